import json,sys
for f in sys.argv[1:]:
    d=json.loads(open(f).read().strip().splitlines()[-1])
    L=d["roofline"]["layers"]
    print(f, d["ms_per_step"], d["roofline"]["achieved"], "conv2 fwd %.1f dgrad %.1f convT4 fwd %.1f" % (L["conv s1 56x56->56x56 fwd"]["avg_us"], L["conv s1 56x56->56x56 dgrad"]["avg_us"], L["convT s2 55x55->111x111 fwd"]["avg_us"]))
