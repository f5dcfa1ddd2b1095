#!/bin/bash
# round 4, first GPU call: correctness of the new ConvT5 kernels + everything else, then their micro-benchmark and a bench line
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/r1_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r1_pytest.log
tail -40 gpurun_out/r1_pytest.log
{
  timeout 200 python tools/kb_convt_out.py 512 3
  SRLZ_OS_FWD_ROWS=8 SRLZ_OS_BWD_ROWS=8 timeout 200 python tools/kb_convt_out.py 512 3
  SRLZ_OS_FWD_ROWS=28 SRLZ_OS_BWD_ROWS=28 timeout 200 python tools/kb_convt_out.py 512 3
  SRLZ_OS_FWD_ROWS=56 SRLZ_OS_BWD_ROWS=56 timeout 200 python tools/kb_convt_out.py 512 3
  timeout 200 python tools/kb_convt_out.py 256 6
} > gpurun_out/r1_kb.log 2>&1
cat gpurun_out/r1_kb.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r1_bench.json 2> gpurun_out/r1_bench.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r1_bench.json").read().strip().splitlines()[-1])
    print("bench ms/step", d["ms_per_step"], "value", d["value"])
    print(json.dumps(d.get("north_star"), indent=0)[:1500])
    for k, v in d["roofline"]["by_symbol"].items():
        print(k, v)
except Exception as e:
    print("bench parse failed", e)
    print(open("gpurun_out/r1_bench.err").read()[-2000:])
PY
