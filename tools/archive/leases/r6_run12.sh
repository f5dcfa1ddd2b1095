#!/bin/bash
# round 6, GPU call 12: why is the VAE step 0.6 ms slower than the AE step at bs = 32 (2.95 against 2.36 ms; 1 % at bs = 256)? kernel tables of both
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
bash tools/prof_quick.sh --batch-size 32 --losses vae 2>&1 | sed -n 1,60p > gpurun_out/r6_run12_vae32.txt
cp gpurun_out/ks.csv gpurun_out/r6_run12_vae32_ks.csv
bash tools/prof_quick.sh --batch-size 32 2>&1 | tail -n 3 > gpurun_out/r6_run12_ae32.txt
cp gpurun_out/ks.csv gpurun_out/r6_run12_ae32_ks.csv
python - <<'PY'
import csv
def load(f):
    return {r['Name'].replace('(anonymous namespace)::','').replace('void ','')[:70]: (int(r['Calls']), float(r['TotalDurationNs'])/13/1e3) for r in csv.DictReader(open(f))}
a, v = load('gpurun_out/r6_run12_ae32_ks.csv'), load('gpurun_out/r6_run12_vae32_ks.csv')
rows = []
for k in set(a) | set(v):
    ca, ta = a.get(k, (0, 0.0)); cv, tv = v.get(k, (0, 0.0))
    rows.append((tv - ta, k, ca / 13, cv / 13, ta, tv))
for d, k, ca, cv, ta, tv in sorted(rows, reverse=True)[:22]:
    print('%+8.1f us/step  %-70s calls/step %5.1f -> %5.1f   %7.1f -> %7.1f' % (d, k, ca, cv, ta, tv))
print('sum', sum(t for _, t in a.values()), sum(t for _, t in v.values()))
PY
