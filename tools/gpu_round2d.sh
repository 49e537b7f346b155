#!/bin/bash
# r02-d: where the round stands — full bench line (extras + cpu baseline) and the whole GPU test tier.
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
echo "== bench (driver form)"
S=$(date +%s)
timeout 900 python bench.py --steps 20 --warmup 5 2> gpurun_out/bench_d.err | tail -1 > gpurun_out/bench_d.json
echo "rc=$? wall=$(( $(date +%s) - S ))s"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_d.json').read())
print('value', d['value'], 'ms/step', d['ms_per_step'], 'frac', d['step_roofline']['frac'])
for k,v in d.items():
    if k.startswith('roofline') or k in ('extra','extras','cpu_baseline','config'):
        print(k, json.dumps(v)[:900])
PY
echo "== tests"
S=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_d.log 2>&1; echo "rc=$? wall=$(( $(date +%s) - S ))s"; tail -4 gpurun_out/pytest_d.log | cut -c1-300
