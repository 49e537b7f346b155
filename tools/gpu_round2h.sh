#!/bin/bash
# r02-h: matrix-core paged attention with the VALU cross-lane maximum: determinism probes, whole GPU tier, bench
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
echo "== determinism probes"
timeout 300 python tools/probe/dbg_attn_variants2.py 2>&1 | grep "plain kernel\|distinct" | cut -c1-200
timeout 300 python tools/probe/dbg_attn_variants.py 2>&1 | grep stable | cut -c1-200
echo "== tests"
S=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_h.log 2>&1; echo "rc=$? wall=$(( $(date +%s) - S ))s"; tail -4 gpurun_out/pytest_h.log | cut -c1-300
echo "== bench (driver form)"
timeout 900 python bench.py --steps 20 --warmup 5 2> gpurun_out/bench_h.err | tail -1 > gpurun_out/bench_h.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_h.json').read())
print('value', d['value'], 'ms/step', d['ms_per_step'], 'frac', d['step_roofline']['frac'], 'prefill', d.get('prefill_tok_s'))
for k in ('roofline_paged_attention','eager','configs1_batch1','configs3_llama2_7b_4x16k'):
    print(k, json.dumps(d.get(k))[:420])
PY
