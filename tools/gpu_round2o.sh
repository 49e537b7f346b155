#!/bin/bash
# r02-o: tiny-batch path without the phase-2 launch (o_proj merges the attention partials)
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
echo "== tests"
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_parity_fullwidth.py -m gpu -q -x -k "tiny or configs1 or in_workgroup" > gpurun_out/pytest_o.log 2>&1; echo "rc=$?"; tail -6 gpurun_out/pytest_o.log | cut -c1-400
for b in 1 2; do
for a in "" "--no-hip-graph"; do timeout 300 python bench.py --batch $b --skip-prefill --steps 48 --warmup 8 --no-cpu-baseline --no-extras $a 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(\"batch$b $a: ms/step\", d[\"ms_per_step\"], \"frac\", d[\"step_roofline\"][\"frac\"])"; done; done
