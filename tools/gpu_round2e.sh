#!/bin/bash
# r02-e: phase-2 merge kernel rewrite (batch 1), prefill GEMM solution probe
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
echo "== tests (paged attention, model)"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -x -k "paged or golden or oracle_model or mixed or split" > gpurun_out/pytest_e.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/pytest_e.log | cut -c1-300
echo "== bench batch 1"
timeout 600 python bench.py --batch 1 --skip-prefill --steps 48 --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > gpurun_out/bench_e_b1.json
python -c "
import json
d=json.loads(open('gpurun_out/bench_e_b1.json').read())
print('batch1: ms/step', d['ms_per_step'], 'frac', d['step_roofline']['frac'])
"
echo "== prefill gemm probe"
timeout 600 python tools/prefill_gemm_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/prefill_gemm_probe.jsonl | cut -c1-400
