#!/bin/bash
# r02-m: three-wave workgroups for the qkv projection
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
echo "== tests"
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -x -k "gemm or packed or splitk or skinny or golden or full_width or tiny or qkv_slabs" > gpurun_out/pytest_m.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/pytest_m.log | cut -c1-300
echo "== micro"
timeout 300 python tools/gemm_wgk_micro.py --shapes qkv --ms 32,1 2>&1 | grep -v amdgpu | cut -c1-300
echo "== bench"
for i in 1 2; do timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('value', d['value'], 'ms/step', d['ms_per_step'], 'frac', d['step_roofline']['frac'])"; done
