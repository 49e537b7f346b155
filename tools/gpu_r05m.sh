#!/bin/bash
# r05 end of round: full GPU suite, smoke, default bench (driver form), kernel traces of configs[2] / configs[1].
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 --durations=30 > gpurun_out/r05m_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -45 gpurun_out/r05m_pytest_gpu.log | cut -c1-250
timeout 300 python __graft_entry__.py smoke > gpurun_out/r05m_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r05m_smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05m_bench_driver_form.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r05m_bench_driver_form.log | cut -c1-600
TRACES="c2 c1" bash tools/gpu_trace.sh 2>&1 | grep -v "^W2026\|amdgpu.ids" | grep -E "rocprof rc|ring_kernel|gemm_rows|paged_attn|splitk_add|rmsnorm_kernel|total kernel" | cut -c1-220
