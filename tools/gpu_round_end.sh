#!/bin/bash
# tools/gpu_round_end.sh — what one end-of-round gpurun call executes: the full GPU suite, smoke, the bench in the driver's form,
# kernel traces of configs[2] / configs[1] (logs under gpurun_out/; ~15 GPU-minutes).
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 --durations=30 > gpurun_out/round_end_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -45 gpurun_out/round_end_pytest_gpu.log | cut -c1-250
timeout 300 python __graft_entry__.py smoke > gpurun_out/round_end_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/round_end_smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/round_end_bench_driver_form.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/round_end_bench_driver_form.log | cut -c1-600
TRACES="c2 c1" bash tools/gpu_trace.sh 2>&1 | grep -v "^W2026\|amdgpu.ids" | grep -E "rocprof rc|ring_kernel|gemm_rows|paged_attn|splitk_add|rmsnorm_kernel|total kernel" | cut -c1-220
