#!/bin/bash
# tools/gpu_check.sh — what one gpurun call executes: GPU parity tests, smoke, a short bench.
# Everything is logged under gpurun_out/ (merged back into the build container).
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== rocminfo" ; rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4
nproc; free -g | head -2
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 ${PYTEST_ARGS:--x} > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -40 gpurun_out/pytest_gpu.log
echo "== smoke"
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -5 gpurun_out/smoke.log
echo "== bench tiny"
timeout 300 python bench.py --model tiny --batch 4 --prompt-len 64 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/bench_tiny.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/bench_tiny.log
echo "== bench"
timeout 900 python bench.py ${BENCH_ARGS:-} > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -5 gpurun_out/bench.log
