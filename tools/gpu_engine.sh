#!/bin/bash
# tools/gpu_engine.sh — one gpurun call for the persistent one-sequence decode step (csrc/decode_engine.hip):
# its parity tests under a watchdog, then batch-1 bench A/B (engine vs the multi-launch path), everything under `timeout`.
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== engine tests"
timeout 900 python -m pytest tests/test_gpu_engine.py -q -x --timeout=600 ${PYTEST_ARGS:-} > gpurun_out/engine_tests.log 2>&1
echo "pytest rc=$?"; tail -30 gpurun_out/engine_tests.log
echo "== bench batch 1, engine"
timeout 600 python bench.py --batch 1 --steps 40 --warmup 8 --no-cpu-baseline --no-extras --no-reference > gpurun_out/engine_b1.json 2> gpurun_out/engine_b1.err
echo "rc=$?"; tail -c 1500 gpurun_out/engine_b1.json; tail -5 gpurun_out/engine_b1.err
echo "== bench batch 1, multi-launch"
timeout 600 python bench.py --batch 1 --steps 40 --warmup 8 --no-cpu-baseline --no-extras --no-reference --no-decode-engine > gpurun_out/engine_b1_off.json 2> gpurun_out/engine_b1_off.err
echo "rc=$?"; tail -c 1500 gpurun_out/engine_b1_off.json; tail -3 gpurun_out/engine_b1_off.err
