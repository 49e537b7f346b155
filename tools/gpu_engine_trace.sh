#!/bin/bash
# tools/gpu_engine_trace.sh — one gpurun call: engine parity tests (watchdog), then the phase anatomy of the persistent decode
# step with the loader thinned during sweeps (flags 1) and not (flags 0).
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_engine.py -q -x --timeout=600 ${PYTEST_ARGS:-} > gpurun_out/engine_tests.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/engine_tests.log
for f in 0; do
  SWL_ENGINE_FLAGS=$f timeout 600 python tools/engine_trace.py --layers 32 > gpurun_out/engine_trace_f$f.json 2> gpurun_out/engine_trace_f$f.err; echo rc=$?
  cat gpurun_out/engine_trace_f$f.json; tail -2 gpurun_out/engine_trace_f$f.err
done
