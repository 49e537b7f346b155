#!/bin/bash
# full GPU suite + smoke, logged
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 --durations=40 > gpurun_out/r05i_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -45 gpurun_out/r05i_pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/r05i_smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/r05i_smoke.log
