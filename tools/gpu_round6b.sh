#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_routes.py -q --timeout=600 -s > gpurun_out/routes_test.log 2>&1; echo "routes rc=$?"; tail -6 gpurun_out/routes_test.log
bash tools/gpu_prefill_ab.sh
