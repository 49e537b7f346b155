#!/bin/bash
# attention variants (default = depth by G; d2; d3), then the new parity tests
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
VARIANTS="default d2 d3" bash tools/gpu_attn_variants.sh
unset SWIFTLLM_HIP_LIB
echo "== new parity tests"
timeout 1500 python -m pytest tests/test_gpu_parity_fullwidth.py tests/test_gpu_reference.py -m gpu -q -s > gpurun_out/pytest_parity.log 2>&1; echo "rc=$?"; grep -E "full-width parity|tier-2|passed|failed|Error|FAILED" gpurun_out/pytest_parity.log | cut -c1-900 | tail -30
