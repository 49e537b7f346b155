#!/bin/bash
# r05: targeted re-runs (large-batch parity, decisive all-steps, full-depth asserts, buckets), the attribution diagnostic,
# serving with graph-capture counts, kernel-file durations.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity_largebatch.py tests/test_gpu_parity_decisive.py tests/test_gpu_parity_fulldepth.py "tests/test_gpu_model.py::test_decode_batch_buckets_replay_equals_exact_eager_launches" -q --timeout=900 -s --durations=15 > gpurun_out/r05k_pytest_targeted.log 2>&1; echo "targeted rc=$?"; grep -E "passed|failed|xfail|\[large-batch|\[decisive|\[full-depth" gpurun_out/r05k_pytest_targeted.log | cut -c1-600 | tail -20
SWIFTLLM_PARITY_FULL_CONTROL=1 timeout 1500 python -m pytest tests/test_gpu_parity_attribution.py -q --timeout=1400 -s > gpurun_out/r05k_pytest_attribution.log 2>&1; echo "attribution rc=$?"; grep -E "passed|failed|\[attribution" gpurun_out/r05k_pytest_attribution.log | cut -c1-900 | tail -6
timeout 900 python tools/serve_bench.py --requests 256 --max-batch 64 --rate 40 --passes 2 --modes plain > gpurun_out/r05k_serve_captures.jsonl 2> gpurun_out/r05k_serve.err; echo "serve rc=$?"; cat gpurun_out/r05k_serve_captures.jsonl | cut -c1-700; grep "Model.profile" gpurun_out/r05k_serve.err
timeout 900 python -m pytest tests/test_gpu_kernels.py -q --durations=0 > gpurun_out/r05k_pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -2 gpurun_out/r05k_pytest_kernels.log
