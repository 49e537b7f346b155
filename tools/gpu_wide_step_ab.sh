#!/bin/bash
# Does a gemm_wide.hip change show in the decode STEP? bench.py decode-only at large batches with the product library and with
# a variant built from another revision of the file (default: csrc/libswiftllm_hip_gwold.so =
#   git show <rev>:swiftllm_amd/csrc/gemm_wide.hip > /tmp/gw_old.hip
#   python -m swiftllm_amd.csrc.build --tag gwold --swap gemm_wide.hip=/tmp/gw_old.hip), interleaved, same box.
mkdir -p gpurun_out; export TMPDIR=/tmp
OLD=$(pwd)/swiftllm_amd/csrc/libswiftllm_hip_${VARIANT:-gwold}.so
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_parity_largebatch.py -q -x -k "wide or large" --timeout=600 > gpurun_out/wide_step_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/wide_step_tests.log
: > gpurun_out/wide_step_ab.jsonl
for round in 1 2; do
  for b in ${BATCHES:-128 192}; do
    for v in product old; do
      LIB=$(pwd)/swiftllm_amd/csrc/libswiftllm_hip.so; [ $v = old ] && LIB=$OLD
      SWIFTLLM_HIP_LIB=$LIB timeout 400 python bench.py --batch $b --skip-prefill --steps 24 --warmup 6 --kv-placement bottom --kv-blocks 24000 --no-cpu-baseline --no-extras --no-reference 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps(dict(lib='$v', batch=$b, round=$round, ms_per_step=d['ms_per_step'], frac=d['step_roofline']['frac'])))" >> gpurun_out/wide_step_ab.jsonl
    done
  done
done
cat gpurun_out/wide_step_ab.jsonl
