#!/bin/bash
# prefill attention A/B: per-row lazy running maximum (tree) vs the exact running maximum (libswiftllm_hip_prelazy.so):
# parity tests, error against fp64 for both, then throughput in interleaved rounds.
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_prefill_long.py -q -x -k "prefill" --timeout=600 > gpurun_out/prefill_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/prefill_tests.log
: > gpurun_out/prefill_lazy_ab.jsonl
for v in lazy prelazy; do
  lib=$PWD/swiftllm_amd/csrc/libswiftllm_hip.so; [ $v = prelazy ] && lib=$PWD/swiftllm_amd/csrc/libswiftllm_hip_prelazy.so
  SWIFTLLM_HIP_LIB=$lib timeout 300 python tools/prefill_attn_err.py 2>/dev/null | sed "s/^{/{\"variant\": \"$v\", /" | tee -a gpurun_out/prefill_lazy_ab.jsonl
done
for round in 1 2 3; do
  for shape in c3 mid c4 ragged; do
    for v in lazy prelazy; do
      lib=$PWD/swiftllm_amd/csrc/libswiftllm_hip.so; [ $v = prelazy ] && lib=$PWD/swiftllm_amd/csrc/libswiftllm_hip_prelazy.so
      SWIFTLLM_HIP_LIB=$lib timeout 300 python tools/prefill_attn_micro.py --shape $shape --iters 20 2>/dev/null | sed "s/^{/{\"variant\": \"$v\", \"round\": $round, /" >> gpurun_out/prefill_lazy_ab.jsonl
    done
  done
done
python - <<'P'
import json, collections
acc = collections.defaultdict(list)
for l in open("gpurun_out/prefill_lazy_ab.jsonl"):
    d = json.loads(l)
    if "TFLOPs" in d: acc[(d["shape"], d["variant"])].append(d["TFLOPs"])
for k in sorted(acc): print(k, acc[k])
P
