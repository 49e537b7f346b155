#!/bin/bash
# prefill attention A/B of the working tree's kernel against a variant library libswiftllm_hip_base.so (built with
# `python -m swiftllm_amd.csrc.build --tag base --swap prefill_attn.hip=<file>`): parity tests on the tree, then interleaved rounds.
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_prefill_long.py -q -x -k "prefill" --timeout=600 2>&1 | tail -1
: > gpurun_out/prefill_variant_ab.jsonl
for round in 1 2 3; do
  for shape in c3 mid ragged c4; do
    for v in tree base; do
      lib=$PWD/swiftllm_amd/csrc/libswiftllm_hip.so; [ $v = base ] && lib=$PWD/swiftllm_amd/csrc/libswiftllm_hip_base.so
      SWIFTLLM_HIP_LIB=$lib timeout 300 python tools/prefill_attn_micro.py --shape $shape --iters 20 2>/dev/null | sed "s/^{/{\"variant\": \"$v\", \"round\": $round, /" >> gpurun_out/prefill_variant_ab.jsonl
    done
  done
done
python - <<'P'
import json, collections
acc = collections.defaultdict(list)
for l in open("gpurun_out/prefill_variant_ab.jsonl"):
    d = json.loads(l); acc[(d["shape"], d["variant"])].append(d["TFLOPs"])
for k in sorted(acc): print(k, acc[k])
P
