#!/bin/bash
# A/B of a prefill-attention variant library (tools/probe/prefill_attn_v*.hip built with `build.py --tag <t> --swap ...`)
# against the product kernel: micro at three shapes + the kernel's parity tests on the variant.
mkdir -p gpurun_out; export TMPDIR=/tmp
for lib in "" ${VARIANTS:-pa2}; do
  if [ -n "$lib" ]; then export SWIFTLLM_HIP_LIB=$PWD/swiftllm_amd/csrc/libswiftllm_hip_${lib}.so; else unset SWIFTLLM_HIP_LIB; fi
  for sh in c3 mid c4 ragged; do
    echo "lib=${lib:-product} $(python tools/prefill_attn_micro.py --shape $sh 2>/dev/null | tail -1)"
  done
done | tee gpurun_out/prefill_ab.jsonl
for lib in ${VARIANTS:-pa2}; do
  SWIFTLLM_HIP_LIB=$PWD/swiftllm_amd/csrc/libswiftllm_hip_${lib}.so timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "prefill" 2>&1 | tail -2
done
