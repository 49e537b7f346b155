#!/bin/bash
# tools/gpu_prefill_ab.sh — prefill attention: parity tests on the shipped kernel, then the LDS-DMA kernel (default at head_dim 128)
# against the register-staged one (SWL_PREFILL_ATTN=v1), interleaved rounds in one session.
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "prefill" --timeout=600 > gpurun_out/prefill_tests.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/prefill_tests.log
: > gpurun_out/prefill_attn_ab.jsonl
for round in 1 2 3; do
  for shape in c3 mid c4 ragged; do
    for v in dma v1; do
      if [ $v = v1 ]; then export SWL_PREFILL_ATTN=v1; else unset SWL_PREFILL_ATTN; fi
      timeout 300 python tools/prefill_attn_micro.py --shape $shape --iters 20 | sed "s/^{/{\"variant\": \"$v\", \"round\": $round, /" >> gpurun_out/prefill_attn_ab.jsonl
    done
  done
done
unset SWL_PREFILL_ATTN
cat gpurun_out/prefill_attn_ab.jsonl
