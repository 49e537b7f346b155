#!/bin/bash
# tools/gpu_r05a.sh — r05: gemm_rows / norm-on-the-fly micro (new small-batch decode kernels), A/B variants of the NF staging.
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in "" nf_last nf_4w nf_rd3; do
  echo "== variant '${v}'"
  if [ -n "$v" ]; then export SWIFTLLM_HIP_LIB=$PWD/swiftllm_amd/csrc/libswiftllm_hip_${v}.so; else unset SWIFTLLM_HIP_LIB; fi
  timeout 600 python tools/gemm_rows_micro.py --layer --m 1,8,16 > gpurun_out/r05c_rows_layer_micro_${v:-default}.jsonl 2> gpurun_out/r05c.err; echo "rc=$?"
  python - <<PY
import json
for l in open("gpurun_out/r05c_rows_layer_micro_${v:-default}.jsonl"):
    r=json.loads(l); print({k:r[k] for k in ("M","qkv_slabs_bit_equal","old_graph_us","rows_graph_us","qkv_nf_us","qkv_splitk_us","silu_nf_us","silu_plain_us","down_rows_us","o_rows_us") if k in r}, r.get("tiny_graph_us"))
PY
  grep -v amdgpu gpurun_out/r05c.err | tail -5
done
