#!/bin/bash
# r06d: the slab-fed paged-attention kernel (rotary + KV store in its prologue) at decode batches beyond 64 as 8-wave workgroups
# (one per CU: every workgroup's prologue stalls the CU's stream) vs 4-wave workgroups (two per CU) — variant library
# csrc/libswiftllm_hip_paw4.so: sed 's/    if (p.seq_block_size >= 32 \* kBlk)$/    if (p.seq_block_size >= 32 * kBlk \&\& Bd <= 64)/'
# swiftllm_amd/csrc/paged_attn.hip > /tmp/pa_w4.hip; python -m swiftllm_amd.csrc.build --tag paw4 --swap paged_attn.hip=/tmp/pa_w4.hip
mkdir -p gpurun_out; export TMPDIR=/tmp
W4=$(pwd)/swiftllm_amd/csrc/libswiftllm_hip_paw4.so
: > gpurun_out/paged_attn_waves_ab.jsonl
for round in 1 2 3; do
  for shape in c3_b128 c3_b256 c3_b64; do
    for q in 4 0; do
      timeout 200 python tools/paged_attn_micro.py --shape $shape --qkv $q 2>/dev/null | grep '^{' >> gpurun_out/paged_attn_waves_ab.jsonl
      SWIFTLLM_HIP_LIB=$W4 timeout 200 python tools/paged_attn_micro.py --shape $shape --qkv $q 2>/dev/null | grep '^{' >> gpurun_out/paged_attn_waves_ab.jsonl
    done
  done
done
python - <<'P'
import json, collections
acc = collections.defaultdict(list)
for l in open("gpurun_out/paged_attn_waves_ab.jsonl"):
    d = json.loads(l)
    us = d.get("us_per_launch") or d.get("us")
    acc[(d["shape"], d["qkv_slabs"], "w4" if "paw4" in d["lib"] else "w8")].append(us)
for k in sorted(acc): print(k, acc[k])
P
