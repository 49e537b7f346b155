#!/bin/bash
# r02-f: how far is the decode attention kernel from its memory-only time? (SWL_PA_PROBE_NO_MATH build)
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
CS=$PWD/swiftllm_amd/csrc
: > gpurun_out/attn_nomath.jsonl
for tag in default nomath; do
  if [ "$tag" = default ]; then unset SWIFTLLM_HIP_LIB; else export SWIFTLLM_HIP_LIB=$CS/libswiftllm_hip_$tag.so; fi
  for args in "--shape c3" "--shape c3 --qkv 4" "--shape c3_b128" "--shape c4" "--shape long"; do
    timeout 300 python tools/paged_attn_micro.py $args --iters 200 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); d['lib']='$tag'; print(json.dumps(d))" >> gpurun_out/attn_nomath.jsonl
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/attn_nomath.jsonl"):
    d=json.loads(l); print(d["lib"].ljust(8), d["shape"].ljust(8), "qkv", d.get("qkv_slabs"), "us", d["us_per_op"], "GB/s", d["GBps"])
PY
