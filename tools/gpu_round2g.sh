#!/bin/bash
# r02-g: matrix-core paged attention — parity tests, then the kernel alone at ring depth 2/3/4
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
CS=$PWD/swiftllm_amd/csrc
echo "== tests"
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -x -k "paged or golden or oracle_model or mixed or split or full_width" > gpurun_out/pytest_g.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/pytest_g.log | cut -c1-400
: > gpurun_out/attn_mfma.jsonl
for tag in default mf2 mf4; do
  if [ "$tag" = default ]; then unset SWIFTLLM_HIP_LIB; else export SWIFTLLM_HIP_LIB=$CS/libswiftllm_hip_$tag.so; fi
  for args in "--shape c3" "--shape c3 --qkv 4" "--shape c3_b128" "--shape long" "--shape c2"; do
    timeout 300 python tools/paged_attn_micro.py $args --iters 200 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); d['lib']='$tag'; print(json.dumps(d))" >> gpurun_out/attn_mfma.jsonl
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/attn_mfma.jsonl"):
    d=json.loads(l); print(d["lib"].ljust(8), d["shape"].ljust(8), "qkv", d.get("qkv_slabs"), "us", d["us_per_op"], "GB/s", d["GBps"])
PY
