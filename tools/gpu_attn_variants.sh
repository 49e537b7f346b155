#!/bin/bash
# tools/gpu_attn_variants.sh — A/B the paged-attention build variants (python -m swiftllm_amd.csrc.build --tag T -D ...)
# on the GPU box: parity subset on the default build, then the attention micro-benchmark and a short bench per variant.
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
CS=swiftllm_amd/csrc
echo "== parity subset (default build)"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -x -k "paged or golden or oracle_model or mixed" > gpurun_out/pytest_attn.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/pytest_attn.log
: > gpurun_out/attn_variants.jsonl
for tag in ${VARIANTS:-default d3 la1 la2 d3la1}; do
  if [ "$tag" = default ]; then unset SWIFTLLM_HIP_LIB; else export SWIFTLLM_HIP_LIB=$PWD/$CS/libswiftllm_hip_$tag.so; fi
  echo "== variant $tag"
  for args in "--shape c3" "--shape c3 --qkv 4" "--shape c2 --qkv 4" "--shape c4" "--shape c3_b128 --qkv 4"; do
    timeout 300 python tools/paged_attn_micro.py $args --iters 256 2>/dev/null | tail -1 | tee -a gpurun_out/attn_variants.jsonl | cut -c1-330
  done
  timeout 600 python bench.py --steps 30 --warmup 6 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_attn_$tag.log
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_attn_$tag.log").read())
print("bench $tag: ms/step", d["ms_per_step"], "tok/s", d["value"], "frac", d["step_roofline"]["frac"])
PY
done
