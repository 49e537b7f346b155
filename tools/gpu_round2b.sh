#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
CS=$PWD/swiftllm_amd/csrc
echo "== tests (default build)"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_reference.py -m gpu -q -x -k "paged or golden or oracle_model or mixed or gemm or packed or splitk or reference_triton or tiny" > gpurun_out/pytest_b.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/pytest_b.log | cut -c1-300
: > gpurun_out/ab.jsonl
for tag in default r01pa r01pa_sload; do
  if [ "$tag" = default ]; then unset SWIFTLLM_HIP_LIB; else export SWIFTLLM_HIP_LIB=$CS/libswiftllm_hip_$tag.so; fi
  for args in "--shape c3" "--shape c3 --qkv 4" "--shape c4" "--shape c2 --qkv 4"; do
    timeout 300 python tools/paged_attn_micro.py $args --iters 256 2>/dev/null | tail -1 >> gpurun_out/ab.jsonl
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/ab.jsonl"):
    d=json.loads(l); print(d["lib"].split("hip_")[-1][:14].ljust(14), d["shape"], "qkv", d["qkv_slabs"], "us", d["us_per_op"], "GB/s", d["GBps"])
PY
for tag in default r01gemm; do
  if [ "$tag" = default ]; then unset SWIFTLLM_HIP_LIB; else export SWIFTLLM_HIP_LIB=$CS/libswiftllm_hip_$tag.so; fi
  echo "== gemm micro $tag"
  timeout 300 python tools/gemm_micro.py --m 32 --iters 200 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['shape'], {k:v for k,v in d.items() if k.startswith('swl_ks0') or k.startswith('blas_us')})
"
  timeout 300 python tools/gemm_silu_micro.py 2>/dev/null | tail -2 | cut -c1-400
  timeout 600 python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > gpurun_out/bench_b_$tag.log
  python -c "
import json
d=json.loads(open('gpurun_out/bench_b_$tag.log').read())
print('bench $tag: ms/step', d['ms_per_step'], 'frac', d['step_roofline']['frac'], 'attn', d.get('roofline_paged_attention',{}).get('us_per_launch'), 'gemm', d['roofline']['us_per_launch'])
"
done
