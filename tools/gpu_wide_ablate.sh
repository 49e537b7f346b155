#!/bin/bash
# Where does gemm_wide.hip's time go at 128 / 256 tokens? Timing-only ablations (WRONG results) as variant libraries built
# beforehand by `python tools/make_gemm_wide_ablations.py` (csrc/libswiftllm_hip_gw<name>.so):
#   plain   W loads without the non-temporal hint (a real candidate, right results)
#   nox     no global loads of x (staging registers filled from a register)        -> what the x path through L2/L1 costs
#   nolds   B fragments taken from registers instead of LDS                        -> what the LDS reads cost
#   wcache  every W ring slot re-reads tile 0 (cache hits, no HBM stream)          -> what the weight stream costs
#   nomfma  one fma per fragment pair instead of the MFMA                          -> what the matrix cores cost
# each in both register tilings (SWL_WIDE_TS=1|2); one JSON line per (form, shape, M) in gpurun_out/wide_ablate.jsonl.
mkdir -p gpurun_out; export TMPDIR=/tmp
: > gpurun_out/wide_ablate.jsonl
for ts in 1 2; do
  for v in product plain nox nolds wcache nomfma; do
    LIB=$(pwd)/swiftllm_amd/csrc/libswiftllm_hip_gw$v.so
    [ $v = product ] && LIB=$(pwd)/swiftllm_amd/csrc/libswiftllm_hip.so
    [ -f $LIB ] || continue
    SWL_WIDE_TS=$ts SWIFTLLM_HIP_LIB=$LIB timeout 300 python tools/gemm_wide_micro.py --m ${MS:-128,256} --shapes ${SHAPES:-qkv,o,up_gate,down} --auto-only --iters 30 2>/dev/null >> gpurun_out/wide_ablate.jsonl
  done
done
python - <<'P'
import json, collections
acc = collections.defaultdict(dict)
for l in open("gpurun_out/wide_ablate.jsonl"):
    d = json.loads(l)
    form = d["lib"].replace("libswiftllm_hip_gw", "").replace("libswiftllm_hip", "product").replace(".so", "")
    for k in ("auto_us", "silu_w0_us"):
        if k in d:
            acc[(d["shape"], k, d["M"], "ts" + d["ts"])][form] = d[k]
for k in sorted(acc):
    print(k, acc[k])
P
