#!/bin/bash
# A/B of gemm_wide.hip's register tiling (r06d): TS = 1 (a wave = 32 rows of W x all tokens, one LDS read per MFMA) vs TS = 2
# (a wave = 64 rows x half the tokens, one LDS read per two MFMAs), the latter also with plain instead of non-temporal W loads
# (variant library csrc/libswiftllm_hip_wideplain.so, built beforehand:
#   sed 's/load8_nt(wsrc + rb_/load8(wsrc + rb_/' swiftllm_amd/csrc/gemm_wide.hip > /tmp/gemm_wide_plain.hip
#   python -m swiftllm_amd.csrc.build --tag wideplain --swap gemm_wide.hip=/tmp/gemm_wide_plain.hip).
# The kernel tests under TS = 2 first, then tools/gemm_wide_micro.py in each form, two rounds interleaved; the digests of the
# results must agree across forms. TRACES=1 also takes kernel traces of decode batches 128 / 256 (product form).
mkdir -p gpurun_out; export TMPDIR=/tmp
SWL_WIDE_TS=2 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "wide" --timeout=600 > gpurun_out/wide_ts2_tests.log 2>&1; echo "wide tests (TS=2) rc=$?"; tail -3 gpurun_out/wide_ts2_tests.log
: > gpurun_out/wide_ts_ab.jsonl
PLAIN=$(pwd)/swiftllm_amd/csrc/libswiftllm_hip_wideplain.so
for round in 1 2; do
  SWL_WIDE_TS=1 timeout 300 python tools/gemm_wide_micro.py --m ${MS:-96,128,192,256} --auto-only --iters 40 2>/dev/null >> gpurun_out/wide_ts_ab.jsonl
  SWL_WIDE_TS=2 timeout 300 python tools/gemm_wide_micro.py --m ${MS:-96,128,192,256} --auto-only --iters 40 2>/dev/null >> gpurun_out/wide_ts_ab.jsonl
  [ -f $PLAIN ] && SWL_WIDE_TS=2 SWIFTLLM_HIP_LIB=$PLAIN timeout 300 python tools/gemm_wide_micro.py --m ${MS:-96,128,192,256} --auto-only --iters 40 2>/dev/null >> gpurun_out/wide_ts_ab.jsonl
done
python - <<'P'
import json, collections
acc, sha = collections.defaultdict(list), collections.defaultdict(set)
for l in open("gpurun_out/wide_ts_ab.jsonl"):
    d = json.loads(l)
    form = f"ts{d['ts']}" + ("" if d["lib"] == "product" else "-plain")
    for k in ("auto_us", "silu_w0_us"):
        if k in d:
            acc[(d["shape"], d["M"], k, form)].append(d[k])
    for k in ("auto_sha", "silu_w0_sha"):
        if k in d:
            sha[(d["shape"], d["M"], k)].add(d[k])
for k in sorted(acc):
    print(k, acc[k])
bad = {k: v for k, v in sha.items() if len(v) > 1}
print("digests agree across forms" if not bad else f"DIGEST MISMATCH {bad}")
P
if [ -n "$TRACES" ]; then TRACES="b128 b256" bash tools/gpu_trace.sh 2>&1 | grep -v "^W2026\|amdgpu.ids" | cut -c1-200 | head -40; fi
