mkdir -p gpurun_out; export TMPDIR=/tmp
: > gpurun_out/wide_pipe_ab.jsonl


for round in 1 2; do
for ts in 1 2; do
  for v in product gwold gwplain; do
    LIB=$(pwd)/swiftllm_amd/csrc/libswiftllm_hip_$v.so
    [ $v = product ] && LIB=$(pwd)/swiftllm_amd/csrc/libswiftllm_hip.so
    [ $v = gwold ] && [ $ts = 2 ] && continue
    SWL_WIDE_TS=$ts SWIFTLLM_HIP_LIB=$LIB timeout 300 python tools/gemm_wide_micro.py --m 96,128,192,256 --shapes ${SHAPES:-qkv,o,up_gate,down,lm_head} --auto-only --iters 30 2>/dev/null >> gpurun_out/wide_pipe_ab.jsonl
  done
done
done
python - <<'P'
import json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list)); sha = collections.defaultdict(set)
for l in open("gpurun_out/wide_pipe_ab.jsonl"):
    d = json.loads(l)
    form = d["lib"].replace("libswiftllm_hip_gw", "").replace("libswiftllm_hip", "pipe").replace(".so", "") + "-ts" + d["ts"]
    for k in ("auto_us", "silu_w0_us"):
        if k in d: acc[(d["shape"], k, d["M"])][form].append(d[k])
    for k in ("auto_sha", "silu_w0_sha"):
        if k in d: sha[(d["shape"], k, d["M"])].add(d[k])
for k in sorted(acc):
    print(k, {f: round(sum(v)/len(v), 1) for f, v in acc[k].items()})
bad = {k: v for k, v in sha.items() if len(v) > 1}
print("digests agree across forms" if not bad else f"DIGEST MISMATCH {bad}")
P
