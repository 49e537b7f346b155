#!/bin/bash
# A/B: SiLU-gate GEMM (norm on the fly, M = 32) as 224 four-wave workgroups vs 448 two-wave ones (SWL_SILU_WAVES=2).
mkdir -p gpurun_out; export TMPDIR=/tmp
SWL_SILU_WAVES=2 timeout 600 python -m pytest tests/test_gpu_rows.py -q -x --timeout=600 2>&1 | tail -2
: > gpurun_out/silu_waves_ab.jsonl
for round in 1 2 3 4; do
  for v in 4 2; do
    for m in 32 8 1; do
      SWL_SILU_WAVES=$v timeout 120 python tools/gemm_silu_micro.py --nf --m $m --iters 128 2>/dev/null | sed "s/^{/{\"waves\": $v, \"round\": $round, /" >> gpurun_out/silu_waves_ab.jsonl
    done
  done
done
python - <<'P'
import json, collections
acc = collections.defaultdict(list)
for l in open("gpurun_out/silu_waves_ab.jsonl"):
    d = json.loads(l); acc[(d["M"], d["waves"])].append(d["us"])
for k in sorted(acc): print(k, acc[k])
P
