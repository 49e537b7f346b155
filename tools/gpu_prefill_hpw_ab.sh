#!/bin/bash
# prefill attention A/B: the q-heads of a kv-head side by side in a workgroup (default) vs 128-row blocks of one head
# (SWL_PREFILL_HPW=1); parity tests under both, then interleaved rounds.
mkdir -p gpurun_out; export TMPDIR=/tmp
for h in 4 1; do
  SWL_PREFILL_HPW=$h timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_prefill_long.py -q -x -k "prefill" --timeout=600 2>&1 | tail -1
done
: > gpurun_out/prefill_hpw_ab.jsonl
for round in 1 2 3; do
  for shape in c3 mid ragged c4; do
    for h in 4 2 1; do
      SWL_PREFILL_HPW=$h timeout 300 python tools/prefill_attn_micro.py --shape $shape --iters 20 2>/dev/null | sed "s/^{/{\"hpw\": $h, \"round\": $round, /" >> gpurun_out/prefill_hpw_ab.jsonl
    done
  done
done
python - <<'P'
import json, collections
acc = collections.defaultdict(list)
for l in open("gpurun_out/prefill_hpw_ab.jsonl"):
    d = json.loads(l); acc[(d["shape"], d["hpw"])].append(d["TFLOPs"])
for k in sorted(acc): print(k, acc[k])
P
