#!/bin/bash
# A/B of the row-owned projections' x path (csrc/gemm_rows.hip): fragment-shaped x loads (r05) vs x through a wave-private
# LDS tile (r06b). Parity tests first, then the layer micro of tools/gemm_rows_micro.py in both forms, three rounds interleaved.
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_rows.py -q -x --timeout=600 > gpurun_out/rows_tests.log 2>&1; echo "rows tests rc=$?"; tail -4 gpurun_out/rows_tests.log
: > gpurun_out/rows_ab.jsonl
for round in 1 2 3; do
  for v in lds frag; do
    SWL_ROWS_X=$v timeout 300 python tools/gemm_rows_micro.py --layer --m 32,24,16,8,1 --iters 40 2>/dev/null | sed "s/^{/{\"x\": \"$v\", \"round\": $round, /" >> gpurun_out/rows_ab.jsonl
  done
done
python - <<'P'
import json, collections
acc = collections.defaultdict(list)
for l in open("gpurun_out/rows_ab.jsonl"):
    d = json.loads(l)
    for k in ("down_rows_us", "down_splitk_us", "o_rows_us", "o_splitk_us", "rows_graph_us", "old_graph_us"):
        acc[(d["M"], d["x"], k)].append(d[k])
    if d["residual_diff_frac"] or not d["qkv_slabs_bit_equal"]:
        print("MISMATCH", d["M"], d["x"], d["residual_diff_frac"], d["qkv_slabs_bit_equal"])
for k in sorted(acc):
    print(k, acc[k])
P
