#!/bin/bash
# r06d: the re-measured wide-vs-library table of kernels/route_tune.py (qkv / o at 129..192 tokens) against the r04 table, in the
# decode STEP: bench.py decode-only at the given batches, both tables in the same call (the r04 table is patched in by a launcher).
mkdir -p gpurun_out; export TMPDIR=/tmp
cat > /tmp/bench_r04_table.py <<'P'
import os, runpy, sys
sys.path.insert(0, os.getcwd())
from swiftllm_amd.worker.kernels import route_tune as R
def r04_table(m, n, k):
    if k >= 2 * n: return True
    if n > 8192: return False
    if m <= 128 or 160 < m <= 192: return True
    return m > 192 and n > 4096
R.table_wide_wins = r04_table
from swiftllm_amd.worker.kernels import linear as L
L._wide_wins = r04_table
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path("bench.py", run_name="__main__")
P
: > gpurun_out/route_step_ab.jsonl
for round in 1 2; do
  for b in ${BATCHES:-144 160}; do
    for v in r06d r04; do
      LAUNCH="bench.py"; [ $v = r04 ] && LAUNCH="/tmp/bench_r04_table.py"
      timeout 400 python $LAUNCH --batch $b --skip-prefill --steps 24 --warmup 6 --kv-placement bottom --kv-blocks 24000 --no-cpu-baseline --no-extras --no-reference 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps(dict(table='$v', batch=$b, round=$round, ms_per_step=d['ms_per_step'], frac=d['step_roofline']['frac'])))" >> gpurun_out/route_step_ab.jsonl
    done
  done
done
cat gpurun_out/route_step_ab.jsonl
