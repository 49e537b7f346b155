#!/bin/bash
# r05 debugging: one-off stall inside the 20-step timed region of bench.py — interpreter GC?
mkdir -p gpurun_out
export TMPDIR=/tmp
for i in 1 2 3 4; do
  for keep in 1 0; do
    SWL_BENCH_KEEP_GC=$keep timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-reference > gpurun_out/r05h_bench_gc${keep}_$i.log 2>&1
    echo "keep_gc=$keep run $i: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r05h_bench_gc${keep}_$i.log | head -1) $(grep -o '"timed_region": {[^}]*}' gpurun_out/r05h_bench_gc${keep}_$i.log)"
  done
done
