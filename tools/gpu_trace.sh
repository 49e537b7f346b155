#!/bin/bash
# tools/gpu_trace.sh — rocprofv3 kernel trace of a short default bench; prints the per-kernel table.
mkdir -p gpurun_out
REPO=$(pwd)
export TMPDIR=/tmp
cd /tmp
rm -rf $REPO/gpurun_out/prof
timeout 900 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof -o bench -- python $REPO/bench.py --steps 24 --warmup 4 --no-cpu-baseline ${TRACE_BENCH_ARGS:-} > $REPO/gpurun_out/prof_bench.log 2>&1
echo "rocprof rc=$?"; tail -1 $REPO/gpurun_out/prof_bench.log | cut -c1-300
cd $REPO
DB=$(find gpurun_out/prof -name "*.db" | head -1); python tools/rocpd_stats.py $DB 24 > gpurun_out/prof_stats.md 2>&1; cat gpurun_out/prof_stats.md
