#!/bin/bash
# tools/gpu_profile.sh — rocprofv3 kernel trace of a short bench run + the default bench line.
mkdir -p gpurun_out
REPO=$(pwd)
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof -o bench -- python $REPO/bench.py --steps 24 --warmup 4 --no-cpu-baseline ${PROF_BENCH_ARGS:-} > $REPO/gpurun_out/prof_bench.log 2>&1
echo "rocprof rc=$?"; tail -2 $REPO/gpurun_out/prof_bench.log | cut -c1-400
find $REPO/gpurun_out/prof -name "*stats*" | head
cd $REPO
if [ -n "$RUN_DEFAULT_BENCH" ]; then
  timeout 1200 python bench.py > gpurun_out/bench_default.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench_default.log
fi
