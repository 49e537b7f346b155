#!/bin/bash
# tools/gpu_round.sh — kernel-trace profile of the default bench + BASELINE configs[1]/[3] lines + serving bench.
mkdir -p gpurun_out
REPO=$(pwd)
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof -o bench -- python $REPO/bench.py --steps 24 --warmup 4 --no-cpu-baseline > $REPO/gpurun_out/prof_bench.log 2>&1
echo "rocprof rc=$?"; tail -1 $REPO/gpurun_out/prof_bench.log | cut -c1-300
cd $REPO
DB=$(find gpurun_out/prof -name "*.db" | head -1); python tools/rocpd_stats.py $DB 30 > gpurun_out/prof_stats.md 2>&1; head -30 gpurun_out/prof_stats.md
echo "== C2 batch 1"
timeout 600 python bench.py --batch 1 --steps 60 --warmup 8 --no-cpu-baseline > gpurun_out/bench_c2.log 2>&1; tail -1 gpurun_out/bench_c2.log | cut -c1-400
echo "== C4 llama2-7b 16k x4"
timeout 900 python bench.py --model llama2-7b --batch 4 --prompt-len 16384 --steps 40 --warmup 4 --no-cpu-baseline > gpurun_out/bench_c4.log 2>&1; tail -1 gpurun_out/bench_c4.log | cut -c1-400
echo "== serving"
timeout 900 python tools/serve_bench.py --requests 128 --max-batch 64 > gpurun_out/serve_bench.log 2>&1; grep -v amdgpu gpurun_out/serve_bench.log | tail -3
