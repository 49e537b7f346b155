#!/bin/bash
# r02-i: kernel traces of the side configs: configs[3] (Llama-2-7B dims, 4 x 16k) and configs[1] (batch 1)
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
REPO=$(pwd)
for cfg in "c3 --model llama2-7b --batch 4 --prompt-len 16384 --gen-len 64 --skip-prefill --no-extras --steps 16 --warmup 4" "c1 --batch 1 --skip-prefill --no-extras --steps 32 --warmup 8"; do
  set -- $cfg; tag=$1; shift
  cd /tmp; rm -rf $REPO/gpurun_out/prof_$tag
  timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_$tag -o bench -- python $REPO/bench.py "$@" --no-cpu-baseline > $REPO/gpurun_out/prof_${tag}_bench.log 2>&1
  echo "$tag rocprof rc=$?"; tail -1 $REPO/gpurun_out/prof_${tag}_bench.log | cut -c1-200
  cd $REPO
  DB=$(find gpurun_out/prof_$tag -name "*.db" | head -1); python tools/rocpd_stats.py $DB 16 > gpurun_out/prof_${tag}_stats.md 2>&1; head -14 gpurun_out/prof_${tag}_stats.md | cut -c1-170
  rm -rf gpurun_out/prof_$tag
done
