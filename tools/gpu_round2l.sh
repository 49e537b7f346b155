#!/bin/bash
# r02-l: end-of-round evidence: kernel trace of the default bench, PMC traffic of the slab-fed matrix-core attention kernel
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$(pwd)
cd /tmp; rm -rf $R/gpurun_out/prof
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o bench -- python $R/bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_bench.log 2>&1
echo "rocprof rc=$?"; tail -1 $R/gpurun_out/prof_bench.log | cut -c1-200
cd $R
DB=$(find gpurun_out/prof -name "*.db" | head -1); python tools/rocpd_stats.py $DB 24 > gpurun_out/prof_stats.md 2>&1; head -26 gpurun_out/prof_stats.md | cut -c1-170
rm -rf gpurun_out/prof
mkdir -p gpurun_out/pmc; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc/$c
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pmc/$c -o pa -- python $R/tools/paged_attn_micro.py --shape c3 --qkv 4 --iters 64 > $R/gpurun_out/pmc/$c.log 2>&1; echo "$c rc=$?"
  DB=$(find $R/gpurun_out/pmc/$c -name "*.db" | head -1); python $R/tools/rocpd_pmc.py $DB paged_attn 2>&1 | tail -3
  rm -rf $R/gpurun_out/pmc/$c
done
