#!/bin/bash
# tools/gpu_trace.sh — rocprofv3 --kernel-trace --stats of short bench runs; prints / saves the per-kernel tables.
#   TRACES="c2 c1 c3" (default): BASELINE configs[2] (batch 32, prompt pass + decode), configs[1] (batch 1), configs[3]
#   (Llama-2-7B dims, 4 x 16k, KV pre-filled); b8: decode-only at batch 8 (the 5-launch layer of r05); b128 / b256: decode-only at batch 128 / 256 (KV pre-filled). Output: gpurun_out/trace_<name>.md (+ the bench line in trace_<name>.log)
mkdir -p gpurun_out
REPO=$(pwd)
export TMPDIR=/tmp
cd /tmp
for t in ${TRACES:-c2 c1 c3}; do
  case $t in
    c2) ARGS="--steps 24 --warmup 4" ;;
    c1) ARGS="--batch 1 --steps 24 --warmup 4" ;;
    b8) ARGS="--batch 8 --skip-prefill --steps 24 --warmup 4" ;;
    c3) ARGS="--model llama2-7b --batch 4 --prompt-len 16384 --gen-len 64 --skip-prefill --steps 24 --warmup 4" ;;
    b128) ARGS="--batch 128 --skip-prefill --steps 16 --warmup 4 --kv-placement bottom" ;;
    b256) ARGS="--batch 256 --skip-prefill --steps 16 --warmup 4 --kv-placement bottom" ;;
  esac
  rm -rf $REPO/gpurun_out/prof_$t
  timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_$t -o bench -- python $REPO/bench.py $ARGS --no-cpu-baseline --no-extras --no-reference --kv-blocks ${KV_BLOCKS:-0} > $REPO/gpurun_out/trace_$t.log 2>&1
  echo "== $t rocprof rc=$?"; tail -1 $REPO/gpurun_out/trace_$t.log | cut -c1-400
  DB=$(find $REPO/gpurun_out/prof_$t -name "*.db" | head -1)
  python $REPO/tools/rocpd_stats.py $DB 28 > $REPO/gpurun_out/trace_$t.md 2>&1; cat $REPO/gpurun_out/trace_$t.md
  rm -rf $REPO/gpurun_out/prof_$t
done
