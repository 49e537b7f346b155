#!/bin/bash
# tools/gpu_r05d.sh — r05: tests of the rows / norm-on-the-fly path, then the decode step with and without it.
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest rows"
timeout 1200 python -m pytest tests/test_gpu_rows.py -x -q --timeout=600 > gpurun_out/r05d_pytest_rows.log 2>&1; echo "rc=$?"; tail -15 gpurun_out/r05d_pytest_rows.log
for b in 32 1 8 16; do
  for f in "" "--no-rows-decode"; do
    echo "== bench batch $b $f"
    timeout 600 python bench.py --batch $b --steps 40 --warmup 6 --no-extras --no-cpu-baseline --no-reference $f > gpurun_out/r05d_bench_b${b}${f:+_norows}.log 2>&1
    tail -1 gpurun_out/r05d_bench_b${b}${f:+_norows}.log | python -c "
import json,sys
try:
    r=json.loads(sys.stdin.read()); print({k:r.get(k) for k in ('value','ms_per_step')}, r.get('step_roofline',{}).get('frac'))
except Exception as e: print('parse failed', e)"
  done
done
