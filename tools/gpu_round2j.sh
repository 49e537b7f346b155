#!/bin/bash
# r02-j: chunked prefill FFN, graph-bucket split widths
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
echo "== tests"
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "golden or chunked or mixed or oracle_model" > gpurun_out/pytest_j.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/pytest_j.log | cut -c1-300
for chunk in 2048 0 4096 1024; do
  echo "== bench prefill chunk=$chunk"
  SWL_PREFILL_FFN_CHUNK=$chunk timeout 600 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > gpurun_out/bench_j_$chunk.json
  python -c "
import json
d=json.loads(open('gpurun_out/bench_j_$chunk.json').read())
print('chunk $chunk: prefill tok/s', d.get('prefill_tok_s'), 'prefill ms', d.get('prefill_ms'), 'decode ms/step', d['ms_per_step'], 'kv_blocks', d['config'].get('kv_blocks'))
"
done
echo "== configs[3]"
timeout 600 python bench.py --model llama2-7b --batch 4 --prompt-len 16384 --gen-len 64 --skip-prefill --no-extras --steps 24 --warmup 4 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('c3 graph: ms/step', d['ms_per_step'], 'frac', d['step_roofline']['frac'])"
timeout 600 python bench.py --model llama2-7b --batch 4 --prompt-len 16384 --gen-len 64 --skip-prefill --no-extras --steps 24 --warmup 4 --no-cpu-baseline --no-hip-graph 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('c3 eager: ms/step', d['ms_per_step'], 'frac', d['step_roofline']['frac'])"
