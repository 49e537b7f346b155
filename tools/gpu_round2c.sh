#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
echo "== tests"
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -x -s -k "deferred or add_scale or golden or oracle_model or mixed or full_width or paged or norm_silu" > gpurun_out/pytest_c.log 2>&1; echo "rc=$?"; grep -E "deferred rmsnorm|passed|failed|FAILED|Error" gpurun_out/pytest_c.log | cut -c1-300 | tail -20
echo "== bench deferred (default) vs exact"
timeout 600 python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > gpurun_out/bench_c_default.log
python -c "
import json
d=json.loads(open('gpurun_out/bench_c_default.log').read())
print('bench default: ms/step', d['ms_per_step'], 'frac', d['step_roofline']['frac'], 'attn', d.get('roofline_paged_attention',{}).get('us_per_launch'), 'gemm', d['roofline']['us_per_launch'])
"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_c -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/gpurun_out/prof_c_bench.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find gpurun_out/prof_c -name "*.db" | head -1); python tools/rocpd_stats.py $DB 24 > gpurun_out/prof_c_stats.md 2>&1; head -22 gpurun_out/prof_c_stats.md | cut -c1-200
echo "== parity full width"
timeout 1500 python -m pytest tests/test_gpu_parity_fullwidth.py -m gpu -q -s > gpurun_out/pytest_parity_c.log 2>&1; echo "rc=$?"; grep -E "full-width parity|passed|failed|FAILED|Error" gpurun_out/pytest_parity_c.log | cut -c1-1100 | tail -12
