#!/bin/bash
# tools/gpu_ref.sh — the staged reference (oracle/_ref, `python -m oracle.make_ref`) on the MI355X box:
# probe, then its decode/prefill throughput at BASELINE configs[2]/[1]/[3] (oracle/ref_triton.py bench).
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
echo "== probe"
timeout 600 python -m oracle.ref_triton probe > gpurun_out/ref_probe.log 2>&1; echo "probe rc=$?"; tail -5 gpurun_out/ref_probe.log
echo "== reference c2 (batch 32, ctx 1024..; prefill 32x1024)"
timeout 900 python -m oracle.ref_triton bench --config c2 --steps ${REF_STEPS:-20} --warmup 5 --prefill-len 1024 > gpurun_out/ref_c2.log 2>&1; echo "rc=$?"; tail -2 gpurun_out/ref_c2.log | cut -c1-600
echo "== reference c1 (batch 1)"
timeout 600 python -m oracle.ref_triton bench --config c1 --steps 40 --warmup 5 > gpurun_out/ref_c1.log 2>&1; echo "rc=$?"; tail -1 gpurun_out/ref_c1.log | cut -c1-600
echo "== reference c3 (llama2-7b 4x16k)"
timeout 900 python -m oracle.ref_triton bench --config c3 --steps 20 --warmup 3 > gpurun_out/ref_c3.log 2>&1; echo "rc=$?"; tail -1 gpurun_out/ref_c3.log | cut -c1-600
