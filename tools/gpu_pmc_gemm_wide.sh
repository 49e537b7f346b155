#!/bin/bash
# tools/gpu_pmc_gemm_wide.sh — PMC passes over swl_gemm_packed_wide (library's own plan) at M tokens (default 128), one
# projection shape of a Llama-3-8B layer per run: HBM traffic (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate passes —
# MI355X_MICROARCH.md), LDS bank conflicts and matrix-pipe busy cycles. Output: gpurun_out/pmc_gemm_wide/counters.txt
export TMPDIR=/tmp; R=$(pwd); O=$R/gpurun_out/pmc_gemm_wide; mkdir -p $O; cd /tmp
M=${M:-128}
: > $O/counters.txt
for shape in ${SHAPES:-qkv up_gate down}; do
  CMD="python $R/tools/gemm_wide_micro.py --auto-only --m $M --iters 12 --shapes $shape"
  $CMD 2>/dev/null | tee -a $O/counters.txt
  for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES"; do
    tag=${shape}_$(echo $pass | cut -d' ' -f1)
    rocprofv3 --pmc $pass --kernel-trace -d $O/$tag -o g -- $CMD > $O/$tag.log 2>&1; echo "$tag rc=$?"
    DB=$(find $O/$tag -name "*.db" | head -1)
    [ -n "$DB" ] && (echo "== $shape M=$M"; python $R/tools/rocpd_pmc.py $DB gemm_packed_wide) | tee -a $O/counters.txt
    rm -rf $O/$tag
  done
done
