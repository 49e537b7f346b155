#!/bin/bash
# tools/gpu_pmc_prefill.sh — PMC passes over the prefill attention micro-benchmark (one counter set per run, --kernel-trace
# only): where do the waves of prefill_attn_kernel spend their cycles? LIBS="<suffix list>" selects variant libraries
# (libswiftllm_hip<suffix>.so; "" = the product). Output: gpurun_out/pmc_prefill/<lib>_<shape>_<set>.txt
export TMPDIR=/tmp; R=$(pwd); mkdir -p $R/gpurun_out/pmc_prefill; cd /tmp
SET1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES"
SET2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"
SET3="TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY"
for lib in ${LIBS:-prod}; do
  [ "$lib" = prod ] && lib=""
  for s in ${SHAPES:-c3 c4}; do
    n=1
    for set in "$SET1" "$SET2" "$SET3"; do
      d=$R/gpurun_out/pmc_prefill/run_${lib:-prod}_${s}_$n; rm -rf $d
      SWIFTLLM_HIP_LIB=$R/swiftllm_amd/csrc/libswiftllm_hip$lib.so rocprofv3 --pmc $set --kernel-trace -d $d -o pf -- python $R/tools/prefill_attn_micro.py --shape $s --iters 4 > $d.log 2>&1
      DB=$(find $d -name "*.db" | head -1)
      python $R/tools/rocpd_pmc.py $DB prefill_attn > $R/gpurun_out/pmc_prefill/${lib:-prod}_${s}_set$n.txt 2>&1
      echo "== ${lib:-prod} $s set$n"; cat $R/gpurun_out/pmc_prefill/${lib:-prod}_${s}_set$n.txt
      rm -rf $d
      n=$((n+1))
    done
  done
done
