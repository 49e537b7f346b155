export TMPDIR=/tmp; R=$(pwd); mkdir -p gpurun_out/pmc_prefill; cd /tmp
for s in c3 c4; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $R/gpurun_out/pmc_prefill/$s -o pf -- python $R/tools/prefill_attn_micro.py --shape $s --iters 5 > $R/gpurun_out/pmc_prefill/$s.log 2>&1; echo "$s rc=$?"; tail -1 $R/gpurun_out/pmc_prefill/$s.log
done
