export TMPDIR=/tmp; R=$(pwd); mkdir -p gpurun_out/pmc_gemm; cd /tmp
python $R/tools/gemm_silu_micro.py 2>&1 | tail -1 | tee $R/gpurun_out/pmc_gemm/micro.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_gemm/fetch -o g -- python $R/tools/gemm_silu_micro.py > $R/gpurun_out/pmc_gemm/fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc_gemm/write -o g -- python $R/tools/gemm_silu_micro.py > $R/gpurun_out/pmc_gemm/write.log 2>&1; echo "write rc=$?"
cd $R
for k in fetch write; do DB=$(find gpurun_out/pmc_gemm/$k -name "*.db" | head -1); python tools/rocpd_pmc.py $DB gemm_skinny_ring; done | tee gpurun_out/pmc_gemm/counters.txt
