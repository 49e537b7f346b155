export TMPDIR=/tmp; R=$(pwd); mkdir -p gpurun_out/pmc; cd /tmp
for s in c3 c2 c4 c3_b128 long; do python $R/tools/paged_attn_micro.py --shape $s 2>&1 | tail -1; done > $R/gpurun_out/pmc/micro_shapes.log
cat $R/gpurun_out/pmc/micro_shapes.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc/fetch -o pa -- python $R/tools/paged_attn_micro.py --shape c3 --iters 64 > $R/gpurun_out/pmc/fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc/write -o pa -- python $R/tools/paged_attn_micro.py --shape c3 --iters 64 > $R/gpurun_out/pmc/write.log 2>&1; echo "write rc=$?"
ls -R $R/gpurun_out/pmc | head -20
