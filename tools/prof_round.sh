R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/p2; cd /tmp; export TMPDIR=/tmp
python $R/bench.py --steps 3 --warmup 1 > $R/gpurun_out/p2/bench.json 2> $R/gpurun_out/p2/bench.err
python $R/bench.py --steps 2 --warmup 1 --batch 8 --no-cpu-baseline --no-roofline --train-steps 0 > $R/gpurun_out/p2/bench_b8.json 2>> $R/gpurun_out/p2/bench.err
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p2/kt -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --train-steps 0 > $R/gpurun_out/p2/bench_prof.json 2> $R/gpurun_out/p2/kt.err
python $R/tools/rocpd_stats.py $(find $R/gpurun_out/p2/kt -name '*.db' | head -1) > $R/gpurun_out/p2/kernel_stats.txt 2>&1
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/p2/pf -o f -- python $R/tools/unet_step.py 3 > /dev/null 2> $R/gpurun_out/p2/pf.err
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/p2/pw -o w -- python $R/tools/unet_step.py 3 > /dev/null 2> $R/gpurun_out/p2/pw.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $R/gpurun_out/p2/ps -o s -- python $R/tools/unet_step.py 3 > /dev/null 2> $R/gpurun_out/p2/ps.err
for k in conv_ksw conv_igemm conv_splitk temporal_attn_fused attention_kernel linattn_fused gn_apply; do for d in pf pw ps; do echo "== $d $k"; python $R/tools/rocpd_pmc.py $(find $R/gpurun_out/p2/$d -name '*.db' | head -1) $k; done; done > $R/gpurun_out/p2/pmc.txt 2>&1
rm -rf $R/gpurun_out/p2/kt $R/gpurun_out/p2/pf $R/gpurun_out/p2/pw $R/gpurun_out/p2/ps
tail -2 $R/gpurun_out/p2/bench.err; cat $R/gpurun_out/p2/bench_b8.json | cut -c1-200
