R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pt; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d $O/pf -o f -- python $R/tools/unet_step.py 2 > /dev/null 2> $O/pf.err
rocprofv3 --pmc WRITE_SIZE -d $O/pw -o w -- python $R/tools/unet_step.py 2 > /dev/null 2> $O/pw.err
python $R/tools/pmc_conv_traffic.py $(find $O/pf -name '*.db' | head -1) $(find $O/pw -name '*.db' | head -1) 2 > $O/traffic.json 2> $O/traffic.err
rm -rf $O/pf $O/pw
cat $O/traffic.json; tail -3 $O/traffic.err
