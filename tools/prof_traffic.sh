# roofline.traffic of bench.py: separate --pmc passes (FETCH_SIZE; WRITE_SIZE) over the Winograd launches of a sampler step
# and over the warp launches of a decode -> gpurun_out/pt/traffic.json (copy to profiles/r02_traffic.json)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pt; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; N=2
for t in wino warp; do
  rocprofv3 --pmc FETCH_SIZE -d $O/${t}_f -o f -- python $R/tools/pmc_targets.py $t $N > $O/${t}_f.log 2> $O/${t}_f.err
  rocprofv3 --pmc WRITE_SIZE -d $O/${t}_w -o w -- python $R/tools/pmc_targets.py $t $N > $O/${t}_w.log 2> $O/${t}_w.err
done
L=$(grep PMC_TARGET $O/wino_f.log | sed 's/.*launches_per_iter=\([0-9]*\) reduce_per_iter=\([0-9]*\).*/\1 \2/')
db() { find $O/$1 -name '*.db' | head -1; }
python $R/tools/pmc_traffic.py $(db wino_f) $(db wino_w) $(db warp_f) $(db warp_w) $L $N > $O/traffic.json 2> $O/traffic.err
rm -rf $O/wino_f $O/wino_w $O/warp_f $O/warp_w
cat $O/traffic.json; tail -n 3 $O/traffic.err
