# round 2, GPU call A: time the built-but-untimed Winograd variants, first GPU run of the wide kernel + GPU fuzz, first GPU runs of the two drivers
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 bash tools/sweep_conv.sh > $O/r02_a_sweep_summary.txt 2>&1
LFDM_FUZZ_GPU=1 timeout 600 python -m pytest tests/test_ops_parity.py -m gpu -x -q -k "wino or fuzz or stage" > $O/r02_a_fuzz.txt 2>&1; echo "fuzz rc=$?" >> $O/r02_a_fuzz.txt
timeout 300 python tools/demo.py --synthetic --steps 5 --frames 8 --out $O/demo_r02 > $O/r02_a_demo.txt 2>&1; echo "demo rc=$?" >> $O/r02_a_demo.txt
timeout 300 python tools/train_dm.py --synthetic --final-step 5 --batch-size 2 --num-workers 0 --out /tmp/train_r02 --frames 8 --print-freq 1 --save-img-freq 3 > $O/r02_a_train.txt 2>&1; echo "train rc=$?" >> $O/r02_a_train.txt
for f in $O/r02_a_fuzz.txt $O/r02_a_demo.txt $O/r02_a_train.txt; do echo "== $f"; tail -n 4 $f; done
ls -la /tmp/train_r02 >> $O/r02_a_train.txt 2>&1; cp /tmp/train_r02/*.png $O/ 2>/dev/null; true
