# parity of the convolution / UNet / sampler paths + one profiled step: the standard "did this kernel change hold" call
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-chk}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_ops_parity.py tests/test_golden_gpu.py tests/test_end_to_end.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -n 5 $O/pytest.txt
bash tools/prof_sequence.sh $TAG > $O/prof.txt 2>&1
python - <<EOF
import json
try:
    print(json.loads(open("$O/bench_prof.json").read().strip().splitlines()[-1])["value"], "videos/s under rocprof")
except Exception as e: print("bench parse failed", e)
EOF
tail -n 2 $O/step_sequence.txt
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --train-steps 0 > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json | head -c 400; echo
