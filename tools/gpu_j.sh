R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02_j; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_ops_parity.py tests/test_golden_gpu.py -m gpu -x -q -k "warp or generator or sample_one_video" 2>&1 | tail -n 2
timeout 300 python bench.py --steps 3 --train-steps 0 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; grep bench $O/bench.err | tail -n 12
python - <<EOF
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["value"], "videos/s"); print("warp", json.dumps(d["warp"])[:400]); print("cpu", json.dumps(d.get("cpu_baseline"))[:900])
EOF
