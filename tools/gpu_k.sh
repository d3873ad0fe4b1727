R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for r in 4 2 1; do echo "LFDM_WARP_R=$r"; LFDM_WARP_R=$r timeout 200 python bench.py --steps 1 --warmup 1 --train-steps 0 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['warp']['value'], d['warp']['us_per_video'], d['warp']['roofline']['frac'])"; done
