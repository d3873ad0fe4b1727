# after the last bench.py edits (six decode warps, guarded extras): the default bench line without the CPU-baseline and training legs
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r03t}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --train-steps 0 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json,sys; d=json.load(open('$O/bench.json')); print(d['value'], d['roofline'].get('frac'), d['roofline'].get('rocprofv3',{}).get('file'), d['roofline'].get('step_traffic',{}).get('file')); print(d['warp'])"
