# last sanity after the focus-mask / skips=False edits on the headline path: smoke() and a short bench
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r03ad}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
timeout 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-roofline --train-steps 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['train']['ms_per_step'] if isinstance(d.get('train'),dict) else d.get('train'))"
