# round 3, first GPU call: the pointwise schedule (conv_pw.hip) - parity, per-shape A/B against the LDS-staged schedules, the
# GroupNorm-apply workgroup-size sweep, and the end-to-end effect (bench with / without, profiled step sequence).
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r03a}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_ops_parity.py -m gpu -x -q -k "pointwise or groupnorm or conv2d or layernorm or gn" > $O/pytest_ops.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_ops.txt; tail -n 3 $O/pytest_ops.txt
timeout 300 python tools/bench_pw.py 2>&1 | grep -v amdgpu.ids > $O/bench_pw.txt; tail -n 25 $O/bench_pw.txt
{ for blk in 256 512 1024; do for f4 in 2 4 8; do LFDM_GN_BLOCK=$blk LFDM_GN_F4=$f4 timeout 120 python tools/bench_gn.py 2>&1 | grep -v amdgpu.ids; done; done; } > $O/sweep_gn.txt; python - <<PY
import collections,re
rows=collections.defaultdict(list)
for ln in open("$O/sweep_gn.txt"):
    m=re.match(r"(F4=\S+ BLOCK=\S+)\s+(res .*chunks\s+\d+):\s+([\d.]+) us",ln)
    if m: rows[m.group(2)].append((float(m.group(3)),m.group(1)))
for k,v in rows.items(): print(k, " | ".join("%s %.2f"%(t,u) for u,t in sorted(v)[:3]), "| default(256,2) %.2f"%[u for u,t in v if t=="F4=2 BLOCK=256"][0])
PY
for cfg in "LFDM_PW=0" "LFDM_PW=1" "LFDM_PW=1 LFDM_PW_MAXM=12000"; do
  echo "=== $cfg"; env $cfg timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --train-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done 2>&1 | tee $O/bench_ab.txt
timeout 600 python -m pytest tests/test_golden_gpu.py -m gpu -x -q > $O/pytest_golden.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_golden.txt; tail -n 3 $O/pytest_golden.txt
bash tools/prof_sequence.sh $TAG > $O/prof.txt 2>&1; tail -n 1 $O/step_sequence.txt
