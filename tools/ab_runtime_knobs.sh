# A/B of HIP runtime environment knobs against the sampler's per-launch floor (160 dependent launches per step, replayed as hipGraphs):
# the default bench (short) under each setting.  Output: one line per setting.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-knobs}; mkdir -p $O; cd $R
run() {
  name=$1; shift
  env "$@" timeout 300 python bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-gpu-eager-baseline --train-steps 0 --lfae-train-steps 0 > $O/$name.json 2> $O/$name.err
  python - "$name" "$O/$name.json" <<'PY'
import json, sys
try:
    b = json.load(open(sys.argv[2])); print("%-44s %.4f videos/s  %.2f ms" % (sys.argv[1], b["value"], b["ms_per_step"]))
except Exception as e:
    print("%-44s FAILED %s" % (sys.argv[1], e))
PY
}
run default LFDM_NOOP=1
run AMD_OPT_FLUSH_0 AMD_OPT_FLUSH=0
run GRAPH_PACKET_CAPTURE_0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run GRAPH_PACKET_CAPTURE_1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run GRAPH_BATCH_SIZE_1024 DEBUG_HIP_GRAPH_BATCH_SIZE=1024
run SYSTEM_SCOPE_SIGNAL_0 ROC_SYSTEM_SCOPE_SIGNAL=0
run FLUSH_ON_EXECUTION_1 GPU_FLUSH_ON_EXECUTION=1
run ACTIVE_WAIT ROC_ACTIVE_WAIT_TIMEOUT=1000
run FORCE_GRAPH_QUEUES DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run default_again LFDM_NOOP=2
