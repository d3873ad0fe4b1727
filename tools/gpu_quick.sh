# quick A/B call: winograd phase probe + conv microbench + winograd/groupnorm parity + one profiled step
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-q}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
python tools/probe_wino_phases.py 2>&1 | grep -v amdgpu.ids > $O/wino_phases.txt; cat $O/wino_phases.txt
python tools/ubench/launch_floor.py 2>&1 | grep -v amdgpu.ids > $O/launch_floor.txt; grep apply $O/launch_floor.txt
timeout 600 python -m pytest tests/test_ops_parity.py -m gpu -x -q -k "winograd or groupnorm or deconv or heads" 2>&1 | tail -n 2
timeout 300 python -m pytest tests/test_golden_gpu.py -m gpu -x -q -k "unet_forward" 2>&1 | tail -n 2
bash tools/prof_sequence.sh $TAG > $O/prof.txt 2>&1; tail -n 1 $O/step_sequence.txt
