# whole-step counter passes (VERDICT r2 #4b, #7): FETCH_SIZE | WRITE_SIZE | SQ set | TCC set, one rocprofv3 --pmc pass each
# (no tracing domains beside --pmc) over tools/pmc_video.py -> gpurun_out/$TAG/step_pmc.{json,txt}
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-pmc}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
run() { timeout 150 rocprofv3 --pmc $2 -d $O/p_$1 -o p -- python $R/tools/pmc_video.py > $O/p_$1.log 2> $O/p_$1.err; echo "pass $1 rc=$?"; }
run f "FETCH_SIZE"
run w "WRITE_SIZE"
run s "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU"
run t "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"
db() { find $O/p_$1 -name '*.db' | head -1; }
python $R/tools/pmc_video_report.py $(db f) $(db w) $(db s) $(db t) $O/step_pmc.json > $O/step_pmc.txt 2> $O/step_pmc.err
rm -rf $O/p_f $O/p_w $O/p_s $O/p_t
head -n 12 $O/step_pmc.txt; tail -n 3 $O/step_pmc.err
