# counter passes over the F(4x4) / F(2x2) Winograd kernels on the LFAE bottleneck shape of a B = 8 training step
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-w4pmc}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for v in 1 0; do
  for pass in s t; do
    if [ $pass = s ]; then C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"; else C="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"; fi
    LFDM_WINO4=$v timeout 120 rocprofv3 --pmc $C -d $O/p_${v}_$pass -o p -- python $R/tools/probe_wino4.py > $O/p_${v}_$pass.log 2>&1; echo "wino4=$v pass $pass rc=$?"
    DB=$(find $O/p_${v}_$pass -name '*.db' | head -1)
    echo "== LFDM_WINO4=$v pass $pass" >> $O/wino4_pmc.txt
    python $R/tools/rocpd_pmc.py $DB conv_wino >> $O/wino4_pmc.txt 2>&1
    rm -rf $O/p_${v}_$pass
  done
  LFDM_WINO4=$v timeout 120 rocprofv3 --pmc FETCH_SIZE -d $O/p_${v}_f -o p -- python $R/tools/probe_wino4.py > $O/p_${v}_f.log 2>&1
  DB=$(find $O/p_${v}_f -name '*.db' | head -1); echo "== LFDM_WINO4=$v pass f" >> $O/wino4_pmc.txt; python $R/tools/rocpd_pmc.py $DB conv_wino >> $O/wino4_pmc.txt 2>&1; rm -rf $O/p_${v}_f
done
cat $O/wino4_pmc.txt
