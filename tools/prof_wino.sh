# PMC passes over single Winograd conv shapes (tools/probe_one_conv.py); summaries -> gpurun_out/pw/pmc.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pw; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail > $O/avail.txt 2>&1
grep -o -E "\b(TA|TCP|TD|TCC)_[A-Za-z0-9_]+" $O/avail.txt | sort -u > $O/avail_mem.txt
i=0
for shape in "256 256 3 8 40" "64 64 3 32 40"; do
  i=$((i+1))
  rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max -d $O/a$i -o a -- python $R/tools/probe_one_conv.py $shape 5 > /dev/null 2> $O/a$i.err
  rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum -d $O/b$i -o b -- python $R/tools/probe_one_conv.py $shape 5 > /dev/null 2> $O/b$i.err
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU -d $O/c$i -o c -- python $R/tools/probe_one_conv.py $shape 5 > /dev/null 2> $O/c$i.err
  for d in a b c; do echo "== shape $shape pass $d"; python $R/tools/rocpd_pmc.py $(find $O/$d$i -name '*.db' | head -1) conv_wino; grep -i -m3 "error\|invalid\|not found" $O/$d$i.err; done
done > $O/pmc2.txt 2>&1
rm -rf $O/a? $O/b? $O/c?
cat $O/pmc2.txt; wc -l $O/avail_mem.txt
