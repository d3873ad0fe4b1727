# FETCH_SIZE calibration per access width (tools/ubench/fetch_calib.hip): counter KB per kernel against the 2 GiB each kernel reads
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-fcal}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 120 rocprofv3 --pmc FETCH_SIZE -d $O/p -o p -- $R/tools/ubench/fetch_calib.bin > $O/run.log 2>&1; echo "rc=$?"
DB=$(find $O/p -name '*.db' | head -1)
python - "$DB" > $O/fetch_calib.txt <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select dispatch_id, name, sum(counter_value), min(duration) from pmc_events where counter_name='FETCH_SIZE' group by dispatch_id order by dispatch_id").fetchall()
print("# rocprofv3 --pmc FETCH_SIZE over tools/ubench/fetch_calib.bin: every kernel reads 2 GiB = 2097152 KB exactly once")
print("%-44s %14s %8s %10s" % ("kernel", "FETCH_SIZE KB", "/ bytes", "us"))
for did, name, val, dur in rows:
    print("%-44s %14.0f %8.3f %10.1f" % (name[:44], val, val / 2097152.0, dur / 1e3))
PY
rm -rf $O/p; cat $O/fetch_calib.txt
