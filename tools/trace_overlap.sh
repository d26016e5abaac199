cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p2 -o b -- python $GRAFT_REPO_ROOT/tools/filt_time.py > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/p2/**/b_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows = [r for r in rows if 'lds_estep' in r['Kernel_Name']]
t0 = int(rows[0]['Start_Timestamp'])
for r in rows[-8:]:
    print(r['Kernel_Name'][:70], 'queue', r.get('Queue_Id'), 'start %.1f us end %.1f us' % ((int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-t0)/1e3))
PY
