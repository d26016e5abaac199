cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_s
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_s -o t -- python $GRAFT_REPO_ROOT/tools/slds_sweep_timeline.py > /dev/null 2>&1
python - <<'PY'
import csv, glob
f=glob.glob('/tmp/prof_s/**/*kernel_trace.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last ascent: find the last occurrence of the first hmm kernel group: print the 40 kernels before the first "hmm_estep2" of the last 40 hmm launches
idx=[i for i,r in enumerate(rows) if 'hmm_estep2' in r['Kernel_Name']]
start=idx[-12]
t0=int(rows[start]['Start_Timestamp'])
for r in rows[max(0,start-40):start+8]:
    print("%9.1f us  %8.1f us  %s" % ((int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, r['Kernel_Name'][:90]))
PY
