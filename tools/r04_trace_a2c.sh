#!/bin/bash
# rocprofv3 kernel trace of a few A2C iterations at BASELINE config 3 and a steady-state excerpt of the rollout's timeline
# (PIPE=1: the rollout as two half-batches on two streams, the default; PIPE=0: unsplit).   usage: r04_trace_a2c.sh <out tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04h}
mkdir -p $O
export TMPDIR=/tmp; cd /tmp
export PIPE=${PIPE:-1}
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_a2c -- python3 $R/tools/prof_a2c_run.py > $O/trace_a2c.log 2>&1
echo "trace rc=$?" | tee -a $O/status.txt
cd $R
find $O -name "*agent_info.csv" -delete
OUT=$O python3 - <<'PY'
import csv,glob,os
f=glob.glob(os.environ["OUT"]+"/trace_a2c/*/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
def short(n):
    for k in ("actor_head","sparse_rows_sum","env_kernel_packed","obs_indices"):
        if k in n: return k
    return n[:30]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'actor_head' in r['Kernel_Name']]
a=idx[-60]; b=idx[-46]
t0=int(rows[a]['Start_Timestamp'])
for r in rows[a:b+1]:
    s=int(r['Start_Timestamp'])-t0; e=int(r['End_Timestamp'])-t0
    print("%8.1f %8.1f %6.1f  q%s  %-20s %s" % (s/1e3,e/1e3,(e-s)/1e3, r['Queue_Id'], short(r['Kernel_Name']), r['Grid_Size_X']))
PY
