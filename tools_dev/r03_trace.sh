#!/bin/bash
# on the GPU box: kernel trace (start / end timestamps) of a few replayed steps of the headline bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03t; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o tr -- python3 bench.py --no-cpu-baseline --no-dropin --steps 6 --warmup 2 ${ARGS} > $O/log.txt 2>&1
f=$(find $O/tr -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# keep the last 2500 rows compactly: name, start, end, stream/queue
out=open('gpurun_out/r03t/trace_tail.csv','w')
t0=int(rows[0]['Start_Timestamp'])
for r in rows[-4000:]:
    n=r['Kernel_Name'].replace('void ','').replace('d3m::','').split('(')[0].split('<')[0]
    out.write(f"{n},{int(r['Start_Timestamp'])-t0},{int(r['End_Timestamp'])-t0},{r.get('Queue_Id','')},{r.get('Stream_Id','')}\n")
print(len(rows), rows[0].keys())
PY
