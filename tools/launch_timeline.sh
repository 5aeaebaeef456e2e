#!/bin/bash
# development helper: per-launch durations of one render (rocprofv3 kernel trace), in launch order per stream
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/timeline; rm -rf $O; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --output-format csv -d $O/out -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 "$@" > $O/log.txt 2>&1
f=$(find $O/out -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "render_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=int(rows[0]["Start_Timestamp"])
byq={}
for r in rows: byq.setdefault(r.get("Queue_Id","?"),[]).append(r)
for q,rs in byq.items():
    print("queue",q,"launches",len(rs))
    print(" start_ms:dur_ms(grid) ", " ".join("%.0f:%.1f(%s)"%((int(r["Start_Timestamp"])-t0)/1e6,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6,r.get("Grid_Size_X", r.get("Grid_Size","?"))) for r in rs))
print("total span ms", (max(int(r["End_Timestamp"]) for r in rows)-t0)/1e6)
PY
rm -rf $O/out
