#!/bin/bash
# tools/ktrace.sh <nranks> : kernel timeline of the last frame rank_prof.py renders (one line per launch)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o p -- python3 $GRAFT_REPO_ROOT/tools/rank_prof.py "$@" > /dev/null 2>&1
g=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python3 - $g <<PY
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last frame: from the last render_kernel on
last=max(i for i,r in enumerate(rows) if "render_kernel" in r["Kernel_Name"])
t0=int(rows[last]["Start_Timestamp"])
for r in rows[last:]:
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    print("%-16s start %8.1f us  dur %8.1f us" % (r["Kernel_Name"].replace("void ","")[:16], (s-t0)/1e3, (e-s)/1e3))
PY
