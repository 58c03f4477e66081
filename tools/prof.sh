#!/bin/bash
# tools/prof.sh <name> <bench args...> : rocprofv3 kernel trace + stats of one bench.py run, written under gpurun_out/<name>/
name="$1"; shift
root="${GRAFT_REPO_ROOT:-$PWD}"
out="$root/gpurun_out/$name"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o prof -- python3 "$root/bench.py" --no-cpu-baseline "$@" > "$out/bench.json" 2> "$out/err.log" || { tail -5 "$out/err.log"; exit 1; }
f=$(find "$out" -name "*kernel_stats.csv" | head -1)
cat "$f"
cat "$out/bench.json"
