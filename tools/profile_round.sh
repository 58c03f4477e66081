#!/bin/bash
# tools/profile_round.sh <prefix> : the round's evidence for profiles/ - per scene a rocprofv3 kernel trace + stats of one
# bench.py run (one frame at a time, so that the per-kernel averages add up to the call) and the PMC passes (instruction
# mix / waits, then FETCH_SIZE, WRITE_SIZE, L2 hits in passes of their own).  Run on the GPU box; summaries land in
# gpurun_out/<prefix>/ and are copied to profiles/ by hand afterwards.
pre="$1"; shift
root="${GRAFT_REPO_ROOT:-$PWD}"
out="$root/gpurun_out/$pre"
mkdir -p "$out"
# (PROFILE_SPECS: a subset, "scene sampling steps;..." - a call of gpurun holds about half of the twelve)
IFS=";" read -ra SPECS <<< "${PROFILE_SPECS:-ch07 shipped 4;ch07 1spp 12;ch11a shipped 6;ch11a 1spp 12;ch11b shipped 6;ch11b 1spp 12;teapot shipped 3;teapot 1spp 12;soft shipped 2;dof shipped 1;dice shipped 1;dice 1spp 6}"
for spec in "${SPECS[@]}"; do
  set -- $spec
  scene=$1; samp=$2; steps=$3
  args="--scene $scene --sampling $samp --steps $steps --warmup 1 --frames-in-flight 1"
  echo "== $scene $samp: kernel trace"
  tools/prof.sh $pre/${scene}_${samp}_f1 $args > "$out/${scene}_${samp}_f1.txt" 2>&1 || { tail -5 "$out/${scene}_${samp}_f1.txt"; exit 1; }
  f=$(find "$out/${scene}_${samp}_f1" -name "*kernel_stats.csv" | head -1)
  cp "$f" "$out/${scene}_${samp}_f1_kernel_stats.csv"
  cp "$out/${scene}_${samp}_f1/bench.json" "$out/${scene}_${samp}_f1_bench.json"
  echo "== $scene $samp: PMC"
  PMC_SETS=all tools/pmc.sh gpurun_out/$pre/pmc_${scene}_${samp} "$args" > "$out/${scene}_${samp}_pmc.txt" 2>&1 || { tail -5 "$out/${scene}_${samp}_pmc.txt"; exit 1; }
  tail -3 "$out/${scene}_${samp}_pmc.txt"
done
