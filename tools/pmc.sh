#!/bin/bash
# PMC passes over the bench (one rocprofv3 run per counter set; see MI355X_MICROARCH.md "rocprofv3 PMC slots").
# usage: tools/pmc.sh <outdir> "<bench args>"
out=$1; args="$2"
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/$out
i=0
SETS=${PMC_SETS:-all}
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_FLAT" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  if [ "$SETS" = "mem" ] && [ $i -lt 4 ]; then continue; fi
  if [ "$SETS" = "sq" ] && [ $i -gt 2 ]; then continue; fi
  (cd /tmp && timeout -s KILL 120 rocprofv3 --pmc $set --output-format csv -d $R/$out/pass$i -- python3 $R/bench.py $args --no-cpu-baseline > $R/$out/pass$i.json 2> $R/$out/pass$i.err) || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$R/$out/pass*/")):
    agg = collections.defaultdict(float)
    frames = 0
    seen = set()
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            kn = next((k for k in ("render_kernel", "resample_kernel", "units_kernel", "sum_up_kernel") if k in row["Kernel_Name"]), None)
            if kn is None: continue
            agg[(kn, row["Counter_Name"])] += float(row["Counter_Value"])
            if kn == "render_kernel" and row["Dispatch_Id"] not in seen: seen.add(row["Dispatch_Id"]); frames += 1
    for (kn, k), v in sorted(agg.items()): print(d.split("/")[-2], kn, k, "per-frame %.6g" % (v / max(frames, 1)), "frames", frames)
PY
