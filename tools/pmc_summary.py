"""tools/pmc_summary.py <prefix> : from the PMC passes and one-frame-at-a-time bench lines of a profile round
(gpurun_out/<prefix>/<scene>_<sampling>_pmc.txt and ..._f1_bench.json, tools/profile_round.sh) writes
profiles/pmc_summary.json - per configuration, for the DOMINANT kernel of the call (the first pass, or the second pass's three
kernels together), what bench.py puts into its `roofline` object next to `frac`:
  valu_busy                SQ_ACTIVE_INST_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)
  lane_util                SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)
  flop_per_lane_op         the kernel's algorithmic FLOPs / SQ_THREAD_CYCLES_VALU (the counter adds the active lanes of every VALU instruction: an FMA on every lane = 2, packed = 4)
  valu_insts_per_wave_ray  SQ_INSTS_VALU / (rays / 64), the call's rays apportioned to the kernel by its share of the algorithmic FLOPs
  wait_frac                SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (share of wave residency spent waiting for an instruction's operands)
  salu_per_valu            SQ_INSTS_SALU / SQ_INSTS_VALU
Counters are per frame (the passes ran `steps` frames; tools/pmc.sh divides)."""
import glob, json, os, re, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pre = sys.argv[1]
src = os.path.join(root, "gpurun_out", pre)
out = {"_what": __doc__.split("writes", 1)[1].strip(), "_build": pre}
for f in sorted(glob.glob(os.path.join(src, "*_pmc.txt"))):
    key = os.path.basename(f)[:-8]
    bj = os.path.join(src, key + "_f1_bench.json")
    if not os.path.exists(bj):
        continue
    bench = json.load(open(bj))
    r = bench["roofline"]
    dom = r["kernel"]
    first = dom.startswith("render_kernel")
    names = ("render_kernel",) if first else ("resample_kernel", "units_kernel", "sum_up_kernel")
    c = {}
    for line in open(f):
        m = re.match(r"pass\d+ (\S+) (\S+) per-frame (\S+)", line)
        if m and m.group(1) in names:
            c[m.group(2)] = c.get(m.group(2), 0.0) + float(m.group(3))
    need = ("SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU")
    if any(k not in c for k in need):
        out[key] = {"error": "counters missing: " + " ".join(k for k in need if k not in c)}
        continue
    flops = r["per_kernel"][dom]["algorithmic_flops"]
    share = flops / max(r["call"]["algorithmic_flops"], 1.0)
    rays = bench["config"]["rays_per_step"] * share
    out[key] = {
        "kernel": dom,
        "valu_busy": round(c["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0), 4),
        "lane_util": round(c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]), 4),
        "flop_per_lane_op": round(flops / c["SQ_THREAD_CYCLES_VALU"], 4),
        "valu_insts_per_wave_ray": round(c["SQ_INSTS_VALU"] / (rays / 64.0), 1),
        "wait_frac": round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 4) if "SQ_WAIT_INST_ANY" in c and c.get("SQ_WAVE_CYCLES") else None,
        "salu_per_valu": round(c["SQ_INSTS_SALU"] / c["SQ_INSTS_VALU"], 3) if "SQ_INSTS_SALU" in c else None,
        "insts_valu": c["SQ_INSTS_VALU"], "ms_kernel_alone": round(r["per_kernel"][dom]["ms"], 3), "frac": round(r["frac"], 4),
        "source": "profiles/%s_%s_pmc.txt" % (pre, key),
    }
json.dump(out, open(os.path.join(root, "profiles", "pmc_summary.json"), "w"), indent=1)
for k, v in out.items():
    if not k.startswith("_"):
        print(k, json.dumps(v))
