"""tools/round_summary.py <prefix> : after tools/scenes_bench.sh and tools/profile_round.sh <prefix> have run on the GPU box and
gpurun_out/ has come back: writes profiles/<round>_all_scenes.json (one flat row per BASELINE configuration), copies the profile
summaries to profiles/<prefix>_* and rewrites profiles/traffic.json (HBM bytes per call = sum over the call's kernels of
2 x FETCH_SIZE + WRITE_SIZE, both in KB per frame, the doubling as MI355X_MICROARCH.md prescribes for gfx950)."""
import glob, json, os, re, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pre = sys.argv[1]
rows = {}
for f in sorted(glob.glob(os.path.join(root, "gpurun_out", "scenes", "*.json"))):
    d = json.load(open(f)); c = d["config"]; r = d["roofline"]; b = d.get("cpu_baseline") or {}
    rows[os.path.basename(f)[:-5]] = {
        "ms_per_frame": round(d["ms_per_step"], 3), "Mrays_s": round(d["value"]), "rays": c["rays_per_step"], "samples": c["samples_per_step"],
        "frac": round(r["frac"], 4), "frac_call_alone": round(r["call"]["frac_alone"], 4), "ms_call_alone": round(r["call"]["ms_alone"], 3),
        "cpu_Mrays_s": round(b.get("value", 0), 1), "cpu_threads": b.get("cores", 0)}
order = ["ch07_1spp", "ch07_shipped", "ch11a_1spp", "ch11a_shipped", "ch11b_1spp", "ch11b_shipped", "teapot_1spp", "teapot_shipped", "soft_shipped", "dof_shipped",
         "dice_1spp", "dice_shipped"]
out = {"_what": "every BASELINE configuration at 3840x2160, f32, one MI355X, bench.py --frames-in-flight 2 (tools/scenes_bench.sh), build %s; frac = the dominant "
                "kernel alone (roofline.frac), frac_call_alone = the whole call alone" % pre,
       "rows": {k: rows[k] for k in order if k in rows}}
json.dump(out, open(os.path.join(root, "profiles", pre[:3] + "_all_scenes.json"), "w"), indent=1)
traffic = {"_what": "HBM bytes per rtmi_render_device call (all its launches), rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes (profiles/%s_*_pmc.txt, "
                    "tools/pmc.sh), KB -> bytes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950; 3840x2160 f32, build %s" % (pre, pre)}
src = os.path.join(root, "gpurun_out", pre)
for f in sorted(glob.glob(os.path.join(src, "*_pmc.txt")) + glob.glob(os.path.join(src, "*_f1_kernel_stats.csv")) + glob.glob(os.path.join(src, "*_f1_bench.json"))):
    shutil.copy(f, os.path.join(root, "profiles", pre + "_" + os.path.basename(f)))
for f in sorted(glob.glob(os.path.join(src, "*_pmc.txt"))):
    kb = 0.0
    for line in open(f):
        m = re.match(r"pass\d+ (\S+) (FETCH_SIZE|WRITE_SIZE) per-frame (\S+)", line)
        if m: kb += float(m.group(3)) * (2.0 if m.group(2) == "FETCH_SIZE" else 1.0)
    traffic[os.path.basename(f)[:-8]] = int(round(kb * 1024))
json.dump(traffic, open(os.path.join(root, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
