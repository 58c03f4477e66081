"""Prints BASELINE.md's results table from an all-scenes file of tools/scenes_bench.sh: tools/results_table.py profiles/r01l_all_scenes.json"""
import json, sys
d = json.load(open(sys.argv[1]))
if "rows" in d:   # round-2 summary (profiles/r02_all_scenes.json): one flat row per configuration
    d = {k: {"ms_per_step": v["ms_per_frame"], "value": v["Mrays_s"], "roofline": {"frac": v["frac"]}, "samples": v["samples"], "rays": v["rays"],
             "cpu_baseline": {"value": v["cpu_Mrays_s"], "cores": v["cpu_threads"]}} for k, v in d["rows"].items()}
names = {"ch07": "1 chapter-07", "ch11a": "2 chapter-11a", "ch11b": "2 chapter-11b", "teapot": "3 teapot (BVH leaf 10, aperture 0)",
         "soft": "4 soft shadows", "dof": "4 depth of field", "dice": "5 chapter-16 dice (17x9, seeded)"}
order = ["ch07_1spp", "ch07_shipped", "ch11a_1spp", "ch11a_shipped", "ch11b_1spp", "ch11b_shipped", "teapot_1spp", "teapot_shipped",
         "soft_shipped", "dof_shipped", "dice_1spp", "dice_shipped"]
print("| config | sampling | GPUs | M samples / M rays per frame | ms/frame | Mrays/s | dominant kernel, % FP32 VALU peak | CPU Mrays/s (threads) | parity |\n|---|---|---|---|---|---|---|---|---|")
for k in order:
    x = d[k]; sc, m = k.rsplit("_", 1); b = x.get("cpu_baseline", {})
    par = "f64 mode: all pixels <= 1/255 vs oracle, equal ray counts; f32 >= 99.8 % px" if m == "1spp" else "PSNR vs oracle: f64 >= 60 dB, f32 >= 35 dB"
    work = "%.1f / %.1f" % (x["samples"] / 1e6, x["rays"] / 1e6) if "samples" in x else "-"
    print("| %s | %s | 1 | %s | %.2f | %s | %.1f | %.1f (%d) | %s |" % (names[sc], "1 spp" if m == "1spp" else "as shipped", work, x["ms_per_step"],
          format(int(round(x["value"])), ","), x["roofline"]["frac"] * 100, b.get("value", 0), b.get("cores", 0), par))
