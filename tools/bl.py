"""tools/bl.py [file ...] : one line per bench.py JSON line (stdin or files): rate, ms per step, per-kernel ms and fractions."""
import json, sys
for src in (sys.argv[1:] or ["-"]):
    txt = sys.stdin.read() if src == "-" else open(src).read()
    for line in txt.splitlines():
        if not line.startswith("{"): continue
        d = json.loads(line); r = d["roofline"]; k = list(r["per_kernel"].values())
        print("%s %s: %.0f Mrays/s  %.3f ms/step | first pass %.3f ms frac %.4f | second %.3f ms | call alone %.3f ms frac %.4f | %.1f GFLOP %d rays %d samples" % (
            d["config"]["scene"], "n%d" % d["n_gpus"], d["value"], d["ms_per_step"], k[0]["ms"], k[0]["frac"], k[1]["ms"], r["call"]["ms_alone"], r["call"]["frac_alone"],
            r["call"]["algorithmic_flops"] / 1e9, d["config"]["rays_per_step"], d["config"]["samples_per_step"]))
        if "pcie_inclusive" in d: print("   host frame %.2f ms" % d["pcie_inclusive"]["ms_per_frame"])
        for kk, v in (d.get("all_configs") or {}).get("rows", {}).items():
            print("   %-15s %s" % (kk, " ".join("%s=%s" % (a, b) for a, b in v.items() if a in ("ms_per_frame", "Mrays_s", "frac", "ms_call_alone", "error"))))
