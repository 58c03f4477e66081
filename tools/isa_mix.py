#!/usr/bin/env python3
"""tools/isa_mix.py <file.s> <kernel-name-substring> [--lines | --stages]: the static instruction mix of one kernel of a
`hipcc --save-temps` assembly file, by class (what issues on the VALU and how much of it is floating-point arithmetic),
and - with a -gline-tables-only build - per source line (--lines) or per source FUNCTION (--stages: every instruction
goes to the function of rtmi_device.hpp / rtmi_kernels.hip whose body its `.loc` line lies in; the per-stage table of
profiles/r03_isa_stages.txt).  Static counts: a loop body counts once, every inlined copy counts."""
import collections
import re
import sys

CLASSES = [
    ("pk_f32", re.compile(r"^v_pk_(fma|mul|add)_f32")),
    ("fma_f32", re.compile(r"^v_(fma|fmac|mad|mac)_f32")),
    ("mul_f32", re.compile(r"^v_mul_f32")),
    ("add_f32", re.compile(r"^v_(add|sub|subrev)_f32")),
    ("minmax_f32", re.compile(r"^v_(min|max|min3|max3|med3)_f32")),
    ("trans_f32", re.compile(r"^v_(rcp|rsq|sqrt|exp|log|sin|cos)_f32")),
    ("other_f32", re.compile(r"^v_(floor|fract|trunc|rndne|ceil|ldexp|frexp\w*|div_\w+|cmp_class)_f32")),
    ("f64", re.compile(r"^v_\w+_f64")),
    ("cvt", re.compile(r"^v_cvt")),
    ("cmp", re.compile(r"^v_cmpx?_")),
    ("cndmask", re.compile(r"^v_cndmask")),
    ("mov", re.compile(r"^v_(mov|accvgpr|swap)")),
    ("lane", re.compile(r"^v_(readlane|readfirstlane|writelane|permlane|bpermute|mbcnt)|^ds_(bpermute|permute|swizzle)")),
    ("int_valu", re.compile(r"^v_")),
    ("lds", re.compile(r"^ds_")),
    ("smem", re.compile(r"^s_(load|buffer_load|store|scratch)")),
    ("vmem", re.compile(r"^(global|flat|buffer|scratch)_")),
    ("waitcnt", re.compile(r"^s_(waitcnt|nop|sleep)")),
    ("branch", re.compile(r"^s_(cbranch|branch|setpc|swappc|call|endpgm|barrier)")),
    ("salu", re.compile(r"^s_")),
]
FLOP = {"pk_f32": 4, "fma_f32": 2, "mul_f32": 1, "add_f32": 1, "trans_f32": 1}  # (v_pk_add / v_pk_mul are 2: counted below)


def classify(op):
    for name, rx in CLASSES:
        if rx.match(op):
            return name
    return "other"


def main():
    path, kernel = sys.argv[1], sys.argv[2]
    by_line = "--lines" in sys.argv
    inside = False
    mix = collections.Counter()
    flops = 0
    per_line = collections.defaultdict(collections.Counter)
    cur = 0
    name = None
    for ln in open(path, errors="replace"):
        s = ln.strip()
        if not inside:
            head = s.split(";")[0].strip()
            if head.endswith(":") and kernel in head and not head.startswith("."):
                inside, name = True, head[:-1]
            continue
        if s.startswith(".loc"):
            parts = s.split()
            cur = (int(parts[1]), int(parts[2]))
            continue
        if s.startswith("s_endpgm"):
            mix["branch"] += 1
            break
        if not s or s.startswith((".", ";", "//")) or s.endswith(":"):
            continue
        op = s.split()[0]
        c = classify(op)
        mix[c] += 1
        per_line[cur][c] += 1
        if c == "pk_f32":
            flops += 4 if "fma" in op else 2
        else:
            flops += FLOP.get(c, 0)
    if not inside:
        sys.exit("kernel not found")
    valu = sum(v for k, v in mix.items() if k in ("pk_f32", "fma_f32", "mul_f32", "add_f32", "minmax_f32", "trans_f32", "other_f32", "f64", "cvt", "cmp",
                                                 "cndmask", "mov", "lane", "int_valu"))
    print("kernel", name)
    for k, _ in CLASSES:
        if mix[k]:
            print("  %-11s %6d" % (k, mix[k]))
    print("  VALU total %d, f32 FLOP (static) %d = %.2f per VALU instruction; arithmetic f32 instructions %.0f %% of VALU" % (
        valu, flops, flops / max(valu, 1), 100.0 * sum(mix[k] for k in ("pk_f32", "fma_f32", "mul_f32", "add_f32", "trans_f32")) / max(valu, 1)))
    if by_line:
        for line in sorted(per_line):
            c = per_line[line]
            print("  line %s: %s" % (line, " ".join("%s=%d" % kv for kv in sorted(c.items()))))
    if "--stages" in sys.argv:
        stages(path, per_line)


VALU_CLASSES = ("pk_f32", "fma_f32", "mul_f32", "add_f32", "minmax_f32", "trans_f32", "other_f32", "f64", "cvt", "cmp", "cndmask", "mov", "lane", "int_valu")


def function_ranges(src):
    """(first line, last line, name) of every function body of a source file, by brace counting from lines that look like
    a definition (a crude parser: good enough for this code base's style)."""
    import os
    out = []
    if not os.path.exists(src):
        return out
    lines = open(src, errors="replace").read().split("\n")
    rx = re.compile(r"(?:^|\s)(\w+)\s*\([^;]*$")
    i = 0
    depth = 0
    start = None
    name = None
    pending = None
    for n, ln in enumerate(lines, 1):
        code = ln.split("//")[0]
        if re.match(r"\s*namespace\b", code) or re.match(r"\s*}\s*//\s*namespace", ln) or re.match(r'\s*extern\s+"C"\s*{', code):
            continue
        if depth == 0 and start is None:
            m = re.search(r"\b(\w+)\s*\(", code)
            if m and ("__device__" in code or "__global__" in code or pending) and m.group(1) not in ("if", "for", "while", "switch", "__launch_bounds__", "defined"):
                pending = pending or (n, m.group(1))
            elif "__device__" in code or "__global__" in code:
                pending = (n, None)
            if pending and pending[1] is None and m and m.group(1) not in ("__launch_bounds__",):
                pending = (pending[0], m.group(1))
        opens, closes = code.count("{"), code.count("}")
        if depth == 0 and opens and pending and pending[1]:
            start, name = pending
        depth += opens - closes
        if depth == 0 and start is not None and closes:
            out.append((start, n, name))
            start = None
            pending = None
        if depth == 0 and ";" in code and start is None:
            pending = None
    return out


def stages(path, per_line):
    import os
    files = {}
    base = None
    for ln in open(path, errors="replace"):
        m = re.match(r'\s*\.file\s+(\d+)\s+(?:"([^"]*)"\s+)?"([^"]*)"', ln)
        if m:
            d, f = m.group(2), m.group(3)
            if m.group(1) == "0":
                base = d
            files[int(m.group(1))] = f if os.path.isabs(f) else os.path.join(d if d and os.path.isabs(d) else (base or "."), d if d and not os.path.isabs(d) else "", f)
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raytracing_amd", "csrc")
    ranges = {}
    for idx, f in files.items():
        b = os.path.basename(f)
        if b in ("rtmi_device.hpp", "rtmi_kernels.hip"):
            ranges[idx] = (b, function_ranges(os.path.join(here, b)))
    agg = collections.defaultdict(collections.Counter)
    for (fi, line), c in per_line.items() if all(isinstance(k, tuple) for k in per_line) else []:
        key = "(other headers)"
        if fi in ranges:
            b, rs = ranges[fi]
            key = b + ": (file scope)"
            for a, z, nm in rs:
                if a <= line <= z:
                    key = nm
                    break
        agg[key].update(c)
    rows = []
    for k, c in agg.items():
        valu = sum(c[x] for x in VALU_CLASSES)
        arith = sum(c[x] for x in ("pk_f32", "fma_f32", "mul_f32", "add_f32", "trans_f32"))
        rows.append((valu, k, arith, c))
    print("  per source function (static): VALU | f32 arithmetic (pk fma mul add trans) | cmp cndmask mov lane int | SALU")
    for valu, k, arith, c in sorted(rows, reverse=True):
        print("  %-28s %5d | %4d (%3d %3d %3d %3d %3d) | %4d %4d %4d %4d %4d | %5d" % (
            k[:28], valu, arith, c["pk_f32"], c["fma_f32"], c["mul_f32"], c["add_f32"], c["trans_f32"], c["cmp"], c["cndmask"], c["mov"], c["lane"], c["int_valu"],
            c["salu"] + c["branch"] + c["waitcnt"]))


if __name__ == "__main__":
    main()
