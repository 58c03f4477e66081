#!/usr/bin/env python3
"""tools/isa_mix.py <file.s> <kernel-name-substring> [--lines]: the static instruction mix of one kernel of a
`hipcc --save-temps` assembly file, by class (what issues on the VALU and how much of it is floating-point arithmetic),
and - with a -g build and --lines - per source line range (the `.loc` directives), which gives the per-stage table
profiles/r03_isa_stages.txt holds.  Static counts: a loop body counts once."""
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
            cur = int(parts[2])
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
            print("  line %5d: %s" % (line, " ".join("%s=%d" % kv for kv in sorted(c.items()))))


if __name__ == "__main__":
    main()
