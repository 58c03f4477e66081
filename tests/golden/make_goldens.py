"""Converts the reference's deterministic Scheme renders (scheme/chapterNN.ppm, P3 text, 320x160) into one small
compressed fixture, tests/golden/scheme_ppm.npz (uint8 [160,320,3] per image).  Data only: pixel values.
Run here (needs /root/reference); the tests read only the .npz.
"""
import os
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/scheme"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scheme_ppm.npz")
names = ["chapter07", "chapter08", "chapter09", "chapter10", "chapter11a", "chapter11b", "chapter13", "chapter16"]
arrays = {}
for n in names:
    with open(os.path.join(REF, n + ".ppm")) as f:
        tok = f.read().split()
    assert tok[0] == "P3" and tok[3] == "255"
    w, h = int(tok[1]), int(tok[2])
    a = np.array(tok[4:], dtype=np.int64)
    assert a.size == w * h * 3, (n, a.size)
    arrays[n] = a.reshape(h, w, 3).astype(np.uint8)
np.savez_compressed(OUT, **arrays)
print("wrote", OUT, os.path.getsize(OUT), "bytes")
