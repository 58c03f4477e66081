"""rtmi_trace_photons of the HOST EMULATION build (ASan + UBSan) against the oracle's orc_trace_photons.  Usage as for
run_emul.py:

    make -C tests/emul && LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 \
        RTMI_NO_TORCH=1 RTMI_LIB=$PWD/tests/emul/librtmi_emul.so python tests/emul/run_emul_photons.py [n_emit]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
assert "emul" in os.environ.get("RTMI_LIB", ""), "set RTMI_LIB to the emulation library"
import oracle_lib as orc  # noqa: E402
import photon_trace_cases as cases  # noqa: E402
from raytracing_amd import abi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
print(abi.describe())
for name, make, flags in cases.CASES:
    world = make()
    f = world.flatten()
    sc = abi.Scene(world)
    for prec in (1, 0):
        opts = abi.make_photon_opts(n, seed=11, precision=prec, **flags)
        ref, rst = orc.trace_photons(f, opts)
        got, gst = abi.trace_photons(sc, opts)
        print("%-12s %s emitted %d  stored %d / %d  rays %d / %d  %s" % (name, "f64" if prec else "f32", n, gst["n_stored"], rst["n_stored"], gst["n_rays"],
                                                                         rst["n_rays"], cases.compare(got, gst, ref, rst, exact=bool(prec))), flush=True)
    # a map in shares (rtmi_photon_opts.first / count): one after the other they are the whole call's photons
    whole, _ = abi.trace_photons(sc, abi.make_photon_opts(n, seed=11, **flags))
    parts = [abi.trace_photons(sc, abi.make_photon_opts(n, seed=11, first=a, count=c, **flags))[0] for a, c in ((0, n // 3), (n // 3, n - n // 3 - 7), (n - 7, 0))]
    assert np.array_equal(np.concatenate(parts), whole)
    sc.close()
print("emulation run clean")
