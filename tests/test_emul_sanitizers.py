"""The lane-independent kernels (World::color_at probe, Camera::ray_for_pixel probe) and the scene layout code compiled for the
host (tests/emul: a HIP stand-in, g++ -fsanitize=address,undefined) and run over random rays into every BASELINE scene:
no out-of-bounds access of a scene record, stack or buffer, and f64 results equal to the oracle's.  CPU only; this is the
check a changed kernel passes before it is run on a GPU box."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_trace_kernels_are_clean_under_asan_and_match_the_oracle():
    emul = os.path.join(ROOT, "tests", "emul")
    subprocess.run(["make", "-C", emul], check=True, capture_output=True)
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], check=True, capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:verify_asan_link_order=0", RTMI_NO_TORCH="1",
               RTMI_LIB=os.path.join(emul, "librtmi_emul.so"))
    r = subprocess.run([sys.executable, os.path.join(emul, "run_emul.py"), "400"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "emulation run clean" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
