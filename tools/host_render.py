"""tools/host_render.py : the PCIe-inclusive rate of rtmi_render (host buffer out: kernels + device->host copy), headline frame."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import abi, scenes
world, cam = scenes.chapter11a(3840, 2160)
sc = abi.Scene(world)
abi.render_host(sc, cam, abi.make_opts())
t0 = time.perf_counter(); n = 5
for _ in range(n):
    img, st = abi.render_host(sc, cam, abi.make_opts())
dt = (time.perf_counter() - t0) / n
rays = st["primary"] + st["reflection"] + st["refraction"] + st["shadow"]
print("rtmi_render, chapter-11a as shipped at 3840x2160: %.1f ms per frame (kernels %.1f ms), %.0f Mrays/s PCIe-inclusive" % (dt * 1e3, st["kernel_ms"], rays / dt / 1e6))
