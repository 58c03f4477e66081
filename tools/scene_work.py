"""tools/scene_work.py <scene> [w h] : work counters per ray of one frame (1 spp and as shipped) - leaf tests, node visits,
slab tests, algorithmic FLOPs per ray - to see where a scene's time can go."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import abi, scenes
name = sys.argv[1]
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (960, 540)
for det in (True, False):
    world, cam = scenes.BASELINE_SCENES[name](w, h, deterministic=det)
    sc = abi.Scene(world)
    img, st = abi.render_host(sc, cam, abi.make_opts())
    sc.close()
    rays = st["primary"] + st["reflection"] + st["refraction"] + st["shadow"]
    print("%s %s: rays %d (p %d rl %d rf %d sh %d) samples %d | per ray: leaf tests %.1f, nodes %.1f, slabs %.1f, flops %.0f | %.2f ms" % (
        name, "1spp" if det else "shipped", rays, st["primary"], st["reflection"], st["refraction"], st["shadow"], st["samples"],
        st["leaf_tests"] / rays, st["node_visits"] / rays, st["aabb_tests"] / rays, st["flops"] / rays, st["kernel_ms"]))
