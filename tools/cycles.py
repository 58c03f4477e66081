"""tools/cycles.py <scene> [w h] [stages] : where a wave's time goes, from a -DRTMI_CYCLES build of the library
(make -C raytracing_amd/csrc BUILD=build_cyc OUT=../librtmi_cyc.so EXTRA=-DRTMI_CYCLES=1 ../librtmi_cyc.so; RTMI_LIB=.../librtmi_cyc.so).
In that build the counters aabb_tests / leaf_tests / node_visits / flops hold shader-clock cycles summed over waves:
  =1          the tree walk (packet_scene, CSG roots included), csg_root's leaf loop, its sort + filter, the whole kernel;
  =2 `stages` the closest-hit queries (World::intersect + hit), the shadow queries (is_shadowed), the container queries (n1 / n2),
              the whole kernel - the rest is camera rays, Philox, prepare_computations, lighting, sums and scheduling."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import abi, scenes
name = sys.argv[1]
stages = "stages" in sys.argv
if stages: sys.argv.remove("stages")
spawn = "spawn" in sys.argv  # a -DRTMI_CYCLES=4 build
if spawn: sys.argv.remove("spawn")
csg = "csg" in sys.argv  # a -DRTMI_CYCLES=5 build: csg_root's bound pass, its full leaf tests, its sort + filter
if csg: sys.argv.remove("csg")
tris = "tris" in sys.argv  # a -DRTMI_CYCLES=6 build: packet_tris' box-pair steps, triangle runs, pops
if tris: sys.argv.remove("tris")
shade = "shade" in sys.argv  # a -DRTMI_CYCLES=3 build: camera ray + Philox, prepare_computations + material + pattern, lighting()
if shade: sys.argv.remove("shade")
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
for det in (True, False):
    world, cam = scenes.BASELINE_SCENES[name](w, h, deterministic=det)
    sc = abi.Scene(world)
    img, st = abi.render_host(sc, cam, abi.make_opts())
    sc.close()
    tot = float(st["flops"])
    if spawn:
        a, b, c = st["aabb_tests"] / tot, st["leaf_tests"] / tot, st["node_visits"] / tot
        print("%s %s: %.2f ms | of the kernels' wave cycles: spawn logic after a hit %.1f %%, popping a pending ray %.1f %%, the sample loop's own work (sums, stop test, policy, queues) %.1f %%" % (
            name, "1spp" if det else "shipped", st["kernel_ms"], 100 * a, 100 * b, 100 * c))
        continue
    if csg:
        a, b, c = st["aabb_tests"] / tot, st["leaf_tests"] / tot, st["node_visits"] / tot
        print("%s %s: %.2f ms | of the kernels' wave cycles: csg_root bound pass %.1f %%, full leaf tests %.1f %%, sort + filter %.1f %%" % (
            name, "1spp" if det else "shipped", st["kernel_ms"], 100 * a, 100 * b, 100 * c))
        continue
    if tris:
        a, b, c = st["aabb_tests"] / tot, st["leaf_tests"] / tot, st["node_visits"] / tot
        print("%s %s: %.2f ms | of the kernels' wave cycles: packet_tris box-pair steps %.1f %%, triangle runs %.1f %%, pops %.1f %%" % (
            name, "1spp" if det else "shipped", st["kernel_ms"], 100 * a, 100 * b, 100 * c))
        continue
    if shade:
        a, b, c = st["aabb_tests"] / tot, st["leaf_tests"] / tot, st["node_visits"] / tot
        print("%s %s: %.2f ms | of the kernels' wave cycles: camera ray + Philox %.1f %%, prepare_computations + material + pattern %.1f %%, lighting() %.1f %%" % (
            name, "1spp" if det else "shipped", st["kernel_ms"], 100 * a, 100 * b, 100 * c))
        continue
    if stages:
        a, b, c = st["aabb_tests"] / tot, st["leaf_tests"] / tot, st["node_visits"] / tot
        rays = st["primary"] + st["reflection"] + st["refraction"] + st["shadow"]
        print("%s %s: %.2f ms | of the kernels' wave cycles: closest-hit queries %.1f %%, shadow queries %.1f %%, container queries %.1f %%, "
              "everything else (camera ray, Philox, prepare, lighting, sums, scheduling) %.1f %%" % (
                  name, "1spp" if det else "shipped", st["kernel_ms"], 100 * a, 100 * b, 100 * c, 100 * (1 - a - b - c)))
        continue
    print("%s %s: %.2f ms | of the kernels' wave cycles: tree walk %.1f %% (csg leaf loop %.1f %%, csg sort+filter %.1f %%), rest (top level, shading, sampling) %.1f %%" % (
        name, "1spp" if det else "shipped", st["kernel_ms"], 100 * st["aabb_tests"] / tot, 100 * st["leaf_tests"] / tot,
        100 * st["node_visits"] / tot, 100 * (1 - st["aabb_tests"] / tot)))
