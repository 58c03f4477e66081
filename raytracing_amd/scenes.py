"""The BASELINE scenes, restated with the front end (same builder calls as the reference's example programs),
and the Scheme twin's deterministic scenes used as golden images.

Each function returns (World, Camera).  Citations: rust/examples/*.rs and scheme/chapterNN.scm of the reference.
Cameras take (hsize, vsize); samplers are set as the cited example ships them unless `deterministic=True`
(min_samples 1, standard error 1.0 -> exactly one centred ray per pixel, camera.rs:378-389).
"""
import math
import os

from . import _front as rt

PI = math.pi
_DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data")


def _camera(h, v, frm, to, se=None, min_samples=None, deterministic=False, fov=PI / 3.0):
    cam = rt.Camera(h, v, fov).with_view_transform(rt.point(*frm), rt.point(*to), rt.vector(0, 1, 0))
    if deterministic:
        cam.set_min_samples(1)
        cam.set_allowed_standard_error(1.0)
    else:
        if se is not None:
            cam.set_allowed_standard_error(se)
        if min_samples is not None:
            cam.set_min_samples(min_samples)
    return cam


# ------------------------------------------------------------------------------------------------
# Rust examples (BASELINE configs 1-5)
# ------------------------------------------------------------------------------------------------
def chapter07(h=900, v=450, deterministic=False):
    """rust/examples/chapter-07.rs:13-83 — BASELINE config 1."""
    w = rt.World.empty()
    w.add_light(rt.AmbientLight(rt.color(0.1, 0.1, 0.1)))
    w.add_light(rt.PointLight(rt.point(-10, 10, -10), rt.color(1, 1, 1)))
    floor_material = rt.Phong.default().with_color(rt.color(1, 0.9, 0.9)).with_specular(0.0).with_shininess(100.0)
    w.add_item(rt.sphere().with_transform(rt.scaling(10, 0.01, 10)).with_material(floor_material))
    w.add_item(rt.sphere().with_transform(
        rt.translation(0, 0, 5) * rt.rotation_y(-PI / 4.0) * rt.rotation_x(-PI / 2.0) * rt.scaling(10, 0.01, 10)).with_material(floor_material))
    w.add_item(rt.sphere().with_transform(
        rt.translation(0, 0, 5) * rt.rotation_y(PI / 4.0) * rt.rotation_x(PI / 2.0) * rt.scaling(10, 0.01, 10)).with_material(floor_material))
    w.add_item(rt.sphere().with_transform(rt.translation(-0.5, 1, 0.5)).with_material(
        rt.Phong.default().with_rgb(0.1, 1.0, 0.5).with_diffuse(0.7).with_specular(0.3)))
    w.add_item(rt.sphere().with_transform(rt.translation(1.5, 0.5, -0.5) * rt.scaling(0.5, 0.5, 0.5)).with_material(
        rt.Phong.default().with_rgb(0.5, 1.0, 0.1).with_diffuse(0.7).with_specular(0.3)))
    w.add_item(rt.sphere().with_transform(rt.translation(-1.5, 0.33, -0.75) * rt.scaling(0.33, 0.33, 0.33)).with_material(
        rt.Phong.default().with_rgb(1.0, 0.8, 0.1).with_diffuse(0.7).with_specular(0.3)))
    return w, _camera(h, v, (0, 1.5, -5), (0, 1, 0), se=1e-2, deterministic=deterministic)


def _sky():
    return rt.plane().with_transform(rt.translation(0, 1000, 0)).with_material(
        rt.Phong.default().with_color(rt.color(0.8, 0.8, 1)).with_emissive(1.0).with_diffuse(1.0).with_specular(0.0))


def chapter11a(h=900, v=450, deterministic=False, max_depth=5):
    """rust/examples/chapter-11a.rs:14-82 — BASELINE config 2 (depth 5; the Rust default is 10, world.rs:61)."""
    w = rt.World.empty()
    w.set_max_reflection_depth(max_depth)
    w.add_light(rt.PointLight(rt.point(-10, 10, -10), rt.color(1, 1, 1)))
    floor_material = rt.Phong.default().with_pattern(
        rt.checkers_pattern(rt.color(0.75, 0.25, 0.5), rt.color(0.25, 0.75, 0.5))).with_diffuse(0.9).with_specular(0.0)
    w.add_item(rt.plane().with_material(floor_material))
    w.add_item(_sky())
    w.add_item(rt.sphere().with_transform(rt.translation(-1.5, 1, 0.5)).with_material(
        rt.Phong.default().with_color(rt.color(0, 0, 0)).with_diffuse(0.0).with_specular(0.8).with_reflective(1.0)))
    w.add_item(rt.sphere().with_transform(rt.translation(1.5, 1, 0.5)).with_material(
        rt.Phong.default().with_color(rt.color(0, 0, 0)).with_diffuse(0.0).with_specular(0.8).with_reflective(1.0)
        .with_transparency(1.0).with_refractive_index(1.5)))
    w.add_item(rt.sphere().with_transform(rt.translation(1.5, 1, 0.5) * rt.scaling(0.5, 0.5, 0.5)).with_material(
        rt.Phong.default().with_color(rt.color(0, 0, 0)).with_diffuse(0.0).with_specular(0.8).with_reflective(1.0)
        .with_transparency(1.0).with_refractive_index(1.0)))
    return w, _camera(h, v, (0, 1.5, -5), (0, 1, 0), se=1e-2, deterministic=deterministic)


def chapter11b(h=900, v=450, deterministic=False, max_depth=5):
    """rust/examples/chapter-11b.rs:13-87 — BASELINE config 2, water variant."""
    w = rt.World.empty()
    w.set_max_reflection_depth(max_depth)
    w.add_light(rt.PointLight(rt.point(-10, 10, -10), rt.color(1, 1, 1)))
    floor_material = rt.Phong.default().with_pattern(
        rt.checkers_pattern(rt.color(0.25, 0.25, 0.75), rt.color(0.25, 0.75, 0.25))).with_diffuse(0.9).with_specular(0.0)
    w.add_item(rt.plane().with_material(floor_material).with_transform(rt.translation(0, -1, 0)))
    water_material = (rt.Phong.default().with_color(rt.color(0.1, 0.1, 0.5)).with_diffuse(1.0).with_specular(0.5)
                      .with_reflective(1.0).with_transparency(1.0).with_refractive_index(1.3))
    w.add_item(rt.plane().with_cast_shadow(False).with_material(water_material).with_transform(rt.translation(0, 0, 0)))
    w.add_item(_sky())
    w.add_item(rt.sphere().with_transform(rt.translation(-1.5, 1, 5.5)).with_material(
        rt.Phong.default().with_color(rt.color(1, 0, 0)).with_diffuse(0.5).with_specular(0.8)))
    w.add_item(rt.sphere().with_transform(rt.translation(1.5, -0.25, 0.5)).with_material(
        rt.Phong.default().with_color(rt.color(0, 1, 0)).with_diffuse(0.1).with_specular(0.8)))
    return w, _camera(h, v, (0, 1.5, -5), (0, 1, 0), se=1e-2, deterministic=deterministic)


def teapot_obj_text(lite=False):
    path = os.path.join(_DATA, "teapot-lite.obj" if lite else "teapot.obj")
    with open(path, "r") as f:
        return f.read()


def chapter15(h=900, v=450, deterministic=False, lite=False, depth_of_field=False, max_leaf=10):
    """rust/examples/chapter-15.rs:15-57 — BASELINE config 3: teapot mesh, host-built BVH (leaf <= 10),
    hard shadows.  The example also switches on depth of field; config 3 does not (`depth_of_field=False`)."""
    w = rt.World.empty()
    w.add_light(rt.AmbientLight(rt.color(0.25, 0.25, 0.25)))
    w.add_light(rt.PointLight(rt.point(-10, 10, -10), rt.color(0.75, 0.75, 0.75)))
    floor_material = rt.Phong.default().with_pattern(
        rt.checkers_pattern(rt.color(0.25, 0.25, 0.75), rt.color(0.25, 0.75, 0.25))).with_diffuse(0.9).with_specular(0.0)
    w.add_item(rt.plane().with_material(floor_material).with_transform(rt.translation(0, -1, 0)))
    teapot = rt.ObjParser.parse_str(teapot_obj_text(lite)).into_group()
    teapot = teapot.with_transform(rt.rotation_y(PI / 4.0) * rt.rotation_x(-PI / 2.0) * rt.scaling(0.2, 0.2, 0.2))
    w.add_item(rt.build_bounding_tree(teapot, max_leaf))
    w.finalize_scene()
    frm, to = (0, 5, -8), (0, 1, 0)
    cam = _camera(h, v, frm, to, se=1e-2, min_samples=10, deterministic=deterministic)
    if depth_of_field:
        cam.set_focal_distance(math.sqrt(sum((a - b) ** 2 for a, b in zip(to, frm))))
        cam.set_aperture_size(0.1)
    return w, cam


def lights_gallery(h=300, v=150, deterministic=False, random_lights=True):
    """Not a reference example: every light kind the Rust API offers (lights.rs: Point, Ambient, RealisticPoint, Beam and,
    if `random_lights`, Sphere and Disc) over every primitive kind, so that the device path of each is compared with the
    oracle (the reference itself has unit tests for Beam only).  (The stripes are shifted so that no cube face lies exactly
    on a stripe boundary, where floor() of a rounded coordinate is a coin toss in any precision.)"""
    w = rt.World.empty()
    w.add_light(rt.AmbientLight(rt.color(0.05, 0.05, 0.05)))
    w.add_light(rt.PointLight(rt.point(-6, 8, -6), rt.color(0.35, 0.3, 0.3)))
    w.add_light(rt.RealisticPointLight(rt.point(4, 5, -3), rt.color(40, 40, 60)))
    w.add_light(rt.Beam(rt.point(0, 9, 0), rt.vector(0, 0, 1.5), rt.vector(1.5, 0, 0), rt.color(0.5, 0.6, 0.3)))
    if random_lights:
        w.add_light(rt.SphereLight(rt.point(-3, 4, -4), 0.7, rt.color(0.3, 0.3, 0.3)))
        w.add_light(rt.DiscLight(rt.point(3, 6, 2), rt.vector(-0.3, -1, -0.2).normalized(), 1.2, rt.color(60, 50, 40)))
    floor = rt.Phong.default().with_pattern(rt.ring_pattern(rt.color(0.9, 0.9, 0.9), rt.color(0.5, 0.6, 0.7))).with_specular(0.1)
    w.add_item(rt.plane().with_material(floor))
    w.add_item(rt.sphere().with_transform(rt.translation(-2.5, 1, 0) * rt.scaling(1, 1, 0.6)).with_material(
        rt.Phong.default().with_rgb(0.9, 0.3, 0.2).with_reflective(0.2)))
    w.add_item(rt.cube().with_transform(rt.translation(0, 0.75, 1.5) * rt.rotation_y(0.6) * rt.scaling(0.75, 0.75, 0.75)).with_material(
        rt.Phong.default().with_pattern(rt.stripe_pattern(rt.color(0.2, 0.7, 0.3), rt.color(0.9, 0.9, 0.2)).with_transform(rt.translation(0.13, 0, 0) * rt.scaling(0.25, 1, 1)))))
    w.add_item(rt.cylinder(0.0, 1.5, True).with_transform(rt.translation(2.5, 0, 0) * rt.scaling(0.6, 1, 0.6)).with_material(
        rt.Phong.default().with_rgb(0.2, 0.4, 0.9).with_shininess(50.0)))
    w.add_item(rt.cone(-1.0, 0.0, True).with_transform(rt.translation(0.5, 1.0, -1.5) * rt.scaling(0.6, 1, 0.6)).with_material(
        rt.Phong.default().with_rgb(0.8, 0.6, 0.1)))
    w.add_item(rt.triangle(rt.point(-4, 0, 3), rt.point(-1, 0, 4), rt.point(-2.5, 3, 3.5)).with_material(
        rt.Phong.default().with_rgb(0.7, 0.2, 0.7)))
    w.add_item(rt.glass_sphere().with_transform(rt.translation(1.2, 0.5, -2.5) * rt.scaling(0.5, 0.5, 0.5)))
    w.finalize_scene()
    return w, _camera(h, v, (0.5, 4.5, -9), (0, 1, 0), se=2e-2, min_samples=4, deterministic=deterministic)


def crowd(h=300, v=150, deterministic=False):
    """Not a reference example: a FLAT world (World::objects all primitives, world.rs:145-156 brute force) that is larger than
    every table the flat-scene kernels stage in LDS - 46 top-level primitives of every kind (the block holds 16), 24
    materials (16), 9 patterned leaves of all four pattern kinds, some of them past the 16th leaf (4 folded pattern slots)
    and 6 lights (4) - so that the paths a small scene never takes (leaves from global records, materials, lights and
    patterns from global memory, a non-shadowing leaf among the pairs) are compared with the oracle too.  Positions and
    sizes come from a fixed Philox stream; nothing touches anything else exactly (no ties in t)."""
    w = rt.World.empty()
    w.set_max_reflection_depth(4)
    w.add_light(rt.AmbientLight(rt.color(0.06, 0.06, 0.06)))
    w.add_light(rt.PointLight(rt.point(-8, 9, -7), rt.color(0.3, 0.3, 0.3)))
    w.add_light(rt.PointLight(rt.point(7, 8, -5), rt.color(0.25, 0.22, 0.2)))
    w.add_light(rt.PointLight(rt.point(0, 12, 4), rt.color(0.2, 0.2, 0.25)))
    w.add_light(rt.RealisticPointLight(rt.point(-2, 6, -6), rt.color(25, 25, 20)))
    w.add_light(rt.PointLight(rt.point(3, 3, -9), rt.color(0.15, 0.15, 0.15)))
    u = lambda i, k: _philox_uniform(0xC20D, i, k)
    pats = [lambda a, b: rt.stripe_pattern(a, b).with_transform(rt.translation(0.07, 0, 0) * rt.scaling(0.3, 1, 1)),
            lambda a, b: rt.gradient_pattern(a, b).with_transform(rt.translation(-1, 0, 0) * rt.scaling(2, 1, 1)),
            lambda a, b: rt.ring_pattern(a, b).with_transform(rt.scaling(0.23, 1, 0.23)),
            lambda a, b: rt.checkers_pattern(a, b).with_transform(rt.translation(0.011, 0.013, 0.017) * rt.scaling(0.41, 0.41, 0.41))]
    mats = []
    for m in range(24):
        col = rt.color(0.25 + 0.7 * u(m, 0), 0.25 + 0.7 * u(m, 1), 0.25 + 0.7 * u(m, 2))
        ph = rt.Phong.default().with_color(col).with_diffuse(0.6 + 0.3 * u(m, 3)).with_specular(0.5 * u(m, 4)).with_shininess(20.0 + 200.0 * u(m, 5))
        if m % 8 == 3:
            ph = ph.with_reflective(0.35)
        if m == 13:
            ph = rt.Phong.default().with_rgb(0.1, 0.1, 0.1).with_diffuse(0.2).with_transparency(0.85).with_reflective(0.5).with_refractive_index(1.45)
        if m in (0, 2, 5, 9, 11, 14, 17, 20, 22):
            ph = ph.with_pattern(pats[m % 4](col, rt.color(0.95 - 0.5 * u(m, 6), 0.9 - 0.5 * u(m, 7), 0.85 - 0.5 * u(m, 8))))
        mats.append(ph)
    w.add_item(rt.plane().with_material(mats[0]))                                                                      # patterned, leaf 0
    w.add_item(rt.plane().with_transform(rt.translation(0, 0, 14) * rt.rotation_x(PI / 2.0)).with_material(mats[1]))
    k = 2
    for i in range(44):
        x, z = -7.0 + 14.0 * u(100 + i, 0), -3.0 + 12.0 * u(100 + i, 1)
        r = 0.25 + 0.45 * u(100 + i, 2)
        y = r + 2.2 * u(100 + i, 3) * (1 if i % 3 else 0)
        mat = mats[k % 24]
        kind = i % 6
        if kind == 0:
            item = rt.sphere().with_transform(rt.translation(x, y, z) * rt.scaling(r, r, r))                           # LEAF_QUICK sphere
        elif kind == 1:
            item = rt.sphere().with_transform(rt.translation(x, y, z) * rt.rotation_z(0.4) * rt.scaling(r, 0.6 * r, 1.3 * r))  # an ellipsoid: global record
        elif kind == 2:
            item = rt.cube().with_transform(rt.translation(x, y, z) * rt.rotation_y(3.0 * u(100 + i, 4)) * rt.scaling(r, r, r))
        elif kind == 3:
            item = rt.cylinder(-1.0, 1.0, True).with_transform(rt.translation(x, y, z) * rt.scaling(0.7 * r, r, 0.7 * r))
        elif kind == 4:
            item = rt.cone(-1.0, 0.0, True).with_transform(rt.translation(x, y + r, z) * rt.scaling(r, 2 * r, r))
        else:
            item = rt.triangle(rt.point(x - r, y - r, z), rt.point(x + r, y - r, z + 0.3 * r), rt.point(x, y + r, z - 0.2 * r))
        item = item.with_material(mat)
        if i in (6, 30):
            item = item.with_cast_shadow(False)                                                                        # (a quick sphere and a late leaf that cast no shadow)
        w.add_item(item)
        k += 1
    return w, _camera(h, v, (0.5, 5.5, -11), (0, 1.2, 2), se=2e-2, min_samples=4, deterministic=deterministic)


def yard(h=300, v=150, deterministic=False, bounded=True):
    """Not a reference example: one Group (or, `bounded`, the BoundingGroup tree build_bounding_tree makes of it) that holds
    every kind of item a subtree can - spheres, cubes, closed and open finite cylinders, cones, flat and smooth triangles, a
    nested Group, CSG unions / intersections / differences (one with a glass part) - plus an unbounded Group (an infinite
    cylinder) next to it, so that the device's own BVH over such subtrees (DBvh2 entries of every kind) is compared with
    the oracle's walk of the reference's tree."""
    w = rt.World.empty()
    w.set_max_reflection_depth(4)
    w.add_light(rt.PointLight(rt.point(-6, 9, -8), rt.color(0.9, 0.9, 0.9)))
    w.add_light(rt.AmbientLight(rt.color(0.1, 0.1, 0.1)))
    w.add_item(rt.plane().with_material(rt.Phong.default().with_pattern(
        rt.checkers_pattern(rt.color(0.3, 0.3, 0.3), rt.color(0.8, 0.8, 0.8)).with_transform(rt.translation(0.07, 0, 0.03))).with_specular(0.0).with_reflective(0.1)))
    g = rt.group()
    k = 0
    for ix in range(-3, 4):
        for iz in range(0, 4):
            u = [_philox_uniform(0xBEEF, k, j) for j in range(6)]
            k += 1
            mat = rt.Phong.default().with_hsv(u[0] * 360.0, 0.7, 0.9).with_reflective(0.15 if k % 5 == 0 else 0.0)
            place = rt.translation(ix * 1.6 + (u[1] - 0.5) * 0.4, 0, iz * 1.7 + (u[2] - 0.5) * 0.4)
            s = 0.35 + 0.25 * u[3]
            kind = k % 9
            if kind == 0:
                g.add_child(rt.sphere().with_material(mat).with_transform(place * rt.translation(0, s, 0) * rt.scaling(s, s, s)))
            elif kind == 1:
                g.add_child(rt.cube().with_material(mat).with_transform(place * rt.translation(0, s, 0) * rt.rotation_y(u[4] * 3.0) * rt.scaling(s, s, s)))
            elif kind == 2:
                g.add_child(rt.cylinder(0.0, 1.0 + u[4], True).with_material(mat).with_transform(place * rt.scaling(s, 1, s)))
            elif kind == 3:
                g.add_child(rt.cone(-1.0, 0.0, True).with_material(mat).with_transform(place * rt.translation(0, 1, 0) * rt.scaling(s, 1, s)))
            elif kind == 4:
                g.add_child(rt.cylinder(0.2, 1.2, False).with_material(mat).with_transform(place * rt.rotation_z(0.3) * rt.scaling(s, 1, s)))
            elif kind == 5:
                a = rt.cube().with_material(mat).with_transform(rt.scaling(s, s, s))
                b = rt.sphere().with_material(rt.Phong.default().with_hsv(u[0] * 360.0 + 180.0, 0.7, 0.9)).with_transform(rt.translation(s * 0.6, s * 0.6, -s * 0.6) * rt.scaling(s, s, s))
                g.add_child(rt.csg_difference(a, b).with_transform(place * rt.translation(0, s, 0) * rt.rotation_y(u[4])))
            elif kind == 6:
                a = rt.sphere().with_material(mat).with_transform(rt.translation(-0.3 * s, 0, 0) * rt.scaling(s, s, s))
                b = rt.glass_sphere().with_transform(rt.translation(0.3 * s, 0, 0) * rt.scaling(s, s, s))
                op = rt.csg_union if k % 2 else rt.csg_intersection
                g.add_child(op(a, b).with_transform(place * rt.translation(0, s, 0)))
            elif kind == 7:
                sub = rt.group()
                p0, p1, p2, p3 = rt.point(-s, 0, -s), rt.point(s, 0, -s), rt.point(0, 0, s), rt.point(0, 1.5 * s, 0)
                for (a, b, c) in ((p0, p1, p3), (p1, p2, p3), (p2, p0, p3)):
                    sub = sub.with_child(rt.triangle(a, b, c).with_material(mat))
                sub = sub.with_child(rt.sphere().with_material(mat).with_transform(rt.translation(0, 1.5 * s, 0) * rt.scaling(0.2 * s, 0.2 * s, 0.2 * s)))
                g.add_child(rt.SceneItem(sub).with_transform(place))
            else:
                n = rt.vector(0, 1, 0)
                g.add_child(rt.smooth_triangle(rt.point(-s, 0.1, 0), rt.point(s, 0.1, 0), rt.point(0, 0.1 + 2 * s, 0.3 * s),
                                               rt.vector(-0.4, 0.2, -1).normalized(), rt.vector(0.4, 0.2, -1).normalized(), n).with_material(mat).with_transform(place))
    w.add_item(rt.build_bounding_tree(g, 2) if bounded else rt.SceneItem(g))
    loose = rt.group().with_child(rt.cylinder().with_material(rt.Phong.default().with_rgb(0.6, 0.6, 0.7)).with_transform(
        rt.translation(6.5, 0, 6) * rt.scaling(0.3, 1, 0.3))).with_child(rt.sphere().with_material(rt.Phong.default().with_rgb(0.9, 0.8, 0.2)).with_transform(
        rt.translation(-6.5, 0.5, 5) * rt.scaling(0.5, 0.5, 0.5)))
    w.add_item(rt.SceneItem(loose))
    w.finalize_scene()
    return w, _camera(h, v, (0.5, 5.0, -7.5), (0, 0.6, 2.5), se=2e-2, min_samples=4, deterministic=deterministic)


def photon_room(h=300, v=150, deterministic=False, direct=True):
    """Not a reference example (examples/photon-map.rs is `photon_box` below): a floor, a back
    wall, a matt sphere, a mirror sphere and a glass sphere under one point light, to which a test attaches a photon map
    (World.set_photon_arrays) so that the gather of Phong::shade_hit (materials.rs:376-384) runs on primary, reflected and
    refracted hits.  `direct` False: World::enable_direct_illumination(false), the map is all the light there is."""
    w = rt.World.empty()
    w.set_max_reflection_depth(4)
    w.add_light(rt.PointLight(rt.point(-3, 6, -4), rt.color(0.8, 0.8, 0.8)))
    w.add_item(rt.plane().with_material(rt.Phong.default().with_rgb(0.8, 0.8, 0.75).with_specular(0.0)))
    w.add_item(rt.plane().with_transform(rt.translation(0, 0, 5) * rt.rotation_x(PI / 2.0)).with_material(
        rt.Phong.default().with_pattern(rt.stripe_pattern(rt.color(0.7, 0.3, 0.3), rt.color(0.3, 0.3, 0.7)).with_transform(rt.translation(0.11, 0, 0))).with_specular(0.0)))
    w.add_item(rt.sphere().with_transform(rt.translation(-1.6, 1, 1)).with_material(rt.Phong.default().with_rgb(0.9, 0.6, 0.2).with_specular(0.1)))
    w.add_item(rt.sphere().with_transform(rt.translation(0.9, 0.8, 0.2) * rt.scaling(0.8, 0.8, 0.8)).with_material(
        rt.Phong.default().with_rgb(0.1, 0.1, 0.1).with_reflective(0.8).with_specular(0.6)))
    w.add_item(rt.glass_sphere().with_transform(rt.translation(-0.2, 0.5, -1.6) * rt.scaling(0.5, 0.5, 0.5)))
    w.enable_direct_illumination(direct)
    w.finalize_scene()
    return w, _camera(h, v, (0, 2.2, -6.5), (0, 0.8, 0), se=2e-2, min_samples=4, deterministic=deterministic)


def synthetic_photons(n, seed=7):
    """A map for photon_room without tracing anything: n photons where the room's point light would leave direct ones - on
    the floor and the back wall, seeded - as the four arrays World.set_photon_arrays takes.  (numpy only: the tests' maps
    come from the oracle's ray casts, tests/photon_cases.py; this one is for measuring the gather on millions of photons.)"""
    import numpy as np
    rng = np.random.default_rng(seed)
    n_floor = int(0.7 * n)
    pos = np.empty((n, 3), dtype=np.float64)
    pos[:n_floor] = np.column_stack([rng.uniform(-8, 8, n_floor), np.zeros(n_floor), rng.uniform(-8, 5, n_floor)])
    pos[n_floor:] = np.column_stack([rng.uniform(-8, 8, n - n_floor), rng.uniform(0, 8, n - n_floor), np.full(n - n_floor, 5.0)])
    to_light = np.array([-3.0, 6.0, -4.0]) - pos
    to_light /= np.linalg.norm(to_light, axis=1, keepdims=True)
    scale = (rng.uniform(0.5, 1.5, n) * 250.0 / n).astype(np.float32)
    rgb = np.tile(np.array([255, 242, 229], dtype=np.uint8), (n, 1))
    return pos.astype(np.float32), to_light.astype(np.float32), scale, rgb


def photon_room_with_map(h=300, v=150, deterministic=False, n_photons=2_000_000, n_nearest=100, max_search_radius=0.3):
    """photon_room + synthetic_photons: the photon-map gather at the scale of examples/photon-map.rs's 100 nearest photons"""
    w, cam = photon_room(h, v, deterministic=deterministic)
    w.set_photon_arrays(*synthetic_photons(n_photons), n_nearest, max_search_radius)
    return w, cam


def photon_box(h=600, v=600, deterministic=False, direct=True):
    """rust/examples/photon-map.rs:15-78: a closed room of six planes (white, one red and one green wall) under a disc light
    of 1000 at the ceiling, a rotated box and a glass ball; SE 1e-2, at least 100 samples.  Its materials are the ones
    Phong::photon_hit accepts (diffuse 0.5, no specular term: materials.rs:443-445), so this is the scene
    World::compute_photon_map (rtmi_trace_photons) is exercised and measured on."""
    w = rt.World.empty()
    w.add_light(rt.DiscLight(rt.point(0, 8, 4), rt.vector(0, -1, 0), 0.5, rt.color(1000, 1000, 1000)))

    def matt(r, g, b):
        return rt.Phong.default().with_rgb(r, g, b).with_emissive(0.0).with_diffuse(0.5).with_specular(0.0).with_shininess(100.0)
    white, red, green = matt(1, 1, 1), matt(1, 0.3, 0.3), matt(0.3, 1, 0.3)
    w.add_item(rt.plane().with_material(white))
    w.add_item(rt.plane().with_material(white).with_transform(rt.translation(0, 10, 0)))
    w.add_item(rt.plane().with_material(white).with_transform(rt.translation(0, 0, 5) * rt.rotation_x(PI / 2.0)))
    w.add_item(rt.plane().with_material(red).with_transform(rt.translation(-5, 0, 0) * rt.rotation_z(PI / 2.0)))
    w.add_item(rt.plane().with_material(green).with_transform(rt.translation(5, 0, 0) * rt.rotation_z(PI / 2.0)))
    w.add_item(rt.plane().with_material(white).with_transform(rt.translation(0, 0, -5) * rt.rotation_x(PI / 2.0)))
    w.add_item(rt.cube().with_material(white).with_transform(rt.translation(-2, 2, 2) * rt.rotation_y(0.8) * rt.scaling(1, 2, 1)))
    w.add_item(rt.sphere().with_material(
        rt.Phong.default().with_rgb(1.0, 1.0, 1.0).with_diffuse(0.0).with_specular(0.0).with_shininess(250.0).with_transparency(0.9)
        .with_refractive_index(1.5).with_reflective(0.0)).with_transform(rt.translation(2, 1.5, 2) * rt.scaling(1.5, 1.5, 1.5)))
    w.enable_direct_illumination(direct)
    w.finalize_scene()
    return w, _camera(h, v, (0, 5, -4.9), (0, 5, 0), se=1e-2, min_samples=100, deterministic=deterministic, fov=PI / 2.0)


def soft_shadow(h=900, v=450, deterministic=False):
    """rust/examples/soft_shadow.rs:12-59 — BASELINE config 4 (area light)."""
    w = rt.World.empty()
    w.add_light(rt.SphereLight(rt.point(-5, 2, -5), 2.0, rt.color(1.5, 1.5, 1.5)))
    w.add_item(rt.sphere().with_transform(rt.translation(-5, 2, -5) * rt.scaling(2, 2, 2)).with_material(
        rt.Phong.default().with_rgb(1.0, 1.0, 1.0).with_emissive(1.0).with_diffuse(0.0).with_specular(0.0).with_reflective(0.0))
        .with_cast_shadow(False))
    floor_material = rt.Phong.default().with_color(rt.color(1.0, 1.0, 1.0)).with_diffuse(0.8).with_specular(0.0)
    w.add_item(rt.plane().with_material(floor_material))
    w.add_item(rt.cube().with_transform(rt.translation(0.0, 1, 0.0)).with_material(
        rt.Phong.default().with_rgb(1.0, 1.0, 1.0).with_diffuse(0.8).with_specular(0.0).with_reflective(0.0)))
    return w, _camera(h, v, (9, 15, -19), (1.5, 1, 1), se=1e-2, min_samples=10, deterministic=deterministic)


def depth_of_field(h=900, v=450, deterministic=False):
    """rust/examples/depth-of-field.rs:13-85 — BASELINE config 4 (thin lens)."""
    w = rt.World.empty()
    w.add_light(rt.PointLight(rt.point(-10, 10, -10), rt.color(0.5, 0.5, 0.5)))
    w.add_light(rt.PointLight(rt.point(-10, 10, 10), rt.color(0.5, 0.5, 0.5)))
    floor_material = rt.Phong.default().with_pattern(
        rt.checkers_pattern(rt.color(0.1, 0.1, 0.1), rt.color(0.9, 0.9, 0.9))).with_specular(0.0)
    w.add_item(rt.plane().with_material(floor_material))
    w.add_item(_sky())
    ball = rt.Phong.default().with_rgb(0.1, 1.0, 0.5).with_diffuse(0.7).with_specular(0.3)
    w.add_item(rt.sphere().with_transform(rt.translation(-1.5, 1, 0.5)).with_material(ball))
    w.add_item(rt.sphere().with_transform(rt.translation(0.5, 1, 10.5)).with_material(ball))
    w.add_item(rt.sphere().with_transform(rt.translation(1.5, 1, -3.5)).with_material(
        rt.Phong.default().with_rgb(0.5, 0.1, 1.0).with_diffuse(0.7).with_specular(0.3)))
    w.add_item(rt.sphere().with_transform(rt.translation(2.0, 1, 4.5)).with_material(
        rt.Phong.default().with_rgb(0.0, 0.0, 0.0).with_diffuse(0.0).with_specular(1.0).with_reflective(1.0)))
    cam = _camera(h, v, (0, 1.5, -5), (0, 1, 0), se=5e-3, deterministic=deterministic)
    cam.set_focal_distance(10.0)
    cam.set_aperture_size(0.25)
    return w, cam


# --- chapter-16 dice (rust/examples/chapter-16.rs:17-107) ---
def _dice_point(i, j, mat):
    return rt.sphere().with_material(mat).with_transform(rt.translation(0.5 * i, 1, 0.5 * j) * rt.scaling(0.2, 0.1, 0.2))


def _dice_side(points, mat):
    if len(points) == 1:
        return rt.SceneItem(_dice_point(points[0][0], points[0][1], mat))
    g = rt.group()
    for (i, j) in points:
        g = g.with_child(_dice_point(i, j, mat))
    return rt.SceneItem(g)


_SIDES = {
    1: [(0.0, 0.0)],
    2: [(-0.8, -0.8), (0.8, 0.8)],
    3: [(0.0, 0.0), (-1.0, -1.0), (1.0, 1.0)],
    4: [(-0.8, -0.8), (-0.8, 0.8), (0.8, -0.8), (0.8, 0.8)],
    5: [(0.0, 0.0), (-1.0, -1.0), (1.0, -1.0), (-1.0, 1.0), (1.0, 1.0)],
    6: [(-1.0, -1.0), (-1.0, 0.0), (-1.0, 1.0), (1.0, -1.0), (1.0, 0.0), (1.0, 1.0)],
}


def dice(mat1, mat2):
    body = rt.csg_intersection(rt.cube().with_material(mat1), rt.sphere().with_material(mat1).with_transform(rt.scaling(1.5, 1.5, 1.5)))
    body = rt.csg_difference(body, _dice_side(_SIDES[1], mat2))
    body = rt.csg_difference(body, _dice_side(_SIDES[6], mat2).with_transform(rt.scaling(1, -1, 1)))
    body = rt.csg_difference(body, _dice_side(_SIDES[2], mat2).with_transform(rt.rotation_x(PI / 2.0)))
    body = rt.csg_difference(body, _dice_side(_SIDES[5], mat2).with_transform(rt.rotation_x(-PI / 2.0)))
    body = rt.csg_difference(body, _dice_side(_SIDES[3], mat2).with_transform(rt.rotation_z(PI / 2.0)))
    body = rt.csg_difference(body, _dice_side(_SIDES[4], mat2).with_transform(rt.rotation_z(-PI / 2.0)))
    return rt.SceneItem(body)


def _philox_uniform(seed, index, k):
    """Same stream as the renderers (oracle/rt_oracle.c orc_uniform): scene randomness of chapter-16 is keyed
    by the die index so that the scene is reproducible (the reference uses thread_rng, chapter-16.rs:145-157)."""
    M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
    c = [index & 0xFFFFFFFF, 0xD1CE, k >> 2, 0x52544D49]
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k0) & 0xFFFFFFFF, p1 & 0xFFFFFFFF, ((p0 >> 32) ^ c[3] ^ k1) & 0xFFFFFFFF, p0 & 0xFFFFFFFF]
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return (c[k & 3] >> 8) / 16777216.0


def chapter16(h=900, v=450, deterministic=False, seed=0x5EED, nx=8, ny=4, max_depth=10):
    """rust/examples/chapter-16.rs:109-176 — BASELINE config 5: CSG dice (17 x 9) + lens, BVH leaf 2."""
    w = rt.World.empty()
    w.set_max_reflection_depth(max_depth)
    w.add_light(rt.AmbientLight(rt.color(0.5, 0.5, 0.5)))
    w.add_light(rt.PointLight(rt.point(-9, 8, -7), rt.color(1, 1, 1)))
    floor_material = rt.Phong.default().with_pattern(
        rt.checkers_pattern(rt.color(0.75, 0.75, 0.75), rt.color(0.9, 0.9, 0.9)).with_transform(rt.scaling(0.1, 0.1, 0.1))
    ).with_diffuse(0.5).with_specular(0.0)
    w.add_item(rt.plane().with_material(floor_material).with_transform(rt.translation(0, 0, 2) * rt.rotation_x(PI / 2.0)))
    glass = rt.Phong.new(rt.color(0, 0, 0), 0.0, 0.0, 0.9, 500.0, 1.0, 1.0, 1.5)
    a = rt.sphere().with_material(glass).with_transform(rt.translation(0, 0, 0.8))
    b = rt.sphere().with_material(glass).with_transform(rt.translation(0, 0, -0.8))
    w.add_item(rt.csg_intersection(a, b).with_transform(rt.translation(0.0, 0, 0)).with_cast_shadow(False))
    dices = rt.group()
    idx = 0
    for i in range(-nx, nx + 1):
        for j in range(-ny, ny + 1):
            u = [_philox_uniform(seed, idx, k) for k in range(8)]
            idx += 1
            hue = u[0] * 360.0
            mat1 = rt.Phong.default().with_hsv(hue, 0.8, 1.0).with_diffuse(1.0)
            mat2 = rt.Phong.default().with_hsv(hue + 180.0, 0.8, 1.0).with_diffuse(1.0)
            size = 0.05 + u[1] * 0.05
            pos_x = i * 0.4 + (-0.1 + u[2] * 0.2)
            pos_y = j * 0.4 + (-0.1 + u[3] * 0.2)
            phi = u[4] * 2.0 * PI
            rotax = rt.vector(u[5], u[6], u[7]).normalized()
            dices.add_child(dice(mat1, mat2).with_transform(
                rt.translation(pos_x, pos_y, 1.8) * rt.Matrix.identity().rotate(phi, rotax) * rt.scaling(size, size, size)))
    w.add_item(rt.build_bounding_tree(dices, 2))
    w.finalize_scene()
    return w, _camera(h, v, (0, 0, -3), (0, 0, 0), se=1e-2, min_samples=10, deterministic=deterministic)


# ------------------------------------------------------------------------------------------------
# Scheme twin scenes (scheme/chapterNN.scm): 320x160, 1 centred ray per pixel, depth 5, white point light.
# Scheme `ambient a` == Rust `emissive a` under a white light.
# (material color ambient diffuse specular shininess [reflective transparency refractive-index])
# ------------------------------------------------------------------------------------------------
def _smat(col, ambient, diffuse, specular, shininess, reflective=0.0, transparency=0.0, ior=1.0, pattern=None):
    m = rt.Phong.new(rt.color(*col) if col is not None else rt.color(0, 0, 0), ambient, diffuse, specular, shininess, reflective,
                     transparency, ior)
    return m.with_pattern(pattern) if pattern is not None else m


def _scheme_world(items, light=(-10, 10, -10)):
    w = rt.World.empty()
    w.set_max_reflection_depth(5)  # MAX-BOUNCE-DEPTH, scheme/raytrace/world.sld:40
    w.add_light(rt.PointLight(rt.point(*light), rt.color(1, 1, 1)))
    for it in items:
        w.add_item(it)
    return w


def _scheme_camera(frm=(0, 1.5, -5), to=(0, 1, 0)):
    return _camera(320, 160, frm, to, deterministic=True)


def _three_balls(kind, mats):
    mk = {"sphere": rt.sphere, "cylinder": rt.cylinder}[kind]
    return [
        mk().with_transform(rt.translation(-0.5, 1, 0.5)).with_material(mats[0]),
        mk().with_transform(rt.translation(1.5, 0.5, -0.5) * rt.scaling(0.5, 0.5, 0.5)).with_material(mats[1]),
        mk().with_transform(rt.translation(-1.5, 0.33, -0.75) * rt.scaling(0.33, 0.33, 0.33)).with_material(mats[2]),
    ]


def scheme_chapter08(shadows=True):
    """scheme/chapter08.scm (== chapter07.scm; chapter07.ppm predates shadows -> shadows=False)."""
    fm = _smat((1, 0.9, 0.9), 0.1, 0.9, 0.0, 100.0)
    walls = [
        rt.sphere().with_transform(rt.scaling(10, 0.01, 10)).with_material(fm),
        rt.sphere().with_transform(rt.translation(0, 0, 5) * rt.rotation_y(-PI / 4) * rt.rotation_x(-PI / 2) * rt.scaling(10, 0.01, 10)).with_material(fm),
        rt.sphere().with_transform(rt.translation(0, 0, 5) * rt.rotation_y(PI / 4) * rt.rotation_x(PI / 2) * rt.scaling(10, 0.01, 10)).with_material(fm),
    ]
    balls = _three_balls("sphere", [_smat((0.1, 1, 0.5), 0.1, 0.7, 0.3, 200.0), _smat((0.5, 1, 0.1), 0.1, 0.7, 0.3, 200.0),
                                    _smat((1, 0.8, 0.1), 0.1, 0.7, 0.3, 200.0)])
    items = walls + balls
    if not shadows:
        items = [s.with_cast_shadow(False) for s in items]
    return _scheme_world(items), _scheme_camera()


def scheme_chapter09():
    """scheme/chapter09.scm."""
    floor = rt.plane().with_material(_smat((1, 0.9, 0.9), 0.1, 0.9, 0.0, 100.0))
    balls = _three_balls("sphere", [_smat((0.1, 1, 0.5), 0.1, 0.7, 0.3, 200.0), _smat((0.5, 1, 0.1), 0.1, 0.7, 0.3, 200.0),
                                    _smat((1, 0.8, 0.1), 0.1, 0.7, 0.3, 200.0)])
    return _scheme_world([floor] + balls), _scheme_camera()


def scheme_chapter10():
    """scheme/chapter10.scm:16-37."""
    c = rt.color
    ground = rt.plane().with_material(_smat(None, 0.1, 0.9, 0.0, 100.0, pattern=rt.stripe_pattern(c(0.5, 0.5, 0.5), c(1, 0.9, 0.9))))
    balls = _three_balls("sphere", [
        _smat(None, 0.1, 0.7, 0.3, 200.0, pattern=rt.gradient_pattern(c(0.9, 0, 0.5), c(0.1, 1, 0.5))),
        _smat(None, 0.1, 0.7, 0.3, 200.0, pattern=rt.stripe_pattern(c(0.5, 0.5, 0.5), c(0.5, 1, 0.1))),
        _smat(None, 0.1, 0.7, 0.3, 200.0, pattern=rt.stripe_pattern(c(0.5, 0.5, 0.5), c(1, 0.8, 0.1))),
    ])
    return _scheme_world([ground] + balls), _scheme_camera()


def scheme_chapter11a():
    """scheme/chapter11a.scm:16-59 (golden only with the Scheme n1/n2 rule)."""
    c = rt.color
    ground = rt.plane().with_material(_smat(None, 1.0, 0.0, 0.0, 100.0, pattern=rt.checkers_pattern(c(0.8, 0.2, 0.2), c(0.2, 0.2, 0.8))))
    ceiling = rt.plane().with_material(_smat((0.8, 0.8, 1), 1.0, 0.0, 0.0, 100.0)).with_transform(rt.translation(0, 11, 0))
    left = rt.sphere().with_transform(rt.translation(-1.5, 1, 0.5)).with_material(_smat((0, 0, 0), 0.0, 0.0, 1.0, 200.0, 1.0, 0.0, 1.0))
    right = rt.sphere().with_transform(rt.translation(1.5, 1, 0.5)).with_material(_smat((0, 0, 0), 0.0, 0.0, 1.0, 200.0, 0.0, 1.0, 1.5))
    inner = rt.sphere().with_transform(rt.translation(1.5, 1, 0.5) * rt.scaling(0.5, 0.5, 0.5)).with_material(
        _smat((0, 0, 0), 0.0, 0.0, 0.0, 200.0, 0.0, 1.0, 0.8))
    return _scheme_world([ground, ceiling, right, inner, left]), _scheme_camera()


def scheme_chapter11b():
    """scheme/chapter11b.scm:16-49."""
    c = rt.color
    ground = rt.plane().with_material(_smat(None, 1.0, 0.0, 0.0, 100.0, pattern=rt.checkers_pattern(c(0.1, 0.4, 0.9), c(0.1, 0.2, 0.5)))
                                      ).with_transform(rt.translation(0, -3, 0))
    water = rt.plane().with_material(_smat((0, 0, 0), 1.0, 0.0, 0.0, 100.0, 0.0, 1.0, 1.33))
    foliage = rt.sphere().with_material(_smat(None, 1.0, 0.0, 0.0, 100.0, pattern=rt.checkers_pattern(c(0.2, 0.6, 0.2), c(0.6, 0.4, 0.1))
                                              .with_transform(rt.scaling(0.1, 0.1, 0.1)))).with_transform(rt.scaling(100, 100, 100))
    sky = rt.plane().with_material(_smat((0.8, 0.8, 1), 1.0, 0.0, 0.0, 100.0)).with_transform(rt.translation(0, 11, 0))
    return _scheme_world([ground, water, sky, foliage]), _scheme_camera((0, 0.5, 0), (0, 0.5, 1))


def scheme_chapter13():
    """scheme/chapter13.scm:15-58 (cylinders + a transparent capped cone; Scheme n1/n2 rule)."""
    floor = rt.plane().with_material(_smat((1, 0.9, 0.9), 0.1, 0.9, 0.0, 100.0))
    middle = rt.cylinder().with_transform(rt.translation(-0.5, 1, 0.5)).with_material(_smat((0.1, 1, 0.5), 0.1, 0.7, 0.3, 200.0))
    right = rt.cylinder(-math.inf, 0.1, True).with_transform(rt.translation(1.5, 0.0, -0.5) * rt.scaling(0.5, 1.0, 0.5)).with_material(
        _smat((0.5, 1, 0.1), 0.1, 0.7, 0.3, 200.0))
    left = rt.cylinder(-math.inf, 1.0, False).with_transform(rt.translation(-1.5, 0.33, -0.75) * rt.scaling(0.33, 0.33, 0.33)).with_material(
        _smat((1, 0.8, 0.1), 0.1, 0.7, 0.3, 200.0))
    top = rt.cone(-1.5, -1.0, True).with_transform(rt.translation(0, 0.5, -0.75) * rt.scaling(0.2, 0.2, 0.2)).with_material(
        _smat((0.9, 0.8, 0.7), 0.0, 0.5, 1.0, 200.0, 0.0, 0.5, 1.5))
    return _scheme_world([floor, middle, right, left, top]), _scheme_camera()


def scheme_chapter16():
    """scheme/chapter16.scm:16-37 (CSG difference cube - sphere)."""
    floor = rt.plane().with_material(_smat((1, 0.9, 0.9), 0.3, 0.9, 0.0, 100.0))
    a = rt.cube().with_material(_smat((0.2, 0.5, 1), 0.3, 0.9, 0.2, 50.0))
    b = rt.sphere().with_material(_smat((1, 0.5, 0.2), 0.3, 0.9, 0.2, 50.0)).with_transform(rt.translation(0.5, 0.5, -0.5))
    c = rt.csg_difference(a, b).with_transform(rt.translation(0, 1, 0))
    w = _scheme_world([floor, c], light=(-8, 10, -9))
    w.finalize_scene()
    return w, _scheme_camera((-1.5, 4, -5), (0, 1, 0))


SCHEME_GOLDENS = {
    # name: (builder, oracle uses the Scheme n1/n2 rule?)
    "chapter07": (lambda: scheme_chapter08(shadows=False), False),
    "chapter08": (scheme_chapter08, False),
    "chapter09": (scheme_chapter09, False),
    "chapter10": (scheme_chapter10, False),
    "chapter11a": (scheme_chapter11a, True),
    "chapter11b": (scheme_chapter11b, False),
    "chapter13": (scheme_chapter13, True),
    "chapter16": (scheme_chapter16, False),
}

def photon_box_with_map(h=600, v=600, deterministic=False, n_photons=2_000_000, n_nearest=100, max_search_radius=0.1):
    """examples/photon-map.rs' third picture (lines 106-118: direct light plus diffuse and caustic photons, 100 nearest within
    0.1) with n_photons instead of its 75 M; the map is traced on the device (World::compute_photon_map -> rtmi_trace_photons)"""
    w, cam = photon_box(h, v, deterministic=deterministic)
    w.enable_direct_photon_map(False); w.enable_diffuse_photon_map(True); w.enable_caustic_photon_map(True)
    w.compute_photon_map(n_photons, n_nearest, max_search_radius)
    return w, cam


def _icosahedron():
    t = (1.0 + math.sqrt(5.0)) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    n = math.sqrt(1.0 + t * t)
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    return [rt.point(x / n, y / n, z / n) for (x, y, z) in v], [list(t3) for t3 in f]


def meshes(h=300, v=150, deterministic=False):
    """Not a reference example (none builds a TriangleMesh): the parity case of the flattened TriangleMesh (triangle.rs:164-231,
    include/rtmi.hpp TriangleMesh) - an icosahedron mesh as a top-level Group, a second one, mirror
    finished, under build_bounding_tree, an octahedron built face by face with add_triangle, and a mesh as the right side of
    a CSG difference.  All opaque: the case in which a flattened mesh and the reference's one-Shape mesh render alike."""
    w = rt.World.empty()
    w.set_max_reflection_depth(4)
    w.add_light(rt.AmbientLight(rt.color(0.08, 0.08, 0.08)))
    w.add_light(rt.PointLight(rt.point(-6, 9, -7), rt.color(0.8, 0.8, 0.75)))
    w.add_light(rt.PointLight(rt.point(5, 4, -6), rt.color(0.25, 0.25, 0.3)))
    floor = rt.Phong.default().with_pattern(rt.checkers_pattern(rt.color(0.3, 0.3, 0.3), rt.color(0.75, 0.75, 0.7)).with_transform(
        rt.translation(0.013, 0.0, 0.017))).with_diffuse(0.8).with_specular(0.1).with_reflective(0.15)
    w.add_item(rt.plane().with_material(floor))
    verts, faces = _icosahedron()
    ico = rt.TriangleMesh.from_mesh(verts, faces)
    red = rt.Phong.default().with_rgb(0.85, 0.25, 0.2).with_diffuse(0.7).with_specular(0.4).with_shininess(80.0)
    w.add_item(ico.into_group(red).with_transform(rt.translation(-2.1, 1.05, 0.4) * rt.rotation_y(0.37) * rt.rotation_x(0.21)))
    mirror = rt.Phong.default().with_rgb(0.1, 0.12, 0.15).with_diffuse(0.3).with_specular(0.8).with_shininess(200.0).with_reflective(0.6)
    w.add_item(rt.build_bounding_tree(ico.into_group(mirror).with_transform(
        rt.translation(0.45, 1.3, 1.6) * rt.rotation_z(0.5) * rt.rotation_y(1.1) * rt.scaling(1.3, 1.3, 1.3)), 4))
    octa = rt.TriangleMesh.new()
    top, bot = rt.point(0, 1, 0), rt.point(0, -1, 0)
    ring = [rt.point(1, 0, 0), rt.point(0, 0, 1), rt.point(-1, 0, 0), rt.point(0, 0, -1)]
    for i in range(4):
        octa.add_triangle(rt.triangle(top, ring[i], ring[(i + 1) % 4]))
        octa.add_triangle(rt.triangle(bot, ring[(i + 1) % 4], ring[i]))
    green = rt.Phong.default().with_rgb(0.2, 0.7, 0.3).with_diffuse(0.75).with_specular(0.2)
    w.add_item(octa.into_group(green).with_transform(rt.translation(2.6, 0.8, -0.3) * rt.rotation_y(0.6) * rt.scaling(0.8, 0.8, 0.8)))
    blue = rt.Phong.default().with_rgb(0.25, 0.35, 0.85).with_diffuse(0.7).with_specular(0.3)
    carved = rt.csg_difference(rt.cube().with_material(blue),
                               octa.into_group(green).with_transform(rt.translation(0.35, 0.4, -0.6) * rt.scaling(1.1, 1.1, 1.1)))
    w.add_item(carved.with_transform(rt.translation(-0.3, 0.55, -1.9) * rt.rotation_y(0.45) * rt.scaling(0.55, 0.55, 0.55)))
    w.finalize_scene()
    return w, _camera(h, v, (0.4, 2.6, -7.5), (0.1, 0.9, 0), se=1e-2, min_samples=5, deterministic=deterministic)


EXTRA_SCENES = {"meshes": meshes, "photons": photon_room_with_map, "yard": yard, "photonbox": photon_box_with_map, "crowd": crowd}  # (bench.py --scene takes these too; not BASELINE configurations)

BASELINE_SCENES = {
    "ch07": chapter07,
    "ch11a": chapter11a,
    "ch11b": chapter11b,
    "teapot": chapter15,
    "soft": soft_shadow,
    "dof": depth_of_field,
    "dice": chapter16,
}
