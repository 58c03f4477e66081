"""The reference's known-answer tests driven through the device path (rtmi_trace_rays = World::color_at + the
IntersectionState of the first hit; rtmi_camera_rays = Camera::ray_for_pixel), in the f64 validation mode (tolerance 1e-5,
the reference's) and on the f32 product path (tolerance 2e-4: f32 rounding through normalisations and pow)."""
import math

import numpy as np
import pytest

import oracle_lib as orc
import raytracing_amd as rt
from raytracing_amd import abi
from kat_cases import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu

PREC = [pytest.param(1, 1e-5, id="f64"), pytest.param(0, 2e-4, id="f32")]


def trace(world, rays, precision, remaining=None):
    sc = abi.Scene(world)
    o = [r[0] for r in rays]
    d = [r[1] for r in rays]
    rgb, recs = abi.trace_rays(sc, o, d, remaining=remaining, opts=abi.make_opts(precision=precision))
    sc.close()
    return rgb, recs


@pytest.mark.parametrize("precision,tol", PREC)
def test_primitive_closest_hits(precision, tol):
    """sphere.rs / plane.rs / cube.rs / cylinder.rs / cone.rs / triangle.rs intersection tables: the first root with t >= 0."""
    def first(ts):
        pos = [t for t in ts if t >= 0]
        return pos[0] if pos else None

    cases = []  # (shape, origin, dir, expected first t or None)
    for o, ts in SPHERE_CASES:
        cases.append((rt.sphere(), o, (0, 0, 1), first(ts) if ts else None))
    cases += [(rt.plane(), (0, 10, 0), (0, 0, 1), None), (rt.plane(), (0, 0, 0), (0, 0, 1), None), (rt.plane(), (0, 1, 0), (0, -1, 0), 1.0),
              (rt.plane(), (0, -1, 0), (0, 1, 0), 1.0)]
    for o, d, t1, t2 in CUBE_HITS:
        cases.append((rt.cube(), o, d, first([t1, t2])))
    for o, d in CUBE_MISSES:
        cases.append((rt.cube(), o, d, None))
    for o, d in CYL_MISSES:
        cases.append((rt.cylinder(), o, norm(d), None))
    for o, d, t0, t1 in CYL_HITS:
        cases.append((rt.cylinder(), o, norm(d), t0))
    for o, d, t0, t1 in CONE_HITS:
        if precision == 0 and t0 == t1 and o == (0, 0, -5) and d == (1, 1, 1):
            continue  # exactly tangent with an inexact direction: the sign of the f32 discriminant is rounding noise
        cases.append((rt.cone(), o, norm(d), t0))
    cases.append((rt.cone(), (0, 0, -1), norm((0, 1, 1)), 0.35355))
    tri = rt.triangle(rt.point(0, 1, 0), rt.point(-1, 0, 0), rt.point(1, 0, 0))
    cases += [(tri, (0, -1, -2), (0, 1, 0), None), (tri, (1, 1, -2), (0, 0, 1), None), (tri, (-1, 1, -2), (0, 0, 1), None),
              (tri, (0, -1, -2), (0, 0, 1), None), (tri, (0, 0.5, -2), (0, 0, 1), 2.0)]
    for shape, o, d, want in cases:
        _, recs = trace(single(shape), [(o, d)], precision)
        if want is None:
            assert recs[0].hit == 0, (o, d)
        else:
            assert recs[0].hit == 1, (o, d)
            assert abs(recs[0].t - want) < max(tol, 1e-5) * max(1.0, abs(want)), (o, d, recs[0].t, want)


@pytest.mark.parametrize("precision,tol", PREC)
def test_counts_of_constrained_and_capped_quadrics(precision, tol):
    """cylinder.rs:195-230, cone.rs:186-200: hit / miss of truncated and capped shapes (count > 0 <=> a hit with t >= 0 here)."""
    for o, d, count in CYL_CONSTRAINED:
        _, recs = trace(single(rt.cylinder(1.0, 2.0, False)), [(o, norm(d))], precision)
        assert (recs[0].hit == 1) == (count > 0)
    for o, d, count in CYL_CLOSED:
        _, recs = trace(single(rt.cylinder(1.0, 2.0, True)), [(o, norm(d))], precision)
        assert (recs[0].hit == 1) == (count > 0)
    for o, d, count in CONE_CAPS:
        _, recs = trace(single(rt.cone(-0.5, 0.5, True)), [(o, norm(d))], precision)
        assert (recs[0].hit == 1) == (count > 0)


@pytest.mark.parametrize("precision,tol", PREC)
def test_normals(precision, tol):
    """Normals of the hit (cube.rs:132-151, cylinder.rs:179-193,232-248, triangle.rs:383-415, shapes/mod.rs:700-719)."""
    _, recs = trace(single(rt.cube()), [((5, 0.5, -0.8), (-1, 0, 0)), ((-0.4, 5, -0.1), (0, -1, 0)), ((0.4, 0.4, -5), (0, 0, 1))], precision)
    near(list(recs[0].normal), (1, 0, 0), tol); near(list(recs[1].normal), (0, 1, 0), tol); near(list(recs[2].normal), (0, 0, -1), tol)
    _, recs = trace(single(rt.cylinder(1.0, 2.0, True)), [((0.5, 5, 0), (0, -1, 0)), ((0, -5, 0.5), (0, 1, 0)), ((5, 1.5, 0), (-1, 0, 0))], precision)
    near(list(recs[0].normal), (0, 1, 0), tol); near(list(recs[1].normal), (0, -1, 0), tol); near(list(recs[2].normal), (1, 0, 0), tol)
    tri = rt.smooth_triangle(rt.point(0, 1, 0), rt.point(-1, 0, 0), rt.point(1, 0, 0), rt.vector(0, 1, 0), rt.vector(-1, 0, 0), rt.vector(1, 0, 0))
    _, recs = trace(single(tri), [((-0.2, 0.3, -2), (0, 0, 1))], precision)
    near(list(recs[0].normal), (-0.5547, 0.83205, 0), max(tol, 1e-5))
    s = rt.sphere().with_transform(rt.scaling(1, 0.5, 1) * rt.rotation_z(math.pi / 5.0))
    f = single(s).flatten()
    rec = orc.hit_record(f, (0, 5, -0.3), (0, -1, 0))
    _, recs = trace(single(s), [((0, 5, -0.3), (0, -1, 0))], precision)
    near(list(recs[0].normal), list(rec.normal), tol)


@pytest.mark.parametrize("precision,tol", PREC)
def test_world_color_at(precision, tol):  # world.rs:330-405
    rgb, recs = trace(default_world(), [((0, 0, -5), (0, 0, 1)), ((0, 0, -5), (0, 1, 0))], precision, remaining=[0, 0])
    near(rgb[0], (0.38066, 0.47583, 0.2855), max(tol, 1e-5))
    assert recs[0].hit == 1 and abs(recs[0].t - 4.0) < tol and not recs[0].inside
    near(list(recs[0].point), (0, 0, -1), tol); near(list(recs[0].normal), (0, 0, -1), tol)
    assert recs[1].hit == 0
    near(rgb[1], (0, 0, 0))
    w = default_world(light=rt.PointLight(rt.point(0, 0.25, 0), rt.color(1, 1, 1)))
    rgb, recs = trace(w, [((0, 0, 0), (0, 0, 1))], precision, remaining=[0])  # shade_inner_intersection
    near(rgb[0], (0.90498, 0.90498, 0.90498), max(tol, 1e-5))
    assert recs[0].inside and abs(recs[0].t - 0.5) < tol
    w = default_world(m1=default_material1().with_emissive(1.0), m2=default_material2().with_emissive(1.0))
    rgb, _ = trace(w, [((0, 0, 0.75), (0, 0, -1))], precision, remaining=[0])  # behind
    near(rgb[0], (1, 1, 1), tol)


@pytest.mark.parametrize("precision,tol", PREC)
def test_shadows(precision, tol):  # world.rs:407-463
    """is_shadowed at a surface point: probe planes through each test point facing the light give the same boolean."""
    w = rt.World.empty()
    w.add_light(rt.PointLight(rt.point(0, 0, -10), rt.color(1, 1, 1)))
    w.add_item(rt.sphere())
    w.add_item(rt.sphere().with_transform(rt.translation(0, 0, 10)))
    rgb, recs = trace(w, [((0, 0, 5), (0, 0, 1))], precision, remaining=[0])  # shadow5: shade in shadow = BLACK
    near(rgb[0], (0, 0, 0))
    assert recs[0].shadowed == 1.0 and recs[0].leaf == 1
    f = default_world().flatten()
    rays = [((0, 0, -5), (0, 0, 1)), ((0, 0, 5), (0, 0, -1)), ((5, 0.2, 0), (-1, 0, 0)), ((0.1, -5, 0.1), (0, 1, 0))]
    _, recs = trace(default_world(), rays, precision)
    for r, (o, d) in zip(recs, rays):
        assert r.shadowed == orc.hit_record(f, o, d).shadowed
    _, recs = trace(default_world(cast_shadow=False), rays, precision)  # shadow6
    assert all(r.shadowed == 0.0 for r in recs)


@pytest.mark.parametrize("precision,tol", PREC)
def test_reflection_and_refraction(precision, tol):  # world.rs:486-729
    o, d = (0, 0, -3), (0, -FRAC_1_SQRT_2, FRAC_1_SQRT_2)
    rgb, _ = trace(default_world(extra=[reflective_plane()]), [(o, d), (o, d)], precision, remaining=[1, 0])
    near(rgb[0], (0.77676, 0.82434, 0.72917), max(tol, 1e-5))  # shade_reflection
    near(rgb[1], np.array((0.77676, 0.82434, 0.72917)) - np.array((0.19033, 0.23792, 0.14274)), max(tol, 2e-5))  # max_reflect: no bounce
    rgb, _ = trace(mirror_box_world(), [((0, 0, 0), (0, 1, 0))], precision, remaining=[10])  # infinite_reflection terminates
    assert np.all(np.isfinite(rgb))
    rgb, _ = trace(refractive_floor_world(False), [(o, d)], precision, remaining=[5])
    near(rgb[0], (0.83642, 0.58642, 0.58642), max(tol, 1e-5))  # refracted_shading
    rgb, _ = trace(refractive_floor_world(True), [(o, d)], precision, remaining=[5])
    near(rgb[0], (0.83391, 0.59643, 0.59243), max(tol, 1e-5))  # schlick_shading
    # refracted_color (world.rs:640-676): the inner glass sphere seen from inside the patterned outer one; whole-pixel
    # value checked against the oracle (the reference asserts only the refracted term)
    w = refracted_color_world()
    want = orc.color_at(w.flatten(), (0, 0, 0.1), (0, 1, 0), 5)[1]
    rgb, _ = trace(w, [((0, 0, 0.1), (0, 1, 0))], precision, remaining=[5])
    near(rgb[0], want, max(tol, 1e-5))


@pytest.mark.parametrize("precision,tol", PREC)
def test_refractive_indices(precision, tol):  # ray.rs:420-462 through real intersections
    rays = [((0, 0, -4 + t - 0.01), (0, 0, 1)) for t in N1N2_TS]
    _, recs = trace(nested_glass_world(), rays, precision)
    for r, want, leaf in zip(recs, N1N2_EXPECT, N1N2_LEAVES):
        assert r.leaf == leaf
        near([r.n1, r.n2], want, 1e-6)


@pytest.mark.parametrize("precision,tol", PREC)
def test_csg(precision, tol):  # csg.rs:185-299
    for wrap in ("plain", "group", "csg"):
        for op, x0, x1 in CSG_FILTER:
            w = csg_filter_world(op, wrap)
            w.finalize_scene()
            want = [1.0, 2.0, 3.0, 4.0][x0]
            _, recs = trace(w, [((0, 0, -2), (0, 0, 1)), ((0, 0, -2 + want + 0.01), (0, 0, 1))], precision)
            assert abs(recs[0].t - want) < 10 * tol, (wrap, op)
            assert abs(recs[1].t - ([1.0, 2.0, 3.0, 4.0][x1] - want - 0.01)) < 10 * tol, (wrap, op)
    w = rt.World.empty()
    w.add_item(rt.csg_union(rt.sphere(), rt.cube()))
    _, recs = trace(w, [((0, 2, -5), (0, 0, 1))], precision)
    assert recs[0].hit == 0
    w = rt.World.empty()
    w.add_item(rt.csg_union(rt.sphere(), rt.sphere().with_transform(rt.translation(0, 0, 0.5))))
    _, recs = trace(w, [((0, 0, -5), (0, 0, 1)), ((0, 0, 0.7), (0, 0, 1))], precision)
    assert abs(recs[0].t - 4.0) < 10 * tol and recs[0].leaf == 0
    assert abs(recs[1].t - 0.8) < 10 * tol and recs[1].leaf == 1


@pytest.mark.parametrize("precision,tol", PREC)
def test_patterns(precision, tol):  # pattern.rs:109-244 via an emissive-only material looked at from above
    def look(pattern, pts, shape_tf=None):
        m = rt.Phong.new(rt.color(0, 0, 0), 1.0, 0.0, 0.0, 200.0, 0.0, 0.0, 1.0).with_pattern(pattern)
        pl = rt.plane().with_material(m)
        if shape_tf is not None:
            pl = pl.with_transform(shape_tf)
        rgb, _ = trace(single(pl), [((p[0], 1.0, p[2]), (0, -1, 0)) for p in pts], precision)
        return rgb

    WH, BL = rt.color(1, 1, 1), rt.color(0, 0, 0)
    rgb = look(rt.stripe_pattern(WH, BL), [(0, 0, 0), (0.9, 0, 0), (1.0, 0, 0), (-0.1, 0, 0), (-1.0, 0, 0), (-1.1, 0, 0), (0, 0, 2)])
    near(rgb[:, 0], [1, 1, 0, 0, 0, 1, 1])
    rgb = look(rt.gradient_pattern(WH, BL), [(0.25, 0, 0), (0.5, 0, 0), (0.75, 0, 0)])
    near(rgb[:, 0], [0.75, 0.5, 0.25], tol)
    rgb = look(rt.ring_pattern(WH, BL), [(0, 0, 0), (1, 0, 0), (0, 0, 1), (0.708, 0, 0.708)])
    near(rgb[:, 0], [1, 0, 0, 0])
    rgb = look(rt.checkers_pattern(WH, BL), [(0, 0, 0), (0.99, 0, 0), (1.01, 0, 0), (0, 0, 0.99), (0, 0, 1.01)])
    near(rgb[:, 0], [1, 1, 0, 1, 0])
    rgb = look(rt.test_pattern().with_transform(rt.translation(0.5, 1, 1.5)), [(2.5, 0, 3.5)], rt.scaling(2, 2, 2))
    near(rgb[0], (0.75, -1.0, 0.25), tol)  # objpat_transform at y = 0 (pattern.rs:127-134 uses a sphere at y = 3)


@pytest.mark.parametrize("precision,tol", PREC)
def test_camera_rays(precision, tol):  # camera.rs:349-375
    c = rt.Camera(201, 101, math.pi / 2.0)
    o, d = abi.camera_rays(c, [100, 0], [50, 0], opts=abi.make_opts(precision=precision))
    near(o[0], (0, 0, 0), tol); near(d[0], (0, 0, -1), tol)
    near(d[1], (0.66519, 0.33259, -0.66851), max(tol, 1e-5))
    c.set_transform(rt.rotation_y(math.pi / 4.0) * rt.translation(0, -2, 5))
    o, d = abi.camera_rays(c, [100], [50], opts=abi.make_opts(precision=precision))
    near(o[0], (0, 2, -5), tol); near(d[0], (FRAC_1_SQRT_2, 0, -FRAC_1_SQRT_2), tol)


@pytest.mark.parametrize("precision,tol", PREC)
def test_jittered_and_lens_rays_match_oracle(precision, tol):
    """Samples >= 1 (jitter) and the thin-lens branch (camera.rs:123-126,137-152) draw the same Philox numbers as the oracle."""
    from raytracing_amd import scenes
    _, cam = scenes.depth_of_field(64, 32)
    px = [0, 13, 63, 31]; py = [0, 7, 31, 16]; s = [0, 1, 2, 77]
    o, d = abi.camera_rays(cam, px, py, s, opts=abi.make_opts(precision=precision, seed=1234))
    for i in range(4):
        oo, dd = orc.camera_ray(cam, px[i], py[i], s[i], seed=1234)
        near(o[i], oo, tol); near(d[i], dd, tol)


@pytest.mark.parametrize("precision,tol", PREC)
def test_camera_render_pixel(precision, tol):  # camera.rs:377-389
    c = rt.Camera(11, 11, math.pi / 2.0).with_view_transform(rt.point(0, 0, -5), rt.point(0, 0, 0), rt.vector(0, 1, 0))
    c.set_min_samples(1)
    c.set_allowed_standard_error(1.0)
    sc = abi.Scene(rt.World.default())
    img, st = abi.render_host(sc, c, abi.make_opts(precision=precision))
    sc.close()
    near(img[5, 5], (0.38066, 0.47583, 0.2855), max(tol, 1e-5))
    assert st["samples"] == 121
    cv = c.render(rt.World.default(), precision=precision)  # the C++ front end's Camera::render (camera.rs:155-161)
    near(cv.get_pixel(5, 5).tuple(), (0.38066, 0.47583, 0.2855), max(tol, 1e-5))


def test_random_rays_against_oracle():
    """10,000 random rays into each BASELINE scene: hit leaf, t, normal, n1/n2 and World::color_at of the f64 device path
    equal the oracle's; the f32 path agrees on >= 99.5 % of the hit leaves.  The `shadowed` probe may differ in one
    direction only: a one-sided surface (plane, triangle) seen from the side away from the light shadows itself in the
    reference (the over_point lies behind it); the device treats the surface the ray left as behind it and reports
    "lit" - Phong::lighting returns the same colour either way (light_dot_normal < 0), which the colour check proves."""
    from raytracing_amd import scenes
    rng = np.random.default_rng(7)
    for name, kw in [("ch11a", {}), ("soft", {}), ("teapot", {"lite": True}), ("dice", {"nx": 1, "ny": 1})]:
        world, _ = scenes.BASELINE_SCENES[name](32, 16, **kw)
        f = world.flatten()
        n = 10000 if name != "dice" else 3000
        o = rng.uniform(-4, 4, (n, 3)) + np.array([0, 2.0, -3.0])
        tgt = rng.uniform(-2, 2, (n, 3)) + np.array([0, 0.5, 0.5])
        d = tgt - o
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        sc = abi.Scene(world)
        c64, r64 = abi.trace_rays(sc, o, d, opts=abi.make_opts(precision=1))
        _, r32 = abi.trace_rays(sc, o, d, opts=abi.make_opts(precision=0))
        sc.close()
        same32 = 0
        for i in range(n):
            ref = orc.hit_record(f, o[i], d[i])
            g = r64[i]
            assert g.hit == ref.hit, (name, i)
            if ref.hit:
                assert g.leaf == ref.leaf and abs(g.t - ref.t) < 1e-7 * max(1.0, abs(ref.t)), (name, i, g.leaf, ref.leaf, g.t, ref.t)
                near(list(g.normal), list(ref.normal), 1e-7)
                assert (g.n1, g.n2) == (ref.n1, ref.n2), (name, i)
                assert g.shadowed == ref.shadowed or (g.shadowed == 0.0 and ref.shadowed == 1.0), (name, i)
            if name != "soft":  # (the sphere light draws random numbers: compared by the image tests)
                near(c64[i], orc.color_at(f, o[i], d[i], world.max_reflection_depth)[1], 1e-9)
            # (coincident surfaces - soft_shadow's floor plane and the cube's bottom face - tie in t: either leaf is accepted in f32)
            same32 += int(r32[i].hit == ref.hit and (not ref.hit or r32[i].leaf == ref.leaf or abs(r32[i].t - ref.t) < 1e-5 * max(1.0, ref.t)))
        assert same32 >= 0.995 * n, (name, same32, n)
