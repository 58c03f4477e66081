"""Shared builders for the known-answer tests transcribed from the reference's `#[cfg(test)]` blocks.
Values and scenario names are the reference's (file:line cited at each test); tolerance = its assert_almost_eq
(1e-5 absolute, rust/src/approx_eq.rs:13,21-29)."""
import math

import numpy as np

import raytracing_amd as rt

EPS = 1e-5
S2 = math.sqrt(2.0)
FRAC_1_SQRT_2 = 1.0 / S2


def near(a, b, tol=EPS):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a, b)
    assert np.all(np.abs(a - b) < tol), "expected %s, got %s" % (b, a)


def default_material1():  # world.rs:43-49
    return rt.Phong.default().with_color(rt.color(0.8, 1.0, 0.6)).with_emissive(0.1).with_diffuse(0.7).with_specular(0.2)


def default_material2():  # world.rs:51-53
    return rt.Phong.default().with_emissive(0.1)


def default_world(m1=None, m2=None, light=None, extra=(), cast_shadow=True):
    """World::default (world.rs:25-41) with optional replacements, plus extra items appended (objects[2], ...)."""
    w = rt.World.empty()
    w.add_light(light if light is not None else rt.PointLight(rt.point(-10, 10, -10), rt.color(1, 1, 1)))
    s1 = rt.sphere().with_material(m1 if m1 is not None else default_material1())
    s2 = rt.sphere().with_material(m2 if m2 is not None else default_material2()).with_transform(rt.scaling(0.5, 0.5, 0.5))
    if not cast_shadow:
        s1 = s1.with_cast_shadow(False)
        s2 = s2.with_cast_shadow(False)
    w.add_item(s1)
    w.add_item(s2)
    for it in extra:
        w.add_item(it)
    return w


def single(shape, light=None):
    """A world holding one shape (leaf 0), for Shape-level tests."""
    w = rt.World.empty()
    if light is not None:
        w.add_light(light)
    w.add_item(shape)
    return w


def norm(v):
    v = np.asarray(v, dtype=np.float64)
    return v / np.linalg.norm(v)


# ---- scenario tables shared by the oracle and the GPU tests -------------------------------------------

# world.rs:408-447 is_shadowed: (point, expected)
SHADOW_CASES = [((0, 10, 0), False), ((10, -10, 10), True), ((-20, 20, -20), False), ((-2, 2, -2), False)]

# ray.rs:421-462: nested glass spheres A(ior 1.5, scale 2), B(2.0, z-0.25), C(2.5, z+0.25); ray (0,0,-4)->(0,0,1)
N1N2_TS = [2.0, 2.75, 3.25, 4.75, 5.25, 6.0]
N1N2_LEAVES = [0, 1, 2, 1, 2, 0]
N1N2_EXPECT = [(1.0, 1.5), (1.5, 2.0), (2.0, 2.5), (2.5, 2.5), (2.5, 1.5), (1.5, 1.0)]


def nested_glass_world():
    glass = rt.glass_sphere().material()
    w = rt.World.empty()
    w.add_item(rt.glass_sphere().with_transform(rt.scaling(2, 2, 2)).with_material(glass.with_refractive_index(1.5)))
    w.add_item(rt.glass_sphere().with_transform(rt.translation(0, 0, -0.25)).with_material(glass.with_refractive_index(2.0)))
    w.add_item(rt.glass_sphere().with_transform(rt.translation(0, 0, 0.25)).with_material(glass.with_refractive_index(2.5)))
    return w


def reflective_plane():  # world.rs:490-493
    return rt.plane().with_material(rt.Phong.default().with_reflective(0.5)).with_transform(rt.translation(0, -1, 0))


def refractive_floor_world(reflective):  # world.rs:680-729
    floor = rt.plane().with_transform(rt.translation(0, -1, 0)).with_material(
        (rt.Phong.default().with_reflective(0.5) if reflective else rt.Phong.default()).with_transparency(0.5).with_refractive_index(1.5))
    ball = rt.sphere().with_transform(rt.translation(0, -3.5, -0.5)).with_material(rt.Phong.default().with_rgb(1.0, 0.0, 0.0).with_emissive(0.5))
    return default_world(extra=[floor, ball])


def refracted_color_world():  # world.rs:642-657
    m1 = default_material1().with_emissive(1.0).with_pattern(rt.test_pattern())
    m2 = default_material2().with_transparency(1.0).with_refractive_index(1.5)
    return default_world(m1=m1, m2=m2)


def mirror_box_world():  # world.rs:549-574
    w = rt.World.empty()
    m = rt.Phong.default().with_color(rt.color(0, 0, 0)).with_emissive(1.0).with_diffuse(0.0).with_specular(0.0).with_reflective(1.0)
    w.add_item(rt.plane().with_transform(rt.translation(0, -1, 0)).with_material(m))
    w.add_item(rt.plane().with_transform(rt.translation(0, 1, 0)).with_material(m))
    w.add_light(rt.PointLight(rt.point(0, 0, 0), rt.color(1, 1, 1)))
    return w


# cube.rs:96-104 (origin, direction, t1, t2)
CUBE_HITS = [((5, 0.5, 0), (-1, 0, 0), 4, 6), ((-5, 0.5, 0), (1, 0, 0), 4, 6), ((0.5, 5, 0), (0, -1, 0), 4, 6), ((0.5, -5, 0), (0, 1, 0), 4, 6),
             ((0.5, 0, 5), (0, 0, -1), 4, 6), ((0.5, 0, -5), (0, 0, 1), 4, 6), ((0, 0.5, 0), (0, 0, 1), -1, 1)]
# cube.rs:117-124
CUBE_MISSES = [((-2, 0, 0), (0.2673, 0.5345, 0.8018)), ((0, -2, 0), (0.8018, 0.2673, 0.5345)), ((0, 0, -2), (0.5345, 0.8018, 0.2673)),
               ((2, 0, 2), (0, 0, -1)), ((0, 2, 2), (0, -1, 0)), ((2, 2, 0), (-1, 0, 0))]
# cube.rs:137-150
CUBE_NORMALS = [((1.0, 0.5, -0.8), (1, 0, 0)), ((-1.0, -0.2, 0.9), (-1, 0, 0)), ((-0.4, 1.0, -0.1), (0, 1, 0)), ((0.3, -1.0, -0.7), (0, -1, 0)),
                ((-0.6, 0.3, 1.0), (0, 0, 1)), ((0.4, 0.4, -1.0), (0, 0, -1)), ((1, 1, 1), (1, 0, 0)), ((-1, -1, -1), (-1, 0, 0))]
# sphere.rs:93-149 (origin, ts or None)
SPHERE_CASES = [((0, 0, -5), (4.0, 6.0)), ((0, 1, -5), (5.0, 5.0)), ((0, 2, -5), None), ((0, 0, 0), (-1.0, 1.0)), ((0, 0, 5), (-6.0, -4.0))]
# cylinder.rs:148-152 / 164-168 / 182-187 / 198-205 / 217-223 / 235-242
CYL_MISSES = [((1, 0, 0), (0, 1, 0)), ((0, 0, 0), (0, 1, 0)), ((0, 0, -5), (1, 1, 1))]
CYL_HITS = [((1, 0, -5), (0, 0, 1), 5, 5), ((0, 0, -5), (0, 0, 1), 4, 6), ((0.5, 0, -5), (0.1, 1, 1), 6.80798, 7.08872)]
CYL_NORMALS = [((1, 0, 0), (1, 0, 0)), ((0, 5, -1), (0, 0, -1)), ((0, 2, -1), (0, 0, -1)), ((-1, 1, 0), (-1, 0, 0))]
CYL_CONSTRAINED = [((0, 1.5, 0), (0.1, 1, 0), 0), ((0, 3, -5), (0, 0, 1), 0), ((0, 0, -5), (0, 0, 1), 0), ((0, 2, -5), (0, 0, 1), 0),
                   ((0, 1, -5), (0, 0, 1), 0), ((0, 1.5, -2), (0, 0, 1), 2)]
CYL_CLOSED = [((0, 3, 0), (0, -1, 0), 2), ((0, 3, -2), (0, -1, 2), 2), ((0, 4, -2), (0, -1, 1), 2), ((0, 0, -2), (0, 1, 2), 2),
              ((0, -1, -2), (0, 1, 1), 2)]
CYL_CAP_NORMALS = [((0, 1, 0), (0, -1, 0)), ((0.5, 1, 0), (0, -1, 0)), ((0, 1, 0.5), (0, -1, 0)), ((0, 2, 0), (0, 1, 0)), ((0.5, 2, 0), (0, 1, 0)),
                   ((0, 2, 0.5), (0, 1, 0))]
# cone.rs:160-164 / 179-183 / 189-193 / 205-209
CONE_HITS = [((0, 0, -5), (0, 0, 1), 5, 5), ((0, 0, -5), (1, 1, 1), 8.66025, 8.66025), ((1, 1, -5), (-0.5, -1, 1), 4.55006, 49.44994)]
CONE_CAPS = [((0, 0, -5), (0, 1, 0), 0), ((0, 0, -0.25), (0, 1, 1), 2), ((0, 0, -0.25), (0, 1, 0), 4)]
CONE_NORMALS = [((0, 0, 0), (0, 0, 0)), ((1, 1, 1), (1, -S2, 1)), ((-1, -1, 0), (-1, 1, 0))]
# csg.rs:154-179 (op, lhit, inl, inr, expected)
CSG_RULES = [("u", 1, 1, 1, 0), ("u", 1, 1, 0, 1), ("u", 1, 0, 1, 0), ("u", 1, 0, 0, 1), ("u", 0, 1, 1, 0), ("u", 0, 1, 0, 0), ("u", 0, 0, 1, 1),
             ("u", 0, 0, 0, 1), ("i", 1, 1, 1, 1), ("i", 1, 1, 0, 0), ("i", 1, 0, 1, 1), ("i", 1, 0, 0, 0), ("i", 0, 1, 1, 1), ("i", 0, 1, 0, 1),
             ("i", 0, 0, 1, 0), ("i", 0, 0, 0, 0), ("d", 1, 1, 1, 0), ("d", 1, 1, 0, 1), ("d", 1, 0, 1, 0), ("d", 1, 0, 0, 1), ("d", 0, 1, 1, 1),
             ("d", 0, 1, 0, 1), ("d", 0, 0, 1, 0), ("d", 0, 0, 0, 0)]
CSG_KIND = {"u": 3, "i": 4, "d": 5}
CSG_MAKE = {"u": rt.csg_union, "i": rt.csg_intersection, "d": rt.csg_difference}
# csg.rs:188-192: which of the four sorted intersections survive
CSG_FILTER = [("u", 0, 3), ("i", 1, 2), ("d", 0, 1)]


def csg_filter_world(op, wrap):
    """Geometry that reproduces csg.rs:185-275's synthetic list: ray (0,0,-2)->(0,0,1) meets the left sphere at t = 1, 3 and
    the right cube (centred z = 1) at t = 2, 4.  wrap: 'plain' | 'group' | 'csg' (children wrapped like the three tests)."""
    s1 = rt.sphere()
    s2 = rt.cube().with_transform(rt.translation(0, 0, 1))
    far = rt.translation(0, 100, 0)
    if wrap == "group":
        a, b = rt.group().with_child(s1), rt.group().with_child(s2)
    elif wrap == "csg":  # csg_union(s, s') where s' is out of the ray's way
        a = rt.csg_union(s1, rt.sphere().with_transform(far))
        b = rt.csg_union(rt.sphere().with_transform(far), s2)
    else:
        a, b = s1, s2
    w = rt.World.empty()
    w.add_item(CSG_MAKE[op](a, b))
    return w
