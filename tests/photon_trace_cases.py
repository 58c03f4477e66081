"""Scenes and the comparison for the photon-tracing tests (rtmi_trace_photons against oracle/rt_oracle.c's
orc_trace_photons): shared by tests/test_gpu_photon_trace.py and tests/emul/run_emul_photons.py."""
import math

import numpy as np

from raytracing_amd import _front as rt
from raytracing_amd import scenes


def box():
    return scenes.photon_box(32, 32)[0]


def lights_room():
    """Every light that emits (lights.rs: Point, RealisticPoint, Beam, Sphere, Disc) plus an ambient one (power 0: never
    chosen), a patterned floor, a mirror, a glass ball with Schlick weights (reflective and transparent) and a mesh-free
    group: all four outcomes of Phong::photon_hit's roulette (materials.rs:447-461)."""
    w = rt.World.empty()
    w.add_light(rt.AmbientLight(rt.color(0.1, 0.1, 0.1)))
    w.add_light(rt.PointLight(rt.point(-3, 6, -4), rt.color(0.8, 0.8, 0.8)))
    w.add_light(rt.RealisticPointLight(rt.point(3, 5, -2), rt.color(0.5, 0.4, 0.3)))
    w.add_light(rt.SphereLight(rt.point(0, 7, 0), 0.5, rt.color(0.6, 0.6, 0.9)))
    w.add_light(rt.DiscLight(rt.point(2, 6, 2), rt.vector(-0.2, -1, -0.1).normalized(), 0.7, rt.color(0.9, 0.7, 0.5)))
    w.add_light(rt.Beam(rt.point(-2, 6, 1), rt.vector(0, 0, 0.3), rt.vector(0.3, 0, 0), rt.color(0.4, 0.9, 0.4)))
    m = rt.Phong.default().with_specular(0.0)
    w.add_item(rt.plane().with_material(m.with_pattern(rt.checkers_pattern(rt.color(0.9, 0.9, 0.9), rt.color(0.2, 0.3, 0.4))).with_diffuse(0.8)))
    w.add_item(rt.plane().with_transform(rt.translation(0, 0, 6) * rt.rotation_x(math.pi / 2.0)).with_material(m.with_rgb(0.7, 0.4, 0.3).with_diffuse(0.7)))
    w.add_item(rt.sphere().with_transform(rt.translation(-1.6, 1, 1)).with_material(m.with_rgb(0.9, 0.6, 0.2).with_diffuse(0.6).with_specular(0.1)))
    w.add_item(rt.sphere().with_transform(rt.translation(1.2, 0.8, 0.2) * rt.scaling(0.8, 0.8, 0.8)).with_material(
        m.with_rgb(0.1, 0.1, 0.1).with_reflective(0.8).with_specular(0.6)))
    w.add_item(rt.sphere().with_transform(rt.translation(-0.2, 0.7, -1.6) * rt.scaling(0.7, 0.7, 0.7)).with_material(
        m.with_rgb(1, 1, 1).with_diffuse(0.0).with_transparency(0.9).with_reflective(0.9).with_refractive_index(1.5)))
    g = rt.group()
    g = g.with_child(rt.cube().with_transform(rt.translation(3, 0.5, 2) * rt.scaling(0.5, 0.5, 0.5)).with_material(m.with_rgb(0.3, 0.8, 0.3).with_diffuse(0.9)))
    g = g.with_child(rt.cylinder(0, 1.5, True).with_transform(rt.translation(-3.2, 0, 2.5) * rt.scaling(0.6, 1, 0.6)).with_material(m.with_rgb(0.4, 0.4, 0.9).with_diffuse(0.8)))
    w.add_item(rt.build_bounding_tree(g, 1))
    w.finalize_scene()
    return w


CASES = [
    ("box", box, {}),
    ("box-direct", box, {"direct": True, "diffuse": False, "caustic": False}),
    ("box-caustic", box, {"direct": True, "diffuse": False, "caustic": True}),
    ("lights", lights_room, {"direct": True}),
]


def power_of(ph):
    return ph["power_rgb"].astype(np.float64) * ph["power_scale"].astype(np.float64)[:, None] / 255.0


def compare(got, gst, ref, rst, exact):
    """exact (the f64 validation mode): the same photons in the same order - positions and directions to 1e-5 (f32
    records of f64 paths), powers to a byte step - but for the few paths a last-bit difference of the two libms sends
    another way (at most 0.5 % of the photons may belong to such paths).  Otherwise (f32): the same population - count
    within 2 %, rays within 2 %, total stored power within 2 %, and 95 % of the photons of the first bounce (direct hits,
    which no earlier rounding can have moved) equal to 1e-3."""
    assert gst["n_truncated"] == rst["n_truncated"] == 0
    if exact:
        if len(got) == len(ref) and gst["n_rays"] == rst["n_rays"]:
            dp = np.abs(got["position"] - ref["position"]).max(axis=1)
            dd = np.abs(got["direction"] - ref["direction"]).max(axis=1)
            dw = np.abs(power_of(got) - power_of(ref)).max(axis=1) / np.maximum(power_of(ref).max(axis=1), 1e-30)
            bad = (dp > 1e-5) | (dd > 1e-5) | (dw > 2.0 / 255.0)
            assert bad.mean() <= 0.005, float(bad.mean())
            return "same order, %d of %d photons off" % (int(bad.sum()), len(ref))
        assert abs(len(got) - len(ref)) <= max(4, 0.005 * len(ref)), (len(got), len(ref))
        return "counts differ by %d (a path went another way)" % (len(got) - len(ref))
    assert abs(len(got) - len(ref)) <= 0.02 * len(ref) + 8, (len(got), len(ref))
    assert abs(gst["n_rays"] - rst["n_rays"]) <= 0.02 * rst["n_rays"] + 8
    pg, pr = power_of(got).sum(), power_of(ref).sum()
    assert abs(pg - pr) <= 0.02 * pr + 1e-12, (pg, pr)
    return "population alike: power %.4g / %.4g" % (pg, pr)
