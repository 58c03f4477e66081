"""rtmi_trace_photons (the tracing half of World::compute_photon_map, world.rs:186-273) on the device against the
oracle's orc_trace_photons on the same Philox stream, through the C ABI.

f64 validation mode: the same photons in the same order (the reference's order: by emission, then by bounce).  f32
product path: the same population (tests/photon_trace_cases.compare gives both bars).  The reference itself draws
from thread_rng and has no test of this half beyond reflect / refract (tests/test_oracle_photon_trace.py)."""
import numpy as np
import pytest

import oracle_lib as orc
import photon_trace_cases as cases
from raytracing_amd import abi, scenes
from raytracing_amd import _front as rt

pytestmark = pytest.mark.gpu
N = 20000


@pytest.mark.parametrize("name,make,flags", cases.CASES, ids=[c[0] for c in cases.CASES])
@pytest.mark.parametrize("precision", [1, 0], ids=["f64", "f32"])
def test_against_the_oracle(name, make, flags, precision):
    world = make()
    sc = abi.Scene(world)
    opts = abi.make_photon_opts(N, seed=11, precision=precision, **flags)
    ref, rst = orc.trace_photons(world.flatten(), opts)
    got, gst = abi.trace_photons(sc, opts)
    sc.close()
    assert rst["n_stored"] > N // 4
    cases.compare(got, gst, ref, rst, exact=bool(precision))
    assert gst["kernel_ms"] > 0


def test_a_call_is_reproducible_and_its_prefix_is_a_prefix():
    world = cases.lights_room()
    sc = abi.Scene(world)
    opts = abi.make_photon_opts(50000, seed=5, direct=True)
    a, ast = abi.trace_photons(sc, opts)
    b, bst = abi.trace_photons(sc, opts)
    head, hst = abi.trace_photons(sc, opts, capacity=1000)
    other, _ = abi.trace_photons(sc, abi.make_photon_opts(50000, seed=6, direct=True))
    sc.close()
    assert np.array_equal(a, b) and {k: v for k, v in ast.items() if k != "kernel_ms"} == {k: v for k, v in bst.items() if k != "kernel_ms"}
    assert hst["n_stored"] == ast["n_stored"] == len(a) and np.array_equal(head, a[:1000])
    assert not np.array_equal(a[:1000]["position"], other[:1000]["position"])


def test_more_hits_than_the_first_guess_holds():
    """the device buffer starts at 2 hits per emitted photon; a long-lived population (diffuse 0.9 everywhere, a closed room)
    stores more and the call runs again with the exact size"""
    w = rt.World.empty()
    w.add_light(rt.PointLight(rt.point(0, 2, 0), rt.color(1, 1, 1)))
    m = rt.Phong.default().with_rgb(1, 1, 1).with_diffuse(0.95).with_specular(0.0)
    w.add_item(rt.cube().with_transform(rt.scaling(4, 4, 4)).with_material(m))
    w.finalize_scene()
    sc = abi.Scene(w)
    opts = abi.make_photon_opts(4000, seed=2, direct=True, precision=1)
    got, gst = abi.trace_photons(sc, opts)
    sc.close()
    ref, rst = orc.trace_photons(w.flatten(), opts)
    assert rst["n_stored"] > 2 * 4000 + 1024
    cases.compare(got, gst, ref, rst, exact=True)


def test_max_bounces_cuts_a_lossless_path():
    """a mirror room: max(specular, reflective) = 1 and nothing else, p_absorb = 0 - the reference never returns; here the
    path ends after max_bounces and is counted"""
    w = rt.World.empty()
    w.add_light(rt.PointLight(rt.point(0.3, 0.2, 0.1), rt.color(1, 1, 1)))
    m = rt.Phong.default().with_rgb(0, 0, 0).with_diffuse(0.0).with_specular(0.0).with_reflective(1.0)
    w.add_item(rt.cube().with_transform(rt.scaling(3, 2, 4)).with_material(m))
    w.finalize_scene()
    sc = abi.Scene(w)
    for prec in (0, 1):
        opts = abi.make_photon_opts(512, seed=2, precision=prec, max_bounces=40)
        got, gst = abi.trace_photons(sc, opts)
        assert gst["n_truncated"] == 512 and gst["n_rays"] == 512 * 40 and gst["n_stored"] == 0 and len(got) == 0
    sc.close()
    ref, rst = orc.trace_photons(w.flatten(), abi.make_photon_opts(512, seed=2, max_bounces=40))
    assert rst["n_truncated"] == 512 and rst["n_rays"] == 512 * 40


def test_the_panics_of_the_reference_are_errors():
    world = scenes.photon_room(16, 16)[0]  # Phong::default() glass: diffuse 0.9 + specular 0.9 + transparency 1 (materials.rs:443-445)
    sc = abi.Scene(world)
    with pytest.raises(abi.RtmiError, match="materials.rs:443"):
        abi.trace_photons(sc, abi.make_photon_opts(5000))
    with pytest.raises(abi.RtmiError, match="n_emit"):
        abi.trace_photons(sc, abi.make_photon_opts(0))
    with pytest.raises(abi.RtmiError, match="max_bounces"):
        abi.trace_photons(sc, abi.make_photon_opts(10, max_bounces=70000))
    sc.close()
    w2 = rt.World.empty()
    w2.add_light(rt.AmbientLight(rt.color(1, 1, 1)))
    w2.add_item(rt.sphere())
    w2.finalize_scene()
    sc = abi.Scene(w2)
    with pytest.raises(abi.RtmiError, match="world.rs:212"):
        abi.trace_photons(sc, abi.make_photon_opts(10))
    sc.close()


def test_compute_photon_map_then_render_against_the_oracle():
    """World::compute_photon_map (world.rs:186-209) as the front end offers it: trace on the device, keep the map, render
    with it (materials.rs:376-384) - the frame against the oracle's render of the same world with the same map."""
    world, cam = scenes.photon_box(96, 96, deterministic=True, direct=False)
    world.enable_direct_photon_map(True)
    res = world.compute_photon_map(60000, 50, 0.5, seed=4)
    assert res["n_stored"] == world.n_photons() > 60000
    ph = np.frombuffer(world.photon_bytes(), dtype=abi.photon_dtype())
    sc0 = abi.Scene(scenes.photon_box(96, 96)[0])  # the same world without a map
    opts = abi.make_photon_opts(60000, seed=4, direct=True)
    again, _ = abi.trace_photons(sc0, opts)
    sc0.close()
    assert np.array_equal(ph, again)
    img, st = abi.render_host(abi.Scene(world), cam, abi.make_opts(precision=1))
    ref, _ = orc.render(world, cam, abi.make_opts(), n_threads=16)
    d = np.abs(orc.to_u8(ref).astype(int) - orc.to_u8(img.astype(np.float64)).astype(int)).max(axis=2)
    assert (d <= 1).mean() >= 0.998, float((d <= 1).mean())
    assert img.mean() > 0.02  # the room is lit by its photons alone


def test_full_size_population():
    """2 M photons into the example room (the reference's example traces 75 M): counts per emitted photon and the stored
    power agree with the oracle's run on a 10 times smaller sample of another part of the same stream to 2 %"""
    world = cases.box()
    sc = abi.Scene(world)
    n = 2_000_000
    got, gst = abi.trace_photons(sc, abi.make_photon_opts(n, seed=8, direct=True))
    sc.close()
    m = 200000
    ref, rst = orc.trace_photons(world.flatten(), abi.make_photon_opts(m, seed=8, direct=True))
    assert gst["n_truncated"] == 0 and len(got) == gst["n_stored"]
    assert abs(gst["n_stored"] / n - rst["n_stored"] / m) < 0.02 * rst["n_stored"] / m
    assert abs(gst["n_rays"] / n - rst["n_rays"] / m) < 0.02 * rst["n_rays"] / m
    assert abs(cases.power_of(got).sum() - cases.power_of(ref).sum()) < 0.02 * cases.power_of(ref).sum()
    assert np.isfinite(got["position"]).all() and np.isfinite(got["direction"]).all()


def test_the_threaded_balance_builds_the_same_tree():
    """PhotonMap::balance on host threads (maps of 200,000 photons and more) against the one-thread order of work: the
    same kd-tree, so the same frame bit for bit"""
    import os
    world, cam = scenes.photon_box(64, 64, deterministic=True, direct=False)
    world.enable_direct_photon_map(True)
    res = world.compute_photon_map(400000, 40, 0.3, seed=12)
    assert res["n_stored"] > 300000
    os.environ["RTMI_HOST_THREADS"] = "1"
    try:
        one = abi.Scene(world)
    finally:
        del os.environ["RTMI_HOST_THREADS"]
    many = abi.Scene(world)
    os.environ["RTMI_PSELECT_MIN"] = "50000"  # the top levels' selections on all threads too (parallel_select; default: ranges above 400,000)
    try:
        more = abi.Scene(world)
    finally:
        del os.environ["RTMI_PSELECT_MIN"]
    c, _ = abi.render_host(more, cam)
    more.close()
    a, _ = abi.render_host(one, cam)
    b, _ = abi.render_host(many, cam)
    one.close(); many.close()
    assert np.array_equal(a, b) and np.array_equal(a, c) and a.mean() > 0.02


@pytest.mark.parametrize("precision", [1, 0], ids=["f64", "f32"])
def test_shares_one_after_the_other_are_the_whole_call(precision):
    """rtmi_photon_opts.first / count (how ranks divide a map): any split gives the one-call photons bit for bit"""
    from raytracing_amd import tiles
    world = cases.lights_room()
    sc = abi.Scene(world)
    n = 30011
    whole, wst = abi.trace_photons(sc, abi.make_photon_opts(n, seed=17, precision=precision, direct=True))
    for k in (2, 5):
        parts, rays = [], 0
        for r in range(k):
            first, count = tiles.photon_share(n, r, k)
            ph, st = abi.trace_photons(sc, abi.make_photon_opts(n, seed=17, precision=precision, direct=True, first=first, count=count))
            parts.append(ph); rays += st["n_rays"]
        assert np.array_equal(np.concatenate(parts), whole) and rays == wst["n_rays"]
    with pytest.raises(abi.RtmiError, match="share"):
        abi.trace_photons(sc, abi.make_photon_opts(n, first=n - 5, count=10))
    sc.close()
