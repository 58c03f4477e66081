"""GPU parity of the photon-map gather in Phong::shade_hit (materials.rs:376-384; PhotonMap::from_vec and find_nearest,
photon_map.rs:230-364) against the oracle, through the C ABI.  The map itself is a fixture (tests/photon_cases.py) so that the gather is
tested on its own; the library's own tracing of photons (rtmi_trace_photons) has tests/test_gpu_photon_trace.py.

Bars: f64 validation mode - colours of random rays within 1e-9 of the oracle's (the same kd-tree, the same photons found, the
same fold), every pixel of a frame within 1/255 and equal ray counts; f32 - >= 99.8 % of pixels within 1/255 (near-ties between
the n-th and the n+1-th photon may swap in f32), PSNR >= 40 dB; adaptive frames PSNR >= 60 / 35 dB."""
import numpy as np
import pytest

import oracle_lib as orc
import photon_cases
from raytracing_amd import abi, scenes

pytestmark = pytest.mark.gpu
RAY_KEYS = ("primary", "reflection", "refraction", "shadow", "samples")


def room(h, v, n_photons=20000, n_nearest=50, radius=0.5, seed=1, **kw):
    world, cam = scenes.photon_room(h, v, **kw)
    photon_cases.attach(world, photon_cases.direct_photons(world, (-3, 6, -4), n_photons, seed), n_nearest, radius)
    return world, cam


def psnr(a, b):
    a = np.clip(a, 0, 1); b = np.clip(b, 0, 1)
    mse = float(np.mean((a - b) ** 2))
    return 99.0 if mse == 0 else 10.0 * np.log10(1.0 / mse)


@pytest.mark.parametrize("direct", [True, False], ids=["with-lights", "map-only"])
@pytest.mark.parametrize("n_nearest,radius", [(50, 0.5), (1, 0.0), (128, 0.05), (7, 1e9)])
def test_random_rays_against_the_oracle(direct, n_nearest, radius):
    world, cam = room(64, 32, n_photons=6000, n_nearest=n_nearest, radius=radius, direct=direct)
    flat = world.flatten()
    sc = abi.Scene(world)
    rng = np.random.default_rng(11)
    n = 1500
    o = rng.uniform(-4, 4, (n, 3)) + np.array([0, 2.5, -3.0])
    d = rng.uniform(-2, 2, (n, 3)) + np.array([0, 0.5, 0.5]) - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    c64, r64 = abi.trace_rays(sc, o, d, opts=abi.make_opts(precision=1))
    c32, r32 = abi.trace_rays(sc, o, d, opts=abi.make_opts(precision=0))
    sc.close()
    ref = np.array([orc.color_at(flat, o[i], d[i], world.max_reflection_depth)[1] for i in range(n)])
    assert np.abs(c64 - ref).max() < 1e-9
    close = np.abs(c32 - ref).max(axis=1) < 2e-3 * np.maximum(1.0, np.abs(ref).max(axis=1))
    assert close.mean() >= 0.995, close.mean()
    if not direct:
        assert ref.max() > 1e-3  # (the map is all the light there is: the comparison is of the estimate)


@pytest.mark.parametrize("n_photons", [1, 2, 3, 4, 5, 6, 7, 8, 9, 15, 16, 17, 64])
def test_small_and_degenerate_maps(n_photons):
    """Maps of a handful of photons (every shape of the left-balanced tree's last level), more neighbours asked for than there
    are photons, and coincident photons (both sides order equal coordinates by index): f64 colours equal the oracle's."""
    world, cam = scenes.photon_room(64, 32, direct=False)
    rng = np.random.default_rng(n_photons)
    pos = np.column_stack([rng.uniform(-2, 2, n_photons), np.zeros(n_photons), rng.uniform(-2, 3, n_photons)])
    if n_photons >= 5:
        pos[1] = pos[0]; pos[3] = pos[0]  # coincident
    ph = orc.make_photons(pos, np.tile([0.0, 1.0, 0.0], (n_photons, 1)), np.full((n_photons, 3), 0.5))
    photon_cases.attach(world, ph, 16, 0.0)
    flat = world.flatten()
    sc = abi.Scene(world)
    n = 300
    o = rng.uniform(-3, 3, (n, 3)) + np.array([0, 3.0, -2.0])
    d = np.column_stack([rng.uniform(-2, 2, n), np.zeros(n), rng.uniform(-2, 3, n)]) - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    c64, _ = abi.trace_rays(sc, o, d, opts=abi.make_opts(precision=1))
    c32, _ = abi.trace_rays(sc, o, d, opts=abi.make_opts(precision=0))
    sc.close()
    ref = np.array([orc.color_at(flat, o[i], d[i], world.max_reflection_depth)[1] for i in range(n)])
    assert ref.max() > 0
    assert np.abs(c64 - ref).max() < 1e-9
    assert (np.abs(c32 - ref).max(axis=1) < 2e-3 * np.maximum(1.0, np.abs(ref).max(axis=1))).mean() >= 0.99


@pytest.mark.parametrize("precision", [1, 0], ids=["f64", "f32"])
@pytest.mark.parametrize("direct", [True, False], ids=["with-lights", "map-only"])
def test_frame_one_sample(precision, direct):
    world, cam = room(192, 96, deterministic=True, direct=direct)
    ref, rst = orc.render(world, cam)
    sc = abi.Scene(world)
    img, st = abi.render_host(sc, cam, abi.make_opts(precision=precision))
    sc.close()
    img = img.astype(np.float64)
    d = np.abs(orc.to_u8(img).astype(int) - orc.to_u8(ref).astype(int)).max(axis=2)
    assert st["overflow"] == 0
    if precision == 1:
        assert int((d > 1).sum()) == 0
        assert {k: st[k] for k in RAY_KEYS} == {k: rst[k] for k in RAY_KEYS}
    else:
        assert (d <= 1).mean() >= 0.998, float((d <= 1).mean())
        assert psnr(img, ref) >= 40.0


@pytest.mark.parametrize("precision", [1, 0], ids=["f64", "f32"])
def test_frame_adaptive(precision):
    """adaptive multisampling with the map: the second pass's kernels (resample / units, PHOTON kind) too"""
    world, cam = room(160, 80, n_photons=8000, n_nearest=30)
    ref, rst = orc.render(world, cam)
    sc = abi.Scene(world)
    img, st = abi.render_host(sc, cam, abi.make_opts(precision=precision))
    sc.close()
    p = psnr(img.astype(np.float64), ref)
    assert st["overflow"] == 0
    assert p >= (60.0 if precision == 1 else 35.0), p
    assert abs(st["samples"] - rst["samples"]) <= (0.001 if precision == 1 else 0.05) * rst["samples"]


def test_the_map_changes_nothing_when_disabled():
    """World::photon_map_enabled() false (all three kinds off, world.rs:124-128): the frame of the same world without a map"""
    world, cam = room(96, 48, deterministic=True, n_photons=2000)
    world.enable_diffuse_photon_map(False); world.enable_caustic_photon_map(False)
    plain, _ = scenes.photon_room(96, 48, deterministic=True)
    a = abi.Scene(world); b = abi.Scene(plain)
    ia, _ = abi.render_host(a, cam, abi.make_opts())
    ib, _ = abi.render_host(b, cam, abi.make_opts())
    a.close(); b.close()
    assert np.array_equal(ia, ib)


def test_sharded_frame_is_bit_identical():
    import torch
    from raytracing_amd import tiles
    world, cam = room(200, 100, n_photons=5000, n_nearest=20)
    sc = abi.Scene(world)
    full, _ = abi.render_host(sc, cam, abi.make_opts())
    h, v = cam.hsize(), cam.vsize()
    out = np.zeros((v, h, 3), dtype=np.float32)
    for r in range(3):
        rows = abi.rank_rows(v, r, 3, 8)
        buf = torch.zeros((rows, h, 3), dtype=torch.float32, device="cuda")
        abi.render_device(sc, cam, abi.make_opts(rank=r, nranks=3, tile_rows=8), buf, None, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        out[tiles.rank_row_index(v, r, 3, 8).numpy()] = buf.cpu().numpy()
    sc.close()
    assert np.array_equal(out, full)
