"""ctypes loader of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE: only tests/, smoke() and
bench.py's cpu_baseline leg may use it (oracle/rt_oracle.h)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
SO = os.path.join(ORACLE_DIR, "liboracle.so")
SCHEME_N1N2 = 1

from raytracing_amd.abi import HitRecord, N_STATS, RenderOpts, make_opts  # noqa: E402  (struct mirrors of include/rtmi.h)

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            subprocess.check_call(["make", "-C", ORACLE_DIR])
        L = C.CDLL(SO)
        L.orc_schlick.restype = C.c_double
        L.orc_schlick.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double]
        L.orc_uniform.restype = C.c_double
        L.orc_uniform.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_render.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(RenderOpts), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_local_intersect.argtypes = [C.c_void_p] * 5
        L.orc_local_normal.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p]
        L.orc_normal_at.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p]
        L.orc_world_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_node_intersect.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_color_at.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_hit_record.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(HitRecord)]
        L.orc_shade_explicit.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double,
                                         C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(HitRecord)]
        L.orc_lighting.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_incoming_at.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_pattern_at.argtypes = [C.c_void_p] * 4
        L.orc_camera_ray.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_csg_eval.argtypes = [C.c_int] * 4
        L.orc_aabb_intersect.argtypes = [C.c_void_p] * 6
        L.orc_philox.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_to_u8.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        L.orc_left_children.restype = C.c_int64
        L.orc_left_children.argtypes = [C.c_int64]
        L.orc_compress_color.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_decompress_color.argtypes = [C.c_float, C.c_void_p, C.c_void_p]
        L.orc_photon_balance.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_find_nearest.argtypes = [C.c_void_p, C.c_int64, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def v3(x):
    return np.ascontiguousarray(x, dtype=np.float64)


class _Args:
    """Keeps temporaries alive for the duration of one foreign call: p = _Args(); f(p(origin), p(direction))."""

    def __init__(self):
        self.keep = []

    def __call__(self, x, dtype=np.float64):
        a = np.ascontiguousarray(x, dtype=dtype)
        self.keep.append(a)
        return a.ctypes.data


def render(world, camera, opts=None, n_threads=8, flags=0, row_stride=1, row_offset=0):
    """Camera::render on the oracle: (float64 [v,h,3], stats dict)."""
    from raytracing_amd.abi import STAT_NAMES
    flat = world.flatten()
    cam = camera.abi()
    opts = opts or make_opts()
    out = np.zeros((camera.vsize(), camera.hsize(), 3), dtype=np.float64)
    st = np.zeros(N_STATS, dtype=np.uint64)
    lib().orc_render(flat.desc_address(), cam.address(), C.byref(opts), n_threads, flags, row_stride, row_offset, out.ctypes.data,
                     st.ctypes.data)
    return out, {k: int(st[i]) for i, k in enumerate(STAT_NAMES)}


def to_u8(img):
    a = np.ascontiguousarray(img, dtype=np.float64)
    out = np.zeros(a.shape, dtype=np.uint8)
    lib().orc_to_u8(a.ctypes.data, a.size, out.ctypes.data)
    return out


def color_at(flat, origin, direction, remaining, flags=0):
    rgb = np.zeros(3)
    p = _Args()
    hit = lib().orc_color_at(flat.desc_address(), p(origin), p(direction), remaining, flags, rgb.ctypes.data)
    return bool(hit), rgb


def hit_record(flat, origin, direction, flags=0):
    rec = HitRecord()
    p = _Args()
    lib().orc_hit_record(flat.desc_address(), p(origin), p(direction), flags, C.byref(rec))
    return rec


def world_intersect(flat, origin, direction, shadow_only=False, cap=64):
    ts = np.zeros(cap)
    leaves = np.zeros(cap, dtype=np.int32)
    p = _Args()
    n = lib().orc_world_intersect(flat.desc_address(), p(origin), p(direction), int(shadow_only), cap, ts.ctypes.data, leaves.ctypes.data)
    return ts[:min(n, cap)], leaves[:min(n, cap)]


def node_intersect(flat, node, origin, direction, cap=64):
    ts = np.zeros(cap)
    leaves = np.zeros(cap, dtype=np.int32)
    p = _Args()
    n = lib().orc_node_intersect(flat.desc_address(), node, p(origin), p(direction), cap, ts.ctypes.data, leaves.ctypes.data)
    return ts[:min(n, cap)], leaves[:min(n, cap)]


def local_intersect(flat, leaf, origin, direction):
    ts = np.zeros(4)
    uv = np.zeros(2)
    p = _Args()
    n = lib().orc_local_intersect(flat.leaf_address(leaf), p(origin), p(direction), ts.ctypes.data, uv.ctypes.data)
    return ts[:n], uv


def local_normal(flat, leaf, point, u=0.0, v=0.0):
    n = np.zeros(3)
    p = _Args()
    lib().orc_local_normal(flat.leaf_address(leaf), p(point), u, v, n.ctypes.data)
    return n


def normal_at(flat, leaf, point, u=0.0, v=0.0):
    n = np.zeros(3)
    p = _Args()
    lib().orc_normal_at(flat.leaf_address(leaf), p(point), u, v, n.ctypes.data)
    return n


def shade_explicit(flat, origin, direction, xs, hit, remaining, what=0, flags=0, uvs=None):
    """xs: list of (t, leaf); hit: (t, leaf).  what: 0 shade_hit, 1 reflected_color, 2 refracted_color."""
    ts = v3([x[0] for x in xs])
    leaves = np.ascontiguousarray([x[1] for x in xs], dtype=np.int32)
    uv = None if uvs is None else v3(uvs)
    rgb = np.zeros(3)
    rec = HitRecord()
    p = _Args()
    ok = lib().orc_shade_explicit(flat.desc_address(), p(origin), p(direction), len(xs), ts.ctypes.data,
                                  leaves.ctypes.data, uv.ctypes.data if uv is not None else None, float(hit[0]), int(hit[1]), remaining,
                                  what, flags, rgb.ctypes.data, C.byref(rec))
    assert ok, "hit not found in intersections"
    return rgb, rec


def camera_ray(camera, px, py, sample=0, seed=0x5EED):
    cam = camera.abi()
    o = np.zeros(3)
    d = np.zeros(3)
    lib().orc_camera_ray(cam.address(), seed, px, py, sample, o.ctypes.data, d.ctypes.data)
    return o, d


# --- photon map (photon_map.rs) ---------------------------------------------------------------------------------
PHOTON_DTYPE = np.dtype([("position", np.float32, 3), ("direction", np.float32, 3), ("power_scale", np.float32), ("power_rgb", np.uint8, 3),
                         ("reserved", np.uint8)])  # rtmi_photon, 32 bytes


def left_children(n):
    return int(lib().orc_left_children(n))


def compress_color(c):
    scale = C.c_float(0)
    rgb = np.zeros(3, dtype=np.uint8)
    cc = v3(c)
    lib().orc_compress_color(cc.ctypes.data, C.byref(scale), rgb.ctypes.data)
    return float(scale.value), rgb


def decompress_color(scale, rgb):
    out = np.zeros(3)
    r = np.ascontiguousarray(rgb, dtype=np.uint8)
    lib().orc_decompress_color(C.c_float(scale), r.ctypes.data, out.ctypes.data)
    return out


def make_photons(positions, directions=None, powers=None):
    """StoredPhoton::new for every row (photon_map.rs:23-30): f32 position / direction, compressed power."""
    positions = np.asarray(positions, dtype=np.float64).reshape(-1, 3)
    n = len(positions)
    ph = np.zeros(n, dtype=PHOTON_DTYPE)
    assert ph.itemsize == 32
    ph["position"] = positions.astype(np.float32)
    ph["direction"] = (np.tile([0.0, 0.0, 1.0], (n, 1)) if directions is None else np.asarray(directions, dtype=np.float64).reshape(-1, 3)).astype(np.float32)
    powers = np.ones((n, 3)) if powers is None else np.asarray(powers, dtype=np.float64).reshape(-1, 3)
    for i in range(n):
        ph["power_scale"][i], ph["power_rgb"][i] = compress_color(powers[i])
    return ph


def photon_balance(photons, lo, hi):
    tree = np.zeros(len(photons), dtype=np.int64)
    lo, hi = v3(lo), v3(hi)
    lib().orc_photon_balance(photons.ctypes.data, len(photons), lo.ctypes.data, hi.ctypes.data, tree.ctypes.data)
    return tree.tolist()


def find_nearest(photons, n_neighbors, p, max_search_radius=0.0):
    found = np.zeros(max(n_neighbors, 1), dtype=np.int64)
    msq = C.c_double(0)
    pp = v3(p)
    k = lib().orc_find_nearest(photons.ctypes.data, len(photons), float(max_search_radius), int(n_neighbors), pp.ctypes.data, found.ctypes.data, C.byref(msq))
    return found[:k].tolist(), float(msq.value)


def trace_photons(flat, opts, capacity=None):
    """orc_trace_photons: (photons in the reference's order, {"n_stored", "n_rays", "n_truncated"}); a negative n_stored is the
    oracle's error code (-1: no light with power; -2 - leaf: absorption probability below zero at that leaf)."""
    L = lib()
    L.orc_trace_photons.restype = C.c_int64
    L.orc_trace_photons.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    rays, trunc = C.c_int64(0), C.c_int64(0)
    if capacity is None:
        capacity = max(int(L.orc_trace_photons(flat.desc_address(), C.addressof(opts), None, 0, C.byref(rays), C.byref(trunc))), 0)
    out = np.zeros(max(capacity, 1), dtype=PHOTON_DTYPE)
    n = int(L.orc_trace_photons(flat.desc_address(), C.addressof(opts), out.ctypes.data, capacity, C.byref(rays), C.byref(trunc)))
    return out[:max(min(n, capacity), 0)], {"n_stored": n, "n_rays": rays.value, "n_truncated": trunc.value}


def photon_reflect(direction, normal):
    out = np.zeros(3)
    L = lib()
    L.orc_photon_reflect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_photon_reflect.restype = None
    d, n = v3(direction), v3(normal)
    L.orc_photon_reflect(d.ctypes.data, n.ctypes.data, out.ctypes.data)
    return out


def photon_refract(direction, normal, n1, n2):
    out = np.zeros(3)
    L = lib()
    L.orc_photon_refract.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p]
    d, n = v3(direction), v3(normal)
    ok = L.orc_photon_refract(d.ctypes.data, n.ctypes.data, n1, n2, out.ctypes.data)
    return out if ok else None


def cosine_sample(direction, seed, index):
    out = np.zeros(3)
    L = lib()
    L.orc_cosine_sample.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]
    L.orc_cosine_sample.restype = None
    d = v3(direction)
    L.orc_cosine_sample(d.ctypes.data, seed, index, out.ctypes.data)
    return out
