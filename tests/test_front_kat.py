"""Host front end (include/rtmi.hpp through the pybind11 binding) against the reference's unit tests for the host-side
rows of SURVEY §8(a): matrix.rs, camera.rs (construction), canvas.rs, color.rs, obj_loader.rs, shapes/mod.rs (transform
push-down, build_bounding_tree), plus the C-ABI surface (every symbol include/rtmi.h declares)."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

import raytracing_amd as rt
from raytracing_amd import abi
from kat_cases import near, FRAC_1_SQRT_2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def M(*rows):
    return rt.Matrix.from_rows([float(x) for x in rows])


def mat_near(a, b, tol=1e-5):
    near(a.into_flat(), b.into_flat(), tol)


A = M(1, 2, 3, 4, 5, 6, 7, 8, 9, 8, 7, 6, 5, 4, 3, 2)
B = M(-2, 1, 2, 3, 3, 2, 1, -1, 4, 3, 6, 5, 1, 2, 7, 8)
AB = M(20, 22, 50, 48, 44, 54, 114, 108, 40, 58, 110, 102, 16, 26, 46, 42)


def test_matrix_basics():  # matrix.rs:361-470
    m = M(1, 2, 3, 4, 5.5, 6.5, 7.5, 8.5, 9, 10, 11, 12, 13.5, 14.5, 15.5, 16.5)
    assert (m(0, 0), m(0, 3), m(1, 0), m(1, 2), m(2, 2), m(3, 0), m(3, 2)) == (1.0, 4.0, 5.5, 7.5, 11.0, 13.5, 15.5)
    assert A.approx_eq(M(1, 2, 3, 4, 5, 6, 7, 8, 9, 8, 7, 6, 5, 4, 3, 2))
    assert not A.approx_eq(M(2, 1, 3, 4, 5, 6, 7, 8, 9, 8, 7, 6, 5, 4, 3, 2))
    mat_near(A * B, AB)
    mat_near((B.transpose() * A.transpose()).transpose(), AB)
    mat_near(A.transpose().transpose() * B, AB)
    mat_near(A * B.transpose().transpose(), AB)


def test_matrix_transform_transpose_identity():  # matrix.rs:472-560
    a = M(1, 2, 3, 4, 2, 4, 4, 2, 8, 6, 4, 1, 0, 0, 0, 1)
    near((a * rt.point(1, 2, 3)).tuple(), (18, 24, 33))
    near((a * rt.vector(1, 2, 3)).tuple(), (14, 22, 32))
    x = M(0, 1, 2, 4, 1, 2, 4, 8, 2, 4, 8, 16, 4, 8, 16, 32)
    mat_near(rt.Matrix.identity() * x * rt.Matrix.identity(), x)
    near((rt.Matrix.identity() * rt.point(1, 2, 4)).tuple(), (1, 2, 4))
    mat_near(a.transpose(), M(1, 2, 8, 0, 2, 4, 6, 0, 3, 4, 4, 0, 4, 2, 1, 1))
    mat_near(rt.Matrix.identity().transpose(), rt.Matrix.identity())


def test_matrix_inverse():  # matrix.rs:562-640
    assert M(6, 4, 4, 4, 5, 5, 7, 6, 4, -9, 3, -7, 9, 1, 7, -6).is_invertible()
    assert not M(6, 4, 4, 4, 5, 5, 7, 6, 4, -9, 3, -7, 0, 0, 0, 0).is_invertible()
    a = M(-5, 2, 6, -8, 1, -5, 1, 8, 7, 7, -6, -7, 1, -3, 7, 4)
    mat_near(a.inverse(), M(0.21805, 0.45113, 0.24060, -0.04511, -0.80827, -1.45677, -0.44361, 0.52068, -0.07895, -0.22368, -0.05263,
                            0.19737, -0.52256, -0.81391, -0.30075, 0.30639))
    mat_near(a.transpose().inverse(), a.inverse().transpose())
    p = M(3, -9, 7, 3, 3, -8, 2, -9, -4, 4, 4, 1, -6, 5, -1, 1)
    q = M(8, 2, 2, 2, 3, -1, 7, 0, 7, 0, 5, 4, 6, -2, 0, 5)
    mat_near((p * q) * q.inverse(), p)


def test_transformations():  # matrix.rs:642-861
    s2 = math.sqrt(2.0) / 2.0
    t = rt.translation(5, -3, 2)
    near((t * rt.point(-3, 4, 5)).tuple(), (2, 1, 7))
    near((t.inverse() * rt.point(-3, 4, 5)).tuple(), (-8, 7, 3))
    near((t * rt.vector(-3, 4, 5)).tuple(), (-3, 4, 5))
    s = rt.scaling(2, 3, 4)
    near((s * rt.point(-4, 6, 8)).tuple(), (-8, 18, 32))
    near((s * rt.vector(-4, 6, 8)).tuple(), (-8, 18, 32))
    near((s.inverse() * rt.vector(-4, 6, 8)).tuple(), (-2, 2, 2))
    near((rt.scaling(-1, 1, 1) * rt.point(2, 3, 4)).tuple(), (-2, 3, 4))
    near((rt.rotation_x(math.pi / 4) * rt.point(0, 1, 0)).tuple(), (0, s2, s2))
    near((rt.rotation_x(math.pi / 2) * rt.point(0, 1, 0)).tuple(), (0, 0, 1))
    near((rt.rotation_x(math.pi / 4).inverse() * rt.point(0, 1, 0)).tuple(), (0, s2, -s2))
    near((rt.rotation_y(math.pi / 4) * rt.point(0, 0, 1)).tuple(), (s2, 0, s2))
    near((rt.rotation_y(math.pi / 2) * rt.point(0, 0, 1)).tuple(), (1, 0, 0))
    near((rt.rotation_z(math.pi / 4) * rt.point(0, 1, 0)).tuple(), (-s2, s2, 0))
    near((rt.rotation_z(math.pi / 2) * rt.point(0, 1, 0)).tuple(), (-1, 0, 0))
    for args, want in [((1, 0, 0, 0, 0, 0), (5, 3, 4)), ((0, 1, 0, 0, 0, 0), (6, 3, 4)), ((0, 0, 1, 0, 0, 0), (2, 5, 4)),
                       ((0, 0, 0, 1, 0, 0), (2, 7, 4)), ((0, 0, 0, 0, 1, 0), (2, 3, 6)), ((0, 0, 0, 0, 0, 1), (2, 3, 7))]:
        near((rt.shearing(*args) * rt.point(2, 3, 4)).tuple(), want)
    a, b, c = rt.rotation_x(math.pi / 2), rt.scaling(5, 5, 5), rt.translation(10, 5, 7)
    p = rt.point(1, 0, 1)
    near((a * p).tuple(), (1, -1, 0))
    near((b * (a * p)).tuple(), (5, -5, 0))
    near((c * (b * (a * p))).tuple(), (15, 0, 7))
    near(((c * b * a) * p).tuple(), (15, 0, 7))
    chain = rt.Matrix.identity().rotate(math.pi / 2, rt.vector(1, 0, 0)).scale(5, 5, 5).translate(10, 5, 7)
    near((chain * p).tuple(), (15, 0, 7))


def test_view_transform():  # matrix.rs:863-909
    up = rt.vector(0, 1, 0)
    mat_near(rt.view_transform(rt.point(0, 0, 0), rt.point(0, 0, -1), up), rt.Matrix.identity())
    mat_near(rt.view_transform(rt.point(0, 0, 0), rt.point(0, 0, 1), up), rt.scaling(-1, 1, -1))
    mat_near(rt.view_transform(rt.point(0, 0, 8), rt.point(0, 0, 0), up), rt.translation(0, 0, -8))
    mat_near(rt.view_transform(rt.point(1, 3, 2), rt.point(4, -2, 8), rt.vector(1, 1, 0)),
             M(-0.50709, 0.50709, 0.67612, -2.36643, 0.76772, 0.60609, 0.12122, -2.82843, -0.35857, 0.59761, -0.71714, 0, 0, 0, 0, 1))


def test_camera_construction():  # camera.rs:322-347
    c = rt.Camera(160, 120, math.pi / 2)
    assert (c.hsize(), c.vsize(), c.field_of_view()) == (160, 120, math.pi / 2)
    assert c.transform().is_identity()
    assert abs(rt.Camera(200, 125, math.pi / 2).pixel_size() - 0.01) < 1e-5
    assert abs(rt.Camera(125, 200, math.pi / 2).pixel_size() - 0.01) < 1e-5


def test_colors():  # color.rs:170-209
    c1, c2 = rt.color(0.9, 0.6, 0.75), rt.color(0.7, 0.1, 0.25)
    near((c1 + c2).tuple(), (1.6, 0.7, 1.0))
    near((c1 - c2).tuple(), (0.2, 0.5, 0.5))
    near((rt.color(0.2, 0.3, 0.4) * 2).tuple(), (0.4, 0.6, 0.8))
    near((2 * rt.color(0.2, 0.3, 0.4)).tuple(), (0.4, 0.6, 0.8))
    near((rt.color(1.0, 0.2, 0.4) * rt.color(0.9, 1, 0.1)).tuple(), (0.9, 0.2, 0.04))
    assert rt.color(1.5, 0.5, -0.5).to_u8() == (255, 128, 0)


def test_canvas_and_ppm():  # canvas.rs:193-271
    c = rt.canvas(10, 20)
    assert (c.width(), c.height()) == (10, 20)
    assert c.get_pixel(3, 4).tuple() == (0, 0, 0)
    c.set_pixel(2, 3, rt.color(1, 0, 0))
    assert c.get_pixel(2, 3).tuple() == (1, 0, 0)
    c = rt.canvas(5, 3)
    assert c.to_ppm().splitlines()[:3] == ["P3", "5 3", "255"]
    c.set_pixel(0, 0, rt.color(1.5, 0, 0))
    c.set_pixel(2, 1, rt.color(0, 0.5, 0))
    c.set_pixel(4, 2, rt.color(-0.5, 0, 1))
    assert c.to_ppm().splitlines()[3:6] == ["255 0 0 0 0 0 0 0 0 0 0 0 0 0 0", "0 0 0 0 0 0 0 128 0 0 0 0 0 0 0", "0 0 0 0 0 0 0 0 0 0 0 0 0 0 255"]
    c = rt.canvas(10, 2)
    c.clear(rt.color(1, 0.8, 0.6))
    assert c.to_ppm().splitlines()[3:7] == ["255 204 153 255 204 153 255 204 153 255 204 153 255 204 153 255 204",
                                            "153 255 204 153 255 204 153 255 204 153 255 204 153"] * 2
    assert rt.canvas(5, 3).to_ppm().endswith("\n")


OBJ = "v -1 1 0\nv -1 0 0\nv 1 0 0\nv 1 1 0\n\ng FirstGroup\nf 1 2 3\ng SecondGroup\nf 1 3 4\n"


def tri_points(item):
    s = item.as_shape()
    return [s.p1.tuple(), s.p2.tuple(), s.p3.tuple()]


def test_obj_parser():  # obj_loader.rs:176-342
    gib = "There was a young lady named Bright\nwho traveled much faster than light.\nShe set out one day\nin a relative way,\nand came back the previous night."
    assert len(rt.ObjParser.parse_str(gib).ignored()) == 5
    p = rt.ObjParser.parse_str("v -1 1 0\nv -1.0000 0.5000 0.0000\nv 1 0 0\nv 1 1 0")
    assert [v.tuple() for v in p.vertices] == [(-1, 1, 0), (-1, 0.5, 0), (1, 0, 0), (1, 1, 0)]
    p = rt.ObjParser.parse_str("v -1 1 0\nv -1 0 0\nv 1 0 0\nv 1 1 0\n\nf 1 2 3\nf 1 3 4")
    g = p.get_group("default_group")
    V = [v.tuple() for v in p.vertices]
    assert tri_points(g.get_child(0)) == [V[0], V[1], V[2]]
    assert tri_points(g.get_child(1)) == [V[0], V[2], V[3]]
    p = rt.ObjParser.parse_str("v -1 1 0\nv -1 0 0\nv 1 0 0\nv 1 1 0\nv 0 2 0\n\nf 1 2 3 4 5")  # fan triangulation
    g = p.get_group("default_group")
    V = [v.tuple() for v in p.vertices]
    assert [tri_points(g.get_child(i)) for i in range(3)] == [[V[0], V[1], V[2]], [V[0], V[2], V[3]], [V[0], V[3], V[4]]]
    p = rt.ObjParser.parse_str(OBJ)
    V = [v.tuple() for v in p.vertices]
    assert tri_points(p.get_group("FirstGroup").get_child(0)) == [V[0], V[1], V[2]]
    assert tri_points(p.get_group("SecondGroup").get_child(0)) == [V[0], V[2], V[3]]
    grp = rt.ObjParser.parse_str(OBJ).into_group()  # convert_to_group: empty default group is dropped
    assert len(grp) == 2
    assert tri_points(grp.get_child(0).as_group().get_child(0)) == [V[0], V[1], V[2]]
    p = rt.ObjParser.parse_str("vn 0 0 1\nvn 0.707 0 -0.707\nvn 1 2 3")
    assert [n.tuple() for n in p.normals] == [(0, 0, 1), (0.707, 0, -0.707), (1, 2, 3)]
    p = rt.ObjParser.parse_str("v 0 1 0\nv -1 0 0\nv 1 0 0\n\nvn -1 0 0\nvn 1 0 0\nvn 0 1 0\n\nf 1//3 2//1 3//2\nf 1/0/3 2/102/1 3/14/2")
    g = p.get_group("default_group")
    for i in range(2):
        s = g.get_child(i).as_shape()
        assert s.kind == 6
        assert [s.n1.tuple(), s.n2.tuple(), s.n3.tuple()] == [(0, 1, 0), (-1, 0, 0), (1, 0, 0)]


def test_group_transform_pushdown():  # shapes/mod.rs:778-804
    g1 = rt.group().with_transform(rt.rotation_y(math.pi / 2))
    g2 = rt.group().with_transform(rt.scaling(2, 2, 2))
    g2.add_child(rt.sphere().with_transform(rt.translation(5, 0, 0)))
    g1.add_child(g2)
    g1.update_transform(rt.Matrix.identity())
    s = g1.get_child(0).as_group().get_child(0).as_shape()
    near(s.world_to_object(rt.point(-2, 0, -10)).tuple(), (0, 0, -1))
    g1 = rt.group().with_transform(rt.rotation_y(math.pi / 2))
    g2 = rt.group().with_transform(rt.scaling(1, 2, 3))
    g2.add_child(rt.sphere().with_transform(rt.translation(5, 0, 0)))
    g1.add_child(g2)
    g1.update_transform(rt.Matrix.identity())
    s = g1.get_child(0).as_group().get_child(0).as_shape()
    s33 = math.sqrt(3.0) / 3.0
    near(s.normal_to_world(rt.vector(s33, s33, s33)).tuple(), (0.28571, 0.42857, -0.85714))


def test_build_bounding_tree_teapot():
    """shapes/mod.rs:489-563 has no reference test ("parity unpinned"); properties: every leaf keeps <= max_leaf shapes
    unless a split was impossible, no triangle is lost, and the root box contains every child box."""
    from raytracing_amd import scenes
    world, _ = scenes.chapter15(64, 32)
    f = world.flatten()
    assert f.n_leaves == 6320 + 1  # 6,320 smooth triangles + the floor plane
    kinds = [f.node(i)[0] for i in range(f.n_nodes)]
    assert kinds.count(0) == f.n_leaves
    sizes = [f.node(i)[2] for i in range(f.n_nodes) if f.node(i)[0] == 2 and all(f.node(c)[0] == 0 for c in f.children[f.node(i)[1]:f.node(i)[1] + f.node(i)[2]])]
    assert max(sizes) <= 10 or sorted(sizes)[-5] <= 10
    root = f.node(f.top_nodes[1])
    assert root[0] == 2
    lo, hi = np.array(root[3]), np.array(root[4])
    for i in range(f.n_nodes):
        k = f.node(i)
        if k[0] == 2:
            assert np.all(np.array(k[3]) >= lo - 1e-9) and np.all(np.array(k[4]) <= hi + 1e-9)


def test_flatten_cast_shadow_and_materials():
    from raytracing_amd import scenes
    w, _ = scenes.chapter11b(32, 16)
    f = w.flatten()
    assert list(f.top_cast_shadow) == [1, 0, 1, 1, 1]  # the water plane does not cast shadows (chapter-11b.rs:39-42)
    w, _ = scenes.chapter16(32, 16, nx=1, ny=1)
    f = w.flatten()
    assert f.n_leaves == 1 + 2 + 9 * 23  # plane + lens + 9 dice x 23 leaves
    assert list(f.top_cast_shadow) == [1, 0, 1]


def test_c_abi_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "rtmi.h")).read()
    declared = set(re.findall(r"\b(rtmi_[a-z0-9_]+)\s*\(", header))
    assert declared == set(abi.SYMBOLS), declared ^ set(abi.SYMBOLS)
    lib = abi.lib()
    for s in declared:
        assert getattr(lib, s) is not None
    assert abi.describe().startswith("rtmi abi=3 target=gfx950")


def test_c_abi_struct_sizes_match():
    """ctypes mirrors vs the compiler's layout (checked through a tiny C probe compiled against include/rtmi.h)."""
    import subprocess
    import tempfile
    src = '#include "rtmi.h"\n#include <stdio.h>\nint main(){printf("%zu %zu %zu %d\\n", sizeof(rtmi_render_opts), sizeof(rtmi_stats), sizeof(rtmi_hit_record), RTMI_N_STATS);return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "p.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "p.c"), "-o", os.path.join(d, "p")])
        out = subprocess.check_output([os.path.join(d, "p")]).split()
    assert [int(x) for x in out] == [C.sizeof(abi.RenderOpts), C.sizeof(abi.Stats), C.sizeof(abi.HitRecord), abi.N_STATS]


def test_rank_rows_partition():
    """Row-tile ownership (rtmi_rank_rows): the ranks' rows partition the frame for any rank count."""
    for v in (1, 7, 8, 100, 2160):
        for n in (1, 2, 3, 4, 8):
            for tr in (8, 16):
                assert sum(abi.rank_rows(v, r, n, tr) for r in range(n)) == v


def test_scene_validation_errors_are_loud():
    """No GPU here: creation must fail with RTMI_ERR_DEVICE for a valid scene, and with a descriptive error for a bad one."""
    w = rt.World.default()
    w.set_max_reflection_depth(99)
    with pytest.raises(abi.RtmiError, match="max_depth"):
        abi.Scene(w)


def test_photon_map_validation_and_stored_photon():
    """The photon fields of the descriptor are checked before any device is touched; StoredPhoton::new / scale_power / power of the
    front end are the reference's (photon_map.rs:23-64,733-750)."""
    import numpy as np
    w = rt.World.default()
    pos = np.zeros((3, 3), dtype=np.float32); dirs = np.zeros((3, 3), dtype=np.float32)
    for bad in (0, 129):
        w.set_photon_arrays(pos, dirs, np.ones(3, dtype=np.float32), np.full((3, 3), 255, dtype=np.uint8), bad, 0.5)
        with pytest.raises(abi.RtmiError, match="n_nearest"):
            abi.Scene(w)
    w.drop_photon_map()
    assert w.n_photons() == 0 and w.photon_map_enabled()
    q = rt.StoredPhoton(rt.point(1, 2, 3), rt.vector(0, 0, 1), rt.color(1e2, 1e3, 1e-3))
    assert q.power_rgb() == [25, 255, 0] and abs(q.power_scale() - 1e3) < 1e-3
    c = q.power()
    assert abs(c.red - 98.03922) < 1e-4 and abs(c.green - 1e3) < 1e-3 and c.blue == 0.0
    assert abs(q.scale_power(0.5).power().green - 500.0) < 1e-3
    w.set_photon_map([q, q.scale_power(0.25)], 2, 1.5)
    assert w.n_photons() == 2


def test_cpp_example_compiles_and_fails_loudly_without_a_gpu(tmp_path):
    """examples/render_ppm.cpp (the C++ front end of INTEGRATION.md section 5) builds against include/ + librtmi.so; on a
    machine without a GPU it must exit non-zero with the library's message instead of producing an image."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "render_ppm")
    libdir = os.path.join(root, "raytracing_amd")
    subprocess.run(["g++", "-std=c++17", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "render_ppm.cpp"),
                    "-L" + libdir, "-lrtmi", "-Wl,-rpath," + libdir, "-o", exe], check=True)
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: the example is run by the gpu tests")
    r = subprocess.run([exe, str(tmp_path / "o.ppm"), "32"], capture_output=True, text=True)
    assert r.returncode != 0 and "no CPU path" in r.stderr


def test_triangle_mesh_flattens_into_a_group_of_its_faces():  # triangle.rs:164-192; include/rtmi.hpp TriangleMesh
    verts = [rt.point(0, 0, 0), rt.point(1, 0, 0), rt.point(0, 1, 0), rt.point(0, 0, 1)]
    mesh = rt.TriangleMesh.from_mesh(verts, [[0, 1, 2], [0, 2, 3], [0, 3, 1], [1, 3, 2]])
    assert mesh.aint_empty() and len(mesh) == 4 and not rt.TriangleMesh.new().aint_empty()
    t = mesh.get_triangle(1)
    assert (t.p1.x, t.p1.y, t.p1.z, t.p2.y, t.p3.z) == (0.0, 0.0, 0.0, 1.0, 1.0)
    mesh.add_triangle(rt.triangle(verts[1], verts[2], verts[3]))
    assert len(mesh) == 5
    with pytest.raises(Exception):
        mesh.add_triangle(rt.smooth_triangle(verts[0], verts[1], verts[2], rt.vector(0, 1, 0), rt.vector(0, 1, 0), rt.vector(0, 1, 0)))
    with pytest.raises(Exception):
        rt.TriangleMesh.from_mesh(verts, [[0, 1]])
    mat = rt.Phong.default().with_rgb(0.2, 0.4, 0.6)
    w = rt.World.empty()
    w.add_item(mesh.into_group(mat).with_transform(rt.translation(1, 2, 3)))
    w.finalize_scene()
    f = w.flatten()
    assert f.n_leaves == 5  # one RTMI_TRIANGLE leaf per face, all with the mesh's transform and material
