"""The steps either side of the path: Canvas::write_png (canvas.rs:95-110) and the OBJ loader's hardening
(obj_loader.rs:35-36 `vt` quirk kept as the default, opt-in fix; negative indices; v/t faces)."""
import io
import struct
import zlib

import numpy as np
import pytest

from raytracing_amd import _front as rt


def _canvas(w, h, seed=3):
    rng = np.random.default_rng(seed)
    cv = rt.Canvas(w, h)
    vals = rng.uniform(-0.2, 1.3, (h, w, 3))  # (out-of-range values too: Color::to_u8 clips)
    for y in range(h):
        for x in range(w):
            cv.set_pixel(x, y, rt.color(*vals[y, x]))
    return cv


def _chunks(png):
    assert png[:8] == b"\x89PNG\r\n\x1a\n"
    pos, out = 8, []
    while pos < len(png):
        n, typ = struct.unpack(">I4s", png[pos:pos + 8])
        body = png[pos + 8:pos + 8 + n]
        (crc,) = struct.unpack(">I", png[pos + 8 + n:pos + 12 + n])
        assert crc == zlib.crc32(typ + body) & 0xFFFFFFFF, typ
        out.append((typ, body))
        pos += 12 + n
    return out


@pytest.mark.parametrize("w,h", [(1, 1), (7, 5), (300, 200)])  # (300x200x3 + 200 filter bytes > one 65535-byte stored block)
def test_png_payload_is_to_rgb8(w, h):
    cv = _canvas(w, h)
    png = cv.to_png()
    ch = _chunks(png)
    assert [t for t, _ in ch] == [b"IHDR", b"IDAT", b"IEND"]
    assert struct.unpack(">IIBBBBB", ch[0][1]) == (w, h, 8, 2, 0, 0, 0)
    raw = np.frombuffer(zlib.decompress(ch[1][1]), dtype=np.uint8).reshape(h, 1 + 3 * w)  # (zlib checks the Adler-32)
    assert (raw[:, 0] == 0).all()  # filter type None on every scanline
    assert np.array_equal(raw[:, 1:].reshape(h, w, 3), cv.to_rgb8())


def test_png_opens_in_an_image_library(tmp_path):
    PIL = pytest.importorskip("PIL.Image")
    cv = _canvas(33, 17)
    path = str(tmp_path / "c.png")
    assert cv.write_png(path)
    img = PIL.open(path)
    assert img.mode == "RGB" and img.size == (33, 17)
    assert np.array_equal(np.asarray(img), cv.to_rgb8())
    assert PIL.open(io.BytesIO(cv.to_png())).size == (33, 17)


def pts(item):
    s = item.as_shape()
    return [s.p1.tuple(), s.p2.tuple(), s.p3.tuple()]


def nrm(item):
    s = item.as_shape()
    return [s.n1.tuple(), s.n2.tuple(), s.n3.tuple()]


OBJ_VT = """v 0 0 0
v 1 0 0
vt 0.5 0.5 0
vt 0.1 0.2 0
v 0 1 0
f 1 2 3
"""


def test_obj_vt_lines_reference_default_and_fix():
    # the reference pushes `vt` lines as vertices (obj_loader.rs:35-36): face 1 2 3 then uses the first vt as its third corner
    p = rt.ObjParser.parse_str(OBJ_VT)
    assert len(p.vertices) == 5
    assert pts(p.get_group("default_group").get_child(0))[2] == (0.5, 0.5, 0.0)
    # opt-in: texture lines are skipped, the face is the triangle the file means
    q = rt.ObjParser.parse_str(OBJ_VT, skip_texture_lines=True)
    assert len(q.vertices) == 3 and len(q.ignored()) == 2
    assert pts(q.get_group("default_group").get_child(0))[2] == (0.0, 1.0, 0.0)


def test_obj_negative_indices_and_v_t_faces():
    src = "v 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\nvn 0 0 1\nvn 0 1 0\nvn 1 0 0\nf -4 -3 -2\nf -3//-1 -2//-2 -1//-3\nf 1/7 2/8 3/9\n"
    p = rt.ObjParser.parse_str(src)
    g = p.get_group("default_group")
    assert len(g) == 3
    kids = [g.get_child(i) for i in range(3)]
    assert pts(kids[0]) == [(0, 0, 0), (1, 0, 0), (0, 1, 0)]  # -4 -3 -2 of four vertices
    assert pts(kids[1]) == [(1, 0, 0), (0, 1, 0), (0, 0, 1)]
    assert kids[1].as_shape().kind == 6 and nrm(kids[1]) == [(1, 0, 0), (0, 1, 0), (0, 0, 1)]  # -1 -2 -3 of three normals
    assert pts(kids[2]) == [(0, 0, 0), (1, 0, 0), (0, 1, 0)] and kids[2].as_shape().kind == 5  # v/t: a flat triangle
    with pytest.raises(Exception):
        rt.ObjParser.parse_str("v 0 0 0\nf -2//1 1//1 1//1\n")  # further back than the file goes
