"""OBJ / MTL / texture export (multimoda_rs_amd.export) against the reference's writers
(io/output.rs, to_object/*.rs) and its interpolation tests (to_object/interpolation.rs:149-533)."""
import os
import struct
import zlib

import numpy as np
import pytest


@pytest.fixture(scope="module")
def EX(mm):
    return mm.export


@pytest.fixture(scope="module")
def FR(mm):
    from multimoda_rs_amd import frames
    return frames


def test_rust_float_formatting(EX):
    f = EX.rust_f64
    assert [f(v) for v in (1.0, -0.0, 0.5, 1e-7, 1.5e21, 123456789.125, 0.1 + 0.2, -2.0)] == \
        ["1", "-0", "0.5", "0.0000001", "1500000000000000000000", "123456789.125", "0.30000000000000004", "-2"]
    assert f(float("nan")) == "NaN" and f(float("inf")) == "inf" and f(float("-inf")) == "-inf"


def _square(FR, cid, z, r=1.0):
    pts = np.array([[0.0, r, z], [-r, 0.0, z], [0.0, -r, z], [r, 0.0, z]])
    c = FR.Contour(cid, cid, pts, None, None, None, "lumen")
    c.compute_centroid()
    return c


def test_write_obj_mesh_text(EX, FR, tmp_path):                # io/output.rs:10-170
    cs = [_square(FR, 0, 0.0), _square(FR, 1, 0.5, r=2.0)]
    path = tmp_path / "sub" / "m.obj"
    EX.write_obj_mesh(cs, EX.compute_uv_coordinates(cs), str(path), "m.mtl", watertight=True)
    lines = path.read_text().splitlines()
    assert lines[:8] == ["v 0 1 0", "v -1 0 0", "v 0 -1 0", "v 1 0 0", "v 0 2 0.5", "v -2 0 0.5", "v 0 -2 0.5", "v 2 0 0.5"]
    assert lines[8:10] == ["mtllib m.mtl", "usemtl displacement_material"]
    assert lines[10:18] == ["vt 0.125 0.25", "vt 0.375 0.25", "vt 0.625 0.25", "vt 0.875 0.25",
                            "vt 0.125 0.75", "vt 0.375 0.75", "vt 0.625 0.75", "vt 0.875 0.75"]
    assert lines[18:22] == ["vn -0 -1 -0", "vn 1 -0 -0", "vn -0 1 -0", "vn -1 -0 -0"]         # inward normals, -0 kept
    assert lines[26:28] == ["f 1/1/1 2/2/2 5/5/5", "f 5/5/5 2/2/2 6/6/6"]
    assert lines[32:34] == ["f 4/4/4 1/1/1 8/8/8", "f 8/8/8 1/1/1 5/5/5"]                   # wrap-around quad
    assert lines[34:40] == ["v 0 0 0", "vt 0.5 0.5", "vn 0.0 0.0 -1.0", "v 0 0 0.5", "vt 0.5 0.5", "vn 0.0 0.0 1.0"]
    assert lines[40] == "f 1/1/1 2/2/2 9/9/9" and lines[44] == "f 10/10/10 6/6/6 5/5/5" and len(lines) == 48
    shell = tmp_path / "shell.obj"
    EX.write_obj_mesh_without_uv(cs, str(shell), "x.mtl", watertight=False)
    sl = shell.read_text().splitlines()
    assert len(sl) == 8 + 2 + 8 + 8 + 8 and sl[10] == "vt 0 0"
    with pytest.raises(RuntimeError, match="at least two contours"):
        EX.write_obj_mesh(cs[:1], [(0, 0)] * 4, str(shell), "x.mtl", False)
    with pytest.raises(RuntimeError, match="same number of points"):
        EX.write_obj_mesh([cs[0], FR.Contour(1, 1, np.zeros((3, 3)), None, None, None, "lumen")], [], str(shell), "x", False)
    with pytest.raises(RuntimeError, match="UV coordinates must match"):
        EX.write_obj_mesh(cs, [(0, 0)], str(shell), "x.mtl", False)


# ---- to_object/interpolation.rs:160-533 ----------------------------------------------------------------
def _mock_frame(FR, fid, off):
    lum = FR.Contour(fid, fid, [[1.0 + off, 2.0 + off, 3.0 + off], [4.0 + off, 5.0 + off, 6.0 + off]],
                     (2.5 + off, 3.5 + off, 4.5 + off), 1.0 + off, 2.0 + off, "lumen", np.array([True, True]))
    cath = FR.Contour(fid, fid, [[10.0 + off, 20.0 + off, 30.0 + off]], (10.0 + off, 20.0 + off, 30.0 + off), None, None, "catheter")
    eem = FR.Contour(fid, fid, [[7.0 + off, 8.0 + off, 9.0 + off]], (7.0 + off, 8.0 + off, 9.0 + off), None, None, "eem")
    return FR.Frame(fid, [5.0 + off, 6.0 + off, 7.0 + off], lum, {"catheter": cath, "eem": eem}, np.array([off, off, off]))


def _mock_geometry(FR, n):
    return [_mock_frame(FR, i, i * 10.0) for i in range(n)]


def test_interpolate_contours(EX, FR):
    res = EX.interpolate_contours(_mock_geometry(FR, 2), _mock_geometry(FR, 2), 2, ["lumen", "catheter", "eem"])   # :259-298
    assert len(res) == 4 and res[0][0].lumen.points[0, 0] == 1.0 and res[-1][0].lumen.points[0, 0] == 1.0
    mid = res[1]
    assert mid[0].lumen.points[0, 0] == pytest.approx(1.0, abs=1e-5) and mid[0].lumen.points[1, 1] == pytest.approx(5.0, abs=1e-5)
    assert mid[0].centroid[0] == pytest.approx(5.0, abs=1e-5)
    assert mid[0].extras["catheter"].points[0, 2] == pytest.approx(30.0, abs=1e-5)
    assert mid[0].extras["eem"].points[0, 0] == pytest.approx(7.0, abs=1e-5)
    res = EX.interpolate_contours(_mock_geometry(FR, 2), _mock_geometry(FR, 3), 1, ["lumen"])                      # :300-312
    assert [len(g) for g in res] == [2, 2, 3]
    res = EX.interpolate_contours(_mock_geometry(FR, 1), _mock_geometry(FR, 1), 1, ["lumen"])                      # :314-333
    assert len(res[1][0].lumen) == 2 and not res[1][0].extras
    assert len(EX.interpolate_contours(_mock_geometry(FR, 1), _mock_geometry(FR, 1), 0, ["lumen"])) == 2
    bad = _mock_geometry(FR, 1)
    bad[0].lumen = FR.Contour(0, 0, np.zeros((3, 3)), None, None, None, "lumen")
    with pytest.raises(RuntimeError, match="point counts do not match"):
        EX.interpolate_contours(_mock_geometry(FR, 1), bad, 2, ["lumen"])
    # t = step / (steps - 1): 0 and 1 for two steps -> the interpolated geometries equal start and end
    s, e = _mock_geometry(FR, 2), [_mock_frame(FR, i, i * 10.0 + 1.0) for i in range(2)]
    r = EX.interpolate_contours(s, e, 2, ["lumen"])
    assert np.array_equal(r[1][1].lumen.points, s[1].lumen.points) and np.array_equal(r[2][1].lumen.points, e[1].lumen.points)
    assert r[1][0].lumen.aortic_thickness == 1.0 and r[2][0].lumen.aortic_thickness == 2.0


def _read_png(path):
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, hdr = 8, b"", None
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + body) & 0xFFFFFFFF
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    w, h, depth, ctype = hdr[:4]
    ch = 3 if ctype == 2 else 4
    raw = zlib.decompress(idat)
    rows = [raw[y * (1 + w * ch) + 1:(y + 1) * (1 + w * ch)] for y in range(h)]
    return np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(h, w, ch)


def test_process_case_files_and_textures(EX, FR, mm, tmp_path):      # to_object/process.rs:13-62, write_mtl.rs, texture.rs
    a = mm.synthetic_pullback(5, 16, pullback_id=0)
    b = a.copy()
    b.lumen[:, 0] += np.linspace(0.0, 0.4, b.lumen.shape[0])        # a displacement that grows along the pullback
    out = tmp_path / "rest"
    with pytest.raises(RuntimeError, match=r"Some \.obj writes failed:\nFailed \[wall_000_dia - sys\.obj\]: Need at least two"):
        EX.process_case("dia - sys", FR.to_frames(a), FR.to_frames(b), str(tmp_path / "w"), 2, True, ["wall"])
    EX.process_case("dia - sys", FR.to_frames(a), FR.to_frames(b), str(out), 2, True, ["lumen", "catheter"])
    names = sorted(os.listdir(out))
    assert [n for n in names if n.startswith("lumen")] == [f"lumen_{i:03d}_dia - sys.{e}" for i in range(4) for e in ("mtl", "obj", "png")]
    assert [n for n in names if n.startswith("catheter")] == [f"catheter_{i:03d}_dia - sys.{e}" for i in range(4) for e in ("mtl", "obj", "png")]
    assert (out / "lumen_001_dia - sys.mtl").read_text() == \
        "newmtl displacement_material\nKa 1 1 1\nKd 1 1 1\nmap_Kd lumen_001_dia - sys.png\n"
    assert (out / "catheter_000_dia - sys.mtl").read_text().startswith("newmtl black_material\nKa 0 0 0\n")
    obj = (out / "lumen_003_dia - sys.obj").read_text().splitlines()
    assert obj[0] == "v " + " ".join(EX.rust_f64(v) for v in b.lumen[0]) and "mtllib lumen_003_dia - sys.mtl" in obj
    assert obj[5 * 16 + 2] == "vt 0.03125 0.1"                       # (0 + 0.5)/16, (0 + 0.5)/5
    first, last = _read_png(out / "lumen_000_dia - sys.png"), _read_png(out / "lumen_003_dia - sys.png")
    assert first.shape == (5, 16, 3) and (first[..., 0] == 0).all() and (first[..., 2] == 255).all()   # no displacement
    assert last[0, -1, 0] == 255 and last[0, -1, 2] == 0             # largest displacement: last frame is the TOP row
    assert last[-1, 0, 0] == 0 and last[-1, 0, 2] == 255 and (last[..., 1] == 0).all()
    assert (_read_png(out / "catheter_002_dia - sys.png") == 0).all()


def test_single_geometry_writers(EX, mm, tmp_path):                  # entry.rs:740-819, functions.rs:1435-1500
    g = mm.synthetic_pullback(4, 12, pullback_id=1)
    g.label = "rest_dia"
    mm.to_obj(g, str(tmp_path / "a"), watertight=False, filename_prefix="case")
    assert sorted(os.listdir(tmp_path / "a")) == ["case_catheter.mtl", "case_catheter.obj", "case_lumen.mtl", "case_lumen.obj"]
    mm.to_obj(g, str(tmp_path / "b"), contour_types=["lumen", "eem"])
    assert sorted(os.listdir(tmp_path / "b")) == ["lumen.mtl", "lumen.obj"]
    assert (tmp_path / "b" / "lumen.mtl").read_text() == "newmtl material\nKa 1.0 1.0 1.0\nKd 1.0 1.0 1.0\nKs 0.0 0.0 0.0\n"
    assert f"mtllib {tmp_path / 'b' / 'lumen.mtl'}" in (tmp_path / "b" / "lumen.obj").read_text()
    EX.write_single_mode(g, str(tmp_path / "c"), True, ["lumen"])
    assert sorted(os.listdir(tmp_path / "c")) == ["lumen_rest_dia.mtl", "lumen_rest_dia.obj"]
    EX.write_single_geometry("pat", g, str(tmp_path / "d"), True, ["catheter"])
    assert sorted(os.listdir(tmp_path / "d")) == ["pat_catheter.mtl", "pat_catheter.obj"]
    assert (tmp_path / "d" / "pat_catheter.mtl").read_text().startswith("newmtl material\nKa 0.0 0.0 0.0")
