"""mm_parse_contour_table (include/mm_hausdorff.h): the native fast path of read_contour_data must give the
same doubles as Python's correctly rounded float() and reject everything that is not the regular form, so that
the row-by-row reader (which skips invalid rows like the reference, input.rs:172-194) takes over.  Host only."""
import glob
import importlib
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def io():
    import __graft_entry__ as ge
    ge.build()
    return importlib.import_module("multimoda_rs_amd.io")


def _write(tmp_path, text, name="c.csv"):
    p = tmp_path / name
    p.write_bytes(text.encode("utf-8") if isinstance(text, str) else text)
    return str(p)


def test_golden_files_equal_the_python_readers(io):
    files = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "**", "*.csv"), recursive=True))
    files = [f for f in files if "records" not in os.path.basename(f) and "manual" not in os.path.basename(f)
             and "centerline" not in os.path.basename(f)]          # (x,y,z centerline table: three columns, not a contour file)
    assert len(files) >= 12
    for f in files:
        d = io._detect_delimiter(f)
        a = io._read_numeric_table_native(f, d)
        assert a is not None, f
        rows = []
        with open(f, newline="") as fh:
            for line in fh.read().replace("\r\n", "\n").splitlines():
                rec = line.split(d)
                rows.append((float(int(rec[0])), float(rec[1]), float(rec[2]), float(rec[3])))
        assert np.array_equal(a, np.array(rows).reshape(-1, 4)), f


def test_random_numbers_in_every_accepted_notation(io, tmp_path):
    rng = np.random.default_rng(11)
    toks, vals = [], []
    for k in range(4000):
        kind = k % 8
        if kind == 0:
            t = repr(float(rng.normal(0, 100)))                       # shortest repr, 17 significant digits at most
        elif kind == 1:
            t = "%.3f" % rng.uniform(-10, 10)
        elif kind == 2:
            t = "%.12e" % rng.uniform(-1, 1)
        elif kind == 3:
            t = "%+.6E" % (rng.uniform(1, 9) * 10.0 ** int(rng.integers(-300, 300)))
        elif kind == 4:
            t = "".join(rng.choice(list("0123456789"), size=int(rng.integers(1, 30)))) + "." + \
                "".join(rng.choice(list("0123456789"), size=int(rng.integers(0, 30))))   # > 19 digits: strtod path
        elif kind == 5:
            t = ["-0", "0.0", "+0.000", "-0.0e5", "00012.50", ".5", "5.", "1e22", "1e23", "9007199254740993"][k // 8 % 10]
        elif kind == 6:
            t = "%d" % int(rng.integers(-10**9, 10**9))
        else:
            t = "2.2250738585072014e-308" if k % 16 == 7 else "4.9e-324"   # smallest normal / subnormal
        toks.append(t)
        vals.append(float(t))
    for delim, eol, final in ((",", "\n", True), ("\t", "\r\n", True), (",", "\n", False)):
        lines = [delim.join([str(i)] + toks[3 * i:3 * i + 3]) for i in range(len(toks) // 3)]
        text = eol.join(lines) + (eol if final else "")
        a = io._read_numeric_table_native(_write(tmp_path, text), delim)
        assert a is not None and a.shape == (len(lines), 4)
        exp = np.array([[float(i)] + vals[3 * i:3 * i + 3] for i in range(len(lines))])
        assert np.array_equal(a, exp) and np.array_equal(np.signbit(a), np.signbit(exp))


@pytest.mark.parametrize("text", [
    '1,"2.0",3,4\n', "1,2,3\n", "1,2,3,4,5\n", "1,2,3,4\n\n5,6,7,8\n", "1,2,abc,4\n", "1,2,3,4\r5,6,7,8\n",
    "-1,2,3,4\n", "1.5,2,3,4\n", "4294967296,2,3,4\n", "1,inf,3,4\n", "1,nan,3,4\n", "1,1e999,3,4\n", "1, 2,3,4\n",
    "1,2,3,4 \n", "1,0x10,3,4\n", "1,1_0,3,4\n", "1,2e,3,4\n", "1,.,3,4\n", "\n", "1,2,3,4\n5,6,7\n"])
def test_irregular_text_is_left_to_the_row_reader(io, tmp_path, text):
    assert io._read_numeric_table_native(_write(tmp_path, text), ",") is None
    # and the public reader still does what the reference does: skip the rows it cannot parse
    got = io.read_contour_data(_write(tmp_path, text, "d.csv"))
    assert got.shape[1] == 4 and got.shape[0] <= text.count("\n") + 1


def test_empty_file_and_read_contour_data_dispatch(io, tmp_path):
    assert io._read_numeric_table_native(_write(tmp_path, ""), ",") is None
    assert io.read_contour_data(_write(tmp_path, "", "e.csv")).shape == (0, 4)
    a = io.read_contour_data(_write(tmp_path, "3\t1.25\t-2.5\t7\r\n4\t0.1\t0.2\t0.3", "t.csv"))
    assert np.array_equal(a, np.array([[3.0, 1.25, -2.5, 7.0], [4.0, 0.1, 0.2, 0.3]]))


# ---------------------------------------------------------------------------------------
# the rows the reference's csv + serde reader skips (ContourPoint { frame_index: u32, x, y, z: f64, aortic: bool })
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("row", ["1.0,2,3,4", "1e0,2,3,4", "-0,2,3,4", "1_0,2,3,4", " 2,2,3,4", "2 ,2,3,4", "1,1_0,3,4",
                                 "1, 2,3,4", "1,2,3,4abc", "1,2,3,0x10", "1,2,3", "1,2,3,4,maybe", "1,2,3,4,True",
                                 "4294967296,2,3,4", "1,,3,4", "1,e5,3,4", "1,.,3,4"])
def test_rows_the_reference_skips_are_skipped(io, tmp_path, row):
    text = "7,1.5,2.5,3.5\n" + row + "\n8,4.5,5.5,6.5\n"
    assert io._read_numeric_table_native(_write(tmp_path, text), ",") is None          # not the regular form
    got = io.read_contour_data(_write(tmp_path, text, "r.csv"))
    assert np.array_equal(got, np.array([[7.0, 1.5, 2.5, 3.5], [8.0, 4.5, 5.5, 6.5]]))


@pytest.mark.parametrize("row,exp", [("+5,1,2,3", [5.0, 1.0, 2.0, 3.0]), ("5,+1,.5,5.", [5.0, 1.0, 0.5, 5.0]),
                                     ("5,1E2,-2e-1,3", [5.0, 100.0, -0.2, 3.0]), ("5,inf,2,3", [5.0, np.inf, 2.0, 3.0]),
                                     ("5,1,2,3,true", [5.0, 1.0, 2.0, 3.0]), ("5,1,2,3,false", [5.0, 1.0, 2.0, 3.0])])
def test_rows_the_reference_accepts_are_kept(io, tmp_path, row, exp):
    # (the first record has the row's field count: csv::ReaderBuilder is not flexible, see the width test below)
    first = "7,1.5,2.5,3.5,false\n" if row.count(",") == 4 else "7,1.5,2.5,3.5\n"
    got = io.read_contour_data(_write(tmp_path, first + row + "\n"))
    assert got.shape == (2, 4) and np.array_equal(got[1], np.array(exp))


def test_fifth_column_is_the_aortic_flag(io, tmp_path):
    arr, flags = io.read_contour_data(_write(tmp_path, "1,1,2,3,true\n1,2,3,4,false\n1,3,4,5\n"), with_aortic=True)
    assert arr.shape == (2, 4) and flags.tolist() == [True, False]      # the 4-field row is of another width: skipped


def test_reference_point_first_record_must_parse(io, tmp_path):
    with pytest.raises(RuntimeError, match="failed to deserialize first reference-point record"):
        io.read_reference_point(_write(tmp_path, "x,1,2,3\n5,1,2,3\n"))
    with pytest.raises(RuntimeError, match="was empty"):
        io.read_reference_point(_write(tmp_path, "", "e.csv"))
    assert io.read_reference_point(_write(tmp_path, "5,1,2,3\nx,1,2,3\n", "ok.csv")).tolist() == [5.0, 1.0, 2.0, 3.0]


def test_aortic_flags_follow_their_points_through_the_builder(io, tmp_path):
    """Per-point flags of the fifth column stay attached to their points through grouping and
    sort_contour_points (contour.rs:368-405)."""
    import math
    d = tmp_path / "case"
    d.mkdir()
    rows = []
    for f in (3, 4):
        for k in range(12):
            a = 2 * math.pi * k / 12
            x, y = 4.5 + 2 * math.cos(a), 4.5 + 1.5 * math.sin(a)
            rows.append(f"{f},{x!r},{y!r},{0.5 * f},{'true' if x > 4.5 else 'false'}")
    (d / "diastolic_contours.csv").write_text("\n".join(rows) + "\n")
    (d / "diastolic_reference_points.csv").write_text("3,6.5,4.5,1.5\n")
    g = io.build_geometry_from_inputdata(None, str(d), "t", True)
    fl = g.meta["lumen_aortic"]
    assert fl.shape[0] == g.lumen.shape[0] == 24
    assert np.array_equal(fl, g.lumen[:, 0] > 4.5)


def test_rows_of_another_width_than_the_first_record_are_skipped(io, tmp_path):
    """input.rs:172-194: csv::ReaderBuilder with its default flexible(false) -- a record whose field count differs from
    the FIRST record's is Err(UnequalLengths) and skipped ("Skipping invalid row"), even if it would deserialize
    (ADVICE r2 #2).  Expected rows worked out by hand; the independent reader of tests/refbuild.py must agree."""
    import refbuild
    mio = io
    cases = {
        # 4-field file with one valid 5-field row: that row is dropped
        "a.csv": ("1,1.0,2.0,3.0\n1,1.5,2.5,3.0,true\n2,4.0,5.0,6.0\n", [(1, 1.0, 2.0, 3.0, False), (2, 4.0, 5.0, 6.0, False)]),
        # 5-field file with 4-field rows: only the 5-field rows stay
        "b.csv": ("1,1.0,2.0,3.0,true\n1,1.5,2.5,3.0\n2,4.0,5.0,6.0,false\n", [(1, 1.0, 2.0, 3.0, True), (2, 4.0, 5.0, 6.0, False)]),
        # the first record decides even if it does not deserialize itself: three fields -> every 4-field row is skipped
        "c.csv": ("junk,1,2\n1,1.0,2.0,3.0\n2,4.0,5.0,6.0\n", []),
        # empty lines do not count as records
        "d.csv": ("\n1,1.0,2.0,3.0\n\n2,4.0,5.0\n3,7.0,8.0,9.0\n", [(1, 1.0, 2.0, 3.0, False), (3, 7.0, 8.0, 9.0, False)]),
    }
    for name, (text, want) in cases.items():
        p = tmp_path / name
        p.write_text(text)
        arr, flags = mio.read_contour_data(str(p), with_aortic=True)
        got = [(int(r[0]), r[1], r[2], r[3], bool(f)) for r, f in zip(arr, flags)]
        assert got == want, name
        ref = [(q["frame"], q["x"], q["y"], q["z"], q["aortic"]) for q in refbuild.read_contour_data(str(p))]
        assert ref == want, name
