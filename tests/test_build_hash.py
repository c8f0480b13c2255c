"""build.py decides staleness by CONTENT (sha256 of sources, headers, generated asm, flags), never by mtime: the built
library travels to the GPU box in a snapshot that may reorder mtimes.  No compiler is run here: the compile
step is replaced by a stub that writes the library file."""
import importlib.util
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_module():
    spec = importlib.util.spec_from_file_location("_mm_build_t", os.path.join(ROOT, "multimoda-rs_amd", "build.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _tree_copy(b, tmp_path):
    pkg = tmp_path / "pkg"
    (pkg / "csrc").mkdir(parents=True)
    inc = tmp_path / "include"
    inc.mkdir()
    src = b.Tree()
    for f in b.SOURCES + b.HEADERS:
        shutil.copy(os.path.join(src.csrc, f), pkg / "csrc" / f)
    for f in b.PUBLIC_HEADERS:
        shutil.copy(os.path.join(src.include, f), inc / f)
    return b.Tree(str(pkg), str(inc))


def test_header_edit_rebuilds_whatever_the_mtimes_say(tmp_path):
    b = _build_module()
    t = _tree_copy(b, tmp_path)
    calls = []

    def fake_compile(tree, verbose):
        calls.append(tree.digest())
        with open(tree.lib, "wb") as f:
            f.write(b"not a library")

    assert t.stale()
    b.build(tree=t, compile_fn=fake_compile)
    assert b.last_status() == "rebuilt" and len(calls) == 1 and not t.stale()
    b.build(tree=t, compile_fn=fake_compile)
    assert b.last_status() == "reused" and len(calls) == 1
    # every source newer than the library: still reused (mtimes do not count)
    future = os.path.getmtime(t.lib) + 1000
    for p in t.inputs():
        os.utime(p, (future, future))
    b.build(tree=t, compile_fn=fake_compile)
    assert b.last_status() == "reused" and len(calls) == 1
    # a header edit whose mtime is OLDER than the library: rebuilt
    hdr = os.path.join(t.include, "mm_hausdorff.h")
    with open(hdr, "a") as f:
        f.write("\n/* edited */\n")
    os.utime(hdr, (1, 1))
    assert t.stale()
    b.build(tree=t, compile_fn=fake_compile)
    assert b.last_status() == "rebuilt" and len(calls) == 2 and calls[0] != calls[1]
    # the generated asm block counts as an input
    with open(os.path.join(t.csrc, "mm_screen_mx_asm.inc"), "a") as f:
        f.write("// x\n")
    assert t.stale()
    # a library without its stamp (built by an older build.py, or a build that died) is stale
    b.build(tree=t, compile_fn=fake_compile)
    os.remove(t.stamp)
    assert t.stale()


def test_in_tree_library_matches_its_sources():
    """after __graft_entry__.build() the committed tree's stamp equals the digest of what is on disk"""
    import sys
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    ge.build()
    b = _build_module()
    assert not b.Tree().stale()
