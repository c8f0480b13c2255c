"""Build the HIP extension (libmm_hausdorff.so) in-tree for gfx950.

hipcc cross-compiles without a GPU.  The shared library is git-ignored but travels to
the GPU box with the snapshot.  -ffp-contract=off is part of the parity contract: the
f64 kernels and the host orchestration must reproduce the reference's (Rust, never
fused) operation order; the f32 screening kernel asks for its FMAs explicitly.

Staleness is decided by CONTENT, not by mtime: a sha256 over every source, header, the
generated asm blocks, the flags and this file is stored beside the library
(`libmm_hausdorff.so.sha256`); a copy of the tree that reorders mtimes (the snapshot that
travels to the GPU box) can neither hide an edit nor force a rebuild.  `build()` returns the
library path; `last_status()` says whether that call `rebuilt` or `reused` it.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["mm_kernels.hip", "mm_nn_kernels.hip", "mm_engine.cpp", "mm_host.cpp", "mm_centerline.cpp", "mm_ccta.cpp",
           "mm_build.cpp", "mm_frames.cpp", "mm_comm.cpp"]
HEADERS = ["mm_device.h", "mm_engine.h", "mm_pool.h", "mm_sort.h", "mm_trace.h", "mm_screen_mx_asm.inc"]
PUBLIC_HEADERS = ["mm_hausdorff.h", "mm_centerline.h", "mm_ccta.h", "mm_build.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]
LINK_FLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-pthread"]

_status = None


class Tree:
    """The files of one checkout: `pkg` is the package directory (csrc/, lib/), `include` the public headers."""

    def __init__(self, pkg: str = HERE, include: str | None = None):
        self.pkg = pkg
        self.csrc = os.path.join(pkg, "csrc")
        self.include = include or os.path.join(pkg, "..", "include")
        self.libdir = os.path.join(pkg, "lib")
        self.lib = os.path.join(self.libdir, "libmm_hausdorff.so")
        self.stamp = self.lib + ".sha256"

    def inputs(self):
        return ([os.path.join(self.csrc, s) for s in SOURCES + HEADERS] +
                [os.path.join(self.include, h) for h in PUBLIC_HEADERS])

    def digest(self) -> str:
        h = hashlib.sha256()
        h.update(("\0".join(FLAGS + ["|"] + LINK_FLAGS)).encode())
        with open(os.path.abspath(__file__), "rb") as f:
            h.update(hashlib.sha256(f.read()).digest())
        for path in self.inputs():
            h.update(os.path.basename(path).encode() + b"\0")
            with open(path, "rb") as f:
                h.update(hashlib.sha256(f.read()).digest())
        return h.hexdigest()

    def stale(self) -> bool:
        if not os.path.exists(self.lib) or not os.path.exists(self.stamp):
            return True
        with open(self.stamp) as f:
            return f.read().strip() != self.digest()


LIB = Tree().lib


def last_status():
    """'rebuilt' or 'reused' for the last build() of this process (None before the first)."""
    return _status


def build(force: bool = False, verbose: bool = False, tree: Tree | None = None, compile_fn=None) -> str:
    global _status
    tree = tree or Tree()
    if not force and not tree.stale():
        _status = "reused"
        return tree.lib
    # several ranks may import at once (torch.distributed.run): one builds, the others wait
    import fcntl
    os.makedirs(tree.libdir, exist_ok=True)
    with open(os.path.join(tree.libdir, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not tree.stale():
                _status = "reused"
                return tree.lib
            digest = tree.digest()
            if os.path.exists(tree.stamp):
                os.remove(tree.stamp)           # a build that dies half way leaves no stamp behind
            (compile_fn or _compile)(tree, verbose)
            with open(tree.stamp, "w") as f:
                f.write(digest + "\n")
            _status = "rebuilt"
            if verbose:
                print(f"libmm_hausdorff.so rebuilt ({digest[:16]})", file=sys.stderr)
            return tree.lib
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _compile(tree: Tree, verbose: bool) -> None:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    objs = []
    procs = []
    for s in SOURCES:
        src = os.path.join(tree.csrc, s)
        obj = os.path.join(tree.libdir, os.path.splitext(s)[0] + ".o")
        cmd = [hipcc, "-x", "hip", *FLAGS, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
        while sum(p.poll() is None for _, p in procs) >= 4:      # four translation units at a time
            for _, p in procs:
                if p.poll() is None:
                    p.wait()
                    break
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = [hipcc, *LINK_FLAGS, "-o", tree.lib, *objs, "-ldl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True), last_status())
