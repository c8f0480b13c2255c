"""Build the HIP extension (libmm_hausdorff.so) in-tree for gfx950.

hipcc cross-compiles without a GPU.  The shared library is git-ignored but travels to
the GPU box with the snapshot.  -ffp-contract=off is part of the parity contract: the
f64 kernels and the host orchestration must reproduce the reference's (Rust, never
fused) operation order; the f32 screening kernel asks for its FMAs explicitly.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libmm_hausdorff.so")
SOURCES = ["mm_kernels.hip", "mm_nn_kernels.hip", "mm_engine.cpp", "mm_host.cpp", "mm_centerline.cpp", "mm_ccta.cpp",
           "mm_build.cpp", "mm_frames.cpp", "mm_comm.cpp"]
HEADERS = ["mm_device.h", "mm_engine.h", "mm_pool.h", "mm_sort.h", "mm_trace.h", "mm_screen_mx_asm.inc", os.path.join("..", "..", "include", "mm_hausdorff.h"),
           os.path.join("..", "..", "include", "mm_centerline.h"), os.path.join("..", "..", "include", "mm_ccta.h"),
           os.path.join("..", "..", "include", "mm_build.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    # several ranks may import at once (torch.distributed.run): one builds, the others wait
    import fcntl
    os.makedirs(LIBDIR, exist_ok=True)
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():
                return LIB
            return _build_locked(verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(verbose: bool) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    os.makedirs(LIBDIR, exist_ok=True)
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(LIBDIR, os.path.splitext(s)[0] + ".o")
        cmd = [hipcc, "-x", "hip", *FLAGS, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", LIB, *objs, "-ldl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
