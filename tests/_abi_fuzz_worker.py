"""Worker of tests/test_abi_null_args.py: calls every function the headers declare with NULL pointers and small / zero
scalars (ones, zeros, NaN, huge / negative values), one after the other from index `start`, printing the name BEFORE each call -- so that the parent sees which
call took the process down, if one does.  mode "engine": a live engine handle goes into every `mm_engine*` first
argument, so the validation BEHIND the engine check is what runs.  A bad argument must come back as a negative status
(or a harmless value), never as a fault."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import multimoda_rs_amd as mm  # noqa: E402

N = mm._native
SKIP = {"mm_device_count", "mm_last_error", "mm_version", "mm_engine_create", "mm_engine_destroy"}


def functions():
    names = sorted(set(N.EXPORTS) | set(N.EXPORTS_CENTERLINE) | set(N.EXPORTS_CCTA) | set(N.EXPORTS_BUILD))
    return [n for n in names if n not in SKIP]


def engine_first():
    """Functions whose first parameter is an `mm_engine*` (from the headers)."""
    import re
    out = set()
    for h in os.listdir(os.path.join(ROOT, "include")):
        txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", h)).read(), flags=re.S)
        out |= set(re.findall(r"\b(mm_[a-z0-9_]+)\s*\(\s*mm_engine\s*\*", txt))
    return out


def main():
    start, mode = int(sys.argv[1]), sys.argv[2]
    L = N.lib()
    eng = mm.Engine() if mode == "engine" else None
    names = functions()
    with_engine = engine_first() if eng is not None else set()
    VARIANTS = [(1, 1.0), (0, 0.0), (1, float("nan")), (-1, 1e300), (1, -1e-300), (3, 1e-300), (2, float("inf"))]     # (integers, floats)
    for k in range(start, len(VARIANTS) * len(names)):
        name, variant = names[k % len(names)], k // len(names)
        f = getattr(L, name)
        args = []
        for j, t in enumerate(f.argtypes):
            is_ptr = t in (C.c_void_p, C.c_char_p) or hasattr(t, "contents")
            if is_ptr:
                args.append(eng.handle if (j == 0 and name in with_engine) else None)
            elif t is C.c_char:
                args.append(b",")
            elif t in (C.c_double, C.c_float):
                args.append(VARIANTS[variant][1])
            elif t in (C.c_uint32, C.c_uint64, C.c_uint8, C.c_size_t):
                args.append(abs(VARIANTS[variant][0]))
            else:
                args.append(VARIANTS[variant][0])
        print(f"CALL {k} {name}", flush=True)
        f(*args)
    print("DONE", flush=True)


if __name__ == "__main__":
    main()
