"""The generated main phases of k_screen_mx (csrc/mm_screen_mx_asm.inc, tools/gen_screen_mx.py), executed symbolically.

One asm block per column-tile count (2 .. 17) in three forms (plain; carry = the row minima go through a row store, for
target sets cut into column blocks; emit = carry + a column store, for the first pick of a bounded search) and two schemes
(SINGLE: one MFMA per tile, the row minima cross the lanes through an LDS transpose; DUAL: two MFMAs per tile on swapped
operands, both minima in-lane), the row-tile count a run-time operand: hundreds to ~1400 hand-scheduled instructions each on
fixed registers, a loop, a tail and two epilogues; nothing in them is checked by the compiler.  This test interprets the
committed text with SETS in the registers, one per half of the wave (lanes 0-31 and 32-63 hold different rows of an MFMA
result) -- an MFMA writes the atoms (row tile, column tile, element, half), or their transposed twins for the dual form's
second MFMA, a minimum is a union, a maximum collects finished minima -- for row-tile counts 1 .. 7, 17 and 33, and requires
that at the end
  * the value that leaves the block is the maximum over exactly: the COMPLETE column minimum of each column tile (all row
    tiles, 16 elements, both halves -- nothing missing, nothing folded twice into a different minimum) and, in the plain
    form, every row tile's minimum,
  * single scheme: every row tile's minima went through the reduction scratch exactly once, complete (all column tiles,
    element by element), and were read back in full; dual scheme: a row tile's minimum holds every transposed atom of its
    column tiles from both halves,
  * carry form: every row tile's minimum met the stored minimum of ITS slot of the row store and was written back there,
    once; emit form: the column store holds every column tile's complete minimum,
  * a candidate's column fragments are read once, into the accumulation registers a[4t : 4t + 3] of their tile, and every MFMA
    takes the fragment of its own tile,
  * no register is read while an LDS load into it is outstanding, no MFMA destination is touched within 12 wait states
    of its MFMA (8 passes: 11 required; the assembler pads nothing inside an asm string), no v_permlane32_swap reads a
    register a vector instruction wrote fewer than 2 wait states before -- on every path through the branches.
No GPU, no compiler: pure text."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "multimoda-rs_amd", "csrc", "mm_screen_mx_asm.inc")
MFMA_STATES = 12
E = frozenset()
_TEXT = {}


def _program(name):
    if "text" not in _TEXT:
        _TEXT["text"] = open(INC).read()
    text = _TEXT["text"]
    i = text.index(f"#define MM_SCREEN_MX_ASM_{name} ")
    lines = []
    for ln in text[i:].split("\n")[1:]:
        m = re.match(r'\s*"(.*)\\n" \\', ln)
        if not m:
            break
        lines.append(m.group(1))
    stride = int(re.search(r"MM_SCREEN_MX_RED_STRIDE (\d+)", text).group(1))
    clob = re.search(rf"#define MM_SCREEN_MX_CLOBBERS_{name} (.*)", text).group(1)
    return lines, stride, {int(x) for x in re.findall(r'"v(\d+)"', clob)} | {AG + int(x) for x in re.findall(r'"a(\d+)"', clob)}


AG = 1000          # accumulation registers: a{i} is register AG + i


def _regs(tok):
    tok = tok.strip()
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        base = AG if m.group(1) == "a" else 0
        return list(range(base + int(m.group(2)), base + int(m.group(3)) + 1))
    m = re.fullmatch(r"([va])(\d+)", tok)
    return [(AG if m.group(1) == "a" else 0) + int(m.group(2))] if m else None


class Min:
    """a running minimum: the set of atoms folded into it, per half of the wave"""

    def __init__(self, lo=E, hi=E):
        self.h = (frozenset(lo), frozenset(hi))

    def __or__(self, o):
        assert isinstance(o, Min), o
        return Min(self.h[0] | o.h[0], self.h[1] | o.h[1])


class Max:
    """a running maximum: the collection of finished minima folded into it, per half"""

    def __init__(self, lo=E, hi=E):
        self.h = (frozenset(lo), frozenset(hi))

    @staticmethod
    def of(x):
        return x if isinstance(x, Max) else Max({x.h[0]} if x.h[0] else E, {x.h[1]} if x.h[1] else E)

    def __or__(self, o):
        o = Max.of(o)
        return Max(self.h[0] | o.h[0], self.h[1] | o.h[1])


def _run(nct, carry, nrt, emit=False, dual=False):
    prog, stride, clobbers = _program(f"{'D' if dual else ''}{nct}{'E' if emit else ('C' if carry else '')}")
    colstore = {}                # emit: 256-byte slot of the column store -> Min written
    nloop, tail = (nrt - 1) // 2, (nrt - 1) & 1
    R = {}                       # vector register -> Min / Max / ("A", row tile) / ("B", column tile) / ("addr", bytes) / ("rs", bytes)
    loading, mfma_at, valu_at = set(), {}, {}
    st = dict(states=0, scc=False, counter=None, n_mfma=0, n_valu=0)
    gens = []                    # reduction scratch, one dict {row v: Min} per row tile
    store = {}                   # carry: row store slot (row tile) -> list of Min written
    result = None
    pc = 0

    def find_label(target):
        num, way = target[:-1], target[-1]
        idx = [i for i, ln in enumerate(prog) if ln == num + ":"]
        if way == "b":
            c = [i for i in idx if i <= pc]
            return c[-1]
        c = [i for i in idx if i > pc]
        return c[0]

    def touch(regs, reads, permlane=False):
        for r in regs:
            assert r in clobbers, f"v{r} is not in the clobber list"
            if r in mfma_at:
                assert st["states"] - mfma_at[r] >= MFMA_STATES, f"v{r} touched {st['states'] - mfma_at[r]} states after its MFMA: {prog[pc]}"
                del mfma_at[r]
            if permlane and r in valu_at:
                assert st["states"] - valu_at[r] >= 3, f"v{r} swapped {st['states'] - valu_at[r]} states after a vector write"
        for r in reads:
            assert r not in loading, f"v{r} read under an outstanding LDS load: {prog[pc]}"

    def wrote(regs):
        for r in regs:
            valu_at[r] = st["states"]

    steps = 0
    while pc < len(prog):
        steps += 1
        assert steps < 400000
        ln = prog[pc]
        op, _, rest = ln.partition(" ")
        args = [a.strip() for a in rest.split(",")] if rest else []
        nxt = pc + 1
        if re.fullmatch(r"\d+:", ln):
            pc = nxt
            continue
        if op == "s_nop":
            st["states"] += int(args[0]) + 1
            pc = nxt
            continue
        st["states"] += 1
        if op == "s_mov_b32":
            assert args == ["%1", "%7"]
            st["counter"] = nloop
        elif op == "s_sub_u32":
            assert args[:2] == ["%1", "%1"]
            st["counter"] -= int(args[2])
        elif op == "s_cmp_lg_u32":
            v = {"%1": st["counter"], "%8": tail}[args[0]]
            st["scc"] = v != int(args[1])
        elif op == "s_cmp_eq_u32":
            assert args[0] == "%1"
            st["scc"] = st["counter"] == int(args[1])
        elif op == "s_cbranch_scc1":
            if st["scc"]:
                nxt = find_label(args[0])
        elif op == "s_branch":
            nxt = find_label(args[0])
        elif op == "s_waitcnt":
            assert rest == "lgkmcnt(0)"
            loading.clear()
        elif op == "v_mov_b32":
            d = _regs(args[0])
            touch(d, [])
            src = _regs(args[1])
            if args[1] == "%3":
                R[d[0]] = ("addr", 0)
            elif args[1] == "%9":
                assert carry
                R[d[0]] = ("rs", 0)
            elif src:
                touch(src, src)
                R[d[0]] = R[src[0]]
            else:
                R[d[0]] = Max() if args[1] == "0" else Min()
            wrote(d)
            st["n_valu"] += 1
        elif op == "v_add_u32" and args[2] == "%4":
            d = _regs(args[0])
            touch(d, [])
            assert int(args[1]) == 8 * stride
            R[d[0]] = ("rw", 8)                      # the reduction scratch from row 8 on
            wrote(d)
            st["n_valu"] += 1
        elif op == "v_add_u32":
            d, src = _regs(args[0]), _regs(args[2])
            assert d == src and R[d[0]][0] in ("addr", "rs")
            touch(d, d)
            R[d[0]] = (R[d[0]][0], R[d[0]][1] + int(args[1]))
            wrote(d)
            st["n_valu"] += 1
        elif op in ("v_min3_i32", "v_min_i32"):
            d = _regs(args[0])
            srcs = [r for a in args[1:] for r in _regs(a)]
            touch(d + srcs, srcs)
            v = Min()
            for r in srcs:
                v = v | R[r]
            R[d[0]] = v
            wrote(d)
            st["n_valu"] += 1
        elif op in ("v_max3_i32", "v_max_i32"):
            srcs = [r for a in args[1:] for r in _regs(a)]
            v = Max()
            for r in srcs:
                v = v | R[r]
            if args[0] == "%0":
                touch(srcs, srcs)
                result = v
            else:
                d = _regs(args[0])
                touch(d + srcs, srcs)
                R[d[0]] = v
                wrote(d)
            st["n_valu"] += 1
        elif op == "v_permlane32_swap_b32":
            a, b = _regs(args[0])[0], _regs(args[1])[0]
            touch([a, b], [a, b], permlane=True)
            va, vb = R[a], R[b]
            assert isinstance(va, Min) and isinstance(vb, Min)
            R[a], R[b] = Min(va.h[0], vb.h[0]), Min(va.h[1], vb.h[1])     # lanes 32-63 of vdst <-> lanes 0-31 of src
            wrote([a, b])
            st["n_valu"] += 1
        elif op == "v_mfma_f32_32x32x16_f16":
            d, a, b = _regs(args[0]), _regs(args[1]), _regs(args[2])
            assert args[3] == "0" and len(d) == 16 and len(a) == 4 and len(b) == 4 and d[0] % 2 == 0
            touch(d + a + b, a + b)
            ta, tb = {R[r] for r in a}, {R[r] for r in b}
            assert len(ta) == 1 and len(tb) == 1, f"mixed operand fragments: {ln}"
            (ka, ia), (kb, ib) = next(iter(ta)), next(iter(tb))
            if ka == "A" and kb == "B":           # D = A B^T: a lane's 16 values belong to its column
                rt, ct = ia, ib
                assert 0 <= rt < nrt, (ln, rt)
                for v in range(16):
                    R[d[v]] = Min({(rt, ct, v, 0)}, {(rt, ct, v, 1)})
                    mfma_at[d[v]] = st["states"]
            else:                                  # D' = B A^T (the dual form): a lane's 16 values belong to its row
                assert dual and ka == "B" and kb == "A" and 0 <= ib < nrt, (ln, ia, ib)
                rt, ct = ib, ia
                for v in range(16):
                    R[d[v]] = Min({("T", rt, ct, v, 0)}, {("T", rt, ct, v, 1)})
                    mfma_at[d[v]] = st["states"]
            st["n_mfma"] += 1
        elif op == "ds_write2_b32":
            a0 = args[0]
            d0 = _regs(args[1])
            d1, o0 = args[2].split(" offset0:")
            o0, o1 = o0.split(" offset1:")
            d1, o0, o1 = _regs(d1), int(o0), int(o1)
            touch(d0 + d1, d0 + d1)
            base = 0
            if a0 != "%4":
                areg = _regs(a0)
                touch(areg, areg)
                assert R[areg[0]] == ("rw", 8)
                base = 8
            for src, o in ((d0, o0), (d1, o1)):
                assert (4 * o) % stride == 0 and o < 256
                v = base + 4 * o // stride
                if v == 0:
                    gens.append({})
                assert v == len(gens[-1]), "rows of the reduction scratch are written in order"
                gens[-1][v] = R[src[0]]
        elif op in ("ds_read_b128", "ds_read_b64", "ds_read_b32"):
            d = _regs(args[0])
            addr, off = args[1].split(" offset:")
            off = int(off)
            touch(d, [])
            areg = _regs(addr)
            if addr == "%3" or (areg and R[areg[0]][0] == "addr"):
                base = 0 if addr == "%3" else R[areg[0]][1]
                if areg:
                    touch(areg, areg)
                assert (base + off) % 1024 == 0 and len(d) == 4
                val = [("A", (base + off) // 1024)] * 4          # (a fragment beyond the last row tile may be requested, never used)
            elif areg and R[areg[0]][0] == "rs":
                touch(areg, areg)
                assert carry and len(d) == 1 and (R[areg[0]][1] + off) % 128 == 0
                slot = (R[areg[0]][1] + off) // 128
                assert slot not in store, "the stored minimum is read after this block wrote the slot"
                val = [Min({("PREV", slot)}, {("PREV", slot)})]
            elif addr == "%2":
                assert off % 512 == 0 and len(d) == 2 and off // 512 < nct
                assert d[0] - AG in (4 * (off // 512), 4 * (off // 512) + 2), "column tile t lives in a[4t : 4t + 3]"
                val = [("B", off // 512)] * 2
            elif addr == "%5":
                assert off % 16 == 0 and len(d) == 4 and d[0] % 2 == 0
                g = len(gens) - 1
                assert len(gens[g]) == 16, "reduction scratch read before all 16 rows are written"
                val = [Min({("R", g, off // 4 + i, 0)}, {("R", g, off // 4 + i, 1)}) for i in range(4)]
            else:
                raise AssertionError(ln)
            for r, x in zip(d, val):
                R[r] = x
                loading.add(r)
        elif op == "ds_write_b32":
            a0 = args[0]
            src, off = args[1].split(" offset:")
            src, off = _regs(src), int(off)
            touch(src, src)
            if a0 == "%10":
                assert emit and off % 256 == 0 and off // 256 not in colstore
                colstore[off // 256] = R[src[0]]
            elif a0 == "%4":
                assert off % stride == 0
                v = off // stride
                if v == 0:
                    gens.append({})
                assert v == len(gens[-1]), "rows of the reduction scratch are written in order"
                gens[-1][v] = R[src[0]]
            else:
                areg = _regs(a0)
                touch(areg, areg)
                assert carry and R[areg[0]][0] == "rs" and (R[areg[0]][1] + off) % 128 == 0
                store.setdefault((R[areg[0]][1] + off) // 128, []).append(R[src[0]])
        elif op == "ds_bpermute_b32":
            d, src = _regs(args[0]), _regs(args[2])
            assert args[1] == "%6"
            touch(d + src, src)
            R[d[0]] = Min(R[src[0]].h[1], R[src[0]].h[0])        # lane ^ 32
            loading.add(d[0])
        else:
            raise AssertionError("unknown instruction: " + ln)
        pc = nxt
    assert not loading and result is not None
    return dict(gens=gens, result=result, store=store, colstore=colstore, **st)


def _check(nct, carry, nrt, emit=False):
    r = _run(nct, carry, nrt, emit)
    assert r["n_mfma"] == nrt * nct
    # row minima: every row tile through the scratch in order, each complete, element by element, half by half
    assert len(r["gens"]) == nrt
    for rt, g in enumerate(r["gens"]):
        for v in range(16):
            for h in range(2):
                assert g[v].h[h] == {(rt, ct, v, h) for ct in range(nct)}, f"row tile {rt} element {v} half {h}"
    want_rows = [frozenset(("R", g, i, h) for i in range(16) for h in range(2)) for g in range(nrt)]
    want_cols = [frozenset((rt, ct, v, h) for rt in range(nrt) for v in range(16) for h in range(2)) for ct in range(nct)]
    lo, hi = r["result"].h
    for w in want_cols:
        assert w in lo or w in hi, "a column tile's minimum is incomplete"
    if not carry:
        # what leaves the block, in either half: the complete column minimum of every column tile in one of the halves,
        # and every row tile's reduction (all 16 dwords of the row, met with the other half's)
        for h in (lo, hi):
            for w in want_rows:
                assert w in h, "a row tile's reduction is missing from one half"
        assert (lo | hi) == set(want_rows) | set(want_cols), "something else was folded into the maximum"
    else:
        assert (lo | hi) == set(want_cols), "the carry form returns the column minima alone"
        assert sorted(r["store"]) == list(range(nrt))
        for rt, w in r["store"].items():
            assert len(w) == 1, "a slot of the row store is written once per block"
            for h in range(2):
                assert w[0].h[h] == want_rows[rt] | {("PREV", rt)}, f"row store slot {rt}"
    if emit:
        # the column store: lane l of slot p holds column 64 p + l -- the lower half column tile 2p, the upper half 2p + 1;
        # an odd last tile in both halves -- each the COMPLETE minimum of its column tile
        assert sorted(r["colstore"]) == list(range((nct + 1) // 2))
        for p_, w in r["colstore"].items():
            assert w.h[0] == want_cols[2 * p_]
            assert w.h[1] == want_cols[2 * p_ + 1 if 2 * p_ + 1 < nct else 2 * p_]
    return r


def _check_dual(nct, carry, nrt, emit=False):
    """The dual form: two MFMAs per tile (D and D'), the column minima from D as in the single form, a row tile's minimum
    from its D' tiles -- every column tile, all 16 elements, both halves (the halves hold different columns of the same
    row), folded once into the running maximum (plain) or met with the stored minimum of its slot and written back (carry)."""
    r = _run(nct, carry, nrt, emit, dual=True)
    assert r["n_mfma"] == 2 * nrt * nct and not r["gens"]
    want_cols = [frozenset((rt, ct, v, h) for rt in range(nrt) for v in range(16) for h in range(2)) for ct in range(nct)]
    want_rows = [frozenset(("T", rt, ct, v, h) for ct in range(nct) for v in range(16) for h in range(2)) for rt in range(nrt)]
    lo, hi = r["result"].h
    for w in want_cols:
        assert w in lo or w in hi, "a column tile's minimum is incomplete"
    if not carry:
        for h in (lo, hi):
            for w in want_rows:
                assert w in h, "a row tile's minimum is missing from one half"
        assert (lo | hi) == set(want_rows) | set(want_cols), "something else was folded into the maximum"
    else:
        assert (lo | hi) == set(want_cols), "the carry form returns the column minima alone"
        assert sorted(r["store"]) == list(range(nrt))
        for rt, w in r["store"].items():
            assert len(w) == 1, "a slot of the row store is written once per block"
            for h in range(2):
                assert w[0].h[h] == want_rows[rt] | {("PREV", rt)}, f"row store slot {rt}"
    if emit:
        assert sorted(r["colstore"]) == list(range((nct + 1) // 2))
        for p_, w in r["colstore"].items():
            assert w.h[0] == want_cols[2 * p_]
            assert w.h[1] == want_cols[2 * p_ + 1 if 2 * p_ + 1 < nct else 2 * p_]
    # 16 minima per tile, five instructions per row tile (+ the loop's address updates), the column fold and the set-up:
    # nothing else on the vector pipe
    assert r["n_valu"] <= 16 * nrt * nct + 7 * nrt + 5 * nct + 12, (r["n_valu"], nrt, nct)
    return r


NRTS = [1, 2, 3, 4, 5, 6, 7, 17, 33]


@pytest.mark.parametrize("form", ["plain", "carry", "emit"])
@pytest.mark.parametrize("nct", range(2, 18))
def test_dual_form_every_tile_once_into_the_right_minima(nct, form):
    for nrt in NRTS:
        _check_dual(nct, form != "plain", nrt, emit=form == "emit")


@pytest.mark.parametrize("nct", range(2, 18))
def test_emit_form_leaves_row_and_column_minima(nct):
    """carry + column store (the pick of a bounded search): the row store holds every row's minimum, the column store every
    column's, and the value that leaves is still the maximum over the column minima"""
    for nrt in NRTS:
        _check(nct, True, nrt, emit=True)


@pytest.mark.parametrize("carry", [False, True])
@pytest.mark.parametrize("nct", range(2, 18))
def test_every_tile_once_into_the_right_minima(nct, carry):
    for nrt in NRTS:
        _check(nct, carry, nrt)


def test_vector_instruction_count_is_near_the_floor():
    """two values per three-operand minimum, every value used twice (row and column): 16 per tile is the floor; the bench
    shape (17 x 17) stays within 16.5, an odd column-tile count within 17.5 from 7 tiles on, an even one (one tile of each
    row tile folded alone: 24 instead of 16) within 18.5"""
    r = _run(17, False, 17)
    assert r["n_valu"] <= 16.5 * r["n_mfma"], (r["n_valu"], r["n_mfma"])
    for nct in range(7, 18):
        r = _run(nct, False, nct)
        assert r["n_valu"] <= (17.5 if nct & 1 else 18.5) * r["n_mfma"], (nct, r["n_valu"], r["n_mfma"])


def _run_bound(nb, nt):
    """the bound kernel's pass (MM_BOUND_MX_ASM_<nb>): nb column tiles of queries (the last nb operands) against nt row tiles"""
    text = open(INC).read()
    i = text.index(f"#define MM_BOUND_MX_ASM_{nb} ")
    prog = []
    for ln in text[i:].split("\n")[1:]:
        m = re.match(r'\s*"(.*)\\n" \\', ln)
        if not m:
            break
        prog.append(m.group(1))
    clobbers = {int(x) for x in re.findall(r'"v(\d+)"', re.search(rf"#define MM_BOUND_MX_CLOBBERS_{nb} (.*)", text).group(1))}
    o_cnt, o_addr, o_nt, o_b = f"%{nb}", f"%{nb + 1}", f"%{nb + 2}", nb + 3
    R, loading, mfma_at = {}, set(), {}
    st = dict(states=0, scc=False, counter=None, n_mfma=0, n_valu=0, n_lds=0)
    result, pc, steps = {}, 0, 0

    def label(target):
        num, way = target[:-1], target[-1]
        idx = [k for k, ln in enumerate(prog) if ln == num + ":"]
        return [k for k in idx if k <= pc][-1] if way == "b" else [k for k in idx if k > pc][0]

    def touch(regs, reads):
        for r in regs:
            assert r in clobbers, f"v{r} is not in the clobber list"
            if r in mfma_at:
                assert st["states"] - mfma_at[r] >= MFMA_STATES, f"v{r} touched {st['states'] - mfma_at[r]} states after its MFMA: {prog[pc]}"
                del mfma_at[r]
        for r in reads:
            assert r not in loading, f"v{r} read under an outstanding LDS load: {prog[pc]}"

    while pc < len(prog):
        steps += 1
        assert steps < 100000
        ln = prog[pc]
        op, _, rest = ln.partition(" ")
        args = [a.strip() for a in rest.split(",")] if rest else []
        nxt = pc + 1
        if re.fullmatch(r"\d+:", ln):
            pc = nxt
            continue
        if op == "s_nop":
            st["states"] += int(args[0]) + 1
            pc = nxt
            continue
        st["states"] += 1
        if op == "s_mov_b32":
            assert args == [o_cnt, o_nt]
            st["counter"] = nt - 1
        elif op == "s_sub_u32":
            st["counter"] -= int(args[2])
            assert st["counter"] >= 0
        elif op == "s_cmp_lt_u32":
            st["scc"] = st["counter"] < int(args[1])
        elif op == "s_cmp_gt_u32":
            st["scc"] = st["counter"] > int(args[1])
        elif op == "s_cmp_eq_u32":
            st["scc"] = st["counter"] == int(args[1])
        elif op == "s_cbranch_scc1":
            if st["scc"]:
                nxt = label(args[0])
        elif op == "s_branch":
            nxt = label(args[0])
        elif op == "s_waitcnt":
            loading.clear()
        elif op == "v_mov_b32":
            d = _regs(args[0])
            touch(d, [])
            R[d[0]] = ("addr", 0) if args[1] == o_addr else Min()
        elif op == "v_add_u32":
            d = _regs(args[0])
            touch(d, d)
            R[d[0]] = ("addr", R[d[0]][1] + int(args[1]))
        elif op == "ds_read_b128":
            d = _regs(args[0])
            addr, off = args[1].split(" offset:")
            a = _regs(addr)
            touch(d + a, a)
            t = (R[a[0]][1] + int(off)) // 1024
            # (requests may run up to two tiles past the set: read, never multiplied -- the MFMA branch checks that)
            assert (R[a[0]][1] + int(off)) % 1024 == 0 and 0 <= t < max(nt + 2, 3), f"row tile {t} of {nt} requested"
            for r in d:
                R[r] = ("A", t)
                loading.add(r)
            st["n_lds"] += 1
        elif op == "v_mfma_f32_32x32x16_f16":
            d, a = _regs(args[0]), _regs(args[1])
            j = int(args[2][1:]) - o_b
            assert args[2][0] == "%" and 0 <= j < nb and args[3] == "0" and len(d) == 16 and len(a) == 4
            touch(d + a, a)
            ta = {R[r] for r in a}
            assert len(ta) == 1
            (_, t), = ta
            assert 0 <= t < nt, f"row tile {t} of {nt} multiplied"
            for v in range(16):
                R[d[v]] = Min({(j, t, v, 0)}, {(j, t, v, 1)})
                mfma_at[d[v]] = st["states"]
            st["n_mfma"] += 1
        elif op in ("v_min3_i32", "v_min_i32"):
            srcs = [r for a in args[1:] for r in _regs(a)]
            v = Min()
            for r in srcs:
                v = v | R[r]
            if args[0].startswith("%"):
                touch(srcs, srcs)
                result[int(args[0][1:])] = v
            else:
                d = _regs(args[0])
                touch(d + srcs, srcs)
                R[d[0]] = v
            st["n_valu"] += 1
        else:
            raise AssertionError("unknown instruction: " + ln)
        pc = nxt
    assert not loading and sorted(result) == list(range(nb))
    return result, st


@pytest.mark.parametrize("nb", [1, 2, 4])
@pytest.mark.parametrize("nt", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 16, 17, 18, 19, 32])
def test_bound_pass_folds_every_tile_once(nb, nt):
    result, st = _run_bound(nb, nt)
    assert st["n_mfma"] == nt * nb and nt <= st["n_lds"] <= max(nt + 2, 3)     # one read of a row fragment feeds nb MFMAs
    for j in range(nb):
        for h in range(2):
            assert result[j].h[h] == {(j, t, v, h) for t in range(nt) for v in range(16)}
    assert st["n_valu"] <= nb * (8 * nt + 4) + 2
