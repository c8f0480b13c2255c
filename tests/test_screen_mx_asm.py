"""The generated main phase of k_screen_mx (csrc/mm_screen_mx_asm.inc, tools/gen_screen_mx.py), executed symbolically.

The asm block is ~1900 hand-scheduled instructions on fixed registers; nothing in it is checked by the compiler.  This test
interprets the committed text for each of its four variants with SETS in the registers -- an MFMA writes the atoms
(row tile, column tile, element), a minimum is a union -- and requires that, at the end,
  * every column-minimum accumulator holds exactly the 16 elements of every tile of its column that the wave owns,
  * every row tile's minima went through the reduction scratch exactly once, complete (all its column tiles, element
    by element), were read back in full and folded into the row maximum (or, for the shared 17th row tile, ds_min'd),
  * no register is read while an LDS load into it is outstanding, and no MFMA destination is touched within 12 wait
    states of its MFMA (8 passes: 11 required; the assembler pads nothing inside an asm string).
No GPU, no compiler: pure text."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "multimoda-rs_amd", "csrc", "mm_screen_mx_asm.inc")
PART = [(0, 5), (5, 9), (9, 13), (13, 17)]
NCT = 17
MFMA_STATES = 12


def _program():
    lines = []
    with open(INC) as f:
        for ln in f:
            m = re.match(r'\s*"(.*)\\n" \\', ln)
            if m:
                lines.append(m.group(1))
    stride = int(re.search(r"MM_SCREEN_MX_RED_STRIDE (\d+)", open(INC).read()).group(1))
    return lines, stride


def _regs(tok):
    tok = tok.strip()
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return [int(m.group(1))]
    return None


def _run(variant):
    prog, stride = _program()
    labels = {m.group(1): i for i, ln in enumerate(prog) if (m := re.fullmatch(r"(\d+):", ln))}
    R = {}                       # register -> frozenset of atoms
    loading = set()
    mfma_at = {}
    states = 0
    scratch, gens = {}, []       # current scratch rows; per generation: {v: atoms}
    colmin, shared16, result = {}, [], None
    scc = False
    n_mfma = n_valu = 0

    def touch(regs, reads):
        for r in regs:
            if r in mfma_at:
                assert states - mfma_at[r] >= MFMA_STATES, f"v{r} touched {states - mfma_at[r]} states after its MFMA: {prog[pc]}"
                del mfma_at[r]
        for r in reads:
            assert r not in loading, f"v{r} read under an outstanding LDS load: {prog[pc]}"

    pc = 0
    while pc < len(prog):
        ln = prog[pc]
        op, _, rest = ln.partition(" ")
        args = [a.strip() for a in rest.split(",")] if rest else []
        nxt = pc + 1
        if re.fullmatch(r"\d+:", ln):
            pc = nxt
            continue
        if op == "s_nop":
            states += int(args[0]) + 1
            pc = nxt
            continue
        states += 1
        if op == "s_cmp_eq_u32":
            assert args[0] == "%9"
            scc = variant == int(args[1])
        elif op == "s_cbranch_scc0":
            if not scc:
                nxt = labels[args[0].rstrip("f")]
        elif op == "s_branch":
            nxt = labels[args[0].rstrip("f")]
        elif op == "s_waitcnt":
            assert rest == "lgkmcnt(0)"
            loading.clear()
        elif op == "v_mov_b32":
            if args[0] == "%0":
                src = _regs(args[1])
                touch(src, src)
                result = R[src[0]]
            else:
                d = _regs(args[0])
                touch(d, [])
                R[d[0]] = frozenset()
                n_valu += 1
        elif op in ("v_min3_i32", "v_min_i32", "v_max_i32"):
            d = _regs(args[0])
            srcs = [r for a in args[1:] for r in _regs(a)]
            touch(d + srcs, srcs)
            R[d[0]] = frozenset().union(*[R[r] for r in srcs])
            n_valu += 1
        elif op == "v_mfma_f32_32x32x16_f16":
            d, a, b = _regs(args[0]), _regs(args[1]), _regs(args[2])
            assert args[3] == "0" and len(d) == 16 and len(a) == 4 and len(b) == 4 and d[0] % 2 == 0
            touch(d + a + b, a + b)
            ta, tb = {R[r] for r in a}, {R[r] for r in b}
            assert len(ta) == 1 and len(tb) == 1, f"mixed operand fragments: {ln}"
            (ka, rt), = next(iter(ta))
            (kb, ct), = next(iter(tb))
            assert ka == "A" and kb == "B"
            for v in range(16):
                R[d[v]] = frozenset({(rt, ct, v)})
                mfma_at[d[v]] = states
            n_mfma += 1
        elif op in ("ds_read_b128", "ds_read_b64"):
            d = _regs(args[0])
            addr, off = args[1].split(" offset:")
            off = int(off)
            touch(d, [])
            if addr == "%3":
                val = [frozenset({("A", 16)})] * 4
            elif addr == "%2":
                assert off % 4096 == 0
                val = [frozenset({("A", "k%d" % (off // 4096))})] * 4
            elif addr == "%1":
                assert off % 512 == 0 and len(d) == 2
                val = [frozenset({("B", off // 512)})] * 2
            elif addr == "%5":
                assert off % 8 == 0 and len(d) == 2
                g = len(gens) - 1
                assert len(gens[g]) == 16, "reduction scratch read before all 16 rows are written"
                val = [frozenset({("R", g, off // 4)}), frozenset({("R", g, off // 4 + 1)})]
            else:
                raise AssertionError(ln)
            for r, x in zip(d, val):
                R[r] = x
                loading.add(r)
        elif op == "ds_write_b32":
            assert args[0] == "%4"
            src, off = args[1].split(" offset:")
            src, off = _regs(src), int(off)
            touch(src, src)
            assert off % stride == 0
            v = off // stride
            if v == 0:
                gens.append({})
            assert v == len(gens[-1]), "rows of the reduction scratch are written in order"
            gens[-1][v] = R[src[0]]
        elif op == "ds_bpermute_b32":
            d, src = _regs(args[0]), _regs(args[2])
            assert args[1] == "%7"
            touch(d + src, src)
            R[d[0]] = frozenset(("P", a) for a in R[src[0]])
            loading.add(d[0])
        elif op == "ds_min_i32":
            if args[0] == "%8":
                src = _regs(args[1])
                touch(src, src)
                shared16.append(R[src[0]])
            else:
                assert args[0] == "%6"
                src, off = args[1].split(" offset:")
                src, off = _regs(src), int(off)
                touch(src, src)
                assert off % 128 == 0 and off // 128 not in colmin
                colmin[off // 128] = R[src[0]]
        else:
            raise AssertionError("unknown instruction: " + ln)
        pc = nxt
    assert not loading and result is not None
    return dict(colmin=colmin, gens=gens, shared16=shared16, result=result, n_mfma=n_mfma, n_valu=n_valu)


@pytest.mark.parametrize("variant", [0, 1, 2, 3])
def test_every_tile_once_into_the_right_minima(variant):
    r = _run(variant)
    full = ["k0", "k1", "k2", "k3"]
    part = list(range(*PART[variant]))
    assert r["n_mfma"] == 4 * NCT + len(part)
    # column minima
    assert sorted(r["colmin"]) == list(range(NCT))
    for ct in range(NCT):
        want = {(rt, ct, v) for rt in full for v in range(16)}
        if ct in part:
            want |= {(16, ct, v) for v in range(16)}
        assert r["colmin"][ct] == want, f"column tile {ct}"
    # row minima: five row tiles through the scratch, each complete, the shared one first
    assert len(r["gens"]) == 5
    order = [16] + full
    for g, rt in zip(r["gens"], order):
        tiles = part if rt == 16 else range(NCT)
        for v in range(16):
            assert g[v] == {(rt, ct, v) for ct in tiles}, f"row tile {rt} element {v}"

    def folded(g):                       # what a lane holds after reading its row back and meeting the other half
        own = {("R", g, i) for i in range(16)}
        return own | {("P", a) for a in own}

    assert r["shared16"] == [folded(0)]
    assert r["result"] == set().union(*[folded(g) for g in range(1, 5)])


def test_vector_instruction_count_is_near_the_floor():
    """two values per three-operand minimum, every value used twice (row and column): 16 per tile is the floor"""
    for variant in range(4):
        r = _run(variant)
        tiles = r["n_mfma"]
        assert r["n_valu"] <= 16.6 * tiles, (variant, r["n_valu"], tiles)
