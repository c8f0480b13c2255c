#!/usr/bin/env python3
"""Generates multimoda-rs_amd/csrc/mm_screen_mx_asm.inc: the main phase of k_screen_mx (mm_kernels.hip) as ONE asm block on
fixed registers, so that the order is exactly the software pipeline we want -- the MFMAs of one group of tiles (1024
squared distances each, on the matrix pipe) issued ahead of the v_min3_i32 that fold the PREVIOUS group on the vector pipe.
The compiler's scheduler does not produce this order (tools/ubench_mfma16*.hip: 46 ns per tile compiler-scheduled, 34 ns
hand-ordered).

One wave, one candidate: its share of the 17th row tile (4 or 5 column tiles, chosen by the scalar operand `variant`) and
its four full row tiles (17 column tiles each), so that the four waves of a workgroup carry 72 or 73 tiles each -- ONE
stream of groups, pipelined across the row-tile boundaries as well.

A row tile is an `init` group (one column tile) followed by pairs of column tiles.  There are five 16-register result
buffers and no separate row-minimum registers: the buffer the init tile's MFMA writes simply BECOMES the running row
minima of that row tile (no instruction), and every pair folds into it with 16 three-operand minima.  Per full row tile:
8 (column fold of the init tile) + 8 x 32 = 264 vector instructions for 17 tiles; the floor of two values per
instruction, each value used twice, is 272 -- the init tile's row half costs nothing.

The cross-lane reduction of a row tile's minima (LDS transpose: write, read back a row per lane, fold, meet the other
half through ds_bpermute) is spread over the steps of the NEXT row tile, one LDS round trip per step, each behind a wait
the pipeline has anyway; its first stage (the write) frees the buffer for the pair after next.

Wait states: nothing in an asm string is padded by the assembler.  The generator tracks every MFMA's destination and
pads (s_nop) where fewer than MFMA_STATES instructions separate it from the first instruction that touches the buffer; it
also refuses to read a register an LDS load is still filling.

Register map (VGPR):
  v[20:35], v[100:163]  five result buffers (a row tile's running row minima live in one of them)
  v36        rowmax     max over rows of the row minima (signed-int order on f32 bits, floored at 0)
  v[40:56]   cm[17]     running column minima per column tile (this lane's column, this wave's rows)
  v[60:63], v[64:67]    A operand fragment of the current / the next row tile
  v[68:71], v72         reduction: four values read back, accumulator
  v[80:87], v[88:95]    B operand fragments, two groups (double buffer)
Operands: %0 out rowmax; %1 vB (LDS byte address of this lane's B fragment in column tile 0); %2 vA (A fragment of the
wave's first row tile; the next ones 4096 bytes apart); %3 vA16 (A fragment of row tile 16); %4 vRW / %5 vRR
(row-reduction scratch: write / read address); %6 vCM (column-minimum array, this lane's column of tile 0); %7 vPERM
(4 * (lane ^ 32)); %8 vR16 (row minima of row tile 16, shared by the four waves: this lane's row); %9 s variant (0..3)."""
import os

INF = "0x7f800000"
BUF = [20, 100, 116, 132, 148]
ROWMAX, CM, ASET, T, ACC, BSET = 36, 40, [60, 64], 68, 72, [80, 88]
NCT = 17
RED_STRIDE = 136          # bytes between rows of the reduction scratch (34 dwords: 8-byte aligned reads, 2-way conflicts)
PART = [(0, 5), (5, 9), (9, 13), (13, 17)]          # column tiles of row tile 16 per variant
MFMA_STATES = 12          # instructions between an 8-pass MFMA and the first touch of its destination (11 required)

DBG_NOP = int(os.environ.get("MX_DBG_NOP", "0"))           # s_nop 15 count in front of every group of minima


class Stream:
    """Straight-line instruction list with the two checks described above."""

    def __init__(self, inherit=None):
        self.out = []
        self.n = 0                              # wait states issued so far
        self.mfma_at = {}                       # register -> state count at which an MFMA writing it was issued
        self.loading = set()                    # registers an LDS load is filling (cleared by s_waitcnt)
        if inherit is not None:
            self.n = inherit.n
            self.mfma_at = dict(inherit.mfma_at)
            self.loading = set(inherit.loading)

    def _touch(self, regs, reads):
        need = 0
        for r in regs:
            if r in self.mfma_at:
                need = max(need, MFMA_STATES - (self.n - self.mfma_at[r]))
        if need > 0:
            self.out.append(f"s_nop {need - 1}")
            self.n += need
        for r in regs:
            self.mfma_at.pop(r, None)
        for r in reads:
            assert r not in self.loading, f"v{r} read while an LDS load is in flight"

    def ins(self, text, reads=(), writes=(), lds_load=False, mfma=False):
        reads, writes = list(reads), list(writes)
        self._touch(reads + writes, reads)
        for r in writes:
            assert r not in self.loading or lds_load, f"v{r} overwritten while an LDS load is in flight"
        self.out.append(text)
        self.n += 1
        if mfma:
            for r in writes:
                self.mfma_at[r] = self.n
        if lds_load:
            self.loading.update(writes)

    def wait(self):
        self.out.append("s_waitcnt lgkmcnt(0)")
        self.n += 1
        self.loading.clear()

    def raw(self, text):                       # scalar / control instructions
        self.out.append(text)
        self.n += 1

    def quiet(self):
        """no MFMA result younger than MFMA_STATES (asserted where two code paths meet)"""
        return all(self.n - at >= MFMA_STATES for at in self.mfma_at.values())


def rng(base, n):
    return list(range(base, base + n))


class Pipe:
    """The software pipeline: step(g) issues the MFMAs of group g and, beside them, the minima of the group before."""

    def __init__(self, s, free, bpar=0):
        self.s = s
        self.free = list(free)              # free result buffers
        self.pending = None                 # group whose minima are still to be issued
        self.rmin = None                    # buffer holding the running row minima of the current row tile
        self.bpar = bpar                    # B operand set of the next group
        self.red = None                     # (buffer, shared, next stage) of the reduction in progress
        self.next_b = None                  # tiles whose B fragments are in flight / loaded for the next group

    # ---- operands -------------------------------------------------------------------------------------------------
    def load_b(self, tiles):
        # a column fragment is stored as 8 bytes per lane -- (x1, x2, y1, y2) or (256, 1, n2h, n2l) -- and read twice: the K
        # slots 4..7 repeat slots 0..3 (the row fragments are laid out to match; for the norm half their slots 4..7 are 0)
        base = BSET[self.bpar]
        for i, t in enumerate(tiles):
            d = base + 4 * i
            self.s.ins(f"ds_read_b64 v[{d}:{d + 1}], %1 offset:{t * 512}", writes=rng(d, 2), lds_load=True)
            self.s.ins(f"ds_read_b64 v[{d + 2}:{d + 3}], %1 offset:{t * 512}", writes=rng(d + 2, 2), lds_load=True)
        self.next_b = list(tiles)

    def load_a(self, reg, operand, offset):
        self.s.ins(f"ds_read_b128 v[{reg}:{reg + 3}], {operand} offset:{offset}", writes=rng(reg, 4), lds_load=True)

    def mfma(self, d, a, b):
        self.s.ins(f"v_mfma_f32_32x32x16_f16 v[{d}:{d + 15}], v[{a}:{a + 3}], v[{b}:{b + 3}], 0",
                   reads=rng(a, 4) + rng(b, 4), writes=rng(d, 16), mfma=True)

    # ---- the minima of one group, as a list of closures -------------------------------------------------------------
    def minima(self, g):
        s, R = self.s, g["rmin"]

        def col(cm, d, q):
            return lambda: s.ins(f"v_min3_i32 v{cm}, v{cm}, v{d + 2 * q}, v{d + 2 * q + 1}",
                                 reads=[cm, d + 2 * q, d + 2 * q + 1], writes=[cm])

        def row3(v, dA, dB):
            return lambda: s.ins(f"v_min3_i32 v{R + v}, v{R + v}, v{dA + v}, v{dB + v}",
                                 reads=[R + v, dA + v, dB + v], writes=[R + v])

        def row2(v, d):
            return lambda: s.ins(f"v_min_i32 v{R + v}, v{R + v}, v{d + v}", reads=[R + v, d + v], writes=[R + v])

        L = []
        if g["kind"] == "init":
            d, = g["bufs"]
            L = [col(CM + g["tiles"][0], d, q) for q in range(8)]
        elif g["kind"] == "pair":
            dA, dB = g["bufs"]
            cA, cB = CM + g["tiles"][0], CM + g["tiles"][1]
            for q in range(8):          # interleaved so that no instruction depends on the one before
                L += [col(cA, dA, q), row3(2 * q, dA, dB), col(cB, dB, q), row3(2 * q + 1, dA, dB)]
        else:                           # a lone tile behind the pairs (the 4-tile shares of row tile 16)
            d, = g["bufs"]
            for q in range(8):
                L += [col(CM + g["tiles"][0], d, q), row2(2 * q, d), row2(2 * q + 1, d)]
        return L

    # ---- the reduction of a finished row tile, in stages --------------------------------------------------------------
    def red_stage(self):
        if self.red is None:
            return
        s = self.s
        buf, shared, st = self.red
        if st == 0:
            for v in range(16):
                s.ins(f"ds_write_b32 %4, v{buf + v} offset:{v * RED_STRIDE}", reads=[buf + v])
            self.free.append(buf)                   # the row minima are on their way to LDS: the buffer is free
        elif st == 6:
            s.ins(f"v_min_i32 v{ACC}, v{ACC}, v{T}", reads=[ACC, T], writes=[ACC])
            if shared:
                s.ins(f"ds_min_i32 %8, v{ACC}", reads=[ACC])
            else:
                s.ins(f"v_max_i32 v{ROWMAX}, v{ROWMAX}, v{ACC}", reads=[ROWMAX, ACC], writes=[ROWMAX])
        else:
            chunk = st - 2                          # fold what stage st - 1 read, then read the next four values
            if chunk == 0:
                s.ins(f"v_min3_i32 v{ACC}, v{T}, v{T + 1}, v{T + 2}", reads=rng(T, 3), writes=[ACC])
                s.ins(f"v_min_i32 v{ACC}, v{ACC}, v{T + 3}", reads=[ACC, T + 3], writes=[ACC])
            elif chunk > 0:
                s.ins(f"v_min3_i32 v{ACC}, v{ACC}, v{T}, v{T + 1}", reads=[ACC, T, T + 1], writes=[ACC])
                s.ins(f"v_min3_i32 v{ACC}, v{ACC}, v{T + 2}, v{T + 3}", reads=[ACC, T + 2, T + 3], writes=[ACC])
            if st <= 4:
                c = st - 1
                s.ins(f"ds_read_b64 v[{T}:{T + 1}], %5 offset:{16 * c}", writes=rng(T, 2), lds_load=True)
                s.ins(f"ds_read_b64 v[{T + 2}:{T + 3}], %5 offset:{16 * c + 8}", writes=rng(T + 2, 2), lds_load=True)
            else:
                s.ins(f"ds_bpermute_b32 v{T}, %7, v{ACC}", reads=[ACC], writes=[T], lds_load=True)
        self.red = (buf, shared, st + 1) if st < 6 else None

    def reduce_blocking(self, buf):
        """The last row tile of the candidate: nothing left to hide it behind.  All 16 values of a row are read back at once
        (two free result buffers as landing zone and scratch), three LDS round trips in all."""
        s = self.s
        assert self.red is None
        t, u = [b for b in self.free if b != buf][:2]
        for v in range(16):
            s.ins(f"ds_write_b32 %4, v{buf + v} offset:{v * RED_STRIDE}", reads=[buf + v])
        s.wait()
        for q in range(8):
            s.ins(f"ds_read_b64 v[{t + 2 * q}:{t + 2 * q + 1}], %5 offset:{8 * q}", writes=rng(t + 2 * q, 2), lds_load=True)
        s.wait()
        for i in range(5):
            s.ins(f"v_min3_i32 v{u + i}, v{t + 3 * i}, v{t + 3 * i + 1}, v{t + 3 * i + 2}", reads=rng(t + 3 * i, 3), writes=[u + i])
        s.ins(f"v_min3_i32 v{u}, v{u}, v{u + 1}, v{u + 2}", reads=rng(u, 3), writes=[u])
        s.ins(f"v_min3_i32 v{u + 3}, v{u + 3}, v{u + 4}, v{t + 15}", reads=[u + 3, u + 4, t + 15], writes=[u + 3])
        s.ins(f"v_min_i32 v{ACC}, v{u}, v{u + 3}", reads=[u, u + 3], writes=[ACC])
        s.ins(f"ds_bpermute_b32 v{T}, %7, v{ACC}", reads=[ACC], writes=[T], lds_load=True)
        s.wait()
        s.ins(f"v_min_i32 v{ACC}, v{ACC}, v{T}", reads=[ACC, T], writes=[ACC])
        s.ins(f"v_max_i32 v{ROWMAX}, v{ROWMAX}, v{ACC}", reads=[ROWMAX, ACC], writes=[ROWMAX])

    # ---- one step -----------------------------------------------------------------------------------------------------
    def step(self, g, nxt, extra=None):
        """g: the group whose MFMAs are issued now (its B fragments were requested a step ago); nxt: the tiles of the group
        after it (their B fragments are requested here) or None; extra: a closure issued with the prefetch (A loads)."""
        s = self.s
        assert self.next_b == g["tiles"]
        b = BSET[self.bpar]
        self.bpar ^= 1
        s.wait()
        self.free.sort()                    # lowest free buffer first: the four variant paths meet in one state
        g["bufs"] = [self.free.pop(0) for _ in g["tiles"]]
        if g["kind"] == "init":
            self.rmin = g["bufs"][0]
        g["rmin"] = self.rmin
        prev, self.pending = self.pending, g
        m = self.minima(prev) if prev is not None else []
        half = len(m) // 2 if len(g["tiles"]) == 2 else 0

        self.mfma(g["bufs"][0], g["a"], b)
        for _ in range(DBG_NOP):
            s.raw("s_nop 15")
        for f in m[:half]:
            f()
        if len(g["tiles"]) == 2:
            # one MFMA, half of the previous group's minima, the other MFMA, the other half: a wave does not queue a second
            # MFMA right behind its own first one (the matrix pipe takes 32 cycles per MFMA, 16 minima take 64)
            self.mfma(g["bufs"][1], g["a"], b + 4)
        if nxt is not None:
            self.load_b(nxt)
        if extra is not None:
            extra()
        # the reduction's write stage reads the previous row tile's minima: they are final once the last group of that row
        # tile is folded, i.e. after this step's minima when this step opens a new row tile -- so stage 0 waits a step
        if not (self.red is not None and self.red[2] == 0 and g["kind"] == "init"):
            self.red_stage()
        for f in m[half:]:
            f()
        if prev is not None and prev["kind"] != "init":
            self.free += prev["bufs"]

    def drain(self):
        prev, self.pending = self.pending, None
        for f in self.minima(prev):
            f()
        if prev["kind"] != "init":
            self.free += prev["bufs"]


def groups_of(rt_a, tiles, init_tile):
    """init group on `init_tile`, then pairs over the rest, a lone tile if one is left"""
    rest = [t for t in tiles if t != init_tile]
    G = [dict(kind="init", a=rt_a, tiles=[init_tile])]
    while len(rest) >= 2:
        G.append(dict(kind="pair", a=rt_a, tiles=rest[:2]))
        rest = rest[2:]
    if rest:
        G.append(dict(kind="single", a=rt_a, tiles=rest))
    return G


def full_row_tile(a):
    return groups_of(a, list(range(NCT)), NCT - 1)


# ---- prologue ------------------------------------------------------------------------------------------------------
head = Stream()
head.ins(f"ds_read_b128 v[{ASET[0]}:{ASET[0] + 3}], %3 offset:0", writes=rng(ASET[0], 4), lds_load=True)
for ct in range(NCT):
    head.ins(f"v_mov_b32 v{CM + ct}, {INF}", writes=[CM + ct])
head.ins(f"v_mov_b32 v{ROWMAX}, 0", writes=[ROWMAX])

# ---- the wave's share of row tile 16 FIRST (its reduction then rides on row tile 0's steps), one code path per variant,
# up to and including the step that opens row tile 0; the paths meet in the same state ---------------------------------
variants, meet = [], None
for var in range(4):
    s = Stream(head)
    p = Pipe(s, BUF)
    G = groups_of(ASET[0], list(range(*PART[var])), PART[var][0])
    first0 = full_row_tile(ASET[1])[0]
    seq = G + [first0]
    p.load_b(seq[0]["tiles"])
    for i, g in enumerate(seq):
        nxt = seq[i + 1]["tiles"] if i + 1 < len(seq) else full_row_tile(ASET[1])[1]["tiles"]
        extra = (lambda: p.load_a(ASET[1], "%2", 0)) if i == 0 else None
        if g is first0:
            p.red = (p.rmin, True, 0)               # row tile 16's minima: final after this step
        p.step(g, nxt, extra)
    assert s.quiet()
    state = (tuple(sorted(p.free)), p.rmin, p.bpar, p.red, tuple(p.next_b), tuple(first0["bufs"]))
    assert meet is None or meet == state, (meet, state)
    meet = state
    variants.append((s, p))

# ---- the four full row tiles (common code) ----------------------------------------------------------------------------
s0, p0 = variants[0]
tail = Stream()
tail.n = max(s.n for s, _ in variants)
q = Pipe(tail, p0.free, p0.bpar)
q.rmin, q.red, q.next_b = p0.rmin, p0.red, p0.next_b
q.pending = dict(p0.pending)
seq = full_row_tile(ASET[1])[1:]
for k in range(1, 4):
    seq += full_row_tile(ASET[(k + 1) & 1])
opens = 0
for i, g in enumerate(seq):
    nxt = seq[i + 1]["tiles"] if i + 1 < len(seq) else None
    extra = None
    if g["kind"] == "init":
        opens += 1
        q.red = (q.rmin, False, 0)
    if g["kind"] == "pair" and g["tiles"][0] == 0 and opens < 3:
        # the A fragment of the next row tile, once the last MFMA of the row tile before this one is out
        k = opens + 1
        extra = (lambda k=k: q.load_a(ASET[(k + 1) & 1], "%2", 4096 * k))
    q.step(g, nxt, extra)
q.drain()
assert q.red is None
q.reduce_blocking(q.rmin)
for ct in range(NCT):
    tail.ins(f"ds_min_i32 %6, v{CM + ct} offset:{ct * 128}", reads=[CM + ct])
tail.wait()
tail.ins(f"v_mov_b32 %0, v{ROWMAX}", reads=[ROWMAX])

# ---- assemble -------------------------------------------------------------------------------------------------------
out = list(head.out)
for var in range(4):
    body = variants[var][0].out
    if var < 3:
        out.append(f"s_cmp_eq_u32 %9, {var}")
        out.append(f"s_cbranch_scc0 {10 + var}f")
    out += body
    if var < 3:
        out.append("s_branch 20f")
        out.append(f"{10 + var}:")
out.append("20:")
out += tail.out

regs = sorted(set(range(20, 37)) | set(range(40, 57)) | set(range(60, 73)) | set(range(80, 96)) | set(range(100, 164)))
here = os.path.dirname(os.path.abspath(__file__))
dst = os.path.join(here, "..", "multimoda-rs_amd", "csrc", "mm_screen_mx_asm.inc")
with open(dst, "w") as f:
    f.write("// GENERATED by tools/gen_screen_mx.py -- do not edit.  Main phase of k_screen_mx: see the generator's docstring.\n")
    f.write(f"// {len(out)} instructions\n")
    f.write("#define MM_SCREEN_MX_ASM \\\n")
    for line in out:
        f.write(f'    "{line}\\n" \\\n')
    f.write('    ""\n')
    f.write("#define MM_SCREEN_MX_CLOBBERS " + ", ".join(f'"v{r}"' for r in regs) + ', "scc", "memory"\n')
    f.write(f"#define MM_SCREEN_MX_RED_STRIDE {RED_STRIDE}\n")
print(len(out), "instructions ->", os.path.normpath(dst))
