#!/usr/bin/env python3
"""Generates multimoda-rs_amd/csrc/mm_screen_mx_asm.inc: the main phase of k_screen_mx (mm_kernels.hip) as ONE asm block on
fixed registers, so that the order is exactly the software pipeline we want -- the MFMAs of one group of tiles (1024
squared distances each, on the matrix pipe) issued ahead of the v_min3_i32 that fold the PREVIOUS group on the vector pipe.
The compiler's scheduler does not produce this order (tools/ubench_mfma16*.hip: 46 ns per tile compiler-scheduled, 34 ns
hand-ordered).

One WAVE, one candidate, all 17 x 17 tiles of it: no other wave touches the candidate, so there is no barrier and no
shared accumulator anywhere in the candidate loop.  The 289 tiles are ONE stream of groups, pipelined across the row-tile
boundaries: row tile 0, then a loop of eight iterations over two row tiles each.

A row tile is an `init` group (column tile 16) followed by eight pairs of column tiles.  Six 16-register result buffers:
the pairs alternate between (P, Q) and (R, S); the init tile's MFMA writes X0 or X1 (row tile parity), and that buffer
simply BECOMES the running row minima of the row tile (no instruction) -- every pair folds into it with 16 three-operand
minima.  Per row tile: 8 (column fold of the init tile) + 8 x 32 = 264 vector instructions for 17 tiles; the floor of two
values per instruction, each value used twice, would be 272 -- the init tile's row half costs nothing.

The cross-lane reduction of a row tile's minima (LDS transpose: write, read back a row per lane, fold, meet the other
half through ds_bpermute) is spread over the steps of the NEXT row tile, one LDS round trip per step, each behind a wait
the pipeline has anyway.  The column minima stay in registers for the whole candidate (this wave has seen every row);
lanes l and l + 32 hold different rows of the same column, so at the end v_permlane32_swap brings the halves of two
column tiles together, one minimum per pair, and the maximum over everything leaves in one register.

Wait states: nothing in an asm string is padded by the assembler.  The generator tracks every MFMA's destination and
pads (s_nop) where fewer than MFMA_STATES instructions separate it from the first instruction that touches the buffer,
pads a vector write ahead of a v_permlane32_swap, and refuses to read a register an LDS load is still filling.

Register map (VGPR):
  v36        rowmax     max over rows of the row minima (signed-int order on f32 bits, floored at 0)
  v[40:56]   cm[17]     running column minima per column tile (this lane's column, this lane's rows)
  v[60:63], v[64:67]    A operand fragment of the even / the odd row tiles
  v[68:71], v72         reduction: four values read back, accumulator
  v73                   LDS address of the A fragments, advanced by two row tiles per loop iteration
  v[80:87], v[88:95]    B operand fragments, two groups (double buffer)
  v[100:195]            six result buffers P, Q, R, S, X0, X1
Operands: %0 out: this lane's maximum; %1 =s loop counter; %2 vB (LDS byte address of this lane's B fragment in column
tile 0, the wave's own copy); %3 vA (A fragment of row tile 0; the next ones 1024 bytes apart); %4 vRW / %5 vRR
(row-reduction scratch: write / read address); %6 vPERM (4 * (lane ^ 32))."""
import os

INF = "0x7f800000"
P, Q, R, S, X = 100, 116, 132, 148, [164, 180]
ROWMAX, CM, ASET, T, ACC, AADDR, BSET = 36, 40, [60, 64], 68, 72, 73, [80, 88]
if os.environ.get("MX_COMPACT") == "1":      # experiment: everything below v168
    P, Q, R, S, X = 72, 88, 104, 120, [136, 152]
    ROWMAX, CM, ASET, T, ACC, AADDR, BSET = 24, 25, [48, 52], 42, 46, 47, [56, 64]
NCT = 17
NRT = 17
RED_STRIDE = 136          # bytes between rows of the reduction scratch (34 dwords: 8-byte aligned reads, 2-way conflicts)
MFMA_STATES = 12          # instructions between an 8-pass MFMA and the first touch of its destination (11 required)
PERMLANE_STATES = 3       # a vector write and a v_permlane32_swap reading it (2 required)

DBG_NOP = int(os.environ.get("MX_DBG_NOP", "0"))           # s_nop 15 count in front of every group of minima


class Stream:
    """Straight-line instruction list with the checks described above."""

    def __init__(self):
        self.out = []
        self.n = 0                              # wait states issued so far
        self.mfma_at = {}                       # register -> state count at which an MFMA writing it was issued
        self.valu_at = {}                       # register -> state count of the last vector write
        self.loading = set()                    # registers an LDS load is filling (cleared by s_waitcnt)

    def _pad(self, need):
        if need > 0:
            self.out.append(f"s_nop {need - 1}")
            self.n += need

    def ins(self, text, reads=(), writes=(), lds_load=False, mfma=False, valu=False, permlane=False):
        reads, writes = list(reads), list(writes)
        need = 0
        for r in reads + writes:
            if r in self.mfma_at:
                need = max(need, MFMA_STATES - (self.n - self.mfma_at[r]))
            if permlane and r in self.valu_at:
                need = max(need, PERMLANE_STATES - (self.n - self.valu_at[r]))
        self._pad(need)
        for r in reads + writes:
            self.mfma_at.pop(r, None)
        for r in reads:
            assert r not in self.loading, f"v{r} read while an LDS load is in flight"
        for r in writes:
            assert r not in self.loading or lds_load, f"v{r} overwritten while an LDS load is in flight"
        self.out.append(text)
        self.n += 1
        if mfma:
            for r in writes:
                self.mfma_at[r] = self.n
        if valu or permlane:
            for r in writes:
                self.valu_at[r] = self.n
        if lds_load:
            self.loading.update(writes)

    def wait(self):
        self.out.append("s_waitcnt lgkmcnt(0)")
        self.n += 1
        self.loading.clear()

    def raw(self, text):                       # scalar / control instructions
        self.out.append(text)
        self.n += 1

    def mark(self):
        return len(self.out)


def rng(base, n):
    return list(range(base, base + n))


class Pipe:
    """The software pipeline: step(g) issues the MFMAs of group g and, beside them, the minima of the group before."""

    def __init__(self, s):
        self.s = s
        self.pending = None                 # group whose minima are still to be issued
        self.bpar = 0                       # B operand set of the next group
        self.red = None                     # (buffer, next stage) of the reduction in progress
        self.next_b = None                  # tiles whose B fragments are in flight / loaded for the next group
        self.busy = set()                   # result buffers holding values still to be folded (bookkeeping check)

    # ---- operands -------------------------------------------------------------------------------------------------
    def load_b(self, tiles):
        # a column fragment is stored as 8 bytes per lane -- (x1, x2, y1, y2) or (256, 1, n2h, n2l) -- and read twice: the K
        # slots 4..7 repeat slots 0..3 (the row fragments are laid out to match; for the norm half their slots 4..7 are 0)
        base = BSET[self.bpar]
        for i, t in enumerate(tiles):
            d = base + 4 * i
            self.s.ins(f"ds_read_b64 v[{d}:{d + 1}], %2 offset:{t * 512}", writes=rng(d, 2), lds_load=True)
            self.s.ins(f"ds_read_b64 v[{d + 2}:{d + 3}], %2 offset:{t * 512}", writes=rng(d + 2, 2), lds_load=True)
        self.next_b = list(tiles)

    def load_a(self, reg, offset):
        self.s.ins(f"ds_read_b128 v[{reg}:{reg + 3}], v{AADDR} offset:{offset}", reads=[AADDR], writes=rng(reg, 4), lds_load=True)

    def mfma(self, d, a, b):
        assert d not in self.busy, f"result buffer v{d} overwritten before it was folded"
        self.busy.add(d)
        self.s.ins(f"v_mfma_f32_32x32x16_f16 v[{d}:{d + 15}], v[{a}:{a + 3}], v[{b}:{b + 3}], 0",
                   reads=rng(a, 4) + rng(b, 4), writes=rng(d, 16), mfma=True)

    # ---- the minima of one group, as a list of closures -------------------------------------------------------------
    def minima(self, g):
        s, Rm = self.s, g["rmin"]

        def col(cm, d, q):
            return lambda: s.ins(f"v_min3_i32 v{cm}, v{cm}, v{d + 2 * q}, v{d + 2 * q + 1}",
                                 reads=[cm, d + 2 * q, d + 2 * q + 1], writes=[cm], valu=True)

        def row3(v, dA, dB):
            return lambda: s.ins(f"v_min3_i32 v{Rm + v}, v{Rm + v}, v{dA + v}, v{dB + v}",
                                 reads=[Rm + v, dA + v, dB + v], writes=[Rm + v], valu=True)

        if g["kind"] == "init":
            d, = g["bufs"]
            return [col(CM + g["tiles"][0], d, q) for q in range(8)]
        dA, dB = g["bufs"]
        cA, cB = CM + g["tiles"][0], CM + g["tiles"][1]
        L = []
        for q in range(8):              # interleaved so that no instruction depends on the one before
            L += [col(cA, dA, q), row3(2 * q, dA, dB), col(cB, dB, q), row3(2 * q + 1, dA, dB)]
        return L

    # ---- the reduction of a finished row tile, in stages --------------------------------------------------------------
    def red_stage(self):
        if self.red is None:
            return
        s = self.s
        buf, st = self.red
        if st == 0:
            for v in range(16):
                s.ins(f"ds_write_b32 %4, v{buf + v} offset:{v * RED_STRIDE}", reads=[buf + v])
            self.busy.discard(buf)                  # the row minima are on their way to LDS: the buffer is free
        elif st == 6:
            s.ins(f"v_min_i32 v{ACC}, v{ACC}, v{T}", reads=[ACC, T], writes=[ACC], valu=True)
            s.ins(f"v_max_i32 v{ROWMAX}, v{ROWMAX}, v{ACC}", reads=[ROWMAX, ACC], writes=[ROWMAX], valu=True)
        else:
            chunk = st - 2                          # fold what stage st - 1 read, then read the next four values
            if chunk == 0:
                s.ins(f"v_min3_i32 v{ACC}, v{T}, v{T + 1}, v{T + 2}", reads=rng(T, 3), writes=[ACC], valu=True)
                s.ins(f"v_min_i32 v{ACC}, v{ACC}, v{T + 3}", reads=[ACC, T + 3], writes=[ACC], valu=True)
            elif chunk > 0:
                s.ins(f"v_min3_i32 v{ACC}, v{ACC}, v{T}, v{T + 1}", reads=[ACC, T, T + 1], writes=[ACC], valu=True)
                s.ins(f"v_min3_i32 v{ACC}, v{ACC}, v{T + 2}, v{T + 3}", reads=[ACC, T + 2, T + 3], writes=[ACC], valu=True)
            if st <= 4:
                c = st - 1
                s.ins(f"ds_read_b64 v[{T}:{T + 1}], %5 offset:{16 * c}", writes=rng(T, 2), lds_load=True)
                s.ins(f"ds_read_b64 v[{T + 2}:{T + 3}], %5 offset:{16 * c + 8}", writes=rng(T + 2, 2), lds_load=True)
            else:
                s.ins(f"ds_bpermute_b32 v{T}, %6, v{ACC}", reads=[ACC], writes=[T], lds_load=True)
        self.red = (buf, st + 1) if st < 6 else None

    def reduce_blocking(self, buf, t, u):
        """The last row tile of the candidate: nothing left to hide it behind.  All 16 values of a row are read back at once
        (two free result buffers as landing zone and scratch), three LDS round trips in all."""
        s = self.s
        assert self.red is None and t not in self.busy and u not in self.busy
        for v in range(16):
            s.ins(f"ds_write_b32 %4, v{buf + v} offset:{v * RED_STRIDE}", reads=[buf + v])
        self.busy.discard(buf)
        s.wait()
        for q in range(8):
            s.ins(f"ds_read_b64 v[{t + 2 * q}:{t + 2 * q + 1}], %5 offset:{8 * q}", writes=rng(t + 2 * q, 2), lds_load=True)
        s.wait()
        for i in range(5):
            s.ins(f"v_min3_i32 v{u + i}, v{t + 3 * i}, v{t + 3 * i + 1}, v{t + 3 * i + 2}", reads=rng(t + 3 * i, 3), writes=[u + i], valu=True)
        s.ins(f"v_min3_i32 v{u}, v{u}, v{u + 1}, v{u + 2}", reads=rng(u, 3), writes=[u], valu=True)
        s.ins(f"v_min3_i32 v{u + 3}, v{u + 3}, v{u + 4}, v{t + 15}", reads=[u + 3, u + 4, t + 15], writes=[u + 3], valu=True)
        s.ins(f"v_min_i32 v{ACC}, v{u}, v{u + 3}", reads=[u, u + 3], writes=[ACC], valu=True)
        s.ins(f"ds_bpermute_b32 v{T}, %6, v{ACC}", reads=[ACC], writes=[T], lds_load=True)
        s.wait()
        s.ins(f"v_min_i32 v{ACC}, v{ACC}, v{T}", reads=[ACC, T], writes=[ACC], valu=True)
        s.ins(f"v_max_i32 v{ROWMAX}, v{ROWMAX}, v{ACC}", reads=[ROWMAX, ACC], writes=[ROWMAX], valu=True)

    # ---- one step -----------------------------------------------------------------------------------------------------
    def step(self, g, nxt, extra=None):
        """g: the group whose MFMAs are issued now (its B fragments were requested a step ago); nxt: the tiles of the group
        after it (their B fragments are requested here); extra: a closure issued with the prefetch (A loads)."""
        s = self.s
        assert self.next_b == g["tiles"]
        b = BSET[self.bpar]
        self.bpar ^= 1
        s.wait()
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
        self.load_b(nxt)
        if extra is not None:
            extra()
        # the reduction's write stage reads the previous row tile's minima: they are final once the last group of that row
        # tile is folded, i.e. after this step's minima when this step opens a new row tile -- so stage 0 waits a step
        if not (self.red is not None and self.red[1] == 0 and g["kind"] == "init"):
            self.red_stage()
        for f in m[half:]:
            f()
        if prev is not None and prev["kind"] != "init":
            for d in prev["bufs"]:
                self.busy.discard(d)

    def drain(self):
        prev, self.pending = self.pending, None
        for f in self.minima(prev):
            f()
        for d in prev["bufs"]:
            if prev["kind"] != "init":
                self.busy.discard(d)


def row_tile(k):
    """groups of row tile k: init on column tile 16 into X[k & 1], then pairs alternating between (P, Q) and (R, S)"""
    a, x = ASET[k & 1], X[k & 1]
    G = [dict(kind="init", a=a, tiles=[NCT - 1], bufs=[x], rmin=x, rt=k)]
    for p in range(8):
        G.append(dict(kind="pair", a=a, tiles=[2 * p, 2 * p + 1], bufs=[R, S] if p & 1 else [P, Q], rmin=x, rt=k, p=p))
    return G


def run_row_tile(pipe, k, a_offset_next, a_reg_next):
    """steps of row tile k; in its first pair the A fragment of row tile k + 1 is requested (offset relative to v73)"""
    G = row_tile(k)
    for i, g in enumerate(G):
        nxt = G[i + 1]["tiles"] if i + 1 < len(G) else [NCT - 1]
        extra = None
        if g["kind"] == "init" and k > 0:
            pipe.red = (X[(k - 1) & 1], 0)           # the row tile before: final after this step
        if g["kind"] == "pair" and g["p"] == 0:
            extra = (lambda: pipe.load_a(a_reg_next, a_offset_next))
        pipe.step(g, nxt, extra)


s = Stream()
pipe = Pipe(s)
# ---- prologue: row tile 0 -------------------------------------------------------------------------------------------
s.ins(f"v_mov_b32 v{AADDR}, %3", writes=[AADDR], valu=True)
s.ins(f"ds_read_b128 v[{ASET[0]}:{ASET[0] + 3}], %3 offset:0", writes=rng(ASET[0], 4), lds_load=True)
pipe.load_b([NCT - 1])
for ct in range(NCT):
    s.ins(f"v_mov_b32 v{CM + ct}, {INF}", writes=[CM + ct], valu=True)
s.ins(f"v_mov_b32 v{ROWMAX}, 0", writes=[ROWMAX], valu=True)
run_row_tile(pipe, 0, 1024, ASET[1])
s.raw("s_mov_b32 %1, 8")
s.raw("1:")
# ---- loop body: row tiles (2i + 1, 2i + 2), v73 = A address of row tile 2i; generated twice, emitted once ------------------
bodies = []
for it in range(2):
    m0 = s.mark()
    run_row_tile(pipe, 1, 2048, ASET[0])
    run_row_tile(pipe, 2, 3072, ASET[1])         # (the last iteration requests a 18th row fragment: read, never used)
    s.ins(f"v_add_u32 v{AADDR}, 2048, v{AADDR}", reads=[AADDR], writes=[AADDR], valu=True)
    bodies.append(s.out[m0:])
    if it == 0:
        keep = s.mark()
assert bodies[0] == bodies[1], "the loop body must leave the pipeline in the state it found it in"
del s.out[keep:]
s.raw("s_sub_u32 %1, %1, 1")
s.raw("s_cmp_lg_u32 %1, 0")
s.raw("s_cbranch_scc1 1b")
# ---- epilogue ---------------------------------------------------------------------------------------------------------
pipe.drain()
assert pipe.red is None
pipe.reduce_blocking(X[0], P, Q)
# column minima: lanes l and l + 32 hold different rows of column l & 31.  The swap leaves the lower halves of two column
# tiles in one register and the upper halves in the other; their minimum is complete for both tiles.
for p in range(8):
    a, b = CM + 2 * p, CM + 2 * p + 1
    s.ins(f"v_permlane32_swap_b32 v{a}, v{b}", reads=[a, b], writes=[a, b], permlane=True)
    s.ins(f"v_min_i32 v{a}, v{a}, v{b}", reads=[a, b], writes=[a], valu=True)
a = CM + 16
s.ins(f"v_mov_b32 v{T}, v{a}", reads=[a], writes=[T], valu=True)
s.ins(f"v_permlane32_swap_b32 v{a}, v{T}", reads=[a, T], writes=[a, T], permlane=True)
s.ins(f"v_min_i32 v{a}, v{a}, v{T}", reads=[a, T], writes=[a], valu=True)
c = [CM + 2 * p for p in range(9)]
s.ins(f"v_max3_i32 v{c[0]}, v{c[0]}, v{c[1]}, v{c[2]}", reads=c[0:3], writes=[c[0]], valu=True)
s.ins(f"v_max3_i32 v{c[3]}, v{c[3]}, v{c[4]}, v{c[5]}", reads=c[3:6], writes=[c[3]], valu=True)
s.ins(f"v_max3_i32 v{c[6]}, v{c[6]}, v{c[7]}, v{c[8]}", reads=c[6:9], writes=[c[6]], valu=True)
s.ins(f"v_max3_i32 v{c[0]}, v{c[0]}, v{c[3]}, v{c[6]}", reads=[c[0], c[3], c[6]], writes=[c[0]], valu=True)
s.ins(f"v_max_i32 %0, v{c[0]}, v{ROWMAX}", reads=[c[0], ROWMAX])
s.wait()                                       # (the unused B and A requests of the last steps)

out = s.out
regs = sorted({ROWMAX} | set(range(CM, CM + NCT)) | set(range(ASET[0], ASET[0] + 8)) | set(range(T, T + 4)) | {ACC, AADDR}
              | set(range(BSET[0], BSET[0] + 16)) | set(range(P, P + 64)) | set(range(X[0], X[0] + 32)))
here = os.path.dirname(os.path.abspath(__file__))
dst = os.environ.get("MX_OUT") or os.path.join(here, "..", "multimoda-rs_amd", "csrc", "mm_screen_mx_asm.inc")
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
