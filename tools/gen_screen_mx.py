#!/usr/bin/env python3
"""Generates multimoda-rs_amd/csrc/mm_screen_mx_asm.inc: the main phase of k_screen_mx (mm_kernels.hip) as asm blocks on
fixed registers, so that the order is exactly the software pipeline we want -- the MFMAs of one group of tiles (1024
squared distances each, on the matrix pipe) issued ahead of the v_min3_i32 that fold the PREVIOUS group on the vector pipe.
The compiler's scheduler does not produce this order (tools/ubench_mfma16*.hip: 46 ns per tile compiler-scheduled, 34 ns
hand-ordered).

One block per COLUMN-TILE COUNT `nct` (2 .. 17; the column minima live in one register per column tile, so the column
loop is unrolled) in two forms; the ROW-tile count is a run-time operand (a loop over pairs of row tiles, an optional tail
row tile):
  plain   the wave sees every column of the candidate: the row minima are final when a row tile is done
  carry   the candidate's columns come in several blocks of `nct` tiles: a row tile's minima are met with those of the
          blocks before through a wave-private row store in LDS (read, min, write back); after the last block the
          kernel takes the maximum over the store

One WAVE, one candidate (or column block of it), all row tiles x nct tiles: no other wave touches it, so there is no
barrier and no shared accumulator anywhere in the candidate loop.  The tiles are ONE stream of groups, pipelined across the
row-tile boundaries: row tile 0, then a loop over two row tiles per iteration, then (even row-tile counts) one more.

A row tile is an `init` group (the last column tile) followed by pairs of column tiles and, for an even `nct`, one single
tile.  Six 16-register result buffers: the pairs alternate between (P, Q) and (R, S); the init tile's MFMA writes X0 or X1
(row tile parity), and that buffer simply BECOMES the running row minima of the row tile (no instruction) -- every pair
folds into it with 16 three-operand minima.  Per row tile at nct = 17: 8 (column fold of the init tile) + 8 x 32 = 264
vector instructions for 17 tiles; the floor of two values per instruction, each value used twice, would be 272 -- the init
tile's row half costs nothing.

The cross-lane reduction of a row tile's minima (LDS transpose: two rows per ds_write2_b32, a row read back per lane 16
bytes at a time, fold, meet the other half through v_permlane32_swap) is spread over the steps of the NEXT row tile, one
stage at the head of a step, where the row tile has steps enough (nct >= 12: seven stages, four values read back at a time;
below that all 16 values of a row are read back at once into a landing zone of their own, three round trips, the surplus
ones behind waits of their own).

A candidate's column (B) fragments are read from LDS ONCE, into accumulation registers a[4t : 4t + 3], and every MFMA of column
tile t takes its B operand from there (the init tile's come with the prologue, the others behind the first MFMA): the
only LDS traffic of the main loop is a row tile's A fragment and its reduction.  The column minima stay in registers for the whole candidate (this wave has seen every row); lanes l
and l + 32 hold different rows of the same column, so at the end v_permlane32_swap brings the halves of two column tiles
together, one minimum per pair, and the maximum over everything leaves in one register.

Wait states: nothing in an asm string is padded by the assembler.  The generator tracks every MFMA's destination and
pads (s_nop) where fewer than MFMA_STATES instructions separate it from the first instruction that touches the buffer,
pads a vector write ahead of a v_permlane32_swap, and refuses to read a register an LDS load is still filling.  Where two
paths meet (the loop's entry, its exit) the code after the join is generated from the MERGED state of the predecessors --
every hazard at the distance of the closer path -- until the text settles.  tests/test_screen_mx_asm.py executes every block
symbolically for row-tile counts 1 .. 7, 17 and 33.

Register map (VGPR):
  v36        rowmax     max over rows of the row minima (signed-int order on f32 bits, floored at 0)
  v[40:56]   cm[nct]    running column minima per column tile (this lane's column, this lane's rows)
  v[60:63], v[64:67]    A operand fragment of the even / the odd row tiles
  v[68:71], v72         reduction: four values read back, accumulator
  v73                   LDS address of the A fragments, advanced by two row tiles per loop iteration
  v74, v75              carry: address of the row store (advanced like v73), the stored minimum read back
  v76                   LDS address of row 8 of this lane's column in the reduction scratch
  v[84:179]             six result buffers P, Q, R, S, X0, X1
  v[180:195]            nct < 12: landing zone of the reduction (16 values of a row)
  a[0 : 4 nct)          the candidate's column (B) fragments
(180 or 196 vector registers + up to 68 accumulation registers: two waves per SIMD, to the last register at nct = 17 --
tests/test_kernel_resources.py.)
Operands: %0 out: this lane's maximum; %1 =s loop counter; %2 vB (LDS byte address of this lane's B fragment in column
tile 0, the wave's own copy); %3 vA (A fragment of row tile 0; the next ones 1024 bytes apart); %4 vRW / %5 vRR
(row-reduction scratch: write / read address); %6 unused (a scalar zero: the halves meet through v_permlane32_swap); %7 s: loop iterations = (row tiles - 1) / 2;
%8 s: 1 if the row-tile count is even (tail row tile); %9 (carry) vRS: this lane's slot in the row store of row tile 0
(the next ones 128 bytes apart)."""
import copy
import os

INF = "0x7f800000"
P, Q, R, S, X = 84, 100, 116, 132, [148, 164]
ROWMAX, CM, ASET, T, ACC, AADDR, RSADDR, TP, LAND = 36, 40, [60, 64], 68, 72, 73, 74, 75, 180
AG = 1000                 # register ids from here on are accumulation registers: AG + i is a{i}
NCT_MIN, NCT_MAX = 2, 17
RED_STRIDE = 144          # bytes between rows of the reduction scratch (36 dwords: 16-byte aligned reads of a row's values)
RW2 = 76                  # LDS address of row 8 of this lane's column in the reduction scratch (ds_write2_b32 reaches 255 dwords)
MFMA_STATES = 12          # instructions between an 8-pass MFMA and the first touch of its destination (11 required)
PERMLANE_STATES = 3       # a vector write and a v_permlane32_swap reading it (2 required)


class Stream:
    """Straight-line instruction list with the checks described above."""

    def __init__(self):
        self.out = []
        self.n = 0                              # wait states issued so far
        self.mfma_at = {}                       # register -> state count at which an MFMA writing it was issued
        self.valu_at = {}                       # register -> state count of the last vector write
        self.loading = set()                    # registers an LDS load is filling (cleared by s_waitcnt)
        self.used = set()                       # every vector register named (the clobber list)

    def _pad(self, need):
        if need > 0:
            self.out.append(f"s_nop {need - 1}")
            self.n += need

    def ins(self, text, reads=(), writes=(), lds_load=False, mfma=False, valu=False, permlane=False):
        reads, writes = list(reads), list(writes)
        self.used.update(reads + writes)
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

    def label(self, text):
        self.out.append(text)

    def mark(self):
        return len(self.out)


def regname(r):
    return f'"a{r - AG}"' if r >= AG else f'"v{r}"'


def rng(base, n):
    return list(range(base, base + n))


class Pipe:
    """The software pipeline: step(g) issues the MFMAs of group g and, beside them, the minima of the group before."""

    def __init__(self, s, nct, carry):
        self.s, self.nct, self.carry = s, nct, carry
        self.pending = None                 # group whose minima are still to be issued
        self.red = None                     # [buffer, list of stages still to run, row-store offset] of the reduction in progress
        self.busy = set()                   # result buffers holding values still to be folded (bookkeeping check)

    # ---- operands -------------------------------------------------------------------------------------------------
    def load_b(self, tiles):
        # a column fragment is stored as 8 bytes per lane -- (x1, x2, y1, y2) or (256, 1, n2h, n2l) -- and read twice: the K
        # slots 4..7 repeat slots 0..3 (the row fragments are laid out to match; for the norm half their slots 4..7 are 0).
        # It goes into ACCUMULATION registers a[4t : 4t + 3] and stays there for the whole candidate: every MFMA of column
        # tile t takes it from there.  (Until round 4 it was read again for every tile: 1 KB of LDS traffic per MFMA, a third
        # of the CU's LDS bandwidth, under which the row reductions queued -- tools/exp_mx2.sh: 7 % of the launch at 17 x 17
        # tiles, 12 % at 7 x 7.)
        for t in tiles:
            d = AG + 4 * t
            self.s.ins(f"ds_read_b64 a[{4 * t}:{4 * t + 1}], %2 offset:{t * 512}", writes=rng(d, 2), lds_load=True)
            self.s.ins(f"ds_read_b64 a[{4 * t + 2}:{4 * t + 3}], %2 offset:{t * 512}", writes=rng(d + 2, 2), lds_load=True)

    def load_a(self, reg, offset):
        self.s.ins(f"ds_read_b128 v[{reg}:{reg + 3}], v{AADDR} offset:{offset}", reads=[AADDR], writes=rng(reg, 4), lds_load=True)

    def mfma(self, d, a, tile):
        assert d not in self.busy, f"result buffer v{d} overwritten before it was folded"
        self.busy.add(d)
        self.s.ins(f"v_mfma_f32_32x32x16_f16 v[{d}:{d + 15}], v[{a}:{a + 3}], a[{4 * tile}:{4 * tile + 3}], 0",
                   reads=rng(a, 4) + rng(AG + 4 * tile, 4), writes=rng(d, 16), mfma=True)

    # ---- the minima of one group, as a list of closures -------------------------------------------------------------
    def minima(self, g):
        s, Rm = self.s, g["rmin"]

        def col(cm, d, q):
            return lambda: s.ins(f"v_min3_i32 v{cm}, v{cm}, v{d + 2 * q}, v{d + 2 * q + 1}",
                                 reads=[cm, d + 2 * q, d + 2 * q + 1], writes=[cm], valu=True)

        def row3(v, dA, dB):
            return lambda: s.ins(f"v_min3_i32 v{Rm + v}, v{Rm + v}, v{dA + v}, v{dB + v}",
                                 reads=[Rm + v, dA + v, dB + v], writes=[Rm + v], valu=True)

        def row2(v, dA):
            return lambda: s.ins(f"v_min_i32 v{Rm + v}, v{Rm + v}, v{dA + v}", reads=[Rm + v, dA + v], writes=[Rm + v], valu=True)

        if g["kind"] == "init":
            d, = g["bufs"]
            return [col(CM + g["tiles"][0], d, q) for q in range(8)]
        if g["kind"] == "single":           # an even nct's last tile: folded alone (24 instructions instead of 16)
            dA, = g["bufs"]
            cA = CM + g["tiles"][0]
            L = []
            for q in range(8):
                L += [col(cA, dA, q), row2(2 * q, dA), row2(2 * q + 1, dA)]
            return L
        dA, dB = g["bufs"]
        cA, cB = CM + g["tiles"][0], CM + g["tiles"][1]
        L = []
        for q in range(8):              # interleaved so that no instruction depends on the one before
            L += [col(cA, dA, q), row3(2 * q, dA, dB), col(cB, dB, q), row3(2 * q + 1, dA, dB)]
        return L

    # ---- the reduction of a finished row tile, in stages --------------------------------------------------------------
    def red_write(self, buf):
        """the 16 row minima of this lane's column to the reduction scratch, two rows per instruction"""
        for v in range(0, 16, 2):
            base, w = ("%4", v) if v < 8 else (f"v{RW2}", v - 8)
            self.s.ins(f"ds_write2_b32 {base}, v{buf + v}, v{buf + v + 1} offset0:{w * RED_STRIDE // 4} offset1:{(w + 1) * RED_STRIDE // 4}",
                       reads=[buf + v, buf + v + 1] + ([RW2] if v >= 8 else []))

    def red_final(self, off):
        """ACC holds this half's minimum of the row, T the other half's (ds_bpermute)."""
        s = self.s
        if self.carry:
            s.ins(f"v_min3_i32 v{ACC}, v{ACC}, v{T}, v{TP}", reads=[ACC, T, TP], writes=[ACC], valu=True)
            s.ins(f"ds_write_b32 v{RSADDR}, v{ACC} offset:{off}", reads=[RSADDR, ACC])
        else:
            s.ins(f"v_min_i32 v{ACC}, v{ACC}, v{T}", reads=[ACC, T], writes=[ACC], valu=True)
            s.ins(f"v_max_i32 v{ROWMAX}, v{ROWMAX}, v{ACC}", reads=[ROWMAX, ACC], writes=[ROWMAX], valu=True)

    def meet_halves(self):
        """lanes l and l + 32 hold the minima of the two halves of a row: afterwards ACC holds the lower half's value in both
        halves and T the upper half's (red_final takes their minimum).  v_permlane32_swap, not ds_bpermute_b32."""
        s = self.s
        s.ins(f"v_mov_b32 v{T}, v{ACC}", reads=[ACC], writes=[T], valu=True)
        s.ins(f"v_permlane32_swap_b32 v{ACC}, v{T}", reads=[ACC, T], writes=[ACC, T], permlane=True)

    def red_prefetch(self, off):
        if self.carry:
            self.s.ins(f"ds_read_b32 v{TP}, v{RSADDR} offset:{off}", reads=[RSADDR], writes=[TP], lds_load=True)

    def red_stages(self, buf, off, narrow):
        """The reduction of the row minima in `buf` as a list of (needs the LDS data of the stage before, closure)."""
        s = self.s

        def write():
            self.red_write(buf)
            self.busy.discard(buf)                  # the row minima are on their way to LDS: the buffer is free

        if narrow:
            def read(c):
                def f():
                    s.ins(f"ds_read_b128 v[{T}:{T + 3}], %5 offset:{16 * c}", writes=rng(T, 4), lds_load=True)
                return f

            def fold(c, then):
                def f():
                    if c == 0:
                        s.ins(f"v_min3_i32 v{ACC}, v{T}, v{T + 1}, v{T + 2}", reads=rng(T, 3), writes=[ACC], valu=True)
                        s.ins(f"v_min_i32 v{ACC}, v{ACC}, v{T + 3}", reads=[ACC, T + 3], writes=[ACC], valu=True)
                    else:
                        s.ins(f"v_min3_i32 v{ACC}, v{ACC}, v{T}, v{T + 1}", reads=[ACC, T, T + 1], writes=[ACC], valu=True)
                        s.ins(f"v_min3_i32 v{ACC}, v{ACC}, v{T + 2}, v{T + 3}", reads=[ACC, T + 2, T + 3], writes=[ACC], valu=True)
                    then()
                return f

            def bperm():
                self.meet_halves()
                self.red_prefetch(off)
            return [(False, write), (False, read(0)), (True, fold(0, read(1))), (True, fold(1, read(2))), (True, fold(2, read(3))),
                    (True, fold(3, bperm)), (True, lambda: self.red_final(off))]

        t = LAND

        def read_all():
            for q in range(4):
                s.ins(f"ds_read_b128 v[{t + 4 * q}:{t + 4 * q + 3}], %5 offset:{16 * q}", writes=rng(t + 4 * q, 4), lds_load=True)

        def fold_all():
            for i in range(5):
                s.ins(f"v_min3_i32 v{t + 3 * i}, v{t + 3 * i}, v{t + 3 * i + 1}, v{t + 3 * i + 2}", reads=rng(t + 3 * i, 3), writes=[t + 3 * i], valu=True)
            s.ins(f"v_min3_i32 v{t}, v{t}, v{t + 3}, v{t + 6}", reads=[t, t + 3, t + 6], writes=[t], valu=True)
            s.ins(f"v_min3_i32 v{t + 9}, v{t + 9}, v{t + 12}, v{t + 15}", reads=[t + 9, t + 12, t + 15], writes=[t + 9], valu=True)
            s.ins(f"v_min_i32 v{ACC}, v{t}, v{t + 9}", reads=[t, t + 9], writes=[ACC], valu=True)
            self.meet_halves()
            self.red_prefetch(off)
        return [(False, write), (False, read_all), (True, fold_all), (True, lambda: self.red_final(off))]

    def red_run(self, n):
        """the next n stages of the reduction in progress; a stage that needs the data of the one before it in the same
        call waits for it (the first stage of a call sits behind the step's own wait)"""
        if self.red is None or n == 0:
            return
        first = True
        for _ in range(n):
            if not self.red[1]:
                break
            needs, f = self.red[1].pop(0)
            if needs and not first:
                self.s.wait()
            f()
            first = False
        if not self.red[1]:
            self.red = None

    def reduce_blocking(self, buf, t, u, off):
        """The last row tile of the candidate: nothing left to hide it behind.  All 16 values of a row are read back at once
        (two free result buffers as landing zone and scratch), three LDS round trips in all."""
        s = self.s
        assert self.red is None and t not in self.busy and u not in self.busy
        self.red_write(buf)
        self.busy.discard(buf)
        self.red_prefetch(off)
        s.wait()
        for q in range(4):
            s.ins(f"ds_read_b128 v[{t + 4 * q}:{t + 4 * q + 3}], %5 offset:{16 * q}", writes=rng(t + 4 * q, 4), lds_load=True)
        s.wait()
        for i in range(5):
            s.ins(f"v_min3_i32 v{u + i}, v{t + 3 * i}, v{t + 3 * i + 1}, v{t + 3 * i + 2}", reads=rng(t + 3 * i, 3), writes=[u + i], valu=True)
        s.ins(f"v_min3_i32 v{u}, v{u}, v{u + 1}, v{u + 2}", reads=rng(u, 3), writes=[u], valu=True)
        s.ins(f"v_min3_i32 v{u + 3}, v{u + 3}, v{u + 4}, v{t + 15}", reads=[u + 3, u + 4, t + 15], writes=[u + 3], valu=True)
        s.ins(f"v_min_i32 v{ACC}, v{u}, v{u + 3}", reads=[u, u + 3], writes=[ACC], valu=True)
        self.meet_halves()
        s.wait()
        self.red_final(off)

    # ---- one step -----------------------------------------------------------------------------------------------------
    def step(self, g, extra=None, red_n=0):
        """g: the group whose MFMAs are issued now; extra: a closure issued with the step's LDS requests (A loads, the
        candidate's other B fragments); red_n: stages of the reduction in progress to run in this step."""
        s = self.s
        if s.loading:
            s.wait()
        prev, self.pending = self.pending, g
        m = self.minima(prev) if prev is not None else []
        half = len(m) // 2 if len(g["tiles"]) == 2 else 0

        # The LDS requests of the step go out FIRST, right behind the step's wait (the A fragment of the next row tile, one
        # stage of the reduction in progress), so that the next step's wait finds them done.
        if extra is not None:
            extra()
        self.red_run(min(red_n, 1))
        self.mfma(g["bufs"][0], g["a"], g["tiles"][0])
        for f in m[:half]:
            f()
        if len(g["tiles"]) == 2:
            # one MFMA, half of the previous group's minima, the other MFMA, the other half: a wave does not queue a second
            # MFMA right behind its own first one (the matrix pipe takes 32 cycles per MFMA, 16 minima take 64)
            self.mfma(g["bufs"][1], g["a"], g["tiles"][1])
        for f in m[half:]:
            f()
        # a short row tile: several stages in one step, each behind a wait of its own
        for _ in range(red_n - 1):
            if self.red is not None:
                if self.red[1][0][0]:
                    s.wait()
                self.red_run(1)
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


def row_tile(nct, k):
    """groups of row tile k: init on the last column tile into X[k & 1], then pairs alternating between (P, Q) and (R, S),
    then (even nct) the single tile left over"""
    a, x = ASET[k & 1], X[k & 1]
    G = [dict(kind="init", a=a, tiles=[nct - 1], bufs=[x], rmin=x, rt=k)]
    npairs = (nct - 1) // 2
    for p in range(npairs):
        G.append(dict(kind="pair", a=a, tiles=[2 * p, 2 * p + 1], bufs=[R, S] if p & 1 else [P, Q], rmin=x, rt=k, p=p))
    if (nct - 1) & 1:
        G.append(dict(kind="single", a=a, tiles=[nct - 2], bufs=[R] if npairs & 1 else [P], rmin=x, rt=k, p=npairs))
    return G


def run_row_tile(pipe, nct, k, a_offset_next, a_reg_next, red_prev, red_off, first=False):
    """steps of row tile k; in its second step the A fragment of row tile k + 1 is requested (offset relative to v73);
    red_prev: the row tile before is reduced beside this one (its row-store offset: red_off)"""
    G = row_tile(nct, k)
    slots = len(G) - 1                            # the steps after the init step: where the reduction's stages go
    narrow = slots >= 6                           # four values at a time (7 stages); else all 16 at once (3 round trips)
    for i, g in enumerate(G):
        extra = None
        red_n = 0
        if i == 0 and red_prev:
            # the row tile before: its minima are final once its last group is folded, i.e. after this step's minima
            pipe.red = [X[(k - 1) & 1], [], red_off]
        if i == 1:
            extra = (lambda: pipe.load_a(a_reg_next, a_offset_next))
            if pipe.red is not None:
                st = pipe.red_stages(pipe.red[0], pipe.red[2], narrow)
                if not (narrow and slots >= 7):
                    # the write and the first read need no wait between them (one wave's LDS operations execute in order)
                    st = [(False, lambda a=st[0][1], b=st[1][1]: (a(), b()))] + st[2:]
                pipe.red[1] = st
        if i >= 1 and pipe.red is not None:
            left, slots_left = len(pipe.red[1]), len(G) - i
            red_n = (left + slots_left - 1) // slots_left
        if i == 0 and first:
            # the candidate's other B fragments go out behind the first MFMA's operands (the init tile's came with the prologue)
            extra = (lambda: pipe.load_b(range(nct - 1)))
        pipe.step(g, extra, red_n)
    assert pipe.red is None, "the reduction must be over before the next row tile starts"


def snapshot(s, pipe):
    return copy.deepcopy((s.n, s.mfma_at, s.valu_at, s.loading, pipe.pending, pipe.red, pipe.busy))


def restore(s, pipe, snap):
    s.n, s.mfma_at, s.valu_at, s.loading, pipe.pending, pipe.red, pipe.busy = copy.deepcopy(snap)


def merge_snapshots(a, b):
    """The state behind a join of two paths: every hazard at the distance of the path that is CLOSER to it (the code
    generated from it pads for the worse of the two and is valid on both); the pipeline's own bookkeeping is taken from
    `a` (the caller generates from merge(a, b) and merge(b, a) and requires the same text)."""
    (na, ma, va, la, *ra), (nb, mb, vb, lb, *rb) = a, b
    n = max(na, nb)

    def closest(xa, xb):
        out = {}
        for r in set(xa) | set(xb):
            d = min(na - xa[r] if r in xa else 1 << 30, nb - xb[r] if r in xb else 1 << 30)
            out[r] = n - d
        return out
    return copy.deepcopy((n, closest(ma, mb), closest(va, vb), la | lb, *ra))


def column_final(s, nct, emit=False):
    """lanes l and l + 32 hold different rows of column l & 31.  The swap leaves the lower halves of two column tiles in one
    register and the upper halves in the other; their minimum is complete for both tiles.
    emit: the complete column minima also go to a column store in LDS (%10: this lane's dword of its first 64 columns) --
    lane l of pair p holds column 64 p + l, 256 bytes per pair; an odd last tile holds its 32 columns in both halves."""
    done = []
    for p in range(nct // 2):
        a, b = CM + 2 * p, CM + 2 * p + 1
        s.ins(f"v_permlane32_swap_b32 v{a}, v{b}", reads=[a, b], writes=[a, b], permlane=True)
        s.ins(f"v_min_i32 v{a}, v{a}, v{b}", reads=[a, b], writes=[a], valu=True)
        if emit:
            s.ins(f"ds_write_b32 %10, v{a} offset:{256 * p}", reads=[a])
        done.append(a)
    if nct & 1:
        a = CM + nct - 1
        s.ins(f"v_mov_b32 v{T}, v{a}", reads=[a], writes=[T], valu=True)
        s.ins(f"v_permlane32_swap_b32 v{a}, v{T}", reads=[a, T], writes=[a, T], permlane=True)
        s.ins(f"v_min_i32 v{a}, v{a}, v{T}", reads=[a, T], writes=[a], valu=True)
        if emit:
            s.ins(f"ds_write_b32 %10, v{a} offset:{256 * (nct // 2)}", reads=[a])
        done.append(a)
    while len(done) > 1:
        nxt = []
        for i in range(0, len(done), 3):
            c = done[i:i + 3]
            if len(c) == 3:
                s.ins(f"v_max3_i32 v{c[0]}, v{c[0]}, v{c[1]}, v{c[2]}", reads=c, writes=[c[0]], valu=True)
            elif len(c) == 2:
                s.ins(f"v_max_i32 v{c[0]}, v{c[0]}, v{c[1]}", reads=c, writes=[c[0]], valu=True)
            nxt.append(c[0])
        done = nxt
    s.ins(f"v_max_i32 %0, v{done[0]}, v{ROWMAX}", reads=[done[0], ROWMAX])


def generate(nct, carry, emit=False):
    assert carry or not emit
    s = Stream()
    pipe = Pipe(s, nct, carry)
    # ---- prologue: row tile 0 -------------------------------------------------------------------------------------------
    s.ins(f"v_mov_b32 v{AADDR}, %3", writes=[AADDR], valu=True)
    s.ins(f"v_add_u32 v{RW2}, {8 * RED_STRIDE}, %4", writes=[RW2], valu=True)
    if carry:
        s.ins(f"v_mov_b32 v{RSADDR}, %9", writes=[RSADDR], valu=True)
    s.ins(f"ds_read_b128 v[{ASET[0]}:{ASET[0] + 3}], %3 offset:0", writes=rng(ASET[0], 4), lds_load=True)
    pipe.load_b([nct - 1])
    for ct in range(nct):
        s.ins(f"v_mov_b32 v{CM + ct}, {INF}", writes=[CM + ct], valu=True)
    s.ins(f"v_mov_b32 v{ROWMAX}, 0", writes=[ROWMAX], valu=True)
    run_row_tile(pipe, nct, 0, 1024, ASET[1], False, 0, first=True)
    s.raw("s_mov_b32 %1, %7")
    s.raw("s_cmp_eq_u32 %1, 0")
    s.raw("s_cbranch_scc1 2f")
    after_prologue = snapshot(s, pipe)
    s.label("1:")
    # ---- loop body: row tiles (2i + 1, 2i + 2), v73 = A address of row tile 2i; generated twice, emitted once ------------------
    # The body is entered from the prologue and from its own end: it is generated from the MERGED state of the two (every
    # hazard at the distance of the closer path), until the text no longer changes -- then that text is valid on both edges.
    def body_from(entry):
        restore(s, pipe, entry)
        m0 = s.mark()
        run_row_tile(pipe, nct, 1, 2048, ASET[0], True, 0)
        run_row_tile(pipe, nct, 2, 3072, ASET[1], True, 128)     # (the last iteration requests one fragment too many: read, never used)
        s.ins(f"v_add_u32 v{AADDR}, 2048, v{AADDR}", reads=[AADDR], writes=[AADDR], valu=True)
        if carry:
            s.ins(f"v_add_u32 v{RSADDR}, 256, v{RSADDR}", reads=[RSADDR], writes=[RSADDR], valu=True)
        s.raw("s_sub_u32 %1, %1, 1")
        s.raw("s_cmp_lg_u32 %1, 0")
        s.raw("s_cbranch_scc1 1b")
        text, end = s.out[m0:], snapshot(s, pipe)
        del s.out[m0:]
        return text, end

    body, after_body = body_from(after_prologue)
    for _ in range(8):
        nxt, end = body_from(merge_snapshots(after_prologue, after_body))
        stable = nxt == body
        body, after_body = nxt, end
        if stable:
            break
    else:
        raise AssertionError("the loop body does not settle")
    s.out += body
    s.label("2:")

    # ---- after the loop: odd row-tile count -> epilogue; even -> one more row tile, then the epilogue -----------------------------
    def tail_and_epilogue(entry):
        restore(s, pipe, entry)
        m0 = s.mark()
        s.raw("s_cmp_lg_u32 %8, 0")
        s.raw("s_cbranch_scc1 3f")
        at_branch = snapshot(s, pipe)
        pipe.drain()
        assert pipe.red is None
        pipe.reduce_blocking(X[0], P, Q, 0)
        s.raw("s_branch 4f")
        s.label("3:")
        restore(s, pipe, at_branch)
        run_row_tile(pipe, nct, 1, 2048, ASET[0], True, 0)
        pipe.drain()
        pipe.reduce_blocking(X[1], P, Q, 128)
        s.label("4:")
        # both paths end behind the final instructions of reduce_blocking: the same distances to every hazard below
        column_final(s, nct, emit)
        s.wait()                                       # (the unused B and A requests of the last steps)
        text = s.out[m0:]
        del s.out[m0:]
        return text

    # the code behind the loop is entered from the prologue (no iteration) and from the loop: generated once, from the merged
    # state (padded for whichever path is closer to each MFMA), and checked to be what either state alone would accept
    t_a = tail_and_epilogue(merge_snapshots(after_prologue, after_body))
    t_b = tail_and_epilogue(merge_snapshots(after_body, after_prologue))
    assert t_a == t_b, "the code behind the loop must not depend on whether the loop ran"
    s.out += t_a
    return s.out, sorted(s.used)


# =====================================================================================================================
# The DUAL form (round 4): both directed minima taken IN-LANE, no cross-lane row reduction.
#
# Per 32 x 32 tile TWO MFMAs on the same operands, swapped:  D = A B^T (a lane's 16 values belong to ITS column: column
# minimum, as above) and D' = B A^T (a lane's 16 values belong to ITS row -- lane l & 31 is row l & 31 of the row tile, the
# half picks 16 of the tile's 32 columns -- : row minimum, 8 v_min3_i32 into one register per row tile).  20 issue slots per
# tile instead of 18, but nothing else: no transpose through LDS, no landing zone, no waits except for the row tile's A
# fragment; at a row tile's end the halves of its row-minimum register meet (v_permlane32_swap) and go into the running
# maximum -- four instructions.  tools/ubench_dual.hip: 39.2 ns per tile per SIMD at two waves against 35.7 for the bare
# loop of the single form -- whose real kernel runs 42 - 43 at 17 x 17 tiles, 51 at 7 x 7 and 67 at 4 x 4 because of its row
# reduction (tools/exp_mx.sh).  Four 16-register result buffers (D, D' of the tile in flight and of the tile being folded).
# The matrix pipe is busy 64 of a tile's 80 cycles.
# Register map: as above up to v75; v70, v71 row-minimum registers of the even / odd row tiles; v75, v77 carry: the stored
# minimum read back (even / odd row tile); v[84:147] the four result buffers.  Operands as above (%4, %5, %6 unused).
# =====================================================================================================================
DD = [[84, 100], [116, 132]]        # [tile parity][0: D, 1: D']
RMIN = [70, 71]
TPD = [75, 77]


class Dual:
    def __init__(self, s, nct, carry):
        self.s, self.nct, self.carry = s, nct, carry
        self.pending = None          # (column tile, buffer parity, row-tile parity, is the row tile's last tile)
        self.par = 0

    def load_b(self, tiles):
        for t in tiles:
            d = AG + 4 * t
            self.s.ins(f"ds_read_b64 a[{4 * t}:{4 * t + 1}], %2 offset:{t * 512}", writes=rng(d, 2), lds_load=True)
            self.s.ins(f"ds_read_b64 a[{4 * t + 2}:{4 * t + 3}], %2 offset:{t * 512}", writes=rng(d + 2, 2), lds_load=True)

    def fold(self, which):
        """8 minima of the pending tile: which = 0 its column minimum (D), 1 its row minimum (D').  Five of them fold the
        buffer into itself, independent of each other; three bring the five partial minima and the last value into the
        accumulator (eight in a row on ONE accumulator, each waiting for the one before, cost 9 % of the loop:
        tools/exp_mx3.sh, variant X1 against the single form without its reduction)."""
        t, par, q, _last = self.pending
        d = DD[par][which]
        acc = CM + t if which == 0 else RMIN[q]
        s = self.s
        for i in range(5):
            s.ins(f"v_min3_i32 v{d + 3 * i}, v{d + 3 * i}, v{d + 3 * i + 1}, v{d + 3 * i + 2}", reads=rng(d + 3 * i, 3), writes=[d + 3 * i], valu=True)
        for a, b in ((d, d + 3), (d + 6, d + 9), (d + 12, d + 15)):
            s.ins(f"v_min3_i32 v{acc}, v{acc}, v{a}, v{b}", reads=[acc, a, b], writes=[acc], valu=True)

    def finish_row_tile(self, q, blocking=False):
        """the halves of row tile q's minimum register meet; the row's minimum goes into the running maximum (carry: meets the
        stored minimum of the blocks before and goes back to the row store -- v74 is the slot of the even row tile that is
        being, or was last, finished: an odd one's is 128 bytes on)"""
        s = self.s
        r, off = RMIN[q], 128 * q
        s.ins(f"v_mov_b32 v{T}, v{r}", reads=[r], writes=[T], valu=True)
        s.ins(f"v_permlane32_swap_b32 v{r}, v{T}", reads=[r, T], writes=[r, T], permlane=True)
        if self.carry:
            if blocking and TPD[q] in s.loading:
                s.wait()
            s.ins(f"v_min3_i32 v{r}, v{r}, v{T}, v{TPD[q]}", reads=[r, T, TPD[q]], writes=[r], valu=True)
            s.ins(f"ds_write_b32 v{RSADDR}, v{r} offset:{off}", reads=[RSADDR, r])
        else:
            s.ins(f"v_min_i32 v{r}, v{r}, v{T}", reads=[r, T], writes=[r], valu=True)
            s.ins(f"v_max_i32 v{ROWMAX}, v{ROWMAX}, v{r}", reads=[ROWMAX, r], writes=[ROWMAX], valu=True)
        s.ins(f"v_mov_b32 v{r}, {INF}", writes=[r], valu=True)

    def step(self, a_reg, t, q, last, extra=None):
        """issue the two MFMAs of (row tile of parity q, column tile t); beside them the 8 + 8 minima of the tile before"""
        s = self.s
        if s.loading:
            s.wait()
        if extra is not None:
            extra()
        par, self.par = self.par, self.par ^ 1
        d, dt = DD[par]
        prev = self.pending
        s.ins(f"v_mfma_f32_32x32x16_f16 v[{d}:{d + 15}], v[{a_reg}:{a_reg + 3}], a[{4 * t}:{4 * t + 3}], 0",
              reads=rng(a_reg, 4) + rng(AG + 4 * t, 4), writes=rng(d, 16), mfma=True)
        if prev is not None:
            self.fold(0)
        s.ins(f"v_mfma_f32_32x32x16_f16 v[{dt}:{dt + 15}], a[{4 * t}:{4 * t + 3}], v[{a_reg}:{a_reg + 3}], 0",
              reads=rng(a_reg, 4) + rng(AG + 4 * t, 4), writes=rng(dt, 16), mfma=True)
        if prev is not None:
            self.fold(1)
            if prev[3]:
                self.finish_row_tile(prev[2])
        self.pending = (t, par, q, last)

    def drain(self):
        self.fold(0)
        self.fold(1)
        _t, _par, q, last = self.pending
        assert last
        self.finish_row_tile(q, blocking=True)
        self.pending = None


def dual_row_tile(pipe, nct, q, a_offset_next, a_reg_next, off, first=False):
    """the steps of a row tile of parity q (its A fragment is in ASET[q]); in its first step the A fragment of the next row
    tile is requested (offset relative to v73) and -- carry -- its own stored minimum (off: its slot relative to v74 NOW)"""
    s = pipe.s
    for t in range(nct):
        extra = None
        if t == 0:
            def extra(q=q):
                s.ins(f"ds_read_b128 v[{a_reg_next}:{a_reg_next + 3}], v{AADDR} offset:{a_offset_next}", reads=[AADDR], writes=rng(a_reg_next, 4), lds_load=True)
                if pipe.carry:
                    s.ins(f"ds_read_b32 v{TPD[q]}, v{RSADDR} offset:{off}", reads=[RSADDR], writes=[TPD[q]], lds_load=True)
                if first:
                    pipe.load_b(range(1, nct))
        pipe.step(ASET[q], t, q, t == nct - 1, extra)


def dual_snapshot(s, pipe):
    return copy.deepcopy((s.n, s.mfma_at, s.valu_at, s.loading, pipe.pending, pipe.par))


def dual_restore(s, pipe, snap):
    s.n, s.mfma_at, s.valu_at, s.loading, pipe.pending, pipe.par = copy.deepcopy(snap)


def generate_dual(nct, carry, emit=False):
    assert carry or not emit
    s = Stream()
    pipe = Dual(s, nct, carry)
    s.ins(f"v_mov_b32 v{AADDR}, %3", writes=[AADDR], valu=True)
    if carry:
        s.ins(f"v_mov_b32 v{RSADDR}, %9", writes=[RSADDR], valu=True)
    s.ins(f"ds_read_b128 v[{ASET[0]}:{ASET[0] + 3}], %3 offset:0", writes=rng(ASET[0], 4), lds_load=True)
    pipe.load_b([0])
    for ct in range(nct):
        s.ins(f"v_mov_b32 v{CM + ct}, {INF}", writes=[CM + ct], valu=True)
    s.ins(f"v_mov_b32 v{ROWMAX}, 0", writes=[ROWMAX], valu=True)
    for r in RMIN:
        s.ins(f"v_mov_b32 v{r}, {INF}", writes=[r], valu=True)
    dual_row_tile(pipe, nct, 0, 1024, ASET[1], 0, first=True)
    s.raw("s_mov_b32 %1, %7")
    s.raw("s_cmp_eq_u32 %1, 0")
    s.raw("s_cbranch_scc1 2f")
    after_prologue = dual_snapshot(s, pipe)
    s.label("1:")

    def body_from(entry):
        dual_restore(s, pipe, entry)
        m0 = s.mark()
        dual_row_tile(pipe, nct, 1, 2048, ASET[0], 128)
        # (the even row tile's slot is 256 bytes on when its stored minimum is requested; v74 has advanced by the time the
        # tile is finished -- in the next iteration's, or the tail's, first step)
        dual_row_tile(pipe, nct, 0, 3072, ASET[1], 256)
        s.ins(f"v_add_u32 v{AADDR}, 2048, v{AADDR}", reads=[AADDR], writes=[AADDR], valu=True)
        if carry:
            s.ins(f"v_add_u32 v{RSADDR}, 256, v{RSADDR}", reads=[RSADDR], writes=[RSADDR], valu=True)
        s.raw("s_sub_u32 %1, %1, 1")
        s.raw("s_cmp_lg_u32 %1, 0")
        s.raw("s_cbranch_scc1 1b")
        text, end = s.out[m0:], dual_snapshot(s, pipe)
        del s.out[m0:]
        return text, end

    body, after_body = body_from(after_prologue)
    for _ in range(8):
        nxt, end = body_from(merge_snapshots(after_prologue, after_body))
        stable = nxt == body
        body, after_body = nxt, end
        if stable:
            break
    else:
        raise AssertionError("the loop body does not settle")
    s.out += body
    s.label("2:")

    def tail_and_epilogue(entry):
        dual_restore(s, pipe, entry)
        m0 = s.mark()
        s.raw("s_cmp_lg_u32 %8, 0")
        s.raw("s_cbranch_scc1 3f")
        at_branch = dual_snapshot(s, pipe)
        pipe.drain()
        a_end = dual_snapshot(s, pipe)
        s.raw("s_branch 4f")
        s.label("3:")
        dual_restore(s, pipe, at_branch)
        dual_row_tile(pipe, nct, 1, 2048, ASET[0], 128)
        pipe.drain()
        b_end = dual_snapshot(s, pipe)
        s.label("4:")
        dual_restore(s, pipe, merge_snapshots(a_end, b_end))
        column_final(s, nct, emit)
        s.wait()                                       # (the unused A request of the last row tile)
        text = s.out[m0:]
        del s.out[m0:]
        return text

    t_a = tail_and_epilogue(merge_snapshots(after_prologue, after_body))
    t_b = tail_and_epilogue(merge_snapshots(after_body, after_prologue))
    assert t_a == t_b, "the code behind the loop must not depend on whether the loop ran"
    s.out += t_a
    return s.out, sorted(s.used)


def generate_bound(nb):
    """The bound kernel's pass (k_bound_mx, mm_kernels.hip): `nb` column tiles of queries held in registers (operands
    %4 .., the B fragments: query tiles of one candidate and / or of several candidates) against the row tiles of a set in
    LDS.  Per row tile ONE ds_read_b128 of the row fragment feeds nb MFMAs, and every MFMA result is folded by 8
    v_min3_i32 into two running column minima (two chains; a lane's 16 values all belong to its query).
    Software pipeline: two sets of result buffers -- the MFMAs of the next row tile are issued ahead of the minima of the
    tile before -- and FOUR row-fragment registers: the fragments of tiles c + 3, c + 4 are requested while c + 1, c + 2 are
    multiplied, so that an LDS round trip has half a loop body (two tiles) to complete behind.  (The first version requested a
    fragment and waited for it on the spot: 295 clocks per tile, most of them LDS latency.)  The loop body is four tiles; what
    is left (0 .. 3 tiles) has its own code.  Requests may run up to two tiles past the set (read, never multiplied).
    The distance between an MFMA and the first read of its result is padded where the instructions in between are not
    enough.  The compiler cannot be trusted with this: it places the minima of a tile right behind that tile's MFMA without
    a wait state (inline asm operands are invisible to its MFMA hazard recogniser), and the hardware does not interlock --
    the minima then fold stale registers.
    Operands: %0 .. %(nb-1) out: min over all row tiles of this lane's column (its half of the rows), per B fragment;
    %nb =s counter; then LDS byte address of this lane's fragment in row tile 0 (the next ones 1024 bytes apart); s: row
    tiles - 1; the nb B fragments (4 VGPRs each)."""
    s = Stream()
    o_cnt, o_addr, o_nt, o_b = nb, nb + 1, nb + 2, nb + 3
    R, ADDR = [60, 64, 68, 76], AADDR                      # four row-fragment registers (v72 .. v75 belong to the screen's blocks)
    D = [[P + 16 * (2 * j), P + 16 * (2 * j + 1)] for j in range(nb)]        # D[j][k]: result buffer k of fragment j
    C = [[CM + 2 * j, CM + 2 * j + 1] for j in range(nb)]

    def mfmas(k, r):
        for j in range(nb):
            d = D[j][k]
            s.ins(f"v_mfma_f32_32x32x16_f16 v[{d}:{d + 15}], v[{R[r]}:{R[r] + 3}], %{o_b + j}, 0", reads=rng(R[r], 4), writes=rng(d, 16), mfma=True)

    def folds(k):
        for q in range(8):              # interleaved over the fragments: no instruction depends on the one before
            for j in range(nb):
                c, d = C[j][q & 1], D[j][k]
                s.ins(f"v_min3_i32 v{c}, v{c}, v{d + 2 * q}, v{d + 2 * q + 1}", reads=[c, d + 2 * q, d + 2 * q + 1], writes=[c], valu=True)

    def load(r, off):
        s.ins(f"ds_read_b128 v[{R[r]}:{R[r] + 3}], v{ADDR} offset:{off}", reads=[ADDR], writes=rng(R[r], 4), lds_load=True)

    def state():
        return copy.deepcopy((s.n, s.mfma_at, s.valu_at, s.loading))

    def set_state(x):
        s.n, s.mfma_at, s.valu_at, s.loading = copy.deepcopy(x)

    strip = lambda b: [x for x in b if not x.startswith("s_nop")]

    def merge_pad(x, y):
        """two versions of the same instruction sequence that differ in their s_nop padding: the larger padding at every place"""
        assert strip(x) == strip(y)
        out, i, j = [], 0, 0
        while i < len(x) or j < len(y):
            px = py = -1
            if i < len(x) and x[i].startswith("s_nop"):
                px = int(x[i].split()[1]); i += 1
            if j < len(y) and y[j].startswith("s_nop"):
                py = int(y[j].split()[1]); j += 1
            if max(px, py) >= 0:
                out.append(f"s_nop {max(px, py)}")
            if i < len(x):
                assert x[i] == y[j]
                out.append(x[i]); i += 1; j += 1
        return out

    # ---- prologue: tiles 0, 1, 2 requested; tile 0 multiplied.  Invariant at the loop top (c = 0): D[.][0] = tile c, R1 = tile
    # c + 1, R2 = tile c + 2 (requested), v73 = address of tile c, counter = tiles after c
    s.ins(f"v_mov_b32 v{ADDR}, %{o_addr}", writes=[ADDR], valu=True)
    load(0, 0); load(1, 1024); load(2, 2048)
    for j in range(nb):
        for c in C[j]:
            s.ins(f"v_mov_b32 v{c}, {INF}", writes=[c], valu=True)
    s.raw(f"s_mov_b32 %{o_cnt}, %{o_nt}")
    s.wait()
    mfmas(0, 0)
    s.raw(f"s_cmp_lt_u32 %{o_cnt}, 4")
    s.raw("s_cbranch_scc1 2f")
    entry = state()
    s.label("1:")
    bodies = []
    for it in range(2):
        m0 = s.mark()
        s.wait()                                  # R1, R2: requested half a body ago
        load(3, 3072); load(0, 4096)
        mfmas(1, 1); folds(0); mfmas(0, 2); folds(1)
        s.wait()                                  # R3, R0
        load(1, 5120); load(2, 6144)
        mfmas(1, 3); folds(0); mfmas(0, 0); folds(1)
        s.ins(f"v_add_u32 v{ADDR}, 4096, v{ADDR}", reads=[ADDR], writes=[ADDR], valu=True)
        s.raw(f"s_sub_u32 %{o_cnt}, %{o_cnt}, 4")
        s.raw(f"s_cmp_gt_u32 %{o_cnt}, 3")
        s.raw("s_cbranch_scc1 1b")
        bodies.append(s.out[m0:])
        if it == 0:
            keep = s.mark()
            after = state()
    # the first iteration comes right behind the prologue's MFMAs and may need more padding; the steady state is safe with it
    # too (more wait states never hurt) -- tests/test_screen_mx_asm.py executes the text for every trip count
    del s.out[keep:]
    s.out[keep - len(bodies[0]):keep] = merge_pad(bodies[0], bodies[1])
    s.label("2:")

    def tails(st):
        """0 .. 3 tiles after tile c: D0 = tile c, R1 = c + 1, R2 = c + 2 requested"""
        set_state(st)
        m0 = s.mark()
        s.raw(f"s_cmp_eq_u32 %{o_cnt}, 0")
        s.raw("s_cbranch_scc1 5f")
        at5 = state()                             # (a taken branch leaves here: the wait states behind it do not count)
        s.raw(f"s_cmp_eq_u32 %{o_cnt}, 1")
        s.raw("s_cbranch_scc1 6f")
        at6 = state()
        s.raw(f"s_cmp_eq_u32 %{o_cnt}, 2")
        s.raw("s_cbranch_scc1 7f")
        at7 = state()
        # three left
        s.wait()
        load(3, 3072)
        mfmas(1, 1); folds(0); mfmas(0, 2); folds(1)
        s.wait()
        mfmas(1, 3); folds(0); folds(1)
        s.raw("s_branch 8f")
        s.label("7:")                             # two left
        set_state(at7)
        s.wait()
        mfmas(1, 1); folds(0); mfmas(0, 2); folds(1); folds(0)
        s.raw("s_branch 8f")
        s.label("6:")                             # one left
        set_state(at6)
        s.wait()
        mfmas(1, 1); folds(0); folds(1)
        s.raw("s_branch 8f")
        s.label("5:")                             # none left
        set_state(at5)
        folds(0)
        s.wait()                                  # (the requests that ran past the set)
        s.label("8:")
        for j in range(nb):
            s.ins(f"v_min_i32 %{j}, v{C[j][0]}, v{C[j][1]}", reads=C[j])
        t = s.out[m0:]
        del s.out[m0:]
        return t

    s.out += merge_pad(tails(entry), tails(after))          # the larger padding at every place serves either entry
    return s.out, sorted(s.used)


BOUND_NB = (1, 2, 4)


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    dst = os.environ.get("MX_OUT") or os.path.join(here, "..", "multimoda-rs_amd", "csrc", "mm_screen_mx_asm.inc")
    total = 0
    with open(dst, "w") as f:
        f.write("// GENERATED by tools/gen_screen_mx.py -- do not edit.  Main phase of k_screen_mx: see the generator's docstring.\n")
        f.write(f"#define MM_SCREEN_MX_RED_STRIDE {RED_STRIDE}\n")
        f.write(f"#define MM_SCREEN_MX_NCT_MIN {NCT_MIN}\n#define MM_SCREEN_MX_NCT_MAX {NCT_MAX}\n")
        for carry in (False, True):
            for nct in range(NCT_MIN, NCT_MAX + 1):
                out, regs = generate(nct, carry)
                total += len(out)
                name = f"{nct}{'C' if carry else ''}"
                f.write(f"// ---- nct = {nct}, {'carry' if carry else 'plain'}: {len(out)} instructions\n")
                f.write(f"#define MM_SCREEN_MX_ASM_{name} \\\n")
                for line in out:
                    f.write(f'    "{line}\\n" \\\n')
                f.write('    ""\n')
                f.write(f"#define MM_SCREEN_MX_CLOBBERS_{name} " + ", ".join(regname(r) for r in regs) + ', "scc", "memory"\n')
        for nct in range(NCT_MIN, NCT_MAX + 1):
            out, regs = generate(nct, True, True)
            total += len(out)
            f.write(f"// ---- nct = {nct}, carry + column store (the pick of a bounded search leaves its row and column minima): {len(out)} instructions\n")
            f.write(f"#define MM_SCREEN_MX_ASM_{nct}E \\\n")
            for line in out:
                f.write(f'    "{line}\\n" \\\n')
            f.write('    ""\n')
            f.write(f"#define MM_SCREEN_MX_CLOBBERS_{nct}E " + ", ".join(regname(r) for r in regs) + ', "scc", "memory"\n')
        for carry, emit, suffix, what in ((False, False, "", "plain"), (True, False, "C", "carry"), (True, True, "E", "carry + column store")):
            for nct in range(NCT_MIN, NCT_MAX + 1):
                out, regs = generate_dual(nct, carry, emit)
                total += len(out)
                f.write(f"// ---- DUAL form, nct = {nct}, {what}: {len(out)} instructions\n")
                f.write(f"#define MM_SCREEN_MX_ASM_D{nct}{suffix} \\\n")
                for line in out:
                    f.write(f'    "{line}\\n" \\\n')
                f.write('    ""\n')
                f.write(f"#define MM_SCREEN_MX_CLOBBERS_D{nct}{suffix} " + ", ".join(regname(r) for r in regs) + ', "scc", "memory"\n')
        for nb in BOUND_NB:
            out, regs = generate_bound(nb)
            total += len(out)
            f.write(f"// ---- the bound kernel's pass, {nb} B fragment(s) per row fragment: {len(out)} instructions\n#define MM_BOUND_MX_ASM_{nb} \\\n")
            for line in out:
                f.write(f'    "{line}\\n" \\\n')
            f.write('    ""\n')
            f.write(f"#define MM_BOUND_MX_CLOBBERS_{nb} " + ", ".join(regname(r) for r in regs) + ', "scc", "memory"\n')
        # the dispatch: one inline function per block, selected at compile time
        f.write("\n#ifdef __HIPCC__\n")
        f.write("template <int NCT, bool CARRY> struct MxMain;\n")
        for carry in (False, True):
            for nct in range(NCT_MIN, NCT_MAX + 1):
                name = f"{nct}{'C' if carry else ''}"
                f.write(f"template <> struct MxMain<{nct}, {'true' if carry else 'false'}> {{\n"
                        "    static __device__ __forceinline__ int run(unsigned vB, unsigned vA, unsigned vRW, unsigned vRR, unsigned vPERM,\n"
                        "                                              int nloop, int tail, unsigned vRS)\n    {\n"
                        "        int m, counter;\n"
                        f"        asm volatile(MM_SCREEN_MX_ASM_{name}\n"
                        "                     : \"=&v\"(m), \"=&s\"(counter)\n"
                        "                     : \"v\"(vB), \"v\"(vA), \"v\"(vRW), \"v\"(vRR), \"s\"(0), \"s\"(nloop), \"s\"(tail), \"v\"(vRS)\n"
                        f"                     : MM_SCREEN_MX_CLOBBERS_{name});\n"
                        "        return m;\n    }\n};\n")
        f.write("template <int NCT> struct MxEmit;\n")
        for nct in range(NCT_MIN, NCT_MAX + 1):
            f.write(f"template <> struct MxEmit<{nct}> {{\n"
                    "    static __device__ __forceinline__ int run(unsigned vB, unsigned vA, unsigned vRW, unsigned vRR, unsigned vPERM,\n"
                    "                                              int nloop, int tail, unsigned vRS, unsigned vCS)\n    {\n"
                    "        int m, counter;\n"
                    f"        asm volatile(MM_SCREEN_MX_ASM_{nct}E\n"
                    "                     : \"=&v\"(m), \"=&s\"(counter)\n"
                    "                     : \"v\"(vB), \"v\"(vA), \"v\"(vRW), \"v\"(vRR), \"s\"(0), \"s\"(nloop), \"s\"(tail), \"v\"(vRS), \"v\"(vCS)\n"
                    f"                     : MM_SCREEN_MX_CLOBBERS_{nct}E);\n"
                    "        return m;\n    }\n};\n")
        f.write("template <int NCT, bool CARRY> struct MxMainD;\ntemplate <int NCT> struct MxEmitD;\n")
        for carry in (False, True):
            for nct in range(NCT_MIN, NCT_MAX + 1):
                name = f"D{nct}{'C' if carry else ''}"
                f.write(f"template <> struct MxMainD<{nct}, {'true' if carry else 'false'}> {{\n"
                        "    static __device__ __forceinline__ int run(unsigned vB, unsigned vA, int nloop, int tail, unsigned vRS)\n    {\n"
                        "        int m, counter;\n"
                        f"        asm volatile(MM_SCREEN_MX_ASM_{name}\n"
                        "                     : \"=&v\"(m), \"=&s\"(counter)\n"
                        "                     : \"v\"(vB), \"v\"(vA), \"s\"(0), \"s\"(0), \"s\"(0), \"s\"(nloop), \"s\"(tail), \"v\"(vRS)\n"
                        f"                     : MM_SCREEN_MX_CLOBBERS_{name});\n"
                        "        return m;\n    }\n};\n")
        for nct in range(NCT_MIN, NCT_MAX + 1):
            f.write(f"template <> struct MxEmitD<{nct}> {{\n"
                    "    static __device__ __forceinline__ int run(unsigned vB, unsigned vA, int nloop, int tail, unsigned vRS, unsigned vCS)\n    {\n"
                    "        int m, counter;\n"
                    f"        asm volatile(MM_SCREEN_MX_ASM_D{nct}E\n"
                    "                     : \"=&v\"(m), \"=&s\"(counter)\n"
                    "                     : \"v\"(vB), \"v\"(vA), \"s\"(0), \"s\"(0), \"s\"(0), \"s\"(nloop), \"s\"(tail), \"v\"(vRS), \"v\"(vCS)\n"
                    f"                     : MM_SCREEN_MX_CLOBBERS_D{nct}E);\n"
                    "        return m;\n    }\n};\n")
        f.write("#endif\n")
    print(total, "instructions in", 6 * (NCT_MAX - NCT_MIN + 1), "blocks ->", os.path.normpath(dst))


if __name__ == "__main__":
    main()
