#!/usr/bin/env python3
"""Generates multimoda-rs_amd/csrc/mm_screen_mx_asm.inc: the main phase of k_screen_mx (mm_kernels.hip) as ONE asm block on
fixed registers, so that the order is exactly the software pipeline we want -- two MFMAs (1024 squared distances each, on
the matrix pipe) issued ahead of the 32 v_min3_i32 that fold the PREVIOUS two tiles on the vector pipe.  The compiler's
scheduler does not produce this order (tools/ubench_mfma16*.hip: 46 ns per tile compiler-scheduled, 34 ns hand-ordered).

One wave, one candidate: its share of the 17th row tile (4 or 5 column tiles, chosen by the scalar operand `variant`) and
its four full row tiles (17 column tiles each), so that the four waves of a workgroup carry 72 or 73 tiles each.
The cross-lane reduction of a row tile's minima (LDS transpose: write, read back a row per lane, fold, meet the other
half through ds_bpermute) is spread over the phases of the NEXT row tile, one LDS round trip per phase, each behind a
wait the pipeline has anyway.
Register map (VGPR):
  v[20:35]   rmin[16]   running row minima of the current row tile (element v of the 32x32 result layout)
  v36        rowmax     max over rows of the row minima (signed-int order on f32 bits, floored at 0)
  v[40:56]   cm[17]     running column minima per column tile (this lane's column, this wave's rows)
  v[60:63], v[64:67]    A operand fragment of the current / the next row tile
  v[68:71], v72         reduction: four values read back, accumulator
  v[80:87], v[88:95]    B operand fragments, two pairs (double buffer)
  v[100:163] D0..D3     four 32x32 result tiles
Operands: %0 out rowmax; %1 vB (LDS byte address of this lane's B fragment in column tile 0); %2 vA (A fragment of the
wave's first row tile; the next ones 4096 bytes apart); %3 vA16 (A fragment of row tile 16); %4 vRW / %5 vRR
(row-reduction scratch: write / read address); %6 vCM (column-minimum array, this lane's column of tile 0); %7 vPERM
(4 * (lane ^ 32)); %8 vR16 (row minima of row tile 16, shared by the four waves: this lane's row); %9 s variant (0..3)."""
import os

INF = "0x7f800000"
RMIN, ROWMAX, CM, ASET, T, ACC, BSET, D = 20, 36, 40, [60, 64], 68, 72, [80, 88], [100, 116, 132, 148]
NCT = 17
RED_STRIDE = 136          # bytes between rows of the reduction scratch (34 dwords: 8-byte aligned reads, 2-way conflicts)
PART = [(0, 5), (5, 9), (9, 13), (13, 17)]          # column tiles of row tile 16 per variant

out = []
emit = out.append
DBG_NOP = int(os.environ.get("MX_DBG_NOP", "0"))           # s_nop 15 count in front of every group of minima
DBG_BLOCKING = os.environ.get("MX_DBG_BLOCKING") == "1"    # no staged reductions


def mfma(d, a, b):
    emit(f"v_mfma_f32_32x32x16_f16 v[{d}:{d + 15}], v[{a}:{a + 3}], v[{b}:{b + 3}], 0")


def load_b(dst, tile):
    # a column fragment is stored as 8 bytes per lane -- (x1, x2, y1, y2) or (256, 1, n2h, n2l) -- and read twice: the K
    # slots 4..7 repeat slots 0..3 (the row fragments are laid out to match; for the norm half their slots 4..7 are 0)
    emit(f"ds_read_b64 v[{dst}:{dst + 1}], %1 offset:{tile * 512}")
    emit(f"ds_read_b64 v[{dst + 2}:{dst + 3}], %1 offset:{tile * 512}")


def mins_pair(dA, dB, cmA, cmB, first):
    """32 minima as a list: column folds of two tiles (8 v_min3_i32 each) and the elementwise row minima over both (16;
    the first pair of a row tile writes rmin, the others fold into it), interleaved so that no instruction depends on the
    one before."""
    col = []
    for q in range(8):
        col.append(f"v_min3_i32 v{cmA}, v{cmA}, v{dA + 2 * q}, v{dA + 2 * q + 1}")
        col.append(f"v_min3_i32 v{cmB}, v{cmB}, v{dB + 2 * q}, v{dB + 2 * q + 1}")
    if first:
        row = [f"v_min_i32 v{RMIN + v}, v{dA + v}, v{dB + v}" for v in range(16)]
    else:
        row = [f"v_min3_i32 v{RMIN + v}, v{RMIN + v}, v{dA + v}, v{dB + v}" for v in range(16)]
    L = []
    for i in range(16):
        L.append(col[i])
        L.append(row[i])
    return L


def mins_single(d, cm):
    col = [f"v_min3_i32 v{cm}, v{cm}, v{d + 2 * q}, v{d + 2 * q + 1}" for q in range(8)]
    row = [f"v_min_i32 v{RMIN + v}, v{RMIN + v}, v{d + v}" for v in range(16)]
    for i in range(8):
        emit(col[i])
        emit(row[2 * i])
        emit(row[2 * i + 1])


# ---- the reduction of a finished row tile, in stages -------------------------------------------------------------
def red_write():
    for v in range(16):
        emit(f"ds_write_b32 %4, v{RMIN + v} offset:{v * RED_STRIDE}")


def red_read(chunk):
    emit(f"ds_read_b64 v[{T}:{T + 1}], %5 offset:{16 * chunk}")
    emit(f"ds_read_b64 v[{T + 2}:{T + 3}], %5 offset:{16 * chunk + 8}")


def red_fold(chunk):
    if chunk == 0:
        emit(f"v_min3_i32 v{ACC}, v{T}, v{T + 1}, v{T + 2}")
        emit(f"v_min_i32 v{ACC}, v{ACC}, v{T + 3}")
    else:
        emit(f"v_min3_i32 v{ACC}, v{ACC}, v{T}, v{T + 1}")
        emit(f"v_min3_i32 v{ACC}, v{ACC}, v{T + 2}, v{T + 3}")


def red_perm():
    emit(f"ds_bpermute_b32 v{T}, %7, v{ACC}")


def red_final(shared):
    emit(f"v_min_i32 v{ACC}, v{ACC}, v{T}")
    if shared:
        emit(f"ds_min_i32 %8, v{ACC}")
    else:
        emit(f"v_max_i32 v{ROWMAX}, v{ROWMAX}, v{ACC}")


# stage s of the reduction of the PREVIOUS row tile, issued in phase s of the current one (after that phase's wait)
RED_STAGES = {
    0: lambda shared: red_write(),
    1: lambda shared: red_read(0),
    2: lambda shared: (red_fold(0), red_read(1)),
    3: lambda shared: (red_fold(1), red_read(2)),
    4: lambda shared: (red_fold(2), red_read(3)),
    5: lambda shared: (red_fold(3), red_perm()),
    6: lambda shared: red_final(shared),
}


def reduction_blocking(shared):
    """The last row tile of the candidate: nothing left to hide it behind.  All 16 values of a row are read back at once
    (the result tiles are free: D2 as the landing zone), three LDS round trips in all."""
    t = D[2]
    red_write()
    emit("s_waitcnt lgkmcnt(0)")
    for q in range(8):
        emit(f"ds_read_b64 v[{t + 2 * q}:{t + 2 * q + 1}], %5 offset:{8 * q}")
    emit("s_waitcnt lgkmcnt(0)")
    u = D[3]
    emit(f"v_min3_i32 v{u}, v{t}, v{t + 1}, v{t + 2}")
    emit(f"v_min3_i32 v{u + 1}, v{t + 3}, v{t + 4}, v{t + 5}")
    emit(f"v_min3_i32 v{u + 2}, v{t + 6}, v{t + 7}, v{t + 8}")
    emit(f"v_min3_i32 v{u + 3}, v{t + 9}, v{t + 10}, v{t + 11}")
    emit(f"v_min3_i32 v{u + 4}, v{t + 12}, v{t + 13}, v{t + 14}")
    emit(f"v_min3_i32 v{u}, v{u}, v{u + 1}, v{u + 2}")
    emit(f"v_min3_i32 v{u + 3}, v{u + 3}, v{u + 4}, v{t + 15}")
    emit(f"v_min_i32 v{ACC}, v{u}, v{u + 3}")
    red_perm()
    emit("s_waitcnt lgkmcnt(0)")
    red_final(shared)


def row_tile(a, tiles, previous, a_next=None, previous_shared=False):
    """All column tiles `tiles` against the row fragment in v[a:a+3].  Pairs of tiles go through D0/D1 and D2/D3 in
    turn: the two MFMAs of a pair are issued, then the 32 minima of the pair before; an odd last tile is folded alone.
    previous: None, 'staged' (the row tile before is reduced along the way, needs 8 pairs) or 'blocking'.
    a_next: (register, LDS operand, offset) of the next row tile's A fragment, loaded along the way."""
    n = len(tiles)
    npairs, odd = n // 2, n % 2
    assert npairs >= 1 and (previous != "staged" or npairs >= 7)

    def bufs(p):
        return (D[2], D[3]) if p & 1 else (D[0], D[1])

    def prefetch(p):                                      # what pair p - 1 leaves in flight: the operands of pair p (or the tail)
        if p < npairs:
            load_b(BSET[p & 1], tiles[2 * p])
            load_b(BSET[p & 1] + 4, tiles[2 * p + 1])
        elif odd:
            load_b(BSET[p & 1], tiles[n - 1])

    load_b(BSET[0], tiles[0])
    load_b(BSET[0] + 4, tiles[1])
    emit("s_waitcnt lgkmcnt(0)")
    mfma(D[0], a, BSET[0])
    mfma(D[1], a, BSET[0] + 4)
    prefetch(1)
    if a_next is not None:
        emit(f"ds_read_b128 v[{a_next[0]}:{a_next[0] + 3}], {a_next[1]} offset:{a_next[2]}")
    if previous == "staged":
        RED_STAGES[0](previous_shared)
    elif previous == "blocking":
        reduction_blocking(previous_shared)
    for p in range(1, npairs):
        # one MFMA, half of the previous pair's minima, the other MFMA, the other half: a wave never queues a second
        # MFMA behind its own first one (the matrix pipe takes 32 cycles per MFMA, 16 minima take 64)
        x, y = bufs(p)
        px, py = bufs(p - 1)
        m = mins_pair(px, py, CM + tiles[2 * p - 2], CM + tiles[2 * p - 1], first=(p == 1))
        emit("s_waitcnt lgkmcnt(0)")
        mfma(x, a, BSET[p & 1])
        for _ in range(DBG_NOP):
            emit("s_nop 15")
        for line in m[:16]:
            emit(line)
        mfma(y, a, BSET[p & 1] + 4)
        prefetch(p + 1)
        if previous == "staged" and p in RED_STAGES:
            RED_STAGES[p](previous_shared)
        for line in m[16:]:
            emit(line)
    lx, ly = bufs(npairs - 1)
    m = mins_pair(lx, ly, CM + tiles[n - 3 if odd else n - 2], CM + tiles[n - 2 if odd else n - 1], first=(npairs == 1))
    if odd:
        tb = bufs(npairs)[0]
        emit("s_waitcnt lgkmcnt(0)")
        mfma(tb, a, BSET[npairs & 1])
    for _ in range(DBG_NOP):
        emit("s_nop 15")
    for line in m:
        emit(line)
    if odd:
        for _ in range(DBG_NOP):
            emit("s_nop 15")
        mins_single(tb, CM + tiles[n - 1])


# The wave's share of row tile 16 FIRST (its reduction then rides on row tile 0's phases), then its four full row tiles;
# only the last one's reduction has nothing to hide behind.
emit(f"ds_read_b128 v[{ASET[0]}:{ASET[0] + 3}], %3 offset:0")
for ct in range(NCT):
    emit(f"v_mov_b32 v{CM + ct}, {INF}")
emit(f"v_mov_b32 v{ROWMAX}, 0")
for var in range(4):
    if var < 3:
        emit(f"s_cmp_eq_u32 %9, {var}")
        emit(f"s_cbranch_scc0 {10 + var}f")
    row_tile(ASET[0], list(range(*PART[var])), None, (ASET[1], "%2", 0))
    emit("s_branch 20f")
    if var < 3:
        emit(f"{10 + var}:")
emit("20:")
for k in range(4):
    nxt = (ASET[k & 1], "%2", 4096 * (k + 1)) if k < 3 else None
    row_tile(ASET[(k + 1) & 1], list(range(NCT)), "blocking" if DBG_BLOCKING else "staged", nxt, previous_shared=(k == 0))
reduction_blocking(False)
for ct in range(NCT):
    emit(f"ds_min_i32 %6, v{CM + ct} offset:{ct * 128}")
emit("s_waitcnt lgkmcnt(0)")
emit(f"v_mov_b32 %0, v{ROWMAX}")

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
