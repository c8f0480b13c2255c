#!/usr/bin/env python3
"""Generates multimoda-rs_amd/csrc/mm_screen_mx_asm.inc: the main phase of k_screen_mx (mm_kernels.hip) as ONE asm block on
fixed registers, so that the order is exactly the software pipeline we want -- two MFMAs (1024 squared distances each, on
the matrix pipe) issued ahead of the 32 v_min3_i32 that fold the PREVIOUS two tiles on the vector pipe.  The compiler's
scheduler does not produce this order (tools/ubench_mfma16*.hip: 46 ns per tile compiler-scheduled, 34 ns hand-ordered).

One wave, one candidate: its four full row tiles (17 column tiles each) and its share of the 17th row tile (4 or 5 column
tiles, chosen by the scalar operand `variant`), so that the four waves of a workgroup carry 72 or 73 tiles each.
Register map (VGPR):
  v[20:35]   rmin[16]   running row minima of the current row tile (element v of the 32x32 result layout)
  v36        rowmax     max over rows of the row minima (signed-int order on f32 bits, floored at 0)
  v[40:56]   cm[17]     running column minima per column tile (this lane's column, this wave's rows)
  v[60:79]   A operand fragments of the wave's 5 row tiles
  v[80:87], v[88:95]    B operand fragments, two pairs (double buffer)
  v[100:163] D0..D3     four 32x32 result tiles
Operands: %0 out rowmax; %1 vB (LDS byte address of this lane's B fragment in column tile 0); %2 vA (A fragment of the
wave's first row tile); %3 vA16 (A fragment of row tile 16); %4 vRW / %5 vRR (row-reduction scratch: write / read
address); %6 vCM (column-minimum array, this lane's column of tile 0); %7 vPERM (4 * (lane ^ 32)); %8 vR16 (row minima of
row tile 16, shared by the four waves: this lane's row); %9 s variant (0..3)."""
import os

INF = "0x7f800000"
RMIN, ROWMAX, CM, A0, BSET, D = 20, 36, 40, 60, [80, 88], [100, 116, 132, 148]
NCT = 17
RED_STRIDE = 136          # bytes between rows of the reduction scratch (34 dwords: 8-byte aligned reads, 2-way conflicts)
PART = [(0, 5), (5, 9), (9, 13), (13, 17)]          # column tiles of row tile 16 per variant

out = []
emit = out.append


def mfma(d, a, b):
    emit(f"v_mfma_f32_32x32x16_f16 v[{d}:{d + 15}], v[{a}:{a + 3}], v[{b}:{b + 3}], 0")


def load_b(dst, tile):
    # a column fragment is stored as 8 bytes per lane -- (x1, x2, y1, y2) or (256, 1, n2h, n2l) -- and read twice: the K
    # slots 4..7 repeat slots 0..3 (the row fragments are laid out to match; for the norm half their slots 4..7 are 0)
    emit(f"ds_read_b64 v[{dst}:{dst + 1}], %1 offset:{tile * 512}")
    emit(f"ds_read_b64 v[{dst + 2}:{dst + 3}], %1 offset:{tile * 512}")


def mins_pair(dA, dB, cmA, cmB):
    """32 v_min3_i32: column folds of two tiles (8 each) and the elementwise row minima over both (16), interleaved so
    that no instruction depends on the one before it."""
    col = []
    for q in range(8):
        col.append(f"v_min3_i32 v{cmA}, v{cmA}, v{dA + 2 * q}, v{dA + 2 * q + 1}")
        col.append(f"v_min3_i32 v{cmB}, v{cmB}, v{dB + 2 * q}, v{dB + 2 * q + 1}")
    row = [f"v_min3_i32 v{RMIN + v}, v{RMIN + v}, v{dA + v}, v{dB + v}" for v in range(16)]
    for i in range(16):
        emit(col[i])
        emit(row[i])


def mins_single(d, cm):
    col = [f"v_min3_i32 v{cm}, v{cm}, v{d + 2 * q}, v{d + 2 * q + 1}" for q in range(8)]
    row = [f"v_min_i32 v{RMIN + v}, v{RMIN + v}, v{d + v}" for v in range(16)]
    for i in range(8):
        emit(col[i])
        emit(row[2 * i])
        emit(row[2 * i + 1])


def reduction(shared):
    """rmin[16] of a finished row tile: through the wave's LDS scratch (row-major, one row = the 32 columns of a half
    wave), each lane folds 16 values of one row, the two halves of a row meet through ds_bpermute.  shared = False:
    rowmax takes the max; True (row tile 16, split over the waves): ds_min_i32 into the workgroup's row array."""
    t, u = D[2], D[3]                                     # free at every place this is emitted
    for v in range(16):
        emit(f"ds_write_b32 %4, v{RMIN + v} offset:{v * RED_STRIDE}")
    emit("s_waitcnt lgkmcnt(0)")
    for q in range(8):
        emit(f"ds_read_b64 v[{t + 2 * q}:{t + 2 * q + 1}], %5 offset:{8 * q}")
    emit("s_waitcnt lgkmcnt(0)")
    emit(f"v_min3_i32 v{u}, v{t}, v{t + 1}, v{t + 2}")
    emit(f"v_min3_i32 v{u + 1}, v{t + 3}, v{t + 4}, v{t + 5}")
    emit(f"v_min3_i32 v{u + 2}, v{t + 6}, v{t + 7}, v{t + 8}")
    emit(f"v_min3_i32 v{u + 3}, v{t + 9}, v{t + 10}, v{t + 11}")
    emit(f"v_min3_i32 v{u + 4}, v{t + 12}, v{t + 13}, v{t + 14}")
    emit(f"v_min3_i32 v{u}, v{u}, v{u + 1}, v{u + 2}")
    emit(f"v_min3_i32 v{u + 3}, v{u + 3}, v{u + 4}, v{t + 15}")
    emit(f"v_min_i32 v{u}, v{u}, v{u + 3}")
    emit(f"ds_bpermute_b32 v{u + 1}, %7, v{u}")
    emit("s_waitcnt lgkmcnt(0)")
    emit(f"v_min_i32 v{u}, v{u}, v{u + 1}")
    if shared:
        emit(f"ds_min_i32 %8, v{u}")
    else:
        emit(f"v_max_i32 v{ROWMAX}, v{ROWMAX}, v{u}")


def row_tile(a, tiles, reduce_previous):
    """All column tiles `tiles` against the row fragment in v[a:a+3].  Pairs of tiles go through D0/D1 and D2/D3 in
    turn: the two MFMAs of a pair are issued, then the 32 minima of the pair before; an odd last tile is folded alone."""
    n = len(tiles)
    npairs, odd = n // 2, n % 2
    assert npairs >= 1

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
    if reduce_previous:
        reduction(False)                                  # of the row tile before, beside this tile's first two MFMAs
    for v in range(16):
        emit(f"v_mov_b32 v{RMIN + v}, {INF}")
    for p in range(1, npairs):
        x, y = bufs(p)
        emit("s_waitcnt lgkmcnt(0)")
        mfma(x, a, BSET[p & 1])
        mfma(y, a, BSET[p & 1] + 4)
        prefetch(p + 1)
        px, py = bufs(p - 1)
        mins_pair(px, py, CM + tiles[2 * p - 2], CM + tiles[2 * p - 1])
    lx, ly = bufs(npairs - 1)
    if odd:
        tb = bufs(npairs)[0]
        emit("s_waitcnt lgkmcnt(0)")
        mfma(tb, a, BSET[npairs & 1])
        mins_pair(lx, ly, CM + tiles[n - 3], CM + tiles[n - 2])
        mins_single(tb, CM + tiles[n - 1])
    else:
        mins_pair(lx, ly, CM + tiles[n - 2], CM + tiles[n - 1])


emit(f"ds_read_b128 v[{A0}:{A0 + 3}], %2 offset:0")
emit(f"ds_read_b128 v[{A0 + 4}:{A0 + 7}], %2 offset:4096")
emit(f"ds_read_b128 v[{A0 + 8}:{A0 + 11}], %2 offset:8192")
emit(f"ds_read_b128 v[{A0 + 12}:{A0 + 15}], %2 offset:12288")
emit(f"ds_read_b128 v[{A0 + 16}:{A0 + 19}], %3 offset:0")
for ct in range(NCT):
    emit(f"v_mov_b32 v{CM + ct}, {INF}")
emit(f"v_mov_b32 v{ROWMAX}, 0")
for k in range(4):
    row_tile(A0 + 4 * k, list(range(NCT)), k > 0)
# the wave's share of row tile 16
for var in range(4):
    if var < 3:
        emit(f"s_cmp_eq_u32 %9, {var}")
        emit(f"s_cbranch_scc0 {10 + var}f")
    row_tile(A0 + 16, list(range(*PART[var])), True)
    emit("s_branch 20f")
    if var < 3:
        emit(f"{10 + var}:")
emit("20:")
reduction(True)
for ct in range(NCT):
    emit(f"ds_min_i32 %6, v{CM + ct} offset:{ct * 128}")
emit("s_waitcnt lgkmcnt(0)")
emit(f"v_mov_b32 %0, v{ROWMAX}")

regs = sorted(set(range(20, 37)) | set(range(40, 57)) | set(range(60, 96)) | set(range(100, 164)))
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
