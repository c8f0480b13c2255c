#!/bin/bash
# AddressSanitizer + UBSan over the HOST code of the library (csrc/*.cpp: builder, frame bookkeeping, centerline
# placement, CCTA host side, engine / plan staging) on the CPU: the device code is compiled as usual (GPU ASan is
# not available on this pool), the .so goes to a scratch directory, and the CPU test suite runs against it through
# MM_LIB_PATH with the sanitizer runtime preloaded.  Usage: tools/asan_host.sh [pytest args]
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$(mktemp -d /tmp/mm_asan.XXXXXX)"
trap 'rm -rf "$OUT"' EXIT
HIPCC=/opt/rocm/bin/hipcc
CSRC="$ROOT/multimoda-rs_amd/csrc"
COMMON="--offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function"
SAN="-fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -g -O1"
objs=()
for s in mm_kernels.hip mm_nn_kernels.hip; do
    $HIPCC -x hip $COMMON -O3 -c "$CSRC/$s" -o "$OUT/${s%.*}.o" & objs+=("$OUT/${s%.*}.o")
done
for s in mm_engine.cpp mm_host.cpp mm_centerline.cpp mm_ccta.cpp mm_build.cpp mm_frames.cpp mm_comm.cpp; do
    $HIPCC -x hip $COMMON $SAN -c "$CSRC/$s" -o "$OUT/${s%.*}.o" & objs+=("$OUT/${s%.*}.o")
done
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC -pthread -fsanitize=address,undefined -shared-libsan -o "$OUT/libmm_hausdorff.so" "${objs[@]}" -ldl
RT="$(dirname "$($HIPCC -print-file-name=libclang_rt.asan-x86_64.so)")"
cd "$ROOT"
# detect_leaks=0: the interpreter itself never frees everything; halt on the first real error
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
LD_PRELOAD="$RT/libclang_rt.asan-x86_64.so" LD_LIBRARY_PATH="$RT:${LD_LIBRARY_PATH:-}" MM_LIB_PATH="$OUT/libmm_hausdorff.so" \
    python -m pytest tests -q -m "not gpu" -x -p no:cacheprovider "$@"
