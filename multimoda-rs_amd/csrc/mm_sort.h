// mm_sort.h -- the stable argsort behind Contour::sort_contour_points (contour.rs:385-390: slice::sort_by on the
// atan2 keys, stable).  Contours arrive in angular order far more often than not -- keys ascending, or ascending
// with a single wrap (a rotation of a sorted sequence: what a sorted contour looks like after Frame::rotate) -- and
// for those the stable order is known without sorting: one pass instead of n log n indirect compares.
#pragma once

#include <algorithm>
#include <cstdint>
#include <numeric>

namespace mm {

// perm[i] = index of the i-th smallest key, equal keys in index order.
inline void stable_argsort(const double* key, int64_t n, int32_t* perm)
{
    int descents = 0;
    int64_t wrap = 0;
    bool ordered = n > 0 && key[0] == key[0];                  // a NaN key: no total order, take the general path
    for (int64_t i = 1; i < n && ordered; ++i) {
        if (key[i] < key[i - 1]) { wrap = i; if (++descents > 1) ordered = false; }
        else if (!(key[i] >= key[i - 1])) ordered = false;     // NaN
    }
    if (ordered && descents == 0) { std::iota(perm, perm + n, 0); return; }
    // one descent at `wrap`: [0, wrap) and [wrap, n) are ascending runs.  If the last key is STRICTLY below the first,
    // every key of the second run precedes every key of the first and no tie straddles the two, so the stable order
    // is the second run followed by the first.  (With key[n-1] == key[0] stability wants the earlier index first.)
    if (ordered && descents == 1 && key[n - 1] < key[0]) {
        std::iota(perm, perm + (n - wrap), (int32_t)wrap);
        std::iota(perm + (n - wrap), perm + n, 0);
        return;
    }
    std::iota(perm, perm + n, 0);
    std::stable_sort(perm, perm + n, [key](int32_t a, int32_t b) { return key[a] < key[b]; });
}

}  // namespace mm
