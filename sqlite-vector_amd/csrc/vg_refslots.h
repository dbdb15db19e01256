// vg_refslots.h - host-side replay of the reference's top-k SLOT algorithm (tie_order = reference).
//
// The reference keeps k unsorted slots: a row enters iff its distance is strictly below the current maximum and takes
// the place of the FIRST slot holding that maximum (sqlite-vector.c:2102-2106 / :2138-2146 / :2218-2223 with
// vFullScanFindMaxIndex :2022-2049); the slots are exchange-sorted at the end (vFullScanSortSlots :2051-2069).  Among
// EQUAL distances the surviving rowids and their order therefore depend on the slot history of the whole stream - no
// order on (distance, position) reproduces it.  The multiset of distances is the k smallest either way.
//
// What makes an exact replay cheap: a row can only enter if its distance is below the k-th smallest distance of the rows
// BEFORE it, and that bound only falls.  The device hands over the first P rows' distances, the host replays them, and
// from then on only rows below the bound reached so far can matter - the device compacts those (vg_reforder.hip), a few
// thousand out of millions, and the host replays them in scan order with the exact rule again.
#pragma once

#include <cmath>
#include <algorithm>
#include <cstdint>
#include <utility>
#include <vector>

struct VgRefSlots {
    int k = 0;
    std::vector<double> dist;          // widened floats, like vFullScanCursor.distance (sqlite-vector.c:1809)
    std::vector<int64_t> pos;          // global scan position of the row in the slot (the caller maps it to a rowid)
    int max_index = 0;                 // a fresh cursor starts at slot 0
    double cur_max = INFINITY;

    void init(int k_) {
        k = k_;
        dist.assign((size_t)k, (double)INFINITY);
        pos.assign((size_t)k, -1);
        max_index = 0;
        cur_max = INFINITY;
    }
    // first slot holding the maximum: both branches of vFullScanFindMaxIndex (k <= 32 and the 4-way unrolled one) return it
    int find_max() const {
        int m = 0;
        for (int i = 1; i < k; ++i) if (dist[(size_t)i] > dist[(size_t)m]) m = i;
        return m;
    }
    inline void offer(float d, int64_t p) {
        if ((double)d < cur_max) {                       // strict '<'; NaN and +Inf never enter
            dist[(size_t)max_index] = (double)d;
            pos[(size_t)max_index] = p;
            max_index = find_max();
            cur_max = dist[(size_t)max_index];
        }
    }
    // the bound a later row has to beat, as a float (the slots hold widened floats, so the narrowing is exact)
    float bound() const { return (float)cur_max; }
    // vFullScanSortSlots: exchange sort, ascending; returns the number of rows (slots that are not +Inf)
    int finish() {
        int inf = 0;
        for (int i = 0; i < k - 1; ++i) {
            if (dist[(size_t)i] == (double)INFINITY) ++inf;
            for (int j = i + 1; j < k; ++j)
                if (dist[(size_t)j] < dist[(size_t)i]) { std::swap(dist[(size_t)i], dist[(size_t)j]); std::swap(pos[(size_t)i], pos[(size_t)j]); }
        }
        if (k > 0 && dist[(size_t)k - 1] == (double)INFINITY) ++inf;
        return k - inf;
    }
};

// rows of the stream the host replays before it asks the device for candidates: balances the host's work on the prefix
// (P rows at ~2 ns each) against the expected number of candidates behind it (~ k N / P, each copied, sorted and replayed:
// ~30 ns) - the optimum is ~4 sqrt(k N); at 10M rows, k = 20: 57k prefix rows, ~3500 candidates (one 64 KiB copy)
static inline int64_t vg_ref_prefix_rows(int64_t n, int k) {
    int64_t p = 4 * (int64_t)std::sqrt((double)k * (double)n);
    if (p < 4096) p = 4096;
    if (p < 4 * (int64_t)k) p = 4 * (int64_t)k;
    return p < n ? p : n;
}

struct VgRefCand { int64_t gpos; float d; };

// The replay driver.  Src provides the stream of one scan's distances in GLOBAL scan order:
//   int fetch(int64_t g0, int64_t cnt, float *out)                       distances of positions [g0, g0 + cnt)
//   int below(int64_t g0, float bound, std::vector<VgRefCand> &out, bool *overflow)
//                                                                        every position >= g0 with distance < bound (any
//                                                                        order), or *overflow = true when there are too many
// Returns 0 or the source's error code; `slots` then holds the reference's slot state before its final sort.
template <class Src>
static int vg_ref_replay(Src &src, int64_t n, int k, VgRefSlots &slots) {
    slots.init(k);
    if (n <= 0 || k <= 0) return 0;
    const int64_t P = vg_ref_prefix_rows(n, k);
    std::vector<float> buf((size_t)P);
    int rc = src.fetch(0, P, buf.data());
    if (rc != 0) return rc;
    for (int64_t i = 0; i < P; ++i) slots.offer(buf[(size_t)i], i);
    int64_t g = P;
    std::vector<VgRefCand> cand;
    while (g < n) {
        bool overflow = false;
        cand.clear();
        const float bound = slots.bound();
        if (bound == INFINITY) overflow = true;              // fewer than k finite rows so far: every finite row still enters
        else if ((rc = src.below(g, bound, cand, &overflow)) != 0) return rc;
        if (!overflow) {
            std::sort(cand.begin(), cand.end(), [](const VgRefCand &a, const VgRefCand &b) { return a.gpos < b.gpos; });
            for (const VgRefCand &c : cand) slots.offer(c.d, c.gpos);
            break;
        }
        // too many rows below the bound (heavy ties / adversarial order): replay the next stretch on the host, ask again
        const int64_t cnt = std::min<int64_t>(n - g, std::max<int64_t>(P, 1 << 20));
        buf.resize((size_t)cnt);
        if ((rc = src.fetch(g, cnt, buf.data())) != 0) return rc;
        for (int64_t i = 0; i < cnt; ++i) slots.offer(buf[(size_t)i], g + i);
        g += cnt;
    }
    return 0;
}

// The same stream CONTINUED: `slots` holds the state the rows before this stretch left (vg_slabscan.hip: a table scanned slab by slab);
// the stretch's rows are positions [0, n) of src and global positions gbase + [0, n).  No prefix: the bound the earlier rows reached is
// already low, so the rows below it are few.  (A first stretch - slots fresh - is replayed by vg_ref_replay itself.)
template <class Src>
static int vg_ref_replay_more(Src &src, int64_t n, int k, VgRefSlots &slots, int64_t gbase) {
    if (n <= 0 || k <= 0) return 0;
    const int64_t P = vg_ref_prefix_rows(n, k);
    std::vector<float> buf;
    std::vector<VgRefCand> cand;
    int64_t g = 0;
    int rc;
    while (g < n) {
        bool overflow = false;
        cand.clear();
        const float bound = slots.bound();
        if (bound == INFINITY) overflow = true;
        else if ((rc = src.below(g, bound, cand, &overflow)) != 0) return rc;
        if (!overflow) {
            std::sort(cand.begin(), cand.end(), [](const VgRefCand &a, const VgRefCand &b) { return a.gpos < b.gpos; });
            for (const VgRefCand &c : cand) slots.offer(c.d, gbase + c.gpos);
            break;
        }
        const int64_t cnt = std::min<int64_t>(n - g, std::max<int64_t>(P, 1 << 20));
        buf.resize((size_t)cnt);
        if ((rc = src.fetch(g, cnt, buf.data())) != 0) return rc;
        for (int64_t i = 0; i < cnt; ++i) slots.offer(buf[(size_t)i], gbase + g + i);
        g += cnt;
    }
    return 0;
}
