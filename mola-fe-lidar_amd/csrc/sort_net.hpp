// sort_net.hpp -- the index logic of the hand-written stable sort behind the prepare chain (map_sort.hip): a bitonic network over
// runs of 8 * T items held eight per thread (three index bits per register group, one LDS round trip per group), and the rank of an
// item among the other sorted runs of its merge group (binary lifting).  Pure functions, host + device: map_sort.hip runs them on
// the GPU, tests/hosts/sort_net_test.cpp runs the SAME functions -- and the same compile-time schedule -- thread by thread on the
// CPU against std::stable_sort.
// Replaces rocPRIM's radix / merge sort (ten launches for a 120k-point scan) in what plays the kd-tree build's role in the
// reference (a new global cloud every scan: src/LidarOdometry.cpp:215-234, 279).
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define MOLA_HD __host__ __device__ __forceinline__
#define MOLA_UNROLL _Pragma("unroll")
#else
#define MOLA_HD inline
#define MOLA_UNROLL
#endif

namespace mola_icp_amd {
namespace sortnet {

constexpr int kE = 8;  // items per thread: index bits (b0 + 2, b0 + 1, b0) of a register group live in the slot number

// item index held in register slot e of thread tid while the group's lowest index bit is b0
MOLA_HD int elem_index(int tid, int e, int b0) { return ((tid >> b0) << (b0 + 3)) | (e << b0) | (tid & ((1 << b0) - 1)); }

// LDS slot of item i: one pad slot per eight keeps every group layout free of bank conflicts (b0 = 0: 9 tid + e; b0 = 1: 18 a + c;
// b0 = 2: 36 a + c; b0 = 3: 72 a + c -- all distinct modulo 32 over 32 consecutive threads)
MOLA_HD int lds_slot(int i) { return i + (i >> 3); }

// One register group of phase k = 2^m: the compare-exchange stages for the index bits b0 + top .. b0 (top <= 2).  Items are
// (key << 32 | index) -- unique, so the network's order IS the stable order and "not less" means "greater".  Direction: ascending
// where bit m of the item index is clear (above the group -- one value per thread -- except in the phases m = 1, 2, whose only
// group is b0 = 0, where it is a slot bit: a constant per pair).  Called with constants: everything but the compares folds.
MOLA_HD void group_stages(uint64_t (&v)[kE], int tid, int m, int b0, int top)
{
    const int base = elem_index(tid, 0, b0);
    MOLA_UNROLL
    for (int s = 2; s >= 0; --s) {
        if (s > top) continue;
        MOLA_UNROLL
        for (int e = 0; e < kE; ++e) {
            if (e & (1 << s)) continue;
            const int f = e | (1 << s);
            const uint64_t a = v[e], b = v[f];
            const bool desc = ((base | (e << b0)) & (1 << m)) != 0;
            const bool sw = (b < a) != desc;
            v[e] = sw ? b : a;
            v[f] = sw ? a : b;
        }
    }
}

// The schedule, at compile time: phase m (k = 2^m) does the bits m - 1 .. 0 in groups of three from the top, the last group always at
// b0 = 0 -- so every phase starts and ends in the layout b0 = 0.  Ctx::group<M, B0, TOP, CUR>() = "bring the items from layout CUR
// to layout B0 (through LDS, if they differ), then group_stages(m = M, b0 = B0, top = TOP)": the device context does it for its own
// thread between barriers, the test's context for every thread in turn.
template <class Ctx, int M, int HI, int CUR>
MOLA_HD void run_groups(Ctx& c)
{
    if constexpr (HI >= 0) {
        constexpr int B0 = HI >= 2 ? HI - 2 : 0;
        constexpr int TOP = HI - B0;
        c.template group<M, B0, TOP, CUR>();
        run_groups<Ctx, M, B0 - 1, B0>(c);
    }
}
template <class Ctx, int M, int LOG>
MOLA_HD void run_phases(Ctx& c)
{
    if constexpr (M <= LOG) {
        run_groups<Ctx, M, M - 1, 0>(c);
        run_phases<Ctx, M + 1, LOG>(c);
    }
}

// number of keys in the ascending arr[0, len) that are < k (incl = false) or <= k (incl = true).  pow2 = a power of two >= len.
MOLA_HD int count_before(const uint32_t* arr, int len, int pow2, uint32_t k, bool incl)
{
    int pos = 0;
    for (int step = pow2; step > 0; step >>= 1) {
        const int q = pos + step;
        if (q <= len) {
            const uint32_t a = arr[q - 1];
            if (a < k || (incl && a == k)) pos = q;
        }
    }
    return pos;
}

// Merge by ranking.  keys[0, n) = runs of L items (the last one shorter), each ascending in (key, original index), run r holding
// lower original indices than run r + 1; groups of F consecutive runs merge into runs of L * F.  The destination of the item at p:
// its group's base + its place in its own run + the items of the other runs of the group that come before it -- keys <= its own in
// earlier runs (they hold lower indices: ties go first), keys < its own in later runs.  Stable, no atomics, no waiting.
MOLA_HD int merge_dest(const uint32_t* keys, int n, int L, int F, int p)
{
    const int a = p / L, g0 = (a / F) * F;
    const uint32_t k = keys[p];
    int rank = p - a * L;
    for (int r = g0; r < g0 + F; ++r) {
        const long long base = (long long)r * L;
        if (base >= n) break;
        if (r == a) continue;
        const int len = (int)((long long)n - base < (long long)L ? (long long)n - base : (long long)L);
        rank += count_before(keys + base, len, L, k, r < a);
    }
    return g0 * L + rank;
}

// The device's form of the same destination, fan-in F (a power of two).  (i) The searches of an item in the F - 1 other runs are
// independent of each other: every step issues their F loads together, then does the F compares -- one chain of dependent loads
// instead of F - 1.  Straight-line: a run that does not take part (the item's own, one past the end of the data) is probed at a
// harmless place and neutralised by a threshold of 0 ("no key is below it"); a first version that skipped such runs with
// `continue` was compiled into load / wait / compare one run after the other.  (ii) The runs are PADDED: keys[0, n_pad), n_pad a
// multiple of L, the slots behind the last real item holding 0xffffffff (the block sort and every merge level write them).  A
// search then needs no length: binary lifting with the steps L / 2 .. 1 finds min(count, L - 1) without leaving the run, one more
// probe at arr[pos] settles a run that lies entirely before the item; a padding key never counts -- it is >= every threshold
// (k + 1 <= 0xffffffff for the contract's keys <= 0xfffffffe).  (iii) `a`, the item's own run, is passed in: on the device it comes
// from the workgroup index (a workgroup never straddles runs), so run bases are scalar and the thresholds are set once.
// Per search step: add, load, compare, select.  L = 1 << logL.
template <int F>
MOLA_HD int merge_dest_padded(const uint32_t* keys, long long n_pad, int logL, int p, int a)
{
    const int L = 1 << logL;
    const int g0 = a & ~(F - 1);
    const uint32_t k = keys[p];
    const uint32_t* arr[F];
    uint32_t kk[F];
    int pos[F];
    MOLA_UNROLL
    for (int g = 0; g < F; ++g) {
        const int r = g0 + g;
        const long long base = (long long)r << logL;
        const bool live = r != a && base < n_pad;
        arr[g] = keys + (live ? base : ((long long)a << logL));
        kk[g] = live ? k + (r < a ? 1u : 0u) : 0u;   // earlier runs hold lower indices: their equal keys go first
        pos[g] = 0;
    }
    for (int step = L >> 1; step >= 0; step = step > 1 ? step >> 1 : step - 1) {
        const int st = step ? step : 1;
        uint32_t av[F];
        MOLA_UNROLL
        for (int g = 0; g < F; ++g) av[g] = arr[g][pos[g] + st - 1];
        MOLA_UNROLL
        for (int g = 0; g < F; ++g) pos[g] = av[g] < kk[g] ? pos[g] + st : pos[g];
    }
    int rank = p - (a << logL);
    MOLA_UNROLL
    for (int g = 0; g < F; ++g) rank += pos[g];
    return (g0 << logL) + rank;
}

}  // namespace sortnet
}  // namespace mola_icp_amd
