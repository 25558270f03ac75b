// sort_net_test.cpp -- the prepare chain's sort (mola-fe-lidar_amd/csrc/sort_net.hpp) run thread by thread on the CPU.
// The device kernels (map_sort.hip: k_sort_runs, k_rank_merge) call the SAME functions with the same schedule; this host
// walks that schedule with an array of per-thread registers and one LDS image, and compares with std::stable_sort.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

#include "sort_net.hpp"

using namespace mola_icp_amd::sortnet;

static int g_fail = 0;
#define CHECK(c, ...)                         \
    do {                                      \
        if (!(c)) {                           \
            if (g_fail < 20) {                \
                std::printf("FAIL " __VA_ARGS__); \
                std::printf("\n");            \
            }                                 \
            ++g_fail;                         \
        }                                     \
    } while (0)

// one run of RUN = 8 T items: what k_sort_runs does between its barriers -- the compile-time schedule (run_phases) drives this
// context exactly as it drives the device's, one group at a time; here every thread takes its turn inside a group
struct HostCtx {
    int T;
    std::vector<uint64_t> lds, regs;
    int exchanges = 0;
    template <int M, int B0, int TOP, int CUR>
    void group()
    {
        if (B0 != CUR) {  // layout change: write (old layout), barrier, read (new layout), barrier
            ++exchanges;
            for (int tid = 0; tid < T; ++tid)
                for (int e = 0; e < kE; ++e) lds[(size_t)lds_slot(elem_index(tid, e, CUR))] = regs[(size_t)tid * kE + e];
            for (int tid = 0; tid < T; ++tid)
                for (int e = 0; e < kE; ++e) regs[(size_t)tid * kE + e] = lds[(size_t)lds_slot(elem_index(tid, e, B0))];
        }
        for (int tid = 0; tid < T; ++tid) {
            uint64_t v[kE];
            for (int e = 0; e < kE; ++e) v[e] = regs[(size_t)tid * kE + e];
            group_stages(v, tid, M, B0, TOP);
            for (int e = 0; e < kE; ++e) regs[(size_t)tid * kE + e] = v[e];
        }
    }
};

template <int LOG>
static void block_sort_log(std::vector<uint64_t>& items)
{
    const int RUN = 1 << LOG, T = RUN / kE;
    HostCtx c;
    c.T = T;
    c.lds.assign((size_t)RUN + RUN / 8, ~0ull);
    c.regs.resize((size_t)T * kE);
    for (int tid = 0; tid < T; ++tid)
        for (int e = 0; e < kE; ++e) c.regs[(size_t)tid * kE + e] = items[(size_t)elem_index(tid, e, 0)];
    run_phases<HostCtx, 1, LOG>(c);
    // (every phase ends in the layout b0 = 0: slot e of thread tid = sorted position 8 tid + e)
    for (int tid = 0; tid < T; ++tid)
        for (int e = 0; e < kE; ++e) items[(size_t)elem_index(tid, e, 0)] = c.regs[(size_t)tid * kE + e];
    if (LOG == 12) CHECK(c.exchanges == 27, "runs of 4096: 27 LDS round trips expected, %d made", c.exchanges);
    if (LOG == 11) CHECK(c.exchanges == 23, "runs of 2048: 23 LDS round trips expected, %d made", c.exchanges);
}

static void block_sort(std::vector<uint64_t>& items /*RUN*/, int T)
{
    switch (T * kE) {
        case 64: block_sort_log<6>(items); break;
        case 128: block_sort_log<7>(items); break;
        case 2048: block_sort_log<11>(items); break;
        case 4096: block_sort_log<12>(items); break;
        default: CHECK(false, "no instantiation for runs of %d", T * kE);
    }
}

// the whole sort: runs of RUN, then merge levels of fan-in F, on PADDED arrays as the device keeps them; returns order[r] =
// original index of the r-th item
static std::vector<uint32_t> full_sort(const std::vector<uint32_t>& keys, int T, int F)
{
    const int n = (int)keys.size(), RUN = 8 * T;
    int logF = 0;
    while ((1 << logF) < F) ++logF;
    CHECK((1 << logF) == F, "fan-in must be a power of two");
    // every level's input is padded to whole runs of ITS run length: the longest one decides the allocation
    long long L_last = RUN;
    while (L_last * F < n) L_last *= F;
    const long long n_alloc = (n + L_last - 1) / L_last * L_last;
    const int n_runs = (int)((n + RUN - 1) / RUN);
    std::vector<uint32_t> ka((size_t)n_alloc, 0xdeadbeefu), ia((size_t)n_alloc, 0u), kb((size_t)n_alloc, 0xdeadbeefu), ib((size_t)n_alloc, 0u);
    for (int b = 0; b < n_runs; ++b) {
        std::vector<uint64_t> items((size_t)RUN);
        for (int j = 0; j < RUN; ++j) {
            const int i = b * RUN + j;
            items[(size_t)j] = ((uint64_t)(i < n ? keys[(size_t)i] : 0xffffffffu) << 32) | (uint32_t)i;
        }
        block_sort(items, T);
        for (int j = 1; j < RUN; ++j) CHECK(items[(size_t)j - 1] < items[(size_t)j], "run %d is not ascending at %d", b, j);
        for (int j = 0; j < RUN; ++j) {   // (the device writes the whole run: the padding keys are what later searches read)
            if (b * RUN + j < n) CHECK((uint32_t)items[(size_t)j] < (uint32_t)n, "a padding item in front of a real one");
            ka[(size_t)(b * RUN + j)] = (uint32_t)(items[(size_t)j] >> 32);
            ia[(size_t)(b * RUN + j)] = (uint32_t)items[(size_t)j];
        }
    }
    long long L = RUN;
    int logL = 0;
    while ((1ll << logL) < L) ++logL;
    for (;;) {   // (at least one level: the device's last level is also the gather)
        const bool last = L * F >= n;
        const long long n_pad_in = (n + L - 1) / L * L;
        const long long n_pad_out = last ? n : (n + L * F - 1) / (L * F) * (L * F);
        std::vector<int> hit((size_t)n, 0);
        for (long long p = 0; p < n_pad_out; ++p) {
            if (p >= n) { kb[(size_t)p] = 0xffffffffu; continue; }   // the padding of the NEXT level's last run
            const int d = merge_dest(ka.data(), n, (int)L, F, (int)p);
            int dp = -1;
            switch (F) {
                case 2: dp = merge_dest_padded<2>(ka.data(), n_pad_in, logL, (int)p, (int)(p >> logL)); break;
                case 4: dp = merge_dest_padded<4>(ka.data(), n_pad_in, logL, (int)p, (int)(p >> logL)); break;
                case 8: dp = merge_dest_padded<8>(ka.data(), n_pad_in, logL, (int)p, (int)(p >> logL)); break;
                case 32: dp = merge_dest_padded<32>(ka.data(), n_pad_in, logL, (int)p, (int)(p >> logL)); break;
                case 64: dp = merge_dest_padded<64>(ka.data(), n_pad_in, logL, (int)p, (int)(p >> logL)); break;
                default: break;
            }
            CHECK(d == dp, "the padded search (the device's form) disagrees at %lld: %d vs %d", p, dp, d);
            CHECK(d >= 0 && d < n, "destination out of range");
            if (d < 0 || d >= n) continue;
            ++hit[(size_t)d];
            kb[(size_t)d] = ka[(size_t)p];
            ib[(size_t)d] = ia[(size_t)p];
        }
        for (int p = 0; p < n; ++p) CHECK(hit[(size_t)p] == 1, "destination %d written %d times", p, hit[(size_t)p]);
        ka.swap(kb);
        ia.swap(ib);
        if (last) break;
        L *= F;
        logL += logF;
    }
    ia.resize((size_t)n);
    return ia;
}

static void one_case(int n, int T, int F, int key_bits, unsigned seed)
{
    std::mt19937 rng(seed);
    std::vector<uint32_t> keys((size_t)n);
    const uint32_t mask = key_bits >= 32 ? 0xffffffffu : ((1u << key_bits) - 1u);
    for (auto& k : keys) k = (uint32_t)rng() & mask;
    if (key_bits == 30 && n > 8) { keys[0] = keys[(size_t)n - 1] = 0x3fffffffu; keys[(size_t)n / 2] = 0u; }
    std::vector<uint32_t> want((size_t)n);
    std::iota(want.begin(), want.end(), 0u);
    std::stable_sort(want.begin(), want.end(), [&](uint32_t a, uint32_t b) { return keys[a] < keys[b]; });
    const std::vector<uint32_t> got = full_sort(keys, T, F);
    const bool same = got == want;
    CHECK(same, "n=%d T=%d F=%d bits=%d: order differs from std::stable_sort", n, T, F, key_bits);
    if (same) std::printf("ok   n=%d runs of %d, fan-in %d, %d-bit keys\n", n, 8 * T, F, key_bits);
}

int main()
{
    // bank-conflict claim of lds_slot: 32 consecutive threads, fixed register slot, layouts b0 = 0..3 -> 32 distinct slots mod 32
    for (int b0 = 0; b0 <= 3; ++b0)
        for (int e = 0; e < kE; ++e) {
            unsigned seen = 0;
            for (int tid = 0; tid < 32; ++tid) seen |= 1u << (lds_slot(elem_index(tid, e, b0)) & 31);
            CHECK(seen == 0xffffffffu, "b0=%d e=%d: LDS slots collide modulo 32", b0, e);
        }
    // every (tid, e) names a distinct item in every layout
    for (int b0 = 0; b0 <= 9; ++b0) {
        std::vector<int> cnt(4096, 0);
        for (int tid = 0; tid < 512; ++tid)
            for (int e = 0; e < kE; ++e) ++cnt[(size_t)elem_index(tid, e, b0)];
        for (int i = 0; i < 4096; ++i) CHECK(cnt[(size_t)i] == 1, "layout b0=%d does not cover item %d once", b0, i);
    }
    // the shipped shape (runs of 2048, fan-in 8) at odometry size, with many equal keys and with 30-bit keys
    one_case(120000, 256, 8, 30, 1);       // two levels
    one_case(120000, 256, 8, 6, 2);
    one_case(131072, 256, 8, 30, 3);
    one_case(2048, 256, 8, 30, 4);
    one_case(2049, 256, 8, 12, 5);
    one_case(16384, 256, 8, 30, 15);       // exactly one full group
    one_case(16385, 256, 8, 9, 16);
    one_case(1, 256, 8, 30, 6);
    one_case(63, 256, 8, 3, 7);
    one_case(300000, 256, 8, 30, 8);       // three levels
    one_case(140000, 256, 8, 4, 9);        // three levels, heavy ties
    one_case(131073, 256, 8, 21, 13);      // three levels, one item in the second run of the top level
    one_case(100000, 512, 32, 30, 14);     // the first shapes this sort had
    one_case(100000, 256, 64, 30, 17);
    // small shapes walk deeper merge trees and odd run counts
    one_case(5000, 8, 4, 8, 10);
    one_case(70000, 16, 8, 10, 11);
    one_case(1023, 8, 2, 5, 12);
    std::printf(g_fail ? "FAILED (%d)\n" : "PASSED\n", g_fail);
    return g_fail ? 1 : 0;
}
