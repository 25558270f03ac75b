// kernels_q4.hpp -- k_nn_q4: the point-to-point matcher for launches with FEWER ITEMS THAN WAVE SLOTS (odometry-size scans, the
// shards of a query-sharded align, the loop-closure Monte-Carlo) -- four lanes per query.
// Device code of the ICP core for gfx950; included by hip_backend.hip only.  Numeric contract: hip_backend.hip / DESIGN.md.
//
// Below ~0.3M queries a launch is as long as ONE item's chain of dependent steps (kernels_coop.hpp), and ten rewrites that took
// WORK out of k_nn_coop's item did not shorten it (LAB_NOTEBOOK.md, round 5).  This kernel changes the item's SHAPE instead:
//   * an item = 16 consecutive sorted queries on ONE wave, lane 4 q + s = query q, sub-lane s.  A sixteenth of k_nn_coop's item
//     in queries and (Hilbert order) roughly an eighth in volume: its tile list is what round 5's quads listed -- 5.3 tiles
//     where the 64-query item met 9.6 -- and there are eight times as many, shorter items for the same wave slots;
//   * the box tests run FOUR boxes per instruction group: sub-lane s of every query tests candidate s of the next four (the boxes
//     reach the lanes by one ds_bpermute per coordinate instead of six v_readlane per box), so the block that round 5 found to be
//     the largest of an item -- ~25 tile tests of 22 instructions -- shrinks fourfold per candidate;
//   * a listed tile's 32 points go from global memory STRAIGHT INTO LDS (global_load_lds_dwordx4: ONE instruction carries two
//     tiles -- 24 lanes x 16 bytes each -- into the wave's ring, nothing passes through registers), all listed tiles of a typical
//     item in flight at once; sub-lane s then evaluates points 8 s .. 8 s + 7 -- exactly one bookkeeping group (kGroup) -- of every
//     listed tile for its query;
//   * the four partial results of a query close with two DPP steps (quad_perm), the winning group is re-read two points per sub-lane.
// Exact for the reason every tiled kernel is: a tile no lane lists is one none of the wave's queries reaches under its LIVE bound
// (the minimum over the query's four sub-lanes).  Bookkeeping per 8-point group as nn_visit_fast / k_nn_coop's merge: the results --
// pairing, kept d2, and the rows of unit-weight sums (ONE workgroup = four waves = the 64 queries of a row; the row is formed
// by item_row_from_records on the fp64 matrix cores in item_row_mfma's order) -- are bit-identical to the other matchers'.
// An exact distance tie between two groups (duplicate points, lattices) sends the WAVE through its 16 queries again with the
// packed-key visitor.
#pragma once
#include "kernels_coop.hpp"

namespace mola_icp_amd {

constexpr int kQ4Bank = 4;                                  // tiles per bank of a wave's ring (two global_load_lds_dwordx4)
constexpr int kQ4TileFloats = 3 * kTileG;                   // x[32] y[32] z[32]
constexpr int kQ4BankFloats = kQ4Bank * kQ4TileFloats;      // 384
constexpr int kQ4RingFloats = 2 * kQ4BankFloats;            // two banks per wave: 3 KB
#ifndef MOLA_Q4_WG_PER_CU
#define MOLA_Q4_WG_PER_CU 6
#endif
constexpr int kQ4WorkgroupsPerCu = MOLA_Q4_WG_PER_CU;     // launch bounds: workgroups of four waves per CU (tuning: -DMOLA_Q4_WG_PER_CU=n)
constexpr int kQ4ListCap = 64;
// -DMOLA_Q4_DIAG: per-wave clock stamps (q4_launch.hip prints their medians after every launch; never in the product build)
#ifdef MOLA_Q4_DIAG
#define Q4_STAMP(k) do { if (dbg_w) dbg_w[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define Q4_STAMP(k) do { } while (0)
#endif                              // list entries live in one vector register, entry n in lane n

__device__ __forceinline__ float quad_min(float v)
{
    v = fminf(v, dpp_f<kDppXor1>(v));
    return fminf(v, dpp_f<kDppXor2>(v));
}
// min / max over the wave's 16 queries (every sub-lane of a query holds the same value), as a scalar: kernels_tiled.hpp's wave
// reduction without its first two rotations (the four sub-lanes of a query already agree)
__device__ __forceinline__ float wave_min_q(float v)
{
    v = fminf(v, dpp_f<kDppRor4>(v)); v = fminf(v, dpp_f<kDppRor8>(v));
    v = fminf(v, dpp_rows_f<kDppBcast15, 0xA>(v)); v = fminf(v, dpp_rows_f<kDppBcast31, 0xC>(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max_q(float v)
{
    v = fmaxf(v, dpp_f<kDppRor4>(v)); v = fmaxf(v, dpp_f<kDppRor8>(v));
    v = fmaxf(v, dpp_rows_f<kDppBcast15, 0xA>(v)); v = fmaxf(v, dpp_rows_f<kDppBcast31, 0xC>(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// a global load at base + a 32-bit byte offset (scalar base + one vector register: no 64-bit vector arithmetic per address)
template <class T> __device__ __forceinline__ T ld_at(const void* base, unsigned int byte_off)
{
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ float bperm_f(int byte_addr, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_addr, __float_as_int(v))); }

// The row of 24 unit-weight sums of 64 pairings whose records [m, m lx, m ly, m lz, m gx, m gy, m gz, m d2] lie in LDS (record r
// = query r of the row): the SAME matrix-core sequence as item_row_mfma (kernels_tiled.hpp) -- pairing 32 h + 16 t + rec goes into
// the (h, t)-th of the four 4x4x4 products, the four blocks are added in block order -- hence the same bits.  One wave; `sd` = 128
// doubles of wave-private LDS.
__device__ __forceinline__ void item_row_from_records(const float* __restrict__ recs, double* __restrict__ sd, int lane, double* __restrict__ row)
{
    const int e = lane & 3, rec = (((lane >> 2) & 3) << 2) + (lane >> 4);
    const int oA = rec * 8 + e;                     // u  = [m, l]
    const int oB = rec * 8 + (e ? 3 + e : 0);       // v  = [m, g]
    const int oA2 = rec * 8 + (e < 3 ? 1 + e : 7);  // u' = [l, d2]
    const int oB2 = rec * 8 + (e < 3 ? 1 + e : 0);  // v' = [l, m]
    double D1 = 0.0, D2 = 0.0;
#pragma unroll
    for (int ht = 0; ht < 4; ++ht) {
        const int base = 16 * ht * 8;
        const double a = (double)recs[base + oA], b = (double)recs[base + oB];
        const double a2 = (double)recs[base + oA2], b2 = (double)recs[base + oB2];
        D1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, D1, 0, 0, 0);
        D2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a2, b2, D2, 0, 0, 0);
    }
    sd[lane] = D1;
    sd[64 + lane] = D2;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < 24) {
        int src;
        if (lane == 0 || lane == 16) src = 0;
        else if (lane < 4) src = 16 * lane;
        else if (lane < 7) src = lane - 3;
        else if (lane < 16) src = 16 * (1 + (lane - 7) / 3) + 1 + (lane - 7) % 3;
        else if (lane == 17) src = 64 + 51;
        else if (lane < 21) src = 64 + (lane - 18);
        else if (lane < 23) src = 64 + 16 + 1 + (lane - 21);
        else src = 64 + 32 + 2;
        row[lane] = ((sd[src] + sd[src + 4]) + sd[src + 8]) + sd[src + 12];
    }
}

struct Q4Result {
    int rpos, roi;      // sorted-map position / original index of the neighbour, -1: none inside the gate
    float rd;           // its squared distance (gate^2 if none)
    float gx, gy, gz;   // its coordinates
};

// scalar bit scans the compiler wraps in a compare + select when written in C: s_ff1_i32_b64 returns -1 for an empty mask by itself,
// and clearing bit (-1 & 63) of an EMPTY mask changes nothing
__device__ __forceinline__ int sff1_b64(unsigned long long m)
{
    int r;
    asm("s_ff1_i32_b64 %0, %1" : "=s"(r) : "s"(m));
    return r;
}
__device__ __forceinline__ void sbitset0_b64(unsigned long long& m, int bit) { asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(bit)); }

// One pass of a wave over its 16 queries with the fast (EX = false) or the packed-key visitor.  Returns (wave-uniform) whether a
// query met an exact distance tie the fast visitor cannot resolve.  `tiles` counts the evaluated tiles (statistics).
template <bool EX>
__device__ __forceinline__ bool q4_pass(const NnProblem& pb, const TiledMap& mp, const lds_f32* lbox /*LDS copy of the upper box levels*/, bool use_lbox,
                                        float* ring /*wave-private, kQ4RingFloats*/, int lane, bool valid, int ic, float qx, float qy, float qz,
                                        int js, float sd, Q4Result& res, unsigned int& tiles, unsigned long long* dbg_w = nullptr)
{
    const float thr2 = pb.thr2;
    const int s = lane & 3, s8 = 8 * s;
    float best = thr2;
    int bpos = -1, cnt = 0;
    unsigned long long key = ((unsigned long long)__float_as_uint(thr2) << 32);   // EX: packed (d2, original index); (gate^2, 0) = "no neighbour"
    const bool seeded = js >= 0 && sd < thr2;
    if (seeded) {   // warm start: last launch's neighbour is an exact candidate
        best = sd;
        bpos = EX ? js : (js & ~(kGroup - 1));
        if (EX) key = ((unsigned long long)__float_as_uint(sd) << 32) | (unsigned int)pb.idx_s[ic];   // (one more trip on this rare path)
    }
    if (!valid) { best = -1.0f; bpos = -1; key = ((unsigned long long)__float_as_uint(thr2) << 32); }   // padding lane: reaches nothing

    // ---- the lists: super-tiles some query reaches, then the tiles; entry n of a list sits in lane n of one register ----
    int sl = -1, n_sl = 0, tl = -1, n_tl = 0;
    const unsigned int arr_stride = (unsigned int)(mp.sy - mp.sx);   // elements between the x, y and z rows of the sorted map
    const unsigned int trow = (unsigned int)mp.n_tiles_p * 4u;       // bytes between the rows of the tile boxes

    // tiles [first, first + 4) of the list (entries of -1: none) -> ring bank `bank`: one instruction per two tiles (24 lanes x 16 bytes
    // each: chunk c of a tile = floats 4 (c & 7) .. of row c >> 3).  Returns the instructions issued (wave-uniform).
    const int lane_c = lane >= 24 ? lane - 24 : lane;   // chunk of a tile this lane carries (lanes 0..23: the first tile of a pair, 24..47: the second)
    const unsigned int lane_off = ((unsigned int)(lane_c >> 3) * arr_stride + (unsigned int)(lane_c & 7) * 4u) * 4u;
    const int lane_e = lane >= 24 ? 1 : 0;
    auto issue_bank = [&](int first, int bank) -> int {
        int issued = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (first + 2 * h < n_tl) {   // (wave-uniform)
                const int e = first + 2 * h + lane_e;
                const int t = __builtin_amdgcn_ds_bpermute((e & 63) << 2, tl);
                const unsigned int off = (unsigned int)t * (unsigned int)(kTileG * 4) + lane_off;
                if (lane < 48 && e < n_tl && t >= 0)
                    __builtin_amdgcn_global_load_lds(reinterpret_cast<const char*>(mp.sx) + off, ring + bank * kQ4BankFloats + h * 2 * kQ4TileFloats, 16, 0, 0);
                ++issued;
            }
        }
        return issued;
    };

    // ---- the seeds' tiles are swept whatever the scan finds (a seed's tile holds a point AT the query's bound), so they are listed
    // and sent for FIRST: their points fly under the wave box, the scan of the box levels and the tile tests; and the tile boxes of the
    // first seed's super-tile -- the one the scan will most likely list first -- are sent for as well.  What the tests add later
    // lands while these tiles are evaluated: the item's chain loses a round trip (tile boxes) and most of another (tile points).
    int n_pre = 0, pre0 = -1, pre1 = -1, pre2 = -1, pre3 = -1, S_spec = -1;
    float f0 = 0.f, f1 = 0.f, f2 = 0.f, f3 = 0.f, f4 = 0.f, f5 = 0.f;   // tile boxes of the next super-tile to test (one per lane)
    if constexpr (!EX) {
        if (pb.use_seed) {   // (wave-uniform)
            const int tq = (valid && seeded) ? (js >> 5) : -1;
            unsigned long long rem = __ballot(tq >= 0) & 0x1111111111111111ull;   // one lane per query
            while (rem && n_tl < kQ4Bank) {
                const int t = __builtin_amdgcn_readlane(tq, sff1_b64(rem));
                rem &= ~__ballot(tq == t);
                tl = lane == n_tl ? t : tl;
                pre3 = n_tl == 3 ? t : pre3; pre2 = n_tl == 2 ? t : pre2; pre1 = n_tl == 1 ? t : pre1; pre0 = n_tl == 0 ? t : pre0;
                ++n_tl;
            }
            n_pre = n_tl;
            if (n_pre) {
                (void)issue_bank(0, 0);
                n_tl = kQ4Bank;   // (the seeds' tiles own bank 0 -- its unused entries stay -1 -- so that what the tests list starts a bank of its own
                                  //  and flies while bank 0, which landed long ago, is evaluated)
                S_spec = pre0 >> 6;
                const unsigned int ti = (unsigned int)(S_spec * kSuper + lane) * 4u;
                f0 = ld_at<float>(mp.tbox, ti); f1 = ld_at<float>(mp.tbox, ti + trow); f2 = ld_at<float>(mp.tbox, ti + 2u * trow);
                f3 = ld_at<float>(mp.tbox, ti + 3u * trow); f4 = ld_at<float>(mp.tbox, ti + 4u * trow); f5 = ld_at<float>(mp.tbox, ti + 5u * trow);
            }
        }
    }
    Q4_STAMP(8);
    bool bank0_in_flight = n_pre > 0;   // the seeds' tiles were sent for above

    // ---- the wave box: union of the 16 query boxes [q - r, q + r], in scalar registers ----
    // (reach_of with the hardware's square root -- 1 ulp -- under the same 1e-5 relative margin: any m with d2 <= best lies inside [q - r, q + r])
    float reach = __builtin_amdgcn_sqrtf(best * 1.000002f) * 1.00001f + fmaxf(fabsf(qx), fmaxf(fabsf(qy), fabsf(qz))) * 2.4e-7f + 1e-30f;
    if (!valid) reach = -1.0f;
    Box w;
    w.lo[0] = wave_min_q(reach >= 0.f ? qx - reach : INFINITY); w.hi[0] = wave_max_q(reach >= 0.f ? qx + reach : -INFINITY);
    w.lo[1] = wave_min_q(reach >= 0.f ? qy - reach : INFINITY); w.hi[1] = wave_max_q(reach >= 0.f ? qy + reach : -INFINITY);
    w.lo[2] = wave_min_q(reach >= 0.f ? qz - reach : INFINITY); w.hi[2] = wave_max_q(reach >= 0.f ? qz + reach : -INFINITY);

    Q4_STAMP(9);
    // the query's live bound: the minimum over its four sub-lanes' bests (all >= 0, or -1 in a padding lane: their bit patterns order as
    // integers, and an integer minimum needs no canonicalising instruction in front of it)
    auto live_bound = [&]() -> float {
        int v = __float_as_int(best);
        v = min(v, dpp_i<kDppXor1>(v));
        v = min(v, dpp_i<kDppXor2>(v));
        return __int_as_float(v);
    };
    // ---- one group of box tests: the next (up to) four candidates of `cand` -- bit i = box i of the 64 the lanes hold, one per
    // lane, in r0..r5 -- sub-lane s of every query takes candidate s; on_pass(i) for every box some query reaches under `bound`
    auto test4 = [&](float r0, float r1, float r2, float r3, float r4, float r5, unsigned long long& cand, float bound, auto&& on_pass) {
        const int c0 = sff1_b64(cand); sbitset0_b64(cand, c0);
        const int c1 = sff1_b64(cand); sbitset0_b64(cand, c1);
        const int c2 = sff1_b64(cand); sbitset0_b64(cand, c2);
        const int c3 = sff1_b64(cand); sbitset0_b64(cand, c3);
        // the four candidates, a byte each (0xff: none) -- a scalar; sub-lane s takes byte s
        const unsigned int pack = ((unsigned int)c0 & 0xffu) | (((unsigned int)c1 & 0xffu) << 8) | (((unsigned int)c2 & 0xffu) << 16) | ((unsigned int)c3 << 24);
        const unsigned int sel = __builtin_amdgcn_ubfe(pack, (unsigned int)s8, 8u);
        const int src = (int)(sel << 2);   // (0xff: lane 63's box, the result is masked below)
        const float m0 = bperm_f(src, r0), m1 = bperm_f(src, r1), m2 = bperm_f(src, r2);
        const float m3 = bperm_f(src, r3), m4 = bperm_f(src, r4), m5 = bperm_f(src, r5);
        // squared distance from the query to the box in the contract's own arithmetic (tiled_sweep::any_reach: exact, no margins)
        const float ax = qx - __builtin_amdgcn_fmed3f(qx, m0, m3);
        const float ay = qy - __builtin_amdgcn_fmed3f(qy, m1, m4);
        const float az = qz - __builtin_amdgcn_fmed3f(qz, m2, m5);
        const unsigned long long m = __ballot(sel != 0xffu && fmaf(az, az, fmaf(ay, ay, ax * ax)) <= bound);
        // candidate j passed iff a lane with sub-lane j is set: fold the 64 bits onto the lowest four
        unsigned int P = (unsigned int)m | (unsigned int)(m >> 32);
        P |= P >> 16; P |= P >> 8; P |= P >> 4;
        if (P & 1u) on_pass(c0);
        if (P & 2u) on_pass(c1);
        if (P & 4u) on_pass(c2);
        if (P & 8u) on_pass(c3);
    };

    // the lane's 8 points of one tile (LDS: x[32] y[32] z[32]) against its query
    auto eval_tile = [&](int t, const float* b) {
        const float* p = b + 8 * s;
        const int gpos = t * kTileG + 8 * s;   // sorted position of the lane's group
        if constexpr (EX) {
            // (the rare path: point by point from LDS -- eight points' keys at once set the whole kernel's register count)
            const int4 P0 = ld_at<int4>(mp.perm, (unsigned int)gpos * 4u), P1 = ld_at<int4>(mp.perm, (unsigned int)gpos * 4u + 16u);
            const unsigned int os[8] = {(unsigned int)P0.x, (unsigned int)P0.y, (unsigned int)P0.z, (unsigned int)P0.w,
                                        (unsigned int)P1.x, (unsigned int)P1.y, (unsigned int)P1.z, (unsigned int)P1.w};
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float d = dist2(qx, qy, qz, p[u], p[32 + u], p[64 + u]);
                const unsigned long long ck = ((unsigned long long)__float_as_uint(d) << 32) | os[u];
                const bool better = ck < key;
                key = better ? ck : key;
                best = better ? d : best;   // (the box tests read it)
                bpos = better ? gpos + u : bpos;
                __builtin_amdgcn_sched_barrier(0);   // (keep the points in sequence)
            }
        } else {
            const float4 X0 = *reinterpret_cast<const float4*>(p), X1 = *reinterpret_cast<const float4*>(p + 4);
            const float4 Y0 = *reinterpret_cast<const float4*>(p + 32), Y1 = *reinterpret_cast<const float4*>(p + 36);
            const float4 Z0 = *reinterpret_cast<const float4*>(p + 64), Z1 = *reinterpret_cast<const float4*>(p + 68);
            const float xs[8] = {X0.x, X0.y, X0.z, X0.w, X1.x, X1.y, X1.z, X1.w};
            const float ys[8] = {Y0.x, Y0.y, Y0.z, Y0.w, Y1.x, Y1.y, Y1.z, Y1.w};
            const float zs[8] = {Z0.x, Z0.y, Z0.z, Z0.w, Z1.x, Z1.y, Z1.z, Z1.w};
            float gm = INFINITY;
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                const v2f mx = {xs[u], xs[u + 1]}, my = {ys[u], ys[u + 1]}, mz = {zs[u], zs[u + 1]};
                const v2f dd = dist2_pk2(qx, qy, qz, mx, my, mz);
                gm = fminf(fminf(gm, dd.x), dd.y);
            }
            // (best, its group, how many groups reached that best): nn_visit_fast's bookkeeping, one group per lane and tile
            const bool lt = gm < best;
            const int eq = (int)(gm == best);
            cnt = lt ? 1 : cnt + eq;
            bpos = lt ? gpos : bpos;
            best = fminf(best, gm);
        }
    };
    // evaluate the listed tiles: bank c + 1 in flight under bank c's distances
    auto run_tiles = [&]() {
        if (n_tl == 0) return;
        if (!bank0_in_flight) (void)issue_bank(0, 0);
        bank0_in_flight = false;
        for (int c = 0; c * kQ4Bank < n_tl; ++c) {
            const int nxt = (c + 1) * kQ4Bank < n_tl ? issue_bank((c + 1) * kQ4Bank, (c + 1) & 1) : 0;
            // (loads land in order: all but the `nxt` just issued have)
            if (nxt == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if (nxt == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const float* bank = ring + (c & 1) * kQ4BankFloats;
#pragma unroll 1
            for (int k = 0; k < kQ4Bank; ++k) {   // (one tile at a time: four tiles' points in registers at once cost the kernel two waves per SIMD)
                const int e = c * kQ4Bank + k;
                if (e >= n_tl) break;
                const int t = __builtin_amdgcn_readlane(tl, e);
                if (t >= 0) { eval_tile(t, bank + k * kQ4TileFloats); ++tiles; }
            }
            __builtin_amdgcn_wave_barrier();   // (this bank is the target of the loads issued in the next iteration)
        }
        n_tl = 0;
    };

    // ---- the scan: top boxes -> super-tile boxes (listed), then the listed super-tiles' tile boxes -> tiles (listed, evaluated).
    // One resumable state machine so that run_tiles() has ONE call site (its body is the distance math).
    const lds_f32* l_ubox = lbox;
    const lds_f32* l_sbox = lbox + 6 * mp.n_top;
    int ub = 0, sb = 0;
    unsigned long long ucand = 0ull, scand = 0ull;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f, c4 = 0.f, c5 = 0.f;
    bool c_valid = false, scan_done = false;
    if (mp.n_top == 1) { ucand = 1ull; ub = 64; }   // (one top box -- any map up to 131 072 points -- holds everything: no test)
    auto load_super_boxes = [&]() {
        const int si = sb + lane;
        if (use_lbox) {
            c0 = l_sbox[si]; c1 = l_sbox[mp.n_super + si]; c2 = l_sbox[2 * mp.n_super + si];
            c3 = l_sbox[3 * mp.n_super + si]; c4 = l_sbox[4 * mp.n_super + si]; c5 = l_sbox[5 * mp.n_super + si];
        } else {
            const unsigned int o = (unsigned int)si * 4u, r = (unsigned int)mp.n_super * 4u;
            c0 = ld_at<float>(mp.sbox, o); c1 = ld_at<float>(mp.sbox, o + r); c2 = ld_at<float>(mp.sbox, o + 2u * r);
            c3 = ld_at<float>(mp.sbox, o + 3u * r); c4 = ld_at<float>(mp.sbox, o + 4u * r); c5 = ld_at<float>(mp.sbox, o + 5u * r);
        }
        c_valid = true;
    };
    for (;;) {
        Q4_STAMP(2);
        // (1) the upper levels, until the super-tile list is (nearly) full or the scan has ended
        {
            const float bound = live_bound();
            while (!scan_done && n_sl <= kQ4ListCap - 4) {
                if (scand) {
                    if (!c_valid) load_super_boxes();   // resumed after a full list
                    test4(c0, c1, c2, c3, c4, c5, scand, bound, [&](int i) { sl = lane == n_sl ? sb + i : sl; ++n_sl; });
                } else if (ucand) {
                    sb = (ub - 64 + sff1_b64(ucand)) * 64;   // first super-tile of this top box (ub already advanced)
                    ucand &= ucand - 1ull;
                    load_super_boxes();
                    scand = __ballot(c0 <= w.hi[0] && c1 <= w.hi[1] && c2 <= w.hi[2] && c3 >= w.lo[0] && c4 >= w.lo[1] && c5 >= w.lo[2]);
                } else if (ub < mp.n_top) {
                    const int ui = ub + lane;
                    float u0 = INFINITY, u1 = INFINITY, u2 = INFINITY, u3 = -INFINITY, u4 = -INFINITY, u5 = -INFINITY;
                    if (ui < mp.n_top) {
                        if (use_lbox) {
                            u0 = l_ubox[ui]; u1 = l_ubox[mp.n_top + ui]; u2 = l_ubox[2 * mp.n_top + ui];
                            u3 = l_ubox[3 * mp.n_top + ui]; u4 = l_ubox[4 * mp.n_top + ui]; u5 = l_ubox[5 * mp.n_top + ui];
                        } else {
                            const unsigned int o = (unsigned int)ui * 4u, r = (unsigned int)mp.n_top * 4u;
                            u0 = ld_at<float>(mp.ubox, o); u1 = ld_at<float>(mp.ubox, o + r); u2 = ld_at<float>(mp.ubox, o + 2u * r);
                            u3 = ld_at<float>(mp.ubox, o + 3u * r); u4 = ld_at<float>(mp.ubox, o + 4u * r); u5 = ld_at<float>(mp.ubox, o + 5u * r);
                        }
                    }
                    ucand = __ballot(u0 <= w.hi[0] && u1 <= w.hi[1] && u2 <= w.hi[2] && u3 >= w.lo[0] && u4 >= w.lo[1] && u5 >= w.lo[2]);
                    if (__builtin_popcountll(ucand) > 4) {   // a spread query group: the top boxes per query first (see tiled_sweep)
                        unsigned long long uc = ucand, keep = 0ull;
                        while (uc) test4(u0, u1, u2, u3, u4, u5, uc, bound, [&](int i) { keep |= 1ull << i; });
                        ucand = keep;
                    }
                    ub += 64;
                } else {
                    scan_done = true;
                }
            }
        }
        // (2) the listed super-tiles: tile boxes of entry e + 1 in flight while entry e's tiles are tested; the tiles that pass are
        // listed and evaluated when the scan of the list ends or the tile list is (nearly) full
        Q4_STAMP(3);
        if (n_sl || n_tl) {   // (n_tl without n_sl: seeds' tiles with no super-tile listed -- a seed's super-tile always is; nothing may stay unevaluated)
            c_valid = false;   // (the super-tile boxes are not held through the tiles' phase: a resumed scan reads them again)
            int e_sl = 0, S = __builtin_amdgcn_readlane(sl, 0), Sc = 0;
            if (n_sl && S != S_spec) {   // (not the super-tile whose tile boxes were sent for with the seeds' tiles)
                const unsigned int ti = (unsigned int)(S * kSuper + lane) * 4u;
                f0 = ld_at<float>(mp.tbox, ti); f1 = ld_at<float>(mp.tbox, ti + trow); f2 = ld_at<float>(mp.tbox, ti + 2u * trow);
                f3 = ld_at<float>(mp.tbox, ti + 3u * trow); f4 = ld_at<float>(mp.tbox, ti + 4u * trow); f5 = ld_at<float>(mp.tbox, ti + 5u * trow);
            }
            S_spec = -1;   // (the prefetched boxes serve once)
            float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f, b4 = 0.f, b5 = 0.f;
            unsigned long long cand = 0ull;
            float bound = live_bound();
            do {
                while (n_tl <= kQ4ListCap - 4 && (cand || e_sl < n_sl)) {
                    if (cand) {
                        test4(b0, b1, b2, b3, b4, b5, cand, bound, [&](int i) { tl = lane == n_tl ? Sc * kSuper + i : tl; ++n_tl; });
                    } else {
                        b0 = f0; b1 = f1; b2 = f2; b3 = f3; b4 = f4; b5 = f5;
                        Sc = S;
                        ++e_sl;
                        if (e_sl < n_sl) {
                            S = __builtin_amdgcn_readlane(sl, e_sl);
                            const unsigned int ti = (unsigned int)(S * kSuper + lane) * 4u;
                            f0 = ld_at<float>(mp.tbox, ti); f1 = ld_at<float>(mp.tbox, ti + trow); f2 = ld_at<float>(mp.tbox, ti + 2u * trow);
                            f3 = ld_at<float>(mp.tbox, ti + 3u * trow); f4 = ld_at<float>(mp.tbox, ti + 4u * trow); f5 = ld_at<float>(mp.tbox, ti + 5u * trow);
                        }
                        // (the seeds' tiles are listed already: a tile evaluated twice would count its groups twice)
                        const int mine = Sc * kSuper + lane;
                        const bool listed = mine == pre0 || mine == pre1 || mine == pre2 || mine == pre3;
                        cand = __ballot(!listed && b0 <= w.hi[0] && b1 <= w.hi[1] && b2 <= w.hi[2] && b3 >= w.lo[0] && b4 >= w.lo[1] && b5 >= w.lo[2]);
                    }
                }
                Q4_STAMP(4);
                run_tiles();
                Q4_STAMP(5);
                bound = live_bound();   // (a resumed list is tested under the bounds the evaluated tiles left)
            } while (cand || e_sl < n_sl);
            n_sl = 0;
        }
        if (scan_done) break;
    }
    // ---- close the query's four partial results (two DPP steps), resolve inside the winning group ----
    res.rpos = -1; res.roi = -1; res.rd = thr2; res.gx = res.gy = res.gz = 0.f;
    bool any_tie = false;
    if constexpr (EX) {
        int p = bpos;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const unsigned int lo = (unsigned int)(key & 0xffffffffull), hi = (unsigned int)(key >> 32);
            const unsigned int lo2 = (unsigned int)(st ? dpp_i<kDppXor2>((int)lo) : dpp_i<kDppXor1>((int)lo));
            const unsigned int hi2 = (unsigned int)(st ? dpp_i<kDppXor2>((int)hi) : dpp_i<kDppXor1>((int)hi));
            const int p2 = st ? dpp_i<kDppXor2>(p) : dpp_i<kDppXor1>(p);
            const unsigned long long k2 = ((unsigned long long)hi2 << 32) | lo2;
            const bool better = k2 < key;   // equal keys = the same point
            key = better ? k2 : key;
            p = better ? p2 : p;
        }
        const float d = __uint_as_float((unsigned int)(key >> 32));
        if (d < thr2 && valid) {
            res.rd = d; res.rpos = p; res.roi = (int)(unsigned int)(key & 0xffffffffull);
            res.gx = mp.sx[p]; res.gy = mp.sx[arr_stride + p]; res.gz = mp.sx[2u * arr_stride + p];   // (rare path: one more trip)
        }
    } else {
        float b = best;
        int p = bpos, t = cnt;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const float b2 = st ? dpp_f<kDppXor2>(b) : dpp_f<kDppXor1>(b);
            const int p2 = st ? dpp_i<kDppXor2>(p) : dpp_i<kDppXor1>(p);
            const int t2 = st ? dpp_i<kDppXor2>(t) : dpp_i<kDppXor1>(t);
            const bool lt = b2 < b, eq = b2 == b;
            // t = number of groups at the best (every group has ONE owner lane, so the counts add up; the seed starts at 0 and its
            // own group counts itself): two groups at an equal minimum are a tie the exact visitor must resolve
            t = lt ? t2 : (eq ? t + t2 : t);
            p = lt ? p2 : p;
            b = fminf(b, b2);
        }
        // the winning group's 8 points, two per sub-lane: the point(s) with d2 == best, lowest original index first
        const int bp = (p >= 0 ? p : 0) + 2 * s;
        const unsigned int bo4 = (unsigned int)bp * 4u;
        const float2 RX = ld_at<float2>(mp.sx, bo4), RY = ld_at<float2>(mp.sx, bo4 + arr_stride * 4u), RZ = ld_at<float2>(mp.sx, bo4 + arr_stride * 8u);
        const int2 RP = ld_at<int2>(mp.perm, bo4);
        unsigned int bo = 0xffffffffu;
        int pos = -1;
        float wx = 0.f, wy = 0.f, wz = 0.f;
        {
            const float d0 = dist2(qx, qy, qz, RX.x, RY.x, RZ.x), d1 = dist2(qx, qy, qz, RX.y, RY.y, RZ.y);
            const bool t0 = d0 == b;
            bo = t0 ? (unsigned int)RP.x : bo; pos = t0 ? bp : pos; wx = t0 ? RX.x : wx; wy = t0 ? RY.x : wy; wz = t0 ? RZ.x : wz;
            const bool t1 = d1 == b && (unsigned int)RP.y < bo;
            bo = t1 ? (unsigned int)RP.y : bo; pos = t1 ? bp + 1 : pos; wx = t1 ? RX.y : wx; wy = t1 ? RY.y : wy; wz = t1 ? RZ.y : wz;
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const unsigned int bo2 = (unsigned int)(st ? dpp_i<kDppXor2>((int)bo) : dpp_i<kDppXor1>((int)bo));
            const int pos2 = st ? dpp_i<kDppXor2>(pos) : dpp_i<kDppXor1>(pos);
            const float wx2 = st ? dpp_f<kDppXor2>(wx) : dpp_f<kDppXor1>(wx);
            const float wy2 = st ? dpp_f<kDppXor2>(wy) : dpp_f<kDppXor1>(wy);
            const float wz2 = st ? dpp_f<kDppXor2>(wz) : dpp_f<kDppXor1>(wz);
            const bool tk = bo2 < bo;
            bo = tk ? bo2 : bo; pos = tk ? pos2 : pos; wx = tk ? wx2 : wx; wy = tk ? wy2 : wy; wz = tk ? wz2 : wz;
        }
        if (p >= 0 && valid) {
            res.rd = b; res.rpos = pos; res.roi = (int)bo;
            res.gx = wx; res.gy = wy; res.gz = wz;
            if (pos < 0) { t = 2; res.roi = -1; }   // cannot happen (same arithmetic); be safe: exact pass
        }
        any_tie = valid && t >= 2;
    }
    Q4_STAMP(6);
    return !EX && __any(any_tie);
}

// grid = (workgroups of 64 queries, problems); a workgroup without queries leaves at once.  `count_pairs`: statistics on.
template <int KMAX>
__global__ __launch_bounds__(256, kQ4WorkgroupsPerCu) void k_nn_q4(const NnBatch<KMAX> batch, int lds_boxes, int count_pairs, unsigned long long* __restrict__ dbg /*diagnostic build only, else null*/)
{
    __shared__ __attribute__((aligned(16))) float s_ring[4][kQ4RingFloats];   // per wave: two banks of four tiles
    __shared__ __attribute__((aligned(16))) float s_rec[64 * 8];              // the row's 64 pairing records
    __shared__ int s_done;                                                    // waves that have written their records
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];             // the upper box levels, if they fit
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (a scalar: the ring is a scalar base, M0 needs no read-back)
    const NnProblem& pb = batch.p[KMAX == 1 ? 0 : blockIdx.y];
    const int N = pb.N;
    const int item = lds_boxes ? (int)blockIdx.x : xcd_item((int)blockIdx.x, (N + 63) / 64);   // (see k_nn_coop)
    if (item * 64 >= N) return;   // nothing for this workgroup (uniform: before any barrier)
    const TiledMap mp = pb.mp;
    float* ring = &s_ring[wave][0];
#ifdef MOLA_Q4_DIAG
    unsigned long long* dbg_w = dbg && blockIdx.y == 0 && blockIdx.x < 2048 ? dbg + 16 * (size_t)(blockIdx.x * 4 + wave) : nullptr;
    if (lane != 0) dbg_w = nullptr;
#else
    unsigned long long* dbg_w = nullptr;
    (void)dbg;
#endif
    Q4_STAMP(0);
    if (lane == 0) ring[0] = 0.f;   // (a plain store to the ring: the tiles arrive by LDS-DMA, which the compiler does not count as one)
    if (threadIdx.x == 0) s_done = 0;

    // round trip A: the lane's query (four lanes read the same words: one fetch) and its seed with coordinates
    const int q = lane >> 2;
    const int qi = item * 64 + wave * 16 + q;
    const bool valid = qi < N;
    const int ic = valid ? qi : N - 1;
    const unsigned int ic4 = (unsigned int)ic * 4u;
    const float lx = ld_at<float>(pb.slx, ic4), ly = ld_at<float>(pb.sly, ic4), lz = ld_at<float>(pb.slz, ic4);
    int js = -1;
    float gsx = 0.f, gsy = 0.f, gsz = 0.f;
    if (pb.use_seed) {
        js = ld_at<int>(pb.pos_s, ic4);
        gsx = ld_at<float>(pb.gsx, ic4); gsy = ld_at<float>(pb.gsy, ic4); gsz = ld_at<float>(pb.gsz, ic4);
    }
    // (the query's own coordinates wait in its record for the row: not held in registers through the sweep, not re-read behind it)
    float* rec = s_rec + (wave * 16 + q) * 8;
    if (lds_boxes) load_boxes_to_lds(mp, (lds_f32*)s_dyn);   // (ends with a barrier)
    else __syncthreads();                                    // (s_done's zero is in place before any wave's ticket)
    Q4_STAMP(1);
    if ((lane & 3) == 0) { rec[1] = lx; rec[2] = ly; rec[3] = lz; }   // (behind the box levels' loads: the first use of the query's own)
    float qx, qy, qz;
    xform(pb.P, lx, ly, lz, qx, qy, qz);
    const float sd = dist2(qx, qy, qz, gsx, gsy, gsz);
    if (!valid) qx = qy = qz = 1.0e18f;   // padding lane

    Q4Result res;
    unsigned int tiles = 0u;
    if (q4_pass<false>(pb, mp, (const lds_f32*)s_dyn, lds_boxes != 0, ring, lane, valid, ic, qx, qy, qz, js, sd, res, tiles, dbg_w))
        (void)q4_pass<true>(pb, mp, (const lds_f32*)s_dyn, lds_boxes != 0, ring, lane, valid, ic, qx, qy, qz, js, sd, res, tiles);

    // ---- the query's record for the row, then the pairing (sorted query order) ----
    const bool paired = valid && res.rpos >= 0;
    if ((lane & 3) == 0) {
        // (plain float stores, the type item_row_from_records reads)
        rec[0] = paired ? 1.0f : 0.0f;
        if (!paired) { rec[1] = 0.0f; rec[2] = 0.0f; rec[3] = 0.0f; }
        rec[4] = paired ? res.gx : 0.0f; rec[5] = paired ? res.gy : 0.0f; rec[6] = paired ? res.gz : 0.0f;
        rec[7] = paired ? res.rd : 0.0f;
    }
    // The row of the workgroup's 64 queries -- the SAME row, bit for bit, that k_nn_tiled / k_nn_coop write for them -- is formed by
    // whichever wave finishes LAST (a ticket in LDS: no wave waits at a barrier for the slowest of the four, and its slot is free at once)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    int ticket = 0;
    if (lane == 0) ticket = atomicAdd(&s_done, 1);
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    if (ticket == 3) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        item_row_from_records(s_rec, reinterpret_cast<double*>(ring), lane, pb.rows + (size_t)item * kNAcc);
    }
    if ((lane & 3) == 0 && valid) {
        pb.pos_s[qi] = res.rpos;
        pb.idx_s[qi] = res.rpos >= 0 ? res.roi : -1;
        pb.d2_s[qi] = res.rd;
        pb.gsx[qi] = res.gx; pb.gsy[qi] = res.gy; pb.gsz[qi] = res.gz;
    }
    if (count_pairs && lane == 0 && tiles)   // executed work in units of 64 (query, point) pairs: a tile = 32 points x 16 queries; slotted
        atomicAdd(pb.staged + (size_t)((blockIdx.x * 4 + wave) & (kStatSlots - 1)) * kStatStride, (unsigned long long)tiles * 8ull);
    Q4_STAMP(7);
}

}  // namespace mola_icp_amd
