// kernels_coop.hpp -- k_nn_coop: the tiled matcher for SMALL and BATCHED problems (odometry-size clouds, the
// loop-closure Monte-Carlo, the nearby-keyframe batch), with the unit-weight accumulation fused in.
// Device code of the ICP core for gfx950; included by hip_backend.hip only.  Numeric contract: hip_backend.hip / DESIGN.md.
//
// k_nn_tiled gives every persistent wave whole 128-query items; with <= ~0.4M queries there are fewer items than
// wave slots, a launch is ONE item long and that item is a chain of dependent memory round trips on a lone wave.
// Here ONE WORKGROUP owns one item: its four waves hold the same 128 queries, run the same (cheap) box scan and
// deal the candidate tiles among themselves (coop_sweep); the four partial results are merged per
// query through LDS and waves 0/1 finish 64 queries each.  No work queue, no atomics, no persistent loop: grid =
// (items, problems), a block that has no item leaves at once.  blockIdx.y selects the problem, so K independent
// problems -- K initial poses on one cloud pair (src/LidarOdometry.cpp:767-788) or K different pairs (cpp:704-741)
// -- share one launch.
//
// The item is a chain of dependent round trips (each ~1 us when the whole grid issues it at once), so the chain is
// kept short:
//   A  queries + last launch's neighbour (position, original index, COORDINATES -- the epilogue stores them, so the
//      seed distance needs no second trip) + the upper box levels into LDS, all issued together;
//   B  tile boxes of the listed super-tiles (the next entry's boxes in flight);  C  the candidate tiles' points;
//   D  the winning 8-point group, re-read once to resolve the exact point.
// Exact distance ties (duplicate points, lattices) are rare: the workgroup that meets one redoes ITS item with the
// exact-key visitor in the same launch (no second kernel, no redo queue).
// Fused accumulation (row a8, stage 0): the finishing lanes hold (l, g, d2) of their pairing; the workgroup sums the
// 24 unit-weight terms in a fixed order and writes ONE row per item -- the first accumulation pass of an iteration is
// then only the fixed-order row reduction (k_reduce_partials), no second pass over the pairing.
// Results are bit-identical to k_nn_tiled / the dense kernels / the CPU checker (same contract, same tie rule).
#pragma once
#include "kernels_tiled.hpp"

namespace mola_icp_amd {

struct NnProblem {
    const float *slx, *sly, *slz;  // Hilbert-sorted local cloud (queries)
    int N;
    TiledMap mp;                   // Hilbert-sorted map + box levels
    PoseF P;
    float thr2;
    int use_seed;                  // pos_s / idx_s / g_s hold the previous launch's neighbours
    int* pos_s;                    // in: seeds, out: neighbour's sorted-map position   } the pairing, in SORTED query order
    int* idx_s;                    // in/out: neighbour's original map index             }
    float* d2_s;                   // out: squared distance                              }
    float *gsx, *gsy, *gsz;        // in/out: the neighbour's coordinates (as stored in the map)
    double* rows;                  // out: kNAcc unit-weight sums per item (fused stage-0 accumulation)
    unsigned long long* staged;    // statistics: kStatSlots counters on separate lines (units of 64 evaluated pairs)
};
constexpr int kCoopMaxBatch = 12;                 // problems per launch (kernel arguments are limited to 4 KB; = kAccMaxBatch)
template <int KMAX> struct NnBatch { NnProblem p[KMAX]; };

constexpr int kCoopParts = 4;  // waves of a workgroup = parts of an item

// what the two finishing waves hold for their 64 queries after the merge
struct CoopResult {
    int rpos, roi;      // sorted-map position / original index of the neighbour, -1: none inside the gate
    float rd;           // its squared distance (gate^2 if none)
    float gx, gy, gz;   // its coordinates
};

// The cooperative sweep.  Same tests and staging as tiled_sweep (kernels_tiled.hpp), split differently:
//   * wave 0 alone derives the wave box and walks the two upper box levels (LDS copy), listing the super-tiles some
//     query reaches in the workgroup's shared list -- the other waves wait at the barrier instead of repeating it (the
//     SIMDs' issue slots are what this kernel is short of: 57 % of all cycles issuing at 100k x 100k);
//   * all four waves stream the list; wave `part` owns the tiles t of super-tile S with (t + S) % 4 == part -- a fixed
//     function of the ids, so a wave visits only its own candidate bits (no walk over the others') and the waves,
//     whose live bounds differ, can never disagree on who evaluates a tile.  A tile its owner's bound cannot reach is
//     exactly cullable for every query, so the merged result is exact.
template <bool NEED_PERM, class Visit>
__device__ __forceinline__ unsigned long long coop_sweep(const TiledMap& mp, const lds_f32* lbox, bool use_lbox, int* s_list,
                                                         float* s_wbox, int* s_ctl, int* s_tick /*kMaxList tickets*/, int lane, int part, float (*sm)[64],
                                                         const float (&qx)[2], const float (&qy)[2], const float (&qz)[2],
                                                         const float (&reach)[2], const float (&bound2)[2], Visit&& visit,
                                                         bool prof, unsigned long long (&pc)[5], unsigned int (&pn)[3])
{
    unsigned long long n_staged = 0;
    // squared distance from each query to the box vs its live bound (see tiled_sweep::any_reach: exact, no margins)
    auto any_reach = [&](float m0, float m1, float m2, float m3, float m4, float m5) -> bool {
        const v2f s_qx = {qx[0], qx[1]}, s_qy = {qy[0], qy[1]}, s_qz = {qz[0], qz[1]};
        const v2f cx = {__builtin_amdgcn_fmed3f(qx[0], m0, m3), __builtin_amdgcn_fmed3f(qx[1], m0, m3)};
        const v2f cy = {__builtin_amdgcn_fmed3f(qy[0], m1, m4), __builtin_amdgcn_fmed3f(qy[1], m1, m4)};
        const v2f cz = {__builtin_amdgcn_fmed3f(qz[0], m2, m5), __builtin_amdgcn_fmed3f(qz[1], m2, m5)};
        const v2f ax = s_qx - cx, ay = s_qy - cy, az = s_qz - cz;   // (q - clamp(q, lo, hi): see tiled_sweep::any_reach)
        const v2f D = __builtin_elementwise_fma(az, az, __builtin_elementwise_fma(ay, ay, ax * ax));
        return __any(D.x <= bound2[0] || D.y <= bound2[1]);
    };

    // ---- wave 0: the wave box (union of the query boxes [q - r, q + r]) and the scan state of the upper levels ----
    const lds_f32* l_ubox = lbox;
    const lds_f32* l_sbox = lbox + 6 * mp.n_top;
    int ub = 0, sb = 0;
    unsigned long long ucand = 0, scand = 0;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f, c4 = 0.f, c5 = 0.f;
    bool c_valid = false;  // c0..c5 hold the super-tile boxes [sb, sb+64)
    Box w;
    if (part == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { w.lo[a] = INFINITY; w.hi[a] = -INFINITY; }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (reach[k] >= 0.f) {
                w.lo[0] = fminf(w.lo[0], qx[k] - reach[k]); w.hi[0] = fmaxf(w.hi[0], qx[k] + reach[k]);
                w.lo[1] = fminf(w.lo[1], qy[k] - reach[k]); w.hi[1] = fmaxf(w.hi[1], qy[k] + reach[k]);
                w.lo[2] = fminf(w.lo[2], qz[k] - reach[k]); w.hi[2] = fmaxf(w.hi[2], qz[k] + reach[k]);
            }
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) { w.lo[a] = wave_min_f(w.lo[a]); w.hi[a] = wave_max_f(w.hi[a]); }
        if (lane == 0) {
            s_wbox[0] = w.lo[0]; s_wbox[1] = w.lo[1]; s_wbox[2] = w.lo[2];
            s_wbox[3] = w.hi[0]; s_wbox[4] = w.hi[1]; s_wbox[5] = w.hi[2];
        }
    }
    auto load_super_boxes = [&]() {
        const int si = sb + lane;
        if (use_lbox) {
            c0 = l_sbox[si]; c1 = l_sbox[mp.n_super + si]; c2 = l_sbox[2 * mp.n_super + si];
            c3 = l_sbox[3 * mp.n_super + si]; c4 = l_sbox[4 * mp.n_super + si]; c5 = l_sbox[5 * mp.n_super + si];
        } else {
            c0 = mp.sbox[si]; c1 = mp.sbox[mp.n_super + si]; c2 = mp.sbox[2 * mp.n_super + si];
            c3 = mp.sbox[3 * mp.n_super + si]; c4 = mp.sbox[4 * mp.n_super + si]; c5 = mp.sbox[5 * mp.n_super + si];
        }
        c_valid = true;
    };
    auto fill_list = [&]() -> int {  // wave 0: resume the scan, collect up to kMaxList super-tiles
        int n_list = 0;
        while (n_list < kMaxList) {
            if (scand) {
                if (!c_valid) load_super_boxes();  // resumed after a full list
                const int sl = __builtin_ctzll(scand);
                scand &= scand - 1;
                if (prof) pn[0] += 1;
                if (any_reach(bcast_lane(c0, sl), bcast_lane(c1, sl), bcast_lane(c2, sl), bcast_lane(c3, sl),
                              bcast_lane(c4, sl), bcast_lane(c5, sl))) {
                    if (lane == 0) s_list[n_list] = sb + sl;
                    ++n_list;
                }
            } else if (ucand) {
                sb = (ub - 64 + __builtin_ctzll(ucand)) * 64;  // first super-tile of this top box (ub already advanced)
                ucand &= ucand - 1;
                load_super_boxes();
                scand = __ballot(c0 <= w.hi[0] && c1 <= w.hi[1] && c2 <= w.hi[2] && c3 >= w.lo[0] && c4 >= w.lo[1] &&
                                 c5 >= w.lo[2]);
            } else if (ub < mp.n_top) {
                const int ui = ub + lane;
                float u0 = INFINITY, u1 = INFINITY, u2 = INFINITY, u3 = -INFINITY, u4 = -INFINITY, u5 = -INFINITY;
                if (ui < mp.n_top) {
                    if (use_lbox) {
                        u0 = l_ubox[ui]; u1 = l_ubox[mp.n_top + ui]; u2 = l_ubox[2 * mp.n_top + ui];
                        u3 = l_ubox[3 * mp.n_top + ui]; u4 = l_ubox[4 * mp.n_top + ui]; u5 = l_ubox[5 * mp.n_top + ui];
                    } else {
                        u0 = mp.ubox[ui]; u1 = mp.ubox[mp.n_top + ui]; u2 = mp.ubox[2 * mp.n_top + ui];
                        u3 = mp.ubox[3 * mp.n_top + ui]; u4 = mp.ubox[4 * mp.n_top + ui]; u5 = mp.ubox[5 * mp.n_top + ui];
                    }
                }
                ucand = __ballot(u0 <= w.hi[0] && u1 <= w.hi[1] && u2 <= w.hi[2] && u3 >= w.lo[0] && u4 >= w.lo[1] &&
                                 u5 >= w.lo[2]);
                // A query group spread over many top boxes -- 64 consecutive sorted queries that straddle an empty stretch of the
                // curve: in a sparse scene two of 15 625 items span ~100 m -- must not descend into every one of them: against a
                // 10M-point map that is 76 dependent box loads and 4 900 per-query tests for ONE item, which then is as long as
                // the whole launch (0.25 ms; profiles/r04/sharded).  So such a group tests the top boxes per query first, as the
                // two levels below always do.  Exact for the same reason (a top box contains its super-tiles' boxes).  Ordinary
                // groups (<= 4 candidate top boxes) do not pay for it.
                if (__builtin_popcountll(ucand) > 4) {
                    unsigned long long uc = ucand, keep = 0ull;
                    while (uc) {
                        const int t = __builtin_ctzll(uc);
                        uc &= uc - 1;
                        if (any_reach(bcast_lane(u0, t), bcast_lane(u1, t), bcast_lane(u2, t), bcast_lane(u3, t), bcast_lane(u4, t),
                                      bcast_lane(u5, t)))
                            keep |= 1ull << t;
                    }
                    ucand = keep;
                }
                ub += 64;
            } else {
                break;
            }
        }
        return n_list;
    };

    // ---- all waves: staging pipeline (two tiles = 64 points per pass, the next pair's loads in flight) ----
    int pend_a = -1, pend_b = -1;  // tile ids whose points sit in the registers below
    float px = 0.f, py = 0.f, pz = 0.f;
    int po = 0;
    auto load_pair = [&](int ta, int tb) {
        const int tt = lane < 32 ? ta : tb;
        px = py = pz = 1.0e18f;  // padding points: d2 ~ 3e36, never a neighbour
        po = 0x7fffffff;
        if (tt >= 0) {
            const int j = tt * kTileG + (lane & 31);
            px = mp.sx[j]; py = mp.sy[j]; pz = mp.sz[j];
            if (NEED_PERM) po = mp.perm[j];
        }
    };
    auto compute_pending = [&](int next_a, int next_b) {
        const int ca = pend_a, cb = pend_b;
        const unsigned long long tp0 = prof ? __builtin_amdgcn_s_memtime() : 0ull;
        sm[0][lane] = px; sm[1][lane] = py; sm[2][lane] = pz;
        if (NEED_PERM) sm[3][lane] = __int_as_float(po);
        pend_a = next_a; pend_b = next_b;
        if (pend_a >= 0) load_pair(pend_a, pend_b);  // next pass's loads fly while this pass computes
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int nm = cb >= 0 ? 64 : 32;
        n_staged += nm;
        const unsigned long long tp1 = prof ? __builtin_amdgcn_s_memtime() : 0ull;
        visit(nm, ca * kTileG, (cb >= 0 ? cb : ca) * kTileG);
        __builtin_amdgcn_wave_barrier();  // the staging area is rewritten by the next pass
        if (prof) { const unsigned long long tp2 = __builtin_amdgcn_s_memtime(); pc[0] += tp1 - tp0; pc[1] += tp2 - tp1; }
    };

    bool have_w = part == 0;
    (void)part;
    for (;;) {
        const unsigned long long tf0 = prof ? __builtin_amdgcn_s_memtime() : 0ull;
        if (part == 0) {
            const int n = fill_list();
            if (lane == 0) s_ctl[0] = n;
            s_tick[lane] = 0;   // (kMaxList = 64 = one per lane) every listed super-tile's candidate tiles are dealt from ticket 0
        }
        __syncthreads();
        if (prof) pc[4] += __builtin_amdgcn_s_memtime() - tf0;   // wave 0: the walk over the upper box levels; the others: waiting for it
        const int n_list = s_ctl[0];  // (workgroup-uniform)
        if (n_list == 0) break;
        if (!have_w) {
            w.lo[0] = s_wbox[0]; w.lo[1] = s_wbox[1]; w.lo[2] = s_wbox[2];
            w.hi[0] = s_wbox[3]; w.hi[1] = s_wbox[4]; w.hi[2] = s_wbox[5];
            have_w = true;
        }
        // the listed super-tiles: tile boxes of entry e+1 in flight while entry e's tiles are processed
        int S = __builtin_amdgcn_readfirstlane(s_list[0]);
        int ti = S * kSuper + lane;
        float n0 = mp.tbox[ti], n1 = mp.tbox[mp.n_tiles_p + ti], n2 = mp.tbox[2 * mp.n_tiles_p + ti],
              n3 = mp.tbox[3 * mp.n_tiles_p + ti], n4 = mp.tbox[4 * mp.n_tiles_p + ti], n5 = mp.tbox[5 * mp.n_tiles_p + ti];
        for (int e = 0; e < n_list; ++e) {
            const unsigned long long tb0 = prof ? __builtin_amdgcn_s_memtime() : 0ull;
            const float b0 = n0, b1 = n1, b2 = n2, b3 = n3, b4 = n4, b5 = n5;
            const int Sc = S;
            if (e + 1 < n_list) {
                S = __builtin_amdgcn_readfirstlane(s_list[e + 1]);
                ti = S * kSuper + lane;
                n0 = mp.tbox[ti]; n1 = mp.tbox[mp.n_tiles_p + ti]; n2 = mp.tbox[2 * mp.n_tiles_p + ti];
                n3 = mp.tbox[3 * mp.n_tiles_p + ti]; n4 = mp.tbox[4 * mp.n_tiles_p + ti]; n5 = mp.tbox[5 * mp.n_tiles_p + ti];
            }
            // The candidate tiles of this super-tile (the same set in all four waves: same boxes, same wave box) are dealt by
            // TICKET: a wave takes the next one nobody has taken (one LDS atomic), whichever it is.  A static deal -- tile t to
            // wave (t + S) % 4, rounds 1-2 -- left the waves up to 3x apart (own tile tests 4 / 7 / 31 per wave at 100k x 100k)
            // and the item as long as its slowest wave.  Still exact: every tile has ONE taker, and a tile its taker's live
            // bounds cannot reach holds nothing that beats the final ones either.
            const unsigned long long cand = __ballot(b0 <= w.hi[0] && b1 <= w.hi[1] && b2 <= w.hi[2] && b3 >= w.lo[0] &&
                                                     b4 >= w.lo[1] && b5 >= w.lo[2]);
            const int n_cand = __popcll(cand);
            const bool my_bit = (cand >> lane) & 1ull;
            const int my_rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(cand >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)cand, 0u));
            const unsigned long long tb1 = prof ? __builtin_amdgcn_s_memtime() : 0ull;
            bool dry = n_cand == 0;
            auto next_tile = [&]() -> int {  // the next candidate nobody has taken that some query still reaches (live bound), or -1
                while (!dry) {
                    int tk = 0;
                    if (lane == 0) tk = atomicAdd(&s_tick[e], 1);
                    tk = __builtin_amdgcn_readfirstlane(tk);
                    if (tk >= n_cand) { dry = true; break; }
                    const int t = __builtin_ctzll(__ballot(my_bit && my_rank == tk));
                    if (prof) pn[2] += 1;
                    if (any_reach(bcast_lane(b0, t), bcast_lane(b1, t), bcast_lane(b2, t), bcast_lane(b3, t),
                                  bcast_lane(b4, t), bcast_lane(b5, t)))
                        return Sc * kSuper + t;
                }
                return -1;
            };
            for (;;) {
                const int t0 = next_tile();
                if (t0 < 0) break;
                const int t1 = next_tile();
                if (pend_a < 0) {  // nothing in flight yet: just issue this pair's loads
                    pend_a = t0; pend_b = t1;
                    load_pair(t0, t1);
                } else {
                    compute_pending(t0, t1);
                }
            }
            if (prof) { const unsigned long long tb2 = __builtin_amdgcn_s_memtime(); pc[2] += tb1 - tb0; pc[3] += tb2 - tb1; pn[1] += 1; }
        }
        if (n_list < kMaxList) break;  // the scan of the upper levels has ended (no barrier: nothing shared is rewritten any more)
        __syncthreads();  // the shared list and its tickets are rewritten from here on
    }
    if (pend_a >= 0) compute_pending(-1, -1);
    return n_staged;
}

// One pass over the workgroup's item with the fast (EXACT = false) or the exact-key visitor.  All four waves call it
// with the same queries; on return waves 0/1 hold the merged result for the queries k = wave of every lane.
// Returns (workgroup-uniform) whether some query met an exact distance tie the fast visitor cannot resolve.
template <bool EXACT>
__device__ __forceinline__ bool coop_item_pass(const NnProblem& pb, const TiledMap& mp, const lds_f32* lbox, bool lds_boxes, int* s_list,
                                               float* s_wbox, int* s_ctl, int* s_tick,
                                               float (*sm)[64], unsigned int (*s_mg)[6][64], unsigned int* s_flag, int lane,
                                               int wave, const float (&qx)[2], const float (&qy)[2], const float (&qz)[2],
                                               const int (&qi)[2], const int (&js)[2],
                                               const float (&sd)[2], float thr2, CoopResult& res, bool prof,
                                               unsigned long long (&dbg)[8])
{
    const int N = pb.N;
    float reach[2], best[2];
    unsigned long long key[2];  // EXACT: packed (d2, original index)
    int bpos[2];                // EXACT: sorted position of the best point; fast: of its kGroup-point group
    int tie[2] = {0, 0};
    int jo[2] = {0, 0};  // EXACT: the seeds' original indices (one more trip on this rare path)
    if constexpr (EXACT) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (js[k] >= 0) jo[k] = pb.idx_s[qi[k] < N ? qi[k] : N - 1];
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        key[k] = ((unsigned long long)__float_as_uint(thr2) << 32);  // (gate^2, index 0): "no neighbour" sentinel
        best[k] = thr2;
        bpos[k] = -1;
        if (js[k] >= 0 && sd[k] < thr2) {  // warm start: last launch's neighbour is an exact candidate
            best[k] = sd[k];
            bpos[k] = EXACT ? js[k] : (js[k] & ~(kGroup - 1));
            if (EXACT) key[k] = ((unsigned long long)__float_as_uint(sd[k]) << 32) | (unsigned int)jo[k];
        }
        reach[k] = reach_of(best[k], qx[k], qy[k], qz[k]);
        if (qi[k] >= N) {  // padding lane: reaches nothing, is never written
            reach[k] = -1.0f;
            best[k] = -1.0f;
            bpos[k] = -1;
            key[k] = ((unsigned long long)__float_as_uint(thr2) << 32);  // (a padding lane loaded the last query's seed: the exact epilogue reads the point at bpos for any key below the gate)
        }
    }
    unsigned long long pc[5] = {0ull, 0ull, 0ull, 0ull, 0ull};
    unsigned int pn[3] = {0u, 0u, 0u};
    const unsigned long long n_staged = coop_sweep<EXACT>(
        mp, lbox, lds_boxes, s_list, s_wbox, s_ctl, s_tick, lane, wave, sm, qx, qy, qz, reach, best,
        [&](int nm, int jb0, int jb1) {
            if constexpr (EXACT) nn_visit_exact<2>(sm, nm, jb0, jb1, qx, qy, qz, key, best, bpos);
            else nn_visit_fast<2>(sm, nm, jb0, jb1, qx, qy, qz, best, bpos, tie);
        },
        prof, pc, pn);
    if (prof) {
        dbg[0] = __builtin_amdgcn_s_memtime();
        dbg[1] = n_staged | ((unsigned long long)pn[1] << 16) | ((unsigned long long)pn[2] << 32) | ((unsigned long long)pn[0] << 48);
        dbg[4] = pc[0]; dbg[5] = pc[1] | (pc[4] << 32); dbg[6] = pc[2]; dbg[7] = pc[3];  // staging, distance passes | list fill + barrier, tile-box wait, tile tests (+ the passes inside)
    }

    // ---- merge the four partial results per query through LDS ----
    unsigned int(*mg)[64] = s_mg[wave];
    if constexpr (EXACT) {
        mg[0][lane] = (unsigned int)(key[0] & 0xffffffffu); mg[1][lane] = (unsigned int)(key[0] >> 32);
        mg[2][lane] = (unsigned int)(key[1] & 0xffffffffu); mg[3][lane] = (unsigned int)(key[1] >> 32);
        mg[4][lane] = (unsigned int)bpos[0]; mg[5][lane] = (unsigned int)bpos[1];
    } else {
        mg[0][lane] = __float_as_uint(best[0]); mg[1][lane] = __float_as_uint(best[1]);
        mg[2][lane] = (unsigned int)bpos[0]; mg[3][lane] = (unsigned int)bpos[1];
        mg[4][lane] = (unsigned int)tie[0]; mg[5][lane] = (unsigned int)tie[1];
    }
    if (lane == 0) atomicAdd(&s_flag[2], (unsigned int)n_staged);  // LDS atomic: the workgroup's staged points
    __syncthreads();
    if (prof) dbg[2] = __builtin_amdgcn_s_memtime();

    bool any_tie = false;
    res.rpos = -1; res.roi = -1; res.rd = thr2; res.gx = res.gy = res.gz = 0.f;
    if (wave < 2) {  // wave w finishes the queries k = w of every lane (64 consecutive queries, coalesced stores)
        const int k = wave;
        const float fqx = k ? qx[1] : qx[0], fqy = k ? qy[1] : qy[0], fqz = k ? qz[1] : qz[0];
        const int fqi = k ? qi[1] : qi[0];
        if constexpr (EXACT) {
            unsigned long long mk = ((unsigned long long)s_mg[0][2 * k + 1][lane] << 32) | s_mg[0][2 * k][lane];
            int mp_ = (int)s_mg[0][4 + k][lane];
#pragma unroll
            for (int w = 1; w < kCoopParts; ++w) {
                const unsigned long long k2 = ((unsigned long long)s_mg[w][2 * k + 1][lane] << 32) | s_mg[w][2 * k][lane];
                const int p2 = (int)s_mg[w][4 + k][lane];
                const bool better = k2 < mk;  // equal keys = the same point (same d2, same original index)
                mk = better ? k2 : mk;
                mp_ = better ? p2 : mp_;
            }
            const float d = __uint_as_float((unsigned int)(mk >> 32));
            if (d < thr2 && fqi < N) {
                res.rd = d; res.rpos = mp_; res.roi = (int)(unsigned int)(mk & 0xffffffffu);
                res.gx = mp.sx[mp_]; res.gy = mp.sy[mp_]; res.gz = mp.sz[mp_];  // (rare path: one more trip)
            }
        } else {
            float b = __uint_as_float(s_mg[0][k][lane]);
            int p = (int)s_mg[0][2 + k][lane];
            int t = (int)s_mg[0][4 + k][lane];
#pragma unroll
            for (int w = 1; w < kCoopParts; ++w) {
                const float b2 = __uint_as_float(s_mg[w][k][lane]);
                const int p2 = (int)s_mg[w][2 + k][lane], t2 = (int)s_mg[w][4 + k][lane];
                const bool lt = b2 < b, eq = b2 == b;
                // t = number of groups at the best (nn_visit_fast): every tile has ONE owner, so the waves' counts add
                // up; two groups at an equal minimum are a tie the exact visitor must resolve
                t = lt ? t2 : (eq ? t + t2 : t);
                p = lt ? p2 : p;
                b = lt ? b2 : b;
            }
            // resolve inside the winning group: the point(s) with d2 == best, lowest original index first
            const int bp = p >= 0 ? p : 0;
            float4 RX[kGroup / 4], RY[kGroup / 4], RZ[kGroup / 4];
            int4 RP[kGroup / 4];
#pragma unroll
            for (int c = 0; c < kGroup / 4; ++c) {
                RX[c] = *reinterpret_cast<const float4*>(mp.sx + bp + 4 * c);
                RY[c] = *reinterpret_cast<const float4*>(mp.sy + bp + 4 * c);
                RZ[c] = *reinterpret_cast<const float4*>(mp.sz + bp + 4 * c);
                RP[c] = *reinterpret_cast<const int4*>(mp.perm + bp + 4 * c);
            }
            unsigned int bo = 0xffffffffu;
            int pos = -1;
            float wx = 0.f, wy = 0.f, wz = 0.f;
#pragma unroll
            for (int c = 0; c < kGroup / 4; ++c) {
                const float xs[4] = {RX[c].x, RX[c].y, RX[c].z, RX[c].w};
                const float ys[4] = {RY[c].x, RY[c].y, RY[c].z, RY[c].w};
                const float zs[4] = {RZ[c].x, RZ[c].y, RZ[c].z, RZ[c].w};
                const int ps[4] = {RP[c].x, RP[c].y, RP[c].z, RP[c].w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float du = dist2(fqx, fqy, fqz, xs[u], ys[u], zs[u]);
                    const bool take = du == b && (unsigned int)ps[u] < bo;
                    bo = take ? (unsigned int)ps[u] : bo;
                    pos = take ? p + 4 * c + u : pos;
                    wx = take ? xs[u] : wx; wy = take ? ys[u] : wy; wz = take ? zs[u] : wz;
                }
            }
            if (p >= 0 && fqi < N) {
                res.rd = b; res.rpos = pos; res.roi = (int)bo;
                res.gx = wx; res.gy = wy; res.gz = wz;
                if (pos < 0) { t = 2; res.roi = -1; }  // cannot happen (same arithmetic); be safe: exact pass
            }
            any_tie = fqi < N && t >= 2;
        }
        if (!EXACT) {
            const bool wt = __any(any_tie);
            if (lane == 0) s_flag[k] = wt ? 1u : 0u;
        }
    }
    __syncthreads();  // the tie flags of both finishing waves are in; s_mg may be rewritten
    const bool redo = !EXACT && (s_flag[0] | s_flag[1]) != 0u;
    if (prof) dbg[3] = __builtin_amdgcn_s_memtime();
    return redo;
}

template <int KMAX, bool DIAG = false /*per-wave phase records (MOLA_ICP_DEBUG_STATS=2); compiled out of the launch path: see k_nn_tiled*/>
__global__ __launch_bounds__(256, 4) void k_nn_coop(const NnBatch<KMAX> batch, int lds_boxes,
                                                    unsigned long long* __restrict__ wave_times /*diagnostics, usually null*/)
{
    __shared__ __attribute__((aligned(16))) float s_m[4][4][64];  // per wave: x, y, z, original index of 64 staged points
    __shared__ int s_list[kMaxList];                  // super-tiles some query reaches (written by wave 0, streamed by all)
    __shared__ float s_wbox[6];                       // the wave box (wave 0's reduction)
    __shared__ int s_ctl[2];                          // [0] entries in s_list
    __shared__ int s_tick[kMaxList];                  // per listed super-tile: the next candidate tile to hand out (coop_sweep)
    __shared__ unsigned int s_mg[kCoopParts][6][64];  // per wave: its partial result for the 128 queries
    __shared__ unsigned int s_flag[4];                // [0..1] tie seen by finishing wave 0/1, [2] staged points
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];  // the upper box levels, if they fit
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float(*sm)[64] = s_m[wave];
    const NnProblem& pb = batch.p[KMAX == 1 ? 0 : blockIdx.y];
    const int N = pb.N;
    const int item = lds_boxes ? (int)blockIdx.x : xcd_item((int)blockIdx.x, (N + kQPW - 1) / kQPW);   // (see k_knn_coop)
    if (item * kQPW >= N) return;  // nothing for this workgroup (uniform: before any barrier)
    const float thr2 = pb.thr2;
    const int use_seed = pb.use_seed;
    const TiledMap mp = pb.mp;
    const PoseF P = pb.P;
    const lds_f32* lbox = (const lds_f32*)s_dyn;
    const bool prof = DIAG && wave_times != nullptr;
    const unsigned long long t_wave0 = prof ? wall_clock64() : 0ull;  // 100 MHz, the same on every XCD
    const unsigned long long c0 = prof ? __builtin_amdgcn_s_memtime() : 0ull;

    // round trip A: the lane's two queries, their seeds with coordinates -- and the box levels, issued behind them
    int qi[2], js[2];
    float lx[2], ly[2], lz[2], gsx[2], gsy[2], gsz[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        qi[k] = item * kQPW + k * 64 + lane;
        if (qi[k] >= N) qi[k] = N;  // padding lane
        const int ic = qi[k] < N ? qi[k] : N - 1;
        lx[k] = pb.slx[ic]; ly[k] = pb.sly[ic]; lz[k] = pb.slz[ic];
        js[k] = -1; gsx[k] = gsy[k] = gsz[k] = 0.f;
        if (use_seed) {
            js[k] = pb.pos_s[ic];
            gsx[k] = pb.gsx[ic]; gsy[k] = pb.gsy[ic]; gsz[k] = pb.gsz[ic];
        }
    }
    if (threadIdx.x < 4) s_flag[threadIdx.x] = 0u;
    if (lds_boxes) load_boxes_to_lds(mp, (lds_f32*)s_dyn);  // (ends with a barrier)
    else __syncthreads();
    const unsigned long long c1 = prof ? __builtin_amdgcn_s_memtime() : 0ull;
    float qx[2], qy[2], qz[2], sd[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        xform(P, lx[k], ly[k], lz[k], qx[k], qy[k], qz[k]);
        sd[k] = dist2(qx[k], qy[k], qz[k], gsx[k], gsy[k], gsz[k]);
        if (qi[k] >= N) qx[k] = qy[k] = qz[k] = 1.0e18f;  // padding lane
    }
    const unsigned long long c2 = prof ? __builtin_amdgcn_s_memtime() : 0ull;

    CoopResult res;
    unsigned long long dbg[8] = {};
    if (coop_item_pass<false>(pb, mp, lbox, lds_boxes != 0, s_list, s_wbox, s_ctl, s_tick, sm, s_mg, s_flag, lane, wave, qx, qy, qz, qi, js, sd, thr2,
                              res, prof, dbg)) {
        // exact distance ties in this item: once more with the per-pair (d2, original index) key
        unsigned long long dbg2[8];
        (void)coop_item_pass<true>(pb, mp, lbox, lds_boxes != 0, s_list, s_wbox, s_ctl, s_tick, sm, s_mg, s_flag, lane, wave, qx, qy, qz, qi, js, sd,
                                   thr2, res, false, dbg2);
    }

    // ---- the pairing (sorted query order, coalesced) and the item's row of unit-weight sums ----
    // Rows of 64 queries, formed by the finishing wave itself on the fp64 matrix cores (item_row_mfma, kernels_tiled.hpp): the
    // SAME row, bit for bit, that k_nn_tiled / k_nn_tiled_batch write for these 64 queries -- a problem's accumulators, and its
    // pose, do not depend on which matcher served it, at any size.  No workgroup barrier: waves 2 / 3 are done.
    if (wave < 2 && (2 * item + wave) * 64 < N) {   // (wave-uniform; a half that is all padding has no row)
        const int k = wave;
        const int fqi = k ? qi[1] : qi[0];
        float al0 = 0.f, al1 = 0.f, al2 = 0.f;
        bool paired = false;
        if (fqi < N) {
            pb.pos_s[fqi] = res.rpos;
            pb.idx_s[fqi] = res.rpos >= 0 ? res.roi : -1;
            pb.d2_s[fqi] = res.rd;
            pb.gsx[fqi] = res.gx; pb.gsy[fqi] = res.gy; pb.gsz[fqi] = res.gz;
            paired = res.rpos >= 0;
            al0 = pb.slx[fqi]; al1 = pb.sly[fqi]; al2 = pb.slz[fqi];  // (re-read: cheaper than six registers held through the sweep)
        }
        item_row_mfma(&sm[0][0], lane, paired, al0, al1, al2, res.gx, res.gy, res.gz, res.rd, pb.rows + (size_t)(2 * item + wave) * kNAcc);
    }

    if (prof && lane == 0 && blockIdx.y == 0 && blockIdx.x * 4 + wave < 8192) {
        // [start, end (wall clock)], then shader cycles: setup (round trip A + boxes -> LDS), sweep, wait for the other
        // waves, finish (group re-read, row sum); packed: staged points | supers entered << 16 | tile tests << 32 | super tests << 48
        unsigned long long* w = wave_times + 8 * (size_t)(blockIdx.x * 4 + wave);
        // (two 32-bit cycle counts per slot from [2] on)
        auto pk = [](unsigned long long lo, unsigned long long hi) { return (lo & 0xffffffffull) | (hi << 32); };
        w[0] = t_wave0; w[1] = wall_clock64(); w[2] = pk(c1 - c0, c2 - c1); w[3] = pk(dbg[0] - c2, dbg[2] - dbg[0]);
        w[4] = pk(__builtin_amdgcn_s_memtime() - dbg[2], dbg[6]); w[5] = pk(dbg[7], dbg[4]); w[6] = pk(dbg[5] & 0xffffffffull, dbg[5] >> 32); w[7] = dbg[1];
    }
    if (threadIdx.x == 0 && s_flag[2])  // executed work in units of 64 (query, point) pairs; slotted: no same-address burst
        atomicAdd(pb.staged + (size_t)(blockIdx.x & (kStatSlots - 1)) * kStatStride, (unsigned long long)s_flag[2] * 2ull);
}

// ---- many problems, throughput mode ----------------------------------------------------------------------------
// With a dozen 100k-point problems in one launch there are ~10^4 items: latency no longer matters, issue slots do, and
// the cooperative kernel spends them four times over on every item's prologue and box scan.  k_nn_tiled_batch is the
// persistent one-wave-per-item matcher of kernels_tiled.hpp over the items of ALL problems (entry -> problem by the
// prefix sums `item_base`): same sweep, same visitors, same tie rule; a wave that meets an exact distance tie redoes
// its item with the exact-key visitor at once.  The upper box levels come from the LDS copy only if every problem
// shares ONE map (`shared_map`: the loop-closure Monte-Carlo), else from global memory.  Every item's row of unit-weight
// sums is formed in the epilogue exactly as k_nn_coop and k_nn_tiled form it (item_row_mfma: the same bits for the same 64
// queries), so a problem's accumulators -- and its pose -- do not depend on which of the matchers served it.
template <int KMAX> struct NnBatchItems { int base[KMAX + 1]; };  // base[k] = first entry of problem k; base[n] = total

template <int QL, bool EXACT, class AfterSweep>
__device__ __forceinline__ bool tiled_batch_item(const NnProblem& pb, const TiledMap& mp, const lds_f32* lbox, bool use_lbox, int* slist,
                                                 float (*sm)[64], int lane, int item, unsigned long long& n_staged_out,
                                                 AfterSweep&& after_sweep)
{
    const int N = pb.N;
    const float thr2 = pb.thr2;
    const PoseF P = pb.P;
    int qi[QL], js[QL];
    float qx[QL], qy[QL], qz[QL], reach[QL], best[QL];
    unsigned long long key[QL];
    int bpos[QL], tie[QL] = {};
    {
        float lx[QL], ly[QL], lz[QL], gsx[QL], gsy[QL], gsz[QL];
        unsigned int gso[QL] = {};
#pragma unroll
        for (int k = 0; k < QL; ++k) {
            qi[k] = item * (64 * QL) + k * 64 + lane;
            if (qi[k] >= N) qi[k] = N;  // padding lane
            const int ic = qi[k] < N ? qi[k] : N - 1;
            lx[k] = pb.slx[ic]; ly[k] = pb.sly[ic]; lz[k] = pb.slz[ic];
            js[k] = -1; gsx[k] = gsy[k] = gsz[k] = 0.f;
            if (pb.use_seed || EXACT) {  // (the exact redo is seeded by the fast pass's result)
                js[k] = pb.pos_s[ic];
                gsx[k] = pb.gsx[ic]; gsy[k] = pb.gsy[ic]; gsz[k] = pb.gsz[ic];
                if (EXACT) gso[k] = (unsigned int)pb.idx_s[ic];
            }
        }
#pragma unroll
        for (int k = 0; k < QL; ++k) {
            xform(P, lx[k], ly[k], lz[k], qx[k], qy[k], qz[k]);
            key[k] = ((unsigned long long)__float_as_uint(thr2) << 32);
            best[k] = thr2;
            bpos[k] = -1;
            const float d = dist2(qx[k], qy[k], qz[k], gsx[k], gsy[k], gsz[k]);
            if (js[k] >= 0 && d < thr2) {
                best[k] = d;
                bpos[k] = EXACT ? js[k] : (js[k] & ~(kGroup - 1));
                if (EXACT) key[k] = ((unsigned long long)__float_as_uint(d) << 32) | gso[k];
            }
            reach[k] = reach_of(best[k], qx[k], qy[k], qz[k]);
            if (qi[k] >= N) {
                qx[k] = qy[k] = qz[k] = 1.0e18f;
                reach[k] = -1.0f;
                best[k] = -1.0f;
                bpos[k] = -1;
                key[k] = ((unsigned long long)__float_as_uint(thr2) << 32);  // (a padding lane loaded the last query's seed: the exact epilogue reads the point at bpos for any key below the gate)
            }
        }
    }
    unsigned long long pa = 0ull, pb2 = 0ull, pf = 0ull, pg = 0ull;
    unsigned int pc = 0u, pd = 0u, pe = 0u;
    n_staged_out += tiled_sweep<QL, EXACT>(mp, lbox, use_lbox, slist, lane, sm, qx, qy, qz, reach, best, [&](int nm, int jb0, int jb1) {
        if constexpr (EXACT) nn_visit_exact<QL>(sm, nm, jb0, jb1, qx, qy, qz, key, best, bpos);
        else nn_visit_fast<QL>(sm, nm, jb0, jb1, qx, qy, qz, best, bpos, tie);
    }, false, pa, pb2, pc, pd, pe, pf, pg);
    after_sweep();

    bool any_tie = false;
    int f_pos = -1;                                    // (QL = 1: the lane's result, for the item's row)
    float f_d = 0.f, f_gx = 0.f, f_gy = 0.f, f_gz = 0.f;
#pragma unroll
    for (int k = 0; k < QL; ++k) {
        int rpos = -1, roi = -1;
        float rd = thr2, wx = 0.f, wy = 0.f, wz = 0.f;
        if constexpr (EXACT) {
            const float d = __uint_as_float((unsigned int)(key[k] >> 32));
            if (d < thr2) {
                rd = d; rpos = bpos[k]; roi = (int)(unsigned int)(key[k] & 0xffffffffu);
                wx = mp.sx[rpos]; wy = mp.sy[rpos]; wz = mp.sz[rpos];
            }
        } else {
            const int bp = bpos[k] >= 0 ? bpos[k] : 0;
            float4 RX[kGroup / 4], RY[kGroup / 4], RZ[kGroup / 4];
            int4 RP[kGroup / 4];
#pragma unroll
            for (int c = 0; c < kGroup / 4; ++c) {
                RX[c] = *reinterpret_cast<const float4*>(mp.sx + bp + 4 * c);
                RY[c] = *reinterpret_cast<const float4*>(mp.sy + bp + 4 * c);
                RZ[c] = *reinterpret_cast<const float4*>(mp.sz + bp + 4 * c);
                RP[c] = *reinterpret_cast<const int4*>(mp.perm + bp + 4 * c);
            }
            unsigned int bo = 0xffffffffu;
            int pos = -1;
#pragma unroll
            for (int c = 0; c < kGroup / 4; ++c) {
                const float xs[4] = {RX[c].x, RX[c].y, RX[c].z, RX[c].w};
                const float ys[4] = {RY[c].x, RY[c].y, RY[c].z, RY[c].w};
                const float zs[4] = {RZ[c].x, RZ[c].y, RZ[c].z, RZ[c].w};
                const int ps[4] = {RP[c].x, RP[c].y, RP[c].z, RP[c].w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float du = dist2(qx[k], qy[k], qz[k], xs[u], ys[u], zs[u]);
                    const bool take = du == best[k] && (unsigned int)ps[u] < bo;
                    bo = take ? (unsigned int)ps[u] : bo;
                    pos = take ? bpos[k] + 4 * c + u : pos;
                    const unsigned int tm = take ? 0xffffffffu : 0u;
                    wx = __uint_as_float((__float_as_uint(xs[u]) & tm) | (__float_as_uint(wx) & ~tm));
                    wy = __uint_as_float((__float_as_uint(ys[u]) & tm) | (__float_as_uint(wy) & ~tm));
                    wz = __uint_as_float((__float_as_uint(zs[u]) & tm) | (__float_as_uint(wz) & ~tm));
                }
            }
            if (bpos[k] >= 0) {
                rd = best[k]; rpos = pos; roi = (int)bo;
                if (pos < 0) tie[k] = 2;  // cannot happen (same arithmetic); be safe: exact pass
            }
        }
        if (qi[k] < N) {
            pb.pos_s[qi[k]] = rpos;
            pb.idx_s[qi[k]] = rpos >= 0 ? roi : -1;
            pb.d2_s[qi[k]] = rd;
            pb.gsx[qi[k]] = wx; pb.gsy[qi[k]] = wy; pb.gsz[qi[k]] = wz;
            any_tie |= tie[k] >= 2;
        }
        if constexpr (QL == 1) { f_pos = rpos; f_d = rd; f_gx = wx; f_gy = wy; f_gz = wz; }
    }
    const bool redo = !EXACT && __any(any_tie);
    if constexpr (QL == 1) {
        if (!redo) {   // (wave-uniform; a tied item's row is written by its exact redo) -- the item's row of unit-weight sums
            const int ic = qi[0] < N ? qi[0] : N - 1;
            item_row_mfma(&sm[0][0], lane, qi[0] < N && f_pos >= 0, pb.slx[ic], pb.sly[ic], pb.slz[ic], f_gx, f_gy, f_gz, f_d,
                          pb.rows + (size_t)item * kNAcc);
        }
    }
    return redo;
}

template <int KMAX, int QL /*queries per lane: items of 64 * QL queries (1: as k_nn_tiled's default, four workgroups per CU)*/>
__global__ __launch_bounds__(256, 3) void k_nn_tiled_batch(const NnBatch<KMAX> batch, const NnBatchItems<KMAX> items, int n_problems,
                                                           int shared_map_lds, unsigned int* __restrict__ queue,
                                                           int early_pop /*tuning knob: reserve the next item at the START of this one*/)
{
    __shared__ __attribute__((aligned(16))) float s_m[4][4][64];
    __shared__ int s_list[4][kMaxList];
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float(*sm)[64] = s_m[wave];
    int* slist = s_list[wave];
    const lds_f32* lbox = (const lds_f32*)s_dyn;
    if (shared_map_lds) load_boxes_to_lds(batch.p[0].mp, (lds_f32*)s_dyn);
    const int n_items = items.base[n_problems];
    WaveQueue wq(queue, lane, n_items);
    unsigned long long wave_staged = 0ull;
    int raw = __builtin_amdgcn_readfirstlane(wq.first());
    while (raw < n_items) {
        int k = 0;
        while (k + 1 < n_problems && raw >= items.base[k + 1]) ++k;  // (wave-uniform: scalar loop over <= KMAX entries)
        const NnProblem& pb = batch.p[k];
        const int item = raw - items.base[k];
        // the next item is reserved LATE, behind the sweep (in flight during the epilogue): reserved at the start, an item was
        // bound to a wave one whole item ahead of its execution and the launch's drain was two items long (see k_nn_tiled)
        int next_raw_v = 0;
        if (early_pop) next_raw_v = wq.pop();
        const TiledMap mp = pb.mp;
        if (tiled_batch_item<QL, false>(pb, mp, lbox, shared_map_lds != 0, slist, sm, lane, item, wave_staged,
                                    [&]() { if (!early_pop) next_raw_v = wq.pop(); })) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // the redo reads this wave's own stores of a moment ago
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            (void)tiled_batch_item<QL, true>(pb, mp, lbox, shared_map_lds != 0, slist, sm, lane, item, wave_staged, []() {});
        }
        raw = __builtin_amdgcn_readfirstlane(wq.settle(__builtin_amdgcn_readfirstlane(next_raw_v)));
    }
    if (lane == 0 && wave_staged)
        atomicAdd(batch.p[0].staged + (size_t)(blockIdx.x & (kStatSlots - 1)) * kStatStride, wave_staged * (unsigned long long)QL);
}

}  // namespace mola_icp_amd
