// kernels_tiled.hpp -- the tiled sweep (box levels, LDS staging), persistent-wave work queue, k_nn_tiled
// Device code of the ICP core for gfx950; included by hip_backend.hip only (one translation unit: the kernels are
// launched from there).  Numeric contract and data layout: hip_backend.hip / DESIGN.md.
#pragma once
#include <type_traits>

#include "kernels_common.hpp"

namespace mola_icp_amd {

// ---- tiled matcher: exact brute force over the map tiles a wave's queries can reach ---------
// Both clouds are put in Hilbert order once (map: per map; local cloud: per cloud; map_sort.hip).  The map is
// cut into TILES of 32 consecutive points with an axis-aligned box; 64 tiles form a super-tile, 64 super-tiles a
// top box.  A wave owns 128 (or 64) consecutive -- hence spatially compact -- queries, two (one) per lane.  Per item:
//   1. transform the queries, warm-start each best from the previous iteration's neighbour; the wave box is the
//      union of the boxes [q - r, q + r], r = sqrt(best) rounded up;
//   2. cull: lanes test 64 boxes of a level at a time against the wave box (ballot); a hit is re-tested against
//      every query of the wave by its squared box distance vs the query's live best (exact, see tiled_sweep);
//   3. every surviving tile is staged in LDS (two tiles = 64 points per pass) and all queries of the wave are
//      evaluated against all its points with the exact contract; the winner is resolved with the lexicographic
//      (d2, lowest ORIGINAL index) rule -- bit-identical to the dense kernels.
// No tree, no per-query traversal, no data-dependent recursion: flat box scans and dense query x point tiles.
constexpr int kTileG = 32;     // map points per tile
constexpr int kSuper = 64;     // tiles per super-tile
constexpr int kQPW = 128;      // queries per wave
constexpr int kGroup = 8;      // fast sweep: points per bookkeeping group (16: epilogue too dear, 4: bookkeeping)

typedef float v2f __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) float lds_f32;  // explicit LDS pointers: ds_read, never flat_load

// dist2 for the two queries of a lane at once (v_pk_add/mul/fma_f32): each half is the same IEEE sequence as dist2
__device__ __forceinline__ v2f dist2_pk(v2f qx, v2f qy, v2f qz, float mx, float my, float mz)
{
    const v2f dx = qx - mx, dy = qy - my, dz = qz - mz;
    return __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
}

// ... and for ONE query against two map points at once (same sequence per half)
__device__ __forceinline__ v2f dist2_pk2(float qx, float qy, float qz, v2f mx, v2f my, v2f mz)
{
    const v2f dx = qx - mx, dy = qy - my, dz = qz - mz;
    return __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
}

struct Box { float lo[3], hi[3]; };

// ---- wave-wide min / max on the DPP path (round 6) ----------------------------------------------------------------------------
// Every tiled kernel starts an item with the box of its wave's queries: six reductions over 64 lanes.  As __shfl_xor butterflies
// (six ds_bpermute with their index arithmetic per value) that was ~250 instructions of the 2 540 a 64-query item of k_nn_tiled
// issues; as DPP row rotations (four steps inside the rows of 16 lanes) + the two row broadcasts of a wave-wide reduction
// (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3) it is seven instructions a value, none through the LDS
// crossbar; lane 63 holds the result, returned as a scalar.  min / max are exact and order-free: no result changes.
template <int CTRL> __device__ __forceinline__ float dpp_f(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL> __device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
template <int CTRL, int ROWS> __device__ __forceinline__ float dpp_rows_f(float v)   // (rows outside ROWS keep v: op(v, v) = v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROWS, 0xf, false));
}
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E;                                   // quad_perm [1,0,3,2] / [2,3,0,1]
constexpr int kDppRor1 = 0x121, kDppRor2 = 0x122, kDppRor4 = 0x124, kDppRor8 = 0x128;   // row_ror:n (a row = 16 lanes)
constexpr int kDppBcast15 = 0x142, kDppBcast31 = 0x143;
__device__ __forceinline__ float wave_min_f(float v)
{
    v = fminf(v, dpp_f<kDppRor1>(v)); v = fminf(v, dpp_f<kDppRor2>(v)); v = fminf(v, dpp_f<kDppRor4>(v)); v = fminf(v, dpp_f<kDppRor8>(v));
    v = fminf(v, dpp_rows_f<kDppBcast15, 0xA>(v)); v = fminf(v, dpp_rows_f<kDppBcast31, 0xC>(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max_f(float v)
{
    v = fmaxf(v, dpp_f<kDppRor1>(v)); v = fmaxf(v, dpp_f<kDppRor2>(v)); v = fmaxf(v, dpp_f<kDppRor4>(v)); v = fmaxf(v, dpp_f<kDppRor8>(v));
    v = fmaxf(v, dpp_rows_f<kDppBcast15, 0xA>(v)); v = fmaxf(v, dpp_rows_f<kDppBcast31, 0xC>(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ bool box_overlap(const float* __restrict__ b, int stride, int i, const Box& w)
{
    // b: SoA [6][stride] = minx,miny,minz,maxx,maxy,maxz ; empty boxes are (+inf,-inf)
    return b[i] <= w.hi[0] && b[stride + i] <= w.hi[1] && b[2 * stride + i] <= w.hi[2] &&
           b[3 * stride + i] >= w.lo[0] && b[4 * stride + i] >= w.lo[1] && b[5 * stride + i] >= w.lo[2];
}

__device__ __forceinline__ float bcast_lane(float v, int lane_uniform)
{
    // lane_uniform is wave-uniform (ctz of a ballot): a v_readlane, not an LDS round trip
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane_uniform));
}

// the sorted map + its three box levels, as the tiled kernels see it
struct TiledMap {
    const float *sx, *sy, *sz;  // Hilbert-sorted points, padded to whole super-tiles
    const int* perm;            // sorted position -> original index (0x7fffffff in the padding)
    const float* tbox; int n_tiles_p;   // tile boxes        SoA [6][n_tiles_p]
    const float* sbox; int n_super;     // super-tile boxes  SoA [6][n_super]  (n_super padded to 64)
    const float* ubox; int n_top;       // top boxes         SoA [6][n_top]
};

constexpr size_t kMaxLdsBoxBytes = 40 * 1024;  // upper box levels kept in LDS up to this size (~3.4M map points)
constexpr size_t kDbgItems = 1u << 17;  // MOLA_ICP_DEBUG_STATS: per-item records for clouds up to 16M points
constexpr int kQueues = 8, kQueueStride = 32;  // work-queue counters, one 128-byte line each
constexpr int kMaxList = 64;   // super-tiles collected before their tiles are streamed
constexpr int kStatSlots = 64, kStatStride = 16;  // slotted statistics counters (u64 units: one 128-byte line per slot)
constexpr int kHalfFlag = 0x40000000;

// One workgroup per item (the cooperative kernels): workgroups are dealt to the 8 XCDs round-robin (XCD = blockIdx & 7), so
// item = blockIdx would send every XCD over the whole cloud -- each of the 8 L2s ends up holding all of it.  Item ranges
// instead: XCD c takes the items [c * per, (c + 1) * per), contiguous = neighbours in space, and its L2 holds an eighth of the
// clouds (what the persistent kernels' per-XCD segments do).  The grid is rounded up to 8 * per workgroups; >= n_items: none.
__device__ __forceinline__ int xcd_item(int block, int n_items)
{
    const int per = (n_items + 7) >> 3, r = block >> 3;
    return r < per ? (block & 7) * per + r : n_items;   // (a grid sized for a larger problem of the batch: nothing for this one)
}
__host__ __device__ __forceinline__ int xcd_grid(int n_items) { return ((n_items + 7) >> 3) << 3; }  // work-list entry = one 64-query half of a 128-query item (id = 2 * item + half)

// LDS copy of the two upper box levels (one per workgroup): [6][n_top] then [6][n_super] floats.  The upper
// levels of the scan then cost LDS reads instead of dependent global round trips.
__device__ __forceinline__ size_t lds_box_floats(int n_top, int n_super) { return 6u * ((size_t)n_top + (size_t)n_super); }
__device__ __forceinline__ void load_boxes_to_lds(const TiledMap& mp, lds_f32* lbox)
{
    const int nu = 6 * mp.n_top, ns = 6 * mp.n_super;
    for (int i = threadIdx.x; i < nu; i += blockDim.x) lbox[i] = mp.ubox[i];
    for (int i = threadIdx.x; i < ns; i += blockDim.x) lbox[nu + i] = mp.sbox[i];
    __syncthreads();
}

// The sweep shared by the tiled kernels: wave box from the per-query reaches; the two upper box levels select
// the super-tiles some query reaches (from the LDS copy `lbox`, or from global memory if it is null) into the
// per-wave list `slist`; the listed super-tiles are then streamed with the NEXT one's tile boxes already in
// flight, their surviving tiles staged through LDS two at a time with the next pair's points in flight too.
// `visit(nm, jb0, jb1)` is called once per staged pass: sm[0..2][0..nm) hold x,y,z of the staged points
// (sm[3] their original indices if NEED_PERM); points [0,32) have sorted positions jb0.., [32,64) jb1...
// Returns the number of staged points.
template <int QPL, bool NEED_PERM, class Visit>
__device__ __forceinline__ unsigned long long tiled_sweep(const TiledMap& mp, const lds_f32* lbox, bool use_lbox, int* slist, int lane,
                                                          float (*sm)[64], const float (&qx)[QPL], const float (&qy)[QPL],
                                                          const float (&qz)[QPL], const float (&reach)[QPL],
                                                          const float (&bound2)[QPL], Visit&& visit,
                                                          bool prof, unsigned long long& p_stage,
                                                          unsigned long long& p_visit, unsigned int& p_supers,
                                                          unsigned int& p_entered, unsigned int& p_tiles,
                                                          unsigned long long& p_boxwait, unsigned long long& p_tiletest)
{
    Box w;
#pragma unroll
    for (int a = 0; a < 3; ++a) { w.lo[a] = INFINITY; w.hi[a] = -INFINITY; }
#pragma unroll
    for (int k = 0; k < QPL; ++k) {
        if (reach[k] >= 0.f) {
            w.lo[0] = fminf(w.lo[0], qx[k] - reach[k]); w.hi[0] = fmaxf(w.hi[0], qx[k] + reach[k]);
            w.lo[1] = fminf(w.lo[1], qy[k] - reach[k]); w.hi[1] = fmaxf(w.hi[1], qy[k] + reach[k]);
            w.lo[2] = fminf(w.lo[2], qz[k] - reach[k]); w.hi[2] = fmaxf(w.hi[2], qz[k] + reach[k]);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) { w.lo[a] = wave_min_f(w.lo[a]); w.hi[a] = wave_max_f(w.hi[a]); }
    unsigned long long n_staged = 0;

    // Can the box (m0..m5 = min xyz, max xyz; wave-uniform values) hold a point with d2 <= bound2 for ANY query of
    // the wave?  Per query: squared distance to the box, computed with the contract's own operation sequence on
    // the per-axis gaps.  Rounding is monotone, so for every point p inside the box gap_a <= |q_a - p_a| after
    // rounding, hence box_d2 <= d2_contract(q, p) EXACTLY as computed -- no margin needed, and bound2 is read
    // live: as a query's best shrinks during the sweep, later boxes are tested against the tighter value.
    // Padding lanes carry bound2 < 0 and reach nothing; empty boxes (+inf, -inf) give inf.
    auto any_reach = [&](float m0, float m1, float m2, float m3, float m4, float m5) -> bool {
        if constexpr (QPL == 2) {  // both queries of the lane per packed instruction
            // gap to the box per axis = q - clamp(q, lo, hi) (v_med3_f32): the same magnitude as max(lo - q, q - hi, 0),
            // one instruction less per axis; only boxes that passed the wave-box test come here (never an empty one)
            const v2f s_qx = {qx[0], qx[1]}, s_qy = {qy[0], qy[1]}, s_qz = {qz[0], qz[1]};
            const v2f cx = {__builtin_amdgcn_fmed3f(qx[0], m0, m3), __builtin_amdgcn_fmed3f(qx[1], m0, m3)};
            const v2f cy = {__builtin_amdgcn_fmed3f(qy[0], m1, m4), __builtin_amdgcn_fmed3f(qy[1], m1, m4)};
            const v2f cz = {__builtin_amdgcn_fmed3f(qz[0], m2, m5), __builtin_amdgcn_fmed3f(qz[1], m2, m5)};
            const v2f ax = s_qx - cx, ay = s_qy - cy, az = s_qz - cz;
            const v2f D = __builtin_elementwise_fma(az, az, __builtin_elementwise_fma(ay, ay, ax * ax));
            return __any(D.x <= bound2[0] || D.y <= bound2[1]);
        } else {
            // (q - clamp(q, lo, hi), as above: two instructions per axis instead of four -- this one-query form is the default now)
            const float ax = qx[0] - __builtin_amdgcn_fmed3f(qx[0], m0, m3);
            const float ay = qy[0] - __builtin_amdgcn_fmed3f(qy[0], m1, m4);
            const float az = qz[0] - __builtin_amdgcn_fmed3f(qz[0], m2, m5);
            return __any(fmaf(az, az, fmaf(ay, ay, ax * ax)) <= bound2[0]);
        }
    };

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
        if (prof) { const unsigned long long tp2 = __builtin_amdgcn_s_memtime(); p_stage += tp1 - tp0; p_visit += tp2 - tp1; }
    };

    // ---- the listed super-tiles: tile boxes of entry e+1 in flight while entry e's tiles are processed ----
    int n_list = 0;
    auto process_list = [&]() {
        if (n_list == 0) return;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int S = __builtin_amdgcn_readfirstlane(slist[0]);
        int ti = S * kSuper + lane;
        float n0 = mp.tbox[ti], n1 = mp.tbox[mp.n_tiles_p + ti], n2 = mp.tbox[2 * mp.n_tiles_p + ti],
              n3 = mp.tbox[3 * mp.n_tiles_p + ti], n4 = mp.tbox[4 * mp.n_tiles_p + ti], n5 = mp.tbox[5 * mp.n_tiles_p + ti];
        for (int e = 0; e < n_list; ++e) {
            const unsigned long long tb0 = prof ? __builtin_amdgcn_s_memtime() : 0ull;
            const float b0 = n0, b1 = n1, b2 = n2, b3 = n3, b4 = n4, b5 = n5;
            const int Sc = S;
            if (e + 1 < n_list) {
                S = __builtin_amdgcn_readfirstlane(slist[e + 1]);
                ti = S * kSuper + lane;
                n0 = mp.tbox[ti]; n1 = mp.tbox[mp.n_tiles_p + ti]; n2 = mp.tbox[2 * mp.n_tiles_p + ti];
                n3 = mp.tbox[3 * mp.n_tiles_p + ti]; n4 = mp.tbox[4 * mp.n_tiles_p + ti]; n5 = mp.tbox[5 * mp.n_tiles_p + ti];
            }
            unsigned long long cand = __ballot(b0 <= w.hi[0] && b1 <= w.hi[1] && b2 <= w.hi[2] && b3 >= w.lo[0] &&
                                               b4 >= w.lo[1] && b5 >= w.lo[2]);
            // Tiles are tested right before they are staged, not all up front: the per-query bound is live, so a
            // tile tested after its neighbours were swept meets the tighter best (a first launch, or one after a
            // large pose step, culls many of a super-tile's later tiles this way).
            const unsigned long long tb1 = prof ? __builtin_amdgcn_s_memtime() : 0ull;
            auto next_tile = [&]() -> int {  // next candidate some query still reaches, or -1
                while (cand) {
                    const int t = __builtin_ctzll(cand);
                    cand &= cand - 1;
                    if (prof) p_tiles += 1;
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
            if (prof) { const unsigned long long tb2 = __builtin_amdgcn_s_memtime(); p_boxwait += tb1 - tb0; p_tiletest += tb2 - tb1; }
        }
        __builtin_amdgcn_wave_barrier();  // the list is rewritten from here on
        n_list = 0;
    };

    // ---- upper levels: top boxes (64 super-tiles = 131072 points each) -> super-tile boxes ----
    // Written as a resumable scan so that process_list() has ONE call site (its body holds the distance
    // passes): collect up to kMaxList super-tiles, stream them, resume where the scan stopped.
    const lds_f32* l_ubox = lbox;
    const lds_f32* l_sbox = lbox + 6 * mp.n_top;
    int ub = 0, sb = 0;
    unsigned long long ucand = 0, scand = 0;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f, c4 = 0.f, c5 = 0.f;
    bool c_valid = false;  // c0..c5 hold the super-tile boxes [sb, sb+64)
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
    for (;;) {
        while (n_list < kMaxList) {
            if (scand) {
                if (!c_valid) load_super_boxes();  // resumed after a full list
                const int sl = __builtin_ctzll(scand);
                scand &= scand - 1;
                if (prof) p_supers += 1;
                // super-tile vs the individual queries: a bimodal query group must not descend everywhere
                if (any_reach(bcast_lane(c0, sl), bcast_lane(c1, sl), bcast_lane(c2, sl), bcast_lane(c3, sl),
                              bcast_lane(c4, sl), bcast_lane(c5, sl))) {
                    if (prof) p_entered += 1;
                    if (lane == 0) slist[n_list] = sb + sl;
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
        if (n_list == 0) break;
        process_list();
        c_valid = false;
    }
    if (pend_a >= 0) compute_pending(-1, -1);
    return n_staged;
}

// ---- the quad sweep: k_nn_tiled's fast attempt on 64-query items (QPL = 1) --------------------------------------------------------
// tiled_sweep evaluates every staged tile for all 64 queries of the wave, although a tile matters only to the queries near it: at
// 1M x 1M an item's queries reach 9.6 tiles between them and 2.5 each, against a 10M-point map 34 and 2.7 (306 / 1 082 evaluated
// pairs per query).  Here the tile TESTS stay per wave (one box against 64 live bounds, as before) but their ballots are kept per
// QUAD of 16 lanes -- 16 consecutive sorted queries, a quarter of the item's volume: a tile that passes is entered in the list of
// every quad with a lane that reaches it (the lists live in two vector registers: entry n of quad g in lane 16 g + n).  The quads walk their OWN lists side
// by side: in round r quad g takes its r-th tile -- four different tiles go from global memory straight into LDS
// (global_load_lds_dword, the next round's in flight under this round's distances), each quad reads its own 32 points -- so every
// lane meets only the tiles its quad listed: 5.3 of the item's 9.6 at 1M x 1M, the fullest quad (= the rounds) 6.8; 12.3 / 15.0 of
// 33.8 against the 10M-point map (LAB_NOTEBOOK.md, round 5).  Exact for the reason tiled_sweep is: a tile a lane does not meet is
// one no lane of its quad reaches.  Bookkeeping per 8-point group as nn_visit_fast (best, group position, groups at the best): the
// results are bit-identical.  The price: all tile tests of a list see the bounds as they stood before the first evaluated point.
constexpr int kQuadBuf = 3 * 2 * 64;   // floats of one round's tiles: x / y / z rows, two blocks of 64 (lanes' first / second point)

__device__ __forceinline__ unsigned long long quad_sweep(const TiledMap& mp, const lds_f32* lbox, bool use_lbox, int* slist, int lane,
                                                         float* sq /*wave-private, 2 * kQuadBuf floats*/,
                                                         float qx, float qy, float qz, float reach, float& best, int& bpos, int& cnt)
{
    Box w;
    w.lo[0] = reach >= 0.f ? qx - reach : INFINITY; w.hi[0] = reach >= 0.f ? qx + reach : -INFINITY;
    w.lo[1] = reach >= 0.f ? qy - reach : INFINITY; w.hi[1] = reach >= 0.f ? qy + reach : -INFINITY;
    w.lo[2] = reach >= 0.f ? qz - reach : INFINITY; w.hi[2] = reach >= 0.f ? qz + reach : -INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a) { w.lo[a] = wave_min_f(w.lo[a]); w.hi[a] = wave_max_f(w.hi[a]); }   // (scalars from here on)
    unsigned long long n_rounds = 0;
    // which lanes can the box (wave-uniform values) hold a point for, under their LIVE bounds (tiled_sweep::any_reach, as a ballot)
    auto reach_ballot = [&](float m0, float m1, float m2, float m3, float m4, float m5) -> unsigned long long {
        const float ax = qx - __builtin_amdgcn_fmed3f(qx, m0, m3);
        const float ay = qy - __builtin_amdgcn_fmed3f(qy, m1, m4);
        const float az = qz - __builtin_amdgcn_fmed3f(qz, m2, m5);
        return __ballot(fmaf(az, az, fmaf(ay, ay, ax * ax)) <= best);
    };

    // ---- the quads' lists: entry n of quad g's list sits in lane 16 g + (n & 15) of qla (n < 16) or qlb (n < 32); n0..n3 entries
    // (wave-uniform).  A lane reads its quad's r-th entry with one ds_bpermute.
    int qla = -1, qlb = -1;
    int n0 = 0, n1 = 0, n2 = 0, n3 = 0;
    const int quad = lane >> 4, sub = lane & 15;
    auto round_loads = [&](int mt, int buf) {   // this lane's two points of its quad's tile -> LDS, no register in between
        if (mt >= 0) {
            // (scalar base + a 32-bit byte offset per lane: the address costs no 64-bit vector arithmetic)
            const unsigned int bo = (unsigned int)(mt * kTileG + sub) * 4u;
            float* b = sq + buf * kQuadBuf;
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const char*>(mp.sx) + bo, b, 4, 0, 0);
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const char*>(mp.sx) + bo, b + 64 - 16, 4, 64, 0);   // (+ 64 bytes on BOTH sides: the second 16 points, the second block)
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const char*>(mp.sy) + bo, b + 128, 4, 0, 0);
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const char*>(mp.sy) + bo, b + 192 - 16, 4, 64, 0);   // (+ 64 bytes on BOTH sides: the second 16 points, the second block)
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const char*>(mp.sz) + bo, b + 256, 4, 0, 0);
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const char*>(mp.sz) + bo, b + 320 - 16, 4, 64, 0);   // (+ 64 bytes on BOTH sides: the second 16 points, the second block)
        }
    };
    int n_mine = 0;   // (run_rounds: this lane's quad's entry count)
    auto my_tile = [&](int r) -> int {   // the r-th entry of this lane's quad's list, -1 behind its end
        const int t = __shfl(r < 16 ? qla : qlb, (lane & 48) | (r & 15));
        return r < n_mine ? t : -1;
    };
    auto run_rounds = [&]() {
        const int nr = max(max(n0, n1), max(n2, n3));
        if (nr == 0) return;
        n_mine = quad == 0 ? n0 : (quad == 1 ? n1 : (quad == 2 ? n2 : n3));
        int mt = my_tile(0);   // (its loads were issued when the quad's list got its first entry: take_tile)
        for (int r = 0; r < nr; ++r) {
            const int mt_next = r + 1 < nr ? my_tile(r + 1) : -1;
            round_loads(mt_next, (r + 1) & 1);            // the next round's tiles fly under this round's distances
            if (r + 1 < nr) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // (in order: all but the six just issued have landed)
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const float* b = sq + (r & 1) * kQuadBuf + 16 * quad;
            const bool live = mt >= 0;
            ++n_rounds;
#pragma unroll
            for (int g = 0; g < 4; ++g) {   // 8-point groups: points 0..15 in the first block, 16..31 in the second
                const float* bx = b + (g >> 1) * 64 + (g & 1) * 8;
                const float4 X0 = *reinterpret_cast<const float4*>(bx);
                const float4 X1 = *reinterpret_cast<const float4*>(bx + 4);
                const float4 Y0 = *reinterpret_cast<const float4*>(bx + 128);
                const float4 Y1 = *reinterpret_cast<const float4*>(bx + 132);
                const float4 Z0 = *reinterpret_cast<const float4*>(bx + 256);
                const float4 Z1 = *reinterpret_cast<const float4*>(bx + 260);
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
                gm = live ? gm : INFINITY;   // (a quad whose list has ended reads what an earlier round left in its part of the buffer)
                const int gpos = mt * kTileG + 8 * g;
                const bool lt = gm < best;
                const int eq = (int)(gm == best);
                cnt = lt ? 1 : cnt + eq;
                bpos = lt ? gpos : bpos;
                best = fminf(best, gm);
            }
            __builtin_amdgcn_wave_barrier();   // (this buffer is the target of the loads issued in the next iteration)
            mt = mt_next;
        }
        n0 = n1 = n2 = n3 = 0;
    };
    auto take_tile = [&](int tile, unsigned long long m) {   // a tile some lane reaches: into the list of every quad that has such a lane
        // (a list's FIRST tile is sent for at once, by the quad's own lanes: the first round then waits for little -- the remaining
        //  tile tests of the item run under its loads, as tiled_sweep's first pair did)
        auto append = [&](int g, int& n) {
            const int at = 16 * g + (n & 15);
            if (n < 16) qla = lane == at ? tile : qla; else qlb = lane == at ? tile : qlb;
            if (n == 0) round_loads(quad == g ? tile : -1, 0);
            ++n;
        };
        if (m & 0xffffull) append(0, n0);
        if ((m >> 16) & 0xffffull) append(1, n1);
        if ((m >> 32) & 0xffffull) append(2, n2);
        if (m >> 48) append(3, n3);
        if (max(max(n0, n1), max(n2, n3)) == 32) run_rounds();   // (a list is full: seen in launches without seeds and against maps many times denser than the scan)
    };

    // ---- the listed super-tiles: tile boxes of entry e+1 in flight while entry e's tiles are tested ----
    int n_list = 0;
    auto process_list = [&]() {
        if (n_list == 0) return;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int S = __builtin_amdgcn_readfirstlane(slist[0]);
        int ti = S * kSuper + lane;
        float f0 = mp.tbox[ti], f1 = mp.tbox[mp.n_tiles_p + ti], f2 = mp.tbox[2 * mp.n_tiles_p + ti],
              f3 = mp.tbox[3 * mp.n_tiles_p + ti], f4 = mp.tbox[4 * mp.n_tiles_p + ti], f5 = mp.tbox[5 * mp.n_tiles_p + ti];
        for (int e = 0; e < n_list; ++e) {
            const float b0 = f0, b1 = f1, b2 = f2, b3 = f3, b4 = f4, b5 = f5;
            const int Sc = S;
            if (e + 1 < n_list) {
                S = __builtin_amdgcn_readfirstlane(slist[e + 1]);
                ti = S * kSuper + lane;
                f0 = mp.tbox[ti]; f1 = mp.tbox[mp.n_tiles_p + ti]; f2 = mp.tbox[2 * mp.n_tiles_p + ti];
                f3 = mp.tbox[3 * mp.n_tiles_p + ti]; f4 = mp.tbox[4 * mp.n_tiles_p + ti]; f5 = mp.tbox[5 * mp.n_tiles_p + ti];
            }
            unsigned long long cand = __ballot(b0 <= w.hi[0] && b1 <= w.hi[1] && b2 <= w.hi[2] && b3 >= w.lo[0] &&
                                               b4 >= w.lo[1] && b5 >= w.lo[2]);
            while (cand) {
                const int t = __builtin_ctzll(cand);
                cand &= cand - 1;
                const unsigned long long m = reach_ballot(bcast_lane(b0, t), bcast_lane(b1, t), bcast_lane(b2, t), bcast_lane(b3, t),
                                                          bcast_lane(b4, t), bcast_lane(b5, t));
                if (m) take_tile(Sc * kSuper + t, m);
            }
        }
        __builtin_amdgcn_wave_barrier();  // the list is rewritten from here on
        n_list = 0;
        run_rounds();
    };

    // ---- upper levels (as tiled_sweep: top boxes -> super-tile boxes, per-query tests, a resumable scan) ----
    auto any_reach = [&](float m0, float m1, float m2, float m3, float m4, float m5) -> bool { return reach_ballot(m0, m1, m2, m3, m4, m5) != 0ull; };
    const lds_f32* l_ubox = lbox;
    const lds_f32* l_sbox = lbox + 6 * mp.n_top;
    int ub = 0, sb = 0;
    unsigned long long ucand = 0, scand = 0;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f, c4 = 0.f, c5 = 0.f;
    bool c_valid = false;
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
    for (;;) {
        while (n_list < kMaxList) {
            if (scand) {
                if (!c_valid) load_super_boxes();
                const int sl = __builtin_ctzll(scand);
                scand &= scand - 1;
                if (any_reach(bcast_lane(c0, sl), bcast_lane(c1, sl), bcast_lane(c2, sl), bcast_lane(c3, sl),
                              bcast_lane(c4, sl), bcast_lane(c5, sl))) {
                    if (lane == 0) slist[n_list] = sb + sl;
                    ++n_list;
                }
            } else if (ucand) {
                sb = (ub - 64 + __builtin_ctzll(ucand)) * 64;
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
                if (__builtin_popcountll(ucand) > 4) {   // (a spread query group: the top boxes per query first -- see tiled_sweep)
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
        if (n_list == 0) break;
        process_list();
        c_valid = false;
    }
    return n_rounds * (unsigned long long)kTileG;   // points every lane met
}

// reach of a query whose current best squared distance is `best`: any m with d2_contract <= best lies inside
// [q - r, q + r] per axis (sqrt rounded up, plus 2 ulp of the largest coordinate)
__device__ __forceinline__ float reach_of(float best, float qx, float qy, float qz)
{
    const float cmax = fmaxf(fabsf(qx), fmaxf(fabsf(qy), fabsf(qz)));
    return sqrtf(best * 1.000002f) * 1.00001f + cmax * 2.4e-7f + 1e-30f;
}

// Work queue of the persistent waves over the entries [0, n).  Same-address atomics serialise device-wide
// (measured: ~13 ns each; one more atomic per item cost 20% of the kernel, the 3072-deep burst of first pops 40 us)
// and every XCD has its own L2, so
//  - the entries are cut into kQueues contiguous segments -- contiguous entries are neighbours in space and share
//    map tiles -- and segment c belongs to XCD c (blocks are dealt to the XCDs round-robin: XCD = blockIdx & 7):
//    the tiles a region needs are then fetched into ONE L2 instead of all eight;
//  - the first entry of every wave is fixed (its rank among the waves of its XCD): no atomics at kernel start;
//  - the rest of a segment is handed out by its own counter (separate cache lines); a wave whose segment has run
//    dry steals from the next ones.
//  - a DRY segment is not popped again.  A failed pop is as dear as a successful one (a device-scope read-modify-write
//    on one of 8 addresses, ~13 ns each, serialised), and at the end of a launch every wave used to walk all 8 dry
//    counters before it left: 8 x 3072 failed pops against the ~6000 useful ones, all inside the drain -- the last
//    entry of a wave "ran" 0.4 of the launch, most of it in that queue of atomics.  (Reading the counters first is no
//    way out: coherent loads of those 8 hot lines serialise with the atomics -- measured, 135 -> 204 us.)  Instead the
//    news travels IN the counters: bits 24..31 of every counter hold the set of segments known to be dry.  The wave
//    whose pop is the FIRST to fail on segment c (its ticket equals the segment's size -- exactly one wave) ORs bit c
//    into the other seven counters; every pop, successful or not, returns the set as of that moment, and a wave pops
//    only segments outside the set it has seen.  A finishing wave pays one failed pop, not eight.
// Callers reserve the next entry late (behind the epilogue's loads): see k_nn_tiled.
constexpr int kQueueCountBits = 24;  // a counter = tickets handed out (low 24 bits: < 16M entries per segment) | dry set << 24
struct WaveQueue {
    unsigned int* q;
    const int* seg;  // kQueues + 1 segment boundaries (k_order_items: equal COST per segment), or null: equal counts
    int lane, n;
    unsigned int dry;  // wave-uniform: bit c = segment c has nothing left to hand out
    int cur;           // wave-uniform: the segment the pop in flight went to (-1: none was worth trying)
    unsigned int seen; // lane 0: the dry set the pop in flight returned
    __device__ __forceinline__ WaveQueue(unsigned int* queue, int lane_, int n_entries, const int* seg_ = nullptr)
        : q(queue), seg(seg_), lane(lane_), n(n_entries), dry(0u), cur(-1), seen(0u) {}
    __device__ __forceinline__ int seg_begin(int c) const { return seg ? seg[c] : (int)(((long long)c * n) / kQueues); }
    __device__ __forceinline__ int n_static(int c) const  // waves of XCD c = fixed first entries of segment c
    {
        return (((int)gridDim.x + kQueues - 1 - c) / kQueues) * 4;
    }
    __device__ __forceinline__ int global_wave() const { return (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6); }
    __device__ __forceinline__ int pop()  // the next entry (meaningful in lane 0; >= n: none there), still in flight
    {
        // first segment not known to be dry, starting at this XCD's own
        const unsigned int own = (unsigned int)blockIdx.x & (kQueues - 1);
        const unsigned int open = ~dry & ((1u << kQueues) - 1u);
        const unsigned int rot = ((open >> own) | (open << (kQueues - own))) & ((1u << kQueues) - 1u);
        cur = rot ? (int)((own + (unsigned int)__builtin_ctz(rot)) & (kQueues - 1)) : -1;
        int r = 0x7fffffff;
        seen = 0u;
        if (cur >= 0 && lane == 0) {
            const unsigned int old = atomicAdd(q + cur * kQueueStride, 1u);
            const int ticket = (int)(old & ((1u << kQueueCountBits) - 1u));
            const int avail = seg_begin(cur + 1) - seg_begin(cur) - n_static(cur);  // entries this segment hands out
            seen = old >> kQueueCountBits;
            if (ticket < avail) r = seg_begin(cur) + n_static(cur) + ticket;
            else if (ticket == (avail > 0 ? avail : 0)) seen |= 0x100u;  // the first pop to fail here: this wave tells the others
        }
        return r;
    }
    __device__ __forceinline__ int settle(int raw)  // raw = readfirstlane(pop()): none there -> the segments still open
    {
        for (;;) {
            const unsigned int sn = (unsigned int)__builtin_amdgcn_readfirstlane((int)seen);
            dry |= sn & ((1u << kQueues) - 1u);
            if (raw < n || cur < 0) return raw;
            dry |= 1u << cur;
            if ((sn & 0x100u) && lane < kQueues && lane != cur)  // one vector atomic, nothing returned
                atomicOr(q + lane * kQueueStride, 1u << (kQueueCountBits + cur));
            raw = __builtin_amdgcn_readfirstlane(pop());
        }
    }
    __device__ __forceinline__ int first()  // wave-uniform; >= n: nothing left anywhere
    {
        const int c = (int)blockIdx.x & (kQueues - 1);
        const int e = seg_begin(c) + ((int)blockIdx.x / kQueues) * 4 + (int)(threadIdx.x >> 6);
        if (e < seg_begin(c + 1)) return e;
        return settle(__builtin_amdgcn_readfirstlane(pop()));  // more waves than entries in this segment
    }
};

// The two visitors of the NN kernels (one staged pass: sm[0..2][0..nm) = x, y, z, sm[3] = original indices; points
// [0,32) have sorted positions jb0.., [32,64) jb1..), shared by k_nn_tiled and the cooperative k_nn_coop.
// exact: per-pair argmin on the packed key (d2 bits << 32 | original index) -- the full lexicographic rule
template <int QPL>
__device__ __forceinline__ void nn_visit_exact(float (*sm)[64], int nm, int jb0, int jb1, const float (&qx)[QPL],
                                               const float (&qy)[QPL], const float (&qz)[QPL],
                                               unsigned long long (&key)[QPL], float (&best)[QPL], int (&bpos)[QPL])
{
    for (int m = 0; m < nm; m += 4) {
        const float4 X = *reinterpret_cast<const float4*>(&sm[0][m]);
        const float4 Y = *reinterpret_cast<const float4*>(&sm[1][m]);
        const float4 Z = *reinterpret_cast<const float4*>(&sm[2][m]);
        const float4 O = *reinterpret_cast<const float4*>(&sm[3][m]);
        const float xs[4] = {X.x, X.y, X.z, X.w}, ys[4] = {Y.x, Y.y, Y.z, Y.w}, zs[4] = {Z.x, Z.y, Z.z, Z.w};
        const unsigned int os[4] = {__float_as_uint(O.x), __float_as_uint(O.y), __float_as_uint(O.z),
                                    __float_as_uint(O.w)};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int k = 0; k < QPL; ++k) {
                const float d = dist2(qx[k], qy[k], qz[k], xs[u], ys[u], zs[u]);
                const unsigned long long ck = ((unsigned long long)__float_as_uint(d) << 32) | os[u];
                const bool better = ck < key[k];
                key[k] = better ? ck : key[k];
                best[k] = better ? d : best[k];  // the sweep's box tests read it
                bpos[k] = better ? ((m + u) < 32 ? jb0 + m + u : jb1 + m + u - 32) : bpos[k];
            }
        }
    }
}

// fast: per kGroup-point group only the group minimum meets the running best; (best, group position, groups at the best)
template <int QPL>
__device__ __forceinline__ void nn_visit_fast(float (*sm)[64], int nm, int jb0, int jb1, const float (&qx)[QPL],
                                              const float (&qy)[QPL], const float (&qz)[QPL], float (&best)[QPL],
                                              int (&bpos)[QPL], int (&cnt)[QPL])
{
    // per kGroup-point group: packed sub/mul/fma (v_pk_*_f32) serve two (query, point) pairs per
    // instruction -- the lane's two queries (QPL = 2) or two consecutive points (QPL = 1) -- the group
    // minimum is a chain of v_min3, and the (best, position, tie) bookkeeping runs once per group
#pragma unroll 2
    for (int m = 0; m < nm; m += kGroup) {
        float gm[QPL];
#pragma unroll
        for (int k = 0; k < QPL; ++k) gm[k] = INFINITY;
#pragma unroll
        for (int h = 0; h < kGroup; h += 8) {
            const float4 X0 = *reinterpret_cast<const float4*>(&sm[0][m + h]);
            const float4 X1 = *reinterpret_cast<const float4*>(&sm[0][m + h + 4]);
            const float4 Y0 = *reinterpret_cast<const float4*>(&sm[1][m + h]);
            const float4 Y1 = *reinterpret_cast<const float4*>(&sm[1][m + h + 4]);
            const float4 Z0 = *reinterpret_cast<const float4*>(&sm[2][m + h]);
            const float4 Z1 = *reinterpret_cast<const float4*>(&sm[2][m + h + 4]);
            const float xs[8] = {X0.x, X0.y, X0.z, X0.w, X1.x, X1.y, X1.z, X1.w};
            const float ys[8] = {Y0.x, Y0.y, Y0.z, Y0.w, Y1.x, Y1.y, Y1.z, Y1.w};
            const float zs[8] = {Z0.x, Z0.y, Z0.z, Z0.w, Z1.x, Z1.y, Z1.z, Z1.w};
            if constexpr (QPL == 2) {
                const v2f q2x = {qx[0], qx[1]}, q2y = {qy[0], qy[1]}, q2z = {qz[0], qz[1]};
#pragma unroll
                for (int u = 0; u < 8; u += 2) {
                    const v2f da = dist2_pk(q2x, q2y, q2z, xs[u], ys[u], zs[u]);
                    const v2f db = dist2_pk(q2x, q2y, q2z, xs[u + 1], ys[u + 1], zs[u + 1]);
                    gm[0] = fminf(fminf(gm[0], da.x), db.x);
                    gm[1] = fminf(fminf(gm[1], da.y), db.y);
                }
            } else {
#pragma unroll
                for (int u = 0; u < 8; u += 2) {
                    const v2f mx = {xs[u], xs[u + 1]}, my = {ys[u], ys[u + 1]}, mz = {zs[u], zs[u + 1]};
                    const v2f dd = dist2_pk2(qx[0], qy[0], qz[0], mx, my, mz);
                    gm[0] = fminf(fminf(gm[0], dd.x), dd.y);
                }
            }
        }
        const int gpos = m < 32 ? jb0 + m : jb1 + m - 32;  // sorted position of this group
        // (best, its group, HOW MANY groups reached that best): a group that beats the best restarts the count, one that
        // equals it adds to it.  The seed starts with a count of 0 -- its own group is always swept (its tile holds a
        // point at exactly the bound) and counts itself -- so "count >= 2 at the end" is an exact distance tie between
        // two groups, which the exact-key pass resolves.  Six instructions per query and group.
#pragma unroll
        for (int k = 0; k < QPL; ++k) {
            const bool lt = gm[k] < best[k];
            const int eq = (int)(gm[k] == best[k]);
            cnt[k] = lt ? 1 : cnt[k] + eq;
            bpos[k] = lt ? gpos : bpos[k];
            best[k] = fminf(best[k], gm[k]);
        }
    }
}

// Row a8 fused into the persistent matcher: the 24 unit-weight sums of ONE 64-query item on the fp64 matrix cores.
// With u = [m, m l] and v = [m, g] per pairing (m = 1 if the query has a neighbour inside the gate, else 0) the sums
// W, sum l, sum g, sum l g^T are the 4 x 4 matrix sum u v^T; sum l l^T and sum d2 come out of a second one,
// u' = [m l, m d2], v' = [m l, m].  v_mfma_f64_4x4x4_4b_f64 adds FOUR such 4 x 4 products over four pairings each per
// instruction: lane L supplies A[block (L >> 2) & 3][row L & 3][k = L >> 4] and B[block][k][column L & 3], lane d receives
// D[block (d >> 2) & 3][row d >> 4][column d & 3] (layout probed on gfx950: tools/microbench/mfma_f64_4x4_layout.hip) --
// 16 pairings per instruction, 2 x 4 instructions per item on a pipe the matcher leaves idle, instead of a second pass
// over the pairing (k_accumulate: 10.7 us at 1M queries).  The lanes' pairings are transposed through the wave's staging
// area (free in the epilogue; 32 records of 8 floats at a time), the four blocks are added in block order and lanes
// 0..23 write the item's row: every row depends on its item's pairing only -- not on which wave ran it, nor when -- so
// the fixed-order row reduction that follows (k_reduce_items) keeps the accumulators bitwise reproducible.  The products
// of two fp32 values are exact in fp64; the sums differ from k_accumulate's by their order only.
__device__ __forceinline__ void item_row_mfma(float* __restrict__ smf /*256 floats of wave-private LDS*/, int lane, bool paired,
                                              float l0, float l1, float l2, float g0, float g1, float g2, float d2,
                                              double* __restrict__ row)
{
    const float m = paired ? 1.0f : 0.0f;
    const float r[8] = {m, paired ? l0 : 0.0f, paired ? l1 : 0.0f, paired ? l2 : 0.0f,
                        paired ? g0 : 0.0f, paired ? g1 : 0.0f, paired ? g2 : 0.0f, paired ? d2 : 0.0f};
    // record [m, m lx, m ly, m lz, gx, gy, gz, m d2] of pairing 16 t + 4 block + k (within a half of 32 pairings)
    const int e = lane & 3, rec = (((lane >> 2) & 3) << 2) + (lane >> 4);
    const int oA = rec * 8 + e;                     // u  = [m, l]
    const int oB = rec * 8 + (e ? 3 + e : 0);       // v  = [m, g]
    const int oA2 = rec * 8 + (e < 3 ? 1 + e : 7);  // u' = [l, d2]
    const int oB2 = rec * 8 + (e < 3 ? 1 + e : 0);  // v' = [l, m]
    double D1 = 0.0, D2 = 0.0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();  // (the reads of the previous half / the sweep's last pass are done)
        if ((lane >> 5) == h) {
            // (plain float stores, the type the reads below use: a float4 store and float loads of the same words are
            //  "no alias" to the compiler, and a release fence alone lets it hoist the loads above the stores -- it did)
#pragma unroll
            for (int q = 0; q < 8; ++q) smf[(lane & 31) * 8 + q] = r[q];
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const double a = (double)smf[oA + 128 * t], b = (double)smf[oB + 128 * t];
            const double a2 = (double)smf[oA2 + 128 * t], b2 = (double)smf[oB2 + 128 * t];
            D1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, D1, 0, 0, 0);
            D2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a2, b2, D2, 0, 0, 0);
        }
    }
    // the four blocks, added in block order; D element (block, i, j) sits in lane 16 i + 4 block + j
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    double* sd = reinterpret_cast<double*>(smf);
    sd[lane] = D1;
    sd[64 + lane] = D2;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < 24) {
        // the accumulator block's layout (kNAcc): W, sum l (3), sum g (3), sum l g^T (9, row-major), n, sum d2, sum l l^T (00 01 02 11 12 22)
        int src;
        if (lane == 0 || lane == 16) src = 0;                                  // D1(0,0): W = n for unit weights
        else if (lane < 4) src = 16 * lane;                                    // D1(i,0): sum l
        else if (lane < 7) src = lane - 3;                                     // D1(0,j): sum g
        else if (lane < 16) src = 16 * (1 + (lane - 7) / 3) + 1 + (lane - 7) % 3;   // D1(i,j): sum l g^T
        else if (lane == 17) src = 64 + 51;                                    // D2(3,3): sum d2
        else if (lane < 21) src = 64 + (lane - 18);                            // D2(0,j): l0 l0, l0 l1, l0 l2
        else if (lane < 23) src = 64 + 16 + 1 + (lane - 21);                   // D2(1,1), D2(1,2)
        else src = 64 + 32 + 2;                                                // D2(2,2)
        row[lane] = ((sd[src] + sd[src + 4]) + sd[src + 8]) + sd[src + 12];
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();  // the staging area goes back to the sweep
}

// Every entry runs the fast sweep: per (query, kGroup = 8 points) only the group minimum meets the running best
//   (nn_visit_fast); the winning group is re-evaluated once at the end to recover the exact point and the
//   lowest-original-index rule inside it.  If a second group reached the SAME minimum (exact ties: duplicate
//   points, lattices) the wave redoes the entry at once with the exact-key sweep: per-pair argmin on the packed key
//   (d2 bits << 32 | original index), i.e. the full lexicographic rule -- from the same seeds, nothing of the fast
//   attempt having been written.  (A second launch over a list of tied entries, as in round 1, cost 5 us per
//   iteration for a list that is almost always empty.)
// DIAG = false (the launch path): the per-phase / per-wave diagnostics are compiled out -- as run-time branches that never fire
// they still held 90 scalar registers' worth of spills and five VGPRs: 101 -> 94 us at C3.
template <int QPL, bool DIAG, bool QUADS = false /*QPL = 1, !DIAG: the fast attempt walks per-quad tile lists (quad_sweep)*/>
__global__ __launch_bounds__(256, QUADS ? 4 : 3) void k_nn_tiled(const float* __restrict__ slx, const float* __restrict__ sly,
                                                  const float* __restrict__ slz, int N, TiledMap mp, PoseF P, float thr2,
                                                  int use_seed, int* __restrict__ pos_s, int* __restrict__ idx_s,
                                                  float* __restrict__ d2_s, float* __restrict__ gs_x, float* __restrict__ gs_y,
                                                  float* __restrict__ gs_z, const int* __restrict__ item_order,
                                                  unsigned int* __restrict__ item_cost, unsigned int* __restrict__ queue,
                                                  unsigned long long* __restrict__ staged_total,
                                                  unsigned long long* __restrict__ dbg_stats, int lds_boxes,
                                                  unsigned long long* __restrict__ wave_times /*diagnostics, usually null*/,
                                                  int early_pop /*tuning knob: reserve the next entry at the START of this one*/,
                                                  double* __restrict__ item_rows /*QPL = 1: one row of kNAcc unit-weight sums per item (item_row_mfma); may be null*/)
{
    __shared__ __attribute__((aligned(16))) float s_m[4][4][64];  // per wave: x, y, z, original index of 64 staged points
    __shared__ int s_list[4][kMaxList];
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];  // the upper box levels, if they fit
    const int lane = threadIdx.x & 63;
    // (the quad flavour: the wave's number as a scalar -- its LDS areas are then scalar bases, M0 of the tile loads needs no read-back)
    const int wave = QUADS ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : (int)(threadIdx.x >> 6);
    if constexpr (!DIAG) { dbg_stats = nullptr; wave_times = nullptr; }  // (constants from here on: the diagnostics fold away)
    // the quad sweep's tiles (QPL = 1, fast attempt): per wave two rounds' worth (double-buffered), x / y / z rows
    __shared__ __attribute__((aligned(16))) float s_q[QUADS ? 4 : 1][QUADS ? 2 * kQuadBuf : 4];
    static_assert(!QUADS || (QPL == 1 && !DIAG), "the quad sweep serves 64-query items of the product build");
    float(*sm)[64] = s_m[wave];
    int* slist = s_list[wave];
    const lds_f32* lbox = (const lds_f32*)s_dyn;
    if (lds_boxes) load_boxes_to_lds(mp, (lds_f32*)s_dyn);
    // Entries of the work list.  QPL = 1: every entry is a 64-query item.  QPL = 2: an entry is a 128-query item (two
    // queries per lane) or, with kHalfFlag, ONE HALF of one (64 queries, one per lane): k_order_entries cuts the items
    // whose cost exceeds what a wave's fair share of the launch allows -- a single 128-query item of a dense region
    // took 1.3x that share, and the launch is as long as its longest wave.
    const int n_items = (N + 64 * QPL - 1) / (64 * QPL);
    const int order_cap = QPL == 2 ? 2 * n_items : n_items;  // layout of item_order: entries, then kQueues + 1 boundaries, then the count
    const int n_entries = (QPL == 2 && item_order) ? item_order[order_cap + kQueues + 1] : n_items;

    WaveQueue wq(queue, lane, n_entries, item_order ? item_order + order_cap : nullptr);  // boundaries follow the order
    auto lookup = [&](int raw) -> int {  // raw is wave-uniform; -1 = past the end
        if (raw >= n_entries) return -1;
        return item_order ? item_order[raw] : raw;  // heaviest entries of the last launch first
    };
    unsigned long long wave_staged = 0ull;
    const unsigned long long t_wave0 = wave_times ? wall_clock64() : 0ull;  // 100 MHz, the same on every XCD
    unsigned int wave_items = 0u;
    unsigned long long ph0 = 0ull, ph1 = 0ull, ph2 = 0ull, ph3 = 0ull, last_start = 0ull;
    unsigned int last_code = 0u;
    // one entry at the granularity QL (queries per lane); `item` counts in units of 64 * QL queries, `code` is the entry
    // as listed (what the cost record carries).  EX = false: the fast attempt (pops the next entry behind its first loads);
    // EX = true: the exact redo of the same entry (`next_in` = what the fast attempt popped).  Returns the next entry's code.
    auto run_item = [&](auto& self, auto ql_tag, auto ex_tag, int item, int code, int next_in) -> int {
        constexpr int QL = decltype(ql_tag)::value;
        constexpr bool EX = decltype(ex_tag)::value;
        constexpr int kQ = 64 * QL;
        if (!EX) ++wave_items;
        const unsigned long long t_item0 = __builtin_amdgcn_s_memtime();
        if (wave_times && !EX) { last_start = wall_clock64(); last_code = (unsigned int)code; }

        float qx[QL], qy[QL], qz[QL], reach[QL];
        unsigned long long key[QL];  // EXACT: packed (d2, original index)
        float best[QL];              // fast: running minimum
        int bpos[QL];                // EXACT: sorted position of the best point; fast: of its kGroup-point group
        int tie[QL] = {};  // fast: number of groups whose minimum equals the best (>= 2: an exact tie)
        // ONE round trip: the two queries of the lane and their seeds -- position, original index and COORDINATES as the
        // last launch's epilogue stored them (gs_*), so the seed distance needs no dependent second trip
        int qi[QL], js[QL];
        float lx[QL], ly[QL], lz[QL];
        float gsx[QL], gsy[QL], gsz[QL];
        unsigned int gso[QL] = {};
#pragma unroll
        for (int k = 0; k < QL; ++k) {
            qi[k] = item * kQ + k * 64 + lane;
            if (qi[k] >= N) qi[k] = N;  // padding lane
            const int ic = qi[k] < N ? qi[k] : N - 1;
            lx[k] = slx[ic]; ly[k] = sly[ic]; lz[k] = slz[ic];
            js[k] = -1; gsx[k] = gsy[k] = gsz[k] = 0.f;
            if (use_seed) {
                js[k] = pos_s[ic];
                gsx[k] = gs_x[ic]; gsy[k] = gs_y[ic]; gsz[k] = gs_z[ic];
                if (EX) gso[k] = (unsigned int)idx_s[ic];
            }
        }
        // The next entry is reserved LATE -- behind the epilogue's loads, consumed after its stores.  Popping at the start of
        // an entry (as this kernel did: the atomic's latency is then hidden for free) binds an entry to a wave one whole
        // entry ahead of its execution: the queue ran dry at 0.55-0.6 of the launch while reserved entries still BEGAN at
        // 0.8 of it -- the drain was two entries long instead of one.
        int next_raw_v = 0;
        if (!EX && early_pop) next_raw_v = wq.pop();  // (behind the loads above: memory results return in order)
#pragma unroll
        for (int k = 0; k < QL; ++k) xform(P, lx[k], ly[k], lz[k], qx[k], qy[k], qz[k]);
#pragma unroll
        for (int k = 0; k < QL; ++k) {
            key[k] = ((unsigned long long)__float_as_uint(thr2) << 32);  // (gate^2, index 0): "no neighbour" sentinel
            best[k] = thr2;
            bpos[k] = -1;
            const float d = dist2(qx[k], qy[k], qz[k], gsx[k], gsy[k], gsz[k]);
            if (js[k] >= 0 && d < thr2) {  // warm start: last iteration's neighbour is an exact candidate
                best[k] = d;
                bpos[k] = EX ? js[k] : (js[k] & ~(kGroup - 1));
                if (EX) key[k] = ((unsigned long long)__float_as_uint(d) << 32) | gso[k];
            }
            reach[k] = reach_of(best[k], qx[k], qy[k], qz[k]);
            if (qi[k] >= N) {  // padding lane: reaches nothing, is never written
                qx[k] = qy[k] = qz[k] = 1.0e18f;
                reach[k] = -1.0f;
                best[k] = -1.0f;
                bpos[k] = -1;
                key[k] = ((unsigned long long)__float_as_uint(thr2) << 32);  // (a padding lane loaded the last query's seed: the exact epilogue reads the point at bpos for any key below the gate)
            }
        }

        unsigned long long p_stage = 0ull, p_visit = 0ull, p_boxwait = 0ull, p_tiletest = 0ull;
        unsigned int p_supers = 0u, p_entered = 0u, p_tiles = 0u;
        const unsigned long long t_sweep0 = (dbg_stats || wave_times) ? __builtin_amdgcn_s_memtime() : 0ull;
        unsigned long long n_staged;
        if constexpr (QUADS && QL == 1 && !EX) {
            n_staged = quad_sweep(mp, lbox, lds_boxes != 0, slist, lane, &s_q[wave][0], qx[0], qy[0], qz[0], reach[0], best[0], bpos[0], tie[0]);
        } else {
        n_staged = tiled_sweep<QL, EX>(mp, lbox, lds_boxes != 0, slist, lane, sm, qx, qy, qz, reach, best, [&](int nm, int jb0, int jb1) {
            if constexpr (EX) nn_visit_exact<QL>(sm, nm, jb0, jb1, qx, qy, qz, key, best, bpos);
            else nn_visit_fast<QL>(sm, nm, jb0, jb1, qx, qy, qz, best, bpos, tie);
        }, dbg_stats != nullptr, p_stage, p_visit, p_supers, p_entered, p_tiles, p_boxwait, p_tiletest);
        }
        const unsigned long long t_sweep1 = (dbg_stats || wave_times) ? __builtin_amdgcn_s_memtime() : 0ull;
        int next_item_v = next_in;
        if (!EX && early_pop) next_item_v = lookup(wq.settle(__builtin_amdgcn_readfirstlane(next_raw_v)));

        bool any_tie = false;
        int rpos[QL], roi[QL];
        float rd[QL], wx[QL], wy[QL], wz[QL];  // (w*: the neighbour's coordinates, next launch's seed)
#pragma unroll
        for (int k = 0; k < QL; ++k) { rpos[k] = -1; roi[k] = -1; rd[k] = thr2; wx[k] = wy[k] = wz[k] = 0.f; }
        // fused accumulation (item_row_mfma): the query's own coordinates again -- re-read here, in flight with the
        // epilogue's other loads, rather than held in three registers through the sweep
        float al0 = 0.f, al1 = 0.f, al2 = 0.f;
        if constexpr (QPL == 1 && QL == 1) {
            if (item_rows) {
                const int ic = qi[0] < N ? qi[0] : N - 1;
                al0 = slx[ic]; al1 = sly[ic]; al2 = slz[ic];
            }
        }
        if constexpr (EX) {
#pragma unroll
            for (int k = 0; k < QL; ++k) {
                const float d = __uint_as_float((unsigned int)(key[k] >> 32));
                if (d < thr2) {
                    rd[k] = d; rpos[k] = bpos[k]; roi[k] = (int)(unsigned int)(key[k] & 0xffffffffu);
                    wx[k] = mp.sx[bpos[k]]; wy[k] = mp.sy[bpos[k]]; wz[k] = mp.sz[bpos[k]];
                }
            }
        } else {
            // resolve inside the winning group: the point(s) with d2 == best, lowest original index first.
            // One round trip: all loads of both queries are issued before the first use.
            float4 RX[QL][kGroup / 4], RY[QL][kGroup / 4], RZ[QL][kGroup / 4];
            int4 RP[QL][kGroup / 4];
#pragma unroll
            for (int k = 0; k < QL; ++k) {
                const int bp = bpos[k] >= 0 ? bpos[k] : 0;
#pragma unroll
                for (int c = 0; c < kGroup / 4; ++c) {
                    RX[k][c] = *reinterpret_cast<const float4*>(mp.sx + bp + 4 * c);
                    RY[k][c] = *reinterpret_cast<const float4*>(mp.sy + bp + 4 * c);
                    RZ[k][c] = *reinterpret_cast<const float4*>(mp.sz + bp + 4 * c);
                    RP[k][c] = *reinterpret_cast<const int4*>(mp.perm + bp + 4 * c);
                }
            }
            if (!early_pop) next_raw_v = wq.pop();  // behind the loads above; the resolution below covers most of its latency
#pragma unroll
            for (int k = 0; k < QL; ++k) {
                unsigned int bo = 0xffffffffu;
                int pos = -1;
#pragma unroll
                for (int c = 0; c < kGroup / 4; ++c) {
                    const float xs[4] = {RX[k][c].x, RX[k][c].y, RX[k][c].z, RX[k][c].w};
                    const float ys[4] = {RY[k][c].x, RY[k][c].y, RY[k][c].z, RY[k][c].w};
                    const float zs[4] = {RZ[k][c].x, RZ[k][c].y, RZ[k][c].z, RZ[k][c].w};
                    const int ps[4] = {RP[k][c].x, RP[k][c].y, RP[k][c].z, RP[k][c].w};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float du = dist2(qx[k], qy[k], qz[k], xs[u], ys[u], zs[u]);
                        const bool take = du == best[k] && (unsigned int)ps[u] < bo;
                        bo = take ? (unsigned int)ps[u] : bo;
                        pos = take ? bpos[k] + 4 * c + u : pos;
                        // (bit-select: three plain selects on one condition were turned into a scratch-array lookup)
                        const unsigned int tm = take ? 0xffffffffu : 0u;
                        wx[k] = __uint_as_float((__float_as_uint(xs[u]) & tm) | (__float_as_uint(wx[k]) & ~tm));
                        wy[k] = __uint_as_float((__float_as_uint(ys[u]) & tm) | (__float_as_uint(wy[k]) & ~tm));
                        wz[k] = __uint_as_float((__float_as_uint(zs[u]) & tm) | (__float_as_uint(wz[k]) & ~tm));
                    }
                }
                if (bpos[k] >= 0) {
                    rd[k] = best[k]; rpos[k] = pos; roi[k] = (int)bo;
                    if (pos < 0) tie[k] = 2;  // cannot happen (same arithmetic); be safe: exact pass
                }
            }
        }
#pragma unroll
        for (int k = 0; k < QL; ++k) any_tie |= qi[k] < N && tie[k] >= 2;
        const bool redo = !EX && __any(any_tie);  // (wave-uniform) an exact distance tie between two groups: nothing is written
        if (!redo) {
#pragma unroll
            for (int k = 0; k < QL; ++k) {
                if (qi[k] < N) {  // coalesced: the pairing stays in sorted query order
                    pos_s[qi[k]] = rpos[k];
                    idx_s[qi[k]] = rpos[k] >= 0 ? roi[k] : -1;
                    d2_s[qi[k]] = rd[k];
                    gs_x[qi[k]] = wx[k]; gs_y[qi[k]] = wy[k]; gs_z[qi[k]] = wz[k];
                }
            }
            if constexpr (QPL == 1 && QL == 1) {
                if (item_rows)   // (wave-uniform)
                    item_row_mfma(&sm[0][0], lane, qi[0] < N && rpos[0] >= 0, al0, al1, al2, wx[0], wy[0], wz[0], rd[0],
                                  item_rows + (size_t)item * kNAcc);
            }
        }
        if (lane == 0) {
            if (!EX && !redo) {
                // (a deterministic proxy -- staged points -- orders no better than the measured cycles; without any
                // order the kernel is 6 % slower)
                const unsigned long long c = __builtin_amdgcn_s_memtime() - t_item0;
                if (item_cost) {
                    const unsigned int cc = c > 0xffffffffull ? 0xffffffffu : (unsigned int)c;
                    if (QPL == 2 && QL == 2) { item_cost[2 * item] = cc; item_cost[2 * item + 1] = 0u; }  // a whole 128-query item
                    else item_cost[item] = cc;   // a 64-query item / half: slot 2 * (its 128-query item) + half
                }
            }
            wave_staged += n_staged * QL;  // executed work in units of 64 (query, point) pairs (one atomic per wave, at exit)
            if (dbg_stats) {
                const unsigned long long t_end = __builtin_amdgcn_s_memtime();
                atomicAdd(&dbg_stats[2], n_staged); atomicAdd(&dbg_stats[3], 1ull); atomicMax(&dbg_stats[4], n_staged);
                atomicAdd(&dbg_stats[5], t_sweep0 - t_item0);                         // prologue
                atomicAdd(&dbg_stats[6], (t_sweep1 - t_sweep0) - p_stage - p_visit);  // box scan
                atomicAdd(&dbg_stats[7], p_stage);                                    // staging
                atomicAdd(&dbg_stats[8], p_visit);                                    // distance passes
                atomicAdd(&dbg_stats[10], t_end - t_sweep1);                          // epilogue
                atomicMax(&dbg_stats[9], t_end - t_item0);
                atomicAdd(&dbg_stats[11], (unsigned long long)p_supers); atomicAdd(&dbg_stats[12], (unsigned long long)p_entered);
                atomicAdd(&dbg_stats[13], (unsigned long long)p_tiles);
                atomicAdd(&dbg_stats[14], p_boxwait); atomicAdd(&dbg_stats[15], p_tiletest);
                unsigned long long* rec = dbg_stats + 16 + 8 * (size_t)item;  // per-item record
                rec[0] = t_end - t_item0; rec[1] = n_staged; rec[2] = p_entered; rec[3] = p_tiles;
                rec[4] = t_sweep0 - t_item0; rec[5] = (t_sweep1 - t_sweep0) - p_stage - p_visit; rec[6] = p_stage + p_visit;
                rec[7] = t_end - t_sweep1;
            }
        }
        if (wave_times && wave_items == 1u && !EX) {  // phases of the wave's first item (shader clock)
            const unsigned long long t_end1 = __builtin_amdgcn_s_memtime();
            ph0 = t_sweep0 - t_item0; ph1 = t_sweep1 - t_sweep0; ph2 = t_end1 - t_sweep1; ph3 = n_staged;
        }
        if (!EX && !early_pop) next_item_v = lookup(wq.settle(__builtin_amdgcn_readfirstlane(next_raw_v)));
        const int next_code = __builtin_amdgcn_readfirstlane(next_item_v);
        if constexpr (!EX) {
            if (redo) return self(self, ql_tag, std::true_type{}, item, code, next_code);
        }
        return next_code;
    };
    int code = __builtin_amdgcn_readfirstlane(lookup(wq.first()));
    while (code >= 0) {
        if constexpr (QPL == 2) {
            if (!(code & kHalfFlag)) {
                code = run_item(run_item, std::integral_constant<int, 2>{}, std::false_type{}, code, code, 0);
                continue;
            }
        }
        code = run_item(run_item, std::integral_constant<int, 1>{}, std::false_type{}, code & ~kHalfFlag, code, 0);
    }
    // (statistics, only when the caller profiles: slotted -- 3072 end-of-wave atomics on ONE address arrive at about the
    //  rate the memory side retires them, ~13 ns each, and the launch cannot end before the last one has landed)
    if (lane == 0 && wave_staged && staged_total)
        atomicAdd(staged_total + (size_t)(blockIdx.x & (kStatSlots - 1)) * kStatStride, wave_staged);
    if (wave_times && lane == 0) {  // [start, end, items, first item: prologue, sweep, epilogue cycles, staged points] per wave
        unsigned long long* w = wave_times + 8 * (size_t)wq.global_wave();
        w[0] = t_wave0; w[1] = wall_clock64(); w[2] = wave_items | ((unsigned long long)last_code << 32); w[3] = ph0; w[4] = ph1; w[5] = ph2; w[6] = ph3;
        w[7] = last_start;  // (the wave's last entry: its code above, when it began here)
    }
}

}  // namespace mola_icp_amd
