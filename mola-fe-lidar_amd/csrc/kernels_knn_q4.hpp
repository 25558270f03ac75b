// kernels_knn_q4.hpp -- k_knn_q4: the point-to-plane matcher (mp2p_icp::Matcher_Point2Plane, params/icp-settings-regular.yaml:33-39) with
// LPQ = four, two or one lane(s) per query -- every launch of the plane matcher except the far launches of odometry-size clouds
// (k_knn_coop) and lists longer than ten entries.
// Device code of the ICP core for gfx950; compiled by knn_q4_launch.hip alone.  Numeric contract: hip_backend.hip / DESIGN.md.
//
// k_nn_q4 (kernels_q4.hpp) showed what shortens a launch that is as long as one item's chain: a smaller item.  The same shape for the
// neighbour LISTS of the plane matcher, at LPQ = 4:
//   * an item = 16 consecutive sorted queries on ONE wave, lane 4 q + s = query q, sub-lane s; box tests four boxes per instruction
//     group; a listed tile goes from global memory straight into the wave's LDS ring -- here with its row of ORIGINAL INDICES (the
//     tie-break half of a list key): 32 lanes x 16 bytes per tile, so ONE global_load_lds_dwordx4 carries two whole tiles;
//   * sub-lane s keeps a sorted list of its own -- seeded like the three others with the query's stored list -- over points 8 s ..
//     8 s + 7 of every listed tile.  Its bound, the last key of ITS list, is never below the final one (a list over fewer points),
//     and the query's live bound for the box tests is the smallest of the four: a tile nobody lists holds nothing that belongs in
//     the merged list (k_knn_coop's argument, with sub-lanes in the place of its waves);
//   * the four lists of a query close in two symmetric DPP steps (every sub-lane ends with the merged list);
//   * ONE workgroup = four waves = 64 queries: a wave leaves its 16 lists in its own ring, and whichever wave finishes LAST (an LDS
//     ticket, no barrier) runs the plane epilogue for all 64 -- one query per lane, the fp64 covariance and eigen-solve on full waves
//     exactly as in k_knn_coop (plane_epilogue: same lists -> same planes, seeds, cached planes, certificates).
// A launch costs ~20 us + 0.45 us per 1 000 queries at four lanes: beyond the kernel's wave slots (~80k queries) fewer, longer waves win --
//   * LPQ = 2: 32 queries per wave, two waves per workgroup; sub-lane s evaluates points 16 s .. 16 s + 15 of a tile in two groups of
//     eight, box tests two per group, one merge step;
//   * LPQ = 1: a wave = a whole row of 64 queries = the workgroup; a test's candidate box reaches the lanes as scalars, nothing to
//     merge, no records, no ticket -- the wave runs plane_epilogue from its registers.  k_knn_planes' item without the persistent
//     kernel's machinery, and faster than it at every size measured (profiles/r06/k_knn_q4_development.txt).
// Which one a launch gets: hip_backend.hip, knn_q4_lanes_per_query.
// Certified lists (KnnCert) as in k_knn_coop; a wave whose queries are all certified skips the sweep (k_knn_coop decides that per
// 64).  Results are identical to k_knn_coop / k_knn_planes: the lists are THE K nearest by (d2, original index), whatever finds them.
#pragma once
#include "kernels_planes.hpp"
#include "kernels_q4.hpp"

namespace mola_icp_amd {

constexpr int kKq4TileFloats = 4 * kTileG;                  // x[32] y[32] z[32] original index[32]
constexpr int kKq4BankFloats = kQ4Bank * kKq4TileFloats;    // 512
constexpr int kKq4RingFloats = 2 * kKq4BankFloats;          // two banks per wave: 4 KB
#ifndef MOLA_KQ4_WG_PER_CU
#define MOLA_KQ4_WG_PER_CU 5
#endif
constexpr int kKq4WorkgroupsPerCu = MOLA_KQ4_WG_PER_CU;
// -DMOLA_KQ4_DIAG: per-wave clock stamps (knn_q4_launch.hip prints their medians after every launch; never in the product build)
#ifdef MOLA_KQ4_DIAG
__device__ unsigned long long* g_kq4_dbg = nullptr;
#define KQ4_STAMP(k) do { if (dbg_w) dbg_w[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define KQ4_STAMP(k) do { } while (0)
#endif
// (a wave-uniform 64-bit mask the compiler may have chosen to keep in vector registers -- it does with the scan's state here -- named
//  as the scalar it is: free where the value already lives in scalar registers)
__device__ __forceinline__ unsigned long long uniform64(unsigned long long m)
{
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(m & 0xffffffffull));
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(m >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ int kq4_ff1(unsigned long long m) { return sff1_b64(uniform64(m)); }
__device__ __forceinline__ void kq4_bitset0(unsigned long long& m, int bit) { m = uniform64(m); sbitset0_b64(m, bit); }
// a wave's records for the epilogue (LPQ > 1): row r of query q at word QW * wave + QW * r + q of its ring, QW = its queries (the skew by the wave
// keeps the 64 lanes of the epilogue on 64 different banks); rows: key low / key high / position per entry, the moved query, the certificate
template <int K> constexpr int kq4_rows() { return 3 * K + 4; }
// workgroups per CU at LPQ lanes per query (a workgroup = LPQ waves = 64 queries): the same waves per SIMD whatever LPQ
template <int LPQ> constexpr int kq4_wg_per_cu() { return kKq4WorkgroupsPerCu * (4 / LPQ); }   // the same waves per SIMD either way
// min / max over the wave's queries (every sub-lane of a query holds the same value) as a scalar
template <int LPQ> __device__ __forceinline__ float wave_min_qn(float v)
{
    if constexpr (LPQ == 1) v = fminf(v, dpp_f<kDppRor1>(v));
    if constexpr (LPQ <= 2) v = fminf(v, dpp_f<kDppRor2>(v));
    return wave_min_q(v);
}
template <int LPQ> __device__ __forceinline__ float wave_max_qn(float v)
{
    if constexpr (LPQ == 1) v = fmaxf(v, dpp_f<kDppRor1>(v));
    if constexpr (LPQ <= 2) v = fmaxf(v, dpp_f<kDppRor2>(v));
    return wave_max_q(v);
}

template <int K /*list length: knn + 1*/, int KMAX /*problems per launch: blockIdx.y*/, int LPQ = 4 /*lanes per query*/>
__global__ __launch_bounds__(64 * LPQ, kKq4WorkgroupsPerCu /*HIP: waves per SIMD -- with four-wave workgroups also workgroups per CU*/) void k_knn_q4(const KnnBatch<KMAX> batch, float thr2, float thr2x, double threshold, double plane_eig_thr,
                                                                     unsigned long long* __restrict__ staged_total /*slotted, may be null*/, int lds_boxes,
                                                                     unsigned long long* __restrict__ cert_stats /*diagnostics, may be null*/)
{
    constexpr int QW = 64 / LPQ;   // queries per wave
    constexpr int NW = LPQ;        // waves per workgroup (64 queries)
    static_assert(LPQ == 4 || LPQ == 2 || LPQ == 1, "four, two or one lane(s) per query");
    static_assert(LPQ == 1 || QW * (NW - 1) + QW * kq4_rows<K>() <= kKq4RingFloats, "the records of a wave must fit its ring");   // (one lane per query: no records, the wave runs its own epilogue)
    __shared__ __attribute__((aligned(16))) float s_ring[NW][kKq4RingFloats];
    __shared__ int s_done;
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const KnnProblem& pb = batch.p[KMAX == 1 ? 0 : blockIdx.y];
    const int N = pb.N;
    const int item = lds_boxes ? (int)blockIdx.x : xcd_item((int)blockIdx.x, (N + 63) / 64);   // (see k_knn_coop)
    if (item * 64 >= N) return;   // (uniform: before any barrier)
    const TiledMap mp = pb.mp;
    float* ring = &s_ring[wave][0];
    if (lane == 0) ring[0] = 0.f;   // (a plain store to the ring: the tiles arrive by LDS-DMA, which the compiler does not count as one)
    if (threadIdx.x == 0) s_done = 0;
#ifdef MOLA_KQ4_DIAG
    unsigned long long* dbg_w = g_kq4_dbg && blockIdx.y == 0 && blockIdx.x < 2048 && lane == 0 ? g_kq4_dbg + 16 * (size_t)(blockIdx.x * NW + wave) : nullptr;
#endif
    KQ4_STAMP(0);

    // ---- round trip A: the lane's query, its stored list (position, coordinates, original index per entry) and its certificate ----
    const int s = lane & (LPQ - 1), q = lane / LPQ;
    const int qi = item * 64 + wave * QW + q;
    const bool valid = qi < N;
    const int ic = valid ? qi : N - 1;
    const unsigned int ic4 = (unsigned int)ic * 4u;
    const float lx = ld_at<float>(pb.slx, ic4), ly = ld_at<float>(pb.sly, ic4), lz = ld_at<float>(pb.slz, ic4);
    const int use_seed = pb.use_seed;
    const bool cert_on = pb.cert_on != 0 && use_seed != 0;
    int js[K];
    float gx[K], gy[K], gz[K];
    unsigned int go[K];
    float lb_old = 0.f;
    if (use_seed) {
        const unsigned int rowb = (unsigned int)pb.seeds.stride * 4u;   // bytes between the entries' rows (the q4 form serves <= 2^17 queries)
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const unsigned int at = (unsigned int)j * rowb + ic4;
            js[j] = ld_at<int>(pb.seeds.pos, at);
            gx[j] = ld_at<float>(pb.seeds.x, at); gy[j] = ld_at<float>(pb.seeds.y, at); gz[j] = ld_at<float>(pb.seeds.z, at);
            go[j] = ld_at<unsigned int>(pb.seeds.oidx, at);
        }
        if (cert_on) lb_old = ld_at<float>(pb.lb, ic4);
    }
    if (lds_boxes) load_boxes_to_lds(mp, (lds_f32*)s_dyn);   // (ends with a barrier)
    else __syncthreads();                                    // (s_done's zero is in place before any wave's ticket)
    const lds_f32* lbox = (const lds_f32*)s_dyn;
    const bool use_lbox = lds_boxes != 0;
    KQ4_STAMP(1);

    float qx, qy, qz;
    xform(pb.P, lx, ly, lz, qx, qy, qz);

    // the sub-lane's list: sorted ascending by the packed key (d2 bits << 32 | original index)
    unsigned long long kk[K];
    int kp[K];
    auto kd_of = [&](int j) -> float { return __uint_as_float((unsigned int)(kk[j] >> 32)); };
    auto insert = [&](float du, unsigned int o, int pos) {
        kk[K - 1] = ((unsigned long long)__float_as_uint(du) << 32) | o; kp[K - 1] = pos;
#pragma unroll
        for (int j = K - 1; j > 0; --j) {
            const bool sw = kk[j] < kk[j - 1];
            if (!__any(sw)) break;
            const unsigned long long tk = kk[j]; const int tp = kp[j];
            kk[j] = sw ? kk[j - 1] : tk; kp[j] = sw ? kp[j - 1] : tp;
            kk[j - 1] = sw ? tk : kk[j - 1]; kp[j - 1] = sw ? tp : kp[j - 1];
        }
    };
#pragma unroll
    for (int j = 0; j < K; ++j) { kk[j] = (unsigned long long)__float_as_uint(thr2x) << 32; kp[j] = -1; }   // (gate'^2, 0): never beaten by d2 >= gate'^2
    int tq0 = -1, tq1 = -1;   // tiles of the nearest and of the K-th stored neighbour (sent for ahead of the scan, below)
    if (use_seed) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const float du = dist2(qx, qy, qz, gx[j], gy[j], gz[j]);
            if (js[j] >= 0 && du < thr2x) insert(du, go[j], js[j]);   // (all four sub-lanes alike: distinct positions, no duplicates)
        }
        tq0 = js[0] >= 0 ? js[0] >> 5 : -1;
        tq1 = js[K - 2] >= 0 ? js[K - 2] >> 5 : -1;
    }
    // certified lists (KnnCert; kernels_planes.hpp): the four sub-lanes of a query derive the same verdict from the same data
    bool certd = false;
    float lbw = -1.0f;   // the certificate's new bound; -1: not certified
    if (cert_on) {
        float ox, oy, oz;
        xform(pb.Pprev, lx, ly, lz, ox, oy, oz);
        const float delta = sqrtf(dist2(qx, qy, qz, ox, oy, oz)) * kCertUp + 1e-18f;
        const float m = fminf((lb_old - delta) * kCertDown, sqrtf(thr2x) * kCertDown);
        const float mm = m * m * kCertDown;
        certd = valid && m > 0.f && mm > fminf(kd_of(K - 2), thr2);
        if (certd) lbw = m;
    }
    const bool open = valid && !certd;   // the query takes part in the sweep
    const unsigned long long cert_mask = __ballot(certd);
    const bool skip_sweep = !__any(open);
    if (!valid) qx = qy = qz = 1.0e18f;   // padding lane: no staged point comes near it
    unsigned int tiles = 0u;
#ifdef MOLA_KQ4_DIAG
    unsigned int dg_slow = 0u, dg_key = 0u, dg_dup = 0u, dg_ins = 0u, dg_tests = 0u;
#endif
    KQ4_STAMP(2);
    bool inserted = false;   // this sub-lane's list is no longer the seeded one

    if (!skip_sweep) {
        float kb = open ? kd_of(K - 1) : -1.0f;   // the sub-lane's own bound (-1: reaches nothing)
        // ---- the lists: super-tiles some query reaches, then the tiles; entry n of a list sits in lane n of one register ----
        int sl = -1, n_sl = 0, tl = -1, n_tl = 0;
        const unsigned int arr_stride = (unsigned int)(mp.sy - mp.sx);
        const unsigned int trow = (unsigned int)mp.n_tiles_p * 4u;
        // tiles [first, first + 4) of the list (entries of -1: none) -> ring bank `bank`: one instruction per two tiles (32 lanes x 16 bytes
        // each: chunk c of a tile = words 4 (c & 7) .. of row c >> 3; row 3 = the original indices)
        const int lane_c = lane & 31, lane_e = lane >> 5;
        const float* lane_src = (lane_c >> 3) == 3 ? reinterpret_cast<const float*>(mp.perm) + (lane_c & 7) * 4
                                                   : mp.sx + (size_t)(lane_c >> 3) * arr_stride + (lane_c & 7) * 4;
        auto issue_bank = [&](int first, int bank) -> int {
            int issued = 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (first + 2 * h < n_tl) {   // (wave-uniform)
                    const int e = first + 2 * h + lane_e;
                    const int t = __builtin_amdgcn_ds_bpermute((e & 63) << 2, tl);
                    if (e < n_tl && t >= 0)
                        __builtin_amdgcn_global_load_lds(lane_src + (size_t)t * kTileG, ring + bank * kKq4BankFloats + h * 2 * kKq4TileFloats, 16, 0, 0);
                    ++issued;
                }
            }
            return issued;
        };
        // ---- the stored neighbours' tiles hold points at the query's bound: they will be listed whatever the scan finds, so up to
        // four of them are sent for FIRST (k_nn_q4), with the tile boxes of the first one's super-tile ----
        int n_pre = 0, pre0 = -1, pre1 = -1, pre2 = -1, pre3 = -1, S_spec = -1;
        float f0 = 0.f, f1 = 0.f, f2 = 0.f, f3 = 0.f, f4 = 0.f, f5 = 0.f;
        if (use_seed) {   // (wave-uniform)
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int tq = open ? (pass ? tq1 : tq0) : -1;
                unsigned long long rem = __ballot(tq >= 0 && tq != pre0 && tq != pre1 && tq != pre2 && tq != pre3) & (LPQ == 4 ? 0x1111111111111111ull : (LPQ == 2 ? 0x5555555555555555ull : ~0ull));   // one lane per query
                while (rem && n_tl < kQ4Bank) {
                    const int t = __builtin_amdgcn_readlane(tq, kq4_ff1(rem));
                    rem &= ~__ballot(tq == t);
                    tl = lane == n_tl ? t : tl;
                    pre3 = n_tl == 3 ? t : pre3; pre2 = n_tl == 2 ? t : pre2; pre1 = n_tl == 1 ? t : pre1; pre0 = n_tl == 0 ? t : pre0;
                    ++n_tl;
                }
            }
            n_pre = n_tl;
            if (n_pre) {
                (void)issue_bank(0, 0);
                n_tl = kQ4Bank;   // (the stored neighbours' tiles own bank 0 -- its unused entries stay -1)
                S_spec = pre0 >> 6;
                const unsigned int ti = (unsigned int)(S_spec * kSuper + lane) * 4u;
                f0 = ld_at<float>(mp.tbox, ti); f1 = ld_at<float>(mp.tbox, ti + trow); f2 = ld_at<float>(mp.tbox, ti + 2u * trow);
                f3 = ld_at<float>(mp.tbox, ti + 3u * trow); f4 = ld_at<float>(mp.tbox, ti + 4u * trow); f5 = ld_at<float>(mp.tbox, ti + 5u * trow);
            }
        }
        bool bank0_in_flight = n_pre > 0;

        // ---- the wave box: union of the open queries' boxes [q - r, q + r], in scalar registers (k_nn_q4) ----
        float reach = __builtin_amdgcn_sqrtf(kb * 1.000002f) * 1.00001f + fmaxf(fabsf(qx), fmaxf(fabsf(qy), fabsf(qz))) * 2.4e-7f + 1e-30f;
        if (!open) reach = -1.0f;
        Box w;
        w.lo[0] = wave_min_qn<LPQ>(reach >= 0.f ? qx - reach : INFINITY); w.hi[0] = wave_max_qn<LPQ>(reach >= 0.f ? qx + reach : -INFINITY);
        w.lo[1] = wave_min_qn<LPQ>(reach >= 0.f ? qy - reach : INFINITY); w.hi[1] = wave_max_qn<LPQ>(reach >= 0.f ? qy + reach : -INFINITY);
        w.lo[2] = wave_min_qn<LPQ>(reach >= 0.f ? qz - reach : INFINITY); w.hi[2] = wave_max_qn<LPQ>(reach >= 0.f ? qz + reach : -INFINITY);

        // the query's live bound: the smallest of its four sub-lanes' (all >= 0, or all -1: the bit patterns order as integers)
        auto live_bound = [&]() -> float {
            int v = __float_as_int(kb);
            if constexpr (LPQ >= 2) v = min(v, dpp_i<kDppXor1>(v));
            if constexpr (LPQ == 4) v = min(v, dpp_i<kDppXor2>(v));
            return __int_as_float(v);
        };
        // one group of box tests: the next (up to) LPQ candidates of `cand`, sub-lane s of every query takes candidate s (k_nn_q4)
        const int s8 = 8 * s;
        auto test4 = [&](float r0, float r1, float r2, float r3, float r4, float r5, unsigned long long& cand, float bound, auto&& on_pass) {
#ifdef MOLA_KQ4_DIAG
            ++dg_tests;
#endif
            const int c0 = kq4_ff1(cand); kq4_bitset0(cand, c0);
            if constexpr (LPQ == 1) {   // one candidate, the same for every query: its box comes as scalars (callers pass a non-empty mask)
                const float m0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r0), c0)), m1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r1), c0));
                const float m2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r2), c0)), m3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r3), c0));
                const float m4 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r4), c0)), m5 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r5), c0));
                const float ax = qx - __builtin_amdgcn_fmed3f(qx, m0, m3);
                const float ay = qy - __builtin_amdgcn_fmed3f(qy, m1, m4);
                const float az = qz - __builtin_amdgcn_fmed3f(qz, m2, m5);
                if (__any(fmaf(az, az, fmaf(ay, ay, ax * ax)) <= bound)) on_pass(c0);
                return;
            }
            const int c1 = kq4_ff1(cand); kq4_bitset0(cand, c1);
            int c2 = -1, c3 = -1;
            if constexpr (LPQ == 4) {
                c2 = kq4_ff1(cand); kq4_bitset0(cand, c2);
                c3 = kq4_ff1(cand); kq4_bitset0(cand, c3);
            }
            const unsigned int pack = LPQ == 4 ? (((unsigned int)c0 & 0xffu) | (((unsigned int)c1 & 0xffu) << 8) | (((unsigned int)c2 & 0xffu) << 16) | ((unsigned int)c3 << 24))
                                               : (((unsigned int)c0 & 0xffu) | (((unsigned int)c1 & 0xffu) << 8));
            const unsigned int sel = __builtin_amdgcn_ubfe(pack, (unsigned int)s8, 8u);
            const int src = (int)(sel << 2);
            const float m0 = bperm_f(src, r0), m1 = bperm_f(src, r1), m2 = bperm_f(src, r2);
            const float m3 = bperm_f(src, r3), m4 = bperm_f(src, r4), m5 = bperm_f(src, r5);
            const float ax = qx - __builtin_amdgcn_fmed3f(qx, m0, m3);
            const float ay = qy - __builtin_amdgcn_fmed3f(qy, m1, m4);
            const float az = qz - __builtin_amdgcn_fmed3f(qz, m2, m5);
            const unsigned long long m = __ballot(sel != 0xffu && fmaf(az, az, fmaf(ay, ay, ax * ax)) <= bound);
            unsigned int P = (unsigned int)m | (unsigned int)(m >> 32);   // candidate j passed iff a lane with sub-lane j is set
            P |= P >> 16; P |= P >> 8; P |= P >> 4;
            if constexpr (LPQ == 2) P |= P >> 2;
            if (P & 1u) on_pass(c0);
            if (P & 2u) on_pass(c1);
            if constexpr (LPQ == 4) {
                if (P & 4u) on_pass(c2);
                if (P & 8u) on_pass(c3);
            }
        };
        // the lane's 32 / LPQ points of one tile (LDS: x[32] y[32] z[32] index[32]), eight at a time, against its query's list
        auto eval_tile = [&](int t, const float* b) {
#pragma unroll
          for (int h = 0; h < 4 / LPQ; ++h) {
            const int g8 = 8 * ((4 / LPQ) * s + h);
            const float* p = b + g8;
            const int gpos = t * kTileG + g8;
            const float4 X0 = *reinterpret_cast<const float4*>(p), X1 = *reinterpret_cast<const float4*>(p + 4);
            const float4 Y0 = *reinterpret_cast<const float4*>(p + 32), Y1 = *reinterpret_cast<const float4*>(p + 36);
            const float4 Z0 = *reinterpret_cast<const float4*>(p + 64), Z1 = *reinterpret_cast<const float4*>(p + 68);
            const float xs[8] = {X0.x, X0.y, X0.z, X0.w, X1.x, X1.y, X1.z, X1.w};
            const float ys[8] = {Y0.x, Y0.y, Y0.z, Y0.w, Y1.x, Y1.y, Y1.z, Y1.w};
            const float zs[8] = {Z0.x, Z0.y, Z0.z, Z0.w, Z1.x, Z1.y, Z1.z, Z1.w};
            float d[8];
            float gm = INFINITY;
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                const v2f mx = {xs[u], xs[u + 1]}, my = {ys[u], ys[u + 1]}, mz = {zs[u], zs[u + 1]};
                const v2f dd = dist2_pk2(qx, qy, qz, mx, my, mz);
                d[u] = dd.x; d[u + 1] = dd.y;
                gm = fminf(fminf(gm, dd.x), dd.y);
            }
            // (a certified query's first K entries are proven and its bound comes from the certificate: it takes no part -- in a late launch nine
            //  lanes in ten are certified, and every tile near them holds their own list members, at distances inside their lists)
            if (__any(open && gm <= kd_of(K - 1))) {   // some lane may have to insert: rare once the lists have tightened
#ifdef MOLA_KQ4_DIAG
                ++dg_slow;
#endif
                const uint4 O0 = *reinterpret_cast<const uint4*>(p + 96), O1 = *reinterpret_cast<const uint4*>(p + 100);
                const unsigned int os[8] = {O0.x, O0.y, O0.z, O0.w, O1.x, O1.y, O1.z, O1.w};
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (open && (((unsigned long long)__float_as_uint(d[u]) << 32) | os[u]) < kk[K - 1]) {
                        bool dup = false;   // a stored neighbour met again by the sweep
#pragma unroll
                        for (int j = 0; j < K; ++j) dup |= kp[j] == gpos + u;
#ifdef MOLA_KQ4_DIAG
                        dg_key += (unsigned int)__popcll(__ballot(true)); dg_dup += (unsigned int)__popcll(__ballot(dup)); dg_ins += (unsigned int)__popcll(__ballot(!dup));
#endif
                        if (!dup) { insert(d[u], os[u], gpos + u); inserted = true; }
                    }
                }
                kb = open ? kd_of(K - 1) : -1.0f;   // live bound (see k_knn_planes)
            }
          }
        };
        auto run_tiles = [&]() {
            if (n_tl == 0) return;
            if (!bank0_in_flight) (void)issue_bank(0, 0);
            bank0_in_flight = false;
            for (int c = 0; c * kQ4Bank < n_tl; ++c) {
                const int nxt = (c + 1) * kQ4Bank < n_tl ? issue_bank((c + 1) * kQ4Bank, (c + 1) & 1) : 0;
                if (nxt == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else if (nxt == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                const float* bank = ring + (c & 1) * kKq4BankFloats;
#pragma unroll 1
                for (int k = 0; k < kQ4Bank; ++k) {
                    const int e = c * kQ4Bank + k;
                    if (e >= n_tl) break;
                    const int t = __builtin_amdgcn_readlane(tl, e);
                    if (t >= 0) { eval_tile(t, bank + k * kKq4TileFloats); ++tiles; }
                }
                __builtin_amdgcn_wave_barrier();
            }
            n_tl = 0;
        };

        // ---- the scan (k_nn_q4's resumable state machine: run_tiles has one call site) ----
        const lds_f32* l_ubox = lbox;
        const lds_f32* l_sbox = lbox + 6 * mp.n_top;
        int ub = 0, sb = 0;
        unsigned long long ucand = 0ull, scand = 0ull;
        float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f, c4 = 0.f, c5 = 0.f;
        bool c_valid = false, scan_done = false;
        if (mp.n_top == 1) { ucand = 1ull; ub = 64; }
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
            {
                const float bound = live_bound();
                while (!scan_done && n_sl <= kQ4ListCap - 4) {
                    if (scand) {
                        if (!c_valid) load_super_boxes();
                        test4(c0, c1, c2, c3, c4, c5, scand, bound, [&](int i) { sl = lane == n_sl ? sb + i : sl; ++n_sl; });
                    } else if (ucand) {
                        sb = (ub - 64 + kq4_ff1(ucand)) * 64;
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
            if (n_sl || n_tl) {
                c_valid = false;
                int e_sl = 0, S = __builtin_amdgcn_readlane(sl, 0), Sc = 0;
                if (n_sl && S != S_spec) {
                    const unsigned int ti = (unsigned int)(S * kSuper + lane) * 4u;
                    f0 = ld_at<float>(mp.tbox, ti); f1 = ld_at<float>(mp.tbox, ti + trow); f2 = ld_at<float>(mp.tbox, ti + 2u * trow);
                    f3 = ld_at<float>(mp.tbox, ti + 3u * trow); f4 = ld_at<float>(mp.tbox, ti + 4u * trow); f5 = ld_at<float>(mp.tbox, ti + 5u * trow);
                }
                S_spec = -1;
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
                            const int mine = Sc * kSuper + lane;
                            const bool listed = mine == pre0 || mine == pre1 || mine == pre2 || mine == pre3;   // (sent for already)
                            cand = __ballot(!listed && b0 <= w.hi[0] && b1 <= w.hi[1] && b2 <= w.hi[2] && b3 >= w.lo[0] && b4 >= w.lo[1] && b5 >= w.lo[2]);
                        }
                    }
                    run_tiles();
                    bound = live_bound();
                } while (cand || e_sl < n_sl);
                n_sl = 0;
            }
            if (scan_done) break;
        }

        KQ4_STAMP(3);
        // ---- the four lists of a query become one: two symmetric steps, both partners end with the same merged list ----
        if (LPQ > 1 && __any(inserted)) {
#pragma unroll
            for (int st = 0; st < (LPQ == 4 ? 2 : (LPQ == 2 ? 1 : 0)); ++st) {
                unsigned int plo[K], phi[K];
                int pp[K];
#pragma unroll
                for (int j = 0; j < K; ++j) {   // the partner's list as it is NOW (before either side inserts)
                    const unsigned int lo = (unsigned int)(kk[j] & 0xffffffffull), hi = (unsigned int)(kk[j] >> 32);
                    plo[j] = (unsigned int)(st ? dpp_i<kDppXor2>((int)lo) : dpp_i<kDppXor1>((int)lo));
                    phi[j] = (unsigned int)(st ? dpp_i<kDppXor2>((int)hi) : dpp_i<kDppXor1>((int)hi));
                    pp[j] = st ? dpp_i<kDppXor2>(kp[j]) : dpp_i<kDppXor1>(kp[j]);
                }
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const unsigned long long ck = ((unsigned long long)phi[j] << 32) | plo[j];
                    bool take = pp[j] >= 0 && ck < kk[K - 1];
                    if (!__any(take)) break;   // (sorted: nothing further down the partner's list can enter either)
#pragma unroll
                    for (int i = 0; i < K; ++i) take &= kp[i] != pp[j];
                    if (__any(take)) {
                        if (take) insert(__uint_as_float(phi[j]), plo[j], pp[j]);
                    }
                }
            }
        }
    }

    if constexpr (LPQ > 1) {
    // ---- the wave's records for the epilogue (row r by sub-lane r & 3: every sub-lane holds the merged list) ----
    {
        int* rec = reinterpret_cast<int*>(ring) + QW * wave + q;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            if (s == ((3 * j) & (LPQ - 1))) rec[QW * (3 * j)] = (int)(unsigned int)(kk[j] & 0xffffffffull);
            if (s == ((3 * j + 1) & (LPQ - 1))) rec[QW * (3 * j + 1)] = (int)(unsigned int)(kk[j] >> 32);
            if (s == ((3 * j + 2) & (LPQ - 1))) rec[QW * (3 * j + 2)] = kp[j];
        }
        if (s == ((3 * K) & (LPQ - 1))) rec[QW * (3 * K)] = __float_as_int(qx);
        if (s == ((3 * K + 1) & (LPQ - 1))) rec[QW * (3 * K + 1)] = __float_as_int(qy);
        if (s == ((3 * K + 2) & (LPQ - 1))) rec[QW * (3 * K + 2)] = __float_as_int(qz);
        if (s == ((3 * K + 3) & (LPQ - 1))) rec[QW * (3 * K + 3)] = __float_as_int(lbw);
    }
    }
    if (lane == 0 && tiles && staged_total)   // units of 64 (query, point) pairs: a tile = 32 points x QW queries
        atomicAdd(staged_total + (size_t)((blockIdx.x * NW + wave) & (kStatSlots - 1)) * kStatStride, (unsigned long long)tiles * (unsigned long long)(QW / 2));
    if (cert_stats && lane == 0 && cert_mask) {   // [1] certified queries, [2] sweeps skipped -- per 16-query wave here, not per 64-query item
        unsigned long long* st = cert_stats + (size_t)((blockIdx.x * NW + wave) & (kStatSlots - 1)) * kStatStride;
        atomicAdd(st + 1, (unsigned long long)(__popcll(cert_mask) / LPQ));
        if (skip_sweep) atomicAdd(st + 2, 1ull);
    }
    KQ4_STAMP(4);
    if constexpr (LPQ == 1) {   // one lane per query: the wave holds a whole row of 64 and runs its epilogue from registers
        float ed[K];
        unsigned int eo[K];
#pragma unroll
        for (int j = 0; j < K; ++j) { ed[j] = kd_of(j); eo[j] = (unsigned int)(kk[j] & 0xffffffffull); }
        if (valid) pb.lb[qi] = lbw > 0.f ? lbw : sqrtf(ed[K - 1]) * kCertDown;
        const bool changed = plane_epilogue<K>(mp, kp, ed, eo, qx, qy, qz, qi, N, thr2, threshold, plane_eig_thr, pb.out, pb.cache, pb.seeds, use_seed, pb.use_cache);
        if (lane == 0 && changed && pb.changed_items) atomicAdd(pb.changed_items + (size_t)(blockIdx.x & (kQueues - 1)) * kQueueStride, 1u);
        KQ4_STAMP(5);
        return;
    }
#ifdef MOLA_KQ4_DIAG
    if (dbg_w) {
        dbg_w[6] = (unsigned long long)tiles | ((unsigned long long)(skip_sweep ? 1 : 0) << 32) | ((unsigned long long)(__popcll(cert_mask) / LPQ) << 40);
        dbg_w[8] = dg_slow; dbg_w[9] = dg_key; dbg_w[10] = dg_dup; dbg_w[11] = dg_ins; dbg_w[12] = dg_tests;
    }
#endif
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    int ticket = 0;
    if (lane == 0) ticket = atomicAdd(&s_done, 1);
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    if (ticket != NW - 1) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");

    // ---- the LAST wave of the workgroup: the plane epilogue of its 64 queries, one per lane (as k_knn_coop's leader) ----
    {
        const int w2 = lane / QW, q2 = lane % QW;
        const int* rec = reinterpret_cast<const int*>(&s_ring[w2][0]) + QW * w2 + q2;
        int ep[K];
        float ed[K];
        unsigned int eo[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            eo[j] = (unsigned int)rec[QW * (3 * j)];
            ed[j] = __int_as_float(rec[QW * (3 * j + 1)]);
            ep[j] = rec[QW * (3 * j + 2)];
        }
        const float ex = __int_as_float(rec[QW * (3 * K)]), ey = __int_as_float(rec[QW * (3 * K + 1)]), ez = __int_as_float(rec[QW * (3 * K + 2)]);
        const float elb = __int_as_float(rec[QW * (3 * K + 3)]);
        const int qe = item * 64 + lane;
        if (qe < N) pb.lb[qe] = elb > 0.f ? elb : sqrtf(ed[K - 1]) * kCertDown;
        const bool changed = plane_epilogue<K>(mp, ep, ed, eo, ex, ey, ez, qe, N, thr2, threshold, plane_eig_thr, pb.out, pb.cache, pb.seeds, use_seed, pb.use_cache);
        if (lane == 0 && changed && pb.changed_items) atomicAdd(pb.changed_items + (size_t)(blockIdx.x & (kQueues - 1)) * kQueueStride, 1u);
#ifdef MOLA_KQ4_DIAG
        if (dbg_w) dbg_w[7] = changed ? 1ull : 0ull;
#endif
    }
    KQ4_STAMP(5);
}

}  // namespace mola_icp_amd
