// kernels_planes.hpp -- point-to-plane matcher k_knn_planes and its quadratic-form accumulation
// Device code of the ICP core for gfx950; included by hip_backend.hip only (one translation unit: the kernels are
// launched from there).  Numeric contract and data layout: hip_backend.hip / DESIGN.md.
#pragma once
#include "kernels_tiled.hpp"
#include "kernels_coop.hpp"

namespace mola_icp_amd {

// ---- row f3: point-to-plane matcher (mp2p_icp::Matcher_Point2Plane, params/icp-settings-regular.yaml:33-39) ----
// Same tiled sweep; the visitor keeps, per query, the K nearest points as a sorted list ordered by
// (d2, original index).  The reach is the gate (distanceThreshold): only neighbours inside it matter.
// Epilogue per query: the neighbours inside the gate (need >= 3) -> mean + covariance in fp64 -> cyclic
// Jacobi eigen-decomposition -> plane iff e0 <= planeEigenThreshold * e2, normal = eigenvector of e0,
// pairing iff |n.(q - mean)| <= distanceThreshold.  [EXT-recalled mp2p_icp behaviour; restated in the
// CPU checker with the same operation order.]
struct PlanePair {      // one per query, sorted query order  (padding it to 64 bytes / 16-byte loads was measured: no gain)
    double c[3];        // plane centroid
    double n[3];        // unit normal
    int valid, n_neigh;
};

__device__ __forceinline__ void eig_sym3_dev(const double Cin[3][3], double ev[3], double V[3][3])
{
    double A[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) { A[i][j] = Cin[i][j]; V[i][j] = (i == j); }
    for (int sweep = 0; sweep < 32; sweep++) {
        const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        const double dg = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
        if (off == 0 || off < 1e-34 * dg) break;
#pragma unroll
        for (int p = 0; p < 2; p++)
#pragma unroll
            for (int q = p + 1; q < 3; q++) {
                if (A[p][q] == 0) continue;
                const double theta = (A[q][q] - A[p][p]) / (2 * A[p][q]);
                const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                const double c = 1 / sqrt(tt * tt + 1), s2 = tt * c;
#pragma unroll
                for (int k = 0; k < 3; k++) { const double a = A[k][p], b = A[k][q]; A[k][p] = c * a - s2 * b; A[k][q] = s2 * a + c * b; }
#pragma unroll
                for (int k = 0; k < 3; k++) { const double a = A[p][k], b = A[q][k]; A[p][k] = c * a - s2 * b; A[q][k] = s2 * a + c * b; }
#pragma unroll
                for (int k = 0; k < 3; k++) { const double a = V[k][p], b = V[k][q]; V[k][p] = c * a - s2 * b; V[k][q] = s2 * a + c * b; }
            }
    }
    // ascending order (bubble on 3 values, with the matching columns)
    double d[3] = {A[0][0], A[1][1], A[2][2]};
    int o0 = 0, o1 = 1, o2 = 2;
    if (d[o0] > d[o1]) { const int t = o0; o0 = o1; o1 = t; }
    if (d[o1] > d[o2]) { const int t = o1; o1 = o2; o2 = t; }
    if (d[o0] > d[o1]) { const int t = o0; o0 = o1; o1 = t; }
    const int o[3] = {o0, o1, o2};
    double Vs[3][3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        ev[k] = o[k] == 0 ? d[0] : (o[k] == 1 ? d[1] : d[2]);
#pragma unroll
        for (int r = 0; r < 3; r++) Vs[r][k] = o[k] == 0 ? V[r][0] : (o[k] == 1 ? V[r][1] : V[r][2]);
    }
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int k = 0; k < 3; k++) V[r][k] = Vs[r][k];
}

// The stored neighbour lists of a cloud pair -- the next launch's seeds.  Entry-major (entry j of sorted query i at [j * stride + i]:
// a wave's 64 queries read an entry with one coalesced load), and each neighbour's COORDINATES and original index sit beside its
// sorted-map position: a launch's prologue is one round trip of coalesced loads.  (Rounds 2-3 kept positions only, query-major:
// 7 strided loads, then 28 gathered words per query for the coordinates and indices, a dependent second trip -- and the
// cooperative kernel does it in all four waves of an item.  A kernel made of just that prologue and the epilogue took 20-24 us at
// 120k queries: most of what a late launch of the Gauss-Newton loop costs.)
struct KnnSeeds {
    int* pos;            // sorted-map position of the neighbour; -1: none
    float *x, *y, *z;    // its coordinates, as stored in the map
    unsigned int* oidx;  // its original index (the tie-break half of the list's keys)
    size_t stride;       // queries per entry row (>= N)
};
__host__ __device__ inline size_t knn_seeds_bytes(size_t stride, int entries) { return 5u * sizeof(int) * stride * (size_t)entries; }
__host__ __device__ inline KnnSeeds knn_seeds_at(void* base, size_t stride, int entries)
{
    int* b = static_cast<int*>(base);
    const size_t a = stride * (size_t)entries;
    return KnnSeeds{b, reinterpret_cast<float*>(b + a), reinterpret_cast<float*>(b + 2 * a), reinterpret_cast<float*>(b + 3 * a),
                    reinterpret_cast<unsigned int*>(b + 4 * a), stride};
}

// Epilogue of one query of the point-to-plane matcher: kp / kd = its sorted neighbour list of KL = K + 1 entries (sorted-map
// positions, squared distances; -1 / gate^2 in the unused tail), (qx, qy, qz) the moved query.  The plane is that of the
// first K; the extra entry is only a seed -- it is what makes the list's last distance a lower bound on every point
// OUTSIDE the K nearest (certified lists, k_knn_planes).  Writes the seeds (next launch's: KL entries per query, only where they
// differ from what is stored), the plane of the list into `cache` when it had to be solved, the pairing into `out`.  Returns whether SOME lane of the wave had to
// solve a plane (= the item's lists changed).
template <int KL>
__device__ __forceinline__ bool plane_epilogue(const TiledMap& mp, const int (&kp)[KL], const float (&kd)[KL], const unsigned int (&ko)[KL] /*original indices*/,
                                               float qx, float qy, float qz,
                                               int i, int N, float thr2, double threshold, double plane_eig_thr,
                                               PlanePair* __restrict__ out, PlanePair* __restrict__ cache, const KnnSeeds& seeds,
                                               int use_seed, int use_cache)
{
    constexpr int K = KL - 1;
    bool item_changed = false;
    // (the sign of the threshold carries a reading switch of mola_icp_params: negative = a plane needs ALL K neighbours inside the gate,
    //  not just >= 3 -- hip_backend.hip, plane_eig_arg)
    const int min_inside = __builtin_signbit(plane_eig_thr) ? K : 3;
    plane_eig_thr = fabs(plane_eig_thr);
    {
            const bool in = i < N;
            const size_t ic = in ? (size_t)i : (size_t)(N - 1);
            int m = 0;
#pragma unroll
            for (int j = 0; j < K; ++j) m += (kp[j] >= 0 && kd[j] < thr2) ? 1 : 0;  // sorted: the first m entries are inside the gate
            // (the list may go on beyond the gate, up to the extended gate of k_knn_planes: seeds, never part of a plane)
            bool same = use_seed != 0 && use_cache != 0;
            if (same) same = cache[ic].n_neigh == m;
            bool differs = use_seed == 0;   // some stored entry (the extra one included) is not this launch's
            if (use_seed) {
#pragma unroll
                for (int j = 0; j < KL; ++j) {
                    const int old = seeds.pos[(size_t)j * seeds.stride + ic];
                    if (j < K) same &= old == kp[j];
                    differs |= old != kp[j];
                }
            }
            PlanePair pl;  // the plane of the list: valid = "is a plane" (before the query-distance test)
            pl.valid = 0; pl.n_neigh = m;
#pragma unroll
            for (int a = 0; a < 3; ++a) { pl.c[a] = 0; pl.n[a] = 0; }
            const bool solve = in && !same;
            const bool fetch = in && (differs || !same);   // the list's points are needed: to be stored as seeds, or for the plane
            if (__any(fetch)) {
                item_changed = __any(solve);
                float fx[KL], fy[KL], fz[KL];
#pragma unroll
                for (int j = 0; j < KL; ++j) {
                    fx[j] = fy[j] = fz[j] = 0.f;
                    if (fetch && kp[j] >= 0) { fx[j] = mp.sx[kp[j]]; fy[j] = mp.sy[kp[j]]; fz[j] = mp.sz[kp[j]]; }
                }
                if (in && differs) {
#pragma unroll
                    for (int j = 0; j < KL; ++j) {
                        const size_t at = (size_t)j * seeds.stride + ic;
                        seeds.pos[at] = kp[j];
                        seeds.x[at] = fx[j]; seeds.y[at] = fy[j]; seeds.z[at] = fz[j];
                        seeds.oidx[at] = ko[j];
                    }
                }
                if (solve && m >= min_inside) {
                    // (the points stay fp32 in registers and are widened where they are used: exact, and 36 registers fewer
                    //  than three arrays of doubles)
                    double mean[3] = {0, 0, 0};
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        if (j < m) { mean[0] += (double)fx[j]; mean[1] += (double)fy[j]; mean[2] += (double)fz[j]; }
                    }
                    const double dm = (double)m;
                    mean[0] /= dm; mean[1] /= dm; mean[2] /= dm;
                    double Cm[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        if (j < m) {
                            const double dd[3] = {(double)fx[j] - mean[0], (double)fy[j] - mean[1], (double)fz[j] - mean[2]};
#pragma unroll
                            for (int r = 0; r < 3; ++r)
#pragma unroll
                                for (int c = 0; c < 3; ++c) Cm[r][c] += dd[r] * dd[c];
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int c = 0; c < 3; ++c) Cm[r][c] /= dm;
                    double ev[3], V[3][3];
                    eig_sym3_dev(Cm, ev, V);
                    if (!(ev[0] > plane_eig_thr * ev[2])) {
                        pl.valid = 1;
                        pl.c[0] = mean[0]; pl.c[1] = mean[1]; pl.c[2] = mean[2];
                        pl.n[0] = V[0][0]; pl.n[1] = V[1][0]; pl.n[2] = V[2][0];
                    }
                }
                if (solve) cache[ic] = pl;
            }
            if (in && same) pl = cache[ic];
            if (in) {
                PlanePair pp = pl;
                if (pl.valid) {
                    const double dist = fabs(pl.n[0] * ((double)qx - pl.c[0]) + pl.n[1] * ((double)qy - pl.c[1]) +
                                             pl.n[2] * ((double)qz - pl.c[2]));
                    if (dist > threshold) {
                        pp.valid = 0;
#pragma unroll
                        for (int a = 0; a < 3; ++a) { pp.c[a] = 0; pp.n[a] = 0; }
                    }
                }
                out[ic] = pp;
            }
    }
    return item_changed;
}

// QL = queries per lane: 2 (items of 128 queries), or 1 (items of 64) when the cloud has no more 128-query items than the
// launch has waves -- every wave then runs ONE item, the launch is as long as the heaviest of them, and halving the
// items nearly halves it (an odometry-size scan pair: 120k points = 938 items of 128 on 3072 waves).
// CERTIFIED LISTS (temporal coherence, exact).  The point-to-plane loop converges in a handful of iterations and then
// keeps launching the matcher at poses that hardly move; proving that a query's K nearest are still the same K is far
// cheaper than sweeping for them.  The lists hold KL = K + 1 entries.  After a launch, every map point OUTSIDE a query's
// stored list has d2 >= the list's last key (the sweep culls and rejects against exactly that value; gate^2 while the
// list is not full): lb = its square root, rounded down, is a lower bound on the distance from the query (as that launch
// transformed it) to every outside point.  The next launch moves the query by delta = |q - q_prev| (both transforms
// recomputed here, bit for bit what the launches use), so every outside point is still >= lb - delta away (triangle
// inequality).  The seeds are re-evaluated with the contract's arithmetic and sorted as always; if (lb - delta)^2 exceeds
// the K-th seed's d2 -- margins for every rounding below -- no outside point can enter or tie with the first K: they ARE
// the K nearest, in the right order (a swap with the K+1-th seed is inside the list).  Such a lane takes no part in the
// sweep (no reach; it keeps evaluating what the others stage, which can only improve its K+1-th entry), an item whose
// lanes are all certified skips the sweep, and lb - delta is stored as the new bound (a seed that left the gate joins
// the outside at >= gate' >= lb).  THE LISTS' OWN GATE: a list that is not full bounds the outside only by the gate it was
// swept with -- and with the matcher's gate itself there is no margin at all: a sparse region's queries (a spinning lidar's
// far rings: fewer than K + 1 points within 0.7 m) would be swept over the whole gate ball at every launch, and they are
// the heaviest items.  So the lists are kept over a slightly larger gate (gate' = 1.1 gate: thr2x); the plane uses the
// entries inside the true gate among the first K, exactly as before (the list is sorted), and a short list certifies as
// long as lb - delta stays beyond the true gate.  Rounding: the contract's d2 = D (1 + e), |e| <= 6u (u = 2^-24), so distances follow
// from d2 within 3.1u; sqrtf within 2u; the products / differences below within u each; every factor (1 -+ 32u) leaves
// several u to spare.
struct KnnCert {
    PoseF Pprev;  // the pose of the launch that wrote lb / the seeds
    float* lb;    // per query (sorted order): read if `on`, always written
    int on;       // lb and Pprev describe the seeds in knn_pos
    unsigned long long* stats;  // diagnostics (slotted counters, may be null): [1] certified queries, [2] items that skipped the sweep
};
constexpr float kCertUp = 1.0f + 2.0e-6f, kCertDown = 1.0f - 2.0e-6f;  // ~ (1 +- 32u)

// DENSE = true: the same kernel compiled for four workgroups per CU (128 VGPRs, ~20 spilled) -- the SEEDED insertion launches gain
// more from the fourth wave per SIMD than they lose to the spills (C3: 521 / 561 / 303 / 194 us -> 458 / 489 / 257 / 143); the
// unseeded launch, where every lane is inserting, does not (554 -> 610 us) and keeps the spill-free build at three.
template <int K /*list length: knn + 1*/, bool VERIFY, int QL, bool DENSE = false>
__global__ __launch_bounds__(256, (K <= 7 ? ((DENSE || VERIFY) ? 4 : 3) : 2)) void k_knn_planes(const float* __restrict__ slx, const float* __restrict__ sly,
                                                    const float* __restrict__ slz, int N, TiledMap mp, PoseF P,
                                                    float thr2, float thr2x /*the lists' own gate^2 >= thr2 (see below)*/, double threshold, double plane_eig_thr,
                                                    PlanePair* __restrict__ out, PlanePair* __restrict__ cache /*the plane of each query's list*/,
                                                    KnnSeeds seeds /*in = last launch's neighbours (use_seed), out = this launch's*/,
                                                    int use_seed, int use_cache /*cached planes were decided with this launch's planeEigenThreshold*/,
                                                    unsigned int* __restrict__ queue,
                                                    unsigned int* __restrict__ redo_count, int* __restrict__ redo_list,
                                                    unsigned int* __restrict__ changed_items,
                                                    unsigned long long* __restrict__ staged_total, int lds_boxes,
                                                    const int* __restrict__ item_order /*heaviest first, range boundaries behind it (k_order_items); may be null*/,
                                                    unsigned int* __restrict__ item_cost /*cycles per item of this launch (full sweeps only)*/,
                                                    int early_pop /*tuning knob: reserve the next item at the START of this one*/,
                                                    KnnCert cert)
{
    __shared__ __attribute__((aligned(16))) float s_m[4][4][64];
    __shared__ int s_list[4][kMaxList];
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float(*sm)[64] = s_m[wave];
    int* slist = s_list[wave];
    // VERIFY = true (warm-started launches): all items; the sweep only COUNTS the points within each query's
    //   K-th seed distance.  Count == number of seeds <=> the neighbour set is exactly the seeds (every seed lies
    //   within that distance and is met once), so the sorted seed list IS the answer and no list is maintained
    //   in the sweep.  Items with a lane whose count differs are queued in redo_list, untouched.
    // VERIFY = false: the full sweep with sorted-list insertion -- over all items (first launch on a cloud pair:
    //   redo_list == nullptr) or over the queued items only.
    const bool from_list = !VERIFY && redo_list != nullptr;
    if (from_list && *redo_count == 0u) return;  // nothing queued (uniform: before any barrier)
    const lds_f32* lbox = (const lds_f32*)s_dyn;
    if (lds_boxes) load_boxes_to_lds(mp, (lds_f32*)s_dyn);
    constexpr int kQ = 64 * QL;  // queries per item
    const int n_items = from_list ? (int)*redo_count : (N + kQ - 1) / kQ;
    unsigned long long wave_staged = 0ull;
    unsigned int wave_changed = 0u;  // items of this wave with a lane whose neighbour list differs from its seeds
    unsigned int wave_certified = 0u, wave_skipped = 0u;  // diagnostics
    // full sweeps are served heaviest item first inside each XCD's range (as k_nn_tiled: a launch is as long as its
    // longest wave, and the dense regions' items take several times the median)
    const int* order = from_list ? nullptr : item_order;
    WaveQueue wq(queue, lane, n_items, order ? order + n_items : nullptr);
    for (int raw = wq.first(); raw < n_items;) {
        // the next item is reserved LATE (behind the sweep, ahead of the long epilogue): reserved at the start, an item was
        // bound to a wave one whole item ahead of its execution and the launch's drain was two items long (see k_nn_tiled)
        int next_raw_v = 0;
        if (early_pop) next_raw_v = wq.pop();
        const int item = from_list ? __builtin_amdgcn_readfirstlane(redo_list[raw]) : (order ? __builtin_amdgcn_readfirstlane(order[raw]) : raw);
        const unsigned long long t_item0 = (item_cost && !from_list) ? __builtin_amdgcn_s_memtime() : 0ull;

        float qx[QL], qy[QL], qz[QL], reach[QL], kbound[QL];
        // the K best of each query, sorted ascending by the packed key (d2 bits << 32 | original index): d2 >= 0, so the
        // unsigned order of the key IS the lexicographic (d2, index) order -- one 64-bit compare per list step
        unsigned long long kk[QL][K];
        auto kd_of = [&](int k, int j) -> float { return __uint_as_float((unsigned int)(kk[k][j] >> 32)); };
        int kp[QL][K];            // sorted-map positions
        // insert (du, o, pos) into the sorted list of query k (caller has checked that it belongs there)
        auto insert = [&](int k, float du, unsigned int o, int pos) {
            kk[k][K - 1] = ((unsigned long long)__float_as_uint(du) << 32) | o; kp[k][K - 1] = pos;
#pragma unroll
            for (int j = K - 1; j > 0; --j) {
                const bool sw = kk[k][j] < kk[k][j - 1];
                if (!__any(sw)) break;  // every inserting lane has found its place (a new entry usually lands near the end)
                const unsigned long long tk = kk[k][j]; const int tp = kp[k][j];
                kk[k][j] = sw ? kk[k][j - 1] : tk; kp[k][j] = sw ? kp[k][j - 1] : tp;
                kk[k][j - 1] = sw ? tk : kk[k][j - 1]; kp[k][j - 1] = sw ? tp : kp[k][j - 1];
            }
        };
        int qi[QL];
        float lx[QL], ly[QL], lz[QL];
#pragma unroll
        for (int k = 0; k < QL; ++k) {
            qi[k] = item * kQ + k * 64 + lane;
            const int ic = qi[k] < N ? qi[k] : N - 1;
            lx[k] = slx[ic]; ly[k] = sly[ic]; lz[k] = slz[ic];
        }
#pragma unroll
        for (int k = 0; k < QL; ++k) {
            xform(P, lx[k], ly[k], lz[k], qx[k], qy[k], qz[k]);
#pragma unroll
            for (int j = 0; j < K; ++j) { kk[k][j] = (unsigned long long)__float_as_uint(thr2x) << 32; kp[k][j] = -1; }  // sentinel: (gate'^2, 0) never beaten by d2 >= gate'^2
        }
        if (use_seed) {
            // warm start: the K neighbours of the last launch are exact candidates; with them in the list the
            // reach is the K-th seed distance instead of the gate, and most tiles are never staged
#pragma unroll
            for (int k = 0; k < QL; ++k) {
                const int ic = qi[k] < N ? qi[k] : N - 1;
                int js[K];
                float gx[K], gy[K], gz[K];
                unsigned int go[K];
#pragma unroll
                for (int j = 0; j < K; ++j) {   // (one trip, coalesced: the seeds carry their coordinates and indices)
                    const size_t at = (size_t)j * seeds.stride + (size_t)ic;
                    js[j] = seeds.pos[at];
                    gx[j] = seeds.x[at]; gy[j] = seeds.y[at]; gz[j] = seeds.z[at]; go[j] = seeds.oidx[at];
                }
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const float du = dist2(qx[k], qy[k], qz[k], gx[j], gy[j], gz[j]);
                    if (js[j] >= 0 && du < thr2x) insert(k, du, go[j], js[j]);  // distinct positions: no duplicates among the seeds
                }
            }
        }
        unsigned long long cert_mask[QL];  // (wave-uniform) lanes whose first K entries are proven to be the K nearest
        bool lane_open = false;
#pragma unroll
        for (int k = 0; k < QL; ++k) {
            reach[k] = reach_of(kd_of(k, K - 1), qx[k], qy[k], qz[k]);  // last entry so far, or the gate while the list is not full
            kbound[k] = kd_of(k, K - 1);  // the sweep's box tests read it LIVE: it shrinks as the list fills (below)
            bool certd = false;
            if (cert.on && use_seed) {
                const int ic = qi[k] < N ? qi[k] : N - 1;
                float ox, oy, oz;
                xform(cert.Pprev, lx[k], ly[k], lz[k], ox, oy, oz);  // where the launch that wrote lb had this query
                const float delta = sqrtf(dist2(qx[k], qy[k], qz[k], ox, oy, oz)) * kCertUp + 1e-18f;
                // every point outside the stored list is at least this far away now -- the seeds dropped above for lying beyond
                // THIS launch's gate' have joined the outside (the gate may differ from the one lb was established under)
                const float m = fminf((cert.lb[ic] - delta) * kCertDown, sqrtf(thr2x) * kCertDown);
                const float mm = m * m * kCertDown;
                // ... beyond the K-th nearest seed, or beyond the gate if fewer than K seeds are inside it
                certd = qi[k] < N && m > 0.f && mm > fminf(kd_of(k, K - 2), thr2);
                if (certd) cert.lb[ic] = m;
            }
            cert_mask[k] = __ballot(certd);
            lane_open |= qi[k] < N && !certd;
            if (certd) { reach[k] = -1.0f; kbound[k] = -1.0f; }   // no reach of its own (it keeps its query: the epilogue needs it)
            if (qi[k] >= N) { qx[k] = qy[k] = qz[k] = 1.0e18f; reach[k] = -1.0f; kbound[k] = -1.0f; }  // padding lane
        }
        const bool skip_sweep = !__any(lane_open);  // every query of the item is certified
        // VERIFY: tau = K-th seed distance (list full), else the largest float below gate^2 ("d2 < gate^2" as "<=")
        float tau[QL];
        int expect[QL], cnt[QL] = {};
#pragma unroll
        for (int k = 0; k < QL; ++k) {
            int sds = 0;
#pragma unroll
            for (int j = 0; j < K; ++j) sds += kp[k][j] >= 0 ? 1 : 0;
            expect[k] = sds;
            tau[k] = sds == K ? kd_of(k, K - 1) : __uint_as_float(__float_as_uint(thr2x) - 1u);
            if (VERIFY && qi[k] < N && !((cert_mask[k] >> lane) & 1ull)) kbound[k] = tau[k];
        }

        // distances of 4 staged points to the lane's queries: QL = 2 pairs the two queries per packed instruction, QL = 1
        // pairs two points (each half the same IEEE sequence as the contract's dist2)
        auto dist4 = [&](const float (&xs)[4], const float (&ys)[4], const float (&zs)[4], float (&d)[QL][4]) {
            if constexpr (QL == 2) {
                const v2f q2x = {qx[0], qx[1]}, q2y = {qy[0], qy[1]}, q2z = {qz[0], qz[1]};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const v2f dv = dist2_pk(q2x, q2y, q2z, xs[u], ys[u], zs[u]);
                    d[0][u] = dv.x; d[1][u] = dv.y;
                }
            } else {
#pragma unroll
                for (int u = 0; u < 4; u += 2) {
                    const v2f mx = {xs[u], xs[u + 1]}, my = {ys[u], ys[u + 1]}, mz = {zs[u], zs[u + 1]};
                    const v2f dv = dist2_pk2(qx[0], qy[0], qz[0], mx, my, mz);
                    d[0][u] = dv.x; d[0][u + 1] = dv.y;
                }
            }
        };
        unsigned long long np_a = 0ull, np_b = 0ull;  // (profiling outputs of the sweep, unused here)
        unsigned int np_c = 0u, np_d = 0u, np_e = 0u;
        const unsigned long long n_staged = skip_sweep ? 0ull : tiled_sweep<QL, !VERIFY>(mp, lbox, lds_boxes != 0, slist, lane, sm, qx, qy, qz, reach, kbound, [&](int nm, int jb0, int jb1) {
            for (int m = 0; m < nm; m += 4) {
                const float4 X = *reinterpret_cast<const float4*>(&sm[0][m]);
                const float4 Y = *reinterpret_cast<const float4*>(&sm[1][m]);
                const float4 Z = *reinterpret_cast<const float4*>(&sm[2][m]);
                const float xs[4] = {X.x, X.y, X.z, X.w}, ys[4] = {Y.x, Y.y, Y.z, Y.w}, zs[4] = {Z.x, Z.y, Z.z, Z.w};
                float d[QL][4];
                dist4(xs, ys, zs, d);
                if constexpr (VERIFY) {
#pragma unroll
                    for (int k = 0; k < QL; ++k)
#pragma unroll
                        for (int u = 0; u < 4; ++u) cnt[k] += d[k][u] <= tau[k] ? 1 : 0;
                    continue;
                }
                bool cand = false;
#pragma unroll
                for (int k = 0; k < QL; ++k) cand |= fminf(fminf(d[k][0], d[k][1]), fminf(d[k][2], d[k][3])) <= kd_of(k, K - 1);
                if (__any(cand)) {  // some lane may have to insert: rare once the lists have tightened
                    const float4 O = *reinterpret_cast<const float4*>(&sm[3][m]);
                    const unsigned int os[4] = {__float_as_uint(O.x), __float_as_uint(O.y), __float_as_uint(O.z),
                                                __float_as_uint(O.w)};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int pos = (m + u) < 32 ? jb0 + m + u : jb1 + m + u - 32;
#pragma unroll
                        for (int k = 0; k < QL; ++k) {
                            const float du = d[k][u];
                            if ((((unsigned long long)__float_as_uint(du) << 32) | os[u]) < kk[k][K - 1]) {
                                bool dup = false;  // a seed met again by the sweep
#pragma unroll
                                for (int j = 0; j < K; ++j) dup |= kp[k][j] == pos;
                                if (!dup) insert(k, du, os[u], pos);
                            }
                        }
                    }
                    // Live bound: tiles tested from here on meet the K-th distance found so far.  On uniform clouds the lists
                    // tighten within the first tiles either way; on a spinning-lidar scan the gate ball of an unseeded query near
                    // the sensor holds ~10^4 points, and a launch was as long as those few items (1.2 ms of a 120k-point pair).
#pragma unroll
                    for (int k = 0; k < QL; ++k)
                        if (qi[k] < N) kbound[k] = kd_of(k, K - 1);
                }
            }
         }, false, np_a, np_b, np_c, np_d, np_e, np_a, np_b);
        if (!early_pop) next_raw_v = wq.pop();

        bool redo = false;
        if constexpr (VERIFY) {
            bool bad = false;
#pragma unroll
            for (int k = 0; k < QL; ++k) bad |= qi[k] < N && !((cert_mask[k] >> lane) & 1ull) && cnt[k] != expect[k];
            redo = __any(bad);
            if (redo && lane == 0) redo_list[atomicAdd(redo_count, 1u)] = item;
        }
        bool item_changed = false;
        if (!redo) {
        // the new lower bound of the lanes that were swept (a certified lane wrote its own above): every point outside the
        // list has d2 >= the list's last key -- in the counting flavour: > tau, nothing but the seeds was counted inside it
#pragma unroll
        for (int k = 0; k < QL; ++k)
            if (qi[k] < N && !((cert_mask[k] >> lane) & 1ull)) cert.lb[qi[k]] = sqrtf(VERIFY ? tau[k] : kd_of(k, K - 1)) * kCertDown;
#pragma unroll
        for (int k = 0; k < QL; ++k) wave_certified += (unsigned int)__popcll(cert_mask[k]);
        wave_skipped += skip_sweep ? 1u : 0u;
        // Epilogue.  The plane (centroid, normal, is-it-planar) depends only on the ordered neighbour list -- map
        // points, fixed for the align -- so when a query's list equals the last launch's, the cached plane is reused
        // bit for bit and the fp64 covariance + Jacobi eigen-solve (dearer than the search itself) is skipped.  Near
        // convergence almost no list changes; a wave pays for the solve only if one of its lanes needs it.
#pragma unroll
        for (int k = 0; k < QL; ++k) {
            float kd[K];
            unsigned int ko[K];
#pragma unroll
            for (int j = 0; j < K; ++j) { kd[j] = kd_of(k, j); ko[j] = (unsigned int)(kk[k][j] & 0xffffffffull); }
            item_changed |= plane_epilogue<K>(mp, kp[k], kd, ko, qx[k], qy[k], qz[k], qi[k], N, thr2, threshold, plane_eig_thr, out, cache, seeds,
                                              use_seed, use_cache);
        }
        }  // (epilogue)
        wave_changed += item_changed ? 1u : 0u;
        wave_staged += n_staged * QL;  // units of 64 (query, point) pairs
        if (item_cost && !from_list && lane == 0) {
            const unsigned long long c = __builtin_amdgcn_s_memtime() - t_item0;
            item_cost[item] = c > 0xffffffffull ? 0xffffffffu : (unsigned int)c;
        }
        raw = wq.settle(__builtin_amdgcn_readfirstlane(next_raw_v));
    }
    if (lane == 0 && wave_staged && staged_total)  // (statistics, slotted, only when the caller profiles: see k_nn_tiled)
        atomicAdd(staged_total + (size_t)(blockIdx.x & (kStatSlots - 1)) * kStatStride, wave_staged);
    if (cert.stats && lane == 0 && wave_certified) {
        unsigned long long* st = cert.stats + (size_t)(blockIdx.x & (kStatSlots - 1)) * kStatStride;
        atomicAdd(st + 1, (unsigned long long)wave_certified);
        if (wave_skipped) atomicAdd(st + 2, (unsigned long long)wave_skipped);
    }
    // (the verify flavour reports its queued items through redo_count; the queued-items launch must not count twice)
    // (8 slots on separate lines -- word 1 of the OTHER launch's queue lines: one address would see 3072 end-of-wave atomics)
    if (!VERIFY && !from_list && lane == 0 && wave_changed) atomicAdd(changed_items + (size_t)(blockIdx.x & (kQueues - 1)) * kQueueStride, wave_changed);
}

// ---- k_knn_coop: the plane matcher for ODOMETRY-SIZE clouds -- one workgroup per 64-query item -----------------------------
// With <= ~0.2M queries k_knn_planes has fewer items than wave slots: a launch is one item per wave and as long as its
// slowest item -- a lone wave walking a chain of dependent round trips and, for an unseeded or far-moved query of a
// spinning lidar's dense core, thousands of points (a KITTI-like 120k-point pair: 75-310 us per launch against 41 us for
// a point-to-point iteration at that size).  Here the four waves of a workgroup hold the SAME 64 queries and deal the
// candidate tiles among themselves exactly as k_nn_coop does (coop_sweep: wave 0 walks the upper box levels into the shared
// list, wave w owns the tiles t of super-tile S with (t + S) % 4 == w).  Every wave keeps its own sorted list of K entries,
// seeded alike; its bound -- the last key of ITS list -- is never below the final one (a list over fewer points), so a
// tile its owner culls holds nothing that belongs in the merged list.  The three other lists go through LDS, wave 0 merges
// them into its own (seeds are in all four: dropped by position) and runs the epilogue.  Same lists, same planes, same
// certified-list logic (KnnCert) as k_knn_planes: results are identical.
// One problem of a (possibly batched) k_knn_coop launch: K initial poses on one cloud pair (the loop-closure Monte-Carlo,
// src/LidarOdometry.cpp:767-788) or K different pairs (the nearby-keyframe checks, cpp:704-741) share one launch through
// blockIdx.y, as the NN matcher's problems do in k_nn_coop -- the reference's own nearby / loop-closure settings select
// THIS matcher (params/icp-settings-loop-closure.yaml:33-39).
struct KnnProblem {
    const float *slx, *sly, *slz;   // Hilbert-sorted local cloud (queries)
    int N;
    TiledMap mp;                    // Hilbert-sorted map + box levels
    PoseF P;                        // this launch's pose
    PoseF Pprev;                    // the pose of the launch that wrote lb / the seeds (KnnCert)
    PlanePair* out;                 // the plane pairing, sorted query order
    PlanePair* cache;               // the plane of each query's stored list
    KnnSeeds seeds;                 // in = last launch's neighbours (use_seed), out = this launch's
    float* lb;                      // per query: lower bound on the distance to every point outside its list (KnnCert)
    int use_seed, use_cache, cert_on;
    unsigned int* changed_items;    // 8 slots, kQueueStride words apart: items whose lists changed
    unsigned int* cost;             // per 64-query item: shader cycles the item took in this launch (null: not recorded) -> the next launch's order
};
constexpr int kKnnMaxBatch = 12;    // problems per launch (kernel arguments are limited to 4 KB; = kCoopMaxBatch)
template <int KMAX> struct KnnBatch { KnnProblem p[KMAX]; };
static_assert(sizeof(KnnBatch<kKnnMaxBatch>) <= 3900, "KnnBatch must fit the kernel-argument segment");

// (knn_q4_launch.hip -- k_knn_q4's translation unit -- needs the types and plane_epilogue above, not the kernels below: the non-template
//  ones among them may be defined in one translation unit only)
#ifndef MOLA_ICP_PLANE_TYPES_ONLY

// LDS of one cooperative item (k_knn_coop, k_knn_coop_groups)
template <int K>
struct KnnCoopLds {
    __attribute__((aligned(16))) float s_m[4][4][64];
    int s_list[kMaxList];
    float s_wbox[6];
    int s_ctl[4];
    int s_tick[kMaxList];
    unsigned long long s_mk[kCoopParts - 1][K][64];
    int s_mpos[kCoopParts - 1][K][64];
};

// One cooperative item: the four waves of the workgroup hold the same 64 queries (lane -> qi; >= N: none), deal the candidate
// tiles among themselves (coop_sweep), merge their lists through LDS; wave 0 runs the plane epilogue.  Called by ALL four waves
// (barriers inside); nothing of S may be touched by the caller before its next barrier.  CERT: derive the certificates (KnnCert).
constexpr int kKnnDiagWords = 14;   // MOLA_ICP_DEBUG_STATS=4: shader-clock phases of every wave of every item (see knn_coop_item)
template <int K, bool CERT, bool DIAG = false>
__device__ __forceinline__ void knn_coop_item(KnnCoopLds<K>& S, const KnnProblem& pb, const TiledMap& mp, const lds_f32* lbox, int lds_boxes, int qi,
                                              float thr2, float thr2x, double threshold, double plane_eig_thr,
                                              unsigned long long* __restrict__ staged_total, unsigned long long* __restrict__ cert_stats,
                                              int lane, int wave, unsigned long long* __restrict__ diag = nullptr /*DIAG: this wave's record*/)
{
    unsigned long long dt[6] = {};
    if (DIAG) dt[0] = __builtin_amdgcn_s_memtime();
    const unsigned long long t_item0 = (wave == 0 && pb.cost) ? __builtin_amdgcn_s_memtime() : 0ull;
    // (the body is written out with plain locals, as the kernel had it before it became a function: the same lists held in a struct
    //  cost this kernel 40 vector registers -- 123 -> 164 -- and with them its fourth workgroup per CU)
    const int N = pb.N;
    const PoseF P = pb.P;
    const float* __restrict__ slx = pb.slx;
    const float* __restrict__ sly = pb.sly;
    const float* __restrict__ slz = pb.slz;
    PlanePair* __restrict__ out = pb.out;
    PlanePair* __restrict__ cache = pb.cache;
    const KnnSeeds seeds = pb.seeds;
    const int use_seed = pb.use_seed, use_cache = pb.use_cache;
    unsigned int* __restrict__ changed_items = pb.changed_items;
    KnnCert cert;
    cert.Pprev = pb.Pprev;
    cert.lb = pb.lb;
    cert.on = CERT ? pb.cert_on : 0;
    cert.stats = cert_stats;

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
    const int ic = qi < N ? qi : N - 1;
    const float lx = slx[ic], ly = sly[ic], lz = slz[ic];
    float qx, qy, qz;
    xform(P, lx, ly, lz, qx, qy, qz);
#pragma unroll
    for (int j = 0; j < K; ++j) { kk[j] = (unsigned long long)__float_as_uint(thr2x) << 32; kp[j] = -1; }
    if (use_seed) {
        int js[K];
        float gx[K], gy[K], gz[K];
        unsigned int go[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {   // (one trip, coalesced: the seeds carry their coordinates and indices)
            const size_t at = (size_t)j * seeds.stride + (size_t)ic;
            js[j] = seeds.pos[at];
            gx[j] = seeds.x[at]; gy[j] = seeds.y[at]; gz[j] = seeds.z[at]; go[j] = seeds.oidx[at];
        }
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const float du = dist2(qx, qy, qz, gx[j], gy[j], gz[j]);
            if (js[j] >= 0 && du < thr2x) insert(du, go[j], js[j]);
        }
    }
    // certified lists (KnnCert): every wave derives the same verdicts from the same data -- which is why NOTHING is written
    // (bounds, seeds, planes: wave 0, in the epilogue) before every wave has read its part: the barrier below
    bool certd = false;
    float lb_new = 0.f;
    if (CERT && cert.on && use_seed) {
        float ox, oy, oz;
        xform(cert.Pprev, lx, ly, lz, ox, oy, oz);
        const float delta = sqrtf(dist2(qx, qy, qz, ox, oy, oz)) * kCertUp + 1e-18f;
        const float m = fminf((cert.lb[ic] - delta) * kCertDown, sqrtf(thr2x) * kCertDown);
        const float mm = m * m * kCertDown;
        certd = qi < N && m > 0.f && mm > fminf(kd_of(K - 2), thr2);
        lb_new = m;
    }
    const unsigned long long cert_mask = __ballot(certd);
    const bool skip_sweep = !__any(qi < N && !certd);
    if (DIAG) dt[1] = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (DIAG) dt[2] = __builtin_amdgcn_s_memtime();
    float q2x[2] = {qx, 1.0e18f}, q2y[2] = {qy, 1.0e18f}, q2z[2] = {qz, 1.0e18f};
    float reach2[2] = {reach_of(kd_of(K - 1), qx, qy, qz), -1.0f};
    float kb2[2] = {kd_of(K - 1), -1.0f};
    if (certd) { reach2[0] = -1.0f; kb2[0] = -1.0f; }
    if (qi >= N) { q2x[0] = q2y[0] = q2z[0] = 1.0e18f; reach2[0] = -1.0f; kb2[0] = -1.0f; }
    unsigned long long n_staged = 0ull;
    unsigned long long pc[5] = {};
    unsigned int pn[3] = {};
    unsigned int dg_groups = 0u, dg_slow = 0u, dg_keypass = 0u, dg_dup = 0u, dg_ins = 0u;   // DIAG: what the visitor did (lane events summed over the wave)
    if (!skip_sweep) {   // (workgroup-uniform: the barriers inside are met by all four waves)
        float(*sm)[64] = S.s_m[wave];
        n_staged = coop_sweep<true>(mp, lbox, lds_boxes != 0, S.s_list, S.s_wbox, S.s_ctl, S.s_tick, lane, wave, sm, q2x, q2y, q2z, reach2, kb2, [&](int nm, int jb0, int jb1) {
            for (int m = 0; m < nm; m += 4) {
                const float4 X = *reinterpret_cast<const float4*>(&sm[0][m]);
                const float4 Y = *reinterpret_cast<const float4*>(&sm[1][m]);
                const float4 Z = *reinterpret_cast<const float4*>(&sm[2][m]);
                const float xs[4] = {X.x, X.y, X.z, X.w}, ys[4] = {Y.x, Y.y, Y.z, Y.w}, zs[4] = {Z.x, Z.y, Z.z, Z.w};
                float d[4];
#pragma unroll
                for (int u = 0; u < 4; u += 2) {
                    const v2f mx = {xs[u], xs[u + 1]}, my = {ys[u], ys[u + 1]}, mz = {zs[u], zs[u + 1]};
                    const v2f dv = dist2_pk2(q2x[0], q2y[0], q2z[0], mx, my, mz);
                    d[u] = dv.x; d[u + 1] = dv.y;
                }
                const bool cand = fminf(fminf(d[0], d[1]), fminf(d[2], d[3])) <= kd_of(K - 1);
                if (DIAG) dg_groups += 1u;
                if (__any(cand)) {
                    if (DIAG) dg_slow += 1u;
                    const float4 O = *reinterpret_cast<const float4*>(&sm[3][m]);
                    const unsigned int os[4] = {__float_as_uint(O.x), __float_as_uint(O.y), __float_as_uint(O.z), __float_as_uint(O.w)};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int pos = (m + u) < 32 ? jb0 + m + u : jb1 + m + u - 32;
                        const bool keypass = (((unsigned long long)__float_as_uint(d[u]) << 32) | os[u]) < kk[K - 1];
                        if (DIAG) dg_keypass += (unsigned int)__popcll(__ballot(keypass));
                        if (keypass) {
                            bool dup = false;  // a seed met again by the sweep
#pragma unroll
                            for (int j = 0; j < K; ++j) dup |= kp[j] == pos;
                            if (DIAG) { dg_dup += (unsigned int)__popcll(__ballot(dup)); dg_ins += (unsigned int)__popcll(__ballot(!dup)); }
                            if (!dup) insert(d[u], os[u], pos);
                        }
                    }
                    if (qi < N && !certd) kb2[0] = kd_of(K - 1);   // live bound (see k_knn_planes)
                }
            }
        }, DIAG, pc, pn);
        if (DIAG) dt[3] = __builtin_amdgcn_s_memtime();
        // merge: the other three waves' lists through LDS into wave 0's
        if (wave > 0) {
#pragma unroll
            for (int j = 0; j < K; ++j) { S.s_mk[wave - 1][j][lane] = kk[j]; S.s_mpos[wave - 1][j][lane] = kp[j]; }
        }
        __syncthreads();
        if (wave == 0) {
            for (int w = 0; w < kCoopParts - 1; ++w) {
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const unsigned long long ck = S.s_mk[w][j][lane];
                    const int cp = S.s_mpos[w][j][lane];
                    bool take = cp >= 0 && ck < kk[K - 1];
#pragma unroll
                    for (int i = 0; i < K; ++i) take &= kp[i] != cp;
                    if (__any(take)) {
                        if (take) insert(__uint_as_float((unsigned int)(ck >> 32)), (unsigned int)(ck & 0xffffffffu), cp);
                    }
                }
            }
        }
    }
    if (lane == 0 && n_staged && staged_total)   // (each wave its own tiles: units of 64 pairs, one query per lane)
        atomicAdd(staged_total + (size_t)(blockIdx.x & (kStatSlots - 1)) * kStatStride, n_staged);
    if (DIAG) { dt[4] = __builtin_amdgcn_s_memtime(); if (skip_sweep) dt[3] = dt[2]; }
    auto diag_record = [&](bool changed) {   // [prologue, wait, sweep, list fill + barrier, tile-box wait, tile tests + passes, staging, distance passes, merge, epilogue, counts, total]
        if (DIAG && diag && lane == 0) {
            const unsigned long long t_end = __builtin_amdgcn_s_memtime();
            const unsigned long long n_open = (unsigned long long)__popcll(__ballot(true));   // (placeholder lane count: whole wave)
            (void)n_open;
            diag[0] = dt[1] - dt[0]; diag[1] = dt[2] - dt[1]; diag[2] = dt[3] - dt[2]; diag[3] = pc[4]; diag[4] = pc[2]; diag[5] = pc[3];
            diag[6] = pc[0]; diag[7] = pc[1]; diag[8] = dt[4] - dt[3]; diag[9] = t_end - dt[4];
            diag[10] = (n_staged & 0xfffffull) | ((unsigned long long)(pn[1] & 0xfffu) << 20) | ((unsigned long long)(pn[2] & 0xfffffu) << 32) |
                       ((unsigned long long)__popcll(~cert_mask) << 52) | ((unsigned long long)(changed ? 1 : 0) << 60) | ((unsigned long long)(skip_sweep ? 1 : 0) << 61);
            diag[11] = t_end - dt[0];
            diag[12] = (unsigned long long)dg_groups | ((unsigned long long)dg_slow << 32);
            diag[13] = (unsigned long long)dg_keypass | ((unsigned long long)dg_dup << 21) | ((unsigned long long)dg_ins << 42);
        }
    };
    if (wave != 0) { diag_record(false); return; }   // (no barrier below)
    if (qi < N) cert.lb[qi] = certd ? lb_new : sqrtf(kd_of(K - 1)) * kCertDown;
    float kd[K];
    unsigned int ko[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { kd[j] = kd_of(j); ko[j] = (unsigned int)(kk[j] & 0xffffffffull); }
    const bool changed = plane_epilogue<K>(mp, kp, kd, ko, qx, qy, qz, qi, N, thr2, threshold, plane_eig_thr, out, cache, seeds, use_seed, use_cache);
    if (lane == 0 && changed && changed_items) atomicAdd(changed_items + (size_t)(blockIdx.x & (kQueues - 1)) * kQueueStride, 1u);   // (batched launches do not count)
    if (cert.stats && lane == 0 && cert_mask) {
        unsigned long long* st = cert.stats + (size_t)(blockIdx.x & (kStatSlots - 1)) * kStatStride;
        atomicAdd(st + 1, (unsigned long long)__popcll(cert_mask));
        if (skip_sweep) atomicAdd(st + 2, 1ull);
    }
    if (pb.cost && lane == 0) {   // (the leader wave is the item's last: its lifetime is the item's)
        const unsigned long long c = __builtin_amdgcn_s_memtime() - t_item0;
        pb.cost[qi >> 6] = (unsigned int)(c < 0xffffffffull ? c : 0xffffffffull);
    }
    diag_record(changed);
}

template <int K /*list length: knn + 1*/, int KMAX /*problems per launch: blockIdx.y*/, bool DIAG = false>
__global__ __launch_bounds__(256, (K <= 7 ? 4 : 3)) void k_knn_coop(const KnnBatch<KMAX> batch, float thr2, float thr2x, double threshold, double plane_eig_thr,
                                                     unsigned long long* __restrict__ staged_total /*slotted, may be null*/, int lds_boxes,
                                                     unsigned long long* __restrict__ cert_stats /*diagnostics, may be null*/,
                                                     unsigned long long* __restrict__ diag = nullptr /*DIAG: [item][wave][kKnnDiagWords]*/)
{
    __shared__ KnnCoopLds<K> S;
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const KnnProblem& pb = batch.p[KMAX == 1 ? 0 : blockIdx.y];
    const int N = pb.N;
    // (a map whose box levels fit LDS fits the L2s whole: items dealt to the XCDs one by one -- a contiguous eighth each left the XCD
    // that holds a scan's dense core with 1.3x the others' work: odometry stream 0.559 -> 0.540 ms per scan; larger maps keep the ranges)
    const int item = lds_boxes ? (int)blockIdx.x : xcd_item((int)blockIdx.x, (N + 63) / 64);
    if (item * 64 >= N) return;  // (the grid is rounded up to whole XCD ranges / sized for the batch's largest problem; uniform: before any barrier)
    const TiledMap mp = pb.mp;
    const lds_f32* lbox = (const lds_f32*)s_dyn;
    if (lds_boxes) load_boxes_to_lds(mp, (lds_f32*)s_dyn);
    knn_coop_item<K, true, DIAG>(S, pb, mp, lbox, lds_boxes, item * 64 + lane, thr2, thr2x, threshold, plane_eig_thr, staged_total, cert_stats, lane, wave,
                                 DIAG && diag ? diag + ((size_t)item * 4 + wave) * kKnnDiagWords : nullptr);
}

// Seeds for a launch that has no lists yet (HipWorkspace::match_planes).  An NN pass has given every query its nearest map point;
// that point names a PLACE in the map -- its tile and the two tiles next to it on the Hilbert curve, 96 points that lie around the
// query if anything does -- and the KL nearest of those 96 are the seeds: exact candidates like any seed (the sweep that follows is
// complete under the bound they give), -1 where the query has no neighbour.  (Round 3 took the KL points AROUND the neighbour on the
// curve, whichever they were: the first launch of an odometry scan then evaluated 2 000 pairs per query where the second, seeded by
// real lists, evaluates 350.)
// Not the KL nearest of the 96 exactly -- the best of each of G = 8 (12 for lists of nine) interleaved groups of candidates, sorted,
// the first KL kept: one compare and two selects per candidate where a sorted insertion, with 64 lanes inserting at different
// times, walked its bubble for nearly every point (31 us at 120k queries against 8).  The bound the sweep starts from is the
// largest of them: a little above the true KL-th distance, far below the curve neighbours'.
// BY_KEY: no NN pass at all -- the place is where the query's own Hilbert key (in the map's frame) falls among the map's sorted keys:
// the map points of its cell, or the cells next to it on the curve.  Close in space most of the time, not always (the curve has
// seams); a poor place only costs a wider sweep.
template <int KL, bool BY_KEY>
__global__ __launch_bounds__(256) void k_bootstrap_seeds(const int* __restrict__ nn_pos /*sorted-map position of the NN per sorted query (!BY_KEY)*/,
                                                         const unsigned int* __restrict__ map_keys /*the map's keys, ascending (BY_KEY)*/,
                                                         const float* __restrict__ map_box /*the box they were quantised in (BY_KEY)*/,
                                                         const float* __restrict__ slx, const float* __restrict__ sly,
                                                         const float* __restrict__ slz, int N, PoseF P, TiledMap mp, int M, KnnSeeds seeds)
{
    constexpr int G = KL <= 8 ? 8 : (KL <= 12 ? 12 : (KL <= 16 ? 16 : 24));
    static_assert(KL <= G && (3 * kTileG) % G == 0 && G % 4 == 0, "groups");
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    float qx, qy, qz;
    xform(P, slx[i], sly[i], slz[i], qx, qy, qz);
    int pos;
    if constexpr (BY_KEY) {
        float box[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) box[k] = map_box[k];
        const unsigned int key = hilbert_key_in_box(box, qx, qy, qz);
        // (a base-8 lifting search -- seven probes per level, six levels instead of seventeen -- and batched loads of the three tiles
        // were measured: 27.5 us against 23.1, profiles/r04/dropped_bootstrap_lifting_search.txt; the kernel is not its search)
        int step = 1;
        while (step < M) step <<= 1;   // (uniform)
        pos = 0;   // number of map keys below the query's
        for (step >>= 1; step > 0; step >>= 1) {
            const int q = pos + step;
            if (q <= M && map_keys[q - 1] < key) pos = q;
        }
        pos = pos < M ? pos : M - 1;
    } else {
        pos = nn_pos[i];
    }
    if (pos < 0) {
#pragma unroll
        for (int j = 0; j < KL; ++j) seeds.pos[(size_t)j * seeds.stride + (size_t)i] = -1;
        return;
    }
    const int n_tiles = (M + kTileG - 1) / kTileG;
    int t0 = (pos / kTileG) - 1;
    t0 = t0 > n_tiles - 3 ? n_tiles - 3 : t0;
    t0 = t0 < 0 ? 0 : t0;
    unsigned long long kk[G];   // per group: the smallest (d2 bits << 32 | sorted position)
#pragma unroll
    for (int j = 0; j < G; ++j) kk[j] = ~0ull;
    for (int c = 0; c < 3 * kTileG; c += G) {
        const int p0 = t0 * kTileG + c;
        if (p0 >= M) break;   // (the sorted arrays are padded to whole super-tiles: the loads below stay inside them)
#pragma unroll
        for (int h = 0; h < G; h += 4) {
            const float4 X = *reinterpret_cast<const float4*>(mp.sx + p0 + h);
            const float4 Y = *reinterpret_cast<const float4*>(mp.sy + p0 + h);
            const float4 Z = *reinterpret_cast<const float4*>(mp.sz + p0 + h);
            const float xs[4] = {X.x, X.y, X.z, X.w}, ys[4] = {Y.x, Y.y, Y.z, Y.w}, zs[4] = {Z.x, Z.y, Z.z, Z.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                unsigned long long key = ((unsigned long long)__float_as_uint(dist2(qx, qy, qz, xs[u], ys[u], zs[u])) << 32) | (unsigned int)(p0 + h + u);
                if (p0 + h + u >= M) key = ~0ull;   // (a padding point)
                kk[h + u] = key < kk[h + u] ? key : kk[h + u];
            }
        }
    }
    // ascending (a bubble network over G values), the first KL are the seeds
#pragma unroll
    for (int a = 0; a < G - 1; ++a)
#pragma unroll
        for (int j = G - 1; j > a; --j) {
            const bool sw = kk[j] < kk[j - 1];
            const unsigned long long x = kk[j - 1], y = kk[j];
            kk[j - 1] = sw ? y : x;
            kk[j] = sw ? x : y;
        }
#pragma unroll
    for (int j = 0; j < KL; ++j) {
        const size_t at = (size_t)j * seeds.stride + (size_t)i;
        if (kk[j] == ~0ull) { seeds.pos[at] = -1; continue; }
        const int ps = (int)(unsigned int)(kk[j] & 0xffffffffull);
        seeds.pos[at] = ps;
        seeds.x[at] = mp.sx[ps]; seeds.y[at] = mp.sy[ps]; seeds.z[at] = mp.sz[ps];
        seeds.oidx[at] = (unsigned int)mp.perm[ps];
    }
}

// The nearest-neighbour matcher's seeds from the plane matcher's lists (the quality pass behind a point-to-plane loop: its queries
// start from the first entry of their lists -- the nearest map point at the loop's last pose -- instead of from nothing)
__global__ __launch_bounds__(256) void k_nn_seeds_from_lists(KnnSeeds seeds, int N, int* __restrict__ pos_s, int* __restrict__ idx_s,
                                                             float* __restrict__ gsx, float* __restrict__ gsy, float* __restrict__ gsz)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int ps = seeds.pos[i];   // (entry 0)
    pos_s[i] = ps;
    idx_s[i] = ps >= 0 ? (int)seeds.oidx[i] : -1;
    gsx[i] = ps >= 0 ? seeds.x[i] : 0.f; gsy[i] = ps >= 0 ? seeds.y[i] : 0.f; gsz[i] = ps >= 0 ? seeds.z[i] : 0.f;
}

// The PairedRatio pass behind a point-to-plane loop, from the lists alone (src/LidarOdometry.cpp:869-871 hands back `quality` with
// every align; params/icp-settings-regular.yaml:33-36: thresholdDistance 0.10 m).  The quality is a COUNT -- queries whose nearest map
// point lies within the threshold at the final pose -- and the lists answer that without a search: (i) an entry of the query's stored
// list closer than the threshold (contract arithmetic on the stored coordinates: the same bits the matcher would compute) => paired,
// whatever else is out there; (ii) no such entry, and every point OUTSIDE the list provably beyond the threshold -- at least lb away at
// the pose that wrote the list, the query has moved by delta since (KnnCert's bound and margins) => not paired; (iii) neither: counted
// as `open`, and the caller runs the nearest-neighbour pass after all (a query whose whole list sits within 2 delta of the threshold:
// practically never).  Entry 0 was the nearest at the list's pose: nearly every paired query stops there.
// One launch of <= kQualityBlocks workgroups; block g publishes {pairs | open << 32, sequence} as record g of the pinned block.
constexpr int kQualityBlocks = 256;   // at most: one 16-byte record each in the pinned block
template <int KL>
__global__ __launch_bounds__(256) void k_quality_from_lists(const float* __restrict__ slx, const float* __restrict__ sly,
                                                            const float* __restrict__ slz, int N, PoseF P, PoseF Pprev, float thr2,
                                                            KnnSeeds seeds, const float* __restrict__ lb,
                                                            unsigned long long* __restrict__ host_out, unsigned long long seq)
{
    __shared__ unsigned int s_pairs[4], s_open[4];
    unsigned int pairs = 0u, open = 0u;
    // four queries per thread and trip, everything the common case needs -- the query, entry 0 of its list, its bound -- loaded in
    // ONE round trip (a first version walked query by query, entry by entry: 26 us at 120k queries for 3 MB of loads; 14 with the
    // trips batched on 32 workgroups)
    constexpr int U = 4;
    const int stride = (int)gridDim.x * 256;
    for (int i0 = (int)blockIdx.x * 256 + (int)threadIdx.x; i0 < N; i0 += U * stride) {
        float lx[U], ly[U], lz[U], ex[U], ey[U], ez[U], bound[U];
        int ep[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * stride;
            const int ic = i < N ? i : N - 1;
            lx[u] = slx[ic]; ly[u] = sly[ic]; lz[u] = slz[ic];
            ep[u] = seeds.pos[ic]; ex[u] = seeds.x[ic]; ey[u] = seeds.y[ic]; ez[u] = seeds.z[ic];
            bound[u] = lb[ic];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * stride;
            if (i >= N) continue;
            float qx, qy, qz;
            xform(P, lx[u], ly[u], lz[u], qx, qy, qz);
            bool paired = ep[u] >= 0 && dist2(qx, qy, qz, ex[u], ey[u], ez[u]) < thr2;
            if (!paired) {   // the other entries, all loads at once (a lane here is the exception: most paired queries stop at entry 0)
                int ps[KL];
                float fx[KL], fy[KL], fz[KL];
#pragma unroll
                for (int e = 1; e < KL; ++e) {
                    const size_t at = (size_t)e * seeds.stride + (size_t)i;
                    ps[e] = seeds.pos[at]; fx[e] = seeds.x[at]; fy[e] = seeds.y[at]; fz[e] = seeds.z[at];
                }
#pragma unroll
                for (int e = 1; e < KL; ++e) paired |= ps[e] >= 0 && dist2(qx, qy, qz, fx[e], fy[e], fz[e]) < thr2;
            }
            if (paired) { ++pairs; continue; }
            float ox, oy, oz;
            xform(Pprev, lx[u], ly[u], lz[u], ox, oy, oz);
            const float delta = sqrtf(dist2(qx, qy, qz, ox, oy, oz)) * kCertUp + 1e-18f;
            const float m = (bound[u] - delta) * kCertDown;
            if (!(m > 0.f && m * m * kCertDown > thr2)) ++open;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { pairs += __shfl_down(pairs, o); open += __shfl_down(open, o); }
    if (lane == 0) { s_pairs[wave] = pairs; s_open[wave] = open; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int tp = 0u, to = 0u;
        for (int w = 0; w < 4; ++w) { tp += s_pairs[w]; to += s_open[w]; }
        volatile unsigned long long* rec = host_out + 2 * (size_t)blockIdx.x;   // {pairs | open << 32, sequence}
        rec[0] = (unsigned long long)tp | ((unsigned long long)to << 32);
        __threadfence_system();
        rec[1] = seq;
        __threadfence_system();
    }
}

// the point-to-plane cost  sum (n.(R l + t - c))^2  is the quadratic form  x^T A x - 2 b^T x + c0  in
// x = [R row-major (9), t (3)]  with  phi = [n (x) l, n],  d = n.c :   A = sum phi phi^T (78 unique),
// b = sum phi d (12), c0 = sum d^2, count.  ONE pass -> the whole Gauss-Newton inner loop runs on the host.
constexpr int kNAccPlane = 92;  // 78 + 12 + 1 + 1
__global__ __launch_bounds__(256) void k_accumulate_planes(const float* __restrict__ slx, const float* __restrict__ sly,
                                                           const float* __restrict__ slz, const PlanePair* __restrict__ pairs,
                                                           int N, double* __restrict__ partials)
{
    double acc[kNAccPlane];
#pragma unroll
    for (int k = 0; k < kNAccPlane; ++k) acc[k] = 0.0;
    // 92 fp64 accumulators leave one wave per SIMD: nothing hides a load but the thread's own next element -- fetched while
    // this one is accumulated (same order of additions: 20 -> ~12 us at 1M)
    const int stride = (int)gridDim.x * 256;
    int i = (int)blockIdx.x * 256 + (int)threadIdx.x;
    PlanePair pp{};
    float l0 = 0.f, l1 = 0.f, l2 = 0.f;
    bool have = i < N;
    if (have) { pp = pairs[i]; l0 = slx[i]; l1 = sly[i]; l2 = slz[i]; }
    while (have) {
        const int inext = i + stride;
        const bool hn = inext < N;
        PlanePair pn{};
        float n0 = 0.f, n1 = 0.f, n2 = 0.f;
        if (hn) { pn = pairs[inext]; n0 = slx[inext]; n1 = sly[inext]; n2 = slz[inext]; }
        if (pp.valid) {
            const double l[3] = {l0, l1, l2};
            double phi[12];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
#pragma unroll
                for (int c = 0; c < 3; ++c) phi[3 * r + c] = pp.n[r] * l[c];
                phi[9 + r] = pp.n[r];
            }
            const double d = pp.n[0] * pp.c[0] + pp.n[1] * pp.c[1] + pp.n[2] * pp.c[2];
            int q = 0;
#pragma unroll
            for (int a = 0; a < 12; ++a)
#pragma unroll
                for (int b = a; b < 12; ++b) acc[q++] += phi[a] * phi[b];
#pragma unroll
            for (int a = 0; a < 12; ++a) acc[78 + a] += phi[a] * d;
            acc[90] += d * d;
            acc[91] += 1.0;
        }
        pp = pn; l0 = n0; l1 = n1; l2 = n2; have = hn; i = inext;
    }
    block_sum_256<kNAccPlane>(acc, partials + (size_t)blockIdx.x * kNAccPlane);  // fixed order
}

// The same sums on the matrix cores.  With v = [phi (12), d, 1, 0, 0] per pairing, the whole form is ONE symmetric 16 x 16 matrix
// D = sum v v^T  (A = D[a][b], b = D[a][12], c0 = D[12][12], count = D[13][13]) -- a rank-4 update per v_mfma_f64_16x16x4_f64:
// lane l supplies v_{l & 15} of pairing (l >> 4) of a chunk of four, and since A[i][k] = B[k][i] here the SAME register is both
// operands.  A wave writes its 64 pairings' vectors to LDS ([pairing][16] doubles, padded rows), reads them back one f64 per lane
// per chunk, and keeps D in the accumulator (4 f64 per lane: col = lane & 15, row = (lane >> 4) + 4 r) across all its batches: no
// per-thread accumulators at all (k_accumulate_planes holds 92 of them -- 184 VGPRs, one wave per SIMD -- which is what made a
// bandwidth-sized kernel latency-bound: 19 us at 1M, 9.7 us at 120k).  The four waves' matrices are added in wave order and the row
// is written in k_accumulate_planes' layout.  Deterministic (fixed order for a given grid); differs from the VALU form by summation
// order only.
typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int kPhiStride = 18;   // doubles per pairing in LDS: 16 + 2 (rows 144 bytes apart: 16-byte aligned, banks spread)
// block `bx` of `nblocks` over one problem's plane pairing -> one row of the form (the summation order depends on (N, nblocks)
// only: the single-problem and the batched launch produce the same bits)
__device__ __forceinline__ void accumulate_planes_mfma_rows(const float* __restrict__ slx, const float* __restrict__ sly,
                                                            const float* __restrict__ slz, const PlanePair* __restrict__ pairs,
                                                            int N, int bx, int nblocks, double* __restrict__ partials)
{
    __shared__ __attribute__((aligned(16))) double s_v[4][64 * kPhiStride];   // per wave: the vectors of its 64 pairings
    __shared__ double s_d[4][256];                                             // per wave: its 16 x 16 result
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double* sv = s_v[wave];
    v4d D = {0.0, 0.0, 0.0, 0.0};
    const int n_batches = (N + 63) / 64, n_waves = nblocks * 4;
    // (a wave's batches are a chain of round trips otherwise -- 2 waves per SIMD at 1M: the next batch is fetched while this one runs)
    int bt = bx * 4 + wave;
    PlanePair pp{};
    float l0 = 0.f, l1 = 0.f, l2 = 0.f;
    if (bt < n_batches && bt * 64 + lane < N) { const int i = bt * 64 + lane; pp = pairs[i]; l0 = slx[i]; l1 = sly[i]; l2 = slz[i]; }
    for (; bt < n_batches; bt += n_waves) {
        const int bn = bt + n_waves;
        PlanePair pn{};
        float n0 = 0.f, n1 = 0.f, n2 = 0.f;
        if (bn < n_batches && bn * 64 + lane < N) { const int i = bn * 64 + lane; pn = pairs[i]; n0 = slx[i]; n1 = sly[i]; n2 = slz[i]; }
        double v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = 0.0;
        if (pp.valid) {   // (a lane past the end holds a zero pairing: valid = 0)
            const double l[3] = {l0, l1, l2};
#pragma unroll
            for (int r = 0; r < 3; ++r) {
#pragma unroll
                for (int c = 0; c < 3; ++c) v[3 * r + c] = pp.n[r] * l[c];
                v[9 + r] = pp.n[r];
            }
            v[12] = pp.n[0] * pp.c[0] + pp.n[1] * pp.c[1] + pp.n[2] * pp.c[2];
            v[13] = 1.0;
        }
        pp = pn; l0 = n0; l1 = n1; l2 = n2;
#pragma unroll
        for (int k = 0; k < 16; k += 2) *reinterpret_cast<double2*>(&sv[lane * kPhiStride + k]) = double2{v[k], v[k + 1]};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const double a = sv[(4 * c + (lane >> 4)) * kPhiStride + (lane & 15)];
            D = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, D, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();   // the vectors are rewritten by the next batch
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) s_d[wave][((lane >> 4) + 4 * r) * 16 + (lane & 15)] = D[r];
    __syncthreads();
    const int row = threadIdx.x >> 4, col = threadIdx.x & 15;
    const double t = ((s_d[0][threadIdx.x] + s_d[1][threadIdx.x]) + s_d[2][threadIdx.x]) + s_d[3][threadIdx.x];
    double* out = partials + (size_t)bx * kNAccPlane;
    if (row <= col && col < 12) out[row * 12 - row * (row - 1) / 2 + (col - row)] = t;   // upper triangle, row-major (a <= b)
    else if (col == 12 && row < 12) out[78 + row] = t;
    else if (col == 12 && row == 12) out[90] = t;
    else if (col == 13 && row == 13) out[91] = t;
}

__global__ __launch_bounds__(256) void k_accumulate_planes_mfma(const float* __restrict__ slx, const float* __restrict__ sly,
                                                                const float* __restrict__ slz, const PlanePair* __restrict__ pairs,
                                                                int N, double* __restrict__ partials)
{
    accumulate_planes_mfma_rows(slx, sly, slz, pairs, N, (int)blockIdx.x, (int)gridDim.x, partials);
}

// K problems in one launch (grid = (max rows, K)): per problem the SAME partition as its own k_accumulate_planes_mfma launch
struct PlaneAccBatch {
    const float* slx[kKnnMaxBatch];
    const float* sly[kKnnMaxBatch];
    const float* slz[kKnnMaxBatch];
    const PlanePair* pairs[kKnnMaxBatch];
    double* partials[kKnnMaxBatch];
    int N[kKnnMaxBatch], nblocks[kKnnMaxBatch], slot[kKnnMaxBatch];
};
__global__ __launch_bounds__(256) void k_accumulate_planes_mfma_batch(const PlaneAccBatch b)
{
    const int y = (int)blockIdx.y, nb = b.nblocks[y];
    if ((int)blockIdx.x >= nb) return;
    accumulate_planes_mfma_rows(b.slx[y], b.sly[y], b.slz[y], b.pairs[y], b.N[y], (int)blockIdx.x, nb, b.partials[y]);
}

// fixed-order sum of [nblocks][n] partial rows (n even, <= 128): 22 slices of rows per accumulator pair with the loads of a
// slice independent of each other, then the 22 slice sums in order.  Deterministic for a given nblocks.
// host_out (pinned, may be null): the n sums + acc[n] written there too, then the sequence number in slot n + 2 -- the
// hand-over k_publish would otherwise make in a launch of its own.
__device__ __forceinline__ void reduce_rows_wide(const double* __restrict__ partials, int nblocks, int n,
                                                 double* __restrict__ acc, const unsigned int* __restrict__ counters,
                                                 double* __restrict__ host_out, unsigned long long seq)
{
    // acc[n] = items of the plane matcher whose neighbour lists changed in this iteration (its next launch picks
    // the counting or the insertion flavour from it): counters[0] (insertion launch) + counters[2] (queued by verify)
    if (counters && threadIdx.x == 0) {
        unsigned int ch = counters[0] + counters[2];
        unsigned int* slots = const_cast<unsigned int*>(counters) + 16 + 1;  // word 1 of the 2 x kQueues queue lines (queues start 8 doubles on)
        for (int q = 0; q < 2 * kQueues; ++q) { ch += slots[q * kQueueStride]; slots[q * kQueueStride] = 0u; }
        acc[n] = (double)ch;
        if (host_out) host_out[n] = acc[n];
        // ... and leave the matcher's counters (kept / redo / ticket, then the work-queue counters) zero for its next launch
        unsigned int* c = const_cast<unsigned int*>(counters);
        c[0] = c[1] = c[2] = c[3] = 0u;
    }
    if (counters && threadIdx.x >= 32 && threadIdx.x < 32 + 2 * kQueues)
        const_cast<unsigned int*>(counters)[16 + (threadIdx.x - 32) * kQueueStride] = 0u;  // (queues start 8 doubles on)
    // 22 slices of rows x 46 column PAIRS (n = 92: a thread adds two accumulators per 16-byte load): 1012 of the 1024 threads busy and
    // 23 rows per thread at 512 rows, where 8 slices x 128 columns left a quarter of the block idle and 64 rows per thread
    constexpr int kSl = 22;
    __shared__ double sm[kSl][128];
    const int half = n >> 1;                       // (n is even: 92)
    const int kp = threadIdx.x % half, sl = threadIdx.x / half;
    double v0 = 0.0, v1 = 0.0;
    if (sl < kSl) {
        int b = sl;
        for (; b + 15 * kSl < nblocks; b += 16 * kSl) {  // sixteen rows in flight
            double2 a[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) a[u] = *reinterpret_cast<const double2*>(partials + (size_t)(b + kSl * u) * n + 2 * kp);
#pragma unroll
            for (int u = 0; u < 16; ++u) { v0 += a[u].x; v1 += a[u].y; }
        }
        for (; b < nblocks; b += kSl) {
            const double2 a = *reinterpret_cast<const double2*>(partials + (size_t)b * n + 2 * kp);
            v0 += a.x; v1 += a.y;
        }
        sm[sl][2 * kp] = v0; sm[sl][2 * kp + 1] = v1;
    }
    __syncthreads();
    const int k = threadIdx.x;
    if (k < n) {
        double t = 0.0;
        for (int s2 = 0; s2 < kSl; ++s2) t += sm[s2][k];
        acc[k] = t;
        if (host_out) host_out[k] = t;
    }
    if (host_out) {  // publish: data first, then the sequence number the host spins on
        if (threadIdx.x < 128) __threadfence_system();  // (the waves that wrote to the host block: see reduce_rows)
        __syncthreads();
        if (threadIdx.x == 0) {
            reinterpret_cast<volatile unsigned long long*>(host_out)[n + 2] = seq;
            __threadfence_system();
        }
    }
}

__global__ __launch_bounds__(1024) void k_reduce_rows(const double* __restrict__ partials, int nblocks, int n,
                                                      double* __restrict__ acc, const unsigned int* __restrict__ counters,
                                                      double* __restrict__ host_out, unsigned long long seq)
{
    reduce_rows_wide(partials, nblocks, n, acc, counters, host_out, seq);
}

// K problems: block y sums problem y's rows (the same order as its own k_reduce_rows launch) into acc + kPlaneAccStride * slot and
// publishes them at host_out + kPlaneAccStride * slot (sequence flag in slot n + 2 of that stride)
constexpr int kPlaneAccStride = 96;
__global__ __launch_bounds__(1024) void k_reduce_rows_batch(const PlaneAccBatch b, int n, double* __restrict__ acc,
                                                            double* __restrict__ host_out, unsigned long long seq)
{
    const int y = (int)blockIdx.x, sl = b.slot[y];
    reduce_rows_wide(b.partials[y], b.nblocks[y], n, acc + (size_t)kPlaneAccStride * sl, nullptr, host_out + (size_t)kPlaneAccStride * sl, seq);
}

// plane pairing in sorted query order -> original order (tests / callers that want the pairing)
__global__ __launch_bounds__(256) void k_unpermute_planes(const int* __restrict__ qperm, const PlanePair* __restrict__ in,
                                                          KnnSeeds seeds, int K, int N, PlanePair* __restrict__ out, int* __restrict__ knn_idx)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int o = qperm[i];
    out[o] = in[i];
    if (knn_idx && seeds.pos)
        for (int j = 0; j < K; ++j) {
            const size_t at = (size_t)j * seeds.stride + (size_t)i;   // (stored lists hold one entry more than knn, and may go on beyond the gate)
            knn_idx[(size_t)o * K + j] = (j < in[i].n_neigh && seeds.pos[at] >= 0) ? (int)seeds.oidx[at] : -1;
        }
}

#endif  // MOLA_ICP_PLANE_TYPES_ONLY

}  // namespace mola_icp_amd
