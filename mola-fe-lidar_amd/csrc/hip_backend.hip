// hip_backend.hip -- gfx950 kernels + HipWorkspace (see hip_backend.hpp).
//
// Hot path rows: a7 nearest-neighbour matcher, a8 weighted
// centroid/covariance accumulation.  The reference reaches both through
// mp2p_icp::ICP::align() at src/LidarOdometry.cpp:869-871 (CPU kd-tree +
// serial sums); here they are brute-force tiled kernels over HBM-resident SoA
// clouds.
//
// Numeric contract (DESIGN.md "numeric contract"; the CPU checker restates it, so NN indices compare
// bit-exactly; this file is compiled with -ffp-contract=off so only the
// explicit fmaf() calls fuse):
//   q  = fmaf-chain R*l+t in fp32;  d2 = fmaf(dz,dz,fmaf(dy,dy,dx*dx));
//   NN = argmin d2, ties -> lowest map index; kept iff d2 < thr2.
#include "hip_backend.hpp"

#include <cmath>
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

namespace mola_icp_amd {

#define HIPCHK(expr)                                                                                          \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess)                                                                                 \
            return fail(e_ == hipErrorOutOfMemory ? MOLA_ICP_E_OOM : MOLA_ICP_E_HIP,                          \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                                   \
    } while (0)

// ------------------------------------------------------------------ device code

struct PoseF {
    float R[9];
    float t[3];
};

__device__ __forceinline__ void xform(const PoseF& P, float lx, float ly, float lz, float& qx, float& qy, float& qz)
{
    float a;
    a = fmaf(P.R[0], lx, P.t[0]); a = fmaf(P.R[1], ly, a); qx = fmaf(P.R[2], lz, a);
    a = fmaf(P.R[3], lx, P.t[1]); a = fmaf(P.R[4], ly, a); qy = fmaf(P.R[5], lz, a);
    a = fmaf(P.R[6], lx, P.t[2]); a = fmaf(P.R[7], ly, a); qz = fmaf(P.R[8], lz, a);
}

__device__ __forceinline__ float dist2(float qx, float qy, float qz, float gx, float gy, float gz)
{
    const float dx = qx - gx, dy = qy - gy, dz = qz - gz;
    return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

// largest float <= x (HIP's __double2float_rd is not relied upon)
__device__ __forceinline__ float down_f32(double x)
{
    float f = (float)x;
    if ((double)f > x) f = __uint_as_float(f > 0.f ? __float_as_uint(f) - 1u : (f < 0.f ? __float_as_uint(f) + 1u : 0x80000001u));
    return f;
}

constexpr float kPadCoord = 1.0e18f;  // padding map points: d2 ~ 3e36, finite, never the minimum

// ---- NN matcher, exact VALU form -------------------------------------------------
// Block = 256 threads, each thread owns QPT queries in registers (coalesced SoA
// loads).  The map streams HBM -> LDS in SoA tiles of TM points; every lane reads
// the same LDS address (broadcast, conflict-free) as ds_read_b128 of 4 points.
// Per pair: 3 sub + mul + 2 fma + ~1 min; the argmin is tracked per 8-point chunk
// (first chunk that lowers the minimum) and resolved to the exact lowest index
// after the sweep by re-evaluating that chunk -- bit-identical arithmetic.
template <int QPT, int TM>
__global__ __launch_bounds__(256) void k_nn_valu(const float* __restrict__ lx, const float* __restrict__ ly,
                                                 const float* __restrict__ lz, int N, const float* __restrict__ gx,
                                                 const float* __restrict__ gy, const float* __restrict__ gz, int M,
                                                 PoseF P, float thr2, int* __restrict__ out_idx,
                                                 float* __restrict__ out_d2, unsigned int* __restrict__ kept_counter)
{
    __shared__ __attribute__((aligned(16))) float sx[TM];
    __shared__ __attribute__((aligned(16))) float sy[TM];
    __shared__ __attribute__((aligned(16))) float sz[TM];
    const int tid = threadIdx.x;
    const int qbase = blockIdx.x * (256 * QPT);

    float qx[QPT], qy[QPT], qz[QPT], best[QPT];
    int bchunk[QPT];
#pragma unroll
    for (int k = 0; k < QPT; ++k) {
        const int i = qbase + k * 256 + tid;
        float x = 0.f, y = 0.f, z = 0.f;
        if (i < N) { x = lx[i]; y = ly[i]; z = lz[i]; }
        xform(P, x, y, z, qx[k], qy[k], qz[k]);
        best[k] = thr2;  // gate: only d2 < thr2 can ever be kept
        bchunk[k] = -1;
    }

    for (int tile0 = 0; tile0 < M; tile0 += TM) {
        __syncthreads();
#pragma unroll
        for (int j = tid; j < TM; j += 256) {
            const int gj = tile0 + j;
            const bool in = gj < M;
            sx[j] = in ? gx[gj] : kPadCoord;
            sy[j] = in ? gy[gj] : kPadCoord;
            sz[j] = in ? gz[gj] : kPadCoord;
        }
        __syncthreads();
        const int lim = min(TM, M - tile0);
        for (int c = 0; c < lim; c += 8) {
            const float4 xa = *reinterpret_cast<const float4*>(&sx[c]);
            const float4 xb = *reinterpret_cast<const float4*>(&sx[c + 4]);
            const float4 ya = *reinterpret_cast<const float4*>(&sy[c]);
            const float4 yb = *reinterpret_cast<const float4*>(&sy[c + 4]);
            const float4 za = *reinterpret_cast<const float4*>(&sz[c]);
            const float4 zb = *reinterpret_cast<const float4*>(&sz[c + 4]);
#pragma unroll
            for (int k = 0; k < QPT; ++k) {
                const float d0 = dist2(qx[k], qy[k], qz[k], xa.x, ya.x, za.x);
                const float d1 = dist2(qx[k], qy[k], qz[k], xa.y, ya.y, za.y);
                const float d2 = dist2(qx[k], qy[k], qz[k], xa.z, ya.z, za.z);
                const float d3 = dist2(qx[k], qy[k], qz[k], xa.w, ya.w, za.w);
                const float d4 = dist2(qx[k], qy[k], qz[k], xb.x, yb.x, zb.x);
                const float d5 = dist2(qx[k], qy[k], qz[k], xb.y, yb.y, zb.y);
                const float d6 = dist2(qx[k], qy[k], qz[k], xb.z, yb.z, zb.z);
                const float d7 = dist2(qx[k], qy[k], qz[k], xb.w, yb.w, zb.w);
                const float m = fminf(fminf(fminf(d0, d1), fminf(d2, d3)), fminf(fminf(d4, d5), fminf(d6, d7)));
                if (m < best[k]) { best[k] = m; bchunk[k] = tile0 + c; }
            }
        }
    }

    unsigned int kept = 0;
#pragma unroll
    for (int k = 0; k < QPT; ++k) {
        const int i = qbase + k * 256 + tid;
        int idx = -1;
        if (bchunk[k] >= 0) {
            for (int r = 7; r >= 0; --r) {  // descending: the lowest matching index wins
                const int gj = bchunk[k] + r;
                if (gj < M) {
                    const float d = dist2(qx[k], qy[k], qz[k], gx[gj], gy[gj], gz[gj]);
                    if (d == best[k]) idx = gj;
                }
            }
        }
        if (i < N) {
            out_idx[i] = idx;
            out_d2[i] = best[k];
            kept += (idx >= 0);
        }
    }
    // one atomic per wave
    for (int off = 32; off > 0; off >>= 1) kept += __shfl_down(kept, off);
    if ((tid & 63) == 0 && kept) atomicAdd(kept_counter, kept);
}

// ---- NN matcher, MFMA filter + exact re-evaluation ------------------------------------
// The N x M x 3 distance contraction in expanded form,
//     e(q,m) = |m'|^2 - 2 q'.m'  = [ -2m'x, -2m'y, -2m'z, |m'|^2 ] . [ q'x, q'y, q'z, 1 ]      (K = 4)
// (primes: coordinates relative to the map's bounding-box centre c) is exactly one
// v_mfma_f32_16x16x4_f32 per 16 map points x 16 queries.  e + |q'|^2 approximates d2 only to
// ~1e-2 m^2 at 100 m range (fp32 cancellation), so the MFMA is used as a FILTER with a
// rigorous error bound (DESIGN.md "MFMA filter bound"):
//     | (e_mfma + |q'|^2_fl) - d2_contract |  <=  u*(18.6|q'|^2 + 18.6|m'|^2 + 6.2 g^2),  u = 2^-24, g = gate
// The |m'|^2 share is folded into the A operand (k=3 row holds |m'|^2 (1 - 20u)), the rest into
// the accumulator input C = -(best - |q'|^2 + 20u|q'|^2 + 8u g^2), so an output <= 0 means
// "d2 may be <= the query's current best".  Only those survivors (a handful per query over the
// whole map) are re-evaluated with the exact direct-difference contract on the original
// coordinates (staged in LDS beside the image) -- the result is bit-identical to k_nn_valu /
// the CPU checker, including the lowest-index tie rule.  The best is warm-started from the
// previous iteration's pairing (an exact candidate), which removes most survivors.
//
// Layout: A = map tile, lane l holds A[i = l&15][k = l>>4]; the map image in HBM/LDS is
// [tile][k][16] so that is word (tile*64 + l): one conflict-free ds_read_b32 feeds QT MFMAs.
// B = 16 queries, lane l holds B[k = l>>4][j = l&15] (register-resident for the whole sweep).
// D: lane l, reg r = pair (map row (l>>4)*4 + r, query l&15): each lane tracks the best of
// "its" rows for query l&15; the four lane groups are merged once at the end.
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr float kMapPadNorm = 1.0e30f;          // |m'|^2 of padding rows: e = 1e30, never a survivor
constexpr float kUnitRoundoff = 5.9604645e-8f;  // u = 2^-24
constexpr float kFoldCoef = 20.0f * kUnitRoundoff;  // >= 18.6u + the rounding of the folding itself
constexpr float kGateCoef = 8.0f * kUnitRoundoff;   // >= 6.2u

// accumulator input for a query with squared norm qq (centred) and current best d2
__device__ __forceinline__ float filter_c(float qq, float best, float gate2)
{
    // -(best - qq + eps_q), rounded towards "more survivors"
    return (qq - best) - (kFoldCoef * qq + kGateCoef * gate2) * 1.0001f - 1e-30f;
}

struct MapFrame {
    float cx, cy, cz;  // bounding-box centre of the map (fp32)
    float radius;      // >= max |m - c| over the map
};

// Work decomposition: the sweep is cut into ITEMS = (group of QT*16 queries) x (map segment).
// Persistent waves pull items from an atomic queue (segment-major, so the waves running at any
// time read the same ~2 MiB slice of the map image: it stays in every XCD's L2).  Each wave is
// autonomous -- no block barrier anywhere: it streams the A operand straight from L2 through a
// 4-deep register prefetch ring (one coalesced 256-B load per 16 map points; ~4 B/clk/CU, far
// below what L2 delivers) and keeps its queries, thresholds and running best in registers.
// Per-segment results are merged by k_nn_merge (lexicographic (d2,index) minimum).
template <int QT>
__global__ __launch_bounds__(256, 2) void k_nn_mfma(const float* __restrict__ lx, const float* __restrict__ ly,
                                                    const float* __restrict__ lz, int N,
                                                    const float* __restrict__ gx, const float* __restrict__ gy,
                                                    const float* __restrict__ gz, int M,
                                                    const float* __restrict__ map_img, int n_tiles, int seg_tiles,
                                                    int n_segs, int n_qgroups, MapFrame F, PoseF P, float thr2,
                                                    const int* __restrict__ seed_idx, int* __restrict__ seg_idx,
                                                    float* __restrict__ seg_d2, unsigned int* __restrict__ queue,
                                                    unsigned long long* __restrict__ dbg_stats)
{
    __shared__ float4 s_q[4 * QT * 16];  // per wave: (qx,qy,qz,|q'|^2) of its queries, for the exact re-evaluation
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, grp = lane >> 4;
    float4* my_q = s_q + wave * (QT * 16);
    const int n_items = n_qgroups * n_segs;

    for (;;) {
        int item = 0;
        if (lane == 0) item = (int)atomicAdd(queue, 1u);
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= n_items) break;
        const int seg = item / n_qgroups, qg = item - seg * n_qgroups;
        const int q0 = qg * (QT * 16);
        const int t_beg = seg * seg_tiles, t_end = min(t_beg + seg_tiles, n_tiles);  // multiples of 4 tiles

        float B[QT], best[QT];
        int bidx[QT];
        f32x4 C[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            const int i = q0 + t * 16 + col;
            float qx = 0.f, qy = 0.f, qz = 0.f, cthr = kMapPadNorm;  // padding query: D = e + 1e30 > 0 always
            float bx = 0.f, by = 0.f, bz = 0.f, qq = 0.f;
            best[t] = thr2;  // gate: only d2 < thr2 can ever be kept
            bidx[t] = -1;
            if (i < N) {
                xform(P, lx[i], ly[i], lz[i], qx, qy, qz);
                bx = qx - F.cx; by = qy - F.cy; bz = qz - F.cz;
                qq = fmaf(bz, bz, fmaf(by, by, bx * bx));
                if (seed_idx) {  // warm start: last iteration's neighbour is an exact candidate
                    const int j = seed_idx[i];
                    if (j >= 0 && j < M) {
                        const float d = dist2(qx, qy, qz, gx[j], gy[j], gz[j]);
                        if (d < thr2) { best[t] = d; bidx[t] = j; }
                    }
                }
                cthr = filter_c(qq, best[t], thr2);
            }
            B[t] = grp == 0 ? bx : (grp == 1 ? by : (grp == 2 ? bz : 1.0f));
            C[t] = f32x4{cthr, cthr, cthr, cthr};
            if (grp == 0) my_q[t * 16 + col] = make_float4(qx, qy, qz, qq);
        }
        // my_q is private to this wave: a wave-level fence is all the ordering it needs
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        const float* img = map_img + (size_t)t_beg * 64 + lane;  // word (tile*64 + lane) = A[lane&15][lane>>4]
        // prefetch ring (the image carries 4 padding tiles past n_tiles, so these never run off the end)
        float a0 = img[0], a1 = img[64], a2 = img[128], a3 = img[192];
        f32x4 Dp[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t) Dp[t] = f32x4{1.f, 1.f, 1.f, 1.f};  // nothing pending before the first step

        // consume(): reduce/test the PREVIOUS step's accumulators while this step's MFMAs run
#define MOLA_NN_CONSUME(TILE)                                                                                     \
    {                                                                                                             \
        int r = min(min(__float_as_int(Dp[0][0]), __float_as_int(Dp[0][1])),                                      \
                    min(__float_as_int(Dp[0][2]), __float_as_int(Dp[0][3])));                                     \
        _Pragma("unroll") for (int t = 1; t < QT; ++t) {                                                          \
            r = min(min(r, __float_as_int(Dp[t][0])), __float_as_int(Dp[t][1]));                                  \
            r = min(min(r, __float_as_int(Dp[t][2])), __float_as_int(Dp[t][3]));                                  \
        }                                                                                                         \
        if (__any(r <= 0)) {                                                                                      \
            const int row0 = (TILE)*16 + grp * 4;                                                                 \
            if (dbg_stats && lane == 0) atomicAdd(&dbg_stats[0], 1ull);                                           \
            _Pragma("unroll") for (int t = 0; t < QT; ++t) {                                                      \
                const int mt = min(min(__float_as_int(Dp[t][0]), __float_as_int(Dp[t][1])),                       \
                                   min(__float_as_int(Dp[t][2]), __float_as_int(Dp[t][3])));                      \
                if (__any(mt <= 0)) {                                                                             \
                    const float4 q = my_q[t * 16 + col];                                                          \
                    _Pragma("unroll") for (int rr = 0; rr < 4; ++rr) {                                            \
                        if (Dp[t][rr] <= 0.0f) {                                                                  \
                            if (dbg_stats) atomicAdd(&dbg_stats[1], 1ull);                                        \
                            const int j = row0 + rr; /* < M: padding rows never survive */                        \
                            const float d = dist2(q.x, q.y, q.z, gx[j], gy[j], gz[j]);                            \
                            if (d < best[t] || (d == best[t] && j < bidx[t])) { best[t] = d; bidx[t] = j; }       \
                        }                                                                                         \
                    }                                                                                             \
                    float nb = best[t];                                                                           \
                    nb = fminf(nb, __shfl_xor(nb, 16));                                                           \
                    nb = fminf(nb, __shfl_xor(nb, 32));                                                           \
                    if (q0 + t * 16 + col < N) {                                                                  \
                        const float cthr = filter_c(q.w, nb, thr2);                                               \
                        C[t] = f32x4{cthr, cthr, cthr, cthr};                                                     \
                    }                                                                                             \
                }                                                                                                 \
            }                                                                                                     \
        }                                                                                                         \
    }
#define MOLA_NN_STEP(AREG, TILE, NEXT_OFF)                                                                        \
    {                                                                                                             \
        f32x4 Dn[QT];                                                                                             \
        const float a_cur = AREG;                                                                                 \
        AREG = img[(NEXT_OFF)];                                                                                   \
        _Pragma("unroll") for (int t = 0; t < QT; ++t)                                                            \
            Dn[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur, B[t], C[t], 0, 0, 0);                             \
        MOLA_NN_CONSUME((TILE)-1)                                                                                 \
        _Pragma("unroll") for (int t = 0; t < QT; ++t) Dp[t] = Dn[t];                                             \
    }
        for (int tile = t_beg; tile < t_end; tile += 4) {
            MOLA_NN_STEP(a0, tile, 4 * 64)
            MOLA_NN_STEP(a1, tile + 1, 5 * 64)
            MOLA_NN_STEP(a2, tile + 2, 6 * 64)
            MOLA_NN_STEP(a3, tile + 3, 7 * 64)
            img += 4 * 64;
        }
        MOLA_NN_CONSUME(t_end - 1)
#undef MOLA_NN_STEP
#undef MOLA_NN_CONSUME

        // merge the four lane groups: lexicographic (d2, index) minimum -> lowest index on ties
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            float d = best[t];
            int j = bidx[t] < 0 ? 0x7fffffff : bidx[t];
#pragma unroll
            for (int off = 16; off <= 32; off <<= 1) {
                const float od = __shfl_xor(d, off);
                const int oj = __shfl_xor(j, off);
                if (od < d || (od == d && oj < j)) { d = od; j = oj; }
            }
            const int i = q0 + t * 16 + col;
            if (grp == 0 && i < N) {
                seg_idx[(size_t)seg * N + i] = j == 0x7fffffff ? -1 : j;
                seg_d2[(size_t)seg * N + i] = d;
            }
        }
        __builtin_amdgcn_wave_barrier();  // my_q is rewritten by the next item
    }
}

// per-segment results -> the pairing: lexicographic (d2, index) minimum over the segments
__global__ __launch_bounds__(256) void k_nn_merge(const int* __restrict__ seg_idx, const float* __restrict__ seg_d2,
                                                  int n_segs, int N, int* __restrict__ out_idx,
                                                  float* __restrict__ out_d2, unsigned int* __restrict__ kept_counter)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    unsigned int kept = 0;
    if (i < N) {
        float d = seg_d2[i];
        int j = seg_idx[i] < 0 ? 0x7fffffff : seg_idx[i];
        for (int s = 1; s < n_segs; ++s) {
            const float od = seg_d2[(size_t)s * N + i];
            const int oj = seg_idx[(size_t)s * N + i] < 0 ? 0x7fffffff : seg_idx[(size_t)s * N + i];
            if (od < d || (od == d && oj < j)) { d = od; j = oj; }
        }
        const int idx = j == 0x7fffffff ? -1 : j;
        out_idx[i] = idx;
        out_d2[i] = d;
        kept = idx >= 0;
    }
    for (int off = 32; off > 0; off >>= 1) kept += __shfl_down(kept, off);
    if ((threadIdx.x & 63) == 0 && kept) atomicAdd(kept_counter, kept);
}

// ---- tiled matcher: exact brute force over the map tiles a wave's queries can reach ---------
// Both clouds are put in Morton order once (map: per map; local cloud: per cloud).  The map is cut
// into TILES of 32 consecutive points with an axis-aligned box, 64 tiles form a super-tile.  A
// wave owns 128 consecutive (hence spatially compact) queries, two per lane.  Per iteration:
//   1. transform the queries, warm-start each best from the previous iteration's neighbour,
//      give each query the box [q - r, q + r], r = sqrt(best) (rounded up), and reduce the
//      boxes to one wave box;
//   2. cull: lanes test 64 super-tile boxes at a time against the wave box (ballot), then the 64
//      tiles of each hit super-tile.  A tile whose box misses the wave box cannot hold, for any of
//      the wave's queries, a point with d2 <= best -- skipping it is exact;
//   3. every surviving tile is staged in LDS (two tiles = 64 points per pass) and ALL 128 queries
//      are evaluated against ALL its points with the exact contract: tiled brute force, per-lane
//      argmin on the packed key (d2 bits << 32 | ORIGINAL map index), i.e. lexicographic
//      (d2, lowest index) -- bit-identical to the untiled kernels.
// No tree, no per-query traversal, no data-dependent recursion: two flat box scans and dense
// 128 x 64 tiles.
constexpr int kTileG = 32;     // map points per tile
constexpr int kSuper = 64;     // tiles per super-tile
constexpr int kQPW = 128;      // queries per wave
#ifndef MOLA_VAR_GROUP
#define MOLA_VAR_GROUP 8
#endif
constexpr int kGroup = MOLA_VAR_GROUP;      // fast sweep: points per bookkeeping group

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

#ifndef MOLA_VAR_LDSBOX_KB
#define MOLA_VAR_LDSBOX_KB 40
#endif
constexpr size_t kMaxLdsBoxBytes = MOLA_VAR_LDSBOX_KB * 1024;  // upper box levels kept in LDS up to this size (~3.4M map points)
constexpr size_t kDbgItems = 1u << 17;  // MOLA_ICP_DEBUG_STATS: per-item records for clouds up to 16M points
constexpr int kQueues = 8, kQueueStride = 32;  // work-queue counters, one 128-byte line each
constexpr int kMaxList = 64;   // super-tiles collected before their tiles are streamed

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
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            w.lo[a] = fminf(w.lo[a], __shfl_xor(w.lo[a], off));
            w.hi[a] = fmaxf(w.hi[a], __shfl_xor(w.hi[a], off));
        }
    }
    unsigned long long n_staged = 0;

    // Can the box (m0..m5 = min xyz, max xyz; wave-uniform values) hold a point with d2 <= bound2 for ANY query of
    // the wave?  Per query: squared distance to the box, computed with the contract's own operation sequence on
    // the per-axis gaps.  Rounding is monotone, so for every point p inside the box gap_a <= |q_a - p_a| after
    // rounding, hence box_d2 <= d2_contract(q, p) EXACTLY as computed -- no margin needed, and bound2 is read
    // live: as a query's best shrinks during the sweep, later boxes are tested against the tighter value.
    // Padding lanes carry bound2 < 0 and reach nothing; empty boxes (+inf, -inf) give inf.
    auto any_reach = [&](float m0, float m1, float m2, float m3, float m4, float m5) -> bool {
        if constexpr (QPL == 2) {  // both queries of the lane per packed instruction
            const v2f s_qx = {qx[0], qx[1]}, s_qy = {qy[0], qy[1]}, s_qz = {qz[0], qz[1]};
            const v2f zero = {0.f, 0.f};
            const v2f ax = __builtin_elementwise_max(__builtin_elementwise_max(m0 - s_qx, s_qx - m3), zero);
            const v2f ay = __builtin_elementwise_max(__builtin_elementwise_max(m1 - s_qy, s_qy - m4), zero);
            const v2f az = __builtin_elementwise_max(__builtin_elementwise_max(m2 - s_qz, s_qz - m5), zero);
            const v2f D = __builtin_elementwise_fma(az, az, __builtin_elementwise_fma(ay, ay, ax * ax));
            return __any(D.x <= bound2[0] || D.y <= bound2[1]);
        } else {
            const float ax = fmaxf(fmaxf(m0 - qx[0], qx[0] - m3), 0.f);
            const float ay = fmaxf(fmaxf(m1 - qy[0], qy[0] - m4), 0.f);
            const float az = fmaxf(fmaxf(m2 - qz[0], qz[0] - m5), 0.f);
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
            unsigned long long tmask = 0;
            const unsigned long long tb1 = prof ? __builtin_amdgcn_s_memtime() : 0ull;
            while (cand) {
                const int t = __builtin_ctzll(cand);
                cand &= cand - 1;
                if (prof) p_tiles += 1;
                if (any_reach(bcast_lane(b0, t), bcast_lane(b1, t), bcast_lane(b2, t), bcast_lane(b3, t),
                              bcast_lane(b4, t), bcast_lane(b5, t)))
                    tmask |= 1ull << t;
            }
            if (prof) { const unsigned long long tb2 = __builtin_amdgcn_s_memtime(); p_boxwait += tb1 - tb0; p_tiletest += tb2 - tb1; }
            while (tmask) {
                const int t0 = Sc * kSuper + __builtin_ctzll(tmask);
                tmask &= tmask - 1;
                int t1 = -1;
                if (tmask) { t1 = Sc * kSuper + __builtin_ctzll(tmask); tmask &= tmask - 1; }
                if (pend_a < 0) {  // nothing in flight yet: just issue this pair's loads
                    pend_a = t0; pend_b = t1;
                    load_pair(t0, t1);
                } else {
                    compute_pending(t0, t1);
                }
            }
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

// reach of a query whose current best squared distance is `best`: any m with d2_contract <= best lies inside
// [q - r, q + r] per axis (sqrt rounded up, plus 2 ulp of the largest coordinate)
__device__ __forceinline__ float reach_of(float best, float qx, float qy, float qz)
{
    const float cmax = fmaxf(fabsf(qx), fmaxf(fabsf(qy), fabsf(qz)));
    return sqrtf(best * 1.000002f) * 1.00001f + cmax * 2.4e-7f + 1e-30f;
}

// Work queue of the persistent waves.  Same-address atomics serialise device-wide (measured: ~13 ns each; one
// more atomic per item cost 20% of the kernel, the 3072-deep burst of first pops 40 us), so
//  - the first entry of every wave is its own index: no atomics at kernel start;
//  - the entries after those are dealt round-robin to kQueues counters on separate cache lines; a wave pops from
//    the counter of its XCD (blockIdx & 7) and moves on to the next counter when that one runs dry.
// Callers keep the next entry's pop in flight while the current item is processed.
struct WaveQueue {
    unsigned int* q;
    int lane, n_waves, tried;
    __device__ __forceinline__ WaveQueue(unsigned int* queue, int lane_) : q(queue), lane(lane_), n_waves((int)gridDim.x * 4), tried(0) {}
    __device__ __forceinline__ int first() const { return (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6); }
    __device__ __forceinline__ int pop()  // the next entry (meaningful in lane 0), still in flight
    {
        const int c = ((int)blockIdx.x + tried) & (kQueues - 1);
        int r = 0;
        if (lane == 0) r = n_waves + c + kQueues * (int)atomicAdd(q + c * kQueueStride, 1u);
        return r;
    }
    __device__ __forceinline__ int settle(int raw, int n_items)  // raw = readfirstlane(pop()): past the end -> other counters
    {
        while (raw >= n_items && ++tried < kQueues) raw = __builtin_amdgcn_readfirstlane(pop());
        return raw;
    }
};

// EXACT = false: the fast sweep.  Per (query, 4-point chunk) only the chunk minimum is compared with the
//   running best (~6 VALU ops per pair); the winning chunk is re-evaluated once at the end to recover the
//   exact point and the lowest-original-index rule inside it.  If an EQUAL minimum showed up in a different
//   chunk (exact ties: duplicate points, lattices) the item is queued for the exact pass.
// EXACT = true : the exact-key sweep over the queued items: per-pair argmin on the packed key
//   (d2 bits << 32 | original index), i.e. the full lexicographic rule (~10 ops per pair).
template <bool EXACT, int QPL>
__global__ __launch_bounds__(256, 3) void k_nn_tiled(const float* __restrict__ slx, const float* __restrict__ sly,
                                                  const float* __restrict__ slz, int N, TiledMap mp, PoseF P, float thr2,
                                                  int use_seed, int* __restrict__ pos_s, int* __restrict__ idx_s,
                                                  float* __restrict__ d2_s, const int* __restrict__ item_order,
                                                  unsigned int* __restrict__ item_cost, unsigned int* __restrict__ queue,
                                                  unsigned int* __restrict__ redo_count, int* __restrict__ redo_list,
                                                  unsigned long long* __restrict__ staged_total,
                                                  unsigned long long* __restrict__ dbg_stats, int lds_boxes,
                                                  unsigned long long* __restrict__ wave_times /*diagnostics, usually null*/)
{
    __shared__ __attribute__((aligned(16))) float s_m[4][4][64];  // per wave: x, y, z, original index of 64 staged points
    __shared__ int s_list[4][kMaxList];
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];  // the upper box levels, if they fit
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float(*sm)[64] = s_m[wave];
    int* slist = s_list[wave];
    if (EXACT && *redo_count == 0u) return;  // the usual case: no exact ties in this launch (uniform: before any barrier)
    const lds_f32* lbox = (const lds_f32*)s_dyn;
    if (lds_boxes) load_boxes_to_lds(mp, (lds_f32*)s_dyn);
    constexpr int kQ = 64 * QPL;  // queries per item: QPL per lane (2 for large clouds, 1 when there are few items per wave)
    const int n_items = EXACT ? (int)*redo_count : (N + kQ - 1) / kQ;

    WaveQueue wq(queue, lane);
    auto lookup = [&](int raw) -> int {  // raw is wave-uniform; -1 = past the end
        if (raw >= n_items) return -1;
        if (EXACT) return redo_list[raw];
        return item_order ? item_order[raw] : raw;  // heaviest items of the last launch first
    };
    unsigned long long wave_staged = 0ull;
    const unsigned long long t_wave0 = wave_times ? wall_clock64() : 0ull;  // 100 MHz, the same on every XCD
    unsigned int wave_items = 0u;
    int item = __builtin_amdgcn_readfirstlane(lookup(wq.first()));
    while (item >= 0) {
        ++wave_items;
        const int next_raw_v = wq.pop();
        const unsigned long long t_item0 = __builtin_amdgcn_s_memtime();

        float qx[QPL], qy[QPL], qz[QPL], reach[QPL];
        unsigned long long key[QPL];  // EXACT: packed (d2, original index)
        float best[QPL];              // fast: running minimum
        int bpos[QPL];                // EXACT: sorted position of the best point; fast: of its kGroup-point group
        int tie[QPL] = {};
        // round trip 1: the two queries of the lane and their seeds (clamped indices: every load is unconditional)
        int qi[QPL], js[QPL];
        float lx[QPL], ly[QPL], lz[QPL];
#pragma unroll
        for (int k = 0; k < QPL; ++k) {
            qi[k] = item * kQ + k * 64 + lane;
            if (qi[k] >= N) qi[k] = N;  // padding lane
            const int ic = qi[k] < N ? qi[k] : N - 1;
            lx[k] = slx[ic]; ly[k] = sly[ic]; lz[k] = slz[ic];
            js[k] = use_seed ? pos_s[ic] : -1;
        }
#pragma unroll
        for (int k = 0; k < QPL; ++k) xform(P, lx[k], ly[k], lz[k], qx[k], qy[k], qz[k]);
        // round trip 2: the seeds' coordinates, and the next item's id
        const int next_item_v = lookup(wq.settle(__builtin_amdgcn_readfirstlane(next_raw_v), n_items));
        float gsx[QPL], gsy[QPL], gsz[QPL];
        unsigned int gso[QPL] = {};
#pragma unroll
        for (int k = 0; k < QPL; ++k) {
            const int jc = js[k] >= 0 ? js[k] : 0;
            gsx[k] = mp.sx[jc]; gsy[k] = mp.sy[jc]; gsz[k] = mp.sz[jc];
            if (EXACT) gso[k] = (unsigned int)mp.perm[jc];
        }
#pragma unroll
        for (int k = 0; k < QPL; ++k) {
            key[k] = ((unsigned long long)__float_as_uint(thr2) << 32);  // (gate^2, index 0): "no neighbour" sentinel
            best[k] = thr2;
            bpos[k] = -1;
            const float d = dist2(qx[k], qy[k], qz[k], gsx[k], gsy[k], gsz[k]);
            if (js[k] >= 0 && d < thr2) {  // warm start: last iteration's neighbour is an exact candidate
                best[k] = d;
                bpos[k] = EXACT ? js[k] : (js[k] & ~(kGroup - 1));
                if (EXACT) key[k] = ((unsigned long long)__float_as_uint(d) << 32) | gso[k];
            }
            reach[k] = reach_of(best[k], qx[k], qy[k], qz[k]);
            if (qi[k] >= N) {  // padding lane: reaches nothing, is never written
                qx[k] = qy[k] = qz[k] = 1.0e18f;
                reach[k] = -1.0f;
                best[k] = -1.0f;
                bpos[k] = -1;
            }
        }

        unsigned long long p_stage = 0ull, p_visit = 0ull, p_boxwait = 0ull, p_tiletest = 0ull;
        unsigned int p_supers = 0u, p_entered = 0u, p_tiles = 0u;
        const unsigned long long t_sweep0 = dbg_stats ? __builtin_amdgcn_s_memtime() : 0ull;
        const unsigned long long n_staged = tiled_sweep<QPL, EXACT>(mp, lbox, lds_boxes != 0, slist, lane, sm, qx, qy, qz, reach, best, [&](int nm, int jb0, int jb1) {
            if constexpr (EXACT) {
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
            } else {
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
#pragma unroll
                    for (int k = 0; k < QPL; ++k) {
                        const bool lt = gm[k] < best[k];
                        const int eq = (int)(gm[k] == best[k]) & (int)(gpos != bpos[k]);
                        tie[k] = lt ? 0 : (tie[k] | eq);
                        best[k] = lt ? gm[k] : best[k];
                        bpos[k] = lt ? gpos : bpos[k];
                    }
                }
            }
        }, dbg_stats != nullptr, p_stage, p_visit, p_supers, p_entered, p_tiles, p_boxwait, p_tiletest);
        const unsigned long long t_sweep1 = dbg_stats ? __builtin_amdgcn_s_memtime() : 0ull;

        bool any_tie = false;
        int rpos[QPL], roi[QPL];
        float rd[QPL];
#pragma unroll
        for (int k = 0; k < QPL; ++k) { rpos[k] = -1; roi[k] = -1; rd[k] = thr2; }
        if constexpr (EXACT) {
#pragma unroll
            for (int k = 0; k < QPL; ++k) {
                const float d = __uint_as_float((unsigned int)(key[k] >> 32));
                if (d < thr2) { rd[k] = d; rpos[k] = bpos[k]; roi[k] = (int)(unsigned int)(key[k] & 0xffffffffu); }
            }
        } else {
            // resolve inside the winning group: the point(s) with d2 == best, lowest original index first.
            // One round trip: all loads of both queries are issued before the first use.
            float4 RX[QPL][kGroup / 4], RY[QPL][kGroup / 4], RZ[QPL][kGroup / 4];
            int4 RP[QPL][kGroup / 4];
#pragma unroll
            for (int k = 0; k < QPL; ++k) {
                const int bp = bpos[k] >= 0 ? bpos[k] : 0;
#pragma unroll
                for (int c = 0; c < kGroup / 4; ++c) {
                    RX[k][c] = *reinterpret_cast<const float4*>(mp.sx + bp + 4 * c);
                    RY[k][c] = *reinterpret_cast<const float4*>(mp.sy + bp + 4 * c);
                    RZ[k][c] = *reinterpret_cast<const float4*>(mp.sz + bp + 4 * c);
                    RP[k][c] = *reinterpret_cast<const int4*>(mp.perm + bp + 4 * c);
                }
            }
#pragma unroll
            for (int k = 0; k < QPL; ++k) {
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
                    }
                }
                if (bpos[k] >= 0) {
                    rd[k] = best[k]; rpos[k] = pos; roi[k] = (int)bo;
                    if (pos < 0) tie[k] = 1;  // cannot happen (same arithmetic); be safe: exact pass
                }
            }
        }
#pragma unroll
        for (int k = 0; k < QPL; ++k) {
            if (qi[k] < N) {  // coalesced: the pairing stays in sorted query order
                pos_s[qi[k]] = rpos[k];
                idx_s[qi[k]] = rpos[k] >= 0 ? roi[k] : -1;
                d2_s[qi[k]] = rd[k];
                any_tie |= tie[k] != 0;
            }
        }
        if (!EXACT && __any(any_tie)) {
            if (lane == 0) redo_list[atomicAdd(redo_count, 1u)] = item;
        }
        if (lane == 0) {
            if (!EXACT) {
                // (a deterministic proxy -- staged points -- orders no better than the measured cycles; without any
                // order the kernel is 6 % slower)
                const unsigned long long c = __builtin_amdgcn_s_memtime() - t_item0;
                if (item_cost) item_cost[item] = c > 0xffffffffull ? 0xffffffffu : (unsigned int)c;
            }
            wave_staged += n_staged * QPL;  // executed work in units of 64 (query, point) pairs (one atomic per wave, at exit)
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
        item = __builtin_amdgcn_readfirstlane(next_item_v);
    }
    if (lane == 0 && wave_staged) atomicAdd(staged_total, wave_staged);
    if (wave_times && lane == 0) {  // [start, end, items] per wave
        unsigned long long* w = wave_times + 3 * (size_t)wq.first();
        w[0] = t_wave0; w[1] = wall_clock64(); w[2] = wave_items;
    }
}

// ---- row f3: point-to-plane matcher (mp2p_icp::Matcher_Point2Plane, params/icp-settings-regular.yaml:33-39) ----
// Same tiled sweep; the visitor keeps, per query, the K nearest points as a sorted list ordered by
// (d2, original index).  The reach is the gate (distanceThreshold): only neighbours inside it matter.
// Epilogue per query: the neighbours inside the gate (need >= 3) -> mean + covariance in fp64 -> cyclic
// Jacobi eigen-decomposition -> plane iff e0 <= planeEigenThreshold * e2, normal = eigenvector of e0,
// pairing iff |n.(q - mean)| <= distanceThreshold.  [EXT-recalled mp2p_icp behaviour; restated in the
// CPU checker with the same operation order.]
struct PlanePair {      // one per query, sorted query order
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

template <int K, bool VERIFY>
__global__ __launch_bounds__(256, (K <= 6 ? 3 : 2)) void k_knn_planes(const float* __restrict__ slx, const float* __restrict__ sly,
                                                    const float* __restrict__ slz, int N, TiledMap mp, PoseF P,
                                                    float thr2, double threshold, double plane_eig_thr,
                                                    PlanePair* __restrict__ out, PlanePair* __restrict__ cache /*the plane of each query's list*/,
                                                    int* __restrict__ knn_pos /*N x K: in = last launch's neighbours (use_seed), out = this launch's*/,
                                                    int use_seed, unsigned int* __restrict__ queue,
                                                    unsigned int* __restrict__ redo_count, int* __restrict__ redo_list,
                                                    unsigned int* __restrict__ changed_items,
                                                    unsigned long long* __restrict__ staged_total, int lds_boxes)
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
    const int n_items = from_list ? (int)*redo_count : (N + kQPW - 1) / kQPW;
    unsigned long long wave_staged = 0ull;
    unsigned int wave_changed = 0u;  // items of this wave with a lane whose neighbour list differs from its seeds
    WaveQueue wq(queue, lane);
    for (int raw = wq.first(); raw < n_items;) {
        const int next_raw_v = wq.pop();
        const int item = from_list ? __builtin_amdgcn_readfirstlane(redo_list[raw]) : raw;

        float qx[2], qy[2], qz[2], reach[2], kbound[2];
        float kd[2][K];          // sorted ascending by (d2, original index)
        unsigned int ko[2][K];   // original indices
        int kp[2][K];            // sorted-map positions
        // insert (du, o, pos) into the sorted list of query k (caller has checked that it belongs there)
        auto insert = [&](int k, float du, unsigned int o, int pos) {
            kd[k][K - 1] = du; ko[k][K - 1] = o; kp[k][K - 1] = pos;
#pragma unroll
            for (int j = K - 1; j > 0; --j) {
                const bool sw = kd[k][j] < kd[k][j - 1] || (kd[k][j] == kd[k][j - 1] && ko[k][j] < ko[k][j - 1]);
                const float td = kd[k][j]; const unsigned int to = ko[k][j]; const int tp = kp[k][j];
                kd[k][j] = sw ? kd[k][j - 1] : td; ko[k][j] = sw ? ko[k][j - 1] : to; kp[k][j] = sw ? kp[k][j - 1] : tp;
                kd[k][j - 1] = sw ? td : kd[k][j - 1]; ko[k][j - 1] = sw ? to : ko[k][j - 1]; kp[k][j - 1] = sw ? tp : kp[k][j - 1];
            }
        };
        int qi[2];
        float lx[2], ly[2], lz[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            qi[k] = item * kQPW + k * 64 + lane;
            const int ic = qi[k] < N ? qi[k] : N - 1;
            lx[k] = slx[ic]; ly[k] = sly[ic]; lz[k] = slz[ic];
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            xform(P, lx[k], ly[k], lz[k], qx[k], qy[k], qz[k]);
#pragma unroll
            for (int j = 0; j < K; ++j) { kd[k][j] = thr2; ko[k][j] = 0u; kp[k][j] = -1; }  // sentinel: (gate^2, 0) never beaten by d2 >= gate^2
        }
        if (use_seed) {
            // warm start: the K neighbours of the last launch are exact candidates; with them in the list the
            // reach is the K-th seed distance instead of the gate, and most tiles are never staged
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int ic = qi[k] < N ? qi[k] : N - 1;
                int js[K];
#pragma unroll
                for (int j = 0; j < K; ++j) js[j] = knn_pos[(size_t)ic * K + j];
                float gx[K], gy[K], gz[K];
                unsigned int go[K];
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const int jc = js[j] >= 0 ? js[j] : 0;
                    gx[j] = mp.sx[jc]; gy[j] = mp.sy[jc]; gz[j] = mp.sz[jc]; go[j] = (unsigned int)mp.perm[jc];
                }
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const float du = dist2(qx[k], qy[k], qz[k], gx[j], gy[j], gz[j]);
                    if (js[j] >= 0 && du < thr2) insert(k, du, go[j], js[j]);  // distinct positions: no duplicates among the seeds
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            reach[k] = reach_of(kd[k][K - 1], qx[k], qy[k], qz[k]);  // K-th best so far, or the gate while the list is not full
            kbound[k] = kd[k][K - 1];  // (fixed during the sweep)
            if (qi[k] >= N) { qx[k] = qy[k] = qz[k] = 1.0e18f; reach[k] = -1.0f; kbound[k] = -1.0f; }  // padding lane
        }
        // VERIFY: tau = K-th seed distance (list full), else the largest float below gate^2 ("d2 < gate^2" as "<=")
        float tau[2];
        int expect[2], cnt[2] = {0, 0};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            int sds = 0;
#pragma unroll
            for (int j = 0; j < K; ++j) sds += kp[k][j] >= 0 ? 1 : 0;
            expect[k] = sds;
            tau[k] = sds == K ? kd[k][K - 1] : __uint_as_float(__float_as_uint(thr2) - 1u);
            if (VERIFY && qi[k] < N) kbound[k] = tau[k];
        }

        const v2f q2x = {qx[0], qx[1]}, q2y = {qy[0], qy[1]}, q2z = {qz[0], qz[1]};
        unsigned long long np_a = 0ull, np_b = 0ull;  // (profiling outputs of the sweep, unused here)
        unsigned int np_c = 0u, np_d = 0u, np_e = 0u;
        const unsigned long long n_staged = tiled_sweep<2, !VERIFY>(mp, lbox, lds_boxes != 0, slist, lane, sm, qx, qy, qz, reach, kbound, [&](int nm, int jb0, int jb1) {
            for (int m = 0; m < nm; m += 4) {
                const float4 X = *reinterpret_cast<const float4*>(&sm[0][m]);
                const float4 Y = *reinterpret_cast<const float4*>(&sm[1][m]);
                const float4 Z = *reinterpret_cast<const float4*>(&sm[2][m]);
                const float xs[4] = {X.x, X.y, X.z, X.w}, ys[4] = {Y.x, Y.y, Y.z, Y.w}, zs[4] = {Z.x, Z.y, Z.z, Z.w};
                if constexpr (VERIFY) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const v2f dv = dist2_pk(q2x, q2y, q2z, xs[u], ys[u], zs[u]);
                        cnt[0] += dv.x <= tau[0] ? 1 : 0;
                        cnt[1] += dv.y <= tau[1] ? 1 : 0;
                    }
                    continue;
                }
                float d[2][4];
                bool cand = false;
#pragma unroll
                for (int u = 0; u < 4; ++u) {  // both queries of the lane per packed instruction
                    const v2f dv = dist2_pk(q2x, q2y, q2z, xs[u], ys[u], zs[u]);
                    d[0][u] = dv.x; d[1][u] = dv.y;
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) cand |= fminf(fminf(d[k][0], d[k][1]), fminf(d[k][2], d[k][3])) <= kd[k][K - 1];
                if (__any(cand)) {  // some lane may have to insert: rare once the lists have tightened
                    const float4 O = *reinterpret_cast<const float4*>(&sm[3][m]);
                    const unsigned int os[4] = {__float_as_uint(O.x), __float_as_uint(O.y), __float_as_uint(O.z),
                                                __float_as_uint(O.w)};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int pos = (m + u) < 32 ? jb0 + m + u : jb1 + m + u - 32;
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            const float du = d[k][u];
                            if (du < kd[k][K - 1] || (du == kd[k][K - 1] && os[u] < ko[k][K - 1])) {
                                bool dup = false;  // a seed met again by the sweep
#pragma unroll
                                for (int j = 0; j < K; ++j) dup |= kp[k][j] == pos;
                                if (!dup) insert(k, du, os[u], pos);
                            }
                        }
                    }
                }
            }
         }, false, np_a, np_b, np_c, np_d, np_e, np_a, np_b);

        bool redo = false;
        if constexpr (VERIFY) {
            bool bad = false;
#pragma unroll
            for (int k = 0; k < 2; ++k) bad |= qi[k] < N && cnt[k] != expect[k];
            redo = __any(bad);
            if (redo && lane == 0) redo_list[atomicAdd(redo_count, 1u)] = item;
        }
        bool item_changed = false;
        if (!redo) {
        // Epilogue.  The plane (centroid, normal, is-it-planar) depends only on the ordered neighbour list -- map
        // points, fixed for the align -- so when a query's list equals the last launch's, the cached plane is reused
        // bit for bit and the fp64 covariance + Jacobi eigen-solve (dearer than the search itself) is skipped.  Near
        // convergence almost no list changes; a wave pays for the solve only if one of its lanes needs it.
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = qi[k];
            const bool in = i < N;
            const size_t ic = in ? (size_t)i : (size_t)(N - 1);
            int m = 0;
#pragma unroll
            for (int j = 0; j < K; ++j) m += (kp[k][j] >= 0 && kd[k][j] < thr2) ? 1 : 0;  // sorted: the first m entries
            bool same = use_seed != 0;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const int now = j < m ? kp[k][j] : -1;
                if (use_seed) same &= knn_pos[ic * K + j] == now;
                if (in) knn_pos[ic * K + j] = now;
            }
            PlanePair pl;  // the plane of the list: valid = "is a plane" (before the query-distance test)
            pl.valid = 0; pl.n_neigh = m;
#pragma unroll
            for (int a = 0; a < 3; ++a) { pl.c[a] = 0; pl.n[a] = 0; }
            const bool solve = in && !same;
            if (__any(solve)) {
                item_changed = true;
                if (solve && m >= 3) {
                    double px[K], py[K], pz[K];
                    double mean[3] = {0, 0, 0};
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        px[j] = py[j] = pz[j] = 0;
                        if (j < m) {
                            px[j] = mp.sx[kp[k][j]]; py[j] = mp.sy[kp[k][j]]; pz[j] = mp.sz[kp[k][j]];
                            mean[0] += px[j]; mean[1] += py[j]; mean[2] += pz[j];
                        }
                    }
                    const double dm = (double)m;
                    mean[0] /= dm; mean[1] /= dm; mean[2] /= dm;
                    double Cm[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        if (j < m) {
                            const double dd[3] = {px[j] - mean[0], py[j] - mean[1], pz[j] - mean[2]};
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
                    const double dist = fabs(pl.n[0] * ((double)qx[k] - pl.c[0]) + pl.n[1] * ((double)qy[k] - pl.c[1]) +
                                             pl.n[2] * ((double)qz[k] - pl.c[2]));
                    if (dist > threshold) {
                        pp.valid = 0;
#pragma unroll
                        for (int a = 0; a < 3; ++a) { pp.c[a] = 0; pp.n[a] = 0; }
                    }
                }
                out[ic] = pp;
            }
        }
        }  // (epilogue)
        wave_changed += item_changed ? 1u : 0u;
        wave_staged += n_staged * 2;  // units of 64 (query, point) pairs
        raw = wq.settle(__builtin_amdgcn_readfirstlane(next_raw_v), n_items);
    }
    if (lane == 0 && wave_staged) atomicAdd(staged_total, wave_staged);
    // (the verify flavour reports its queued items through redo_count; the queued-items launch must not count twice)
    if (!VERIFY && !from_list && lane == 0 && wave_changed) atomicAdd(changed_items, wave_changed);
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
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) {
        const PlanePair pp = pairs[i];
        if (!pp.valid) continue;
        const double l[3] = {slx[i], sly[i], slz[i]};
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
    __shared__ double sm[4][kNAccPlane];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < kNAccPlane; ++k) {
        double v = acc[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if (lane == 0) sm[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < kNAccPlane) {
        double v = 0.0;
        for (int w = 0; w < 4; ++w) v += sm[w][threadIdx.x];
        partials[(size_t)blockIdx.x * kNAccPlane + threadIdx.x] = v;
    }
}

// fixed-order sum of [nblocks][n] partial rows (n <= 128): 8 slices of rows per accumulator with the loads of a
// slice independent of each other, then the 8 slice sums in order.  Deterministic for a given nblocks.
__global__ __launch_bounds__(1024) void k_reduce_rows(const double* __restrict__ partials, int nblocks, int n,
                                                      double* __restrict__ acc, const unsigned int* __restrict__ counters)
{
    // acc[n] = items of the plane matcher whose neighbour lists changed in this iteration (its next launch picks
    // the counting or the insertion flavour from it): counters[0] (insertion launch) + counters[2] (queued by verify)
    if (counters && threadIdx.x == 0) acc[n] = (double)(counters[0] + counters[2]);
    __shared__ double sm[8][128];
    const int k = threadIdx.x & 127, sl = threadIdx.x >> 7;
    double v = 0.0;
    if (k < n) {
        int b = sl;
        for (; b + 24 < nblocks; b += 32) {  // four rows in flight
            const double a0 = partials[(size_t)b * n + k], a1 = partials[(size_t)(b + 8) * n + k];
            const double a2 = partials[(size_t)(b + 16) * n + k], a3 = partials[(size_t)(b + 24) * n + k];
            v += a0; v += a1; v += a2; v += a3;
        }
        for (; b < nblocks; b += 8) v += partials[(size_t)b * n + k];
    }
    sm[sl][k] = v;
    __syncthreads();
    if (sl == 0 && k < n) {
        double t = 0.0;
        for (int s2 = 0; s2 < 8; ++s2) t += sm[s2][k];
        acc[k] = t;
    }
}

// plane pairing in sorted query order -> original order (tests / callers that want the pairing)
__global__ __launch_bounds__(256) void k_unpermute_planes(const int* __restrict__ qperm, const PlanePair* __restrict__ in,
                                                          const int* __restrict__ perm, const int* __restrict__ knn_pos,
                                                          int K, int N, PlanePair* __restrict__ out, int* __restrict__ knn_idx)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int o = qperm[i];
    out[o] = in[i];
    if (knn_idx && knn_pos)
        for (int j = 0; j < K; ++j) {
            const int ps = knn_pos[(size_t)i * K + j];
            knn_idx[(size_t)o * K + j] = ps >= 0 ? perm[ps] : -1;
        }
}

// number of kept pairs of a stored pairing (only when a caller asks for it)
__global__ __launch_bounds__(256) void k_count_kept(const int* __restrict__ idx, int N, unsigned int* __restrict__ counter)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    unsigned int kept = (i < N && idx[i] >= 0) ? 1u : 0u;
    for (int off = 32; off > 0; off >>= 1) kept += __shfl_down(kept, off);
    if ((threadIdx.x & 63) == 0 && kept) atomicAdd(counter, kept);
}

// heavy-first work order for the next launches: counting sort of the 128-query items by the cycles they took in
// the last launch (32 buckets relative to the maximum), one 1024-thread block.  Longest-processing-time-first
// keeps the persistent waves' tail short when a few query groups are much heavier than the rest.
// (Cutting the heavy groups into smaller items was measured and dropped: an item's cost is mostly fixed
// overhead -- box scan, staging and epilogue round trips -- so halves cost nearly as much as the whole.)
__global__ __launch_bounds__(1024) void k_order_items(const unsigned int* __restrict__ cost, int n_items,
                                                      int* __restrict__ order)
{
    __shared__ unsigned int s_max, s_cnt[32], s_off[32];
    if (threadIdx.x == 0) s_max = 1u;
    if (threadIdx.x < 32) s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    unsigned int mx = 1u;
    for (int i = threadIdx.x; i < n_items; i += 1024) mx = max(mx, cost[i]);
    atomicMax(&s_max, mx);
    __syncthreads();
    const float scale = 32.0f / (float)s_max;
    for (int i = threadIdx.x; i < n_items; i += 1024) {
        const int b = 31 - min(31, (int)((float)cost[i] * scale));  // bucket 0 = heaviest
        atomicAdd(&s_cnt[b], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int o = 0;
        for (int b = 0; b < 32; ++b) { s_off[b] = o; o += s_cnt[b]; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_items; i += 1024) {
        const int b = 31 - min(31, (int)((float)cost[i] * scale));
        order[atomicAdd(&s_off[b], 1u)] = i;
    }
}

// sorted-order pairing -> original query order (only when a caller asks for the pairing)
__global__ __launch_bounds__(256) void k_unpermute_pairing(const int* __restrict__ qperm, const int* __restrict__ idx_s,
                                                           const float* __restrict__ d2_s, int N,
                                                           int* __restrict__ out_idx, float* __restrict__ out_d2)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int o = qperm[i];
    out_idx[o] = idx_s[i];
    out_d2[o] = d2_s[i];
}

// boxes of the map tiles (one thread per tile) and super-tiles (one thread per super-tile); SoA [6][n]
__global__ __launch_bounds__(256) void k_tile_boxes(const float* __restrict__ sx, const float* __restrict__ sy,
                                                    const float* __restrict__ sz, int M, int n_tiles_p,
                                                    float* __restrict__ tbox)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_tiles_p) return;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    const int j0 = t * kTileG, j1 = min(j0 + kTileG, M);
    for (int j = j0; j < j1; ++j) {
        mn[0] = fminf(mn[0], sx[j]); mx[0] = fmaxf(mx[0], sx[j]);
        mn[1] = fminf(mn[1], sy[j]); mx[1] = fmaxf(mx[1], sy[j]);
        mn[2] = fminf(mn[2], sz[j]); mx[2] = fmaxf(mx[2], sz[j]);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { tbox[k * n_tiles_p + t] = mn[k]; tbox[(3 + k) * n_tiles_p + t] = mx[k]; }
}

__global__ __launch_bounds__(256) void k_super_boxes(const float* __restrict__ tbox, int n_tiles_p, int n_super,
                                                     float* __restrict__ sbox)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n_super) return;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int t = s * kSuper; t < (s + 1) * kSuper; ++t) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            mn[k] = fminf(mn[k], tbox[k * n_tiles_p + t]);
            mx[k] = fmaxf(mx[k], tbox[(3 + k) * n_tiles_p + t]);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { sbox[k * n_super + s] = mn[k]; sbox[(3 + k) * n_super + s] = mx[k]; }
}

// ---- map preparation for the MFMA matcher (once per map) -------------------------------
// bounding box: per-block partial min/max -> [nblocks][6]; second stage on one block
__global__ __launch_bounds__(256) void k_bbox_partial(const float* __restrict__ gx, const float* __restrict__ gy,
                                                      const float* __restrict__ gz, int M, float* __restrict__ part)
{
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < M; i += gridDim.x * 256) {
        const float v[3] = {gx[i], gy[i], gz[i]};
#pragma unroll
        for (int k = 0; k < 3; ++k) { mn[k] = fminf(mn[k], v[k]); mx[k] = fmaxf(mx[k], v[k]); }
    }
    __shared__ float sm[4][6];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        for (int off = 32; off > 0; off >>= 1) {
            mn[k] = fminf(mn[k], __shfl_down(mn[k], off));
            mx[k] = fmaxf(mx[k], __shfl_down(mx[k], off));
        }
        if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6][k] = mn[k]; sm[threadIdx.x >> 6][3 + k] = mx[k]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = sm[0][threadIdx.x];
        for (int w = 1; w < 4; ++w) v = threadIdx.x < 3 ? fminf(v, sm[w][threadIdx.x]) : fmaxf(v, sm[w][threadIdx.x]);
        part[blockIdx.x * 6 + threadIdx.x] = v;
    }
}

__global__ __launch_bounds__(64) void k_bbox_final(const float* __restrict__ part, int nblocks, float* __restrict__ out)
{
    if (threadIdx.x < 6) {
        float v = part[threadIdx.x];
        for (int b = 1; b < nblocks; ++b)
            v = threadIdx.x < 3 ? fminf(v, part[b * 6 + threadIdx.x]) : fmaxf(v, part[b * 6 + threadIdx.x]);
        out[threadIdx.x] = v;
    }
}

// map image [tile][k][16]: k<3 -> -2*(m_k - c_k), k=3 -> |m - c|^2 (1 - 20u) (the map-point share of
// the filter's error bound, folded in).  Rows >= M are padding.
__global__ __launch_bounds__(256) void k_map_image(const float* __restrict__ gx, const float* __restrict__ gy,
                                                   const float* __restrict__ gz, int M, int M_padded, MapFrame F,
                                                   float* __restrict__ img)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= M_padded) return;
    float ax = 0.f, ay = 0.f, az = 0.f, mm = kMapPadNorm;
    if (p < M) {
        const float x = gx[p] - F.cx, y = gy[p] - F.cy, z = gz[p] - F.cz;
        const float n = fmaf(z, z, fmaf(y, y, x * x));
        mm = n - kFoldCoef * n;
        ax = -2.0f * x; ay = -2.0f * y; az = -2.0f * z;
    }
    float* t = img + (size_t)(p >> 4) * 64 + (p & 15);
    t[0] = ax; t[16] = ay; t[32] = az; t[48] = mm;
}

// ---- accumulation (row a8) ---------------------------------------------------------
struct AccArgs {
    const float *lx, *ly, *lz, *gx, *gy, *gz;
    const int* idx;
    const float* d2;
    unsigned char* outlier;
    int N;
    int stage;
    int use_scale;
    int use_robust;
    double scale_thr, rk_param, rk_scale;
    double cl[3], cg[3];
    double R[9];
};

constexpr int kAccThreads = 256;
constexpr int kAccMaxBlocks = 512;   // rows of partial sums (fixed for a given N: deterministic reduction)

__global__ __launch_bounds__(kAccThreads) void k_accumulate(AccArgs a, double* __restrict__ partials)
{
    double s[kNAcc];
#pragma unroll
    for (int k = 0; k < kNAcc; ++k) s[k] = 0.0;
    const int stride = gridDim.x * kAccThreads;
    // one pairing -> the 24 sums; elements are taken in ascending i per thread (fixed summation order)
    auto element = [&](int i, int j, unsigned char out, double l0, double l1, double l2, double g0, double g1, double g2,
                       float d2v) {
        if (j < 0 || out) return;
        double w = 1.0;
        if (a.stage == 1) {
            double b0 = g0 - a.cg[0], b1 = g1 - a.cg[1], b2 = g2 - a.cg[2];
            double r0 = l0 - a.cl[0], r1 = l1 - a.cl[1], r2 = l2 - a.cl[2];
            const double bn = sqrt(b0 * b0 + b1 * b1 + b2 * b2);
            const double rn = sqrt(r0 * r0 + r1 * r1 + r2 * r2);
            if (bn < 1e-4 || rn < 1e-4) return;
            if (a.use_scale) {
                const double hi = bn > rn ? bn : rn, lo = bn > rn ? rn : bn;
                if (hi / lo > a.scale_thr) {
                    a.outlier[i] = 1;
                    return;
                }
            }
            if (a.use_robust) {
                b0 /= bn; b1 /= bn; b2 /= bn;
                r0 /= rn; r1 /= rn; r2 /= rn;
                const double x = a.R[0] * r0 + a.R[1] * r1 + a.R[2] * r2;
                const double y = a.R[3] * r0 + a.R[4] * r1 + a.R[5] * r2;
                const double z = a.R[6] * r0 + a.R[7] * r1 + a.R[8] * r2;
                double c = x * b0 + y * b1 + z * b2;
                c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
                const double ang = acos(c);
                if (ang > a.rk_param) {
                    const double e = ang - a.rk_param;
                    w *= 1.0 / (1.0 + a.rk_scale * e * e);
                }
            }
        }
        s[0] += w;
        s[1] += w * l0; s[2] += w * l1; s[3] += w * l2;
        s[4] += w * g0; s[5] += w * g1; s[6] += w * g2;
        s[7] += w * l0 * g0; s[8] += w * l0 * g1; s[9] += w * l0 * g2;
        s[10] += w * l1 * g0; s[11] += w * l1 * g1; s[12] += w * l1 * g2;
        s[13] += w * l2 * g0; s[14] += w * l2 * g1; s[15] += w * l2 * g2;
        s[16] += 1.0;
        s[17] += (double)d2v;
        s[18] += w * l0 * l0; s[19] += w * l0 * l1; s[20] += w * l0 * l2;
        s[21] += w * l1 * l1; s[22] += w * l1 * l2; s[23] += w * l2 * l2;
    };
    // two elements per trip with all their loads issued up front (the gather by neighbour position is a dependent
    // load: this halves the exposed latency); they are summed in the same order as a one-by-one loop
    for (int i = blockIdx.x * kAccThreads + threadIdx.x; i < a.N; i += 2 * stride) {
        const int i2 = i + stride;
        const bool in2 = i2 < a.N;
        const int ic2 = in2 ? i2 : i;
        const int jA = a.idx[i], jB = in2 ? a.idx[ic2] : -1;
        const unsigned char oA = a.outlier[i], oB = a.outlier[ic2];
        const float lA0 = a.lx[i], lA1 = a.ly[i], lA2 = a.lz[i], dA = a.d2[i];
        const float lB0 = a.lx[ic2], lB1 = a.ly[ic2], lB2 = a.lz[ic2], dB = a.d2[ic2];
        const int jcA = jA >= 0 ? jA : 0, jcB = jB >= 0 ? jB : 0;
        const float gA0 = a.gx[jcA], gA1 = a.gy[jcA], gA2 = a.gz[jcA];
        const float gB0 = a.gx[jcB], gB1 = a.gy[jcB], gB2 = a.gz[jcB];
        element(i, jA, oA, lA0, lA1, lA2, gA0, gA1, gA2, dA);
        element(i2, jB, oB, lB0, lB1, lB2, gB0, gB1, gB2, dB);
    }
    // fixed-order reduction: lanes (shuffle tree) -> waves (LDS, in wave order) -> one row per block
    __shared__ double sm[kAccThreads / 64][kNAcc];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < kNAcc; ++k) {
        double v = s[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if (lane == 0) sm[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < kNAcc) {
        double v = 0.0;
        for (int w = 0; w < kAccThreads / 64; ++w) v += sm[w][threadIdx.x];
        partials[(size_t)blockIdx.x * kNAcc + threadIdx.x] = v;
    }
}

// sums the per-block rows in a fixed order: 32 interleaved slices per accumulator, then the slices in order
constexpr int kRedSlices = 32;
__global__ __launch_bounds__(kNAcc * kRedSlices) void k_reduce_partials(const double* __restrict__ partials, int nblocks,
                                                                        double* __restrict__ acc,
                                                                        double* __restrict__ host_out /*pinned, may be null*/,
                                                                        unsigned long long seq)
{
    __shared__ double sm[kRedSlices][kNAcc];
    const int k = threadIdx.x % kNAcc, sl = threadIdx.x / kNAcc;
    double v = 0.0;
    for (int b = sl; b < nblocks; b += kRedSlices) v += partials[(size_t)b * kNAcc + k];
    sm[sl][k] = v;
    __syncthreads();
    if (threadIdx.x < kNAcc) {
        double t = 0.0;
        for (int s = 0; s < kRedSlices; ++s) t += sm[s][threadIdx.x];
        acc[threadIdx.x] = t;
        if (host_out) host_out[threadIdx.x] = t;  // straight into the host's pinned block: no copy engine, no extra launch gap
    }
    if (host_out) {  // publish: data first, then the sequence number the host spins on
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
            reinterpret_cast<volatile unsigned long long*>(host_out)[kNAcc + 6] = seq;  // (slots 24..29 serve other read-backs)
            __threadfence_system();
        }
    }
    // the matcher's work-queue / kept / redo counters sit right behind the block: leave them zero for its next launch
    if (threadIdx.x == kNAcc) { acc[kNAcc] = 0.0; acc[kNAcc + 1] = 0.0; }
    if (threadIdx.x >= 32 && threadIdx.x < 32 + 2 * kQueues)  // the tiled matcher's work-queue counters
        reinterpret_cast<unsigned int*>(acc + kNAcc + 8)[(threadIdx.x - 32) * kQueueStride] = 0u;
}

// ------------------------------------------------------------------ host code

int DevBuf::reserve(size_t bytes)
{
    if (bytes <= cap && p) return MOLA_ICP_OK;
    release();
    const size_t want = bytes < 256 ? 256 : bytes;
    HIPCHK(hipMalloc(&p, want));
    cap = want;
    return MOLA_ICP_OK;
}

void DevBuf::release()
{
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}

HipWorkspace::HipWorkspace(int device)
    : device_(device), map_sc_(std::make_shared<SortedCloud>()), loc_sc_(std::make_shared<SortedCloud>())
{
}

HipWorkspace::~HipWorkspace()
{
    if (!inited_) return;
    (void)hipSetDevice(device_);
    if (stream_) (void)hipStreamSynchronize(stream_);
    for (hipEvent_t e : ev_) (void)hipEventDestroy(e);
    map_own_.release(); loc_own_.release(); map_img_.release(); map_meta_.release();
    map_sc_.reset();
    loc_sc_.reset();
    ts_pos_.release(); ts_idx_.release(); ts_d2_.release(); item_cost_.release(); item_order_.release(); redo_list_.release();
    planes_.release(); knn_pos_.release(); plane_acc_.release(); plane_cache_.release();
    if (plane_acc_host_) (void)hipHostFree(plane_acc_host_);
    sort_scratch_.release();
    idx_.release(); d2_.release(); seg_idx_.release(); seg_d2_.release(); outlier_.release(); partials_.release(); acc_dev_.release();
    if (acc_host_) (void)hipHostFree(acc_host_);
    if (meta_host_) (void)hipHostFree(meta_host_);
    if (own_stream_ && stream_) (void)hipStreamDestroy(stream_);
}

int HipWorkspace::init()
{
    if (inited_) return MOLA_ICP_OK;
    int count = 0;
    const hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(MOLA_ICP_E_NODEVICE, std::string("no HIP device available (") +
                                             (e == hipSuccess ? "0 devices" : hipGetErrorString(e)) +
                                             "); this library has no CPU fallback");
    if (device_ < 0) HIPCHK(hipGetDevice(&device_));
    if (device_ >= count) return fail(MOLA_ICP_E_BADARG, "device index out of range");
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device_));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(MOLA_ICP_E_NODEVICE,
                    std::string("device is ") + prop.gcnArchName + " but the kernels are built for gfx950 only");
    num_cus_ = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIPCHK(hipSetDevice(device_));
    HIPCHK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    own_stream_ = true;
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&acc_host_), sizeof(double) * (kNAcc + 8), hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(acc_host_, 0, sizeof(double) * (kNAcc + 8));
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&meta_host_), sizeof(float) * 16, hipHostMallocDefault));
    int rc;
    if ((rc = acc_dev_.reserve(sizeof(double) * (kNAcc + 8) + sizeof(unsigned int) * 2 * kQueues * kQueueStride))) return rc;
    if (std::getenv("MOLA_ICP_DEBUG_STATS")) {  // diagnostic builds of a run, never on by default
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&dbg_stats_), (16 + 8 * kDbgItems) * sizeof(unsigned long long)));
        HIPCHK(hipMemset(dbg_stats_, 0, (16 + 8 * kDbgItems) * sizeof(unsigned long long)));
        if (std::atoi(std::getenv("MOLA_ICP_DEBUG_STATS")) == 2) {  // light mode: per-wave start/end of the tiled matcher only
            HIPCHK(hipMalloc(reinterpret_cast<void**>(&wave_times_), 3 * 8192 * sizeof(unsigned long long)));
            HIPCHK(hipMemset(wave_times_, 0, 3 * 8192 * sizeof(unsigned long long)));
        }
    }
    inited_ = true;
    return MOLA_ICP_OK;
}

int HipWorkspace::set_external_stream(void* s)
{
    int rc = init();
    if (rc) return rc;
    HIPCHK(hipSetDevice(device_));
    if (own_stream_ && stream_) {
        HIPCHK(hipStreamSynchronize(stream_));
        HIPCHK(hipStreamDestroy(stream_));
    }
    if (s) {
        stream_ = static_cast<hipStream_t>(s);
        own_stream_ = false;
    } else {
        HIPCHK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
        own_stream_ = true;
    }
    return MOLA_ICP_OK;
}

static int upload_soa(DevBuf& buf, hipStream_t st, const float* x, const float* y, const float* z, size_t n,
                      const float** dx, const float** dy, const float** dz)
{
    // padded to a multiple of 64 floats per component so the three arrays stay 256-B aligned
    const size_t np = (n + 63) / 64 * 64;
    int rc = buf.reserve(sizeof(float) * 3 * (np ? np : 64));
    if (rc) return rc;
    float* base = buf.as<float>();
    if (n) {
        HIPCHK(hipMemcpyAsync(base, x, sizeof(float) * n, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(base + np, y, sizeof(float) * n, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(base + 2 * np, z, sizeof(float) * n, hipMemcpyHostToDevice, st));
    }
    *dx = base; *dy = base + np; *dz = base + 2 * np;
    return MOLA_ICP_OK;
}

int HipWorkspace::set_map_host(const float* x, const float* y, const float* z, size_t M)
{
    int rc = init();
    if (rc) return rc;
    if (M && (!x || !y || !z)) return fail(MOLA_ICP_E_BADARG, "null map pointer");
    if (M > (size_t)0x7fff0000) return fail(MOLA_ICP_E_BADARG, "map too large for 32-bit indices");
    HIPCHK(hipSetDevice(device_));
    if ((rc = upload_soa(map_own_, stream_, x, y, z, M, &gx_, &gy_, &gz_))) return rc;
    // the host buffers may be pageable: finish the copies before returning (never retain caller pointers)
    HIPCHK(hipStreamSynchronize(stream_));
    M_ = M;
    planes_valid_ = false;
    map_img_valid_ = false;
    if (map_sc_->cached) map_sc_ = std::make_shared<SortedCloud>();  // a cached cloud is immutable: use an own one
    map_sc_->ready = false;
    pairing_valid_ = false;
    seed_valid_ = false;
    knn_seed_valid_ = false;
    return MOLA_ICP_OK;
}

int HipWorkspace::set_map_device(const float* x, const float* y, const float* z, size_t M)
{
    int rc = init();
    if (rc) return rc;
    if (M && (!x || !y || !z)) return fail(MOLA_ICP_E_BADARG, "null map pointer");
    if (M > (size_t)0x7fff0000) return fail(MOLA_ICP_E_BADARG, "map too large for 32-bit indices");
    gx_ = x; gy_ = y; gz_ = z;
    M_ = M;
    planes_valid_ = false;
    map_img_valid_ = false;
    if (map_sc_->cached) map_sc_ = std::make_shared<SortedCloud>();  // a cached cloud is immutable: use an own one
    map_sc_->ready = false;
    pairing_valid_ = false;
    seed_valid_ = false;
    knn_seed_valid_ = false;
    return MOLA_ICP_OK;
}

int HipWorkspace::set_local_host(const float* x, const float* y, const float* z, size_t N)
{
    int rc = init();
    if (rc) return rc;
    if (N && (!x || !y || !z)) return fail(MOLA_ICP_E_BADARG, "null local-cloud pointer");
    if (N > (size_t)0x7fff0000) return fail(MOLA_ICP_E_BADARG, "local cloud too large for 32-bit indices");
    HIPCHK(hipSetDevice(device_));
    if ((rc = upload_soa(loc_own_, stream_, x, y, z, N, &lx_, &ly_, &lz_))) return rc;
    HIPCHK(hipStreamSynchronize(stream_));
    N_ = N;
    planes_valid_ = false;
    if (loc_sc_->cached) loc_sc_ = std::make_shared<SortedCloud>();
    loc_sc_->ready = false;
    cost_valid_ = false;
    order_valid_ = false;
    knn_seed_valid_ = false;
    pairing_valid_ = false;
    seed_valid_ = false;
    knn_seed_valid_ = false;
    return MOLA_ICP_OK;
}

int HipWorkspace::set_local_device(const float* x, const float* y, const float* z, size_t N)
{
    int rc = init();
    if (rc) return rc;
    if (N && (!x || !y || !z)) return fail(MOLA_ICP_E_BADARG, "null local-cloud pointer");
    if (N > (size_t)0x7fff0000) return fail(MOLA_ICP_E_BADARG, "local cloud too large for 32-bit indices");
    lx_ = x; ly_ = y; lz_ = z;
    N_ = N;
    planes_valid_ = false;
    if (loc_sc_->cached) loc_sc_ = std::make_shared<SortedCloud>();
    loc_sc_->ready = false;
    cost_valid_ = false;
    order_valid_ = false;
    knn_seed_valid_ = false;
    pairing_valid_ = false;
    seed_valid_ = false;
    knn_seed_valid_ = false;
    return MOLA_ICP_OK;
}

constexpr int kMfmaQT = 8;             // 128 queries per wave
constexpr int kSegTilesTarget = 8192;  // 131072 map points = 2 MiB of image per segment: L2-resident per XCD

// Builds the derived map image for the MFMA matcher: bounding box -> centre/radius, then
// [tile][4][16] fp32 rows padded to whole LDS chunks.  Once per map.
int HipWorkspace::prepare_map()
{
    if (map_img_valid_) return MOLA_ICP_OK;
    int rc;
    const int M = (int)M_;
    const int nb = 256;
    if ((rc = map_meta_.reserve(sizeof(float) * (6 * nb + 8)))) return rc;
    float* part = map_meta_.as<float>();
    float* bbox = part + 6 * nb;
    hipLaunchKernelGGL(k_bbox_partial, dim3(nb), dim3(256), 0, stream_, gx_, gy_, gz_, M, part);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_bbox_final, dim3(1), dim3(64), 0, stream_, part, nb, bbox);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(meta_host_, bbox, sizeof(float) * 6, hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    double r2 = 0;
    for (int k = 0; k < 3; ++k) {
        const float lo = meta_host_[k], hi = meta_host_[3 + k];
        if (!std::isfinite(lo) || !std::isfinite(hi))
            return fail(MOLA_ICP_E_BADARG, "the map has non-finite coordinates");
        const float c = 0.5f * lo + 0.5f * hi;
        map_center_[k] = c;
        const double h = std::fmax((double)hi - (double)c, (double)c - (double)lo);
        r2 += h * h;
    }
    map_radius_ = (float)(std::sqrt(r2) * 1.00001) + 1e-30f;  // >= max |fl(m - c)|
    if (!(map_radius_ < 1e15f)) return fail(MOLA_ICP_E_BADARG, "map extent too large for the fp32 MFMA filter");
    // tiles of 16 points, rounded up to whole 4-tile steps, plus 4 tiles of prefetch slack
    map_tiles_ = (int)(((M_ + 15) / 16 + 3) / 4 * 4);
    map_segs_ = (map_tiles_ + kSegTilesTarget - 1) / kSegTilesTarget;
    if (map_segs_ > 64) map_segs_ = 64;
    map_seg_tiles_ = ((map_tiles_ + map_segs_ - 1) / map_segs_ + 3) / 4 * 4;
    map_segs_ = (map_tiles_ + map_seg_tiles_ - 1) / map_seg_tiles_;
    const size_t padded = (size_t)(map_tiles_ + 4) * 16;
    if ((rc = map_img_.reserve(sizeof(float) * 4 * padded))) return rc;
    MapFrame F{map_center_[0], map_center_[1], map_center_[2], map_radius_};
    hipLaunchKernelGGL(k_map_image, dim3((unsigned)((padded + 255) / 256)), dim3(256), 0, stream_, gx_, gy_, gz_, M,
                       (int)padded, F, map_img_.as<float>());
    HIPCHK(hipGetLastError());
    map_img_valid_ = true;
    return MOLA_ICP_OK;
}

int morton_sort_points(hipStream_t stream, const float* gx, const float* gy, const float* gz, size_t M, size_t M_padded,
                       const float bbox[6], DevBuf& scratch, float* sxyz, int* perm);

int HipWorkspace::bbox_of(const float* x, const float* y, const float* z, size_t n, float out[6])
{
    int rc;
    const int nb = 256;
    if ((rc = map_meta_.reserve(sizeof(float) * (6 * nb + 8)))) return rc;
    float* part = map_meta_.as<float>();
    float* bbox = part + 6 * nb;
    hipLaunchKernelGGL(k_bbox_partial, dim3(nb), dim3(256), 0, stream_, x, y, z, (int)n, part);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_bbox_final, dim3(1), dim3(64), 0, stream_, part, nb, bbox);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(meta_host_, bbox, sizeof(float) * 6, hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    for (int k = 0; k < 6; ++k) {
        if (!std::isfinite(meta_host_[k])) return fail(MOLA_ICP_E_BADARG, "a cloud has non-finite coordinates");
        out[k] = meta_host_[k];
    }
    return MOLA_ICP_OK;
}

// Once per map: Morton order, tile boxes, super-tile boxes.
int HipWorkspace::prepare_tiles()
{
    if (map_sc_->ready) return MOLA_ICP_OK;
    int rc;
    float bbox[6];
    if ((rc = bbox_of(gx_, gy_, gz_, M_, bbox))) return rc;
    const size_t super_pts = (size_t)kTileG * kSuper;
    map_sc_->n_super = (int)((M_ + super_pts - 1) / super_pts);
    map_sc_->n_super = (map_sc_->n_super + 63) / 64 * 64;  // whole top boxes (the padding tiles get empty boxes)
    map_sc_->n_top = map_sc_->n_super / 64;
    map_sc_->n_tiles_p = map_sc_->n_super * kSuper;
    map_sc_->padded = (size_t)map_sc_->n_tiles_p * kTileG;
    if ((rc = map_sc_->sorted.reserve(sizeof(float) * 3 * map_sc_->padded))) return rc;
    if ((rc = map_sc_->perm.reserve(sizeof(int) * map_sc_->padded))) return rc;
    if ((rc = map_sc_->tbox.reserve(sizeof(float) * 6 * (size_t)map_sc_->n_tiles_p))) return rc;
    if ((rc = map_sc_->sbox.reserve(sizeof(float) * 6 * (size_t)map_sc_->n_super))) return rc;
    if ((rc = map_sc_->ubox.reserve(sizeof(float) * 6 * (size_t)map_sc_->n_top))) return rc;
    if ((rc = morton_sort_points(stream_, gx_, gy_, gz_, M_, map_sc_->padded, bbox, sort_scratch_, map_sc_->sorted.as<float>(),
                                 map_sc_->perm.as<int>())))
        return rc;
    const float* sx = map_sc_->sorted.as<float>();
    hipLaunchKernelGGL(k_tile_boxes, dim3((unsigned)((map_sc_->n_tiles_p + 255) / 256)), dim3(256), 0, stream_, sx, sx + map_sc_->padded,
                       sx + 2 * map_sc_->padded, (int)M_, map_sc_->n_tiles_p, map_sc_->tbox.as<float>());
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_super_boxes, dim3((unsigned)((map_sc_->n_super + 255) / 256)), dim3(256), 0, stream_, map_sc_->tbox.as<float>(),
                       map_sc_->n_tiles_p, map_sc_->n_super, map_sc_->sbox.as<float>());
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_super_boxes, dim3((unsigned)((map_sc_->n_top + 255) / 256)), dim3(256), 0, stream_, map_sc_->sbox.as<float>(),
                       map_sc_->n_super, map_sc_->n_top, map_sc_->ubox.as<float>());
    HIPCHK(hipGetLastError());
    map_sc_->ready = true;
    return MOLA_ICP_OK;
}

// Once per local cloud: Morton order of the queries (a rigid motion keeps them compact).
int HipWorkspace::prepare_queries()
{
    if (loc_sc_->ready) return MOLA_ICP_OK;
    int rc;
    float bbox[6];
    if ((rc = bbox_of(lx_, ly_, lz_, N_, bbox))) return rc;
    loc_sc_->padded = (N_ + kQPW - 1) / kQPW * kQPW;
    if ((rc = loc_sc_->sorted.reserve(sizeof(float) * 3 * loc_sc_->padded))) return rc;
    if ((rc = loc_sc_->perm.reserve(sizeof(int) * loc_sc_->padded))) return rc;
    if ((rc = morton_sort_points(stream_, lx_, ly_, lz_, N_, loc_sc_->padded, bbox, sort_scratch_, loc_sc_->sorted.as<float>(),
                                 loc_sc_->perm.as<int>())))
        return rc;
    loc_sc_->ready = true;
    return MOLA_ICP_OK;
}

int voxel_downsample_device(hipStream_t stream, const float* x, const float* y, const float* z, size_t n, const float bbox[6],
                            float voxel, DevBuf& scratch, float* out_x, float* out_y, float* out_z, size_t capacity,
                            size_t* n_out_host);

// row f4: voxel-grid downsample of a host cloud -> host cloud (centroid per occupied voxel, ascending voxel key)
int HipWorkspace::voxel_downsample(const float* x, const float* y, const float* z, size_t n, double voxel_size,
                                   float* out_x, float* out_y, float* out_z, size_t capacity, size_t* n_out)
{
    int rc = init();
    if (rc) return rc;
    if (!n_out) return fail(MOLA_ICP_E_BADARG, "null n_out");
    *n_out = 0;
    if (!(voxel_size > 0) || !std::isfinite(voxel_size)) return fail(MOLA_ICP_E_BADARG, "voxel size must be > 0");
    if (n && (!x || !y || !z)) return fail(MOLA_ICP_E_BADARG, "null cloud pointer");
    if (n > (size_t)0x7fff0000) return fail(MOLA_ICP_E_BADARG, "cloud too large for 32-bit indices");
    if (n == 0) return MOLA_ICP_OK;
    HIPCHK(hipSetDevice(device_));
    DevBuf in, out;
    const float *dx, *dy, *dz;
    if ((rc = upload_soa(in, stream_, x, y, z, n, &dx, &dy, &dz))) return rc;
    float bbox[6];
    if ((rc = bbox_of(dx, dy, dz, n, bbox))) { in.release(); return rc; }
    if ((rc = out.reserve(sizeof(float) * 3 * n))) { in.release(); return rc; }
    float* o = out.as<float>();
    size_t nv = 0;
    rc = voxel_downsample_device(stream_, dx, dy, dz, n, bbox, (float)voxel_size, sort_scratch_, o, o + n, o + 2 * n, n, &nv);
    if (!rc) {
        *n_out = nv;
        const size_t m = nv < capacity ? nv : capacity;
        if (m && out_x && out_y && out_z) {
            hipError_t e = hipMemcpyAsync(out_x, o, sizeof(float) * m, hipMemcpyDeviceToHost, stream_);
            if (e == hipSuccess) e = hipMemcpyAsync(out_y, o + n, sizeof(float) * m, hipMemcpyDeviceToHost, stream_);
            if (e == hipSuccess) e = hipMemcpyAsync(out_z, o + 2 * n, sizeof(float) * m, hipMemcpyDeviceToHost, stream_);
            if (e == hipSuccess) e = hipStreamSynchronize(stream_);
            if (e != hipSuccess) rc = fail(MOLA_ICP_E_HIP, std::string("voxel_downsample copy: ") + hipGetErrorString(e));
        }
    }
    in.release();
    out.release();
    return rc;
}

// ---- row f4: device-resident cloud cache --------------------------------------------------------------
// A cached cloud lives in HBM in raw AND Hilbert-sorted form with its tile boxes, so it can serve as the map
// (`from`) or as the local cloud (`to`) of any later align without upload or sort: the reference keeps
// keyframe clouds in the world model and re-reads them per nearby-KF / loop-closure ICP
// (src/LidarOdometry.cpp:384-388, 658-666), and in odometry each scan is `to` once and `from` once (cpp:278-279).
int HipWorkspace::build_cached(SortedCloud& sc, const float* x, const float* y, const float* z, size_t n)
{
    int rc = init();
    if (rc) return rc;
    if (n && (!x || !y || !z)) return fail(MOLA_ICP_E_BADARG, "null cloud pointer");
    if (n > (size_t)0x7fff0000) return fail(MOLA_ICP_E_BADARG, "cloud too large for 32-bit indices");
    HIPCHK(hipSetDevice(device_));
    if ((rc = upload_soa(sc.raw, stream_, x, y, z, n, &sc.x, &sc.y, &sc.z))) return rc;
    sc.n = n;
    sc.cached = true;
    // build the sorted form through the map-role path of this workspace, then hand the buffers over
    std::shared_ptr<SortedCloud> keep_sc = map_sc_;
    const float *kx = gx_, *ky = gy_, *kz = gz_;
    const size_t kM = M_;
    std::shared_ptr<SortedCloud> tmp(&sc, [](SortedCloud*) {});
    map_sc_ = tmp;
    gx_ = sc.x; gy_ = sc.y; gz_ = sc.z; M_ = n;
    sc.ready = false;
    rc = n ? prepare_tiles() : MOLA_ICP_OK;
    if (!rc) HIPCHK(hipStreamSynchronize(stream_));
    map_sc_ = keep_sc;
    gx_ = kx; gy_ = ky; gz_ = kz; M_ = kM;
    return rc;
}

void HipWorkspace::use_cached_map(const std::shared_ptr<SortedCloud>& sc)
{
    map_sc_ = sc;
    gx_ = sc->x; gy_ = sc->y; gz_ = sc->z;
    M_ = sc->n;
    planes_valid_ = false;
    map_img_valid_ = false;
    pairing_valid_ = false;
    seed_valid_ = false;
    knn_seed_valid_ = false;
}

void HipWorkspace::use_cached_local(const std::shared_ptr<SortedCloud>& sc)
{
    loc_sc_ = sc;
    lx_ = sc->x; ly_ = sc->y; lz_ = sc->z;
    N_ = sc->n;
    planes_valid_ = false;
    cost_valid_ = false;
    order_valid_ = false;
    knn_seed_valid_ = false;
    pairing_valid_ = false;
    seed_valid_ = false;
    knn_seed_valid_ = false;
}

TiledMap HipWorkspace::tiled_map() const
{
    const float* sx = map_sc_->sorted.as<float>();
    return TiledMap{sx, sx + map_sc_->padded, sx + 2 * map_sc_->padded, map_sc_->perm.as<int>(), map_sc_->tbox.as<float>(), map_sc_->n_tiles_p,
                    map_sc_->sbox.as<float>(), map_sc_->n_super, map_sc_->ubox.as<float>(), map_sc_->n_top};
}

int HipWorkspace::launch_tiled(const PoseF& P, float thr2, bool use_seed, unsigned int* counter)
{
    // blocks per CU, measured at C3: 2 -> 0.159 ms, 3 -> 0.150, 4 -> 0.156
    int per_cu = 3;
    if (const char* e = std::getenv("MOLA_ICP_BLOCKS_PER_CU")) per_cu = std::atoi(e) > 0 ? std::atoi(e) : 3;  // tuning knob
    const TiledMap mp = tiled_map();
    const size_t box_bytes = sizeof(float) * 6u * ((size_t)mp.n_top + (size_t)mp.n_super);
    const int lds_boxes = box_bytes <= kMaxLdsBoxBytes ? 1 : 0;  // else the upper levels are read from global memory
    const size_t dyn_lds = lds_boxes ? box_bytes : 0;
    // Queries per lane.  2 (items of 128 queries) amortises the per-item box scan best.  When 128-query items
    // would give every persistent wave between one and two items -- the launch is then two items long for a
    // work of one and a bit -- items of 64 queries (1 per lane; the packed math pairs map points instead) balance
    // better: measured at 500k queries 0.128 -> 0.105 ms.  Below one item per wave the launch is one item long
    // either way and the larger items do less total work (250k: 0.083 vs 0.091 ms).
    const size_t n128 = (N_ + kQPW - 1) / kQPW, slots = (size_t)num_cus_ * per_cu * 4;
    int qpl = (n128 >= slots && n128 < 2 * slots) ? 1 : 2;
    if (const char* e = std::getenv("MOLA_ICP_QPL")) qpl = std::atoi(e) == 1 ? 1 : 2;  // tuning knob
    const int n_items = (int)((N_ + (size_t)(64 * qpl) - 1) / (size_t)(64 * qpl));
    {   // persistent waves with a static first item: every block of the grid must be resident from the start
        int fit = 0;
        if (qpl == 2) HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, k_nn_tiled<false, 2>, 256, dyn_lds));
        else HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, k_nn_tiled<false, 1>, 256, dyn_lds));
        if (fit >= 1 && per_cu > fit) per_cu = fit;
    }
    int grid = num_cus_ * per_cu;
    if (grid > (n_items + 3) / 4) grid = (n_items + 3) / 4;
    int rc;
    if ((rc = item_cost_.reserve(sizeof(unsigned int) * (size_t)n_items))) return rc;
    if ((rc = item_order_.reserve(sizeof(int) * (size_t)n_items))) return rc;
    const int* order = nullptr;
    if (cost_valid_ && !std::getenv("MOLA_ICP_NO_LPT")) {
        // the cost profile drifts slowly with the pose: re-sort at launch 1, 2, 4, 8 after the clouds were set,
        // then every 16th; the order is reused in between
        if (!order_valid_ || launches_since_order_ >= plan_interval_) {
            hipLaunchKernelGGL(k_order_items, dim3(1), dim3(1024), 0, stream_, item_cost_.as<unsigned int>(), n_items,
                               item_order_.as<int>());
            HIPCHK(hipGetLastError());
            plan_interval_ = order_valid_ ? (plan_interval_ < 16 ? plan_interval_ * 2 : 16) : 1;
            order_valid_ = true;
            launches_since_order_ = 0;
        }
        ++launches_since_order_;
        order = item_order_.as<int>();
    }
    if ((rc = ts_pos_.reserve(sizeof(int) * loc_sc_->padded))) return rc;
    if ((rc = ts_idx_.reserve(sizeof(int) * loc_sc_->padded))) return rc;
    if ((rc = ts_d2_.reserve(sizeof(float) * loc_sc_->padded))) return rc;
    const float* sl = loc_sc_->sorted.as<float>();
    if ((rc = redo_list_.reserve(sizeof(int) * (size_t)n_items))) return rc;
    unsigned long long* staged = reinterpret_cast<unsigned long long*>(acc_dev_.as<double>() + kNAcc + 4);
    // counter[2] = redo count; tq = the fast pass's queue counters, tq + kQueues * kQueueStride the exact pass's
    unsigned int* tq = reinterpret_cast<unsigned int*>(acc_dev_.as<double>() + kNAcc + 8);
    unsigned long long* dbg = dbg_stats_ && !wave_times_ ? dbg_stats_ : nullptr;
    // fast pass, then the exact pass over the items it queued (exact ties: duplicate points, lattices; usually
    // none: a few waves that read the count and leave)
#define MOLA_LAUNCH_TILED(QPL)                                                                                        \
    do {                                                                                                              \
        hipLaunchKernelGGL((k_nn_tiled<false, QPL>), dim3(grid), dim3(256), dyn_lds, stream_, sl, sl + loc_sc_->padded,  \
                           sl + 2 * loc_sc_->padded, (int)N_, mp, P, thr2, use_seed ? 1 : 0, ts_pos_.as<int>(),        \
                           ts_idx_.as<int>(), ts_d2_.as<float>(), order, item_cost_.as<unsigned int>(), tq, counter + 2, \
                           redo_list_.as<int>(), staged, dbg, lds_boxes, wave_times_);                                \
        HIPCHK(hipGetLastError());                                                                                    \
        hipLaunchKernelGGL((k_nn_tiled<true, QPL>), dim3(grid < 64 ? grid : 64), dim3(256), dyn_lds, stream_, sl,     \
                           sl + loc_sc_->padded, sl + 2 * loc_sc_->padded, (int)N_, mp, P, thr2,                       \
                           /*seed = fast pass's result*/ 1, ts_pos_.as<int>(), ts_idx_.as<int>(), ts_d2_.as<float>(), \
                           (const int*)nullptr, (unsigned int*)nullptr, tq + kQueues * kQueueStride, counter + 2,     \
                           redo_list_.as<int>(), staged, dbg, lds_boxes, (unsigned long long*)nullptr);               \
    } while (0)
    if (qpl == 2) MOLA_LAUNCH_TILED(2);
    else MOLA_LAUNCH_TILED(1);
#undef MOLA_LAUNCH_TILED
    cost_valid_ = true;
    HIPCHK(hipGetLastError());
    return MOLA_ICP_OK;
}

int HipWorkspace::match_planes(const Mat4& T, const mola_icp_params& p)
{
    int rc = init();
    if (rc) return rc;
    HIPCHK(hipSetDevice(device_));
    if (p.knn < 3 || p.knn > 8) return fail(MOLA_ICP_E_UNSUPPORTED, "Matcher_Point2Plane: knn must be in [3, 8] in this build");
    planes_valid_ = false;
    if (N_ == 0 || M_ == 0) { planes_valid_ = true; planes_empty_ = true; return MOLA_ICP_OK; }
    planes_empty_ = false;
    if ((rc = prepare_tiles())) return rc;
    if ((rc = prepare_queries())) return rc;
    if ((rc = planes_.reserve(sizeof(PlanePair) * loc_sc_->padded))) return rc;
    if ((rc = knn_pos_.reserve(sizeof(int) * loc_sc_->padded * 8))) return rc;
    if ((rc = plane_cache_.reserve(sizeof(PlanePair) * loc_sc_->padded))) return rc;
    PoseF P;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) P.R[3 * r + c] = (float)T(r, c);
        P.t[r] = (float)T(r, 3);
    }
    const float thr2 = (float)(p.matcher_threshold * p.matcher_threshold);
    while (ev_.size() < ev_used_ + 2) {
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        ev_.push_back(e);
    }
    unsigned int* counter = reinterpret_cast<unsigned int*>(acc_dev_.as<double>() + kNAcc);
    HIPCHK(hipMemsetAsync(counter, 0, 4 * sizeof(unsigned int), stream_));
    HIPCHK(hipMemsetAsync(acc_dev_.as<double>() + kNAcc + 8, 0, sizeof(unsigned int) * 2 * kQueues * kQueueStride, stream_));
    counters_clean_ = false;
    HIPCHK(hipEventRecord(ev_[ev_used_], stream_));
    const int n_items = (int)((N_ + kQPW - 1) / kQPW);
    const float* sl = loc_sc_->sorted.as<float>();
    const TiledMap mp = tiled_map();
    unsigned long long* staged = reinterpret_cast<unsigned long long*>(acc_dev_.as<double>() + kNAcc + 4);
    const size_t box_bytes = sizeof(float) * 6u * ((size_t)mp.n_top + (size_t)mp.n_super);
    const int lds_boxes = box_bytes <= kMaxLdsBoxBytes ? 1 : 0;  // else the upper levels are read from global memory
    const size_t dyn_lds = lds_boxes ? box_bytes : 0;
    // persistent waves with a static first item: every block of the grid must be resident from the start
    int fit = 0;
    switch (p.knn) {  // (the verify flavour needs fewer registers than the insertion flavour)
        case 3: HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, k_knn_planes<3, false>, 256, dyn_lds)); break;
        case 4: HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, k_knn_planes<4, false>, 256, dyn_lds)); break;
        case 5: HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, k_knn_planes<5, false>, 256, dyn_lds)); break;
        case 6: HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, k_knn_planes<6, false>, 256, dyn_lds)); break;
        case 7: HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, k_knn_planes<7, false>, 256, dyn_lds)); break;
        default: HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, k_knn_planes<8, false>, 256, dyn_lds)); break;
    }
    int grid = num_cus_ * (fit < 1 ? 1 : (fit > 3 ? 3 : fit));
    if (grid > (n_items + 3) / 4) grid = (n_items + 3) / 4;
    const int knn_seed = (knn_seed_valid_ && planes_knn_ == (int)p.knn && !std::getenv("MOLA_ICP_NO_KNN_SEED")) ? 1 : 0;
    // the counting flavour pays off when few items will need the insertion flavour afterwards: judged by the
    // number of items whose lists changed in the previous iteration (read back with its accumulators)
    const bool verify = knn_seed && knn_changed_items_ >= 0.0 && knn_changed_items_ < 0.3 * (double)n_items &&
                        !std::getenv("MOLA_ICP_NO_KNN_VERIFY");
    knn_changed_items_ = -1.0;  // consumed: only an accumulate_planes() after this launch renews it
    if ((rc = redo_list_.reserve(sizeof(int) * (size_t)n_items))) return rc;
    unsigned int* tq = reinterpret_cast<unsigned int*>(acc_dev_.as<double>() + kNAcc + 8);
    // warm-started launches: the counting flavour over all items, then the insertion flavour over the items it
    // queued (counter[2] = their number); first launch on a cloud pair: the insertion flavour over all items
#define MOLA_LAUNCH_KNN(KK, VER, QUEUE, LIST)                                                                        \
    hipLaunchKernelGGL((k_knn_planes<KK, VER>), dim3(grid), dim3(256), dyn_lds, stream_, sl, sl + loc_sc_->padded,   \
                       sl + 2 * loc_sc_->padded, (int)N_, mp, P, thr2, p.matcher_threshold, p.plane_eigen_threshold,   \
                       planes_.as<PlanePair>(), plane_cache_.as<PlanePair>(), knn_pos_.as<int>(), knn_seed, QUEUE,    \
                       counter + 2, LIST, counter, staged, lds_boxes)
#define MOLA_LAUNCH_KNN_ALL(KK)                                                                                      \
    do {                                                                                                             \
        if (verify) {                                                                                                \
            MOLA_LAUNCH_KNN(KK, true, tq, redo_list_.as<int>());                                                     \
            MOLA_LAUNCH_KNN(KK, false, tq + kQueues * kQueueStride, redo_list_.as<int>());                           \
        } else {                                                                                                     \
            MOLA_LAUNCH_KNN(KK, false, tq, (int*)nullptr);                                                           \
        }                                                                                                            \
    } while (0)
    switch (p.knn) {
        case 3: MOLA_LAUNCH_KNN_ALL(3); break;
        case 4: MOLA_LAUNCH_KNN_ALL(4); break;
        case 5: MOLA_LAUNCH_KNN_ALL(5); break;
        case 6: MOLA_LAUNCH_KNN_ALL(6); break;
        case 7: MOLA_LAUNCH_KNN_ALL(7); break;
        default: MOLA_LAUNCH_KNN_ALL(8); break;
    }
#undef MOLA_LAUNCH_KNN_ALL
#undef MOLA_LAUNCH_KNN
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ev_[ev_used_ + 1], stream_));
    ev_used_ += 2;
    last_kernel_ = MOLA_ICP_NN_TILED;
    planes_knn_ = (int)p.knn;
    planes_valid_ = true;
    knn_seed_valid_ = true;
    return MOLA_ICP_OK;
}

int HipWorkspace::accumulate_planes(double acc[kNAccPlaneHost])
{
    int rc = init();
    if (rc) return rc;
    if (!planes_valid_) return fail(MOLA_ICP_E_BADARG, "accumulate_planes() called before match_planes()");
    HIPCHK(hipSetDevice(device_));
    if ((rc = plane_acc_.reserve(sizeof(double) * kNAccPlane * 514))) return rc;
    double* dacc = plane_acc_.as<double>() + (size_t)512 * kNAccPlane;
    if (planes_empty_) {
        HIPCHK(hipMemsetAsync(dacc, 0, sizeof(double) * (kNAccPlane + 1), stream_));
    } else {
        int nblocks = (int)((N_ + 255) / 256);
        if (nblocks > 512) nblocks = 512;
        const float* sl = loc_sc_->sorted.as<float>();
        hipLaunchKernelGGL(k_accumulate_planes, dim3(nblocks), dim3(256), 0, stream_, sl, sl + loc_sc_->padded, sl + 2 * loc_sc_->padded,
                           planes_.as<PlanePair>(), (int)N_, plane_acc_.as<double>());
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(k_reduce_rows, dim3(1), dim3(1024), 0, stream_, plane_acc_.as<double>(), nblocks, kNAccPlane, dacc,
                           reinterpret_cast<const unsigned int*>(acc_dev_.as<double>() + kNAcc));
        HIPCHK(hipGetLastError());
    }
    if (comm_) {
        const int rc2 = rccl_allreduce_sum_f64(comm_, dacc, kNAccPlane, stream_);
        if (rc2) return rc2;
    }
    if (!plane_acc_host_) HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&plane_acc_host_), sizeof(double) * (kNAccPlane + 1), hipHostMallocDefault));
    HIPCHK(hipMemcpyAsync(plane_acc_host_, dacc, sizeof(double) * (kNAccPlane + 1), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    std::memcpy(acc, plane_acc_host_, sizeof(double) * kNAccPlane);
    knn_changed_items_ = planes_empty_ ? -1.0 : plane_acc_host_[kNAccPlane];
    if (!comm_ && ar_fn_) {
        const int r = ar_fn_(acc, kNAccPlane, 0, ar_user_);
        if (r) return fail(MOLA_ICP_E_COMM, "all-reduce hook failed with code " + std::to_string(r));
    }
    return MOLA_ICP_OK;
}

// plane pairing to host, original query order: valid[N], centroid[N*3], normal[N*3], knn_idx[N*knn] (each may be null)
int HipWorkspace::copy_planes(uint8_t* valid, double* centroid, double* normal, int32_t* knn_idx)
{
    if (!planes_valid_) return fail(MOLA_ICP_E_BADARG, "no plane pairing stored: call match_planes() first");
    if (planes_empty_ || N_ == 0) return MOLA_ICP_OK;
    HIPCHK(hipSetDevice(device_));
    int rc;
    DevBuf tmp_pairs, tmp_knn;
    if ((rc = tmp_pairs.reserve(sizeof(PlanePair) * N_))) return rc;
    if ((rc = tmp_knn.reserve(sizeof(int) * N_ * 8))) { tmp_pairs.release(); return rc; }
    hipLaunchKernelGGL(k_unpermute_planes, dim3((unsigned)((N_ + 255) / 256)), dim3(256), 0, stream_, loc_sc_->perm.as<int>(),
                       planes_.as<PlanePair>(), map_sc_->perm.as<int>(), knn_pos_.as<int>(), planes_knn_, (int)N_,
                       tmp_pairs.as<PlanePair>(), tmp_knn.as<int>());
    std::vector<PlanePair> hp(N_);
    std::vector<int> hk(N_ * (size_t)planes_knn_);
    hipError_t e1 = hipMemcpyAsync(hp.data(), tmp_pairs.p, sizeof(PlanePair) * N_, hipMemcpyDeviceToHost, stream_);
    hipError_t e2 = hipMemcpyAsync(hk.data(), tmp_knn.p, sizeof(int) * N_ * planes_knn_, hipMemcpyDeviceToHost, stream_);
    hipError_t e3 = hipStreamSynchronize(stream_);
    tmp_pairs.release();
    tmp_knn.release();
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) return fail(MOLA_ICP_E_HIP, "copy_planes: HIP copy failed");
    for (size_t i = 0; i < N_; ++i) {
        if (valid) valid[i] = (uint8_t)hp[i].valid;
        for (int a = 0; a < 3; ++a) {
            if (centroid) centroid[3 * i + a] = hp[i].c[a];
            if (normal) normal[3 * i + a] = hp[i].n[a];
        }
    }
    if (knn_idx) std::memcpy(knn_idx, hk.data(), sizeof(int) * hk.size());
    return MOLA_ICP_OK;
}

void HipWorkspace::reset_stats()
{
    ev_used_ = 0;
    last_kernel_ = 0;
    dense_pairs_ = 0;
    if (inited_) (void)hipMemsetAsync(acc_dev_.as<double>() + kNAcc + 4, 0, sizeof(unsigned long long), stream_);
}

int HipWorkspace::collect_stats(double* ms_total, uint32_t* launches, uint32_t* kernel_used, uint64_t* pairs)
{
    if (pairs) {
        unsigned long long staged = 0;
        if (inited_) {
            HIPCHK(hipSetDevice(device_));
            HIPCHK(hipMemcpyAsync(acc_host_ + kNAcc + 4, acc_dev_.as<double>() + kNAcc + 4, sizeof staged,
                                  hipMemcpyDeviceToHost, stream_));
            HIPCHK(hipStreamSynchronize(stream_));
            std::memcpy(&staged, acc_host_ + kNAcc + 4, sizeof staged);
        }
        *pairs = dense_pairs_ + (uint64_t)staged * 64u;  // the tiled kernels count in units of 64 pairs
    }
    double tot = 0;
    if (ev_used_) {
        HIPCHK(hipSetDevice(device_));
        HIPCHK(hipStreamSynchronize(stream_));
        for (size_t i = 0; i + 1 < ev_used_; i += 2) {
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, ev_[i], ev_[i + 1]));
            tot += ms;
        }
    }
    if (wave_times_) {
        std::vector<unsigned long long> w(3 * 8192);
        HIPCHK(hipStreamSynchronize(stream_));
        HIPCHK(hipMemcpy(w.data(), wave_times_, w.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, t1 = 0ull;
        std::vector<unsigned long long> ends;
        for (size_t i = 0; i < 8192; ++i)
            if (w[3 * i + 1]) { t0 = std::min(t0, w[3 * i]); t1 = std::max(t1, w[3 * i + 1]); ends.push_back(w[3 * i + 1]); }
        if (!ends.empty()) {
            std::sort(ends.begin(), ends.end());
            const double span = (double)(t1 - t0);
            double busy = 0;
            for (unsigned long long e : ends) busy += (double)(e - t0);
            std::fprintf(stderr, "[mola_icp debug] last tiled launch: %zu waves, span %.0f ticks; waves done at 10/25/50/75/90/99%%: "
                                 "%.2f %.2f %.2f %.2f %.2f %.2f of the span; mean wave lifetime %.2f of the span\n",
                         ends.size(), span, (ends[ends.size() / 10] - t0) / span, (ends[ends.size() / 4] - t0) / span,
                         (ends[ends.size() / 2] - t0) / span, (ends[ends.size() * 3 / 4] - t0) / span,
                         (ends[ends.size() * 9 / 10] - t0) / span, (ends[ends.size() * 99 / 100] - t0) / span,
                         busy / ends.size() / span);
        }
        HIPCHK(hipMemset(wave_times_, 0, w.size() * sizeof(unsigned long long)));
    }
    if (dbg_stats_ && !wave_times_) {
        unsigned long long h[16] = {};
        HIPCHK(hipStreamSynchronize(stream_));
        HIPCHK(hipMemcpy(h, dbg_stats_, sizeof h, hipMemcpyDeviceToHost));
        std::fprintf(stderr, "[mola_icp debug] nn launches=%zu slow-path entries=%llu survivors=%llu (N=%zu M=%zu); "
                             "tiled: staged points per wave item=%.1f (items=%llu, max=%llu); cycles per item: prologue %.0f "
                             "scan %.0f stage %.0f compute %.0f epilogue %.0f, longest item %llu; per item: super tests %.1f, supers entered %.1f, tile tests %.1f; of scan: tile-box wait %.0f, tile tests %.0f\n",
                     ev_used_ / 2, h[0], h[1], N_, M_, h[3] ? (double)h[2] / (double)h[3] : 0.0, h[3], h[4],
                     h[3] ? (double)h[5] / h[3] : 0.0, h[3] ? (double)h[6] / h[3] : 0.0, h[3] ? (double)h[7] / h[3] : 0.0,
                     h[3] ? (double)h[8] / h[3] : 0.0, h[3] ? (double)h[10] / h[3] : 0.0, h[9],
                     h[3] ? (double)h[11] / h[3] : 0.0, h[3] ? (double)h[12] / h[3] : 0.0, h[3] ? (double)h[13] / h[3] : 0.0,
                     h[3] ? (double)h[14] / h[3] : 0.0, h[3] ? (double)h[15] / h[3] : 0.0);
        HIPCHK(hipMemset(dbg_stats_, 0, sizeof h));
        const size_t n_items = std::min((N_ + 63) / 64, item_cost_.cap / sizeof(unsigned int));  // upper bound (64-query items)
        if (cost_valid_ && N_ > 0 && n_items <= kDbgItems) {  // the heaviest items of the last tiled launch
            std::vector<unsigned long long> rec(8 * n_items);
            HIPCHK(hipMemcpy(rec.data(), dbg_stats_ + 16, rec.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            std::vector<size_t> ord;
            for (size_t i = 0; i < n_items; ++i) if (rec[8 * i]) ord.push_back(i);
            std::sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return rec[8 * a] > rec[8 * b]; });
            for (size_t r = 0; r < ord.size(); r = (r < 4 ? r + 1 : r * 2)) {
                const unsigned long long* q = &rec[8 * ord[r]];
                std::fprintf(stderr, "[mola_icp debug]   rank %zu item %zu: cycles %llu staged %llu supers %llu tile tests %llu | prologue %llu scan %llu passes %llu epilogue %llu\n",
                             r, ord[r], q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7]);
            }
            HIPCHK(hipMemset(dbg_stats_ + 16, 0, rec.size() * sizeof(unsigned long long)));
            std::vector<unsigned int> c(n_items);
            HIPCHK(hipMemcpy(c.data(), item_cost_.p, n_items * sizeof(unsigned int), hipMemcpyDeviceToHost));
            std::sort(c.begin(), c.end());
            double sum = 0;
            for (unsigned int v : c) sum += v;
            std::fprintf(stderr, "[mola_icp debug] item cost: n=%zu mean %.0f p50 %u p90 %u p99 %u max %u; sum/3072 slots = %.0f\n",
                         n_items, sum / n_items, c[n_items / 2], c[n_items * 9 / 10], c[n_items * 99 / 100], c[n_items - 1],
                         sum / 3072.0);
        }
    }
    if (ms_total) *ms_total = tot;
    if (launches) *launches = (uint32_t)(ev_used_ / 2);
    if (kernel_used) *kernel_used = last_kernel_;
    return MOLA_ICP_OK;
}

int HipWorkspace::launch_nn(const Mat4& T, float thr2, int kernel)
{
    PoseF P;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) P.R[3 * r + c] = (float)T(r, c);
        P.t[r] = (float)T(r, 3);
    }
    while (ev_.size() < ev_used_ + 2) {
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        ev_.push_back(e);
    }
    // auto: the tiled matcher wherever sorting pays (it prepares both clouds once), else the dense kernels
    if (kernel == MOLA_ICP_NN_AUTO) {
        if (N_ >= 8192 && M_ >= 8192) kernel = MOLA_ICP_NN_TILED;
        else if (N_ >= 4096 && M_ >= 1024) kernel = MOLA_ICP_NN_MFMA;
        else kernel = MOLA_ICP_NN_VALU;
    }
    if (kernel == MOLA_ICP_NN_TILED) {
        int rc = prepare_tiles();
        if (rc) return rc;
        if ((rc = prepare_queries())) return rc;
    } else if (kernel == MOLA_ICP_NN_MFMA) {
        const int rc = prepare_map();
        if (rc) return rc;
    }
    unsigned int* counter = reinterpret_cast<unsigned int*>(acc_dev_.as<double>() + kNAcc);
    if (!counters_clean_) {  // [0] kept [1] queue (dense kernels) [2] redo count; then the tiled kernels' queue counters
        HIPCHK(hipMemsetAsync(counter, 0, 4 * sizeof(unsigned int), stream_));
        HIPCHK(hipMemsetAsync(acc_dev_.as<double>() + kNAcc + 8, 0, sizeof(unsigned int) * 2 * kQueues * kQueueStride, stream_));
    }
    HIPCHK(hipEventRecord(ev_[ev_used_], stream_));
    if (kernel == MOLA_ICP_NN_TILED) {
        const bool use_seed = seed_valid_ && pairing_sorted_ && !std::getenv("MOLA_ICP_NO_WARM_START");
        const int rc = launch_tiled(P, thr2, use_seed, counter);
        if (rc) return rc;
        last_kernel_ = MOLA_ICP_NN_TILED;
        pairing_sorted_ = true;
        counters_clean_ = false;
        HIPCHK(hipEventRecord(ev_[ev_used_ + 1], stream_));
        ev_used_ += 2;
        return MOLA_ICP_OK;
    }
    const bool use_mfma = kernel == MOLA_ICP_NN_MFMA;
    if (use_mfma) {
        const int n_qgroups = (int)((N_ + kMfmaQT * 16 - 1) / (kMfmaQT * 16));
        const int n_items = n_qgroups * map_segs_;
        int rc;
        if ((rc = seg_idx_.reserve(sizeof(int) * (size_t)map_segs_ * N_))) return rc;
        if ((rc = seg_d2_.reserve(sizeof(float) * (size_t)map_segs_ * N_))) return rc;
        MapFrame F{map_center_[0], map_center_[1], map_center_[2], map_radius_};
        // warm start from the pairing this workspace computed last for the same clouds
        const int* seed = (seed_valid_ && !pairing_sorted_ && !std::getenv("MOLA_ICP_NO_WARM_START")) ? idx_.as<int>()
                                                                                                        : nullptr;
        // persistent grid: every CU gets its resident blocks (2-3 per CU at this register count)
        int per_cu = 3;
        if (const char* e = std::getenv("MOLA_ICP_BLOCKS_PER_CU")) per_cu = std::atoi(e) > 0 ? std::atoi(e) : 3;  // tuning knob
        int grid = num_cus_ * per_cu;
        if (grid > (n_items + 3) / 4) grid = (n_items + 3) / 4;
        hipLaunchKernelGGL((k_nn_mfma<kMfmaQT>), dim3(grid), dim3(256), 0, stream_, lx_, ly_, lz_, (int)N_, gx_, gy_,
                           gz_, (int)M_, map_img_.as<float>(), map_tiles_, map_seg_tiles_, map_segs_, n_qgroups, F, P,
                           thr2, seed, seg_idx_.as<int>(), seg_d2_.as<float>(), counter + 1, dbg_stats_);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(k_nn_merge, dim3((unsigned)((N_ + 255) / 256)), dim3(256), 0, stream_, seg_idx_.as<int>(),
                           seg_d2_.as<float>(), map_segs_, (int)N_, idx_.as<int>(), d2_.as<float>(), counter);
        last_kernel_ = MOLA_ICP_NN_MFMA;
    } else {
        constexpr int QPT = 4, TM = 1024;
        const int grid = (int)((N_ + 256 * QPT - 1) / (256 * QPT));
        hipLaunchKernelGGL((k_nn_valu<QPT, TM>), dim3(grid), dim3(256), 0, stream_, lx_, ly_, lz_, (int)N_, gx_, gy_,
                           gz_, (int)M_, P, thr2, idx_.as<int>(), d2_.as<float>(), counter);
        last_kernel_ = MOLA_ICP_NN_VALU;
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ev_[ev_used_ + 1], stream_));
    ev_used_ += 2;
    pairing_sorted_ = false;
    counters_clean_ = false;
    dense_pairs_ += (uint64_t)N_ * (uint64_t)M_;
    return MOLA_ICP_OK;
}

int HipWorkspace::match(const Mat4& T, double threshold, const mola_icp_params& p, uint64_t* n_pairs)
{
    int rc = init();
    if (rc) return rc;
    HIPCHK(hipSetDevice(device_));
    if (!(threshold > 0)) return fail(MOLA_ICP_E_BADARG, "matcher threshold must be > 0");
    if ((rc = idx_.reserve(sizeof(int) * (N_ ? N_ : 1)))) return rc;
    if ((rc = d2_.reserve(sizeof(float) * (N_ ? N_ : 1)))) return rc;
    {
        const void* before = outlier_.p;
        if ((rc = outlier_.reserve(N_ ? N_ : 1))) return rc;
        if (outlier_.p != before) outlier_cleared_for_ = 0;  // fresh allocation: contents undefined
    }
    const float thr2 = (float)(threshold * threshold);
    if (N_ == 0 || M_ == 0) {
        if (N_) HIPCHK(hipMemsetAsync(idx_.p, 0xff, sizeof(int) * N_, stream_));
        pairing_valid_ = true;
        seed_valid_ = false;
    knn_seed_valid_ = false;
        pairing_sorted_ = false;
        if (n_pairs) *n_pairs = 0;
        return MOLA_ICP_OK;
    }
    if ((rc = launch_nn(T, thr2, p.nn_kernel))) return rc;
    pairing_valid_ = true;
    seed_valid_ = true;
    if (n_pairs) {
        unsigned int* counter = reinterpret_cast<unsigned int*>(acc_dev_.as<double>() + kNAcc);
        unsigned int* hc = reinterpret_cast<unsigned int*>(acc_host_ + kNAcc);
        if (pairing_sorted_) {  // the tiled kernels do not count: do it now
            HIPCHK(hipMemsetAsync(counter, 0, sizeof(unsigned int), stream_));
            hipLaunchKernelGGL(k_count_kept, dim3((unsigned)((N_ + 255) / 256)), dim3(256), 0, stream_, ts_idx_.as<int>(),
                               (int)N_, counter);
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipMemcpyAsync(hc, counter, sizeof(unsigned int), hipMemcpyDeviceToHost, stream_));
        HIPCHK(hipStreamSynchronize(stream_));
        *n_pairs = *hc;
    }
    return MOLA_ICP_OK;
}

int HipWorkspace::accumulate(const mola_icp_params& p, const Mat4& Tcur, int stage, const double cl[3],
                             const double cg[3], bool reset_outliers, double acc[kNAcc])
{
    int rc = init();
    if (rc) return rc;
    if (!pairing_valid_) return fail(MOLA_ICP_E_BADARG, "accumulate() called before match()");
    if (stage != 0 && stage != 1) return fail(MOLA_ICP_E_BADARG, "stage must be 0 or 1");
    if (stage == 1 && (!cl || !cg)) return fail(MOLA_ICP_E_BADARG, "stage 1 needs the centroids");
    HIPCHK(hipSetDevice(device_));
    if (N_ == 0 && !comm_) {
        for (int k = 0; k < kNAcc; ++k) acc[k] = 0;
        return MOLA_ICP_OK;
    }
    if (N_ == 0) {  // an empty shard still joins the all-reduce
        HIPCHK(hipMemsetAsync(acc_dev_.p, 0, sizeof(double) * kNAcc, stream_));
        const int rc2 = rccl_allreduce_sum_f64(comm_, acc_dev_.as<double>(), kNAcc, stream_);
        if (rc2) return rc2;
        HIPCHK(hipMemcpyAsync(acc_host_, acc_dev_.p, sizeof(double) * kNAcc, hipMemcpyDeviceToHost, stream_));
        HIPCHK(hipStreamSynchronize(stream_));
        std::memcpy(acc, acc_host_, sizeof(double) * kNAcc);
        return MOLA_ICP_OK;
    }
    if (reset_outliers && (outliers_dirty_ || outlier_cleared_for_ != N_)) {  // nothing sets a flag unless stage 1 ran
        HIPCHK(hipMemsetAsync(outlier_.p, 0, N_, stream_));
        outliers_dirty_ = false;
        outlier_cleared_for_ = N_;
    }
    if (stage == 1 && p.use_scale_outlier_detector) outliers_dirty_ = true;
    int nblocks = (int)((N_ + kAccThreads - 1) / kAccThreads);
    if (nblocks > kAccMaxBlocks) nblocks = kAccMaxBlocks;
    if ((rc = partials_.reserve(sizeof(double) * kNAcc * kAccMaxBlocks))) return rc;
    AccArgs a{};
    if (pairing_sorted_) {  // tiled matcher: sorted local cloud, neighbour = sorted-map position (local gathers)
        const float* sl = loc_sc_->sorted.as<float>();
        const float* sm = map_sc_->sorted.as<float>();
        a.lx = sl; a.ly = sl + loc_sc_->padded; a.lz = sl + 2 * loc_sc_->padded;
        a.gx = sm; a.gy = sm + map_sc_->padded; a.gz = sm + 2 * map_sc_->padded;
        a.idx = ts_pos_.as<int>(); a.d2 = ts_d2_.as<float>();
    } else {
        a.lx = lx_; a.ly = ly_; a.lz = lz_; a.gx = gx_; a.gy = gy_; a.gz = gz_;
        a.idx = idx_.as<int>(); a.d2 = d2_.as<float>();
    }
    a.outlier = outlier_.as<unsigned char>();
    a.N = (int)N_; a.stage = stage;
    a.use_scale = p.use_scale_outlier_detector; a.use_robust = p.use_robust_kernel;
    a.scale_thr = p.scale_outlier_threshold; a.rk_param = p.robust_kernel_param; a.rk_scale = p.robust_kernel_scale;
    for (int k = 0; k < 3; ++k) { a.cl[k] = cl ? cl[k] : 0.0; a.cg[k] = cg ? cg[k] : 0.0; }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) a.R[3 * r + c] = Tcur(r, c);
    hipLaunchKernelGGL(k_accumulate, dim3(nblocks), dim3(kAccThreads), 0, stream_, a, partials_.as<double>());
    HIPCHK(hipGetLastError());
    // Single GPU: the reduction writes the 24 sums straight into the pinned host block and then a sequence number;
    // the host spins on that number instead of a copy + stream synchronisation (both cost a launch gap and the
    // driver's wake-up latency on a ~0.2 ms iteration).  Sharded over RCCL: the collective runs in between on the
    // device block, then the block is copied.
    const bool direct = !comm_ && !std::getenv("MOLA_ICP_NO_DIRECT_READBACK");
    const unsigned long long seq = ++readback_seq_;
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(kNAcc * kRedSlices), 0, stream_, partials_.as<double>(), nblocks,
                       acc_dev_.as<double>(), direct ? acc_host_ : (double*)nullptr, seq);
    HIPCHK(hipGetLastError());
    counters_clean_ = true;
    if (direct) {
        volatile unsigned long long* flag = reinterpret_cast<volatile unsigned long long*>(acc_host_) + kNAcc + 6;
        bool seen = false;
        for (unsigned long long spins = 0; spins < 400000000ull; ++spins) {  // ~ seconds; then fall back to a real wait
            if (*flag == seq) { seen = true; break; }
            __builtin_ia32_pause();
        }
        if (!seen) {
            HIPCHK(hipStreamSynchronize(stream_));
            if (*flag != seq) return fail(MOLA_ICP_E_HIP, "accumulate(): the reduction kernel did not publish its result");
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        for (int k = 0; k < kNAcc; ++k) acc[k] = acc_host_[k];
        return MOLA_ICP_OK;
    }
    if (comm_) {  // query-sharded: the one collective of the path, in place on the device block (RCCL over xGMI)
        const int rc2 = rccl_allreduce_sum_f64(comm_, acc_dev_.as<double>(), kNAcc, stream_);
        if (rc2) return rc2;
    }
    HIPCHK(hipMemcpyAsync(acc_host_, acc_dev_.p, sizeof(double) * kNAcc, hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    std::memcpy(acc, acc_host_, sizeof(double) * kNAcc);
    return MOLA_ICP_OK;
}

int HipWorkspace::allreduce(double acc[kNAcc])
{
    if (comm_) return MOLA_ICP_OK;  // accumulate() already reduced the device block over RCCL
    if (!ar_fn_) return MOLA_ICP_OK;
    const int rc = ar_fn_(acc, kNAcc, 0, ar_user_);
    if (rc) return fail(MOLA_ICP_E_COMM, "all-reduce hook failed with code " + std::to_string(rc));
    return MOLA_ICP_OK;
}

int HipWorkspace::copy_pairing(int32_t* idx_out, float* d2_out)
{
    if (!pairing_valid_) return fail(MOLA_ICP_E_BADARG, "no pairing stored: call match() first");
    HIPCHK(hipSetDevice(device_));
    if (N_ && pairing_sorted_) {
        hipLaunchKernelGGL(k_unpermute_pairing, dim3((unsigned)((N_ + 255) / 256)), dim3(256), 0, stream_,
                           loc_sc_->perm.as<int>(), ts_idx_.as<int>(), ts_d2_.as<float>(), (int)N_, idx_.as<int>(),
                           d2_.as<float>());
        HIPCHK(hipGetLastError());
    }
    if (N_) {
        if (idx_out) HIPCHK(hipMemcpyAsync(idx_out, idx_.p, sizeof(int) * N_, hipMemcpyDeviceToHost, stream_));
        if (d2_out) HIPCHK(hipMemcpyAsync(d2_out, d2_.p, sizeof(float) * N_, hipMemcpyDeviceToHost, stream_));
    }
    HIPCHK(hipStreamSynchronize(stream_));
    return MOLA_ICP_OK;
}

int HipWorkspace::sync()
{
    if (!inited_) return MOLA_ICP_OK;
    HIPCHK(hipSetDevice(device_));
    HIPCHK(hipStreamSynchronize(stream_));
    return MOLA_ICP_OK;
}

}  // namespace mola_icp_amd
