// kernels_dense.hpp -- dense N x M matchers: k_nn_valu, k_nn_mfma (+ k_nn_merge)
// Device code of the ICP core for gfx950; included by hip_backend.hip only (one translation unit: the kernels are
// launched from there).  Numeric contract and data layout: hip_backend.hip / DESIGN.md.
#pragma once
#include "kernels_common.hpp"

namespace mola_icp_amd {

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

}  // namespace mola_icp_amd
