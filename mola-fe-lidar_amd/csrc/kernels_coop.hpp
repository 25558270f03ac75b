// kernels_coop.hpp -- k_nn_coop: the tiled matcher for SMALL and BATCHED problems (odometry-size clouds, the
// loop-closure Monte-Carlo, the nearby-keyframe batch).
// Device code of the ICP core for gfx950; included by hip_backend.hip only.  Numeric contract: hip_backend.hip / DESIGN.md.
//
// k_nn_tiled gives every persistent wave whole 128-query items; with <= ~0.4M queries there are fewer items than
// wave slots, a launch is ONE item long and that item is a chain of dependent round trips on a lone wave (measured at
// 100k x 100k: 24 us for the median item, 49 us for the launch).  Here ONE WORKGROUP owns one item: its four waves
// hold the same 128 queries, run the same (cheap) box scan and deal the candidate tiles round-robin among themselves
// (tiled_sweep<.., NPARTS = 4>); the four partial results are merged per query through LDS and waves 0/1 finish 64
// queries each.  No work queue, no atomics on the launch path, no persistent loop: grid = (items, problems), a block
// that has no item leaves at once.  blockIdx.y selects the problem, so K independent problems -- K initial poses on
// one cloud pair (src/LidarOdometry.cpp:767-788) or K different pairs (cpp:704-741) -- share one launch.
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
    int use_seed;                  // pos_s holds the previous launch's neighbours (sorted-map positions)
    int* pos_s;                    // in: seeds, out: neighbour position   } the pairing, in SORTED query order
    int* idx_s;                    // out: neighbour's original map index   }
    float* d2_s;                   // out: squared distance                 }
    unsigned int* redo_count;      // items with exact distance ties: queued by the fast flavour, consumed by the exact one
    int* redo_list;
    unsigned long long* staged;    // statistics: kStatSlots counters on separate lines (units of 64 evaluated pairs)
};
constexpr int kStatSlots = 64, kStatStride = 16;  // (u64 units: one 128-byte line per slot)
constexpr int kCoopMaxBatch = 12;                 // problems per launch (kernel arguments are limited to 4 KB; = kAccMaxBatch)
template <int KMAX> struct NnBatch { NnProblem p[KMAX]; };

constexpr int kCoopParts = 4;  // waves of a workgroup = parts of an item

template <bool EXACT, int KMAX>
__global__ __launch_bounds__(256, 4) void k_nn_coop(const NnBatch<KMAX> batch, int lds_boxes,
                                                    unsigned long long* __restrict__ wave_times /*diagnostics, usually null*/)
{
    __shared__ __attribute__((aligned(16))) float s_m[4][4][64];  // per wave: x, y, z, original index of 64 staged points
    __shared__ int s_list[4][kMaxList];
    __shared__ unsigned int s_mg[kCoopParts][6][64];  // per wave: its partial result for the 128 queries
    __shared__ unsigned int s_flag[4];                // [0..1] tie seen by finishing wave 0/1, [2..3] staged points (lo/hi not needed: < 2^32)
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];  // the upper box levels, if they fit
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float(*sm)[64] = s_m[wave];
    int* slist = s_list[wave];
    const NnProblem& pb = batch.p[KMAX == 1 ? 0 : blockIdx.y];
    const int N = pb.N;
    const float thr2 = pb.thr2;
    const int use_seed = pb.use_seed;
    const unsigned int n_work = EXACT ? *pb.redo_count : (unsigned int)((N + kQPW - 1) / kQPW);
    if (blockIdx.x >= n_work) return;  // nothing for this workgroup (uniform: before any barrier)
    const TiledMap mp = pb.mp;
    const PoseF P = pb.P;
    const lds_f32* lbox = (const lds_f32*)s_dyn;
    const unsigned long long t_wave0 = wave_times ? wall_clock64() : 0ull;  // 100 MHz, the same on every XCD
    const unsigned long long c0 = wave_times ? __builtin_amdgcn_s_memtime() : 0ull;
    if (threadIdx.x < 4) s_flag[threadIdx.x] = 0u;
    if (lds_boxes) load_boxes_to_lds(mp, (lds_f32*)s_dyn);  // (ends with a barrier)
    else __syncthreads();
    unsigned int block_staged = 0u;
    const unsigned long long c1 = wave_times ? __builtin_amdgcn_s_memtime() : 0ull;
    unsigned long long c2 = 0ull, c3 = 0ull, c4 = 0ull, st_dbg = 0ull;

    for (unsigned int e = blockIdx.x; e < n_work; e += gridDim.x) {  // (fast flavour: grid.x >= items, one trip)
        const int item = EXACT ? pb.redo_list[e] : (int)e;
        float qx[2], qy[2], qz[2], reach[2];
        unsigned long long key[2];  // EXACT: packed (d2, original index)
        float best[2];              // running minimum (the sweep's box tests read it)
        int bpos[2];                // EXACT: sorted position of the best point; fast: of its kGroup-point group
        int tie[2] = {0, 0};
        int qi[2], js[2];
        float lx[2], ly[2], lz[2];
        // round trip 1: the lane's two queries and their seeds (every wave of the workgroup loads the same 128)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            qi[k] = item * kQPW + k * 64 + lane;
            if (qi[k] >= N) qi[k] = N;  // padding lane
            const int ic = qi[k] < N ? qi[k] : N - 1;
            lx[k] = pb.slx[ic]; ly[k] = pb.sly[ic]; lz[k] = pb.slz[ic];
            js[k] = use_seed ? pb.pos_s[ic] : -1;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) xform(P, lx[k], ly[k], lz[k], qx[k], qy[k], qz[k]);
        // round trip 2: the seeds' coordinates
        float gsx[2], gsy[2], gsz[2];
        unsigned int gso[2] = {0u, 0u};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int jc = js[k] >= 0 ? js[k] : 0;
            gsx[k] = mp.sx[jc]; gsy[k] = mp.sy[jc]; gsz[k] = mp.sz[jc];
            if (EXACT) gso[k] = (unsigned int)mp.perm[jc];
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            key[k] = ((unsigned long long)__float_as_uint(thr2) << 32);  // (gate^2, index 0): "no neighbour" sentinel
            best[k] = thr2;
            bpos[k] = -1;
            const float d = dist2(qx[k], qy[k], qz[k], gsx[k], gsy[k], gsz[k]);
            if (js[k] >= 0 && d < thr2) {  // warm start: last launch's neighbour is an exact candidate
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

        if (wave_times) c2 = __builtin_amdgcn_s_memtime();
        unsigned long long np_a = 0ull, np_b = 0ull;  // (profiling outputs of the sweep, unused here)
        unsigned int np_c = 0u, np_d = 0u, np_e = 0u;
        const unsigned long long n_staged = tiled_sweep<2, EXACT, kCoopParts>(
            mp, lbox, lds_boxes != 0, slist, lane, wave, sm, qx, qy, qz, reach, best,
            [&](int nm, int jb0, int jb1) {
                if constexpr (EXACT) nn_visit_exact<2>(sm, nm, jb0, jb1, qx, qy, qz, key, best, bpos);
                else nn_visit_fast<2>(sm, nm, jb0, jb1, qx, qy, qz, best, bpos, tie);
            },
            false, np_a, np_b, np_c, np_d, np_e, np_a, np_b);

        if (wave_times) { c3 = __builtin_amdgcn_s_memtime(); st_dbg = n_staged; }
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
        if (wave_times) c4 = __builtin_amdgcn_s_memtime();

        if (wave < 2) {  // wave w finishes the queries k = w of every lane (64 consecutive queries, coalesced stores)
            const int k = wave;
            const float fqx = k ? qx[1] : qx[0], fqy = k ? qy[1] : qy[0], fqz = k ? qz[1] : qz[0];
            const int fqi = k ? qi[1] : qi[0];
            int rpos = -1, roi = -1;
            float rd = thr2;
            bool any_tie = false;
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
                if (d < thr2) { rd = d; rpos = mp_; roi = (int)(unsigned int)(mk & 0xffffffffu); }
            } else {
                float b = __uint_as_float(s_mg[0][k][lane]);
                int p = (int)s_mg[0][2 + k][lane];
                int t = (int)s_mg[0][4 + k][lane];
#pragma unroll
                for (int w = 1; w < kCoopParts; ++w) {
                    const float b2 = __uint_as_float(s_mg[w][k][lane]);
                    const int p2 = (int)s_mg[w][2 + k][lane], t2 = (int)s_mg[w][4 + k][lane];
                    const bool lt = b2 < b, eq = b2 == b;
                    // an EQUAL minimum in a different group is a tie the exact flavour must resolve; the same group on
                    // both sides (the common seed group, or a group two waves met) is the same candidate
                    t = lt ? t2 : (eq ? (t | t2 | (int)(p2 != p)) : t);
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
                    }
                }
                if (p >= 0) {
                    rd = b; rpos = pos; roi = (int)bo;
                    if (pos < 0) t = 1;  // cannot happen (same arithmetic); be safe: exact pass
                }
                any_tie = fqi < N && t != 0;
            }
            if (fqi < N) {  // coalesced: the pairing stays in sorted query order
                pb.pos_s[fqi] = rpos;
                pb.idx_s[fqi] = rpos >= 0 ? roi : -1;
                pb.d2_s[fqi] = rd;
            }
            if (!EXACT) {
                const bool wt = __any(any_tie);
                if (lane == 0) s_flag[k] = wt ? 1u : 0u;
            }
        }
        __syncthreads();  // s_mg / s_flag are rewritten by the next trip; the tie flags of both finishing waves are in
        if (threadIdx.x == 0) {
            if (!EXACT && (s_flag[0] | s_flag[1])) pb.redo_list[atomicAdd(pb.redo_count, 1u)] = item;
            block_staged += s_flag[2];
            s_flag[2] = 0u;
        }
        if (e + gridDim.x < n_work) __syncthreads();  // (exact flavour only: thread 0's reset before the next trip's adds)
    }
    if (wave_times && lane == 0 && blockIdx.y == 0 && blockIdx.x * 4 + wave < 8192) {
        // [start, end (wall clock)], then shader cycles: setup (boxes -> LDS), prologue, sweep, wait for the other waves, epilogue; staged points
        unsigned long long* w = wave_times + 8 * (size_t)(blockIdx.x * 4 + wave);
        w[0] = t_wave0; w[1] = wall_clock64(); w[2] = c1 - c0; w[3] = c2 - c1; w[4] = c3 - c2; w[5] = c4 - c3;
        w[6] = __builtin_amdgcn_s_memtime() - c4; w[7] = st_dbg;
    }
    if (threadIdx.x == 0 && block_staged)  // executed work in units of 64 (query, point) pairs; slotted: no same-address burst
        atomicAdd(pb.staged + (size_t)(blockIdx.x & (kStatSlots - 1)) * kStatStride, (unsigned long long)block_staged * 2ull);
}

}  // namespace mola_icp_amd
