// kernels_prepare.hpp -- small kernels around the matchers: pairing counters / order / un-permutation, box levels, bounding box, MFMA map image
// Device code of the ICP core for gfx950; included by hip_backend.hip only (one translation unit: the kernels are
// launched from there).  Numeric contract and data layout: hip_backend.hip / DESIGN.md.
#pragma once
#include "kernels_tiled.hpp"

namespace mola_icp_amd {

// number of kept pairs of a stored pairing (only when a caller asks for it).  Grid-stride, one atomic per BLOCK: one per
// wave was 15 625 atomics on one address at 1M queries -- 173 us for a count (they retire one after another, ~13 ns each).
__global__ __launch_bounds__(256) void k_count_kept(const int* __restrict__ idx, int N, unsigned int* __restrict__ counter)
{
    __shared__ unsigned int s_part[4];
    unsigned int kept = 0u;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) kept += idx[i] >= 0 ? 1u : 0u;
    for (int off = 32; off > 0; off >>= 1) kept += __shfl_down(kept, off);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = kept;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int t = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        if (t) atomicAdd(counter, t);
    }
}

// heavy-first work order for the next launches, per segment of the work queue (WaveQueue: kQueues contiguous
// ranges of items, one per XCD).  The ranges are cut so that each holds an equal share of the COST (cycles of the
// last launch) -- regions of the cloud differ in density -- and their boundaries are stored behind the order
// (order[n_items .. n_items + kQueues]).  Inside a range: counting sort by cost (32 buckets relative to the
// maximum): longest-processing-time-first keeps the persistent waves' tail short.  One 1024-thread block.
// (Cutting the heavy groups into smaller items was measured and dropped: an item's cost is mostly fixed
// overhead -- box scan, staging and epilogue round trips -- so halves cost nearly as much as the whole.)
__global__ __launch_bounds__(1024) void k_order_items(const unsigned int* __restrict__ cost, int n_items,
                                                      int* __restrict__ order)
{
    __shared__ unsigned int s_max, s_cnt[kQueues][32], s_off[kQueues][32];
    __shared__ unsigned long long s_pre[1024];
    __shared__ int s_seg[kQueues + 1];
    if (threadIdx.x == 0) s_max = 1u;
    if (threadIdx.x < kQueues * 32) (&s_cnt[0][0])[threadIdx.x] = 0u;
    if (threadIdx.x <= kQueues) s_seg[threadIdx.x] = threadIdx.x == kQueues ? n_items : 0;
    // prefix of the cost in item (= spatial) order: contiguous chunk per thread, Hillis-Steele over the chunk sums
    const int chunk = (n_items + 1023) / 1024, i0 = min(n_items, (int)threadIdx.x * chunk), i1 = min(n_items, i0 + chunk);
    unsigned long long mine = 0;
    unsigned int mx = 1u;
    for (int i = i0; i < i1; ++i) { mine += cost[i]; mx = max(mx, cost[i]); }
    s_pre[threadIdx.x] = mine;
    __syncthreads();
    atomicMax(&s_max, mx);
    for (int off = 1; off < 1024; off <<= 1) {
        const unsigned long long add = threadIdx.x >= (unsigned)off ? s_pre[threadIdx.x - off] : 0ull;
        __syncthreads();
        s_pre[threadIdx.x] += add;
        __syncthreads();
    }
    const unsigned long long total = s_pre[1023];
    unsigned long long run = s_pre[threadIdx.x] - mine;  // cost before my chunk
    for (int i = i0; i < i1; ++i) {  // boundary c = first item whose preceding cost reaches c/kQueues of the total
        for (int c = 1; c < kQueues; ++c) {
            const unsigned long long want = total / kQueues * (unsigned long long)c;
            if (run < want && run + cost[i] >= want) s_seg[c] = i + 1;
        }
        run += cost[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // degenerate profiles (all-zero cost): keep the boundaries monotone
        for (int c = 1; c <= kQueues; ++c) if (s_seg[c] < s_seg[c - 1]) s_seg[c] = s_seg[c - 1];
    }
    __syncthreads();
    if (threadIdx.x <= kQueues) order[n_items + threadIdx.x] = s_seg[threadIdx.x];
    auto seg_of = [&](int i) -> int {
        int c = 0;
        while (c + 1 < kQueues && s_seg[c + 1] <= i) ++c;
        return c;
    };
    const float scale = 32.0f / (float)s_max;
    for (int i = threadIdx.x; i < n_items; i += 1024) {
        const int b = 31 - min(31, (int)((float)cost[i] * scale));  // bucket 0 = heaviest
        atomicAdd(&s_cnt[seg_of(i)][b], 1u);
    }
    __syncthreads();
    if (threadIdx.x < kQueues) {
        unsigned int o = (unsigned int)s_seg[threadIdx.x];
        for (int b = 0; b < 32; ++b) { s_off[threadIdx.x][b] = o; o += s_cnt[threadIdx.x][b]; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_items; i += 1024) {
        const int b = 31 - min(31, (int)((float)cost[i] * scale));
        order[atomicAdd(&s_off[seg_of(i)][b], 1u)] = i;
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

// The work list of k_nn_tiled<., 2> (128-query items): as k_order_items -- kQueues contiguous ranges of equal cost, one
// per XCD, heaviest first inside a range -- but an item whose cost exceeds kSplitShare of a wave's fair share of the
// launch (total cost / resident waves) is listed as its two 64-query HALVES (kHalfFlag | 2 * item + half).  A launch is
// as long as its longest wave: at 1M x 1M the heaviest 1 % of the items alone took 1.2-1.3 x the fair share, and the
// waves sat idle 30 % of the launch.  cost2[2 i], cost2[2 i + 1]: last launch's cycles of item i (whole: second slot 0)
// or of its halves.  Layout of `order`: 2 n_items entry slots, kQueues + 1 boundaries (in entries), the entry count.
constexpr float kSplitShare = 0.55f;
__global__ __launch_bounds__(1024) void k_order_entries(const unsigned int* __restrict__ cost2, int n_items, int n_slots,
                                                        int* __restrict__ order)
{
    __shared__ unsigned int s_max, s_cnt[kQueues][32], s_off[kQueues][32];
    __shared__ unsigned long long s_pre[1024];
    __shared__ int s_seg[kQueues + 1], s_ebase[kQueues + 1];
    const int cap = 2 * n_items;
    if (threadIdx.x == 0) s_max = 1u;
    if (threadIdx.x < kQueues * 32) (&s_cnt[0][0])[threadIdx.x] = 0u;
    if (threadIdx.x <= kQueues) s_seg[threadIdx.x] = threadIdx.x == kQueues ? n_items : 0;
    auto cost = [&](int i) -> unsigned long long { return (unsigned long long)cost2[2 * i] + cost2[2 * i + 1]; };
    const int chunk = (n_items + 1023) / 1024, i0 = min(n_items, (int)threadIdx.x * chunk), i1 = min(n_items, i0 + chunk);
    unsigned long long mine = 0;
    unsigned int mx = 1u;
    for (int i = i0; i < i1; ++i) { const unsigned long long c = cost(i); mine += c; mx = max(mx, (unsigned int)min(c, 0xffffffffull)); }
    s_pre[threadIdx.x] = mine;
    __syncthreads();
    atomicMax(&s_max, mx);
    for (int off = 1; off < 1024; off <<= 1) {
        const unsigned long long add = threadIdx.x >= (unsigned)off ? s_pre[threadIdx.x - off] : 0ull;
        __syncthreads();
        s_pre[threadIdx.x] += add;
        __syncthreads();
    }
    const unsigned long long total = s_pre[1023];
    unsigned long long run = s_pre[threadIdx.x] - mine;
    for (int i = i0; i < i1; ++i) {
        const unsigned long long c = cost(i);
        for (int q = 1; q < kQueues; ++q) {
            const unsigned long long want = total / kQueues * (unsigned long long)q;
            if (run < want && run + c >= want) s_seg[q] = i + 1;
        }
        run += c;
    }
    __syncthreads();
    if (threadIdx.x == 0)
        for (int q = 1; q <= kQueues; ++q) if (s_seg[q] < s_seg[q - 1]) s_seg[q] = s_seg[q - 1];
    __syncthreads();
    auto seg_of = [&](int i) -> int {
        int q = 0;
        while (q + 1 < kQueues && s_seg[q + 1] <= i) ++q;
        return q;
    };
    const unsigned long long thr = (unsigned long long)((double)total / (double)(n_slots > 0 ? n_slots : 1) * (double)kSplitShare);
    const float scale = 32.0f / (float)s_max;
    // (Listing the LIGHTEST items -- the ones served last, whose latency is the launch's tail -- as halves too was measured:
    //  no gain; a half is still a chain of the same dependent round trips.)
    // an item -> one entry, or two: (cost, code) of entry e in {0, 1}; returns the number of entries
    auto entries_of = [&](int i, unsigned int (&ec)[2], int (&code)[2]) -> int {
        const unsigned long long c = cost(i);
        if (total == 0 || c <= thr) { ec[0] = (unsigned int)min(c, 0xffffffffull); code[0] = i; return 1; }
        const unsigned int h1 = cost2[2 * i + 1];
        // halves measured last time, or 0.58 of the whole each (a half costs more than half: the per-item work is shared)
        ec[0] = h1 ? cost2[2 * i] : (unsigned int)min(c * 58ull / 100ull, 0xffffffffull);
        ec[1] = h1 ? h1 : ec[0];
        code[0] = kHalfFlag | (2 * i); code[1] = kHalfFlag | (2 * i + 1);
        return 2;
    };
    for (int i = threadIdx.x; i < n_items; i += 1024) {
        unsigned int ec[2]; int code[2];
        const int ne = entries_of(i, ec, code), q = seg_of(i);
        for (int e = 0; e < ne; ++e) atomicAdd(&s_cnt[q][31 - min(31, (int)((float)ec[e] * scale))], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int o = 0;
        for (int q = 0; q < kQueues; ++q) {
            s_ebase[q] = (int)o;
            for (int b = 0; b < 32; ++b) { s_off[q][b] = o; o += s_cnt[q][b]; }
        }
        s_ebase[kQueues] = (int)o;
    }
    __syncthreads();
    if (threadIdx.x <= kQueues) order[cap + threadIdx.x] = s_ebase[threadIdx.x];
    if (threadIdx.x == 0) order[cap + kQueues + 1] = s_ebase[kQueues];
    for (int i = threadIdx.x; i < n_items; i += 1024) {
        unsigned int ec[2]; int code[2];
        const int ne = entries_of(i, ec, code), q = seg_of(i);
        for (int e = 0; e < ne; ++e) order[atomicAdd(&s_off[q][31 - min(31, (int)((float)ec[e] * scale))], 1u)] = code[e];
    }
}

// map-slab point -> original map index (row e); -1 stays -1
__global__ __launch_bounds__(256) void k_remap_indices(const int* __restrict__ idx, const int* __restrict__ orig, int N,
                                                       int* __restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int j = idx[i];
    out[i] = j >= 0 ? orig[j] : -1;
}

// boxes of the map tiles and of the two levels above them; SoA [6][n].  One 32-lane group per tile (a point per lane, coalesced;
// a thread per tile walking its 32 points took 13 us for a 120k-point map), one wave per super-tile (a tile box per lane; a thread
// per super-tile took 19 + 9 us for the two upper levels: 384 dependent loads each).  min / max: the boxes do not depend on the order.
__global__ __launch_bounds__(256) void k_tile_boxes(const float* __restrict__ sx, const float* __restrict__ sy,
                                                    const float* __restrict__ sz, int M, int n_tiles_p,
                                                    float* __restrict__ tbox)
{
    static_assert(kTileG == 32, "one 32-lane group per tile");
    const int t = blockIdx.x * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
    if (t >= n_tiles_p) return;   // (whole 32-lane groups leave together)
    const int j = t * kTileG + l;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (j < M) { mn[0] = mx[0] = sx[j]; mn[1] = mx[1] = sy[j]; mn[2] = mx[2] = sz[j]; }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) {
            mn[k] = fminf(mn[k], __shfl_xor(mn[k], off, 32));
            mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], off, 32));
        }
    }
    if (l < 3) tbox[l * n_tiles_p + t] = l == 0 ? mn[0] : l == 1 ? mn[1] : mn[2];
    else if (l < 6) tbox[l * n_tiles_p + t] = l == 3 ? mx[0] : l == 4 ? mx[1] : mx[2];
}

__global__ __launch_bounds__(256) void k_super_boxes(const float* __restrict__ tbox, int n_tiles_p, int n_super,
                                                     float* __restrict__ sbox)
{
    static_assert(kSuper == 64, "one wave per super-tile, a tile box per lane");
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
    if (s >= n_super) return;   // (whole waves leave together)
    const int t = s * kSuper + l;
    float v[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) v[k] = tbox[k * n_tiles_p + t];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float o = __shfl_xor(v[k], off);
            v[k] = k < 3 ? fminf(v[k], o) : fmaxf(v[k], o);
        }
    }
    if (l < 6) sbox[l * n_super + s] = l == 0 ? v[0] : l == 1 ? v[1] : l == 2 ? v[2] : l == 3 ? v[3] : l == 4 ? v[4] : v[5];
}

// ---- map preparation for the MFMA matcher (once per map) -------------------------------
// bounding box: per-block partial min/max -> [nblocks][6]; second stage on one block
__global__ __launch_bounds__(256) void k_bbox_partial(const float* __restrict__ gx, const float* __restrict__ gy,
                                                      const float* __restrict__ gz, int M, float* __restrict__ part)
{
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < M; i += gridDim.x * 256) {
        const float v[3] = {gx[i], gy[i], gz[i]};
        // (fminf / fmaxf drop a NaN: a coordinate that is not a finite number poisons the maximum instead -- +inf survives the
        // whole reduction and fails the host's finiteness check of the box)
#pragma unroll
        for (int k = 0; k < 3; ++k) { mn[k] = fminf(mn[k], v[k]); mx[k] = fabsf(v[k]) <= 3.4028234e38f ? fmaxf(mx[k], v[k]) : INFINITY; }
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

// The prepare chain's form of the same rows: 1024 points per workgroup and trip, a thread's four loads issued together (the cloud
// has just been uploaded: every load is a trip to HBM, and k_bbox_partial's grid-stride loop made them one after another -- 8.8 us
// for a 120k-point scan).  Same semantics per value (a non-finite coordinate poisons the maximum).  Rows [gridDim.x][6].
__global__ __launch_bounds__(256) void k_bbox_rows(const float* __restrict__ gx, const float* __restrict__ gy,
                                                   const float* __restrict__ gz, int M, float* __restrict__ part)
{
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i0 = (int)blockIdx.x * 1024 + (int)threadIdx.x; i0 < M; i0 += (int)gridDim.x * 1024) {
        float v[4][3];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 256 * u;
            ok[u] = i < M;
            v[u][0] = v[u][1] = v[u][2] = 0.f;
            if (ok[u]) { v[u][0] = gx[i]; v[u][1] = gy[i]; v[u][2] = gz[i]; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (ok[u]) {
#pragma unroll
                for (int k = 0; k < 3; ++k) { mn[k] = fminf(mn[k], v[u][k]); mx[k] = fabsf(v[u][k]) <= 3.4028234e38f ? fmaxf(mx[k], v[u][k]) : INFINITY; }
            }
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

// (256 threads, a partial row each, then a fixed-shape min / max tree: six lanes walking all rows one after the other took 22 us)
__global__ __launch_bounds__(256) void k_bbox_final(const float* __restrict__ part, int nblocks, float* __restrict__ out)
{
    float v[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int b = threadIdx.x; b < nblocks; b += 256) {
#pragma unroll
        for (int k = 0; k < 6; ++k) v[k] = k < 3 ? fminf(v[k], part[b * 6 + k]) : fmaxf(v[k], part[b * 6 + k]);
    }
    __shared__ float sm[4][6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float o = __shfl_xor(v[k], off);
            v[k] = k < 3 ? fminf(v[k], o) : fmaxf(v[k], o);
        }
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6][k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        float r = sm[0][threadIdx.x];
        for (int w = 1; w < 4; ++w) r = threadIdx.x < 3 ? fminf(r, sm[w][threadIdx.x]) : fmaxf(r, sm[w][threadIdx.x]);
        out[threadIdx.x] = r;
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

}  // namespace mola_icp_amd
