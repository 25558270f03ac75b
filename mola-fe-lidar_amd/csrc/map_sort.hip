// map_sort.hip -- one-time spatial ordering of a cloud for the tiled matcher: 30-bit Hilbert-curve keys (isotropic 10-bit cells
// on the cloud's bounding box; the Hilbert curve is continuous, so every run of consecutive points is spatially compact --
// Morton order has scene-sized jumps), a stable sort, the gather, and the three box levels of a map.
// Plays the role the kd-tree build plays in the reference's CPU path (once per new map -- in odometry once per SCAN,
// src/LidarOdometry.cpp:215-234, 279), so its launch count is on the critical path of every scan:
//
//   k_bbox_partial (hip_backend.hip)  per-block min / max rows
//   k_hilbert_keys                    final box from the rows (every workgroup, redundantly: no launch in between) + the keys
//   k_sort_runs                       a bitonic sort of runs of 2048 (key << 32 | index) items, eight per thread (sort_net.hpp)
//   k_rank_merge<GATHER>  (x levels) 8-way merge by ranking: an item's place = its place in its own run + binary searches in the
//                                      seven others, in flight together; two levels up to 131072 points; the last level writes
//                                      the permutation and gathers the coordinates itself
//   k_boxes                           tile boxes, super-tile boxes and top boxes (float atomics on min / max: order-free) in one
//
// Six launches for a scan of up to 131072 points, enqueued back to back (rounds 1-3: rocPRIM's sort through hipCUB -- eighteen
// launches, 170 us at 120k, the host's enqueue rate between them).  Larger clouds add one merge level per factor 8.  No atomics in
// the sort, no waiting between workgroups: every kernel ends by itself whatever the data.  A stable sort on the same keys gives the
// same permutation as the library did: every result downstream is unchanged bit for bit.  The same sort serves the voxel filter
// (three stable passes, one per 21-bit axis index of its keys), and a hand-written stable compaction (count / scan / scatter)
// serves the map slab and the voxel heads.
// Sizes (measured at 120k, rocprofv3): runs of 4096 on 30 workgroups 23.6 us + one 32-way merge 34 us (12 400 search steps per
// item's 31 searches; its run bookkeeping in vector registers) -> runs of 2048 on 59 workgroups + two 8-way levels over padded runs
// (189 search steps per item, four vector instructions each): see DESIGN.md.
#include <hip/hip_runtime.h>

#include <string>
#include <utility>

#include "hip_backend.hpp"
#include "kernels_common.hpp"
#include "sort_net.hpp"

namespace mola_icp_amd {

#define HIPCHK(expr)                                                                                          \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess) {                                                                               \
            (void)hipGetLastError(); /* (not latched for the next call's check) */                            \
            return fail(e_ == hipErrorOutOfMemory ? MOLA_ICP_E_OOM : MOLA_ICP_E_HIP,                          \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                                   \
        }                                                                                                     \
    } while (0)

constexpr int kSortRun = 2048;                 // items per sorted run (one workgroup of 256 threads, 18 KB of LDS)
constexpr int kSortThreads = kSortRun / sortnet::kE;
constexpr int kSortLog = 11;
constexpr int kSortFanLog = 3, kSortFan = 1 << kSortFanLog;   // runs merged per level: 131072 items with two levels, 1M with three
static_assert((1 << kSortLog) == kSortRun && kSortRun % 256 == 0, "run length (a merge workgroup of 256 never straddles runs)");

// ---- Hilbert keys of a cloud ---------------------------------------------------------------------------------------------
// The bounding box arrives as rows [n_rows][6] (min xyz, max xyz): k_bbox_partial's per-block rows, or one finished row.  Every
// workgroup reduces them itself (6 KB from the L2: cheaper than a launch in between); workgroup 0 publishes the box to the
// device block and -- straight into the pinned slot, no copy engine -- to the host, which looks at it at the next wait it makes
// anyway (HipWorkspace::check_bboxes: finite coordinates?).
__global__ __launch_bounds__(256) void k_hilbert_keys(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                                                      int n, const float* __restrict__ rows, int n_rows, float* __restrict__ box_dev,
                                                      float* __restrict__ box_host, unsigned int* __restrict__ keys)
{
    __shared__ float s_b[4][6];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float v[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int b = tid; b < n_rows; b += 256) {
#pragma unroll
        for (int k = 0; k < 6; ++k) v[k] = k < 3 ? fminf(v[k], rows[b * 6 + k]) : fmaxf(v[k], rows[b * 6 + k]);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float o = __shfl_xor(v[k], off);
            v[k] = k < 3 ? fminf(v[k], o) : fmaxf(v[k], o);
        }
        if (lane == 0) s_b[wave][k] = v[k];
    }
    __syncthreads();
    float box[6];
#pragma unroll
    for (int k = 0; k < 6; ++k)
        box[k] = k < 3 ? fminf(fminf(s_b[0][k], s_b[1][k]), fminf(s_b[2][k], s_b[3][k])) : fmaxf(fmaxf(s_b[0][k], s_b[1][k]), fmaxf(s_b[2][k], s_b[3][k]));
    if (blockIdx.x == 0 && tid < 6) {
        if (box_dev) box_dev[tid] = box[tid];
        if (box_host) box_host[tid] = box[tid];
    }
    const int i = (int)blockIdx.x * 256 + tid;
    if (i >= n) return;
    keys[i] = hilbert_key_in_box(box, x[i], y[i], z[i]);
}

// keys from an array -- optionally through an indirection (the second pass of a two-pass sort reads them in the first pass's order)
struct ArrayKeys {
    const unsigned int* k;
    const unsigned int* via;   // may be null
    __device__ __forceinline__ unsigned int key(int i) const { return via ? k[via[i]] : k[i]; }
};

// ---- runs of 2048: a bitonic network on (key << 32 | index), eight items per thread ----------------------------------------
// Three index bits of a register group are the slot number, so three compare-exchange stages cost one LDS round trip (23 round
// trips instead of 66 stage passes for 2048 items; padded slots: conflict-free in every layout).  The schedule is sort_net.hpp's
// run_phases -- compile-time, so every shift, mask and LDS offset below is a constant -- driven through this context; the stages
// are its group_stages.  tests/hosts/sort_net_test.cpp drives the same schedule and the same functions on the CPU.  Items past the
// end carry the key 0xffffffff and a higher index than any real item: they sort behind all of them -- and they ARE written: the
// merge's searches run over whole, padded runs (sort_net.hpp: merge_dest_padded).  Keys must be <= 0xfffffffe.
struct SortRunCtx {
    uint64_t v[sortnet::kE];
    uint64_t* lds;
    int tid;
    template <int M, int B0, int TOP, int CUR>
    __device__ __forceinline__ void group()
    {
        using namespace sortnet;
        if constexpr (B0 != CUR) {
#pragma unroll
            for (int e = 0; e < kE; ++e) lds[lds_slot(elem_index(tid, e, CUR))] = v[e];
            __syncthreads();
#pragma unroll
            for (int e = 0; e < kE; ++e) v[e] = lds[lds_slot(elem_index(tid, e, B0))];
            __syncthreads();
        }
        group_stages(v, tid, M, B0, TOP);
    }
};

__global__ __launch_bounds__(kSortThreads) void k_sort_runs(const ArrayKeys kg, int n, unsigned int* __restrict__ keys_out,
                                                            unsigned int* __restrict__ idx_out)
{
    using namespace sortnet;
    __shared__ uint64_t s_it[kSortRun + kSortRun / 8];
    SortRunCtx c;
    c.lds = s_it;
    c.tid = (int)threadIdx.x;
    const int base = (int)blockIdx.x * kSortRun;
#pragma unroll
    for (int e = 0; e < kE; ++e) {
        const int i = base + c.tid * kE + e;   // (layout b0 = 0: eight consecutive items per thread)
        const unsigned int key = i < n ? kg.key(i) : 0xffffffffu;
        c.v[e] = ((uint64_t)key << 32) | (unsigned int)i;
    }
    run_phases<SortRunCtx, 1, kSortLog>(c);
    // (every phase ends in the layout b0 = 0 again: slot e of thread tid = sorted position 8 tid + e of the run)
#pragma unroll
    for (int e = 0; e < kE; ++e) {
        const int i = base + c.tid * kE + e;   // (whole runs: the arrays are padded to them)
        keys_out[i] = (unsigned int)(c.v[e] >> 32);
        idx_out[i] = (unsigned int)c.v[e];
    }
}

// ---- one merge level: every item finds its place among the <= 8 runs of its group (sort_net.hpp: merge_dest_padded) ----------
// The seven searches of an item run together: log2(L) + 1 rounds of independent loads instead of seven chains of dependent ones.
// The item's own run comes from the workgroup index, so run bases are scalar and the thresholds are set once; the runs are padded,
// so a search step is add / load / compare / select.  The 64 lanes of a wave hold consecutive items of one sorted run: their search
// positions in another run stay within a few cache lines of each other.
// Not the last level: the slots [n, n_pad_out) of keys_out get the padding key (the next level's runs are whole).
// GATHER = the last level of a cloud's sort: the destination gets the original index (perm) and the point itself; the slots
// [n, n_pad_out) are filled with padding points; the first threads also reset the super-tile and top boxes k_boxes accumulates into.
template <bool GATHER>
__global__ __launch_bounds__(256) void k_rank_merge(const unsigned int* __restrict__ keys_in, const unsigned int* __restrict__ idx_in,
                                                    int n, long long n_pad_in, int n_pad_out, int logL, unsigned int* __restrict__ keys_out,
                                                    unsigned int* __restrict__ idx_out, const float* __restrict__ gx,
                                                    const float* __restrict__ gy, const float* __restrict__ gz,
                                                    float* __restrict__ sx, float* __restrict__ sy, float* __restrict__ sz,
                                                    float* __restrict__ sbox, int n_super, float* __restrict__ ubox, int n_top)
{
    const int p = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (GATHER && sbox) {   // (6 n_super <= n_pad_out: every box has a thread)
        if (p < 6 * n_super) sbox[p] = p < 3 * n_super ? INFINITY : -INFINITY;
        if (p < 6 * n_top) ubox[p] = p < 3 * n_top ? INFINITY : -INFINITY;
    }
    if (p >= n) {
        if (p < n_pad_out) {
            if (GATHER) {
                sx[p] = sy[p] = sz[p] = 1.0e18f;   // padding members: d2 ~ 3e36, never a neighbour
                idx_out[p] = 0x7fffffffu;          // (perm has n_pad_out entries)
            } else {
                keys_out[p] = 0xffffffffu;
            }
        }
        return;
    }
    const int a = (int)(((unsigned int)blockIdx.x * 256u) >> logL);   // (scalar: the workgroup's run)
    const int dest = sortnet::merge_dest_padded<kSortFan>(keys_in, n_pad_in, logL, p, a);
    const unsigned int idx = idx_in[p];
    idx_out[dest] = idx;
    if (GATHER) {
        sx[dest] = gx[idx]; sy[dest] = gy[idx]; sz[dest] = gz[idx];
        if (keys_out) keys_out[dest] = keys_in[p];   // (a map keeps its sorted keys: a query's place in it is a binary search away)
    } else {
        keys_out[dest] = keys_in[p];
    }
}

// Scratch of one sort: two (keys, indices) array pairs, each padded to whole runs of the LAST level's input run length (every
// level's input must consist of whole runs; the last level's are the longest)
struct SortScratch {
    unsigned int *kA, *iA, *kB, *iB;
};
static size_t sort_padded_items(size_t n)
{
    unsigned long long L = kSortRun;
    while (L * kSortFan < n) L *= kSortFan;
    return (size_t)((n + L - 1) / L * L);
}
static size_t sort_scratch_bytes(size_t n) { return 4 * ((sizeof(unsigned int) * sort_padded_items(n) + 255) / 256 * 256); }
static SortScratch sort_scratch_at(char* base, size_t n)
{
    const size_t a = (sizeof(unsigned int) * sort_padded_items(n) + 255) / 256 * 256;
    return SortScratch{reinterpret_cast<unsigned int*>(base), reinterpret_cast<unsigned int*>(base + a),
                       reinterpret_cast<unsigned int*>(base + 2 * a), reinterpret_cast<unsigned int*>(base + 3 * a)};
}

// runs -> merge levels.  The last level writes order_out (the original index of every sorted position; n_padded entries when
// it gathers) and, with gather arguments, the points.  Keys must be <= 0xfffffffe.
static int sort_by_key(hipStream_t stream, const ArrayKeys& kg, size_t n, const SortScratch& s, unsigned int* order_out, size_t n_padded,
                       const float* gx, const float* gy, const float* gz, float* sx, float* sy, float* sz, float* sbox, int n_super,
                       float* ubox, int n_top, unsigned int* keys_sorted_out = nullptr /*with the gather: the keys in sorted order (n entries)*/)
{
    const int ni = (int)n;
    const unsigned n_runs = (unsigned)((n + kSortRun - 1) / kSortRun);
    hipLaunchKernelGGL(k_sort_runs, dim3(n_runs), dim3(kSortThreads), 0, stream, kg, ni, s.kA, s.iA);
    HIPCHK(hipGetLastError());
    unsigned int *kin = s.kA, *iin = s.iA, *kout = s.kB, *iout = s.iB;
    int logL = kSortLog;
    for (unsigned long long L = kSortRun;; L *= kSortFan, logL += kSortFanLog) {
        const bool last = L * kSortFan >= (unsigned long long)n;
        const long long n_pad_in = (long long)((n + L - 1) / L * L);
        if (last && gx) {
            hipLaunchKernelGGL((k_rank_merge<true>), dim3((unsigned)((n_padded + 255) / 256)), dim3(256), 0, stream, kin, iin, ni, n_pad_in,
                               (int)n_padded, logL, keys_sorted_out, order_out, gx, gy, gz, sx, sy, sz, sbox, n_super, ubox, n_top);
        } else {
            // (not the last level: the output is the next level's input -- padded to whole runs of L * fan-in)
            const unsigned long long Ln = L * kSortFan;
            const size_t n_pad_out = last ? n : (size_t)((n + Ln - 1) / Ln * Ln);
            hipLaunchKernelGGL((k_rank_merge<false>), dim3((unsigned)((n_pad_out + 255) / 256)), dim3(256), 0, stream, kin, iin, ni, n_pad_in,
                               (int)n_pad_out, logL, kout, last ? order_out : iout, (const float*)nullptr, (const float*)nullptr,
                               (const float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, 0, (float*)nullptr, 0);
        }
        HIPCHK(hipGetLastError());
        if (last) break;
        std::swap(kin, kout);
        std::swap(iin, iout);
    }
    return MOLA_ICP_OK;
}

// sorted copies: sxyz = 3 * M_padded floats (SoA), perm[M_padded] = original index of sorted position (0x7fffffff in the padding).
// box_rows: [n_box_rows][6] on the device (k_bbox_partial's rows, or one finished row); the finished box goes to box_dev / box_host
// (either may be null).  sbox / ubox (may be null): [6][n_super] / [6][n_top], reset for k_boxes.
int hilbert_sort_points(hipStream_t stream, const float* gx, const float* gy, const float* gz, size_t M, size_t M_padded,
                        const float* box_rows, int n_box_rows, float* box_dev, float* box_host, DevBuf& scratch, float* sxyz, int* perm,
                        float* sbox, int n_super, float* ubox, int n_top, unsigned int* keys_sorted /*M entries, may be null*/)
{
    if (M == 0) return MOLA_ICP_OK;
    if (M_padded > (size_t)0x7fffff00) return fail(MOLA_ICP_E_BADARG, "cloud too large for the 32-bit sort");
    const size_t a = (sizeof(unsigned int) * M + 255) / 256 * 256;
    int rc = scratch.reserve(a + sort_scratch_bytes(M) + 256);
    if (rc) return rc;
    unsigned int* keys = scratch.as<unsigned int>();
    const SortScratch s = sort_scratch_at(scratch.as<char>() + a, M);
    hipLaunchKernelGGL(k_hilbert_keys, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, stream, gx, gy, gz, (int)M, box_rows, n_box_rows, box_dev,
                       box_host, keys);
    HIPCHK(hipGetLastError());
    return sort_by_key(stream, ArrayKeys{keys, nullptr}, M, s, reinterpret_cast<unsigned int*>(perm), M_padded, gx, gy, gz, sxyz,
                       sxyz + M_padded, sxyz + 2 * M_padded, sbox, n_super, ubox, n_top, keys_sorted);
}

// ---- the three box levels of a sorted map, one launch -------------------------------------------------------------------
// Four workgroups per super-tile (64 tiles of 32 points): a 32-lane group per tile (a point per lane, coalesced); the super-tile's
// box and the top box (64 super-tiles) by float atomics on min / max -- the result does not depend on the order, so the boxes are
// the same bits whichever workgroup arrives first.  SoA [6][n] per level; empty boxes are (+inf, -inf) (the sort's last level
// left the two upper levels so).  One workgroup per super-tile walking its 64 tiles took 9.5 us at 120k (64 workgroups).
__device__ __forceinline__ void atomic_min_f32(float* a, float v)
{
    if (v >= 0.f) atomicMin(reinterpret_cast<int*>(a), __float_as_int(v));
    else atomicMax(reinterpret_cast<unsigned int*>(a), __float_as_uint(v));
}
__device__ __forceinline__ void atomic_max_f32(float* a, float v)
{
    if (v >= 0.f) atomicMax(reinterpret_cast<int*>(a), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned int*>(a), __float_as_uint(v));
}

__global__ __launch_bounds__(256) void k_boxes(const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz,
                                               int M, int n_tiles_p, int n_super, int n_top, float* __restrict__ tbox,
                                               float* __restrict__ sbox, float* __restrict__ ubox,
                                               const float* __restrict__ cloud_box /*the whole cloud's box, or null*/)
{
    // One top box = the cloud's own bounding box (the same min / max over the same points), which the keys kernel left on the
    // device: 256 workgroups' atomics on six addresses retire one after another (~13 ns each) and were a quarter of this launch.
    const bool top_from_cloud = n_top == 1 && cloud_box != nullptr;
    if (top_from_cloud && blockIdx.x == 0 && threadIdx.x < 6) ubox[threadIdx.x] = cloud_box[threadIdx.x];
    const int s = (int)blockIdx.x >> 2, quarter = (int)blockIdx.x & 3;
    const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, l = lane & 31;
    float smn[3] = {INFINITY, INFINITY, INFINITY}, smx[3] = {-INFINITY, -INFINITY, -INFINITY};
    float px[2], py[2], pz[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {   // (all loads first)
        const int j = (s * 64 + quarter * 16 + wave * 4 + it * 2 + half) * 32 + l;
        px[it] = py[it] = pz[it] = 0.f;
        if (j < M) { px[it] = sx[j]; py[it] = sy[j]; pz[it] = sz[j]; }
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int t = s * 64 + quarter * 16 + wave * 4 + it * 2 + half;
        const bool in = (t * 32 + l) < M;
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        if (in) { mn[0] = mx[0] = px[it]; mn[1] = mx[1] = py[it]; mn[2] = mx[2] = pz[it]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) {
                mn[k] = fminf(mn[k], __shfl_xor(mn[k], off, 32));
                mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], off, 32));
            }
            smn[k] = fminf(smn[k], mn[k]);
            smx[k] = fmaxf(smx[k], mx[k]);
        }
        if (l < 3) tbox[l * n_tiles_p + t] = l == 0 ? mn[0] : l == 1 ? mn[1] : mn[2];
        else if (l < 6) tbox[l * n_tiles_p + t] = l == 3 ? mx[0] : l == 4 ? mx[1] : mx[2];
    }
    __shared__ float s_w[4][6];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        smn[k] = fminf(smn[k], __shfl_xor(smn[k], 32));
        smx[k] = fmaxf(smx[k], __shfl_xor(smx[k], 32));
        if (lane == 0) { s_w[wave][k] = smn[k]; s_w[wave][3 + k] = smx[k]; }
    }
    __syncthreads();
    if (tid < 6) {
        float r = s_w[0][tid];
        for (int w = 1; w < 4; ++w) r = tid < 3 ? fminf(r, s_w[w][tid]) : fmaxf(r, s_w[w][tid]);
        if (tid < 3) {
            if (r < INFINITY) { atomic_min_f32(sbox + tid * n_super + s, r); if (!top_from_cloud) atomic_min_f32(ubox + tid * n_top + (s >> 6), r); }
        } else {
            if (r > -INFINITY) { atomic_max_f32(sbox + tid * n_super + s, r); if (!top_from_cloud) atomic_max_f32(ubox + tid * n_top + (s >> 6), r); }
        }
    }
}

// tbox [6][n_tiles_p], sbox [6][n_super], ubox [6][n_top] of the sorted cloud (sbox / ubox must hold (+inf, -inf): the sort's last
// level left them so)
int boxes_of_sorted(hipStream_t stream, const float* sxyz, size_t M, size_t M_padded, int n_tiles_p, int n_super, int n_top, float* tbox,
                    float* sbox, float* ubox, const float* cloud_box)
{
    if (n_super <= 0) return MOLA_ICP_OK;
    hipLaunchKernelGGL(k_boxes, dim3((unsigned)n_super * 4u), dim3(256), 0, stream, sxyz, sxyz + M_padded, sxyz + 2 * M_padded, (int)M, n_tiles_p,
                       n_super, n_top, tbox, sbox, ubox, cloud_box);
    HIPCHK(hipGetLastError());
    return MOLA_ICP_OK;
}

// ---- stable compaction: sel[k] = index of the k-th item with pred(i), ascending; count / scan / scatter ----------------------
constexpr int kCompPer = 8;   // consecutive items per thread: 2048 per workgroup

struct InBoxPred {
    const float *x, *y, *z;
    float lx, ly, lz, hx, hy, hz;
    __device__ __forceinline__ bool operator()(int i) const
    {
        const float a = x[i], b = y[i], c = z[i];
        return a >= lx && a <= hx && b >= ly && b <= hy && c >= lz && c <= hz;
    }
};
struct HeadPred {   // first item of a run of equal 64-bit keys
    const unsigned long long* keys;
    __device__ __forceinline__ bool operator()(int i) const { return i == 0 || keys[i] != keys[i - 1]; }
};

template <class Pred>
__global__ __launch_bounds__(256) void k_comp_count(const Pred pr, int n, unsigned int* __restrict__ counts)
{
    const int base = ((int)blockIdx.x * 256 + (int)threadIdx.x) * kCompPer;
    unsigned int c = 0;
#pragma unroll
    for (int e = 0; e < kCompPer; ++e) c += (base + e < n && pr(base + e)) ? 1u : 0u;
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    __shared__ unsigned int s_c[4];
    if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = s_c[0] + s_c[1] + s_c[2] + s_c[3];
}

// exclusive scan of counts[0, nb) in place; counts[nb] = the total (one workgroup: nb is n / 2048)
__global__ __launch_bounds__(1024) void k_comp_scan(unsigned int* __restrict__ counts, int nb)
{
    __shared__ unsigned int s_w[16];
    __shared__ unsigned int s_carry;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry = 0u;
    __syncthreads();
    for (int b0 = 0; b0 < nb; b0 += 1024) {
        const int i = b0 + tid;
        const unsigned int v = i < nb ? counts[i] : 0u;
        unsigned int inc = v;   // inclusive scan inside the wave
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned int o = __shfl_up(inc, off);
            if (lane >= off) inc += o;
        }
        if (lane == 63) s_w[wave] = inc;
        __syncthreads();
        unsigned int before = s_carry;
        for (int w = 0; w < wave; ++w) before += s_w[w];
        if (i < nb) counts[i] = before + inc - v;
        __syncthreads();
        if (tid == 1023) s_carry = before + inc;
        __syncthreads();
    }
    if (tid == 0) counts[nb] = s_carry;
}

template <class Pred>
__global__ __launch_bounds__(256) void k_comp_scatter(const Pred pr, int n, const unsigned int* __restrict__ offsets, int* __restrict__ sel)
{
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int base = ((int)blockIdx.x * 256 + tid) * kCompPer;
    bool f[kCompPer];
    unsigned int c = 0;
#pragma unroll
    for (int e = 0; e < kCompPer; ++e) { f[e] = base + e < n && pr(base + e); c += f[e] ? 1u : 0u; }
    unsigned int inc = c;
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned int o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
    }
    __shared__ unsigned int s_w[4];
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    unsigned int at = offsets[blockIdx.x] + inc - c;
    for (int w = 0; w < wave; ++w) at += s_w[w];
#pragma unroll
    for (int e = 0; e < kCompPer; ++e)
        if (f[e]) sel[at++] = base + e;
}

// sel[0, *n_kept_host) on the device; waits for the count.  `counts` needs n / 2048 + 2 words.
template <class Pred>
static int compact_indices(hipStream_t stream, const Pred& pr, size_t n, unsigned int* counts, int* sel, size_t* n_kept_host)
{
    const int ni = (int)n;
    const int nb = (int)((n + 256 * kCompPer - 1) / (256 * kCompPer));
    hipLaunchKernelGGL((k_comp_count<Pred>), dim3((unsigned)nb), dim3(256), 0, stream, pr, ni, counts);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_comp_scan, dim3(1), dim3(1024), 0, stream, counts, nb);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL((k_comp_scatter<Pred>), dim3((unsigned)nb), dim3(256), 0, stream, pr, ni, counts, sel);
    HIPCHK(hipGetLastError());
    unsigned int h = 0;
    HIPCHK(hipMemcpyAsync(&h, counts + nb, sizeof h, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    *n_kept_host = (size_t)h;
    return MOLA_ICP_OK;
}

// ---- row e: the part of a map a query shard can reach ------------------------------------------------------
// Stable compaction of the points inside an axis-aligned box: sel[k] = original index of the k-th kept point (ascending),
// *n_kept_host = their number.  The caller gathers the coordinates.
__global__ __launch_bounds__(256) void k_gather_by_index(const float* __restrict__ x, const float* __restrict__ y,
                                                         const float* __restrict__ z, const int* __restrict__ sel, int n,
                                                         float* __restrict__ ox, float* __restrict__ oy, float* __restrict__ oz)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int j = sel[i];
    ox[i] = x[j]; oy[i] = y[j]; oz[i] = z[j];
}

int select_in_box(hipStream_t stream, const float* x, const float* y, const float* z, size_t n, const float lo[3], const float hi[3],
                  DevBuf& scratch, int* sel, size_t* n_kept_host)
{
    *n_kept_host = 0;
    if (n == 0) return MOLA_ICP_OK;
    int rc = scratch.reserve(sizeof(unsigned int) * (n / (256 * kCompPer) + 4));
    if (rc) return rc;
    const InBoxPred pr{x, y, z, lo[0], lo[1], lo[2], hi[0], hi[1], hi[2]};
    return compact_indices(stream, pr, n, scratch.as<unsigned int>(), sel, n_kept_host);
}

int gather_by_index(hipStream_t stream, const float* x, const float* y, const float* z, const int* sel, size_t n, float* ox, float* oy,
                    float* oz)
{
    if (n == 0) return MOLA_ICP_OK;
    hipLaunchKernelGGL(k_gather_by_index, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, y, z, sel, (int)n, ox, oy, oz);
    HIPCHK(hipGetLastError());
    return MOLA_ICP_OK;
}

// ---- row f4: voxel-grid downsample (one centroid per occupied voxel) ------------------------------------
// The reference decimates clouds before the ICP with mp2p_icp_filters (src/LidarOdometry.cpp:215-224; voxel
// parameters include/mola-fe-lidar/LidarOdometry.h:76-80, params/kitti-default.yaml:25-32) [EXT: that library is
// not in the tree; this is the plain voxel-centroid filter].  Key = (ix, iy, iz) of floor((p - min) * (1/size)) in
// fp32, 21 bits per axis; a stable sort by key -- three stable passes of the 32-bit sort above, iz then iy then ix (every
// pass's keys < 2^21: inside the sort's key contract); the voxel heads compacted; one thread per voxel sums its run in fp64
// (ascending original index) -> centroid.  Output order = ascending key.
__global__ __launch_bounds__(256) void k_voxel_keys(const float* __restrict__ x, const float* __restrict__ y,
                                                    const float* __restrict__ z, int n, float ox, float oy, float oz,
                                                    float inv, unsigned long long* __restrict__ keys, unsigned int* __restrict__ kx,
                                                    unsigned int* __restrict__ ky, unsigned int* __restrict__ kz)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long ix = (unsigned long long)fminf(floorf((x[i] - ox) * inv), 2097151.f);
    const unsigned long long iy = (unsigned long long)fminf(floorf((y[i] - oy) * inv), 2097151.f);
    const unsigned long long iz = (unsigned long long)fminf(floorf((z[i] - oz) * inv), 2097151.f);
    keys[i] = (ix << 42) | (iy << 21) | iz;
    kx[i] = (unsigned int)ix; ky[i] = (unsigned int)iy; kz[i] = (unsigned int)iz;
}

// out[r] = a[b[r]] (the order of two stable passes, composed); with keys: the sorted keys alongside
__global__ __launch_bounds__(256) void k_compose(const unsigned int* __restrict__ a, const unsigned int* __restrict__ b, int n,
                                                 unsigned int* __restrict__ out, const unsigned long long* __restrict__ keys,
                                                 unsigned long long* __restrict__ keys_sorted)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const unsigned int o = a[b[r]];
    out[r] = o;
    if (keys) keys_sorted[r] = keys[o];
}

__global__ __launch_bounds__(256) void k_voxel_centroids(const float* __restrict__ x, const float* __restrict__ y,
                                                         const float* __restrict__ z, const unsigned long long* __restrict__ keys,
                                                         const unsigned int* __restrict__ order, const int* __restrict__ heads, int n_heads, int n,
                                                         float* __restrict__ ox, float* __restrict__ oy, float* __restrict__ oz)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n_heads) return;
    const int i = heads[s];
    double sx = 0, sy = 0, sz = 0;
    int c = 0;
    for (int j = i; j < n && keys[j] == keys[i]; ++j) {
        const unsigned int o = order[j];
        sx += x[o]; sy += y[o]; sz += z[o];
        ++c;
    }
    ox[s] = (float)(sx / c); oy[s] = (float)(sy / c); oz[s] = (float)(sz / c);
}

// device in (x,y,z,n) -> device out (capacity floats each); *n_out_host = number of voxels (may exceed capacity)
int voxel_downsample_device(hipStream_t stream, const float* x, const float* y, const float* z, size_t n, const float bbox[6],
                            float voxel, DevBuf& scratch, float* out_x, float* out_y, float* out_z, size_t capacity,
                            size_t* n_out_host)
{
    *n_out_host = 0;
    if (n == 0) return MOLA_ICP_OK;
    const int ni = (int)n;
    const float inv = 1.0f / voxel;
    for (int k = 0; k < 3; ++k)
        if (!((bbox[3 + k] - bbox[k]) * inv < 2097151.f))
            return fail(MOLA_ICP_E_BADARG, "voxel size too small for the cloud extent (more than 2^21 voxels per axis)");
    const size_t a8 = (sizeof(unsigned long long) * n + 255) / 256 * 256, a4 = (sizeof(int) * n + 255) / 256 * 256;
    const size_t cnt_bytes = (sizeof(unsigned int) * (n / (256 * kCompPer) + 4) + 255) / 256 * 256;
    int rc = scratch.reserve(2 * a8 + 8 * a4 + cnt_bytes + sort_scratch_bytes(n) + 512);
    if (rc) return rc;
    char* base = scratch.as<char>();
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(base);
    unsigned long long* keys_sorted = reinterpret_cast<unsigned long long*>(base + a8);
    unsigned int* kx = reinterpret_cast<unsigned int*>(base + 2 * a8);
    unsigned int* ky = reinterpret_cast<unsigned int*>(base + 2 * a8 + a4);
    unsigned int* kz = reinterpret_cast<unsigned int*>(base + 2 * a8 + 2 * a4);
    unsigned int* o1 = reinterpret_cast<unsigned int*>(base + 2 * a8 + 3 * a4);
    unsigned int* o2 = reinterpret_cast<unsigned int*>(base + 2 * a8 + 4 * a4);
    unsigned int* o12 = reinterpret_cast<unsigned int*>(base + 2 * a8 + 5 * a4);
    unsigned int* order = reinterpret_cast<unsigned int*>(base + 2 * a8 + 6 * a4);
    int* heads = reinterpret_cast<int*>(base + 2 * a8 + 7 * a4);
    unsigned int* counts = reinterpret_cast<unsigned int*>(base + 2 * a8 + 8 * a4);
    const SortScratch ss = sort_scratch_at(base + 2 * a8 + 8 * a4 + cnt_bytes, n);
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_voxel_keys, dim3(nb), dim3(256), 0, stream, x, y, z, ni, bbox[0], bbox[1], bbox[2], inv, keys, kx, ky, kz);
    HIPCHK(hipGetLastError());
    // least significant index first; every later pass is stable over the order the earlier ones left = stable by the whole key
    auto plain_sort = [&](const ArrayKeys& kg, unsigned int* out) {
        return sort_by_key(stream, kg, n, ss, out, n, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0);
    };
    if ((rc = plain_sort(ArrayKeys{kz, nullptr}, o1))) return rc;
    if ((rc = plain_sort(ArrayKeys{ky, o1}, o2))) return rc;
    hipLaunchKernelGGL(k_compose, dim3(nb), dim3(256), 0, stream, o1, o2, ni, o12, (const unsigned long long*)nullptr, (unsigned long long*)nullptr);
    HIPCHK(hipGetLastError());
    if ((rc = plain_sort(ArrayKeys{kx, o12}, o2))) return rc;
    hipLaunchKernelGGL(k_compose, dim3(nb), dim3(256), 0, stream, o12, o2, ni, order, keys, keys_sorted);
    HIPCHK(hipGetLastError());
    size_t n_heads = 0;
    if ((rc = compact_indices(stream, HeadPred{keys_sorted}, n, counts, heads, &n_heads))) return rc;   // (waits for the count)
    const size_t n_emit = n_heads < capacity ? n_heads : capacity;
    if (n_emit) {
        hipLaunchKernelGGL(k_voxel_centroids, dim3((unsigned)((n_emit + 255) / 256)), dim3(256), 0, stream, x, y, z, keys_sorted, order, heads,
                           (int)n_emit, ni, out_x, out_y, out_z);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(stream));
    }
    *n_out_host = n_heads;
    return MOLA_ICP_OK;
}

}  // namespace mola_icp_amd
