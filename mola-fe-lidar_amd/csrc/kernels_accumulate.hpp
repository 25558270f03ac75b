// kernels_accumulate.hpp -- row a8: weighted centroid / covariance accumulation and its fixed-order reduction
// Device code of the ICP core for gfx950; included by hip_backend.hip only (one translation unit: the kernels are
// launched from there).  Numeric contract and data layout: hip_backend.hip / DESIGN.md.
#pragma once
#include "kernels_tiled.hpp"

namespace mola_icp_amd {

// ---- accumulation (row a8) ---------------------------------------------------------
struct AccArgs {
    const float *lx, *ly, *lz, *gx, *gy, *gz;
    const float *nx, *ny, *nz;  // non-null: the neighbour's coordinates per query (tiled matchers store them) -- no gather
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

// block `bx` of `nblocks` over one problem's pairing -> one row of partial sums (the summation order depends on
// (N, nblocks) only: the single-problem and the batched launch produce the same bits)
__device__ __forceinline__ void accumulate_rows(const AccArgs& a, int bx, int nblocks, double* __restrict__ partials)
{
    double s[kNAcc];
#pragma unroll
    for (int k = 0; k < kNAcc; ++k) s[k] = 0.0;
    const int stride = nblocks * kAccThreads;
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
    for (int i = bx * kAccThreads + threadIdx.x; i < a.N; i += 2 * stride) {
        const int i2 = i + stride;
        const bool in2 = i2 < a.N;
        const int ic2 = in2 ? i2 : i;
        const int jA = a.idx[i], jB = in2 ? a.idx[ic2] : -1;
        const unsigned char oA = a.outlier[i], oB = a.outlier[ic2];
        const float lA0 = a.lx[i], lA1 = a.ly[i], lA2 = a.lz[i], dA = a.d2[i];
        const float lB0 = a.lx[ic2], lB1 = a.ly[ic2], lB2 = a.lz[ic2], dB = a.d2[ic2];
        float gA0, gA1, gA2, gB0, gB1, gB2;
        if (a.nx) {  // (uniform) same values as the gather: the matcher copied them from the map
            gA0 = a.nx[i]; gA1 = a.ny[i]; gA2 = a.nz[i];
            gB0 = a.nx[ic2]; gB1 = a.ny[ic2]; gB2 = a.nz[ic2];
        } else {
            const int jcA = jA >= 0 ? jA : 0, jcB = jB >= 0 ? jB : 0;
            gA0 = a.gx[jcA]; gA1 = a.gy[jcA]; gA2 = a.gz[jcA];
            gB0 = a.gx[jcB]; gB1 = a.gy[jcB]; gB2 = a.gz[jcB];
        }
        element(i, jA, oA, lA0, lA1, lA2, gA0, gA1, gA2, dA);
        element(i2, jB, oB, lB0, lB1, lB2, gB0, gB1, gB2, dB);
    }
    // fixed-order block sum -> one row per block
    static_assert(kAccThreads == 256, "block_sum_256");
    block_sum_256<kNAcc>(s, partials + (size_t)bx * kNAcc);
}

__global__ __launch_bounds__(kAccThreads) void k_accumulate(AccArgs a, double* __restrict__ partials)
{
    accumulate_rows(a, (int)blockIdx.x, (int)gridDim.x, partials);
}

// K problems in one launch (grid = (max rows, K)): per problem the SAME partition as its own k_accumulate launch
constexpr int kAccMaxBatch = 12;   // 12 x sizeof(AccArgs) stays below the 4 KB of kernel arguments
struct AccBatch {
    AccArgs a[kAccMaxBatch];
    int nblocks[kAccMaxBatch];
    double* partials[kAccMaxBatch];
};
__global__ __launch_bounds__(kAccThreads) void k_accumulate_batch(const AccBatch b)
{
    const int y = (int)blockIdx.y;
    const int nb = b.nblocks[y];
    if ((int)blockIdx.x >= nb) return;
    accumulate_rows(b.a[y], (int)blockIdx.x, nb, b.partials[y]);
}

// sums the per-block rows in a fixed order: 32 interleaved slices per accumulator, then the slices in order
constexpr int kRedSlices = 32;
__device__ __forceinline__ void reduce_rows(const double* __restrict__ partials, int nblocks, double* __restrict__ acc,
                                            double* __restrict__ host_out, unsigned long long seq)
{
    __shared__ double sm[kRedSlices][kNAcc];
    const int k = threadIdx.x % kNAcc, sl = threadIdx.x / kNAcc;
    // rows sl, sl + 32, sl + 64, ... added in order.  One block, latency-bound: all loads of up to 16 rows per thread are
    // issued before the first add (a load-add-load chain over the 25 rows of a 100k-point cloud took 11 us)
    double v = 0.0;
    int b = sl;
    for (; b < nblocks; b += 16 * kRedSlices) {
        double t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int r = b + u * kRedSlices;
            t[u] = r < nblocks ? partials[(size_t)r * kNAcc + k] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (b + u * kRedSlices < nblocks) v += t[u];
    }
    sm[sl][k] = v;
    __syncthreads();
    if (threadIdx.x < kNAcc) {
        double t = 0.0;
        for (int s = 0; s < kRedSlices; ++s) t += sm[s][threadIdx.x];
        acc[threadIdx.x] = t;
        if (host_out) host_out[threadIdx.x] = t;  // straight into the host's pinned block: no copy engine, no extra launch gap
    }
    if (host_out) {  // publish: data first, then the sequence number the host spins on
        // (only the waves that wrote to the host block fence at system scope: the fence is an L2 write-back per wave, and
        //  twelve waves doing it one after another was most of this kernel)
        if (threadIdx.x < ((kNAcc + 63) / 64) * 64) __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
            reinterpret_cast<volatile unsigned long long*>(host_out)[kNAcc + 6] = seq;  // (slots 24..29 serve other read-backs)
            __threadfence_system();
        }
    }
}

__global__ __launch_bounds__(kNAcc * kRedSlices) void k_reduce_partials(const double* __restrict__ partials, int nblocks,
                                                                        double* __restrict__ acc,
                                                                        double* __restrict__ host_out /*pinned, may be null*/,
                                                                        unsigned long long seq)
{
    reduce_rows(partials, nblocks, acc, host_out, seq);
    // the matcher's work-queue / kept / redo counters sit right behind the block: leave them zero for its next launch
    if (threadIdx.x == kNAcc) { acc[kNAcc] = 0.0; acc[kNAcc + 1] = 0.0; }
    if (threadIdx.x >= 32 && threadIdx.x < 32 + 2 * kQueues)  // the tiled matcher's work-queue counters
        reinterpret_cast<unsigned int*>(acc + kNAcc + 8)[(threadIdx.x - 32) * kQueueStride] = 0u;
}

// The persistent matcher's item rows (item_row_mfma: one row per 64-query item, 15 625 of them at 1M queries): block g of G
// sums the rows [g n / G, (g + 1) n / G) in reduce_rows' fixed order into row g of `part` (stride kNAcc) and, single GPU,
// publishes it at host_out + 32 g with its own sequence flag -- the host adds the G rows in order itself (a second,
// one-block launch for a few dozen rows would cost more than the sums).  Block 0 also leaves the matcher's counters zero.
// `final_acc` (sharded over RCCL: the collective runs on ONE device block): the block that finishes LAST adds the G rows in order
// 0 .. G-1 -- the same bits whichever block that is -- into final_acc[0, kNAcc): no second, one-block launch in front of ncclAllReduce.
// `done` counts the finished blocks (a device word that is zero between launches: the last block leaves it so).
constexpr int kItemRedBlocks = 32;
__global__ __launch_bounds__(kNAcc * kRedSlices) void k_reduce_items(const double* __restrict__ rows, int n_rows, double* __restrict__ part,
                                                                     double* __restrict__ host_out /*pinned, may be null*/,
                                                                     unsigned long long seq, double* __restrict__ counters_block /*acc_dev_*/,
                                                                     double* __restrict__ final_acc = nullptr, unsigned int* __restrict__ done = nullptr)
{
    const int g = (int)blockIdx.x, G = (int)gridDim.x;
    const int lo = (int)(((long long)g * n_rows) / G), hi = (int)(((long long)(g + 1) * n_rows) / G);
    reduce_rows(rows + (size_t)lo * kNAcc, hi - lo, part + (size_t)g * kNAcc, host_out ? host_out + 32 * (size_t)g : (double*)nullptr, seq);
    if (g == 0) {  // (as k_reduce_partials: the matcher's kept / redo counters and its work-queue counters)
        if (threadIdx.x == kNAcc) { counters_block[kNAcc] = 0.0; counters_block[kNAcc + 1] = 0.0; }
        if (threadIdx.x >= 32 && threadIdx.x < 32 + 2 * kQueues)
            reinterpret_cast<unsigned int*>(counters_block + kNAcc + 8)[(threadIdx.x - 32) * kQueueStride] = 0u;
    }
    if (final_acc) {
        __shared__ unsigned int s_ticket;
        __threadfence();        // this block's row is visible device-wide before its ticket is
        __syncthreads();
        if (threadIdx.x == 0) s_ticket = atomicAdd(done, 1u);
        __syncthreads();
        if (s_ticket == (unsigned int)(G - 1)) {   // (block-uniform)
            __threadfence();    // ... and the other blocks' rows are read behind their tickets
            if (threadIdx.x < kNAcc) {
                double t = 0.0;
                for (int r = 0; r < G; ++r) t += __builtin_nontemporal_load(part + (size_t)r * kNAcc + threadIdx.x);
                final_acc[threadIdx.x] = t;
            }
            if (threadIdx.x == 0) *done = 0u;
        }
    }
}

// ... and for K problems in one launch (grid = (kItemRedBlocks, K)): per problem the partition and the order of its own
// k_reduce_items launch (the G rows land at host_out + 32 (kItemRedBlocks slot + g), each with its sequence flag)
__host__ __device__ __forceinline__ int item_red_blocks(int n_rows)
{
    const int g = (n_rows + 63) / 64;
    return g < 1 ? 1 : (g > kItemRedBlocks ? kItemRedBlocks : g);
}
struct ReduceItemsBatch {
    const double* rows[kAccMaxBatch];
    int n_rows[kAccMaxBatch];
    int slot[kAccMaxBatch];
};
__global__ __launch_bounds__(kNAcc * kRedSlices) void k_reduce_items_batch(const ReduceItemsBatch b, double* __restrict__ part,
                                                                           double* __restrict__ host_out, unsigned long long seq)
{
    const int y = (int)blockIdx.y, g = (int)blockIdx.x, n_rows = b.n_rows[y], G = item_red_blocks(n_rows);
    if (g >= G) return;
    const int lo = (int)(((long long)g * n_rows) / G), hi = (int)(((long long)(g + 1) * n_rows) / G);
    const size_t o = (size_t)kItemRedBlocks * (size_t)b.slot[y] + (size_t)g;
    reduce_rows(b.rows[y] + (size_t)lo * kNAcc, hi - lo, part + o * kNAcc, host_out + 32 * o, seq);
}

// K problems: block y reduces problem y's rows into acc + 32 y and publishes them at host_out + 32 y (flag in slot 30
// of that stride)
struct ReduceBatch {
    const double* partials[kAccMaxBatch];
    int nblocks[kAccMaxBatch];
    int slot[kAccMaxBatch];              // the problem's index in acc / host_out / counters
};
__global__ __launch_bounds__(kNAcc * kRedSlices) void k_reduce_partials_batch(const ReduceBatch b, double* __restrict__ acc,
                                                                              double* __restrict__ host_out,
                                                                              unsigned long long seq)
{
    const int y = (int)blockIdx.x, s = b.slot[y];
    reduce_rows(b.partials[y], b.nblocks[y], acc + 32 * (size_t)s, host_out + 32 * (size_t)s, seq);
}

// hands a device block of n doubles to the host through its pinned block: data, then the sequence number the host
// spins on (after an RCCL all-reduce on the device block, or for the plane form)
__global__ __launch_bounds__(128) void k_publish(const double* __restrict__ src, int n, double* __restrict__ host_out,
                                                 int flag_slot, unsigned long long seq)
{
    for (int k = threadIdx.x; k < n; k += 128) host_out[k] = src[k];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        reinterpret_cast<volatile unsigned long long*>(host_out)[flag_slot] = seq;
        __threadfence_system();
    }
}

}  // namespace mola_icp_amd
