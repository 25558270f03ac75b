// map_sort.hip -- one-time spatial ordering of a cloud for the tiled matcher: 30-bit Hilbert-curve keys
// (isotropic 10-bit cells on the cloud's bounding box; the Hilbert curve is continuous, so every run of
// consecutive points is spatially compact -- Morton order has scene-sized jumps), radix sort (rocPRIM via
// hipCUB: preprocessing, not the hot loop), gather.
// Plays the role the kd-tree build plays in the reference's CPU path (once per new map).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "hip_backend.hpp"

namespace mola_icp_amd {

#define HIPCHK(expr)                                                                                          \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess)                                                                                 \
            return fail(e_ == hipErrorOutOfMemory ? MOLA_ICP_E_OOM : MOLA_ICP_E_HIP,                          \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                                   \
    } while (0)

__device__ __forceinline__ unsigned int spread10(unsigned int v)
{
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// Skilling's axes -> transposed Hilbert index (J. Skilling, "Programming the Hilbert curve", 2004), 3-D, 10 bits
__device__ __forceinline__ unsigned int hilbert30(unsigned int x, unsigned int y, unsigned int z)
{
    unsigned int X[3] = {x, y, z};
    const unsigned int Mtop = 1u << 9;
    for (unsigned int Q = Mtop; Q > 1; Q >>= 1) {
        const unsigned int Pm = Q - 1;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (X[i] & Q) X[0] ^= Pm;
            else { const unsigned int t = (X[0] ^ X[i]) & Pm; X[0] ^= t; X[i] ^= t; }
        }
    }
    X[1] ^= X[0];
    X[2] ^= X[1];
    unsigned int t = 0;
    for (unsigned int Q = Mtop; Q > 1; Q >>= 1)
        if (X[2] & Q) t ^= Q - 1;
    X[0] ^= t; X[1] ^= t; X[2] ^= t;
    return (spread10(X[0]) << 2) | (spread10(X[1]) << 1) | spread10(X[2]);
}

// (the bounding box is read from device memory, where the reduction in front of this kernel left it: no host round trip between them)
__global__ __launch_bounds__(256) void k_curve_keys(const float* __restrict__ gx, const float* __restrict__ gy,
                                                    const float* __restrict__ gz, int M, const float* __restrict__ bbox,
                                                    unsigned int* __restrict__ keys, int* __restrict__ vals)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const float x0 = bbox[0], y0 = bbox[1], z0 = bbox[2];
    float ext = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) ext = fmaxf(ext, bbox[3 + k] - bbox[k]);
    const float scale = ext > 0 ? 1023.999f / ext : 0.f;  // isotropic cells
    const unsigned int ix = (unsigned int)fminf(fmaxf((gx[i] - x0) * scale, 0.f), 1023.f);
    const unsigned int iy = (unsigned int)fminf(fmaxf((gy[i] - y0) * scale, 0.f), 1023.f);
    const unsigned int iz = (unsigned int)fminf(fmaxf((gz[i] - z0) * scale, 0.f), 1023.f);
    keys[i] = hilbert30(ix, iy, iz);
    vals[i] = i;
}

__global__ __launch_bounds__(256) void k_gather_sorted(const float* __restrict__ gx, const float* __restrict__ gy,
                                                       const float* __restrict__ gz, int* __restrict__ perm, int M,
                                                       int M_padded, float* __restrict__ sx, float* __restrict__ sy,
                                                       float* __restrict__ sz)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M_padded) return;
    float x = 1.0e18f, y = 1.0e18f, z = 1.0e18f;  // padding members: d2 ~ 3e36, never a neighbour
    if (i < M) {
        const int j = perm[i];
        x = gx[j]; y = gy[j]; z = gz[j];
    } else {
        perm[i] = 0x7fffffff;  // padding slots (perm has M_padded entries)
    }
    sx[i] = x; sy[i] = y; sz[i] = z;
}

// sorted copies: sxyz = 3 * M_padded floats (SoA), perm[M] = original index of sorted position
int morton_sort_points(hipStream_t stream, const float* gx, const float* gy, const float* gz, size_t M, size_t M_padded,
                    const float* bbox /*device: min xyz, max xyz*/, DevBuf& scratch, float* sxyz, int* perm)
{
    if (M == 0) return MOLA_ICP_OK;
    const int Mi = (int)M;
    size_t tmp_bytes = 0;
    unsigned int* nk = nullptr;
    int* nv = nullptr;
    HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, nk, nk, nv, nv, Mi, 0, 30, stream));
    const size_t a = (sizeof(unsigned int) * M + 255) / 256 * 256;
    int rc = scratch.reserve(3 * a + tmp_bytes + 256);
    if (rc) return rc;
    char* base = scratch.as<char>();
    unsigned int* k_in = reinterpret_cast<unsigned int*>(base);
    unsigned int* k_out = reinterpret_cast<unsigned int*>(base + a);
    int* v_in = reinterpret_cast<int*>(base + 2 * a);
    void* tmp = base + 3 * a;
    const unsigned nb = (unsigned)((M + 255) / 256);
    hipLaunchKernelGGL(k_curve_keys, dim3(nb), dim3(256), 0, stream, gx, gy, gz, Mi, bbox, k_in, v_in);
    HIPCHK(hipGetLastError());
    HIPCHK(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, k_in, k_out, v_in, perm, Mi, 0, 30, stream));
    hipLaunchKernelGGL(k_gather_sorted, dim3((unsigned)((M_padded + 255) / 256)), dim3(256), 0, stream, gx, gy, gz, perm,
                       Mi, (int)M_padded, sxyz, sxyz + M_padded, sxyz + 2 * M_padded);
    HIPCHK(hipGetLastError());
    return MOLA_ICP_OK;
}

// ---- row e: the part of a map a query shard can reach ------------------------------------------------------
// Stable compaction of the points inside an axis-aligned box: sel[k] = original index of the k-th kept point (ascending),
// *n_kept_host = their number.  The caller gathers the coordinates.
__global__ __launch_bounds__(256) void k_flag_in_box(const float* __restrict__ x, const float* __restrict__ y,
                                                     const float* __restrict__ z, int n, float lx, float ly, float lz, float hx,
                                                     float hy, float hz, unsigned char* __restrict__ flag)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float a = x[i], b = y[i], c = z[i];
    flag[i] = (a >= lx && a <= hx && b >= ly && b <= hy && c >= lz && c <= hz) ? 1 : 0;
}

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
    const int ni = (int)n;
    size_t tmp_bytes = 0;
    hipcub::CountingInputIterator<int> iota(0);
    unsigned char* fl = nullptr;
    int* cnt = nullptr;
    HIPCHK(hipcub::DeviceSelect::Flagged(nullptr, tmp_bytes, iota, fl, sel, cnt, ni, stream));
    const size_t a = (n + 255) / 256 * 256;
    int rc = scratch.reserve(a + 256 + tmp_bytes + 256);
    if (rc) return rc;
    char* base = scratch.as<char>();
    fl = reinterpret_cast<unsigned char*>(base);
    cnt = reinterpret_cast<int*>(base + a);
    void* tmp = base + a + 256;
    hipLaunchKernelGGL(k_flag_in_box, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, y, z, ni, lo[0], lo[1], lo[2],
                       hi[0], hi[1], hi[2], fl);
    HIPCHK(hipGetLastError());
    HIPCHK(hipcub::DeviceSelect::Flagged(tmp, tmp_bytes, iota, fl, sel, cnt, ni, stream));
    int h = 0;
    HIPCHK(hipMemcpyAsync(&h, cnt, sizeof(int), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    *n_kept_host = (size_t)h;
    return MOLA_ICP_OK;
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
// fp32, 21 bits per axis; stable radix sort by key; one thread per voxel head sums its run in fp64 (ascending
// original index) -> centroid.  Output order = ascending key.
__global__ __launch_bounds__(256) void k_voxel_keys(const float* __restrict__ x, const float* __restrict__ y,
                                                    const float* __restrict__ z, int n, float ox, float oy, float oz,
                                                    float inv, unsigned long long* __restrict__ keys, int* __restrict__ vals)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long ix = (unsigned long long)fminf(floorf((x[i] - ox) * inv), 2097151.f);
    const unsigned long long iy = (unsigned long long)fminf(floorf((y[i] - oy) * inv), 2097151.f);
    const unsigned long long iz = (unsigned long long)fminf(floorf((z[i] - oz) * inv), 2097151.f);
    keys[i] = (ix << 42) | (iy << 21) | iz;
    vals[i] = i;
}

__global__ __launch_bounds__(256) void k_voxel_heads(const unsigned long long* __restrict__ keys, int n, int* __restrict__ head)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_voxel_centroids(const float* __restrict__ x, const float* __restrict__ y,
                                                         const float* __restrict__ z, const unsigned long long* __restrict__ keys,
                                                         const int* __restrict__ order, const int* __restrict__ head,
                                                         const int* __restrict__ slot, int n, int capacity,
                                                         float* __restrict__ ox, float* __restrict__ oy, float* __restrict__ oz)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !head[i]) return;
    const int s = slot[i];
    if (s >= capacity) return;
    double sx = 0, sy = 0, sz = 0;
    int c = 0;
    for (int j = i; j < n && keys[j] == keys[i]; ++j) {
        const int o = order[j];
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
    size_t sort_tmp = 0, scan_tmp = 0;
    unsigned long long* nk = nullptr;
    int* nv = nullptr;
    HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_tmp, nk, nk, nv, nv, ni, 0, 63, stream));
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_tmp, nv, nv, ni, stream));
    const size_t a8 = (sizeof(unsigned long long) * n + 255) / 256 * 256, a4 = (sizeof(int) * n + 255) / 256 * 256;
    const size_t tmp_bytes = sort_tmp > scan_tmp ? sort_tmp : scan_tmp;
    int rc = scratch.reserve(2 * a8 + 4 * a4 + tmp_bytes + 512);
    if (rc) return rc;
    char* base = scratch.as<char>();
    unsigned long long* k_in = reinterpret_cast<unsigned long long*>(base);
    unsigned long long* k_out = reinterpret_cast<unsigned long long*>(base + a8);
    int* v_in = reinterpret_cast<int*>(base + 2 * a8);
    int* order = reinterpret_cast<int*>(base + 2 * a8 + a4);
    int* head = reinterpret_cast<int*>(base + 2 * a8 + 2 * a4);
    int* slot = reinterpret_cast<int*>(base + 2 * a8 + 3 * a4);
    void* tmp = base + 2 * a8 + 4 * a4;
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_voxel_keys, dim3(nb), dim3(256), 0, stream, x, y, z, ni, bbox[0], bbox[1], bbox[2], inv, k_in, v_in);
    HIPCHK(hipGetLastError());
    HIPCHK(hipcub::DeviceRadixSort::SortPairs(tmp, sort_tmp, k_in, k_out, v_in, order, ni, 0, 63, stream));
    hipLaunchKernelGGL(k_voxel_heads, dim3(nb), dim3(256), 0, stream, k_out, ni, head);
    HIPCHK(hipGetLastError());
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(tmp, scan_tmp, head, slot, ni, stream));
    hipLaunchKernelGGL(k_voxel_centroids, dim3(nb), dim3(256), 0, stream, x, y, z, k_out, order, head, slot, ni, (int)capacity,
                       out_x, out_y, out_z);
    HIPCHK(hipGetLastError());
    int last[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(&last[0], slot + (ni - 1), sizeof(int), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipMemcpyAsync(&last[1], head + (ni - 1), sizeof(int), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    *n_out_host = (size_t)last[0] + (size_t)last[1];
    return MOLA_ICP_OK;
}

}  // namespace mola_icp_amd
