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

__global__ __launch_bounds__(256) void k_curve_keys(const float* __restrict__ gx, const float* __restrict__ gy,
                                                    const float* __restrict__ gz, int M, float x0, float y0, float z0,
                                                    float scale, unsigned int* __restrict__ keys, int* __restrict__ vals)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
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
                    const float bbox[6], DevBuf& scratch, float* sxyz, int* perm)
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
    float ext = 0.f;
    for (int k = 0; k < 3; ++k) ext = fmaxf(ext, bbox[3 + k] - bbox[k]);
    const float scale = ext > 0 ? 1023.999f / ext : 0.f;  // isotropic cells
    const unsigned nb = (unsigned)((M + 255) / 256);
    hipLaunchKernelGGL(k_curve_keys, dim3(nb), dim3(256), 0, stream, gx, gy, gz, Mi, bbox[0], bbox[1], bbox[2], scale, k_in,
                       v_in);
    HIPCHK(hipGetLastError());
    HIPCHK(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, k_in, k_out, v_in, perm, Mi, 0, 30, stream));
    hipLaunchKernelGGL(k_gather_sorted, dim3((unsigned)((M_padded + 255) / 256)), dim3(256), 0, stream, gx, gy, gz, perm,
                       Mi, (int)M_padded, sxyz, sxyz + M_padded, sxyz + 2 * M_padded);
    HIPCHK(hipGetLastError());
    return MOLA_ICP_OK;
}

}  // namespace mola_icp_amd
