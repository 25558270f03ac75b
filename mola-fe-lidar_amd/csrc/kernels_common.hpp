// kernels_common.hpp -- pose transform and the distance contract shared by every matcher kernel
// Device code of the ICP core for gfx950; included by hip_backend.hip only (one translation unit: the kernels are
// launched from there).  Numeric contract and data layout: hip_backend.hip / DESIGN.md.
#pragma once
#include <hip/hip_runtime.h>

#include "hip_backend.hpp"

namespace mola_icp_amd {


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

// the key of a point in a cloud's frame: box = min xyz, max xyz of the cloud the keys belong to (isotropic 10-bit cells; a point
// outside the box lands in the nearest cell)
__device__ __forceinline__ unsigned int hilbert_key_in_box(const float (&box)[6], float x, float y, float z)
{
    float ext = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) ext = fmaxf(ext, box[3 + k] - box[k]);
    const float scale = ext > 0 ? 1023.999f / ext : 0.f;  // isotropic cells
    const unsigned int ix = (unsigned int)fminf(fmaxf((x - box[0]) * scale, 0.f), 1023.f);
    const unsigned int iy = (unsigned int)fminf(fmaxf((y - box[1]) * scale, 0.f), 1023.f);
    const unsigned int iz = (unsigned int)fminf(fmaxf((z - box[2]) * scale, 0.f), 1023.f);
    return hilbert30(ix, iy, iz);
}

constexpr float kPadCoord = 1.0e18f;  // padding map points: d2 ~ 3e36, finite, never the minimum

// Sum of NV per-thread doubles over a 256-thread block, in a fixed order: eight values at a time are laid out in LDS
// ([value][thread]), 32 threads per value add 8 entries each in index order, then a 32-wide shuffle tree.  ~3 LDS
// operations per value and thread instead of the 12 of a 64-wide shuffle tree per value (ds_bpermute pairs for fp64),
// which dominated the accumulation kernels.  out[v] is written by one thread per value; all threads must call.
template <int NV>
__device__ __forceinline__ void block_sum_256(const double (&v)[NV], double* __restrict__ out)
{
    __shared__ double s_bs[8][256];
    const int tid = threadIdx.x, j = tid >> 5, l = tid & 31;
#pragma unroll
    for (int c = 0; c < NV; c += 8) {
#pragma unroll
        for (int q = 0; q < 8; ++q) s_bs[q][tid] = c + q < NV ? v[c + q < NV ? c + q : 0] : 0.0;
        __syncthreads();
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < 8; ++r) t += s_bs[j][l + 32 * r];
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) t += __shfl_down(t, off, 32);
        if (l == 0 && c + j < NV) out[c + j] = t;
        __syncthreads();
    }
}

}  // namespace mola_icp_amd
