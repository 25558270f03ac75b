// Microbenchmark: cycles per MFMA for the NN matcher's steady-state loop, by schedule.
// Build: hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define STEPS 4096

// f32 16x16x4, QT=8 query tiles, software-pipelined: step s issues 8 MFMAs into one set while the
// previous set is reduced with v_min3_i32.  SCHED: 0 = compiler's order, 1 = {1 MFMA, 2 VALU} x 8
template <int SCHED, int NMIN>
__global__ __launch_bounds__(256) void k_f32(const float* __restrict__ in, float* out, unsigned long long* cyc)
{
    const int lane = threadIdx.x & 63;
    float b[8];
    f32x4 C[8], Dp[8];
    for (int t = 0; t < 8; ++t) { b[t] = 1.0f + t + lane; C[t] = f32x4{1.f + t, 2.f, 3.f, 4.f}; Dp[t] = C[t]; }
    int acc = 0x7fffffff, hits = 0;
    const float* p = in + lane;
    float a0 = p[0], a1 = p[64];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 2
    for (int s = 0; s < STEPS; ++s) {
        const float a = (s & 1) ? a1 : a0;
        if (s & 1) a1 = p[(s + 2) * 64]; else a0 = p[(s + 2) * 64];
        f32x4 Dn[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) Dn[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[t], C[t], 0, 0, 0);
        int r = 0x7fffffff;
        if (NMIN) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                r = min(min(r, __float_as_int(Dp[t][0])), __float_as_int(Dp[t][1]));
                r = min(min(r, __float_as_int(Dp[t][2])), __float_as_int(Dp[t][3]));
            }
        }
        if (SCHED == 1) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);  // 2 VALU
            }
        }
        if (__any(r <= 0)) { hits++; acc = min(acc, r); }
#pragma unroll
        for (int t = 0; t < 8; ++t) Dp[t] = Dn[t];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sres = 0;
    for (int t = 0; t < 8; ++t) sres += Dp[t][0];
    out[blockIdx.x * 256 + threadIdx.x] = sres + acc + hits;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// f16 32x32x16, QT=4 query tiles of 32, C = 0; 8 v_min3_i32 per MFMA on the previous set
template <int SCHED, int NMIN>
__global__ __launch_bounds__(256) void k_f16(const float* __restrict__ in, float* out, unsigned long long* cyc)
{
    const int lane = threadIdx.x & 63;
    f16x8 b[4];
    for (int t = 0; t < 4; ++t) for (int j = 0; j < 8; ++j) b[t][j] = (_Float16)(1.0f + t + j + lane);
    const f32x16 Z = {};
    f32x16 Dp[4];
    for (int t = 0; t < 4; ++t) Dp[t] = Z + 1.0f;
    int acc = 0x7fffffff, hits = 0;
    const float4* p = reinterpret_cast<const float4*>(in) + lane;
    float4 a0 = p[0], a1 = p[64];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 2
    for (int s = 0; s < STEPS; ++s) {
        const float4 af = (s & 1) ? a1 : a0;
        if (s & 1) a1 = p[(s + 2) * 64]; else a0 = p[(s + 2) * 64];
        f16x8 a;
        __builtin_memcpy(&a, &af, 16);
        f32x16 Dn[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) Dn[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[t], Z, 0, 0, 0);
        int r = 0x7fffffff;
        if (NMIN) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int k = 0; k < 16; k += 2) r = min(min(r, __float_as_int(Dp[t][k])), __float_as_int(Dp[t][k + 1]));
            }
        }
        if (SCHED == 1) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);  // 8 VALU
            }
        }
        if (__any(r <= 0)) { hits++; acc = min(acc, r); }
#pragma unroll
        for (int t = 0; t < 4; ++t) Dp[t] = Dn[t];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sres = 0;
    for (int t = 0; t < 4; ++t) sres += Dp[t][0];
    out[blockIdx.x * 256 + threadIdx.x] = sres + acc + hits;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <class K>
static void run(const char* name, K kern, int blocks, int mfma_per_step, const float* in)
{
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, sizeof(float) * blocks * 256);
    (void)hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, in, out, cyc);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, in, out, cyc);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    (void)hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    const double waves = blocks / 256.0;
    printf("%-44s %.0f wave/SIMD: %6.1f ticks per MFMA per wave -> %5.1f per MFMA per SIMD; wall %.3f ms\n", name, waves,
           avg / ((double)STEPS * mfma_per_step), avg / ((double)STEPS * mfma_per_step) / waves, ms);
    (void)hipFree(out); (void)hipFree(cyc);
}

int main()
{
    float* in;
    const size_t n = (size_t)(STEPS + 8) * 64 * 4;
    (void)hipMalloc(&in, n * sizeof(float));
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = 1.0f + (float)(i % 97) * 0.25f;  // positive data: never "survivors"
    (void)hipMemcpy(in, h.data(), n * sizeof(float), hipMemcpyHostToDevice);
    for (int blocks : {256, 512, 768}) {
        run("f32 16x16x4 x8, no mins", k_f32<0, 0>, blocks, 8, in);
        run("f32 16x16x4 x8 + 16 min3, compiler order", k_f32<0, 1>, blocks, 8, in);
        run("f32 16x16x4 x8 + 16 min3, {1 MFMA,2 VALU}", k_f32<1, 1>, blocks, 8, in);
        run("f16 32x32x16 x4, no mins", k_f16<0, 0>, blocks, 4, in);
        run("f16 32x32x16 x4 + 32 min3, compiler order", k_f16<0, 1>, blocks, 4, in);
        run("f16 32x32x16 x4 + 32 min3, {1 MFMA,8 VALU}", k_f16<1, 1>, blocks, 4, in);
    }
    return 0;
}
