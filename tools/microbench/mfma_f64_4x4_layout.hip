// Which lane supplies / receives which element of v_mfma_f64_4x4x4_4b_f64 on gfx950?  One-hot probe: experiment (a, b) sets
// A = 1 in lane a only and B = 1 in lane b only; D's non-zero lane (if any) is where A[a] * B[b] lands.  Prints the map.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/mfma_f64_4x4_layout.hip -o tools/microbench/mfma_f64_4x4_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(int* out)
{
    const int a = blockIdx.x >> 6, b = blockIdx.x & 63, lane = threadIdx.x;
    const double A = lane == a ? 1.0 : 0.0, B = lane == b ? 1.0 : 0.0;
    double D = 0.0;
    D = __builtin_amdgcn_mfma_f64_4x4x4f64(A, B, D, 0, 0, 0);
    if (D != 0.0) out[blockIdx.x] = lane;
}

// cycles per DEPENDENT instruction (one wave alone), and with a second wave of the same SIMD-set doing packed-fp32 VALU work
template <int KIND>
__global__ void rate(unsigned long long* out, int n)
{
    const int wave = threadIdx.x >> 6;
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0;
    typedef double v4d __attribute__((ext_vector_type(4)));
    v4d D16 = {0, 0, 0, 0};
    double D4 = 0.0;
    float f0 = threadIdx.x, f1 = 1.0f, f2 = 0.5f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave == 0 || KIND >= 2) {
        if (wave == 0) {
            for (int i = 0; i < n; ++i) {
                if (KIND & 1) D16 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, D16, 0, 0, 0);
                else D4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, D4, 0, 0, 0);
            }
        } else {
            for (int i = 0; i < 8 * n; ++i) { f0 = fmaf(f0, f1, f2); f1 = fmaf(f1, f2, f0); }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;
    if (D16[0] + D4 + f0 + f1 == 12345.678) out[8] = 1;
}

int main()
{
    {
        unsigned long long* o;
        hipMalloc(&o, 16 * sizeof(unsigned long long));
        unsigned long long h[16];
        const int n = 4096;
        for (int kind = 0; kind < 4; ++kind) {
            for (int rep = 0; rep < 2; ++rep) {
                // 64 threads: the MFMA wave alone; 512 threads (kind >= 2): 8 waves = 2 per SIMD, wave 0 runs the MFMA chain, the others VALU
                const int threads = kind >= 2 ? 512 : 64;
                if (kind == 0) hipLaunchKernelGGL(rate<0>, dim3(1), dim3(threads), 0, 0, o, n);
                if (kind == 1) hipLaunchKernelGGL(rate<1>, dim3(1), dim3(threads), 0, 0, o, n);
                if (kind == 2) hipLaunchKernelGGL(rate<2>, dim3(1), dim3(threads), 0, 0, o, n);
                if (kind == 3) hipLaunchKernelGGL(rate<3>, dim3(1), dim3(threads), 0, 0, o, n);
                hipMemcpy(h, o, sizeof h, hipMemcpyDeviceToHost);
            }
            std::printf("%s f64 MFMA, %s: %.1f memtime ticks per dependent MFMA (wave 0); VALU wave 4 (same SIMD as wave 0): %.2f ticks per 16 fma\n",
                        (kind & 1) ? "16x16x4" : "4x4x4_4b", kind >= 2 ? "with VALU waves beside it" : "alone", (double)h[0] / n,
                        kind >= 2 ? (double)h[4] / n : 0.0);
        }
    }
    int* d;
    hipMalloc(&d, 4096 * sizeof(int));
    hipMemset(d, 0xff, 4096 * sizeof(int));
    hipLaunchKernelGGL(probe, dim3(4096), dim3(64), 0, 0, d);
    std::vector<int> h(4096);
    hipMemcpy(h.data(), d, 4096 * sizeof(int), hipMemcpyDeviceToHost);
    // for every A lane: the B lanes it meets and the D lane of each product
    for (int a = 0; a < 64; ++a) {
        std::printf("A lane %2d:", a);
        for (int b = 0; b < 64; ++b)
            if (h[a * 64 + b] >= 0) std::printf(" (B %2d -> D %2d)", b, h[a * 64 + b]);
        std::printf("\n");
    }
    return 0;
}
