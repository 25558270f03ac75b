// q4_launch.hip -- k_nn_q4's instantiations and launches (a translation unit of its own: the kernel is rebuilt in seconds)
#include "q4_launch.hpp"

#include "kernels_q4.hpp"

#include <algorithm>
#include <cstdio>
#include <vector>

namespace mola_icp_amd {

size_t q4_static_lds()
{
    static size_t bytes = 0;   // (the kernel's own footprint: nothing here is hand-copied from the kernel)
    if (!bytes) {
        hipFuncAttributes fa{};
        bytes = hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_nn_q4<1>)) == hipSuccess && fa.sharedSizeBytes
                    ? fa.sharedSizeBytes
                    : sizeof(float) * (4 * kQ4RingFloats + 64 * 8);
    }
    return bytes;
}

int q4_workgroups_per_cu() { return kQ4WorkgroupsPerCu; }

#ifdef MOLA_Q4_DIAG
// diagnostic build only (-DMOLA_Q4_DIAG): every launch is followed by a synchronisation and a line of per-wave phase medians
static void q4_diag_report(unsigned long long* dbg, int n_waves)
{
    std::vector<unsigned long long> w(16 * (size_t)n_waves);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(w.data(), dbg, w.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    const char* names[9] = {"A+boxes", "init+wavebox", "upper scan", "tile tests", "tiles", "merge+resolve", "stores+row", "(of init: seeds+sends", "wave box)"};
    std::vector<double> ph[9], life;
    for (int i = 0; i < n_waves; ++i) {
        const unsigned long long* r = &w[16 * (size_t)i];
        if (!r[0] || !r[7]) continue;
        for (int k = 0; k < 7; ++k) ph[k].push_back(r[k + 1] >= r[k] ? (double)(r[k + 1] - r[k]) : 0.0);
        ph[7].push_back(r[8] >= r[1] ? (double)(r[8] - r[1]) : 0.0);
        ph[8].push_back(r[9] >= r[8] ? (double)(r[9] - r[8]) : 0.0);
        life.push_back((double)(r[7] - r[0]));
    }
    if (life.empty()) return;
    auto med = [](std::vector<double>& v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
    std::fprintf(stderr, "[q4 diag] %zu waves, lifetime (100 MHz ticks) p50 %.0f p90 %.0f max %.0f |", life.size(), med(life, 0.5), med(life, 0.9), med(life, 1.0));
    for (int k = 0; k < 9; ++k) std::fprintf(stderr, " %s %.0f/%.0f", names[k], med(ph[k], 0.5), med(ph[k], 0.9));
    std::fprintf(stderr, "\n");
}
#endif

hipError_t q4_launch(hipStream_t stream, const NnBatch<1>& b, int grid, size_t dyn_lds, int lds_boxes, int count_pairs)
{
    unsigned long long* dbg = nullptr;
#ifdef MOLA_Q4_DIAG
    static unsigned long long* dbg_buf = nullptr;
    if (!dbg_buf) (void)hipMalloc(reinterpret_cast<void**>(&dbg_buf), 16 * 8192 * sizeof(unsigned long long));
    (void)hipMemsetAsync(dbg_buf, 0, 16 * 8192 * sizeof(unsigned long long), stream);
    dbg = dbg_buf;
#endif
    hipLaunchKernelGGL((k_nn_q4<1>), dim3(grid), dim3(256), dyn_lds, stream, b, lds_boxes, count_pairs, dbg);
    const hipError_t e = hipGetLastError();
#ifdef MOLA_Q4_DIAG
    q4_diag_report(dbg_buf, grid * 4 < 8192 ? grid * 4 : 8192);
#endif
    return e;
}

hipError_t q4_launch_batch(hipStream_t stream, const NnBatch<kCoopMaxBatch>& b, int grid_x, int n_problems, size_t dyn_lds, int lds_boxes, int count_pairs)
{
    hipLaunchKernelGGL((k_nn_q4<kCoopMaxBatch>), dim3(grid_x, n_problems), dim3(256), dyn_lds, stream, b, lds_boxes, count_pairs, (unsigned long long*)nullptr);
    return hipGetLastError();
}

}  // namespace mola_icp_amd
