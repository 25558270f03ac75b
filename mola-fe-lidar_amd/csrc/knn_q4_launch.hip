// knn_q4_launch.hip -- k_knn_q4's instantiations and launches (a translation unit of its own: the kernel is rebuilt without the big one)
#define MOLA_ICP_PLANE_TYPES_ONLY   // (kernels_planes.hpp: types + plane_epilogue; its kernels belong to hip_backend.hip)
#include "knn_q4_launch.hpp"

#include "kernels_knn_q4.hpp"

#include <algorithm>
#include <cstdio>
#include <vector>

namespace mola_icp_amd {

// list lengths 4 .. 10 = knn 3 .. 9 (the reference's settings use 6: params/icp-settings-regular.yaml:37); longer lists keep k_knn_coop
#define MOLA_KQ4_LENGTHS(X) X(4) X(5) X(6) X(7) X(8) X(9) X(10)
// ... at two lanes per query 4 .. 9 (a wave's 32 lists must fit its ring)
#define MOLA_KQ4_LENGTHS2(X) X(4) X(5) X(6) X(7) X(8) X(9)

bool knn_q4_has(int list_len, int lpq) { return list_len >= 4 && list_len <= (lpq == 2 ? 9 : 10) && (lpq == 1 || lpq == 2 || lpq == 4); }
int knn_q4_workgroups_per_cu(int lpq) { return lpq == 1 ? kq4_wg_per_cu<1>() : (lpq == 2 ? kq4_wg_per_cu<2>() : kq4_wg_per_cu<4>()); }

size_t knn_q4_static_lds(int list_len, int lpq)
{
    static size_t bytes[3][18] = {};
    if (!knn_q4_has(list_len, lpq)) return 0;
    size_t& b = bytes[lpq == 1 ? 2 : (lpq == 2 ? 1 : 0)][list_len];
    if (!b) {
        hipFuncAttributes fa{};
        hipError_t e = hipErrorInvalidValue;
        if (lpq == 1) {
            switch (list_len) {
#define X(KK) case KK: e = hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_knn_q4<KK, 1, 1>)); break;
                MOLA_KQ4_LENGTHS(X)
#undef X
            }
        } else if (lpq == 2) {
            switch (list_len) {
#define X(KK) case KK: e = hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_knn_q4<KK, 1, 2>)); break;
                MOLA_KQ4_LENGTHS2(X)
#undef X
            }
        } else {
            switch (list_len) {
#define X(KK) case KK: e = hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_knn_q4<KK, 1, 4>)); break;
                MOLA_KQ4_LENGTHS(X)
#undef X
            }
        }
        b = e == hipSuccess && fa.sharedSizeBytes ? fa.sharedSizeBytes : sizeof(float) * (size_t)lpq * kKq4RingFloats + 16;
        (void)hipGetLastError();
    }
    return b;
}

#ifdef MOLA_KQ4_DIAG
// diagnostic build only (-DMOLA_KQ4_DIAG): every single-problem launch is followed by a synchronisation and a few lines of per-wave phases
static unsigned long long* kq4_dbg_buf()
{
    static unsigned long long* buf = nullptr;
    if (!buf) {
        (void)hipMalloc(reinterpret_cast<void**>(&buf), 16 * 8192 * sizeof(unsigned long long));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_kq4_dbg), &buf, sizeof buf);
    }
    return buf;
}
static void kq4_diag_report(int n_waves, int use_seed, int cert_on)
{
    std::vector<unsigned long long> w(16 * (size_t)n_waves);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(w.data(), kq4_dbg_buf(), w.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    const char* names[5] = {"loads+boxes", "seeds+cert", "sweep", "merge+records", "epilogue"};
    std::vector<double> ph[3][5], life[3], tiles;   // [all | sweeping | last wave of its workgroup]
    unsigned long long t_min = ~0ull, t_max = 0ull;
    size_t skipped = 0, solved = 0, lasts = 0;
    unsigned long long c_tiles = 0, c_slow = 0, c_key = 0, c_dup = 0, c_ins = 0, c_tests = 0;
    for (int i = 0; i < n_waves; ++i) {
        const unsigned long long* r = &w[16 * (size_t)i];
        if (!r[0] || !r[4]) continue;
        const bool skip = (r[6] >> 32) & 1ull, last = r[5] != 0;
        skipped += skip; lasts += last; solved += last && r[7];
        if (!skip) tiles.push_back((double)(r[6] & 0xffffffffull));
        c_tiles += r[6] & 0xffffffffull; c_slow += r[8]; c_key += r[9]; c_dup += r[10]; c_ins += r[11]; c_tests += r[12];
        const unsigned long long end = last ? r[5] : r[4];
        t_min = std::min(t_min, r[0]); t_max = std::max(t_max, end);
        for (int cls = 0; cls < 3; ++cls) {
            if ((cls == 1 && skip) || (cls == 2 && !last)) continue;
            ph[cls][0].push_back((double)(r[1] - r[0])); ph[cls][1].push_back((double)(r[2] - r[1]));
            ph[cls][2].push_back(skip ? 0.0 : (double)(r[3] - r[2])); ph[cls][3].push_back((double)(r[4] - (skip ? r[2] : r[3])));
            ph[cls][4].push_back(last ? (double)(r[5] - r[4]) : 0.0);
            life[cls].push_back((double)(end - r[0]));
        }
    }
    if (life[0].empty()) return;
    auto med = [](std::vector<double>& v, double q) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
    std::fprintf(stderr, "[kq4 diag] seed=%d cert=%d: %zu waves (%zu skipped the sweep; %zu last waves, %zu with a plane solve), launch span %.0f ticks (100 MHz), tiles per sweeping wave p50 %.0f p90 %.0f max %.0f\n",
                 use_seed, cert_on, life[0].size(), skipped, lasts, solved, (double)(t_max - t_min), med(tiles, 0.5), med(tiles, 0.9), med(tiles, 1.0));
    std::fprintf(stderr, "[kq4 diag]   %llu tiles evaluated, %llu took the insertion path; lane events in it: %llu keys below the lane's last (%llu list members met again, %llu insertions); %llu groups of four box tests\n", c_tiles, c_slow, c_key, c_dup, c_ins, c_tests);
    const char* cls_names[3] = {"all waves", "sweeping", "last of wg"};
    for (int cls = 0; cls < 3; ++cls) {
        std::fprintf(stderr, "[kq4 diag]   %-10s lifetime p50 %.0f p90 %.0f max %.0f |", cls_names[cls], med(life[cls], 0.5), med(life[cls], 0.9), med(life[cls], 1.0));
        for (int k = 0; k < 5; ++k) std::fprintf(stderr, " %s %.0f/%.0f", names[k], med(ph[cls][k], 0.5), med(ph[cls][k], 0.9));
        std::fprintf(stderr, "\n");
    }
}
#endif

hipError_t knn_q4_launch(hipStream_t stream, int list_len, const KnnBatch<1>& b, int grid, size_t dyn_lds, float thr2, float thr2x, double threshold,
                         double plane_eig_thr, unsigned long long* staged, int lds_boxes, unsigned long long* cert_stats, int lpq)
{
#ifdef MOLA_KQ4_DIAG
    (void)hipMemsetAsync(kq4_dbg_buf(), 0, 16 * 8192 * sizeof(unsigned long long), stream);
#endif
    if (!knn_q4_has(list_len, lpq)) return hipErrorInvalidValue;
    if (lpq == 1) {
        switch (list_len) {
#define X(KK) case KK: hipLaunchKernelGGL((k_knn_q4<KK, 1, 1>), dim3(grid), dim3(64), dyn_lds, stream, b, thr2, thr2x, threshold, plane_eig_thr, staged, lds_boxes, cert_stats); break;
            MOLA_KQ4_LENGTHS(X)
#undef X
        }
    } else if (lpq == 2) {
        switch (list_len) {
#define X(KK) case KK: hipLaunchKernelGGL((k_knn_q4<KK, 1, 2>), dim3(grid), dim3(128), dyn_lds, stream, b, thr2, thr2x, threshold, plane_eig_thr, staged, lds_boxes, cert_stats); break;
            MOLA_KQ4_LENGTHS2(X)
#undef X
        }
    } else {
        switch (list_len) {
#define X(KK) case KK: hipLaunchKernelGGL((k_knn_q4<KK, 1, 4>), dim3(grid), dim3(256), dyn_lds, stream, b, thr2, thr2x, threshold, plane_eig_thr, staged, lds_boxes, cert_stats); break;
            MOLA_KQ4_LENGTHS(X)
#undef X
        }
    }
    const hipError_t e = hipGetLastError();
#ifdef MOLA_KQ4_DIAG
    kq4_diag_report(grid * lpq < 8192 ? grid * lpq : 8192, b.p[0].use_seed, b.p[0].cert_on);
#endif
    return e;
}

hipError_t knn_q4_launch_batch(hipStream_t stream, int list_len, const KnnBatch<kKnnMaxBatch>& b, int grid_x, int n_problems, size_t dyn_lds, float thr2,
                               float thr2x, double threshold, double plane_eig_thr, unsigned long long* staged, int lds_boxes, unsigned long long* cert_stats, int lpq)
{
    if (!knn_q4_has(list_len, lpq)) return hipErrorInvalidValue;
    if (lpq == 1) {
        switch (list_len) {
#define X(KK) case KK: hipLaunchKernelGGL((k_knn_q4<KK, kKnnMaxBatch, 1>), dim3(grid_x, n_problems), dim3(64), dyn_lds, stream, b, thr2, thr2x, threshold, plane_eig_thr, staged, lds_boxes, cert_stats); break;
            MOLA_KQ4_LENGTHS(X)
#undef X
        }
    } else if (lpq == 2) {
        switch (list_len) {
#define X(KK) case KK: hipLaunchKernelGGL((k_knn_q4<KK, kKnnMaxBatch, 2>), dim3(grid_x, n_problems), dim3(128), dyn_lds, stream, b, thr2, thr2x, threshold, plane_eig_thr, staged, lds_boxes, cert_stats); break;
            MOLA_KQ4_LENGTHS2(X)
#undef X
        }
    } else {
        switch (list_len) {
#define X(KK) case KK: hipLaunchKernelGGL((k_knn_q4<KK, kKnnMaxBatch, 4>), dim3(grid_x, n_problems), dim3(256), dyn_lds, stream, b, thr2, thr2x, threshold, plane_eig_thr, staged, lds_boxes, cert_stats); break;
            MOLA_KQ4_LENGTHS(X)
#undef X
        }
    }
    return hipGetLastError();
}

}  // namespace mola_icp_amd
