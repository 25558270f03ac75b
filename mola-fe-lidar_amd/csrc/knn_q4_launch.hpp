// knn_q4_launch.hpp -- host entry points of k_knn_q4 (kernels_knn_q4.hpp), which lives in a translation unit of its own (knn_q4_launch.hip)
#pragma once
#include "kernels_planes.hpp"

namespace mola_icp_amd {

// lpq = lanes per query: 4 (16 queries per wave, four waves per workgroup) or 2 (32 per wave, two waves per workgroup)
bool knn_q4_has(int list_len, int lpq = 4);             // instantiated list lengths (knn + 1)
size_t knn_q4_static_lds(int list_len, int lpq = 4);    // the kernel's static LDS footprint (hipFuncGetAttributes, asked once per instantiation)
int knn_q4_workgroups_per_cu(int lpq = 4);              // its launch bounds
// one problem: grid = workgroups of 64 queries
hipError_t knn_q4_launch(hipStream_t stream, int list_len, const KnnBatch<1>& b, int grid, size_t dyn_lds, float thr2, float thr2x, double threshold,
                         double plane_eig_thr, unsigned long long* staged, int lds_boxes, unsigned long long* cert_stats, int lpq = 4);
// up to kKnnMaxBatch problems: grid = (workgroups of the largest, problems)
hipError_t knn_q4_launch_batch(hipStream_t stream, int list_len, const KnnBatch<kKnnMaxBatch>& b, int grid_x, int n_problems, size_t dyn_lds, float thr2,
                               float thr2x, double threshold, double plane_eig_thr, unsigned long long* staged, int lds_boxes, unsigned long long* cert_stats, int lpq = 4);

}  // namespace mola_icp_amd
