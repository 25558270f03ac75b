// q4_launch.hpp -- host entry points of k_nn_q4 (kernels_q4.hpp), which lives in a translation unit of its own (q4_launch.hip)
#pragma once
#include "kernels_coop.hpp"

namespace mola_icp_amd {

size_t q4_static_lds();   // the kernel's static LDS footprint (hipFuncGetAttributes, asked once)
int q4_workgroups_per_cu();   // what the kernel's registers allow (its launch bounds)
// one problem: grid = workgroups of 64 queries
hipError_t q4_launch(hipStream_t stream, const NnBatch<1>& b, int grid, size_t dyn_lds, int lds_boxes, int count_pairs);
// up to kCoopMaxBatch problems: grid = (workgroups of the largest, problems)
hipError_t q4_launch_batch(hipStream_t stream, const NnBatch<kCoopMaxBatch>& b, int grid_x, int n_problems, size_t dyn_lds, int lds_boxes, int count_pairs);

}  // namespace mola_icp_amd
