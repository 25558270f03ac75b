// icp_loop.hpp -- the iteration-control loop of the ICP core (host, fp64).
//
// SURVEY.md §8 rows a1 (driver), a10 (iteration control), a11 (quality), a12
// (results).  Restates what `mp2p_icp::ICP::align()` does around its matcher
// and solver stages, as invoked at src/LidarOdometry.cpp:869-871, over an
// abstract `Stages` interface so the same code drives
//   - the HIP stages (hip_backend.hip), single GPU or one query shard per rank
//     with the accumulator all-reduce in between, and
//   - caller-supplied stages (mola_icp_run_loop), which is how the host logic
//     and the sharded reduction are tested on CPU.
#pragma once
#include <cstdint>
#include <string>

#include "../../include/mola_icp_amd.h"
#include "se3_math.hpp"

namespace mola_icp_amd {

// thread-local last-error string behind mola_icp_last_error()
void set_error(const std::string& msg);
const char* last_error();
int fail(int code, const std::string& msg);  // records msg, returns code

constexpr int kNAccPlaneHost = 92;  // 78 (upper triangle of the 12x12 form) + 12 + 1 + count

struct Stages {
    virtual ~Stages() = default;
    // matcher: transform + NN + gate at pose T; stores the pairing; n_pairs = this rank's count
    virtual int match(const Mat4& T, double threshold, const mola_icp_params& p, uint64_t* n_pairs) = 0;
    // accumulate over the stored pairing -> acc[24] (this rank's partial sums, host memory)
    virtual int accumulate(const mola_icp_params& p, const Mat4& Tcur, int stage, const double cl[3],
                           const double cg[3], bool reset_outliers, double acc[kNAcc]) = 0;
    // row f3 (point-to-plane + Gauss-Newton): plane pairing at pose T, then the quadratic form of its cost
    // (already summed over the ranks).  Default: not available.
    virtual int match_planes(const Mat4& T, const mola_icp_params& p)
    {
        (void)T; (void)p;
        return fail(MOLA_ICP_E_UNSUPPORTED, "these stages do not provide the point-to-plane matcher");
    }
    virtual int accumulate_planes(double acc[kNAccPlaneHost])
    {
        (void)acc;
        return fail(MOLA_ICP_E_UNSUPPORTED, "these stages do not provide the point-to-plane matcher");
    }
    // sum acc across ranks in place (no-op for a single rank)
    virtual int allreduce(double acc[kNAcc]) { (void)acc; return MOLA_ICP_OK; }
    // row a11: this rank's pair count of a PairedRatio pass at pose T (acc[16]; the other entries zero), if the stages can give it
    // WITHOUT a matcher pass -- *done = false (the default): the loop runs match() + accumulate() as always
    virtual int quality_pairs(const Mat4& T, double threshold, const mola_icp_params& p, double acc[kNAcc], bool* done)
    {
        (void)T; (void)threshold; (void)p; (void)acc;
        *done = false;
        return MOLA_ICP_OK;
    }
    virtual uint64_t n_local_total() const = 0;
    virtual uint64_t n_map_total() const = 0;
};

// K independent problems (point-to-point + Horn, or the shipped point-to-plane + Gauss-Newton pipeline) advanced in lockstep, every stage ONE batched launch over the problems still
// iterating: the loop-closure Monte-Carlo (K initial poses on one cloud pair, src/LidarOdometry.cpp:767-788) and the
// nearby-keyframe batch (K pairs, cpp:704-741).  `active[k] != 0` selects the problems a call works on.
struct BatchStages {
    virtual ~BatchStages() = default;
    virtual int size() const = 0;
    virtual int match(const uint8_t* active, const Mat4* T, double threshold, const mola_icp_params& p) = 0;
    virtual int accumulate(const uint8_t* active, const mola_icp_params& p, const Mat4* Tcur, int stage,
                           const double (*cl)[3], const double (*cg)[3], bool reset_outliers, double (*acc)[kNAcc]) = 0;
    // row f3 (point-to-plane + Gauss-Newton), batched: plane pairings of the active problems at their poses, then the
    // quadratic form of each one's cost.  Default: not available.
    virtual int match_planes(const uint8_t* active, const Mat4* T, const mola_icp_params& p)
    {
        (void)active; (void)T; (void)p;
        return fail(MOLA_ICP_E_UNSUPPORTED, "these batched stages do not provide the point-to-plane matcher");
    }
    virtual int accumulate_planes(const uint8_t* active, double (*acc)[kNAccPlaneHost])
    {
        (void)active; (void)acc;
        return fail(MOLA_ICP_E_UNSUPPORTED, "these batched stages do not provide the point-to-plane matcher");
    }
    virtual uint64_t n_local_total(int k) const = 0;
    virtual uint64_t n_map_total(int k) const = 0;
};
// Per problem the same sequence of operations as run_icp_loop (either pipeline): results are bit-identical to
// K separate runs over stages that compute the same sums.  out = K results.
int run_icp_loop_batch(BatchStages& st, const Mat4* init, const mola_icp_params& p, mola_icp_result* out);

// Runs the loop; fills T, quality, n_iterations, termination, n_pairs, rmse, cov,
// ms_iterations, ms_quality of *out (other fields untouched).
int run_icp_loop(Stages& st, const Mat4& init, const mola_icp_params& p, mola_icp_result* out);

// adds the share of a point-to-point pairing (its 24 unit-weight sums, accumulated at pose T) to the 92-term quadratic form of the
// Gauss-Newton cost (mixed pairings in one solve: icp_loop.cpp)
void mixed_form(const double acc[kNAcc], const Mat4& T, double pacc[kNAccPlaneHost]);

int validate_params(const mola_icp_params& p);
// the single-entry parameter set in force at iteration `it` of a staged pipeline (include/mola_icp_amd.h: mola_icp_matcher_entry);
// false: no matcher's range holds the iteration
bool stage_params(const mola_icp_params& p, uint32_t it, mola_icp_params& eff);
// ... and the general form: the number of active matchers (0, 1, 2: eff, eff2), whether a solver's range holds the iteration
int stage_params(const mola_icp_params& p, uint32_t it, mola_icp_params& eff, mola_icp_params* eff2, bool* solver_in_range);

}  // namespace mola_icp_amd
