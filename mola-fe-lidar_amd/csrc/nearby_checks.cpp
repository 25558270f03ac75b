// nearby_checks.cpp -- the host policy around the nearby-keyframe / loop-closure ICPs:
//   * which keyframes of the local pose graph get checked against the current one
//     (LidarOdometry::checkForNearbyKFs, src/LidarOdometry.cpp:570-741),
//   * the loop-closure Monte-Carlo's perturbed initial guesses (doCheckForNonAdjacentKFs, cpp:767-783),
//   * the check itself: run the ICP(s), keep the first best goodness, accept-the-edge test (cpp:751-815).
// Pure host logic on plain arrays; the registrations go through mola_icp_align / mola_icp_align_multi_init (the K
// guesses of a loop closure are ONE batched device problem) or an injected align function (tests without a GPU).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <vector>

#include "../../include/mola_icp_amd.h"
#include "icp_loop.hpp"
#include "se3_math.hpp"

using namespace mola_icp_amd;

namespace {

// splitmix64 + Box-Muller: the in-repo seeded generator (the reference draws from a time-seeded
// mrpt::random::CRandomGenerator, cpp:772 -- not reproducible by design; the draw ORDER x, y, z, yaw per sample is kept)
struct SplitMix {
    uint64_t s;
    explicit SplitMix(uint64_t seed) : s(seed) {}
    uint64_t next()
    {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double uniform() { return ((double)(next() >> 11) + 0.5) * (1.0 / 9007199254740992.0); }  // (0, 1)
    double gaussian(double mean, double sigma)
    {
        const double u1 = uniform(), u2 = uniform();
        return mean + sigma * std::sqrt(-2.0 * std::log(u1)) * std::cos(2.0 * M_PI * u2);
    }
};

}  // namespace

extern "C" {

int mola_lo_select_checks(const mola_lo_params* p, const mola_lo_kf_candidate* kfs, size_t n_kfs, uint64_t* nearby_ids,
                          size_t nearby_capacity, size_t* n_nearby, uint64_t* loop_closure_id, int* has_loop_closure)
{
    if (!p || (n_kfs && !kfs) || !n_nearby || !loop_closure_id || !has_loop_closure)
        return fail(MOLA_ICP_E_BADARG, "null argument");
    if (p->max_nearby_align_checks == 0) return fail(MOLA_ICP_E_BADARG, "max_nearby_align_checks must be >= 1");
    *n_nearby = 0;
    *has_loop_closure = 0;
    *loop_closure_id = 0;
    try {
        // KF_distances: keyed by the Euclidean distance, so two keyframes at exactly the same distance collapse into
        // the one inserted last (cpp:551: `KF_distances[kfs.second.norm()] = ...` in node order)
        std::map<double, const mola_lo_kf_candidate*> by_dist;
        for (size_t i = 0; i < n_kfs; ++i) by_dist[kfs[i].eucl_dist] = &kfs[i];
        const auto it1 = by_dist.lower_bound(p->min_dist_to_matching);                                           // cpp:574
        const auto it2 = by_dist.upper_bound(std::max(p->max_dist_to_loop_closure, p->max_dist_to_matching));    // cpp:575-576
        std::vector<uint64_t> nearby;
        bool have_lc = false;
        uint64_t lc = 0;
        for (auto it = it1; it != it2; ++it) {
            const mola_lo_kf_candidate& c = *it->second;
            const bool is_lc = c.topo_dist >= p->min_topo_dist_to_consider_loopclosure;   // cpp:588-589
            if (!is_lc && c.eucl_dist > p->max_dist_to_matching) continue;                // cpp:592-594
            if (c.already_checked) continue;                                              // cpp:600-604
            if (!is_lc) nearby.push_back(c.kf_id);                                        // cpp:679-686, in distance order
            else if (!have_lc) { have_lc = true; lc = c.kf_id; }   // loop_closure_checks.begin(): the smallest distance, cpp:727-729
        }
        // "send a maximum of N" (cpp:704-711): a stride, so up to ceil(n / decim) checks go out
        const size_t nn = nearby.size();
        const size_t decim = std::max<size_t>(1, nn / p->max_nearby_align_checks);
        size_t k = 0;
        for (size_t idx = 0; idx < nn; idx += decim, ++k)
            if (k < nearby_capacity && nearby_ids) nearby_ids[k] = nearby[idx];
        *n_nearby = k;
        *has_loop_closure = have_lc ? 1 : 0;
        *loop_closure_id = lc;
        if (k > nearby_capacity) return fail(MOLA_ICP_E_BADARG, "nearby_ids too small (n_nearby holds the needed size)");
        return MOLA_ICP_OK;
    } catch (const std::exception& e) {
        return fail(MOLA_ICP_E_INTERNAL, e.what());
    }
}

int mola_lo_montecarlo_guesses(const double init_xyzypr[6], double max_dist_to_loop_closure, uint32_t n_samples,
                               uint64_t seed, double* guesses_xyzypr, double* guesses_T)
{
    if (!init_xyzypr || (n_samples && !guesses_xyzypr && !guesses_T)) return fail(MOLA_ICP_E_BADARG, "null argument");
    const double std_xyz = max_dist_to_loop_closure * 0.1;   // cpp:768
    const double std_rot = 2.0 * M_PI / 180.0;               // cpp:769
    SplitMix rnd(seed);
    for (uint32_t i = 0; i < n_samples; ++i) {
        double g[6];
        std::memcpy(g, init_xyzypr, sizeof g);               // d->init_guess_to_wrt_from = original_guess, cpp:776
        g[0] += rnd.gaussian(0, std_xyz);                    // cpp:777-780: x, y, z, yaw in this order
        g[1] += rnd.gaussian(0, std_xyz);
        g[2] += rnd.gaussian(0, std_xyz);
        g[3] += rnd.gaussian(0, std_rot);
        if (guesses_xyzypr) std::memcpy(guesses_xyzypr + 6 * (size_t)i, g, sizeof g);
        if (guesses_T) {
            const Mat4 T = pose_from_xyzypr(g);
            std::memcpy(guesses_T + 16 * (size_t)i, T.m, sizeof T.m);
        }
    }
    return MOLA_ICP_OK;
}

int mola_lo_check_nonadjacent(mola_icp_handle* icp, mola_lo_align_fn align_cb, void* user, const mola_lo_params* lp,
                              int is_loop_closure, const float* from_x, const float* from_y, const float* from_z, size_t M,
                              const float* to_x, const float* to_y, const float* to_z, size_t N,
                              const double init_xyzypr[6], uint64_t seed, mola_lo_check_result* out)
{
    if ((!icp && !align_cb) || !lp || !init_xyzypr || !out) return fail(MOLA_ICP_E_BADARG, "null argument");
    try {
        std::memset(out, 0, sizeof *out);
        out->best_guess = -1;
        // run_one_icp with d->align_kind's ICP object AND its own Parameters (cpp:682-683, 693-694, 869)
        const mola_icp_params& ip = is_loop_closure ? lp->icp_loop_closure : lp->icp_without_vel;
        double last_guess[6];
        std::memcpy(last_guess, init_xyzypr, sizeof last_guess);
        mola_icp_result best;
        std::memset(&best, 0, sizeof best);   // ICP_Output{}: goodness = 0, pose = identity-less default (h:128-132)
        const Mat4 I = Mat4::identity();
        std::memcpy(best.T, I.m, sizeof best.T);
        int rc;
        if (!is_loop_closure) {  // cpp:756-760
            const Mat4 T0 = pose_from_xyzypr(init_xyzypr);
            rc = align_cb ? align_cb(user, from_x, from_y, from_z, M, to_x, to_y, to_z, N, T0.m, &ip, &best)
                          : mola_icp_align(icp, from_x, from_y, from_z, M, to_x, to_y, to_z, N, T0.m, &ip, &best);
            if (rc) return rc < 0 ? rc : fail(MOLA_ICP_E_INTERNAL, "align function failed");
            out->n_attempts = 1;
            out->best_guess = 0;
        } else {  // cpp:762-788
            const uint32_t K = lp->loop_closure_montecarlo_samples;
            std::vector<double> g6(6 * (size_t)K), gT(16 * (size_t)K);
            if ((rc = mola_lo_montecarlo_guesses(init_xyzypr, lp->max_dist_to_loop_closure, K, seed, g6.data(), gT.data()))) return rc;
            if (K) std::memcpy(last_guess, &g6[6 * (size_t)(K - 1)], sizeof last_guess);  // d->init_guess keeps the LAST sample (cpp:776-780)
            out->n_attempts = K;
            if (K && align_cb) {
                for (uint32_t i = 0; i < K; ++i) {
                    mola_icp_result r;
                    std::memset(&r, 0, sizeof r);
                    rc = align_cb(user, from_x, from_y, from_z, M, to_x, to_y, to_z, N, &gT[16 * (size_t)i], &ip, &r);
                    if (rc) return rc < 0 ? rc : fail(MOLA_ICP_E_INTERNAL, "align function failed");
                    if (r.quality > best.quality) { best = r; out->best_guess = (int32_t)i; }   // cpp:785-786
                }
            } else if (K) {  // the K guesses as one batched device problem
                mola_icp_result b;
                int bi = -1;
                if ((rc = mola_icp_align_multi_init(icp, from_x, from_y, from_z, M, to_x, to_y, to_z, N, K, gT.data(), &ip, nullptr, &b, &bi)))
                    return rc;
                if (bi >= 0) { best = b; out->best_guess = bi; }
            }
        }
        out->icp = best;
        // accept the new edge? (cpp:791-815)
        Mat4 rel;
        std::memcpy(rel.m, best.T, sizeof rel.m);
        const Mat4 guess = pose_from_xyzypr(last_guess);
        const Mat4 d = mul(inverse_rigid(guess), rel);   // rel_pose - init_guess (CPose3D inverse composition)
        const double pos_correction = std::sqrt(d(0, 3) * d(0, 3) + d(1, 3) * d(1, 3) + d(2, 3) * d(2, 3));
        const double gnorm = std::sqrt(last_guess[0] * last_guess[0] + last_guess[1] * last_guess[1] + last_guess[2] * last_guess[2]);
        out->correction_percent = pos_correction / (gnorm + 0.01);                              // cpp:797-798
        std::memcpy(out->init_guess_used, last_guess, sizeof last_guess);
        const double thres = is_loop_closure ? lp->min_icp_goodness_lc : lp->min_icp_goodness;  // cpp:810-813
        out->edge_accepted = (best.quality > thres && (out->correction_percent < 0.2 || is_loop_closure)) ? 1 : 0;  // cpp:815-817
        return MOLA_ICP_OK;
    } catch (const std::bad_alloc&) {
        return fail(MOLA_ICP_E_OOM, "host allocation failed");
    } catch (const std::exception& e) {
        return fail(MOLA_ICP_E_INTERNAL, e.what());
    }
}

}  // extern "C"
