// icp_loop.cpp -- see icp_loop.hpp.
#include "icp_loop.hpp"
#include "trace.hpp"

#include <chrono>
#include <cmath>
#include <cstring>
#include <vector>

namespace mola_icp_amd {

namespace {
thread_local std::string g_last_error;
double now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}
}  // namespace

void set_error(const std::string& msg) { g_last_error = msg; }
const char* last_error() { return g_last_error.c_str(); }
int fail(int code, const std::string& msg)
{
    g_last_error = msg;
    return code;
}

namespace {
bool in_range(uint32_t it, uint32_t from, uint32_t upto) { return it >= from && (upto == 0 || it <= upto); }

// the one-matcher / one-solver parameter set an iteration runs with (entry k of `matchers:`, entry j of `solvers:`)
mola_icp_params with_entries(const mola_icp_params& p, int k, int j)
{
    mola_icp_params e = p;
    if (k > 0) {
        const mola_icp_matcher_entry& m = p.extra_matchers[k - 1];
        e.matcher_class = m.matcher_class; e.matcher_threshold = m.matcher_threshold; e.plane_eigen_threshold = m.plane_eigen_threshold;
        e.knn = m.knn; e.run_from_iteration = m.run_from_iteration; e.run_up_to_iteration = m.run_up_to_iteration;
    }
    if (j > 0) {
        const mola_icp_solver_entry& sv = p.extra_solvers[j - 1];
        e.solver_class = sv.solver_class; e.solver_max_iterations = sv.solver_max_iterations;
        e.solver_run_from_iteration = sv.run_from_iteration; e.solver_run_up_to_iteration = sv.run_up_to_iteration;
    }
    e.n_extra_matchers = 0;
    e.n_extra_solvers = 0;
    return e;
}
int validate_single(const mola_icp_params& p);
}  // namespace

// The `matchers:` / `solvers:` entries in force at iteration `it` as single-entry parameter sets (see mola_icp_matcher_entry).
// Returns the number of matchers whose range holds the iteration: 0 = none (no pairings); 1 = `eff`; 2 = `eff` and `eff2`, the
// first two active entries in the order listed -- validate_params admits that only for one Matcher_Points_DistanceThreshold plus
// one Matcher_Point2Plane, whose pairings feed ONE Gauss-Newton solve (mixed_form below).  *solver_in_range = false when no
// solver's range holds the iteration: nothing can solve it, the align ends with SolverError ([EXT] mp2p_icp tries its solvers in
// order and reports SolverError when none succeeds); `eff` then carries the last solver listed.
int stage_params(const mola_icp_params& p, uint32_t it, mola_icp_params& eff, mola_icp_params* eff2, bool* solver_in_range)
{
    int k_act[2] = {-1, -1}, n_act = 0;
    for (int k = 0; k <= (int)p.n_extra_matchers && k <= MOLA_ICP_MAX_EXTRA_STAGES; ++k) {
        const uint32_t from = k ? p.extra_matchers[k - 1].run_from_iteration : p.run_from_iteration;
        const uint32_t upto = k ? p.extra_matchers[k - 1].run_up_to_iteration : p.run_up_to_iteration;
        if (in_range(it, from, upto)) { if (n_act < 2) k_act[n_act] = k; ++n_act; }
    }
    if (solver_in_range) *solver_in_range = true;
    if (n_act == 0) return 0;
    int j_act = -1;
    for (int j = 0; j <= (int)p.n_extra_solvers && j <= MOLA_ICP_MAX_EXTRA_STAGES; ++j) {
        const uint32_t from = j ? p.extra_solvers[j - 1].run_from_iteration : p.solver_run_from_iteration;
        const uint32_t upto = j ? p.extra_solvers[j - 1].run_up_to_iteration : p.solver_run_up_to_iteration;
        if (in_range(it, from, upto)) { j_act = j; break; }
    }
    if (j_act < 0) {
        if (solver_in_range) *solver_in_range = false;
        j_act = (int)p.n_extra_solvers;
    }
    eff = with_entries(p, k_act[0], j_act);
    if (n_act >= 2 && eff2) *eff2 = with_entries(p, k_act[1], j_act);
    return n_act;
}
bool stage_params(const mola_icp_params& p, uint32_t it, mola_icp_params& eff) { return stage_params(p, it, eff, nullptr, nullptr) > 0; }

int validate_params(const mola_icp_params& p)
{
    if (p.n_extra_matchers > MOLA_ICP_MAX_EXTRA_STAGES || p.n_extra_solvers > MOLA_ICP_MAX_EXTRA_STAGES)
        return fail(MOLA_ICP_E_BADARG, "n_extra_matchers / n_extra_solvers exceed MOLA_ICP_MAX_EXTRA_STAGES");
    if (p.n_extra_quality > MOLA_ICP_MAX_EXTRA_STAGES) return fail(MOLA_ICP_E_BADARG, "n_extra_quality exceeds MOLA_ICP_MAX_EXTRA_STAGES");
    for (uint32_t e = 0; e < p.n_extra_quality; ++e) {
        const mola_icp_quality_entry& q = p.extra_quality[e];
        if (q.quality_class != MOLA_ICP_QUALITY_PAIRED_RATIO) return fail(MOLA_ICP_E_BADARG, "unknown quality_class");
        if (!(q.quality_threshold > 0) || !std::isfinite(q.quality_threshold))
            return fail(MOLA_ICP_E_BADARG, "quality thresholdDistance must be a positive finite distance");
        if (!(q.weight >= 0) || !std::isfinite(q.weight)) return fail(MOLA_ICP_E_BADARG, "quality weight must be finite and >= 0");
    }
    if (p.n_extra_quality && !(p.quality_weight >= 0)) return fail(MOLA_ICP_E_BADARG, "quality weight must be finite and >= 0");
    if (p.n_extra_matchers == 0 && p.n_extra_solvers == 0) return validate_single(p);
    // Several entries.  Which entries are active changes only where a range begins or ends: every distinct combination an align
    // can meet is met at iteration 0 or at one of those boundaries (no walk over the iterations, no cut-off).
    std::vector<uint32_t> probes{0u};
    auto add_range = [&](uint32_t from, uint32_t upto) {
        probes.push_back(from);
        if (upto != 0 && upto != 0xffffffffu) probes.push_back(upto + 1);
    };
    add_range(p.run_from_iteration, p.run_up_to_iteration);
    add_range(p.solver_run_from_iteration, p.solver_run_up_to_iteration);
    for (uint32_t k = 0; k < p.n_extra_matchers; ++k) add_range(p.extra_matchers[k].run_from_iteration, p.extra_matchers[k].run_up_to_iteration);
    for (uint32_t j = 0; j < p.n_extra_solvers; ++j) add_range(p.extra_solvers[j].run_from_iteration, p.extra_solvers[j].run_up_to_iteration);
    bool any = false;
    for (uint32_t it : probes) {
        if (it >= p.max_iterations) continue;
        any = true;
        mola_icp_params eff, eff2;
        bool solver_ok = true;
        const int n_act = stage_params(p, it, eff, &eff2, &solver_ok);
        if (n_act == 0) continue;
        int rc = validate_single(eff);
        if (rc) return rc;
        if (n_act == 1) continue;
        if ((rc = validate_single(eff2))) return rc;
        const bool one_of_each = n_act == 2 && eff.matcher_class != eff2.matcher_class;
        if (!one_of_each)
            return fail(MOLA_ICP_E_UNSUPPORTED, "matchers: " + std::to_string(n_act) + " entries are active in iteration " + std::to_string(it) +
                                                    ": this build merges the pairings of ONE mp2p_icp::Matcher_Points_DistanceThreshold and ONE "
                                                    "mp2p_icp::Matcher_Point2Plane in a solve -- give other combinations disjoint runFromIteration / "
                                                    "runUpToIteration ranges");
        if (eff.solver_class != MOLA_ICP_SOLVER_GAUSS_NEWTON)
            return fail(MOLA_ICP_E_UNSUPPORTED, "point-to-point and point-to-plane pairings in one iteration (" + std::to_string(it) +
                                                    ") need mp2p_icp::Solver_GaussNewton (Solver_Horn only consumes point-to-point pairings)");
        // pairingsWeightParameters with two matchers: the scale outlier detector prunes the POINT matcher's pairings (two passes of
        // the centroid-relative test, as in front of Horn) before they enter the Gauss-Newton form -- what the reference's own
        // params block asks for (params/icp-settings-regular.yaml:14-17); plane pairings carry no such test.  The robust kernel's
        // weights are defined on point pairings relative to the centroids of a Horn solve: not available here.
        if (p.use_robust_kernel)
            return fail(MOLA_ICP_E_UNSUPPORTED, "pairingsWeightParameters.use_robust_kernel is not available when two matchers feed one solve "
                                                    "(use_scale_outlier_detector is)");
    }
    if (!any) return validate_single(with_entries(p, 0, 0));
    return MOLA_ICP_OK;
}

namespace {
int validate_single(const mola_icp_params& p)
{
    if (!(p.matcher_threshold > 0) || !std::isfinite(p.matcher_threshold))
        return fail(MOLA_ICP_E_BADARG, "matcher threshold must be a positive finite distance");
    if (!(p.quality_threshold > 0) || !std::isfinite(p.quality_threshold))
        return fail(MOLA_ICP_E_BADARG, "quality thresholdDistance must be a positive finite distance");
    if (p.use_scale_outlier_detector && !(p.scale_outlier_threshold >= 1.0))
        return fail(MOLA_ICP_E_BADARG, "scale_outlier_threshold must be >= 1");
    if (p.matcher_class != MOLA_ICP_MATCHER_POINTS_DISTANCE_THRESHOLD && p.matcher_class != MOLA_ICP_MATCHER_POINT2PLANE)
        return fail(MOLA_ICP_E_BADARG, "unknown matcher_class");
    if (p.solver_class != MOLA_ICP_SOLVER_HORN && p.solver_class != MOLA_ICP_SOLVER_GAUSS_NEWTON)
        return fail(MOLA_ICP_E_BADARG, "unknown solver_class");
    if (p.matcher_class == MOLA_ICP_MATCHER_POINT2PLANE) {
        // the reference's shipped pipeline (params/icp-settings-regular.yaml:23-39)
        if (p.solver_class != MOLA_ICP_SOLVER_GAUSS_NEWTON)
            return fail(MOLA_ICP_E_UNSUPPORTED,
                        "mp2p_icp::Matcher_Point2Plane pairings need mp2p_icp::Solver_GaussNewton (Solver_Horn only "
                        "consumes point-to-point pairings)");
        // pairingsWeightParameters.use_robust_kernel with plane pairings only: refused by name, or -- under the reading that the flag acts on
        // point pairings and finds none, as use_scale_outlier_detector does in the reference's own params block -- accepted without effect
        // (mola_icp_params.reading_robust_kernel_skips_planes; the plane pipeline never consults the weight parameters)
        if (p.use_robust_kernel && !p.reading_robust_kernel_skips_planes)
            return fail(MOLA_ICP_E_UNSUPPORTED, "use_robust_kernel is not available with mp2p_icp::Matcher_Point2Plane (readings: robust_kernel_skips_planes "
                                                "accepts it with unit weights on the plane pairings)");
        if (p.knn < 3 || p.knn > 16) return fail(MOLA_ICP_E_UNSUPPORTED, "Matcher_Point2Plane: knn must be in [3, 16]");
        if (!(p.plane_eigen_threshold > 0)) return fail(MOLA_ICP_E_BADARG, "planeEigenThreshold must be > 0");
        if (p.solver_max_iterations == 0) return fail(MOLA_ICP_E_BADARG, "Solver_GaussNewton: maxIterations must be > 0");
    }
    // point-to-point pairings + Solver_GaussNewton: the Gauss-Newton minimiser of sum |T l - g|^2 is Horn's
    // closed-form solution, which is what runs.
    if (p.quality_class != MOLA_ICP_QUALITY_PAIRED_RATIO) return fail(MOLA_ICP_E_BADARG, "unknown quality_class");
    return MOLA_ICP_OK;
}
}  // namespace

// One solver invocation on the stored pairing (row a8 + a9).
//  unweighted: acc0 -> Horn.
//  weighted  : [mp2p_icp optimal_tf_horn behaviour, EXT-recalled] centroids from
//              a unit-weight pass, then the centroid-relative tests/weights; with
//              the scale outlier detector the whole thing runs twice, the second
//              time with the centroids recomputed without the first pass's outliers.
// pairs_global = acc0[16] of the first unit-weight pass (all gated pairings).
static int solve_on_pairing(Stages& st, const mola_icp_params& p, const Mat4& Tcur, Mat4& Tnew, double acc[kNAcc],
                            double* pairs_global, bool* solver_error)
{
    *solver_error = false;
    int rc = st.accumulate(p, Tcur, 0, nullptr, nullptr, true, acc);
    if (rc) return rc;
    if ((rc = st.allreduce(acc))) return rc;
    *pairs_global = acc[16];
    if (!(acc[16] > 0)) return MOLA_ICP_OK;  // caller decides: NoPairings
    const bool weighted = p.use_scale_outlier_detector || p.use_robust_kernel;
    if (!weighted) {
        if (!solve_horn(acc, nullptr, nullptr, Tnew)) *solver_error = true;
        return MOLA_ICP_OK;
    }
    const int passes = (p.use_scale_outlier_detector && !p.reading_outlier_single_pass) ? 2 : 1;   // (the reading: mola_icp_params)
    for (int pass = 0; pass < passes; ++pass) {
        if (pass > 0) {
            if ((rc = st.accumulate(p, Tcur, 0, nullptr, nullptr, false, acc))) return rc;
            if ((rc = st.allreduce(acc))) return rc;
        }
        if (!(acc[0] > 0)) { *solver_error = true; return MOLA_ICP_OK; }
        const double cl[3] = {acc[1] / acc[0], acc[2] / acc[0], acc[3] / acc[0]};
        const double cg[3] = {acc[4] / acc[0], acc[5] / acc[0], acc[6] / acc[0]};
        if ((rc = st.accumulate(p, Tcur, 1, cl, cg, false, acc))) return rc;
        if ((rc = st.allreduce(acc))) return rc;
        if (!solve_horn(acc, cl, cg, Tnew)) { *solver_error = true; return MOLA_ICP_OK; }
    }
    return MOLA_ICP_OK;
}

// The point matcher's share of a MIXED solve under pairingsWeightParameters.use_scale_outlier_detector
// (params/icp-settings-regular.yaml:14-17): the pairings that would reach Horn's final solve -- solve_on_pairing's sequence without
// its solves: unit-weight sums -> centroids -> the centroid-relative test flags outliers; once more with the centroids of what is
// left.  acc = the 24 sums over the survivors of the second test (unit weights: validate_params refuses the robust kernel here);
// all zero when nothing is left -- the plane pairings may still carry the solve.
static int point_sums_without_outliers(Stages& st, const mola_icp_params& p, const Mat4& Tcur, double acc[kNAcc], double* gated)
{
    int rc = st.accumulate(p, Tcur, 0, nullptr, nullptr, true, acc);
    if (rc) return rc;
    if ((rc = st.allreduce(acc))) return rc;
    *gated = acc[16];   // every pairing the matcher gated: what the iteration REPORTS (as solve_on_pairing does); the survivors feed the solve
    for (int pass = 0; pass < (p.reading_outlier_single_pass ? 1 : 2); ++pass) {
        if (pass > 0) {
            if ((rc = st.accumulate(p, Tcur, 0, nullptr, nullptr, false, acc))) return rc;
            if ((rc = st.allreduce(acc))) return rc;
        }
        if (!(acc[0] > 0)) {
            for (int k = 0; k < kNAcc; ++k) acc[k] = 0.0;
            return MOLA_ICP_OK;
        }
        const double cl[3] = {acc[1] / acc[0], acc[2] / acc[0], acc[3] / acc[0]};
        const double cg[3] = {acc[4] / acc[0], acc[5] / acc[0], acc[6] / acc[0]};
        if ((rc = st.accumulate(p, Tcur, 1, cl, cg, false, acc))) return rc;
        if ((rc = st.allreduce(acc))) return rc;
    }
    return MOLA_ICP_OK;
}

// Mixed pairings in one solve (the reference's `matchers:` is "a sequence of one or more", params/icp-settings-regular.yaml:28-39,
// all initialised together at src/LidarOdometry.cpp:83-84; [EXT] mp2p_icp hands the pairings of every active matcher to the solver).
// The Gauss-Newton cost of the iteration is  sum_planes (n.(R l + t - c))^2 + sum_points |R l + t - g|^2 -- and a point-to-point
// pairing is three plane terms with the normals e_x, e_y, e_z and d = g_x, g_y, g_z: its share of the quadratic form
// x^T A x - 2 b^T x + c0 (x = [R row-major, t]) follows from the 24 sums the point-to-point accumulation already delivers --
//   A[3k+i][3k+j] += sum l_i l_j,  A[3k+i][9+k] += sum l_i,  A[9+k][9+k] += W,  b[3k+i] += sum l_i g_k,  b[9+k] += sum g_k  (k = x, y, z)
// -- no kernel of its own.  sum |g|^2 is not among the sums; c0 is set so that the form's value at the matcher's pose equals the
// accumulated sum of squared distances (acc[17]; it only enters the reported cost / rmse, never the step).  Unit weights; under the scale
// outlier detector the sums hold the point pairings that survive it (point_sums_without_outliers).
void mixed_form(const double acc[kNAcc], const Mat4& T, double pacc[kNAccPlaneHost])
{
    auto at = [](int a, int b) { if (a > b) { const int t = a; a = b; b = t; } return a * 12 - a * (a - 1) / 2 + (b - a); };   // upper triangle, row-major
    const double W = acc[0];
    const double* sl = acc + 1;
    const double* sg = acc + 4;
    const double* slg = acc + 7;                         // sum l_i g_k at [3 i + k]
    const double sll[3][3] = {{acc[18], acc[19], acc[20]}, {acc[19], acc[21], acc[22]}, {acc[20], acc[22], acc[23]}};
    double A[12][12] = {}, b[12] = {};
    for (int k = 0; k < 3; ++k) {
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) A[3 * k + i][3 * k + j] = sll[i][j];
            A[3 * k + i][9 + k] = A[9 + k][3 * k + i] = sl[i];
            b[3 * k + i] = slg[3 * i + k];
        }
        A[9 + k][9 + k] = W;
        b[9 + k] = sg[k];
    }
    double x[12];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) x[3 * r + c] = T(r, c);
        x[9 + r] = T(r, 3);
    }
    double q = 0;   // x^T A x - 2 b^T x at the matcher's pose
    for (int i = 0; i < 12; ++i) {
        double v = 0;
        for (int j = 0; j < 12; ++j) v += A[i][j] * x[j];
        q += x[i] * (v - 2 * b[i]);
    }
    for (int i = 0; i < 12; ++i)
        for (int j = i; j < 12; ++j) pacc[at(i, j)] += A[i][j];
    for (int i = 0; i < 12; ++i) pacc[78 + i] += b[i];
    pacc[90] += acc[17] - q;
    pacc[91] += acc[16];
}

int run_icp_loop(Stages& st, const Mat4& init, const mola_icp_params& p_all, mola_icp_result* out)
{
    int rc = validate_params(p_all);
    if (rc) return rc;
    mola_icp_params p = p_all;          // the single-entry set of the current iteration (stage_params)
    p.n_extra_matchers = p.n_extra_solvers = 0;
    const bool staged = p_all.n_extra_matchers != 0 || p_all.n_extra_solvers != 0;
    bool last_planes = p_all.matcher_class == MOLA_ICP_MATCHER_POINT2PLANE;   // which pipeline fed the last solve
    Mat4 T = init, Tprev = init;
    uint32_t term = MOLA_ICP_TERM_UNDEFINED;
    uint32_t it = 0;
    double acc[kNAcc] = {};
    double last_acc[kNAcc] = {};
    bool have_solution = false;
    double plane_pairs = 0, plane_rmse = 0;
    double last_pacc[kNAccPlaneHost] = {};
    const double t0 = now_ms();
    {
    TraceRange tr_loop("mola_icp.iterations");
    for (; it < p.max_iterations; ++it) {
        bool run_matcher;
        int n_active = 1;
        bool solver_in_range = true;
        mola_icp_params p2 = p;   // (two matchers active: the second one's single-entry set)
        if (staged) {
            n_active = stage_params(p_all, it, p, &p2, &solver_in_range);
            run_matcher = n_active > 0;
        } else {
            run_matcher = it >= p.run_from_iteration && (p.run_up_to_iteration == 0 || it <= p.run_up_to_iteration);
        }
        if (!run_matcher || st.n_local_total() == 0 || st.n_map_total() == 0) {
            term = MOLA_ICP_TERM_NO_PAIRINGS;
            break;
        }
        if (!solver_in_range) { term = MOLA_ICP_TERM_SOLVER_ERROR; break; }   // no `solvers:` entry covers this iteration
        last_planes = p.matcher_class == MOLA_ICP_MATCHER_POINT2PLANE || n_active == 2;
        Mat4 Tn = T;
        double pairs_global = 0;
        bool solver_error = false;
        if (n_active == 2) {
            // one Matcher_Points_DistanceThreshold + one Matcher_Point2Plane (validate_params): both pair ALL local points at this
            // pose ([EXT] no "already matched" exclusion between matchers), one Gauss-Newton solve over the sum of their costs
            const mola_icp_params& pp = p.matcher_class == MOLA_ICP_MATCHER_POINT2PLANE ? p : p2;
            const mola_icp_params& pq = p.matcher_class == MOLA_ICP_MATCHER_POINT2PLANE ? p2 : p;
            double pacc[kNAccPlaneHost];
            if ((rc = st.match_planes(T, pp))) return rc;
            if ((rc = st.accumulate_planes(pacc))) return rc;
            if ((rc = st.match(T, pq.matcher_threshold, pq, nullptr))) return rc;
            // The iteration's pairings = everything the two matchers gated (what it reports and what decides NoPairings: the same
            // rule as solve_on_pairing's); under pairingsWeightParameters.use_scale_outlier_detector the point pairings the detector
            // flags leave the SOLVE only -- A and b hold the survivors' terms, the rmse is taken over them (plane pairings carry no
            // such test: the detector is defined on point pairs).
            double gated_points = 0;
            if (pq.use_scale_outlier_detector) {
                if ((rc = point_sums_without_outliers(st, pq, T, acc, &gated_points))) return rc;
            } else {
                if ((rc = st.accumulate(pq, T, 0, nullptr, nullptr, true, acc))) return rc;
                if ((rc = st.allreduce(acc))) return rc;
                gated_points = acc[16];
            }
            pairs_global = pacc[91] + gated_points;
            mixed_form(acc, T, pacc);
            const double in_solve = pacc[91];
            if (!(pairs_global > 0)) { term = MOLA_ICP_TERM_NO_PAIRINGS; break; }
            double cost = 0;
            if (!(in_solve > 0) || !solve_gauss_newton_planes(pacc, T, pp.solver_max_iterations, Tn, &cost)) { term = MOLA_ICP_TERM_SOLVER_ERROR; break; }
            plane_pairs = pairs_global;
            plane_rmse = std::sqrt((cost > 0 ? cost : 0.0) / in_solve);
            std::memcpy(last_pacc, pacc, sizeof last_pacc);
        } else if (p.matcher_class == MOLA_ICP_MATCHER_POINT2PLANE) {
            // row f3: plane pairings -> ONE accumulation pass (the quadratic form of the cost) -> host Gauss-Newton
            double pacc[kNAccPlaneHost];
            if ((rc = st.match_planes(T, p))) return rc;
            if ((rc = st.accumulate_planes(pacc))) return rc;
            pairs_global = pacc[91];
            if (!(pairs_global > 0)) { term = MOLA_ICP_TERM_NO_PAIRINGS; break; }
            double cost = 0;
            if (!solve_gauss_newton_planes(pacc, T, p.solver_max_iterations, Tn, &cost)) solver_error = true;
            if (solver_error) { term = MOLA_ICP_TERM_SOLVER_ERROR; break; }
            plane_pairs = pairs_global;
            plane_rmse = std::sqrt((cost > 0 ? cost : 0.0) / pairs_global);
            std::memcpy(last_pacc, pacc, sizeof last_pacc);
        } else {
            if ((rc = st.match(T, p.matcher_threshold, p, nullptr))) return rc;  // count comes from acc[16]
            if ((rc = solve_on_pairing(st, p, T, Tn, acc, &pairs_global, &solver_error))) return rc;
            if (!(pairs_global > 0)) { term = MOLA_ICP_TERM_NO_PAIRINGS; break; }
            if (solver_error) { term = MOLA_ICP_TERM_SOLVER_ERROR; break; }
            std::memcpy(last_acc, acc, sizeof acc);
        }
        have_solution = true;
        T = Tn;
        double d_xyz, d_rot;
        stall_deltas(T, Tprev, d_xyz, d_rot);
        if (!p.fixed_iterations && std::fabs(d_xyz) < p.min_abs_step_trans && std::fabs(d_rot) < p.min_abs_step_rot) {
            term = MOLA_ICP_TERM_STALLED;
            ++it;  // this iteration did run
            break;
        }
        Tprev = T;
    }
    }
    if (term == MOLA_ICP_TERM_UNDEFINED) term = MOLA_ICP_TERM_MAX_ITERATIONS;
    const double t1 = now_ms();

    // quality (row a11): PairedRatio at the final pose = pairings / min(N, M)
    double quality = 0;
    if (p.skip_quality) quality = -1.0;
    else if (st.n_local_total() > 0 && st.n_map_total() > 0) {
        TraceRange tr_q("mola_icp.quality");
        const double denom = (double)((p_all.reading_quality_denominator_local || st.n_local_total() < st.n_map_total()) ? st.n_local_total() : st.n_map_total());
        // one PairedRatio pass per `quality:` entry, combined by weight (a single entry: that entry's ratio, whatever its weight)
        double wsum = 0, qsum = 0;
        for (uint32_t e = 0; e <= p_all.n_extra_quality && e <= MOLA_ICP_MAX_EXTRA_STAGES; ++e) {
            const double thr = e ? p_all.extra_quality[e - 1].quality_threshold : p_all.quality_threshold;
            double w = e ? p_all.extra_quality[e - 1].weight : p_all.quality_weight;
            if (e == 0 && w == 0) w = 1.0;
            double qacc[kNAcc];
            bool counted = false;
            if ((rc = st.quality_pairs(T, thr, p, qacc, &counted))) return rc;
            if (!counted) {
                if ((rc = st.match(T, thr, p, nullptr))) return rc;
                if ((rc = st.accumulate(p, T, 0, nullptr, nullptr, true, qacc))) return rc;
            }
            if ((rc = st.allreduce(qacc))) return rc;
            if (p_all.n_extra_quality == 0) { qsum = qacc[16] / denom; wsum = 1.0; break; }
            qsum += w * (qacc[16] / denom);
            wsum += w;
        }
        quality = wsum > 0 ? qsum / wsum : 0.0;
    }
    const double t2 = now_ms();

    std::memcpy(out->T, T.m, sizeof out->T);
    out->quality = quality;
    out->n_iterations = it;
    out->termination = term;
    if (last_planes) {
        out->n_pairs = have_solution ? (uint64_t)plane_pairs : 0;
        out->rmse = have_solution ? plane_rmse : 0.0;   // rms point-to-plane distance at the last linearisation
        if (!have_solution || !pose_covariance_planes(last_pacc, T, out->cov)) std::memset(out->cov, 0, sizeof out->cov);
    } else {
        out->n_pairs = have_solution ? (uint64_t)last_acc[16] : 0;
        out->rmse = (have_solution && last_acc[16] > 0) ? std::sqrt(last_acc[17] / last_acc[16]) : 0.0;
        if (!have_solution || !pose_covariance(last_acc, T, out->cov)) std::memset(out->cov, 0, sizeof out->cov);
    }
    out->ms_iterations = t1 - t0;
    out->ms_quality = t2 - t1;
    return MOLA_ICP_OK;
}

// ---- K problems in lockstep (see icp_loop.hpp).  Mirrors run_icp_loop + solve_on_pairing step by step; a problem that
// terminates simply drops out of the active set of the following launches.
int run_icp_loop_batch(BatchStages& st, const Mat4* init, const mola_icp_params& p, mola_icp_result* out)
{
    int rc = validate_params(p);
    if (rc) return rc;
    if (p.n_extra_matchers != 0 || p.n_extra_solvers != 0 || p.n_extra_quality != 0)
        return fail(MOLA_ICP_E_UNSUPPORTED, "the batched loop runs single-entry `matchers:` / `solvers:` / `quality:` pipelines (staged pipelines: align the pairs one by one)");
    const bool planes = p.matcher_class == MOLA_ICP_MATCHER_POINT2PLANE;
    const int K = st.size();
    if (K <= 0) return MOLA_ICP_OK;
    struct State {
        Mat4 T, Tprev;
        uint32_t term = MOLA_ICP_TERM_UNDEFINED, its = 0;
        double last_acc[kNAcc] = {};
        double last_pacc[kNAccPlaneHost] = {};   // row f3: the last linearisation's quadratic form
        double plane_pairs = 0, plane_rmse = 0;
        bool have_solution = false, done = false;
    };
    std::vector<State> s((size_t)K);
    std::vector<Mat4> Tn((size_t)K), Tcur((size_t)K);
    std::vector<uint8_t> active((size_t)K), sub((size_t)K);
    std::vector<double> accv((size_t)K * kNAcc), clv((size_t)K * 3), cgv((size_t)K * 3), paccv;
    auto acc = reinterpret_cast<double(*)[kNAcc]>(accv.data());
    auto cl = reinterpret_cast<double(*)[3]>(clv.data());
    auto cg = reinterpret_cast<double(*)[3]>(cgv.data());
    for (int k = 0; k < K; ++k) { s[k].T = init[k]; s[k].Tprev = init[k]; }
    auto any = [&](const std::vector<uint8_t>& m) { for (uint8_t v : m) if (v) return true; return false; };
    const bool weighted = p.use_scale_outlier_detector || p.use_robust_kernel;
    const double t0 = now_ms();
    for (uint32_t it = 0; it < p.max_iterations; ++it) {
        const bool run_matcher = it >= p.run_from_iteration && (p.run_up_to_iteration == 0 || it <= p.run_up_to_iteration);
        for (int k = 0; k < K; ++k) {
            active[k] = s[k].done ? 0 : 1;
            if (active[k] && (!run_matcher || st.n_local_total(k) == 0 || st.n_map_total(k) == 0)) {
                s[k].term = MOLA_ICP_TERM_NO_PAIRINGS; s[k].its = it; s[k].done = true; active[k] = 0;
            }
            Tcur[k] = s[k].T;
            Tn[k] = s[k].T;
        }
        if (!any(active)) break;
        if (planes) {
            // row f3, batched: plane pairings -> ONE accumulation pass per problem (the quadratic form of its cost) -> K host
            // Gauss-Newton solves -- per problem exactly run_icp_loop's point-to-plane branch
            if ((rc = st.match_planes(active.data(), Tcur.data(), p))) return rc;
            if (paccv.empty()) paccv.resize((size_t)K * kNAccPlaneHost);
            auto pacc = reinterpret_cast<double(*)[kNAccPlaneHost]>(paccv.data());
            if ((rc = st.accumulate_planes(active.data(), pacc))) return rc;
            for (int k = 0; k < K; ++k) {
                if (!active[k]) continue;
                const double pairs_global = pacc[k][91];
                if (!(pairs_global > 0)) { s[k].term = MOLA_ICP_TERM_NO_PAIRINGS; s[k].its = it; s[k].done = true; continue; }
                double cost = 0;
                if (!solve_gauss_newton_planes(pacc[k], s[k].T, p.solver_max_iterations, Tn[k], &cost)) {
                    s[k].term = MOLA_ICP_TERM_SOLVER_ERROR; s[k].its = it; s[k].done = true;
                    continue;
                }
                s[k].plane_pairs = pairs_global;
                s[k].plane_rmse = std::sqrt((cost > 0 ? cost : 0.0) / pairs_global);
                std::memcpy(s[k].last_pacc, pacc[k], sizeof s[k].last_pacc);
                s[k].have_solution = true;
                s[k].T = Tn[k];
                double d_xyz, d_rot;
                stall_deltas(s[k].T, s[k].Tprev, d_xyz, d_rot);
                if (!p.fixed_iterations && std::fabs(d_xyz) < p.min_abs_step_trans && std::fabs(d_rot) < p.min_abs_step_rot) {
                    s[k].term = MOLA_ICP_TERM_STALLED; s[k].its = it + 1; s[k].done = true;
                    continue;
                }
                s[k].Tprev = s[k].T;
            }
            continue;
        }
        if ((rc = st.match(active.data(), Tcur.data(), p.matcher_threshold, p))) return rc;
        // solve_on_pairing, batched
        if ((rc = st.accumulate(active.data(), p, Tcur.data(), 0, nullptr, nullptr, true, acc))) return rc;
        sub = active;  // problems still inside this iteration's solve
        std::vector<uint8_t> solver_error((size_t)K, 0);
        for (int k = 0; k < K; ++k) {
            if (!sub[k]) continue;
            if (!(acc[k][16] > 0)) {  // NoPairings
                s[k].term = MOLA_ICP_TERM_NO_PAIRINGS; s[k].its = it; s[k].done = true; sub[k] = 0; active[k] = 0;
            } else if (!weighted) {
                if (!solve_horn(acc[k], nullptr, nullptr, Tn[k])) solver_error[k] = 1;
                sub[k] = 0;
            }
        }
        if (weighted) {
            const int passes = (p.use_scale_outlier_detector && !p.reading_outlier_single_pass) ? 2 : 1;
            for (int pass = 0; pass < passes && any(sub); ++pass) {
                if (pass > 0)
                    if ((rc = st.accumulate(sub.data(), p, Tcur.data(), 0, nullptr, nullptr, false, acc))) return rc;
                for (int k = 0; k < K; ++k) {
                    if (!sub[k]) continue;
                    if (!(acc[k][0] > 0)) { solver_error[k] = 1; sub[k] = 0; continue; }
                    for (int a = 0; a < 3; ++a) { cl[k][a] = acc[k][1 + a] / acc[k][0]; cg[k][a] = acc[k][4 + a] / acc[k][0]; }
                }
                if (!any(sub)) break;
                if ((rc = st.accumulate(sub.data(), p, Tcur.data(), 1, cl, cg, false, acc))) return rc;
                for (int k = 0; k < K; ++k) {
                    if (!sub[k]) continue;
                    if (!solve_horn(acc[k], cl[k], cg[k], Tn[k])) { solver_error[k] = 1; sub[k] = 0; }
                }
            }
        }
        for (int k = 0; k < K; ++k) {
            if (!active[k]) continue;
            if (solver_error[k]) { s[k].term = MOLA_ICP_TERM_SOLVER_ERROR; s[k].its = it; s[k].done = true; continue; }
            std::memcpy(s[k].last_acc, acc[k], sizeof s[k].last_acc);
            s[k].have_solution = true;
            s[k].T = Tn[k];
            double d_xyz, d_rot;
            stall_deltas(s[k].T, s[k].Tprev, d_xyz, d_rot);
            if (!p.fixed_iterations && std::fabs(d_xyz) < p.min_abs_step_trans && std::fabs(d_rot) < p.min_abs_step_rot) {
                s[k].term = MOLA_ICP_TERM_STALLED; s[k].its = it + 1; s[k].done = true;
                continue;
            }
            s[k].Tprev = s[k].T;
        }
    }
    for (int k = 0; k < K; ++k)
        if (!s[k].done) { s[k].term = MOLA_ICP_TERM_MAX_ITERATIONS; s[k].its = p.max_iterations; }
    const double t1 = now_ms();
    // quality (row a11) of every problem at its final pose
    std::vector<double> quality((size_t)K, 0.0);
    if (p.skip_quality) {
        for (int k = 0; k < K; ++k) quality[k] = -1.0;
    } else {
        for (int k = 0; k < K; ++k) { active[k] = (st.n_local_total(k) > 0 && st.n_map_total(k) > 0) ? 1 : 0; Tcur[k] = s[k].T; }
        if (any(active)) {
            if ((rc = st.match(active.data(), Tcur.data(), p.quality_threshold, p))) return rc;
            if ((rc = st.accumulate(active.data(), p, Tcur.data(), 0, nullptr, nullptr, true, acc))) return rc;
            for (int k = 0; k < K; ++k)
                if (active[k]) {
                    const uint64_t nl = st.n_local_total(k), nm = st.n_map_total(k);
                    quality[k] = acc[k][16] / (double)((p.reading_quality_denominator_local || nl < nm) ? nl : nm);
                }
        }
    }
    const double t2 = now_ms();
    for (int k = 0; k < K; ++k) {
        mola_icp_result& o = out[k];
        std::memcpy(o.T, s[k].T.m, sizeof o.T);
        o.quality = quality[k];
        o.n_iterations = s[k].its;
        o.termination = s[k].term;
        if (planes) {
            o.n_pairs = s[k].have_solution ? (uint64_t)s[k].plane_pairs : 0;
            o.rmse = s[k].have_solution ? s[k].plane_rmse : 0.0;   // rms point-to-plane distance at the last linearisation
            if (!s[k].have_solution || !pose_covariance_planes(s[k].last_pacc, s[k].T, o.cov)) std::memset(o.cov, 0, sizeof o.cov);
        } else {
            o.n_pairs = s[k].have_solution ? (uint64_t)s[k].last_acc[16] : 0;
            o.rmse = (s[k].have_solution && s[k].last_acc[16] > 0) ? std::sqrt(s[k].last_acc[17] / s[k].last_acc[16]) : 0.0;
            if (!s[k].have_solution || !pose_covariance(s[k].last_acc, s[k].T, o.cov)) std::memset(o.cov, 0, sizeof o.cov);
        }
        o.ms_iterations = t1 - t0;   // the whole batch's loop / quality pass (the problems ran together)
        o.ms_quality = t2 - t1;
    }
    return MOLA_ICP_OK;
}

}  // namespace mola_icp_amd
