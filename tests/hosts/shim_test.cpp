// shim_test.cpp -- compiles shim/mola_icp_amd_shim.h with plain stand-in value types and walks the five members the
// reference uses (src/LidarOdometry.cpp:78-87, 869-871).  Test infrastructure: built and run by
// tests/test_boundary_hosts.py with `g++ -std=c++17 -Wall -Wextra -Werror`.
//   shim_test config                 no GPU needed: settings, unknown classes, call-order errors
//   shim_test align <clouds.bin>     GPU: one align through the shim; prints the result for the Python side to compare
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mola_icp_amd_shim.h"

namespace {

struct TestTraits {
    struct Yaml { std::string text; };
    struct MetricMap { std::vector<float> x, y, z; };
    struct Pose { double x = 0, y = 0, z = 0, yaw = 0, pitch = 0, roll = 0; };
    struct Weights {
        bool use_scale_outlier_detector = false;
        double scale_outlier_threshold = 1.2;
        bool use_robust_kernel = false;
        double robust_kernel_param = 0.0, robust_kernel_scale = 400.0;
    };
    struct Parameters {
        uint32_t maxIterations = 40;
        double minAbsStep_trans = 5e-4, minAbsStep_rot = 1e-4;
        Weights pairingsWeightParameters;
    };
    struct Results {
        double T[16], cov[36], quality = 0;
        uint32_t nIterations = 0, terminationReason = 0;
    };
    static std::string yaml_text(const Yaml& y) { return y.text; }
    static void points(const MetricMap& m, const float*& x, const float*& y, const float*& z, size_t& n)
    {
        x = m.x.data(); y = m.y.data(); z = m.z.data(); n = m.x.size();
    }
    static void store(Results& o, const mola_icp_result& r)
    {
        std::memcpy(o.T, r.T, sizeof o.T);
        std::memcpy(o.cov, r.cov, sizeof o.cov);
        o.quality = r.quality; o.nIterations = r.n_iterations; o.terminationReason = r.termination;
    }
};
using Shim = mola_icp_amd::IcpShim<TestTraits>;

// the sub-trees of params/icp-settings-regular.yaml:23-46, as mrpt::containers::yaml::printAsYAML would hand them over
const char* kSolvers = "- class: mp2p_icp::Solver_GaussNewton\n  params:\n    maxIterations: 20\n";
const char* kMatchers =
    "- class: mp2p_icp::Matcher_Point2Plane\n  params:\n    distanceThreshold: 0.70\n    planeEigenThreshold: 0.07\n"
    "    knn: 6\n    runFromIteration: 0\n    runUpToIteration: 0\n";
const char* kQuality = "- class: mp2p_icp::QualityEvaluator_PairedRatio\n  params:\n    thresholdDistance: 0.10\n";

int fails = 0;
void expect(bool ok, const char* what)
{
    std::printf("%s %s\n", ok ? "ok  " : "FAIL", what);
    if (!ok) ++fails;
}
template <class F>
std::string thrown(F&& f)
{
    try { f(); } catch (const std::exception& e) { return e.what(); }
    return "";
}

int run_config()
{
    Shim icp;
    TestTraits::MetricMap a, b;
    TestTraits::Results res;
    // align before the three initialize_* calls: a clear error, no crash
    expect(thrown([&] { icp.align(a, b, {}, {}, res); }).find("initialize_solvers") != std::string::npos, "align before initialize_*");
    icp.initialize_solvers({kSolvers});                          // cpp:81
    expect(icp.settings().solver_class == MOLA_ICP_SOLVER_GAUSS_NEWTON && icp.settings().solver_max_iterations == 20, "solvers");
    icp.initialize_matchers({kMatchers});                        // cpp:84
    expect(icp.settings().matcher_class == MOLA_ICP_MATCHER_POINT2PLANE && icp.settings().matcher_threshold == 0.70 &&
               icp.settings().plane_eigen_threshold == 0.07 && icp.settings().knn == 6, "matchers");
    icp.initialize_quality_evaluators({kQuality});               // cpp:87
    expect(icp.settings().quality_threshold == 0.10 && icp.settings().solver_class == MOLA_ICP_SOLVER_GAUSS_NEWTON, "quality (earlier stages kept)");
    // unknown classes fail at the call that brings them, naming the class (cpp:66-75's rule, per stage)
    const std::string e1 = thrown([&] { Shim s; s.initialize_solvers({"- class: mp2p_icp::Solver_OLAE\n"}); });
    expect(e1.find("Solver_OLAE") != std::string::npos, "unknown solver class named in the exception");
    const std::string e2 = thrown([&] { Shim s; s.initialize_matchers({"- class: foo::Matcher\n"}); });
    expect(e2.find("foo::Matcher") != std::string::npos, "unknown matcher class named in the exception");
    const std::string e3 = thrown([&] { Shim s; s.initialize_quality_evaluators({"- class: foo::Quality\n"}); });
    expect(e3.find("foo::Quality") != std::string::npos, "unknown quality class named in the exception");
    // no usable device: the failure of mola_icp_create surfaces as an exception at the first align (there is no CPU path)
    int n_dev_err = 0;
    {
        mola_icp_handle* h = nullptr;
        const int rc = mola_icp_create(-1, &h);
        if (rc == MOLA_ICP_OK) mola_icp_destroy(h); else n_dev_err = 1;
    }
    if (n_dev_err) {
        a.x = a.y = a.z = {0.f, 1.f, 2.f}; b = a;
        const std::string e4 = thrown([&] { icp.align(a, b, {}, {}, res); });
        expect(!e4.empty() && e4.find("[mola_icp_amd]") == 0, "no device: exception from align");
        std::printf("     (%s)\n", e4.c_str());
    }
    return fails;
}

int run_align(const char* path)
{
    FILE* f = std::fopen(path, "rb");
    if (!f) { std::perror(path); return 2; }
    uint64_t hdr[2];
    if (std::fread(hdr, sizeof hdr, 1, f) != 1) return 2;
    TestTraits::MetricMap from, to;
    auto rd = [&](std::vector<float>& v, size_t n) { v.resize(n); return std::fread(v.data(), sizeof(float), n, f) == n; };
    if (!rd(from.x, hdr[0]) || !rd(from.y, hdr[0]) || !rd(from.z, hdr[0]) || !rd(to.x, hdr[1]) || !rd(to.y, hdr[1]) || !rd(to.z, hdr[1])) return 2;
    std::fclose(f);
    Shim icp;
    icp.initialize_solvers({kSolvers});
    icp.initialize_matchers({kMatchers});
    icp.initialize_quality_evaluators({kQuality});
    TestTraits::Parameters par;            // icp-settings-regular.yaml:10-21
    par.maxIterations = 100; par.minAbsStep_trans = 5e-5; par.minAbsStep_rot = 1e-5;
    TestTraits::Pose guess;
    guess.x = 0.05; guess.yaw = 0.004;
    TestTraits::Results r;
    icp.align(from, to, guess, par, r);
    std::printf("RESULT");
    for (double v : r.T) std::printf(" %.17g", v);
    std::printf(" %.17g %u %u\n", r.quality, r.nIterations, r.terminationReason);
    return 0;
}

}  // namespace

int main(int argc, char** argv)
{
    if (argc >= 2 && !std::strcmp(argv[1], "config")) return run_config() ? 1 : 0;
    if (argc >= 3 && !std::strcmp(argv[1], "align")) return run_align(argv[2]);
    std::fprintf(stderr, "usage: shim_test config | align <clouds.bin>\n");
    return 2;
}
