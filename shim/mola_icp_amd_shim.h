// mola_icp_amd_shim.h -- the reference-side binding of libmola_icp_amd.so.
//
// mola-fe-lidar reaches its ICP through a C++ object created by NAME (`mrpt::rtti::classFactory(icp_class)` ->
// `ptr_cast<mp2p_icp::ICP>`, src/LidarOdometry.cpp:66-68) and five members:
//   Parameters::load_from(yaml)            cpp:78      (mp2p_icp::Parameters -- stays the reference's own type)
//   initialize_solvers(yaml)               cpp:81
//   initialize_matchers(yaml)              cpp:84
//   initialize_quality_evaluators(yaml)    cpp:87
//   align(from, to, init_to_wrt_from, parameters, results)   cpp:869-871
// This header implements exactly those members over the C-ABI of include/mola_icp_amd.h.  All logic lives in
// IcpShim<Traits>; Traits names the host's types (YAML node, point-cloud container, pose, Parameters, Results), so
// the same code is compiled in two places:
//   * in the reference tree with MRPT / mp2p_icp types: shim/mola_icp_amd_shim_mrpt.h (derives from mp2p_icp::ICP,
//     DEFINE_MRPT_OBJECT, registered in MRPT_INITIALIZER next to cpp:46-53);
//   * in this repository's tests with plain stand-in value types (tests/hosts/shim_test.cpp), which is how the
//     member bodies are parsed by a compiler and exercised here, where MRPT and mp2p_icp do not exist.
// Error convention: the C-ABI's status codes come back as std::runtime_error carrying mola_icp_last_error(), i.e.
// the reference's own convention (THROW_EXCEPTION / ASSERT_, cpp:70-75, 860-861), caught per task at cpp:510-513.
#pragma once
#include <mola_icp_amd.h>

#include <cstring>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <string>

namespace mola_icp_amd {

// What Traits must provide (all static):
//   using Yaml / MetricMap / Pose / Parameters / Results
//   std::string yaml_text(const Yaml&)                         block-style YAML of the node (MRPT: printAsYAML)
//   void points(const MetricMap&, const float*& x, const float*& y, const float*& z, size_t& n)   SoA fp32 view
//   void store(Results&, const mola_icp_result&)               optimal_tf (mean + cov), quality, nIterations, terminationReason
// Pose has public x, y, z, yaw, pitch, roll (mrpt::math::TPose3D); Parameters has the eight fields of
// mp2p_icp::Parameters the YAML pins (icp-settings-regular.yaml:11-21).
template <class Traits>
class IcpShim {
   public:
    using Yaml = typename Traits::Yaml;
    using MetricMap = typename Traits::MetricMap;
    using Pose = typename Traits::Pose;
    using Parameters = typename Traits::Parameters;
    using Results = typename Traits::Results;

    explicit IcpShim(int device = -1) : device_(device) { check(mola_icp_params_default(&settings_)); }
    ~IcpShim()
    {
        if (h_) (void)mola_icp_destroy(h_);
    }
    IcpShim(const IcpShim&) = delete;
    IcpShim& operator=(const IcpShim&) = delete;

    // cpp:81, 84, 87.  Each call re-reads the object's settings through the core's own loader, so an unknown class
    // fails HERE, naming the class, as the reference's factories do.
    void initialize_solvers(const Yaml& y) { solvers_ = Traits::yaml_text(y); have_solvers_ = true; reparse(); }
    void initialize_matchers(const Yaml& y) { matchers_ = Traits::yaml_text(y); have_matchers_ = true; reparse(); }
    void initialize_quality_evaluators(const Yaml& y) { quality_ = Traits::yaml_text(y); have_quality_ = true; reparse(); }

    // cpp:869-871: `icp->align(*in.from_pc, *in.to_pc, current_solution, in.icp_params, icp_result)`
    // `par` is per call and may differ from the ICP_case's own (cpp:287-290); the object contributes its
    // matcher / solver / quality settings.  Re-entrant: the handle leases a workspace per call (cpp:94-96, 869).
    void align(const MetricMap& from, const MetricMap& to, const Pose& init_to_wrt_from, const Parameters& par,
               Results& out)
    {
        if (!have_solvers_ || !have_matchers_ || !have_quality_)
            throw std::runtime_error("mola_icp_amd::ICP_MI355X::align(): initialize_solvers/matchers/quality_evaluators first");
        const float *gx, *gy, *gz, *lx, *ly, *lz;
        size_t M = 0, N = 0;
        Traits::points(from, gx, gy, gz, M);
        Traits::points(to, lx, ly, lz, N);
        mola_icp_params call = settings_;  // the eight mp2p_icp::Parameters fields, all of them
        call.max_iterations = (uint32_t)par.maxIterations;
        call.min_abs_step_trans = par.minAbsStep_trans;
        call.min_abs_step_rot = par.minAbsStep_rot;
        call.use_scale_outlier_detector = par.pairingsWeightParameters.use_scale_outlier_detector ? 1 : 0;
        call.scale_outlier_threshold = par.pairingsWeightParameters.scale_outlier_threshold;
        call.use_robust_kernel = par.pairingsWeightParameters.use_robust_kernel ? 1 : 0;
        call.robust_kernel_param = par.pairingsWeightParameters.robust_kernel_param;  // radians on both sides
        call.robust_kernel_scale = par.pairingsWeightParameters.robust_kernel_scale;
        mola_icp_params p;
        check(mola_icp_params_compose(&settings_, &call, &p));
        const double g6[6] = {init_to_wrt_from.x, init_to_wrt_from.y, init_to_wrt_from.z,
                              init_to_wrt_from.yaw, init_to_wrt_from.pitch, init_to_wrt_from.roll};
        double T0[16];
        check(mola_icp_pose_from_xyzypr(g6, T0));
        mola_icp_result r;
        std::memset(&r, 0, sizeof r);
        check(mola_icp_align(handle(), gx, gy, gz, M, lx, ly, lz, N, T0, &p, &r));
        Traits::store(out, r);
    }

    const mola_icp_params& settings() const { return settings_; }

   private:
    static void check(int status)
    {
        if (status != MOLA_ICP_OK) {
            const char* m = mola_icp_last_error();
            throw std::runtime_error(std::string("[mola_icp_amd] ") + ((m && *m) ? m : mola_icp_status_string(status)));
        }
    }
    mola_icp_handle* handle()  // created at the first align: loading a config needs no GPU
    {
        std::lock_guard<std::mutex> lk(mtx_);
        if (!h_) check(mola_icp_create(device_, &h_));
        return h_;
    }
    static std::string indented(const std::string& text)
    {
        std::istringstream in(text);
        std::string line, out;
        while (std::getline(in, line)) out += "  " + line + "\n";
        return out;
    }
    // the three sub-trees (+ the class name and an empty `params`, which stays with mp2p_icp::Parameters) as ONE
    // icp-settings document for mola_icp_params_from_yaml; a stage not initialised yet is stood in for by its
    // default class so that the others can be validated in the order the reference calls them
    void reparse()
    {
        std::string doc = "icp_class: mola_icp_amd::ICP_MI355X\nparams:\n  maxIterations: 40\n";
        doc += "solvers:\n" + (have_solvers_ ? indented(solvers_) : std::string("  - class: mp2p_icp::Solver_Horn\n"));
        doc += "matchers:\n" + (have_matchers_ ? indented(matchers_) : std::string("  - class: mp2p_icp::Matcher_Points_DistanceThreshold\n"));
        doc += "quality:\n" + (have_quality_ ? indented(quality_) : std::string("  - class: mp2p_icp::QualityEvaluator_PairedRatio\n"));
        mola_icp_params p;
        check(mola_icp_params_from_yaml(doc.c_str(), &p));
        settings_ = p;
    }

    int device_;
    mola_icp_handle* h_ = nullptr;
    std::mutex mtx_;
    mola_icp_params settings_{};
    std::string solvers_, matchers_, quality_;
    bool have_solvers_ = false, have_matchers_ = false, have_quality_ = false;
};

}  // namespace mola_icp_amd
