// mola_icp_amd_shim_mrpt.h -- IcpShim bound to the reference's own types.  This file belongs in the mola-fe-lidar
// tree (next to src/LidarOdometry.cpp); it needs MRPT >= 2.1 and mp2p_icp (CMakeLists.txt:17-24 of the reference),
// neither of which exists in this repository's build image, so it is NOT compiled here -- the member bodies it
// forwards to are (shim/mola_icp_amd_shim.h, tests/hosts/shim_test.cpp).
//
//   icp-settings-*.yaml:   icp_class: mola_icp_amd::ICP_MI355X          (instead of mp2p_icp::ICP, icpreg:7)
//   LidarOdometry.cpp:46-53, inside MRPT_INITIALIZER(do_register_LidarOdometry):
//                          mrpt::rtti::registerClass(CLASS_ID(mola_icp_amd::ICP_MI355X));
//   CMakeLists.txt:        target_link_libraries(${PROJECT_NAME} PRIVATE mola_icp_amd)   + this repo's include/ and shim/
// load_icp_set_of_params() (cpp:57-88) and run_one_icp() (cpp:851-895) then compile and run unchanged.
#pragma once
#if !__has_include(<mp2p_icp/ICP.h>)
#error "mola_icp_amd_shim_mrpt.h needs mp2p_icp and MRPT (build it inside the mola-fe-lidar tree)"
#endif
#include <mp2p_icp/ICP.h>
#include <mrpt/containers/yaml.h>
#include <mrpt/maps/CPointsMap.h>
#include <mrpt/math/CMatrixFixed.h>
#include <mrpt/poses/CPose3D.h>
#include <mrpt/rtti/CObject.h>

#include <sstream>

#include "mola_icp_amd_shim.h"

namespace mola_icp_amd {

struct MrptTraits {
    using Yaml = mrpt::containers::yaml;
    using MetricMap = mp2p_icp::metric_map_t;
    using Pose = mrpt::math::TPose3D;
    using Parameters = mp2p_icp::Parameters;
    using Results = mp2p_icp::Results;
    static std::string yaml_text(const Yaml& y)
    {
        std::stringstream ss;
        y.printAsYAML(ss);
        return ss.str();
    }
    static void points(const MetricMap& m, const float*& x, const float*& y, const float*& z, size_t& n)
    {
        // the "raw" layer is the one the front-end fills (cpp:215-224): CPointsMap keeps fp32 SoA buffers
        const auto pts = m.point_layer(mp2p_icp::metric_map_t::PT_LAYER_RAW);
        ASSERT_(pts);
        x = pts->getPointsBufferRef_x().data();
        y = pts->getPointsBufferRef_y().data();
        z = pts->getPointsBufferRef_z().data();
        n = pts->size();
    }
    static void store(Results& out, const mola_icp_result& r)
    {
        mrpt::math::CMatrixDouble44 T;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) T(i, j) = r.T[4 * i + j];
        out.optimal_tf.mean = mrpt::poses::CPose3D(T);                                   // read at cpp:876, 879
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j) out.optimal_tf.cov(i, j) = r.cov[6 * i + j];
        out.quality = r.quality;                                                         // cpp:873, 880
        out.nIterations = r.n_iterations;                                                // cpp:886
        out.terminationReason = static_cast<mp2p_icp::IterTermReason>(r.termination);    // cpp:888 (same enumerators, same order)
    }
};

class ICP_MI355X : public mp2p_icp::ICP {
    DEFINE_MRPT_OBJECT(ICP_MI355X, mola_icp_amd)
   public:
    void initialize_solvers(const mrpt::containers::yaml& y) override { impl_.initialize_solvers(y); }
    void initialize_matchers(const mrpt::containers::yaml& y) override { impl_.initialize_matchers(y); }
    void initialize_quality_evaluators(const mrpt::containers::yaml& y) override { impl_.initialize_quality_evaluators(y); }
    void align(const mp2p_icp::metric_map_t& from, const mp2p_icp::metric_map_t& to,
               const mrpt::math::TPose3D& init_to_wrt_from, const mp2p_icp::Parameters& p,
               mp2p_icp::Results& result) override
    {
        impl_.align(from, to, init_to_wrt_from, p, result);
    }

   private:
    IcpShim<MrptTraits> impl_;
};

}  // namespace mola_icp_amd
// in ONE translation unit of the host:  IMPLEMENTS_MRPT_OBJECT(ICP_MI355X, mp2p_icp::ICP, mola_icp_amd)
