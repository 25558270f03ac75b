// oracle/ref_probe/mp2p_icp_time.cpp -- TEST INFRASTRUCTURE ONLY (never part of the product library).
//
// The only thing that could pin this repository's oracle is the reference's real arithmetic: the third-party
// mp2p_icp (+ MRPT) that MOLAorg/mola-fe-lidar calls at src/LidarOdometry.cpp:869-871 and requires at
// CMakeLists.txt:17-24.  It is absent from every machine this build has seen, so THIS FILE HAS NEVER BEEN
// COMPILED; bench.py's `_probe_reference()` tries to build it only when it finds an mp2p_icp installation on the
// box it runs on, and reports what happened (`reference_probe` in the bench line).  API as recalled [EXT] from
// mp2p_icp (ICP::align(pcLocal, pcGlobal, initialGuessLocalWrtGlobal, Parameters, Results); matchers()/solvers()/
// quality_evaluators() containers; metric_map_t::layers; Matcher_Points_DistanceThreshold::threshold).
//
//   mp2p_icp_time <clouds.bin> <iterations> <gate_m>
//   clouds.bin: u64 M, u64 N, then M x (x,y,z) fp32 of the global cloud, N x (x,y,z) fp32 of the local cloud
//   prints one line:  iterations=<n> seconds=<s> T=<16 numbers row-major> quality=<q>
#include <mp2p_icp/ICP.h>
#include <mp2p_icp/Matcher_Points_DistanceThreshold.h>
#include <mp2p_icp/QualityEvaluator_PairedRatio.h>
#include <mp2p_icp/Solver_Horn.h>
#include <mp2p_icp/metricmap.h>
#include <mrpt/maps/CSimplePointsMap.h>
#include <mrpt/poses/CPose3D.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

static mrpt::maps::CSimplePointsMap::Ptr read_cloud(std::FILE* f, uint64_t n)
{
    auto pc = mrpt::maps::CSimplePointsMap::Create();
    std::vector<float> xyz(3 * n);
    if (std::fread(xyz.data(), sizeof(float), xyz.size(), f) != xyz.size()) std::exit(3);
    pc->reserve(n);
    for (uint64_t i = 0; i < n; ++i) pc->insertPointFast(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    pc->mark_as_modified();
    return pc;
}

int main(int argc, char** argv)
{
    if (argc < 4) return 2;
    std::FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    uint64_t M = 0, N = 0;
    if (std::fread(&M, 8, 1, f) != 1 || std::fread(&N, 8, 1, f) != 1) return 3;
    mp2p_icp::metric_map_t global, local;
    global.layers[mp2p_icp::metric_map_t::PT_LAYER_RAW] = read_cloud(f, M);
    local.layers[mp2p_icp::metric_map_t::PT_LAYER_RAW] = read_cloud(f, N);
    std::fclose(f);

    mp2p_icp::ICP icp;
    auto m = mp2p_icp::Matcher_Points_DistanceThreshold::Create();
    m->threshold = std::atof(argv[3]);
    icp.matchers().push_back(m);
    icp.solvers().push_back(mp2p_icp::Solver_Horn::Create());
    icp.quality_evaluators().clear();
    icp.quality_evaluators().emplace_back(mp2p_icp::QualityEvaluator_PairedRatio::Create(), 1.0);

    mp2p_icp::Parameters p;
    p.maxIterations = (uint32_t)std::atoi(argv[2]);
    p.minAbsStep_trans = 0;   // fixed iteration count: the stall test never fires
    p.minAbsStep_rot = 0;
    mp2p_icp::Results r;
    const auto t0 = std::chrono::steady_clock::now();
    icp.align(local, global, mrpt::math::TPose3D(0, 0, 0, 0, 0, 0), p, r);
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const auto H = mrpt::poses::CPose3D(r.optimal_tf.mean).getHomogeneousMatrixVal<mrpt::math::CMatrixDouble44>();
    std::printf("iterations=%u seconds=%.6f T=", (unsigned)r.nIterations, s);
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) std::printf("%.17g ", H(i, j));
    std::printf("quality=%.9f\n", r.quality);
    return 0;
}
