// config.cpp -- `icp-settings-*.yaml` -> mola_icp_params.
//
// Replaces load_icp_set_of_params() (src/LidarOdometry.cpp:57-88): the same
// five required entries (`icp_class` cpp:63, `params` cpp:77, `solvers` cpp:80,
// `matchers` cpp:83, `quality` cpp:86), the same "class must be known" rule
// (cpp:66-75), and every key of params/icp-settings-regular.yaml:7-46.
#include <cmath>
#include <string>

#include "icp_loop.hpp"
#include "yaml_lite.hpp"

namespace mola_icp_amd {

void params_default(mola_icp_params& p)
{
    p = mola_icp_params{};
    // [EXT] mp2p_icp::Parameters defaults
    p.max_iterations = 40;
    p.min_abs_step_trans = 5e-4;
    p.min_abs_step_rot = 1e-4;
    p.use_scale_outlier_detector = 0;
    p.scale_outlier_threshold = 1.20;
    p.use_robust_kernel = 0;
    p.robust_kernel_param = 0.05 * M_PI / 180.0;
    p.robust_kernel_scale = 400.0;
    p.solver_class = MOLA_ICP_SOLVER_HORN;
    p.solver_max_iterations = 0;
    p.matcher_class = MOLA_ICP_MATCHER_POINTS_DISTANCE_THRESHOLD;
    p.matcher_threshold = 0.50;
    p.plane_eigen_threshold = 0.07;
    p.knn = 6;
    p.run_from_iteration = 0;
    p.run_up_to_iteration = 0;
    p.quality_class = MOLA_ICP_QUALITY_PAIRED_RATIO;
    p.quality_threshold = 0.10;
    p.quality_weight = 1.0;
    p.fixed_iterations = 0;
    p.nn_kernel = MOLA_ICP_NN_AUTO;
    p.skip_quality = 0;
}

// mp2p_icp::Parameters (cpp:77-78) vs the ICP object's own pipelines (cpp:80-87): see mola_icp_params_compose()
void params_compose(const mola_icp_params& object_settings, const mola_icp_params& call_parameters, mola_icp_params& out)
{
    mola_icp_params r = object_settings;
    r.max_iterations = call_parameters.max_iterations;
    r.min_abs_step_trans = call_parameters.min_abs_step_trans;
    r.min_abs_step_rot = call_parameters.min_abs_step_rot;
    r.use_scale_outlier_detector = call_parameters.use_scale_outlier_detector;
    r.scale_outlier_threshold = call_parameters.scale_outlier_threshold;
    r.use_robust_kernel = call_parameters.use_robust_kernel;
    r.robust_kernel_param = call_parameters.robust_kernel_param;
    r.robust_kernel_scale = call_parameters.robust_kernel_scale;
    r.fixed_iterations = call_parameters.fixed_iterations;
    out = r;
}

static const char* kKnownIcpClasses[] = {"mp2p_icp::ICP", "mola_icp_amd::ICP_MI355X"};

static void require(const YamlNode& cfg, const char* key)
{
    if (!cfg.is_map() || !cfg.has(key)) throw std::runtime_error(std::string("Missing YAML required entry `") + key + "`");
}

// entry i of a `solvers:` / `matchers:` / `quality:` sequence ("a sequence of one or more" {class, params}: icpreg:28-31)
static const YamlNode& entry_of_seq(const YamlNode& n, const char* what, size_t i, size_t max_entries)
{
    if (!n.is_seq() || n.seq.empty())
        throw std::runtime_error(std::string("`") + what + "` must be a non-empty sequence of {class, params}");
    if (n.seq.size() > max_entries)
        throw std::runtime_error(std::string("`") + what + "`: at most " + std::to_string(max_entries) + (max_entries == 1 ? " entry" : " entries") +
                                 " in this build (" + std::to_string(n.seq.size()) + " given)");
    const YamlNode& e = n.seq[i];
    if (!e.is_map() || !e.has("class"))
        throw std::runtime_error(std::string("`") + what + "[" + std::to_string(i) + "]` lacks a `class`");
    return e;
}

static void parse_solver(const YamlNode& s, int32_t& cls_out, uint32_t& max_its, uint32_t& from, uint32_t& upto)
{
    const std::string cls = s.at("class").as_string();
    if (cls == "mp2p_icp::Solver_Horn") cls_out = MOLA_ICP_SOLVER_HORN;
    else if (cls == "mp2p_icp::Solver_GaussNewton") cls_out = MOLA_ICP_SOLVER_GAUSS_NEWTON;
    else throw std::runtime_error("solver class=`" + cls + "` is a non-registered class. Known: "
                                  "`mp2p_icp::Solver_Horn`, `mp2p_icp::Solver_GaussNewton`.");
    if (auto* sp = s.find("params")) {
        if (auto* n = sp->find("maxIterations")) max_its = (uint32_t)n->as_int();
        if (auto* n = sp->find("runFromIteration")) from = (uint32_t)n->as_int();
        if (auto* n = sp->find("runUpToIteration")) upto = (uint32_t)n->as_int();
    }
}

static void parse_matcher(const YamlNode& m, int32_t& cls_out, double& thr, double& eig, uint32_t& knn, uint32_t& from, uint32_t& upto)
{
    const std::string cls = m.at("class").as_string();
    const YamlNode* mp = m.find("params");
    if (cls == "mp2p_icp::Matcher_Points_DistanceThreshold") {
        cls_out = MOLA_ICP_MATCHER_POINTS_DISTANCE_THRESHOLD;
        if (mp)
            if (auto* n = mp->find("threshold")) thr = n->as_double();
    } else if (cls == "mp2p_icp::Matcher_Point2Plane") {
        cls_out = MOLA_ICP_MATCHER_POINT2PLANE;
        if (mp) {
            if (auto* n = mp->find("distanceThreshold")) thr = n->as_double();
            if (auto* n = mp->find("planeEigenThreshold")) eig = n->as_double();
            if (auto* n = mp->find("knn")) knn = (uint32_t)n->as_int();
        }
    } else
        throw std::runtime_error("matcher class=`" + cls + "` is a non-registered class. Known: "
                                 "`mp2p_icp::Matcher_Points_DistanceThreshold`, `mp2p_icp::Matcher_Point2Plane`.");
    if (mp) {
        if (auto* n = mp->find("runFromIteration")) from = (uint32_t)n->as_int();
        if (auto* n = mp->find("runUpToIteration")) upto = (uint32_t)n->as_int();
    }
}

void params_from_yaml_node(const YamlNode& cfg, mola_icp_params& p)
{
    params_default(p);
    if (!cfg.is_map()) throw std::runtime_error("ICP settings must be a YAML map");
    require(cfg, "icp_class");  // YAML_LOAD_REQ(icp_class) cpp:63
    const std::string icp_class = cfg.at("icp_class").as_string();
    bool known = false;
    for (const char* k : kKnownIcpClasses) known |= (icp_class == k);
    if (!known)
        throw std::runtime_error("icp_class=`" + icp_class +
                                 "` is a non-registered or incompatible class. Known classes: `mp2p_icp::ICP`, "
                                 "`mola_icp_amd::ICP_MI355X`.");
    require(cfg, "params");
    require(cfg, "solvers");
    require(cfg, "matchers");
    require(cfg, "quality");

    const YamlNode& pr = cfg.at("params");
    if (pr.is_map()) {
        if (auto* n = pr.find("maxIterations")) p.max_iterations = (uint32_t)n->as_int();
        if (auto* n = pr.find("minAbsStep_trans")) p.min_abs_step_trans = n->as_double();
        if (auto* n = pr.find("minAbsStep_rot")) p.min_abs_step_rot = n->as_double();
        if (auto* w = pr.find("pairingsWeightParameters")) {
            if (auto* n = w->find("use_scale_outlier_detector")) p.use_scale_outlier_detector = n->as_bool();
            if (auto* n = w->find("scale_outlier_threshold")) p.scale_outlier_threshold = n->as_double();
            if (auto* n = w->find("use_robust_kernel")) p.use_robust_kernel = n->as_bool();
            if (auto* n = w->find("robust_kernel_param")) p.robust_kernel_param = n->as_double() * M_PI / 180.0;  // [deg]
            if (auto* n = w->find("robust_kernel_scale")) p.robust_kernel_scale = n->as_double();
        }
        // keys of this implementation (absent in the reference -> defaults)
        if (auto* n = pr.find("fixed_iterations")) p.fixed_iterations = n->as_bool();
        if (auto* n = pr.find("skip_quality")) p.skip_quality = n->as_bool();
        if (auto* n = pr.find("nn_kernel")) {
            const std::string s = n->as_string();
            if (s == "auto") p.nn_kernel = MOLA_ICP_NN_AUTO;
            else if (s == "valu") p.nn_kernel = MOLA_ICP_NN_VALU;
            else if (s == "mfma") p.nn_kernel = MOLA_ICP_NN_MFMA;
            else if (s == "tiled") p.nn_kernel = MOLA_ICP_NN_TILED;
            else throw std::runtime_error("nn_kernel=`" + s + "` is not one of auto|valu|mfma|tiled");
        }
    }

    // the readings (a key of this implementation: absent in the reference -> all 0 = the default readings)
    if (auto* rd = cfg.find("readings")) {
        if (!rd->is_map()) throw std::runtime_error("`readings` must be a map of booleans");
        if (auto* n = rd->find("outlier_single_pass")) p.reading_outlier_single_pass = n->as_bool();
        if (auto* n = rd->find("p2pl_all_inside_gate")) p.reading_p2pl_all_inside_gate = n->as_bool();
        if (auto* n = rd->find("quality_denominator_local")) p.reading_quality_denominator_local = n->as_bool();
        if (auto* n = rd->find("robust_kernel_skips_planes")) p.reading_robust_kernel_skips_planes = n->as_bool();
    }

    {
        const YamlNode& seq = cfg.at("solvers");
        parse_solver(entry_of_seq(seq, "solvers", 0, 1 + MOLA_ICP_MAX_EXTRA_STAGES), p.solver_class, p.solver_max_iterations,
                     p.solver_run_from_iteration, p.solver_run_up_to_iteration);
        p.n_extra_solvers = (uint32_t)seq.seq.size() - 1;
        for (uint32_t i = 0; i < p.n_extra_solvers; ++i) {
            mola_icp_solver_entry& e = p.extra_solvers[i];
            e = mola_icp_solver_entry{};
            parse_solver(entry_of_seq(seq, "solvers", i + 1, 1 + MOLA_ICP_MAX_EXTRA_STAGES), e.solver_class, e.solver_max_iterations,
                         e.run_from_iteration, e.run_up_to_iteration);
        }
    }
    {
        const YamlNode& seq = cfg.at("matchers");
        parse_matcher(entry_of_seq(seq, "matchers", 0, 1 + MOLA_ICP_MAX_EXTRA_STAGES), p.matcher_class, p.matcher_threshold,
                      p.plane_eigen_threshold, p.knn, p.run_from_iteration, p.run_up_to_iteration);
        p.n_extra_matchers = (uint32_t)seq.seq.size() - 1;
        for (uint32_t i = 0; i < p.n_extra_matchers; ++i) {
            mola_icp_matcher_entry& e = p.extra_matchers[i];
            e = mola_icp_matcher_entry{};
            e.matcher_threshold = 0.50; e.plane_eigen_threshold = 0.07; e.knn = 6;   // (the defaults of entry 0: params_default)
            parse_matcher(entry_of_seq(seq, "matchers", i + 1, 1 + MOLA_ICP_MAX_EXTRA_STAGES), e.matcher_class, e.matcher_threshold,
                          e.plane_eigen_threshold, e.knn, e.run_from_iteration, e.run_up_to_iteration);
        }
    }
    {
        const YamlNode& seq = cfg.at("quality");
        auto parse_quality = [](const YamlNode& q, int32_t& cls_out, double& thr, double& weight) {
            const std::string cls = q.at("class").as_string();
            if (cls != "mp2p_icp::QualityEvaluator_PairedRatio")
                throw std::runtime_error("quality class=`" + cls + "` is a non-registered class. Known: "
                                         "`mp2p_icp::QualityEvaluator_PairedRatio`.");
            cls_out = MOLA_ICP_QUALITY_PAIRED_RATIO;
            if (auto* qp = q.find("params"))
                if (auto* n = qp->find("thresholdDistance")) thr = n->as_double();
            if (auto* n = q.find("weight")) weight = n->as_double();
        };
        p.quality_weight = 1.0;
        parse_quality(entry_of_seq(seq, "quality", 0, 1 + MOLA_ICP_MAX_EXTRA_STAGES), p.quality_class, p.quality_threshold, p.quality_weight);
        p.n_extra_quality = (uint32_t)seq.seq.size() - 1;
        for (uint32_t i = 0; i < p.n_extra_quality; ++i) {
            mola_icp_quality_entry& e = p.extra_quality[i];
            e = mola_icp_quality_entry{};
            e.quality_threshold = 0.10; e.weight = 1.0;
            parse_quality(entry_of_seq(seq, "quality", i + 1, 1 + MOLA_ICP_MAX_EXTRA_STAGES), e.quality_class, e.quality_threshold, e.weight);
        }
    }
}

}  // namespace mola_icp_amd
