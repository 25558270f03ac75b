// lidar_odometry_core.cpp -- the LidarOdometry front-end logic AROUND the ICP call:
// what LidarOdometry::doProcessNewObservation() does between receiving a point cloud and publishing a
// relative pose / keyframe decision (src/LidarOdometry.cpp:190-514), minus MOLA's back-end, world model and
// GUI plumbing.  Pure host logic; the registration itself is the C-ABI's mola_icp_align (or an injected
// function, which is how this logic is tested without a GPU).
#include <atomic>
#include <cmath>
#include <cstring>
#include <memory>
#include <vector>

#include "../../include/mola_icp_amd.h"
#include "icp_loop.hpp"
#include "se3_math.hpp"
#include "yaml_lite.hpp"

namespace mola_icp_amd {
void params_from_yaml_node(const YamlNode& cfg, mola_icp_params& p);
void params_compose(const mola_icp_params& object_settings, const mola_icp_params& call_parameters, mola_icp_params& out);
}
using namespace mola_icp_amd;

struct mola_lo {
    mola_lo_params params{};
    mola_icp_handle* icp = nullptr;   // AlignKind::LidarOdometry's ICP object (LidarOdometry.h:118, cpp:869)
    mola_lo_align_fn align_cb = nullptr;
    void* align_user = nullptr;

    // MethodState (LidarOdometry.h:136-160), the fields this path uses
    bool have_last_tim = false;
    double last_obs_tim = 0;
    std::vector<float> last_x, last_y, last_z;   // last_points
    bool have_last_points = false;               // a (possibly empty) cloud has been stored
    double twist[4] = {0, 0, 0, 0};              // last_iter_twist: vx, vy, vz, wz
    bool twist_is_good = false;
    Mat4 accum_since_last_kf = Mat4::identity();
    bool have_kf = false;
    uint64_t last_kf = 0, next_kf_id = 0;

    // with a GPU ICP handle the scans live in its cloud cache (row f4): a scan is uploaded + sorted once, is the `to`
    // cloud of this step and the `from` cloud of the next, then dropped.  Ids are taken from a reserved range.
    uint64_t cache_base = 0, scan_no = 0;
    bool have_cached_last = false;
    uint64_t cached_last_id = 0;
};

namespace {
void fill_step_pose(double dst[16], const Mat4& m) { std::memcpy(dst, m.m, sizeof m.m); }
}

extern "C" {

int mola_lo_params_default(mola_lo_params* p)
{
    if (!p) return fail(MOLA_ICP_E_BADARG, "null params");
    std::memset(p, 0, sizeof *p);
    p->min_time_between_scans = 0.2;                       // LidarOdometry.h:57
    p->min_dist_xyz_between_keyframes = 1.0;               // h:61
    p->min_rotation_between_keyframes = 30.0 * M_PI / 180; // h:66
    p->min_icp_goodness = 0.4;                             // h:70
    mola_icp_params_default(&p->icp_with_vel);
    mola_icp_params_default(&p->icp_without_vel);
    mola_icp_params_default(&p->icp_loop_closure);
    p->min_icp_goodness_lc = 0.6;                          // h:73
    p->min_dist_to_matching = 6.0;                         // h:83
    p->max_dist_to_matching = 12.0;                        // h:84
    p->max_dist_to_loop_closure = 30.0;                    // h:85
    p->loop_closure_montecarlo_samples = 10;               // h:86
    p->max_nearby_align_checks = 2;                        // h:87
    p->min_topo_dist_to_consider_loopclosure = 20;         // h:88
    p->max_kfs_local_graph = 50000;                        // h:90
    return MOLA_ICP_OK;
}

// Same keys as LidarOdometry::initialize() reads (src/LidarOdometry.cpp:105-128); `params:` wrapper optional.
int mola_lo_params_from_yaml_file(const char* path, const char* mola_dir, mola_lo_params* p)
{
    if (!path || !p) return fail(MOLA_ICP_E_BADARG, "null argument");
    try {
        mola_lo_params_default(p);
        const std::string sp(path);
        YamlNode root = yaml_parse(read_text_file(sp));
        const size_t slash = sp.find_last_of('/');
        yaml_resolve_includes(root, slash == std::string::npos ? std::string(".") : sp.substr(0, slash),
                              mola_dir ? std::string(mola_dir) : std::string());
        const YamlNode* cfg = &root;
        if (root.has("params") && root.at("params").is_map()) cfg = &root.at("params");  // cfg = c["params"] cpp:102
        p->min_dist_xyz_between_keyframes = cfg->at("min_dist_xyz_between_keyframes").as_double();  // YAML_LOAD_REQ cpp:105
        if (auto* n = cfg->find("min_rotation_between_keyframes"))
            p->min_rotation_between_keyframes = n->as_double() * M_PI / 180.0;                       // _OPT_DEG cpp:106
        if (auto* n = cfg->find("min_time_between_scans")) p->min_time_between_scans = n->as_double();
        if (auto* n = cfg->find("min_icp_goodness")) p->min_icp_goodness = n->as_double();
        if (auto* n = cfg->find("min_icp_goodness_lc")) p->min_icp_goodness_lc = n->as_double();             // cpp:110
        if (auto* n = cfg->find("min_dist_to_matching")) p->min_dist_to_matching = n->as_double();           // cpp:112
        if (auto* n = cfg->find("max_dist_to_matching")) p->max_dist_to_matching = n->as_double();           // cpp:113
        if (auto* n = cfg->find("max_dist_to_loop_closure")) p->max_dist_to_loop_closure = n->as_double();   // cpp:114
        if (auto* n = cfg->find("max_nearby_align_checks")) p->max_nearby_align_checks = (uint32_t)n->as_int();  // cpp:115
        if (auto* n = cfg->find("min_topo_dist_to_consider_loopclosure"))
            p->min_topo_dist_to_consider_loopclosure = (uint32_t)n->as_int();                                // cpp:116
        if (auto* n = cfg->find("loop_closure_montecarlo_samples"))
            p->loop_closure_montecarlo_samples = (uint32_t)n->as_int();                                      // cpp:117
        // (keys the reference's file carries but its code never reads -- decimate_to_point_count, pointcloud_filter_*,
        //  debug_save_*: SURVEY.md §5 "stale-config warning" -- are accepted and ignored, as the reference does)
        if (!cfg->has("icp_settings_with_vel")) throw std::runtime_error("Missing YAML required entry `icp_settings_with_vel`");
        params_from_yaml_node(cfg->at("icp_settings_with_vel"), p->icp_with_vel);          // cpp:122-124
        params_from_yaml_node(cfg->at("icp_settings_without_vel"), p->icp_without_vel);    // cpp:125-126
        params_from_yaml_node(cfg->at("icp_settings_loop_closure"), p->icp_loop_closure);  // cpp:127-128
        return MOLA_ICP_OK;
    } catch (const std::exception& e) {
        return fail(MOLA_ICP_E_CONFIG, e.what());
    }
}

int mola_lo_create(mola_icp_handle* icp, mola_lo_align_fn align_cb, void* user, const mola_lo_params* params,
                   mola_lo** out)
{
    if (!out || !params) return fail(MOLA_ICP_E_BADARG, "null argument");
    if (!icp && !align_cb) return fail(MOLA_ICP_E_BADARG, "need an ICP handle or an align function");
    try {
        std::unique_ptr<mola_lo> lo(new mola_lo);
        lo->params = *params;
        lo->icp = icp;
        lo->align_cb = align_cb;
        lo->align_user = user;
        static std::atomic<uint64_t> instance{0};
        lo->cache_base = (0xF1ull << 56) | ((instance.fetch_add(1) & 0xFFFFFFull) << 32);
        *out = lo.release();
        return MOLA_ICP_OK;
    } catch (const std::exception& e) {
        return fail(MOLA_ICP_E_INTERNAL, e.what());
    }
}

int mola_lo_destroy(mola_lo* lo)
{
    if (lo && lo->icp && lo->have_cached_last) (void)mola_icp_cloud_drop(lo->icp, lo->cached_last_id);
    delete lo;
    return MOLA_ICP_OK;
}

// LidarOdometry::reset(): state_ = MethodState() (src/LidarOdometry.cpp:160)
int mola_lo_reset(mola_lo* lo)
{
    if (!lo) return fail(MOLA_ICP_E_BADARG, "null handle");
    if (lo->icp && lo->have_cached_last) (void)mola_icp_cloud_drop(lo->icp, lo->cached_last_id);
    lo->have_cached_last = false;
    lo->have_last_tim = false;
    lo->last_x.clear(); lo->last_y.clear(); lo->last_z.clear();
    lo->have_last_points = false;
    lo->twist[0] = lo->twist[1] = lo->twist[2] = lo->twist[3] = 0;
    lo->twist_is_good = false;
    lo->accum_since_last_kf = Mat4::identity();
    lo->have_kf = false;
    lo->last_kf = 0;
    lo->next_kf_id = 0;
    return MOLA_ICP_OK;
}

// doProcessNewObservation(): src/LidarOdometry.cpp:190-514
int mola_lo_process_scan(mola_lo* lo, double timestamp, const float* x, const float* y, const float* z, size_t n,
                         mola_lo_step* out)
{
    if (!lo || !out || (n && (!x || !y || !z))) return fail(MOLA_ICP_E_BADARG, "null argument");
    // the odometry step is the call with a deadline (the sensor's rate; the reference sheds load when it falls behind: cpp:171-179): its
    // device work runs at the greatest stream priority, ahead of queued launches of nearby / loop-closure checks on the same handle
    struct HighPriority {
        int before = 0;
        HighPriority() { (void)mola_icp_get_thread_priority(&before); (void)mola_icp_set_thread_priority(1); }
        ~HighPriority() { (void)mola_icp_set_thread_priority(before); }
    } high_priority;
    try {
        std::memset(out, 0, sizeof *out);
        fill_step_pose(out->rel_pose, Mat4::identity());
        fill_step_pose(out->kf_factor_pose, Mat4::identity());
        // time gate (cpp:202-212)
        if (lo->have_last_tim && (timestamp - lo->last_obs_tim) < lo->params.min_time_between_scans) {
            out->status = MOLA_LO_DROPPED_TOO_SOON;
            fill_step_pose(out->accum_since_last_kf, lo->accum_since_last_kf);
            out->reference_kf = lo->last_kf;
            return MOLA_ICP_OK;
        }
        // "Store for next step" (cpp:229-234): the new cloud replaces the old one even when it is empty
        const bool had_last_tim = lo->have_last_tim;
        const double last_obs_tim = lo->last_obs_tim;
        std::vector<float> px, py, pz;
        px.swap(lo->last_x); py.swap(lo->last_y); pz.swap(lo->last_z);
        lo->last_obs_tim = timestamp;
        lo->have_last_tim = true;
        const bool use_cache = lo->icp && !lo->align_cb;
        const bool had_cached = lo->have_cached_last;
        // (with the device cache, "there is a last cloud" means it is IN the cache: a scan whose upload was refused -- non-finite
        // coordinates -- leaves no partner for the next one, which then starts over like a first scan)
        const bool had_points = lo->have_last_points && !px.empty() && (!use_cache || had_cached);
        const uint64_t prev_id = lo->cached_last_id;
        uint64_t cur_id = 0;
        // (a scan that will be aligned against the previous one is put in the cache BY that align: mola_icp_align_cached_put -- the
        // new cloud's prepare chain and the align's first launches share a stream, no host wait in between)
        const bool put_with_align = use_cache && n && had_points && had_cached;
        if (use_cache) {
            lo->last_x.assign(n ? 1 : 0, 0.f);  // the host copy is not needed: only "is there a last cloud"
            lo->last_y.clear(); lo->last_z.clear();
            lo->have_cached_last = false;
            if (n) {
                cur_id = lo->cache_base | (lo->scan_no++ & 0xFFFFFFFFull);
                if (!put_with_align) {
                    const int rc = mola_icp_cloud_put(lo->icp, cur_id, x, y, z, n);
                    if (rc) {
                        if (had_cached) (void)mola_icp_cloud_drop(lo->icp, prev_id);
                        return rc;
                    }
                    lo->have_cached_last = true;
                    lo->cached_last_id = cur_id;
                }
            }
        } else {
            lo->last_x.assign(x, x + n); lo->last_y.assign(y, y + n); lo->last_z.assign(z, z + n);
        }
        lo->have_last_points = true;
        struct DropPrev {  // the previous scan leaves the cache when this step is over, whatever path it takes
            mola_icp_handle* h; bool on; uint64_t id;
            ~DropPrev() { if (on) (void)mola_icp_cloud_drop(h, id); }
        } drop_prev{lo->icp, use_cache && had_cached, prev_id};

        if (n == 0) {  // cpp:238-245: "could not be converted into a pointcloud. Doing nothing."
            out->status = MOLA_LO_EMPTY_CLOUD;
            fill_step_pose(out->accum_since_last_kf, lo->accum_since_last_kf);
            out->reference_kf = lo->last_kf;
            return MOLA_ICP_OK;
        }

        bool create_keyframe = false;
        if (!had_points) {
            // first pointcloud: skip ICP, still create a first KF at the origin (cpp:250-257)
            out->status = MOLA_LO_FIRST_SCAN;
            create_keyframe = true;
        } else {
            out->status = MOLA_LO_ICP_RAN;
            double dt = 0.0;
            if (had_last_tim) dt = timestamp - last_obs_tim;  // cpp:268-269
            // constant-velocity guess: (vx,vy,vz)*dt and yaw = wz*dt only (cpp:272-275, "do omega_xyz part!" TODO)
            const double guess6[6] = {lo->twist[0] * dt, lo->twist[1] * dt, lo->twist[2] * dt, lo->twist[3] * dt, 0, 0};
            const Mat4 guess = pose_from_xyzypr(guess6);
            // Without a trustworthy twist the reference swaps ONLY icp_in.icp_params -- the mp2p_icp::Parameters half
            // (maxIterations, minAbsStep_*, pairingsWeightParameters) -- for the NearbyAlign case's (cpp:287-290);
            // the ICP object stays the AlignKind::LidarOdometry one (in.align_kind keeps its default, h:118; cpp:869),
            // i.e. icp_with_vel's matcher / solver / quality settings run either way.
            mola_icp_params ip = lo->params.icp_with_vel;
            if (!lo->twist_is_good) params_compose(lo->params.icp_with_vel, lo->params.icp_without_vel, ip);
            out->used_with_vel_params = lo->twist_is_good ? 1 : 0;
            // run_one_icp: to = this scan, from = previous scan (cpp:278-279, 299, 869-871)
            int rc;
            if (lo->align_cb)
                rc = lo->align_cb(lo->align_user, px.data(), py.data(), pz.data(), px.size(), x, y, z, n, guess.m, &ip,
                                  &out->icp);
            else if (put_with_align) {  // the previous scan is prepared in HBM; this one is prepared, aligned and cached in one go
                int put_done = 0;
                rc = mola_icp_align_cached_put(lo->icp, prev_id, cur_id, x, y, z, n, guess.m, &ip, &out->icp, &put_done);
                if (put_done) {
                    lo->have_cached_last = true;
                    lo->cached_last_id = cur_id;
                }
            } else  // both scans are already prepared in HBM
                rc = mola_icp_align_cached(lo->icp, prev_id, cur_id, guess.m, &ip, &out->icp);
            if (rc) return rc < 0 ? rc : fail(MOLA_ICP_E_INTERNAL, "align function failed");
            Mat4 rel;
            std::memcpy(rel.m, out->icp.T, sizeof rel.m);  // out.found_pose_to_wrt_from = optimal_tf (cpp:879)
            fill_step_pose(out->rel_pose, rel);
            out->dt = dt;
            // twist update (cpp:305-311); dt == 0 gives inf exactly as the reference's division does
            double rp[6];
            pose_to_xyzypr(rel, rp);
            lo->twist[0] = rp[0] / dt; lo->twist[1] = rp[1] / dt; lo->twist[2] = rp[2] / dt; lo->twist[3] = rp[3] / dt;
            lo->twist_is_good = true;
            // accumulate and decide (cpp:321-337)
            lo->accum_since_last_kf = mul(lo->accum_since_last_kf, rel);
            const Mat4& A = lo->accum_since_last_kf;
            const double dist = std::sqrt(A(0, 3) * A(0, 3) + A(1, 3) * A(1, 3) + A(2, 3) * A(2, 3));
            double lg[6];
            se3_log(A, lg);
            const double rot = std::sqrt(lg[3] * lg[3] + lg[4] * lg[4] + lg[5] * lg[5]);
            out->dist_since_last_kf = dist;
            out->rot_since_last_kf = rot;
            create_keyframe = out->icp.quality > lo->params.min_icp_goodness &&
                              (dist > lo->params.min_dist_xyz_between_keyframes ||
                               rot > lo->params.min_rotation_between_keyframes);
        }
        if (create_keyframe) {  // cpp:342-475, without the back-end / world-model calls
            const uint64_t new_id = lo->next_kf_id++;
            out->keyframe_created = 1;
            if (lo->have_kf) {  // FactorRelativePose3(last_kf, new_kf, accum_since_last_kf) cpp:436-443
                out->kf_factor_valid = 1;
                out->kf_factor_from = lo->last_kf;
                out->kf_factor_to = new_id;
                fill_step_pose(out->kf_factor_pose, lo->accum_since_last_kf);
            }
            lo->accum_since_last_kf = Mat4::identity();  // cpp:472-474
            lo->last_kf = new_id;
            lo->have_kf = true;
        }
        // advertiseUpdatedLocalization (cpp:484-490): reference KF + pose since it
        fill_step_pose(out->accum_since_last_kf, lo->accum_since_last_kf);
        out->reference_kf = lo->last_kf;
        for (int k = 0; k < 4; ++k) out->twist[k] = lo->twist[k];
        return MOLA_ICP_OK;
    } catch (const std::bad_alloc&) {
        return fail(MOLA_ICP_E_OOM, "host allocation failed");
    } catch (const std::exception& e) {
        return fail(MOLA_ICP_E_INTERNAL, e.what());
    }
}

}  // extern "C"
