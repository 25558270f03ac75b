// fake_backend.cpp -- TEST INFRASTRUCTURE, never part of the product library.
//
// A host-only stand-in for the device translation units (csrc/hip_backend.hip, map_sort.hip, rccl_dl.cpp): it defines the members of
// HipWorkspace / HipBatch / DevBuf that the library's HOST code calls, on plain host memory, so that the REAL csrc/c_api.cpp,
// icp_loop.cpp, config.cpp, lidar_odometry_core.cpp, nearby_checks.cpp and local_comm.cpp can be linked into a CPU executable and
// their threaded scaffolding -- the workspace lease pool, the lanes and prepare-ahead tasks of mola_icp_align_batch, the deferred
// builds of mola_icp_align_cached_put, the cloud cache under concurrent put / align / drop, the multi-init batch -- run under
// ThreadSanitizer and AddressSanitizer (those paths only exist with a device, where no sanitizer runs).
// The reference's threading contract they implement: src/LidarOdometry.cpp:94-96 (pool threads), :869 (one ICP object, many callers).
//
// The stages are a small brute-force point-to-point ICP (exact NN over tiny clouds, the 24 sums, the same outlier / robust-kernel
// rules): results are deterministic, so the threaded run can be compared bit for bit with a serial one.  Sleeps of a few
// microseconds inside the stages widen the windows in which threads interleave.  This file is compiled by tests/hosts/Makefile.race
// only; the product has no CPU path (mola_icp_create fails without a gfx950 device).
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../mola-fe-lidar_amd/csrc/hip_backend.hpp"

// ---- the three HIP entry points csrc/c_api.cpp calls itself ------------------------------------------------------------
extern "C" {
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
}

namespace mola_icp_amd {

std::atomic<long> g_fake_live_workspaces{0}, g_fake_peak_workspaces{0}, g_fake_builds{0}, g_fake_matches{0};

static void jitter()
{
    static std::atomic<unsigned> c{0};
    const unsigned k = c.fetch_add(1, std::memory_order_relaxed);
    if ((k & 3u) == 0u) std::this_thread::sleep_for(std::chrono::microseconds(5 + (k % 7u) * 3u));
    else std::this_thread::yield();
}

// ---- RCCL: not there ------------------------------------------------------------------------------------------------------
int rccl_set_library(const char*) { return MOLA_ICP_OK; }
int rccl_unique_id(RcclUniqueId*) { return fail(MOLA_ICP_E_COMM, "fake backend: no RCCL"); }
int rccl_comm_init(void**, int, const RcclUniqueId&, int) { return fail(MOLA_ICP_E_COMM, "fake backend: no RCCL"); }
int rccl_allreduce_sum_f64(void*, double*, size_t, hipStream_t) { return fail(MOLA_ICP_E_COMM, "fake backend: no RCCL"); }
int rccl_comm_count(void*, int*) { return fail(MOLA_ICP_E_COMM, "fake backend: no RCCL"); }
int rccl_comm_destroy(void*) { return MOLA_ICP_OK; }
void reload_env_knobs() {}
static std::atomic<int> g_fake_wait_policy{0};
void set_wait_policy(int policy) { g_fake_wait_policy = policy; }
int wait_policy() { return g_fake_wait_policy; }

// ---- "device" memory = host memory ---------------------------------------------------------------------------------------------
int DevBuf::reserve(size_t bytes)
{
    if (bytes <= cap && p) return MOLA_ICP_OK;
    release();
    const size_t want = bytes < 256 ? 256 : bytes;
    p = std::malloc(want);
    if (!p) return fail(MOLA_ICP_E_OOM, "fake backend: malloc failed");
    std::memset(p, 0xcd, want);   // (uninitialised "device" memory must never be read as data)
    cap = want;
    return MOLA_ICP_OK;
}
void DevBuf::release()
{
    std::free(p);
    p = nullptr;
    cap = 0;
}
void device_pool_trim(size_t, int) {}
size_t device_pool_bytes(int) { return 0; }
void BatchScratch::release_all()
{
    for (BatchBuffers& b : bufs) {
        b.pos.release(); b.idx.release(); b.d2.release(); b.gs.release(); b.rows.release(); b.outlier.release(); b.partials.release();
        b.planes.release(); b.plane_cache.release(); b.knn_pos.release(); b.knn_lb.release(); b.plane_partials.release();
    }
    bufs.clear();
}

// ---- the brute-force stages ---------------------------------------------------------------------------------------------------
namespace {
void nn_brute(const float* gx, const float* gy, const float* gz, size_t M, const float* lx, const float* ly, const float* lz, size_t N,
              const Mat4& T, float thr2, int* idx, float* d2)
{
    float R[9], t[3];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) R[3 * r + c] = (float)T(r, c);
        t[r] = (float)T(r, 3);
    }
    for (size_t i = 0; i < N; ++i) {
        const float qx = std::fmaf(R[2], lz[i], std::fmaf(R[1], ly[i], std::fmaf(R[0], lx[i], t[0])));
        const float qy = std::fmaf(R[5], lz[i], std::fmaf(R[4], ly[i], std::fmaf(R[3], lx[i], t[1])));
        const float qz = std::fmaf(R[8], lz[i], std::fmaf(R[7], ly[i], std::fmaf(R[6], lx[i], t[2])));
        int best = -1;
        float bd = thr2;
        for (size_t j = 0; j < M; ++j) {
            const float dx = qx - gx[j], dy = qy - gy[j], dz = qz - gz[j];
            const float d = std::fmaf(dz, dz, std::fmaf(dy, dy, dx * dx));
            if (d < bd) { bd = d; best = (int)j; }
        }
        idx[i] = best;
        d2[i] = bd;
    }
}

void sums(const float* gx, const float* gy, const float* gz, const float* lx, const float* ly, const float* lz, size_t N, const int* idx,
          const float* d2, unsigned char* outlier, const mola_icp_params& p, const Mat4& Tcur, int stage, const double cl[3],
          const double cg[3], double acc[kNAcc])
{
    for (int k = 0; k < kNAcc; ++k) acc[k] = 0.0;
    for (size_t i = 0; i < N; ++i) {
        const int j = idx[i];
        if (j < 0 || outlier[i]) continue;
        const double l[3] = {lx[i], ly[i], lz[i]}, g[3] = {gx[j], gy[j], gz[j]};
        double w = 1.0;
        if (stage == 1) {
            double b[3] = {g[0] - cg[0], g[1] - cg[1], g[2] - cg[2]}, r[3] = {l[0] - cl[0], l[1] - cl[1], l[2] - cl[2]};
            const double bn = std::sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]), rn = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
            if (bn < 1e-4 || rn < 1e-4) continue;
            if (p.use_scale_outlier_detector && (bn > rn ? bn / rn : rn / bn) > p.scale_outlier_threshold) { outlier[i] = 1; continue; }
            if (p.use_robust_kernel) {
                for (int k = 0; k < 3; ++k) { b[k] /= bn; r[k] /= rn; }
                double c = 0;
                for (int a = 0; a < 3; ++a) c += (Tcur(a, 0) * r[0] + Tcur(a, 1) * r[1] + Tcur(a, 2) * r[2]) * b[a];
                c = c > 1 ? 1 : (c < -1 ? -1 : c);
                const double ang = std::acos(c);
                if (ang > p.robust_kernel_param) w = 1.0 / (1.0 + p.robust_kernel_scale * (ang - p.robust_kernel_param) * (ang - p.robust_kernel_param));
            }
        }
        acc[0] += w;
        for (int k = 0; k < 3; ++k) { acc[1 + k] += w * l[k]; acc[4 + k] += w * g[k]; }
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) acc[7 + 3 * r + c] += w * l[r] * g[c];
        acc[16] += 1.0;
        acc[17] += (double)d2[i];
        acc[18] += w * l[0] * l[0]; acc[19] += w * l[0] * l[1]; acc[20] += w * l[0] * l[2];
        acc[21] += w * l[1] * l[1]; acc[22] += w * l[1] * l[2]; acc[23] += w * l[2] * l[2];
    }
}

int copy_soa(DevBuf& buf, const float* x, const float* y, const float* z, size_t n, const float** dx, const float** dy, const float** dz)
{
    const size_t np = (n + 63) / 64 * 64;
    const int rc = buf.reserve(sizeof(float) * 3 * (np ? np : 64));
    if (rc) return rc;
    float* b = buf.as<float>();
    if (n) {
        std::memcpy(b, x, sizeof(float) * n);
        std::memcpy(b + np, y, sizeof(float) * n);
        std::memcpy(b + 2 * np, z, sizeof(float) * n);
    }
    *dx = b; *dy = b + np; *dz = b + 2 * np;
    return MOLA_ICP_OK;
}
}  // namespace

// ---- HipWorkspace -----------------------------------------------------------------------------------------------------------------
HipWorkspace::HipWorkspace(int device, int priority) : device_(device), priority_(priority), map_sc_(std::make_shared<SortedCloud>()), loc_sc_(std::make_shared<SortedCloud>())
{
    const long live = g_fake_live_workspaces.fetch_add(1) + 1;
    long peak = g_fake_peak_workspaces.load();
    while (live > peak && !g_fake_peak_workspaces.compare_exchange_weak(peak, live)) {}
}
HipWorkspace::~HipWorkspace()
{
    g_fake_live_workspaces.fetch_sub(1);
    map_own_.release(); loc_own_.release(); idx_.release(); d2_.release(); outlier_.release();
    batch_scratch_.release_all();
}
int HipWorkspace::init() { inited_ = true; return MOLA_ICP_OK; }
int HipWorkspace::set_external_stream(void*) { return init(); }
int HipWorkspace::set_map_host(const float* x, const float* y, const float* z, size_t M, bool)
{
    if (M && (!x || !y || !z)) return fail(MOLA_ICP_E_BADARG, "null map pointer");
    const int rc = copy_soa(map_own_, x, y, z, M, &gx_, &gy_, &gz_);
    if (rc) return rc;
    M_ = M;
    if (map_sc_->cached) map_sc_ = std::make_shared<SortedCloud>();
    pairing_valid_ = false;
    return MOLA_ICP_OK;
}
int HipWorkspace::set_map_device(const float* x, const float* y, const float* z, size_t M) { gx_ = x; gy_ = y; gz_ = z; M_ = M; pairing_valid_ = false; return MOLA_ICP_OK; }
int HipWorkspace::set_local_host(const float* x, const float* y, const float* z, size_t N, bool)
{
    if (N && (!x || !y || !z)) return fail(MOLA_ICP_E_BADARG, "null local-cloud pointer");
    const int rc = copy_soa(loc_own_, x, y, z, N, &lx_, &ly_, &lz_);
    if (rc) return rc;
    N_ = N;
    if (loc_sc_->cached) loc_sc_ = std::make_shared<SortedCloud>();
    pairing_valid_ = false;
    return MOLA_ICP_OK;
}
int HipWorkspace::set_local_device(const float* x, const float* y, const float* z, size_t N) { lx_ = x; ly_ = y; lz_ = z; N_ = N; pairing_valid_ = false; return MOLA_ICP_OK; }
int HipWorkspace::set_local_shard(const float*, const float*, const float*, size_t, int, int, bool) { return fail(MOLA_ICP_E_UNSUPPORTED, "fake backend: no shards"); }
int HipWorkspace::set_local_shard_range(const float*, const float*, const float*, size_t, size_t, size_t, bool) { return fail(MOLA_ICP_E_UNSUPPORTED, "fake backend: no shards"); }
int HipWorkspace::copy_shard_indices(int32_t*) { return fail(MOLA_ICP_E_UNSUPPORTED, "fake backend: no shards"); }
int HipWorkspace::set_map_slab(const float*, const float*, const float*, size_t, const double[3], const double[3], bool, size_t*) { return fail(MOLA_ICP_E_UNSUPPORTED, "fake backend: no slabs"); }
int HipWorkspace::shard_reach_box(const Mat4&, double, double[3], double[3]) { return fail(MOLA_ICP_E_UNSUPPORTED, "fake backend: no shards"); }

int HipWorkspace::match(const Mat4& T, double threshold, const mola_icp_params&, uint64_t* n_pairs)
{
    g_fake_matches.fetch_add(1);
    int rc;
    if ((rc = idx_.reserve(sizeof(int) * (N_ ? N_ : 1)))) return rc;
    if ((rc = d2_.reserve(sizeof(float) * (N_ ? N_ : 1)))) return rc;
    const size_t before = outlier_.cap;
    if ((rc = outlier_.reserve(N_ ? N_ : 1))) return rc;
    if (outlier_.cap != before) std::memset(outlier_.p, 0, outlier_.cap);
    jitter();
    nn_brute(gx_, gy_, gz_, M_, lx_, ly_, lz_, N_, T, (float)(threshold * threshold), idx_.as<int>(), d2_.as<float>());
    pairing_valid_ = true;
    if (n_pairs) {
        uint64_t n = 0;
        for (size_t i = 0; i < N_; ++i) n += idx_.as<int>()[i] >= 0;
        *n_pairs = n;
    }
    return MOLA_ICP_OK;
}
int HipWorkspace::accumulate(const mola_icp_params& p, const Mat4& Tcur, int stage, const double cl[3], const double cg[3], bool reset_outliers,
                             double acc[kNAcc])
{
    if (!pairing_valid_) return fail(MOLA_ICP_E_BADARG, "accumulate() called before match()");
    if (reset_outliers && N_) std::memset(outlier_.p, 0, N_);
    jitter();
    sums(gx_, gy_, gz_, lx_, ly_, lz_, N_, idx_.as<int>(), d2_.as<float>(), outlier_.as<unsigned char>(), p, Tcur, stage, cl, cg, acc);
    return MOLA_ICP_OK;
}
int HipWorkspace::allreduce(double acc[kNAcc])
{
    if (!ar_fn_) return MOLA_ICP_OK;
    const int rc = ar_fn_(acc, kNAcc, 0, ar_user_);
    return rc ? fail(MOLA_ICP_E_COMM, "all-reduce hook failed") : MOLA_ICP_OK;
}
int HipWorkspace::quality_pairs(const Mat4&, double, const mola_icp_params&, double[kNAcc], bool* done) { *done = false; return MOLA_ICP_OK; }
int HipWorkspace::copy_pairing(int32_t* idx_out, float* d2_out)
{
    if (!pairing_valid_) return fail(MOLA_ICP_E_BADARG, "no pairing stored");
    if (idx_out) std::memcpy(idx_out, idx_.p, sizeof(int) * N_);
    if (d2_out) std::memcpy(d2_out, d2_.p, sizeof(float) * N_);
    return MOLA_ICP_OK;
}
int HipWorkspace::match_planes(const Mat4&, const mola_icp_params&) { return fail(MOLA_ICP_E_UNSUPPORTED, "fake backend: point-to-point only"); }
int HipWorkspace::accumulate_planes(double[kNAccPlaneHost]) { return fail(MOLA_ICP_E_UNSUPPORTED, "fake backend: point-to-point only"); }
int HipWorkspace::copy_planes(uint8_t*, double*, double*, int32_t*) { return fail(MOLA_ICP_E_UNSUPPORTED, "fake backend: point-to-point only"); }

int HipWorkspace::build_cached(SortedCloud& sc, const float* x, const float* y, const float* z, size_t n)
{
    std::shared_ptr<SortedCloud> alias(&sc, [](SortedCloud*) {});
    return build_cached(alias, x, y, z, n, true);
}
int HipWorkspace::build_cached(const std::shared_ptr<SortedCloud>& scp, const float* x, const float* y, const float* z, size_t n, bool)
{
    g_fake_builds.fetch_add(1);
    if (n && (!x || !y || !z)) return fail(MOLA_ICP_E_BADARG, "null cloud pointer");
    SortedCloud& sc = *scp;
    jitter();
    const int rc = copy_soa(sc.raw, x, y, z, n, &sc.x, &sc.y, &sc.z);
    if (rc) return rc;
    for (size_t i = 0; i < n; ++i)
        if (!std::isfinite(x[i]) || !std::isfinite(y[i]) || !std::isfinite(z[i])) { sc.ready = false; return fail(MOLA_ICP_E_BADARG, "a cloud has non-finite coordinates"); }
    sc.n = n;
    sc.cached = true;
    sc.ready = true;
    return MOLA_ICP_OK;
}
int HipWorkspace::finish_build(SortedCloud&) { jitter(); return MOLA_ICP_OK; }
hipError_t HipWorkspace::quick_sync() { return hipSuccess; }
void HipWorkspace::use_cached_map(const std::shared_ptr<SortedCloud>& sc) { map_sc_ = sc; gx_ = sc->x; gy_ = sc->y; gz_ = sc->z; M_ = sc->n; pairing_valid_ = false; }
void HipWorkspace::use_cached_local(const std::shared_ptr<SortedCloud>& sc) { loc_sc_ = sc; lx_ = sc->x; ly_ = sc->y; lz_ = sc->z; N_ = sc->n; pairing_valid_ = false; }
int HipWorkspace::voxel_downsample(const float*, const float*, const float*, size_t, double, float*, float*, float*, size_t, size_t*) { return fail(MOLA_ICP_E_UNSUPPORTED, "fake backend: no voxel filter"); }
int HipWorkspace::sync() { return MOLA_ICP_OK; }
void HipWorkspace::forget_warm_start(bool) {}
void HipWorkspace::reset_stats() { nn_launches_ = 0; }
int HipWorkspace::collect_stats(double* ms_total, uint32_t* launches, uint32_t* kernel_used, uint64_t* pairs)
{
    if (ms_total) *ms_total = 0;
    if (launches) *launches = nn_launches_;
    if (kernel_used) *kernel_used = MOLA_ICP_NN_VALU;
    if (pairs) *pairs = 0;
    return MOLA_ICP_OK;
}

// ---- HipBatch: K problems, one after the other inside every "launch" ---------------------------------------------------------------
HipBatch::HipBatch(HipWorkspace& ws, std::vector<BatchProblem> probs) : ws_(ws), sc_(ws.batch_scratch_), probs_(std::move(probs)) {}
HipBatch::~HipBatch() {}
int HipBatch::init()
{
    const size_t K = probs_.size();
    if (sc_.bufs.size() < K) sc_.bufs.resize(K);
    for (size_t k = 0; k < K; ++k) {
        const BatchProblem& pr = probs_[k];
        if (!pr.map || !pr.loc) return fail(MOLA_ICP_E_BADARG, "batch problem without clouds");
        if (pr.loc->n == 0 || pr.map->n == 0) continue;
        if (!pr.map->ready || !pr.loc->ready) return fail(MOLA_ICP_E_INTERNAL, "batch problem with an unprepared cloud");
        BatchBuffers& b = sc_.bufs[k];
        int rc;
        if ((rc = b.idx.reserve(sizeof(int) * pr.loc->n))) return rc;
        if ((rc = b.d2.reserve(sizeof(float) * pr.loc->n))) return rc;
        if ((rc = b.outlier.reserve(pr.loc->n))) return rc;
        std::memset(b.outlier.p, 0, pr.loc->n);
    }
    inited_ = true;
    return MOLA_ICP_OK;
}
int HipBatch::init_planes(int) { return fail(MOLA_ICP_E_UNSUPPORTED, "fake backend: point-to-point only"); }
int HipBatch::match(const uint8_t* active, const Mat4* T, double threshold, const mola_icp_params&)
{
    ++nn_launches_;
    jitter();
    for (size_t k = 0; k < probs_.size(); ++k) {
        if (!active[k]) continue;
        const BatchProblem& pr = probs_[k];
        BatchBuffers& b = sc_.bufs[k];
        nn_brute(pr.map->x, pr.map->y, pr.map->z, pr.map->n, pr.loc->x, pr.loc->y, pr.loc->z, pr.loc->n, T[k], (float)(threshold * threshold), b.idx.as<int>(), b.d2.as<float>());
    }
    return MOLA_ICP_OK;
}
int HipBatch::accumulate(const uint8_t* active, const mola_icp_params& p, const Mat4* Tcur, int stage, const double (*cl)[3], const double (*cg)[3],
                         bool reset_outliers, double (*acc)[kNAcc])
{
    jitter();
    for (size_t k = 0; k < probs_.size(); ++k) {
        if (!active[k]) continue;
        const BatchProblem& pr = probs_[k];
        BatchBuffers& b = sc_.bufs[k];
        if (reset_outliers) std::memset(b.outlier.p, 0, pr.loc->n);
        sums(pr.map->x, pr.map->y, pr.map->z, pr.loc->x, pr.loc->y, pr.loc->z, pr.loc->n, b.idx.as<int>(), b.d2.as<float>(), b.outlier.as<unsigned char>(), p,
             Tcur[k], stage, cl ? cl[k] : nullptr, cg ? cg[k] : nullptr, acc[k]);
    }
    return MOLA_ICP_OK;
}
int HipBatch::match_planes(const uint8_t*, const Mat4*, const mola_icp_params&) { return fail(MOLA_ICP_E_UNSUPPORTED, "fake backend: point-to-point only"); }
int HipBatch::accumulate_planes(const uint8_t*, double (*)[kNAccPlaneHost]) { return fail(MOLA_ICP_E_UNSUPPORTED, "fake backend: point-to-point only"); }
int HipBatch::collect_stats(double* ms_total, uint32_t* launches, uint64_t* pairs)
{
    if (ms_total) *ms_total = 0;
    if (launches) *launches = nn_launches_;
    if (pairs) *pairs = 0;
    return MOLA_ICP_OK;
}

}  // namespace mola_icp_amd
