// c_api.cpp -- the extern "C" boundary declared in include/mola_icp_amd.h.
// No exception, C++ type or HIP type crosses it.
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <future>
#include <map>
#include <chrono>
#include <cstring>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mola_icp_amd.h"
#include "hip_backend.hpp"
#include "icp_loop.hpp"
#include "trace.hpp"
#include "yaml_lite.hpp"

namespace mola_icp_amd {
void params_default(mola_icp_params& p);
void params_from_yaml_node(const YamlNode& cfg, mola_icp_params& p);
void params_compose(const mola_icp_params& object_settings, const mola_icp_params& call_parameters, mola_icp_params& out);
}  // namespace mola_icp_amd

using namespace mola_icp_amd;

struct mola_icp_handle {
    int device = -1;
    std::mutex mtx;                                   // guards pool + resident
    std::vector<std::unique_ptr<HipWorkspace>> pool;  // idle workspaces for mola_icp_align()
    std::vector<std::unique_ptr<HipWorkspace>> pool_hi;  // ... whose streams run at the device's greatest priority (mola_icp_set_thread_priority)
    std::unique_ptr<HipWorkspace> resident;           // the resident-cloud API's workspace
    std::atomic<int> profiling{0};                    // mola_icp_set_profiling
    // prepared-cloud objects of finished batched calls, kept for the next one (their device buffers are reused:
    // hipMalloc / hipFree of a chunk's clouds cost milliseconds and synchronise the device)
    std::mutex spare_mtx;
    std::vector<std::shared_ptr<SortedCloud>> spare_clouds;
    std::shared_ptr<SortedCloud> take_cloud()
    {
        std::lock_guard<std::mutex> lk(spare_mtx);
        if (spare_clouds.empty()) return std::make_shared<SortedCloud>();
        std::shared_ptr<SortedCloud> c = std::move(spare_clouds.back());
        spare_clouds.pop_back();
        return c;
    }
    void give_cloud(std::shared_ptr<SortedCloud>&& c)
    {
        if (!c || c.use_count() != 1) return;  // still shared (a cache entry, another problem): not ours to recycle
        std::lock_guard<std::mutex> lk(spare_mtx);
        if (spare_clouds.size() < 64) spare_clouds.push_back(std::move(c));
    }
    mola_icp_allreduce_fn ar_fn = nullptr;
    mola_icp_local_comm* local_comm = nullptr;        // mola_icp_comm_attach_local (not owned)
    void* ar_user = nullptr;
    void* comm = nullptr;  // RCCL communicator of the query-sharded path
    std::mutex cache_mtx;  // guards the cloud cache (row f4)
    std::map<uint64_t, std::shared_ptr<SortedCloud>> cache;
    // worker threads of mola_icp_align_batch (the analogue of worker_pool_past_KFs_, src/LidarOdometry.cpp:94-96):
    // started on first use, kept for the handle's lifetime
    std::mutex job_mtx;
    std::condition_variable job_cv;
    std::deque<std::function<void()>> jobs;
    std::vector<std::thread> workers;
    bool stopping = false;
    void ensure_workers(size_t n)
    {
        std::lock_guard<std::mutex> lk(job_mtx);
        while (workers.size() < n)
            workers.emplace_back([this]() {
                for (;;) {
                    std::function<void()> job;
                    {
                        std::unique_lock<std::mutex> lk2(job_mtx);
                        job_cv.wait(lk2, [this]() { return stopping || !jobs.empty(); });
                        if (jobs.empty()) return;  // stopping
                        job = std::move(jobs.front());
                        jobs.pop_front();
                    }
                    job();
                }
            });
    }
    void submit(std::function<void()> job)
    {
        {
            std::lock_guard<std::mutex> lk(job_mtx);
            jobs.push_back(std::move(job));
        }
        job_cv.notify_one();
    }
    ~mola_icp_handle()
    {
        {
            std::lock_guard<std::mutex> lk(job_mtx);
            stopping = true;
        }
        job_cv.notify_all();
        for (auto& t : workers) t.join();
        if (resident) resident->sync();
        if (comm) (void)rccl_comm_destroy(comm);
    }
};

// One handle per device slot and one PERSISTENT worker thread per slot (round 6; a std::thread per device per call before).  A call
// posts ONE job: the workers pull chunks of pairs from a shared cursor until it runs dry -- a slot whose pairs stall early takes more
// chunks instead of idling while the others finish a static share (pairs differ in iteration count: src/LidarOdometry.cpp:704-741's
// batch mixes nearby aligns and loop closures).  Every pair's result is that of its stand-alone align, whichever slot served it.
struct mola_icp_pool {
    std::vector<mola_icp_handle*> handles;
    struct Job {
        size_t n_pairs = 0, chunk = 1;
        const float* const* fx = nullptr; const float* const* fy = nullptr; const float* const* fz = nullptr; const size_t* M = nullptr;
        const float* const* tx = nullptr; const float* const* ty = nullptr; const float* const* tz = nullptr; const size_t* N = nullptr;
        const double* init_T = nullptr;
        const mola_icp_params* p = nullptr;
        mola_icp_result* out = nullptr;
    };
    std::mutex m;                 // guards everything below
    std::condition_variable cv_work, cv_done;
    std::mutex call_m;            // one align_batch call at a time per pool
    Job job;
    unsigned long long generation = 0;   // bumped per posted job
    size_t cursor = 0;            // next pair to hand out
    int busy = 0;                 // workers that have not finished the current generation
    bool stop = false;
    std::vector<int> rcs;
    std::vector<std::string> msgs;
    std::vector<size_t> served;   // pairs each slot served in the last call (statistics)
    std::vector<std::thread> workers;

    void worker(int d)
    {
        unsigned long long seen = 0;
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lk(m);
                cv_work.wait(lk, [&] { return stop || generation != seen; });
                if (stop) return;
                seen = generation;
                j = job;
            }
            int rc = MOLA_ICP_OK;
            std::string msg;
            size_t n_served = 0;
            for (;;) {
                size_t lo, hi;
                {
                    std::lock_guard<std::mutex> lk(m);
                    if (cursor >= j.n_pairs) break;
                    lo = cursor;
                    hi = std::min(j.n_pairs, lo + j.chunk);
                    cursor = hi;
                }
                const size_t n = hi - lo;
                rc = mola_icp_align_batch(handles[(size_t)d], n, j.fx + lo, j.fy + lo, j.fz + lo, j.M + lo, j.tx + lo, j.ty + lo, j.tz + lo,
                                          j.N + lo, j.init_T + 16 * lo, j.p, j.out + lo);
                if (rc) { msg = mola_icp_last_error(); break; }
                n_served += n;
            }
            {
                std::lock_guard<std::mutex> lk(m);
                rcs[(size_t)d] = rc;
                msgs[(size_t)d] = msg;
                served[(size_t)d] = n_served;
                if (rc) cursor = j.n_pairs;   // (a failed slot ends the call: the others stop pulling)
                if (--busy == 0) cv_done.notify_all();   // (under the lock: the caller may unwind as soon as it sees busy == 0)
            }
        }
    }
    void start()
    {
        const size_t nd = handles.size();
        rcs.assign(nd, MOLA_ICP_OK); msgs.assign(nd, std::string()); served.assign(nd, 0);
        for (size_t d = 0; d < nd; ++d) workers.emplace_back([this, d] { worker((int)d); });
    }
    ~mola_icp_pool()
    {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
            cv_work.notify_all();
        }
        for (std::thread& t : workers) t.join();
        for (mola_icp_handle* h : handles) delete h;
    }
};

namespace {

constexpr int kBatchChunk = 12;  // pairs advanced together by mola_icp_align_batch (= problems per batched launch)
constexpr size_t kBatchPlanesMaxQueries = 256 * 8 * 64;   // = the stand-alone path's range of k_knn_coop (hip_backend.hip: match_planes)

double now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// Runs `f` when the scope dies, by return or by exception: every frame below that hands references to itself to worker
// threads waits for those workers through one of these, so no worker ever writes into a frame that has been unwound.
template <class F> struct AtScopeExit {
    F f;
    explicit AtScopeExit(F&& fn) : f(std::move(fn)) {}
    AtScopeExit(const AtScopeExit&) = delete;
    AtScopeExit& operator=(const AtScopeExit&) = delete;
    ~AtScopeExit() { f(); }
};
template <class F> AtScopeExit<F> at_scope_exit(F&& f) { return AtScopeExit<F>(std::move(f)); }

template <class F> int guarded(F&& f)
{
    try {
        return f();
    } catch (const std::bad_alloc&) {
        return fail(MOLA_ICP_E_OOM, "host allocation failed");
    } catch (const std::exception& e) {
        return fail(MOLA_ICP_E_INTERNAL, e.what());
    } catch (...) {
        return fail(MOLA_ICP_E_INTERNAL, "unknown C++ exception");
    }
}

Mat4 mat_from(const double T[16])
{
    Mat4 m;
    std::memcpy(m.m, T, sizeof m.m);
    return m;
}

int check_pose(const double T[16])
{
    if (!T) return fail(MOLA_ICP_E_BADARG, "null pose");
    for (int i = 0; i < 16; ++i)
        if (!std::isfinite(T[i])) return fail(MOLA_ICP_E_BADARG, "pose has a non-finite entry");
    return MOLA_ICP_OK;
}

// borrows a workspace from the handle's pool for one align() call
thread_local int t_call_priority = 0;   // mola_icp_set_thread_priority: the class of the workspaces this thread's calls lease

struct Lease {
    mola_icp_handle* h;
    std::unique_ptr<HipWorkspace> ws;
    int rc = MOLA_ICP_OK;
    bool high = false;
    explicit Lease(mola_icp_handle* hh) : h(hh), high(t_call_priority > 0)
    {
        {
            std::lock_guard<std::mutex> lk(h->mtx);
            auto& pool = high ? h->pool_hi : h->pool;
            if (!pool.empty()) {
                ws = std::move(pool.back());
                pool.pop_back();
            }
        }
        if (!ws) {
            ws.reset(new HipWorkspace(h->device, high ? 1 : 0));
            rc = ws->init();
        }
        ws->set_profiling(h->profiling.load() != 0);
    }
    ~Lease()
    {
        if (ws && rc == MOLA_ICP_OK) {
            std::lock_guard<std::mutex> lk(h->mtx);
            (high ? h->pool_hi : h->pool).push_back(std::move(ws));
        }
    }
};

int align_on(HipWorkspace& ws, const double init_T[16], const mola_icp_params* p, mola_icp_result* out)
{
    ws.reset_stats();
    int rc = run_icp_loop(ws, mat_from(init_T), *p, out);
    if (rc) return rc;
    double ms = 0;
    uint32_t n = 0, k = 0;
    uint64_t pairs = 0;
    if ((rc = ws.collect_stats(&ms, &n, &k, &pairs))) return rc;
    out->ms_nn_kernel = ms;
    out->n_nn_launches = n;
    out->nn_kernel_used = k;
    out->nn_pairs_evaluated = pairs;
    return MOLA_ICP_OK;
}

int align_host_clouds(mola_icp_handle* h, const float* fx, const float* fy, const float* fz, size_t M, const float* tx,
                      const float* ty, const float* tz, size_t N, const double init_T[16], const mola_icp_params* p,
                      mola_icp_result* out)
{
    int rc;
    if ((rc = check_pose(init_T))) return rc;
    if ((rc = validate_params(*p))) return rc;
    Lease lease(h);
    if (lease.rc) return lease.rc;
    HipWorkspace& ws = *lease.ws;
    std::memset(out, 0, sizeof *out);
    TraceRange tr("mola_icp.align");   // = the reference's profiler entry "run_one_icp" (cpp:858)
    const double t0 = now_ms();
    {
        TraceRange tr_up("mola_icp.upload");
        // (no host wait behind the uploads: the caller's buffers outlive this frame, and the frame does not end before the stream has
        // drained -- the align's last hand-over on success, the explicit wait below otherwise.  Two waits here cost config 0 ~20 us.)
        if ((rc = ws.set_map_host(fx, fy, fz, M, false))) { (void)ws.sync(); lease.rc = rc; return rc; }
        if ((rc = ws.set_local_host(tx, ty, tz, N, false))) { (void)ws.sync(); lease.rc = rc; return rc; }
    }
    ws.set_global_sizes(0, 0);
    ws.set_allreduce(nullptr, nullptr);
    const double t1 = now_ms();
    rc = align_on(ws, init_T, p, out);
    out->ms_upload = t1 - t0;
    const int rcs = ws.sync();   // (nothing may still read the caller's buffers; cheap when the stream has drained)
    if (!rc) rc = rcs;
    if (rc) lease.rc = rc;  // a workspace that failed is dropped, not pooled
    return rc;
}

// Would a stand-alone align of this problem run the tiled matcher (sorted pairing)?  Only then is the batched path
// (k_nn_coop + batched accumulation) bit-identical to it: the dense kernels sum the pairing in another order.
// The shipped point-to-plane pipeline (params/icp-settings-loop-closure.yaml:23-39 -- what the reference's own nearby / loop-closure
// checks run, src/LidarOdometry.cpp:704-741, 767-788) always works on the tiled structures and its lists are exact whatever
// kernel computes them: batched = stand-alone at every size.  Its quality pass runs the point-to-point matcher, whose batched
// form counts the same pairings (an integer) as any of the stand-alone kernels.
bool batch_eligible(const mola_icp_params& p, size_t N, size_t M)
{
    if (N == 0 || M == 0) return false;
    if (p.n_extra_matchers != 0 || p.n_extra_solvers != 0 || p.n_extra_quality != 0) return false;   // staged pipelines: stand-alone aligns (stream per pair)
    // (the batched plane matcher is the cooperative kernel, one workgroup per 64 queries: it serves the sizes the stand-alone path
    //  gives to it -- up to ~131k queries; larger pairs go one by one through the persistent kernel, a stream per pair)
    // (knn 9 .. 16: the stand-alone path -- the batched launch is instantiated for the list lengths of knn 3 .. 8)
    if (p.matcher_class == MOLA_ICP_MATCHER_POINT2PLANE) return N <= (size_t)kBatchPlanesMaxQueries && p.knn <= 8;
    if (p.matcher_class != MOLA_ICP_MATCHER_POINTS_DISTANCE_THRESHOLD) return false;
    if (p.nn_kernel == MOLA_ICP_NN_TILED) return true;
    return p.nn_kernel == MOLA_ICP_NN_AUTO && N >= 8192 && M >= 8192;
}

// K prepared problems through the lockstep loop on one leased workspace; fills out[0..K)
int run_batch_on(HipWorkspace& ws, const std::vector<BatchProblem>& probs, const double* init_T, const mola_icp_params* p,
                 mola_icp_result* out)
{
    const size_t K = probs.size();
    HipBatch batch(ws, probs);
    int rc = batch.init();
    if (rc) return rc;
    std::vector<Mat4> inits(K);
    for (size_t k = 0; k < K; ++k) inits[k] = mat_from(init_T + 16 * k);
    for (size_t k = 0; k < K; ++k) std::memset(&out[k], 0, sizeof out[k]);
    if ((rc = run_icp_loop_batch(batch, inits.data(), *p, out))) return rc;
    double ms = 0;
    uint32_t n = 0;
    uint64_t pairs = 0;
    if ((rc = batch.collect_stats(&ms, &n, &pairs))) return rc;
    for (size_t k = 0; k < K; ++k) {  // the matcher launches served the whole batch: its totals, on every result
        out[k].ms_nn_kernel = ms;
        out[k].n_nn_launches = n;
        out[k].nn_kernel_used = MOLA_ICP_NN_TILED;
        out[k].nn_pairs_evaluated = pairs;
    }
    return MOLA_ICP_OK;
}

struct CallbackStages final : Stages {
    const mola_icp_stage_callbacks* cb;
    explicit CallbackStages(const mola_icp_stage_callbacks* c) : cb(c) {}
    int match(const Mat4& T, double thr, const mola_icp_params&, uint64_t* n) override
    {
        uint64_t dummy = 0;
        const int rc = cb->match(cb->user, T.m, thr, n ? n : &dummy);
        return rc ? fail(rc < 0 ? rc : MOLA_ICP_E_INTERNAL, "match callback failed") : MOLA_ICP_OK;
    }
    int accumulate(const mola_icp_params& p, const Mat4& Tcur, int stage, const double cl[3], const double cg[3],
                   bool reset, double acc[kNAcc]) override
    {
        const int rc = cb->accumulate(cb->user, &p, Tcur.m, stage, cl, cg, reset ? 1 : 0, acc);
        return rc ? fail(rc < 0 ? rc : MOLA_ICP_E_INTERNAL, "accumulate callback failed") : MOLA_ICP_OK;
    }
    int allreduce(double acc[kNAcc]) override
    {
        if (!cb->allreduce) return MOLA_ICP_OK;
        const int rc = cb->allreduce(acc, kNAcc, 0, cb->user);
        return rc ? fail(MOLA_ICP_E_COMM, "all-reduce hook failed with code " + std::to_string(rc)) : MOLA_ICP_OK;
    }
    uint64_t n_local_total() const override { return cb->n_local_total; }
    uint64_t n_map_total() const override { return cb->n_map_total; }
};

struct CallbackBatchStages final : BatchStages {
    const mola_icp_stage_callbacks* cb;
    size_t K;
    CallbackBatchStages(const mola_icp_stage_callbacks* c, size_t k) : cb(c), K(k) {}
    int size() const override { return (int)K; }
    int match(const uint8_t* active, const Mat4* T, double thr, const mola_icp_params&) override
    {
        for (size_t k = 0; k < K; ++k) {
            if (!active[k]) continue;
            uint64_t n = 0;
            const int rc = cb[k].match(cb[k].user, T[k].m, thr, &n);
            if (rc) return fail(rc < 0 ? rc : MOLA_ICP_E_INTERNAL, "match callback failed");
        }
        return MOLA_ICP_OK;
    }
    int accumulate(const uint8_t* active, const mola_icp_params& p, const Mat4* Tcur, int stage, const double (*cl)[3],
                   const double (*cg)[3], bool reset, double (*acc)[kNAcc]) override
    {
        for (size_t k = 0; k < K; ++k) {
            if (!active[k]) continue;
            const int rc = cb[k].accumulate(cb[k].user, &p, Tcur[k].m, stage, cl ? cl[k] : nullptr, cg ? cg[k] : nullptr,
                                            reset ? 1 : 0, acc[k]);
            if (rc) return fail(rc < 0 ? rc : MOLA_ICP_E_INTERNAL, "accumulate callback failed");
        }
        return MOLA_ICP_OK;
    }
    uint64_t n_local_total(int k) const override { return cb[k].n_local_total; }
    uint64_t n_map_total(int k) const override { return cb[k].n_map_total; }
};

}  // namespace

extern "C" {

int mola_icp_abi_version(void) { return MOLA_ICP_ABI_VERSION; }
const char* mola_icp_last_error(void) { return last_error(); }

const char* mola_icp_status_string(int s)
{
    switch (s) {
        case MOLA_ICP_OK: return "ok";
        case MOLA_ICP_E_BADARG: return "bad argument";
        case MOLA_ICP_E_CONFIG: return "configuration error";
        case MOLA_ICP_E_HIP: return "HIP runtime error";
        case MOLA_ICP_E_OOM: return "out of memory";
        case MOLA_ICP_E_NODEVICE: return "no usable gfx950 device";
        case MOLA_ICP_E_UNSUPPORTED: return "unsupported configuration";
        case MOLA_ICP_E_COMM: return "all-reduce failure";
        case MOLA_ICP_E_INTERNAL: return "internal error";
        default: return "unknown status";
    }
}

int mola_icp_device_count(int* count)
{
    return guarded([&]() -> int {
        if (!count) return fail(MOLA_ICP_E_BADARG, "null count");
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
        *count = n;
        return MOLA_ICP_OK;
    });
}

int mola_icp_set_thread_priority(int high)
{
    t_call_priority = high > 0 ? 1 : 0;
    return MOLA_ICP_OK;
}

int mola_icp_get_thread_priority(int* high)
{
    if (!high) return fail(MOLA_ICP_E_BADARG, "null argument");
    *high = t_call_priority;
    return MOLA_ICP_OK;
}

int mola_icp_set_wait_policy(int policy)
{
    if (policy < MOLA_ICP_WAIT_SPIN || policy > MOLA_ICP_WAIT_BLOCK) return fail(MOLA_ICP_E_BADARG, "wait policy must be MOLA_ICP_WAIT_SPIN, _YIELD or _BLOCK");
    set_wait_policy(policy);
    return MOLA_ICP_OK;
}

int mola_icp_get_wait_policy(int* policy)
{
    if (!policy) return fail(MOLA_ICP_E_BADARG, "null argument");
    *policy = wait_policy();
    return MOLA_ICP_OK;
}

int mola_icp_debug_reload_env(void)
{
    reload_env_knobs();
    return MOLA_ICP_OK;
}

int mola_icp_params_default(mola_icp_params* p)
{
    if (!p) return fail(MOLA_ICP_E_BADARG, "null params");
    params_default(*p);
    return MOLA_ICP_OK;
}

int mola_icp_params_from_yaml(const char* yaml_text, mola_icp_params* p)
{
    if (!yaml_text || !p) return fail(MOLA_ICP_E_BADARG, "null argument");
    try {
        YamlNode root = yaml_parse(yaml_text);
        params_from_yaml_node(root, *p);
        return MOLA_ICP_OK;
    } catch (const std::exception& e) {
        return fail(MOLA_ICP_E_CONFIG, e.what());
    }
}

int mola_icp_params_from_yaml_file(const char* path, const char* mola_dir, const char* key, mola_icp_params* p)
{
    if (!path || !p) return fail(MOLA_ICP_E_BADARG, "null argument");
    try {
        const std::string sp(path);
        YamlNode root = yaml_parse(read_text_file(sp));
        const size_t slash = sp.find_last_of('/');
        yaml_resolve_includes(root, slash == std::string::npos ? std::string(".") : sp.substr(0, slash),
                              mola_dir ? std::string(mola_dir) : std::string());
        const YamlNode* n = &root;
        if (key && *key) {
            // the reference nests the settings under `params:` in a full MOLA system file
            // (cfg = c["params"], src/LidarOdometry.cpp:102); accept both layouts
            if (!root.has(key) && root.has("params") && root.at("params").has(key)) n = &root.at("params");
            n = &n->at(key);
        }
        params_from_yaml_node(*n, *p);
        return MOLA_ICP_OK;
    } catch (const std::exception& e) {
        return fail(MOLA_ICP_E_CONFIG, e.what());
    }
}

int mola_icp_params_compose(const mola_icp_params* object_settings, const mola_icp_params* call_parameters,
                            mola_icp_params* out)
{
    if (!object_settings || !call_parameters || !out) return fail(MOLA_ICP_E_BADARG, "null argument");
    params_compose(*object_settings, *call_parameters, *out);
    return MOLA_ICP_OK;
}

int mola_icp_create(int device, mola_icp_handle** out)
{
    return guarded([&]() -> int {
        if (!out) return fail(MOLA_ICP_E_BADARG, "null out");
        *out = nullptr;
        std::unique_ptr<mola_icp_handle> h(new mola_icp_handle);
        h->resident.reset(new HipWorkspace(device));
        const int rc = h->resident->init();
        if (rc) return rc;
        h->device = h->resident->device();
        *out = h.release();
        return MOLA_ICP_OK;
    });
}

int mola_icp_destroy(mola_icp_handle* h)
{
    return guarded([&]() -> int {
        delete h;
        return MOLA_ICP_OK;
    });
}

int mola_icp_set_stream(mola_icp_handle* h, void* hip_stream)
{
    return guarded([&]() -> int {
        if (!h) return fail(MOLA_ICP_E_BADARG, "null handle");
        std::lock_guard<std::mutex> lk(h->mtx);
        return h->resident->set_external_stream(hip_stream);
    });
}

int mola_icp_set_profiling(mola_icp_handle* h, int on)
{
    if (!h) return fail(MOLA_ICP_E_BADARG, "null handle");
    std::lock_guard<std::mutex> lk(h->mtx);
    h->profiling = on ? 1 : 0;
    if (h->resident) h->resident->set_profiling(on != 0);
    return MOLA_ICP_OK;
}

int mola_icp_forget_warm_start(mola_icp_handle* h)
{
    if (!h) return fail(MOLA_ICP_E_BADARG, "null handle");
    std::lock_guard<std::mutex> lk(h->mtx);
    if (h->resident) h->resident->forget_warm_start(false);
    return MOLA_ICP_OK;
}

int mola_icp_forget_cloud_schedule(mola_icp_handle* h)
{
    if (!h) return fail(MOLA_ICP_E_BADARG, "null handle");
    std::lock_guard<std::mutex> lk(h->mtx);
    if (h->resident) h->resident->forget_warm_start(true);
    return MOLA_ICP_OK;
}

int mola_icp_set_allreduce(mola_icp_handle* h, mola_icp_allreduce_fn fn, void* user)
{
    if (!h) return fail(MOLA_ICP_E_BADARG, "null handle");
    std::lock_guard<std::mutex> lk(h->mtx);
    h->ar_fn = fn;
    h->ar_user = user;
    h->resident->set_allreduce(fn, user);
    return MOLA_ICP_OK;
}

int mola_icp_comm_set_library(const char* path) { return rccl_set_library(path); }

int mola_icp_comm_unique_id(uint8_t id_out[128])
{
    return guarded([&]() -> int {
        if (!id_out) return fail(MOLA_ICP_E_BADARG, "null id");
        RcclUniqueId id;
        const int rc = rccl_unique_id(&id);
        if (rc) return rc;
        std::memcpy(id_out, id.internal, 128);
        return MOLA_ICP_OK;
    });
}

int mola_icp_comm_init(mola_icp_handle* h, const uint8_t id[128], int nranks, int rank)
{
    return guarded([&]() -> int {
        if (!h || !id) return fail(MOLA_ICP_E_BADARG, "null argument");
        if (nranks < 1 || rank < 0 || rank >= nranks) return fail(MOLA_ICP_E_BADARG, "bad rank / nranks");
        std::lock_guard<std::mutex> lk(h->mtx);
        if (h->comm) return fail(MOLA_ICP_E_BADARG, "communicator already initialised");
        if (hipSetDevice(h->device) != hipSuccess) return fail(MOLA_ICP_E_HIP, "hipSetDevice failed");
        RcclUniqueId uid;
        std::memcpy(uid.internal, id, 128);
        const int rc = rccl_comm_init(&h->comm, nranks, uid, rank);
        if (rc) return rc;
        h->resident->set_comm(h->comm);
        return MOLA_ICP_OK;
    });
}

int mola_icp_comm_attach_local(mola_icp_handle* h, mola_icp_local_comm* c)
{
    return guarded([&]() -> int {
        if (!h) return fail(MOLA_ICP_E_BADARG, "null handle");
        std::lock_guard<std::mutex> lk(h->mtx);
        if (h->comm) return fail(MOLA_ICP_E_BADARG, "the handle has an RCCL communicator: destroy it first");
        h->resident->sync();
        h->local_comm = c;
        if (c) h->resident->set_allreduce(&local_comm_hook, c);
        else h->resident->set_allreduce(nullptr, nullptr);
        return MOLA_ICP_OK;
    });
}

int mola_icp_comm_nranks(mola_icp_handle* h, int* nranks_out)
{
    return guarded([&]() -> int {
        if (!h || !nranks_out) return fail(MOLA_ICP_E_BADARG, "null argument");
        std::lock_guard<std::mutex> lk(h->mtx);
        if (!h->comm && h->local_comm) return mola_icp_local_comm_nranks(h->local_comm, nranks_out);
        if (!h->comm) return fail(MOLA_ICP_E_BADARG, "no communicator: call mola_icp_comm_init first");
        return rccl_comm_count(h->comm, nranks_out);
    });
}

int mola_icp_comm_destroy(mola_icp_handle* h)
{
    return guarded([&]() -> int {
        if (!h) return fail(MOLA_ICP_E_BADARG, "null handle");
        std::lock_guard<std::mutex> lk(h->mtx);
        h->resident->sync();
        h->resident->set_comm(nullptr);
        if (h->local_comm) {  // (not owned: the caller destroys it with mola_icp_local_comm_destroy)
            h->local_comm = nullptr;
            h->resident->set_allreduce(nullptr, nullptr);
        }
        const int rc = h->comm ? rccl_comm_destroy(h->comm) : MOLA_ICP_OK;
        h->comm = nullptr;
        return rc;
    });
}

int mola_icp_align(mola_icp_handle* h, const float* fx, const float* fy, const float* fz, size_t M, const float* tx,
                   const float* ty, const float* tz, size_t N, const double init_T[16], const mola_icp_params* p,
                   mola_icp_result* out)
{
    return guarded([&]() -> int {
        if (!h || !p || !out) return fail(MOLA_ICP_E_BADARG, "null argument");
        return align_host_clouds(h, fx, fy, fz, M, tx, ty, tz, N, init_T, p, out);
    });
}

int mola_icp_align_batch(mola_icp_handle* h, size_t n_pairs, const float* const* fx, const float* const* fy,
                         const float* const* fz, const size_t* M, const float* const* tx, const float* const* ty,
                         const float* const* tz, const size_t* N, const double* init_T, const mola_icp_params* p,
                         mola_icp_result* out)
{
    return guarded([&]() -> int {
        if (!h || !p || !out || !init_T) return fail(MOLA_ICP_E_BADARG, "null argument");
        if (n_pairs && (!fx || !fy || !fz || !M || !tx || !ty || !tz || !N))
            return fail(MOLA_ICP_E_BADARG, "null batch array");
        int rc;
        if ((rc = validate_params(*p))) return rc;
        for (size_t i = 0; i < n_pairs; ++i)
            if ((rc = check_pose(init_T + 16 * i))) return rc;
        // Pairs whose stand-alone align would run the tiled matcher advance TOGETHER, kCoopMaxBatch at a time: every
        // stage is one launch over the pairs still iterating (blockIdx.y = pair), so a launch that is "one item long"
        // for a 100k-point cloud serves a dozen of them.  The others (tiny clouds, the point-to-plane pipeline) go to
        // the handle's worker threads, stream-per-pair (worker_pool_past_KFs_, src/LidarOdometry.cpp:94-96).
        std::vector<size_t> together, apart;
        for (size_t i = 0; i < n_pairs; ++i) (batch_eligible(*p, N[i], M[i]) ? together : apart).push_back(i);
        if (together.size() == 1) { apart.push_back(together[0]); together.clear(); }
        std::atomic<int> first_err{MOLA_ICP_OK};
        std::string err_msg;
        std::mutex err_mtx;
        auto record = [&](int rc2, size_t i) {
            std::lock_guard<std::mutex> lk(err_mtx);
            if (first_err.load() == MOLA_ICP_OK) {
                first_err = rc2;
                err_msg = "pair " + std::to_string(i) + ": " + last_error();
            }
        };
        // the stream-per-pair jobs run on the pool while this thread drives the lockstep chunks
        std::mutex done_mtx;
        std::condition_variable done_cv;
        size_t pending = 0;   // jobs on the pool that reference this frame
        // (declared before the first submit: whatever throws below, the frame outlives its jobs)
        auto wait_jobs = at_scope_exit([&]() {
            std::unique_lock<std::mutex> lk(done_mtx);
            done_cv.wait(lk, [&]() { return pending == 0; });
        });
        auto submit_counted = [&](std::function<void()> job) {   // the job must end with job_done()
            { std::lock_guard<std::mutex> lk(done_mtx); ++pending; }
            try {
                h->submit(std::move(job));
            } catch (...) {
                { std::lock_guard<std::mutex> lk(done_mtx); --pending; }
                throw;
            }
        };
        // (the notification is sent UNDER the lock: the waiter owns this frame -- released first, it could see pending == 0, return and
        // unwind mutex and condition variable while this worker was still inside notify_all() on them.  Found by ThreadSanitizer through
        // tests/hosts/race_host.cpp, round 5; with the lock held the waiter cannot leave wait() before the worker is done with both.)
        auto job_done = [&]() {
            std::lock_guard<std::mutex> lk(done_mtx);
            --pending;
            done_cv.notify_all();
        };
        if (!apart.empty()) {
            h->ensure_workers(apart.size() < 8 ? apart.size() : 8);
            for (size_t i : apart)
                submit_counted([&, i]() {
                    const int rc2 = guarded([&]() -> int {
                        return align_host_clouds(h, fx[i], fy[i], fz[i], M[i], tx[i], ty[i], tz[i], N[i], init_T + 16 * i, p,
                                                 &out[i]);
                    });
                    if (rc2) record(rc2, i);
                    job_done();
                });
        }
        // Lockstep chunks, software-pipelined: while chunk c iterates on this thread, chunk c+1's clouds are uploaded and
        // prepared (Hilbert sort, box levels) by a worker on another workspace / stream.
        struct Prepared {
            std::vector<BatchProblem> probs;
            std::vector<double> inits;
            double upload_ms = 0;
            int rc = MOLA_ICP_OK;
            std::string err;
        };
        // balanced chunks (16 pairs = 8 + 8, not 12 + 4: a lockstep launch over 4 problems is nearly as long as one over 12)
        std::vector<size_t> chunk_begin;
        {
            const size_t n_chunks = (together.size() + kBatchChunk - 1) / kBatchChunk;
            for (size_t c = 0; c <= n_chunks; ++c) chunk_begin.push_back(n_chunks ? c * together.size() / n_chunks : 0);
        }
        auto prepare = [&](size_t ci) -> std::shared_ptr<Prepared> {
            auto pr = std::make_shared<Prepared>();
            const size_t c0 = chunk_begin[ci], K = chunk_begin[ci + 1] - c0;
            pr->rc = guarded([&]() -> int {
                Lease lease(h);
                if (lease.rc) return lease.rc;
                const double t0 = now_ms();
                pr->probs.resize(K);
                pr->inits.resize(16 * K);
                int rc3;
                for (size_t k = 0; k < K; ++k) {
                    const size_t i = together[c0 + k];
                    pr->probs[k].map = h->take_cloud();
                    pr->probs[k].loc = h->take_cloud();
                    if ((rc3 = lease.ws->build_cached(*pr->probs[k].map, fx[i], fy[i], fz[i], M[i]))) { lease.rc = rc3; return rc3; }
                    if ((rc3 = lease.ws->build_cached(*pr->probs[k].loc, tx[i], ty[i], tz[i], N[i]))) { lease.rc = rc3; return rc3; }
                    std::memcpy(&pr->inits[16 * k], init_T + 16 * i, sizeof(double) * 16);
                }
                pr->upload_ms = now_ms() - t0;
                return MOLA_ICP_OK;
            });
            if (pr->rc) pr->err = last_error();
            return pr;
        };
        // TWO lockstep loops side by side (chunks 0, 2, 4, ... and 1, 3, 5, ...), each with its own prepare-ahead: while one
        // loop's host thread runs its dozen Horn solves and stall tests (the GPU would idle ~40 us of every ~185-us lockstep
        // iteration), the other loop's launches run.  The persistent matcher has no inter-block dependencies, so two of them
        // sharing the device is safe; each is a little slower, the pair is faster.
        const size_t n_chunks = chunk_begin.size() - 1;
        // A lane never lets an exception out (it may run on a thread of its own) and never leaves a prepare-ahead task
        // behind: the task references this frame, so its future is always collected -- also when the loop body throws.
        auto lane_body = [&](size_t first) {
            if (first >= n_chunks) return;
            std::shared_ptr<Prepared> cur = prepare(first);
            for (size_t ci = first; ci < n_chunks; ci += 2) {
                const size_t c0 = chunk_begin[ci], K = chunk_begin[ci + 1] - c0;
                std::future<std::shared_ptr<Prepared>> next;
                auto collect = at_scope_exit([&]() { if (next.valid()) next.wait(); });
                if (ci + 2 < n_chunks && first_err.load() == MOLA_ICP_OK) {
                    auto task = std::make_shared<std::packaged_task<std::shared_ptr<Prepared>()>>([&prepare, ci]() { return prepare(ci + 2); });
                    next = task->get_future();
                    // the prepare-ahead runs on a pool worker; it waits for nothing on the pool (no job of this library
                    // blocks on another pool job: lane 1 below has a thread of its own), so it always gets its turn
                    try { h->submit([task]() { (*task)(); }); } catch (...) { (*task)(); }
                }
                if (cur->rc) {
                    set_error(cur->err);
                    record(cur->rc, together[c0]);
                } else if (first_err.load() == MOLA_ICP_OK) {
                    const int rc2 = guarded([&]() -> int {
                        Lease lease(h);
                        if (lease.rc) return lease.rc;
                        std::vector<mola_icp_result> res(K);
                        int rc3;
                        if ((rc3 = run_batch_on(*lease.ws, cur->probs, cur->inits.data(), p, res.data()))) { lease.rc = rc3; return rc3; }
                        for (size_t k = 0; k < K; ++k) {
                            res[k].ms_upload = cur->upload_ms;  // the chunk's uploads + preparation (overlapped with the previous chunk's loop)
                            out[together[c0 + k]] = res[k];
                        }
                        return MOLA_ICP_OK;
                    });
                    if (rc2) record(rc2, together[c0]);
                }
                for (BatchProblem& bp : cur->probs) { h->give_cloud(std::move(bp.map)); h->give_cloud(std::move(bp.loc)); }
                cur = next.valid() ? next.get() : nullptr;   // (always collected: the worker references this frame)
                if (!cur) break;
            }
            if (cur) for (BatchProblem& bp : cur->probs) { h->give_cloud(std::move(bp.map)); h->give_cloud(std::move(bp.loc)); }
        };
        auto lane = [&](size_t first) {
            const int rc2 = guarded([&]() -> int { lane_body(first); return MOLA_ICP_OK; });
            if (rc2) record(rc2, first < n_chunks ? together[chunk_begin[first]] : 0);
        };
        if (n_chunks >= 1) {
            h->ensure_workers(n_chunks >= 2 ? 2 : 1);   // a prepare-ahead per loop (+ the stream-per-pair jobs above)
            // The second loop runs on a thread of ITS OWN, not on the pool: it blocks on prepare-ahead tasks that sit in the
            // pool's queue -- as a pool job it could occupy the very worker its task needs (several concurrent batch calls
            // on one handle filled every worker with a waiting lane: nothing ever ran the tasks).
            std::thread lane1;
            auto join_lane1 = at_scope_exit([&]() { if (lane1.joinable()) lane1.join(); });
            bool lane1_inline = false;
            if (n_chunks >= 2) {
                try { lane1 = std::thread([&]() { lane(1); }); } catch (const std::system_error&) { lane1_inline = true; }
            }
            lane(0);
            if (lane1_inline) lane(1);
        }   // (lane 1 joined here)
        {
            std::unique_lock<std::mutex> lk(done_mtx);
            done_cv.wait(lk, [&]() { return pending == 0; });
        }
        if (first_err.load()) return fail(first_err.load(), err_msg);
        return MOLA_ICP_OK;
    });
}

int mola_icp_pool_assignment(size_t n_pairs, int n_devices, int* device_of_pair)
{
    if (n_devices < 1 || (n_pairs && !device_of_pair)) return fail(MOLA_ICP_E_BADARG, "bad pool assignment arguments");
    for (size_t i = 0; i < n_pairs; ++i) device_of_pair[i] = (int)(i % (size_t)n_devices);  // round-robin
    return MOLA_ICP_OK;
}

int mola_icp_pool_create(const int* devices, int n_devices, mola_icp_pool** out)
{
    return guarded([&]() -> int {
        if (!out) return fail(MOLA_ICP_E_BADARG, "null out");
        *out = nullptr;
        std::vector<int> devs;
        if (!devices || n_devices <= 0) {
            int n = 0;
            if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
                return fail(MOLA_ICP_E_NODEVICE, "no HIP device available; this library has no CPU fallback");
            for (int d = 0; d < n; ++d) devs.push_back(d);
        } else {
            devs.assign(devices, devices + n_devices);
        }
        std::unique_ptr<mola_icp_pool> pool(new mola_icp_pool);
        for (int d : devs) {
            if (d < 0) return fail(MOLA_ICP_E_BADARG, "negative device index in a pool");
            mola_icp_handle* h = nullptr;
            const int rc = mola_icp_create(d, &h);
            if (rc) return rc;
            pool->handles.push_back(h);
        }
        pool->start();
        *out = pool.release();
        return MOLA_ICP_OK;
    });
}

int mola_icp_pool_destroy(mola_icp_pool* pool)
{
    return guarded([&]() -> int {
        delete pool;
        return MOLA_ICP_OK;
    });
}

int mola_icp_pool_size(const mola_icp_pool* pool, int* n_handles)
{
    if (!pool || !n_handles) return fail(MOLA_ICP_E_BADARG, "null argument");
    *n_handles = (int)pool->handles.size();
    return MOLA_ICP_OK;
}

int mola_icp_pool_handle(mola_icp_pool* pool, int i, mola_icp_handle** h)
{
    if (!pool || !h || i < 0 || i >= (int)pool->handles.size()) return fail(MOLA_ICP_E_BADARG, "bad pool handle index");
    *h = pool->handles[(size_t)i];
    return MOLA_ICP_OK;
}

int mola_icp_pool_align_batch(mola_icp_pool* pool, size_t n_pairs, const float* const* fx, const float* const* fy,
                              const float* const* fz, const size_t* M, const float* const* tx, const float* const* ty,
                              const float* const* tz, const size_t* N, const double* init_T, const mola_icp_params* p,
                              mola_icp_result* out)
{
    return guarded([&]() -> int {
        if (!pool || pool->handles.empty() || !p || !out || !init_T) return fail(MOLA_ICP_E_BADARG, "null argument");
        if (n_pairs && (!fx || !fy || !fz || !M || !tx || !ty || !tz || !N))
            return fail(MOLA_ICP_E_BADARG, "null batch array");
        const int nd = (int)pool->handles.size();
        if (n_pairs == 0) return MOLA_ICP_OK;
        std::lock_guard<std::mutex> call(pool->call_m);
        // chunks: half a slot's fair share (a slot takes two on average, more if the others' pairs run long), at most what one
        // lockstep launch advances together
        size_t chunk = (n_pairs + (size_t)(2 * nd) - 1) / (size_t)(2 * nd);
        if (chunk > (size_t)kBatchChunk) chunk = (size_t)kBatchChunk;
        if (chunk < 1) chunk = 1;
        {
            std::unique_lock<std::mutex> lk(pool->m);
            mola_icp_pool::Job& j = pool->job;
            j.n_pairs = n_pairs; j.chunk = chunk;
            j.fx = fx; j.fy = fy; j.fz = fz; j.M = M; j.tx = tx; j.ty = ty; j.tz = tz; j.N = N;
            j.init_T = init_T; j.p = p; j.out = out;
            pool->cursor = 0;
            pool->busy = nd;
            ++pool->generation;
            pool->cv_work.notify_all();
            pool->cv_done.wait(lk, [&] { return pool->busy == 0; });
            for (int d = 0; d < nd; ++d)
                if (pool->rcs[(size_t)d]) return fail(pool->rcs[(size_t)d], "device slot " + std::to_string(d) + ": " + pool->msgs[(size_t)d]);
        }
        return MOLA_ICP_OK;
    });
}

int mola_icp_pool_last_shares(const mola_icp_pool* pool, size_t* pairs_per_slot, int n_slots)
{
    if (!pool || !pairs_per_slot || n_slots < (int)pool->handles.size()) return fail(MOLA_ICP_E_BADARG, "bad argument");
    std::lock_guard<std::mutex> lk(const_cast<mola_icp_pool*>(pool)->m);
    for (size_t d = 0; d < pool->handles.size(); ++d) pairs_per_slot[d] = pool->served[d];
    return MOLA_ICP_OK;
}

int mola_icp_align_multi_init(mola_icp_handle* h, const float* fx, const float* fy, const float* fz, size_t M,
                              const float* tx, const float* ty, const float* tz, size_t N, size_t n_init,
                              const double* init_T, const mola_icp_params* p, mola_icp_result* out,
                              mola_icp_result* best, int* best_index)
{
    return guarded([&]() -> int {
        if (!h || !p || !init_T || !best_index) return fail(MOLA_ICP_E_BADARG, "null argument");
        int rc;
        for (size_t k = 0; k < n_init; ++k)
            if ((rc = check_pose(init_T + 16 * k))) return rc;
        if ((rc = validate_params(*p))) return rc;
        Lease lease(h);
        if (lease.rc) return lease.rc;
        HipWorkspace& ws = *lease.ws;
        const double t0 = now_ms();
        *best_index = -1;
        if (n_init >= 2 && batch_eligible(*p, N, M)) {
            // the K guesses as a batch dimension on the device: the pair is uploaded and prepared once, every stage of
            // every iteration is ONE launch over the guesses still iterating (blockIdx.y = guess)
            auto map = h->take_cloud(), loc = h->take_cloud();
            if ((rc = ws.build_cached(*map, fx, fy, fz, M))) { lease.rc = rc; return rc; }
            if ((rc = ws.build_cached(*loc, tx, ty, tz, N))) { lease.rc = rc; return rc; }
            const double upload_ms = now_ms() - t0;
            std::vector<mola_icp_result> res(n_init);
            {
                std::vector<BatchProblem> probs(n_init);
                for (auto& pr : probs) { pr.map = map; pr.loc = loc; }
                rc = run_batch_on(ws, probs, init_T, p, res.data());
            }
            h->give_cloud(std::move(map));
            h->give_cloud(std::move(loc));
            if (rc) { lease.rc = rc; return rc; }
            double best_q = 0.0;  // ICP_Output::goodness starts at .0 (LidarOdometry.h:130); strictly greater wins (cpp:785)
            for (size_t k = 0; k < n_init; ++k) {
                res[k].ms_upload = k == 0 ? upload_ms : 0.0;
                if (out) out[k] = res[k];
                if (res[k].quality > best_q) {
                    best_q = res[k].quality;
                    *best_index = (int)k;
                    if (best) *best = res[k];
                }
            }
            return MOLA_ICP_OK;
        }
        // Other pipelines (the shipped point-to-plane settings, tiny clouds): the pair is still uploaded and prepared ONCE;
        // the attempts then run side by side on the handle's worker threads, a workspace and stream each, sharing the two
        // prepared clouds (as concurrent mola_icp_align_cached calls would) -- at odometry sizes a launch leaves most of
        // the GPU idle, so the streams overlap.  Each attempt is the stand-alone align, bit for bit.
        auto map = h->take_cloud(), loc = h->take_cloud();
        if ((rc = ws.build_cached(*map, fx, fy, fz, M))) { lease.rc = rc; return rc; }
        if ((rc = ws.build_cached(*loc, tx, ty, tz, N))) { lease.rc = rc; return rc; }
        const double upload_ms = now_ms() - t0;
        std::vector<mola_icp_result> res(n_init);
        std::vector<int> rcs(n_init, MOLA_ICP_OK);
        std::vector<std::string> errs(n_init);
        auto attempt = [&](HipWorkspace& w, size_t k) {
            std::memset(&res[k], 0, sizeof res[k]);
            w.use_cached_map(map);
            w.use_cached_local(loc);
            w.set_global_sizes(0, 0);
            w.set_allreduce(nullptr, nullptr);
            rcs[k] = align_on(w, init_T + 16 * k, p, &res[k]);
            if (rcs[k]) errs[k] = last_error();
        };
        if (n_init == 1) {
            attempt(ws, 0);
            if (rcs[0]) lease.rc = rcs[0];
        } else {
            std::mutex dm;
            std::condition_variable dcv;
            size_t pending = 0;   // jobs on the pool that reference this frame: it outlives them whatever throws below
            auto wait_jobs = at_scope_exit([&]() {
                std::unique_lock<std::mutex> lk(dm);
                dcv.wait(lk, [&]() { return pending == 0; });
            });
            h->ensure_workers(std::min<size_t>(n_init - 1, 7));
            for (size_t k = 1; k < n_init; ++k) {
                { std::lock_guard<std::mutex> lk(dm); ++pending; }
                try {
                h->submit([&, k]() {
                    rcs[k] = guarded([&]() -> int {
                        Lease l2(h);
                        if (l2.rc) return l2.rc;
                        attempt(*l2.ws, k);
                        l2.ws->use_cached_map(std::make_shared<SortedCloud>());   // (no reference kept in the pooled workspace)
                        l2.ws->use_cached_local(std::make_shared<SortedCloud>());
                        if (rcs[k]) l2.rc = rcs[k];
                        return rcs[k];
                    });
                    if (rcs[k] && errs[k].empty()) errs[k] = last_error();
                    {   // (notified under the lock: see mola_icp_align_batch's job_done)
                        std::lock_guard<std::mutex> lk(dm);
                        --pending;
                        dcv.notify_all();
                    }
                });
                } catch (...) {
                    { std::lock_guard<std::mutex> lk(dm); --pending; }
                    throw;
                }
            }
            attempt(ws, 0);   // this thread takes the first attempt
            if (rcs[0]) lease.rc = rcs[0];
            std::unique_lock<std::mutex> lk(dm);
            dcv.wait(lk, [&]() { return pending == 0; });
        }
        ws.use_cached_map(std::make_shared<SortedCloud>());    // (drop this workspace's references before recycling the clouds)
        ws.use_cached_local(std::make_shared<SortedCloud>());
        for (size_t k = 0; k < n_init; ++k)
            if (rcs[k]) { set_error(errs[k]); return rcs[k]; }
        h->give_cloud(std::move(map));
        h->give_cloud(std::move(loc));
        double best_q = 0.0;  // ICP_Output::goodness starts at .0 (LidarOdometry.h:130)
        for (size_t k = 0; k < n_init; ++k) {
            res[k].ms_upload = k == 0 ? upload_ms : 0.0;
            if (out) out[k] = res[k];
            if (res[k].quality > best_q) {
                best_q = res[k].quality;
                *best_index = (int)k;
                if (best) *best = res[k];
            }
        }
        return MOLA_ICP_OK;
    });
}

int mola_icp_cloud_put(mola_icp_handle* h, uint64_t id, const float* x, const float* y, const float* z, size_t n)
{
    return guarded([&]() -> int {
        if (!h) return fail(MOLA_ICP_E_BADARG, "null handle");
        Lease lease(h);
        if (lease.rc) return lease.rc;
        auto sc = std::make_shared<SortedCloud>();
        const int rc = lease.ws->build_cached(*sc, x, y, z, n);
        if (rc) { lease.rc = rc; return rc; }
        std::lock_guard<std::mutex> lk(h->cache_mtx);
        h->cache[id] = std::move(sc);
        return MOLA_ICP_OK;
    });
}

int mola_icp_cloud_drop(mola_icp_handle* h, uint64_t id)
{
    return guarded([&]() -> int {
        if (!h) return fail(MOLA_ICP_E_BADARG, "null handle");
        std::lock_guard<std::mutex> lk(h->cache_mtx);
        if (!h->cache.erase(id)) return fail(MOLA_ICP_E_BADARG, "no cached cloud with id " + std::to_string(id));
        return MOLA_ICP_OK;
    });
}

int mola_icp_device_pool_trim(int device, size_t keep_bytes, size_t* parked_bytes_out)
{
    return guarded([&]() -> int {
        if (device < -1 || device >= 16) return fail(MOLA_ICP_E_BADARG, "device index out of range (-1 = every device)");
        device_pool_trim(keep_bytes, device);   // (only that device's parked blocks: hipFree synchronises the device it frees on)
        if (parked_bytes_out) *parked_bytes_out = device >= 0 ? device_pool_bytes(device) : 0;
        return MOLA_ICP_OK;
    });
}

int mola_icp_cloud_count(mola_icp_handle* h, size_t* count_out, size_t* device_bytes_out)
{
    return guarded([&]() -> int {
        if (!h) return fail(MOLA_ICP_E_BADARG, "null handle");
        std::lock_guard<std::mutex> lk(h->cache_mtx);
        size_t bytes = 0;
        for (auto& kv : h->cache) {
            const SortedCloud& c = *kv.second;
            bytes += c.raw.cap + c.sorted.cap + c.perm.cap + c.tbox.cap + c.sbox.cap + c.ubox.cap;
        }
        if (count_out) *count_out = h->cache.size();
        if (device_bytes_out) *device_bytes_out = bytes;
        return MOLA_ICP_OK;
    });
}

int mola_icp_align_cached(mola_icp_handle* h, uint64_t from_id, uint64_t to_id, const double init_T[16],
                          const mola_icp_params* p, mola_icp_result* out)
{
    return guarded([&]() -> int {
        if (!h || !p || !out) return fail(MOLA_ICP_E_BADARG, "null argument");
        int rc;
        if ((rc = check_pose(init_T))) return rc;
        if ((rc = validate_params(*p))) return rc;
        std::shared_ptr<SortedCloud> from, to;
        {
            std::lock_guard<std::mutex> lk(h->cache_mtx);
            auto a = h->cache.find(from_id), b = h->cache.find(to_id);
            if (a == h->cache.end()) return fail(MOLA_ICP_E_BADARG, "no cached cloud with id " + std::to_string(from_id));
            if (b == h->cache.end()) return fail(MOLA_ICP_E_BADARG, "no cached cloud with id " + std::to_string(to_id));
            from = a->second;
            to = b->second;
        }
        Lease lease(h);
        if (lease.rc) return lease.rc;
        HipWorkspace& ws = *lease.ws;
        std::memset(out, 0, sizeof *out);
        ws.use_cached_map(from);
        ws.use_cached_local(to);
        ws.set_global_sizes(0, 0);
        ws.set_allreduce(nullptr, nullptr);
        rc = align_on(ws, init_T, p, out);
        if (rc) lease.rc = rc;
        return rc;
    });
}

int mola_icp_align_cached_put(mola_icp_handle* h, uint64_t from_id, uint64_t to_id, const float* tx, const float* ty, const float* tz,
                              size_t N, const double init_T[16], const mola_icp_params* p, mola_icp_result* out, int* put_done)
{
    if (put_done) *put_done = 0;
    return guarded([&]() -> int {
        if (!h || !p || !out) return fail(MOLA_ICP_E_BADARG, "null argument");
        int rc;
        if ((rc = check_pose(init_T))) return rc;
        if ((rc = validate_params(*p))) return rc;
        std::shared_ptr<SortedCloud> from;
        {
            std::lock_guard<std::mutex> lk(h->cache_mtx);
            auto a = h->cache.find(from_id);
            if (a == h->cache.end()) return fail(MOLA_ICP_E_BADARG, "no cached cloud with id " + std::to_string(from_id));
            from = a->second;
        }
        Lease lease(h);
        if (lease.rc) return lease.rc;
        HipWorkspace& ws = *lease.ws;
        std::memset(out, 0, sizeof *out);
        const double t0 = now_ms();
        // the new cloud's prepare chain is only enqueued: the align's first launches follow it down the same stream, the device
        // never idles between the two (a host wait there cost 16-24 us of an odometry step's ~500) -- and nobody else can see the
        // cloud before it is finished: it enters the cache at the end
        auto to = std::make_shared<SortedCloud>();
        if ((rc = ws.build_cached(to, tx, ty, tz, N, false))) { lease.rc = rc; return rc; }
        const double upload_ms = now_ms() - t0;
        ws.use_cached_map(from);
        ws.use_cached_local(to);
        ws.set_global_sizes(0, 0);
        ws.set_allreduce(nullptr, nullptr);
        rc = align_on(ws, init_T, p, out);
        out->ms_upload = upload_ms;
        const int rc2 = ws.finish_build(*to);   // (whatever the align did: nothing of the build in flight, the box looked at)
        if (!rc2 && (to->ready || N == 0)) {   // (an empty cloud is cached as mola_icp_cloud_put caches it)
            std::lock_guard<std::mutex> lk(h->cache_mtx);
            h->cache[to_id] = to;
            if (put_done) *put_done = 1;
        } else {
            ws.use_cached_local(std::make_shared<SortedCloud>());   // (no reference to a refused cloud stays in the pooled workspace)
        }
        if (rc || rc2) lease.rc = rc ? rc : rc2;
        return rc ? rc : rc2;
    });
}

int mola_icp_voxel_downsample(mola_icp_handle* h, const float* x, const float* y, const float* z, size_t n,
                              double voxel_size, float* out_x, float* out_y, float* out_z, size_t capacity, size_t* n_out)
{
    return guarded([&]() -> int {
        if (!h) return fail(MOLA_ICP_E_BADARG, "null handle");
        Lease lease(h);
        if (lease.rc) return lease.rc;
        const int rc = lease.ws->voxel_downsample(x, y, z, n, voxel_size, out_x, out_y, out_z, capacity, n_out);
        if (rc) lease.rc = rc;
        return rc;
    });
}

#define RESIDENT_CALL(expr)                                             \
    return guarded([&]() -> int {                                       \
        if (!h) return fail(MOLA_ICP_E_BADARG, "null handle");          \
        std::lock_guard<std::mutex> lk(h->mtx);                         \
        HipWorkspace& ws = *h->resident;                                \
        (void)ws;                                                       \
        return (expr);                                                  \
    })

int mola_icp_set_map_host(mola_icp_handle* h, const float* x, const float* y, const float* z, size_t M)
{
    RESIDENT_CALL(ws.set_map_host(x, y, z, M));
}
int mola_icp_set_map_device(mola_icp_handle* h, const float* x, const float* y, const float* z, size_t M)
{
    RESIDENT_CALL(ws.set_map_device(x, y, z, M));
}
int mola_icp_set_local_host(mola_icp_handle* h, const float* x, const float* y, const float* z, size_t N)
{
    RESIDENT_CALL(ws.set_local_host(x, y, z, N));
}
int mola_icp_set_local_device(mola_icp_handle* h, const float* x, const float* y, const float* z, size_t N)
{
    RESIDENT_CALL(ws.set_local_device(x, y, z, N));
}
int mola_icp_set_global_sizes(mola_icp_handle* h, uint64_t nl, uint64_t nm)
{
    RESIDENT_CALL((ws.set_global_sizes(nl, nm), MOLA_ICP_OK));
}

int mola_icp_set_local_shard_host(mola_icp_handle* h, const float* x, const float* y, const float* z, size_t n_total, int rank,
                                  int nranks, size_t* n_shard_out)
{
    return guarded([&]() -> int {
        if (!h) return fail(MOLA_ICP_E_BADARG, "null handle");
        std::lock_guard<std::mutex> lk(h->mtx);
        const int rc = h->resident->set_local_shard(x, y, z, n_total, rank, nranks, false);
        if (!rc && n_shard_out) *n_shard_out = h->resident->N();
        return rc;
    });
}

int mola_icp_set_local_shard_device(mola_icp_handle* h, const float* dx, const float* dy, const float* dz, size_t n_total, int rank,
                                    int nranks, size_t* n_shard_out)
{
    return guarded([&]() -> int {
        if (!h) return fail(MOLA_ICP_E_BADARG, "null handle");
        std::lock_guard<std::mutex> lk(h->mtx);
        const int rc = h->resident->set_local_shard(dx, dy, dz, n_total, rank, nranks, true);
        if (!rc && n_shard_out) *n_shard_out = h->resident->N();
        return rc;
    });
}

int mola_icp_set_local_shard_range_host(mola_icp_handle* h, const float* x, const float* y, const float* z, size_t n_total, size_t lo,
                                        size_t hi, size_t* n_shard_out)
{
    return guarded([&]() -> int {
        if (!h) return fail(MOLA_ICP_E_BADARG, "null handle");
        std::lock_guard<std::mutex> lk(h->mtx);
        const int rc = h->resident->set_local_shard_range(x, y, z, n_total, lo, hi, false);
        if (!rc && n_shard_out) *n_shard_out = h->resident->N();
        return rc;
    });
}

int mola_icp_set_local_shard_range_device(mola_icp_handle* h, const float* dx, const float* dy, const float* dz, size_t n_total, size_t lo,
                                          size_t hi, size_t* n_shard_out)
{
    return guarded([&]() -> int {
        if (!h) return fail(MOLA_ICP_E_BADARG, "null handle");
        std::lock_guard<std::mutex> lk(h->mtx);
        const int rc = h->resident->set_local_shard_range(dx, dy, dz, n_total, lo, hi, true);
        if (!rc && n_shard_out) *n_shard_out = h->resident->N();
        return rc;
    });
}

int mola_icp_local_shard_indices(mola_icp_handle* h, int32_t* idx_out)
{
    return guarded([&]() -> int {
        if (!h) return fail(MOLA_ICP_E_BADARG, "null handle");
        std::lock_guard<std::mutex> lk(h->mtx);
        return h->resident->copy_shard_indices(idx_out);
    });
}

int mola_icp_shard_reach_box(mola_icp_handle* h, const double T[16], double margin, double lo_out[3], double hi_out[3])
{
    return guarded([&]() -> int {
        if (!h || !T || !lo_out || !hi_out) return fail(MOLA_ICP_E_BADARG, "null argument");
        if (!(margin >= 0)) return fail(MOLA_ICP_E_BADARG, "margin must be >= 0");
        int rc;
        if ((rc = check_pose(T))) return rc;
        std::lock_guard<std::mutex> lk(h->mtx);
        return h->resident->shard_reach_box(mat_from(T), margin, lo_out, hi_out);
    });
}

int mola_icp_set_map_slab_host(mola_icp_handle* h, const float* x, const float* y, const float* z, size_t M, const double lo[3],
                               const double hi[3], size_t* n_kept_out)
{
    return guarded([&]() -> int {
        if (!h) return fail(MOLA_ICP_E_BADARG, "null handle");
        std::lock_guard<std::mutex> lk(h->mtx);
        return h->resident->set_map_slab(x, y, z, M, lo, hi, false, n_kept_out);
    });
}

int mola_icp_set_map_slab_device(mola_icp_handle* h, const float* dx, const float* dy, const float* dz, size_t M, const double lo[3],
                                 const double hi[3], size_t* n_kept_out)
{
    return guarded([&]() -> int {
        if (!h) return fail(MOLA_ICP_E_BADARG, "null handle");
        std::lock_guard<std::mutex> lk(h->mtx);
        return h->resident->set_map_slab(dx, dy, dz, M, lo, hi, true, n_kept_out);
    });
}

int mola_icp_align_resident(mola_icp_handle* h, const double init_T[16], const mola_icp_params* p,
                            mola_icp_result* out)
{
    return guarded([&]() -> int {
        if (!h || !p || !out) return fail(MOLA_ICP_E_BADARG, "null argument");
        int rc;
        if ((rc = check_pose(init_T))) return rc;
        std::lock_guard<std::mutex> lk(h->mtx);
        std::memset(out, 0, sizeof *out);
        return align_on(*h->resident, init_T, p, out);
    });
}

int mola_icp_match(mola_icp_handle* h, const double T[16], double threshold, int nn_kernel, int32_t* idx_out,
                   float* d2_out, uint64_t* n_pairs_out)
{
    return guarded([&]() -> int {
        if (!h) return fail(MOLA_ICP_E_BADARG, "null handle");
        int rc;
        if ((rc = check_pose(T))) return rc;
        std::lock_guard<std::mutex> lk(h->mtx);
        mola_icp_params p;
        params_default(p);
        p.nn_kernel = nn_kernel;
        uint64_t n = 0;
        if ((rc = h->resident->match(mat_from(T), threshold, p, &n))) return rc;
        if (n_pairs_out) *n_pairs_out = n;
        if (idx_out || d2_out) return h->resident->copy_pairing(idx_out, d2_out);
        return MOLA_ICP_OK;
    });
}

int mola_icp_accumulate(mola_icp_handle* h, const mola_icp_params* p, const double Tcur[16], int stage,
                        const double cl[3], const double cg[3], int reset_outliers, double acc_out[MOLA_ICP_NACC])
{
    return guarded([&]() -> int {
        if (!h || !p || !acc_out) return fail(MOLA_ICP_E_BADARG, "null argument");
        int rc;
        if ((rc = check_pose(Tcur))) return rc;
        std::lock_guard<std::mutex> lk(h->mtx);
        return h->resident->accumulate(*p, mat_from(Tcur), stage, cl, cg, reset_outliers != 0, acc_out);
    });
}

int mola_icp_match_planes(mola_icp_handle* h, const double T[16], const mola_icp_params* p, uint8_t* valid,
                          double* centroid, double* normal, int32_t* knn_idx, uint64_t* n_pairs_out)
{
    return guarded([&]() -> int {
        if (!h || !p) return fail(MOLA_ICP_E_BADARG, "null argument");
        int rc;
        if ((rc = check_pose(T))) return rc;
        std::lock_guard<std::mutex> lk(h->mtx);
        if ((rc = h->resident->match_planes(mat_from(T), *p))) return rc;
        if (n_pairs_out) {
            double acc[kNAccPlaneHost];
            if ((rc = h->resident->accumulate_planes(acc))) return rc;
            *n_pairs_out = (uint64_t)acc[91];
        }
        if (valid || centroid || normal || knn_idx) return h->resident->copy_planes(valid, centroid, normal, knn_idx);
        return MOLA_ICP_OK;
    });
}

int mola_icp_accumulate_planes(mola_icp_handle* h, double acc_out[MOLA_ICP_NACC_PLANES])
{
    return guarded([&]() -> int {
        if (!h || !acc_out) return fail(MOLA_ICP_E_BADARG, "null argument");
        std::lock_guard<std::mutex> lk(h->mtx);
        return h->resident->accumulate_planes(acc_out);
    });
}

int mola_icp_mixed_form(const double acc_p2p[MOLA_ICP_NACC], const double T[16], double form_inout[MOLA_ICP_NACC_PLANES])
{
    if (!acc_p2p || !T || !form_inout) return fail(MOLA_ICP_E_BADARG, "null argument");
    mixed_form(acc_p2p, mat_from(T), form_inout);
    return MOLA_ICP_OK;
}

int mola_icp_solve_gauss_newton_planes(const double acc[MOLA_ICP_NACC_PLANES], const double T0[16],
                                       uint32_t max_iterations, double T_out[16], double* final_cost,
                                       uint32_t* iterations_done)
{
    if (!acc || !T0 || !T_out) return fail(MOLA_ICP_E_BADARG, "null argument");
    Mat4 T;
    unsigned its = 0;
    double cost = 0;
    if (!solve_gauss_newton_planes(acc, mat_from(T0), max_iterations, T, &cost, &its))
        return fail(MOLA_ICP_E_BADARG, "Gauss-Newton: fewer than 3 pairings or singular normal equations");
    std::memcpy(T_out, T.m, sizeof T.m);
    if (final_cost) *final_cost = cost;
    if (iterations_done) *iterations_done = its;
    return MOLA_ICP_OK;
}

int mola_icp_solve_horn(const double acc[MOLA_ICP_NACC], const double* cl, const double* cg, double T_out[16])
{
    if (!acc || !T_out) return fail(MOLA_ICP_E_BADARG, "null argument");
    Mat4 T;
    if (!solve_horn(acc, cl, cg, T)) return fail(MOLA_ICP_E_BADARG, "Horn: no weight / degenerate accumulators");
    std::memcpy(T_out, T.m, sizeof T.m);
    return MOLA_ICP_OK;
}

int mola_icp_stall_deltas(const double T[16], const double Tprev[16], double* d_xyz, double* d_rot)
{
    if (!T || !Tprev || !d_xyz || !d_rot) return fail(MOLA_ICP_E_BADARG, "null argument");
    stall_deltas(mat_from(T), mat_from(Tprev), *d_xyz, *d_rot);
    return MOLA_ICP_OK;
}

int mola_icp_se3_log(const double T[16], double out6[6])
{
    if (!T || !out6) return fail(MOLA_ICP_E_BADARG, "null argument");
    se3_log(mat_from(T), out6);
    return MOLA_ICP_OK;
}

int mola_icp_pose_from_xyzypr(const double p[6], double T_out[16])
{
    if (!p || !T_out) return fail(MOLA_ICP_E_BADARG, "null argument");
    const Mat4 T = pose_from_xyzypr(p);
    std::memcpy(T_out, T.m, sizeof T.m);
    return MOLA_ICP_OK;
}

int mola_icp_pose_to_xyzypr(const double T[16], double out[6])
{
    if (!T || !out) return fail(MOLA_ICP_E_BADARG, "null argument");
    pose_to_xyzypr(mat_from(T), out);
    return MOLA_ICP_OK;
}

int mola_icp_run_loop(const mola_icp_stage_callbacks* cb, const double init_T[16], const mola_icp_params* p,
                      mola_icp_result* out)
{
    return guarded([&]() -> int {
        if (!cb || !cb->match || !cb->accumulate || !p || !out) return fail(MOLA_ICP_E_BADARG, "null argument");
        int rc;
        if ((rc = check_pose(init_T))) return rc;
        std::memset(out, 0, sizeof *out);
        CallbackStages st(cb);
        return run_icp_loop(st, mat_from(init_T), *p, out);
    });
}

int mola_icp_run_loop_batch(const mola_icp_stage_callbacks* cb, size_t n_problems, const double* init_T,
                            const mola_icp_params* p, mola_icp_result* out)
{
    return guarded([&]() -> int {
        if (!p || (n_problems && (!cb || !init_T || !out))) return fail(MOLA_ICP_E_BADARG, "null argument");
        int rc;
        std::vector<Mat4> inits(n_problems);
        for (size_t k = 0; k < n_problems; ++k) {
            if (!cb[k].match || !cb[k].accumulate) return fail(MOLA_ICP_E_BADARG, "null stage callback");
            if ((rc = check_pose(init_T + 16 * k))) return rc;
            inits[k] = mat_from(init_T + 16 * k);
            std::memset(&out[k], 0, sizeof out[k]);
        }
        CallbackBatchStages st(cb, n_problems);
        return run_icp_loop_batch(st, inits.data(), *p, out);
    });
}

}  // extern "C"
