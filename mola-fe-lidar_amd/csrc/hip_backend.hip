// hip_backend.hip -- gfx950 kernels + HipWorkspace (see hip_backend.hpp).
//
// Hot path rows: a7 nearest-neighbour matcher, a8 weighted
// centroid/covariance accumulation.  The reference reaches both through
// mp2p_icp::ICP::align() at src/LidarOdometry.cpp:869-871 (CPU kd-tree +
// serial sums); here they are brute-force tiled kernels over HBM-resident SoA
// clouds.
//
// Numeric contract (DESIGN.md "numeric contract"; the CPU checker restates it, so NN indices compare
// bit-exactly; this file is compiled with -ffp-contract=off so only the
// explicit fmaf() calls fuse):
//   q  = fmaf-chain R*l+t in fp32;  d2 = fmaf(dz,dz,fmaf(dy,dy,dx*dx));
//   NN = argmin d2, ties -> lowest map index; kept iff d2 < thr2.
#include "hip_backend.hpp"

#include <cmath>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sched.h>
#include <iterator>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace mola_icp_amd {

#define HIPCHK(expr)                                                                                          \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess) {                                                                               \
            (void)hipGetLastError(); /* the runtime latches the error: a later hipGetLastError() check of a  \
                                        healthy call would report it again (found by tests/test_gpu_oom.py) */ \
            return fail(e_ == hipErrorOutOfMemory ? MOLA_ICP_E_OOM : MOLA_ICP_E_HIP,                          \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                                   \
        }                                                                                                     \
    } while (0)

}  // namespace mola_icp_amd

// ------------------------------------------------------------------ device code (one translation unit)
#include "kernels_common.hpp"
#include "kernels_dense.hpp"
#include "kernels_tiled.hpp"
#include "kernels_coop.hpp"
#include "q4_launch.hpp"
#include "knn_q4_launch.hpp"
#include "kernels_planes.hpp"
#include "kernels_prepare.hpp"
#include "kernels_accumulate.hpp"

namespace mola_icp_amd {

// ------------------------------------------------------------------ host code

// Diagnostic / tuning environment variables: read ONCE per process -- never on a launch path -- and again only
// when a test asks for it through mola_icp_debug_reload_env().
struct Knobs {
    int blocks_per_cu = 0;     // MOLA_ICP_BLOCKS_PER_CU (0 = default)
    int qpl = 0;               // MOLA_ICP_QPL (0 = by cloud size)
    int coop = -1;             // MOLA_ICP_COOP (-1 = by cloud size, 0 = one item per wave, 1 = one item per block)
    int batch_tiled = -1;      // MOLA_ICP_BATCH_TILED (-1 = by item count; batched launches: 0 = k_nn_coop, 1 = k_nn_tiled_batch)
    bool no_split = false;     // MOLA_ICP_NO_SPLIT: never list a heavy 128-query item as its two halves
    bool no_certify = false;   // MOLA_ICP_NO_CERTIFY: the point-to-plane matcher sweeps for every query at every launch
    double split_share = 0.55; // MOLA_ICP_SPLIT_SHARE: a 128-query item dearer than this share of a wave's fair share is listed as its two halves
    int knn_coop = -1;         // MOLA_ICP_KNN_COOP (-1 = by cloud size, 0 = k_knn_planes, 1 = k_knn_coop: one workgroup per item)
    bool planes_valu = false;  // MOLA_ICP_PLANES_VALU: the plane form accumulated by k_accumulate_planes (VALU) instead of the fp64-MFMA kernel
    bool early_pop = false;    // MOLA_ICP_EARLY_POP: the persistent kernels reserve the next entry at the start of the current one
    bool no_bootstrap = false;   // MOLA_ICP_NO_BOOTSTRAP: the first plane-matcher launch of an align sweeps without seeds
    bool bootstrap_nn = false;   // MOLA_ICP_BOOTSTRAP_NN: ... is seeded around each query's nearest neighbour (an NN pass first) instead of around its Hilbert key's place
    bool no_side_prepare = false;   // MOLA_ICP_NO_SIDE_PREPARE: the two clouds' prepare chains one after the other on one stream
    bool no_quality_lists = false;  // MOLA_ICP_NO_QUALITY_LISTS: the PairedRatio pass behind a point-to-plane loop is always a matcher pass
    bool no_fused_rows = false;  // MOLA_ICP_NO_FUSED_ROWS: k_nn_tiled writes no item rows (k_accumulate sums the pairing, as in round 2)
    bool no_lpt = false, no_knn_seed = false, no_knn_verify = false, no_direct_readback = false, no_warm_start = false;
    int debug_stats = 0;       // MOLA_ICP_DEBUG_STATS
    int lds_boxes_kb = 40;        // MOLA_ICP_LDS_BOXES_KB: the cooperative / plane kernels keep the upper box levels in LDS up to this size (tuning knob; <= 40)
    int quad_lds_boxes_kb = 22;   // MOLA_ICP_QUAD_LDS_BOXES_KB: the quad flavour keeps the upper box levels in LDS up to this size (tuning knob)
    int q4 = -1;               // MOLA_ICP_Q4 (-1 = by cloud size, 0 = never, 1 = always: k_nn_q4, four lanes per query, instead of k_nn_coop / k_nn_tiled)
    int knn_q4 = -1;           // MOLA_ICP_KNN_Q4 (-1 = k_knn_coop's sizes near the previous pose + every launch beyond them; 0 = never; 1 = every launch): k_knn_q4, four / two / one lane(s) per query
    int knn_q4_lpq = 0;        // MOLA_ICP_KNN_Q4_LPQ (0 = by the launch's size, 1 / 2 / 4: k_knn_q4's lanes per query)
    int q4_lds_boxes_kb = -1;  // MOLA_ICP_Q4_LDS_BOXES_KB: k_nn_q4 keeps the upper box levels in LDS up to this size (-1: what costs it no workgroup per CU)
    int quads = -1;            // MOLA_ICP_QUADS (-1 = by cloud sizes, 0 = never, 1 = always: k_nn_tiled's quad flavour)
    bool no_stream_priority = false;   // MOLA_ICP_NO_STREAM_PRIORITY: every workspace's streams at the default priority (A/B of bench.py's mixed_load leg)
    int wait_policy = 0;       // MOLA_ICP_WAIT=spin|yield|block: how the host thread waits for a pass's sums (0 spin -- the default --, 1 yield, 2 block)
    bool turn_clock = false;   // MOLA_ICP_TURN_CLOCK: print the host's side of an iteration's turn (product kernels; stderr, every 200 turns)
};
static Knobs read_knobs()
{
    Knobs k;
    auto geti = [](const char* name) { const char* e = std::getenv(name); return e ? std::atoi(e) : 0; };
    k.blocks_per_cu = geti("MOLA_ICP_BLOCKS_PER_CU") > 0 ? geti("MOLA_ICP_BLOCKS_PER_CU") : 0;
    k.qpl = std::getenv("MOLA_ICP_QPL") ? (geti("MOLA_ICP_QPL") == 1 ? 1 : 2) : 0;
    k.coop = std::getenv("MOLA_ICP_COOP") ? (geti("MOLA_ICP_COOP") != 0 ? 1 : 0) : -1;
    k.batch_tiled = std::getenv("MOLA_ICP_BATCH_TILED") ? (geti("MOLA_ICP_BATCH_TILED") != 0 ? 1 : 0) : -1;
    k.no_lpt = std::getenv("MOLA_ICP_NO_LPT") != nullptr;
    k.no_split = std::getenv("MOLA_ICP_NO_SPLIT") != nullptr;
    k.early_pop = std::getenv("MOLA_ICP_EARLY_POP") != nullptr;
    k.no_fused_rows = std::getenv("MOLA_ICP_NO_FUSED_ROWS") != nullptr;
    k.no_quality_lists = std::getenv("MOLA_ICP_NO_QUALITY_LISTS") != nullptr;
    k.no_side_prepare = std::getenv("MOLA_ICP_NO_SIDE_PREPARE") != nullptr;
    k.no_bootstrap = std::getenv("MOLA_ICP_NO_BOOTSTRAP") != nullptr;
    k.bootstrap_nn = std::getenv("MOLA_ICP_BOOTSTRAP_NN") != nullptr;
    k.planes_valu = std::getenv("MOLA_ICP_PLANES_VALU") != nullptr;
    k.knn_coop = std::getenv("MOLA_ICP_KNN_COOP") ? (geti("MOLA_ICP_KNN_COOP") != 0 ? 1 : 0) : -1;
    if (const char* e = std::getenv("MOLA_ICP_SPLIT_SHARE")) { const double v = std::atof(e); if (v > 0.01 && v < 100.0) k.split_share = v; }
    k.no_certify = std::getenv("MOLA_ICP_NO_CERTIFY") != nullptr;
    k.no_knn_seed = std::getenv("MOLA_ICP_NO_KNN_SEED") != nullptr;
    k.no_knn_verify = std::getenv("MOLA_ICP_NO_KNN_VERIFY") != nullptr;
    k.no_direct_readback = std::getenv("MOLA_ICP_NO_DIRECT_READBACK") != nullptr;
    k.no_warm_start = std::getenv("MOLA_ICP_NO_WARM_START") != nullptr;
    k.debug_stats = geti("MOLA_ICP_DEBUG_STATS");
    k.turn_clock = std::getenv("MOLA_ICP_TURN_CLOCK") != nullptr;
    k.no_stream_priority = std::getenv("MOLA_ICP_NO_STREAM_PRIORITY") != nullptr;
    if (const char* e = std::getenv("MOLA_ICP_WAIT")) k.wait_policy = std::strcmp(e, "yield") == 0 ? 1 : (std::strcmp(e, "block") == 0 ? 2 : 0);
    if (std::getenv("MOLA_ICP_LDS_BOXES_KB")) { k.lds_boxes_kb = geti("MOLA_ICP_LDS_BOXES_KB"); if (k.lds_boxes_kb > 40) k.lds_boxes_kb = 40; if (k.lds_boxes_kb < 0) k.lds_boxes_kb = 0; }
    if (std::getenv("MOLA_ICP_QUAD_LDS_BOXES_KB")) k.quad_lds_boxes_kb = geti("MOLA_ICP_QUAD_LDS_BOXES_KB");
    k.knn_q4 = std::getenv("MOLA_ICP_KNN_Q4") ? (geti("MOLA_ICP_KNN_Q4") != 0 ? 1 : 0) : -1;
    if (std::getenv("MOLA_ICP_KNN_Q4_LPQ")) k.knn_q4_lpq = geti("MOLA_ICP_KNN_Q4_LPQ") == 2 ? 2 : (geti("MOLA_ICP_KNN_Q4_LPQ") == 4 ? 4 : (geti("MOLA_ICP_KNN_Q4_LPQ") == 1 ? 1 : 0));
    k.q4 = std::getenv("MOLA_ICP_Q4") ? (geti("MOLA_ICP_Q4") != 0 ? 1 : 0) : -1;
    if (std::getenv("MOLA_ICP_Q4_LDS_BOXES_KB")) k.q4_lds_boxes_kb = geti("MOLA_ICP_Q4_LDS_BOXES_KB");
    k.quads = std::getenv("MOLA_ICP_QUADS") ? (geti("MOLA_ICP_QUADS") != 0 ? 1 : 0) : -1;
    return k;
}
// The upper box levels go into LDS only while that leaves the kernel the workgroups per CU its registers allow (160 KB of LDS per CU):
// a copy that costs the fourth workgroup costs more than the global reads it saves -- a 120k-point scan against a 2M-point map through
// k_knn_coop 0.227 -> 0.193 ms per iteration, against 3M points 0.359 -> 0.243; k_nn_tiled's quad flavour 1M x 3M 0.170 -> 0.145
// (MOLA_ICP_LDS_BOXES_KB caps it further; 40 KB was the limit for every kernel until round 5).
static size_t lds_box_limit(size_t static_lds_bytes, int workgroups_per_cu);
static const char* const kSlabMsg =
    "the pose moves a shard's reach outside its map slab: cut the slab again with a larger margin";
static Knobs g_knobs = read_knobs();
void reload_env_knobs() { g_knobs = read_knobs(); }
static size_t g_lds_per_cu = (size_t)160 * 1024;   // (gfx950; HipWorkspace::init reads the device's own figure)
static size_t lds_box_limit(size_t static_lds_bytes, int workgroups_per_cu)
{
    const size_t per_wg = g_lds_per_cu / (size_t)(workgroups_per_cu > 0 ? workgroups_per_cu : 1);
    const size_t room = per_wg > static_lds_bytes + 512 ? per_wg - static_lds_bytes - 512 : 0;   // (512 B of slack for the allocation granule)
    const size_t cap = (size_t)g_knobs.lds_boxes_kb * 1024;
    return room < cap ? room : cap;
}
// What the plane kernels receive as planeEigenThreshold: its SIGN carries the reading `p2pl_all_inside_gate` (mola_icp_params: a plane
// needs ALL knn neighbours inside the gate instead of >= 3; plane_epilogue decodes it) -- so the flag also takes part wherever the
// threshold is compared to decide whether cached planes may be reused.
static inline double plane_eig_arg(const mola_icp_params& p)
{
    return p.reading_p2pl_all_inside_gate ? -std::fabs(p.plane_eigen_threshold) : std::fabs(p.plane_eigen_threshold);
}
constexpr size_t kQ4MaxQueries = 230000;   // k_nn_q4 serves launches up to this many queries (HipWorkspace::launch_nn)
// static LDS of a kernel, asked of the code object itself (once per kernel): nothing here is hand-copied from the kernels' __shared__
// declarations, so an edit there moves the occupancy cliff with it
template <class Kernel> static size_t static_lds_of(Kernel kernel, size_t fallback)
{
    hipFuncAttributes fa{};
    return hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kernel)) == hipSuccess && fa.sharedSizeBytes ? (size_t)fa.sharedSizeBytes : fallback;
}
static size_t nn_coop_static_lds()
{
    static const size_t v = static_lds_of(&k_nn_coop<1, false>, 10800);
    return v;
}
static size_t persistent_static_lds()   // (k_nn_tiled_batch: s_m + s_list, as the other persistent kernels without the quad buffers)
{
    static const size_t v = static_lds_of(&k_nn_tiled_batch<kCoopMaxBatch, 1>, 5120);
    return v;
}
static size_t knn_coop_static_lds(int list_len)   // sizeof(KnnCoopLds<K>), K = knn + 1
{
    switch (list_len) {
        case 4: return sizeof(KnnCoopLds<4>); case 5: return sizeof(KnnCoopLds<5>); case 6: return sizeof(KnnCoopLds<6>); case 7: return sizeof(KnnCoopLds<7>);
        case 8: return sizeof(KnnCoopLds<8>); case 9: return sizeof(KnnCoopLds<9>); case 10: return sizeof(KnnCoopLds<10>); case 11: return sizeof(KnnCoopLds<11>);
        case 12: return sizeof(KnnCoopLds<12>); case 13: return sizeof(KnnCoopLds<13>); case 14: return sizeof(KnnCoopLds<14>); case 15: return sizeof(KnnCoopLds<15>);
        case 16: return sizeof(KnnCoopLds<16>); default: return sizeof(KnnCoopLds<17>);
    }
}
// k_knn_coop's box levels: its workgroups per CU by list length (launch bounds: four up to seven entries, else three), and a workgroup's
// static + dynamic LDS within 64 KB
static size_t knn_coop_lds_box_limit(int list_len)
{
    const size_t st = knn_coop_static_lds(list_len);
    const size_t a = lds_box_limit(st, list_len <= 7 ? 4 : 3);
    const size_t b = st + 1024 < (size_t)64 * 1024 ? (size_t)64 * 1024 - st - 1024 : 0;
    return a < b ? a : b;
}

// k_knn_q4's box levels (kernels_knn_q4.hpp): what its own static LDS leaves at its launch bounds
static size_t knn_q4_lds_box_limit(int list_len, int lpq)
{
    const size_t st = knn_q4_static_lds(list_len, lpq);
    const size_t a = lds_box_limit(st, knn_q4_workgroups_per_cu(lpq));
    const size_t b = st + 1024 < (size_t)64 * 1024 ? (size_t)64 * 1024 - st - 1024 : 0;
    return a < b ? a : b;
}
// ... and whether a launch k_knn_coop would serve goes to k_knn_q4 instead (the diagnostic flavours are k_knn_coop's)
constexpr double kKnnQ4MaxStep = 0.25;   // metres of pose step (HipWorkspace::match_planes)
constexpr size_t kKnnQ4MaxQueries = 16000000;   // (beyond: the persistent kernel -- untested territory for k_knn_q4, not a measured crossover)
// k_knn_q4's lanes per query for a launch of `workgroups` 64-query workgroups.  FOUR while their waves (four each) fit the wave slots the kernel
// has, and for a launch on key-bootstrapped seeds of an odometry-size scan, which is bound by its insertions (a KITTI-like 120k scan: 95-100 us at
// four lanes, 119-122 at two).  Beyond that TWO -- half the waves, each ~1.3x as long -- up to ~245k queries, then ONE (a wave = a whole row of 64,
// no lists to merge, the epilogue from registers).  20-iteration shipped aligns, ms per iteration four | two | one lane(s): 100k 0.054 | 0.051 | -,
// 120k - | 0.057 | 0.061, 300k 0.098 | 0.085 | 0.079, 1M 0.262 | 0.181 | 0.169, 3M - | 0.547 | 0.490, 5M - | 0.967 | 0.849 (the persistent
// k_knn_planes: 1M 0.221, 3M 0.566, 5M 0.938); 24 pairs of 100k in lockstep 1 774 | 2 196 | 2 254 pairs/s.
static int knn_q4_lanes_per_query(int list_len, size_t workgroups, int num_cus, bool insertion_bound)
{
    if (g_knobs.knn_q4_lpq && knn_q4_has(list_len, g_knobs.knn_q4_lpq)) return g_knobs.knn_q4_lpq;
    const size_t slots = (size_t)num_cus * 4u * (size_t)knn_q4_workgroups_per_cu(4);
    if (workgroups * 4u <= slots || insertion_bound) return 4;
    if (workgroups * 4u > slots * 3u || !knn_q4_has(list_len, 2)) return 1;
    return 2;
}
static bool use_knn_q4(int list_len)
{
    return g_knobs.knn_q4 != 0 && knn_q4_has(list_len) && g_knobs.debug_stats != 4 && g_knobs.debug_stats != 5;
}

// ---- parked device blocks (DevBuf::pooled) ------------------------------------------------------------------
// An odometry stream drops one cached cloud per scan: six hipFree calls, ~90 us of a 0.75-ms scan (HIP API trace), each of
// them a device-wide synchronisation that also stalls every other handle's stream.  Parked blocks are handed out again to
// requests they fit without much waste; beyond MOLA_ICP_POOL_MB (default 1024) per device a released block is freed at once.
// The lists are never destroyed (the HIP runtime may be gone before static destructors run; the process's teardown
// returns the memory).
namespace {
constexpr int kPoolDevices = 16;
struct BlockPool {
    std::mutex m;
    std::multimap<size_t, void*> free_[kPoolDevices];
    size_t bytes[kPoolDevices] = {};
    size_t limit = 1024ull << 20;
    BlockPool()
    {
        if (const char* e = std::getenv("MOLA_ICP_POOL_MB")) limit = (size_t)std::strtoull(e, nullptr, 10) << 20;
    }
};
BlockPool& block_pool()
{
    static BlockPool* pool = new BlockPool;   // (deliberately leaked: see above)
    return *pool;
}
}  // namespace

void device_pool_trim(size_t keep_bytes, int device)
{
    BlockPool& bp = block_pool();
    std::lock_guard<std::mutex> lk(bp.m);
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (int d = 0; d < kPoolDevices; ++d) {
        if (device >= 0 && d != device) continue;
        if (bp.free_[d].empty()) continue;
        (void)hipSetDevice(d);
        while (bp.bytes[d] > keep_bytes && !bp.free_[d].empty()) {   // the largest blocks first
            auto it = std::prev(bp.free_[d].end());
            (void)hipFree(it->second);
            bp.bytes[d] -= it->first;
            bp.free_[d].erase(it);
        }
    }
    (void)hipSetDevice(cur);
}

size_t device_pool_bytes(int device)
{
    BlockPool& bp = block_pool();
    std::lock_guard<std::mutex> lk(bp.m);
    return device >= 0 && device < kPoolDevices ? bp.bytes[device] : 0;
}

int DevBuf::reserve(size_t bytes)
{
    if (bytes <= cap && p) return MOLA_ICP_OK;
    release();
    size_t want = bytes < 256 ? 256 : bytes;
    if (pooled) {
        want = (want + 4095) / 4096 * 4096;
        int d = 0;
        HIPCHK(hipGetDevice(&d));
        dev = d;
        if (d >= 0 && d < kPoolDevices) {
            BlockPool& bp = block_pool();
            std::lock_guard<std::mutex> lk(bp.m);
            auto it = bp.free_[d].lower_bound(want);
            if (it != bp.free_[d].end() && it->first <= want + want / 2 + 65536) {   // (a fit without much waste)
                p = it->second;
                cap = it->first;
                bp.bytes[d] -= cap;
                bp.free_[d].erase(it);
                return MOLA_ICP_OK;
            }
        }
    }
    hipError_t e = hipMalloc(&p, want);
    if (e == hipErrorOutOfMemory) {   // parked blocks are free memory as far as the caller is concerned
        (void)hipGetLastError();
        device_pool_trim(0, -1);
        e = hipMalloc(&p, want);
    }
    if (e != hipSuccess) {
        p = nullptr;
        (void)hipGetLastError();   // (not latched for the next call's check: the handle stays usable -- SURVEY section 5, "never abort")
        return fail(e == hipErrorOutOfMemory ? MOLA_ICP_E_OOM : MOLA_ICP_E_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
    }
    cap = want;
    return MOLA_ICP_OK;
}

void DevBuf::release()
{
    if (p && pooled && dev >= 0 && dev < kPoolDevices) {
        BlockPool& bp = block_pool();
        std::lock_guard<std::mutex> lk(bp.m);
        if (bp.bytes[dev] + cap <= bp.limit) {
            bp.free_[dev].emplace(cap, p);
            bp.bytes[dev] += cap;
            p = nullptr;
            cap = 0;
            return;
        }
    }
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}

HipWorkspace::HipWorkspace(int device, int priority)
    : device_(device), priority_(priority), map_sc_(std::make_shared<SortedCloud>()), loc_sc_(std::make_shared<SortedCloud>())
{
}

HipWorkspace::~HipWorkspace()
{
    if (!inited_) return;
    (void)hipSetDevice(device_);
    if (stream_) (void)hipStreamSynchronize(stream_);
    if (aux_stream_) { (void)hipStreamSynchronize(aux_stream_); (void)hipStreamDestroy(aux_stream_); }
    if (ev_order_a_) (void)hipEventDestroy(ev_order_a_);
    if (ev_prep_) (void)hipEventDestroy(ev_prep_);
    if (ev_order_b_) (void)hipEventDestroy(ev_order_b_);
    for (hipEvent_t e : ev_) (void)hipEventDestroy(e);
    map_own_.release(); loc_own_.release(); map_img_.release(); map_meta_.release();
    shard_idx_.release(); slab_orig_.release(); stage_in_.release();
    batch_scratch_.release_all();
    map_sc_.reset();
    loc_sc_.reset();
    ts_pos_.release(); ts_idx_.release(); ts_d2_.release(); ts_gs_.release(); rows_.release(); item_cost_.release(); item_order_.release(); redo_list_.release(); knn_cost_.release(); knn_order_.release();
    planes_.release(); knn_pos_.release(); plane_acc_.release(); plane_cache_.release();
    if (plane_acc_host_) (void)hipHostFree(plane_acc_host_);
    if (item_part_host_) (void)hipHostFree(item_part_host_);
    if (quality_host_) (void)hipHostFree(quality_host_);
    item_part_.release();
    sort_scratch_.release(); sort_scratch_loc_.release(); loc_meta_.release();
    idx_.release(); d2_.release(); seg_idx_.release(); seg_d2_.release(); outlier_.release(); partials_.release(); acc_dev_.release();
    if (acc_host_) (void)hipHostFree(acc_host_);
    if (meta_host_) (void)hipHostFree(meta_host_);
    if (stats_host_) (void)hipHostFree(stats_host_);
    stats_.release();
    if (own_stream_ && stream_) (void)hipStreamDestroy(stream_);
}

// MOLA_ICP_TURN_CLOCK: the host's side of an iteration's turn (result block seen -> matcher call entered -> its launch call returned)
namespace {
struct TurnClock {
    std::chrono::steady_clock::time_point seen, entered;
    bool have_seen = false, have_entered = false;
    double sum_host = 0, sum_launch = 0;
    unsigned long long n = 0;
    void on_seen() { seen = std::chrono::steady_clock::now(); have_seen = true; }
    void on_enter() { if (have_seen) { entered = std::chrono::steady_clock::now(); have_entered = true; } }
    void on_launched()
    {
        if (!have_seen || !have_entered) return;
        const auto t = std::chrono::steady_clock::now();
        sum_host += std::chrono::duration<double, std::micro>(entered - seen).count();
        sum_launch += std::chrono::duration<double, std::micro>(t - entered).count();
        have_seen = have_entered = false;
        if (++n % 200 == 0)
            std::fprintf(stderr, "[mola_icp debug] host turn over %llu iterations: result seen -> matcher call entered %.2f us, entered -> launch call returned %.2f us\n",
                         n, sum_host / (double)n, sum_launch / (double)n);
    }
};
thread_local TurnClock g_turn;
}  // namespace

int HipWorkspace::init()
{
    if (inited_) return MOLA_ICP_OK;
    int count = 0;
    const hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(MOLA_ICP_E_NODEVICE, std::string("no HIP device available (") +
                                             (e == hipSuccess ? "0 devices" : hipGetErrorString(e)) +
                                             "); this library has no CPU fallback");
    if (device_ < 0) HIPCHK(hipGetDevice(&device_));
    if (device_ >= count) return fail(MOLA_ICP_E_BADARG, "device index out of range");
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device_));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(MOLA_ICP_E_NODEVICE,
                    std::string("device is ") + prop.gcnArchName + " but the kernels are built for gfx950 only");
    num_cus_ = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (prop.maxSharedMemoryPerMultiProcessor >= (size_t)64 * 1024) g_lds_per_cu = prop.maxSharedMemoryPerMultiProcessor;   // (160 KB on gfx950)
    HIPCHK(hipSetDevice(device_));
    // Stream priority: the reference's odometry thread and its max(2, hw/2) pool threads call align() on the same ICP object at the
    // same time (src/LidarOdometry.cpp:94-96, 183-184, 711-712, 869); the odometry step is the one with a deadline (10 Hz), so the
    // workspaces it leases run on streams of the device's greatest priority: their launches are picked ahead of queued launches
    // of the nearby / loop-closure batches (mola_icp_set_thread_priority; running waves are not preempted).
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    const int prio = (priority_ > 0 && !g_knobs.no_stream_priority) ? prio_greatest : 0;
    HIPCHK(hipStreamCreateWithPriority(&stream_, hipStreamNonBlocking, prio));
    own_stream_ = true;
    HIPCHK(hipStreamCreateWithPriority(&aux_stream_, hipStreamNonBlocking, prio));
    HIPCHK(hipEventCreateWithFlags(&ev_order_a_, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&ev_prep_, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&ev_order_b_, hipEventDisableTiming));
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&acc_host_), sizeof(double) * (kNAcc + 8), hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(acc_host_, 0, sizeof(double) * (kNAcc + 8));
    // (the sort's first kernel writes a cloud's bounding box straight into this block: mapped, coherent)
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&meta_host_), sizeof(float) * 16, hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(meta_host_, 0, sizeof(float) * 16);
    int rc;
    if ((rc = acc_dev_.reserve(sizeof(double) * (kNAcc + 8) + sizeof(unsigned int) * 2 * kQueues * kQueueStride))) return rc;
    HIPCHK(hipMemsetAsync(acc_dev_.p, 0, sizeof(double) * (kNAcc + 8) + sizeof(unsigned int) * 2 * kQueues * kQueueStride, stream_));   // (k_reduce_items' finished-blocks word among them)
    if ((rc = stats_.reserve(sizeof(unsigned long long) * kStatSlots * kStatStride))) return rc;
    HIPCHK(hipMemsetAsync(stats_.p, 0, sizeof(unsigned long long) * kStatSlots * kStatStride, stream_));
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&stats_host_), sizeof(unsigned long long) * kStatSlots * kStatStride, hipHostMallocDefault));
    if (g_knobs.debug_stats) {  // diagnostic builds of a run, never on by default
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&dbg_stats_), (16 + 8 * kDbgItems) * sizeof(unsigned long long)));
        HIPCHK(hipMemset(dbg_stats_, 0, (16 + 8 * kDbgItems) * sizeof(unsigned long long)));
        if (g_knobs.debug_stats == 2) {  // light mode: per-wave start/end of the tiled matcher only
            HIPCHK(hipMalloc(reinterpret_cast<void**>(&wave_times_), 8 * 8192 * sizeof(unsigned long long)));
            HIPCHK(hipMemset(wave_times_, 0, 8 * 8192 * sizeof(unsigned long long)));
        }
    }
    inited_ = true;
    return MOLA_ICP_OK;
}

int HipWorkspace::set_external_stream(void* s)
{
    int rc = init();
    if (rc) return rc;
    HIPCHK(hipSetDevice(device_));
    if (own_stream_ && stream_) {
        HIPCHK(hipStreamSynchronize(stream_));
        HIPCHK(hipStreamDestroy(stream_));
    }
    if (s) {
        stream_ = static_cast<hipStream_t>(s);
        own_stream_ = false;
    } else {
        HIPCHK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
        own_stream_ = true;
    }
    return MOLA_ICP_OK;
}

static int upload_soa(DevBuf& buf, hipStream_t st, const float* x, const float* y, const float* z, size_t n,
                      const float** dx, const float** dy, const float** dz)
{
    // padded to a multiple of 64 floats per component so the three arrays stay 256-B aligned
    const size_t np = (n + 63) / 64 * 64;
    int rc = buf.reserve(sizeof(float) * 3 * (np ? np : 64));
    if (rc) return rc;
    float* base = buf.as<float>();
    if (n && y == x + n && z == y + n) {
        // one [3][n] block on the host (what the Python binding and a row-major 3 x n matrix hand over): one copy instead of
        // three (each pageable copy is a staged, synchronous call of its own); the rows land np floats apart
        if (n == np) HIPCHK(hipMemcpyAsync(base, x, sizeof(float) * 3 * n, hipMemcpyHostToDevice, st));
        else HIPCHK(hipMemcpy2DAsync(base, sizeof(float) * np, x, sizeof(float) * n, sizeof(float) * n, 3, hipMemcpyHostToDevice, st));
    } else if (n) {
        HIPCHK(hipMemcpyAsync(base, x, sizeof(float) * n, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(base + np, y, sizeof(float) * n, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(base + 2 * np, z, sizeof(float) * n, hipMemcpyHostToDevice, st));
    }
    *dx = base; *dy = base + np; *dz = base + 2 * np;
    return MOLA_ICP_OK;
}

int HipWorkspace::set_map_host(const float* x, const float* y, const float* z, size_t M, bool wait)
{
    int rc = init();
    if (rc) return rc;
    if (M && (!x || !y || !z)) return fail(MOLA_ICP_E_BADARG, "null map pointer");
    if (M > (size_t)0x7fff0000) return fail(MOLA_ICP_E_BADARG, "map too large for 32-bit indices");
    HIPCHK(hipSetDevice(device_));
    if ((rc = upload_soa(map_own_, stream_, x, y, z, M, &gx_, &gy_, &gz_))) return rc;
    // the host buffers may be pageable: finish the copies before returning (never retain caller pointers) -- unless the caller
    // keeps them alive until it has waited for this stream itself (wait = false: mola_icp_align's own frame)
    if (wait) HIPCHK(hipStreamSynchronize(stream_));
    M_ = M;
    slab_active_ = false;
    slab_violation_ = false;
    planes_valid_ = false;
    map_img_valid_ = false;
    if (map_sc_->cached) map_sc_ = std::make_shared<SortedCloud>();  // a cached cloud is immutable: use an own one
    map_sc_->ready = false;
    pairing_valid_ = false;
    seed_valid_ = false;
    knn_seed_valid_ = false;
    return MOLA_ICP_OK;
}

int HipWorkspace::set_map_device(const float* x, const float* y, const float* z, size_t M)
{
    int rc = init();
    if (rc) return rc;
    if (M && (!x || !y || !z)) return fail(MOLA_ICP_E_BADARG, "null map pointer");
    if (M > (size_t)0x7fff0000) return fail(MOLA_ICP_E_BADARG, "map too large for 32-bit indices");
    gx_ = x; gy_ = y; gz_ = z;
    M_ = M;
    slab_active_ = false;
    slab_violation_ = false;
    planes_valid_ = false;
    map_img_valid_ = false;
    if (map_sc_->cached) map_sc_ = std::make_shared<SortedCloud>();  // a cached cloud is immutable: use an own one
    map_sc_->ready = false;
    pairing_valid_ = false;
    seed_valid_ = false;
    knn_seed_valid_ = false;
    return MOLA_ICP_OK;
}

int HipWorkspace::set_local_host(const float* x, const float* y, const float* z, size_t N, bool wait)
{
    int rc = init();
    if (rc) return rc;
    if (N && (!x || !y || !z)) return fail(MOLA_ICP_E_BADARG, "null local-cloud pointer");
    if (N > (size_t)0x7fff0000) return fail(MOLA_ICP_E_BADARG, "local cloud too large for 32-bit indices");
    HIPCHK(hipSetDevice(device_));
    if ((rc = upload_soa(loc_own_, stream_, x, y, z, N, &lx_, &ly_, &lz_))) return rc;
    if (wait) HIPCHK(hipStreamSynchronize(stream_));
    N_ = N;
    loc_bbox_valid_ = false;
    shard_n_ = 0;
    planes_valid_ = false;
    if (loc_sc_->cached) loc_sc_ = std::make_shared<SortedCloud>();
    loc_sc_->ready = false;
    cost_valid_ = false;
    order_valid_ = false;
    knn_cost_valid_ = false;
    knn_order_valid_ = false;
    knn_seed_valid_ = false;
    pairing_valid_ = false;
    seed_valid_ = false;
    knn_seed_valid_ = false;
    return MOLA_ICP_OK;
}

int HipWorkspace::set_local_device(const float* x, const float* y, const float* z, size_t N)
{
    int rc = init();
    if (rc) return rc;
    if (N && (!x || !y || !z)) return fail(MOLA_ICP_E_BADARG, "null local-cloud pointer");
    if (N > (size_t)0x7fff0000) return fail(MOLA_ICP_E_BADARG, "local cloud too large for 32-bit indices");
    lx_ = x; ly_ = y; lz_ = z;
    N_ = N;
    loc_bbox_valid_ = false;
    shard_n_ = 0;
    planes_valid_ = false;
    if (loc_sc_->cached) loc_sc_ = std::make_shared<SortedCloud>();
    loc_sc_->ready = false;
    cost_valid_ = false;
    order_valid_ = false;
    knn_cost_valid_ = false;
    knn_order_valid_ = false;
    knn_seed_valid_ = false;
    pairing_valid_ = false;
    seed_valid_ = false;
    knn_seed_valid_ = false;
    return MOLA_ICP_OK;
}

int hilbert_sort_points(hipStream_t stream, const float* gx, const float* gy, const float* gz, size_t M, size_t M_padded,
                        const float* box_rows, int n_box_rows, float* box_dev, float* box_host, DevBuf& scratch, float* sxyz, int* perm,
                        float* sbox, int n_super, float* ubox, int n_top, unsigned int* keys_sorted);
int boxes_of_sorted(hipStream_t stream, const float* sxyz, size_t M, size_t M_padded, int n_tiles_p, int n_super, int n_top, float* tbox,
                    float* sbox, float* ubox, const float* cloud_box);
int select_in_box(hipStream_t stream, const float* x, const float* y, const float* z, size_t n, const float lo[3], const float hi[3],
                  DevBuf& scratch, int* sel, size_t* n_kept_host);
int gather_by_index(hipStream_t stream, const float* x, const float* y, const float* z, const int* sel, size_t n, float* ox, float* oy,
                    float* oz);

// Row e.  Every rank sees the whole scan (it is small: 12 B per point) but keeps only a spatially compact shard of it: the
// slice [lo, hi) of the scan's Hilbert order.  A random 1/W subsample would be W times sparser than the map, and every
// 128-query group would sweep W times more map tiles.  The order is computed HERE, on the device, with the whole scan's
// bounding box, so every rank cuts the same order; the full copy is transient.
int HipWorkspace::set_local_shard(const float* x, const float* y, const float* z, size_t n_total, int rank, int nranks, bool on_device)
{
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(MOLA_ICP_E_BADARG, "bad rank / nranks");
    const size_t base = n_total / (size_t)nranks, rem = n_total % (size_t)nranks;
    const size_t lo = (size_t)rank * base + std::min((size_t)rank, rem), n = base + ((size_t)rank < rem ? 1 : 0);
    return set_local_shard_range(x, y, z, n_total, lo, lo + n, on_device);
}

// ... and any slice [lo, hi) of that order (cost-balanced cuts: sharded.balanced_cuts)
int HipWorkspace::set_local_shard_range(const float* x, const float* y, const float* z, size_t n_total, size_t lo, size_t hi, bool on_device)
{
    int rc = init();
    if (rc) return rc;
    if (lo > hi || hi > n_total) return fail(MOLA_ICP_E_BADARG, "shard range outside the scan");
    if (n_total && (!x || !y || !z)) return fail(MOLA_ICP_E_BADARG, "null local-cloud pointer");
    if (n_total > (size_t)0x7fff0000) return fail(MOLA_ICP_E_BADARG, "local cloud too large for 32-bit indices");
    HIPCHK(hipSetDevice(device_));
    const size_t n = hi - lo;
    const float *fx = x, *fy = y, *fz = z;
    if (!on_device) {
        if ((rc = upload_soa(stage_in_, stream_, x, y, z, n_total, &fx, &fy, &fz))) return rc;
        HIPCHK(hipStreamSynchronize(stream_));  // (pageable host buffers: never retain caller pointers)
    }
    DevBuf sorted, perm;
    float* own = nullptr;
    if (n_total) {
        float bbox[6];
        if ((rc = bbox_of(fx, fy, fz, n_total, bbox))) return rc;
        const size_t padded = (n_total + kQPW - 1) / kQPW * kQPW;
        if ((rc = sorted.reserve(sizeof(float) * 3 * padded))) return rc;
        if ((rc = perm.reserve(sizeof(int) * padded))) { sorted.release(); return rc; }   // (an OOM path: leak nothing)
        // (bbox_of left the finished box in the device block: one row)
        if ((rc = hilbert_sort_points(stream_, fx, fy, fz, n_total, padded, bbox_dev(), 1, nullptr, nullptr, sort_scratch_, sorted.as<float>(),
                                      perm.as<int>(), nullptr, 0, nullptr, 0, nullptr))) {
            sorted.release(); perm.release();
            return rc;
        }
        const size_t np = (n + 63) / 64 * 64;
        rc = loc_own_.reserve(sizeof(float) * 3 * (np ? np : 64));
        if (!rc) rc = shard_idx_.reserve(sizeof(int) * (n ? n : 1));
        if (rc) { sorted.release(); perm.release(); return rc; }
        own = loc_own_.as<float>();
        const float* sp = sorted.as<float>();
        hipError_t e = hipSuccess;
        for (int a = 0; a < 3 && e == hipSuccess && n; ++a)
            e = hipMemcpyAsync(own + (size_t)a * np, sp + (size_t)a * padded + lo, sizeof(float) * n, hipMemcpyDeviceToDevice, stream_);
        if (e == hipSuccess && n) e = hipMemcpyAsync(shard_idx_.p, perm.as<int>() + lo, sizeof(int) * n, hipMemcpyDeviceToDevice, stream_);
        if (e == hipSuccess) e = hipStreamSynchronize(stream_);
        sorted.release(); perm.release();
        stage_in_.release();  // the full scan was transient
        if (e != hipSuccess) return fail(MOLA_ICP_E_HIP, std::string("shard copy: ") + hipGetErrorString(e));
        rc = set_local_device(own, own + np, own + 2 * np, n);
    } else {
        rc = set_local_device(nullptr, nullptr, nullptr, 0);
    }
    if (rc) return rc;
    shard_n_ = n;
    return MOLA_ICP_OK;
}

int HipWorkspace::copy_shard_indices(int32_t* idx_out)
{
    if (shard_n_ == 0) return MOLA_ICP_OK;
    if (!idx_out) return fail(MOLA_ICP_E_BADARG, "null output");
    HIPCHK(hipSetDevice(device_));
    HIPCHK(hipMemcpyAsync(idx_out, shard_idx_.p, sizeof(int) * shard_n_, hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    return MOLA_ICP_OK;
}

// AABB of the local cloud's bounding box moved by T, grown by `margin` on every side: where this rank's queries can
// find neighbours at pose T with a gate <= margin.
int HipWorkspace::shard_reach_box(const Mat4& T, double margin, double lo[3], double hi[3])
{
    int rc = init();
    if (rc) return rc;
    if (N_ == 0) {  // an empty shard (more ranks than points) reaches nothing: an empty box, no map point kept
        for (int a = 0; a < 3; ++a) { lo[a] = 1e30; hi[a] = -1e30; }
        return MOLA_ICP_OK;
    }
    if (!loc_bbox_valid_) {
        HIPCHK(hipSetDevice(device_));
        if ((rc = bbox_of(lx_, ly_, lz_, N_, loc_bbox_))) return rc;
        loc_bbox_valid_ = true;
    }
    for (int a = 0; a < 3; ++a) { lo[a] = INFINITY; hi[a] = -INFINITY; }
    for (int c = 0; c < 8; ++c) {
        const double p[3] = {loc_bbox_[(c & 1) ? 3 : 0], loc_bbox_[(c & 2) ? 4 : 1], loc_bbox_[(c & 4) ? 5 : 2]};
        for (int a = 0; a < 3; ++a) {
            const double v = T(a, 0) * p[0] + T(a, 1) * p[1] + T(a, 2) * p[2] + T(a, 3);
            lo[a] = std::min(lo[a], v);
            hi[a] = std::max(hi[a], v);
        }
    }
    // (the matcher moves the points in fp32: a few ulp of the coordinates on top of the margin)
    for (int a = 0; a < 3; ++a) {
        const double slack = 1e-5 * std::max(std::fabs(lo[a]), std::fabs(hi[a])) + 1e-6;
        lo[a] -= margin + slack;
        hi[a] += margin + slack;
    }
    return MOLA_ICP_OK;
}

// The part of the map inside [lo, hi] becomes this rank's map (C5: 10M points are not uploaded-sorted-held eight times
// over; a rank keeps what its shard can reach).  Points keep their ORIGINAL indices in every pairing that leaves the
// library.  Exactness: a pairing inside the gate can only involve map points within `gate` of a moved query, so the
// result equals the full map's as long as shard_reach_box(T, gate) stays inside the box -- match() checks that at every
// pose and fails with MOLA_ICP_E_BADARG otherwise (the caller re-cuts with a larger margin).
int HipWorkspace::set_map_slab(const float* x, const float* y, const float* z, size_t M, const double lo[3], const double hi[3],
                               bool on_device, size_t* n_kept)
{
    int rc = init();
    if (rc) return rc;
    if (M && (!x || !y || !z)) return fail(MOLA_ICP_E_BADARG, "null map pointer");
    if (M > (size_t)0x7fff0000) return fail(MOLA_ICP_E_BADARG, "map too large for 32-bit indices");
    if (!lo || !hi) return fail(MOLA_ICP_E_BADARG, "null box");
    HIPCHK(hipSetDevice(device_));
    const float *fx = x, *fy = y, *fz = z;
    if (!on_device) {
        if ((rc = upload_soa(stage_in_, stream_, x, y, z, M, &fx, &fy, &fz))) return rc;
        HIPCHK(hipStreamSynchronize(stream_));
    }
    // box in fp32, rounded outwards
    float flo[3], fhi[3];
    for (int a = 0; a < 3; ++a) {
        flo[a] = (float)lo[a]; if ((double)flo[a] > lo[a]) flo[a] = std::nextafter(flo[a], -INFINITY);
        fhi[a] = (float)hi[a]; if ((double)fhi[a] < hi[a]) fhi[a] = std::nextafter(fhi[a], INFINITY);
    }
    if ((rc = slab_orig_.reserve(sizeof(int) * (M ? M : 1)))) return rc;
    size_t kept = 0;
    if ((rc = select_in_box(stream_, fx, fy, fz, M, flo, fhi, sort_scratch_, slab_orig_.as<int>(), &kept))) return rc;
    const size_t kp = (kept + 63) / 64 * 64;
    if ((rc = map_own_.reserve(sizeof(float) * 3 * (kp ? kp : 64)))) return rc;
    float* own = map_own_.as<float>();
    if ((rc = gather_by_index(stream_, fx, fy, fz, slab_orig_.as<int>(), kept, own, own + kp, own + 2 * kp))) return rc;
    HIPCHK(hipStreamSynchronize(stream_));
    stage_in_.release();  // the full map was transient
    if ((rc = set_map_device(own, own + kp, own + 2 * kp, kept))) return rc;
    slab_active_ = true;
    for (int a = 0; a < 3; ++a) { slab_lo_[a] = (double)flo[a]; slab_hi_[a] = (double)fhi[a]; }
    if (n_kept) *n_kept = kept;
    return MOLA_ICP_OK;
}

constexpr int kMfmaQT = 8;             // 128 queries per wave
constexpr int kSegTilesTarget = 8192;  // 131072 map points = 2 MiB of image per segment: L2-resident per XCD

// Builds the derived map image for the MFMA matcher: bounding box -> centre/radius, then
// [tile][4][16] fp32 rows padded to whole LDS chunks.  Once per map.
int HipWorkspace::prepare_map()
{
    if (map_img_valid_) return MOLA_ICP_OK;
    int rc;
    const int M = (int)M_;
    const int nb = 256;
    if ((rc = map_meta_.reserve(sizeof(float) * (6 * nb + 8)))) return rc;
    float* part = map_meta_.as<float>();
    float* bbox = part + 6 * nb;
    hipLaunchKernelGGL(k_bbox_partial, dim3(nb), dim3(256), 0, stream_, gx_, gy_, gz_, M, part);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_bbox_final, dim3(1), dim3(256), 0, stream_, part, nb, bbox);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(meta_host_, bbox, sizeof(float) * 6, hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    double r2 = 0;
    for (int k = 0; k < 3; ++k) {
        const float lo = meta_host_[k], hi = meta_host_[3 + k];
        if (!std::isfinite(lo) || !std::isfinite(hi))
            return fail(MOLA_ICP_E_BADARG, "the map has non-finite coordinates");
        const float c = 0.5f * lo + 0.5f * hi;
        map_center_[k] = c;
        const double h = std::fmax((double)hi - (double)c, (double)c - (double)lo);
        r2 += h * h;
    }
    map_radius_ = (float)(std::sqrt(r2) * 1.00001) + 1e-30f;  // >= max |fl(m - c)|
    if (!(map_radius_ < 1e15f)) return fail(MOLA_ICP_E_BADARG, "map extent too large for the fp32 MFMA filter");
    // tiles of 16 points, rounded up to whole 4-tile steps, plus 4 tiles of prefetch slack
    map_tiles_ = (int)(((M_ + 15) / 16 + 3) / 4 * 4);
    map_segs_ = (map_tiles_ + kSegTilesTarget - 1) / kSegTilesTarget;
    if (map_segs_ > 64) map_segs_ = 64;
    map_seg_tiles_ = ((map_tiles_ + map_segs_ - 1) / map_segs_ + 3) / 4 * 4;
    map_segs_ = (map_tiles_ + map_seg_tiles_ - 1) / map_seg_tiles_;
    const size_t padded = (size_t)(map_tiles_ + 4) * 16;
    if ((rc = map_img_.reserve(sizeof(float) * 4 * padded))) return rc;
    MapFrame F{map_center_[0], map_center_[1], map_center_[2], map_radius_};
    hipLaunchKernelGGL(k_map_image, dim3((unsigned)((padded + 255) / 256)), dim3(256), 0, stream_, gx_, gy_, gz_, M,
                       (int)padded, F, map_img_.as<float>());
    HIPCHK(hipGetLastError());
    map_img_valid_ = true;
    return MOLA_ICP_OK;
}

// Bounding box of a device cloud into the device block bbox_dev() and, by an asynchronous copy, into slot `slot` of the pinned block:
// nothing waits here.  The host looks at its copy (finite coordinates?) at the next wait it makes anyway (check_bboxes(): behind
// the first accumulation of an align, or behind a cache build's synchronisation).  `owner`: the prepared-cloud object that must
// not count as prepared if the box turns out not to be finite (may be null).
int HipWorkspace::bbox_async(const float* x, const float* y, const float* z, size_t n, int slot, const std::shared_ptr<SortedCloud>& owner)
{
    int rc;
    if ((rc = bbox_rows_async(x, y, z, n, slot, owner))) return rc;
    hipLaunchKernelGGL(k_bbox_final, dim3(1), dim3(256), 0, stream_, map_meta_.as<float>(), bbox_n_rows_, bbox_dev());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(meta_host_ + 8 * slot, bbox_dev(), sizeof(float) * 6, hipMemcpyDeviceToHost, stream_));
    return MOLA_ICP_OK;
}

// The prepare chain's form: only the per-block rows (map_meta_: [kBboxRows][6]); the sort's first kernel finishes the box itself
// and writes it to bbox_dev() and to the pinned slot (map_sort.hip, HilbertKeys) -- no launch and no copy in between.
int HipWorkspace::bbox_rows_async(const float* x, const float* y, const float* z, size_t n, int slot, const std::shared_ptr<SortedCloud>& owner,
                                  hipStream_t st, DevBuf* meta)
{
    int rc;
    if (!st) st = stream_;
    if (!meta) meta = &map_meta_;
    if ((rc = meta->reserve(sizeof(float) * (6 * kBboxRows + 8)))) return rc;
    // (1024 points per workgroup and trip, the four loads of a thread issued together: a 120k-point scan is one trip of 118 workgroups)
    bbox_n_rows_ = (int)std::min<size_t>((size_t)kBboxRows, (n + 1023) / 1024);
    if (bbox_n_rows_ < 1) bbox_n_rows_ = 1;
    hipLaunchKernelGGL(k_bbox_rows, dim3(bbox_n_rows_), dim3(256), 0, st, x, y, z, (int)n, meta->as<float>());
    HIPCHK(hipGetLastError());
    bbox_pending_ |= 1u << slot;
    bbox_owner_[slot] = owner;
    return MOLA_ICP_OK;
}

// after a wait that covers everything enqueued on stream_ so far
int HipWorkspace::check_bboxes()
{
    const unsigned int pending = bbox_pending_;
    bbox_pending_ = 0;
    int rc = MOLA_ICP_OK;
    for (int slot = 0; slot < 2; ++slot) {
        const std::shared_ptr<SortedCloud> owner = bbox_owner_[slot].lock();   // (gone: nothing to mark)
        bbox_owner_[slot].reset();
        if (!(pending & (1u << slot))) continue;
        for (int k = 0; k < 6; ++k)
            if (!std::isfinite(meta_host_[8 * slot + k])) {
                // (not "prepared": another align on the same resident clouds must run into the same refusal -- the cloud the box
                // belongs to, whichever role it was prepared in)
                if (owner) owner->ready = false;
                if (!rc) rc = fail(MOLA_ICP_E_BADARG, "a cloud has non-finite coordinates");
                break;
            }
    }
    return rc;
}

int HipWorkspace::bbox_of(const float* x, const float* y, const float* z, size_t n, float out[6])
{
    int rc = bbox_async(x, y, z, n, 0, std::shared_ptr<SortedCloud>());
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(stream_));
    if ((rc = check_bboxes())) return rc;
    for (int k = 0; k < 6; ++k) out[k] = meta_host_[k];
    return MOLA_ICP_OK;
}

// Once per map: Hilbert order, tile boxes, super-tile boxes, top boxes -- five launches enqueued back to back (map_sort.hip).
int HipWorkspace::prepare_tiles()
{
    if (map_sc_->ready) return MOLA_ICP_OK;
    int rc;
    SortedCloud& sc = *map_sc_;
    if ((rc = bbox_rows_async(gx_, gy_, gz_, M_, 0, map_sc_))) return rc;
    const size_t super_pts = (size_t)kTileG * kSuper;
    sc.n_super = (int)((M_ + super_pts - 1) / super_pts);
    sc.n_super = (sc.n_super + 63) / 64 * 64;  // whole top boxes (the padding tiles get empty boxes)
    sc.n_top = sc.n_super / 64;
    sc.n_tiles_p = sc.n_super * kSuper;
    sc.padded = (size_t)sc.n_tiles_p * kTileG;
    if ((rc = sc.sorted.reserve(sizeof(float) * 3 * sc.padded))) return rc;
    if ((rc = sc.perm.reserve(sizeof(int) * sc.padded))) return rc;
    if ((rc = sc.tbox.reserve(sizeof(float) * 6 * (size_t)sc.n_tiles_p))) return rc;
    if ((rc = sc.sbox.reserve(sizeof(float) * 6 * (size_t)sc.n_super))) return rc;
    if ((rc = sc.ubox.reserve(sizeof(float) * 6 * (size_t)sc.n_top))) return rc;
    if ((rc = sc.keys.reserve(sizeof(unsigned int) * (M_ ? M_ : 1)))) return rc;
    if ((rc = sc.box.reserve(sizeof(float) * 8))) return rc;
    if ((rc = hilbert_sort_points(stream_, gx_, gy_, gz_, M_, sc.padded, map_meta_.as<float>(), bbox_n_rows_, sc.box.as<float>(), meta_host_, sort_scratch_,
                                  sc.sorted.as<float>(), sc.perm.as<int>(), sc.sbox.as<float>(), sc.n_super, sc.ubox.as<float>(), sc.n_top,
                                  sc.keys.as<unsigned int>())))
        return rc;
    if ((rc = boxes_of_sorted(stream_, sc.sorted.as<float>(), M_, sc.padded, sc.n_tiles_p, sc.n_super, sc.n_top, sc.tbox.as<float>(),
                              sc.sbox.as<float>(), sc.ubox.as<float>(), sc.box.as<float>())))
        return rc;
    sc.ready = true;
    return MOLA_ICP_OK;
}

// Once per local cloud: Hilbert order of the queries (a rigid motion keeps them compact) -- four launches.
int HipWorkspace::prepare_queries(bool aside)
{
    if (loc_sc_->ready) return MOLA_ICP_OK;
    int rc;
    SortedCloud& sc = *loc_sc_;
    // aside: on the second stream, with rows / scratch of its own -- next to the map's chain (prepare_both)
    const hipStream_t st = aside ? aux_stream_ : stream_;
    DevBuf& meta = aside ? loc_meta_ : map_meta_;
    DevBuf& scratch = aside ? sort_scratch_loc_ : sort_scratch_;
    if ((rc = bbox_rows_async(lx_, ly_, lz_, N_, 1, loc_sc_, st, &meta))) return rc;
    const int n_rows = bbox_n_rows_;
    sc.padded = (N_ + kQPW - 1) / kQPW * kQPW;
    if ((rc = sc.sorted.reserve(sizeof(float) * 3 * sc.padded))) return rc;
    if ((rc = sc.perm.reserve(sizeof(int) * sc.padded))) return rc;
    if ((rc = hilbert_sort_points(st, lx_, ly_, lz_, N_, sc.padded, meta.as<float>(), n_rows, meta.as<float>() + 6 * kBboxRows, meta_host_ + 8, scratch,
                                  sc.sorted.as<float>(), sc.perm.as<int>(), nullptr, 0, nullptr, 0, nullptr)))
        return rc;
    sc.ready = true;
    return MOLA_ICP_OK;
}

// Both clouds new (mola_icp_align from host buffers: two uploads, two chains of six small launches each): the chains are independent
// and neither fills the device -- the queries' chain runs on the second stream beside the map's, the matcher waits for both.
int HipWorkspace::prepare_both()
{
    int rc;
    if (map_sc_->ready || loc_sc_->ready || g_knobs.no_side_prepare || N_ == 0 || M_ == 0) {
        if ((rc = prepare_tiles())) return rc;
        return prepare_queries(false);
    }
    HIPCHK(hipEventRecord(ev_order_a_, stream_));              // (behind the uploads)
    HIPCHK(hipStreamWaitEvent(aux_stream_, ev_order_a_, 0));
    if ((rc = prepare_queries(true))) return rc;
    HIPCHK(hipEventRecord(ev_prep_, aux_stream_));
    rc = prepare_tiles();
    HIPCHK(hipStreamWaitEvent(stream_, ev_prep_, 0));          // (also on a failure above: nothing of the side chain may outrun the main stream's waits)
    return rc;
}

int voxel_downsample_device(hipStream_t stream, const float* x, const float* y, const float* z, size_t n, const float bbox[6],
                            float voxel, DevBuf& scratch, float* out_x, float* out_y, float* out_z, size_t capacity,
                            size_t* n_out_host);

// row f4: voxel-grid downsample of a host cloud -> host cloud (centroid per occupied voxel, ascending voxel key)
int HipWorkspace::voxel_downsample(const float* x, const float* y, const float* z, size_t n, double voxel_size,
                                   float* out_x, float* out_y, float* out_z, size_t capacity, size_t* n_out)
{
    int rc = init();
    if (rc) return rc;
    if (!n_out) return fail(MOLA_ICP_E_BADARG, "null n_out");
    *n_out = 0;
    if (!(voxel_size > 0) || !std::isfinite(voxel_size)) return fail(MOLA_ICP_E_BADARG, "voxel size must be > 0");
    if (n && (!x || !y || !z)) return fail(MOLA_ICP_E_BADARG, "null cloud pointer");
    if (n > (size_t)0x7fff0000) return fail(MOLA_ICP_E_BADARG, "cloud too large for 32-bit indices");
    if (n == 0) return MOLA_ICP_OK;
    HIPCHK(hipSetDevice(device_));
    DevBuf in, out;
    const float *dx, *dy, *dz;
    if ((rc = upload_soa(in, stream_, x, y, z, n, &dx, &dy, &dz))) return rc;
    float bbox[6];
    if ((rc = bbox_of(dx, dy, dz, n, bbox))) { in.release(); return rc; }
    if ((rc = out.reserve(sizeof(float) * 3 * n))) { in.release(); return rc; }
    float* o = out.as<float>();
    size_t nv = 0;
    rc = voxel_downsample_device(stream_, dx, dy, dz, n, bbox, (float)voxel_size, sort_scratch_, o, o + n, o + 2 * n, n, &nv);
    if (!rc) {
        *n_out = nv;
        const size_t m = nv < capacity ? nv : capacity;
        if (m && out_x && out_y && out_z) {
            hipError_t e = hipMemcpyAsync(out_x, o, sizeof(float) * m, hipMemcpyDeviceToHost, stream_);
            if (e == hipSuccess) e = hipMemcpyAsync(out_y, o + n, sizeof(float) * m, hipMemcpyDeviceToHost, stream_);
            if (e == hipSuccess) e = hipMemcpyAsync(out_z, o + 2 * n, sizeof(float) * m, hipMemcpyDeviceToHost, stream_);
            if (e == hipSuccess) e = hipStreamSynchronize(stream_);
            if (e != hipSuccess) rc = fail(MOLA_ICP_E_HIP, std::string("voxel_downsample copy: ") + hipGetErrorString(e));
        }
    }
    in.release();
    out.release();
    return rc;
}

// ---- row f4: device-resident cloud cache --------------------------------------------------------------
// A cached cloud lives in HBM in raw AND Hilbert-sorted form with its tile boxes, so it can serve as the map
// (`from`) or as the local cloud (`to`) of any later align without upload or sort: the reference keeps
// keyframe clouds in the world model and re-reads them per nearby-KF / loop-closure ICP
// (src/LidarOdometry.cpp:384-388, 658-666), and in odometry each scan is `to` once and `from` once (cpp:278-279).
// Waits for the stream the way the accumulator hand-over does -- by looking -- for work that is known to be tens of microseconds
// long (a cloud's prepare chain): hipStreamSynchronize's wake-up alone cost 8-10 us between the chain's last kernel and the align's
// first launch.  Falls back to the blocking wait after ~0.3 ms of looking (large clouds, a busy device).
hipError_t HipWorkspace::quick_sync()
{
    for (int spins = 0; spins < 4000; ++spins) {
        const hipError_t e = hipStreamQuery(stream_);
        if (e != hipErrorNotReady) return e;
        for (int k = 0; k < 8; ++k) __builtin_ia32_pause();
    }
    return hipStreamSynchronize(stream_);
}

int HipWorkspace::build_cached(SortedCloud& sc, const float* x, const float* y, const float* z, size_t n)
{
    std::shared_ptr<SortedCloud> alias(&sc, [](SortedCloud*) {});   // (the caller owns it)
    return build_cached(alias, x, y, z, n, true);
}

// wait = false: the chain is only ENQUEUED -- the caller goes on to use the cloud on THIS workspace (same stream: ordered) and calls
// finish_build() before anything else may touch it; the box is looked at by the first wait the host makes (spin_for / finish_build),
// which needs `scp` to be the owning pointer (the box is recorded against it).
int HipWorkspace::build_cached(const std::shared_ptr<SortedCloud>& scp, const float* x, const float* y, const float* z, size_t n, bool wait)
{
    int rc = init();
    if (rc) return rc;
    if (n && (!x || !y || !z)) return fail(MOLA_ICP_E_BADARG, "null cloud pointer");
    if (n > (size_t)0x7fff0000) return fail(MOLA_ICP_E_BADARG, "cloud too large for 32-bit indices");
    HIPCHK(hipSetDevice(device_));
    SortedCloud& sc = *scp;
    if ((rc = upload_soa(sc.raw, stream_, x, y, z, n, &sc.x, &sc.y, &sc.z))) return rc;
    sc.n = n;
    sc.cached = true;
    // build the sorted form through the map-role path of this workspace, then hand the buffers over
    std::shared_ptr<SortedCloud> keep_sc = map_sc_;
    const float *kx = gx_, *ky = gy_, *kz = gz_;
    const size_t kM = M_;
    map_sc_ = scp;
    gx_ = sc.x; gy_ = sc.y; gz_ = sc.z; M_ = n;
    sc.ready = false;
    rc = n ? prepare_tiles() : MOLA_ICP_OK;
    // (the workspace's own map is put back BEFORE any return)
    // (on a failure half-way the caller destroys the cloud, and its parked blocks may be handed to another handle at once --
    // nothing of this build may still be running on them: wait)
    hipError_t es = hipSuccess;
    if (wait || rc) es = quick_sync();
    map_sc_ = keep_sc;
    gx_ = kx; gy_ = ky; gz_ = kz; M_ = kM;
    if (es != hipSuccess && !rc) return fail(es == hipErrorOutOfMemory ? MOLA_ICP_E_OOM : MOLA_ICP_E_HIP, std::string("build_cached: ") + hipGetErrorString(es));
    if (!wait && !rc) return MOLA_ICP_OK;
    // (the cloud's bounding box arrived with that synchronisation; a box that is not finite clears `ready` on THIS cloud -- the
    // owner recorded with the box -- not on whichever cloud sits in the workspace's map role)
    if (!rc) rc = check_bboxes();
    else { bbox_pending_ = 0; bbox_owner_[0].reset(); bbox_owner_[1].reset(); }
    if (rc) sc.ready = false;
    return rc;
}

// the end of a build_cached(..., wait = false): nothing of the chain in flight any more, the box looked at (if no wait did yet)
int HipWorkspace::finish_build(SortedCloud& sc)
{
    const hipError_t es = quick_sync();
    if (es != hipSuccess) { sc.ready = false; return fail(MOLA_ICP_E_HIP, std::string("finish_build: ") + hipGetErrorString(es)); }
    const int rc = bbox_pending_ ? check_bboxes() : MOLA_ICP_OK;
    if (rc) sc.ready = false;
    return rc;
}

void HipWorkspace::use_cached_map(const std::shared_ptr<SortedCloud>& sc)
{
    map_sc_ = sc;
    gx_ = sc->x; gy_ = sc->y; gz_ = sc->z;
    M_ = sc->n;
    planes_valid_ = false;
    map_img_valid_ = false;
    pairing_valid_ = false;
    seed_valid_ = false;
    knn_seed_valid_ = false;
}

void HipWorkspace::use_cached_local(const std::shared_ptr<SortedCloud>& sc)
{
    loc_sc_ = sc;
    lx_ = sc->x; ly_ = sc->y; lz_ = sc->z;
    N_ = sc->n;
    planes_valid_ = false;
    cost_valid_ = false;
    order_valid_ = false;
    knn_cost_valid_ = false;
    knn_order_valid_ = false;
    knn_seed_valid_ = false;
    pairing_valid_ = false;
    seed_valid_ = false;
    knn_seed_valid_ = false;
}

// What an align LEAVES for the next one on the same clouds is dropped: the pairing (next launch's seeds), the neighbour lists, their
// certificates and the plane cache -- results of matches at poses of that align.  What is kept is what belongs to the CLOUDS: their
// prepared (sorted) form, and the per-item cost order of the work queue -- how dear each stretch of the scan's Hilbert order is
// against this map is a property of the two clouds (their densities along the curve), it schedules the items and can never change a
// result; like the prepared form it is made once per cloud pair and used by every align on it (`everything` drops it too: the very
// first align on a pair).
void HipWorkspace::forget_warm_start(bool everything)
{
    planes_valid_ = false;
    knn_seed_valid_ = false;
    pairing_valid_ = false;
    seed_valid_ = false;
    if (everything) {
        cost_valid_ = false;
        order_valid_ = false;
        knn_cost_valid_ = false;
        knn_order_valid_ = false;
    }
}

int HipWorkspace::order_begin()
{
    HIPCHK(hipEventRecord(ev_order_a_, stream_));
    HIPCHK(hipStreamWaitEvent(aux_stream_, ev_order_a_, 0));
    return MOLA_ICP_OK;
}
int HipWorkspace::order_end()
{
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ev_order_b_, aux_stream_));
    order_pending_ = true;
    return MOLA_ICP_OK;
}
int HipWorkspace::order_join()
{
    if (order_pending_) {
        HIPCHK(hipStreamWaitEvent(stream_, ev_order_b_, 0));
        order_pending_ = false;
    }
    return MOLA_ICP_OK;
}

TiledMap HipWorkspace::tiled_map() const
{
    const float* sx = map_sc_->sorted.as<float>();
    return TiledMap{sx, sx + map_sc_->padded, sx + 2 * map_sc_->padded, map_sc_->perm.as<int>(), map_sc_->tbox.as<float>(), map_sc_->n_tiles_p,
                    map_sc_->sbox.as<float>(), map_sc_->n_super, map_sc_->ubox.as<float>(), map_sc_->n_top};
}

int HipWorkspace::launch_tiled(const PoseF& P, float thr2, bool use_seed, unsigned int* counter)
{
    // Items of 64 queries (one per lane; the packed math pairs map points), four workgroups per CU: measured at every size
    // once the end-of-wave statistics atomics were out of the way (they had hidden it) -- 1M x 1M: 64-query items 4 / 3 per
    // CU 105-108 / 112-114 us, 128-query items (heavy ones listed as halves) 123-125 us; 2M x 2M 188 / 206 / 214 us;
    // 1M vs a 10M-point map 277 / 293 / 413 us.  Smaller items evaluate fewer pairs per query (309 vs 420: a tile staged
    // for a wave is evaluated by ALL its queries) and balance better (5 per wave); the 118-VGPR one-query body fits four
    // waves per SIMD (five with spills: no gain).  MOLA_ICP_QPL=2 / MOLA_ICP_BLOCKS_PER_CU bring the older form back.
    int qpl = g_knobs.qpl ? g_knobs.qpl : 1;
    int per_cu = g_knobs.blocks_per_cu > 0 ? g_knobs.blocks_per_cu : (qpl == 1 ? 4 : 3);  // tuning knob
    const TiledMap mp = tiled_map();
    const size_t box_bytes = sizeof(float) * 6u * ((size_t)mp.n_top + (size_t)mp.n_super);
    const int n_items = (int)((N_ + (size_t)(64 * qpl) - 1) / (size_t)(64 * qpl));
    // The quad flavour (kernels_tiled.hpp: quad_sweep -- per-quad tile lists, four tiles per round straight into LDS): every lane meets
    // only the tiles its 16-query quad reaches.  Same lease, ms per iteration plain / quads: 1M x 1M 0.1168-0.1202 / 0.1110-0.1170,
    // 1M x 3M 0.173 / 0.167, 500k x 5M 0.167 / 0.141, 250k x 2.5M 0.109 / 0.100, 1M x 10M 0.285 / 0.234 (profiles/r05/quad_sweep_ab.txt).
    // The product build's default for seeded launches; the diagnostic builds keep the pass-by-pass sweep and its clocks.  MOLA_ICP_QUADS=0|1 forces one.
    const bool diag_build = (dbg_stats_ && !wave_times_) || wave_times_ != nullptr;
    // (a launch without seeds keeps the pass-by-pass sweep unless the knob forces the other: its bounds start at the gate and tighten from
    //  pass to pass, the quad sweep would test every tile of an item against the gate -- one launch, 1M queries x 10M points: 0.68 / 0.81 ms)
    const bool quads = qpl == 1 && !diag_build && (g_knobs.quads >= 0 ? g_knobs.quads != 0 : use_seed);
    // the upper box levels in LDS, else read from global memory.  The quad flavour's own 17.4 KB leave 22 KB per workgroup at four
    // workgroups per CU, and the fourth workgroup is worth more than the LDS copy: ms per iteration with a 40-KB / 22-KB limit,
    // 1M x 2M 0.143-0.149 / 0.125-0.128, 1M x 3M 0.168-0.177 / 0.143-0.148, 2M x 2M 0.214-0.222 / 0.188-0.189 (1M x 1.5M, 17 KB: the same)
    const size_t lds_box_limit = quads ? (size_t)g_knobs.quad_lds_boxes_kb * 1024 : kMaxLdsBoxBytes;
    const int lds_boxes = box_bytes <= lds_box_limit ? 1 : 0;
    const size_t dyn_lds = lds_boxes ? box_bytes : 0;
    {   // persistent waves with a static first item: every block of the grid must be resident from the start
        // (the query is a runtime call of tens of microseconds on the launch path: once per kernel flavour and LDS size)
        const int slot = qpl == 2 ? 0 : (quads ? 2 : 1);
        int& fit = fit_cache_[slot];
        size_t& fit_lds = fit_cache_lds_[slot];
        if (fit == 0 || fit_lds != dyn_lds) {
            if (qpl == 2) HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, (k_nn_tiled<2, true>), 256, dyn_lds));
            else if (quads) HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, (k_nn_tiled<1, false, true>), 256, dyn_lds));
            else HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, (k_nn_tiled<1, true>), 256, dyn_lds));   // (the diagnostic build: the larger one)
            fit_lds = dyn_lds;
        }
        if (fit >= 1 && per_cu > fit) per_cu = fit;
    }
    int grid = num_cus_ * per_cu;
    if (grid > (n_items + 3) / 4) grid = (n_items + 3) / 4;
    int rc;
    // 128-query items keep two cost slots each (the item, or its two halves) and up to two list entries
    const size_t cost_slots = qpl == 2 ? 2 * (size_t)n_items : (size_t)n_items;
    {
        const void* before = item_cost_.p;
        if ((rc = item_cost_.reserve(sizeof(unsigned int) * cost_slots))) return rc;
        if (item_cost_.p != before) cost_valid_ = false;
    }
    if ((rc = item_order_.reserve(sizeof(int) * (cost_slots + kQueues + 2)))) return rc;  // + the range boundaries + the entry count
    const int* order = nullptr;
    if ((rc = order_join())) return rc;   // a re-sort launched behind the last matcher launch (below)
    if (cost_valid_ && !g_knobs.no_lpt) {
        // the cost profile drifts slowly with the pose: re-sort at launch 1, 2, 4, 8 after the clouds were set,
        // then every 16th; the order is reused in between
        if (!order_valid_ || launches_since_order_ >= plan_interval_) {
            if (qpl == 2)
                hipLaunchKernelGGL(k_order_entries, dim3(1), dim3(1024), 0, stream_, item_cost_.as<unsigned int>(), n_items,
                                   g_knobs.no_split ? 1 : (int)((double)(grid * 4) * 0.55 / g_knobs.split_share), item_order_.as<int>());
            else
                hipLaunchKernelGGL(k_order_items, dim3(1), dim3(1024), 0, stream_, item_cost_.as<unsigned int>(), n_items,
                                   item_order_.as<int>());
            HIPCHK(hipGetLastError());
            plan_interval_ = order_valid_ ? (plan_interval_ < 16 ? plan_interval_ * 2 : 16) : 1;
            order_valid_ = true;
            launches_since_order_ = 0;
        }
        ++launches_since_order_;
        order = item_order_.as<int>();
    }
    if ((rc = ts_pos_.reserve(sizeof(int) * loc_sc_->padded))) return rc;
    if ((rc = ts_idx_.reserve(sizeof(int) * loc_sc_->padded))) return rc;
    if ((rc = ts_d2_.reserve(sizeof(float) * loc_sc_->padded))) return rc;
    if ((rc = ts_gs_.reserve(sizeof(float) * 3 * loc_sc_->padded))) return rc;
    float* gs = ts_gs_.as<float>();
    const size_t gs_n = loc_sc_->padded;
    const float* sl = loc_sc_->sorted.as<float>();
    unsigned long long* staged = profiling_ ? stats_.as<unsigned long long>() : nullptr;  // evaluated pairs, slotted (statistics only)
    // counter[2] = redo count; tq = the fast pass's queue counters, tq + kQueues * kQueueStride the exact pass's
    unsigned int* tq = reinterpret_cast<unsigned int*>(acc_dev_.as<double>() + kNAcc + 8);
    unsigned long long* dbg = dbg_stats_ && !wave_times_ ? dbg_stats_ : nullptr;
    // 64-query items: the unit-weight sums of every item are formed in the matcher's epilogue (item_row_mfma) -- the first
    // accumulation pass of the iteration is then a row reduction, no second pass over the pairing
    double* item_rows = nullptr;
    if (qpl == 1 && !g_knobs.no_fused_rows) {
        if ((rc = rows_.reserve(sizeof(double) * kNAcc * (size_t)n_items))) return rc;
        item_rows = rows_.as<double>();
    }
    // one launch: an entry that meets an exact distance tie (duplicate points, lattices) is redone by its wave with the exact-key sweep
#define MOLA_LAUNCH_TILED(QPL, DIAG) MOLA_LAUNCH_TILED_Q(QPL, DIAG, false)
#define MOLA_LAUNCH_TILED_Q(QPL, DIAG, QUADS)                                                                         \
    do {                                                                                                              \
        hipLaunchKernelGGL((k_nn_tiled<QPL, DIAG, QUADS>), dim3(grid), dim3(256), dyn_lds, stream_, sl, sl + loc_sc_->padded, \
                           sl + 2 * loc_sc_->padded, (int)N_, mp, P, thr2, use_seed ? 1 : 0, ts_pos_.as<int>(),        \
                           ts_idx_.as<int>(), ts_d2_.as<float>(), gs, gs + gs_n, gs + 2 * gs_n, order, item_cost_.as<unsigned int>(), tq, \
                           staged, dbg, lds_boxes, wave_times_, g_knobs.early_pop ? 1 : 0, item_rows);                 \
        HIPCHK(hipGetLastError());                                                                                    \
    } while (0)
    const bool diag = dbg != nullptr || wave_times_ != nullptr;
    if (qpl == 2) { if (diag) MOLA_LAUNCH_TILED(2, true); else MOLA_LAUNCH_TILED(2, false); }
    else { if (diag) MOLA_LAUNCH_TILED(1, true); else if (quads) MOLA_LAUNCH_TILED_Q(1, false, true); else MOLA_LAUNCH_TILED(1, false); }
#undef MOLA_LAUNCH_TILED
#undef MOLA_LAUNCH_TILED_Q
    cost_valid_ = true;
    // the NEXT launch's re-sort, if one is due: now, on the side stream -- it reads this launch's costs and runs beside the
    // accumulation and the host's turn-around instead of in front of the next matcher launch
    if (!g_knobs.no_lpt && (!order_valid_ || launches_since_order_ >= plan_interval_)) {
        if ((rc = order_begin())) return rc;
        if (qpl == 2)
            hipLaunchKernelGGL(k_order_entries, dim3(1), dim3(1024), 0, aux_stream_, item_cost_.as<unsigned int>(), n_items,
                               g_knobs.no_split ? 1 : (int)((double)(grid * 4) * 0.55 / g_knobs.split_share), item_order_.as<int>());
        else
            hipLaunchKernelGGL(k_order_items, dim3(1), dim3(1024), 0, aux_stream_, item_cost_.as<unsigned int>(), n_items,
                               item_order_.as<int>());
        if ((rc = order_end())) return rc;
        plan_interval_ = order_valid_ ? (plan_interval_ < 16 ? plan_interval_ * 2 : 16) : 1;
        order_valid_ = true;
        launches_since_order_ = 0;
    }
    rows_valid_ = item_rows != nullptr;
    rows_count_ = n_items;
    HIPCHK(hipGetLastError());
    return MOLA_ICP_OK;
}

// One NN problem of a cooperative / batched launch on this workspace's clouds and pairing buffers.
int HipWorkspace::fill_nn_problem(const PoseF& P, float thr2, bool use_seed, NnProblem& pb)
{
    int rc;
    if ((rc = ts_pos_.reserve(sizeof(int) * loc_sc_->padded))) return rc;
    if ((rc = ts_idx_.reserve(sizeof(int) * loc_sc_->padded))) return rc;
    if ((rc = ts_d2_.reserve(sizeof(float) * loc_sc_->padded))) return rc;
    if ((rc = ts_gs_.reserve(sizeof(float) * 3 * loc_sc_->padded))) return rc;
    const size_t n_rows = (N_ + 63) / 64;   // one row per 64 queries
    if ((rc = rows_.reserve(sizeof(double) * kNAcc * n_rows))) return rc;
    const float* sl = loc_sc_->sorted.as<float>();
    pb.slx = sl; pb.sly = sl + loc_sc_->padded; pb.slz = sl + 2 * loc_sc_->padded;
    pb.N = (int)N_;
    pb.mp = tiled_map();
    pb.P = P;
    pb.thr2 = thr2;
    pb.use_seed = use_seed ? 1 : 0;
    pb.pos_s = ts_pos_.as<int>(); pb.idx_s = ts_idx_.as<int>(); pb.d2_s = ts_d2_.as<float>();
    pb.gsx = ts_gs_.as<float>(); pb.gsy = pb.gsx + loc_sc_->padded; pb.gsz = pb.gsx + 2 * loc_sc_->padded;
    pb.rows = rows_.as<double>();
    pb.staged = stats_.as<unsigned long long>();
    return MOLA_ICP_OK;
}

// Small clouds: one workgroup per 128-query item (k_nn_coop): pairing + the item's row of unit-weight sums, one launch.
int HipWorkspace::launch_coop(const PoseF& P, float thr2, bool use_seed)
{
    NnBatch<1> b;
    int rc;
    if ((rc = fill_nn_problem(P, thr2, use_seed, b.p[0]))) return rc;
    const TiledMap& mp = b.p[0].mp;
    const size_t box_bytes = sizeof(float) * 6u * ((size_t)mp.n_top + (size_t)mp.n_super);
    const int lds_boxes = box_bytes <= lds_box_limit(nn_coop_static_lds(), 4) ? 1 : 0;   // (k_nn_coop: four workgroups per CU)
    const size_t dyn_lds = lds_boxes ? box_bytes : 0;
    const int n_items = (int)((N_ + kQPW - 1) / kQPW);
    if (wave_times_) hipLaunchKernelGGL((k_nn_coop<1, true>), dim3(xcd_grid(n_items)), dim3(256), dyn_lds, stream_, b, lds_boxes, wave_times_);
    else hipLaunchKernelGGL((k_nn_coop<1, false>), dim3(xcd_grid(n_items)), dim3(256), dyn_lds, stream_, b, lds_boxes, wave_times_);
    HIPCHK(hipGetLastError());
    wave_times_coop_ = true;
    rows_valid_ = true;
    rows_count_ = (int)((N_ + 63) / 64);
    return MOLA_ICP_OK;
}

// Launches with fewer items than wave slots: k_nn_q4 (kernels_q4.hpp) -- one WAVE per 16 queries, four lanes per query, one
// workgroup per row of 64; pairing + the row of unit-weight sums, one launch.
static int q4_lds_boxes(size_t box_bytes)
{
    if (g_knobs.q4_lds_boxes_kb >= 0) return box_bytes <= (size_t)g_knobs.q4_lds_boxes_kb * 1024 ? 1 : 0;
    return box_bytes <= lds_box_limit(q4_static_lds(), q4_workgroups_per_cu()) ? 1 : 0;
}
int HipWorkspace::launch_q4(const PoseF& P, float thr2, bool use_seed)
{
    NnBatch<1> b;
    int rc;
    if ((rc = fill_nn_problem(P, thr2, use_seed, b.p[0]))) return rc;
    const TiledMap& mp = b.p[0].mp;
    const size_t box_bytes = sizeof(float) * 6u * ((size_t)mp.n_top + (size_t)mp.n_super);
    const int lds_boxes = q4_lds_boxes(box_bytes);
    const size_t dyn_lds = lds_boxes ? box_bytes : 0;
    const int n_items = (int)((N_ + 63) / 64);
    HIPCHK(q4_launch(stream_, b, lds_boxes ? n_items : xcd_grid(n_items), dyn_lds, lds_boxes, profiling_ ? 1 : 0));
    rows_valid_ = true;
    rows_count_ = n_items;
    return MOLA_ICP_OK;
}

int HipWorkspace::match_planes(const Mat4& T, const mola_icp_params& p)
{
    if (g_knobs.turn_clock) g_turn.on_enter();
    int rc = init();
    if (rc) return rc;
    HIPCHK(hipSetDevice(device_));
    if (p.knn < 3 || p.knn > 16) return fail(MOLA_ICP_E_UNSUPPORTED, "Matcher_Point2Plane: knn must be in [3, 16] in this build");
    // knn 9 .. 16 (the schema names no bound, params/icp-settings-regular.yaml:37; the shipped files say 6): served by the cooperative
    // kernel alone, at every cloud size -- one instantiation per list length instead of the persistent kernel's four flavours; the first
    // launch seeded by key, the PairedRatio pass by the matcher (no k_quality_from_lists of that length)
    const bool wide_knn = p.knn > 8;
    planes_valid_ = false;
    if ((rc = check_slab(T, p.matcher_threshold))) return rc;
    if (N_ == 0 || M_ == 0) { planes_valid_ = true; planes_empty_ = true; return MOLA_ICP_OK; }
    planes_empty_ = false;
    if ((rc = prepare_both())) return rc;
    if ((rc = planes_.reserve(sizeof(PlanePair) * loc_sc_->padded))) return rc;
    // the stored lists (knn + 1 entries per query: positions, coordinates, original indices; kernels_planes.hpp: KnnSeeds)
    if ((rc = knn_pos_.reserve(knn_seeds_bytes(loc_sc_->padded, (int)p.knn + 1)))) return rc;
    const KnnSeeds seeds = knn_seeds_at(knn_pos_.p, loc_sc_->padded, (int)p.knn + 1);
    if ((rc = knn_lb_.reserve(sizeof(float) * loc_sc_->padded))) return rc;
    if ((rc = plane_cache_.reserve(sizeof(PlanePair) * loc_sc_->padded))) return rc;
    // The first launch of an align has no lists to start from: its sweep begins at the whole gate, and on an odometry-size scan it is
    // the longest kernel of the align by far (269 us of the 660 us of kernels of a scan in the odometry stream).  The caller's
    // guess is a motion model's (src/LidarOdometry.cpp:264-276): the queries sit near their neighbours already.  So: one
    // nearest-neighbour pass first (k_nn_coop, ~35 us at 120k), and every query starts from the knn + 1 map points AROUND its
    // nearest neighbour on the Hilbert curve (k_bootstrap_seeds) -- candidates like any seed (exact: the sweep that follows is
    // complete under the bound they give), no certificates, no cached planes.  Only the first launch: as a general replacement
    // for stale lists the same idea lost everywhere ('measured and dropped').  At 1M x 1M the seeded launch that follows
    // is the dense-build flavour: shipped pipeline 4 770-4 810 -> 4 910 it/s over 20 iterations.
    bool bootstrapped = false;
    {
        const bool have_seed = knn_seed_valid_ && planes_knn_ == (int)p.knn && !g_knobs.no_knn_seed;
        if (!have_seed && !g_knobs.no_bootstrap && !g_knobs.no_knn_seed && M_ >= 64) {
            const bool by_key = (wide_knn || !g_knobs.bootstrap_nn) && map_sc_->keys.p != nullptr && map_sc_->box.p != nullptr;
            if (!by_key && !wide_knn) {
                mola_icp_params pn = p;
                pn.nn_kernel = MOLA_ICP_NN_AUTO;
                // (under HALF the plane matcher's gate: a query whose nearest neighbour is farther than that starts without seeds --
                // the pass costs with its gate, 79 us at 0.7 m on a 120k scan, and seeds that far away bound little; measured 1 / 0.5 /
                // 0.25 of the gate: config 0's first iteration 193-197 / 180 / 201 us, odometry stream 0.65-0.71 / 0.63-0.67 / 0.64-0.67 ms)
                if ((rc = match(T, 0.5 * p.matcher_threshold, pn, nullptr))) return rc;
            }
            if (by_key || (pairing_sorted_ && !wide_knn)) {
                PoseF Pb;
                for (int r = 0; r < 3; ++r) {
                    for (int c = 0; c < 3; ++c) Pb.R[3 * r + c] = (float)T(r, c);
                    Pb.t[r] = (float)T(r, 3);
                }
                const float* slq = loc_sc_->sorted.as<float>();
#define MOLA_LAUNCH_BOOTSTRAP(KK)                                                                                                     \
    do {                                                                                                                              \
        if (by_key)                                                                                                                   \
            hipLaunchKernelGGL((k_bootstrap_seeds<KK, true>), dim3((unsigned)((N_ + 255) / 256)), dim3(256), 0, stream_, (const int*)nullptr, \
                               map_sc_->keys.as<unsigned int>(), map_sc_->box.as<float>(), slq, slq + loc_sc_->padded,              \
                               slq + 2 * loc_sc_->padded, (int)N_, Pb, tiled_map(), (int)M_, seeds);                                \
        else                                                                                                                          \
            hipLaunchKernelGGL((k_bootstrap_seeds<KK, false>), dim3((unsigned)((N_ + 255) / 256)), dim3(256), 0, stream_, ts_pos_.as<int>(), \
                               (const unsigned int*)nullptr, (const float*)nullptr, slq, slq + loc_sc_->padded,                      \
                               slq + 2 * loc_sc_->padded, (int)N_, Pb, tiled_map(), (int)M_, seeds);                                \
    } while (0)
#define MOLA_LAUNCH_BOOTSTRAP_KEY(KK)                                                                                                 \
    hipLaunchKernelGGL((k_bootstrap_seeds<KK, true>), dim3((unsigned)((N_ + 255) / 256)), dim3(256), 0, stream_, (const int*)nullptr,  \
                       map_sc_->keys.as<unsigned int>(), map_sc_->box.as<float>(), slq, slq + loc_sc_->padded,                       \
                       slq + 2 * loc_sc_->padded, (int)N_, Pb, tiled_map(), (int)M_, seeds)
                switch (p.knn) {
                    case 3: MOLA_LAUNCH_BOOTSTRAP(4); break;
                    case 4: MOLA_LAUNCH_BOOTSTRAP(5); break;
                    case 5: MOLA_LAUNCH_BOOTSTRAP(6); break;
                    case 6: MOLA_LAUNCH_BOOTSTRAP(7); break;
                    case 7: MOLA_LAUNCH_BOOTSTRAP(8); break;
                    case 8: MOLA_LAUNCH_BOOTSTRAP(9); break;
                    case 9: MOLA_LAUNCH_BOOTSTRAP_KEY(10); break;
                    case 10: MOLA_LAUNCH_BOOTSTRAP_KEY(11); break;
                    case 11: MOLA_LAUNCH_BOOTSTRAP_KEY(12); break;
                    case 12: MOLA_LAUNCH_BOOTSTRAP_KEY(13); break;
                    case 13: MOLA_LAUNCH_BOOTSTRAP_KEY(14); break;
                    case 14: MOLA_LAUNCH_BOOTSTRAP_KEY(15); break;
                    case 15: MOLA_LAUNCH_BOOTSTRAP_KEY(16); break;
                    default: MOLA_LAUNCH_BOOTSTRAP_KEY(17); break;
                }
#undef MOLA_LAUNCH_BOOTSTRAP_KEY
#undef MOLA_LAUNCH_BOOTSTRAP
                HIPCHK(hipGetLastError());
                bootstrapped = true;
            }
        }
    }
    PoseF P;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) P.R[3 * r + c] = (float)T(r, c);
        P.t[r] = (float)T(r, 3);
    }
    const float thr2 = (float)(p.matcher_threshold * p.matcher_threshold);
    while (ev_.size() < ev_used_ + 2) {
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        ev_.push_back(e);
    }
    unsigned int* counter = reinterpret_cast<unsigned int*>(acc_dev_.as<double>() + kNAcc);
    if (!counters_clean_) {  // (the reductions of both pipelines leave them zero: no memset inside an ICP loop)
        HIPCHK(hipMemsetAsync(counter, 0, 4 * sizeof(unsigned int), stream_));
        HIPCHK(hipMemsetAsync(acc_dev_.as<double>() + kNAcc + 8, 0, sizeof(unsigned int) * 2 * kQueues * kQueueStride, stream_));
    }
    counters_clean_ = false;
    ++nn_launches_;
    if (profiling_) HIPCHK(hipEventRecord(ev_[ev_used_], stream_));
    const float* sl = loc_sc_->sorted.as<float>();
    const TiledMap mp = tiled_map();
    unsigned long long* staged = (profiling_ || g_knobs.debug_stats == 3) ? stats_.as<unsigned long long>() : nullptr;  // evaluated pairs, slotted (statistics only)
    const size_t box_bytes = sizeof(float) * 6u * ((size_t)mp.n_top + (size_t)mp.n_super);
    // Odometry-size clouds: fewer 64-query items than wave slots -- a persistent launch is one item per wave and as long as its
    // slowest item; one WORKGROUP per item instead (k_knn_coop: four waves deal the tiles, lists merged through LDS).
    const int n_items64 = (int)((N_ + 63) / 64);   // (the cooperative kernel's items hold 64 queries whatever MOLA_ICP_QPL says)
    // (crossover, ms per 8-iteration align cooperative / persistent -- uniform synthetic clouds: 60k 0.54 / 0.67, 120k 0.83 / 0.78,
    //  160k 1.02 / 0.84, 200k 1.21 / 0.92; a KITTI-like 120k scan pair, dense near the sensor: 1.76 / 1.96.  Up to 131k queries.)
    // the pose step against the previous launch on these clouds (a heuristic input only -- rotation weighed with a 30 m lever; no result depends on it)
    double step = 0.0;
    {
        double dr = 0.0, dt = 0.0;
        for (int k = 0; k < 9; ++k) { const double d = (double)P.R[k] - (double)knn_last_P_[k]; dr += d * d; }
        for (int k = 0; k < 3; ++k) { const double d = (double)P.t[k] - (double)knn_last_P_[9 + k]; dt += d * d; }
        step = std::sqrt(dt) + 30.0 * std::sqrt(dr);
    }
    // four lanes per query (kernels_knn_q4.hpp) where the launch is as long as an item's chain: near the previous pose.  Far from it the lists change
    // wholesale, a launch is bound by its insertions and k_knn_coop's wider items are the better shape (a KITTI-like 120k pair, us per launch
    // k_knn_q4 / k_knn_coop by step: 2 m 218 / 172, 1 m 156 / 158, 0.3-0.8 m 178-180 / 169-171, 0.13 m 78 / 84, 0.05 m 66 / 68, <= 0.01 m 49-55 / 62-64).
    // Beyond k_knn_coop's range the alternative is the persistent kernel, which k_knn_q4 beats at any step and at every size measured (150k ... 5M
    // queries: knn_q4_lanes_per_query above).
    const bool coop_size = (size_t)n_items64 <= (size_t)num_cus_ * 8;
    const bool knn_q4 = use_knn_q4((int)p.knn + 1) &&
                        (g_knobs.knn_q4 == 1 || (g_knobs.knn_coop != 0 && (coop_size ? step <= kKnnQ4MaxStep : N_ <= kKnnQ4MaxQueries)));
    const bool knn_coop = wide_knn || knn_q4 || (g_knobs.knn_coop >= 0 ? g_knobs.knn_coop != 0 : coop_size);   // one workgroup per item (either kernel)
    // the upper box levels in LDS while that costs the kernel no workgroup per CU (lds_box_limit), else read from global memory
    const int kq4_lpq = knn_q4 ? knn_q4_lanes_per_query((int)p.knn + 1, (size_t)n_items64, num_cus_, coop_size && bootstrapped) : 4;
    const int lds_boxes = box_bytes <= (knn_q4 ? knn_q4_lds_box_limit((int)p.knn + 1, kq4_lpq) : (knn_coop ? knn_coop_lds_box_limit((int)p.knn + 1) : lds_box_limit(persistent_static_lds(), 4))) ? 1 : 0;
    const size_t dyn_lds = lds_boxes ? box_bytes : 0;
    // Queries per lane: ONE (64-query items).  A lane's K-entry lists for two queries push the insertion flavour to 168
    // VGPR + spills; with one query per lane there are none, items are twice as many and half as long -- better balance
    // on large clouds (1M x 1M shipped pipeline 3090 -> 3370 it/s), and where a cloud has no more 128-query items than
    // the launch has waves (an odometry-size pair) the launch, one item long, is nearly halved.  MOLA_ICP_QPL=2 forces
    // 128-query items.
    int ql = 1;
    if (g_knobs.qpl) ql = g_knobs.qpl;
    // persistent waves with a static first item: every block of a grid must be resident from the start -- per flavour
    // (the counting flavour holds ~100 VGPR, the insertion flavour ~140: the first fits four workgroups per CU)
    const int kslot = p.knn < 3 ? 0 : (p.knn > 8 ? 5 : (int)p.knn - 3);
    int& fit_ins = knn_fit_[ql - 1][0][kslot];
    int& fit_ver = knn_fit_[ql - 1][1][kslot];
    int& fit_ins4 = knn_fit_[ql - 1][2][kslot];
    size_t& fit_lds = knn_fit_lds_[ql - 1][kslot];
    if (!wide_knn && (fit_ins == 0 || fit_lds != dyn_lds)) {
#define MOLA_KNN_FIT(KK)                                                                                                   \
    do {                                                                                                                   \
        if (ql == 1) {                                                                                                     \
            HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit_ins, (k_knn_planes<KK, false, 1>), 256, dyn_lds));    \
            HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit_ver, (k_knn_planes<KK, true, 1>), 256, dyn_lds));     \
            HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit_ins4, (k_knn_planes<KK, false, 1, true>), 256, dyn_lds)); \
        } else {                                                                                                           \
            HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit_ins, (k_knn_planes<KK, false, 2>), 256, dyn_lds));    \
            HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit_ver, (k_knn_planes<KK, true, 2>), 256, dyn_lds));     \
            fit_ins4 = fit_ins;   /* (two queries per lane: no dense build) */                                              \
        }                                                                                                                  \
    } while (0)
        switch (p.knn) {  // (the kernels are instantiated on the list length: knn + 1)
            case 3: MOLA_KNN_FIT(4); break;
            case 4: MOLA_KNN_FIT(5); break;
            case 5: MOLA_KNN_FIT(6); break;
            case 6: MOLA_KNN_FIT(7); break;
            case 7: MOLA_KNN_FIT(8); break;
            default: MOLA_KNN_FIT(9); break;
        }
#undef MOLA_KNN_FIT
        fit_lds = dyn_lds;
    }
    auto clampi = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
    int grid = num_cus_ * clampi(fit_ins, 1, g_knobs.blocks_per_cu > 0 ? g_knobs.blocks_per_cu : 3);
    int grid_ver = num_cus_ * clampi(fit_ver, 1, g_knobs.blocks_per_cu > 0 ? g_knobs.blocks_per_cu : 4);
    int grid4 = num_cus_ * clampi(fit_ins4, 1, g_knobs.blocks_per_cu > 0 ? g_knobs.blocks_per_cu : 4);   // seeded insertion launches (DENSE build)
    const int n_items = (int)((N_ + (size_t)(64 * ql) - 1) / (size_t)(64 * ql));
    if (grid > (n_items + 3) / 4) grid = (n_items + 3) / 4;
    if (grid_ver > (n_items + 3) / 4) grid_ver = (n_items + 3) / 4;
    if (grid4 > (n_items + 3) / 4) grid4 = (n_items + 3) / 4;
    const int knn_seed = (bootstrapped || (knn_seed_valid_ && planes_knn_ == (int)p.knn && !g_knobs.no_knn_seed)) ? 1 : 0;
    // the cached plane of an unchanged neighbour list carries the planar / non-planar decision of the launch that
    // solved it: reusable only under the same planeEigenThreshold (the seeds themselves do not depend on it)
    const int plane_cache_ok = (knn_seed && !bootstrapped && planes_eig_thr_ == plane_eig_arg(p)) ? 1 : 0;
    // the counting flavour pays off when few items will need the insertion flavour afterwards: judged by the
    // number of items whose lists changed in the previous iteration (read back with its accumulators)
    // ... and only inside a converging sequence of poses: the first launch of another align on the same clouds starts far from
    // where the last one ended -- every list changes, the counting pass (0.27 ms at C3) would queue every item.  Judged by the
    // pose step against the previous one (a heuristic: rotation weighed with a 30 m lever; results do not depend on it).
    const bool converging = step <= 4.0 * knn_last_step_ + 1e-9;
    knn_last_step_ = knn_seed ? step : 1e30;   // (after an unseeded launch any step counts as "converging")
    const bool verify = knn_seed && !bootstrapped && knn_changed_items_ >= 0.0 && knn_changed_items_ < 0.3 * (double)n_items && converging &&
                        !g_knobs.no_knn_verify;
    knn_changed_items_ = -1.0;  // consumed: only an accumulate_planes() after this launch renews it
    if ((rc = redo_list_.reserve(sizeof(int) * (size_t)n_items))) return rc;
    // heavy-first order of the full sweeps (cycles per item of the last full sweep; re-sorted at launch 1, 2, 4, 8, then every 16th)
    {
        const void* before = knn_cost_.p;
        // (the cooperative kernel records a lifetime per 64-query item whatever MOLA_ICP_QPL makes of the persistent kernel's items)
        if ((rc = knn_cost_.reserve(sizeof(unsigned int) * std::max((size_t)n_items, (N_ + 63) / 64)))) return rc;
        if (knn_cost_.p != before) knn_cost_valid_ = false;
    }
    if ((rc = knn_order_.reserve(sizeof(int) * ((size_t)n_items + kQueues + 1)))) return rc;
    const int* knn_order = nullptr;
    if ((rc = order_join())) return rc;   // a re-sort launched behind the last launch (below)
    // (with at most one item per wave there is nothing to balance, and heavy-first would put the heaviest four on ONE CU)
    // (the cooperative kernel has no queue and consumes no order: no re-sort for it, and no per-item clock reads outside MOLA_ICP_DEBUG_STATS=5)
    if (knn_cost_valid_ && !g_knobs.no_lpt && !knn_coop && n_items > grid * 4) {
        if (!knn_order_valid_ || knn_launches_since_order_ >= knn_plan_interval_) {
            hipLaunchKernelGGL(k_order_items, dim3(1), dim3(1024), 0, stream_, knn_cost_.as<unsigned int>(), n_items, knn_order_.as<int>());
            HIPCHK(hipGetLastError());
            knn_plan_interval_ = knn_order_valid_ ? (knn_plan_interval_ < 16 ? knn_plan_interval_ * 2 : 16) : 1;
            knn_order_valid_ = true;
            knn_launches_since_order_ = 0;
        }
        ++knn_launches_since_order_;
        knn_order = knn_order_.as<int>();
    }
    knn_cost_valid_ = !knn_coop;   // (only the persistent flavours record item costs in the product build)
    unsigned int* tq = reinterpret_cast<unsigned int*>(acc_dev_.as<double>() + kNAcc + 8);
    // certified lists (kernels_planes.hpp, KnnCert): the bound and the pose of the launch that wrote these seeds
    KnnCert cert{};
    for (int k = 0; k < 9; ++k) cert.Pprev.R[k] = knn_last_P_[k];
    for (int k = 0; k < 3; ++k) cert.Pprev.t[k] = knn_last_P_[9 + k];
    cert.lb = knn_lb_.as<float>();
    cert.on = (knn_seed && !bootstrapped && !g_knobs.no_certify) ? 1 : 0;
    cert.stats = (profiling_ || g_knobs.debug_stats == 3) ? stats_.as<unsigned long long>() : nullptr;
    // the lists' own gate (KnnCert): 1.1 x the matcher's; seeds kept under one gate are not reused under another
    const float thr2x = g_knobs.no_certify ? thr2 : thr2 * 1.21f;
    for (int k = 0; k < 9; ++k) knn_last_P_[k] = P.R[k];
    for (int k = 0; k < 3; ++k) knn_last_P_[9 + k] = P.t[k];
    // warm-started launches: the counting flavour over all items, then the insertion flavour over the items it
    // queued (counter[2] = their number); first launch on a cloud pair: the insertion flavour over all items
#define MOLA_LAUNCH_KNN(KK, VER, QLL, QUEUE, LIST) MOLA_LAUNCH_KNN_D(KK, VER, QLL, false, QUEUE, LIST)
#define MOLA_LAUNCH_KNN_D(KK, VER, QLL, DENSE, QUEUE, LIST)                                                               \
    hipLaunchKernelGGL((k_knn_planes<KK, VER, QLL, DENSE>), dim3(VER ? grid_ver : (DENSE ? grid4 : grid)), dim3(256), dyn_lds, stream_, sl, sl + loc_sc_->padded,   \
                       sl + 2 * loc_sc_->padded, (int)N_, mp, P, thr2, thr2x, p.matcher_threshold, plane_eig_arg(p),   \
                       planes_.as<PlanePair>(), plane_cache_.as<PlanePair>(), seeds, knn_seed, plane_cache_ok, QUEUE, \
                       counter + 2, LIST, ((QUEUE) == tq ? tq + kQueues * kQueueStride : tq) + 1, staged, lds_boxes, knn_order, knn_cost_.as<unsigned int>(), g_knobs.early_pop ? 1 : 0, cert)
#define MOLA_LAUNCH_KNN_QL(KK, QLL)                                                                                  \
    do {                                                                                                             \
        if (verify) {                                                                                                \
            MOLA_LAUNCH_KNN(KK, true, QLL, tq, redo_list_.as<int>());                                                \
            if (QLL == 1) MOLA_LAUNCH_KNN_D(KK, false, 1, true, tq + kQueues * kQueueStride, redo_list_.as<int>());   \
            else MOLA_LAUNCH_KNN(KK, false, QLL, tq + kQueues * kQueueStride, redo_list_.as<int>());                 \
        } else if (knn_seed && QLL == 1) {   /* seeded: the dense build, four workgroups per CU */                   \
            MOLA_LAUNCH_KNN_D(KK, false, 1, true, tq, (int*)nullptr);                                                \
        } else {                                                                                                     \
            MOLA_LAUNCH_KNN(KK, false, QLL, tq, (int*)nullptr);                                                      \
        }                                                                                                            \
    } while (0)
#define MOLA_LAUNCH_KNN_ALL(KK)                                                                                      \
    do {                                                                                                             \
        if (ql == 1) MOLA_LAUNCH_KNN_QL(KK, 1);                                                                      \
        else MOLA_LAUNCH_KNN_QL(KK, 2);                                                                              \
    } while (0)
#define MOLA_LAUNCH_KNN_COOP(KK)                                                                                       \
    hipLaunchKernelGGL((k_knn_coop<KK, 1>), dim3(xcd_grid(n_items64)), dim3(256), dyn_lds, stream_, kb, thr2, thr2x,    \
                       p.matcher_threshold, plane_eig_arg(p), staged, lds_boxes, cert.stats, (unsigned long long*)nullptr)
    if (knn_coop) {
        KnnBatch<1> kb;
        KnnProblem& kp = kb.p[0];
        kp.slx = sl; kp.sly = sl + loc_sc_->padded; kp.slz = sl + 2 * loc_sc_->padded;
        kp.N = (int)N_;
        kp.mp = mp;
        kp.P = P;
        kp.Pprev = cert.Pprev;
        kp.out = planes_.as<PlanePair>(); kp.cache = plane_cache_.as<PlanePair>(); kp.seeds = seeds;
        kp.lb = cert.lb;
        kp.use_seed = knn_seed; kp.use_cache = plane_cache_ok; kp.cert_on = cert.on;
        kp.changed_items = tq + kQueues * kQueueStride + 1;
        kp.cost = g_knobs.debug_stats == 5 ? knn_cost_.as<unsigned int>() : nullptr;   // (diagnostics only: two clock reads and a store per item)
        if (knn_q4) {
            HIPCHK(knn_q4_launch(stream_, (int)p.knn + 1, kb, xcd_grid(n_items64), dyn_lds, thr2, thr2x, p.matcher_threshold, plane_eig_arg(p), staged, lds_boxes, cert.stats, kq4_lpq));
        } else if (g_knobs.debug_stats == 4 && p.knn == 6) {   // diagnostics: where the waves of every item spend their cycles
            DevBuf dg;
            if ((rc = dg.reserve(sizeof(unsigned long long) * kKnnDiagWords * 4 * (size_t)n_items64))) return rc;
            HIPCHK(hipMemsetAsync(dg.p, 0, sizeof(unsigned long long) * kKnnDiagWords * 4 * (size_t)n_items64, stream_));
            hipLaunchKernelGGL((k_knn_coop<7, 1, true>), dim3(xcd_grid(n_items64)), dim3(256), dyn_lds, stream_, kb, thr2, thr2x, p.matcher_threshold,
                               plane_eig_arg(p), staged, lds_boxes, cert.stats, dg.as<unsigned long long>());
            HIPCHK(hipGetLastError());
            std::vector<unsigned long long> h((size_t)kKnnDiagWords * 4 * (size_t)n_items64);
            HIPCHK(hipMemcpyAsync(h.data(), dg.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream_));
            HIPCHK(hipStreamSynchronize(stream_));
            dg.release();
            static const char* const names[12] = {"prologue (seeds, sort, certificate)", "wait at the first barrier", "sweep (all of it)", "  list fill + its barrier",
                                                  "  tile-box wait", "  tile tests + passes", "    staging (wait for points)", "    distance passes + insertions",
                                                  "merge (+ barrier)", "epilogue (planes, stores)", "", "whole item"};
            auto pct = [](std::vector<unsigned long long>& v, double q) { if (v.empty()) return 0ull; std::sort(v.begin(), v.end()); return v[(size_t)(q * (double)(v.size() - 1))]; };
            std::fprintf(stderr, "[mola_icp debug] k_knn_coop<7> item phases, shader cycles (median / p90 / max): leader wave | other waves; N=%zu seed=%d cert=%d step=%.5f\n", N_, knn_seed, cert.on, step);
            for (int w = 0; w < 12; ++w) {
                if (w == 10) continue;
                std::vector<unsigned long long> a, b;
                for (int it = 0; it < n_items64; ++it)
                    for (int wv = 0; wv < 4; ++wv) (wv == 0 ? a : b).push_back(h[((size_t)it * 4 + wv) * kKnnDiagWords + w]);
                std::fprintf(stderr, "[mola_icp debug]   %-38s %7llu / %7llu / %7llu | %7llu / %7llu / %7llu\n", names[w], pct(a, 0.5), pct(a, 0.9), pct(a, 1.0), pct(b, 0.5),
                             pct(b, 0.9), pct(b, 1.0));
            }
            std::vector<unsigned long long> staged_v, tests_v, open_v, supers_v;
            unsigned long long n_changed = 0, n_skip = 0;
            for (int it = 0; it < n_items64; ++it) {
                unsigned long long st = 0, tt = 0;
                for (int wv = 0; wv < 4; ++wv) { const unsigned long long c = h[((size_t)it * 4 + wv) * kKnnDiagWords + 10]; st += c & 0xfffffull; tt += (c >> 32) & 0xfffffull; }
                const unsigned long long c0 = h[((size_t)it * 4) * kKnnDiagWords + 10];
                staged_v.push_back(st); tests_v.push_back(tt); open_v.push_back((c0 >> 52) & 0xffull); supers_v.push_back((c0 >> 20) & 0xfffull);
                n_changed += (c0 >> 60) & 1ull; n_skip += (c0 >> 61) & 1ull;
            }
            std::fprintf(stderr, "[mola_icp debug]   per item (median / p90 / max): staged points %llu / %llu / %llu, tile tests %llu / %llu / %llu, super-tiles entered %llu / %llu / %llu, lanes not certified %llu / %llu / %llu; items with a plane solve %llu, items that skipped the sweep %llu of %d\n",
                         pct(staged_v, 0.5), pct(staged_v, 0.9), pct(staged_v, 1.0), pct(tests_v, 0.5), pct(tests_v, 0.9), pct(tests_v, 1.0), pct(supers_v, 0.5), pct(supers_v, 0.9),
                         pct(supers_v, 1.0), pct(open_v, 0.5), pct(open_v, 0.9), pct(open_v, 1.0), n_changed, n_skip, n_items64);
            {   // what the sweep's visitor did: groups of four staged points, how many took the slow path, lane events in it
                unsigned long long groups = 0, slow = 0, keypass = 0, dup = 0, ins = 0;
                for (int it = 0; it < n_items64; ++it)
                    for (int wv = 0; wv < 4; ++wv) {
                        const unsigned long long a = h[((size_t)it * 4 + wv) * kKnnDiagWords + 12], b = h[((size_t)it * 4 + wv) * kKnnDiagWords + 13];
                        groups += a & 0xffffffffull; slow += a >> 32; keypass += b & 0x1fffffull; dup += (b >> 21) & 0x1fffffull; ins += b >> 42;
                    }
                std::fprintf(stderr, "[mola_icp debug]   visitor, whole launch: %llu groups of four points, %llu (%.1f %%) took the slow path; lane events inside it: %llu keys below the lane's K-th "
                                     "(%.2f per query), of which %llu seeds met again, %llu insertions (%.2f per query)\n",
                             groups, slow, 100.0 * (double)slow / (double)(groups ? groups : 1), keypass, (double)keypass / (double)N_, dup, ins, (double)ins / (double)N_);
            }
            // ... and is a heavy item a SPREAD one?  (64 consecutive sorted queries that are not compact in space: the sparse far rings
            // of a scan, an empty stretch of the curve): the item's own bounding box, from the sorted queries
            {
                const size_t np_ = loc_sc_->padded;
                std::vector<float> sq(3 * np_);
                HIPCHK(hipMemcpy(sq.data(), loc_sc_->sorted.p, sizeof(float) * 3 * np_, hipMemcpyDeviceToHost));
                std::vector<double> diag((size_t)n_items64, 0.0);
                for (int it = 0; it < n_items64; ++it) {
                    double lo[3] = {1e30, 1e30, 1e30}, hi[3] = {-1e30, -1e30, -1e30};
                    for (size_t q = (size_t)it * 64; q < std::min((size_t)N_, (size_t)(it + 1) * 64); ++q)
                        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], (double)sq[a * np_ + q]); hi[a] = std::max(hi[a], (double)sq[a * np_ + q]); }
                    diag[(size_t)it] = std::sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
                }
                std::vector<int> ord((size_t)n_items64);
                for (int it = 0; it < n_items64; ++it) ord[(size_t)it] = it;
                auto whole = [&](int it) { return h[((size_t)it * 4) * kKnnDiagWords + 11]; };
                std::sort(ord.begin(), ord.end(), [&](int a, int b) { return whole(a) > whole(b); });
                for (int r = 0; r < 8 && r < n_items64; ++r) {
                    const int it = ord[(size_t)r];
                    unsigned long long st = 0, tt = 0;
                    for (int wv = 0; wv < 4; ++wv) { const unsigned long long c = h[((size_t)it * 4 + wv) * kKnnDiagWords + 10]; st += c & 0xfffffull; tt += (c >> 32) & 0xfffffull; }
                    std::fprintf(stderr, "[mola_icp debug]   heaviest item %d: %llu cycles, box diagonal of its 64 queries %.2f m, staged points %llu, tile tests %llu\n", it, whole(it),
                                 diag[(size_t)it], st, tt);
                }
                std::vector<unsigned long long> compact, spread;
                for (int it = 0; it < n_items64; ++it) (diag[(size_t)it] > 3.0 ? spread : compact).push_back(whole(it));
                std::fprintf(stderr, "[mola_icp debug]   whole item, queries within 3 m of each other (%zu items): %llu / %llu / %llu; spread wider (%zu items): %llu / %llu / %llu\n",
                             compact.size(), pct(compact, 0.5), pct(compact, 0.9), pct(compact, 1.0), spread.size(), pct(spread, 0.5), pct(spread, 0.9), pct(spread, 1.0));
            }
        } else
        switch (p.knn) {
            case 3: MOLA_LAUNCH_KNN_COOP(4); break;
            case 4: MOLA_LAUNCH_KNN_COOP(5); break;
            case 5: MOLA_LAUNCH_KNN_COOP(6); break;
            case 6: MOLA_LAUNCH_KNN_COOP(7); break;
            case 7: MOLA_LAUNCH_KNN_COOP(8); break;
            case 8: MOLA_LAUNCH_KNN_COOP(9); break;
            case 9: MOLA_LAUNCH_KNN_COOP(10); break;
            case 10: MOLA_LAUNCH_KNN_COOP(11); break;
            case 11: MOLA_LAUNCH_KNN_COOP(12); break;
            case 12: MOLA_LAUNCH_KNN_COOP(13); break;
            case 13: MOLA_LAUNCH_KNN_COOP(14); break;
            case 14: MOLA_LAUNCH_KNN_COOP(15); break;
            case 15: MOLA_LAUNCH_KNN_COOP(16); break;
            default: MOLA_LAUNCH_KNN_COOP(17); break;
        }
    } else
    switch (p.knn) {
        case 3: MOLA_LAUNCH_KNN_ALL(4); break;
        case 4: MOLA_LAUNCH_KNN_ALL(5); break;
        case 5: MOLA_LAUNCH_KNN_ALL(6); break;
        case 6: MOLA_LAUNCH_KNN_ALL(7); break;
        case 7: MOLA_LAUNCH_KNN_ALL(8); break;
        default: MOLA_LAUNCH_KNN_ALL(9); break;
    }
#undef MOLA_LAUNCH_KNN_COOP
#undef MOLA_LAUNCH_KNN_ALL
#undef MOLA_LAUNCH_KNN_QL
#undef MOLA_LAUNCH_KNN
#undef MOLA_LAUNCH_KNN_D
    HIPCHK(hipGetLastError());
    if (profiling_) {
        HIPCHK(hipEventRecord(ev_[ev_used_ + 1], stream_));
        ev_used_ += 2;
    }
    // the next launch's re-sort, if one is due, on the side stream (see launch_tiled)
    if (!g_knobs.no_lpt && !knn_coop && n_items > grid * 4 && (!knn_order_valid_ || knn_launches_since_order_ >= knn_plan_interval_)) {
        if ((rc = order_begin())) return rc;
        hipLaunchKernelGGL(k_order_items, dim3(1), dim3(1024), 0, aux_stream_, knn_cost_.as<unsigned int>(), n_items, knn_order_.as<int>());
        if ((rc = order_end())) return rc;
        knn_plan_interval_ = knn_order_valid_ ? (knn_plan_interval_ < 16 ? knn_plan_interval_ * 2 : 16) : 1;
        knn_order_valid_ = true;
        knn_launches_since_order_ = 0;
    }
    if (g_knobs.debug_stats == 5 && (knn_coop || (!verify && ql == 1))) {   // diagnostics: the items' lifetimes in THIS launch of the product kernel (two clock reads per item)
        std::vector<unsigned int> c((size_t)n_items64);
        HIPCHK(hipMemcpyAsync(c.data(), knn_cost_.p, c.size() * sizeof(unsigned int), hipMemcpyDeviceToHost, stream_));
        HIPCHK(hipStreamSynchronize(stream_));
        std::vector<unsigned int> v = c;
        std::sort(v.begin(), v.end());
        double sum = 0;
        for (unsigned int x : v) sum += x;
        const size_t n = v.size();
        const int slots = num_cus_ * 4;
        std::fprintf(stderr, "[mola_icp debug] %s item lifetimes (shader cycles): N=%zu items=%zu seed=%d cert=%d bootstrapped=%d step=%.4f | mean %.0f p10 %u p50 %u p90 %u p99 %u max %u | "
                             "sum / %d workgroup slots = %.0f = %.2f x the heaviest item; items above 2x the median: %zu\n",
                     knn_coop ? "k_knn_coop (one workgroup per item)" : "k_knn_planes (one wave per item)", N_, n, knn_seed, cert.on, (int)bootstrapped, step, sum / (double)n, v[n / 10], v[n / 2], v[n * 9 / 10], v[n * 99 / 100], v[n - 1], slots, sum / slots,
                     sum / slots / (double)v[n - 1], (size_t)(v.end() - std::upper_bound(v.begin(), v.end(), 2u * v[n / 2])));
        // where along the curve the heavy ones sit (16 stretches of the item order: share of the total cost)
        std::fprintf(stderr, "[mola_icp debug]   cost share of 16 stretches of the item order:");
        for (int b = 0; b < 16; ++b) {
            double sb = 0;
            for (size_t i = n * (size_t)b / 16; i < n * (size_t)(b + 1) / 16; ++i) sb += c[i];
            std::fprintf(stderr, " %.3f", sb / sum);
        }
        std::fprintf(stderr, "\n");
    }
    if (g_knobs.debug_stats == 3) {   // diagnostics: what the certificates did in THIS launch (a synchronisation per launch)
        HIPCHK(hipMemcpyAsync(stats_host_, stats_.p, sizeof(unsigned long long) * kStatSlots * kStatStride, hipMemcpyDeviceToHost, stream_));
        HIPCHK(hipStreamSynchronize(stream_));
        unsigned long long certified = 0, skipped = 0, staged64 = 0;
        for (int k = 0; k < kStatSlots; ++k) { staged64 += stats_host_[(size_t)k * kStatStride]; certified += stats_host_[(size_t)k * kStatStride + 1]; skipped += stats_host_[(size_t)k * kStatStride + 2]; }
        HIPCHK(hipMemsetAsync(stats_.p, 0, sizeof(unsigned long long) * kStatSlots * kStatStride, stream_));
        std::fprintf(stderr, "[mola_icp debug] plane matcher launch: N=%zu items=%d %s seed=%d cert=%d bootstrapped=%d step=%.5f m | certified queries %llu (%.1f %%), items that skipped the sweep %llu (%.1f %%), pairs/query %.1f\n",
                     N_, n_items64, knn_q4 ? (kq4_lpq == 1 ? "q4, one lane per query" : (kq4_lpq == 2 ? "q4, two lanes per query (sweeps skipped: per 32-query wave)" : "q4 (sweeps skipped: per 16-query wave)")) : (knn_coop ? "coop" : (verify ? "persistent+count" : "persistent")), knn_seed, cert.on, (int)bootstrapped, step, certified,
                     100.0 * (double)certified / (double)N_, skipped, 100.0 * (double)skipped / (double)(knn_q4 ? kq4_lpq * n_items64 : n_items64), 64.0 * (double)staged64 / (double)N_);
    }
    last_kernel_ = MOLA_ICP_NN_TILED;
    planes_knn_ = (int)p.knn;
    planes_eig_thr_ = plane_eig_arg(p);
    if (g_knobs.turn_clock) g_turn.on_launched();
    planes_valid_ = true;
    knn_seed_valid_ = true;
    return MOLA_ICP_OK;
}

// An all-reduce hook returned non-zero.  The node-local communicator (local_comm.cpp) has already recorded WHICH rank is ahead,
// gave up or timed out in this thread's error string -- the diagnostic include/mola_icp_amd.h promises: it is kept, with the code in
// front; a foreign hook that set nothing gets the plain message.
static int hook_failed(int r, const std::string& error_before_the_call)
{
    const std::string detail = last_error();
    std::string msg = "all-reduce hook failed with code " + std::to_string(r);
    if (!detail.empty() && detail != error_before_the_call) msg += ": " + detail;   // (what the hook itself recorded)
    return fail(MOLA_ICP_E_COMM, msg);
}

int HipWorkspace::accumulate_planes(double acc[kNAccPlaneHost])
{
    int rc = init();
    if (rc) return rc;
    if (!planes_valid_) return fail(MOLA_ICP_E_BADARG, "accumulate_planes() called before match_planes()");
    HIPCHK(hipSetDevice(device_));
    if ((rc = plane_acc_.reserve(sizeof(double) * kNAccPlane * 514))) return rc;
    double* dacc = plane_acc_.as<double>() + (size_t)512 * kNAccPlane;
    if (!plane_acc_host_) {
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&plane_acc_host_), sizeof(double) * (kNAccPlane + 4),
                             hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(plane_acc_host_, 0, sizeof(double) * (kNAccPlane + 4));
    }
    // single GPU: the reduction writes the form (+ the changed-lists count) straight into the pinned block and then the
    // sequence number the host spins on; sharded over RCCL the collective runs on the device block first
    const bool direct = !comm_ && !g_knobs.no_direct_readback && !planes_empty_;
    const unsigned long long seq = ++readback_seq_;
    if (planes_empty_) {
        HIPCHK(hipMemsetAsync(dacc, 0, sizeof(double) * (kNAccPlane + 1), stream_));
    } else {
        int nblocks = (int)((N_ + 255) / 256);
        if (nblocks > 512) nblocks = 512;
        const float* sl = loc_sc_->sorted.as<float>();
        if (g_knobs.planes_valu)
            hipLaunchKernelGGL(k_accumulate_planes, dim3(nblocks), dim3(256), 0, stream_, sl, sl + loc_sc_->padded, sl + 2 * loc_sc_->padded,
                               planes_.as<PlanePair>(), (int)N_, plane_acc_.as<double>());
        else
            hipLaunchKernelGGL(k_accumulate_planes_mfma, dim3(nblocks), dim3(256), 0, stream_, sl, sl + loc_sc_->padded, sl + 2 * loc_sc_->padded,
                               planes_.as<PlanePair>(), (int)N_, plane_acc_.as<double>());
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(k_reduce_rows, dim3(1), dim3(1024), 0, stream_, plane_acc_.as<double>(), nblocks, kNAccPlane, dacc,
                           reinterpret_cast<const unsigned int*>(acc_dev_.as<double>() + kNAcc), direct ? plane_acc_host_ : (double*)nullptr, seq);
        HIPCHK(hipGetLastError());
        counters_clean_ = true;
    }
    if (comm_) {
        if (slab_violation_) {  // see check_slab(): every rank learns of it through the sum (slot 91 = the pair count)
            acc_host_[kNAcc + 7] = std::nan("");
            HIPCHK(hipMemcpyAsync(dacc + kNAccPlane - 1, acc_host_ + kNAcc + 7, sizeof(double), hipMemcpyHostToDevice, stream_));
        }
        const int rc2 = rccl_allreduce_sum_f64(comm_, dacc, kNAccPlane, stream_);
        if (rc2) return rc2;
    }
    if (g_knobs.no_direct_readback) {
        HIPCHK(hipMemcpyAsync(plane_acc_host_, dacc, sizeof(double) * (kNAccPlane + 1), hipMemcpyDeviceToHost, stream_));
        HIPCHK(hipStreamSynchronize(stream_));
        if (bbox_pending_) { const int rcb = check_bboxes(); if (rcb) return rcb; }
    } else {
        if (!direct) {  // (after the collective, or an empty shard's zeros)
            hipLaunchKernelGGL(k_publish, dim3(1), dim3(128), 0, stream_, dacc, kNAccPlane + 1, plane_acc_host_, kNAccPlane + 2, seq);
            HIPCHK(hipGetLastError());
        }
        if ((rc = spin_for(reinterpret_cast<volatile unsigned long long*>(plane_acc_host_) + kNAccPlane + 2, seq))) return rc;
    }
    std::memcpy(acc, plane_acc_host_, sizeof(double) * kNAccPlane);
    knn_changed_items_ = planes_empty_ ? -1.0 : plane_acc_host_[kNAccPlane];
    if (!comm_ && ar_fn_) {
        if (slab_violation_) acc[kNAccPlane - 1] = std::nan("");
        const std::string err0 = last_error();
        const int r = ar_fn_(acc, kNAccPlane, 0, ar_user_);
        if (r) return hook_failed(r, err0);
    }
    if ((comm_ || ar_fn_) && std::isnan(acc[kNAccPlane - 1])) { slab_violation_ = false; return fail(MOLA_ICP_E_BADARG, kSlabMsg); }
    return MOLA_ICP_OK;
}

// plane pairing to host, original query order: valid[N], centroid[N*3], normal[N*3], knn_idx[N*knn] (each may be null)
int HipWorkspace::copy_planes(uint8_t* valid, double* centroid, double* normal, int32_t* knn_idx)
{
    if (!planes_valid_) return fail(MOLA_ICP_E_BADARG, "no plane pairing stored: call match_planes() first");
    if (planes_empty_ || N_ == 0) return MOLA_ICP_OK;
    HIPCHK(hipSetDevice(device_));
    int rc;
    DevBuf tmp_pairs, tmp_knn;
    if ((rc = tmp_pairs.reserve(sizeof(PlanePair) * N_))) return rc;
    if ((rc = tmp_knn.reserve(sizeof(int) * N_ * (size_t)(planes_knn_ > 8 ? planes_knn_ : 8)))) { tmp_pairs.release(); return rc; }
    hipLaunchKernelGGL(k_unpermute_planes, dim3((unsigned)((N_ + 255) / 256)), dim3(256), 0, stream_, loc_sc_->perm.as<int>(),
                       planes_.as<PlanePair>(), knn_seeds_at(knn_pos_.p, loc_sc_->padded, planes_knn_ + 1), planes_knn_, (int)N_,
                       tmp_pairs.as<PlanePair>(), tmp_knn.as<int>());
    std::vector<PlanePair> hp(N_);
    std::vector<int> hk(N_ * (size_t)planes_knn_);
    hipError_t e1 = hipMemcpyAsync(hp.data(), tmp_pairs.p, sizeof(PlanePair) * N_, hipMemcpyDeviceToHost, stream_);
    hipError_t e2 = hipMemcpyAsync(hk.data(), tmp_knn.p, sizeof(int) * N_ * planes_knn_, hipMemcpyDeviceToHost, stream_);
    hipError_t e3 = hipStreamSynchronize(stream_);
    tmp_pairs.release();
    tmp_knn.release();
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) return fail(MOLA_ICP_E_HIP, "copy_planes: HIP copy failed");
    for (size_t i = 0; i < N_; ++i) {
        if (valid) valid[i] = (uint8_t)hp[i].valid;
        for (int a = 0; a < 3; ++a) {
            if (centroid) centroid[3 * i + a] = hp[i].c[a];
            if (normal) normal[3 * i + a] = hp[i].n[a];
        }
    }
    if (knn_idx) std::memcpy(knn_idx, hk.data(), sizeof(int) * hk.size());
    return MOLA_ICP_OK;
}

void HipWorkspace::reset_stats()
{
    ev_used_ = 0;
    nn_launches_ = 0;
    last_kernel_ = 0;
    dense_pairs_ = 0;
    if (inited_ && profiling_) {
        (void)hipMemsetAsync(acc_dev_.as<double>() + kNAcc + 4, 0, sizeof(unsigned long long), stream_);
        (void)hipMemsetAsync(stats_.p, 0, sizeof(unsigned long long) * kStatSlots * kStatStride, stream_);
    }
}

int HipWorkspace::collect_stats(double* ms_total, uint32_t* launches, uint32_t* kernel_used, uint64_t* pairs)
{
    if (pairs) {
        unsigned long long staged = 0;
        if (inited_ && profiling_) {  // (a copy + stream synchronisation: only when the caller asked for statistics)
            HIPCHK(hipSetDevice(device_));
            HIPCHK(hipMemcpyAsync(acc_host_ + kNAcc + 4, acc_dev_.as<double>() + kNAcc + 4, sizeof staged,
                                  hipMemcpyDeviceToHost, stream_));
            HIPCHK(hipMemcpyAsync(stats_host_, stats_.p, sizeof(unsigned long long) * kStatSlots * kStatStride,
                                  hipMemcpyDeviceToHost, stream_));
            HIPCHK(hipStreamSynchronize(stream_));
            std::memcpy(&staged, acc_host_ + kNAcc + 4, sizeof staged);
            for (int k = 0; k < kStatSlots; ++k) staged += stats_host_[(size_t)k * kStatStride];  // the cooperative kernel's slotted counters
        }
        *pairs = dense_pairs_ + (uint64_t)staged * 64u;  // the tiled kernels count in units of 64 pairs
    }
    double tot = 0;
    if (ev_used_) {
        HIPCHK(hipSetDevice(device_));
        HIPCHK(hipStreamSynchronize(stream_));
        for (size_t i = 0; i + 1 < ev_used_; i += 2) {
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, ev_[i], ev_[i + 1]));
            tot += ms;
        }
    }
    if (wave_times_) {
        std::vector<unsigned long long> w(8 * 8192);
        HIPCHK(hipStreamSynchronize(stream_));
        HIPCHK(hipMemcpy(w.data(), wave_times_, w.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, t1 = 0ull;
        std::vector<unsigned long long> ends, pro, swp, epi, stg, setup, mwait, starts, entered, tiles, supers, boxwait, tiletest, stagec, visitc, fillw, fill0;
        for (size_t i = 0; i < 8192; ++i)
            if (w[8 * i + 1]) {
                t0 = std::min(t0, w[8 * i]); t1 = std::max(t1, w[8 * i + 1]); ends.push_back(w[8 * i + 1]);
                starts.push_back(w[8 * i]);
                if (wave_times_coop_) {  // k_nn_coop's layout: two 32-bit cycle counts per slot
                    auto lo = [&](int k) { return w[8 * i + k] & 0xffffffffull; };
                    auto hi = [&](int k) { return w[8 * i + k] >> 32; };
                    setup.push_back(lo(2)); pro.push_back(hi(2)); swp.push_back(lo(3)); mwait.push_back(hi(3)); epi.push_back(lo(4));
                    boxwait.push_back(hi(4)); tiletest.push_back(lo(5)); stagec.push_back(hi(5)); visitc.push_back(lo(6));
                    ((i & 3) == 0 ? fill0 : fillw).push_back(hi(6));   // wave 0 walks the upper box levels, the others wait for it
                    stg.push_back(w[8 * i + 7] & 0xffffull);
                    entered.push_back((w[8 * i + 7] >> 16) & 0xffffull); tiles.push_back((w[8 * i + 7] >> 32) & 0xffffull);
                    supers.push_back(w[8 * i + 7] >> 48);
                } else {
                    pro.push_back(w[8 * i + 3]); swp.push_back(w[8 * i + 4]); epi.push_back(w[8 * i + 5]); stg.push_back(w[8 * i + 6]);
                }
            }
        if (!ends.empty()) {
            std::sort(ends.begin(), ends.end());
            const double span = (double)(t1 - t0);
            double busy = 0;
            for (unsigned long long e : ends) busy += (double)(e - t0);
            std::fprintf(stderr, "[mola_icp debug] last tiled launch: %zu waves, span %.0f ticks; waves done at 10/25/50/75/90/99%%: "
                                 "%.2f %.2f %.2f %.2f %.2f %.2f of the span; mean wave lifetime %.2f of the span\n",
                         ends.size(), span, (ends[ends.size() / 10] - t0) / span, (ends[ends.size() / 4] - t0) / span,
                         (ends[ends.size() / 2] - t0) / span, (ends[ends.size() * 3 / 4] - t0) / span,
                         (ends[ends.size() * 9 / 10] - t0) / span, (ends[ends.size() * 99 / 100] - t0) / span,
                         busy / ends.size() / span);
            auto med = [](std::vector<unsigned long long>& v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
            if (!wave_times_coop_) {  // the persistent kernel: per XCD (block & 7) when its waves end and how many items they took
                double last[8] = {}, sum[8] = {}, items[8] = {};
                int cnt[8] = {};
                unsigned long long imin = ~0ull, imax = 0;
                for (size_t i = 0; i < 8192; ++i)
                    if (w[8 * i + 1]) {
                        const int x = (int)((i / 4) & 7);
                        const double e = (double)(w[8 * i + 1] - t0) / span;
                        const unsigned long long ni = w[8 * i + 2] & 0xffffffffull;
                        last[x] = std::max(last[x], e); sum[x] += e; items[x] += (double)ni; ++cnt[x];
                        imin = std::min(imin, ni); imax = std::max(imax, ni);
                    }
                std::fprintf(stderr, "[mola_icp debug]   per XCD: last wave ends at / mean end / items per wave:");
                for (int x = 0; x < 8; ++x) std::fprintf(stderr, " %.2f/%.2f/%.1f", last[x], cnt[x] ? sum[x] / cnt[x] : 0.0, cnt[x] ? items[x] / cnt[x] : 0.0);
                std::fprintf(stderr, "; items per wave min %llu max %llu\n", imin, imax);
                // the waves that end last: how long their LAST entry ran, when it began, which entry it was
                std::vector<size_t> late;
                for (size_t i = 0; i < 8192; ++i) if (w[8 * i + 1]) late.push_back(i);
                std::sort(late.begin(), late.end(), [&](size_t a, size_t b) { return w[8 * a + 1] > w[8 * b + 1]; });
                for (size_t r = 0; r < late.size() && r < 400; r = (r < 4 ? r + 1 : r * 3)) {
                    const size_t i = late[r];
                    const unsigned int code = (unsigned int)(w[8 * i + 2] >> 32);
                    std::fprintf(stderr, "[mola_icp debug]   end rank %zu: wave ends %.2f, %llu entries, last entry %s%u began %.2f ran %.2f of the span\n", r,
                                 (double)(w[8 * i + 1] - t0) / span, w[8 * i + 2] & 0xffffffffull, (code & 0x40000000u) ? "half " : "", code & 0x3fffffffu,
                                 (double)(w[8 * i + 7] - t0) / span, (double)(w[8 * i + 1] - w[8 * i + 7]) / span);
                }
                std::vector<double> lastdur;
                for (size_t i : late) lastdur.push_back((double)(w[8 * i + 1] - w[8 * i + 7]) / span);
                std::sort(lastdur.begin(), lastdur.end());
                std::fprintf(stderr, "[mola_icp debug]   duration of a wave's last entry (of the span): p10 %.2f p50 %.2f p90 %.2f max %.2f\n",
                             lastdur[lastdur.size() / 10], lastdur[lastdur.size() / 2], lastdur[lastdur.size() * 9 / 10], lastdur.back());
            }
            if (wave_times_coop_) {
                std::sort(starts.begin(), starts.end());
                std::fprintf(stderr, "[mola_icp debug]   cooperative kernel: last wave starts %.0f ticks after the first; per wave, shader cycles "
                                     "(median / p90 / max): setup %llu / %llu / %llu, merge wait %llu / %llu / %llu, sweep max %llu, staged max %llu\n",
                             (double)(starts.back() - t0), med(setup, 0.5), med(setup, 0.9), med(setup, 1.0), med(mwait, 0.5), med(mwait, 0.9),
                             med(mwait, 1.0), med(swp, 1.0), med(stg, 1.0));
                std::fprintf(stderr, "[mola_icp debug]   per wave (median / p90 / max): super tests %llu / %llu / %llu, supers entered %llu / %llu / %llu, "
                                     "own tile tests %llu / %llu / %llu\n",
                             med(supers, 0.5), med(supers, 0.9), med(supers, 1.0), med(entered, 0.5), med(entered, 0.9), med(entered, 1.0),
                             med(tiles, 0.5), med(tiles, 0.9), med(tiles, 1.0));
                if (!fill0.empty() && !fillw.empty())
                    std::fprintf(stderr, "[mola_icp debug]   list fill + barrier (median / p90 / max cycles): wave 0 (the walk over the upper box levels) %llu / %llu / %llu, waves 1-3 (waiting for it) %llu / %llu / %llu\n",
                                 med(fill0, 0.5), med(fill0, 0.9), med(fill0, 1.0), med(fillw, 0.5), med(fillw, 0.9), med(fillw, 1.0));
                std::fprintf(stderr, "[mola_icp debug]   inside the sweep (median / p90 / max cycles): tile-box wait %llu / %llu / %llu, tile tests + passes %llu / %llu / %llu, "
                                     "of which staging (wait for points) %llu / %llu / %llu, distance passes %llu / %llu / %llu\n",
                             med(boxwait, 0.5), med(boxwait, 0.9), med(boxwait, 1.0), med(tiletest, 0.5), med(tiletest, 0.9), med(tiletest, 1.0),
                             med(stagec, 0.5), med(stagec, 0.9), med(stagec, 1.0), med(visitc, 0.5), med(visitc, 0.9), med(visitc, 1.0));
            }
            std::fprintf(stderr, "[mola_icp debug]   first item of a wave, shader cycles (median / p90): prologue %llu / %llu, sweep %llu / %llu, "
                                 "epilogue %llu / %llu; staged points %llu / %llu\n",
                         med(pro, 0.5), med(pro, 0.9), med(swp, 0.5), med(swp, 0.9), med(epi, 0.5), med(epi, 0.9), med(stg, 0.5), med(stg, 0.9));
        }
        HIPCHK(hipMemset(wave_times_, 0, w.size() * sizeof(unsigned long long)));
        wave_times_coop_ = false;
    }
    if (dbg_stats_ && !wave_times_) {
        unsigned long long h[16] = {};
        HIPCHK(hipStreamSynchronize(stream_));
        HIPCHK(hipMemcpy(h, dbg_stats_, sizeof h, hipMemcpyDeviceToHost));
        std::fprintf(stderr, "[mola_icp debug] nn launches=%zu slow-path entries=%llu survivors=%llu (N=%zu M=%zu); "
                             "tiled: staged points per wave item=%.1f (items=%llu, max=%llu); cycles per item: prologue %.0f "
                             "scan %.0f stage %.0f compute %.0f epilogue %.0f, longest item %llu; per item: super tests %.1f, supers entered %.1f, tile tests %.1f; of scan: tile-box wait %.0f, tile tests %.0f\n",
                     ev_used_ / 2, h[0], h[1], N_, M_, h[3] ? (double)h[2] / (double)h[3] : 0.0, h[3], h[4],
                     h[3] ? (double)h[5] / h[3] : 0.0, h[3] ? (double)h[6] / h[3] : 0.0, h[3] ? (double)h[7] / h[3] : 0.0,
                     h[3] ? (double)h[8] / h[3] : 0.0, h[3] ? (double)h[10] / h[3] : 0.0, h[9],
                     h[3] ? (double)h[11] / h[3] : 0.0, h[3] ? (double)h[12] / h[3] : 0.0, h[3] ? (double)h[13] / h[3] : 0.0,
                     h[3] ? (double)h[14] / h[3] : 0.0, h[3] ? (double)h[15] / h[3] : 0.0);
        HIPCHK(hipMemset(dbg_stats_, 0, sizeof h));
        const size_t n_items = std::min((N_ + 63) / 64, item_cost_.cap / sizeof(unsigned int));  // upper bound (64-query items)
        if (cost_valid_ && N_ > 0 && n_items <= kDbgItems) {  // the heaviest items of the last tiled launch
            std::vector<unsigned long long> rec(8 * n_items);
            HIPCHK(hipMemcpy(rec.data(), dbg_stats_ + 16, rec.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            std::vector<size_t> ord;
            for (size_t i = 0; i < n_items; ++i) if (rec[8 * i]) ord.push_back(i);
            std::sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return rec[8 * a] > rec[8 * b]; });
            for (size_t r = 0; r < ord.size(); r = (r < 4 ? r + 1 : r * 2)) {
                const unsigned long long* q = &rec[8 * ord[r]];
                std::fprintf(stderr, "[mola_icp debug]   rank %zu item %zu: cycles %llu staged %llu supers %llu tile tests %llu | prologue %llu scan %llu passes %llu epilogue %llu\n",
                             r, ord[r], q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7]);
            }
            HIPCHK(hipMemset(dbg_stats_ + 16, 0, rec.size() * sizeof(unsigned long long)));
            std::vector<unsigned int> c(n_items);
            HIPCHK(hipMemcpy(c.data(), item_cost_.p, n_items * sizeof(unsigned int), hipMemcpyDeviceToHost));
            std::sort(c.begin(), c.end());
            double sum = 0;
            for (unsigned int v : c) sum += v;
            std::fprintf(stderr, "[mola_icp debug] item cost: n=%zu mean %.0f p50 %u p90 %u p99 %u max %u; sum/3072 slots = %.0f\n",
                         n_items, sum / n_items, c[n_items / 2], c[n_items * 9 / 10], c[n_items * 99 / 100], c[n_items - 1],
                         sum / 3072.0);
        }
    }
    if (ms_total) *ms_total = tot;
    if (launches) *launches = nn_launches_;
    if (kernel_used) *kernel_used = last_kernel_;
    return MOLA_ICP_OK;
}

int HipWorkspace::launch_nn(const Mat4& T, float thr2, int kernel)
{
    PoseF P;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) P.R[3 * r + c] = (float)T(r, c);
        P.t[r] = (float)T(r, 3);
    }
    while (ev_.size() < ev_used_ + 2) {
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        ev_.push_back(e);
    }
    // auto: the tiled matcher wherever sorting pays (it prepares both clouds once), else the dense kernels
    if (kernel == MOLA_ICP_NN_AUTO) {
        if (N_ >= 8192 && M_ >= 8192) kernel = MOLA_ICP_NN_TILED;
        else if (N_ >= 4096 && M_ >= 1024) kernel = MOLA_ICP_NN_MFMA;
        else kernel = MOLA_ICP_NN_VALU;
    }
    if (kernel == MOLA_ICP_NN_TILED) {
        const int rc = prepare_both();
        if (rc) return rc;
    } else if (kernel == MOLA_ICP_NN_MFMA) {
        const int rc = prepare_map();
        if (rc) return rc;
    }
    unsigned int* counter = reinterpret_cast<unsigned int*>(acc_dev_.as<double>() + kNAcc);
    if (!counters_clean_) {  // [0] kept [1] queue (dense kernels) [2] redo count; then the tiled kernels' queue counters
        HIPCHK(hipMemsetAsync(counter, 0, 4 * sizeof(unsigned int), stream_));
        HIPCHK(hipMemsetAsync(acc_dev_.as<double>() + kNAcc + 8, 0, sizeof(unsigned int) * 2 * kQueues * kQueueStride, stream_));
    }
    ++nn_launches_;
    if (profiling_) HIPCHK(hipEventRecord(ev_[ev_used_], stream_));
    if (kernel == MOLA_ICP_NN_TILED) {
        const bool use_seed = seed_valid_ && pairing_sorted_ && !g_knobs.no_warm_start;
        // fewer 128-query items than persistent wave slots: the launch would be one item long -> one WORKGROUP per item
        const size_t n128 = (N_ + kQPW - 1) / kQPW;
        // (crossover measured with the 64-query persistent kernel, us per ICP iteration cooperative / persistent: 50k 34 / 42,
        //  100k 41 / 43, 150k 50 / 47, 200k 58 / 50, 390k 93 / 65)
        const bool coop = g_knobs.coop >= 0 ? g_knobs.coop != 0 : n128 <= (size_t)num_cus_ * 4;
        // ... and within that range four lanes per query (k_nn_q4): an item a sixteenth of the cooperative kernel's, the box tests four
        // at a time.  MOLA_ICP_Q4=0|1 forces either; the diagnostic builds keep the cooperative kernel and its clocks.
        // (crossover measured against the 64-query persistent kernel, us per ICP iteration q4 / persistent: 200k 42.6 / 46.4, 260k 49.2 / 48.8;
        //  against k_nn_coop: 12k 24.1 / 30.0, 100k 28.0 / 39.0, 125k x a 1M-point map 55 / 78 -- profiles/r06/q4_ab.txt)
        const bool q4 = !wave_times_ && (g_knobs.q4 >= 0 ? g_knobs.q4 != 0 : (g_knobs.coop < 0 && N_ <= kQ4MaxQueries));
        const int rc = q4 ? launch_q4(P, thr2, use_seed) : (coop ? launch_coop(P, thr2, use_seed) : launch_tiled(P, thr2, use_seed, counter));
        if (rc) return rc;
        last_kernel_ = MOLA_ICP_NN_TILED;
        pairing_sorted_ = true;
        if (!coop && !q4) counters_clean_ = false;   // (the cooperative kernel has no queue and no redo list: it leaves the counters as they are)
        if (profiling_) {
            HIPCHK(hipEventRecord(ev_[ev_used_ + 1], stream_));
            ev_used_ += 2;
        }
        return MOLA_ICP_OK;
    }
    const bool use_mfma = kernel == MOLA_ICP_NN_MFMA;
    if (use_mfma) {
        const int n_qgroups = (int)((N_ + kMfmaQT * 16 - 1) / (kMfmaQT * 16));
        const int n_items = n_qgroups * map_segs_;
        int rc;
        if ((rc = seg_idx_.reserve(sizeof(int) * (size_t)map_segs_ * N_))) return rc;
        if ((rc = seg_d2_.reserve(sizeof(float) * (size_t)map_segs_ * N_))) return rc;
        MapFrame F{map_center_[0], map_center_[1], map_center_[2], map_radius_};
        // warm start from the pairing this workspace computed last for the same clouds
        const int* seed = (seed_valid_ && !pairing_sorted_ && !g_knobs.no_warm_start) ? idx_.as<int>()
                                                                                                        : nullptr;
        // persistent grid: every CU gets its resident blocks (2-3 per CU at this register count)
        const int per_cu = g_knobs.blocks_per_cu > 0 ? g_knobs.blocks_per_cu : 3;  // tuning knob
        int grid = num_cus_ * per_cu;
        if (grid > (n_items + 3) / 4) grid = (n_items + 3) / 4;
        hipLaunchKernelGGL((k_nn_mfma<kMfmaQT>), dim3(grid), dim3(256), 0, stream_, lx_, ly_, lz_, (int)N_, gx_, gy_,
                           gz_, (int)M_, map_img_.as<float>(), map_tiles_, map_seg_tiles_, map_segs_, n_qgroups, F, P,
                           thr2, seed, seg_idx_.as<int>(), seg_d2_.as<float>(), counter + 1, dbg_stats_);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(k_nn_merge, dim3((unsigned)((N_ + 255) / 256)), dim3(256), 0, stream_, seg_idx_.as<int>(),
                           seg_d2_.as<float>(), map_segs_, (int)N_, idx_.as<int>(), d2_.as<float>(), counter);
        last_kernel_ = MOLA_ICP_NN_MFMA;
    } else {
        constexpr int QPT = 4, TM = 1024;
        const int grid = (int)((N_ + 256 * QPT - 1) / (256 * QPT));
        hipLaunchKernelGGL((k_nn_valu<QPT, TM>), dim3(grid), dim3(256), 0, stream_, lx_, ly_, lz_, (int)N_, gx_, gy_,
                           gz_, (int)M_, P, thr2, idx_.as<int>(), d2_.as<float>(), counter);
        last_kernel_ = MOLA_ICP_NN_VALU;
    }
    HIPCHK(hipGetLastError());
    if (profiling_) {
        HIPCHK(hipEventRecord(ev_[ev_used_ + 1], stream_));
        ev_used_ += 2;
    }
    pairing_sorted_ = false;
    rows_valid_ = false;
    counters_clean_ = false;
    dense_pairs_ += (uint64_t)N_ * (uint64_t)M_;
    return MOLA_ICP_OK;
}

int HipWorkspace::match(const Mat4& T, double threshold, const mola_icp_params& p, uint64_t* n_pairs)
{
    if (g_knobs.turn_clock) g_turn.on_enter();
    int rc = init();
    if (rc) return rc;
    HIPCHK(hipSetDevice(device_));
    if (!(threshold > 0)) return fail(MOLA_ICP_E_BADARG, "matcher threshold must be > 0");
    if ((rc = idx_.reserve(sizeof(int) * (N_ ? N_ : 1)))) return rc;
    if ((rc = d2_.reserve(sizeof(float) * (N_ ? N_ : 1)))) return rc;
    {
        const void* before = outlier_.p;
        if ((rc = outlier_.reserve(N_ ? N_ : 1))) return rc;
        if (outlier_.p != before) outlier_cleared_for_ = 0;  // fresh allocation: contents undefined
    }
    const float thr2 = (float)(threshold * threshold);
    if ((rc = check_slab(T, threshold))) return rc;
    if (N_ == 0 || M_ == 0) {
        if (N_) HIPCHK(hipMemsetAsync(idx_.p, 0xff, sizeof(int) * N_, stream_));
        pairing_valid_ = true;
        seed_valid_ = false;
    knn_seed_valid_ = false;
        pairing_sorted_ = false;
        if (n_pairs) *n_pairs = 0;
        return MOLA_ICP_OK;
    }
    // No pairing of these clouds to start from, but the plane matcher's lists (the quality pass behind a point-to-plane loop whose
    // first launch was seeded by key, without an NN pass): the first entry of every list -- the nearest map point at the loop's last
    // pose -- seeds the tiled matcher.
    if (!seed_valid_ && knn_seed_valid_ && planes_knn_ >= 3 && !g_knobs.no_warm_start && !planes_empty_ && loc_sc_->ready && map_sc_->ready &&
        (p.nn_kernel == MOLA_ICP_NN_TILED || (p.nn_kernel == MOLA_ICP_NN_AUTO && N_ >= 8192 && M_ >= 8192))) {
        if ((rc = ts_pos_.reserve(sizeof(int) * loc_sc_->padded))) return rc;
        if ((rc = ts_idx_.reserve(sizeof(int) * loc_sc_->padded))) return rc;
        if ((rc = ts_d2_.reserve(sizeof(float) * loc_sc_->padded))) return rc;
        if ((rc = ts_gs_.reserve(sizeof(float) * 3 * loc_sc_->padded))) return rc;
        float* gs = ts_gs_.as<float>();
        hipLaunchKernelGGL(k_nn_seeds_from_lists, dim3((unsigned)((N_ + 255) / 256)), dim3(256), 0, stream_,
                           knn_seeds_at(knn_pos_.p, loc_sc_->padded, planes_knn_ + 1), (int)N_, ts_pos_.as<int>(), ts_idx_.as<int>(), gs,
                           gs + loc_sc_->padded, gs + 2 * loc_sc_->padded);
        HIPCHK(hipGetLastError());
        seed_valid_ = true;
        pairing_sorted_ = true;
        pairing_valid_ = false;   // (seeds, not a pairing)
    }
    if ((rc = launch_nn(T, thr2, p.nn_kernel))) return rc;
    if (g_knobs.turn_clock) g_turn.on_launched();
    pairing_valid_ = true;
    seed_valid_ = true;
    if (n_pairs) {
        unsigned int* counter = reinterpret_cast<unsigned int*>(acc_dev_.as<double>() + kNAcc);
        unsigned int* hc = reinterpret_cast<unsigned int*>(acc_host_ + kNAcc);
        if (pairing_sorted_) {  // the tiled kernels do not count: do it now
            HIPCHK(hipMemsetAsync(counter, 0, sizeof(unsigned int), stream_));
            hipLaunchKernelGGL(k_count_kept, dim3((unsigned)std::min<size_t>((N_ + 255) / 256, 256)), dim3(256), 0, stream_, ts_idx_.as<int>(),
                               (int)N_, counter);
            HIPCHK(hipGetLastError());
            counters_clean_ = false;   // (the count stays in counter[0]: a dense kernel's next launch must find it zero)
        }
        HIPCHK(hipMemcpyAsync(hc, counter, sizeof(unsigned int), hipMemcpyDeviceToHost, stream_));
        HIPCHK(hipStreamSynchronize(stream_));
        if (bbox_pending_) { const int rcb = check_bboxes(); if (rcb) return rcb; }
        *n_pairs = *hc;
    }
    return MOLA_ICP_OK;
}

// Row e: the map is only the part a box holds -- every gated neighbour of a moved query must lie inside it.
int HipWorkspace::check_slab(const Mat4& T, double threshold)
{
    if (!slab_active_ || N_ == 0) return MOLA_ICP_OK;
    double lo[3], hi[3];
    const int rc = shard_reach_box(T, threshold, lo, hi);
    if (rc) return rc;
    bool out = false;
    for (int a = 0; a < 3; ++a) out |= lo[a] < slab_lo_[a] || hi[a] > slab_hi_[a];
    // Sharded over ranks: a rank that returned here would stop joining the accumulator all-reduce and leave its peers
    // waiting.  It keeps going instead and poisons its pair count (NaN), which the sum carries to EVERY rank: they all
    // fail together after the collective (accumulate / accumulate_planes / allreduce).
    if (out && !comm_ && !ar_fn_) return fail(MOLA_ICP_E_BADARG, kSlabMsg);
    slab_violation_ = slab_violation_ || out;
    return MOLA_ICP_OK;
}

// waits for `seq` in a pinned flag written by the device (k_reduce_partials / k_publish); a real stream wait as fallback
// Wait policy (MOLA_ICP_WAIT, mola_icp_set_wait_policy): spin -- the default: the lowest latency, one host core at 100 % per in-flight
// align (the reference runs max(2, hw/2) pool threads plus the odometry thread on one ICP object: src/LidarOdometry.cpp:94-96, 869);
// yield -- spin for ~10 us, then sched_yield() between polls: the core is shared with whoever is runnable; block -- spin briefly,
// then SLEEP between polls (no core is burnt; every hand-over pays the timer's wake-up, ~55-60 us).  Costs per iteration: INTEGRATION.md.
void set_wait_policy(int policy) { g_knobs.wait_policy = policy == 1 ? 1 : (policy == 2 ? 2 : 0); }
int wait_policy() { return g_knobs.wait_policy; }

int HipWorkspace::spin_for(volatile unsigned long long* flag, unsigned long long seq)
{
    const int policy = g_knobs.wait_policy;
    const unsigned long long spin_limit = policy == 0 ? 400000000ull /* ~ seconds */ : (policy == 1 ? 4000ull : 600ull);
    for (unsigned long long spins = 0; spins < spin_limit; ++spins) {
        if (*flag == seq) {
            std::atomic_thread_fence(std::memory_order_acquire);
            if (g_knobs.turn_clock) g_turn.on_seen();
            // (the block was published behind everything enqueued before it: a bounding box copied for the host has landed too)
            return bbox_pending_ ? check_bboxes() : MOLA_ICP_OK;
        }
        __builtin_ia32_pause();
    }
    if (policy == 1) {   // yield: the device has a bounded amount of work in front of the flag; a stream wait ends a hang
        const auto t0 = std::chrono::steady_clock::now();
        while (*flag != seq) {
            sched_yield();
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) break;
        }
        if (*flag == seq) {
            std::atomic_thread_fence(std::memory_order_acquire);
            if (g_knobs.turn_clock) g_turn.on_seen();
            return bbox_pending_ ? check_bboxes() : MOLA_ICP_OK;
        }
    } else if (policy == 2) {
        // block: the thread SLEEPS between polls (20 us requested; the kernel's timer slack makes it ~55-60).  The first form -- a
        // blocking-sync HIP event recorded behind the publishing kernel + hipEventSynchronize -- cost +6 us per iteration and left the calling
        // thread's CPU share at 1.00: on ROCm 7.2 that wait polls too (profiles/r06/wait_policy.txt).
        const auto t0 = std::chrono::steady_clock::now();
        while (*flag != seq) {
            std::this_thread::sleep_for(std::chrono::microseconds(20));
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) break;
        }
        if (*flag == seq) {
            std::atomic_thread_fence(std::memory_order_acquire);
            if (g_knobs.turn_clock) g_turn.on_seen();
            return bbox_pending_ ? check_bboxes() : MOLA_ICP_OK;
        }
    }
    HIPCHK(hipStreamSynchronize(stream_));
    if (*flag != seq) return fail(MOLA_ICP_E_HIP, "the device did not publish its result block");
    std::atomic_thread_fence(std::memory_order_acquire);
    return bbox_pending_ ? check_bboxes() : MOLA_ICP_OK;
}

int HipWorkspace::accumulate(const mola_icp_params& p, const Mat4& Tcur, int stage, const double cl[3],
                             const double cg[3], bool reset_outliers, double acc[kNAcc])
{
    int rc = init();
    if (rc) return rc;
    if (!pairing_valid_) return fail(MOLA_ICP_E_BADARG, "accumulate() called before match()");
    if (stage != 0 && stage != 1) return fail(MOLA_ICP_E_BADARG, "stage must be 0 or 1");
    if (stage == 1 && (!cl || !cg)) return fail(MOLA_ICP_E_BADARG, "stage 1 needs the centroids");
    HIPCHK(hipSetDevice(device_));
    if (N_ == 0 && !comm_) {
        for (int k = 0; k < kNAcc; ++k) acc[k] = 0;
        return MOLA_ICP_OK;
    }
    if (N_ == 0) {  // an empty shard still joins the all-reduce
        HIPCHK(hipMemsetAsync(acc_dev_.p, 0, sizeof(double) * kNAcc, stream_));
        const int rc2 = rccl_allreduce_sum_f64(comm_, acc_dev_.as<double>(), kNAcc, stream_);
        if (rc2) return rc2;
        HIPCHK(hipMemcpyAsync(acc_host_, acc_dev_.p, sizeof(double) * kNAcc, hipMemcpyDeviceToHost, stream_));
        HIPCHK(hipStreamSynchronize(stream_));
        if (bbox_pending_) { const int rcb = check_bboxes(); if (rcb) return rcb; }
        std::memcpy(acc, acc_host_, sizeof(double) * kNAcc);
        return MOLA_ICP_OK;
    }
    if (reset_outliers && (outliers_dirty_ || outlier_cleared_for_ != N_)) {  // nothing sets a flag unless stage 1 ran
        HIPCHK(hipMemsetAsync(outlier_.p, 0, N_, stream_));
        outliers_dirty_ = false;
        outlier_cleared_for_ = N_;
    }
    if (stage == 1 && p.use_scale_outlier_detector) outliers_dirty_ = true;
    int nblocks = (int)((N_ + kAccThreads - 1) / kAccThreads);
    if (nblocks > kAccMaxBlocks) nblocks = kAccMaxBlocks;
    if ((rc = partials_.reserve(sizeof(double) * kNAcc * kAccMaxBlocks))) return rc;
    AccArgs a{};
    if (pairing_sorted_) {  // tiled matcher: sorted local cloud, neighbour = sorted-map position (local gathers)
        const float* sl = loc_sc_->sorted.as<float>();
        const float* sm = map_sc_->sorted.as<float>();
        a.lx = sl; a.ly = sl + loc_sc_->padded; a.lz = sl + 2 * loc_sc_->padded;
        a.gx = sm; a.gy = sm + map_sc_->padded; a.gz = sm + 2 * map_sc_->padded;
        a.idx = ts_pos_.as<int>(); a.d2 = ts_d2_.as<float>();
        a.nx = ts_gs_.as<float>(); a.ny = a.nx + loc_sc_->padded; a.nz = a.nx + 2 * loc_sc_->padded;
    } else {
        a.lx = lx_; a.ly = ly_; a.lz = lz_; a.gx = gx_; a.gy = gy_; a.gz = gz_;
        a.idx = idx_.as<int>(); a.d2 = d2_.as<float>();
    }
    a.outlier = outlier_.as<unsigned char>();
    a.N = (int)N_; a.stage = stage;
    a.use_scale = p.use_scale_outlier_detector; a.use_robust = p.use_robust_kernel;
    a.scale_thr = p.scale_outlier_threshold; a.rk_param = p.robust_kernel_param; a.rk_scale = p.robust_kernel_scale;
    for (int k = 0; k < 3; ++k) { a.cl[k] = cl ? cl[k] : 0.0; a.cg[k] = cg ? cg[k] : 0.0; }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) a.R[3 * r + c] = Tcur(r, c);
    // the first pass of an iteration after a cooperative match: the matcher already summed the unit-weight terms per item
    const bool fused = pairing_sorted_ && rows_valid_ && stage == 0 && reset_outliers;
    const double* rows = partials_.as<double>();
    bool rows_final = false;
    if (fused) {
        // the matchers' item rows (one per 64 queries: thousands): G blocks sum a slice each; on a single GPU they publish their G
        // rows to the pinned block themselves and THIS thread adds them in order -- else the one-block reduction below takes them
        const int G = item_red_blocks(rows_count_);
        if ((rc = item_part_.reserve(sizeof(double) * kNAcc * kItemRedBlocks))) return rc;
        if (!item_part_host_) {
            HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&item_part_host_), sizeof(double) * 32 * kItemRedBlocks, hipHostMallocMapped | hipHostMallocCoherent));
            std::memset(item_part_host_, 0, sizeof(double) * 32 * kItemRedBlocks);
        }
        const bool direct_items = !comm_ && !g_knobs.no_direct_readback;
        const unsigned long long seq_i = ++readback_seq_;
        // (sharded over RCCL: the last block to finish also adds the G rows into the device block the collective runs on)
        unsigned int* done_word = reinterpret_cast<unsigned int*>(acc_dev_.as<double>() + kNAcc + 6);
        hipLaunchKernelGGL(k_reduce_items, dim3(G), dim3(kNAcc * kRedSlices), 0, stream_, rows_.as<double>(), rows_count_, item_part_.as<double>(),
                           direct_items ? item_part_host_ : (double*)nullptr, seq_i, acc_dev_.as<double>(),
                           comm_ ? acc_dev_.as<double>() : (double*)nullptr, comm_ ? done_word : (unsigned int*)nullptr);
        HIPCHK(hipGetLastError());
        counters_clean_ = true;
        if (direct_items) {
            for (int k = 0; k < kNAcc; ++k) acc[k] = 0.0;
            for (int g = 0; g < G; ++g) {
                const double* row = item_part_host_ + 32 * (size_t)g;
                if ((rc = spin_for(reinterpret_cast<volatile unsigned long long*>(const_cast<double*>(row)) + kNAcc + 6, seq_i))) return rc;
                for (int k = 0; k < kNAcc; ++k) acc[k] += row[k];
            }
            if (ar_fn_ && slab_violation_) acc[16] = std::nan("");  // the hook's sum carries it to every rank (allreduce below)
            return MOLA_ICP_OK;
        }
        rows = item_part_.as<double>();
        nblocks = G;
        rows_final = comm_ != nullptr;   // (acc_dev_ holds the sums already: k_reduce_items' last block)
    } else {
        hipLaunchKernelGGL(k_accumulate, dim3(nblocks), dim3(kAccThreads), 0, stream_, a, partials_.as<double>());
        HIPCHK(hipGetLastError());
    }
    // Single GPU: the reduction writes the 24 sums straight into the pinned host block and then a sequence number;
    // the host spins on that number instead of a copy + stream synchronisation (both cost a launch gap and the
    // driver's wake-up latency on a ~0.2 ms iteration).  Sharded over RCCL: the collective runs in between on the
    // device block, then the block is copied.
    const bool direct = !comm_ && !g_knobs.no_direct_readback;
    const unsigned long long seq = ++readback_seq_;
    if (!rows_final) {
        hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(kNAcc * kRedSlices), 0, stream_, rows, nblocks,
                           acc_dev_.as<double>(), direct ? acc_host_ : (double*)nullptr, seq);
        HIPCHK(hipGetLastError());
    }
    counters_clean_ = true;
    volatile unsigned long long* flag = reinterpret_cast<volatile unsigned long long*>(acc_host_) + kNAcc + 6;
    if (!direct) {
        if (comm_) {  // query-sharded: the one collective of the path, in place on the device block (RCCL over xGMI)
            if (slab_violation_) {  // see match(): every rank learns of it through the sum
                acc_host_[kNAcc + 7] = std::nan("");
                HIPCHK(hipMemcpyAsync(acc_dev_.as<double>() + 16, acc_host_ + kNAcc + 7, sizeof(double), hipMemcpyHostToDevice, stream_));
            }
            const int rc2 = rccl_allreduce_sum_f64(comm_, acc_dev_.as<double>(), kNAcc, stream_);
            if (rc2) return rc2;
        }
        if (g_knobs.no_direct_readback) {
            HIPCHK(hipMemcpyAsync(acc_host_, acc_dev_.p, sizeof(double) * kNAcc, hipMemcpyDeviceToHost, stream_));
            HIPCHK(hipStreamSynchronize(stream_));
            if (bbox_pending_) { const int rcb = check_bboxes(); if (rcb) return rcb; }
            std::memcpy(acc, acc_host_, sizeof(double) * kNAcc);
            return MOLA_ICP_OK;
        }
        hipLaunchKernelGGL(k_publish, dim3(1), dim3(128), 0, stream_, acc_dev_.as<double>(), kNAcc, acc_host_, kNAcc + 6, seq);
        HIPCHK(hipGetLastError());
    }
    if ((rc = spin_for(flag, seq))) return rc;
    for (int k = 0; k < kNAcc; ++k) acc[k] = acc_host_[k];
    if (comm_ && std::isnan(acc[16])) { slab_violation_ = false; return fail(MOLA_ICP_E_BADARG, kSlabMsg); }
    if (ar_fn_ && slab_violation_) acc[16] = std::nan("");  // the hook's sum carries it to every rank (allreduce below)
    return MOLA_ICP_OK;
}

int HipWorkspace::allreduce(double acc[kNAcc])
{
    if (comm_) return MOLA_ICP_OK;  // accumulate() already reduced the device block over RCCL
    if (!ar_fn_) return MOLA_ICP_OK;
    const std::string err0 = last_error();
    const int rc = ar_fn_(acc, kNAcc, 0, ar_user_);
    if (rc) return hook_failed(rc, err0);
    if (std::isnan(acc[16])) { slab_violation_ = false; return fail(MOLA_ICP_E_BADARG, kSlabMsg); }
    return MOLA_ICP_OK;
}

// The PairedRatio count from the plane matcher's certified lists (k_quality_from_lists): available right behind a point-to-plane loop
// on the clouds in place; any query the lists cannot decide sends the caller back to the matcher pass (same count either way).
int HipWorkspace::quality_pairs(const Mat4& T, double threshold, const mola_icp_params& p, double acc[kNAcc], bool* done)
{
    *done = false;
    // Sharded over a native RCCL communicator the count must go through the collective accumulate() issues (allreduce() is a no-op
    // there), and EVERY rank must issue it: a rank answering from its lists while another -- undecided queries, an empty shard --
    // falls back to match() + accumulate() would leave that one alone in ncclAllReduce.  So with comm_ the matcher pass answers on
    // all ranks.  (The hook / node-local transports sum in allreduce(), which every rank calls once per pass either way.)
    if (comm_) return MOLA_ICP_OK;
    if (!inited_ || !knn_seed_valid_ || planes_knn_ < 1 || planes_empty_ || g_knobs.no_certify || g_knobs.no_knn_seed || g_knobs.no_quality_lists ||
        !loc_sc_ || !loc_sc_->ready || N_ == 0 || g_knobs.no_direct_readback)
        return MOLA_ICP_OK;
    (void)p;
    int rc;
    HIPCHK(hipSetDevice(device_));
    // a shard's map slab must hold the reach of the FINAL pose too (the last solve may have moved it out): the count below would
    // silently be one against a truncated map.  A violation goes the way match() sends it: the matcher pass runs, poisons the pair
    // count, every rank fails together and the caller re-cuts.
    if ((rc = check_slab(T, threshold))) return rc;
    if (slab_violation_) return MOLA_ICP_OK;
    if (!quality_host_) {   // (a block of its own: a record's data word must never sit where another kernel's sequence flag is awaited)
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&quality_host_), sizeof(unsigned long long) * 2 * kQualityBlocks, hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(quality_host_, 0, sizeof(unsigned long long) * 2 * kQualityBlocks);
    }
    PoseF P, Pprev;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) P.R[3 * r + c] = (float)T(r, c);
        P.t[r] = (float)T(r, 3);
    }
    for (int k = 0; k < 9; ++k) Pprev.R[k] = knn_last_P_[k];
    for (int k = 0; k < 3; ++k) Pprev.t[k] = knn_last_P_[9 + k];
    const float thr2 = (float)(threshold * threshold);
    const int G = (int)std::min<size_t>((N_ + 1023) / 1024, (size_t)kQualityBlocks);
    const unsigned long long seq = ++readback_seq_;
    const float* sl = loc_sc_->sorted.as<float>();
    unsigned long long* recs = quality_host_;
    const KnnSeeds seeds = knn_seeds_at(knn_pos_.p, loc_sc_->padded, planes_knn_ + 1);
#define MOLA_QUALITY_CASE(KL_)                                                                                                              \
    case KL_:                                                                                                                               \
        hipLaunchKernelGGL(k_quality_from_lists<KL_>, dim3(G), dim3(256), 0, stream_, sl, sl + loc_sc_->padded, sl + 2 * loc_sc_->padded, (int)N_, \
                           P, Pprev, thr2, seeds, knn_lb_.as<float>(), recs, seq);                                                          \
        break;
    switch (planes_knn_ + 1) {
        MOLA_QUALITY_CASE(4) MOLA_QUALITY_CASE(5) MOLA_QUALITY_CASE(6) MOLA_QUALITY_CASE(7) MOLA_QUALITY_CASE(8) MOLA_QUALITY_CASE(9)
        default: return MOLA_ICP_OK;   // (no such list length: the matcher pass answers)
    }
#undef MOLA_QUALITY_CASE
    HIPCHK(hipGetLastError());
    double pairs = 0.0, open = 0.0;
    for (int g = 0; g < G; ++g) {
        if ((rc = spin_for(recs + 2 * (size_t)g + 1, seq))) return rc;
        const unsigned long long v = const_cast<volatile unsigned long long*>(recs)[2 * (size_t)g];
        pairs += (double)(unsigned int)(v & 0xffffffffull);
        open += (double)(unsigned int)(v >> 32);
    }
    if (open > 0.0) return MOLA_ICP_OK;   // undecided queries: the matcher pass answers (exactly the same count)
    for (int k = 0; k < kNAcc; ++k) acc[k] = 0.0;
    acc[16] = pairs;
    *done = true;
    return MOLA_ICP_OK;
}

int HipWorkspace::copy_pairing(int32_t* idx_out, float* d2_out)
{
    if (!pairing_valid_) return fail(MOLA_ICP_E_BADARG, "no pairing stored: call match() first");
    HIPCHK(hipSetDevice(device_));
    if (N_ && pairing_sorted_) {
        hipLaunchKernelGGL(k_unpermute_pairing, dim3((unsigned)((N_ + 255) / 256)), dim3(256), 0, stream_,
                           loc_sc_->perm.as<int>(), ts_idx_.as<int>(), ts_d2_.as<float>(), (int)N_, idx_.as<int>(),
                           d2_.as<float>());
        HIPCHK(hipGetLastError());
    }
    if (N_) {
        const int* src = idx_.as<int>();
        if (idx_out && slab_active_) {  // slab point -> ORIGINAL map index (idx_ itself stays: it seeds the dense kernels)
            int rc = stage_in_.reserve(sizeof(int) * N_);
            if (rc) return rc;
            hipLaunchKernelGGL(k_remap_indices, dim3((unsigned)((N_ + 255) / 256)), dim3(256), 0, stream_, idx_.as<int>(),
                               slab_orig_.as<int>(), (int)N_, stage_in_.as<int>());
            HIPCHK(hipGetLastError());
            src = stage_in_.as<int>();
        }
        if (idx_out) HIPCHK(hipMemcpyAsync(idx_out, src, sizeof(int) * N_, hipMemcpyDeviceToHost, stream_));
        if (d2_out) HIPCHK(hipMemcpyAsync(d2_out, d2_.p, sizeof(float) * N_, hipMemcpyDeviceToHost, stream_));
    }
    HIPCHK(hipStreamSynchronize(stream_));
    if (bbox_pending_) { const int rcb = check_bboxes(); if (rcb) return rcb; }
    return MOLA_ICP_OK;
}

// ------------------------------------------------------------------ HipBatch: K problems per launch
static TiledMap tiled_map_of(const SortedCloud& sc)
{
    const float* sx = sc.sorted.as<float>();
    return TiledMap{sx, sx + sc.padded, sx + 2 * sc.padded, sc.perm.as<int>(), sc.tbox.as<float>(), sc.n_tiles_p,
                    sc.sbox.as<float>(), sc.n_super, sc.ubox.as<float>(), sc.n_top};
}

void BatchScratch::release_all()
{
    for (BatchBuffers& b : bufs) {
        b.pos.release(); b.idx.release(); b.d2.release(); b.gs.release(); b.rows.release(); b.outlier.release(); b.partials.release();
        b.planes.release(); b.plane_cache.release(); b.knn_pos.release(); b.knn_lb.release(); b.plane_partials.release();
    }
    bufs.clear();
    acc_dev.release(); stats.release(); queue.release(); plane_acc_dev.release(); item_part_dev.release();
    if (item_part_host) (void)hipHostFree(item_part_host);
    item_part_host = nullptr; item_part_host_problems = 0;
    if (plane_acc_host) (void)hipHostFree(plane_acc_host);
    plane_acc_host = nullptr; plane_acc_host_problems = 0;
    if (acc_host) (void)hipHostFree(acc_host);
    if (stats_host) (void)hipHostFree(stats_host);
    acc_host = nullptr; stats_host = nullptr; acc_host_problems = 0;
    for (void* e : events) (void)hipEventDestroy(static_cast<hipEvent_t>(e));
    events.clear();
}

HipBatch::HipBatch(HipWorkspace& ws, std::vector<BatchProblem> probs) : ws_(ws), sc_(ws.batch_scratch_), probs_(std::move(probs)) {}

HipBatch::~HipBatch()
{
    if (!inited_) return;
    (void)hipSetDevice(ws_.device_);
    (void)hipStreamSynchronize(ws_.stream_);  // nothing of this batch is in flight when the scratch is used again
}

int HipBatch::init()
{
    if (inited_) return MOLA_ICP_OK;
    int rc = ws_.init();
    if (rc) return rc;
    HIPCHK(hipSetDevice(ws_.device_));
    const size_t K = probs_.size();
    if (sc_.bufs.size() < K) sc_.bufs.resize(K);
    for (size_t k = 0; k < K; ++k) {
        const BatchProblem& pr = probs_[k];
        if (!pr.map || !pr.loc) return fail(MOLA_ICP_E_BADARG, "batch problem without clouds");
        Buffers& b = sc_.bufs[k];
        b.seed_valid = false;
        b.outliers_dirty = true;   // (contents of a reused flag buffer are another problem's)
        b.outlier_cleared_for = 0;
        if (pr.loc->n == 0 || pr.map->n == 0) continue;  // never matched (the loop ends such a problem at once)
        if (!pr.map->ready || !pr.loc->ready) return fail(MOLA_ICP_E_INTERNAL, "batch problem with an unprepared cloud");
        const size_t np = (pr.loc->n + kQPW - 1) / kQPW * kQPW;
        if ((rc = b.pos.reserve(sizeof(int) * np))) return rc;
        if ((rc = b.idx.reserve(sizeof(int) * np))) return rc;
        if ((rc = b.d2.reserve(sizeof(float) * np))) return rc;
        if ((rc = b.gs.reserve(sizeof(float) * 3 * np))) return rc;
        if ((rc = b.rows.reserve(sizeof(double) * kNAcc * (np / 64)))) return rc;   // one row per 64 queries
        if ((rc = b.outlier.reserve(pr.loc->n))) return rc;
        if ((rc = b.partials.reserve(sizeof(double) * kNAcc * kAccMaxBlocks))) return rc;
    }
    if ((rc = sc_.acc_dev.reserve(sizeof(double) * 32 * (K ? K : 1)))) return rc;
    if ((rc = sc_.stats.reserve(sizeof(unsigned long long) * kStatSlots * kStatStride))) return rc;
    if ((rc = sc_.queue.reserve(sizeof(unsigned int) * kQueues * kQueueStride))) return rc;
    HIPCHK(hipMemsetAsync(sc_.stats.p, 0, sizeof(unsigned long long) * kStatSlots * kStatStride, ws_.stream_));
    if (sc_.acc_host_problems < (K ? K : 1)) {
        if (sc_.acc_host) (void)hipHostFree(sc_.acc_host);
        sc_.acc_host = nullptr;
        const size_t cap = (K ? K : 1) < 16 ? 16 : (K ? K : 1);
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&sc_.acc_host), sizeof(double) * 32 * cap, hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(sc_.acc_host, 0, sizeof(double) * 32 * cap);
        sc_.acc_host_problems = cap;
    }
    if (!sc_.stats_host)
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&sc_.stats_host), sizeof(unsigned long long) * kStatSlots * kStatStride, hipHostMallocDefault));
    if ((rc = sc_.item_part_dev.reserve(sizeof(double) * kNAcc * kItemRedBlocks * (K ? K : 1)))) return rc;
    if (sc_.item_part_host_problems < (K ? K : 1)) {
        if (sc_.item_part_host) (void)hipHostFree(sc_.item_part_host);
        sc_.item_part_host = nullptr;
        const size_t cap = (K ? K : 1) < 16 ? 16 : (K ? K : 1);
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&sc_.item_part_host), sizeof(double) * 32 * kItemRedBlocks * cap, hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(sc_.item_part_host, 0, sizeof(double) * 32 * kItemRedBlocks * cap);
        sc_.item_part_host_problems = cap;
    }
    inited_ = true;
    return MOLA_ICP_OK;
}

int HipBatch::match(const uint8_t* active, const Mat4* T, double threshold, const mola_icp_params& p)
{
    (void)p;
    int rc = init();
    if (rc) return rc;
    if (!(threshold > 0)) return fail(MOLA_ICP_E_BADARG, "matcher threshold must be > 0");
    HIPCHK(hipSetDevice(ws_.device_));
    const float thr2 = (float)(threshold * threshold);
    while (sc_.events.size() < ev_used_ + 2) {
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        sc_.events.push_back(e);
    }
    ++nn_launches_;
    if (ws_.profiling_) HIPCHK(hipEventRecord(static_cast<hipEvent_t>(sc_.events[ev_used_]), ws_.stream_));
    const int K = (int)probs_.size();
    for (int k0 = 0; k0 < K;) {  // chunks of up to kCoopMaxBatch active problems per launch
        NnBatch<kCoopMaxBatch> b;
        NnBatchItems<kCoopMaxBatch> bi;
        std::memset(&b, 0, sizeof b);
        std::memset(&bi, 0, sizeof bi);
        int n = 0, max_items = 0, max_items_t = 0, total_items = 0, total_items_t = 0;
        size_t max_box_bytes = 0;
        bool same_map = true;
        const SortedCloud* first_map = nullptr;
        for (; k0 < K && n < kCoopMaxBatch; ++k0) {
            if (!active[k0]) continue;
            const BatchProblem& pr = probs_[(size_t)k0];
            Buffers& bf = sc_.bufs[(size_t)k0];
            if (pr.loc->n == 0 || pr.map->n == 0) return fail(MOLA_ICP_E_INTERNAL, "empty problem in a batched match");
            NnProblem& pb = b.p[n++];
            const float* sl = pr.loc->sorted.as<float>();
            pb.slx = sl; pb.sly = sl + pr.loc->padded; pb.slz = sl + 2 * pr.loc->padded;
            pb.N = (int)pr.loc->n;
            pb.mp = tiled_map_of(*pr.map);
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c) pb.P.R[3 * r + c] = (float)T[k0](r, c);
                pb.P.t[r] = (float)T[k0](r, 3);
            }
            pb.thr2 = thr2;
            pb.use_seed = (bf.seed_valid && !g_knobs.no_warm_start) ? 1 : 0;
            pb.pos_s = bf.pos.as<int>(); pb.idx_s = bf.idx.as<int>(); pb.d2_s = bf.d2.as<float>();
            {
                const size_t np = (pr.loc->n + kQPW - 1) / kQPW * kQPW;
                pb.gsx = bf.gs.as<float>(); pb.gsy = pb.gsx + np; pb.gsz = pb.gsx + 2 * np;
            }
            pb.rows = bf.rows.as<double>();
            pb.staged = sc_.stats.as<unsigned long long>();
            bf.seed_valid = true;
            const int items = (int)((pr.loc->n + kQPW - 1) / kQPW);   // 128-query items: the cooperative kernel's, and the rows'
            if (items > max_items) max_items = items;
            const int items_t = (int)((pr.loc->n + 63) / 64);         // 64-query items: the persistent batched matcher's, and k_nn_q4's workgroups
            if (items_t > max_items_t) max_items_t = items_t;
            bi.base[n - 1] = total_items_t;
            total_items += items;
            total_items_t += items_t;
            bi.base[n] = total_items_t;
            if (!first_map) first_map = pr.map.get();
            same_map = same_map && pr.map.get() == first_map;
            const size_t bb = sizeof(float) * 6u * ((size_t)pb.mp.n_top + (size_t)pb.mp.n_super);
            if (bb > max_box_bytes) max_box_bytes = bb;
        }
        if (n == 0) break;
        // Few items: latency counts -> one WORKGROUP per item (k_nn_coop, rows fused).  Many (a dozen 100k-point problems
        // are ~10^4): issue slots count -> the persistent one-wave-per-item matcher over all problems' items (rows fused too:
        // the same sums, bit for bit).  MOLA_ICP_BATCH_TILED=0|1 forces either.
        const bool tiled = g_knobs.batch_tiled >= 0 ? g_knobs.batch_tiled != 0 : total_items >= 2 * 1024;
        const int lds_boxes = max_box_bytes <= lds_box_limit(tiled ? persistent_static_lds() : nn_coop_static_lds(), 4) ? 1 : 0;
        const size_t dyn_lds = lds_boxes ? max_box_bytes : 0;
        const bool q4 = !tiled && (g_knobs.q4 >= 0 ? g_knobs.q4 != 0 : true);   // (four lanes per query: see HipWorkspace::launch_nn)
        if (q4) {
            const int lb = q4_lds_boxes(max_box_bytes);
            HIPCHK(q4_launch_batch(ws_.stream_, b, xcd_grid(max_items_t), n, lb ? max_box_bytes : 0, lb, ws_.profiling_ ? 1 : 0));
        } else if (!tiled) {
            hipLaunchKernelGGL((k_nn_coop<kCoopMaxBatch>), dim3(xcd_grid(max_items), n), dim3(256), dyn_lds, ws_.stream_, b, lds_boxes,
                               (unsigned long long*)nullptr);
            HIPCHK(hipGetLastError());
        } else {
            const int shared = (same_map && lds_boxes) ? 1 : 0;
            const size_t lds = shared ? dyn_lds : 0;
            int& fit_tiled_ = sc_.fit_tiled;
            size_t& fit_tiled_lds_ = sc_.fit_tiled_lds;
            if (fit_tiled_ == 0 || fit_tiled_lds_ != lds) {  // persistent waves with fixed first entries: the whole grid must be resident
                HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit_tiled_, (k_nn_tiled_batch<kCoopMaxBatch, 1>), 256, lds));
                fit_tiled_lds_ = lds;
            }
            int per_cu = g_knobs.blocks_per_cu > 0 ? g_knobs.blocks_per_cu : 4;   // (64-query items, four workgroups per CU: see launch_tiled)
            if (fit_tiled_ >= 1 && per_cu > fit_tiled_) per_cu = fit_tiled_;
            int grid = ws_.num_cus_ * per_cu;
            if (grid > (total_items_t + 3) / 4) grid = (total_items_t + 3) / 4;
            HIPCHK(hipMemsetAsync(sc_.queue.p, 0, sizeof(unsigned int) * kQueues * kQueueStride, ws_.stream_));
            hipLaunchKernelGGL((k_nn_tiled_batch<kCoopMaxBatch, 1>), dim3(grid), dim3(256), lds, ws_.stream_, b, bi, n, shared,
                               sc_.queue.as<unsigned int>(), g_knobs.early_pop ? 1 : 0);
            HIPCHK(hipGetLastError());
        }
    }
    if (ws_.profiling_) {
        HIPCHK(hipEventRecord(static_cast<hipEvent_t>(sc_.events[ev_used_ + 1]), ws_.stream_));
        ev_used_ += 2;
    }
    return MOLA_ICP_OK;
}

int HipBatch::accumulate(const uint8_t* active, const mola_icp_params& p, const Mat4* Tcur, int stage, const double (*cl)[3],
                         const double (*cg)[3], bool reset_outliers, double (*acc)[kNAcc])
{
    int rc = init();
    if (rc) return rc;
    if (stage != 0 && stage != 1) return fail(MOLA_ICP_E_BADARG, "stage must be 0 or 1");
    if (stage == 1 && (!cl || !cg)) return fail(MOLA_ICP_E_BADARG, "stage 1 needs the centroids");
    HIPCHK(hipSetDevice(ws_.device_));
    const int K = (int)probs_.size();
    const unsigned long long seq = ++sc_.seq;
    const bool fused = stage == 0 && reset_outliers;  // (every batched match writes its item rows)
    if (fused) {
        // per problem the partition and the order of HipWorkspace::accumulate's k_reduce_items + host sum: the same bits
        for (int k0 = 0; k0 < K;) {
            ReduceItemsBatch rb;
            std::memset(&rb, 0, sizeof rb);
            int n = 0;
            for (; k0 < K && n < kAccMaxBatch; ++k0) {
                if (!active[k0]) continue;
                Buffers& bf = sc_.bufs[(size_t)k0];
                if (!bf.seed_valid) return fail(MOLA_ICP_E_BADARG, "batched accumulate() before match()");
                const size_t N = probs_[(size_t)k0].loc->n;
                if (bf.outliers_dirty || bf.outlier_cleared_for != N) {
                    HIPCHK(hipMemsetAsync(bf.outlier.p, 0, N, ws_.stream_));
                    bf.outliers_dirty = false;
                    bf.outlier_cleared_for = N;
                }
                rb.rows[n] = bf.rows.as<double>();
                rb.n_rows[n] = (int)((N + 63) / 64);
                rb.slot[n] = k0;
                ++n;
            }
            if (n == 0) break;
            hipLaunchKernelGGL(k_reduce_items_batch, dim3(kItemRedBlocks, n), dim3(kNAcc * kRedSlices), 0, ws_.stream_, rb,
                               sc_.item_part_dev.as<double>(), sc_.item_part_host, seq);
            HIPCHK(hipGetLastError());
        }
        for (int k = 0; k < K; ++k) {
            if (!active[k]) continue;
            const int G = item_red_blocks((int)((probs_[(size_t)k].loc->n + 63) / 64));
            for (int c = 0; c < kNAcc; ++c) acc[k][c] = 0.0;
            for (int g = 0; g < G; ++g) {
                const double* row = sc_.item_part_host + 32 * ((size_t)kItemRedBlocks * (size_t)k + (size_t)g);
                if ((rc = ws_.spin_for(reinterpret_cast<volatile unsigned long long*>(const_cast<double*>(row)) + kNAcc + 6, seq))) return rc;
                for (int c = 0; c < kNAcc; ++c) acc[k][c] += row[c];
            }
        }
        return MOLA_ICP_OK;
    }
    int n_active = 0;
    for (int k0 = 0; k0 < K;) {
        AccBatch ab;
        ReduceBatch rb;
        std::memset(&ab, 0, sizeof ab);
        std::memset(&rb, 0, sizeof rb);
        int n = 0, max_blocks = 0;
        for (; k0 < K && n < kAccMaxBatch; ++k0) {
            if (!active[k0]) continue;
            const BatchProblem& pr = probs_[(size_t)k0];
            Buffers& bf = sc_.bufs[(size_t)k0];
            if (!bf.seed_valid) return fail(MOLA_ICP_E_BADARG, "batched accumulate() before match()");
            const size_t N = pr.loc->n;
            if (reset_outliers && (bf.outliers_dirty || bf.outlier_cleared_for != N)) {
                HIPCHK(hipMemsetAsync(bf.outlier.p, 0, N, ws_.stream_));
                bf.outliers_dirty = false;
                bf.outlier_cleared_for = N;
            }
            if (stage == 1 && p.use_scale_outlier_detector) bf.outliers_dirty = true;
            int nblocks = (int)((N + kAccThreads - 1) / kAccThreads);
            if (nblocks > kAccMaxBlocks) nblocks = kAccMaxBlocks;
            AccArgs& a = ab.a[n];
            const float* sl = pr.loc->sorted.as<float>();
            const float* sm = pr.map->sorted.as<float>();
            a.lx = sl; a.ly = sl + pr.loc->padded; a.lz = sl + 2 * pr.loc->padded;
            a.gx = sm; a.gy = sm + pr.map->padded; a.gz = sm + 2 * pr.map->padded;
            a.idx = bf.pos.as<int>(); a.d2 = bf.d2.as<float>();
            {
                const size_t np = (N + kQPW - 1) / kQPW * kQPW;
                a.nx = bf.gs.as<float>(); a.ny = a.nx + np; a.nz = a.nx + 2 * np;
            }
            a.outlier = bf.outlier.as<unsigned char>();
            a.N = (int)N; a.stage = stage;
            a.use_scale = p.use_scale_outlier_detector; a.use_robust = p.use_robust_kernel;
            a.scale_thr = p.scale_outlier_threshold; a.rk_param = p.robust_kernel_param; a.rk_scale = p.robust_kernel_scale;
            for (int c = 0; c < 3; ++c) { a.cl[c] = cl ? cl[k0][c] : 0.0; a.cg[c] = cg ? cg[k0][c] : 0.0; }
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) a.R[3 * r + c] = Tcur[k0](r, c);
            ab.nblocks[n] = nblocks;
            ab.partials[n] = bf.partials.as<double>();
            rb.partials[n] = bf.partials.as<double>();
            rb.nblocks[n] = nblocks;
            rb.slot[n] = k0;
            if (nblocks > max_blocks) max_blocks = nblocks;
            ++n;
        }
        if (n == 0) break;
        n_active += n;
        hipLaunchKernelGGL(k_accumulate_batch, dim3(max_blocks, n), dim3(kAccThreads), 0, ws_.stream_, ab);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(k_reduce_partials_batch, dim3(n), dim3(kNAcc * kRedSlices), 0, ws_.stream_, rb, sc_.acc_dev.as<double>(),
                           sc_.acc_host, seq);
        HIPCHK(hipGetLastError());
    }
    if (n_active == 0) return MOLA_ICP_OK;
    for (int k = 0; k < K; ++k) {
        if (!active[k]) continue;
        volatile unsigned long long* flag = reinterpret_cast<volatile unsigned long long*>(sc_.acc_host + 32 * (size_t)k) + kNAcc + 6;
        if ((rc = ws_.spin_for(flag, seq))) return rc;
        for (int c = 0; c < kNAcc; ++c) acc[k][c] = sc_.acc_host[32 * (size_t)k + c];
    }
    return MOLA_ICP_OK;
}

// ---- the shipped pipeline, batched (row f3 in the lockstep loop) ----------------------------------------------------
int HipBatch::init_planes(int knn)
{
    int rc = init();
    if (rc) return rc;
    if (planes_inited_) return MOLA_ICP_OK;
    HIPCHK(hipSetDevice(ws_.device_));
    const size_t K = probs_.size();
    for (size_t k = 0; k < K; ++k) {
        const BatchProblem& pr = probs_[k];
        Buffers& b = sc_.bufs[k];
        b.knn_seed_valid = false;   // (a reused buffer holds another problem's lists)
        b.planes_valid = false;
        b.planes_knn = knn;
        b.planes_eig_thr = std::nan("");   // (never equal to a threshold: no cached planes)
        if (pr.loc->n == 0 || pr.map->n == 0) continue;
        const size_t np = pr.loc->padded;
        if ((rc = b.planes.reserve(sizeof(PlanePair) * np))) return rc;
        if ((rc = b.plane_cache.reserve(sizeof(PlanePair) * np))) return rc;
        if ((rc = b.knn_pos.reserve(knn_seeds_bytes(np, knn + 1)))) return rc;   // lists of knn + 1 entries (KnnSeeds)
        if ((rc = b.knn_lb.reserve(sizeof(float) * np))) return rc;
        if ((rc = b.plane_partials.reserve(sizeof(double) * kNAccPlane * 512))) return rc;
    }
    if ((rc = sc_.plane_acc_dev.reserve(sizeof(double) * kPlaneAccStride * (K ? K : 1)))) return rc;
    if (sc_.plane_acc_host_problems < (K ? K : 1)) {
        if (sc_.plane_acc_host) (void)hipHostFree(sc_.plane_acc_host);
        sc_.plane_acc_host = nullptr;
        const size_t cap = (K ? K : 1) < 16 ? 16 : (K ? K : 1);
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&sc_.plane_acc_host), sizeof(double) * kPlaneAccStride * cap, hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(sc_.plane_acc_host, 0, sizeof(double) * kPlaneAccStride * cap);
        sc_.plane_acc_host_problems = cap;
    }
    planes_inited_ = true;
    return MOLA_ICP_OK;
}

// One k_knn_coop launch per chunk of up to kKnnMaxBatch active problems (blockIdx.y = problem): per problem the kernel, the
// seeds, the certified lists and the plane cache of HipWorkspace::match_planes' cooperative path -- the lists are exact and the
// planes a function of the lists, so every problem's pairing is the one its stand-alone align computes, whichever kernel that uses.
int HipBatch::match_planes(const uint8_t* active, const Mat4* T, const mola_icp_params& p)
{
    if (p.knn < 3 || p.knn > 8) return fail(MOLA_ICP_E_UNSUPPORTED, "Matcher_Point2Plane: knn must be in [3, 8] in this build");
    int rc = init_planes((int)p.knn);
    if (rc) return rc;
    HIPCHK(hipSetDevice(ws_.device_));
    const float thr2 = (float)(p.matcher_threshold * p.matcher_threshold);
    const float thr2x = g_knobs.no_certify ? thr2 : thr2 * 1.21f;   // the lists' own gate (KnnCert)
    while (sc_.events.size() < ev_used_ + 2) {
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        sc_.events.push_back(e);
    }
    ++nn_launches_;
    if (ws_.profiling_) HIPCHK(hipEventRecord(static_cast<hipEvent_t>(sc_.events[ev_used_]), ws_.stream_));
    const int K = (int)probs_.size();
    for (int k0 = 0; k0 < K;) {
        KnnBatch<kKnnMaxBatch> kb;
        std::memset(&kb, 0, sizeof kb);
        int n = 0, max_items = 0;
        size_t max_box_bytes = 0;
        for (; k0 < K && n < kKnnMaxBatch; ++k0) {
            if (!active[k0]) continue;
            const BatchProblem& pr = probs_[(size_t)k0];
            Buffers& bf = sc_.bufs[(size_t)k0];
            if (pr.loc->n == 0 || pr.map->n == 0) return fail(MOLA_ICP_E_INTERNAL, "empty problem in a batched match");
            KnnProblem& kp = kb.p[n++];
            const float* sl = pr.loc->sorted.as<float>();
            kp.slx = sl; kp.sly = sl + pr.loc->padded; kp.slz = sl + 2 * pr.loc->padded;
            kp.N = (int)pr.loc->n;
            kp.mp = tiled_map_of(*pr.map);
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c) kp.P.R[3 * r + c] = (float)T[k0](r, c);
                kp.P.t[r] = (float)T[k0](r, 3);
            }
            for (int q = 0; q < 9; ++q) kp.Pprev.R[q] = bf.knn_last_P[q];
            for (int q = 0; q < 3; ++q) kp.Pprev.t[q] = bf.knn_last_P[9 + q];
            kp.out = bf.planes.as<PlanePair>(); kp.cache = bf.plane_cache.as<PlanePair>();
            kp.seeds = knn_seeds_at(bf.knn_pos.p, pr.loc->padded, (int)p.knn + 1);
            kp.lb = bf.knn_lb.as<float>();
            const int seed = (bf.knn_seed_valid && bf.planes_knn == (int)p.knn && !g_knobs.no_knn_seed) ? 1 : 0;
            kp.use_seed = seed;
            kp.use_cache = (seed && bf.planes_eig_thr == plane_eig_arg(p)) ? 1 : 0;
            kp.cert_on = (seed && !g_knobs.no_certify) ? 1 : 0;
            kp.changed_items = nullptr;   // (only the persistent kernel's counting flavour reads the count)
            for (int q = 0; q < 9; ++q) bf.knn_last_P[q] = kp.P.R[q];
            for (int q = 0; q < 3; ++q) bf.knn_last_P[9 + q] = kp.P.t[q];
            bf.knn_seed_valid = true;
            bf.planes_valid = true;
            bf.planes_knn = (int)p.knn;
            bf.planes_eig_thr = plane_eig_arg(p);
            const int items = (int)((pr.loc->n + 63) / 64);
            if (items > max_items) max_items = items;
            const size_t bb = sizeof(float) * 6u * ((size_t)kp.mp.n_top + (size_t)kp.mp.n_super);
            if (bb > max_box_bytes) max_box_bytes = bb;
        }
        if (n == 0) break;
        const bool knn_q4 = use_knn_q4((int)p.knn + 1);
        const int kq4_lpq = knn_q4 ? knn_q4_lanes_per_query((int)p.knn + 1, (size_t)max_items * (size_t)n, ws_.num_cus_, false) : 4;
        const int lds_boxes = max_box_bytes <= (knn_q4 ? knn_q4_lds_box_limit((int)p.knn + 1, kq4_lpq) : knn_coop_lds_box_limit((int)p.knn + 1)) ? 1 : 0;
        const size_t dyn_lds = lds_boxes ? max_box_bytes : 0;
        unsigned long long* staged = ws_.profiling_ ? sc_.stats.as<unsigned long long>() : nullptr;
#define MOLA_LAUNCH_KNN_COOP_B(KK)                                                                                          \
    hipLaunchKernelGGL((k_knn_coop<KK, kKnnMaxBatch>), dim3(xcd_grid(max_items), n), dim3(256), dyn_lds, ws_.stream_, kb, thr2, thr2x, \
                       p.matcher_threshold, plane_eig_arg(p), staged, lds_boxes, (unsigned long long*)nullptr, (unsigned long long*)nullptr)
        if (knn_q4) {
            HIPCHK(knn_q4_launch_batch(ws_.stream_, (int)p.knn + 1, kb, xcd_grid(max_items), n, dyn_lds, thr2, thr2x, p.matcher_threshold, plane_eig_arg(p), staged, lds_boxes, nullptr, kq4_lpq));
        } else
        switch (p.knn) {
            case 3: MOLA_LAUNCH_KNN_COOP_B(4); break;
            case 4: MOLA_LAUNCH_KNN_COOP_B(5); break;
            case 5: MOLA_LAUNCH_KNN_COOP_B(6); break;
            case 6: MOLA_LAUNCH_KNN_COOP_B(7); break;
            case 7: MOLA_LAUNCH_KNN_COOP_B(8); break;
            default: MOLA_LAUNCH_KNN_COOP_B(9); break;
        }
#undef MOLA_LAUNCH_KNN_COOP_B
        HIPCHK(hipGetLastError());
    }
    if (ws_.profiling_) {
        HIPCHK(hipEventRecord(static_cast<hipEvent_t>(sc_.events[ev_used_ + 1]), ws_.stream_));
        ev_used_ += 2;
    }
    return MOLA_ICP_OK;
}

// The plane form of every active problem: per problem k_accumulate_planes_mfma's partition and k_reduce_rows' order (same bits
// as the stand-alone accumulate_planes), one launch each over the chunk, one pinned read-back per problem.
int HipBatch::accumulate_planes(const uint8_t* active, double (*acc)[kNAccPlaneHost])
{
    int rc = init();
    if (rc) return rc;
    if (!planes_inited_) return fail(MOLA_ICP_E_BADARG, "batched accumulate_planes() before match_planes()");
    HIPCHK(hipSetDevice(ws_.device_));
    const int K = (int)probs_.size();
    const unsigned long long seq = ++sc_.seq;
    int n_active = 0;
    for (int k0 = 0; k0 < K;) {
        PlaneAccBatch ab;
        std::memset(&ab, 0, sizeof ab);
        int n = 0, max_blocks = 0;
        for (; k0 < K && n < kKnnMaxBatch; ++k0) {
            if (!active[k0]) continue;
            const BatchProblem& pr = probs_[(size_t)k0];
            Buffers& bf = sc_.bufs[(size_t)k0];
            if (!bf.planes_valid) return fail(MOLA_ICP_E_BADARG, "batched accumulate_planes() before match_planes()");
            const float* sl = pr.loc->sorted.as<float>();
            int nblocks = (int)((pr.loc->n + 255) / 256);
            if (nblocks > 512) nblocks = 512;
            ab.slx[n] = sl; ab.sly[n] = sl + pr.loc->padded; ab.slz[n] = sl + 2 * pr.loc->padded;
            ab.pairs[n] = bf.planes.as<PlanePair>();
            ab.partials[n] = bf.plane_partials.as<double>();
            ab.N[n] = (int)pr.loc->n; ab.nblocks[n] = nblocks; ab.slot[n] = k0;
            if (nblocks > max_blocks) max_blocks = nblocks;
            ++n;
        }
        if (n == 0) break;
        n_active += n;
        hipLaunchKernelGGL(k_accumulate_planes_mfma_batch, dim3(max_blocks, n), dim3(256), 0, ws_.stream_, ab);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(k_reduce_rows_batch, dim3(n), dim3(1024), 0, ws_.stream_, ab, kNAccPlane, sc_.plane_acc_dev.as<double>(),
                           sc_.plane_acc_host, seq);
        HIPCHK(hipGetLastError());
    }
    if (n_active == 0) return MOLA_ICP_OK;
    for (int k = 0; k < K; ++k) {
        if (!active[k]) continue;
        const double* hb = sc_.plane_acc_host + (size_t)kPlaneAccStride * (size_t)k;
        if ((rc = ws_.spin_for(reinterpret_cast<volatile unsigned long long*>(const_cast<double*>(hb)) + kNAccPlane + 2, seq))) return rc;
        std::memcpy(acc[k], hb, sizeof(double) * kNAccPlane);
    }
    return MOLA_ICP_OK;
}

int HipBatch::collect_stats(double* ms_total, uint32_t* launches, uint64_t* pairs)
{
    if (!inited_) { if (ms_total) *ms_total = 0; if (launches) *launches = 0; if (pairs) *pairs = 0; return MOLA_ICP_OK; }
    HIPCHK(hipSetDevice(ws_.device_));
    unsigned long long staged = 0;
    if (ws_.profiling_) {
        HIPCHK(hipMemcpyAsync(sc_.stats_host, sc_.stats.p, sizeof(unsigned long long) * kStatSlots * kStatStride, hipMemcpyDeviceToHost, ws_.stream_));
        HIPCHK(hipStreamSynchronize(ws_.stream_));
        for (int k = 0; k < kStatSlots; ++k) staged += sc_.stats_host[(size_t)k * kStatStride];
    }
    double tot = 0;
    for (size_t i = 0; i + 1 < ev_used_; i += 2) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, static_cast<hipEvent_t>(sc_.events[i]), static_cast<hipEvent_t>(sc_.events[i + 1])));
        tot += ms;
    }
    if (ms_total) *ms_total = tot;
    if (launches) *launches = nn_launches_;
    if (pairs) *pairs = (uint64_t)staged * 64u;
    return MOLA_ICP_OK;
}

int HipWorkspace::sync()
{
    if (!inited_) return MOLA_ICP_OK;
    HIPCHK(hipSetDevice(device_));
    HIPCHK(hipStreamSynchronize(stream_));
    if (bbox_pending_) { const int rcb = check_bboxes(); if (rcb) return rcb; }
    return MOLA_ICP_OK;
}

}  // namespace mola_icp_amd
