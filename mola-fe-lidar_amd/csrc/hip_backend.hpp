// hip_backend.hpp -- device side of the ICP core: HBM-resident clouds, the NN
// matcher kernels, the accumulation kernels.  One HipWorkspace = one in-flight
// align (stream + scratch); handles keep a pool of them so mola_icp_align() is
// re-entrant (the reference calls align() on one ICP object from several pool
// threads: src/LidarOdometry.cpp:94-96, 869).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <memory>
#include <vector>

#include "icp_loop.hpp"

namespace mola_icp_amd {

struct PoseF;
struct TiledMap;
struct NnProblem;

// RCCL, loaded at run time (rccl_dl.cpp)
struct RcclUniqueId { char internal[128]; };
int rccl_set_library(const char* path);
int rccl_unique_id(RcclUniqueId* id);
int rccl_comm_init(void** comm, int nranks, const RcclUniqueId& id, int rank);
int rccl_allreduce_sum_f64(void* comm, double* dev_buf, size_t n, hipStream_t stream);
int rccl_comm_count(void* comm, int* nranks);   // what RCCL itself reports for the communicator
int rccl_comm_destroy(void* comm);
// the node-local shared-memory all-reduce (local_comm.cpp) in the shape of the all-reduce hook; user = the LocalComm
int local_comm_hook(double* buf, int n, int device_ptr, void* user);

void reload_env_knobs();  // re-reads the MOLA_ICP_* diagnostic variables (tests); they are otherwise read once per process
void set_wait_policy(int policy);   // 0 spin, 1 yield, 2 block (HipWorkspace::spin_for)
int wait_policy();

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    // pooled: release() parks the block in a per-device free list instead of hipFree (which synchronises the whole device and
    // takes ~15 us a call), reserve() looks there first.  Only for buffers that are released when no work on them is in flight
    // on ANY stream -- the clouds (SortedCloud): every entry point that uses one returns after its work has completed.
    bool pooled = false;
    int dev = -1;               // (pooled blocks: the device they live on)
    int reserve(size_t bytes);  // grows (never shrinks); contents are NOT preserved
    void release();
    template <class T> T* as() const { return static_cast<T*>(p); }
};
void device_pool_trim(size_t keep_bytes, int device = -1);   // hipFree parked blocks of one device (-1: every device) down to keep_bytes (0: all)
size_t device_pool_bytes(int device);

// A cloud in Hilbert order with its three box levels (the tiled kernels' view of a map; its sorted coordinates and
// permutation also serve the query role).  Owned by a workspace, or shared through the handle's cloud cache.
struct SortedCloud {
    DevBuf raw;                                     // cached clouds own their raw SoA too
    const float *x = nullptr, *y = nullptr, *z = nullptr;
    size_t n = 0;
    DevBuf sorted, perm, tbox, sbox, ubox;
    DevBuf keys;                                    // a map's Hilbert keys in sorted order (n entries) ...
    DevBuf box;                                     // ... and the box they were quantised in (min xyz, max xyz): a query's place in the map
    size_t padded = 0;
    int n_tiles_p = 0, n_super = 0, n_top = 0;
    bool ready = false, cached = false;
    SortedCloud() { raw.pooled = sorted.pooled = perm.pooled = tbox.pooled = sbox.pooled = ubox.pooled = keys.pooled = box.pooled = true; }
    ~SortedCloud() { raw.release(); sorted.release(); perm.release(); tbox.release(); sbox.release(); ubox.release(); keys.release(); box.release(); }
};

class HipBatch;

// Device / pinned scratch of the batched path (HipBatch), kept in the workspace between calls: allocating and freeing a
// dozen problems' pairing buffers per call cost a third of mola_icp_align_batch (hipMalloc / hipFree synchronise).
struct BatchBuffers {
    DevBuf pos, idx, d2, gs, rows, outlier, partials;
    bool seed_valid = false, outliers_dirty = false;
    size_t outlier_cleared_for = 0;
    // the point-to-plane pipeline's per-problem state (K guesses on one cloud pair share the clouds, never these)
    DevBuf planes, plane_cache, knn_pos, knn_lb, plane_partials;
    bool knn_seed_valid = false, planes_valid = false;
    int planes_knn = 0;
    double planes_eig_thr = __builtin_nan("");   // planeEigenThreshold (sign = the all-inside-gate reading) the cached planes were decided with; NaN: none
    float knn_last_P[12] = {};
};
struct BatchScratch {
    std::vector<BatchBuffers> bufs;
    DevBuf acc_dev, stats, queue;
    double* acc_host = nullptr;                 // pinned: 32 doubles per problem (24 sums, flag in slot 30)
    size_t acc_host_problems = 0;
    DevBuf item_part_dev;                       // k_reduce_items_batch: 32 partial rows per problem
    double* item_part_host = nullptr;           // pinned: the same, 32 doubles apart, each with its sequence flag
    size_t item_part_host_problems = 0;
    DevBuf plane_acc_dev;                       // 96 doubles per problem: the 92-term plane form
    double* plane_acc_host = nullptr;           // pinned: the same, sequence flag in slot 94
    size_t plane_acc_host_problems = 0;
    unsigned long long* stats_host = nullptr;   // pinned
    std::vector<void*> events;                  // hipEvent_t
    unsigned long long seq = 0;                 // read-back sequence numbers keep growing across uses
    int fit_tiled = 0;                          // resident workgroups per CU of k_nn_tiled_batch (occupancy query, cached)
    size_t fit_tiled_lds = 0;
    void release_all();
};

class HipWorkspace final : public Stages {
    friend class HipBatch;
   public:
    explicit HipWorkspace(int device, int priority = 0);   // priority > 0: its streams are created at the device's greatest priority
    ~HipWorkspace() override;
    HipWorkspace(const HipWorkspace&) = delete;
    HipWorkspace& operator=(const HipWorkspace&) = delete;

    int init();  // creates the stream / pinned buffers; MOLA_ICP_E_NODEVICE if no GPU
    int set_external_stream(void* s);
    void set_allreduce(mola_icp_allreduce_fn fn, void* user) { ar_fn_ = fn; ar_user_ = user; }
    // native RCCL communicator (owned by the handle): accumulate() then all-reduces the device block itself
    void set_comm(void* comm) { comm_ = comm; }
    void set_global_sizes(uint64_t nl, uint64_t nm) { n_local_total_ = nl; n_map_total_ = nm; }

    int set_map_host(const float* x, const float* y, const float* z, size_t M, bool wait = true);
    int set_map_device(const float* x, const float* y, const float* z, size_t M);
    int set_local_host(const float* x, const float* y, const float* z, size_t N, bool wait = true);
    int set_local_device(const float* x, const float* y, const float* z, size_t N);
    // row e (query sharding): this rank's spatially compact shard of a scan every rank sees in full -- the scan is put in
    // Hilbert order on the device (a transient copy) and the slice [lo, hi) of that order is kept as the local cloud
    int set_local_shard(const float* x, const float* y, const float* z, size_t n_total, int rank, int nranks, bool on_device);
    int set_local_shard_range(const float* x, const float* y, const float* z, size_t n_total, size_t lo, size_t hi, bool on_device);
    int copy_shard_indices(int32_t* idx_out);  // original scan indices of the shard's points, in the shard's order
    // ... and the part of the map that shard can reach: the points inside [lo, hi] (original indices are kept: pairings
    // still name ORIGINAL map points).  match() refuses a pose that moves the shard's reach out of the box.
    int set_map_slab(const float* x, const float* y, const float* z, size_t M, const double lo[3], const double hi[3], bool on_device,
                     size_t* n_kept);
    int shard_reach_box(const Mat4& T, double margin, double lo[3], double hi[3]);  // AABB of T (+) (local bbox), grown by margin

    // Stages
    int match(const Mat4& T, double threshold, const mola_icp_params& p, uint64_t* n_pairs) override;
    int accumulate(const mola_icp_params& p, const Mat4& Tcur, int stage, const double cl[3], const double cg[3],
                   bool reset_outliers, double acc[kNAcc]) override;
    int allreduce(double acc[kNAcc]) override;
    int quality_pairs(const Mat4& T, double threshold, const mola_icp_params& p, double acc[kNAcc], bool* done) override;
    uint64_t n_local_total() const override { return n_local_total_ ? n_local_total_ : N_; }
    uint64_t n_map_total() const override { return n_map_total_ ? n_map_total_ : M_; }

    int copy_pairing(int32_t* idx_out, float* d2_out);  // after match(); syncs

    // row f3: point-to-plane matcher + the quadratic form of its cost (Stages overrides)
    int match_planes(const Mat4& T, const mola_icp_params& p) override;
    int accumulate_planes(double acc[kNAccPlaneHost]) override;
    int copy_planes(uint8_t* valid, double* centroid, double* normal, int32_t* knn_idx);

    // row f4: device-resident cloud cache
    int build_cached(SortedCloud& sc, const float* x, const float* y, const float* z, size_t n);  // host pointers
    int build_cached(const std::shared_ptr<SortedCloud>& sc, const float* x, const float* y, const float* z, size_t n, bool wait);
    int finish_build(SortedCloud& sc);
    hipError_t quick_sync();
    void use_cached_map(const std::shared_ptr<SortedCloud>& sc);
    void use_cached_local(const std::shared_ptr<SortedCloud>& sc);
    int voxel_downsample(const float* x, const float* y, const float* z, size_t n, double voxel_size, float* out_x,
                         float* out_y, float* out_z, size_t capacity, size_t* n_out);
    int sync();

    // NN-kernel timing (HIP events on this workspace's stream)
    // per-launch HIP events + executed-pair counters (ms_nn_kernel, nn_pairs_evaluated of the results).  Off by default:
    // the two event packets cost ~8 us per iteration at odometry sizes, the counter read-back a stream synchronisation.
    void set_profiling(bool on) { profiling_ = on; }
    // drop what earlier matches left for the clouds in place (neighbour lists, pairing seeds, plane cache); the prepared (sorted)
    // clouds and their work-queue order stay -- `everything`: the order too (what the very first align on these clouds paid)
    void forget_warm_start(bool everything = false);
    void reset_stats();
    int collect_stats(double* ms_total, uint32_t* launches, uint32_t* kernel_used, uint64_t* pairs = nullptr);

    size_t N() const { return N_; }
    size_t M() const { return M_; }
    int device() const { return device_; }

   private:
    int prepare_map();    // derived map image for the MFMA matcher
    int prepare_tiles();    // Morton-sorted map + tile boxes for the tiled matcher
    int prepare_queries(bool aside = false);  // Hilbert-sorted local cloud (aside: on the second stream, beside the map's chain)
    int prepare_both();
    int bbox_of(const float* x, const float* y, const float* z, size_t n, float out[6]);   // (waits)
    static constexpr int kBboxRows = 256;   // workgroups of k_bbox_partial = partial rows in map_meta_
    int bbox_async(const float* x, const float* y, const float* z, size_t n, int slot, const std::shared_ptr<SortedCloud>& owner);   // device block + pinned slot, no wait
    int bbox_rows_async(const float* x, const float* y, const float* z, size_t n, int slot, const std::shared_ptr<SortedCloud>& owner,
                        hipStream_t st = nullptr, DevBuf* meta = nullptr);   // the partial rows only (the sort finishes the box)
    int check_bboxes();                                                                     // after the next wait on stream_
    float* bbox_dev() { return map_meta_.as<float>() + 6 * kBboxRows; }
    unsigned int bbox_pending_ = 0;
    int bbox_n_rows_ = 0;   // rows k_bbox_rows wrote last
    std::weak_ptr<SortedCloud> bbox_owner_[2];   // the prepared-cloud object each pending box belongs to (marked "not prepared" if the box is not finite)
    int launch_tiled(const struct PoseF& P, float thr2, bool use_seed, unsigned int* counter);
    int fill_nn_problem(const struct PoseF& P, float thr2, bool use_seed, NnProblem& pb);
    int launch_coop(const struct PoseF& P, float thr2, bool use_seed);
    int launch_q4(const struct PoseF& P, float thr2, bool use_seed);   // k_nn_q4: four lanes per query (kernels_q4.hpp)
    TiledMap tiled_map() const;
    int spin_for(volatile unsigned long long* flag, unsigned long long seq);
    int check_slab(const Mat4& T, double threshold);
    int launch_nn(const Mat4& T, float thr2, int kernel);

    int device_;
    int priority_ = 0;   // stream priority class (0 normal, 1 high: the odometry path beside batches of checks)
    bool inited_ = false;
    hipStream_t stream_ = nullptr;
    bool own_stream_ = false;

    size_t N_ = 0, M_ = 0;
    uint64_t n_local_total_ = 0, n_map_total_ = 0;
    // clouds: owned copies or borrowed device pointers
    DevBuf map_own_, loc_own_;
    BatchScratch batch_scratch_;
    DevBuf shard_idx_, slab_orig_, stage_in_;   // row e: the shard's original scan indices; slab point -> original map index
    size_t shard_n_ = 0;
    bool slab_active_ = false, slab_violation_ = false;
    double slab_lo_[3] = {0, 0, 0}, slab_hi_[3] = {0, 0, 0};
    float loc_bbox_[6] = {0, 0, 0, 0, 0, 0};
    bool loc_bbox_valid_ = false;
    const float *gx_ = nullptr, *gy_ = nullptr, *gz_ = nullptr;
    const float *lx_ = nullptr, *ly_ = nullptr, *lz_ = nullptr;
    // derived map image for the MFMA kernel ([tile][4][16] fp32) + its bounds
    DevBuf map_img_, map_meta_;
    bool map_img_valid_ = false;
    float map_center_[3] = {0, 0, 0};
    float map_radius_ = 0;
    int map_tiles_ = 0, map_segs_ = 0, map_seg_tiles_ = 0;
    int num_cus_ = 256;
    // tiled matcher: the map and the local cloud in Hilbert order (own, or borrowed from the handle's cache)
    std::shared_ptr<SortedCloud> map_sc_, loc_sc_;
    DevBuf sort_scratch_;
    DevBuf sort_scratch_loc_, loc_meta_;   // the queries' prepare chain when it runs beside the map's (prepare_both)
    DevBuf ts_pos_, ts_idx_, ts_d2_;  // the tiled matcher's pairing, in SORTED query order
    DevBuf ts_gs_;                    // ... and each neighbour's coordinates (3 x padded floats): next launch's seeds, accumulate's g
    DevBuf rows_;                     // the matchers' fused stage-0 sums, one row of kNAcc doubles per 64 queries (item_row_mfma)
    bool rows_valid_ = false;         // rows_ belongs to the pairing in place
    int rows_count_ = 0;              // ... and holds this many rows
    DevBuf item_part_;                // k_reduce_items' partial rows
    double* item_part_host_ = nullptr;  // pinned: the same, 32 doubles apart, each with its sequence flag
    unsigned long long* quality_host_ = nullptr;  // pinned: k_quality_from_lists' per-workgroup records {pairs | open << 32, sequence}
    bool pairing_sorted_ = false;     // which representation the stored pairing / warm start is in
    DevBuf knn_lb_;   // per query: lower bound on the distance to every map point outside its stored neighbour list (KnnCert)
    double knn_last_step_ = 1e30;   // size of the pose step between the last two launches of the plane matcher (flavour heuristic)
    float knn_last_P_[12] = {};  // the pose of the launch that wrote knn_pos_ / knn_lb_ (PoseF: R row-major, then t)
    DevBuf planes_, knn_pos_, plane_acc_, plane_cache_;  // point-to-plane pairing (sorted query order; knn_pos_: knn + 1 positions per query) + its accumulators
    double* plane_acc_host_ = nullptr;
    bool planes_valid_ = false, planes_empty_ = false;
    int planes_knn_ = 0;
    double planes_eig_thr_ = __builtin_nan("");  // planeEigenThreshold (sign = the all-inside-gate reading) the planes in plane_cache_ were decided with; NaN: none
    double knn_changed_items_ = -1.0;  // items whose neighbour lists changed in the last iteration (-1: unknown)
    bool knn_seed_valid_ = false;  // knn_pos_ holds the last launch's neighbours for the clouds in place
    DevBuf stats_;                    // the cooperative matcher's slotted statistics counters
    unsigned long long* stats_host_ = nullptr;  // pinned
    DevBuf redo_list_;                // work items with exact distance ties: redone with the full lexicographic key
    DevBuf item_cost_, item_order_;   // per work item: cycles in the last launch -> heavy-first order of the next
    int knn_fit_[2][3][6] = {};       // resident workgroups per CU of k_knn_planes<K, flavour, QL> (insertion, counting, dense insertion; occupancy queries, cached)
    size_t knn_fit_lds_[2][6] = {};
    DevBuf knn_cost_, knn_order_;     // ... of the kNN (point-to-plane) kernels' full sweeps
    bool knn_cost_valid_ = false, knn_order_valid_ = false;
    int knn_plan_interval_ = 1, knn_launches_since_order_ = 0;
    // the work lists are re-sorted on a side stream, behind the matcher launch whose costs they read and beside the
    // accumulation that follows it (a single-block 22-30 us kernel otherwise on the next matcher's critical path)
    hipStream_t aux_stream_ = nullptr;
    hipEvent_t ev_order_a_ = nullptr, ev_order_b_ = nullptr, ev_prep_ = nullptr;
    bool order_pending_ = false;
    int order_begin();   // aux_stream_ waits for what stream_ holds so far
    int order_end();     // ... and the next matcher launch will wait for what aux_stream_ holds
    int order_join();    // (called by that launch)
    bool cost_valid_ = false, order_valid_ = false;
    unsigned int launches_since_order_ = 0, plan_interval_ = 1;
    int fit_cache_[8] = {};            // resident blocks per CU of the tiled kernels (0 = not queried yet)
    size_t fit_cache_lds_[8] = {};     // ... for this dynamic-LDS size
    unsigned long long readback_seq_ = 0;  // accumulate(): sequence number the reduction publishes next to its sums
    // pairing + scratch
    DevBuf idx_, d2_, seg_idx_, seg_d2_, outlier_, partials_, acc_dev_;
    double* acc_host_ = nullptr;  // pinned
    float* meta_host_ = nullptr;  // pinned
    bool pairing_valid_ = false;
    bool counters_clean_ = false;      // the matcher's device counters are known to be zero
    bool outliers_dirty_ = false;      // stage 1 may have flagged outliers since the last clear
    size_t outlier_cleared_for_ = 0;   // the flag buffer is known to be all-zero for this many queries
    bool seed_valid_ = false;  // idx_ holds a pairing of the CURRENT clouds: usable as the next match's warm start

    mola_icp_allreduce_fn ar_fn_ = nullptr;
    void* ar_user_ = nullptr;
    void* comm_ = nullptr;

    std::vector<hipEvent_t> ev_;  // pairs: start, stop
    size_t ev_used_ = 0;
    uint32_t last_kernel_ = 0;
    uint32_t nn_launches_ = 0;
    bool profiling_ = false;
    uint64_t dense_pairs_ = 0;
    unsigned long long* wave_times_ = nullptr; // MOLA_ICP_DEBUG_STATS=2 only
    bool wave_times_coop_ = false;             // ... written by k_nn_coop (its own record layout)
    unsigned long long* dbg_stats_ = nullptr;  // MOLA_ICP_DEBUG_STATS=1 only
};

// K point-to-point problems on ONE device advanced by batched launches (BatchStages of icp_loop.hpp): k_nn_coop with
// blockIdx.y = problem, k_accumulate_batch, k_reduce_partials_batch, one pinned read-back per pass.  Problems are
// pairs of prepared clouds (shared: the K guesses of the loop-closure Monte-Carlo use the same two).  Runs on the
// stream and scratch of a workspace the caller holds for the batch's lifetime.
struct BatchProblem {
    std::shared_ptr<SortedCloud> map, loc;
};
class HipBatch final : public BatchStages {
   public:
    HipBatch(HipWorkspace& ws, std::vector<BatchProblem> probs);
    ~HipBatch() override;
    int init();  // per-problem pairing buffers, pinned read-back block
    int init_planes(int knn);  // ... and the point-to-plane pipeline's (on its first use)
    int size() const override { return (int)probs_.size(); }
    int match(const uint8_t* active, const Mat4* T, double threshold, const mola_icp_params& p) override;
    int accumulate(const uint8_t* active, const mola_icp_params& p, const Mat4* Tcur, int stage, const double (*cl)[3],
                   const double (*cg)[3], bool reset_outliers, double (*acc)[kNAcc]) override;
    // row f3, batched: the plane matcher (k_knn_coop, blockIdx.y = problem) and the plane form of every active problem
    int match_planes(const uint8_t* active, const Mat4* T, const mola_icp_params& p) override;
    int accumulate_planes(const uint8_t* active, double (*acc)[kNAccPlaneHost]) override;
    uint64_t n_local_total(int k) const override { return probs_[(size_t)k].loc->n; }
    uint64_t n_map_total(int k) const override { return probs_[(size_t)k].map->n; }
    // statistics of the matcher launches since init(): total HIP-event time, launches, evaluated pairs (whole batch)
    int collect_stats(double* ms_total, uint32_t* launches, uint64_t* pairs);

   private:
    using Buffers = BatchBuffers;
    HipWorkspace& ws_;
    BatchScratch& sc_;           // the workspace's: buffers survive this object
    std::vector<BatchProblem> probs_;
    size_t ev_used_ = 0;
    uint32_t nn_launches_ = 0;
    bool inited_ = false, planes_inited_ = false;
};

}  // namespace mola_icp_amd
