/*
 * mola_icp_amd.h -- C-ABI of the MI355X-native ICP registration core.
 *
 * This is the drop-in boundary for the one hot path of MOLAorg/mola-fe-lidar:
 * `LidarOdometry::run_one_icp()` (src/LidarOdometry.cpp:851-895) delegating to
 * `mp2p_icp::ICP::align(from, to, init_guess_to_wrt_from, params, result)`
 * (src/LidarOdometry.cpp:869-871).  Plain pointers and sizes only: no C++,
 * torch, MRPT or HIP types cross it.  Reference-side binding: INTEGRATION.md.
 *
 * Conventions (all from the reference):
 *  - `from` = global cloud / map (M points), `to` = local cloud / queries
 *    (N points): include/mola-fe-lidar/LidarOdometry.h:121, src/LidarOdometry.cpp:278-279.
 *  - poses are `to` w.r.t. `from` (LidarOdometry.h:122,131): g ~= T (+) l.
 *    4x4 row-major doubles; helpers convert from/to MRPT TPose3D
 *    (x,y,z,yaw,pitch,roll), R = Rz(yaw)Ry(pitch)Rx(roll) (src/LidarOdometry.cpp:272-275).
 *  - clouds are fp32 structure-of-arrays x[], y[], z[].
 *  - every function returns MOLA_ICP_OK (0) or a negative MOLA_ICP_E_* code; the
 *    message of the calling thread's last error is mola_icp_last_error().
 *    Nothing aborts or throws across this boundary (the reference's own error
 *    convention is C++ exceptions caught per task: src/LidarOdometry.cpp:510-513,
 *    845-848; the C++ shim turns codes back into exceptions).
 *  - mola_icp_align() is re-entrant per handle (the reference calls align() on
 *    one ICP object from several pool threads: src/LidarOdometry.cpp:94-96,869)
 *    and takes the parameters per call (src/LidarOdometry.cpp:287-290).
 */
#ifndef MOLA_ICP_AMD_H
#define MOLA_ICP_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MOLA_ICP_ABI_VERSION 5

/* ---- status codes ------------------------------------------------------ */
enum {
    MOLA_ICP_OK            = 0,
    MOLA_ICP_E_BADARG      = -1,  /* null pointer, bad size, bad enum          */
    MOLA_ICP_E_CONFIG      = -2,  /* YAML / class-name error (cf. cpp:70-75)   */
    MOLA_ICP_E_HIP         = -3,  /* a HIP runtime call failed                 */
    MOLA_ICP_E_OOM         = -4,  /* device or host allocation failed          */
    MOLA_ICP_E_NODEVICE    = -5,  /* no gfx950 device / HIP kernels unusable   */
    MOLA_ICP_E_UNSUPPORTED = -6,  /* valid config this build does not run      */
    MOLA_ICP_E_COMM        = -7,  /* the all-reduce hook reported a failure    */
    MOLA_ICP_E_INTERNAL    = -8
};

/* ---- termination reasons: mp2p_icp::IterTermReason, read at cpp:888 ---- */
enum {
    MOLA_ICP_TERM_UNDEFINED      = 0,
    MOLA_ICP_TERM_NO_PAIRINGS    = 1,
    MOLA_ICP_TERM_SOLVER_ERROR   = 2,
    MOLA_ICP_TERM_MAX_ITERATIONS = 3,
    MOLA_ICP_TERM_STALLED        = 4
};

/* ---- matcher / solver / quality classes (params/icp-settings-regular.yaml:23-46) */
enum { MOLA_ICP_MATCHER_POINTS_DISTANCE_THRESHOLD = 0, MOLA_ICP_MATCHER_POINT2PLANE = 1 };
enum { MOLA_ICP_SOLVER_HORN = 0, MOLA_ICP_SOLVER_GAUSS_NEWTON = 1 };
enum { MOLA_ICP_QUALITY_PAIRED_RATIO = 0 };
/* nearest-neighbour kernel selection (new key `nn_kernel`, default auto) */
enum {
    MOLA_ICP_NN_AUTO  = 0,
    MOLA_ICP_NN_VALU  = 1,  /* exact VALU brute force (reference kernel)                            */
    MOLA_ICP_NN_MFMA  = 2,  /* dense N x M MFMA filter + exact re-evaluation of survivors           */
    MOLA_ICP_NN_TILED = 3   /* exact VALU brute force over the Hilbert-sorted map tiles a query group can reach */
};

/* Further entries of the `matchers:` / `solvers:` sequences -- "a sequence of one or more" pairs of class + params
 * (params/icp-settings-regular.yaml:28-31), which is what runFromIteration / runUpToIteration (icpreg:38-39) exist for.
 * Entry 0 of each sequence lives in the flat fields of mola_icp_params; entries 1.. here.  Semantics ([EXT-recalled]
 * mp2p_icp: a matcher outside its iteration range contributes no pairings; the solvers are tried in order):
 *   - at iteration `it` the ACTIVE matcher is the one whose [runFromIteration, runUpToIteration (0 = no limit)] holds it;
 *     none active = no pairings (terminates as NoPairings), as with a single entry;
 *   - ONE Points_DistanceThreshold and ONE Point2Plane matcher may be active in the same iteration when its solver is
 *     Solver_GaussNewton and no robust kernel / outlier weights are set: their pairings go into one solve (the point-to-point
 *     sums enter the Gauss-Newton form as three plane terms per pairing: mola_icp_mixed_form); any other overlap is
 *     MOLA_ICP_E_UNSUPPORTED, named;
 *   - the solver of an iteration is the first `solvers:` entry whose range holds it (entries without a range: always); an
 *     iteration no solver covers ends the align with MOLA_ICP_TERM_SOLVER_ERROR.
 * Typical use: point-to-point + Horn for the first iterations, then point-to-plane + Gauss-Newton. */
#define MOLA_ICP_MAX_EXTRA_STAGES 3
typedef struct mola_icp_matcher_entry {
    int32_t  matcher_class;              /* MOLA_ICP_MATCHER_*                       */
    double   matcher_threshold;          /* threshold / distanceThreshold [m]        */
    double   plane_eigen_threshold;      /* planeEigenThreshold                      */
    uint32_t knn;                        /* knn                                      */
    uint32_t run_from_iteration;         /* runFromIteration                         */
    uint32_t run_up_to_iteration;        /* runUpToIteration (0 = no limit)          */
} mola_icp_matcher_entry;
/* entries 1.. of `quality:` (icpreg:40-46 is a sequence too): Results::quality = sum w_i q_i / sum w_i over all entries ([EXT-recalled]
 * mp2p_icp weighs its quality evaluators; `weight` defaults to 1).  One more matcher pass per entry. */
typedef struct mola_icp_quality_entry {
    int32_t  quality_class;              /* MOLA_ICP_QUALITY_*                       */
    double   quality_threshold;          /* thresholdDistance [m]                    */
    double   weight;                     /* weight (default 1)                       */
} mola_icp_quality_entry;
typedef struct mola_icp_solver_entry {
    int32_t  solver_class;               /* MOLA_ICP_SOLVER_*                        */
    uint32_t solver_max_iterations;      /* params.maxIterations (Gauss-Newton)      */
    uint32_t run_from_iteration;         /* runFromIteration                         */
    uint32_t run_up_to_iteration;        /* runUpToIteration (0 = no limit)          */
} mola_icp_solver_entry;

/* ---- per-call parameters == mp2p_icp::Parameters + the per-object pipeline
 *      settings the YAML carries (params/icp-settings-regular.yaml:10-46).
 *      Replaces `mp2p_icp::Parameters` (include/mola-fe-lidar/LidarOdometry.h:99,123). */
typedef struct mola_icp_params {
    uint32_t max_iterations;             /* params.maxIterations            icpreg:11 */
    double   min_abs_step_trans;         /* params.minAbsStep_trans [m]     icpreg:12 */
    double   min_abs_step_rot;           /* params.minAbsStep_rot [rad]     icpreg:13 */
    int32_t  use_scale_outlier_detector; /* pairingsWeightParameters        icpreg:16 */
    double   scale_outlier_threshold;    /*                                 icpreg:17 */
    int32_t  use_robust_kernel;          /*                                 icpreg:19 */
    double   robust_kernel_param;        /* RADIANS (YAML is degrees)       icpreg:20 */
    double   robust_kernel_scale;        /*                                 icpreg:21 */
    int32_t  solver_class;               /* MOLA_ICP_SOLVER_*               icpreg:24 */
    uint32_t solver_max_iterations;      /* solvers[].params.maxIterations  icpreg:26 */
    int32_t  matcher_class;              /* MOLA_ICP_MATCHER_*              icpreg:33 */
    double   matcher_threshold;          /* threshold / distanceThreshold [m] icpreg:35 */
    double   plane_eigen_threshold;      /* planeEigenThreshold             icpreg:36 */
    uint32_t knn;                        /* knn                             icpreg:37 */
    uint32_t run_from_iteration;         /* runFromIteration                icpreg:38 */
    uint32_t run_up_to_iteration;        /* runUpToIteration (0 = no limit) icpreg:39 */
    int32_t  quality_class;              /* MOLA_ICP_QUALITY_*              icpreg:44 */
    double   quality_threshold;          /* thresholdDistance [m]           icpreg:46 */
    /* keys that do not exist in the reference (default 0 = reference behaviour) */
    int32_t  fixed_iterations;           /* !=0: never stop on the stall test (benchmarks) */
    int32_t  nn_kernel;                  /* MOLA_ICP_NN_*                                */
    int32_t  skip_quality;               /* !=0: skip the quality pass, quality = -1 (benchmarks) */
    /* matchers[1..] / solvers[1..] of the YAML sequences (see mola_icp_matcher_entry); 0 = the single-entry pipelines above */
    uint32_t solver_run_from_iteration;  /* solvers[0].params.runFromIteration (absent: 0)          */
    uint32_t solver_run_up_to_iteration; /* solvers[0].params.runUpToIteration (absent / 0: no limit) */
    uint32_t n_extra_matchers;
    uint32_t n_extra_solvers;
    mola_icp_matcher_entry extra_matchers[MOLA_ICP_MAX_EXTRA_STAGES];
    mola_icp_solver_entry  extra_solvers[MOLA_ICP_MAX_EXTRA_STAGES];
    double   quality_weight;             /* quality[0].weight (0 is read as 1: a zeroed / pre-ABI-3 struct means one evaluator) */
    uint32_t n_extra_quality;
    mola_icp_quality_entry extra_quality[MOLA_ICP_MAX_EXTRA_STAGES];
    /* ABI 4 -- alternative READINGS of three behaviours of mp2p_icp that nothing in the reference pins and that are NOT pose- or
     * goodness-neutral (INTEGRATION.md "When a real mp2p_icp disagrees").  0 = the reading this library
     * implements by default.  YAML: an optional top-level `readings:` map beside `params:` -- a key outside the reference's schema. */
    int32_t  reading_outlier_single_pass;        /* scale-outlier detector: ONE weighted pass instead of two     icpreg:14-17 */
    int32_t  reading_p2pl_all_inside_gate;       /* Matcher_Point2Plane: a plane needs ALL knn inside the gate   icpreg:33-39 */
    int32_t  reading_quality_denominator_local;  /* PairedRatio = pairings / N(local), not / min(N, M)           icpreg:44-46 */
    /* ABI 5 -- pairingsWeightParameters.use_robust_kernel (icpreg:18-21) with Matcher_Point2Plane ALONE.  0: refused by name (the
     * kernel's weights are defined on POINT pairings relative to a Horn solve's centroids; what mp2p_icp does with the flag when a
     * solve holds plane pairings only is not pinned by anything in the reference).  !=0: accepted, and plane pairings keep unit
     * weights -- the reading under which the flag, like use_scale_outlier_detector in the reference's own params block, acts on point
     * pairings and finds none.  (Point AND plane pairings in one solve with the flag set stay refused.) */
    int32_t  reading_robust_kernel_skips_planes;
} mola_icp_params;

/* ---- result == the fields of mp2p_icp::Results that the reference consumes
 *      (src/LidarOdometry.cpp:873-888) + per-call statistics. */
typedef struct mola_icp_result {
    double   T[16];          /* optimal_tf.mean (cpp:876,879), row-major 4x4               */
    double   cov[36];        /* optimal_tf.cov, 6x6 row-major, order (x,y,z,wx,wy,wz)      */
    double   quality;        /* Results::quality (cpp:873,880) in [0,1]                    */
    uint32_t n_iterations;   /* Results::nIterations (cpp:886)                             */
    uint32_t termination;    /* Results::terminationReason (cpp:888), MOLA_ICP_TERM_*      */
    uint64_t n_pairs;        /* pairings that fed the last solve                           */
    double   rmse;           /* sqrt(mean d^2) over those pairings [m]                     */
    double   ms_upload;      /* host->HBM copies + map preparation                         */
    double   ms_iterations;  /* the iteration loop                                         */
    double   ms_quality;     /* the quality pass                                           */
    double   ms_nn_kernel;   /* sum of HIP-event durations of the NN kernel launches (0 unless mola_icp_set_profiling) */
    uint32_t n_nn_launches;  /* number of NN kernel launches (matcher + quality passes)     */
    uint32_t nn_kernel_used; /* MOLA_ICP_NN_VALU / _MFMA / _TILED                              */
    uint64_t nn_pairs_evaluated; /* exact (query, map point) distance evaluations over all launches;
                                    N*M per launch for the dense kernels, fewer under exact tile culling */
} mola_icp_result;

#define MOLA_ICP_NACC 24
/* Per-iteration accumulator block (fp64) -- the ONLY data reduced across GPUs:
 *  [0] W = sum w          [1..3] sum w*l        [4..6] sum w*g
 *  [7..15] sum w*l*g^T (row-major, l_r*g_c)     [16] pairs with w>0
 *  [17] sum d^2 over those pairs                [18..23] sum w*l*l^T (xx,xy,xz,yy,yz,zz) */

/* All-reduce hook for the query-sharded multi-GPU path: must sum `n` doubles
 * in place across ranks and return 0.  `device_ptr`!=0 means buf is in HBM
 * (ordered on the handle's stream), else it is host memory. */
typedef int (*mola_icp_allreduce_fn)(double* buf, int n, int device_ptr, void* user);

typedef struct mola_icp_handle mola_icp_handle; /* one per `mp2p_icp::ICP` object (LidarOdometry.h:98) */

/* ---- library ---------------------------------------------------------- */
int         mola_icp_abi_version(void);
const char* mola_icp_last_error(void);            /* thread-local, never NULL */
const char* mola_icp_status_string(int status);
int         mola_icp_device_count(int* count);    /* gfx950 devices visible   */
/* The MOLA_ICP_* diagnostic / tuning environment variables are read once, when the library is loaded --
 * never on a launch path.  Tests that toggle one on a live process call this to have them read again. */
int         mola_icp_debug_reload_env(void);
/* How a host thread waits for the sums of an accumulation pass (one hand-over per ICP iteration).  The reference runs align() from
 * the odometry thread AND max(2, hw/2) pool threads on one ICP object (src/LidarOdometry.cpp:94-96, 183-184, 711-712, 869): under the
 * default, MOLA_ICP_WAIT_SPIN, every one of them holds a core at 100 % while its align is in flight.  _YIELD: ~10 us of spinning,
 * then sched_yield() between polls.  _BLOCK: the thread sleeps between polls (no core burnt, ~55-60 us per hand-over).  Process-wide;
 * also the environment variable MOLA_ICP_WAIT=spin|yield|block (read at load).  Costs per iteration: INTEGRATION.md. */
/* Priority class of the CALLING THREAD's subsequent calls on any handle: high > 0 = the workspaces (streams) those calls lease run at
 * the device's greatest stream priority -- their launches are picked ahead of the queued launches of normal-priority calls (the
 * odometry step beside batches of nearby / loop-closure checks on the same ICP object: src/LidarOdometry.cpp:94-96, 183-184, 711-712,
 * 869).  Thread-local, default 0.  mola_lo_process_scan raises it for its own duration. */
int         mola_icp_set_thread_priority(int high);
int         mola_icp_get_thread_priority(int* high);
#define MOLA_ICP_WAIT_SPIN  0
#define MOLA_ICP_WAIT_YIELD 1
#define MOLA_ICP_WAIT_BLOCK 2
int         mola_icp_set_wait_policy(int policy);
int         mola_icp_get_wait_policy(int* policy);

/* ---- parameters ------------------------------------------------------- */
/* the defaults of mp2p_icp::Parameters + Points_DistanceThreshold/Horn/PairedRatio */
int mola_icp_params_default(mola_icp_params* p);

/* Replaces load_icp_set_of_params() (src/LidarOdometry.cpp:57-88): parses one
 * `icp-settings-*.yaml` document (text), accepts every key of
 * params/icp-settings-regular.yaml:7-46, checks `icp_class`, and fills *p.
 * Unknown class names -> MOLA_ICP_E_CONFIG naming the class (cf. cpp:70-75). */
int mola_icp_params_from_yaml(const char* yaml_text, mola_icp_params* p);
/* same from a file; resolves `$include{...}` and `$(mola-dir PKG)` (-> mola_dir)
 * like params/kitti-default.yaml:43,46,50; `key` selects a sub-map
 * (e.g. "icp_settings_with_vel"), NULL/"" = the document root. */
int mola_icp_params_from_yaml_file(const char* path, const char* mola_dir, const char* key, mola_icp_params* p);

/* The reference keeps matchers / solvers / quality evaluators inside the ICP OBJECT (LidarOdometry.h:98,
 * initialised once at src/LidarOdometry.cpp:80-87) and hands `mp2p_icp::Parameters` (cpp:77-78: maxIterations,
 * minAbsStep_trans, minAbsStep_rot, pairingsWeightParameters{...}) to align() PER CALL -- and at cpp:287-290 it
 * passes the NearbyAlign case's Parameters to the AlignKind::LidarOdometry object (cpp:869).  The flat
 * mola_icp_params carries both halves; this composes them the way that call does:
 *   *out = *object_settings  (solver_*, matcher_*, plane_eigen_threshold, knn, run_*_iteration, quality_*,
 *                             nn_kernel, skip_quality)
 *   with the mp2p_icp::Parameters fields of *call_parameters (max_iterations, min_abs_step_trans,
 *   min_abs_step_rot, use_scale_outlier_detector, scale_outlier_threshold, use_robust_kernel, robust_kernel_param,
 *   robust_kernel_scale; and fixed_iterations, which this implementation reads from the same `params:` block). */
int mola_icp_params_compose(const mola_icp_params* object_settings, const mola_icp_params* call_parameters,
                            mola_icp_params* out);

/* ---- handle ----------------------------------------------------------- */
/* device < 0: current HIP device.  Replaces mrpt::rtti::classFactory(icp_class)
 * (src/LidarOdometry.cpp:66-68).  Fails with MOLA_ICP_E_NODEVICE when no GPU
 * is present: there is no CPU fallback. */
int mola_icp_create(int device, mola_icp_handle** out);
int mola_icp_destroy(mola_icp_handle* h);
/* run this handle's work on an existing HIP stream (hipStream_t as void*); NULL = own stream */
int mola_icp_set_stream(mola_icp_handle* h, void* hip_stream);
/* query-sharded multi-GPU: install the accumulator all-reduce (NULL = single GPU) */
int mola_icp_set_allreduce(mola_icp_handle* h, mola_icp_allreduce_fn fn, void* user);

/* Native RCCL for the query-sharded path (one process per GPU).  Rank 0 calls
 * mola_icp_comm_unique_id() and ships the 128 bytes to the other ranks by any means (MPI,
 * torch.distributed, a file); then EVERY rank calls mola_icp_comm_init() (collective).  From then on the
 * resident-cloud API all-reduces the accumulator block on the device, on the handle's stream, and the
 * mola_icp_set_allreduce hook is ignored.  RCCL is loaded at run time: `path` (optional) names the
 * library this process already uses (e.g. the one bundled with PyTorch). */
int mola_icp_comm_set_library(const char* path);
int mola_icp_comm_unique_id(uint8_t id_out[128]);
int mola_icp_comm_init(mola_icp_handle* h, const uint8_t id[128], int nranks, int rank);
/* the size RCCL itself reports for this handle's communicator (ncclCommCount): what a launcher prints to show the
 * collective really spans the ranks it started */
int mola_icp_comm_nranks(mola_icp_handle* h, int* nranks_out);
int mola_icp_comm_destroy(mola_icp_handle* h);

/* The node-local communicator: the all-reduce of ONE node's ranks through a POSIX shared-memory mailbox (csrc/local_comm.cpp).
 * The reduced block is consumed by each rank's host thread (the fp64 solve), and a single-GPU iteration already ends with the
 * device writing the block to pinned host memory: the ranks exchange the 24 (92) doubles there, between cores -- no launch, no
 * device-side wait, nothing measurable per step where a 192-byte RCCL all-reduce costs 13-19 us; every rank adds the rows in rank order, so all
 * ranks hold the same bits.  What `bench.py --gpus N` uses on one node; RCCL above stays for ranks on different nodes.
 *  - create: COLLECTIVE over the ranks of the node; `name` = a POSIX shm name unique to this job (rank 0 picks it and ships it
 *    like the RCCL id); returns once all `nranks` ranks have joined (then the name is unlinked: nothing is left in /dev/shm),
 *    MOLA_ICP_E_COMM after `timeout_s` (<= 0: 30 s) -- the same time-out bounds every wait of an all-reduce;
 *  - allreduce: sums buf[0, n) (n <= 120, host memory) over the ranks in place; the ranks must make the same calls in the same
 *    order (a rank found ahead, or with another n, is MOLA_ICP_E_COMM on every rank, not a hang).  After MOLA_ICP_E_COMM the
 *    communicator is finished (this end is out of step with the others): every later call fails at once -- destroy it and create a
 *    new one on every rank.  A name that still leads to the leftover of a run that crashed inside `create` is harmless: a rank that
 *    mapped the leftover notices rank 0 replacing it and joins the new segment;
 *  - abort: this rank cannot go on (an error outside the collective): the others' next all-reduce fails at once;
 *  - attach_local: the handle's resident-cloud aligns reduce through `c` (NULL detaches); `c` stays the caller's to destroy,
 *    after mola_icp_comm_destroy(h) or mola_icp_destroy(h).  mola_icp_comm_nranks() then reports the ranks that joined `c`. */
typedef struct mola_icp_local_comm mola_icp_local_comm;
int mola_icp_local_comm_create(const char* name, int nranks, int rank, double timeout_s, mola_icp_local_comm** out);
int mola_icp_local_comm_allreduce(mola_icp_local_comm* c, double* buf, int n);
int mola_icp_local_comm_nranks(mola_icp_local_comm* c, int* nranks_out);
int mola_icp_local_comm_abort(mola_icp_local_comm* c);
int mola_icp_local_comm_destroy(mola_icp_local_comm* c);
int mola_icp_comm_attach_local(mola_icp_handle* h, mola_icp_local_comm* c);

/* ---- the hot path ------------------------------------------------------ */
/* Replaces mp2p_icp::ICP::align() as called at src/LidarOdometry.cpp:869-871.
 * Host pointers; copies both clouds to HBM, runs every iteration on the GPU,
 * never retains caller pointers.  Thread-safe per handle.
 * A cloud with a coordinate that is not a finite number (NaN, +-inf) is refused with MOLA_ICP_E_BADARG
 * ("non-finite coordinates") by the call that brought it in -- every align entry point, mola_icp_cloud_put --
 * decided on the device from the cloud's bounding box; the handle stays usable. */
int mola_icp_align(mola_icp_handle* h,
                   const float* from_x, const float* from_y, const float* from_z, size_t M,
                   const float* to_x, const float* to_y, const float* to_z, size_t N,
                   const double init_T[16], const mola_icp_params* p, mola_icp_result* out);

/* Loop-closure / nearby-KF batch (src/LidarOdometry.cpp:704-741, 767-788): n_pairs independent problems on this
 * handle's device.  Pairs advance in LOCKSTEP, a dozen at a time -- every stage of an iteration is one launch over the
 * pairs still iterating -- for both single-entry pipelines: point-to-point + Horn and the reference's own nearby /
 * loop-closure settings, Matcher_Point2Plane + Solver_GaussNewton (params/icp-settings-loop-closure.yaml:23-39); the
 * next chunk's uploads and preparation overlap the current chunk's loop.  Pairs the lockstep path does not serve (tiny
 * clouds under the dense NN kernels, staged multi-entry pipelines) run stream-per-pair on the handle's worker threads.
 * Every result equals the pair's stand-alone mola_icp_align, bit for bit.
 * Arrays of per-pair pointers/sizes; init_T = n_pairs x 16; out = n_pairs results. */
int mola_icp_align_batch(mola_icp_handle* h, size_t n_pairs,
                         const float* const* from_x, const float* const* from_y, const float* const* from_z,
                         const size_t* M,
                         const float* const* to_x, const float* const* to_y, const float* const* to_z,
                         const size_t* N,
                         const double* init_T, const mola_icp_params* p, mola_icp_result* out);

/* ---- device pool: the replica / task parallelism of the nearby-keyframe + loop-closure checks across the GPUs of one
 * node (BASELINE config[3]: 64 independent pairs over 8 x MI355X, no collective).  The reference runs these checks on
 * `worker_pool_past_KFs_`, max(2, hw/2) CPU threads (src/LidarOdometry.cpp:94-96, 704-741); here one handle and one
 * persistent host thread per device slot; a call's pairs are cut into chunks that the slots pull from a shared cursor (a
 * slot whose pairs terminate early serves more of them), each chunk advancing in lockstep through mola_icp_align_batch.
 * Results are those of stand-alone aligns, whichever slot served a pair. */
typedef struct mola_icp_pool mola_icp_pool;
/* devices == NULL or n_devices == 0: every visible device.  A device index may repeat (two handles on one GPU). */
int mola_icp_pool_create(const int* devices, int n_devices, mola_icp_pool** out);
int mola_icp_pool_destroy(mola_icp_pool* pool);
int mola_icp_pool_size(const mola_icp_pool* pool, int* n_handles);
int mola_icp_pool_handle(mola_icp_pool* pool, int i, mola_icp_handle** h);   /* borrowed: owned by the pool */
/* the STATIC round-robin rule (device_of_pair[i] = i mod n_devices): what rounds 1-5 dealt by and what a caller may still use to deal
 * pairs itself; since round 6 mola_icp_pool_align_batch lets the slots pull chunks of pairs from a shared cursor instead */
int mola_icp_pool_assignment(size_t n_pairs, int n_devices, int* device_of_pair);
/* pairs each slot served in the pool's last align_batch call (statistics; n_slots >= the pool's size) */
int mola_icp_pool_last_shares(const mola_icp_pool* pool, size_t* pairs_per_slot, int n_slots);
int mola_icp_pool_align_batch(mola_icp_pool* pool, size_t n_pairs,
                              const float* const* from_x, const float* const* from_y, const float* const* from_z,
                              const size_t* M,
                              const float* const* to_x, const float* const* to_y, const float* const* to_z,
                              const size_t* N,
                              const double* init_T, const mola_icp_params* p, mola_icp_result* out);

/* Loop-closure Monte-Carlo (src/LidarOdometry.cpp:767-788): the SAME pair aligned from n_init initial
 * poses (init_T = n_init x 16), keeping the attempt with the highest goodness (strictly greater, i.e. the
 * first best, as `if (this_icp_out.goodness > icp_out.goodness)` cpp:785).  The clouds are uploaded and
 * prepared once and the guesses are a batch dimension on the device (lockstep, as mola_icp_align_batch; each attempt
 * bit-equal to a stand-alone align from that guess).  out = n_init results (may be NULL), *best_index = winner, -1 if
 * every goodness is 0. */
int mola_icp_align_multi_init(mola_icp_handle* h,
                              const float* from_x, const float* from_y, const float* from_z, size_t M,
                              const float* to_x, const float* to_y, const float* to_z, size_t N,
                              size_t n_init, const double* init_T, const mola_icp_params* p,
                              mola_icp_result* out, mola_icp_result* best, int* best_index);

/* ---- device-resident cloud cache -------------------------------------------
 * Keyframe / scan clouds stay in HBM keyed by a caller id (the reference keeps them in the world model and
 * re-reads them for every nearby-KF / loop-closure ICP: src/LidarOdometry.cpp:384-388, 658-666; in odometry a
 * scan is `to` once and `from` once: cpp:278-279).  A cached cloud is stored raw AND prepared (Hilbert order +
 * tile boxes), so mola_icp_align_cached() does no upload and no sort.  put() with an existing id replaces it.
 * Thread-safe; a cloud being used by a running align stays alive until that align returns. */
int mola_icp_cloud_put(mola_icp_handle* h, uint64_t id, const float* x, const float* y, const float* z, size_t n);
/* mola_icp_cloud_put(to_id, ...) followed by mola_icp_align_cached(from_id, to_id, ...), as ONE call -- the odometry step: every scan is
 * `to` now and `from` for the next one (src/LidarOdometry.cpp:278-279, 384-388).  The new cloud's prepare chain and the align's first
 * launches go down the same stream without a host wait in between (the two separate calls leave the device idle 16-24 us there); the
 * cloud enters the cache under to_id when the call is over.  *put_done (may be NULL) = 1 if it did -- also when the align itself
 * failed; a cloud with non-finite coordinates is refused (MOLA_ICP_E_BADARG) and not cached.  Result = the two calls' result. */
int mola_icp_align_cached_put(mola_icp_handle* h, uint64_t from_id, uint64_t to_id, const float* tx, const float* ty, const float* tz,
                              size_t N, const double init_T[16], const mola_icp_params* p, mola_icp_result* out, int* put_done);
int mola_icp_cloud_drop(mola_icp_handle* h, uint64_t id);        /* MOLA_ICP_E_BADARG if the id is unknown */
int mola_icp_cloud_count(mola_icp_handle* h, size_t* count_out, size_t* device_bytes_out);
/* The device blocks of dropped / replaced clouds are parked per device (up to MOLA_ICP_POOL_MB, default 1024) and handed to
 * the next cloud instead of being freed: hipFree synchronises the whole device, and an odometry stream drops a cloud per scan.
 * mola_icp_device_pool_trim() frees the blocks parked on `device` (-1: on every device) until at most keep_bytes remain there
 * (0: all of them); *parked_bytes_out (may be null) = what is still parked on `device` afterwards (0 for -1).  Only that
 * device is touched: hipFree synchronises the device it frees on.  Process-wide, thread-safe. */
int mola_icp_device_pool_trim(int device, size_t keep_bytes, size_t* parked_bytes_out);
/* mola_icp_align() between two cached clouds (`from` = map, `to` = local) */
int mola_icp_align_cached(mola_icp_handle* h, uint64_t from_id, uint64_t to_id, const double init_T[16],
                          const mola_icp_params* p, mola_icp_result* out);

/* Voxel-grid downsample on the GPU (the decimation step BEFORE the ICP: src/LidarOdometry.cpp:215-224,
 * LidarOdometry.h:76-80, kitti-default.yaml:25-32): one centroid per occupied voxel of edge `voxel_size`, voxel
 * grid anchored at the cloud's minimum corner, output ordered by (ix, iy, iz).  *n_out = number of voxels; at most
 * `capacity` points are written (call with capacity = n to get them all). */
int mola_icp_voxel_downsample(mola_icp_handle* h, const float* x, const float* y, const float* z, size_t n,
                              double voxel_size, float* out_x, float* out_y, float* out_z, size_t capacity,
                              size_t* n_out);

/* ---- resident-cloud API (inputs already in HBM; bench + sharded path) ---
 * *_device take DEVICE pointers (fp32 SoA) that must stay valid until the
 * next set_* / destroy; *_host copy from host memory.
 * ORDERING CONTRACT of *_device: the handle works on its own non-blocking HIP stream (or the one given to
 * mola_icp_set_stream), which has NO implicit ordering against any other stream, the null stream included.  The
 * buffers must be completely written -- the producing stream synchronised, or the producer run on the stream handed
 * to mola_icp_set_stream() -- BEFORE the call that first uses them (the next match / align), and must not be
 * rewritten while an align runs.  (The Python binding synchronises torch's current stream in set_map / set_local.) */
/* Kernel statistics of the following aligns on this handle: HIP events around every matcher launch (ms_nn_kernel) and
 * the executed-pair counters (nn_pairs_evaluated).  OFF by default -- the event packets cost ~8 us per ICP iteration at
 * odometry sizes and the counter read-back a stream synchronisation per align; n_nn_launches is always filled.
 * (The reference's counterpart is its mrpt::system::CTimeLogger profiler, src/LidarOdometry.cpp:296-297, 858.) */
int mola_icp_set_profiling(mola_icp_handle* h, int on);
/* Drop what earlier ALIGNS left behind for the resident clouds in place -- the last pairing (the next launch's seeds), the
 * neighbour lists with their certificates, the plane cache: results of matches at that align's poses -- so that the next
 * align is stateless, as mp2p_icp::ICP::align() keeps nothing between calls (src/LidarOdometry.cpp:869-871).  What belongs to
 * the CLOUDS stays: their prepared (sorted) form and the per-item cost order of the work queue (a schedule made once per cloud
 * pair, like the sort; it never changes a result).  For measurements (bench.py's `value`) and for callers that re-register the
 * same pair from an unrelated guess.
 * mola_icp_forget_cloud_schedule: all of the above AND the cost order -- the next align costs what the very FIRST align on this
 * pair cost (bench.py reports it as `value_first_align_on_pair`). */
int mola_icp_forget_warm_start(mola_icp_handle* h);
int mola_icp_forget_cloud_schedule(mola_icp_handle* h);
int mola_icp_set_map_host(mola_icp_handle* h, const float* x, const float* y, const float* z, size_t M);
int mola_icp_set_map_device(mola_icp_handle* h, const float* dx, const float* dy, const float* dz, size_t M);
int mola_icp_set_local_host(mola_icp_handle* h, const float* x, const float* y, const float* z, size_t N);
int mola_icp_set_local_device(mola_icp_handle* h, const float* dx, const float* dy, const float* dz, size_t N);
/* ---- query sharding helpers (no reference analogue: one align() is serial there) ----------
 * Every rank passes the SAME full scan and keeps a spatially compact shard of it: the slice of rank `rank` of the scan's
 * Hilbert order, computed on the device with the whole scan's bounding box (identical on every rank; the full copy is
 * transient).  A random 1/W subsample would be W times sparser than the map: every query group would sweep W times
 * more map tiles.  mola_icp_local_shard_indices: the shard's points as indices into the full scan (n_shard of them). */
int mola_icp_set_local_shard_host(mola_icp_handle* h, const float* x, const float* y, const float* z, size_t n_total,
                                  int rank, int nranks, size_t* n_shard_out);
int mola_icp_set_local_shard_device(mola_icp_handle* h, const float* dx, const float* dy, const float* dz, size_t n_total,
                                    int rank, int nranks, size_t* n_shard_out);
/* ... or any slice [lo, hi) of that order (0 <= lo <= hi <= n_total): cuts of equal COST instead of equal count.  Where the guess is
 * far off (a rotation error moves distant queries by metres) the matcher's cost per query varies several-fold across the scan and
 * a launch is as long as its slowest rank: the ranks time one iteration on equal-count shards, exchange the W times (one
 * all-reduce of a W-vector) and cut again where the cumulated cost is k / W of the total (sharded.balanced_cuts). */
int mola_icp_set_local_shard_range_host(mola_icp_handle* h, const float* x, const float* y, const float* z, size_t n_total,
                                        size_t lo, size_t hi, size_t* n_shard_out);
int mola_icp_set_local_shard_range_device(mola_icp_handle* h, const float* dx, const float* dy, const float* dz, size_t n_total,
                                          size_t lo, size_t hi, size_t* n_shard_out);
int mola_icp_local_shard_indices(mola_icp_handle* h, int32_t* idx_out);
/* The part of the map this rank's shard can reach: [lo, hi] = the shard's bounding box moved by T and grown by `margin`
 * (mola_icp_shard_reach_box), then only the map points inside it are kept, prepared and searched
 * (mola_icp_set_map_slab_*; configs[4]: a rank does not hold, sort and sweep the 10M-point map eight times over).
 * Pairings still name ORIGINAL map indices.  Exact as long as every pose of the align keeps the shard's reach (its moved
 * box grown by the matcher's gate) inside [lo, hi]: the matcher checks this at every pose and fails with
 * MOLA_ICP_E_BADARG ("outside its map slab") otherwise -- cut again with a larger margin.  A margin of
 * gate + the largest pose correction expected (metres) is the natural choice. */
int mola_icp_shard_reach_box(mola_icp_handle* h, const double T[16], double margin, double lo_out[3], double hi_out[3]);
int mola_icp_set_map_slab_host(mola_icp_handle* h, const float* x, const float* y, const float* z, size_t M,
                               const double lo[3], const double hi[3], size_t* n_kept_out);
int mola_icp_set_map_slab_device(mola_icp_handle* h, const float* dx, const float* dy, const float* dz, size_t M,
                                 const double lo[3], const double hi[3], size_t* n_kept_out);
/* total sizes for the quality ratio when this rank holds only a shard (0 = use own sizes) */
int mola_icp_set_global_sizes(mola_icp_handle* h, uint64_t n_local_total, uint64_t n_map_total);
/* full align on the resident clouds (uses the all-reduce hook if installed) */
int mola_icp_align_resident(mola_icp_handle* h, const double init_T[16], const mola_icp_params* p,
                            mola_icp_result* out);

/* ---- single stages, for parity tests (rows a7, a8 of SURVEY.md §8) ------ */
/* matcher: transform + NN + gate at pose T; optionally copies the pairing to
 * host: idx[i] = map index or -1, d2[i] = squared distance of the NN (N each, may be NULL). */
int mola_icp_match(mola_icp_handle* h, const double T[16], double threshold, int nn_kernel,
                   int32_t* idx_out, float* d2_out, uint64_t* n_pairs_out);
/* accumulation over the stored pairing.  stage 0: unit weights, skips pairs
 * flagged outlier; stage 1: centroid-relative tests/weights with centroids
 * cl,cg (flags new outliers).  reset_outliers!=0 clears the flags first.
 * No all-reduce is applied here. */
int mola_icp_accumulate(mola_icp_handle* h, const mola_icp_params* p, const double Tcur[16], int stage,
                        const double cl[3], const double cg[3], int reset_outliers,
                        double acc_out[MOLA_ICP_NACC]);

/* point-to-plane matcher (mp2p_icp::Matcher_Point2Plane, icpreg:33-39) on the resident clouds at pose T, for
 * parity tests: valid[N], centroid[N*3], normal[N*3], knn_idx[N*p->knn] (original map indices, -1 padded);
 * any output may be NULL.  *n_pairs_out = number of plane pairings. */
int mola_icp_match_planes(mola_icp_handle* h, const double T[16], const mola_icp_params* p, uint8_t* valid,
                          double* centroid, double* normal, int32_t* knn_idx, uint64_t* n_pairs_out);

/* ---- host-side math (no GPU needed) ------------------------------------ */
#define MOLA_ICP_NACC_PLANES 92
/* the quadratic form of the stored plane pairing (after mola_icp_match_planes), for parity tests: 78 terms of the upper
 * triangle of sum(phi phi^T) row-major (a <= b), phi = [n (x) l, n]; 12 of sum(phi d), d = n.c; sum(d^2); the pairing count --
 * what Solver_GaussNewton (icpreg:23-26) iterates on.  No all-reduce is applied here. */
int mola_icp_accumulate_planes(mola_icp_handle* h, double acc_out[MOLA_ICP_NACC_PLANES]);
/* Gauss-Newton (mp2p_icp::Solver_GaussNewton, icpreg:23-26) on the point-to-plane cost given as the quadratic
 * form x^T A x - 2 b^T x + c0 in x = [R row-major, t]: acc = A's upper triangle (78, row-major a<=b), b (12),
 * c0, pair count.  This is what the loop runs after the single device accumulation pass. */
int mola_icp_solve_gauss_newton_planes(const double acc[MOLA_ICP_NACC_PLANES], const double T0[16],
                                       uint32_t max_iterations, double T_out[16], double* final_cost,
                                       uint32_t* iterations_done);
/* Mixed pairings in one solve (`matchers:` is a sequence, params/icp-settings-regular.yaml:28-39; all entries initialised together,
 * src/LidarOdometry.cpp:83-84): adds the share of a point-to-point pairing -- its MOLA_ICP_NACC unit-weight sums, accumulated at
 * pose T -- to the quadratic form above (a point-to-point residual is three plane residuals with the normals e_x, e_y, e_z). */
int mola_icp_mixed_form(const double acc_p2p[MOLA_ICP_NACC], const double T[16], double form_inout[MOLA_ICP_NACC_PLANES]);
/* Horn closed form on an accumulator block (row a9).  cl/cg may be NULL
 * (weighted means of acc).  Returns MOLA_ICP_E_BADARG if W<=0. */
int mola_icp_solve_horn(const double acc[MOLA_ICP_NACC], const double* cl, const double* cg, double T_out[16]);
/* stall test quantities (row a10): |v|,|w| of log(Tprev^-1 * T) */
int mola_icp_stall_deltas(const double T[16], const double Tprev[16], double* d_xyz, double* d_rot);
int mola_icp_se3_log(const double T[16], double out6[6]);
int mola_icp_pose_from_xyzypr(const double xyzypr[6], double T_out[16]);
int mola_icp_pose_to_xyzypr(const double T[16], double xyzypr_out[6]);

/* The iteration-control loop (rows a1, a10, a11, a12) over caller-supplied
 * stages -- the SAME host code mola_icp_align* runs over the HIP stages.
 * Lets the host logic and the sharded reduction be exercised without a GPU. */
typedef struct mola_icp_stage_callbacks {
    /* matcher at pose T with gate `threshold`; stores the pairing; *n_pairs = local count */
    int (*match)(void* user, const double T[16], double threshold, uint64_t* n_pairs);
    /* accumulate over the stored pairing (see mola_icp_accumulate) */
    int (*accumulate)(void* user, const mola_icp_params* p, const double Tcur[16], int stage,
                      const double cl[3], const double cg[3], int reset_outliers,
                      double acc_out[MOLA_ICP_NACC]);
    mola_icp_allreduce_fn allreduce; /* may be NULL */
    void*    user;
    uint64_t n_local_total, n_map_total; /* for the quality ratio */
} mola_icp_stage_callbacks;
int mola_icp_run_loop(const mola_icp_stage_callbacks* cb, const double init_T[16], const mola_icp_params* p,
                      mola_icp_result* out);
/* The LOCKSTEP loop behind mola_icp_align_multi_init / mola_icp_align_batch (n_problems independent point-to-point
 * problems, every stage issued for all problems still iterating) over caller-supplied stages: cb[k] serves problem k
 * (its allreduce member is ignored), init_T = n_problems x 16, out = n_problems results.  Each result is bit-identical
 * to mola_icp_run_loop(&cb[k], ...). */
int mola_icp_run_loop_batch(const mola_icp_stage_callbacks* cb, size_t n_problems, const double* init_T,
                            const mola_icp_params* p, mola_icp_result* out);

/* ======================================================================================
 * Front-end logic around the ICP: LidarOdometry::doProcessNewObservation()
 * (src/LidarOdometry.cpp:190-514) without MOLA's back-end / world-model / GUI plumbing: time gate,
 * constant-velocity guess, run_one_icp, twist update, keyframe decision, relative-pose factor.
 * ====================================================================================== */
typedef struct mola_lo_params {
    double min_time_between_scans;          /* [s]   LidarOdometry.h:57,  kitti-default.yaml:5, used cpp:202-206 */
    double min_dist_xyz_between_keyframes;  /* [m]   h:61,  kitti:8,  used cpp:336 */
    double min_rotation_between_keyframes;  /* [rad] h:66 (YAML in degrees, cpp:106), used cpp:337 */
    double min_icp_goodness;                /* h:70, kitti:12, used cpp:334 */
    /* the three ICP cases (LidarOdometry.h:96-102): each = {ICP object settings, default mp2p_icp::Parameters}.
     * The odometry path ALWAYS runs the LidarOdometry object (icp_with_vel's matcher/solver/quality: cpp:869 with
     * in.align_kind left at its default, h:118) and swaps only the mp2p_icp::Parameters half (cpp:287-290). */
    mola_icp_params icp_with_vel;           /* params_.icp[LidarOdometry], cpp:122-124 */
    mola_icp_params icp_without_vel;        /* params_.icp[NearbyAlign],   cpp:125-126 */
    mola_icp_params icp_loop_closure;       /* params_.icp[LoopClosure],   cpp:127-128 */
    /* nearby-KF / loop-closure policy (h:70-93, read at cpp:110-117) */
    double   min_icp_goodness_lc;                     /* h:73,  kitti:14, used cpp:805-807 */
    double   min_dist_to_matching;                    /* h:83,  kitti:35, used cpp:574 */
    double   max_dist_to_matching;                    /* h:84,  kitti:36, used cpp:575-576, 592-594 */
    double   max_dist_to_loop_closure;                /* h:85,  kitti:37, used cpp:575-576, 768 */
    uint32_t loop_closure_montecarlo_samples;         /* h:86,  kitti:49, used cpp:774 */
    uint32_t max_nearby_align_checks;                 /* h:87,  kitti:38, used cpp:706-708 */
    uint32_t min_topo_dist_to_consider_loopclosure;   /* h:88,  kitti:39, used cpp:588-589 */
    uint32_t max_kfs_local_graph;                     /* h:90,  used cpp:558 */
} mola_lo_params;

enum {
    MOLA_LO_DROPPED_TOO_SOON = 0,  /* min_time_between_scans gate, cpp:202-212 */
    MOLA_LO_FIRST_SCAN       = 1,  /* no previous cloud: no ICP, first keyframe, cpp:250-257 */
    MOLA_LO_ICP_RAN          = 2,
    MOLA_LO_EMPTY_CLOUD      = 3   /* cpp:238-245 */
};

typedef struct mola_lo_step {
    int32_t  status;                  /* MOLA_LO_* */
    int32_t  used_with_vel_params;    /* which parameter set the ICP got (cpp:287-290) */
    double   dt;                      /* seconds since the previous processed scan (cpp:268-269) */
    double   rel_pose[16];            /* icp_out.found_pose_to_wrt_from mean (cpp:301-302) */
    double   twist[4];                /* last_iter_twist vx,vy,vz,wz after the update (cpp:305-311) */
    double   dist_since_last_kf;      /* cpp:322-323 */
    double   rot_since_last_kf;       /* cpp:324-327 */
    int32_t  keyframe_created;        /* cpp:333-337 / 250-257 */
    int32_t  kf_factor_valid;         /* a FactorRelativePose3 would be emitted (cpp:433-443) */
    uint64_t kf_factor_from, kf_factor_to;
    double   kf_factor_pose[16];      /* accum_since_last_kf at keyframe creation */
    uint64_t reference_kf;            /* advertiseUpdatedLocalization.reference_kf (cpp:487) */
    double   accum_since_last_kf[16]; /* ... .pose (cpp:488), after a possible reset (cpp:472-474) */
    mola_icp_result icp;              /* the ICP's own result (goodness = icp.quality) */
} mola_lo_step;

typedef struct mola_lo mola_lo;
/* align function with mola_icp_align's meaning, for hosts/tests that supply their own ICP */
typedef int (*mola_lo_align_fn)(void* user, const float* from_x, const float* from_y, const float* from_z, size_t M,
                                const float* to_x, const float* to_y, const float* to_z, size_t N,
                                const double init_T[16], const mola_icp_params* p, mola_icp_result* out);

/* ---- nearby-keyframe / loop-closure policy around the ICP -------------------------
 * Host logic on plain arrays; the pose graph itself (MRPT CNetworkOfPoses, Dijkstra) stays with the caller, who
 * hands in, per keyframe of the local graph, its Euclidean and topological distance to the current keyframe. */
typedef struct mola_lo_kf_candidate {
    uint64_t kf_id;
    double   eucl_dist;        /* |pose of the KF w.r.t. the current KF|, cpp:551 */
    uint32_t topo_dist;        /* Dijkstra hop count, cpp:542-550 */
    int32_t  already_checked;  /* pair in checked_KF_pairs, cpp:600-604 */
} mola_lo_kf_candidate;
/* LidarOdometry::checkForNearbyKFs, the selection half (src/LidarOdometry.cpp:572-599, 679-697, 700-729): keyframes in
 * the band [min_dist_to_matching, max(max_dist_to_loop_closure, max_dist_to_matching)] in order of distance; one whose
 * topological distance reaches min_topo_dist_to_consider_loopclosure is a loop-closure candidate, the others are
 * nearby checks only up to max_dist_to_matching.  Nearby checks are thinned with the stride
 * max(1, n / max_nearby_align_checks); of the loop-closure candidates only the closest is sent.
 * *n_nearby = number selected (MOLA_ICP_E_BADARG if it exceeds nearby_capacity; the count is still set). */
int mola_lo_select_checks(const mola_lo_params* p, const mola_lo_kf_candidate* kfs, size_t n_kfs,
                          uint64_t* nearby_ids, size_t nearby_capacity, size_t* n_nearby,
                          uint64_t* loop_closure_id, int* has_loop_closure);
/* The Monte-Carlo guesses of a loop-closure check (cpp:767-783): sample i = init + N(0, 0.1*max_dist_to_loop_closure)
 * on x, y, z and N(0, 2 deg) on yaw, drawn in that order from a seeded in-repo generator (the reference's is
 * time-seeded).  guesses_xyzypr (n x 6) and guesses_T (n x 16, row-major) may each be NULL. */
int mola_lo_montecarlo_guesses(const double init_xyzypr[6], double max_dist_to_loop_closure, uint32_t n_samples,
                               uint64_t seed, double* guesses_xyzypr, double* guesses_T);
typedef struct mola_lo_check_result {
    mola_icp_result icp;          /* the kept attempt (goodness = icp.quality); quality 0 if none was better than 0 */
    int32_t  best_guess;          /* index of the kept Monte-Carlo sample, 0 for a nearby check, -1: none */
    uint32_t n_attempts;
    double   init_guess_used[6];  /* d->init_guess_to_wrt_from as the accept test sees it: for a loop closure the LAST
                                     sample's guess (the reference overwrites it in the loop, cpp:776-780) */
    double   correction_percent;  /* cpp:795-798 */
    int32_t  edge_accepted;       /* a FactorRelativePose3(from, to, rel_pose) would be added, cpp:815-817 */
} mola_lo_check_result;
/* LidarOdometry::doCheckForNonAdjacentKFs (cpp:743-848) without the back-end calls: a nearby check is one ICP with the
 * NearbyAlign case; a loop closure runs loop_closure_montecarlo_samples perturbed guesses with the LoopClosure case
 * as ONE batched device problem (mola_icp_align_multi_init) and keeps the first best goodness.  `icp` may be NULL if
 * align_cb is given (then the attempts run one after another through it). */
int mola_lo_check_nonadjacent(mola_icp_handle* icp, mola_lo_align_fn align_cb, void* user, const mola_lo_params* lp,
                              int is_loop_closure,
                              const float* from_x, const float* from_y, const float* from_z, size_t M,
                              const float* to_x, const float* to_y, const float* to_z, size_t N,
                              const double init_xyzypr[6], uint64_t seed, mola_lo_check_result* out);

int mola_lo_params_default(mola_lo_params* p);
/* reads the keys LidarOdometry::initialize() reads (cpp:105-128) from a kitti-default.yaml-style file */
int mola_lo_params_from_yaml_file(const char* path, const char* mola_dir, mola_lo_params* p);
/* icp: the AlignKind::LidarOdometry ICP object (may be NULL if align_cb is given) */
int mola_lo_create(mola_icp_handle* icp, mola_lo_align_fn align_cb, void* user, const mola_lo_params* params,
                   mola_lo** out);
int mola_lo_destroy(mola_lo* lo);
int mola_lo_reset(mola_lo* lo);   /* LidarOdometry::reset(), cpp:160 */
/* one observation: timestamp [s] + the (already filtered) cloud in the sensor/vehicle frame */
int mola_lo_process_scan(mola_lo* lo, double timestamp, const float* x, const float* y, const float* z, size_t n,
                         mola_lo_step* out);

#ifdef __cplusplus
}
#endif
#endif /* MOLA_ICP_AMD_H */
