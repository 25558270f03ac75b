/*
 * icp_oracle.c -- CPU ORACLE for the ICP registration hot path.
 *
 * >>> TEST INFRASTRUCTURE ONLY. <<<
 * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
 * load this file's shared object, and only as the checker / reported CPU
 * baseline.  The product (mola-fe-lidar_amd/csrc) never links or calls it.
 *
 * >>> PARITY UNPINNED. <<<
 * The reference repo (MOLAorg/mola-fe-lidar) contains no arithmetic for this
 * path: `LidarOdometry::run_one_icp()` (src/LidarOdometry.cpp:851-895) only
 * calls `mp2p_icp::ICP::align()` (src/LidarOdometry.cpp:869-871).  mp2p_icp
 * and MRPT are un-vendored, un-pinned third-party dependencies
 * (CMakeLists.txt:17-24, mola-module.yml:1-2) that are absent from this image,
 * and the reference ships no tests / golden vectors / datasets.  So this file
 * restates the *published algorithm* of that path (mp2p_icp ~2021 API
 * generation: Matcher_Points_DistanceThreshold -> Solver_Horn ->
 * SE(3)-log stall test -> QualityEvaluator_PairedRatio) and is anchored on
 *   - the call contract at src/LidarOdometry.cpp:851-895 (inputs, pose
 *     convention "to w.r.t. from": include/mola-fe-lidar/LidarOdometry.h:122,131),
 *   - the configuration constants in params/icp-settings-regular.yaml:7-46,
 *   - an independent numpy/scipy implementation (tests/golden/make_golden.py).
 * Items recalled from mp2p_icp/MRPT that cannot be cited into /root/reference
 * are tagged [EXT].
 *
 * Numeric contract shared with the HIP path (so NN indices can be compared
 * bit-exactly):
 *   q   = R*l + t in fp32, each component a left-to-right fmaf chain
 *         a = fmaf(R_r0,lx,t_r); a = fmaf(R_r1,ly,a); a = fmaf(R_r2,lz,a)
 *         with R,t = (float) of the fp64 pose;
 *   d2  = fmaf(dz,dz, fmaf(dy,dy, dx*dx)),  dx = qx - gx (fp32);
 *   NN  = argmin_j d2, ties -> lowest j;   kept iff d2 < (float)(thr*thr);
 *   all sums over pairs in fp64.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ types */

/* mirrors params/icp-settings-regular.yaml:10-21,35-39,46 */
typedef struct orc_params {
    uint32_t max_iterations;        /* params.maxIterations        icpreg:11 */
    double   min_abs_step_trans;    /* params.minAbsStep_trans     icpreg:12 */
    double   min_abs_step_rot;      /* params.minAbsStep_rot       icpreg:13 */
    int32_t  use_scale_outlier_detector; /* icpreg:16 */
    double   scale_outlier_threshold;    /* icpreg:17 */
    int32_t  use_robust_kernel;          /* icpreg:19 */
    double   robust_kernel_param;        /* icpreg:20, RADIANS here */
    double   robust_kernel_scale;        /* icpreg:21 */
    double   matcher_threshold;     /* matchers[].params.threshold / distanceThreshold icpreg:35 */
    uint32_t run_from_iteration;    /* icpreg:38 */
    uint32_t run_up_to_iteration;   /* icpreg:39 (0 = no limit) */
    double   quality_threshold;     /* quality[].params.thresholdDistance icpreg:46 */
    int32_t  fixed_iterations;      /* !=0: disable the stall test (benchmark mode) */
    int32_t  use_kdtree;            /* !=0: kd-tree NN, else O(N*M) brute force */
} orc_params;

/* [EXT] mp2p_icp::IterTermReason */
enum { ORC_TERM_UNDEFINED = 0, ORC_TERM_NO_PAIRINGS = 1, ORC_TERM_SOLVER_ERROR = 2,
       ORC_TERM_MAX_ITERATIONS = 3, ORC_TERM_STALLED = 4 };

/* what run_one_icp consumes: src/LidarOdometry.cpp:873-888 */
typedef struct orc_result {
    double   T[16];          /* optimal_tf.mean, row-major 4x4, pose of `to` wrt `from` */
    double   quality;        /* Results::quality      cpp:873,880 */
    uint32_t n_iterations;   /* Results::nIterations  cpp:886 */
    uint32_t termination;    /* Results::terminationReason cpp:888 */
    uint64_t n_pairs;        /* pairings of the last iteration */
    double   rmse;           /* sqrt(mean d2) of the last iteration's pairings */
    double   kdtree_build_s; /* seconds spent building the kd-tree (0 if brute) */
    double   iter_s;         /* seconds in the iteration loop */
} orc_result;

/* ---- alternative READINGS of the [EXT]-recalled mp2p_icp semantics ------------------------------------------
 * Nothing in /root/reference pins these (the arithmetic lives in the absent mp2p_icp); the defaults (all 0) are the
 * readings the product implements.  tests/test_readings.py switches each one and reports how far the final pose
 * moves on the golden pairs, so the cost of a wrong reading is known.  Process-global: test
 * infrastructure only. */
typedef struct orc_readings {
    int32_t stall_max_abs;        /* 0: |v| and |w| of log(Tprev^-1 T) (norms) vs minAbsStep_*   1: max |component|      */
    int32_t quality_denominator;  /* 0: PairedRatio = pairings / min(N, M)                      1: pairings / N (local) */
    int32_t outlier_single_pass;  /* 0: scale-outlier detector runs the weighted solve twice    1: once                 */
    int32_t gn_right_perturbation;/* 0: Gauss-Newton update T <- exp(d) T (left)                1: T <- T exp(d)        */
    int32_t p2pl_all_inside_gate; /* 0: plane from the >= 3 neighbours inside distanceThreshold 1: needs ALL knn inside  */
} orc_readings;
static orc_readings g_readings = {0, 0, 0, 0, 0};
void orc_set_readings(const orc_readings* r) { if (r) g_readings = *r; else memset(&g_readings, 0, sizeof g_readings); }

#define ORC_NACC 24
/* accumulator block (fp64):
 *  [0] W=sum w  [1..3] sum w*l  [4..6] sum w*g  [7..15] sum w*l*g^T (row-major l_r*g_c)
 *  [16] n pairs with w>0  [17] sum d2 over those pairs  [18..23] sum w*l*l^T (xx,xy,xz,yy,yz,zz) */

/* --------------------------------------------------------- SE(3) helpers */

/* [EXT] MRPT TPose3D(x,y,z,yaw,pitch,roll): R = Rz(yaw)*Ry(pitch)*Rx(roll)
 * (constructed at src/LidarOdometry.cpp:272-275) */
void orc_pose_from_xyzypr(const double p[6], double T[16])
{
    const double cy = cos(p[3]), sy = sin(p[3]);
    const double cp = cos(p[4]), sp = sin(p[4]);
    const double cr = cos(p[5]), sr = sin(p[5]);
    T[0] = cy * cp;  T[1] = cy * sp * sr - sy * cr;  T[2]  = cy * sp * cr + sy * sr;  T[3]  = p[0];
    T[4] = sy * cp;  T[5] = sy * sp * sr + cy * cr;  T[6]  = sy * sp * cr - cy * sr;  T[7]  = p[1];
    T[8] = -sp;      T[9] = cp * sr;                 T[10] = cp * cr;                 T[11] = p[2];
    T[12] = 0; T[13] = 0; T[14] = 0; T[15] = 1;
}

void orc_pose_to_xyzypr(const double T[16], double p[6])
{
    p[0] = T[3]; p[1] = T[7]; p[2] = T[11];
    const double sp = -T[8];
    if (fabs(sp) < 1.0 - 1e-12) {
        p[4] = asin(sp);
        p[3] = atan2(T[4], T[0]);
        p[5] = atan2(T[9], T[10]);
    } else { /* gimbal lock: roll := 0 */
        p[4] = sp > 0 ? M_PI / 2 : -M_PI / 2;
        p[3] = atan2(-T[1], T[5]);
        p[5] = 0;
    }
}

static void mat4_mul(const double A[16], const double B[16], double C[16])
{
    double r[16];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double s = 0;
            for (int k = 0; k < 4; k++) s += A[4 * i + k] * B[4 * k + j];
            r[4 * i + j] = s;
        }
    memcpy(C, r, sizeof r);
}

static void se3_inv(const double T[16], double Ti[16])
{
    double r[16];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r[4 * i + j] = T[4 * j + i];
    for (int i = 0; i < 3; i++)
        r[4 * i + 3] = -(r[4 * i + 0] * T[3] + r[4 * i + 1] * T[7] + r[4 * i + 2] * T[11]);
    r[12] = r[13] = r[14] = 0; r[15] = 1;
    memcpy(Ti, r, sizeof r);
}

/* SE(3) logarithm -> out[0..2] = v (translation part, V^-1 t), out[3..5] = w.
 * [EXT] mrpt::poses::Lie::SE<3>::log ordering (used at src/LidarOdometry.cpp:326-329) */
void orc_se3_log(const double T[16], double out[6])
{
    const double R00 = T[0], R01 = T[1], R02 = T[2], R10 = T[4], R11 = T[5], R12 = T[6],
                 R20 = T[8], R21 = T[9], R22 = T[10];
    double ctr = 0.5 * (R00 + R11 + R22 - 1.0);
    if (ctr > 1) ctr = 1;
    if (ctr < -1) ctr = -1;
    const double theta = acos(ctr);
    double w[3];
    if (theta < 1e-9) {
        w[0] = 0.5 * (R21 - R12); w[1] = 0.5 * (R02 - R20); w[2] = 0.5 * (R10 - R01);
    } else if (M_PI - theta < 1e-6) {
        /* near pi: axis from the diagonal of (R+I)/2 */
        double ax = sqrt(fmax(0, (R00 + 1) / 2)), ay = sqrt(fmax(0, (R11 + 1) / 2)),
               az = sqrt(fmax(0, (R22 + 1) / 2));
        if (ax >= ay && ax >= az) { ay = copysign(ay, R01 + R10); az = copysign(az, R02 + R20); }
        else if (ay >= az)        { ax = copysign(ax, R01 + R10); az = copysign(az, R12 + R21); }
        else                      { ax = copysign(ax, R02 + R20); ay = copysign(ay, R12 + R21); }
        const double n = sqrt(ax * ax + ay * ay + az * az);
        w[0] = theta * ax / n; w[1] = theta * ay / n; w[2] = theta * az / n;
    } else {
        const double k = theta / (2 * sin(theta));
        w[0] = k * (R21 - R12); w[1] = k * (R02 - R20); w[2] = k * (R10 - R01);
    }
    /* V^-1 = I - 1/2 [w]x + c [w]x^2,  c = 1/th^2 * (1 - th*sin(th)/(2(1-cos th))) */
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    double c;
    if (th2 < 1e-4) c = 1.0 / 12.0 + th2 / 720.0 + th2 * th2 / 30240.0; /* series: the closed form cancels */
    else {
        const double th = sqrt(th2);
        c = (1.0 - th * sin(th) / (2 * (1 - cos(th)))) / th2;
    }
    const double t[3] = {T[3], T[7], T[11]};
    /* wxt = w x t ; wxwxt = w x (w x t) */
    const double wxt[3] = {w[1] * t[2] - w[2] * t[1], w[2] * t[0] - w[0] * t[2], w[0] * t[1] - w[1] * t[0]};
    const double wxwxt[3] = {w[1] * wxt[2] - w[2] * wxt[1], w[2] * wxt[0] - w[0] * wxt[2],
                             w[0] * wxt[1] - w[1] * wxt[0]};
    for (int i = 0; i < 3; i++) {
        out[i] = t[i] - 0.5 * wxt[i] + c * wxwxt[i];
        out[3 + i] = w[i];
    }
}

/* delta = T (-) Tprev = Tprev^-1 * T ; returns |v| and |w| of its log.
 * [EXT] mp2p_icp::ICP::align stall test; thresholds icpreg:12-13 */
void orc_stall_deltas(const double T[16], const double Tprev[16], double* d_xyz, double* d_rot)
{
    double Ti[16], D[16], lg[6];
    se3_inv(Tprev, Ti);
    mat4_mul(Ti, T, D);
    orc_se3_log(D, lg);
    if (g_readings.stall_max_abs) {  /* alternative reading: the largest component against the thresholds */
        *d_xyz = fmax(fabs(lg[0]), fmax(fabs(lg[1]), fabs(lg[2])));
        *d_rot = fmax(fabs(lg[3]), fmax(fabs(lg[4]), fabs(lg[5])));
        return;
    }
    *d_xyz = sqrt(lg[0] * lg[0] + lg[1] * lg[1] + lg[2] * lg[2]);
    *d_rot = sqrt(lg[3] * lg[3] + lg[4] * lg[4] + lg[5] * lg[5]);
}

/* ------------------------------------------------ transform + distance */

typedef struct { float R[9]; float t[3]; } pose_f32;

static pose_f32 pose_to_f32(const double T[16])
{
    pose_f32 P;
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) P.R[3 * r + c] = (float)T[4 * r + c];
        P.t[r] = (float)T[4 * r + 3];
    }
    return P;
}

static inline void xform(const pose_f32* P, float lx, float ly, float lz, float* qx, float* qy, float* qz)
{
    float a;
    a = fmaf(P->R[0], lx, P->t[0]); a = fmaf(P->R[1], ly, a); *qx = fmaf(P->R[2], lz, a);
    a = fmaf(P->R[3], lx, P->t[1]); a = fmaf(P->R[4], ly, a); *qy = fmaf(P->R[5], lz, a);
    a = fmaf(P->R[6], lx, P->t[2]); a = fmaf(P->R[7], ly, a); *qz = fmaf(P->R[8], lz, a);
}

static inline float dist2(float qx, float qy, float qz, float gx, float gy, float gz)
{
    const float dx = qx - gx, dy = qy - gy, dz = qz - gz;
    return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

void orc_transform_f32(const double T[16], const float* lx, const float* ly, const float* lz, size_t n,
                       float* qx, float* qy, float* qz)
{
    const pose_f32 P = pose_to_f32(T);
    for (size_t i = 0; i < n; i++) xform(&P, lx[i], ly[i], lz[i], &qx[i], &qy[i], &qz[i]);
}

/* exact O(N*M) nearest neighbour, ties -> lowest index; idx=-1,d2=+inf if M==0 */
void orc_nn_brute(const float* gx, const float* gy, const float* gz, size_t M, const float* qx,
                  const float* qy, const float* qz, size_t N, int32_t* idx, float* d2)
{
    for (size_t i = 0; i < N; i++) {
        float best = INFINITY;
        int32_t bi = -1;
        const float x = qx[i], y = qy[i], z = qz[i];
        for (size_t j = 0; j < M; j++) {
            const float d = dist2(x, y, z, gx[j], gy[j], gz[j]);
            if (d < best) { best = d; bi = (int32_t)j; }
        }
        idx[i] = bi;
        d2[i] = best;
    }
}

/* ------------------------------------------------------------- kd-tree
 * [EXT] the reference's NN search is nanoflann's single-index kd-tree inside
 * mrpt::maps::CPointsMap (leaf_max_size 10, split at the widest dimension).
 * This is an independent implementation with the same structure; it returns
 * exactly the brute-force answer above (incl. the tie rule). */

typedef struct kd_node {
    int32_t left, right;  /* children, -1 for leaf */
    int32_t lo, hi;       /* point range [lo,hi) in the permuted arrays */
    int32_t dim;
    float   split_lo, split_hi; /* max of left child / min of right child along dim */
} kd_node;

typedef struct orc_kdtree {
    size_t   n;
    float   *x, *y, *z;   /* permuted copies */
    int32_t* perm;        /* original index of permuted point */
    kd_node* nodes;
    int32_t  n_nodes, cap_nodes;
    float    bbmin[3], bbmax[3];
} orc_kdtree;

#define KD_LEAF 10

static int32_t kd_new_node(orc_kdtree* t)
{
    if (t->n_nodes == t->cap_nodes) {
        t->cap_nodes = t->cap_nodes ? 2 * t->cap_nodes : 1024;
        t->nodes = (kd_node*)realloc(t->nodes, sizeof(kd_node) * (size_t)t->cap_nodes);
    }
    return t->n_nodes++;
}

static inline float kd_coord(const orc_kdtree* t, int dim, int32_t i)
{
    return dim == 0 ? t->x[i] : (dim == 1 ? t->y[i] : t->z[i]);
}

static inline void kd_swap(orc_kdtree* t, int32_t a, int32_t b)
{
    float f;
    int32_t p;
    f = t->x[a]; t->x[a] = t->x[b]; t->x[b] = f;
    f = t->y[a]; t->y[a] = t->y[b]; t->y[b] = f;
    f = t->z[a]; t->z[a] = t->z[b]; t->z[b] = f;
    p = t->perm[a]; t->perm[a] = t->perm[b]; t->perm[b] = p;
}

/* quickselect: place the k-th smallest (by dim) at position k within [lo,hi) */
static void kd_select(orc_kdtree* t, int dim, int32_t lo, int32_t hi, int32_t k)
{
    while (hi - lo > 1) {
        const int32_t mid = lo + (hi - lo) / 2;
        /* median of three */
        float a = kd_coord(t, dim, lo), b = kd_coord(t, dim, mid), c = kd_coord(t, dim, hi - 1);
        int32_t pi = (a < b) ? ((b < c) ? mid : (a < c ? hi - 1 : lo)) : ((a < c) ? lo : (b < c ? hi - 1 : mid));
        kd_swap(t, pi, hi - 1);
        const float pv = kd_coord(t, dim, hi - 1);
        int32_t s = lo;
        for (int32_t i = lo; i < hi - 1; i++)
            if (kd_coord(t, dim, i) < pv) { kd_swap(t, i, s); s++; }
        kd_swap(t, s, hi - 1);
        if (k == s) return;
        if (k < s) hi = s; else lo = s + 1;
    }
}

static int32_t kd_build_rec(orc_kdtree* t, int32_t lo, int32_t hi)
{
    const int32_t id = kd_new_node(t);
    kd_node nd;
    nd.left = nd.right = -1; nd.lo = lo; nd.hi = hi; nd.dim = 0; nd.split_lo = nd.split_hi = 0;
    if (hi - lo > KD_LEAF) {
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int32_t i = lo; i < hi; i++) {
            if (t->x[i] < mn[0]) mn[0] = t->x[i]; if (t->x[i] > mx[0]) mx[0] = t->x[i];
            if (t->y[i] < mn[1]) mn[1] = t->y[i]; if (t->y[i] > mx[1]) mx[1] = t->y[i];
            if (t->z[i] < mn[2]) mn[2] = t->z[i]; if (t->z[i] > mx[2]) mx[2] = t->z[i];
        }
        int dim = 0;
        if (mx[1] - mn[1] > mx[dim] - mn[dim]) dim = 1;
        if (mx[2] - mn[2] > mx[dim] - mn[dim]) dim = 2;
        if (mx[dim] > mn[dim]) {
            const int32_t mid = lo + (hi - lo) / 2;
            kd_select(t, dim, lo, hi, mid);
            float slo = -INFINITY, shi = INFINITY;
            for (int32_t i = lo; i < mid; i++) if (kd_coord(t, dim, i) > slo) slo = kd_coord(t, dim, i);
            for (int32_t i = mid; i < hi; i++) if (kd_coord(t, dim, i) < shi) shi = kd_coord(t, dim, i);
            nd.dim = dim; nd.split_lo = slo; nd.split_hi = shi;
            t->nodes[id] = nd;
            const int32_t L = kd_build_rec(t, lo, mid);
            const int32_t R = kd_build_rec(t, mid, hi);
            t->nodes[id].left = L;
            t->nodes[id].right = R;
            return id;
        }
    }
    t->nodes[id] = nd;
    return id;
}

orc_kdtree* orc_kdtree_build(const float* gx, const float* gy, const float* gz, size_t M)
{
    orc_kdtree* t = (orc_kdtree*)calloc(1, sizeof *t);
    t->n = M;
    t->x = (float*)malloc(sizeof(float) * (M ? M : 1));
    t->y = (float*)malloc(sizeof(float) * (M ? M : 1));
    t->z = (float*)malloc(sizeof(float) * (M ? M : 1));
    t->perm = (int32_t*)malloc(sizeof(int32_t) * (M ? M : 1));
    for (int k = 0; k < 3; k++) { t->bbmin[k] = INFINITY; t->bbmax[k] = -INFINITY; }
    for (size_t i = 0; i < M; i++) {
        t->x[i] = gx[i]; t->y[i] = gy[i]; t->z[i] = gz[i]; t->perm[i] = (int32_t)i;
        if (gx[i] < t->bbmin[0]) t->bbmin[0] = gx[i]; if (gx[i] > t->bbmax[0]) t->bbmax[0] = gx[i];
        if (gy[i] < t->bbmin[1]) t->bbmin[1] = gy[i]; if (gy[i] > t->bbmax[1]) t->bbmax[1] = gy[i];
        if (gz[i] < t->bbmin[2]) t->bbmin[2] = gz[i]; if (gz[i] > t->bbmax[2]) t->bbmax[2] = gz[i];
    }
    if (M) kd_build_rec(t, 0, (int32_t)M);
    return t;
}

void orc_kdtree_free(orc_kdtree* t)
{
    if (!t) return;
    free(t->x); free(t->y); free(t->z); free(t->perm); free(t->nodes); free(t);
}

typedef struct { float qx, qy, qz; float best; int32_t bi; } kd_query;

/* Lower bounds are kept in fp64 and a branch is pruned only when its bound,
 * deflated by 1e-6 relative, is still > best: a subtree that could hold an
 * fp32 d2 equal to (or rounding below) best is always visited, so the result
 * equals orc_nn_brute including the lowest-index tie rule. */
static void kd_search(const orc_kdtree* t, int32_t id, kd_query* q, double off[3], double bound)
{
    const kd_node* nd = &t->nodes[id];
    if (nd->left < 0) {
        for (int32_t i = nd->lo; i < nd->hi; i++) {
            const float d = dist2(q->qx, q->qy, q->qz, t->x[i], t->y[i], t->z[i]);
            const int32_t oi = t->perm[i];
            if (d < q->best || (d == q->best && oi < q->bi)) { q->best = d; q->bi = oi; }
        }
        return;
    }
    const int dim = nd->dim;
    const double v = dim == 0 ? q->qx : (dim == 1 ? q->qy : q->qz);
    const double d_lo = v - (double)nd->split_lo; /* distance past the left child's max  */
    const double d_hi = v - (double)nd->split_hi; /* distance past the right child's min */
    int32_t near_c, far_c;
    double cut;
    if (d_lo + d_hi < 0) { near_c = nd->left;  far_c = nd->right; cut = d_hi * d_hi; }
    else                 { near_c = nd->right; far_c = nd->left;  cut = d_lo * d_lo; }
    /* the near child inherits the parent's bound */
    kd_search(t, near_c, q, off, bound);
    const double save = off[dim];
    const double fb = bound - save + cut;
    if (fb * (1.0 - 1e-6) <= (double)q->best) {
        off[dim] = cut;
        kd_search(t, far_c, q, off, fb);
        off[dim] = save;
    }
}

void orc_kdtree_nn(const orc_kdtree* t, const float* qx, const float* qy, const float* qz, size_t N,
                   int32_t* idx, float* d2)
{
    for (size_t i = 0; i < N; i++) {
        kd_query q = {qx[i], qy[i], qz[i], INFINITY, -1};
        if (t->n) {
            double off[3] = {0, 0, 0}, b = 0;
            const float qq[3] = {q.qx, q.qy, q.qz};
            for (int k = 0; k < 3; k++) {
                double d = 0;
                if (qq[k] < t->bbmin[k]) d = (double)t->bbmin[k] - qq[k];
                else if (qq[k] > t->bbmax[k]) d = (double)qq[k] - t->bbmax[k];
                off[k] = d * d;
                b += off[k];
            }
            kd_search(t, 0, &q, off, b);
        }
        idx[i] = q.bi;
        d2[i] = q.best;
    }
}

/* ------------------------------------------------ matcher (row a7)
 * [EXT] mp2p_icp::Matcher_Points_DistanceThreshold: for every local point,
 * transform by the current pose, nearest global point, keep iff d < threshold.
 * idx[i] = -1 where rejected.  Returns the number of pairings. */
size_t orc_match(const float* gx, const float* gy, const float* gz, size_t M, const orc_kdtree* tree,
                 const float* lx, const float* ly, const float* lz, size_t N, const double T[16],
                 double threshold, int32_t* idx, float* d2)
{
    const pose_f32 P = pose_to_f32(T);
    const float thr2 = (float)(threshold * threshold);
    size_t kept = 0;
    for (size_t i = 0; i < N; i++) {
        float qx, qy, qz, bd;
        int32_t bi;
        xform(&P, lx[i], ly[i], lz[i], &qx, &qy, &qz);
        if (tree) orc_kdtree_nn(tree, &qx, &qy, &qz, 1, &bi, &bd);
        else orc_nn_brute(gx, gy, gz, M, &qx, &qy, &qz, 1, &bi, &bd);
        if (bi >= 0 && bd < thr2) { idx[i] = bi; d2[i] = bd; kept++; }
        else { idx[i] = -1; d2[i] = bd; }
    }
    return kept;
}

/* ------------------------------------------------ weights + accumulation (row a8)
 * [EXT] mp2p_icp visit_correspondences()/optimal_tf_horn(): unit pair weights;
 * optional scale-based outlier detector on the centroid-relative vectors
 * (|g-cg| vs |l-cl|, ratio > scale_outlier_threshold -> discard; vectors
 * shorter than 1e-4 skipped); optional robust kernel on the angle between the
 * (current-pose-rotated) unit vectors.  With the detector on the solve runs
 * twice: the second pass recomputes the centroids without the first pass's
 * outliers (and may flag more).
 *
 * stage 0: acc over all pairs with idx>=0 and !outlier[i], weight 1, no tests.
 * stage 1: given centroids (cl,cg): apply tests, set outlier[i]=1 for
 *          discarded pairs, accumulate the survivors with their weights. */
void orc_accumulate(const float* lx, const float* ly, const float* lz, const float* gx, const float* gy,
                    const float* gz, const int32_t* idx, const float* d2, size_t N, const orc_params* p,
                    const double Tcur[16], int stage, const double cl[3], const double cg[3],
                    uint8_t* outlier, double acc[ORC_NACC])
{
    for (int k = 0; k < ORC_NACC; k++) acc[k] = 0;
    for (size_t i = 0; i < N; i++) {
        const int32_t j = idx[i];
        if (j < 0) continue;
        if (outlier && outlier[i]) continue;
        const double l[3] = {lx[i], ly[i], lz[i]};
        const double g[3] = {gx[j], gy[j], gz[j]};
        double w = 1.0;
        if (stage == 1) {
            double b[3] = {g[0] - cg[0], g[1] - cg[1], g[2] - cg[2]};
            double r[3] = {l[0] - cl[0], l[1] - cl[1], l[2] - cl[2]};
            const double bn = sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
            const double rn = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
            if (bn < 1e-4 || rn < 1e-4) continue;
            if (p->use_scale_outlier_detector) {
                const double mismatch = (bn > rn ? bn : rn) / (bn > rn ? rn : bn);
                if (mismatch > p->scale_outlier_threshold) {
                    if (outlier) outlier[i] = 1;
                    continue;
                }
            }
            if (p->use_robust_kernel) {
                for (int k = 0; k < 3; k++) { b[k] /= bn; r[k] /= rn; }
                const double r2[3] = {Tcur[0] * r[0] + Tcur[1] * r[1] + Tcur[2] * r[2],
                                      Tcur[4] * r[0] + Tcur[5] * r[1] + Tcur[6] * r[2],
                                      Tcur[8] * r[0] + Tcur[9] * r[1] + Tcur[10] * r[2]};
                double c = r2[0] * b[0] + r2[1] * b[1] + r2[2] * b[2];
                if (c > 1) c = 1;
                if (c < -1) c = -1;
                const double ang = acos(c);
                if (ang > p->robust_kernel_param) {
                    const double e = ang - p->robust_kernel_param;
                    w *= 1.0 / (1.0 + p->robust_kernel_scale * e * e);
                }
            }
        }
        acc[0] += w;
        for (int k = 0; k < 3; k++) { acc[1 + k] += w * l[k]; acc[4 + k] += w * g[k]; }
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) acc[7 + 3 * r + c] += w * l[r] * g[c];
        acc[16] += 1.0;
        acc[17] += (double)d2[i];
        acc[18] += w * l[0] * l[0]; acc[19] += w * l[0] * l[1]; acc[20] += w * l[0] * l[2];
        acc[21] += w * l[1] * l[1]; acc[22] += w * l[1] * l[2]; acc[23] += w * l[2] * l[2];
    }
}

/* ------------------------------------------------ Horn solve (row a9)
 * Horn 1987 closed form: S = sum w (l-cl)(g-cg)^T ; N(S) 4x4 symmetric ;
 * q = eigenvector of the largest eigenvalue ; R(q) ; t = cg - R cl.
 * [EXT] mp2p_icp::Solver_Horn / optimal_tf_horn (scale forced to 1).
 * cl/cg may be NULL (then the weighted means from acc are used). */

static void jacobi_sym4(double A[4][4], double V[4][4], double ev[4])
{
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) V[i][j] = (i == j);
    for (int sweep = 0; sweep < 64; sweep++) {
        double off = 0;
        for (int i = 0; i < 4; i++)
            for (int j = i + 1; j < 4; j++) off += A[i][j] * A[i][j];
        double diag = 0;
        for (int i = 0; i < 4; i++) diag += A[i][i] * A[i][i];
        if (off <= 1e-300 || off < 1e-32 * diag) break;
        for (int p = 0; p < 4; p++)
            for (int q = p + 1; q < 4; q++) {
                if (A[p][q] == 0) continue;
                const double theta = (A[q][q] - A[p][p]) / (2 * A[p][q]);
                const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                const double c = 1 / sqrt(tt * tt + 1), s = tt * c;
                for (int k = 0; k < 4; k++) {
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 4; k++) {
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 4; k++) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < 4; i++) ev[i] = A[i][i];
}

/* Degenerate geometry, judged relatively (the same rule as the product's csrc/se3_math.hpp, kSingularRel): Horn refuses
 * fewer than 3 pairings, an S that is zero to rounding (all queries in one point) and a largest eigenvalue of N(S) that is
 * not separated from the second (a line: S of rank 1); the 6 x 6 solves refuse a pivot not above ORC_SINGULAR_REL of the matrix' largest entry (one
 * plane: rank 3).  The align then ends with SolverError at the last pose solved (src/LidarOdometry.cpp:873-877 is the
 * caller's soft-failure branch). */
#define ORC_SINGULAR_REL 1e-11

int orc_horn(const double acc[ORC_NACC], const double* cl_in, const double* cg_in, double T[16])
{
    const double W = acc[0];
    if (!(W > 0)) return -1;
    if (!(acc[16] >= 3.0)) return -3;
    double cl[3], cg[3];
    for (int k = 0; k < 3; k++) {
        cl[k] = cl_in ? cl_in[k] : acc[1 + k] / W;
        cg[k] = cg_in ? cg_in[k] : acc[4 + k] / W;
    }
    double S[3][3];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
            S[r][c] = acc[7 + 3 * r + c] - cl[r] * acc[4 + c] - acc[1 + r] * cg[c] + W * cl[r] * cg[c];
    const double Sxx = S[0][0], Sxy = S[0][1], Sxz = S[0][2], Syx = S[1][0], Syy = S[1][1], Syz = S[1][2],
                 Szx = S[2][0], Szy = S[2][1], Szz = S[2][2];
    double Nm[4][4] = {{Sxx + Syy + Szz, Syz - Szy, Szx - Sxz, Sxy - Syx},
                       {Syz - Szy, Sxx - Syy - Szz, Sxy + Syx, Szx + Sxz},
                       {Szx - Sxz, Sxy + Syx, -Sxx + Syy - Szz, Syz + Szy},
                       {Sxy - Syx, Szx + Sxz, Syz + Szy, -Sxx - Syy + Szz}};
    {   /* S is a difference of sums: all queries in one point leave rounding noise around zero -- judged against the sums */
        double raw = 0, n_mag = 0;
        for (int k = 0; k < 9; k++) if (fabs(acc[7 + k]) > raw) raw = fabs(acc[7 + k]);
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) if (fabs(W * cl[r] * cg[c]) > raw) raw = fabs(W * cl[r] * cg[c]);
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) if (fabs(Nm[i][j]) > n_mag) n_mag = fabs(Nm[i][j]);
        if (!(n_mag > ORC_SINGULAR_REL * raw)) return -5;
    }
    double V[4][4], ev[4];
    jacobi_sym4(Nm, V, ev);
    int best = 0;
    for (int i = 1; i < 4; i++) if (ev[i] > ev[best]) best = i;
    {
        double second = -1e300, mag = 0;
        for (int i = 0; i < 4; i++) {
            if (i != best && ev[i] > second) second = ev[i];
            if (fabs(ev[i]) > mag) mag = fabs(ev[i]);
        }
        const double gap = mag > 0 ? (ev[best] - second) / mag : 0.0;
        if (!(gap > ORC_SINGULAR_REL)) return -4;
    }
    double qw = V[0][best], qx = V[1][best], qy = V[2][best], qz = V[3][best];
    const double qn = sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
    if (!(qn > 0)) return -2;
    qw /= qn; qx /= qn; qy /= qn; qz /= qn;
    if (qw < 0) { qw = -qw; qx = -qx; qy = -qy; qz = -qz; }
    double R[3][3] = {{1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy)},
                      {2 * (qx * qy + qw * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qw * qx)},
                      {2 * (qx * qz - qw * qy), 2 * (qy * qz + qw * qx), 1 - 2 * (qx * qx + qy * qy)}};
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) T[4 * r + c] = R[r][c];
        T[4 * r + 3] = cg[r] - (R[r][0] * cl[0] + R[r][1] * cl[1] + R[r][2] * cl[2]);
    }
    T[12] = T[13] = T[14] = 0; T[15] = 1;
    return 0;
}

/* one solver invocation on a set of pairings: (optionally two-pass) weights +
 * accumulation + Horn.  Returns 0 ok, <0 solver error.  acc_out = the final
 * accumulators that fed Horn. */
int orc_solve_pairs(const float* lx, const float* ly, const float* lz, const float* gx, const float* gy,
                    const float* gz, const int32_t* idx, const float* d2, size_t N, const orc_params* p,
                    const double Tcur[16], double Tnew[16], double acc_out[ORC_NACC])
{
    double acc[ORC_NACC];
    const int weighted = p->use_scale_outlier_detector || p->use_robust_kernel;
    if (!weighted) {
        orc_accumulate(lx, ly, lz, gx, gy, gz, idx, d2, N, p, Tcur, 0, NULL, NULL, NULL, acc);
        if (acc_out) memcpy(acc_out, acc, sizeof acc);
        return orc_horn(acc, NULL, NULL, Tnew);
    }
    uint8_t* outl = (uint8_t*)calloc(N ? N : 1, 1);
    double cl[3], cg[3];
    int rc = 0;
    const int passes = (p->use_scale_outlier_detector && !g_readings.outlier_single_pass) ? 2 : 1;
    for (int pass = 0; pass < passes; pass++) {
        orc_accumulate(lx, ly, lz, gx, gy, gz, idx, d2, N, p, Tcur, 0, NULL, NULL, outl, acc);
        if (!(acc[0] > 0)) { rc = -1; break; }
        for (int k = 0; k < 3; k++) { cl[k] = acc[1 + k] / acc[0]; cg[k] = acc[4 + k] / acc[0]; }
        orc_accumulate(lx, ly, lz, gx, gy, gz, idx, d2, N, p, Tcur, 1, cl, cg, outl, acc);
        rc = orc_horn(acc, cl, cg, Tnew);
        if (rc) break;
    }
    if (acc_out) memcpy(acc_out, acc, sizeof acc);
    free(outl);
    return rc;
}

/* ------------------------------------------------ quality (row a11)
 * [EXT] mp2p_icp::QualityEvaluator_PairedRatio: run the distance-threshold
 * matcher at the final pose with thresholdDistance (icpreg:46) and return
 * pairings / min(N, M). */
double orc_quality_paired_ratio(const float* gx, const float* gy, const float* gz, size_t M,
                                const orc_kdtree* tree, const float* lx, const float* ly, const float* lz,
                                size_t N, const double T[16], double threshold, int32_t* idx_tmp,
                                float* d2_tmp)
{
    if (!N || !M) return 0;
    const size_t kept = orc_match(gx, gy, gz, M, tree, lx, ly, lz, N, T, threshold, idx_tmp, d2_tmp);
    if (g_readings.quality_denominator) return (double)kept / (double)N;
    return (double)kept / (double)(N < M ? N : M);
}

#include <time.h>
static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* ================================================================================
 * Row f3: the reference's SHIPPED pipeline (params/icp-settings-regular.yaml:23-39):
 * mp2p_icp::Matcher_Point2Plane (knn, distanceThreshold, planeEigenThreshold) +
 * mp2p_icp::Solver_GaussNewton (maxIterations).  [EXT] restated from mp2p_icp ~2021:
 *   matcher: for every local point q = T(+)l: its knn nearest global points (ties: lowest
 *     index), keep those with d2 < thr2, need >= 3; mean + covariance (fp64) of them, eigen
 *     values e0<=e1<=e2; plane iff e0 <= planeEigenThreshold*e2; normal = eigenvector of e0;
 *     pairing iff |n.(q - mean)| <= distanceThreshold;  stored: (l, centroid, normal);
 *   solver: Gauss-Newton on sum (n.(T(+)l - c))^2 from the current pose, <= maxIterations
 *     steps, stop when |delta| < 1e-7.
 * ================================================================================ */

/* k nearest neighbours, brute force; out sorted by (d2, index); n_out <= k */
void orc_knn_brute(const float* gx, const float* gy, const float* gz, size_t M, float qx, float qy, float qz, int k,
                   int32_t* idx, float* d2, int* n_out)
{
    int n = 0;
    for (size_t j = 0; j < M; j++) {
        const float d = dist2(qx, qy, qz, gx[j], gy[j], gz[j]);
        if (n < k || d < d2[n - 1]) { /* strict: on ties the earlier (lower) index stays */
            int pos = n < k ? n : k - 1;
            while (pos > 0 && d < d2[pos - 1]) { d2[pos] = d2[pos - 1]; idx[pos] = idx[pos - 1]; pos--; }
            d2[pos] = d; idx[pos] = (int32_t)j;
            if (n < k) n++;
        }
    }
    *n_out = n;
}

typedef struct { float qx, qy, qz; int k, n; int32_t* idx; float* d2; } kd_knn_query;

static inline int knn_less(float d, int32_t i, float d2, int32_t i2) { return d < d2 || (d == d2 && i < i2); }

static void kd_knn_search(const orc_kdtree* t, int32_t id, kd_knn_query* q, double off[3], double bound)
{
    const kd_node* nd = &t->nodes[id];
    if (nd->left < 0) {
        for (int32_t i = nd->lo; i < nd->hi; i++) {
            const float d = dist2(q->qx, q->qy, q->qz, t->x[i], t->y[i], t->z[i]);
            const int32_t oi = t->perm[i];
            if (q->n < q->k || knn_less(d, oi, q->d2[q->n - 1], q->idx[q->n - 1])) {
                int pos = q->n < q->k ? q->n : q->k - 1;
                while (pos > 0 && knn_less(d, oi, q->d2[pos - 1], q->idx[pos - 1])) {
                    q->d2[pos] = q->d2[pos - 1]; q->idx[pos] = q->idx[pos - 1]; pos--;
                }
                q->d2[pos] = d; q->idx[pos] = oi;
                if (q->n < q->k) q->n++;
            }
        }
        return;
    }
    const int dim = nd->dim;
    const double v = dim == 0 ? q->qx : (dim == 1 ? q->qy : q->qz);
    const double d_lo = v - (double)nd->split_lo, d_hi = v - (double)nd->split_hi;
    int32_t near_c, far_c;
    double cut;
    if (d_lo + d_hi < 0) { near_c = nd->left; far_c = nd->right; cut = d_hi * d_hi; }
    else { near_c = nd->right; far_c = nd->left; cut = d_lo * d_lo; }
    kd_knn_search(t, near_c, q, off, bound);
    const double save = off[dim];
    const double fb = bound - save + cut;
    const double worst = q->n < q->k ? INFINITY : (double)q->d2[q->n - 1];
    if (fb * (1.0 - 1e-6) <= worst) {
        off[dim] = cut;
        kd_knn_search(t, far_c, q, off, fb);
        off[dim] = save;
    }
}

void orc_kdtree_knn(const orc_kdtree* t, float qx, float qy, float qz, int k, int32_t* idx, float* d2, int* n_out)
{
    kd_knn_query q = {qx, qy, qz, k, 0, idx, d2};
    if (t->n) {
        double off[3] = {0, 0, 0}, b = 0;
        const float qq[3] = {qx, qy, qz};
        for (int a = 0; a < 3; a++) {
            double d = 0;
            if (qq[a] < t->bbmin[a]) d = (double)t->bbmin[a] - qq[a];
            else if (qq[a] > t->bbmax[a]) d = (double)qq[a] - t->bbmax[a];
            off[a] = d * d;
            b += off[a];
        }
        kd_knn_search(t, 0, &q, off, b);
    }
    *n_out = q.n;
}

/* symmetric 3x3 eigen decomposition, cyclic Jacobi in fp64; values ascending, vectors in columns */
static void eig_sym3(const double Cin[3][3], double ev[3], double V[3][3])
{
    double A[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) { A[i][j] = Cin[i][j]; V[i][j] = (i == j); }
    for (int sweep = 0; sweep < 32; sweep++) {
        const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        const double dg = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
        if (off == 0 || off < 1e-34 * dg) break;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                if (A[p][q] == 0) continue;
                const double theta = (A[q][q] - A[p][p]) / (2 * A[p][q]);
                const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                const double c = 1 / sqrt(tt * tt + 1), s2 = tt * c;
                for (int k = 0; k < 3; k++) { const double a = A[k][p], b = A[k][q]; A[k][p] = c * a - s2 * b; A[k][q] = s2 * a + c * b; }
                for (int k = 0; k < 3; k++) { const double a = A[p][k], b = A[q][k]; A[p][k] = c * a - s2 * b; A[q][k] = s2 * a + c * b; }
                for (int k = 0; k < 3; k++) { const double a = V[k][p], b = V[k][q]; V[k][p] = c * a - s2 * b; V[k][q] = s2 * a + c * b; }
            }
    }
    int o[3] = {0, 1, 2};
    double d[3] = {A[0][0], A[1][1], A[2][2]};
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2 - i; j++)
            if (d[o[j]] > d[o[j + 1]]) { const int t = o[j]; o[j] = o[j + 1]; o[j + 1] = t; }
    double Vs[3][3];
    for (int k = 0; k < 3; k++) { ev[k] = d[o[k]]; for (int r = 0; r < 3; r++) Vs[r][k] = V[r][o[k]]; }
    memcpy(V, Vs, sizeof Vs);
}

/* the matcher: for every local point its plane (or none).  valid[i]=1 and centroid/normal filled where paired;
 * knn_idx (N x knn, -1 padded) optional.  Returns the number of pairings. */
size_t orc_match_point2plane(const float* gx, const float* gy, const float* gz, size_t M, const orc_kdtree* tree,
                             const float* lx, const float* ly, const float* lz, size_t N, const double T[16],
                             double threshold, double plane_eigen_threshold, int knn, uint8_t* valid,
                             double* centroid /*N x 3*/, double* normal /*N x 3*/, int32_t* knn_idx)
{
    const pose_f32 P = pose_to_f32(T);
    const float thr2 = (float)(threshold * threshold);
    size_t kept = 0;
    int32_t idx[64];
    float d2[64];
    if (knn > 64) knn = 64;
    for (size_t i = 0; i < N; i++) {
        float qx, qy, qz;
        xform(&P, lx[i], ly[i], lz[i], &qx, &qy, &qz);
        int n = 0;
        if (tree) orc_kdtree_knn(tree, qx, qy, qz, knn, idx, d2, &n);
        else orc_knn_brute(gx, gy, gz, M, qx, qy, qz, knn, idx, d2, &n);
        int m = 0;
        while (m < n && d2[m] < thr2) m++;  /* sorted ascending: the neighbours inside the gate */
        if (knn_idx) for (int k = 0; k < knn; k++) knn_idx[i * (size_t)knn + k] = k < m ? idx[k] : -1;
        valid[i] = 0;
        if (g_readings.p2pl_all_inside_gate && m < knn) continue;
        if (m < 3) continue;
        double mean[3] = {0, 0, 0};
        for (int k = 0; k < m; k++) { mean[0] += gx[idx[k]]; mean[1] += gy[idx[k]]; mean[2] += gz[idx[k]]; }
        for (int a = 0; a < 3; a++) mean[a] /= m;
        double C[3][3] = {{0}};
        for (int k = 0; k < m; k++) {
            const double d[3] = {gx[idx[k]] - mean[0], gy[idx[k]] - mean[1], gz[idx[k]] - mean[2]};
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) C[r][c] += d[r] * d[c];
        }
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) C[r][c] /= m;
        double ev[3], V[3][3];
        eig_sym3(C, ev, V);
        if (ev[0] > plane_eigen_threshold * ev[2]) continue;
        const double nrm[3] = {V[0][0], V[1][0], V[2][0]};
        const double dist = fabs(nrm[0] * (qx - mean[0]) + nrm[1] * (qy - mean[1]) + nrm[2] * (qz - mean[2]));
        if (dist > threshold) continue;
        valid[i] = 1;
        for (int a = 0; a < 3; a++) { centroid[3 * i + a] = mean[a]; normal[3 * i + a] = nrm[a]; }
        kept++;
    }
    return kept;
}

static void solve6(double H[6][6], const double g[6], double x[6], int* ok)
{
    double A[6][7];
    for (int i = 0; i < 6; i++) { for (int j = 0; j < 6; j++) A[i][j] = H[i][j]; A[i][6] = g[i]; }
    *ok = 1;
    double scale = 0;
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) if (fabs(H[i][j]) > scale) scale = fabs(H[i][j]);
    for (int c = 0; c < 6; c++) {
        int piv = c;
        for (int r = c + 1; r < 6; r++) if (fabs(A[r][c]) > fabs(A[piv][c])) piv = r;
        if (!(fabs(A[piv][c]) > ORC_SINGULAR_REL * scale)) { *ok = 0; return; }
        if (piv != c) for (int k = 0; k < 7; k++) { const double t = A[c][k]; A[c][k] = A[piv][k]; A[piv][k] = t; }
        for (int r = c + 1; r < 6; r++) {
            const double f = A[r][c] / A[c][c];
            for (int k = c; k < 7; k++) A[r][k] -= f * A[c][k];
        }
    }
    for (int i = 5; i >= 0; i--) {
        double s = A[i][6];
        for (int j = i + 1; j < 6; j++) s -= A[i][j] * x[j];
        x[i] = s / A[i][i];
    }
}

static void se3_exp(const double d[6], double T[16])
{
    const double v[3] = {d[0], d[1], d[2]}, w[3] = {d[3], d[4], d[5]};
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = sqrt(th2);
    double a, b, c;  /* R = I + a W + b W^2 ; V = I + b W + c W^2 */
    if (th < 1e-6) { a = 1 - th2 / 6; b = 0.5 - th2 / 24; c = 1.0 / 6 - th2 / 120; }
    else { a = sin(th) / th; b = (1 - cos(th)) / th2; c = (th - sin(th)) / (th2 * th); }
    const double W[3][3] = {{0, -w[2], w[1]}, {w[2], 0, -w[0]}, {-w[1], w[0], 0}};
    double W2[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) { W2[i][j] = 0; for (int k = 0; k < 3; k++) W2[i][j] += W[i][k] * W[k][j]; }
    for (int i = 0; i < 3; i++) {
        double t = 0;
        for (int j = 0; j < 3; j++) {
            T[4 * i + j] = (i == j) + a * W[i][j] + b * W2[i][j];
            t += ((i == j) + b * W[i][j] + c * W2[i][j]) * v[j];
        }
        T[4 * i + 3] = t;
    }
    T[12] = T[13] = T[14] = 0; T[15] = 1;
}

/* Gauss-Newton on the point-to-plane pairings -- and, when pp_idx is given, on point-to-point pairings as well (mixed pairings in
 * one solve: the reference's `matchers:` is a sequence, params/icp-settings-regular.yaml:28-39, src/LidarOdometry.cpp:83-84; [EXT]
 * mp2p_icp hands every active matcher's pairings to the solver; unit weights): cost = sum (n.(p - c))^2 + sum |p - g|^2, p = T l.
 * A point-to-point residual is three plane residuals with the normals e_x, e_y, e_z.  Left perturbation T <- exp(delta) T.
 * Returns 0 ok. */
int orc_solve_gauss_newton_mixed(const float* lx, const float* ly, const float* lz, size_t N, const uint8_t* valid,
                                 const double* centroid, const double* normal, const int32_t* pp_idx, const float* gx, const float* gy,
                                 const float* gz, const double Tcur[16], uint32_t max_iters, double Tnew[16], double* final_cost,
                                 uint32_t* iters_done)
{
    double T[16];
    memcpy(T, Tcur, sizeof T);
    uint32_t it = 0;
    double cost = 0;
    for (; it < max_iters; it++) {
        double H[6][6] = {{0}}, g[6] = {0};
        cost = 0;
        size_t n = 0;
        for (size_t i = 0; i < N; i++) {
            if (!valid[i]) continue;
            const double l[3] = {lx[i], ly[i], lz[i]};
            double p[3];
            for (int r = 0; r < 3; r++) p[r] = T[4 * r] * l[0] + T[4 * r + 1] * l[1] + T[4 * r + 2] * l[2] + T[4 * r + 3];
            const double* nn = normal + 3 * i;
            const double* cc = centroid + 3 * i;
            const double r0 = nn[0] * (p[0] - cc[0]) + nn[1] * (p[1] - cc[1]) + nn[2] * (p[2] - cc[2]);
            /* d p / d delta = [ I | -[p]x ]  ->  J = [ n , p x n ] */
            double J[6] = {nn[0], nn[1], nn[2], p[1] * nn[2] - p[2] * nn[1], p[2] * nn[0] - p[0] * nn[2],
                           p[0] * nn[1] - p[1] * nn[0]};
            if (g_readings.gn_right_perturbation) {  /* T exp(d): p = R (l + w x l + v) + t  ->  J = [ R^T n , l x R^T n ] */
                double m[3];
                for (int c = 0; c < 3; c++) m[c] = T[c] * nn[0] + T[4 + c] * nn[1] + T[8 + c] * nn[2];
                J[0] = m[0]; J[1] = m[1]; J[2] = m[2];
                J[3] = l[1] * m[2] - l[2] * m[1]; J[4] = l[2] * m[0] - l[0] * m[2]; J[5] = l[0] * m[1] - l[1] * m[0];
            }
            for (int a = 0; a < 6; a++) { g[a] += J[a] * r0; for (int b = 0; b < 6; b++) H[a][b] += J[a] * J[b]; }
            cost += r0 * r0;
            n++;
        }
        for (size_t i = 0; pp_idx && i < N; i++) {
            if (pp_idx[i] < 0) continue;
            const double l[3] = {lx[i], ly[i], lz[i]};
            const double gq[3] = {gx[pp_idx[i]], gy[pp_idx[i]], gz[pp_idx[i]]};
            double p[3];
            for (int r = 0; r < 3; r++) p[r] = T[4 * r] * l[0] + T[4 * r + 1] * l[1] + T[4 * r + 2] * l[2] + T[4 * r + 3];
            for (int k = 0; k < 3; k++) {
                double nn[3] = {0, 0, 0};
                nn[k] = 1.0;
                const double r0 = p[k] - gq[k];
                const double J[6] = {nn[0], nn[1], nn[2], p[1] * nn[2] - p[2] * nn[1], p[2] * nn[0] - p[0] * nn[2], p[0] * nn[1] - p[1] * nn[0]};
                for (int a = 0; a < 6; a++) { g[a] += J[a] * r0; for (int b = 0; b < 6; b++) H[a][b] += J[a] * J[b]; }
                cost += r0 * r0;
            }
            n++;
        }
        if (n < 3) return -1;
        double mg[6], d[6];
        int ok;
        for (int a = 0; a < 6; a++) mg[a] = -g[a];
        solve6(H, mg, d, &ok);
        if (!ok) return -2;
        double E[16], R[16];
        se3_exp(d, E);
        if (g_readings.gn_right_perturbation) mat4_mul(T, E, R);
        else mat4_mul(E, T, R);
        memcpy(T, R, sizeof T);
        double nd = 0;
        for (int a = 0; a < 6; a++) nd += d[a] * d[a];
        if (sqrt(nd) < 1e-7) { it++; break; }
    }
    memcpy(Tnew, T, sizeof T);
    if (final_cost) *final_cost = cost;
    if (iters_done) *iters_done = it;
    return 0;
}

int orc_solve_gauss_newton(const float* lx, const float* ly, const float* lz, size_t N, const uint8_t* valid,
                           const double* centroid, const double* normal, const double Tcur[16], uint32_t max_iters,
                           double Tnew[16], double* final_cost, uint32_t* iters_done)
{
    return orc_solve_gauss_newton_mixed(lx, ly, lz, N, valid, centroid, normal, NULL, NULL, NULL, NULL, Tcur, max_iters, Tnew, final_cost,
                                        iters_done);
}

/* The point matcher's share of a mixed solve under pairingsWeightParameters.use_scale_outlier_detector
 * (params/icp-settings-regular.yaml:14-17): the pairings that would reach Horn's final solve in orc_solve_pairs -- unit-weight
 * centroids, the centroid-relative test (orc_accumulate, stage 1), once more with the centroids of what is left -- stay; every
 * other pairing is removed from idx (-1).  Plane pairings carry no such test.  [EXT-recalled: mp2p_icp applies the detector
 * inside its Horn / OLAE weighting; what its Gauss-Newton solver does with these parameters in the reference's unpinned version
 * is unknown -- DESIGN.md section 8.]  Returns the number of pairings left. */
static size_t drop_scale_outliers(const float* lx, const float* ly, const float* lz, const float* gx, const float* gy, const float* gz,
                                  int32_t* idx, const float* d2, size_t N, const orc_params* p, const double Tcur[16])
{
    uint8_t* outl = (uint8_t*)calloc(N ? N : 1, 1);
    double acc[ORC_NACC], cl[3] = {0, 0, 0}, cg[3] = {0, 0, 0};
    int have = 0;
    for (int pass = 0; pass < 2; pass++) {
        orc_accumulate(lx, ly, lz, gx, gy, gz, idx, d2, N, p, Tcur, 0, NULL, NULL, outl, acc);
        if (!(acc[0] > 0)) { have = 0; break; }
        for (int k = 0; k < 3; k++) { cl[k] = acc[1 + k] / acc[0]; cg[k] = acc[4 + k] / acc[0]; }
        orc_accumulate(lx, ly, lz, gx, gy, gz, idx, d2, N, p, Tcur, 1, cl, cg, outl, acc);
        have = 1;
    }
    size_t kept = 0;
    for (size_t i = 0; i < N; i++) {
        if (idx[i] < 0) continue;
        int keep = have && !outl[i];
        if (keep) {   /* stage 1 also passes over pairings closer than 1e-4 to a centroid, without flagging them */
            const int32_t j = idx[i];
            const double b[3] = {gx[j] - cg[0], gy[j] - cg[1], gz[j] - cg[2]}, r[3] = {lx[i] - cl[0], ly[i] - cl[1], lz[i] - cl[2]};
            if (sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]) < 1e-4 || sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]) < 1e-4) keep = 0;
        }
        if (keep) kept++;
        else idx[i] = -1;
    }
    free(outl);
    return kept;
}

/* align with BOTH matchers active in every iteration (Matcher_Points_DistanceThreshold at p->matcher_threshold +
 * Matcher_Point2Plane at plane_threshold) feeding one Gauss-Newton solve; same loop / stall test / quality as orc_align */
int orc_align_mixed(const float* gx, const float* gy, const float* gz, size_t M, const float* lx, const float* ly,
                    const float* lz, size_t N, const double Tinit[16], const orc_params* p, double plane_threshold,
                    double plane_eigen_threshold, int knn, uint32_t solver_max_iters, orc_result* res)
{
    memset(res, 0, sizeof *res);
    double T[16], Tprev[16];
    memcpy(T, Tinit, sizeof T);
    memcpy(Tprev, Tinit, sizeof T);
    uint8_t* valid = (uint8_t*)calloc(N ? N : 1, 1);
    double* cen = (double*)malloc(sizeof(double) * 3 * (N ? N : 1));
    double* nor = (double*)malloc(sizeof(double) * 3 * (N ? N : 1));
    int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * (N ? N : 1));
    float* d2 = (float*)malloc(sizeof(float) * (N ? N : 1));
    orc_kdtree* tree = NULL;
    if (p->use_kdtree) tree = orc_kdtree_build(gx, gy, gz, M);
    res->termination = ORC_TERM_UNDEFINED;
    uint32_t it = 0;
    const double t0 = now_s();
    for (; it < p->max_iterations; it++) {
        size_t kept_pl = 0, kept_pp = 0;
        if (N && M) {
            kept_pl = orc_match_point2plane(gx, gy, gz, M, tree, lx, ly, lz, N, T, plane_threshold, plane_eigen_threshold, knn, valid, cen,
                                            nor, NULL);
            kept_pp = orc_match(gx, gy, gz, M, tree, lx, ly, lz, N, T, p->matcher_threshold, idx, d2);
        }
        /* the pairings of the iteration = everything the matchers gated (what orc_align reports, and what decides NoPairings); the
         * detector then removes point pairings from the SOLVE only: its survivors' terms form the cost, rmse is taken over them */
        const size_t gated = kept_pl + kept_pp;
        if (N && M && p->use_scale_outlier_detector) kept_pp = drop_scale_outliers(lx, ly, lz, gx, gy, gz, idx, d2, N, p, T);
        if (!gated) { res->termination = ORC_TERM_NO_PAIRINGS; break; }
        if (!(kept_pl + kept_pp)) { res->termination = ORC_TERM_SOLVER_ERROR; break; }   /* gated pairings, none left for the solve */
        double Tn[16], cost;
        if (orc_solve_gauss_newton_mixed(lx, ly, lz, N, valid, cen, nor, idx, gx, gy, gz, T, solver_max_iters, Tn, &cost, NULL)) {
            res->termination = ORC_TERM_SOLVER_ERROR;
            break;
        }
        res->n_pairs = gated;
        res->rmse = sqrt(cost / (double)(kept_pl + kept_pp));
        memcpy(T, Tn, sizeof T);
        double dxyz, drot;
        orc_stall_deltas(T, Tprev, &dxyz, &drot);
        if (!p->fixed_iterations && fabs(dxyz) < p->min_abs_step_trans && fabs(drot) < p->min_abs_step_rot) {
            res->termination = ORC_TERM_STALLED;
            it++;
            break;
        }
        memcpy(Tprev, T, sizeof T);
    }
    if (res->termination == ORC_TERM_UNDEFINED) res->termination = ORC_TERM_MAX_ITERATIONS;
    res->n_iterations = it;
    res->iter_s = now_s() - t0;
    memcpy(res->T, T, sizeof T);
    res->quality = orc_quality_paired_ratio(gx, gy, gz, M, tree, lx, ly, lz, N, T, p->quality_threshold, idx, d2);
    orc_kdtree_free(tree);
    free(idx); free(d2); free(valid); free(cen); free(nor);
    return 0;
}

/* align with the point-to-plane + Gauss-Newton pipeline (same loop / stall test / quality as orc_align) */
int orc_align_p2pl(const float* gx, const float* gy, const float* gz, size_t M, const float* lx, const float* ly,
                   const float* lz, size_t N, const double Tinit[16], const orc_params* p, double plane_eigen_threshold,
                   int knn, uint32_t solver_max_iters, orc_result* res)
{
    memset(res, 0, sizeof *res);
    double T[16], Tprev[16];
    memcpy(T, Tinit, sizeof T);
    memcpy(Tprev, Tinit, sizeof T);
    uint8_t* valid = (uint8_t*)calloc(N ? N : 1, 1);
    double* cen = (double*)malloc(sizeof(double) * 3 * (N ? N : 1));
    double* nor = (double*)malloc(sizeof(double) * 3 * (N ? N : 1));
    orc_kdtree* tree = NULL;
    if (p->use_kdtree) tree = orc_kdtree_build(gx, gy, gz, M);
    res->termination = ORC_TERM_UNDEFINED;
    uint32_t it = 0;
    const double t0 = now_s();
    for (; it < p->max_iterations; it++) {
        size_t kept = 0;
        if (N && M) kept = orc_match_point2plane(gx, gy, gz, M, tree, lx, ly, lz, N, T, p->matcher_threshold,
                                                 plane_eigen_threshold, knn, valid, cen, nor, NULL);
        if (!kept) { res->termination = ORC_TERM_NO_PAIRINGS; break; }
        double Tn[16], cost;
        if (orc_solve_gauss_newton(lx, ly, lz, N, valid, cen, nor, T, solver_max_iters, Tn, &cost, NULL)) {
            res->termination = ORC_TERM_SOLVER_ERROR;
            break;
        }
        res->n_pairs = kept;
        res->rmse = sqrt(cost / (double)kept);
        memcpy(T, Tn, sizeof T);
        double dxyz, drot;
        orc_stall_deltas(T, Tprev, &dxyz, &drot);
        if (!p->fixed_iterations && fabs(dxyz) < p->min_abs_step_trans && fabs(drot) < p->min_abs_step_rot) {
            res->termination = ORC_TERM_STALLED;
            it++;
            break;
        }
        memcpy(Tprev, T, sizeof T);
    }
    if (res->termination == ORC_TERM_UNDEFINED) res->termination = ORC_TERM_MAX_ITERATIONS;
    res->n_iterations = it;
    res->iter_s = now_s() - t0;
    memcpy(res->T, T, sizeof T);
    int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * (N ? N : 1));
    float* d2 = (float*)malloc(sizeof(float) * (N ? N : 1));
    res->quality = orc_quality_paired_ratio(gx, gy, gz, M, tree, lx, ly, lz, N, T, p->quality_threshold, idx, d2);
    orc_kdtree_free(tree);
    free(idx); free(d2); free(valid); free(cen); free(nor);
    return 0;
}

/* ------------------------------------------------ align (rows a1, a10, a12) */

/* [EXT] mp2p_icp::ICP::align(from=global, to=local, init_to_wrt_from, params, result)
 * as called at src/LidarOdometry.cpp:869-871.
 * pose_trace (optional): n_iterations x 16 doubles, the pose after every iteration. */
int orc_align(const float* gx, const float* gy, const float* gz, size_t M, const float* lx, const float* ly,
              const float* lz, size_t N, const double Tinit[16], const orc_params* p, orc_result* res,
              double* pose_trace)
{
    memset(res, 0, sizeof *res);
    double T[16], Tprev[16];
    memcpy(T, Tinit, sizeof T);
    memcpy(Tprev, Tinit, sizeof T);
    int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * (N ? N : 1));
    float* d2 = (float*)malloc(sizeof(float) * (N ? N : 1));
    orc_kdtree* tree = NULL;
    if (p->use_kdtree) {
        const double t0 = now_s();
        tree = orc_kdtree_build(gx, gy, gz, M);
        res->kdtree_build_s = now_s() - t0;
    }
    res->termination = ORC_TERM_UNDEFINED;
    const double t0 = now_s();
    uint32_t it = 0;
    for (; it < p->max_iterations; it++) {
        size_t kept = 0;
        const int run_matcher = it >= p->run_from_iteration &&
                                (p->run_up_to_iteration == 0 || it <= p->run_up_to_iteration);
        if (run_matcher && N && M)
            kept = orc_match(gx, gy, gz, M, tree, lx, ly, lz, N, T, p->matcher_threshold, idx, d2);
        if (!kept) { res->termination = ORC_TERM_NO_PAIRINGS; break; }
        double Tn[16], acc[ORC_NACC];
        if (orc_solve_pairs(lx, ly, lz, gx, gy, gz, idx, d2, N, p, T, Tn, acc)) {
            res->termination = ORC_TERM_SOLVER_ERROR;
            break;
        }
        res->n_pairs = (uint64_t)acc[16];
        res->rmse = acc[16] > 0 ? sqrt(acc[17] / acc[16]) : 0;
        memcpy(T, Tn, sizeof T);
        if (pose_trace) memcpy(pose_trace + 16 * (size_t)it, T, sizeof T);
        double dxyz, drot;
        orc_stall_deltas(T, Tprev, &dxyz, &drot);
        if (!p->fixed_iterations && fabs(dxyz) < p->min_abs_step_trans && fabs(drot) < p->min_abs_step_rot) {
            res->termination = ORC_TERM_STALLED;
            it++; /* this iteration did run */
            break;
        }
        memcpy(Tprev, T, sizeof T);
    }
    if (res->termination == ORC_TERM_UNDEFINED) res->termination = ORC_TERM_MAX_ITERATIONS;
    res->n_iterations = it;
    res->iter_s = now_s() - t0;
    memcpy(res->T, T, sizeof T);
    res->quality = orc_quality_paired_ratio(gx, gy, gz, M, tree, lx, ly, lz, N, T, p->quality_threshold, idx, d2);
    orc_kdtree_free(tree);
    free(idx);
    free(d2);
    return 0;
}

size_t orc_sizeof_params(void) { return sizeof(orc_params); }
size_t orc_sizeof_result(void) { return sizeof(orc_result); }

#ifdef __cplusplus
}
#endif
