/* c_abi_walk.c -- a plain C11 host of include/mola_icp_amd.h: proves the header is C (no C++ leaks), that the library
 * links from C, and walks create -> params_from_yaml -> align along their ERROR paths, which need no GPU.
 * Test infrastructure: built by tests/test_boundary_hosts.py with `gcc -std=c11 -Wall -Wextra -Werror -pedantic`. */
#include <math.h>
#include <stdio.h>
#include <unistd.h>
#include <string.h>

#include "mola_icp_amd.h"

static int fails = 0;
static void expect(int ok, const char* what)
{
    printf("%s %s\n", ok ? "ok  " : "FAIL", what);
    if (!ok) ++fails;
}

static const char* kGood =
    "icp_class: mp2p_icp::ICP\n"
    "params:\n  maxIterations: 100\n  minAbsStep_trans: 5e-5\n  minAbsStep_rot: 1e-5\n"
    "solvers:\n  - class: mp2p_icp::Solver_Horn\n"
    "matchers:\n  - class: mp2p_icp::Matcher_Points_DistanceThreshold\n    params:\n      threshold: 0.75\n"
    "quality:\n  - class: mp2p_icp::QualityEvaluator_PairedRatio\n    params:\n      thresholdDistance: 0.1\n";

int main(void)
{
    mola_icp_params p, q, r;
    mola_icp_result res;
    mola_icp_handle* h = NULL;
    double T[16], g6[6] = {1.0, -2.0, 0.5, 0.3, -0.1, 0.05}, back[6];
    int rc, i;

    expect(mola_icp_abi_version() == MOLA_ICP_ABI_VERSION, "ABI version of the header = the library's");
    expect(mola_icp_status_string(MOLA_ICP_E_NODEVICE) != NULL && strlen(mola_icp_status_string(MOLA_ICP_E_NODEVICE)) > 0, "status strings");
    expect(mola_icp_params_default(&p) == MOLA_ICP_OK && p.max_iterations == 40, "params_default");
    expect(mola_icp_params_default(NULL) == MOLA_ICP_E_BADARG, "params_default(NULL) -> BADARG");

    /* load_icp_set_of_params (src/LidarOdometry.cpp:57-88) */
    expect(mola_icp_params_from_yaml(kGood, &q) == MOLA_ICP_OK && q.max_iterations == 100 && q.matcher_threshold == 0.75, "params_from_yaml");
    rc = mola_icp_params_from_yaml("icp_class: foo::Bar\nparams:\nsolvers:\nmatchers:\nquality:\n", &r);
    expect(rc == MOLA_ICP_E_CONFIG && strstr(mola_icp_last_error(), "foo::Bar") != NULL, "unknown icp_class -> E_CONFIG naming it (cpp:70-75)");
    rc = mola_icp_params_from_yaml("icp_class: mp2p_icp::ICP\nparams:\n  maxIterations: 3\n", &r);
    expect(rc == MOLA_ICP_E_CONFIG && strstr(mola_icp_last_error(), "solvers") != NULL, "missing `solvers` -> E_CONFIG (cpp:80)");
    expect(mola_icp_params_compose(&p, &q, &r) == MOLA_ICP_OK && r.max_iterations == 100 && r.matcher_threshold == p.matcher_threshold,
           "params_compose: Parameters half from the call, pipeline half from the object (cpp:287-290 vs 869)");

    /* pose convention (cpp:272-275) */
    expect(mola_icp_pose_from_xyzypr(g6, T) == MOLA_ICP_OK && mola_icp_pose_to_xyzypr(T, back) == MOLA_ICP_OK, "pose_from/to_xyzypr");
    for (i = 0; i < 6; ++i) expect(fabs(back[i] - g6[i]) < 1e-12, "  round trip component");

    /* create / align error paths */
    expect(mola_icp_create(0, NULL) == MOLA_ICP_E_BADARG, "create(NULL out) -> BADARG");
    memset(&res, 0, sizeof res);
    expect(mola_icp_align(NULL, NULL, NULL, NULL, 0, NULL, NULL, NULL, 0, T, &q, &res) == MOLA_ICP_E_BADARG, "align(NULL handle) -> BADARG");
    rc = mola_icp_create(-1, &h);
    if (rc != MOLA_ICP_OK) {
        expect(rc == MOLA_ICP_E_NODEVICE && h == NULL && strlen(mola_icp_last_error()) > 0, "no GPU: create -> E_NODEVICE with a message, no fallback");
        printf("     (%s)\n", mola_icp_last_error());
    } else {
        const float x[4] = {0.f, 1.f, 0.f, 1.f}, y[4] = {0.f, 0.f, 1.f, 1.f}, z[4] = {0.f, 0.f, 0.f, 0.f};
        q.matcher_threshold = -1.0;
        expect(mola_icp_align(h, x, y, z, 4, x, y, z, 4, T, &q, &res) == MOLA_ICP_E_BADARG, "GPU: negative matcher threshold -> BADARG");
        q.matcher_threshold = 0.75;
        q.max_iterations = 5;
        T[0] = NAN;
        expect(mola_icp_align(h, x, y, z, 4, x, y, z, 4, T, &q, &res) == MOLA_ICP_E_BADARG, "GPU: NaN pose -> BADARG");
        mola_icp_pose_from_xyzypr(g6, T);
        expect(mola_icp_align(h, x, y, z, 4, NULL, y, z, 4, T, &q, &res) == MOLA_ICP_E_BADARG, "GPU: NULL cloud pointer -> BADARG");
        memset(T, 0, sizeof T); T[0] = T[5] = T[10] = T[15] = 1.0;
        expect(mola_icp_align(h, x, y, z, 4, x, y, z, 4, T, &q, &res) == MOLA_ICP_OK && res.quality == 1.0, "GPU: a 4-point align runs");
        expect(mola_icp_destroy(h) == MOLA_ICP_OK, "destroy");
    }

    /* the dealing rule of the device pool and the loop-closure policy: host-only */
    {
        int dev[10];
        mola_lo_params lp;
        mola_lo_kf_candidate kf[3];
        uint64_t ids[3], lc = 0;
        size_t n = 0;
        int has = 0;
        double guesses[2 * 6];
        expect(mola_icp_pool_assignment(10, 4, dev) == MOLA_ICP_OK && dev[0] == 0 && dev[5] == 1 && dev[9] == 1, "pool_assignment round-robin");
        expect(mola_lo_params_default(&lp) == MOLA_ICP_OK, "lo_params_default");
        memset(kf, 0, sizeof kf);
        kf[0].kf_id = 7; kf[0].eucl_dist = 8.0; kf[0].topo_dist = 3;
        kf[1].kf_id = 8; kf[1].eucl_dist = 25.0; kf[1].topo_dist = 40;
        kf[2].kf_id = 9; kf[2].eucl_dist = 2.0; kf[2].topo_dist = 1;
        expect(mola_lo_select_checks(&lp, kf, 3, ids, 3, &n, &lc, &has) == MOLA_ICP_OK && n == 1 && ids[0] == 7 && has && lc == 8, "select_checks");
        expect(mola_lo_montecarlo_guesses(g6, 30.0, 2, 1u, guesses, NULL) == MOLA_ICP_OK && guesses[4] == g6[4] && guesses[0] != g6[0], "montecarlo_guesses");
    }
    /* the node-local communicator with one rank: create (collective of one), reduce, the error paths, destroy: host-only */
    {
        mola_icp_local_comm* c = NULL;
        double v[24];
        int n = 0, k;
        char name[64];
        snprintf(name, sizeof name, "mola_icp_c_walk_%ld", (long)getpid());
        expect(mola_icp_local_comm_create(NULL, 1, 0, 1.0, &c) == MOLA_ICP_E_BADARG, "local_comm_create: NULL name -> BADARG");
        expect(mola_icp_local_comm_create(name, 2, 2, 1.0, &c) == MOLA_ICP_E_BADARG && c == NULL, "local_comm_create: rank out of range -> BADARG");
        expect(mola_icp_local_comm_create(name, 1, 0, 1.0, &c) == MOLA_ICP_OK && c != NULL, "local_comm_create: one rank");
        expect(mola_icp_local_comm_nranks(c, &n) == MOLA_ICP_OK && n == 1, "local_comm_nranks == 1");
        for (k = 0; k < 24; ++k) v[k] = 0.5 * k;
        expect(mola_icp_local_comm_allreduce(c, v, 24) == MOLA_ICP_OK && v[7] == 3.5, "local_comm_allreduce: one rank leaves the block as it is");
        expect(mola_icp_local_comm_allreduce(c, v, 121) == MOLA_ICP_E_BADARG, "local_comm_allreduce: payload beyond 120 -> BADARG");
        expect(mola_icp_local_comm_allreduce(NULL, v, 24) == MOLA_ICP_E_BADARG, "local_comm_allreduce: NULL -> BADARG");
        expect(mola_icp_local_comm_destroy(c) == MOLA_ICP_OK, "local_comm_destroy");
    }
    printf("%s (%d failure%s)\n", fails ? "FAILED" : "PASSED", fails, fails == 1 ? "" : "s");
    return fails ? 1 : 0;
}
