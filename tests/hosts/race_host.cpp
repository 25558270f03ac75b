// race_host.cpp -- TEST INFRASTRUCTURE: drives the library's C-ABI from 8 threads at once over the REAL host code (csrc/c_api.cpp,
// icp_loop.cpp, ...) linked against tests/hosts/fake_backend.cpp (host-memory stages), so that the threaded scaffolding that only runs
// with a device -- workspace leases, mola_icp_align_batch's lockstep lanes + prepare-ahead tasks + worker pool, mola_icp_align_multi_init,
// the cloud cache under concurrent put / align_cached / align_cached_put / drop -- runs under ThreadSanitizer and ASan + UBSan
// (tests/hosts/Makefile.race, tools/sanitize.sh).  The reference's contract: one ICP object, many caller threads
// (src/LidarOdometry.cpp:94-96, 869).  Every threaded result is compared bit for bit with the serial run of the same call.
// Prints the entry points it went through (the coverage list of the sanitizer reports) and exits non-zero on any mismatch.
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "../../include/mola_icp_amd.h"

namespace mola_icp_amd { extern std::atomic<long> g_fake_peak_workspaces, g_fake_builds, g_fake_matches; }

namespace {
struct Cloud { std::vector<float> x, y, z; };
struct Pair { Cloud g, l; double T0[16]; };

void pose(double x, double y, double z, double yaw, double T[16])
{
    const double xyzypr[6] = {x, y, z, yaw, 0.002, -0.001};
    mola_icp_pose_from_xyzypr(xyzypr, T);
}

Pair make_pair(unsigned seed, int M, int N)
{
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> u(-5.f, 5.f);
    std::normal_distribution<float> noise(0.f, 0.01f);
    Pair p;
    for (int i = 0; i < M; ++i) { p.g.x.push_back(u(rng)); p.g.y.push_back(u(rng)); p.g.z.push_back(0.3f * u(rng)); }
    double Tgt[16];
    pose(0.12, -0.07, 0.02, 0.02, Tgt);
    // l = Tgt^-1 (g + noise) for the first N map points
    for (int i = 0; i < N; ++i) {
        const double g[3] = {p.g.x[i] + noise(rng) - Tgt[3], p.g.y[i] + noise(rng) - Tgt[7], p.g.z[i] + noise(rng) - Tgt[11]};
        p.l.x.push_back((float)(Tgt[0] * g[0] + Tgt[4] * g[1] + Tgt[8] * g[2]));
        p.l.y.push_back((float)(Tgt[1] * g[0] + Tgt[5] * g[1] + Tgt[9] * g[2]));
        p.l.z.push_back((float)(Tgt[2] * g[0] + Tgt[6] * g[1] + Tgt[10] * g[2]));
    }
    pose(0, 0, 0, 0, p.T0);
    return p;
}

bool same(const mola_icp_result& a, const mola_icp_result& b)
{
    return std::memcmp(a.T, b.T, sizeof a.T) == 0 && a.n_iterations == b.n_iterations && a.termination == b.termination && a.quality == b.quality &&
           a.n_pairs == b.n_pairs;
}

std::atomic<int> g_fail{0};
#define CHECK(cond, what)                                                                             \
    do {                                                                                              \
        if (!(cond)) { g_fail.fetch_add(1); std::fprintf(stderr, "MISMATCH %s (line %d): %s\n", what, __LINE__, mola_icp_last_error()); } \
    } while (0)
}  // namespace

int main(int argc, char** argv)
{
    const int n_threads = argc > 1 ? std::atoi(argv[1]) : 8, rounds = argc > 2 ? std::atoi(argv[2]) : 6;
    mola_icp_handle* h = nullptr;
    if (mola_icp_create(0, &h)) { std::fprintf(stderr, "create: %s\n", mola_icp_last_error()); return 2; }
    mola_icp_params p, pa, pw;
    mola_icp_params_default(&p);
    p.matcher_threshold = 0.8; p.max_iterations = 25; p.quality_threshold = 0.1;
    p.nn_kernel = MOLA_ICP_NN_TILED;      // the lockstep path of mola_icp_align_batch / multi_init (chunks, lanes, prepare-ahead)
    pa = p; pa.nn_kernel = MOLA_ICP_NN_AUTO;   // small clouds under AUTO: the stream-per-pair jobs on the handle's worker pool
    pw = p; pw.use_scale_outlier_detector = 1; pw.scale_outlier_threshold = 1.2;   // the weighted solve's extra passes

    const int n_pairs = 30;   // (> 2 chunks of 12: both lanes and the prepare-ahead tasks run)
    std::vector<Pair> pairs;
    for (int i = 0; i < n_pairs; ++i) pairs.push_back(make_pair(100 + i, 140 + 7 * (i % 9), 90 + 5 * (i % 7)));
    std::vector<const float*> fx, fy, fz, tx, ty, tz;
    std::vector<size_t> M, N;
    std::vector<double> inits;
    for (const Pair& q : pairs) {
        fx.push_back(q.g.x.data()); fy.push_back(q.g.y.data()); fz.push_back(q.g.z.data());
        tx.push_back(q.l.x.data()); ty.push_back(q.l.y.data()); tz.push_back(q.l.z.data());
        M.push_back(q.g.x.size()); N.push_back(q.l.x.size());
        inits.insert(inits.end(), q.T0, q.T0 + 16);
    }
    // ---- the serial reference: every call on its own
    std::vector<mola_icp_result> ref(n_pairs), refw(n_pairs);
    for (int i = 0; i < n_pairs; ++i) {
        CHECK(mola_icp_align(h, fx[i], fy[i], fz[i], M[i], tx[i], ty[i], tz[i], N[i], &inits[16 * i], &p, &ref[i]) == 0, "serial align");
        CHECK(mola_icp_align(h, fx[i], fy[i], fz[i], M[i], tx[i], ty[i], tz[i], N[i], &inits[16 * i], &pw, &refw[i]) == 0, "serial weighted align");
        CHECK(ref[i].n_iterations > 1 && ref[i].quality > 0.5, "the fake stages converge");
    }
    const int n_guess = 6;
    std::vector<double> guesses;
    for (int k = 0; k < n_guess; ++k) { double T[16]; pose(0.02 * k, -0.01 * k, 0, 0.004 * k, T); guesses.insert(guesses.end(), T, T + 16); }
    std::vector<mola_icp_result> ref_mi(n_guess);
    mola_icp_result ref_best;
    int ref_bi = -1;
    CHECK(mola_icp_align_multi_init(h, fx[0], fy[0], fz[0], M[0], tx[0], ty[0], tz[0], N[0], n_guess, guesses.data(), &p, ref_mi.data(), &ref_best, &ref_bi) == 0, "serial multi_init");

    // ---- 8 threads on ONE handle
    std::atomic<long> calls_batch{0}, calls_mi{0}, calls_put{0}, calls_cached{0}, calls_cached_put{0}, calls_drop{0}, calls_align{0}, dropped_under_us{0};
    auto worker = [&](int t) {
        std::mt19937 rng(7000 + t);
        for (int r = 0; r < rounds; ++r) {
            switch ((t + r) % 4) {
                case 0: {   // the whole batch through the lockstep lanes; then a slice of it through the worker pool
                    std::vector<mola_icp_result> out(n_pairs);
                    CHECK(mola_icp_align_batch(h, n_pairs, fx.data(), fy.data(), fz.data(), M.data(), tx.data(), ty.data(), tz.data(), N.data(), inits.data(), &p, out.data()) == 0, "align_batch");
                    for (int i = 0; i < n_pairs; ++i) CHECK(same(out[i], ref[i]), "align_batch result = stand-alone align");
                    const int a = (int)(rng() % (n_pairs - 9));
                    std::vector<mola_icp_result> out2(9);
                    CHECK(mola_icp_align_batch(h, 9, fx.data() + a, fy.data() + a, fz.data() + a, M.data() + a, tx.data() + a, ty.data() + a, tz.data() + a, N.data() + a, inits.data() + 16 * a, &pa, out2.data()) == 0, "align_batch (worker pool)");
                    for (int i = 0; i < 9; ++i) CHECK(same(out2[i], ref[a + i]), "pooled align_batch result = stand-alone align");
                    calls_batch += 2;
                    break;
                }
                case 1: {   // the loop-closure Monte-Carlo batch + a weighted batch
                    std::vector<mola_icp_result> out(n_guess);
                    mola_icp_result best;
                    int bi = -1;
                    CHECK(mola_icp_align_multi_init(h, fx[0], fy[0], fz[0], M[0], tx[0], ty[0], tz[0], N[0], n_guess, guesses.data(), &p, out.data(), &best, &bi) == 0, "multi_init");
                    CHECK(bi == ref_bi && same(best, ref_best), "multi_init best");
                    for (int k = 0; k < n_guess; ++k) CHECK(same(out[k], ref_mi[k]), "multi_init attempt");
                    std::vector<mola_icp_result> outw(14);
                    CHECK(mola_icp_align_batch(h, 14, fx.data(), fy.data(), fz.data(), M.data(), tx.data(), ty.data(), tz.data(), N.data(), inits.data(), &pw, outw.data()) == 0, "weighted align_batch");
                    for (int i = 0; i < 14; ++i) CHECK(same(outw[i], refw[i]), "weighted align_batch result");
                    calls_mi += 1; calls_batch += 1;
                    break;
                }
                case 2: {   // the odometry chain on ids of this thread's own: put, align_cached_put scan after scan, drop behind
                    const uint64_t base = 1000u * (uint64_t)(t + 1) + 100u * (uint64_t)r;
                    CHECK(mola_icp_cloud_put(h, base, fx[1], fy[1], fz[1], M[1]) == 0, "cloud_put");
                    for (int k = 0; k < 5; ++k) {
                        const int i = 1;   // (the same pair: map = pair 1's map under a new id every step)
                        mola_icp_result out;
                        int put_done = 0;
                        CHECK(mola_icp_align_cached_put(h, base, base + 1 + (uint64_t)k, tx[i], ty[i], tz[i], N[i], &inits[16 * i], &p, &out, &put_done) == 0, "align_cached_put");
                        CHECK(put_done == 1 && same(out, ref[i]), "align_cached_put result = stand-alone align");
                        mola_icp_result out2;
                        CHECK(mola_icp_align_cached(h, base, base + 1 + (uint64_t)k, &inits[16 * i], &p, &out2) == 0, "align_cached");
                        CHECK(same(out2, ref[i]), "align_cached result");
                        if (k > 0) CHECK(mola_icp_cloud_drop(h, base + (uint64_t)k) == 0, "cloud_drop (own id)");
                        calls_cached_put += 1; calls_cached += 1; calls_drop += (k > 0);
                    }
                    CHECK(mola_icp_cloud_drop(h, base) == 0 && mola_icp_cloud_drop(h, base + 5) == 0, "cloud_drop (chain ends)");
                    calls_put += 1; calls_drop += 2;
                    break;
                }
                default: {  // SHARED ids, fought over: put / align / drop race; an align either sees both clouds (right result) or says so
                    for (int k = 0; k < 6; ++k) {
                        const uint64_t a = 5 + (rng() % 3), b = 8 + (rng() % 3);
                        (void)mola_icp_cloud_put(h, a, fx[2], fy[2], fz[2], M[2]);
                        (void)mola_icp_cloud_put(h, b, tx[2], ty[2], tz[2], N[2]);
                        mola_icp_result out;
                        const int rc = mola_icp_align_cached(h, a, b, &inits[32], &p, &out);
                        if (rc == 0) CHECK(same(out, ref[2]), "align_cached on shared ids");
                        else { CHECK(rc == MOLA_ICP_E_BADARG, "align_cached on a dropped id is BADARG"); dropped_under_us += 1; }
                        (void)mola_icp_cloud_drop(h, (rng() & 1) ? a : b);   // (may already be gone: BADARG)
                        mola_icp_result o2;
                        CHECK(mola_icp_align(h, fx[3], fy[3], fz[3], M[3], tx[3], ty[3], tz[3], N[3], &inits[48], &pw, &o2) == 0 && same(o2, refw[3]), "align beside the cache traffic");
                        calls_put += 2; calls_cached += 1; calls_drop += 1; calls_align += 1;
                    }
                }
            }
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t) th.emplace_back(worker, t);
    // ... and, beside them, the device pool (round 6: persistent slot threads pulling chunks from a shared cursor) called from two
    // threads at once, one of them in the high-priority class (mola_icp_set_thread_priority is thread-local)
    std::atomic<long> calls_pool{0};
    mola_icp_pool* pool = nullptr;
    const int slots[3] = {0, 0, 0};
    CHECK(mola_icp_pool_create(slots, 3, &pool) == 0, "pool_create");
    auto pool_worker = [&](int t) {
        CHECK(mola_icp_set_thread_priority(t & 1) == 0, "set_thread_priority");
        for (int r = 0; r < rounds; ++r) {
            const int n = 7 + 5 * ((t + r) % 3);
            std::vector<mola_icp_result> out((size_t)n);
            CHECK(mola_icp_pool_align_batch(pool, (size_t)n, fx.data(), fy.data(), fz.data(), M.data(), tx.data(), ty.data(), tz.data(), N.data(), inits.data(), &p, out.data()) == 0, "pool_align_batch");
            for (int i = 0; i < n; ++i) CHECK(same(out[(size_t)i], ref[(size_t)i]), "pool result = stand-alone align");
            size_t shares[3] = {0, 0, 0};
            CHECK(mola_icp_pool_last_shares(pool, shares, 3) == 0, "pool_last_shares");
            calls_pool += 1;
        }
        int hi = -1;
        CHECK(mola_icp_get_thread_priority(&hi) == 0 && hi == (t & 1), "the priority class is the thread's own");
    };
    for (int t = 0; t < 2; ++t) th.emplace_back(pool_worker, t);
    for (auto& x : th) x.join();
    CHECK(mola_icp_pool_destroy(pool) == 0, "pool_destroy");
    CHECK(calls_pool.load() == 2 * rounds, "pool calls");
    for (uint64_t id = 5; id <= 10; ++id) (void)mola_icp_cloud_drop(h, id);
    size_t count = 99, bytes = 0;
    CHECK(mola_icp_cloud_count(h, &count, &bytes) == 0 && count == 0, "every cloud dropped at the end");
    CHECK(mola_icp_destroy(h) == 0, "destroy");
    std::printf("race_host: %d threads x %d rounds on one handle; calls: align_batch %ld, align_multi_init %ld, cloud_put %ld, align_cached %ld, "
                "align_cached_put %ld, cloud_drop %ld, align %ld (aligns that met a dropped id: %ld); fake backend: %ld builds, %ld matches, "
                "%ld workspaces alive at the peak; mismatches: %d\n",
                n_threads, rounds, calls_batch.load(), calls_mi.load(), calls_put.load(), calls_cached.load(), calls_cached_put.load(), calls_drop.load(),
                calls_align.load(), dropped_under_us.load(), mola_icp_amd::g_fake_builds.load(), mola_icp_amd::g_fake_matches.load(),
                mola_icp_amd::g_fake_peak_workspaces.load(), g_fail.load());
    return g_fail.load() ? 1 : 0;
}
