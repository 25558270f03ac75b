// Microbenchmark: how long the device waits for the host between two dependent kernels of an iteration loop
//   (a) the plain turn: kernel A publishes to pinned memory -> the host sees it, "solves" -> hipLaunchKernel(B) -> B starts
//   (b) the gated turn: [A][gate][B] are all in the queue; the host sees A's flag, "solves", writes the pose + a flag into pinned
//       memory; the one-wave gate kernel spinning on that flag copies the pose to device memory and ends; B starts behind it
// Reported: A's end -> B's start on the device clock (wall_clock64, 100 MHz), median over the repetitions, for a B of 1 024 workgroups.
// Build: hipcc -O3 --offload-arch=gfx950 gate_latency.hip -o gate_latency
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct HostSlot { volatile unsigned long long flag; volatile unsigned int ctl; float pose[12]; volatile unsigned long long a_done; };
struct DevSlot { float pose[12]; unsigned int ctl; unsigned long long t_a_end, t_b_start, t_gate_seen; };

__global__ void k_a(HostSlot* h, DevSlot* d, unsigned long long seq, int spin)
{
    // a stand-in for the row reduction: a little work, then the hand-over to the host
    unsigned long long t = wall_clock64();
    while (wall_clock64() - t < (unsigned long long)spin) {}
    if (threadIdx.x == 0) {
        d->t_a_end = wall_clock64();
        h->a_done = seq;
        __threadfence_system();
    }
}

__global__ void k_gate(HostSlot* h, DevSlot* d, unsigned long long seq, unsigned long long timeout_ticks)
{
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    unsigned int ctl = 3u;   // timed out
    for (;;) {
        if (__atomic_load_n(&h->flag, __ATOMIC_ACQUIRE) >= seq) { ctl = h->ctl; break; }
        if (wall_clock64() - t0 > timeout_ticks) break;
        __builtin_amdgcn_s_sleep(2);
    }
    d->t_gate_seen = wall_clock64();
    for (int k = 0; k < 12; ++k) d->pose[k] = h->pose[k];
    d->ctl = ctl;
}

__global__ __launch_bounds__(256) void k_b(const DevSlot* gate, DevSlot* d, float* out, float px)
{
    float p0 = px;
    if (gate) { if (gate->ctl != 1u) return; p0 = gate->pose[0]; }
    if (blockIdx.x == 0 && threadIdx.x == 0) d->t_b_start = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = p0 + (float)threadIdx.x;
}

// (c) no gate kernel: B itself waits -- its first workgroup polls the pinned flag, copies the pose into device memory and raises a device
//     flag there; every other workgroup polls that one (L2)
struct DevGate { unsigned long long flag; unsigned int ctl; float pose[12]; };
__global__ __launch_bounds__(256) void k_b_self(HostSlot* h, DevGate* g, DevSlot* d, float* out, unsigned long long seq, unsigned long long timeout_ticks)
{
    __shared__ float s_pose[12];
    __shared__ unsigned int s_ctl;
    if (threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        unsigned int ctl = 3u;
        if (blockIdx.x == 0) {
            for (;;) {
                if (__atomic_load_n(&h->flag, __ATOMIC_ACQUIRE) >= seq) { ctl = h->ctl; break; }
                if (wall_clock64() - t0 > timeout_ticks) break;
                __builtin_amdgcn_s_sleep(1);
            }
            for (int k = 0; k < 12; ++k) { const float v = h->pose[k]; g->pose[k] = v; s_pose[k] = v; }
            g->ctl = ctl;
            __atomic_store_n(&g->flag, seq, __ATOMIC_RELEASE);
        } else {
            for (;;) {
                if (__atomic_load_n(&g->flag, __ATOMIC_ACQUIRE) >= seq) { ctl = g->ctl; break; }
                if (wall_clock64() - t0 > timeout_ticks) break;
                __builtin_amdgcn_s_sleep(1);
            }
            for (int k = 0; k < 12; ++k) s_pose[k] = g->pose[k];
        }
        s_ctl = ctl;
    }
    __syncthreads();
    if (s_ctl != 1u) return;
    if (blockIdx.x == 1023 && threadIdx.x == 0) d->t_b_start = wall_clock64();   // (the LAST workgroup's release)
    out[blockIdx.x * 256 + threadIdx.x] = s_pose[0] + (float)threadIdx.x;
}

int main()
{
    HostSlot* h; DevSlot* d; float* out; DevSlot* hd;
    CHK(hipHostMalloc((void**)&h, sizeof(HostSlot), hipHostMallocMapped | hipHostMallocCoherent));
    CHK(hipHostMalloc((void**)&hd, sizeof(DevSlot), hipHostMallocMapped | hipHostMallocCoherent));
    CHK(hipMalloc((void**)&d, sizeof(DevSlot)));
    CHK(hipMalloc((void**)&out, sizeof(float) * 1024 * 256));
    CHK(hipMemset(d, 0, sizeof(DevSlot)));
    h->flag = 0; h->ctl = 0; h->a_done = 0;
    hipStream_t st; CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int reps = 200;
    auto solve = []() {   // ~2 us of host work between seeing the sums and knowing the pose
        const auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < 2.0) {}
    };
    DevGate* dg; CHK(hipMalloc((void**)&dg, sizeof(DevGate))); CHK(hipMemset(dg, 0, sizeof(DevGate)));
    for (int mode = 0; mode < 3; ++mode) {
        std::vector<double> turn, seen;
        unsigned long long seq = 1000ull * (mode + 1);
        for (int r = 0; r < reps; ++r) {
            ++seq;
            hipLaunchKernelGGL(k_a, dim3(1), dim3(64), 0, st, h, d, seq, 500 /*5 us*/);
            if (mode == 2) hipLaunchKernelGGL(k_b_self, dim3(1024), dim3(256), 0, st, h, dg, d, out, seq, 100000000ull / 2);
            if (mode == 1) {
                hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, st, h, d, seq, 100000000ull / 2 /*0.5 s*/);
                hipLaunchKernelGGL(k_b, dim3(1024), dim3(256), 0, st, (const DevSlot*)d, d, out, 0.f);
            }
            while (h->a_done < seq) {}
            solve();
            if (mode == 0) {
                hipLaunchKernelGGL(k_b, dim3(1024), dim3(256), 0, st, (const DevSlot*)nullptr, d, out, 1.0f);
            } else {
                h->pose[0] = 1.0f; h->ctl = 1u;
                __atomic_store_n((unsigned long long*)&h->flag, seq, __ATOMIC_RELEASE);
            }
            CHK(hipMemcpyAsync(hd, d, sizeof(DevSlot), hipMemcpyDeviceToHost, st));
            CHK(hipStreamSynchronize(st));
            if (r >= 20) {
                turn.push_back((double)(hd->t_b_start - hd->t_a_end) / 100.0);
                if (mode == 1) seen.push_back((double)(hd->t_gate_seen - hd->t_a_end) / 100.0);
            }
        }
        std::sort(turn.begin(), turn.end());
        printf("%s: A's end -> B's start  median %.2f us  p10 %.2f  p90 %.2f", mode == 0 ? "plain launch behind the host's turn" : (mode == 1 ? "gate + B already queued       " : "B queued, waits by itself     "),
               turn[turn.size() / 2], turn[turn.size() / 10], turn[turn.size() * 9 / 10]);
        if (mode == 1) { std::sort(seen.begin(), seen.end()); printf("   (gate saw the flag %.2f us after A's end)", seen[seen.size() / 2]); }
        printf("\n");
    }
    // a cancelled B: how long the queue is held by a launch that returns at once
    {
        unsigned long long seq = 5000;
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        CHK(hipMemset(d, 0, sizeof(DevSlot)));   // ctl = 0: every workgroup of B returns at once
        CHK(hipEventRecord(e0, st));
        for (int r = 0; r < 100; ++r) hipLaunchKernelGGL(k_b, dim3(1875), dim3(256), 0, st, (const DevSlot*)d, d, out, 0.f);
        CHK(hipEventRecord(e1, st));
        CHK(hipStreamSynchronize(st));
        float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("a cancelled launch of 1 875 workgroups (returns at its first instruction): %.2f us each, back to back\n", ms * 10.0);
        (void)seq;
    }
    return 0;
}
