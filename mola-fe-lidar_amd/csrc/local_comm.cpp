// local_comm.cpp -- the node-local all-reduce of the query-sharded path: every rank of ONE node adds
// its fp64 accumulator block (24 sums point-to-point, 92 point-to-plane) to the others' through a POSIX shared-memory mailbox.
//
// Why on the host.  The consumer of the reduced block is each rank's HOST thread (the Horn / Gauss-Newton solve is fp64 on the
// CPU: icp_loop.cpp), and a single-GPU iteration already ends with the reduction kernel writing its block into pinned host memory
// while the host spins on a sequence word (hip_backend.hip, "direct readback").  A device-side collective puts a second hop in
// front of that: RCCL's all-reduce of 192 bytes costs 13-19 us per iteration on this path even with one rank, a hand-written
// peer-write exchange over xGMI still needs an in-kernel wait on the slowest rank before the same host publish.  Here the hop is
// the one the single-GPU loop already pays: every rank takes its own block off its GPU as always, stores it into its mailbox row
// (one cache-line group) and reads the other rows -- sub-microsecond between cores of one node, no GPU work, no launch, no kernel
// that waits on another process.  Every rank adds the rows in rank order 0 .. n-1: the sums are bit-identical on all ranks (they
// must be: each rank solves its own pose from them and the poses have to stay in lockstep).
// Across nodes there is no shared memory: that is what the RCCL communicator (rccl_dl.cpp) stays for.
//
// No reference analogue: mp2p_icp::ICP::align() is one serial call (src/LidarOdometry.cpp:869-871).
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <new>
#include <string>
#include <thread>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/mola_icp_amd.h"
#include "icp_loop.hpp"

namespace mola_icp_amd {

namespace {
constexpr uint64_t kMagic = 0x4d4f4c414c434f4dull;  // "MOLALCOM"
constexpr int kRowDoubles = 120;                    // payload capacity (the plane block is 92)
constexpr int kMaxRanks = 64;

struct alignas(64) Row {
    double v[kRowDoubles];
    std::atomic<uint64_t> seq;   // the all-reduce this row belongs to, stored (release) after v
    uint32_t n;                  // payload length of that all-reduce
    uint32_t status;             // != 0: the rank gave up (its error reaches the others instead of a time-out)
    uint8_t pad[1024 - sizeof(double) * kRowDoubles - sizeof(std::atomic<uint64_t>) - 8];
};
static_assert(sizeof(Row) == 1024, "one row = 16 cache lines");

struct alignas(64) Header {
    std::atomic<uint64_t> magic;     // written last by rank 0: the segment is initialised
    uint32_t nranks;
    std::atomic<uint32_t> joined;    // ranks that mapped the segment
    std::atomic<uint32_t> left;      // ranks that destroyed their end
    std::atomic<uint32_t> ack;       // rank 0, AFTER it has seen every rank join and has unlinked the name: this is the live segment
    uint8_t pad[64 - 8 - 4 * 4];
};
static_assert(sizeof(Header) == 64, "header = one cache line");

inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
}
}  // namespace

class LocalComm {
public:
    ~LocalComm()
    {
        if (base_) {
            hdr()->left.fetch_add(1, std::memory_order_acq_rel);
            munmap(base_, bytes_);
        }
        if (owner_ && !unlinked_) shm_unlink(name_.c_str());
    }

    int init(const char* name, int nranks, int rank, double timeout_s)
    {
        if (!name || !*name) return fail(MOLA_ICP_E_BADARG, "local communicator: empty name");
        if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks)
            return fail(MOLA_ICP_E_BADARG, "local communicator: bad rank / nranks (at most 64 ranks)");
        name_ = name[0] == '/' ? std::string(name) : "/" + std::string(name);
        nranks_ = nranks;
        rank_ = rank;
        timeout_s_ = timeout_s > 0 ? timeout_s : 30.0;
        bytes_ = sizeof(Header) + sizeof(Row) * 2 * (size_t)nranks;
        const auto t_end = std::chrono::steady_clock::now() + std::chrono::duration<double>(timeout_s_);
        if (rank == 0) {
            shm_unlink(name_.c_str());  // a leftover of a crashed run under the same name
            const int fd = shm_open(name_.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
            if (fd < 0) return fail(MOLA_ICP_E_COMM, "local communicator: shm_open(" + name_ + ") failed: " + std::strerror(errno));
            owner_ = true;
            if (ftruncate(fd, (off_t)bytes_) != 0) {
                close(fd);
                return fail(MOLA_ICP_E_COMM, std::string("local communicator: ftruncate failed: ") + std::strerror(errno));
            }
            void* p = mmap(nullptr, bytes_, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            close(fd);
            if (p == MAP_FAILED) return fail(MOLA_ICP_E_COMM, std::string("local communicator: mmap failed: ") + std::strerror(errno));
            base_ = p;
            // (a fresh segment is zero-filled: rows start at seq 0, all-reduces count from 1, nobody has joined, no ack)
            hdr()->nranks = (uint32_t)nranks;
            hdr()->magic.store(kMagic, std::memory_order_release);
            hdr()->joined.fetch_add(1, std::memory_order_acq_rel);
            // collective: nobody leaves before everybody has mapped the segment -- then the name can go (nothing stays in /dev/shm
            // if a rank dies later; the memory lives until the last mapping is gone)
            while (hdr()->joined.load(std::memory_order_acquire) < (uint32_t)nranks) {
                if (std::chrono::steady_clock::now() > t_end)
                    return fail(MOLA_ICP_E_COMM, "local communicator: only " + std::to_string(hdr()->joined.load()) + " of " +
                                                     std::to_string(nranks) + " ranks joined " + name_);
                std::this_thread::sleep_for(std::chrono::microseconds(50));
            }
            // the name goes BEFORE the acknowledgement: a name left behind by a crash can therefore never lead to a segment whose
            // `ack` is set -- which is what lets a late-comer tell a leftover from the live segment (below)
            shm_unlink(name_.c_str());
            unlinked_ = true;
            hdr()->ack.store(1u, std::memory_order_release);
            return MOLA_ICP_OK;
        }
        // Ranks > 0.  The name may not exist yet, may not have its size yet -- or may still lead to the LEFTOVER of a run that crashed
        // during its own init (rank 0 removes it only when it gets here): its size, magic and rank count all check out, and its stale
        // `joined` count can even be complete.  What a leftover never has is rank 0's acknowledgement.  So a rank waits for `ack`
        // in the segment it mapped, and while it waits it keeps looking at what the name leads to NOW: another inode = rank 0 has
        // replaced the segment this rank sits in -- drop it and join the new one.
        for (;;) {
            int fd = -1;
            struct stat st {};
            for (;;) {
                fd = shm_open(name_.c_str(), O_RDWR, 0600);
                if (fd >= 0) {
                    if (fstat(fd, &st) == 0 && (size_t)st.st_size >= bytes_) break;
                    close(fd);
                    fd = -1;
                }
                if (std::chrono::steady_clock::now() > t_end)
                    return fail(MOLA_ICP_E_COMM, "local communicator: rank 0 never created " + name_);
                std::this_thread::sleep_for(std::chrono::milliseconds(1));
            }
            void* p = mmap(nullptr, bytes_, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            close(fd);
            if (p == MAP_FAILED) return fail(MOLA_ICP_E_COMM, std::string("local communicator: mmap failed: ") + std::strerror(errno));
            base_ = p;
            bool joined = false, replaced = false, mismatch = false;
            unsigned spins = 0;
            for (;;) {
                if (!joined && !mismatch && hdr()->magic.load(std::memory_order_acquire) == kMagic) {
                    // another rank count than this rank was told: the LEFTOVER of a crashed run of another size (rank 0 will replace it:
                    // the inode check below) -- or a live segment this rank has no place in.  Only `ack` tells the two apart (a leftover
                    // never has it), so the verdict waits for it; the segment is not joined meanwhile.
                    if (hdr()->nranks != (uint32_t)nranks) mismatch = true;
                    else {
                        hdr()->joined.fetch_add(1, std::memory_order_acq_rel);
                        joined = true;
                    }
                }
                if (mismatch && hdr()->ack.load(std::memory_order_acquire) != 0u) {
                    const uint32_t theirs = hdr()->nranks;
                    munmap(base_, bytes_);
                    base_ = nullptr;
                    return fail(MOLA_ICP_E_COMM, "local communicator: rank 0 created " + name_ + " for " + std::to_string(theirs) +
                                                     " ranks, this rank was told " + std::to_string(nranks));
                }
                if (joined && hdr()->ack.load(std::memory_order_acquire) != 0u) break;
                if ((++spins & 0xff) == 0) {
                    struct stat now {};
                    const int fd2 = shm_open(name_.c_str(), O_RDWR, 0600);
                    if (fd2 >= 0) {
                        const bool other = fstat(fd2, &now) == 0 && (now.st_ino != st.st_ino || now.st_dev != st.st_dev);
                        close(fd2);
                        if (other) { replaced = true; break; }
                    }
                    if (std::chrono::steady_clock::now() > t_end) {
                        // (this rank leaves: its place in the count goes with it, so that a rank 0 that is still waiting fails at ITS
                        //  time-out instead of acknowledging a segment with a rank missing)
                        if (joined) hdr()->joined.fetch_sub(1, std::memory_order_acq_rel);
                        const std::string what = mismatch ? name_ + " was made for " + std::to_string(hdr()->nranks) + " ranks (this rank was told " + std::to_string(nranks) +
                                                                ") and rank 0 never replaced it"
                                                 : joined ? "only " + std::to_string(hdr()->joined.load()) + " of " + std::to_string(nranks) + " ranks joined " + name_ +
                                                              " (or it is the leftover of a crashed run and rank 0 never came)"
                                                          : "rank 0 never initialised " + name_;
                        munmap(base_, bytes_);
                        base_ = nullptr;
                        return fail(MOLA_ICP_E_COMM, "local communicator: " + what);
                    }
                    std::this_thread::sleep_for(std::chrono::microseconds(50));
                } else {
                    cpu_relax();
                }
            }
            if (!replaced) return MOLA_ICP_OK;
            munmap(base_, bytes_);   // (a leftover: its counters are nobody's any more)
            base_ = nullptr;
        }
    }

    // sum buf[0, n) over the ranks, in place; every rank gets the same bits.  A call that fails (MOLA_ICP_E_COMM: a peer ahead, gone
    // or late) leaves this end one all-reduce out of step with the others for good: the communicator is finished -- destroy it and
    // create a new one on every rank (include/mola_icp_amd.h says the same).
    int allreduce(double* buf, int n)
    {
        if (broken_) return fail(MOLA_ICP_E_COMM, "local communicator: an earlier all-reduce failed on this rank -- destroy the communicator and create a new one on every rank");
        if (!base_) return fail(MOLA_ICP_E_BADARG, "local communicator: not initialised");
        if (!buf || n < 1 || n > kRowDoubles) return fail(MOLA_ICP_E_BADARG, "local communicator: payload of 1 .. 120 doubles");
        const uint64_t seq = ++seq_;
        Row* rows = row0() + (seq & 1) * (size_t)nranks_;
        Row& mine = rows[rank_];
        std::memcpy(mine.v, buf, sizeof(double) * (size_t)n);
        mine.n = (uint32_t)n;
        mine.seq.store(seq, std::memory_order_release);
        // (two parities are enough: a rank can only be one all-reduce ahead of the slowest -- it cannot finish seq + 1 before every
        // rank has published seq + 1, i.e. has finished reading seq)
        double sum[kRowDoubles];
        for (int k = 0; k < n; ++k) sum[k] = 0.0;
        std::chrono::steady_clock::time_point t_end{};
        bool timed = false;
        for (int r = 0; r < nranks_; ++r) {
            Row& row = rows[r];
            unsigned spins = 0;
            for (;;) {
                const uint64_t s = row.seq.load(std::memory_order_acquire);
                if (s == seq) break;
                if (s > seq && s != UINT64_MAX)
                    return give_up("local communicator: rank " + std::to_string(r) + " is at all-reduce " + std::to_string(s) + ", this rank at " +
                                   std::to_string(seq) + " (the ranks did not make the same calls)");
                if (s == UINT64_MAX)
                    return give_up("local communicator: rank " + std::to_string(r) + " gave up (its own error says why)", false);
                cpu_relax();
                if ((++spins & 0x3ff) == 0) {
                    const auto now = std::chrono::steady_clock::now();
                    if (!timed) { t_end = now + std::chrono::duration_cast<std::chrono::steady_clock::duration>(std::chrono::duration<double>(timeout_s_)); timed = true; }
                    else if (now > t_end)
                        return give_up("local communicator: rank " + std::to_string(r) + " did not reach all-reduce " + std::to_string(seq) +
                                       " within " + std::to_string(timeout_s_) + " s");
                    if (spins > (1u << 16)) std::this_thread::yield();  // (more ranks than cores: let the others run)
                }
            }
            if (row.n != (uint32_t)n)
                return give_up("local communicator: rank " + std::to_string(r) + " reduces " + std::to_string(row.n) + " values, this rank " +
                               std::to_string(n));
            for (int k = 0; k < n; ++k) sum[k] += row.v[k];
        }
        std::memcpy(buf, sum, sizeof(double) * (size_t)n);
        return MOLA_ICP_OK;
    }

    // this rank cannot go on (an error outside the collective): the others' next all-reduce fails at once instead of timing out
    void abort_peers()
    {
        if (!base_) return;
        for (int par = 0; par < 2; ++par) row0()[par * (size_t)nranks_ + rank_].seq.store(UINT64_MAX, std::memory_order_release);
    }

    int nranks() const { return base_ ? (int)hdr()->joined.load(std::memory_order_acquire) : 0; }
    int rank() const { return rank_; }

private:
    Header* hdr() const { return static_cast<Header*>(base_); }
    Row* row0() const { return reinterpret_cast<Row*>(static_cast<char*>(base_) + sizeof(Header)); }
    int give_up(const std::string& msg, bool tell = true)
    {
        if (tell) abort_peers();
        broken_ = true;
        return fail(MOLA_ICP_E_COMM, msg);
    }

    std::string name_;
    void* base_ = nullptr;
    size_t bytes_ = 0;
    int nranks_ = 0, rank_ = 0;
    double timeout_s_ = 30.0;
    uint64_t seq_ = 0;
    bool owner_ = false, unlinked_ = false, broken_ = false;
};

int local_comm_hook(double* buf, int n, int device_ptr, void* user)
{
    if (device_ptr || !user) return MOLA_ICP_E_BADARG;
    return static_cast<LocalComm*>(user)->allreduce(buf, n);
}

}  // namespace mola_icp_amd

using mola_icp_amd::LocalComm;

struct mola_icp_local_comm {
    LocalComm c;
};

extern "C" {

int mola_icp_local_comm_create(const char* name, int nranks, int rank, double timeout_s, mola_icp_local_comm** out)
{
    try {
        if (!out) return mola_icp_amd::fail(MOLA_ICP_E_BADARG, "null argument");
        *out = nullptr;
        mola_icp_local_comm* c = new mola_icp_local_comm();
        const int rc = c->c.init(name, nranks, rank, timeout_s);
        if (rc) {
            delete c;
            return rc;
        }
        *out = c;
        return MOLA_ICP_OK;
    } catch (const std::bad_alloc&) {
        return mola_icp_amd::fail(MOLA_ICP_E_OOM, "out of host memory");
    } catch (const std::exception& e) {
        return mola_icp_amd::fail(MOLA_ICP_E_INTERNAL, e.what());
    }
}

int mola_icp_local_comm_allreduce(mola_icp_local_comm* c, double* buf, int n)
{
    if (!c) return mola_icp_amd::fail(MOLA_ICP_E_BADARG, "null communicator");
    try {
        return c->c.allreduce(buf, n);
    } catch (const std::exception& e) {
        return mola_icp_amd::fail(MOLA_ICP_E_INTERNAL, e.what());
    }
}

int mola_icp_local_comm_nranks(mola_icp_local_comm* c, int* nranks_out)
{
    if (!c || !nranks_out) return mola_icp_amd::fail(MOLA_ICP_E_BADARG, "null argument");
    *nranks_out = c->c.nranks();
    return MOLA_ICP_OK;
}

int mola_icp_local_comm_abort(mola_icp_local_comm* c)
{
    if (!c) return mola_icp_amd::fail(MOLA_ICP_E_BADARG, "null communicator");
    c->c.abort_peers();
    return MOLA_ICP_OK;
}

int mola_icp_local_comm_destroy(mola_icp_local_comm* c)
{
    delete c;
    return MOLA_ICP_OK;
}

}  // extern "C"
