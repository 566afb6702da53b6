// RCCL (collectives over xGMI) bound at run time with dlopen, so that the library loads on a
// single-GPU box without RCCL being touched and shares the RCCL copy torch already loaded when
// the process is a torch.distributed rank.  No reference counterpart (the reference is
// single-process).
#include "common.hpp"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <dlfcn.h>
#include <mutex>
#include <string.h>

namespace padne {

// minimal mirror of the parts of rccl.h we use (ABI-stable since NCCL 2.x)
typedef struct { char internal[128]; } ncclUniqueId;
typedef void *ncclComm_t;
enum { ncclSuccess = 0 };
enum { ncclFloat32 = 7, ncclFloat64 = 8 };   // ncclDataType_t: float, double
enum { ncclSum = 0 };

struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(ncclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*CommAbort)(ncclComm_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};

static Rccl g_rccl;

// collectives issued by this process since load (all-reduce, all-gather f64, all-gather f32) and the bytes this rank
// contributed: what a multi-GPU solve costs in launches is counted, not estimated (padne_comm_call_counts)
static std::atomic<long long> g_calls[4], g_bytes[4];     // [3]: peer-to-peer halo exchanges (no collective)

static int load_rccl() {
    if (g_rccl.lib) return PADNE_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void *h = nullptr;
    for (const char *n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);   // prefer the copy the process already has (torch's)
        if (h) break;
    }
    if (!h)
        for (const char *n : names) {
            h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
    if (!h) {
        set_error("RCCL not found: %s", dlerror());
        return PADNE_E_COMM;
    }
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
    g_rccl.CommAbort = (decltype(g_rccl.CommAbort))dlsym(h, "ncclCommAbort");      // optional
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(h, "ncclAllReduce");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(h, "ncclAllGather");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.AllGather) {
        set_error("RCCL library lacks required symbols");
        return PADNE_E_COMM;
    }
    g_rccl.lib = h;
    return PADNE_OK;
}

static int check_nccl(int rc, const char *what) {
    if (rc == ncclSuccess) return PADNE_OK;
    set_error("%s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "rccl error");
    return PADNE_E_COMM;
}

// ---- in-process team: several contexts ("ranks") of ONE process on ONE device, one host thread per rank -----
// RCCL refuses two ranks on the same GPU, so the row-partitioned solver cannot be exercised with world > 1 on a
// single-GPU box through RCCL.  A team offers the same two collectives between the contexts of one process
// (host barrier + peer reads on the shared device) so that the multi-rank control flow, halo plan and block
// preconditioner run for real in the test-suite.  Results are summed in rank order: identical on every rank.
struct Team {
    int world = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    long long generation = 0;
    std::vector<const double *> ptrs;
    std::vector<double> stage;      // host staging for the reduced values
    std::vector<void *> mbox;       // every rank's mailbox ring of the peer-to-peer halo exchange (device pointers)
    bool failed = false;
};

// A rank that leaves a solve early (an error return, an exception in its driver thread) must not leave its peers
// waiting for ever: padne_team_abort marks the team failed and wakes everybody; from then on every barrier returns
// PADNE_E_COMM at once.
static int team_barrier(Team *t) {
    std::unique_lock<std::mutex> lk(t->mu);
    if (t->failed) {
        set_error("team aborted: another rank left the collective with an error");
        return PADNE_E_COMM;
    }
    const long long gen = t->generation;
    if (++t->arrived == t->world) {
        t->arrived = 0;
        ++t->generation;
        t->cv.notify_all();
    } else {
        t->cv.wait(lk, [&] { return t->generation != gen || t->failed; });
        if (t->generation == gen) {       // woken by an abort, not by the last arrival
            set_error("team aborted: another rank left the collective with an error");
            return PADNE_E_COMM;
        }
    }
    return PADNE_OK;
}

static void team_fail(Team *t) {
    {
        std::lock_guard<std::mutex> lk(t->mu);
        t->failed = true;
    }
    t->cv.notify_all();
}

// a HIP error inside a team collective takes the whole team down (the peers would wait for this rank otherwise)
#define PADNE_TEAM_HIP(t, expr)                                                            \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            team_fail(t);                                                                  \
            set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return PADNE_E_HIP;                                                            \
        }                                                                                  \
    } while (0)

static int team_allreduce(padne_ctx *ctx, double *dev_buf, int count) {
    Team *t = (Team *)ctx->team;
    PADNE_REQUIRE(count <= 16, "team all-reduce is for a handful of scalars");
    double mine[16];
    PADNE_TEAM_HIP(t, hipMemcpyAsync(mine, dev_buf, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, ctx->stream));
    PADNE_TEAM_HIP(t, hipStreamSynchronize(ctx->stream));
    {
        std::lock_guard<std::mutex> lk(t->mu);
        for (int c = 0; c < count; ++c) t->stage[(size_t)ctx->rank * 16 + c] = mine[c];
    }
    PADNE_TRY(team_barrier(t));
    double sum[16];
    for (int c = 0; c < count; ++c) {
        double s = 0.0;
        for (int r = 0; r < t->world; ++r) s += t->stage[(size_t)r * 16 + c];   // rank order: same bits everywhere
        sum[c] = s;
    }
    PADNE_TRY(team_barrier(t));      // everybody has read the stage before anyone overwrites it
    PADNE_TEAM_HIP(t, hipMemcpyAsync(dev_buf, sum, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, ctx->stream));
    PADNE_TEAM_HIP(t, hipStreamSynchronize(ctx->stream));
    return PADNE_OK;
}

static int team_allgather(padne_ctx *ctx, const void *send_v, void *recv_v, size_t bytes_per_rank) {
    const char *send = (const char *)send_v;
    char *recv = (char *)recv_v;
    Team *t = (Team *)ctx->team;
    PADNE_TEAM_HIP(t, hipStreamSynchronize(ctx->stream));          // my segment is complete
    {
        std::lock_guard<std::mutex> lk(t->mu);
        t->ptrs[(size_t)ctx->rank] = (const double *)send_v;
    }
    PADNE_TRY(team_barrier(t));
    for (int r = 0; r < t->world; ++r) {
        if (r == ctx->rank && send == recv + (size_t)r * bytes_per_rank) continue;   // in place
        PADNE_TEAM_HIP(t, hipMemcpyAsync(recv + (size_t)r * bytes_per_rank, t->ptrs[(size_t)r], bytes_per_rank,
                                         hipMemcpyDeviceToDevice, ctx->stream));
    }
    PADNE_TEAM_HIP(t, hipStreamSynchronize(ctx->stream));
    PADNE_TRY(team_barrier(t));      // peers may now overwrite their segments
    return PADNE_OK;
}

// ---- collectives through a transport of the caller (padne_ctx_comm_init_host): device -> host -> callback -> device ----
static int hostcoll_call(padne_ctx *ctx, const void *send, void *recv, size_t bytes_per_rank) {
    const int rc = ctx->hostcoll_fn(ctx->hostcoll_user, send, recv, (int64_t)bytes_per_rank);
    if (rc != 0) {
        set_error("the caller's all-gather failed (%d): a rank has left the collective", rc);
        return PADNE_E_COMM;
    }
    return PADNE_OK;
}

static int hostcoll_allreduce(padne_ctx *ctx, double *dev_buf, int count) {
    PADNE_REQUIRE(count <= 16, "all-reduce is for a handful of scalars");
    std::vector<double> all((size_t)ctx->world * 16, 0.0);
    double mine[16];
    PADNE_HIP_CHECK(hipMemcpyAsync(mine, dev_buf, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, ctx->stream));
    PADNE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    PADNE_TRY(hostcoll_call(ctx, mine, all.data(), sizeof(double) * (size_t)count));
    double sum[16];
    for (int c = 0; c < count; ++c) {
        double t = 0.0;
        for (int r = 0; r < ctx->world; ++r) t += all[(size_t)r * count + c];      // rank order: same bits everywhere
        sum[c] = t;
    }
    PADNE_HIP_CHECK(hipMemcpyAsync(dev_buf, sum, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, ctx->stream));
    PADNE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PADNE_OK;
}

static int hostcoll_allgather(padne_ctx *ctx, const void *send, void *recv, size_t bytes_per_rank, hipStream_t stream) {
    std::vector<char> mine(bytes_per_rank), all(bytes_per_rank * (size_t)ctx->world);
    PADNE_HIP_CHECK(hipMemcpyAsync(mine.data(), send, bytes_per_rank, hipMemcpyDeviceToHost, stream));
    PADNE_HIP_CHECK(hipStreamSynchronize(stream));
    PADNE_TRY(hostcoll_call(ctx, mine.data(), all.data(), bytes_per_rank));
    PADNE_HIP_CHECK(hipMemcpyAsync(recv, all.data(), all.size(), hipMemcpyHostToDevice, stream));
    PADNE_HIP_CHECK(hipStreamSynchronize(stream));
    return PADNE_OK;
}

int comm_allreduce_sum_f64(padne_ctx *ctx, double *dev_buf, int count) {
    if (comm_active(ctx)) {
        ++g_calls[0];
        g_bytes[0] += 8LL * count;
    }
    // (a communicator of one rank: the sum over the ranks is the value itself.  Not a call into the library for it: on a
    // one-rank communicator every ncclAllReduce / ncclAllGather allocates and frees a small buffer -- hipExtMallocWithFlags on
    // the calling thread, hipFree on a helper thread, 780 of each in ten solves of the bench (profiles/r06_hip_api_dist.csv) --
    // and a hipFree waits for the device while it holds the runtime's lock: the launches of the solve queued behind it.)
    if (ctx->world == 1) return PADNE_OK;
    if (ctx->team != nullptr) return team_allreduce(ctx, dev_buf, count);
    if (ctx->hostcoll_fn != nullptr) return hostcoll_allreduce(ctx, dev_buf, count);
    if (ctx->comm == nullptr) return PADNE_OK;
    return check_nccl(g_rccl.AllReduce(dev_buf, dev_buf, (size_t)count, ncclFloat64, ncclSum, (ncclComm_t)ctx->comm,
                                       ctx->stream), "ncclAllReduce");
}

static int allgather_impl(padne_ctx *ctx, const void *send, void *recv, int count_per_rank, bool f64, hipStream_t stream) {
    if (count_per_rank == 0) return PADNE_OK;
    const int kind = f64 ? 1 : 2;
    const size_t elem = f64 ? sizeof(double) : sizeof(float);
    if (comm_active(ctx)) {
        ++g_calls[kind];
        g_bytes[kind] += (long long)elem * count_per_rank;
    }
    if (ctx->world == 1) {      // one rank: what is gathered is what was sent (see comm_allreduce_sum_f64)
        if (send != recv)
            PADNE_HIP_CHECK(hipMemcpyAsync(recv, send, elem * (size_t)count_per_rank, hipMemcpyDeviceToDevice, stream));
        return PADNE_OK;
    }
    if (ctx->team != nullptr) return team_allgather(ctx, send, recv, elem * (size_t)count_per_rank);
    if (ctx->hostcoll_fn != nullptr) return hostcoll_allgather(ctx, send, recv, elem * (size_t)count_per_rank, stream);
    if (ctx->comm == nullptr) return PADNE_OK;
    return check_nccl(g_rccl.AllGather(send, recv, (size_t)count_per_rank, f64 ? ncclFloat64 : ncclFloat32, (ncclComm_t)ctx->comm,
                                       stream), "ncclAllGather");
}

int comm_allgather_f64(padne_ctx *ctx, const double *send, double *recv, int count_per_rank) {
    return allgather_impl(ctx, send, recv, count_per_rank, true, ctx->stream);
}

int comm_allgather_f32(padne_ctx *ctx, const float *send, float *recv, int count_per_rank) {
    return allgather_impl(ctx, send, recv, count_per_rank, false, ctx->stream);
}

// ---- peer-to-peer halo exchange ------------------------------------------------------------------------------------
// In-process team: the ranks are contexts of one process on one device, so a peer's mailbox is an ordinary device
// pointer, registered with the team; "the stores have landed" = every rank has drained its stream (host barrier).  One
// barrier per exchange instead of the two barriers and world peer copies of the team all-gather, and nothing of it is a
// collective.  The ring makes the second barrier unnecessary: entry e is rewritten kP2pRing exchanges later, and every
// exchange in between ends with all ranks' streams drained.
// One process per GPU (RCCL, or a host transport): the mailboxes are uncached device memory shared through hipIpc handles
// (padne_ctx_p2p_export / _import below) and the arrival is signalled by flags the receiver's unpack kernel waits for
// (pcg.hip: halo_store_peers_kernel / halo_unpack_kernel with their IPC arguments) -- no barrier, no host.  The ring makes
// reuse safe there too: a rank stores exchange e + 1 only after its own unpack of exchange e (stream order), so when a
// sender has completed its receive of e + 1 every rank has consumed e, and entry e % kP2pRing is rewritten at e + kP2pRing.
// Tested with two PROCESSES on one GPU (tests/test_two_processes_gpu.py); between two GPUs the same stores cross xGMI.
bool comm_p2p_enabled(const padne_ctx *ctx) {
    return (ctx->team != nullptr || ctx->p2p_ipc) && !ctx->opt.no_p2p;
}

// the in-process team regrows its rings on demand; mailboxes shared between processes have the size they were exported
// with, a plan with more slots per rank keeps the all-gather (m is the same number on every rank: all take the same path)
bool comm_p2p_fits(const padne_ctx *ctx, int m) {
    return ctx->team != nullptr || (ctx->p2p_ipc && m <= ctx->p2p_m_cap);
}

// does a halo exchange of this context run beside what is queued between its two halves?  (Peer-to-peer stores: the arrival
// is all that is left for the second half.)  Where it does not -- an all-gather on the same stream -- splitting a product
// into interior and boundary tiles buys nothing and costs a launch: the split plans are built only where this holds
bool comm_exchange_overlaps(const padne_ctx *ctx) { return comm_p2p_enabled(ctx); }

void comm_p2p_release(padne_ctx *ctx) {
    for (void *p : ctx->p2p_ipc_mapped)
        if (p != nullptr) (void)hipIpcCloseMemHandle(p);
    ctx->p2p_ipc_mapped.clear();
    ctx->p2p_ipc = false;
    if (ctx->p2p_mbox != nullptr) (void)hipFree(ctx->p2p_mbox);
    if (ctx->p2p_peers != nullptr) (void)hipFree(ctx->p2p_peers);
    ctx->p2p_mbox = nullptr;
    ctx->p2p_peers = nullptr;
    ctx->p2p_m_cap = 0;
}

// a receiver of this context gave up waiting for a sender's flag (the sender died, or never got to its stores): the
// values it unpacked are not the exchange's, the solve that contains it must fail
int comm_p2p_check(padne_ctx *ctx) {
    if (!ctx->p2p_ipc || ctx->p2p_mbox == nullptr) return PADNE_OK;
    unsigned long long err = 0;
    PADNE_TRY(read_back(ctx, (const char *)ctx->p2p_mbox + kP2pErrorOff, sizeof(err), &err));
    if (err != 0) {
        set_error("peer-to-peer halo exchange: rank %d waited more than %u ms for the stores of rank %llu (exchange %llu)",
                  ctx->rank, ctx->p2p_timeout_ms, (err >> 48) - 1, err & 0xffffffffffffull);
        // reported once: a peer that was slow rather than dead must not fail every later solve of this context with the
        // stale rank and exchange number
        (void)hipMemsetAsync((char *)ctx->p2p_mbox + kP2pErrorOff, 0, 8, ctx->stream);
        (void)hipStreamSynchronize(ctx->stream);
        return PADNE_E_COMM;
    }
    return PADNE_OK;
}

int comm_p2p_begin(padne_ctx *ctx, int m, void ***peers_dev, size_t *entry_offset) {
    if (ctx->p2p_ipc) {
        PADNE_REQUIRE(m > 0 && m <= ctx->p2p_m_cap, "peer-to-peer exchange larger than the shared mailboxes");
        const unsigned long long seq = ctx->p2p_seq++;
        *peers_dev = ctx->p2p_peers;
        *entry_offset = (size_t)kP2pHeaderBytes + (size_t)(seq % kP2pRing) * (size_t)ctx->world * (size_t)ctx->p2p_m_cap * 8;
        ++g_calls[3];
        g_bytes[3] += 8LL * m * (ctx->world - 1);
        return PADNE_OK;
    }
    Team *t = (Team *)ctx->team;
    PADNE_REQUIRE(t != nullptr && m > 0, "peer-to-peer exchange without a team");
    if (m > ctx->p2p_m_cap) {
        // first exchange, or a plan with more slots than the ring holds: (re)build the mailboxes -- all ranks get here
        // together, they run the same sequence of exchanges with the same plans
        PADNE_TEAM_HIP(t, hipStreamSynchronize(ctx->stream));
        PADNE_TRY(team_barrier(t));                  // nobody stores into a ring that is about to go
        comm_p2p_release(ctx);
        const int cap = std::max(2 * m, 4096);
        const size_t bytes = (size_t)kP2pRing * (size_t)ctx->world * (size_t)cap * 8;
        PADNE_TEAM_HIP(t, hipMalloc(&ctx->p2p_mbox, bytes));
        PADNE_TEAM_HIP(t, hipMalloc((void **)&ctx->p2p_peers, sizeof(void *) * (size_t)ctx->world));
        PADNE_TEAM_HIP(t, hipMemsetAsync(ctx->p2p_mbox, 0, bytes, ctx->stream));
        {
            std::lock_guard<std::mutex> lk(t->mu);
            t->mbox[(size_t)ctx->rank] = ctx->p2p_mbox;
        }
        PADNE_TRY(team_barrier(t));
        std::vector<void *> peers;
        {
            std::lock_guard<std::mutex> lk(t->mu);
            peers = t->mbox;
        }
        PADNE_TEAM_HIP(t, hipMemcpyAsync(ctx->p2p_peers, peers.data(), sizeof(void *) * (size_t)ctx->world, hipMemcpyHostToDevice,
                                         ctx->stream));
        PADNE_TEAM_HIP(t, hipStreamSynchronize(ctx->stream));
        PADNE_TRY(team_barrier(t));                  // every ring is zeroed and every table complete before the first store
        ctx->p2p_m_cap = cap;
        ctx->p2p_seq = 0;
    }
    const unsigned long long seq = ctx->p2p_seq++;
    *peers_dev = ctx->p2p_peers;
    *entry_offset = (size_t)(seq % kP2pRing) * (size_t)ctx->world * (size_t)ctx->p2p_m_cap * 8;
    ++g_calls[3];
    g_bytes[3] += 8LL * m * (ctx->world - 1);
    return PADNE_OK;
}

int comm_p2p_arrive(padne_ctx *ctx) {
    if (ctx->p2p_ipc) return PADNE_OK;      // the unpack kernel waits for the senders' flags itself
    Team *t = (Team *)ctx->team;
    PADNE_REQUIRE(t != nullptr, "peer-to-peer exchange without a team");
    PADNE_TEAM_HIP(t, hipStreamSynchronize(ctx->stream));      // my stores are out ...
    return team_barrier(t);                                   // ... and so are everybody else's
}

// a rank that fails locally (allocation, HIP error, argument check) inside a row-partitioned solve tells its team
// (in-process team: the peers' barriers return PADNE_E_COMM at once).  Over RCCL a rank cannot reach into its peers: it
// aborts ITS communicator -- queued collectives are torn down instead of waiting for ever, the context is left without one,
// and the error return makes the Python driver raise, so the process ends non-zero and the launcher (torchrun) ends the
// other ranks, which are blocked in a collective this rank will never enter.
void comm_abort(padne_ctx *ctx) {
    if (ctx->team != nullptr) team_fail((Team *)ctx->team);
    if (ctx->comm != nullptr && g_rccl.CommAbort != nullptr) {
        (void)g_rccl.CommAbort((ncclComm_t)ctx->comm);
        ctx->comm = nullptr;
    }
}

void comm_destroy(padne_ctx *ctx) {
    comm_p2p_release(ctx);
    if (ctx->comm && g_rccl.CommDestroy) g_rccl.CommDestroy((ncclComm_t)ctx->comm);
    ctx->comm = nullptr;
    ctx->team = nullptr;
    ctx->hostcoll_fn = nullptr;
    ctx->hostcoll_user = nullptr;
}

}  // namespace padne

using namespace padne;

extern "C" int padne_comm_unique_id(void *id128) {
    PADNE_REQUIRE(id128, "id128");
    PADNE_TRY(load_rccl());
    ncclUniqueId id;
    PADNE_TRY(check_nccl(g_rccl.GetUniqueId(&id), "ncclGetUniqueId"));
    memcpy(id128, &id, sizeof(id));
    return PADNE_OK;
}

extern "C" int padne_ctx_comm_init(padne_ctx *ctx, const void *id128, int rank, int world_size) {
    PADNE_REQUIRE(ctx && id128, "null argument");
    PADNE_REQUIRE(world_size >= 1 && rank >= 0 && rank < world_size, "rank/world_size");
    PADNE_REQUIRE(!comm_active(ctx), "communicator already initialised");
    PADNE_TRY(load_rccl());
    PADNE_HIP_CHECK(hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t comm = nullptr;
    PADNE_TRY(check_nccl(g_rccl.CommInitRank(&comm, world_size, id, rank), "ncclCommInitRank"));
    ctx->comm = comm;
    ctx->rank = rank;
    ctx->world = world_size;
    return PADNE_OK;
}

extern "C" int padne_ctx_comm_init_host(padne_ctx *ctx, int rank, int world_size, padne_allgather_fn allgather, void *user) {
    PADNE_REQUIRE(ctx && allgather, "null argument");
    PADNE_REQUIRE(world_size >= 1 && world_size <= kP2pMaxWorld && rank >= 0 && rank < world_size, "rank/world_size");
    PADNE_REQUIRE(!comm_active(ctx), "context already has a communicator");
    ctx->hostcoll_fn = allgather;
    ctx->hostcoll_user = user;
    ctx->rank = rank;
    ctx->world = world_size;
    return PADNE_OK;
}

// ---- mailboxes shared between processes (hipIpc) -------------------------------------------------------------------
extern "C" int padne_ctx_p2p_export(padne_ctx *ctx, int32_t slots_per_rank, void *handle64) {
    PADNE_REQUIRE(ctx && handle64, "null argument");
    PADNE_REQUIRE((ctx->comm != nullptr || ctx->hostcoll_fn != nullptr) && ctx->team == nullptr,
                  "mailboxes are shared between the processes of a communicator");
    PADNE_REQUIRE(slots_per_rank > 0 && ctx->world <= kP2pMaxWorld, "slots_per_rank / world size");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the handle travels as 64 bytes");
    PADNE_HIP_CHECK(hipSetDevice(ctx->device));
    PADNE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    comm_p2p_release(ctx);
    const size_t bytes = (size_t)kP2pHeaderBytes + (size_t)kP2pRing * (size_t)ctx->world * (size_t)slots_per_rank * 8;
    // uncached (MTYPE_UC) device memory: a peer's stores go to memory and this rank's polls read memory -- a line of
    // ordinary (coarse-grained) device memory may sit in an XCD's L2 for the whole kernel that polls it.  It is what
    // RCCL allocates for its own flags and buffers; fine-grained memory is the second choice, ordinary memory is refused.
    void *p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error("no uncached / fine-grained device memory for the mailbox: %s", hipGetErrorString(e));
        return PADNE_E_COMM;
    }
    ctx->p2p_mbox = p;
    e = hipMemsetAsync(p, 0, bytes, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    hipIpcMemHandle_t h;
    if (e == hipSuccess) e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        comm_p2p_release(ctx);
        set_error("the mailbox cannot be shared (hipIpcGetMemHandle: %s)", hipGetErrorString(e));
        return PADNE_E_COMM;
    }
    memcpy(handle64, &h, 64);
    ctx->p2p_m_cap = slots_per_rank;
    ctx->p2p_seq = 0;
    ctx->p2p_timeout_ms = ctx->opt.p2p_timeout_ms;
    return PADNE_OK;
}

extern "C" int padne_ctx_p2p_import(padne_ctx *ctx, const void *handles, int32_t n_handles) {
    PADNE_REQUIRE(ctx && handles, "null argument");
    PADNE_REQUIRE(ctx->p2p_mbox != nullptr && ctx->p2p_m_cap > 0 && !ctx->p2p_ipc, "padne_ctx_p2p_export comes first");
    PADNE_REQUIRE(n_handles == ctx->world, "one handle per rank");
    PADNE_HIP_CHECK(hipSetDevice(ctx->device));
    std::vector<void *> peers((size_t)ctx->world, nullptr);
    for (int r = 0; r < ctx->world; ++r) {
        if (r == ctx->rank) {
            peers[(size_t)r] = ctx->p2p_mbox;
            continue;
        }
        hipIpcMemHandle_t h;
        memcpy(&h, (const char *)handles + (size_t)r * 64, 64);
        void *p = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            for (void *q : ctx->p2p_ipc_mapped) (void)hipIpcCloseMemHandle(q);
            ctx->p2p_ipc_mapped.clear();
            set_error("the mailbox of rank %d cannot be mapped (hipIpcOpenMemHandle: %s)", r, hipGetErrorString(e));
            return PADNE_E_COMM;
        }
        ctx->p2p_ipc_mapped.push_back(p);
        peers[(size_t)r] = p;
    }
    if (ctx->p2p_peers == nullptr) PADNE_HIP_CHECK(hipMalloc((void **)&ctx->p2p_peers, sizeof(void *) * (size_t)ctx->world));
    PADNE_HIP_CHECK(hipMemcpyAsync(ctx->p2p_peers, peers.data(), sizeof(void *) * (size_t)ctx->world, hipMemcpyHostToDevice, ctx->stream));
    PADNE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ctx->p2p_ipc = true;
    return PADNE_OK;
}

// Real exchanges through the shared mailboxes -- the kernels, flags and ring entries a solve uses -- with values every rank
// can check: nine of them (twice around the ring and once more), as WIDE as the mailboxes were exported (every cell of every
// ring entry is written and read), doubles and floats in turn; in exchange t rank r exports f(r, t, k) for its k-th cell.
// A node on which the peers' stores or flags do not arrive (peer access that maps but does not deliver, a runtime that
// ignores the memory type, a link that reorders the flag in front of the data) shows here, within 2 s per exchange and
// before any solve depends on it: the caller gathers the verdicts and keeps the all-gather unless every rank says yes.
extern "C" int padne_ctx_p2p_selftest(padne_ctx *ctx, int32_t *ok_out) {
    PADNE_REQUIRE(ctx && ok_out, "null argument");
    *ok_out = 0;
    PADNE_REQUIRE(ctx->p2p_ipc, "padne_ctx_p2p_import comes first");
    const int m = ctx->p2p_m_cap;
    constexpr int kRounds = 2 * kP2pRing + 1;
    PADNE_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t n = (size_t)m + (size_t)ctx->world * (size_t)m;
    // (integers below 2^24 up to 126 ranks: exact in single precision too; every round has values of its own -- rounds t and
    // t + kP2pRing use the same ring entry, and stale data of the earlier one must not pass for the later one's)
    static_assert(kRounds <= 16, "a round's values lie 8192 apart inside a rank's 131072");
    auto value = [](int r, int t, int k) { return 131072.0 * (r + 1) + 8192.0 * t + (double)(k % 8192); };
    std::vector<double> h(n, -1.0);
    std::vector<float> hf(n, -1.f);
    std::vector<int> idx((size_t)m);
    for (int k = 0; k < m; ++k) idx[(size_t)k] = k;
    double *d_v = (double *)pool_alloc(ctx, sizeof(double) * n);
    int *d_idx = (int *)pool_alloc(ctx, sizeof(int) * (size_t)m);
    if (d_v == nullptr || d_idx == nullptr) {
        pool_free(ctx, d_v);
        pool_free(ctx, d_idx);
        return PADNE_E_NOMEM;
    }
    int rc = PADNE_OK;
    bool ok = comm_p2p_enabled(ctx);                         // (PADNE_NO_P2P: the exchanges below are all-gathers; nothing is tested)
    hipError_t e = hipMemcpyAsync(d_idx, idx.data(), sizeof(int) * (size_t)m, hipMemcpyHostToDevice, ctx->stream);
    const unsigned keep_timeout = ctx->p2p_timeout_ms;
    ctx->p2p_timeout_ms = 2000;
    HaloPlan plan;
    plan.n_owned = m;
    plan.m = m;
    plan.n_export = m;
    plan.export_idx = d_idx;
    // (a rank that has seen a wrong value goes through the remaining exchanges all the same: its peers would otherwise sit
    // out the time limit of every round it skipped)
    const bool run = ok;
    for (int t = 0; t < kRounds && e == hipSuccess && rc == PADNE_OK && run; ++t) {
        const bool f64 = (t & 1) == 0;
        if (f64) {
            std::fill(h.begin(), h.end(), -1.0);
            for (int k = 0; k < m; ++k) h[(size_t)k] = value(ctx->rank, t, k);
            e = hipMemcpyAsync(d_v, h.data(), sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) rc = halo_exchange_plan(ctx, plan, d_v, nullptr);
            if (e == hipSuccess && rc == PADNE_OK) e = hipMemcpyAsync(h.data(), d_v, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream);
        } else {
            std::fill(hf.begin(), hf.end(), -1.f);
            for (int k = 0; k < m; ++k) hf[(size_t)k] = (float)value(ctx->rank, t, k);
            e = hipMemcpyAsync(d_v, hf.data(), sizeof(float) * n, hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) rc = halo_exchange_plan_f32(ctx, plan, (float *)d_v, nullptr);
            if (e == hipSuccess && rc == PADNE_OK) e = hipMemcpyAsync(hf.data(), d_v, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream);
        }
        if (e == hipSuccess && rc == PADNE_OK) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess || rc != PADNE_OK) break;
        for (int r = 0; r < ctx->world && ok; ++r)
            for (int k = 0; k < m && ok; ++k) {
                const size_t at = (size_t)m + (size_t)r * (size_t)m + (size_t)k;
                ok = f64 ? h[at] == value(r, t, k) : hf[at] == (float)value(r, t, k);
            }
        if (comm_p2p_check(ctx) != PADNE_OK) ok = false;     // a wait ran out (the word is cleared with the report): here, not in the first solve
    }
    ctx->p2p_timeout_ms = keep_timeout;
    pool_free(ctx, d_v);
    pool_free(ctx, d_idx);
    if (e != hipSuccess) {
        set_error("mailbox self-test failed: %s", hipGetErrorString(e));
        return PADNE_E_HIP;
    }
    PADNE_TRY(rc);
    *ok_out = ok || !comm_p2p_enabled(ctx) ? 1 : 0;
    return PADNE_OK;
}

extern "C" int padne_ctx_p2p_close(padne_ctx *ctx) {
    PADNE_REQUIRE(ctx, "ctx");
    if (ctx->team != nullptr) return PADNE_OK;      // (the team's rings belong to the team path)
    PADNE_HIP_CHECK(hipSetDevice(ctx->device));
    PADNE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    comm_p2p_release(ctx);
    return PADNE_OK;
}

extern "C" int padne_comm_call_counts(long long calls[4], long long bytes[4]) {
    PADNE_REQUIRE(calls && bytes, "null argument");
    for (int k = 0; k < 4; ++k) {
        calls[k] = g_calls[k].load();
        bytes[k] = g_bytes[k].load();
    }
    return PADNE_OK;
}

extern "C" int padne_ctx_comm_rank(padne_ctx *ctx, int *rank, int *world_size) {
    PADNE_REQUIRE(ctx, "ctx");
    if (rank) *rank = ctx->rank;
    if (world_size) *world_size = ctx->world;
    return PADNE_OK;
}

// ---- in-process team (tests / single-GPU rehearsal of the multi-rank path) --------------------------------------
extern "C" int padne_team_create(int world_size, void **team_out) {
    PADNE_REQUIRE(team_out && world_size >= 1 && world_size <= 64, "team size");
    Team *t = new Team();
    t->world = world_size;
    t->ptrs.assign((size_t)world_size, nullptr);
    t->stage.assign((size_t)world_size * 16, 0.0);
    t->mbox.assign((size_t)world_size, nullptr);
    *team_out = t;
    return PADNE_OK;
}

extern "C" int padne_team_abort(void *team) {
    PADNE_REQUIRE(team != nullptr, "team");
    team_fail((Team *)team);
    return PADNE_OK;
}

extern "C" int padne_team_destroy(void *team) {
    delete (Team *)team;
    return PADNE_OK;
}

extern "C" int padne_ctx_join_team(padne_ctx *ctx, void *team, int rank) {
    PADNE_REQUIRE(ctx && team, "null argument");
    Team *t = (Team *)team;
    PADNE_REQUIRE(rank >= 0 && rank < t->world, "rank");
    PADNE_REQUIRE(!comm_active(ctx), "context already has a communicator");
    ctx->team = t;
    ctx->rank = rank;
    ctx->world = t->world;
    return PADNE_OK;
}
