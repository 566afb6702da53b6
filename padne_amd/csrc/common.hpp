// Shared declarations for libpadne_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/padne_hip.h"
#include "../../include/padne_hip_test.h"

#include <atomic>

namespace padne {

// Kernels and asynchronous fills this library has queued since it was loaded (padne_launch_count, test header): what a
// solve costs in launches is read off this counter -- the floor of a small system is launches, not bytes.
extern std::atomic<long long> g_launch_count;

void set_error(const char *fmt, ...);

#define PADNE_HIP_CHECK(expr)                                                              \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            ::padne::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),      \
                               __FILE__, __LINE__);                                        \
            return PADNE_E_HIP;                                                            \
        }                                                                                  \
    } while (0)

#define PADNE_REQUIRE(cond, msg)                                                           \
    do {                                                                                   \
        if (!(cond)) {                                                                     \
            ::padne::set_error("invalid argument: %s (%s)", msg, #cond);                   \
            return PADNE_E_INVALID;                                                        \
        }                                                                                  \
    } while (0)

#define PADNE_TRY(expr)                                                                    \
    do {                                                                                   \
        int _rc = (expr);                                                                  \
        if (_rc != PADNE_OK) return _rc;                                                   \
    } while (0)

}  // namespace padne
// Every kernel launch of the library goes through hipLaunchKernelGGL (no <<< >>> in the sources) and is counted here
// (padne_launch_count: the launches per step / setup / iteration of the bench line).  The macro is restated on the
// public launch syntax -- not on a private macro of one HIP release.  Counted: kernels and hipMemsetAsync (a fill
// kernel).  NOT counted: hipMemcpyAsync -- the device-to-device copies a setup queues (row pointers of a result copied
// out of a scan, a blit kernel each in a trace; the one-GPU CG iteration queues none) and the small copies to and
// from the host.  A trace therefore shows a few more dispatches than the count: 510 against 494 per C4 setup.
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...)   \
    do {                                                                                   \
        ::padne::g_launch_count.fetch_add(1, std::memory_order_relaxed);                   \
        kernelName<<<(numBlocks), (numThreads), (memPerBlock), (streamId)>>>(__VA_ARGS__); \
    } while (0)
#define hipMemsetAsync(...) (::padne::g_launch_count.fetch_add(1, std::memory_order_relaxed), hipMemsetAsync(__VA_ARGS__))
namespace padne {

constexpr int kNumXcd = 8;           // MI355X: 8 XCDs, each with a private L2

// SpMV tile geometry (see spmv.hip)
constexpr int kSpmvThreads = 256;
constexpr int kSpmvRows = 256;       // rows per workgroup turn: four wave-private tiles of 64 rows
constexpr int kPadNnz = 4096;        // cols/vals allocations are padded by this many zero entries

// number of partial sums every reduction kernel emits (one per workgroup)
constexpr int kMaxPartials = 2048;
// pinned polling buffer of a context: [0, 4096) status mirrors of the CG loops, then kPinnedSlots slots of 512 bytes for
// the Lanczos histories queued on the second stream (one per possible level of a hierarchy, kMaxLevels = 16 in amg.hip)
constexpr int kPinnedSlots = 16;
constexpr int kPinnedSlotBase = 4096;
constexpr int kPinnedBytes = kPinnedSlotBase + 512 * kPinnedSlots;

constexpr int kTinySystem = 32;      // systems up to this size are solved with the diagonal preconditioner (no hierarchy is built)
constexpr int kSpmmK = 8;            // right-hand sides of the batched path (spmm.hip), interleaved [n][8]

}  // namespace padne

struct padne_csr {
    int64_t n_rows = 0, n_cols = 0, nnz = 0;
    int32_t *rowptr = nullptr;   // [n_rows + 1]
    int32_t *cols = nullptr;     // [nnz + pad]   padding entries are column 0
    double *vals = nullptr;      // [nnz + pad]   padding entries are 0.0
    double *dinv = nullptr;      // [n_rows] 1/diag, built on first use (square matrices)
    int device = 0;
    padne_ctx *owner = nullptr;  // context whose pool the arrays came from (must outlive the matrix)
    void *amg = nullptr;         // cached multigrid hierarchy (padne::Amg*), owned
    float *vals32 = nullptr, *dinv32 = nullptr;   // single-precision copies for the multigrid cycle (csr_build_f32)
    // x-window plan of the SpMV (csr_build_xw_plan): per 64-row tile up to three runs of x that cover all its columns
    int4 *xw_desc = nullptr;             // [n_tiles] run starts in .x .y .z, .w = 1 if the tile qualifies
    unsigned short *xw_lidx = nullptr;   // [nnz + pad] position of every column inside its tile's staged runs (bytes when xw_run == 72)
    int xw_state = 0;                    // 0 = not examined, 1 = in use, -1 = examined and not worth it
    int xw_run = 0;                      // entries per staged run (72 for scan-line meshes, 128 for strip-ordered ones)
    int xw_nruns = 3;                    // staged runs per tile: 3, or 12 short ones of 20 (the wide plan of a fused up-leg operator W,
                                         // csr_build_xw_plan_wide: xw_desc then holds four int4 per tile -- twelve run starts, spare, flag)
    // interior / boundary split of a row-partitioned operator (csr_build_split_plan): the 64-row tiles whose columns are all
    // owned, and the tiles that read an exchange slot.  The product of the interior tiles needs no remote value and is
    // launched while the halo exchange is under way; the boundary tiles follow once it has landed.
    int *split_tiles = nullptr;          // [n_tiles]: interior tiles first, then the boundary tiles
    int split_n_int = 0, split_n_bnd = 0;
    int split_state = 0;                 // 0 = not examined, 1 = in use, -1 = not a row-partitioned operator / switched off
    bool hierarchy_operator = false;   // multigrid-internal operator: may use the wave-per-row SpMV
    bool cols_unsorted = false;        // an uploaded matrix (padne_csr_from_host) with a row whose columns do not ascend: paths
                                       // that count on column order (the order-preserving relabel) leave it to the general ones
    padne_csr *prec_block = nullptr;   // borrowed: owned x owned diagonal block for the preconditioner
    // the mesh the system was assembled from stays on the device with it (padne_assemble_system), so that the
    // post-processing of the solution (padne_csr_power_density) does not upload 40 bytes per vertex again
    double *mesh_xy = nullptr, *mesh_sigma = nullptr;
    int32_t *mesh_tri = nullptr;
    long long *mesh_voff = nullptr, *mesh_toff = nullptr;
    int64_t mesh_n_vert = 0, mesh_n_tri = 0, mesh_n_mesh = 0;
};

// The PADNE_* environment switches, read ONCE when a context is created (padne_ctx_reload_options of the test header reads
// them again): what a context does never depends on a getenv in the middle of a call.  INTEGRATION.md lists them.
struct padne_options {
    // alternatives of the product path that tests compare against the default
    bool amg_f64 = false;              // PADNE_AMG_F64=1: the multigrid cycle in double precision
    int amg_w = 2;                     // PADNE_AMG_W=none|fine: fused up-leg operator W on no level / the fine level only (default: all)
    bool amg_exchange_all = false;     // PADNE_AMG_EXCHANGE_ALL=1: the last partitioned level exchanges instead of computing from the tail
    bool pcg_p64 = false;              // PADNE_PCG_P64=1: the search direction of the loop stays in double precision
    bool pcg_no_xhist = false;         // PADNE_PCG_NO_XHIST=1: x updated in every iteration instead of from the kept search directions
    bool gj_vector = false;            // PADNE_GJ_VECTOR=1: the dense inverse by the vector kernel (16 pivots per launch)
    bool no_batch = false;             // PADNE_NO_BATCH=1: right-hand sides one at a time
    bool no_mailbox = false;           // PADNE_NO_MAILBOX=1: host looks by copy + synchronise
    bool no_p2p = false;               // PADNE_NO_P2P=1: halo exchanges as all-gathers
    bool no_split = false;             // PADNE_NO_SPLIT=1: products behind an exchange in one launch
    bool no_xwindow = false;           // PADNE_NO_XWINDOW=1: no x-window plans (SpMV and the setup kernels that use them)
    bool setup_one_stream = false;     // PADNE_SETUP_ONE_STREAM=1: the side work of the multigrid setup on the main stream (traces with standalone kernel times)
    int cg_single_reduction = -1;      // PADNE_CG_SINGLE_REDUCTION=0|1: force the loop form (default: by communicator)
    int lockstep_narrow = -1;          // PADNE_LOCKSTEP_NARROW=0|2: never / always the narrow lockstep widths
    // sizes
    int amg_coarse_n = 2048;           // PADNE_AMG_COARSE_N: coarsest-level size the dense inverse takes
    long long amg_gather_n = 0;        // PADNE_AMG_GATHER_N: level size below which a partitioned hierarchy is gathered (0: default)
    unsigned p2p_timeout_ms = 20000;   // PADNE_P2P_TIMEOUT_MS
    // PADNE_FORCE=<path>[,<path>...]: send everything through a path that the data takes only rarely (tests)
    bool force_asm_hash = false, force_asm_two_pass = false, force_relabel_slots = false, force_transpose_cursors = false;
    bool force_relabel_lanes = false;  // relabel_lanes: the order-preserving relabel with a lane per row (the form of round 5)
    bool force_xhist_small = false;    // xhist_small: eight places for the kept search directions (the ring wraps within a solve)
    long long force_spgemm_split = 0;  // spgemm_split:<slots>
    // PADNE_VERBOSE=amg,xw,pool: diagnostics on stderr
    bool verbose_amg = false, verbose_xw = false, verbose_pool = false;
};
namespace padne { void options_from_env(padne_options *o); }

struct padne_ctx {
    int device = 0;
    padne_options opt;
    hipStream_t stream = nullptr;
    // reduction scratch
    double *partials = nullptr;      // [8][kMaxPartials]
    double *scalars = nullptr;       // device scalars used by the PCG loop
    int32_t *status = nullptr;       // device status words
    void *pinned = nullptr;          // small pinned host buffer for polling
    // grow-on-demand workspace for PCG vectors
    void *ws = nullptr;
    size_t ws_bytes = 0;
    // RCCL
    void *comm = nullptr;
    void *team = nullptr;            // in-process team of contexts (single-GPU rehearsal of the multi-rank path)
    int rank = 0, world = 1;
    // halo plan of a row-partitioned matrix: vectors are [owned | world * halo_m exchanged values]
    bool halo_on = false;
    long long halo_n_owned = 0;
    int halo_m = 0, halo_n_export = 0;
    int32_t *halo_export = nullptr;
    // peer-to-peer halo exchange (comm.hip, comm_p2p_*): this rank's mailbox ring, the device table of all ranks' rings
    void *p2p_mbox = nullptr;
    void **p2p_peers = nullptr;      // device array [world]
    int p2p_m_cap = 0;               // exchange slots per rank a ring entry holds (8 bytes each)
    unsigned long long p2p_seq = 0;  // exchanges so far: ring entry = seq % kP2pRing
    // the same between processes (padne_ctx_p2p_export / _import): the mailbox is uncached device memory shared through
    // hipIpc, [kP2pHeaderBytes of arrival flags | ring]; p2p_ipc is set once every peer's mailbox is mapped
    bool p2p_ipc = false;
    std::vector<void *> p2p_ipc_mapped;   // what hipIpcOpenMemHandle returned for the other ranks (closed with the mailbox)
    unsigned p2p_timeout_ms = 20000;
    // collectives through a transport of the caller (padne_ctx_comm_init_host)
    int (*hostcoll_fn)(void *, const void *, void *, int64_t) = nullptr;
    void *hostcoll_user = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // second stream of the context (its own pool, workspace and reduction scratch): independent chains of short,
    // latency-bound kernels of the multigrid setup run there next to the main chain (aux_context, stream_order)
    // the streams host vectors of a solve_system plan travel on (kkt.hip: parallel_copy), made on first use and kept: a stream
    // costs 2 ms to create on a fresh process, four of them per plan were 8 of the 11 ms of a cold solve_system at 10 k unknowns
    hipStream_t copy_stream[4] = {nullptr, nullptr, nullptr, nullptr};
    padne_ctx *aux = nullptr, *parent = nullptr;
    bool is_aux = false;
    unsigned long long lockstep_groups = 0;      // groups of right-hand sides this context has solved in lockstep (test introspection)
    unsigned pinned_busy = 0;        // bit k: the k-th 512-byte Lanczos slot of `pinned` holds a job in flight (second stream only)
    // mailbox: a page of host-coherent memory the device posts small results into (read_back / mail_ticket)
    unsigned long long *mailbox = nullptr, *mailbox_dev = nullptr;
    unsigned long long mail_seq = 0;
    hipEvent_t ev_order = nullptr;
    // caching device allocator (see pool_alloc): hipMalloc/hipFree of GB-sized blocks cost up to hundreds of
    // milliseconds, which would dominate the multigrid setup that runs inside every solve
    std::multimap<size_t, void *> pool_free_blocks;
    std::unordered_map<void *, size_t> pool_sizes;
    size_t pool_cached_bytes = 0;
};

namespace padne {

int ensure_workspace(padne_ctx *ctx, size_t bytes);
int csr_alloc(padne_ctx *ctx, int64_t n_rows, int64_t n_cols, int64_t nnz, padne_csr **out);
int csr_shrink_nnz(padne_ctx *ctx, padne_csr *m, int64_t nnz);      // allocated for a bound, nnz known later
int csr_build_dinv(padne_ctx *ctx, padne_csr *m);

// kernels launched from several translation units
int launch_spmv(padne_ctx *ctx, const padne_csr *m, const double *x, double *y,
                const double *dot_with /* may be null */, double *partials /* may be null */,
                const int32_t *done_flag /* may be null */);
int spmv_grid(const padne_csr *m);
enum { SPMV_PLAIN = 0, SPMV_DOT = 1, SPMV_RESID = 2, SPMV_ADD = 3, SPMV_JACOBI = 4,
       SPMV_DOT_AUX = 5,     // same as SPMV_DOT; used outside the CG loop (Lanczos estimates) so that kernel profiles keep the two apart
       SPMV_WUP = 6,         // y = aux0 + scale * aux2 * aux1 + W x  (spmv.hip)
       SPMV_RESTRICT = 7,    // y = A x ; y2 = scale * aux2 * y
       SPMV_RESID_PRE = 8 }; // y = x - A (scale * aux2 .* x): residual of the sweep from zero, x the right-hand side (spmv.hip)
int launch_spmv_mode(padne_ctx *ctx, const padne_csr *m, int mode, const double *x, double *y,
                     const double *dot_with, double *partials, const int32_t *done_flag, const double *aux1,
                     const double *aux2, double scale);

int launch_spmv_f32(padne_ctx *ctx, const padne_csr *m, int mode, const float *x, float *y, double *partials,
                    const int32_t *done_flag, const float *aux1, const float *aux2, float scale);
int launch_spmv_f32_restrict(padne_ctx *ctx, const padne_csr *R, const float *r, float *b_c, float *x_c,
                             const int32_t *done_flag, const float *dinv_c, float c);
int launch_spmv_f32_exit(padne_ctx *ctx, const padne_csr *m, const float *x, double *y, const double *dot_with,
                         double *partials, const int32_t *done_flag, const float *aux1, const float *aux2, float scale,
                         const double *out_scale2, float *z32 = nullptr);
int launch_spmv_f32_wup_exit(padne_ctx *ctx, const padne_csr *w, const float *e, double *z, const double *dot_with,
                             double *partials, const int32_t *done_flag, const float *x_pre, const float *r_pre,
                             const float *dinv32, float scale, const double *out_scale2, float *z32 = nullptr,
                             const float *dot_b32 = nullptr);
int csr_build_f32(padne_ctx *ctx, padne_csr *m);
int csr_build_xw_plan(padne_ctx *ctx, padne_csr *m);
int csr_build_xw_plan_wide(padne_ctx *ctx, padne_csr *m, int grid_cap = 0);      // grid_cap: workgroups at most (a build that runs beside latency-bound work of the other stream)      // twelve runs of 20 (single-precision operators with float values only: W)
// interior / boundary tiles of a row-partitioned operator whose first n_owned columns are the rank's own unknowns
int csr_build_split_plan(padne_ctx *ctx, padne_csr *m, long long n_owned);
// number of per-workgroup partial sums a product with a dot epilogue on `m` writes (spmv_grid, or the grids of the
// interior and the boundary launch together)
int spmv_partials(const padne_csr *m);
// which part of a split operator a launch covers: everything (one after the other), the interior tiles, the boundary
// tiles.  On an operator without a split plan SPMV_INTERIOR does nothing and SPMV_BOUNDARY is the whole product.
enum { SPMV_ALL = 0, SPMV_INTERIOR = 1, SPMV_BOUNDARY = 2 };
int launch_spmv_part(padne_ctx *ctx, const padne_csr *m, int mode, int part, const double *x, double *y, const double *dot_with,
                     double *partials, const int32_t *done_flag, const double *aux1, const double *aux2, double scale);
int launch_spmv_f32_part(padne_ctx *ctx, const padne_csr *m, int mode, int part, const float *x, float *y, double *partials,
                         const int32_t *done_flag, const float *aux1, const float *aux2, float scale);
int launch_spmv_f32_exit_part(padne_ctx *ctx, const padne_csr *m, int part, const float *x, double *y, const double *dot_with,
                              double *partials, const int32_t *done_flag, const float *aux1, const float *aux2, float scale,
                              const double *out_scale2, float *z32 = nullptr);

// spmm.hip: the same products for 8 interleaved right-hand sides (vectors [n][8]; aux2 = 1/diag stays [n]);
// dot partials are [8][kMaxPartials]
int spmm8_grid(const padne_csr *m);
int launch_spmm8_mode(padne_ctx *ctx, const padne_csr *m, int mode, const double *x, double *y, const double *dot_with,
                      double *partials, const int32_t *done_flag, const double *aux1, const double *aux2, double scale);
// the same for k = 8, 4 or 2 interleaved right-hand sides (vectors [n][k]; dot partials [k][kMaxPartials])
int launch_spmm_mode(padne_ctx *ctx, const padne_csr *m, int k, int mode, const double *x, double *y, const double *dot_with,
                     double *partials, const int32_t *done_flag, const double *aux1, const double *aux2, double scale);
int launch_spmm_f32(padne_ctx *ctx, const padne_csr *m, int k, int mode, const float *x, float *y, double *partials,
                    const int32_t *done_flag, const float *aux1, const float *aux2, float scale);
int launch_spmm_f32_exit(padne_ctx *ctx, const padne_csr *m, int k, const float *x, double *y, const double *dot_with,
                         double *partials, const int32_t *done_flag, const float *aux1, const float *aux2,
                         float scale, const double *out_scale2, float *y32 = nullptr);
int launch_spmm_f32_wup_exit(padne_ctx *ctx, const padne_csr *w, int k, const float *e, double *z, const double *dot_with,
                             double *partials, const int32_t *done_flag, const float *x_pre, const float *r_pre,
                             const float *dinv32, float scale, const double *out_scale2, float *z32 = nullptr,
                             const float *rhs = nullptr);
bool spmv_resid_pre_ok(const padne_csr *m);
int launch_spmv_f32_resid_pre(padne_ctx *ctx, const padne_csr *m, const float *b, float *resid, const int32_t *done_flag,
                              const float *dinv32, float c);
bool spmv_x32_ok(const padne_csr *m);
int launch_spmv_dot_x32(padne_ctx *ctx, const padne_csr *m, const float *x, double *y, double *partials, const int32_t *done_flag);
int launch_spmv_f32_wup(padne_ctx *ctx, const padne_csr *w, const float *e, float *x_out, const int32_t *done_flag,
                        const float *x_pre, const float *r_pre, const float *dinv32, float scale);
int launch_spmm_f32_wup(padne_ctx *ctx, const padne_csr *w, int k, const float *e, float *x_out, const int32_t *done_flag,
                        const float *x_pre, const float *r_pre, const float *dinv32, float scale);
int interleave(padne_ctx *ctx, long long n, int k, const double *src, double *dst, bool to_interleaved);

// exclusive scan of int32 counts into int32 offsets (n+1 outputs); returns total via host
int exclusive_scan_i32(padne_ctx *ctx, const int32_t *in, int32_t *out, int64_t n, int64_t *total);
// Small results the host has to see before it can go on (counts that size the next allocation, flags that choose the
// next kernel).  hipMemcpyAsync to pageable memory + hipStreamSynchronize costs ~28 us of idle GPU per round trip and the
// multigrid setup makes ~75 of them: instead the device stores the words into a host-coherent page (one 64-byte slot per
// request: sequence number + up to 56 bytes) and the host polls the sequence number.  mail_ticket hands a kernel the slot
// to post to (mail_post, from one thread, after the data is final); read_back queues a one-wave kernel that posts a
// device buffer.  PADNE_NO_MAILBOX=1: memcpy + synchronise as before.
struct MailTicket {
    unsigned long long *slot_host = nullptr, *slot_dev = nullptr;   // slot_dev == nullptr: the mailbox is off
    unsigned long long seq = 0;
};
MailTicket mail_ticket(padne_ctx *ctx);
int mail_wait(padne_ctx *ctx, const MailTicket &t, void *out, size_t bytes);
int read_back(padne_ctx *ctx, const void *dev, size_t bytes, void *host_out);
int read_back2(padne_ctx *ctx, const void *dev, size_t bytes, void *host_out, const void *dev2, size_t bytes2, void *host_out2);
// the exclusive scan in two halves (assemble.hip): what the caller launches between them overlaps with the scan
struct ScanTicket {
    int nb = 0;
    long long *bs = nullptr;
    bool want_total = false;
    MailTicket mail;
};
int scan_i32_begin(padne_ctx *ctx, const int32_t *in, int32_t *out, int64_t n, ScanTicket *t, bool want_total);
int scan_i32_end(padne_ctx *ctx, ScanTicket *t, long long total_and_negative[2]);
__device__ __forceinline__ void mail_post(unsigned long long *slot, unsigned long long seq, const unsigned long long *words, int n_words) {
    for (int k = 0; k < n_words; ++k) __hip_atomic_store(slot + 1 + k, words[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(slot, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// the same without reading the total back (no host synchronisation; the caller knows the total fits 32 bits)
int exclusive_scan_i32_async(padne_ctx *ctx, const int32_t *in, int32_t *out, int64_t n);
// negative inputs are a message from the producer (a row that asks for another path), not an error: reported, and the
// offsets are meaningless then
int exclusive_scan_i32_flagged(padne_ctx *ctx, const int32_t *in, int32_t *out, int64_t n, int64_t *total, bool *negative);

// The context's second stream, as a context of its own (created on first use, destroyed with its parent).
// Memory first touched on that stream must come from ITS pool: a block of the parent's pool may still be in use by
// kernels queued on the parent's stream.  pool_free(parent, p) finds blocks of either pool.
padne_ctx *aux_context(padne_ctx *ctx);
// work queued on `later` after this call starts only when everything queued on `earlier` so far has finished
int stream_order(padne_ctx *earlier, padne_ctx *later);

// comm.cpp
int comm_allreduce_sum_f64(padne_ctx *ctx, double *dev_buf, int count);
int comm_allgather_f64(padne_ctx *ctx, const double *send, double *recv, int count_per_rank);
int comm_allgather_f32(padne_ctx *ctx, const float *send, float *recv, int count_per_rank);
void comm_destroy(padne_ctx *ctx);
void comm_abort(padne_ctx *ctx);   // a rank that leaves a collective phase with an error: wake the team / abort the communicator
// Peer-to-peer halo exchange: instead of packing its exported values into its own segment and taking part in an
// all-gather, a rank STORES them straight into every rank's mailbox (a ring of kP2pRing entries of world * m values; entry
// = exchange number % kP2pRing) and the receiver copies its mailbox entry behind its owned values once the stores have
// landed.  comm_p2p_begin: true if the context can do that for m slots per rank (sets up / grows the mailboxes on first
// use -- collectively: every rank runs the same sequence of exchanges); returns the device table of the ranks' rings and
// the byte offset of this exchange's entry.  comm_p2p_arrive: all ranks' stores of the current exchange are visible.
constexpr int kP2pRing = 4;
// header of a mailbox shared between processes: [0, 1024) one 8-byte arrival flag per sender (the sequence number + 1 of
// the last exchange whose stores of that sender have landed), [1024] the block counter of the sender's store kernel,
// [1536] the error word a receiver sets when a wait runs out
constexpr int kP2pHeaderBytes = 4096, kP2pCounterOff = 1024, kP2pErrorOff = 1536, kP2pMaxWorld = 128;
inline bool comm_active(const padne_ctx *c) { return c->comm != nullptr || c->team != nullptr || c->hostcoll_fn != nullptr; }
bool comm_p2p_enabled(const padne_ctx *ctx);
bool comm_p2p_fits(const padne_ctx *ctx, int m);       // can an exchange with m slots per rank go peer to peer?
int comm_p2p_check(padne_ctx *ctx);                    // PADNE_E_COMM if a receiver of this context gave up waiting
bool comm_exchange_overlaps(const padne_ctx *ctx);
int comm_p2p_begin(padne_ctx *ctx, int m, void ***peers_dev, size_t *entry_offset);
int comm_p2p_arrive(padne_ctx *ctx);
void comm_p2p_release(padne_ctx *ctx);

// exchange plan of a row-partitioned operator: vectors are [n_owned | world * m exchanged values]; every
// rank packs its n_export (<= m) exported owned entries into its segment and one all-gather fills the rest
struct HaloPlan {
    long long n_owned = 0;
    int m = 0, n_export = 0;
    const int32_t *export_idx = nullptr;   // device
};
int halo_exchange_plan(padne_ctx *ctx, const HaloPlan &plan, double *v, const int32_t *done_flag);
int halo_exchange_plan_f32(padne_ctx *ctx, const HaloPlan &plan, float *v, const int32_t *done_flag);
// the same exchange in two halves (pcg.hip): what needs no remote value goes between them
struct HaloTicket {
    bool p2p = false;
    size_t entry_off = 0;
    unsigned long long seq1 = 0;     // mailboxes shared between processes: the exchange's sequence number + 1 (0: in-process team)
};
int halo_send(padne_ctx *ctx, const HaloPlan &plan, double *v, const int32_t *done_flag, HaloTicket *tk);
int halo_send_f32(padne_ctx *ctx, const HaloPlan &plan, float *v, const int32_t *done_flag, HaloTicket *tk);
int halo_recv(padne_ctx *ctx, const HaloPlan &plan, double *v, const int32_t *done_flag, const HaloTicket &tk);
int halo_recv_f32(padne_ctx *ctx, const HaloPlan &plan, float *v, const int32_t *done_flag, const HaloTicket &tk);

// amg.hip
void amg_destroy(void *amg);
// pcg.hip: largest eigenvalue of D^-1 A from `steps` Lanczos (Jacobi-PCG) steps; with a plan the matrix is
// this rank's rows of a row-partitioned operator and the estimate (identical on all ranks) is the global one
int estimate_lambda_max(padne_ctx *ctx, const padne_csr *a, int steps, double *lambda, const HaloPlan *plan = nullptr);
// the same in two halves: queue the steps (no host synchronisation), later wait for them and evaluate
struct LanczosJob {
    padne_ctx *ctx = nullptr;
    int steps = 0;
    double *hist = nullptr;          // device history of the step scalars (pool of ctx)
    std::vector<double> host;        // its host copy, valid after lanczos_finish
    double *host_dst = nullptr;      // where the queue copies it first (pinned memory on the second stream)
    int pinned_slot = -1;            // its slot there, released by lanczos_finish
};
int lanczos_enqueue(padne_ctx *ctx, const padne_csr *a, int steps, LanczosJob *job, const HaloPlan *plan = nullptr);
int lanczos_finish(LanczosJob *job, double *lambda);
// assemble.hip: rows of `top` followed by the rows of `bottom`, n_cols columns
int csr_vstack(padne_ctx *ctx, const padne_csr *top, const padne_csr *bottom, int64_t n_cols, padne_csr **out);

// Per-context caching allocator.  All work of a context is ordered on its one stream, so a block handed back
// may be reused by later launches without synchronisation.  Blocks are kept (up to kPoolCacheLimit) until the
// context is destroyed.
void *pool_alloc(padne_ctx *ctx, size_t bytes);     // nullptr on failure (error message set)
void pool_free(padne_ctx *ctx, void *p);
void pool_release_all(padne_ctx *ctx);

// assemble.hip helpers shared with amg.hip
struct Scratch {   // device allocations returned to the pool on scope exit
    padne_ctx *ctx;
    std::vector<void *> ptrs;
    explicit Scratch(padne_ctx *c) : ctx(c) {}
    ~Scratch() { release(); }
    void release() {
        for (void *p : ptrs) if (p) pool_free(ctx, p);
        ptrs.clear();
    }
    void disown(void *p) {          // the caller keeps p beyond this scope
        for (void *&q : ptrs) if (q == p) q = nullptr;
    }
    template <typename T> int alloc(T **out, size_t count) {
        void *p = pool_alloc(ctx, sizeof(T) * (count ? count : 1));
        if (p == nullptr) return PADNE_E_NOMEM;
        ptrs.push_back(p);
        *out = (T *)p;
        return PADNE_OK;
    }
};
static inline unsigned nblk(long long n, int bs = 256) {
    const long long b = (n + bs - 1) / bs;
    return (unsigned)(b > 0 ? b : 1);
}
// sort each row's slots by key = (col << 32 | seq), add duplicates in key order, compact in place
int merge_slots_generic(padne_ctx *ctx, long long n_rows, const int *slot_ptr, long long *key, double *val,
                        int *row_len);
// slots without duplicates (transposes): sort each row by key, nothing else
int sort_slots_exact(padne_ctx *ctx, long long n_rows, const int *slot_ptr, long long *key, double *val, int *row_len_scratch);
// rows already compacted at their slot offsets (key >> 32 = column) -> new CSR matrix
int csr_from_slots(padne_ctx *ctx, long long n_rows, long long n_cols, const int *slot_ptr, const long long *key,
                   const double *val, const int *row_len, padne_csr **out,
                   int (*while_scanning)(void *) = nullptr, void *while_scanning_arg = nullptr);
// the same when every row fills its slots exactly (slot_ptr is the row pointer): no scan, no synchronisation
int csr_from_exact_slots(padne_ctx *ctx, long long n_rows, long long n_cols, long long nnz, const int *slot_ptr,
                         const long long *key, const double *val, padne_csr **out);

}  // namespace padne
