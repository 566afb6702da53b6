// C ABI plumbing: contexts, device memory, CSR hand-off, SpMV entry points, timing.
#include "common.hpp"

#include <algorithm>

#include <math.h>
#include <stdarg.h>
#include <string.h>
#include <chrono>

namespace padne {

std::atomic<long long> g_launch_count{0};

void options_from_env(padne_options *o) {
    *o = padne_options();
    auto on = [](const char *name) { const char *e = getenv(name); return e != nullptr && e[0] != '\0' && !(e[0] == '0' && e[1] == '\0'); };
    auto has = [](const char *list, const char *word) {
        const size_t n = strlen(word);
        for (const char *p = list; p != nullptr && *p != '\0';) {
            const char *q = strchr(p, ',');
            const size_t len = q != nullptr ? (size_t)(q - p) : strlen(p);
            if (len >= n && strncmp(p, word, n) == 0 && (len == n || p[n] == ':')) return p;
            p = q != nullptr ? q + 1 : nullptr;
        }
        return (const char *)nullptr;
    };
    o->amg_f64 = on("PADNE_AMG_F64");
    if (const char *e = getenv("PADNE_AMG_W")) o->amg_w = strcmp(e, "none") == 0 ? 0 : (strcmp(e, "fine") == 0 ? 1 : 2);
    o->amg_exchange_all = on("PADNE_AMG_EXCHANGE_ALL");
    o->pcg_p64 = on("PADNE_PCG_P64");
    o->pcg_no_xhist = on("PADNE_PCG_NO_XHIST");
    o->gj_vector = on("PADNE_GJ_VECTOR");
    o->no_batch = on("PADNE_NO_BATCH");
    o->no_mailbox = on("PADNE_NO_MAILBOX");
    o->no_p2p = on("PADNE_NO_P2P");
    o->no_split = on("PADNE_NO_SPLIT");
    o->no_xwindow = on("PADNE_NO_XWINDOW");
    o->setup_one_stream = on("PADNE_SETUP_ONE_STREAM");
    if (const char *e = getenv("PADNE_CG_SINGLE_REDUCTION")) o->cg_single_reduction = atoi(e) != 0 ? 1 : 0;
    if (const char *e = getenv("PADNE_LOCKSTEP_NARROW")) o->lockstep_narrow = atoi(e);
    if (const char *e = getenv("PADNE_AMG_COARSE_N")) o->amg_coarse_n = std::min(4096, std::max(16, atoi(e)));
    if (const char *e = getenv("PADNE_AMG_GATHER_N")) o->amg_gather_n = atoll(e);
    if (const char *e = getenv("PADNE_P2P_TIMEOUT_MS")) {
        const long v = atol(e);
        if (v > 0 && v <= 600000) o->p2p_timeout_ms = (unsigned)v;
    }
    if (const char *f = getenv("PADNE_FORCE")) {
        o->force_asm_hash = has(f, "asm_hash") != nullptr;
        o->force_asm_two_pass = has(f, "asm_two_pass") != nullptr;
        o->force_relabel_slots = has(f, "relabel_slots") != nullptr;
        o->force_transpose_cursors = has(f, "transpose_cursors") != nullptr;
        o->force_xhist_small = has(f, "xhist_small") != nullptr;
        o->force_relabel_lanes = has(f, "relabel_lanes") != nullptr;
        if (const char *w = has(f, "spgemm_split")) o->force_spgemm_split = w[12] == ':' ? atoll(w + 13) : 30000;
    }
    if (const char *v = getenv("PADNE_VERBOSE")) {
        o->verbose_amg = has(v, "amg") != nullptr;
        o->verbose_xw = has(v, "xw") != nullptr;
        o->verbose_pool = has(v, "pool") != nullptr;
    }
}

static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int ensure_workspace(padne_ctx *ctx, size_t bytes) {
    if (ctx->ws_bytes >= bytes) return PADNE_OK;
    if (ctx->ws) {
        PADNE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        PADNE_HIP_CHECK(hipFree(ctx->ws));
        ctx->ws = nullptr;
        ctx->ws_bytes = 0;
    }
    if (hipMalloc(&ctx->ws, bytes) != hipSuccess) {
        set_error("hipMalloc of %zu workspace bytes failed", bytes);
        return PADNE_E_NOMEM;
    }
    ctx->ws_bytes = bytes;
    return PADNE_OK;
}

constexpr size_t kPoolCacheLimit = (size_t)240 << 30;  // bytes kept for reuse (the card has 288 GB; a failed hipMalloc releases them and retries)

static size_t pool_round(size_t bytes) {
    if (bytes < 256) bytes = 256;
    if (bytes >= ((size_t)1 << 20)) return (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);   // MiB granules
    size_t p = 256;
    while (p < bytes) p <<= 1;
    return p;
}

void *pool_alloc(padne_ctx *ctx, size_t bytes) {
    const size_t want = pool_round(bytes);
    auto it = ctx->pool_free_blocks.lower_bound(want);
    if (it != ctx->pool_free_blocks.end() && it->first <= want + want / 2 + ((size_t)1 << 20)) {
        void *p = it->second;
        ctx->pool_cached_bytes -= it->first;
        ctx->pool_free_blocks.erase(it);
        return p;
    }
    void *p = nullptr;
    const bool trace = ctx->opt.verbose_pool;      // every miss of the cache (steady state: none)
    if (trace) fprintf(stderr, "[pool] %s hipMalloc %zu bytes (cached %zu)\n", ctx->is_aux ? "aux" : "main", want, ctx->pool_cached_bytes);
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {   // give cached blocks back to the driver and retry once
        (void)hipGetLastError();
        pool_release_all(ctx);
        e = hipMalloc(&p, want);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error("hipMalloc of %zu bytes failed: %s", want, hipGetErrorString(e));
        return nullptr;
    }
    ctx->pool_sizes[p] = want;
    return p;
}

void pool_free(padne_ctx *ctx, void *p) {
    if (p == nullptr) return;
    auto it = ctx->pool_sizes.find(p);
    if (it == ctx->pool_sizes.end()) {
        padne_ctx *other = ctx->is_aux ? ctx->parent : ctx->aux;     // a block of the sibling stream's pool
        if (other != nullptr && other->pool_sizes.count(p)) {
            pool_free(other, p);
            return;
        }
        (void)hipFree(p);                // not ours: plain free
        return;
    }
    if (ctx->pool_cached_bytes + it->second > kPoolCacheLimit) {
        if (ctx->opt.verbose_pool) fprintf(stderr, "[pool] cache limit: hipFree %zu bytes\n", it->second);
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(p);
        ctx->pool_sizes.erase(it);
        return;
    }
    ctx->pool_free_blocks.emplace(it->second, p);
    ctx->pool_cached_bytes += it->second;
}

void pool_release_all(padne_ctx *ctx) {
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (auto &kv : ctx->pool_free_blocks) {
        (void)hipFree(kv.second);
        ctx->pool_sizes.erase(kv.second);
    }
    ctx->pool_free_blocks.clear();
    ctx->pool_cached_bytes = 0;
}

static int ctx_init_resources(padne_ctx *ctx) {
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void **)&ctx->partials, sizeof(double) * 8 * kMaxPartials) != hipSuccess ||
        hipMalloc((void **)&ctx->scalars, sizeof(double) * 64) != hipSuccess ||
        hipMalloc((void **)&ctx->status, 1024) != hipSuccess ||
        hipHostMalloc(&ctx->pinned, kPinnedBytes, hipHostMallocDefault) != hipSuccess ||
        hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_order, hipEventDisableTiming) != hipSuccess) {
        set_error("context creation failed: %s", hipGetErrorString(hipGetLastError()));
        return PADNE_E_HIP;
    }
    hipMemsetAsync(ctx->partials, 0, sizeof(double) * 8 * kMaxPartials, ctx->stream);
    hipMemsetAsync(ctx->scalars, 0, sizeof(double) * 64, ctx->stream);
    hipMemsetAsync(ctx->status, 0, 1024, ctx->stream);
    hipStreamSynchronize(ctx->stream);
    // the mailbox is an optimisation: without host-coherent memory (or with PADNE_NO_MAILBOX=1) read_back copies and synchronises
    if (hipHostMalloc((void **)&ctx->mailbox, 4096, hipHostMallocCoherent | hipHostMallocMapped) == hipSuccess) {
        if (hipHostGetDevicePointer((void **)&ctx->mailbox_dev, ctx->mailbox, 0) == hipSuccess) {
            memset(ctx->mailbox, 0, 4096);
        } else {
            (void)hipHostFree(ctx->mailbox);
            ctx->mailbox = nullptr;
            ctx->mailbox_dev = nullptr;
        }
    }
    (void)hipGetLastError();
    return PADNE_OK;
}

// up to two device buffers (the second starts at the next 8-byte boundary of the payload)
__global__ void mail_post_kernel(const unsigned char *__restrict__ src, int n_bytes, const unsigned char *__restrict__ src2,
                                 int n_bytes2, unsigned long long *slot, unsigned long long seq) {
    __shared__ unsigned long long w[7];
    if (threadIdx.x < 7) w[threadIdx.x] = 0ull;
    __syncthreads();
    const int off2 = (n_bytes + 7) & ~7;
    if ((int)threadIdx.x < n_bytes) ((unsigned char *)w)[threadIdx.x] = src[threadIdx.x];
    if ((int)threadIdx.x < n_bytes2) ((unsigned char *)w)[off2 + threadIdx.x] = src2[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) mail_post(slot, seq, w, (off2 + n_bytes2 + 7) / 8);
}

MailTicket mail_ticket(padne_ctx *ctx) {
    MailTicket t;
    if (ctx->mailbox_dev == nullptr || ctx->opt.no_mailbox) return t;
    t.seq = ++ctx->mail_seq;
    const size_t off = (size_t)(t.seq & 31ull) * 8;       // 32 slots of 64 bytes; one request is in flight at a time
    t.slot_host = ctx->mailbox + off;
    t.slot_dev = ctx->mailbox_dev + off;
    return t;
}

int mail_wait(padne_ctx *ctx, const MailTicket &t, void *out, size_t bytes) {
    if (t.slot_dev == nullptr || bytes > 56) {
        set_error("mailbox request without a slot");
        return PADNE_E_INVALID;
    }
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (__atomic_load_n(t.slot_host, __ATOMIC_ACQUIRE) != t.seq) {
        __builtin_ia32_pause();
        if ((++spins & 4095u) == 0u &&
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 5.0) {
            // never seen in the tests: the stream is drained and the slot looked at once more before giving up
            const hipError_t e = hipStreamSynchronize(ctx->stream);
            if (e == hipSuccess && __atomic_load_n(t.slot_host, __ATOMIC_ACQUIRE) == t.seq) break;
            set_error("mailbox: no answer from the device (%s)", hipGetErrorString(e));
            return PADNE_E_HIP;
        }
    }
    memcpy(out, t.slot_host + 1, bytes);
    return PADNE_OK;
}

int read_back2(padne_ctx *ctx, const void *dev, size_t bytes, void *host_out, const void *dev2, size_t bytes2, void *host_out2) {
    const size_t off2 = (bytes + 7) & ~(size_t)7;
    const MailTicket t = off2 + bytes2 <= 56 ? mail_ticket(ctx) : MailTicket();
    if (t.slot_dev == nullptr) {
        PADNE_HIP_CHECK(hipMemcpyAsync(host_out, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
        if (bytes2 > 0) PADNE_HIP_CHECK(hipMemcpyAsync(host_out2, dev2, bytes2, hipMemcpyDeviceToHost, ctx->stream));
        PADNE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        return PADNE_OK;
    }
    hipLaunchKernelGGL(mail_post_kernel, dim3(1), dim3(64), 0, ctx->stream, (const unsigned char *)dev, (int)bytes,
                       (const unsigned char *)dev2, (int)bytes2, t.slot_dev, t.seq);
    PADNE_HIP_CHECK(hipGetLastError());
    unsigned char buf[56];
    PADNE_TRY(mail_wait(ctx, t, buf, off2 + bytes2));
    memcpy(host_out, buf, bytes);
    if (bytes2 > 0) memcpy(host_out2, buf + off2, bytes2);
    return PADNE_OK;
}

int read_back(padne_ctx *ctx, const void *dev, size_t bytes, void *host_out) {
    return read_back2(ctx, dev, bytes, host_out, nullptr, 0, nullptr);
}

padne_ctx *aux_context(padne_ctx *ctx) {
    if (ctx->is_aux) return ctx;
    if (ctx->aux != nullptr) return ctx->aux;
    padne_ctx *a = new padne_ctx();
    a->device = ctx->device;
    a->is_aux = true;
    a->parent = ctx;
    a->rank = ctx->rank;
    a->world = ctx->world;
    a->opt = ctx->opt;
    if (ctx_init_resources(a) != PADNE_OK) {
        padne_ctx_destroy(a);
        return nullptr;
    }
    ctx->aux = a;
    return a;
}

int stream_order(padne_ctx *earlier, padne_ctx *later) {
    if (earlier == later) return PADNE_OK;
    PADNE_HIP_CHECK(hipEventRecord(earlier->ev_order, earlier->stream));
    PADNE_HIP_CHECK(hipStreamWaitEvent(later->stream, earlier->ev_order, 0));
    return PADNE_OK;
}

__global__ void csr_zero_pads(int32_t *__restrict__ cols_pad, double *__restrict__ vals_pad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // grid = kPadNnz threads
    cols_pad[i] = 0;
    vals_pad[i] = 0.0;
}

int csr_alloc(padne_ctx *ctx, int64_t n_rows, int64_t n_cols, int64_t nnz, padne_csr **out) {
    PADNE_REQUIRE(n_rows >= 0 && n_cols >= 0 && nnz >= 0, "negative size");
    PADNE_REQUIRE(nnz < (int64_t)2147483647 - kPadNnz && n_rows < 2147483647 && n_cols < 2147483647,
                  "matrix too large for int32 indices");
    padne_csr *m = new padne_csr();
    m->n_rows = n_rows;
    m->n_cols = n_cols;
    m->nnz = nnz;
    m->device = ctx->device;
    m->owner = ctx;
    const size_t ne = (size_t)nnz + kPadNnz;
    m->rowptr = (int32_t *)pool_alloc(ctx, sizeof(int32_t) * (size_t)(n_rows + 1));
    m->cols = (int32_t *)pool_alloc(ctx, sizeof(int32_t) * ne);
    m->vals = (double *)pool_alloc(ctx, sizeof(double) * ne);
    if (!m->rowptr || !m->cols || !m->vals) {
        padne_csr_destroy(m);
        return PADNE_E_NOMEM;
    }
    // zero the padding (column 0 / value 0.0) so the SpMV tile loads never need a bounds check: one launch for both arrays
    hipLaunchKernelGGL(csr_zero_pads, dim3(kPadNnz / 256), dim3(256), 0, ctx->stream, m->cols + nnz, m->vals + nnz);
    PADNE_HIP_CHECK(hipGetLastError());
    *out = m;
    return PADNE_OK;
}

// a matrix allocated for an upper bound of its entries learns the real count: the padding moves behind the real end
int csr_shrink_nnz(padne_ctx *ctx, padne_csr *m, int64_t nnz) {
    PADNE_REQUIRE(nnz >= 0 && nnz <= m->nnz, "entry count beyond the allocation");
    m->nnz = nnz;
    hipLaunchKernelGGL(csr_zero_pads, dim3(kPadNnz / 256), dim3(256), 0, ctx->stream, m->cols + nnz, m->vals + nnz);
    PADNE_HIP_CHECK(hipGetLastError());
    return PADNE_OK;
}

__global__ void dot_partial_kernel(const long long n, const double *__restrict__ a, const double *__restrict__ b,
                                   double *__restrict__ partials) {
    __shared__ double red[4];
    double s = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const double d = a[i] - b[i];
        s += d * d;
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

}  // namespace padne

using namespace padne;

// (test header) the environment switches again, for a context that lives across a change of them
extern "C" int padne_ctx_reload_options(padne_ctx *ctx) {
    PADNE_REQUIRE(ctx, "ctx");
    options_from_env(&ctx->opt);
    ctx->p2p_timeout_ms = ctx->opt.p2p_timeout_ms;
    if (ctx->aux != nullptr) ctx->aux->opt = ctx->opt;
    return PADNE_OK;
}

extern "C" int padne_launch_count(long long *count) {
    PADNE_REQUIRE(count, "null argument");
    *count = g_launch_count.load(std::memory_order_relaxed);
    return PADNE_OK;
}

extern "C" {

int padne_abi_version(void) { return PADNE_ABI_VERSION; }

const char *padne_last_error(void) { return g_err; }

int padne_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return PADNE_E_HIP;
    }
    return n;
}

int padne_ctx_create(int device, padne_ctx **out) {
    PADNE_REQUIRE(out != nullptr, "out");
    int n = 0;
    PADNE_HIP_CHECK(hipGetDeviceCount(&n));
    if (n <= 0) {
        set_error("no HIP device visible: libpadne_hip has no CPU fallback");
        return PADNE_E_HIP;
    }
    PADNE_REQUIRE(device >= 0 && device < n, "device index out of range");
    PADNE_HIP_CHECK(hipSetDevice(device));
    padne_ctx *ctx = new padne_ctx();
    ctx->device = device;
    options_from_env(&ctx->opt);       // the PADNE_* switches are read here, once
    ctx->p2p_timeout_ms = ctx->opt.p2p_timeout_ms;
    if (ctx_init_resources(ctx) != PADNE_OK) {
        padne_ctx_destroy(ctx);
        return PADNE_E_HIP;
    }
    *out = ctx;
    return PADNE_OK;
}

int padne_ctx_destroy(padne_ctx *ctx) {
    if (!ctx) return PADNE_OK;
    hipSetDevice(ctx->device);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    if (ctx->aux) {
        padne_ctx_destroy(ctx->aux);
        ctx->aux = nullptr;
    }
    comm_destroy(ctx);
    pool_release_all(ctx);
    for (auto &kv : ctx->pool_sizes) (void)hipFree(kv.first);   // blocks still held by live matrices
    ctx->pool_sizes.clear();
    if (ctx->halo_export) hipFree(ctx->halo_export);
    if (ctx->ws) hipFree(ctx->ws);
    if (ctx->partials) hipFree(ctx->partials);
    if (ctx->scalars) hipFree(ctx->scalars);
    if (ctx->status) hipFree(ctx->status);
    if (ctx->pinned) hipHostFree(ctx->pinned);
    if (ctx->mailbox) hipHostFree(ctx->mailbox);
    if (ctx->ev0) hipEventDestroy(ctx->ev0);
    if (ctx->ev1) hipEventDestroy(ctx->ev1);
    if (ctx->ev_order) hipEventDestroy(ctx->ev_order);
    for (hipStream_t &cs : ctx->copy_stream)
        if (cs != nullptr) {
            (void)hipStreamSynchronize(cs);
            (void)hipStreamDestroy(cs);
            cs = nullptr;
        }
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return PADNE_OK;
}

int padne_ctx_synchronize(padne_ctx *ctx) {
    PADNE_REQUIRE(ctx, "ctx");
    PADNE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PADNE_OK;
}

void *padne_ctx_stream(padne_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int padne_dev_alloc(padne_ctx *ctx, int64_t bytes, void **dev_out) {
    PADNE_REQUIRE(ctx && dev_out && bytes >= 0, "arguments");
    PADNE_HIP_CHECK(hipSetDevice(ctx->device));
    if (hipMalloc(dev_out, bytes > 0 ? (size_t)bytes : 8) != hipSuccess) {
        set_error("hipMalloc(%lld) failed", (long long)bytes);
        return PADNE_E_NOMEM;
    }
    return PADNE_OK;
}

int padne_dev_free(padne_ctx *ctx, void *dev) {
    PADNE_REQUIRE(ctx, "ctx");
    if (dev) {
        PADNE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        PADNE_HIP_CHECK(hipFree(dev));
    }
    return PADNE_OK;
}

int padne_dev_upload(padne_ctx *ctx, void *dev_dst, const void *host_src, int64_t bytes) {
    PADNE_REQUIRE(ctx && bytes >= 0 && (bytes == 0 || (dev_dst && host_src)), "arguments");
    if (bytes == 0) return PADNE_OK;
    PADNE_HIP_CHECK(hipMemcpyAsync(dev_dst, host_src, (size_t)bytes, hipMemcpyHostToDevice, ctx->stream));
    PADNE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PADNE_OK;
}

int padne_dev_download(padne_ctx *ctx, void *host_dst, const void *dev_src, int64_t bytes) {
    PADNE_REQUIRE(ctx && bytes >= 0 && (bytes == 0 || (host_dst && dev_src)), "arguments");
    if (bytes == 0) return PADNE_OK;
    PADNE_HIP_CHECK(hipMemcpyAsync(host_dst, dev_src, (size_t)bytes, hipMemcpyDeviceToHost, ctx->stream));
    PADNE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PADNE_OK;
}

int padne_dev_memset(padne_ctx *ctx, void *dev, int value, int64_t bytes) {
    PADNE_REQUIRE(ctx && bytes >= 0 && (bytes == 0 || dev), "arguments");
    if (bytes == 0) return PADNE_OK;
    PADNE_HIP_CHECK(hipMemsetAsync(dev, value, (size_t)bytes, ctx->stream));
    PADNE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PADNE_OK;
}

int padne_csr_from_host(padne_ctx *ctx, int64_t n_rows, int64_t n_cols, const int32_t *indptr,
                        const int32_t *indices, const double *data, padne_csr **out) {
    PADNE_REQUIRE(ctx && out && indptr, "null argument");
    PADNE_REQUIRE(n_rows >= 0 && n_cols >= 0, "negative size");
    const int64_t nnz = indptr[n_rows];
    PADNE_REQUIRE(indptr[0] == 0 && nnz >= 0, "indptr must start at 0");
    PADNE_REQUIRE(nnz == 0 || (indices && data), "null indices/data");
    for (int64_t i = 0; i < n_rows; ++i) PADNE_REQUIRE(indptr[i] <= indptr[i + 1], "indptr not monotone");
    for (int64_t k = 0; k < nnz; ++k)
        PADNE_REQUIRE(indices[k] >= 0 && indices[k] < n_cols, "column index out of range");
    // (scipy's canonical form has ascending columns, but the layout does not demand it: remembered, not refused)
    bool unsorted = false;
    for (int64_t i = 0; i < n_rows && !unsorted; ++i)
        for (int64_t k = indptr[i] + 1; k < indptr[i + 1]; ++k)
            if (indices[k - 1] >= indices[k]) {
                unsorted = true;
                break;
            }
    PADNE_HIP_CHECK(hipSetDevice(ctx->device));
    padne_csr *m = nullptr;
    PADNE_TRY(csr_alloc(ctx, n_rows, n_cols, nnz, &m));
    m->cols_unsorted = unsorted;
    hipError_t e = hipMemcpyAsync(m->rowptr, indptr, sizeof(int32_t) * (size_t)(n_rows + 1),
                                  hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && nnz > 0)
        e = hipMemcpyAsync(m->cols, indices, sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && nnz > 0)
        e = hipMemcpyAsync(m->vals, data, sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        set_error("matrix upload failed: %s", hipGetErrorString(e));
        padne_csr_destroy(m);
        return PADNE_E_HIP;
    }
    *out = m;
    return PADNE_OK;
}

int padne_csr_destroy(padne_csr *m) {
    if (!m) return PADNE_OK;
    hipSetDevice(m->device);
    if (m->amg) amg_destroy(m->amg);
    // the arrays go back to the owner's pool: later work on the same stream may reuse them at once
    if (m->owner) {
        pool_free(m->owner, m->rowptr);
        pool_free(m->owner, m->cols);
        pool_free(m->owner, m->vals);
        pool_free(m->owner, m->dinv);
        pool_free(m->owner, m->vals32);
        pool_free(m->owner, m->dinv32);
        pool_free(m->owner, m->xw_desc);
        pool_free(m->owner, m->mesh_xy);
        pool_free(m->owner, m->mesh_sigma);
        pool_free(m->owner, m->mesh_tri);
        pool_free(m->owner, m->mesh_voff);
        pool_free(m->owner, m->mesh_toff);
        pool_free(m->owner, m->xw_lidx);
        pool_free(m->owner, m->split_tiles);
    }
    delete m;
    return PADNE_OK;
}

int padne_csr_shape(const padne_csr *m, int64_t *n_rows, int64_t *n_cols, int64_t *nnz) {
    PADNE_REQUIRE(m, "matrix");
    if (n_rows) *n_rows = m->n_rows;
    if (n_cols) *n_cols = m->n_cols;
    if (nnz) *nnz = m->nnz;
    return PADNE_OK;
}

int padne_csr_to_host(padne_ctx *ctx, const padne_csr *m, int32_t *indptr, int32_t *indices, double *data) {
    PADNE_REQUIRE(ctx && m && indptr, "null argument");
    PADNE_REQUIRE(m->nnz == 0 || (indices && data), "null indices/data");
    PADNE_HIP_CHECK(hipMemcpyAsync(indptr, m->rowptr, sizeof(int32_t) * (size_t)(m->n_rows + 1),
                                   hipMemcpyDeviceToHost, ctx->stream));
    if (m->nnz > 0) {
        PADNE_HIP_CHECK(hipMemcpyAsync(indices, m->cols, sizeof(int32_t) * (size_t)m->nnz,
                                       hipMemcpyDeviceToHost, ctx->stream));
        PADNE_HIP_CHECK(hipMemcpyAsync(data, m->vals, sizeof(double) * (size_t)m->nnz, hipMemcpyDeviceToHost,
                                       ctx->stream));
    }
    PADNE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PADNE_OK;
}

int padne_spmv_dev(padne_ctx *ctx, const padne_csr *m, const void *x_dev, void *y_dev, int repeat) {
    PADNE_REQUIRE(ctx && m && x_dev && y_dev, "null argument");
    PADNE_REQUIRE(repeat >= 1, "repeat");
    PADNE_HIP_CHECK(hipSetDevice(ctx->device));
    PADNE_TRY(csr_build_xw_plan(ctx, const_cast<padne_csr *>(m)));
    for (int i = 0; i < repeat; ++i)
        PADNE_TRY(launch_spmv(ctx, m, (const double *)x_dev, (double *)y_dev, nullptr, nullptr, nullptr));
    PADNE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PADNE_OK;
}

int padne_spmv(padne_ctx *ctx, const padne_csr *m, const double *x_host, double *y_host) {
    PADNE_REQUIRE(ctx && m && x_host && y_host, "null argument");
    PADNE_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t xb = sizeof(double) * (size_t)m->n_cols, yb = sizeof(double) * (size_t)m->n_rows;
    PADNE_TRY(ensure_workspace(ctx, xb + yb + 512));
    PADNE_TRY(csr_build_xw_plan(ctx, const_cast<padne_csr *>(m)));
    double *x = (double *)ctx->ws;
    double *y = (double *)((char *)ctx->ws + ((xb + 255) & ~(size_t)255));
    PADNE_HIP_CHECK(hipMemcpyAsync(x, x_host, xb, hipMemcpyHostToDevice, ctx->stream));
    PADNE_TRY(launch_spmv(ctx, m, x, y, nullptr, nullptr, nullptr));
    PADNE_HIP_CHECK(hipMemcpyAsync(y_host, y, yb, hipMemcpyDeviceToHost, ctx->stream));
    PADNE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PADNE_OK;
}

int padne_residual_norm(padne_ctx *ctx, const padne_csr *m, const double *x_host, const double *b_host,
                        double *norm_out) {
    PADNE_REQUIRE(ctx && m && x_host && b_host && norm_out, "null argument");
    PADNE_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t xb = ((sizeof(double) * (size_t)m->n_cols) + 255) & ~(size_t)255;
    const size_t yb = ((sizeof(double) * (size_t)m->n_rows) + 255) & ~(size_t)255;
    PADNE_TRY(ensure_workspace(ctx, xb + 2 * yb + 512));
    double *x = (double *)ctx->ws;
    double *y = (double *)((char *)ctx->ws + xb);
    double *b = (double *)((char *)ctx->ws + xb + yb);
    PADNE_HIP_CHECK(hipMemcpyAsync(x, x_host, sizeof(double) * (size_t)m->n_cols, hipMemcpyHostToDevice, ctx->stream));
    PADNE_HIP_CHECK(hipMemcpyAsync(b, b_host, sizeof(double) * (size_t)m->n_rows, hipMemcpyHostToDevice, ctx->stream));
    PADNE_TRY(launch_spmv(ctx, m, x, y, nullptr, nullptr, nullptr));
    long long g = (m->n_rows + 255) / 256;
    if (g > 1024) g = 1024;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(dot_partial_kernel, dim3((unsigned)g), dim3(256), 0, ctx->stream, (long long)m->n_rows, y, b,
                       ctx->partials + 7 * kMaxPartials);
    PADNE_HIP_CHECK(hipGetLastError());
    std::vector<double> h((size_t)g);
    PADNE_HIP_CHECK(hipMemcpyAsync(h.data(), ctx->partials + 7 * kMaxPartials, sizeof(double) * (size_t)g,
                                   hipMemcpyDeviceToHost, ctx->stream));
    PADNE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    double s = 0.0;
    for (double v : h) s += v;
    *norm_out = sqrt(s);
    return PADNE_OK;
}

int64_t padne_spmv_algorithmic_bytes(const padne_csr *m) {
    if (!m) return 0;
    return 12 * m->nnz + 20 * m->n_rows + 4;
}

int padne_spmv_time(padne_ctx *ctx, const padne_csr *m, const void *x_dev, void *y_dev, int warmup, int repeat,
                    double *seconds_per_launch) {
    PADNE_REQUIRE(ctx && m && x_dev && y_dev && seconds_per_launch, "null argument");
    PADNE_REQUIRE(repeat >= 1 && warmup >= 0, "repeat/warmup");
    PADNE_HIP_CHECK(hipSetDevice(ctx->device));
    PADNE_TRY(csr_build_xw_plan(ctx, const_cast<padne_csr *>(m)));
    for (int i = 0; i < warmup; ++i)
        PADNE_TRY(launch_spmv(ctx, m, (const double *)x_dev, (double *)y_dev, nullptr, nullptr, nullptr));
    PADNE_HIP_CHECK(hipEventRecord(ctx->ev0, ctx->stream));
    for (int i = 0; i < repeat; ++i)
        PADNE_TRY(launch_spmv(ctx, m, (const double *)x_dev, (double *)y_dev, nullptr, nullptr, nullptr));
    PADNE_HIP_CHECK(hipEventRecord(ctx->ev1, ctx->stream));
    PADNE_HIP_CHECK(hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    PADNE_HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *seconds_per_launch = (double)ms * 1e-3 / repeat;
    return PADNE_OK;
}

}  // extern "C"
