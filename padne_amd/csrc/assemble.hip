// System assembly on the device.
//
//   padne_assemble_system : triangle soup + lumped COO stamps  ->  CSR of L (reference layout/sign)
//   padne_csr_reduce      : scale * P^T M P (index relabel + merge) -> CSR of the reduced SPD system
//   padne_power_density   : per-face sigma*|grad V|^2 (+ the per-mesh scatter of the potentials)
//
// Reference arithmetic restated here: HalfEdge.cotan (mesh.py:124-139), laplace_operator
// (solver.py:171-213), process_mesh_laplace_operators (solver.py:563-575), the += semantics of
// stamp_network_into_system / setup_ground_node on a lil_matrix (solver.py:469-560),
// compute_triangle_gradient / compute_power_density (solver.py:689-745).
//
// Pipeline (no global sort, no float atomics, bitwise reproducible):
//   1 lists   one pass over the triangles validates them and gives every vertex the list of its incident triangles,
//             each as the pair of its two other corners (integer atomics hand out list positions; nothing depends on
//             the order inside a list)
//   2 listed  rows that cannot be built from a list alone -- stamps, hubs of more triangles than a list holds, the
//             unknowns behind the vertices -- get slots (diagonal placeholder, two per triangle, one per stamp); their
//             terms are sorted by (column, sequence) and merged by one wave per row: the two mesh terms of an edge,
//             scaled by the sheet conductance, stamps in stamp order, the diagonal -(w_1 + w_2 + ...) in ascending
//             column order, exact zeros dropped.  Vertices with 9..12 triangles are built like step 3 into slots.
//   3 rows    one lane per mesh vertex builds its row in registers from the list (pairwise match of the fan's edges,
//             compare-exchange network by column) -- the same additions in the same order as step 2 -- and the rows
//             are written ONCE, in place: the row kernel is persistent, a scanner wave inside it turns the entry counts
//             its workgroups publish into offsets while they build their next tile (asm_rows_in_place)
//   4 place   the listed rows move from their slots to the room step 3 left for them
// Because the sort key fixes the summation order, the values do not depend on the order in which the atomics handed out
// list positions or slots.
//
// This file is compiled with -ffp-contract=off: the cotangent and gradient expressions must
// round exactly like the reference's Python floats (no fused multiply-add).
#include "common.hpp"

#include <atomic>

#include <cmath>

#include <algorithm>

#include <string.h>

namespace padne {

// ------------------------------------------------------------------------------------------
// exclusive scan (int32 in, int32 out[n+1], total as int64)
// ------------------------------------------------------------------------------------------
constexpr int kScanItems = 16;
constexpr int kScanChunk = 256 * kScanItems;

__device__ __forceinline__ int block_exclusive_scan_256(int v, int *lds /*[5]*/, int *total) {
    // inclusive scan inside the wave
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(inc, off, 64);
        if (lane >= off) inc += t;
    }
    if (lane == 63) lds[w] = inc;
    __syncthreads();
    int wave_off = 0;
    for (int i = 0; i < w; ++i) wave_off += lds[i];
    if (total) *total = lds[0] + lds[1] + lds[2] + lds[3];
    __syncthreads();
    return wave_off + inc - v;
}

// per-chunk sums in 64 bits (the counts may be saturated upper bounds); a chunk with a negative input leaves a marker
constexpr long long kScanNegative = (long long)0x8000000000000000ull;
// (both passes over the data read it in 16-byte pieces, lane after lane: a thread walking 16 consecutive ints of its own
// touched 64 different cache lines per load instruction and the two kernels ran at 1.0-1.3 TB/s -- 30 + 65 us for the 10 M
// counts of a fine level, nine times per multigrid setup)
__global__ __launch_bounds__(256) void scan_block_sums(const int *__restrict__ in, long long n,
                                                       long long *__restrict__ block_sums) {
    __shared__ long long red[4];
    __shared__ int any_neg;
    if (threadIdx.x == 0) any_neg = 0;
    __syncthreads();
    const long long chunk = (long long)blockIdx.x * kScanChunk;
    long long s = 0;
    bool neg = false;
    const bool vec = (reinterpret_cast<unsigned long long>(in) & 15ull) == 0ull;
#pragma unroll
    for (int j = 0; j < kScanItems / 4; ++j) {
        const long long e = chunk + ((long long)j * 256 + threadIdx.x) * 4;
        if (vec && e + 3 < n) {
            const int4 q = *reinterpret_cast<const int4 *>(in + e);
            neg |= (q.x | q.y | q.z | q.w) < 0;
            s += (long long)q.x + q.y + q.z + q.w;
        } else {
            for (int t = 0; t < 4; ++t)
                if (e + t < n) {
                    const int v = in[e + t];
                    neg |= v < 0;
                    s += v;
                }
        }
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    if (neg) any_neg = 1;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = any_neg ? kScanNegative : (red[0] + red[1]) + (red[2] + red[3]);
}

// exclusive scan of the chunk sums by one workgroup: every thread takes a contiguous piece.  total_out[0] = the exact
// total, [1] = 1 if some input was negative; both are also posted to the host's mailbox slot when there is one (the
// host needs nothing else from the scan and does not wait for scan_apply).
__global__ __launch_bounds__(256) void scan_block_offsets(long long *block_sums, int nb, long long *total_out,
                                                          unsigned long long *mail_slot, unsigned long long mail_seq) {
    __shared__ long long piece[256];
    __shared__ int any_neg;
    if (threadIdx.x == 0) any_neg = 0;
    __syncthreads();
    const int per = (nb + 255) / 256;
    const int b0 = threadIdx.x * per, b1 = min(b0 + per, nb);
    long long s = 0;
    bool neg = false;
    for (int i = b0; i < b1; ++i) {
        const long long v = block_sums[i];
        neg |= v == kScanNegative;
        s += v == kScanNegative ? 0 : v;
    }
    piece[threadIdx.x] = s;
    if (neg) any_neg = 1;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long run = 0;
        for (int t = 0; t < 256; ++t) {
            const long long v = piece[t];
            piece[t] = run;
            run += v;
        }
        total_out[0] = run;
        total_out[1] = any_neg;
        if (mail_slot != nullptr) {
            const unsigned long long w[2] = {(unsigned long long)run, (unsigned long long)any_neg};
            mail_post(mail_slot, mail_seq, w, 2);
        }
    }
    __syncthreads();
    long long run = piece[threadIdx.x];
    for (int i = b0; i < b1; ++i) {
        const long long v = block_sums[i];
        block_sums[i] = run;
        run += v == kScanNegative ? 0 : v;
    }
}

__global__ __launch_bounds__(256) void scan_apply(const int *__restrict__ in, long long n,
                                                  const long long *__restrict__ block_offs,
                                                  const long long *__restrict__ total, int *__restrict__ out) {
    __shared__ int lds[5];
    const long long chunk = (long long)blockIdx.x * kScanChunk;
    const bool vec = ((reinterpret_cast<unsigned long long>(in) | reinterpret_cast<unsigned long long>(out)) & 15ull) == 0ull;
    int carry = (int)block_offs[blockIdx.x];
    // the chunk in slabs of 1024 entries: a thread takes 4 consecutive ones (one 16-byte load, one 16-byte store), the
    // slab's 256 partial sums are scanned over the workgroup, the slab total carries over to the next slab
#pragma unroll
    for (int j = 0; j < kScanItems / 4; ++j) {
        const long long e = chunk + ((long long)j * 256 + threadIdx.x) * 4;
        int v0 = 0, v1 = 0, v2 = 0, v3 = 0;
        const bool whole = vec && e + 3 < n;
        if (whole) {
            const int4 q = *reinterpret_cast<const int4 *>(in + e);
            v0 = q.x; v1 = q.y; v2 = q.z; v3 = q.w;
        } else {
            if (e < n) v0 = in[e];
            if (e + 1 < n) v1 = in[e + 1];
            if (e + 2 < n) v2 = in[e + 2];
            if (e + 3 < n) v3 = in[e + 3];
        }
        int slab_total = 0;
        const int run = carry + block_exclusive_scan_256(v0 + v1 + v2 + v3, lds, &slab_total);
        if (whole) {
            *reinterpret_cast<int4 *>(out + e) = make_int4(run, run + v0, run + v0 + v1, run + v0 + v1 + v2);
        } else {
            if (e < n) out[e] = run;
            if (e + 1 < n) out[e + 1] = run + v0;
            if (e + 2 < n) out[e + 2] = run + v0 + v1;
            if (e + 3 < n) out[e + 3] = run + v0 + v1 + v2;
        }
        carry += slab_total;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = (int)(*total);
}

// Short inputs (the coarse levels of a multigrid setup, lists of a few thousand rows: two dozen scans per setup): ONE workgroup
// of 1024 threads does all three steps -- every thread a contiguous piece, the pieces scanned over the workgroup -- in one
// launch instead of three that took 5 us each whatever the length.  Same results as the three kernels where no input is
// negative; with a negative input the flag is what the callers look at.
constexpr int kScanSmall = 32768;
__global__ __launch_bounds__(1024) void scan_small(const int *in, const int n, int *out,      // (in == out is allowed)
                                                   long long *__restrict__ total_out, unsigned long long *mail_slot,
                                                   unsigned long long mail_seq) {
    __shared__ long long wave_tot[16];
    __shared__ int any_neg;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t == 0) any_neg = 0;
    __syncthreads();
    const int per = (n + 1023) / 1024;
    const int b0 = min(t * per, n), b1 = min(b0 + per, n);
    long long s = 0;
    bool neg = false;
    for (int i = b0; i < b1; ++i) {
        const int v = in[i];
        neg |= v < 0;
        s += v;
    }
    if (neg) any_neg = 1;
    long long inc = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const long long u = __shfl_up(inc, off, 64);
        if (lane >= off) inc += u;
    }
    if (lane == 63) wave_tot[w] = inc;
    __syncthreads();
    long long run = inc - s, total = 0;
    for (int q = 0; q < 16; ++q) {
        if (q < w) run += wave_tot[q];
        total += wave_tot[q];
    }
    for (int i = b0; i < b1; ++i) {
        const int v = in[i];
        out[i] = (int)run;
        run += v;
    }
    if (t == 0) {
        const long long tot = any_neg ? 0 : total;
        out[n] = (int)tot;
        total_out[0] = tot;
        total_out[1] = any_neg;
        if (mail_slot != nullptr) {
            const unsigned long long wd[2] = {(unsigned long long)tot, (unsigned long long)any_neg};
            mail_post(mail_slot, mail_seq, wd, 2);
        }
    }
}

// Two halves: scan_i32_begin queues the three kernels (and the mailbox post of the total), scan_i32_end waits for the total
// -- whatever the caller launches in between (on any stream) is launched while the scan runs.
int scan_i32_begin(padne_ctx *ctx, const int32_t *in, int32_t *out, int64_t n, ScanTicket *t, bool want_total) {
    const int nb = (int)((n + kScanChunk - 1) / kScanChunk);
    hipStream_t s = ctx->stream;
    t->nb = nb;
    t->bs = nullptr;
    t->want_total = want_total;
    t->mail = MailTicket();
    if (nb == 0) {
        const hipError_t e = hipMemsetAsync(out, 0, sizeof(int32_t), s);
        if (e != hipSuccess) {
            set_error("scan failed: %s", hipGetErrorString(e));
            return PADNE_E_HIP;
        }
        return PADNE_OK;
    }
    long long *bs = (long long *)pool_alloc(ctx, sizeof(long long) * (size_t)(nb + 2));
    if (bs == nullptr) return PADNE_E_NOMEM;
    t->bs = bs;
    long long *tot = bs + nb;          // [0] exact 64-bit total, [1] negative-input flag
    if (want_total) t->mail = mail_ticket(ctx);
    if (n <= kScanSmall) {
        hipLaunchKernelGGL(scan_small, dim3(1), dim3(1024), 0, s, in, (int)n, out, tot, t->mail.slot_dev, t->mail.seq);
    } else {
        hipLaunchKernelGGL(scan_block_sums, dim3(nb), dim3(256), 0, s, in, (long long)n, bs);
        hipLaunchKernelGGL(scan_block_offsets, dim3(1), dim3(256), 0, s, bs, nb, tot, t->mail.slot_dev, t->mail.seq);
        // a total beyond int32 makes the 32-bit offsets below meaningless: it is detected from the 64-bit total
        hipLaunchKernelGGL(scan_apply, dim3(nb), dim3(256), 0, s, in, (long long)n, bs, tot, out);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        pool_free(ctx, bs);
        t->bs = nullptr;
        set_error("scan failed: %s", hipGetErrorString(e));
        return PADNE_E_HIP;
    }
    return PADNE_OK;
}

int scan_i32_end(padne_ctx *ctx, ScanTicket *t, long long h[2]) {
    h[0] = h[1] = 0;
    if (t->nb == 0) return PADNE_OK;
    hipStream_t s = ctx->stream;
    long long *tot = t->bs + t->nb;
    hipError_t e = hipSuccess;
    int rc = PADNE_OK;
    if (t->want_total) {
        if (t->mail.slot_dev != nullptr) {
            rc = mail_wait(ctx, t->mail, h, 2 * sizeof(long long));
        } else {
            e = hipMemcpyAsync(h, tot, 2 * sizeof(long long), hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
        }
    }
    pool_free(ctx, t->bs);      // reuse is ordered on the context's stream
    t->bs = nullptr;
    if (e != hipSuccess) {
        set_error("scan failed: %s", hipGetErrorString(e));
        return PADNE_E_HIP;
    }
    return rc;
}

static int scan_i32_impl(padne_ctx *ctx, const int32_t *in, int32_t *out, int64_t n, long long h[2], bool want_total) {
    ScanTicket t;
    PADNE_TRY(scan_i32_begin(ctx, in, out, n, &t, want_total));
    return scan_i32_end(ctx, &t, h);
}

int exclusive_scan_i32(padne_ctx *ctx, const int32_t *in, int32_t *out, int64_t n, int64_t *total) {
    long long h[2] = {0, 0};
    PADNE_TRY(scan_i32_impl(ctx, in, out, n, h, true));
    if (h[0] < 0 || h[1] != 0) {
        set_error("scan of negative counts");
        return PADNE_E_INVALID;
    }
    if (h[0] >= 2147483647LL) {
        // callers that can split their work (the sparse products of the multigrid setup) look at *total and do so
        *total = h[0];
        set_error("%lld entries exceed the 32-bit index space", h[0]);
        return PADNE_E_TOOLARGE;
    }
    *total = h[0];
    return PADNE_OK;
}

int exclusive_scan_i32_async(padne_ctx *ctx, const int32_t *in, int32_t *out, int64_t n) {
    long long h[2] = {0, 0};
    return scan_i32_impl(ctx, in, out, n, h, false);
}

int exclusive_scan_i32_flagged(padne_ctx *ctx, const int32_t *in, int32_t *out, int64_t n, int64_t *total, bool *negative) {
    long long h[2] = {0, 0};
    PADNE_TRY(scan_i32_impl(ctx, in, out, n, h, true));
    *negative = h[1] != 0;
    *total = h[0];
    if (!*negative && h[0] >= 2147483647LL) {
        set_error("%lld entries exceed the 32-bit index space", h[0]);
        return PADNE_E_TOOLARGE;
    }
    return PADNE_OK;
}

// ------------------------------------------------------------------------------------------
// assembly kernels
// ------------------------------------------------------------------------------------------
enum { ERR_BAD_INDEX = 0, ERR_NONMANIFOLD = 1, ERR_LONG_ROWS = 2, ERR_HUB = 4, ERR_SCAN = 5,
       ERR_GAVE_UP = 6,      // a bounded wait of asm_rows_in_place's in-kernel scan ran out: not an error of the input, the
                             // host builds the rows again in two passes (asm_rows_two_pass)
       ERR_WORDS = 8 };

__device__ __forceinline__ int find_segment(const long long *__restrict__ offs, int n_seg, long long i) {
    // largest m with offs[m] <= i   (offs has n_seg+1 entries, offs[0] = 0)
    int lo = 0, hi = n_seg;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (offs[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

// Vertex -> incident triangles, counted and listed in ONE pass: every vertex owns kIncCap list entries (a structured
// grid has 6 triangles around a vertex, Delaunay meshes rarely more than 10); the counter keeps counting beyond that,
// such a vertex takes the slot path.  A list entry is the triangle as the vertex sees it: (j, k), the corner the edge
// leaving the vertex points at and the corner the arriving edge comes from, as global vertex numbers -- the row kernel
// then goes from the list straight to the coordinates (with the triangle's number in the list it went list -> corners ->
// coordinates, three dependent gathers, and read every triangle three times).  The order inside a list is left to the
// atomics: nothing below depends on it.
constexpr int kIncCap = 12;
// The first kFanShort entries of all lists lie together, 64 bytes per vertex -- what almost every vertex needs and all the
// row kernel reads; the remaining entries of all lists follow behind them (touched for the few vertices that have more).
constexpr int kFanShort = 8;
__device__ __forceinline__ long long inc_at(const long long v, const int pos, const long long n_vert) {
    return pos < kFanShort ? v * kFanShort + pos : n_vert * kFanShort + v * (kIncCap - kFanShort) + (pos - kFanShort);
}
// The triangles of a workgroup usually touch a short range of vertices (512 consecutive triangles of a scan-line or
// strip-ordered mesh: two mesh lines, about as many distinct vertices as triangles, each touched three times).  Then
// the corners are counted with LDS atomics first and the global counter of a vertex is advanced ONCE per workgroup by
// the number of its corners here: a third of the global atomics.  A workgroup whose vertices span more than
// kCntRange indices (unordered triangle lists, very long mesh lines) does the same through a hash table of the vertices
// it meets.
constexpr int kCntRange = 4096;
struct __attribute__((packed, aligned(4))) Int3 { int a, b, c; };
constexpr int kCntTris = 2;           // triangles per thread (1: 441 us, 2: 417 us, 4: 443 us at N = 10 M): the dependent steps below (range,
                                      // LDS counts, global counters, list entries) are paid once per 512 triangles
__global__ __launch_bounds__(256) void asm_count_tri(long long n_tri, const int *__restrict__ tri, int n_mesh,
                                                     const long long *__restrict__ mesh_voff,
                                                     const long long *__restrict__ mesh_toff, int *__restrict__ cnt,
                                                     int2 *__restrict__ inc, int *__restrict__ err, const int no_range) {
    // no_range (PADNE_FORCE=asm_hash, tests): every workgroup takes the hash path of the wide ranges
    __shared__ int lcnt[kCntRange], lbase[kCntRange];
    __shared__ int s_min, s_max;
    if (threadIdx.x == 0) {
        s_min = 0x7fffffff;
        s_max = -1;
    }
    __syncthreads();
    // every XCD takes one contiguous eighth of the triangles (workgroups are dealt to the XCDs round robin): the two or
    // three workgroups that fill a vertex's list then share an L2, and the list leaves it once, complete
    const long long n_vert = mesh_voff[n_mesh];
    const unsigned per_xcd = gridDim.x >> 3;
    const unsigned wg = blockIdx.x >= (per_xcd << 3) ? blockIdx.x : (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    bool ok[kCntTris];
    int g[kCntTris][3];
    int lo = 0x7fffffff, hi = -1;
    // the mesh of the workgroup's first and last triangle: almost always the same one, then nobody searches the table
    const long long t_first = (long long)wg * kCntTris * 256;
    const long long t_last = min(t_first + kCntTris * 256, n_tri) - 1;
    const int m_first = find_segment(mesh_toff, n_mesh, t_first), m_last = find_segment(mesh_toff, n_mesh, t_last);
#pragma unroll
    for (int u = 0; u < kCntTris; ++u) {
        const long long t = t_first + u * 256 + threadIdx.x;
        ok[u] = false;
        g[u][0] = g[u][1] = g[u][2] = 0;
        if (t < n_tri) {
            const int m = m_first == m_last ? m_first : find_segment(mesh_toff, n_mesh, t);
            const long long v0 = mesh_voff[m];
            const long long nv = mesh_voff[m + 1] - v0;
            const Int3 c3 = *reinterpret_cast<const Int3 *>(tri + 3 * t);      // one 12-byte load
            const int a = c3.a, b = c3.b, c = c3.c;
            if (a < 0 || b < 0 || c < 0 || a >= nv || b >= nv || c >= nv || a == b || b == c || a == c) {
                atomicExch(&err[ERR_BAD_INDEX], 1);
            } else {
                ok[u] = true;
                g[u][0] = (int)(v0 + a);
                g[u][1] = (int)(v0 + b);
                g[u][2] = (int)(v0 + c);
                lo = min(lo, min(g[u][0], min(g[u][1], g[u][2])));
                hi = max(hi, max(g[u][0], max(g[u][1], g[u][2])));
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        lo = min(lo, __shfl_down(lo, off, 64));
        hi = max(hi, __shfl_down(hi, off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&s_min, lo);
        atomicMax(&s_max, hi);
    }
    __syncthreads();
    const int base = s_min;
    const long long range = (long long)s_max - base + 1;
    if (s_max < 0) return;                                // no valid triangle in this workgroup
    if (range > kCntRange || no_range) {
        // vertices spread over a wider range (unordered triangle lists; a mesh line longer than the table: 160 M vertices):
        // the same through a hash table of the vertices met -- at most 3 * 256 * kCntTris of them, kCntRange places
        static_assert(3 * 256 * kCntTris <= kCntRange / 2 && kCntRange == 4096, "hash of 12 bits, load at most 3/8");
        int slot[kCntTris][3], hpos[kCntTris][3];
        for (int j = threadIdx.x; j < kCntRange; j += 256) {
            lbase[j] = -1;                                 // (the keys, while the corners are counted)
            lcnt[j] = 0;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kCntTris; ++u)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                slot[u][q] = 0;
                hpos[u][q] = 0;
                if (ok[u]) {
                    unsigned h = ((unsigned)g[u][q] * 2654435761u) >> 20;
                    for (;;) {
                        h &= kCntRange - 1;
                        const int old = atomicCAS(&lbase[h], -1, g[u][q]);
                        if (old == -1 || old == g[u][q]) break;
                        ++h;
                    }
                    slot[u][q] = (int)h;
                    hpos[u][q] = atomicAdd(&lcnt[h], 1);
                }
            }
        __syncthreads();
        for (int j = threadIdx.x; j < kCntRange; j += 256) {
            const int c = lcnt[j];
            lcnt[j] = c > 0 ? atomicAdd(&cnt[lbase[j]], c) : 0;      // (in place: the counts are in the lanes' registers)
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kCntTris; ++u)
            if (ok[u])
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int pos = lcnt[slot[u][q]] + hpos[u][q];
                    if (pos < kIncCap) inc[inc_at(g[u][q], pos, n_vert)] = make_int2(g[u][(q + 1) % 3], g[u][(q + 2) % 3]);
                }
        return;
    }
    for (int j = threadIdx.x; j < (int)range; j += 256) lcnt[j] = 0;
    __syncthreads();
    int lpos[kCntTris][3];
#pragma unroll
    for (int u = 0; u < kCntTris; ++u)
#pragma unroll
        for (int q = 0; q < 3; ++q) lpos[u][q] = ok[u] ? atomicAdd(&lcnt[g[u][q] - base], 1) : 0;
    __syncthreads();
    for (int j = threadIdx.x; j < (int)range; j += 256) {
        const int c = lcnt[j];
        lbase[j] = c > 0 ? atomicAdd(&cnt[base + j], c) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kCntTris; ++u)
        if (ok[u])
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int pos = lbase[g[u][q] - base] + lpos[u][q];
                if (pos < kIncCap) inc[inc_at(g[u][q], pos, n_vert)] = make_int2(g[u][(q + 1) % 3], g[u][(q + 2) % 3]);
            }
}

// Which rows go through the slots.  A mesh vertex without stamps and with at most kFanShort triangles is built by
// asm_rows_in_place straight from its incidence list and takes no slots.  One with up to kIncCap triangles (its list is
// still complete) is built the same way by asm_rows_long_fans, into T + 2 slots.  Every other row (stamps, a hub of more
// than kIncCap triangles, internal nodes and extra unknowns behind the vertices) owns its diagonal placeholder, two slots
// per incident triangle and one per stamp, and is merged by the slot kernels.  Two lists.
__global__ __launch_bounds__(256) void asm_classify_rows(long long n, long long n_rows, long long n_vert,
                                                         const int *__restrict__ n_inc, const int *__restrict__ n_coo,
                                                         int *__restrict__ cnt, int *__restrict__ slow_list,
                                                         int *__restrict__ fan_list, int *__restrict__ n_listed /*[2]*/,
                                                         int *__restrict__ err) {
    // four rows per thread, 16-byte loads and stores (one row per thread: 83 us for 120 MB at N = 10 M).  The lists are
    // appended to once per workgroup and list: an agent-scope atomic on one address is served about every 12 ns, and an
    // unstructured mesh lists a few per cent of its rows
    __shared__ int s_part[2][4], s_base[2];
    const long long r0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    int T[4] = {0, 0, 0, 0}, S[4] = {0, 0, 0, 0}, C[4] = {0, 0, 0, 0};
    if (r0 + 3 < n_vert) {
        const int4 t4 = *reinterpret_cast<const int4 *>(n_inc + r0), s4 = *reinterpret_cast<const int4 *>(n_coo + r0);
        T[0] = t4.x; T[1] = t4.y; T[2] = t4.z; T[3] = t4.w;
        S[0] = s4.x; S[1] = s4.y; S[2] = s4.z; S[3] = s4.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            T[j] = r0 + j < n_vert ? n_inc[r0 + j] : 0;
            S[j] = r0 + j < n_rows ? n_coo[r0 + j] : 0;
        }
    }
    bool fan[4], slow[4];
    int n_slow = 0, n_fan = 0;
    bool hub = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long long r = r0 + j;
        fan[j] = r < n_vert && S[j] == 0 && T[j] > kFanShort && T[j] <= kIncCap;
        slow[j] = r < n_rows && !fan[j] && (r >= n_vert || S[j] != 0 || T[j] > kIncCap);
        C[j] = slow[j] ? 1 + 2 * T[j] + S[j] : fan[j] ? T[j] + 2 : 0;
        n_slow += slow[j] ? 1 : 0;
        n_fan += fan[j] ? 1 : 0;
        hub = hub || (slow[j] && T[j] > kIncCap);
    }
    if (r0 + 3 < n) {                                      // n = n_rows + 1: the scan wants the entry behind the end
        *reinterpret_cast<int4 *>(cnt + r0) = make_int4(C[0], C[1], C[2], C[3]);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (r0 + j < n) cnt[r0 + j] = C[j];
    }
    // places in the lists: inside the wave, among the waves, then one atomic per list
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int is = n_slow, jf = n_fan;
    for (int off = 1; off < 64; off <<= 1) {
        const int a = __shfl_up(is, off, 64), b = __shfl_up(jf, off, 64);
        if (lane >= off) {
            is += a;
            jf += b;
        }
    }
    if (lane == 63) {
        s_part[0][wv] = is;
        s_part[1][wv] = jf;
    }
    if (__ballot(hub) != 0ull && lane == 0) *(volatile int *)&err[ERR_HUB] = 1;
    __syncthreads();
    if (threadIdx.x < 2) {
        const int total = s_part[threadIdx.x][0] + s_part[threadIdx.x][1] + s_part[threadIdx.x][2] + s_part[threadIdx.x][3];
        s_base[threadIdx.x] = total > 0 ? atomicAdd(&n_listed[threadIdx.x], total) : 0;
    }
    __syncthreads();
    int ps = s_base[0] + is - n_slow, pf = s_base[1] + jf - n_fan;
    for (int w2 = 0; w2 < wv; ++w2) {
        ps += s_part[0][w2];
        pf += s_part[1][w2];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (slow[j]) slow_list[ps++] = (int)(r0 + j);
        if (fan[j]) fan_list[pf++] = (int)(r0 + j);
    }
}

__global__ void asm_count_coo(long long n_coo, const int *__restrict__ row, int *__restrict__ cnt) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_coo) return;
    atomicAdd(&cnt[row[k]], 1);
}

// |cot(theta_o)| / 2 for the edge (i,k) seen from the opposite corner o   -- mesh.py:136-138
__device__ __forceinline__ double cot_half(double ix, double iy, double kx, double ky, double ox, double oy) {
    const double vix = ix - ox, viy = iy - oy;
    const double vkx = kx - ox, vky = ky - oy;
    const double dot = vix * vkx + viy * vky;
    const double cross = vix * vky - viy * vkx;
    return fabs(dot / cross) / 2;
}

// slot key: column in the high word; low word orders duplicates: 0 = mesh term stored at the
// origin of the directed triangle edge, 1 = mesh term stored at its target, 2 = the row's
// diagonal placeholder, 3+k = stamp k
__device__ __forceinline__ long long make_key(int col, int seq) { return ((long long)col << 32) | (unsigned)seq; }

__global__ void asm_fill_tri(long long n_tri, const int *__restrict__ tri, const double *__restrict__ xy,
                             int n_mesh, const long long *__restrict__ mesh_voff,
                             const long long *__restrict__ mesh_toff, const int *__restrict__ slot_ptr,
                             int *__restrict__ cursor, long long *__restrict__ key, double *__restrict__ val,
                             const int *__restrict__ hubs_of) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tri) return;
    const int m = find_segment(mesh_toff, n_mesh, t);
    const long long v0 = mesh_voff[m];
    const int ga = (int)(v0 + tri[3 * t]), gb = (int)(v0 + tri[3 * t + 1]), gc = (int)(v0 + tri[3 * t + 2]);
    // hubs_of (the triangle counts of the vertices): only the rows of more than kIncCap triangles take slots here -- the
    // incidence list of such a vertex is incomplete; every other listed row is filled from its list (asm_fill_listed)
    const bool fa = hubs_of == nullptr || hubs_of[ga] > kIncCap;
    const bool fb = hubs_of == nullptr || hubs_of[gb] > kIncCap;
    const bool fc = hubs_of == nullptr || hubs_of[gc] > kIncCap;
    if (!(fa || fb || fc)) return;
    const double ax = xy[2 * (long long)ga], ay = xy[2 * (long long)ga + 1];
    const double bx = xy[2 * (long long)gb], by = xy[2 * (long long)gb + 1];
    const double cx = xy[2 * (long long)gc], cy = xy[2 * (long long)gc + 1];
    const double wab = cot_half(ax, ay, bx, by, cx, cy);   // edge a->b, opposite c
    const double wbc = cot_half(bx, by, cx, cy, ax, ay);   // edge b->c, opposite a
    const double wca = cot_half(cx, cy, ax, ay, bx, by);   // edge c->a, opposite b
    // row a: (a,b) forward, (a,c) backward ; row b: (b,c) fwd, (b,a) bwd ; row c: (c,a) fwd, (c,b) bwd
    int s;
    if (fa) {
        s = slot_ptr[ga] + atomicAdd(&cursor[ga], 2);
        key[s] = make_key(gb, 0); val[s] = wab;
        key[s + 1] = make_key(gc, 1); val[s + 1] = wca;
    }
    if (fb) {
        s = slot_ptr[gb] + atomicAdd(&cursor[gb], 2);
        key[s] = make_key(gc, 0); val[s] = wbc;
        key[s + 1] = make_key(ga, 1); val[s + 1] = wab;
    }
    if (fc) {
        s = slot_ptr[gc] + atomicAdd(&cursor[gc], 2);
        key[s] = make_key(ga, 0); val[s] = wca;
        key[s + 1] = make_key(gb, 1); val[s + 1] = wbc;
    }
}

// The listed rows of mesh vertices with a complete incidence list: their two cotangent terms per triangle go to fixed
// slots (1 + 2q, 2 + 2q; slot 0 is the diagonal placeholder), the stamps follow from the cursor.  The same cot_half calls
// with the same arguments as asm_fill_tri, which now only serves the hubs.
__global__ __launch_bounds__(256) void asm_fill_listed(const int *__restrict__ n_list, const int *__restrict__ row_list,
                                                       long long n_vert, const double *__restrict__ xy,
                                                       const int *__restrict__ n_inc, const int2 *__restrict__ inc,
                                                       const int *__restrict__ slot_ptr, int *__restrict__ cursor,
                                                       long long *__restrict__ key, double *__restrict__ val) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= *n_list) return;
    const int r = row_list[idx];
    const int T = r < n_vert ? n_inc[r] : 0;
    if (T > kIncCap) {                                     // a hub: asm_fill_tri appends its terms
        cursor[r] = 1;
        return;
    }
    cursor[r] = 1 + 2 * T;
    if (T == 0) return;
    const double2 pv = reinterpret_cast<const double2 *>(xy)[r];
    const int s0 = slot_ptr[r];
    for (int q = 0; q < T; ++q) {
        const int2 jk = inc[inc_at(r, q, n_vert)];
        const double2 pj = reinterpret_cast<const double2 *>(xy)[jk.x], pk = reinterpret_cast<const double2 *>(xy)[jk.y];
        key[s0 + 1 + 2 * q] = make_key(jk.x, 0);
        val[s0 + 1 + 2 * q] = cot_half(pv.x, pv.y, pj.x, pj.y, pk.x, pk.y);      // edge r -> j, opposite k
        key[s0 + 2 + 2 * q] = make_key(jk.y, 1);
        val[s0 + 2 + 2 * q] = cot_half(pk.x, pk.y, pv.x, pv.y, pj.x, pj.y);      // edge k -> r, opposite j
    }
}

// Offsets of the row tiles without a second pass.  A worker workgroup publishes the number of entries of a tile as soon as
// it has built the tile's rows (tile_agg), parks the rows, builds its NEXT tile, and only then asks for the first tile's
// offset -- by then it is almost always there; nobody waits for the slowest workgroup of the chip the way a look-back
// inside every workgroup does (that cost 0.6 ms of 1.4).  Two levels: tiles form chunks of 64; the worker that completes
// a chunk (a counter per chunk) adds up its 64 counts and publishes the chunk's; one scanner wave (block 0) walks the
// chunk counts in order, 64 chunks a look, and publishes the chunk offsets; a tile's offset is its chunk's plus the
// counts of the earlier tiles of the chunk (one 64-lane look).  Tiles are handed out by a ticket, so a tile somebody waits
// for is always in the hands of a running workgroup, and an offset depends on earlier tiles only.  Every wait is bounded
// (kScanMaxPolls); a wait that gives up raises the abort word, which ends all other waits and fails the assembly loudly.
constexpr unsigned long long kScanValueMask = (1ull << 62) - 1ull;
constexpr unsigned long long kScanKnown = 1ull << 62;
constexpr int kScanMaxPolls = 1 << 20;      // (about a second of polling: a wait that runs out costs a second pass, not the call)
constexpr int kChunkTiles = 64;
__device__ __forceinline__ unsigned long long scan_word_load(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void scan_word_store(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
struct TileScan {
    unsigned long long *tile_agg;        // [n_tiles]   known | entries of the tile
    unsigned long long *chunk_agg;       // [n_chunks]  known | entries of the chunk
    unsigned long long *chunk_prefix;    // [n_chunks + 1]  known | entries before the chunk
    int *chunk_count;                    // [n_chunks]  tiles of the chunk that have published
    int *ticket, *abort_word;
    long long *nnz_out;
    int n_tiles, n_chunks;
};

// block 0 of asm_rows_in_place, one wave
__device__ __forceinline__ void chunk_offset_scanner(const TileScan sc) {
    const int lane = threadIdx.x & 63;
    long long running = 0;
    int done = 0;                                          // chunk_prefix[0 .. done] are out
    if (lane == 0) scan_word_store(&sc.chunk_prefix[0], kScanKnown);
    int idle = 0;
    while (done < sc.n_chunks) {
        const int idx = done + lane;
        const unsigned long long w = idx < sc.n_chunks ? scan_word_load(&sc.chunk_agg[idx]) : 0ull;
        const unsigned long long gaps = __ballot((w >> 62) == 0ull);
        const int lead = gaps != 0ull ? __ffsll((long long)gaps) - 1 : 64;
        if (lead == 0) {
            // (the abort word is looked at now and then only: a word every waiting wave of the chip reads at every poll is a
            //  hot spot that slowed the whole kernel sevenfold)
            if (++idle > kScanMaxPolls || ((idle & 255) == 0 && __hip_atomic_load(sc.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                if (lane == 0) __hip_atomic_store(sc.abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            __builtin_amdgcn_s_sleep(2);
            continue;
        }
        idle = 0;
        long long incl = lane < lead ? (long long)(w & kScanValueMask) : 0;
        for (int off = 1; off < 64; off <<= 1) {
            const long long up = __shfl_up(incl, off, 64);
            if (lane >= off) incl += up;
        }
        if (lane < lead) scan_word_store(&sc.chunk_prefix[idx + 1], kScanKnown | (unsigned long long)(running + incl));
        running += __shfl(incl, 63, 64);
        done += lead;
    }
    if (lane == 0) *sc.nnz_out = running;
}

// a worker wave publishes a tile's count and arrives at the tile's chunk; returns how many tiles of the chunk have arrived
// (in lane 0; the caller looks at it later, the round trip of the counter travels behind other work)
__device__ __forceinline__ int tile_publish(const TileScan sc, const int tile, const int total, const int lane) {
    int arrived = 0;
    if (lane == 0) {
        scan_word_store(&sc.tile_agg[tile], kScanKnown | (unsigned long long)total);
        arrived = __hip_atomic_fetch_add(&sc.chunk_count[tile / kChunkTiles], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    }
    return arrived;
}
// ... and the wave that arrived last adds up the chunk.  Nothing relies on the order in which another wave's count and its
// arrival reach memory: the counts are read with their known bit, and waited for if need be.
__device__ __forceinline__ bool chunk_publish(const TileScan sc, const int tile, const int arrived_lane0, const int lane) {
    const int c = tile / kChunkTiles;
    const int in_chunk = min(kChunkTiles, sc.n_tiles - c * kChunkTiles);
    if (__shfl(arrived_lane0, 0, 64) != in_chunk) return true;
    unsigned long long w = lane < in_chunk ? scan_word_load(&sc.tile_agg[c * kChunkTiles + lane]) : kScanKnown;
    for (int polls = 0; __ballot((w >> 62) == 0ull) != 0ull; ++polls) {
        if (polls > kScanMaxPolls) {
            if (lane == 0) __hip_atomic_store(sc.abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
        if ((w >> 62) == 0ull) w = scan_word_load(&sc.tile_agg[c * kChunkTiles + lane]);
    }
    long long v = (long long)(w & kScanValueMask);
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) scan_word_store(&sc.chunk_agg[c], kScanKnown | (unsigned long long)v);
    return true;
}

// a worker wave asks for a tile's offset: the words are requested early (first look), examined after other work, and
// waited for only if they were not there yet; false when the wait was given up
__device__ __forceinline__ const unsigned long long *tile_offset_word(const TileScan sc, const int tile, const int lane, bool *mine) {
    const int c = tile / kChunkTiles, pos = tile % kChunkTiles;
    *mine = lane == 63 || lane < pos;                       // (pos <= 63: lane 63 never holds a tile of the sum)
    return lane == 63 ? &sc.chunk_prefix[c] : &sc.tile_agg[c * kChunkTiles + lane];
}
__device__ __forceinline__ bool tile_offset_collect(const TileScan sc, const unsigned long long *p, const bool mine,
                                                    unsigned long long w, const int lane, long long *before) {
    int polls = 0;
    while (__ballot(mine && (w >> 62) == 0ull) != 0ull) {
        if (++polls > kScanMaxPolls || ((polls & 255) == 0 && __hip_atomic_load(sc.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
            if (lane == 0) __hip_atomic_store(sc.abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(8);
        if (mine && (w >> 62) == 0ull) w = scan_word_load(p);
    }
    long long v = mine ? (long long)(w & kScanValueMask) : 0;
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    *before = v;
    return true;
}

// Compare-exchange network over N (column, value) pairs held in registers (merge exchange, Knuth 5.2.2 M: any N, about
// N log^2 N / 4 exchanges, the list is made at compile time so that every index below is a constant).
template <int N>
struct SortNet {
    int a[N * N], b[N * N], n;
    constexpr SortNet() : a(), b(), n(0) {
        int t = 0;
        while ((1 << t) < N) ++t;
        for (int p = t > 0 ? 1 << (t - 1) : 0; p > 0; p >>= 1) {
            int q = 1 << (t - 1), r = 0, d = p;
            for (;;) {
                for (int i = 0; i + d < N; ++i)
                    if ((i & p) == r) {
                        a[n] = i;
                        b[n] = i + d;
                        ++n;
                    }
                if (q == p) break;
                d = q - p;
                q >>= 1;
                r = p;
            }
        }
    }
};

// The row of mesh vertex r from its incidence list, in registers: the fan's triangles (j_u, k_u) with their two terms
// wf_u (edge r -> j_u) and wb_u (edge k_u -> r).  The neighbour j_u holds wf_u plus the backward term of the triangle on
// the other side of that edge (the u' with k_u' == j_u), found by comparing every pair -- no search loop, no LDS, no
// divergence; a boundary fan has one j without partner and one k without partner, which is the extra entry.  Sorted by
// column with a compare-exchange network.  The additions are those of the slot path (forward + backward; the diagonal
// -(w_1 + w_2 + ...) in ascending column order), hence the same bits.  RC: list entries examined (8 covers almost every
// vertex of a triangulation, 12 is the capacity of a list); the unused ones carry distinct negative numbers and match
// nothing.  Out: col/w[0..RC] sorted, INT_MAX behind the last neighbour; w already scaled by the conductance.
template <int RC>
__device__ __forceinline__ void fan_row(const int T, const double vx, const double vy, const int4 *__restrict__ pairs,
                                        const double *__restrict__ xy, const double sig, int (&col)[RC + 1], double (&w)[RC + 1],
                                        double &dval, int &len, bool &bad) {
    // pairs: the list, two entries per int4 (what lies behind entry T - 1 is never looked at)
    int cj[RC], ck[RC];
#pragma unroll
    for (int q = 0; q < RC / 2; ++q) {
        cj[2 * q] = pairs[q].x;
        ck[2 * q] = pairs[q].y;
        cj[2 * q + 1] = pairs[q].z;
        ck[2 * q + 1] = pairs[q].w;
    }
    double wf[RC], wb[RC];
#pragma unroll
    for (int u = 0; u < RC; ++u) {
        wf[u] = wb[u] = 0.0;
        if (u < T) {
            const double2 pj = reinterpret_cast<const double2 *>(xy)[cj[u]], pk = reinterpret_cast<const double2 *>(xy)[ck[u]];
            wf[u] = cot_half(vx, vy, pj.x, pj.y, pk.x, pk.y);      // edge r -> j, opposite k
            wb[u] = cot_half(pk.x, pk.y, vx, vy, pj.x, pj.y);      // edge k -> r, opposite j
        } else {
            cj[u] = -1 - u;
            ck[u] = -101 - u;
        }
    }
    // hit(a, b): triangle a arrives over the edge triangle b leaves by.  A manifold fan has at most one partner per edge on
    // either side; a count beyond one (two triangles on the same side of an edge) is the non-manifold mesh of mesh.py:342
    int src[RC];
#pragma unroll
    for (int u = 0; u < RC; ++u) src[u] = 0;
    int over = 0, fwd_only = 0, bwd_only = 0;
#pragma unroll
    for (int u = 0; u < RC; ++u) {
        int partners = 0;
        double wbm = 0.0;
#pragma unroll
        for (int u2 = 0; u2 < RC; ++u2) {
            const bool hit = ck[u2] == cj[u];
            partners += hit ? 1 : 0;
            src[u2] += hit ? 1 : 0;
            wbm = hit ? wb[u2] : wbm;
        }
        over |= partners;
        col[u] = u < T ? cj[u] : 0x7fffffff;
        w[u] = wf[u] + wbm;                                 // forward + backward, as the merge adds them (w + 0.0 == w: w >= +0)
        fwd_only += (u < T && partners == 0) ? 1 : 0;
    }
    int extra_col = 0x7fffffff;
    double extra_w = 0.0;
#pragma unroll
    for (int u = 0; u < RC; ++u) {
        over |= src[u];
        const bool alone = u < T && src[u] == 0;            // nobody leaves by the edge k_u - r: the boundary on that side
        bwd_only += alone ? 1 : 0;
        extra_col = alone ? ck[u] : extra_col;
        extra_w = alone ? wb[u] : extra_w;
    }
    col[RC] = extra_col;
    w[RC] = extra_w;
    if ((over & ~1) != 0 || fwd_only > 1 || bwd_only > 1) bad = true;      // (more than one boundary on a side: not a manifold fan)
    constexpr SortNet<RC + 1> net;
#pragma unroll
    for (int e = 0; e < net.n; ++e) {
        const int ia = net.a[e], ib = net.b[e];
        const bool swap = col[ia] > col[ib];
        const int ca = col[ia], cb = col[ib];
        const double wa = w[ia], wbb = w[ib];
        col[ia] = swap ? cb : ca;
        col[ib] = swap ? ca : cb;
        w[ia] = swap ? wbb : wa;
        w[ib] = swap ? wa : wbb;
    }
    double dacc = 0.0;
    len = 0;
#pragma unroll
    for (int i = 0; i <= RC; ++i) {
        if (col[i] != 0x7fffffff) {
            const double wm = w[i];
            if (wm != 0.0) dacc = dacc - wm;
            const double v = sig * wm;
            w[i] = v;                                       // what the row stores
            if (v != 0.0) ++len;                            // exact zeros are not stored
        }
    }
    dval = sig * dacc;
    if (dval != 0.0) ++len;
}

__device__ __forceinline__ void put_col(int *p, int c) { *p = c; }
__device__ __forceinline__ void put_col(long long *p, int c) { *p = (long long)c << 32; }      // a slot key: column in the high word

template <int RC, typename CT>
__device__ __forceinline__ void fan_row_store(const int r, const int (&col)[RC + 1], const double (&w)[RC + 1], const double dval,
                                              CT *__restrict__ cols, double *__restrict__ vals) {
    int o = 0;
    bool diag_done = false;
#pragma unroll
    for (int i = 0; i <= RC + 1; ++i) {
        const int c = i <= RC ? col[i <= RC ? i : RC] : 0x7fffffff;
        if (!diag_done && c > r) {                          // (INT_MAX behind the last neighbour)
            if (dval != 0.0) {
                put_col(cols + o, r);
                vals[o] = dval;
                ++o;
            }
            diag_done = true;
        }
        if (i <= RC && c != 0x7fffffff) {
            const double v = w[i <= RC ? i : RC];
            if (v != 0.0) {
                put_col(cols + o, c);
                vals[o] = v;
                ++o;
            }
        }
    }
}

// Vertices with more than kFanShort triangles and no stamps (a few per cent of a Delaunay mesh, none of a structured
// one): the same register algorithm over the whole list, the row goes to the vertex's slots -- finished, nothing left to
// merge -- and is placed with the listed rows.  Keeps the long variant's registers out of the kernel every row runs.
__global__ __launch_bounds__(128) void asm_rows_long_fans(const int *__restrict__ n_list, const int *__restrict__ row_list,
                                                          long long n_vert, int n_mesh, const long long *__restrict__ mesh_voff,
                                                          const double *__restrict__ sigma, const double *__restrict__ xy,
                                                          const int *__restrict__ n_inc, const int2 *__restrict__ inc,
                                                          const int *__restrict__ slot_ptr, long long *__restrict__ key,
                                                          double *__restrict__ val, int *__restrict__ row_len,
                                                          int *__restrict__ err) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= *n_list) return;
    const int r = row_list[idx];
    const int T = n_inc[r];
    int4 pairs[kIncCap / 2];
    const int4 *lst = reinterpret_cast<const int4 *>(inc + inc_at(r, 0, n_vert)), *more = reinterpret_cast<const int4 *>(inc + inc_at(r, kFanShort, n_vert));
#pragma unroll
    for (int q = 0; q < kFanShort / 2; ++q) pairs[q] = lst[q];
#pragma unroll
    for (int q = kFanShort / 2; q < kIncCap / 2; ++q) pairs[q] = more[q - kFanShort / 2];
    const double2 pv = reinterpret_cast<const double2 *>(xy)[r];
    const double sig = sigma[find_segment(mesh_voff, n_mesh, r)];
    int col[kIncCap + 1];
    double w[kIncCap + 1];
    double dval = 0.0;
    int len = 0;
    bool bad = false;
    fan_row<kIncCap>(T, pv.x, pv.y, pairs, xy, sig, col, w, dval, len, bad);
    if (bad) atomicExch(&err[ERR_NONMANIFOLD], 1);
    const int s0 = slot_ptr[r];
    fan_row_store<kIncCap>(r, col, w, dval, key + s0, val + s0);
    row_len[r] = len;
}

// The rows of the assembled system, written once and in place.  Worker workgroups take tiles of 128 rows by ticket; the
// lane of mesh vertex r builds its row from the incidence list (fan_row) and counts what it will store; the tile's count
// is published, the rows wait in registers while the workgroup builds its next tile, then learn their offset from the
// scanner (above) and are written straight into the CSR arrays -- no slots, no second scan, no compaction pass.  The
// listed rows (stamps, hubs, long fans, the unknowns behind the vertices) were merged at their slot offsets before this
// kernel runs; it leaves room for them, asm_place_listed moves them in.
// Tickets.  One counter for all workgroups limits the kernel: an agent-scope atomic on ONE address is served about every
// 12 ns.  kTicketSeqs counters, each on a line of its own, hand out interleaved sequences of tiles (sequence q owns the
// tiles q, q + kTicketSeqs, ...), a workgroup draws from the sequence of its number: the sequences advance together, so
// the tiles still start nearly in order, which is all the scan needs to be quick.
// What its CORRECTNESS needs: a workgroup must never wait for an offset that depends on a tile it holds itself and has not
// published.  While it waits for the offset of tile H it has published everything it built; what it may still hold is a
// ticket drawn ahead of time.  Drawn from its own sequence that ticket is larger than H (a sequence only grows) and H's
// offset does not depend on it.  When the own sequence is used up the workgroup draws from the others -- those tickets
// can be smaller than H -- and then only at the top of a turn, when everything it holds is published (kLateTicket marks
// a turn whose ticket is drawn that way).  (Tickets drawn ahead from all sequences in turn made six assemblies side by
// side wait for themselves until the poll limit.)
constexpr int kTicketSeqs = 32, kTicketStride = 16, kLateTicket = -1;
__device__ __forceinline__ int take_own_ticket(int *__restrict__ counters, const int seq, const int n_tiles) {
    const int k = atomicAdd(&counters[seq * kTicketStride], 1);
    const long long tile = (long long)k * kTicketSeqs + seq;
    return tile < n_tiles ? (int)tile : kLateTicket;
}
__device__ __forceinline__ int take_any_ticket(int *__restrict__ counters, int &seq, unsigned &used_up, const int n_tiles) {
    // n_tiles when every sequence is used up
    static_assert(kTicketSeqs == 32, "one bit per sequence in used_up");
    while (used_up != 0xffffffffu) {
        const int q = seq;
        seq = (seq + 1) % kTicketSeqs;
        if ((used_up >> q) & 1u) continue;
        const int k = atomicAdd(&counters[q * kTicketStride], 1);
        const long long tile = (long long)k * kTicketSeqs + q;
        if (tile < n_tiles) return (int)tile;
        used_up |= 1u << q;
    }
    return n_tiles;
}

struct RowsInPlace {
    long long n_rows, n_vert;
    int n_mesh;
    const long long *mesh_voff;
    const double *sigma, *xy;
    const int *n_inc;
    const int2 *inc;
    const int *slot_ptr, *row_len;
    int *rowptr, *cols;
    double *vals;
    long long nnz_cap;              // entries cols / vals hold: an offset beyond it is refused (and reported), never written to
    int *err;
    TileScan scan;
};

__global__ __launch_bounds__(128) void asm_rows_in_place(const RowsInPlace a) {
    static_assert(kFanShort == 8, "the first 64 bytes of a list are its first eight entries");
    if (blockIdx.x == 0) {
        if (threadIdx.x < 64) chunk_offset_scanner(a.scan);
        return;
    }
    // the rows this kernel built of the tile that waits for its offset, one behind the other.  A tile without listed rows
    // (four in five) lies in the matrix exactly like that, and the whole workgroup copies it out with consecutive lanes on
    // consecutive entries (every lane storing its own row, 28 and 56 bytes from its neighbour's, cost 275 us of 1030);
    // in a tile with listed rows every lane moves its own row past the room of the listed ones.
    constexpr int kStage = 128 * (kFanShort + 2);           // a row of this kernel holds at most kFanShort + 1 neighbours and the diagonal
    __shared__ int stage_c[kStage];
    __shared__ double stage_v[kStage];
    __shared__ int s_ticket[2], s_late, s_wave_total[2][2], s_wave_built[2][2], s_abort;      // (tickets and wave totals alternate between two sets by turn)
    __shared__ int s_gave_up[2];                            // a wait given up at the END of a turn, by turn parity: read behind the next turn's barrier
    __shared__ long long s_before;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int n_tiles = a.scan.n_tiles;
    // the slow words of the protocol (agent-scope atomics and loads, a few us each) are asked for early and looked at late,
    // and the two waves share them: wave 1 fetches the tickets, wave 0 publishes and collects the offsets
    const int own_seq = blockIdx.x % kTicketSeqs;
    int any_seq = (own_seq + 1) % kTicketSeqs;              // (thread 64's) where the late tickets come from next
    unsigned used_up = 1u << own_seq;                       // (thread 64's) sequences found used up (the own one is never asked late)
    int ticket_ahead = kLateTicket;                         // (thread 64's) the ticket of the turn after this one
    if (t == 64) {
        s_ticket[0] = take_own_ticket(a.scan.ticket, own_seq, n_tiles);
        s_abort = 0;
        s_gave_up[0] = s_gave_up[1] = 0;
        ticket_ahead = s_ticket[0] != kLateTicket ? take_own_ticket(a.scan.ticket, own_seq, n_tiles) : kLateTicket;
    }
    __syncthreads();
    int tile = s_ticket[0];
    bool held = false, held_direct = false;
    int held_tile = 0, held_len = 0, held_local = 0, held_total = 0, held_blocal = 0, held_built = 0;
    for (int turn = 0;; ++turn) {
        if (tile == kLateTicket) {                          // (uniform) the own sequence is used up: see above
            if (t == 64) s_late = take_any_ticket(a.scan.ticket, any_seq, used_up, n_tiles);
            __syncthreads();
            tile = s_late;
            __syncthreads();                                // (s_late is rewritten at the top of a later turn)
        }
        const bool work = tile < n_tiles;
        if (!work && !held) break;
        // first look at the offset of the tile that waits: asked for now, examined after this tile's rows are built
        bool off_mine = false;
        const unsigned long long *off_p = nullptr;
        unsigned long long off_w = kScanKnown;
        if (held && wv == 0) {
            off_p = tile_offset_word(a.scan, held_tile, lane, &off_mine);
            if (off_mine) off_w = scan_word_load(off_p);
        }
        bool direct = false;
        int len = 0, local = 0, blocal = 0;                  // (blocal: offset among the rows this kernel builds)
        int col[kFanShort + 1];
        double w[kFanShort + 1], dval = 0.0;
#pragma unroll
        for (int i = 0; i <= kFanShort; ++i) {
            col[i] = 0x7fffffff;
            w[i] = 0.0;
        }
        if (work) {
            const long long r = (long long)tile * 128 + t;
            const bool live = r < a.n_rows;
            // everything that does not depend on another load is requested at once: the row's slot span, its triangle count,
            // its own coordinates and the first eight list entries (whether they are needed or not)
            int sp0 = 0, sp1 = 1, T = 0, listed_len = 0;
            int4 pairs[kFanShort / 2];
            double2 pv = make_double2(0.0, 0.0);
#pragma unroll
            for (int q = 0; q < kFanShort / 2; ++q) pairs[q] = make_int4(0, 0, 0, 0);
            if (live) {
                sp0 = a.slot_ptr[r];
                sp1 = a.slot_ptr[r + 1];
                listed_len = a.row_len[r];
                if (r < a.n_vert) {
                    T = a.n_inc[r];
                    pv = reinterpret_cast<const double2 *>(a.xy)[r];
                    const int4 *lst = reinterpret_cast<const int4 *>(a.inc + r * kFanShort);
#pragma unroll
                    for (int q = 0; q < kFanShort / 2; ++q) pairs[q] = lst[q];
                }
            }
            direct = live && sp1 == sp0;                   // a listed row owns at least its diagonal placeholder
            len = live && !direct ? listed_len : 0;
            if (len < 0) {                                 // a listed row no merge pass reached: never expected
                atomicExch(&a.err[ERR_SCAN], 1);
                len = 0;
            }
            if (!direct) T = 0;
            const bool six = __ballot(T > 6) == 0ull;      // (one code path per wave; a structured mesh never needs the other)
            bool bad = false;
            if (direct) {
                const double sig = a.sigma[find_segment(a.mesh_voff, a.n_mesh, r)];
                if (six) {
                    int c6[7];
                    double w6[7];
                    fan_row<6>(T, pv.x, pv.y, pairs, a.xy, sig, c6, w6, dval, len, bad);
#pragma unroll
                    for (int i = 0; i < 7; ++i) {
                        col[i] = c6[i];
                        w[i] = w6[i];
                    }
                } else {
                    fan_row<kFanShort>(T, pv.x, pv.y, pairs, a.xy, sig, col, w, dval, len, bad);
                }
                if (bad) atomicExch(&a.err[ERR_NONMANIFOLD], 1);
            }
            int incl = len;                                // offsets inside the wave
            for (int off = 1; off < 64; off <<= 1) {
                const int up = __shfl_up(incl, off, 64);
                if (lane >= off) incl += up;
            }
            if (lane == 63) s_wave_total[turn & 1][wv] = incl;
            local = incl - len;
            const int blen = direct ? len : 0;
            int bincl = blen;
            for (int off = 1; off < 64; off <<= 1) {
                const int up = __shfl_up(bincl, off, 64);
                if (lane >= off) bincl += up;
            }
            if (lane == 63) s_wave_built[turn & 1][wv] = bincl;
            blocal = bincl - blen;
        }
        if (t == 64) s_ticket[(turn + 1) & 1] = ticket_ahead;      // (asked for at the end of the previous turn)
        __syncthreads();                                   // wave totals, the next ticket
        if (turn > 0 && s_gave_up[(turn - 1) & 1]) {        // (uniform: written before this barrier, and the word of THIS
            if (t == 0) atomicExch(&a.err[ERR_GAVE_UP], 1);   //  turn's parity is not written before the next one)
            return;
        }
        const int tile_after = work ? s_ticket[(turn + 1) & 1] : tile;
        int arrived = 0;
        if (work) {
            if (wv == 1) local += s_wave_total[turn & 1][0];
            if (wv == 1) blocal += s_wave_built[turn & 1][0];
            if (wv == 0) arrived = tile_publish(a.scan, tile, s_wave_total[turn & 1][0] + s_wave_total[turn & 1][1], lane);
        }
        bool chunk_done = !work;
        if (held) {
            if (wv == 0) {
                // Before this wave WAITS for anything it does its own duty: if it arrived last at its tile's chunk, the chunk's
                // count is what the offset it is about to wait for may depend on (at the end of the tiles a workgroup can hold
                // a tile of a later chunk than the one it has just built).  When the first look found everything, nothing waits.
                bool ok = true;
                if (!chunk_done && __ballot(off_mine && (off_w >> 62) == 0ull) != 0ull) {
                    ok = chunk_publish(a.scan, tile, arrived, lane);
                    chunk_done = true;
                }
                long long before = 0;
                ok = ok && tile_offset_collect(a.scan, off_p, off_mine, off_w, lane, &before);
                if (lane == 0) {
                    s_before = before;
                    if (!ok) s_abort = 1;
                }
            }
            __syncthreads();
            if (s_abort) {                                  // (uniform: read behind the barrier by both waves)
                if (t == 0) atomicExch(&a.err[ERR_GAVE_UP], 1);
                return;
            }
            const long long hr = (long long)held_tile * 128 + t;
            const long long base = s_before;
            if (base < 0 || base + held_total > a.nnz_cap) {      // refused and reported, never written
                if (t == 0) atomicExch(&a.err[ERR_SCAN], 1);
            } else {
                if (hr < a.n_rows) {
                    a.rowptr[hr] = (int)(base + held_local);
                    if (hr == a.n_rows - 1) a.rowptr[a.n_rows] = (int)(base + held_local + held_len);
                }
                if (held_built == held_total) {
                    for (int i = t; i < held_total; i += 128) {
                        a.cols[base + i] = stage_c[i];
                        a.vals[base + i] = stage_v[i];
                    }
                } else if (held_direct) {
                    for (int i = 0; i < held_len; ++i) {
                        a.cols[base + held_local + i] = stage_c[held_blocal + i];
                        a.vals[base + held_local + i] = stage_v[held_blocal + i];
                    }
                }
            }
            __syncthreads();                               // the stage is rewritten below
        }
        if (!chunk_done && wv == 0 && !chunk_publish(a.scan, tile, arrived, lane)) {
            // the wait was given up: both waves leave TOGETHER, behind the next barrier they share (the one in front of
            // the publish of the next turn), never one wave alone in front of a barrier the other still goes to
            if (lane == 0) s_gave_up[turn & 1] = 1;
        }
        // the ticket after the next: ahead of time from the own sequence only (larger than everything held); once that is
        // used up every ticket is drawn late.  Nothing is drawn behind the end.
        if (t == 64)
            ticket_ahead = !work ? n_tiles : (tile_after == kLateTicket || tile_after >= n_tiles) ? tile_after
                                               : take_own_ticket(a.scan.ticket, own_seq, n_tiles);
        held = work;
        held_tile = tile;
        held_len = len;
        held_local = local;
        held_direct = direct;
        held_blocal = blocal;
        held_total = work ? s_wave_total[turn & 1][0] + s_wave_total[turn & 1][1] : 0;
        held_built = work ? s_wave_built[turn & 1][0] + s_wave_built[turn & 1][1] : 0;
        if (work && direct) fan_row_store<kFanShort>((int)((long long)tile * 128 + t), col, w, dval, stage_c + blocal, stage_v + blocal);
        tile = tile_after;
    }
    __syncthreads();                                       // (the loop is left by both waves in the same turn)
    if ((s_gave_up[0] | s_gave_up[1]) != 0 && t == 0) atomicExch(&a.err[ERR_GAVE_UP], 1);
}

// The second path of the row kernel: the same rows in two passes, for when the in-kernel scan gives up (a chip shared
// with other work can starve its scanner or a ticket holder until a bounded wait runs out -- that is a property of the
// moment, not of the input).  COUNT leaves every row's length (a listed row's is its merged length), an ordinary
// exclusive scan turns the lengths into the row pointer, FILL builds the rows again -- the same fan_row, hence the same
// bits -- and every lane stores its row at its offset.  Twice the arithmetic and uncoalesced stores: about 3x the time of
// the single pass, which is why it is the second path and not the first.
template <bool FILL>
__global__ __launch_bounds__(128) void asm_rows_two_pass(const RowsInPlace a, int *__restrict__ len_out) {
    const long long r = (long long)blockIdx.x * 128 + threadIdx.x;
    if (r >= a.n_rows) return;
    const bool direct = a.slot_ptr[r + 1] == a.slot_ptr[r];      // a listed row owns at least its diagonal placeholder
    if (!direct) {
        if (!FILL) {
            int len = a.row_len[r];
            if (len < 0) {                                 // a listed row no merge pass reached: never expected
                atomicExch(&a.err[ERR_SCAN], 1);
                len = 0;
            }
            len_out[r] = len;
        }
        return;
    }
    int T = 0;
    int4 pairs[kFanShort / 2];
    double2 pv = make_double2(0.0, 0.0);
    double sig = 0.0;
#pragma unroll
    for (int q = 0; q < kFanShort / 2; ++q) pairs[q] = make_int4(0, 0, 0, 0);
    if (r < a.n_vert) {
        T = a.n_inc[r];
        pv = reinterpret_cast<const double2 *>(a.xy)[r];
        const int4 *lst = reinterpret_cast<const int4 *>(a.inc + r * kFanShort);
#pragma unroll
        for (int q = 0; q < kFanShort / 2; ++q) pairs[q] = lst[q];
        sig = a.sigma[find_segment(a.mesh_voff, a.n_mesh, r)];
    }
    int col[kFanShort + 1], len = 0;
    double w[kFanShort + 1], dval = 0.0;
    bool bad = false;
    fan_row<kFanShort>(T, pv.x, pv.y, pairs, a.xy, sig, col, w, dval, len, bad);
    if (bad) atomicExch(&a.err[ERR_NONMANIFOLD], 1);
    if (!FILL) {
        len_out[r] = len;
        return;
    }
    const long long at = a.rowptr[r];
    if (at < 0 || at + len > a.nnz_cap || a.rowptr[r + 1] - at != len) {      // refused and reported, never written
        atomicExch(&a.err[ERR_SCAN], 1);
        return;
    }
    fan_row_store<kFanShort>((int)r, col, w, dval, a.cols + at, a.vals + at);
}

// the listed rows, merged at their slot offsets, move to the place asm_rows_in_place left for them: one wave per row
__global__ __launch_bounds__(256) void asm_place_listed(const int *__restrict__ n_list, const int *__restrict__ row_list,
                                                        const int *__restrict__ slot_ptr, const long long *__restrict__ key,
                                                        const double *__restrict__ val, const int *__restrict__ rowptr,
                                                        int *__restrict__ cols, double *__restrict__ vals, const long long nnz_cap,
                                                        int *__restrict__ err) {
    const int lane = threadIdx.x & 63;
    const int n = *n_list;
    for (int idx = blockIdx.x * 4 + (threadIdx.x >> 6); idx < n; idx += gridDim.x * 4) {
        const int r = row_list[idx];
        const int s0 = slot_ptr[r], at = rowptr[r], len = rowptr[r + 1] - at;
        if (at < 0 || len < 0 || len > slot_ptr[r + 1] - s0 || (long long)at + len > nnz_cap) {      // refused and reported, never written
            if (lane == 0) atomicExch(&err[ERR_SCAN], 1);
            continue;
        }
        for (int i = lane; i < len; i += 64) {
            cols[at + i] = (int)(key[s0 + i] >> 32);
            vals[at + i] = val[s0 + i];
        }
    }
}

__global__ void asm_fill_coo(long long n_coo, const int *__restrict__ row, const int *__restrict__ col,
                             const double *__restrict__ v, const int *__restrict__ slot_ptr,
                             int *__restrict__ cursor, long long *__restrict__ key, double *__restrict__ val) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_coo) return;
    const int r = row[k];
    const int s = slot_ptr[r] + atomicAdd(&cursor[r], 1);
    key[s] = make_key(col[k], (int)(k + 3));
    val[s] = v[k];
}

// in-place insertion sort of a row's slots by key
__device__ __forceinline__ void sort_slots(long long *key, double *val, int n) {
    for (int i = 1; i < n; ++i) {
        const long long k = key[i];
        const double v = val[i];
        int j = i - 1;
        while (j >= 0 && key[j] > k) {
            key[j + 1] = key[j];
            val[j + 1] = val[j];
            --j;
        }
        key[j + 1] = k;
        val[j + 1] = v;
    }
}

// Long rows of the slot path (a vertex that a via ring's resistors all snapped to, a hub of lumped elements): the
// one-lane insertion sort of merge_rows costs O(n^2) dependent global accesses (0.9 ms for the few hundred 70-slot rows
// of config C4).  One wave per such row sorts its slots first: keys and values into LDS, every lane ranks its elements
// by counting smaller keys (ties by position: keys are unique for a manifold mesh, repeated ones must keep their
// order), sorted slots back in place.  merge_rows then finds the row sorted -- its insertion sort degenerates to one
// pass -- and walks it as before: same operations in the same order.
constexpr int kWaveSortCap = 1024;
template <int CAPW, int CHUNK = 64>      // slots per wave in LDS (1024 for the long rows, 64 when every row of a transpose is sorted
                                          // this way); rows a wave examines per turn (few when most of them qualify)
__global__ __launch_bounds__(256) void sort_long_rows_wave(long long n_list, const int *__restrict__ row_list,
                                                           const int *__restrict__ slot_ptr, long long *__restrict__ key,
                                                           double *__restrict__ val, const int min_len, const int mesh,
                                                           int *__restrict__ longer_count = nullptr,
                                                           const int *__restrict__ run_if_nonzero = nullptr,
                                                           int *__restrict__ longer_list = nullptr) {
    // longer_count / longer_list: the pass counts and lists the rows that exceed its capacity (left to the next pass);
    // run_if_nonzero: a pass behind such a count returns at once when there is nothing for it to do
    __shared__ long long Ks[4][CAPW];
    __shared__ double Vs[4][CAPW];
    if (run_if_nonzero != nullptr && *run_if_nonzero == 0) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long n_waves = (long long)gridDim.x * 4;
    // a wave looks at 64 rows at a time (one lane each, coalesced reads of the offsets) and sorts the long ones it finds
    for (long long c0 = ((long long)blockIdx.x * 4 + w) * CHUNK; c0 < n_list; c0 += n_waves * CHUNK) {
        const long long idx = c0 + lane;
        long long r_l = 0;
        int s0_l = 0, n_l = 0;
        if (lane < CHUNK && idx < n_list) {
            r_l = row_list != nullptr ? row_list[idx] : idx;
            s0_l = slot_ptr[r_l];
            n_l = slot_ptr[r_l + 1] - s0_l;
        }
        unsigned long long todo = __ballot(n_l >= min_len && n_l <= CAPW);
        if (longer_count != nullptr) {
            const unsigned long long longer = __ballot(n_l > CAPW);
            if (longer != 0ull) {
                int base = 0;
                if (lane == 0) base = atomicAdd(longer_count, __popcll(longer));
                base = __shfl(base, 0, 64);
                if (longer_list != nullptr && n_l > CAPW)
                    longer_list[base + __popcll(longer & ((1ull << lane) - 1ull))] = (int)r_l;
            }
        }
        while (todo != 0ull) {
            const int b = __ffsll((long long)todo) - 1;
            todo &= todo - 1ull;
            const long long r = __shfl(r_l, b, 64);
            const int s0 = __shfl(s0_l, b, 64), n = __shfl(n_l, b, 64);
            for (int e = lane; e < n; e += 64) {
                // slot 0 of an assembly row is its diagonal placeholder, which nobody wrote (see merge_rows)
                Ks[w][e] = (mesh && e == 0) ? make_key((int)r, 2) : key[s0 + e];
                Vs[w][e] = (mesh && e == 0) ? 0.0 : val[s0 + e];
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            for (int e = lane; e < n; e += 64) {
                const long long k = Ks[w][e];
                int rank = 0;
                for (int f = 0; f < n; ++f) {
                    const long long kf = Ks[w][f];
                    rank += (kf < k || (kf == k && f < e)) ? 1 : 0;
                }
                key[s0 + rank] = k;
                val[s0 + rank] = Vs[w][e];
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // LDS is reused by the next row
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// The listed rows, one WORKGROUP each (rows of up to 1024 slots): keys and values into LDS, every thread ranks its elements
// by counting smaller keys, sorted slots back in place -- the rank sort of sort_long_rows_wave over 256 threads.  The long
// rows of a transposed prolongator (aggregates next to a via ring: several hundred fine rows interpolate from them) are a
// few dozen; a wave that met them one after the other in its chunk took 1.2 ms, spread over workgroups they take 60 us.
__global__ __launch_bounds__(256) void sort_listed_rows_block(const int *__restrict__ n_list, const int *__restrict__ row_list,
                                                              const int *__restrict__ slot_ptr, long long *__restrict__ key,
                                                              double *__restrict__ val) {
    __shared__ long long Ks[kWaveSortCap];
    __shared__ double Vs[kWaveSortCap];
    const int cnt = *n_list;
    for (int j = blockIdx.x; j < cnt; j += gridDim.x) {
        const int r = row_list[j];
        const int s0 = slot_ptr[r], n = slot_ptr[r + 1] - s0;
        if (n > kWaveSortCap) continue;               // beyond the LDS: the one-lane merge sorts it
        for (int e = threadIdx.x; e < n; e += 256) {
            Ks[e] = key[s0 + e];
            Vs[e] = val[s0 + e];
        }
        __syncthreads();
        for (int e = threadIdx.x; e < n; e += 256) {
            const long long k = Ks[e];
            int rank = 0;
            for (int f = 0; f < n; ++f) {
                const long long kf = Ks[f];
                rank += (kf < k || (kf == k && f < e)) ? 1 : 0;
            }
            key[s0 + rank] = k;
            val[s0 + rank] = Vs[e];
        }
        __syncthreads();
    }
}

// One lane per row.  MESH=true: slots with sequence 0/1 are raw cotangent terms of the row's
// mesh, sequence 2 is the (zero valued) diagonal placeholder every row owns, 3+k is stamp k.
// Every column group owns at least one slot and emits at most one entry, so the in-place
// compaction (output cursor o <= input cursor i) never overtakes unread slots.
template <bool MESH>
__global__ void merge_rows(long long n_rows, long long n_vert, int n_mesh,
                           const long long *__restrict__ mesh_voff, const double *__restrict__ sigma,
                           const int *__restrict__ slot_ptr, long long *__restrict__ key,
                           double *__restrict__ val, int *__restrict__ row_len, int *__restrict__ err,
                           const int min_len, const int *__restrict__ row_list = nullptr, const int presorted_upto = 0) {
    // row_list: the kernel runs over these n_rows rows only (the assembly's slot rows), otherwise over rows 0..n_rows-1;
    // presorted_upto: rows of up to that many slots were sorted by sort_long_rows_wave (placeholder included)
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_rows) return;
    const long long r = row_list != nullptr ? row_list[idx] : idx;
    const int s0 = slot_ptr[r];
    const int n = slot_ptr[r + 1] - s0;
    if (n < min_len) return;                       // short rows were merged by merge_rows_lds
    long long *K = key + s0;
    double *V = val + s0;
    if (n > presorted_upto) {
        if (MESH) {                                // the diagonal placeholder of the assembled system (see above)
            K[0] = make_key((int)r, 2);
            V[0] = 0.0;
        }
        sort_slots(K, V, n);
    }
    double sig = 0.0;
    double dacc = 0.0;  // -(w_1 + w_2 + ...) in ascending column order   (diag[i] -= ratio, solver.py:203)
    if (MESH && r < n_vert) {
        sig = sigma[find_segment(mesh_voff, n_mesh, r)];
        int fwd_only = 0, bwd_only = 0;
        int i = 0;
        while (i < n) {
            const int col = (int)(K[i] >> 32);
            int terms = 0, seq_sum = 0;
            double wm = 0.0;
            while (i < n && (int)(K[i] >> 32) == col && (unsigned)(K[i] & 0xffffffffLL) < 2u) {
                wm = (terms == 0) ? V[i] : wm + V[i];          // 0. + c1 + c2   mesh.py:131-138
                seq_sum += (int)(K[i] & 0xffffffffLL);
                ++terms;
                ++i;
            }
            while (i < n && (int)(K[i] >> 32) == col) ++i;      // stamps: second walk
            // an interior edge has one forward and one backward term; anything else is the
            // "Non-manifold mesh" of mesh.py:342-343 (or a doubly used directed edge)
            if (terms > 2 || (terms == 2 && seq_sum != 1)) atomicExch(&err[ERR_NONMANIFOLD], 1);
            if (terms == 1) { if (seq_sum == 0) ++fwd_only; else ++bwd_only; }
            if (terms > 0 && wm != 0.0) dacc = dacc - wm;       // zero weights are skipped  solver.py:187-190
        }
        if (fwd_only > 1 || bwd_only > 1) atomicExch(&err[ERR_NONMANIFOLD], 1);
    }
    int o = 0;
    int i = 0;
    while (i < n) {
        const int col = (int)(K[i] >> 32);
        double v = 0.0;
        if (MESH) {
            if (col == (int)r) {
                v = sig * dacc;                                   // conductance * (diagonal entry)
            } else {
                int terms = 0;
                double wm = 0.0;
                while (i < n && (int)(K[i] >> 32) == col && (unsigned)(K[i] & 0xffffffffLL) < 2u) {
                    wm = (terms == 0) ? V[i] : wm + V[i];
                    ++terms;
                    ++i;
                }
                if (terms > 0) v = sig * wm;                      // conductance * laplace_operator(msh)
            }
        }
        while (i < n && (int)(K[i] >> 32) == col) {               // L[i,j] += stamp, in stamp order
            v = v + V[i];
            ++i;
        }
        if (v != 0.0) {                                           // exact zeros are not stored
            K[o] = (long long)col << 32;
            V[o] = v;
            ++o;
        }
    }
    row_len[r] = o;
}

// The listed rows of the assembled system, one WAVE per row (rows of up to kMergeWaveCap slots; the few beyond it are
// flagged and go through sort_long_rows_wave + merge_rows<true>).  The slots are rank-sorted by key in LDS, every lane
// that holds the first slot of a column walks that column's slots in key order -- the two mesh terms, then the stamps,
// exactly as merge_rows<true> adds them -- one lane adds up the diagonal over the columns in ascending order, the kept
// entries are written back compactly at the slot offset.  Same operations in the same order as the one-lane kernels,
// hence the same bits; it replaces three passes of one lane per row (33 + 64 + 91 us for the 18 k listed rows of config
// C4, each followed by a look at a flag on the host) by one of a few us per wave.
constexpr int kMergeWaveCap = 256;
__global__ __launch_bounds__(256) void merge_rows_mesh_wave(const int *__restrict__ n_list, const int *__restrict__ row_list,
                                                            long long n_vert, int n_mesh, const long long *__restrict__ mesh_voff,
                                                            const double *__restrict__ sigma, const int *__restrict__ slot_ptr,
                                                            long long *__restrict__ key, double *__restrict__ val,
                                                            int *__restrict__ row_len, int *__restrict__ err) {
    __shared__ long long Ku[4][kMergeWaveCap], Ks[4][kMergeWaveCap];      // unsorted (later: mesh sums per column) / sorted keys
    __shared__ double Vu[4][kMergeWaveCap], Vs[4][kMergeWaveCap];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    long long *ku = Ku[wv], *ks = Ks[wv];
    double *vu = Vu[wv], *vs = Vs[wv];
    double *gm = reinterpret_cast<double *>(ku);            // mesh sum of the column that starts at a slot
    const int n_rows = *n_list;
    for (int idx = blockIdx.x * 4 + wv; idx < n_rows; idx += gridDim.x * 4) {
        const int r = row_list[idx];
        const int s0 = slot_ptr[r], n = slot_ptr[r + 1] - s0;
        if (n > kMergeWaveCap) {
            if (lane == 0) *(volatile int *)&err[ERR_LONG_ROWS] = 1;
            continue;
        }
        for (int e = lane; e < n; e += 64) {
            // slot 0 is the row's diagonal placeholder (sequence 2, value 0): nobody wrote it
            ku[e] = e == 0 ? make_key(r, 2) : key[s0 + e];
            vu[e] = e == 0 ? 0.0 : val[s0 + e];
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        for (int e = lane; e < n; e += 64) {
            const long long k = ku[e];
            int rank = 0;
            for (int f = 0; f < n; ++f) {
                const long long kf = ku[f];
                rank += (kf < k || (kf == k && f < e)) ? 1 : 0;
            }
            ks[rank] = k;
            vs[rank] = vu[e];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const bool mesh_row = r < n_vert;
        const double sig = mesh_row ? sigma[find_segment(mesh_voff, n_mesh, r)] : 0.0;
        // pass 1: the mesh terms of every column (held by the lane of the column's first slot)
        int fwd_only = 0, bwd_only = 0;
        bool bad = false;
        for (int e = lane; e < n; e += 64) {
            const int c = (int)(ks[e] >> 32);
            const bool head = e == 0 || (int)(ks[e - 1] >> 32) != c;
            double wm = 0.0;
            if (head && mesh_row) {
                int terms = 0, seq_sum = 0, i = e;
                while (i < n && (int)(ks[i] >> 32) == c && (unsigned)(ks[i] & 0xffffffffLL) < 2u) {
                    wm = (terms == 0) ? vs[i] : wm + vs[i];              // 0. + c1 + c2   mesh.py:131-138
                    seq_sum += (int)(ks[i] & 0xffffffffLL);
                    ++terms;
                    ++i;
                }
                // an interior edge has one forward and one backward term; anything else is the "Non-manifold mesh" of
                // mesh.py:342-343 (or a doubly used directed edge)
                if (terms > 2 || (terms == 2 && seq_sum != 1)) bad = true;
                if (terms == 1) { if (seq_sum == 0) ++fwd_only; else ++bwd_only; }
                if (terms == 0) wm = 0.0;
            }
            gm[e] = wm;                                                 // (0 where no column starts: adds nothing below)
        }
        for (int off = 32; off > 0; off >>= 1) {
            fwd_only += __shfl_xor(fwd_only, off, 64);
            bwd_only += __shfl_xor(bwd_only, off, 64);
        }
        if (__ballot(bad) != 0ull || fwd_only > 1 || bwd_only > 1) {
            if (lane == 0) atomicExch(&err[ERR_NONMANIFOLD], 1);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // the diagonal: -(w_1 + w_2 + ...) in ascending column order, zero weights skipped   (solver.py:187-190, 203)
        double dacc = 0.0;
        if (mesh_row) {
            if (lane == 0)
                for (int e = 0; e < n; ++e) {
                    const double wm = gm[e];
                    if (wm != 0.0) dacc = dacc - wm;
                }
            dacc = __shfl(dacc, 0, 64);
        }
        // pass 2: value of every column, stamps added in stamp order; exact zeros are not stored
        int base = 0;
        for (int e0 = 0; e0 < n; e0 += 64) {
            const int e = e0 + lane;
            bool keep = false;
            int c = 0;
            double v = 0.0;
            if (e < n) {
                c = (int)(ks[e] >> 32);
                const bool head = e == 0 || (int)(ks[e - 1] >> 32) != c;
                if (head) {
                    int i = e;
                    while (i < n && (int)(ks[i] >> 32) == c && (unsigned)(ks[i] & 0xffffffffLL) < 2u) ++i;
                    if (mesh_row) {
                        if (c == r) v = sig * dacc;                     // conductance * (diagonal entry)
                        else if (i > e) v = sig * gm[e];                // conductance * laplace_operator(msh)
                    }
                    while (i < n && (int)(ks[i] >> 32) == c) {          // L[i,j] += stamp, in stamp order
                        v = v + vs[i];
                        ++i;
                    }
                    keep = v != 0.0;
                }
            }
            const unsigned long long mk = __ballot(keep);
            if (keep) {
                const int o = base + __popcll(mk & ((1ull << lane) - 1ull));
                key[s0 + o] = (long long)c << 32;
                val[s0 + o] = v;
            }
            base += __popcll(mk);
        }
        if (lane == 0) row_len[r] = base;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // LDS is reused by the next row
        __builtin_amdgcn_wave_barrier();
    }
}

// Short rows of the generic merge: sort and add duplicates inside LDS ([slot][thread] layout), one lane per row.
// Same order of additions as merge_rows<false> (key = column, then sequence), hence the same bits.
template <int CAP>
__global__ __launch_bounds__(128) void merge_rows_lds(long long n_rows, const int *__restrict__ slot_ptr,
                                                      long long *__restrict__ key, double *__restrict__ val,
                                                      int *__restrict__ row_len) {
    __shared__ long long Kc[CAP][128];
    __shared__ double Vc[CAP][128];
    const int t = threadIdx.x;
    const long long r = (long long)blockIdx.x * 128 + t;
    if (r >= n_rows) return;
    const int s0 = slot_ptr[r];
    const int n = slot_ptr[r + 1] - s0;
    if (n > CAP) return;                           // left to merge_rows<false>
    for (int i = 0; i < n; ++i) {                  // insertion sort while loading
        const long long k = key[s0 + i];
        const double v = val[s0 + i];
        int j = i - 1;
        while (j >= 0 && Kc[j][t] > k) {
            Kc[j + 1][t] = Kc[j][t];
            Vc[j + 1][t] = Vc[j][t];
            --j;
        }
        Kc[j + 1][t] = k;
        Vc[j + 1][t] = v;
    }
    int o = 0, i = 0;
    while (i < n) {
        const int col = (int)(Kc[i][t] >> 32);
        double v = 0.0;
        while (i < n && (int)(Kc[i][t] >> 32) == col) {
            v = v + Vc[i][t];
            ++i;
        }
        if (v != 0.0) {
            key[s0 + o] = (long long)col << 32;
            val[s0 + o] = v;
            ++o;
        }
    }
    row_len[r] = o;
}

// Rows already compacted at their slot offsets -> final CSR arrays.  A wave moves 64 consecutive rows: both the
// source span [slot_ptr[r0], slot_ptr[r0+64]) and the destination span [rowptr[r0], rowptr[r0+64]) are
// contiguous, lanes walk the destination (coalesced stores) and find their row by bisection in LDS.
static int compact_rows_per_wave(long long n_rows, long long nnz) {
    if (n_rows <= 0 || n_rows > 100000 || nnz <= 16 * n_rows) return 64;     // big levels: bandwidth-bound, 64 rows per wave
    return nnz <= 128 * n_rows ? 8 : 1;
}

__global__ __launch_bounds__(256) void compact_rows(long long n_rows, const int *__restrict__ slot_ptr,
                                                    const int *__restrict__ rowptr, const long long *__restrict__ key,
                                                    const double *__restrict__ val, int *__restrict__ cols,
                                                    double *__restrict__ vals, const int rpw) {
    // rpw: rows per wave and turn (64 for mesh-like rows; 8 or 1 for the long rows of the small coarse operators, where 64
    // rows of hundreds of entries kept a wave busy for 80 us while most of the chip had nothing to do)
    __shared__ int rp_all[4][65];
    __shared__ int sp_all[4][65];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int *rp = rp_all[w], *sp = sp_all[w];
    const long long n_wt = (n_rows + rpw - 1) / rpw;
    for (long long wt = (long long)blockIdx.x * 4 + w; wt < n_wt; wt += (long long)gridDim.x * 4) {
        const long long r0 = wt * rpw;
        const int nr = (int)((n_rows - r0) < rpw ? (n_rows - r0) : rpw);
        if (lane <= nr) rp[lane] = rowptr[r0 + lane];
        if (lane < nr) sp[lane] = slot_ptr[r0 + lane];
        if (lane == 0 && nr == 64) rp[64] = rowptr[r0 + 64];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const int d0 = rp[0], d1 = rp[nr];
        for (int k = d0 + lane; k < d1; k += 64) {
            int lo = 0, hi = nr;                     // largest row with rp[row] <= k
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (rp[mid] <= k) lo = mid; else hi = mid;
            }
            const int src = sp[lo] + (k - rp[lo]);
            cols[k] = (int)(key[src] >> 32);
            vals[k] = val[src];
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ void fill_value_i32(int *p, long long n, int v) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ---- reduce:  out = scale * P^T M P ------------------------------------------------------------
__global__ void reduce_count(long long n_rows, const int *__restrict__ rowptr, const int *__restrict__ cols,
                             const int *__restrict__ map, const int *__restrict__ cmap, int *__restrict__ cnt) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const int t = map[r];
    if (t < 0) return;
    int c = 0;
    for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) c += (cmap[cols[k]] >= 0) ? 1 : 0;
    if (c) atomicAdd(&cnt[t], c);
}

__global__ void reduce_fill(long long n_rows, const int *__restrict__ rowptr, const int *__restrict__ cols,
                            const double *__restrict__ vals, const int *__restrict__ map,
                            const int *__restrict__ cmap, double scale,
                            const int *__restrict__ slot_ptr, int *__restrict__ cursor,
                            long long *__restrict__ key, double *__restrict__ val) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const int t = map[r];
    if (t < 0) return;
    int c = 0;
    for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) c += (cmap[cols[k]] >= 0) ? 1 : 0;
    if (!c) return;
    int s = slot_ptr[t] + atomicAdd(&cursor[t], c);
    for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) {
        const int tc = cmap[cols[k]];
        if (tc < 0) continue;
        key[s] = make_key(tc, k);       // ties are broken by the position in the source matrix
        val[s] = scale * vals[k];
        ++s;
    }
}


// ---- relabel with injective maps: every output row is ONE relabelled source row -------------------------------------
// (ground elimination, locality permutations, the row / column split of the row-partitioned solver; only tied groups
// of unknowns -- voltage sources -- merge rows.)  No slots, no merge, no compaction: count, scan, write.
// flag[0] = 1: two indices share a target (not injective); flag[1] = 1: an entry outside [-1, n_out) (checked here, on
// the device copy of the map, instead of in a host loop over 10 M entries)
__global__ void map_is_injective(long long n, const int *__restrict__ map, int n_out, int *__restrict__ hist,
                                 int *__restrict__ flag) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int t = map[i];
    if (t < -1 || t >= n_out) {
        *(volatile int *)(flag + 1) = 1;
        return;
    }
    if (t >= 0 && atomicAdd(&hist[t], 1) > 0) *(volatile int *)flag = 1;
}

// (the count does not look at the values: a kept entry whose scaled value is exactly zero -- an explicit zero in a
// matrix handed in by the caller; the assembly stores none -- is noticed by the fill kernel, and the whole relabel is
// then redone through the slots, which drop such entries)
__global__ void relabel_count_direct(long long n_rows, const int *__restrict__ rowptr, const int *__restrict__ cols,
                                     const int *__restrict__ map, const int *__restrict__ cmap, int *__restrict__ cnt) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const int t = map[r];
    if (t < 0) return;
    int c = 0;
    for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) c += (cmap[cols[k]] >= 0) ? 1 : 0;
    cnt[t] = c;
}

template <int CAP>
__global__ __launch_bounds__(128) void relabel_fill_direct(long long n_rows, const int *__restrict__ rowptr,
                                                           const int *__restrict__ cols, const double *__restrict__ vals,
                                                           const int *__restrict__ map, const int *__restrict__ cmap,
                                                           double scale, const int *__restrict__ out_rowptr,
                                                           int *__restrict__ out_cols, double *__restrict__ out_vals,
                                                           int *__restrict__ zero_seen) {
    __shared__ int Cc[CAP][128];
    __shared__ double Vc[CAP][128];
    const int th = threadIdx.x;
    const long long r = (long long)blockIdx.x * 128 + th;
    if (r >= n_rows) return;
    const int t = map[r];
    if (t < 0) return;
    const int o0 = out_rowptr[t], n_out = out_rowptr[t + 1] - o0;
    if (n_out <= CAP) {
        // the usual case: the row's entries sorted by their new column while being collected in LDS
        int m = 0;
        for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) {
            const int tc = cmap[cols[k]];
            const double v = scale * vals[k];
            if (tc < 0) continue;
            if (v == 0.0) *(volatile int *)zero_seen = 1;
            int u = m - 1;
            while (u >= 0 && Cc[u][th] > tc) {
                Cc[u + 1][th] = Cc[u][th];
                Vc[u + 1][th] = Vc[u][th];
                --u;
            }
            Cc[u + 1][th] = tc;
            Vc[u + 1][th] = v;
            ++m;
        }
        for (int u = 0; u < m; ++u) {
            out_cols[o0 + u] = Cc[u][th];
            out_vals[o0 + u] = Vc[u][th];
        }
    } else {
        // long rows (hubs): written in source order, then sorted in place
        int m = 0;
        for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) {
            const int tc = cmap[cols[k]];
            const double v = scale * vals[k];
            if (tc < 0) continue;
            if (v == 0.0) *(volatile int *)zero_seen = 1;
            out_cols[o0 + m] = tc;
            out_vals[o0 + m] = v;
            ++m;
        }
        for (int i = 1; i < m; ++i) {
            const int c = out_cols[o0 + i];
            const double v = out_vals[o0 + i];
            int u = i - 1;
            while (u >= 0 && out_cols[o0 + u] > c) {
                out_cols[o0 + u + 1] = out_cols[o0 + u];
                out_vals[o0 + u + 1] = out_vals[o0 + u];
                --u;
            }
            out_cols[o0 + u + 1] = c;
            out_vals[o0 + u + 1] = v;
        }
    }
}

// ---- order-preserving maps: eliminations without a permutation -------------------------------------------------
// The reduction of the reference's system to its potential block (solver.py:544-560: drop the ground vertex and the
// multiplier row, flip the sign) is a map that only DROPS indices: the kept ones keep their order.  Then a row's kept entries
// are in column order as they stand -- nothing to sort, nothing to check for duplicates -- and the relabel is a copy with
// holes: count, scan, copy.  (Through the general direct path the same call cost 1.8 ms at 10 M rows: a histogram for the
// injectivity test, a thread walking every row entry by entry, an insertion sort in LDS per row.)
// flag[0] = 1: not such a map; flag[1] = 1: an entry outside [-1, n_out); flag[2] = 1: some index maps to 0
__global__ void map_is_compaction(long long n, const int *__restrict__ map, int n_out, int *__restrict__ flag) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int t = map[i];
    if (t < -1 || t >= n_out) {
        *(volatile int *)(flag + 1) = 1;
        return;
    }
    if (t < 0) return;
    if (t == 0) *(volatile int *)(flag + 2) = 1;
    // the next kept index carries t + 1 (the last one n_out - 1): with an index that maps to 0 this makes the kept values
    // 0, 1, ..., n_out - 1 in order.  A run of more than 64 dropped indices is not followed: the general path takes such maps
    long long j = i + 1;
    int tn = -1;
    for (int step = 0; step < 64 && j < n; ++step, ++j) {
        tn = map[j];
        if (tn >= 0) break;
    }
    if (j >= n) {
        if (t != n_out - 1) *(volatile int *)flag = 1;
    } else if (tn < 0 || tn != t + 1) {
        *(volatile int *)flag = 1;
    }
}

struct __attribute__((packed, aligned(4))) RlI4 { int x, y, z, w; };
struct __attribute__((packed, aligned(4))) RlD2 { double x, y; };

// kept entries per row, one lane per row, eight entries per step (two unaligned 16-byte loads)
__global__ __launch_bounds__(256) void relabel_count_ordered(long long n_rows, const int *__restrict__ rowptr,
                                                             const int *__restrict__ cols, const int *__restrict__ map,
                                                             const int *__restrict__ cmap, const int n_out, int *__restrict__ cnt) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const int t = map[r];
    // (launched before the host has read map_is_compaction's verdict: an entry outside the result is that kernel's to report,
    // not this one's to write through)
    if (t < 0 || t >= n_out) return;
    const int k0 = rowptr[r], k1 = rowptr[r + 1];
    int c = 0;
    for (int k = k0; k < k1; k += 8) {
        const RlI4 a = *reinterpret_cast<const RlI4 *>(cols + k), b = *reinterpret_cast<const RlI4 *>(cols + k + 4);   // (the arrays are padded)
        const int cc[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        int tc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) tc[u] = k + u < k1 ? cmap[cc[u]] : -1;
#pragma unroll
        for (int u = 0; u < 8; ++u) c += tc[u] >= 0 ? 1 : 0;
    }
    cnt[t] = c;
}

// the copy: a lane takes its row eight entries at a time, maps the columns, closes the holes in registers and writes the
// kept entries at the row's place in the result
__global__ __launch_bounds__(256) void relabel_fill_ordered(long long n_rows, const int *__restrict__ rowptr,
                                                            const int *__restrict__ cols, const double *__restrict__ vals,
                                                            const int *__restrict__ map, const int *__restrict__ cmap,
                                                            double scale, const int *__restrict__ out_rowptr,
                                                            int *__restrict__ out_cols, double *__restrict__ out_vals,
                                                            int *__restrict__ zero_seen) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const int t = map[r];
    if (t < 0) return;
    const int k0 = rowptr[r], k1 = rowptr[r + 1];
    int o = out_rowptr[t];
    bool zero = false;
    for (int k = k0; k < k1; k += 8) {
        const RlI4 a = *reinterpret_cast<const RlI4 *>(cols + k), b = *reinterpret_cast<const RlI4 *>(cols + k + 4);
        const RlD2 v0 = *reinterpret_cast<const RlD2 *>(vals + k), v1 = *reinterpret_cast<const RlD2 *>(vals + k + 2),
                   v2 = *reinterpret_cast<const RlD2 *>(vals + k + 4), v3 = *reinterpret_cast<const RlD2 *>(vals + k + 6);
        const int cc[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        const double vv[8] = {v0.x, v0.y, v1.x, v1.y, v2.x, v2.y, v3.x, v3.y};
        int tc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) tc[u] = k + u < k1 ? cmap[cc[u]] : -1;
#pragma unroll
        for (int h = 0; h < 8; h += 4) {
            if (tc[h] >= 0 && tc[h + 1] >= 0 && tc[h + 2] >= 0 && tc[h + 3] >= 0) {
                // no hole in these four (all rows but those next to a dropped unknown): wide stores
                double w[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    w[u] = scale * vv[h + u];
                    zero = zero || w[u] == 0.0;
                }
                *reinterpret_cast<RlI4 *>(out_cols + o) = RlI4{tc[h], tc[h + 1], tc[h + 2], tc[h + 3]};
                *reinterpret_cast<RlD2 *>(out_vals + o) = RlD2{w[0], w[1]};
                *reinterpret_cast<RlD2 *>(out_vals + o + 2) = RlD2{w[2], w[3]};
                o += 4;
            } else {
#pragma unroll
                for (int u = h; u < h + 4; ++u)
                    if (tc[u] >= 0) {
                        const double w = scale * vv[u];
                        zero = zero || w == 0.0;
                        out_cols[o] = tc[u];
                        out_vals[o] = w;
                        ++o;
                    }
            }
        }
    }
    if (zero) *(volatile int *)zero_seen = 1;
}

// The same two passes with a WAVE per 64 rows, in the shape of the SpMV's matrix stream: the rows of a tile are one
// contiguous range of the source and -- the map keeps the order -- their kept entries one contiguous range of the result.
// The wave streams the range lane-consecutively (lane l takes entries l, l + 64, ...: 256 / 512 contiguous bytes per load)
// and a kept entry's place is the running count of kept entries in front of it (a ballot and a population count), so the
// stores of an instruction are contiguous too, holes closed: full lines, where a lane per row wrote 16-byte pieces at a
// stride of 56 / 112 bytes (1.64 x the bytes of the result by the write counter, 593 us at 10 M rows).
// Dropped rows and columns are rare (the ground vertex, the multiplier row): the count pass only looks for DROPPED entries and
// finds their rows by bisection of the tile's row starts in LDS; the copy masks the entries of a dropped row by its range.
constexpr int kRlEpl = 8;      // entries per lane and pass

__global__ __launch_bounds__(256) void relabel_count_wave(const int n_rows, const int *__restrict__ rowptr,
                                                          const int *__restrict__ cols, const int *__restrict__ map,
                                                          const int *__restrict__ cmap, const int n_out, int *__restrict__ cnt) {
    __shared__ int rs_s[4][65];
    __shared__ int drop_s[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n_tiles = (n_rows + 63) / 64;
    for (int tile = blockIdx.x * 4 + w; tile < n_tiles; tile += gridDim.x * 4) {
        const int r = tile * 64 + lane;
        const int last = min(63, n_rows - 1 - tile * 64);
        int rs = 0, re = 0, t = -1;
        if (r < n_rows) {
            rs = rowptr[r];
            re = rowptr[r + 1];
            t = map[r];
            if (t >= n_out) t = -1;      // (map_is_compaction reports it)
        }
        const int k0 = __shfl(rs, 0, 64), k1 = __shfl(re, last, 64);
        bool any = false;
        bool staged = false;
        for (int base = k0; base < k1; base += 64 * kRlEpl) {
            int c[kRlEpl], tc[kRlEpl];
#pragma unroll
            for (int j = 0; j < kRlEpl; ++j) {
                const int e = base + lane + 64 * j;
                c[j] = e < k1 ? cols[e] : -1;
            }
#pragma unroll
            for (int j = 0; j < kRlEpl; ++j) tc[j] = c[j] >= 0 ? cmap[c[j]] : 0;
            bool dropped = false;
#pragma unroll
            for (int j = 0; j < kRlEpl; ++j) dropped = dropped || tc[j] < 0;
            if (__ballot(dropped) != 0ull) {          // wave-uniform, rare
                if (!staged) {
                    rs_s[w][lane] = rs;
                    if (lane == last) rs_s[w][last + 1] = re;
                    drop_s[w][lane] = 0;
                    staged = true;
                    any = true;
                    __builtin_amdgcn_wave_barrier();
                }
#pragma unroll
                for (int j = 0; j < kRlEpl; ++j)
                    if (tc[j] < 0) {
                        const int e = base + lane + 64 * j;
                        int lo = 0, hi = last + 1;             // the row with rs_s[row] <= e < rs_s[row + 1]
                        while (hi - lo > 1) {
                            const int mid = (lo + hi) >> 1;
                            if (rs_s[w][mid] <= e) lo = mid;
                            else hi = mid;
                        }
                        atomicAdd(&drop_s[w][lo], 1);
                    }
            }
        }
        int kept = re - rs;
        if (any) {
            __builtin_amdgcn_wave_barrier();
            kept -= drop_s[w][lane];
            __builtin_amdgcn_wave_barrier();
        }
        if (t >= 0) cnt[t] = kept;
    }
}

__global__ __launch_bounds__(256) void relabel_fill_wave(const int n_rows, const int *__restrict__ rowptr,
                                                         const int *__restrict__ cols, const double *__restrict__ vals,
                                                         const int *__restrict__ map, const int *__restrict__ cmap,
                                                         const double scale, const int *__restrict__ out_rowptr,
                                                         int *__restrict__ out_cols, double *__restrict__ out_vals,
                                                         int *__restrict__ zero_seen) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n_tiles = (n_rows + 63) / 64;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));      // the lanes below this one
    bool zero = false;
    for (int tile = blockIdx.x * 4 + w; tile < n_tiles; tile += gridDim.x * 4) {
        const int r = tile * 64 + lane;
        const int last = min(63, n_rows - 1 - tile * 64);
        int rs = 0, re = 0, t = -1;
        if (r < n_rows) {
            rs = rowptr[r];
            re = rowptr[r + 1];
            t = map[r];
        }
        const unsigned long long kept_rows = __ballot(t >= 0);
        if (kept_rows == 0ull) continue;
        unsigned long long dropped_rows = __ballot(r < n_rows && t < 0 && re > rs);
        const int first = __ffsll((long long)kept_rows) - 1;
        int o = out_rowptr[__shfl(t, first, 64)];              // where the tile's kept entries start in the result
        const int k0 = __shfl(rs, 0, 64), k1 = __shfl(re, last, 64);
        for (int base = k0; base < k1; base += 64 * kRlEpl) {
            int c[kRlEpl], tc[kRlEpl];
            double v[kRlEpl];
#pragma unroll
            for (int j = 0; j < kRlEpl; ++j) {
                const int e = base + lane + 64 * j;
                c[j] = e < k1 ? cols[e] : -1;
            }
#pragma unroll
            for (int j = 0; j < kRlEpl; ++j) {
                const int e = base + lane + 64 * j;
                v[j] = e < k1 ? vals[e] : 0.0;
            }
#pragma unroll
            for (int j = 0; j < kRlEpl; ++j) tc[j] = c[j] >= 0 ? cmap[c[j]] : -1;
            if (dropped_rows != 0ull) {                        // wave-uniform, rare: the entries of a dropped row go
                for (unsigned long long m = dropped_rows; m != 0ull; m &= m - 1) {
                    const int d = __ffsll((long long)m) - 1;
                    const int ds = __shfl(rs, d, 64), de = __shfl(re, d, 64);
#pragma unroll
                    for (int j = 0; j < kRlEpl; ++j) {
                        const int e = base + lane + 64 * j;
                        if (e >= ds && e < de) tc[j] = -1;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < kRlEpl; ++j) {
                const bool keep = tc[j] >= 0;
                const unsigned long long b = __ballot(keep);
                if (keep) {
                    const int pos = o + __popcll(b & lt);
                    const double wv = scale * v[j];
                    zero = zero || wv == 0.0;
                    out_cols[pos] = tc[j];
                    out_vals[pos] = wv;
                }
                o += __popcll(b);
            }
        }
    }
    if (zero) *(volatile int *)zero_seen = 1;
}

// ---- power density ---------------------------------------------------------------------------
// compute_triangle_gradient (solver.py:689-725) with the face vertex order of the reference:
// Face.edge is the last interior half-edge created (v3->v1, mesh.py:320-325) so face.vertices
// yields (v3, v1, v2).
__device__ __forceinline__ double interp(double x1, double y1, double x2, double y2, double x3, double y3,
                                         double f1, double f2, double f3, double x, double y) {
    const double D = (y2 - y3) * (x1 - x3) + (x3 - x2) * (y1 - y3);
    const double l1 = ((y2 - y3) * (x - x3) + (x3 - x2) * (y - y3)) / D;
    const double l2 = ((y3 - y1) * (x - x3) + (x1 - x3) * (y - y3)) / D;
    const double l3 = 1 - l1 - l2;
    return l1 * f1 + l2 * f2 + l3 * f3;
}

__global__ void power_density_kernel(long long n_tri, const int *__restrict__ tri, const double *__restrict__ xy,
                                     int n_mesh, const long long *__restrict__ mesh_voff,
                                     const long long *__restrict__ mesh_toff, const double *__restrict__ sigma,
                                     const double *__restrict__ pot, double *__restrict__ out,
                                     double *__restrict__ gx_out, double *__restrict__ gy_out, int *__restrict__ err) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tri) return;
    const int m = find_segment(mesh_toff, n_mesh, t);
    const long long v0 = mesh_voff[m];
    const long long nv = mesh_voff[m + 1] - v0;
    const int l1 = tri[3 * t + 2], l2 = tri[3 * t], l3 = tri[3 * t + 1];
    if (l1 < 0 || l2 < 0 || l3 < 0 || l1 >= nv || l2 >= nv || l3 >= nv) {      // checked here instead of in a host loop over all triangles
        *(volatile int *)err = 1;
        return;
    }
    const long long g1 = v0 + l1, g2 = v0 + l2, g3 = v0 + l3;
    const double x1 = xy[2 * g1], y1 = xy[2 * g1 + 1];
    const double x2 = xy[2 * g2], y2 = xy[2 * g2 + 1];
    const double x3 = xy[2 * g3], y3 = xy[2 * g3 + 1];
    const double f1 = pot[g1], f2 = pot[g2], f3 = pot[g3];
    const double gx = interp(x1, y1, x2, y2, x3, y3, f1, f2, f3, x1 + 1, y1) - f1;
    const double gy = interp(x1, y1, x2, y2, x3, y3, f1, f2, f3, x1, y1 + 1) - f1;
    if (gx_out) {
        gx_out[t] = gx;
        gy_out[t] = gy;
    }
    if (out) {
        const double s = sigma[m];
        const double jx = gx * s, jy = gy * s;      // J = E * conductivity
        out[t] = jx * gx + jy * gy;                 // J.dot(E)
    }
}

// ---- host orchestration ----------------------------------------------------------------------
// shared tail: slots (key,val,slot_ptr) already filled -> merged CSR
// padne_assemble_system_ex(flags & 1): the triangles are a rank's piece of a larger mesh (owned vertices + the ring of
// vertices around them); the fans of the ring vertices are incomplete by construction, so the manifold test is off
static thread_local bool t_partial_mesh = false;
// assemblies of this process whose rows were built by the two-pass second path (forced, or after the single pass gave up)
static std::atomic<long long> g_two_pass_fallbacks{0};

// the listed rows of the assembled system, merged in place at their slot offsets (row_len = what each keeps)
static int merge_listed_mesh_rows(padne_ctx *ctx, long long n_vert, int n_mesh, const long long *d_voff, const double *d_sigma,
                                  const int *slot_ptr, long long *key, double *val, int *row_len, int *d_err,
                                  const int *row_list, const int *n_list_dev, long long n_merge) {
    hipStream_t s = ctx->stream;
    if (n_merge <= 0) return PADNE_OK;
    // one wave per row; a row beyond its capacity (a hub of hundreds of lumped elements) is flagged and takes the wave
    // sort + the one-lane global-memory merge
    hipLaunchKernelGGL(merge_rows_mesh_wave, dim3(std::min(nblk(n_merge, 4), 8192u)), dim3(256), 0, s, n_list_dev, row_list,
                       n_vert, n_mesh, d_voff, d_sigma, slot_ptr, key, val, row_len, d_err);
    PADNE_HIP_CHECK(hipGetLastError());
    int h_long[ERR_WORDS];
    PADNE_TRY(read_back(ctx, d_err, sizeof(h_long), h_long));
    if (h_long[ERR_LONG_ROWS]) {
        hipLaunchKernelGGL((sort_long_rows_wave<kWaveSortCap, 4>), dim3(std::min(nblk(n_merge, 16), 8192u)), dim3(256), 0, s,
                           n_merge, row_list, slot_ptr, key, val, kMergeWaveCap + 1, 1);
        hipLaunchKernelGGL(merge_rows<true>, dim3(nblk(n_merge, 128)), dim3(128), 0, s, n_merge, n_vert, n_mesh, d_voff,
                           d_sigma, slot_ptr, key, val, row_len, d_err, kMergeWaveCap + 1, row_list, kWaveSortCap);
        PADNE_HIP_CHECK(hipGetLastError());
    }
    return PADNE_OK;
}

// shared tail of the generic merge (padne_csr_reduce): slots already filled -> merged CSR
static int finish_rows(padne_ctx *ctx, Scratch &sc, long long n_rows, long long n_cols, const int *slot_ptr, long long *key,
                       double *val, padne_csr **out) {
    hipStream_t s = ctx->stream;
    int *row_len = nullptr;
    PADNE_TRY(sc.alloc(&row_len, (size_t)n_rows + 1));
    PADNE_TRY(merge_slots_generic(ctx, n_rows, slot_ptr, key, val, row_len));
    int *rowptr_tmp = nullptr;
    PADNE_TRY(sc.alloc(&rowptr_tmp, (size_t)n_rows + 1));
    int64_t nnz = 0;
    PADNE_TRY(exclusive_scan_i32(ctx, row_len, rowptr_tmp, n_rows, &nnz));
    padne_csr *m = nullptr;
    PADNE_TRY(csr_alloc(ctx, n_rows, n_cols, nnz, &m));
    hipError_t e = hipMemcpyAsync(m->rowptr, rowptr_tmp, sizeof(int32_t) * (size_t)(n_rows + 1),
                                  hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) {
        const int rpw = compact_rows_per_wave(n_rows, nnz);
        hipLaunchKernelGGL(compact_rows, dim3(std::min(nblk((n_rows + rpw - 1) / rpw, 4), 65536u)), dim3(256), 0, s, n_rows, slot_ptr,
                           m->rowptr, key, val, m->cols, m->vals, rpw);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
        set_error("row compaction failed: %s", hipGetErrorString(e));
        padne_csr_destroy(m);
        return PADNE_E_HIP;
    }
    *out = m;
    return PADNE_OK;
}


// exported to amg.hip --------------------------------------------------------------------------
int merge_slots_generic(padne_ctx *ctx, long long n_rows, const int *slot_ptr, long long *key, double *val,
                        int *row_len) {
    constexpr int kLdsCap = 32;
    hipLaunchKernelGGL(merge_rows_lds<kLdsCap>, dim3(nblk(n_rows, 128)), dim3(128), 0, ctx->stream, n_rows, slot_ptr, key,
                       val, row_len);
    // rows beyond the LDS pass (transposed prolongators and products of the coarse levels: hundreds of entries): one
    // wave sorts each, the one-lane merge then walks a sorted row
    hipLaunchKernelGGL(sort_long_rows_wave<kWaveSortCap>, dim3(std::min(nblk(n_rows, 256), 2048u)), dim3(256), 0, ctx->stream,
                       n_rows, (const int *)nullptr, slot_ptr, key, val, kLdsCap + 1, 0);
    hipLaunchKernelGGL(merge_rows<false>, dim3(nblk(n_rows, 128)), dim3(128), 0, ctx->stream, n_rows, 0LL, 0,
                       (const long long *)nullptr, (const double *)nullptr, slot_ptr, key, val, row_len,
                       (int *)nullptr, kLdsCap + 1, (const int *)nullptr, kWaveSortCap);
    PADNE_HIP_CHECK(hipGetLastError());
    return PADNE_OK;
}

// while_scanning: called between the launch of the scan of the row lengths and the look at its total (what the caller
// launches there, for another stream, is launched while this stream works)
int csr_from_slots(padne_ctx *ctx, long long n_rows, long long n_cols, const int *slot_ptr, const long long *key,
                   const double *val, const int *row_len, padne_csr **out, int (*while_scanning)(void *), void *while_scanning_arg) {
    hipStream_t s = ctx->stream;
    Scratch sc(ctx);
    int *rowptr_tmp = nullptr;
    PADNE_TRY(sc.alloc(&rowptr_tmp, (size_t)n_rows + 1));
    int64_t nnz = 0;
    {
        ScanTicket ticket;
        long long h[2] = {0, 0};
        PADNE_TRY(scan_i32_begin(ctx, row_len, rowptr_tmp, n_rows, &ticket, true));
        const int rc_cb = while_scanning != nullptr ? while_scanning(while_scanning_arg) : PADNE_OK;
        const int rc_scan = scan_i32_end(ctx, &ticket, h);
        PADNE_TRY(rc_cb);
        PADNE_TRY(rc_scan);
        if (h[0] < 0 || h[1] != 0) {
            set_error("scan of negative counts");
            return PADNE_E_INVALID;
        }
        if (h[0] >= 2147483647LL) {
            set_error("%lld entries exceed the 32-bit index space", h[0]);
            return PADNE_E_TOOLARGE;
        }
        nnz = h[0];
    }
    padne_csr *m = nullptr;
    PADNE_TRY(csr_alloc(ctx, n_rows, n_cols, nnz, &m));
    hipError_t e = hipMemcpyAsync(m->rowptr, rowptr_tmp, sizeof(int32_t) * (size_t)(n_rows + 1),
                                  hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess && n_rows > 0) {
        const int rpw = compact_rows_per_wave(n_rows, nnz);
        hipLaunchKernelGGL(compact_rows, dim3(std::min(nblk((n_rows + rpw - 1) / rpw, 4), 65536u)), dim3(256), 0, s, n_rows, slot_ptr,
                           m->rowptr, key, val, m->cols, m->vals, rpw);
        e = hipGetLastError();
    }
    // no synchronisation: the caller's scratch goes back to the pool, whose reuse is ordered on the same stream
    if (e != hipSuccess) {
        set_error("row compaction failed: %s", hipGetErrorString(e));
        padne_csr_destroy(m);
        return PADNE_E_HIP;
    }
    *out = m;
    return PADNE_OK;
}

// slots that are already exact (row i occupies [slot_ptr[i], slot_ptr[i+1]) completely, e.g. a transpose whose
// counts are exact and whose rows have been sorted): no scan, no per-row compaction, one streaming pass
__global__ void unpack_slots_kernel(long long nnz, const long long *__restrict__ key, const double *__restrict__ val,
                                    int *__restrict__ cols, double *__restrict__ vals) {
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (long long)gridDim.x * blockDim.x) {
        cols[k] = (int)(key[k] >> 32);
        vals[k] = val[k];
    }
}

// Slots that only need sorting (a transpose: no duplicates, nothing to add or drop): every row by one wave -- rank by
// counting smaller keys in LDS, rows of up to 64 slots at full occupancy, longer ones with the 1024-slot variant, the
// rare rest by the one-lane insertion sort.
int sort_slots_exact(padne_ctx *ctx, long long n_rows, const int *slot_ptr, long long *key, double *val, int *row_len_scratch) {
    // rows of up to 64 slots: eight rows per wave and turn, sorted in 2 KiB of LDS per wave; the rows beyond that are
    // listed and sorted by a workgroup each (sort_listed_rows_block); what exceeds even its 1024 slots is left to the
    // one-lane merge.  Nothing here needs the host.
    Scratch sc(ctx);
    int *long_list = nullptr;
    PADNE_TRY(sc.alloc(&long_list, (size_t)n_rows + 1));
    int *n_long = row_len_scratch + n_rows;          // the scratch has n_rows + 1 entries
    PADNE_HIP_CHECK(hipMemsetAsync(n_long, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL((sort_long_rows_wave<64, 8>), dim3(std::min(nblk(n_rows, 32), 8192u)), dim3(256), 0, ctx->stream, n_rows,
                       (const int *)nullptr, slot_ptr, key, val, 2, 0, n_long, (const int *)nullptr, long_list);
    hipLaunchKernelGGL(sort_listed_rows_block, dim3(1024), dim3(256), 0, ctx->stream, (const int *)n_long, (const int *)long_list,
                       slot_ptr, key, val);
    hipLaunchKernelGGL(merge_rows<false>, dim3(nblk(n_rows, 128)), dim3(128), 0, ctx->stream, n_rows, 0LL, 0,
                       (const long long *)nullptr, (const double *)nullptr, slot_ptr, key, val, row_len_scratch,
                       (int *)nullptr, kWaveSortCap + 1);
    PADNE_HIP_CHECK(hipGetLastError());
    return PADNE_OK;
}

int csr_from_exact_slots(padne_ctx *ctx, long long n_rows, long long n_cols, long long nnz, const int *slot_ptr,
                         const long long *key, const double *val, padne_csr **out) {
    hipStream_t s = ctx->stream;
    padne_csr *m = nullptr;
    PADNE_TRY(csr_alloc(ctx, n_rows, n_cols, nnz, &m));
    hipError_t e = hipMemcpyAsync(m->rowptr, slot_ptr, sizeof(int32_t) * (size_t)(n_rows + 1), hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess && nnz > 0) {
        hipLaunchKernelGGL(unpack_slots_kernel, dim3((unsigned)std::min<long long>((nnz + 255) / 256, 8192)), dim3(256), 0, s,
                           nnz, key, val, m->cols, m->vals);
        e = hipGetLastError();
    }
    if (e != hipSuccess) {
        set_error("slot unpacking failed: %s", hipGetErrorString(e));
        padne_csr_destroy(m);
        return PADNE_E_HIP;
    }
    *out = m;
    return PADNE_OK;
}

}  // namespace padne

using namespace padne;

extern "C" int padne_assemble_system_ex(padne_ctx *ctx, int64_t n_unknowns, int64_t n_vert, const double *xy_host,
                                        int64_t n_tri, const int32_t *tri_host, int64_t n_mesh,
                                        const int64_t *mesh_vertex_offset, const int64_t *mesh_tri_offset,
                                        const double *conductance, int64_t n_coo, const int64_t *coo_row,
                                        const int64_t *coo_col, const double *coo_val, int32_t flags, padne_csr **out) {
    struct Guard {
        ~Guard() { t_partial_mesh = false; }
    } guard;
    t_partial_mesh = (flags & 1) != 0;
    return padne_assemble_system(ctx, n_unknowns, n_vert, xy_host, n_tri, tri_host, n_mesh, mesh_vertex_offset,
                                 mesh_tri_offset, conductance, n_coo, coo_row, coo_col, coo_val, out);
}

extern "C" int padne_assemble_system(padne_ctx *ctx, int64_t n_unknowns, int64_t n_vert, const double *xy_host,
                                     int64_t n_tri, const int32_t *tri_host, int64_t n_mesh,
                                     const int64_t *mesh_vertex_offset, const int64_t *mesh_tri_offset,
                                     const double *conductance, int64_t n_coo, const int64_t *coo_row,
                                     const int64_t *coo_col, const double *coo_val, padne_csr **out) {
    PADNE_REQUIRE(ctx && out, "null argument");
    PADNE_REQUIRE(n_unknowns >= 0 && n_vert >= 0 && n_tri >= 0 && n_mesh >= 0 && n_coo >= 0, "negative size");
    PADNE_REQUIRE(n_vert <= n_unknowns, "more vertices than unknowns");
    PADNE_REQUIRE(n_unknowns < 2147483647LL, "too many unknowns for int32 indices");
    PADNE_REQUIRE(n_vert == 0 || xy_host, "xy");
    PADNE_REQUIRE(n_tri == 0 || tri_host, "tri");
    PADNE_REQUIRE(n_mesh == 0 || (mesh_vertex_offset && mesh_tri_offset && conductance), "mesh tables");
    PADNE_REQUIRE(n_coo == 0 || (coo_row && coo_col && coo_val), "coo arrays");
    PADNE_REQUIRE(n_mesh > 0 || (n_vert == 0 && n_tri == 0), "vertices without a mesh");
    if (n_mesh > 0) {
        PADNE_REQUIRE(mesh_vertex_offset[0] == 0 && mesh_tri_offset[0] == 0, "offset tables must start at 0");
        PADNE_REQUIRE(mesh_vertex_offset[n_mesh] == n_vert && mesh_tri_offset[n_mesh] == n_tri,
                      "offset tables must end at n_vert / n_tri");
        for (int64_t m = 0; m < n_mesh; ++m)
            PADNE_REQUIRE(mesh_vertex_offset[m] <= mesh_vertex_offset[m + 1] &&
                              mesh_tri_offset[m] <= mesh_tri_offset[m + 1], "offset tables not monotone");
    }
    std::vector<int32_t> row32((size_t)n_coo), col32((size_t)n_coo);
    for (int64_t k = 0; k < n_coo; ++k) {
        PADNE_REQUIRE(coo_row[k] >= 0 && coo_row[k] < n_unknowns && coo_col[k] >= 0 && coo_col[k] < n_unknowns,
                      "stamp index out of range");
        row32[(size_t)k] = (int32_t)coo_row[k];
        col32[(size_t)k] = (int32_t)coo_col[k];
    }
    PADNE_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    Scratch sc(ctx);
    double *d_xy = nullptr, *d_sigma = nullptr, *d_cval = nullptr;
    int *d_tri = nullptr, *d_crow = nullptr, *d_ccol = nullptr, *d_cnt = nullptr, *d_slot = nullptr, *d_err = nullptr;
    long long *d_voff = nullptr, *d_toff = nullptr;
    // the mesh arrays outlive this call: they are handed to the matrix at the end (MeshKeep frees them on any error path)
    struct MeshKeep {
        padne_ctx *ctx;
        void *p[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
        bool released = false;
        ~MeshKeep() { if (!released) for (void *q : p) pool_free(ctx, q); }
    } keep{ctx};
    keep.p[0] = d_xy = (double *)pool_alloc(ctx, sizeof(double) * ((size_t)n_vert * 2 + 1));
    keep.p[1] = d_tri = (int *)pool_alloc(ctx, sizeof(int) * ((size_t)n_tri * 3 + 1));
    keep.p[2] = d_sigma = (double *)pool_alloc(ctx, sizeof(double) * ((size_t)n_mesh + 1));
    keep.p[3] = d_voff = (long long *)pool_alloc(ctx, sizeof(long long) * ((size_t)n_mesh + 1));
    keep.p[4] = d_toff = (long long *)pool_alloc(ctx, sizeof(long long) * ((size_t)n_mesh + 1));
    if (!d_xy || !d_tri || !d_sigma || !d_voff || !d_toff) return PADNE_E_NOMEM;
    PADNE_TRY(sc.alloc(&d_crow, (size_t)n_coo));
    PADNE_TRY(sc.alloc(&d_ccol, (size_t)n_coo));
    PADNE_TRY(sc.alloc(&d_cval, (size_t)n_coo));
    PADNE_TRY(sc.alloc(&d_cnt, (size_t)n_unknowns + 1));
    PADNE_TRY(sc.alloc(&d_slot, (size_t)n_unknowns + 1));
    PADNE_TRY(sc.alloc(&d_err, (size_t)ERR_WORDS));
    const long long zero_off[1] = {0};
    // The two big arrays may already live on the device (padne_generate_grid_mesh, a caller with device-resident meshes).
    // Then the kernels read the CALLER's arrays, and the copy the matrix keeps for the post-processing (400 MB at N = 10 M:
    // 0.19 ms of copy engine in front of the first kernel) is made on the context's second stream beside them; it is joined
    // before this call returns.  Host arrays cross PCIe into the kept arrays first, as before.
    auto on_device = [&](const void *p) {
        hipPointerAttribute_t at;
        if (p == nullptr || hipPointerGetAttributes(&at, p) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        return at.type == hipMemoryTypeDevice && at.device == ctx->device;
    };
    padne_ctx *side = nullptr;
    if (n_vert > 0 && n_tri > 0 && on_device(xy_host) && on_device(tri_host) && !ctx->is_aux) side = aux_context(ctx);
    double *keep_xy = d_xy;
    int *keep_tri = d_tri;
    if (side != nullptr) {
        PADNE_TRY(stream_order(ctx, side));                  // (what filled the caller's arrays was queued on this context's stream, or is complete)
        PADNE_HIP_CHECK(hipMemcpyAsync(keep_xy, xy_host, sizeof(double) * 2 * (size_t)n_vert, hipMemcpyDeviceToDevice, side->stream));
        PADNE_HIP_CHECK(hipMemcpyAsync(keep_tri, tri_host, sizeof(int) * 3 * (size_t)n_tri, hipMemcpyDeviceToDevice, side->stream));
        d_xy = const_cast<double *>(xy_host);               // read-only from here on
        d_tri = const_cast<int *>(tri_host);
    } else {
        PADNE_HIP_CHECK(hipMemcpyAsync(d_xy, xy_host, sizeof(double) * 2 * (size_t)n_vert, hipMemcpyDefault, s));
        PADNE_HIP_CHECK(hipMemcpyAsync(d_tri, tri_host, sizeof(int) * 3 * (size_t)n_tri, hipMemcpyDefault, s));
    }
    struct SideJoin {                                       // every return path waits for the side stream's copies: they read
        padne_ctx *side;                                    // the caller's arrays and write blocks MeshKeep may hand back
        ~SideJoin() { if (side != nullptr) (void)hipStreamSynchronize(side->stream); }
    } side_join{side};
    if (n_mesh > 0) {
        PADNE_HIP_CHECK(hipMemcpyAsync(d_sigma, conductance, sizeof(double) * (size_t)n_mesh, hipMemcpyHostToDevice, s));
        PADNE_HIP_CHECK(hipMemcpyAsync(d_voff, mesh_vertex_offset, sizeof(long long) * (size_t)(n_mesh + 1), hipMemcpyHostToDevice, s));
        PADNE_HIP_CHECK(hipMemcpyAsync(d_toff, mesh_tri_offset, sizeof(long long) * (size_t)(n_mesh + 1), hipMemcpyHostToDevice, s));
    } else {
        PADNE_HIP_CHECK(hipMemcpyAsync(d_voff, zero_off, sizeof(long long), hipMemcpyHostToDevice, s));
        PADNE_HIP_CHECK(hipMemcpyAsync(d_toff, zero_off, sizeof(long long), hipMemcpyHostToDevice, s));
    }
    if (n_coo > 0) {
        PADNE_HIP_CHECK(hipMemcpyAsync(d_crow, row32.data(), sizeof(int) * (size_t)n_coo, hipMemcpyHostToDevice, s));
        PADNE_HIP_CHECK(hipMemcpyAsync(d_ccol, col32.data(), sizeof(int) * (size_t)n_coo, hipMemcpyHostToDevice, s));
        PADNE_HIP_CHECK(hipMemcpyAsync(d_cval, coo_val, sizeof(double) * (size_t)n_coo, hipMemcpyHostToDevice, s));
    }
    PADNE_HIP_CHECK(hipMemsetAsync(d_err, 0, sizeof(int) * ERR_WORDS, s));
    // 1 one pass over the triangles: validation, incident triangles per vertex and their lists; stamps per row
    int *d_ninc = nullptr, *d_ncoo = nullptr, *d_rowlen = nullptr, *d_list = nullptr, *d_fans = nullptr, *d_nlisted = nullptr;
    int2 *d_inc = nullptr;
    PADNE_TRY(sc.alloc(&d_ninc, (size_t)n_unknowns + 1));
    PADNE_TRY(sc.alloc(&d_ncoo, (size_t)n_unknowns + 1));
    PADNE_TRY(sc.alloc(&d_inc, (size_t)n_vert * kIncCap + 2));
    PADNE_TRY(sc.alloc(&d_rowlen, (size_t)n_unknowns + 1));
    PADNE_TRY(sc.alloc(&d_list, (size_t)n_unknowns + 1));
    PADNE_TRY(sc.alloc(&d_fans, (size_t)n_vert + 1));
    PADNE_TRY(sc.alloc(&d_nlisted, 2));
    PADNE_HIP_CHECK(hipMemsetAsync(d_ninc, 0, sizeof(int) * (size_t)(n_unknowns + 1), s));
    PADNE_HIP_CHECK(hipMemsetAsync(d_ncoo, 0, sizeof(int) * (size_t)(n_unknowns + 1), s));
    PADNE_HIP_CHECK(hipMemsetAsync(d_nlisted, 0, sizeof(int) * 2, s));
    if (n_tri > 0)
        hipLaunchKernelGGL(asm_count_tri, dim3(nblk(n_tri, 256 * kCntTris)), dim3(256), 0, s, (long long)n_tri, d_tri, (int)n_mesh,
                           d_voff, d_toff, d_ninc, d_inc, d_err, ctx->opt.force_asm_hash ? 1 : 0);
    if (n_coo > 0)
        hipLaunchKernelGGL(asm_count_coo, dim3(nblk(n_coo)), dim3(256), 0, s, (long long)n_coo, d_crow, d_ncoo);
    // 2 the rows that go through the slots (stamps, hubs, the unknowns behind the vertices; long fans): lists, slot counts, offsets
    hipLaunchKernelGGL(asm_classify_rows, dim3(nblk((n_unknowns + 4) / 4)), dim3(256), 0, s, (long long)n_unknowns + 1,
                       (long long)n_unknowns, (long long)n_vert, d_ninc, d_ncoo, d_cnt, d_list, d_fans, d_nlisted, d_err);
    PADNE_HIP_CHECK(hipGetLastError());
    int64_t n_slots = 0;
    PADNE_TRY(exclusive_scan_i32(ctx, d_cnt, d_slot, n_unknowns, &n_slots));
    int h_err[ERR_WORDS];
    int h_listed[2] = {0, 0};
    PADNE_TRY(read_back2(ctx, d_err, sizeof(h_err), h_err, d_nlisted, sizeof(h_listed), h_listed));
    if (h_err[ERR_BAD_INDEX]) {
        set_error("triangle refers to a vertex outside its mesh, or repeats a vertex");
        return PADNE_E_INVALID;
    }
    const int h_slow = h_listed[0], h_fans = h_listed[1];
    long long *d_key = nullptr;
    double *d_val = nullptr;
    PADNE_TRY(sc.alloc(&d_key, (size_t)n_slots + 1));
    PADNE_TRY(sc.alloc(&d_val, (size_t)n_slots + 1));
    // 3 the listed rows: mesh terms from the incidence lists (the triangles again only when a hub's list is incomplete),
    //   stamps behind them, merged in place at their slot offsets.  d_cnt has been scanned; it now serves as the cursor.
    if (h_fans > 0)
        hipLaunchKernelGGL(asm_rows_long_fans, dim3(nblk(h_fans, 128)), dim3(128), 0, s, (const int *)(d_nlisted + 1),
                           (const int *)d_fans, (long long)n_vert, (int)n_mesh, d_voff, d_sigma, d_xy, d_ninc, d_inc, d_slot, d_key, d_val, d_rowlen,
                           d_err);
    if (h_slow > 0) {
        hipLaunchKernelGGL(asm_fill_listed, dim3(nblk(h_slow)), dim3(256), 0, s, (const int *)d_nlisted, (const int *)d_list,
                           (long long)n_vert, d_xy, d_ninc, d_inc, d_slot, d_cnt, d_key, d_val);
        if (h_err[ERR_HUB] && n_tri > 0)
            hipLaunchKernelGGL(asm_fill_tri, dim3(nblk(n_tri)), dim3(256), 0, s, (long long)n_tri, d_tri, d_xy, (int)n_mesh,
                               d_voff, d_toff, d_slot, d_cnt, d_key, d_val, (const int *)d_ninc);
        if (n_coo > 0)
            hipLaunchKernelGGL(asm_fill_coo, dim3(nblk(n_coo)), dim3(256), 0, s, (long long)n_coo, d_crow, d_ccol, d_cval,
                               d_slot, d_cnt, d_key, d_val);
        PADNE_HIP_CHECK(hipGetLastError());
        PADNE_TRY(merge_listed_mesh_rows(ctx, (long long)n_vert, (int)n_mesh, d_voff, d_sigma, d_slot, d_key, d_val, d_rowlen,
                                         d_err, d_list, d_nlisted, (long long)h_slow));
    }
    // 4 every row, written once and in place: a mesh row holds at most one entry per triangle plus two, a listed row at
    //   most its slots -- the arrays are sized by that bound, the single-pass scan inside the kernel finds the offsets
    const long long nnz_bound = 3LL * n_tri + 2LL * n_vert + n_slots;
    if (nnz_bound >= 2147483647LL - kPadNnz) {
        set_error("%lld entries exceed the 32-bit index space", nnz_bound);
        return PADNE_E_TOOLARGE;
    }
    padne_csr *m = nullptr;
    PADNE_TRY(csr_alloc(ctx, n_unknowns, n_unknowns, nnz_bound, &m));
    long long h_nnz = 0;
    long long *d_nnz = nullptr;
    int rc = PADNE_OK;
    if (n_unknowns > 0) {
        const long long n_tiles = (n_unknowns + 127) / 128, n_chunks = (n_tiles + kChunkTiles - 1) / kChunkTiles;
        // one block of scan state: tile counts, chunk counts, chunk offsets (64-bit words), arrivals per chunk, ticket, abort
        const size_t words64 = (size_t)n_tiles + 2 * (size_t)n_chunks + 64 + 2;
        const size_t state_bytes = words64 * 8 + ((size_t)n_chunks + 16 + (size_t)kTicketSeqs * kTicketStride + 16) * 4;
        unsigned char *d_state = nullptr;
        rc = sc.alloc(&d_state, state_bytes);
        hipError_t e = hipSuccess;
        if (rc == PADNE_OK) e = hipMemsetAsync(d_state, 0, state_bytes, s);
        if (rc == PADNE_OK && e == hipSuccess) {
            RowsInPlace args;
            args.scan.tile_agg = (unsigned long long *)d_state;
            args.scan.chunk_agg = args.scan.tile_agg + n_tiles;
            args.scan.chunk_prefix = args.scan.chunk_agg + n_chunks + 64;     // (the scanner looks 64 chunks ahead)
            args.scan.nnz_out = (long long *)(args.scan.chunk_prefix + n_chunks + 1);
            args.scan.chunk_count = (int *)(d_state + words64 * 8);
            args.scan.ticket = args.scan.chunk_count + ((n_chunks + 15) / 16) * 16;      // kTicketSeqs counters, a line apart
            args.scan.abort_word = args.scan.ticket + kTicketSeqs * kTicketStride;
            args.scan.n_tiles = (int)n_tiles;
            args.scan.n_chunks = (int)n_chunks;
            // as many workers as the chip holds at once (a worker that is not resident simply takes no tickets), one scanner
            int per_cu = 0, n_cu = 0;
            e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, asm_rows_in_place, 128, 0);
            if (e == hipSuccess) e = hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, ctx->device);
            const long long resident = (long long)std::max(per_cu, 1) * std::max(n_cu, 1);
            const unsigned workers = (unsigned)std::max(1LL, std::min(n_tiles, resident - 1));
            args.n_rows = n_unknowns;
            args.n_vert = n_vert;
            args.n_mesh = (int)n_mesh;
            args.mesh_voff = d_voff;
            args.sigma = d_sigma;
            args.xy = d_xy;
            args.n_inc = d_ninc;
            args.inc = d_inc;
            args.slot_ptr = d_slot;
            args.row_len = d_rowlen;
            args.rowptr = m->rowptr;
            args.cols = m->cols;
            args.vals = m->vals;
            args.err = d_err;
            args.nnz_cap = nnz_bound;
            d_nnz = args.scan.nnz_out;
            auto place_listed_rows = [&]() {
                if (h_slow > 0)
                    hipLaunchKernelGGL(asm_place_listed, dim3(std::min(nblk(h_slow, 4), 4096u)), dim3(256), 0, s, (const int *)d_nlisted,
                                       (const int *)d_list, (const int *)d_slot, (const long long *)d_key, (const double *)d_val,
                                       (const int *)m->rowptr, m->cols, m->vals, nnz_bound, d_err);
                if (h_fans > 0)
                    hipLaunchKernelGGL(asm_place_listed, dim3(std::min(nblk(h_fans, 4), 4096u)), dim3(256), 0, s,
                                       (const int *)(d_nlisted + 1), (const int *)d_fans, (const int *)d_slot, (const long long *)d_key,
                                       (const double *)d_val, (const int *)m->rowptr, m->cols, m->vals, nnz_bound, d_err);
            };
            // PADNE_FORCE=asm_two_pass takes the second path at once (its test; a caller that knows the chip is oversubscribed)
            bool two_pass = ctx->opt.force_asm_two_pass;
            if (e == hipSuccess && !two_pass) {
                hipLaunchKernelGGL(asm_rows_in_place, dim3(workers + 1), dim3(128), 0, s, args);
                place_listed_rows();
                e = hipGetLastError();
                if (e == hipSuccess) {
                    rc = read_back2(ctx, d_err, sizeof(h_err), h_err, d_nnz, sizeof(long long), &h_nnz);
                    // gave up: the abort word is the diagnostic, not the result.  (What asm_place_listed refused on the
                    // unfinished row pointer is not an error either: both words are cleared for the second path, which
                    // finds a real inconsistency again.)
                    if (rc == PADNE_OK && h_err[ERR_GAVE_UP]) {
                        two_pass = true;
                        ++g_two_pass_fallbacks;
                        e = hipMemsetAsync(d_err + ERR_SCAN, 0, sizeof(int) * 2, s);
                        static_assert(ERR_GAVE_UP == ERR_SCAN + 1, "the two words are cleared together");
                    }
                }
            }
            if (rc == PADNE_OK && e == hipSuccess && two_pass) {
                if (!h_err[ERR_GAVE_UP]) ++g_two_pass_fallbacks;      // (forced; a give-up was counted above)
                h_err[ERR_GAVE_UP] = 0;
                int *d_len = nullptr;
                rc = sc.alloc(&d_len, (size_t)n_unknowns + 1);
                if (rc == PADNE_OK) {
                    hipLaunchKernelGGL(asm_rows_two_pass<false>, dim3(nblk(n_unknowns, 128)), dim3(128), 0, s, args, d_len);
                    e = hipGetLastError();
                }
                int64_t total = 0;
                if (rc == PADNE_OK && e == hipSuccess) rc = exclusive_scan_i32(ctx, d_len, m->rowptr, n_unknowns, &total);
                if (rc == PADNE_OK && e == hipSuccess && total > nnz_bound) {
                    set_error("assembled %lld entries where at most %lld fit", (long long)total, nnz_bound);
                    rc = PADNE_E_HIP;
                }
                if (rc == PADNE_OK && e == hipSuccess) {
                    hipLaunchKernelGGL(asm_rows_two_pass<true>, dim3(nblk(n_unknowns, 128)), dim3(128), 0, s, args, d_len);
                    place_listed_rows();
                    e = hipGetLastError();
                    h_nnz = total;
                }
                if (rc == PADNE_OK && e == hipSuccess) rc = read_back(ctx, d_err, sizeof(h_err), h_err);
            }
        }
        if (rc == PADNE_OK && e != hipSuccess) {
            set_error("row kernel failed: %s", hipGetErrorString(e));
            rc = PADNE_E_HIP;
        }
        if (rc == PADNE_OK && (h_err[ERR_SCAN] || h_err[ERR_GAVE_UP])) {
            set_error("the rows of the assembled system do not fit their offsets (inconsistent row lengths)");
            rc = PADNE_E_HIP;
        }
        if (rc == PADNE_OK && h_err[ERR_NONMANIFOLD] && !t_partial_mesh) {
            set_error("Non-manifold mesh");
            rc = PADNE_E_NONMANIFOLD;
        }
    } else {
        PADNE_HIP_CHECK(hipMemsetAsync(m->rowptr, 0, sizeof(int32_t), s));
    }
    if (rc == PADNE_OK && (h_nnz < 0 || h_nnz > nnz_bound)) {
        set_error("assembled %lld entries where at most %lld fit", h_nnz, nnz_bound);
        rc = PADNE_E_HIP;
    }
    if (rc == PADNE_OK) {
        // the padding behind the real end (column 0 / value 0.0, see csr_alloc), then the scratch goes back to the pool
        rc = csr_shrink_nnz(ctx, m, h_nnz);
        if (rc == PADNE_OK && hipStreamSynchronize(s) != hipSuccess) {
            set_error("assembly failed: %s", hipGetErrorString(hipGetLastError()));
            rc = PADNE_E_HIP;
        }
    }
    if (rc != PADNE_OK) {
        padne_csr_destroy(m);
        return rc;
    }
    *out = m;
    padne_csr *res = *out;
    res->mesh_xy = keep_xy;
    res->mesh_tri = keep_tri;
    res->mesh_sigma = d_sigma;
    res->mesh_voff = d_voff;
    res->mesh_toff = d_toff;
    res->mesh_n_vert = n_vert;
    res->mesh_n_tri = n_tri;
    res->mesh_n_mesh = n_mesh;
    keep.released = true;
    return PADNE_OK;
}

extern "C" int padne_asm_second_path_count(int64_t *count) {
    PADNE_REQUIRE(count != nullptr, "count");
    *count = (int64_t)g_two_pass_fallbacks.load();
    return PADNE_OK;
}

// Power density of a solution on the mesh the matrix was assembled from (kept on the device): uploads the potentials,
// downloads one value per triangle.  compute_power_density, solver.py:728-745, for all meshes in one launch.
extern "C" int padne_csr_power_density(padne_ctx *ctx, const padne_csr *m, const double *potential_host,
                                       double *power_out_host) {
    PADNE_REQUIRE(ctx && m, "null argument");
    PADNE_REQUIRE(m->mesh_n_mesh > 0 || m->mesh_n_tri == 0, "the matrix does not carry a mesh (only padne_assemble_system keeps it)");
    if (m->mesh_n_tri == 0) return PADNE_OK;
    PADNE_REQUIRE(potential_host && power_out_host, "null argument");
    PADNE_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    Scratch sc(ctx);
    double *d_pot = nullptr, *d_out = nullptr;
    int *d_bad = nullptr;
    PADNE_TRY(sc.alloc(&d_pot, (size_t)m->mesh_n_vert));
    PADNE_TRY(sc.alloc(&d_out, (size_t)m->mesh_n_tri));
    PADNE_TRY(sc.alloc(&d_bad, 1));
    PADNE_HIP_CHECK(hipMemsetAsync(d_bad, 0, sizeof(int), s));
    PADNE_HIP_CHECK(hipMemcpyAsync(d_pot, potential_host, sizeof(double) * (size_t)m->mesh_n_vert, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(power_density_kernel, dim3(nblk(m->mesh_n_tri)), dim3(256), 0, s, (long long)m->mesh_n_tri, m->mesh_tri,
                       m->mesh_xy, (int)m->mesh_n_mesh, m->mesh_voff, m->mesh_toff, m->mesh_sigma, d_pot, d_out,
                       (double *)nullptr, (double *)nullptr, d_bad);
    PADNE_HIP_CHECK(hipGetLastError());
    PADNE_HIP_CHECK(hipMemcpyAsync(power_out_host, d_out, sizeof(double) * (size_t)m->mesh_n_tri, hipMemcpyDeviceToHost, s));
    PADNE_HIP_CHECK(hipStreamSynchronize(s));
    return PADNE_OK;
}

// out = scale * R^T M C: entry (i, j, v) becomes (row_map[i], col_map[j], scale*v) when both maps are >= 0,
// duplicates are added in the order of their position in M.
// The maps may be host or device arrays (hipMemcpyDefault): padne_kkt_create builds its map on the device.
namespace padne {
int csr_relabel(padne_ctx *ctx, const padne_csr *m, const int32_t *row_map_host, int64_t n_rows_out,
                const int32_t *col_map_host, int64_t n_cols_out, double scale, padne_csr **out) {
    PADNE_REQUIRE(n_rows_out >= 0 && n_rows_out < 2147483647LL && n_cols_out >= 0 && n_cols_out < 2147483647LL,
                  "output shape");
    PADNE_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    Scratch sc(ctx);
    int *d_map = nullptr, *d_cmap = nullptr, *d_cnt = nullptr, *d_slot = nullptr, *d_err = nullptr;
    PADNE_TRY(sc.alloc(&d_map, (size_t)m->n_rows));
    PADNE_TRY(sc.alloc(&d_cnt, (size_t)n_rows_out + 1));
    PADNE_TRY(sc.alloc(&d_slot, (size_t)n_rows_out + 1));
    PADNE_TRY(sc.alloc(&d_err, (size_t)ERR_WORDS));
    // a map that already lies on this device is read where it is (padne_kkt_create builds its maps there; a caller that
    // reduces one layout again and again keeps the map resident): no 40 MB copy per call at 10 M unknowns
    auto on_this_device = [&](const int32_t *p) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, p) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        return at.type == hipMemoryTypeDevice && at.device == ctx->device;
    };
    if (on_this_device(row_map_host)) {
        d_map = const_cast<int *>(row_map_host);
    } else {
        PADNE_HIP_CHECK(hipMemcpyAsync(d_map, row_map_host, sizeof(int) * (size_t)m->n_rows, hipMemcpyDefault, s));
    }
    if (col_map_host == row_map_host) {
        d_cmap = d_map;
    } else if (on_this_device(col_map_host)) {
        d_cmap = const_cast<int *>(col_map_host);
    } else {
        PADNE_TRY(sc.alloc(&d_cmap, (size_t)m->n_cols));
        PADNE_HIP_CHECK(hipMemcpyAsync(d_cmap, col_map_host, sizeof(int) * (size_t)m->n_cols, hipMemcpyDefault, s));
    }
    PADNE_HIP_CHECK(hipMemsetAsync(d_err, 0, sizeof(int) * ERR_WORDS, s));
    if (m->n_rows > 0 && m->n_cols > 0 && n_rows_out > 0 && n_cols_out > 0 && !ctx->opt.force_relabel_slots && !m->cols_unsorted) {
        // (a copy keeps the order of a row's entries: the source's columns must ascend, as everything built on the device
        // does and padne_csr_from_host checks for what it uploads)
        // maps that only drop indices (the reduction to the potential block): count, scan, copy -- see map_is_compaction
        static_assert(ERR_WORDS >= 6, "two triples of flag words");
        hipLaunchKernelGGL(map_is_compaction, dim3(nblk(m->n_rows)), dim3(256), 0, s, (long long)m->n_rows, (const int *)d_map,
                           (int)n_rows_out, d_err);
        if (d_cmap != d_map)
            hipLaunchKernelGGL(map_is_compaction, dim3(nblk(m->n_cols)), dim3(256), 0, s, (long long)m->n_cols,
                               (const int *)d_cmap, (int)n_cols_out, d_err + 3);
        // (the counts are those of any one-to-one map; taken before the verdict is known, for one look at the host less)
        // (a wave per 64 rows, 2048 workgroups sweep the tiles; the lane-per-row kernels relabel_count_ordered /
        // relabel_fill_ordered remain for PADNE_FORCE=relabel_lanes: the tests compare the two)
        const int g_wave = (int)std::min<long long>(2048, ((m->n_rows + 63) / 64 + 3) / 4);
        if (ctx->opt.force_relabel_lanes)
            hipLaunchKernelGGL(relabel_count_ordered, dim3(nblk(m->n_rows)), dim3(256), 0, s, (long long)m->n_rows, m->rowptr, m->cols,
                               (const int *)d_map, (const int *)d_cmap, (int)n_rows_out, d_cnt);
        else
            hipLaunchKernelGGL(relabel_count_wave, dim3(g_wave), dim3(256), 0, s, (int)m->n_rows, m->rowptr, m->cols,
                               (const int *)d_map, (const int *)d_cmap, (int)n_rows_out, d_cnt);
        PADNE_HIP_CHECK(hipGetLastError());
        int h_flags[6] = {0, 0, 0, 0, 0, 0};
        PADNE_TRY(read_back(ctx, d_err, sizeof(h_flags), h_flags));
        if (h_flags[1] || h_flags[4]) {
            set_error("invalid argument: index map entry out of range");
            return PADNE_E_INVALID;
        }
        const bool rows_ok = h_flags[0] == 0 && h_flags[2] == 1;
        const bool cols_ok = d_cmap == d_map ? rows_ok && n_rows_out <= n_cols_out : (h_flags[3] == 0 && h_flags[5] == 1);
        if (rows_ok && cols_ok) {
            int64_t nnz = 0;
            PADNE_TRY(exclusive_scan_i32(ctx, d_cnt, d_slot, n_rows_out, &nnz));
            padne_csr *res = nullptr;
            PADNE_TRY(csr_alloc(ctx, n_rows_out, n_cols_out, nnz, &res));
            hipError_t e = hipMemcpyAsync(res->rowptr, d_slot, sizeof(int32_t) * (size_t)(n_rows_out + 1), hipMemcpyDeviceToDevice, s);
            if (e == hipSuccess) {
                if (ctx->opt.force_relabel_lanes)
                    hipLaunchKernelGGL(relabel_fill_ordered, dim3(nblk(m->n_rows)), dim3(256), 0, s, (long long)m->n_rows, m->rowptr,
                                       m->cols, m->vals, (const int *)d_map, (const int *)d_cmap, scale, (const int *)res->rowptr,
                                       res->cols, res->vals, d_err + 6);
                else
                    hipLaunchKernelGGL(relabel_fill_wave, dim3(g_wave), dim3(256), 0, s, (int)m->n_rows, m->rowptr, m->cols, m->vals,
                                       (const int *)d_map, (const int *)d_cmap, scale, (const int *)res->rowptr, res->cols,
                                       res->vals, d_err + 6);
                e = hipGetLastError();
            }
            int h_zero = 0;
            int rc = e == hipSuccess ? read_back(ctx, d_err + 6, sizeof(int), &h_zero) : PADNE_E_HIP;
            if (rc != PADNE_OK) {
                if (e != hipSuccess) set_error("relabel failed: %s", hipGetErrorString(e));
                (void)hipStreamSynchronize(s);
                padne_csr_destroy(res);
                return rc;
            }
            if (h_zero == 0) {
                *out = res;
                return PADNE_OK;
            }
            padne_csr_destroy(res);           // explicit zeros in the source: the slot path drops them
        }
        PADNE_HIP_CHECK(hipMemsetAsync(d_err, 0, sizeof(int) * ERR_WORDS, s));
    }
    PADNE_HIP_CHECK(hipMemsetAsync(d_cnt, 0, sizeof(int) * (size_t)(n_rows_out + 1), s));
    {
        // injective row and column maps (everything but tied groups of unknowns): direct relabel, no slots
        int *d_hist = nullptr;
        const long long n_hist = std::max<long long>(n_rows_out, n_cols_out) + 1;
        PADNE_TRY(sc.alloc(&d_hist, (size_t)n_hist));
        PADNE_HIP_CHECK(hipMemsetAsync(d_hist, 0, sizeof(int) * (size_t)n_hist, s));
        if (m->n_rows > 0)
            hipLaunchKernelGGL(map_is_injective, dim3(nblk(m->n_rows)), dim3(256), 0, s, (long long)m->n_rows, d_map,
                               (int)n_rows_out, d_hist, d_err + ERR_LONG_ROWS);
        if (d_cmap != d_map && m->n_cols > 0) {
            PADNE_HIP_CHECK(hipMemsetAsync(d_hist, 0, sizeof(int) * (size_t)n_hist, s));
            hipLaunchKernelGGL(map_is_injective, dim3(nblk(m->n_cols)), dim3(256), 0, s, (long long)m->n_cols, d_cmap,
                               (int)n_cols_out, d_hist, d_err + ERR_LONG_ROWS);
        } else if (d_cmap == d_map) {
            PADNE_REQUIRE(n_rows_out <= n_cols_out, "shared index map needs n_rows_out <= n_cols_out");
        }
        PADNE_HIP_CHECK(hipGetLastError());
        int h_flags[2] = {0, 0};
        static_assert(ERR_LONG_ROWS + 1 < ERR_WORDS, "two flag words");
        PADNE_TRY(read_back(ctx, d_err + ERR_LONG_ROWS, sizeof(h_flags), h_flags));
        if (h_flags[1]) {
            set_error("invalid argument: index map entry out of range");
            return PADNE_E_INVALID;
        }
        const int h_dup = h_flags[0];
        if (h_dup == 0 && !ctx->opt.force_relabel_slots) {
            if (m->n_rows > 0)
                hipLaunchKernelGGL(relabel_count_direct, dim3(nblk(m->n_rows)), dim3(256), 0, s, (long long)m->n_rows, m->rowptr,
                                   m->cols, d_map, d_cmap, d_cnt);
            PADNE_HIP_CHECK(hipGetLastError());
            int64_t nnz = 0;
            PADNE_TRY(exclusive_scan_i32(ctx, d_cnt, d_slot, n_rows_out, &nnz));
            padne_csr *res = nullptr;
            PADNE_TRY(csr_alloc(ctx, n_rows_out, n_cols_out, nnz, &res));
            hipError_t e = hipMemcpyAsync(res->rowptr, d_slot, sizeof(int32_t) * (size_t)(n_rows_out + 1), hipMemcpyDeviceToDevice, s);
            if (e == hipSuccess && m->n_rows > 0) {
                hipLaunchKernelGGL(relabel_fill_direct<16>, dim3(nblk(m->n_rows, 128)), dim3(128), 0, s, (long long)m->n_rows,
                                   m->rowptr, m->cols, m->vals, d_map, d_cmap, scale, res->rowptr, res->cols, res->vals,
                                   d_err + ERR_LONG_ROWS);
                e = hipGetLastError();
            }
            int h_zero = 0;
            if (e == hipSuccess) e = hipMemcpyAsync(&h_zero, d_err + ERR_LONG_ROWS, sizeof(int), hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) {
                set_error("relabel failed: %s", hipGetErrorString(e));
                padne_csr_destroy(res);
                return PADNE_E_HIP;
            }
            if (h_zero == 0) {
                *out = res;
                return PADNE_OK;
            }
            padne_csr_destroy(res);           // explicit zeros in the source: the slot path drops them
            PADNE_HIP_CHECK(hipMemsetAsync(d_cnt, 0, sizeof(int) * (size_t)(n_rows_out + 1), s));
        }
        PADNE_HIP_CHECK(hipMemsetAsync(d_err, 0, sizeof(int) * ERR_WORDS, s));
    }
    if (m->n_rows > 0)
        hipLaunchKernelGGL(reduce_count, dim3(nblk(m->n_rows)), dim3(256), 0, s, (long long)m->n_rows, m->rowptr,
                           m->cols, d_map, d_cmap, d_cnt);
    PADNE_HIP_CHECK(hipGetLastError());
    int64_t n_slots = 0;
    PADNE_TRY(exclusive_scan_i32(ctx, d_cnt, d_slot, n_rows_out, &n_slots));
    long long *d_key = nullptr;
    double *d_val = nullptr;
    PADNE_TRY(sc.alloc(&d_key, (size_t)n_slots));
    PADNE_TRY(sc.alloc(&d_val, (size_t)n_slots));
    PADNE_HIP_CHECK(hipMemsetAsync(d_cnt, 0, sizeof(int) * (size_t)(n_rows_out + 1), s));
    if (m->n_rows > 0)
        hipLaunchKernelGGL(reduce_fill, dim3(nblk(m->n_rows)), dim3(256), 0, s, (long long)m->n_rows, m->rowptr, m->cols,
                           m->vals, d_map, d_cmap, scale, d_slot, d_cnt, d_key, d_val);
    PADNE_HIP_CHECK(hipGetLastError());
    return finish_rows(ctx, sc, n_rows_out, n_cols_out, d_slot, d_key, d_val, out);
}
}  // namespace padne

extern "C" int padne_csr_relabel(padne_ctx *ctx, const padne_csr *m, const int32_t *row_map_host, int64_t n_rows_out,
                                 const int32_t *col_map_host, int64_t n_cols_out, double scale, padne_csr **out) {
    PADNE_REQUIRE(ctx && m && row_map_host && col_map_host && out, "null argument");
    return csr_relabel(ctx, m, row_map_host, n_rows_out, col_map_host, n_cols_out, scale, out);
}

// rows of `top` followed by the rows of `bottom` (same number of columns)
__global__ void vstack_shift(int n, const int *__restrict__ src, int shift, int *__restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i] + shift;
}

namespace padne {
int csr_vstack(padne_ctx *ctx, const padne_csr *top, const padne_csr *bottom, int64_t n_cols, padne_csr **out) {
    PADNE_REQUIRE(top->n_cols <= n_cols && bottom->n_cols <= n_cols, "column count");
    PADNE_REQUIRE(top->nnz + bottom->nnz < 2147483647LL, "too many entries");
    hipStream_t s = ctx->stream;
    padne_csr *m = nullptr;
    PADNE_TRY(csr_alloc(ctx, top->n_rows + bottom->n_rows, n_cols, top->nnz + bottom->nnz, &m));
    hipError_t e = hipMemcpyAsync(m->rowptr, top->rowptr, sizeof(int) * (size_t)(top->n_rows + 1), hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(vstack_shift, dim3(nblk(bottom->n_rows + 1)), dim3(256), 0, s, (int)bottom->n_rows + 1,
                           bottom->rowptr, (int)top->nnz, m->rowptr + top->n_rows);
        e = hipGetLastError();
    }
    if (e == hipSuccess && top->nnz > 0) {
        e = hipMemcpyAsync(m->cols, top->cols, sizeof(int) * (size_t)top->nnz, hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(m->vals, top->vals, sizeof(double) * (size_t)top->nnz, hipMemcpyDeviceToDevice, s);
    }
    if (e == hipSuccess && bottom->nnz > 0) {
        e = hipMemcpyAsync(m->cols + top->nnz, bottom->cols, sizeof(int) * (size_t)bottom->nnz, hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess)
            e = hipMemcpyAsync(m->vals + top->nnz, bottom->vals, sizeof(double) * (size_t)bottom->nnz, hipMemcpyDeviceToDevice, s);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
        set_error("vstack failed: %s", hipGetErrorString(e));
        padne_csr_destroy(m);
        return PADNE_E_HIP;
    }
    *out = m;
    return PADNE_OK;
}
}  // namespace padne

extern "C" int padne_csr_vstack(padne_ctx *ctx, const padne_csr *top, const padne_csr *bottom, padne_csr **out) {
    PADNE_REQUIRE(ctx && top && bottom && out, "null argument");
    PADNE_REQUIRE(top->n_cols == bottom->n_cols, "column counts differ");
    PADNE_HIP_CHECK(hipSetDevice(ctx->device));
    return csr_vstack(ctx, top, bottom, top->n_cols, out);
}

extern "C" int padne_csr_reduce(padne_ctx *ctx, const padne_csr *m, const int32_t *map_host, int64_t n_out,
                                double scale, padne_csr **out) {
    PADNE_REQUIRE(ctx && m && map_host && out, "null argument");
    PADNE_REQUIRE(m->n_rows == m->n_cols, "matrix must be square");
    return csr_relabel(ctx, m, map_host, n_out, map_host, n_out, scale, out);
}

static int face_fields(padne_ctx *ctx, int64_t n_vert, const double *xy_host, int64_t n_tri,
                       const int32_t *tri_host, int64_t n_mesh, const int64_t *mesh_vertex_offset,
                       const int64_t *mesh_tri_offset, const double *conductance, const double *potential_host,
                       double *power_out_host, double *gx_out_host, double *gy_out_host) {
    PADNE_REQUIRE(ctx, "ctx");
    PADNE_REQUIRE(n_vert >= 0 && n_tri >= 0 && n_mesh >= 0, "negative size");
    if (n_tri == 0) return PADNE_OK;
    PADNE_REQUIRE(xy_host && tri_host && mesh_vertex_offset && mesh_tri_offset && potential_host && n_mesh > 0,
                  "null argument");
    PADNE_REQUIRE(power_out_host == nullptr || conductance != nullptr, "conductance");
    PADNE_REQUIRE(mesh_vertex_offset[n_mesh] == n_vert && mesh_tri_offset[n_mesh] == n_tri, "offset tables");
    PADNE_REQUIRE(mesh_vertex_offset[0] == 0 && mesh_tri_offset[0] == 0, "offset tables must start at 0");
    for (int64_t m = 0; m < n_mesh; ++m)
        PADNE_REQUIRE(mesh_vertex_offset[m] <= mesh_vertex_offset[m + 1] && mesh_tri_offset[m] <= mesh_tri_offset[m + 1],
                      "offset tables not monotone");
    PADNE_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    Scratch sc(ctx);
    double *d_xy = nullptr, *d_sigma = nullptr, *d_pot = nullptr, *d_out = nullptr, *d_gx = nullptr, *d_gy = nullptr;
    int *d_tri = nullptr, *d_bad = nullptr;
    long long *d_voff = nullptr, *d_toff = nullptr;
    PADNE_TRY(sc.alloc(&d_bad, 1));
    PADNE_HIP_CHECK(hipMemsetAsync(d_bad, 0, sizeof(int), s));
    PADNE_TRY(sc.alloc(&d_xy, (size_t)n_vert * 2));
    PADNE_TRY(sc.alloc(&d_tri, (size_t)n_tri * 3));
    PADNE_TRY(sc.alloc(&d_sigma, (size_t)n_mesh));
    PADNE_TRY(sc.alloc(&d_voff, (size_t)n_mesh + 1));
    PADNE_TRY(sc.alloc(&d_toff, (size_t)n_mesh + 1));
    PADNE_TRY(sc.alloc(&d_pot, (size_t)n_vert));
    if (power_out_host) PADNE_TRY(sc.alloc(&d_out, (size_t)n_tri));
    if (gx_out_host) {
        PADNE_TRY(sc.alloc(&d_gx, (size_t)n_tri));
        PADNE_TRY(sc.alloc(&d_gy, (size_t)n_tri));
    }
    // (hipMemcpyDefault: the two big arrays may already live on the device -- padne_generate_grid_mesh, padne_assemble_system_ex)
    PADNE_HIP_CHECK(hipMemcpyAsync(d_xy, xy_host, sizeof(double) * 2 * (size_t)n_vert, hipMemcpyDefault, s));
    PADNE_HIP_CHECK(hipMemcpyAsync(d_tri, tri_host, sizeof(int) * 3 * (size_t)n_tri, hipMemcpyDefault, s));
    if (conductance)
        PADNE_HIP_CHECK(hipMemcpyAsync(d_sigma, conductance, sizeof(double) * (size_t)n_mesh, hipMemcpyHostToDevice, s));
    PADNE_HIP_CHECK(hipMemcpyAsync(d_voff, mesh_vertex_offset, sizeof(long long) * (size_t)(n_mesh + 1), hipMemcpyHostToDevice, s));
    PADNE_HIP_CHECK(hipMemcpyAsync(d_toff, mesh_tri_offset, sizeof(long long) * (size_t)(n_mesh + 1), hipMemcpyHostToDevice, s));
    PADNE_HIP_CHECK(hipMemcpyAsync(d_pot, potential_host, sizeof(double) * (size_t)n_vert, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(power_density_kernel, dim3(nblk(n_tri)), dim3(256), 0, s, (long long)n_tri, d_tri, d_xy, (int)n_mesh,
                       d_voff, d_toff, d_sigma, d_pot, d_out, d_gx, d_gy, d_bad);
    PADNE_HIP_CHECK(hipGetLastError());
    int h_bad = 0;
    PADNE_HIP_CHECK(hipMemcpyAsync(&h_bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, s));
    if (power_out_host)
        PADNE_HIP_CHECK(hipMemcpyAsync(power_out_host, d_out, sizeof(double) * (size_t)n_tri, hipMemcpyDeviceToHost, s));
    if (gx_out_host) {
        PADNE_HIP_CHECK(hipMemcpyAsync(gx_out_host, d_gx, sizeof(double) * (size_t)n_tri, hipMemcpyDeviceToHost, s));
        PADNE_HIP_CHECK(hipMemcpyAsync(gy_out_host, d_gy, sizeof(double) * (size_t)n_tri, hipMemcpyDeviceToHost, s));
    }
    PADNE_HIP_CHECK(hipStreamSynchronize(s));
    if (h_bad) {
        set_error("invalid argument: triangle index out of range");
        return PADNE_E_INVALID;
    }
    return PADNE_OK;
}

extern "C" int padne_power_density(padne_ctx *ctx, int64_t n_vert, const double *xy_host, int64_t n_tri,
                                   const int32_t *tri_host, int64_t n_mesh, const int64_t *mesh_vertex_offset,
                                   const int64_t *mesh_tri_offset, const double *conductance,
                                   const double *potential_host, double *power_out_host) {
    PADNE_REQUIRE(n_tri == 0 || (power_out_host && conductance), "null argument");
    return face_fields(ctx, n_vert, xy_host, n_tri, tri_host, n_mesh, mesh_vertex_offset, mesh_tri_offset,
                       conductance, potential_host, power_out_host, nullptr, nullptr);
}

extern "C" int padne_face_gradient(padne_ctx *ctx, int64_t n_vert, const double *xy_host, int64_t n_tri,
                                   const int32_t *tri_host, int64_t n_mesh, const int64_t *mesh_vertex_offset,
                                   const int64_t *mesh_tri_offset, const double *potential_host,
                                   double *gx_out_host, double *gy_out_host) {
    PADNE_REQUIRE(n_tri == 0 || (gx_out_host && gy_out_host), "null argument");
    return face_fields(ctx, n_vert, xy_host, n_tri, tri_host, n_mesh, mesh_vertex_offset, mesh_tri_offset,
                       nullptr, potential_host, nullptr, gx_out_host, gy_out_host);
}

// ---- connection snapping: nearest mesh vertex of every query point ------------------------------------------
// The reference builds a KD-tree per layer (solver.py:356-396) and queries it once per connection
// (solver.py:425); with a million vertices and a few hundred connections the tree build is two thirds of the host
// time of solve().  Brute force on the device instead: every workgroup takes a chunk of kNnChunk vertices and,
// for every query, the closest vertex of its chunk; a second kernel takes the minimum over the chunks.  Ties go to
// the smallest vertex index (a KD-tree leaves them to its traversal order).
constexpr int kNnChunk = 4096;

__global__ __launch_bounds__(256) void nn_chunk_kernel(long long n, const double *__restrict__ xy, int nq,
                                                       const double *__restrict__ q, double *__restrict__ part_d,
                                                       long long *__restrict__ part_i, int *__restrict__ part_c) {
    // part_c: how many vertices of the chunk sit at exactly the minimum distance (ties: the reference's KD-tree resolves
    // them by traversal order, the caller re-resolves those queries with the tree -- padne_nearest_vertex_ties)
    __shared__ double sd[4];
    __shared__ long long si[4];
    __shared__ int sc[4];
    const long long base = (long long)blockIdx.x * kNnChunk;
    double px[kNnChunk / 256], py[kNnChunk / 256];
#pragma unroll
    for (int k = 0; k < kNnChunk / 256; ++k) {
        const long long i = base + threadIdx.x + 256LL * k;
        px[k] = i < n ? xy[2 * i] : 0.0;
        py[k] = i < n ? xy[2 * i + 1] : 0.0;
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int j = 0; j < nq; ++j) {
        const double qx = q[2 * j], qy = q[2 * j + 1];
        double best = 1e300;
        long long bi = 0x7fffffffffffffffLL;
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < kNnChunk / 256; ++k) {
            const long long i = base + threadIdx.x + 256LL * k;
            const double dx = px[k] - qx, dy = py[k] - qy;
            const double d = dx * dx + dy * dy;
            if (i < n) {
                if (d < best) {
                    best = d;
                    bi = i;
                    cnt = 1;
                } else if (d == best) {
                    cnt += 1;
                    if (i < bi) bi = i;
                }
            }
        }
        for (int off = 32; off > 0; off >>= 1) {
            const double od = __shfl_down(best, off, 64);
            const long long oi = __shfl_down(bi, off, 64);
            const int oc = __shfl_down(cnt, off, 64);
            if (od < best) {
                best = od;
                bi = oi;
                cnt = oc;
            } else if (od == best) {
                cnt += oc;
                if (oi < bi) bi = oi;
            }
        }
        if (lane == 0) {
            sd[w] = best;
            si[w] = bi;
            sc[w] = cnt;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int t = 1; t < 4; ++t) {
                if (sd[t] < best) {
                    best = sd[t];
                    bi = si[t];
                    cnt = sc[t];
                } else if (sd[t] == best) {
                    cnt += sc[t];
                    if (si[t] < bi) bi = si[t];
                }
            }
            part_d[(size_t)blockIdx.x * nq + j] = best;
            part_i[(size_t)blockIdx.x * nq + j] = bi;
            part_c[(size_t)blockIdx.x * nq + j] = cnt;
        }
        __syncthreads();
    }
}

__global__ void nn_reduce_kernel(int n_chunks, int nq, const double *__restrict__ part_d,
                                 const long long *__restrict__ part_i, const int *__restrict__ part_c,
                                 long long *__restrict__ out, int *__restrict__ out_c) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nq) return;
    double best = 1e300;
    long long bi = 0x7fffffffffffffffLL;
    int cnt = 0;
    for (int c = 0; c < n_chunks; ++c) {
        const double d = part_d[(size_t)c * nq + j];
        const long long i = part_i[(size_t)c * nq + j];
        if (d < best) {
            best = d;
            bi = i;
            cnt = part_c[(size_t)c * nq + j];
        } else if (d == best) {
            cnt += part_c[(size_t)c * nq + j];
            if (i < bi) bi = i;
        }
    }
    out[j] = bi;
    out_c[j] = cnt;
}

extern "C" int padne_nearest_vertex(padne_ctx *ctx, int64_t n_points, const double *xy_host, int64_t n_query,
                                    const double *query_host, int64_t *index_out_host) {
    return padne_nearest_vertex_ties(ctx, n_points, xy_host, n_query, query_host, index_out_host, nullptr);
}

extern "C" int padne_nearest_vertex_ties(padne_ctx *ctx, int64_t n_points, const double *xy_host, int64_t n_query,
                                         const double *query_host, int64_t *index_out_host, int32_t *tie_count_out_host) {
    PADNE_REQUIRE(ctx && (n_query == 0 || (query_host && index_out_host)), "null argument");
    PADNE_REQUIRE(n_points >= 1 && xy_host, "at least one point is needed");
    PADNE_REQUIRE(n_query >= 0 && n_query < (1 << 24), "number of queries");
    if (n_query == 0) return PADNE_OK;
    // a NaN / inf coordinate compares false with everything and would leave "no vertex" (INT64_MAX) in the output
    for (int64_t k = 0; k < 2 * n_query; ++k) PADNE_REQUIRE(std::isfinite(query_host[k]), "query coordinates must be finite");
    for (int64_t k = 0; k < 2 * n_points; ++k) PADNE_REQUIRE(std::isfinite(xy_host[k]), "vertex coordinates must be finite");
    PADNE_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    Scratch sc(ctx);
    const int n_chunks = (int)((n_points + kNnChunk - 1) / kNnChunk);
    double *d_xy = nullptr, *d_q = nullptr, *d_pd = nullptr;
    long long *d_pi = nullptr, *d_out = nullptr;
    int *d_pc = nullptr, *d_outc = nullptr;
    PADNE_TRY(sc.alloc(&d_pc, (size_t)n_chunks * n_query));
    PADNE_TRY(sc.alloc(&d_outc, (size_t)n_query));
    PADNE_TRY(sc.alloc(&d_xy, (size_t)2 * n_points));
    PADNE_TRY(sc.alloc(&d_q, (size_t)2 * n_query));
    PADNE_TRY(sc.alloc(&d_pd, (size_t)n_chunks * n_query));
    PADNE_TRY(sc.alloc(&d_pi, (size_t)n_chunks * n_query));
    PADNE_TRY(sc.alloc(&d_out, (size_t)n_query));
    PADNE_HIP_CHECK(hipMemcpyAsync(d_xy, xy_host, sizeof(double) * 2 * (size_t)n_points, hipMemcpyHostToDevice, s));
    PADNE_HIP_CHECK(hipMemcpyAsync(d_q, query_host, sizeof(double) * 2 * (size_t)n_query, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(nn_chunk_kernel, dim3(n_chunks), dim3(256), 0, s, (long long)n_points, d_xy, (int)n_query, d_q, d_pd,
                       d_pi, d_pc);
    hipLaunchKernelGGL(nn_reduce_kernel, dim3(nblk(n_query)), dim3(256), 0, s, n_chunks, (int)n_query, d_pd, d_pi, d_pc, d_out,
                       d_outc);
    PADNE_HIP_CHECK(hipGetLastError());
    PADNE_HIP_CHECK(hipMemcpyAsync(index_out_host, d_out, sizeof(long long) * (size_t)n_query, hipMemcpyDeviceToHost, s));
    if (tie_count_out_host != nullptr)
        PADNE_HIP_CHECK(hipMemcpyAsync(tie_count_out_host, d_outc, sizeof(int) * (size_t)n_query, hipMemcpyDeviceToHost, s));
    PADNE_HIP_CHECK(hipStreamSynchronize(s));
    return PADNE_OK;
}
