// solve_system on the device (solver.py:767-780 with everything around the solve call): the reference hands L and r to
// SuperLU and gets v back; here the KKT system is first reduced to an SPD one, and the bookkeeping of
// that reduction -- index map, right-hand side b = -P^T (r - L c), expansion v = c + P y, residual rows for the multiplier
// recovery, ||L v - r|| -- used to be numpy passes over N-element arrays plus three vector round trips over PCIe
// (0.10 s of a 0.14 s call at N = 10 M).  A padne_kkt plan keeps all of it on the device: the host describes the reduction
// by its O(#constraints) lists, r crosses PCIe once (while the reduced matrix and its multigrid hierarchy are being
// built), v once.
#include "common.hpp"

#include <string.h>
#include <rocprim/device/device_radix_sort.hpp>      // the one stable key-value sort of the locality ordering (section "ordering")

#include <algorithm>
#include <math.h>
#include <string.h>
#include <thread>
#include <vector>

namespace padne {

int csr_relabel(padne_ctx *ctx, const padne_csr *m, const int32_t *row_map, int64_t n_rows_out, const int32_t *col_map,
                int64_t n_cols_out, double scale, padne_csr **out);
int amg_setup(padne_ctx *ctx, padne_csr *A0);
void amg_info(const padne_csr *A0, int *levels, double *complexity, double *setup_seconds, long long *coarse_n);

constexpr int kCopyStreams = 4;

}  // namespace padne

struct padne_kkt {
    padne_ctx *ctx = nullptr;
    const padne_csr *L = nullptr;        // borrowed: the caller keeps the assembled system alive
    long long N = 0, n_pot = 0, n_free = 0;
    int32_t *imap = nullptr;             // [N] reduced unknown of every unknown, -1 = none
    int32_t *src_of = nullptr;           // [n_free] the unknown whose row opens the sum of reduced row t (its representative)
    int32_t *tied_member = nullptr;      // [n_tied] further members of source-tied groups, ascending ...
    int32_t *tied_target = nullptr;      // [n_tied] ... and the reduced unknown they add into
    long long n_tied = 0;
    // the same list grouped by target (= by representative): entries tied_order[tied_gptr[g] .. tied_gptr[g + 1]) of the
    // two arrays above add into one reduced unknown, in ascending member order (kkt_rhs_tied: one thread per group)
    int32_t *tied_order = nullptr, *tied_gptr = nullptr;
    long long n_tied_groups = 0;
    padne_csr *A = nullptr;              // -P^T L P, owned (with its hierarchy once a solve has built it)
    double *r = nullptr, *v = nullptr, *w = nullptr, *c = nullptr;      // [N] device vectors: right-hand side, solution, scratch, known part
    double *b = nullptr, *y = nullptr;   // [(1 + n_extra) * n_free]
    double *Z = nullptr;                 // [n_extra * N] expanded extra solutions (regulators)
    int n_extra_cap = 0, n_extra = 0;
    bool has_c = false, solved = false;
    double setup_seconds_last = 0.0;
};

namespace padne {

// ---- index map from the sparse description -----------------------------------------------------------------------
// imap[i] = i - #{e in elim : e < i} for potentials that are not eliminated, -1 otherwise (elim sorted, in LDS when short)
__global__ __launch_bounds__(256) void kkt_build_imap(const long long N, const long long n_pot, const long long *__restrict__ elim,
                                                      const int n_elim, int32_t *__restrict__ imap) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    if (i >= n_pot) {
        imap[i] = -1;
        return;
    }
    int lo = 0, hi = n_elim;                  // first position with elim[pos] >= i
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (elim[mid] < i) lo = mid + 1; else hi = mid;
    }
    imap[i] = (lo < n_elim && elim[lo] == i) ? -1 : (int32_t)(i - lo);
}

__global__ void kkt_tie_members(const int n_tied, const long long *__restrict__ member, const long long *__restrict__ rep,
                                int32_t *__restrict__ imap, int32_t *__restrict__ tied_member, int32_t *__restrict__ tied_target) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_tied) return;
    const int32_t t = imap[rep[k]];           // representatives are never members themselves: their entry is final
    imap[member[k]] = t;
    tied_member[k] = (int32_t)member[k];
    tied_target[k] = t;
}

// src_of[imap[i]] = i; tied members may race with their representative for a slot -- kkt_fix_sources settles it afterwards
__global__ __launch_bounds__(256) void kkt_sources(const long long N, const int32_t *__restrict__ imap, int32_t *__restrict__ src_of) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int32_t t = imap[i];
    if (t >= 0) src_of[t] = (int32_t)i;
}

__global__ void kkt_fix_sources(const int n_tied, const long long *__restrict__ rep, const int32_t *__restrict__ imap,
                                int32_t *__restrict__ src_of) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_tied) src_of[imap[rep[k]]] = (int32_t)rep[k];
}

__global__ void kkt_scatter_f64(const int n, const long long *__restrict__ idx, const double *__restrict__ val, double *__restrict__ dst) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) dst[idx[k]] = val[k];
}

// ---- right-hand side: b = -P^T (r - L c) ---------------------------------------------------------------------------
// b[t] = -(r - Lc)[src_of[t]]; the other members of a tied group are added by kkt_rhs_tied, one thread, in index order
__global__ __launch_bounds__(256) void kkt_rhs(const long long n_free, const int32_t *__restrict__ src_of, const double *__restrict__ r,
                                               const double *__restrict__ Lc, double *__restrict__ b) {
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n_free; t += (long long)gridDim.x * 256) {
        const int32_t i = src_of[t];
        b[t] = -(Lc != nullptr ? r[i] - Lc[i] : r[i]);
    }
}

// One thread per tied GROUP (all further members of one representative): it subtracts its members' terms from the
// group's reduced row one after the other in ascending member order -- the additions of a single thread walking the whole
// list in index order (what this kernel was: 0.2 s for 1e5 tied members, a chain of dependent read-modify-writes), in
// the same order per row, hence the same bits, in parallel over the rows.
__global__ __launch_bounds__(256) void kkt_rhs_tied(const int n_groups, const int32_t *__restrict__ gptr, const int32_t *__restrict__ order,
                                                    const int32_t *__restrict__ member, const int32_t *__restrict__ target,
                                                    const double *__restrict__ r, const double *__restrict__ Lc, double *__restrict__ b) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    const int e0 = gptr[g], e1 = gptr[g + 1];
    const int32_t t = target[order[e0]];
    double acc = b[t];
    for (int e = e0; e < e1; ++e) {
        const int32_t i = member[order[e]];
        acc -= (Lc != nullptr ? r[i] - Lc[i] : r[i]);
    }
    b[t] = acc;
}

// extra right-hand sides (regulator gain columns): b_k = P^T gamma_k, a handful of entries each, added in list order
__global__ void kkt_rhs_extra(const int n_extra, const long long *__restrict__ ptr, const long long *__restrict__ row,
                              const double *__restrict__ val, const int32_t *__restrict__ imap, const long long n_free,
                              double *__restrict__ b_extra) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    for (int k = 0; k < n_extra; ++k)
        for (long long e = ptr[k]; e < ptr[k + 1]; ++e) {
            const int32_t t = imap[row[e]];
            if (t >= 0) b_extra[(long long)k * n_free + t] += val[e];
        }
}

// per-workgroup partial sums of a.a for up to 8 vectors laid out one after the other (stride n)
__global__ __launch_bounds__(256) void kkt_norm2(const long long n, const double *__restrict__ a, double *__restrict__ partials) {
    __shared__ double red[4];
    const double *v = a + (long long)blockIdx.y * n;
    double s = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += v[i] * v[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[(long long)blockIdx.y * kMaxPartials + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[j] = sum of partials[j][0..P) in a fixed order (one workgroup per vector)
__global__ __launch_bounds__(256) void kkt_fold(const double *__restrict__ partials, const int P, double *__restrict__ out) {
    __shared__ double red[4];
    const double *p = partials + (long long)blockIdx.x * kMaxPartials;
    double s = 0.0;
    for (int i = threadIdx.x; i < P; i += 256) s += p[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- expansion: v = c + P y (multipliers zero) ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void kkt_expand(const long long N, const int32_t *__restrict__ imap, const double *__restrict__ y,
                                                  const double *__restrict__ c, double *__restrict__ v) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long long)gridDim.x * 256) {
        const int32_t t = imap[i];
        const double known = c != nullptr ? c[i] : 0.0;
        v[i] = t >= 0 ? known + y[t] : known;
    }
}

// w = r - w  (w holds L v on entry): the KCL residual rows of the multiplier recovery
__global__ __launch_bounds__(256) void kkt_rho(const long long N, const double *__restrict__ r, double *__restrict__ w) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long long)gridDim.x * 256) w[i] = r[i] - w[i];
}

__global__ void kkt_gather_f64(const int n, const long long *__restrict__ idx, const double *__restrict__ src, double *__restrict__ dst) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) dst[k] = src[idx[k]];
}

// v += sum_k coeff[k] Z_k
__global__ __launch_bounds__(256) void kkt_add_extras(const long long N, const int n_extra, const double *__restrict__ coeff,
                                                      const double *__restrict__ Z, double *__restrict__ v) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long long)gridDim.x * 256) {
        double s = v[i];
        for (int k = 0; k < n_extra; ++k) s += coeff[k] * Z[(long long)k * N + i];
        v[i] = s;
    }
}

// per-workgroup partial sums of (a - b)^2
__global__ __launch_bounds__(256) void kkt_diff2(const long long n, const double *__restrict__ a, const double *__restrict__ b,
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

// ---- locality ordering of the reduced unknowns ("Ordering") ----------------------------------------
// CGAL numbers vertices in insertion order; the reduced system is solved in a band numbering by horizontal strips: the mesh
// unknowns sorted by (mesh, strip of about three vertex spacings, x), the others behind them in their old order.  The host
// form (reduction.apply_locality_ordering: numpy sorts over N-element arrays) was 0.09 s of a 0.14 s solve_system call on a
// 2 M-vertex mesh; here the keys are formed and sorted on the device from the mesh the system was assembled from.  Per-mesh
// constants (bounding box of the owners, strip height) come from a small reduction and the host's own formula, so that
// the keys -- and with a stable sort the permutation -- are the host's bit for bit (tested).
// stats[m] = {x min, x max, y min, y max, count} over the mesh vertices that represent a reduced unknown
__global__ __launch_bounds__(256) void kkt_mesh_stats(const long long *__restrict__ voff, const double *__restrict__ xy,
                                                      const int32_t *__restrict__ imap, const int32_t *__restrict__ src_of,
                                                      double *__restrict__ stats) {
    __shared__ double red[5][4];
    const int m = blockIdx.x;
    double xmin = 1e300, xmax = -1e300, ymin = 1e300, ymax = -1e300, cnt = 0.0;
    for (long long v = voff[m] + threadIdx.x; v < voff[m + 1]; v += 256) {
        const int32_t t = imap[v];
        if (t < 0 || src_of[t] != (int32_t)v) continue;
        const double x = xy[2 * v], y = xy[2 * v + 1];
        xmin = fmin(xmin, x);
        xmax = fmax(xmax, x);
        ymin = fmin(ymin, y);
        ymax = fmax(ymax, y);
        cnt += 1.0;
    }
    for (int off = 32; off > 0; off >>= 1) {
        xmin = fmin(xmin, __shfl_down(xmin, off, 64));
        xmax = fmax(xmax, __shfl_down(xmax, off, 64));
        ymin = fmin(ymin, __shfl_down(ymin, off, 64));
        ymax = fmax(ymax, __shfl_down(ymax, off, 64));
        cnt += __shfl_down(cnt, off, 64);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[0][w] = xmin; red[1][w] = xmax; red[2][w] = ymin; red[3][w] = ymax; red[4][w] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        stats[5 * m + 0] = fmin(fmin(red[0][0], red[0][1]), fmin(red[0][2], red[0][3]));
        stats[5 * m + 1] = fmax(fmax(red[1][0], red[1][1]), fmax(red[1][2], red[1][3]));
        stats[5 * m + 2] = fmin(fmin(red[2][0], red[2][1]), fmin(red[2][2], red[2][3]));
        stats[5 * m + 3] = fmax(fmax(red[3][0], red[3][1]), fmax(red[3][2], red[3][3]));
        stats[5 * m + 4] = (red[4][0] + red[4][1]) + (red[4][2] + red[4][3]);
    }
}

// key[t] = mesh << (32 + strip_bits) | strip << 32 | x quantised to 32 bits inside the mesh; unknowns that are no mesh
// vertex: n_mesh in the mesh field | t.  The same ORDER as the host's keys [mesh 16 | strip 16 | x 32] with 0xFFFF for the
// unknowns behind the vertices -- every field keeps its rank -- in as few bits as the system needs, so that the radix
// sort behind it walks 40-odd bits instead of 64.
// par[m] = {x lo, x span, y0, strip height} as the host computes them (reduction._strip_order / strip_index)
__global__ __launch_bounds__(256) void kkt_strip_keys(const long long n_free, const long long n_vert, const int n_mesh,
                                                      const long long *__restrict__ voff, const double *__restrict__ xy,
                                                      const int32_t *__restrict__ src_of, const double *__restrict__ par,
                                                      unsigned long long *__restrict__ key, int *__restrict__ val, int *__restrict__ bad,
                                                      const int strip_bits) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_free) return;
    val[t] = (int)t;
    const long long v = src_of[t];
    if (v >= n_vert) {
        key[t] = ((unsigned long long)n_mesh << (32 + strip_bits)) | (unsigned long long)t;
        return;
    }
    int lo = 0, hi = n_mesh;                       // mesh of v: largest m with voff[m] <= v
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (voff[mid] <= v) lo = mid; else hi = mid;
    }
    const int m = lo;
    const double x = xy[2 * v], y = xy[2 * v + 1];
    const double x_lo = par[4 * m], span = par[4 * m + 1], y0 = par[4 * m + 2], height = par[4 * m + 3];
    long long strip = 0;
    if (height > 0.0) strip = (long long)floor((y - y0) / height);
    if (strip < 0 || strip >= (1ll << strip_bits)) {        // (the host's bound of the field, computed from the same numbers)
        atomicExch(bad, 1);
        strip = 0;
    }
    double q = (x - x_lo) / span * 4294967295.0;
    if (q > 4294967295.0) q = 4294967295.0;
    const unsigned long long xq = (unsigned long long)q;
    key[t] = ((unsigned long long)m << (32 + strip_bits)) | ((unsigned long long)strip << 32) | xq;
}

__global__ void kkt_invert_perm(const long long n, const int *__restrict__ order, int32_t *__restrict__ new_of_old) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) new_of_old[order[k]] = (int32_t)k;
}

__global__ void kkt_relabel_map(const long long N, const int32_t *__restrict__ new_of_old, int32_t *__restrict__ imap) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int32_t t = imap[i];
    if (t >= 0) imap[i] = new_of_old[t];
}

__global__ void kkt_permute_i32(const long long n, const int *__restrict__ order, const int32_t *__restrict__ src, int32_t *__restrict__ dst) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) dst[k] = src[order[k]];
}

__global__ void kkt_relabel_list(const int n, const int32_t *__restrict__ new_of_old, int32_t *__restrict__ list) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) list[k] = new_of_old[list[k]];
}

// relabels k->imap / src_of / tied_target in place; PADNE_E_INVALID (nothing changed) when the keys do not fit their fields
static int kkt_apply_strip_order(padne_kkt *k) {
    padne_ctx *ctx = k->ctx;
    const padne_csr *L = k->L;
    hipStream_t s = ctx->stream;
    const long long nf = k->n_free, nv = L->mesh_n_vert;
    const int n_mesh = (int)L->mesh_n_mesh;
    if (nf == 0) return PADNE_OK;
    PADNE_REQUIRE(n_mesh > 0 && L->mesh_xy != nullptr && L->mesh_voff != nullptr, "the system matrix carries no mesh to order by");
    PADNE_REQUIRE(n_mesh < 0xFFFF && nf < 2147483647LL && nv <= k->n_pot, "mesh count / size beyond the key fields");
    Scratch sc(ctx);
    double *d_stats = nullptr, *d_par = nullptr;
    unsigned long long *key_a = nullptr, *key_b = nullptr;
    int *val_a = nullptr, *val_b = nullptr, *d_bad = nullptr;
    int32_t *new_of_old = nullptr, *src_new = nullptr;
    PADNE_TRY(sc.alloc(&d_stats, (size_t)5 * n_mesh));
    PADNE_TRY(sc.alloc(&d_par, (size_t)4 * n_mesh));
    PADNE_TRY(sc.alloc(&key_a, (size_t)nf));
    PADNE_TRY(sc.alloc(&key_b, (size_t)nf));
    PADNE_TRY(sc.alloc(&val_a, (size_t)nf));
    PADNE_TRY(sc.alloc(&val_b, (size_t)nf));
    PADNE_TRY(sc.alloc(&d_bad, 1));
    PADNE_TRY(sc.alloc(&new_of_old, (size_t)nf));
    PADNE_TRY(sc.alloc(&src_new, (size_t)nf));
    hipLaunchKernelGGL(kkt_mesh_stats, dim3(n_mesh), dim3(256), 0, s, L->mesh_voff, L->mesh_xy, k->imap, k->src_of, d_stats);
    PADNE_HIP_CHECK(hipGetLastError());
    std::vector<double> st((size_t)5 * n_mesh), par((size_t)4 * n_mesh);
    PADNE_HIP_CHECK(hipMemcpyAsync(st.data(), d_stats, sizeof(double) * st.size(), hipMemcpyDeviceToHost, s));
    PADNE_HIP_CHECK(hipStreamSynchronize(s));
    double strips_max = 1.0;                                // largest strip index any mesh can produce (+ 1 of slack for the rounding of floor)
    for (int m = 0; m < n_mesh; ++m) {
        const double xmin = st[5 * m], xmax = st[5 * m + 1], ymin = st[5 * m + 2], ymax = st[5 * m + 3], cnt = st[5 * m + 4];
        double x_lo = 0.0, span = 1.0, y0 = 0.0, height = 0.0;
        if (cnt >= 1.0) {
            x_lo = xmin;
            span = std::max(xmax - xmin, 1e-300);                               // reduction._strip_order
            y0 = ymin;
            if (cnt >= 2.0) {                                                   // reduction.strip_index (a single point: strip 0)
                const double area = std::max((xmax - xmin) * (ymax - y0), 1e-300);
                height = 3.4 * sqrt(area / cnt);
            }
        }
        if (height > 0.0) strips_max = std::max(strips_max, floor((ymax - y0) / height) + 2.0);
        par[4 * m] = x_lo;
        par[4 * m + 1] = span;
        par[4 * m + 2] = y0;
        par[4 * m + 3] = height;
    }
    PADNE_HIP_CHECK(hipMemcpyAsync(d_par, par.data(), sizeof(double) * par.size(), hipMemcpyHostToDevice, s));
    PADNE_HIP_CHECK(hipMemsetAsync(d_bad, 0, sizeof(int), s));
    // the fields of the sort key, as narrow as this system allows: strips (at most 16 bits, the host's field), meshes + 1
    int strip_bits = 1, mesh_bits = 1;
    while (strip_bits < 16 && (double)(1ll << strip_bits) <= strips_max) ++strip_bits;
    while ((1ll << mesh_bits) <= (long long)n_mesh) ++mesh_bits;
    const int key_bits = 32 + strip_bits + mesh_bits;      // <= 32 + 16 + 16
    hipLaunchKernelGGL(kkt_strip_keys, dim3(nblk(nf)), dim3(256), 0, s, nf, nv, n_mesh, L->mesh_voff, L->mesh_xy, k->src_of, d_par,
                       key_a, val_a, d_bad, strip_bits);
    PADNE_HIP_CHECK(hipGetLastError());
    int h_bad = 0;
    PADNE_HIP_CHECK(hipMemcpyAsync(&h_bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, s));
    PADNE_HIP_CHECK(hipStreamSynchronize(s));              // (also: par's host buffer may go)
    if (h_bad) {
        set_error("strip index beyond 16 bits: the host orders this system");
        return PADNE_E_INVALID;
    }
    // stable sort of (key, old index): equal keys keep their index order, as the host's tie repair leaves them
    size_t tmp_bytes = 0;
    PADNE_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, key_a, key_b, val_a, val_b, (size_t)nf, 0, key_bits, s));
    void *tmp = nullptr;
    PADNE_TRY(sc.alloc((char **)&tmp, tmp_bytes));
    PADNE_HIP_CHECK(rocprim::radix_sort_pairs(tmp, tmp_bytes, key_a, key_b, val_a, val_b, (size_t)nf, 0, key_bits, s));
    const int *order = val_b;                              // new position -> old reduced index
    hipLaunchKernelGGL(kkt_invert_perm, dim3(nblk(nf)), dim3(256), 0, s, nf, order, new_of_old);
    hipLaunchKernelGGL(kkt_relabel_map, dim3(nblk(k->N)), dim3(256), 0, s, k->N, (const int32_t *)new_of_old, k->imap);
    hipLaunchKernelGGL(kkt_permute_i32, dim3(nblk(nf)), dim3(256), 0, s, nf, order, (const int32_t *)k->src_of, src_new);
    if (k->n_tied > 0)
        hipLaunchKernelGGL(kkt_relabel_list, dim3(nblk(k->n_tied)), dim3(256), 0, s, (int)k->n_tied, (const int32_t *)new_of_old,
                           k->tied_target);
    PADNE_HIP_CHECK(hipGetLastError());
    PADNE_HIP_CHECK(hipMemcpyAsync(k->src_of, src_new, sizeof(int32_t) * (size_t)nf, hipMemcpyDeviceToDevice, s));
    return PADNE_OK;
}

static int vgrid(long long n) {
    long long g = (n + 255) / 256;
    if (g > 1024) g = 1024;
    return (int)(g < 1 ? 1 : g);
}

// A pageable host buffer crosses PCIe through the runtime's pinned staging area at the speed of ONE host core's memcpy
// (8-10 GB/s: 8-10 ms for the 80 MB of a 10 M-unknown vector); kCopyStreams threads, each with a stream and a quarter
// of the vector, bring it close to the link.  Blocks the caller until all parts have arrived.
static int copy_streams(padne_ctx *ctx, int count) {      // the context's first `count` copy streams exist
    static_assert(kCopyStreams == sizeof(ctx->copy_stream) / sizeof(ctx->copy_stream[0]), "the context holds the copy streams");
    for (int t = 0; t < count; ++t)
        if (ctx->copy_stream[t] == nullptr && hipStreamCreateWithFlags(&ctx->copy_stream[t], hipStreamNonBlocking) != hipSuccess) {
            ctx->copy_stream[t] = nullptr;
            set_error("stream creation failed");
            return PADNE_E_HIP;
        }
    return PADNE_OK;
}

static int parallel_copy(padne_kkt *k, void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
    const int device = k->ctx->device;
    hipStream_t *copy_stream = k->ctx->copy_stream;
    if (bytes < ((size_t)8 << 20)) {
        PADNE_TRY(copy_streams(k->ctx, 1));
        PADNE_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, kind, copy_stream[0]));
        PADNE_HIP_CHECK(hipStreamSynchronize(copy_stream[0]));
        return PADNE_OK;
    }
    PADNE_TRY(copy_streams(k->ctx, kCopyStreams));
    hipError_t err[kCopyStreams];
    std::thread th[kCopyStreams];
    const size_t chunk = ((bytes / kCopyStreams) + 4095) & ~(size_t)4095;
    for (int t = 0; t < kCopyStreams; ++t) {
        const size_t off = std::min(bytes, (size_t)t * chunk), len = std::min(bytes - off, chunk);
        err[t] = hipSuccess;
        th[t] = std::thread([=, &err]() {
            if (len == 0) return;
            hipError_t e = hipSetDevice(device);
            if (e == hipSuccess) e = hipMemcpyAsync((char *)dst + off, (const char *)src + off, len, kind, copy_stream[t]);
            if (e == hipSuccess) e = hipStreamSynchronize(copy_stream[t]);
            err[t] = e;
        });
    }
    for (int t = 0; t < kCopyStreams; ++t) th[t].join();
    for (int t = 0; t < kCopyStreams; ++t)
        if (err[t] != hipSuccess) {
            set_error("vector transfer failed: %s", hipGetErrorString(err[t]));
            return PADNE_E_HIP;
        }
    return PADNE_OK;
}

static void kkt_free(padne_kkt *k) {
    if (k == nullptr) return;
    padne_ctx *ctx = k->ctx;
    if (ctx != nullptr && ctx->stream != nullptr) (void)hipStreamSynchronize(ctx->stream);
    if (ctx != nullptr)
        for (hipStream_t cs : ctx->copy_stream)
            if (cs != nullptr) (void)hipStreamSynchronize(cs);
    if (k->A != nullptr) padne_csr_destroy(k->A);
    for (void *p : {(void *)k->imap, (void *)k->src_of, (void *)k->tied_member, (void *)k->tied_target, (void *)k->tied_order,
                    (void *)k->tied_gptr, (void *)k->r, (void *)k->v,
                    (void *)k->w, (void *)k->c, (void *)k->b, (void *)k->y, (void *)k->Z})
        if (p != nullptr) pool_free(ctx, p);
    delete k;
}

}  // namespace padne

using namespace padne;

extern "C" int padne_kkt_create(padne_ctx *ctx, const padne_csr *L, int64_t n_potential, int64_t n_elim,
                                const int64_t *elim_sorted, int64_t n_tied, const int64_t *tied_member,
                                const int64_t *tied_rep, const int32_t *index_map_host, int64_t n_free, int32_t flags,
                                padne_kkt **out) {
    PADNE_REQUIRE(ctx && L && out, "null argument");
    PADNE_REQUIRE(L->n_rows == L->n_cols, "the system matrix must be square");
    const long long N = L->n_rows;
    PADNE_REQUIRE(n_potential >= 0 && n_potential <= N, "n_potential");
    PADNE_REQUIRE(n_elim >= 0 && n_tied >= 0 && n_tied <= n_elim && n_elim < 2147483647LL, "list sizes");
    PADNE_REQUIRE(n_elim == 0 || elim_sorted != nullptr, "elim list");
    PADNE_REQUIRE(n_tied == 0 || (tied_member != nullptr && tied_rep != nullptr), "tied lists");
    PADNE_REQUIRE(n_free >= 0 && n_free <= n_potential, "n_free");
    PADNE_REQUIRE(index_map_host != nullptr || n_free == n_potential - n_elim, "n_free does not match the lists");
    for (int64_t k = 0; k < n_elim; ++k)
        PADNE_REQUIRE(elim_sorted[k] >= 0 && elim_sorted[k] < n_potential && (k == 0 || elim_sorted[k - 1] < elim_sorted[k]),
                      "elim list must be sorted, unique and inside the potentials");
    for (int64_t k = 0; k < n_tied; ++k) {
        PADNE_REQUIRE(tied_member[k] >= 0 && tied_member[k] < n_potential && tied_rep[k] >= 0 && tied_rep[k] < n_potential &&
                      tied_rep[k] != tied_member[k] && (k == 0 || tied_member[k - 1] < tied_member[k]),
                      "tied lists must be sorted by member and inside the potentials");
        PADNE_REQUIRE(std::binary_search(elim_sorted, elim_sorted + n_elim, tied_member[k]) &&
                      !std::binary_search(elim_sorted, elim_sorted + n_elim, tied_rep[k]),
                      "a tied member must be eliminated and its representative must not be");
    }
    PADNE_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    padne_kkt *k = new padne_kkt();
    k->ctx = ctx;
    k->L = L;
    k->N = N;
    k->n_pot = n_potential;
    k->n_free = n_free;
    k->n_tied = n_tied;
    int rc = PADNE_OK;
    auto fail = [&](int code) {
        kkt_free(k);
        return code;
    };
    // (the copy streams of the context, made here on the calling thread: the copies themselves run on threads of their own)
    if (copy_streams(ctx, sizeof(double) * (size_t)(N > 0 ? N : 1) >= ((size_t)8 << 20) ? kCopyStreams : 1) != PADNE_OK) return fail(PADNE_E_HIP);
    const size_t nN = (size_t)(N > 0 ? N : 1), nF = (size_t)(n_free > 0 ? n_free : 1), nT = (size_t)(n_tied > 0 ? n_tied : 1);
    k->imap = (int32_t *)pool_alloc(ctx, sizeof(int32_t) * nN);
    k->src_of = (int32_t *)pool_alloc(ctx, sizeof(int32_t) * nF);
    k->tied_member = (int32_t *)pool_alloc(ctx, sizeof(int32_t) * nT);
    k->tied_target = (int32_t *)pool_alloc(ctx, sizeof(int32_t) * nT);
    k->r = (double *)pool_alloc(ctx, sizeof(double) * nN);
    k->v = (double *)pool_alloc(ctx, sizeof(double) * nN);
    k->w = (double *)pool_alloc(ctx, sizeof(double) * nN);
    k->c = (double *)pool_alloc(ctx, sizeof(double) * nN);
    k->b = (double *)pool_alloc(ctx, sizeof(double) * nF);
    k->y = (double *)pool_alloc(ctx, sizeof(double) * nF);
    if (!k->imap || !k->src_of || !k->tied_member || !k->tied_target || !k->r || !k->v || !k->w || !k->c || !k->b || !k->y)
        return fail(PADNE_E_NOMEM);
    Scratch sc(ctx);
    long long *d_elim = nullptr, *d_mem = nullptr, *d_rep = nullptr;
    if ((rc = sc.alloc(&d_elim, (size_t)n_elim)) != PADNE_OK || (rc = sc.alloc(&d_mem, (size_t)n_tied)) != PADNE_OK ||
        (rc = sc.alloc(&d_rep, (size_t)n_tied)) != PADNE_OK)
        return fail(rc);
    hipError_t e = hipSuccess;
    std::vector<int32_t> h_order, h_gptr;                  // (outlive the asynchronous copies below: synchronised before return)
    if (n_tied > 0) {
        e = hipMemcpyAsync(d_mem, tied_member, sizeof(long long) * (size_t)n_tied, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(d_rep, tied_rep, sizeof(long long) * (size_t)n_tied, hipMemcpyHostToDevice, s);
        // the list grouped by representative; inside a group the members keep their (ascending) order
        h_order.resize((size_t)n_tied);
        for (int64_t q = 0; q < n_tied; ++q) h_order[(size_t)q] = (int32_t)q;
        std::stable_sort(h_order.begin(), h_order.end(), [&](int32_t a, int32_t b) { return tied_rep[a] < tied_rep[b]; });
        for (int64_t q = 0; q < n_tied; ++q)
            if (q == 0 || tied_rep[h_order[(size_t)q]] != tied_rep[h_order[(size_t)q - 1]]) h_gptr.push_back((int32_t)q);
        k->n_tied_groups = (long long)h_gptr.size();
        h_gptr.push_back((int32_t)n_tied);
        k->tied_order = (int32_t *)pool_alloc(ctx, sizeof(int32_t) * h_order.size());
        k->tied_gptr = (int32_t *)pool_alloc(ctx, sizeof(int32_t) * h_gptr.size());
        if (!k->tied_order || !k->tied_gptr) return fail(PADNE_E_NOMEM);
        if (e == hipSuccess) e = hipMemcpyAsync(k->tied_order, h_order.data(), sizeof(int32_t) * h_order.size(), hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(k->tied_gptr, h_gptr.data(), sizeof(int32_t) * h_gptr.size(), hipMemcpyHostToDevice, s);
    }
    if (e == hipSuccess && index_map_host != nullptr) {
        // a map the host made (locality reordering of a scattered numbering): used as it is
        e = hipMemcpyAsync(k->imap, index_map_host, sizeof(int32_t) * (size_t)N, hipMemcpyHostToDevice, s);
        if (e == hipSuccess && n_tied > 0) {
            // tied_target straight from the uploaded map
            hipLaunchKernelGGL(kkt_tie_members, dim3(nblk(n_tied)), dim3(256), 0, s, (int)n_tied, d_mem, d_mem, k->imap,
                               k->tied_member, k->tied_target);
            e = hipGetLastError();
        }
    } else if (e == hipSuccess) {
        if (n_elim > 0) e = hipMemcpyAsync(d_elim, elim_sorted, sizeof(long long) * (size_t)n_elim, hipMemcpyHostToDevice, s);
        if (e == hipSuccess && N > 0) {
            hipLaunchKernelGGL(kkt_build_imap, dim3(nblk(N)), dim3(256), 0, s, N, (long long)n_potential, d_elim, (int)n_elim, k->imap);
            if (n_tied > 0)
                hipLaunchKernelGGL(kkt_tie_members, dim3(nblk(n_tied)), dim3(256), 0, s, (int)n_tied, d_mem, d_rep, k->imap,
                                   k->tied_member, k->tied_target);
            e = hipGetLastError();
        }
    }
    if (e == hipSuccess && N > 0) {
        hipLaunchKernelGGL(kkt_sources, dim3(nblk(N)), dim3(256), 0, s, N, k->imap, k->src_of);
        if (n_tied > 0)
            hipLaunchKernelGGL(kkt_fix_sources, dim3(nblk(n_tied)), dim3(256), 0, s, (int)n_tied, d_rep, k->imap, k->src_of);
        e = hipGetLastError();
    }
    if (e != hipSuccess) {
        set_error("building the index map failed: %s", hipGetErrorString(e));
        return fail(PADNE_E_HIP);
    }
    if ((flags & 1) != 0 && index_map_host == nullptr) {
        // the reduced unknowns in the strip numbering (mesh, strip, x) of the mesh the system was assembled from
        if ((rc = kkt_apply_strip_order(k)) != PADNE_OK) return fail(rc);
    }
    // A = -P^T L P from the device-resident map (csr_relabel takes host or device maps)
    if ((rc = csr_relabel(ctx, L, k->imap, n_free, k->imap, n_free, -1.0, &k->A)) != PADNE_OK) return fail(rc);
    if (hipStreamSynchronize(s) != hipSuccess) {      // the scratch lists go out of scope
        set_error("plan creation failed: %s", hipGetErrorString(hipGetLastError()));
        return fail(PADNE_E_HIP);
    }
    *out = k;
    return PADNE_OK;
}

extern "C" int padne_kkt_destroy(padne_kkt *k) {
    if (k != nullptr && k->ctx != nullptr) (void)hipSetDevice(k->ctx->device);
    kkt_free(k);
    return PADNE_OK;
}

extern "C" int padne_kkt_matrix(const padne_kkt *k, const padne_csr **reduced_out) {
    PADNE_REQUIRE(k && reduced_out, "null argument");
    *reduced_out = k->A;
    return PADNE_OK;
}

extern "C" int padne_kkt_solve(padne_ctx *ctx, padne_kkt *k, const double *r_host, int64_t n_known, const int64_t *known_idx,
                               const double *known_val, int32_t n_extra, const int64_t *extra_ptr, const int64_t *extra_row,
                               const double *extra_val, int64_t n_probe, const int64_t *probe_idx, double *probe_out,
                               const padne_solve_opts *opts, double abs_residual_target, padne_solve_info *info) {
    PADNE_REQUIRE(ctx && k && r_host && opts, "null argument");
    PADNE_REQUIRE(k->ctx == ctx, "the plan belongs to another context");
    PADNE_REQUIRE(n_known >= 0 && (n_known == 0 || (known_idx && known_val)), "known potentials");
    PADNE_REQUIRE(n_extra >= 0 && n_extra <= 4096 && (n_extra == 0 || (extra_ptr && extra_ptr[0] == 0)), "extra right-hand sides");
    PADNE_REQUIRE(n_probe >= 0 && (n_probe == 0 || (probe_idx && probe_out)), "probes");
    const long long N = k->N, nf = k->n_free;
    for (int64_t j = 0; j < n_known; ++j) PADNE_REQUIRE(known_idx[j] >= 0 && known_idx[j] < k->n_pot, "known potential out of range");
    for (int64_t j = 0; j < n_probe; ++j) PADNE_REQUIRE(probe_idx[j] >= 0 && probe_idx[j] < N, "probe out of range");
    const long long n_ex_entries = n_extra > 0 ? extra_ptr[n_extra] : 0;
    for (int j = 0; j < n_extra; ++j) PADNE_REQUIRE(extra_ptr[j] <= extra_ptr[j + 1], "extra_ptr must be monotone");
    PADNE_REQUIRE(n_ex_entries == 0 || (extra_row && extra_val), "extra entries");
    for (long long e = 0; e < n_ex_entries; ++e) PADNE_REQUIRE(extra_row[e] >= 0 && extra_row[e] < N, "extra row out of range");
    PADNE_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    padne_solve_info local;
    memset(&local, 0, sizeof(local));
    local.n_rhs = 1 + n_extra;
    k->solved = false;
    k->n_extra = n_extra;
    // more right-hand sides than last time: grow b, y, Z
    if (n_extra > k->n_extra_cap) {
        PADNE_HIP_CHECK(hipStreamSynchronize(s));
        pool_free(ctx, k->b);
        pool_free(ctx, k->y);
        pool_free(ctx, k->Z);
        k->b = k->y = k->Z = nullptr;
        const size_t nF = (size_t)(nf > 0 ? nf : 1) * (size_t)(1 + n_extra);
        k->b = (double *)pool_alloc(ctx, sizeof(double) * nF);
        k->y = (double *)pool_alloc(ctx, sizeof(double) * nF);
        k->Z = (double *)pool_alloc(ctx, sizeof(double) * (size_t)(N > 0 ? N : 1) * (size_t)n_extra);
        if (!k->b || !k->y || !k->Z) return PADNE_E_NOMEM;
        k->n_extra_cap = n_extra;
    }
    // 1. r crosses PCIe on its own streams while this thread builds what does not depend on it: 1/diag, the x-window
    //    plan and the multigrid hierarchy of A (the counterpart of the factorisation)
    int up_rc = PADNE_OK;
    std::thread uploader([&]() {
        (void)hipSetDevice(ctx->device);
        up_rc = parallel_copy(k, k->r, r_host, sizeof(double) * (size_t)N, hipMemcpyHostToDevice);
    });
    struct Join {
        std::thread &t;
        ~Join() { if (t.joinable()) t.join(); }
    } join_guard{uploader};
    if ((opts->flags & 4) != 0 && k->A->amg != nullptr) {
        amg_destroy(k->A->amg);
        k->A->amg = nullptr;
    }
    const bool want_amg = opts->precond == 1 && nf > kTinySystem;
    double setup_s = 0.0;
    if (nf > 0) {
        PADNE_TRY(csr_build_dinv(ctx, k->A));
        PADNE_TRY(csr_build_xw_plan(ctx, k->A));
        if (want_amg && k->A->amg == nullptr) {
            const int rc_setup = amg_setup(ctx, k->A);
            if (rc_setup != PADNE_OK && rc_setup != PADNE_E_NOCOARSEN) return rc_setup;
            if (rc_setup == PADNE_OK) amg_info(k->A, nullptr, nullptr, &setup_s, nullptr);
        }
    }
    k->setup_seconds_last = setup_s;
    // known part of the potentials: c (zero unless sources fix potentials against the ground or against each other)
    Scratch sc(ctx);
    k->has_c = n_known > 0;
    if (k->has_c) {
        long long *d_idx = nullptr;
        double *d_val = nullptr;
        PADNE_TRY(sc.alloc(&d_idx, (size_t)n_known));
        PADNE_TRY(sc.alloc(&d_val, (size_t)n_known));
        PADNE_HIP_CHECK(hipMemsetAsync(k->c, 0, sizeof(double) * (size_t)N, s));
        PADNE_HIP_CHECK(hipMemcpyAsync(d_idx, known_idx, sizeof(long long) * (size_t)n_known, hipMemcpyHostToDevice, s));
        PADNE_HIP_CHECK(hipMemcpyAsync(d_val, known_val, sizeof(double) * (size_t)n_known, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(kkt_scatter_f64, dim3(nblk(n_known)), dim3(256), 0, s, (int)n_known, d_idx, d_val, k->c);
        PADNE_HIP_CHECK(hipGetLastError());
        PADNE_TRY(csr_build_xw_plan(ctx, const_cast<padne_csr *>(k->L)));
        PADNE_TRY(launch_spmv(ctx, k->L, k->c, k->w, nullptr, nullptr, nullptr));       // w = L c
    }
    uploader.join();
    PADNE_TRY(up_rc);
    // 2. b = -P^T (r - L c), the extra right-hand sides, their norms
    const double *Lc = k->has_c ? k->w : nullptr;
    std::vector<double> norm2_buf((size_t)n_extra + 8, 0.0);      // (one per right-hand side: any number of regulators)
    double *h_norm2 = norm2_buf.data();
    if (nf > 0) {
        hipLaunchKernelGGL(kkt_rhs, dim3(vgrid(nf)), dim3(256), 0, s, nf, k->src_of, k->r, Lc, k->b);
        if (k->n_tied > 0)
            hipLaunchKernelGGL(kkt_rhs_tied, dim3(nblk(k->n_tied_groups)), dim3(256), 0, s, (int)k->n_tied_groups, k->tied_gptr,
                               k->tied_order, k->tied_member, k->tied_target, k->r, Lc, k->b);
        PADNE_HIP_CHECK(hipGetLastError());
        if (n_extra > 0) {
            long long *d_ptr = nullptr, *d_row = nullptr;
            double *d_val = nullptr;
            PADNE_TRY(sc.alloc(&d_ptr, (size_t)n_extra + 1));
            PADNE_TRY(sc.alloc(&d_row, (size_t)n_ex_entries));
            PADNE_TRY(sc.alloc(&d_val, (size_t)n_ex_entries));
            PADNE_HIP_CHECK(hipMemsetAsync(k->b + nf, 0, sizeof(double) * (size_t)nf * (size_t)n_extra, s));
            PADNE_HIP_CHECK(hipMemcpyAsync(d_ptr, extra_ptr, sizeof(long long) * (size_t)(n_extra + 1), hipMemcpyHostToDevice, s));
            if (n_ex_entries > 0) {
                PADNE_HIP_CHECK(hipMemcpyAsync(d_row, extra_row, sizeof(long long) * (size_t)n_ex_entries, hipMemcpyHostToDevice, s));
                PADNE_HIP_CHECK(hipMemcpyAsync(d_val, extra_val, sizeof(double) * (size_t)n_ex_entries, hipMemcpyHostToDevice, s));
            }
            hipLaunchKernelGGL(kkt_rhs_extra, dim3(1), dim3(1), 0, s, (int)n_extra, d_ptr, d_row, d_val, k->imap, nf, k->b + nf);
            PADNE_HIP_CHECK(hipGetLastError());
        }
        // norms of all right-hand sides (the tolerance rule below; zero right-hand sides are not solved for)
        const int g = vgrid(nf);
        for (int first = 0; first < 1 + n_extra; first += 8) {
            const int cnt = std::min(8, 1 + n_extra - first);
            hipLaunchKernelGGL(kkt_norm2, dim3(g, cnt), dim3(256), 0, s, nf, k->b + (long long)first * nf, ctx->partials);
            hipLaunchKernelGGL(kkt_fold, dim3(cnt), dim3(256), 0, s, ctx->partials, g, ctx->scalars + 32);
            PADNE_HIP_CHECK(hipGetLastError());
            if (cnt <= 7) {
                PADNE_TRY(read_back(ctx, ctx->scalars + 32, sizeof(double) * (size_t)cnt, h_norm2 + first));
            } else {
                PADNE_TRY(read_back2(ctx, ctx->scalars + 32, sizeof(double) * 4, h_norm2 + first, ctx->scalars + 36,
                                     sizeof(double) * 3, h_norm2 + first + 4));
                PADNE_TRY(read_back(ctx, ctx->scalars + 39, sizeof(double), h_norm2 + first + 7));
            }
        }
    }
    // 3. the reference judges a solve by the ABSOLUTE residual of the whole system (tests/test_solver.py:2083-2089): when
    //    rtol ||b|| is looser than the target the relative tolerance is tightened (never below what binary64 resolves)
    double norm_max = 0.0;
    for (int j = 0; j < 1 + n_extra; ++j) norm_max = std::max(norm_max, sqrt(h_norm2[j]));
    padne_solve_opts o = *opts;
    o.flags &= ~(1 | 4);                     // x0 = 0; the hierarchy was (re)built above
    if (abs_residual_target > 0.0 && norm_max > 0.0 && o.rtol * norm_max > abs_residual_target)
        o.rtol = std::max(abs_residual_target / norm_max, 2e-15);
    // right-hand sides that vanish are not solved for; the live ones are packed to the front (they already are unless a
    // regulator's gain column projects to zero)
    int rc_solve = PADNE_OK;
    if (nf > 0) {
        PADNE_HIP_CHECK(hipMemsetAsync(k->y, 0, sizeof(double) * (size_t)nf * (size_t)(1 + n_extra), s));
        int j = 0;
        while (j < 1 + n_extra) {
            if (!(h_norm2[j] > 0.0)) {
                ++j;
                continue;
            }
            int j1 = j;
            while (j1 < 1 + n_extra && h_norm2[j1] > 0.0) ++j1;        // a run of live right-hand sides: one call
            padne_solve_info part;
            memset(&part, 0, sizeof(part));
            const int rc = padne_solve_spd_dev(ctx, k->A, k->b + (long long)j * nf, k->y + (long long)j * nf, j1 - j, &o, &part);
            if (rc != PADNE_OK && rc != PADNE_E_NOTCONVERGED) return rc;
            if (rc != PADNE_OK) rc_solve = rc;
            local.iterations += part.iterations;
            local.restarts += part.restarts;
            local.rel_residual = std::max(local.rel_residual, part.rel_residual);
            local.abs_residual = std::max(local.abs_residual, part.abs_residual);
            local.solve_seconds += part.solve_seconds;
            local.spmv_seconds = std::max(local.spmv_seconds, part.spmv_seconds);
            local.precond_fallbacks += part.precond_fallbacks;
            local.levels = part.levels;
            local.operator_complexity = part.operator_complexity;
            if (part.status != PADNE_OK) local.status = part.status;
            j = j1;
        }
    }
    local.precond_setup_seconds = setup_s;
    // 4. v = c + P y (multipliers still zero), Z_k = P z_k, and the KCL residual rows the host peels the multipliers from
    const double *c = k->has_c ? k->c : nullptr;
    hipLaunchKernelGGL(kkt_expand, dim3(vgrid(N)), dim3(256), 0, s, N, k->imap, k->y, c, k->v);
    for (int j = 0; j < n_extra; ++j)
        hipLaunchKernelGGL(kkt_expand, dim3(vgrid(N)), dim3(256), 0, s, N, k->imap, k->y + (long long)(1 + j) * nf,
                           (const double *)nullptr, k->Z + (long long)j * N);
    PADNE_HIP_CHECK(hipGetLastError());
    if (n_probe > 0) {
        long long *d_idx = nullptr;
        double *d_out = nullptr;
        PADNE_TRY(sc.alloc(&d_idx, (size_t)n_probe));
        PADNE_TRY(sc.alloc(&d_out, (size_t)n_probe * (size_t)(1 + n_extra)));
        PADNE_HIP_CHECK(hipMemcpyAsync(d_idx, probe_idx, sizeof(long long) * (size_t)n_probe, hipMemcpyHostToDevice, s));
        PADNE_TRY(csr_build_xw_plan(ctx, const_cast<padne_csr *>(k->L)));
        PADNE_TRY(launch_spmv(ctx, k->L, k->v, k->w, nullptr, nullptr, nullptr));
        hipLaunchKernelGGL(kkt_rho, dim3(vgrid(N)), dim3(256), 0, s, N, k->r, k->w);                    // rho = r - L v
        hipLaunchKernelGGL(kkt_gather_f64, dim3(nblk(n_probe)), dim3(256), 0, s, (int)n_probe, d_idx, k->w, d_out);
        for (int j = 0; j < n_extra; ++j) {
            PADNE_TRY(launch_spmv(ctx, k->L, k->Z + (long long)j * N, k->w, nullptr, nullptr, nullptr));  // L Z_k
            hipLaunchKernelGGL(kkt_gather_f64, dim3(nblk(n_probe)), dim3(256), 0, s, (int)n_probe, d_idx, k->w,
                               d_out + (size_t)(1 + j) * (size_t)n_probe);
        }
        PADNE_HIP_CHECK(hipGetLastError());
        PADNE_HIP_CHECK(hipMemcpyAsync(probe_out, d_out, sizeof(double) * (size_t)n_probe * (size_t)(1 + n_extra),
                                       hipMemcpyDeviceToHost, s));
    }
    PADNE_HIP_CHECK(hipStreamSynchronize(s));
    k->solved = true;
    if (info) *info = local;
    return rc_solve;
}

extern "C" int padne_kkt_finish(padne_ctx *ctx, padne_kkt *k, int32_t n_extra, const double *extra_coeff, int64_t n_mult,
                                const int64_t *mult_idx, const double *mult_val, double *v_host, double *residual_norm_out) {
    PADNE_REQUIRE(ctx && k && v_host && residual_norm_out, "null argument");
    PADNE_REQUIRE(k->ctx == ctx && k->solved, "padne_kkt_finish follows padne_kkt_solve on the same plan");
    PADNE_REQUIRE(n_extra == k->n_extra && (n_extra == 0 || extra_coeff), "one coefficient per extra right-hand side");
    PADNE_REQUIRE(n_mult >= 0 && (n_mult == 0 || (mult_idx && mult_val)), "multipliers");
    const long long N = k->N;
    for (int64_t j = 0; j < n_mult; ++j) PADNE_REQUIRE(mult_idx[j] >= 0 && mult_idx[j] < N, "multiplier index out of range");
    PADNE_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    Scratch sc(ctx);
    if (n_extra > 0) {
        double *d_coeff = nullptr;
        PADNE_TRY(sc.alloc(&d_coeff, (size_t)n_extra));
        PADNE_HIP_CHECK(hipMemcpyAsync(d_coeff, extra_coeff, sizeof(double) * (size_t)n_extra, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(kkt_add_extras, dim3(vgrid(N)), dim3(256), 0, s, N, (int)n_extra, d_coeff, k->Z, k->v);
        PADNE_HIP_CHECK(hipGetLastError());
    }
    if (n_mult > 0) {
        long long *d_idx = nullptr;
        double *d_val = nullptr;
        PADNE_TRY(sc.alloc(&d_idx, (size_t)n_mult));
        PADNE_TRY(sc.alloc(&d_val, (size_t)n_mult));
        PADNE_HIP_CHECK(hipMemcpyAsync(d_idx, mult_idx, sizeof(long long) * (size_t)n_mult, hipMemcpyHostToDevice, s));
        PADNE_HIP_CHECK(hipMemcpyAsync(d_val, mult_val, sizeof(double) * (size_t)n_mult, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(kkt_scatter_f64, dim3(nblk(n_mult)), dim3(256), 0, s, (int)n_mult, d_idx, d_val, k->v);
        PADNE_HIP_CHECK(hipGetLastError());
    }
    // v is final: it travels home on the copy streams while the main stream evaluates ||L v - r|| (solver.py:775)
    PADNE_HIP_CHECK(hipStreamSynchronize(s));
    int down_rc = PADNE_OK;
    std::thread downloader([&]() {
        (void)hipSetDevice(ctx->device);
        down_rc = parallel_copy(k, v_host, k->v, sizeof(double) * (size_t)N, hipMemcpyDeviceToHost);
    });
    struct Join {
        std::thread &t;
        ~Join() { if (t.joinable()) t.join(); }
    } join_guard{downloader};
    double norm2 = 0.0;
    if (N > 0) {
        PADNE_TRY(csr_build_xw_plan(ctx, const_cast<padne_csr *>(k->L)));
        PADNE_TRY(launch_spmv(ctx, k->L, k->v, k->w, nullptr, nullptr, nullptr));
        const int g = vgrid(N);
        hipLaunchKernelGGL(kkt_diff2, dim3(g), dim3(256), 0, s, N, k->w, k->r, ctx->partials);
        hipLaunchKernelGGL(kkt_fold, dim3(1), dim3(256), 0, s, ctx->partials, g, ctx->scalars + 32);
        PADNE_HIP_CHECK(hipGetLastError());
        PADNE_TRY(read_back(ctx, ctx->scalars + 32, sizeof(double), &norm2));
    }
    downloader.join();
    PADNE_TRY(down_rc);
    *residual_norm_out = sqrt(norm2);
    k->solved = false;
    return PADNE_OK;
}
