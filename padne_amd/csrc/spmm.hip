// CSR SpMM for gfx950: Y = A X for kSpmmK = 8 right-hand sides at once (config C5: the same matrix, 8 current
// source configurations; SURVEY.md section 8d: 12*nnz + 4*N + 16*N*k algorithmic bytes instead of k SpMVs).
//
// Layout: the k vectors are interleaved, X[i*8 + j] is entry i of right-hand side j, so the gather for one
// non-zero is ONE contiguous 64-byte line instead of eight scattered doubles.
//
// Kernel: the SpMV skeleton (spmv.hip) with the roles in the second half swapped.
//   * a wave owns 64 consecutive rows and streams their cols/vals lane-consecutively into its LDS slice
//     (coalesced, predicated, every byte of the matrix fetched once);
//   * then lane = (row-in-group r, pair of right-hand sides jp), 16 x 4: the four lanes of a row read the same
//     (col, val) from LDS (broadcast) and gather the 64-byte line X[col*8 .. col*8+7] with one 16-byte load each;
//     a wave works through its 64 rows in 4 groups of 16.  Every (row, j) sum runs in CSR order in one lane, so
//     each column of Y is bit-identical to the single-vector product.
//   * epilogues as in spmv.hip; the dot partials are per right-hand side.
#include "common.hpp"

#include <algorithm>

namespace padne {

constexpr int kSpmmChunk = 512;      // non-zeros staged per wave per pass (6 KiB: 4 B col + 8 B val)

// K = 8, 4 or 2 right-hand sides (the narrow forms serve one to three regulators, pcg.hip): K / 2 lanes share a row,
// 128 / K rows form a group, a wave works through its 64 rows in K / 2 groups.
// ST (default XT): the type the multiplied vectors are STORED in.  <K, SPMV_DOT, double, double, double, float> -- q = A p with the
// search directions kept as floats, multiplied in double, the single loop's form -- was built and measured in round 5 and is NOT
// used: two conversions per non-zero and lane run at the rate of double-precision arithmetic, the product became bound by
// them (512 against 330 us at N = 5 M).  The parameter stays for the gathers' type.
template <int K, int MODE, typename VT, typename XT, typename YT, typename ST = XT>
__global__ __launch_bounds__(kSpmvThreads) void csr_spmm_kernel(
    const int n_rows, const int n_wtiles, const int *__restrict__ rowptr, const int *__restrict__ cols,
    const VT *__restrict__ vals, const ST *__restrict__ x, YT *__restrict__ y, const double *__restrict__ dot_with,
    double *__restrict__ partials /* [8][kMaxPartials] */, const int *__restrict__ done_flag,
    const XT *__restrict__ aux1, const XT *__restrict__ aux2, const XT scale,
    const double *__restrict__ out_scale2 /* [K] or null */, const XT *__restrict__ aux0 /* SPMV_WUP: [n][K] */,
    const XT *__restrict__ rhs /* SPMV_WUP exit without aux0: the fine level's right-hand side [n][K] */) {
    static_assert(K == 8 || K == 4 || K == 2, "lockstep widths");
    constexpr int LPR = K / 2;                              // lanes per row (each takes two right-hand sides)
    constexpr int RPG = 64 / LPR;                           // rows per group
    constexpr int NG = LPR;                                 // groups per 64-row tile
    constexpr bool WITH_DOT = (MODE == SPMV_DOT) || (MODE == SPMV_JACOBI) || (MODE == SPMV_WUP);
    __shared__ int cs_all[4 * kSpmmChunk];
    __shared__ VT vs_all[4 * kSpmmChunk];
    __shared__ double red[4][K];

    if (done_flag != nullptr && *done_flag != 0) return;

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int rsub = lane / LPR, j = (lane % LPR) * 2;     // this lane's columns: j, j + 1
    int *cs = cs_all + w * kSpmmChunk;
    VT *vs = vs_all + w * kSpmmChunk;
    double out_mul0 = 1.0, out_mul1 = 1.0;
    if (out_scale2 != nullptr) {
        const double a0 = out_scale2[j], a1 = out_scale2[j + 1];
        out_mul0 = a0 > 0.0 ? sqrt(a0) : 1.0;
        out_mul1 = a1 > 0.0 ? sqrt(a1) : 1.0;
    }
    const int G = gridDim.x;
    const int nslab = (G % kNumXcd == 0) ? kNumXcd : 1;
    const int slab = blockIdx.x % nslab;
    const int wx = (blockIdx.x / nslab) * 4 + w;
    const int wps = (G / nslab) * 4;
    const int s0 = (int)((long long)slab * n_wtiles / nslab);
    const int s1 = (int)((long long)(slab + 1) * n_wtiles / nslab);
    static_assert(sizeof(ST) == sizeof(XT) || MODE == SPMV_DOT || MODE == SPMV_PLAIN, "stored type: products only");
    struct alignas(2 * sizeof(XT)) X2 { XT a, b; };
    struct alignas(2 * sizeof(ST)) S2 { ST a, b; };
    struct alignas(2 * sizeof(YT)) Y2 { YT a, b; };

    double dot0 = 0.0, dot1 = 0.0;
    for (int wt = s0 + wx; wt < s1; wt += wps) {
        const int row0 = wt * 64;
        const int row1 = min(row0 + 64, n_rows);
        int rs = 0, re = 0;
        if (row0 + lane < row1) {
            rs = rowptr[row0 + lane];
            re = rowptr[row0 + lane + 1];
        }
        const int k0 = __shfl(rs, 0, 64);
        const int k1 = __shfl(re, row1 - row0 - 1, 64);
        XT acc0[NG], acc1[NG];
        int grs[NG], gre[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            acc0[g] = 0;
            acc1[g] = 0;
            grs[g] = __shfl(rs, g * RPG + rsub, 64);
            gre[g] = __shfl(re, g * RPG + rsub, 64);
        }
        for (int base = k0; base < k1; base += kSpmmChunk) {
#pragma unroll
            for (int q = 0; q < kSpmmChunk / 64; ++q) {
                const int e = base + lane + 64 * q;
                if (e < k1) {
                    cs[lane + 64 * q] = cols[e];
                    vs[lane + 64 * q] = vals[e];
                }
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int lo = max(grs[g], base), hi = min(gre[g], base + kSpmmChunk);
                XT a0 = acc0[g], a1 = acc1[g];
                for (int k = lo; k < hi; k += 8) {             // up to eight 16-byte gathers in flight per lane
                    S2 xv[8];
                    XT vv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        xv[u].a = 0;
                        xv[u].b = 0;
                        vv[u] = 0;
                        if (k + u < hi) {
                            xv[u] = *reinterpret_cast<const S2 *>(x + (size_t)cs[k + u - base] * K + j);
                            vv[u] = (XT)vs[k + u - base];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (k + u < hi) {
                            a0 += vv[u] * (XT)xv[u].a;
                            a1 += vv[u] * (XT)xv[u].b;
                        }
                }
                acc0[g] = a0;
                acc1[g] = a1;
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int r = row0 + g * RPG + rsub;
            if (r >= row1) continue;
            const size_t o = (size_t)r * K + j;
            const XT a0 = acc0[g], a1 = acc1[g];
            Y2 out2;
            if (MODE == SPMV_PLAIN) {
                out2.a = (YT)a0;
                out2.b = (YT)a1;
            } else if (MODE == SPMV_DOT) {
                out2.a = (YT)a0;
                out2.b = (YT)a1;
                if (sizeof(ST) != sizeof(XT)) {                    // p.q with the stored p
                    const S2 xs = *reinterpret_cast<const S2 *>(x + o);
                    dot0 += (double)xs.a * (double)a0;
                    dot1 += (double)xs.b * (double)a1;
                } else {
                    dot0 += dot_with[o] * (double)a0;
                    dot1 += dot_with[o + 1] * (double)a1;
                }
            } else if (MODE == SPMV_RESID) {
                const X2 b = *reinterpret_cast<const X2 *>(aux1 + o);
                out2.a = (YT)(b.a - a0);
                out2.b = (YT)(b.b - a1);

            } else if (MODE == SPMV_ADD) {
                const Y2 old = *reinterpret_cast<const Y2 *>(y + o);
                out2.a = old.a + (YT)a0;
                out2.b = old.b + (YT)a1;
            } else if (MODE == SPMV_WUP) {
                // exit stage of the cycle in the W form (spmv.hip): z = x_pre + c D^-1 r_pre + W e, r.z partials
                const X2 rp = *reinterpret_cast<const X2 *>(aux1 + o);
                const XT d = scale * aux2[r];
                XT o0, o1;
                X2 rb;
                rb.a = 0;
                rb.b = 0;
                if (rhs != nullptr) {
                    // exit stage of the fine level formed from its right-hand side (spmv.hip, SPMV_WUP with y2): the pre-smoothed
                    // iterate is c D^-1 of it, so x_pre + c D^-1 r_pre = c D^-1 (b + r_pre) and x_pre is not read; r.z is taken
                    // against it too (4 instead of 8 bytes per row and right-hand side).  (The residual of the sweep from
                    // zero formed from the right-hand side as well -- b and 1/diag gathered per non-zero, the single
                    // cycle's form -- was measured in round 5 and is not used here: 69.3 against 66.9 ms for config C5, the
                    // 8-wide product pays for every instruction per non-zero.)
                    rb = *reinterpret_cast<const X2 *>(rhs + o);
                    o0 = d * (rb.a + rp.a) + a0;
                    o1 = d * (rb.b + rp.b) + a1;
                } else {
                    const X2 xp = *reinterpret_cast<const X2 *>(aux0 + o);
                    o0 = xp.a + d * rp.a + a0;
                    o1 = xp.b + d * rp.b + a1;
                }
                if (dot_with != nullptr || rhs != nullptr) {
                    const double d0 = (double)o0 * out_mul0, d1 = (double)o1 * out_mul1;
                    // (a float result leaves unscaled, as in spmv.hip: the consumer multiplies, the same double comes out)
                    out2.a = sizeof(YT) == 4 ? (YT)o0 : (YT)d0;
                    out2.b = sizeof(YT) == 4 ? (YT)o1 : (YT)d1;
                    if (rhs != nullptr) {
                        dot0 += ((double)rb.a * out_mul0) * d0;
                        dot1 += ((double)rb.b * out_mul1) * d1;
                    } else {
                        dot0 += dot_with[o] * d0;
                        dot1 += dot_with[o + 1] * d1;
                    }
                } else {      // inner level of the cycle
                    out2.a = (YT)o0;
                    out2.b = (YT)o1;
                }
            } else {
                const X2 b = *reinterpret_cast<const X2 *>(aux1 + o);
                const S2 xo = *reinterpret_cast<const S2 *>(x + o);
                const XT d = scale * aux2[r];
                const XT o0 = (XT)xo.a + d * (b.a - a0), o1 = (XT)xo.b + d * (b.b - a1);
                if (dot_with != nullptr) {
                    const double d0 = (double)o0 * out_mul0, d1 = (double)o1 * out_mul1;
                    out2.a = sizeof(YT) == 4 ? (YT)o0 : (YT)d0;
                    out2.b = sizeof(YT) == 4 ? (YT)o1 : (YT)d1;
                    dot0 += dot_with[o] * d0;
                    dot1 += dot_with[o + 1] * d1;
                } else {
                    out2.a = (YT)o0;
                    out2.b = (YT)o1;
                    dot0 += (double)(b.a * o0);
                    dot1 += (double)(b.b * o1);
                }
            }
            *reinterpret_cast<Y2 *>(y + o) = out2;
        }
    }
    if (WITH_DOT && partials != nullptr) {
        // lanes with the same column pair: LPR, 2 LPR, ... 32 apart
#pragma unroll
        for (int d = LPR; d < 64; d <<= 1) {
            dot0 += __shfl_xor(dot0, d, 64);
            dot1 += __shfl_xor(dot1, d, 64);
        }
        if (lane < LPR) {
            red[w][lane * 2] = dot0;
            red[w][lane * 2 + 1] = dot1;
        }
        __syncthreads();
        if (threadIdx.x < K)
            partials[(size_t)threadIdx.x * kMaxPartials + blockIdx.x] =
                (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    }
}

template <int K, typename VT, typename XT, typename YT>
static int launch_spmm_typed(padne_ctx *ctx, const padne_csr *m, const VT *vals, int mode, const XT *x, YT *y,
                             const double *dot_with, double *partials, const int32_t *done_flag, const XT *aux1,
                             const XT *aux2, XT scale, const double *out_scale2, const XT *aux0 = nullptr,
                             const XT *rhs = nullptr) {
    if (m->n_rows == 0) return PADNE_OK;
    const int n_tiles = (int)((m->n_rows + 63) / 64);
    long long g = (m->n_rows + kSpmvRows - 1) / kSpmvRows;
    if (g > kMaxPartials) g = kMaxPartials;
    if (g >= kNumXcd) g -= g % kNumXcd;
    if (g < 1) g = 1;
#define PADNE_SPMM_LAUNCH(M)                                                                                      \
    hipLaunchKernelGGL((csr_spmm_kernel<K, M, VT, XT, YT>), dim3((unsigned)g), dim3(kSpmvThreads), 0, ctx->stream, \
                       (int)m->n_rows, n_tiles, m->rowptr, m->cols, vals, x, y, dot_with, partials, done_flag,     \
                       aux1, aux2, scale, out_scale2, aux0, rhs)
    switch (mode) {
        case SPMV_PLAIN: PADNE_SPMM_LAUNCH(SPMV_PLAIN); break;
        case SPMV_DOT: PADNE_SPMM_LAUNCH(SPMV_DOT); break;
        case SPMV_RESID: PADNE_SPMM_LAUNCH(SPMV_RESID); break;
        case SPMV_ADD: PADNE_SPMM_LAUNCH(SPMV_ADD); break;
        case SPMV_JACOBI: PADNE_SPMM_LAUNCH(SPMV_JACOBI); break;
        case SPMV_WUP: PADNE_SPMM_LAUNCH(SPMV_WUP); break;
        default: set_error("bad SpMM mode %d", mode); return PADNE_E_INVALID;
    }
#undef PADNE_SPMM_LAUNCH
    PADNE_HIP_CHECK(hipGetLastError());
    return PADNE_OK;
}

int spmm8_grid(const padne_csr *m) {
    long long g = (m->n_rows + kSpmvRows - 1) / kSpmvRows;
    if (g > kMaxPartials) g = kMaxPartials;
    if (g >= kNumXcd) g -= g % kNumXcd;
    return (int)(g < 1 ? 1 : g);
}

// the width is a run-time choice of the caller (8 for config C5 and groups of regulators, 4 / 2 for one to three)
#define PADNE_SPMM_WIDTH(k, CALL8, CALL4, CALL2)                       \
    switch (k) {                                                       \
        case 8: return CALL8;                                          \
        case 4: return CALL4;                                          \
        case 2: return CALL2;                                          \
        default: set_error("lockstep width %d", k); return PADNE_E_INVALID; \
    }

int launch_spmm_mode(padne_ctx *ctx, const padne_csr *m, int k, int mode, const double *x, double *y, const double *dot_with,
                     double *partials, const int32_t *done_flag, const double *aux1, const double *aux2, double scale) {
#define ARGS ctx, m, m->vals, mode, x, y, dot_with, partials, done_flag, aux1, aux2, scale, nullptr
    PADNE_SPMM_WIDTH(k, (launch_spmm_typed<8, double, double, double>(ARGS)), (launch_spmm_typed<4, double, double, double>(ARGS)),
                     (launch_spmm_typed<2, double, double, double>(ARGS)))
#undef ARGS
}

int launch_spmm_f32(padne_ctx *ctx, const padne_csr *m, int k, int mode, const float *x, float *y, double *partials,
                    const int32_t *done_flag, const float *aux1, const float *aux2, float scale) {
    PADNE_REQUIRE(m->vals32 != nullptr, "single-precision copy missing");
#define ARGS ctx, m, m->vals32, mode, x, y, nullptr, partials, done_flag, aux1, aux2, scale, nullptr
    PADNE_SPMM_WIDTH(k, (launch_spmm_typed<8, float, float, float>(ARGS)), (launch_spmm_typed<4, float, float, float>(ARGS)),
                     (launch_spmm_typed<2, float, float, float>(ARGS)))
#undef ARGS
}

// (y32 instead of y: z leaves in single precision and without its factor, the r.z partials are those of the double)
int launch_spmm_f32_exit(padne_ctx *ctx, const padne_csr *m, int k, const float *x, double *y, const double *dot_with,
                         double *partials, const int32_t *done_flag, const float *aux1, const float *aux2,
                         float scale, const double *out_scale2, float *y32) {
    PADNE_REQUIRE(m->vals32 != nullptr && dot_with != nullptr, "single-precision exit stage");
    if (y32 != nullptr) {
#define ARGS ctx, m, m->vals32, SPMV_JACOBI, x, y32, dot_with, partials, done_flag, aux1, aux2, scale, out_scale2
        PADNE_SPMM_WIDTH(k, (launch_spmm_typed<8, float, float, float>(ARGS)), (launch_spmm_typed<4, float, float, float>(ARGS)),
                         (launch_spmm_typed<2, float, float, float>(ARGS)))
#undef ARGS
    }
#define ARGS ctx, m, m->vals32, SPMV_JACOBI, x, y, dot_with, partials, done_flag, aux1, aux2, scale, out_scale2
    PADNE_SPMM_WIDTH(k, (launch_spmm_typed<8, float, float, double>(ARGS)), (launch_spmm_typed<4, float, float, double>(ARGS)),
                     (launch_spmm_typed<2, float, float, double>(ARGS)))
#undef ARGS
}

// last stage of the lockstep cycle in the W form: z = (x_pre + c D^-1 r_pre + W e) * sqrt(out_scale2[j]) in double, with the
// partial sums of dot_with . z per right-hand side (the lockstep counterpart of launch_spmv_f32_wup_exit)
int launch_spmm_f32_wup_exit(padne_ctx *ctx, const padne_csr *w, int k, const float *e, double *z, const double *dot_with,
                             double *partials, const int32_t *done_flag, const float *x_pre, const float *r_pre,
                             const float *dinv32, float scale, const double *out_scale2, float *z32, const float *rhs) {
    PADNE_REQUIRE(w->vals32 != nullptr && (dot_with != nullptr || rhs != nullptr) && (x_pre != nullptr || rhs != nullptr),
                  "single-precision W stage");
    if (z32 != nullptr) {
#define ARGS ctx, w, w->vals32, SPMV_WUP, e, z32, dot_with, partials, done_flag, r_pre, dinv32, scale, out_scale2, x_pre, rhs
        PADNE_SPMM_WIDTH(k, (launch_spmm_typed<8, float, float, float>(ARGS)), (launch_spmm_typed<4, float, float, float>(ARGS)),
                         (launch_spmm_typed<2, float, float, float>(ARGS)))
#undef ARGS
    }
#define ARGS ctx, w, w->vals32, SPMV_WUP, e, z, dot_with, partials, done_flag, r_pre, dinv32, scale, out_scale2, x_pre, rhs
    PADNE_SPMM_WIDTH(k, (launch_spmm_typed<8, float, float, double>(ARGS)), (launch_spmm_typed<4, float, float, double>(ARGS)),
                     (launch_spmm_typed<2, float, float, double>(ARGS)))
#undef ARGS
}

// up-leg of an inner level in the W form: x = x_pre + c D^-1 r_pre + W e, single precision throughout
int launch_spmm_f32_wup(padne_ctx *ctx, const padne_csr *w, int k, const float *e, float *x_out, const int32_t *done_flag,
                        const float *x_pre, const float *r_pre, const float *dinv32, float scale) {
    PADNE_REQUIRE(w->vals32 != nullptr, "single-precision W stage");
#define ARGS ctx, w, w->vals32, SPMV_WUP, e, x_out, nullptr, nullptr, done_flag, r_pre, dinv32, scale, nullptr, x_pre
    PADNE_SPMM_WIDTH(k, (launch_spmm_typed<8, float, float, float>(ARGS)), (launch_spmm_typed<4, float, float, float>(ARGS)),
                     (launch_spmm_typed<2, float, float, float>(ARGS)))
#undef ARGS
}

int launch_spmm8_mode(padne_ctx *ctx, const padne_csr *m, int mode, const double *x, double *y, const double *dot_with,
                      double *partials, const int32_t *done_flag, const double *aux1, const double *aux2,
                      double scale) {
    return launch_spmm_mode(ctx, m, kSpmmK, mode, x, y, dot_with, partials, done_flag, aux1, aux2, scale);
}

// [k][n] (one vector after the other) <-> [n][k] (interleaved)
__global__ void interleave_kernel(long long n, int k, const double *__restrict__ src, double *__restrict__ dst, int to_interleaved) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * k) return;
    const long long i = t / k;
    const int j = (int)(t % k);
    if (to_interleaved) dst[t] = src[(size_t)j * n + i];
    else dst[(size_t)j * n + i] = src[t];
}

int interleave(padne_ctx *ctx, long long n, int k, const double *src, double *dst, bool to_interleaved) {
    if (n <= 0) return PADNE_OK;
    hipLaunchKernelGGL(interleave_kernel, dim3((unsigned)((n * k + 255) / 256)), dim3(256), 0, ctx->stream, n, k, src,
                       dst, to_interleaved ? 1 : 0);
    PADNE_HIP_CHECK(hipGetLastError());
    return PADNE_OK;
}

}  // namespace padne

using namespace padne;

// Y = M X for 8 interleaved vectors on the device (X: n_cols x 8, Y: n_rows x 8, row-major), `repeat` launches
extern "C" int padne_spmm8_dev(padne_ctx *ctx, const padne_csr *m, const void *x_dev, void *y_dev, int repeat) {
    PADNE_REQUIRE(ctx && m && x_dev && y_dev, "null argument");
    PADNE_REQUIRE(repeat >= 1, "repeat");
    PADNE_HIP_CHECK(hipSetDevice(ctx->device));
    for (int i = 0; i < repeat; ++i)
        PADNE_TRY(launch_spmm8_mode(ctx, m, SPMV_PLAIN, (const double *)x_dev, (double *)y_dev, nullptr, nullptr, nullptr,
                                    nullptr, nullptr, 0.0));
    PADNE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PADNE_OK;
}

extern "C" int padne_spmm8_time(padne_ctx *ctx, const padne_csr *m, const void *x_dev, void *y_dev, int warmup,
                                int repeat, double *seconds_per_launch) {
    PADNE_REQUIRE(ctx && m && x_dev && y_dev && seconds_per_launch, "null argument");
    PADNE_REQUIRE(repeat >= 1 && warmup >= 0, "repeat / warmup");
    PADNE_HIP_CHECK(hipSetDevice(ctx->device));
    for (int i = 0; i < warmup; ++i)
        PADNE_TRY(launch_spmm8_mode(ctx, m, SPMV_PLAIN, (const double *)x_dev, (double *)y_dev, nullptr, nullptr, nullptr,
                                    nullptr, nullptr, 0.0));
    PADNE_HIP_CHECK(hipEventRecord(ctx->ev0, ctx->stream));
    for (int i = 0; i < repeat; ++i)
        PADNE_TRY(launch_spmm8_mode(ctx, m, SPMV_PLAIN, (const double *)x_dev, (double *)y_dev, nullptr, nullptr, nullptr,
                                    nullptr, nullptr, 0.0));
    PADNE_HIP_CHECK(hipEventRecord(ctx->ev1, ctx->stream));
    PADNE_HIP_CHECK(hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    PADNE_HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *seconds_per_launch = ms * 1e-3 / repeat;
    return PADNE_OK;
}

// algorithmic bytes of one 8-vector product: 12*nnz + 4*n_rows + 16*n_rows*8 + 4  
extern "C" int64_t padne_spmm8_algorithmic_bytes(const padne_csr *m) {
    if (!m) return 0;
    return 12 * m->nnz + 4 * m->n_rows + 16 * m->n_rows * kSpmmK + 4;
}
