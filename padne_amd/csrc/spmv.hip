// CSR SpMV for gfx950 (MI355X): y = A x, f64 values, i32 indices.
//
// Bandwidth-bound (0.135 flop/B), so no MFMA: the whole design is about moving
// 12 B per non-zero + 20 B per row exactly once at full HBM rate.
//
//  * a workgroup (4 waves of 64) owns a contiguous run of 256-row tiles; tiles
//    are dealt to workgroups so that the workgroups of one XCD (blockIdx % 8)
//    cover one contiguous slab of the matrix: the x entries a tile gathers are
//    shared with the neighbouring tiles, and each XCD has a private L2.
//  * the non-zeros of a tile are one contiguous range of `cols`/`vals`; the
//    workgroup streams that range with 16-byte-per-lane loads (int4 columns,
//    2 x double2 values per 4 non-zeros), fully coalesced and independent of
//    the row structure.  Arrays are padded so no bounds checks are needed.
//  * each lane gathers x[col] (L1/L2 hits: mesh neighbours are close in
//    index), multiplies, and parks the products in LDS (16 KiB per pass).
//  * one lane per row then adds its row segment out of LDS in CSR order, so
//    the result is bit-identical to a sequential CSR product (scipy's
//    csr_matvec); rows longer than a pass simply span several passes.
//  * optional epilogue: partial sums of dot_with[row]*y[row], reduced over the
//    wave with DPP shuffles and over the workgroup through LDS, one partial
//    per workgroup (deterministic, no atomics).
#include "common.hpp"

namespace padne {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// sum over a 256-thread workgroup; result valid in thread 0. `red` = 4 doubles of LDS.
__device__ __forceinline__ double block_sum_256(double v, double *red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) red[w] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0) s = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return s;
}

template <bool WITH_DOT>
__global__ __launch_bounds__(kSpmvThreads) void csr_spmv_kernel(
    const int n_rows, const int n_tiles, const int *__restrict__ rowptr,
    const int *__restrict__ cols, const double *__restrict__ vals,
    const double *__restrict__ x, double *__restrict__ y,
    const double *__restrict__ dot_with, double *__restrict__ partials,
    const int *__restrict__ done_flag) {
    __shared__ __attribute__((aligned(16))) double prod[kSpmvTileNnz];
    __shared__ double red[4];

    if (done_flag != nullptr && *done_flag != 0) return;

    const int tid = threadIdx.x;
    // XCD-aware tile assignment: workgroups b, b+8, b+16, ... share an XCD (and its L2);
    // give them consecutive slabs of tiles.
    const int G = gridDim.x;
    const int per_xcd = G / kNumXcd;                     // G is a multiple of 8 (or < 8)
    int vb = blockIdx.x;
    if (per_xcd > 0 && G % kNumXcd == 0) vb = (blockIdx.x % kNumXcd) * per_xcd + blockIdx.x / kNumXcd;
    const long long t0 = (long long)vb * n_tiles / G;
    const long long t1 = (long long)(vb + 1) * n_tiles / G;

    double dot_acc = 0.0;

    for (int tile = (int)t0; tile < (int)t1; ++tile) {
        const int row0 = tile * kSpmvRows;
        const int row1 = min(row0 + kSpmvRows, n_rows);
        const int k0 = rowptr[row0];
        const int k1 = rowptr[row1];
        const int r = row0 + tid;
        int rs = 0, re = 0;
        if (r < row1) {
            rs = rowptr[r];
            re = rowptr[r + 1];
        }
        double acc = 0.0;
        for (int base = k0 & ~3; base < k1; base += kSpmvTileNnz) {
#pragma unroll
            for (int j = 0; j < kSpmvTileNnz / (4 * kSpmvThreads); ++j) {
                const int l = 4 * (tid + kSpmvThreads * j);
                const int e = base + l;
                const int4 c = *reinterpret_cast<const int4 *>(cols + e);
                const double2 v01 = *reinterpret_cast<const double2 *>(vals + e);
                const double2 v23 = *reinterpret_cast<const double2 *>(vals + e + 2);
                const double x0 = x[c.x], x1 = x[c.y], x2 = x[c.z], x3 = x[c.w];
                double2 p01, p23;
                p01.x = v01.x * x0;
                p01.y = v01.y * x1;
                p23.x = v23.x * x2;
                p23.y = v23.y * x3;
                *reinterpret_cast<double2 *>(prod + l) = p01;
                *reinterpret_cast<double2 *>(prod + l + 2) = p23;
            }
            __syncthreads();
            const int lo = max(rs, base), hi = min(re, base + kSpmvTileNnz);
            for (int k = lo; k < hi; ++k) acc += prod[k - base];
            __syncthreads();
        }
        if (r < row1) {
            y[r] = acc;
            if (WITH_DOT) dot_acc += dot_with[r] * acc;
        }
    }
    if (WITH_DOT) {
        const double s = block_sum_256(dot_acc, red);
        if (tid == 0) partials[blockIdx.x] = s;
    }
}

int spmv_grid(const padne_csr *m) {
    const long long n_tiles = (m->n_rows + kSpmvRows - 1) / kSpmvRows;
    long long g = n_tiles < kMaxPartials ? n_tiles : kMaxPartials;
    if (g >= kNumXcd) g -= g % kNumXcd;
    if (g < 1) g = 1;
    return (int)g;
}

int launch_spmv(padne_ctx *ctx, const padne_csr *m, const double *x, double *y,
                const double *dot_with, double *partials, const int32_t *done_flag) {
    if (m->n_rows == 0) return PADNE_OK;
    const int n_tiles = (int)((m->n_rows + kSpmvRows - 1) / kSpmvRows);
    const int g = spmv_grid(m);
    if (dot_with != nullptr) {
        hipLaunchKernelGGL(csr_spmv_kernel<true>, dim3(g), dim3(kSpmvThreads), 0, ctx->stream,
                           (int)m->n_rows, n_tiles, m->rowptr, m->cols, m->vals, x, y, dot_with,
                           partials, done_flag);
    } else {
        hipLaunchKernelGGL(csr_spmv_kernel<false>, dim3(g), dim3(kSpmvThreads), 0, ctx->stream,
                           (int)m->n_rows, n_tiles, m->rowptr, m->cols, m->vals, x, y, nullptr,
                           nullptr, done_flag);
    }
    PADNE_HIP_CHECK(hipGetLastError());
    return PADNE_OK;
}

// ---- 1/diag ---------------------------------------------------------------------------------
__global__ void csr_dinv_kernel(int n_rows, const int *__restrict__ rowptr,
                                const int *__restrict__ cols, const double *__restrict__ vals,
                                double *__restrict__ dinv) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    double d = 0.0;
    for (int k = rowptr[r]; k < rowptr[r + 1]; ++k)
        if (cols[k] == r) d += vals[k];
    dinv[r] = 1.0 / d;
}

int csr_build_dinv(padne_ctx *ctx, padne_csr *m) {
    if (m->dinv != nullptr) return PADNE_OK;
    PADNE_REQUIRE(m->n_rows <= m->n_cols, "Jacobi needs a diagonal");
    PADNE_HIP_CHECK(hipMalloc((void **)&m->dinv, sizeof(double) * (size_t)(m->n_rows > 0 ? m->n_rows : 1)));
    if (m->n_rows > 0) {
        const int bs = 256;
        hipLaunchKernelGGL(csr_dinv_kernel, dim3((unsigned)((m->n_rows + bs - 1) / bs)), dim3(bs), 0,
                           ctx->stream, (int)m->n_rows, m->rowptr, m->cols, m->vals, m->dinv);
        PADNE_HIP_CHECK(hipGetLastError());
    }
    return PADNE_OK;
}

}  // namespace padne
