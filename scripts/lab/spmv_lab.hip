// SpMV kernel laboratory (not part of the product): times design variants of the CSR SpMV on a
// synthetic 7-point triangular-grid matrix of the C4 shape (8 layers of 1118x1118).
//   hipcc -O3 --offload-arch=gfx950 -o spmv_lab scripts/spmv_lab.hip && ./spmv_lab
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int NXCD = 8;

__device__ __forceinline__ int vblock(int G) {
    const int per = G / NXCD;
    if (per > 0 && G % NXCD == 0) return (blockIdx.x % NXCD) * per + blockIdx.x / NXCD;
    return blockIdx.x;
}

// ---- V0/V1: LDS-staged, TILE nnz per pass, ROWS rows per tile, optional non-temporal streams ----
template <int TILE, int ROWS, bool NT, bool PRED = false>
__global__ __launch_bounds__(256) void spmv_lds(int n_rows, int n_tiles, const int *__restrict__ rowptr,
                                                const int *__restrict__ cols, const double *__restrict__ vals,
                                                const double *__restrict__ x, double *__restrict__ y) {
    __shared__ __attribute__((aligned(16))) double prod[TILE];
    const int tid = threadIdx.x;
    const int G = gridDim.x;
    const int vb = vblock(G);
    const long long t0 = (long long)vb * n_tiles / G, t1 = (long long)(vb + 1) * n_tiles / G;
    for (int tile = (int)t0; tile < (int)t1; ++tile) {
        const int row0 = tile * ROWS;
        const int row1 = min(row0 + ROWS, n_rows);
        const int k0 = rowptr[row0], k1 = rowptr[row1];
        int rs[ROWS / 256], re[ROWS / 256];
#pragma unroll
        for (int q = 0; q < ROWS / 256; ++q) {
            const int r = row0 + tid + 256 * q;
            rs[q] = re[q] = 0;
            if (r < row1) { rs[q] = rowptr[r]; re[q] = rowptr[r + 1]; }
        }
        double acc[ROWS / 256];
#pragma unroll
        for (int q = 0; q < ROWS / 256; ++q) acc[q] = 0.0;
        for (int base = k0 & ~3; base < k1; base += TILE) {
#pragma unroll
            for (int j = 0; j < TILE / 1024; ++j) {
                const int l = 4 * (tid + 256 * j);
                const int e = base + l;
                int4 c = make_int4(0, 0, 0, 0); double2 v01 = make_double2(0, 0), v23 = make_double2(0, 0);
                if (PRED && e >= k1) {
                    // nothing to fetch: this quad belongs to the next tile
                } else if (NT) {
                    c.x = __builtin_nontemporal_load(cols + e); c.y = __builtin_nontemporal_load(cols + e + 1);
                    c.z = __builtin_nontemporal_load(cols + e + 2); c.w = __builtin_nontemporal_load(cols + e + 3);
                    v01.x = __builtin_nontemporal_load(vals + e); v01.y = __builtin_nontemporal_load(vals + e + 1);
                    v23.x = __builtin_nontemporal_load(vals + e + 2); v23.y = __builtin_nontemporal_load(vals + e + 3);
                } else {
                    c = *reinterpret_cast<const int4 *>(cols + e);
                    v01 = *reinterpret_cast<const double2 *>(vals + e);
                    v23 = *reinterpret_cast<const double2 *>(vals + e + 2);
                }
                double2 p01, p23;
                p01.x = v01.x * x[c.x]; p01.y = v01.y * x[c.y];
                p23.x = v23.x * x[c.z]; p23.y = v23.y * x[c.w];
                *reinterpret_cast<double2 *>(prod + l) = p01;
                *reinterpret_cast<double2 *>(prod + l + 2) = p23;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < ROWS / 256; ++q) {
                const int lo = max(rs[q], base), hi = min(re[q], base + TILE);
                for (int k = lo; k < hi; ++k) acc[q] += prod[k - base];
            }
            __syncthreads();
        }
#pragma unroll
        for (int q = 0; q < ROWS / 256; ++q) {
            const int r = row0 + tid + 256 * q;
            if (r < row1) y[r] = acc[q];
        }
    }
}


// ---- V6: LDS-staged, lane-consecutive element mapping (element = base + tid + 256*j): adjacent lanes
// gather adjacent non-zeros, whose columns are mostly adjacent in x -> far fewer L1 tag lookups ----
template <int TILE, int ROWS, int UNROLL>
__global__ __launch_bounds__(256) void spmv_lds_lc(int n_rows, int n_tiles, const int *__restrict__ rowptr,
                                                   const int *__restrict__ cols, const double *__restrict__ vals,
                                                   const double *__restrict__ x, double *__restrict__ y) {
    __shared__ double prod[TILE];
    const int tid = threadIdx.x;
    const int G = gridDim.x;
    const int vb = vblock(G);
    const long long t0 = (long long)vb * n_tiles / G, t1 = (long long)(vb + 1) * n_tiles / G;
    for (int tile = (int)t0; tile < (int)t1; ++tile) {
        const int row0 = tile * ROWS;
        const int row1 = min(row0 + ROWS, n_rows);
        const int k0 = rowptr[row0], k1 = rowptr[row1];
        int rs[ROWS / 256], re[ROWS / 256];
#pragma unroll
        for (int q = 0; q < ROWS / 256; ++q) {
            const int r = row0 + tid + 256 * q;
            rs[q] = re[q] = 0;
            if (r < row1) { rs[q] = rowptr[r]; re[q] = rowptr[r + 1]; }
        }
        double acc[ROWS / 256];
#pragma unroll
        for (int q = 0; q < ROWS / 256; ++q) acc[q] = 0.0;
        for (int base = k0; base < k1; base += TILE) {
            int c[TILE / 256];
            double v[TILE / 256];
#pragma unroll
            for (int j = 0; j < TILE / 256; ++j) {
                c[j] = cols[base + tid + 256 * j];
                v[j] = vals[base + tid + 256 * j];
            }
#pragma unroll
            for (int j = 0; j < TILE / 256; ++j) prod[tid + 256 * j] = v[j] * x[c[j]];
            __syncthreads();
#pragma unroll
            for (int q = 0; q < ROWS / 256; ++q) {
                const int lo = max(rs[q], base), hi = min(re[q], base + TILE);
                for (int k = lo; k < hi; ++k) acc[q] += prod[k - base];
            }
            __syncthreads();
        }
#pragma unroll
        for (int q = 0; q < ROWS / 256; ++q) {
            const int r = row0 + tid + 256 * q;
            if (r < row1) y[r] = acc[q];
        }
    }
}


// ---- V8: wave-private tiles (64 rows per wave), no workgroup barrier: each wave streams its own
// non-zero range (lane-consecutive, predicated), parks products in its private 4 KiB LDS slice and
// reduces its 64 rows.  LDS operations of one wave execute in order, so write->read needs no barrier.
template <int EPL>
__global__ __launch_bounds__(256) void spmv_wave(int n_rows, int n_wtiles, const int *__restrict__ rowptr,
                                                 const int *__restrict__ cols, const double *__restrict__ vals,
                                                 const double *__restrict__ x, double *__restrict__ y) {
    constexpr int CH = 64 * EPL;
    __shared__ double prod_all[4 * CH];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double *prod = prod_all + w * CH;
    const int G = gridDim.x;
    const int vb = vblock(G);
    const long long W = (long long)G * 4, gw = (long long)vb * 4 + w;
    const int t0 = (int)(gw * n_wtiles / W), t1 = (int)((gw + 1) * n_wtiles / W);
    for (int wt = t0; wt < t1; ++wt) {
        const int row0 = wt * 64;
        const int row1 = min(row0 + 64, n_rows);
        const int r = row0 + lane;
        int rs = 0, re = 0;
        if (r < row1) { rs = rowptr[r]; re = rowptr[r + 1]; }
        const int k0 = __shfl(rs, 0, 64);
        const int k1 = __shfl(re, row1 - row0 - 1, 64);
        double acc = 0.0;
        for (int base = k0; base < k1; base += CH) {
            int c[EPL];
            double v[EPL];
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                const int e = base + lane + 64 * j;
                c[j] = 0; v[j] = 0.0;
                if (e < k1) { c[j] = cols[e]; v[j] = vals[e]; }
            }
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                const int e = base + lane + 64 * j;
                double xv = 0.0;
                if (e < k1) xv = x[c[j]];
                prod[lane + 64 * j] = v[j] * xv;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const int lo = max(rs, base), hi = min(re, base + CH);
            for (int k = lo; k < hi; ++k) acc += prod[k - base];
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        if (r < row1) y[r] = acc;
    }
}


// ---- V9: V8 with interleaved assignment: the waves of one XCD sweep that XCD's slab of wave-tiles
// together, CHUNK wave-tiles per wave per turn, so co-resident waves work on neighbouring rows ----
template <int EPL, int CHUNK>
__global__ __launch_bounds__(256) void spmv_wave_rr(int n_rows, int n_wtiles, const int *__restrict__ rowptr,
                                                    const int *__restrict__ cols, const double *__restrict__ vals,
                                                    const double *__restrict__ x, double *__restrict__ y) {
    constexpr int CH = 64 * EPL;
    __shared__ double prod_all[4 * CH];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double *prod = prod_all + w * CH;
    const int G = gridDim.x;                       // multiple of 8
    const int xcd = blockIdx.x % NXCD;
    const int wx = (blockIdx.x / NXCD) * 4 + w;    // wave index inside the XCD
    const int wpx = (G / NXCD) * 4;                // waves per XCD
    const int s0 = (int)((long long)xcd * n_wtiles / NXCD), s1 = (int)((long long)(xcd + 1) * n_wtiles / NXCD);
    for (int first = s0 + wx * CHUNK; first < s1; first += wpx * CHUNK) {
        const int last = min(first + CHUNK, s1);
        for (int wt = first; wt < last; ++wt) {
            const int row0 = wt * 64;
            const int row1 = min(row0 + 64, n_rows);
            const int r = row0 + lane;
            int rs = 0, re = 0;
            if (r < row1) { rs = rowptr[r]; re = rowptr[r + 1]; }
            const int k0 = __shfl(rs, 0, 64);
            const int k1 = __shfl(re, row1 - row0 - 1, 64);
            double acc = 0.0;
            for (int base = k0; base < k1; base += CH) {
                int c[EPL];
                double v[EPL];
#pragma unroll
                for (int j = 0; j < EPL; ++j) {
                    const int e = base + lane + 64 * j;
                    c[j] = 0; v[j] = 0.0;
                    if (e < k1) { c[j] = cols[e]; v[j] = vals[e]; }
                }
#pragma unroll
                for (int j = 0; j < EPL; ++j) {
                    const int e = base + lane + 64 * j;
                    double xv = 0.0;
                    if (e < k1) xv = x[c[j]];
                    prod[lane + 64 * j] = v[j] * xv;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                const int lo = max(rs, base), hi = min(re, base + CH);
                for (int k = lo; k < hi; ++k) acc += prod[k - base];
                asm volatile("" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
            if (r < row1) y[r] = acc;
        }
    }
}


// ---- V10: V9 with 16-bit column offsets relative to the first row of the 64-row wave tile ----
template <int EPL>
__global__ __launch_bounds__(256) void spmv_wave_rr16(int n_rows, int n_wtiles, const int *__restrict__ rowptr,
                                                      const short *__restrict__ cols16, const double *__restrict__ vals,
                                                      const double *__restrict__ x, double *__restrict__ y) {
    constexpr int CH = 64 * EPL;
    __shared__ double prod_all[4 * CH];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double *prod = prod_all + w * CH;
    const int G = gridDim.x;
    const int xcd = blockIdx.x % NXCD;
    const int wx = (blockIdx.x / NXCD) * 4 + w;
    const int wpx = (G / NXCD) * 4;
    const int s0 = (int)((long long)xcd * n_wtiles / NXCD), s1 = (int)((long long)(xcd + 1) * n_wtiles / NXCD);
    for (int wt = s0 + wx; wt < s1; wt += wpx) {
        const int row0 = wt * 64;
        const int row1 = min(row0 + 64, n_rows);
        const int r = row0 + lane;
        int rs = 0, re = 0;
        if (r < row1) { rs = rowptr[r]; re = rowptr[r + 1]; }
        const int k0 = __shfl(rs, 0, 64);
        const int k1 = __shfl(re, row1 - row0 - 1, 64);
        double acc = 0.0;
        for (int base = k0; base < k1; base += CH) {
            short c[EPL];
            double v[EPL];
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                const int e = base + lane + 64 * j;
                c[j] = 0; v[j] = 0.0;
                if (e < k1) { c[j] = cols16[e]; v[j] = vals[e]; }
            }
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                const int e = base + lane + 64 * j;
                double xv = 0.0;
                if (e < k1) xv = x[row0 + (int)c[j]];
                prod[lane + 64 * j] = v[j] * xv;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const int lo = max(rs, base), hi = min(re, base + CH);
            for (int k = lo; k < hi; ++k) acc += prod[k - base];
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        if (r < row1) y[r] = acc;
    }
}


// ---- V11: V9 software-pipelined per wave: while tile t is gathered/reduced, the stream loads of tile t+1 and
// the row pointers of tile t+2 are already in flight (issued AFTER the gathers of t so that vmcnt ordering lets
// the gathers complete first) ----
template <int EPL>
__global__ __launch_bounds__(256) void spmv_wave_pipe(int n_rows, int n_wtiles, const int *__restrict__ rowptr,
                                                      const int *__restrict__ cols, const double *__restrict__ vals,
                                                      const double *__restrict__ x, double *__restrict__ y) {
    constexpr int CH = 64 * EPL;
    __shared__ double prod_all[4 * CH];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double *prod = prod_all + w * CH;
    const int G = gridDim.x;
    const int xcd = blockIdx.x % NXCD;
    const int wx = (blockIdx.x / NXCD) * 4 + w;
    const int wpx = (G / NXCD) * 4;
    const int s0 = (int)((long long)xcd * n_wtiles / NXCD), s1 = (int)((long long)(xcd + 1) * n_wtiles / NXCD);
    int wt = s0 + wx;
    if (wt >= s1) return;
    // ---- prologue: row pointers of tile 0 and 1, stream loads of tile 0
    int rs, re, rsn = 0, ren = 0;
    {
        const int r = wt * 64 + lane;
        rs = re = 0;
        if (r < n_rows) { rs = rowptr[r]; re = rowptr[min(r + 1, n_rows)]; }
    }
    int c[EPL]; double v[EPL];
    int k0 = __shfl(rs, 0, 64);
    int k1 = __shfl(re, min(wt * 64 + 64, n_rows) - wt * 64 - 1, 64);
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
        const int e = k0 + lane + 64 * j;
        c[j] = 0; v[j] = 0.0;
        if (e < k1) { c[j] = cols[e]; v[j] = vals[e]; }
    }
    {
        const int wn = wt + wpx;
        const int r = wn * 64 + lane;
        if (wn < s1 && r < n_rows) { rsn = rowptr[r]; ren = rowptr[min(r + 1, n_rows)]; }
    }
    for (; wt < s1; wt += wpx) {
        const int row0 = wt * 64;
        const int row1 = min(row0 + 64, n_rows);
        const int r = row0 + lane;
        // 1. gathers of the current tile (first pass)
        double xv[EPL];
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const int e = k0 + lane + 64 * j;
            xv[j] = 0.0;
            if (e < k1) xv[j] = x[c[j]];
        }
        double p[EPL];
        // 2. prefetch: stream loads of the next tile, row pointers of the one after
        const int wn = wt + wpx;
        int k0n = 0, k1n = 0;
        int cn[EPL]; double vn[EPL];
        if (wn < s1) {
            k0n = __shfl(rsn, 0, 64);
            k1n = __shfl(ren, min(wn * 64 + 64, n_rows) - wn * 64 - 1, 64);
        }
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const int e = k0n + lane + 64 * j;
            cn[j] = 0; vn[j] = 0.0;
            if (e < k1n) { cn[j] = cols[e]; vn[j] = vals[e]; }
        }
        int rsnn = 0, renn = 0;
        {
            const int wnn = wn + wpx;
            const int rr = wnn * 64 + lane;
            if (wnn < s1 && rr < n_rows) { rsnn = rowptr[rr]; renn = rowptr[min(rr + 1, n_rows)]; }
        }
        // 3. products of the current tile -> LDS, reduce
#pragma unroll
        for (int j = 0; j < EPL; ++j) { p[j] = v[j] * xv[j]; prod[lane + 64 * j] = p[j]; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        double acc = 0.0;
        {
            const int lo = max(rs, k0), hi = min(re, k0 + CH);
            for (int k = lo; k < hi; ++k) acc += prod[k - k0];
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        for (int base = k0 + CH; base < k1; base += CH) {       // rare: more than CH non-zeros in the tile
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                const int e = base + lane + 64 * j;
                double t = 0.0;
                if (e < k1) t = vals[e] * x[cols[e]];
                prod[lane + 64 * j] = t;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const int lo = max(rs, base), hi = min(re, base + CH);
            for (int k = lo; k < hi; ++k) acc += prod[k - base];
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        if (r < row1) y[r] = acc;
        // 4. rotate
#pragma unroll
        for (int j = 0; j < EPL; ++j) { c[j] = cn[j]; v[j] = vn[j]; }
        rs = rsn; re = ren; rsn = rsnn; ren = renn;
        k0 = k0n; k1 = k1n;
    }
}

// ---- V2: as V0 (TILE 2048, ROWS 256) with the next tile's stream loads issued before the reduce phase
__global__ __launch_bounds__(256) void spmv_pipe(int n_rows, int n_tiles, const int *__restrict__ rowptr,
                                                 const int *__restrict__ cols, const double *__restrict__ vals,
                                                 const double *__restrict__ x, double *__restrict__ y) {
    constexpr int TILE = 2048, ROWS = 256;
    __shared__ __attribute__((aligned(16))) double prod[TILE];
    const int tid = threadIdx.x;
    const int G = gridDim.x;
    const int vb = vblock(G);
    const int t0 = (int)((long long)vb * n_tiles / G), t1 = (int)((long long)(vb + 1) * n_tiles / G);
    if (t0 >= t1) return;
    int4 c[2]; double2 v[4];
    int k0 = rowptr[t0 * ROWS];
    {
        const int base = k0 & ~3;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int e = base + 4 * (tid + 256 * j);
            c[j] = *reinterpret_cast<const int4 *>(cols + e);
            v[2 * j] = *reinterpret_cast<const double2 *>(vals + e);
            v[2 * j + 1] = *reinterpret_cast<const double2 *>(vals + e + 2);
        }
    }
    for (int tile = t0; tile < t1; ++tile) {
        const int row0 = tile * ROWS;
        const int row1 = min(row0 + ROWS, n_rows);
        const int k1 = rowptr[row1];
        const int r = row0 + tid;
        int rs = 0, re = 0;
        if (r < row1) { rs = rowptr[r]; re = rowptr[r + 1]; }
        double acc = 0.0;
        int base = k0 & ~3;
        // first pass: registers were loaded one tile ahead
        {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int l = 4 * (tid + 256 * j);
                double2 p01, p23;
                p01.x = v[2 * j].x * x[c[j].x]; p01.y = v[2 * j].y * x[c[j].y];
                p23.x = v[2 * j + 1].x * x[c[j].z]; p23.y = v[2 * j + 1].y * x[c[j].w];
                *reinterpret_cast<double2 *>(prod + l) = p01;
                *reinterpret_cast<double2 *>(prod + l + 2) = p23;
            }
            // prefetch the first pass of the next tile (its range starts at k1)
            if (tile + 1 < t1) {
                const int nb = k1 & ~3;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int e = nb + 4 * (tid + 256 * j);
                    c[j] = *reinterpret_cast<const int4 *>(cols + e);
                    v[2 * j] = *reinterpret_cast<const double2 *>(vals + e);
                    v[2 * j + 1] = *reinterpret_cast<const double2 *>(vals + e + 2);
                }
            }
            __syncthreads();
            const int lo = max(rs, base), hi = min(re, base + TILE);
            for (int k = lo; k < hi; ++k) acc += prod[k - base];
            __syncthreads();
        }
        for (base += TILE; base < k1; base += TILE) {   // rare: tile longer than one pass
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int l = 4 * (tid + 256 * j);
                const int e = base + l;
                const int4 cc = *reinterpret_cast<const int4 *>(cols + e);
                const double2 a = *reinterpret_cast<const double2 *>(vals + e);
                const double2 b = *reinterpret_cast<const double2 *>(vals + e + 2);
                double2 p01, p23;
                p01.x = a.x * x[cc.x]; p01.y = a.y * x[cc.y];
                p23.x = b.x * x[cc.z]; p23.y = b.y * x[cc.w];
                *reinterpret_cast<double2 *>(prod + l) = p01;
                *reinterpret_cast<double2 *>(prod + l + 2) = p23;
            }
            __syncthreads();
            const int lo = max(rs, base), hi = min(re, base + TILE);
            for (int k = lo; k < hi; ++k) acc += prod[k - base];
            __syncthreads();
        }
        if (r < row1) y[r] = acc;
        k0 = k1;
    }
}

// ---- V3: no LDS, LPR lanes cooperate on a row, shuffle reduce (CSR-vector with sub-waves) ----
template <int LPR>
__global__ __launch_bounds__(256) void spmv_subwave(int n_rows, const int *__restrict__ rowptr,
                                                    const int *__restrict__ cols, const double *__restrict__ vals,
                                                    const double *__restrict__ x, double *__restrict__ y) {
    const long long gt = (long long)blockIdx.x * 256 + threadIdx.x;
    const int sub = threadIdx.x & (LPR - 1);
    const long long nsub = (long long)gridDim.x * 256 / LPR;
    for (long long r = gt / LPR; r < n_rows; r += nsub) {
        const int rs = rowptr[r], re = rowptr[r + 1];
        double acc = 0.0;
        for (int k = rs + sub; k < re; k += LPR) acc += vals[k] * x[cols[k]];
#pragma unroll
        for (int off = LPR / 2; off > 0; off >>= 1) acc += __shfl_down(acc, off, LPR);
        if (sub == 0) y[r] = acc;
    }
}

// ---- V5: plain streaming read of the same bytes (upper bound for this access mix) ----
__global__ __launch_bounds__(256) void stream_ref(long long nnz4, const int4 *__restrict__ cols, const double2 *__restrict__ vals,
                                                  double *__restrict__ y) {
    double s = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nnz4; i += (long long)gridDim.x * 256) {
        const int4 c = cols[i];
        const double2 a = vals[2 * i], b = vals[2 * i + 1];
        s += a.x + a.y + b.x + b.y + (double)(c.x ^ c.y ^ c.z ^ c.w);
    }
    if (s == 12345.678) y[0] = s;
}

// ---- V12: wave-private tiles, cols/vals parked in LDS, then ONE LANE PER ROW gathers x: adjacent lanes hold
// adjacent rows, whose k-th neighbours are adjacent entries of x, so a gather instruction touches a few
// contiguous runs instead of 64 scattered addresses ----
template <int EPL, int U>
__global__ __launch_bounds__(256) void spmv_wave_rg(int n_rows, int n_wtiles, const int *__restrict__ rowptr,
                                                    const int *__restrict__ cols, const double *__restrict__ vals,
                                                    const double *__restrict__ x, double *__restrict__ y) {
    constexpr int CH = 64 * EPL;
    __shared__ double vs_all[4 * CH];
    __shared__ int cs_all[4 * CH];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double *vs = vs_all + w * CH;
    int *cs = cs_all + w * CH;
    const int G = gridDim.x;
    const int xcd = blockIdx.x % NXCD;
    const int wx = (blockIdx.x / NXCD) * 4 + w;
    const int wpx = (G / NXCD) * 4;
    const int s0 = (int)((long long)xcd * n_wtiles / NXCD), s1 = (int)((long long)(xcd + 1) * n_wtiles / NXCD);
    for (int wt = s0 + wx; wt < s1; wt += wpx) {
        const int row0 = wt * 64;
        const int row1 = min(row0 + 64, n_rows);
        const int r = row0 + lane;
        int rs = 0, re = 0;
        if (r < row1) { rs = rowptr[r]; re = rowptr[r + 1]; }
        const int k0 = __shfl(rs, 0, 64);
        const int k1 = __shfl(re, row1 - row0 - 1, 64);
        double acc = 0.0;
        for (int base = k0; base < k1; base += CH) {
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                const int e = base + lane + 64 * j;
                if (e < k1) { cs[lane + 64 * j] = cols[e]; vs[lane + 64 * j] = vals[e]; }
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const int lo = max(rs, base), hi = min(re, base + CH);
            for (int k = lo; k < hi; k += U) {
                int c[U];
                double v[U], xv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    c[u] = 0; v[u] = 0.0;
                    if (k + u < hi) { c[u] = cs[k + u - base]; v[u] = vs[k + u - base]; }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) { xv[u] = 0.0; if (k + u < hi) xv[u] = x[c[u]]; }
#pragma unroll
                for (int u = 0; u < U; ++u) if (k + u < hi) acc += v[u] * xv[u];
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        if (r < row1) y[r] = acc;
    }
}

// ---- V13: x windows staged in LDS.  Per 64-row tile a descriptor lists up to 3 contiguous runs of x that cover
// every column of the tile; the runs are loaded with wide coalesced loads into the wave's LDS slice and the
// per-non-zero column becomes a 16-bit index into that slice (2 B instead of 4 B per non-zero, and no scattered
// global gathers at all) ----
constexpr int XW = 72;   // entries reserved per run
template <int EPL, int PHASE = 0>
__global__ __launch_bounds__(256) void spmv_wave_xw(int n_rows, int n_wtiles, const int *__restrict__ rowptr,
                                                    const unsigned short *__restrict__ lidx, const double *__restrict__ vals,
                                                    const int4 *__restrict__ desc, const double *__restrict__ x,
                                                    double *__restrict__ y) {
    constexpr int CH = 64 * EPL;
    __shared__ double prod_all[4 * CH];
    __shared__ double xs_all[4 * 3 * XW];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double *prod = prod_all + w * CH;
    double *xs = xs_all + w * 3 * XW;
    const int G = gridDim.x;
    const int xcd = blockIdx.x % NXCD;
    const int wx = (blockIdx.x / NXCD) * 4 + w;
    const int wpx = (G / NXCD) * 4;
    const int s0 = (int)((long long)xcd * n_wtiles / NXCD), s1 = (int)((long long)(xcd + 1) * n_wtiles / NXCD);
    for (int wt = s0 + wx; wt < s1; wt += wpx) {
        const int row0 = wt * 64;
        const int row1 = min(row0 + 64, n_rows);
        const int r = row0 + lane;
        int rs = 0, re = 0;
        if (r < row1) { rs = rowptr[r]; re = rowptr[r + 1]; }
        const int4 d = desc[wt];                   // three run starts (clipped to [0, n_rows)), .w unused
        const int k0 = __shfl(rs, 0, 64);
        const int k1 = __shfl(re, row1 - row0 - 1, 64);
        // stage the runs: 72 entries each, two loads per run (64 + 8)
        if (PHASE != 2) {
            const int st[3] = {d.x, d.y, d.z};
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int a = st[q] + lane;
                xs[q * XW + lane] = (a < n_rows) ? x[a] : 0.0;
                if (lane < XW - 64) {
                    const int b = st[q] + 64 + lane;
                    xs[q * XW + 64 + lane] = (b < n_rows) ? x[b] : 0.0;
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        double acc = 0.0;
        for (int base = k0; base < k1; base += CH) {
            int c[EPL];
            double v[EPL];
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                const int e = base + lane + 64 * j;
                c[j] = 0; v[j] = 0.0;
                if (e < k1) { c[j] = lidx[e]; v[j] = vals[e]; }
            }
            if (PHASE == 1 || PHASE == 3) {
#pragma unroll
                for (int j = 0; j < EPL; ++j) acc += v[j] * (PHASE == 3 ? (double)c[j] : xs[c[j]]);
                continue;
            }
#pragma unroll
            for (int j = 0; j < EPL; ++j) prod[lane + 64 * j] = v[j] * xs[c[j]];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const int lo = max(rs, base), hi = min(re, base + CH);
            for (int k = lo; k < hi; ++k) acc += prod[k - base];
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        if (r < row1) y[r] = acc;
    }
}

// ---- V14: V13 with all global loads of a tile issued back to back (matrix stream first, then the x runs) ----
template <int EPL>
__global__ __launch_bounds__(256) void spmv_wave_xw2(int n_rows, int n_wtiles, const int *__restrict__ rowptr,
                                                     const unsigned short *__restrict__ lidx, const double *__restrict__ vals,
                                                     const int4 *__restrict__ desc, const double *__restrict__ x,
                                                     double *__restrict__ y) {
    constexpr int CH = 64 * EPL;
    __shared__ double prod_all[4 * CH];
    __shared__ double xs_all[4 * 3 * XW];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double *prod = prod_all + w * CH;
    double *xs = xs_all + w * 3 * XW;
    const int G = gridDim.x;
    const int xcd = blockIdx.x % NXCD;
    const int wx = (blockIdx.x / NXCD) * 4 + w;
    const int wpx = (G / NXCD) * 4;
    const int s0 = (int)((long long)xcd * n_wtiles / NXCD), s1 = (int)((long long)(xcd + 1) * n_wtiles / NXCD);
    for (int wt = s0 + wx; wt < s1; wt += wpx) {
        const int row0 = wt * 64;
        const int row1 = min(row0 + 64, n_rows);
        const int r = row0 + lane;
        int rs = 0, re = 0;
        if (r < row1) { rs = rowptr[r]; re = rowptr[r + 1]; }
        const int4 d = desc[wt];
        const int k0 = __shfl(rs, 0, 64);
        const int k1 = __shfl(re, row1 - row0 - 1, 64);
        int c[EPL];
        double v[EPL];
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const int e = k0 + lane + 64 * j;
            c[j] = 0; v[j] = 0.0;
            if (e < k1) { c[j] = lidx[e]; v[j] = vals[e]; }
        }
        {
            const int st[3] = {d.x, d.y, d.z};
            double xa[3], xb[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int a = st[q] + lane, b = st[q] + 64 + lane;
                xa[q] = (a < n_rows) ? x[a] : 0.0;
                xb[q] = (lane < XW - 64 && b < n_rows) ? x[b] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                xs[q * XW + lane] = xa[q];
                if (lane < XW - 64) xs[q * XW + 64 + lane] = xb[q];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        double acc = 0.0;
        for (int base = k0; base < k1; base += CH) {
            if (base != k0) {
#pragma unroll
                for (int j = 0; j < EPL; ++j) {
                    const int e = base + lane + 64 * j;
                    c[j] = 0; v[j] = 0.0;
                    if (e < k1) { c[j] = lidx[e]; v[j] = vals[e]; }
                }
            }
#pragma unroll
            for (int j = 0; j < EPL; ++j) prod[lane + 64 * j] = v[j] * xs[c[j]];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const int lo = max(rs, base), hi = min(re, base + CH);
            for (int k = lo; k < hi; ++k) acc += prod[k - base];
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        if (r < row1) y[r] = acc;
    }
}

// ---- V15: V9 with 16-byte loads of the matrix stream: a lane takes PAIRS of consecutive non-zeros ----
template <int EPL>   // EPL pairs per lane per pass
__global__ __launch_bounds__(256) void spmv_wave_rr_v2(int n_rows, int n_wtiles, const int *__restrict__ rowptr,
                                                       const int *__restrict__ cols, const double *__restrict__ vals,
                                                       const double *__restrict__ x, double *__restrict__ y) {
    constexpr int CH = 128 * EPL;
    __shared__ double prod_all[4 * CH];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double *prod = prod_all + w * CH;
    const int G = gridDim.x;
    const int xcd = blockIdx.x % NXCD;
    const int wx = (blockIdx.x / NXCD) * 4 + w;
    const int wpx = (G / NXCD) * 4;
    const int s0 = (int)((long long)xcd * n_wtiles / NXCD), s1 = (int)((long long)(xcd + 1) * n_wtiles / NXCD);
    for (int wt = s0 + wx; wt < s1; wt += wpx) {
        const int row0 = wt * 64;
        const int row1 = min(row0 + 64, n_rows);
        const int r = row0 + lane;
        int rs = 0, re = 0;
        if (r < row1) { rs = rowptr[r]; re = rowptr[r + 1]; }
        const int k0 = __shfl(rs, 0, 64);
        const int k1 = __shfl(re, row1 - row0 - 1, 64);
        double acc = 0.0;
        for (int base = k0 & ~1; base < k1; base += CH) {
            int2 c[EPL];
            double2 v[EPL];
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                const int e = base + 2 * lane + 128 * j;
                c[j] = make_int2(0, 0); v[j] = make_double2(0.0, 0.0);
                if (e < k1) {   // the arrays are padded: reading one element past k1 is safe
                    c[j] = *reinterpret_cast<const int2 *>(cols + e);
                    v[j] = *reinterpret_cast<const double2 *>(vals + e);
                }
            }
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                const int e = base + 2 * lane + 128 * j;
                double x0 = 0.0, x1 = 0.0;
                if (e < k1 && e >= k0) x0 = x[c[j].x];
                if (e + 1 < k1) x1 = x[c[j].y];
                *reinterpret_cast<double2 *>(prod + 2 * lane + 128 * j) = make_double2(v[j].x * x0, v[j].y * x1);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const int lo = max(rs, base), hi = min(re, base + CH);
            for (int k = lo; k < hi; ++k) acc += prod[k - base];
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        if (r < row1) y[r] = acc;
    }
}

// ---- V16: V14 software-pipelined: the matrix stream and the x runs of the NEXT tile are in flight while the
// current tile goes through LDS ----
template <int EPL>
__global__ __launch_bounds__(256) void spmv_wave_xw3(int n_rows, int n_wtiles, const int *__restrict__ rowptr,
                                                     const unsigned short *__restrict__ lidx, const double *__restrict__ vals,
                                                     const int4 *__restrict__ desc, const double *__restrict__ x,
                                                     double *__restrict__ y) {
    constexpr int CH = 64 * EPL;
    __shared__ double prod_all[4 * CH];
    __shared__ double xs_all[4 * 3 * XW];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double *prod = prod_all + w * CH;
    double *xs = xs_all + w * 3 * XW;
    const int G = gridDim.x;
    const int xcd = blockIdx.x % NXCD;
    const int wx = (blockIdx.x / NXCD) * 4 + w;
    const int wpx = (G / NXCD) * 4;
    const int s0 = (int)((long long)xcd * n_wtiles / NXCD), s1 = (int)((long long)(xcd + 1) * n_wtiles / NXCD);
    int wt = s0 + wx;
    if (wt >= s1) return;
    // state of the tile whose loads are in flight
    int rs, re, k0, k1, row1;
    int c[EPL];
    double v[EPL], xa[3], xb[3];
    auto issue = [&](int t) {
        const int row0 = t * 64;
        row1 = min(row0 + 64, n_rows);
        const int r = row0 + lane;
        rs = 0; re = 0;
        if (r < row1) { rs = rowptr[r]; re = rowptr[r + 1]; }
        const int4 d = desc[t];
        k0 = __shfl(rs, 0, 64);
        k1 = __shfl(re, row1 - row0 - 1, 64);
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const int e = k0 + lane + 64 * j;
            c[j] = 0; v[j] = 0.0;
            if (e < k1) { c[j] = lidx[e]; v[j] = vals[e]; }
        }
        const int st[3] = {d.x, d.y, d.z};
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int a = st[q] + lane, b = st[q] + 64 + lane;
            xa[q] = (a < n_rows) ? x[a] : 0.0;
            xb[q] = (lane < XW - 64 && b < n_rows) ? x[b] : 0.0;
        }
    };
    issue(wt);
    while (wt < s1) {
        // take over the in-flight tile
        const int crs = rs, cre = re, ck0 = k0, ck1 = k1, crow1 = row1, cwt = wt;
        int cc[EPL];
        double cv[EPL], cxa[3], cxb[3];
#pragma unroll
        for (int j = 0; j < EPL; ++j) { cc[j] = c[j]; cv[j] = v[j]; }
#pragma unroll
        for (int q = 0; q < 3; ++q) { cxa[q] = xa[q]; cxb[q] = xb[q]; }
        wt += wpx;
        if (wt < s1) issue(wt);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            xs[q * XW + lane] = cxa[q];
            if (lane < XW - 64) xs[q * XW + 64 + lane] = cxb[q];
        }
        __builtin_amdgcn_wave_barrier();
        double acc = 0.0;
        for (int base = ck0; base < ck1; base += CH) {
            if (base != ck0) {
#pragma unroll
                for (int j = 0; j < EPL; ++j) {
                    const int e = base + lane + 64 * j;
                    cc[j] = 0; cv[j] = 0.0;
                    if (e < ck1) { cc[j] = lidx[e]; cv[j] = vals[e]; }
                }
            }
#pragma unroll
            for (int j = 0; j < EPL; ++j) prod[lane + 64 * j] = cv[j] * xs[cc[j]];
            __builtin_amdgcn_wave_barrier();
            const int lo = max(crs, base), hi = min(cre, base + CH);
            for (int k = lo; k < hi; ++k) acc += prod[k - base];
            __builtin_amdgcn_wave_barrier();
        }
        const int r = cwt * 64 + lane;
        if (r < crow1) y[r] = acc;
    }
}

// ---- V17: V9 in single precision (values, x, y, products): what a mixed-precision V-cycle would run ----
template <int EPL, typename VT, typename XT>
__global__ __launch_bounds__(256) void spmv_wave_rr_mp(int n_rows, int n_wtiles, const int *__restrict__ rowptr,
                                                       const int *__restrict__ cols, const VT *__restrict__ vals,
                                                       const XT *__restrict__ x, XT *__restrict__ y) {
    constexpr int CH = 64 * EPL;
    __shared__ XT prod_all[4 * CH];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    XT *prod = prod_all + w * CH;
    const int G = gridDim.x;
    const int xcd = blockIdx.x % NXCD;
    const int wx = (blockIdx.x / NXCD) * 4 + w;
    const int wpx = (G / NXCD) * 4;
    const int s0 = (int)((long long)xcd * n_wtiles / NXCD), s1 = (int)((long long)(xcd + 1) * n_wtiles / NXCD);
    for (int wt = s0 + wx; wt < s1; wt += wpx) {
        const int row0 = wt * 64;
        const int row1 = min(row0 + 64, n_rows);
        const int r = row0 + lane;
        int rs = 0, re = 0;
        if (r < row1) { rs = rowptr[r]; re = rowptr[r + 1]; }
        const int k0 = __shfl(rs, 0, 64);
        const int k1 = __shfl(re, row1 - row0 - 1, 64);
        XT acc = 0;
        for (int base = k0; base < k1; base += CH) {
            int c[EPL];
            VT v[EPL];
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                const int e = base + lane + 64 * j;
                c[j] = 0; v[j] = 0;
                if (e < k1) { c[j] = cols[e]; v[j] = vals[e]; }
            }
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                const int e = base + lane + 64 * j;
                XT xv = 0;
                if (e < k1) xv = x[c[j]];
                prod[lane + 64 * j] = (XT)v[j] * xv;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const int lo = max(rs, base), hi = min(re, base + CH);
            for (int k = lo; k < hi; ++k) acc += prod[k - base];
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        if (r < row1) y[r] = acc;
    }
}

// ---- V18: x windows + 16-byte matrix loads: a lane takes PAIRS of consecutive non-zeros (double2 values, two
// 16-bit window indices in one 4-byte load) ----
template <int EPL>   // pairs per lane per pass
__global__ __launch_bounds__(256) void spmv_wave_xw_pairs(int n_rows, int n_wtiles, const int *__restrict__ rowptr,
                                                          const unsigned short *__restrict__ lidx, const double *__restrict__ vals,
                                                          const int4 *__restrict__ desc, const double *__restrict__ x,
                                                          double *__restrict__ y) {
    constexpr int CH = 128 * EPL;
    __shared__ double prod_all[4 * CH];
    __shared__ double xs_all[4 * 3 * XW];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double *prod = prod_all + w * CH;
    double *xs = xs_all + w * 3 * XW;
    const int G = gridDim.x;
    const int xcd = blockIdx.x % NXCD;
    const int wx = (blockIdx.x / NXCD) * 4 + w;
    const int wpx = (G / NXCD) * 4;
    const int s0 = (int)((long long)xcd * n_wtiles / NXCD), s1 = (int)((long long)(xcd + 1) * n_wtiles / NXCD);
    for (int wt = s0 + wx; wt < s1; wt += wpx) {
        const int row0 = wt * 64;
        const int row1 = min(row0 + 64, n_rows);
        const int r = row0 + lane;
        int rs = 0, re = 0;
        if (r < row1) { rs = rowptr[r]; re = rowptr[r + 1]; }
        const int4 d = desc[wt];
        const int k0 = __shfl(rs, 0, 64);
        const int k1 = __shfl(re, row1 - row0 - 1, 64);
        const int base0 = k0 & ~1;
        unsigned int c[EPL];
        double2 v[EPL];
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const int e = base0 + 2 * lane + 128 * j;
            c[j] = 0u; v[j] = make_double2(0.0, 0.0);
            if (e < k1) {
                c[j] = *reinterpret_cast<const unsigned int *>(lidx + e);
                v[j] = *reinterpret_cast<const double2 *>(vals + e);
            }
        }
        {
            const int st[3] = {d.x, d.y, d.z};
            double xa[3], xb[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int a = st[q] + lane, b = st[q] + 64 + lane;
                xa[q] = (a < n_rows) ? x[a] : 0.0;
                xb[q] = (lane < XW - 64 && b < n_rows) ? x[b] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                xs[q * XW + lane] = xa[q];
                if (lane < XW - 64) xs[q * XW + 64 + lane] = xb[q];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        double acc = 0.0;
        for (int base = base0; base < k1; base += CH) {
            if (base != base0) {
#pragma unroll
                for (int j = 0; j < EPL; ++j) {
                    const int e = base + 2 * lane + 128 * j;
                    c[j] = 0u; v[j] = make_double2(0.0, 0.0);
                    if (e < k1) {
                        c[j] = *reinterpret_cast<const unsigned int *>(lidx + e);
                        v[j] = *reinterpret_cast<const double2 *>(vals + e);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                const double p0 = v[j].x * xs[c[j] & 0xffffu], p1 = v[j].y * xs[c[j] >> 16];
                *reinterpret_cast<double2 *>(prod + 2 * lane + 128 * j) = make_double2(p0, p1);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const int lo = max(rs, base), hi = min(re, base + CH);
            for (int k = lo; k < hi; ++k) acc += prod[k - base];
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        if (r < row1) y[r] = acc;
    }
}

struct Mat { int n; long long nnz; int *rowptr, *cols; double *vals; short *cols16; unsigned short *lidx; int4 *desc; };

static Mat build(int layers, int nx, int ny) {
    const long long n = (long long)layers * nx * ny;
    std::vector<int> rp(n + 1); std::vector<int> cl; std::vector<double> vl;
    cl.reserve(n * 7); vl.reserve(n * 7);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (double)(s >> 8) / (1 << 24); };
    for (int l = 0; l < layers; ++l)
        for (int iy = 0; iy < ny; ++iy)
            for (int ix = 0; ix < nx; ++ix) {
                const long long i = ((long long)l * ny + iy) * nx + ix;
                rp[i] = (int)cl.size();
                const int dx[7] = {-1, 0, -1, 0, 1, 0, 1}, dy[7] = {-1, -1, 0, 0, 0, 1, 1};
                for (int k = 0; k < 7; ++k) {
                    const int jx = ix + dx[k], jy = iy + dy[k];
                    if (jx < 0 || jy < 0 || jx >= nx || jy >= ny) continue;
                    cl.push_back((int)(((long long)l * ny + jy) * nx + jx));
                    vl.push_back(k == 3 ? 6.0 + rnd() : -rnd());
                }
            }
    rp[n] = (int)cl.size();
    Mat m; m.n = (int)n; m.nnz = (long long)cl.size();
    const size_t pad = 8192;
    CK(hipMalloc(&m.rowptr, sizeof(int) * (n + 1)));
    CK(hipMalloc(&m.cols, sizeof(int) * (cl.size() + pad)));
    CK(hipMalloc(&m.vals, sizeof(double) * (cl.size() + pad)));
    CK(hipMemset(m.cols, 0, sizeof(int) * (cl.size() + pad)));
    CK(hipMemset(m.vals, 0, sizeof(double) * (cl.size() + pad)));
    CK(hipMemcpy(m.rowptr, rp.data(), sizeof(int) * (n + 1), hipMemcpyHostToDevice));
    CK(hipMemcpy(m.cols, cl.data(), sizeof(int) * cl.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(m.vals, vl.data(), sizeof(double) * vl.size(), hipMemcpyHostToDevice));
    std::vector<short> c16(cl.size() + pad, 0);
    for (long long i = 0; i < n; ++i)
        for (int k = rp[i]; k < rp[i + 1]; ++k) c16[k] = (short)(cl[k] - (int)((i / 64) * 64));
    CK(hipMalloc(&m.cols16, sizeof(short) * c16.size()));
    CK(hipMemcpy(m.cols16, c16.data(), sizeof(short) * c16.size(), hipMemcpyHostToDevice));
    // x-window descriptors: runs start at row0 - nx - 1, row0 - 1, row0 + nx - 1 (72 entries each)
    const long long nt = (n + 63) / 64;
    std::vector<int4> ds(nt);
    std::vector<unsigned short> li(cl.size() + pad, 0);
    long long bad = 0;
    for (long long t = 0; t < nt; ++t) {
        const long long row0 = t * 64;
        long long st[3] = {row0 - nx - 1, row0 - 1, row0 + nx - 1};
        for (int q = 0; q < 3; ++q) st[q] = st[q] < 0 ? 0 : st[q];
        ds[t] = make_int4((int)st[0], (int)st[1], (int)st[2], 0);
        for (long long i = row0; i < row0 + 64 && i < n; ++i)
            for (int k = rp[i]; k < rp[i + 1]; ++k) {
                int found = -1;
                for (int q = 0; q < 3; ++q)
                    if (cl[k] >= st[q] && cl[k] < st[q] + 72) { found = q * 72 + (int)(cl[k] - st[q]); break; }
                if (found < 0) { ++bad; found = 0; }
                li[k] = (unsigned short)found;
            }
    }
    if (bad) printf("WARNING: %lld columns outside the staged windows\n", bad);
    CK(hipMalloc(&m.lidx, sizeof(unsigned short) * li.size()));
    CK(hipMemcpy(m.lidx, li.data(), sizeof(unsigned short) * li.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&m.desc, sizeof(int4) * nt));
    CK(hipMemcpy(m.desc, ds.data(), sizeof(int4) * nt, hipMemcpyHostToDevice));
    return m;
}

static const char *g_only = nullptr;
static const char *g_name = nullptr;
template <typename F> static double timeit(F launch, int reps = 30) {
    if (g_only && !strstr(g_name, g_only)) return -1.0;
    if (g_only) reps = 20;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 5; ++i) launch();
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return ms * 1e-3 / reps;
}

int main(int argc, char **argv) {
    const int layers = argc > 1 ? atoi(argv[1]) : 8, nx = argc > 2 ? atoi(argv[2]) : 1118;
    const char *only = argc > 3 ? argv[3] : nullptr;
    Mat m = build(layers, nx, nx);
    double *x, *y, *yref;
    CK(hipMalloc(&x, sizeof(double) * m.n)); CK(hipMalloc(&y, sizeof(double) * m.n)); CK(hipMalloc(&yref, sizeof(double) * m.n));
    std::vector<double> hx(m.n);
    for (int i = 0; i < m.n; ++i) hx[i] = sin(0.001 * i) + 0.5;
    CK(hipMemcpy(x, hx.data(), sizeof(double) * m.n, hipMemcpyHostToDevice));
    const double bytes = 12.0 * m.nnz + 20.0 * m.n + 4;
    printf("n=%d nnz=%lld bytes=%.0f\n", m.n, m.nnz, bytes);
    auto report = [&](const char *name, double t, bool check) {
        if (t < 0) return;
        double err = -1;
        if (check) {
            std::vector<double> a(m.n), b(m.n);
            CK(hipMemcpy(a.data(), y, sizeof(double) * m.n, hipMemcpyDeviceToHost));
            CK(hipMemcpy(b.data(), yref, sizeof(double) * m.n, hipMemcpyDeviceToHost));
            err = 0;
            for (int i = 0; i < m.n; ++i) err = fmax(err, fabs(a[i] - b[i]));
        }
        printf("%-34s %8.1f us  %7.1f GB/s  %5.1f%%  maxdiff=%g\n", name, t * 1e6, bytes / t / 1e9, bytes / t / 8e10, err);
        fflush(stdout);
    };
    g_only = only;
#define RUN(NAME, ...) do { g_name = NAME; report(NAME, timeit(__VA_ARGS__), true); } while (0)
    const int nt256 = (m.n + 255) / 256, nt512 = (m.n + 511) / 512;
    auto G = [](int nt, int cap) { int g = nt < cap ? nt : cap; if (g >= 8) g -= g % 8; return g < 1 ? 1 : g; };
    // reference result
    spmv_lds<2048, 256, false><<<G(nt256, 2048), 256>>>(m.n, nt256, m.rowptr, m.cols, m.vals, x, yref);
    CK(hipDeviceSynchronize());
    for (int cap : {1024, 2048, 4096, 8192}) {
        char nm[64];
        snprintf(nm, 64, "lds 2048/256 grid<=%d", cap);
        RUN(nm, [&] { spmv_lds<2048, 256, false><<<G(nt256, cap), 256>>>(m.n, nt256, m.rowptr, m.cols, m.vals, x, y); });
    }
    RUN("lanecons 2048/256", [&] { spmv_lds_lc<2048, 256, 0><<<G(nt256, 2048), 256>>>(m.n, nt256, m.rowptr, m.cols, m.vals, x, y); });
    RUN("lanecons 2048/256 grid 8192", [&] { spmv_lds_lc<2048, 256, 0><<<G(nt256, 8192), 256>>>(m.n, nt256, m.rowptr, m.cols, m.vals, x, y); });
    RUN("lanecons 4096/512", [&] { spmv_lds_lc<4096, 512, 0><<<G(nt512, 2048), 256>>>(m.n, nt512, m.rowptr, m.cols, m.vals, x, y); });
    RUN("lanecons 4096/512 grid 8192", [&] { spmv_lds_lc<4096, 512, 0><<<G(nt512, 8192), 256>>>(m.n, nt512, m.rowptr, m.cols, m.vals, x, y); });
    RUN("lds 2048/256 predicated", [&] { spmv_lds<2048, 256, false, true><<<G(nt256, 2048), 256>>>(m.n, nt256, m.rowptr, m.cols, m.vals, x, y); });
    RUN("lds 2048/256 predicated grid 8192", [&] { spmv_lds<2048, 256, false, true><<<G(nt256, 8192), 256>>>(m.n, nt256, m.rowptr, m.cols, m.vals, x, y); });
    { const int nwt = (m.n + 63) / 64;
      RUN("wave-private epl8 grid 2048", [&] { spmv_wave<8><<<2048, 256>>>(m.n, nwt, m.rowptr, m.cols, m.vals, x, y); });
      RUN("wave-private epl8 grid 8192", [&] { spmv_wave<8><<<8192, 256>>>(m.n, nwt, m.rowptr, m.cols, m.vals, x, y); });
      RUN("wave-private epl8 grid 16384", [&] { spmv_wave<8><<<16384, 256>>>(m.n, nwt, m.rowptr, m.cols, m.vals, x, y); }); }
    { const int nwt = (m.n + 63) / 64;
      RUN("wave-rr chunk1 grid 2048", [&] { spmv_wave_rr<8, 1><<<2048, 256>>>(m.n, nwt, m.rowptr, m.cols, m.vals, x, y); });
      RUN("wave-rr chunk2 grid 2048", [&] { spmv_wave_rr<8, 2><<<2048, 256>>>(m.n, nwt, m.rowptr, m.cols, m.vals, x, y); });
      RUN("wave-rr chunk4 grid 2048", [&] { spmv_wave_rr<8, 4><<<2048, 256>>>(m.n, nwt, m.rowptr, m.cols, m.vals, x, y); });
      RUN("wave-rr chunk8 grid 2048", [&] { spmv_wave_rr<8, 8><<<2048, 256>>>(m.n, nwt, m.rowptr, m.cols, m.vals, x, y); });
      RUN("wave-rr chunk4 grid 1024", [&] { spmv_wave_rr<8, 4><<<1024, 256>>>(m.n, nwt, m.rowptr, m.cols, m.vals, x, y); });
      RUN("wave-rr chunk4 grid 1536", [&] { spmv_wave_rr<8, 4><<<1536, 256>>>(m.n, nwt, m.rowptr, m.cols, m.vals, x, y); }); }
    { const int nwt = (m.n + 63) / 64;
      // single-precision copies (contents irrelevant for the timing: reinterpret the double arrays)
      const float *vf = (const float *)m.vals; const float *xf = (const float *)x; float *yf = (float *)y;
      g_name = "mixed f32 vals+x grid 2048"; report(g_name, timeit([&] { spmv_wave_rr_mp<8, float, float><<<2048, 256>>>(m.n, nwt, m.rowptr, m.cols, vf, xf, yf); }), false);
      g_name = "mixed f32 vals+x grid 4096"; report(g_name, timeit([&] { spmv_wave_rr_mp<8, float, float><<<4096, 256>>>(m.n, nwt, m.rowptr, m.cols, vf, xf, yf); }), false);
      g_name = "mixed f32 vals, f64 x grid 2048"; report(g_name, timeit([&] { spmv_wave_rr_mp<8, float, double><<<2048, 256>>>(m.n, nwt, m.rowptr, m.cols, vf, x, y); }), false);
      g_name = "mixed f64 vals, f64 x (check) grid 2048"; report(g_name, timeit([&] { spmv_wave_rr_mp<8, double, double><<<2048, 256>>>(m.n, nwt, m.rowptr, m.cols, m.vals, x, y); }), true); }
    { const int nwt = (m.n + 63) / 64;
      RUN("pairs epl4 grid 2048", [&] { spmv_wave_rr_v2<4><<<2048, 256>>>(m.n, nwt, m.rowptr, m.cols, m.vals, x, y); });
      RUN("pairs epl4 grid 1536", [&] { spmv_wave_rr_v2<4><<<1536, 256>>>(m.n, nwt, m.rowptr, m.cols, m.vals, x, y); });
      RUN("pairs epl2 grid 2048", [&] { spmv_wave_rr_v2<2><<<2048, 256>>>(m.n, nwt, m.rowptr, m.cols, m.vals, x, y); });
      RUN("pairs epl2 grid 4096", [&] { spmv_wave_rr_v2<2><<<4096, 256>>>(m.n, nwt, m.rowptr, m.cols, m.vals, x, y); }); }
    { const int nwt = (m.n + 63) / 64;
      RUN("xwindow epl8 grid 2048", [&] { spmv_wave_xw<8><<<2048, 256>>>(m.n, nwt, m.rowptr, m.lidx, m.vals, m.desc, x, y); });
      RUN("xwindow epl8 grid 1536", [&] { spmv_wave_xw<8><<<1536, 256>>>(m.n, nwt, m.rowptr, m.lidx, m.vals, m.desc, x, y); });
      RUN("xwindow epl8 grid 4096", [&] { spmv_wave_xw<8><<<4096, 256>>>(m.n, nwt, m.rowptr, m.lidx, m.vals, m.desc, x, y); });
      RUN("xwindow epl4 grid 2048", [&] { spmv_wave_xw<4><<<2048, 256>>>(m.n, nwt, m.rowptr, m.lidx, m.vals, m.desc, x, y); });
      RUN("xwpairs epl4 grid 1536", [&] { spmv_wave_xw_pairs<4><<<1536, 256>>>(m.n, nwt, m.rowptr, m.lidx, m.vals, m.desc, x, y); });
      RUN("xwpairs epl4 grid 2048", [&] { spmv_wave_xw_pairs<4><<<2048, 256>>>(m.n, nwt, m.rowptr, m.lidx, m.vals, m.desc, x, y); });
      RUN("xwpairs epl2 grid 2048", [&] { spmv_wave_xw_pairs<2><<<2048, 256>>>(m.n, nwt, m.rowptr, m.lidx, m.vals, m.desc, x, y); });
      RUN("xwpairs epl2 grid 4096", [&] { spmv_wave_xw_pairs<2><<<4096, 256>>>(m.n, nwt, m.rowptr, m.lidx, m.vals, m.desc, x, y); });
      RUN("xwindow3 epl8 grid 1024", [&] { spmv_wave_xw3<8><<<1024, 256>>>(m.n, nwt, m.rowptr, m.lidx, m.vals, m.desc, x, y); });
      RUN("xwindow3 epl8 grid 1536", [&] { spmv_wave_xw3<8><<<1536, 256>>>(m.n, nwt, m.rowptr, m.lidx, m.vals, m.desc, x, y); });
      RUN("xwindow3 epl8 grid 2048", [&] { spmv_wave_xw3<8><<<2048, 256>>>(m.n, nwt, m.rowptr, m.lidx, m.vals, m.desc, x, y); });
      RUN("xwindow2 epl8 grid 1536", [&] { spmv_wave_xw2<8><<<1536, 256>>>(m.n, nwt, m.rowptr, m.lidx, m.vals, m.desc, x, y); });
      RUN("xwindow2 epl8 grid 2048", [&] { spmv_wave_xw2<8><<<2048, 256>>>(m.n, nwt, m.rowptr, m.lidx, m.vals, m.desc, x, y); });
      RUN("xwindow2 epl4 grid 2048", [&] { spmv_wave_xw2<4><<<2048, 256>>>(m.n, nwt, m.rowptr, m.lidx, m.vals, m.desc, x, y); });
      RUN("xwindow2 epl4 grid 4096", [&] { spmv_wave_xw2<4><<<4096, 256>>>(m.n, nwt, m.rowptr, m.lidx, m.vals, m.desc, x, y); });
      RUN("xwindow-norowsum epl4 grid 2048", [&] { spmv_wave_xw<4, 1><<<2048, 256>>>(m.n, nwt, m.rowptr, m.lidx, m.vals, m.desc, x, y); });
      RUN("xwindow-nostage epl4 grid 2048", [&] { spmv_wave_xw<4, 2><<<2048, 256>>>(m.n, nwt, m.rowptr, m.lidx, m.vals, m.desc, x, y); });
      RUN("xwindow-streamonly epl4 grid 2048", [&] { spmv_wave_xw<4, 3><<<2048, 256>>>(m.n, nwt, m.rowptr, m.lidx, m.vals, m.desc, x, y); });
      RUN("xwindow-streamonly epl8 grid 2048", [&] { spmv_wave_xw<8, 3><<<2048, 256>>>(m.n, nwt, m.rowptr, m.lidx, m.vals, m.desc, x, y); });
      RUN("xwindow epl4 grid 4096", [&] { spmv_wave_xw<4><<<4096, 256>>>(m.n, nwt, m.rowptr, m.lidx, m.vals, m.desc, x, y); }); }
    { const int nwt = (m.n + 63) / 64;
      RUN("rowgather epl8 u8 grid 2048", [&] { spmv_wave_rg<8, 8><<<2048, 256>>>(m.n, nwt, m.rowptr, m.cols, m.vals, x, y); });
      RUN("rowgather epl8 u4 grid 2048", [&] { spmv_wave_rg<8, 4><<<2048, 256>>>(m.n, nwt, m.rowptr, m.cols, m.vals, x, y); });
      RUN("rowgather epl8 u8 grid 1536", [&] { spmv_wave_rg<8, 8><<<1536, 256>>>(m.n, nwt, m.rowptr, m.cols, m.vals, x, y); });
      RUN("rowgather epl8 u8 grid 4096", [&] { spmv_wave_rg<8, 8><<<4096, 256>>>(m.n, nwt, m.rowptr, m.cols, m.vals, x, y); });
      RUN("rowgather epl16 u8 grid 2048", [&] { spmv_wave_rg<16, 8><<<2048, 256>>>(m.n, nwt, m.rowptr, m.cols, m.vals, x, y); }); }
    { const int nwt = (m.n + 63) / 64;
      RUN("wave-rr16 chunk1 grid 2048", [&] { spmv_wave_rr16<8><<<2048, 256>>>(m.n, nwt, m.rowptr, m.cols16, m.vals, x, y); });
      RUN("wave-rr16 chunk1 grid 1536", [&] { spmv_wave_rr16<8><<<1536, 256>>>(m.n, nwt, m.rowptr, m.cols16, m.vals, x, y); });
      RUN("wave-rr16 chunk1 grid 4096", [&] { spmv_wave_rr16<8><<<4096, 256>>>(m.n, nwt, m.rowptr, m.cols16, m.vals, x, y); }); }
    { const int nwt = (m.n + 63) / 64;
      RUN("wave-pipe epl8 grid 2048", [&] { spmv_wave_pipe<8><<<2048, 256>>>(m.n, nwt, m.rowptr, m.cols, m.vals, x, y); });
      RUN("wave-pipe epl8 grid 1536", [&] { spmv_wave_pipe<8><<<1536, 256>>>(m.n, nwt, m.rowptr, m.cols, m.vals, x, y); });
      RUN("wave-pipe epl8 grid 1024", [&] { spmv_wave_pipe<8><<<1024, 256>>>(m.n, nwt, m.rowptr, m.cols, m.vals, x, y); }); }
    RUN("lds 2048/256 nontemporal", [&] { spmv_lds<2048, 256, true><<<G(nt256, 2048), 256>>>(m.n, nt256, m.rowptr, m.cols, m.vals, x, y); });
    RUN("lds 4096/512", [&] { spmv_lds<4096, 512, false><<<G(nt512, 2048), 256>>>(m.n, nt512, m.rowptr, m.cols, m.vals, x, y); });
    RUN("lds 4096/512 grid 1024", [&] { spmv_lds<4096, 512, false><<<G(nt512, 1024), 256>>>(m.n, nt512, m.rowptr, m.cols, m.vals, x, y); });
    for (int cap : {1024, 2048, 4096})  {
        char nm[64];
        snprintf(nm, 64, "pipelined 2048/256 grid<=%d", cap);
        RUN(nm, [&] { spmv_pipe<<<G(nt256, cap), 256>>>(m.n, nt256, m.rowptr, m.cols, m.vals, x, y); });
    }
    RUN("subwave 8 lanes/row", [&] { spmv_subwave<8><<<8192, 256>>>(m.n, m.rowptr, m.cols, m.vals, x, y); });
    RUN("subwave 4 lanes/row", [&] { spmv_subwave<4><<<8192, 256>>>(m.n, m.rowptr, m.cols, m.vals, x, y); });
    RUN("subwave 2 lanes/row", [&] { spmv_subwave<2><<<8192, 256>>>(m.n, m.rowptr, m.cols, m.vals, x, y); });
    RUN("subwave 1 lane/row", [&] { spmv_subwave<1><<<8192, 256>>>(m.n, m.rowptr, m.cols, m.vals, x, y); });
    g_name = "stream"; report("stream cols+vals only (bound)", timeit([&] { stream_ref<<<2048, 256>>>(m.nnz / 4, (const int4 *)m.cols, (const double2 *)m.vals, y); }), false);
    return 0;
}
