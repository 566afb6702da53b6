// CSR SpMV for gfx950 (MI355X): y = A x, i32 indices; f64 for the solver's products, f32 copies for the multigrid cycle.
//
// Bandwidth-bound (0.135 flop/B), so no MFMA: the design is about moving 12 B per non-zero +
// 20 B per row exactly once, at the rate the HBM delivers (measured with rocprofv3 FETCH_SIZE, see
// DESIGN.md "SpMV"):
//
//  * wave-private tiles: a wavefront (64 lanes) owns 64 consecutive rows.  Their non-zeros are one
//    contiguous range of `cols`/`vals`, which the wave streams lane-consecutively (lane l takes
//    elements l, l+64, ...: every load instruction covers 256/512 contiguous bytes) and predicated
//    on the end of the range, so no byte of the matrix is fetched twice.
//  * each lane gathers x[col] (adjacent lanes hold adjacent non-zeros, whose columns are mostly
//    adjacent mesh neighbours: L1 hits), multiplies, and parks the product in the wave's private
//    4 KiB slice of LDS.
//  * one lane per row then adds its row segment out of LDS in CSR order, so the result is
//    bit-identical to a sequential CSR product (scipy's csr_matvec).  LDS operations of one wave
//    execute in order: no workgroup barrier anywhere, waves never wait for each other.  Rows longer
//    than the slice simply span several passes.
//  * XCD-aware sweep: workgroups b, b+8, b+16, ... share an XCD and its private 4 MiB L2.  Each XCD
//    gets one contiguous slab of the matrix and ALL its waves sweep that slab together, one 64-row
//    tile per wave per turn.  The rows in flight on an XCD are therefore neighbours, and the x
//    entries gathered for mesh row i (columns i-nx-1 .. i+nx+1) are still in L2 when rows i+-nx use
//    them: FETCH_SIZE drops from 1.31x to 1.05x of the algorithmic bytes.
//  * optional epilogue: dot_with[row]*y[row] summed over the wave with shuffles, over the workgroup
//    through LDS, one partial per workgroup (deterministic, no float atomics).
//  * x windows (csr_build_xw_plan): what bounds the kernel after all that is the gather path (the L1/TA unit
//    spends about two accesses per gathered double while the matrix stream alone runs at 6 TB/s).  A tile whose
//    columns are covered by at most three runs of 72 (or 128) consecutive x entries -- every tile of a band matrix
//    such as a scan-line numbered mesh, most tiles of a strip-ordered unstructured mesh -- stages those runs
//    into LDS with wide coalesced loads and addresses them through a 16-bit index per non-zero (2 instead of 4 bytes of index traffic, no scattered global loads at
//    all); other tiles (rows with lumped couplings to far-away unknowns, unstructured numberings) take the
//    gather path, tile by tile, inside the same launch.  Products and their summation order are unchanged.
#include "common.hpp"

#include <algorithm>
#include <stdlib.h>

namespace padne {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
template <typename T> __device__ __forceinline__ T wave_sum_t(T v) {
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

constexpr int kXwRuns = 3;              // staged runs of x per tile
constexpr int kXwRunShort = 72;         // entries per run: 64 rows + the mesh neighbours on both sides (scan-line grids),
constexpr int kXwRunLong = 128;         // or twice the tile for strip-ordered unstructured meshes
// The wide plan (csr_build_xw_plan_wide): twelve short runs.  The rows of the fused up-leg operator W = P - c D^-1 A P are fine
// rows, its columns aggregates: 64 consecutive fine rows reach the aggregates rooted within four mesh lines of theirs, and
// with the aggregates numbered in root order every mesh line contributes ONE short run of consecutive columns -- up to
// nine or ten runs of about ten, far apart.  Three runs of 72 never cover that; twelve runs of 20 do (ten of 24 covered
// 91 % of the tiles of config C4, twelve of 20 cover 95 %, fifteen of 16 97 % at the same speed), and 240 positions
// still fit one byte: 5 instead of 8 bytes per non-zero and no scattered loads for the largest product of the cycle.
constexpr int kXwRunsWide = 12;
constexpr int kXwRunWide = 20;
constexpr int kXwDescWide = 16;         // ints per tile: twelve run starts, three spare, the flag
constexpr int kEpl = 8;                 // elements per lane per pass
constexpr int kWaveChunk = 64 * kEpl;   // non-zeros parked in LDS per wave per pass (4 KiB)

// The streaming part of a windowed tile: G consecutive non-zeros per lane and load (one 4-byte word of G positions --
// 16-bit ones in pairs, 8-bit ones in fours -- and one or two 16-byte loads of G values), products parked in the
// wave's LDS slice, one lane per row summing its segment in CSR order.  A pass starts on a multiple of G; the stray
// elements in front of k0 or behind k1 are multiplied like the others but never summed (both arrays are padded).
template <int G, typename IT, typename VT, typename XT, typename ST = XT>      // ST: the type x is stored in (float under a double product: pcg.hip)
__device__ __forceinline__ XT xw_stream_tile(const VT *__restrict__ vals, const IT *__restrict__ lidx, const ST *xs, XT *prod,
                                             const int k0, const int k1, const int rs, const int re, const int lane,
                                             const int top) {
    static_assert(G * sizeof(IT) == 4, "one index word per lane and load");
    constexpr int NJ = kEpl / G;
    struct alignas(G * sizeof(VT) < 16 ? G * sizeof(VT) : 16) VG { VT v[G]; };
    unsigned int cw[NJ];
    VG vg[NJ];
    XT acc = 0;
    for (int base = k0 & ~(G - 1);;) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int e = base + G * lane + 64 * G * j;
            cw[j] = 0u;
#pragma unroll
            for (int t = 0; t < G; ++t) vg[j].v[t] = 0;
            if (e < k1) {
                cw[j] = *reinterpret_cast<const unsigned int *>(lidx + e);
                vg[j] = *reinterpret_cast<const VG *>(vals + e);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the staged runs are in LDS
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
            for (int t = 0; t < G; ++t) {
                const int pos = min((int)((cw[j] >> (8 * sizeof(IT) * t)) & ((1u << (8 * sizeof(IT))) - 1u)), top);
                prod[G * lane + 64 * G * j + t] = (XT)vg[j].v[t] * (XT)xs[pos];
            }
        }
        // same-wave LDS traffic is processed in issue order; keep the compiler from reordering
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const int lo = max(rs, base), hi = min(re, base + kWaveChunk);
        {
            int k = lo;                                    // two entries per turn (one 8-byte LDS read where they pair up)
            for (; k + 1 < hi; k += 2) {
                const XT p0 = prod[k - base], p1 = prod[k + 1 - base];
                acc += p0;
                acc += p1;
            }
            if (k < hi) acc += prod[k - base];
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        base += kWaveChunk;
        if (base >= k1) break;
    }
    return acc;
}

// The gather path of a tile in the same shape: four consecutive non-zeros per lane and load (one 16-byte load of columns,
// one or two of values) instead of eight 4-byte loads of each, x gathered from global memory, products parked four at a time.
// A pass starts on a multiple of 4; stray elements in front of k0 / behind k1 are multiplied (their columns are valid: both
// arrays are padded with column 0 / value 0) but never summed.  Same products, same order of the row sums.
template <typename VT, typename XT, int EPL, typename ST = XT, bool PRE = false>      // PRE: x stands for scale * mul .* x
__device__ __forceinline__ XT gather_stream_tile(const int *__restrict__ cols, const VT *__restrict__ vals, const ST *__restrict__ x,
                                                 XT *prod, const int k0, const int k1, const int rs, const int re, const int lane,
                                                 const XT *__restrict__ mul = nullptr, const XT scale = 0) {
    constexpr int G = 4, NJ = EPL / G, CHUNK = 64 * EPL;
    struct alignas(16) VG { VT v[G]; };
    int4 cw[NJ];
    VG vg[NJ];
    XT acc = 0;
    for (int base = k0 & ~(G - 1);;) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int e = base + G * lane + 64 * G * j;
            cw[j] = make_int4(0, 0, 0, 0);
#pragma unroll
            for (int t = 0; t < G; ++t) vg[j].v[t] = 0;
            if (e < k1) {
                cw[j] = *reinterpret_cast<const int4 *>(cols + e);
                vg[j] = *reinterpret_cast<const VG *>(vals + e);
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int e = base + G * lane + 64 * G * j;
            XT xv[G] = {0, 0, 0, 0};
            if (e < k1) {
                xv[0] = (XT)x[cw[j].x];
                xv[1] = (XT)x[cw[j].y];
                xv[2] = (XT)x[cw[j].z];
                xv[3] = (XT)x[cw[j].w];
                if (PRE) {
                    xv[0] *= scale * mul[cw[j].x];
                    xv[1] *= scale * mul[cw[j].y];
                    xv[2] *= scale * mul[cw[j].z];
                    xv[3] *= scale * mul[cw[j].w];
                }
            }
#pragma unroll
            for (int t = 0; t < G; ++t) prod[G * lane + 64 * G * j + t] = (XT)vg[j].v[t] * xv[t];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const int lo = max(rs, base), hi = min(re, base + CHUNK);
        {
            int k = lo;                                    // two entries per turn (one 8-byte LDS read where they pair up)
            for (; k + 1 < hi; k += 2) {
                const XT p0 = prod[k - base], p1 = prod[k + 1 - base];
                acc += p0;
                acc += p1;
            }
            if (k < hi) acc += prod[k - base];
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        base += CHUNK;
        if (base >= k1) break;
    }
    return acc;
}

// The tile's three runs of x into the wave's LDS slice in 16-byte pieces: 54 lanes of ONE load and ONE store instruction
// bring the 216 single-precision entries of three runs of 72 (two rounds for doubles or runs of 128) where lane-per-entry
// loads took six of each.  A piece that would reach past the end of x is fetched entry by entry.
template <int RUN, typename XT, bool PRE = false>      // PRE: what is staged is scale * mul .* x
__device__ __forceinline__ void stage_windows(const XT *__restrict__ x, const int n_cols, const int4 d, XT *xs, const int lane,
                                              const XT *__restrict__ mul = nullptr, const XT scale = 0) {
    constexpr int PER = 16 / (int)sizeof(XT), PPR = RUN / PER, NP = kXwRuns * PPR;
    static_assert(RUN % PER == 0, "runs are whole 16-byte pieces");
    struct alignas(sizeof(XT)) PieceG { XT v[PER]; };      // in global memory a run starts at any entry
    struct alignas(16) PieceL { XT v[PER]; };
#pragma unroll
    for (int c0 = 0; c0 < NP; c0 += 64) {
        const int c = c0 + lane;
        if (c < NP) {
            const int q = c / PPR, i = c - q * PPR;
            const int g0 = (q == 0 ? d.x : (q == 1 ? d.y : d.z)) + PER * i;
            PieceL pl;
            if (g0 + PER - 1 < n_cols) {
                const PieceG pg = *reinterpret_cast<const PieceG *>(x + g0);
#pragma unroll
                for (int t = 0; t < PER; ++t) pl.v[t] = pg.v[t];
                if (PRE) {
                    const PieceG mg = *reinterpret_cast<const PieceG *>(mul + g0);
#pragma unroll
                    for (int t = 0; t < PER; ++t) pl.v[t] *= scale * mg.v[t];
                }
            } else {
#pragma unroll
                for (int t = 0; t < PER; ++t) pl.v[t] = (g0 + t < n_cols) ? (PRE ? scale * mul[g0 + t] * x[g0 + t] : x[g0 + t]) : (XT)0;
            }
            *reinterpret_cast<PieceL *>(xs + q * RUN + PER * i) = pl;
        }
    }
}

// the same for the wide plan: lane k < 10 holds the start of run k (`mine`), a piece asks its run's start by shuffle
template <typename XT>
__device__ __forceinline__ void stage_windows_wide(const XT *__restrict__ x, const int n_cols, const int mine, XT *xs, const int lane) {
    constexpr int PER = 16 / (int)sizeof(XT), PPR = kXwRunWide / PER, NP = kXwRunsWide * PPR;
    static_assert(kXwRunWide % PER == 0, "runs are whole 16-byte pieces");
    struct alignas(sizeof(XT)) PieceG { XT v[PER]; };
    struct alignas(16) PieceL { XT v[PER]; };
#pragma unroll
    for (int c0 = 0; c0 < NP; c0 += 64) {
        const int c = c0 + lane;
        const int q = min(c / PPR, kXwRunsWide - 1), i = c - q * PPR;
        const int g0 = __shfl(mine, q, 64) + PER * i;       // (every lane takes part in the shuffle)
        if (c < NP) {
            PieceL pl;
            if (g0 + PER - 1 < n_cols) {
                const PieceG pg = *reinterpret_cast<const PieceG *>(x + g0);
#pragma unroll
                for (int t = 0; t < PER; ++t) pl.v[t] = pg.v[t];
            } else {
#pragma unroll
                for (int t = 0; t < PER; ++t) pl.v[t] = (g0 + t < n_cols) ? x[g0 + t] : (XT)0;
            }
            *reinterpret_cast<PieceL *>(xs + q * kXwRunWide + PER * i) = pl;
        }
    }
}

// Epilogues (acc = (A x)[row]):
//   SPMV_PLAIN   y = acc
//   SPMV_DOT     y = acc ; partial sums of dot_with[row] * acc
//   SPMV_RESID   y = aux1[row] - acc                                   (residual b - A x)
//   SPMV_ADD     y += acc                                              (prolongation: x += P xc)
//   SPMV_JACOBI  y = x[row] + scale * aux2[row] * (aux1[row] - acc)    (damped-Jacobi sweep, aux2 = 1/diag)
//                optional partial sums of aux1[row] * y[row]           (r.z of the preconditioned CG)
//   SPMV_RESTRICT y = acc ; y2 = scale * aux2[row] * acc                 (restriction, and the first damped-Jacobi sweep of
//                the level it restricts to from a zero start: aux2 = that level's 1/diag)
//   SPMV_WUP     y = aux0[row] + scale * aux2[row] * aux1[row] + acc   (coarse correction and post-smoothing in one
//                product with W = P - c D^-1 A P, see amg.hip: aux0 = pre-smoothed iterate, aux1 = its residual)
//
// Scalar types: VT matrix values, XT the vectors x / aux1 / aux2 and the arithmetic, YT the output.  The solver's
// own products are <double, double, double>; the multigrid cycle runs <float, float, float> (single-precision
// copies of its operators: 8 instead of 12 bytes per non-zero, half the vector traffic), and its last stage
// <float, float, double> hands z back to CG in double, multiplied by sqrt(*out_scale2) (the cycle works on
// r / ||b||, see amg.hip) and with the r.z partials taken against the double residual `dot_with`.
// ST (default XT): the type the multiplied vector is STORED in.  <SPMV_DOT, double, double, double, ..., float> is the CG loop's
// q = A p with the search direction kept in single precision (pcg.hip): staged and gathered as floats, multiplied in double,
// and the p.q partials taken against x itself.
template <int MODE, typename VT, typename XT, typename YT, bool LIST = false, bool LONG = false, bool WIDE = false,
          typename ST = XT>
__global__ __launch_bounds__(kSpmvThreads) void csr_spmv_kernel(
    const int n_rows, const int n_cols, const int n_wtiles, const int *__restrict__ rowptr,
    const int *__restrict__ cols, const VT *__restrict__ vals,
    const ST *__restrict__ x, YT *__restrict__ y,
    const double *__restrict__ dot_with, double *__restrict__ partials,
    const int *__restrict__ done_flag, const XT *__restrict__ aux1,
    const XT *__restrict__ aux2, const XT scale, const double *__restrict__ out_scale2,
    const int4 *__restrict__ xw_desc, const void *__restrict__ xw_lidx, const int xw_run,
    const XT *__restrict__ aux0, XT *__restrict__ y2, const int *__restrict__ tile_list, const int n_list,
    const int partial_off) {
    // LIST: the launch covers the n_list tiles of tile_list (interior or boundary tiles of a row-partitioned operator,
    // csr_build_split_plan) instead of all n_wtiles, and its partial sums start at partial_off.  A template parameter, not a
    // run-time test: with the test in the sweep loop the one-GPU kernels lost 6 % (the W product 45 %)
    constexpr bool WITH_DOT = (MODE == SPMV_DOT) || (MODE == SPMV_DOT_AUX) || (MODE == SPMV_JACOBI) || (MODE == SPMV_WUP);
    // LONG (single-precision operators with more than 8 entries per row on average: restrictions, coarse operators): 16
    // elements per lane and pass of the gather path -- the same 4 KiB of LDS per wave as 8 doubles, one pass less per tile.
    // Not for shorter rows: the passes are unrolled, and the W product (5 entries per row) lost 30 % to the idle half
    constexpr int kEplGather = LONG ? 2 * kEpl : kEpl;
    static_assert(!LONG || sizeof(XT) == 4, "16 elements per lane only for single-precision vectors");
    __shared__ XT prod_all[4 * 64 * kEplGather];
    extern __shared__ __attribute__((aligned(16))) unsigned char xs_dyn[];      // 4 * kXwRuns * xw_run entries of XT when the plan is in use
    ST *xs_all = reinterpret_cast<ST *>(xs_dyn);
    __shared__ double red[4];

    // (the early exit of a converged solve is taken BEHIND the first loads of the kernel -- the stop word is a dependent load of
    // its own, a microsecond in front of everything else when it is waited for first; the loads it overtakes are harmless)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    XT *prod = prod_all + w * 64 * kEplGather;
    ST *xs = xs_all + w * (WIDE ? kXwRunsWide * kXwRunWide : kXwRuns * xw_run);
    double out_mul = 1.0;
    if (out_scale2 != nullptr) {
        const double s2 = *out_scale2;
        out_mul = s2 > 0.0 ? sqrt(s2) : 1.0;
    }
    const int G = gridDim.x;
    // slabs: one per XCD when the grid is a multiple of 8, otherwise a single slab
    const int nslab = (G % kNumXcd == 0) ? kNumXcd : 1;
    const int slab = blockIdx.x % nslab;
    const int wx = (blockIdx.x / nslab) * 4 + w;       // wave index inside the slab
    const int wps = (G / nslab) * 4;                   // waves per slab
    const int n_sweep = LIST ? n_list : n_wtiles;
    const int s0 = (int)((long long)slab * n_sweep / nslab);
    const int s1 = (int)((long long)(slab + 1) * n_sweep / nslab);

    double dot_acc = 0.0;
    int rs = 0, re = 0;
    int4 d = make_int4(0, 0, 0, 0);
    // A tile is a chain of dependent accesses: its row pointers and descriptor, then the runs of x and the stream of the
    // matrix they delimit, then LDS.  The head of the chain is asked for one tile ahead (and, for a list of tiles, the
    // tile's number one further ahead): it arrives while the tile in front of it is being multiplied.
    int wt_n = -1, wt_nn = -1, rs_n = 0, re_n = 0, wide_n = 0;
    int4 d_n = make_int4(0, 0, 0, 0);
    auto tile_at = [&](const int it) -> int { return it < s1 ? (LIST ? tile_list[it] : it) : -1; };
    auto fetch_head = [&](const int wt) {
        rs_n = 0;
        re_n = 0;
        wide_n = 0;
        d_n = make_int4(0, 0, 0, 0);
        if (wt < 0) return;
        const int r = wt * 64 + lane;
        if (r < n_rows) {
            rs_n = rowptr[r];
            re_n = rowptr[r + 1];
        }
        if (WIDE) {
            if (lane < kXwDescWide) wide_n = reinterpret_cast<const int *>(xw_desc)[(size_t)wt * kXwDescWide + lane];
        } else if (xw_desc != nullptr) {
            d_n = xw_desc[wt];
        }
    };
    wt_n = tile_at(s0 + wx);
    wt_nn = tile_at(s0 + wx + wps);
    fetch_head(wt_n);
    if (done_flag != nullptr && *done_flag != 0) return;
    for (int it = s0 + wx; it < s1; it += wps) {
        const int wt = wt_n;
        const int row0 = wt * 64;
        const int row1 = min(row0 + 64, n_rows);
        const int r = row0 + lane;
        rs = rs_n;
        re = re_n;
        d = d_n;
        const int wide_mine = wide_n;                       // WIDE: lane k holds int k of the tile's descriptor
        if (WIDE) d.w = __shfl(wide_mine, kXwDescWide - 1, 64);
        wt_n = wt_nn;
        wt_nn = tile_at(it + 2 * wps);
        fetch_head(wt_n);
        const int k0 = __shfl(rs, 0, 64);
        const int k1 = __shfl(re, row1 - row0 - 1, 64);
        XT acc = 0;
        const bool windowed = xw_desc != nullptr && d.w != 0;      // wave-uniform
        if (WIDE && windowed) {
            stage_windows_wide<ST>(x, n_cols, wide_mine, xs, lane);
            acc = xw_stream_tile<4, unsigned char, VT, XT, ST>(vals, (const unsigned char *)xw_lidx, xs, prod, k0, k1, rs, re, lane,
                                                               kXwRunsWide * kXwRunWide - 1);
        } else if (windowed) {
            // the tile's runs of x (stage_windows; the stream loops wait for these LDS stores before their first read)
            constexpr bool PRE = MODE == SPMV_RESID_PRE && sizeof(ST) == sizeof(XT);
            if (xw_run <= kXwRunShort) stage_windows<kXwRunShort, ST, PRE>(x, n_cols, d, xs, lane, (const ST *)aux2, (ST)scale);
            else stage_windows<kXwRunLong, ST, PRE>(x, n_cols, d, xs, lane, (const ST *)aux2, (ST)scale);
            const int top = kXwRuns * xw_run - 1;
            // three runs of 72 are 216 positions: one byte each, four non-zeros per lane and load; runs of 128 need 16 bits
            if (xw_run <= kXwRunShort)
                acc = xw_stream_tile<4, unsigned char, VT, XT, ST>(vals, (const unsigned char *)xw_lidx, xs, prod, k0, k1, rs, re, lane, top);
            else
                acc = xw_stream_tile<2, unsigned short, VT, XT, ST>(vals, (const unsigned short *)xw_lidx, xs, prod, k0, k1, rs, re, lane, top);
        } else if (k1 > k0) {
            acc = gather_stream_tile<VT, XT, kEplGather, ST, MODE == SPMV_RESID_PRE>(cols, vals, x, prod, k0, k1, rs, re, lane, aux2, scale);
        }
        if (r < row1) {
            if (MODE == SPMV_PLAIN) {
                y[r] = (YT)acc;
            } else if (MODE == SPMV_RESTRICT) {
                y[r] = (YT)acc;
                y2[r] = scale * aux2[r] * acc;
            } else if (MODE == SPMV_DOT || MODE == SPMV_DOT_AUX) {
                y[r] = (YT)acc;
                if (sizeof(ST) != sizeof(XT)) dot_acc += (double)x[r] * (double)acc;      // p.q with the stored p
                else dot_acc += dot_with[r] * (double)acc;
            } else if (MODE == SPMV_RESID) {
                y[r] = (YT)(aux1[r] - acc);
            } else if (MODE == SPMV_RESID_PRE) {
                y[r] = (YT)((XT)x[r] - acc);
            } else if (MODE == SPMV_ADD) {
                y[r] += (YT)acc;
            } else if (MODE == SPMV_WUP) {
                // y2 (optional, exit stage of the fine level): the level's right-hand side, r / ||b|| in single precision.  The
                // pre-smoothed iterate is c D^-1 of it (a sweep from zero), so x_pre + c D^-1 r_pre = c D^-1 (b + r_pre) and x_pre
                // need not be read; and r.z is taken against it -- 4 instead of 8 bytes per row, its rounding (6e-8 of every term)
                // goes into alpha and beta like the cycle's own
                const XT rhs = y2 != nullptr ? y2[r] : (XT)0;
                const XT out = y2 != nullptr ? scale * aux2[r] * (rhs + aux1[r]) + acc : aux0[r] + scale * aux2[r] * aux1[r] + acc;
                if (dot_with != nullptr) {
                    const double outd = (double)out * out_mul;
                    // a float result leaves unscaled (the consumer multiplies: the same double comes out)
                    y[r] = sizeof(YT) == 4 ? (YT)out : (YT)outd;
                    dot_acc += (y2 != nullptr ? (double)rhs * out_mul : dot_with[r]) * outd;
                } else {
                    y[r] = (YT)out;
                }
            } else {
                const XT b = aux1[r];
                const XT out = (XT)x[r] + scale * aux2[r] * (b - acc);
                if (dot_with != nullptr) {
                    const double outd = (double)out * out_mul;
                    y[r] = sizeof(YT) == 4 ? (YT)out : (YT)outd;
                    dot_acc += dot_with[r] * outd;
                } else {
                    y[r] = (YT)out;
                    dot_acc += (double)(b * out);
                }
            }
        }
    }
    if (WITH_DOT && partials != nullptr) {
        const double s = block_sum_256(dot_acc, red);
        if (threadIdx.x == 0) partials[(LIST ? partial_off : 0) + blockIdx.x] = s;
    }
}

// Wave-per-row variant for the small, dense operators at the bottom of the multigrid hierarchy (tens to
// hundreds of non-zeros per row, a few thousand rows): there the tile kernel above is latency-bound on a
// handful of waves.  The lanes of a wave stride over one row and the partial sums are combined with a
// fixed shuffle tree, so results are reproducible (but not in CSR order: never used for the fine matrix).
template <int MODE, typename VT, typename XT, typename YT>
__global__ __launch_bounds__(kSpmvThreads) void csr_spmv_wpr_kernel(
    const int n_rows, const int *__restrict__ rowptr, const int *__restrict__ cols, const VT *__restrict__ vals,
    const XT *__restrict__ x, YT *__restrict__ y, const double *__restrict__ dot_with,
    double *__restrict__ partials, const int *__restrict__ done_flag, const XT *__restrict__ aux1,
    const XT *__restrict__ aux2, const XT scale, const double *__restrict__ out_scale2, XT *__restrict__ y2,
    const XT *__restrict__ aux0) {
    constexpr bool WITH_DOT = (MODE == SPMV_DOT) || (MODE == SPMV_DOT_AUX) || (MODE == SPMV_JACOBI);
    __shared__ double red[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int W = gridDim.x * 4;
    // (the first row's bounds are asked for in front of the stop word, as in the tile kernel)
    const int r_first = blockIdx.x * 4 + w;
    int rs_first = 0, re_first = 0;
    if (r_first < n_rows) {
        rs_first = rowptr[r_first];
        re_first = rowptr[r_first + 1];
    }
    if (done_flag != nullptr && *done_flag != 0) return;
    double out_mul = 1.0;
    if (out_scale2 != nullptr) {
        const double s2 = *out_scale2;
        out_mul = s2 > 0.0 ? sqrt(s2) : 1.0;
    }
    double dot_acc = 0.0;
    for (int r = r_first; r < n_rows; r += W) {
        const int rs = r == r_first ? rs_first : rowptr[r], re = r == r_first ? re_first : rowptr[r + 1];
        XT acc = 0;
        for (int k = rs + lane; k < re; k += 64) acc += (XT)vals[k] * x[cols[k]];
        acc = wave_sum_t(acc);
        if (lane == 0) {
            if (MODE == SPMV_PLAIN) {
                y[r] = (YT)acc;
            } else if (MODE == SPMV_RESTRICT) {
                y[r] = (YT)acc;
                y2[r] = scale * aux2[r] * acc;
            } else if (MODE == SPMV_DOT || MODE == SPMV_DOT_AUX) {
                y[r] = (YT)acc;
                dot_acc += dot_with[r] * (double)acc;
            } else if (MODE == SPMV_RESID) {
                y[r] = (YT)(aux1[r] - acc);
            } else if (MODE == SPMV_ADD) {
                y[r] += (YT)acc;
            } else if (MODE == SPMV_WUP) {      // inner levels only: no exit stage here
                y[r] = (YT)(aux0[r] + scale * aux2[r] * aux1[r] + acc);
            } else {
                const XT b = aux1[r];
                const XT out = x[r] + scale * aux2[r] * (b - acc);
                if (dot_with != nullptr) {
                    const double outd = (double)out * out_mul;
                    y[r] = sizeof(YT) == 4 ? (YT)out : (YT)outd;
                    dot_acc += dot_with[r] * outd;
                } else {
                    y[r] = (YT)out;
                    dot_acc += (double)(b * out);
                }
            }
        }
    }
    if (WITH_DOT && partials != nullptr) {
        const double s = block_sum_256(dot_acc, red);
        if (threadIdx.x == 0) partials[blockIdx.x] = s;
    }
}

static inline bool use_wave_per_row(const padne_csr *m) {
    // (a few thousand rows are a handful of 256-row workgroups for the tile kernel: the restriction onto the coarsest
    // level, 1617 rows of 21 entries, took 14 us there and takes 5 us with a wave per row)
    return m->hierarchy_operator && m->n_rows > 0 && m->n_rows <= 65536 &&
           (m->nnz >= 24 * m->n_rows || (m->n_rows <= 4096 && m->nnz >= 8 * m->n_rows));
}

int spmv_grid(const padne_csr *m) {
    if (use_wave_per_row(m)) {
        long long g = (m->n_rows + 3) / 4;
        return (int)(g < kMaxPartials ? g : kMaxPartials);
    }
    const long long n_tiles = (m->n_rows + kSpmvRows - 1) / kSpmvRows;   // 4 wave-tiles per workgroup-turn
    long long g = n_tiles < kMaxPartials ? n_tiles : kMaxPartials;
    // with the x windows a double-precision workgroup holds 23 KiB of LDS: six fit on a CU, so the persistent
    // sweep uses 6 x 256 workgroups (a seventh and eighth would run as a second wave of work)
    // (the wide plan of a single-precision operator stages under 4 KiB per workgroup: the full grid)
    if (m->xw_state == 1 && m->xw_nruns == kXwRuns && g > (m->xw_run > kXwRunShort ? 1280 : 1536)) g = m->xw_run > kXwRunShort ? 1280 : 1536;
    if (g >= kNumXcd) g -= g % kNumXcd;
    if (g < 1) g = 1;
    return (int)g;
}

static int split_grid(const padne_csr *m, int n_list) {
    long long g = ((long long)n_list + 3) / 4;               // four wave-tiles per workgroup turn
    const int cap = spmv_grid(m);
    if (g > cap) g = cap;
    if (g >= kNumXcd) g -= g % kNumXcd;
    return (int)(g < 1 ? 1 : g);
}

static bool split_in_use(const padne_csr *m) { return m->split_state == 1 && !use_wave_per_row(m); }

int spmv_partials(const padne_csr *m) {
    if (!split_in_use(m)) return spmv_grid(m);
    return split_grid(m, m->split_n_int) + (m->split_n_bnd > 0 ? split_grid(m, m->split_n_bnd) : 0);
}

template <typename VT, typename XT, typename YT>
static int launch_spmv_typed(padne_ctx *ctx, const padne_csr *m, const VT *vals, int mode, const XT *x, YT *y,
                             const double *dot_with, double *partials, const int32_t *done_flag, const XT *aux1,
                             const XT *aux2, XT scale, const double *out_scale2, const XT *aux0 = nullptr, XT *y2 = nullptr,
                             int part = SPMV_ALL) {
    if (m->n_rows == 0) return PADNE_OK;
    const int n_tiles = (int)((m->n_rows + 63) / 64);   // wave-tiles of 64 rows
    if (!split_in_use(m)) {
        if (part == SPMV_INTERIOR) return PADNE_OK;      // no split plan: the whole product follows the exchange
        part = SPMV_ALL;
    } else if (part == SPMV_ALL && partials == nullptr) {
        part = -1;                                       // one sweep over all tiles: nobody counts partial sums
    }
    if (part == SPMV_ALL && split_in_use(m)) {
        // a product with partial sums on a split operator always leaves them in the layout of the two launches
        PADNE_TRY((launch_spmv_typed<VT, XT, YT>(ctx, m, vals, mode, x, y, dot_with, partials, done_flag, aux1, aux2, scale,
                                                 out_scale2, aux0, y2, SPMV_INTERIOR)));
        return launch_spmv_typed<VT, XT, YT>(ctx, m, vals, mode, x, y, dot_with, partials, done_flag, aux1, aux2, scale,
                                             out_scale2, aux0, y2, SPMV_BOUNDARY);
    }
    int g = spmv_grid(m);
    const int *tile_list = nullptr;
    int n_list = 0, partial_off = 0;
    if (part == SPMV_INTERIOR) {
        tile_list = m->split_tiles;
        n_list = m->split_n_int;
        g = split_grid(m, n_list);
    } else if (part == SPMV_BOUNDARY) {
        if (m->split_n_bnd == 0) return PADNE_OK;
        tile_list = m->split_tiles + m->split_n_int;
        n_list = m->split_n_bnd;
        partial_off = split_grid(m, m->split_n_int);
        g = split_grid(m, n_list);
    }
    if (use_wave_per_row(m) && !(mode == SPMV_WUP && dot_with != nullptr)) {
#define PADNE_SPMV_WPR(M)                                                                                           \
    hipLaunchKernelGGL((csr_spmv_wpr_kernel<M, VT, XT, YT>), dim3(g), dim3(kSpmvThreads), 0, ctx->stream,            \
                       (int)m->n_rows, m->rowptr, m->cols, vals, x, y, dot_with, partials, done_flag, aux1, aux2,    \
                       scale, out_scale2, y2, aux0)
        switch (mode) {
            case SPMV_PLAIN: PADNE_SPMV_WPR(SPMV_PLAIN); break;
            case SPMV_DOT: PADNE_SPMV_WPR(SPMV_DOT); break;
            case SPMV_DOT_AUX: PADNE_SPMV_WPR(SPMV_DOT_AUX); break;
            case SPMV_RESID: PADNE_SPMV_WPR(SPMV_RESID); break;
            case SPMV_ADD: PADNE_SPMV_WPR(SPMV_ADD); break;
            case SPMV_JACOBI: PADNE_SPMV_WPR(SPMV_JACOBI); break;
            case SPMV_RESTRICT: PADNE_SPMV_WPR(SPMV_RESTRICT); break;
            case SPMV_WUP: PADNE_SPMV_WPR(SPMV_WUP); break;
            default: set_error("bad SpMV mode %d", mode); return PADNE_E_INVALID;
        }
#undef PADNE_SPMV_WPR
        PADNE_HIP_CHECK(hipGetLastError());
        return PADNE_OK;
    }
    const bool wide = m->xw_state == 1 && m->xw_nruns == kXwRunsWide;
    const size_t xs_bytes = m->xw_state == 1 ? sizeof(XT) * 4 * (wide ? (size_t)kXwRunsWide * kXwRunWide : kXwRuns * (size_t)m->xw_run) : 0;
    const int4 *xw_desc = m->xw_state == 1 ? m->xw_desc : nullptr;
    const void *xw_lidx = m->xw_state == 1 ? (const void *)m->xw_lidx : nullptr;
#define PADNE_SPMV_ARGS                                                                                          \
    (int)m->n_rows, (int)m->n_cols, n_tiles, m->rowptr, m->cols, vals, x, y, dot_with, partials, done_flag, aux1, aux2, \
        scale, out_scale2, xw_desc, xw_lidx, m->xw_run, aux0, y2, tile_list, n_list, partial_off
#define PADNE_SPMV_LAUNCH(M)                                                                                     \
    hipLaunchKernelGGL((csr_spmv_kernel<M, VT, XT, YT, false>), dim3(g), dim3(kSpmvThreads), xs_bytes, ctx->stream, PADNE_SPMV_ARGS)
#define PADNE_SPMV_LAUNCH_LONG(M)                                                                                \
    hipLaunchKernelGGL((csr_spmv_kernel<M, VT, XT, YT, false, sizeof(XT) == 4 && sizeof(YT) == 4>), dim3(g), dim3(kSpmvThreads), xs_bytes, ctx->stream, PADNE_SPMV_ARGS)
#define PADNE_SPMV_LAUNCH_LIST(M)                                                                                \
    hipLaunchKernelGGL((csr_spmv_kernel<M, VT, XT, YT, true>), dim3(g), dim3(kSpmvThreads), xs_bytes, ctx->stream, PADNE_SPMV_ARGS)
    if (tile_list != nullptr) {
        // the products that follow a halo exchange: q = A p of the CG loop, the Lanczos steps, residual and smoothing of the cycle
        switch (mode) {
            case SPMV_PLAIN: PADNE_SPMV_LAUNCH_LIST(SPMV_PLAIN); break;
            case SPMV_DOT: PADNE_SPMV_LAUNCH_LIST(SPMV_DOT); break;
            case SPMV_DOT_AUX: PADNE_SPMV_LAUNCH_LIST(SPMV_DOT_AUX); break;
            case SPMV_RESID: PADNE_SPMV_LAUNCH_LIST(SPMV_RESID); break;
            case SPMV_JACOBI: PADNE_SPMV_LAUNCH_LIST(SPMV_JACOBI); break;
            default: set_error("SpMV mode %d has no interior / boundary form", mode); return PADNE_E_INVALID;
        }
        PADNE_HIP_CHECK(hipGetLastError());
        return PADNE_OK;
    }
    // single-precision operators of the cycle with long rows, no x windows and few tiles per wave (the levels below the
    // first coarse one: 139 k rows at C4): the 16-per-lane form of the gather path, -11 us per cycle there.  On the million-row
    // operators (the fine restriction, the first coarse level) it loses 2-5 us each: the shorter form stays
    const bool long_rows = sizeof(XT) == 4 && sizeof(YT) == 4 && xw_desc == nullptr && m->n_rows < 500000 &&
                           m->nnz > 8 * m->n_rows + 4 * (m->n_rows >> 3);
    if (long_rows && (mode == SPMV_PLAIN || mode == SPMV_RESID || mode == SPMV_ADD || mode == SPMV_JACOBI || mode == SPMV_RESTRICT)) {
        switch (mode) {
            case SPMV_PLAIN: PADNE_SPMV_LAUNCH_LONG(SPMV_PLAIN); break;
            case SPMV_RESID: PADNE_SPMV_LAUNCH_LONG(SPMV_RESID); break;
            case SPMV_ADD: PADNE_SPMV_LAUNCH_LONG(SPMV_ADD); break;
            case SPMV_JACOBI: PADNE_SPMV_LAUNCH_LONG(SPMV_JACOBI); break;
            default: PADNE_SPMV_LAUNCH_LONG(SPMV_RESTRICT); break;
        }
        PADNE_HIP_CHECK(hipGetLastError());
        return PADNE_OK;
    }
    if (wide) {
        // the wide plan exists for the fused up-leg product only (single-precision values, csr_build_xw_plan_wide)
        PADNE_REQUIRE(mode == SPMV_WUP && sizeof(VT) == 4 && sizeof(XT) == 4, "the wide x-window plan serves the W product");
        hipLaunchKernelGGL((csr_spmv_kernel<SPMV_WUP, VT, XT, YT, false, false, true>), dim3(g), dim3(kSpmvThreads), xs_bytes,
                           ctx->stream, PADNE_SPMV_ARGS);
        PADNE_HIP_CHECK(hipGetLastError());
        return PADNE_OK;
    }
    switch (mode) {
        case SPMV_PLAIN: PADNE_SPMV_LAUNCH(SPMV_PLAIN); break;
        case SPMV_DOT: PADNE_SPMV_LAUNCH(SPMV_DOT); break;
        case SPMV_DOT_AUX: PADNE_SPMV_LAUNCH(SPMV_DOT_AUX); break;
        case SPMV_RESID: PADNE_SPMV_LAUNCH(SPMV_RESID); break;
        case SPMV_ADD: PADNE_SPMV_LAUNCH(SPMV_ADD); break;
        case SPMV_JACOBI: PADNE_SPMV_LAUNCH(SPMV_JACOBI); break;
        case SPMV_WUP: PADNE_SPMV_LAUNCH(SPMV_WUP); break;
        case SPMV_RESTRICT: PADNE_SPMV_LAUNCH(SPMV_RESTRICT); break;
        case SPMV_RESID_PRE: PADNE_SPMV_LAUNCH(SPMV_RESID_PRE); break;
        default: set_error("bad SpMV mode %d", mode); return PADNE_E_INVALID;
    }
#undef PADNE_SPMV_LAUNCH_LONG
#undef PADNE_SPMV_LAUNCH_LIST
#undef PADNE_SPMV_ARGS
#undef PADNE_SPMV_LAUNCH
    PADNE_HIP_CHECK(hipGetLastError());
    return PADNE_OK;
}

// Residual of the first sweep of a level from a zero start, without that sweep's result in memory: the iterate is
// c D^-1 b, so  r = b - A (c D^-1 b)  is formed from b alone -- the windows of x are staged as c * dinv .* b (gathered entries
// likewise).  What it saves is the store and the staged read of the iterate (amg.hip: the fine level of the float cycle, whose
// up-leg does not read the iterate either).  Plain tile kernel only (spmv_resid_pre_ok).
bool spmv_resid_pre_ok(const padne_csr *m) {
    const bool long_rows = m->xw_state != 1 && m->n_rows < 500000 && m->nnz > 8 * m->n_rows + 4 * (m->n_rows >> 3);
    return m->vals32 != nullptr && m->dinv32 != nullptr && !split_in_use(m) && !use_wave_per_row(m) && !long_rows &&
           !(m->xw_state == 1 && m->xw_nruns == kXwRunsWide);
}
int launch_spmv_f32_resid_pre(padne_ctx *ctx, const padne_csr *m, const float *b, float *resid, const int32_t *done_flag,
                              const float *dinv32, float c) {
    PADNE_REQUIRE(spmv_resid_pre_ok(m), "residual of the sweep from zero on this operator");
    return launch_spmv_typed<float, float, float>(ctx, m, m->vals32, SPMV_RESID_PRE, b, resid, nullptr, nullptr, done_flag, nullptr,
                                                  dinv32, c, nullptr);
}

// q = A p with p stored in single precision, p.q partials (one GPU, no split plan; same grid and partial layout as the
// double form).  false from spmv_x32_ok: the caller keeps p in double.
bool spmv_x32_ok(const padne_csr *m) {
    return m->vals != nullptr && !split_in_use(m) && !use_wave_per_row(m) && !(m->xw_state == 1 && m->xw_nruns == kXwRunsWide) &&
           !(m->owner != nullptr && m->owner->opt.pcg_p64);
}
int launch_spmv_dot_x32(padne_ctx *ctx, const padne_csr *m, const float *x, double *y, double *partials, const int32_t *done_flag) {
    PADNE_REQUIRE(spmv_x32_ok(m), "single-precision search direction on this operator");
    if (m->n_rows == 0) return PADNE_OK;
    const int n_tiles = (int)((m->n_rows + 63) / 64);
    const int g = spmv_grid(m);      // (= spmv_partials: the consumers of the p.q partials count on it)
    const size_t xs_bytes = m->xw_state == 1 ? sizeof(float) * 4 * kXwRuns * (size_t)m->xw_run : 0;
    const int4 *xw_desc = m->xw_state == 1 ? m->xw_desc : nullptr;
    const void *xw_lidx = m->xw_state == 1 ? (const void *)m->xw_lidx : nullptr;
    hipLaunchKernelGGL((csr_spmv_kernel<SPMV_DOT, double, double, double, false, false, false, float>), dim3(g), dim3(kSpmvThreads),
                       xs_bytes, ctx->stream, (int)m->n_rows, (int)m->n_cols, n_tiles, m->rowptr, m->cols, m->vals, x, y,
                       (const double *)nullptr, partials, done_flag, (const double *)nullptr, (const double *)nullptr, 0.0,
                       (const double *)nullptr, xw_desc, xw_lidx, m->xw_run, (const double *)nullptr, (double *)nullptr,
                       (const int *)nullptr, 0, 0);
    PADNE_HIP_CHECK(hipGetLastError());
    return PADNE_OK;
}

int launch_spmv_mode(padne_ctx *ctx, const padne_csr *m, int mode, const double *x, double *y,
                     const double *dot_with, double *partials, const int32_t *done_flag, const double *aux1,
                     const double *aux2, double scale) {
    return launch_spmv_typed<double, double, double>(ctx, m, m->vals, mode, x, y, dot_with, partials, done_flag, aux1,
                                                     aux2, scale, nullptr);
}

int launch_spmv_part(padne_ctx *ctx, const padne_csr *m, int mode, int part, const double *x, double *y, const double *dot_with,
                     double *partials, const int32_t *done_flag, const double *aux1, const double *aux2, double scale) {
    return launch_spmv_typed<double, double, double>(ctx, m, m->vals, mode, x, y, dot_with, partials, done_flag, aux1,
                                                     aux2, scale, nullptr, nullptr, nullptr, part);
}

int launch_spmv_f32_part(padne_ctx *ctx, const padne_csr *m, int mode, int part, const float *x, float *y, double *partials,
                         const int32_t *done_flag, const float *aux1, const float *aux2, float scale) {
    PADNE_REQUIRE(m->vals32 != nullptr, "single-precision copy missing");
    return launch_spmv_typed<float, float, float>(ctx, m, m->vals32, mode, x, y, nullptr, partials, done_flag, aux1,
                                                  aux2, scale, nullptr, nullptr, nullptr, part);
}

int launch_spmv_f32_exit_part(padne_ctx *ctx, const padne_csr *m, int part, const float *x, double *y, const double *dot_with,
                              double *partials, const int32_t *done_flag, const float *aux1, const float *aux2, float scale,
                              const double *out_scale2, float *z32) {
    PADNE_REQUIRE(m->vals32 != nullptr && dot_with != nullptr, "single-precision exit stage");
    if (z32 != nullptr)
        return launch_spmv_typed<float, float, float>(ctx, m, m->vals32, SPMV_JACOBI, x, z32, dot_with, partials, done_flag,
                                                      aux1, aux2, scale, out_scale2, nullptr, nullptr, part);
    return launch_spmv_typed<float, float, double>(ctx, m, m->vals32, SPMV_JACOBI, x, y, dot_with, partials, done_flag,
                                                   aux1, aux2, scale, out_scale2, nullptr, nullptr, part);
}

// single-precision operator copy (csr_build_f32) on single-precision vectors
int launch_spmv_f32(padne_ctx *ctx, const padne_csr *m, int mode, const float *x, float *y, double *partials,
                    const int32_t *done_flag, const float *aux1, const float *aux2, float scale) {
    PADNE_REQUIRE(m->vals32 != nullptr, "single-precision copy missing");
    return launch_spmv_typed<float, float, float>(ctx, m, m->vals32, mode, x, y, nullptr, partials, done_flag, aux1,
                                                  aux2, scale, nullptr);
}

// restriction b_c = R r and, in the same pass, the pre-smoothed start of the coarse level x_c = c D_c^-1 b_c
int launch_spmv_f32_restrict(padne_ctx *ctx, const padne_csr *R, const float *r, float *b_c, float *x_c,
                             const int32_t *done_flag, const float *dinv_c, float c) {
    PADNE_REQUIRE(R->vals32 != nullptr && x_c != nullptr && dinv_c != nullptr, "single-precision restriction");
    return launch_spmv_typed<float, float, float>(ctx, R, R->vals32, SPMV_RESTRICT, r, b_c, nullptr, nullptr, done_flag,
                                                  nullptr, dinv_c, c, nullptr, nullptr, x_c);
}

// last stage of the single-precision cycle: damped-Jacobi sweep whose result goes out in double, multiplied by
// sqrt(*out_scale2), with partial sums of dot_with . y
// z32 != nullptr: the result goes there in single precision and UNSCALED instead (the dot product is the same)
int launch_spmv_f32_exit(padne_ctx *ctx, const padne_csr *m, const float *x, double *y, const double *dot_with,
                         double *partials, const int32_t *done_flag, const float *aux1, const float *aux2, float scale,
                         const double *out_scale2, float *z32) {
    PADNE_REQUIRE(m->vals32 != nullptr && dot_with != nullptr, "single-precision exit stage");
    if (z32 != nullptr)
        return launch_spmv_typed<float, float, float>(ctx, m, m->vals32, SPMV_JACOBI, x, z32, dot_with, partials, done_flag,
                                                      aux1, aux2, scale, out_scale2);
    return launch_spmv_typed<float, float, double>(ctx, m, m->vals32, SPMV_JACOBI, x, y, dot_with, partials, done_flag,
                                                   aux1, aux2, scale, out_scale2);
}

// last stage of the single-precision cycle in the W form: z = (x_pre + c D^-1 r_pre + W e) * sqrt(*out_scale2) in double,
// with partial sums of dot_with . z.  W carries single-precision values only (m->vals32).
int launch_spmv_f32_wup_exit(padne_ctx *ctx, const padne_csr *w, const float *e, double *z, const double *dot_with,
                             double *partials, const int32_t *done_flag, const float *x_pre, const float *r_pre,
                             const float *dinv32, float scale, const double *out_scale2, float *z32, const float *dot_b32) {
    PADNE_REQUIRE(w->vals32 != nullptr && dot_with != nullptr, "single-precision W stage");
    if (z32 != nullptr)
        return launch_spmv_typed<float, float, float>(ctx, w, w->vals32, SPMV_WUP, e, z32, dot_with, partials, done_flag,
                                                      r_pre, dinv32, scale, out_scale2, x_pre, const_cast<float *>(dot_b32));
    return launch_spmv_typed<float, float, double>(ctx, w, w->vals32, SPMV_WUP, e, z, dot_with, partials, done_flag, r_pre,
                                                   dinv32, scale, out_scale2, x_pre, const_cast<float *>(dot_b32));
}

// up-leg of an inner level in the W form: x = x_pre + c D^-1 r_pre + W e, single precision throughout
int launch_spmv_f32_wup(padne_ctx *ctx, const padne_csr *w, const float *e, float *x_out, const int32_t *done_flag,
                        const float *x_pre, const float *r_pre, const float *dinv32, float scale) {
    PADNE_REQUIRE(w->vals32 != nullptr, "single-precision W stage");
    return launch_spmv_typed<float, float, float>(ctx, w, w->vals32, SPMV_WUP, e, x_out, nullptr, nullptr, done_flag, r_pre,
                                                  dinv32, scale, nullptr, x_pre);
}

// ---- x-window plan -------------------------------------------------------------------------------------
// One wave per 64-row tile: greedy cover of the tile's columns by runs of kXwRun entries starting at the smallest
// uncovered column; at most kXwRuns runs or the tile keeps the gather path.  A qualifying tile gets, for every
// non-zero, the 16-bit position of its column inside the staged runs.
template <typename IT>      // unsigned char for runs of 72 (216 positions), unsigned short for runs of 128
__global__ __launch_bounds__(256) void xw_plan_kernel(int n_rows, int n_wtiles, int run, const int *__restrict__ rowptr,
                                                      const int *__restrict__ cols, int4 *__restrict__ desc,
                                                      IT *__restrict__ lidx, int *__restrict__ n_ok) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long W = (long long)gridDim.x * 4, gw = (long long)blockIdx.x * 4 + w;
    int ok_count = 0;
    for (long long wt = gw; wt < n_wtiles; wt += W) {
        const int row0 = (int)wt * 64;
        const int row1 = min(row0 + 64, n_rows);
        const int k0 = rowptr[row0], k1 = rowptr[row1];
        int start[kXwRuns];
        int bound = -1;                      // columns <= bound are covered
        bool fits = true;
        // the tile's columns in registers when there are at most 512 of them (a mesh Laplacian: ~450): the four passes
        // for the run starts and the pass that writes the positions then need no memory round trip of their own
        constexpr int kReg = 8;
        const bool in_regs = k1 - k0 <= 64 * kReg;      // wave-uniform
        int creg[kReg];
#pragma unroll
        for (int j = 0; j < kReg; ++j) {
            const int e = k0 + lane + 64 * j;
            creg[j] = (in_regs && e < k1) ? cols[e] : 0x7fffffff;
        }
#pragma unroll
        for (int q = 0; q <= kXwRuns; ++q) {
            int mn = 0x7fffffff;
            if (in_regs) {
#pragma unroll
                for (int j = 0; j < kReg; ++j)
                    if (creg[j] > bound && creg[j] < mn) mn = creg[j];
            } else {
                for (int e = k0 + lane; e < k1; e += 64) {
                    const int c = cols[e];
                    if (c > bound && c < mn) mn = c;
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) mn = min(mn, __shfl_xor(mn, off, 64));
            if (q == kXwRuns) {
                fits = mn == 0x7fffffff;     // nothing left beyond the last run
            } else {
                start[q] = mn == 0x7fffffff ? (q > 0 ? start[q - 1] : 0) : mn;
                if (mn != 0x7fffffff) bound = mn + run - 1;
            }
        }
        if (fits) {
            if (in_regs) {
#pragma unroll
                for (int j = 0; j < kReg; ++j) {
                    const int e = k0 + lane + 64 * j;
                    if (e < k1) {
                        const int c = creg[j];
                        int pos = 0;
#pragma unroll
                        for (int q = kXwRuns - 1; q >= 0; --q)
                            if (c >= start[q] && c < start[q] + run) pos = q * run + (c - start[q]);
                        lidx[e] = (IT)pos;
                    }
                }
            } else {
                for (int e = k0 + lane; e < k1; e += 64) {
                    const int c = cols[e];
                    int pos = 0;
#pragma unroll
                    for (int q = kXwRuns - 1; q >= 0; --q)
                        if (c >= start[q] && c < start[q] + run) pos = q * run + (c - start[q]);
                    lidx[e] = (IT)pos;
                }
            }
            ++ok_count;
        }
        if (lane == 0) desc[wt] = make_int4(start[0], start[1], start[2], fits ? 1 : 0);
    }
    // one atomic per workgroup, spread over 8 counters: 32 768 waves adding to ONE address are serialised in the L2 -- that
    // alone was 0.3 ms of this kernel's 0.4
    __shared__ int ok_block[4];
    if (lane == 0) ok_block[w] = ok_count;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t = ok_block[0] + ok_block[1] + ok_block[2] + ok_block[3];
        if (t) atomicAdd(n_ok + (blockIdx.x & 7), t);
    }
}

int csr_build_xw_plan(padne_ctx *ctx, padne_csr *m) {
    if (m->xw_state != 0) return PADNE_OK;
    m->xw_state = -1;
    // worth examining only for the big streaming operators.  Hierarchy operators: the first coarse operators of a mesh
    // problem do qualify with runs of 128 (96 % of the tiles of A_1 of config C4), the prolongators do not (five bands of
    // aggregates); measured, the plan of A_1 / A_2 costs 0.3-0.8 ms of setup and buys nothing per iteration (1030 us
    // either way: those passes are not gather-bound), so they keep the gather path.
    if (m->n_rows < 65536 || m->hierarchy_operator || m->nnz > 64LL * m->n_rows || ctx->opt.no_xwindow)
        return PADNE_OK;
    padne_ctx *owner = m->owner ? m->owner : ctx;
    const int n_tiles = (int)((m->n_rows + 63) / 64);
    int4 *desc = (int4 *)pool_alloc(owner, sizeof(int4) * (size_t)n_tiles);
    unsigned short *lidx = (unsigned short *)pool_alloc(owner, sizeof(unsigned short) * ((size_t)m->nnz + kPadNnz));
    int *d_ok = (int *)pool_alloc(ctx, 8 * sizeof(int));
    if (!desc || !lidx || !d_ok) {
        pool_free(owner, desc);
        pool_free(owner, lidx);
        pool_free(ctx, d_ok);
        return PADNE_E_NOMEM;
    }
    int h_ok = 0, run = 0;
    hipError_t e = hipSuccess;
    const unsigned g = (unsigned)std::min<long long>(((long long)n_tiles + 3) / 4, 8192);
    for (int attempt = 0; attempt < 2 && e == hipSuccess && 2LL * h_ok < n_tiles; ++attempt) {
        run = attempt == 0 ? kXwRunShort : kXwRunLong;
        e = hipMemsetAsync(d_ok, 0, 8 * sizeof(int), ctx->stream);
        if (e != hipSuccess) break;
        static_assert(kXwRuns * kXwRunShort <= 256, "8-bit positions");
        if (run == kXwRunShort)
            hipLaunchKernelGGL(xw_plan_kernel<unsigned char>, dim3(g), dim3(256), 0, ctx->stream, (int)m->n_rows, n_tiles, run,
                               m->rowptr, m->cols, desc, (unsigned char *)lidx, d_ok);
        else
            hipLaunchKernelGGL(xw_plan_kernel<unsigned short>, dim3(g), dim3(256), 0, ctx->stream, (int)m->n_rows, n_tiles, run,
                               m->rowptr, m->cols, desc, lidx, d_ok);
        e = hipGetLastError();
        int h8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (e == hipSuccess && read_back(ctx, d_ok, sizeof(h8), h8) != PADNE_OK) e = hipErrorUnknown;
        h_ok = h8[0] + h8[1] + h8[2] + h8[3] + h8[4] + h8[5] + h8[6] + h8[7];
    }
    pool_free(ctx, d_ok);
    if (ctx->opt.verbose_xw)
        fprintf(stderr, "[spmv] x-window plan: %d of %d tiles qualify with runs of %d\n", h_ok, n_tiles, run);
    if (e != hipSuccess || 2LL * h_ok < n_tiles) {     // fewer than half of the tiles qualify: not worth the extra array
        pool_free(owner, desc);
        pool_free(owner, lidx);
        if (e != hipSuccess) {
            set_error("x-window plan failed: %s", hipGetErrorString(e));
            return PADNE_E_HIP;
        }
        return PADNE_OK;
    }
    m->xw_run = run;
    m->xw_desc = desc;
    m->xw_lidx = lidx;
    m->xw_state = 1;
    return PADNE_OK;
}

// minimum over the 64 lanes of a wave (all active), as a scalar: four shifts within the rows of 16 lanes, the rows' last lanes
// handed on (row_bcast:15, row_bcast:31), lane 63 read -- seven instructions where six rounds of __shfl_xor are thirty
__device__ __forceinline__ int wave_min_i32(int v) {
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x111, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x112, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x114, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x118, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x142, 0xa, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x143, 0xc, 0xf, false));
    return __builtin_amdgcn_readlane(v, 63);
}

// The wide plan: greedy cover of a tile's columns by up to twelve runs of 20; every tile decides for itself (no count comes
// back to the host: the plan is built on the second stream beside the Galerkin product, a look at the host there would
// hold the main chain up), tiles that need more runs -- the two mesh lines a tile at a line's end touches -- keep the
// gather path inside the same launch.
__global__ __launch_bounds__(256) void xw_plan_wide_kernel(int n_rows, int n_wtiles, const int *__restrict__ rowptr,
                                                           const int *__restrict__ cols, int *__restrict__ desc,
                                                           unsigned char *__restrict__ lidx) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long W = (long long)gridDim.x * 4, gw = (long long)blockIdx.x * 4 + w;
    for (long long wt = gw; wt < n_wtiles; wt += W) {
        const int row0 = (int)wt * 64;
        const int row1 = min(row0 + 64, n_rows);
        const int k0 = rowptr[row0], k1 = rowptr[row1];
        constexpr int kReg = 8;
        const bool in_regs = k1 - k0 <= 64 * kReg;      // wave-uniform
        int creg[kReg];
#pragma unroll
        for (int j = 0; j < kReg; ++j) {
            const int e = k0 + lane + 64 * j;
            creg[j] = (in_regs && e < k1) ? cols[e] : 0x7fffffff;
        }
        int start[kXwRunsWide];
        int bound = -1;
        bool fits = true;
#pragma unroll
        for (int q = 0; q <= kXwRunsWide; ++q) {
            int mn = 0x7fffffff;
            if (in_regs) {
#pragma unroll
                for (int j = 0; j < kReg; ++j)
                    if (creg[j] > bound && creg[j] < mn) mn = creg[j];
            } else {
                for (int e = k0 + lane; e < k1; e += 64) {
                    const int c = cols[e];
                    if (c > bound && c < mn) mn = c;
                }
            }
            mn = wave_min_i32(mn);
            if (q == kXwRunsWide) {
                fits = mn == 0x7fffffff;
            } else {
                start[q] = mn == 0x7fffffff ? (q > 0 ? start[q - 1] : 0) : mn;
                if (mn != 0x7fffffff) bound = mn + kXwRunWide - 1;
            }
        }
        if (fits) {
            if (in_regs) {
#pragma unroll
                for (int j = 0; j < kReg; ++j) {
                    const int e = k0 + lane + 64 * j;
                    if (e < k1) {
                        const int c = creg[j];
                        int pos = 0;
#pragma unroll
                        for (int q = kXwRunsWide - 1; q >= 0; --q)
                            if (c >= start[q] && c < start[q] + kXwRunWide) pos = q * kXwRunWide + (c - start[q]);
                        lidx[e] = (unsigned char)pos;
                    }
                }
            } else {
                for (int e = k0 + lane; e < k1; e += 64) {
                    const int c = cols[e];
                    int pos = 0;
#pragma unroll
                    for (int q = kXwRunsWide - 1; q >= 0; --q)
                        if (c >= start[q] && c < start[q] + kXwRunWide) pos = q * kXwRunWide + (c - start[q]);
                    lidx[e] = (unsigned char)pos;
                }
            }
        }
        if (lane < kXwDescWide) {
            int v = 0;
#pragma unroll
            for (int q = 0; q < kXwRunsWide; ++q) v = lane == q ? start[q] : v;
            if (lane == kXwDescWide - 1) v = fits ? 1 : 0;
            desc[(size_t)wt * kXwDescWide + lane] = v;
        }
    }
}

int csr_build_xw_plan_wide(padne_ctx *ctx, padne_csr *m, int grid_cap) {
    static_assert(kXwRunsWide * kXwRunWide <= 256, "8-bit positions");
    if (m->xw_state == 1 || m->n_rows < 65536 || m->vals32 == nullptr || ctx->opt.no_xwindow)
        return PADNE_OK;
    padne_ctx *owner = m->owner ? m->owner : ctx;
    const int n_tiles = (int)((m->n_rows + 63) / 64);
    int4 *desc = (int4 *)pool_alloc(owner, sizeof(int) * kXwDescWide * (size_t)n_tiles);
    unsigned short *lidx = (unsigned short *)pool_alloc(owner, (size_t)m->nnz + kPadNnz);      // (bytes: one per entry)
    if (!desc || !lidx) {
        pool_free(owner, desc);
        pool_free(owner, lidx);
        return PADNE_E_NOMEM;
    }
    // (positions of the tiles that keep the gather path, and of the padding, are never read as positions; zeroed so that
    //  a stray word of a pass's last load is a valid index)
    PADNE_HIP_CHECK(hipMemsetAsync(lidx, 0, (size_t)m->nnz + kPadNnz, ctx->stream));
    const unsigned g = (unsigned)std::min<long long>(((long long)n_tiles + 3) / 4, grid_cap > 0 ? grid_cap : 8192);
    hipLaunchKernelGGL(xw_plan_wide_kernel, dim3(g), dim3(256), 0, ctx->stream, (int)m->n_rows, n_tiles, m->rowptr, m->cols,
                       (int *)desc, (unsigned char *)lidx);
    PADNE_HIP_CHECK(hipGetLastError());
    if (ctx->opt.verbose_xw) {            // (diagnostics only: this look at the host stalls the stream)
        std::vector<int> h((size_t)n_tiles * kXwDescWide);
        PADNE_HIP_CHECK(hipMemcpyAsync(h.data(), desc, sizeof(int) * h.size(), hipMemcpyDeviceToHost, ctx->stream));
        PADNE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        long long ok = 0;
        for (int t = 0; t < n_tiles; ++t) ok += h[(size_t)t * kXwDescWide + kXwDescWide - 1];
        fprintf(stderr, "[spmv] wide x-window plan: %lld of %d tiles qualify with %d runs of %d\n", ok, n_tiles, kXwRunsWide, kXwRunWide);
    }
    m->xw_run = kXwRunWide;
    m->xw_nruns = kXwRunsWide;
    m->xw_desc = desc;
    m->xw_lidx = lidx;
    m->xw_state = 1;
    return PADNE_OK;
}

// ---- interior / boundary tiles of a row-partitioned operator -------------------------------------------------------
// flag[t] = 1 if tile t reads an exchange slot (a column >= n_owned), one wave per tile
__global__ __launch_bounds__(256) void split_flag_kernel(int n_rows, int n_wtiles, int n_owned, const int *__restrict__ rowptr,
                                                         const int *__restrict__ cols, int *__restrict__ flag) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (long long wt = (long long)blockIdx.x * 4 + w; wt < n_wtiles; wt += (long long)gridDim.x * 4) {
        const int row0 = (int)wt * 64, row1 = min(row0 + 64, n_rows);
        const int k0 = rowptr[row0], k1 = rowptr[row1];
        int far = 0;
        for (int e = k0 + lane; e < k1; e += 64) far |= cols[e] >= n_owned ? 1 : 0;
        far = __any(far) ? 1 : 0;
        if (lane == 0) flag[wt] = far;
    }
}

// interior tiles to the front (in order), boundary tiles behind them (in order): pos = exclusive scan of the flags
__global__ void split_scatter_kernel(int n_wtiles, const int *__restrict__ flag, const int *__restrict__ pos, int n_bnd,
                                     int *__restrict__ tiles) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_wtiles) return;
    const int n_int = n_wtiles - n_bnd;
    if (flag[t]) tiles[n_int + pos[t]] = t;
    else tiles[t - pos[t]] = t;
}

int csr_build_split_plan(padne_ctx *ctx, padne_csr *m, long long n_owned) {
    if (m->split_state != 0) return PADNE_OK;
    m->split_state = -1;
    // (only where the exchange really runs beside the interior tiles: otherwise the second launch is pure overhead)
    if (m->n_cols <= n_owned || m->n_rows < 64 * 64 || ctx->opt.no_split || !comm_exchange_overlaps(ctx))
        return PADNE_OK;
    padne_ctx *owner = m->owner ? m->owner : ctx;
    const int n_tiles = (int)((m->n_rows + 63) / 64);
    int *tiles = (int *)pool_alloc(owner, sizeof(int) * (size_t)n_tiles);
    int *flag = (int *)pool_alloc(ctx, sizeof(int) * ((size_t)n_tiles + 1));
    int *pos = (int *)pool_alloc(ctx, sizeof(int) * ((size_t)n_tiles + 1));
    int rc = (!tiles || !flag || !pos) ? PADNE_E_NOMEM : PADNE_OK;
    int64_t n_bnd = 0;
    if (rc == PADNE_OK) {
        const unsigned g = (unsigned)std::min<long long>(((long long)n_tiles + 3) / 4, 4096);
        hipLaunchKernelGGL(split_flag_kernel, dim3(g), dim3(256), 0, ctx->stream, (int)m->n_rows, n_tiles, (int)n_owned, m->rowptr,
                           m->cols, flag);
        rc = exclusive_scan_i32(ctx, flag, pos, n_tiles, &n_bnd);
    }
    if (rc == PADNE_OK) {
        hipLaunchKernelGGL(split_scatter_kernel, dim3((unsigned)((n_tiles + 255) / 256)), dim3(256), 0, ctx->stream, n_tiles,
                           (const int *)flag, (const int *)pos, (int)n_bnd, tiles);
        if (hipGetLastError() != hipSuccess) rc = PADNE_E_HIP;
    }
    pool_free(ctx, flag);
    pool_free(ctx, pos);
    if (rc != PADNE_OK) {
        pool_free(owner, tiles);
        return rc;
    }
    m->split_tiles = tiles;
    m->split_n_bnd = (int)n_bnd;
    m->split_n_int = n_tiles - (int)n_bnd;
    m->split_state = 1;
    return PADNE_OK;
}

__global__ void f32_copy_kernel(long long n, const double *__restrict__ src, float *__restrict__ dst) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        dst[i] = (float)src[i];
}

// single-precision copies of the values and of 1/diag (kept with the matrix, freed with it).  The copies come from
// the pool of the context whose stream writes them first (pool_free of the matrix' owner finds either pool).
int csr_build_f32(padne_ctx *ctx, padne_csr *m) {
    padne_ctx *owner = ctx->is_aux ? ctx : (m->owner ? m->owner : ctx);
    if (m->vals32 == nullptr) {          // (else: an earlier solve's copy, or the one the strength pass of the setup has written)
        m->vals32 = (float *)pool_alloc(owner, sizeof(float) * ((size_t)m->nnz + kPadNnz));   // padded like vals
        if (m->vals32 == nullptr) return PADNE_E_NOMEM;
        if (m->nnz > 0)
            hipLaunchKernelGGL(f32_copy_kernel, dim3((unsigned)std::min<long long>((m->nnz + 255) / 256, 8192)), dim3(256), 0,
                               ctx->stream, (long long)m->nnz, m->vals, m->vals32);
    }
    if (m->dinv != nullptr && m->dinv32 == nullptr) {
        m->dinv32 = (float *)pool_alloc(owner, sizeof(float) * (size_t)(m->n_rows > 0 ? m->n_rows : 1));
        if (m->dinv32 == nullptr) return PADNE_E_NOMEM;
        if (m->n_rows > 0)
            hipLaunchKernelGGL(f32_copy_kernel, dim3((unsigned)std::min<long long>((m->n_rows + 255) / 256, 8192)),
                               dim3(256), 0, ctx->stream, (long long)m->n_rows, m->dinv, m->dinv32);
    }
    PADNE_HIP_CHECK(hipGetLastError());
    return PADNE_OK;
}

int launch_spmv(padne_ctx *ctx, const padne_csr *m, const double *x, double *y,
                const double *dot_with, double *partials, const int32_t *done_flag) {
    return launch_spmv_mode(ctx, m, dot_with != nullptr ? SPMV_DOT : SPMV_PLAIN, x, y, dot_with, partials,
                            done_flag, nullptr, nullptr, 0.0);
}

// ---- 1/diag ---------------------------------------------------------------------------------
__global__ void csr_dinv_kernel(int n_rows, const int *__restrict__ rowptr,
                                const int *__restrict__ cols, const double *__restrict__ vals,
                                double *__restrict__ dinv) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    double d = 0.0;
    const int k0 = rowptr[r], k1 = rowptr[r + 1];
    // eight columns per look (two unaligned 16-byte loads; behind the row's end lies the next row or the zero padding of the
    // array, and is not used): entry by entry the thirteen entries of a first-coarse-level row were thirteen dependent steps
    struct __attribute__((packed, aligned(4))) I4u { int x, y, z, w; };
    for (int kb = k0; kb < k1; kb += 8) {
        const I4u a = *reinterpret_cast<const I4u *>(cols + kb), b = *reinterpret_cast<const I4u *>(cols + kb + 4);
        const int c[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (kb + u < k1 && c[u] == r) d += vals[kb + u];
    }
    dinv[r] = 1.0 / d;
}

// the same with a wave per row, for the small operators with long rows at the bottom of a hierarchy (a thread walking a
// row of a hundred entries on its own: 28-39 us for the two coarsest levels of config C4)
__global__ __launch_bounds__(256) void csr_dinv_wpr_kernel(int n_rows, const int *__restrict__ rowptr, const int *__restrict__ cols,
                                                           const double *__restrict__ vals, double *__restrict__ dinv) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    double d = 0.0;
    for (int k = rowptr[r] + lane; k < rowptr[r + 1]; k += 64)
        if (cols[k] == r) d += vals[k];           // columns are unique in a row: exactly one lane finds the diagonal
    d = wave_sum(d);                              // (zeros from the other lanes: the sum is that entry, bit for bit)
    if (lane == 0) dinv[r] = 1.0 / d;
}

int csr_build_dinv(padne_ctx *ctx, padne_csr *m) {
    if (m->dinv != nullptr) return PADNE_OK;
    PADNE_REQUIRE(m->n_rows <= m->n_cols, "Jacobi needs a diagonal");
    m->dinv = (double *)pool_alloc(m->owner ? m->owner : ctx, sizeof(double) * (size_t)(m->n_rows > 0 ? m->n_rows : 1));
    if (m->dinv == nullptr) return PADNE_E_NOMEM;
    if (m->n_rows > 0) {
        const int bs = 256;
        if (m->n_rows <= 65536 && m->nnz >= 24 * m->n_rows)
            hipLaunchKernelGGL(csr_dinv_wpr_kernel, dim3((unsigned)((m->n_rows + 3) / 4)), dim3(bs), 0, ctx->stream, (int)m->n_rows,
                               m->rowptr, m->cols, m->vals, m->dinv);
        else
            hipLaunchKernelGGL(csr_dinv_kernel, dim3((unsigned)((m->n_rows + bs - 1) / bs)), dim3(bs), 0,
                               ctx->stream, (int)m->n_rows, m->rowptr, m->cols, m->vals, m->dinv);
        PADNE_HIP_CHECK(hipGetLastError());
    }
    return PADNE_OK;
}

}  // namespace padne
