// Smoothed-aggregation algebraic multigrid, built and applied entirely on the device, used as the
// preconditioner of the CG solve that replaces scipy's spsolve (solver.py:773).  No reference
// counterpart (the reference factorises with SuperLU); it only changes how fast
// ||b - A x|| <= rtol ||b|| is reached, not what is solved.
//
// Setup per level (all deterministic -- priorities are hashes of the index, sums have a fixed order):
//   strength   j is a strong neighbour of i  <=>  a_ij^2 >= theta^2 a_ii a_jj          (theta = 0.08)
//   aggregate  distance-2 maximal independent set of the strength graph (MIS-k rounds on unique 32-bit
//              priority words, two neighbour-max passes per round) -> roots; every other vertex joins the
//              aggregate of its strongest aggregated neighbour (two passes); vertices without strong
//              neighbours become singletons
//   prolong    P = (I - omega D_F^-1 A_F) T,  T = piecewise constant, A_F = A with the weak entries lumped into
//              the diagonal,  omega = 1.5 / lambda,  lambda = Gershgorin bound of D_F^-1 A_F
//   restrict   R = P^T stored explicitly (CSR, placed in column order without a sort) so that restriction is the same SpMV kernel
//   coarse     A_c = R (A P) by two row-wise sparse products: one thread per row with a short list in LDS for the fine
//              level's A P, 16 / 32 / 64 lanes per row with a lane mask per column everywhere else, a dense LDS
//              accumulator for the long rows of the coarse levels; products are always added in generation order
// until n <= 2048, where the dense inverse is formed by a blocked Gauss-Jordan on the f64 matrix cores (SPD: no pivoting).
//
// Apply: V(1,1) cycle with damped Jacobi (first-degree Chebyshev on [lambda/10, lambda]); every
// stage is the SpMV kernel of spmv.hip with a different epilogue (residual formed from the right-hand side alone,
// restriction with the first sweep of the level below, coarse correction + post-smoothing in ONE product with
// W = P - c D^-1 A P), so the fine level costs two and a half matrix passes per CG iteration.  The cycle is a
// fixed symmetric positive definite linear operator, hence plain PCG applies.  By default it runs in
// single precision on float copies of its operators (amg_apply_f32), also for 8 / 4 / 2 interleaved right-hand
// sides at once (amg_apply_batch).
//
// Row-partitioned runs (one rank per GPU): one hierarchy over all ranks, see amg_setup_dist below.
#include "common.hpp"

#include <functional>
#include <type_traits>
#include <memory>

#include <algorithm>
#include <chrono>
#include <math.h>
#include <stdlib.h>
#include <string.h>

namespace padne {

struct AmgLevel {
    const padne_csr *A = nullptr;   // level matrix (level 0 is borrowed)
    padne_csr *A_owned = nullptr;
    padne_csr *P = nullptr;         // n_l x n_{l+1}
    padne_csr *R = nullptr;         // n_{l+1} x n_l
    padne_csr *W = nullptr;         // P - c D^-1 A P, single-precision values only (the fine level of the float cycle)
    double lambda = 2.0;            // Gershgorin bound of D^-1 A
    double jac = 0.0;               // Jacobi damping 1/theta_c
    long long n = 0;
    double *b = nullptr, *xa = nullptr, *xb = nullptr, *tmp = nullptr;   // work vectors (levels >= 1: all; level 0: xa, tmp)
    float *b8 = nullptr, *xa8 = nullptr, *xb8 = nullptr, *tmp8 = nullptr;   // [n][8] vectors of the batched cycle
    // row-partitioned hierarchy: A is this rank's rows, columns [owned | world * halo.m exchange slots]
    HaloPlan halo;
    int32_t *export_owned = nullptr;   // device copy of the export list (levels >= 1; level 0 borrows the context's)
    std::vector<int> export_host;
    // the other ranks' rows of P for the exchanged vertices ([world * halo.m] x [n_{l+1} | world * m_{l+1}]): kept on the
    // last partitioned level, whose coarse level is the gathered one -- every rank holds the whole tail solution, so the
    // neighbours' corrected values there are computed, not exchanged (amg_apply_f32)
    padne_csr *P_halo = nullptr;
    float *e_ext = nullptr;            // [n_{l+1} | world * m_{l+1}] coarse correction incl. the other ranks' exported aggregates
};

struct Amg {
    padne_ctx *ctx = nullptr;
    std::vector<AmgLevel> levels;
    double *coarse_inv = nullptr;   // dense n_c x n_c
    int n_coarse = 0;
    // single-precision cycle: the operators of every level have float copies and all cycle vectors are float
    bool f32 = false;
    float *coarse_inv32 = nullptr;
    // row-partitioned hierarchy: below the gather level every rank holds the whole operator (`tail`, with its
    // own single-GPU hierarchy) and runs the rest of the cycle redundantly
    bool dist = false;
    padne_csr *tail = nullptr;
    int n_pad = 0;                     // longest per-rank piece of the gather level
    long long tail_n = 0, tail_off = 0;   // gathered size, first row of this rank's piece
    int *seg_off = nullptr;            // device [world + 1]: first gathered row of every rank
    double *coarse_gather = nullptr;   // [world * n_pad] padded pieces as exchanged
    double *tail_r = nullptr, *tail_z = nullptr;   // [tail_n]
    int *tail_slot_idx = nullptr;      // device [world * m]: gathered row of every exchange slot of the gather level
    double setup_seconds = 0.0;
    double operator_complexity = 0.0;
    int device = 0;
};

// strength threshold, prolongator smoothing, smoother interval, Lanczos steps: the values the parameter studies of rounds
// 1-4 settled on (DESIGN_HISTORY.md); the size the dense inverse takes comes from the context's options (tests vary it)
constexpr double kTheta = 0.08;
constexpr double kOmegaNum = 1.5;
constexpr double kChebRatio = 10.0;
constexpr int kLanczosSteps = 8;
constexpr int kMaxLevels = 16;

// unaligned multi-dword loads: global loads of 8 and 16 bytes only need their address to be a multiple of 4.  What bounds a
// kernel of gathers is the number of its memory instructions (the address unit takes about a cycle per lane and instruction
// whatever the width), so rows are read four indices / two values at a time; what lies behind a row's end is the next row
// or the zero padding of the arrays, and is not used.
struct __attribute__((packed, aligned(4))) I4u { int x, y, z, w; };
struct __attribute__((packed, aligned(4))) I2u { int x, y; };
struct __attribute__((packed, aligned(4))) D2u { double x, y; };
struct __attribute__((packed, aligned(4))) L2u { long long x, y; };
__device__ __forceinline__ int4 load_i4_unaligned(const int *p) {
    const I4u v = *reinterpret_cast<const I4u *>(p);
    return make_int4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ double2 load_d2_unaligned(const double *p) {
    const D2u v = *reinterpret_cast<const D2u *>(p);
    return make_double2(v.x, v.y);
}

// Inclusive prefix sum over groups of LANES (16, 32 or 64) lanes, all lanes active: four shifts within the rows of 16 lanes,
// then the last lane of a row added to the whole next row (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3)
// -- six vector instructions instead of five rounds of index arithmetic, ds_bpermute and select.
template <int LANES>
__device__ __forceinline__ int scan_incl_lanes(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
    if (LANES >= 32) x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
    if (LANES == 64) x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
    return x;
}

// ---- XCD-aware row order --------------------------------------------------------------------------------------------
// Workgroups are dealt to the 8 XCDs round-robin (b, b + 8, b + 16, ... share an XCD and its private 4 MiB L2).  A kernel
// that walks the rows in workgroup order therefore shows every XCD rows from all over the matrix, and whatever the rows
// of a mesh neighbourhood share -- the rows of P around an aggregate, the slots of A P a coarse row gathers, 1/diag of
// the neighbours -- is fetched by up to 8 L2s and has left each of them long before the neighbouring scan line comes by
// (rocprofv3 FETCH_SIZE: 19x the algorithmic bytes over the setup of round 2).  As in the SpMV kernel (spmv.hip) every XCD
// gets one contiguous eighth of the rows and sweeps it front to back: xcd_bid maps the hardware workgroup index to the
// logical one (a bijection; the last gridDim % 8 workgroups keep their place).
__device__ __forceinline__ unsigned xcd_bid() {
    const unsigned b = blockIdx.x, per = gridDim.x >> 3;
    if (b >= (per << 3)) return b;
    return (b & 7u) * per + (b >> 3);
}
// the same for kernels whose waves stride over 64-row tiles: [first, last) tile range of this workgroup's XCD, the wave's
// start inside it and its stride (gridDim a multiple of 8, else one slab)
struct XcdSweep { long long t0, t1, stride; };
__device__ __forceinline__ XcdSweep xcd_sweep(const long long n_tiles, const int waves_per_block, const int w) {
    const unsigned G = gridDim.x;
    const unsigned nslab = (G % kNumXcd == 0) ? kNumXcd : 1;
    const unsigned slab = blockIdx.x % nslab;
    XcdSweep sw;
    const long long s0 = (long long)slab * n_tiles / nslab;
    sw.t1 = (long long)(slab + 1) * n_tiles / nslab;
    sw.t0 = s0 + (long long)(blockIdx.x / nslab) * waves_per_block + w;
    sw.stride = (long long)(G / nslab) * waves_per_block;
    return sw;
}

// ---- small kernels -----------------------------------------------------------------------------


// 32-bit competition word of the MIS rounds: 0 = decided (covered), kMisRoot = root, otherwise a priority that
// is unique per vertex (bijective 31-bit mix of the index, + 1)
constexpr unsigned int kMisRoot = 0xffffffffu;
__device__ __forceinline__ unsigned int prio32_of(int i) {
    unsigned int h = (unsigned int)i & 0x7fffffffu;
    h = (h * 0x5bd1e995u) & 0x7fffffffu;      // odd multiplier: a bijection modulo 2^31
    h ^= h >> 15;
    h = (h * 0x2c1b3c6du) & 0x7fffffffu;
    h ^= h >> 13;
    h = (h * 0x297a2d39u) & 0x7fffffffu;
    h ^= h >> 16;
    return h + 1u;
}

__device__ __forceinline__ bool strong(double a, double di, double dj, double theta2) {
    return a * a * di * dj >= theta2;   // a_ij^2 >= theta^2 a_ii a_jj with d = 1/a_ii
}

// Strength graph S on the pattern of A: scol[k] = column of entry k if it is a strong off-diagonal coupling,
// otherwise the row itself (a self loop is neutral for the neighbour maxima and is skipped by the joins).  No
// compaction, no second array of weights (the joins read |a_ij| from A): one coalesced pass over the matrix.
// Wave-private tiles of 64 rows as in the SpMV kernel; the row of every staged element comes from LDS.
// Position of column `c` inside the staged x-window runs of a tile (spmv.hip, csr_build_xw_plan): the runs start at
// d.x, d.y, d.z and hold `run` consecutive columns each.
__device__ __forceinline__ int xw_position(const int4 d, const int run, const int c) {
    int pos = 0;                       // where runs overlap the FIRST run wins, exactly as xw_plan_kernel numbers xw_lidx
    if (c >= d.z && c < d.z + run) pos = 2 * run + (c - d.z);
    if (c >= d.y && c < d.y + run) pos = run + (c - d.y);
    if (c >= d.x && c < d.x + run) pos = c - d.x;
    return pos;
}

// With an x-window plan of A (xw_desc / xw_lidx, runs of at most 85 columns so that a position fits a byte) the pass
// also writes spos[k]: the window position of scol[k].  The neighbour-maximum passes of the independent-set rounds
// then read one byte per entry and take the neighbours' words from LDS-staged runs instead of gathering them.
__global__ __launch_bounds__(256) void strength_mark(int n, int n_wtiles, const int *__restrict__ rowptr,
                                                     const int *__restrict__ cols, const double *__restrict__ vals,
                                                     const double *__restrict__ dinv, double theta2,
                                                     int *__restrict__ scol, double *__restrict__ bound_partial,
                                                     const int4 *__restrict__ xw_desc = nullptr,
                                                     const unsigned char *__restrict__ xw_lidx = nullptr,
                                                     const int xw_run = 0, unsigned char *__restrict__ spos = nullptr,
                                                     float *__restrict__ vals32 = nullptr) {
    // vals32 (optional): the single-precision copy of the values the multigrid cycle multiplies with, written while the
    // values stream by (the copy as a pass of its own read the fine matrix once more: 140 us)
    // bound_partial (optional): per-workgroup maxima of the Gershgorin bounds of D^-1 A (second row of kMaxPartials) and of D_F^-1 A_F, the filtered operator
    // the prolongator is smoothed with -- the same sums, in the same order, as gershgorin_filtered_kernel forms them
    // one lane per row, taken here from the values this pass streams anyway (a separate pass over A cost 0.4 ms).
    constexpr int CH = 512;
    __shared__ unsigned char rid_all[4 * CH];      // bits 0-5: lane of the element's row, bit 6: diagonal, bit 7: strong
    __shared__ double pv_all[4 * CH];
    __shared__ double red[4], red2[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned char *rid = rid_all + w * CH;
    double *pv = pv_all + w * CH;
    const XcdSweep sw = xcd_sweep(n_wtiles, 4, w);
    double my_max = 0.0, my_plain = 0.0;
    for (long long wt = sw.t0; wt < sw.t1; wt += sw.stride) {
        const int row0 = (int)wt * 64;
        const int row1 = min(row0 + 64, n);
        const int r = row0 + lane;
        int rs = 0, re = 0;
        double di = 0.0;
        if (r < row1) {
            rs = rowptr[r];
            re = rowptr[r + 1];
            di = dinv[r];
        }
        const int k0 = __shfl(rs, 0, 64);
        const int k1 = __shfl(re, row1 - row0 - 1, 64);
        double dF = r < row1 ? 1.0 / di : 1.0, off_strong = 0.0, off_all = 0.0, abs_all = 0.0;
        int4 dsc = make_int4(0, 0, 0, 0);
        if (spos != nullptr) dsc = xw_desc[wt];
        const bool windowed = spos != nullptr && dsc.w != 0;
        const int self_pos = windowed ? xw_position(dsc, xw_run, r) : 0;
        for (int base = k0; base < k1; base += CH) {
            const int lo = max(rs, base), hi = min(re, base + CH);
            for (int k = lo; k < hi; ++k) rid[k - base] = (unsigned char)lane;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < CH / 64; ++q) {
                const int e = base + lane + 64 * q;
                const int rl = e < k1 ? rid[e - base] : 0;
                const double dr = __shfl(di, rl, 64);          // all lanes take part: the source lane may be past k1
                const int sp = __shfl(self_pos, rl, 64);
                if (e < k1) {
                    const int i = row0 + rl;
                    const int c = cols[e];
                    const double v = vals[e];
                    if (vals32 != nullptr) vals32[e] = (float)v;
                    const bool st = c != i && strong(v, dr, dinv[c], theta2);
                    scol[e] = st ? c : i;
                    if (windowed) spos[e] = st ? xw_lidx[e] : (unsigned char)sp;
                    if (bound_partial != nullptr) {
                        rid[e - base] = (unsigned char)(rl | (c == i ? 0x40 : 0) | (st ? 0x80 : 0));
                        pv[e - base] = v;
                    }
                }
            }
            if (bound_partial != nullptr) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                for (int k = lo; k < hi; ++k) {                // one lane per row, in CSR order
                    const unsigned char f = rid[k - base];
                    const double v = pv[k - base];
                    abs_all += fabs(v);                        // every entry, diagonal included: gershgorin_kernel's sum
                    if (f & 0x40) continue;
                    off_all += fabs(v);
                    if (f & 0x80) off_strong += fabs(v);
                    else dF += v;
                }
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        if (bound_partial != nullptr && r < row1) {
            double off = off_strong;
            if (!(dF * di > 0.05)) {                           // same rule as prolong_fill: such a row is not filtered
                dF = 1.0 / di;
                off = off_all;
            }
            const double sgm = (off + fabs(dF)) / fabs(dF);
            my_max = sgm > my_max ? sgm : my_max;
            const double plain = abs_all * fabs(di);           // bound of D^-1 A itself, as gershgorin_kernel forms it
            my_plain = plain > my_plain ? plain : my_plain;
        }
    }
    if (bound_partial != nullptr) {
        for (int o = 32; o > 0; o >>= 1) {
            my_max = fmax(my_max, __shfl_down(my_max, o, 64));
            my_plain = fmax(my_plain, __shfl_down(my_plain, o, 64));
        }
        if (lane == 0) {
            red[w] = my_max;
            red2[w] = my_plain;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            bound_partial[blockIdx.x] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
            bound_partial[kMaxPartials + blockIdx.x] = fmax(fmax(red2[0], red2[1]), fmax(red2[2], red2[3]));
        }
    }
}

// out[i] = max(in[i], max over the strong neighbours j of in[j]).  Same wave-private, lane-consecutive
// streaming as the SpMV kernel (spmv.hip): 64 rows per wave, neighbour values parked in LDS, one lane per row.
template <typename T>
__global__ __launch_bounds__(256) void nbr_max(int n, int n_wtiles, const int *__restrict__ srow,
                                               const int *__restrict__ scol, const T *__restrict__ in,
                                               T *__restrict__ out) {
    constexpr int CH = 512;
    __shared__ T park_all[4 * CH];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    T *park = park_all + w * CH;
    // (every XCD sweeps one contiguous eighth of the tiles, as the SpMV does: the words the rows of a tile gather are those of
    // the tiles next to it, still in this XCD's L2)
    const XcdSweep sw = xcd_sweep(n_wtiles, 4, w);
    for (long long wt = sw.t0; wt < sw.t1; wt += sw.stride) {
        const int row0 = (int)wt * 64;
        const int row1 = min(row0 + 64, n);
        const int r = row0 + lane;
        int rs = 0, re = 0;
        T m = 0;
        if (r < row1) {
            rs = srow[r];
            re = srow[r + 1];
            m = in[r];
        }
        const int k0 = __shfl(rs, 0, 64);
        const int k1 = __shfl(re, row1 - row0 - 1, 64);
        for (int base = k0; base < k1; base += CH) {
#pragma unroll
            for (int j = 0; j < CH / 64; ++j) {
                const int e = base + lane + 64 * j;
                park[lane + 64 * j] = (e < k1) ? in[scol[e]] : (T)0;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const int lo = max(rs, base), hi = min(re, base + CH);
            for (int k = lo; k < hi; ++k) {
                const T v = park[k - base];
                m = v > m ? v : m;
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        if (r < row1) out[r] = m;
    }
}

// Wave-per-row form for the small levels with long rows (level 3 of config C4: 14 k rows of 50-100 entries).  The tile
// kernel above gives a wave 64 rows, i.e. eight passes of 512 entries one after the other, and all of a 14 k-row level is
// 224 waves: 23 us per pass, forty passes per aggregation.  Here the lanes stride over one row and fold with shuffles.
template <typename T>
__global__ __launch_bounds__(256) void nbr_max_wpr(int n, const int *__restrict__ srow, const int *__restrict__ scol,
                                                   const T *__restrict__ in, T *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const int rs = srow[r], re = srow[r + 1];
    T m = in[r];
    for (int k = rs + lane; k < re; k += 64) {
        const T v = in[scol[k]];
        m = v > m ? v : m;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const T o = __shfl_down(m, off, 64);
        m = o > m ? o : m;
    }
    if (lane == 0) out[r] = m;
}

__device__ __forceinline__ bool mis_decide_one(int i, unsigned int top, unsigned int *__restrict__ word,
                                               signed char *__restrict__ state) {
    // returns true if vertex i is still undecided after this round
    const unsigned int mine = word[i];
    if (top == kMisRoot) {
        state[i] = 2;
        word[i] = 0u;
        return false;
    }
    if (top == mine) {
        state[i] = 1;
        word[i] = kMisRoot;
        return false;
    }
    return true;
}

// The same pass over a matrix with an x-window plan: a tile's neighbours live in (at most) three runs of consecutive
// vertices, which are staged in LDS with coalesced loads; an entry is then one byte (spos, written by strength_mark)
// instead of a 4-byte column and a scattered 4-byte gather.  Tiles without a plan take the gather path above.
// DECIDE: the second pass of a round ends with the decision of the round (mis_decide) while the two-hop maximum is in
// a register: `out` is not written, the words / states are updated in place (the pass itself reads only `in`, the
// one-hop maxima), the vertices still open are counted.
template <typename T, bool DECIDE>
__global__ __launch_bounds__(256) void nbr_max_xw(int n, int n_wtiles, const int *__restrict__ srow,
                                                  const int *__restrict__ scol, const unsigned char *__restrict__ spos,
                                                  const int4 *__restrict__ xw_desc, const int run,
                                                  const T *__restrict__ in, T *__restrict__ out,
                                                  unsigned int *__restrict__ word, signed char *__restrict__ state,
                                                  int *__restrict__ undecided) {
    constexpr int CH = 512;
    __shared__ int open_red[4];
    int open = 0;
    __shared__ T park_all[4 * CH];
    __shared__ T xs_all[4 * 3 * 88];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    T *park = park_all + w * CH;
    T *xs = xs_all + w * 3 * 88;
    const XcdSweep sw = xcd_sweep(n_wtiles, 4, w);      // (an eighth of the tiles per XCD: neighbouring tiles stage the same runs)
    for (long long wt = sw.t0; wt < sw.t1; wt += sw.stride) {
        const int row0 = (int)wt * 64;
        const int row1 = min(row0 + 64, n);
        const int r = row0 + lane;
        int rs = 0, re = 0;
        T m = 0;
        if (r < row1) {
            rs = srow[r];
            re = srow[r + 1];
            m = in[r];
        }
        const int4 d = xw_desc[wt];
        const int k0 = __shfl(rs, 0, 64);
        const int k1 = __shfl(re, row1 - row0 - 1, 64);
        if (d.w != 0) {
            const int st[3] = {d.x, d.y, d.z};
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int a = st[q] + lane, b = st[q] + 64 + lane;
                xs[q * run + lane] = a < n ? in[a] : (T)0;
                if (lane < run - 64) xs[q * run + 64 + lane] = b < n ? in[b] : (T)0;
            }
            const int top = 3 * run - 1;
            for (int base = k0 & ~3; base < k1; base += CH) {
                unsigned int cw[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int e = base + 4 * lane + 256 * j;
                    cw[j] = e < k1 ? *reinterpret_cast<const unsigned int *>(spos + e) : 0u;     // (spos is padded like cols)
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int t = 0; t < 4; ++t) park[4 * lane + 256 * j + t] = xs[min((int)((cw[j] >> (8 * t)) & 255u), top)];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                const int lo = max(rs, base), hi = min(re, base + CH);
                for (int k = lo; k < hi; ++k) {
                    const T v = park[k - base];
                    m = v > m ? v : m;
                }
                asm volatile("" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
        } else {
            for (int base = k0; base < k1; base += CH) {
#pragma unroll
                for (int j = 0; j < CH / 64; ++j) {
                    const int e = base + lane + 64 * j;
                    park[lane + 64 * j] = (e < k1) ? in[scol[e]] : (T)0;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                const int lo = max(rs, base), hi = min(re, base + CH);
                for (int k = lo; k < hi; ++k) {
                    const T v = park[k - base];
                    m = v > m ? v : m;
                }
                asm volatile("" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (r < row1) {
            if (DECIDE) {
                if (state[r] == 0 && mis_decide_one(r, (unsigned int)m, word, state)) ++open;
            } else {
                out[r] = m;
            }
        }
    }
    if (DECIDE) {
        for (int off = 32; off > 0; off >>= 1) open += __shfl_down(open, off, 64);
        if (lane == 0) open_red[w] = open;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int t = open_red[0] + open_red[1] + open_red[2] + open_red[3];
            if (t) atomicAdd(undecided, t);
        }
    }
}

// One round of the distance-2 independent set on the competition words (Bell, Dalton, Olson: MIS-k).  m2 is the
// maximum of the words within two strong hops.  An undecided vertex that sees a root is covered; one that sees
// nothing above its own priority becomes a root (the vertices around it are covered in the next round).
// start of a level's aggregation in one launch: the competition words, the states, the (pre-zeroed) counters of the rounds
// and the padding behind the window positions -- three memsets and a kernel before
__global__ void mis_init_words(int n, unsigned int *__restrict__ word, signed char *__restrict__ state,
                               int *__restrict__ counters, int n_counters, unsigned char *__restrict__ pad, int n_pad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        word[i] = prio32_of(i);
        state[i] = 0;
    }
    if (i < n_counters) counters[i] = 0;
    if (pad != nullptr && i < n_pad) pad[i] = 0;
}

// The LAST rounds of a level in one launch: once a few thousand vertices are open, a round is two launches of a handful
// of workgroups and every third round a look at the host -- half a dozen rounds per level, four levels per setup.  One
// workgroup takes the list through all remaining rounds: two-hop maxima of the open vertices (eight lanes per vertex as in
// mis_two_hop_max), barrier, decisions (mis_decide_one) and the next list, barrier.  Same words, same decisions.
constexpr int kMisTailCap = 512;
__global__ __launch_bounds__(1024) void mis_tail_rounds(const int *__restrict__ cnt_ptr, int *list_a, int *list_b,
                                                        const int *__restrict__ srow, const int *__restrict__ scol,
                                                        unsigned int *word, signed char *state, unsigned int *m2,
                                                        int *__restrict__ count_out, const int max_rounds) {
    __shared__ int s_next;
    int cnt = *cnt_ptr;
    if (threadIdx.x == 0) s_next = 0;
    __syncthreads();
    const int q = threadIdx.x & 7, slot = threadIdx.x >> 3;      // 128 vertices per pass, eight lanes each
    for (int round = 0; round < max_rounds && cnt > 0; ++round) {
        for (int t0 = 0; t0 < cnt; t0 += 128) {
            const int t = t0 + slot;
            const bool live = t < cnt;
            unsigned int m = 0u;
            if (live) {
                const int i = list_a[t];
                m = word[i];
                const int a1 = srow[i + 1];
                for (int a = srow[i] + q; a < a1; a += 8) {
                    const int j = scol[a];
                    if (j == i) continue;
                    const unsigned int wj = word[j];
                    m = wj > m ? wj : m;
                    for (int b = srow[j]; b < srow[j + 1]; ++b) {
                        const unsigned int wk = word[scol[b]];
                        m = wk > m ? wk : m;
                    }
                }
            }
#pragma unroll
            for (int d = 1; d < 8; d <<= 1) {
                const unsigned int o = (unsigned int)__shfl_xor((int)m, d, 64);
                m = o > m ? o : m;
            }
            if (live && q == 0) m2[t] = m;
        }
        __syncthreads();                                  // every maximum is taken from the words of the round's start
        for (int t = threadIdx.x; t < cnt; t += 1024) {
            const int i = list_a[t];
            if (mis_decide_one(i, m2[t], word, state)) list_b[atomicAdd(&s_next, 1)] = i;
        }
        __syncthreads();
        cnt = s_next;
        __syncthreads();
        if (threadIdx.x == 0) s_next = 0;
        int *tmp = list_a;
        list_a = list_b;
        list_b = tmp;
        __syncthreads();
    }
    if (threadIdx.x == 0) *count_out = cnt;
}

// full round: every vertex looks at its two-hop maximum m2; *undecided = number of vertices still open
// (grid-stride, one atomic per workgroup: a per-wave atomic on one address costs 1.7 ms at N = 10 M)
__global__ __launch_bounds__(256) void mis_decide(int n, unsigned int *__restrict__ word, const unsigned int *__restrict__ m2,
                                                  signed char *__restrict__ state, int *__restrict__ undecided) {
    __shared__ int red[4];
    int open = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
        if (state[i] == 0 && mis_decide_one(i, m2[i], word, state)) ++open;
    for (int off = 32; off > 0; off >>= 1) open += __shfl_down(open, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = open;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t = red[0] + red[1] + red[2] + red[3];
        if (t) atomicAdd(undecided, t);
    }
}

// Late rounds touch only the vertices that are still open (a few per cent after three or four rounds, while a
// full neighbour-max pass streams the whole strength graph): compact list + direct two-hop maximum per vertex.
// Same words, same decisions as the full rounds, so the aggregates do not depend on where the switch happens.
// Every workgroup compacts one contiguous slice: count, one atomic for the base, then write.
__global__ __launch_bounds__(256) void mis_collect_open(int n, const signed char *__restrict__ state, int *__restrict__ list,
                                                        int *__restrict__ count) {
    __shared__ int red[4];
    __shared__ int base_s;
    const int per = (n + gridDim.x - 1) / gridDim.x;
    const int i0 = blockIdx.x * per, i1 = min(i0 + per, n);
    int mine = 0;
    for (int i = i0 + threadIdx.x; i < i1; i += 256) mine += state[i] == 0 ? 1 : 0;
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t = red[0] + red[1] + red[2] + red[3];
        base_s = t ? atomicAdd(count, t) : 0;
    }
    __syncthreads();
    int base = base_s;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int c0 = i0; c0 < i1; c0 += 256) {
        const int i = c0 + threadIdx.x;
        const bool open = i < i1 && state[i] == 0;
        const unsigned long long mask = __ballot(open);
        __syncthreads();
        if (lane == 0) red[w] = __popcll(mask);
        __syncthreads();
        int before = 0;
        for (int q = 0; q < w; ++q) before += red[q];
        if (open) list[base + before + __popcll(mask & ((1ull << lane) - 1ull))] = i;
        base += red[0] + red[1] + red[2] + red[3];
    }
}

// (the list length is read on the device: several rounds are enqueued per host synchronisation, on a grid sized by the
// last length the host has seen; lists only shrink)
// Eight lanes per list entry: lane q walks the neighbours q, q + 8, ... (each one's row serially) and the eight maxima are
// folded by shuffles -- the chain of dependent loads of one thread per entry (row x row of them, hundreds next to a hub
// row of a coarse level) was what the short lists of the late rounds waited for.
constexpr int kHopLanes = 8;
__global__ void mis_two_hop_max(const int *__restrict__ cnt_ptr, const int *__restrict__ list, const int *__restrict__ srow,
                                const int *__restrict__ scol, const unsigned int *__restrict__ word,
                                unsigned int *__restrict__ m2) {
    const int t = (blockIdx.x * blockDim.x + threadIdx.x) / kHopLanes;
    const int q = threadIdx.x & (kHopLanes - 1);
    const bool live = t < *cnt_ptr;
    unsigned int m = 0u;
    if (live) {
        const int i = list[t];
        m = word[i];
        const int a1 = srow[i + 1];
        for (int a = srow[i] + q; a < a1; a += kHopLanes) {
            const int j = scol[a];
            if (j == i) continue;
            const unsigned int wj = word[j];
            m = wj > m ? wj : m;
            for (int b = srow[j]; b < srow[j + 1]; ++b) {
                const unsigned int wk = word[scol[b]];
                m = wk > m ? wk : m;
            }
        }
    }
#pragma unroll
    for (int d = 1; d < kHopLanes; d <<= 1) {
        const unsigned int o = (unsigned int)__shfl_xor((int)m, d, 64);
        m = o > m ? o : m;
    }
    if (live && q == 0) m2[t] = m;
}

__global__ void mis_decide_list(const int *__restrict__ cnt_ptr, const int *__restrict__ list, const unsigned int *__restrict__ m2,
                                unsigned int *__restrict__ word, signed char *__restrict__ state,
                                int *__restrict__ list_next, int *__restrict__ count_next) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    bool open = false;
    int i = 0;
    if (t < *cnt_ptr) {
        i = list[t];
        open = mis_decide_one(i, m2[t], word, state);
    }
    // one atomic per workgroup (its four waves' counts are added in LDS first): thousands of waves asking ONE address for
    // a return value are served one after the other -- 85 us for the 434 k entries of the first list of config C4
    __shared__ int wave_cnt[4], block_base;
    const unsigned long long mask = __ballot(open);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) wave_cnt[w] = __popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        block_base = t ? atomicAdd(count_next, t) : 0;
    }
    __syncthreads();
    int base = block_base;
    for (int q = 0; q < w; ++q) base += wave_cnt[q];
    if (open) list_next[base + __popcll(mask & ((1ull << lane) - 1ull))] = i;
}

__global__ void flag_state(int n, const signed char *__restrict__ state, int *__restrict__ flag, int which) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = state[i] == which ? 1 : 0;
}

__global__ void agg_from_roots(int n, const signed char *__restrict__ state, const int *__restrict__ scan,
                               int *__restrict__ agg) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) agg[i] = state[i] == 1 ? scan[i] : -1;
}

__global__ void agg_join(int n, const int *__restrict__ srow, const int *__restrict__ scol,
                         const double *__restrict__ vals, const int *__restrict__ agg_in, int *__restrict__ agg_out) {
    const int i = xcd_bid() * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int a = agg_in[i];
    if (a < 0) {
        double best = -1.0;
        const int k0 = srow[i], k1 = srow[i + 1];
        // the first eight entries together (columns and values in six loads, then eight independent looks at the
        // neighbours' aggregates); one entry after the other every entry waited for its own chain of three loads
        {
            const int4 ca = load_i4_unaligned(scol + k0), cb = load_i4_unaligned(scol + k0 + 4);
            const double2 v0 = load_d2_unaligned(vals + k0), v1 = load_d2_unaligned(vals + k0 + 2),
                          v2 = load_d2_unaligned(vals + k0 + 4), v3 = load_d2_unaligned(vals + k0 + 6);
            const int j[8] = {ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w};
            const double v[8] = {v0.x, v0.y, v1.x, v1.y, v2.x, v2.y, v3.x, v3.y};
            int aj[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) aj[u] = (k0 + u < k1 && j[u] != i) ? agg_in[j[u]] : -1;      // (j == i: weak or diagonal entry)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const double w = fabs(v[u]);
                if (aj[u] >= 0 && w > best) {   // ties: columns are sorted, the smaller index wins
                    best = w;
                    a = aj[u];
                }
            }
        }
        for (int k = k0 + 8; k < k1; ++k) {
            const int j = scol[k];
            if (j == i) continue;
            const int aj = agg_in[j];
            if (aj < 0) continue;
            const double w = fabs(vals[k]);
            if (w > best) {
                best = w;
                a = aj;
            }
        }
    }
    agg_out[i] = a;
}

__global__ void flag_unaggregated(int n, const int *__restrict__ agg, int *__restrict__ flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = agg[i] < 0 ? 1 : 0;
}

// (base_dev: the number of roots, where the scan that numbered them left its 64-bit total)
__global__ void agg_singletons(int n, const int *__restrict__ scan, const long long *__restrict__ base_dev, int *__restrict__ agg) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && agg[i] < 0) agg[i] = (int)*base_dev + scan[i];
}

// lambda = max_i dinv_i * sum_j |a_ij|  (one partial max per workgroup)
__global__ __launch_bounds__(256) void gershgorin_kernel(int n, const int *__restrict__ rowptr,
                                                         const double *__restrict__ vals,
                                                         const double *__restrict__ dinv,
                                                         double *__restrict__ partial_max) {
    __shared__ double red[256];
    double m = 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        double s = 0.0;
        for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) s += fabs(vals[k]);
        s *= fabs(dinv[i]);
        m = s > m ? s : m;
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) partial_max[blockIdx.x] = red[0];
}

// slots of P = (I - omega D_F^-1 A_F) T with the *filtered* matrix A_F (weak off-diagonals lumped into the
// diagonal, Vanek et al.): row i owns rowlen(A_i) + 1 slots at rowptr[i] + i.  Filtering keeps the coarse
// stencils from filling in (without it the 8-layer system reaches 200+ entries per row by level 3).
__global__ void prolong_fill(int n, const int *__restrict__ rowptr, const int *__restrict__ cols,
                             const double *__restrict__ vals, const double *__restrict__ dinv, double theta2,
                             double omega, const int *__restrict__ agg, int *__restrict__ slot_ptr,
                             long long *__restrict__ key, double *__restrict__ val) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    if (i == n) {
        slot_ptr[n] = rowptr[n] + n;
        return;
    }
    const double di = dinv[i];
    double dF = 1.0 / di;                        // a_ii + sum of the weak off-diagonals
    for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
        const int j = cols[k];
        if (j != i && !strong(vals[k], di, dinv[j], theta2)) dF += vals[k];
    }
    // a row whose lumped diagonal collapses keeps all its entries (no filtering): P must keep unit row sums
    const bool keep_all = !(dF * di > 0.05);
    if (keep_all) dF = 1.0 / di;
    int s = rowptr[i] + i;
    slot_ptr[i] = s;
    const int ai = agg[i];
    key[s] = ((long long)ai << 32);
    val[s] = 1.0;
    ++s;
    const double w = -omega / dF;
    int seq = 1;
    for (int k = rowptr[i]; k < rowptr[i + 1]; ++k, ++s, ++seq) {
        const int j = cols[k];
        if (j == i) {
            key[s] = ((long long)ai << 32) | (unsigned)seq;
            val[s] = -omega;                      // a^F_ii / d^F_i = 1
        } else if (keep_all || strong(vals[k], di, dinv[j], theta2)) {
            key[s] = ((long long)agg[j] << 32) | (unsigned)seq;
            val[s] = w * vals[k];
        } else {
            key[s] = ((long long)ai << 32) | (unsigned)seq;
            val[s] = 0.0;                         // lumped: contributes nothing to P, dropped by the merge
        }
    }
}

__global__ void prolong_slot_ptr(int n, const int *__restrict__ rowptr, int *__restrict__ slot_ptr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= n) slot_ptr[i] = rowptr[i] + i;
}

// Gershgorin bound of D_F^-1 A_F (filtered matrix)
__global__ __launch_bounds__(256) void gershgorin_filtered_kernel(int n, const int *__restrict__ rowptr,
                                                                  const int *__restrict__ cols,
                                                                  const double *__restrict__ vals,
                                                                  const double *__restrict__ dinv, double theta2,
                                                                  double *__restrict__ partial_max) {
    __shared__ double red[256];
    double m = 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const double di = dinv[i];
        double dF = 1.0 / di, off = 0.0;
        for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
            const int j = cols[k];
            if (j == i) continue;
            if (strong(vals[k], di, dinv[j], theta2)) off += fabs(vals[k]);
            else dF += vals[k];
        }
        if (!(dF * di > 0.05)) {                  // same rule as prolong_fill: such a row is not filtered
            dF = 1.0 / di;
            off = 0.0;
            for (int k = rowptr[i]; k < rowptr[i + 1]; ++k)
                if (cols[k] != i) off += fabs(vals[k]);
        }
        const double sgm = (off + fabs(dF)) / fabs(dF);
        m = sgm > m ? sgm : m;
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) partial_max[blockIdx.x] = red[0];
}

// ---- transposes whose rows come out in column order: no sort ---------------------------------------------------------
// Entry (i, c) of M lands in row c of M^T behind every row i' < i that holds c.  A wave takes 64 consecutive rows of M (one
// lane per row) and knows the order of ITS rows; what it does not know is how many entries the waves before it put into
// the same row of M^T.  One atomic on a cursor per (block, column) hands out places in the order the blocks happen to
// run -- the rows then had to be sorted afterwards (sort_csr_rows_seg: 1.6 ms beside the product A P of config C4's fine
// level, the main stream waited 1.1 ms per setup for the restrictions).  Instead the counting pass leaves, per column,
// the list of the waves that hold it with their counts -- (wave << 6 | count - 1), in the order of arrival, at most kTpK
// of them: the rows of a mesh neighbourhood lie in a few runs of consecutive rows -- and the filling pass adds up the
// counts of the waves in front of its own: a place that depends on nothing but the matrix.  Inside a wave the rows that
// hold a column are a 64-bit MASK in a wave-private LDS hash table of the columns met (atomic OR: the result does not
// depend on the order of the lanes); an entry's place among them is the number of lower bits.  No workgroup barrier, no
// scan, nothing to sort.  Columns met by more than kTpK waves (hubs, scattered numberings) take a cursor as before and
// are sorted by sort_csr_rows_seg, which skips all other rows; a wave whose rows hold more entries than its table takes
// marks all its columns as such.  Same matrix either way, bit for bit.
constexpr int kTpK = 16;              // waves per column the lists hold (64 bytes per column)
constexpr int kTpSlots = 320;         // places of a wave's hash table: 16 bytes each, 20 KiB per workgroup of four waves
constexpr int kTpMaxEntries = 240;    // entries of a wave's 64 rows the table takes (3.75 per row)

__device__ __forceinline__ unsigned tp_hash(const int c) {
    return (unsigned)(((unsigned long long)((unsigned)c * 2654435761u) * (unsigned long long)kTpSlots) >> 32);
}
__device__ __forceinline__ int tp_insert(int *key, const int c) {
    unsigned h = tp_hash(c);
    for (;;) {
        const int old = atomicCAS(&key[h], -1, c);
        if (old == -1 || old == c) return (int)h;
        h = h + 1 == (unsigned)kTpSlots ? 0u : h + 1;
    }
}
__device__ __forceinline__ int tp_find(const int *key, const int c) {
    unsigned h = tp_hash(c);
    while (key[h] != c) h = h + 1 == (unsigned)kTpSlots ? 0u : h + 1;
    return (int)h;
}
// the wave's table: every column its rows hold -> the lanes that hold it.  slot[0..3]: places of the row's first entries
__device__ __forceinline__ void tp_build(int *key, unsigned long long *mask, const int *__restrict__ cols, const int k0,
                                         const int k1, const int c[4], int slot[4]) {
    const int lane = threadIdx.x & 63;
    for (int j = lane; j < kTpSlots; j += 64) {
        key[j] = -1;
        mask[j] = 0ull;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const unsigned long long bit = 1ull << lane;
    const int ln = k1 - k0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        slot[u] = 0;
        if (u < ln) {
            slot[u] = tp_insert(key, c[u]);
            atomicOr(&mask[slot[u]], bit);
        }
    }
    for (int k = k0 + 4; k < k1; ++k) atomicOr(&mask[tp_insert(key, cols[k])], bit);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(256) void transpose_count_pairs(int n_rows, const int *__restrict__ rowptr,
                                                             const int *__restrict__ cols, int *__restrict__ cnt,
                                                             int *__restrict__ npairs, int *__restrict__ pairs,
                                                             const int all_cursors) {
    // all_cursors (PADNE_FORCE=transpose_cursors, tests): every wave takes the path of the waves whose rows exceed the table
    __shared__ int Key[4][kTpSlots];
    __shared__ unsigned long long Mask[4][kTpSlots];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wb = (int)xcd_bid() * 4 + w;
    const int r0 = wb * 64, i = r0 + lane;
    if (r0 >= n_rows) return;
    const bool live = i < n_rows;
    const int k0 = live ? rowptr[i] : 0, k1 = live ? rowptr[i + 1] : 0;
    const int ln = k1 - k0;
    const int tot = __shfl(live ? k1 : 0, min(63, n_rows - 1 - r0), 64) - __shfl(k0, 0, 64);
    if (tot > kTpMaxEntries || all_cursors) {
        // more entries than the table takes: every column of these rows goes through its cursor, in every wave that holds it
        for (int k = k0; k < k1; ++k) {
            const int cc = cols[k];
            atomicAdd(&cnt[cc], 1);
            atomicAdd(&npairs[cc], kTpK + 1);
        }
        return;
    }
    int c[4] = {0, 0, 0, 0}, slot[4];
    if (ln > 0) {
        const int4 c4 = load_i4_unaligned(cols + k0);
        c[0] = c4.x; c[1] = c4.y; c[2] = c4.z; c[3] = c4.w;
    }
    tp_build(Key[w], Mask[w], cols, k0, k1, c, slot);
    const unsigned long long below = (1ull << lane) - 1ull;
    // the first row that holds a column reports the wave's count of it
    auto report = [&](const int cc, const int sl) {
        const unsigned long long m = Mask[w][sl];
        if ((m & below) != 0ull) return;
        const int n_here = __popcll(m);
        atomicAdd(&cnt[cc], n_here);
        const int at = atomicAdd(&npairs[cc], 1);
        if (at < kTpK) pairs[(long long)cc * kTpK + at] = (int)(((unsigned)wb << 6) | (unsigned)(n_here - 1));
    };
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (u < ln) report(c[u], slot[u]);
    for (int k = k0 + 4; k < k1; ++k) {
        const int cc = cols[k];
        report(cc, tp_find(Key[w], cc));
    }
}

__global__ __launch_bounds__(256) void transpose_fill_ordered(int n_rows, const int *__restrict__ rowptr,
                                                              const int *__restrict__ cols, const double *__restrict__ vals,
                                                              const int *__restrict__ t_rowptr, const int *__restrict__ npairs,
                                                              const int *__restrict__ pairs, int *__restrict__ cursor,
                                                              int *__restrict__ key, double *__restrict__ val,
                                                              const int all_cursors) {
    // (key / val: the cols / vals arrays of the transposed matrix)
    __shared__ int Key[4][kTpSlots], Base[4][kTpSlots];
    __shared__ unsigned long long Mask[4][kTpSlots];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wb = (int)xcd_bid() * 4 + w;
    const int r0 = wb * 64, i = r0 + lane;
    if (r0 >= n_rows) return;
    const bool live = i < n_rows;
    const int k0 = live ? rowptr[i] : 0, k1 = live ? rowptr[i + 1] : 0;
    const int ln = k1 - k0;
    const int tot = __shfl(live ? k1 : 0, min(63, n_rows - 1 - r0), 64) - __shfl(k0, 0, 64);
    if (tot > kTpMaxEntries || all_cursors) {
        for (int k = k0; k < k1; ++k) {                    // (the counting pass has sent all these columns to their cursors)
            const int cc = cols[k];
            const int at = t_rowptr[cc] + atomicAdd(&cursor[cc], 1);
            key[at] = i;
            val[at] = vals[k];
        }
        return;
    }
    int c[4] = {0, 0, 0, 0}, slot[4];
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    if (ln > 0) {
        const int4 c4 = load_i4_unaligned(cols + k0);
        const double2 v01 = load_d2_unaligned(vals + k0), v23 = load_d2_unaligned(vals + k0 + 2);
        c[0] = c4.x; c[1] = c4.y; c[2] = c4.z; c[3] = c4.w;
        v[0] = v01.x; v[1] = v01.y; v[2] = v23.x; v[3] = v23.y;
    }
    int sp[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) sp[u] = u < ln ? t_rowptr[c[u]] : 0;      // (on their way while the table is built)
    tp_build(Key[w], Mask[w], cols, k0, k1, c, slot);
    const unsigned long long below = (1ull << lane) - 1ull;
    // the first row that holds a column finds out how many entries the waves in front of this one put into it: from the
    // column's list of (wave, count), or from its cursor
    auto fetch = [&](const int cc, const int sl) {
        const unsigned long long m = Mask[w][sl];
        if ((m & below) != 0ull) return;
        // (the list is requested with its length, not behind it: one round trip)
        const int4 *pp = reinterpret_cast<const int4 *>(pairs + (long long)cc * kTpK);
        const int np = npairs[cc];
        int4 q[kTpK / 4];
#pragma unroll
        for (int t = 0; t < kTpK / 4; ++t) q[t] = pp[t];
        int got = 0;
        if (np <= kTpK) {
#pragma unroll
            for (int t = 0; t < kTpK / 4; ++t) {
                const int kk[4] = {q[t].x, q[t].y, q[t].z, q[t].w};
#pragma unroll
                for (int z = 0; z < 4; ++z)
                    if (4 * t + z < np && (int)((unsigned)kk[z] >> 6) < wb) got += (kk[z] & 63) + 1;
            }
        } else {
            got = atomicAdd(&cursor[cc], __popcll(m));
        }
        Base[w][sl] = got;
    };
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (u < ln) fetch(c[u], slot[u]);
    for (int k = k0 + 4; k < k1; ++k) {
        const int cc = cols[k];
        fetch(cc, tp_find(Key[w], cc));
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (u < ln) {
            const int at = sp[u] + Base[w][slot[u]] + __popcll(Mask[w][slot[u]] & below);
            key[at] = i;
            val[at] = v[u];
        }
    for (int k = k0 + 4; k < k1; ++k) {
        const int cc = cols[k], sl = tp_find(Key[w], cc);
        const int at = t_rowptr[cc] + Base[w][sl] + __popcll(Mask[w][sl] & below);
        key[at] = i;
        val[at] = vals[k];
    }
}

// The rows of a freshly transposed matrix into column order, in place.  A wave takes kSegRows consecutive rows at a time:
// their entries are ONE contiguous range of cols / vals, loaded lane-consecutively into LDS (one round trip for ~140
// entries of a transposed prolongator, every lane busy), every ENTRY then counts the smaller columns of its own row --
// rows of a transpose hold no column twice -- and goes back to memory at row start + rank.  (The slot-based predecessor
// sorted one row at a time, 17 of 64 lanes at work and two memory round trips per row: 0.8 + 0.4 ms for the 1.4 M rows of
// config C4's first restriction, then 0.4 ms to unpack the slots; the main stream waited 0.4-0.5 ms per level for it.)
// Rows of more than kSegCap entries are listed for sort_listed_csr_rows (a workgroup each).
constexpr int kSegRows = 8, kSegCap = 64;
__global__ __launch_bounds__(256) void sort_csr_rows_seg(const int n_rows, const int *__restrict__ rowptr, int *__restrict__ cols,
                                                         double *__restrict__ vals, int *__restrict__ n_long,
                                                         int *__restrict__ long_list, const int *__restrict__ npairs) {
    // (behind transpose_fill_ordered only the rows that took a cursor -- more than kTpK waves -- are out of order)
    __shared__ int Cs[4][kSegRows * kSegCap];
    __shared__ double Vs[4][kSegRows * kSegCap];
    __shared__ int Rs[4][kSegRows + 1];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long n_chunks = ((long long)n_rows + kSegRows - 1) / kSegRows;
    const XcdSweep sw = xcd_sweep(n_chunks, 4, w);
    for (long long ch = sw.t0; ch < sw.t1; ch += sw.stride) {
        const int r0 = (int)(ch * kSegRows);
        if (__ballot(lane < kSegRows && r0 + lane < n_rows && npairs[r0 + lane] > kTpK) == 0ull) continue;
        int rp = 0;
        if (lane <= kSegRows) rp = rowptr[min(r0 + lane, n_rows)];
        // a chunk with a row beyond the LDS capacity (rare: aggregates next to a via ring, hubs) is listed as a whole --
        // its entries need not fit the wave's window then -- and sorted row by row by sort_listed_csr_rows
        const int len = __shfl_down(rp, 1, 64) - rp;
        const bool is_row = lane < kSegRows && r0 + lane < n_rows;
        if (__ballot(is_row && len > kSegCap) != 0ull) {       // (wave-uniform)
            const unsigned long long rows = __ballot(is_row);
            int base = 0;
            if (lane == 0) base = atomicAdd(n_long, __popcll(rows));
            base = __shfl(base, 0, 64);
            if (is_row) long_list[base + lane] = r0 + lane;
            continue;
        }
        if (lane <= kSegRows) Rs[w][lane] = rp;
        const int e0 = __shfl(rp, 0, 64), e1 = __shfl(rp, kSegRows, 64);
        // (LDS position of an entry: its row's slot of kSegCap places + its place in the row)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        int my_row[ (kSegRows * kSegCap) / 64 ], my_pos[ (kSegRows * kSegCap) / 64 ];
#pragma unroll
        for (int j = 0; j < (kSegRows * kSegCap) / 64; ++j) {
            const int e = e0 + lane + 64 * j;
            my_row[j] = -1;
            my_pos[j] = 0;
            if (e < e1) {
                int q = 0;                                   // the entry's row: the last of the chunk's rows that starts at or before e
#pragma unroll
                for (int t = 1; t < kSegRows; ++t) q += (Rs[w][t] <= e) ? 1 : 0;
                const int rs = Rs[w][q];
                my_row[j] = q;
                my_pos[j] = e - rs;
                Cs[w][q * kSegCap + (e - rs)] = cols[e];
                Vs[w][q * kSegCap + (e - rs)] = vals[e];
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // every entry of the chunk is in LDS before one goes back
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < (kSegRows * kSegCap) / 64; ++j) {
            const int q = my_row[j];
            if (q >= 0) {
                const int rs = Rs[w][q], rl = Rs[w][q + 1] - rs;
                const int c = Cs[w][q * kSegCap + my_pos[j]];
                int rank = 0;
                for (int f = 0; f < rl; ++f) rank += Cs[w][q * kSegCap + f] < c ? 1 : 0;
                cols[rs + rank] = c;
                vals[rs + rank] = Vs[w][q * kSegCap + my_pos[j]];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // LDS is rewritten by the next chunk
        __builtin_amdgcn_wave_barrier();
    }
}

// the listed rows (more than kSegCap entries: aggregates next to a via ring, hubs), a workgroup each: rank sort in LDS up to
// kListCap entries; what exceeds even that is sorted by one thread's insertion sort in memory (a handful of rows at most)
constexpr int kListCap = 2048;
__global__ __launch_bounds__(256) void sort_listed_csr_rows(const int *__restrict__ n_list, const int *__restrict__ row_list,
                                                            const int *__restrict__ rowptr, int *__restrict__ cols,
                                                            double *__restrict__ vals) {
    __shared__ int Cs[kListCap];
    __shared__ double Vs[kListCap];
    const int cnt = *n_list;
    for (int j = blockIdx.x; j < cnt; j += gridDim.x) {
        const int r = row_list[j];
        const int s0 = rowptr[r], n = rowptr[r + 1] - s0;
        if (n > kListCap) {
            if (threadIdx.x == 0) {
                for (int a = 1; a < n; ++a) {
                    const int c = cols[s0 + a];
                    const double v = vals[s0 + a];
                    int b = a - 1;
                    for (; b >= 0 && cols[s0 + b] > c; --b) {
                        cols[s0 + b + 1] = cols[s0 + b];
                        vals[s0 + b + 1] = vals[s0 + b];
                    }
                    cols[s0 + b + 1] = c;
                    vals[s0 + b + 1] = v;
                }
            }
            continue;                                  // (uniform over the workgroup)
        }
        for (int e = threadIdx.x; e < n; e += 256) {
            Cs[e] = cols[s0 + e];
            Vs[e] = vals[s0 + e];
        }
        __syncthreads();
        for (int e = threadIdx.x; e < n; e += 256) {
            const int c = Cs[e];
            int rank = 0;
            for (int f = 0; f < n; ++f) rank += Cs[f] < c ? 1 : 0;
            cols[s0 + rank] = c;
            vals[s0 + rank] = Vs[e];
        }
        __syncthreads();
    }
}

// C = X * Y, row-wise: upper bound of the row length, then sorted-insert accumulation.  Row m of Y is [yr[m], ye[m]) with
// its columns at yc[q * ycs]: a CSR matrix (ye = yr + 1, ycs = 1), or rows still sitting in their merge slots
// (ye = slot start + length, columns in the upper halves of the 64-bit keys: ycs = 2) -- A P is consumed once, by
// R (A P), which reads it by rows anyway, so it is never compacted.
__global__ void spgemm_count(int n_rows, const int *__restrict__ xr, const int *__restrict__ xc,
                             const int *__restrict__ yr, const int *__restrict__ ye, int *__restrict__ cnt) {
    const int i = xcd_bid() * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    long long c = 0;
    const int k0 = xr[i], k1 = xr[i + 1];
    // the first eight entries of the row together: their columns in two loads, then eight independent looks at the row
    // pointers of Y (one 8-byte load each when Y is a CSR matrix); the rest one by one
    {
        const int4 ca = load_i4_unaligned(xc + k0), cb = load_i4_unaligned(xc + k0 + 4);
        const int m[8] = {ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w};
        int ln[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            ln[u] = 0;
            if (k0 + u < k1) {
                if (ye == yr + 1) {
                    const I2u be = *reinterpret_cast<const I2u *>(yr + m[u]);
                    ln[u] = be.y - be.x;
                } else {
                    ln[u] = ye[m[u]] - yr[m[u]];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) c += ln[u];
    }
    for (int k = k0 + 8; k < k1; ++k) {
        const int m = xc[k];
        c += ye[m] - yr[m];
    }
    cnt[i] = c > 2000000000LL ? 2000000000 : (int)c;   // the scan rejects totals beyond int32
}

// The same count with LPR lanes per row, for rows of X beyond the eight entries the kernel above takes at once (the
// restriction: 22 entries per row on the fine level of C4, where a thread walked the other 14 one dependent load after
// the other -- 203 us for 30 M entries).
template <int LPR>
__global__ __launch_bounds__(256) void spgemm_count_lanes(int n_rows, const int *__restrict__ xr, const int *__restrict__ xc,
                                                          const int *__restrict__ yr, const int *__restrict__ ye, int *__restrict__ cnt) {
    const long long tid = (long long)xcd_bid() * blockDim.x + threadIdx.x;
    const long long i = tid / LPR;
    const int l = (int)(tid % LPR);
    long long c = 0;
    if (i < n_rows) {
        const int k0 = xr[i], k1 = xr[i + 1];
        int k = k0 + l;
        for (; k + LPR < k1; k += 2 * LPR) {
            const int m0 = xc[k], m1 = xc[k + LPR];
            const int b0 = yr[m0], e0 = ye[m0], b1 = yr[m1], e1 = ye[m1];
            c += (e0 - b0) + (e1 - b1);
        }
        if (k < k1) {
            const int m0 = xc[k];
            c += ye[m0] - yr[m0];
        }
    }
#pragma unroll
    for (int d = 1; d < LPR; d <<= 1) c += __shfl_xor(c, d, 64);
    if (i < n_rows && l == 0) cnt[i] = c > 2000000000LL ? 2000000000 : (int)c;
}

__global__ void spgemm_rows(int n_rows, const int *__restrict__ xr, const int *__restrict__ xc,
                            const double *__restrict__ xv, const int *__restrict__ yr, const int *__restrict__ yc,
                            const double *__restrict__ yv, const int *__restrict__ ye, const int ycs, const int *__restrict__ slot_ptr,
                            long long *__restrict__ key, double *__restrict__ val, int *__restrict__ row_len) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    long long *K = key + slot_ptr[i];
    double *V = val + slot_ptr[i];
    int m = 0;   // distinct columns so far, kept sorted
    for (int k = xr[i]; k < xr[i + 1]; ++k) {
        const int mid = xc[k];
        const double a = xv[k];
        for (int q = yr[mid]; q < ye[mid]; ++q) {
            const long long c = (long long)yc[(long long)q * ycs] << 32;
            const double v = a * yv[q];
            // binary search for c in K[0..m)
            int lo = 0, hi = m;
            while (lo < hi) {
                const int h = (lo + hi) >> 1;
                if (K[h] < c) lo = h + 1; else hi = h;
            }
            if (lo < m && K[lo] == c) {
                V[lo] += v;                       // products are added in generation order: deterministic
            } else {
                for (int t = m; t > lo; --t) {
                    K[t] = K[t - 1];
                    V[t] = V[t - 1];
                }
                K[lo] = c;
                V[lo] = v;
                ++m;
            }
        }
    }
    row_len[i] = m;
}

// Short rows (A P on the fine levels: a dozen products per row): one thread per row, the sorted distinct list of the row in
// LDS ([slot][thread] layout, bank-conflict free), a thread takes KC entries of its row of X at once -- columns and values,
// then the row bounds of Y they select -- and requests the products of entry u + 1 while those of entry u go into the list.
// Rows that would exceed CAP distinct columns are flagged (row_len = -1) and redone by the wave kernels / spgemm_rows_redo.
// `begin` (may be null): the rows of a wave's 64-row tile are written BACK TO BACK from the start of the tile's slot range
// instead of each into its own range of product-count size -- the merged rows are a third of their products (5.2 of 17 on
// the fine level of C4), and what reads them afterwards (R (A P) by rows, the W build) touched every line of a 2.6 GB arena
// for 0.83 GB of entries.  begin[i] receives the row's place; a row that overflows its list (redone later, up to its product
// count long) gets its place from the END of the tile's range: merged lengths from the front and product counts from the
// back cannot meet, their sum is at most the range.
template <int CAP, int KC, int QC, int YCS>
__global__ __launch_bounds__(128) void spgemm_rows_lds_pipe(int n_rows, const int *__restrict__ xr, const int *__restrict__ xc,
                                                            const double *__restrict__ xv, const int *__restrict__ yr,
                                                            const int *__restrict__ yc, const double *__restrict__ yv,
                                                            const int *__restrict__ ye, const int *__restrict__ slot_ptr,
                                                            long long *__restrict__ key, double *__restrict__ val,
                                                            int *__restrict__ row_len, int *__restrict__ begin) {
    static_assert(CAP % 2 == 0, "keys are kept in pairs");
    __shared__ int2 Kc[CAP / 2][128];
    __shared__ double Vc[CAP][128];
    const int t = threadIdx.x;
    const int i = xcd_bid() * 128 + t;
    const bool live = i < n_rows;
    if (begin == nullptr && !live) return;
    int m = 0;
    bool overflow = false;
    // The list of a row is kept in order of APPEARANCE and sorted once at the end.  (It used to be kept sorted: a search
    // loop, a shift loop and three-way branching per product, each lane with its own trip counts -- 4150 scalar and 1700
    // vector instructions per wave of 64 rows, the scalar unit 72 % busy: the kernel was bound by the issue of the
    // instructions that steer divergent lanes.)  Now a product is compared with the keys of the list as far as the longest
    // list of the wave reaches -- free places hold -1, no column -- and is either added to the sum it found or appended:
    // straight-line code under a predicate, the sums of a column still in generation order.
#pragma unroll
    for (int u = 0; u < CAP / 2; ++u) Kc[u][t] = make_int2(-1, -1);
    auto insert = [&](const bool act, const int c, const double v) {
        int pos = -1;
#pragma unroll
        for (int u = 0; u < CAP; u += 2) {
            if (__all(u >= m)) break;
            const int2 kk = Kc[u >> 1][t];
            pos = kk.x == c ? u : pos;
            pos = kk.y == c ? u + 1 : pos;
        }
        if (act && !overflow) {
            const bool found = pos >= 0;
            if (!found && m == CAP) {
                overflow = true;
            } else {
                const int p = found ? pos : m;
                const double base = found ? Vc[p][t] : -0.0;      // -0 + v == v: the first product is taken as it is
                Vc[p][t] = base + v;
                if (!found) {
                    reinterpret_cast<int *>(&Kc[p >> 1][t])[p & 1] = c;
                    ++m;
                }
            }
        }
    };
    const int x0 = live ? xr[i] : 0, x1 = live ? xr[i + 1] : 0;
    for (int kb = x0; kb < x1 && !overflow; kb += KC) {
        const int nk = min(KC, x1 - kb);
        int mid[KC], ys[KC], ln[KC];
        double a[KC];
        // What bounds the loads is their number: every lane reads somewhere else, and the address unit takes about a cycle
        // per lane and instruction whatever the width.  So the rows are read four indices / two doubles at a time (unaligned
        // 16-byte loads; past a row's end lies the next row or the zero padding of the arrays, and is not used).
        static_assert(KC % 4 == 1 && QC == 4, "the loads below are written for 4 k + 1 entries of X and rows of Y read four at a time");
#pragma unroll
        for (int u = 0; u < KC - 1; u += 4) {
            const I4u c4 = *reinterpret_cast<const I4u *>(xc + kb + u);
            mid[u] = c4.x; mid[u + 1] = c4.y; mid[u + 2] = c4.z; mid[u + 3] = c4.w;
            const D2u v0 = *reinterpret_cast<const D2u *>(xv + kb + u), v1 = *reinterpret_cast<const D2u *>(xv + kb + u + 2);
            a[u] = v0.x; a[u + 1] = v0.y; a[u + 2] = v1.x; a[u + 3] = v1.y;
        }
        mid[KC - 1] = xc[kb + KC - 1];
        a[KC - 1] = xv[kb + KC - 1];
#pragma unroll
        for (int u = 0; u < KC; ++u) {
            mid[u] = u < nk ? mid[u] : 0;                  // (behind the row's end: row 0 of Y, not used)
            if (ye == yr + 1) {
                const I2u be = *reinterpret_cast<const I2u *>(yr + mid[u]);
                ys[u] = be.x;
                ln[u] = u < nk ? be.y - be.x : 0;
            } else {
                ys[u] = yr[mid[u]];
                ln[u] = u < nk ? ye[mid[u]] - ys[u] : 0;
            }
        }
        int cb[2][QC];
        double vb[2][QC];
        auto fetch = [&](const int u, int (&c)[QC], double (&v)[QC]) {
            if (YCS == 1) {
                const I4u c4 = *reinterpret_cast<const I4u *>(yc + ys[u]);
                c[0] = c4.x; c[1] = c4.y; c[2] = c4.z; c[3] = c4.w;
            } else {                                       // slots: yc points at the high word of the first 64-bit key, the column
                const L2u k0 = *reinterpret_cast<const L2u *>(reinterpret_cast<const long long *>(yc) + ys[u]);
                const L2u k1 = *reinterpret_cast<const L2u *>(reinterpret_cast<const long long *>(yc) + ys[u] + 2);
                c[0] = (int)k0.x; c[1] = (int)k0.y; c[2] = (int)k1.x; c[3] = (int)k1.y;      // (the low word of a read that starts there)
            }
            const D2u v0 = *reinterpret_cast<const D2u *>(yv + ys[u]), v1 = *reinterpret_cast<const D2u *>(yv + ys[u] + 2);
            v[0] = v0.x; v[1] = v0.y; v[2] = v1.x; v[3] = v1.y;
        };
        fetch(0, cb[0], vb[0]);
#pragma unroll
        for (int u = 0; u < KC; ++u) {
            if (u + 1 < KC) fetch(u + 1, cb[(u + 1) & 1], vb[(u + 1) & 1]);
#pragma unroll
            for (int w = 0; w < QC; ++w)
                if (__any(w < ln[u])) insert(w < ln[u], cb[u & 1][w], a[u] * vb[u & 1][w]);
            for (int q = QC; q < ln[u] && !overflow; ++q)          // a Y row longer than the buffer: the rest one by one
                insert(true, yc[(long long)(ys[u] + q) * YCS], a[u] * yv[ys[u] + q]);
        }
    }
    int place = live ? slot_ptr[i] : 0;
    if (begin != nullptr) {
        // (every lane of the wave is here: the tile's rows back to back, the overflowed ones from the end of its range)
        const int lane = t & 63;
        const int tile0 = i - lane, tile1 = min(tile0 + 64, n_rows);
        const int len = (live && !overflow) ? m : 0;
        const int bound = (live && overflow) ? slot_ptr[i + 1] - slot_ptr[i] : 0;
        // (prefix sums of the lengths from the front, of the bounds from the back: total - prefix + own)
        const int incl = scan_incl_lanes<64>(len);
        const int bound_incl = scan_incl_lanes<64>(bound);
        const int back = __builtin_amdgcn_readlane(bound_incl, 63) - bound_incl + bound;
        if (tile0 < n_rows) {
            const int r_begin = slot_ptr[tile0], r_end = slot_ptr[tile1];
            place = overflow ? r_end - back : r_begin + (incl - len);
        }
        if (live) begin[i] = place;
    }
    if (!live) return;
    if (overflow) {
        row_len[i] = -1;                                   // redo in global memory
        return;
    }
    long long *K = key + place;
    double *V = val + place;
    // the list goes out in column order: an entry's place is the number of smaller keys
    int kr[CAP];
#pragma unroll
    for (int u = 0; u < CAP; u += 2) {
        const int2 kk = Kc[u >> 1][t];
        kr[u] = u < m ? kk.x : 2147483647;
        kr[u + 1] = u + 1 < m ? kk.y : 2147483647;
    }
#pragma unroll
    for (int u = 0; u < CAP; ++u) {
        if (__all(u >= m)) break;
        int rank = 0;
#pragma unroll
        for (int w = 0; w < CAP; ++w) rank += kr[w] < kr[u] ? 1 : 0;
        if (u < m) {
            K[rank] = (long long)kr[u] << 32;
            V[rank] = Vc[u][t];
        }
    }
    row_len[i] = m;
}

// Wave-per-row (or half-wave-per-row) kernels for rows with tens to hundreds of products (R * (A P) on every level,
// A * P below the finest): the thread-per-row list kernels above serialise ~100 sorted insertions per thread, these
// spread a row over the lanes of a wave.  A row is worked through in chunks of LANES products, in generation order:
//   1. lanes = entries of the X row: lengths of the Y rows they select, prefix sum -> product offsets.  The entries
//      that have products are packed, and a bit per entry at its first product tells product p which entry it belongs
//      to (a population count instead of a bisection);
//   2. a chunk's products (column c, x*y) look c up in an open-addressing table in LDS (integer CAS only).  The lane
//      that claims a free slot numbers the column (in order of appearance) and leaves the number beside the key;
//   3. every product sets ITS lane's bit in the mask of its column, and lane j -- owner of column number j -- adds the
//      products its mask names, lowest lane first, to a register.  Chunks in order and bits in lane order: every
//      column sees its products in generation order, hence bit-identical to spgemm_rows with no float atomics;
//   4. after the last chunk the owners rank their columns among the row's and store (column, sum) in rank order.
// (Before: the products were stored, the table compacted and rank-sorted, every product bisected the sorted columns
// and one ballot per column told its owner which products to add -- ~2.5 times the wave instructions, which is what
// bounds these kernels: R (A P) of C4's fine level 1.35 ms.)
// Rows that do not fit (X row > LANES entries, > CAPP products, > HT/2 distinct columns) are flagged (row_len -1).
template <int CAPP, int HT, int LANES>
struct SpgemmRowLds {
    using Mask = typename std::conditional<LANES == 64, unsigned long long, unsigned>::type;
    int ht[HT];                        // column keys
    int hidx[HT];                      // number of the column in that slot
    int dk[HT / 2];                    // column by number
    Mask mask[HT / 2];                 // lanes of the current chunk whose product belongs to the column
    double pvb[LANES];                 // the chunk's products
    double xs[LANES];                  // packed X entries: value, first Y position, first product
    int ys[LANES];
    int offc[LANES];
    unsigned long long bits[CAPP / 64];
};

// All lanes of the wave call this together (`has_row` false: the lane group only keeps step).  Returns the number of
// distinct columns stored at K / V, or -1 when the row does not fit.
template <int CAPP, int HT, int LANES>
__device__ __forceinline__ int spgemm_row_by_masks(SpgemmRowLds<CAPP, HT, LANES> &S, const int sl, const int sub,
                                                   const bool has_row, const int nx, const int len, const int ystart,
                                                   const double a, const int *__restrict__ yc, const double *__restrict__ yv,
                                                   const int ycs, long long *__restrict__ K, double *__restrict__ V) {
    using Mask = typename SpgemmRowLds<CAPP, HT, LANES>::Mask;
    static_assert(CAPP % 64 == 0 && CAPP / 64 <= LANES && 64 % LANES == 0, "bits of the product offsets");
    constexpr int NC = (HT / 2 + LANES - 1) / LANES;          // columns per owner lane
    constexpr int EMPTY = -1;
    const unsigned long long sub_mask = (LANES == 64 ? ~0ull : ((1ull << LANES) - 1ull));
    const int shift = sub * LANES;
    const unsigned long long below = (1ull << sl) - 1ull;
    static_assert(LANES == 16 || LANES == 32 || LANES == 64, "the scan below");
    const int incl = scan_incl_lanes<LANES>(len);
    const int np = LANES == 64   ? __builtin_amdgcn_readlane(incl, 63)
                   : LANES == 32 ? (sub != 0 ? __builtin_amdgcn_readlane(incl, 63) : __builtin_amdgcn_readlane(incl, 31))
                                 : __builtin_amdgcn_update_dpp(0, incl, 0x15F, 0xf, 0xf, false);      // row_newbcast:15
    const bool live = has_row && nx <= LANES && np <= CAPP;
    const unsigned long long have = (__ballot(live && len > 0) >> shift) & sub_mask;
    if (live) {
        for (int h = sl; h < HT; h += LANES) S.ht[h] = EMPTY;
        for (int h = sl; h < HT / 2; h += LANES) S.mask[h] = (Mask)0;
        if (sl < CAPP / 64) S.bits[sl] = 0ull;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (len > 0) {
            const int pos = __popcll(have & below);
            const int o = incl - len;
            S.offc[pos] = o;
            S.ys[pos] = ystart;
            S.xs[pos] = a;
            atomicOr(&S.bits[o >> 6], 1ull << (o & 63));
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    int np_w = live ? np : 0;                                 // chunks the wave walks: those of its longest row
    if (LANES == 64) {
        np_w = __builtin_amdgcn_readfirstlane(np_w);
    } else {
        int m = 0;
#pragma unroll
        for (int g = 0; g < 64 / LANES; ++g) m = max(m, __builtin_amdgcn_readlane(np_w, g * LANES));
        np_w = m;
    }
    int cnt = 0, pre = 0;
    bool overflow = false;
    double acc[NC];
#pragma unroll
    for (int u = 0; u < NC; ++u) acc[u] = -0.0;               // -0 + v == v for every v: the first product is taken as it is
    for (int base = 0; base < np_w; base += LANES) {
        const int p = base + sl;
        const bool mine = live && p < np;
        unsigned long long word = 0ull;
        if (live && base < np) word = S.bits[base >> 6];
        int c = 0;
        if (mine) {
            const int kk = pre + __popcll(word & (~0ull >> (63 - (p & 63)))) - 1;
            const int q = S.ys[kk] + (p - S.offc[kk]);
            c = yc[(long long)q * ycs];
            S.pvb[sl] = S.xs[kk] * yv[q];
        }
        if (((base + LANES) & 63) == 0) pre += __popcll(word);
        bool pending = mine;
        unsigned h = ((unsigned)c * 2654435761u) >> 7;
        int idx = -1;
        for (int round = 0; __any(pending); ++round) {
            if (round == HT) {                                 // the table is full of other columns
                overflow = overflow || pending;
                break;
            }
            h &= (HT - 1);
            int old = EMPTY - 1;
            if (pending) old = atomicCAS(&S.ht[h], EMPTY, c);
            const bool won = old == EMPTY;
            const unsigned long long wm = (__ballot(won) >> shift) & sub_mask;
            if (won) {
                idx = cnt + __popcll(wm & below);
                S.hidx[h] = idx;
                if (idx < HT / 2) S.dk[idx] = c;
            }
            cnt += __popcll(wm);
            // (LDS serves a wave's accesses in the order they were issued: the number stored by the winner is there
            // for the lanes that met its key in the same round)
            asm volatile("" ::: "memory");
            const bool hit = pending && old == c;
            if (hit) idx = S.hidx[h];
            pending = pending && !won && !hit;
            ++h;
        }
        if (mine && idx >= 0 && idx < HT / 2) atomicOr(&S.mask[idx], (Mask)1 << sl);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < NC; ++u) {
            const int j = sl + u * LANES;
            if (live && j < cnt && j < HT / 2) {
                Mask m = S.mask[j];
                if (m != (Mask)0) {
                    S.mask[j] = (Mask)0;
                    do {
                        const int b = (LANES == 64 ? __ffsll((long long)m) : __ffs((int)m)) - 1;
                        acc[u] += S.pvb[b];
                        m &= m - (Mask)1;
                    } while (m != (Mask)0);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
    const bool any_overflow = ((__ballot(overflow) >> shift) & sub_mask) != 0ull;
    const int nd = cnt;
    const bool ok = live && !any_overflow && nd <= HT / 2;
#pragma unroll
    for (int u = 0; u < NC; ++u) {
        const int j = sl + u * LANES;
        if (ok && j < nd) {
            const int kq = S.dk[j];
            int rank = 0;
            for (int t = 0; t < nd; ++t) rank += S.dk[t] < kq ? 1 : 0;
            K[rank] = (long long)kq << 32;
            V[rank] = acc[u];
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    return ok ? nd : -1;
}

// The kernel around it: 256 / LANES lane groups per workgroup, each working through its share of the rows -- every XCD
// a contiguous eighth of them (see xcd_bid: the rows of Y that neighbouring rows of X gather, the slots of A P around an
// aggregate, are then still in that XCD's L2 when the next row asks for them), or the list of rows an earlier pass
// flagged (row_list / list_count, collected by collect_pending_rows; rows of the list that a pass in between has
// finished are skipped).
// The loads that lead to a row's products form a chain of dependent global accesses (row number -> xr -> xc / xv ->
// yr / ye -> yc / yv).  All but the last run as a pipeline over the rows of the sweep: an iteration asks for the number of
// the row four ahead, the bounds of the row three ahead, the entries of the row two ahead and the Y bounds (and the
// place) of the next row, each from what the iteration before received -- independent loads, in flight while the
// current row is worked on, instead of a chain the wave waits through row by row (five exposed latencies per row
// before, one now).
template <int CAPP, int HT, int LANES, bool LIST>
__global__ __launch_bounds__(256) void spgemm_rows_lanes(int n_rows, const int *__restrict__ xr, const int *__restrict__ xc,
                                                         const double *__restrict__ xv, const int *__restrict__ yr,
                                                         const int *__restrict__ yc, const double *__restrict__ yv, const int *__restrict__ ye, const int ycs,
                                                         const int *__restrict__ slot_ptr, long long *__restrict__ key,
                                                         double *__restrict__ val, int *__restrict__ row_len,
                                                         const int *__restrict__ row_list = nullptr,
                                                         const int *__restrict__ list_count = nullptr) {
    constexpr int G = 256 / LANES;                    // rows (lane groups) per workgroup
    __shared__ SpgemmRowLds<CAPP, HT, LANES> s_rows[G];
    const int lane = threadIdx.x & 63;
    const int sl = lane % LANES, sub = lane / LANES, g = threadIdx.x / LANES;
    constexpr bool listed = LIST;                      // (a template parameter: the sweep over all rows carries no lengths of an earlier pass)
    int t_l, t_end, stride;                            // positions in the sweep: t_l, t_l + stride, ... < t_end
    if (listed) {
        t_end = *list_count;
        t_l = (int)blockIdx.x * G + g;
        stride = (int)gridDim.x * G;
    } else {
        const int nslab = (gridDim.x % kNumXcd == 0) ? kNumXcd : 1;
        const int slab = blockIdx.x % nslab;
        t_end = (int)((long long)(slab + 1) * n_rows / nslab);
        t_l = (int)((long long)slab * n_rows / nslab) + (int)(blockIdx.x / nslab) * G + g;
        stride = (int)(gridDim.x / nslab) * G;
    }
    if (t_l > t_end) t_l = t_end;
    // row < 0: no row.  len_*: the row's length as an earlier pass left it (listed rows only; >= 0: finished)
    int row_l = -1;                                                                 // stage L: the row's number
    int row_a = -1, x0_a = 0, x1_a = 0, len_a = -1;                                 // stage A: bounds of the row of X
    int row_b = -1, x0_b = 0, x1_b = 0, len_b = -1, mid_b = 0;                      // stage B: its entries
    double a_b = 0.0;
    int row_c = -1, nx_c = 0, len_c = -1, ys_c = 0, ye_c = 0, place_c = 0;          // stage C: bounds in Y, place of the result
    double a_c = 0.0;
    auto advance = [&]() {
        const int nx_b = x1_b - x0_b;
        row_c = row_b; nx_c = nx_b; len_c = len_b; a_c = a_b; ys_c = 0; ye_c = 0; place_c = 0;
        if (row_b >= 0) {
            place_c = slot_ptr[row_b];
            if (nx_b <= LANES && sl < nx_b) {
                ys_c = yr[mid_b];
                ye_c = ye[mid_b];
            }
        }
        const int nx_a = x1_a - x0_a;
        row_b = row_a; x0_b = x0_a; x1_b = x1_a; len_b = len_a; mid_b = 0; a_b = 0.0;
        if (row_a >= 0 && nx_a <= LANES && sl < nx_a) {
            mid_b = xc[x0_a + sl];
            a_b = xv[x0_a + sl];
        }
        row_a = row_l; x0_a = 0; x1_a = 0; len_a = -1;
        if (row_a >= 0) {
            x0_a = xr[row_a];
            x1_a = xr[row_a + 1];
            if (listed) len_a = row_len[row_a];
        }
        row_l = -1;
        if (t_l < t_end) row_l = listed ? row_list[t_l] : t_l;
        t_l = t_l < t_end - stride ? t_l + stride : t_end;
    };
    advance();
    advance();
    advance();
    advance();                                     // (stage C now holds the first row)
    // (the lane groups of a wave stay together to the end of the longer sweep: the row routine is called wave-wide)
    while (__any(row_c >= 0)) {
        const int i = row_c;
        const bool has_row = i >= 0 && !(listed && len_c >= 0);
        const int nx = nx_c, ystart = ys_c, len = ye_c - ys_c, place = place_c;
        const double a = a_c;
        advance();
        const int nd = spgemm_row_by_masks<CAPP, HT, LANES>(s_rows[g], sl, sub, has_row, nx, len, ystart, a, yc, yv, ycs,
                                                            key + place, val + place);
        if (has_row && sl == 0) row_len[i] = nd;
    }
}

// One workgroup per row with a dense accumulator in LDS (coarse levels: few thousand columns, long rows).
// The X entries of the row are processed one after another and the lanes spread over the Y row, whose
// columns are distinct, so every accumulator cell sees its products in the same order as spgemm_rows.
__global__ __launch_bounds__(256) void spgemm_rows_dense(int n_cols, const int *__restrict__ xr, const int *__restrict__ xc,
                                                         const double *__restrict__ xv, const int *__restrict__ yr,
                                                         const int *__restrict__ yc, const double *__restrict__ yv, const int *__restrict__ ye, const int ycs,
                                                         const int *__restrict__ slot_ptr, long long *__restrict__ key,
                                                         double *__restrict__ val, int *__restrict__ row_len,
                                                         const int only_pending, const int *__restrict__ row_list = nullptr,
                                                         const int *__restrict__ list_count = nullptr) {
    // row_list / list_count: behind the wave kernels the workgroups walk the list of the rows those left (collected by
    // collect_pending_rows) -- one workgroup per row of the level just to find that its row was done cost 150 us on 140 k rows
    extern __shared__ double acc_and_flag[];               // n_cols doubles + n_cols bytes
    double *acc = acc_and_flag;
    unsigned char *hit = (unsigned char *)(acc + n_cols);
    __shared__ int wave_cnt[4];
    const int n_turns = row_list != nullptr ? *list_count : 1;
    for (int turn = row_list != nullptr ? (int)blockIdx.x : 0; turn < n_turns; turn += row_list != nullptr ? (int)gridDim.x : 1) {
    const int i = row_list != nullptr ? row_list[turn] : (int)blockIdx.x;
    if (only_pending && row_len[i] >= 0) continue;         // behind the wave kernels: the rows they left (-1)
    __syncthreads();                                       // (the previous turn's compaction has read acc / hit)
    for (int c = threadIdx.x; c < n_cols; c += 256) {
        acc[c] = 0.0;
        hit[c] = 0;
    }
    __syncthreads();
    for (int k = xr[i]; k < xr[i + 1]; ++k) {
        const int mid = xc[k];
        const double a = xv[k];
        for (int q = yr[mid] + threadIdx.x; q < ye[mid]; q += 256) {
            const int c = yc[(long long)q * ycs];
            acc[c] += a * yv[q];
            hit[c] = 1;
        }
        __syncthreads();
    }
    // ordered compaction of the touched columns
    long long *K = key + slot_ptr[i];
    double *V = val + slot_ptr[i];
    int base = 0;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int c0 = 0; c0 < n_cols; c0 += 256) {
        const int c = c0 + threadIdx.x;
        const bool on = c < n_cols && hit[c];
        const unsigned long long bal = __ballot(on);
        if (lane == 0) wave_cnt[w] = __popcll(bal);
        __syncthreads();
        int off = base;
        for (int u = 0; u < w; ++u) off += wave_cnt[u];
        const int total = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        if (on) {
            const int pos = off + __popcll(bal & ((1ull << lane) - 1ull));
            K[pos] = (long long)c << 32;
            V[pos] = acc[c];
        }
        base += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) row_len[i] = base;
    }
}

// rows an earlier pass left (-1), as a list: the passes behind it walk the list instead of all rows
__global__ __launch_bounds__(256) void collect_pending_rows(int n, const int *__restrict__ row_len, int *__restrict__ list,
                                                            int *__restrict__ count) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool pend = i < n && row_len[i] < 0;
    __shared__ int wave_cnt[4], block_base;          // one atomic per workgroup (see mis_decide_list)
    const unsigned long long mask = __ballot(pend);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) wave_cnt[w] = __popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        block_base = t ? atomicAdd(count, t) : 0;
    }
    __syncthreads();
    int base = block_base;
    for (int q = 0; q < w; ++q) base += wave_cnt[q];
    if (pend) list[base + __popcll(mask & ((1ull << lane) - 1ull))] = i;
}

// only the rows the LDS variant gave up on
__global__ void spgemm_rows_redo(int n_rows, const int *__restrict__ xr, const int *__restrict__ xc,
                                 const double *__restrict__ xv, const int *__restrict__ yr, const int *__restrict__ yc,
                                 const double *__restrict__ yv, const int *__restrict__ ye, const int ycs, const int *__restrict__ slot_ptr,
                                 long long *__restrict__ key, double *__restrict__ val, int *__restrict__ row_len) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows || row_len[i] >= 0) return;
    long long *K = key + slot_ptr[i];
    double *V = val + slot_ptr[i];
    int m = 0;
    for (int k = xr[i]; k < xr[i + 1]; ++k) {
        const int mid = xc[k];
        const double a = xv[k];
        for (int q = yr[mid]; q < ye[mid]; ++q) {
            const long long c = (long long)yc[(long long)q * ycs] << 32;
            const double v = a * yv[q];
            int lo = 0, hi = m;
            while (lo < hi) {
                const int h = (lo + hi) >> 1;
                if (K[h] < c) lo = h + 1; else hi = h;
            }
            if (lo < m && K[lo] == c) {
                V[lo] += v;
            } else {
                for (int u = m; u > lo; --u) {
                    K[u] = K[u - 1];
                    V[u] = V[u - 1];
                }
                K[lo] = c;
                V[lo] = v;
                ++m;
            }
        }
    }
    row_len[i] = m;
}

// dense coarse matrix and its inverse
__global__ void dense_from_csr(int n, const int *__restrict__ rowptr, const int *__restrict__ cols,
                               const double *__restrict__ vals, double *__restrict__ W) {
    // W = A, n x n row-major
    const int i = blockIdx.x;
    for (int c = threadIdx.x; c < n; c += blockDim.x) W[(size_t)i * n + c] = 0.0;
    __syncthreads();
    for (int k = rowptr[i] + threadIdx.x; k < rowptr[i + 1]; k += blockDim.x) W[(size_t)i * n + cols[k]] = vals[k];
}

// Blocked in-place Gauss-Jordan inversion, ping-pong between two n x n copies: one launch eliminates kGjBlock pivots
// (the first version, one pivot per launch on [A | I], was 10 % of the multigrid setup; eight pivots per launch on
// [A | I] still moved 17 GB for n = 1600).  With K the pivot indices and D = W[K, K]:
//     W[K, K] <- D^-1            W[K, j] <- D^-1 W[K, j]
//     W[i, K] <- -W[i, K] D^-1   W[i, j] <- W[i, j] - W[i, K] D^-1 W[K, j]          (i, j outside K)
// Each thread owns one column of a strip of rows, forms its column R of the new pivot rows once (for a pivot
// column that is the column of D^-1 itself, and the old entry counts as zero) and applies it down the strip.  No
// pivoting (the coarse operators are symmetric positive definite); every entry reads the OLD matrix and writes the new.
constexpr int kGjBlock = 16;
constexpr int kGjStrip = 32;
__device__ __forceinline__ double Rsel(const double (&R)[kGjBlock], int a) {
    double v = 0.0;
#pragma unroll
    for (int b = 0; b < kGjBlock; ++b) v = (a == b) ? R[b] : v;
    return v;
}
// The 16 x 16 pivot block, inverted in place by ONE wave: entry (a, b) sits in lane (a & 3) << 4 | b, register a >> 2, and
// travels between the lanes by shuffles (per pivot one look at the pivot, one at the pivot row, four at the pivot column).
// The arithmetic of gj_block_step's inversion, entry for entry.
__device__ __forceinline__ void gj_invert_block16(double (&dd)[4], const int lane) {
    const int eb = lane & 15, ea0 = lane >> 4;
#pragma unroll
    for (int p = 0; p < kGjBlock; ++p) {
        const double dpp = __shfl(dd[p >> 2], ((p & 3) << 4) | p, 64);
        const double drow = __shfl(dd[p >> 2], ((p & 3) << 4) | eb, 64);
        double dcol[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) dcol[q] = __shfl(dd[q], (lane & 48) | p, 64);
        // 1 / pivot by the hardware reciprocal + two Newton steps (the pivots of a symmetric positive definite block are
        // positive and far from the ends of the exponent range): the correctly rounded division is a dozen dependent
        // instructions in a chain of 64 pivots that a launch of the big step waits for
        double inv = __builtin_amdgcn_rcp(dpp);
        inv = fma(inv, fma(-dpp, inv, 1.0), inv);
        inv = fma(inv, fma(-dpp, inv, 1.0), inv);
        const double rp = drow * inv;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ea = ea0 + 4 * q;
            const double cp = dcol[q];
            dd[q] = (ea == p) ? (eb == p ? inv : rp) : (eb == p ? -cp * inv : dd[q] - cp * rp);
        }
    }
}

template <bool kFull>      // kFull: all 16 pivots exist (every launch but possibly the last)
__global__ __launch_bounds__(256) void gj_block_step(int n, int k0, const double *__restrict__ in, double *__restrict__ out) {
    static_assert(kGjBlock == 16, "the pivot block inversion below maps a 16 x 16 block onto one wave");
    __shared__ __attribute__((aligned(16))) double Dinv[kGjBlock][kGjBlock];      // read as broadcasts: no padding, 16-byte reads
    __shared__ __attribute__((aligned(16))) double Lcol[kGjStrip][kGjBlock];     // W[i, K] of the strip's rows
    const int bs = kFull ? kGjBlock : min(kGjBlock, n - k0);
    const int r0 = blockIdx.y * kGjStrip;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = c < n;
    const bool pivot_col = c >= k0 && c < k0 + bs;
    // everything this thread needs from the old matrix is requested before the pivot block is inverted, so the
    // inversion (a serial chain of 16 steps) runs under the memory latency: its column of the pivot rows, the
    // strip's old entries, the strip's multipliers
    double P[kGjBlock], V[kGjStrip];
#pragma unroll
    for (int b = 0; b < kGjBlock; ++b) P[b] = (live && !pivot_col && b < bs) ? in[(size_t)(k0 + b) * n + c] : 0.0;
#pragma unroll
    for (int q = 0; q < kGjStrip; ++q) {
        const int i = r0 + q;
        V[q] = (live && i < n && !pivot_col) ? in[(size_t)i * n + c] : 0.0;
    }
    for (int t = threadIdx.x; t < kGjStrip * kGjBlock; t += 256) {
        const int i = r0 + t / kGjBlock, b = t % kGjBlock;
        Lcol[t / kGjBlock][b] = (i < n && (kFull || b < bs)) ? in[(size_t)i * n + k0 + b] : 0.0;
    }
    if (threadIdx.x < 64) {
        // in-place Gauss-Jordan of the pivot block D by ONE wave, entries in registers (lane: column lane % 16, rows
        // lane / 16 + 4 q), exchanged through LDS whose operations a wave sees in order: no workgroup barrier in the
        // chain.  Padded with the identity when fewer than 16 pivots are left.
        const int lane = threadIdx.x, eb = lane & 15, ea0 = lane >> 4;
        double dd[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ea = ea0 + 4 * q;
            dd[q] = (ea < bs && eb < bs) ? in[(size_t)(k0 + ea) * n + k0 + eb] : (ea == eb ? 1.0 : 0.0);
        }
        gj_invert_block16(dd, lane);      // (by shuffles: no store / wait / load round trip through LDS per pivot)
#pragma unroll
        for (int q = 0; q < 4; ++q) Dinv[ea0 + 4 * q][eb] = dd[q];
    }
    __syncthreads();
    if (!live) return;
    double R[kGjBlock];
    if (pivot_col) {
#pragma unroll
        for (int a = 0; a < kGjBlock; ++a) R[a] = Dinv[a][c - k0];
    } else {
#pragma unroll
        for (int a = 0; a < kGjBlock; ++a) {
            double sum = 0.0;
#pragma unroll
            for (int b = 0; b < kGjBlock; ++b) sum = fma(Dinv[a][b], P[b], sum);
            R[a] = sum;
        }
    }
#pragma unroll
    for (int q = 0; q < kGjStrip; ++q) {
        const int i = r0 + q;
        double v = V[q];
#pragma unroll
        for (int b = 0; b < kGjBlock; ++b) v = fma(-Lcol[q][b], R[b], v);
        if (i >= k0 && i < k0 + bs) v = Rsel(R, i - k0);
        if (i < n) out[(size_t)i * n + c] = v;
    }
}

// ---- the same inversion with 64 pivots per launch, the rank-64 updates on the double-precision matrix cores ------------------
// With K the 64 pivots of a step, D = W[K, K], the step is  W' = C~ - L~ (D^-1 P~)  with the inputs modified so that one formula
// serves every entry:  C~ = W with the pivot rows and columns zeroed,  P~ = W[K, :] with the identity in the pivot columns,
// L~ = W[:, K] with MINUS the identity in the pivot rows  (pivot block: 0 + I D^-1 I = D^-1; pivot rows: D^-1 W[K, j]; pivot
// columns: -W[i, K] D^-1; elsewhere the rank-64 update).  One WAVE owns a tile of 96 x 32 entries: it forms U = D^-1 P~ for its
// 32 columns (4 x 2 blocks of v_mfma_f64_16x16x4_f64, D^-1 read from the side buffer in operand order) and then takes its six
// row blocks through 16 k-steps each, operands straight from global memory into registers -- no LDS, no barrier.  The result
// of the first product leaves the matrix cores in the register layout the second one wants its B operand in.
// Order of the 64 pivots inside the products: the A operand of the update (the tile's rows of the pivot columns, contiguous
// in memory along k) is read 16 bytes per lane, lane (i, g) taking k = 16 q + 4 g + {0..3}; the k-step (q, m) therefore pairs
// lane group g with pivot 16 q + 4 g + m, and U must come out of the first product with row 4 g + m of block q in register m of
// lane group g: its A operand (D^-1) has its rows permuted, row i of a block standing for pivot 4 (i & 3) + (i >> 2)
// (gj64_side_off).
// Measured (scripts/lab/gj_mfma_lab.hip, mfma_f64_rate.hip; n = 1617): the instruction issues every 62 ns per SIMD with one wave
// there, 45 ns with two -- 33 to 47 TFLOP/s over the chip, not above the vector rate -- and a launch of 64 pivots takes 40 us
// against 2 x 32 us of the vector kernel above: what it saves is a pass over the matrix, not arithmetic.
// The pivot block of the NEXT step is prepared by workgroup 0 of the launch: its four waves take
// the next 64 x 64 block through this step, invert it in LDS by four 16-pivot block steps and leave D^-1 in the other side buffer.
// Sums are formed in the order of the matrix cores: the result agrees with the vector kernels to rounding, not bit for bit
// (PADNE_GJ_VECTOR=1 selects those).
typedef double gj_v4d __attribute__((ext_vector_type(4)));
constexpr int kGjM = 64;                      // pivots per launch
constexpr int kGjTR = 96, kGjTC = 32;         // a wave's tile
__device__ __host__ inline int gj64_side_off(int a, int b) {      // D^-1[a][b] -> position in the side buffer
    const int blk = a >> 4, w = a & 15, i = ((w & 3) << 2) | (w >> 2);      // row i of the block stands for pivot 4 (i & 3) + (i >> 2)
    return (blk * 16 + (b >> 2)) * 64 + (b & 3) * 16 + i;
}

// One wave: the tile of 16 NRB rows from r0, 16 NCB columns from c0, through the step with the pivots [k0, k0 + bs).
// sink(row, col, value) receives every entry of the tile (also those outside the matrix: the caller decides).
// u_lds (may be null): the four waves of the workgroup own four row tiles of ONE column block and share U = D^-1 P~ through LDS --
// wave w forms its quarter (the 16 rows of U of pivot block w: a quarter of the 128 products every wave made for itself) and
// takes the other three from there; ALL four waves must call (the barrier), `active` tells whether the wave's row tile exists.
template <int NRB, int NCB, typename Sink>
__device__ __forceinline__ void gj64_tile(const int n, const int k0, const int bs, const double *__restrict__ in,
                                          const double *__restrict__ side, const int r0, const int c0, const int lane, Sink sink,
                                          double *u_lds = nullptr, const int w = 0, const bool active = true) {
    const int j = lane & 15, g = lane >> 4;
    auto load_a = [&](int rb, double (&a)[16]) {
        const int row = r0 + 16 * rb + j;
        const bool prow = row >= k0 && row < k0 + kGjM;
        const double *p = in + (size_t)(row < n ? row : 0) * n + k0 + 4 * g;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (!prow && row < n && 16 * q + 4 * g + 3 < bs) {
                const D2u lo = *reinterpret_cast<const D2u *>(p + 16 * q), hi = *reinterpret_cast<const D2u *>(p + 16 * q + 2);
                a[4 * q + 0] = -lo.x; a[4 * q + 1] = -lo.y; a[4 * q + 2] = -hi.x; a[4 * q + 3] = -hi.y;
            } else {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int k = 16 * q + 4 * g + m;
                    double v = 0.0;
                    if (prow) v = (row - k0 == k) ? 1.0 : 0.0;            // -(-I)
                    else if (row < n && k < bs) v = -p[16 * q + m];
                    a[4 * q + m] = v;
                }
            }
        }
    };
    auto load_c = [&](int rb, gj_v4d (&C)[NCB]) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = r0 + 16 * rb + g + 4 * r, col = c0 + 16 * cb + j;
                const bool piv = (row >= k0 && row < k0 + kGjM) || (col >= k0 && col < k0 + kGjM);
                C[cb][r] = (row < n && col < n && !piv) ? in[(size_t)row * n + col] : 0.0;
            }
    };
    double a_cur[16], a_nxt[16];
    gj_v4d C_cur[NCB], C_nxt[NCB];
    load_c(0, C_cur);
    load_a(0, a_cur);
    // U = D^-1 P~ for the tile's columns
    gj_v4d U[4][NCB];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) U[kb][cb] = (gj_v4d){0.0, 0.0, 0.0, 0.0};
    double pb[16][NCB];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const int col = c0 + 16 * cb + j, k = 4 * ks + g;
            double v = 0.0;
            if (col >= k0 && col < k0 + kGjM) v = (col - k0 == k) ? 1.0 : 0.0;
            else if (col < n && k < bs) v = in[(size_t)(k0 + k) * n + col];
            pb[ks][cb] = v;
        }
    if (u_lds == nullptr) {
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const double a = side[(kb * 16 + ks) * 64 + lane];
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) U[kb][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, pb[ks][cb], U[kb][cb], 0, 0, 0);
            }
    } else {
        gj_v4d mine[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) mine[cb] = (gj_v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const double a = side[(w * 16 + ks) * 64 + lane];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) mine[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, pb[ks][cb], mine[cb], 0, 0, 0);
        }
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) u_lds[((w * NCB + cb) * 4 + r) * 64 + lane] = mine[cb][r];
        __syncthreads();
        if (!active) return;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) U[kb][cb][r] = u_lds[((kb * NCB + cb) * 4 + r) * 64 + lane];
    }
    // the tile, a row block at a time, the next block's operands in flight
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
        if (rb + 1 < NRB) {
            load_c(rb + 1, C_nxt);
            load_a(rb + 1, a_nxt);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb)
                    C_cur[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[4 * q + m], U[q][cb][m], C_cur[cb], 0, 0, 0);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) sink(r0 + 16 * rb + g + 4 * r, c0 + 16 * cb + j, C_cur[cb][r]);
        if (rb + 1 < NRB) {
#pragma unroll
            for (int e = 0; e < 16; ++e) a_cur[e] = a_nxt[e];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) C_cur[cb] = C_nxt[cb];
        }
    }
}

// The 16 x 16 pivot block of gj64_invert_lds, inverted in place by one wave with (almost) no LDS traffic: lane 16 c + r holds
// row r, columns 4 c .. 4 c + 3.  The pivot row reaches the lanes of its 16-lane row by DPP (row_newbcast: one move per
// 32-bit half, no memory pipe), the pivot by a scalar read, and only the lane's own entry of the pivot column crosses the
// 16-lane rows through ds_bpermute -- two per pivot where gj_invert_block16 makes twelve.  Same formulas.
template <int P>
__device__ __forceinline__ double gj_row_bcast(const double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + P, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + P, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int P>
__device__ __forceinline__ void gj_pivot_rows16(double (&a)[4], const int lane) {
    constexpr int pc = P >> 2, pm = P & 3;
    const int r = lane & 15, c = lane >> 4;
    double rp[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) rp[m] = gj_row_bcast<P>(a[m]);
    const double dpp = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a[pm]), pc * 16 + P),
                                        __builtin_amdgcn_readlane(__double2loint(a[pm]), pc * 16 + P));
    const double cp = __shfl(a[pm], pc * 16 + r, 64);
    double inv = __builtin_amdgcn_rcp(dpp);
    inv = fma(inv, fma(-dpp, inv, 1.0), inv);
    inv = fma(inv, fma(-dpp, inv, 1.0), inv);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const bool pcol = c == pc && m == pm;
        const double rm = rp[m] * inv;
        a[m] = r == P ? (pcol ? inv : rm) : (pcol ? -cp * inv : a[m] - cp * rm);
    }
}
__device__ __forceinline__ void gj_invert_rows16(double (&a)[4], const int lane) {
    gj_pivot_rows16<0>(a, lane);  gj_pivot_rows16<1>(a, lane);  gj_pivot_rows16<2>(a, lane);  gj_pivot_rows16<3>(a, lane);
    gj_pivot_rows16<4>(a, lane);  gj_pivot_rows16<5>(a, lane);  gj_pivot_rows16<6>(a, lane);  gj_pivot_rows16<7>(a, lane);
    gj_pivot_rows16<8>(a, lane);  gj_pivot_rows16<9>(a, lane);  gj_pivot_rows16<10>(a, lane); gj_pivot_rows16<11>(a, lane);
    gj_pivot_rows16<12>(a, lane); gj_pivot_rows16<13>(a, lane); gj_pivot_rows16<14>(a, lane); gj_pivot_rows16<15>(a, lane);
}

// In-place inversion of a 64 x 64 block in LDS by one workgroup: four block steps of 16 pivots.  The 16 x 16 pivot block is
// inverted by wave 0 (gj_invert_rows16); the rest of a step is the unified formula of the big step on 16 x 16 blocks,
//     next = C~ - L~ (D16 P~)        (C~: pivot rows and columns zeroed, P~: pivot rows with the identity in the pivot columns,
//                                     L~: pivot columns with MINUS the identity in the pivot rows)
// on the matrix cores: wave w owns the 16 columns 16 w .., forms R = D16 P~ for them (four v_mfma_f64_16x16x4_f64; the rows of
// D16 are read permuted so that the result's register layout is the B-operand layout of the second product, as in gj64_tile)
// and takes its four row blocks through four k-steps each.  Everything a wave needs of the old block that does not depend on
// D16 is in registers before the barrier behind the inversion.  Until round 5 both products were vector FMAs fed from LDS
// (~270 LDS reads per thread and block step: the step was bound by LDS bandwidth, 7 us of it beside a 1.6 us inversion), and
// a launch of the big step waited 35 us for this workgroup while its tiles needed 26.  N0 holds the block and receives the
// inverse; N1, D16 are scratch.  Ends with a barrier.  Sums in the order of the matrix cores.
constexpr int kGjLd = kGjM + 1;
__device__ __forceinline__ void gj64_invert_lds(double (*N0)[kGjLd], double (*N1)[kGjLd], double (*D16)[kGjBlock]) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, j = lane & 15, g = lane >> 4;
    double (*cur)[kGjLd] = N0, (*nxt)[kGjLd] = N1;
    for (int bb = 0; bb < kGjM / kGjBlock; ++bb) {
        const int pb = bb * kGjBlock;
        gj_v4d acc[4];
        double aop[4][4], bop[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[rb][r] = (rb == bb || w == bb) ? 0.0 : cur[16 * rb + g + 4 * r][16 * w + j];
            // (A operand: row j of the block, pivot 4 g + m in k-step m; minus L~)
#pragma unroll
            for (int m = 0; m < 4; ++m) aop[rb][m] = rb == bb ? (j == 4 * g + m ? 1.0 : 0.0) : -cur[16 * rb + j][pb + 4 * g + m];
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) bop[m] = w == bb ? (4 * m + g == j ? 1.0 : 0.0) : cur[pb + 4 * m + g][16 * w + j];
        if (t < 64) {
            double dd[4];                                  // (row j, columns 4 g .. 4 g + 3: the layout of gj_invert_rows16)
#pragma unroll
            for (int m = 0; m < 4; ++m) dd[m] = cur[pb + j][pb + 4 * g + m];
            gj_invert_rows16(dd, lane);
#pragma unroll
            for (int m = 0; m < 4; ++m) D16[j][4 * g + m] = dd[m];
        }
        __syncthreads();
        gj_v4d rr = {0.0, 0.0, 0.0, 0.0};
        const int prow = 4 * (j & 3) + (j >> 2);          // row j of the first product stands for pivot 4 (j & 3) + (j >> 2)
#pragma unroll
        for (int m = 0; m < 4; ++m) rr = __builtin_amdgcn_mfma_f64_16x16x4f64(D16[prow][4 * m + g], bop[m], rr, 0, 0, 0);
        // rr[m] of lane (g, j) = R[4 g + m][16 w + j]: the B operand of k-step m
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[rb][m], rr[m], acc[rb], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) nxt[16 * rb + g + 4 * r][16 * w + j] = acc[rb][r];
        }
        __syncthreads();
        double (*sw)[kGjLd] = cur;
        cur = nxt;
        nxt = sw;
    }
    // (four steps: the result is back in N0)
}

// D^-1 of the first step's pivot block -> side (operand order)
__global__ __launch_bounds__(256) void gj64_prepare(int n, int k0, int bs, const double *__restrict__ in, double *__restrict__ side) {
    __shared__ double N0[kGjM][kGjLd], N1[kGjM][kGjLd], D16[kGjBlock][kGjBlock];
    const int t = threadIdx.x;
    for (int e = t; e < kGjM * kGjM; e += 256) {
        const int a = e / kGjM, b = e % kGjM;
        N0[a][b] = (a < bs && b < bs) ? in[(size_t)(k0 + a) * n + k0 + b] : (a == b ? 1.0 : 0.0);
    }
    __syncthreads();
    gj64_invert_lds(N0, N1, D16);
    for (int e = t; e < kGjM * kGjM; e += 256) side[gj64_side_off(e / kGjM, e % kGjM)] = N0[e / kGjM][e % kGjM];
}

// One step: workgroups 1.. carry four wave tiles each; workgroup 0 prepares the pivot block of the next step (next_bs of its
// pivots exist; 0: there is no next step).
__global__ __launch_bounds__(256) void gj64_step(int n, int k0, int bs, const double *__restrict__ in, double *__restrict__ out,
                                                 const double *__restrict__ side, double *__restrict__ side_next, int next_bs,
                                                 int n_ct, int n_tiles) {
    __shared__ double N0[kGjM][kGjLd], N1[kGjM][kGjLd], D16[kGjBlock][kGjBlock];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (blockIdx.x > 0) {
        // the four waves of a workgroup: four row tiles of one column block (they share U through LDS: N0 is free here)
        const int b = (int)blockIdx.x - 1, n_rt = n_tiles / n_ct;
        const int c0 = (b % n_ct) * kGjTC, rt = (b / n_ct) * 4 + w;
        static_assert(sizeof(N0) >= sizeof(double) * 4 * (kGjTC / 16) * 4 * 64, "room for U");
        gj64_tile<kGjTR / 16, kGjTC / 16>(n, k0, bs, in, side, rt * kGjTR, c0, lane, [&](int row, int col, double v) {
            if (row < n && col < n) out[(size_t)row * n + col] = v;
        }, &N0[0][0], w, rt < n_rt);
        return;
    }
    if (next_bs <= 0) return;
    // the next pivot block as this step leaves it: 64 rows x 16 columns per wave, into LDS, padded with the identity
    const int kn = k0 + kGjM;
    gj64_tile<kGjM / 16, 1>(n, k0, bs, in, side, kn, kn + 16 * w, lane, [&](int row, int col, double v) {
        const int a = row - kn, b = col - kn;
        N0[a][b] = (a < next_bs && b < next_bs) ? v : (a == b ? 1.0 : 0.0);
    });
    __syncthreads();
    gj64_invert_lds(N0, N1, D16);
    for (int e = t; e < kGjM * kGjM; e += 256) side_next[gj64_side_off(e / kGjM, e % kGjM)] = N0[e / kGjM][e % kGjM];
}

// y = Inv * b : one wave per row
template <typename T>
__global__ __launch_bounds__(256) void dense_gemv(int n_rows, int n, const T *__restrict__ inv,
                                                  const T *__restrict__ b, T *__restrict__ y) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n_rows) return;
    // four independent partial sums per lane: the loads of 256 columns are in flight together (a row is 1617 columns on
    // the coarsest level of config C4: one dependent load after the other took 13 us for 10 MB)
    const T *ir = inv + (size_t)row * n;
    T s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int c = lane;
    for (; c + 192 < n; c += 256) {
        s0 += ir[c] * b[c];
        s1 += ir[c + 64] * b[c + 64];
        s2 += ir[c + 128] * b[c + 128];
        s3 += ir[c + 192] * b[c + 192];
    }
    for (; c < n; c += 64) s0 += ir[c] * b[c];
    T s = (s0 + s1) + (s2 + s3);
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) y[row] = s;
}

// the same product as the WHOLE preconditioner of a system the dense inverse takes as it stands (a hierarchy of one level:
// boards of a few hundred to two thousand unknowns): z = Inv r with the partial sums of r . z, one per workgroup of four rows
__global__ __launch_bounds__(256) void dense_gemv_dot(int n, const double *__restrict__ inv, const double *__restrict__ r,
                                                      double *__restrict__ z, double *__restrict__ partials,
                                                      const int *__restrict__ done_flag) {
    __shared__ double red[4];
    if (done_flag != nullptr && *done_flag != 0) return;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + w;
    double rz = 0.0;
    if (row < n) {
        const double *ir = inv + (size_t)row * n;
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        int c = lane;
        for (; c + 192 < n; c += 256) {
            s0 += ir[c] * r[c];
            s1 += ir[c + 64] * r[c + 64];
            s2 += ir[c + 128] * r[c + 128];
            s3 += ir[c + 192] * r[c + 192];
        }
        for (; c < n; c += 64) s0 += ir[c] * r[c];
        double s = (s0 + s1) + (s2 + s3);
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) {
            z[row] = s;
            rz = r[row] * s;
        }
    }
    if (lane == 0) red[w] = rz;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

template <typename T>
__global__ void scale_dinv_kernel(long long n, T c, const T *__restrict__ dinv, const T *__restrict__ b,
                                  T *__restrict__ x, const int *__restrict__ done_flag) {
    if (done_flag != nullptr && *done_flag != 0) return;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        x[i] = c * dinv[i] * b[i];
}

// entry of the single-precision cycle: b = r / ||b_rhs|| in float (the cycle is linear, the last stage multiplies
// the norm back in; keeps every intermediate far from the float range limits whatever the units of the system),
// and the first Jacobi sweep from a zero guess
__global__ void amg_entry_f32_kernel(long long n, const double *__restrict__ r, const double *__restrict__ bb2, float c,
                                     const float *__restrict__ dinv, float *__restrict__ b, float *__restrict__ x,
                                     const int *__restrict__ done_flag) {
    if (done_flag != nullptr && *done_flag != 0) return;
    double s_inv = 1.0;
    if (bb2 != nullptr) {
        const double s2 = *bb2;
        if (s2 > 0.0) s_inv = 1.0 / sqrt(s2);
    }
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = (float)(r[i] * s_inv);
        b[i] = v;
        x[i] = c * dinv[i] * v;
    }
}

__global__ void f32_copy_amg(long long n, const double *__restrict__ src, float *__restrict__ dst) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (float)src[i];
}

// per-workgroup min / max of |v| (range check before switching a hierarchy to single precision)
__global__ __launch_bounds__(256) void abs_range_kernel(long long n, const double *__restrict__ v, double *__restrict__ mins,
                                                        double *__restrict__ maxs) {
    __shared__ double lo[256], hi[256];
    double a = 1e300, b = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const double t = fabs(v[i]);
        a = t < a ? t : a;
        b = (t > b || !(t == t)) ? t : b;
    }
    lo[threadIdx.x] = a;
    hi[threadIdx.x] = b;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            lo[threadIdx.x] = fmin(lo[threadIdx.x], lo[threadIdx.x + o]);
            const double h2 = hi[threadIdx.x + o];
            if (h2 > hi[threadIdx.x] || !(h2 == h2)) hi[threadIdx.x] = h2;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        mins[blockIdx.x] = lo[0];
        maxs[blockIdx.x] = hi[0];
    }
}

// smallest of mins, largest of maxs (NaN if any is NaN), by one workgroup: 16 bytes for the host
__global__ __launch_bounds__(256) void fold_range_kernel(const double *__restrict__ mins, const double *__restrict__ maxs, int g,
                                                         double *__restrict__ out2) {
    __shared__ double lo_s[256], hi_s[256];
    double lo = 1e300, hi = 0.0;
    for (int q = threadIdx.x; q < g; q += 256) {
        const double a = mins[q], b = maxs[q];
        lo = a < lo ? a : lo;
        if (b > hi || !(b == b)) hi = b;
    }
    lo_s[threadIdx.x] = lo;
    hi_s[threadIdx.x] = hi;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            const double a = lo_s[threadIdx.x + off], b = hi_s[threadIdx.x + off];
            if (a < lo_s[threadIdx.x]) lo_s[threadIdx.x] = a;
            if (!(hi_s[threadIdx.x] == hi_s[threadIdx.x])) {
            } else if (b > hi_s[threadIdx.x] || !(b == b)) {
                hi_s[threadIdx.x] = b;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out2[0] = lo_s[0];
        out2[1] = hi_s[0];
    }
}

// ---- host side ----------------------------------------------------------------------------------

static bool amg_verbose();

// maxima of up to two arrays of non-negative partial results, by one workgroup (so that 16 bytes go to the host)
__global__ __launch_bounds__(256) void fold_max2_kernel(const double *__restrict__ a, const double *__restrict__ b, int g,
                                                        double *__restrict__ out2) {
    __shared__ double red[2][256];
    double ma = 0.0, mb = 0.0;
    for (int q = threadIdx.x; q < g; q += 256) {
        ma = a[q] > ma ? a[q] : ma;
        if (b != nullptr) mb = b[q] > mb ? b[q] : mb;
    }
    red[0][threadIdx.x] = ma;
    red[1][threadIdx.x] = mb;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            red[0][threadIdx.x] = fmax(red[0][threadIdx.x], red[0][threadIdx.x + off]);
            red[1][threadIdx.x] = fmax(red[1][threadIdx.x], red[1][threadIdx.x + off]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out2[0] = red[0][0];
        out2[1] = red[1][0];
    }
}

static int gershgorin(padne_ctx *ctx, const padne_csr *A, double *lambda, bool filtered = false) {
    const int g = (int)std::min<long long>((A->n_rows + 255) / 256, 1024);
    double *part = ctx->partials + 6 * kMaxPartials;
    if (filtered)
        hipLaunchKernelGGL(gershgorin_filtered_kernel, dim3(g > 0 ? g : 1), dim3(256), 0, ctx->stream, (int)A->n_rows,
                           A->rowptr, A->cols, A->vals, A->dinv, kTheta * kTheta, part);
    else
        hipLaunchKernelGGL(gershgorin_kernel, dim3(g > 0 ? g : 1), dim3(256), 0, ctx->stream, (int)A->n_rows, A->rowptr,
                           A->vals, A->dinv, part);
    hipLaunchKernelGGL(fold_max2_kernel, dim3(1), dim3(256), 0, ctx->stream, (const double *)part, (const double *)nullptr,
                       g > 0 ? g : 1, part + kMaxPartials - 2);
    PADNE_HIP_CHECK(hipGetLastError());
    double h[2] = {0.0, 0.0};
    PADNE_TRY(read_back(ctx, part + kMaxPartials - 2, sizeof(h), h));
    double m = h[0];
    if (!(m > 0.0) || !(m < 1e6)) m = 2.0;
    *lambda = m;
    return PADNE_OK;
}

// aggregates of A -> device array agg[n], count n_agg
static int aggregate(padne_ctx *ctx, Scratch &sc, const padne_csr *A, int **agg_out, int *n_agg, double *lambda_f = nullptr,
                     double *lambda_plain = nullptr, unsigned char **spos_out = nullptr, int **scol_out = nullptr,
                     float *vals32_out = nullptr) {
    hipStream_t s = ctx->stream;
    const int n = (int)A->n_rows;
    const double theta2 = kTheta * kTheta;
    signed char *state = nullptr;
    unsigned int *w0 = nullptr, *w1 = nullptr, *w2 = nullptr;
    int *flag = nullptr, *scan = nullptr, *agg0 = nullptr, *agg1 = nullptr, *counter = nullptr;
    PADNE_TRY(sc.alloc(&state, (size_t)n));
    PADNE_TRY(sc.alloc(&w0, (size_t)n));
    PADNE_TRY(sc.alloc(&w1, (size_t)n));
    PADNE_TRY(sc.alloc(&w2, (size_t)n));
    PADNE_TRY(sc.alloc(&flag, (size_t)n + 1));
    PADNE_TRY(sc.alloc(&scan, (size_t)n + 1));
    PADNE_TRY(sc.alloc(&agg0, (size_t)n));
    PADNE_TRY(sc.alloc(&agg1, (size_t)n));
    // one zeroed counter per round (at most 256 rounds) instead of a memset in front of every round
    constexpr int kMaxRounds = 256;
    PADNE_TRY(sc.alloc(&counter, (size_t)2 * kMaxRounds + 4));
    const dim3 g(nblk(n)), b(256);
    // strength graph on the pattern of A, built once per level
    const int *srow = A->rowptr;
    int *scol = nullptr;
    PADNE_TRY(sc.alloc(&scol, (size_t)(A->nnz > 0 ? A->nnz : 1)));
    const int n_wt = (n + 63) / 64;
    int gm_x = std::min(2048, (n_wt + 3) / 4 > 0 ? (n_wt + 3) / 4 : 1);
    if (gm_x >= kNumXcd) gm_x -= gm_x % kNumXcd;      // (a multiple of the XCDs: the tile kernels sweep one slab per XCD)
    const dim3 gm((unsigned)gm_x);
    // lambda_f: the Gershgorin bound of the filtered operator comes out of the same pass
    double *bound_part = lambda_f != nullptr ? ctx->partials + 6 * kMaxPartials : nullptr;
    double *bound2 = nullptr;            // their maxima: filtered | plain
    if (lambda_f != nullptr) PADNE_TRY(sc.alloc(&bound2, 2));
    // x-window plan of A (fine-level band matrices): one byte per entry and LDS-staged neighbours in the rounds below
    const bool xw = A->xw_state == 1 && A->xw_run <= 85;
    unsigned char *spos = nullptr;
    if (xw) {
        PADNE_TRY(sc.alloc(&spos, (size_t)A->nnz + kPadNnz));
    }
    static_assert(kPadNnz >= 2 * kMaxRounds + 4, "one grid for everything the start clears");
    hipLaunchKernelGGL(mis_init_words, dim3(nblk(std::max(n, kPadNnz))), b, 0, s, n, w0, state, counter, 2 * kMaxRounds + 4,
                       spos != nullptr ? spos + A->nnz : (unsigned char *)nullptr, kPadNnz);
    hipLaunchKernelGGL(strength_mark, gm, b, 0, s, n, n_wt, A->rowptr, A->cols, A->vals, A->dinv, theta2, scol, bound_part,
                       xw ? A->xw_desc : (const int4 *)nullptr, xw ? (const unsigned char *)A->xw_lidx : (const unsigned char *)nullptr,
                       xw ? A->xw_run : 0, spos, vals32_out);
    PADNE_HIP_CHECK(hipGetLastError());
    // one round = one-hop maxima, two-hop maxima, decision; on a windowed matrix the decision rides on the second pass;
    // small levels with long rows (a few thousand rows of dozens of entries) take a wave per row
    const bool long_rows = !xw && n <= 65536 && A->nnz >= 24LL * n;
    auto launch_round = [&](int *open_counter) {
        if (xw) {
            hipLaunchKernelGGL((nbr_max_xw<unsigned int, false>), gm, b, 0, s, n, n_wt, srow, scol, spos, A->xw_desc, A->xw_run,
                               (const unsigned int *)w0, w1, (unsigned int *)nullptr, (signed char *)nullptr, (int *)nullptr);
            hipLaunchKernelGGL((nbr_max_xw<unsigned int, true>), gm, b, 0, s, n, n_wt, srow, scol, spos, A->xw_desc, A->xw_run,
                               (const unsigned int *)w1, (unsigned int *)nullptr, w0, state, open_counter);
        } else if (long_rows) {
            hipLaunchKernelGGL(nbr_max_wpr<unsigned int>, dim3((unsigned)((n + 3) / 4)), b, 0, s, n, srow, (const int *)scol,
                               (const unsigned int *)w0, w1);
            hipLaunchKernelGGL(nbr_max_wpr<unsigned int>, dim3((unsigned)((n + 3) / 4)), b, 0, s, n, srow, (const int *)scol,
                               (const unsigned int *)w1, w2);
            hipLaunchKernelGGL(mis_decide, dim3(std::min<unsigned>(g.x, 1024u)), b, 0, s, n, w0, w2, state, open_counter);
        } else {
            hipLaunchKernelGGL(nbr_max<unsigned int>, gm, b, 0, s, n, n_wt, srow, scol, w0, w1);
            hipLaunchKernelGGL(nbr_max<unsigned int>, gm, b, 0, s, n, n_wt, srow, scol, w1, w2);
            hipLaunchKernelGGL(mis_decide, dim3(std::min<unsigned>(g.x, 1024u)), b, 0, s, n, w0, w2, state, open_counter);
        }
    };
    if (lambda_f != nullptr)
        hipLaunchKernelGGL(fold_max2_kernel, dim3(1), dim3(256), 0, s, (const double *)bound_part,
                           (const double *)(bound_part + kMaxPartials), (int)gm.x, bound2);
    // the two bounds travel to the host with the first open count
    double h_bound2[2] = {0.0, 0.0};
    bool bound_pending = lambda_f != nullptr;
    auto count_back = [&](const int *dev_count, int *host_count) -> int {
        if (!bound_pending) return read_back(ctx, dev_count, sizeof(int), host_count);
        bound_pending = false;
        return read_back2(ctx, dev_count, sizeof(int), host_count, bound2, sizeof(h_bound2), h_bound2);
    };
    int open_count = n;
    int round = 0;
    // the compact rounds pay off on sparse rows only: a direct two-hop maximum visits (nnz/row)^2 words per vertex
    const bool compact_ok = A->nnz <= 16LL * n;
    // Rounds are enqueued in batches between two reads of the open count (a host synchronisation costs ~30 us, as
    // much as a round on a small level): one round per batch while a pass over the level is expensive, four on the
    // small levels, where a superfluous round after the last vertex was decided is cheaper than the wait.
    // (the first two rounds are never the last ones of a large level -- the test below asks for two -- so they go together)
    const int full_batch = n > 200000 ? 1 : 4;
    while (round < kMaxRounds && open_count > 0 && (!compact_ok || round < 2 || open_count > n / 8)) {
        const int reps = (round == 0 && compact_ok && full_batch < 2) ? 2 : full_batch;
        for (int rep = 0; rep < reps && round < kMaxRounds; ++rep, ++round) {
            launch_round(counter + round);
        }
        PADNE_HIP_CHECK(hipGetLastError());
        PADNE_TRY(count_back(counter + round - 1, &open_count));
    }
    if (open_count > 0) {
        // compact rounds; the word buffers of the full passes are free now and serve as the two lists (w1, w2 hold
        // n entries each, the open vertices are fewer)
        int *list_a = (int *)w1, *list_b = (int *)w2;
        unsigned int *m2 = nullptr;
        PADNE_TRY(sc.alloc(&m2, (size_t)open_count));
        // list lengths: one (pre-zeroed) word per compact round behind the words of the full rounds
        int *counters = counter + kMaxRounds;
        hipLaunchKernelGGL(mis_collect_open, dim3(std::min<unsigned>(g.x, 1024u)), b, 0, s, n, state, list_a, counters);
        PADNE_HIP_CHECK(hipGetLastError());
        int cnt = open_count, cur = 0;       // counters[cur]: length of list_a, on the device
        while (cur + 1 < kMaxRounds && cnt > 0) {
            if (cnt <= kMisTailCap) {
                // the rest of the rounds in one launch of one workgroup
                hipLaunchKernelGGL(mis_tail_rounds, dim3(1), dim3(1024), 0, s, (const int *)(counters + cur), list_a, list_b, srow,
                                   (const int *)scol, w0, state, m2, counters + cur + 1, kMaxRounds - cur - 1);
                PADNE_HIP_CHECK(hipGetLastError());
                ++cur;
                round += kMaxRounds;                       // (not counted one by one: `round` only bounds the loops here)
                PADNE_TRY(count_back(counters + cur, &cnt));
                break;
            }
            const dim3 gl(nblk(cnt));
            // four rounds between two looks at the host: the grids of the second and third are those of the first, their
            // lists are shorter (the kernels read the length on the device), and what an idle stream costs while the
            // host looks is more than the lanes that find nothing to do
            const int batch = 4;
            for (int rep = 0; rep < batch && cur + 1 < kMaxRounds; ++rep, ++round) {
                hipLaunchKernelGGL(mis_two_hop_max, dim3(nblk((long long)cnt * kHopLanes)), b, 0, s, counters + cur, list_a, srow, scol, w0, m2);
                hipLaunchKernelGGL(mis_decide_list, gl, b, 0, s, counters + cur, list_a, m2, w0, state, list_b, counters + cur + 1);
                std::swap(list_a, list_b);
                ++cur;
            }
            PADNE_HIP_CHECK(hipGetLastError());
            PADNE_TRY(count_back(counters + cur, &cnt));
        }
        open_count = cnt;
    }
    PADNE_REQUIRE(open_count == 0, "independent-set rounds did not terminate");
    if (lambda_f != nullptr) {
        if (bound_pending) PADNE_TRY(read_back(ctx, bound2, sizeof(h_bound2), h_bound2));      // no round ran (n == 0)
        double m = h_bound2[0], mp = h_bound2[1];
        if (!(m > 0.0) || !(m < 1e6)) m = 2.0;                           // as gershgorin()
        if (!(mp > 0.0) || !(mp < 1e6)) mp = 2.0;
        *lambda_f = m;
        if (lambda_plain != nullptr) *lambda_plain = mp;
    }
    // number the roots
    hipLaunchKernelGGL(flag_state, g, b, 0, s, n, state, flag, 1);
    // The host needs the two totals (roots, singletons) only as their sum: both scans are queued, the kernel that numbers
    // the singletons reads the number of roots on the device, and the host looks once
    ScanTicket t_roots, t_single;
    long long h_roots[2] = {0, 0}, h_single[2] = {0, 0};
    PADNE_TRY(scan_i32_begin(ctx, flag, scan, n, &t_roots, true));
    hipLaunchKernelGGL(agg_from_roots, g, b, 0, s, n, state, scan, agg0);
    hipLaunchKernelGGL(agg_join, g, b, 0, s, n, srow, scol, A->vals, agg0, agg1);
    hipLaunchKernelGGL(agg_join, g, b, 0, s, n, srow, scol, A->vals, agg1, agg0);
    hipLaunchKernelGGL(flag_unaggregated, g, b, 0, s, n, agg0, flag);
    int rc_scan = scan_i32_begin(ctx, flag, scan, n, &t_single, true);
    if (rc_scan == PADNE_OK && t_roots.bs != nullptr)
        hipLaunchKernelGGL(agg_singletons, g, b, 0, s, n, scan, (const long long *)(t_roots.bs + t_roots.nb), agg0);
    const int rc_roots = scan_i32_end(ctx, &t_roots, h_roots);
    if (rc_scan == PADNE_OK) rc_scan = scan_i32_end(ctx, &t_single, h_single);
    PADNE_TRY(rc_roots);
    PADNE_TRY(rc_scan);
    PADNE_HIP_CHECK(hipGetLastError());
    const long long n_roots = h_roots[0], n_single = h_single[0];
    PADNE_REQUIRE(n_roots >= 0 && n_single >= 0 && n_roots + n_single < 2147483647LL, "aggregate count out of range");
    *agg_out = agg0;
    *n_agg = (int)(n_roots + n_single);
    if (spos_out != nullptr) *spos_out = spos;      // lives in the caller's scratch, like agg
    if (scol_out != nullptr) *scol_out = scol;
    return PADNE_OK;
}

// LDS variant of the prolongator rows: sorted (aggregate, value) lists per lane, products added in the
// order prolong_fill + merge would add them.  Rows with more than CAP aggregates return -1 and take the
// slot path below.
template <int CAP>
__global__ __launch_bounds__(128) void prolong_rows_lds(int n, const int *__restrict__ rowptr, const int *__restrict__ cols,
                                                        const double *__restrict__ vals, const double *__restrict__ dinv,
                                                        double theta2, double omega, const int *__restrict__ agg,
                                                        const int *__restrict__ slot_ptr, long long *__restrict__ key,
                                                        double *__restrict__ val, int *__restrict__ row_len) {
    static_assert(CAP % 2 == 0, "keys are kept in pairs");
    __shared__ int2 Kc[CAP / 2][128];
    __shared__ double Vc[CAP][128];
    const int t = threadIdx.x;
    const int i = blockIdx.x * 128 + t;
    if (i >= n) return;
    const double di = dinv[i];
    double dF = 1.0 / di;
    const int k0 = rowptr[i], k1 = rowptr[i + 1];
    // Both walks over the row take eight entries at a time: their columns and values in six loads, then the eight
    // neighbours' 1/diag (and aggregates) asked for together -- entry by entry every entry waited for its own chain of
    // loads, twice per row.  Same operations in the same order on the same values.
    for (int kb = k0; kb < k1; kb += 8) {
        const int4 ca = load_i4_unaligned(cols + kb), cb = load_i4_unaligned(cols + kb + 4);
        const double2 v0 = load_d2_unaligned(vals + kb), v1 = load_d2_unaligned(vals + kb + 2),
                      v2 = load_d2_unaligned(vals + kb + 4), v3 = load_d2_unaligned(vals + kb + 6);
        const int j[8] = {ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w};
        const double a[8] = {v0.x, v0.y, v1.x, v1.y, v2.x, v2.y, v3.x, v3.y};
        double dj[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) dj[u] = kb + u < k1 ? dinv[j[u]] : 1.0;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (kb + u < k1 && j[u] != i && !strong(a[u], di, dj[u], theta2)) dF += a[u];
    }
    const bool keep_all = !(dF * di > 0.05);
    if (keep_all) dF = 1.0 / di;
    const double w = -omega / dF;
    const int ai = agg[i];
    // The row's list in order of APPEARANCE (as in spgemm_rows_lds_pipe: a contribution is compared with the list as far
    // as the longest list of the wave reaches and added or appended under a predicate; a sorted list cost a search loop, a
    // shift loop and three-way branching per contribution, every lane with its own trip counts).  The sums of an aggregate
    // still add up in the order of the row; the list is put in order once, at the end.
#pragma unroll
    for (int u = 0; u < CAP / 2; ++u) Kc[u][t] = make_int2(-1, -1);
    int m = 1;
    reinterpret_cast<int *>(&Kc[0][t])[0] = ai;
    Vc[0][t] = 1.0;
    bool overflow = false;
    auto insert = [&](const bool act, const int c, const double v) {
        int pos = -1;
#pragma unroll
        for (int u = 0; u < CAP; u += 2) {
            if (__all(u >= m)) break;
            const int2 kk = Kc[u >> 1][t];
            pos = kk.x == c ? u : pos;
            pos = kk.y == c ? u + 1 : pos;
        }
        if (act && !overflow) {
            const bool found = pos >= 0;
            if (!found && m == CAP) {
                overflow = true;
            } else {
                const int p = found ? pos : m;
                const double base = found ? Vc[p][t] : -0.0;      // -0 + v == v: a new aggregate starts with its first contribution
                Vc[p][t] = base + v;
                if (!found) {
                    reinterpret_cast<int *>(&Kc[p >> 1][t])[p & 1] = c;
                    ++m;
                }
            }
        }
    };
    for (int kb = k0; kb < k1 && !overflow; kb += 8) {
        const int4 ca = load_i4_unaligned(cols + kb), cb = load_i4_unaligned(cols + kb + 4);
        const double2 v0 = load_d2_unaligned(vals + kb), v1 = load_d2_unaligned(vals + kb + 2),
                      v2 = load_d2_unaligned(vals + kb + 4), v3 = load_d2_unaligned(vals + kb + 6);
        const int j[8] = {ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w};
        const double a[8] = {v0.x, v0.y, v1.x, v1.y, v2.x, v2.y, v3.x, v3.y};
        double dj[8];
        int aj[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool on = kb + u < k1;
            dj[u] = on ? dinv[j[u]] : 1.0;
            aj[u] = on ? agg[j[u]] : 0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool on = kb + u < k1;
            const bool diag = on && j[u] == i;
            const bool act = on && (diag || keep_all || strong(a[u], di, dj[u], theta2));
            if (__any(act)) insert(act, diag ? ai : aj[u], diag ? -omega : w * a[u]);
        }
    }
    if (overflow) {
        row_len[i] = -1;
        return;
    }
    // aggregates whose contributions cancelled are dropped (their key sorts behind all others), the rest goes out in order
    int o = 0;
    for (int u = 0; u < m; ++u) {
        if (Vc[u][t] != 0.0) ++o;
        else reinterpret_cast<int *>(&Kc[u >> 1][t])[u & 1] = 0x7fffffff;
    }
    long long *K = key + slot_ptr[i];
    double *V = val + slot_ptr[i];
    for (int u = 0; u < CAP; ++u) {
        if (__all(u >= m)) break;
        const int ku = u < m ? reinterpret_cast<const int *>(&Kc[u >> 1][t])[u & 1] : 0x7fffffff;
        int rank = 0;
        for (int q = 0; q < CAP; q += 2) {
            if (__all(q >= m)) break;
            const int2 kk = Kc[q >> 1][t];
            rank += (q < m && kk.x < ku) ? 1 : 0;
            rank += (q + 1 < m && kk.y < ku) ? 1 : 0;
        }
        if (ku != 0x7fffffff) {
            K[rank] = (long long)ku << 32;
            V[rank] = Vc[u][t];
        }
    }
    row_len[i] = o;
}

// The prolongator rows of a matrix with an x-window plan (the fine level: 10 M rows of 7 entries).  The one-thread-per-
// row kernel above walks its row with strided, dependent global loads, keeps a list per lane in LDS and parks the rows in
// 16-byte slots that a second kernel compacts.  Here a wave streams the 64 rows of a tile into LDS with coalesced loads, stages the aggregates of the
// tile's three runs of neighbours, and every lane merges its row IN REGISTERS without data-dependent control flow: the
// row's contributions (aggregate, value) sit in fixed slots, `first` marks the first slot of every aggregate, its sum
// runs over the later slots of the same aggregate in slot order (the order prolong_rows_lds adds them in), and the
// rank of an aggregate among the kept ones is where its entry goes -- all O(m^2) compares on m <= 14 registers.
// One pass (prolong_rows_xw below: the rows of a tile staged back to back, then moved); tiles without a plan (rows with far
// couplings) take cols / scol from global memory.
constexpr int kPxSlots = 14;        // identity + up to 13 entries of the row of A (longer rows: the whole matrix falls back)
constexpr int kPxChunk = 640;       // entries of a 64-row tile staged at once (10 per row)
// (merge: the row's kept aggregates c[k] / sums / keep flags in registers, returns their number; store: an entry's place is
// the number of kept aggregates below it)
template <int SLOTS>
__device__ __forceinline__ int prolong_row_merge(const bool live, const int w, const int rs, const int re, const int k0, const int self,
                                                 const bool windowed, const double di, const double omega, const int r,
                                                 const unsigned char (*Ls)[kPxChunk], const unsigned char (*Ss)[kPxChunk],
                                                 const double (*Vs)[kPxChunk], const int (*As)[3 * 88],
                                                 const int *__restrict__ cols, const int *__restrict__ scol,
                                                 const double *__restrict__ vals, const int *__restrict__ agg,
                                                 int (&c)[SLOTS], double (&sum)[SLOTS], bool (&keep)[SLOTS]) {
    // (every lane of the wave runs this, a lane without a row -- `live` false -- as a row of no entries that keeps nothing:
    // straight-line code up to the prefix sum of the lengths, which all lanes take part in)
    const int len = live ? re - rs : 0;
    // the row in registers: slot 0 is the identity part of P, slot k + 1 entry k of the row of A
    int el[SLOTS - 1], es[SLOTS - 1];
    double ev[SLOTS - 1];
    double dF = 1.0 / di;
#pragma unroll
    for (int k = 0; k < SLOTS - 1; ++k) {
        const bool in = k < len;
        // (a tile without a plan -- rows with far couplings, one or two per cent of the tiles -- reads its row from global memory)
        el[k] = in ? (windowed ? (int)Ls[w][rs - k0 + k] : cols[rs + k]) : -1;
        es[k] = in ? (windowed ? (int)Ss[w][rs - k0 + k] : scol[rs + k]) : -1;
        ev[k] = in ? (windowed ? Vs[w][rs - k0 + k] : vals[rs + k]) : 0.0;
        if (in && el[k] != self && es[k] == self) dF += ev[k];       // weak entry: lumped into the diagonal
    }
    const bool keep_all = !(dF * di > 0.05);
    if (keep_all) dF = 1.0 / di;
    const double wgt = -omega / dF;
    const int ai = live ? (windowed ? As[w][self] : agg[r]) : 0;
    double v[SLOTS];
    c[0] = ai;
    v[0] = 1.0;
#pragma unroll
    for (int k = 0; k < SLOTS - 1; ++k) {
        const bool in = k < len;
        const bool diag = in && el[k] == self;
        const bool used = in && (diag || keep_all || es[k] != self);
        int cj = 0x7fffffff;                                         // unused slot: sorts last, matches nothing
        if (used && !diag) cj = windowed ? As[w][el[k]] : agg[el[k]];
        c[k + 1] = diag ? ai : cj;
        v[k + 1] = diag ? -omega : (used ? wgt * ev[k] : 0.0);
    }
    int o = 0;
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
        bool first = c[k] != 0x7fffffff;
#pragma unroll
        for (int j = 0; j < k; ++j) first = first && c[j] != c[k];
        double t = v[k];
#pragma unroll
        for (int j = k + 1; j < SLOTS; ++j) t = c[j] == c[k] ? t + v[j] : t;
        sum[k] = t;
        keep[k] = live && first && t != 0.0;
        o += keep[k] ? 1 : 0;
    }
    return o;
}

template <int SLOTS>
__device__ __forceinline__ void prolong_row_store(const int (&c)[SLOTS], const double (&sum)[SLOTS], const bool (&keep)[SLOTS],
                                                  const int base, int *__restrict__ out_cols, double *__restrict__ out_vals) {
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
        int rank = 0;
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) rank += (keep[j] && c[j] < c[k]) ? 1 : 0;
        if (keep[k]) {
            out_cols[base + rank] = c[k];
            out_vals[base + rank] = sum[k];
        }
    }
}

// ONE pass: the finished rows of a tile go back to back into a staging area, from the tile's own place there -- the tile's
// first entry of A plus its first row number: a row of P has at most one entry more than its row of A, so the places of
// the tiles need no scan -- with their lengths; the scan of the lengths then gives the row pointers and `prolong_unstage`
// moves every tile's run (contiguous on both sides) to its place in the CSR arrays.  Before: a counting pass with the
// same arithmetic as the filling pass (385 us of instruction issue on the fine level of C4) to learn the row pointers
// first; the move is 150 us of streaming.
__global__ __launch_bounds__(256, 5) void prolong_rows_xw(int n, int n_wtiles, const int *__restrict__ rowptr,
                                                       const int *__restrict__ cols, const double *__restrict__ vals,
                                                       const double *__restrict__ dinv, const int *__restrict__ scol,
                                                       const unsigned char *__restrict__ lidx,
                                                       const unsigned char *__restrict__ spos,
                                                       const int4 *__restrict__ xw_desc, const int run, const double omega,
                                                       const int *__restrict__ agg, int *__restrict__ row_len,
                                                       int *__restrict__ st_cols, double *__restrict__ st_vals) {
    __shared__ unsigned char Ls[4][kPxChunk], Ss[4][kPxChunk];     // window position / strength position of an entry
    __shared__ double Vs[4][kPxChunk];
    __shared__ int As[4][3 * 88];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long W = (long long)gridDim.x * 4, gw = (long long)blockIdx.x * 4 + w;
    for (long long wt = gw; wt < n_wtiles; wt += W) {
        const int row0 = (int)wt * 64;
        const int row1 = min(row0 + 64, n);
        const int r = row0 + lane;
        int rs = 0, re = 0;
        double di = 1.0;
        if (r < row1) {
            rs = rowptr[r];
            re = rowptr[r + 1];
            di = dinv[r];
        }
        const int4 d = xw_desc[wt];
        const int k0 = __shfl(rs, 0, 64);
        const int k1 = __shfl(re, row1 - row0 - 1, 64);
        const bool too_long = re - rs > kPxSlots - 1;
        if (__any(too_long)) {                                // wave-uniform; the scan of the lengths reports the -1 to the host
            if (r < row1) row_len[r] = -1;
            continue;
        }
        const bool windowed = d.w != 0 && k1 - k0 <= kPxChunk;
        if (windowed) {
            const int st[3] = {d.x, d.y, d.z};
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int a = st[q] + lane, b = st[q] + 64 + lane;
                As[w][q * run + lane] = a < n ? agg[a] : 0;
                if (lane < run - 64) As[w][q * run + 64 + lane] = b < n ? agg[b] : 0;
            }
            for (int e = k0 + lane; e < k1; e += 64) {
                Ls[w][e - k0] = lidx[e];
                Ss[w][e - k0] = spos[e];
                Vs[w][e - k0] = vals[e];
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const int self = windowed ? xw_position(d, run, r) : r;
        const int tile_base = k0 + row0;
        // nearly every row of a mesh operator has at most 9 entries: the short instantiation does a third of the compares
        if (__all(re - rs <= 9)) {
            int c[10];
            double sum[10];
            bool keep[10];
            const int o = prolong_row_merge<10>(r < row1, w, rs, re, k0, self, windowed, di, omega, r, Ls, Ss, Vs, As, cols, scol, vals, agg, c, sum, keep);
            const int before = scan_incl_lanes<64>(o) - o;
            if (r < row1) row_len[r] = o;
            prolong_row_store<10>(c, sum, keep, tile_base + before, st_cols, st_vals);
        } else {
            int c[kPxSlots];
            double sum[kPxSlots];
            bool keep[kPxSlots];
            const int o = prolong_row_merge<kPxSlots>(r < row1, w, rs, re, k0, self, windowed, di, omega, r, Ls, Ss, Vs, As, cols, scol, vals, agg, c, sum, keep);
            const int before = scan_incl_lanes<64>(o) - o;
            if (r < row1) row_len[r] = o;
            prolong_row_store<kPxSlots>(c, sum, keep, tile_base + before, st_cols, st_vals);
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
}

// the staged runs of the tiles to their places in the CSR arrays (a wave per tile; source and destination contiguous)
__global__ __launch_bounds__(256) void prolong_unstage(int n, int n_wtiles, const int *__restrict__ a_rowptr,
                                                       const int *__restrict__ rowptr, const int *__restrict__ st_cols,
                                                       const double *__restrict__ st_vals, int *__restrict__ cols,
                                                       double *__restrict__ vals) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const XcdSweep sw = xcd_sweep(n_wtiles, 4, w);
    for (long long wt = sw.t0; wt < sw.t1; wt += sw.stride) {
        const int row0 = (int)wt * 64;
        const int row1 = min(row0 + 64, n);
        const int src = a_rowptr[row0] + row0;
        const int d0 = rowptr[row0], d1 = rowptr[row1];
        for (int k = lane; k < d1 - d0; k += 64) {
            cols[d0 + k] = st_cols[src + k];
            vals[d0 + k] = st_vals[src + k];
        }
    }
}

// fix-up of rows the LDS kernel gave up on: merge their slots (filled by prolong_fill) the slow way
__global__ void prolong_redo_flag(int n, const int *__restrict__ row_len, int *__restrict__ any) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && row_len[i] < 0) atomicExch(any, 1);
}

static int build_prolongator(padne_ctx *ctx, const padne_csr *A, const int *agg, int n_agg, double omega,
                             padne_csr **P, const unsigned char *spos = nullptr, const int *scol = nullptr) {
    hipStream_t s = ctx->stream;
    const int n = (int)A->n_rows;
    Scratch sc(ctx);
    // (the staging places are 32-bit: first entry of the tile + first row of the tile)
    if (spos != nullptr && scol != nullptr && A->xw_state == 1 && A->xw_run <= 85 && n > 0 &&
        (long long)A->nnz + (long long)n < 2147483647LL) {
        // windowed fine level: the rows staged tile by tile, the scan of their lengths, the tiles' runs moved into the CSR arrays
        int *row_len = nullptr, *rowptr_tmp = nullptr, *st_cols = nullptr;
        double *st_vals = nullptr;
        PADNE_TRY(sc.alloc(&row_len, (size_t)n + 1));
        PADNE_TRY(sc.alloc(&rowptr_tmp, (size_t)n + 1));
        PADNE_TRY(sc.alloc(&st_cols, (size_t)A->nnz + (size_t)n));
        PADNE_TRY(sc.alloc(&st_vals, (size_t)A->nnz + (size_t)n));
        const int n_wt = (n + 63) / 64;
        const dim3 g((unsigned)std::min((n_wt + 3) / 4, 8192)), b(256);
        hipLaunchKernelGGL(prolong_rows_xw, g, b, 0, s, n, n_wt, A->rowptr, A->cols, A->vals, A->dinv, scol,
                           (const unsigned char *)A->xw_lidx, spos, A->xw_desc, A->xw_run, omega, agg, row_len, st_cols, st_vals);
        PADNE_HIP_CHECK(hipGetLastError());
        int64_t nnz = 0;
        bool h_gave_up = false;          // a row that gave up left the length -1
        PADNE_TRY(exclusive_scan_i32_flagged(ctx, row_len, rowptr_tmp, n, &nnz, &h_gave_up));
        if (!h_gave_up) {
            padne_csr *m = nullptr;
            PADNE_TRY(csr_alloc(ctx, n, n_agg, nnz, &m));
            PADNE_HIP_CHECK(hipMemcpyAsync(m->rowptr, rowptr_tmp, sizeof(int32_t) * (size_t)(n + 1), hipMemcpyDeviceToDevice, s));
            hipLaunchKernelGGL(prolong_unstage, dim3((unsigned)std::min((n_wt + 3) / 4, 2048)), b, 0, s, n, n_wt, A->rowptr,
                               (const int *)rowptr_tmp, (const int *)st_cols, (const double *)st_vals, m->cols, m->vals);
            PADNE_HIP_CHECK(hipGetLastError());
            *P = m;
            return PADNE_OK;
        }
        // a tile without a plan, or a row with more than kPxCap aggregates: the general path below
    }
    const size_t n_slots = (size_t)A->nnz + (size_t)n;
    int *slot_ptr = nullptr, *row_len = nullptr, *any = nullptr;
    long long *key = nullptr;
    double *val = nullptr;
    PADNE_TRY(sc.alloc(&slot_ptr, (size_t)n + 1));
    PADNE_TRY(sc.alloc(&row_len, (size_t)n + 1));
    PADNE_TRY(sc.alloc(&any, 1));
    PADNE_TRY(sc.alloc(&key, n_slots));
    PADNE_TRY(sc.alloc(&val, n_slots));
    PADNE_HIP_CHECK(hipMemsetAsync(any, 0, sizeof(int), s));
    hipLaunchKernelGGL(prolong_slot_ptr, dim3(nblk((long long)n + 1)), dim3(256), 0, s, n, A->rowptr, slot_ptr);
    hipLaunchKernelGGL(prolong_rows_lds<24>, dim3(nblk(n, 128)), dim3(128), 0, s, n, A->rowptr, A->cols, A->vals, A->dinv,
                       kTheta * kTheta, omega, agg, slot_ptr, key, val, row_len);
    hipLaunchKernelGGL(prolong_redo_flag, dim3(nblk(n)), dim3(256), 0, s, n, row_len, any);
    PADNE_HIP_CHECK(hipGetLastError());
    int h_any = 0;
    PADNE_TRY(read_back(ctx, any, sizeof(int), &h_any));
    if (h_any) {
        // rare (rows touching more than 24 aggregates): redo everything through the slot + merge path
        hipLaunchKernelGGL(prolong_fill, dim3(nblk((long long)n + 1)), dim3(256), 0, s, n, A->rowptr, A->cols, A->vals,
                           A->dinv, kTheta * kTheta, omega, agg, slot_ptr, key, val);
        PADNE_HIP_CHECK(hipGetLastError());
        PADNE_TRY(merge_slots_generic(ctx, n, slot_ptr, key, val, row_len));
    }
    return csr_from_slots(ctx, n, n_agg, slot_ptr, key, val, row_len, P);
}

// No host synchronisation (the number of entries is known): may be queued on the context's second stream.
static int transpose(padne_ctx *ctx, const padne_csr *M, padne_csr **T) {
    // count the entries per column -- and, per column, the waves of 64 rows that hold it (transpose_count_pairs) --, scan
    // the counts into the row pointer of the transpose, place every entry in its row in the order of the rows
    // (transpose_fill_ordered): straight into the CSR arrays of the result, in column order, nothing of it needs the
    // host.  The sort behind it touches only the rows that went through a cursor (see above).
    hipStream_t s = ctx->stream;
    Scratch sc(ctx);
    const long long nc = M->n_cols;
    PADNE_REQUIRE(M->n_rows < 2147483647LL && nc < 2147483647LL, "transpose of a matrix beyond 32-bit indices");
    int *zeroed = nullptr, *long_list = nullptr, *pairs = nullptr;
    const size_t n_zero = 3 * (size_t)nc + 4;
    PADNE_TRY(sc.alloc(&zeroed, n_zero));                  // [counts nc + 1 | cursors nc | waves per column nc | n_long]
    PADNE_TRY(sc.alloc(&long_list, (size_t)nc + 1));
    PADNE_TRY(sc.alloc(&pairs, (size_t)nc * kTpK + 4));
    int *cnt = zeroed, *cursor = zeroed + nc + 1, *npairs = cursor + nc, *n_long = npairs + nc;
    padne_csr *m = nullptr;
    PADNE_TRY(csr_alloc(ctx, nc, M->n_rows, M->nnz, &m));
    const int all_cursors = ctx->opt.force_transpose_cursors ? 1 : 0;
    hipError_t e = hipMemsetAsync(zeroed, 0, sizeof(int) * n_zero, s);
    if (e == hipSuccess && M->n_rows > 0)
        hipLaunchKernelGGL(transpose_count_pairs, dim3(nblk(M->n_rows)), dim3(256), 0, s, (int)M->n_rows, M->rowptr, M->cols, cnt,
                           npairs, pairs, all_cursors);
    int rc = e == hipSuccess ? exclusive_scan_i32_async(ctx, cnt, m->rowptr, nc) : PADNE_E_HIP;
    if (rc == PADNE_OK && e == hipSuccess && M->n_rows > 0) {
        hipLaunchKernelGGL(transpose_fill_ordered, dim3(nblk(M->n_rows)), dim3(256), 0, s, (int)M->n_rows, M->rowptr, M->cols,
                           M->vals, (const int *)m->rowptr, (const int *)npairs, (const int *)pairs, cursor, m->cols, m->vals,
                           all_cursors);
        const long long n_chunks = (nc + kSegRows - 1) / kSegRows;
        unsigned g = (unsigned)std::min<long long>((n_chunks + 3) / 4, 8192);
        if (g >= (unsigned)kNumXcd) g -= g % kNumXcd;
        hipLaunchKernelGGL(sort_csr_rows_seg, dim3(g > 0 ? g : 1), dim3(256), 0, s, (int)nc, (const int *)m->rowptr, m->cols, m->vals,
                           n_long, long_list, (const int *)npairs);
        hipLaunchKernelGGL(sort_listed_csr_rows, dim3(2048), dim3(256), 0, s, (const int *)n_long, (const int *)long_list,
                           (const int *)m->rowptr, m->cols, m->vals);
        e = hipGetLastError();
    }
    if (rc == PADNE_OK && e != hipSuccess) {
        set_error("transpose failed: %s", hipGetErrorString(e));
        rc = PADNE_E_HIP;
    }
    if (rc != PADNE_OK) {
        padne_csr_destroy(m);
        return rc;
    }
    // (the scratch arrays go back to the pool without a synchronisation: reuse is ordered on the context's stream)
    *T = m;
    return PADNE_OK;
}

// rows of a product left in their merge slots (row i = [begin[i], end[i]) of key / val), owned by this object
struct SlotRows {
    padne_ctx *ctx = nullptr;
    long long n_rows = 0, n_cols = 0, n_slots = 0;
    int *begin = nullptr, *end = nullptr;
    long long *key = nullptr;
    double *val = nullptr;
    bool valid = false;
    ~SlotRows() { release(); }
    void release() {
        if (ctx != nullptr) {
            pool_free(ctx, begin);
            pool_free(ctx, end);
            pool_free(ctx, key);
            pool_free(ctx, val);
        }
        begin = end = nullptr;
        key = nullptr;
        val = nullptr;
        valid = false;
    }
};

__global__ void slot_row_ends(int n, const int *__restrict__ slot_ptr, const int *__restrict__ row_len, int *__restrict__ end) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) end[i] = slot_ptr[i] + row_len[i];
}

// C = X * Y.  `y_slots`: Y is given by its uncompacted rows instead of the arrays of the handle (only n_rows / n_cols of
// Y are read then).  `keep_slots`: leave the product in ITS slots (no scan, no compaction) and return no matrix.
// while_counting: called once the count and the scan of the slot offsets are queued and before the host waits for their
// total -- what it launches (on another stream) is launched while they run instead of in front of them
// while_multiplying: called once the row kernels are queued, before the product is compacted (again: launches for the
// other stream made while this one has work, not while it waits for the host)
static int spgemm(padne_ctx *ctx, const padne_csr *X, const padne_csr *Y, padne_csr **C, const SlotRows *y_slots = nullptr,
                  SlotRows *keep_slots = nullptr, const std::function<int()> *while_counting = nullptr,
                  const std::function<int()> *while_multiplying = nullptr) {
    const int *y_begin = y_slots != nullptr ? y_slots->begin : Y->rowptr;
    const int *y_end = y_slots != nullptr ? y_slots->end : Y->rowptr + 1;
    const int *y_cols = y_slots != nullptr ? (const int *)y_slots->key + 1 : Y->cols;      // upper half of a little-endian key
    const double *y_vals = y_slots != nullptr ? y_slots->val : Y->vals;
    const int y_cs = y_slots != nullptr ? 2 : 1;
    hipStream_t s = ctx->stream;
    Scratch sc(ctx);
    const int n = (int)X->n_rows;
    int *cnt = nullptr, *slot_ptr = nullptr, *row_len = nullptr;
    PADNE_TRY(sc.alloc(&cnt, (size_t)n + 1));
    PADNE_TRY(sc.alloc(&slot_ptr, (size_t)n + 1));
    PADNE_TRY(sc.alloc(&row_len, (size_t)n + 1));
    // (no zeroing: spgemm_count stores every one of the n entries the scan reads)
    if (n > 0) {
        const double x_avg = (double)X->nnz / (double)n;       // (0 for the row views of the split below: thread per row)
        if (x_avg > 48.0)
            hipLaunchKernelGGL((spgemm_count_lanes<32>), dim3(nblk((long long)n * 32)), dim3(256), 0, s, n, X->rowptr, X->cols, y_begin, y_end, cnt);
        else if (x_avg > 16.0)
            hipLaunchKernelGGL((spgemm_count_lanes<8>), dim3(nblk((long long)n * 8)), dim3(256), 0, s, n, X->rowptr, X->cols, y_begin, y_end, cnt);
        else
            hipLaunchKernelGGL(spgemm_count, dim3(nblk(n)), dim3(256), 0, s, n, X->rowptr, X->cols, y_begin, y_end, cnt);
    }
    PADNE_HIP_CHECK(hipGetLastError());
    int64_t n_slots = 0;
    int rc_scan = PADNE_OK;
    {
        ScanTicket ticket;
        long long h[2] = {0, 0};
        PADNE_TRY(scan_i32_begin(ctx, cnt, slot_ptr, n, &ticket, true));
        const int rc_cb = while_counting != nullptr ? (*while_counting)() : PADNE_OK;
        rc_scan = scan_i32_end(ctx, &ticket, h);
        PADNE_TRY(rc_cb);
        if (rc_scan == PADNE_OK) {
            if (h[0] < 0 || h[1] != 0) {
                set_error("scan of negative counts");
                rc_scan = PADNE_E_INVALID;
            } else if (h[0] >= 2147483647LL) {
                set_error("%lld entries exceed the 32-bit index space", h[0]);
                rc_scan = PADNE_E_TOOLARGE;
            }
            n_slots = h[0];
        }
    }
    // (PADNE_FORCE=spgemm_split:<slots> lowers the limit so that tests reach the split path on small systems)
    const long long split_limit = ctx->opt.force_spgemm_split;
    if ((rc_scan == PADNE_E_TOOLARGE || (rc_scan == PADNE_OK && split_limit > 0 && n_slots > split_limit)) && n >= 2) {
        // more product slots than 32-bit offsets address (A*P of a 130 M-row mesh Laplacian: 17 per row) although the
        // product itself fits: multiply the two halves of the rows separately and stack the results.  A half is a
        // view of X -- row pointers are absolute positions in cols / vals.
        sc.release();
        const int h = n / 2;
        padne_csr top = *X, bottom = *X;
        top.n_rows = h;
        bottom.n_rows = n - h;
        bottom.rowptr = X->rowptr + h;
        top.nnz = bottom.nnz = 0;      // not used by the product kernels
        top.amg = bottom.amg = nullptr;
        if (amg_verbose()) fprintf(stderr, "[amg]   spgemm %lld rows: %lld product slots, split in two\n", (long long)n, (long long)n_slots);
        padne_csr *Ct = nullptr, *Cb = nullptr;
        int rc = spgemm(ctx, &top, Y, &Ct, y_slots);
        if (rc == PADNE_OK) rc = spgemm(ctx, &bottom, Y, &Cb, y_slots);
        if (rc == PADNE_OK) rc = csr_vstack(ctx, Ct, Cb, Y->n_cols, C);
        if (Ct) padne_csr_destroy(Ct);
        if (Cb) padne_csr_destroy(Cb);
        return rc;
    }
    PADNE_TRY(rc_scan);
    if (amg_verbose()) fprintf(stderr, "[amg]   spgemm %lld rows, %lld product slots\n", (long long)n, (long long)n_slots);
    long long *key = nullptr;
    double *val = nullptr;
    int *row_begin = nullptr;      // where the rows start when that is not slot_ptr (spgemm_rows_lds_pipe)
    PADNE_TRY(sc.alloc(&key, (size_t)n_slots + 4));      // (+4: kernels that read a row four slots at a time, as the Y of a later product)
    PADNE_TRY(sc.alloc(&val, (size_t)n_slots + 4));
    if (n > 0) {
        const double avg = (double)n_slots / (double)n;
        const size_t dense_lds = (size_t)Y->n_cols * 9 + 16;
        if (avg > 256.0 && dense_lds <= 150 * 1024) {
            // long rows over a small column space: dense LDS accumulator, one workgroup per row
            (void)hipFuncSetAttribute((const void *)spgemm_rows_dense, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)dense_lds);
            hipLaunchKernelGGL(spgemm_rows_dense, dim3(n), dim3(256), dense_lds, s, (int)Y->n_cols, X->rowptr, X->cols,
                               X->vals, y_begin, y_cols, y_vals, y_end, y_cs, slot_ptr, key, val, row_len, 0);
        } else if (avg <= 24.0) {
            // A*P on the fine levels: a dozen products per row -> one thread per row with small sorted lists in LDS (12 entries:
            // 18 KiB per workgroup, four waves per SIMD -- with 16 the LDS allowed three: 2.47 -> 2.16 ms on the fine level of config C4;
            // the rare longer row is redone in global memory)
            {
                // the rows of a 64-row tile back to back
                PADNE_TRY(sc.alloc(&row_begin, (size_t)n + 1));
                if (y_cs == 1)
                    hipLaunchKernelGGL((spgemm_rows_lds_pipe<12, 9, 4, 1>), dim3(nblk(n, 128)), dim3(128), 0, s, n, X->rowptr, X->cols,
                                       X->vals, y_begin, y_cols, y_vals, y_end, slot_ptr, key, val, row_len, row_begin);
                else
                    hipLaunchKernelGGL((spgemm_rows_lds_pipe<12, 9, 4, 2>), dim3(nblk(n, 128)), dim3(128), 0, s, n, X->rowptr, X->cols,
                                       X->vals, y_begin, y_cols, y_vals, y_end, slot_ptr, key, val, row_len, row_begin);
            }
            {
                // the rows whose list overflowed (next to a via ring: hundreds of products): collected, a wave each; one thread per
                // row in global memory (spgemm_rows_redo: 98 us on the fine level of C4 for a few hundred rows) only for what is left
                const int *place = row_begin != nullptr ? row_begin : slot_ptr;
                int *pend = nullptr, *pend_count = nullptr;
                PADNE_TRY(sc.alloc(&pend, (size_t)n));
                PADNE_TRY(sc.alloc(&pend_count, 1));
                PADNE_HIP_CHECK(hipMemsetAsync(pend_count, 0, sizeof(int), s));
                hipLaunchKernelGGL(collect_pending_rows, dim3(nblk(n)), dim3(256), 0, s, n, (const int *)row_len, pend, pend_count);
                hipLaunchKernelGGL((spgemm_rows_lanes<1024, 512, 64, true>), dim3(256), dim3(256), 0, s, n, X->rowptr, X->cols, X->vals, y_begin,
                                   y_cols, y_vals, y_end, y_cs, place, key, val, row_len, (const int *)pend, (const int *)pend_count);
                hipLaunchKernelGGL(spgemm_rows_redo, dim3(nblk(n, 128)), dim3(128), 0, s, n, X->rowptr, X->cols, X->vals,
                                   y_begin, y_cols, y_vals, y_end, y_cs, place, key, val, row_len);
            }
        } else if (avg <= 256.0) {
            // tens to hundreds of products per row: one wave per row, the rows that do not fit are redone in global memory
            const unsigned gw = (unsigned)std::min<long long>(((long long)n + 3) / 4, 16384);
            const unsigned gl = std::min(gw, 4096u);       // passes over the list of rows the first pass left
            int *pend = nullptr, *pend_count = nullptr;
            PADNE_TRY(sc.alloc(&pend, (size_t)n));
            PADNE_TRY(sc.alloc(&pend_count, 1));
            PADNE_HIP_CHECK(hipMemsetAsync(pend_count, 0, sizeof(int), s));
            if (avg <= 110.0) {
                // short rows: two rows per wave with small limits first (1.5 KiB of LDS per row), or four where the rows
                // of X are short as well (A P below the finest level: 13 entries, 30 products); then one row per wave for
                // the rows that did not fit
                const double x_avg = (double)X->nnz / (double)n;
                if (x_avg > 0.0 && x_avg <= 14.0 && avg <= 48.0) {
                    unsigned gs = (unsigned)std::min<long long>(((long long)n + 15) / 16, 16384);
                    hipLaunchKernelGGL((spgemm_rows_lanes<128, 32, 16, false>), dim3(gs), dim3(256), 0, s, n, X->rowptr, X->cols, X->vals,
                                       y_begin, y_cols, y_vals, y_end, y_cs, slot_ptr, key, val, row_len);
                } else {
                    unsigned gs = (unsigned)std::min<long long>(((long long)n + 7) / 8, 16384);
                    hipLaunchKernelGGL((spgemm_rows_lanes<256, 64, 32, false>), dim3(gs), dim3(256), 0, s, n, X->rowptr, X->cols, X->vals,
                                       y_begin, y_cols, y_vals, y_end, y_cs, slot_ptr, key, val, row_len);
                }
                hipLaunchKernelGGL(collect_pending_rows, dim3(nblk(n)), dim3(256), 0, s, n, (const int *)row_len, pend, pend_count);
                hipLaunchKernelGGL((spgemm_rows_lanes<256, 128, 64, true>), dim3(gl), dim3(256), 0, s, n, X->rowptr, X->cols, X->vals,
                                   y_begin, y_cols, y_vals, y_end, y_cs, slot_ptr, key, val, row_len, (const int *)pend,
                                   (const int *)pend_count);
            } else {
                // coarse levels: rows of a few hundred products, some of a thousand
                hipLaunchKernelGGL((spgemm_rows_lanes<512, 256, 64, false>), dim3(gw), dim3(256), 0, s, n, X->rowptr, X->cols, X->vals,
                                   y_begin, y_cols, y_vals, y_end, y_cs, slot_ptr, key, val, row_len);
                hipLaunchKernelGGL(collect_pending_rows, dim3(nblk(n)), dim3(256), 0, s, n, (const int *)row_len, pend, pend_count);
            }
            // a second wave pass with doubled limits keeps the few long rows (aggregates next to a via hub) away from the
            // serial fallback (0.5 ms for a handful of rows)
            hipLaunchKernelGGL((spgemm_rows_lanes<1024, 512, 64, true>), dim3(std::min(gl, 1024u)), dim3(256), 0, s, n, X->rowptr, X->cols,
                               X->vals, y_begin, y_cols, y_vals, y_end, y_cs, slot_ptr, key, val, row_len, (const int *)pend,
                               (const int *)pend_count);
            if (dense_lds <= 150 * 1024) {
                // rows beyond the wave kernels' limits over a small column space: the dense accumulator, a workgroup each
                (void)hipFuncSetAttribute((const void *)spgemm_rows_dense, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)dense_lds);
                hipLaunchKernelGGL(spgemm_rows_dense, dim3(std::min(n, 1024)), dim3(256), dense_lds, s, (int)Y->n_cols, X->rowptr, X->cols,
                                   X->vals, y_begin, y_cols, y_vals, y_end, y_cs, slot_ptr, key, val, row_len, 1, (const int *)pend,
                                   (const int *)pend_count);
            }
            hipLaunchKernelGGL(spgemm_rows_redo, dim3(nblk(n, 128)), dim3(128), 0, s, n, X->rowptr, X->cols, X->vals,
                               y_begin, y_cols, y_vals, y_end, y_cs, slot_ptr, key, val, row_len);
        } else {
            hipLaunchKernelGGL(spgemm_rows, dim3(nblk(n, 128)), dim3(128), 0, s, n, X->rowptr, X->cols, X->vals,
                               y_begin, y_cols, y_vals, y_end, y_cs, slot_ptr, key, val, row_len);
        }
    }
    PADNE_HIP_CHECK(hipGetLastError());
    // (with a compaction behind the row kernels the call is made once its scan is queued as well: on a small level the row
    // kernels are over before the host has made forty launches for the other stream)
    if (while_multiplying != nullptr && keep_slots != nullptr) PADNE_TRY((*while_multiplying)());
    if (keep_slots != nullptr) {
        keep_slots->release();
        int *end = (int *)pool_alloc(ctx, sizeof(int) * (size_t)(n > 0 ? n : 1));
        if (end == nullptr) return PADNE_E_NOMEM;
        int *first = row_begin != nullptr ? row_begin : slot_ptr;
        if (n > 0) hipLaunchKernelGGL(slot_row_ends, dim3(nblk(n)), dim3(256), 0, s, n, (const int *)first, row_len, end);
        PADNE_HIP_CHECK(hipGetLastError());
        keep_slots->ctx = ctx;
        keep_slots->n_rows = n;
        keep_slots->n_cols = Y->n_cols;
        keep_slots->n_slots = n_slots;
        keep_slots->begin = first;
        keep_slots->end = end;
        keep_slots->key = key;
        keep_slots->val = val;
        keep_slots->valid = true;
        sc.disown(first);
        sc.disown(key);
        sc.disown(val);
        if (C != nullptr) *C = nullptr;
        return PADNE_OK;
    }
    auto trampoline = [](void *f) -> int { return (*static_cast<const std::function<int()> *>(f))(); };
    return csr_from_slots(ctx, n, Y->n_cols, row_begin != nullptr ? row_begin : slot_ptr, key, val, row_len, C,
                          while_multiplying != nullptr ? +trampoline : nullptr, const_cast<std::function<int()> *>(while_multiplying));
}

// ---- W = P - c D^-1 (A P): coarse correction and post-smoothing of a level in ONE product ----------------------------
// The up-leg of the V(1,1) cycle is  x2 = x1 + P e ;  x3 = x2 + c D^-1 (b - A x2).  With r1 = b - A x1 (the residual the
// down-leg has just restricted, still in memory)
//     x3 = x1 + c D^-1 r1 + (P - c D^-1 A P) e :
// one product with W (the pattern of A P: 52 M entries on the fine level of C4) instead of one with P (24 M) and one
// with A (70 M).  The setup forms A P anyway; W is built from its merge slots while they are still around, in single
// precision only (it exists for the float cycle).  Algebraically the same operator, so still symmetric.
__global__ void slot_row_lengths(int n, const int *__restrict__ begin, const int *__restrict__ end, int *__restrict__ len) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) len[i] = end[i] - begin[i];
}

// destination-ordered like compact_rows (assemble.hip): a wave owns 64 rows, its lanes walk the OUTPUT entries of those
// rows (coalesced writes), find the row of an entry by bisection in the staged row pointers, read its slot and look the
// column up in the (two or three entries of the) row of P
__global__ __launch_bounds__(256) void w_from_slots_kernel(int n, const int *__restrict__ ap_begin,
                                                           const long long *__restrict__ ap_key,
                                                           const double *__restrict__ ap_val, const int *__restrict__ pr,
                                                           const int *__restrict__ pc, const double *__restrict__ pv,
                                                           const double *__restrict__ dinv, const double c,
                                                           const int *__restrict__ wr, int *__restrict__ wc,
                                                           float *__restrict__ wv) {
    __shared__ int rp_all[4][65];
    __shared__ int sp_all[4][65];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int *rp = rp_all[w], *sp = sp_all[w];
    const long long n_wt = ((long long)n + 63) / 64;
    for (long long wt = (long long)blockIdx.x * 4 + w; wt < n_wt; wt += (long long)gridDim.x * 4) {
        const long long r0 = wt * 64;
        const int nr = (int)((n - r0) < 64 ? (n - r0) : 64);
        if (lane <= nr) rp[lane] = wr[r0 + lane];
        if (lane < nr) sp[lane] = ap_begin[r0 + lane];
        if (lane == 0 && nr == 64) rp[64] = wr[r0 + 64];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const int d0 = rp[0], d1 = rp[nr];
        for (int k = d0 + lane; k < d1; k += 64) {
            int lo = 0, hi = nr;                     // largest row with rp[row] <= k
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (rp[mid] <= k) lo = mid; else hi = mid;
            }
            const int i = (int)r0 + lo;
            const int src = sp[lo] + (k - rp[lo]);
            const int col = (int)(ap_key[src] >> 32);
            double v = -c * dinv[i] * ap_val[src];
            const I2u be = *reinterpret_cast<const I2u *>(pr + i);
            if (be.y - be.x <= 4) {
                // the row of P in three loads instead of a dependent pair per entry (behind its end: the next row or the padding)
                const I4u c4 = *reinterpret_cast<const I4u *>(pc + be.x);
                const D2u p0 = *reinterpret_cast<const D2u *>(pv + be.x), p1 = *reinterpret_cast<const D2u *>(pv + be.x + 2);
                const int ln = be.y - be.x;
                if (ln > 0 && c4.x == col) v = p0.x + v;      // (the columns of a row are distinct: at most one of these)
                if (ln > 1 && c4.y == col) v = p0.y + v;
                if (ln > 2 && c4.z == col) v = p1.x + v;
                if (ln > 3 && c4.w == col) v = p1.y + v;
            } else {
                for (int q = be.x; q < be.y; ++q)
                    if (pc[q] == col) v = pv[q] + v;
            }
            wc[k] = col;
            wv[k] = (float)v;
        }
        // the SpMV streams four non-zeros per lane and load: the three entries behind the last one must be valid columns
        // (csr_alloc zeroes the padding of ordinary matrices; W's arrays are sized by the slot count, its end is known here)
        if (r0 + nr == n && lane < 4) {
            wc[d1 + lane] = 0;
            wv[d1 + lane] = 0.f;
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
}

// No host synchronisation (the arrays are sized by the slot count, an upper bound of the entries): queued on the
// context's second stream next to the next level's setup; the slots must stay alive until that stream has been joined.
// grid_cap > 0: the two large kernels use that many workgroups at most -- the W of the inner levels are built while the
// other stream inverts the coarsest operator, 51 dependent launches of seven workgroups each, and a launch that finds every
// CU full of this build's waves waits for them (149 instead of 36 us behind the full grid of level 1)
static int build_w_operator(padne_ctx *ctx, const padne_csr *A, const padne_csr *P, const SlotRows &ap, long long n_slots,
                            double c, padne_csr **W_out, int grid_cap = 0, bool with_plan = true) {
    hipStream_t s = ctx->stream;
    const int n = (int)A->n_rows;
    Scratch sc(ctx);
    int *len = nullptr;
    PADNE_TRY(sc.alloc(&len, (size_t)n + 1));
    struct CsrDrop { void operator()(padne_csr *m) const { if (m) padne_csr_destroy(m); } };
    std::unique_ptr<padne_csr, CsrDrop> w_owner(new padne_csr());      // (every way out but the last destroys it)
    padne_csr *W = w_owner.get();
    W->n_rows = n;
    W->n_cols = P->n_cols;
    W->nnz = n_slots;                      // capacity; the row pointers hold the truth
    W->device = ctx->device;
    W->owner = ctx;
    W->hierarchy_operator = true;
    W->xw_state = -1;                      // five bands of aggregates: no x-window plan (measured on P)
    W->rowptr = (int32_t *)pool_alloc(ctx, sizeof(int32_t) * ((size_t)n + 1));
    W->cols = (int32_t *)pool_alloc(ctx, sizeof(int32_t) * ((size_t)n_slots + kPadNnz));
    W->vals32 = (float *)pool_alloc(ctx, sizeof(float) * ((size_t)n_slots + kPadNnz));
    if (!W->rowptr || !W->cols || !W->vals32) return PADNE_E_NOMEM;
    hipLaunchKernelGGL(slot_row_lengths, dim3(nblk(n)), dim3(256), 0, s, n, ap.begin, ap.end, len);
    PADNE_HIP_CHECK(hipGetLastError());
    PADNE_TRY(exclusive_scan_i32_async(ctx, len, W->rowptr, n));
    unsigned gw = nblk(((long long)n + 63) / 64, 4);
    if (grid_cap > 0 && gw > (unsigned)grid_cap) gw = (unsigned)grid_cap;
    hipLaunchKernelGGL(w_from_slots_kernel, dim3(gw), dim3(256), 0, s, n, ap.begin, ap.key, ap.val,
                       P->rowptr, P->cols, P->vals, A->dinv, c, (const int *)W->rowptr, W->cols, W->vals32);
    PADNE_HIP_CHECK(hipGetLastError());
    // its x-window plan: twelve short runs per tile (spmv.hip, csr_build_xw_plan_wide), on this stream, no look at the host
    // (the fine level only: below it the aggregates are numbered in the order of their roots, 64 consecutive rows reach
    // into a dozen far-apart stretches of columns and 1 % of the tiles qualified -- C4, level 1: 144 of 21 486)
    W->xw_state = 0;
    if (with_plan) PADNE_TRY(csr_build_xw_plan_wide(ctx, W, grid_cap));
    if (W->xw_state != 1) W->xw_state = -1;
    *W_out = w_owner.release();
    return PADNE_OK;
}

static int dense_inverse(padne_ctx *ctx, const padne_csr *A, double **inv_out) {
    hipStream_t s = ctx->stream;
    const int n = (int)A->n_rows;
    Scratch sc(ctx);
    double *W = nullptr, *W2 = nullptr, *inv = nullptr;
    PADNE_TRY(sc.alloc(&W, (size_t)n * n));
    PADNE_TRY(sc.alloc(&W2, (size_t)n * n));
    inv = (double *)pool_alloc(ctx, sizeof(double) * (size_t)(n > 0 ? n : 1) * (size_t)(n > 0 ? n : 1));
    if (inv == nullptr) return PADNE_E_NOMEM;
    if (n > 0) {
        hipLaunchKernelGGL(dense_from_csr, dim3(n), dim3(256), 0, s, n, A->rowptr, A->cols, A->vals, W);
        const dim3 ge(nblk(n), (unsigned)((n + kGjStrip - 1) / kGjStrip));
        double *src = W, *dst = W2;
        // 64 pivots per launch on the matrix cores (gj64_step); PADNE_GJ_VECTOR=1, or a matrix of at most 64 unknowns: the
        // vector kernel, 16 pivots per launch -- the form the matrix-core inverse is tested against
        const bool mfma = !ctx->opt.gj_vector && n > kGjM;
        if (mfma) {
            double *side = nullptr;
            PADNE_TRY(sc.alloc(&side, (size_t)2 * kGjM * kGjM));
            const int n_ct = (n + kGjTC - 1) / kGjTC, n_rt = (n + kGjTR - 1) / kGjTR, n_tiles = n_ct * n_rt;
            const unsigned g64 = 1u + (unsigned)(n_ct * ((n_rt + 3) / 4));      // (a workgroup: four row tiles of one column block)
            hipLaunchKernelGGL(gj64_prepare, dim3(1), dim3(256), 0, s, n, 0, std::min(kGjM, n), (const double *)src, side);
            int par = 0;
            for (int k = 0; k < n; k += kGjM) {
                const int bs = std::min(kGjM, n - k);
                const int next_bs = k + kGjM < n ? std::min(kGjM, n - k - kGjM) : 0;
                double *to = k + kGjM >= n ? inv : dst;
                hipLaunchKernelGGL(gj64_step, dim3(g64), dim3(256), 0, s, n, k, bs, (const double *)src, to,
                                   (const double *)(side + (size_t)par * kGjM * kGjM), side + (size_t)(par ^ 1) * kGjM * kGjM,
                                   next_bs, n_ct, n_tiles);
                par ^= 1;
                std::swap(src, dst);
            }
        } else {
            for (int k = 0; k < n; k += kGjBlock) {
                // the last step writes the finished inverse where it stays
                if (n - k >= kGjBlock)
                    hipLaunchKernelGGL(gj_block_step<true>, ge, dim3(256), 0, s, n, k, src, k + kGjBlock >= n ? inv : dst);
                else
                    hipLaunchKernelGGL(gj_block_step<false>, ge, dim3(256), 0, s, n, k, src, inv);
                std::swap(src, dst);
            }
        }
    }
    const hipError_t e = hipGetLastError();      // (no synchronisation: the scratch goes back to the stream-ordered pool)
    if (e != hipSuccess) {
        pool_free(ctx, inv);
        set_error("coarse inverse failed: %s", hipGetErrorString(e));
        return PADNE_E_HIP;
    }
    *inv_out = inv;
    return PADNE_OK;
}

void amg_destroy(void *p) {
    Amg *amg = (Amg *)p;
    if (!amg) return;
    (void)hipSetDevice(amg->device);
    for (AmgLevel &L : amg->levels) {
        if (L.A_owned) padne_csr_destroy(L.A_owned);
        if (L.P) padne_csr_destroy(L.P);
        if (L.P_halo) padne_csr_destroy(L.P_halo);
        pool_free(amg->ctx, L.e_ext);
        if (L.R) padne_csr_destroy(L.R);
        if (L.W) padne_csr_destroy(L.W);
        pool_free(amg->ctx, L.b);
        pool_free(amg->ctx, L.xa);
        pool_free(amg->ctx, L.xb);
        pool_free(amg->ctx, L.tmp);
        pool_free(amg->ctx, L.b8);
        pool_free(amg->ctx, L.xa8);
        pool_free(amg->ctx, L.xb8);
        pool_free(amg->ctx, L.tmp8);
        pool_free(amg->ctx, L.export_owned);
    }
    pool_free(amg->ctx, amg->coarse_inv);
    pool_free(amg->ctx, amg->coarse_inv32);
    pool_free(amg->ctx, amg->coarse_gather);
    pool_free(amg->ctx, amg->seg_off);
    pool_free(amg->ctx, amg->tail_r);
    pool_free(amg->ctx, amg->tail_z);
    pool_free(amg->ctx, amg->tail_slot_idx);
    if (amg->tail) padne_csr_destroy(amg->tail);
    delete amg;
}

static int alloc_vec(padne_ctx *ctx, double **p, long long n) {
    *p = (double *)pool_alloc(ctx, sizeof(double) * (size_t)(n > 0 ? n : 1));
    return *p != nullptr ? PADNE_OK : PADNE_E_NOMEM;
}

struct PhaseTimer {   // wall-clock phase timing, only when PADNE_AMG_VERBOSE is set
    padne_ctx *ctx;
    std::chrono::steady_clock::time_point t0;
    bool on;
    explicit PhaseTimer(padne_ctx *c, bool enabled) : ctx(c), on(enabled) {
        if (on) { (void)hipStreamSynchronize(ctx->stream); t0 = std::chrono::steady_clock::now(); }
    }
    void lap(const char *what) {
        if (!on) return;
        (void)hipStreamSynchronize(ctx->stream);
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[amg]     %-22s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

static thread_local bool t_amg_verbose = false;      // PADNE_VERBOSE=amg of the context whose setup runs on this thread
static bool amg_verbose() { return t_amg_verbose; }

static int amg_setup_dist(padne_ctx *ctx, padne_csr *A0);
static int host_allgather(padne_ctx *ctx, const std::vector<double> &mine, std::vector<double> &all);

// Single-precision cycle (default; PADNE_AMG_F64=1 keeps double): float copies of every level operator.  Skipped
// when 1/diag of the fine matrix leaves [1e-15, 1e15] (the cycle input is normalised, the operator is not).
static int f32_range(padne_ctx *ctx, const padne_csr *A0, double *lo_out, double *hi_out) {
    hipStream_t s = ctx->stream;
    const int g = (int)std::min<long long>((A0->n_rows + 255) / 256, 1024);
    double *mins = ctx->partials + 6 * kMaxPartials, *maxs = ctx->partials + 7 * kMaxPartials;
    hipLaunchKernelGGL(abs_range_kernel, dim3(g), dim3(256), 0, s, (long long)A0->n_rows, (const double *)A0->dinv, mins,
                       maxs);
    PADNE_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(fold_range_kernel, dim3(1), dim3(256), 0, s, (const double *)mins, (const double *)maxs, g, mins + g);
    PADNE_HIP_CHECK(hipGetLastError());
    double h[2] = {0.0, 0.0};
    PADNE_TRY(read_back(ctx, mins + g, sizeof(h), h));
    const double lo = h[0], hi = h[1];
    *lo_out = lo;
    *hi_out = hi;
    return PADNE_OK;
}

// Does the cycle run in single precision?  Decided from the range of 1/diag of the fine matrix -- in a row-partitioned run
// from the range over ALL ranks: the same decision everywhere.
static int decide_f32(padne_ctx *ctx, const padne_csr *A0, bool dist, bool *want) {
    *want = false;
    if (ctx->opt.amg_f64 || A0->hierarchy_operator) return PADNE_OK;   // (hierarchy_operator: the gathered tail of a row-partitioned hierarchy)
    double lo = 0.0, hi = 0.0;
    PADNE_TRY(f32_range(ctx, A0, &lo, &hi));
    if (dist) {
        std::vector<double> mine = {lo, hi}, all;
        PADNE_TRY(host_allgather(ctx, mine, all));
        for (int q = 0; q < ctx->world; ++q) {
            lo = std::min(lo, all[(size_t)q * 2]);
            if (all[(size_t)q * 2 + 1] > hi || !(all[(size_t)q * 2 + 1] == all[(size_t)q * 2 + 1])) hi = all[(size_t)q * 2 + 1];
        }
    }
    *want = lo >= 1e-15 && hi <= 1e15;
    return PADNE_OK;
}

// known: the decision if the caller has taken it already (amg_setup_dist, before it builds the up-leg operators), -1 otherwise
static int enable_f32(padne_ctx *ctx, Amg *amg, int known = -1) {
    if (ctx->opt.amg_f64 || amg->levels.size() < 2) return PADNE_OK;
    if (amg->levels[0].A->hierarchy_operator) return PADNE_OK;   // the gathered tail of a row-partitioned hierarchy
    hipStream_t s = ctx->stream;
    bool want = known == 1;
    if (known < 0) PADNE_TRY(decide_f32(ctx, amg->levels[0].A, amg->dist, &want));
    if (!want) return PADNE_OK;
    for (AmgLevel &L : amg->levels) {
        PADNE_TRY(csr_build_f32(ctx, const_cast<padne_csr *>(L.A)));
        if (L.P) PADNE_TRY(csr_build_f32(ctx, L.P));
        if (L.P_halo) PADNE_TRY(csr_build_f32(ctx, L.P_halo));
        if (L.R) PADNE_TRY(csr_build_f32(ctx, L.R));
    }
    if (amg->levels[0].b == nullptr) PADNE_TRY(alloc_vec(ctx, &amg->levels[0].b, amg->levels[0].n));
    if (!amg->dist && amg->n_coarse > 0) {
        const size_t cnt = (size_t)amg->n_coarse * (size_t)amg->n_coarse;
        amg->coarse_inv32 = (float *)pool_alloc(ctx, sizeof(float) * cnt);
        if (amg->coarse_inv32 == nullptr) return PADNE_E_NOMEM;
        hipLaunchKernelGGL(f32_copy_amg, dim3(nblk((long long)cnt)), dim3(256), 0, s, (long long)cnt, amg->coarse_inv,
                           amg->coarse_inv32);
        PADNE_HIP_CHECK(hipGetLastError());
    }
    amg->f32 = true;
    return PADNE_OK;
}

// Single-GPU setup.  Two chains run side by side on the context's two streams (the kernels of the coarse levels are
// short and latency-bound, so a second queue is nearly free):
//   main:    aggregate -> prolongator P -> A P ..................... -> R (A P) -> next level ... -> dense inverse
//   second:  Lanczos bound of the level, float copy of A | wait P -> R = P^T, float copies of P and R
// The smoother bounds are only read when the cycle is applied, so they are collected at the very end.
int amg_setup(padne_ctx *ctx, padne_csr *A0) {
    if (A0->amg) return PADNE_OK;
    if (ctx->halo_on && A0->n_cols != A0->n_rows) return amg_setup_dist(ctx, A0);
    PADNE_REQUIRE(A0->n_rows == A0->n_cols, "multigrid needs a square matrix");
    PADNE_TRY(csr_build_dinv(ctx, A0));
    PADNE_HIP_CHECK(hipEventRecord(ctx->ev0, ctx->stream));
    t_amg_verbose = ctx->opt.verbose_amg;
    padne_ctx *aux = ctx->opt.setup_one_stream ? nullptr : aux_context(ctx);
    if (aux == nullptr) aux = ctx;
    const bool two = aux != ctx;
    // single-precision cycle?  (decided from 1/diag of the fine matrix, before anything is queued)
    bool want_f32 = !ctx->opt.amg_f64 && !A0->hierarchy_operator;
    if (want_f32) {
        double lo = 0.0, hi = 0.0;
        PADNE_TRY(f32_range(ctx, A0, &lo, &hi));
        want_f32 = lo >= 1e-15 && hi <= 1e15;
    }
    Amg *amg = new Amg();
    amg->device = ctx->device;
    amg->ctx = ctx;
    int rc = PADNE_OK;
    const padne_csr *A = A0;
    double nnz_total = 0.0;
    struct Pending { int level; LanczosJob job; };
    std::vector<Pending *> pending;      // Lanczos estimates in flight on the second stream
    SlotRows ap_keep;                    // slots of the fine level's A P while W is built from them on the second stream
    // slots of A P of the levels below it: their W needs the level's damping, which the Lanczos estimate settles at the
    // end of the setup -- built there, on the second stream, next to the dense inverse of the coarsest operator
    struct KeptSlots { int level; SlotRows rows; };
    std::vector<std::unique_ptr<KeptSlots>> ap_inner;
    const bool w_inner = ctx->opt.amg_w == 2;
    auto drop_pending = [&]() {
        for (Pending *pj : pending) {
            double unused = 0.0;
            (void)lanczos_finish(&pj->job, &unused);
            delete pj;
        }
        pending.clear();
    };
    for (int lvl = 0; lvl < kMaxLevels; ++lvl) {
        AmgLevel L;
        L.A = A;
        L.A_owned = (lvl == 0) ? nullptr : const_cast<padne_csr *>(A);
        L.n = A->n_rows;
        nnz_total += (double)A->nnz;
        const bool coarsest = A->n_rows <= ctx->opt.amg_coarse_n || lvl == kMaxLevels - 1;
        if ((rc = alloc_vec(ctx, &L.xa, L.n)) != PADNE_OK || (rc = alloc_vec(ctx, &L.tmp, L.n)) != PADNE_OK) { amg->levels.push_back(L); break; }
        if (lvl > 0 && ((rc = alloc_vec(ctx, &L.b, L.n)) != PADNE_OK || (rc = alloc_vec(ctx, &L.xb, L.n)) != PADNE_OK)) { amg->levels.push_back(L); break; }
        // second stream, everything that needs only A_l: the Lanczos bound and the float copy.  Neither is read before the
        // cycle runs, so they are queued LATE in the level -- behind the transposes the main stream waits for, and while
        // the main stream is busy with R (A P): the one host thread that launches for both streams then starts the level's
        // aggregation without first spending 30 launches on the other stream (the small levels are launch-bound)
        const bool lanczos = lvl > 0 && A->n_rows > ctx->opt.amg_coarse_n;
        bool w_expected = false;             // the level's fused up-leg operator is on its way: P is not multiplied with in single precision
        auto queue_level_extras = [&](const bool ordered = false) -> int {
            if (two && !ordered) PADNE_TRY(stream_order(ctx, aux));
            if (lanczos) {
                // the Gershgorin bound is loose on the coarse operators (2.8 against ~1.7): it is tightened with the largest
                // Ritz value of 8 Lanczos steps (converges from below; 8 % margin keeps the sweep stable; 12 steps gave the
                // same iteration counts on the configs, on 300 random systems and at 40 / 160 M unknowns, for 0.6 ms more).
                // Level 0: the bound is tight (1.99 by Lanczos).
                Pending *pj = new Pending();
                pj->level = lvl;
                pending.push_back(pj);
                PADNE_TRY(lanczos_enqueue(aux, A, kLanczosSteps, &pj->job));
            }
            if (want_f32) PADNE_TRY(csr_build_f32(aux, const_cast<padne_csr *>(A)));
            // (and of the level's R and P, once they exist; P only where the cycle will have no W to go up with: see the end of the setup)
            if (want_f32 && L.P != nullptr && !w_expected) PADNE_TRY(csr_build_f32(aux, L.P));
            if (want_f32 && L.R != nullptr) PADNE_TRY(csr_build_f32(aux, L.R));
            return PADNE_OK;
        };
        if (coarsest && (rc = queue_level_extras()) != PADNE_OK) { amg->levels.push_back(L); break; }
        if (coarsest) {
            // (no bound of D^-1 A here: the coarsest operator is inverted, never smoothed -- the bound cost a pass of one thread
            // per row over rows of hundreds of entries and a look at the host in front of the inverse, 50 us)
            amg->levels.push_back(L);
            break;
        }
        Scratch sc(ctx);
        int *agg = nullptr, n_agg = 0;
        PhaseTimer pt(ctx, amg_verbose());
        double lambda_f = 2.0;      // Gershgorin bound of the filtered operator, a by-product of the strength pass
        unsigned char *spos = nullptr;
        int *scol = nullptr;
        // (the single-precision copy of A's values comes out of the strength pass, which streams them anyway)
        float *v32 = nullptr;
        if (want_f32 && A->vals32 == nullptr && A->nnz > 0) {
            padne_csr *Am = const_cast<padne_csr *>(A);
            v32 = (float *)pool_alloc(Am->owner ? Am->owner : ctx, sizeof(float) * ((size_t)A->nnz + kPadNnz));      // padded like vals
            if (v32 == nullptr) { rc = PADNE_E_NOMEM; amg->levels.push_back(L); break; }
            Am->vals32 = v32;
        }
        if ((rc = aggregate(ctx, sc, A, &agg, &n_agg, &lambda_f, &L.lambda, &spos, &scol, v32)) != PADNE_OK) { amg->levels.push_back(L); break; }
        pt.lap("aggregate");
        if (n_agg == 0 || (double)n_agg > 0.8 * (double)A->n_rows) {   // coarsening stalled: stop here
            rc = queue_level_extras();
            amg->levels.push_back(L);
            break;
        }
        // omega uses Gershgorin bounds (filtered operator, capped by the unfiltered one): the sharper Lanczos
        // estimate of lambda(D^-1 A) over-relaxes the prolongator (44 instead of 34 CG iterations at N = 0.5 M)
        if (L.lambda < lambda_f) lambda_f = L.lambda;
        const double omega = kOmegaNum / lambda_f;
        if (amg_verbose())
            fprintf(stderr, "[amg] level %d: n=%lld nnz=%lld lambda=%.3f (P: %.3f) -> %d aggregates\n", lvl,
                    (long long)A->n_rows, (long long)A->nnz, L.lambda, lambda_f, n_agg);
        padne_csr *AP = nullptr, *Ac = nullptr;
        if ((rc = build_prolongator(ctx, A, agg, n_agg, omega, &L.P, spos, scol)) != PADNE_OK) { amg->levels.push_back(L); break; }
        pt.lap("prolongator");
        if (amg_verbose()) fprintf(stderr, "[amg]   P: %lld x %lld nnz=%lld\n", (long long)L.P->n_rows, (long long)L.P->n_cols, (long long)L.P->nnz);
        // second stream: R = P^T next to A P on the main stream.  Its dozen launches are made after the count pass of A P
        // has been queued (the host would otherwise sit in front of the scan's total anyway): P is complete at the event
        // recorded here
        if (two && (rc = stream_order(ctx, aux)) != PADNE_OK) { amg->levels.push_back(L); break; }
        const std::function<int()> queue_transpose = [&]() -> int { return transpose(aux, L.P, &L.R); };
        // A P stays in the slots its rows were merged in: R (A P) reads it by rows, a compacted copy would be written
        // and read once for nothing (the product comes back as a matrix only when it had to be split, see spgemm)
        SlotRows ap_rows;
        if ((rc = spgemm(ctx, A, L.P, &AP, nullptr, &ap_rows, &queue_transpose)) != PADNE_OK) { amg->levels.push_back(L); break; }
        if (L.R == nullptr && (rc = transpose(aux, L.P, &L.R)) != PADNE_OK) { amg->levels.push_back(L); break; }   // (the product was split: its halves do not call back)
        pt.lap("A*P (P^T queued beside it)");
        // fine level of the float cycle: W = P - c D^-1 A P from the slots of A P, on the second stream
        const bool with_w = lvl == 0 && want_f32 && ap_rows.valid && ctx->opt.amg_w >= 1;
        if (amg_verbose() && AP != nullptr) fprintf(stderr, "[amg]   AP nnz=%lld\n", (long long)AP->nnz);
        if (two && (rc = stream_order(aux, ctx)) != PADNE_OK) { if (AP) padne_csr_destroy(AP); amg->levels.push_back(L); break; }
        // What the second stream gets next -- W of the fine level, the level's extras: some forty launches -- is queued
        // once the row kernels of R (A P) are: made in front of that product they kept this stream waiting for the host
        // (0.4 ms per level in a trace)
        // (the second stream may start on it now -- the event of what it reads, the slots of A P included, is recorded here, in
        // front of the product, not behind its row kernels)
        if (two && (rc = stream_order(ctx, aux)) != PADNE_OK) { if (AP) padne_csr_destroy(AP); amg->levels.push_back(L); break; }
        w_expected = with_w || (lvl > 0 && want_f32 && w_inner && ap_rows.valid);
        bool side_queued = false;
        const std::function<int()> queue_side = [&]() -> int {
            side_queued = true;
            if (with_w) {                   // after the join above: the main stream waits for R, not for W
                const double c0 = 1.0 / (0.5 * (L.lambda + L.lambda / kChebRatio));      // level 0 keeps its Gershgorin bound: L.jac below
                PADNE_TRY(build_w_operator(aux, A, L.P, ap_rows, ap_rows.n_slots, c0, &L.W));
            }
            return queue_level_extras(true);
        };
        if (ap_rows.valid) {
            padne_csr ap_shape;
            ap_shape.n_rows = ap_rows.n_rows;
            ap_shape.n_cols = ap_rows.n_cols;
            rc = spgemm(ctx, L.R, &ap_shape, &Ac, &ap_rows, nullptr, nullptr, &queue_side);
        } else {
            rc = spgemm(ctx, L.R, AP, &Ac, nullptr, nullptr, nullptr, &queue_side);
        }
        if (rc == PADNE_OK && !side_queued) rc = queue_side();      // (the product was split: its halves do not call back)
        if (rc != PADNE_OK && two) (void)hipStreamSynchronize(aux->stream);      // it may have queued reads of the slots that go with this scope
        pt.lap("R*(AP)");
        if (amg_verbose() && rc == PADNE_OK) fprintf(stderr, "[amg]   Ac: n=%lld nnz=%lld\n", (long long)Ac->n_rows, (long long)Ac->nnz);
        if (AP) padne_csr_destroy(AP);
        auto take_slots = [](SlotRows &to, SlotRows &from) {
            std::swap(to.ctx, from.ctx);
            std::swap(to.begin, from.begin);
            std::swap(to.end, from.end);
            std::swap(to.key, from.key);
            std::swap(to.val, from.val);
            to.n_rows = from.n_rows;
            to.n_cols = from.n_cols;
            to.n_slots = from.n_slots;
            to.valid = from.valid;
        };
        if (with_w) {                       // the second stream may still be reading the slots: they go when it has been joined
            take_slots(ap_keep, ap_rows);
        } else if (lvl > 0 && rc == PADNE_OK && want_f32 && w_inner && ap_rows.valid) {
            ap_inner.emplace_back(new KeptSlots());
            ap_inner.back()->level = lvl;
            take_slots(ap_inner.back()->rows, ap_rows);
        }
        ap_rows.release();
        amg->levels.push_back(L);
        if (rc != PADNE_OK) break;
        if ((rc = csr_build_dinv(ctx, Ac)) != PADNE_OK) { padne_csr_destroy(Ac); break; }
        Ac->hierarchy_operator = true;
        amg->levels.back().P->hierarchy_operator = true;
        amg->levels.back().R->hierarchy_operator = true;
        A = Ac;
    }
    if (rc == PADNE_OK) {
        const AmgLevel &last = amg->levels.back();
        if (last.P != nullptr) {
            rc = PADNE_E_NOCOARSEN;
            set_error("multigrid setup did not reach a coarsest level");
        } else if (last.n > 4096) {
            rc = PADNE_E_NOCOARSEN;
            set_error("multigrid coarsening stalled at %lld unknowns", last.n);
        } else {
            amg->n_coarse = (int)last.n;
            PhaseTimer pd(ctx, amg_verbose());
            // the W operators of the inner levels are built on the second stream while this stream inverts: they read P,
            // 1/diag and the kept slots of A P, all written on this stream -- ordered behind what is queued here so far (an
            // event wait in front of the inverse, not behind it)
            if (two && !ap_inner.empty()) rc = stream_order(ctx, aux);
            if (rc == PADNE_OK) rc = dense_inverse(ctx, last.A, &amg->coarse_inv);
            pd.lap("dense inverse");
        }
    }
    if (rc != PADNE_OK) {
        drop_pending();
        if (two) (void)hipStreamSynchronize(aux->stream);
        amg_destroy(amg);
        return rc;
    }
    // smoother damping of every level: first-degree Chebyshev on [lambda / ratio, lambda]
    for (Pending *pj : pending) {
        double ritz = 0.0;
        const int rcl = lanczos_finish(&pj->job, &ritz);
        if (rcl != PADNE_OK && rc == PADNE_OK) rc = rcl;
        AmgLevel &L = amg->levels[(size_t)pj->level];
        const double est = 1.08 * ritz;
        if (rcl == PADNE_OK && est > 0.0 && est < L.lambda) L.lambda = est;
        delete pj;
    }
    pending.clear();
    for (AmgLevel &L : amg->levels) L.jac = 1.0 / (0.5 * (L.lambda + L.lambda / kChebRatio));
    // W of the inner levels (the second stream is idle here, the main one is inverting the coarsest operator)
    for (auto &ks : ap_inner) {
        if (rc != PADNE_OK) break;
        AmgLevel &L = amg->levels[(size_t)ks->level];
        if (L.P == nullptr || L.A->dinv == nullptr) continue;
        rc = build_w_operator(aux, L.A, L.P, ks->rows, ks->rows.n_slots, L.jac, &L.W, 512, false);
    }
    // a level that came out without W goes up with P: its single-precision copy, left out above
    for (AmgLevel &L : amg->levels)
        if (rc == PADNE_OK && want_f32 && L.P != nullptr && L.W == nullptr) rc = csr_build_f32(aux, L.P);
    if (rc == PADNE_OK && two) rc = stream_order(aux, ctx);      // the cycle runs on the main stream
    if (rc != PADNE_OK) {
        if (two) (void)hipStreamSynchronize(aux->stream);
        amg_destroy(amg);
        return rc;
    }
    amg->operator_complexity = nnz_total / (double)(A0->nnz > 0 ? A0->nnz : 1);
    if (want_f32 && amg->levels.size() >= 2) {
        hipStream_t s = ctx->stream;
        if (amg->levels[0].b == nullptr && (rc = alloc_vec(ctx, &amg->levels[0].b, amg->levels[0].n)) != PADNE_OK) {
            amg_destroy(amg);
            return rc;
        }
        if (amg->n_coarse > 0) {
            const size_t cnt = (size_t)amg->n_coarse * (size_t)amg->n_coarse;
            amg->coarse_inv32 = (float *)pool_alloc(ctx, sizeof(float) * cnt);
            if (amg->coarse_inv32 == nullptr) {
                amg_destroy(amg);
                return PADNE_E_NOMEM;
            }
            hipLaunchKernelGGL(f32_copy_amg, dim3(nblk((long long)cnt)), dim3(256), 0, s, (long long)cnt, amg->coarse_inv,
                               amg->coarse_inv32);
        }
        amg->f32 = true;
    }
    // (every exit below leaves through one place: a hierarchy that is not handed to the matrix is destroyed, after both
    // streams have drained)
    float ms = 0.f;
    hipError_t he = hipGetLastError();
    if (he == hipSuccess) he = hipEventRecord(ctx->ev1, ctx->stream);
    if (he == hipSuccess) he = hipEventSynchronize(ctx->ev1);
    if (he == hipSuccess) he = hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    if (he != hipSuccess) {
        set_error("multigrid setup: %s", hipGetErrorString(he));
        (void)hipStreamSynchronize(ctx->stream);
        if (two) (void)hipStreamSynchronize(aux->stream);
        amg_destroy(amg);
        return PADNE_E_HIP;
    }
    amg->setup_seconds = ms * 1e-3;
    A0->amg = amg;
    return PADNE_OK;
}

// ---- row-partitioned hierarchy ---------------------------------------------------------------------
// One global hierarchy over all ranks whose aggregates never cross a rank boundary: the prolongator is block
// diagonal (built from the rank's own block of A_l), so restriction and prolongation stay local and the
// Galerkin product needs the remote rows of P only for the exchanged vertices.  Every level is again a
// row-partitioned operator [owned | world * m_l exchange slots] with its own exchange plan; the V-cycle
// exchanges the smoothed iterate twice per level.  The coarsest operator is gathered to every rank and
// inverted densely.

// cnt[i] = entries of row i in the owned columns
__global__ void block_count(int n, const int *__restrict__ rowptr, const int *__restrict__ cols, int n_own,
                            int *__restrict__ cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int c = 0;
    for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) c += cols[k] < n_own ? 1 : 0;
    cnt[i] = c;
}

// owned x owned block of the rank's rows.  The dropped couplings are lumped onto the diagonal (row sums of
// the block equal those of the full operator, so the smoothed prolongator keeps its unit row sums at the
// exchanged vertices) unless that would eat more than 70 % of the diagonal.
__global__ void block_fill(int n, const int *__restrict__ rowptr, const int *__restrict__ cols,
                           const double *__restrict__ vals, int n_own, const int *__restrict__ optr,
                           int *__restrict__ ocols, double *__restrict__ ovals) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int o = optr[i], dpos = -1;
    double dropped = 0.0;
    for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
        const int c = cols[k];
        if (c < n_own) {
            ocols[o] = c;
            ovals[o] = vals[k];
            if (c == i) dpos = o;
            ++o;
        } else {
            dropped += vals[k];
        }
    }
    if (dpos >= 0 && dropped != 0.0) {
        const double d = ovals[dpos];
        if (d + dropped > 0.3 * d) ovals[dpos] = d + dropped;
    }
}

// rows idx[0..ne) of a CSR matrix into fixed-width records of K entries
__global__ void extract_rows(int ne, const int *__restrict__ idx, const int *__restrict__ rowptr,
                             const int *__restrict__ cols, const double *__restrict__ vals, int K,
                             int *__restrict__ len, int *__restrict__ oc, double *__restrict__ ov) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ne) return;
    const int i = idx[e];
    const int L = rowptr[i + 1] - rowptr[i];
    len[e] = L;
    for (int k = 0; k < L && k < K; ++k) {
        oc[(size_t)e * K + k] = cols[rowptr[i] + k];
        ov[(size_t)e * K + k] = vals[rowptr[i] + k];
    }
}

// all-gather of equally sized host records through the context's communicator
static int host_allgather(padne_ctx *ctx, const std::vector<double> &mine, std::vector<double> &all) {
    const size_t cnt = mine.size();
    const int W = ctx->world;
    all.assign(cnt * (size_t)W, 0.0);
    if (cnt == 0) return PADNE_OK;
    PADNE_REQUIRE(cnt < (1u << 30), "exchange record too large");
    Scratch sc(ctx);
    double *d = nullptr;
    PADNE_TRY(sc.alloc(&d, cnt * (size_t)W));
    PADNE_HIP_CHECK(hipMemcpyAsync(d + cnt * (size_t)ctx->rank, mine.data(), sizeof(double) * cnt, hipMemcpyHostToDevice,
                                   ctx->stream));
    if (W > 1) PADNE_TRY(comm_allgather_f64(ctx, d + cnt * (size_t)ctx->rank, d, (int)cnt));
    PADNE_HIP_CHECK(hipMemcpyAsync(all.data(), d, sizeof(double) * cnt * (size_t)W, hipMemcpyDeviceToHost, ctx->stream));
    PADNE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PADNE_OK;
}

static int upload_csr(padne_ctx *ctx, long long n_rows, long long n_cols, const std::vector<int> &rowptr,
                      const std::vector<int> &cols, const std::vector<double> &vals, padne_csr **out) {
    padne_csr *m = nullptr;
    PADNE_TRY(csr_alloc(ctx, n_rows, n_cols, (long long)cols.size(), &m));
    hipError_t e = hipMemcpyAsync(m->rowptr, rowptr.data(), sizeof(int) * (size_t)(n_rows + 1), hipMemcpyHostToDevice,
                                  ctx->stream);
    if (e == hipSuccess && !cols.empty()) {
        e = hipMemcpyAsync(m->cols, cols.data(), sizeof(int) * cols.size(), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync(m->vals, vals.data(), sizeof(double) * vals.size(), hipMemcpyHostToDevice, ctx->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        set_error("matrix upload failed: %s", hipGetErrorString(e));
        padne_csr_destroy(m);
        return PADNE_E_HIP;
    }
    *out = m;
    return PADNE_OK;
}

static int owned_block(padne_ctx *ctx, const padne_csr *A, long long n_own, padne_csr **blk) {
    hipStream_t s = ctx->stream;
    const int n = (int)n_own;
    Scratch sc(ctx);
    int *cnt = nullptr, *optr = nullptr;
    PADNE_TRY(sc.alloc(&cnt, (size_t)n + 1));
    PADNE_TRY(sc.alloc(&optr, (size_t)n + 1));
    PADNE_HIP_CHECK(hipMemsetAsync(cnt, 0, sizeof(int) * (size_t)(n + 1), s));
    hipLaunchKernelGGL(block_count, dim3(nblk(n)), dim3(256), 0, s, n, A->rowptr, A->cols, n, cnt);
    PADNE_HIP_CHECK(hipGetLastError());
    int64_t nnz = 0;
    PADNE_TRY(exclusive_scan_i32(ctx, cnt, optr, n, &nnz));
    padne_csr *m = nullptr;
    PADNE_TRY(csr_alloc(ctx, n, n, nnz, &m));
    hipError_t e = hipMemcpyAsync(m->rowptr, optr, sizeof(int) * (size_t)(n + 1), hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(block_fill, dim3(nblk(n)), dim3(256), 0, s, n, A->rowptr, A->cols, A->vals, n, m->rowptr,
                           m->cols, m->vals);
        e = hipGetLastError();
    }
    if (e != hipSuccess) {
        set_error("owned block extraction failed: %s", hipGetErrorString(e));
        padne_csr_destroy(m);
        return PADNE_E_HIP;
    }
    *blk = m;
    return PADNE_OK;
}

// The rows of P that belong to exchanged vertices, as every other rank needs them: rows of the exchange
// slots [world * m] over the coarse columns [n_agg | world * m_c].  Also returns this rank's coarse export
// list (the aggregates its exported rows touch) and the common coarse segment length m_c.
static int exchange_prolongator_rows(padne_ctx *ctx, const AmgLevel &L, const padne_csr *P, int n_agg,
                                     std::vector<int> &export_c, int *m_c_out, padne_csr **P_halo) {
    const int W = ctx->world, rank = ctx->rank, m = L.halo.m, ne = L.halo.n_export;
    hipStream_t s = ctx->stream;
    int K = 16, kmax = 0;
    std::vector<int> len((size_t)ne), pc;
    std::vector<double> pv;
    if (ne > 0) {
        for (int pass = 0; pass < 2; ++pass) {
            Scratch sc(ctx);
            int *d_len = nullptr, *d_pc = nullptr;
            double *d_pv = nullptr;
            PADNE_TRY(sc.alloc(&d_len, (size_t)ne));
            PADNE_TRY(sc.alloc(&d_pc, (size_t)ne * K));
            PADNE_TRY(sc.alloc(&d_pv, (size_t)ne * K));
            hipLaunchKernelGGL(extract_rows, dim3(nblk(ne)), dim3(256), 0, s, ne, L.halo.export_idx, P->rowptr, P->cols,
                               P->vals, K, d_len, d_pc, d_pv);
            PADNE_HIP_CHECK(hipGetLastError());
            pc.resize((size_t)ne * K);
            pv.resize((size_t)ne * K);
            PADNE_HIP_CHECK(hipMemcpyAsync(len.data(), d_len, sizeof(int) * (size_t)ne, hipMemcpyDeviceToHost, s));
            PADNE_HIP_CHECK(hipMemcpyAsync(pc.data(), d_pc, sizeof(int) * pc.size(), hipMemcpyDeviceToHost, s));
            PADNE_HIP_CHECK(hipMemcpyAsync(pv.data(), d_pv, sizeof(double) * pv.size(), hipMemcpyDeviceToHost, s));
            PADNE_HIP_CHECK(hipStreamSynchronize(s));
            kmax = 0;
            for (int e = 0; e < ne; ++e) kmax = std::max(kmax, len[(size_t)e]);
            if (kmax <= K) break;
            K = kmax;
        }
    }
    export_c.clear();
    for (int e = 0; e < ne; ++e)
        for (int k = 0; k < len[(size_t)e]; ++k) export_c.push_back(pc[(size_t)e * K + k]);
    std::sort(export_c.begin(), export_c.end());
    export_c.erase(std::unique(export_c.begin(), export_c.end()), export_c.end());
    std::vector<double> head = {(double)ne, (double)export_c.size(), (double)kmax}, heads;
    PADNE_TRY(host_allgather(ctx, head, heads));
    int m_c = 0, Kg = 1;
    for (int q = 0; q < W; ++q) {
        PADNE_REQUIRE((int)heads[(size_t)q * 3] <= m, "export list longer than the exchange segment");
        m_c = std::max(m_c, (int)heads[(size_t)q * 3 + 1]);
        Kg = std::max(Kg, (int)heads[(size_t)q * 3 + 2]);
    }
    // fixed-width records (position in my coarse export list, value); position -1 = unused
    std::vector<double> rec((size_t)m * Kg * 2, -1.0), recs;
    for (int e = 0; e < ne; ++e)
        for (int k = 0; k < len[(size_t)e]; ++k) {
            const int c = pc[(size_t)e * K + k];
            const int pos = (int)(std::lower_bound(export_c.begin(), export_c.end(), c) - export_c.begin());
            rec[((size_t)e * Kg + k) * 2] = (double)pos;
            rec[((size_t)e * Kg + k) * 2 + 1] = pv[(size_t)e * K + k];
        }
    PADNE_TRY(host_allgather(ctx, rec, recs));
    std::vector<int> rowptr((size_t)W * m + 1, 0), cols;
    std::vector<double> vals;
    for (int q = 0; q < W; ++q)
        for (int e = 0; e < m; ++e) {
            if (q != rank) {
                const double *r = recs.data() + ((size_t)q * m + e) * Kg * 2;
                for (int k = 0; k < Kg; ++k)
                    if (r[2 * k] >= 0.0) {
                        cols.push_back(n_agg + q * m_c + (int)r[2 * k]);
                        vals.push_back(r[2 * k + 1]);
                    }
            }
            rowptr[(size_t)q * m + e + 1] = (int)cols.size();
        }
    *m_c_out = m_c;
    return upload_csr(ctx, (long long)W * m, (long long)n_agg + (long long)W * m_c, rowptr, cols, vals, P_halo);
}

// padded per-rank pieces [world][n_pad] -> the gathered vector [sum n_q]
template <typename T>
__global__ void compact_pieces(int world, int n_pad, const int *__restrict__ seg_off, const T *__restrict__ src,
                               double *__restrict__ dst, const int *__restrict__ done_flag) {
    if (done_flag != nullptr && *done_flag != 0) return;
    const int q = blockIdx.y;
    const int nq = seg_off[q + 1] - seg_off[q];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += gridDim.x * blockDim.x)
        dst[seg_off[q] + i] = (double)src[(size_t)q * n_pad + i];
}

// the gathered solution as the level above sees its coarse vector: [this rank's piece | every exchange slot of the level]
__global__ void tail_to_ext_kernel(long long n, int n_slots, const double *__restrict__ tail_z, long long tail_off,
                                   const int *__restrict__ slot_idx, float *__restrict__ e_ext,
                                   const int *__restrict__ done_flag) {
    if (done_flag != nullptr && *done_flag != 0) return;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) e_ext[i] = (float)tail_z[tail_off + i];
    else if (i < n + n_slots) e_ext[i] = (float)tail_z[slot_idx[i - n]];
}

__global__ void f32_from_f64_kernel(long long n, const double *__restrict__ src, float *__restrict__ dst,
                                    const int *__restrict__ done_flag) {
    if (done_flag != nullptr && *done_flag != 0) return;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (float)src[i];
}

// this rank's piece of the gather level as one record of doubles: rowptr[n_pad + 1] | cols[nnz_max] (already in
// gathered numbering) | vals[nnz_max]
__global__ void tail_pack(int n, int nnz, const int *__restrict__ rowptr, const int *__restrict__ cols,
                          const double *__restrict__ vals, int my_off, const int *__restrict__ slot_to_global,
                          long long o_cols, long long o_vals, double *__restrict__ rec) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t <= n) rec[t] = (double)rowptr[t];
    if (t < nnz) {
        const int j = cols[t];
        rec[o_cols + t] = (double)(j < n ? my_off + j : slot_to_global[j - n]);
        rec[o_vals + t] = vals[t];
    }
}

// records of all ranks -> one CSR matrix, rows in rank order
__global__ void tail_unpack(const double *__restrict__ recs, long long rec_len, long long o_cols, long long o_vals,
                            const int *__restrict__ seg_off, const int *__restrict__ nnz_off, int *__restrict__ rowptr,
                            int *__restrict__ cols, double *__restrict__ vals) {
    const int q = blockIdx.y;
    const double *rec = recs + (size_t)q * rec_len;
    const int nq = seg_off[q + 1] - seg_off[q], zq = nnz_off[q + 1] - nnz_off[q];
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < nq || t < zq; t += gridDim.x * blockDim.x) {
        if (t < nq) rowptr[seg_off[q] + t] = nnz_off[q] + (int)rec[t];
        if (t < zq) {
            cols[nnz_off[q] + t] = (int)rec[o_cols + t];
            vals[nnz_off[q] + t] = rec[o_vals + t];
        }
    }
    if (q == gridDim.y - 1 && blockIdx.x == 0 && threadIdx.x == 0) rowptr[seg_off[q + 1]] = nnz_off[q + 1];
}

// Gather the last row-partitioned level to every rank (rows in rank order) and build an ordinary hierarchy
// on it: the small levels are cheaper to run redundantly than to exchange, and their aggregates may then
// cross rank boundaries, which matters once the couplings between ranks dominate a coarse operator.
static int gather_tail(padne_ctx *ctx, Amg *amg) {
    AmgLevel &L = amg->levels.back();
    const int W = ctx->world, m = L.halo.m, n = (int)L.n;
    hipStream_t s = ctx->stream;
    const padne_csr *A = L.A;
    std::vector<double> head = {(double)n, (double)A->nnz, (double)L.halo.n_export}, heads;
    PADNE_TRY(host_allgather(ctx, head, heads));
    int n_pad = 1;
    long long nnz_max = 1;
    std::vector<int> off((size_t)W + 1, 0), zoff((size_t)W + 1, 0);
    for (int q = 0; q < W; ++q) {
        n_pad = std::max(n_pad, (int)heads[(size_t)q * 3]);
        nnz_max = std::max(nnz_max, (long long)heads[(size_t)q * 3 + 1]);
        off[(size_t)q + 1] = off[(size_t)q] + (int)heads[(size_t)q * 3];
        PADNE_REQUIRE((long long)zoff[(size_t)q] + (long long)heads[(size_t)q * 3 + 1] < 2147483647LL,
                      "gathered operator too large");
        zoff[(size_t)q + 1] = zoff[(size_t)q] + (int)heads[(size_t)q * 3 + 1];
    }
    const long long N = off[(size_t)W], nnz_sum = zoff[(size_t)W];
    // where every exchange slot lives in the gathered numbering
    std::vector<double> ex((size_t)m, 0.0), exs;
    for (int e = 0; e < L.halo.n_export; ++e) ex[(size_t)e] = (double)L.export_host[(size_t)e];
    PADNE_TRY(host_allgather(ctx, ex, exs));
    std::vector<int> slot_to_global((size_t)W * m + 1, 0);
    for (int p = 0; p < W; ++p)
        for (int e = 0; e < m; ++e) slot_to_global[(size_t)p * m + e] = off[(size_t)p] + (int)exs[(size_t)p * m + e];
    const long long o_cols = (long long)n_pad + 1, o_vals = o_cols + nnz_max, rec_len = o_vals + nnz_max;
    PADNE_REQUIRE(rec_len < (1LL << 30), "gather record too large");
    {
        Scratch sc(ctx);
        int *d_s2g = nullptr, *d_off = nullptr, *d_zoff = nullptr;
        double *recs = nullptr;
        PADNE_TRY(sc.alloc(&d_s2g, slot_to_global.size()));
        PADNE_TRY(sc.alloc(&d_off, (size_t)W + 1));
        PADNE_TRY(sc.alloc(&d_zoff, (size_t)W + 1));
        PADNE_TRY(sc.alloc(&recs, (size_t)rec_len * W));
        PADNE_HIP_CHECK(hipMemcpyAsync(d_s2g, slot_to_global.data(), sizeof(int) * slot_to_global.size(),
                                       hipMemcpyHostToDevice, s));
        PADNE_HIP_CHECK(hipMemcpyAsync(d_off, off.data(), sizeof(int) * ((size_t)W + 1), hipMemcpyHostToDevice, s));
        PADNE_HIP_CHECK(hipMemcpyAsync(d_zoff, zoff.data(), sizeof(int) * ((size_t)W + 1), hipMemcpyHostToDevice, s));
        double *mine = recs + (size_t)ctx->rank * rec_len;
        const long long work = std::max<long long>((long long)n + 1, A->nnz);
        hipLaunchKernelGGL(tail_pack, dim3(nblk(work)), dim3(256), 0, s, n, (int)A->nnz, A->rowptr, A->cols, A->vals,
                           off[(size_t)ctx->rank], d_s2g, o_cols, o_vals, mine);
        PADNE_HIP_CHECK(hipGetLastError());
        if (W > 1) PADNE_TRY(comm_allgather_f64(ctx, mine, recs, (int)rec_len));
        PADNE_TRY(csr_alloc(ctx, N, N, nnz_sum, &amg->tail));
        const long long per_rank = std::max<long long>(n_pad, nnz_max);
        hipLaunchKernelGGL(tail_unpack, dim3((unsigned)std::min<long long>(nblk(per_rank), 4096), W), dim3(256), 0, s, recs,
                           rec_len, o_cols, o_vals, d_off, d_zoff, amg->tail->rowptr, amg->tail->cols, amg->tail->vals);
        PADNE_HIP_CHECK(hipGetLastError());
        PADNE_HIP_CHECK(hipStreamSynchronize(s));   // the staging arrays go back to the pool
    }
    amg->tail->hierarchy_operator = true;
    PADNE_TRY(amg_setup(ctx, amg->tail));
    {
        // the level above the gather level computes its neighbours' corrected values from the tail solution: where every
        // exchange slot of the gather level lives in the gathered numbering, and room for [own aggregates | those slots]
        const size_t nl = amg->levels.size();
        for (size_t l = 0; l + 2 < nl; ++l)
            if (amg->levels[l].P_halo != nullptr) {
                padne_csr_destroy(amg->levels[l].P_halo);
                amg->levels[l].P_halo = nullptr;
            }
        if (nl >= 2 && amg->levels[nl - 2].P_halo != nullptr && ctx->opt.amg_exchange_all) {
            padne_csr_destroy(amg->levels[nl - 2].P_halo);      // (A/B and tests: every level exchanges twice per cycle)
            amg->levels[nl - 2].P_halo = nullptr;
        }
        if (nl >= 2 && amg->levels[nl - 2].P_halo != nullptr) {
            amg->tail_slot_idx = (int *)pool_alloc(ctx, sizeof(int) * ((size_t)W * m + 1));
            amg->levels[nl - 2].e_ext = (float *)pool_alloc(ctx, sizeof(float) * ((size_t)n + (size_t)W * m + 1));
            if (amg->tail_slot_idx == nullptr || amg->levels[nl - 2].e_ext == nullptr) return PADNE_E_NOMEM;
            PADNE_HIP_CHECK(hipMemcpyAsync(amg->tail_slot_idx, slot_to_global.data(), sizeof(int) * ((size_t)W * m + 1),
                                           hipMemcpyHostToDevice, s));
            PADNE_HIP_CHECK(hipStreamSynchronize(s));
        }
    }
    amg->n_pad = n_pad;
    amg->tail_n = N;
    amg->tail_off = off[(size_t)ctx->rank];
    amg->seg_off = (int *)pool_alloc(ctx, sizeof(int) * ((size_t)W + 1));
    amg->coarse_gather = (double *)pool_alloc(ctx, sizeof(double) * (size_t)W * n_pad);
    amg->tail_r = (double *)pool_alloc(ctx, sizeof(double) * (size_t)(N > 0 ? N : 1));
    amg->tail_z = (double *)pool_alloc(ctx, sizeof(double) * (size_t)(N > 0 ? N : 1));
    if (!amg->seg_off || !amg->coarse_gather || !amg->tail_r || !amg->tail_z) return PADNE_E_NOMEM;
    PADNE_HIP_CHECK(hipMemcpyAsync(amg->seg_off, off.data(), sizeof(int) * ((size_t)W + 1), hipMemcpyHostToDevice, s));
    PADNE_HIP_CHECK(hipMemsetAsync(amg->coarse_gather, 0, sizeof(double) * (size_t)W * n_pad, s));
    PADNE_HIP_CHECK(hipStreamSynchronize(s));
    return PADNE_OK;
}

// global size at which the row-partitioned levels hand over to the gathered tail: the tail is replicated work (a
// fixed cost per rank), a partitioned level costs two exchanges per cycle -- with 4+ ranks the 100 k-unknown level
// is still cheaper partitioned
static int gather_n_limit(const padne_ctx *ctx, int world) {
    int v = ctx->opt.amg_gather_n > 0 ? (int)std::min<long long>(ctx->opt.amg_gather_n, 1LL << 30) : (world >= 4 ? 65536 : 262144);
    if (v < 64) v = 64;
    return v;
}

static int amg_setup_dist(padne_ctx *ctx, padne_csr *A0) {
    const int W = ctx->world;
    PADNE_REQUIRE(A0->n_rows == ctx->halo_n_owned && A0->n_cols == ctx->halo_n_owned + (long long)W * ctx->halo_m,
                  "row-partitioned multigrid: the matrix must be owned rows x [owned | world * m] columns");
    PADNE_TRY(csr_build_dinv(ctx, A0));
    hipStream_t s = ctx->stream;
    PADNE_HIP_CHECK(hipStreamSynchronize(s));
    const auto t_begin = std::chrono::steady_clock::now();
    t_amg_verbose = ctx->opt.verbose_amg;
    const long long gather_n = gather_n_limit(ctx, W);
    // single-precision cycle?  Decided up front: the fused up-leg operators W are read by that cycle only
    bool want_f32 = false;
    PADNE_TRY(decide_f32(ctx, A0, true, &want_f32));
    Amg *amg = new Amg();
    amg->device = ctx->device;
    amg->ctx = ctx;
    amg->dist = true;
    int rc = PADNE_OK;
    const padne_csr *A = A0;
    HaloPlan plan;
    plan.n_owned = ctx->halo_n_owned;
    plan.m = ctx->halo_m;
    plan.n_export = ctx->halo_n_export;
    plan.export_idx = ctx->halo_export;
    std::vector<int> export_host((size_t)plan.n_export);
    int32_t *export_owned = nullptr;       // device list owned by the level being built (levels >= 1)
    if (plan.n_export > 0) {
        hipError_t e = hipMemcpyAsync(export_host.data(), plan.export_idx, sizeof(int) * export_host.size(),
                                      hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) rc = PADNE_E_HIP;
    }
    double nnz_total = 0.0;
    bool stalled = false;
    for (int lvl = 0; lvl < kMaxLevels && rc == PADNE_OK; ++lvl) {
        AmgLevel L;
        L.A = A;
        L.A_owned = (lvl == 0) ? nullptr : const_cast<padne_csr *>(A);
        L.n = plan.n_owned;
        L.halo = plan;
        L.export_owned = export_owned;
        L.export_host = export_host;
        export_owned = nullptr;
        nnz_total += (double)A->nnz;
        amg->levels.push_back(L);
        AmgLevel &Lr = amg->levels.back();
        // interior / boundary tiles of this level's operator: its products overlap the halo exchange (amg_apply_f32)
        if (lvl > 0 && (rc = csr_build_split_plan(ctx, const_cast<padne_csr *>(A), plan.n_owned)) != PADNE_OK) break;
        double lam = 2.0;
        if ((rc = gershgorin(ctx, A, &lam)) != PADNE_OK) break;
        std::vector<double> st = {(double)Lr.n, lam}, sts;
        if ((rc = host_allgather(ctx, st, sts)) != PADNE_OK) break;
        long long n_glob = 0, n_max = 0;
        for (int q = 0; q < W; ++q) {
            n_glob += (long long)sts[(size_t)q * 2];
            n_max = std::max(n_max, (long long)sts[(size_t)q * 2]);
            lam = std::max(lam, sts[(size_t)q * 2 + 1]);
        }
        Lr.lambda = lam;
        // level 0 always stays row-partitioned (the CG vectors are); below it the levels are gathered as soon
        // as they are small enough to run redundantly
        const bool coarsest = lvl > 0 && (n_glob <= gather_n || lvl == kMaxLevels - 1);
        if (lvl > 0 && !coarsest) {
            double ritz = 0.0;
            if ((rc = estimate_lambda_max(ctx, A, kLanczosSteps, &ritz, &plan)) != PADNE_OK) break;
            const double est = 1.08 * ritz;
            if (est > 0.0 && est < Lr.lambda) Lr.lambda = est;
        }
        Lr.jac = 1.0 / (0.5 * (Lr.lambda + Lr.lambda / kChebRatio));
        if ((rc = alloc_vec(ctx, &Lr.xa, Lr.n + (long long)W * plan.m)) != PADNE_OK) break;
        if ((rc = alloc_vec(ctx, &Lr.tmp, A->n_rows)) != PADNE_OK) break;
        // (xb: the level's solution WITH its exchange area -- the W up-leg of the level above multiplies it)
        if (lvl > 0 && ((rc = alloc_vec(ctx, &Lr.b, Lr.n)) != PADNE_OK ||
                        (rc = alloc_vec(ctx, &Lr.xb, Lr.n + (long long)W * plan.m)) != PADNE_OK)) break;
        if (lvl > 0 && (rc = hipMemsetAsync(Lr.xb, 0, sizeof(double) * (size_t)(Lr.n + (long long)W * plan.m), s) == hipSuccess
                                 ? PADNE_OK : PADNE_E_HIP) != PADNE_OK) break;
        if ((rc = hipMemsetAsync(Lr.xa, 0, sizeof(double) * (size_t)(Lr.n + (long long)W * plan.m), s) == hipSuccess
                      ? PADNE_OK : PADNE_E_HIP) != PADNE_OK) break;
        if (coarsest) break;
        // aggregates and prolongator from the rank's own block
        PhaseTimer pt(ctx, amg_verbose() && ctx->rank == 0);
        padne_csr *blk = nullptr;
        if ((rc = owned_block(ctx, A, Lr.n, &blk)) != PADNE_OK) break;
        Scratch sc(ctx);
        int *agg = nullptr, n_agg = 0;
        double lambda_f = 2.0, lambda_g = 2.0;
        if ((rc = csr_build_dinv(ctx, blk)) == PADNE_OK && (rc = aggregate(ctx, sc, blk, &agg, &n_agg, &lambda_f)) == PADNE_OK)
            rc = gershgorin(ctx, blk, &lambda_g, false);
        if (rc != PADNE_OK) { padne_csr_destroy(blk); break; }
        std::vector<double> ag = {(double)n_agg}, ags;
        if ((rc = host_allgather(ctx, ag, ags)) != PADNE_OK) { padne_csr_destroy(blk); break; }
        long long agg_glob = 0;
        for (int q = 0; q < W; ++q) agg_glob += (long long)ags[(size_t)q];
        if (amg_verbose() && ctx->rank == 0)
            fprintf(stderr, "[amg] level %d: global n=%lld (this rank %lld, nnz %lld, m=%d) lambda=%.3f -> %lld aggregates\n",
                    lvl, n_glob, Lr.n, (long long)A->nnz, plan.m, Lr.lambda, agg_glob);
        if (agg_glob == 0 || (double)agg_glob > 0.8 * (double)n_glob) {
            // coarsening inside the ranks stalled: gather this level (level 0 cannot be: diagonal preconditioner,
            // the same decision on every rank)
            padne_csr_destroy(blk);
            stalled = true;
            if (lvl == 0) {
                rc = PADNE_E_NOCOARSEN;
                set_error("row-partitioned multigrid: no coarsening on the finest level");
            }
            break;
        }
        if (lambda_g < lambda_f) lambda_f = lambda_g;
        pt.lap("block+aggregate");
        rc = build_prolongator(ctx, blk, agg, n_agg, kOmegaNum / lambda_f, &Lr.P);
        padne_csr_destroy(blk);
        if (rc != PADNE_OK) break;
        if ((rc = transpose(ctx, Lr.P, &Lr.R)) != PADNE_OK) break;
        pt.lap("prolongator+transpose");
        padne_csr *P_halo = nullptr, *P_ext = nullptr, *AP = nullptr, *Ac = nullptr;
        std::vector<int> export_c;
        int m_c = 0;
        if ((rc = exchange_prolongator_rows(ctx, Lr, Lr.P, n_agg, export_c, &m_c, &P_halo)) != PADNE_OK) break;
        rc = csr_vstack(ctx, Lr.P, P_halo, (long long)n_agg + (long long)W * m_c, &P_ext);
        P_halo->hierarchy_operator = true;
        Lr.P_halo = P_halo;                 // kept: see AmgLevel (dropped again below on all but the last partitioned level)
        if (rc != PADNE_OK) break;
        pt.lap("exchange P rows");
        // A [P ; P_halo] stays in its merge slots as on one GPU (consumed by R (A P) row by row), and the up-leg operator of
        // the float cycle is formed from them: W = [P | 0] - c D^-1 A [P ; P_halo], columns [aggregates | exchange slots of
        // the coarse level] -- coarse correction and post-smoothing of the owned rows in ONE product with the coarse
        // solution and its exchanged values (amg_apply_f32), instead of prolongation, exchange of the corrected iterate and
        // a product with A
        SlotRows ap_rows;
        rc = spgemm(ctx, A, P_ext, &AP, nullptr, &ap_rows);
        if (rc != PADNE_OK) { padne_csr_destroy(P_ext); break; }
        if (ap_rows.valid) {
            padne_csr ap_shape;
            ap_shape.n_rows = ap_rows.n_rows;
            ap_shape.n_cols = ap_rows.n_cols;
            rc = spgemm(ctx, Lr.R, &ap_shape, &Ac, &ap_rows);
            if (rc == PADNE_OK && want_f32 && A->dinv != nullptr && (ctx->opt.amg_w == 2 || (ctx->opt.amg_w == 1 && lvl == 0))) {
                // (the twelve-run window plan pays on the fine level only, as on one GPU: ~1 % of an inner level's tiles qualify)
                rc = build_w_operator(ctx, A, Lr.P, ap_rows, ap_rows.n_slots, Lr.jac, &Lr.W, 0, lvl == 0);
                if (rc == PADNE_OK) Lr.W->n_cols = P_ext->n_cols;
            }
        } else {
            rc = spgemm(ctx, Lr.R, AP, &Ac);
        }
        ap_rows.release();
        padne_csr_destroy(P_ext);
        if (AP) padne_csr_destroy(AP);
        if (rc != PADNE_OK) break;
        if ((rc = csr_build_dinv(ctx, Ac)) != PADNE_OK) { padne_csr_destroy(Ac); break; }
        pt.lap("galerkin");
        Ac->hierarchy_operator = true;
        Lr.P->hierarchy_operator = true;
        Lr.R->hierarchy_operator = true;
        // exchange plan of the coarse level
        export_owned = (int32_t *)pool_alloc(ctx, sizeof(int32_t) * (export_c.empty() ? 1 : export_c.size()));
        if (export_owned == nullptr) { padne_csr_destroy(Ac); rc = PADNE_E_NOMEM; break; }
        if (!export_c.empty()) {
            hipError_t e = hipMemcpyAsync(export_owned, export_c.data(), sizeof(int) * export_c.size(),
                                          hipMemcpyHostToDevice, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) { padne_csr_destroy(Ac); pool_free(ctx, export_owned); export_owned = nullptr; rc = PADNE_E_HIP; break; }
        }
        plan.n_owned = n_agg;
        plan.m = m_c;
        plan.n_export = (int)export_c.size();
        plan.export_idx = export_owned;
        export_host = export_c;
        A = Ac;
    }
    if (export_owned != nullptr) pool_free(ctx, export_owned);
    if (rc == PADNE_OK) {
        if (amg->levels.back().P != nullptr && !stalled) {
            rc = PADNE_E_NOCOARSEN;
            set_error("multigrid setup did not reach a coarsest level");
        } else {
            PhaseTimer pg(ctx, amg_verbose() && ctx->rank == 0);
            rc = gather_tail(ctx, amg);
            pg.lap("gather + tail hierarchy");
        }
    }
    if (rc != PADNE_OK) {
        amg_destroy(amg);
        return rc;
    }
    amg->operator_complexity = nnz_total / (double)(A0->nnz > 0 ? A0->nnz : 1);
    if ((rc = enable_f32(ctx, amg, want_f32 ? 1 : 0)) != PADNE_OK) {
        amg_destroy(amg);
        return rc;
    }
    PADNE_HIP_CHECK(hipStreamSynchronize(s));
    amg->setup_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    A0->amg = amg;
    return PADNE_OK;
}

// z = M^-1 r on level 0 ; optional partial sums of r.z (written by the last kernel of the cycle)
int amg_apply(padne_ctx *ctx, const padne_csr *A0, const double *r, double *z, double *partials_rz,
              const int32_t *done_flag, const double *bb2, bool entry_done, float *z32 = nullptr);

// the same cycle on the single-precision copies; r comes in and z goes out in double -- or, with z32, in single precision
// and without the final multiplication by ||b|| (the caller's p update does it: 40 % less traffic for z)
static int amg_apply_f32(padne_ctx *ctx, Amg *amg, const double *r, double *z, double *partials_rz,
                         const int32_t *done_flag, const double *bb2, bool entry_done, float *z32) {
    hipStream_t s = ctx->stream;
    const int nl = (int)amg->levels.size();
    if (partials_rz == nullptr) {
        set_error("the single-precision cycle is only used as the CG preconditioner");
        return PADNE_E_INVALID;
    }
    bool fused_start = false;      // the restriction onto this level has already written its pre-smoothed start
    // last partitioned level of a row-partitioned hierarchy: the values of the other ranks' vertices after the coarse
    // correction, x1 + P e, are computed here -- x1 came with the down-leg exchange, their rows of P with the setup, e is
    // the tail solution every rank holds -- instead of a second exchange (same arithmetic as on the owning rank: bitwise
    // the same cycle; PADNE_AMG_EXCHANGE_ALL=1 exchanges as before)
    const bool local_halo = amg->dist && nl >= 2 && amg->levels[nl - 2].P_halo != nullptr &&
                            amg->levels[nl - 2].P_halo->vals32 != nullptr && amg->levels[nl - 2].e_ext != nullptr &&
                            amg->tail_slot_idx != nullptr;
    for (int l = 0; l < nl; ++l) {
        AmgLevel &L = amg->levels[l];
        float *b = (float *)L.b, *xa = (float *)L.xa, *tmp = (float *)L.tmp;
        if (l == nl - 1) {
            if (amg->dist) {
                // gather level: float pieces travel, the redundant tail cycle runs in double (launch-bound anyway)
                float *cg = (float *)amg->coarse_gather;
                float *seg = cg + (size_t)ctx->rank * amg->n_pad;
                PADNE_HIP_CHECK(hipMemcpyAsync(seg, b, sizeof(float) * (size_t)L.n, hipMemcpyDeviceToDevice, s));
                PADNE_TRY(comm_allgather_f32(ctx, seg, cg, amg->n_pad));
                hipLaunchKernelGGL(compact_pieces<float>, dim3(nblk(amg->n_pad), ctx->world), dim3(256), 0, s, ctx->world,
                                   amg->n_pad, amg->seg_off, (const float *)cg, amg->tail_r, done_flag);
                PADNE_HIP_CHECK(hipGetLastError());
                PADNE_TRY(amg_apply(ctx, amg->tail, amg->tail_r, amg->tail_z, nullptr, done_flag, nullptr, false));
                hipLaunchKernelGGL(f32_from_f64_kernel, dim3(nblk(L.n)), dim3(256), 0, s, L.n,
                                   (const double *)(amg->tail_z + amg->tail_off), (float *)L.xb, done_flag);
                if (local_halo) {
                    const int n_slots = ctx->world * L.halo.m;
                    hipLaunchKernelGGL(tail_to_ext_kernel, dim3(nblk(L.n + n_slots)), dim3(256), 0, s, L.n, n_slots,
                                       (const double *)amg->tail_z, amg->tail_off, (const int *)amg->tail_slot_idx,
                                       amg->levels[nl - 2].e_ext, done_flag);
                }
                PADNE_HIP_CHECK(hipGetLastError());
                break;
            }
            hipLaunchKernelGGL(dense_gemv<float>, dim3((amg->n_coarse + 3) / 4), dim3(256), 0, s, amg->n_coarse,
                               amg->n_coarse, (const float *)amg->coarse_inv32, (const float *)b, (float *)L.xb);
            PADNE_HIP_CHECK(hipGetLastError());
            break;
        }
        const int gv = (int)std::min<long long>((L.n + 255) / 256, 1024);
        if (l == 0 && entry_done) {
            // the caller's x / r update already wrote b = r / ||b|| and the first sweep (pcg_update_xr_entry_kernel)
        } else if (l == 0)
            hipLaunchKernelGGL(amg_entry_f32_kernel, dim3(gv > 0 ? gv : 1), dim3(256), 0, s, L.n, r, bb2, (float)L.jac,
                               (const float *)L.A->dinv32, b, xa, done_flag);
        else if (!fused_start)
            hipLaunchKernelGGL(scale_dinv_kernel<float>, dim3(gv > 0 ? gv : 1), dim3(256), 0, s, L.n, (float)L.jac,
                               (const float *)L.A->dinv32, (const float *)b, xa, done_flag);
        PADNE_HIP_CHECK(hipGetLastError());
        if (amg->dist) {
            // the pre-smoothed iterate goes out, the interior tiles of the residual run while it travels
            HaloTicket tk;
            PADNE_TRY(halo_send_f32(ctx, L.halo, xa, done_flag, &tk));
            PADNE_TRY(launch_spmv_f32_part(ctx, L.A, SPMV_RESID, SPMV_INTERIOR, xa, tmp, nullptr, done_flag, b, nullptr, 0.f));
            PADNE_TRY(halo_recv_f32(ctx, L.halo, xa, done_flag, tk));
            PADNE_TRY(launch_spmv_f32_part(ctx, L.A, SPMV_RESID, SPMV_BOUNDARY, xa, tmp, nullptr, done_flag, b, nullptr, 0.f));
        } else if (l == 0 && L.W != nullptr && spmv_resid_pre_ok(L.A)) {
            // the fine level never looks at its pre-smoothed iterate: the residual is formed from the right-hand side alone
            // (xa = c D^-1 b inside the staging of the product), the up-leg takes c D^-1 (b + residual) (spmv.hip)
            PADNE_TRY(launch_spmv_f32_resid_pre(ctx, L.A, b, tmp, done_flag, L.A->dinv32, (float)L.jac));
        } else {
            PADNE_TRY(launch_spmv_f32(ctx, L.A, SPMV_RESID, xa, tmp, nullptr, done_flag, b, nullptr, 0.f));
        }
        // the restriction also leaves the first sweep of the level below (from a zero start: x = c D^-1 b), unless that
        // level is the coarsest (solved directly) -- one short launch less per level
        AmgLevel &Lc = amg->levels[l + 1];
        fused_start = l + 1 < nl - 1 && Lc.A->dinv32 != nullptr;
        if (fused_start)
            PADNE_TRY(launch_spmv_f32_restrict(ctx, L.R, tmp, (float *)Lc.b, (float *)Lc.xa, done_flag, Lc.A->dinv32,
                                               (float)Lc.jac));
        else
            PADNE_TRY(launch_spmv_f32(ctx, L.R, SPMV_PLAIN, tmp, (float *)Lc.b, nullptr, done_flag, nullptr, nullptr, 0.f));
    }
    for (int l = nl - 2; l >= 0; --l) {
        AmgLevel &L = amg->levels[l];
        float *b = (float *)L.b, *xa = (float *)L.xa;
        if (L.W != nullptr && amg->dist) {
            // row-partitioned level: the same product, its input the coarse solution with the other ranks' exported values
            // behind it -- computed from the tail solution on the last partitioned level (e_ext), exchanged otherwise: one
            // exchange of the COARSE level's vector instead of one of this level's corrected iterate
            float *e = (float *)amg->levels[l + 1].xb;
            if (local_halo && l == nl - 2)
                e = L.e_ext;
            else
                PADNE_TRY(halo_exchange_plan_f32(ctx, amg->levels[l + 1].halo, e, done_flag));
            if (l == 0)
                PADNE_TRY(launch_spmv_f32_wup_exit(ctx, L.W, e, z, r, partials_rz, done_flag, xa, (const float *)L.tmp,
                                                   L.A->dinv32, (float)L.jac, bb2, z32, nullptr));
            else
                PADNE_TRY(launch_spmv_f32_wup(ctx, L.W, e, (float *)L.xb, done_flag, xa, (const float *)L.tmp, L.A->dinv32,
                                              (float)L.jac));
            continue;
        }
        if (L.W != nullptr && !amg->dist) {
            // coarse correction + post-smoothing (+ exit) in one product with W = P - c D^-1 A P (tmp still holds the
            // residual of the pre-smoothed iterate that the down-leg restricted)
            if (l == 0)
                PADNE_TRY(launch_spmv_f32_wup_exit(ctx, L.W, (const float *)amg->levels[1].xb, z, r, partials_rz, done_flag, xa,
                                                   (const float *)L.tmp, L.A->dinv32, (float)L.jac, bb2, z32,
                                                   spmv_resid_pre_ok(L.A) ? (const float *)b : nullptr));      // (as the down-leg decided)
            else
                PADNE_TRY(launch_spmv_f32_wup(ctx, L.W, (const float *)amg->levels[l + 1].xb, (float *)L.xb, done_flag, xa,
                                              (const float *)L.tmp, L.A->dinv32, (float)L.jac));
            continue;
        }
        PADNE_TRY(launch_spmv_f32(ctx, L.P, SPMV_ADD, (const float *)amg->levels[l + 1].xb, xa, nullptr, done_flag,
                                  nullptr, nullptr, 0.f));
        HaloTicket tk;
        const bool exchange = amg->dist && !(local_halo && l == nl - 2);
        if (amg->dist && !exchange)
            PADNE_TRY(launch_spmv_f32(ctx, L.P_halo, SPMV_ADD, (const float *)L.e_ext, xa + L.n, nullptr, done_flag, nullptr,
                                      nullptr, 0.f));
        // post-smoothing: with an exchange of the corrected iterate its interior tiles run while the halo travels
        for (int part = exchange ? SPMV_INTERIOR : SPMV_ALL; part <= (exchange ? SPMV_BOUNDARY : SPMV_ALL); ++part) {
            if (exchange && part == SPMV_INTERIOR) PADNE_TRY(halo_send_f32(ctx, L.halo, xa, done_flag, &tk));
            if (exchange && part == SPMV_BOUNDARY) PADNE_TRY(halo_recv_f32(ctx, L.halo, xa, done_flag, tk));
            if (l > 0)
                PADNE_TRY(launch_spmv_f32_part(ctx, L.A, SPMV_JACOBI, part, xa, (float *)L.xb, nullptr, done_flag, b, L.A->dinv32,
                                               (float)L.jac));
            else
                PADNE_TRY(launch_spmv_f32_exit_part(ctx, L.A, part, xa, z, r, partials_rz, done_flag, b, L.A->dinv32,
                                                    (float)L.jac, bb2, z32));
        }
    }
    return PADNE_OK;
}

// ---- the single-precision cycle on K = 8, 4 or 2 interleaved right-hand sides ([n][K] vectors, spmm.hip) ---------------
template <int K>
__global__ void amg_entry_f32xk_kernel(long long n, const double *__restrict__ r, const double *__restrict__ bb2, float c,
                                       const float *__restrict__ dinv, float *__restrict__ b, float *__restrict__ x,
                                       const int *__restrict__ done_flag) {
    if (done_flag != nullptr && *done_flag != 0) return;
    const int j = threadIdx.x & (K - 1);
    double s_inv = 1.0;
    if (bb2 != nullptr) {
        const double s2 = bb2[j];
        if (s2 > 0.0) s_inv = 1.0 / sqrt(s2);
    }
    // blockDim.x is a multiple of K and so is every stride: a thread keeps its right-hand side j
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n * K; t += (long long)gridDim.x * blockDim.x) {
        const float v = (float)(r[t] * s_inv);
        b[t] = v;
        if (x != nullptr) x[t] = c * dinv[t / K] * v;      // (null: the level forms its first sweep from b itself)
    }
}

template <int K>
__global__ void scale_dinv_xk_kernel(long long n, float c, const float *__restrict__ dinv, const float *__restrict__ b,
                                     float *__restrict__ x, const int *__restrict__ done_flag) {
    if (done_flag != nullptr && *done_flag != 0) return;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n * K; t += (long long)gridDim.x * blockDim.x)
        x[t] = c * dinv[t / K] * b[t];
}

// Y[row][j] = sum_c Inv[row][c] B[c][j] : one wave per row, lane = (c mod 64 / K, j)
template <int K>
__global__ __launch_bounds__(256) void dense_gemm_xk(int n, const float *__restrict__ inv, const float *__restrict__ b,
                                                     float *__restrict__ y) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    constexpr int CS = 64 / K;                              // columns of Inv a wave takes per turn
    const int cs = lane / K, j = lane & (K - 1);
    float s = 0.f;
    for (int c = cs; c < n; c += CS) s += inv[(size_t)row * n + c] * b[(size_t)c * K + j];
#pragma unroll
    for (int d = K; d < 64; d <<= 1) s += __shfl_xor(s, d, 64);
    if (lane < K) y[(size_t)row * K + j] = s;
}

static int amg_batch_vectors(padne_ctx *ctx, Amg *amg) {
    for (AmgLevel &L : amg->levels) {
        if (L.b8 != nullptr) continue;
        const size_t bytes = sizeof(float) * 8 * (size_t)(L.n > 0 ? L.n : 1);      // (room for the widest form)
        L.b8 = (float *)pool_alloc(ctx, bytes);
        L.xa8 = (float *)pool_alloc(ctx, bytes);
        L.xb8 = (float *)pool_alloc(ctx, bytes);
        L.tmp8 = (float *)pool_alloc(ctx, bytes);
        if (!L.b8 || !L.xa8 || !L.xb8 || !L.tmp8) return PADNE_E_NOMEM;
    }
    return PADNE_OK;
}

// buffers of the batched cycle's entry stage, for the caller that fuses it into its x / r update (pcg.hip, solve_batch)
int amg_batch_entry_args(padne_ctx *ctx, const padne_csr *A0, float *jac, const float **dinv32, float **b8, float **xa8) {
    Amg *amg = (Amg *)A0->amg;
    PADNE_REQUIRE(amg != nullptr && amg->f32 && !amg->dist && amg->levels.size() >= 2, "batched cycle");
    PADNE_TRY(amg_batch_vectors(ctx, amg));
    const AmgLevel &L = amg->levels[0];
    *jac = (float)L.jac;
    *dinv32 = L.A->dinv32;
    *b8 = L.b8;
    *xa8 = L.xa8;
    return PADNE_OK;
}

// z8 = M^-1 r8 for k interleaved right-hand sides; partials_rz [k][kMaxPartials]; bb2 [k]
template <int K>
static int amg_apply_batch_k(padne_ctx *ctx, const padne_csr *A0, const double *r8, double *z8, double *partials_rz,
                             const int32_t *done_flag, const double *bb2, const bool entry_done, float *z32) {
    Amg *amg = (Amg *)A0->amg;
    PADNE_REQUIRE(amg != nullptr && amg->f32 && !amg->dist, "the batched cycle needs the single-GPU single-precision hierarchy");
    hipStream_t s = ctx->stream;
    const int nl = (int)amg->levels.size();
    PADNE_TRY(amg_batch_vectors(ctx, amg));
    for (int l = 0; l < nl; ++l) {
        AmgLevel &L = amg->levels[l];
        if (l == nl - 1) {
            hipLaunchKernelGGL(dense_gemm_xk<K>, dim3((amg->n_coarse + 3) / 4), dim3(256), 0, s, amg->n_coarse,
                               (const float *)amg->coarse_inv32, (const float *)L.b8, L.xb8);
            PADNE_HIP_CHECK(hipGetLastError());
            break;
        }
        const int gv = (int)std::min<long long>((L.n * K + 255) / 256, 2048);
        if (l == 0 && entry_done) {
            // the caller's x / r update has written b = r / ||b|| and the first sweep (pcg8_update_xr_kernel)
        } else if (l == 0)
            hipLaunchKernelGGL(amg_entry_f32xk_kernel<K>, dim3(gv > 0 ? gv : 1), dim3(256), 0, s, L.n, r8, bb2, (float)L.jac,
                               (const float *)L.A->dinv32, L.b8, L.xa8, done_flag);
        else
            hipLaunchKernelGGL(scale_dinv_xk_kernel<K>, dim3(gv > 0 ? gv : 1), dim3(256), 0, s, L.n, (float)L.jac,
                               (const float *)L.A->dinv32, (const float *)L.b8, L.xa8, done_flag);
        PADNE_HIP_CHECK(hipGetLastError());
        PADNE_TRY(launch_spmm_f32(ctx, L.A, K, SPMV_RESID, L.xa8, L.tmp8, nullptr, done_flag, L.b8, nullptr, 0.f));
        PADNE_TRY(launch_spmm_f32(ctx, L.R, K, SPMV_PLAIN, L.tmp8, amg->levels[l + 1].b8, nullptr, done_flag, nullptr,
                                  nullptr, 0.f));
    }
    for (int l = nl - 2; l >= 0; --l) {
        AmgLevel &L = amg->levels[l];
        if (L.W != nullptr) {
            // coarse correction + post-smoothing (+ exit) in one product with W = P - c D^-1 A P, as in the single cycle (tmp8
            // still holds the residual of the pre-smoothed iterate that the down-leg restricted): 52 M instead of 24 + 70 M
            // non-zeros of the fine level per lockstep iteration
            if (l == 0)
                PADNE_TRY(launch_spmm_f32_wup_exit(ctx, L.W, K, amg->levels[1].xb8, z8, r8, partials_rz, done_flag,
                                                   L.A->dinv32 != nullptr ? (const float *)nullptr : (const float *)L.xa8, L.tmp8,
                                                   L.A->dinv32, (float)L.jac, bb2, z32,
                                                   L.A->dinv32 != nullptr ? (const float *)L.b8 : (const float *)nullptr));
            else
                PADNE_TRY(launch_spmm_f32_wup(ctx, L.W, K, amg->levels[l + 1].xb8, L.xb8, done_flag, L.xa8, L.tmp8,
                                              L.A->dinv32, (float)L.jac));
            continue;
        }
        PADNE_TRY(launch_spmm_f32(ctx, L.P, K, SPMV_ADD, amg->levels[l + 1].xb8, L.xa8, nullptr, done_flag, nullptr, nullptr,
                                  0.f));
        if (l > 0)
            PADNE_TRY(launch_spmm_f32(ctx, L.A, K, SPMV_JACOBI, L.xa8, L.xb8, nullptr, done_flag, L.b8, L.A->dinv32,
                                      (float)L.jac));
        else
            PADNE_TRY(launch_spmm_f32_exit(ctx, L.A, K, L.xa8, z8, r8, partials_rz, done_flag, L.b8, L.A->dinv32,
                                           (float)L.jac, bb2, z32));
    }
    return PADNE_OK;
}

// z32 (optional, instead of z8): z in single precision and without its factor sqrt(bb2[j]) -- the r.z partials are those of
// the double it stands for (the lockstep loop keeps z and the search directions as floats, like the single loop)
int amg_apply_batch(padne_ctx *ctx, const padne_csr *A0, int k, const double *r8, double *z8, double *partials_rz,
                    const int32_t *done_flag, const double *bb2, bool entry_done, float *z32) {
    switch (k) {
        case 8: return amg_apply_batch_k<8>(ctx, A0, r8, z8, partials_rz, done_flag, bb2, entry_done, z32);
        case 4: return amg_apply_batch_k<4>(ctx, A0, r8, z8, partials_rz, done_flag, bb2, entry_done, z32);
        case 2: return amg_apply_batch_k<2>(ctx, A0, r8, z8, partials_rz, done_flag, bb2, entry_done, z32);
        default: set_error("lockstep width %d", k); return PADNE_E_INVALID;
    }
}

bool amg_supports_batch8(const padne_csr *A0) {
    const Amg *amg = (const Amg *)A0->amg;
    return amg != nullptr && amg->f32 && !amg->dist && amg->levels.size() >= 2 && amg->coarse_inv32 != nullptr;
}

// buffers of the single-precision entry stage, for callers that fuse it into their own kernel (false: double cycle)
// number of per-workgroup r.z partials the last stage of the cycle writes (the grid of that launch)
int amg_rz_partials(const padne_csr *A0) {
    const Amg *amg = (const Amg *)A0->amg;
    if (amg != nullptr && !amg->dist && amg->levels.size() == 1 && amg->n_coarse > 0) return (amg->n_coarse + 3) / 4;      // dense_gemv_dot
    if (amg != nullptr && amg->f32 && !amg->levels.empty() && amg->levels[0].W != nullptr)
        return spmv_grid(amg->levels[0].W);
    return spmv_partials(A0);
}

bool amg_f32_entry_args(const padne_csr *A0, float *jac, const float **dinv32, float **b32, float **xa32) {
    const Amg *amg = (const Amg *)A0->amg;
    if (amg == nullptr || !amg->f32 || amg->levels.size() < 2) return false;
    const AmgLevel &L = amg->levels[0];
    *jac = (float)L.jac;
    *dinv32 = L.A->dinv32;
    *b32 = (float *)L.b;
    // (null: nothing reads the pre-smoothed iterate of the fine level -- residual and up-leg form it from b, amg_apply_f32)
    *xa32 = (!amg->dist && L.W != nullptr && spmv_resid_pre_ok(L.A)) ? nullptr : (float *)L.xa;
    return true;
}

int amg_apply(padne_ctx *ctx, const padne_csr *A0, const double *r, double *z, double *partials_rz,
              const int32_t *done_flag, const double *bb2, bool entry_done, float *z32) {
    Amg *amg = (Amg *)A0->amg;
    if (amg->f32) return amg_apply_f32(ctx, amg, r, z, partials_rz, done_flag, bb2, entry_done, z32);
    PADNE_REQUIRE(z32 == nullptr, "single-precision output of the double-precision cycle");
    hipStream_t s = ctx->stream;
    const int nl = (int)amg->levels.size();
    // downward sweep
    for (int l = 0; l < nl; ++l) {
        AmgLevel &L = amg->levels[l];
        const double *b = (l == 0) ? r : L.b;
        if (l == nl - 1) {
            double *out = (l == 0) ? z : L.xb;
            if (amg->dist) {
                // gather the restricted residual and run the rest of the cycle redundantly on every rank
                double *seg = amg->coarse_gather + (size_t)ctx->rank * amg->n_pad;
                PADNE_HIP_CHECK(hipMemcpyAsync(seg, b, sizeof(double) * (size_t)L.n, hipMemcpyDeviceToDevice, s));
                PADNE_TRY(comm_allgather_f64(ctx, seg, amg->coarse_gather, amg->n_pad));
                hipLaunchKernelGGL(compact_pieces<double>, dim3(nblk(amg->n_pad), ctx->world), dim3(256), 0, s, ctx->world,
                                   amg->n_pad, amg->seg_off, (const double *)amg->coarse_gather, amg->tail_r, done_flag);
                PADNE_HIP_CHECK(hipGetLastError());
                PADNE_TRY(amg_apply(ctx, amg->tail, amg->tail_r, amg->tail_z, nullptr, done_flag, nullptr, false));
                PADNE_HIP_CHECK(hipMemcpyAsync(out, amg->tail_z + amg->tail_off, sizeof(double) * (size_t)L.n,
                                               hipMemcpyDeviceToDevice, s));
            } else if (amg->n_coarse > 0 && l == 0 && partials_rz != nullptr)
                // a hierarchy of one level: the inverse IS the preconditioner (amg_rz_partials: one partial per four rows)
                hipLaunchKernelGGL(dense_gemv_dot, dim3((amg->n_coarse + 3) / 4), dim3(256), 0, s, amg->n_coarse,
                                   (const double *)amg->coarse_inv, b, out, partials_rz, done_flag);
            else if (amg->n_coarse > 0)
                hipLaunchKernelGGL(dense_gemv<double>, dim3((amg->n_coarse + 3) / 4), dim3(256), 0, s, amg->n_coarse,
                                   amg->n_coarse, (const double *)amg->coarse_inv, b, out);
            PADNE_HIP_CHECK(hipGetLastError());
            break;
        }
        const int gv = (int)std::min<long long>((L.n + 255) / 256, 1024);
        hipLaunchKernelGGL(scale_dinv_kernel<double>, dim3(gv > 0 ? gv : 1), dim3(256), 0, s, L.n, L.jac,
                           (const double *)L.A->dinv, b, L.xa, done_flag);
        PADNE_HIP_CHECK(hipGetLastError());
        if (amg->dist) PADNE_TRY(halo_exchange_plan(ctx, L.halo, L.xa, done_flag));
        PADNE_TRY(launch_spmv_mode(ctx, L.A, SPMV_RESID, L.xa, L.tmp, nullptr, nullptr, done_flag, b, nullptr, 0.0));
        PADNE_TRY(launch_spmv_mode(ctx, L.R, SPMV_PLAIN, L.tmp, amg->levels[l + 1].b, nullptr, nullptr, done_flag,
                                   nullptr, nullptr, 0.0));
    }
    // upward sweep
    for (int l = nl - 2; l >= 0; --l) {
        AmgLevel &L = amg->levels[l];
        const double *b = (l == 0) ? r : L.b;
        const double *xc = amg->levels[l + 1].xb;
        double *out = (l == 0) ? z : L.xb;
        PADNE_TRY(launch_spmv_mode(ctx, L.P, SPMV_ADD, xc, L.xa, nullptr, nullptr, done_flag, nullptr, nullptr, 0.0));
        if (amg->dist) PADNE_TRY(halo_exchange_plan(ctx, L.halo, L.xa, done_flag));
        PADNE_TRY(launch_spmv_mode(ctx, L.A, SPMV_JACOBI, L.xa, out, nullptr, (l == 0) ? partials_rz : nullptr,
                                   done_flag, b, L.A->dinv, L.jac));
    }
    if (nl == 1 && partials_rz != nullptr && (amg->dist || amg->n_coarse <= 0)) {
        set_error("a hierarchy of one level serves as a preconditioner only through its dense inverse");
        return PADNE_E_INVALID;
    }
    return PADNE_OK;
}

const padne_csr *amg_level_matrix(const padne_csr *A0, int level, int which) {
    const Amg *amg = (const Amg *)A0->amg;
    if (!amg || level < 0 || level >= (int)amg->levels.size()) return nullptr;
    const AmgLevel &L = amg->levels[(size_t)level];
    return which == 0 ? L.A : (which == 1 ? L.P : L.R);
}

int csr_matmul(padne_ctx *ctx, const padne_csr *X, const padne_csr *Y, padne_csr **C) { return spgemm(ctx, X, Y, C); }
const padne_csr *amg_level_matrix(const padne_csr *A0, int level, int which);
int csr_transpose(padne_ctx *ctx, const padne_csr *M, padne_csr **T) { return transpose(ctx, M, T); }

void amg_info(const padne_csr *A0, int *levels, double *complexity, double *setup_seconds, long long *coarse_n) {
    const Amg *amg = (const Amg *)A0->amg;
    if (!amg) return;
    if (levels) {
        *levels = (int)amg->levels.size();
        if (amg->tail && amg->tail->amg) *levels += (int)((const Amg *)amg->tail->amg)->levels.size() - 1;
    }
    if (complexity) *complexity = amg->operator_complexity;
    if (setup_seconds) *setup_seconds = amg->setup_seconds;
    if (coarse_n) *coarse_n = amg->n_coarse;
}

}  // namespace padne

extern "C" int padne_csr_split_tiles(const padne_csr *m, int level, int64_t *interior, int64_t *boundary) {
    PADNE_REQUIRE(m && interior && boundary, "null argument");
    const padne_csr *a = level < 0 ? m : padne::amg_level_matrix(m, level, 0);
    *interior = *boundary = 0;
    if (a != nullptr && a->split_state == 1) {
        *interior = a->split_n_int;
        *boundary = a->split_n_bnd;
    }
    return PADNE_OK;
}
