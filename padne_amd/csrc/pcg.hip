// Preconditioned conjugate gradients on the device, replacing the direct solve
// scipy.sparse.linalg.spsolve of the reference (solver.py:773) on the reduced SPD system.
// Preconditioners: Jacobi (below) or one smoothed-aggregation multigrid V-cycle (amg.hip; the default).
//
// One Jacobi-PCG iteration = three kernels, all HBM-streaming, no host synchronisation:
//   K1  q = A p            + partial sums of p.q                (spmv.hip, 104*N bytes)
//   K2  alpha = rz/pq ; x += alpha p ; r -= alpha q             (56*N bytes)
//       + partial sums of r.(D^-1 r) and r.r
//   K3  beta = rz'/rz ; p = D^-1 r + beta p                      (32*N bytes)
//       + workgroup 0 does the bookkeeping: iteration count, convergence / breakdown flags.
// z = D^-1 r is never stored.  With multigrid K2 stops at r.r, the cycle produces z and the r.z partials
// (its last stage), and K3 reads z.  Scalars never visit the host: every workgroup of the consuming
// kernel re-adds the producer's per-workgroup partials (<= 2048 doubles, L2 resident) in a
// fixed order, so results are bitwise reproducible run to run (no float atomics).
// In multi-GPU runs a one-workgroup kernel folds the partials into a scalar which is then
// summed over ranks by RCCL on the same stream; consumers then read a single "partial".
//
// The host enqueues `check_every` iterations at a time and then polls one status word; kernels
// launched after convergence return immediately on the `done` flag.  After convergence the true
// residual is re-evaluated and the iteration restarted from it if the recurrence drifted.
// solve_batch8 runs the same recurrences for 8 right-hand sides in lockstep (spmm.hip).
#include "common.hpp"

#include <algorithm>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace padne {

__device__ __forceinline__ double wave_sum_v(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// every thread of a 256-thread workgroup gets the sum of partials[0..P) (fixed order)
__device__ __forceinline__ double block_total(const double *__restrict__ partials, int P, double *red) {
    double s = 0.0;
    for (int i = threadIdx.x; i < P; i += 256) s += partials[i];
    s = wave_sum_v(s);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) red[w] = s;
    __syncthreads();
    const double t = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return t;
}

// two such sums in ONE pass over the workgroup (the loads of both arrays in flight together, one pair of barriers): every
// sum adds the same terms in the same order as block_total, hence the same bits.  `red2` = 8 doubles of LDS.
__device__ __forceinline__ void block_total2(const double *__restrict__ pa, const int Pa, const double *__restrict__ pb, const int Pb,
                                             double *red2, double *ta, double *tb) {
    double sa = 0.0, sb = 0.0;
    const int P = Pa > Pb ? Pa : Pb;
    for (int i = threadIdx.x; i < P; i += 256) {
        if (i < Pa) sa += pa[i];
        if (i < Pb) sb += pb[i];
    }
    sa = wave_sum_v(sa);
    sb = wave_sum_v(sb);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) {
        red2[w] = sa;
        red2[4 + w] = sb;
    }
    __syncthreads();
    *ta = (red2[0] + red2[1]) + (red2[2] + red2[3]);
    *tb = (red2[4] + red2[5]) + (red2[6] + red2[7]);
    __syncthreads();
}

__device__ __forceinline__ void block_store_partial(double v, double *red, double *out) {
    v = wave_sum_v(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) red[w] = v;
    __syncthreads();
    if (threadIdx.x == 0) *out = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
}

struct PcgStatus {       // lives in device memory, mirrored to pinned host memory when polled
    int32_t done;        // 1 = stop iterating
    int32_t code;        // PADNE_OK / PADNE_E_BREAKDOWN
    int32_t iters;       // iterations completed
    int32_t done_seen;   // `done` as the x/r update of the current iteration read it.  The p update of the multigrid loop
                         // (which sets `done` itself, in its workgroup 0) takes its early exit from THIS word: a workgroup of
                         // that launch dispatched after workgroup 0's store must still apply its share of x += alpha p
    double rr;           // recurrence ||r||^2 after the last completed iteration
    double tol2;         // stop when rr <= tol2
    double bb;           // ||b||^2
};

// r = b - ax (ax may be null: x0 = 0) ; p = dinv*r ; partial sums rz, rr, bb
__global__ __launch_bounds__(256) void pcg_init_kernel(
    const long long n, const double *__restrict__ b, const double *__restrict__ ax,
    const double *__restrict__ dinv, double *__restrict__ r, double *__restrict__ p,
    double *__restrict__ part_rz, double *__restrict__ part_rr, double *__restrict__ part_bb) {
    __shared__ double red[4];
    double rz = 0.0, rr = 0.0, bb = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const double bi = b[i];
        const double ri = ax ? bi - ax[i] : bi;
        const double zi = dinv[i] * ri;
        r[i] = ri;
        p[i] = zi;
        rz += ri * zi;
        rr += ri * ri;
        bb += bi * bi;
    }
    block_store_partial(rz, red, part_rz + blockIdx.x);
    block_store_partial(rr, red, part_rr + blockIdx.x);
    block_store_partial(bb, red, part_bb + blockIdx.x);
}

// folds partials into scalars (used once after init, and per reduction in multi-GPU mode)
__global__ __launch_bounds__(256) void fold_partials_kernel(const double *__restrict__ partials, int P,
                                                            int stride, int count, double *__restrict__ out) {
    __shared__ double red[4];
    for (int c = 0; c < count; ++c) {
        const double t = block_total(partials + (size_t)c * stride, P, red);
        if (threadIdx.x == 0) out[c] = t;
    }
}

__global__ void pcg_set_tolerance_kernel(PcgStatus *st, const double *__restrict__ scal_rr_bb, double rtol,
                                         double atol, int use_existing_bb) {
    // scal_rr_bb[0] = rr, [1] = bb
    const double bb = use_existing_bb ? st->bb : scal_rr_bb[1];
    const double rr = scal_rr_bb[0];
    double tol = rtol * sqrt(bb);
    if (atol > tol) tol = atol;
    st->bb = bb;
    st->tol2 = tol * tol;
    st->rr = rr;
    st->code = PADNE_OK;
    st->done = (rr <= tol * tol) ? 1 : 0;
    if (!(rr == rr)) {  // NaN in the input
        st->done = 1;
        st->code = PADNE_E_BREAKDOWN;
    }
}

// K2
__global__ __launch_bounds__(256) void pcg_update_xr_kernel(
    const long long n, const double *__restrict__ part_rz, const int P_rz,
    const double *__restrict__ part_pq, const int P_pq, const double *__restrict__ p,
    const double *__restrict__ q, const double *__restrict__ dinv, double *__restrict__ x,
    double *__restrict__ r, double *__restrict__ part_rz_new, double *__restrict__ part_rr,
    PcgStatus *__restrict__ st) {
    __shared__ double red[4];
    if (st->done) return;
    const double rz = block_total(part_rz, P_rz, red);
    const double pq = block_total(part_pq, P_pq, red);
    const double alpha = rz / pq;
    double s_rz = 0.0, s_rr = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const double pi = p[i];
        const double ri = r[i] - alpha * q[i];
        if (x != nullptr) x[i] += alpha * pi;      // (null: the Lanczos steps of the multigrid setup have no iterate)
        r[i] = ri;
        const double zi = dinv[i] * ri;
        s_rz += ri * zi;
        s_rr += ri * ri;
    }
    block_store_partial(s_rz, red, part_rz_new + blockIdx.x);
    block_store_partial(s_rr, red, part_rr + blockIdx.x);
}

// K3
__global__ __launch_bounds__(256) void pcg_update_p_kernel(
    const long long n, const double *__restrict__ part_rz_new, const double *__restrict__ part_rz_old,
    const int P_rz, const double *__restrict__ part_rr, const int P_rr,
    const double *__restrict__ part_pq, const int P_pq, const double *__restrict__ r,
    const double *__restrict__ dinv, double *__restrict__ p, PcgStatus *__restrict__ st,
    const int max_iter, double *__restrict__ rec_next = nullptr, double *__restrict__ rec_pq = nullptr) {
    // rec_next / rec_pq (the Lanczos estimates of the multigrid setup): the step's scalars are left for the host --
    // r.z and r.r after the step, p.q of the step -- instead of two one-workgroup fold launches per step
    __shared__ double red[4];
    if (st->done) return;
    const double rz_new = block_total(part_rz_new, P_rz, red);
    const double rz_old = block_total(part_rz_old, P_rz, red);
    const double beta = rz_new / rz_old;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        p[i] = dinv[i] * r[i] + beta * p[i];
    }
    if (blockIdx.x == 0) {
        const double rr = block_total(part_rr, P_rr, red);
        const double pq = block_total(part_pq, P_pq, red);
        if (threadIdx.x == 0 && rec_next != nullptr) {
            rec_next[0] = rz_new;
            rec_next[1] = rr;
            rec_pq[0] = pq;
        }
        // all other workgroups have already passed (or will pass) their own `done` read with the
        // value 0 only if they were dispatched before this store lands; either way x and r are
        // complete (K2), and p is dead once `done` is set.
        __syncthreads();
        if (threadIdx.x == 0) {
            const int it = st->iters + 1;
            st->iters = it;
            st->rr = rr;
            if (!(pq > 0.0) || !(rr == rr)) {
                st->code = PADNE_E_BREAKDOWN;
                st->done = 1;
            } else if (rr <= st->tol2 || it >= max_iter) {
                st->done = 1;
            }
        }
    }
}

// r = b - ax ; partial rr   (true-residual check)
__global__ __launch_bounds__(256) void residual_kernel(const long long n, const double *__restrict__ b,
                                                       const double *__restrict__ ax, double *__restrict__ r,
                                                       double *__restrict__ part_rr) {
    __shared__ double red[4];
    double rr = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const double ri = ax[i] - b[i];
        if (r) r[i] = -ri;
        rr += ri * ri;
    }
    block_store_partial(rr, red, part_rr + blockIdx.x);
}

// ---- kernels of the multigrid-preconditioned loop (z comes from amg_apply) --------------------------
// alpha = rz/pq ; x += alpha p ; r -= alpha q ; partial r.r
__global__ __launch_bounds__(256) void pcg_update_xr_plain_kernel(
    const long long n, const double *__restrict__ part_rz, const int P_rz, const double *__restrict__ part_pq,
    const int P_pq, const double *__restrict__ p, const double *__restrict__ q, double *__restrict__ x,
    double *__restrict__ r, double *__restrict__ part_rr, PcgStatus *__restrict__ st) {
    __shared__ double red[4];
    const int stop = st->done;              // written by an EARLIER launch: every workgroup of this one reads the same value
    if (blockIdx.x == 0 && threadIdx.x == 0) st->done_seen = stop;
    if (stop) return;
    const double rz = block_total(part_rz, P_rz, red);
    const double pq = block_total(part_pq, P_pq, red);
    const double alpha = rz / pq;
    double s_rr = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const double ri = r[i] - alpha * q[i];
        x[i] += alpha * p[i];
        r[i] = ri;
        s_rr += ri * ri;
    }
    block_store_partial(s_rr, red, part_rr + blockIdx.x);
}

// the same update with the entry stage of the single-precision multigrid cycle fused in: while r is in registers,
// b32 = r / ||b|| and the first damped-Jacobi sweep xa32 = c * dinv * b32 are written as well (amg.hip, amg_entry_f32)
__global__ __launch_bounds__(256) void pcg_update_xr_entry_kernel(
    const long long n, const double *__restrict__ part_rz, const int P_rz, const double *__restrict__ part_pq,
    const int P_pq, const double *__restrict__ p, const double *__restrict__ q, double *__restrict__ x,
    double *__restrict__ r, double *__restrict__ part_rr, PcgStatus *__restrict__ st, const double *__restrict__ bb2,
    const float c, const float *__restrict__ dinv32, float *__restrict__ b32, float *__restrict__ xa32, const int p_hat) {
    __shared__ double red[8];
    const int stop = st->done;              // written by an EARLIER launch: every workgroup of this one reads the same value
    if (blockIdx.x == 0 && threadIdx.x == 0) st->done_seen = stop;
    if (stop) return;
    double rz, pq;
    block_total2(part_rz, P_rz, part_pq, P_pq, red, &rz, &pq);
    const double s2 = *bb2;
    const double s_inv = s2 > 0.0 ? 1.0 / sqrt(s2) : 1.0;
    // p_hat: the search direction is stored as p / ||b|| in single precision (solve_one), q and p.q are those of the stored
    // vector: the step along it is alpha ||b||
    const double alpha = p_hat ? rz / (pq * (s2 > 0.0 ? sqrt(s2) : 1.0)) : rz / pq;
    double s_rr = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const double ri = r[i] - alpha * q[i];
        if (x != nullptr) x[i] += alpha * p[i];      // null: pcg_update_p_z_kernel does it while it has p in registers
        r[i] = ri;
        s_rr += ri * ri;
        const float v = (float)(ri * s_inv);
        b32[i] = v;
        if (xa32 != nullptr) xa32[i] = c * dinv32[i] * v;      // (null: the cycle forms its first sweep from b32 itself)
    }
    block_store_partial(s_rr, red, part_rr + blockIdx.x);
}

// p^ = z / ||b|| in single precision (the first search direction of a (re)start)
__global__ void p_hat_from_z_kernel(const long long n, const double *__restrict__ z, const double *__restrict__ bb2, float *__restrict__ p32) {
    const double s2 = *bb2;
    const double s_inv = s2 > 0.0 ? 1.0 / sqrt(s2) : 1.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        p32[i] = (float)(z[i] * s_inv);
}

// beta = rz'/rz ; p = z + beta p ; bookkeeping
__global__ __launch_bounds__(256) void pcg_update_p_z_kernel(
    const long long n, const double *__restrict__ part_rz_new, const double *__restrict__ part_rz_old,
    const int P_rz, const double *__restrict__ part_rr, const int P_rr, const double *__restrict__ part_pq,
    const int P_pq, const double *__restrict__ z, double *__restrict__ p, PcgStatus *__restrict__ st,
    const int max_iter, double *__restrict__ x_deferred, const float *__restrict__ z32, const double *__restrict__ bb2,
    float *__restrict__ p32, float *__restrict__ p32_next = nullptr, double *__restrict__ alpha_out = nullptr) {
    // p32_next / alpha_out (the search directions of a solve KEPT, solve_one): the new direction goes to a place of its own
    // and the step length along the old one is left for pcg_x_flush_kernel -- x is not touched here (x_deferred is null)
    __shared__ double red[8];
    // NOT st->done: workgroup 0 of this very launch sets it, and this kernel carries the deferred x += alpha p -- a
    // workgroup dispatched after that store would skip its slice of the last update.  done_seen is what the x/r update of
    // this iteration read, i.e. a value from before this launch
    if (st->done_seen) return;
    double rz_new, rz_old;
    block_total2(part_rz_new, P_rz, part_rz_old, P_rz, red, &rz_new, &rz_old);
    const double beta = rz_new / rz_old;
    if (z32 != nullptr) {
        // the cycle left z in single precision and without its factor ||b|| (amg_apply, z32): the same double the exit stage
        // took its r.z from; with the deferred x update
        const double s2 = *bb2;
        const double z_mul = s2 > 0.0 ? sqrt(s2) : 1.0;
        // (the step length: every workgroup's where x rides on this update, the first workgroup's alone where the search
        // directions are kept and it is only recorded)
        const bool need_alpha = p32_next == nullptr || blockIdx.x == 0;
        const double alpha = need_alpha ? rz_old / block_total(part_pq, P_pq, red) : 0.0;
        if (p32 != nullptr) {
            // the search direction is KEPT in single precision, in the units of the cycle (p^ = p / ||b||, like z32: right-hand
            // sides of 1e-30 A or 1e+30 A stay in range): q = A p^ was formed from this very float, so x += alpha^ p^ and
            // r -= alpha^ q (alpha^ = alpha ||b||) stay consistent to double rounding; what the rounding of p costs is conjugacy
            // at the 1e-7 level, which the cycle's own single precision costs already
            const double alpha_hat = alpha / z_mul;
            if (p32_next != nullptr) {
                if (blockIdx.x == 0 && threadIdx.x == 0) *alpha_out = alpha_hat;
                for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
                    p32_next[i] = (float)((double)z32[i] + beta * (double)p32[i]);
            } else {
                for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
                    const double pi = (double)p32[i];
                    x_deferred[i] += alpha_hat * pi;
                    p32[i] = (float)((double)z32[i] + beta * pi);
                }
            }
        } else {
            for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
                const double pi = p[i];
                x_deferred[i] += alpha * pi;
                p[i] = (double)z32[i] * z_mul + beta * pi;
            }
        }
    } else if (x_deferred != nullptr) {
        // x += alpha p of this iteration (the alpha pcg_update_xr_entry_kernel applied to r), with the p that is about to be
        // replaced: p is read once per iteration instead of twice
        const double alpha = rz_old / block_total(part_pq, P_pq, red);
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
            const double pi = p[i];
            x_deferred[i] += alpha * pi;
            p[i] = z[i] + beta * pi;
        }
    } else {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
            p[i] = z[i] + beta * p[i];
    }
    if (blockIdx.x == 0) {
        const double rr = block_total(part_rr, P_rr, red);
        const double pq = block_total(part_pq, P_pq, red);
        __syncthreads();
        if (threadIdx.x == 0) {
            const int it = st->iters + 1;
            st->iters = it;
            st->rr = rr;
            if (!(pq > 0.0) || !(rr == rr) || !(rz_new > 0.0)) {
                st->code = PADNE_E_BREAKDOWN;
                st->done = 1;
            } else if (rr <= st->tol2 || it >= max_iter) {
                st->done = 1;
            }
        }
    }
}

// x += sum over the iterations j in [j0, min(j1, completed)) of alpha_j p_j, each row's terms added in the order of the
// iterations: the bits of the running update x += alpha_j p_j, which read and wrote x in every iteration (16 bytes per
// row and iteration; this reads 4).  p_j lies in slot j % n_slots of `hist`, alpha_j in alpha[j % n_slots]; `completed`
// is the status word's count (iterations queued behind the one that converged returned at their first kernel).
__global__ __launch_bounds__(256) void pcg_x_flush_kernel(const long long n, const float *__restrict__ hist, const size_t slot_stride,
                                                          const int n_slots, const double *__restrict__ alpha, const int j0,
                                                          const int j1, const PcgStatus *__restrict__ st, double *__restrict__ x) {
    __shared__ double a_s[64];
    __shared__ int s_s[64];
    const int hi = min(j1, st->iters);
    const int cnt = hi - j0;
    if (cnt <= 0) return;
    if ((int)threadIdx.x < cnt && threadIdx.x < 64) {
        a_s[threadIdx.x] = alpha[(j0 + threadIdx.x) % n_slots];
        s_s[threadIdx.x] = (j0 + threadIdx.x) % n_slots;
    }
    __syncthreads();
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        double xi = x[i];
        int k = 0;
        for (; k + 4 <= cnt; k += 4) {      // four loads in flight, the additions in order
            const float p0 = hist[(size_t)s_s[k] * slot_stride + i], p1 = hist[(size_t)s_s[k + 1] * slot_stride + i],
                        p2 = hist[(size_t)s_s[k + 2] * slot_stride + i], p3 = hist[(size_t)s_s[k + 3] * slot_stride + i];
            xi += a_s[k] * (double)p0;
            xi += a_s[k + 1] * (double)p1;
            xi += a_s[k + 2] * (double)p2;
            xi += a_s[k + 3] * (double)p3;
        }
        for (; k < cnt; ++k) xi += a_s[k] * (double)hist[(size_t)s_s[k] * slot_stride + i];
        x[i] = xi;
    }
}

// r = b - ax (ax may be null) ; partial rr, bb
__global__ __launch_bounds__(256) void pcg_init_plain_kernel(const long long n, const double *__restrict__ b,
                                                             const double *__restrict__ ax, double *__restrict__ r,
                                                             double *__restrict__ part_rr,
                                                             double *__restrict__ part_bb) {
    __shared__ double red[4];
    double rr = 0.0, bb = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const double bi = b[i];
        const double ri = ax ? bi - ax[i] : bi;
        r[i] = ri;
        rr += ri * ri;
        bb += bi * bi;
    }
    block_store_partial(rr, red, part_rr + blockIdx.x);
    block_store_partial(bb, red, part_bb + blockIdx.x);
}

static int vec_grid(long long n) {
    long long g = (n + 255) / 256;
    if (g > 1024) g = 1024;  // 4 workgroups per CU, grid-stride the rest
    if (g < 1) g = 1;
    return (int)g;
}

// partial slots inside ctx->partials
enum { SLOT_PQ = 0, SLOT_RZ0 = 1, SLOT_RZ1 = 2, SLOT_RR = 3, SLOT_BB = 4, SLOT_TMP = 5 };
static inline double *slot(padne_ctx *ctx, int s) { return ctx->partials + (size_t)s * kMaxPartials; }

// gathers the owned values other ranks need into this rank's segment of the exchange buffer that
// sits behind the owned entries:  v[n_owned + rank*M + k] = v[export_idx[k]]
template <typename T>
__global__ void halo_pack_kernel(T *__restrict__ v, const int *__restrict__ export_idx, int n_export,
                                 long long dst_offset, const int *__restrict__ done_flag) {
    if (done_flag != nullptr && *done_flag != 0) return;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_export) v[dst_offset + k] = v[export_idx[k]];
}

// scalar slots inside ctx->scalars used when the reductions go through RCCL
enum { S_RR = 0, S_BB = 1, S_TRUE = 2,
       S_NRM = 3,      // what the single-precision cycle and the stored search direction are normalised with: ||r_0||^2 of a warm start
       S_PQ = 8, S_RZRR0 = 10, S_RZRR1 = 12 };

// peer-to-peer form of the pack: the exported values go straight into every rank's mailbox entry of this exchange
// (peers[q] + entry_off, laid out [world][m_cap] 8-byte cells; narrower types use the front of their cell)
// seq1 != 0: the mailboxes are shared between PROCESSES (comm.hip, padne_ctx_p2p_export): when all stores of this launch are
// out, the sender writes seq1 -- the exchange's sequence number + 1 -- into its arrival flag in every rank's mailbox.  Every
// thread fences its own stores at system scope; the workgroup that counts itself last (an agent-scope counter in the
// sender's own mailbox header) therefore signals after ALL stores of the launch, with system-scope release stores.  A
// launch with nothing to export (grid of one workgroup, n_export = 0) signals just the same: the receivers wait for
// every rank's flag.
template <typename T>
__global__ void halo_store_peers_kernel(const T *__restrict__ v, const int *__restrict__ export_idx, int n_export, void *const *peers,
                                        size_t entry_off, int world, int rank, int m_cap, const int *__restrict__ done_flag,
                                        unsigned long long seq1) {
    if (done_flag != nullptr && *done_flag != 0) return;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_export) {
        const T val = v[export_idx[k]];
        for (int q = 0; q < world; ++q) {
            T *cell = reinterpret_cast<T *>(static_cast<char *>(peers[q]) + entry_off + ((size_t)rank * m_cap + k) * 8);
            if (seq1 != 0ull)
                __hip_atomic_store(cell, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            else
                *cell = val;
        }
    }
    if (seq1 == 0ull) return;
    __threadfence_system();
    __syncthreads();
    __shared__ int s_last;
    if (threadIdx.x == 0) {
        unsigned int *counter = reinterpret_cast<unsigned int *>(static_cast<char *>(peers[rank]) + kP2pCounterOff);
        const unsigned int prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_last = prev + 1u == gridDim.x ? 1 : 0;
        if (s_last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (for the next exchange: stream order)
    }
    __syncthreads();
    if (s_last && (int)threadIdx.x < world) {
        unsigned long long *flag = reinterpret_cast<unsigned long long *>(static_cast<char *>(peers[threadIdx.x])) + rank;
        __hip_atomic_store(flag, seq1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// the receiver's half: mailbox entry -> the exchange area behind the owned values, v[n_owned + r * m + k].  seq1 != 0
// (mailboxes shared between processes): first wait until every sender's flag has reached seq1 -- a flag only grows, so a
// sender that is already an exchange ahead passes too.  The wait is bounded by the wall clock (timeout_ticks of the
// 100 MHz constant clock); a wait that runs out leaves (sender + 1) << 48 | seq1 in the header's error word, which fails
// the solve (comm_p2p_check), and the kernel goes on so that the stream drains.
template <typename T>
__global__ void halo_unpack_kernel(T *__restrict__ v, long long n_owned, int m, int world, const void *mbox, size_t entry_off,
                                   int m_cap, const int *__restrict__ done_flag, unsigned long long seq1,
                                   unsigned long long timeout_ticks) {
    if (done_flag != nullptr && *done_flag != 0) return;
    if (seq1 != 0ull) {
        if ((int)threadIdx.x < world) {
            const unsigned long long *flag = static_cast<const unsigned long long *>(mbox) + threadIdx.x;
            const unsigned long long t0 = wall_clock64();
            while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq1) {
                if (wall_clock64() - t0 > timeout_ticks) {
                    unsigned long long *err = reinterpret_cast<unsigned long long *>(const_cast<char *>(static_cast<const char *>(mbox)) + kP2pErrorOff);
                    __hip_atomic_store(err, ((unsigned long long)(threadIdx.x + 1) << 48) | (seq1 & 0xffffffffffffull), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
                __builtin_amdgcn_s_sleep(32);
            }
        }
        __syncthreads();
    }
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= world * m) return;
    const int r = j / m, k = j - r * m;
    const T *cell = reinterpret_cast<const T *>(static_cast<const char *>(mbox) + entry_off + ((size_t)r * m_cap + k) * 8);
    v[n_owned + j] = seq1 != 0ull ? __hip_atomic_load(cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : *cell;
}

// An exchange in two halves, so that the caller can put work that needs no remote value between them (the interior
// tiles of the product the exchange is for): halo_send packs and sends, halo_recv returns when v's exchange area is
// complete (stream order).  Peer-to-peer stores where the context can do them (comm_p2p_enabled), otherwise pack into
// the rank's own segment + one all-gather.
template <typename T>
static int halo_send_t(padne_ctx *ctx, const HaloPlan &plan, T *v, const int32_t *done_flag, HaloTicket *tk) {
    tk->p2p = false;
    if (plan.m <= 0) return PADNE_OK;
    if (comm_p2p_enabled(ctx) && comm_p2p_fits(ctx, plan.m)) {
        void **peers = nullptr;
        PADNE_TRY(comm_p2p_begin(ctx, plan.m, &peers, &tk->entry_off));
        tk->p2p = true;
        tk->seq1 = ctx->p2p_ipc ? ctx->p2p_seq : 0ull;      // (p2p_seq was advanced by comm_p2p_begin: this exchange's number + 1)
        if (plan.n_export > 0 || ctx->p2p_ipc) {             // between processes a rank with nothing to export still signals
            hipLaunchKernelGGL(halo_store_peers_kernel<T>, dim3(plan.n_export > 0 ? (plan.n_export + 255) / 256 : 1), dim3(256), 0,
                               ctx->stream, (const T *)v, plan.export_idx, plan.n_export, (void *const *)peers, tk->entry_off,
                               ctx->world, ctx->rank, ctx->p2p_m_cap, done_flag, tk->seq1);
            PADNE_HIP_CHECK(hipGetLastError());
        }
        return PADNE_OK;
    }
    const long long seg_off = plan.n_owned + (long long)ctx->rank * plan.m;
    if (plan.n_export > 0) {
        hipLaunchKernelGGL(halo_pack_kernel<T>, dim3((plan.n_export + 255) / 256), dim3(256), 0, ctx->stream, v,
                           plan.export_idx, plan.n_export, seg_off, done_flag);
        PADNE_HIP_CHECK(hipGetLastError());
    }
    return PADNE_OK;      // (the all-gather itself is halo_recv's: same stream, nothing runs beside it)
}

static int allgather_t(padne_ctx *ctx, const double *send, double *recv, int count) { return comm_allgather_f64(ctx, send, recv, count); }
static int allgather_t(padne_ctx *ctx, const float *send, float *recv, int count) { return comm_allgather_f32(ctx, send, recv, count); }

template <typename T>
static int halo_recv_t(padne_ctx *ctx, const HaloPlan &plan, T *v, const int32_t *done_flag, const HaloTicket &tk) {
    if (plan.m <= 0) return PADNE_OK;
    if (tk.p2p) {
        PADNE_TRY(comm_p2p_arrive(ctx));
        const int n = ctx->world * plan.m;
        hipLaunchKernelGGL(halo_unpack_kernel<T>, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, v, plan.n_owned, plan.m,
                           ctx->world, (const void *)ctx->p2p_mbox, tk.entry_off, ctx->p2p_m_cap, done_flag, tk.seq1,
                           (unsigned long long)ctx->p2p_timeout_ms * 100000ull);
        PADNE_HIP_CHECK(hipGetLastError());
        return PADNE_OK;
    }
    const long long seg_off = plan.n_owned + (long long)ctx->rank * plan.m;
    return allgather_t(ctx, v + seg_off, v + plan.n_owned, plan.m);
}

int halo_send(padne_ctx *ctx, const HaloPlan &plan, double *v, const int32_t *done_flag, HaloTicket *tk) { return halo_send_t(ctx, plan, v, done_flag, tk); }
int halo_send_f32(padne_ctx *ctx, const HaloPlan &plan, float *v, const int32_t *done_flag, HaloTicket *tk) { return halo_send_t(ctx, plan, v, done_flag, tk); }
int halo_recv(padne_ctx *ctx, const HaloPlan &plan, double *v, const int32_t *done_flag, const HaloTicket &tk) { return halo_recv_t(ctx, plan, v, done_flag, tk); }
int halo_recv_f32(padne_ctx *ctx, const HaloPlan &plan, float *v, const int32_t *done_flag, const HaloTicket &tk) { return halo_recv_t(ctx, plan, v, done_flag, tk); }

int halo_exchange_plan(padne_ctx *ctx, const HaloPlan &plan, double *v, const int32_t *done_flag) {
    HaloTicket tk;
    PADNE_TRY(halo_send(ctx, plan, v, done_flag, &tk));
    return halo_recv(ctx, plan, v, done_flag, tk);
}

int halo_exchange_plan_f32(padne_ctx *ctx, const HaloPlan &plan, float *v, const int32_t *done_flag) {
    HaloTicket tk;
    PADNE_TRY(halo_send_f32(ctx, plan, v, done_flag, &tk));
    return halo_recv_f32(ctx, plan, v, done_flag, tk);
}

static int halo_exchange(padne_ctx *ctx, double *v, const int32_t *done_flag) {
    if (!ctx->halo_on) return PADNE_OK;
    HaloPlan plan;
    plan.n_owned = ctx->halo_n_owned;
    plan.m = ctx->halo_m;
    plan.n_export = ctx->halo_n_export;
    plan.export_idx = ctx->halo_export;
    return halo_exchange_plan(ctx, plan, v, done_flag);
}

}  // namespace padne
// (test header) average device time of one halo exchange of the context's plan -- pack / store to the peers / wait / unpack,
// or pack / all-gather -- over `repeats` exchanges queued back to back between two events.  Collective: every rank calls it
// with the same count.
extern "C" int padne_ctx_halo_exchange_time(padne_ctx *ctx, int32_t repeats, double *seconds_out) {
    using namespace padne;
    PADNE_REQUIRE(ctx && seconds_out && repeats > 0, "argument");
    PADNE_REQUIRE(ctx->halo_on, "no halo plan on this context");
    PADNE_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t n = (size_t)ctx->halo_n_owned + (size_t)ctx->world * (size_t)ctx->halo_m;
    double *v = (double *)pool_alloc(ctx, sizeof(double) * (n ? n : 1));
    if (v == nullptr) return PADNE_E_NOMEM;
    int rc = PADNE_OK;
    hipError_t e = hipMemsetAsync(v, 0, sizeof(double) * n, ctx->stream);
    for (int k = 0; k < 3 && rc == PADNE_OK && e == hipSuccess; ++k) rc = halo_exchange(ctx, v, nullptr);
    if (e == hipSuccess) e = hipEventRecord(ctx->ev0, ctx->stream);
    for (int k = 0; k < repeats && rc == PADNE_OK && e == hipSuccess; ++k) rc = halo_exchange(ctx, v, nullptr);
    if (e == hipSuccess) e = hipEventRecord(ctx->ev1, ctx->stream);
    if (e == hipSuccess) e = hipEventSynchronize(ctx->ev1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    pool_free(ctx, v);
    if (e != hipSuccess) {
        set_error("halo exchange timing failed: %s", hipGetErrorString(e));
        return PADNE_E_HIP;
    }
    PADNE_TRY(rc);
    PADNE_TRY(comm_p2p_check(ctx));
    *seconds_out = (double)ms * 1e-3 / repeats;
    return PADNE_OK;
}
namespace padne {

// q = A v for a vector whose exchange area has to be refreshed first: the halo goes out, the interior tiles of the product
// (no remote value in their columns) run while it travels, the boundary tiles follow when it has landed
static int halo_product_dot(padne_ctx *ctx, const padne_csr *a, double *v, double *q, double *partials, const int32_t *done_flag) {
    if (!ctx->halo_on) return launch_spmv(ctx, a, v, q, v, partials, done_flag);
    HaloPlan plan;
    plan.n_owned = ctx->halo_n_owned;
    plan.m = ctx->halo_m;
    plan.n_export = ctx->halo_n_export;
    plan.export_idx = ctx->halo_export;
    HaloTicket tk;
    PADNE_TRY(halo_send(ctx, plan, v, done_flag, &tk));
    PADNE_TRY(launch_spmv_part(ctx, a, SPMV_DOT, SPMV_INTERIOR, v, q, v, partials, done_flag, nullptr, nullptr, 0.0));
    PADNE_TRY(halo_recv(ctx, plan, v, done_flag, tk));
    return launch_spmv_part(ctx, a, SPMV_DOT, SPMV_BOUNDARY, v, q, v, partials, done_flag, nullptr, nullptr, 0.0);
}

int amg_setup(padne_ctx *ctx, padne_csr *A0);
int amg_apply(padne_ctx *ctx, const padne_csr *A0, const double *r, double *z, double *partials_rz,
              const int32_t *done_flag, const double *bb2 = nullptr, bool entry_done = false, float *z32 = nullptr);
bool amg_f32_entry_args(const padne_csr *A0, float *jac, const float **dinv32, float **b32, float **xa32);
int amg_rz_partials(const padne_csr *A0);
void amg_info(const padne_csr *A0, int *levels, double *complexity, double *setup_seconds, long long *coarse_n);
const padne_csr *amg_level_matrix(const padne_csr *A0, int level, int which);

// One preconditioned CG solve.  Preconditioner: Jacobi (prec == nullptr) or one multigrid V-cycle of the
// hierarchy cached on `prec` (the matrix itself on one GPU; the rank's own diagonal block in a
// row-partitioned run, i.e. block-Jacobi with multigrid blocks: no communication inside the cycle).
// Reductions go through RCCL when the context has a communicator; the halo plan (if any) is applied
// before every product.
static thread_local bool t_last_solve_stagnated = false;   // the last solve_one ended at the evaluation floor of b - A x

// the sampled events of a timed solve: destroyed on every way out of the function that made them
struct EventList {
    std::vector<hipEvent_t> ev;
    ~EventList() { clear(); }
    void clear() {
        for (hipEvent_t e : ev) (void)hipEventDestroy(e);
        ev.clear();
    }
    bool empty() const { return ev.empty(); }
    size_t size() const { return ev.size(); }
    hipEvent_t operator[](size_t i) const { return ev[i]; }
    void push_back(hipEvent_t e) { ev.push_back(e); }
    hipEvent_t back() const { return ev.back(); }
};

static int solve_one(padne_ctx *ctx, const padne_csr *a, const padne_csr *prec, const double *b, double *x,
                     const padne_solve_opts *o, padne_solve_info *info, bool x_is_guess) {
    t_last_solve_stagnated = false;
    const bool dist = comm_active(ctx);
    const bool halo = ctx->halo_on;
    const bool amg = prec != nullptr;
    const long long nr = a->n_rows;                       // matrix rows (owned rows + empty exchange rows)
    const long long n = halo ? ctx->halo_n_owned : nr;    // owned unknowns = length of b, x, r, z
    const long long nc = a->n_cols;                       // length of any vector the matrix multiplies
    if (halo) {
        PADNE_REQUIRE(nc == ctx->halo_n_owned + (long long)ctx->world * ctx->halo_m && nr <= nc,
                      "matrix shape does not match the halo plan");
    } else {
        PADNE_REQUIRE(nr == nc, "matrix must be square");
    }
    if (amg)
        PADNE_REQUIRE(prec->n_rows == n && (prec->n_cols == n || prec == a),
                      "preconditioner must be the matrix itself or its owned x owned block");
    // The search directions of a one-GPU multigrid solve are kept (single precision, 4 bytes per row and iteration) and x is
    // formed from them when the loop has ended -- or when the ring of kXHist places is full -- instead of being read and
    // written in every iteration: pcg_x_flush_kernel.  A place per iteration up to 2 GiB of them, at least eight.
    constexpr int kXHist = 32;
    static_assert(kXHist <= 64, "pcg_x_flush_kernel holds the step lengths of one flush in 64 words of LDS");
    int n_hist = 0;
    if (amg && !dist && !halo && !ctx->opt.pcg_p64 && !ctx->opt.pcg_no_xhist && n > 0) {
        n_hist = (int)std::min<long long>(kXHist, std::max<long long>(8, ((long long)2 << 30) / (4 * n)));
        if (ctx->opt.force_xhist_small) n_hist = 8;
    }
    const size_t hist_stride = ((size_t)n + 63) & ~(size_t)63;      // floats per place
    const size_t base_bytes = (sizeof(double) * (size_t)(2 * n + 2 * nc + nr) + 4096 + 255) & ~(size_t)255;
    PADNE_TRY(ensure_workspace(ctx, base_bytes + (n_hist > 0 ? sizeof(float) * hist_stride * (size_t)n_hist + sizeof(double) * kXHist : 0)));
    double *r = (double *)ctx->ws;
    double *z = r + n;          // [n]   multigrid output
    double *p = z + n;          // [nc]
    double *q = p + nc;         // [nr]
    double *xe = q + nr;        // [nc]  extended copy of x for products A x (halo runs only)
    PcgStatus *st = (PcgStatus *)ctx->status;
    PcgStatus *hst = (PcgStatus *)ctx->pinned;
    hipStream_t s = ctx->stream;
    const int gv = vec_grid(n);
    const int gs = spmv_partials(a);
    const int P_rz = amg ? amg_rz_partials(prec) : gv;    // workgroups that emit r.z partials
    const int max_iter = o->max_iter > 0 ? o->max_iter : 100000;
    const int check_every = o->check_every > 0 ? o->check_every : (amg ? 4 : 50);
    const int sample_stride = amg ? 4 : 16;
    double *scal = ctx->scalars;
    // the single-precision vectors of the loop (cycle input, z, the stored search direction) are kept in units of ||b||.  From
    // an initial guess the first residual may lie dozens of orders below ||b|| (1e-30 ||b||: its floats would be denormals
    // or zero and p.q = 0 a breakdown where the double loop iterated): the unit is then ||r_0|| of that start.
    const double *bb_scalar = x_is_guess ? scal + S_NRM : scal + S_BB;
    float e_jac = 0.f;
    const float *e_dinv32 = nullptr;
    float *e_b32 = nullptr, *e_xa32 = nullptr;
    const bool fuse_entry = amg && amg_f32_entry_args(prec, &e_jac, &e_dinv32, &e_b32, &e_xa32);
    const bool defer_x = fuse_entry;      // x += alpha p rides on the p update (12 us per iteration at 10 M unknowns)
    float *z32 = defer_x && !dist && !halo ? (float *)z : nullptr;      // z of the loop: single precision, in z's own memory
    // ... and the search direction with it: p is stored as floats (in p's own memory), multiplied in double (csr_spmv_kernel
    // <..., float>): 8 bytes per row less in q = A p and in the p update, the same iteration counts (scripts/lab/exp_p32.py)
    float *p32 = z32 != nullptr && spmv_x32_ok(a) ? (float *)p : nullptr;
    if (p32 == nullptr) n_hist = 0;
    float *hist = n_hist > 0 ? (float *)((char *)ctx->ws + base_bytes) : nullptr;
    double *alpha_hist = n_hist > 0 ? (double *)(hist + hist_stride * (size_t)n_hist) : nullptr;
    if (hist != nullptr) p32 = hist;      // direction j of a (re)start lies in place j % n_hist
    auto p_place = [&](long long j) { return hist + hist_stride * (size_t)(j % n_hist); };

    PADNE_HIP_CHECK(hipMemsetAsync(st, 0, sizeof(PcgStatus), s));
    if (halo) PADNE_HIP_CHECK(hipMemsetAsync(p + n, 0, sizeof(double) * (size_t)(nc - n), s));
    PADNE_HIP_CHECK(hipEventRecord(ctx->ev0, s));

    auto product_Ax = [&](double *out) -> int {            // out = A x for the owned-length x
        if (!halo) return launch_spmv(ctx, a, x, out, nullptr, nullptr, nullptr);
        PADNE_HIP_CHECK(hipMemcpyAsync(xe, x, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, s));
        PADNE_HIP_CHECK(hipMemsetAsync(xe + n, 0, sizeof(double) * (size_t)(nc - n), s));
        PADNE_TRY(halo_exchange(ctx, xe, nullptr));
        return launch_spmv(ctx, a, xe, out, nullptr, nullptr, nullptr);
    };
    auto fold = [&](const double *first_slot, int P, long long stride, int count, double *out) -> int {
        hipLaunchKernelGGL(fold_partials_kernel, dim3(1), dim3(256), 0, s, first_slot, P, (int)stride, count, out);
        PADNE_HIP_CHECK(hipGetLastError());
        return PADNE_OK;
    };
    auto allreduce = [&](double *buf, int count) -> int { return dist ? comm_allreduce_sum_f64(ctx, buf, count) : PADNE_OK; };

    int restarts = 0, total_iters = 0, code = PADNE_OK;
    const bool sample_spmv = (o->flags & 2) != 0;
    EventList ev_a, ev_b;
    long long launched = 0;
    double true_rr = 0.0, bb = 0.0, tol2 = 0.0, prev_true_rr = 0.0;
    bool have_ax = false, stagnated = false;
    if (x_is_guess) {
        PADNE_TRY(product_Ax(q));
        have_ax = true;
    } else {
        PADNE_HIP_CHECK(hipMemsetAsync(x, 0, sizeof(double) * (size_t)n, s));
    }
    for (;;) {
        // ---- (re)start: r = b - A x ; z = M^-1 r ; p = z ------------------------------------------------
        if (amg) {
            hipLaunchKernelGGL(pcg_init_plain_kernel, dim3(gv), dim3(256), 0, s, n, b, have_ax ? q : nullptr, r,
                               slot(ctx, SLOT_RR), slot(ctx, SLOT_BB));
            PADNE_HIP_CHECK(hipGetLastError());
            // ||b||^2 first: the single-precision cycle normalises its input with it
            PADNE_TRY(fold(slot(ctx, SLOT_RR), gv, kMaxPartials, 2, scal + S_RR));   // RR, BB adjacent
            PADNE_TRY(allreduce(scal + S_RR, 2));
            if (x_is_guess && restarts == 0)
                PADNE_HIP_CHECK(hipMemcpyAsync(scal + S_NRM, scal + S_RR, sizeof(double), hipMemcpyDeviceToDevice, s));
            PADNE_TRY(amg_apply(ctx, prec, r, z, slot(ctx, SLOT_RZ0), nullptr, bb_scalar));
            if (p32 != nullptr) {
                if (hist != nullptr) p32 = hist;      // direction 0 of this (re)start
                hipLaunchKernelGGL(p_hat_from_z_kernel, dim3(gv), dim3(256), 0, s, n, (const double *)z, bb_scalar, p32);
                PADNE_HIP_CHECK(hipGetLastError());
            } else {
                PADNE_HIP_CHECK(hipMemcpyAsync(p, z, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, s));
            }
        } else {
            hipLaunchKernelGGL(pcg_init_kernel, dim3(gv), dim3(256), 0, s, n, b, have_ax ? q : nullptr, a->dinv, r, p,
                               slot(ctx, SLOT_RZ0), slot(ctx, SLOT_RR), slot(ctx, SLOT_BB));
            PADNE_HIP_CHECK(hipGetLastError());
            PADNE_TRY(fold(slot(ctx, SLOT_RR), gv, kMaxPartials, 2, scal + S_RR));   // RR, BB adjacent
            PADNE_TRY(allreduce(scal + S_RR, 2));
            if (x_is_guess && restarts == 0)
                PADNE_HIP_CHECK(hipMemcpyAsync(scal + S_NRM, scal + S_RR, sizeof(double), hipMemcpyDeviceToDevice, s));
        }
        if (dist) {
            PADNE_TRY(fold(slot(ctx, SLOT_RZ0), P_rz, kMaxPartials, 1, scal + S_RZRR0));
            PADNE_TRY(allreduce(scal + S_RZRR0, 1));
        }
        hipLaunchKernelGGL(pcg_set_tolerance_kernel, dim3(1), dim3(1), 0, s, st, scal + S_RR, o->rtol, o->atol,
                           restarts > 0 ? 1 : 0);
        PADNE_HIP_CHECK(hipGetLastError());
        int parity = 0;
        bool done = false;
        long long jq = 0, jf = 0;      // iterations of this (re)start queued / covered by a flush of x
        auto flush_x = [&]() -> int {
            if (hist == nullptr || jq == jf) return PADNE_OK;
            hipLaunchKernelGGL(pcg_x_flush_kernel, dim3(gv), dim3(256), 0, s, n, (const float *)hist, hist_stride, n_hist,
                               (const double *)alpha_hist, (int)jf, (int)jq, (const PcgStatus *)st, x);
            PADNE_HIP_CHECK(hipGetLastError());
            jf = jq;
            return PADNE_OK;
        };
        while (!done) {
            for (int k = 0; k < check_every; ++k) {
                const int rz_old_slot = parity ? SLOT_RZ1 : SLOT_RZ0;
                const int rz_new_slot = parity ? SLOT_RZ0 : SLOT_RZ1;
                double *s_old = scal + (parity ? S_RZRR1 : S_RZRR0);   // {rz, rr} reduced over ranks
                double *s_new = scal + (parity ? S_RZRR0 : S_RZRR1);
                // how the consumers read their scalars: per-workgroup partials, or one reduced value
                const double *rz_old = dist ? s_old : slot(ctx, rz_old_slot);
                const double *rz_new = dist ? s_new : slot(ctx, rz_new_slot);
                const double *pq = dist ? scal + S_PQ : slot(ctx, SLOT_PQ);
                const double *rr = dist ? s_new + 1 : slot(ctx, SLOT_RR);
                const int Pz = dist ? 1 : P_rz, Pq = dist ? 1 : gs, Pr = dist ? 1 : gv;

                const bool sampled = sample_spmv && (launched++ % sample_stride) == 1 && ev_a.size() < 512;
                if (sampled) {
                    hipEvent_t e0, e1;
                    PADNE_HIP_CHECK(hipEventCreate(&e0));
                    PADNE_HIP_CHECK(hipEventCreate(&e1));
                    ev_a.push_back(e0);
                    ev_b.push_back(e1);
                    PADNE_HIP_CHECK(hipEventRecord(e0, s));
                }
                if (hist != nullptr) p32 = p_place(jq);
                if (p32 != nullptr) PADNE_TRY(launch_spmv_dot_x32(ctx, a, p32, q, slot(ctx, SLOT_PQ), &st->done));
                else PADNE_TRY(halo_product_dot(ctx, a, p, q, slot(ctx, SLOT_PQ), &st->done));
                if (sampled) PADNE_HIP_CHECK(hipEventRecord(ev_b.back(), s));
                if (dist) {
                    PADNE_TRY(fold(slot(ctx, SLOT_PQ), gs, kMaxPartials, 1, scal + S_PQ));
                    PADNE_TRY(allreduce(scal + S_PQ, 1));
                }
                if (amg) {
                    if (fuse_entry)
                        hipLaunchKernelGGL(pcg_update_xr_entry_kernel, dim3(gv), dim3(256), 0, s, n, rz_old, Pz, pq, Pq, p, q,
                                           defer_x ? (double *)nullptr : x, r, slot(ctx, SLOT_RR), st, bb_scalar, e_jac, e_dinv32,
                                           e_b32, e_xa32, p32 != nullptr ? 1 : 0);
                    else
                        hipLaunchKernelGGL(pcg_update_xr_plain_kernel, dim3(gv), dim3(256), 0, s, n, rz_old, Pz, pq, Pq, p, q,
                                           x, r, slot(ctx, SLOT_RR), st);
                    PADNE_HIP_CHECK(hipGetLastError());
                    PADNE_TRY(amg_apply(ctx, prec, r, z, slot(ctx, rz_new_slot), &st->done, bb_scalar, fuse_entry, z32));
                    if (dist) {
                        PADNE_TRY(fold(slot(ctx, rz_new_slot), P_rz, kMaxPartials, 1, s_new));
                        PADNE_TRY(fold(slot(ctx, SLOT_RR), gv, kMaxPartials, 1, s_new + 1));
                        PADNE_TRY(allreduce(s_new, 2));
                    }
                    if (hist != nullptr) {
                        hipLaunchKernelGGL(pcg_update_p_z_kernel, dim3(gv), dim3(256), 0, s, n, rz_new, rz_old, Pz, rr, Pr,
                                           pq, Pq, z, p, st, max_iter - total_iters, (double *)nullptr, (const float *)z32,
                                           bb_scalar, p32, p_place(jq + 1), alpha_hist + (jq % n_hist));
                        ++jq;
                        // the place of direction jf is written again by iteration jf + n_hist - 1
                        if (jq - jf >= n_hist - 1) PADNE_TRY(flush_x());
                    } else {
                        hipLaunchKernelGGL(pcg_update_p_z_kernel, dim3(gv), dim3(256), 0, s, n, rz_new, rz_old, Pz, rr, Pr,
                                           pq, Pq, z, p, st, max_iter - total_iters, defer_x ? x : (double *)nullptr,
                                           (const float *)z32, bb_scalar, p32);
                    }
                } else {
                    hipLaunchKernelGGL(pcg_update_xr_kernel, dim3(gv), dim3(256), 0, s, n, rz_old, Pz, pq, Pq, p, q,
                                       a->dinv, x, r, slot(ctx, rz_new_slot), slot(ctx, SLOT_RR), st);
                    if (dist) {
                        PADNE_TRY(fold(slot(ctx, rz_new_slot), gv, (long long)(SLOT_RR - rz_new_slot) * kMaxPartials, 2,
                                       s_new));
                        PADNE_TRY(allreduce(s_new, 2));
                    }
                    hipLaunchKernelGGL(pcg_update_p_kernel, dim3(gv), dim3(256), 0, s, n, rz_new, rz_old, Pz, rr, Pr, pq,
                                       Pq, r, a->dinv, p, st, max_iter - total_iters);
                }
                parity ^= 1;
            }
            PADNE_HIP_CHECK(hipGetLastError());
            // (through the mailbox: the host polls instead of sleeping in hipStreamSynchronize, whose wake-up is at the mercy of
            // whatever else keeps the cores busy -- spinning BLAS workers of the caller tripled this loop's wall time)
            static_assert(sizeof(PcgStatus) <= 56, "one mailbox slot");
            PADNE_TRY(read_back(ctx, st, sizeof(PcgStatus), hst));
            done = hst->done != 0;
        }
        PADNE_TRY(flush_x());
        total_iters += hst->iters;
        code = hst->code;
        bb = hst->bb;
        tol2 = hst->tol2;
        // ---- true residual ---------------------------------------------------------------------------
        PADNE_TRY(product_Ax(q));
        hipLaunchKernelGGL(residual_kernel, dim3(gv), dim3(256), 0, s, n, b, q, (double *)nullptr, slot(ctx, SLOT_TMP));
        PADNE_TRY(fold(slot(ctx, SLOT_TMP), gv, kMaxPartials, 1, scal + S_TRUE));
        PADNE_TRY(allreduce(scal + S_TRUE, 1));
        PADNE_TRY(read_back(ctx, scal + S_TRUE, sizeof(double), &true_rr));
        if (code != PADNE_OK) break;
        if (true_rr <= tol2 * 1.0000001 || total_iters >= max_iter || restarts >= 8) break;
        // A restart that does not even halve the true residual means b - A x has reached what binary64 can
        // evaluate (about eps |A||x| per row, ~1e-12 ||b|| at N = 5 M): stop instead of spinning
        if (restarts > 0 && true_rr >= 0.25 * prev_true_rr) {
            stagnated = true;
            t_last_solve_stagnated = true;
            break;
        }
        prev_true_rr = true_rr;
        // the recurrence residual drifted from the true one: restart from the true residual
        ++restarts;
        have_ax = true;  // q = A x is current
        PADNE_HIP_CHECK(hipMemsetAsync(st, 0, 4 * sizeof(int32_t), s));   // done = code = iters = done_seen = 0
    }
    PADNE_HIP_CHECK(hipEventRecord(ctx->ev1, s));
    PADNE_HIP_CHECK(hipEventSynchronize(ctx->ev1));
    PADNE_TRY(comm_p2p_check(ctx));      // (mailboxes shared between processes: did a receiver give up waiting?)
    float ms = 0.f;
    PADNE_HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    if (!ev_a.empty()) {
        // launches enqueued after convergence are no-ops: keep the samples within 2x of the median
        std::vector<double> t_s;
        for (size_t i = 0; i < ev_a.size(); ++i) {
            float t = 0.f;
            if (hipEventElapsedTime(&t, ev_a[i], ev_b[i]) == hipSuccess) t_s.push_back(t * 1e-3);
        }
        if (info && !t_s.empty()) {
            std::vector<double> sorted = t_s;
            std::sort(sorted.begin(), sorted.end());
            const double med = sorted[sorted.size() / 2];
            double sum = 0.0;
            int cnt = 0;
            for (double t : t_s)
                if (t >= 0.5 * med && t <= 2.0 * med) {
                    sum += t;
                    ++cnt;
                }
            if (cnt > 0) info->spmv_seconds = sum / cnt;
        }
    }
    if (info) {
        info->iterations += total_iters;
        info->restarts += restarts;
        const double rel = bb > 0 ? sqrt(true_rr / bb) : sqrt(true_rr);
        if (rel > info->rel_residual) info->rel_residual = rel;
        if (sqrt(true_rr) > info->abs_residual) info->abs_residual = sqrt(true_rr);
        info->solve_seconds += ms * 1e-3;
        if (code != PADNE_OK) info->status = code;
        else if (true_rr > tol2 * 1.0000001 && info->status == PADNE_OK) {
            // at the evaluation floor a residual within 10x of the request counts as converged; the achieved
            // value is reported in rel_residual either way
            if (!(stagnated && true_rr <= 100.0 * tol2)) info->status = PADNE_E_NOTCONVERGED;
        }
    }
    return PADNE_OK;
}

// ---- single-reduction CG (Chronopoulos / Gear) for row-partitioned runs ------------------------------------
// The textbook loop needs two global reductions per iteration (p.q before the update of x and r; r.z and r.r after
// the preconditioner), each an all-reduce of one or two doubles whose cost across ranks is pure latency.  Rearranged
// so that the product follows the preconditioner,
//     x += alpha p ; r -= alpha s ; z = M r ; w = A z ;
//     gamma' = r.z , delta = z.w , rr = r.r              <- ONE all-reduce of three doubles
//     beta = gamma' / gamma ; alpha = gamma' / (delta - beta gamma' / alpha) ; p = z + beta p ; s = w + beta s
// (s = A p by recurrence), one iteration costs one all-reduce and one exchange of z.  The stopping test sees r.r only
// together with the other two sums, i.e. after the cycle and the product of the iteration that converged: one
// superfluous V-cycle + product per solve buys one collective less in every iteration.  Scalars live in `sr`:
// (gamma and alpha of the previous iteration in two alternating pairs: the kernel that forms the new ones reads the old ones
// in all its workgroups while its first workgroup writes)
enum { SR_STATE = 0 /* [2][2]: gamma, alpha */, SR_RED = 4 /* gamma', delta, rr */ };

__global__ __launch_bounds__(256) void sr_fold3_kernel(const double *__restrict__ p0, int n0, const double *__restrict__ p1,
                                                       int n1, const double *__restrict__ p2, int n2,
                                                       double *__restrict__ out) {
    __shared__ double red[4];
    const double *src = blockIdx.x == 0 ? p0 : (blockIdx.x == 1 ? p1 : p2);
    const int cnt = blockIdx.x == 0 ? n0 : (blockIdx.x == 1 ? n1 : n2);
    const double t = block_total(src, cnt, red);
    if (threadIdx.x == 0) out[blockIdx.x] = t;
}

// One iteration's scalar and vector work in ONE launch (the three reduced sums are in, the next cycle needs r):
//     beta = gamma' / gamma ; alpha = gamma' / (delta - beta gamma' / alpha)      (first step of a start: alpha = gamma' / delta)
//     p = z + beta p ; s = w + beta s ; x += alpha p ; r -= alpha s ; partial r.r ;
//     optionally the entry stage of the single-precision cycle (as pcg_update_xr_entry_kernel)
// Every workgroup forms the scalars and the stopping decision from the same numbers (the reduced sums, the previous gamma and
// alpha, the launch's number `j` -- nothing this launch writes), the first one records them.  Until round 5 these were three
// launches (scalars; p, s; x, r -- the latter two on either side of NO reduction) and p and s crossed memory twice.
__global__ __launch_bounds__(256) void sr_step_kernel(
    const long long n, double *__restrict__ sr, const double *__restrict__ z, const double *__restrict__ w,
    double *__restrict__ p, double *__restrict__ s, double *__restrict__ x, double *__restrict__ r,
    double *__restrict__ part_rr, PcgStatus *__restrict__ st, const int first, const int j, const int max_iter,
    const double *__restrict__ bb2, const float c, const float *__restrict__ dinv32, float *__restrict__ b32,
    float *__restrict__ xa32) {
    __shared__ double red[4];
    if (st->done) return;                              // (set by an EARLIER launch: every workgroup reads the same value)
    const double gn = sr[SR_RED + 0], delta = sr[SR_RED + 1], rr = sr[SR_RED + 2];
    const double *old = sr + SR_STATE + 2 * (j & 1);
    double alpha, beta = 0.0;
    if (first) {
        alpha = gn / delta;
    } else {
        beta = gn / old[0];
        alpha = gn / (delta - beta * gn / old[1]);
    }
    bool stop = !first && (rr <= st->tol2 || j >= max_iter);
    int code = PADNE_OK;
    if (!stop && (!(alpha > 0.0) || !(alpha == alpha) || !(gn > 0.0) || !(rr == rr))) {
        code = PADNE_E_BREAKDOWN;      // p.A p = gamma' / alpha <= 0, or the cycle lost definiteness, or NaN
        stop = true;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double *nw = sr + SR_STATE + 2 * ((j & 1) ^ 1);
        nw[0] = gn;
        nw[1] = alpha;
        if (!first) {
            st->iters = j;
            st->rr = rr;
        }
        if (code != PADNE_OK) st->code = code;
        if (stop) st->done = 1;
    }
    if (stop) return;
    double s_inv = 1.0;
    if (b32 != nullptr) {
        const double s2 = *bb2;
        s_inv = s2 > 0.0 ? 1.0 / sqrt(s2) : 1.0;
    }
    double s_rr = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        // (first step: p and s hold whatever the workspace held -- never read)
        const double pi = first ? z[i] : z[i] + beta * p[i];
        const double si = first ? w[i] : w[i] + beta * s[i];
        p[i] = pi;
        s[i] = si;
        const double ri = r[i] - alpha * si;
        x[i] += alpha * pi;
        r[i] = ri;
        s_rr += ri * ri;
        if (b32 != nullptr) {
            const float v = (float)(ri * s_inv);
            b32[i] = v;
            if (xa32 != nullptr) xa32[i] = c * dinv32[i] * v;
        }
    }
    block_store_partial(s_rr, red, part_rr + blockIdx.x);
}


static int solve_one_single_reduction(padne_ctx *ctx, const padne_csr *a, const padne_csr *prec, const double *b,
                                      double *x, const padne_solve_opts *o, padne_solve_info *info, bool x_is_guess) {
    t_last_solve_stagnated = false;
    const bool dist = comm_active(ctx);
    const bool halo = ctx->halo_on;
    const long long nr = a->n_rows;
    const long long n = halo ? ctx->halo_n_owned : nr;
    const long long nc = a->n_cols;
    if (halo) {
        PADNE_REQUIRE(nc == ctx->halo_n_owned + (long long)ctx->world * ctx->halo_m && nr <= nc,
                      "matrix shape does not match the halo plan");
    } else {
        PADNE_REQUIRE(nr == nc, "matrix must be square");
    }
    PADNE_REQUIRE(prec != nullptr && prec->n_rows == n && (prec->n_cols == n || prec == a),
                  "preconditioner must be the matrix itself or its owned x owned block");
    PADNE_TRY(ensure_workspace(ctx, sizeof(double) * (size_t)(3 * n + 2 * nc + 2 * nr) + 4096));
    double *r = (double *)ctx->ws;
    double *p = r + n;          // [n]
    double *sv = p + n;         // [n]   s = A p (recurrence)
    double *z = sv + n;         // [nc]  multigrid output with room for the exchanged values
    double *w = z + nc;         // [nr]  A z
    double *q = w + nr;         // [nr]  A x of the true-residual checks
    double *xe = q + nr;        // [nc]
    PcgStatus *st = (PcgStatus *)ctx->status;
    PcgStatus *hst = (PcgStatus *)ctx->pinned;
    hipStream_t s = ctx->stream;
    const int gv = vec_grid(n);
    const int gs = spmv_partials(a);
    const int P_rz = amg_rz_partials(prec);
    const int max_iter = o->max_iter > 0 ? o->max_iter : 100000;
    const int check_every = o->check_every > 0 ? o->check_every : 4;
    double *scal = ctx->scalars;
    double *sr = scal + 16;                      // SR_* slots (ctx->scalars holds 64 doubles)
    // the single-precision vectors of the loop (cycle input, z, the stored search direction) are kept in units of ||b||.  From
    // an initial guess the first residual may lie dozens of orders below ||b|| (1e-30 ||b||: its floats would be denormals
    // or zero and p.q = 0 a breakdown where the double loop iterated): the unit is then ||r_0|| of that start.
    const double *bb_scalar = x_is_guess ? scal + S_NRM : scal + S_BB;
    float e_jac = 0.f;
    const float *e_dinv32 = nullptr;
    float *e_b32 = nullptr, *e_xa32 = nullptr;
    const bool fuse_entry = amg_f32_entry_args(prec, &e_jac, &e_dinv32, &e_b32, &e_xa32);

    PADNE_HIP_CHECK(hipMemsetAsync(st, 0, sizeof(PcgStatus), s));
    if (halo) PADNE_HIP_CHECK(hipMemsetAsync(z + n, 0, sizeof(double) * (size_t)(nc - n), s));
    PADNE_HIP_CHECK(hipEventRecord(ctx->ev0, s));
    auto product_Ax = [&](double *out) -> int {
        if (!halo) return launch_spmv(ctx, a, x, out, nullptr, nullptr, nullptr);
        PADNE_HIP_CHECK(hipMemcpyAsync(xe, x, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, s));
        PADNE_HIP_CHECK(hipMemsetAsync(xe + n, 0, sizeof(double) * (size_t)(nc - n), s));
        PADNE_TRY(halo_exchange(ctx, xe, nullptr));
        return launch_spmv(ctx, a, xe, out, nullptr, nullptr, nullptr);
    };
    auto fold = [&](const double *first_slot, int P, long long stride, int count, double *out) -> int {
        hipLaunchKernelGGL(fold_partials_kernel, dim3(1), dim3(256), 0, s, first_slot, P, (int)stride, count, out);
        PADNE_HIP_CHECK(hipGetLastError());
        return PADNE_OK;
    };
    auto allreduce = [&](double *buf, int count) -> int { return dist ? comm_allreduce_sum_f64(ctx, buf, count) : PADNE_OK; };
    // z = M r, exchange, w = A z and the three sums of the iteration in one reduction
    const bool sample_spmv = (o->flags & 2) != 0;
    EventList ev_a, ev_b;
    long long launched = 0;
    auto cycle_product_reduce = [&](bool entry_done, const int32_t *done_flag) -> int {
        PADNE_TRY(amg_apply(ctx, prec, r, z, slot(ctx, SLOT_RZ0), done_flag, bb_scalar, entry_done));
        const bool sampled = sample_spmv && (launched++ % 4) == 1 && ev_a.size() < 512;
        if (sampled) {
            hipEvent_t e0, e1;
            PADNE_HIP_CHECK(hipEventCreate(&e0));
            PADNE_HIP_CHECK(hipEventCreate(&e1));
            ev_a.push_back(e0);
            ev_b.push_back(e1);
            PADNE_HIP_CHECK(hipEventRecord(e0, s));
        }
        PADNE_TRY(halo_product_dot(ctx, a, z, w, slot(ctx, SLOT_PQ), done_flag));
        if (sampled) PADNE_HIP_CHECK(hipEventRecord(ev_b.back(), s));
        hipLaunchKernelGGL(sr_fold3_kernel, dim3(3), dim3(256), 0, s, slot(ctx, SLOT_RZ0), P_rz, slot(ctx, SLOT_PQ), gs,
                           slot(ctx, SLOT_RR), gv, sr + SR_RED);
        PADNE_HIP_CHECK(hipGetLastError());
        return allreduce(sr + SR_RED, 3);
    };

    int restarts = 0, total_iters = 0, code = PADNE_OK;
    double true_rr = 0.0, bb = 0.0, tol2 = 0.0, prev_true_rr = 0.0;
    bool have_ax = false, stagnated = false;
    if (x_is_guess) {
        PADNE_TRY(product_Ax(q));
        have_ax = true;
    } else {
        PADNE_HIP_CHECK(hipMemsetAsync(x, 0, sizeof(double) * (size_t)n, s));
    }
    for (;;) {
        // ---- (re)start: r = b - A x ; z = M r ; w = A z ; p = z ; s = w --------------------------------
        hipLaunchKernelGGL(pcg_init_plain_kernel, dim3(gv), dim3(256), 0, s, n, b, have_ax ? q : nullptr, r,
                           slot(ctx, SLOT_RR), slot(ctx, SLOT_BB));
        PADNE_HIP_CHECK(hipGetLastError());
        PADNE_TRY(fold(slot(ctx, SLOT_RR), gv, kMaxPartials, 2, scal + S_RR));   // RR, BB adjacent: ||b|| normalises the cycle
        PADNE_TRY(allreduce(scal + S_RR, 2));
        if (x_is_guess && restarts == 0)
            PADNE_HIP_CHECK(hipMemcpyAsync(scal + S_NRM, scal + S_RR, sizeof(double), hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(pcg_set_tolerance_kernel, dim3(1), dim3(1), 0, s, st, scal + S_RR, o->rtol, o->atol,
                           restarts > 0 ? 1 : 0);
        PADNE_HIP_CHECK(hipGetLastError());
        PADNE_TRY(cycle_product_reduce(false, &st->done));
        int step_no = 0;                       // steps queued since this (re)start: the j-th one completes iteration j
        hipLaunchKernelGGL(sr_step_kernel, dim3(gv), dim3(256), 0, s, n, sr, (const double *)z, (const double *)w, p, sv, x, r,
                           slot(ctx, SLOT_RR), st, 1, step_no, max_iter - total_iters, bb_scalar, e_jac, e_dinv32,
                           fuse_entry ? e_b32 : (float *)nullptr, e_xa32);
        PADNE_HIP_CHECK(hipGetLastError());
        bool done = false;
        while (!done) {
            for (int k = 0; k < check_every; ++k) {
                PADNE_TRY(cycle_product_reduce(fuse_entry, &st->done));
                ++step_no;
                hipLaunchKernelGGL(sr_step_kernel, dim3(gv), dim3(256), 0, s, n, sr, (const double *)z, (const double *)w, p, sv, x,
                                   r, slot(ctx, SLOT_RR), st, 0, step_no, max_iter - total_iters, bb_scalar, e_jac, e_dinv32,
                                   fuse_entry ? e_b32 : (float *)nullptr, e_xa32);
            }
            PADNE_HIP_CHECK(hipGetLastError());
            // (through the mailbox: the host polls instead of sleeping in hipStreamSynchronize, whose wake-up is at the mercy of
            // whatever else keeps the cores busy -- spinning BLAS workers of the caller tripled this loop's wall time)
            static_assert(sizeof(PcgStatus) <= 56, "one mailbox slot");
            PADNE_TRY(read_back(ctx, st, sizeof(PcgStatus), hst));
            done = hst->done != 0;
        }
        total_iters += hst->iters;
        code = hst->code;
        bb = hst->bb;
        tol2 = hst->tol2;
        // ---- true residual ---------------------------------------------------------------------------
        PADNE_TRY(product_Ax(q));
        hipLaunchKernelGGL(residual_kernel, dim3(gv), dim3(256), 0, s, n, b, q, (double *)nullptr, slot(ctx, SLOT_TMP));
        PADNE_TRY(fold(slot(ctx, SLOT_TMP), gv, kMaxPartials, 1, scal + S_TRUE));
        PADNE_TRY(allreduce(scal + S_TRUE, 1));
        PADNE_TRY(read_back(ctx, scal + S_TRUE, sizeof(double), &true_rr));
        if (code != PADNE_OK) break;
        if (true_rr <= tol2 * 1.0000001 || total_iters >= max_iter || restarts >= 8) break;
        if (restarts > 0 && true_rr >= 0.25 * prev_true_rr) {      // the evaluation floor of b - A x (see solve_one)
            stagnated = true;
            t_last_solve_stagnated = true;
            break;
        }
        prev_true_rr = true_rr;
        ++restarts;
        have_ax = true;
        PADNE_HIP_CHECK(hipMemsetAsync(st, 0, 4 * sizeof(int32_t), s));
    }
    PADNE_HIP_CHECK(hipEventRecord(ctx->ev1, s));
    PADNE_HIP_CHECK(hipEventSynchronize(ctx->ev1));
    PADNE_TRY(comm_p2p_check(ctx));      // (mailboxes shared between processes: did a receiver give up waiting?)
    float ms = 0.f;
    PADNE_HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    if (!ev_a.empty()) {       // launches queued after convergence are no-ops: keep the samples within 2x of the median
        std::vector<double> t_s;
        for (size_t i = 0; i < ev_a.size(); ++i) {
            float t = 0.f;
            if (hipEventElapsedTime(&t, ev_a[i], ev_b[i]) == hipSuccess) t_s.push_back(t * 1e-3);
        }
        if (info && !t_s.empty()) {
            std::vector<double> sorted = t_s;
            std::sort(sorted.begin(), sorted.end());
            const double med = sorted[sorted.size() / 2];
            double sum = 0.0;
            int cnt = 0;
            for (double t : t_s)
                if (t >= 0.5 * med && t <= 2.0 * med) {
                    sum += t;
                    ++cnt;
                }
            if (cnt > 0) info->spmv_seconds = sum / cnt;
        }
    }
    if (info) {
        info->iterations += total_iters;
        info->restarts += restarts;
        const double rel = bb > 0 ? sqrt(true_rr / bb) : sqrt(true_rr);
        if (rel > info->rel_residual) info->rel_residual = rel;
        if (sqrt(true_rr) > info->abs_residual) info->abs_residual = sqrt(true_rr);
        info->solve_seconds += ms * 1e-3;
        if (code != PADNE_OK) info->status = code;
        else if (true_rr > tol2 * 1.0000001 && info->status == PADNE_OK) {
            if (!(stagnated && true_rr <= 100.0 * tol2)) info->status = PADNE_E_NOTCONVERGED;
        }
    }
    return PADNE_OK;
}

// ---- 8 right-hand sides in lockstep (config C5) ----------------------------------------------------------
// The same preconditioned CG recurrences, one per right-hand side, advanced together so that the matrix and the
// multigrid operators are streamed once per iteration for all of them (spmm.hip; vectors interleaved [n][8]).
// Every column keeps its own alpha / beta / stopping test; a column that has converged is frozen (alpha = 0)
// while the others go on.  Reductions: per-workgroup partials [8][kMaxPartials] folded by an 8-workgroup kernel
// into 8 device scalars that the consumers read.
// (K = 8: config C5, groups of regulators; K = 4 / 2: the two to four right-hand sides of one to three regulators, whose
// lockstep iteration costs less than as many single ones -- the width is a template parameter of every kernel below and a
// run-time argument of the host functions)
int amg_apply_batch(padne_ctx *ctx, const padne_csr *A0, int k, const double *r8, double *z8, double *partials_rz,
                    const int32_t *done_flag, const double *bb2, bool entry_done = false, float *z32 = nullptr);
int amg_batch_entry_args(padne_ctx *ctx, const padne_csr *A0, float *jac, const float **dinv32, float **b8, float **xa8);
bool amg_supports_batch8(const padne_csr *A0);

struct Pcg8Status {
    int32_t done, code, iters, pad;
    int32_t col_done[8];
    int32_t col_iters[8];
    double rr[8], tol2[8], bb[8];
    // what the x / r update of an iteration read of `done` and `col_done` (values from BEFORE its launch): the p update of the
    // same iteration carries the deferred x += alpha p and must take the same decisions in every workgroup, whatever its
    // workgroup 0 has written to `done` / `col_done` in the meantime
    int32_t done_seen, pad2;
    int32_t col_seen[8];
};

// sums over this thread's strided share of an interleaved vector end up per column j = threadIdx.x & (K - 1);
// combine the threads of a workgroup that share j and store one partial per column
template <int K>
__device__ __forceinline__ void block_store_partial8(double v, double (*red)[8], double *part /* [K][kMaxPartials] */) {
#pragma unroll
    for (int d = K; d < 64; d <<= 1) v += __shfl_xor(v, d, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane < K) red[w][lane] = v;
    __syncthreads();
    if (threadIdx.x < K)
        part[(size_t)threadIdx.x * kMaxPartials + blockIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    __syncthreads();
}

__global__ __launch_bounds__(256) void fold8_kernel(const double *__restrict__ partials, int P, double *__restrict__ out) {
    __shared__ double red[4];
    const double t = block_total(partials + (size_t)blockIdx.x * kMaxPartials, P, red);
    if (threadIdx.x == 0) out[blockIdx.x] = t;
}

template <int K>
__global__ __launch_bounds__(256) void pcg8_init_kernel(const long long n, const double *__restrict__ b,
                                                        const double *__restrict__ ax, double *__restrict__ r,
                                                        double *__restrict__ part_rr, double *__restrict__ part_bb) {
    __shared__ double red[4][8];
    double rr = 0.0, bb = 0.0;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n * K; t += (long long)gridDim.x * 256) {
        const double bi = b[t];
        const double ri = ax ? bi - ax[t] : bi;
        r[t] = ri;
        rr += ri * ri;
        bb += bi * bi;
    }
    block_store_partial8<K>(rr, red, part_rr);
    block_store_partial8<K>(bb, red, part_bb);
}

__global__ void pcg8_set_tolerance_kernel(Pcg8Status *st, const double *__restrict__ rr, const double *__restrict__ bbv,
                                          double rtol, double atol, int use_existing_bb, const int K) {
    const int j = threadIdx.x;
    if (j < K) {
        const double bb = use_existing_bb ? st->bb[j] : bbv[j];
        double tol = rtol * sqrt(bb);
        if (atol > tol) tol = atol;
        st->bb[j] = bb;
        st->tol2[j] = tol * tol;
        st->rr[j] = rr[j];
        st->col_done[j] = (rr[j] <= tol * tol) ? 1 : 0;
        if (!(rr[j] == rr[j])) st->code = PADNE_E_BREAKDOWN;
    }
    __syncthreads();
    if (j == 0) {
        int all = 1;
        for (int c = 0; c < K; ++c) all &= st->col_done[c];
        st->done = (all || st->code != PADNE_OK) ? 1 : 0;
    }
}

template <int K>
__global__ __launch_bounds__(256) void pcg8_update_xr_kernel(const long long n, const double *__restrict__ rz,
                                                             const double *__restrict__ pq, const double *__restrict__ p,
                                                             const double *__restrict__ q, double *__restrict__ x,
                                                             double *__restrict__ r, double *__restrict__ part_rr,
                                                             const Pcg8Status *__restrict__ st,
                                                             // entry stage of the batched cycle (b32 may be null): b = r / ||b||
                                                             // and the first sweep, as amg_entry_f32xk_kernel -- the residual
                                                             // is in a register here and need not be read again
                                                             const double *__restrict__ bb2, const float c,
                                                             const float *__restrict__ dinv32, float *__restrict__ b32,
                                                             float *__restrict__ xa32) {
    __shared__ double red[4][8];
    if (st->done) return;
    const int j = threadIdx.x & (K - 1);
    const double alpha = st->col_done[j] ? 0.0 : rz[j] / pq[j];
    double s_inv = 1.0;
    if (b32 != nullptr) {
        const double s2 = bb2[j];
        if (s2 > 0.0) s_inv = 1.0 / sqrt(s2);
    }
    double s_rr = 0.0;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n * K; t += (long long)gridDim.x * 256) {
        const double ri = r[t] - alpha * q[t];
        x[t] += alpha * p[t];
        r[t] = ri;
        s_rr += ri * ri;
        if (b32 != nullptr) {
            const float v = (float)(ri * s_inv);
            b32[t] = v;
            if (xa32 != nullptr) xa32[t] = c * dinv32[t / K] * v;
        }
    }
    block_store_partial8<K>(s_rr, red, part_rr);
}

// p = z of a (re)start, z as the cycle leaves it: floats in units of sqrt(unit2[j])
template <int K>
__global__ void p8_from_z32_kernel(const long long n, const float *__restrict__ z32, const double *__restrict__ unit2,
                                   double *__restrict__ p) {
    const double s2 = unit2[threadIdx.x & (K - 1)];
    const double z_mul = s2 > 0.0 ? sqrt(s2) : 1.0;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n * K; t += (long long)gridDim.x * 256)
        p[t] = (double)z32[t] * z_mul;
}

// The two vector updates of the lockstep loop with z in single precision (in units of ||b_j||, as the cycle leaves it) and
// x += alpha p riding on the p update: p is read once per iteration instead of twice, z crosses memory as floats (per
// iteration and right-hand side 24 bytes less in the x / r update, 8 less in the cycle's exit, 4 more in the p update).
// The search directions stay doubles here: stored as floats (the single loop's form) the 8-wide product converts two
// floats per non-zero and lane at the rate of double-precision arithmetic and is bound by it -- 512 against 330 us at
// N = 5 M, more than the vector kernels save.
template <int K>
__global__ __launch_bounds__(256) void pcg8_update_r_entry_kernel(const long long n, const double *__restrict__ rz,
                                                                const double *__restrict__ pq, const double *__restrict__ q,
                                                                double *__restrict__ r, double *__restrict__ part_rr,
                                                                Pcg8Status *__restrict__ st, const double *__restrict__ unit2,
                                                                const float c, const float *__restrict__ dinv32,
                                                                float *__restrict__ b32, float *__restrict__ xa32) {
    __shared__ double red[4][8];
    const int stop = st->done;              // written by an EARLIER launch: every workgroup of this one reads the same value
    const int j = threadIdx.x & (K - 1);
    const int frozen = st->col_done[j];
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) st->done_seen = stop;
        if (threadIdx.x < K) st->col_seen[threadIdx.x] = frozen;
    }
    if (stop) return;
    const double s2 = unit2[j];
    const double s_inv = s2 > 0.0 ? 1.0 / sqrt(s2) : 1.0;
    const double alpha = frozen ? 0.0 : rz[j] / pq[j];
    double s_rr = 0.0;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n * K; t += (long long)gridDim.x * 256) {
        const double ri = r[t] - alpha * q[t];
        r[t] = ri;
        s_rr += ri * ri;
        const float v = (float)(ri * s_inv);
        b32[t] = v;
        if (xa32 != nullptr) xa32[t] = c * dinv32[t / K] * v;      // (null: the cycle forms its first sweep from b32 itself)
    }
    block_store_partial8<K>(s_rr, red, part_rr);
}

template <int K>
__global__ __launch_bounds__(256) void pcg8_update_p_z_kernel(const long long n, const double *__restrict__ rz_new,
                                                                const double *__restrict__ rz_old, const double *__restrict__ rr,
                                                                const double *__restrict__ pq, const double *__restrict__ unit2,
                                                                const float *__restrict__ z32, double *__restrict__ p,
                                                                double *__restrict__ x, Pcg8Status *__restrict__ st,
                                                                const int max_iter) {
    if (st->done_seen) return;
    const int j = threadIdx.x & (K - 1);
    const int frozen = st->col_seen[j];
    const double s2 = unit2[j];
    const double z_mul = s2 > 0.0 ? sqrt(s2) : 1.0;
    const double beta = frozen ? 0.0 : rz_new[j] / rz_old[j];
    const double alpha = frozen ? 0.0 : rz_old[j] / pq[j];
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n * K; t += (long long)gridDim.x * 256) {
        const double pi = p[t];
        x[t] += alpha * pi;
        p[t] = (double)z32[t] * z_mul + beta * pi;
    }
    if (blockIdx.x == 0) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const int it = st->iters + 1;
            st->iters = it;
            int all = 1;
            for (int c = 0; c < K; ++c) {
                if (st->col_done[c]) continue;
                st->col_iters[c] += 1;
                st->rr[c] = rr[c];
                if (!(pq[c] > 0.0) || !(rr[c] == rr[c]) || !(rz_new[c] > 0.0)) {
                    st->code = PADNE_E_BREAKDOWN;
                } else if (rr[c] <= st->tol2[c]) {
                    st->col_done[c] = 1;
                    continue;
                }
                all = 0;
            }
            if (all || st->code != PADNE_OK || it >= max_iter) st->done = 1;
        }
    }
}

template <int K>
__global__ __launch_bounds__(256) void pcg8_update_p_kernel(const long long n, const double *__restrict__ rz_new,
                                                            const double *__restrict__ rz_old, const double *__restrict__ rr,
                                                            const double *__restrict__ pq, const double *__restrict__ z,
                                                            double *__restrict__ p, Pcg8Status *__restrict__ st,
                                                            const int max_iter) {
    if (st->done) return;
    const int j = threadIdx.x & (K - 1);
    const double beta = st->col_done[j] ? 0.0 : rz_new[j] / rz_old[j];
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n * K; t += (long long)gridDim.x * 256)
        p[t] = z[t] + beta * p[t];
    if (blockIdx.x == 0) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const int it = st->iters + 1;
            st->iters = it;
            int all = 1;
            for (int c = 0; c < K; ++c) {
                if (st->col_done[c]) continue;
                st->col_iters[c] += 1;
                st->rr[c] = rr[c];
                if (!(pq[c] > 0.0) || !(rr[c] == rr[c]) || !(rz_new[c] > 0.0)) {
                    st->code = PADNE_E_BREAKDOWN;
                } else if (rr[c] <= st->tol2[c]) {
                    st->col_done[c] = 1;
                    continue;
                }
                all = 0;
            }
            if (all || st->code != PADNE_OK || it >= max_iter) st->done = 1;
        }
    }
}

template <int K>
__global__ __launch_bounds__(256) void residual8_kernel(const long long n, const double *__restrict__ b,
                                                        const double *__restrict__ ax, double *__restrict__ part_rr) {
    __shared__ double red[4][8];
    double rr = 0.0;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n * K; t += (long long)gridDim.x * 256) {
        const double ri = ax[t] - b[t];
        rr += ri * ri;
    }
    block_store_partial8<K>(rr, red, part_rr);
}

// b_cols / x_cols: K vectors of n doubles one after the other (the C ABI layout)
template <int K>
static int solve_batch(padne_ctx *ctx, const padne_csr *a, const double *b_cols, double *x_cols,
                       const padne_solve_opts *o, padne_solve_info *info, bool x_is_guess) {
    const long long n = a->n_rows;
    PADNE_REQUIRE(a->n_rows == a->n_cols && !ctx->halo_on, "the batched solve is single-GPU");
    hipStream_t s = ctx->stream;
    Scratch sc(ctx);
    double *b8 = nullptr, *x8 = nullptr, *r8 = nullptr, *z8 = nullptr, *p8 = nullptr, *q8 = nullptr, *part = nullptr,
           *scal = nullptr;
    const size_t nv = (size_t)n * K;
    PADNE_TRY(sc.alloc(&b8, nv));
    PADNE_TRY(sc.alloc(&x8, nv));
    PADNE_TRY(sc.alloc(&r8, nv));
    PADNE_TRY(sc.alloc(&z8, nv));
    PADNE_TRY(sc.alloc(&p8, nv));
    PADNE_TRY(sc.alloc(&q8, nv));
    PADNE_TRY(sc.alloc(&part, (size_t)6 * 8 * kMaxPartials));
    PADNE_TRY(sc.alloc(&scal, 64));
    enum { P_PQ = 0, P_RZ0 = 1, P_RZ1 = 2, P_RR = 3, P_BB = 4, P_TMP = 5 };
    auto pslot = [&](int k) { return part + (size_t)k * 8 * kMaxPartials; };
    enum { C_PQ = 0, C_RZ0 = 8, C_RZ1 = 16, C_RR = 24, C_BB = 32, C_TRUE = 40, C_NRM = 48 };
    Pcg8Status *st = (Pcg8Status *)ctx->status;
    Pcg8Status *hst = (Pcg8Status *)ctx->pinned;
    const int gv = vec_grid(n * K);
    const int gs = spmm8_grid(a);
    const int max_iter = o->max_iter > 0 ? o->max_iter : 100000;
    const int check_every = o->check_every > 0 ? o->check_every : 4;
    auto fold = [&](const double *partials, int P, double *out) -> int {
        hipLaunchKernelGGL(fold8_kernel, dim3(K), dim3(256), 0, s, partials, P, out);
        PADNE_HIP_CHECK(hipGetLastError());
        return PADNE_OK;
    };
    PADNE_HIP_CHECK(hipMemsetAsync(st, 0, sizeof(Pcg8Status), s));
    PADNE_TRY(interleave(ctx, n, K, b_cols, b8, true));
    PADNE_HIP_CHECK(hipEventRecord(ctx->ev0, s));
    bool have_ax = false;
    if (x_is_guess) {
        PADNE_TRY(interleave(ctx, n, K, x_cols, x8, true));
        PADNE_TRY(launch_spmm_mode(ctx, a, K, SPMV_PLAIN, x8, q8, nullptr, nullptr, nullptr, nullptr, nullptr, 0.0));
        have_ax = true;
    } else {
        PADNE_HIP_CHECK(hipMemsetAsync(x8, 0, sizeof(double) * nv, s));
    }
    // (the entry stage of the cycle rides on the x / r update of the loop)
    float e_jac = 0.f, *e_b8 = nullptr, *e_xa8 = nullptr;
    const float *e_dinv32 = nullptr;
    PADNE_TRY(amg_batch_entry_args(ctx, a, &e_jac, &e_dinv32, &e_b8, &e_xa8));
    // z in single precision, in z8's own memory, with x += alpha p on the p update (PADNE_PCG_P64=1: the loop of rounds 3-4,
    // z in double and x updated beside r); its unit is ||b_j||, from an initial guess ||r_0,j|| of the first start (see solve_one)
    const bool hat = !ctx->opt.pcg_p64;
    float *z32 = hat ? (float *)z8 : nullptr;
    const double *unit2 = (hat && x_is_guess) ? scal + C_NRM : scal + C_BB;
    int restarts = 0, total_iters = 0, code = PADNE_OK;
    double true_rr[8] = {0}, prev_true_rr[8] = {0}, bb[8] = {0}, tol2[8] = {0};
    int col_iters[8] = {0};
    bool stagnated[8] = {false};
    for (;;) {
        hipLaunchKernelGGL(pcg8_init_kernel<K>, dim3(gv), dim3(256), 0, s, n, b8, have_ax ? q8 : nullptr, r8, pslot(P_RR),
                           pslot(P_BB));
        PADNE_HIP_CHECK(hipGetLastError());
        PADNE_TRY(fold(pslot(P_RR), gv, scal + C_RR));
        PADNE_TRY(fold(pslot(P_BB), gv, scal + C_BB));
        if (hat && x_is_guess && restarts == 0)
            PADNE_HIP_CHECK(hipMemcpyAsync(scal + C_NRM, scal + C_RR, sizeof(double) * 8, hipMemcpyDeviceToDevice, s));
        PADNE_TRY(amg_apply_batch(ctx, a, K, r8, z8, pslot(P_RZ0), nullptr, unit2, false, z32));
        PADNE_TRY(fold(pslot(P_RZ0), gs, scal + C_RZ0));
        if (hat) {
            hipLaunchKernelGGL(p8_from_z32_kernel<K>, dim3(gv), dim3(256), 0, s, n, (const float *)z32, unit2, p8);
            PADNE_HIP_CHECK(hipGetLastError());
        } else {
            PADNE_HIP_CHECK(hipMemcpyAsync(p8, z8, sizeof(double) * nv, hipMemcpyDeviceToDevice, s));
        }
        hipLaunchKernelGGL(pcg8_set_tolerance_kernel, dim3(1), dim3(64), 0, s, st, scal + C_RR, scal + C_BB, o->rtol,
                           o->atol, restarts > 0 ? 1 : 0, K);
        PADNE_HIP_CHECK(hipGetLastError());
        int parity = 0;
        bool done = false;
        while (!done) {
            for (int k = 0; k < check_every; ++k) {
                double *rz_old = scal + (parity ? C_RZ1 : C_RZ0), *rz_new = scal + (parity ? C_RZ0 : C_RZ1);
                const int rz_new_slot = parity ? P_RZ0 : P_RZ1;
                if (hat) {
                    PADNE_TRY(launch_spmm_mode(ctx, a, K, SPMV_DOT, p8, q8, p8, pslot(P_PQ), &st->done, nullptr, nullptr, 0.0));
                    PADNE_TRY(fold(pslot(P_PQ), gs, scal + C_PQ));
                    hipLaunchKernelGGL(pcg8_update_r_entry_kernel<K>, dim3(gv), dim3(256), 0, s, n, rz_old, scal + C_PQ, q8, r8,
                                       pslot(P_RR), st, unit2, e_jac, e_dinv32, e_b8, e_xa8);
                    PADNE_HIP_CHECK(hipGetLastError());
                    PADNE_TRY(amg_apply_batch(ctx, a, K, r8, z8, pslot(rz_new_slot), &st->done, unit2, true, z32));
                    PADNE_TRY(fold(pslot(rz_new_slot), gs, rz_new));
                    PADNE_TRY(fold(pslot(P_RR), gv, scal + C_RR));
                    hipLaunchKernelGGL(pcg8_update_p_z_kernel<K>, dim3(gv), dim3(256), 0, s, n, rz_new, rz_old, scal + C_RR,
                                       scal + C_PQ, unit2, (const float *)z32, p8, x8, st, max_iter - total_iters);
                    PADNE_HIP_CHECK(hipGetLastError());
                } else {
                    PADNE_TRY(launch_spmm_mode(ctx, a, K, SPMV_DOT, p8, q8, p8, pslot(P_PQ), &st->done, nullptr, nullptr, 0.0));
                    PADNE_TRY(fold(pslot(P_PQ), gs, scal + C_PQ));
                    hipLaunchKernelGGL(pcg8_update_xr_kernel<K>, dim3(gv), dim3(256), 0, s, n, rz_old, scal + C_PQ, p8, q8, x8, r8,
                                       pslot(P_RR), st, (const double *)(scal + C_BB), e_jac, e_dinv32, e_b8, e_xa8);
                    PADNE_HIP_CHECK(hipGetLastError());
                    PADNE_TRY(amg_apply_batch(ctx, a, K, r8, z8, pslot(rz_new_slot), &st->done, scal + C_BB, true));
                    PADNE_TRY(fold(pslot(rz_new_slot), gs, rz_new));
                    PADNE_TRY(fold(pslot(P_RR), gv, scal + C_RR));
                    hipLaunchKernelGGL(pcg8_update_p_kernel<K>, dim3(gv), dim3(256), 0, s, n, rz_new, rz_old, scal + C_RR,
                                       scal + C_PQ, z8, p8, st, max_iter - total_iters);
                    PADNE_HIP_CHECK(hipGetLastError());
                }
                parity ^= 1;
            }
            PADNE_HIP_CHECK(hipMemcpyAsync(hst, st, sizeof(Pcg8Status), hipMemcpyDeviceToHost, s));
            PADNE_HIP_CHECK(hipStreamSynchronize(s));
            done = hst->done != 0;
        }
        total_iters += hst->iters;
        code = hst->code;
        for (int c = 0; c < K; ++c) {
            bb[c] = hst->bb[c];
            tol2[c] = hst->tol2[c];
            col_iters[c] += hst->col_iters[c];
        }
        // true residuals
        PADNE_TRY(launch_spmm_mode(ctx, a, K, SPMV_PLAIN, x8, q8, nullptr, nullptr, nullptr, nullptr, nullptr, 0.0));
        hipLaunchKernelGGL(residual8_kernel<K>, dim3(gv), dim3(256), 0, s, n, b8, q8, pslot(P_TMP));
        PADNE_HIP_CHECK(hipGetLastError());
        PADNE_TRY(fold(pslot(P_TMP), gv, scal + C_TRUE));
        double *hd = (double *)((char *)ctx->pinned + 1024);
        PADNE_HIP_CHECK(hipMemcpyAsync(hd, scal + C_TRUE, K * sizeof(double), hipMemcpyDeviceToHost, s));
        PADNE_HIP_CHECK(hipStreamSynchronize(s));
        bool all_final = true;
        for (int c = 0; c < K; ++c) {
            true_rr[c] = hd[c];
            if (true_rr[c] <= tol2[c] * 1.0000001) continue;
            if (restarts > 0 && true_rr[c] >= 0.25 * prev_true_rr[c]) {
                stagnated[c] = true;      // the evaluation floor of b - A x in binary64 (see solve_one)
                continue;
            }
            all_final = false;
        }
        if (code != PADNE_OK || all_final || total_iters >= max_iter || restarts >= 8) break;
        for (int c = 0; c < K; ++c) prev_true_rr[c] = true_rr[c];
        ++restarts;
        have_ax = true;
        PADNE_HIP_CHECK(hipMemsetAsync(st, 0, 4 * sizeof(int32_t) + 16 * sizeof(int32_t), s));   // flags and counters
    }
    PADNE_TRY(interleave(ctx, n, K, x8, x_cols, false));
    PADNE_HIP_CHECK(hipEventRecord(ctx->ev1, s));
    PADNE_HIP_CHECK(hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    PADNE_HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    if (info) {
        for (int c = 0; c < K; ++c) {
            info->iterations += col_iters[c];
            const double rel = bb[c] > 0 ? sqrt(true_rr[c] / bb[c]) : sqrt(true_rr[c]);
            if (rel > info->rel_residual) info->rel_residual = rel;
            if (sqrt(true_rr[c]) > info->abs_residual) info->abs_residual = sqrt(true_rr[c]);
            if (code == PADNE_OK && true_rr[c] > tol2[c] * 1.0000001 && info->status == PADNE_OK &&
                !(stagnated[c] && true_rr[c] <= 100.0 * tol2[c]))
                info->status = PADNE_E_NOTCONVERGED;
        }
        info->restarts += restarts;
        info->solve_seconds += ms * 1e-3;
        if (code != PADNE_OK) info->status = code;
    }
    return PADNE_OK;
}

// ---- largest eigenvalue of D^-1 A from the Lanczos coefficients of a few Jacobi-PCG steps ------------
// start of the Lanczos steps in one launch: r = b (pseudo-random, never stored), p = D^-1 r and their partial sums; the first
// workgroup clears the history and the status words, the exchange area behind p is cleared (four memsets and two kernels
// before)
__global__ __launch_bounds__(256) void lanczos_init_kernel(const long long n, const double *__restrict__ dinv,
                                                           double *__restrict__ r, double *__restrict__ p,
                                                           const long long n_tail, double *__restrict__ hist, const int n_hist,
                                                           PcgStatus *__restrict__ st, double *__restrict__ part_rz,
                                                           double *__restrict__ part_rr, double *__restrict__ part_bb) {
    __shared__ double red[4];
    if (blockIdx.x == 0) {
        for (int j = threadIdx.x; j < n_hist; j += 256) hist[j] = 0.0;
        if (threadIdx.x < (int)(sizeof(PcgStatus) / sizeof(int))) ((int *)st)[threadIdx.x] = 0;
    }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_tail; i += (long long)gridDim.x * 256) p[n + i] = 0.0;
    double rz = 0.0, rr = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        unsigned int h = (unsigned int)i * 2654435761u + 0x9e3779b9u;
        h ^= h >> 16;
        h *= 2246822519u;
        h ^= h >> 13;
        h *= 3266489917u;
        h ^= h >> 16;
        const double ri = (double)h * (2.0 / 4294967296.0) - 1.0;
        const double zi = dinv[i] * ri;
        r[i] = ri;
        p[i] = zi;
        rz += ri * zi;
        rr += ri * ri;
    }
    block_store_partial(rz, red, part_rz + blockIdx.x);
    block_store_partial(rr, red, part_rr + blockIdx.x);
    block_store_partial(rr, red, part_bb + blockIdx.x);
}

static double tridiag_max_eig(const std::vector<double> &d, const std::vector<double> &e) {
    // bisection on the Sturm count; e[k] couples d[k] and d[k+1]
    const int m = (int)d.size();
    double lo = d[0], hi = d[0];
    for (int k = 0; k < m; ++k) {
        const double r = (k > 0 ? fabs(e[k - 1]) : 0.0) + (k + 1 < m ? fabs(e[k]) : 0.0);
        lo = std::min(lo, d[k] - r);
        hi = std::max(hi, d[k] + r);
    }
    auto count_below = [&](double x) {   // number of eigenvalues < x
        int c = 0;
        double q = d[0] - x;
        if (q < 0) ++c;
        for (int k = 1; k < m; ++k) {
            const double den = (fabs(q) < 1e-300) ? 1e-300 : q;
            q = d[k] - x - e[k - 1] * e[k - 1] / den;
            if (q < 0) ++c;
        }
        return c;
    };
    for (int it = 0; it < 100; ++it) {
        const double mid = 0.5 * (lo + hi);
        if (count_below(mid) >= m) hi = mid; else lo = mid;
    }
    return 0.5 * (lo + hi);
}

// Enqueue `steps` Jacobi-PCG steps on a pseudo-random right-hand side; every step's scalars stay on the device
// (kernels consume them there) and the whole history is copied to `job->host` at the end of the queue -- no host
// synchronisation here, so a caller may queue this on the context's second stream and go on with other work.
int lanczos_enqueue(padne_ctx *ctx, const padne_csr *a, int steps, LanczosJob *job, const HaloPlan *plan) {
    const bool dist = plan != nullptr;
    const long long n = dist ? plan->n_owned : a->n_rows;   // owned unknowns
    const long long nc = dist ? a->n_cols : n;              // length of the vector the matrix multiplies
    const long long nr = a->n_rows;
    PADNE_REQUIRE(nr >= n && nc >= n, "operator shape");
    PADNE_REQUIRE(steps >= 1 && steps <= 60, "Lanczos steps");
    PADNE_TRY(csr_build_dinv(ctx, const_cast<padne_csr *>(a)));
    PADNE_TRY(ensure_workspace(ctx, sizeof(double) * (size_t)(3 * n + nc + nr) + 4096));
    double *r = (double *)ctx->ws, *x = r + n, *b = x + n, *p = b + n, *q = p + nc;
    PcgStatus *st = (PcgStatus *)ctx->status;
    hipStream_t s = ctx->stream;
    const int gv = vec_grid(n), gs = spmv_partials(a);
    job->ctx = ctx;
    job->steps = steps;
    job->host.assign((size_t)3 * steps + 4, 0.0);
    // on the second stream the history lands in pinned memory (a copy to pageable memory would hold the host until the
    // queue has drained, which is exactly what queuing there is meant to avoid): one 512-byte slot per job in flight,
    // kPinnedSlots of them (one per possible level of a hierarchy: the setup keeps every job pending until its end); a job
    // that finds no free slot copies to its own pageable buffer -- slower, never another job's data
    job->host_dst = job->host.data();
    job->pinned_slot = -1;
    if (ctx->is_aux && sizeof(double) * job->host.size() <= 512) {
        for (int k = 0; k < kPinnedSlots; ++k)
            if (!(ctx->pinned_busy & (1u << k))) {
                ctx->pinned_busy |= 1u << k;
                job->pinned_slot = k;
                job->host_dst = (double *)((char *)ctx->pinned + kPinnedSlotBase + 512 * k);
                break;
            }
    }
    // (rz, rr)[0..steps] interleaved | pq[0..steps)
    job->hist = (double *)pool_alloc(ctx, sizeof(double) * ((size_t)3 * steps + 4));
    if (job->hist == nullptr) return PADNE_E_NOMEM;
    double *hist = job->hist;
    double *H_rz = hist, *H_pq = hist + 2 * steps + 2;      // rz of step k at H_rz[2 k], its r.r right behind it
    static_assert(sizeof(PcgStatus) % sizeof(int) == 0 && sizeof(PcgStatus) / sizeof(int) <= 256, "cleared by one workgroup");
    hipLaunchKernelGGL(lanczos_init_kernel, dim3(gv), dim3(256), 0, s, n, (const double *)a->dinv, r, p, nc - n, hist,
                       3 * steps + 4, st, slot(ctx, SLOT_RZ0), slot(ctx, SLOT_RR), slot(ctx, SLOT_BB));
    PADNE_HIP_CHECK(hipGetLastError());
    (void)x;
    (void)b;
    auto fold = [&](const double *first_slot, int P, double *out) -> int {
        hipLaunchKernelGGL(fold_partials_kernel, dim3(1), dim3(256), 0, s, first_slot, P, kMaxPartials, 1, out);
        PADNE_HIP_CHECK(hipGetLastError());
        return PADNE_OK;
    };
    PADNE_TRY(fold(slot(ctx, SLOT_RZ0), gv, H_rz));
    if (dist) PADNE_TRY(comm_allreduce_sum_f64(ctx, H_rz, 1));
    int parity = 0;
    for (int k = 0; k < steps; ++k) {
        double *rz_old_part = slot(ctx, parity ? SLOT_RZ1 : SLOT_RZ0);
        double *rz_new_part = slot(ctx, parity ? SLOT_RZ0 : SLOT_RZ1);
        if (dist) PADNE_TRY(halo_exchange_plan(ctx, *plan, p, nullptr));
        PADNE_TRY(launch_spmv_mode(ctx, a, SPMV_DOT_AUX, p, q, p, slot(ctx, SLOT_PQ), nullptr, nullptr, nullptr, 0.0));
        if (dist) {
            PADNE_TRY(fold(slot(ctx, SLOT_PQ), gs, H_pq + k));
            PADNE_TRY(comm_allreduce_sum_f64(ctx, H_pq + k, 1));
        }
        // consumers read per-workgroup partials on one GPU and the reduced scalars across ranks
        const double *rz_old = dist ? H_rz + 2 * k : rz_old_part, *pq = dist ? H_pq + k : slot(ctx, SLOT_PQ);
        const int Pz = dist ? 1 : gv, Pq = dist ? 1 : gs;
        hipLaunchKernelGGL(pcg_update_xr_kernel, dim3(gv), dim3(256), 0, s, n, rz_old, Pz, pq, Pq, p, q, a->dinv,
                           (double *)nullptr, r, rz_new_part, slot(ctx, SLOT_RR), st);
        PADNE_HIP_CHECK(hipGetLastError());
        if (dist) {
            // r.z and r.r of the step travel in ONE all-reduce (two per Lanczos step in all, three before)
            PADNE_TRY(fold(rz_new_part, gv, H_rz + 2 * (k + 1)));
            PADNE_TRY(fold(slot(ctx, SLOT_RR), gv, H_rz + 2 * (k + 1) + 1));
            PADNE_TRY(comm_allreduce_sum_f64(ctx, H_rz + 2 * (k + 1), 2));
        }
        const double *rz_new = dist ? H_rz + 2 * (k + 1) : rz_new_part, *rr = dist ? H_rz + 2 * (k + 1) + 1 : slot(ctx, SLOT_RR);
        // (one GPU: the kernel leaves the step's scalars in the history itself)
        hipLaunchKernelGGL(pcg_update_p_kernel, dim3(gv), dim3(256), 0, s, n, rz_new, rz_old, Pz, rr, Pz, pq, Pq, r,
                           a->dinv, p, st, 1 << 30, dist ? (double *)nullptr : H_rz + 2 * (k + 1),
                           dist ? (double *)nullptr : H_pq + k);
        PADNE_HIP_CHECK(hipGetLastError());
        parity ^= 1;
    }
    PADNE_HIP_CHECK(hipMemcpyAsync(job->host_dst, hist, sizeof(double) * job->host.size(), hipMemcpyDeviceToHost, s));
    return PADNE_OK;
}

// Wait for the job's stream, then the largest Ritz value of the Lanczos tridiagonal.
int lanczos_finish(LanczosJob *job, double *lambda) {
    PADNE_REQUIRE(job->ctx != nullptr, "Lanczos job was not enqueued");
    const hipError_t e = hipStreamSynchronize(job->ctx->stream);
    pool_free(job->ctx, job->hist);
    job->hist = nullptr;
    if (job->pinned_slot >= 0) {
        if (e == hipSuccess) memcpy(job->host.data(), job->host_dst, sizeof(double) * job->host.size());
        job->ctx->pinned_busy &= ~(1u << job->pinned_slot);
        job->pinned_slot = -1;
        job->host_dst = job->host.data();
    }
    if (e != hipSuccess) {
        set_error("Lanczos estimate failed: %s", hipGetErrorString(e));
        return PADNE_E_HIP;
    }
    const int steps = job->steps;
    const std::vector<double> &hh = job->host;
    std::vector<double> alpha, beta;
    for (int k = 0; k < steps; ++k) {
        const double pqv = hh[(size_t)2 * steps + 2 + k], rzo = hh[(size_t)2 * k], rzn = hh[(size_t)2 * k + 2];
        if (!(pqv > 0.0) || !(rzo > 0.0)) break;      // also stops at the first NaN after a breakdown
        alpha.push_back(rzo / pqv);
        beta.push_back(rzn / rzo);
        if (!(rzn > 0.0) || rzn < 1e-30 * rzo) break;
    }
    const int m = (int)alpha.size();
    if (m == 0) {
        *lambda = 2.0;
        return PADNE_OK;
    }
    std::vector<double> d((size_t)m), e2((size_t)(m > 1 ? m - 1 : 0));
    for (int k = 0; k < m; ++k) {
        d[(size_t)k] = 1.0 / alpha[(size_t)k] + (k > 0 ? beta[(size_t)k - 1] / alpha[(size_t)k - 1] : 0.0);
        if (k + 1 < m) e2[(size_t)k] = sqrt(beta[(size_t)k]) / alpha[(size_t)k];
    }
    *lambda = tridiag_max_eig(d, e2);
    return PADNE_OK;
}

int estimate_lambda_max(padne_ctx *ctx, const padne_csr *a, int steps, double *lambda, const HaloPlan *plan) {
    LanczosJob job;
    const int rc = lanczos_enqueue(ctx, a, steps, &job, plan);
    if (rc != PADNE_OK) {
        if (job.hist != nullptr) {
            (void)hipStreamSynchronize(ctx->stream);
            pool_free(ctx, job.hist);
        }
        return rc;
    }
    return lanczos_finish(&job, lambda);
}

}  // namespace padne

using namespace padne;

// z = M^-1 r with the multigrid V-cycle (builds the hierarchy if needed); host vectors
extern "C" int padne_amg_apply(padne_ctx *ctx, padne_csr *a, const double *r_host, double *z_host) {
    PADNE_REQUIRE(ctx && a && r_host && z_host, "null argument");
    PADNE_REQUIRE(a->n_rows == a->n_cols && a->n_rows > 2048, "multigrid needs a square matrix with more than 2048 rows");
    PADNE_HIP_CHECK(hipSetDevice(ctx->device));
    PADNE_TRY(amg_setup(ctx, a));
    const size_t bytes = sizeof(double) * (size_t)a->n_rows;
    double *r = nullptr, *z = nullptr;
    PADNE_HIP_CHECK(hipMalloc((void **)&r, bytes));
    if (hipMalloc((void **)&z, bytes) != hipSuccess) {
        (void)hipFree(r);
        set_error("hipMalloc failed");
        return PADNE_E_NOMEM;
    }
    int rc = PADNE_OK;
    if (hipMemcpyAsync(r, r_host, bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = PADNE_E_HIP;
    if (rc == PADNE_OK) rc = amg_apply(ctx, a, r, z, slot(ctx, SLOT_TMP), nullptr);
    if (rc == PADNE_OK && (hipMemcpyAsync(z_host, z, bytes, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                           hipStreamSynchronize(ctx->stream) != hipSuccess))
        rc = PADNE_E_HIP;
    (void)hipFree(r);
    (void)hipFree(z);
    if (rc == PADNE_E_HIP) set_error("multigrid apply failed: %s", hipGetErrorString(hipGetLastError()));
    return rc;
}

// borrowed handle of a hierarchy matrix (which: 0 = A_l, 1 = P_l, 2 = R_l); valid while `a` lives
extern "C" int padne_amg_level(padne_ctx *ctx, padne_csr *a, int level, int which, const padne_csr **out) {
    PADNE_REQUIRE(ctx && a && out, "null argument");
    PADNE_TRY(amg_setup(ctx, a));
    *out = amg_level_matrix(a, level, which);
    if (*out == nullptr) {
        set_error("no such multigrid level / operator");
        return PADNE_E_INVALID;
    }
    return PADNE_OK;
}

// attach the owned x owned diagonal block used to build the multigrid preconditioner of a row-partitioned
// matrix (borrowed: the caller keeps both handles alive; pass null to detach)
extern "C" int padne_csr_set_preconditioner_block(padne_csr *a, padne_csr *block) {
    PADNE_REQUIRE(a != nullptr, "matrix");
    PADNE_REQUIRE(block == nullptr || block->n_rows == block->n_cols, "block must be square");
    a->prec_block = block;
    return PADNE_OK;
}

extern "C" int padne_ctx_set_halo(padne_ctx *ctx, int64_t n_owned, int32_t m, int32_t n_export,
                                  const int32_t *export_idx_host) {
    PADNE_REQUIRE(ctx, "ctx");
    PADNE_HIP_CHECK(hipSetDevice(ctx->device));
    PADNE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->halo_export) {
        PADNE_HIP_CHECK(hipFree(ctx->halo_export));
        ctx->halo_export = nullptr;
    }
    ctx->halo_on = false;
    if (n_owned < 0) return PADNE_OK;                 // negative n_owned clears the plan
    PADNE_REQUIRE(m >= 0 && n_export >= 0 && n_export <= m, "halo sizes");
    PADNE_REQUIRE(n_export == 0 || export_idx_host, "export list");
    for (int k = 0; k < n_export; ++k)
        PADNE_REQUIRE(export_idx_host[k] >= 0 && export_idx_host[k] < n_owned, "export index out of range");
    PADNE_HIP_CHECK(hipMalloc((void **)&ctx->halo_export, sizeof(int32_t) * (size_t)(n_export > 0 ? n_export : 1)));
    if (n_export > 0)
        PADNE_HIP_CHECK(hipMemcpy(ctx->halo_export, export_idx_host, sizeof(int32_t) * (size_t)n_export,
                                  hipMemcpyHostToDevice));
    ctx->halo_n_owned = n_owned;
    ctx->halo_m = m;
    ctx->halo_n_export = n_export;
    ctx->halo_on = true;
    return PADNE_OK;
}

static int solve_spd_dev_impl(padne_ctx *ctx, const padne_csr *a, const void *b_dev, void *x_dev, int32_t n_rhs,
                              const padne_solve_opts *opts, padne_solve_info *info);

extern "C" int padne_solve_spd_dev(padne_ctx *ctx, const padne_csr *a, const void *b_dev, void *x_dev,
                                   int32_t n_rhs, const padne_solve_opts *opts, padne_solve_info *info) {
    const int rc = solve_spd_dev_impl(ctx, a, b_dev, x_dev, n_rhs, opts, info);
    // "not converged" and "breakdown" are decided from globally reduced scalars, identically on every rank; any other
    // failure is local to this rank, whose peers would wait for it in their next collective
    if (rc != PADNE_OK && rc != PADNE_E_NOTCONVERGED && rc != PADNE_E_BREAKDOWN && ctx != nullptr) comm_abort(ctx);
    return rc;
}

static int solve_spd_dev_impl(padne_ctx *ctx, const padne_csr *a, const void *b_dev, void *x_dev, int32_t n_rhs,
                              const padne_solve_opts *opts, padne_solve_info *info) {
    PADNE_REQUIRE(ctx && a && b_dev && x_dev && opts, "null argument");
    PADNE_REQUIRE(n_rhs >= 1, "n_rhs");
    PADNE_REQUIRE(opts->precond == 0 || opts->precond == 1, "precond must be 0 (Jacobi) or 1 (multigrid)");
    PADNE_HIP_CHECK(hipSetDevice(ctx->device));
    if ((opts->flags & 4) != 0) {
        // "rebuild": everything derived from the matrix is recomputed inside this call, as for a matrix seen for the
        // first time -- the hierarchy (below), and also the single-precision copy and the x-window plan cached on it
        padne_csr *m = const_cast<padne_csr *>(a);
        padne_ctx *owner = m->owner ? m->owner : ctx;
        PADNE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        pool_free(owner, m->vals32);
        pool_free(owner, m->dinv32);
        pool_free(owner, m->xw_desc);
        pool_free(owner, m->xw_lidx);
        pool_free(owner, m->split_tiles);
        m->vals32 = nullptr;
        m->dinv32 = nullptr;
        m->xw_desc = nullptr;
        m->xw_lidx = nullptr;
        m->xw_state = 0;
        m->split_tiles = nullptr;
        m->split_state = 0;
    }
    PADNE_TRY(csr_build_dinv(ctx, const_cast<padne_csr *>(a)));
    PADNE_TRY(csr_build_xw_plan(ctx, const_cast<padne_csr *>(a)));
    if (ctx->halo_on) PADNE_TRY(csr_build_split_plan(ctx, const_cast<padne_csr *>(a), ctx->halo_n_owned));
    padne_solve_info local;
    memset(&local, 0, sizeof(local));
    local.n_rhs = n_rhs;
    // multigrid: on the matrix itself (one GPU: square; row-partitioned: this rank's rows of one global
    // hierarchy, see amg.hip), or block-Jacobi on the rank's diagonal block when one is attached with
    // padne_csr_set_preconditioner_block; small single-GPU systems stay with Jacobi
    padne_csr *pm = a->prec_block ? a->prec_block : const_cast<padne_csr *>(a);
    const long long n_owned = ctx->halo_on ? ctx->halo_n_owned : a->n_rows;
    const bool global_hierarchy = ctx->halo_on && a->prec_block == nullptr;
    PADNE_REQUIRE(!(opts->precond == 1 && global_hierarchy && a->n_rows != n_owned),
                  "row-partitioned multigrid needs a matrix with exactly the owned rows");
    // (systems of a few dozen unknowns stay with Jacobi; up to the size the dense inverse takes -- 2048 -- the "hierarchy" is that
    // inverse alone: an exact preconditioner, the loop ends after one or two iterations where Jacobi took hundreds)
    bool use_amg = opts->precond == 1 && (global_hierarchy || n_owned > kTinySystem);
    if (use_amg) {
        PADNE_REQUIRE(pm->n_rows == n_owned && (global_hierarchy || pm->n_cols == n_owned),
                      "preconditioner block must be owned x owned");
        const bool fresh = pm->amg == nullptr || (opts->flags & 4) != 0;
        if (fresh && pm->amg) {
            amg_destroy(pm->amg);
            pm->amg = nullptr;
        }
        const int rc_setup = amg_setup(ctx, pm);
        if (rc_setup == PADNE_E_NOCOARSEN) {
            // coarsening stalled on this matrix (decided identically on every rank): the solve proceeds with the
            // diagonal preconditioner (levels = 0).  Every other failure of the setup -- a shape check, a HIP error --
            // is an error of the call, not a reason to run 200x more iterations quietly.
            use_amg = false;
        } else if (rc_setup != PADNE_OK) {
            return rc_setup;
        } else {
            double setup_s = 0.0;
            amg_info(pm, &local.levels, &local.operator_complexity, &setup_s, nullptr);
            if (fresh) local.precond_setup_seconds = setup_s;
            if (local.levels < 2 && global_hierarchy) {      // (a row-partitioned hierarchy of one level: nothing to cycle over, use Jacobi)
                use_amg = false;
                local.levels = 0;
                local.operator_complexity = 0.0;
            }
        }
    }
    const long long n = ctx->halo_on ? ctx->halo_n_owned : a->n_rows;
    int k_first = 0;
    if (use_amg && !ctx->halo_on && !comm_active(ctx) && pm == a && amg_supports_batch8(a) &&
        !ctx->opt.no_batch) {
        // Groups of right-hand sides advance in lockstep (one pass over the operators per iteration for all of them); a
        // group whose cycle breaks down falls through to the one-at-a-time path below.  The kernels exist in widths 8, 4
        // and 2; MEASURED at N = 5 M (scripts/exp_lockstep_widths.py, profiles/r04_lockstep_widths.json) a lockstep solve
        // of 8 / 4 / 2 right-hand sides costs 5.3 / 3.3 / 2.3 single solves -- the single path has the x windows, the fused
        // W up-leg and the fused vector kernels, the lockstep cycle is the plain V(1,1) on interleaved vectors -- so:
        // eight at a time while they last, a remainder of 5-7 zero-padded to eight (zero columns are converged from the
        // start), exactly 4 (three regulators, solver.py:512-538) in width 4, and 2-3 one at a time, where lockstep LOSES
        // (2.3 against 2.0, 3.3 against 3.0).  PADNE_LOCKSTEP_NARROW=2 sends 2-3 through the narrow widths anyway (tests),
        // =0 switches width 4 off as well.
        auto solve_group = [&](const int width, const int first, const int count, bool *ok) -> int {
            *ok = false;
            Scratch pad(ctx);
            const double *bsrc = (const double *)b_dev + (size_t)first * n;
            double *xdst = (double *)x_dev + (size_t)first * n;
            double *bp = nullptr, *xp = nullptr;
            if (count < width) {
                PADNE_TRY(pad.alloc(&bp, (size_t)width * n));
                PADNE_TRY(pad.alloc(&xp, (size_t)width * n));
                PADNE_HIP_CHECK(hipMemsetAsync(bp, 0, sizeof(double) * (size_t)width * n, ctx->stream));
                PADNE_HIP_CHECK(hipMemsetAsync(xp, 0, sizeof(double) * (size_t)width * n, ctx->stream));
                PADNE_HIP_CHECK(hipMemcpyAsync(bp, bsrc, sizeof(double) * (size_t)count * n, hipMemcpyDeviceToDevice, ctx->stream));
                if ((opts->flags & 1) != 0)
                    PADNE_HIP_CHECK(hipMemcpyAsync(xp, xdst, sizeof(double) * (size_t)count * n, hipMemcpyDeviceToDevice, ctx->stream));
            }
            padne_solve_info grp = local;
            grp.status = PADNE_OK;
            const double *bb = count < width ? bp : bsrc;
            double *xx = count < width ? xp : xdst;
            const bool guess = (opts->flags & 1) != 0;
            if (width == 8) PADNE_TRY(solve_batch<8>(ctx, a, bb, xx, opts, &grp, guess));
            else if (width == 4) PADNE_TRY(solve_batch<4>(ctx, a, bb, xx, opts, &grp, guess));
            else PADNE_TRY(solve_batch<2>(ctx, a, bb, xx, opts, &grp, guess));
            if (grp.status != PADNE_OK) return PADNE_OK;             // (not ok: the caller solves these one at a time)
            if (count < width) {
                PADNE_HIP_CHECK(hipMemcpyAsync(xdst, xp, sizeof(double) * (size_t)count * n, hipMemcpyDeviceToDevice, ctx->stream));
                PADNE_HIP_CHECK(hipStreamSynchronize(ctx->stream));   // (the padded copies go back to the pool)
            }
            const int keep = local.status;
            local = grp;
            local.status = keep;
            ++ctx->lockstep_groups;
            *ok = true;
            return PADNE_OK;
        };
        const int narrow = ctx->opt.lockstep_narrow < 0 ? 1 : ctx->opt.lockstep_narrow;
        bool ok = true;
        while (ok && n_rhs - k_first >= 2) {
            const int rest = n_rhs - k_first;
            int width = 0, count = 0;
            if (rest >= 8) { width = 8; count = 8; }
            else if (rest >= 5) { width = 8; count = rest; }
            else if (rest == 4 && narrow >= 1) { width = 4; count = 4; }
            else if (rest == 3 && narrow >= 2) { width = 4; count = 3; }
            else if (rest == 2 && narrow >= 2) { width = 2; count = 2; }
            else break;
            PADNE_TRY(solve_group(width, k_first, count, &ok));
            if (ok) k_first += count;
        }
    }
    // row-partitioned multigrid runs use the single-reduction loop (one all-reduce per iteration); PADNE_CG_SINGLE_REDUCTION
    // = 1 / 0 forces it on (also on one GPU, for tests) or off
    const bool dist_run = comm_active(ctx);
    const bool single_reduction = use_amg && (dist_run || local.levels >= 2) &&      // (a hierarchy of one level: the plain loop)
                                  (ctx->opt.cg_single_reduction >= 0 ? ctx->opt.cg_single_reduction != 0 : dist_run);
    for (int k = k_first; k < n_rhs; ++k) {
        const int status_before = local.status;
        if (single_reduction)
            PADNE_TRY(solve_one_single_reduction(ctx, a, pm, (const double *)b_dev + (size_t)k * n,
                                                 (double *)x_dev + (size_t)k * n, opts, &local, (opts->flags & 1) != 0));
        else
            PADNE_TRY(solve_one(ctx, a, use_amg ? pm : nullptr, (const double *)b_dev + (size_t)k * n,
                                (double *)x_dev + (size_t)k * n, opts, &local, (opts->flags & 1) != 0));
        // (a solve that stalled at the binary64 floor of b - A x is not redone: no preconditioner gets below it)
        if (use_amg && local.status != PADNE_OK && status_before == PADNE_OK &&
            !(local.status == PADNE_E_NOTCONVERGED && t_last_solve_stagnated)) {
            // the V-cycle lost positive definiteness or stalled far from the tolerance on this system:
            // redo this right-hand side from scratch with the diagonal preconditioner
            padne_solve_opts retry = *opts;
            retry.flags &= ~1;
            local.status = PADNE_OK;
            local.rel_residual = 0.0;
            local.abs_residual = 0.0;
            local.precond_fallbacks += 1;
            PADNE_TRY(solve_one(ctx, a, nullptr, (const double *)b_dev + (size_t)k * n,
                                (double *)x_dev + (size_t)k * n, &retry, &local, false));
        }
    }
    if (info) *info = local;
    if (local.status == PADNE_E_BREAKDOWN) {
        set_error("PCG breakdown: matrix is not symmetric positive definite or contains NaN");
        return PADNE_E_BREAKDOWN;
    }
    if (local.status == PADNE_E_NOTCONVERGED) {
        set_error("PCG did not reach the tolerance in %d iterations (rel. residual %.3e)", local.iterations,
                  local.rel_residual);
        return PADNE_E_NOTCONVERGED;
    }
    return PADNE_OK;
}

extern "C" int padne_solve_spd(padne_ctx *ctx, const padne_csr *a, const double *b_host, double *x_host,
                               int32_t n_rhs, const padne_solve_opts *opts, padne_solve_info *info) {
    PADNE_REQUIRE(ctx && a && b_host && x_host && opts, "null argument");
    PADNE_REQUIRE(n_rhs >= 1, "n_rhs");
    PADNE_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t bytes = sizeof(double) * (size_t)(ctx->halo_on ? ctx->halo_n_owned : a->n_rows) * (size_t)n_rhs;
    // (from the context's cache: hipMalloc / hipFree of two 80 MB vectors cost several milliseconds per call at 10 M unknowns)
    double *b = (double *)pool_alloc(ctx, bytes ? bytes : 8), *x = (double *)pool_alloc(ctx, bytes ? bytes : 8);
    if (b == nullptr || x == nullptr) {
        pool_free(ctx, b);
        pool_free(ctx, x);
        return PADNE_E_NOMEM;
    }
    int rc = PADNE_OK;
    if (hipMemcpyAsync(b, b_host, bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = PADNE_E_HIP;
    if (rc == PADNE_OK && (opts->flags & 1) &&
        hipMemcpyAsync(x, x_host, bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        rc = PADNE_E_HIP;
    if (rc == PADNE_OK) rc = padne_solve_spd_dev(ctx, a, b, x, n_rhs, opts, info);
    if (rc == PADNE_OK || rc == PADNE_E_NOTCONVERGED) {
        if (hipMemcpyAsync(x_host, x, bytes, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) {
            set_error("copy-back of the solution failed");
            rc = PADNE_E_HIP;
        }
    }
    if (rc != PADNE_OK && rc != PADNE_E_NOTCONVERGED) (void)hipStreamSynchronize(ctx->stream);
    pool_free(ctx, b);
    pool_free(ctx, x);
    return rc;
}

// test introspection (include/padne_hip_test.h): groups of right-hand sides this context has solved in lockstep so far
extern "C" int padne_ctx_lockstep_groups(const padne_ctx *ctx, int64_t *groups) {
    PADNE_REQUIRE(ctx && groups, "null argument");
    *groups = (int64_t)ctx->lockstep_groups;
    return PADNE_OK;
}
