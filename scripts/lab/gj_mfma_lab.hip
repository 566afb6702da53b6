// Lab: blocked Gauss-Jordan inversion with 64 pivots per launch, trailing update on the f64 matrix cores
// (v_mfma_f64_16x16x4_f64).  Standalone: hipcc --offload-arch=gfx950 -O3 -o gj_mfma_lab gj_mfma_lab.hip ; ./gj_mfma_lab [n]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int B = 64;            // pivots per step
constexpr int TR = 128, TC = 32; // rows x columns of a wave's tile
constexpr int NCB = TC / 16;

// Dinv in fragment order: element (a, b) at ((a >> 4) * 16 + (b >> 2)) * 64 + (b & 3) * 16 + (a & 15)
__device__ __host__ inline int frag_off(int a, int b) { return ((a >> 4) * 16 + (b >> 2)) * 64 + (b & 3) * 16 + (a & 15); }

__device__ __host__ inline int gj_side_off(int a, int b) {      // Dinv[a][b] -> position in the side buffer
    const int blk = a >> 4, w = a & 15, i = ((w & 3) << 2) | (w >> 2);      // pi^-1: pivot w = 4 (i & 3) + (i >> 2)
    return ((blk * 16 + (b >> 2)) * 64 + (b & 3) * 16 + i);
}

// naive: one workgroup inverts the bs x bs pivot block at k0 (padded with the identity to 64 x 64) -> side (fragment order)
__global__ __launch_bounds__(256) void pivot_block_naive(int n, int k0, int bs, const double *__restrict__ in, double *__restrict__ side) {
    __shared__ double D[B][B + 1];
    const int t = threadIdx.x;
    for (int e = t; e < B * B; e += 256) {
        const int a = e / B, b = e % B;
        D[a][b] = (a < bs && b < bs) ? in[(size_t)(k0 + a) * n + k0 + b] : (a == b ? 1.0 : 0.0);
    }
    __syncthreads();
    for (int p = 0; p < B; ++p) {
        const double inv = 1.0 / D[p][p];
        __syncthreads();
        double newv[16];
        for (int h = 0; h < 16; ++h) {
            const int e = t + 256 * h, a = e / B, b = e % B;
            const double rp = D[p][b] * inv, cp = D[a][p];
            newv[h] = (a == p) ? (b == p ? inv : rp) : (b == p ? -cp * inv : D[a][b] - cp * rp);
        }
        __syncthreads();
        for (int h = 0; h < 16; ++h) {
            const int e = t + 256 * h;
            D[e / B][e % B] = newv[h];
        }
        __syncthreads();
    }
    for (int e = t; e < B * B; e += 256) side[gj_side_off(e / B, e % B)] = D[e / B][e % B];
}

// one wave per tile of TR rows x TC columns; W' = C~ - L~ (Dinv P~).
// Order of the 64 pivots inside the products: the A operand of the update (the tile's rows of the pivot columns, contiguous
// in memory along k) is read 16 bytes per lane, lane (i, g) taking k = 16 q + 4 g + {0..3}; the k-step (q, m) of the
// matrix-core instruction therefore pairs lane group g with pivot 16 q + 4 g + m, and U must come out of the first product
// with row 4 g + m of block q in register m of lane group g: its A operand (Dinv) has its rows permuted, row i of a block
// standing for pivot 4 (i & 3) + (i >> 2) (side buffer: gj_side_off).
struct __attribute__((packed, aligned(4))) D2u { double x, y; };
template <int MODE>
__global__ __launch_bounds__(64) void gj64_update(int n, int k0, int bs, const double *__restrict__ in, double *__restrict__ out,
                                                  const double *__restrict__ side) {
    const int lane = threadIdx.x, j = lane & 15, g = lane >> 4;
    const int c0 = blockIdx.x * TC, r0 = blockIdx.y * TR;
    constexpr int NRB = TR / 16;
    // A operand of the update for one row block: 16 doubles per lane (4 quads of 4 consecutive pivots)
    auto load_a = [&](int rb, double (&a)[16]) {
        const int row = r0 + 16 * rb + j;
        const bool prow = row >= k0 && row < k0 + B;
        const double *p = in + (size_t)row * n + k0 + 4 * g;
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
                    if (prow) v = (row - k0 == k) ? 1.0 : 0.0;
                    else if (row < n && k < bs) v = -p[16 * q + m];
                    a[4 * q + m] = v;
                }
            }
        }
    };
    auto load_c = [&](int rb, v4d (&C)[NCB]) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = r0 + 16 * rb + g + 4 * r, col = c0 + 16 * cb + j;
                const bool piv = (row >= k0 && row < k0 + B) || (col >= k0 && col < k0 + B);
                C[cb][r] = (row < n && col < n && !piv) ? in[(size_t)row * n + col] : 0.0;
            }
    };
    double a_cur[16], a_nxt[16];
    v4d C_cur[NCB], C_nxt[NCB];
    load_c(0, C_cur);
    load_a(0, a_cur);
    // ---- U = Dinv P~ for the tile's columns: 4 pivot blocks x 2 column blocks
    v4d U[4][NCB];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) U[kb][cb] = (v4d){0.0, 0.0, 0.0, 0.0};
    double pb[16][NCB];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const int col = c0 + 16 * cb + j, k = 4 * ks + g;
            double v = 0.0;
            if (col >= k0 && col < k0 + B) v = (col - k0 == k) ? 1.0 : 0.0;
            else if (col < n && k < bs) v = in[(size_t)(k0 + k) * n + col];
            pb[ks][cb] = v;
        }
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const double a = side[(kb * 16 + ks) * 64 + lane];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) { if (MODE & 2) U[kb][cb][0] += a * pb[ks][cb]; else U[kb][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, pb[ks][cb], U[kb][cb], 0, 0, 0); }
        }
    // ---- the tile: C~ - L~ U, a row block at a time, the next block's operands in flight
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
                for (int cb = 0; cb < NCB; ++cb) { if (MODE & 1) C_cur[cb][0] += a_cur[4 * q + m] * U[q][cb][m]; else C_cur[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[4 * q + m], U[q][cb][m], C_cur[cb], 0, 0, 0); }
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = r0 + 16 * rb + g + 4 * r, col = c0 + 16 * cb + j;
                if (row < n && col < n) out[(size_t)row * n + col] = C_cur[cb][r];
            }
        if (rb + 1 < NRB) {
#pragma unroll
            for (int e = 0; e < 16; ++e) a_cur[e] = a_nxt[e];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) C_cur[cb] = C_nxt[cb];
        }
    }
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1617;
    std::vector<double> A((size_t)n * n, 0.0);
    srand(5);
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < 24; ++k) {
            const int jx = (k < 6) ? (i + k + 1) % n : rand() % n;
            if (jx == i) continue;
            const double v = -(0.1 + (rand() % 1000) / 1000.0);
            A[(size_t)i * n + jx] += v;
            A[(size_t)jx * n + i] += v;
        }
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int jx = 0; jx < n; ++jx) if (jx != i) s += fabs(A[(size_t)i * n + jx]);
        A[(size_t)i * n + i] = s * 1.0001 + 1e-3;
    }
    double *d0, *d1, *side;
    CHECK(hipMalloc(&d0, sizeof(double) * n * n));
    CHECK(hipMalloc(&d1, sizeof(double) * n * n));
    CHECK(hipMalloc(&side, sizeof(double) * B * B));
    CHECK(hipMemcpy(d0, A.data(), sizeof(double) * n * n, hipMemcpyHostToDevice));
    const dim3 grid((n + TC - 1) / TC, (n + TR - 1) / TR);
    double *src = d0, *dst = d1;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemcpy(d0, A.data(), sizeof(double) * n * n, hipMemcpyHostToDevice));
        src = d0; dst = d1;
        CHECK(hipEventRecord(e0));
        for (int k0 = 0; k0 < n; k0 += B) {
            const int bs = n - k0 < B ? n - k0 : B;
            hipLaunchKernelGGL(pivot_block_naive, dim3(1), dim3(256), 0, 0, n, k0, bs, src, side);
            hipLaunchKernelGGL(gj64_update<0>, grid, dim3(64), 0, 0, n, k0, bs, src, dst, side);
            std::swap(src, dst);
        }
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("full inversion (naive pivot blocks in front): %.3f ms\n", ms);
    }
    std::vector<double> inv((size_t)n * n);
    CHECK(hipMemcpy(inv.data(), src, sizeof(double) * n * n, hipMemcpyDeviceToHost));
    // check: inv (A x) = x for a few x
    double worst = 0.0;
    for (int trial = 0; trial < 3; ++trial) {
        std::vector<double> x(n), y(n, 0.0), z(n, 0.0);
        for (int i = 0; i < n; ++i) x[i] = (rand() % 2001 - 1000) / 1000.0;
        for (int i = 0; i < n; ++i) { double s = 0; for (int jx = 0; jx < n; ++jx) s += A[(size_t)i * n + jx] * x[jx]; y[i] = s; }
        for (int i = 0; i < n; ++i) { double s = 0; for (int jx = 0; jx < n; ++jx) s += inv[(size_t)i * n + jx] * y[jx]; z[i] = s; }
        for (int i = 0; i < n; ++i) worst = fmax(worst, fabs(z[i] - x[i]));
    }
    printf("n = %d: max |inv (A x) - x| = %.3e\n", n, worst);
    // timing of the update kernel alone (fixed side)
    const int steps = (n + B - 1) / B;
    auto time_mode = [&](int mode) {
        CHECK(hipEventRecord(e0));
        for (int rep = 0; rep < 10; ++rep)
            for (int s = 0; s < steps; ++s) {
                const int k0 = s * B, bs = n - k0 < B ? n - k0 : B;
                if (mode == 0) hipLaunchKernelGGL(gj64_update<0>, grid, dim3(64), 0, 0, n, k0, bs, src, dst, side);
                if (mode == 1) hipLaunchKernelGGL(gj64_update<1>, grid, dim3(64), 0, 0, n, k0, bs, src, dst, side);
                if (mode == 2) hipLaunchKernelGGL(gj64_update<2>, grid, dim3(64), 0, 0, n, k0, bs, src, dst, side);
                if (mode == 3) hipLaunchKernelGGL(gj64_update<3>, grid, dim3(64), 0, 0, n, k0, bs, src, dst, side);
                std::swap(src, dst);
            }
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("mode %d (1: no update MFMA, 2: no U MFMA): %d launches per inversion, %.2f us per launch, %.3f ms per inversion\n", mode, steps, 1e3 * ms / (10 * steps), ms / 10);
    };
    for (int mode = 0; mode < 4; ++mode) time_mode(mode);
    return worst < 1e-6 ? 0 : 2;
}
