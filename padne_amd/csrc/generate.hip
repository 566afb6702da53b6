// Structured triangulations generated on the device: the synthetic inputs of BASELINE.json's configs C2-C5 (jittered
// nx * ny grids, cells split by alternating diagonals) without a host array or a PCIe transfer.  Stands where the
// reference's CGAL mesher stands (padne/mesh.py:662-795, padne/cpp/_cgal.cpp -- out of scope): it only produces the
// xy / triangle arrays the assembly consumes.  Bit-identical to padne_amd.synthetic.jittered_grid, including the
// jitter: numpy's default_rng(seed).uniform(-j h, +j h, (ny, nx, 2)) is PCG64 (XSL-RR 128/64), one 64-bit output per
// double, so vertex (ix, iy) needs outputs 2 (iy nx + ix) + 1 and + 2 of the stream, reached with the O(log k)
// jump-ahead of a linear congruential generator.  The caller passes the generator's 128-bit state and increment
// (numpy: default_rng(seed).bit_generator.state), so the seeding itself stays numpy's.
#include "common.hpp"

namespace padne {

typedef unsigned __int128 u128;

__device__ __forceinline__ u128 make128(unsigned long long hi, unsigned long long lo) { return ((u128)hi << 64) | (u128)lo; }

// state after `delta` steps of s -> s * mult + inc  (Brown, "Random number generation with arbitrary strides")
__device__ __forceinline__ u128 lcg_advance(u128 state, unsigned long long delta, u128 mult, u128 inc) {
    u128 acc_mult = 1, acc_plus = 0, cur_mult = mult, cur_plus = inc;
    while (delta > 0) {
        if (delta & 1ull) {
            acc_mult *= cur_mult;
            acc_plus = acc_plus * cur_mult + cur_plus;
        }
        cur_plus = (cur_mult + 1) * cur_plus;
        cur_mult *= cur_mult;
        delta >>= 1;
    }
    return acc_mult * state + acc_plus;
}

__device__ __forceinline__ double pcg64_double(u128 state) {      // output function of numpy's PCG64, then next_double
    const unsigned long long hi = (unsigned long long)(state >> 64), lo = (unsigned long long)state;
    const unsigned long long x = hi ^ lo;
    const unsigned rot = (unsigned)(hi >> 58);
    const unsigned long long out = (x >> rot) | (x << ((64u - rot) & 63u));
    return (double)(out >> 11) * (1.0 / 9007199254740992.0);
}

__global__ void grid_vertices_kernel(long long nx, long long ny, double h, double jitter, double ox, double oy,
                                     unsigned long long s_hi, unsigned long long s_lo, unsigned long long i_hi,
                                     unsigned long long i_lo, int jittered, double *__restrict__ xy) {
    const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nx * ny) return;
    const long long ix = v % nx, iy = v / nx;
    double x = ox + (double)ix * h, y = oy + (double)iy * h;
    if (jittered && ix > 0 && iy > 0 && ix < nx - 1 && iy < ny - 1) {      // border vertices stay on the lattice
        const u128 mult = make128(0x2360ED051FC65DA4ull, 0x4385DF649FCCF645ull);
        const u128 inc = make128(i_hi, i_lo);
        u128 st = lcg_advance(make128(s_hi, s_lo), 2ull * (unsigned long long)v + 1ull, mult, inc);
        const double lo = -jitter * h, scale = jitter * h - lo;             // numpy: low + (high - low) * u
        x += lo + scale * pcg64_double(st);
        st = st * mult + inc;
        y += lo + scale * pcg64_double(st);
    }
    xy[2 * v] = x;
    xy[2 * v + 1] = y;
}

// two counter-clockwise triangles per cell, diagonal v00-v11 in even cells, v10-v01 in odd ones
__global__ void grid_triangles_kernel(long long nx, long long ny, int *__restrict__ tri) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (nx - 1) * (ny - 1)) return;
    const long long cx = c % (nx - 1), cy = c / (nx - 1);
    const int v00 = (int)(cy * nx + cx), v10 = v00 + 1, v01 = v00 + (int)nx, v11 = v01 + 1;
    const bool even = ((cx + cy) & 1) == 0;
    int *t = tri + 6 * c;
    t[0] = v00;
    t[1] = v10;
    t[2] = even ? v11 : v01;
    t[3] = even ? v00 : v10;
    t[4] = v11;
    t[5] = v01;
}

}  // namespace padne

using namespace padne;

extern "C" int padne_generate_grid_mesh(padne_ctx *ctx, int64_t nx, int64_t ny, double h, double jitter, double origin_x,
                                        double origin_y, const uint64_t *pcg64_state_inc, void *xy_dev, void *tri_dev) {
    PADNE_REQUIRE(ctx && xy_dev && tri_dev, "null argument");
    PADNE_REQUIRE(nx >= 2 && ny >= 2 && nx * ny < 2147483647LL, "grid needs at least 2 x 2 and fewer than 2^31 vertices");
    PADNE_REQUIRE(jitter == 0.0 || pcg64_state_inc != nullptr, "a jittered grid needs the generator state");
    PADNE_HIP_CHECK(hipSetDevice(ctx->device));
    const uint64_t zero[4] = {0, 0, 0, 0};
    const uint64_t *g = pcg64_state_inc ? pcg64_state_inc : zero;
    hipLaunchKernelGGL(grid_vertices_kernel, dim3(nblk(nx * ny)), dim3(256), 0, ctx->stream, (long long)nx, (long long)ny, h,
                       jitter, origin_x, origin_y, (unsigned long long)g[0], (unsigned long long)g[1],
                       (unsigned long long)g[2], (unsigned long long)g[3], jitter != 0.0 ? 1 : 0, (double *)xy_dev);
    hipLaunchKernelGGL(grid_triangles_kernel, dim3(nblk((nx - 1) * (ny - 1))), dim3(256), 0, ctx->stream, (long long)nx,
                       (long long)ny, (int *)tri_dev);
    PADNE_HIP_CHECK(hipGetLastError());
    return PADNE_OK;
}
