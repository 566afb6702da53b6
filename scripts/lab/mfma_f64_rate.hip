// Lab: issue rate and dependent latency of v_mfma_f64_16x16x4_f64 on gfx950.  One wave per SIMD (grid 1024 x 64), NACC independent
// accumulator chains per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double v4d __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(64) void chains(int iters, const double *__restrict__ in, double *__restrict__ out) {
    const double a = in[threadIdx.x], b = in[64 + threadIdx.x];
    v4d acc[NACC];
#pragma unroll
    for (int c = 0; c < NACC; ++c) acc[c] = (v4d){0.0, 0.0, 0.0, (double)c};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < NACC; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < NACC; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int NACC>
static void run(int waves_per_simd, const double *in, double *out) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int total = 16384;                      // MFMAs per wave
    const int iters = total / NACC;
    hipLaunchKernelGGL(chains<NACC>, dim3(1024 * waves_per_simd), dim3(64), 0, 0, iters, in, out);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(chains<NACC>, dim3(1024 * waves_per_simd), dim3(64), 0, 0, iters, in, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double per = 1e6 * ms / total / waves_per_simd;      // ns per MFMA per SIMD
    printf("chains %d, waves/SIMD %d: %.1f ns per instruction and SIMD (%.0f cycles at 2.4 GHz), %.1f TFLOP/s\n", NACC, waves_per_simd, per,
           per * 2.4, 2048.0 * total * 1024 * waves_per_simd / (ms * 1e-3) * 1e-12);
}
int main() {
    double *in, *out;
    CHECK(hipMalloc(&in, 1024));
    CHECK(hipMalloc(&out, sizeof(double) * 64 * 4096));
    CHECK(hipMemset(in, 0, 1024));
    run<1>(1, in, out); run<2>(1, in, out); run<4>(1, in, out); run<8>(1, in, out); run<16>(1, in, out);
    run<1>(2, in, out); run<4>(2, in, out); run<8>(2, in, out);
    return 0;
}
