// How much fp32 VALU work fits between f16 MFMAs of the SAME wave for free?
// One wave per SIMD (256-thread workgroups, 256 of them), 32 x v_mfma_f32_32x32x16_f16 per iteration with
// N filler instructions of one kind after every MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int N>   // KIND 0: v_fma_f32, 1: v_pk_fma_f32, 2: v_exp_f32, 3: v_cvt_pk_f16_f32
__device__ __forceinline__ void fill(float (&x)[8], f32x2 (&p)[8], float a, float b) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (KIND == 0) x[i & 7] = fmaf(x[i & 7], b, a);
        if (KIND == 1) p[i & 7] = p[i & 7] * f32x2{b, b} + f32x2{a, a};
        if (KIND == 2) x[i & 7] = __builtin_amdgcn_exp2f(x[i & 7]);
        if (KIND == 3) {
            typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
            f16x2 hh = __builtin_convertvector(p[i & 7], f16x2);
            p[i & 7] = p[i & 7] + __builtin_convertvector(hh, f32x2);
        }
    }
}

template <int KIND, int N, bool MFMA>
__global__ __launch_bounds__(256) void k(int iters, float *out) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    f16x8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(a + i); hb[i] = (_Float16)(b + i); }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float x[8];
    f32x2 p[8];
    for (int i = 0; i < 8; ++i) { x[i] = a + i; p[i] = f32x2{a + i, a}; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MFMA) c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c0, 0, 0, 0);
            fill<KIND, N>(x, p, a, b);
            if (MFMA) c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c1, 0, 0, 0);
            fill<KIND, N>(x, p, a, b);
            if (MFMA) c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c2, 0, 0, 0);
            fill<KIND, N>(x, p, a, b);
            if (MFMA) c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c3, 0, 0, 0);
            fill<KIND, N>(x, p, a, b);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = c0[0] + c1[1] + c2[2] + c3[3];
    for (int i = 0; i < 8; ++i) s += x[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND, int N, bool MFMA>
float run(float *out, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, N, MFMA>), dim3(256), dim3(256), 0, 0, 10, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, N, MFMA>), dim3(256), dim3(256), 0, 0, iters, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / iters;   // us per iteration (32 MFMAs)
}

#define ROW(KIND, N, NAME) \
    printf("%-14s x%2d per MFMA: with MFMA %7.3f us/iter   fillers alone %7.3f us/iter\n", NAME, N, \
           run<KIND, N, true>(out, iters), run<KIND, N, false>(out, iters));

int main() {
    float *out;
    (void)hipMalloc(&out, 256 * 256 * 4);
    const int iters = 2000;
    printf("MFMA only: %7.3f us/iter (32 MFMAs)\n", run<0, 0, true>(out, iters));
    ROW(0, 2, "v_fma_f32") ROW(0, 4, "v_fma_f32") ROW(0, 6, "v_fma_f32") ROW(0, 8, "v_fma_f32") ROW(0, 12, "v_fma_f32")
    ROW(1, 2, "v_pk_fma_f32") ROW(1, 4, "v_pk_fma_f32") ROW(1, 6, "v_pk_fma_f32")
    ROW(2, 1, "v_exp_f32") ROW(2, 2, "v_exp_f32") ROW(2, 4, "v_exp_f32")
    ROW(3, 2, "cvt_pk+cvt+add") ROW(3, 4, "cvt_pk+cvt+add")
    return 0;
}
