// Issue rate of fp32 VALU forms on one SIMD: v_fma_f32 vs v_pk_fma_f32 (two fp32 per lane) vs v_pk_mul/add,
// 16 independent chains, 1 or 2 waves per SIMD.  Reports ns per instruction per SIMD and implied TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(int iters, float *out) {
    const float a = threadIdx.x * 1e-3f + 1.0f, b = 0.999f;
    float r = 0.f;
    if (MODE == 0) {
        float x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = a + i;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaf(x[i], b, a);
#pragma unroll
        for (int i = 0; i < 16; ++i) r += x[i];
    } else {
        f32x2 x[16];
        const f32x2 bb = {b, b}, aa = {a, a};
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = f32x2{a + i, a - i};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (MODE == 1) x[i] = __builtin_elementwise_fma(x[i], bb, aa);
                    if (MODE == 2) x[i] = x[i] * bb;
                    if (MODE == 3) x[i] = x[i] + aa;
                }
#pragma unroll
        for (int i = 0; i < 16; ++i) r += x[i].x + x[i].y;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
void run(const char *name, int threads, float *out) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<256, threads>>>(100, out);
    (void)hipEventRecord(e0);
    k<MODE><<<256, threads>>>(iters, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double n_inst = iters * 64.0, waves = threads / 256.0;      // per wave
    const double ns = ms * 1e6 / (n_inst * waves);
    const double flop = (MODE == 0 ? 2 : (MODE == 1 ? 4 : 2)) * 64.0;  // per wave instruction
    printf("%-14s %d wave(s)/SIMD: %.2f ns per instruction per SIMD -> %.1f TFLOP/s chip\n", name, threads / 256, ns,
           flop / ns * 1e9 * 1024 / 1e12);
}

int main() {
    float *out;
    (void)hipMalloc(&out, 256 * 512 * 4);
    for (int threads : {256, 512}) {
        run<0>("v_fma_f32", threads, out);
        run<1>("v_pk_fma_f32", threads, out);
        run<2>("v_pk_mul_f32", threads, out);
        run<3>("v_pk_add_f32", threads, out);
    }
    return 0;
}
