// Clean co-issue measurements (everything compile-time, no branches between instructions).
//   A: one wave per SIMD runs  { 1 f16 MFMA (32x32x16) ; NV fp32 VALU ops } x N   -> ns per MFMA slot vs NV
//   B: two waves per SIMD, both run A (what the edge kernels do)
//   C: two waves per SIMD, one only MFMAs, the other only VALU (specialised waves)
// VALU ops are independent chains (8 accumulators), KIND 0 = v_fma_f32, 1 = v_pk_fma_f32, 2 = v_exp_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NV, int KIND>
__device__ inline void valu(float (&x)[8], f32x2 (&p)[8], float a, float b) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (KIND == 0) x[i & 7] = __builtin_fmaf(x[i & 7], b, a);
        if (KIND == 1) p[i & 7] = __builtin_elementwise_fma(p[i & 7], f32x2{b, b}, f32x2{a, a});
        if (KIND == 2) x[i & 7] = __builtin_amdgcn_exp2f(x[i & 7]);
    }
}

// ROLE 0: MFMA + VALU in every wave; 1: waves 0-3 MFMA only, waves 4-7 VALU only; 2: MFMA only; 3: VALU only
template <int NV, int KIND, int ROLE>
__global__ void k(int iters, float *out) {
    const int wave = threadIdx.x >> 6;
    const float a = threadIdx.x * 1e-3f + 1.0f, b = 0.999f;
    f16x8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(a + i); hb[i] = (_Float16)(b + i); }
    f32x16 c[4] = {{0}, {0}, {0}, {0}};
    float x[8];
    f32x2 p[8];
    for (int i = 0; i < 8; ++i) { x[i] = a + i; p[i] = f32x2{a + i, a - i}; }
    const bool do_m = ROLE == 0 || ROLE == 2 || (ROLE == 1 && wave < 4);
    const bool do_v = ROLE == 0 || ROLE == 3 || (ROLE == 1 && wave >= 4);
    if (do_m && do_v) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                c[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c[u & 3], 0, 0, 0);
                valu<NV, KIND>(x, p, a, b);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if (do_m) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) c[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c[u & 3], 0, 0, 0);
        }
    } else if (do_v) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) valu<NV, KIND>(x, p, a, b);
        }
    }
    float r = 0.f;
    for (int i = 0; i < 8; ++i) r += x[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r + c[0][0] + c[1][1] + c[2][2] + c[3][3];
}

template <int NV, int KIND, int ROLE>
float run(int threads, float *out) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<NV, KIND, ROLE><<<256, threads>>>(50, out);
    (void)hipEventRecord(e0);
    k<NV, KIND, ROLE><<<256, threads>>>(iters, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6f / (iters * 16.0f);   // ns per (MFMA + NV valu) slot per wave
}

template <int KIND>
void sweep(const char *name, float *out) {
    printf("%s: ns per slot {1 MFMA + NV ops}; 1 wave/SIMD | 2 waves/SIMD (per wave) | specialised pair (MFMA wave + VALU wave)\n", name);
#define ROW(NV) printf("  NV=%2d   %6.1f | %6.1f | %6.1f   (VALU alone, 1 wave: %6.1f)\n", NV, run<NV, KIND, 0>(256, out), \
                       run<NV, KIND, 0>(512, out), run<NV, KIND, 1>(512, out), run<NV, KIND, 3>(256, out));
    ROW(0) ROW(2) ROW(4) ROW(6) ROW(8) ROW(12) ROW(16) ROW(24)
}

int main() {
    float *out;
    (void)hipMalloc(&out, 256 * 512 * 4);
    sweep<0>("v_fma_f32", out);
    sweep<1>("v_pk_fma_f32", out);
    sweep<2>("v_exp_f32", out);
    return 0;
}
