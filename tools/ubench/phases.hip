// Two waves per SIMD, each alternating a pure VALU phase (NV independent-chain v_fma_f32) and a pure
// matrix phase (NM v_mfma_f32_32x32x16_f16, 3-long accumulate chains), no memory traffic at all:
// does the SIMD overlap one wave's VALU phase with the other's matrix phase?
//   OFFSET 0: both waves start with the VALU phase; 1: waves 4-7 start with the matrix phase.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NV, int NM>
__device__ inline void phase_pair(float (&x)[8], f32x16 (&c)[4], f16x8 ha, f16x8 hb, float a, float b, bool m_first) {
    if (m_first) {
#pragma unroll
        for (int u = 0; u < NM; ++u) c[(u / 3) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c[(u / 3) & 3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) x[i & 7] = __builtin_fmaf(x[i & 7], b, a);
    __builtin_amdgcn_sched_barrier(0);
    if (!m_first) {
#pragma unroll
        for (int u = 0; u < NM; ++u) c[(u / 3) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c[(u / 3) & 3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int NV, int NM, int OFFSET>
__global__ void k(int iters, float *out) {
    const int wave = threadIdx.x >> 6;
    const float a = threadIdx.x * 1e-3f + 1.0f, b = 0.999f;
    f16x8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(a + i); hb[i] = (_Float16)(b + i); }
    f32x16 c[4] = {{0}, {0}, {0}, {0}};
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = a + i;
    const bool m_first = OFFSET && wave >= 4;
    if (m_first) for (int it = 0; it < iters; ++it) phase_pair<NV, NM>(x, c, ha, hb, a, b, true);
    else for (int it = 0; it < iters; ++it) phase_pair<NV, NM>(x, c, ha, hb, a, b, false);
    float r = 0.f;
    for (int i = 0; i < 8; ++i) r += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r + c[0][0] + c[1][1] + c[2][2] + c[3][3];
}

template <int NV, int NM, int OFFSET>
float run(int threads, float *out) {
    const int iters = 400;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<NV, NM, OFFSET><<<256, threads>>>(20, out);
    (void)hipEventRecord(e0);
    k<NV, NM, OFFSET><<<256, threads>>>(iters, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / iters;   // us per (V + M) phase pair per wave
}

int main() {
    float *out;
    (void)hipMalloc(&out, 256 * 512 * 4);
#define ROW(NV, NM) printf("NV=%4d NM=%3d: 1 wave/SIMD %6.2f us | 2 waves in phase %6.2f us | 2 waves anti-phase %6.2f us\n", NV, NM, \
                           run<NV, NM, 0>(256, out), run<NV, NM, 0>(512, out), run<NV, NM, 1>(512, out));
    ROW(0, 96) ROW(1000, 0) ROW(1000, 96) ROW(500, 96) ROW(2000, 96) ROW(1000, 48)
    return 0;
}
