// Does v_mfma_f32_32x32x2_f32 co-execute with ordinary fp32 VALU work of ANOTHER wave on the
// same SIMD?  512-thread workgroups: waves 0-3 and 4-7 are SIMD partners.
//   mode 0: waves 0-3 MFMA only            mode 1: waves 4-7 VALU (v_fma_f32) only
//   mode 2: both                            mode 3: waves 4-7 VALU packed (v_pk_fma_f32)
//   mode 4: MFMA + packed VALU              mode 5: ONE wave interleaves 4 MFMA + 16 v_fma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(512) void k(int mode, int iters, float *out) {
    const int wave = threadIdx.x >> 6;
    const bool mf = wave < 4;
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    if ((mode == 0 || mode == 2 || mode == 4) && mf) {
        f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
            }
        }
        out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    } else if ((mode == 1 || mode == 2) && !mf) {
        float x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 32; ++u) {   // 256 v_fma per iteration = 1024 cycles at 4 cyc/instr
                x0 = fmaf(x0, b, a); x1 = fmaf(x1, b, a); x2 = fmaf(x2, b, a); x3 = fmaf(x3, b, a);
                x4 = fmaf(x4, b, a); x5 = fmaf(x5, b, a); x6 = fmaf(x6, b, a); x7 = fmaf(x7, b, a);
            }
        }
        out[blockIdx.x * 512 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    } else if ((mode == 3 || mode == 4) && !mf) {
        f32x2 x0 = {a, a}, x1 = {a + 1, a}, x2 = {a + 2, a}, x3 = {a + 3, a}, x4 = {a + 4, a}, x5 = {a + 5, a},
              x6 = {a + 6, a}, x7 = {a + 7, a};
        const f32x2 bb = {b, b}, aa = {a, a};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                x0 = x0 * bb + aa; x1 = x1 * bb + aa; x2 = x2 * bb + aa; x3 = x3 * bb + aa;
                x4 = x4 * bb + aa; x5 = x5 * bb + aa; x6 = x6 * bb + aa; x7 = x7 * bb + aa;
            }
        }
        out[blockIdx.x * 512 + threadIdx.x] = x0.x + x1.y + x2.x + x3.y + x4.x + x5.y + x6.x + x7.y;
    } else if ((mode == 6 || mode == 7 || mode == 8 || mode == 9 || mode == 10) && mf) {
        if (mode == 9) __builtin_amdgcn_s_setprio(3);
        typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
        f16x8 ha, hb;
        for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(a + i); hb[i] = (_Float16)(b + i); }
        f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        float x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {   // 32 f16 MFMAs per iteration = 1024 matrix cycles
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c0, 0, 0, 0);
                if (mode == 8) { x0 = fmaf(x0, b, a); x1 = fmaf(x1, b, a); x2 = fmaf(x2, b, a); x3 = fmaf(x3, b, a); }
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c1, 0, 0, 0);
                if (mode == 8) { x0 = fmaf(x0, b, a); x1 = fmaf(x1, b, a); x2 = fmaf(x2, b, a); x3 = fmaf(x3, b, a); }
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c2, 0, 0, 0);
                if (mode == 8) { x0 = fmaf(x0, b, a); x1 = fmaf(x1, b, a); x2 = fmaf(x2, b, a); x3 = fmaf(x3, b, a); }
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c3, 0, 0, 0);
                if (mode == 8) { x0 = fmaf(x0, b, a); x1 = fmaf(x1, b, a); x2 = fmaf(x2, b, a); x3 = fmaf(x3, b, a); }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + x0 + x1 + x2 + x3;
    } else if ((mode == 7 || mode == 9 || mode == 10) && !mf) {
        if (mode == 10) __builtin_amdgcn_s_setprio(3);
        float x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                x0 = fmaf(x0, b, a); x1 = fmaf(x1, b, a); x2 = fmaf(x2, b, a); x3 = fmaf(x3, b, a);
                x4 = fmaf(x4, b, a); x5 = fmaf(x5, b, a); x6 = fmaf(x6, b, a); x7 = fmaf(x7, b, a);
            }
        }
        out[blockIdx.x * 512 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    } else if (mode >= 11 && mode <= 14 && mf) {
        typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
        f16x8 ha, hb;
        for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(a + i); hb[i] = (_Float16)(b + i); }
        f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        float x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
        f32x2 p0 = {a, a}, p1 = {a + 1, a}, p2 = {a + 2, a}, p3 = {a + 3, a};
        const f32x2 bb = {b, b}, aa = {a, a};
#define FILL()                                                                                         \
    if (mode == 11) { x0 = fmaf(x0, b, a); x1 = fmaf(x1, b, a); x2 = fmaf(x2, b, a); x3 = fmaf(x3, b, a);  \
                      x4 = fmaf(x4, b, a); x5 = fmaf(x5, b, a); x6 = fmaf(x6, b, a); x7 = fmaf(x7, b, a); } \
    if (mode == 12) { p0 = p0 * bb + aa; p1 = p1 * bb + aa; p2 = p2 * bb + aa; p3 = p3 * bb + aa; }      \
    if (mode == 13) { x0 = __builtin_amdgcn_exp2f(x0); x1 = __builtin_amdgcn_exp2f(x1); }               \
    if (mode == 14) { x0 = fmaf(x0, b, a); x1 = fmaf(x1, b, a); x2 = fmaf(x2, b, a); x3 = fmaf(x3, b, a);  \
                      x4 = fmaf(x4, b, a); x5 = fmaf(x5, b, a); }
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c0, 0, 0, 0);
                FILL()
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c1, 0, 0, 0);
                FILL()
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c2, 0, 0, 0);
                FILL()
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c3, 0, 0, 0);
                FILL()
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 +
                                              p0.x + p1.y + p2.x + p3.y;
    } else if (mode == 5 && mf) {
        f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        float x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
                x0 = fmaf(x0, b, a); x1 = fmaf(x1, b, a); x2 = fmaf(x2, b, a); x3 = fmaf(x3, b, a);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
                x0 = fmaf(x0, b, a); x1 = fmaf(x1, b, a); x2 = fmaf(x2, b, a); x3 = fmaf(x3, b, a);
                c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
                x0 = fmaf(x0, b, a); x1 = fmaf(x1, b, a); x2 = fmaf(x2, b, a); x3 = fmaf(x3, b, a);
                c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
                x0 = fmaf(x0, b, a); x1 = fmaf(x1, b, a); x2 = fmaf(x2, b, a); x3 = fmaf(x3, b, a);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + x0 + x1 + x2 + x3;
    }
}

int main() {
    float *out;
    hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    const char *names[] = {"MFMA only (16/iter)", "v_fma only (256/iter)", "MFMA + v_fma partner waves",
                           "v_pk_fma only (256/iter)", "MFMA + v_pk_fma partner waves",
                           "one wave: 16 MFMA + 64 v_fma interleaved", "f16 MFMA only (32/iter)",
                           "f16 MFMA + v_fma(256/iter) partner waves", "one wave: 32 f16 MFMA + 128 v_fma interleaved",
                           "f16 MFMA (prio 3) + v_fma partner", "f16 MFMA + v_fma (prio 3) partner",
                           "one wave: 32 f16 MFMA + 8 v_fma per gap", "one wave: 32 f16 MFMA + 4 v_pk_fma per gap",
                           "one wave: 32 f16 MFMA + 2 v_exp per gap", "one wave: 32 f16 MFMA + 6 v_fma per gap"};
    for (int mode = 0; mode < 15; ++mode) {
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, 10, out);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, iters, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d  %-45s %8.3f ms  -> %.1f cycles/iter @2.4GHz\n", mode, names[mode], ms,
               ms * 1e-3 * 2.4e9 / iters);
    }
    return 0;
}
