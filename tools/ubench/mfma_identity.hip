// Is one f16 MFMA with a single non-zero product per output, D = C + a * b, the correctly rounded fp32 fma(a, b, C)?
// (What the one-wave edge update needs to form its residual C + 2^E hi + 2^E lo on the matrix pipe, bit for bit equal to
// two v_fma_mix_f32 on the vector pipe.)  A = scale * I (32 x 16 slice of the identity), B = random fp16 values (normal,
// subnormal, both signs), C = random fp32.  build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_identity.hip -o tools/ubench/mfma_identity
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k(const _Float16 *b, const float *c, float scale, float *d_mfma, float *d_fma, int s) {
    const int lane = threadIdx.x, h = lane >> 5, m = lane & 31;
    // A operand: row m, k = 8 h + j; identity slice: non-zero where m == 16 s + 8 (j >> 2) + 4 h + (j & 3)
    f16x8 a;
    for (int j = 0; j < 8; ++j) a[j] = (m == 16 * s + 8 * (j >> 2) + 4 * h + (j & 3)) ? (_Float16)scale : (_Float16)0.f;
    // B operand: column n = lane & 31, k = 8 h + j
    f16x8 bb;
    for (int j = 0; j < 8; ++j) bb[j] = b[(lane & 31) * 16 + 8 * h + j];
    f32x16 cc;
    for (int r = 0; r < 16; ++r) cc[r] = c[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + (lane & 31)];
    f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bb, cc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        d_mfma[row * 32 + (lane & 31)] = d[r];
        // the k that feeds this row: row == 16 s + 8 (j >> 2) + 4 h' + (j & 3)
        float ref = cc[r];
        if (row >= 16 * s && row < 16 * s + 16) {
            const int q = row - 16 * s, j = 4 * (q >> 3) + (q & 3), hh = (q >> 2) & 1;
            ref = fmaf((float)b[(lane & 31) * 16 + 8 * hh + j], scale, cc[r]);
        }
        d_fma[row * 32 + (lane & 31)] = ref;
    }
}

int main() {
    _Float16 hb[32 * 16]; float hc[32 * 32];
    _Float16 *b; float *c, *d1, *d2;
    (void)hipMalloc(&b, sizeof(hb)); (void)hipMalloc(&c, sizeof(hc)); (void)hipMalloc(&d1, sizeof(hc)); (void)hipMalloc(&d2, sizeof(hc));
    srand(3);
    long bad = 0, total = 0;
    const float scales[] = {1.f, 0.5f, 4.f, 32768.f, 1.f / 16384.f, 256.f};
    for (int trial = 0; trial < 2000; ++trial) {
        const float scale = scales[trial % 6];
        for (int i = 0; i < 32 * 16; ++i) {
            const int mode = rand() % 4;
            float v = (rand() / (float)RAND_MAX - 0.5f) * 8.f;
            if (mode == 1) v *= 1e-3f;          // small
            if (mode == 2) v *= 1e-6f;          // fp16 subnormal range
            hb[i] = (_Float16)v;
        }
        for (int i = 0; i < 32 * 32; ++i) {
            float v = (rand() / (float)RAND_MAX - 0.5f) * 4.f * scale;
            if (rand() % 8 == 0) v *= 1e-4f;
            hc[i] = v;
        }
        (void)hipMemcpy(b, hb, sizeof(hb), hipMemcpyHostToDevice);
        (void)hipMemcpy(c, hc, sizeof(hc), hipMemcpyHostToDevice);
        k<<<1, 64>>>(b, c, scale, d1, d2, trial & 1);
        float r1[32 * 32], r2[32 * 32];
        (void)hipMemcpy(r1, d1, sizeof(r1), hipMemcpyDeviceToHost);
        (void)hipMemcpy(r2, d2, sizeof(r2), hipMemcpyDeviceToHost);
        for (int i = 0; i < 32 * 32; ++i) {
            ++total;
            if (memcmp(&r1[i], &r2[i], 4) != 0) {
                if (bad < 10) printf("differs: mfma %.9g fma %.9g (scale %g)\n", r1[i], r2[i], scale);
                ++bad;
            }
        }
    }
    printf("%ld of %ld results differ\n", bad, total);
    return bad != 0;
}
