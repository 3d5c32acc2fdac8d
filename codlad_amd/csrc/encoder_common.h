// Helpers shared by the two tensor-product convolution kernels (encoder_kernels.hip, encoder_mfma_kernel.hip).
#pragma once
#include "common.h"
#include "../../include/codlad_hip.h"

namespace {

typedef const __attribute__((address_space(4))) float *kfloat_p;
DEV kfloat_p uni(const float *p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    return (kfloat_p)p;
#pragma clang diagnostic pop
}

constexpr int NS = 12, NV = 4;              // 12 scalars, 4 vectors per irrep block (reference vae_model.py:37)
constexpr float INV_SQRT3 = 0.57735026918962576f, INV_SQRT6 = 0.40824829046386302f;
constexpr float W_A = 0.31622776601683794f, W_B = 0.18257418583505536f;   // 1/sqrt 10, 1/sqrt 30: wigner_3j(1, 2, 1)

// feature width of depth d (irreps 12x0e | 4x1o | 4x1e | 12x0o, cumulative): 12, 24, 36, 48
__host__ __device__ constexpr int width_of(int depth) { return 12 * (depth + 1); }

struct Vec3 {
    float x, y, z;
};
DEV float dot3(Vec3 a, Vec3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
DEV Vec3 cross3(Vec3 a, Vec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
// sum_ij v_i Y2_j w3j(1,2,1)[i,j,k]
DEV Vec3 w121(Vec3 v, const float *s) {
    return {W_A * (v.z * s[0] + v.y * s[1] - v.x * s[4]) - W_B * v.x * s[2],
            W_A * (v.x * s[1] + v.z * s[3]) + 2.0f * W_B * v.y * s[2],
            W_A * (v.x * s[0] + v.y * s[3] + v.z * s[4]) - W_B * v.z * s[2]};
}

}  // namespace
