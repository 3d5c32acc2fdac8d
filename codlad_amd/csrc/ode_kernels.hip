// Element-wise update of the ODE samplers (flow matching), its own small translation unit.
#include "common.h"
#include "../../include/codlad_hip.h"

// out = y + h * sum_i coef[i] * k[i]: the stage / step update of an explicit Runge-Kutta method
struct OdeCombineArgs {
    const float *y;
    const float *k[7];
    float coef[7];
    int n_k;
    float h;
    float *out;
    size_t n;
};
__global__ void ode_combine_kernel(OdeCombineArgs a) {
#pragma clang fp contract(off)
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    // torchdiffeq forms sum_i k_i * (beta_i * dt) left to right and adds it to y: separately rounded ops
    float acc = a.k[0][i] * (a.coef[0] * a.h);
    for (int j = 1; j < a.n_k; ++j) acc = acc + a.k[j][i] * (a.coef[j] * a.h);
    a.out[i] = a.y[i] + acc;
}

extern "C" int codlad_ode_combine(const float *y, const float *const *k_host, const float *coef_host, int n_k, float h,
                                  size_t n, float *out, void *stream) {
    CODLAD_REQUIRE(y && k_host && coef_host && out, "null pointer");
    CODLAD_REQUIRE(n_k >= 1 && n_k <= 7 && n > 0, "1 to 7 stages");
    OdeCombineArgs a = {};
    a.y = y; a.n_k = n_k; a.h = h; a.out = out; a.n = n;
    for (int j = 0; j < n_k; ++j) {
        CODLAD_REQUIRE(k_host[j], "null stage pointer");
        a.k[j] = k_host[j];
        a.coef[j] = coef_host[j];
    }
    hipLaunchKernelGGL(ode_combine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return codlad_check_launch("codlad_ode_combine");
}
