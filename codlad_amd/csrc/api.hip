// Host-side plumbing of libcodlad_hip.so: error reporting and the weight-block packer.
#include "common.h"
#include "../../include/codlad_hip.h"

#include <stdarg.h>
#include <stddef.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void codlad_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

hipError_t codlad_take_attr_error();   // denoiser_kernels.hip: a failed hipFuncSetAttribute, if any

int codlad_check_launch(const char *what) {
    hipError_t e = codlad_take_attr_error();
    if (e != hipSuccess) {
        codlad_set_error("%s: raising the dynamic LDS limit of a kernel failed: %s", what, hipGetErrorString(e));
        (void)hipGetLastError();
        return (int)e;
    }
    e = hipGetLastError();
    if (e != hipSuccess) {
        codlad_set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

extern "C" int codlad_abi_version(void) { return CODLAD_ABI_VERSION; }

extern "C" const char *codlad_last_error(void) { return g_err; }

extern "C" void codlad_struct_sizes(int *out5) {
    out5[0] = (int)sizeof(codlad_denoiser_weights);
    out5[1] = (int)sizeof(codlad_decoder_weights);
    out5[2] = (int)sizeof(codlad_workspace);
    out5[3] = (int)offsetof(codlad_denoiser_weights, precision);
    out5[4] = (int)offsetof(codlad_denoiser_weights, enc_h);
}

extern "C" void codlad_pack_block_host(const float *src, int ld, float scale, float *dst) {
    for (int b = 0; b < 4; ++b)
        for (int r = 0; r < 16; ++r)
            for (int lane = 0; lane < 64; ++lane)
                for (int bo = 0; bo < 4; ++bo) {
                    const int row = 32 * bo + (lane & 31);
                    const int col = 32 * b + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    dst[((16 * b + r) * 64 + lane) * 4 + bo] = scale * src[(size_t)row * ld + col];
                }
}
