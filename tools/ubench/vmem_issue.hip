// What does ONE vector memory instruction cost a CU when its data is already in the L1 (every lane of every wave reads
// the same 128-byte line)?  8 waves per CU, loads written as inline asm so that nothing is merged or hoisted; 16 loads,
// one s_waitcnt, repeat.  Widths: dword (4 B per lane), dwordx2, dwordx4 (16 B per lane = 1 KB per instruction).
#include <hip/hip_runtime.h>
#include <cstdio>

template <int WIDTH>
__global__ __launch_bounds__(512) void k(const float *src, int iters, float *sink) {
    const float *p = src + (threadIdx.x & 1) * 4;        // two 16-byte pieces of one line
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (WIDTH == 4) {
                float4 v;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
                asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
                (void)v;
            } else if (WIDTH == 2) {
                float2 v;
                asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
                asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
                (void)v;
            } else {
                float v;
                asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
                asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
                acc += 0.f * (float)i;
                (void)v;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 12345.678f) sink[threadIdx.x] = acc;
}

template <int WIDTH>
void run(const char *name, const float *src, float *sink, int threads) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<WIDTH><<<256, threads>>>(src, 10, sink);
    (void)hipEventRecord(e0);
    k<WIDTH><<<256, threads>>>(src, iters, sink);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_cu = (threads / 64.0) * iters * 16.0;
    printf("%-22s %d waves per CU: %6.1f ns per wave instruction per CU = %5.1f bytes per ns per CU\n", name, threads / 64,
           ms * 1e6 / inst_per_cu, 64.0 * 4 * WIDTH / (ms * 1e6 / inst_per_cu));
}

int main() {
    float *src, *sink;
    (void)hipMalloc(&src, 4096); (void)hipMalloc(&sink, 4096);
    (void)hipMemset(src, 0, 4096);
    for (int threads : {512, 256}) {
        if (threads == 512) {
            run<1>("global_load_dword", src, sink, 512);
            run<2>("global_load_dwordx2", src, sink, 512);
            run<4>("global_load_dwordx4", src, sink, 512);
        } else {
            run<4>("global_load_dwordx4", src, sink, 256);
        }
    }
    return 0;
}
