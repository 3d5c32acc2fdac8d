// Throughput of 16-byte-per-lane vector memory instructions from one CU with 8 waves (2 per SIMD), the shape of the edge
// kernels: per "tile" a wave issues NLOAD global_load_dwordx4 (+ optionally 16 stores), waits for all of them, does a
// token amount of arithmetic, and goes on.  Patterns: SAME (every lane the same 32 bytes: always an L1 hit), ROWS (32 rows
// of 512 B picked per lane, as the Q gather: L2 hits), STREAM (1 KB contiguous per instruction through a 1.1 GB array: HBM).
// Reports ns per wave-level instruction per CU and the implied chip-wide TB/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

enum { SAME = 0, ROWS = 1, STREAM = 2, UPD = 3, ROWS_SKEW = 4, ROWS2 = 5, UPD2 = 6 };   // ROWS2: two WHOLE rows per instruction (lanes 0-31 one 512-byte row, lanes 32-63 the next); UPD2: UPD with ROWS2 gathers
   // ROWS_SKEW: the four 128-byte lines of a row rotated by the row index      // UPD: 16 gathered-row loads + 16 streaming loads (+ stores in place)

template <int PATTERN, int NLOAD, bool STORE>
__global__ __launch_bounds__(512) void k(const float4 *src, float4 *dst, const int *rows, int tiles, size_t tile_f4, float *sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const size_t w = (size_t)blockIdx.x * 8 + wave, nw = (size_t)gridDim.x * 8;
    float acc = 0.f;
    for (int t = 0; t < tiles; ++t) {
        const size_t tile = (w + (size_t)t * nw) % (tile_f4 ? tile_f4 : 1);
        float4 v[NLOAD];
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const float4 *p;
            if (PATTERN == SAME) p = src + h + 2 * ((t + i) & 3);                 // one 128-byte line, whatever t and i
            else if (PATTERN == ROWS_SKEW) {
                const int r = rows[(t * 32 + c) & 4095];
                p = src + (size_t)r * 32 + ((((i & 15) >> 2) + r) & 3) * 8 + 2 * (i & 3) + h;
            }
            else if (PATTERN == ROWS2) p = src + (size_t)rows[(t * 32 + 2 * (i & 15) + h) & 4095] * 32 + c;
            else if (PATTERN == UPD2 && i >= 16) p = dst + (size_t)rows[(t * 32 + 2 * (i & 15) + h) & 4095] * 32 + c;
            else if (PATTERN == ROWS) p = src + (size_t)rows[(t * 32 + c) & 4095] * 32 + 2 * (i & 15) + h;
            else if (PATTERN == UPD && i >= 16) p = dst + (size_t)rows[(t * 32 + c) & 4095] * 32 + 2 * (i & 15) + h;
            else p = src + tile * 1024 + (size_t)(2 * (i & 15) + h) * 32 + c;           // 16 KB per tile, 1 KB per instruction
            v[i] = *p;
        }
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) acc += v[i].x + v[i].w;
        if (STORE) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                ((PATTERN == UPD || PATTERN == UPD2) ? const_cast<float4 *>(src) : dst)[tile * 1024 + (size_t)(2 * i + h) * 32 + c] = make_float4(acc, v[i % NLOAD].y, v[i % NLOAD].z, acc);
        }
    }
    if (acc == 12345.678f) sink[threadIdx.x] = acc;
}

template <int PATTERN, int NLOAD, bool STORE>
void run(const char *name, const float4 *src, float4 *dst, const int *rows, size_t n_tiles, float *sink) {
    const int tiles = 64;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<PATTERN, NLOAD, STORE><<<256, 512>>>(src, dst, rows, 4, n_tiles, sink);
    (void)hipEventRecord(e0);
    k<PATTERN, NLOAD, STORE><<<256, 512>>>(src, dst, rows, tiles, n_tiles, sink);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_cu = 8.0 * tiles * (NLOAD + (STORE ? 16 : 0));
    const double bytes = 256.0 * inst_per_cu * 1024.0;
    printf("%-34s %6.1f ns per wave instruction per CU   %5.2f TB/s of 16-byte lanes chip-wide   (%.3f ms)\n", name,
           ms * 1e6 / inst_per_cu, bytes / (ms * 1e-3) / 1e12, ms);
}

int main() {
    const size_t n_tiles = 68000;                       // x 16 KB = 1.1 GB
    float4 *src, *dst; int *rows; float *sink;
    (void)hipMalloc(&src, n_tiles * 16384); (void)hipMalloc(&dst, n_tiles * 16384);
    (void)hipMalloc(&rows, 4096 * sizeof(int)); (void)hipMalloc(&sink, 4096);
    (void)hipMemset(src, 0, n_tiles * 16384);
    int hrows[4096];
    srand(1);
    for (int i = 0; i < 4096; ++i) hrows[i] = rand() % 35400;      // 35 400 rows of 512 B = 18 MB (the Q array of cfg 2)
    (void)hipMemcpy(rows, hrows, sizeof(hrows), hipMemcpyHostToDevice);
    run<SAME, 16, false>("16 loads, same line", src, dst, rows, n_tiles, sink);
    run<SAME, 32, false>("32 loads, same line", src, dst, rows, n_tiles, sink);
    run<ROWS, 16, false>("16 loads, 32 rows per instruction", src, dst, rows, n_tiles, sink);
    run<ROWS_SKEW, 16, false>("16 loads, 32 rows, lines rotated", src, dst, rows, n_tiles, sink);
    run<STREAM, 16, false>("16 loads, streaming", src, dst, rows, n_tiles, sink);
    run<STREAM, 16, true>("16 loads + 16 stores, streaming", src, dst, rows, n_tiles, sink);
    run<STREAM, 32, true>("32 loads + 16 stores, streaming", src, dst, rows, n_tiles, sink);
    run<UPD, 32, false>("16 streaming + 16 row loads", src, dst, rows, n_tiles, sink);
    run<UPD, 32, true>("the same + 16 stores in place", src, dst, rows, n_tiles, sink);
    run<ROWS2, 16, false>("16 loads, 2 whole rows per instruction", src, dst, rows, n_tiles, sink);
    run<UPD2, 32, false>("16 streaming + 16 whole-row loads", src, dst, rows, n_tiles, sink);
    run<UPD2, 32, true>("the same + 16 stores in place", src, dst, rows, n_tiles, sink);
    return 0;
}
