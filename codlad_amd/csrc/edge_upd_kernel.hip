// The edge-update kernel of the split-fp16 modes (precision 1 = f16x4, 2 = f16x3).
// One hot kernel per translation unit: hipcc's register allocation for these kernels (written to sit exactly at the
// 256-register limit of two waves per SIMD) changed with whatever else was compiled beside them - another kernel in
// the same file was enough to move 20-130 registers into scratch inside the contraction loops (-9 %).
#include "edge_args.h"

// ---------------------------------------------------------------------------------------------
// Split-fp16 edge kernels (precision 1 = f16x4, 2 = f16x3).  Persistent 512-thread workgroups, one per CU: the two (three)
// 64 KB weight blocks of the MLP live in LDS for the whole launch - 128 KB for the message kernel,
// 152 KB for the edge update (W12, W13 and the first three k-steps of W11e; its last five k-steps
// stream from L2, see below) - and every wave walks its nodes with a stride (wave_node_span).  The
// contraction runs on the f16 matrix pipe, split / bias / GELU / reduction on the fp32 lanes.
// ---------------------------------------------------------------------------------------------
// Edge update, split-fp16 contractions:
//   h_E[n,j] <- mod3(LN(h_E[n,j] + W13 GELU(W12 GELU(P_i + Q_j + W11e h_E[n,j]) + b12) + b13))
// HOISTED: encoder layer 0 with a.E1 given - layer 1's edge contraction comes precomputed.
template <int NWAVES, bool HOISTED, int TERMS>
__global__ __launch_bounds__(NWAVES * 64, NWAVES / 4) void upd_kernel_h(EdgeArgs a) {
    extern __shared__ __align__(16) u32x4 wl[];
    constexpr int NT = NWAVES * 64;
    constexpr int W1_U4 = UPD_W1_KS * 512;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u32x4 *consts = wl + 2 * LDS_BLOCK_U4 + W1_U4;
    {
        const u32x4 *g1 = reinterpret_cast<const u32x4 *>(a.W1h);
        const u32x4 *g2 = reinterpret_cast<const u32x4 *>(a.W2h);
        const u32x4 *g3 = reinterpret_cast<const u32x4 *>(a.W3h);
        for (int i = tid; i < LDS_BLOCK_U4; i += NT) {
            wl[i] = g2[i];
            wl[LDS_BLOCK_U4 + i] = g3[i];
        }
        if (!HOISTED)
            for (int i = tid; i < W1_U4; i += NT) wl[2 * LDS_BLOCK_U4 + i] = g1[i];
        if (tid < 32) consts[tid] = reinterpret_cast<const u32x4 *>(a.b2)[tid];
        if (tid >= 64 && tid < 96) consts[32 + (tid & 31)] = reinterpret_cast<const u32x4 *>(a.b3)[tid & 31];
        if (tid >= 128 && tid < 160) {
            // modulate folded to one multiply-add: A = gate (1 + scale), B = gate shift
            const float4 *m = reinterpret_cast<const float4 *>(a.mods3);
            const int i = tid & 31;
            const float4 s = m[i], c = m[32 + i], g = m[64 + i];
            float4 *cf = reinterpret_cast<float4 *>(consts);
            cf[64 + i] = make_float4(g.x * (1.0f + c.x), g.y * (1.0f + c.y), g.z * (1.0f + c.z), g.w * (1.0f + c.w));
            cf[96 + i] = make_float4(g.x * s.x, g.y * s.y, g.z * s.z, g.w * s.w);
        }
    }
    __syncthreads();
    const u32x4 *w2 = wl, *w3 = wl + LDS_BLOCK_U4, *w1 = wl + 2 * LDS_BLOCK_U4;
    const float *c_base = reinterpret_cast<const float *>(consts);
    float *Pslot = reinterpret_cast<float *>(consts + EDGE_CONST_U4 + wave * 32);
    const int h = lane >> 5, c = lane & 31;
    const NodeSpan span = wave_node_span(a.n_nodes, NWAVES, wave);
    for (int n = span.first; n < span.end; n += span.stride) {
        const int4 info = a.node_info[n];
        const int src = info.x, base = info.y, K = info.z;
        const float *rows = a.hE_in + (size_t)(a.in_by_src ? src : n) * EDGE_BLOCK;
        float *out_rows = a.hE_out + (size_t)n * EDGE_BLOCK;
        const bool validA = c < K, validB = 32 + c < K;
        const int colA = validA ? c : 0, colB = validB ? 32 + c : 0;
        const int jA = a.E_idx[(size_t)src * 64 + colA], jB = a.E_idx[(size_t)src * 64 + colB];
        // this node's P row: one coalesced 512-byte read, staged in the wave's own LDS slot
        // (same wave writes and reads: program order + lgkmcnt, no barrier)
        reinterpret_cast<float2 *>(Pslot)[lane] = reinterpret_cast<const float2 *>(a.P + (size_t)n * HD)[lane];

        for (int half = 0; half < 2; ++half) {
            if (32 * half >= K) break;
            const bool valid = half ? validB : validA;
            const int colc = half ? colB : colA, col = 32 * half + c;
            const int j = half ? jB : jA;
            Tile x, acc, t2;
            StreamedGemm<TERMS, UPD_W1_KS, 8 - UPD_W1_KS, false, 8, true> tail1;
            if (!HOISTED) tail1.start(a.W1h, lane);
            // the constants never change, so the compiler would read them once, before the node
            // loop, into ~300 registers and spill those; an opaque zero offset keeps the reads here
            int lds_off = 0;
            asm volatile("" : "+v"(lds_off));
            const float *c_b2 = c_base + lds_off, *c_b3 = c_b2 + HD;
            const float *c_modA = c_b2 + 2 * HD, *c_modB = c_b2 + 3 * HD;
            tile_load_row(acc, a.Q + (size_t)(base + j) * HD, h);
            tile_add_row(acc, Pslot, h);
            if (!HOISTED) tile_load_edge<true>(x, rows, colc, h);                // layer-1 operand and residual
            if (HOISTED) {
                tile_add_edge(acc, a.E1 + (size_t)src * EDGE_BLOCK, colc, h);
                // the tile itself (h_E0, shared by the ensemble members: default cache policy) is only the residual here:
                // requested after the hoisted term has been added, it travels under layer 2.  (Requested first, as the
                // non-hoisted variant must, hipcc serialised the sixteen E1 loads behind it, one s_waitcnt vmcnt(0) each:
                // +25 % on this launch.)
                __builtin_amdgcn_sched_barrier(0);
                tile_load_edge<false>(x, rows, colc, h);
            } else {
                gemm_h_lds<TERMS, 0, UPD_W1_KS, false, false, true>(acc, x, w1, lane, a.gelu_a);   // layer 1, resident k-steps
                tail1.run(acc, x, lane, a.gelu_a);                                    // layer 1, streamed k-steps
            }
            tile_load_row(t2, c_b2, h);
            gemm128_h_lds<TERMS, true>(t2, acc, w2, lane, a.gelu_a);   // layer 2 on GELU(layer 1)
            // layer 3 accumulates onto (h_E + b13) * 2^E: the input tile stays in registers for the
            // residual instead of being fetched from HBM a second time (c_b3 holds b13 * 2^E)
            // (x arrives in the stored, pre-split form: its value is hi + lo)
            tile_unsplit_scale_add_row(x, a.res_scale, c_b3, h);
            gemm128_h_lds<TERMS, true>(x, t2, w3, lane, a.gelu_b);     // layer 3 on GELU(layer 2)
            tile_layernorm_affine(x, a.ln_eps, c_modA, c_modB, h);
            tile_presplit(x);                                          // stored as the halves the next contraction reads
            if (valid) tile_store_edge<true>(x, out_rows, col, h);
        }
    }
}


template <int TERMS>
static void launch_upd_t(const EdgeArgs &ea, hipStream_t st) {
    static bool attr_set = false;     // one flag per TERMS instantiation
    constexpr int NW = 8;
    const size_t lds = 16 * edge_lds_u4<true, NW>();
    static_assert(16 * edge_lds_u4<true, NW>() <= 160 * 1024, "kernel exceeds the CU's LDS");
    if (!attr_set) {
        set_max_lds(reinterpret_cast<const void *>(upd_kernel_h<NW, false, TERMS>), lds);
        set_max_lds(reinterpret_cast<const void *>(upd_kernel_h<NW, true, TERMS>), lds);
        attr_set = true;
    }
    const int groups = (ea.n_nodes + NW - 1) / NW;
    dim3 grid(groups < edge_cus() ? groups : edge_cus()), block(NW * 64);
    if (ea.E1 != nullptr) hipLaunchKernelGGL((upd_kernel_h<NW, true, TERMS>), grid, block, lds, st, ea);
    else hipLaunchKernelGGL((upd_kernel_h<NW, false, TERMS>), grid, block, lds, st, ea);
}

void launch_edge_upd(int terms, const EdgeArgs &ea, hipStream_t st) {
    if (terms == 3) launch_upd_t<3>(ea, st);
    else launch_upd_t<4>(ea, st);
}
