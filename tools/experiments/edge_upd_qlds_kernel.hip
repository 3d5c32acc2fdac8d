// The edge-update kernel of the split-fp16 modes with the neighbours' Q rows fetched as whole rows and transposed through
// LDS (variant 1 of CODLAD_OPT_EDGE_UPD_VARIANT; edge_upd_kernel.hip holds variant 0 and says what the kernel computes).
// Its own translation unit: hipcc's register allocation for these kernels changes with whatever is compiled beside them
// (edge_args.h).
#include "edge_args.h"

// Edge update, split-fp16 contractions:
//   h_E[n,j] <- mod3(LN(h_E[n,j] + W13 GELU(W12 GELU(P_i + Q_j + W11e h_E[n,j]) + b12) + b13))
// HOISTED: encoder layer 0 with a.E1 given - layer 1's edge contraction comes precomputed.
//
// QLDS: the neighbours' Q rows arrive through LDS.  A lane owns a column (an edge), so fetched straight into the chain
// layout every one of the 16 gather instructions reaches into 32 different rows for 32 bytes each - 512 line touches per
// tile for 16 KB, 56 ns per instruction and CU in isolation (tools/ubench/vmem_rate.hip) and, in this kernel, the reason
// its texture-address FIFO runs full (SQ_VMEM_TA_ADDR_FIFO_FULL 1.7e7 against 0 in the message kernel).  With QLDS a wave
// instruction fetches two WHOLE rows (lanes 0-31 one 512-byte row, lanes 32-63 the next: 8 lines per instruction, 128
// per tile), four rows at a time are bounced through a padded per-wave LDS slot (544 bytes per row: the four rows land
// 8 banks apart, reads and writes conflict-free) and the two lanes that own each of those rows read their 16 float4s
// back in chain layout.  The LDS for the eight bounce slots (17 KB) is paid for with resident k-steps of W11e
// (KS = 1 instead of 3; each streamed k-step costs 8 coalesced 1 KB fragment loads per tile).
constexpr int QB_ROW_F = 136;                 // floats per bounce row: 512 bytes + 32 bytes of padding
constexpr int QB_WAVE_F = 4 * QB_ROW_F;       // four rows per pass

// Eight wave instructions, two whole rows each: instruction k fetches the rows of columns C0 + 2k (lanes 0-31) and
// C0 + 2k + 1 (lanes 32-63).  `j` holds, in lanes c and c + 32 alike, the neighbour of column c.  Buffer addressing:
// the base of Q sits in SGPRs and a lane's offset is one 32-bit register (flat addressing would keep a 64-bit address
// pair per load alive).
struct QStage8 {
    f32x4 v[8];
};

template <int C0>
DEV void q_rows_issue(QStage8 &st, __amdgpu_buffer_rsrc_t qrsrc, int base, int j, int lane) {
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int ja = __builtin_amdgcn_readlane(j, C0 + 2 * k), jb = __builtin_amdgcn_readlane(j, C0 + 2 * k + 1);
        const int row = base + (h ? jb : ja);
        st.v[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(qrsrc, row * (HD * 4) + 16 * c, 0, 0));
    }
}

// staged rows -> acc in chain layout (lane (c, h): the float4 chunks 2i + h of row c), passes P0 .. P0+3 of 4 rows.
// One wave writes and reads its own slot: the LDS serves a wave's instructions in order, so neither the read-back after
// the writes nor the next pass's writes after the read-back need a barrier.
template <int P0>
DEV void q_rows_to_tile(Tile &acc, const QStage8 &st, float *bounce, int lane) {
    const int c = lane & 31, h = lane >> 5;
    float *wr = bounce + h * QB_ROW_F + 4 * c;
    const float *rd = bounce + (c & 3) * QB_ROW_F + 4 * h;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        *reinterpret_cast<f32x4 *>(wr) = st.v[2 * p];                      // columns 4 (P0 + p) (h = 0), + 1 (h = 1)
        *reinterpret_cast<f32x4 *>(wr + 2 * QB_ROW_F) = st.v[2 * p + 1];   // columns 4 (P0 + p) + 2, + 3
        __builtin_amdgcn_wave_barrier();
        if ((c >> 2) == P0 + p) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(rd + 8 * i);
                acc.b[i >> 2][4 * (i & 3) + 0] = v.x;
                acc.b[i >> 2][4 * (i & 3) + 1] = v.y;
                acc.b[i >> 2][4 * (i & 3) + 2] = v.z;
                acc.b[i >> 2][4 * (i & 3) + 3] = v.w;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <int KS, int NWAVES>
constexpr int upd_q_lds_u4() {
    return 2 * LDS_BLOCK_U4 + KS * 512 + EDGE_CONST_U4 + NWAVES * 32 + NWAVES * QB_WAVE_F / 4;
}

template <int NWAVES, bool HOISTED, int TERMS, int KS>
__global__ __launch_bounds__(NWAVES * 64, NWAVES / 4) void upd_kernel_q(EdgeArgs a) {
    extern __shared__ __align__(16) u32x4 wl[];
    constexpr int NT = NWAVES * 64;
    constexpr int W1_U4 = KS * 512;
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
    float *bounce0 = reinterpret_cast<float *>(consts + EDGE_CONST_U4 + NWAVES * 32) + wave * QB_WAVE_F;
    const int h = lane >> 5, c = lane & 31;
    // all of Q as one buffer (the launcher holds n_nodes * 512 bytes below 2^31)
    const __amdgpu_buffer_rsrc_t qrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.Q), 0, a.n_nodes * (HD * 4), 0x00020000);
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
            StreamedGemm<TERMS, KS, 8 - KS, false, 8> tail1;
            // the constants never change, so the compiler would read them once, before the node
            // loop, into ~300 registers and spill those; an opaque zero offset keeps the reads here
            int lds_off = 0;
            asm volatile("" : "+v"(lds_off));
            const float *c_b2 = c_base + lds_off, *c_b3 = c_b2 + HD;
            const float *c_modA = c_b2 + 2 * HD, *c_modB = c_b2 + 3 * HD;
            float *bounce = bounce0 + lds_off;
            {
                // all 32 Q rows requested first (16 instructions, L2 hits), the first half passes through LDS, then the
                // tile's own rows (HBM) are requested into the registers that half has freed and travel while the
                // second half passes; the streamed part of W11e is requested last (its ring and the staging registers
                // do not fit side by side)
                QStage8 s0, s1;
                q_rows_issue<0>(s0, qrsrc, base, j, lane);
                q_rows_issue<16>(s1, qrsrc, base, j, lane);
                __builtin_amdgcn_sched_barrier(0);
                q_rows_to_tile<0>(acc, s0, bounce, lane);
                __builtin_amdgcn_sched_barrier(0);
                tile_load_edge<!HOISTED>(x, rows, colc, h);                      // layer-1 operand and residual
                __builtin_amdgcn_sched_barrier(0);
                q_rows_to_tile<4>(acc, s1, bounce, lane);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!HOISTED) tail1.start(a.W1h, lane);
            tile_add_row(acc, Pslot, h);
            if (HOISTED) {
                tile_add_edge(acc, a.E1 + (size_t)src * EDGE_BLOCK, colc, h);
            } else {
                if (KS > 0) gemm_h_lds<TERMS, 0, KS, false>(acc, x, w1, lane, a.gelu_a);   // layer 1, resident k-steps
                tail1.run(acc, x, lane, a.gelu_a);                                    // layer 1, streamed k-steps
            }
            tile_load_row(t2, c_b2, h);
            gemm128_h_lds<TERMS, true>(t2, acc, w2, lane, a.gelu_a);   // layer 2 on GELU(layer 1)
            // layer 3 accumulates onto (h_E + b13) * 2^E: the input tile stays in registers for the
            // residual instead of being fetched from HBM a second time (c_b3 holds b13 * 2^E)
            tile_scale_add_row(x, a.res_scale, c_b3, h);
            gemm128_h_lds<TERMS, true>(x, t2, w3, lane, a.gelu_b);     // layer 3 on GELU(layer 2)
            tile_layernorm_affine(x, a.ln_eps, c_modA, c_modB, h);
            if (valid) tile_store_edge<true>(x, out_rows, col, h);
        }
    }
}

template <int TERMS, int KS>
static void launch_upd_q(const EdgeArgs &ea, hipStream_t st) {
    static bool attr_set = false;     // one flag per instantiation
    constexpr int NW = 8;
    constexpr size_t lds = 16 * upd_q_lds_u4<KS, NW>();
    static_assert(lds <= 160 * 1024, "kernel exceeds the CU's LDS");
    if (!attr_set) {
        set_max_lds(reinterpret_cast<const void *>(upd_kernel_q<NW, false, TERMS, KS>), lds);
        set_max_lds(reinterpret_cast<const void *>(upd_kernel_q<NW, true, TERMS, KS>), lds);
        attr_set = true;
    }
    const int groups = (ea.n_nodes + NW - 1) / NW;
    dim3 grid(groups < num_cu() ? groups : num_cu()), block(NW * 64);
    if (ea.E1 != nullptr) hipLaunchKernelGGL((upd_kernel_q<NW, true, TERMS, KS>), grid, block, lds, st, ea);
    else hipLaunchKernelGGL((upd_kernel_q<NW, false, TERMS, KS>), grid, block, lds, st, ea);
}

void launch_edge_upd_qlds(int terms, const EdgeArgs &ea, hipStream_t st) {
    if (terms == 3) launch_upd_q<3, 1>(ea, st);
    else launch_upd_q<4, 1>(ea, st);
}
