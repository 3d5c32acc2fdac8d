// The message kernel of the split-fp16 modes (precision 1 = f16x4, 2 = f16x3): the dominant kernel of a DDPM step.
// One hot kernel per translation unit: hipcc's register allocation for these kernels (written to sit exactly at the
// 256-register limit of two waves per SIMD) changed with whatever else was compiled beside them - another kernel in
// the same file was enough to move 20-130 registers into scratch inside the contraction loops (-9 %).
#include "edge_args.h"

// Message kernel, split-fp16 contractions: S[n] = sum_j GELU(W2 GELU(P_i + Q_j + W1e h_E[i,j]) + b2) over the K
// neighbours.  Same LDS residency scheme as upd_kernel_h; on top of that
//   * the LAST contraction is issued with swapped MFMA operands, so its output block arrives
//     transposed (lane = feature, registers = the tile's 32 edges): the sum over neighbours is
//     then 15 register adds per block instead of a 5-step cross-lane reduction per register;
//   * with the 64-register running sum gone, the next tile's edge rows are fetched while layer 2
//     runs and its Q rows while the epilogue runs (a wave walks its (node, half) tiles in order).
template <int NWAVES, bool HOISTED, int TERMS>
__global__ __launch_bounds__(NWAVES * 64, NWAVES / 4) void msg_kernel_h(EdgeArgs a) {
    extern __shared__ __align__(16) u32x4 wl[];
    constexpr int NT = NWAVES * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u32x4 *consts = wl + 2 * LDS_BLOCK_U4;
    {
        const u32x4 *g1 = reinterpret_cast<const u32x4 *>(a.W1h);
        const u32x4 *g2 = reinterpret_cast<const u32x4 *>(a.W2h);
        for (int i = tid; i < LDS_BLOCK_U4; i += NT) {
            if (!HOISTED) wl[i] = g1[i];
            wl[LDS_BLOCK_U4 + i] = g2[i];
        }
        if (tid < 32) consts[tid] = reinterpret_cast<const u32x4 *>(a.b2)[tid];
        // b2 once more, every value four times over in each of four copies: layer 2's accumulators start as 16 registers
        // of one value per lane (the output is transposed: lane = feature), which four 16-byte LDS reads deliver without a
        // vector instruction (64 v_mov per tile otherwise)
        u32x4 *rep = consts + EDGE_CONST_U4 + NWAVES * 32;
        if (tid < 128) {
            const unsigned bv = reinterpret_cast<const unsigned *>(a.b2)[tid];
            for (int q = 0; q < 4; ++q) rep[q * 128 + tid] = u32x4{bv, bv, bv, bv};
        }
    }
    __syncthreads();
    const u32x4 *w1 = wl, *w2 = wl + LDS_BLOCK_U4;
    const float *c_b2 = reinterpret_cast<const float *>(consts);
    float2 *Pslot = reinterpret_cast<float2 *>(consts + EDGE_CONST_U4 + wave * 32);
    const int h = lane >> 5, c = lane & 31;
    const NodeSpan span = wave_node_span(a.n_nodes, NWAVES, wave);
    const int stride = span.stride, n_end = span.end;
    int n = span.first;                              // wave-uniform
    if (n >= n_end) return;

    const float *xsrc = HOISTED ? a.E1 : a.hE_in;    // layer-1 edge operand: hoisted term or h_E
    auto block_of = [&](int node, int s) {
        return xsrc + (size_t)((HOISTED || a.in_by_src) ? s : node) * EDGE_BLOCK;
    };
    const float2 *Prows = reinterpret_cast<const float2 *>(a.P);

    int4 info = a.node_info[n];
    int src = __builtin_amdgcn_readfirstlane(info.x), base = __builtin_amdgcn_readfirstlane(info.y);
    int K = __builtin_amdgcn_readfirstlane(info.z);
    int jA = a.E_idx[(size_t)src * 64 + (c < K ? c : 0)];
    int jB = a.E_idx[(size_t)src * 64 + (32 + c < K ? 32 + c : 0)];
    Pslot[lane] = Prows[(size_t)n * 64 + lane];
    int half = 0;
    Tile x, acc, t2;
    tile_load_edge<!HOISTED>(x, block_of(n, src), c < K ? c : 0, h);
    tile_load_row(acc, a.Q + (size_t)(base + jA) * HD, h);
    float sum[4] = {0.f, 0.f, 0.f, 0.f};
    const float4 *bias_rep = reinterpret_cast<const float4 *>(consts + EDGE_CONST_U4 + NWAVES * 32) + c;

    // the node after this one (kept equal to the current node when there is none, so that the
    // prefetch below always has a valid address and needs no branch)
    int4 ninfo = info;
    int nsrc = src, nbase = base, nK = K, njA = jA, njB = jB;
    float2 npv = {0.f, 0.f};
    for (;;) {
        const int n2 = n + stride;
        const bool next_node = n2 < n_end;
        const bool first_half = half == 0;
        if (first_half && next_node) ninfo = a.node_info[n2];
        tile_add_row(acc, reinterpret_cast<const float *>(Pslot), h);
        if (HOISTED) {
#pragma unroll
            for (int bo = 0; bo < 4; ++bo) acc.b[bo] += x.b[bo];
        } else {
            gemm128_h_lds<TERMS, false, false, true>(acc, x, w1, lane, a.gelu_a);      // layer 1, h_E tile as stored (pre-split)
        }
        if (first_half && next_node) {                   // next node's neighbour list and P row
            nsrc = __builtin_amdgcn_readfirstlane(ninfo.x);
            nbase = __builtin_amdgcn_readfirstlane(ninfo.y);
            nK = __builtin_amdgcn_readfirstlane(ninfo.z);
            njA = a.E_idx[(size_t)nsrc * 64 + (c < nK ? c : 0)];
            njB = a.E_idx[(size_t)nsrc * 64 + (32 + c < nK ? 32 + c : 0)];
            npv = Prows[(size_t)n2 * 64 + lane];
        }
        const bool next_half = first_half && K > 32;
        {
            int rep_off = 0;
            asm volatile("" : "+v"(rep_off));   // or the 64 copies are read once, outside the loop, and spilled
            const float4 *br = bias_rep + rep_off;
#pragma unroll
            for (int bo = 0; bo < 4; ++bo)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 bv = br[q * 128 + 32 * bo];
                    t2.b[bo][4 * q + 0] = bv.x; t2.b[bo][4 * q + 1] = bv.y;
                    t2.b[bo][4 * q + 2] = bv.z; t2.b[bo][4 * q + 3] = bv.w;
                }
        }
        gemm128_h_lds<TERMS, true, true>(t2, acc, w2, lane, a.gelu_a);    // layer 2 on GELU(layer 1), output transposed
        {   // edge rows and Q rows of the next tile, in flight during the epilogue
            const int pn = next_half ? n : (next_node ? n2 : n), ps = next_half ? src : nsrc;
            const int pe = next_half ? (32 + c < K ? 32 + c : 0) : (c < nK ? c : 0);
            const int pq = next_half ? base + jB : nbase + njA;
            tile_load_edge<!HOISTED>(x, block_of(pn, ps), pe, h);
            tile_load_row(acc, a.Q + (size_t)pq * HD, h);
            __builtin_amdgcn_sched_barrier(0);
        }
        tile_gelu(t2, a.gelu_b);
        const int cnt = K - 32 * half;                   // valid edges of this tile (wave-uniform)
        if (cnt >= 32) {
#pragma unroll
            for (int bo = 0; bo < 4; ++bo) {
                f32x2 s2 = tile_pair(t2.b[bo], 0);      // packed adds: two partial sums per block
#pragma unroll
                for (int r = 2; r < 16; r += 2) s2 += tile_pair(t2.b[bo], r);
                sum[bo] += s2.x + s2.y;
            }
        } else {
#pragma unroll
            for (int bo = 0; bo < 4; ++bo) {
                float s0 = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) s0 += ((r & 3) + 8 * (r >> 2) + 4 * h < cnt) ? t2.b[bo][r] : 0.f;
                sum[bo] += s0;
            }
        }
        if (next_half) {
            half = 1;
            continue;
        }
#pragma unroll
        for (int bo = 0; bo < 4; ++bo) {
            const float tot = sum[bo] + __shfl_xor(sum[bo], 32);
            if (h == 0) a.S[(size_t)n * HD + 32 * bo + c] = tot;
            sum[bo] = 0.f;
        }
        if (!next_node) break;
        n = n2; src = nsrc; base = nbase; K = nK; jA = njA; jB = njB;
        Pslot[lane] = npv;
        half = 0;
    }
}

// ---------------------------------------------------------------------------------------------
// Node kernel: one wave = 32 nodes (columns).
//   MODE_IN : h_V = x_in(x)
//   MODE_UPD: h_V = mod2(LN(v + FFN(v))),  v = mod1(LN(h_V + (W3 @ S + K b3) / 30))
// then up to four 128x128 projections of the new h_V for the next edge kernels.
// ---------------------------------------------------------------------------------------------

template <int TERMS>
static void launch_msg_t(const EdgeArgs &ea, hipStream_t st) {
    static bool attr_set = false;     // one flag per TERMS instantiation
    constexpr int NW = 8;
    const size_t lds = 16 * (edge_lds_u4<false, NW>() + 4 * 128);     // + the replicated bias table
    static_assert(16 * (edge_lds_u4<false, NW>() + 4 * 128) <= 160 * 1024, "kernel exceeds the CU's LDS");
    if (!attr_set) {
        set_max_lds(reinterpret_cast<const void *>(msg_kernel_h<NW, false, TERMS>), lds);
        set_max_lds(reinterpret_cast<const void *>(msg_kernel_h<NW, true, TERMS>), lds);
        attr_set = true;
    }
    const int groups = (ea.n_nodes + NW - 1) / NW;
    dim3 grid(groups < edge_cus() ? groups : edge_cus()), block(NW * 64);
    if (ea.E1 != nullptr) hipLaunchKernelGGL((msg_kernel_h<NW, true, TERMS>), grid, block, lds, st, ea);
    else hipLaunchKernelGGL((msg_kernel_h<NW, false, TERMS>), grid, block, lds, st, ea);
}

void launch_edge_msg(int terms, const EdgeArgs &ea, hipStream_t st) {
    if (terms == 3) launch_msg_t<3>(ea, st);
    else launch_msg_t<4>(ea, st);
}
