// Shared by the translation units that hold edge kernels: argument struct, LDS layout and the XCD-aware
// node placement.  (The small-job tile kernels live in their own translation unit, edge_tile_kernels.hip: with
// them in the same unit hipcc allocated the per-node kernels' registers differently and spilled in their inner
// loops - scratch 20 -> 88 and 24 -> 116 bytes, -9 % - although their source was unchanged.)
#pragma once
#include "common.h"
#include "../../include/codlad_hip.h"

#define HD 128

void set_max_lds(const void *fn, size_t bytes);   // denoiser_kernels.hip: hipFuncSetAttribute, failure kept for the next check
int num_cu();
int edge_cus();     // persistent workgroups of an edge kernel: num_cu() unless CODLAD_OPT_EDGE_CUS says fewer

// ---------------------------------------------------------------------------------------------
// Edge kernels: one wave = one node = up to 64 neighbour columns (two 32-column passes).
// ---------------------------------------------------------------------------------------------
struct EdgeArgs {
    const int4 *node_info;
    const int32_t *E_idx;  // [n_snodes][64]
    const float *hE_in;    // [rows][64][128]
    int in_by_src;         // 1: rows indexed by structure node (h_E0), 0: by sample node
    const float *E1;       // encoder layer 0 only (may be null): W1e @ h_E0 per structure edge, hoisted
    float *hE_out;         // edge update only
    const float *P, *Q;    // [n_nodes][128]: own-node term (+bias), neighbour term
    const float *W1, *W2, *W3;
    const void *W1h, *W2h, *W3h;  // split-fp16 copies (precision 1, 2)
    const float *b2, *b3;
    const float *mods3;    // edge update: shift3, scale3, gate3 (3 x 128)
    float *S;              // message: [n_nodes][128]
    int n_nodes;
    // split-fp16 modes (block exponents, common.h): GELU constants for the input of layer 2 (scale 2^E1) and of
    // layer 3 / the message epilogue (2^(E1+E2)); edge update: the residual enters layer 3's accumulator as
    // h_E * res_scale (= 2^(E1+E2+E3)) and the LayerNorm runs with eps * res_scale^2 (exactly equivalent).
    // b2 / b3 point at biases pre-scaled to match.
    GeluK gelu_a, gelu_b;
    float res_scale, ln_eps;
};

// small jobs: the work dealt out per non-empty 32-edge tile {node, half} (edge_tile_kernels.hip, edge_wide_kernels.hip)
struct EdgeTileArgs : EdgeArgs {
    const int2 *tile_list;
    int n_tiles;
};

#define LDS_BLOCK_U4 4096   // one 64 KB packed block in 16-byte words

// XCD-aware node placement.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8
// share one, MI355X_MICROARCH.md "Workgroup dispatch"), each with a private 4 MB L2.  The node list
// is cut into 8 contiguous chunks and chunk b % 8 is served by the workgroups of that residue
// class, in the node kernel and in both edge kernels alike: the P/Q rows the node kernel writes,
// the neighbour rows an edge tile gathers (neighbours are nodes of the same sample, i.e. of the
// same chunk) and the edge state written by one edge kernel and read by the next then stay within
// one XCD's L2 instead of being pulled into all eight.  Placement is a speed matter only.
// Chunk granularity: large jobs use 256 nodes (the largest node-kernel workgroup: 8 waves x 32), so
// that the node kernel can follow the same chunks; below 32 768 nodes that would leave whole XCDs
// without work (eight chunks of a multiple of 256 nodes), so chunks are cut to 32 nodes and the node
// kernel keeps its plain order - at that size everything fits in any L2 anyway.
constexpr int NODE_WG_TILE = 256;
constexpr int XCD_CHUNKED_NODE_KERNEL_MIN = 32768;
__host__ __device__ inline int xcd_chunk_nodes(int n_nodes) {
    const int g = n_nodes >= XCD_CHUNKED_NODE_KERNEL_MIN ? NODE_WG_TILE : 32;
    const int tiles = (n_nodes + g - 1) / g;
    return g * ((tiles + 7) / 8);
}

struct NodeSpan {
    int first, end, stride;
};
// nodes first, first + stride, ... < end for wave `wave` of this workgroup
DEV NodeSpan wave_node_span(int n_nodes, int nwaves, int wave) {
    const int nb = gridDim.x, b = blockIdx.x;
    if (nb % 8) return {b * nwaves + wave, n_nodes, nb * nwaves};
    const int chunk = xcd_chunk_nodes(n_nodes);
    const int lo = (b % 8) * chunk, hi = lo + chunk < n_nodes ? lo + chunk : n_nodes;
    return {lo + (b / 8) * nwaves + wave, hi, (nb / 8) * nwaves};
}

// Small wave-uniform vectors (the centre node's P row, biases, modulation) are NOT read with
// per-lane global loads: a 128-float vector costs 16 dwordx4 instructions per lane whatever the
// addresses, and eight of those per tile were half of the kernel's traffic through the texture
// addresser (tools/ablate_edge.py).  They sit in LDS instead and are read as broadcasts.
//   message LDS (16-byte words): [W1e 4096][W2 4096][consts][P slots NWAVES x 32]
//   update  LDS                : [W12 4096][W13 4096][W11e first UPD_W1_KS k-steps][consts][P slots]
// Three blocks do not fit in 160 KB, so the update kernel streams part of ONE block from L2.  That
// block is W11e, the first contraction of a tile: its fragment loads are issued before the tile's
// own rows are even requested and run eight groups ahead (the registers are free at that point),
// which hides the L2 latency that a streamed W13 - needed last, with three tiles live - could not.
constexpr int UPD_W1_KS = 3;                              // k-steps of W11e that fit in LDS
constexpr int EDGE_CONST_U4 = 4 * 32;                     // b2, b3, modulate A, modulate B
template <bool EDGE_UPDATE, int NWAVES>
constexpr int edge_lds_u4() {
    return 2 * LDS_BLOCK_U4 + (EDGE_UPDATE ? UPD_W1_KS * 512 : 0) + EDGE_CONST_U4 + NWAVES * 32;
}

