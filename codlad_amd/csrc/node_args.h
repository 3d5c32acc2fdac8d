// Arguments of the node kernels (shared by denoiser_kernels.hip and node_wide_kernels.hip).
#pragma once
#include "edge_args.h"

struct NodeArgs {
    const int4 *node_info;
    int n_nodes;
    const float *x, *x_in_w, *x_in_b;  // MODE_IN
    const float *x_sc;                 // MODE_IN with self-conditioning: previous pred_xstart (null = zeros)
    int in_dim;                        // 3, or 6 = [x_self_cond | x] (latent_model.py:210-212)
    const float *S;                    // MODE_UPD
    float *hV;
    const float *W3, *b3;
    const float *mods;                 // shift1, scale1, gate1, shift2, scale2, gate2
    const float *Win[4], *Wout[4];
    const float *b_in, *b_out;
    int n_proj;
    const float *proj_w[4];
    const float *proj_b[4];            // may be null
    float *proj_out[4];
    int proj_flags[4];                 // bit0: input = h_V + h_Venc; bit1: += TS[z]
    const float *TS;                   // [30][128]
    const float *hVenc_in;
    float *hVenc_out;                  // if set: also store the new h_V here (h_Venc := h_V)
    int venc_is_self;                  // h_Venc == new h_V (first decoder layer's Q)
    // precision 1, 2: split-fp16 copies of the blocks in execution order: [W3, Win0, Wout0, .., Wout3,] proj0..
    const void *blk_h[13];
    // block exponents (split-fp16 modes; all 1 / plain in the fp32 mode): S arrives scaled by the message MLP's
    // accumulated exponent and is contracted as S * s_scale (= 2^-(E1+E2) / 64); the W3 term comes out as
    // t * 2^e3 / 64 and is added as (t * t_scale) / 30 (t_scale = 64 * 2^-e3); the FFN output carries
    // 2^(e_in+e_out) and is added as t * ffn_scale.  b3, b_in, b_out, proj_b, TS are pre-scaled to match.
    float s_scale, t_scale, ffn_scale;
    GeluK gelu_ffn;
    int s_partials;                    // 1: S holds one partial sum per half, S[2][n_nodes][128] (tile-wise message kernel)
};

