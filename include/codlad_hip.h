/*
 * libcodlad_hip.so - C ABI of the MI355X (gfx950) sampling hot path of CODLAD.
 *
 * The reference (pure PyTorch) has no FFI layer; its boundary for this path is a set of
 * Python call signatures.  Each entry point below names the reference
 * code it replaces.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - fp32 row-major, indices int32 unless stated, VQ indices int64;
 *   - the caller owns every buffer (inputs, outputs, workspace); nothing is allocated here;
 *   - `stream` is a hipStream_t (NULL = default stream); calls only enqueue work;
 *   - return 0 = ok, < 0 = argument error, > 0 = hipError_t; codlad_last_error() has text;
 *   - not thread-safe per handle; one process per GPU.
 *
 * Ragged layout ("job")
 *   A job is S samples, sample s has L_s residues ("nodes").  Node arrays are flat,
 *   sample-major: n_nodes = sum L_s.  Several samples may share one structure
 *   (ensemble members of one frame): structure arrays are flat over structure nodes
 *   (n_snodes = sum over structures of L_f).  No padding anywhere, so there are no masks.
 *   node_info[n] = {src, base, K, z}:
 *       src  = flat structure-node index of node n (row of E_idx / h_E0 / cg_z),
 *       base = flat index of the first node of n's sample,
 *       K    = min(64, L_s) neighbours, z = residue type (0..29).
 *   E_idx[src][k], k < K, is the neighbour's index inside its own structure (0..L-1).
 */
#ifndef CODLAD_HIP_H
#define CODLAD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CODLAD_ABI_VERSION 13
#define CODLAD_H 128          /* hidden width of the denoiser                          */
#define CODLAD_KNN 64         /* k_neighbors (reference models/latent_model.py:86)      */
#define CODLAD_MODS_PER_STEP 6016 /* 3*9*128 (enc) + 3*6*128 (dec) + 2*128 (final)      */
#define CODLAD_PACKED_BLOCK 16384 /* floats of one 128x128 block in MFMA operand order  */

int codlad_abi_version(void);
const char *codlad_last_error(void);
/* sizeof(codlad_denoiser_weights), sizeof(codlad_decoder_weights), sizeof(codlad_workspace),
 * offsetof(codlad_denoiser_weights, precision), offsetof(.., enc_h): lets a binding verify its
 * struct mirrors. */
void codlad_struct_sizes(int *out5);

/* 128x128 weight block -> MFMA A-operand order (host helper; src/dst are HOST pointers).
 * dst[(16*b + r)*256 + lane*4 + bo] = src[(32*bo + (lane&31))*ld + 32*b + (r&3) + 8*(r>>2) + 4*(lane>>5)]
 * scaled by `scale`. */
void codlad_pack_block_host(const float *src_host, int ld, float scale, float *dst_host);

/* Encoder layer weights.  W* are packed 128x128 blocks (codlad_pack_block_host), b* plain. */
typedef struct {
    const float *W1e, *W2, *W3;        /* message MLP: W1[:,128:256], W2, W3                 */
    const float *W11e, *W12, *W13;     /* edge-update MLP: W11[:,128:256], W12, W13           */
    const float *W1a, *W1c;            /* W1[:,0:128] (own node), W1[:,256:384] (neighbour)   */
    const float *W11a, *W11c;
    const float *Win[4], *Wout[4];     /* dense.W_in rows 128c.., dense.W_out cols 128c..     */
    const float *b1, *b2, *b3, *b11, *b12, *b13, *b_in /*512*/, *b_out;
} codlad_enc_layer;

typedef struct {
    const float *W1e;                  /* 2 * W1[:,128:256]  (h_ESV doubles h_E)              */
    const float *W2, *W3;
    const float *W1a;                  /* W1[:,0:128]                                          */
    const float *W1v;                  /* W1[:,384:512]  applied to h_V_j + h_Venc_j           */
    const float *TS;                   /* [30][128]: W1[:,256:384] @ (2*W_s[z])                */
    const float *Win[4], *Wout[4];
    const float *b1, *b2, *b3, *b_in, *b_out;
} codlad_dec_layer;

/* Split-fp16 copies of the 128x128 blocks: each weight split into hi + lo fp16 halves, 64 KB per
 * block in the order [k-step 0..7][out block 0..3][hi,lo][lane 0..63][8 halves]
 * (codlad_amd/csrc/common.h).  Used when codlad_denoiser_weights.precision is 1 or 2.
 *
 * BLOCK EXPONENTS.  hi + lo reproduces a value to max(2^-22 |x|, 2^-25): the bound is the relative one only
 * while the `lo` half is a normal fp16 (|x| >~ 2^-3).  So that this holds whatever the scale of a layer's
 * weights, every block is stored multiplied by a power of two 2^e chosen when the weights are packed (one e
 * per weight matrix of the reference: e1 for the three slices of W1, e2 for W2, ...; codlad_amd/weights.py),
 * the biases are stored pre-multiplied by the power of two their accumulator carries, and the activations
 * between the layers of an MLP carry the accumulated exponent through an exactly scale-equivariant GELU
 * evaluation.  The scale leaves where that is free (a LayerNorm, a constant multiply).  Exact power-of-two
 * scaling throughout: with every e = 0 the arithmetic is the unscaled one bit for bit.  Consequences visible
 * at this boundary: in the split modes the workspace's S and PQ buffers and E1 hold scaled values
 * (S: 2^(e1+e2) of the layer that wrote it; PQ: 2^e1 / 2^e11; E1[0] / E1[1]: 2^e1 / 2^e11 of encoder layer 0). */
typedef struct {
    const void *W1e, *W2, *W3, *W11e, *W12, *W13, *W1a, *W1c, *W11a, *W11c;
    const void *Win[4], *Wout[4];
    /* 2^e1 b1, 2^(e1+e2) b2, 2^e3 b3, 2^e11 b11, 2^(e11+e12) b12, 2^(e11+e12+e13) b13, 2^e_in b_in [512],
     * 2^(e_in+e_out) b_out */
    const float *b1, *b2, *b3, *b11, *b12, *b13, *b_in, *b_out;
    int e1, e2, e3, e11, e12, e13, e_in, e_out;
} codlad_enc_layer_h;

typedef struct {
    const void *W1e, *W2, *W3, *W1a, *W1v;
    const void *Win[4], *Wout[4];
    const float *TS;                   /* 2^e1 TS                                              */
    const float *b1, *b2, *b3, *b_in, *b_out;   /* scaled like the encoder's                  */
    int e1, e2, e3, e_in, e_out;
} codlad_dec_layer_h;

/* Replaces the parameters of reference models/latent_model.py:119-148 (ProteinMPNN_diffusion_new). */
typedef struct {
    const float *freqs;                /* [128] exp(-ln(1e4) k/128)  (latent_model.py:62-64)   */
    const float *rbf_mu;               /* [16] linspace(2, 22, 16)  (protein_mpnn_utils.py:464-465) */
    const float *t_w0, *t_b0;          /* t_embedder.mlp.0  [128][256], [128]                  */
    const float *t_w2, *t_b2;          /* t_embedder.mlp.2  [128][128], [128]                  */
    const float *ada_w[7], *ada_b[7];  /* adaLN heads: enc0..2 [1152][128], dec0..2 [768][128], final [256][128] */
    const float *x_in_w, *x_in_b;      /* [128][3] ([128][6] with self_condition), [128]       */
    const float *pos_w, *pos_b;        /* features.embeddings.linear [16][66], [16]            */
    const float *edge_wT;              /* features.edge_embedding.weight TRANSPOSED [167][128] */
    const float *norm_w, *norm_b;      /* features.norm_edges                                   */
    const float *We_wT, *We_b;         /* W_e.weight TRANSPOSED [128][128], [128]              */
    const float *out_w, *out_b;        /* W_out.linear [out_dim][128], [out_dim]               */
    codlad_enc_layer enc[3];
    codlad_dec_layer dec[3];
    /* 0: contractions on v_mfma_f32_32x32x2_f32 (exact fp32 products);
     * 1: f16x4 - both operands split into two fp16 halves (representation error max(2^-22 |x|, 2^-25),
     *    kept in its relative regime by the block exponents below), the four cross products on
     *    v_mfma_f32_32x32x16_f16, fp32 accumulate;
     * 2: f16x3 - the same without the lo x lo product, which is itself <= 2^-22 of the result: a
     *    quarter fewer matrix instructions at the same measured deviation from mode 0.
     *    The default of the Python host layer.
     * Operands beyond the fp16 range (65504) cannot be split: weights are refused when packed, activations
     * are caught by the status word (codlad_status_check). */
    int precision;
    codlad_enc_layer_h enc_h[3];
    codlad_dec_layer_h dec_h[3];
    /* 1: the model was built with self_condition=True (latent_model.py:112-116): x_in takes
     * cat(x_self_cond, x) and the sampler feeds each step the previous step's pred_xstart
     * (gaussian_diffusion.py:530-547). */
    int self_condition;
    /* rows of W_out.linear: 6 = (eps | variance logits) of a diffusion model, 3 = the velocity of a flow-matching
     * model (latent_model.py:142-143: input_size is doubled for diffusion == "diffusion" only) */
    int out_dim;
} codlad_denoiser_weights;

/* Row 4: CA_ProteinFeatures.forward + W_e
 * (reference models/protein_mpnn_utils.py:478-523, latent_model.py:208,216).
 * snode_info[m] = {start, L} of the structure that structure-node m belongs to.
 * Writes E_idx [n_snodes][64] (ascending distance, self first) and h_E0, one EDGE BLOCK per
 * structure node.  An edge block holds 64 edges x 128 features, the two halves of 32 edges one after the
 * other and "chunk-major" within a half: [2 halves][32 chunks of 4 features][32 edges][4 floats] (feature f
 * of edge e at float 4096*(e/32) + 128*(f/4) + 4*(e%32) + f%4), so that the 16 bytes neighbouring lanes
 * (edges) move per instruction are neighbours in memory and a 32-edge tile is 16 contiguous KB; slots of
 * edges e >= K are never written.  E1 and the workspace's hE use the same block layout.
 * Contents of a 16-byte slot: four fp32 features in the fp32-MFMA mode (precision 0) and for E1 in every mode; in the
 * split-fp16 modes (precision 1, 2) h_E0 and hE are stored as the fp16 halves the contractions consume ("pre-split", same
 * bytes): for b = 0..3, s = 0..1, h = 0..1 the eight features 32 b + 16 s + 4 h + {0,1,2,3, 8,9,10,11} of an edge keep
 * their eight `hi` halves (fp16 of the value, round to nearest) in chunk slot 8 b + 4 s + h and their eight `lo` halves
 * (fp16 of value - hi) in chunk slot 8 b + 4 s + 2 + h; the value is hi + lo (22 significant bits).  The edge state is
 * internal to a (structures, workspace) pair: a caller never reads it, and one that changes w->precision must call
 * codlad_features_prepass again (the Python layer does). */
int codlad_features_prepass(const codlad_denoiser_weights *w, const float *cg_xyz,
                            const int32_t *snode_info, int n_snodes, int max_len,
                            int32_t *E_idx, float *h_E0, void *stream);

/* Row 3: TimestepEmbedder + every adaLN head for n_t timesteps
 * (latent_model.py:37-75; protein_mpnn_utils.py:238,298; latent_model.py:32).
 * t_values: DEVICE int64 [n_t] (already mapped through timestep_map).  mods [n_t][6016]. */
int codlad_step_mods(const codlad_denoiser_weights *w, const int64_t *t_values, int n_t,
                     float *mods, void *stream);

/* The same for FRACTIONAL timesteps (device float [n_t]): the flow-matching sampler evaluates the model at
 * t in [0, 1] (reference test.py:214-250; latent_model.py:66 multiplies t.float() into the frequencies). */
int codlad_step_mods_f(const codlad_denoiser_weights *w, const float *t_values, int n_t, float *mods, void *stream);

/* Next row 8f-4 (flow matching / ODE sampling, reference test.py:214-250 -> torchdiffeq.odeint): the update of
 * an explicit Runge-Kutta step, out[i] = y[i] + sum_j k[j][i] * (coef[j] * h), j < n_k <= 7, each operation
 * rounded separately, summed left to right.  k_host: HOST array of n_k DEVICE pointers; coef_host: HOST floats. */
int codlad_ode_combine(const float *y, const float *const *k_host, const float *coef_host, int n_k, float h,
                       size_t n, float *out, void *stream);

/* Workspace of one job, all caller-allocated. */
typedef struct {
    float *hV;      /* [n_nodes][128]                 */
    float *hVenc;   /* [n_nodes][128]                 */
    float *S;       /* [4][n_nodes][128]: neighbour sums (planes 1-3: partial sums of the small-job kernels) */
    float *PQ;      /* [4][n_nodes][128]              */
    float *hE;      /* [n_nodes] edge blocks (64 x 128) */
    int32_t *status; /* [1] sticky status word (CODLAD_STATUS_*), may be NULL: set by the kernels, never
                      * cleared by them; read and cleared by codlad_status_check                       */
    /* optional (NULL, 0 = none): the job's non-empty 32-edge tiles {node, half}, node-major - half 1 listed
     * only for nodes with K > 32.  Small jobs (CODLAD_OPT_EDGE_TILE_MAX_NODES) deal the edge kernels' work out
     * per tile with it, which doubles the number of busy waves; results are bit-identical to the per-node
     * order (the partial neighbour sums are kept apart and added in that order). */
    const int32_t *tile_list;
    int32_t n_tiles;
} codlad_workspace;

/* Status bits.  NONFINITE: a denoiser output (eps | variance logits) was inf or NaN.  In the split-fp16
 * contraction modes that is also what an operand beyond the fp16 range (|x| > 65504) ends in: its halves
 * become hi = +-inf, lo = -+inf and every product they enter NaN, which LayerNorm spreads over the node and
 * message passing over the sample - an overflow cannot come out as a finite number, so this one test at the
 * end of every forward is the overflow sentinel, at no cost in the contraction loops. */
#define CODLAD_STATUS_NONFINITE 1
#define CODLAD_E_NONFINITE (-3)

/* Synchronises `stream`, reads the status word and clears it.  Returns 0 if it was clear,
 * CODLAD_E_NONFINITE (with codlad_last_error() text) if CODLAD_STATUS_NONFINITE was set, > 0 = hipError_t.
 * The only entry point that waits for the device. */
int codlad_status_check(int32_t *status, void *stream);

/* Step- and member-invariant part of encoder layer 0: E1[0] = W1[:,128:256] @ h_E0 (message) and
 * E1[1] = W11[:,128:256] @ h_E0 (edge update) per structure edge, E1 [2][n_snodes] edge blocks.
 * Optional: passing E1 = NULL below makes the layer-0 kernels contract h_E0 themselves. */
int codlad_layer0_edge_terms(const codlad_denoiser_weights *w, const int32_t *snode_info,
                             int n_snodes, const float *h_E0, float *E1, void *stream);

/* Rows 5-7: one denoiser forward (latent_model.py:175-268): x [n_nodes][3] -> out [n_nodes][6].
 * (out [n_nodes][3] for a flow-matching model, out_dim 3).
 * mods_t = the 6016 modulation floats of this timestep.  E1 (may be NULL) from
 * codlad_layer0_edge_terms, n_snodes = its structure-node count.  x_self_cond [n_nodes][3]: only for
 * a self_condition model, NULL = zeros (latent_model.py:211). */
int codlad_denoiser_forward(const codlad_denoiser_weights *w, const int32_t *node_info,
                            int n_nodes, const int32_t *E_idx, const float *h_E0,
                            const float *E1, int n_snodes, const float *x, const float *x_self_cond,
                            const float *mods_t, float *out, const codlad_workspace *ws, void *stream);

/* Row 2: one reverse step given the model output (gaussian_diffusion.py:404-449, 262-360).
 * coef_host[8] = {sqrt_recip_acp, sqrt_recipm1_acp, post_coef1, post_coef2,
 *                 post_log_var_clipped, log_beta, nonzero(0/1), mode} for this step.
 * mode selects p_mean_variance's branches (gaussian_diffusion.py:303-349), as a small integer kept in a float:
 *   bit 1  the model predicts x_0 (ModelMeanType.START_X; test.py --predict_xstart) instead of the noise
 *   bit 2  fixed variance (ModelVarType.FIXED_SMALL / FIXED_LARGE; create_diffusion(learn_sigma=False)): entry 4 holds the
 *          step's log variance itself and model_out is [n_nodes][3] (no variance channels); otherwise [n_nodes][6]
 *   bit 4  clip_denoised: pred_xstart clamped into [-1, 1]
 * mode 0 = epsilon prediction, learned-range variance, no clipping: what test.py samples with. */
int codlad_ddpm_update(const float *x, const float *model_out, const float *noise,
                       const float *coef_host, int n_nodes, float *x_out, float *x_start_out /* pred_xstart, may be NULL */,
                       void *stream);

/* Rows 2-7 fused: p_sample_loop (gaussian_diffusion.py:451-547, respace.py:124-129).
 * x [n_nodes][3] holds x_T on entry and x_0 on return.  noise [T][n_nodes][3] is consumed in
 * loop order (entry 0 at step T-1).  mods [T][6016] and coef [T][8] (device) are indexed by
 * respaced step i; the loop runs i = T-1 .. 0.  x_start [n_nodes][3] (may be NULL unless the model
 * is self-conditioned) receives every step's pred_xstart and is what the next step is conditioned on. */
int codlad_sample_loop(const codlad_denoiser_weights *w, const int32_t *node_info, int n_nodes,
                       const int32_t *E_idx, const float *h_E0, const float *E1, int n_snodes,
                       float *x, float *x_start, const float *noise, const float *mods, const float *coef,
                       int T, const codlad_workspace *ws, void *stream);

/* Row 8: get_norm_feature(norm_in=False) + nearest code
 * (utils/dataset_module.py:253; utils/vq_module.py:61-68 / VectorQuantize eval lookup).
 * x [n][3] normalised samples -> latent = x*std+mean; idx int64 [n]; z_q [n][3]. */
int codlad_vq_lookup(const float *x, int n, const float *mean3, const float *std3,
                     const float *codebook, int n_codes, int64_t *idx, float *z_q,
                     float *latent_out /* may be NULL */, void *stream);

/* IC decoder weights (reference models/vae_model.py:318-373, 414-465), plain row-major. */
typedef struct {
    int angle;                          /* 0 = IC_Decoder (N6), 1 = IC_Decoder_angle (K3/K4)  */
    const float *map_out_w, *map_out_b; /* [36][3], [36]                                       */
    const float *res_embed;             /* [25][4]                                             */
    const float *inv0_w[4], *inv0_b[4], *inv1_w[4], *inv1_b[4];   /* [40][40]                  */
    const float *dist_w[4], *dist_b[4];                           /* [40][15]                  */
    const float *dense1_w[4], *dense1_b[4], *dense3_w[4], *dense3_b[4];
    const float *bb_dist, *sc_dist;     /* [25][3], [25][10]                                   */
    const float *bb_ang1_w, *bb_ang1_b, *bb_ang3_w, *bb_ang3_b;   /* [3][40], [3][3]           */
    const float *sc_angle_emb;          /* [25][10]   (angle == 0)                             */
    const float *sc_ang1_w, *sc_ang1_b, *sc_ang3_w, *sc_ang3_b;   /* [10][40], [10][10] (angle == 1) */
    const float *bb_tor1_w, *bb_tor1_b, *bb_tor3_w, *bb_tor3_b;   /* [3][43], [3][3]           */
    const float *tor1_w[4], *tor1_b[4], *tor3_w[4], *tor3_b[4];   /* [F][F], F = 40 or 50      */
    const float *fin1_w, *fin1_b, *fin3_w, *fin3_b;               /* [10][F], [10][10]         */
} codlad_decoder_weights;

/* Row 9: VAE.decoder = map_out + IC_Decoder[_angle].forward (vae_model.py:759-764, 375-412, 467-503).
 * z_q [M][3] (w->map_out_w given: N6 / K3 / K4) or [M][36] (map_out_w NULL: the C2 model, whose IC decoder takes the
 * 36-wide latent as it is, vae_model.py:556-561), cg_z int32 [M], cg_xyz [M][3]; directed CG graph in CSR over receiving node:
 * csr_ptr int32 [M+1], csr_src int32 [E_dir] (sending node of each incoming edge, flat index).
 * scratch: float [M][200].  ic_out [M][13][3]. */
int codlad_ic_decode(const codlad_decoder_weights *w, const float *z_q, const int32_t *cg_z,
                     const float *cg_xyz, const int32_t *csr_ptr, const int32_t *csr_src,
                     int M, float *scratch, float *ic_out, void *stream);

/* Next row 8f-3 (host preprocessing): CG neighbour list within `cutoff` (reference
 * utils/protein_module.py:567-584) + make_directed + scatter order (models/gcn_nn.py:54-64,
 * models/vae_model.py:485-488), as the CSR codlad_ic_decode consumes.  sample_range[i] = {first, L}
 * of the sample that flat node i belongs to.  Two passes: degree != NULL counts the directed edges
 * arriving at every node; after an exclusive scan into csr_ptr, the second call (degree == NULL)
 * writes csr_src (for node i: senders j > i ascending, then j < i ascending). */
int codlad_cg_graph(const float *cg_xyz, const int32_t *sample_range, int M, float cutoff,
                    int32_t *degree, const int32_t *csr_ptr, int32_t *csr_src, void *stream);

/* Next row 8f-1: the e3nn encoder / CG prior in front of the decoder (reference models/vae_model.py:21-311; what
 * VAE.get_latent_wovq and get_latent_cg run, reference test.py:495,501).
 *
 * codlad_tp_conv = one TensorProductConvLayer.forward (reference models/gcn_nn.py:176-219; residual=False, no batch
 * norm, reduce='mean') over a set of receiving nodes, fused with everything that feeds it per edge: the distance
 * r = r_sign * (xyz_snd[s] - xyz_recv[n]), its Gaussian smearing over [0, smear_stop] (8 centres, gcn_nn.py:163-173),
 * the edge-embedding MLP Linear(emb_in, 12) -> ReLU -> Linear(12, 12) on [type_recv, type_snd, 0, 0, 0, 0, smearing]
 * (emb_in 14; vae_model.py:164-194) or on the smearing alone (emb_in 8: the atom <-> bead cross graph, :196-201), the
 * real spherical harmonics of r up to l = 2 ('component' normalisation), fc = Linear(36, 36) -> ReLU ->
 * Linear(36, weight_numel) on [edge embedding | h[.., :12] | h[.., :12]] and o3.FullyConnectedTensorProduct(irreps of
 * `depth`, 1x0e + 1x1o + 1x2e, irreps of depth + 1) restated from e3nn 0.5.1's definition (e3nn is not part of the
 * reference tree: parity unpinned except for the Wigner symbols, see oracle/e3nn_lite.py).
 *   feature layout of depth d: [12x0e | 4x1o | 4x1e | 12x0o] truncated to 12 (d + 1) floats, vectors as (u, xyz);
 *   ptr int32 [n_recv + 1], snd int32 [E]: the receivers' CSR (edges of receiver n: ptr[n] .. ptr[n+1]), snd = the
 *     node whose features h_snd travel along the edge (e3nn's edge_dst; the receiver is its edge_src);
 *   attr_recv_first: 1 = fc sees [e | h_recv[:12] | h_snd[:12]] (intra graphs, bead -> atom), 0 = [e | h_snd | h_recv]
 *     (atom -> bead: reference vae_model.py:140-142 passes the same concatenation to both cross directions);
 *   out [n_recv][12 (depth + 2)]: accumulate 0: out = pad(h_recv) + mean, 1: out += mean (the layer's second update);
 *   group: lanes per receiving node (64, 16 or 1: pick >= the typical degree; any degree is correct with any group);
 *     64 selects the matrix-pipe kernel (CODLAD_OPT_TP_CONV_VARIANT). */
typedef struct {
    const int32_t *ptr, *snd;
    int32_t n_recv;
    const float *xyz_recv, *xyz_snd;       /* [n][3] */
    const float *typ_recv, *typ_snd;       /* node types as floats (emb_in 14) or NULL (emb_in 8) */
    float r_sign, smear_stop;
    const float *emb0_w, *emb0_b, *emb3_w, *emb3_b;
    int32_t emb_in;
    const float *h_recv; int32_t d_recv;   /* receiving nodes' features, row stride d_recv (>= 12) */
    const float *h_snd; int32_t d_snd;     /* sending nodes' features, row stride = width of `depth` */
    int32_t attr_recv_first;
    const float *fc0_w, *fc0_b, *fc3_w, *fc3_b;
    int32_t depth;                         /* 0, 1, 2 */
    float *out;
    int32_t accumulate, group;
    const void *packed;                    /* NULL, or the layer's weights as codlad_tp_conv_pack left them (below) */
} codlad_tp_conv_args;
int codlad_tp_conv(const codlad_tp_conv_args *args, void *stream);
int codlad_tp_conv_args_size(void);      /* sizeof(codlad_tp_conv_args), for bindings to check their layout */
/* The matrix-pipe kernel keeps a layer's weights (fc.0, fc.3, the edge embedding) in LDS as split-fp16 operand fragments.
 * With packed == NULL every workgroup of every launch builds that image from the fp32 weights (18-35 us, most of a small
 * graph's launch); codlad_tp_conv_pack builds it ONCE into `image` (codlad_tp_conv_image_bytes(depth) bytes of device
 * memory, 16-byte aligned) from the weight pointers, depth and emb_in of `args` (graph fields are not read), and launches that
 * pass it as `packed` copy it.  The image depends on nothing else; results are bit-identical either way. */
int codlad_tp_conv_image_bytes(int depth);
int codlad_tp_conv_pack(const codlad_tp_conv_args *args, void *image, void *stream);

/* The receivers' CSR codlad_tp_conv reads, from a pair list (int64 [n_pairs][2], node indices < n_nodes; what the reference's
 * make_directed, models/gcn_nn.py:54-64, and its scatter do on the host).  mode 0: edge (a, b) = receiver a, sender b, and -
 * unless the list already holds pairs with a > b AND pairs with b > a - the reversed edges as well (make_directed's rule);
 * mode 1: the list is directed as given.  ptr int32 [n_nodes + 1], snd int32 [2 n_pairs] (ptr[n_nodes] entries used), senders
 * ascending inside a receiver (a fixed order: results do not depend on the atomics).  work: int32 [2 n_nodes + 2 + 4 n_pairs].
 * Node indices are not range-checked on the device: the caller guarantees 0 <= index < n_nodes. */
int codlad_receiver_csr(const int64_t *pairs, int n_pairs, int n_nodes, int mode, int32_t *ptr, int32_t *snd, int32_t *work,
                        void *stream);

/* y[i] = W2 act(W1 x[i] + b1) + b2 (hidden <= 36; hidden 0: y = W2 x + b2), act 0 tanh / 1 relu, in_dim 84 / 48 / 36,
 * out_dim <= 36; mode 1: y = 1e-9 + exp(y / 2) (the prior's H_sigma, vae_model.py:263-265). */
int codlad_mlp_rows(const float *x, int n, int in_dim, const float *w1, const float *b1, int hidden, const float *w2,
                    const float *b2, int out_dim, int act, int mode, float *y, void *stream);
/* node[I] = [mean of h_atom over the atoms of bead I (48) | h_cg[I] (36)]  (vae_model.py:158-160) */
int codlad_bead_mean(const float *h_atom, const float *h_cg, const int32_t *bead_ptr, const int32_t *bead_atoms,
                     int n_cg, float *node, void *stream);
/* out[i] = table[idx[i]] (nn.Embedding rows of `width` floats) */
int codlad_embed_rows(const float *table, const int32_t *idx, int n, int width, float *out, void *stream);

/* Row 10: ic_to_xyz (utils/utils_ic.py:242-268).  ca_full [B][L+2][3] (flanking residues
 * included), ic [B][L][13][3], orders int32 [10][L][3] (atom_orders), slot_to_out int32 [L*14]
 * (output atom index of each residue slot, -1 = slot unused; derived from info's
 * atom_idx/permute).  xyz_out [B][n_atoms][3]. */
int codlad_ic_to_xyz(const float *ca_full, const float *ic, const int32_t *orders,
                     const int32_t *slot_to_out, int B, int L, int n_atoms, float *xyz_out,
                     void *stream);

/* The same for several proteins (each its own L, atom tables and output) in ONE launch: `groups` is a DEVICE array of
 * n_groups descriptors, first_row = the number of (frame, residue) rows of the groups before it (ascending from 0),
 * total_rows = the sum of B * L.  A job of many short proteins otherwise pays one latency-bound launch per protein. */
typedef struct {
    const float *ca_full, *ic;
    const int32_t *orders, *slot_to_out;
    float *xyz_out;
    int32_t B, L, n_atoms, first_row;
} codlad_xyz_group;
int codlad_ic_to_xyz_groups(const codlad_xyz_group *groups, int n_groups, int total_rows, void *stream);

/* The inverse, for building data sets from coordinates (utils/protein_module.py:770-774: get_backbone_ic / get_sidechain_ic
 * of utils/utils_ic.py:141-196, there on mdtraj / numpy).  xyz [n_frames][n_atoms][3]; quads int32 [n_quads][4] = atoms
 * (A1, A2, A3, A4) of each internal coordinate, any index < 0 = the slot does not exist (zeros); ic_out
 * [n_frames][n_quads][3] = (|A1 - A2|, angle(A1 - A2, A3 - A2), dihedral(A1, A2, A3, A4)), angle and dihedral in [0, 2 pi). */
int codlad_xyz_to_ic(const float *xyz, int n_frames, int n_atoms, const int32_t *quads, int n_quads, float *ic_out,
                     void *stream);

/* Tuning switches (speed only: every setting computes the same values).  Defaults suit MI355X; the environment
 * variable of the same name (CODLAD_ prefix, upper case) sets the initial value.
 *   CODLAD_OPT_NODEQ_MAX_TILES   jobs of up to this many 32-node tiles run the node update on the small-job
 *                                "quarter" kernel (one tile per 4-wave workgroup), larger ones on the streaming one
 *   CODLAD_OPT_NODE_QUAD_MAX_TILES  jobs of up to this many 32-node tiles run the node UPDATE on four waves per tile with the
 *                                whole register file (a ring of six weight quarters in flight: node_quad_kernels.hip); same bits
 *   CODLAD_OPT_EDGE_TILE_MAX_NODES  jobs of up to this many nodes deal the edge kernels' work out per 32-edge
 *                                tile instead of per node (twice the waves for the same work)
 *   (slot 2 is unused: the round-2 header reserved it for a captured / persistent step loop that was never built)
 *   CODLAD_OPT_DEC_EDGE_VARIANT  IC decoder messages: 0 = one sine / cosine + recurrence, 15 -> 40 filter on the f16 matrix
 *                                pipe (split fp16, fp32-equivalent); 1 = 15 library sines and fp32 FMAs (round-2 kernel).
 *                                NOT bit-identical to each other (both within the decoder's parity tolerance)
 *   CODLAD_OPT_TP_CONV_VARIANT   codlad_tp_conv with group = 64: 0 = fc.0 / fc.3 on the f16 matrix pipe (split fp16,
 *                                fp32-equivalent; a wave per receiving node, 32 edges per step); 1 = the scalar-operand
 *                                kernel that also serves groups 1 and 16.  Not bit-identical to each other either
 *   CODLAD_OPT_EDGE_UPD_VARIANT  edge update of large jobs: 0 = two waves per SIMD (upd_kernel_h); 1 = one wave per SIMD with
 *                                the next tile prefetched (upd1_kernel_h) where nearly every node has two 32-edge tiles;
 *                                2 = that kernel for every job.  Bit-identical; same speed on MI355X (profiles/r04_upd1_*)
 *   CODLAD_OPT_EDGE_CUS          persistent workgroups of the per-node edge kernels (0 / >= CU count: one per CU)
 *   CODLAD_OPT_EDGE_WIDE_MAX_TILES  tile-wise jobs of up to this many 32-edge tiles give a tile to FOUR waves (one output
 *                                block each, weight quarters in registers: edge_wide_kernels.hip) instead of one; same bits */
#define CODLAD_OPT_NODEQ_MAX_TILES 0
#define CODLAD_OPT_EDGE_TILE_MAX_NODES 1
#define CODLAD_OPT_NODE_QUAD_MAX_TILES 2
#define CODLAD_OPT_DEC_EDGE_VARIANT 3
#define CODLAD_OPT_TP_CONV_VARIANT 4
#define CODLAD_OPT_EDGE_UPD_VARIANT 5
#define CODLAD_OPT_EDGE_CUS 6
#define CODLAD_OPT_EDGE_WIDE_MAX_TILES 7
#define CODLAD_N_OPTIONS 8
int codlad_set_option(int option, int value);

/* Measurement aid for bench.py (not part of the reference's interface): while enabled, every edge-kernel launch made
 * by codlad_denoiser_forward / codlad_sample_loop is bracketed by a pair of HIP events on its stream (the first 4096
 * launches after enabling; enabling clears earlier records).  codlad_probe_read waits for the recorded events and
 * returns how many launches of `kind` were recorded (0 message, 1 edge update, +2 = the hoisted layer-0 variant) with
 * their summed duration in *total_ms; negative = error.  Not thread-safe; one stream at a time. */
int codlad_probe_edge_launches(int enable);
int codlad_probe_read(int kind, double *total_ms);

/* Measurement hook: ONE launch of the message kernel (which = 0) or the edge-update kernel
 * (which = 1) of encoder layer `layer` (0 or 1) on a job whose workspace holds the state of a
 * previous forward; both GEMM layers are executed (no E1 shortcut).  Layer 0 reads the shared h_E0,
 * layer 1 the per-sample edge state (the HBM-resident case that 5 of a step's 6 message launches are).
 * Used by bench.py to time the dominant kernel. */
int codlad_bench_edge_launch(const codlad_denoiser_weights *w, const int32_t *node_info,
                             int n_nodes, const int32_t *E_idx, const float *h_E0,
                             const float *mods_t, const codlad_workspace *ws, int which,
                             int layer, void *stream);

/* Next row 8f-2: the evaluation helpers that follow the path in the reference's loop (test.py:589-593):
 * recon_result (test.py:153-166), xyz_result (:148-151), ged_result (:141-146), clash_result (:118-139),
 * inter_result (:97-116).  Index lists are int64 as in the reference batch; a list may be empty (NULL, 0).
 * clash_list = the rows of cat(edge_list, nbr_list) that occur exactly once (the reference derives them
 * with unique(dim=0, return_counts=True) on every call; they depend on the topology only).
 * ic / ic_recon [n_ic][3] = (bond, angle, torsion) per slot, ic_mask [n_ic]. */
typedef struct {
    const float *xyz_recon, *xyz;      /* [n_atoms][3] */
    int64_t n_atoms;
    const int64_t *edge_list;          /* [n_edges][2] */
    int64_t n_edges;
    const int64_t *clash_list;         /* [n_clash][2] */
    int64_t n_clash;
    const int64_t *bb_NO_list;         /* [n_bb][2] */
    int64_t n_bb;
    const int64_t *interaction_list;   /* [n_inter][2] */
    int64_t n_inter;
    const int64_t *pi_pi_list;         /* [n_pipi][4] */
    int64_t n_pipi;
    const float *ic, *ic_recon, *ic_mask;
    int64_t n_ic;
} codlad_metric_inputs;

/* out8 = {loss_bond, loss_angle, loss_torsion, loss_xyz, loss_graph, loss_nbr, loss_inter, loss_pi_pi}
 * (device floats).  scratch: codlad_metrics_scratch_bytes() of device memory.  Deterministic. */
int codlad_metrics_scratch_bytes(void);
int codlad_eval_metrics(const codlad_metric_inputs *in, float *out8, void *scratch, void *stream);

/* Next row 8f-2: bond-graph validity, valid_ratio_and_cut_off_result (test.py:168-188) ->
 * eval_sample_qualities / count_valid_graphs / get_bond_graphs (utils/protein_module.py:251-364).  Atoms are flat
 * over structures, struct_ptr int32 [n_struct + 1] = atom offsets; radius [n_atoms] = covalent cut-off radius of
 * each atom's element (COVCUTOFFTABLE), heavy int32 [n_atoms] = 1 for Z != 1.  counts int32 [n_struct][6] =
 * {bonds in xyz, bonds in xyz_recon, pairs on which the graphs differ} over all atoms, then over heavy atoms
 * (unordered pairs; the reference's full matrices count each twice, which cancels in its ratios). */
int codlad_bond_graph_counts(const float *xyz, const float *xyz_recon, const float *radius, const int32_t *heavy,
                             const int32_t *struct_ptr, int n_struct, int max_atoms, float scale, int32_t *counts,
                             void *stream);

/* Self-test of the MFMA chain primitive: Y[n][:] = act(W @ X[n][:] + bias), n < 32*tiles.
 * act: 0 = none, 1 = exact-erf GELU. */
int codlad_selftest_gemm128(const float *W_packed, const float *bias, const float *X, int n_rows,
                            int act, float *Y, void *stream);

/* Same for the split-fp16 contraction: Y[n][:] = W @ act_in(X[n][:]) + bias with W packed in the
 * split order (codlad_amd.weights.pack_block_h); act_in: 0 = none, 1 = GELU applied to the input;
 * terms: 4 (f16x4) or 3 (f16x3). */
int codlad_selftest_gemm128_h(const void *W_split, const float *bias, const float *X, int n_rows,
                              int act_in, int terms, float *Y, void *stream);

#ifdef __cplusplus
}
#endif
#endif
