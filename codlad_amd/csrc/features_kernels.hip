// Step-invariant CA features: k-NN graph, RBF / positional / orientation
// edge features, edge embedding, LayerNorm and W_e.  Depends on the CA trace only, so it runs
// once per structure and is shared by all steps and all ensemble members of that frame
// (the reference recomputes it in every denoiser call: latent_model.py:208).
//
// One 256-thread workgroup per structure node i:
//   A  distances to every node of the structure -> LDS
//   B  rank selection of the K = min(64, L) nearest (ascending, ties by lower index)
//   C  167 raw features per edge -> LDS
//   D  edge_embedding (167 -> 128)      E  LayerNorm(affine, 1e-5)      F  W_e (128 -> 128)
#include "common.h"
#include "../../include/codlad_hip.h"

#define HD 128
#define NFEAT 167
#define FSTR 169  // odd LDS row stride: edge-per-lane reads are conflict free
#define YSTR 129

DEV float dist_eps(float ax, float ay, float az, float bx, float by, float bz) {
    // sqrt(sum((a-b)^2) + 1e-6) with every operation rounded separately, in ATen's order
    const float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
    const float s = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
    return __fsqrt_rn(__fadd_rn(s, 1e-6f));
}

DEV void normalize3(float v[3]) {  // F.normalize: v / max(|v|, 1e-12)
    const float n = fmaxf(sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), 1e-12f);
    v[0] /= n; v[1] /= n; v[2] /= n;
}

DEV void cross3(const float a[3], const float b[3], float o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

// unit CA(p) -> CA(p+1) vector, zeroed when the step is not a 3.6-4.0 A virtual bond
DEV void bond_unit(const float *X, int p, float u[3]) {
    u[0] = X[3 * (p + 1) + 0] - X[3 * p + 0];
    u[1] = X[3 * (p + 1) + 1] - X[3 * p + 1];
    u[2] = X[3 * (p + 1) + 2] - X[3 * p + 2];
    const float n = sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    const float m = (3.6f < n && n < 4.0f) ? 1.0f : 0.0f;
    u[0] *= m; u[1] *= m; u[2] *= m;
    normalize3(u);
}

// local frame of residue p (rows o_1, n_2, o_1 x n_2); zero for p = 0 and the last two residues
DEV void frame_of(const float *X, int L, int p, float O[9]) {
    if (p < 1 || p > L - 3) {
#pragma unroll
        for (int e = 0; e < 9; ++e) O[e] = 0.f;
        return;
    }
    float u2[3], u1[3], n2[3], o1[3], t[3];
    bond_unit(X, p - 1, u2);
    bond_unit(X, p, u1);
    cross3(u2, u1, n2);
    normalize3(n2);
    o1[0] = u2[0] - u1[0]; o1[1] = u2[1] - u1[1]; o1[2] = u2[2] - u1[2];
    normalize3(o1);
    cross3(o1, n2, t);
#pragma unroll
    for (int e = 0; e < 3; ++e) { O[e] = o1[e]; O[3 + e] = n2[e]; O[6 + e] = t[e]; }
}

DEV float sgnf(float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); }

__global__ __launch_bounds__(256) void features_kernel(codlad_denoiser_weights w, const float *xyz,
                                                      const int2 *snode_info, int32_t *E_idx,
                                                      float *hE0, int lpad) {
    extern __shared__ __align__(16) float smem[];
    const int m = blockIdx.x, tid = threadIdx.x;
    const int start = snode_info[m].x, L = snode_info[m].y;
    const int i = m - start;
    const int K = L < 64 ? L : 64;
    const float *X = xyz + (size_t)start * 3;

    float *D = smem;                                   // [lpad]
    int *nb = reinterpret_cast<int *>(D + lpad);       // [64]
    float *dnb = reinterpret_cast<float *>(nb + 64);   // [64]
    float *feat = dnb + 64;                            // [64][FSTR]
    float *y = feat + 64 * FSTR;                       // [64][YSTR]

    const float xi = X[3 * i], yi = X[3 * i + 1], zi = X[3 * i + 2];
    for (int j = tid; j < L; j += 256)
        D[j] = dist_eps(X[3 * j], X[3 * j + 1], X[3 * j + 2], xi, yi, zi);
    __syncthreads();
    for (int j = tid; j < L; j += 256) {
        const float dj = D[j];
        int rank = 0;
        for (int q = 0; q < L; ++q) {
            const float dq = D[q];
            rank += (dq < dj || (dq == dj && q < j)) ? 1 : 0;
        }
        if (rank < K) { nb[rank] = j; dnb[rank] = dj; }
    }
    __syncthreads();

    const int k = tid & 63, part = tid >> 6;
    if (k < K) {
        const int j = nb[k];
        float *f = feat + k * FSTR;
        // neighbour triplets: A = {prev, self, next} of i, B = same of j; missing residues are
        // the zero vector (protein_mpnn_utils.py:485-489)
        float A[3][3], B[3][3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int pi = i + s - 1, pj = j + s - 1;
            const bool oki = pi >= 0 && pi < L, okj = pj >= 0 && pj < L;
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                A[s][e] = oki ? X[3 * pi + e] : 0.f;
                B[s][e] = okj ? X[3 * pj + e] : 0.f;
            }
        }
        // RBF block order (protein_mpnn_utils.py:494-505): (1,1) (0,0) (2,2) (0,1) (0,2) (1,0) (1,2) (2,0) (2,1)
        const int pa[9] = {1, 0, 2, 0, 0, 1, 1, 2, 2};
        const int pb[9] = {1, 0, 2, 1, 2, 0, 2, 0, 1};
        const int blk_lo = part == 0 ? 0 : (part == 1 ? 1 : (part == 2 ? 4 : 7));
        const int blk_hi = part == 0 ? 1 : (part == 1 ? 4 : (part == 2 ? 7 : 9));
        for (int bk = blk_lo; bk < blk_hi; ++bk) {
            const float d = bk == 0 ? dnb[k]
                                    : dist_eps(A[pa[bk]][0], A[pa[bk]][1], A[pa[bk]][2],
                                               B[pb[bk]][0], B[pb[bk]][1], B[pb[bk]][2]);
            for (int g = 0; g < 16; ++g) {
                const float u = (d - w.rbf_mu[g]) / 1.25f;
                f[16 + 16 * bk + g] = expf(-(u * u));
            }
        }
        if (part == 0) {
            int d = i - j + 32;
            d = d < 0 ? 0 : (d > 64 ? 64 : d);
            for (int g = 0; g < 16; ++g) f[g] = w.pos_w[g * 66 + d] + w.pos_b[g];
        }
        if (part == 3) {
            float Oi[9], Oj[9];
            frame_of(X, L, i, Oi);
            frame_of(X, L, j, Oj);
            const float dX[3] = {B[1][0] - A[1][0], B[1][1] - A[1][1], B[1][2] - A[1][2]};
            float dU[3];
#pragma unroll
            for (int r = 0; r < 3; ++r)
                dU[r] = Oi[3 * r] * dX[0] + Oi[3 * r + 1] * dX[1] + Oi[3 * r + 2] * dX[2];
            normalize3(dU);
            float R[3][3];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b)
                    R[a][b] = Oi[a] * Oj[b] + Oi[3 + a] * Oj[3 + b] + Oi[6 + a] * Oj[6 + b];
            const float Rxx = R[0][0], Ryy = R[1][1], Rzz = R[2][2];
            float q[4];
            q[0] = sgnf(R[2][1] - R[1][2]) * (0.5f * sqrtf(fabsf(1.f + (Rxx - Ryy - Rzz))));
            q[1] = sgnf(R[0][2] - R[2][0]) * (0.5f * sqrtf(fabsf(1.f + (-Rxx + Ryy - Rzz))));
            q[2] = sgnf(R[1][0] - R[0][1]) * (0.5f * sqrtf(fabsf(1.f + (-Rxx - Ryy + Rzz))));
            q[3] = sqrtf(fmaxf(1.f + (Rxx + Ryy + Rzz), 0.f)) / 2.f;
            const float qn = fmaxf(sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), 1e-12f);
            f[160] = dU[0]; f[161] = dU[1]; f[162] = dU[2];
            f[163] = q[0] / qn; f[164] = q[1] / qn; f[165] = q[2] / qn; f[166] = q[3] / qn;
        }
    }
    __syncthreads();

    // D: y[k][32*part .. +31] = feat[k] @ edge_wT
    float acc[32];
    {
#pragma unroll
        for (int e = 0; e < 32; ++e) acc[e] = 0.f;
        const float *fk = feat + k * FSTR;
        const float *wcol = w.edge_wT + 32 * part;
        if (k < K)
            for (int g = 0; g < NFEAT; ++g) {
                const float fv = fk[g];
                const float4 *wr = reinterpret_cast<const float4 *>(wcol + (size_t)g * HD);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float4 ww = wr[e];
                    acc[4 * e + 0] = fmaf(fv, ww.x, acc[4 * e + 0]);
                    acc[4 * e + 1] = fmaf(fv, ww.y, acc[4 * e + 1]);
                    acc[4 * e + 2] = fmaf(fv, ww.z, acc[4 * e + 2]);
                    acc[4 * e + 3] = fmaf(fv, ww.w, acc[4 * e + 3]);
                }
            }
#pragma unroll
        for (int e = 0; e < 32; ++e) y[k * YSTR + 32 * part + e] = acc[e];
    }
    __syncthreads();
    // E: LayerNorm over the 128 embedded features of each edge, affine, eps 1e-5
    {
        const float *yk = y + k * YSTR;
        float s = 0.f;
        for (int g = 0; g < HD; ++g) s += yk[g];
        const float mean = s * (1.0f / 128.0f);
        float v = 0.f;
        for (int g = 0; g < HD; ++g) { const float d = yk[g] - mean; v += d * d; }
        const float rstd = 1.0f / sqrtf(v * (1.0f / 128.0f) + 1e-5f);
#pragma unroll
        for (int e = 0; e < 32; ++e) {
            const int g = 32 * part + e;
            feat[k * FSTR + g] = (acc[e] - mean) * rstd * w.norm_w[g] + w.norm_b[g];
        }
    }
    __syncthreads();
    // F: h_E0 = W_e(E)
    if (k < K) {
        const float *fk = feat + k * FSTR;
        const float *wcol = w.We_wT + 32 * part;
#pragma unroll
        for (int e = 0; e < 32; ++e) acc[e] = w.We_b[32 * part + e];
        for (int g = 0; g < HD; ++g) {
            const float fv = fk[g];
            const float4 *wr = reinterpret_cast<const float4 *>(wcol + (size_t)g * HD);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float4 ww = wr[e];
                acc[4 * e + 0] = fmaf(fv, ww.x, acc[4 * e + 0]);
                acc[4 * e + 1] = fmaf(fv, ww.y, acc[4 * e + 1]);
                acc[4 * e + 2] = fmaf(fv, ww.z, acc[4 * e + 2]);
                acc[4 * e + 3] = fmaf(fv, ww.w, acc[4 * e + 3]);
            }
        }
        // edge block (common.h, EDGE_BLOCK): [2 halves][32 chunks][32 edges][4 floats]
        float4 *out = reinterpret_cast<float4 *>(hE0 + (size_t)m * (64 * HD)) + (k >> 5) * 1024 + (8 * part) * 32 + (k & 31);
        if (w.precision == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                out[e * 32] = make_float4(acc[4 * e], acc[4 * e + 1], acc[4 * e + 2], acc[4 * e + 3]);
        } else {
            // split-fp16 modes: the stored form the contractions read (common.h, "pre-split edge state"): for s, hh the
            // eight features 16 s + 4 hh + {0..3} and 16 s + 8 + 4 hh + {0..3} of this 32-feature block go as eight `hi`
            // halves to chunk 4 s + hh and as eight `lo` halves to chunk 4 s + 2 + hh
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    unsigned hi[4], lo[4];
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const int f = 16 * s + 8 * (p >> 1) + 4 * hh + 2 * (p & 1);
                        const f32x2 x = {acc[f], acc[f + 1]};
                        const f16x2 hv = __builtin_convertvector(x, f16x2);
                        hi[p] = __builtin_bit_cast(unsigned, hv);
                        lo[p] = __builtin_bit_cast(unsigned, split_lo_pair(hv, x));
                    }
                    reinterpret_cast<uint4 *>(out)[(4 * s + hh) * 32] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
                    reinterpret_cast<uint4 *>(out)[(4 * s + 2 + hh) * 32] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
                }
        }
        if (part == 0) E_idx[(size_t)m * 64 + k] = nb[k];
    }
}

extern "C" int codlad_features_prepass(const codlad_denoiser_weights *w, const float *cg_xyz,
                                       const int32_t *snode_info, int n_snodes, int max_len,
                                       int32_t *E_idx, float *h_E0, void *stream) {
    CODLAD_REQUIRE(w && cg_xyz && snode_info && E_idx && h_E0, "null pointer");
    CODLAD_REQUIRE(n_snodes > 0 && max_len > 0, "n_snodes and max_len must be positive");
    const int lpad = (max_len + 3) & ~3;
    const size_t lds = sizeof(float) * ((size_t)lpad + 128 + 64 * FSTR + 64 * YSTR);
    CODLAD_REQUIRE(lds <= 160 * 1024, "structure too long for the LDS distance row");
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(features_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { codlad_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    }
    hipLaunchKernelGGL(features_kernel, dim3(n_snodes), dim3(256), lds, (hipStream_t)stream, *w,
                       cg_xyz, reinterpret_cast<const int2 *>(snode_info), E_idx, h_E0, lpad);
    return codlad_check_launch("codlad_features_prepass");
}
