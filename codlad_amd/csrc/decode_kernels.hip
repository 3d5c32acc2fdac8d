// Decoder tail (SURVEY.md 8a rows 8 and 10, and the CG graph of 8f-3): de-normalise + VQ lookup, CG neighbour
// list, internal coordinates -> Cartesian.  Row 9 (the IC decoder) is ic_decoder_kernels.hip.
#include "common.h"
#include "../../include/codlad_hip.h"

// ---------------------------------------------------------------------------------------------
// Row 8.  d(z, e) = (|z|^2 + |e|^2) - 2 * dot(z, e), evaluated in exactly the association the
// reference's CPU path uses so the argmin is bit-identical on identical latents:
//   |v|^2 = (v0*v0 + v1*v1) + v2*v2 (separately rounded),  dot = fma(z2,e2, fma(z1,e1, z0*e0)).
// First index wins ties.  A workgroup = 64 latents x 16 waves: the codebook (with |e|^2) sits in LDS, each
// wave scans one sixteenth of it (a broadcast read per code), and wave 0 merges the sixteen (distance, index)
// candidates in index order with a strict <, which is the first-minimum rule of the sequential scan.
// (Round 3: 16 waves instead of 8.  Two workgroups fit a CU's LDS either way, so cfg 5's 553 workgroups need a second,
// nearly empty round on the 512 slots; with twice the waves per workgroup each round takes half as long.)
// ---------------------------------------------------------------------------------------------
#define VQ_WAVES 16
__global__ __launch_bounds__(64 * VQ_WAVES) void vq_kernel(const float *x, int n, const float *mean3,
                                                          const float *std3, const float *codebook,
                                                          int n_codes, int64_t *idx, float *z_q,
                                                          float *latent_out) {
    extern __shared__ __align__(16) float4 code[];  // (e0, e1, e2, |e|^2) x n_codes, then the merge area
    float *cand_d = reinterpret_cast<float *>(code + n_codes);
    int *cand_i = reinterpret_cast<int *>(cand_d + 64 * VQ_WAVES);
    for (int cidx = threadIdx.x; cidx < n_codes; cidx += blockDim.x) {
        const float e0 = codebook[3 * cidx], e1 = codebook[3 * cidx + 1], e2 = codebook[3 * cidx + 2];
        const float se = __fadd_rn(__fadd_rn(__fmul_rn(e0, e0), __fmul_rn(e1, e1)), __fmul_rn(e2, e2));
        code[cidx] = make_float4(e0, e1, e2, se);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane, ii = i < n ? i : n - 1;
    const float z0 = __fadd_rn(__fmul_rn(x[3 * ii], std3[0]), mean3[0]);
    const float z1 = __fadd_rn(__fmul_rn(x[3 * ii + 1], std3[1]), mean3[1]);
    const float z2 = __fadd_rn(__fmul_rn(x[3 * ii + 2], std3[2]), mean3[2]);
    const float sz = __fadd_rn(__fadd_rn(__fmul_rn(z0, z0), __fmul_rn(z1, z1)), __fmul_rn(z2, z2));
    const int per = (n_codes + VQ_WAVES - 1) / VQ_WAVES;
    const int c0 = wave * per < n_codes ? wave * per : n_codes, c1 = c0 + per < n_codes ? c0 + per : n_codes;
    float best = INFINITY;
    int bi = 0;
    for (int cidx = c0; cidx < c1; ++cidx) {
        const float4 e = code[cidx];
        const float dot = __fmaf_rn(z2, e.z, __fmaf_rn(z1, e.y, __fmul_rn(z0, e.x)));
        const float d = __fsub_rn(__fadd_rn(sz, e.w), __fmul_rn(2.0f, dot));
        if (d < best) { best = d; bi = cidx; }
    }
    cand_d[wave * 64 + lane] = best;
    cand_i[wave * 64 + lane] = bi;
    __syncthreads();
    if (wave != 0 || i >= n) return;
    for (int w = 1; w < VQ_WAVES; ++w) {
        const float d = cand_d[w * 64 + lane];
        if (d < best) { best = d; bi = cand_i[w * 64 + lane]; }
    }
    idx[i] = bi;
    const float4 e = code[bi];
    z_q[3 * i] = e.x; z_q[3 * i + 1] = e.y; z_q[3 * i + 2] = e.z;
    if (latent_out) { latent_out[3 * i] = z0; latent_out[3 * i + 1] = z1; latent_out[3 * i + 2] = z2; }
}

extern "C" int codlad_vq_lookup(const float *x, int n, const float *mean3, const float *std3,
                                const float *codebook, int n_codes, int64_t *idx, float *z_q,
                                float *latent_out, void *stream) {
    CODLAD_REQUIRE(x && mean3 && std3 && codebook && idx && z_q, "null pointer");
    CODLAD_REQUIRE(n > 0 && n_codes > 0, "n and n_codes must be positive");
    const size_t lds = (size_t)n_codes * sizeof(float4) + 64 * VQ_WAVES * (sizeof(float) + sizeof(int));
    CODLAD_REQUIRE(lds <= 160 * 1024, "codebook does not fit in LDS");
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(vq_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { codlad_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    }
    hipLaunchKernelGGL(vq_kernel, dim3((n + 63) / 64), dim3(64 * VQ_WAVES), lds, (hipStream_t)stream, x, n,
                       mean3, std3, codebook, n_codes, idx, z_q, latent_out);
    return codlad_check_launch("codlad_vq_lookup");
}

// ---------------------------------------------------------------------------------------------
// SURVEY 8f-3 (host preprocessing moved to the device): the CG neighbour list of the IC decoder.
// Reference: get_neighbor_list(xyz, cutoff=cg_cutoff) keeps the pairs j > i with
// sqrt(sum((x_i - x_j)^2)) <= cutoff (utils/protein_module.py:567-584); make_directed appends the
// flipped pairs (models/gcn_nn.py:54-64) and scatter_add sums per receiving node in list order
// (models/vae_model.py:485-488).  Built directly as CSR over the receiver: for node i first the
// senders j > i in ascending order, then the senders j < i in ascending order.
// One wave per node; pass 1 counts, pass 2 (after an exclusive scan of the counts) fills.
// ---------------------------------------------------------------------------------------------
DEV bool cg_within(const float *xyz, int i, int j, float cutoff) {
    const float dx = xyz[3 * i] - xyz[3 * j], dy = xyz[3 * i + 1] - xyz[3 * j + 1], dz = xyz[3 * i + 2] - xyz[3 * j + 2];
    return sqrtf((dx * dx + dy * dy) + dz * dz) <= cutoff;     // -ffp-contract=off: rounded like the reference
}

template <bool FILL>
__global__ __launch_bounds__(256) void cg_graph_kernel(const float *xyz, const int2 *range, int M, float cutoff,
                                                      int32_t *degree, const int32_t *csr_ptr, int32_t *csr_src) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= M) return;
    const int first = range[i].x, last = first + range[i].y;   // nodes of i's sample: [first, last)
    int count = FILL ? csr_ptr[i] : 0;
    // senders j > i, then j < i, each ascending; 64 candidates per step, order kept by ballot ranks
    for (int phase = 0; phase < 2; ++phase) {
        const int lo = phase == 0 ? i + 1 : first, hi = phase == 0 ? last : i;
        for (int j0 = lo; j0 < hi; j0 += 64) {
            const int j = j0 + lane;
            const bool in = j < hi && cg_within(xyz, i, j, cutoff);
            const unsigned long long m = __ballot(in);
            if (FILL && in) csr_src[count + __popcll(m & ((1ull << lane) - 1ull))] = j;
            count += __popcll(m);
        }
    }
    if (!FILL && lane == 0) degree[i] = count;
}

extern "C" int codlad_cg_graph(const float *cg_xyz, const int32_t *sample_range, int M, float cutoff,
                               int32_t *degree, const int32_t *csr_ptr, int32_t *csr_src, void *stream) {
    CODLAD_REQUIRE(cg_xyz && sample_range && M > 0, "bad arguments");
    CODLAD_REQUIRE((degree != nullptr) != (csr_ptr != nullptr && csr_src != nullptr),
                   "pass either degree (count pass) or csr_ptr + csr_src (fill pass)");
    dim3 grid((M + 3) / 4), block(256);
    const int2 *rg = reinterpret_cast<const int2 *>(sample_range);
    if (degree) hipLaunchKernelGGL(cg_graph_kernel<false>, grid, block, 0, (hipStream_t)stream, cg_xyz, rg, M, cutoff, degree, nullptr, nullptr);
    else hipLaunchKernelGGL(cg_graph_kernel<true>, grid, block, 0, (hipStream_t)stream, cg_xyz, rg, M, cutoff, nullptr, csr_ptr, csr_src);
    return codlad_check_launch("codlad_cg_graph");
}

// ---------------------------------------------------------------------------------------------
// Row 10.  One thread per (frame, residue): 13 sequential z-matrix placements.
// ---------------------------------------------------------------------------------------------
struct V3 { float x, y, z; };
DEV V3 v3(float x, float y, float z) { V3 r = {x, y, z}; return r; }
DEV V3 sub3(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV float dot3(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }

// Euler-Rodrigues rotation of d about `axis` by `angle` (utils_ic.py:197-210)
DEV V3 rotate(V3 axis, float angle, V3 d) {
    const float nrm = sqrtf(dot3(axis, axis));
    const float a = cosf(angle / 2.0f), sn = sinf(angle / 2.0f);
    const float b = -(axis.x / nrm) * sn, c = -(axis.y / nrm) * sn, e = -(axis.z / nrm) * sn;
    const float r00 = a * a + b * b - c * c - e * e, r01 = 2 * (b * c - a * e), r02 = 2 * (b * e + a * c);
    const float r10 = 2 * (b * c + a * e), r11 = a * a + c * c - b * b - e * e, r12 = 2 * (c * e - a * b);
    const float r20 = 2 * (b * e - a * c), r21 = 2 * (c * e + a * b), r22 = a * a + e * e - b * b - c * c;
    return v3(r00 * d.x + r01 * d.y + r02 * d.z, r10 * d.x + r11 * d.y + r12 * d.z,
              r20 * d.x + r21 * d.y + r22 * d.z);
}

// add_atom_to_xyz (utils_ic.py:213-239)
DEV V3 place_atom(const float *ic3, V3 atom1, V3 atom2, V3 atom3) {
    V3 a = sub3(atom2, atom1), b = sub3(atom2, atom3);
    a.x = a.x == 0.f ? a.x + 1e-8f : a.x; a.y = a.y == 0.f ? a.y + 1e-8f : a.y; a.z = a.z == 0.f ? a.z + 1e-8f : a.z;
    b.x = b.x == 0.f ? b.x + 1e-8f : b.x; b.y = b.y == 0.f ? b.y + 1e-8f : b.y; b.z = b.z == 0.f ? b.z + 1e-8f : b.z;
    const float na = sqrtf(dot3(a, a)), dist = fabsf(ic3[0]);
    V3 d = v3(dist * a.x / na, dist * a.y / na, dist * a.z / na);
    const V3 normal = v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
    d = rotate(normal, ic3[1], d);
    d = rotate(a, ic3[2], d);
    return v3(atom1.x + d.x, atom1.y + d.y, atom1.z + d.z);
}

// one (frame, residue): the 14 atoms of the residue in LDS, one column per thread - the side-chain placements pick their
// three reference atoms by index (orders), and a register array indexed at run time would live in scratch memory
DEV void ic_to_xyz_row(const float *ca_full, const float *ic, const int32_t *orders, const int32_t *slot_to_out, int b,
                       int r, int L, int n_atoms, float *xyz_out, float (*ax)[128], float (*ay)[128], float (*az)[128],
                       int tid) {
    const float *ca = ca_full + ((size_t)b * (L + 2) + r) * 3;  // residue r-1 (flanking) .. r+1
    const V3 prv = v3(ca[0], ca[1], ca[2]), mid = v3(ca[3], ca[4], ca[5]), nxt = v3(ca[6], ca[7], ca[8]);
    const float *icr = ic + ((size_t)b * L + r) * 39;
    auto put = [&](int s, V3 a) { ax[s][tid] = a.x; ay[s][tid] = a.y; az[s][tid] = a.z; };
    auto get = [&](int s) { return v3(ax[s][tid], ay[s][tid], az[s][tid]); };
    const V3 an = place_atom(icr + 0, mid, prv, nxt);        // N
    const V3 ac = place_atom(icr + 3, mid, nxt, prv);        // C
    put(1, an);
    put(2, ac);
    put(0, place_atom(icr + 6, ac, mid, an));                // O
    put(3, mid);                                             // CA
    for (int i = 0; i < 10; ++i) {
        const int32_t *o = orders + ((size_t)i * L + r) * 3;
        put(4 + i, place_atom(icr + 9 + 3 * i, get(o[2]), get(o[1]), get(o[0])));
    }
    float *out = xyz_out + (size_t)b * n_atoms * 3;
    for (int s = 0; s < 14; ++s) {
        const int p = slot_to_out[r * 14 + s];
        if (p >= 0) { out[3 * p] = ax[s][tid]; out[3 * p + 1] = ay[s][tid]; out[3 * p + 2] = az[s][tid]; }
    }
}

__global__ __launch_bounds__(128) void ic_to_xyz_kernel(const float *ca_full, const float *ic,
                                                       const int32_t *orders,
                                                       const int32_t *slot_to_out, int B, int L,
                                                       int n_atoms, float *xyz_out) {
    __shared__ float ax[14][128], ay[14][128], az[14][128];
    const int tid = threadIdx.x;
    const int t = blockIdx.x * blockDim.x + tid;
    if (t >= B * L) return;
    const int b = t / L, r = t - b * L;
    ic_to_xyz_row(ca_full, ic, orders, slot_to_out, b, r, L, n_atoms, xyz_out, ax, ay, az, tid);
}

// several proteins (different L, atom tables) in ONE launch: thread t belongs to the group g with
// first_row[g] <= t < first_row[g + 1]
__global__ __launch_bounds__(128) void ic_to_xyz_groups_kernel(const codlad_xyz_group *groups, int n_groups, int total) {
    __shared__ float ax[14][128], ay[14][128], az[14][128];
    const int tid = threadIdx.x;
    const int t = blockIdx.x * blockDim.x + tid;
    if (t >= total) return;
    int lo = 0, hi = n_groups - 1;                       // last group whose first_row <= t
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (groups[mid].first_row <= t) lo = mid; else hi = mid - 1;
    }
    const codlad_xyz_group g = groups[lo];
    const int k = t - g.first_row, b = k / g.L, r = k - b * g.L;
    ic_to_xyz_row(g.ca_full, g.ic, g.orders, g.slot_to_out, b, r, g.L, g.n_atoms, g.xyz_out, ax, ay, az, tid);
}

extern "C" int codlad_ic_to_xyz(const float *ca_full, const float *ic, const int32_t *orders,
                                const int32_t *slot_to_out, int B, int L, int n_atoms,
                                float *xyz_out, void *stream) {
    CODLAD_REQUIRE(ca_full && ic && orders && slot_to_out && xyz_out, "null pointer");
    CODLAD_REQUIRE(B > 0 && L > 0 && n_atoms > 0, "B, L and n_atoms must be positive");
    hipLaunchKernelGGL(ic_to_xyz_kernel, dim3((B * L + 127) / 128), dim3(128), 0, (hipStream_t)stream,
                       ca_full, ic, orders, slot_to_out, B, L, n_atoms, xyz_out);
    return codlad_check_launch("codlad_ic_to_xyz");
}

extern "C" int codlad_ic_to_xyz_groups(const codlad_xyz_group *groups_dev, int n_groups, int total_rows, void *stream) {
    CODLAD_REQUIRE(groups_dev && n_groups > 0 && total_rows > 0, "bad arguments");
    hipLaunchKernelGGL(ic_to_xyz_groups_kernel, dim3((total_rows + 127) / 128), dim3(128), 0, (hipStream_t)stream,
                       groups_dev, n_groups, total_rows);
    return codlad_check_launch("codlad_ic_to_xyz_groups");
}

// ----------------------------------------------------------------------------------------------------------------------
// The inverse direction, for building data sets from coordinates: the internal coordinates the reference
// computes with mdtraj / numpy in build_ic_peptide_dataset (utils/protein_module.py:770-774 -> utils/utils_ic.py:141-196).
// A "quad" names four atoms of a frame: ic = (|A1 - A2|, angle(A1 - A2, A3 - A2), dihedral(A1, A2, A3, A4)), angle and
// dihedral reduced to [0, 2 pi) as protein_module.py:773 does; a quad with a negative index (a side-chain slot the
// residue does not have) gives zeros.  The host tables say which atoms: backbone N = (N, CA, CA-, CA+), C = (C, CA, CA+, CA-),
// O = (O, C, CA, N); side chain j = (core[j + 4], core[order[2]], core[order[1]], core[order[0]]).
DEV float wrap_2pi(float a) {
    const float two_pi = 6.283185307179586f;
    float m = fmodf(a, two_pi);
    if (m < 0.f) m += two_pi;
    return m;
}
__global__ __launch_bounds__(256) void xyz_to_ic_kernel(const float *xyz, int n_frames, int n_atoms, const int32_t *quads,
                                                       int n_quads, float *ic) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)n_frames * n_quads) return;
    const int f = (int)(t / n_quads), q = (int)(t - (size_t)f * n_quads);
    const int32_t *qa = quads + (size_t)q * 4;
    float *o = ic + t * 3;
    if (qa[0] < 0 || qa[1] < 0 || qa[2] < 0 || qa[3] < 0) { o[0] = o[1] = o[2] = 0.f; return; }
    const float *fr = xyz + (size_t)f * n_atoms * 3;
    auto at = [&](int i) { return v3(fr[3 * i], fr[3 * i + 1], fr[3 * i + 2]); };
    const V3 p0 = at(qa[0]), p1 = at(qa[1]), p2 = at(qa[2]), p3 = at(qa[3]);
    const V3 u = v3(p0.x - p1.x, p0.y - p1.y, p0.z - p1.z), w = v3(p2.x - p1.x, p2.y - p1.y, p2.z - p1.z);
    const float nu = sqrtf(dot3(u, u)), nw = sqrtf(dot3(w, w));
    // angle_between (utils_ic.py:95-106) is arccos of the clipped dot product of the unit vectors; the same angle from
    // atan2(|u x w|, u . w), which keeps fp32 accuracy where arccos loses it (angles near 0 and pi)
    const V3 uxw = v3(u.y * w.z - u.z * w.y, u.z * w.x - u.x * w.z, u.x * w.y - u.y * w.x);
    const float angle = atan2f(sqrtf(dot3(uxw, uxw)), dot3(u, w));
    // dihedral (utils_ic.py:109-138): b1 normalised, rejections of b0 and b2 from it, atan2
    const V3 b0 = u, b2 = v3(p3.x - p2.x, p3.y - p2.y, p3.z - p2.z);
    const V3 b1 = v3(w.x / nw, w.y / nw, w.z / nw);
    const float d0 = dot3(b0, b1), d2 = dot3(b2, b1);
    const V3 v = v3(b0.x - b1.x * d0, b0.y - b1.y * d0, b0.z - b1.z * d0);
    const V3 ww = v3(b2.x - b1.x * d2, b2.y - b1.y * d2, b2.z - b1.z * d2);
    const V3 cr = v3(b1.y * v.z - b1.z * v.y, b1.z * v.x - b1.x * v.z, b1.x * v.y - b1.y * v.x);
    o[0] = nu;
    o[1] = angle;                                            // in [0, pi]: nothing to wrap
    o[2] = wrap_2pi(atan2f(dot3(cr, ww), dot3(v, ww)));
}

extern "C" int codlad_xyz_to_ic(const float *xyz, int n_frames, int n_atoms, const int32_t *quads, int n_quads,
                                float *ic_out, void *stream) {
    CODLAD_REQUIRE(xyz && quads && ic_out, "null pointer");
    CODLAD_REQUIRE(n_frames > 0 && n_atoms > 0 && n_quads > 0, "n_frames, n_atoms and n_quads must be positive");
    const size_t total = (size_t)n_frames * n_quads;
    hipLaunchKernelGGL(xyz_to_ic_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       xyz, n_frames, n_atoms, quads, n_quads, ic_out);
    return codlad_check_launch("codlad_xyz_to_ic");
}
