// Layer-stack engine for the narrow widths (dim = 16 / 32 / 64: the reference's RNA configurations,
// inference_rna_puzzles.py:29-30, main_rna_puzzles.py:52-53): the n_layer x (global, local) loop of PAMNet.forward
// (models.py:196-204) as ONE call per direction, the counterpart of engine.hip for dim = 128.
//
// Per layer the node-side chains are single launches (narrow_chain.h), the edge / triplet-row work uses the row kernels
// of narrow_core.h, every weight gradient goes from the kernels' partial rows straight into the caller's gradient
// buffers (column blocks of the 3d-wide message weights included), and gradients of the tensors that feed every layer
// (e_g, rbf_e, e_sbf) are accumulated inside the kernels that produce them.  Against the per-operator path this removes
// the host work of ~50 autograd nodes per layer pair (the d = 16 step was bound by it) and ~70 % of the launches.
//
// Forward, global layer k (layers/global_message_passing.py:33-56):
//   npre_fwd      x -> x1 = SiLU(mlp_x1 x), P = [W_i x1 | W_j x1]
//   nglobal_fwd   msg = SiLU(P_i[tgt] + P_j[src] + W_e e + b) * (W_ea e)          (E_g rows)
//   segment_sum   x2 = x1 + sum_{e -> i} msg
//   ntail_fwd     x2, x -> x_out, out, att
// local layer k (layers/local_message_passing.py:36-66):
//   npre_fwd      x -> x1, P = [ji_i | kj_i | ji_j | kj_j]
//   npre_fwd      rbf_e -> Q = [ji_e | kj_e | lin_rbf | lin_rbf_out]               (E_l rows, no first layer)
//   nlocal_gate   m_ji = SiLU(..), m_nb = SiLU(..) * Q_2
//   nmlp2_fwd     s = mlp_sbf(e_sbf)                                               (T+P rows)
//   segment_sum   m_other = sum_{rows of edge} m_nb[col] * s
//   nlocal_msg    m = Q_3 * (m_ji + m_other);  segment_sum x2 = x1 + sum m;  ntail_fwd
// The backward runs the same list in reverse with the backward kernels.
#include "narrow_chain.h"

namespace {

#define TRY(x)                              \
    do {                                    \
        const int rc_ = (x);                \
        if (rc_ != PAMNET_OK) return rc_;   \
    } while (0)

// m = Q[:, 3d:4d] * (m_ji + m_other)   (local_message_passing.py:53), one float4 per thread
__global__ __launch_bounds__(256) void nlocal_msg_fwd_kernel(const float4* __restrict__ Q, const float4* __restrict__ m_ji,
                                                             const float4* __restrict__ m_other, int64_t el, int d4,
                                                             float4* __restrict__ msg) {
    const int64_t total = el * d4;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t q = t / d4;
        const int c = (int)(t - q * d4);
        const float4 g = Q[q * (4 * d4) + 3 * d4 + c], a = m_ji[t], b = m_other[t];
        msg[t] = make_float4(g.x * (a.x + b.x), g.y * (a.y + b.y), g.z * (a.z + b.z), g.w * (a.w + b.w));
    }
}

// gm = d_x2[tgt[q]];  dQ[:, 3d:4d] = gm * (m_ji + m_other);  dmm = gm * Q[:, 3d:4d]  (= d m_ji = d m_other)
__global__ __launch_bounds__(256) void nlocal_msg_bwd_kernel(const float4* __restrict__ d_x2, const int32_t* __restrict__ tgt,
                                                             const float4* __restrict__ Q, const float4* __restrict__ m_ji,
                                                             const float4* __restrict__ m_other, int64_t el, int d4,
                                                             float4* __restrict__ dQ, float4* __restrict__ dmm) {
    const int64_t total = el * d4;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t q = t / d4;
        const int c = (int)(t - q * d4);
        const float4 gm = d_x2[(int64_t)tgt[q] * d4 + c];
        const float4 g = Q[q * (4 * d4) + 3 * d4 + c], a = m_ji[t], b = m_other[t];
        dQ[q * (4 * d4) + 3 * d4 + c] = make_float4(gm.x * (a.x + b.x), gm.y * (a.y + b.y), gm.z * (a.z + b.z), gm.w * (a.w + b.w));
        dmm[t] = make_float4(gm.x * g.x, gm.y * g.y, gm.z * g.z, gm.w * g.w);
    }
}

inline int ew_grid(int64_t total) {
    const int64_t want = (total + 255) / 256;
    return (int)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
}

// ---- launches --------------------------------------------------------------------------------------------------------
int launch_tail_fwd(int d, const NTailFwd& p, hipStream_t st) {
    if (p.m == 0) return PAMNET_OK;
    const int grid = chain_grid(p.m);
#define CALL(DD)                                                                                          \
    {                                                                                                     \
        const hipError_t e_ = allow_lds(ntail_fwd_kernel<DD>, ntail_fwd_lds<DD>());                       \
        if (e_ != hipSuccess) return (int)e_;                                                             \
        hipLaunchKernelGGL((ntail_fwd_kernel<DD>), dim3(grid), dim3(64 * CHW), ntail_fwd_lds<DD>(), st, p); \
    }
    NARROW_DISPATCH(d, CALL)
#undef CALL
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

inline int tail_stride(int d) { return d == 16 ? TailRow<16>::STRIDE : (d == 32 ? TailRow<32>::STRIDE : TailRow<64>::STRIDE); }

// gW / gb: 10 + 10 gradient buffers in chain order, head gradients gw_out [d], gb_out [1], gw_att [d]
int launch_tail_bwd(int d, NTailBwd p, float* const* gW, float* const* gb, float* gw_out, float* gb_out, float* gw_att,
                    Reducer& R, hipStream_t st) {
    if (p.m == 0) return PAMNET_EINVAL;
    const int grid = chain_grid(p.m);
    p.stride = tail_stride(d);
    {
        const int rc = R.take(grid, p.stride, 23, &p.partial);
        if (rc) return rc;
    }
#define CALL(DD)                                                                                          \
    {                                                                                                     \
        const hipError_t e_ = allow_lds(ntail_bwd_kernel<DD>, ntail_bwd_lds<DD>());                       \
        if (e_ != hipSuccess) return (int)e_;                                                             \
        hipLaunchKernelGGL((ntail_bwd_kernel<DD>), dim3(grid), dim3(64 * CHW), ntail_bwd_lds<DD>(), st, p); \
    }
    NARROW_DISPATCH(d, CALL)
#undef CALL
    PAMNET_LAUNCH_CHECK();
    const int MAT = d * d, LIN = MAT + d, MLP = 2 * MAT + 2 * d;
    SegTable& T = R.T;
    T.mat(gW[0], 0, d, d, d, d);
    T.vec(gb[0], MAT, d);
    for (int k = 0; k < 4; ++k) {
        const int off = LIN + k * MLP;
        T.mat(gW[2 + 2 * k], off, d, d, d, d);                 // segment layout: [dW2][db2][dW1][db1]
        T.vec(gb[2 + 2 * k], off + MAT, d);
        T.mat(gW[1 + 2 * k], off + LIN, d, d, d, d);
        T.vec(gb[1 + 2 * k], off + LIN + MAT, d);
    }
    const int l5 = LIN + 4 * MLP;
    T.mat(gW[9], l5, d, d, d, d);
    T.vec(gb[9], l5 + MAT, d);
    const int h = l5 + LIN;
    T.vec(gw_out, h, d);
    T.vec(gw_att, h + d, d);
    T.vec(gb_out, h + 2 * d, 1);
    return PAMNET_OK;
}

int launch_pre_fwd(int d, const NPreFwd& p, hipStream_t st) {
    if (p.m == 0) return PAMNET_OK;
    const int grid = grid_for(p.m, fwd_per_cu(d));
#define CALL(DD)                                                                                      \
    {                                                                                                 \
        const hipError_t e_ = allow_lds(npre_fwd_kernel<DD>, npre_fwd_lds<DD>());                     \
        if (e_ != hipSuccess) return (int)e_;                                                         \
        hipLaunchKernelGGL((npre_fwd_kernel<DD>), dim3(grid), dim3(NWG), npre_fwd_lds<DD>(), st, p);  \
    }
    NARROW_DISPATCH(d, CALL)
#undef CALL
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

// gWp[k]: gradient destination of projection block k (row stride ldg[k]); gW1 [d, d], gb1 [d]
int launch_pre_bwd(int d, NPreBwd p, float* const* gWp, const int* ldg, float* gW1, float* gb1, Reducer& R, hipStream_t st) {
    if (p.m == 0) return PAMNET_EINVAL;
    const int grid = chain_grid(p.m);
    const int MAT = d * d;
    p.stride = (p.nb + 1) * MAT + d;
    {
        const int rc = R.take(grid, p.stride, p.nb + 2, &p.partial);
        if (rc) return rc;
    }
#define CALL(DD)                                                                                         \
    {                                                                                                    \
        const hipError_t e_ = allow_lds(npre_bwd_kernel<DD>, npre_bwd_lds<DD>());                        \
        if (e_ != hipSuccess) return (int)e_;                                                            \
        hipLaunchKernelGGL((npre_bwd_kernel<DD>), dim3(grid), dim3(64 * CHW), npre_bwd_lds<DD>(), st, p); \
    }
    NARROW_DISPATCH(d, CALL)
#undef CALL
    PAMNET_LAUNCH_CHECK();
    SegTable& T = R.T;
    for (int k = 0; k < p.nb; ++k) T.mat(gWp[k], k * MAT, d, d, d, ldg[k]);
    T.mat(gW1, p.nb * MAT, d, d, d, d);
    T.vec(gb1, (p.nb + 1) * MAT, d);
    return PAMNET_OK;
}

constexpr int EROW_BLOCKS = 256;            // backward row kernels: at most one workgroup per CU (narrow_core.h)

int launch_global_fwd(int d, const float* e, int64_t m, const int32_t* tgt, const int32_t* src, const float* P,
                      const float* We, int ldwe, const float* bias, const float* Wea, int ldwea, float* msg,
                      hipStream_t st) {
    if (m == 0) return PAMNET_OK;
    const int grid = grid_for(m, fwd_per_cu(d));
#define CALL(DD)                                                                                                       \
    hipLaunchKernelGGL((nglobal_fwd_kernel<DD>), dim3(grid), dim3(NWG), 2 * wimg_bytes(DD), st, e, m, tgt, \
                       src, P, We, ldwe, bias, Wea, ldwea, msg);
    NARROW_DISPATCH(d, CALL)
#undef CALL
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

int launch_global_bwd(int d, const float* e, int64_t m, const int32_t* tgt, const int32_t* src, const float* P,
                      const float* We, int ldwe, const float* bias, const float* Wea, int ldwea, const float* dagg,
                      float* dz, float* de, int acc_de, Reducer& R, float* gWe, int ldgwe, float* gWea, float* gb,
                      hipStream_t st) {
    const int grid = grid_for(m, 1, bwd_waves(d));
    const int stride = 2 * d * d + d;
    float* partial = nullptr;
    {
        const int rc = R.take(grid, stride, 3, &partial);
        if (rc) return rc;
    }
#define CALL(DD)                                                                                                     \
    {                                                                                                                \
        const size_t lds = 4 * wimg_bytes(DD) + bwd_waves(DD) * 16 * (DD + 4) * sizeof(float);      \
        const hipError_t e_ = allow_lds(nglobal_bwd_kernel<DD>, lds);                                                \
        if (e_ != hipSuccess) return (int)e_;                                                                        \
        hipLaunchKernelGGL((nglobal_bwd_kernel<DD>), dim3(grid), dim3(64 * bwd_waves(DD)), lds, st, e, m, tgt, src, P, We, \
                           ldwe, bias, Wea, ldwea, dagg, dz, de, partial, stride, acc_de);                           \
    }
    NARROW_DISPATCH(d, CALL)
#undef CALL
    PAMNET_LAUNCH_CHECK();
    SegTable& T = R.T;
    T.mat(gWe, 0, d, d, d, ldgwe);
    T.mat(gWea, d * d, d, d, d, d);
    T.vec(gb, 2 * d * d, d);
    return PAMNET_OK;
}

int launch_mlp2_fwd(int d, const float* x, int64_t m, const float* W1, const float* b1, const float* W2, const float* b2,
                    float* y, hipStream_t st) {
    if (m == 0) return PAMNET_OK;
    const int grid = grid_for(m, fwd_per_cu(d));
#define CALL(DD)                                                                                                  \
    {                                                                                                             \
        const size_t lds = 2 * wimg_bytes(DD) + 4 * 16 * (DD + 4) * sizeof(float);               \
        hipLaunchKernelGGL((nmlp2_fwd_kernel<DD>), dim3(grid), dim3(NWG), lds, st, x, m, W1, b1, W2, b2, 0,       \
                           (const float*)nullptr, y);                                                             \
    }
    NARROW_DISPATCH(d, CALL)
#undef CALL
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

int launch_mlp2_bwd(int d, const float* x, int64_t m, const float* W1, const float* b1, const float* W2, const float* b2,
                    const float* dy, float* dx, int acc_dx, Reducer& R, float* gW1, float* gb1, float* gW2, float* gb2,
                    hipStream_t st) {
    const int grid = grid_for(m, 1, bwd_waves(d));
    const int stride = 2 * d * d + 2 * d;
    float* partial = nullptr;
    {
        const int rc = R.take(grid, stride, 4, &partial);
        if (rc) return rc;
    }
#define CALL(DD)                                                                                                       \
    {                                                                                                                  \
        const size_t lds = 4 * wimg_bytes(DD) + bwd_waves(DD) * 16 * (DD + 4) * sizeof(float);        \
        const hipError_t e_ = allow_lds(nmlp2_bwd_kernel<DD>, lds);                                                    \
        if (e_ != hipSuccess) return (int)e_;                                                                          \
        hipLaunchKernelGGL((nmlp2_bwd_kernel<DD>), dim3(grid), dim3(64 * bwd_waves(DD)), lds, st, x, m, W1, b1, W2, b2, dy, 0, \
                           dx, partial, stride, acc_dx);                                                               \
    }
    NARROW_DISPATCH(d, CALL)
#undef CALL
    PAMNET_LAUNCH_CHECK();
    SegTable& T = R.T;
    T.mat(gW1, 0, d, d, d, d);
    T.mat(gW2, d * d, d, d, d, d);
    T.vec(gb1, 2 * d * d, d);
    T.vec(gb2, 2 * d * d + d, d);
    return PAMNET_OK;
}

// one bias-free projection block of the edge-side Q = rbf [W_0 | W_1 | W_2 | W_3]^T: dW_k and (blocks whose gradient is the
// pre-activation gradient of a biased layer) the bias gradient = column sums of dQ_k; d rbf accumulated
int launch_qblock_bwd(int d, const float* x, int64_t m, const float* W, int ldw, const float* dy, int64_t lddy, float* dx,
                      int accumulate, float* partial, int stride, hipStream_t st) {
    const int grid = grid_for(m, 1, lin_bwd_waves(d));
#define CALL(DD)                                                                                                        \
    {                                                                                                                   \
        const size_t lds = 2 * wimg_bytes(DD) + lin_bwd_waves(DD) * 16 * (DD + 4) * sizeof(float);     \
        hipLaunchKernelGGL((nlinear_bwd_kernel<DD>), dim3(grid), dim3(64 * lin_bwd_waves(DD)), lds, st, x, m, W, ldw,   \
                           (const float*)nullptr, 0, dy, lddy, dx, accumulate, partial, stride);                        \
    }
    NARROW_DISPATCH(d, CALL)
#undef CALL
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

int launch_gate_bwd(int d, const float* P, const float* Q, const int32_t* tgt, const int32_t* src, const float* bji,
                    const float* bkj, int64_t m, const float* g_ji, const float* g_nb, float* dz, float* dQ, hipStream_t st) {
    const int grid = ew_grid(m * (d / 4));
#define CALL(DD) \
    hipLaunchKernelGGL((nlocal_gate_bwd_kernel<DD>), dim3(grid), dim3(256), 0, st, P, Q, tgt, src, bji, bkj, m, g_ji, g_nb, dz, dQ, 0);
    NARROW_DISPATCH(d, CALL)
#undef CALL
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

// ---- arenas ------------------------------------------------------------------------------------------------------------
inline int64_t up64(int64_t v) { return (v + 63) & ~(int64_t)63; }

struct Lay {
    int64_t n, eg, el, tp, d;
    // saved, per layer pair
    int64_t g_x1, g_P, g_x2, g_H0, g_R1, g_R2, g_R3, g_T, g_O;
    int64_t l_x1, l_P, l_x2, l_H0, l_R1, l_R2, l_R3, l_T, l_O, l_Q, l_mji, l_mnb, l_mother, l_s;
    int64_t pair;
    // temp
    int64_t t_msg, t_dz, t_ds, t_dQ, t_dzl, t_dmm, t_dmnb, t_dpi, t_dpj, t_dx2, t_dresx, t_gx0, t_gx1, t_partial, t_pack, temp;
    int64_t partial_floats;
};

Lay make_layout(int64_t n, int64_t eg, int64_t el, int64_t tp, int64_t d, int64_t n_layer = 1) {
    Lay L;
    L.n = n, L.eg = eg, L.el = el, L.tp = tp, L.d = d;
    int64_t o = 0;
    auto take = [&](int64_t floats) {
        const int64_t at = o;
        o += up64(floats);
        return at;
    };
    const int64_t nd = n * d;
    L.g_x1 = take(nd), L.g_P = take(2 * nd), L.g_x2 = take(nd), L.g_H0 = take(nd), L.g_R1 = take(nd), L.g_R2 = take(nd);
    L.g_R3 = take(nd), L.g_T = take(nd), L.g_O = take(nd);
    L.l_x1 = take(nd), L.l_P = take(4 * nd), L.l_x2 = take(nd), L.l_H0 = take(nd), L.l_R1 = take(nd), L.l_R2 = take(nd);
    L.l_R3 = take(nd), L.l_T = take(nd), L.l_O = take(nd);
    L.l_Q = take(4 * el * d), L.l_mji = take(el * d), L.l_mnb = take(el * d), L.l_mother = take(el * d), L.l_s = take(tp * d);
    L.pair = o;
    o = 0;
    const int64_t big = eg > el ? eg : el;
    L.t_msg = take(big * d);                 // forward: messages of the layer being aggregated
    L.t_dz = take(eg * d), L.t_ds = take(tp * d), L.t_dQ = take(4 * el * d), L.t_dzl = take(2 * el * d);
    L.t_dmm = take(el * d), L.t_dmnb = take(el * d), L.t_dpi = take(2 * nd), L.t_dpj = take(2 * nd);
    L.t_dx2 = take(nd), L.t_dresx = take(nd), L.t_gx0 = take(nd), L.t_gx1 = take(nd);
    const int64_t chain = (int64_t)chain_grid(n) * tail_stride((int)d);
    const int64_t pre = (int64_t)chain_grid(n) * (5 * d * d + d);
    const int64_t erow = (int64_t)EROW_BLOCKS * (2 * d * d + 2 * d);
    const int64_t qb = 4 * (int64_t)EROW_BLOCKS * (d * d + d);
    // every backward kernel of a layer pair keeps its partial rows until the pair's reductions run (Reducer): two chain
    // tails, two heads, the global row kernel, the triplet / pair MLP and the four projection blocks
    const int64_t pf = 2 * (chain + 64) + 2 * (pre + 64) + 2 * (erow + 64) + (qb + 64);
    L.partial_floats = pf;
    L.t_partial = take(pf);
    L.t_pack = take(n_layer * 2 * PACK_SLOTS * d * d);       // weight images of every layer pair (npack_kernel)
    L.temp = o;
    return L;
}

struct Idx {
    const int32_t *g_ptr, *g_row, *g_col, *gT_ptr, *gT_perm, *l_ptr, *l_row, *l_col, *lT_ptr, *lT_perm, *tp_ptr, *tp_row,
        *tp_col, *tpT_ptr, *tpT_perm;
};

inline Idx make_idx(const int32_t* const* g) {
    return Idx{g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7], g[8], g[9], g[10], g[11], g[12], g[13], g[14]};
}

constexpr int GP = 28, LP = 35;            // pointers per global / local layer in the parameter tables (pamnet_hip.h)

// slots of a layer pair's [d, d] weight blocks in the packed-image area
enum Slot { G_W1 = 0, G_P = 1, G_TAIL = 3, L_W1 = 13, L_P = 14, L_Q = 18, L_TAIL = 22 };

struct Images {
    const float* base;
    int d;
    const float4* at(int64_t pair, int slot, int transposed) const {
        return reinterpret_cast<const float4*>(base + ((pair * PACK_SLOTS + slot) * 2 + transposed) * (int64_t)d * d);
    }
};

int pack_all(int d, int64_t n_layer, const float* const* gparams, const float* const* lparams, float* out, hipStream_t st) {
    for (int64_t k = 0; k < n_layer; ++k) {
        const float* const* g = gparams + k * GP;
        const float* const* l = lparams + k * LP;
        PackJobs J;
        auto set = [&](int slot, const float* W, int ld) { J.W[slot] = W, J.ld[slot] = ld; };
        set(G_W1, g[0], d), set(G_P, g[2], 3 * d), set(G_P + 1, g[2] + d, 3 * d);
        for (int i = 0; i < 10; ++i) set(G_TAIL + i, g[5 + i], d), set(L_TAIL + i, l[12 + i], d);
        set(L_W1, l[0], d);
        set(L_P, l[2], 3 * d), set(L_P + 1, l[4], 3 * d), set(L_P + 2, l[2] + d, 3 * d), set(L_P + 3, l[4] + d, 3 * d);
        set(L_Q, l[2] + 2 * d, 3 * d), set(L_Q + 1, l[4] + 2 * d, 3 * d), set(L_Q + 2, l[10], d), set(L_Q + 3, l[11], d);
        float4* dst = reinterpret_cast<float4*>(out + k * 2 * PACK_SLOTS * (int64_t)d * d);
#define CALL(DD) hipLaunchKernelGGL((npack_kernel<DD>), dim3(PACK_SLOTS, 2), dim3(256), 0, st, J, dst);
        NARROW_DISPATCH(d, CALL)
#undef CALL
        PAMNET_LAUNCH_CHECK();
    }
    return PAMNET_OK;
}

inline void tail_fwd_params(NTailFwd& t, const float* const* tp, const Images& im, int64_t pair, int slot0) {
    for (int i = 0; i < 10; ++i) t.img[i] = im.at(pair, slot0 + i, 0), t.b[i] = tp[10 + i];
    t.w_out = tp[20], t.b_out = tp[21], t.w_att = tp[22];
}

}  // namespace

extern "C" int pamnet_narrow_stack_workspace(int64_t n, int64_t eg, int64_t el, int64_t tp, int64_t n_layer, int64_t d,
                                             int64_t* saved_floats, int64_t* temp_floats) {
    if (n < 0 || eg < 0 || el < 0 || tp < 0 || n_layer < 1 || !width_ok(d)) return PAMNET_EINVAL;
    if (!saved_floats || !temp_floats) return PAMNET_ENULL;
    const Lay L = make_layout(n, eg, el, tp, d, n_layer);
    *saved_floats = L.pair * n_layer;
    *temp_floats = L.temp;
    return PAMNET_OK;
}

// layout[0] = floats per layer pair in `saved`; [1], [2] = offsets of the global / local layer's node output in a pair
extern "C" int pamnet_narrow_stack_layout(int64_t n, int64_t eg, int64_t el, int64_t tp, int64_t d, int64_t* layout) {
    if (!width_ok(d)) return PAMNET_EINVAL;
    if (!layout) return PAMNET_ENULL;
    const Lay L = make_layout(n, eg, el, tp, d);
    layout[0] = L.pair, layout[1] = L.g_R3, layout[2] = L.l_R3;
    return PAMNET_OK;
}

extern "C" int pamnet_narrow_stack_fwd_f32(const int64_t* sizes, const int32_t* const* graph_idx, int64_t n_layer, int64_t d,
                                           const float* x0, const float* e_g, const float* rbf_e, const float* e_sbf,
                                           const float* const* gparams, const float* const* lparams, float* saved,
                                           float* temp, float* outs, float* atts, pamnet_stream_t stream) {
    if (!sizes || !graph_idx || !gparams || !lparams) return PAMNET_ENULL;
    const int64_t n = sizes[0], eg = sizes[1], el = sizes[2], tp = sizes[3];
    if (n < 0 || eg < 0 || el < 0 || tp < 0 || n_layer < 1 || !width_ok(d)) return PAMNET_EINVAL;
    if (n == 0) return PAMNET_OK;
    if (!x0 || !saved || !temp || !outs || !atts || (eg > 0 && !e_g) || (el > 0 && !rbf_e) || (tp > 0 && !e_sbf)) return PAMNET_ENULL;
    const Lay L = make_layout(n, eg, el, tp, d, n_layer);
    const Idx ix = make_idx(graph_idx);
    hipStream_t st = as_stream(stream);
    const int D = (int)d;
    TRY(pack_all(D, n_layer, gparams, lparams, temp + L.t_pack, st));
    const Images im{temp + L.t_pack, D};
    const float* x = x0;
    for (int64_t k = 0; k < n_layer; ++k) {
        float* S = saved + k * L.pair;
        const float* const* g = gparams + k * GP;
        const float* const* l = lparams + k * LP;
        // ---------------- global layer
        {
            NPreFwd p = {};
            p.x = x, p.img1 = im.at(k, G_W1, 0), p.b1 = g[1], p.nb = 2, p.imgp[0] = im.at(k, G_P, 0), p.imgp[1] = im.at(k, G_P + 1, 0);
            p.x1 = S + L.g_x1, p.P = S + L.g_P, p.m = n;
            TRY(launch_pre_fwd(D, p, st));
            TRY(launch_global_fwd(D, e_g, eg, ix.g_row, ix.g_col, S + L.g_P, g[2] + 2 * D, 3 * D, g[3], g[4], D, temp + L.t_msg, st));
            TRY(pamnet_segment_sum_f32(S + L.g_x2, S + L.g_x1, temp + L.t_msg, nullptr, nullptr, nullptr, nullptr, ix.g_ptr, n,
                                       d, stream));
            NTailFwd t = {};
            tail_fwd_params(t, g + 5, im, k, G_TAIL);
            t.x2 = S + L.g_x2, t.res_x = x, t.H0 = S + L.g_H0, t.R1 = S + L.g_R1, t.R2 = S + L.g_R2, t.R3 = S + L.g_R3;
            t.T = S + L.g_T, t.O = S + L.g_O, t.out = outs + (2 * k) * n, t.att = atts + (2 * k) * n, t.m = n;
            TRY(launch_tail_fwd(D, t, st));
            x = S + L.g_R3;
        }
        // ---------------- local layer
        {
            NPreFwd p = {};
            p.x = x, p.img1 = im.at(k, L_W1, 0), p.b1 = l[1], p.nb = 4, p.m = n, p.x1 = S + L.l_x1, p.P = S + L.l_P;
            for (int b = 0; b < 4; ++b) p.imgp[b] = im.at(k, L_P + b, 0);
            TRY(launch_pre_fwd(D, p, st));
            NPreFwd q = {};
            q.x = rbf_e, q.nb = 4, q.m = el, q.P = S + L.l_Q;
            for (int b = 0; b < 4; ++b) q.imgp[b] = im.at(k, L_Q + b, 0);
            TRY(launch_pre_fwd(D, q, st));
            TRY(pamnet_narrow_local_gate_fwd_f32(S + L.l_P, S + L.l_Q, ix.l_row, ix.l_col, l[3], l[5], el, d, S + L.l_mji,
                                                 S + L.l_mnb, stream));
            TRY(launch_mlp2_fwd(D, e_sbf, tp, l[6], l[7], l[8], l[9], S + L.l_s, st));
            TRY(pamnet_segment_sum_f32(S + L.l_mother, nullptr, S + L.l_mnb, ix.tp_col, S + L.l_s, nullptr, nullptr, ix.tp_ptr,
                                       el, d, stream));
            if (el > 0) {
                hipLaunchKernelGGL(nlocal_msg_fwd_kernel, dim3(ew_grid(el * (d / 4))), dim3(256), 0, st,
                                   (const float4*)(S + L.l_Q), (const float4*)(S + L.l_mji), (const float4*)(S + L.l_mother),
                                   el, (int)(d / 4), (float4*)(temp + L.t_msg));
                PAMNET_LAUNCH_CHECK();
            }
            TRY(pamnet_segment_sum_f32(S + L.l_x2, S + L.l_x1, temp + L.t_msg, nullptr, nullptr, nullptr, nullptr, ix.l_ptr, n,
                                       d, stream));
            NTailFwd t = {};
            tail_fwd_params(t, l + 12, im, k, L_TAIL);
            t.x2 = S + L.l_x2, t.res_x = x, t.H0 = S + L.l_H0, t.R1 = S + L.l_R1, t.R2 = S + L.l_R2, t.R3 = S + L.l_R3;
            t.T = S + L.l_T, t.O = S + L.l_O, t.out = outs + (2 * k + 1) * n, t.att = atts + (2 * k + 1) * n, t.m = n;
            TRY(launch_tail_fwd(D, t, st));
            x = S + L.l_R3;
        }
    }
    return PAMNET_OK;
}

extern "C" int pamnet_narrow_stack_bwd_f32(const int64_t* sizes, const int32_t* const* graph_idx, int64_t n_layer, int64_t d,
                                           const float* x0, const float* e_g, const float* rbf_e, const float* e_sbf,
                                           const float* const* gparams, const float* const* lparams, const float* saved,
                                           float* temp, const float* d_outs, const float* d_atts, float* const* ggrads,
                                           float* const* lgrads, float* d_x0, float* d_eg, float* d_rbf, float* d_sbf,
                                           void* const* layer_done, pamnet_stream_t stream) {
    if (!sizes || !graph_idx || !gparams || !lparams || !ggrads || !lgrads) return PAMNET_ENULL;
    const int64_t n = sizes[0], eg = sizes[1], el = sizes[2], tp = sizes[3];
    if (n <= 0 || eg <= 0 || el <= 0 || tp <= 0 || n_layer < 1 || !width_ok(d)) return PAMNET_EINVAL;   // (empty parts: the
    //                                                     caller zero-fills; a batch without edges has nothing to train on)
    if (!x0 || !e_g || !rbf_e || !e_sbf || !saved || !temp || !d_outs || !d_atts || !d_x0 || !d_eg || !d_rbf || !d_sbf)
        return PAMNET_ENULL;
    const Lay L = make_layout(n, eg, el, tp, d, n_layer);
    const Idx ix = make_idx(graph_idx);
    hipStream_t st = as_stream(stream);
    const int D = (int)d;
    TRY(pack_all(D, n_layer, gparams, lparams, temp + L.t_pack, st));
    const Images im{temp + L.t_pack, D};
    Reducer R(temp + L.t_partial, L.partial_floats, st);    // partial rows of the backward kernels: reduced a few kernels at a time
    float* d_x2 = temp + L.t_dx2;
    float* d_resx = temp + L.t_dresx;
    float* gx[2] = {temp + L.t_gx0, temp + L.t_gx1};
    const float* g_x = nullptr;              // gradient w.r.t. the node output of the layer being differentiated
    for (int64_t k = n_layer - 1; k >= 0; --k) {
        const float* S = saved + k * L.pair;
        const float* const* g = gparams + k * GP;
        const float* const* l = lparams + k * LP;
        float* const* gg = ggrads + k * GP;
        float* const* lg = lgrads + k * LP;
        const bool first = k == n_layer - 1;                 // first layer pair to be differentiated: overwrite d e_*
        const float* x_glob = k == 0 ? x0 : saved + (k - 1) * L.pair + L.l_R3;       // input of global layer k
        const float* x_loc = S + L.g_R3;                                              // input of local layer k
        // ---------------- local layer
        {
            NTailBwd t = {};
            for (int i = 0; i < 10; ++i) t.img[i] = im.at(k, L_TAIL + i, 0), t.imgt[i] = im.at(k, L_TAIL + i, 1), t.b[i] = l[22 + i];
            t.w_out = l[32], t.w_att = l[34];
            t.x2 = S + L.l_x2, t.H0 = S + L.l_H0, t.R1 = S + L.l_R1, t.R2 = S + L.l_R2, t.R3 = S + L.l_R3, t.T = S + L.l_T;
            t.O = S + L.l_O, t.g_x = g_x, t.g_out = d_outs + (2 * k + 1) * n, t.g_att = d_atts + (2 * k + 1) * n;
            t.d_x2 = d_x2, t.d_resx = d_resx, t.m = n;
            TRY(launch_tail_bwd(D, t, lg + 12, lg + 22, lg[32], lg[33], lg[34], R, st));
            // x2 = x1 + sum m,  m = Q_3 (m_ji + m_other)
            float* dQ = temp + L.t_dQ;
            float* dmm = temp + L.t_dmm;
            hipLaunchKernelGGL(nlocal_msg_bwd_kernel, dim3(ew_grid(el * (d / 4))), dim3(256), 0, st, (const float4*)d_x2,
                               ix.l_row, (const float4*)(S + L.l_Q), (const float4*)(S + L.l_mji),
                               (const float4*)(S + L.l_mother), el, (int)(d / 4), (float4*)dQ, (float4*)dmm);
            PAMNET_LAUNCH_CHECK();
            // m_other[r] = sum_{q in row r} m_nb[col q] * s[q]
            float* ds = temp + L.t_ds;
            float* dmnb = temp + L.t_dmnb;
            TRY(pamnet_gather_mul_f32(ds, S + L.l_mnb, ix.tp_col, dmm, ix.tp_row, tp, d, stream));
            TRY(pamnet_segment_sum_f32(dmnb, nullptr, S + L.l_s, nullptr, dmm, ix.tp_row, ix.tpT_perm, ix.tpT_ptr, el, d, stream));
            TRY(launch_mlp2_bwd(D, e_sbf, tp, l[6], l[7], l[8], l[9], ds, d_sbf, first ? 0 : 1, R, lg[6], lg[7], lg[8],
                                lg[9], st));
            // gates: dz_l [el, 2d] and dQ blocks 0..2 (block 3 is already there)
            float* dzl = temp + L.t_dzl;
            TRY(launch_gate_bwd(D, S + L.l_P, S + L.l_Q, ix.l_row, ix.l_col, l[3], l[5], el, dmm, dmnb, dzl, dQ, st));
            // node side: d P_i over the edges into i, d P_j over the edges out of j
            float* dpi = temp + L.t_dpi;
            float* dpj = temp + L.t_dpj;
            TRY(pamnet_segment_sum_f32(dpi, nullptr, dzl, nullptr, nullptr, nullptr, nullptr, ix.l_ptr, n, 2 * d, stream));
            TRY(pamnet_segment_sum_f32(dpj, nullptr, dzl, nullptr, nullptr, nullptr, ix.lT_perm, ix.lT_ptr, n, 2 * d, stream));
            // edge side: the four projection blocks of Q; their partial rows interleave so that one reduce serves all
            const float* Wq[4] = {l[2] + 2 * D, l[4] + 2 * D, l[10], l[11]};
            const int ldq[4] = {3 * D, 3 * D, D, D};
            const int qs = D * D + D;
            float* qpart = nullptr;
            TRY(R.take(grid_for(el, 1, lin_bwd_waves(D)), 4 * qs, 6, &qpart));
            if (D <= 32) {                                    // one pass over the rows for the four blocks (narrow_core.h)
                const int qgrid = grid_for(el, 1, lin_bwd_waves(D));
#define CALL(DD)                                                                                                        \
    {                                                                                                                   \
        const size_t lds = 4 * wimg_bytes(DD) + lin_bwd_waves(DD) * 16 * (DD + 4) * sizeof(float);                      \
        const size_t need = 4 * (size_t)qs * sizeof(float);                                                             \
        hipLaunchKernelGGL((nqblock4_bwd_kernel<DD>), dim3(qgrid), dim3(64 * lin_bwd_waves(DD)), lds > need ? lds : need, st, \
                           rbf_e, el, Wq[0], Wq[1], Wq[2], Wq[3], ldq[0], ldq[1], ldq[2], ldq[3], dQ, (int64_t)(4 * D), d_rbf, \
                           first ? 0 : 1, qpart, 4 * qs);                                                               \
    }
                if (D == 16) CALL(16) else CALL(32)
#undef CALL
                PAMNET_LAUNCH_CHECK();
            } else {
                for (int b = 0; b < 4; ++b)
                    TRY(launch_qblock_bwd(D, rbf_e, el, Wq[b], ldq[b], dQ + b * D, 4 * D, d_rbf, (first && b == 0) ? 0 : 1,
                                          qpart + b * qs, 4 * qs, st));
            }
            {
                SegTable& T = R.T;
                T.mat(lg[2] + 2 * D, 0 * qs, D, D, D, 3 * D);
                T.vec(lg[3], 0 * qs + D * D, D);                 // d b_ji = column sums of d z_ji
                T.mat(lg[4] + 2 * D, 1 * qs, D, D, D, 3 * D);
                T.vec(lg[5], 1 * qs + D * D, D);
                T.mat(lg[10], 2 * qs, D, D, D, D);
                T.mat(lg[11], 3 * qs, D, D, D, D);
            }
            NPreBwd p = {};
            p.x = x_loc, p.x1 = S + L.l_x1, p.img1 = im.at(k, L_W1, 0), p.img1t = im.at(k, L_W1, 1), p.b1 = l[1], p.nb = 4, p.m = n;
            for (int b = 0; b < 4; ++b) p.imgpt[b] = im.at(k, L_P + b, 1);
            p.dP[0] = dpi, p.dP[1] = dpi + D, p.dP[2] = dpj, p.dP[3] = dpj + D;
            p.lddp[0] = p.lddp[1] = p.lddp[2] = p.lddp[3] = 2 * D;
            p.d_direct = d_x2, p.d_add = d_resx, p.dx = gx[0];
            float* gWp[4] = {lg[2], lg[4], lg[2] + D, lg[4] + D};
            const int ldg[4] = {3 * D, 3 * D, 3 * D, 3 * D};
            TRY(launch_pre_bwd(D, p, gWp, ldg, lg[0], lg[1], R, st));
        }
        // ---------------- global layer
        {
            NTailBwd t = {};
            for (int i = 0; i < 10; ++i) t.img[i] = im.at(k, G_TAIL + i, 0), t.imgt[i] = im.at(k, G_TAIL + i, 1), t.b[i] = g[15 + i];
            t.w_out = g[25], t.w_att = g[27];
            t.x2 = S + L.g_x2, t.H0 = S + L.g_H0, t.R1 = S + L.g_R1, t.R2 = S + L.g_R2, t.R3 = S + L.g_R3, t.T = S + L.g_T;
            t.O = S + L.g_O, t.g_x = gx[0], t.g_out = d_outs + (2 * k) * n, t.g_att = d_atts + (2 * k) * n;
            t.d_x2 = d_x2, t.d_resx = d_resx, t.m = n;
            TRY(launch_tail_bwd(D, t, gg + 5, gg + 15, gg[25], gg[26], gg[27], R, st));
            float* dz = temp + L.t_dz;
            TRY(launch_global_bwd(D, e_g, eg, ix.g_row, ix.g_col, S + L.g_P, g[2] + 2 * D, 3 * D, g[3], g[4], D, d_x2, dz, d_eg,
                                  first ? 0 : 1, R, gg[2] + 2 * D, 3 * D, gg[4], gg[3], st));
            float* dpi = temp + L.t_dpi;
            float* dpj = temp + L.t_dpj;
            TRY(pamnet_segment_sum_f32(dpi, nullptr, dz, nullptr, nullptr, nullptr, nullptr, ix.g_ptr, n, d, stream));
            TRY(pamnet_segment_sum_f32(dpj, nullptr, dz, nullptr, nullptr, nullptr, ix.gT_perm, ix.gT_ptr, n, d, stream));
            NPreBwd p = {};
            p.x = x_glob, p.x1 = S + L.g_x1, p.img1 = im.at(k, G_W1, 0), p.img1t = im.at(k, G_W1, 1), p.b1 = g[1], p.nb = 2, p.m = n;
            p.imgpt[0] = im.at(k, G_P, 1), p.imgpt[1] = im.at(k, G_P + 1, 1);
            p.dP[0] = dpi, p.dP[1] = dpj, p.lddp[0] = p.lddp[1] = D;
            p.d_direct = d_x2, p.d_add = d_resx, p.dx = k == 0 ? d_x0 : gx[1];
            float* gWp[2] = {gg[2], gg[2] + D};
            const int ldg[2] = {3 * D, 3 * D};
            TRY(launch_pre_bwd(D, p, gWp, ldg, gg[0], gg[1], R, st));
            g_x = gx[1];
        }
        TRY(R.flush());                   // this pair's gradients are complete before its event is recorded
        if (layer_done) {
            const hipError_t e_ = hipEventRecord(static_cast<hipEvent_t>(layer_done[k]), st);
            if (e_ != hipSuccess) return (int)e_;
        }
    }
    return PAMNET_OK;
}
