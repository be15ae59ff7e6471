// Fused edge-level kernels of PAMNet's message passing (dim = 128), forward and backward.
//
// With the message MLPs split algebraically (W[x_i|x_j|e] = W_i x_i + W_j x_j + W_e e) the per-edge work is
//   global (layers/global_message_passing.py:52-56):
//       z = W_e e + b_m + P_i[i] + P_j[j],   m = SiLU(z) * (W_ea e)
//   local  (layers/local_message_passing.py:46-48, 53):
//       z_ji = W_ji,e r + b_ji + P0[i] + P2[j]      m_ji = SiLU(z_ji)
//       z_kj = W_kj,e r + b_kj + P1[i] + P3[j]      m_nb = SiLU(z_kj) * (lin_rbf r)       q3 = lin_rbf_out r
//   triplet/pair MLP (layers/local_message_passing.py:49):  s = SiLU(W2 SiLU(W1 sbf + b1) + b2)
// Each kernel keeps a 64-row tile in LDS, runs its 2-4 fp32-MFMA GEMMs on it and applies the gather-add / SiLU / gate
// epilogue while the tile is written out in coalesced 512-byte rows -- the [E,3d] concatenations and the per-edge
// GEMM outputs of the reference never exist in memory.  P_* are the node-level projections (node_chain.hip), edges
// are sorted by target so P_i rows repeat within a tile (L1/L2 hits).
// Weight slices are prefetched into registers (gemm_core.h WFrag): the first two matrices at kernel entry, the next
// ones as soon as the MFMAs that consumed a slice have been issued.
#include "common.h"
#include "gemm_core.h"

using namespace pamnet;

namespace {

constexpr int BME = 64;                 // edge rows per workgroup
constexpr int MTE = BME / 16;
constexpr int SLOTE = BME * LDT;

__device__ __forceinline__ int wave_col0() { return (threadIdx.x >> 6) * 32; }
__device__ __forceinline__ Bias2 no_bias() { return load_bias2(nullptr, 0); }

// -------------------------------------------------------------------------------------------------- global edges
__global__ __launch_bounds__(WG) void global_edge_fwd_kernel(const float* __restrict__ e, int64_t m,
                                                             const float* __restrict__ We, int ld_we,
                                                             const float* __restrict__ bm,
                                                             const float* __restrict__ Wea, int ld_wea,
                                                             const float* __restrict__ Pi, const float* __restrict__ Pj,
                                                             const int32_t* __restrict__ row_of,
                                                             const int32_t* __restrict__ col, float* __restrict__ z,
                                                             float* __restrict__ ea, float* __restrict__ msg) {
    __shared__ __attribute__((aligned(16))) float lds[2 * SLOTE];
    float* S0 = lds;
    float* S1 = lds + SLOTE;
    const int64_t row0 = (int64_t)blockIdx.x * BME;
    const int wc = wave_col0();
    const Bias2 bv = load_bias2(bm, wc);
    WFrag f1, f2;
    load_wfrag<false>(f1, We, ld_we, wc);
    load_wfrag<false>(f2, Wea, ld_wea, wc);
    sweep_rows<BME>([&](int r, int c4) { st_lds4(S0, r, c4, ldg4z(e, row0 + r, m, DIM, c4)); });
    __syncthreads();
    f32x4 au[MTE][2], aa[MTE][2];
    acc_zero<MTE>(au);
    acc_zero<MTE>(aa);
    mma_tile_frag<MTE>(S0, f1, au);
    mma_tile_frag<MTE>(S0, f2, aa);
    __syncthreads();                                   // every wave is done reading the e tile
    acc_to_lds<MTE>(au, S0, wc, bv);
    acc_to_lds<MTE>(aa, S1, wc, no_bias());
    __syncthreads();
    sweep_rows<BME>([&](int r, int c4) {
        const int64_t g = row0 + r;
        if (g >= m) return;
        const int64_t i = row_of[g], j = col[g];
        const float4 zz = f4add(f4add(lds4(S0, r, c4), ldg4(Pi, i, DIM, c4)), ldg4(Pj, j, DIM, c4));
        const float4 gate = lds4(S1, r, c4);
        stg4(z, g, DIM, c4, zz);
        stg4(ea, g, DIM, c4, gate);
        stg4(msg, g, DIM, c4, f4mul(f4silu(zz), gate));
    });
}

// dm[e] = d_agg[i(e)];  dz = dm * ea * SiLU'(z);  dea = dm * SiLU(z);  d_e (+)= dz * W_e + dea * W_ea
__global__ __launch_bounds__(WG) void global_edge_bwd_kernel(const float* __restrict__ d_agg,
                                                             const int32_t* __restrict__ row_of, int64_t m,
                                                             const float* __restrict__ z, const float* __restrict__ ea,
                                                             const float* __restrict__ We, int ld_we,
                                                             const float* __restrict__ Wea, int ld_wea,
                                                             float* __restrict__ dz, float* __restrict__ dea,
                                                             float* __restrict__ d_e, int accumulate) {
    __shared__ __attribute__((aligned(16))) float lds[2 * SLOTE];
    float* S0 = lds;
    float* S1 = lds + SLOTE;
    const int64_t row0 = (int64_t)blockIdx.x * BME;
    const int wc = wave_col0();
    WFrag f1, f2;
    load_wfrag<true>(f1, We, ld_we, wc);
    load_wfrag<true>(f2, Wea, ld_wea, wc);
    sweep_rows<BME>([&](int r, int c4) {
        const int64_t g = row0 + r;
        float4 a = f4zero(), b = f4zero();
        if (g < m) {
            const float4 dm = ldg4(d_agg, row_of[g], DIM, c4);
            const float4 zz = ldg4(z, g, DIM, c4);
            a = f4mul(f4mul(dm, ldg4(ea, g, DIM, c4)), f4dsilu(zz));
            b = f4mul(dm, f4silu(zz));
            stg4(dz, g, DIM, c4, a);
            stg4(dea, g, DIM, c4, b);
        }
        st_lds4(S0, r, c4, a);
        st_lds4(S1, r, c4, b);
    });
    __syncthreads();
    f32x4 acc[MTE][2];
    acc_zero<MTE>(acc);
    mma_tile_frag<MTE>(S0, f1, acc);
    mma_tile_frag<MTE>(S1, f2, acc);
    __syncthreads();
    acc_to_lds<MTE>(acc, S0, wc, no_bias());
    __syncthreads();
    sweep_rows<BME>([&](int r, int c4) {
        const int64_t g = row0 + r;
        if (g >= m) return;
        float4 v = lds4(S0, r, c4);
        if (accumulate) v = f4add(v, ldg4(d_e, g, DIM, c4));
        stg4(d_e, g, DIM, c4, v);
    });
}

// -------------------------------------------------------------------------------------------------- local edges
struct LocalW {
    const float* W[4];      // ji_e, kj_e, lin_rbf, lin_rbf_out   ([128][*] blocks)
    int ld[4];
    const float* P[4];      // node planes: ji_i, kj_i, ji_j, kj_j  ([N][128] each)
};

__global__ __launch_bounds__(WG) void local_edge_fwd_kernel(const float* __restrict__ rbf, int64_t m, LocalW w,
                                                            const float* __restrict__ b_ji,
                                                            const float* __restrict__ b_kj,
                                                            const int32_t* __restrict__ row_of,
                                                            const int32_t* __restrict__ col, float* __restrict__ z_ji,
                                                            float* __restrict__ z_kj, float* __restrict__ q2,
                                                            float* __restrict__ q3, float* __restrict__ m_ji,
                                                            float* __restrict__ m_nb) {
    __shared__ __attribute__((aligned(16))) float lds[3 * SLOTE];
    float* S0 = lds;
    float* S1 = lds + SLOTE;
    float* S2 = lds + 2 * SLOTE;
    const int64_t row0 = (int64_t)blockIdx.x * BME;
    const int wc = wave_col0();
    const Bias2 bkj = load_bias2(b_kj, wc), bji = load_bias2(b_ji, wc);
    WFrag fa, fb;
    load_wfrag<false>(fa, w.W[2], w.ld[2], wc);
    load_wfrag<false>(fb, w.W[1], w.ld[1], wc);
    sweep_rows<BME>([&](int r, int c4) { st_lds4(S0, r, c4, ldg4z(rbf, row0 + r, m, DIM, c4)); });
    __syncthreads();
    // one GEMM on the rbf tile with the slice in `f`; then `f` is refilled with the slice of matrix `next` (or left)
    auto gemm = [&](WFrag& f, Bias2 bias, float* D, int next) {
        f32x4 acc[MTE][2];
        acc_zero<MTE>(acc);
        mma_tile_frag<MTE>(S0, f, acc);
        if (next >= 0) load_wfrag<false>(f, w.W[next], w.ld[next], wc);
        acc_to_lds<MTE>(acc, D, wc, bias);
        __syncthreads();
    };
    gemm(fa, no_bias(), S2, 0);                         // q2 = lin_rbf r (kept for the gate);   fa <- W_ji,e
    gemm(fb, bkj, S1, 3);                               // W_kj,e r + b_kj;                      fb <- lin_rbf_out
    sweep_rows<BME>([&](int r, int c4) {
        const int64_t g = row0 + r;
        if (g >= m) return;
        const int64_t i = row_of[g], j = col[g];
        const float4 zz = f4add(f4add(lds4(S1, r, c4), ldg4(w.P[1], i, DIM, c4)), ldg4(w.P[3], j, DIM, c4));
        const float4 gate = lds4(S2, r, c4);
        stg4(z_kj, g, DIM, c4, zz);
        stg4(q2, g, DIM, c4, gate);
        stg4(m_nb, g, DIM, c4, f4mul(f4silu(zz), gate));
    });
    __syncthreads();
    gemm(fa, bji, S1, -1);                              // W_ji,e r + b_ji
    gemm(fb, no_bias(), S2, -1);                        // q3 = lin_rbf_out r
    sweep_rows<BME>([&](int r, int c4) {
        const int64_t g = row0 + r;
        if (g >= m) return;
        const int64_t i = row_of[g], j = col[g];
        const float4 zz = f4add(f4add(lds4(S1, r, c4), ldg4(w.P[0], i, DIM, c4)), ldg4(w.P[2], j, DIM, c4));
        stg4(z_ji, g, DIM, c4, zz);
        stg4(m_ji, g, DIM, c4, f4silu(zz));
        stg4(q3, g, DIM, c4, lds4(S2, r, c4));
    });
}

// dz_ji = d_mji * SiLU'(z_ji);  dz_kj = d_mnb * q2 * SiLU'(z_kj);  dq2 = d_mnb * SiLU(z_kj);  dq3 given.
// d_rbf (+)= dz_ji W_ji,e + dz_kj W_kj,e + dq2 W_lin_rbf + dq3 W_lin_rbf_out
__global__ __launch_bounds__(WG) void local_edge_bwd_kernel(const float* __restrict__ d_mji,
                                                            const float* __restrict__ d_mnb,
                                                            const float* __restrict__ d_q3, int64_t m,
                                                            const float* __restrict__ z_ji,
                                                            const float* __restrict__ z_kj,
                                                            const float* __restrict__ q2, LocalW w,
                                                            float* __restrict__ dz_ji, float* __restrict__ dz_kj,
                                                            float* __restrict__ dq2, float* __restrict__ d_rbf,
                                                            int accumulate) {
    __shared__ __attribute__((aligned(16))) float lds[2 * SLOTE];
    float* S0 = lds;
    float* S1 = lds + SLOTE;
    const int64_t row0 = (int64_t)blockIdx.x * BME;
    const int wc = wave_col0();
    WFrag fa, fb;
    load_wfrag<true>(fa, w.W[0], w.ld[0], wc);
    load_wfrag<true>(fb, w.W[1], w.ld[1], wc);
    f32x4 acc[MTE][2];
    acc_zero<MTE>(acc);
    // pass A: dz_ji -> S0, dz_kj -> S1
    sweep_rows<BME>([&](int r, int c4) {
        const int64_t g = row0 + r;
        float4 a = f4zero(), b = f4zero();
        if (g < m) {
            a = f4mul(ldg4(d_mji, g, DIM, c4), f4dsilu(ldg4(z_ji, g, DIM, c4)));
            b = f4mul(f4mul(ldg4(d_mnb, g, DIM, c4), ldg4(q2, g, DIM, c4)), f4dsilu(ldg4(z_kj, g, DIM, c4)));
            stg4(dz_ji, g, DIM, c4, a);
            stg4(dz_kj, g, DIM, c4, b);
        }
        st_lds4(S0, r, c4, a);
        st_lds4(S1, r, c4, b);
    });
    __syncthreads();
    mma_tile_frag<MTE>(S0, fa, acc);
    load_wfrag<true>(fa, w.W[2], w.ld[2], wc);
    mma_tile_frag<MTE>(S1, fb, acc);
    load_wfrag<true>(fb, w.W[3], w.ld[3], wc);
    __syncthreads();
    // pass B: dq2 -> S0, dq3 -> S1
    sweep_rows<BME>([&](int r, int c4) {
        const int64_t g = row0 + r;
        float4 a = f4zero(), b = f4zero();
        if (g < m) {
            a = f4mul(ldg4(d_mnb, g, DIM, c4), f4silu(ldg4(z_kj, g, DIM, c4)));
            b = ldg4(d_q3, g, DIM, c4);
            stg4(dq2, g, DIM, c4, a);
        }
        st_lds4(S0, r, c4, a);
        st_lds4(S1, r, c4, b);
    });
    __syncthreads();
    mma_tile_frag<MTE>(S0, fa, acc);
    mma_tile_frag<MTE>(S1, fb, acc);
    __syncthreads();
    acc_to_lds<MTE>(acc, S0, wc, no_bias());
    __syncthreads();
    sweep_rows<BME>([&](int r, int c4) {
        const int64_t g = row0 + r;
        if (g >= m) return;
        float4 v = lds4(S0, r, c4);
        if (accumulate) v = f4add(v, ldg4(d_rbf, g, DIM, c4));
        stg4(d_rbf, g, DIM, c4, v);
    });
}

// -------------------------------------------------------------------------------------------------- 2-layer MLP
__global__ __launch_bounds__(WG) void mlp2_fwd_kernel(const float* __restrict__ x, int64_t m,
                                                      const float* __restrict__ W1, const float* __restrict__ b1,
                                                      const float* __restrict__ W2, const float* __restrict__ b2,
                                                      float* __restrict__ z1, float* __restrict__ z2,
                                                      float* __restrict__ y) {
    __shared__ __attribute__((aligned(16))) float lds[2 * SLOTE];
    float* S0 = lds;
    float* S1 = lds + SLOTE;
    const int64_t row0 = (int64_t)blockIdx.x * BME;
    const int wc = wave_col0();
    const Bias2 bv1 = load_bias2(b1, wc), bv2 = load_bias2(b2, wc);
    WFrag f;
    load_wfrag<false>(f, W1, DIM, wc);
    sweep_rows<BME>([&](int r, int c4) { st_lds4(S0, r, c4, ldg4z(x, row0 + r, m, DIM, c4)); });
    __syncthreads();
    {
        f32x4 acc[MTE][2];
        acc_zero<MTE>(acc);
        mma_tile_frag<MTE>(S0, f, acc);
        load_wfrag<false>(f, W2, DIM, wc);              // in flight during the activation sweep
        acc_to_lds<MTE>(acc, S1, wc, bv1);
        __syncthreads();
    }
    sweep_rows<BME>([&](int r, int c4) {
        const int64_t g = row0 + r;
        const float4 zz = lds4(S1, r, c4);
        st_lds4(S1, r, c4, f4silu(zz));
        if (g < m) stg4(z1, g, DIM, c4, zz);
    });
    __syncthreads();
    {
        f32x4 acc[MTE][2];
        acc_zero<MTE>(acc);
        mma_tile_frag<MTE>(S1, f, acc);
        acc_to_lds<MTE>(acc, S0, wc, bv2);
        __syncthreads();
    }
    sweep_rows<BME>([&](int r, int c4) {
        const int64_t g = row0 + r;
        if (g >= m) return;
        const float4 zz = lds4(S0, r, c4);
        stg4(z2, g, DIM, c4, zz);
        stg4(y, g, DIM, c4, f4silu(zz));
    });
}

__global__ __launch_bounds__(WG) void mlp2_bwd_kernel(const float* __restrict__ dy, int64_t m,
                                                      const float* __restrict__ z1, const float* __restrict__ z2,
                                                      const float* __restrict__ W1, const float* __restrict__ W2,
                                                      float* __restrict__ dz1, float* __restrict__ dz2,
                                                      float* __restrict__ dx, int accumulate) {
    __shared__ __attribute__((aligned(16))) float lds[2 * SLOTE];
    float* S0 = lds;
    float* S1 = lds + SLOTE;
    const int64_t row0 = (int64_t)blockIdx.x * BME;
    const int wc = wave_col0();
    WFrag f;
    load_wfrag<true>(f, W2, DIM, wc);
    sweep_rows<BME>([&](int r, int c4) {
        const int64_t g = row0 + r;
        float4 a = f4zero();
        if (g < m) {
            a = f4mul(ldg4(dy, g, DIM, c4), f4dsilu(ldg4(z2, g, DIM, c4)));
            stg4(dz2, g, DIM, c4, a);
        }
        st_lds4(S0, r, c4, a);
    });
    __syncthreads();
    {
        f32x4 acc[MTE][2];
        acc_zero<MTE>(acc);
        mma_tile_frag<MTE>(S0, f, acc);
        load_wfrag<true>(f, W1, DIM, wc);
        acc_to_lds<MTE>(acc, S1, wc, no_bias());
        __syncthreads();
    }
    sweep_rows<BME>([&](int r, int c4) {
        const int64_t g = row0 + r;
        float4 a = f4zero();
        if (g < m) {
            a = f4mul(lds4(S1, r, c4), f4dsilu(ldg4(z1, g, DIM, c4)));
            stg4(dz1, g, DIM, c4, a);
        }
        st_lds4(S1, r, c4, a);
    });
    __syncthreads();
    {
        f32x4 acc[MTE][2];
        acc_zero<MTE>(acc);
        mma_tile_frag<MTE>(S1, f, acc);
        __syncthreads();
        acc_to_lds<MTE>(acc, S0, wc, no_bias());
        __syncthreads();
    }
    sweep_rows<BME>([&](int r, int c4) {
        const int64_t g = row0 + r;
        if (g >= m) return;
        float4 v = lds4(S0, r, c4);
        if (accumulate) v = f4add(v, ldg4(dx, g, DIM, c4));
        stg4(dx, g, DIM, c4, v);
    });
}

inline unsigned tiles(int64_t m) { return (unsigned)ceil_div(m, BME); }

}  // namespace

extern "C" int pamnet_global_edge_fwd_f32(const float* e, int64_t n_edges, const float* We, int64_t ld_we,
                                          const float* bm, const float* Wea, int64_t ld_wea, const float* Pi,
                                          const float* Pj, const int32_t* row_of, const int32_t* col, float* z,
                                          float* ea, float* msg, pamnet_stream_t stream) {
    if (n_edges < 0) return PAMNET_EINVAL;
    if (n_edges == 0) return PAMNET_OK;
    if (!e || !We || !bm || !Wea || !Pi || !Pj || !row_of || !col || !z || !ea || !msg) return PAMNET_ENULL;
    hipLaunchKernelGGL(global_edge_fwd_kernel, dim3(tiles(n_edges)), dim3(WG), 0, as_stream(stream), e, n_edges, We,
                       (int)ld_we, bm, Wea, (int)ld_wea, Pi, Pj, row_of, col, z, ea, msg);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_global_edge_bwd_f32(const float* d_agg, const int32_t* row_of, int64_t n_edges, const float* z,
                                          const float* ea, const float* We, int64_t ld_we, const float* Wea,
                                          int64_t ld_wea, float* dz, float* dea, float* d_e, int32_t accumulate,
                                          pamnet_stream_t stream) {
    if (n_edges < 0) return PAMNET_EINVAL;
    if (n_edges == 0) return PAMNET_OK;
    if (!d_agg || !row_of || !z || !ea || !We || !Wea || !dz || !dea || !d_e) return PAMNET_ENULL;
    hipLaunchKernelGGL(global_edge_bwd_kernel, dim3(tiles(n_edges)), dim3(WG), 0, as_stream(stream), d_agg, row_of,
                       n_edges, z, ea, We, (int)ld_we, Wea, (int)ld_wea, dz, dea, d_e, (int)accumulate);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

static int fill_local(LocalW& w, const float* const* Wq, const int64_t* ldq, const float* const* P) {
    for (int b = 0; b < 4; ++b) {
        if (!Wq[b]) return PAMNET_ENULL;
        w.W[b] = Wq[b];
        w.ld[b] = (int)ldq[b];
        w.P[b] = P ? P[b] : nullptr;
        if (P && !P[b]) return PAMNET_ENULL;
    }
    return PAMNET_OK;
}

extern "C" int pamnet_local_edge_fwd_f32(const float* rbf, int64_t n_edges, const float* const* Wq, const int64_t* ldq,
                                         const float* b_ji, const float* b_kj, const float* const* P,
                                         const int32_t* row_of, const int32_t* col, float* z_ji, float* z_kj, float* q2,
                                         float* q3, float* m_ji, float* m_nb, pamnet_stream_t stream) {
    if (n_edges < 0) return PAMNET_EINVAL;
    if (n_edges == 0) return PAMNET_OK;
    if (!rbf || !Wq || !ldq || !b_ji || !b_kj || !P || !row_of || !col || !z_ji || !z_kj || !q2 || !q3 || !m_ji || !m_nb)
        return PAMNET_ENULL;
    LocalW w;
    int rc = fill_local(w, Wq, ldq, P);
    if (rc) return rc;
    hipLaunchKernelGGL(local_edge_fwd_kernel, dim3(tiles(n_edges)), dim3(WG), 0, as_stream(stream), rbf, n_edges, w,
                       b_ji, b_kj, row_of, col, z_ji, z_kj, q2, q3, m_ji, m_nb);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_local_edge_bwd_f32(const float* d_mji, const float* d_mnb, const float* d_q3, int64_t n_edges,
                                         const float* z_ji, const float* z_kj, const float* q2, const float* const* Wq,
                                         const int64_t* ldq, float* dz_ji, float* dz_kj, float* dq2, float* d_rbf,
                                         int32_t accumulate, pamnet_stream_t stream) {
    if (n_edges < 0) return PAMNET_EINVAL;
    if (n_edges == 0) return PAMNET_OK;
    if (!d_mji || !d_mnb || !d_q3 || !z_ji || !z_kj || !q2 || !Wq || !ldq || !dz_ji || !dz_kj || !dq2 || !d_rbf)
        return PAMNET_ENULL;
    LocalW w;
    int rc = fill_local(w, Wq, ldq, nullptr);
    if (rc) return rc;
    hipLaunchKernelGGL(local_edge_bwd_kernel, dim3(tiles(n_edges)), dim3(WG), 0, as_stream(stream), d_mji, d_mnb, d_q3,
                       n_edges, z_ji, z_kj, q2, w, dz_ji, dz_kj, dq2, d_rbf, (int)accumulate);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_mlp2_fwd_f32(const float* x, int64_t rows, const float* W1, const float* b1, const float* W2,
                                   const float* b2, float* z1, float* z2, float* y, pamnet_stream_t stream) {
    if (rows < 0) return PAMNET_EINVAL;
    if (rows == 0) return PAMNET_OK;
    if (!x || !W1 || !b1 || !W2 || !b2 || !z1 || !z2 || !y) return PAMNET_ENULL;
    hipLaunchKernelGGL(mlp2_fwd_kernel, dim3(tiles(rows)), dim3(WG), 0, as_stream(stream), x, rows, W1, b1, W2, b2, z1,
                       z2, y);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_mlp2_bwd_f32(const float* dy, int64_t rows, const float* z1, const float* z2, const float* W1,
                                   const float* W2, float* dz1, float* dz2, float* dx, int32_t accumulate,
                                   pamnet_stream_t stream) {
    if (rows < 0) return PAMNET_EINVAL;
    if (rows == 0) return PAMNET_OK;
    if (!dy || !z1 || !z2 || !W1 || !W2 || !dz1 || !dz2 || !dx) return PAMNET_ENULL;
    hipLaunchKernelGGL(mlp2_bwd_kernel, dim3(tiles(rows)), dim3(WG), 0, as_stream(stream), dy, rows, z1, z2, W1, W2,
                       dz1, dz2, dx, (int)accumulate);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}
