// Fused node-update chains of PAMNet's message-passing layers (dim = 128), forward and backward.
//
// Both layer kinds end with the same 10-Linear stack per node (layers/global_message_passing.py:39-50 /
// layers/local_message_passing.py:55-66):
//     h0 = SiLU(L0 x)                                 mlp_x2
//     r1 = SiLU(L2 SiLU(L1 h0)) + h0 + res_x          res1 (+ the layer's residual input)
//     r2 = SiLU(L4 SiLU(L3 r1)) + r1                  res2
//     r3 = SiLU(L6 SiLU(L5 r2)) + r2                  res3  -> x_out
//     o3 = SiLU(L9 SiLU(L8 SiLU(L7 r3)))              mlp_out
//     out = W_out . o3 + b_out,  att = W . o3
// and begin with   x1 = SiLU(Lx1 x),  P = x1 * Wp^T   (the node-level halves of the split message MLPs).
//
// One workgroup keeps a 16-row tile in LDS from the first GEMM to the last: ten fp32-MFMA GEMMs, bias / SiLU /
// residual epilogues and both heads in ONE launch (the reference issues ~40 kernels for the same work).  Per layer the
// pre-activation z_k is written once (coalesced 512 B rows) for the backward pass; r1, r2 are the only extra saves.
#include "common.h"
#include "gemm_core.h"

using namespace pamnet;

namespace {

struct TailParams {
    const float* W[10];
    const float* b[10];
    const float* w_out;   // [128]
    const float* b_out;   // [1]
    const float* w_att;   // [128]
};

constexpr int BMN = 16;                       // rows per workgroup in node-level chains
constexpr int SLOT = BMN * LDT;               // floats per LDS slot

// One GEMM of a chain with the weights already in registers (f); as soon as its MFMAs are issued the NEXT layer's
// weight slice is requested into the same registers, so the L2 latency overlaps the epilogue + activation sweep.
template <bool TRANS>
__device__ __forceinline__ void gemm16(const float* As, WFrag& f, const float* bias, float* Ds, const float* Wnext,
                                       int ld_next = DIM) {
    f32x4 acc[1][2];
    acc_zero<1>(acc);
    const int wcol0 = (threadIdx.x >> 6) * 32;
    const Bias2 bv = load_bias2(bias, wcol0);             // before the prefetch (in-order vmcnt)
    mma_tile_frag<1>(As, f, acc);
    if (Wnext) load_wfrag<TRANS>(f, Wnext, ld_next, wcol0);
    acc_to_lds<1>(acc, Ds, wcol0, bv);
    __syncthreads();
}

__global__ __launch_bounds__(WG) void node_tail_fwd_kernel(const float* __restrict__ x2,
                                                           const float* __restrict__ res_x, int64_t n, TailParams p,
                                                           float* __restrict__ Z, float* __restrict__ R,
                                                           float* __restrict__ x_out, float* __restrict__ out,
                                                           float* __restrict__ att) {
    __shared__ __attribute__((aligned(16))) float lds[4 * SLOT];
    float* S0 = lds;
    float* S1 = lds + SLOT;
    float* S2 = lds + 2 * SLOT;
    float* S3 = lds + 3 * SLOT;
    const int64_t row0 = (int64_t)blockIdx.x * BMN;
    const int64_t plane = n * DIM;

    WFrag wf;
    load_wfrag<false>(wf, p.W[0], DIM, (threadIdx.x >> 6) * 32);
    sweep_rows<BMN>([&](int r, int c4) {
        const int64_t g = row0 + r;
        st_lds4(S0, r, c4, ldg4z(x2, g, n, DIM, c4));
        st_lds4(S3, r, c4, ldg4z(res_x, g, n, DIM, c4));
    });
    __syncthreads();

    // z_k (in slot T) -> save, a = SiLU(z) (+ add1 + add2) -> T ; optional save of a
    auto act = [&](float* T, int k, const float* add1, const float* add2, float* save_a) {
        sweep_rows<BMN>([&](int r, int c4) {
            const int64_t g = row0 + r;
            const float4 z = lds4(T, r, c4);
            float4 a = f4silu(z);
            if (add1) a = f4add(a, lds4(add1, r, c4));
            if (add2) a = f4add(a, lds4(add2, r, c4));
            st_lds4(T, r, c4, a);
            if (g < n) {
                stg4(Z + (int64_t)k * plane, g, DIM, c4, z);
                if (save_a) stg4(save_a, g, DIM, c4, a);
            }
        });
        __syncthreads();
    };

    gemm16<false>(S0, wf, p.b[0], S1, p.W[1]); act(S1, 0, nullptr, nullptr, nullptr);       // h0      -> S1
    gemm16<false>(S1, wf, p.b[1], S2, p.W[2]); act(S2, 1, nullptr, nullptr, nullptr);       // a1      -> S2
    gemm16<false>(S2, wf, p.b[2], S0, p.W[3]); act(S0, 2, S1, S3, R);                       // r1      -> S0
    gemm16<false>(S0, wf, p.b[3], S1, p.W[4]); act(S1, 3, nullptr, nullptr, nullptr);       // a3      -> S1
    gemm16<false>(S1, wf, p.b[4], S2, p.W[5]); act(S2, 4, S0, nullptr, R + plane);          // r2      -> S2
    gemm16<false>(S2, wf, p.b[5], S0, p.W[6]); act(S0, 5, nullptr, nullptr, nullptr);       // a5      -> S0
    gemm16<false>(S0, wf, p.b[6], S1, p.W[7]); act(S1, 6, S2, nullptr, x_out);              // r3      -> S1
    gemm16<false>(S1, wf, p.b[7], S0, p.W[8]); act(S0, 7, nullptr, nullptr, nullptr);       // o1      -> S0
    gemm16<false>(S0, wf, p.b[8], S2, p.W[9]); act(S2, 8, nullptr, nullptr, nullptr);       // o2      -> S2
    gemm16<false>(S2, wf, p.b[9], S0, nullptr); act(S0, 9, nullptr, nullptr, nullptr);      // o3      -> S0

    // heads: 16 lanes per row, 8 columns each, butterfly over the 16-lane group
    {
        const int r = threadIdx.x >> 4, part = threadIdx.x & 15;
        float so = 0.f, sa = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float v = S0[r * LDT + part * 8 + c];
            so += v * p.w_out[part * 8 + c];
            sa += v * p.w_att[part * 8 + c];
        }
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) {
            so += __shfl_xor(so, o, 64);
            sa += __shfl_xor(sa, o, 64);
        }
        const int64_t g = row0 + r;
        if (part == 0 && g < n) {
            out[g] = so + p.b_out[0];
            att[g] = sa;
        }
    }
}

// Backward of the chain above.  Produces dZ_k for every layer (consumed by the batched weight-gradient kernel),
// d x2 (gradient of the chain input), d res_x, and per-workgroup partial sums for the two head vectors.
__global__ __launch_bounds__(WG) void node_tail_bwd_kernel(const float* __restrict__ d_xout /* may be null */,
                                                           const float* __restrict__ d_out,
                                                           const float* __restrict__ d_att, int64_t n, TailParams p,
                                                           const float* __restrict__ Z, float* __restrict__ dZ,
                                                           float* __restrict__ d_x2, float* __restrict__ d_resx,
                                                           float* __restrict__ head_partial /* [grid][257] */) {
    __shared__ __attribute__((aligned(16))) float lds[3 * SLOT + 8 * 256];
    float* S0 = lds;
    float* S1 = lds + SLOT;
    float* S2 = lds + 2 * SLOT;
    float* red = lds + 3 * SLOT;
    const int64_t row0 = (int64_t)blockIdx.x * BMN;
    const int64_t plane = n * DIM;

    // S0 = d o3 = d_out * w_out + d_att * w_att ; S2 = d x_out (gradient arriving at r3 from the next layer)
    sweep_rows<BMN>([&](int r, int c4) {
        const int64_t g = row0 + r;
        float4 v = f4zero(), u = f4zero();
        if (g < n) {
            const float go = d_out[g], ga = d_att[g];
            const float4 wo = *reinterpret_cast<const float4*>(p.w_out + 4 * c4);
            const float4 wa = *reinterpret_cast<const float4*>(p.w_att + 4 * c4);
            v = make_float4(go * wo.x + ga * wa.x, go * wo.y + ga * wa.y, go * wo.z + ga * wa.z, go * wo.w + ga * wa.w);
            if (d_xout) u = ldg4(d_xout, g, DIM, c4);
        }
        st_lds4(S0, r, c4, v);
        st_lds4(S2, r, c4, u);
    });
    __syncthreads();

    // d a (S0) [+ S2, optionally re-stored to S2 / a global tensor]  ->  dz_k = d a * SiLU'(z_k)  -> S1 and dZ_k
    WFrag wf;
    load_wfrag<true>(wf, p.W[9], DIM, (threadIdx.x >> 6) * 32);
    auto back = [&](int k, bool add_s2, bool keep_s2, float* extra_out) {
        sweep_rows<BMN>([&](int r, int c4) {
            const int64_t g = row0 + r;
            float4 da = lds4(S0, r, c4);
            if (add_s2) da = f4add(da, lds4(S2, r, c4));
            if (keep_s2) st_lds4(S2, r, c4, da);
            float4 dz = f4zero();
            if (g < n) {
                dz = f4mul(da, f4dsilu(ldg4(Z + (int64_t)k * plane, g, DIM, c4)));
                stg4(dZ + (int64_t)k * plane, g, DIM, c4, dz);
                if (extra_out) stg4(extra_out, g, DIM, c4, da);
            }
            st_lds4(S1, r, c4, dz);
        });
        __syncthreads();
        gemm16<true>(S1, wf, nullptr, S0, k > 0 ? p.W[k - 1] : nullptr);   // d(input of layer k) = dz_k * W_k
    };

    // head-vector partials: sum_rows d_out * o3, sum_rows d_att * o3, sum_rows d_out  (o3 = SiLU(z9))
    {
        const int c4 = threadIdx.x & 31, r0 = threadIdx.x >> 5;
        float4 so = f4zero(), sa = f4zero();
        float sb = 0.f;
#pragma unroll
        for (int i = 0; i < BMN / 8; ++i) {
            const int64_t g = row0 + r0 + 8 * i;
            if (g < n) {
                const float4 o3 = f4silu(ldg4(Z + 9 * plane, g, DIM, c4));
                const float go = d_out[g], ga = d_att[g];
                so = f4add(so, make_float4(go * o3.x, go * o3.y, go * o3.z, go * o3.w));
                sa = f4add(sa, make_float4(ga * o3.x, ga * o3.y, ga * o3.z, ga * o3.w));
                if (c4 == 0) sb += go;
            }
        }
        float* mine = red + r0 * 256;
        *reinterpret_cast<float4*>(mine + 4 * c4) = so;
        *reinterpret_cast<float4*>(mine + 128 + 4 * c4) = sa;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) tot += red[q * 256 + threadIdx.x];
        head_partial[(int64_t)blockIdx.x * 257 + threadIdx.x] = tot;
        // d b_out: rows r0 + 8i with c4 == 0 hold the pieces -> lanes with c4 == 0 are threads 0,32,...,224
        __syncthreads();
        if (c4 == 0) red[r0] = sb;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.f;
            for (int q = 0; q < 8; ++q) t += red[q];
            head_partial[(int64_t)blockIdx.x * 257 + 256] = t;
        }
        __syncthreads();
    }

    back(9, false, false, nullptr);
    back(8, false, false, nullptr);
    back(7, false, false, nullptr);
    back(6, true, true, nullptr);            // d r3 = S0 + d x_out            -> kept in S2
    back(5, false, false, nullptr);
    back(4, true, true, nullptr);            // d r2 = S0 + d r3
    back(3, false, false, nullptr);
    back(2, true, true, d_resx);             // d r1 = S0 + d r2  (= d res_x)
    back(1, false, false, nullptr);
    back(0, true, false, nullptr);           // d h0 = S0 + d r1
    sweep_rows<BMN>([&](int r, int c4) {
        const int64_t g = row0 + r;
        if (g < n) stg4(d_x2, g, DIM, c4, lds4(S0, r, c4));
    });
}

// columns: [0,128) d w_out, [128,256) d w_att, [256] d b_out
// one workgroup per 16 columns: 16 row-slices x 16 columns, fixed-order tree in LDS (deterministic)
__global__ __launch_bounds__(WG) void head_reduce_kernel(const float* __restrict__ partial, int nblocks,
                                                         float* __restrict__ d_wout, float* __restrict__ d_watt,
                                                         float* __restrict__ d_bout) {
    __shared__ float red[16][17];
    const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    float s = 0.f;
    if (c < 257)
        for (int b = sl; b < nblocks; b += 16) s += partial[(int64_t)b * 257 + c];
    red[sl][cl] = s;
    __syncthreads();
    if (sl == 0 && c < 257) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += red[q][cl];
        if (c < 128) d_wout[c] = t;
        else if (c < 256) d_watt[c - 128] = t;
        else d_bout[0] = t;
    }
}

// ---- head of a layer: x1 = SiLU(Lx1 x) ; P[:, 128*b : 128*(b+1)] = x1 * Wp_b^T (no bias), b < nblk ----------------
__global__ __launch_bounds__(WG) void node_pre_fwd_kernel(const float* __restrict__ x, int64_t n,
                                                          const float* __restrict__ Wx1,
                                                          const float* __restrict__ bx1,
                                                          const float* wp0, const float* wp1, const float* wp2,
                                                          const float* wp3, int ldwp, int nblk,
                                                          float* __restrict__ Zx1, float* __restrict__ x1,
                                                          float* __restrict__ P) {
    __shared__ __attribute__((aligned(16))) float lds[3 * SLOT];
    float* S0 = lds;
    float* S1 = lds + SLOT;
    float* S2 = lds + 2 * SLOT;
    const int64_t row0 = (int64_t)blockIdx.x * BMN;
    sweep_rows<BMN>([&](int r, int c4) {
        const int64_t g = row0 + r;
        st_lds4(S0, r, c4, ldg4z(x, g, n, DIM, c4));
    });
    __syncthreads();
    const float* wps[4] = {wp0, wp1, wp2, wp3};
    WFrag wf;
    load_wfrag<false>(wf, Wx1, DIM, (threadIdx.x >> 6) * 32);
    gemm16<false>(S0, wf, bx1, S1, wps[0], ldwp);
    sweep_rows<BMN>([&](int r, int c4) {
        const int64_t g = row0 + r;
        const float4 z = lds4(S1, r, c4);
        const float4 a = f4silu(z);
        st_lds4(S1, r, c4, a);
        if (g < n) {
            stg4(Zx1, g, DIM, c4, z);
            stg4(x1, g, DIM, c4, a);
        }
    });
    __syncthreads();
    const int64_t plane_p = n * DIM;                       // P is stored as nblk planes [N][128]
    for (int b = 0; b < nblk; ++b) {
        gemm16<false>(S1, wf, nullptr, S2, b + 1 < nblk ? wps[b + 1] : nullptr, ldwp);
        sweep_rows<BMN>([&](int r, int c4) {
            const int64_t g = row0 + r;
            if (g < n) stg4(P + (int64_t)b * plane_p, g, DIM, c4, lds4(S2, r, c4));
        });
        __syncthreads();
    }
}

// backward of the head: d x1 = dP * Wp + d x1_direct ; dz = d x1 * SiLU'(z_x1) ; d x = dz * Wx1 (+ d_add)
__global__ __launch_bounds__(WG) void node_pre_bwd_kernel(const float* __restrict__ dP, const float* __restrict__ dx1_direct,
                                                          const float* __restrict__ d_add /* may be null */, int64_t n,
                                                          const float* __restrict__ Wx1, const float* wp0,
                                                          const float* wp1, const float* wp2, const float* wp3,
                                                          int ldwp, int nblk, const float* __restrict__ Zx1,
                                                          float* __restrict__ dZx1, float* __restrict__ dx) {
    __shared__ __attribute__((aligned(16))) float lds[3 * SLOT];
    float* S0 = lds;
    float* S1 = lds + SLOT;
    float* S2 = lds + 2 * SLOT;
    const int64_t row0 = (int64_t)blockIdx.x * BMN;
    const float* wps[4] = {wp0, wp1, wp2, wp3};
    const int64_t plane_p = n * DIM;                       // dP: nblk planes [N][128]
    const int wcol0 = (threadIdx.x >> 6) * 32;
    f32x4 acc[1][2];
    acc_zero<1>(acc);
    WFrag wf;
    load_wfrag<true>(wf, wps[0], ldwp, wcol0);
    for (int b = 0; b < nblk; ++b) {
        sweep_rows<BMN>([&](int r, int c4) {
            const int64_t g = row0 + r;
            st_lds4(S0, r, c4, ldg4z(dP + (int64_t)b * plane_p, g, n, DIM, c4));
        });
        __syncthreads();
        mma_tile_frag<1>(S0, wf, acc);                        // accumulate over the projection blocks
        if (b + 1 < nblk) load_wfrag<true>(wf, wps[b + 1], ldwp, wcol0);
        else load_wfrag<true>(wf, Wx1, DIM, wcol0);           // weights of the final GEMM: in flight during the sweep
        __syncthreads();
    }
    acc_to_lds<1>(acc, S1, wcol0, load_bias2(nullptr, wcol0));
    __syncthreads();
    sweep_rows<BMN>([&](int r, int c4) {
        const int64_t g = row0 + r;
        float4 dz = f4zero();
        if (g < n) {
            float4 d1 = lds4(S1, r, c4);
            if (dx1_direct) d1 = f4add(d1, ldg4(dx1_direct, g, DIM, c4));
            dz = f4mul(d1, f4dsilu(ldg4(Zx1, g, DIM, c4)));
            stg4(dZx1, g, DIM, c4, dz);
        }
        st_lds4(S2, r, c4, dz);
    });
    __syncthreads();
    gemm16<true>(S2, wf, nullptr, S0, nullptr);
    sweep_rows<BMN>([&](int r, int c4) {
        const int64_t g = row0 + r;
        if (g < n) {
            float4 v = lds4(S0, r, c4);
            if (d_add) v = f4add(v, ldg4(d_add, g, DIM, c4));
            stg4(dx, g, DIM, c4, v);
        }
    });
}

inline TailParams make_tail(const float* const* weights, const float* const* biases, const float* w_out,
                            const float* b_out, const float* w_att) {
    TailParams p;
    for (int k = 0; k < 10; ++k) {
        p.W[k] = weights[k];
        p.b[k] = biases ? biases[k] : nullptr;
    }
    p.w_out = w_out;
    p.b_out = b_out;
    p.w_att = w_att;
    return p;
}

}  // namespace

extern "C" int pamnet_node_tail_fwd_f32(const float* x2, const float* res_x, int64_t n, const float* const* weights,
                                        const float* const* biases, const float* w_out, const float* b_out,
                                        const float* w_att, float* Z, float* R, float* x_out, float* out, float* att,
                                        pamnet_stream_t stream) {
    if (n < 0) return PAMNET_EINVAL;
    if (n == 0) return PAMNET_OK;
    if (!x2 || !res_x || !weights || !biases || !w_out || !b_out || !w_att || !Z || !R || !x_out || !out || !att)
        return PAMNET_ENULL;
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(node_tail_fwd_kernel, dim3((unsigned)ceil_div(n, BMN)), dim3(WG), 0, st, x2, res_x, n,
                       make_tail(weights, biases, w_out, b_out, w_att), Z, R, x_out, out, att);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_node_tail_bwd_f32(const float* d_xout, const float* d_out, const float* d_att, int64_t n,
                                        const float* const* weights, const float* w_out, const float* w_att,
                                        const float* Z, float* dZ, float* d_x2, float* d_resx, float* head_partial,
                                        float* d_wout, float* d_watt, float* d_bout, pamnet_stream_t stream) {
    if (n < 0) return PAMNET_EINVAL;
    if (n == 0) return PAMNET_OK;
    if (!d_out || !d_att || !weights || !w_out || !w_att || !Z || !dZ || !d_x2 || !d_resx || !head_partial || !d_wout ||
        !d_watt || !d_bout)
        return PAMNET_ENULL;
    hipStream_t st = as_stream(stream);
    const unsigned grid = (unsigned)ceil_div(n, BMN);
    hipLaunchKernelGGL(node_tail_bwd_kernel, dim3(grid), dim3(WG), 0, st, d_xout, d_out, d_att, n,
                       make_tail(weights, nullptr, w_out, nullptr, w_att), Z, dZ, d_x2, d_resx, head_partial);
    PAMNET_LAUNCH_CHECK();
    hipLaunchKernelGGL(head_reduce_kernel, dim3(17), dim3(WG), 0, st, head_partial, (int)grid, d_wout, d_watt, d_bout);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_node_pre_fwd_f32(const float* x, int64_t n, const float* Wx1, const float* bx1,
                                       const float* const* wp, int64_t ldwp, int64_t nblk, float* Zx1, float* x1,
                                       float* P, pamnet_stream_t stream) {
    if (n < 0 || nblk < 1 || nblk > 4) return PAMNET_EINVAL;
    if (n == 0) return PAMNET_OK;
    if (!x || !Wx1 || !bx1 || !wp || !Zx1 || !x1 || !P) return PAMNET_ENULL;
    const float* w[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int b = 0; b < nblk; ++b) w[b] = wp[b];
    hipLaunchKernelGGL(node_pre_fwd_kernel, dim3((unsigned)ceil_div(n, BMN)), dim3(WG), 0, as_stream(stream), x, n, Wx1,
                       bx1, w[0], w[1], w[2], w[3], (int)ldwp, (int)nblk, Zx1, x1, P);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_node_pre_bwd_f32(const float* dP, const float* dx1_direct, const float* d_add, int64_t n,
                                       const float* Wx1, const float* const* wp, int64_t ldwp, int64_t nblk,
                                       const float* Zx1, float* dZx1, float* dx, pamnet_stream_t stream) {
    if (n < 0 || nblk < 1 || nblk > 4) return PAMNET_EINVAL;
    if (n == 0) return PAMNET_OK;
    if (!dP || !Wx1 || !wp || !Zx1 || !dZx1 || !dx) return PAMNET_ENULL;
    const float* w[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int b = 0; b < nblk; ++b) w[b] = wp[b];
    hipLaunchKernelGGL(node_pre_bwd_kernel, dim3((unsigned)ceil_div(n, BMN)), dim3(WG), 0, as_stream(stream), dP,
                       dx1_direct, d_add, n, Wx1, w[0], w[1], w[2], w[3], (int)ldwp, (int)nblk, Zx1, dZx1, dx);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}
