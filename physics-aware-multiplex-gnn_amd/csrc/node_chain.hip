// Fused node-level head of PAMNet's message-passing layers (dim = 128), forward and backward (the 10-Linear tail
// lives in node_tail.hip).
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
            if (Zx1) stg4(Zx1, g, DIM, c4, z);                // backward-only save: null in inference mode
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
template <bool PACKED>
__global__ __launch_bounds__(WG) void node_pre_bwd_kernel(const float* __restrict__ dP, const float* __restrict__ dx1_direct,
                                                          const float* __restrict__ d_add /* may be null */, int64_t n,
                                                          const float* __restrict__ Wx1, const float* wp0,
                                                          const float* wp1, const float* wp2, const float* wp3,
                                                          int ldwp, int nblk, const float* __restrict__ Zx1,
                                                          float* __restrict__ dZx1, float* __restrict__ dx) {
    // every input tile (up to 4 projection-gradient planes, z_x1, the direct d x1 and the residual gradient) is
    // requested in ONE burst up front: the chain below then holds no global load except the weight slices
    __shared__ __attribute__((aligned(16))) float lds[9 * SLOT];
    float* S0 = lds;                       // GEMM output / final d x
    float* S1 = lds + SLOT;
    float* S2 = lds + 2 * SLOT;
    float* PL = lds + 3 * SLOT;            // [4] dP planes
    float* ZT = lds + 7 * SLOT;            // z_x1 tile
    float* DT = lds + 8 * SLOT;            // d x1_direct tile
    const int64_t row0 = (int64_t)blockIdx.x * BMN;
    const float* wps[4] = {wp0, wp1, wp2, wp3};
    const int64_t plane_p = n * DIM;                       // dP: nblk planes [N][128]
    const int wcol0 = (threadIdx.x >> 6) * 32;
    f32x4 acc[1][2];
    acc_zero<1>(acc);
    WFrag wf;
    load_w<PACKED, true>(wf, wps[0], ldwp, wcol0);
    sweep_rows<BMN>([&](int r, int c4) {
        const int64_t g = row0 + r;
        for (int b = 0; b < nblk; ++b) st_lds4(PL + b * SLOT, r, c4, ldg4z(dP + (int64_t)b * plane_p, g, n, DIM, c4));
        st_lds4(ZT, r, c4, ldg4z(Zx1, g, n, DIM, c4));
        st_lds4(DT, r, c4, dx1_direct ? ldg4z(dx1_direct, g, n, DIM, c4) : f4zero());
        st_lds4(S0, r, c4, d_add ? ldg4z(d_add, g, n, DIM, c4) : f4zero());
    });
    __syncthreads();
    for (int b = 0; b < nblk; ++b) {
        mma_tile_frag<1>(PL + b * SLOT, wf, acc);             // accumulate over the projection blocks
        if (b + 1 < nblk) load_w<PACKED, true>(wf, wps[b + 1], ldwp, wcol0);
        else load_w<PACKED, true>(wf, Wx1, DIM, wcol0);       // weights of the final GEMM
    }
    acc_to_lds<1>(acc, S1, wcol0, load_bias2(nullptr, wcol0));
    __syncthreads();
    sweep_rows<BMN>([&](int r, int c4) {
        const float4 d1 = f4add(lds4(S1, r, c4), lds4(DT, r, c4));
        st_lds4(S2, r, c4, f4mul(d1, f4dsilu(lds4(ZT, r, c4))));      // dz (zero rows beyond n: dP, DT are zero there)
    });
    __syncthreads();
    {
        f32x4 a2[1][2];
        acc_zero<1>(a2);
        mma_tile_frag<1>(S2, wf, a2);
        acc_to_lds<1>(a2, S1, wcol0, load_bias2(nullptr, wcol0));
    }
    __syncthreads();
    sweep_rows<BMN>([&](int r, int c4) {
        const int64_t g = row0 + r;
        if (g >= n) return;
        stg4(dZx1, g, DIM, c4, lds4(S2, r, c4));
        stg4(dx, g, DIM, c4, f4add(lds4(S1, r, c4), lds4(S0, r, c4)));
    });
}

}  // namespace

extern "C" int pamnet_node_pre_fwd_f32(const float* x, int64_t n, const float* Wx1, const float* bx1,
                                       const float* const* wp, int64_t ldwp, int64_t nblk, float* Zx1, float* x1,
                                       float* P, pamnet_stream_t stream) {
    if (n < 0 || nblk < 1 || nblk > 4) return PAMNET_EINVAL;
    if (n == 0) return PAMNET_OK;
    if (!x || !Wx1 || !bx1 || !wp || !x1 || !P) return PAMNET_ENULL;           // Zx1: optional save
    const float* w[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int b = 0; b < nblk; ++b) w[b] = wp[b];
    hipLaunchKernelGGL(node_pre_fwd_kernel, dim3((unsigned)ceil_div(n, BMN)), dim3(WG), 0, as_stream(stream), x, n, Wx1,
                       bx1, w[0], w[1], w[2], w[3], (int)ldwp, (int)nblk, Zx1, x1, P);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_node_pre_bwd_f32(const float* dP, const float* dx1_direct, const float* d_add, int64_t n,
                                       const float* Wx1, const float* const* wp, int64_t ldwp, int64_t nblk,
                                       const float* Zx1, float* dZx1, float* dx, int32_t packed,
                                       pamnet_stream_t stream) {
    if (n < 0 || nblk < 1 || nblk > 4) return PAMNET_EINVAL;
    if (n == 0) return PAMNET_OK;
    if (!dP || !Wx1 || !wp || !Zx1 || !dZx1 || !dx) return PAMNET_ENULL;
    const float* w[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int b = 0; b < nblk; ++b) w[b] = wp[b];
    if (packed)
        hipLaunchKernelGGL(node_pre_bwd_kernel<true>, dim3((unsigned)ceil_div(n, BMN)), dim3(WG), 0, as_stream(stream), dP,
                           dx1_direct, d_add, n, Wx1, w[0], w[1], w[2], w[3], (int)ldwp, (int)nblk, Zx1, dZx1, dx);
    else
        hipLaunchKernelGGL(node_pre_bwd_kernel<false>, dim3((unsigned)ceil_div(n, BMN)), dim3(WG), 0, as_stream(stream),
                           dP, dx1_direct, d_add, n, Wx1, w[0], w[1], w[2], w[3], (int)ldwp, (int)nblk, Zx1, dZx1, dx);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}
