// Node-level chains of the narrow-width (dim = 16 / 32 / 64) layers as single launches.
//
// A PAMNet layer at these widths spends its node-side work (N rows: a few thousand to a few ten thousand) in ~20 dense
// d x d layers per direction: mlp_x1 + the node-side projections of the message weights before the edge kernels, and
// mlp_x2 -> Res1..3 -> mlp_out -> the two heads after them (layers/global_message_passing.py:33-50,
// layers/local_message_passing.py:36-66).  One launch per dense layer is ~7-35 us of latency for ~1 us of arithmetic
// (rocprof: 0.19 ms forward / 0.6 ms backward per layer at d = 64, N = 17.7 k).  Here a wavefront carries its 16-row
// tile through the whole chain in registers (the row kernels' design, narrow_core.h) and the workgroup rebuilds the
// weight images stage by stage:
//   npre_fwd / npre_bwd   : x1 = SiLU(W1 x + b1);  P[:, k d:(k+1) d] = x1 Wp_k^T   (k < NB <= 4)
//   ntail_fwd / ntail_bwd : h0 = SiLU(L0 x2); r1 = Res1(h0) + x; r2 = Res2(r1); r3 = Res3(r2);
//                           o = mlp_out(r3); out = o . w_out + b_out; att = o . w_att
// The backward kernels recompute pre-activations from the saved stage inputs, form every weight gradient in the same pass
// and leave one partial-gradient row per workgroup (waves summed in wave order); narrow_reduce_multi_kernel adds the rows
// in a fixed order and scatters the matrices / vectors straight into the parameters' gradient buffers (column blocks of
// the 3d-wide message weights included) -- deterministic, no atomics.
#pragma once
#include "narrow_core.h"

namespace {

// Waves per workgroup of the chain kernels (128 rows): two per SIMD.  A wave walks ~10 dependent stages for its one tile
// (load -> MFMA -> LDS relayout -> MFMA ...), so a second wave per SIMD is what hides the latencies; and 17.7 k rows are
// 139 workgroups -- one round on 256 CUs, where 64-row workgroups (277) took two (ntail_bwd, d = 64: 220 us).
constexpr int CHW = 8;
// widths whose whole tail chain (ten [d, d] images, twenty with the transposes) stays in LDS
__host__ __device__ constexpr bool tail_resident(int d) { return d <= 32; }

// The chain kernels visit a new weight matrix every stage and only a handful of row tiles per workgroup, so building the
// fragment-ordered images from the row-major weights inside them (strided 4-byte reads) was most of their time
// (ntail_bwd at d = 64, N = 17.7 k: 220 us with in-kernel builds).  npack_kernel builds every image of a layer pair once per
// call -- plain and transposed -- into a scratch area; the chains copy them into LDS with coalesced 16-byte loads.
template <int NT>
__device__ __forceinline__ void copy_image(float4* dst, const float4* __restrict__ src) {
    for (int i = threadIdx.x; i < NT * NT * 64; i += blockDim.x) dst[i] = src[i];
}

constexpr int PACK_SLOTS = 32;              // [D, D] weight blocks of one layer pair (see narrow_engine.hip)
struct PackJobs {
    const float* W[PACK_SLOTS];
    int ld[PACK_SLOTS];
};
// grid (PACK_SLOTS, 2): image (slot, transposed?) -> out[(2 * slot + t) * IMG]
template <int D>
__global__ __launch_bounds__(256) void npack_kernel(const PackJobs jobs, float4* __restrict__ out) {
    constexpr int NT = D / 16;
    constexpr int IMG = NT * NT * 64;
    const int slot = blockIdx.x, t = blockIdx.y;
    const float* W = jobs.W[slot];
    if (!W) return;
    float4* img = out + (size_t)(2 * slot + t) * IMG;
    if (t == 0) build_image<NT, NT, false>(img, W, jobs.ld[slot], D);
    else build_image<NT, NT, true>(img, W, jobs.ld[slot], D);
}

template <int NT>
__device__ __forceinline__ void load_bias(float (&bj)[NT], const float* __restrict__ b, int c) {
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) bj[jt] = b ? b[16 * jt + c] : 0.f;
}

template <int NT>
__device__ __forceinline__ void add_bias_silu(f32x4 (&v)[NT], const float (&bj)[NT]) {
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float z = v[jt][r] + bj[jt];
            v[jt][r] = z * sigmoidf_fast(z);
        }
}

// g *= SiLU'(z + b)
template <int NT>
__device__ __forceinline__ void mul_dsilu(f32x4 (&g)[NT], const f32x4 (&z)[NT], const float (&bj)[NT]) {
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float zz = z[jt][r] + bj[jt];
            const float s = sigmoidf_fast(zz);
            g[jt][r] *= s * (1.0f + zz * (1.0f - s));
        }
}

// Sum the waves' contributions (wave order), then hand the n floats to the workgroup's partial row (`add`: this is a later
// row group of the same workgroup).  Ends with a barrier: `red`, the images and the tiles may be rewritten afterwards.
//   WIDE (d <= 32): every wave writes its part to a slot of its own, one barrier, then the threads add the CHW slots of an
//   element in wave order -- the same sums, bit for bit, as the narrow form's chain, in two barriers instead of CHW + 1 (the
//   tail chain's backward runs eleven of these per row group: a third of its time at the RNA batch).
//   narrow (d = 64, no room for CHW slots): `put(red, first)` writes / adds this wave's part in its turn.
template <bool WIDE, typename F>
__device__ __forceinline__ void wg_reduce(float* red, int n, float* __restrict__ dst, bool add, F&& put) {
    const int wave = threadIdx.x >> 6;
    if constexpr (WIDE) {
        put(red + wave * n, true);
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += 64 * CHW) {
            float s = red[i];
#pragma unroll
            for (int w = 1; w < CHW; ++w) s += red[w * n + i];
            dst[i] = add ? dst[i] + s : s;
        }
    } else {
        for (int w = 0; w < CHW; ++w) {
            if (wave == w) put(red, w == 0);
            __syncthreads();
        }
        for (int i = threadIdx.x; i < n; i += 64 * CHW) dst[i] = add ? dst[i] + red[i] : red[i];
    }
    __syncthreads();
}

// ---- backward stages: g (accumulator layout, zero on rows >= m) is the gradient w.r.t. the stage's output on entry and
// w.r.t. its input on exit ---------------------------------------------------------------------------------------------
// y = SiLU(x W^T + b) (act) or x W^T (+ b);  partial segment: [dW fragments D x D][db D]
template <int D>
__device__ __forceinline__ void lin_bwd_stage(f32x4 (&g)[D / 16], const float* __restrict__ X, int64_t row0, int64_t m,
                                              const float4* img, const float4* imgt, const float* __restrict__ b,
                                              bool act, float* tile, float* red, float* dst, bool add, int lane) {
    constexpr int NT = D / 16;
    float4 a[NT];
    load_a<D>(a, X, row0, m, lane);
    if (act) {
        f32x4 z[NT];
        float bj[NT];
        load_bias<NT>(bj, b, lane & 15);
        zero(z);
        mma_img<NT, NT>(z, a, img, lane);
        mul_dsilu<NT>(g, z, bj);
    }
    f32x4 xd[NT], gw[NT][NT];
    a_to_d<D>(xd, a, tile, lane);
    float dbs[NT];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
        dbs[jt] = 0.f;
        zero(gw[jt]);
    }
    wgrad_acc<NT, NT>(gw, g, xd);
    colsum_acc<NT>(dbs, g);
    f32x4 dx[NT];
    d_to_a<D>(a, g, tile, lane);
    zero(dx);
    mma_img<NT, NT>(dx, a, imgt, lane);
    wg_reduce<tail_resident(D)>(red, D * D + D, dst, add, [&](float* r, bool first) {
        red_add_mat<NT, NT>(r, gw, lane, first);
        red_add_bias<NT>(r + D * D, dbs, lane, first);
    });
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) g[jt] = dx[jt];
}

// y = SiLU(W2 SiLU(W1 x + b1) + b2) (+ x);  partial segment: [dW2][db2][dW1][db1] (the second layer's gradients leave
// the registers before the first layer's are formed: 256 registers per lane at two waves per SIMD)
template <int D>
__device__ __forceinline__ void mlp2_bwd_stage(f32x4 (&g)[D / 16], const float* __restrict__ X, int64_t row0, int64_t m,
                                               const float4* img1, const float4* img2, const float4* img1t,
                                               const float4* img2t, const float* __restrict__ b1,
                                               const float* __restrict__ b2, bool res, float* tile, float* red,
                                               float* dst, bool add, int lane) {
    constexpr int NT = D / 16;
    constexpr int MAT = D * D;
    float bj[NT], dbs[NT];
    float4 a[NT], ah[NT];
    load_a<D>(a, X, row0, m, lane);
    f32x4 z1[NT], h[NT], q[NT];
    zero(z1);
    mma_img<NT, NT>(z1, a, img1, lane);
    load_bias<NT>(bj, b1, lane & 15);
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            z1[jt][r] += bj[jt];
            h[jt][r] = z1[jt][r] * sigmoidf_fast(z1[jt][r]);
        }
    d_to_a<D>(ah, h, tile, lane);
    zero(q);
    mma_img<NT, NT>(q, ah, img2, lane);                      // z2 - b2
    load_bias<NT>(bj, b2, lane & 15);
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float z = q[jt][r] + bj[jt];
            const float s = sigmoidf_fast(z);
            q[jt][r] = g[jt][r] * (s * (1.0f + z * (1.0f - s)));   // dz2 (zero on padded rows: g = 0)
        }
    {
        f32x4 gw[NT][NT];
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            dbs[jt] = 0.f;
            zero(gw[jt]);
        }
        wgrad_acc<NT, NT>(gw, q, h);
        colsum_acc<NT>(dbs, q);
        wg_reduce<tail_resident(D)>(red, MAT + D, dst, add, [&](float* r, bool first) {
            red_add_mat<NT, NT>(r, gw, lane, first);
            red_add_bias<NT>(r + MAT, dbs, lane, first);
        });
    }
    d_to_a<D>(ah, q, tile, lane);
    zero(q);
    mma_img<NT, NT>(q, ah, img2t, lane);                     // dh1
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float z = z1[jt][r];
            const float s = sigmoidf_fast(z);
            q[jt][r] *= s * (1.0f + z * (1.0f - s));         // dz1
        }
    a_to_d<D>(h, a, tile, lane);                             // x in accumulator layout
    {
        f32x4 gw[NT][NT];
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            dbs[jt] = 0.f;
            zero(gw[jt]);
        }
        wgrad_acc<NT, NT>(gw, q, h);
        colsum_acc<NT>(dbs, q);
        wg_reduce<tail_resident(D)>(red, MAT + D, dst + MAT + D, add, [&](float* r, bool first) {
            red_add_mat<NT, NT>(r, gw, lane, first);
            red_add_bias<NT>(r + MAT, dbs, lane, first);
        });
    }
    d_to_a<D>(ah, q, tile, lane);
    zero(q);
    mma_img<NT, NT>(q, ah, img1t, lane);                     // dx
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) g[jt] = res ? q[jt] + g[jt] : q[jt];
}

// ====================================================================================================================
// Tail: mlp_x2 -> Res1 (+ layer input) -> Res2 -> Res3 -> mlp_out -> heads
// W / b in chain order: [0] mlp_x2, [1,2] res1, [3,4] res2, [5,6] res3, [7,8,9] mlp_out
// ====================================================================================================================
struct NTailFwd {
    const float4* img[10];                  // packed images of the ten weights (npack_kernel)
    const float* b[10];
    const float *w_out, *b_out, *w_att;
    const float *x2, *res_x;
    float *H0, *R1, *R2, *R3, *T, *O;       // saved stage inputs of the backward; R3 is the layer's node output
    float *out, *att;
    int64_t m;
};

template <int D>
__device__ __forceinline__ void mlp2_fwd_stage(f32x4 (&v)[D / 16], const float4* img1, const float4* img2,
                                               const float* __restrict__ b1, const float* __restrict__ b2, bool res,
                                               float* tile, int lane) {
    constexpr int NT = D / 16;
    float bj[NT];
    float4 a[NT];
    f32x4 h[NT], o[NT];
    d_to_a<D>(a, v, tile, lane);
    zero(h);
    mma_img<NT, NT>(h, a, img1, lane);
    load_bias<NT>(bj, b1, lane & 15);
    add_bias_silu<NT>(h, bj);
    d_to_a<D>(a, h, tile, lane);
    zero(o);
    mma_img<NT, NT>(o, a, img2, lane);
    load_bias<NT>(bj, b2, lane & 15);
    add_bias_silu<NT>(o, bj);
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) v[jt] = res ? o[jt] + v[jt] : o[jt];
}

template <int D>
__global__ __launch_bounds__(64 * CHW) void ntail_fwd_kernel(const NTailFwd p) {
    constexpr int NT = D / 16;
    constexpr int IMG = NT * NT * 64;
    // d <= 32: all ten images of the chain are resident in LDS (10 / 40 KB), copied once ahead of the rows.  Stage by stage
    // into two slots -- what d = 64 has room for -- every stage began with a global round trip between two barriers: six per
    // group, ~40 % of the kernel at the RNA batch (139 workgroups of one group each).
    constexpr bool RES = tail_resident(D);
    extern __shared__ float4 lds4[];
    float4* img0 = lds4;
    float4* img1 = lds4 + IMG;
    float* tile = reinterpret_cast<float*>(lds4 + (RES ? 10 : 2) * IMG) + (threadIdx.x >> 6) * 16 * (D + 4);
    // the ten bias vectors wait in LDS too: as global loads inside the stages each sat behind the previous stage's stores
    // (the compiler cannot move it above them) -- one more round trip per layer of the chain
    float* bl = reinterpret_cast<float*>(lds4 + (RES ? 10 : 2) * IMG) + CHW * 16 * (D + 4);
    const int lane = threadIdx.x & 63, c = lane & 15, kg = lane >> 4, wave = threadIdx.x >> 6;
    const int64_t m = p.m, ntiles = (m + 15) / 16;
    for (int i = threadIdx.x; i < 10 * D; i += 64 * CHW) bl[i] = p.b[i / D] ? p.b[i / D][i % D] : 0.f;
    if constexpr (RES) {
#pragma unroll 1
        for (int k = 0; k < 10; ++k) copy_image<NT>(lds4 + k * IMG, p.img[k]);
    }
    __syncthreads();
    for (int64_t grp = blockIdx.x; grp * CHW < ntiles; grp += gridDim.x) {
        const int64_t row0 = (grp * CHW + wave) * 16;
        if constexpr (!RES) {
            __syncthreads();
            copy_image<NT>(img0, p.img[0]);
            __syncthreads();
        }
        f32x4 v[NT];
        {
            float4 a[NT];
            float bj[NT];
            load_a<D>(a, p.x2, row0, m, lane);
            zero(v);
            mma_img<NT, NT>(v, a, img0, lane);
            load_bias<NT>(bj, bl, c);
            add_bias_silu<NT>(v, bj);
        }
        store_d<D>(v, p.H0, row0, m, lane);
#pragma unroll 1
        for (int k = 0; k < 4; ++k) {                        // res1, res2, res3, first two layers of mlp_out
            if constexpr (!RES) {
                __syncthreads();
                copy_image<NT>(img0, p.img[1 + 2 * k]);
                copy_image<NT>(img1, p.img[2 + 2 * k]);
                __syncthreads();
            }
            mlp2_fwd_stage<D>(v, RES ? lds4 + (1 + 2 * k) * IMG : img0, RES ? lds4 + (2 + 2 * k) * IMG : img1,
                              bl + (1 + 2 * k) * D, bl + (2 + 2 * k) * D, k < 3, tile, lane);
            if (k == 0) {                                    // Res1(h0) + the layer's input (basic.py:32, *_message_passing.py:41)
                f32x4 rx[NT];
                load_d<D>(rx, p.res_x, row0, m, lane);
#pragma unroll
                for (int jt = 0; jt < NT; ++jt) v[jt] += rx[jt];
            }
            float* dst = k == 0 ? p.R1 : (k == 1 ? p.R2 : (k == 2 ? p.R3 : p.T));
            store_d<D>(v, dst, row0, m, lane);
        }
        if constexpr (!RES) {
            __syncthreads();
            copy_image<NT>(img0, p.img[9]);
            __syncthreads();
        }
        {
            float4 a[NT];
            float bj[NT];
            f32x4 o[NT];
            d_to_a<D>(a, v, tile, lane);
            zero(o);
            mma_img<NT, NT>(o, a, RES ? lds4 + 9 * IMG : img0, lane);
            load_bias<NT>(bj, bl + 9 * D, c);
            add_bias_silu<NT>(o, bj);
            store_d<D>(o, p.O, row0, m, lane);
            // heads: per-row dot products; a row's 16 column lanes share kg
            float so[4] = {0.f, 0.f, 0.f, 0.f}, sa[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                const float wo = p.w_out[16 * jt + c], wa = p.w_att[16 * jt + c];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    so[r] = fmaf(o[jt][r], wo, so[r]);
                    sa[r] = fmaf(o[jt][r], wa, sa[r]);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int s = 8; s >= 1; s >>= 1) {
                    so[r] += __shfl_xor(so[r], s, 64);
                    sa[r] += __shfl_xor(sa[r], s, 64);
                }
            }
            if (c == 0) {
                const float bo = p.b_out[0];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t row = row0 + 4 * kg + r;
                    if (row < m) {
                        p.out[row] = so[r] + bo;
                        p.att[row] = sa[r];
                    }
                }
            }
        }
    }
}

struct NTailBwd {
    const float4* img[10];                  // packed images, plain and transposed
    const float4* imgt[10];
    const float* b[10];
    const float *w_out, *w_att;
    const float *x2, *H0, *R1, *R2, *R3, *T, *O;
    const float *g_x;                       // [m, D] gradient w.r.t. the layer's node output (nullable)
    const float *g_out, *g_att;             // [m]
    float *d_x2, *d_resx;                   // [m, D]
    float* partial;
    int stride;
    int64_t m;
};

// offsets of the stages' segments in a partial row
template <int D>
struct TailRow {
    static constexpr int LIN = D * D + D, MLP = 2 * D * D + 2 * D;
    static constexpr int L0 = 0, B1 = LIN, B2 = B1 + MLP, B3 = B2 + MLP, B4 = B3 + MLP, L5 = B4 + MLP, H = L5 + LIN;
    static constexpr int FLOATS = H + 2 * D + 1;
    static constexpr int STRIDE = (FLOATS + 3) & ~3;
};

template <int D>
__global__ __launch_bounds__(64 * CHW) void ntail_bwd_kernel(const NTailBwd p) {
    constexpr int NT = D / 16;
    constexpr int IMG = NT * NT * 64;
    using Row = TailRow<D>;
    // (d <= 32: all twenty images -- plain at slot k, transposed at slot 10 + k -- resident, as in the forward)
    constexpr bool RES = tail_resident(D);
    constexpr int NSLOT = RES ? 20 : 4;
    extern __shared__ float4 lds4[];
    float4* img = lds4;                                      // (!RES) 4 images
    float* tile = reinterpret_cast<float*>(lds4 + NSLOT * IMG) + (threadIdx.x >> 6) * 16 * (D + 4);
    float* red = reinterpret_cast<float*>(lds4 + NSLOT * IMG) + CHW * 16 * (D + 4);
    float* bl = red + (RES ? CHW : 1) * (D * D + 2 * D + 4);    // the ten bias vectors (as in the forward)
    const int lane = threadIdx.x & 63, c = lane & 15, kg = lane >> 4, wave = threadIdx.x >> 6;
    const int64_t m = p.m, ntiles = (m + 15) / 16;
    float* prow = p.partial + (size_t)blockIdx.x * p.stride;
    for (int i = threadIdx.x; i < 10 * D; i += 64 * CHW) bl[i] = p.b[i / D] ? p.b[i / D][i % D] : 0.f;
    if constexpr (RES) {
#pragma unroll 1
        for (int k = 0; k < 10; ++k) {
            copy_image<NT>(lds4 + k * IMG, p.img[k]);
            copy_image<NT>(lds4 + (10 + k) * IMG, p.imgt[k]);
        }
    }
    __syncthreads();
    for (int64_t grp = blockIdx.x; grp * CHW < ntiles; grp += gridDim.x) {
        const bool add = grp != (int64_t)blockIdx.x;
        const int64_t row0 = (grp * CHW + wave) * 16;
        // ---- heads (global_message_passing.py:47-50): d o = g_out w_out + g_att w_att
        f32x4 g[NT];
        {
            f32x4 o[NT];
            load_d<D>(o, p.O, row0, m, lane);
            float go[4], ga[4], sb = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = row0 + 4 * kg + r;
                const bool ok = row < m;
                go[r] = ok ? p.g_out[row] : 0.f;
                ga[r] = ok ? p.g_att[row] : 0.f;
                sb += go[r];
            }
            float so[NT], sa[NT];
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                const float wo = p.w_out[16 * jt + c], wa = p.w_att[16 * jt + c];
                so[jt] = sa[jt] = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    g[jt][r] = go[r] * wo + ga[r] * wa;
                    so[jt] = fmaf(go[r], o[jt][r], so[jt]);
                    sa[jt] = fmaf(ga[r], o[jt][r], sa[jt]);
                }
            }
            sb += __shfl_xor(sb, 16, 64);
            sb += __shfl_xor(sb, 32, 64);
            wg_reduce<tail_resident(D)>(red, 2 * D + 1, prow + Row::H, add, [&](float* r, bool first) {
                red_add_bias<NT>(r, so, lane, first);
                red_add_bias<NT>(r + D, sa, lane, first);
                if (lane == 0) r[2 * D] = first ? sb : r[2 * D] + sb;
            });
        }
        // ---- mlp_out[2]
        if constexpr (!RES) {
            copy_image<NT>(img, p.img[9]);
            copy_image<NT>(img + IMG, p.imgt[9]);
            __syncthreads();
        }
        lin_bwd_stage<D>(g, p.T, row0, m, RES ? lds4 + 9 * IMG : img, RES ? lds4 + 19 * IMG : img + IMG, bl + 9 * D, true, tile,
                         red, prow + Row::L5, add, lane);
        // ---- mlp_out[0:2], Res3, Res2, Res1
#pragma unroll 1
        for (int k = 3; k >= 0; --k) {
            if constexpr (!RES) {
                copy_image<NT>(img, p.img[1 + 2 * k]);
                copy_image<NT>(img + IMG, p.img[2 + 2 * k]);
                copy_image<NT>(img + 2 * IMG, p.imgt[1 + 2 * k]);
                copy_image<NT>(img + 3 * IMG, p.imgt[2 + 2 * k]);
                __syncthreads();
            }
            if (k == 0) store_d<D>(g, p.d_resx, row0, m, lane);      // r1 = Res1(h0) + res_x
            const float* X = k == 3 ? p.R3 : (k == 2 ? p.R2 : (k == 1 ? p.R1 : p.H0));
            const int off = k == 3 ? Row::B4 : (k == 2 ? Row::B3 : (k == 1 ? Row::B2 : Row::B1));
            mlp2_bwd_stage<D>(g, X, row0, m, RES ? lds4 + (1 + 2 * k) * IMG : img, RES ? lds4 + (2 + 2 * k) * IMG : img + IMG,
                              RES ? lds4 + (11 + 2 * k) * IMG : img + 2 * IMG, RES ? lds4 + (12 + 2 * k) * IMG : img + 3 * IMG,
                              bl + (1 + 2 * k) * D, bl + (2 + 2 * k) * D, k < 3, tile, red, prow + off, add, lane);
            if (k == 3 && p.g_x) {                                   // r3 is also the layer's node output
                f32x4 gx[NT];
                load_d<D>(gx, p.g_x, row0, m, lane);
#pragma unroll
                for (int jt = 0; jt < NT; ++jt) g[jt] += gx[jt];
            }
        }
        // ---- mlp_x2
        if constexpr (!RES) {
            copy_image<NT>(img, p.img[0]);
            copy_image<NT>(img + IMG, p.imgt[0]);
            __syncthreads();
        }
        lin_bwd_stage<D>(g, p.x2, row0, m, RES ? lds4 : img, RES ? lds4 + 10 * IMG : img + IMG, bl, true, tile, red,
                         prow + Row::L0, add, lane);
        store_d<D>(g, p.d_x2, row0, m, lane);
    }
}

// ====================================================================================================================
// Pre: x1 = SiLU(W1 x + b1) (optional) and NB projections P[:, k D:(k+1) D] = x1 Wp_k^T, Wp_k a [D, D] block with row
// stride ldp (column blocks of mlp_m / mlp_m_ji / mlp_m_kj / lin_rbf...).  Without the first layer (W1 null) this is the
// edge-side projection Q of the local layer on E_l rows.
// ====================================================================================================================
constexpr int NPB = 4;
struct NPreFwd {
    const float* x;
    const float4* img1;                     // packed image of W1 (null: no first layer)
    const float* b1;
    const float4* imgp[NPB];                // packed images of the projection blocks
    int nb;
    float* x1;                              // [m, D] (with W1)
    float* P;                               // [m, nb * D]
    int64_t m;
};

template <int D>
__global__ __launch_bounds__(NWG) void npre_fwd_kernel(const NPreFwd p) {
    constexpr int NT = D / 16;
    constexpr int IMG = NT * NT * 64;
    extern __shared__ float4 lds4[];
    float4* img1 = lds4;
    float4* imgp = lds4 + IMG;
    float* tile = reinterpret_cast<float*>(lds4 + (1 + NPB) * IMG) + (threadIdx.x >> 6) * 16 * (D + 4);
    const bool first = p.img1 != nullptr;
    if (first) copy_image<NT>(img1, p.img1);
    for (int k = 0; k < p.nb; ++k) copy_image<NT>(imgp + k * IMG, p.imgp[k]);
    __syncthreads();
    const int lane = threadIdx.x & 63, c = lane & 15;
    const int64_t m = p.m, ntiles = (m + 15) / 16;
    const int64_t ldP = (int64_t)p.nb * D;
    float bj[NT];
    load_bias<NT>(bj, first ? p.b1 : nullptr, c);
    for (int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); t < ntiles; t += (int64_t)gridDim.x * 4) {
        const int64_t row0 = t * 16;
        float4 a[NT];
        load_a<D>(a, p.x, row0, m, lane);
        if (first) {
            f32x4 v[NT];
            zero(v);
            mma_img<NT, NT>(v, a, img1, lane);
            add_bias_silu<NT>(v, bj);
            store_d<D>(v, p.x1, row0, m, lane);
            d_to_a<D>(a, v, tile, lane);
        }
#pragma unroll 1
        for (int k = 0; k < p.nb; ++k) {
            f32x4 y[NT];
            zero(y);
            mma_img<NT, NT>(y, a, imgp + k * IMG, lane);
            store_d<D>(y, p.P + k * D, row0, m, lane, ldP);
        }
    }
}

struct NPreBwd {
    const float* x;                          // [m, D] input of mlp_x1
    const float* x1;                         // [m, D] saved SiLU(W1 x + b1)
    const float4 *img1, *img1t;              // packed images of W1, plain and transposed
    const float* b1;
    const float4* imgpt[NPB];                // packed transposed images of the projection blocks
    int nb;
    const float* dP[NPB];                    // gradient of projection block k: [m, D] with row stride lddp[k]
    int lddp[NPB];
    const float* d_direct;                   // [m, D] gradient arriving at x1 directly (x2 = x1 + aggregate); nullable
    const float* d_add;                      // [m, D] added to dx (the residual branch of the layer input); nullable
    float* dx;                               // [m, D]
    float* partial;                          // row: [dWp_0 .. dWp_{nb-1} (D x D fragments each)][dW1][db1]
    int stride;
    int64_t m;
};

template <int D>
__global__ __launch_bounds__(64 * CHW) void npre_bwd_kernel(const NPreBwd p) {
    constexpr int NT = D / 16;
    constexpr int IMG = NT * NT * 64;
    constexpr int MAT = D * D;
    extern __shared__ float4 lds4[];
    float4* img1 = lds4;                                     // W1, W1^T, then Wp_k^T
    float4* img1t = lds4 + IMG;
    float4* imgpt = lds4 + 2 * IMG;
    float* tile = reinterpret_cast<float*>(lds4 + (2 + NPB) * IMG) + (threadIdx.x >> 6) * 16 * (D + 4);
    float* red = reinterpret_cast<float*>(lds4 + (2 + NPB) * IMG) + CHW * 16 * (D + 4);
    copy_image<NT>(img1, p.img1);
    copy_image<NT>(img1t, p.img1t);
    for (int k = 0; k < p.nb; ++k) copy_image<NT>(imgpt + k * IMG, p.imgpt[k]);
    __syncthreads();
    const int lane = threadIdx.x & 63, c = lane & 15, wave = threadIdx.x >> 6;
    const int64_t m = p.m, ntiles = (m + 15) / 16;
    float* prow = p.partial + (size_t)blockIdx.x * p.stride;
    for (int64_t grp = blockIdx.x; grp * CHW < ntiles; grp += gridDim.x) {
        const bool add = grp != (int64_t)blockIdx.x;
        const int64_t row0 = (grp * CHW + wave) * 16;
        f32x4 gx1[NT], x1d[NT];
        if (p.d_direct) load_d<D>(gx1, p.d_direct, row0, m, lane);
        else zero(gx1);
        load_d<D>(x1d, p.x1, row0, m, lane);
#pragma unroll 1
        for (int k = 0; k < p.nb; ++k) {
            f32x4 gk[NT], gw[NT][NT];
            load_d<D>(gk, p.dP[k], row0, m, lane, p.lddp[k]);
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) zero(gw[jt]);
            wgrad_acc<NT, NT>(gw, gk, x1d);
            float4 a[NT];
            d_to_a<D>(a, gk, tile, lane);
            mma_img<NT, NT>(gx1, a, imgpt + k * IMG, lane);
            wg_reduce<tail_resident(D)>(red, MAT, prow + k * MAT, add,
                                        [&](float* r, bool first) { red_add_mat<NT, NT>(r, gw, lane, first); });
        }
        // through mlp_x1
        float* dst = prow + p.nb * MAT;
        lin_bwd_stage<D>(gx1, p.x, row0, m, img1, img1t, p.b1, true, tile, red, dst, add, lane);
        if (p.d_add) {
            f32x4 ad[NT];
            load_d<D>(ad, p.d_add, row0, m, lane);
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) gx1[jt] += ad[jt];
        }
        store_d<D>(gx1, p.dx, row0, m, lane);
    }
}

// ====================================================================================================================
// Fixed-order sum of the workgroups' partial rows, scattered to up to 32 destinations in one launch.
//   rows > 0: a rows x KP matrix in fragment order (narrow_core.h) -> dst[o * ldo + k], k < kvalid
//   rows = 0: a plain vector of `len` floats                      -> dst[i]
// ====================================================================================================================
constexpr int MAXSEG = 32;
struct RSeg {
    float* dst;
    const float* src;     // the partial rows this destination is summed from: nblk rows of `stride` floats
    int nblk, stride;
    int off, rows, KP, kvalid, len, ldo;
};
struct RSegs {
    RSeg s[MAXSEG];
    int n;
};

__global__ __launch_bounds__(512) void narrow_reduce_multi_kernel(const RSegs S) {
    __shared__ float part[8][64];
    const RSeg& sg = S.s[blockIdx.y];
    const int total = sg.rows > 0 ? sg.rows * sg.KP : sg.len;
    if ((int)blockIdx.x * 64 >= total) return;
    const int x = threadIdx.x, y = threadIdx.y;
    const int p = blockIdx.x * 64 + x;
    const int nblk = sg.nblk, stride = sg.stride;
    const float* __restrict__ src = sg.src + sg.off;
    float s = 0.f;
    if (p < total) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;        // four loads in flight per thread, fixed association
        int b = y;
        for (; b + 24 < nblk; b += 32) {
            s0 += src[(size_t)b * stride + p];
            s1 += src[(size_t)(b + 8) * stride + p];
            s2 += src[(size_t)(b + 16) * stride + p];
            s3 += src[(size_t)(b + 24) * stride + p];
        }
        for (; b < nblk; b += 8) s0 += src[(size_t)b * stride + p];
        s = (s0 + s1) + (s2 + s3);
    }
    part[y][x] = s;
    __syncthreads();
    if (y != 0 || p >= total) return;
#pragma unroll
    for (int u = 1; u < 8; ++u) s += part[u][x];
    if (sg.rows > 0) {
        const int t = p >> 8, r = (p >> 6) & 3, lane = p & 63;
        const int nk = sg.KP / 16;
        const int jo = t / nk, jk = t % nk;
        const int o = 16 * jo + 4 * (lane >> 4) + r, k = 16 * jk + (lane & 15);
        if (k < sg.kvalid) sg.dst[(size_t)o * sg.ldo + k] = s;
    } else {
        sg.dst[p] = s;
    }
}

// host: segment table builder
struct SegTable {
    RSegs S;
    int gx = 1;
    const float* cur = nullptr;          // source of the entries added next (source())
    int cur_nblk = 0, cur_stride = 0;
    SegTable() { S.n = 0; }
    // the entries added from here on are summed from `partial`: nblk rows of `stride` floats
    void source(const float* partial, int nblk, int stride) { cur = partial, cur_nblk = nblk, cur_stride = stride; }
    void mat(float* dst, int off, int rows, int KP, int kvalid, int ldo) {
        RSeg& s = S.s[S.n++];
        s.dst = dst, s.off = off, s.rows = rows, s.KP = KP, s.kvalid = kvalid, s.len = 0, s.ldo = ldo;
        s.src = cur, s.nblk = cur_nblk, s.stride = cur_stride;
        const int g = (rows * KP + 63) / 64;
        gx = g > gx ? g : gx;
    }
    void vec(float* dst, int off, int len) {
        if (!dst) return;
        RSeg& s = S.s[S.n++];
        s.dst = dst, s.off = off, s.rows = 0, s.KP = 0, s.kvalid = 0, s.len = len, s.ldo = 0;
        s.src = cur, s.nblk = cur_nblk, s.stride = cur_stride;
        const int g = (len + 63) / 64;
        gx = g > gx ? g : gx;
    }
    int launch(hipStream_t st) {
        if (S.n == 0) return PAMNET_OK;
        hipLaunchKernelGGL(narrow_reduce_multi_kernel, dim3(gx, S.n), dim3(64, 8), 0, st, S);
        PAMNET_LAUNCH_CHECK();
        S.n = 0, gx = 1;
        return PAMNET_OK;
    }
};

// The partial rows of SEVERAL backward kernels wait in one arena and are reduced by one launch (a reduction is a ~5 us
// launch of a few dozen small workgroups; seven per layer pair were 35 us of a 1.1 ms RNA step).  take(): the region for a
// kernel's nblk x stride partial rows with `segs` destinations -- when the table (32 entries) or the arena cannot hold them
// the pending reductions are launched first (stream order: they run before the kernel that reuses the space).
struct Reducer {
    SegTable T;
    float* base;
    int64_t cap, used = 0;
    hipStream_t st;
    Reducer(float* b, int64_t c, hipStream_t s) : base(b), cap(c), st(s) {}
    int take(int nblk, int stride, int segs, float** out) {
        const int64_t need = ((int64_t)nblk * stride + 63) & ~(int64_t)63;
        if (need > cap || segs > MAXSEG) return PAMNET_EINVAL;
        if (T.S.n + segs > MAXSEG || used + need > cap) {
            const int rc = flush();
            if (rc) return rc;
        }
        *out = base + used;
        used += need;
        T.source(*out, nblk, stride);
        return PAMNET_OK;
    }
    int flush() {
        used = 0;
        return T.launch(st);
    }
};

// chain kernels: one 64-row workgroup per 4 tiles, at most CHAIN_CAP workgroups (later row groups add into the rows)
constexpr int CHAIN_CAP = 512;
inline int chain_grid(int64_t m) {
    const int64_t want = ((m + 15) / 16 + CHW - 1) / CHW;
    return (int)(want < 1 ? 1 : (want > CHAIN_CAP ? CHAIN_CAP : want));
}

template <int D>
constexpr size_t ntail_fwd_lds() { return (tail_resident(D) ? 10 : 2) * (size_t)D * D * 4 + CHW * 16 * (D + 4) * 4 + 10 * D * 4; }
template <int D>
constexpr size_t ntail_bwd_lds() {
    return (tail_resident(D) ? 20 : 4) * (size_t)D * D * 4 + CHW * 16 * (D + 4) * 4 +
           (tail_resident(D) ? CHW : 1) * ((size_t)D * D + 2 * D + 4) * 4 +      // (wg_reduce: a slot per wave when resident)
           10 * D * 4;                                                           // the bias vectors
}
template <int D>
constexpr size_t npre_fwd_lds() { return (1 + NPB) * (size_t)D * D * 4 + 4 * 16 * (D + 4) * 4; }
template <int D>
constexpr size_t npre_bwd_lds() {
    return (2 + NPB) * (size_t)D * D * 4 + CHW * 16 * (D + 4) * 4 + (tail_resident(D) ? CHW : 1) * ((size_t)D * D + D) * 4;
}

}  // namespace
