// Narrow-width (dim = 16 / 32 / 64) message kernels: the RNA configurations of the reference (inference_rna_puzzles.py:
// dim 16, n_layer 1; main_rna_puzzles.py: dim 64, n_layer 2) have a few thousand nodes but ~10^6 global edges and
// triplet/pair rows per batch, so the row-wise work is HBM-bound, not MFMA-bound as at dim = 128.
//
// Design (different from the 128-wide kernels on purpose):
//   * one wavefront owns a 16-row tile end to end; workgroups are just 4 independent waves that share the weight
//     images, there is no barrier inside the row loop;
//   * rows are read once as MFMA A fragments straight from HBM (lane l: row l&15, floats 16q + 4(l>>4) .. +3 = one
//     16-byte load per 16 columns), all GEMMs of a stage chain through registers, and the D->A relayout between two
//     GEMMs goes through a wave-private LDS tile;
//   * backward kernels recompute the forward pre-activations from the row tile instead of reading saved ones
//     (a [rows, D] store + load costs more than D/16 extra MFMA groups), and form the weight gradients in the same
//     pass: the D-layout accumulators of dZ are exactly the A operand of dW = dZ^T X with the row index as k;
//   * weight gradients are reduced wave -> workgroup (LDS, fixed order) -> grid (second kernel, fixed order):
//     deterministic, no atomics.
//
// Reference semantics: layers/global_message_passing.py:52-53 (message), layers/local_message_passing.py:48-49
// (mlp_sbf), models.py:185-188 (edge-embedding MLPs).
#pragma once
#include <stdlib.h>
#include "common.h"
#include "gemm_core.h"

namespace {

using pamnet::f32x4;
using pamnet::sigmoidf_fast;
using pamnet::sincos_turns;
using pamnet::Frag3;

constexpr int NWG = 256;                      // forward kernels: 4 independent waves per workgroup
// Backward kernels end with one partial gradient row per workgroup, so they run at most one workgroup per CU and get
// their occupancy from more waves per workgroup where the register budget allows (d = 64 needs ~350 registers a lane).
__host__ __device__ constexpr int bwd_waves(int d) { return d == 16 ? 16 : (d == 32 ? 8 : 4); }
// (two waves per SIMD for the single-layer backward at d = 64 fit in 256 registers but measured slower on node-sized
// inputs: 25.7 vs 23.3 us at 17.7 k rows, 230 vs 239 us at 669 k)
__host__ __device__ constexpr int lin_bwd_waves(int d) { return bwd_waves(d); }

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---- weight images in LDS ------------------------------------------------------------------------------------------
// img[(jt * NQ + q) * 64 + lane] = the float4 lane feeds to the four MFMAs of (output tile jt, k-group q).
//   TRANS = false: Y = X W^T, W [out][in] (row stride ld, `kin` valid input columns): b.t = W[16jt + c][16q + 4kg + t]
//   TRANS = true : Y = X W,   W [k][out]:                                            b.t = W[16q + 4kg + t][16jt + c]
template <int NJ, int NQ, bool TRANS>
__device__ __forceinline__ void build_image(float4* img, const float* __restrict__ W, int ld, int kin) {
    for (int idx = threadIdx.x; idx < NJ * NQ * 64; idx += blockDim.x) {
        const int lane = idx & 63, t = idx >> 6;
        const int q = t % NQ, jt = t / NQ;
        const int c = lane & 15, kg = lane >> 4;
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!TRANS) {
                const int k = 16 * q + 4 * kg + u;
                v[u] = (k < kin) ? W[(size_t)(16 * jt + c) * ld + k] : 0.f;
            } else {
                const int col = 16 * jt + c;
                v[u] = (col < kin) ? W[(size_t)(16 * q + 4 * kg + u) * ld + col] : 0.f;
            }
        }
        img[idx] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// consecutive MFMAs go to different accumulators (a dependent 16x16x4 pair issues 40 cycles apart, independent ones 32)
template <int NJ, int NQ>
__device__ __forceinline__ void mma_img(f32x4 (&acc)[NJ], const float4 (&a)[NQ], const float4* img, int lane) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        float4 b[NJ];
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt) b[jt] = img[(jt * NQ + q) * 64 + lane];
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt) acc[jt] = mfma4(a[q].x, b[jt].x, acc[jt]);
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt) acc[jt] = mfma4(a[q].y, b[jt].y, acc[jt]);
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt) acc[jt] = mfma4(a[q].z, b[jt].z, acc[jt]);
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt) acc[jt] = mfma4(a[q].w, b[jt].w, acc[jt]);
    }
}

// ---- the same GEMMs on the bf16 matrix pipe at fp32 accuracy (gemm_core.h "bf16x6"), K >= 32 ----------------------------
// A row tile is owned by ONE wave from load to store, so each element is split into its three bf16 pieces exactly once, in
// registers, where it is consumed.  k order of k-step s: the lane's two float4 A fragments 2s and 2s + 1, i.e.
// k(t) = 32 s + 16 (t >> 2) + 4 kg + (t & 3); the image holds the weights' pieces in the same order:
// imgb[((jt * NS + s) * 3 + piece) * 64 + lane], NS = NQ / 2 k-steps -- 1.5x the bytes of the fp32 image.
// Six v_mfma_f32_16x16x32_bf16 (~17 cycles each) per output tile and k-step replace eight fp32 MFMAs of 32.
template <int NJ, int NQ>
constexpr int img_units() { return NQ >= 2 ? NJ * (NQ / 2) * 3 * 64 : NJ * NQ * 64; }     // in 16-byte units

template <int NJ, int NQ, bool TRANS>
__device__ __forceinline__ void build_image_b(uint4* img, const float* __restrict__ W, int ld, int kin) {
    constexpr int NS = NQ / 2;
    for (int idx = threadIdx.x; idx < NJ * NS * 64; idx += blockDim.x) {
        const int lane = idx & 63, t = idx >> 6;
        const int s = t % NS, jt = t / NS;
        const int c = lane & 15, kg = lane >> 4;
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = 32 * s + 16 * (u >> 2) + 4 * kg + (u & 3);
            if (!TRANS) v[u] = (k < kin) ? W[(size_t)(16 * jt + c) * ld + k] : 0.f;
            else v[u] = (16 * jt + c < kin) ? W[(size_t)k * ld + 16 * jt + c] : 0.f;
        }
        const Frag3 f = pamnet::split_frag(v);
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
            img[((jt * NS + s) * 3 + pc) * 64 + lane] = make_uint4(f.p[pc][0], f.p[pc][1], f.p[pc][2], f.p[pc][3]);
    }
}

__device__ __forceinline__ f32x4 mfma_b(const uint32_t (&a)[4], const uint4& b, const f32x4& c) {
    const pamnet::u32x4 av = {a[0], a[1], a[2], a[3]}, bv = {b.x, b.y, b.z, b.w};
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pamnet::bf16x8, av), __builtin_bit_cast(pamnet::bf16x8, bv), c, 0, 0,
                                                   0);
}

template <int NJ, int NQ>
__device__ __forceinline__ void mma_img_b(f32x4 (&acc)[NJ], const float4 (&a)[NQ], const uint4* img, int lane) {
    constexpr int NS = NQ / 2;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const float v[8] = {a[2 * s].x, a[2 * s].y, a[2 * s].z, a[2 * s].w, a[2 * s + 1].x, a[2 * s + 1].y, a[2 * s + 1].z,
                            a[2 * s + 1].w};
        const Frag3 fa = pamnet::split_frag(v);
        // two output tiles at a time: their B fragments are 24 registers, and consecutive MFMAs alternate between the two
        // accumulators (smallest products first)
        constexpr int JB = NJ >= 2 ? 2 : 1;
#pragma unroll
        for (int j0 = 0; j0 < NJ; j0 += JB) {
            uint4 b[JB][3];
#pragma unroll
            for (int jj = 0; jj < JB; ++jj)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) b[jj][pc] = img[(((j0 + jj) * NS + s) * 3 + pc) * 64 + lane];
#pragma unroll
            for (int jj = 0; jj < JB; ++jj) acc[j0 + jj] = mfma_b(fa.p[2], b[jj][0], acc[j0 + jj]);
#pragma unroll
            for (int jj = 0; jj < JB; ++jj) acc[j0 + jj] = mfma_b(fa.p[1], b[jj][1], acc[j0 + jj]);
#pragma unroll
            for (int jj = 0; jj < JB; ++jj) acc[j0 + jj] = mfma_b(fa.p[0], b[jj][2], acc[j0 + jj]);
#pragma unroll
            for (int jj = 0; jj < JB; ++jj) acc[j0 + jj] = mfma_b(fa.p[1], b[jj][0], acc[j0 + jj]);
#pragma unroll
            for (int jj = 0; jj < JB; ++jj) acc[j0 + jj] = mfma_b(fa.p[0], b[jj][1], acc[j0 + jj]);
#pragma unroll
            for (int jj = 0; jj < JB; ++jj) acc[j0 + jj] = mfma_b(fa.p[0], b[jj][0], acc[j0 + jj]);
        }
    }
}

// width dispatch: bf16x6 from K = 32 on, the fp32 MFMA for K = 16 (half a bf16 k-step)
template <int NJ, int NQ, bool TRANS>
__device__ __forceinline__ void build_w(float4* img, const float* __restrict__ W, int ld, int kin) {
    if constexpr (NQ >= 2) build_image_b<NJ, NQ, TRANS>(reinterpret_cast<uint4*>(img), W, ld, kin);
    else build_image<NJ, NQ, TRANS>(img, W, ld, kin);
}
template <int NJ, int NQ>
__device__ __forceinline__ void mma_w(f32x4 (&acc)[NJ], const float4 (&a)[NQ], const float4* img, int lane) {
    if constexpr (NQ >= 2) mma_img_b<NJ, NQ>(acc, a, reinterpret_cast<const uint4*>(img), lane);
    else mma_img<NJ, NQ>(acc, a, img, lane);
}
// bytes of one [d, d] weight image as the row kernels below lay it out
__host__ __device__ constexpr size_t wimg_bytes(int d) { return (size_t)(d >= 32 ? 6 : 4) * d * d; }

template <int N>
__device__ __forceinline__ void zero(f32x4 (&acc)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// Row loads never sit under a branch: out-of-range rows read row m - 1 (a valid address, m > 0) and are zeroed by a
// select afterwards.  (`ok ? *p : 0` made the compiler wrap every element load in its own exec-masked block with a
// vmcnt(0) wait and whole-fragment register copies: 3x slower kernels.)
// rows as A fragments (row stride D floats, 16-byte aligned); rows >= m read as zero
template <int D>
__device__ __forceinline__ void load_a(float4 (&a)[D / 16], const float* __restrict__ X, int64_t row0, int64_t m, int lane) {
    const int64_t row = row0 + (lane & 15);
    const bool ok = row < m;
    const float* p = X + (ok ? row : m - 1) * D + 4 * (lane >> 4);
#pragma unroll
    for (int q = 0; q < D / 16; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(p + 16 * q);
        a[q] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
    }
}

// rows in accumulator ("D") layout: v[jt][r] = X[row0 + 4kg + r][16jt + c]; rows >= m read as zero
template <int D>
__device__ __forceinline__ void load_d(f32x4 (&v)[D / 16], const float* __restrict__ X, int64_t row0, int64_t m, int lane,
                                       int64_t ld = D) {
    const int c = lane & 15, kg = lane >> 4;
    float t[4][D / 16];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t row = row0 + 4 * kg + r;
        const float* p = X + (row < m ? row : m - 1) * ld + c;
#pragma unroll
        for (int jt = 0; jt < D / 16; ++jt) t[r][jt] = p[16 * jt];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const bool ok = row0 + 4 * kg + r < m;
#pragma unroll
        for (int jt = 0; jt < D / 16; ++jt) v[jt][r] = ok ? t[r][jt] : 0.f;
    }
}

template <int D>
__device__ __forceinline__ void store_d(const f32x4 (&v)[D / 16], float* __restrict__ Y, int64_t row0, int64_t m, int lane,
                                        int64_t ld = D) {
    const int c = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t row = row0 + 4 * kg + r;
        if (row < m) {
            float* p = Y + row * ld + c;
#pragma unroll
            for (int jt = 0; jt < D / 16; ++jt) p[16 * jt] = v[jt][r];
        }
    }
}

// accumulator layout -> A fragments through the wave's private LDS tile ([16][D + 4] floats)
template <int D>
__device__ __forceinline__ void d_to_a(float4 (&a)[D / 16], const f32x4 (&v)[D / 16], float* tile, int lane) {
    constexpr int LD = D + 4;
    const int c = lane & 15, kg = lane >> 4;
    wave_lds_sync();
#pragma unroll
    for (int jt = 0; jt < D / 16; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) tile[(4 * kg + r) * LD + 16 * jt + c] = v[jt][r];
    wave_lds_sync();
#pragma unroll
    for (int q = 0; q < D / 16; ++q) a[q] = *reinterpret_cast<const float4*>(tile + c * LD + 16 * q + 4 * kg);
}

// A fragments -> accumulator layout through the wave's private LDS tile (the reverse of d_to_a): lets a row tile that
// was fetched with four coalesced 16-byte loads per lane also serve as the weight-gradient operand, instead of
// sixteen more 4-byte loads per lane
template <int D>
__device__ __forceinline__ void a_to_d(f32x4 (&v)[D / 16], const float4 (&a)[D / 16], float* tile, int lane) {
    constexpr int LD = D + 4;
    const int c = lane & 15, kg = lane >> 4;
    wave_lds_sync();
#pragma unroll
    for (int q = 0; q < D / 16; ++q) *reinterpret_cast<float4*>(tile + c * LD + 16 * q + 4 * kg) = a[q];
    wave_lds_sync();
#pragma unroll
    for (int jt = 0; jt < D / 16; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[jt][r] = tile[(4 * kg + r) * LD + 16 * jt + c];
}

// dW[o][k] += sum_rows dz[row][o] * x[row][k], both operands in accumulator layout (rows are the MFMA k index)
template <int NJ, int NK>
__device__ __forceinline__ void wgrad_acc(f32x4 (&w)[NJ][NK], const f32x4 (&dz)[NJ], const f32x4 (&x)[NK]) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int jo = 0; jo < NJ; ++jo)
#pragma unroll
            for (int jk = 0; jk < NK; ++jk) w[jo][jk] = mfma4(dz[jo][r], x[jk][r], w[jo][jk]);
}

// column sums of an accumulator-layout tile over the wave's rows: result valid on every lane for column 16jt + c
template <int NJ>
__device__ __forceinline__ void colsum_acc(float (&s)[NJ], const f32x4 (&dz)[NJ]) {
#pragma unroll
    for (int jt = 0; jt < NJ; ++jt) s[jt] += (dz[jt][0] + dz[jt][1]) + (dz[jt][2] + dz[jt][3]);
}

// ---- workgroup reduction of per-wave gradient accumulators -----------------------------------------------------------
// `red` (LDS, reused image area) is laid out [matrix fragments ...][bias columns ...]; waves add in wave order.
template <int NJ, int NK>
__device__ __forceinline__ void red_add_mat(float* red, const f32x4 (&w)[NJ][NK], int lane, bool first) {
#pragma unroll
    for (int jo = 0; jo < NJ; ++jo)
#pragma unroll
        for (int jk = 0; jk < NK; ++jk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float* p = red + ((jo * NK + jk) * 4 + r) * 64 + lane;
                *p = first ? w[jo][jk][r] : (*p + w[jo][jk][r]);
            }
}

template <int NJ>
__device__ __forceinline__ void red_add_bias(float* red, float (&s)[NJ], int lane, bool first) {
#pragma unroll
    for (int jt = 0; jt < NJ; ++jt) {
        float v = s[jt];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (lane < 16) red[16 * jt + lane] = first ? v : (red[16 * jt + lane] + v);
    }
}

// Workgroup sum of the waves' parts of a partial row (n floats) in LDS.  The waves used to take turns on ONE copy -- NW read-
// modify-write rounds, a barrier each: 16 at d = 16.  Now S copies (as many as the kernel's LDS layout, `avail` floats, holds;
// a power of two dividing NW): wave w owns copy w % S and adds in turn w / S, then the threads add the S copies of an element
// in order.  S = NW is the old order bit for bit; fewer copies change the association (still fixed, run to run).
__host__ __device__ constexpr int pick_slots(int nw, int n, int avail) {
    int s = nw;
    while (s > 1 && s * n > avail) s >>= 1;
    return s;
}
template <int NW, int S, typename F>
__device__ __forceinline__ void wg_sum_out(float* red, int n, float* __restrict__ dst, F&& put) {
    const int wave = threadIdx.x >> 6;
    float* mine = red + (wave % S) * n;
#pragma unroll 1
    for (int t = 0; t < NW / S; ++t) {
        if (wave / S == t) put(mine, t == 0);
        __syncthreads();
    }
    for (int i = threadIdx.x; i < n; i += 64 * NW) {
        float v = red[i];
#pragma unroll
        for (int k = 1; k < S; ++k) v += red[k * n + i];
        dst[i] = v;
    }
}

// out[mat][o][k] (k < kvalid) = sum_b partial[b][fragment(o, k)];  bias[j] = sum_b partial[b][bias_off + j].
// Block (64, 8): 64 consecutive positions of the partial row (coalesced) x 8 slices over the workgroup rows, combined
// through LDS in slice order -- fixed summation order.
__global__ __launch_bounds__(512) void narrow_reduce_kernel(const float* __restrict__ partial, int nblk, int stride,
                                                            int nmat, int D, int KP, int kvalid, int nbias,
                                                            float* __restrict__ mats, float* __restrict__ bias) {
    __shared__ float part[8][64];
    const int x = threadIdx.x, y = threadIdx.y;
    const int p = blockIdx.x * 64 + x;
    const int per = D * KP;
    const int total = nmat * per + nbias;
    float s = 0.f;
    if (p < total) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;        // four loads in flight per thread, fixed association
        int b = y;
        for (; b + 24 < nblk; b += 32) {
            s0 += partial[(size_t)b * stride + p];
            s1 += partial[(size_t)(b + 8) * stride + p];
            s2 += partial[(size_t)(b + 16) * stride + p];
            s3 += partial[(size_t)(b + 24) * stride + p];
        }
        for (; b < nblk; b += 8) s0 += partial[(size_t)b * stride + p];
        s = (s0 + s1) + (s2 + s3);
    }
    part[y][x] = s;
    __syncthreads();
    if (y != 0 || p >= total) return;
#pragma unroll
    for (int u = 1; u < 8; ++u) s += part[u][x];
    if (p < nmat * per) {
        const int mat = p / per, rem = p % per;
        const int t = rem >> 8, r = (rem >> 6) & 3, lane = rem & 63;
        const int nk = KP / 16;
        const int jo = t / nk, jk = t % nk;
        const int o = 16 * jo + 4 * (lane >> 4) + r, k = 16 * jk + (lane & 15);
        if (k < kvalid) mats[(size_t)mat * D * kvalid + (size_t)o * kvalid + k] = s;
    } else {
        bias[p - nmat * per] = s;
    }
}

// ====================================================================================================================
// Global message (layers/global_message_passing.py:52-53 with W_m split into node and edge blocks):
//   z = P[tgt, :D] + P[src, D:] + e We^T + b ;  msg = SiLU(z) * (e Wea^T)
// ====================================================================================================================
// d = 64: capped at 256 registers so that the two workgroups a CU is given are co-resident (the unconstrained build keeps
// both weight images in registers, 272 per lane, one wave per SIMD: 321 us against 252 us at 867 k edges)
#ifndef NGF_CAP
#define NGF_CAP 2
#endif
template <int D>
__global__ __launch_bounds__(NWG, D == 64 ? NGF_CAP : 1) void nglobal_fwd_kernel(const float* __restrict__ e, int64_t m,
                                                          const int32_t* __restrict__ tgt, const int32_t* __restrict__ src,
                                                          const float* __restrict__ P, const float* __restrict__ We, int ldwe,
                                                          const float* __restrict__ bias, const float* __restrict__ Wea,
                                                          int ldwea, float* __restrict__ msg) {
    constexpr int NT = D / 16;
    extern __shared__ float4 lds4[];
    float4* img_e = lds4;
    float4* img_a = lds4 + img_units<NT, NT>();
    build_w<NT, NT, false>(img_e, We, ldwe, D);
    build_w<NT, NT, false>(img_a, Wea, ldwea, D);
    __syncthreads();
    const int lane = threadIdx.x & 63, c = lane & 15, kg = lane >> 4;
    const int64_t ntiles = (m + 15) / 16;
    float bj[NT];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) bj[jt] = bias[16 * jt + c];
    for (int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); tile < ntiles; tile += (int64_t)gridDim.x * 4) {
        // the weight images are loop invariant: without this the compiler keeps their fragments in registers across
        // tiles (192 at d = 64) and spills to fit the two-workgroups-per-CU cap; LDS reads are cheap, registers are not
        asm volatile("" ::: "memory");
        const int64_t row0 = tile * 16;
        float4 a[NT];
        load_a<D>(a, e, row0, m, lane);
        // all index and projection loads of the tile are issued before the GEMMs (their latency hides behind the MFMAs)
        int ti[4], sj[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = row0 + 4 * kg + r;
            ti[r] = tgt[row < m ? row : m - 1];
            sj[r] = src[row < m ? row : m - 1];
        }
        float pv[4][NT], qv[4][NT];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* pi = P + (size_t)ti[r] * (2 * D) + c;
            const float* pj = P + (size_t)sj[r] * (2 * D) + D + c;
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                pv[r][jt] = pi[16 * jt];
                qv[r][jt] = pj[16 * jt];
            }
        }
        f32x4 q1[NT], q2[NT];
        zero(q1);
        zero(q2);
        mma_w<NT, NT>(q1, a, img_e, lane);
        mma_w<NT, NT>(q2, a, img_a, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = row0 + 4 * kg + r;
            if (row < m) {
                float* out = msg + row * D + c;
#pragma unroll
                for (int jt = 0; jt < NT; ++jt) {
                    const float z = q1[jt][r] + bj[jt] + pv[r][jt] + qv[r][jt];
                    out[16 * jt] = z * sigmoidf_fast(z) * q2[jt][r];
                }
            }
        }
    }
}

// backward: dmsg[row] = dagg[tgt[row]].  Outputs dz [m, D] (for the two node-side segment sums), de [m, D],
// and per-workgroup partials of dWe, dWea (fragment order) and db.
template <int D>
__global__ __launch_bounds__(64 * bwd_waves(D)) void nglobal_bwd_kernel(const float* __restrict__ e, int64_t m,
                                                          const int32_t* __restrict__ tgt, const int32_t* __restrict__ src,
                                                          const float* __restrict__ P, const float* __restrict__ We, int ldwe,
                                                          const float* __restrict__ bias, const float* __restrict__ Wea,
                                                          int ldwea, const float* __restrict__ dagg,
                                                          float* __restrict__ dz_out, float* __restrict__ de,
                                                          float* __restrict__ partial, int stride, int acc_de) {
    constexpr int NT = D / 16;
    constexpr int IMG = img_units<NT, NT>();
    constexpr int NW = bwd_waves(D);
    extern __shared__ float4 lds4[];
    float4* img_e = lds4;
    float4* img_a = lds4 + IMG;
    float4* img_et = lds4 + 2 * IMG;
    float4* img_at = lds4 + 3 * IMG;
    float* tile = reinterpret_cast<float*>(lds4 + 4 * IMG) + (threadIdx.x >> 6) * 16 * (D + 4);
    build_w<NT, NT, false>(img_e, We, ldwe, D);
    build_w<NT, NT, false>(img_a, Wea, ldwea, D);
    build_w<NT, NT, true>(img_et, We, ldwe, D);
    build_w<NT, NT, true>(img_at, Wea, ldwea, D);
    __syncthreads();
    const int lane = threadIdx.x & 63, c = lane & 15, kg = lane >> 4;
    const int64_t ntiles = (m + 15) / 16;
    float bj[NT], dbs[NT];
    f32x4 gwe[NT][NT], gwa[NT][NT];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
        bj[jt] = bias[16 * jt + c];
        dbs[jt] = 0.f;
        zero(gwe[jt]);
        zero(gwa[jt]);
    }
    // d <= 32: the rows, node indices and accumulate operand of a tile are fetched a tile ahead (12 / 24 registers), so a tile
    // begins with its gathers instead of an index round trip ahead of them (four waves per SIMD hide little of it).  d = 64 has
    // no registers for it (measured there: 496 -> 550 us at 867 k edges).
    constexpr bool PF = D <= 32;
    struct Raw {
        int ti[4], sj[4];
        float4 a[NT];
        f32x4 de[NT];
    };
    auto fetch = [&](int64_t t, Raw& w) {
        const int64_t r0 = t * 16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = r0 + 4 * kg + r;
            w.ti[r] = tgt[row < m ? row : m - 1];
            w.sj[r] = src[row < m ? row : m - 1];
        }
        load_a<D>(w.a, e, r0, m, lane);
        if (PF && acc_de) load_d<D>(w.de, de, r0, m, lane);
    };
    const int64_t tstep = (int64_t)gridDim.x * NW;
    int64_t tile_id = (int64_t)blockIdx.x * NW + (threadIdx.x >> 6);
    Raw cur;
    if (PF && tile_id < ntiles) fetch(tile_id, cur);
    for (; tile_id < ntiles; tile_id += tstep) {
        const int64_t row0 = tile_id * 16;
        if constexpr (!PF) fetch(tile_id, cur);
        float4 (&a)[NT] = cur.a;
        int (&ti)[4] = cur.ti;
        int (&sj)[4] = cur.sj;
        // all gathers of the tile are issued before the GEMMs
        float pv[4][NT], qv[4][NT], gv[4][NT];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* pi = P + (size_t)ti[r] * (2 * D) + c;
            const float* pj = P + (size_t)sj[r] * (2 * D) + D + c;
            const float* dg = dagg + (size_t)ti[r] * D + c;
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                pv[r][jt] = pi[16 * jt];
                qv[r][jt] = pj[16 * jt];
                gv[r][jt] = dg[16 * jt];
            }
        }
        Raw nxt;
        if constexpr (PF) fetch(tile_id + tstep < ntiles ? tile_id + tstep : tile_id, nxt);   // (the last tile again: unused)
        f32x4 q1[NT], q2[NT], ed[NT];
        // (the accumulator-layout copy of e through the wave's LDS tile replaces a second, 4-byte-strided read of the rows)
        a_to_d<D>(ed, a, tile, lane);
        zero(q1);
        zero(q2);
        mma_w<NT, NT>(q1, a, img_e, lane);
        mma_w<NT, NT>(q2, a, img_a, lane);
        // q1 -> dz, q2 -> dq2 (in place)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool ok = row0 + 4 * kg + r < m;
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                const float z = q1[jt][r] + bj[jt] + pv[r][jt] + qv[r][jt];
                const float s = sigmoidf_fast(z);
                const float dm = ok ? gv[r][jt] : 0.f;
                const float gate = q2[jt][r];
                q1[jt][r] = dm * gate * (s * (1.0f + z * (1.0f - s)));
                q2[jt][r] = dm * (z * s);
            }
        }
        store_d<D>(q1, dz_out, row0, m, lane);
        wgrad_acc<NT, NT>(gwe, q1, ed);
        wgrad_acc<NT, NT>(gwa, q2, ed);
        colsum_acc<NT>(dbs, q1);
        f32x4 dx[NT];
        zero(dx);
        d_to_a<D>(a, q1, tile, lane);
        mma_w<NT, NT>(dx, a, img_et, lane);
        d_to_a<D>(a, q2, tile, lane);
        mma_w<NT, NT>(dx, a, img_at, lane);
        if (acc_de) {                                        // the edge embedding feeds every layer: later calls add
            if constexpr (!PF) load_d<D>(cur.de, de, row0, m, lane);
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) dx[jt] += cur.de[jt];
        }
        store_d<D>(dx, de, row0, m, lane);
        if constexpr (PF) cur = nxt;
    }
    // workgroup reduction in wave order, then one partial row per workgroup
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds4);
    constexpr int MAT = D * D, N = 2 * MAT + D;
    constexpr int S = pick_slots(NW, N, 4 * IMG * 4 + NW * 16 * (D + 4));
    wg_sum_out<NW, S>(red, N, partial + (size_t)blockIdx.x * stride, [&](float* r, bool first) {
        red_add_mat<NT, NT>(r, gwe, lane, first);
        red_add_mat<NT, NT>(r + MAT, gwa, lane, first);
        red_add_bias<NT>(r + 2 * MAT, dbs, lane, first);
    });
}

// ====================================================================================================================
// Two-layer SiLU MLP on rows (mlp_sbf of the local layer, layers/local_message_passing.py:24,49):
//   y = SiLU(W2 SiLU(W1 x + b1) + b2)
// ====================================================================================================================
template <int D>
__global__ __launch_bounds__(NWG) void nmlp2_fwd_kernel(const float* __restrict__ x, int64_t m,
                                                        const float* __restrict__ W1, const float* __restrict__ b1,
                                                        const float* __restrict__ W2, const float* __restrict__ b2,
                                                        int res_x, const float* __restrict__ res,
                                                        float* __restrict__ y) {
    constexpr int NT = D / 16;
    constexpr int IMG = img_units<NT, NT>();
    extern __shared__ float4 lds4[];
    float4* img1 = lds4;
    float4* img2 = lds4 + IMG;
    float* tile = reinterpret_cast<float*>(lds4 + 2 * IMG) + (threadIdx.x >> 6) * 16 * (D + 4);
    build_w<NT, NT, false>(img1, W1, D, D);
    build_w<NT, NT, false>(img2, W2, D, D);
    __syncthreads();
    const int lane = threadIdx.x & 63, c = lane & 15;
    const int64_t ntiles = (m + 15) / 16;
    float bj1[NT], bj2[NT];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
        bj1[jt] = b1[16 * jt + c];
        bj2[jt] = b2[16 * jt + c];
    }
    for (int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); t < ntiles; t += (int64_t)gridDim.x * 4) {
        const int64_t row0 = t * 16;
        float4 a[NT];
        load_a<D>(a, x, row0, m, lane);
        f32x4 h[NT], o[NT];
        zero(h);
        mma_w<NT, NT>(h, a, img1, lane);
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float z = h[jt][r] + bj1[jt];
                h[jt][r] = z * sigmoidf_fast(z);
            }
        d_to_a<D>(a, h, tile, lane);
        zero(o);
        mma_w<NT, NT>(o, a, img2, lane);
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float z = o[jt][r] + bj2[jt];
                o[jt][r] = z * sigmoidf_fast(z);
            }
        if (res_x) {                                         // Res block: MLP2(x) + x (layers/basic.py:32-33)
            load_d<D>(h, x, row0, m, lane);
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) o[jt] += h[jt];
        }
        if (res) {
            load_d<D>(h, res, row0, m, lane);
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) o[jt] += h[jt];
        }
        store_d<D>(o, y, row0, m, lane);
    }
}

template <int D>
__global__ __launch_bounds__(64 * bwd_waves(D)) void nmlp2_bwd_kernel(const float* __restrict__ x, int64_t m,
                                                        const float* __restrict__ W1, const float* __restrict__ b1,
                                                        const float* __restrict__ W2, const float* __restrict__ b2,
                                                        const float* __restrict__ dy, int res_x,
                                                        float* __restrict__ dx, float* __restrict__ partial,
                                                        int stride, int acc_dx) {
    constexpr int NT = D / 16;
    constexpr int IMG = img_units<NT, NT>();
    constexpr int NW = bwd_waves(D);
    extern __shared__ float4 lds4[];
    float4* img1 = lds4;
    float4* img2 = lds4 + IMG;
    float4* img1t = lds4 + 2 * IMG;
    float4* img2t = lds4 + 3 * IMG;
    float* tile = reinterpret_cast<float*>(lds4 + 4 * IMG) + (threadIdx.x >> 6) * 16 * (D + 4);
    build_w<NT, NT, false>(img1, W1, D, D);
    build_w<NT, NT, false>(img2, W2, D, D);
    build_w<NT, NT, true>(img1t, W1, D, D);
    build_w<NT, NT, true>(img2t, W2, D, D);
    __syncthreads();
    const int lane = threadIdx.x & 63, c = lane & 15;
    const int64_t ntiles = (m + 15) / 16;
    float bj1[NT], bj2[NT], db1[NT], db2[NT];
    f32x4 gw1[NT][NT], gw2[NT][NT];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
        bj1[jt] = b1[16 * jt + c];
        bj2[jt] = b2[16 * jt + c];
        db1[jt] = db2[jt] = 0.f;
        zero(gw1[jt]);
        zero(gw2[jt]);
    }
    // Row tiles arrive as A fragments (four coalesced 16-byte loads per lane for x and for dy; the accumulator-layout copies
    // the weight-gradient GEMMs need go through the wave's LDS tile), and the NEXT tile's are requested before this tile's
    // GEMMs: at one wave per SIMD nothing else hides the HBM latency (415 -> 3xx us at 669 k rows, d = 64).
    const int64_t tstep = (int64_t)gridDim.x * NW;
    int64_t t = (int64_t)blockIdx.x * NW + (threadIdx.x >> 6);
    float4 a[NT], ga[NT], a_n[NT], ga_n[NT];
    if (t < ntiles) {
        load_a<D>(a, x, t * 16, m, lane);
        load_a<D>(ga, dy, t * 16, m, lane);
    }
    for (; t < ntiles; t += tstep) {
        const int64_t row0 = t * 16;
        const bool more = t + tstep < ntiles;
        if (more) {
            load_a<D>(a_n, x, (t + tstep) * 16, m, lane);
            load_a<D>(ga_n, dy, (t + tstep) * 16, m, lane);
        }
        f32x4 z1[NT], h[NT], g[NT], xd[NT], dyv[NT];
        a_to_d<D>(xd, a, tile, lane);
        a_to_d<D>(dyv, ga, tile, lane);
        zero(z1);
        mma_w<NT, NT>(z1, a, img1, lane);
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                z1[jt][r] += bj1[jt];
                h[jt][r] = z1[jt][r] * sigmoidf_fast(z1[jt][r]);
            }
        d_to_a<D>(a, h, tile, lane);
        zero(g);
        mma_w<NT, NT>(g, a, img2, lane);                   // z2 - b2
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float z = g[jt][r] + bj2[jt];
                const float s = sigmoidf_fast(z);
                g[jt][r] = dyv[jt][r] * (s * (1.0f + z * (1.0f - s)));      // dz2 (zero on padded rows: dy = 0)
            }
        wgrad_acc<NT, NT>(gw2, g, h);
        colsum_acc<NT>(db2, g);
        d_to_a<D>(a, g, tile, lane);
        zero(g);
        mma_w<NT, NT>(g, a, img2t, lane);                  // dh1
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float z = z1[jt][r];
                const float s = sigmoidf_fast(z);
                g[jt][r] *= s * (1.0f + z * (1.0f - s));     // dz1
            }
        wgrad_acc<NT, NT>(gw1, g, xd);
        colsum_acc<NT>(db1, g);
        if (dx) {
            d_to_a<D>(a, g, tile, lane);
            zero(g);
            mma_w<NT, NT>(g, a, img1t, lane);
            if (res_x) {
#pragma unroll
                for (int jt = 0; jt < NT; ++jt) g[jt] += dyv[jt];
            }
            if (acc_dx) {
                load_d<D>(dyv, dx, row0, m, lane);
#pragma unroll
                for (int jt = 0; jt < NT; ++jt) g[jt] += dyv[jt];
            }
            store_d<D>(g, dx, row0, m, lane);
        }
        if (more) {
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                a[jt] = a_n[jt];
                ga[jt] = ga_n[jt];
            }
        }
    }
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds4);
    constexpr int MAT = D * D, N = 2 * MAT + 2 * D;
    constexpr int S = pick_slots(NW, N, 4 * IMG * 4 + NW * 16 * (D + 4));
    wg_sum_out<NW, S>(red, N, partial + (size_t)blockIdx.x * stride, [&](float* r, bool first) {
        red_add_mat<NT, NT>(r, gw1, lane, first);
        red_add_mat<NT, NT>(r + MAT, gw2, lane, first);
        red_add_bias<NT>(r + 2 * MAT, db1, lane, first);
        red_add_bias<NT>(r + 2 * MAT + D, db2, lane, first);
    });
}

// ====================================================================================================================
// One dense layer on rows: y = act(x W^T + b), W a [D, D] block with row stride ldw (slices of the 3d-wide message
// weights included), y with row stride ldy (so several blocks can fill one [rows, n D] projection).
// ====================================================================================================================
template <int D>
__global__ __launch_bounds__(NWG) void nlinear_fwd_kernel(const float* __restrict__ x, int64_t m,
                                                          const float* __restrict__ W, int ldw,
                                                          const float* __restrict__ b, int act, float* __restrict__ y,
                                                          int64_t ldy) {
    constexpr int NT = D / 16;
    extern __shared__ float4 lds4[];
    build_w<NT, NT, false>(lds4, W, ldw, D);
    __syncthreads();
    const int lane = threadIdx.x & 63, c = lane & 15;
    const int64_t ntiles = (m + 15) / 16;
    float bj[NT];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) bj[jt] = b ? b[16 * jt + c] : 0.f;
    for (int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); t < ntiles; t += (int64_t)gridDim.x * 4) {
        const int64_t row0 = t * 16;
        float4 a[NT];
        load_a<D>(a, x, row0, m, lane);
        f32x4 o[NT];
        zero(o);
        mma_w<NT, NT>(o, a, lds4, lane);
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float z = o[jt][r] + bj[jt];
                o[jt][r] = act ? z * sigmoidf_fast(z) : z;
            }
        store_d<D>(o, y, row0, m, lane, ldy);
    }
}

// dx (+)= dz W;  partial = [dW fragments (D x D)][db (D)]
template <int D>
__global__ __launch_bounds__(64 * lin_bwd_waves(D)) void nlinear_bwd_kernel(const float* __restrict__ x, int64_t m,
                                                          const float* __restrict__ W, int ldw,
                                                          const float* __restrict__ b, int act,
                                                          const float* __restrict__ dy, int64_t lddy,
                                                          float* __restrict__ dx, int accumulate,
                                                          float* __restrict__ partial, int stride) {
    constexpr int NT = D / 16;
    constexpr int IMG = img_units<NT, NT>();
    constexpr int NW = lin_bwd_waves(D);
    extern __shared__ float4 lds4[];
    float4* img = lds4;
    float4* imgt = lds4 + IMG;
    float* tile = reinterpret_cast<float*>(lds4 + 2 * IMG) + (threadIdx.x >> 6) * 16 * (D + 4);
    build_w<NT, NT, false>(img, W, ldw, D);
    build_w<NT, NT, true>(imgt, W, ldw, D);
    __syncthreads();
    const int lane = threadIdx.x & 63, c = lane & 15;
    const int64_t ntiles = (m + 15) / 16;
    float bj[NT], dbs[NT];
    f32x4 gw[NT][NT];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
        bj[jt] = b ? b[16 * jt + c] : 0.f;
        dbs[jt] = 0.f;
        zero(gw[jt]);
    }
    // the next tile's rows are requested before this tile's GEMMs: at d = 64 a wave holds ~350 registers, one wave per
    // SIMD, so nothing else hides the HBM latency of its loads
    const int64_t tstep = (int64_t)gridDim.x * NW;
    int64_t t = (int64_t)blockIdx.x * NW + (threadIdx.x >> 6);
    float4 a[NT], ga[NT], a_n[NT], ga_n[NT];                // x and dy row tiles as A fragments (coalesced 16-byte loads)
    f32x4 g[NT], xd[NT];
    const bool dense_dy = lddy == D;
    if (t < ntiles) {
        load_a<D>(a, x, t * 16, m, lane);
        if (dense_dy) load_a<D>(ga, dy, t * 16, m, lane);
    }
    for (; t < ntiles; t += tstep) {
        const int64_t row0 = t * 16;
        const bool more = t + tstep < ntiles;
        if (more) {
            load_a<D>(a_n, x, (t + tstep) * 16, m, lane);
            if (dense_dy) load_a<D>(ga_n, dy, (t + tstep) * 16, m, lane);
        }
        if (dense_dy) a_to_d<D>(g, ga, tile, lane);
        else load_d<D>(g, dy, row0, m, lane, lddy);          // a column block of a wider gradient
        a_to_d<D>(xd, a, tile, lane);
        if (act) {
            f32x4 z[NT];
            zero(z);
            mma_w<NT, NT>(z, a, img, lane);
#pragma unroll
            for (int jt = 0; jt < NT; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float zz = z[jt][r] + bj[jt];
                    const float s = sigmoidf_fast(zz);
                    g[jt][r] *= s * (1.0f + zz * (1.0f - s));
                }
        }
        wgrad_acc<NT, NT>(gw, g, xd);
        colsum_acc<NT>(dbs, g);
        if (dx) {
            d_to_a<D>(a, g, tile, lane);
            zero(g);
            mma_w<NT, NT>(g, a, imgt, lane);
            if (accumulate) {
                load_d<D>(xd, dx, row0, m, lane);
#pragma unroll
                for (int jt = 0; jt < NT; ++jt) g[jt] += xd[jt];
            }
            store_d<D>(g, dx, row0, m, lane);
        }
        if (more) {
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                a[jt] = a_n[jt];
                ga[jt] = ga_n[jt];
            }
        }
    }
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds4);
    constexpr int MAT = D * D, N = MAT + D;
    constexpr int S = pick_slots(NW, N, 2 * IMG * 4 + NW * 16 * (D + 4));
    wg_sum_out<NW, S>(red, N, partial + (size_t)blockIdx.x * stride, [&](float* r, bool first) {
        red_add_mat<NT, NT>(r, gw, lane, first);
        red_add_bias<NT>(r + MAT, dbs, lane, first);
    });
}

// The four bias-free projection blocks of the edge-side Q = rbf [W_0 | W_1 | W_2 | W_3]^T in ONE pass over the rows
// (d <= 32: four D x D accumulator sets fit the registers): per 16-row tile x is read once, d rbf (+)= sum_k dQ_k W_k is
// written once, dW_k = dQ_k^T x and the column sums of dQ_k (the bias gradients of the layers those blocks feed) accumulate
// in registers.  partial row of a workgroup: [dW_0][db_0] ... [dW_3][db_3].  (Four nlinear_bwd launches re-read x and
// read-modify-write d rbf four times; at d = 16 they are ~10 us each, launch-bound.)
template <int D>
__global__ __launch_bounds__(64 * lin_bwd_waves(D)) void nqblock4_bwd_kernel(const float* __restrict__ x, int64_t m,
                                                          const float* __restrict__ W0, const float* __restrict__ W1,
                                                          const float* __restrict__ W2, const float* __restrict__ W3,
                                                          int ld0, int ld1, int ld2, int ld3,
                                                          const float* __restrict__ dy, int64_t lddy,
                                                          float* __restrict__ dx, int accumulate,
                                                          float* __restrict__ partial, int stride) {
    constexpr int NT = D / 16;
    constexpr int IMG = img_units<NT, NT>();
    constexpr int NW = lin_bwd_waves(D);
    extern __shared__ float4 lds4[];
    float4* imgt = lds4;                                         // [4] transposed images
    float* tile = reinterpret_cast<float*>(lds4 + 4 * IMG) + (threadIdx.x >> 6) * 16 * (D + 4);
    build_w<NT, NT, true>(imgt, W0, ld0, D);
    build_w<NT, NT, true>(imgt + IMG, W1, ld1, D);
    build_w<NT, NT, true>(imgt + 2 * IMG, W2, ld2, D);
    build_w<NT, NT, true>(imgt + 3 * IMG, W3, ld3, D);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t ntiles = (m + 15) / 16;
    float dbs[4][NT];
    f32x4 gw[4][NT][NT];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            dbs[k][jt] = 0.f;
            zero(gw[k][jt]);
        }
    const int64_t tstep = (int64_t)gridDim.x * NW;
    for (int64_t t = (int64_t)blockIdx.x * NW + (threadIdx.x >> 6); t < ntiles; t += tstep) {
        const int64_t row0 = t * 16;
        float4 a[NT];
        f32x4 xd[NT], g[4][NT], acc[NT];
        load_a<D>(a, x, row0, m, lane);
#pragma unroll
        for (int k = 0; k < 4; ++k) load_d<D>(g[k], dy + k * D, row0, m, lane, lddy);
        if (accumulate) load_d<D>(acc, dx, row0, m, lane);
        else zero(acc);
        a_to_d<D>(xd, a, tile, lane);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            wgrad_acc<NT, NT>(gw[k], g[k], xd);
            colsum_acc<NT>(dbs[k], g[k]);
            d_to_a<D>(a, g[k], tile, lane);
            mma_w<NT, NT>(acc, a, imgt + k * IMG, lane);
        }
        store_d<D>(acc, dx, row0, m, lane);
    }
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds4);
    constexpr int MAT = D * D, QS = MAT + D;
    constexpr int LAYOUT = 4 * IMG * 4 + NW * 16 * (D + 4);                // (the launch asks for max(layout, 4 QS floats))
    constexpr int S = pick_slots(NW, 4 * QS, LAYOUT > 4 * QS ? LAYOUT : 4 * QS);
    wg_sum_out<NW, S>(red, 4 * QS, partial + (size_t)blockIdx.x * stride, [&](float* r, bool first) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            red_add_mat<NT, NT>(r + k * QS, gw[k], lane, first);
            red_add_bias<NT>(r + k * QS + MAT, dbs[k], lane, first);
        }
    });
}

// ====================================================================================================================
// Layer heads (layers/global_message_passing.py:47-50): out[n] = o[n] . w_out + b_out, att[n] = o[n] . w_att
// D/4 lanes per row (one float4 each); backward: d o = g_out w_out + g_att w_att and the three parameter gradients
// reduced lane -> workgroup (LDS, fixed order) -> grid (narrow_reduce_kernel).
// ====================================================================================================================
template <int D>
__global__ __launch_bounds__(256) void nheads_fwd_kernel(const float* __restrict__ o, int64_t m,
                                                         const float* __restrict__ w_out, const float* __restrict__ b_out,
                                                         const float* __restrict__ w_att, float* __restrict__ out,
                                                         float* __restrict__ att) {
    constexpr int LPR = D / 4, RPB = 256 / LPR;
    const int sub = threadIdx.x % LPR, rl = threadIdx.x / LPR;
    const float4 wo = *reinterpret_cast<const float4*>(w_out + 4 * sub);
    const float4 wa = *reinterpret_cast<const float4*>(w_att + 4 * sub);
    const float bo = b_out[0];
    for (int64_t row = (int64_t)blockIdx.x * RPB + rl; row < m; row += (int64_t)gridDim.x * RPB) {   // uniform per lane group
        const float4 v = *reinterpret_cast<const float4*>(o + row * D + 4 * sub);
        float so = (v.x * wo.x + v.y * wo.y) + (v.z * wo.z + v.w * wo.w);
        float sa = (v.x * wa.x + v.y * wa.y) + (v.z * wa.z + v.w * wa.w);
#pragma unroll
        for (int s = LPR / 2; s >= 1; s >>= 1) {
            so += __shfl_xor(so, s, 64);
            sa += __shfl_xor(sa, s, 64);
        }
        if (sub == 0) {
            out[row] = so + bo;
            att[row] = sa;
        }
    }
}

template <int D>
__global__ __launch_bounds__(256) void nheads_bwd_kernel(const float* __restrict__ o, int64_t m,
                                                         const float* __restrict__ w_out, const float* __restrict__ w_att,
                                                         const float* __restrict__ g_out, const float* __restrict__ g_att,
                                                         float* __restrict__ d_o, float* __restrict__ partial) {
    constexpr int LPR = D / 4, RPB = 256 / LPR;
    __shared__ float red[256][9];
    const int sub = threadIdx.x % LPR, rl = threadIdx.x / LPR;
    const float4 wo = *reinterpret_cast<const float4*>(w_out + 4 * sub);
    const float4 wa = *reinterpret_cast<const float4*>(w_att + 4 * sub);
    float4 so = make_float4(0.f, 0.f, 0.f, 0.f), sa = so;
    float sb = 0.f;
    for (int64_t row = (int64_t)blockIdx.x * RPB + rl; row < m; row += (int64_t)gridDim.x * RPB) {
        const float4 v = *reinterpret_cast<const float4*>(o + row * D + 4 * sub);
        const float go = g_out[row], ga = g_att[row];
        *reinterpret_cast<float4*>(d_o + row * D + 4 * sub) =
            make_float4(go * wo.x + ga * wa.x, go * wo.y + ga * wa.y, go * wo.z + ga * wa.z, go * wo.w + ga * wa.w);
        so.x += go * v.x; so.y += go * v.y; so.z += go * v.z; so.w += go * v.w;
        sa.x += ga * v.x; sa.y += ga * v.y; sa.z += ga * v.z; sa.w += ga * v.w;
        sb += go;
    }
    float* mine = red[threadIdx.x];
    mine[0] = so.x; mine[1] = so.y; mine[2] = so.z; mine[3] = so.w;
    mine[4] = sa.x; mine[5] = sa.y; mine[6] = sa.z; mine[7] = sa.w;
    mine[8] = sb;
    __syncthreads();
    // partial row: [dw_out (D)][dw_att (D)][db_out]
    for (int p = threadIdx.x; p < 2 * D + 1; p += 256) {
        float s = 0.f;
        if (p < 2 * D) {
            const int which = p / D, c = p % D, su = c / 4, comp = which * 4 + (c & 3);
            for (int r = 0; r < RPB; ++r) s += red[r * LPR + su][comp];
        } else {
            for (int r = 0; r < RPB; ++r) s += red[r * LPR][8];
        }
        partial[(size_t)blockIdx.x * (2 * D + 1) + p] = s;
    }
}

// ====================================================================================================================
// Local-edge gates (layers/local_message_passing.py:46-48 after the W[x_i | x_j | rbf] split): per local edge q = (j -> i)
//   z1 = P[i, 0:D] + P[j, 2D:3D] + Q[q, 0:D] + b_ji          m_ji = SiLU(z1)
//   z2 = P[i, D:2D] + P[j, 3D:4D] + Q[q, D:2D] + b_kj         m_nb = SiLU(z2) * Q[q, 2D:3D]      (mlp_m_kj * lin_rbf)
// P [N, 4D] node-side projections, Q [E, 4D] edge-side projections (column block 3 = lin_rbf_out is used later).
// Elementwise: D/4 lanes per edge, float4 each.  Backward writes dz [E, 2D] (the caller segment-sums it into dP and
// column-sums it into the biases) and dQ[:, 0:3D] (block 3 zeroed).
// ====================================================================================================================
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float silu1(float z) { return z * sigmoidf_fast(z); }
__device__ __forceinline__ float dsilu1(float z) { const float s = sigmoidf_fast(z); return s * (1.0f + z * (1.0f - s)); }

template <int D>
__global__ __launch_bounds__(256) void nlocal_gate_fwd_kernel(const float* __restrict__ P, const float* __restrict__ Q,
                                                              const int32_t* __restrict__ tgt, const int32_t* __restrict__ src,
                                                              const float* __restrict__ bji, const float* __restrict__ bkj,
                                                              int64_t m, float* __restrict__ m_ji, float* __restrict__ m_nb) {
    constexpr int LPR = D / 4;
    const int64_t total = m * LPR;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t q = t / LPR;
        const int c = (int)(t % LPR) * 4;
        const float* pi = P + (size_t)tgt[q] * (4 * D) + c;
        const float* pj = P + (size_t)src[q] * (4 * D) + 2 * D + c;
        const float* qq = Q + q * (4 * D) + c;
        const float4 a1 = ld4(pi), a2 = ld4(pi + D), c1 = ld4(pj), c2 = ld4(pj + D);
        const float4 q1 = ld4(qq), q2 = ld4(qq + D), q3 = ld4(qq + 2 * D), b1 = ld4(bji + c), b2 = ld4(bkj + c);
        st4(m_ji + q * D + c, make_float4(silu1(a1.x + c1.x + q1.x + b1.x), silu1(a1.y + c1.y + q1.y + b1.y),
                                          silu1(a1.z + c1.z + q1.z + b1.z), silu1(a1.w + c1.w + q1.w + b1.w)));
        st4(m_nb + q * D + c, make_float4(silu1(a2.x + c2.x + q2.x + b2.x) * q3.x, silu1(a2.y + c2.y + q2.y + b2.y) * q3.y,
                                          silu1(a2.z + c2.z + q2.z + b2.z) * q3.z, silu1(a2.w + c2.w + q2.w + b2.w) * q3.w));
    }
}

template <int D>
__global__ __launch_bounds__(256) void nlocal_gate_bwd_kernel(const float* __restrict__ P, const float* __restrict__ Q,
                                                              const int32_t* __restrict__ tgt, const int32_t* __restrict__ src,
                                                              const float* __restrict__ bji, const float* __restrict__ bkj,
                                                              int64_t m, const float* __restrict__ g_ji,
                                                              const float* __restrict__ g_nb, float* __restrict__ dz,
                                                              float* __restrict__ dQ, int zero_q3) {
    constexpr int LPR = D / 4;
    const int64_t total = m * LPR;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t q = t / LPR;
        const int c = (int)(t % LPR) * 4;
        const float* pi = P + (size_t)tgt[q] * (4 * D) + c;
        const float* pj = P + (size_t)src[q] * (4 * D) + 2 * D + c;
        const float* qq = Q + q * (4 * D) + c;
        const float4 a1 = ld4(pi), a2 = ld4(pi + D), c1 = ld4(pj), c2 = ld4(pj + D);
        const float4 q1 = ld4(qq), q2 = ld4(qq + D), q3 = ld4(qq + 2 * D), b1 = ld4(bji + c), b2 = ld4(bkj + c);
        const float4 gj = ld4(g_ji + q * D + c), gn = ld4(g_nb + q * D + c);
        const float z1[4] = {a1.x + c1.x + q1.x + b1.x, a1.y + c1.y + q1.y + b1.y, a1.z + c1.z + q1.z + b1.z, a1.w + c1.w + q1.w + b1.w};
        const float z2[4] = {a2.x + c2.x + q2.x + b2.x, a2.y + c2.y + q2.y + b2.y, a2.z + c2.z + q2.z + b2.z, a2.w + c2.w + q2.w + b2.w};
        const float gjv[4] = {gj.x, gj.y, gj.z, gj.w}, gnv[4] = {gn.x, gn.y, gn.z, gn.w}, q3v[4] = {q3.x, q3.y, q3.z, q3.w};
        float d1[4], d2[4], d3[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            d1[u] = gjv[u] * dsilu1(z1[u]);
            d2[u] = gnv[u] * q3v[u] * dsilu1(z2[u]);
            d3[u] = gnv[u] * silu1(z2[u]);
        }
        st4(dz + q * (2 * D) + c, make_float4(d1[0], d1[1], d1[2], d1[3]));
        st4(dz + q * (2 * D) + D + c, make_float4(d2[0], d2[1], d2[2], d2[3]));
        float* dq = dQ + q * (4 * D) + c;
        st4(dq, make_float4(d1[0], d1[1], d1[2], d1[3]));
        st4(dq + D, make_float4(d2[0], d2[1], d2[2], d2[3]));
        st4(dq + 2 * D, make_float4(d3[0], d3[1], d3[2], d3[3]));
        if (zero_q3) st4(dq + 3 * D, make_float4(0.f, 0.f, 0.f, 0.f));
    }
}

// ====================================================================================================================
// Edge-embedding MLPs (models.py:185-188): y = SiLU(W f + b), f [rows, K] with K = 16 (Bessel) or 42 (spherical);
// with `kind` the row picks (Wa, ba) for kind 0 (triplet rows, mlp_sbf2) or (Wb, bb) for kind 1 (pair rows, mlp_sbf1).
// ====================================================================================================================
template <int K>
__device__ __forceinline__ void load_feat_a(float4 (&a)[(K + 15) / 16], const float* __restrict__ F, int64_t row0,
                                            int64_t m, int lane) {
    const int64_t row = row0 + (lane & 15);
    const bool ok = row < m;
    const float* p = F + (ok ? row : m - 1) * K;
#pragma unroll
    for (int q = 0; q < (K + 15) / 16; ++q) {
        const int k = 16 * q + 4 * (lane >> 4);
        // columns beyond K: read the row's last pair instead (valid address), zeroed by the selects
        const float2 lo = *reinterpret_cast<const float2*>(p + (k + 1 < K ? k : K - 2));
        const float2 hi = *reinterpret_cast<const float2*>(p + (k + 3 < K ? k + 2 : K - 2));
        const bool okl = ok && k + 1 < K, okh = ok && k + 3 < K;
        a[q] = make_float4(okl ? lo.x : 0.f, okl ? lo.y : 0.f, okh ? hi.x : 0.f, okh ? hi.y : 0.f);
    }
}

template <int K>
__device__ __forceinline__ void load_feat_d(f32x4 (&v)[(K + 15) / 16], const float* __restrict__ F, int64_t row0,
                                            int64_t m, int lane) {
    const int c = lane & 15, kg = lane >> 4;
    float t[4][(K + 15) / 16];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t row = row0 + 4 * kg + r;
        const float* p = F + (row < m ? row : m - 1) * K;
#pragma unroll
        for (int jk = 0; jk < (K + 15) / 16; ++jk) t[r][jk] = p[16 * jk + c < K ? 16 * jk + c : K - 1];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const bool ok = row0 + 4 * kg + r < m;
#pragma unroll
        for (int jk = 0; jk < (K + 15) / 16; ++jk) v[jk][r] = (ok && 16 * jk + c < K) ? t[r][jk] : 0.f;
    }
}

// Bessel rows formed inside the forward embedding kernel (RBF = true, K = 16; inference): the row of edge e is
// u(x) sin(freq_n x), x = dist[e] / cutoff (layers/basic.py:74-76) -- the arithmetic of rbf_fwd_kernel, so the values are
// the same floats; the [E, 16] tensor (55 MB at the RNA batch) is neither written nor read.
__device__ __forceinline__ float narrow_envelope(float x) {
    // layers/basic.py:36-51 with p = 5: 1/x - 21 x^5 + 35 x^6 - 15 x^7 for x < 1, else 0 (as envelope_f in basis.hip)
    if (!(x < 1.0f)) return 0.0f;
    const float x2 = x * x, x5 = x2 * x2 * x;
    return 1.0f / x + x5 * (-21.0f + x * (35.0f - 15.0f * x));
}

// the backward's recomputation: the hardware reciprocal (1 ulp) and no branch, so the loads around it stay in flight
__device__ __forceinline__ float narrow_envelope_rcp(float x) {
    const float x2 = x * x, x5 = x2 * x2 * x;
    const float u = __builtin_amdgcn_rcpf(x) + x5 * (-21.0f + x * (35.0f - 15.0f * x));
    return x < 1.0f ? u : 0.0f;
}

__device__ __forceinline__ void rbf_feat_a(float4 (&a)[1], const float* __restrict__ dist, const float (&f)[4],
                                           float inv_cutoff, int64_t row0, int64_t m, int lane) {
    const int64_t row = row0 + (lane & 15);
    const bool ok = row < m;
    const float x = dist[ok ? row : m - 1] * inv_cutoff;
    const float u = narrow_envelope(x);
    a[0] = ok ? make_float4(u * sinf(f[0] * x), u * sinf(f[1] * x), u * sinf(f[2] * x), u * sinf(f[3] * x))
              : make_float4(0.f, 0.f, 0.f, 0.f);
}

template <int D, int K, bool TWO, bool RBF = false>
__global__ __launch_bounds__(NWG) void nembed_fwd_kernel(const float* __restrict__ F, int64_t m,
                                                         const int32_t* __restrict__ kind,
                                                         const float* __restrict__ Wa, const float* __restrict__ ba,
                                                         const float* __restrict__ Wb, const float* __restrict__ bb,
                                                         float* __restrict__ y, const float* __restrict__ rbf_freq,
                                                         float rbf_inv_cutoff) {
    static_assert(!RBF || (K == 16 && !TWO), "Bessel rows are 16 wide, one weight set");
    constexpr int NT = D / 16, NQ = (K + 15) / 16;
    constexpr int IMG = NT * NQ * 64;
    extern __shared__ float4 lds4[];
    build_image<NT, NQ, false>(lds4, Wa, K, K);
    if (TWO) build_image<NT, NQ, false>(lds4 + IMG, Wb, K, K);
    __syncthreads();
    const int lane = threadIdx.x & 63, c = lane & 15, kg = lane >> 4;
    const int64_t ntiles = (m + 15) / 16;
    float bja[NT], bjb[NT];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
        bja[jt] = ba[16 * jt + c];
        bjb[jt] = TWO ? bb[16 * jt + c] : 0.f;
    }
    float fr[4] = {0.f, 0.f, 0.f, 0.f};                      // RBF: F is the edge-length vector [m]
    if constexpr (RBF) {
#pragma unroll
        for (int j = 0; j < 4; ++j) fr[j] = rbf_freq[4 * kg + j];
    }
    for (int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); t < ntiles; t += (int64_t)gridDim.x * 4) {
        const int64_t row0 = t * 16;
        float4 a[NQ];
        if constexpr (RBF) rbf_feat_a(a, F, fr, rbf_inv_cutoff, row0, m, lane);
        else load_feat_a<K>(a, F, row0, m, lane);
        f32x4 acc[NT];
        zero(acc);
        if (!TWO) {
            mma_img<NT, NQ>(acc, a, lds4, lane);
        } else {
            const int64_t row = row0 + c;
            const bool second = kind[row < m ? row : m - 1] != 0;
            float4 a0[NQ], a1[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                a0[q] = second ? make_float4(0.f, 0.f, 0.f, 0.f) : a[q];
                a1[q] = second ? a[q] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            mma_img<NT, NQ>(acc, a0, lds4, lane);
            mma_img<NT, NQ>(acc, a1, lds4 + IMG, lane);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = row0 + 4 * kg + r;
            const bool second = TWO && kind[row < m ? row : m - 1] != 0;
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                const float z = acc[jt][r] + (second ? bjb[jt] : bja[jt]);
                acc[jt][r] = z * sigmoidf_fast(z);
            }
        }
        store_d<D>(acc, y, row0, m, lane);
    }
}

// backward: partial = [dWa fragments (D x KP)][dWb fragments if TWO][dba (D)][dbb (D) if TWO]; df [m, K] only for
// the single-set K = 16 case (the Bessel frequencies are trainable, layers/basic.py:65-72).
// RBF (with DX): F is the edge-length vector [m]; the Bessel rows are formed here (as in the forward), and instead of
// storing df [m, 16] for a separate kernel the frequency gradient d freq_n = sum_e df[e][n] u(x) x cos(freq_n x)
// (layers/basic.py:76) is accumulated per lane and leaves in 16 more floats of the workgroup's partial row.
template <int D, int K, bool TWO, bool DX, bool RBF = false>
__global__ __launch_bounds__(64 * bwd_waves(D)) void nembed_bwd_kernel(const float* __restrict__ F, int64_t m,
                                                         const int32_t* __restrict__ kind,
                                                         const float* __restrict__ Wa, const float* __restrict__ ba,
                                                         const float* __restrict__ Wb, const float* __restrict__ bb,
                                                         const float* __restrict__ dy, float* __restrict__ df,
                                                         float* __restrict__ partial, int stride,
                                                         const float* __restrict__ rbf_freq, float rbf_inv_cutoff) {
    static_assert(!DX || (K == 16 && !TWO), "df only for the single-set 16-wide embedding");
    static_assert(!RBF || DX, "Bessel rows: the 16-wide single-set embedding with the frequency gradient");
    constexpr int NT = D / 16, NQ = (K + 15) / 16, KP = NQ * 16;
    constexpr int IMG = NT * NQ * 64;
    constexpr int IMGT = NQ * NT * 64;
    constexpr int NW = bwd_waves(D);
    extern __shared__ float4 lds4[];
    float4* img_t = lds4 + (TWO ? 2 : 1) * IMG;
    constexpr int TW = (KP > D ? KP : D) + 4;                // the wave's tile also transposes the feature rows (KP wide)
    float* tile = reinterpret_cast<float*>(lds4 + (TWO ? 2 : 1) * IMG + (DX ? IMGT : 0)) + (threadIdx.x >> 6) * 16 * TW;
    build_image<NT, NQ, false>(lds4, Wa, K, K);
    if (TWO) build_image<NT, NQ, false>(lds4 + IMG, Wb, K, K);
    if constexpr (DX) build_image<NQ, NT, true>(img_t, Wa, K, K);       // df = dz Wa: output tiles over K, k-groups over D
    __syncthreads();
    const int lane = threadIdx.x & 63, c = lane & 15, kg = lane >> 4;
    const int64_t ntiles = (m + 15) / 16;
    float bja[NT], bjb[NT], dba[NT], dbb[NT];
    f32x4 gwa[NT][NQ], gwb[TWO ? NT : 1][NQ];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
        bja[jt] = ba[16 * jt + c];
        bjb[jt] = TWO ? bb[16 * jt + c] : 0.f;
        dba[jt] = dbb[jt] = 0.f;
        zero(gwa[jt]);
        if constexpr (TWO) zero(gwb[jt]);
    }
    float fr[4] = {0.f, 0.f, 0.f, 0.f};
    float fc = 0.f, facc = 0.f;                              // RBF: this lane's column frequency, its d freq partial sum
    if constexpr (RBF) {
#pragma unroll
        for (int j = 0; j < 4; ++j) fr[j] = rbf_freq[4 * kg + j];
        fc = rbf_freq[c];
    }
    // One tile's global operands, fetched a tile ahead of their use: every load of a tile is issued before the previous
    // tile's arithmetic (no branch between them, so the waits are counted), instead of five dependent round trips per tile
    // (edge length -> envelope branch -> ... -> dy), which is what bounded the kernel (61 us at the RNA batch's 867 k edges
    // against 7 us of traffic).
    struct Raw {
        float xd[4];                                          // RBF: the edge lengths of rows 4 kg + r
        float4 a[NQ];                                         // feature rows, A layout
        f32x4 fd[NQ];                                         // feature rows, accumulator layout (K = 16 table rows)
        f32x4 dy[NT];
        int kc, kr[4];                                        // TWO: the kind of row c / of rows 4 kg + r
    };
    auto fetch = [&](int64_t t, Raw& w) {
        const int64_t row0 = t * 16;
        if constexpr (RBF) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = row0 + 4 * kg + r;
                w.xd[r] = F[row < m ? row : m - 1];
            }
        } else {
            load_feat_a<K>(w.a, F, row0, m, lane);
            if constexpr (K != 42) load_feat_d<K>(w.fd, F, row0, m, lane);
        }
        load_d<D>(w.dy, dy, row0, m, lane);
        if constexpr (TWO) {
            const int64_t rc = row0 + c;
            w.kc = kind[rc < m ? rc : m - 1];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = row0 + 4 * kg + r;
                w.kr[r] = kind[row < m ? row : m - 1];
            }
        }
    };
    const int64_t tstep = (int64_t)gridDim.x * NW;
    int64_t t = (int64_t)blockIdx.x * NW + (threadIdx.x >> 6);
    Raw cur, nxt;
    if (t < ntiles) fetch(t, cur);
    for (; t < ntiles; t += tstep) {
        const int64_t row0 = t * 16;
        fetch(t + tstep < ntiles ? t + tstep : t, nxt);       // (the last tile again: a valid address, never used)
        float4 a[NQ];
        f32x4 fd[NQ];                                         // the same rows in accumulator layout (rows 4 kg + r, column c)
        float xr[4], ur[4], cr[4];                            // RBF: x, u(x), cos(freq_c x) of this lane's four rows
        if constexpr (RBF) {
            // Accumulator layout first: sin and cos of one argument share the range reduction, and every Bessel value is
            // formed once -- the A operand is the tile's transpose through the wave's LDS scratch.  (Forming both layouts
            // on their own cost 8 sinf + 4 cosf per lane and tile and made the kernel VALU bound: 77 us at the RNA batch.)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = row0 + 4 * kg + r < m;
                xr[r] = cur.xd[r] * rbf_inv_cutoff;
                ur[r] = ok ? narrow_envelope_rcp(xr[r]) : 0.f;
                float sn;
                sincos_turns(fc * xr[r], &sn, &cr[r]);
                fd[0][r] = ur[r] * sn;
            }
            float4 a1[1];
            d_to_a<16>(a1, fd, tile, lane);
            a[0] = a1[0];
        } else {
#pragma unroll
            for (int q = 0; q < NQ; ++q) a[q] = cur.a[q];
        }
        f32x4 acc[NT];
        zero(acc);
        if (!TWO) {
            mma_img<NT, NQ>(acc, a, lds4, lane);
        } else {
            const bool second = cur.kc != 0;
            float4 a0[NQ], a1[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                a0[q] = second ? make_float4(0.f, 0.f, 0.f, 0.f) : a[q];
                a1[q] = second ? a[q] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            mma_img<NT, NQ>(acc, a0, lds4, lane);
            mma_img<NT, NQ>(acc, a1, lds4 + IMG, lane);
        }
        f32x4 dza[NT], dzb[TWO ? NT : 1];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool second = TWO && cur.kr[r] != 0;
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                const float z = acc[jt][r] + (second ? bjb[jt] : bja[jt]);
                const float s = sigmoidf_fast(z);
                const float dz = cur.dy[jt][r] * (s * (1.0f + z * (1.0f - s)));
                dza[jt][r] = second ? 0.f : dz;
                if constexpr (TWO) dzb[jt][r] = second ? dz : 0.f;
            }
        }
        if constexpr (!RBF) {
            // the rows were fetched once (A layout, 8-byte loads); their accumulator layout is the transpose through the wave's
            // LDS tile instead of twelve more 4-byte loads per lane
            if constexpr (K == 42) {
                a_to_d<KP>(fd, a, tile, lane);
            } else {
#pragma unroll
                for (int q = 0; q < NQ; ++q) fd[q] = cur.fd[q];
            }
        }
        wgrad_acc<NT, NQ>(gwa, dza, fd);
        colsum_acc<NT>(dba, dza);
        if constexpr (TWO) {
            wgrad_acc<NT, NQ>(gwb, dzb, fd);
            colsum_acc<NT>(dbb, dzb);
        }
        if constexpr (DX) {
            float4 az[NT];
            d_to_a<D>(az, dza, tile, lane);
            f32x4 o[NQ];
            zero(o);
            mma_img<NQ, NT>(o, az, img_t, lane);
            // K = 16: one 16-column tile
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = row0 + 4 * kg + r;
                if constexpr (RBF) facc += o[0][r] * ur[r] * xr[r] * cr[r];                // (u = 0 beyond the last row)
                else if (row < m) df[row * K + c] = o[0][r];
            }
        }
        cur = nxt;
    }
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds4);
    constexpr int MAT = D * KP;
    constexpr int NM = TWO ? 2 : 1;
    constexpr int N = NM * (MAT + D) + (RBF ? 16 : 0);
    constexpr int S = pick_slots(NW, N, ((TWO ? 2 : 1) * IMG + (DX ? IMGT : 0)) * 4 + NW * 16 * TW);
    wg_sum_out<NW, S>(red, N, partial + (size_t)blockIdx.x * stride, [&](float* r, bool first) {
        red_add_mat<NT, NQ>(r, gwa, lane, first);
        red_add_bias<NT>(r + NM * MAT, dba, lane, first);
        if constexpr (TWO) {
            red_add_mat<NT, NQ>(r + MAT, gwb, lane, first);
            red_add_bias<NT>(r + NM * MAT + D, dbb, lane, first);
        }
        if constexpr (RBF) {
            float fa[1] = {facc};
            red_add_bias<1>(r + NM * (MAT + D), fa, lane, first);
        }
    });
}

// ---- host side -------------------------------------------------------------------------------------------------------
inline int grid_for(int64_t m, int per_cu, int waves = 4) {
    const int64_t tiles = (m + 15) / 16;
    const int64_t want = (tiles + waves - 1) / waves;        // one tile per wave
    const int64_t cap = 256 * (int64_t)per_cu;
    return (int)(want < cap ? (want > 0 ? want : 1) : cap);
}

inline bool width_ok(int64_t d) { return d == 16 || d == 32 || d == 64; }

// Workgroups that are co-resident per CU: the backward kernels at d = 64 hold ~350 registers per lane and ~80 KB of
// weight images, so one 4-wave workgroup fills a CU; a grid beyond that only adds a second, nearly empty round.
inline int fwd_per_cu(int64_t d) { return d == 64 ? 2 : (d == 16 ? 8 : 4); }


template <typename Kern>
inline hipError_t allow_lds(Kern k, size_t bytes) {
    if (bytes <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

#define NARROW_DISPATCH(d, CALL)     \
    switch ((int)(d)) {              \
        case 16: { CALL(16); } break; \
        case 32: { CALL(32); } break; \
        default: { CALL(64); } break; \
    }

}  // namespace
