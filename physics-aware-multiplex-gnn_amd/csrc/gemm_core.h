// fp32 MFMA tile-GEMM core shared by the fused PAMNet kernels (gfx950).
//
// Every dense layer of PAMNet is  Y[rows, 128] = X[rows, K] * W[128, K]^T  with K = 128 (layers/basic.py:19-22 with
// dim=128; the 3*dim-wide message MLPs are split algebraically into 128-wide blocks).  fp32 inputs are mandatory
// (1e-5 parity), so the matrix unit is v_mfma_f32_16x16x4_f32: exact fp32, 32-cycle issue, 157 TF/s chip peak.
//
// Work split: a workgroup = 4 waves owns a tile of BM = 16*MT rows; wave w owns output columns [32w, 32w+32) (two
// 16-wide N tiles).  The X tile lives in LDS ([BM][LDT] floats, LDT = 132: the 4-float pad spreads ds_read_b128 rows
// over the 64 banks).  W is streamed straight from L2 as MFMA B fragments -- it is shared by every workgroup and
// stays L2 resident (64 KB per matrix) -- so LDS holds activations only and chains of layers keep their row tile
// on-chip from the first GEMM to the last.
//
// Fragment mapping (cdna_hip_programming.md section 3): for 16x16x4, lane l supplies A[i = l&15][k = l>>4] and
// B[k = l>>4][j = l&15]; D[reg r] is row (l>>4)*4 + r, column l&15.  k is a summation index, so each lane may fetch
// four consecutive k (one 16-byte load) and feed them to four consecutive MFMAs: lane l covers
// k = 16q + 4*(l>>4) + t, t = 0..3, identically for A (LDS row-major) and W ([out][in] row-major) -- both operands are
// read as float4 with no transposition anywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pamnet {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int DIM = 128;          // feature width handled by the fused kernels
constexpr int LDT = 132;          // LDS tile leading dimension (floats)
constexpr int WG = 256;           // threads per workgroup (4 waves)

// sigmoid through the hardware exp2 / rcp units (v_exp_f32, v_rcp_f32: ~1 ulp each): 4 VALU.  The activation epilogues run
// once per GEMM output element; __frcp_rn looked like the hardware reciprocal but compiles to the correctly rounded
// division sequence (v_div_scale x2, v_rcp, 5 fma, v_div_fmas, v_div_fixup: 14 VALU for the sigmoid, ~100 VALU per lane
// and layer of a node chain) -- __builtin_amdgcn_rcpf is the single instruction.
__device__ __forceinline__ float sigmoidf_fast(float z) { return __builtin_amdgcn_rcpf(1.0f + __expf(-z)); }
__device__ __forceinline__ float silu(float z) { return z * sigmoidf_fast(z); }
// d/dz [z * sigmoid(z)] = s * (1 + z * (1 - s))
__device__ __forceinline__ float dsilu(float z) {
    const float s = sigmoidf_fast(z);
    return s * (1.0f + z * (1.0f - s));
}

// sin and cos of one argument on the hardware units (v_sin_f32 / v_cos_f32 take REVOLUTIONS): the product with 1/(2 pi) is
// formed to ~2^-48 (head + fma tail), its whole turns are removed exactly, and the two instructions see |t| <= 0.5 -- about
// ten VALU for the pair where sincosf's range reduction and polynomials cost ~100.  Backward-only (the recomputed Bessel
// rows and their frequency derivative): forward values keep sinf, so outputs stay the floats of the separate basis kernel.
__device__ __forceinline__ float turns_of(float p) {
    constexpr float INV2PI_HI = 0.15915494f, INV2PI_LO = 6.4206382e-09f;     // 1/(2 pi) = HI + LO
    const float t = p * INV2PI_HI;
    const float e = fmaf(p, INV2PI_HI, -t) + p * INV2PI_LO;
    return (t - rintf(t)) + e;
}
// The gradients are then formed from rows that differ from the forward's by the units' absolute error (~1e-6 of the row's
// scale): bounded against the fp64 oracle at the band edges -- frequency 16 pi, x -> 0 and x -> 1 -- by
// tests/test_hip_fused.py::test_bessel_gradients_at_the_band_edges (measured worst 2e-6 relative).
// Build switch -DPAMNET_EXACT_SINCOS: libm's sinf / cosf / sincosf instead (the forward's own floats; ~10x the instructions).
#ifdef PAMNET_EXACT_SINCOS
__device__ __forceinline__ float sin_turns(float p) { return sinf(p); }
__device__ __forceinline__ float cos_turns(float p) { return cosf(p); }
__device__ __forceinline__ void sincos_turns(float p, float* sn, float* cs) { sincosf(p, sn, cs); }
#else
__device__ __forceinline__ float sin_turns(float p) { return __builtin_amdgcn_sinf(turns_of(p)); }
__device__ __forceinline__ float cos_turns(float p) { return __builtin_amdgcn_cosf(turns_of(p)); }
__device__ __forceinline__ void sincos_turns(float p, float* sn, float* cs) {
    const float f = turns_of(p);
    *sn = __builtin_amdgcn_sinf(f);
    *cs = __builtin_amdgcn_cosf(f);
}
#endif

// ---- fp32-accurate GEMMs on the bf16 matrix pipe ("bf16x6") -----------------------------------------------------------
// The f32-input MFMA runs at the fp32 VECTOR rate (64 FLOP/clk/SIMD: 32 cycles per 16x16x4) and shares the issue port
// with the VALU; v_mfma_f32_16x16x32_bf16 does 8x the work in ~17 cycles on the matrix pipe proper.  A float splits
// EXACTLY into three bf16 pieces (8 + 8 + 8 significand bits, each piece the round-to-nearest bf16 of the running
// residual; bf16 has fp32's exponent range, so no scaling is involved):  x = p0 + p1 + p2.  A product needs the six
// piece products with i + j <= 2 -- the three dropped ones are below 2^-24 of |x y|, the rounding error of one fp32
// multiply -- each exact in the fp32 accumulator (8 x 8 bit significands).  Six bf16 MFMAs per 32 k against eight fp32
// MFMAs of twice the cycles: 2.5x fewer matrix-pipe cycles at fp32 accuracy (tests/test_hip_kernels.py::test_wgrad_*:
// error against fp64 at the level of an fp32 GEMM; a numpy restatement of the arithmetic is in tools/bf16x6_check.py).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// pieces of the pair (x0, x1), each as one packed dword (x0 in the low half): v_cvt_pk_bf16_f32 rounds to nearest even
__device__ __forceinline__ void split3(float x0, float x1, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
    const f32x2 v = {x0, x1};
    p0 = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
    const f32x2 r = {x0 - __uint_as_float(p0 << 16), x1 - __uint_as_float(p0 & 0xffff0000u)};      // exact
    p1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2));
    const f32x2 q = {r[0] - __uint_as_float(p1 << 16), r[1] - __uint_as_float(p1 & 0xffff0000u)};  // exact
    p2 = __builtin_bit_cast(uint32_t, __builtin_convertvector(q, bf16x2));
}

// the same in three stages (4 / 4 / 1 VALU) that a kernel can place between MFMAs one at a time; `r` carries the residual
template <int STAGE>
__device__ __forceinline__ void split3_stage(float x0, float x1, uint32_t& p0, uint32_t& p1, uint32_t& p2, f32x2& r) {
    if constexpr (STAGE == 0) {
        const f32x2 v = {x0, x1};
        p0 = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
        r = f32x2{x0 - __uint_as_float(p0 << 16), x1 - __uint_as_float(p0 & 0xffff0000u)};
    } else if constexpr (STAGE == 1) {
        p1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2));
        r = f32x2{r[0] - __uint_as_float(p1 << 16), r[1] - __uint_as_float(p1 & 0xffff0000u)};
    } else {
        p2 = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2));
    }
}

// An MFMA operand fragment (8 consecutive-k values of one row / column per lane) as three piece planes.
struct Frag3 {
    uint32_t p[3][4];
};
__device__ __forceinline__ Frag3 split_frag(const float (&v)[8]) {
    Frag3 f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t a, b, c;
        split3(v[2 * i], v[2 * i + 1], a, b, c);
        f.p[0][i] = a, f.p[1][i] = b, f.p[2][i] = c;
    }
    return f;
}
__device__ __forceinline__ f32x4 mfma_bf16(const uint32_t (&a)[4], const uint32_t (&b)[4], const f32x4& c) {
    const u32x4 av = {a[0], a[1], a[2], a[3]}, bv = {b[0], b[1], b[2], b[3]};
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), c, 0, 0, 0);
}

template <int MT>
__device__ __forceinline__ void acc_zero(f32x4 (&acc)[MT][2]) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        acc[m][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc[m][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// acc[m][n] += As[m*16 .. , 0:128] * W(block)^T for this wave's 32 output columns [wcol0, wcol0 + 32).
//   TRANS = false: W is [out][in] (row stride ldw): Y = X * W^T   (forward:  Linear)
//   TRANS = true : W is [in'][out'] and we need Y = X * W, i.e. B[k][j] = W[k][j] (backward: dX = dZ * W)
template <int MT, bool TRANS>
__device__ __forceinline__ void mma_tile(const float* __restrict__ As, const float* __restrict__ W, int ldw, int wcol0,
                                         f32x4 (&acc)[MT][2]) {
    const int lane = threadIdx.x & 63;
    const int r16 = lane & 15, kg = lane >> 4;
    const float* ap = As + r16 * LDT + 4 * kg;
#pragma unroll
    for (int q = 0; q < DIM / 16; ++q) {
        float4 b0, b1;
        if (!TRANS) {
            const float* wp = W + (size_t)(wcol0 + r16) * ldw + 4 * kg + 16 * q;
            b0 = *reinterpret_cast<const float4*>(wp);
            b1 = *reinterpret_cast<const float4*>(wp + (size_t)16 * ldw);
        } else {
            const float* wp = W + (size_t)(16 * q + 4 * kg) * ldw + wcol0 + r16;
            b0 = make_float4(wp[0], wp[ldw], wp[2 * (size_t)ldw], wp[3 * (size_t)ldw]);
            b1 = make_float4(wp[16], wp[ldw + 16], wp[2 * (size_t)ldw + 16], wp[3 * (size_t)ldw + 16]);
        }
        float4 a[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) a[m] = *reinterpret_cast<const float4*>(ap + m * 16 * LDT + 16 * q);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].x, b0.x, acc[m][0], 0, 0, 0);
            acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].x, b1.x, acc[m][1], 0, 0, 0);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].y, b0.y, acc[m][0], 0, 0, 0);
            acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].y, b1.y, acc[m][1], 0, 0, 0);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].z, b0.z, acc[m][0], 0, 0, 0);
            acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].z, b1.z, acc[m][1], 0, 0, 0);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].w, b0.w, acc[m][0], 0, 0, 0);
            acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].w, b1.w, acc[m][1], 0, 0, 0);
        }
    }
}

// ---- explicit weight-fragment prefetch ------------------------------------------------------------------------------
// Left to itself the compiler fetches each B fragment right before the MFMAs that use it, so every k-step of a GEMM
// waits a full L2 round trip (8 per layer; measured 44 us for the 10-layer node chain whose MFMAs need ~9 us).  A
// WFrag holds one whole 128x32 weight slice of a wave (16 x float4 = 64 VGPRs): all 16 loads are issued back to back,
// and the NEXT layer's slice is requested as soon as the current MFMAs are issued, so its latency hides behind the
// epilogue (LDS scatter, barrier, activation sweep).
struct WFrag {
    float4 b[DIM / 16][2];
};

template <bool TRANS>
__device__ __forceinline__ void load_wfrag(WFrag& f, const float* __restrict__ W, int ldw, int wcol0) {
    const int lane = threadIdx.x & 63;
    const int r16 = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int q = 0; q < DIM / 16; ++q) {
        if (!TRANS) {
            const float* wp = W + (size_t)(wcol0 + r16) * ldw + 4 * kg + 16 * q;
            f.b[q][0] = *reinterpret_cast<const float4*>(wp);
            f.b[q][1] = *reinterpret_cast<const float4*>(wp + (size_t)16 * ldw);
        } else {
            const float* wp = W + (size_t)(16 * q + 4 * kg) * ldw + wcol0 + r16;
            f.b[q][0] = make_float4(wp[0], wp[ldw], wp[2 * (size_t)ldw], wp[3 * (size_t)ldw]);
            f.b[q][1] = make_float4(wp[16], wp[ldw + 16], wp[2 * (size_t)ldw + 16], wp[3 * (size_t)ldw + 16]);
        }
    }
}

// Fragment-ordered weight image (pamnet_pack_weights_f32): img4[(j*8 + q)*64 + lane] holds, for 16-column tile j and
// k-group q, the float4 each lane feeds to the MFMAs -- a wave's request for one k-group is one contiguous 1 KB read.
constexpr int IMG = DIM * DIM;               // floats per image
__device__ __forceinline__ void load_wfrag_img(WFrag& f, const float* __restrict__ img) {       // 4 waves x 32 columns
    const float4* p4 = reinterpret_cast<const float4*>(img) + (threadIdx.x & 63);
    const int j = (threadIdx.x >> 6) * 2;
#pragma unroll
    for (int q = 0; q < DIM / 16; ++q) {
        f.b[q][0] = p4[(j * 8 + q) * 64];
        f.b[q][1] = p4[((j + 1) * 8 + q) * 64];
    }
}

// next weight slice of a 4-wave chain (TRANS selects the row-major access pattern; images carry their orientation): fragment-ordered image, or row-major matrix with row stride ld
template <bool PACKED, bool TRANS = false>
__device__ __forceinline__ void load_w(WFrag& f, const float* __restrict__ W, int ld, int wc) {
    if constexpr (PACKED) load_wfrag_img(f, W);
    else load_wfrag<TRANS>(f, W, ld, wc);
}


template <int MT>
__device__ __forceinline__ void mma_tile_frag(const float* __restrict__ As, const WFrag& f, f32x4 (&acc)[MT][2]) {
    const int lane = threadIdx.x & 63;
    const float* ap = As + (lane & 15) * LDT + 4 * (lane >> 4);
    // A fragments: for the 16-row chains (MT = 1) the whole K = 128 strip is 8 x float4 = 32 VGPRs -- read it with
    // eight back-to-back ds_read_b128 so the LDS latency is paid once per GEMM, not once per k-step.
    constexpr int QA = (MT <= 2) ? DIM / 16 : 1;
    float4 apre[QA][MT];
    if (MT <= 2) {
#pragma unroll
        for (int q = 0; q < DIM / 16; ++q)
#pragma unroll
            for (int m = 0; m < MT; ++m) apre[q % QA][m] = *reinterpret_cast<const float4*>(ap + m * 16 * LDT + 16 * q);
    }
#pragma unroll
    for (int q = 0; q < DIM / 16; ++q) {
        float4 a[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
            a[m] = (MT <= 2) ? apre[q % QA][m] : *reinterpret_cast<const float4*>(ap + m * 16 * LDT + 16 * q);
        const float4 b0 = f.b[q][0], b1 = f.b[q][1];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].x, b0.x, acc[m][0], 0, 0, 0);
            acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].x, b1.x, acc[m][1], 0, 0, 0);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].y, b0.y, acc[m][0], 0, 0, 0);
            acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].y, b1.y, acc[m][1], 0, 0, 0);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].z, b0.z, acc[m][0], 0, 0, 0);
            acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].z, b1.z, acc[m][1], 0, 0, 0);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].w, b0.w, acc[m][0], 0, 0, 0);
            acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].w, b1.w, acc[m][1], 0, 0, 0);
        }
    }
}

// The same products with the operands of the MFMA swapped (weights as A, activations as B): the accumulator of lane l then
// holds out[row l & 15][channels 16 n2 + 4 (l >> 4) + 0..3] -- four CONSECUTIVE channels of one row instead of four rows of one
// channel, so an epilogue parks a lane's values with one 16-byte LDS access where it took four 4-byte ones.  Every element is the
// same sum of the same products in the same order (the k mapping of both operands is unchanged): bitwise the values of
// mma_tile_frag.  16-row tiles.
__device__ __forceinline__ void mma_tile_frag_t(const float* __restrict__ As, const WFrag& f, f32x4 (&acc)[2]) {
    const int lane = threadIdx.x & 63;
    const float* ap = As + (lane & 15) * LDT + 4 * (lane >> 4);
    float4 apre[DIM / 16];
#pragma unroll
    for (int q = 0; q < DIM / 16; ++q) apre[q] = *reinterpret_cast<const float4*>(ap + 16 * q);
#pragma unroll
    for (int q = 0; q < DIM / 16; ++q) {
        const float4 a = apre[q], b0 = f.b[q][0], b1 = f.b[q][1];
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0.x, a.x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1.x, a.x, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0.y, a.y, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1.y, a.y, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0.z, a.z, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1.z, a.z, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0.w, a.w, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1.w, a.w, acc[1], 0, 0, 0);
    }
}
// the eight bias values of such a lane: channels dcol0 + 16 n2 + 4 (l >> 4) + 0..3
struct Bias8 {
    float4 v[2];
};
__device__ __forceinline__ Bias8 load_bias8(const float* __restrict__ bias, int dcol0) {
    Bias8 b;
    const int kg = (threadIdx.x & 63) >> 4;
    b.v[0] = bias ? *reinterpret_cast<const float4*>(bias + dcol0 + 4 * kg) : make_float4(0.f, 0.f, 0.f, 0.f);
    b.v[1] = bias ? *reinterpret_cast<const float4*>(bias + dcol0 + 16 + 4 * kg) : make_float4(0.f, 0.f, 0.f, 0.f);
    return b;
}

// The two bias values a lane needs (columns dcol0 + r16 and + 16).  Fetch them BEFORE issuing a weight prefetch: vmcnt
// retires in order, so a bias load issued after the prefetch would make its s_waitcnt drain the whole prefetch.
struct Bias2 {
    float v[2];
};
__device__ __forceinline__ Bias2 load_bias2(const float* __restrict__ bias, int dcol0) {
    Bias2 b;
    const int r16 = threadIdx.x & 15;
    b.v[0] = bias ? bias[dcol0 + r16] : 0.f;
    b.v[1] = bias ? bias[dcol0 + 16 + r16] : 0.f;
    return b;
}

template <int MT>
__device__ __forceinline__ void acc_to_lds(const f32x4 (&acc)[MT][2], float* __restrict__ Ds, int dcol0, Bias2 bias) {
    const int lane = threadIdx.x & 63;
    const int r16 = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int col = dcol0 + 16 * n + r16;
        const float bv = bias.v[n];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            float* d = Ds + (m * 16 + kg * 4) * LDT + col;
            d[0 * LDT] = acc[m][n][0] + bv;
            d[1 * LDT] = acc[m][n][1] + bv;
            d[2 * LDT] = acc[m][n][2] + bv;
            d[3 * LDT] = acc[m][n][3] + bv;
        }
    }
}

// Cooperative row-major sweep over a [BM][128] tile: thread t owns float4 column c4 = t & 31 of rows (t >> 5) + 8*i.
// f(row_in_tile, c4) is called BM/8 times per thread; global accesses inside f are 512-byte coalesced rows.
template <int BM, typename F>
__device__ __forceinline__ void sweep_rows(F&& f) {
    const int c4 = threadIdx.x & 31, r0 = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < BM / 8; ++i) f(r0 + 8 * i, c4);
}

__device__ __forceinline__ float4 lds4(const float* tile, int row, int c4) {
    return *reinterpret_cast<const float4*>(tile + row * LDT + 4 * c4);
}
__device__ __forceinline__ void st_lds4(float* tile, int row, int c4, float4 v) {
    *reinterpret_cast<float4*>(tile + row * LDT + 4 * c4) = v;
}
__device__ __forceinline__ float4 ldg4(const float* p, int64_t row, int ld, int c4) {
    return *reinterpret_cast<const float4*>(p + row * ld + 4 * c4);
}
// Row-guarded load without a branch: out-of-range rows read row 0 and are zeroed, so the loads of a sweep stay
// independent and are issued back to back (a branch per row makes the compiler wait for each load separately).
__device__ __forceinline__ float4 ldg4z(const float* p, int64_t row, int64_t nrows, int ld, int c4) {
    const bool ok = row < nrows;
    float4 v = *reinterpret_cast<const float4*>(p + (ok ? row : 0) * ld + 4 * c4);
    if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
    return v;
}
__device__ __forceinline__ void stg4(float* p, int64_t row, int ld, int c4, float4 v) {
    *reinterpret_cast<float4*>(p + row * ld + 4 * c4) = v;
}
// Streaming forms for rows that are touched exactly once per launch (edge-level inputs read once, backward-only saves
// written once): the non-temporal hint keeps them from pushing the re-used node-plane rows out of the XCD's 4 MB L2.
__device__ __forceinline__ float4 ldg4z_nt(const float* p, int64_t row, int64_t nrows, int ld, int c4) {
    const bool ok = row < nrows;
    const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + (ok ? row : 0) * ld + 4 * c4));
    return ok ? make_float4(t[0], t[1], t[2], t[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
}
// the same, out-of-range rows reading the LAST valid row (nrows >= 1): for loads every workgroup issues behind its own range
// (row 0 would be one cache line hammered by all of them)
__device__ __forceinline__ float4 ldg4zl_nt(const float* p, int64_t row, int64_t nrows, int ld, int c4) {
    const bool ok = row < nrows;
    const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + (ok ? row : nrows - 1) * ld + 4 * c4));
    return ok ? make_float4(t[0], t[1], t[2], t[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ float4 ldg4_nt(const float* p, int64_t row, int ld, int c4) {
    const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + row * ld + 4 * c4));
    return make_float4(t[0], t[1], t[2], t[3]);
}
__device__ __forceinline__ void st_nt4(float4* p, const float4& v) {
    const f32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(p));
}
__device__ __forceinline__ void stg4_nt(float* p, int64_t row, int ld, int c4, float4 v) {
    const f32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(p + row * ld + 4 * c4));
}
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 f4silu(float4 z) { return make_float4(silu(z.x), silu(z.y), silu(z.z), silu(z.w)); }
__device__ __forceinline__ float4 f4dsilu(float4 z) { return make_float4(dsilu(z.x), dsilu(z.y), dsilu(z.z), dsilu(z.w)); }
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

}  // namespace pamnet
