// Shared device helpers of the edge-level kernels (edge_chain.hip, edge_agg.hip): one 8- or 4-wave workgroup per CU,
// wave w owns 128 / NW output columns, weight slices resident in registers, row tiles of 16 in LDS (see edge_chain.hip).
#pragma once
#include "common.h"
#include "gemm_core.h"

namespace {
namespace edge {
using namespace pamnet;

constexpr int WG8 = 512;                  // 8 waves
constexpr int N_CU = 256;                 // MI355X
constexpr int MT2 = 8;                    // largest chunk (16-row tiles) of the two-slot kernels (2 x 128 rows = 132 KB)

// 16-byte reads of a weight image (round 6) as buffer loads: resource = base + size in bytes (reads past the image return 0)
typedef __amdgpu_buffer_rsrc_t ImgRsrc;
__device__ __forceinline__ ImgRsrc img_rsrc(const float* img, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(img), 0, bytes, 0x00020000);   // gfx9 raw buffer, 32-bit format
}
__device__ __forceinline__ u32x4 img_load16(ImgRsrc r, int byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
}

// one 128 x 16 weight slice of a wave: 8 x float4 = 32 VGPRs
struct WFrag1 {
    float4 b[DIM / 16];
};
//   TRANS = false: W is [out][in] (row stride ldw): Y = X * W^T   (forward:  Linear)
//   TRANS = true : Y = X * W                                       (backward: dX = dZ * W)
// ldw == 0 (round 6): `W` is the matrix's fp32 fragment image (pamnet_pack_weights_f32 / _mixed_f32 kind 0, made in the
// orientation the kernel wants): img4[(tile * 8 + q) * 64 + lane] -- the same float4 values, one contiguous 1 KB read per wave
// and k-group instead of (transposed) four dword loads per lane across 64-byte segments.
template <bool TRANS>
__device__ __forceinline__ void load_wfrag1(WFrag1& f, const float* __restrict__ W, int ldw, int wc) {
    const int lane = threadIdx.x & 63;
    const int r16 = lane & 15, kg = lane >> 4;
    if (ldw == 0) {
        // (buffer loads: as plain 16-byte loads the compiler scalarises them and sinks them into the other arm's 32 dword loads,
        // addresses selected by the branch -- four 16-byte-strided dword requests per float4, 26 -> 33 us for the backward pair)
        const ImgRsrc r = img_rsrc(W, DIM * DIM * 4);
        const int off = ((wc >> 4) * (DIM / 16 * 64) + lane) * 16;
#pragma unroll
        for (int q = 0; q < DIM / 16; ++q) {
            const u32x4 u = img_load16(r, off + q * 1024);
            f.b[q] = make_float4(__uint_as_float(u[0]), __uint_as_float(u[1]), __uint_as_float(u[2]), __uint_as_float(u[3]));
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < DIM / 16; ++q) {
        if (!TRANS) {
            f.b[q] = *reinterpret_cast<const float4*>(W + (size_t)(wc + r16) * ldw + 4 * kg + 16 * q);
        } else {
            const float* wp = W + (size_t)(16 * q + 4 * kg) * ldw + wc + r16;
            f.b[q] = make_float4(wp[0], wp[ldw], wp[2 * (size_t)ldw], wp[3 * (size_t)ldw]);
        }
    }
}

template <int MTX>
struct Acc {
    f32x4 v[MTX];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < MTX; ++i) v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
};

// acc[0..MT) += As[16 m .. 16 m + 16, 0:128] * slice   (MT independent MFMA chains per k-step)
template <int MT, int MTX>
__device__ __forceinline__ void mma_strip(const float* __restrict__ As, const WFrag1& f, Acc<MTX>& acc) {
    const int lane = threadIdx.x & 63;
    const float* ap = As + (lane & 15) * LDT + 4 * (lane >> 4);
#pragma unroll
    for (int q = 0; q < DIM / 16; ++q) {
        float4 a[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) a[m] = *reinterpret_cast<const float4*>(ap + m * 16 * LDT + 16 * q);
        const float4 b = f.b[q];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc.v[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].x, b.x, acc.v[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < MT; ++m) acc.v[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].y, b.y, acc.v[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < MT; ++m) acc.v[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].z, b.z, acc.v[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < MT; ++m) acc.v[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].w, b.w, acc.v[m], 0, 0, 0);
    }
}
// mt (1..MTX) live 16-row tiles; mt is uniform over the workgroup
template <int MTX>
__device__ __forceinline__ void mma_n(const float* __restrict__ As, const WFrag1& f, Acc<MTX>& acc, int mt) {
    switch (mt) {
        case 1: mma_strip<1, MTX>(As, f, acc); break;
        case 2: mma_strip<2, MTX>(As, f, acc); break;
        case 3: mma_strip<3, MTX>(As, f, acc); break;
        case 4: mma_strip<4, MTX>(As, f, acc); break;
        case 5: mma_strip<5, MTX>(As, f, acc); break;
        case 6: mma_strip<6, MTX>(As, f, acc); break;
        case 7:
            if constexpr (MTX >= 7) mma_strip<7, MTX>(As, f, acc);
            break;
        case 8:
            if constexpr (MTX >= 8) mma_strip<8, MTX>(As, f, acc);
            break;
        default:
            if constexpr (MTX >= 9) mma_strip<9, MTX>(As, f, acc);
            else if constexpr (MTX >= 8) mma_strip<8, MTX>(As, f, acc);
            break;
    }
}
// D[row][wc + col] = acc + bias for the live tiles (accumulator layout: rows 4*(lane>>4) + r, column lane & 15)
// (tstride: floats between consecutive 16-row tiles -- 16 LDT for a plain tile array)
template <int MTX>
__device__ __forceinline__ void acc_store(const Acc<MTX>& acc, float* __restrict__ Ds, int wc, float bias, int mt,
                                          int tstride = 16 * LDT) {
    const int lane = threadIdx.x & 63;
    const int col = wc + (lane & 15), kg = lane >> 4;
#pragma unroll
    for (int m = 0; m < MTX; ++m) {
        if (m < mt) {
            float* d = Ds + m * tstride + kg * 4 * LDT + col;
            d[0 * LDT] = acc.v[m][0] + bias;
            d[1 * LDT] = acc.v[m][1] + bias;
            d[2 * LDT] = acc.v[m][2] + bias;
            d[3 * LDT] = acc.v[m][3] + bias;
        }
    }
}
// ---- the same tile GEMMs on the bf16 matrix pipe at fp32 accuracy ("bf16x6", gemm_core.h) ------------------------------
// The f32-input MFMA above issues at the fp32 VECTOR rate (32 cycles per 16x16x4 on a SIMD); v_mfma_f32_16x16x32_bf16 does
// 8x the k in ~17.  A wave's 128 x 16 weight slice is split ONCE into its three exact bf16 pieces when it is loaded
// (48 registers instead of 32, resident for the whole launch); the activation tile stays fp32 in LDS -- no piece planes, no
// extra LDS -- and every wave splits the A fragments it reads (8 consecutive k of a row per lane: two ds_read_b128, four
// split3 = 36 VALU per 16 x 32 fragment).  That split is redundant across the 8 waves, so it pays where the fragment feeds
// more than one GEMM or where the MFMA share dominates: per 16 rows x 32 k the fused global-edge forward (two GEMMs on the
// same rows) issues 12 bf16 MFMAs + 36 VALU (~276 cycles) against 16 fp32 MFMAs (512 cycles).
// Fragment maps of v_mfma_f32_16x16x32_bf16: lane l supplies A[i = l & 15][k = 8 (l >> 4) + 0..7] and
// B[k = 8 (l >> 4) + 0..7][j = l & 15]; the accumulator layout is that of the 16x16x4 form (rows 4 (l >> 4) + r, column
// l & 15), so epilogues and stores are shared.
struct WFragB1 {
    uint32_t p[DIM / 32][3][4];               // [k-step of 32][piece][4 dwords = 8 bf16]
};
//   TRANS = false: W is [out][in] (row stride ldw): B[k][j] = W[wc + j][k]   (forward:  Y = X W^T)
//   TRANS = true : B[k][j] = W[k][wc + j]                                     (backward: dX = dZ W)
// the pieces of k-step q of the 16-column tile at wc, as lane `lane` holds them
template <bool TRANS>
__device__ __forceinline__ Frag3 wfragb1_q(const float* __restrict__ W, int ldw, int wc, int q, int lane) {
    const int j = lane & 15, kg = lane >> 4;
    float v[8];
    if (!TRANS) {
        const float* wp = W + (size_t)(wc + j) * ldw + 32 * q + 8 * kg;
        const float4 a = *reinterpret_cast<const float4*>(wp), b = *reinterpret_cast<const float4*>(wp + 4);
        v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
    } else {
        const float* wp = W + (size_t)(32 * q + 8 * kg) * ldw + wc + j;
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = wp[(size_t)t * ldw];
    }
    return split_frag(v);
}
// Ready-made image of these fragments (round 6; pamnet_pack_weights_mixed_f32 kind 1, node_tail.hip): every workgroup of every
// edge-level launch loaded its two to four 128 x 128 slices as fp32 (dword loads in the transposed orientation) and split them
// into pieces -- ~300 VALU and 64-128 loads per lane ahead of the first row, 2 us of a 30 us kernel at the QM9 batch.  The image
// holds the SAME pieces in the order the lanes want them: uint4 image[((tile * 4 + q) * 3 + piece) * 64 + lane], tile = wc / 16;
// a matrix is 8 x 4 x 3 x 64 x 16 bytes = 96 KB.  Selected by ldw == 0 (`W` then IS the image); results are bitwise the same.
// ARM: -1 = by ldw at run time; 1 = images only, 0 = matrices only (compile-time, for loads inside a row loop: with both arms
// there the compiler keeps the fragments in scratch)
constexpr int64_t EDGE_IMG_FLOATS = 3 * DIM * DIM / 2;
template <bool TRANS, int ARM = -1>
__device__ __forceinline__ void load_wfragb1(WFragB1& f, const float* __restrict__ W, int ldw, int wc) {
    const int lane = threadIdx.x & 63;
    if (ARM == 1 || (ARM < 0 && ldw == 0)) {
        const ImgRsrc r = img_rsrc(W, EDGE_IMG_FLOATS * 4);
        const int off = ((wc >> 4) * (DIM / 32 * 3 * 64) + lane) * 16;
#pragma unroll
        for (int q = 0; q < DIM / 32; ++q)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
                const u32x4 u = img_load16(r, off + (q * 3 + pc) * 1024);
                f.p[q][pc][0] = u[0], f.p[q][pc][1] = u[1], f.p[q][pc][2] = u[2], f.p[q][pc][3] = u[3];
            }
        return;
    }
    if (ARM == 0 && TRANS) {
        // inside a row loop: the 32 row addresses of a slice are loop invariants, and hoisted as 64-bit pairs they are 64 VGPRs a
        // slice (696 bytes of scratch for the four of the local edge backward) -- one lane offset and scalar row offsets instead
        const ImgRsrc r = img_rsrc(W, ldw * DIM * 4);
        const int voff = ((8 * (lane >> 4)) * ldw + wc + (lane & 15)) * 4;
#pragma unroll
        for (int q = 0; q < DIM / 32; ++q) {
            float v[8];
#pragma unroll
            for (int t = 0; t < 8; ++t)
                v[t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, (32 * q + t) * ldw * 4, 0));
            const Frag3 fr = split_frag(v);
#pragma unroll
            for (int pc = 0; pc < 3; ++pc)
#pragma unroll
                for (int t = 0; t < 4; ++t) f.p[q][pc][t] = fr.p[pc][t];
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < DIM / 32; ++q) {
        const Frag3 fr = wfragb1_q<TRANS>(W, ldw, wc, q, lane);
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
#pragma unroll
            for (int t = 0; t < 4; ++t) f.p[q][pc][t] = fr.p[pc][t];
    }
}
// A fragment (rows 16 m + (l & 15), k = 32 q + 8 (l >> 4) + 0..7) of an fp32 LDS tile, split into its pieces
__device__ __forceinline__ Frag3 lds_frag3(const float* __restrict__ As, int m, int q) {
    const int lane = threadIdx.x & 63;
    const float* ap = As + (m * 16 + (lane & 15)) * LDT + 32 * q + 8 * (lane >> 4);
    const float4 a = *reinterpret_cast<const float4*>(ap), b = *reinterpret_cast<const float4*>(ap + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    return split_frag(v);
}
// the six piece products with i + j <= 2 of one fragment pair, small ones first; G independent accumulators interleaved
template <int G, int MTX>
__device__ __forceinline__ void mfma6(const Frag3 (&a)[G], const uint32_t (&b)[3][4], Acc<MTX>& acc, int m0) {
#pragma unroll
    for (int g = 0; g < G; ++g) acc.v[m0 + g] = mfma_bf16(a[g].p[2], b[0], acc.v[m0 + g]);
#pragma unroll
    for (int g = 0; g < G; ++g) acc.v[m0 + g] = mfma_bf16(a[g].p[1], b[1], acc.v[m0 + g]);
#pragma unroll
    for (int g = 0; g < G; ++g) acc.v[m0 + g] = mfma_bf16(a[g].p[0], b[2], acc.v[m0 + g]);
#pragma unroll
    for (int g = 0; g < G; ++g) acc.v[m0 + g] = mfma_bf16(a[g].p[1], b[0], acc.v[m0 + g]);
#pragma unroll
    for (int g = 0; g < G; ++g) acc.v[m0 + g] = mfma_bf16(a[g].p[0], b[1], acc.v[m0 + g]);
#pragma unroll
    for (int g = 0; g < G; ++g) acc.v[m0 + g] = mfma_bf16(a[g].p[0], b[0], acc.v[m0 + g]);
}
// acc1[m0 .. m0 + G) += As * f1,  acc2[...] += As * f2 (f2 nullable at compile time: ONE = true): G <= 3 row tiles at a time
template <int G, int MTX, bool ONE>
__device__ __forceinline__ void mma_b16_group(const float* __restrict__ As, int m0, const WFragB1& f1, Acc<MTX>& acc1,
                                              const WFragB1& f2, Acc<MTX>& acc2) {
#pragma unroll
    for (int q = 0; q < DIM / 32; ++q) {
        Frag3 a[G];
#pragma unroll
        for (int g = 0; g < G; ++g) a[g] = lds_frag3(As, m0 + g, q);
        mfma6<G, MTX>(a, f1.p[q], acc1, m0);
        if constexpr (!ONE) mfma6<G, MTX>(a, f2.p[q], acc2, m0);
    }
}
// mt (1..MTX, workgroup-uniform) live 16-row tiles, in groups of GMAX <= 3 (the A pieces of a group are 12 GMAX registers;
// a group of one still has two independent accumulator chains when it feeds two GEMMs)
template <int MTX, bool ONE, int GMAX = 3>
__device__ __forceinline__ void mma_b16(const float* __restrict__ As, const WFragB1& f1, Acc<MTX>& acc1, const WFragB1& f2,
                                        Acc<MTX>& acc2, int mt) {
#pragma unroll
    for (int m0 = 0; m0 < MTX; m0 += GMAX) {
        const int left = mt - m0;
        if (left <= 0) break;
        if constexpr (GMAX >= 3) {
            if (m0 + 3 <= MTX && left >= 3) {
                mma_b16_group<3, MTX, ONE>(As, m0, f1, acc1, f2, acc2);
                continue;
            }
        }
        if constexpr (GMAX >= 2) {
            if (m0 + 2 <= MTX && left >= 2) {
                mma_b16_group<2, MTX, ONE>(As, m0, f1, acc1, f2, acc2);
                continue;
            }
        }
        mma_b16_group<1, MTX, ONE>(As, m0, f1, acc1, f2, acc2);
    }
}

// ---- piece planes: the split done ONCE per workgroup ------------------------------------------------------------------
// With the fp32 tile in LDS every wave splits every A fragment it reads: 36 VALU per 16 x 32 fragment and wave, eight
// times over in an 8-wave workgroup -- measured (profiles/r04_issue_slots_pmc.txt) 9 VALU per MFMA in the fused global-edge
// backward, two thirds of them these splits, and the SIMDs' issue slots -- not the matrix pipe, not HBM -- are what the
// multi-chunk kernels run out of.  Here the thread that WRITES a tile element (the cooperative sweep: a float4 of one row)
// splits it and stores the three bf16 pieces in MFMA A-fragment order; a reader's fragment is then three ds_read_b128 and
// no VALU.  A 16-row tile is 12 KB: [piece][k-step q][chunk] x 16 bytes, chunk = the lane that reads it, swizzled
//     chunk(rho, kg, q) = (rho + 16 kg) ^ (2 kg + (q & 1))        (row rho = l & 15, k-group kg = l >> 4)
// so that both sides are conflict-free: a ds_read_b128 group ({rows 0-3, 12-15} of one kg with {rows 4-11} of the next)
// still covers 16 distinct 16-byte bank groups (the xor keeps each of the two row sets in place or swaps them), and the 16
// lanes of a ds_write_b64 group -- one row, float4 columns 0..15 or 16..31: four kg x two q x two halves -- land on 16
// distinct 8-byte bank pairs (the low three bits of the chunk differ for every (kg, q & 1)).
constexpr int PTILE = 3 * 4 * 1024;           // bytes of a 16-row tile as piece planes
__device__ __forceinline__ int piece_chunk(int rho, int kg, int q) { return (rho + 16 * kg) ^ (2 * kg + (q & 1)); }
// row r (of the chunk: tile r >> 4), float4 column c4 of a [rows][128] tile -> its 8 bytes in each piece plane
__device__ __forceinline__ void st_pieces4(char* __restrict__ P, int r, int c4, const float4& v) {
    const int q = c4 >> 3, kg = (c4 >> 1) & 3;
    char* d = P + (r >> 4) * PTILE + q * 1024 + piece_chunk(r & 15, kg, q) * 16 + (c4 & 1) * 8;
    uint32_t a0, b0, c0, a1, b1, c1;
#ifdef PAMNET_SPLIT_PROBE
    // Development probe (never in libpamnet_hip.so; tools/agg_probe.py with PAMNET_PROBE_FLAGS=-DPAMNET_SPLIT_PROBE): what the
    // kernels would cost if the pieces arrived ready-made -- one conversion instead of the exact three-piece split (WRONG
    // numbers, right instruction count for an upper bound on "split the edge embeddings once per step").
    a0 = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2{v.x, v.y}), bf16x2));
    a1 = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2{v.z, v.w}), bf16x2));
    b0 = c0 = a0, b1 = c1 = a1;
#else
    split3(v.x, v.y, a0, b0, c0);
    split3(v.z, v.w, a1, b1, c1);
#endif
    *reinterpret_cast<uint2*>(d) = make_uint2(a0, a1);
    *reinterpret_cast<uint2*>(d + 4096) = make_uint2(b0, b1);
    *reinterpret_cast<uint2*>(d + 8192) = make_uint2(c0, c1);
}
// A fragment (rows 16 m + (l & 15), k = 32 q + 8 (l >> 4) + 0..7) of a piece-plane tile: three 16-byte reads
__device__ __forceinline__ Frag3 lds_frag3p(const char* __restrict__ P, int m, int q) {
    const int lane = threadIdx.x & 63;
    const char* s = P + m * PTILE + q * 1024 + piece_chunk(lane & 15, lane >> 4, q) * 16;
    Frag3 f;
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) {
        const uint4 u = *reinterpret_cast<const uint4*>(s + pc * 4096);
        f.p[pc][0] = u.x, f.p[pc][1] = u.y, f.p[pc][2] = u.z, f.p[pc][3] = u.w;
    }
    return f;
}
template <int G, int MTX, bool ONE>
__device__ __forceinline__ void mma_p16_group(const char* __restrict__ P, int m0, const WFragB1& f1, Acc<MTX>& acc1,
                                              const WFragB1& f2, Acc<MTX>& acc2) {
#pragma unroll
    for (int q = 0; q < DIM / 32; ++q) {
        Frag3 a[G];
#pragma unroll
        for (int g = 0; g < G; ++g) a[g] = lds_frag3p(P, m0 + g, q);
        mfma6<G, MTX>(a, f1.p[q], acc1, m0);
        if constexpr (!ONE) mfma6<G, MTX>(a, f2.p[q], acc2, m0);
    }
}
// mma_b16 on piece planes (same grouping: G <= GMAX row tiles share a weight fragment's issue)
template <int MTX, bool ONE, int GMAX = 3>
__device__ __forceinline__ void mma_p16(const char* __restrict__ P, const WFragB1& f1, Acc<MTX>& acc1, const WFragB1& f2,
                                        Acc<MTX>& acc2, int mt) {
#pragma unroll
    for (int m0 = 0; m0 < MTX; m0 += GMAX) {
        const int left = mt - m0;
        if (left <= 0) break;
        if constexpr (GMAX >= 3) {
            if (m0 + 3 <= MTX && left >= 3) {
                mma_p16_group<3, MTX, ONE>(P, m0, f1, acc1, f2, acc2);
                continue;
            }
        }
        if constexpr (GMAX >= 2) {
            if (m0 + 2 <= MTX && left >= 2) {
                mma_p16_group<2, MTX, ONE>(P, m0, f1, acc1, f2, acc2);
                continue;
            }
        }
        mma_p16_group<1, MTX, ONE>(P, m0, f1, acc1, f2, acc2);
    }
}

// A workgroup is NW waves (8 or 4); wave w owns NS = 8 / NW consecutive 16-column slices.  Two 4-wave workgroups with
// half the rows each share a CU: same waves per SIMD as one 8-wave workgroup, but their barriers are independent, so one
// workgroup's load / epilogue sweeps overlap the other's GEMMs (measured -16 % on the triplet/pair MLP).
// NW*64 threads sweep a [16 mt][128] tile: thread t owns float4 column t & 31 of rows (t >> 5) + 2 NW i.
// f(row_in_chunk, c4); global accesses inside f are 512-byte coalesced rows.
template <int MTX, int NW, typename F>
__device__ __forceinline__ void sweep(int mt, F&& f) {
    const int c4 = threadIdx.x & 31, r0 = threadIdx.x >> 5;
    constexpr int RPP = 2 * NW;                               // rows per pass
#pragma unroll
    for (int i = 0; i < 16 * MTX / RPP; ++i)
        if (RPP * i < 16 * mt) f(r0 + RPP * i, c4);
}

template <int NS>
struct WSet {                                                 // a wave's NS slices of one weight matrix
    WFrag1 s[NS];
};
template <>
struct WSet<0> {};                                            // (placeholder where a kernel holds bf16x3 pieces instead)
template <bool TRANS, int NS>
__device__ __forceinline__ void load_wset(WSet<NS>& w, const float* __restrict__ W, int ldw, int wc) {
#pragma unroll
    for (int h = 0; h < NS; ++h) load_wfrag1<TRANS>(w.s[h], W, ldw, wc + 16 * h);
}
template <int MTX, int NS>
struct AccSet {
    Acc<MTX> a[NS];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int h = 0; h < NS; ++h) a[h].zero();
    }
};
template <int MTX, int NS>
__device__ __forceinline__ void mma_set(const float* __restrict__ As, const WSet<NS>& w, AccSet<MTX, NS>& acc, int mt) {
#pragma unroll
    for (int h = 0; h < NS; ++h) mma_n<MTX>(As, w.s[h], acc.a[h], mt);
}
template <int NS>
struct BiasSet {
    float v[NS];
};
template <int NS>
__device__ __forceinline__ BiasSet<NS> lane_biases(const float* __restrict__ b, int wc) {
    BiasSet<NS> r;
#pragma unroll
    for (int h = 0; h < NS; ++h) r.v[h] = b ? b[wc + 16 * h + (threadIdx.x & 15)] : 0.f;
    return r;
}
template <int MTX, int NS>
__device__ __forceinline__ void store_set(const AccSet<MTX, NS>& acc, float* __restrict__ Ds, int wc, const BiasSet<NS>& b,
                                          int mt, int tstride = 16 * LDT) {
#pragma unroll
    for (int h = 0; h < NS; ++h) acc_store<MTX>(acc.a[h], Ds, wc + 16 * h, b.v[h], mt, tstride);
}

// rows [beg, end) of this workgroup and its chunking: per = 16-row tiles per workgroup, cmt = tiles per chunk
// workgroup b owns base (+1 for the first `rem` workgroups) consecutive 16-row tiles
// NW = 4 (paired split, two co-resident workgroups per CU): "CU" c owns pa consecutive tiles; workgroup c takes the first
// pc of them, workgroup pb + c the rest (pb = number of pairs).
struct Span {
    int64_t beg, end;
    int cmt;
    template <int NW>
    static __device__ __forceinline__ Span make(int64_t m, int pa, int pb, int pc, int cmt_) {
        return make_at<NW>(m, pa, pb, pc, cmt_, (int)blockIdx.x);
    }
    // the same for workgroup `b` of a plan that is one part of a launch (edge_chain.hip local_bwd_pair_kernel)
    template <int NW>
    static __device__ __forceinline__ Span make_at(int64_t m, int pa, int pb, int pc, int cmt_, int b) {
        Span sp;
        sp.cmt = cmt_;
        int64_t t0, cnt;
        if (NW == 8) {                                        // pa = base, pb = rem
            t0 = (int64_t)b * pa + (b < pb ? b : pb);
            cnt = pa + (b < pb ? 1 : 0);
        } else {                                              // pa = tiles per pair, pb = pairs, pc = first share
            const bool second = b >= pb;
            const int c = second ? b - pb : b;
            t0 = (int64_t)c * pa + (second ? pc : 0);
            cnt = second ? pa - pc : pc;
        }
        sp.beg = t0 * 16;
        const int64_t e = (t0 + cnt) * 16;
        sp.end = e < m ? e : m;
        if (sp.beg > sp.end) sp.beg = sp.end;
        return sp;
    }
};
#define CHUNK_LOOP(sp)                                                             \
    for (int64_t row0 = (sp).beg; row0 < (sp).end; row0 += (int64_t)(sp).cmt * 16)
__device__ __forceinline__ int chunk_mt(const Span& sp, int64_t row0) {
    const int64_t left = sp.end - row0;
    const int rows = left < (int64_t)sp.cmt * 16 ? (int)left : sp.cmt * 16;
    return (rows + 15) >> 4;
}

template <int NW>
__device__ __forceinline__ int wave_col() { return (threadIdx.x >> 6) * (128 / NW); }
__device__ __forceinline__ float lane_bias(const float* __restrict__ b, int wc) {
    return b ? b[wc + (threadIdx.x & 15)] : 0.f;
}


// -------------------------------------------------------------------------------------------------- 2-layer MLP
// y = SiLU(W2 SiLU(W1 x + b1) + b2) on the rows of `sp` (layers/local_message_passing.py:49 `mlp_sbf`); z1, z2: optional
// saves.  Called by mlp2_fwd_kernel (edge_chain.hip) and, as a rider, by node_tail_fwd_kernel (node_tail.hip): the
// triplet/pair MLPs do not depend on the node features, so their row tiles fill the CUs a node chain leaves idle.
struct Mlp2Set {
    const float *W1, *b1, *W2, *b2;
    float *z1, *z2, *y;
};
// Both GEMMs run on the bf16 matrix pipe at fp32 accuracy ("bf16x6" above).  8-wave geometry: a wave's slice of each
// matrix stays resident as bf16x3 pieces.  4-wave geometry (two slices per wave; the riders of the node-chain launches):
// the pieces of ONE matrix at a time -- a wave's two slices share the split of every A fragment, and a matrix is re-read
// (16 KB per wave, L2-resident) and re-split per chunk, ~1 us against the ~5 us of matrix-pipe time a 48-row chunk saves.
constexpr int MLP2_TILE_B = PTILE + 16 * LDT * 4;            // LDS bytes per 16-row tile of a chunk: piece planes + one fp32 tile
template <int MTX, int NW>
__device__ __forceinline__ void mlp2_fwd_body(const float* __restrict__ x, const Mlp2Set& set, const Span& sp, float* lds) {
    const float* __restrict__ W1 = set.W1;
    const float* __restrict__ b1 = set.b1;
    const float* __restrict__ W2 = set.W2;
    const float* __restrict__ b2 = set.b2;
    float* __restrict__ z1 = set.z1;
    float* __restrict__ z2 = set.z2;
    float* __restrict__ y = set.y;
    // Both GEMM inputs wait in LDS as piece planes written by the sweep that produces them (the rows as they arrive, then
    // SiLU(z1)): the split once per workgroup ("piece planes" above); the accumulators pass through one fp32 tile array.
    char* P = reinterpret_cast<char*>(lds);
    float* S1 = reinterpret_cast<float*>(P + MTX * PTILE);
    constexpr int NS = 8 / NW;
    const int wc = wave_col<NW>();
    const BiasSet<NS> bv1 = lane_biases<NS>(b1, wc), bv2 = lane_biases<NS>(b2, wc);
    WFragB1 r1, r2;                                           // NS == 1: resident pieces of both matrices
    if constexpr (NS == 1) {
        load_wfragb1<false>(r1, W1, DIM, wc);
        load_wfragb1<false>(r2, W2, DIM, wc);
    }
    auto gemm = [&](const float* W, const WFragB1& resident, AccSet<MTX, NS>& acc, int mt) {
        acc.zero();
        if constexpr (NS == 1) {
            mma_p16<MTX, true, 3>(P, resident, acc.a[0], resident, acc.a[0], mt);
        } else {
            WFragB1 wa, wb;
            load_wfragb1<false>(wa, W, DIM, wc);
            load_wfragb1<false>(wb, W, DIM, wc + 16);
            mma_p16<MTX, false, 3>(P, wa, acc.a[0], wb, acc.a[NS - 1], mt);
        }
    };
    CHUNK_LOOP(sp) {
        const int mt = chunk_mt(sp, row0);
        sweep<MTX, NW>(mt, [&](int r, int c4) { st_pieces4(P, r, c4, ldg4z(x, row0 + r, sp.end, DIM, c4)); });
        __syncthreads();
        AccSet<MTX, NS> acc;
        gemm(W1, r1, acc, mt);
        store_set<MTX, NS>(acc, S1, wc, bv1, mt);
        __syncthreads();                                      // (every wave is done with the x pieces: they take SiLU(z1))
        sweep<MTX, NW>(mt, [&](int r, int c4) {
            const int64_t g = row0 + r;
            const float4 zz = lds4(S1, r, c4);
            st_pieces4(P, r, c4, f4silu(zz));
            if (z1 && g < sp.end) stg4_nt(z1, g, DIM, c4, zz);       // (backward-only saves: streamed past the caches)
        });
        __syncthreads();
        gemm(W2, r2, acc, mt);
        store_set<MTX, NS>(acc, S1, wc, bv2, mt);
        __syncthreads();
        sweep<MTX, NW>(mt, [&](int r, int c4) {
            const int64_t g = row0 + r;
            if (g >= sp.end) return;
            const float4 zz = lds4(S1, r, c4);
            if (z2) stg4_nt(z2, g, DIM, c4, zz);
            stg4(y, g, DIM, c4, f4silu(zz));
        });
        // (no barrier here: the next chunk's first sweep writes the pieces, which every wave finished reading before the
        // barrier above, and its accumulators reach S1 only behind that sweep's own barrier)
    }
}

}  // namespace edge
}  // namespace
