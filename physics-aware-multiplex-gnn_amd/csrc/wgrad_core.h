// Split-K weight-gradient core shared by wgrad.hip (the batched launches) and node_tail.hip (slots of a batch riding as
// extra workgroups of a node-chain backward launch: the chain occupies only ceil(n/16) of the 256 CUs).
//   dW[128,128] = dZ[rows,128]^T * A[rows,128],  db[128] = column sums of dZ
// A workgroup ("slot") reduces a contiguous row chunk of one job into a full 128x128 fp32 tile and writes it to its
// slot of a partial buffer; finish_body sums each job's slots in a fixed order -> deterministic, atomics-free.
//
// The GEMM runs on the bf16 matrix pipe at fp32 accuracy ("bf16x6", gemm_core.h): both operands are split exactly into
// three bf16 pieces, six v_mfma_f32_16x16x32_bf16 per 32 rows and tile (the row index is the MFMA k dimension).
//   * The split happens ONCE per element, on the way from global memory to LDS (a reader-side split repeats it in every
//     wave that shares the fragment and made the loop VALU-issue bound: 7 900 cycles per 64 rows against 10 500 for the
//     fp32 MFMA form, tools/wgrad_probe.py).  LDS holds the pieces as ready-made MFMA fragments: per operand, piece and
//     16-column tile one 1 KB image of 64 lane slots x 16 bytes (8 consecutive rows of one column).
//   * Slot order inside an image: column c, row group kg -> slot (c % 4) * 4 + c / 4 + 16 kg.  A staging thread owns
//     8 rows x 4 adjacent columns (eight coalesced 16-byte global loads), so its neighbours in the wave write adjacent
//     slots; images are 64 bytes apart modulo the bank width (TILE_B = 1088), so the two images an 8-lane store group
//     touches do not collide; the reader's 16-lane groups of ds_read_b128 each cover 16 distinct slots modulo 16.
//   * One block = 32 rows = one MFMA k-step.  Per block: fragments -> registers, barrier (the images are free from here
//     on), then the MFMAs with the staging of the NEXT block's rows (already in registers) placed between them by hand,
//     one small unit per MFMA, fenced by scheduling barriers: SiLU of a value, one stage of the split of a row pair
//     (4 / 4 / 1 VALU), a column's share of the bias sum, the 16-byte stores of a finished column.  A second barrier
//     publishes the images.  A single wave issues one instruction per ~4.5 cycles whatever its kind and VALU work does
//     not hide behind an MFMA of the same wave unless it is spread thinly (measured: 96 MFMAs 2 016 cycles, with 48
//     split units bunched behind every second MFMA 2 660), so the loop keeps ~2.5 instructions per MFMA, evenly.
//   * <= 256 registers (accumulators included) so that two workgroups share a CU: the small reduction workgroups of the
//     fused launches (wgrad.hip) run beside the slots, not after them.
#pragma once
#include <type_traits>
#include <utility>

#include "common.h"
#include "gemm_core.h"

#ifndef WPROBE
#define WPROBE(i)
#endif

namespace {
using namespace pamnet;

// f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N - 1>): a loop whose index is a constant
// expression inside the body
template <typename F, int... T>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, T...>) {
    (f(std::integral_constant<int, T>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

constexpr int RB = 64;            // slot chunks are multiples of this many rows (host-side planning, wgrad.hip)
constexpr int KB = 32;            // rows per block = one k-step of v_mfma_f32_16x16x32_bf16
constexpr int TILE_B = 1024 + 64; // bytes per fragment image (+ 64: consecutive images start 16 banks apart)
constexpr int PLANES_B = 2 * 3 * 8 * TILE_B;                       // {dZ, A} x 3 pieces x 8 column tiles = 52 224 bytes
constexpr int WGRAD_LDS_FLOATS = DIM * LDT + 4 * DIM;              // the epilogue's transposed tile + bias parts (69 632 B)
static_assert(WGRAD_LDS_FLOATS * 4 >= PLANES_B, "the fragment images alias the epilogue tile");

// zero rows for the ragged end of a job: out-of-range rows are read from here instead of being masked after the load
__device__ const float4 pamnet_wgrad_zero_row[DIM / 4] = {};

struct WJob {
    const float* dZ;
    const float* A;
    float* dW;
    float* db;        // may be null
    int64_t rows;
    int ld_dz, ld_a, ld_dw, a_mode;
};
template <int MJ>
struct WBatchT {
    WJob job[MJ];
    int start[MJ + 1];            // slot prefix: job j owns slots [start[j], start[j+1])
    int njobs;
};
constexpr int MAXJ = 24, MAXJ_S = 16;
using WBatch = WBatchT<MAXJ>;     // stand-alone launches (up to 24 jobs)
using WBatchS = WBatchT<MAXJ_S>;  // compact form: three of them fit the 4 KB kernel-argument block (fused launches, riders)

// A planned rider batch (host side): slots that run as extra workgroups of a node-chain backward launch
// (pamnet_wgrad_rider_plan_f32 fills it, pamnet_node_pre_tail_bwd_f32 launches it, the deferred finish reduces it).
struct WgradRider {
    WBatchS batch;
    float* partial;
    int slots;
};

__device__ __forceinline__ float f4get(const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }
__device__ __forceinline__ void f4set(float4& v, int e, float x) {
    if (e == 0) v.x = x;
    else if (e == 1) v.y = x;
    else if (e == 2) v.z = x;
    else v.w = x;
}

// Staging roles of a wave (uniform): what it does with its 8 rows x 4 columns between the MFMAs of a block.
constexpr int ROLE_NONE = 0, ROLE_DZ = 1, ROLE_A = 2, ROLE_A_SILU = 3;
// Units per column e of the thread's float4 columns, in order:
//   ROLE_A_SILU: 8 x SiLU of one value;  all: 4 row pairs x 3 split stages;  ROLE_DZ: the column's bias sum;
//   all: 3 x 16-byte store (one per piece)
template <int ROLE>
struct Units {
    static constexpr int silu = ROLE == ROLE_A_SILU ? 8 : 0, bias = ROLE == ROLE_DZ ? 1 : 0;
    static constexpr int per_col = silu + 12 + bias + 3, total = ROLE == ROLE_NONE ? 0 : 4 * per_col;
};

struct Stage {
    float4 raw[8];                // the thread's 8 rows x 4 columns of the next block
    uint32_t pk[3][4];            // pieces of the column being split: [piece][row pair]
    f32x2 resid;
    double bsum[4];               // bias gradient: fp32 inside a block (8 rows), fp64 across blocks
    char* wr;                     // image address of column 0, piece 0 (column e: + 64 e, piece p: + 8 p TILE_B)
};

template <int ROLE, int U>
__device__ __forceinline__ void stage_unit(Stage& st) {
    using Un = Units<ROLE>;
    constexpr int e = U / Un::per_col, v = U % Un::per_col;
    if constexpr (v < Un::silu) {
        f4set(st.raw[v], e, silu(f4get(st.raw[v], e)));       // A = SiLU(Z_prev) applied while staging (a_mode 1)
    } else if constexpr (v < Un::silu + 12) {
        constexpr int k2 = (v - Un::silu) / 3, stg = (v - Un::silu) % 3;
        split3_stage<stg>(f4get(st.raw[2 * k2], e), f4get(st.raw[2 * k2 + 1], e), st.pk[0][k2], st.pk[1][k2], st.pk[2][k2],
                          st.resid);
    } else if constexpr (v < Un::silu + 12 + Un::bias) {
        float t = ((f4get(st.raw[0], e) + f4get(st.raw[1], e)) + (f4get(st.raw[2], e) + f4get(st.raw[3], e))) +
                  ((f4get(st.raw[4], e) + f4get(st.raw[5], e)) + (f4get(st.raw[6], e) + f4get(st.raw[7], e)));
        st.bsum[e] += (double)t;
    } else {
        constexpr int p = v - (Un::silu + 12 + Un::bias);
        *reinterpret_cast<uint4*>(st.wr + p * 8 * TILE_B + e * 64) = make_uint4(st.pk[p][0], st.pk[p][1], st.pk[p][2], st.pk[p][3]);
    }
}

// Slot `bid` of `batch`, computed by a workgroup of NW waves (4: wave tile 64x64; 8: 32x64).  The first four waves stage.
template <int NW, typename Batch>
__device__ __forceinline__ void wgrad_body(const Batch& batch, float* __restrict__ partial, const int bid, float* lds) {
    constexpr int NT = 64 * NW;               // threads
    constexpr int AI = NW == 4 ? 4 : 2;       // 16-row tiles of dW rows per wave
    constexpr int NMFMA = AI * 4 * 6;         // MFMAs per block and wave
    char* ldsb = reinterpret_cast<char*>(lds);
    int j = 0;
    while (j + 1 < batch.njobs && bid >= batch.start[j + 1]) ++j;       // wave-uniform scalar search
    const WJob jb = batch.job[j];
    const int s = bid - batch.start[j];
    const int js = batch.start[j + 1] - batch.start[j];
    const int64_t chunk = ((jb.rows + js - 1) / js + RB - 1) / RB * RB;
    const int64_t beg = (int64_t)s * chunk;
    const int64_t end = beg + chunk < jb.rows ? beg + chunk : jb.rows;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r16 = lane & 15, kg = lane >> 4;
    const int i0 = (w >> 1) * (16 * AI), j0 = (w & 1) * 64;   // wave tile: dW rows [i0, i0 + 16 AI), cols [j0, j0 + 64)
    f32x4 acc[AI][4];
#pragma unroll
    for (int a = 0; a < AI; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- staging role (threads 0..255): operand `op` (0: dZ, 1: A), rows 8 g .. 8 g + 7 of a block, columns 4 c4 .. + 3
    const bool stager = threadIdx.x < 256;                     // wave-uniform
    const int op = (threadIdx.x >> 7) & 1, c4 = threadIdx.x & 31, g = (threadIdx.x >> 5) & 3;
    const int role = !stager ? ROLE_NONE : op == 0 ? ROLE_DZ : jb.a_mode == 1 ? ROLE_A_SILU : ROLE_A;
    const float* src = op ? jb.A : jb.dZ;
    const int64_t ld = op ? jb.ld_a : jb.ld_dz;
    const float* zero = reinterpret_cast<const float*>(pamnet_wgrad_zero_row);
    Stage st;
    st.bsum[0] = st.bsum[1] = st.bsum[2] = st.bsum[3] = 0.0;
    // image address of this thread's column e, piece p: ((op * 3 + p) * 8 + c4 / 4) * TILE_B + (e * 4 + c4 % 4 + 16 g) * 16
    st.wr = ldsb + (op * 24 + (c4 >> 2)) * TILE_B + ((c4 & 3) + 16 * g) * 16;
    // reader: piece p of tile ct of operand o at ((o * 3 + p) * 8 + ct) * TILE_B + ((r16 % 4) * 4 + r16 / 4 + 16 kg) * 16
    const char* rdz = ldsb + (i0 >> 4) * TILE_B + ((r16 & 3) * 4 + (r16 >> 2) + 16 * kg) * 16;
    const char* rda = ldsb + (24 + (j0 >> 4)) * TILE_B + ((r16 & 3) * 4 + (r16 >> 2) + 16 * kg) * 16;

    auto fetch = [&](int64_t r0) {            // rows r0 + 8 g + u of the thread's operand; rows past the end read zeros
        const float* p = src + (r0 + 8 * g) * ld + 4 * c4;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool ok = r0 + 8 * g + u < end;
            st.raw[u] = *reinterpret_cast<const float4*>(ok ? p + u * ld : zero + 4 * c4);
        }
    };
    auto stage_all = [&](auto role_c) {       // first block of a slot: nothing to hide behind
        constexpr int ROLE = decltype(role_c)::value;
        static_for<Units<ROLE>::total>([&](auto uc) { stage_unit<ROLE, decltype(uc)::value>(st); });
    };

    // One block: fragments -> registers, barrier, MFMAs with the staging units of the next block's rows between them.
    auto block = [&](auto role_c) {
        constexpr int ROLE = decltype(role_c)::value;
        uint4 fz[AI][3], fa[4][3];
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int p = 0; p < 3; ++p) fa[b][p] = *reinterpret_cast<const uint4*>(rda + (p * 8 + b) * TILE_B);
#pragma unroll
        for (int a = 0; a < AI; ++a)
#pragma unroll
            for (int p = 0; p < 3; ++p) fz[a][p] = *reinterpret_cast<const uint4*>(rdz + (p * 8 + a) * TILE_B);
        __syncthreads();                                       // every wave holds its fragments: the images are free
        // units start a sixth of the way in (the rows were requested at the end of the previous block), spread evenly
        constexpr int U0 = NMFMA / 6, USPAN = NMFMA - U0, NU = Units<ROLE>::total;
        __builtin_amdgcn_sched_barrier(0);
        static_for<NMFMA>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            // MFMA order: tile row a outermost, then the six products smallest first, then the four column tiles --
            // consecutive MFMAs go to different accumulators
            constexpr int a = i / 24, term = (i % 24) / 4, b = i % 4;
            constexpr int pz = term == 0 ? 2 : (term == 1 || term == 3) ? 1 : 0;
            constexpr int pa = term == 2 ? 2 : (term == 1 || term == 4) ? 1 : 0;
            const u32x4 zv = {fz[a][pz].x, fz[a][pz].y, fz[a][pz].z, fz[a][pz].w};
            const u32x4 av = {fa[b][pa].x, fa[b][pa].y, fa[b][pa].z, fa[b][pa].w};
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, zv), __builtin_bit_cast(bf16x8, av),
                                                                acc[a][b], 0, 0, 0);
            if constexpr (NU > 0 && i >= U0) {
                constexpr int u0 = ((i - U0) * NU + USPAN - 1) / USPAN, u1 = ((i - U0 + 1) * NU + USPAN - 1) / USPAN;
                static_for<u1 - u0>([&](auto uc) { stage_unit<ROLE, u0 + decltype(uc)::value>(st); });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    auto with_role = [&](auto&& f) {          // wave-uniform dispatch to the role's instantiation
        if (role == ROLE_DZ) f(std::integral_constant<int, ROLE_DZ>{});
        else if (role == ROLE_A) f(std::integral_constant<int, ROLE_A>{});
        else if (role == ROLE_A_SILU) f(std::integral_constant<int, ROLE_A_SILU>{});
        else f(std::integral_constant<int, ROLE_NONE>{});
    };

    int it = 0;
    if (beg < end) {
        WPROBE(0);
        if (stager) {
            fetch(beg);
            with_role(stage_all);
            if (beg + KB < end) fetch(beg + KB);
        }
        __syncthreads();
        WPROBE(1);
        for (int64_t r0 = beg; r0 < end; r0 += KB, ++it) {
            WPROBE(2 + 3 * it);
            if (stager && r0 + KB < end) {
                with_role(block);
                WPROBE(3 + 3 * it);
                if (r0 + 2 * KB < end) fetch(r0 + 2 * KB);
            } else {
                block(std::integral_constant<int, ROLE_NONE>{});
                WPROBE(3 + 3 * it);
            }
            WPROBE(4 + 3 * it);
            __syncthreads();
        }
    }
    WPROBE(2 + 3 * it);
    // partial[slot][128*128 + 2*128]: the tile, then two row-part bias partials
    // The accumulator layout (4 rows x 16 columns per store) would hit memory as 64-byte fragments; transpose through
    // LDS (the fragment images are free now) and write the tile as coalesced 512-byte rows.
    float* out = partial + (int64_t)bid * (DIM * DIM + 2 * DIM);
    float* T = lds;
#pragma unroll
    for (int a = 0; a < AI; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                T[(i0 + 16 * a + kg * 4 + r) * LDT + j0 + 16 * b + r16] = acc[a][b][r];
    float* bp = lds + DIM * LDT;                               // [4 row groups][128] bias parts behind the tile
    if (stager && op == 0) {
        *reinterpret_cast<float4*>(bp + g * DIM + 4 * c4) =
            make_float4((float)st.bsum[0], (float)st.bsum[1], (float)st.bsum[2], (float)st.bsum[3]);
    }
    __syncthreads();
    {
        const int oc4 = threadIdx.x & 31, rr = threadIdx.x >> 5;
        constexpr int RP = NT / 32;
#pragma unroll
        for (int i = 0; i < DIM / RP; ++i) {
            const int row = rr + RP * i;
            *reinterpret_cast<float4*>(out + row * DIM + 4 * oc4) = *reinterpret_cast<const float4*>(T + row * LDT + 4 * oc4);
        }
    }
    if (threadIdx.x < 2 * DIM) {                               // 4 row groups -> the 2 parts the finish pass expects
        const int h = threadIdx.x >> 7, bc = threadIdx.x & 127;
        out[DIM * DIM + threadIdx.x] = bp[(2 * h) * DIM + bc] + bp[(2 * h + 1) * DIM + bc];
    }
    WPROBE(3 + 3 * it);
}

}  // namespace
