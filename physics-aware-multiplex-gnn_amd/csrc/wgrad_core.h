// Split-K weight-gradient core shared by wgrad.hip (the batched launches) and node_tail.hip (slots of a batch riding as
// extra workgroups of a node-chain backward launch: the chain occupies only ceil(n/16) of the 256 CUs).
//   dW[128,128] = dZ[rows,128]^T * A[rows,128],  db[128] = column sums of dZ
// A workgroup ("slot") reduces a contiguous row chunk of one job into a full 128x128 fp32 tile with
// v_mfma_f32_16x16x4_f32 (the row index is the MFMA k dimension) and writes it to its slot of a partial buffer;
// finish_body sums each job's slots in a fixed order -> deterministic, atomics-free.
#pragma once
#include "common.h"
#include "gemm_core.h"

#ifndef WPROBE
#define WPROBE(i)
#endif

namespace {
using namespace pamnet;

constexpr int RB = 64;            // rows staged per step (2 x 36 KB LDS: 64 rows of dZ and A in flight per fetch)
constexpr int LDW = 144;          // LDS leading dim: 144 mod 32 = 16 -> conflict-free ds_read_b32 fragment reads
constexpr int WGRAD_LDS_FLOATS = 2 * RB * LDW;

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

// Slot `bid` of `batch`, computed by a workgroup of NW waves (4: wave tile 64x64; 8: 32x64).
template <int NW, typename Batch>
__device__ __forceinline__ void wgrad_body(const Batch& batch, float* __restrict__ partial, const int bid, float* lds) {
    constexpr int NT = 64 * NW;               // threads
    constexpr int RP = NT / 32;               // rows fetched per pass (float4 column per lane, 32 lanes per row)
    constexpr int NP = RB / RP;               // passes per 64-row step
    constexpr int AI = NW == 4 ? 4 : 2;       // 16-row tiles of dW rows per wave
    constexpr int NH = NT / 128;              // row parts of the bias column sums
    float* Zs = lds;
    float* As = lds + RB * LDW;
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
    // bias gradient: thread t = (column t & 127, row part t >> 7); fp32 inside a 64 / NH-row block, fp64 across blocks
    double colsum = 0.0;
    const int bc = threadIdx.x & 127, bh = threadIdx.x >> 7;
    const bool want_bias = jb.db != nullptr;

    // register double buffer: the next 64-row block is in flight from L2/HBM while the MFMAs chew on the current one
    const int c4 = threadIdx.x & 31, rr = threadIdx.x >> 5;
    float4 zr[NP], ar[NP];
    auto fetch = [&](int64_t r0) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int64_t g = r0 + rr + RP * i;
            const bool ok = g < end;
            const int64_t gg = ok ? g : beg;                  // clamp instead of branching: loads stay unconditional
            zr[i] = ldg4(jb.dZ, gg, jb.ld_dz, c4);
            ar[i] = ldg4(jb.A, gg, jb.ld_a, c4);
            if (!ok) { zr[i] = f4zero(); ar[i] = f4zero(); }
        }
    };
    if (beg < end) fetch(beg);
    int it = 0;
    for (int64_t r0 = beg; r0 < end; r0 += RB, ++it) {
        WPROBE(4 * it);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int r = rr + RP * i;
            float4 a = ar[i];
            if (jb.a_mode == 1) a = f4silu(a);                 // SiLU(0) = 0 keeps the zero padding
            *reinterpret_cast<float4*>(Zs + r * LDW + 4 * c4) = zr[i];
            *reinterpret_cast<float4*>(As + r * LDW + 4 * c4) = a;
        }
        __syncthreads();
        WPROBE(4 * it + 1);
        if (r0 + RB < end) fetch(r0 + RB);
        if (want_bias) {
            constexpr int RH = RB / NH;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
            for (int r = 0; r < RH; r += 4) {
                s0 += Zs[(RH * bh + r) * LDW + bc];
                s1 += Zs[(RH * bh + r + 1) * LDW + bc];
                s2 += Zs[(RH * bh + r + 2) * LDW + bc];
                s3 += Zs[(RH * bh + r + 3) * LDW + bc];
            }
            colsum += (double)((s0 + s1) + (s2 + s3));
        }
        // operands of k-step st+1 are requested before the MFMAs of step st are issued (explicit register double
        // buffer): left alone, the compiler issues each ds_read right before its s_waitcnt and the LDS latency shows up
        // twice per k-step (12 400 instead of 8 192 cycles per 64-row block, tools/wgrad_probe.py)
        float za[2][AI], ab[2][4];
#pragma unroll
        for (int t = 0; t < AI; ++t) za[0][t] = Zs[kg * LDW + i0 + 16 * t + r16];
#pragma unroll
        for (int t = 0; t < 4; ++t) ab[0][t] = As[kg * LDW + j0 + 16 * t + r16];
#pragma unroll
        for (int st = 0; st < RB / 4; ++st) {
            const int cur = st & 1, nxt = cur ^ 1;
            if (st + 1 < RB / 4) {
                const int r = 4 * (st + 1) + kg;
#pragma unroll
                for (int t = 0; t < AI; ++t) za[nxt][t] = Zs[r * LDW + i0 + 16 * t + r16];
#pragma unroll
                for (int t = 0; t < 4; ++t) ab[nxt][t] = As[r * LDW + j0 + 16 * t + r16];
            }
            __builtin_amdgcn_sched_barrier(0);                 // keep the requests above this step's MFMAs
#pragma unroll
            for (int a = 0; a < AI; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(za[cur][a], ab[cur][b], acc[a][b], 0, 0, 0);
        }
        WPROBE(4 * it + 2);
        __syncthreads();
        WPROBE(4 * it + 3);
    }
    WPROBE(4 * it);
    // partial[slot][128*128 + 2*128]: the tile, then two row-part bias partials
    // The accumulator layout (4 rows x 16 columns per store) would hit memory as 64-byte fragments; transpose through
    // LDS (the staging buffers are free now: 128 x 132 floats fit) and write the tile as coalesced 512-byte rows.
    float* out = partial + (int64_t)bid * (DIM * DIM + 2 * DIM);
    float* T = lds;
    static_assert(2 * RB * LDW >= DIM * LDT + 4 * DIM, "tile (+ bias parts) must fit in the staging buffers");
#pragma unroll
    for (int a = 0; a < AI; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                T[(i0 + 16 * a + kg * 4 + r) * LDT + j0 + 16 * b + r16] = acc[a][b][r];
    float* bp = lds + DIM * LDT;                               // [NH][128] bias parts behind the tile
    if (NH > 2) bp[bh * DIM + bc] = (float)colsum;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < DIM / RP; ++i) {
        const int row = rr + RP * i;
        *reinterpret_cast<float4*>(out + row * DIM + 4 * c4) = *reinterpret_cast<const float4*>(T + row * LDT + 4 * c4);
    }
    if (NH == 2) {
        out[DIM * DIM + threadIdx.x] = (float)colsum;
    } else if (threadIdx.x < 2 * DIM) {                        // 4 row parts -> the 2 the finish pass expects (fixed order)
        const int h = threadIdx.x >> 7;
        out[DIM * DIM + threadIdx.x] = bp[(2 * h) * DIM + bc] + bp[(2 * h + 1) * DIM + bc];
    }
    WPROBE(4 * it + 1);
}

}  // namespace
