// Batched weight-gradient GEMMs:  dW[128,128] = dZ[rows,128]^T * A[rows,128],  db[128] = column sums of dZ.
//
// The backward of every Linear in a PAMNet layer needs one of these; the reference launches one GEMM + one reduce per
// Linear (~60 per layer pair, most with only 2-4 K rows: launch-latency bound).  Here all jobs of a layer are ONE
// launch over a compact 1-D grid: job j owns ceil(rows_j / 256) consecutive workgroups ("slots"); each reduces its
// contiguous <=256-row chunk into a full 128x128 fp32 tile with v_mfma_f32_16x16x4_f32 (the row index is the MFMA k
// dimension) and writes it to its slot of a partial buffer; a second small kernel sums each job's slots in a fixed
// order -> deterministic, atomics-free.
// A may be given as a pre-activation (a_mode 1: A = SiLU(Z_prev) applied while staging), so activations that are a
// pure SiLU of a saved z are never stored twice.
#include <stdlib.h>

#include "common.h"
#include "gemm_core.h"

using namespace pamnet;

#ifdef PAMNET_PHASE_PROBE
// Development aid (tools/wgrad_probe.py): shader-clock timestamps of one workgroup, 4 per 64-row iteration.
__device__ long long pamnet_wgrad_probe[64];
#define WPROBE(i)                                                                                        \
    do {                                                                                                 \
        if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0 && (i) < 64) pamnet_wgrad_probe[i] = clock64(); \
    } while (0)
extern "C" int pamnet_wgrad_probe_read(long long* host64) {
    return (int)hipMemcpyFromSymbol(host64, HIP_SYMBOL(pamnet_wgrad_probe), sizeof(long long) * 64);
}
#else
#define WPROBE(i)
#endif

#include "wgrad_core.h"

namespace {

constexpr int ROWS_PER_WG = 256;
constexpr int MAX_SLOTS_PER_JOB = 256;

inline int job_slots(int64_t rows, int64_t chunk) {
    const int64_t want = (rows + chunk - 1) / chunk;
    return (int)(want < 1 ? 1 : (want > MAX_SLOTS_PER_JOB ? MAX_SLOTS_PER_JOB : want));
}

// Rows per slot for a batch: 256, or the smallest multiple of 64 above it for which the whole batch fits in
// TARGET_SLOTS workgroups -- one co-resident round over the 256 CUs instead of a full round plus a ragged tail
// (345 workgroups at the QM9 B=128 shape ran as 2 rounds: 49 us; one round of 256 x 384 rows: see DESIGN.md).
// Two slots per CU (launch bound 2, 2 x 68 KB of LDS): one's staging runs under the other's MFMAs.  Round 4, same box,
// alternated three times: 256 -> 512 slots  PDBbind step 8.77-8.80 -> 8.63-8.69 ms (its 700 k-row jobs), QM9 2.207-2.212 ->
// 2.205-2.211 (neutral: the pair launch has 137 k rows); 1 024 slots (four per CU in turn): +0.03 ms on both.
constexpr int target_slots() { return 512; }
// rows per rider slot to start from (a rider should not outlast the node chain it rides with: ~26 us)
constexpr int64_t rider_rows() { return 256; }
// smallest chunk a plan may use (scratch is sized for it)
constexpr int MIN_ROWS_PER_WG = 128;
constexpr int64_t first_rows() { return ROWS_PER_WG < MIN_ROWS_PER_WG ? MIN_ROWS_PER_WG : ROWS_PER_WG; }
inline int64_t plan_chunk(int64_t njobs, const int64_t* rows, int64_t max_slots, int64_t first_chunk = 0) {
    int64_t chunk = first_chunk > 0 ? first_chunk : first_rows();
    for (;; chunk += RB) {
        int64_t slots = 0;
        for (int64_t j = 0; j < njobs; ++j) slots += job_slots(rows[j], chunk);
        if (slots <= max_slots || chunk >= 16384) return chunk;
    }
}

// Slot workgroups: 4 waves, one per SIMD, <= 256 registers each, so that a second workgroup -- the small reductions of the
// fused launches -- shares the CU.  An 8-wave form in which every thread stages (two waves per SIMD) was measured and
// dropped: no faster per slot -- an MFMA in flight blocks the VALU issue of BOTH waves of its SIMD, the block time is the
// sum of all instructions either way (3 880 vs 3 950 cycles per 32 rows) -- and slower per launch (46.9 vs 31.7 us: with
// 8 x 256 registers resident the reductions no longer fit beside the slots).  The allocation sits right at 256: small
// changes to the body tip it into spilling (142 spilled registers doubled the launch to 62 us) -- check
// -Rpass-analysis=kernel-resource-usage after touching wgrad_core.h.
constexpr int SNW = 4, SWG = 64 * SNW;
__global__ __launch_bounds__(SWG, 2) void wgrad_kernel(WBatch batch, float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float lds[WGRAD_LDS_FLOATS];
    wgrad_body<SNW>(batch, partial, (int)blockIdx.x, lds);
}

// Second pass, ONE launch per batch (grid 66 x (njobs + 1)), every sum in a fixed order (deterministic):
//   x <  64, y < njobs : 256 consecutive elements of job y's 128x128 tile, summed over its slots -- the slots are split
//                         over 4 groups (group g takes slots s0+g, s0+g+4, ...), then the 4 group sums are added;
//   x == 64, y < njobs : job y's bias gradient (two row-half partials per slot), fp64 across slots;
//   y == njobs, x < 34 : optional extra reductions riding along -- up to two node chains' head-vector partials
//                          ([blocks][257]: d w_out | d w_att | d b_out), so the chain's backward needs no launch of its own.
constexpr int FIN_TILES = DIM * DIM / 256;      // reduction workgroups per job: 64 float4 each
constexpr int FIN_X = FIN_TILES + 2;            // + the bias gradient + a spare (the head vectors use a row of 34 of their own)
static_assert(FIN_X >= 34, "the head-vector reductions take workgroups [0, 34) of the extra row");
struct HeadOne {
    const float* partial;     // null: none
    int blocks;
    float *d_wout, *d_watt, *d_bout;
};
// up to two chains' head vectors per batch (a layer pair's merged launch carries the local and the global chain's)
struct HeadJob {
    HeadOne h[2];
};
constexpr HeadOne NO_HEAD{nullptr, 0, nullptr, nullptr, nullptr};

template <typename Batch>
__device__ __forceinline__ void finish_body(const Batch& batch, const float* __restrict__ partial, const HeadJob& head,
                                            const int bx, const int by, float* lds) {
    float4(*red)[16] = reinterpret_cast<float4(*)[16]>(lds);                 // [16][16] float4 = 4 KB
    constexpr int64_t SLOT = DIM * DIM + 2 * DIM;
    if (by == batch.njobs) {
        if (bx >= 34) return;
        const HeadOne hd = head.h[bx / 17];                                // workgroups [0, 17): first chain, [17, 34): second
        if (!hd.partial) return;
        const int hx = bx % 17;
        float(*r1)[17] = reinterpret_cast<float(*)[17]>(&red[0][0]);      // 16 x 17 floats fit in the float4 array
        const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
        const int c = hx * 16 + cl;
        float s = 0.f;
        if (c < 257)
            for (int b = sl; b < hd.blocks; b += 16) s += hd.partial[(int64_t)b * 257 + c];
        r1[sl][cl] = s;
        __syncthreads();
        if (sl == 0 && c < 257) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += r1[q][cl];
            if (c < 128) hd.d_wout[c] = t;
            else if (c < 256) hd.d_watt[c - 128] = t;
            else hd.d_bout[0] = t;
        }
        return;
    }
    const WJob jb = batch.job[by];
    const int s0 = batch.start[by], s1 = batch.start[by + 1];
    if (bx < FIN_TILES) {
        // 64 float4 columns x 4 slot groups per workgroup (a quarter of the workgroups of the 16 x 16 form: a reduction is
        // launch-bound -- node-level jobs have ~9 slots -- and up to 36 jobs wait for one weight-gradient launch)
        float4(*red4)[64] = reinterpret_cast<float4(*)[64]>(lds);           // [4][64] float4 = 4 KB
        const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
        const int e4 = bx * 64 + c;                       // float4 index inside the 128x128 tile
        float4 s = f4zero(), s2 = f4zero();
        int q = s0 + g;
        for (; q + 4 < s1; q += 8) {                      // two independent loads in flight, added in slot order
            const float4 u = *reinterpret_cast<const float4*>(partial + (int64_t)q * SLOT + 4 * e4);
            const float4 v = *reinterpret_cast<const float4*>(partial + (int64_t)(q + 4) * SLOT + 4 * e4);
            s = f4add(s, u);
            s2 = f4add(s2, v);
        }
        if (q < s1) s = f4add(s, *reinterpret_cast<const float4*>(partial + (int64_t)q * SLOT + 4 * e4));
        red4[g][c] = f4add(s, s2);
        __syncthreads();
        if (g == 0) {
            const float4 t = f4add(f4add(red4[0][c], red4[1][c]), f4add(red4[2][c], red4[3][c]));
            const int el = 4 * e4;
            *reinterpret_cast<float4*>(jb.dW + (int64_t)(el >> 7) * jb.ld_dw + (el & 127)) = t;
        }
    } else if (bx == FIN_TILES) {
        if (!jb.db) return;
        // 32 float4 columns x 8 slot groups, fp64 across slots, fixed order
        double(*rd)[128] = reinterpret_cast<double(*)[128]>(lds + 1024);          // [8][128] doubles behind `red`
        const int c4 = threadIdx.x & 31, g = threadIdx.x >> 5;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        for (int q = s0 + g; q < s1; q += 8) {
            const float* p = partial + (int64_t)q * SLOT + DIM * DIM;
            const float4 u = *reinterpret_cast<const float4*>(p + 4 * c4);
            const float4 v = *reinterpret_cast<const float4*>(p + DIM + 4 * c4);
            a0 += (double)u.x + (double)v.x;
            a1 += (double)u.y + (double)v.y;
            a2 += (double)u.z + (double)v.z;
            a3 += (double)u.w + (double)v.w;
        }
        rd[g][4 * c4] = a0, rd[g][4 * c4 + 1] = a1, rd[g][4 * c4 + 2] = a2, rd[g][4 * c4 + 3] = a3;
        __syncthreads();
        if (threadIdx.x < 128) {
            double t = rd[0][threadIdx.x];
#pragma unroll
            for (int k = 1; k < 8; ++k) t += rd[k][threadIdx.x];
            jb.db[threadIdx.x] = (float)t;
        }
    }
}

__global__ __launch_bounds__(WG) void wgrad_finish_kernel(WBatch batch, const float* __restrict__ partial, HeadJob head) {
    __shared__ __attribute__((aligned(16))) float lds[1024 + 2 * 8 * 128];
    finish_body(batch, partial, head, (int)blockIdx.x, (int)blockIdx.y, lds);
}

// One launch = the split-K pass of the current batch (blocks [0, cur slots)) + the fixed-order reductions of up to two
// EARLIER batches (the remaining blocks, 258 per job): the previous layer's main batch (with the node chain's
// head-vector partials) and its rider batch (the slots that ran as extra workgroups of a node-chain launch).  A reduction
// is ~16 MB of L2-resident reads that ran as a launch of its own between two weight-gradient passes; here its small
// workgroups fill in beside the current pass.  Compact batches (<= 16 jobs): the three descriptors share the 4 KB
// kernel-argument block.

// `edge` (round 5): the partial tiles the fused global-edge backward left behind (edge_agg.hip global_edge_agg_bwd_wg_kernel: one
// slot per workgroup and job, dW_e with the bias parts and dW_ea) -- a third earlier batch, reduced by the LAST blocks of the grid.
using WBatchE = WBatchT<2>;
__global__ __launch_bounds__(SWG, 2) void wgrad_fused_kernel(WBatchS cur, float* __restrict__ cur_partial, WBatchS prev,
                                                         const float* __restrict__ prev_partial, HeadJob prev_head,
                                                         WBatch prev2, const float* __restrict__ prev2_partial, WBatchE edge,
                                                         const float* __restrict__ edge_partial) {
    __shared__ __attribute__((aligned(16))) float lds[WGRAD_LDS_FLOATS];
    const int slots = cur.start[cur.njobs];
    int fb = (int)blockIdx.x - slots;
    if (fb < 0) {
        wgrad_body<SNW>(cur, cur_partial, (int)blockIdx.x, lds);
        return;
    }
    if (threadIdx.x >= WG) return;                             // the reductions are 4-wave workgroups
    const int fin1 = prev_partial ? FIN_X * (prev.njobs + 1) : 0;
    const int fin2 = prev2_partial ? FIN_X * (prev2.njobs + 1) : 0;
    if (fb < fin1) {
        finish_body(prev, prev_partial, prev_head, fb % FIN_X, fb / FIN_X, lds);
    } else if (fb < fin1 + fin2) {
        fb -= fin1;
        finish_body(prev2, prev2_partial, HeadJob{{NO_HEAD, NO_HEAD}}, fb % FIN_X, fb / FIN_X, lds);
    } else {
        fb -= fin1 + fin2;
        finish_body(edge, edge_partial, HeadJob{{NO_HEAD, NO_HEAD}}, fb % FIN_X, fb / FIN_X, lds);
    }
}

// the same with one earlier batch and wide descriptors (up to 24 jobs each): no rider batch pending
__global__ __launch_bounds__(SWG, 2) void wgrad_fused_wide_kernel(WBatch cur, float* __restrict__ cur_partial, WBatch prev,
                                                              const float* __restrict__ prev_partial, HeadJob prev_head,
                                                              WBatchE edge, const float* __restrict__ edge_partial) {
    __shared__ __attribute__((aligned(16))) float lds[WGRAD_LDS_FLOATS];
    const int slots = cur.start[cur.njobs];
    if ((int)blockIdx.x < slots) {
        wgrad_body<SNW>(cur, cur_partial, (int)blockIdx.x, lds);
    } else {
        if (threadIdx.x >= WG) return;
        int fb = (int)blockIdx.x - slots;
        const int fin1 = FIN_X * (prev.njobs + 1);
        if (fb < fin1) {
            finish_body(prev, prev_partial, prev_head, fb % FIN_X, fb / FIN_X, lds);
        } else {
            fb -= fin1;
            finish_body(edge, edge_partial, HeadJob{{NO_HEAD, NO_HEAD}}, fb % FIN_X, fb / FIN_X, lds);
        }
    }
}

}  // namespace

// Scratch needed for a batch: sum_j clamp(ceil(rows_j/128), 1, 256) slots of (128*128 + 256) floats.
extern "C" int pamnet_wgrad_scratch_floats(int64_t njobs, const int64_t* rows, int64_t* floats) {
    if (njobs < 0 || njobs > MAXJ || !floats || (njobs > 0 && !rows)) return PAMNET_EINVAL;
    int64_t slots = 0;
    for (int64_t j = 0; j < njobs; ++j) slots += job_slots(rows[j], MIN_ROWS_PER_WG);   // upper bound for any plan
    *floats = slots * (int64_t)(DIM * DIM + 2 * DIM);
    return PAMNET_OK;
}

namespace {
// caller-owned host state of a sequence of deferred batches: up to two batches wait for their reduction
struct WgradPending {
    WBatch batch[2];              // [0]: main batch (with head job), [1]: rider batch
    const float* partial[2];
    HeadJob head;
    int valid[2];
    WBatchE edge;                 // partial tiles left by the fused global-edge backward (pamnet_wgrad_edge_enqueue_f32)
    const float* edge_partial;
    int edge_valid;
};

template <typename Batch>
inline int build_batch(Batch& b, int64_t njobs, const float* const* dZ, const int64_t* ld_dz, const float* const* A,
                       const int64_t* ld_a, const int32_t* a_mode, const int64_t* rows, float* const* dW,
                       const int64_t* ld_dw, float* const* db, int64_t max_slots, int64_t first_chunk = 0) {
    b.njobs = (int)njobs;
    b.start[0] = 0;
    const int64_t chunk = plan_chunk(njobs, rows, max_slots, first_chunk);
    for (int j = 0; j < njobs; ++j) {
        if (!dZ[j] || !A[j] || !dW[j]) return PAMNET_ENULL;
        b.job[j] = WJob{dZ[j], A[j], dW[j], db[j], rows[j], (int)ld_dz[j], (int)ld_a[j], (int)ld_dw[j], a_mode[j]};
        b.start[j + 1] = b.start[j] + job_slots(rows[j], chunk);
    }
    return PAMNET_OK;
}

inline WBatchS compact(const WBatch& b) {                      // njobs <= MAXJ_S
    WBatchS c;
    c.njobs = b.njobs;
    for (int j = 0; j < b.njobs; ++j) c.job[j] = b.job[j], c.start[j] = b.start[j];
    c.start[b.njobs] = b.start[b.njobs];
    return c;
}
template <typename B>
inline WBatch widen(const B& b) {
    WBatch c;
    c.njobs = b.njobs;
    for (int j = 0; j < b.njobs; ++j) c.job[j] = b.job[j], c.start[j] = b.start[j];
    c.start[b.njobs] = b.start[b.njobs];
    return c;
}
}  // namespace

// jobs described by parallel host arrays (njobs <= 24).  partial: pamnet_wgrad_scratch_floats(njobs, rows) floats.
extern "C" int pamnet_wgrad_batched_f32(int64_t njobs, const float* const* dZ, const int64_t* ld_dz,
                                        const float* const* A, const int64_t* ld_a, const int32_t* a_mode,
                                        const int64_t* rows, float* const* dW, const int64_t* ld_dw, float* const* db,
                                        float* partial, const float* head_partial, int64_t head_blocks,
                                        float* d_wout, float* d_watt, float* d_bout, pamnet_stream_t stream) {
    if (njobs < 0 || njobs > MAXJ) return PAMNET_EINVAL;
    if (njobs == 0) return head_partial ? PAMNET_EINVAL : PAMNET_OK;
    if (head_partial && (!d_wout || !d_watt || !d_bout || head_blocks < 0)) return PAMNET_ENULL;
    if (!dZ || !ld_dz || !A || !ld_a || !a_mode || !rows || !dW || !ld_dw || !db || !partial) return PAMNET_ENULL;
    WBatch b;
    const int rc = build_batch(b, njobs, dZ, ld_dz, A, ld_a, a_mode, rows, dW, ld_dw, db, target_slots());
    if (rc) return rc;
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(wgrad_kernel, dim3((unsigned)b.start[njobs]), dim3(SWG), 0, st, b, partial);
    PAMNET_LAUNCH_CHECK();
    const HeadJob head{{HeadOne{head_partial, (int)head_blocks, d_wout, d_watt, d_bout}, NO_HEAD}};
    hipLaunchKernelGGL(wgrad_finish_kernel, dim3(FIN_X, (unsigned)njobs + 1), dim3(WG), 0, st, b, partial, head);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

// Deferred form for a sequence of batches (one per layer of a backward pass): the reduction of batch i rides along in
// the launch of batch i+1's split-K pass; pamnet_wgrad_flush_f32 reduces what is left.  `ctx`: caller-owned host memory
// of pamnet_wgrad_ctx_bytes bytes, zeroed before the first call (the library itself keeps no state); consecutive calls
// must use different `partial` buffers (the previous one is still being read).
extern "C" int pamnet_wgrad_ctx_bytes(int64_t* bytes) {
    if (!bytes) return PAMNET_ENULL;
    *bytes = (int64_t)sizeof(WgradPending);
    return PAMNET_OK;
}

static int finish_pending(WgradPending* pend, hipStream_t st) {
    const HeadJob none{{NO_HEAD, NO_HEAD}};
    for (int k = 0; k < 2; ++k) {
        if (!pend->valid[k]) continue;
        hipLaunchKernelGGL(wgrad_finish_kernel, dim3(FIN_X, (unsigned)pend->batch[k].njobs + 1), dim3(WG), 0, st,
                           pend->batch[k], pend->partial[k], k == 0 ? pend->head : none);
        PAMNET_LAUNCH_CHECK();
        pend->valid[k] = 0;
    }
    if (pend->edge_valid) {
        hipLaunchKernelGGL(wgrad_finish_kernel, dim3(FIN_X, (unsigned)pend->edge.njobs), dim3(WG), 0, st, widen(pend->edge),
                           pend->edge_partial, none);
        PAMNET_LAUNCH_CHECK();
        pend->edge_valid = 0;
    }
    return PAMNET_OK;
}

extern "C" int pamnet_wgrad_deferred_f32(int64_t njobs, const float* const* dZ, const int64_t* ld_dz,
                                         const float* const* A, const int64_t* ld_a, const int32_t* a_mode,
                                         const int64_t* rows, float* const* dW, const int64_t* ld_dw, float* const* db,
                                         float* partial, const float* head_partial, int64_t head_blocks, float* d_wout,
                                         float* d_watt, float* d_bout, const float* head2_partial, float* d_wout2,
                                         float* d_watt2, float* d_bout2, void* ctx, pamnet_stream_t stream) {
    if (njobs < 1 || njobs > MAXJ) return PAMNET_EINVAL;
    if (head_partial && (!d_wout || !d_watt || !d_bout || head_blocks < 0)) return PAMNET_ENULL;
    if (head2_partial && (!d_wout2 || !d_watt2 || !d_bout2 || head_blocks < 0)) return PAMNET_ENULL;
    if (!dZ || !ld_dz || !A || !ld_a || !a_mode || !rows || !dW || !ld_dw || !db || !partial || !ctx) return PAMNET_ENULL;
    WgradPending* pend = static_cast<WgradPending*>(ctx);
    if ((pend->valid[0] && pend->partial[0] == partial) || (pend->valid[1] && pend->partial[1] == partial)) return PAMNET_EINVAL;
    WBatch b;
    const int rc = build_batch(b, njobs, dZ, ld_dz, A, ld_a, a_mode, rows, dW, ld_dw, db, target_slots());
    if (rc) return rc;
    hipStream_t st = as_stream(stream);
    const bool any = pend->valid[0] || pend->valid[1] || pend->edge_valid;
    const bool fits = njobs <= MAXJ_S && (!pend->valid[0] || pend->batch[0].njobs <= MAXJ_S);   // (the rider batch: wide)
    WBatchE edge;
    edge.njobs = 0, edge.start[0] = 0;
    if (pend->edge_valid) edge = pend->edge;
    const float* edge_partial = pend->edge_valid ? pend->edge_partial : nullptr;
    const unsigned efin = (unsigned)(FIN_X * edge.njobs);
    if (pend->valid[0] && !pend->valid[1]) {                   // one earlier batch: wide descriptors
        const unsigned grid = (unsigned)b.start[njobs] + (unsigned)(FIN_X * (pend->batch[0].njobs + 1)) + efin;
        hipLaunchKernelGGL(wgrad_fused_wide_kernel, dim3(grid), dim3(SWG), 0, st, b, partial, pend->batch[0], pend->partial[0],
                           pend->head, edge, edge_partial);
        PAMNET_LAUNCH_CHECK();
        pend->valid[0] = 0;
        pend->edge_valid = 0;
    } else if (any && fits) {
        WBatchS none;
        none.njobs = 0, none.start[0] = 0;
        const WBatchS p0 = pend->valid[0] ? compact(pend->batch[0]) : none;
        WBatch p1;
        p1.njobs = 0, p1.start[0] = 0;
        if (pend->valid[1]) p1 = pend->batch[1];
        const unsigned fin = (pend->valid[0] ? FIN_X * (p0.njobs + 1) : 0) + (pend->valid[1] ? FIN_X * (p1.njobs + 1) : 0);
        const HeadJob nohead{{NO_HEAD, NO_HEAD}};
        hipLaunchKernelGGL(wgrad_fused_kernel, dim3((unsigned)b.start[njobs] + fin + efin), dim3(SWG), 0, st, compact(b), partial,
                           p0, pend->valid[0] ? pend->partial[0] : nullptr, pend->valid[0] ? pend->head : nohead, p1,
                           pend->valid[1] ? pend->partial[1] : nullptr, edge, edge_partial);
        PAMNET_LAUNCH_CHECK();
        pend->valid[0] = pend->valid[1] = 0;
        pend->edge_valid = 0;
    } else {
        if (any) {                                             // too many jobs for the compact descriptors: plain launches
            const int frc = finish_pending(pend, st);
            if (frc) return frc;
        }
        hipLaunchKernelGGL(wgrad_kernel, dim3((unsigned)b.start[njobs]), dim3(SWG), 0, st, b, partial);
        PAMNET_LAUNCH_CHECK();
    }
    pend->batch[0] = b;
    pend->partial[0] = partial;
    pend->head = HeadJob{{HeadOne{head_partial, (int)head_blocks, d_wout, d_watt, d_bout},
                          HeadOne{head2_partial, (int)head_blocks, d_wout2, d_watt2, d_bout2}}};
    pend->valid[0] = 1;
    return PAMNET_OK;
}

extern "C" int pamnet_wgrad_flush_f32(void* ctx, pamnet_stream_t stream) {
    if (!ctx) return PAMNET_ENULL;
    return finish_pending(static_cast<WgradPending*>(ctx), as_stream(stream));
}

// ---- riders: slots of a batch that run as extra workgroups of a node-chain backward launch -----------------------------
// pamnet_wgrad_rider_plan_f32 lays a batch (<= 12 jobs) out over <= max_slots slots and stores the plan in `rider`
// (caller-owned host memory, pamnet_wgrad_rider_bytes); pamnet_node_pre_tail_bwd_f32 takes the plan and appends the
// slots to its grid; pamnet_wgrad_rider_enqueue_f32 then registers the batch with `ctx` so that the next deferred launch
// (or the flush) reduces its slots.  *slots_out = workgroups the plan adds.
extern "C" int pamnet_wgrad_rider_bytes(int64_t* bytes) {
    if (!bytes) return PAMNET_ENULL;
    *bytes = (int64_t)sizeof(WgradRider);
    return PAMNET_OK;
}

extern "C" int pamnet_wgrad_rider_plan_f32(int64_t njobs, const float* const* dZ, const int64_t* ld_dz,
                                           const float* const* A, const int64_t* ld_a, const int32_t* a_mode,
                                           const int64_t* rows, float* const* dW, const int64_t* ld_dw, float* const* db,
                                           float* partial, int64_t max_slots, void* rider, int64_t* slots_out) {
    if (njobs < 1 || njobs > MAXJ_S || max_slots < njobs) return PAMNET_EINVAL;
    if (!dZ || !ld_dz || !A || !ld_a || !a_mode || !rows || !dW || !ld_dw || !db || !partial || !rider) return PAMNET_ENULL;
    WgradRider* r = static_cast<WgradRider*>(rider);
    const int rc = build_batch(r->batch, njobs, dZ, ld_dz, A, ld_a, a_mode, rows, dW, ld_dw, db, max_slots, rider_rows());
    if (rc) return rc;
    r->partial = partial;
    r->slots = r->batch.start[njobs];
    if (slots_out) *slots_out = r->slots;
    return PAMNET_OK;
}

extern "C" int pamnet_wgrad_rider_enqueue_f32(void* ctx, const void* rider) {
    if (!ctx || !rider) return PAMNET_ENULL;
    WgradPending* pend = static_cast<WgradPending*>(ctx);
    const WgradRider* r = static_cast<const WgradRider*>(rider);
    if (pend->valid[1]) {
        // A second rider batch before the next reduction (a layer pair's merged launch follows TWO chain launches): appended
        // to the first one's descriptor.  Its slots must lie right behind the first one's in the same scratch buffer.
        WBatch& b = pend->batch[1];
        const int s0 = b.start[b.njobs];
        if (b.njobs + r->batch.njobs > MAXJ) return PAMNET_EINVAL;
        if (r->partial != pend->partial[1] + (int64_t)s0 * (DIM * DIM + 2 * DIM)) return PAMNET_EINVAL;
        for (int j = 0; j < r->batch.njobs; ++j) {
            b.job[b.njobs + j] = r->batch.job[j];
            b.start[b.njobs + j + 1] = s0 + r->batch.start[j + 1];
        }
        b.njobs += r->batch.njobs;
        return PAMNET_OK;
    }
    pend->batch[1] = widen(r->batch);
    pend->partial[1] = r->partial;
    pend->valid[1] = 1;
    return PAMNET_OK;
}

// ---- the fused global-edge backward's own weight gradients (edge_agg.hip pamnet_global_edge_agg_bwd_wg_f32) -------------------
// That launch left 2 * slots partial tiles in `partial` (slot format above): slots [0, slots) = shares of dW_e with the bias
// parts, [slots, 2 slots) = shares of dW_ea.  Registered here, they are summed in slot order by the next
// pamnet_wgrad_deferred_f32 launch (or the flush) like any other batch; `partial` must stay untouched until then.
extern "C" int pamnet_wgrad_edge_enqueue_f32(void* ctx, int64_t slots, float* dW_e, int64_t ld_e, float* db, float* dW_ea,
                                             int64_t ld_ea, const float* partial) {
    if (!ctx || !dW_e || !dW_ea || !partial) return PAMNET_ENULL;
    if (slots < 1 || slots > MAX_SLOTS_PER_JOB) return PAMNET_EINVAL;
    WgradPending* pend = static_cast<WgradPending*>(ctx);
    if (pend->edge_valid) return PAMNET_EINVAL;                 // one at a time: the next launch consumes it
    WBatchE& b = pend->edge;
    b.njobs = 2;
    b.start[0] = 0, b.start[1] = (int)slots, b.start[2] = 2 * (int)slots;
    b.job[0] = WJob{nullptr, nullptr, dW_e, db, 0, 0, 0, (int)ld_e, 0};
    b.job[1] = WJob{nullptr, nullptr, dW_ea, nullptr, 0, 0, 0, (int)ld_ea, 0};
    pend->edge_partial = partial;
    pend->edge_valid = 1;
    return PAMNET_OK;
}
