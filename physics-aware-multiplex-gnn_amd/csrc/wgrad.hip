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

namespace {

constexpr int MAXJ = 24;
constexpr int RB = 64;            // rows staged per step (2 x 36 KB LDS: 64 rows of dZ and A in flight per fetch)
constexpr int LDW = 144;          // LDS leading dim: 144 mod 32 = 16 -> conflict-free ds_read_b32 fragment reads
constexpr int ROWS_PER_WG = 256;
constexpr int MAX_SLOTS_PER_JOB = 256;

struct WJob {
    const float* dZ;
    const float* A;
    float* dW;
    float* db;        // may be null
    int64_t rows;
    int ld_dz, ld_a, ld_dw, a_mode;
};
struct WBatch {
    WJob job[MAXJ];
    int start[MAXJ + 1];          // slot prefix: job j owns slots [start[j], start[j+1])
    int njobs;
};

inline int job_slots(int64_t rows, int64_t chunk) {
    const int64_t want = (rows + chunk - 1) / chunk;
    return (int)(want < 1 ? 1 : (want > MAX_SLOTS_PER_JOB ? MAX_SLOTS_PER_JOB : want));
}

// Rows per slot for a batch: 256, or the smallest multiple of 64 above it for which the whole batch fits in
// TARGET_SLOTS workgroups -- one wave of workgroups over the 256 CUs instead of a full round plus a ragged tail
// (345 workgroups at the QM9 B=128 shape ran as 2 rounds: 49 us; one round of 256 x 384 rows: see DESIGN.md).
inline int target_slots() {
    static int t = [] { const char* e = getenv("PAMNET_WGRAD_SLOTS"); return e ? atoi(e) : 256; }();
    return t;
}
inline int64_t plan_chunk(int64_t njobs, const int64_t* rows) {
    int64_t chunk = ROWS_PER_WG;
    for (;; chunk += RB) {
        int64_t slots = 0;
        for (int64_t j = 0; j < njobs; ++j) slots += job_slots(rows[j], chunk);
        if (slots <= target_slots() || chunk >= 16384) return chunk;
    }
}

__device__ __forceinline__ void wgrad_body(const WBatch& batch, float* __restrict__ partial, const int bid, float* lds) {
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
    const int i0 = (w >> 1) * 64, j0 = (w & 1) * 64;      // wave tile: dW rows (n) [i0, i0+64), cols (k) [j0, j0+64)
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // bias gradient: thread t = (column t & 127, row half t >> 7); fp32 inside a 16-row half block, fp64 across blocks
    double colsum = 0.0;
    const int bc = threadIdx.x & 127, bh = threadIdx.x >> 7;
    const bool want_bias = jb.db != nullptr;

    // register double buffer: the next 32-row block is in flight from L2/HBM while the MFMAs chew on the current one
    const int c4 = threadIdx.x & 31, rr = threadIdx.x >> 5;
    float4 zr[RB / 8], ar[RB / 8];
    auto fetch = [&](int64_t r0) {
#pragma unroll
        for (int i = 0; i < RB / 8; ++i) {
            const int64_t g = r0 + rr + 8 * i;
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
        for (int i = 0; i < RB / 8; ++i) {
            const int r = rr + 8 * i;
            float4 a = ar[i];
            if (jb.a_mode == 1) a = f4silu(a);                 // SiLU(0) = 0 keeps the zero padding
            *reinterpret_cast<float4*>(Zs + r * LDW + 4 * c4) = zr[i];
            *reinterpret_cast<float4*>(As + r * LDW + 4 * c4) = a;
        }
        __syncthreads();
        WPROBE(4 * it + 1);
        if (r0 + RB < end) fetch(r0 + RB);
        if (want_bias) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
            for (int r = 0; r < RB / 2; r += 4) {
                s0 += Zs[((RB / 2) * bh + r) * LDW + bc];
                s1 += Zs[((RB / 2) * bh + r + 1) * LDW + bc];
                s2 += Zs[((RB / 2) * bh + r + 2) * LDW + bc];
                s3 += Zs[((RB / 2) * bh + r + 3) * LDW + bc];
            }
            colsum += (double)((s0 + s1) + (s2 + s3));
        }
        // operands of k-step st+1 are requested before the MFMAs of step st are issued (explicit register double
        // buffer): left alone, the compiler issues each ds_read right before its s_waitcnt and the LDS latency shows up
        // twice per k-step (12 400 instead of 8 192 cycles per 64-row block, tools/wgrad_probe.py)
        float za[2][4], ab[2][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            za[0][t] = Zs[kg * LDW + i0 + 16 * t + r16];
            ab[0][t] = As[kg * LDW + j0 + 16 * t + r16];
        }
#pragma unroll
        for (int st = 0; st < RB / 4; ++st) {
            const int cur = st & 1, nxt = cur ^ 1;
            if (st + 1 < RB / 4) {
                const int r = 4 * (st + 1) + kg;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    za[nxt][t] = Zs[r * LDW + i0 + 16 * t + r16];
                    ab[nxt][t] = As[r * LDW + j0 + 16 * t + r16];
                }
            }
            __builtin_amdgcn_sched_barrier(0);                 // keep the requests above this step's MFMAs
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(za[cur][a], ab[cur][b], acc[a][b], 0, 0, 0);
        }
        WPROBE(4 * it + 2);
        __syncthreads();
        WPROBE(4 * it + 3);
    }
    WPROBE(4 * it);
    // partial[slot][128*128 + 2*128]: the tile, then the two row-half bias partials
    // The accumulator layout (4 rows x 16 columns per store) would hit memory as 64-byte fragments; transpose through
    // LDS (the staging buffers are free now: 128 x 132 floats fit) and write the tile as coalesced 512-byte rows.
    float* out = partial + (int64_t)bid * (DIM * DIM + 2 * DIM);
    float* T = lds;
    static_assert(2 * RB * LDW >= DIM * LDT, "tile must fit in the staging buffers");
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                T[(i0 + 16 * a + kg * 4 + r) * LDT + j0 + 16 * b + r16] = acc[a][b][r];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < DIM / 8; ++i) {
        const int row = rr + 8 * i;
        *reinterpret_cast<float4*>(out + row * DIM + 4 * c4) = *reinterpret_cast<const float4*>(T + row * LDT + 4 * c4);
    }
    out[DIM * DIM + threadIdx.x] = (float)colsum;
    WPROBE(4 * it + 1);
}

__global__ __launch_bounds__(WG) void wgrad_kernel(WBatch batch, float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float lds[2 * RB * LDW];
    wgrad_body(batch, partial, (int)blockIdx.x, lds);
}

// Second pass, ONE launch per batch (grid 258 x (njobs + 1)), every sum in a fixed order (deterministic):
//   x <  256, y < njobs : 64 consecutive elements of job y's 128x128 tile, summed over its slots -- the slots are split
//                          over 16 groups (group g takes slots s0+g, s0+g+16, ...), then the 16 group sums are added;
//   x == 256, y < njobs : job y's bias gradient (two row-half partials per slot), fp64 across slots;
//   y == njobs, x < 17  : optional extra reduction riding along -- the node chain's head-vector partials
//                          ([blocks][257]: d w_out | d w_att | d b_out), so the chain's backward needs no launch of its own.
struct HeadJob {
    const float* partial;     // null: none
    int blocks;
    float *d_wout, *d_watt, *d_bout;
};

__device__ __forceinline__ void finish_body(const WBatch& batch, const float* __restrict__ partial, const HeadJob& head,
                                            const int bx, const int by, float* lds) {
    float4(*red)[16] = reinterpret_cast<float4(*)[16]>(lds);                 // [16][16] float4 = 4 KB
    constexpr int64_t SLOT = DIM * DIM + 2 * DIM;
    if (by == batch.njobs) {
        if (!head.partial || bx >= 17) return;
        float(*r1)[17] = reinterpret_cast<float(*)[17]>(&red[0][0]);      // 16 x 17 floats fit in the float4 array
        const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
        const int c = bx * 16 + cl;
        float s = 0.f;
        if (c < 257)
            for (int b = sl; b < head.blocks; b += 16) s += head.partial[(int64_t)b * 257 + c];
        r1[sl][cl] = s;
        __syncthreads();
        if (sl == 0 && c < 257) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += r1[q][cl];
            if (c < 128) head.d_wout[c] = t;
            else if (c < 256) head.d_watt[c - 128] = t;
            else head.d_bout[0] = t;
        }
        return;
    }
    const WJob jb = batch.job[by];
    const int s0 = batch.start[by], s1 = batch.start[by + 1];
    if (bx < 256) {
        const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
        const int e4 = bx * 16 + c;                       // float4 index inside the 128x128 tile
        float4 s = f4zero();
        for (int q = s0 + g; q < s1; q += 16)
            s = f4add(s, *reinterpret_cast<const float4*>(partial + (int64_t)q * SLOT + 4 * e4));
        red[g][c] = s;
        __syncthreads();
        if (g == 0) {
            float4 t = red[0][c];
#pragma unroll
            for (int k = 1; k < 16; ++k) t = f4add(t, red[k][c]);
            const int el = 4 * e4;
            *reinterpret_cast<float4*>(jb.dW + (int64_t)(el >> 7) * jb.ld_dw + (el & 127)) = t;
        }
    } else if (bx == 256) {
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

// One launch = the split-K pass of the current batch (blocks [0, cur slots)) + the fixed-order reduction of the
// PREVIOUS batch (the remaining blocks, 258 per job): the reduction is ~16 MB of L2-resident reads that ran as a
// launch of its own between two weight-gradient passes; here its small workgroups fill in beside the current pass.
constexpr int FIN_X = DIM * DIM / 64 + 2;
__global__ __launch_bounds__(WG) void wgrad_fused_kernel(WBatch cur, float* __restrict__ cur_partial, WBatch prev,
                                                         const float* __restrict__ prev_partial, HeadJob prev_head) {
    __shared__ __attribute__((aligned(16))) float lds[2 * RB * LDW];
    const int slots = cur.start[cur.njobs];
    if ((int)blockIdx.x < slots) {
        wgrad_body(cur, cur_partial, (int)blockIdx.x, lds);
    } else {
        const int fb = (int)blockIdx.x - slots;
        finish_body(prev, prev_partial, prev_head, fb % FIN_X, fb / FIN_X, lds);
    }
}

}  // namespace

// Scratch needed for a batch: sum_j clamp(ceil(rows_j/256), 1, 256) slots of (128*128 + 256) floats.
extern "C" int pamnet_wgrad_scratch_floats(int64_t njobs, const int64_t* rows, int64_t* floats) {
    if (njobs < 0 || njobs > MAXJ || !floats || (njobs > 0 && !rows)) return PAMNET_EINVAL;
    int64_t slots = 0;
    for (int64_t j = 0; j < njobs; ++j) slots += job_slots(rows[j], ROWS_PER_WG);       // upper bound for any plan
    *floats = slots * (int64_t)(DIM * DIM + 2 * DIM);
    return PAMNET_OK;
}

namespace {
struct WgradPending {
    WBatch batch;
    const float* partial;
    HeadJob head;
    int valid;
};

inline int build_batch(WBatch& b, int64_t njobs, const float* const* dZ, const int64_t* ld_dz, const float* const* A,
                       const int64_t* ld_a, const int32_t* a_mode, const int64_t* rows, float* const* dW,
                       const int64_t* ld_dw, float* const* db) {
    b.njobs = (int)njobs;
    b.start[0] = 0;
    const int64_t chunk = plan_chunk(njobs, rows);
    for (int j = 0; j < njobs; ++j) {
        if (!dZ[j] || !A[j] || !dW[j]) return PAMNET_ENULL;
        b.job[j] = WJob{dZ[j], A[j], dW[j], db[j], rows[j], (int)ld_dz[j], (int)ld_a[j], (int)ld_dw[j], a_mode[j]};
        b.start[j + 1] = b.start[j] + job_slots(rows[j], chunk);
    }
    return PAMNET_OK;
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
    const int rc = build_batch(b, njobs, dZ, ld_dz, A, ld_a, a_mode, rows, dW, ld_dw, db);
    if (rc) return rc;
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(wgrad_kernel, dim3((unsigned)b.start[njobs]), dim3(WG), 0, st, b, partial);
    PAMNET_LAUNCH_CHECK();
    const HeadJob head{head_partial, (int)head_blocks, d_wout, d_watt, d_bout};
    hipLaunchKernelGGL(wgrad_finish_kernel, dim3(FIN_X, (unsigned)njobs + 1), dim3(WG), 0, st, b, partial, head);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

// Deferred form for a sequence of batches (one per layer of a backward pass): the reduction of batch i rides along in
// the launch of batch i+1's split-K pass; pamnet_wgrad_flush_f32 reduces the last one.  `ctx`: caller-owned host memory
// of pamnet_wgrad_ctx_bytes bytes, zeroed before the first call (the library itself keeps no state); consecutive calls
// must use different `partial` buffers (the previous one is still being read).
extern "C" int pamnet_wgrad_ctx_bytes(int64_t* bytes) {
    if (!bytes) return PAMNET_ENULL;
    *bytes = (int64_t)sizeof(WgradPending);
    return PAMNET_OK;
}

extern "C" int pamnet_wgrad_deferred_f32(int64_t njobs, const float* const* dZ, const int64_t* ld_dz,
                                         const float* const* A, const int64_t* ld_a, const int32_t* a_mode,
                                         const int64_t* rows, float* const* dW, const int64_t* ld_dw, float* const* db,
                                         float* partial, const float* head_partial, int64_t head_blocks, float* d_wout,
                                         float* d_watt, float* d_bout, void* ctx, pamnet_stream_t stream) {
    if (njobs < 1 || njobs > MAXJ) return PAMNET_EINVAL;
    if (head_partial && (!d_wout || !d_watt || !d_bout || head_blocks < 0)) return PAMNET_ENULL;
    if (!dZ || !ld_dz || !A || !ld_a || !a_mode || !rows || !dW || !ld_dw || !db || !partial || !ctx) return PAMNET_ENULL;
    WgradPending* pend = static_cast<WgradPending*>(ctx);
    if (pend->valid && pend->partial == partial) return PAMNET_EINVAL;
    WBatch b;
    const int rc = build_batch(b, njobs, dZ, ld_dz, A, ld_a, a_mode, rows, dW, ld_dw, db);
    if (rc) return rc;
    hipStream_t st = as_stream(stream);
    if (pend->valid) {
        const unsigned grid = (unsigned)b.start[njobs] + (unsigned)(FIN_X * (pend->batch.njobs + 1));
        hipLaunchKernelGGL(wgrad_fused_kernel, dim3(grid), dim3(WG), 0, st, b, partial, pend->batch, pend->partial,
                           pend->head);
    } else {
        hipLaunchKernelGGL(wgrad_kernel, dim3((unsigned)b.start[njobs]), dim3(WG), 0, st, b, partial);
    }
    PAMNET_LAUNCH_CHECK();
    pend->batch = b;
    pend->partial = partial;
    pend->head = HeadJob{head_partial, (int)head_blocks, d_wout, d_watt, d_bout};
    pend->valid = 1;
    return PAMNET_OK;
}

extern "C" int pamnet_wgrad_flush_f32(void* ctx, pamnet_stream_t stream) {
    if (!ctx) return PAMNET_ENULL;
    WgradPending* pend = static_cast<WgradPending*>(ctx);
    if (!pend->valid) return PAMNET_OK;
    hipLaunchKernelGGL(wgrad_finish_kernel, dim3(FIN_X, (unsigned)pend->batch.njobs + 1), dim3(WG), 0, as_stream(stream),
                       pend->batch, pend->partial, pend->head);
    PAMNET_LAUNCH_CHECK();
    pend->valid = 0;
    return PAMNET_OK;
}
