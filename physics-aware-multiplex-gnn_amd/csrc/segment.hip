// Sorted-CSR segment reduction and row gather: the scatter-add / gather of PAMNet's message passing.
//
// Replaces torch_scatter.scatter(..., reduce='add') (layers/local_message_passing.py:50,54,107,111), PyG's
// MessagePassing add-aggregation (layers/global_message_passing.py:38) and the x[i] / m[idx] row gathers.
//
// Design (HBM-bound; MI355X_MICROARCH: 16 B/lane coalesced loads, many loads in flight, no atomics):
//   * a row of d floats is covered by LPR = d/4 lanes holding one float4 each, so a wave64 owns 64/LPR output rows
//     (d=128: two rows per wave, every global access is a full 512 B row segment = 4 cache lines, 16 B per lane);
//   * each lane group walks its CSR segment with a 4-deep unrolled software pipeline (4 independent 16 B loads in
//     flight per lane, 8 waves/SIMD at this register footprint) and accumulates in registers -> one store per row;
//   * deterministic: summation order = CSR order, run-to-run bitwise identical.
#include "common.h"

namespace {

typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_stream(const float4* p) {
    const f32x4_t t = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p));
    return make_float4(t[0], t[1], t[2], t[3]);
}

template <bool HAS_IA, bool HAS_B, bool HAS_IB, bool HAS_PERM>
__device__ __forceinline__ float4 load_term(const float4* __restrict__ A, const int32_t* __restrict__ ia,
                                            const float4* __restrict__ B, const int32_t* __restrict__ ib,
                                            const int32_t* __restrict__ perm, int64_t q, int64_t d4, int c) {
    int64_t k = HAS_PERM ? (int64_t)perm[q] : q;
    int64_t ra = HAS_IA ? (int64_t)ia[k] : k;
    // rows walked in storage order are read exactly once per launch: streamed (non-temporal), so that they do not push
    // re-used lines (gathered rows, the CSR pointer) out of the XCD's L2
    float4 a = (!HAS_IA && !HAS_PERM) ? ld_stream(A + ra * d4 + c) : A[ra * d4 + c];
    if (HAS_B) {
        int64_t rb = HAS_IB ? (int64_t)ib[k] : k;
        float4 b = (!HAS_IB && !HAS_PERM) ? ld_stream(B + rb * d4 + c) : B[rb * d4 + c];
        a.x *= b.x; a.y *= b.y; a.z *= b.z; a.w *= b.w;
    }
    return a;
}

// rows * lanes-per-row at or below this use the split kernel (four lane groups share a row): up to four waves of lanes per
// SIMD on 256 CUs.  (One wave per SIMD until round 4; the node-level reductions of a PDBbind batch -- 19 000 rows of ~37 terms,
// 608 k lanes -- are 4 x shorter dependent walks with the split: gather form 72-76 -> 66-70 us, tools/perm_probe.py.)
constexpr int64_t SPLIT_MAX_LANES = 256 * 4 * 64 * 4 * 4;

__device__ __forceinline__ void acc4(float4& s, const float4& v) { s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }

// LPR lanes per row (power of two <= 64).  blockDim.x = 256.
template <int LPR, bool HAS_IA, bool HAS_B, bool HAS_IB, bool HAS_PERM>
__global__ __launch_bounds__(256) void segment_sum_kernel(float4* __restrict__ out, const float4* __restrict__ init,
                                                          const float4* __restrict__ A, const int32_t* __restrict__ ia,
                                                          const float4* __restrict__ B, const int32_t* __restrict__ ib,
                                                          const int32_t* __restrict__ perm,
                                                          const int32_t* __restrict__ ptr, int64_t rows, int64_t d4) {
    constexpr int ROWS_PER_BLOCK = 256 / LPR;
    const int c = threadIdx.x % LPR;                       // float4 column owned by this lane
    const int sub = threadIdx.x / LPR;                     // row slot inside the block
    for (int64_t r = (int64_t)blockIdx.x * ROWS_PER_BLOCK + sub; r < rows; r += (int64_t)gridDim.x * ROWS_PER_BLOCK) {
        const int64_t beg = ptr[r], end = ptr[r + 1];
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
        if (init) s0 = init[r * d4 + c];
        int64_t q = beg;
        for (; q + 8 <= end; q += 8) {                     // 8 independent 16 B loads in flight per lane; the adds are the
            float4 v[8];                                   // two 4-blocks below in sequence: the same summation order
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = load_term<HAS_IA, HAS_B, HAS_IB, HAS_PERM>(A, ia, B, ib, perm, q + u, d4, c);
            acc4(s0, v[0]); acc4(s1, v[1]); acc4(s2, v[2]); acc4(s3, v[3]);
            acc4(s0, v[4]); acc4(s1, v[5]); acc4(s2, v[6]); acc4(s3, v[7]);
        }
        for (; q + 4 <= end; q += 4) {                     // 4 independent 16 B loads (x2 with B) in flight per lane
            float4 v0 = load_term<HAS_IA, HAS_B, HAS_IB, HAS_PERM>(A, ia, B, ib, perm, q + 0, d4, c);
            float4 v1 = load_term<HAS_IA, HAS_B, HAS_IB, HAS_PERM>(A, ia, B, ib, perm, q + 1, d4, c);
            float4 v2 = load_term<HAS_IA, HAS_B, HAS_IB, HAS_PERM>(A, ia, B, ib, perm, q + 2, d4, c);
            float4 v3 = load_term<HAS_IA, HAS_B, HAS_IB, HAS_PERM>(A, ia, B, ib, perm, q + 3, d4, c);
            acc4(s0, v0); acc4(s1, v1); acc4(s2, v2); acc4(s3, v3);
        }
        for (; q < end; ++q) acc4(s0, load_term<HAS_IA, HAS_B, HAS_IB, HAS_PERM>(A, ia, B, ib, perm, q, d4, c));
        acc4(s0, s1); acc4(s2, s3); acc4(s0, s2);
        out[r * d4 + c] = s0;                              // (a non-temporal store here measured slower: 5.68 vs 5.85 TB/s)
    }
}

// Few output rows (node-level reductions at batch size 128: ~2 K rows x ~14 terms) leave most of the chip idle and make
// the serial CSR walk the whole cost.  Here SPLIT lane groups share a row: each walks a contiguous quarter of the
// segment, the partial sums meet in LDS and are added in part order (fixed order -> deterministic).
template <int LPR, int SPLIT, bool HAS_IA, bool HAS_B, bool HAS_IB, bool HAS_PERM>
__global__ __launch_bounds__(256) void segment_sum_split_kernel(float4* __restrict__ out,
                                                                const float4* __restrict__ init,
                                                                const float4* __restrict__ A,
                                                                const int32_t* __restrict__ ia,
                                                                const float4* __restrict__ B,
                                                                const int32_t* __restrict__ ib,
                                                                const int32_t* __restrict__ perm,
                                                                const int32_t* __restrict__ ptr, int64_t rows,
                                                                int64_t d4) {
    constexpr int SLOTS = 256 / LPR;
    constexpr int RPB = SLOTS / SPLIT;
    static_assert(RPB >= 1, "SPLIT too large for this row width");
    __shared__ float4 red[SLOTS][LPR];
    const int c = threadIdx.x % LPR;
    const int slot = threadIdx.x / LPR;
    const int part = slot % SPLIT, rsub = slot / SPLIT;
    for (int64_t r0 = (int64_t)blockIdx.x * RPB; r0 < rows; r0 += (int64_t)gridDim.x * RPB) {
        const int64_t r = r0 + rsub;
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
        if (r < rows) {
            const int64_t beg = ptr[r], end = ptr[r + 1];
            const int64_t per = (end - beg + SPLIT - 1) / SPLIT;
            int64_t q = beg + part * per;
            const int64_t stop = q + per < end ? q + per : end;
            for (; q + 4 <= stop; q += 4) {
                float4 v0 = load_term<HAS_IA, HAS_B, HAS_IB, HAS_PERM>(A, ia, B, ib, perm, q + 0, d4, c);
                float4 v1 = load_term<HAS_IA, HAS_B, HAS_IB, HAS_PERM>(A, ia, B, ib, perm, q + 1, d4, c);
                float4 v2 = load_term<HAS_IA, HAS_B, HAS_IB, HAS_PERM>(A, ia, B, ib, perm, q + 2, d4, c);
                float4 v3 = load_term<HAS_IA, HAS_B, HAS_IB, HAS_PERM>(A, ia, B, ib, perm, q + 3, d4, c);
                acc4(s0, v0); acc4(s1, v1); acc4(s2, v2); acc4(s3, v3);
            }
            if (q < stop) acc4(s0, load_term<HAS_IA, HAS_B, HAS_IB, HAS_PERM>(A, ia, B, ib, perm, q, d4, c));
            if (q + 1 < stop) acc4(s1, load_term<HAS_IA, HAS_B, HAS_IB, HAS_PERM>(A, ia, B, ib, perm, q + 1, d4, c));
            if (q + 2 < stop) acc4(s2, load_term<HAS_IA, HAS_B, HAS_IB, HAS_PERM>(A, ia, B, ib, perm, q + 2, d4, c));
            acc4(s0, s1); acc4(s2, s3); acc4(s0, s2);
        }
        red[slot][c] = s0;
        __syncthreads();
        if (part == 0 && r < rows) {
            float4 t = init ? init[r * d4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int p = 0; p < SPLIT; ++p) acc4(t, red[rsub * SPLIT + p][c]);
            out[r * d4 + c] = t;
        }
        __syncthreads();
    }
}

// Fallback for d/4 not a power of two (e.g. d = 12, 20, 40): one wave per row, lanes stride over float4 columns;
// optional operands resolved at run time (rare widths, not a tuned path).
__global__ __launch_bounds__(256) void segment_sum_generic_kernel(float4* __restrict__ out,
                                                                  const float4* __restrict__ init,
                                                                  const float4* __restrict__ A,
                                                                  const int32_t* __restrict__ ia,
                                                                  const float4* __restrict__ B,
                                                                  const int32_t* __restrict__ ib,
                                                                  const int32_t* __restrict__ perm,
                                                                  const int32_t* __restrict__ ptr, int64_t rows,
                                                                  int64_t d4) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r < rows; r += n_waves) {
        const int64_t beg = ptr[r], end = ptr[r + 1];
        for (int c = lane; c < d4; c += 64) {
            float4 s = init ? init[r * d4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
            for (int64_t q = beg; q < end; ++q) {
                const int64_t k = perm ? (int64_t)perm[q] : q;
                float4 a = A[(ia ? (int64_t)ia[k] : k) * d4 + c];
                if (B) {
                    const float4 b = B[(ib ? (int64_t)ib[k] : k) * d4 + c];
                    a.x *= b.x; a.y *= b.y; a.z *= b.z; a.w *= b.w;
                }
                acc4(s, a);
            }
            out[r * d4 + c] = s;
        }
    }
}

template <bool HAS_IA, bool HAS_B, bool HAS_IB>
__global__ __launch_bounds__(256) void gather_mul_kernel(float4* __restrict__ out, const float4* __restrict__ A,
                                                         const int32_t* __restrict__ ia, const float4* __restrict__ B,
                                                         const int32_t* __restrict__ ib, int64_t m, int64_t d4) {
    const int64_t total = m * d4;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = t / d4;
        const int c = (int)(t - k * d4);
        out[t] = load_term<HAS_IA, HAS_B, HAS_IB, false>(A, ia, B, ib, nullptr, k, d4, c);
    }
}

template <int LPR>
int launch_segment_sum(float* out, const float* init, const float* A, const int32_t* ia, const float* B,
                       const int32_t* ib, const int32_t* perm, const int32_t* ptr, int64_t rows, int64_t d4,
                       hipStream_t st) {
    constexpr int SPLIT = (256 / LPR >= 4) ? 4 : 1;
    constexpr int SPLIT_WIDE = 256 / LPR;                  // a whole workgroup per row
    // few rows: split every segment over SPLIT lane groups (see segment_sum_split_kernel); a handful of rows (the
    // gradient of the 5-row atom-type embedding sums ~450 node rows per type): the whole workgroup shares one row
    const bool wide = SPLIT_WIDE > SPLIT && rows <= 64;
    const bool split = SPLIT > 1 && rows * LPR <= (int64_t)SPLIT_MAX_LANES;
    constexpr int RPB_FULL = 256 / LPR;
    const int RPB = wide ? 1 : (split ? RPB_FULL / SPLIT : RPB_FULL);
    int64_t grid = ceil_div(rows, RPB);
    if (grid > 256 * 64) grid = 256 * 64;                  // grid-stride above 64 blocks / CU
    if (grid < 1) grid = 1;
#define PAMNET_SEG_CASE(IA, BB, IB, PM)                                                                              \
    if (wide)                                                                                                        \
        hipLaunchKernelGGL((segment_sum_split_kernel<LPR, SPLIT_WIDE, IA, BB, IB, PM>), dim3((unsigned)grid),        \
                           dim3(256), 0, st, (float4*)out, (const float4*)init, (const float4*)A, ia,                \
                           (const float4*)B, ib, perm, ptr, rows, d4);                                               \
    else if (split)                                                                                                  \
        hipLaunchKernelGGL((segment_sum_split_kernel<LPR, SPLIT, IA, BB, IB, PM>), dim3((unsigned)grid), dim3(256),  \
                           0, st, (float4*)out, (const float4*)init, (const float4*)A, ia, (const float4*)B, ib,     \
                           perm, ptr, rows, d4);                                                                     \
    else                                                                                                             \
        hipLaunchKernelGGL((segment_sum_kernel<LPR, IA, BB, IB, PM>), dim3((unsigned)grid), dim3(256), 0, st,        \
                           (float4*)out, (const float4*)init, (const float4*)A, ia, (const float4*)B, ib, perm, ptr, \
                           rows, d4)
    const int key = (ia ? 8 : 0) | (B ? 4 : 0) | ((B && ib) ? 2 : 0) | (perm ? 1 : 0);
    switch (key) {
        case 0: PAMNET_SEG_CASE(false, false, false, false); break;
        case 1: PAMNET_SEG_CASE(false, false, false, true); break;
        case 4: PAMNET_SEG_CASE(false, true, false, false); break;
        case 5: PAMNET_SEG_CASE(false, true, false, true); break;
        case 6: PAMNET_SEG_CASE(false, true, true, false); break;
        case 7: PAMNET_SEG_CASE(false, true, true, true); break;
        case 8: PAMNET_SEG_CASE(true, false, false, false); break;
        case 9: PAMNET_SEG_CASE(true, false, false, true); break;
        case 12: PAMNET_SEG_CASE(true, true, false, false); break;
        case 13: PAMNET_SEG_CASE(true, true, false, true); break;
        case 14: PAMNET_SEG_CASE(true, true, true, false); break;
        case 15: PAMNET_SEG_CASE(true, true, true, true); break;
        default: return PAMNET_EINVAL;
    }
#undef PAMNET_SEG_CASE
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

// Up to 4 plain segment sums (d = 128) over the same number of output rows in one launch (blockIdx.y = job): the
// backward of a layer reduces the same edge gradients by target and by source (transposed CSR) into node planes --
// 2 (global) / 4 (local) launches become one.
struct MultiSeg {
    float4* out[4];
    const float4* A[4];
    const int32_t* perm[4];     // nullable per job
    const int32_t* ptr[4];
};
__global__ __launch_bounds__(256) void segment_sum_multi_kernel(MultiSeg js, int64_t rows) {
    constexpr int LPR = 32, SPLIT = 4, SLOTS = 8, RPB = 2;
    __shared__ float4 red[SLOTS][LPR];
    float4* __restrict__ out = js.out[blockIdx.y];
    const float4* __restrict__ A = js.A[blockIdx.y];
    const int32_t* __restrict__ perm = js.perm[blockIdx.y];
    const int32_t* __restrict__ ptr = js.ptr[blockIdx.y];
    const int c = threadIdx.x % LPR;
    const int slot = threadIdx.x / LPR;
    const int part = slot % SPLIT, rsub = slot / SPLIT;
    for (int64_t r0 = (int64_t)blockIdx.x * RPB; r0 < rows; r0 += (int64_t)gridDim.x * RPB) {
        const int64_t r = r0 + rsub;
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
        if (r < rows) {
            const int64_t beg = ptr[r], end = ptr[r + 1];
            const int64_t per = (end - beg + SPLIT - 1) / SPLIT;
            int64_t q = beg + part * per;
            const int64_t stop = q + per < end ? q + per : end;
            if (perm) {
                for (; q + 4 <= stop; q += 4) {
                    const int64_t k0 = perm[q], k1 = perm[q + 1], k2 = perm[q + 2], k3 = perm[q + 3];
                    acc4(s0, A[k0 * LPR + c]); acc4(s1, A[k1 * LPR + c]); acc4(s2, A[k2 * LPR + c]); acc4(s3, A[k3 * LPR + c]);
                }
                for (; q < stop; ++q) acc4(s0, A[(int64_t)perm[q] * LPR + c]);
            } else {
                for (; q + 4 <= stop; q += 4) {
                    acc4(s0, A[q * LPR + c]); acc4(s1, A[(q + 1) * LPR + c]); acc4(s2, A[(q + 2) * LPR + c]); acc4(s3, A[(q + 3) * LPR + c]);
                }
                for (; q < stop; ++q) acc4(s0, A[q * LPR + c]);
            }
            acc4(s0, s1); acc4(s2, s3); acc4(s0, s2);
        }
        red[slot][c] = s0;
        __syncthreads();
        if (part == 0 && r < rows) {
            float4 t = red[rsub * SPLIT][c];
#pragma unroll
            for (int p = 1; p < SPLIT; ++p) acc4(t, red[rsub * SPLIT + p][c]);
            out[r * LPR + c] = t;
        }
        __syncthreads();
    }
}

// out1 = A[ia] * B1, out2 = A[ia] * B2 (one gather of A feeding two products), d = 128
__global__ __launch_bounds__(256) void gather_mul2_kernel(float4* __restrict__ out1, float4* __restrict__ out2,
                                                          const float4* __restrict__ A, const int32_t* __restrict__ ia,
                                                          const float4* __restrict__ B1, const float4* __restrict__ B2,
                                                          int64_t m) {
    const int64_t total = m * 32;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = t >> 5;
        const int c = (int)(t & 31);
        const float4 a = A[(int64_t)ia[k] * 32 + c], b1 = B1[t], b2 = B2[t];
        out1[t] = make_float4(a.x * b1.x, a.y * b1.y, a.z * b1.z, a.w * b1.w);
        out2[t] = make_float4(a.x * b2.x, a.y * b2.y, a.z * b2.z, a.w * b2.w);
    }
}

}  // namespace

// njobs <= 4 plain segment sums out_j[r,:] = sum_{q in [ptr_j[r], ptr_j[r+1])} A_j[perm_j ? perm_j[q] : q, :], d = 128.
// Summation order inside a row: four contiguous quarters of the segment, added in order (deterministic).
extern "C" int pamnet_segment_sum_multi_f32(int64_t njobs, float* const* out, const float* const* A,
                                            const int32_t* const* perm, const int32_t* const* ptr, int64_t rows,
                                            int64_t d, pamnet_stream_t stream) {
    if (njobs < 1 || njobs > 4 || rows < 0 || d != 128) return PAMNET_EINVAL;
    if (rows == 0) return PAMNET_OK;
    if (!out || !A || !perm || !ptr) return PAMNET_ENULL;
    MultiSeg js;
    for (int j = 0; j < 4; ++j) {
        const int s = j < njobs ? j : 0;
        if (!out[s] || !ptr[s]) return PAMNET_ENULL;
        js.out[j] = (float4*)out[s];
        js.A[j] = (const float4*)A[s];
        js.perm[j] = perm[s];
        js.ptr[j] = ptr[s];
    }
    int64_t grid = ceil_div(rows, 2);
    if (grid > 256 * 32) grid = 256 * 32;
    hipLaunchKernelGGL(segment_sum_multi_kernel, dim3((unsigned)grid, (unsigned)njobs), dim3(256), 0, as_stream(stream),
                       js, rows);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

// out1[k,:] = A[ia[k],:] * B1[k,:],  out2[k,:] = A[ia[k],:] * B2[k,:]   (d = 128)
extern "C" int pamnet_gather_mul2_f32(float* out1, float* out2, const float* A, const int32_t* ia, const float* B1,
                                      const float* B2, int64_t m, int64_t d, pamnet_stream_t stream) {
    if (m < 0 || d != 128) return PAMNET_EINVAL;
    if (m == 0) return PAMNET_OK;
    if (!out1 || !out2 || !A || !ia || !B1 || !B2) return PAMNET_ENULL;
    int64_t grid = ceil_div(m * 32, 256);
    if (grid > 256 * 64) grid = 256 * 64;
    hipLaunchKernelGGL(gather_mul2_kernel, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), (float4*)out1,
                       (float4*)out2, (const float4*)A, ia, (const float4*)B1, (const float4*)B2, m);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_segment_sum_f32(float* out, const float* init, const float* A, const int32_t* ia,
                                      const float* B, const int32_t* ib, const int32_t* perm, const int32_t* ptr,
                                      int64_t rows, int64_t d, pamnet_stream_t stream) {
    if (rows < 0 || d <= 0 || (d & 3)) return PAMNET_EINVAL;
    if (rows == 0) return PAMNET_OK;
    if (!out || !ptr) return PAMNET_ENULL;          // A may be null when every segment is empty (m == 0)
    hipStream_t st = as_stream(stream);
    const int64_t d4 = d / 4;
    switch (d4) {
        case 1: return launch_segment_sum<1>(out, init, A, ia, B, ib, perm, ptr, rows, d4, st);
        case 2: return launch_segment_sum<2>(out, init, A, ia, B, ib, perm, ptr, rows, d4, st);
        case 4: return launch_segment_sum<4>(out, init, A, ia, B, ib, perm, ptr, rows, d4, st);
        case 8: return launch_segment_sum<8>(out, init, A, ia, B, ib, perm, ptr, rows, d4, st);
        case 16: return launch_segment_sum<16>(out, init, A, ia, B, ib, perm, ptr, rows, d4, st);
        case 32: return launch_segment_sum<32>(out, init, A, ia, B, ib, perm, ptr, rows, d4, st);
        case 64: return launch_segment_sum<64>(out, init, A, ia, B, ib, perm, ptr, rows, d4, st);
        default: break;
    }
    int64_t grid = ceil_div(rows, 4);
    if (grid > 256 * 32) grid = 256 * 32;
    hipLaunchKernelGGL(segment_sum_generic_kernel, dim3((unsigned)grid), dim3(256), 0, st, (float4*)out,
                       (const float4*)init, (const float4*)A, ia, (const float4*)B, ib, perm, ptr, rows, d4);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_gather_mul_f32(float* out, const float* A, const int32_t* ia, const float* B,
                                     const int32_t* ib, int64_t m, int64_t d, pamnet_stream_t stream) {
    if (m < 0 || d <= 0 || (d & 3)) return PAMNET_EINVAL;
    if (m == 0) return PAMNET_OK;
    if (!out || !A) return PAMNET_ENULL;
    hipStream_t st = as_stream(stream);
    const int64_t d4 = d / 4;
    int64_t grid = ceil_div(m * d4, 256);
    if (grid > 256 * 64) grid = 256 * 64;
#define PAMNET_GM_CASE(IA, BB, IB)                                                                               \
    hipLaunchKernelGGL((gather_mul_kernel<IA, BB, IB>), dim3((unsigned)grid), dim3(256), 0, st, (float4*)out,    \
                       (const float4*)A, ia, (const float4*)B, ib, m, d4)
    const int key = (ia ? 4 : 0) | (B ? 2 : 0) | ((B && ib) ? 1 : 0);
    switch (key) {
        case 0: PAMNET_GM_CASE(false, false, false); break;
        case 2: PAMNET_GM_CASE(false, true, false); break;
        case 3: PAMNET_GM_CASE(false, true, true); break;
        case 4: PAMNET_GM_CASE(true, false, false); break;
        case 6: PAMNET_GM_CASE(true, true, false); break;
        case 7: PAMNET_GM_CASE(true, true, true); break;
        default: return PAMNET_EINVAL;
    }
#undef PAMNET_GM_CASE
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}
