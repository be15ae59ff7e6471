// Input-embedding layers of PAMNet (dim = 128 outputs, tiny K):
//   edge_attr_rbf_{l,g} = SiLU(Linear(16 -> d)(rbf))                       models.py:185-186, layers/basic.py:19-22
//   edge_attr_sbf{1,2}  = SiLU(Linear(42 -> d)(sbf))  (mlp_sbf1 on pair rows, mlp_sbf2 on triplet rows)  models.py:187-188
//   x = init_linear(x_raw[:, 3:])   (18 -> d, no bias, no activation; PDBbind)                           models.py:119
// K is 16 / 18 / 42: too thin for a matrix-core tile to pay off, so this is a VALU kernel shaped for bandwidth:
// thread t owns output column c = t & 127 with its whole weight row (K floats) in registers; a workgroup stages 32
// input rows in LDS and reads them as wave-wide broadcasts; outputs leave as coalesced 512-byte rows.  The combined
// triplet/pair row list selects between two weight sets per row (`kind`), which replaces the reference's two separate
// Linear calls + the index_select / index_copy traffic of a host-side split.
// Backward: z is recomputed (cheaper than storing it), dW / db accumulate per thread across a grid-stride loop and are
// reduced over workgroups in a fixed order (deterministic); dx (only needed for the trainable Bessel frequencies) is a
// second LDS phase.
#include "common.h"

namespace {

constexpr int DOUT = 128;
constexpr int TR = 32;                 // rows per tile
constexpr int RU = 4;                  // rows in flight per thread

__device__ __forceinline__ float sigmoid_fast(float z) { return __frcp_rn(1.0f + __expf(-z)); }

template <int K>
__device__ __forceinline__ void stage_rows(const float* __restrict__ x, int64_t row0, int64_t rows, float* xs) {
    // TR x K floats, contiguous in memory (row stride K): coalesced linear copy, zero padded
    const int64_t base = row0 * K;
    const int64_t lim = rows * K;
    for (int i = threadIdx.x; i < TR * K; i += 256) xs[i] = (base + i < lim) ? x[base + i] : 0.f;
}

template <int K, bool TWO>
__global__ __launch_bounds__(256) void embed_fwd_kernel(const float* __restrict__ x, int64_t rows,
                                                        const int32_t* __restrict__ kind,
                                                        const float* __restrict__ W0, const float* __restrict__ b0,
                                                        const float* __restrict__ W1, const float* __restrict__ b1,
                                                        int act, float* __restrict__ out) {
    __shared__ float xs[TR * K];
    __shared__ int ks[TR];
    const int c = threadIdx.x & 127, half = threadIdx.x >> 7;
    float w0[K], w1[TWO ? K : 1];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        w0[k] = W0[c * K + k];
        if (TWO) w1[k] = W1[c * K + k];
    }
    const float bias0 = b0 ? b0[c] : 0.f, bias1 = (TWO && b1) ? b1[c] : 0.f;
    const int64_t ntiles = (rows + TR - 1) / TR;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row0 = tile * TR;
        __syncthreads();
        stage_rows<K>(x, row0, rows, xs);
        if (TWO && threadIdx.x < TR) ks[threadIdx.x] = (row0 + threadIdx.x < rows) ? kind[row0 + threadIdx.x] : 0;
        __syncthreads();
        // RU rows in flight per thread: independent accumulators hide the LDS-broadcast + FMA latency
        for (int r0 = half * (TR / 2); r0 < (half + 1) * (TR / 2); r0 += RU) {
            float z[RU];
            bool k1[RU];
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                k1[u] = TWO && ks[r0 + u] != 0;
                z[u] = k1[u] ? bias1 : bias0;
            }
#pragma unroll
            for (int k = 0; k < K; ++k) {
#pragma unroll
                for (int u = 0; u < RU; ++u) {
                    const float w = TWO ? (k1[u] ? w1[k] : w0[k]) : w0[k];
                    z[u] = fmaf(w, xs[(r0 + u) * K + k], z[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int64_t g = row0 + r0 + u;
                if (g < rows) out[g * DOUT + c] = act ? z[u] * sigmoid_fast(z[u]) : z[u];
            }
        }
    }
}

// partial layout per workgroup: [2 kinds][128][K] dW, then [2][128] db
template <int K, bool TWO, bool DX>
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* __restrict__ x, int64_t rows,
                                                        const int32_t* __restrict__ kind,
                                                        const float* __restrict__ W0, const float* __restrict__ b0,
                                                        const float* __restrict__ W1, const float* __restrict__ b1,
                                                        int act, const float* __restrict__ gout,
                                                        float* __restrict__ partial, float* __restrict__ dx) {
    constexpr int RED = DOUT * (2 * K + 2);           // final cross-half reduction scratch
    constexpr int DZ = DX ? TR * (DOUT + 1) : 0;      // dz tile for the dx phase
    __shared__ float xs[TR * K];
    __shared__ int ks[TR];
    __shared__ float dzs[RED > DZ ? RED : DZ];
    __shared__ float ws[DX ? DOUT * K : 1];           // W0 for the dx phase (single-kind layers only)
    const int c = threadIdx.x & 127, half = threadIdx.x >> 7;
    float w0[K], g0[K], w1[TWO ? K : 1], g1[TWO ? K : 1];
    float gb0 = 0.f, gb1 = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        w0[k] = W0[c * K + k];
        g0[k] = 0.f;
        if (TWO) {
            w1[k] = W1[c * K + k];
            g1[k] = 0.f;
        }
    }
    if (DX)
        for (int i = threadIdx.x; i < DOUT * K; i += 256) ws[i] = W0[i];
    const float bias0 = b0 ? b0[c] : 0.f, bias1 = (TWO && b1) ? b1[c] : 0.f;
    const int64_t ntiles = (rows + TR - 1) / TR;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row0 = tile * TR;
        __syncthreads();
        stage_rows<K>(x, row0, rows, xs);
        if (TWO && threadIdx.x < TR) ks[threadIdx.x] = (row0 + threadIdx.x < rows) ? kind[row0 + threadIdx.x] : 0;
        __syncthreads();
        for (int r0 = half * (TR / 2); r0 < (half + 1) * (TR / 2); r0 += RU) {
            float z[RU], dz[RU];
            bool k1[RU];
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int64_t g = row0 + r0 + u;
                k1[u] = TWO && ks[r0 + u] != 0;
                z[u] = k1[u] ? bias1 : bias0;
                dz[u] = g < rows ? gout[g * DOUT + c] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < K; ++k) {
#pragma unroll
                for (int u = 0; u < RU; ++u) {
                    const float w = TWO ? (k1[u] ? w1[k] : w0[k]) : w0[k];
                    z[u] = fmaf(w, xs[(r0 + u) * K + k], z[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                if (act) {
                    const float sg = sigmoid_fast(z[u]);
                    dz[u] *= sg * (1.0f + z[u] * (1.0f - sg));
                }
                if (DX) dzs[(r0 + u) * (DOUT + 1) + c] = dz[u];
                if (k1[u]) gb1 += dz[u];
                else gb0 += dz[u];
            }
#pragma unroll
            for (int k = 0; k < K; ++k) {
#pragma unroll
                for (int u = 0; u < RU; ++u) {
                    const float xv = xs[(r0 + u) * K + k];
                    if (TWO) {
                        g0[k] = fmaf(k1[u] ? 0.f : dz[u], xv, g0[k]);
                        g1[k] = fmaf(k1[u] ? dz[u] : 0.f, xv, g1[k]);
                    } else {
                        g0[k] = fmaf(dz[u], xv, g0[k]);
                    }
                }
            }
        }
        if (DX) {
            __syncthreads();
            // dx[row][k] = sum_c dz[row][c] * W0[c][k]
            for (int o = threadIdx.x; o < TR * K; o += 256) {
                const int r = o / K, k = o - r * K;
                const int64_t g = row0 + r;
                if (g >= rows) continue;
                float s0 = 0.f, s1 = 0.f;
#pragma unroll 8
                for (int cc = 0; cc < DOUT; cc += 2) {
                    s0 = fmaf(dzs[r * (DOUT + 1) + cc], ws[cc * K + k], s0);
                    s1 = fmaf(dzs[r * (DOUT + 1) + cc + 1], ws[(cc + 1) * K + k], s1);
                }
                dx[g * K + k] = s0 + s1;
            }
        }
    }
    // the two row-halves of a column are combined through LDS (reuse dzs), then written as this workgroup's partial
    __syncthreads();
    float* red = dzs;
    if (half == 1) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            red[c * (2 * K + 2) + k] = g0[k];
            if (TWO) red[c * (2 * K + 2) + K + k] = g1[k];
        }
        red[c * (2 * K + 2) + 2 * K] = gb0;
        red[c * (2 * K + 2) + 2 * K + 1] = gb1;
    }
    __syncthreads();
    if (half == 0) {
        float* p = partial + (int64_t)blockIdx.x * (2 * DOUT * K + 2 * DOUT);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            p[c * K + k] = g0[k] + red[c * (2 * K + 2) + k];
            if (TWO) p[DOUT * K + c * K + k] = g1[k] + red[c * (2 * K + 2) + K + k];
        }
        p[2 * DOUT * K + c] = gb0 + red[c * (2 * K + 2) + 2 * K];
        p[2 * DOUT * K + DOUT + c] = gb1 + red[c * (2 * K + 2) + 2 * K + 1];
    }
}

// 32 output elements x 8 slices of the workgroup partials per block; fixed-order tree over the slices (deterministic).
// Elements: [2][128][K] dW then [2][128] db; the second kind is skipped when it has no destination.
__global__ __launch_bounds__(256) void embed_reduce_kernel(const float* __restrict__ partial, int nblocks, int K,
                                                           float* __restrict__ dW0, float* __restrict__ db0,
                                                           float* __restrict__ dW1, float* __restrict__ db1) {
    __shared__ float sm[8][33];
    const int per = 2 * DOUT * K + 2 * DOUT;
    const int lane = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + lane;
    float* dst = nullptr;
    if (e < DOUT * K) dst = dW0 + e;
    else if (e < 2 * DOUT * K) dst = dW1 ? dW1 + (e - DOUT * K) : nullptr;
    else if (e < 2 * DOUT * K + DOUT) dst = db0 ? db0 + (e - 2 * DOUT * K) : nullptr;
    else if (e < per) dst = db1 ? db1 + (e - 2 * DOUT * K - DOUT) : nullptr;
    float s = 0.f;
    if (dst)
        for (int b = slice; b < nblocks; b += 8) s += partial[(int64_t)b * per + e];
    sm[slice][lane] = s;
    __syncthreads();
    if (slice == 0 && dst) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += sm[q][lane];
        *dst = t;
    }
}

inline int grid_for(int64_t rows, int cap) {
    const int64_t tiles = (rows + TR - 1) / TR;
    return (int)(tiles < 1 ? 1 : (tiles > cap ? cap : tiles));
}
constexpr int FWD_CAP = 1024, BWD_CAP = 256;

template <int K>
int launch_fwd(const float* x, int64_t rows, const int32_t* kind, const float* W0, const float* b0, const float* W1,
               const float* b1, int act, float* out, hipStream_t st) {
    if (kind)
        hipLaunchKernelGGL((embed_fwd_kernel<K, true>), dim3(grid_for(rows, FWD_CAP)), dim3(256), 0, st, x, rows, kind,
                           W0, b0, W1, b1, act, out);
    else
        hipLaunchKernelGGL((embed_fwd_kernel<K, false>), dim3(grid_for(rows, FWD_CAP)), dim3(256), 0, st, x, rows, kind,
                           W0, b0, W1, b1, act, out);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

template <int K>
int launch_bwd(const float* x, int64_t rows, const int32_t* kind, const float* W0, const float* b0, const float* W1,
               const float* b1, int act, const float* gout, float* dW0, float* db0, float* dW1, float* db1, float* dx,
               float* partial, hipStream_t st) {
    const int nb = grid_for(rows, BWD_CAP);
    if (dx) {
        if constexpr (K == 16)
            hipLaunchKernelGGL((embed_bwd_kernel<K, false, true>), dim3(nb), dim3(256), 0, st, x, rows, kind, W0, b0,
                               W1, b1, act, gout, partial, dx);
        else
            return PAMNET_EINVAL;
    } else if (W1) {
        hipLaunchKernelGGL((embed_bwd_kernel<K, true, false>), dim3(nb), dim3(256), 0, st, x, rows, kind, W0, b0, W1,
                           b1, act, gout, partial, dx);
    } else {
        hipLaunchKernelGGL((embed_bwd_kernel<K, false, false>), dim3(nb), dim3(256), 0, st, x, rows, kind, W0, b0, W1,
                           b1, act, gout, partial, dx);
    }
    PAMNET_LAUNCH_CHECK();
    const int per = 2 * DOUT * K + 2 * DOUT;
    hipLaunchKernelGGL(embed_reduce_kernel, dim3((per + 31) / 32), dim3(256), 0, st, partial, nb, K, dW0, db0, dW1,
                       db1);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

}  // namespace

// scratch floats for the backward: grid * (2*128*K + 2*128)
extern "C" int pamnet_embed_scratch_floats(int64_t rows, int64_t K, int64_t* floats) {
    if (rows < 0 || K <= 0 || !floats) return PAMNET_EINVAL;
    *floats = (int64_t)grid_for(rows, BWD_CAP) * (2 * DOUT * K + 2 * DOUT);
    return PAMNET_OK;
}

extern "C" int pamnet_embed_fwd_f32(const float* x, int64_t rows, int64_t K, const int32_t* kind, const float* W0,
                                    const float* b0, const float* W1, const float* b1, int32_t act, float* out,
                                    pamnet_stream_t stream) {
    if (rows < 0 || (K != 16 && K != 18 && K != 42)) return PAMNET_EINVAL;
    if (rows == 0) return PAMNET_OK;
    if (!x || !W0 || !out || (kind && !W1)) return PAMNET_ENULL;
    hipStream_t st = as_stream(stream);
    if (K == 16) return launch_fwd<16>(x, rows, kind, W0, b0, W1, b1, act, out, st);
    if (K == 18) return launch_fwd<18>(x, rows, kind, W0, b0, W1, b1, act, out, st);
    return launch_fwd<42>(x, rows, kind, W0, b0, W1, b1, act, out, st);
}

extern "C" int pamnet_embed_bwd_f32(const float* x, int64_t rows, int64_t K, const int32_t* kind, const float* W0,
                                    const float* b0, const float* W1, const float* b1, int32_t act, const float* gout,
                                    float* dW0, float* db0, float* dW1, float* db1, float* dx, float* partial,
                                    pamnet_stream_t stream) {
    if (rows < 0 || (K != 16 && K != 18 && K != 42)) return PAMNET_EINVAL;
    if (rows > 0 && (!x || !gout)) return PAMNET_ENULL;   // rows == 0: gradients are written as zeros
    if (!W0 || !dW0 || !partial || (kind && !W1) || (W1 && (!dW1 || (rows > 0 && !kind)))) return PAMNET_ENULL;
    if (dx && (kind || K != 16)) return PAMNET_EINVAL;     // input gradients only for the single-kind K = 16 (rbf) layer
    hipStream_t st = as_stream(stream);
    if (K == 16) return launch_bwd<16>(x, rows, kind, W0, b0, W1, b1, act, gout, dW0, db0, dW1, db1, dx, partial, st);
    if (K == 18) return launch_bwd<18>(x, rows, kind, W0, b0, W1, b1, act, gout, dW0, db0, dW1, db1, dx, partial, st);
    return launch_bwd<42>(x, rows, kind, W0, b0, W1, b1, act, gout, dW0, db0, dW1, db1, dx, partial, st);
}
