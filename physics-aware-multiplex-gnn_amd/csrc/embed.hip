// Input-embedding layers of PAMNet (dim = 128 outputs, tiny K):
//   edge_attr_rbf_{l,g} = SiLU(Linear(16 -> d)(rbf))                       models.py:185-186, layers/basic.py:19-22
//   edge_attr_sbf{1,2}  = SiLU(Linear(42 -> d)(sbf))  (mlp_sbf1 on pair rows, mlp_sbf2 on triplet rows)  models.py:187-188
//   x = init_linear(x_raw[:, 3:])   (18 -> d, no bias, no activation; PDBbind)                           models.py:119
// K is 16 / 18 / 42: too thin for a matrix-core tile to pay off, so this is a VALU kernel shaped for bandwidth:
// thread t owns output column c = t & 127 with its whole weight row (K floats) in registers; a workgroup stages 64
// input rows in LDS and reads them as wave-wide broadcasts; outputs leave as coalesced 512-byte rows.  The combined
// triplet/pair row list selects between two weight sets per row (`kind`), which replaces the reference's two separate
// Linear calls + the index_select / index_copy traffic of a host-side split.
// Backward: z is recomputed (cheaper than storing it), dW / db accumulate per thread across a grid-stride loop and are
// reduced over workgroups in a fixed order (deterministic); dx (only needed for the trainable Bessel frequencies) is a
// second LDS phase.
#include "common.h"

namespace {

constexpr int DOUT = 128;
constexpr int TR = 64;                 // rows per tile

__device__ __forceinline__ float sigmoid_fast(float z) { return __frcp_rn(1.0f + __expf(-z)); }

template <int K>
__device__ __forceinline__ void stage_rows(const float* __restrict__ x, int64_t row0, int64_t rows, float* xs) {
    // TR x K floats, contiguous in memory (row stride K): coalesced linear copy, zero padded
    const int64_t base = row0 * K;
    const int64_t lim = rows * K;
    for (int i = threadIdx.x; i < TR * K; i += 256) xs[i] = (base + i < lim) ? x[base + i] : 0.f;
}

template <int K>
__global__ __launch_bounds__(256) void embed_fwd_kernel(const float* __restrict__ x, int64_t rows,
                                                        const int32_t* __restrict__ kind,
                                                        const float* __restrict__ W0, const float* __restrict__ b0,
                                                        const float* __restrict__ W1, const float* __restrict__ b1,
                                                        int act, float* __restrict__ out) {
    __shared__ float xs[TR * K];
    __shared__ int ks[TR];
    const int c = threadIdx.x & 127, half = threadIdx.x >> 7;
    float w0[K], w1[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        w0[k] = W0[c * K + k];
        w1[k] = kind ? W1[c * K + k] : 0.f;
    }
    const float bias0 = b0 ? b0[c] : 0.f, bias1 = (kind && b1) ? b1[c] : 0.f;
    const int64_t ntiles = (rows + TR - 1) / TR;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row0 = tile * TR;
        __syncthreads();
        stage_rows<K>(x, row0, rows, xs);
        if (threadIdx.x < TR) ks[threadIdx.x] = (kind && row0 + threadIdx.x < rows) ? kind[row0 + threadIdx.x] : 0;
        __syncthreads();
        for (int r = half * (TR / 2); r < (half + 1) * (TR / 2); ++r) {
            const int64_t g = row0 + r;
            if (g >= rows) break;
            float z;
            if (ks[r] == 0) {
                z = bias0;
#pragma unroll
                for (int k = 0; k < K; ++k) z = fmaf(w0[k], xs[r * K + k], z);
            } else {
                z = bias1;
#pragma unroll
                for (int k = 0; k < K; ++k) z = fmaf(w1[k], xs[r * K + k], z);
            }
            out[g * DOUT + c] = act ? z * sigmoid_fast(z) : z;
        }
    }
}

// partial layout per workgroup: [2 kinds][128][K] dW, then [2][128] db
template <int K, bool DX>
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* __restrict__ x, int64_t rows,
                                                        const int32_t* __restrict__ kind,
                                                        const float* __restrict__ W0, const float* __restrict__ b0,
                                                        const float* __restrict__ W1, const float* __restrict__ b1,
                                                        int act, const float* __restrict__ gout,
                                                        float* __restrict__ partial, float* __restrict__ dx) {
    __shared__ float xs[TR * K];
    __shared__ int ks[TR];
    constexpr int RED = DOUT * (2 * K + 2);           // final cross-half reduction scratch
    constexpr int DZ = DX ? TR * (DOUT + 1) : 0;      // dz tile for the dx phase
    __shared__ float dzs[RED > DZ ? RED : DZ];
    __shared__ float ws[DX ? DOUT * K : 1];           // W0 for the dx phase (single-kind layers only)
    const int c = threadIdx.x & 127, half = threadIdx.x >> 7;
    float w0[K], w1[K], g0[K], g1[K];
    float gb0 = 0.f, gb1 = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        w0[k] = W0[c * K + k];
        w1[k] = kind ? W1[c * K + k] : 0.f;
        g0[k] = 0.f;
        g1[k] = 0.f;
    }
    if (DX)
        for (int i = threadIdx.x; i < DOUT * K; i += 256) ws[i] = W0[i];
    const float bias0 = b0 ? b0[c] : 0.f, bias1 = (kind && b1) ? b1[c] : 0.f;
    const int64_t ntiles = (rows + TR - 1) / TR;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row0 = tile * TR;
        __syncthreads();
        stage_rows<K>(x, row0, rows, xs);
        if (threadIdx.x < TR) ks[threadIdx.x] = (kind && row0 + threadIdx.x < rows) ? kind[row0 + threadIdx.x] : 0;
        __syncthreads();
        for (int r = half * (TR / 2); r < (half + 1) * (TR / 2); ++r) {
            const int64_t g = row0 + r;
            float dz = 0.f;
            if (g < rows) {
                const bool k0 = ks[r] == 0;
                float z = k0 ? bias0 : bias1;
                if (k0) {
#pragma unroll
                    for (int k = 0; k < K; ++k) z = fmaf(w0[k], xs[r * K + k], z);
                } else {
#pragma unroll
                    for (int k = 0; k < K; ++k) z = fmaf(w1[k], xs[r * K + k], z);
                }
                dz = gout[g * DOUT + c];
                if (act) {
                    const float s = sigmoid_fast(z);
                    dz *= s * (1.0f + z * (1.0f - s));
                }
                if (k0) {
                    gb0 += dz;
#pragma unroll
                    for (int k = 0; k < K; ++k) g0[k] = fmaf(dz, xs[r * K + k], g0[k]);
                } else {
                    gb1 += dz;
#pragma unroll
                    for (int k = 0; k < K; ++k) g1[k] = fmaf(dz, xs[r * K + k], g1[k]);
                }
            }
            if (DX) dzs[r * (DOUT + 1) + c] = dz;
        }
        if (DX) {
            __syncthreads();
            // dx[row][k] = sum_c dz[row][c] * W0[c][k]
            for (int o = threadIdx.x; o < TR * K; o += 256) {
                const int r = o / K, k = o - r * K;
                const int64_t g = row0 + r;
                if (g >= rows) continue;
                float s = 0.f;
#pragma unroll 8
                for (int cc = 0; cc < DOUT; ++cc) s = fmaf(dzs[r * (DOUT + 1) + cc], ws[cc * K + k], s);
                dx[g * K + k] = s;
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
            red[c * (2 * K + 2) + K + k] = g1[k];
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
            p[DOUT * K + c * K + k] = g1[k] + red[c * (2 * K + 2) + K + k];
        }
        p[2 * DOUT * K + c] = gb0 + red[c * (2 * K + 2) + 2 * K];
        p[2 * DOUT * K + DOUT + c] = gb1 + red[c * (2 * K + 2) + 2 * K + 1];
    }
}

// out element e of [2][128][K] + [2][128]: fixed-order sum over the workgroup partials
__global__ __launch_bounds__(256) void embed_reduce_kernel(const float* __restrict__ partial, int nblocks, int K,
                                                           float* __restrict__ dW0, float* __restrict__ db0,
                                                           float* __restrict__ dW1, float* __restrict__ db1) {
    const int per = 2 * DOUT * K + 2 * DOUT;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= per) return;
    float s = 0.f;
    for (int b = 0; b < nblocks; ++b) s += partial[(int64_t)b * per + e];
    if (e < DOUT * K) dW0[e] = s;
    else if (e < 2 * DOUT * K) { if (dW1) dW1[e - DOUT * K] = s; }
    else if (e < 2 * DOUT * K + DOUT) { if (db0) db0[e - 2 * DOUT * K] = s; }
    else if (db1) db1[e - 2 * DOUT * K - DOUT] = s;
}

inline int grid_for(int64_t rows) {
    const int64_t tiles = (rows + TR - 1) / TR;
    return (int)(tiles < 1 ? 1 : (tiles > 512 ? 512 : tiles));
}

template <int K>
int launch_fwd(const float* x, int64_t rows, const int32_t* kind, const float* W0, const float* b0, const float* W1,
               const float* b1, int act, float* out, hipStream_t st) {
    hipLaunchKernelGGL((embed_fwd_kernel<K>), dim3(grid_for(rows)), dim3(256), 0, st, x, rows, kind, W0, b0, W1, b1, act,
                       out);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

template <int K>
int launch_bwd(const float* x, int64_t rows, const int32_t* kind, const float* W0, const float* b0, const float* W1,
               const float* b1, int act, const float* gout, float* dW0, float* db0, float* dW1, float* db1, float* dx,
               float* partial, hipStream_t st) {
    const int nb = grid_for(rows);
    if (dx) {
        if constexpr (K == 16)
            hipLaunchKernelGGL((embed_bwd_kernel<K, true>), dim3(nb), dim3(256), 0, st, x, rows, kind, W0, b0, W1, b1,
                               act, gout, partial, dx);
        else
            return PAMNET_EINVAL;
    } else {
        hipLaunchKernelGGL((embed_bwd_kernel<K, false>), dim3(nb), dim3(256), 0, st, x, rows, kind, W0, b0, W1, b1, act,
                           gout, partial, dx);
    }
    PAMNET_LAUNCH_CHECK();
    const int per = 2 * DOUT * K + 2 * DOUT;
    hipLaunchKernelGGL(embed_reduce_kernel, dim3((per + 255) / 256), dim3(256), 0, st, partial, nb, K, dW0, db0, dW1,
                       db1);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

}  // namespace

// scratch floats for the backward: grid * (2*128*K + 2*128)
extern "C" int pamnet_embed_scratch_floats(int64_t rows, int64_t K, int64_t* floats) {
    if (rows < 0 || K <= 0 || !floats) return PAMNET_EINVAL;
    *floats = (int64_t)grid_for(rows) * (2 * DOUT * K + 2 * DOUT);
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
    if (!W0 || !dW0 || !partial || (kind && (!W1 || !dW1))) return PAMNET_ENULL;
    if (dx && (kind || K != 16)) return PAMNET_EINVAL;     // input gradients only for the single-kind K = 16 (rbf) layer
    hipStream_t st = as_stream(stream);
    if (K == 16) return launch_bwd<16>(x, rows, kind, W0, b0, W1, b1, act, gout, dW0, db0, dW1, db1, dx, partial, st);
    if (K == 18) return launch_bwd<18>(x, rows, kind, W0, b0, W1, b1, act, gout, dW0, db0, dW1, db1, dx, partial, st);
    return launch_bwd<42>(x, rows, kind, W0, b0, W1, b1, act, gout, dW0, db0, dW1, db1, dx, partial, st);
}
