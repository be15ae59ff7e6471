// Small whole-buffer reductions of the training step, fixed summation order (deterministic):
//   * pamnet_sumsq_partials_f32 : 256 fp64 partial sums of squares of the flat gradient (clip_grad_norm_,
//                                 main_qm9.py:111); the optimiser kernel (optim.hip) adds them itself -- no finish launch
//   * pamnet_l1_loss_f32 / pamnet_mse_loss_f32 / pamnet_smooth_l1_loss_f32 : the three drivers' losses (main_qm9.py:108,
//                                 main_pdbbind.py:93, main_rna_puzzles.py:92) with their gradient w.r.t. out, one launch
//   * pamnet_type_rows_grad_f32 : gradient of `embeddings[x]` (models.py:107,140): rows of d x summed per atom type
//                                 (partials per workgroup + a one-workgroup finish, workgroup order)
// (A single launch with a "last workgroup finishes" counter was measured and rejected: the device-scope fence it needs
// writes back the whole L2 on this multi-XCD part -- 40 us for the norm against 7.5 us for the partials launch.)
#include "common.h"
#include "type_rows_core.h"

namespace {

constexpr int NORM_BLOCKS = 256;

__global__ __launch_bounds__(256) void sumsq_partials_kernel(const float4* __restrict__ g, int64_t n4,
                                                             double* __restrict__ partial) {
    __shared__ double red[256];
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    const int64_t per = (n4 + gridDim.x - 1) / gridDim.x;
    const int64_t beg = blockIdx.x * per, end = beg + per < n4 ? beg + per : n4;
    int64_t i = beg + threadIdx.x;
    for (; i + 256 < end; i += 512) {                  // two independent 16-byte loads in flight per lane
        const float4 u = g[i], v = g[i + 256];
        a0.x = fmaf(u.x, u.x, a0.x), a0.y = fmaf(u.y, u.y, a0.y), a0.z = fmaf(u.z, u.z, a0.z), a0.w = fmaf(u.w, u.w, a0.w);
        a1.x = fmaf(v.x, v.x, a1.x), a1.y = fmaf(v.y, v.y, a1.y), a1.z = fmaf(v.z, v.z, a1.z), a1.w = fmaf(v.w, v.w, a1.w);
    }
    if (i < end) {
        const float4 u = g[i];
        a0.x = fmaf(u.x, u.x, a0.x), a0.y = fmaf(u.y, u.y, a0.y), a0.z = fmaf(u.z, u.z, a0.z), a0.w = fmaf(u.w, u.w, a0.w);
    }
    red[threadIdx.x] = ((double)a0.x + (double)a0.y) + ((double)a0.z + (double)a0.w) + ((double)a1.x + (double)a1.y) +
                       ((double)a1.z + (double)a1.w);
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// one workgroup: loss = mean_i l(out_i - y_i) ; d_out_i = l'(out_i - y_i) * grad_scale / n     (n graphs: a few hundred)
//   KIND 0: l(d) = |d|                                   F.l1_loss        (main_qm9.py:108)
//   KIND 1: l(d) = d^2                                   F.mse_loss       (main_pdbbind.py:93)
//   KIND 2: l(d) = d^2 / 2 if |d| < 1 else |d| - 1/2     F.smooth_l1_loss (main_rna_puzzles.py:92; beta = 1)
template <int KIND>
__global__ __launch_bounds__(256) void loss_kernel(const float* __restrict__ out, const float* __restrict__ y, int64_t n,
                                                   float grad_scale, float* __restrict__ loss, float* __restrict__ d_out) {
    __shared__ double red[256];
    double s = 0.0;
    const float gs = grad_scale / (float)n;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const float d = out[i] - y[i];
        const float a = fabsf(d);
        float l, dl;
        if (KIND == 0) {
            l = a, dl = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        } else if (KIND == 1) {
            l = d * d, dl = 2.f * d;
        } else {
            l = a < 1.f ? 0.5f * d * d : a - 0.5f, dl = a < 1.f ? d : (d > 0.f ? 1.f : -1.f);
        }
        s += (double)l;
        if (d_out) d_out[i] = dl * gs;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = (float)(red[0] / (double)n);
}

template <int KIND>
int loss_launch(const float* out, const float* y, int64_t n, float grad_scale, float* loss, float* d_out,
                pamnet_stream_t stream) {
    if (n < 1) return PAMNET_EINVAL;
    if (!out || !y || !loss) return PAMNET_ENULL;
    hipLaunchKernelGGL(loss_kernel<KIND>, dim3(1), dim3(256), 0, as_stream(stream), out, y, n, grad_scale, loss, d_out);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

using type_rows::TYPE_BLOCKS;
using type_rows::TYPE_MAX;

__global__ __launch_bounds__(256) void type_rows_grad_kernel(const float4* __restrict__ g, const int32_t* __restrict__ idx,
                                                             int64_t n, int n_types, int d4, float4* __restrict__ partial) {
    __shared__ float4 red[256];
    type_rows::grad_body(g, idx, n, n_types, d4, partial, blockIdx.x, gridDim.x, red);
}

__global__ __launch_bounds__(256) void type_rows_finish_kernel(const float4* __restrict__ partial, int blocks, int cells,
                                                               float4* __restrict__ out) {
    type_rows::finish_body(partial, blocks, cells, out);
}

}  // namespace

// scratch (caller-owned, device) of pamnet_type_rows_grad_f32: pamnet_reduce_scratch_bytes bytes, no initialisation needed
typedef float f32x4 __attribute__((ext_vector_type(4)));
// A plain 16-byte-per-lane copy, 8 loads in flight per lane: what this box's memory system delivers to the simplest streaming
// kernel (bench.py's roofline calibration; not part of any model path).
__global__ __launch_bounds__(256) void stream_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int64_t n4) {
    constexpr int U = 8;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + i + u * stride));
            v[u] = make_float4(t[0], t[1], t[2], t[3]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const f32x4 t = {v[u].x, v[u].y, v[u].z, v[u].w};
            __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(dst + i + u * stride));
        }
    }
    for (; i < n4; i += stride) dst[i] = src[i];
}
extern "C" int pamnet_stream_copy_f32(const float* src, float* dst, int64_t n, pamnet_stream_t stream) {
    if (n < 0 || (n & 3)) return PAMNET_EINVAL;
    if (n == 0) return PAMNET_OK;
    if (!src || !dst) return PAMNET_ENULL;
    const int64_t n4 = n / 4;
    int64_t blocks = (n4 + 256 * 8 - 1) / (256 * 8);
    blocks = blocks < 1 ? 1 : (blocks > 256 * 16 ? 256 * 16 : blocks);
    hipLaunchKernelGGL(stream_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float4*>(src), reinterpret_cast<float4*>(dst), n4);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_reduce_scratch_bytes(int64_t* bytes) {
    if (!bytes) return PAMNET_ENULL;
    *bytes = (int64_t)TYPE_BLOCKS * TYPE_MAX * 64 * 16;
    return PAMNET_OK;
}

// partials[0:256] (fp64) = sums of squares of 256 contiguous slices of g[0:n] (unused slices: 0).  n % 4 == 0.
// || g ||_2 = sqrt(sum of the partials in index order) -- pamnet_adam_ema_norm_f32 does that sum itself.
extern "C" int pamnet_sumsq_partials_f32(const float* g, int64_t n, double* partials, pamnet_stream_t stream) {
    if (n < 0 || (n & 3)) return PAMNET_EINVAL;
    if (!g || !partials) return PAMNET_ENULL;
    hipLaunchKernelGGL(sumsq_partials_kernel, dim3(NORM_BLOCKS), dim3(256), 0, as_stream(stream), (const float4*)g, n / 4,
                       partials);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

// loss[0] = mean_i |out[i] - y[i]|;  d_out[i] (nullable) = grad_scale * sign(out[i] - y[i]) / n   (sign(0) = 0 as torch)
extern "C" int pamnet_l1_loss_f32(const float* out, const float* y, int64_t n, float grad_scale, float* loss,
                                  float* d_out, pamnet_stream_t stream) {
    return loss_launch<0>(out, y, n, grad_scale, loss, d_out, stream);
}

// loss[0] = mean_i (out[i] - y[i])^2;  d_out[i] (nullable) = grad_scale * 2 (out[i] - y[i]) / n      (main_pdbbind.py:93)
extern "C" int pamnet_mse_loss_f32(const float* out, const float* y, int64_t n, float grad_scale, float* loss,
                                   float* d_out, pamnet_stream_t stream) {
    return loss_launch<1>(out, y, n, grad_scale, loss, d_out, stream);
}

// loss[0] = mean_i h(out[i] - y[i]), h(d) = d^2/2 for |d| < 1, |d| - 1/2 otherwise (beta = 1);  d_out[i] (nullable) =
// grad_scale * clamp(out[i] - y[i], -1, 1) / n                                                  (main_rna_puzzles.py:92)
extern "C" int pamnet_smooth_l1_loss_f32(const float* out, const float* y, int64_t n, float grad_scale, float* loss,
                                         float* d_out, pamnet_stream_t stream) {
    return loss_launch<2>(out, y, n, grad_scale, loss, d_out, stream);
}

// out[t, :] = sum_{r < n: idx[r] = t} g[r, :]  for t < n_types (<= 8), d in {16, 32, 64, 128, 256}; idx values outside
// [0, n_types) contribute nothing.
extern "C" int pamnet_type_rows_grad_f32(const float* g, const int32_t* idx, int64_t n, int64_t n_types, int64_t d,
                                         void* scratch, float* out, pamnet_stream_t stream) {
    const int64_t d4 = d / 4;
    if (n < 0 || n_types < 1 || n_types > TYPE_MAX || d < 4 || (d & 3) || d4 > 64 || (d4 & (d4 - 1))) return PAMNET_EINVAL;
    if (!out || !scratch || (n > 0 && (!g || !idx))) return PAMNET_ENULL;
    const int blocks = type_rows::blocks_for_rows(n);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(type_rows_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const float4*)g, idx, n,
                       (int)n_types, (int)d4, static_cast<float4*>(scratch));
    hipLaunchKernelGGL(type_rows_finish_kernel, dim3(1), dim3(256), 0, st, static_cast<const float4*>(scratch), (int)blocks,
                       (int)(n_types * d4), (float4*)out);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}
