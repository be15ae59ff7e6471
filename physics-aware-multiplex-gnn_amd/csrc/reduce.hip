// Small whole-buffer reductions of the training step, each ONE launch with a fixed summation order (deterministic):
//   * pamnet_grad_norm_f32   : L2 norm of the flat gradient (clip_grad_norm_, main_qm9.py:111) -- 14 MB, HBM-bound
//   * pamnet_l1_loss_f32     : F.l1_loss(out, y) and its gradient w.r.t. out (main_qm9.py:108), a few hundred floats
//   * pamnet_type_rows_grad_f32 : gradient of `embeddings[x]` (models.py:107,140): rows of d x summed per atom type
// Pattern: every workgroup writes a partial, the LAST one to finish (device counter, reset for the next call) adds the
// partials in workgroup order -- the order of arrival only decides who does the final sum, never its value.
#include "common.h"

namespace {

__device__ __forceinline__ bool last_block(unsigned* counter) {
    __shared__ bool last;
    __threadfence();                                   // this block's partial is visible device-wide
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned done = atomicAdd(counter, 1u);
        last = done + 1 == gridDim.x;
        if (last) *counter = 0;                        // ready for the next call on this stream
    }
    __syncthreads();
    __threadfence();
    return last;
}

constexpr int NORM_BLOCKS = 256;

__global__ __launch_bounds__(256) void grad_norm_kernel(const float4* __restrict__ g, int64_t n4, double* __restrict__ partial,
                                                        unsigned* __restrict__ counter, float* __restrict__ norm_out) {
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
    if (!last_block(counter)) return;
    red[threadIdx.x] = threadIdx.x < gridDim.x ? partial[threadIdx.x] : 0.0;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) norm_out[0] = (float)sqrt(red[0]);
}

// one workgroup: loss = mean |out - y| ; d_out = sign(out - y) * grad_scale / n     (n graphs: a few hundred)
__global__ __launch_bounds__(256) void l1_loss_kernel(const float* __restrict__ out, const float* __restrict__ y, int64_t n,
                                                      float grad_scale, float* __restrict__ loss,
                                                      float* __restrict__ d_out) {
    __shared__ double red[256];
    double s = 0.0;
    const float gs = grad_scale / (float)n;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const float d = out[i] - y[i];
        s += (double)fabsf(d);
        if (d_out) d_out[i] = d > 0.f ? gs : (d < 0.f ? -gs : 0.f);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = (float)(red[0] / (double)n);
}

constexpr int TYPE_MAX = 8, TYPE_BLOCKS = 64;

// out[t, :] = sum_{r: idx[r] = t} g[r, :],  t < n_types <= 8.  Lane group (d4 lanes) per row slot; a workgroup owns a
// contiguous slice of rows, slot s walks rows s, s + slots, ... (fixed), slots meet in LDS in slot order, workgroups in
// the last block in workgroup order.
__global__ __launch_bounds__(256) void type_rows_grad_kernel(const float4* __restrict__ g, const int32_t* __restrict__ idx,
                                                             int64_t n, int n_types, int d4, float4* __restrict__ partial,
                                                             unsigned* __restrict__ counter, float4* __restrict__ out) {
    __shared__ float4 red[256];
    const int c = threadIdx.x % d4, slot = threadIdx.x / d4, slots = 256 / d4;
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t beg = blockIdx.x * per, end = beg + per < n ? beg + per : n;
    float4 acc[TYPE_MAX];
#pragma unroll
    for (int t = 0; t < TYPE_MAX; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t r = beg + slot; r < end; r += slots) {
        const int ty = idx[r];
        const float4 v = g[r * d4 + c];
#pragma unroll
        for (int t = 0; t < TYPE_MAX; ++t) {
            const float m = ty == t ? 1.f : 0.f;
            acc[t].x = fmaf(m, v.x, acc[t].x), acc[t].y = fmaf(m, v.y, acc[t].y);
            acc[t].z = fmaf(m, v.z, acc[t].z), acc[t].w = fmaf(m, v.w, acc[t].w);
        }
    }
#pragma unroll
    for (int t = 0; t < TYPE_MAX; ++t) {
        if (t >= n_types) break;
        red[threadIdx.x] = acc[t];
        __syncthreads();
        if (slot == 0) {
            float4 s = red[c];
            for (int k = 1; k < slots; ++k) {
                const float4 v = red[k * d4 + c];
                s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
            }
            partial[((int64_t)blockIdx.x * n_types + t) * d4 + c] = s;
        }
        __syncthreads();
    }
    if (!last_block(counter)) return;
    for (int e = threadIdx.x; e < n_types * d4; e += 256) {
        float4 s = partial[e];
        for (int b = 1; b < (int)gridDim.x; ++b) {
            const float4 v = partial[(int64_t)b * n_types * d4 + e];
            s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
        }
        out[e] = s;
    }
}

}  // namespace

// scratch (caller-owned, device): pamnet_reduce_scratch_bytes bytes, ZEROED ONCE before the first call (the kernels
// leave the counter at zero); one scratch per stream.  Layout: [0,8) counter | [256, ...) partials.
extern "C" int pamnet_reduce_scratch_bytes(int64_t* bytes) {
    if (!bytes) return PAMNET_ENULL;
    *bytes = 256 + (int64_t)TYPE_BLOCKS * TYPE_MAX * 64 * 16;      // >= NORM_BLOCKS doubles, >= type partials up to d = 256
    return PAMNET_OK;
}

// norm_out[0] = || g[0:n] ||_2  (fp64 across lanes / workgroups).  n % 4 == 0, g 16-byte aligned.
extern "C" int pamnet_grad_norm_f32(const float* g, int64_t n, void* scratch, float* norm_out, pamnet_stream_t stream) {
    if (n < 0 || (n & 3)) return PAMNET_EINVAL;
    if (!g || !scratch || !norm_out) return PAMNET_ENULL;
    const int64_t n4 = n / 4;
    int64_t blocks = ceil_div(n4, 2048);
    blocks = blocks < 1 ? 1 : (blocks > NORM_BLOCKS ? NORM_BLOCKS : blocks);
    hipLaunchKernelGGL(grad_norm_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), (const float4*)g, n4,
                       reinterpret_cast<double*>(static_cast<char*>(scratch) + 256), static_cast<unsigned*>(scratch),
                       norm_out);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

// loss[0] = mean_i |out[i] - y[i]|;  d_out[i] (nullable) = grad_scale * sign(out[i] - y[i]) / n   (sign(0) = 0 as torch)
extern "C" int pamnet_l1_loss_f32(const float* out, const float* y, int64_t n, float grad_scale, float* loss,
                                  float* d_out, pamnet_stream_t stream) {
    if (n < 1) return PAMNET_EINVAL;
    if (!out || !y || !loss) return PAMNET_ENULL;
    hipLaunchKernelGGL(l1_loss_kernel, dim3(1), dim3(256), 0, as_stream(stream), out, y, n, grad_scale, loss, d_out);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

// out[t, :] = sum_{r < n: idx[r] = t} g[r, :]  for t < n_types (<= 8), d in {16, 32, 64, 128, 256}; idx values outside
// [0, n_types) contribute nothing.
extern "C" int pamnet_type_rows_grad_f32(const float* g, const int32_t* idx, int64_t n, int64_t n_types, int64_t d,
                                         void* scratch, float* out, pamnet_stream_t stream) {
    const int64_t d4 = d / 4;
    if (n < 0 || n_types < 1 || n_types > TYPE_MAX || d < 4 || (d & 3) || d4 > 64 || (d4 & (d4 - 1))) return PAMNET_EINVAL;
    if (!out || !scratch || (n > 0 && (!g || !idx))) return PAMNET_ENULL;
    int64_t blocks = ceil_div(n, 32);
    blocks = blocks < 1 ? 1 : (blocks > TYPE_BLOCKS ? TYPE_BLOCKS : blocks);
    hipLaunchKernelGGL(type_rows_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), (const float4*)g, idx,
                       n, (int)n_types, (int)d4, reinterpret_cast<float4*>(static_cast<char*>(scratch) + 256),
                       static_cast<unsigned*>(scratch), (float4*)out);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}
