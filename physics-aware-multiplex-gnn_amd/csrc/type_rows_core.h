// Per-type row sums (gradient of `embeddings[x]`, models.py:107,140) as device bodies, shared by the stand-alone entry
// (reduce.hip) and the input-embedding launches (embed.hip):
//   out[t, :] = sum_{r: idx[r] = t} g[r, :],  t < n_types <= 8.
// Lane group (d4 lanes) per row slot; a workgroup owns a contiguous slice of rows, slot s walks rows s, s + slots, ...
// (fixed), slots meet in LDS in slot order, workgroups in finish_body in workgroup order -- deterministic.
#pragma once
#include "common.h"

namespace {
namespace type_rows {

constexpr int TYPE_MAX = 8, TYPE_BLOCKS = 64;

inline int blocks_for_rows(int64_t n) {
    int64_t blocks = (n + 127) / 128;                        // >= 128 rows per workgroup: few partials for the finish
    return (int)(blocks < 1 ? 1 : (blocks > TYPE_BLOCKS ? TYPE_BLOCKS : blocks));
}

// 256 threads; red: 256 float4 of LDS
__device__ __forceinline__ void grad_body(const float4* __restrict__ g, const int32_t* __restrict__ idx, int64_t n,
                                          int n_types, int d4, float4* __restrict__ partial, int bid, int nblk,
                                          float4* red) {
    const int c = threadIdx.x % d4, slot = threadIdx.x / d4, slots = 256 / d4;
    const int64_t per = (n + nblk - 1) / nblk;
    const int64_t beg = bid * per, end = beg + per < n ? beg + per : n;
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
    __syncthreads();
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
            partial[((int64_t)bid * n_types + t) * d4 + c] = s;
        }
        __syncthreads();
    }
}

// one workgroup (256 threads): out[e] = sum over workgroup partials, in workgroup order
__device__ __forceinline__ void finish_body(const float4* __restrict__ partial, int blocks, int cells,
                                            float4* __restrict__ out) {
    for (int e = threadIdx.x; e < cells; e += 256) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        int b = 0;
        for (; b + 8 <= blocks; b += 8) {                  // 8 independent loads in flight, added in workgroup order
            float4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = partial[(int64_t)(b + k) * cells + e];
#pragma unroll
            for (int k = 0; k < 8; ++k) s.x += v[k].x, s.y += v[k].y, s.z += v[k].z, s.w += v[k].w;
        }
        for (; b < blocks; ++b) {
            const float4 v = partial[(int64_t)b * cells + e];
            s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
        }
        out[e] = s;
    }
}

}  // namespace type_rows
}  // namespace
