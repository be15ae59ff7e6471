// Edge lengths and angles of graph construction, shared by the step-by-step kernels (graph.hip) and the molecule-local
// builder (graph_mol.hip) so that both produce the same floats.
#pragma once
#include "common.h"

__device__ __forceinline__ float dist3_xyz(float ax, float ay, float az, float bx, float by, float bz) {
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    // same association as (pos_i - pos_j).pow(2).sum(-1).sqrt() (models.py:65); no fma contraction
    const float s = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
    return __fsqrt_rn(s);
}

__device__ __forceinline__ float dist3(const float* __restrict__ pos, int64_t a, int64_t b) {
    return dist3_xyz(pos[3 * a + 0], pos[3 * a + 1], pos[3 * a + 2], pos[3 * b + 0], pos[3 * b + 1], pos[3 * b + 2]);
}

__device__ __forceinline__ float angle3(float ax, float ay, float az, float bx, float by, float bz) {
    // atan2(|a x b|, a.b)  (models.py:165-168); every product and sum rounded on its own (no contraction), so that the
    // value does not depend on the kernel the expression is inlined into
    const float dot = __fadd_rn(__fadd_rn(__fmul_rn(ax, bx), __fmul_rn(ay, by)), __fmul_rn(az, bz));
    const float cx = __fsub_rn(__fmul_rn(ay, bz), __fmul_rn(az, by));
    const float cy = __fsub_rn(__fmul_rn(az, bx), __fmul_rn(ax, bz));
    const float cz = __fsub_rn(__fmul_rn(ax, by), __fmul_rn(ay, bx));
    const float n2 = __fadd_rn(__fadd_rn(__fmul_rn(cx, cx), __fmul_rn(cy, cy)), __fmul_rn(cz, cz));
    return atan2f(__fsqrt_rn(n2), dot);
}
