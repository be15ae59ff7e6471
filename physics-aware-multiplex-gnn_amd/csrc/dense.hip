// Dense layers of ANY width: fp32-accurate GEMMs on the bf16 matrix pipe ("bf16x6", gemm_core.h).
//
// The engines (engine.hip, narrow_engine.hip) are built for hidden sizes up to 128 and the input widths of the default
// bases (16 / 18 / 42).  Everything else the reference accepts -- models.py:25 `dim` above 128, models.py:187-188 with
// another num_spherical * num_radial -- is a Sequential(Linear, SiLU) (layers/basic.py:19-22) or a bare F.linear of some
// [rows, k] x [m, k]^T shape.  One tiled kernel serves the three products of such a layer:
//     forward   Z = X W^T + b,  Y = SiLU(Z)            C[i][j] = sum_k a(i, k) b(j, k):  a = X,        b = W
//     backward  dX = dZ W                                                               a = dZ,       b = W^T
//               dW = dZ^T X,  db = column sums of dZ                                    a = dZ^T,     b = X^T
// with dZ = G * SiLU'(Z) formed while the A operand is staged (it is never stored).  Operands are addressed through two
// strides (outer index, summation index), so no transposed copy of anything is made.
//
// Tiling: a workgroup of 4 waves owns a 64 x 64 tile of C, a wave 32 x 32 of it (2 x 2 accumulators of
// v_mfma_f32_16x16x32_bf16); one block = 32 summation indices.  Per block every thread fetches ONE fragment slot of each
// operand (8 consecutive k of one outer index: two float4 when k is the contiguous index, eight coalesced dwords across
// the wave when the outer index is), splits it exactly into three bf16 pieces once (4.5 VALU per element) and stores the
// pieces as ready-made MFMA fragments: per operand, 16-index tile and piece a 1 KB lane-linear image, read back by the
// waves with conflict-free ds_read_b128.  The fetch of block t+1 is in flight during the 24 MFMAs of block t.
// The weight-gradient product sums over the ROWS (up to ~10^6): split over grid.z, partial tiles in caller-owned
// scratch, summed in a fixed order by a second launch -- deterministic, atomics-free, like wgrad.hip.
#include "common.h"
#include "gemm_core.h"

using namespace pamnet;

namespace {

constexpr int TM = 64, TN = 64, TK = 32;
constexpr int IMG_B = 1024;                     // one piece of one 16-index tile: 64 lane slots x 16 bytes
constexpr int OPND_B = 4 * 3 * IMG_B;           // an operand's 64 x 32 block: 4 tiles x 3 pieces
constexpr int DWG = 256;
constexpr int TLD = TN + 4;                       // row stride of the epilogue's transposed tile (floats)
static_assert(TM * TLD * 4 <= 2 * OPND_B, "the epilogue tile aliases the fragment images");
constexpr int MAX_SPLITS = 512;

struct Operand {
    const float* p;           // element (o, k) at p[o * so + k * sk]
    const float* z;           // nullable: the element is multiplied by SiLU'(z[o * so + k * sk])
    int64_t so, sk;           // one of them is 1
    int64_t no, nk;           // extents
    int vec;                  // k contiguous and every 8-group 16-byte aligned: two float4 per slot
};

struct Gemm {
    Operand a, b;
    float* C;                 // [no_a][ldc] (+ split * c_split)
    float* Zout;              // nullable: C before the activation (same layout)
    const float* bias;        // nullable, per column
    float* rowsum;            // nullable: sum_k a(i, k) per split: rowsum[split * no_a + i]
    int64_t ldc, c_split;
    int64_t k_chunk;          // summation indices per split (multiple of TK)
    int act;
};

// the thread's fragment slot inside a 64 x 32 block: outer index o in [0, 64), k-group kg in [0, 4)
__device__ __forceinline__ void slot_of(const Operand& op, int t, int& o, int& kg) {
    if (op.sk == 1) o = t >> 2, kg = t & 3;     // 4 neighbouring threads: 128 contiguous bytes of one row
    else o = t & 63, kg = t >> 6;               // 64 neighbouring threads: 256 contiguous bytes of one k
}

__device__ __forceinline__ void fetch8(const float* __restrict__ p, const Operand& op, int64_t o, int64_t k0, int64_t kend,
                                       float (&v)[8]) {
    const bool ook = o < op.no;
    if (op.sk == 1) {
        const float* q = p + (ook ? o : 0) * op.so;
        if (op.vec && k0 + 8 <= kend) {
            const float4 u0 = *reinterpret_cast<const float4*>(q + k0), u1 = *reinterpret_cast<const float4*>(q + k0 + 4);
            v[0] = u0.x, v[1] = u0.y, v[2] = u0.z, v[3] = u0.w, v[4] = u1.x, v[5] = u1.y, v[6] = u1.z, v[7] = u1.w;
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = q[k0 + u < kend ? k0 + u : 0];  // clamped address, masked below
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = k0 + u < kend ? v[u] : 0.f;
        }
        if (!ook) {
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = 0.f;
        }
    } else {
        const float* q = p + (ook ? o : 0) * op.so;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool ok = k0 + u < kend;
            v[u] = q[(ok ? k0 + u : 0) * op.sk];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (ook && k0 + u < kend) ? v[u] : 0.f;
    }
}

__device__ __forceinline__ void fetch_operand(const Operand& op, int64_t o, int64_t k0, int64_t kend, float (&v)[8]) {
    fetch8(op.p, op, o, k0, kend, v);
    if (op.z) {                                                // dZ = G * SiLU'(Z); masked elements are 0 * SiLU'(0)
        float zv[8];
        fetch8(op.z, op, o, k0, kend, zv);
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] *= dsilu(zv[u]);
    }
}

__device__ __forceinline__ void store_slot(char* base, int o, int kg, const float (&v)[8]) {
    const Frag3 f = split_frag(v);
    char* d = base + (o >> 4) * 3 * IMG_B + ((o & 15) + 16 * kg) * 16;
#pragma unroll
    for (int pc = 0; pc < 3; ++pc)
        *reinterpret_cast<uint4*>(d + pc * IMG_B) = make_uint4(f.p[pc][0], f.p[pc][1], f.p[pc][2], f.p[pc][3]);
}

__global__ __launch_bounds__(DWG, 4) void dense_gemm_kernel(Gemm g) {
    __shared__ __attribute__((aligned(16))) char lds[2 * OPND_B + 4 * 64 * 4];
    char* la = lds;
    char* lb = lds + OPND_B;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wi = w >> 1, wj = w & 1;
    const int64_t i0 = (int64_t)blockIdx.x * TM, j0 = (int64_t)blockIdx.y * TN;
    const int split = blockIdx.z;
    const int64_t kbeg = (int64_t)split * g.k_chunk;
    const int64_t kend = kbeg + g.k_chunk < g.a.nk ? kbeg + g.k_chunk : g.a.nk;

    int oa, kga, ob, kgb;
    slot_of(g.a, t, oa, kga);
    slot_of(g.b, t, ob, kgb);
    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float rs = 0.f;
    float va[8], vb[8];
    auto sum8 = [](const float (&v)[8]) { return ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])); };

    if (kbeg < kend) {
        fetch_operand(g.a, i0 + oa, kbeg + 8 * kga, kend, va);
        fetch_operand(g.b, j0 + ob, kbeg + 8 * kgb, kend, vb);
        if (g.rowsum) rs += sum8(va);
        store_slot(la, oa, kga, va);
        store_slot(lb, ob, kgb, vb);
    }
    __syncthreads();
    for (int64_t k0 = kbeg; k0 < kend; k0 += TK) {
        uint4 fa[2][3], fb[2][3];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc)
                fa[a][pc] = *reinterpret_cast<const uint4*>(la + ((wi * 2 + a) * 3 + pc) * IMG_B + lane * 16);
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc)
                fb[b][pc] = *reinterpret_cast<const uint4*>(lb + ((wj * 2 + b) * 3 + pc) * IMG_B + lane * 16);
        __syncthreads();                                       // every wave holds its fragments: the images are free
        const bool more = k0 + TK < kend;
        if (more) {
            fetch_operand(g.a, i0 + oa, k0 + TK + 8 * kga, kend, va);
            fetch_operand(g.b, j0 + ob, k0 + TK + 8 * kgb, kend, vb);
        }
        // six piece products, smallest first; consecutive MFMAs go to different accumulators
#pragma unroll
        for (int term = 0; term < 6; ++term) {
            const int pa = term == 0 ? 2 : (term == 1 || term == 3) ? 1 : 0;
            const int pb = term == 2 ? 2 : (term == 1 || term == 4) ? 1 : 0;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const u32x4 av = {fa[a][pa].x, fa[a][pa].y, fa[a][pa].z, fa[a][pa].w};
                    const u32x4 bv = {fb[b][pb].x, fb[b][pb].y, fb[b][pb].z, fb[b][pb].w};
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av),
                                                                        __builtin_bit_cast(bf16x8, bv), acc[a][b], 0, 0, 0);
                }
        }
        if (more) {
            if (g.rowsum) rs += sum8(va);
            store_slot(la, oa, kga, va);
            store_slot(lb, ob, kgb, vb);
        }
        __syncthreads();
    }

    // ---- epilogue: accumulator element r of a lane is row 4 (lane / 16) + r, column lane % 16 of its 16 x 16 tile -- stored
    // from there a wave-instruction would write sixteen 64-byte pieces.  The tile goes through LDS (the fragment images are
    // free now) and leaves as rows: a wave writes 256 contiguous bytes per instruction.
    const int r16 = lane & 15, kg = lane >> 4;
    float* T = reinterpret_cast<float*>(lds);                  // [64][TLD] floats = 17 KB of the 24 KB of images
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) T[(wi * 32 + a * 16 + 4 * kg + r) * TLD + wj * 32 + b * 16 + r16] = acc[a][b][r];
    __syncthreads();
    {
        float* C = g.C + (int64_t)split * g.c_split;
        const int64_t j = j0 + lane;
        const bool jok = j < g.b.no;
        const float bv = (g.bias && jok) ? g.bias[j] : 0.f;
#pragma unroll 4
        for (int ii = 0; ii < TM / 4; ++ii) {
            const int row = w + 4 * ii;
            const int64_t i = i0 + row;
            if (!jok || i >= g.a.no) continue;
            const float z = T[row * TLD + lane] + bv;
            if (g.Zout) g.Zout[i * g.ldc + j] = z;
            C[i * g.ldc + j] = g.act ? silu(z) : z;
        }
    }
    __syncthreads();
    if (g.rowsum && blockIdx.y == 0) {                         // bias gradient: the A operand's sums over this split's k
        float* red = reinterpret_cast<float*>(lds + 2 * OPND_B);        // [4 k-groups][64 outer indices]
        red[kga * 64 + oa] = rs;
        __syncthreads();
        if (t < 64 && i0 + t < g.a.no)
            g.rowsum[(int64_t)split * g.a.no + i0 + t] = (red[t] + red[64 + t]) + (red[128 + t] + red[192 + t]);
    }
}

// out[e] = sum over splits of partial[s * stride + e] in a fixed order: a workgroup owns 64 consecutive elements, its four
// waves take the splits s = w, w + 4, ... (four loads in flight per thread: a thin weight gradient has few elements and
// up to 512 splits, so the walk over the splits is the whole cost), the four wave sums are added in wave order.
__global__ __launch_bounds__(256) void dense_reduce_kernel(const float* __restrict__ partial, int splits, int64_t stride,
                                                           int64_t count, float* __restrict__ out) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t e = (int64_t)blockIdx.x * 64 + lane;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < count) {
        const float* p = partial + e;
        int s = w;
        for (; s + 12 < splits; s += 16) {
            s0 += p[(int64_t)s * stride];
            s1 += p[(int64_t)(s + 4) * stride];
            s2 += p[(int64_t)(s + 8) * stride];
            s3 += p[(int64_t)(s + 12) * stride];
        }
        for (; s < splits; s += 4) s0 += p[(int64_t)s * stride];
    }
    red[w][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (w == 0 && e < count) out[e] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

inline Operand operand(const float* p, const float* z, int64_t so, int64_t sk, int64_t no, int64_t nk) {
    Operand op{p, z, so, sk, no, nk, 0};
    const bool al = (reinterpret_cast<uintptr_t>(p) & 15) == 0 && (!z || (reinterpret_cast<uintptr_t>(z) & 15) == 0);
    op.vec = (sk == 1 && (so & 3) == 0 && al) ? 1 : 0;
    return op;
}

inline int splits_for(int64_t n, int64_t k, int64_t m) {
    const int64_t tiles = ceil_div(m, TM) * ceil_div(k, TN);
    int64_t s = ceil_div(1024, tiles);                        // ~4 workgroups per CU over the whole launch
    const int64_t by_rows = ceil_div(n, 4 * TK);              // at least 128 rows per split
    if (s > by_rows) s = by_rows;
    if (s > MAX_SPLITS) s = MAX_SPLITS;
    return (int)(s < 1 ? 1 : s);
}

inline int launch(const Gemm& g, int splits, hipStream_t st) {
    const dim3 grid((unsigned)ceil_div(g.a.no, TM), (unsigned)ceil_div(g.b.no, TN), (unsigned)splits);
    if (grid.y > 65535u) return PAMNET_EINVAL;
    hipLaunchKernelGGL(dense_gemm_kernel, grid, dim3(DWG), 0, st, g);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

}  // namespace

extern "C" int pamnet_dense_fwd_f32(const float* X, int64_t ldx, const float* W, int64_t ldw, const float* bias, int64_t n,
                                    int64_t k, int64_t m, int32_t act, float* Z, float* Y, pamnet_stream_t stream) {
    if (n < 0 || k < 1 || m < 1 || ldx < k || ldw < k) return PAMNET_EINVAL;
    if (n == 0) return PAMNET_OK;
    if (!X || !W || !Y) return PAMNET_ENULL;
    Gemm g{};
    g.a = operand(X, nullptr, ldx, 1, n, k);
    g.b = operand(W, nullptr, ldw, 1, m, k);
    g.C = Y, g.Zout = Z, g.bias = bias, g.rowsum = nullptr, g.ldc = m, g.c_split = 0;
    g.k_chunk = ceil_div(k, TK) * TK, g.act = act ? 1 : 0;
    return launch(g, 1, as_stream(stream));
}

extern "C" int pamnet_dense_scratch_floats(int64_t n, int64_t k, int64_t m, int64_t* floats) {
    if (n < 0 || k < 1 || m < 1 || !floats) return PAMNET_EINVAL;
    *floats = (int64_t)splits_for(n, k, m) * (m * k + m);
    return PAMNET_OK;
}

// dZ = act ? G * SiLU'(Z) : G.   dX [n][k] (nullable) = dZ W;   dW [m][k] (nullable) = dZ^T X, db [m] (nullable, with dW).
// partial: pamnet_dense_scratch_floats(n, k, m) floats when dW is asked for.
extern "C" int pamnet_dense_bwd_f32(const float* G, const float* Z, const float* X, int64_t ldx, const float* W, int64_t ldw,
                                    int64_t n, int64_t k, int64_t m, int32_t act, float* dX, float* dW, float* db,
                                    float* partial, pamnet_stream_t stream) {
    if (n < 0 || k < 1 || m < 1 || ldx < k || ldw < k) return PAMNET_EINVAL;
    if (!G || (act && !Z)) return PAMNET_ENULL;
    if ((dX && !W) || (dW && (!X || !partial)) || (db && !dW)) return PAMNET_ENULL;
    hipStream_t st = as_stream(stream);
    const float* z = act ? Z : nullptr;
    if (dX && n > 0) {
        Gemm g{};
        g.a = operand(G, z, m, 1, n, m);
        g.b = operand(W, nullptr, 1, ldw, k, m);                // b(j, kk) = W[kk][j]
        g.C = dX, g.ldc = k, g.k_chunk = ceil_div(m, TK) * TK;
        const int rc = launch(g, 1, st);
        if (rc) return rc;
    }
    if (dW) {
        const int splits = splits_for(n, k, m);
        Gemm g{};
        g.a = operand(G, z, 1, m, m, n);                        // a(i, kk) = dZ[kk][i]
        g.b = operand(X, nullptr, 1, ldx, k, n);                // b(j, kk) = X[kk][j]
        g.C = partial, g.ldc = k, g.c_split = m * k;
        g.rowsum = db ? partial + (int64_t)splits * m * k : nullptr;
        g.k_chunk = ceil_div(ceil_div(n > 0 ? n : 1, splits), TK) * TK;
        const int rc = launch(g, splits, st);
        if (rc) return rc;
        hipLaunchKernelGGL(dense_reduce_kernel, dim3((unsigned)ceil_div(m * k, 64)), dim3(256), 0, st, partial, splits, m * k,
                           m * k, dW);
        PAMNET_LAUNCH_CHECK();
        if (db) {
            hipLaunchKernelGGL(dense_reduce_kernel, dim3((unsigned)ceil_div(m, 64)), dim3(256), 0, st,
                               partial + (int64_t)splits * m * k, splits, m, m, db);
            PAMNET_LAUNCH_CHECK();
        }
    }
    return PAMNET_OK;
}
