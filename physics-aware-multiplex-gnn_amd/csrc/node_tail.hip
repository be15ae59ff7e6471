// Fused node-update tail of PAMNet's message-passing layers (dim = 128), forward and backward.
//
// Both layer kinds end with the same 10-Linear stack per node (layers/global_message_passing.py:39-50 /
// layers/local_message_passing.py:55-66; Res: layers/basic.py:25-33):
//     h0 = SiLU(L0 x)                                 mlp_x2
//     r1 = SiLU(L2 SiLU(L1 h0)) + h0 + res_x          res1 (+ the layer's residual input)
//     r2 = SiLU(L4 SiLU(L3 r1)) + r1                  res2
//     r3 = SiLU(L6 SiLU(L5 r2)) + r2                  res3  -> x_out
//     o3 = SiLU(L9 SiLU(L8 SiLU(L7 r3)))              mlp_out
//     out = W_out . o3 + b_out,  att = W . o3
// One workgroup keeps a 16-row tile on-chip from the first GEMM to the last: ten fp32-MFMA GEMMs, bias / SiLU /
// residual epilogues and both heads in ONE launch (the reference issues ~40 kernels for the same work).
//
// The chain is latency-bound (a 16x128x128 GEMM is 256 MFMAs in total), so each layer is the shortest dependent
// sequence:  A fragments (ds_read_b128)  ->  MFMAs  ->  epilogue IN THE ACCUMULATOR REGISTERS (bias, SiLU, residuals
// read from LDS in fragment layout, z_k / taps stored to global)  ->  one LDS write  ->  ONE barrier.
// What bounds it: a 16-row tile re-reads the whole 64 KB weight matrix for every layer (8 FLOP per L2 byte), so the
// chain runs at L2 / texture-path rate, not at MFMA rate (measured ~29 us for 10 layers on 143 workgroups; the MFMAs
// alone need ~9 us).  The next layer's weight slice is therefore requested right after the MFMAs are issued, and the
// backward -- whose transposed weight reads are scalar, 4x more load instructions -- uses 8 waves x 16 columns so two
// waves per SIMD overlap loads, MFMAs and epilogue (27 us vs 29 us); its z_k are fetched before the MFMAs start.
#include "common.h"
#include "edge_core.h"
#include "gemm_core.h"
#include "wgrad_core.h"

using namespace pamnet;

#ifdef PAMNET_PHASE_PROBE
// Development aid (tools/tail_probe.py): shader-clock timestamps of the middle workgroup, 4 per layer.
__device__ long long pamnet_tail_probe[64];
#define TPROBE(i)                                                                                      \
    do {                                                                                               \
        if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) pamnet_tail_probe[i] = clock64();          \
    } while (0)
extern "C" int pamnet_tail_probe_read(long long* host64) {
    return (int)hipMemcpyFromSymbol(host64, HIP_SYMBOL(pamnet_tail_probe), sizeof(long long) * 64);
}
#else
#define TPROBE(i)
#endif

namespace {

struct TailParams {
    const float* W[10];       // row-major [128][128] matrices, or fragment-ordered images when `packed`
    const float* b[10];
    int packed;
    const float* w_out;   // [128]
    const float* b_out;   // [1]
    const float* w_att;   // [128]
};

// Head of the NEXT layer applied to this chain's x_out while the tile is still in LDS (nblk = 0: none):
//   x1 = SiLU(Wx1 x_out + bx1),  P_b = x1 * wp[b]^T  (the node-level halves of the next layer's split message MLPs).
struct PreNext {
    const float* Wx1;
    const float* bx1;
    const float* wp[4];
    int ldwp, nblk;
    float *Zx1, *x1, *P;      // [n][128], [n][128], [nblk][n][128]
};

// Rider of a forward chain launch: row tiles [tile0, tile0 + ntiles) of one triplet/pair MLP (edge::mlp2_fwd_body) dealt to
// the workgroups behind the chain's ceil(n/16) -- the MLP does not depend on the node features, and the chain leaves
// 256 - ceil(n/16) CUs idle.
struct Mlp2Rider {
    const float* x;
    int64_t m;
    edge::Mlp2Set set;
    int tile0, ntiles, n_chain;
};

constexpr int BMN = 16;                       // rows per workgroup
constexpr int SLOT = BMN * LDT;               // floats per LDS slot
constexpr int TWG = 512;                      // 8 waves
// more row tiles than CUs: a CU gets several workgroups in turn -- the lean chain forms (co-resident workgroups) from here
constexpr unsigned LEAN_FROM_TILES = 256;
// PAMNET_CHAIN_LEAN=0: never, 1: forward only (default: both directions) -- read once, for A/B timing
inline int lean_mode() {
    static const int v = [] { const char* e = getenv("PAMNET_CHAIN_LEAN"); return e ? atoi(e) : 2; }();
    return v;
}

// fragment coordinates of accumulator element r of this lane: row = 4*(lane>>4) + r, col = 16*wave + (lane&15)
struct Frag {
    int wc, r16, kg;
    __device__ __forceinline__ Frag() {
        const int lane = threadIdx.x & 63;
        wc = (threadIdx.x >> 6) * 16;
        r16 = lane & 15;
        kg = lane >> 4;
    }
    __device__ __forceinline__ int row(int r) const { return 4 * kg + r; }
    __device__ __forceinline__ int col() const { return wc + r16; }
};

// A pointer the caller guarantees: `if (Wnext) <request the next layer's weights>` inside a layer body that is inlined with a
// run-time pointer keeps its branch, and behind the join the compiler must assume the requests were NOT made -- the wait for the
// layer's bias (requested before them; vmcnt retires in order) becomes s_waitcnt vmcnt(0): the epilogue waited for the whole next
// weight slice, ~1 000 cycles of every forward layer (seen in the ISA, round 6).  With the pointer known non-null the requests are
// unconditional and the bias wait counts them (vmcnt(12 / 16 / 24)).
__device__ __forceinline__ const float* nn(const float* p) {
    __builtin_assume(p != nullptr);
    return p;
}

// ---- fragment-ordered weight images (pamnet_pack_weights_f32) -----------------------------------------------------------
// Requesting a weight slice from the row-major matrix costs ~1 150 cycles of every ~5 000-cycle layer: each of the 16
// 1 KB requests touches 16 half-used cache lines.  An image stores, for 16-column tile j and k-group q, the 64 lanes'
// float4 fragments back to back: img4[(j*8 + q)*64 + lane] -- every request is one contiguous 1 KB read.
//   forward orientation  (Y = X W^T): lane's float4 = W[16j + (lane&15)][16q + 4(lane>>4) + 0..3]
//   transposed orientation (Y = X W): lane's float4 = W[16q + 4(lane>>4) + 0..3][16j + (lane&15)]
// one 128 x 16 weight slice of a wave: 8 x float4 = 32 VGPRs
struct WFrag1 {
    float4 b[DIM / 16];
};

template <bool TRANS>
__device__ __forceinline__ void load_wfrag1(WFrag1& f, const float* __restrict__ W, int wc) {
    const int lane = threadIdx.x & 63;
    const int r16 = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int q = 0; q < DIM / 16; ++q) {
        if (!TRANS) {
            f.b[q] = *reinterpret_cast<const float4*>(W + (size_t)(wc + r16) * DIM + 4 * kg + 16 * q);
        } else {
            const float* wp = W + (size_t)(16 * q + 4 * kg) * DIM + wc + r16;
            f.b[q] = make_float4(wp[0], wp[DIM], wp[2 * DIM], wp[3 * DIM]);
        }
    }
}

__device__ __forceinline__ void load_wfrag1_img(WFrag1& f, const float* __restrict__ img) {     // 8 waves x 16 columns
    const float4* p4 = reinterpret_cast<const float4*>(img) + (threadIdx.x & 63);
    const int j = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < DIM / 16; ++q) f.b[q] = p4[(j * 8 + q) * 64];
}

// [16 x 128] (LDS) x [128 x 16] (registers) -> one 16x16 accumulator; two interleaved chains
__device__ __forceinline__ f32x4 mma_strip(const float* __restrict__ As, const WFrag1& f) {
    const int lane = threadIdx.x & 63;
    const float* ap = As + (lane & 15) * LDT + 4 * (lane >> 4);
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < DIM / 16; ++q) {
        const float4 a = *reinterpret_cast<const float4*>(ap + 16 * q);
        const float4 b = f.b[q];
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, c1, 0, 0, 0);
    }
    return c0 + c1;
}

// mma_strip with the MFMA operands swapped (gemm_core.h mma_tile_frag_t): lane l holds out[row l & 15][channels wc + 4 (l >> 4) +
// 0..3] -- a float4 of one row -- and every epilogue access to an LDS tile is 16 bytes; same sums of the same products in the same
// order (two interleaved chains, added at the end), bitwise mma_strip's values.
__device__ __forceinline__ f32x4 mma_strip_t(const float* __restrict__ As, const WFrag1& f) {
    const int lane = threadIdx.x & 63;
    const float* ap = As + (lane & 15) * LDT + 4 * (lane >> 4);
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < DIM / 16; ++q) {
        const float4 a = *reinterpret_cast<const float4*>(ap + 16 * q);
        const float4 b = f.b[q];
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b.x, a.x, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b.y, a.y, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b.z, a.z, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b.w, a.w, c1, 0, 0, 0);
    }
    return c0 + c1;
}
__device__ __forceinline__ f32x4 lds_f32x4(const float* p) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    return f32x4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ void st_f32x4(float* p, const f32x4& v) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }

// Forward: 4 waves x 32 columns (float4 weight loads are cheap per byte, so fewer, wider waves win; an 8-wave x 16-column
// version measured 35 us against 31 us: two waves per SIMD only take turns on the matrix pipe and then idle at the
// barrier -- tools/tail_probe.py).
// HEADS = false: the head branch (mlp_out + the two dot products, layers 7-9) is left to node_heads_fwd_kernel, which
// runs it for every layer of the model in one launch after the layer loop: nothing downstream of a layer depends on
// its heads, the chain gets three dependent GEMMs shorter, and the batched launch has 2L x ceil(n/16) workgroups
// instead of the chain's ceil(n/16) (143 of 256 CUs at the QM9 batch).
// Round 6: the chain input x2 of the LOCAL layer may be FORMED by this launch's own row tiles instead of read -- the two chained
// aggregations of layers/local_message_passing.py:49-54 that pamnet_local_agg_fwd_f32 computes in a launch of its own ahead of the
// chain:   m_t[e] = m_ji[e] + sum_{r in rows(e)} m_nb[t_col[r]] * s[r],   x2[i] = x1[i] + sum_{e -> i} q3[e] * m_t[e]
// (the same operations in the same order: bit for bit that kernel's rows).  x2 and m_t are still written (saved activations).
struct LocalAgg {
    const float4 *m_ji, *m_nb, *s, *q3, *init;
    const int32_t *t_ptr, *t_col, *l_ptr;
    float4* m_t;                  // nullable: backward-only save
    float4* x2_out;
};
// x2 rows of NJ nodes (this thread's float4 column) at once.  Node by node, edge by edge, four rows at a time the sums of one row
// were a chain of ~10 dependent round trips (edge range -> row ranges -> columns -> rows, per edge), twice per thread, ahead of the
// chain's first layer.  Here the edge ranges of all nodes are one request, every step takes NK edges of every node together (their
// row ranges, m_ji and gate rows in one request), and their rows NU at a time with the NEXT columns in flight beside the rows.
// The additions are those of local_agg_fwd_kernel in the same order (edge_agg.hip): rows in triplet / pair order onto m_ji, edges in
// CSR order onto init -- bitwise the stand-alone kernel's x2 and m_t (tests: PAMNET_FUSE_LOCAL_AGG=0 / 1 hash-equal).
template <int NJ>
__device__ __forceinline__ void local_agg_rows(const LocalAgg& la, const int64_t (&node)[NJ], const bool (&ok)[NJ], int c,
                                               float4 (&out)[NJ]) {
    constexpr int NK = 2, NU = 2;
    int e0[NJ], e1[NJ];
    float4 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        e0[j] = e1[j] = 0;
        acc[j] = f4zero();
        if (ok[j]) {
            e0[j] = la.l_ptr[node[j]], e1[j] = la.l_ptr[node[j] + 1];
            if (la.init) acc[j] = la.init[node[j] * 32 + c];
        }
    }
    auto any_edges = [&]() __attribute__((always_inline)) {
        bool m = false;
#pragma unroll
        for (int j = 0; j < NJ; ++j) m = m || e0[j] < e1[j];
        return m;
    };
    while (any_edges()) {
        int t0[NJ][NK], t1[NJ][NK];
        float4 v[NJ][NK], gate[NJ][NK];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const int e = e0[j] + k;
                t0[j][k] = t1[j][k] = 0;
                if (e < e1[j]) {
                    t0[j][k] = la.t_ptr[e], t1[j][k] = la.t_ptr[e + 1];
                    v[j][k] = la.m_ji[(int64_t)e * 32 + c];
                    gate[j][k] = la.q3[(int64_t)e * 32 + c];
                }
            }
        int col[NJ][NK][NU];
        auto fetch_cols = [&](int (&o)[NJ][NK][NU]) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int k = 0; k < NK; ++k)
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        o[j][k][u] = -1;
                        if (t0[j][k] + u < t1[j][k]) o[j][k][u] = la.t_col[t0[j][k] + u];
                    }
        };
        auto any_rows = [&]() __attribute__((always_inline)) {
            bool m = false;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int k = 0; k < NK; ++k) m = m || t0[j][k] < t1[j][k];
            return m;
        };
        fetch_cols(col);
        while (any_rows()) {
            float4 ra[NJ][NK][NU], rb[NJ][NK][NU];
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int k = 0; k < NK; ++k)
#pragma unroll
                    for (int u = 0; u < NU; ++u)
                        if (col[j][k][u] >= 0) {
                            ra[j][k][u] = la.m_nb[(int64_t)col[j][k][u] * 32 + c];
                            rb[j][k][u] = la.s[(int64_t)(t0[j][k] + u) * 32 + c];
                        }
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int k = 0; k < NK; ++k) t0[j][k] += NU;
            int ncol[NJ][NK][NU];
            fetch_cols(ncol);                                 // (in flight beside the rows)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int k = 0; k < NK; ++k)
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        if (col[j][k][u] >= 0) v[j][k] = f4add(v[j][k], f4mul(ra[j][k][u], rb[j][k][u]));
                        col[j][k][u] = ncol[j][k][u];
                    }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const int e = e0[j] + k;
                if (e < e1[j]) {
                    if (la.m_t) st_nt4(la.m_t + (int64_t)e * 32 + c, v[j][k]);      // (backward-only save: streamed)
                    acc[j] = f4add(acc[j], f4mul(v[j][k], gate[j][k]));
                }
            }
            e0[j] += NK;
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        if (ok[j]) la.x2_out[node[j] * 32 + c] = acc[j];
        out[j] = acc[j];
    }
}

template <bool PACKED, bool HEADS, bool RIDER = false>
__global__ __launch_bounds__(WG) void node_tail_fwd_kernel(const float* __restrict__ x2,
                                                           const float* __restrict__ res_x, int64_t n, TailParams p,
                                                           float* __restrict__ Z, float* __restrict__ R,
                                                           float* __restrict__ x_out, float* __restrict__ out,
                                                           float* __restrict__ att, PreNext nx,
                                                           Mlp2Rider rd = Mlp2Rider{}, LocalAgg la = LocalAgg{}) {
    // 5 working slots + 10 pre-activation tiles + 3 residual taps: everything the backward needs is parked in LDS and
    // written out once, as coalesced 512-byte rows, after the chain -- no global store (and no wait for its
    // acknowledgement, vmcnt retires in order) sits between one layer's MFMAs and the next layer's weight slice.
    __shared__ __attribute__((aligned(16))) float lds[18 * SLOT];
    if constexpr (RIDER) {
        if ((int)blockIdx.x >= rd.n_chain) {
            constexpr int RMT = 3;                            // 3-tile chunks: the 4-wave geometry without spills
            static_assert(18 * SLOT * 4 >= RMT * edge::MLP2_TILE_B, "rider tiles must fit the chain's LDS");
            const int b = (int)blockIdx.x - rd.n_chain, nr = (int)gridDim.x - rd.n_chain;
            const int base = rd.ntiles / nr, rem = rd.ntiles % nr;
            const int64_t t0 = rd.tile0 + (int64_t)b * base + (b < rem ? b : rem);
            const int cnt = base + (b < rem ? 1 : 0);
            edge::Span sp;
            sp.beg = t0 * 16;
            const int64_t e = (t0 + cnt) * 16;
            sp.end = e < rd.m ? e : rd.m;
            if (sp.beg > sp.end) sp.beg = sp.end;
            const int nch = (cnt + RMT - 1) / RMT;
            sp.cmt = nch > 0 ? (cnt + nch - 1) / nch : 1;
            edge::mlp2_fwd_body<RMT, 4>(rd.x, rd.set, sp, lds);
            return;
        }
    }
    float* X0 = lds;
    float* RX = lds + SLOT;
    float* A = lds + 2 * SLOT;
    float* B = lds + 3 * SLOT;
    float* C = lds + 4 * SLOT;
    float* ZL = lds + 5 * SLOT;               // [10] z_k tiles
    float* TL = lds + 15 * SLOT;              // [3]  r1, r2, x_out tiles
    const int64_t row0 = (int64_t)blockIdx.x * BMN;
    const int64_t plane = n * DIM;
    const int lane = threadIdx.x & 63, r16 = lane & 15, kg = lane >> 4;
    const int wc = (threadIdx.x >> 6) * 32;

    TPROBE(40);
    WFrag wf;
    load_w<PACKED>(wf, p.W[0], DIM, wc);
    if (la.m_ji) {                                            // (workgroup-uniform) the x2 rows are formed here
        constexpr int NJ = BMN / 8;                           // rows per thread of the sweep
        const int c4 = threadIdx.x & 31, rq = threadIdx.x >> 5;
        int64_t node[NJ];
        bool ok[NJ];
        float4 xin[NJ], rx[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) node[j] = row0 + rq + 8 * j, ok[j] = node[j] < n;
#pragma unroll
        for (int j = 0; j < NJ; ++j) rx[j] = ldg4z(res_x, node[j], n, DIM, c4);
        local_agg_rows<NJ>(la, node, ok, c4, xin);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            st_lds4(RX, rq + 8 * j, c4, rx[j]);
            st_lds4(X0, rq + 8 * j, c4, xin[j]);
        }
    } else {
        sweep_rows<BMN>([&](int r, int c4) {
            const int64_t g = row0 + r;
            st_lds4(RX, r, c4, ldg4z(res_x, g, n, DIM, c4));
            st_lds4(X0, r, c4, ldg4z(x2, g, n, DIM, c4));
        });
    }
    __syncthreads();

    // layer k: in -> dst = SiLU(W_k in + b_k) (+ add1 + add2); z_k parked; optional tap of the result
    auto layer = [&](const float* in, float* dst, int k, const float* add1, const float* add2, float* tap,
                     const float* Wnext) {
        TPROBE(4 * k);
        const Bias8 bv = load_bias8(p.b[k], wc);               // before the prefetch (in-order vmcnt)
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        mma_tile_frag_t(in, wf, acc);                          // lane: row r16, channels wc + 16 n2 + 4 kg + 0..3
        TPROBE(4 * k + 1);
        if (Wnext) load_w<PACKED>(wf, Wnext, DIM, wc);
        float* zk = ZL + k * SLOT;
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2) {
            const int o = r16 * LDT + wc + 16 * n2 + 4 * kg;   // (16-byte aligned: LDT * 4 = 33 x 16 bytes)
            const float4 z = make_float4(acc[n2][0] + bv.v[n2].x, acc[n2][1] + bv.v[n2].y, acc[n2][2] + bv.v[n2].z,
                                         acc[n2][3] + bv.v[n2].w);
            float4 a = make_float4(silu(z.x), silu(z.y), silu(z.z), silu(z.w));
            if (add1) {
                const float4 t = *reinterpret_cast<const float4*>(add1 + o);
                a.x += t.x, a.y += t.y, a.z += t.z, a.w += t.w;
            }
            if (add2) {
                const float4 t = *reinterpret_cast<const float4*>(add2 + o);
                a.x += t.x, a.y += t.y, a.z += t.z, a.w += t.w;
            }
            *reinterpret_cast<float4*>(dst + o) = a;
            *reinterpret_cast<float4*>(zk + o) = z;
            if (tap) *reinterpret_cast<float4*>(tap + o) = a;
        }
        TPROBE(4 * k + 2);
        __syncthreads();
        TPROBE(4 * k + 3);
    };

    layer(X0, A, 0, nullptr, nullptr, nullptr, nn(p.W[1]));        // h0 -> A
    layer(A, B, 1, nullptr, nullptr, nullptr, nn(p.W[2]));         // a1 -> B
    layer(B, C, 2, A, RX, TL, nn(p.W[3]));                         // r1 -> C   (+ h0 + res_x)
    layer(C, A, 3, nullptr, nullptr, nullptr, nn(p.W[4]));         // a3 -> A
    layer(A, B, 4, C, nullptr, TL + SLOT, nn(p.W[5]));             // r2 -> B   (+ r1)
    layer(B, A, 5, nullptr, nullptr, nullptr, nn(p.W[6]));         // a5 -> A
    constexpr int NZ = HEADS ? 10 : 7;                         // z_k tiles this kernel produces
    layer(A, C, 6, B, nullptr, TL + 2 * SLOT, nn(HEADS ? p.W[7] : (nx.nblk > 0 ? nx.Wx1 : p.W[6])));   // r3 -> C = x_out  (no next head: any valid image)
    if constexpr (HEADS) {
        layer(C, A, 7, nullptr, nullptr, nullptr, nn(p.W[8]));         // o1 -> A
        layer(A, B, 8, nullptr, nullptr, nullptr, nn(p.W[9]));         // o2 -> B
        layer(B, A, 9, nullptr, nullptr, nullptr, nn(nx.nblk > 0 ? nx.Wx1 : p.W[9]));   // o3 -> A
    }

    // park -> memory: Z[10][n][128], R[2][n][128], x_out[n][128]
    sweep_rows<BMN>([&](int r, int c4) {
        const int64_t g = row0 + r;
        if (g >= n) return;
        if (Z) {                                              // backward-only saves: null in inference mode
#pragma unroll
            for (int k = 0; k < NZ; ++k) stg4(Z + (int64_t)k * plane, g, DIM, c4, lds4(ZL + k * SLOT, r, c4));
            stg4(R, g, DIM, c4, lds4(TL, r, c4));
            stg4(R + plane, g, DIM, c4, lds4(TL + SLOT, r, c4));
        }
        stg4(x_out, g, DIM, c4, lds4(TL + 2 * SLOT, r, c4));
    });
    TPROBE(41);

    // heads: 16 lanes per row, 8 columns each, butterfly over the 16-lane group
    if constexpr (HEADS) {
        const int r = threadIdx.x >> 4, part = threadIdx.x & 15;
        float so = 0.f, sa = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float v = A[r * LDT + part * 8 + c];
            so += v * p.w_out[part * 8 + c];
            sa += v * p.w_att[part * 8 + c];
        }
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) {
            so += __shfl_xor(so, o, 64);
            sa += __shfl_xor(sa, o, 64);
        }
        const int64_t g = row0 + r;
        if (part == 0 && g < n) {
            out[g] = so + p.b_out[0];
            att[g] = sa;
        }
    }

    // ---- the next layer's head on the x_out tile (still in TL[2]); its outputs reuse the z_k parking slots, which the
    // sweep above has already read -> barrier, then one GEMM for x1 and nblk for the projections, then a second flush.
    if (nx.nblk > 0) {
        __syncthreads();
        const float* xin = TL + 2 * SLOT;
        {
            const Bias8 bv = load_bias8(nx.bx1, wc);
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            mma_tile_frag_t(xin, wf, acc);
            load_w<PACKED>(wf, nx.wp[0], nx.ldwp, wc);
#pragma unroll
            for (int n2 = 0; n2 < 2; ++n2) {
                const int o = r16 * LDT + wc + 16 * n2 + 4 * kg;
                const float4 z = make_float4(acc[n2][0] + bv.v[n2].x, acc[n2][1] + bv.v[n2].y, acc[n2][2] + bv.v[n2].z,
                                             acc[n2][3] + bv.v[n2].w);
                *reinterpret_cast<float4*>(ZL + o) = z;                                                       // Zx1
                *reinterpret_cast<float4*>(ZL + SLOT + o) = make_float4(silu(z.x), silu(z.y), silu(z.z), silu(z.w));   // x1
            }
            __syncthreads();
        }
        for (int b = 0; b < nx.nblk; ++b) {
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            mma_tile_frag_t(ZL + SLOT, wf, acc);
            if (b + 1 < nx.nblk) load_w<PACKED>(wf, nx.wp[b + 1], nx.ldwp, wc);
            float* pb = ZL + (2 + b) * SLOT;
#pragma unroll
            for (int n2 = 0; n2 < 2; ++n2)
                *reinterpret_cast<float4*>(pb + r16 * LDT + wc + 16 * n2 + 4 * kg) =
                    make_float4(acc[n2][0], acc[n2][1], acc[n2][2], acc[n2][3]);
        }
        __syncthreads();
        TPROBE(42);
        sweep_rows<BMN>([&](int r, int c4) {
            const int64_t g = row0 + r;
            if (g >= n) return;
            if (nx.Zx1) stg4(nx.Zx1, g, DIM, c4, lds4(ZL, r, c4));
            stg4(nx.x1, g, DIM, c4, lds4(ZL + SLOT, r, c4));
            for (int b = 0; b < nx.nblk; ++b) stg4(nx.P + (int64_t)b * plane, g, DIM, c4, lds4(ZL + (2 + b) * SLOT, r, c4));
        });
        TPROBE(43);
    }
}

// ---- the forward chain for batches of several rounds (more row tiles than CUs) ------------------------------------------
// The kernel above parks everything the backward needs in LDS (152 KB: one workgroup per CU) so that no global store sits
// on a layer's critical path -- right when every workgroup is the only one its CU will ever see (143 tiles at the QM9
// batch).  With 1 188 tiles (PDBbind B = 32) a CU works through ~4.6 of them one after the other, and each one's
// latencies -- LDS before the first MFMA, the weight requests, the SiLU sequence, the barrier -- are exposed in turn
// (111 us per launch against 40 us of matrix-pipe time).  This form keeps only the five working tiles in LDS (42 KB:
// three workgroups per CU) and writes z_k / the taps / the next head's outputs straight from the accumulators, behind the
// next weight request (vmcnt retires in order: the request does not wait for them), the way node_heads_fwd_kernel does:
// one workgroup's epilogue runs under another's MFMAs.  Packed weights, deferred heads; same arithmetic, same results.
__global__ __launch_bounds__(WG, 3) void node_tail_fwd_lean_kernel(const float* __restrict__ x2, const float* __restrict__ res_x,
                                                                   int64_t n, TailParams p, float* __restrict__ Z,
                                                                   float* __restrict__ R, float* __restrict__ x_out,
                                                                   PreNext nx) {
    __shared__ __attribute__((aligned(16))) float lds[5 * SLOT];
    float* X0 = lds;
    float* RX = lds + SLOT;
    float* A = lds + 2 * SLOT;
    float* B = lds + 3 * SLOT;
    float* C = lds + 4 * SLOT;
    const int64_t row0 = (int64_t)blockIdx.x * BMN;
    const int64_t plane = n * DIM;
    const int lane = threadIdx.x & 63, r16 = lane & 15, kg = lane >> 4;
    const int wc = (threadIdx.x >> 6) * 32;

    WFrag wf;
    load_w<true>(wf, p.W[0], DIM, wc);
    sweep_rows<BMN>([&](int r, int c4) {
        const int64_t g = row0 + r;
        st_lds4(X0, r, c4, ldg4z(x2, g, n, DIM, c4));
        st_lds4(RX, r, c4, ldg4z(res_x, g, n, DIM, c4));
    });
    __syncthreads();

    // layer k: in -> dst = SiLU(W_k in + b_k) (+ add1 + add2); z_k and the optional tap go to memory from the accumulators
    // (operands swapped, gemm_core.h mma_tile_frag_t: a lane holds four consecutive channels of a row -- 16-byte stores)
    const bool trow = row0 + r16 < n;
    auto layer = [&](const float* in, float* dst, int k, const float* add1, const float* add2, float* tap,
                     const float* Wnext) {
        const Bias8 bv = load_bias8(p.b[k], wc);
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        mma_tile_frag_t(in, wf, acc);
        if (Wnext) load_w<true>(wf, Wnext, DIM, wc);
        float* zg = Z ? Z + (int64_t)k * plane : nullptr;
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2) {
            const int c0 = wc + 16 * n2 + 4 * kg, o = r16 * LDT + c0;
            const float4 z = make_float4(acc[n2][0] + bv.v[n2].x, acc[n2][1] + bv.v[n2].y, acc[n2][2] + bv.v[n2].z,
                                         acc[n2][3] + bv.v[n2].w);
            float4 a = make_float4(silu(z.x), silu(z.y), silu(z.z), silu(z.w));
            if (add1) {
                const float4 t = *reinterpret_cast<const float4*>(add1 + o);
                a.x += t.x, a.y += t.y, a.z += t.z, a.w += t.w;
            }
            if (add2) {
                const float4 t = *reinterpret_cast<const float4*>(add2 + o);
                a.x += t.x, a.y += t.y, a.z += t.z, a.w += t.w;
            }
            *reinterpret_cast<float4*>(dst + o) = a;
            if (trow) {
                if (zg) st_nt4(reinterpret_cast<float4*>(zg + (row0 + r16) * DIM + c0), z);      // (backward-only saves: streamed)
                if (tap) st_nt4(reinterpret_cast<float4*>(tap + (row0 + r16) * DIM + c0), a);
            }
        }
        __syncthreads();
    };
    float* const tap1 = Z ? R : nullptr;                      // r1, r2: backward-only saves (null in inference mode)
    float* const tap2 = Z ? R + plane : nullptr;
    layer(X0, A, 0, nullptr, nullptr, nullptr, nn(p.W[1]));        // h0 -> A
    layer(A, B, 1, nullptr, nullptr, nullptr, nn(p.W[2]));         // a1 -> B
    layer(B, C, 2, A, RX, tap1, nn(p.W[3]));                       // r1 -> C   (+ h0 + res_x)
    layer(C, A, 3, nullptr, nullptr, nullptr, nn(p.W[4]));         // a3 -> A
    layer(A, B, 4, C, nullptr, tap2, nn(p.W[5]));                  // r2 -> B   (+ r1)
    layer(B, A, 5, nullptr, nullptr, nullptr, nn(p.W[6]));         // a5 -> A
    layer(A, C, 6, B, nullptr, x_out, nn(nx.nblk > 0 ? nx.Wx1 : p.W[6]));   // r3 -> C = x_out

    // the next layer's head on the x_out tile (C): x1 -> A (and memory), the projections straight to memory
    if (nx.nblk > 0) {
        {
            const Bias8 bv = load_bias8(nx.bx1, wc);
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            mma_tile_frag_t(C, wf, acc);
            load_w<true>(wf, nx.wp[0], nx.ldwp, wc);
#pragma unroll
            for (int n2 = 0; n2 < 2; ++n2) {
                const int c0 = wc + 16 * n2 + 4 * kg;
                const float4 z = make_float4(acc[n2][0] + bv.v[n2].x, acc[n2][1] + bv.v[n2].y, acc[n2][2] + bv.v[n2].z,
                                             acc[n2][3] + bv.v[n2].w);
                const float4 a = make_float4(silu(z.x), silu(z.y), silu(z.z), silu(z.w));
                *reinterpret_cast<float4*>(A + r16 * LDT + c0) = a;
                if (trow) {
                    if (nx.Zx1) st_nt4(reinterpret_cast<float4*>(nx.Zx1 + (row0 + r16) * DIM + c0), z);
                    *reinterpret_cast<float4*>(nx.x1 + (row0 + r16) * DIM + c0) = a;
                }
            }
            __syncthreads();
        }
        for (int b = 0; b < nx.nblk; ++b) {
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            mma_tile_frag_t(A, wf, acc);
            if (b + 1 < nx.nblk) load_w<true>(wf, nx.wp[b + 1], nx.ldwp, wc);
            float* pb = nx.P + (int64_t)b * plane;
#pragma unroll
            for (int n2 = 0; n2 < 2; ++n2)
                if (trow)
                    *reinterpret_cast<float4*>(pb + (row0 + r16) * DIM + wc + 16 * n2 + 4 * kg) =
                        make_float4(acc[n2][0], acc[n2][1], acc[n2][2], acc[n2][3]);
        }
    }
}

// ---- the forward chain on the bf16 matrix pipe at fp32 accuracy ("bf16x6", gemm_core.h) ---------------------------------
// The production form of the kernel above (packed weights, deferred heads) with every GEMM as six bf16 piece products: a layer
// is 48 v_mfma_f32_16x16x32_bf16 per wave (768 matrix-pipe cycles) instead of 64 fp32 MFMAs (2 048).  Every GEMM input tile
// lives in LDS as three bf16 piece planes (edge_core.h st_pieces4 / lds_frag3p: 12 KB per 16-row tile, conflict-free on both
// sides), written by the epilogue that produces it -- the split is done once per element; the weights arrive as the edge-level
// kernels' bf16x3 fragment images (pamnet_pack_weights_mixed_f32 kind 1; 96 KB per matrix).
// Operands swapped like the fp32 form's (mma_tile_frag_t): the weight pieces are the A operand (lane: out channel wc + (l & 15),
// k = 8 (l >> 4) + 0..7 -- exactly what the image holds for the lane), the activation pieces the B operand (lane: node row
// l & 15, same k), so a lane's accumulator is row r16, channels wc + 16 n2 + 4 kg + 0..3: its pieces leave as three 8-byte
// stores, its pre-activations as one 16-byte store.  (Until round 6 this kernel had the operands the other way round and a
// k order of its own: 12 four-byte piece stores per lane and layer, epilogue 2 500 cycles against the fp32 form's 1 250.)
template <int NT>
struct WFragB {                               // a wave's NT 16-channel tiles: 48 VGPRs each
    edge::WFragB1 t[NT];
};
template <int NT>
__device__ __forceinline__ void load_wfragb(WFragB<NT>& f, const float* __restrict__ img, int wc) {
#pragma unroll
    for (int t = 0; t < NT; ++t) edge::load_wfragb1<false, 1>(f.t[t], img, 0, wc + 16 * t);
}
// acc[t] += (tile as piece planes at `in`) x (channels of slice t): 4 k-steps x 6 products, small products first; two independent
// accumulator chains either way (two tiles, or one tile's k-steps {0, 1} and {2, 3})
__device__ __forceinline__ void mma_planes_t(const char* __restrict__ in, const WFragB<2>& f, f32x4 (&acc)[2]) {
    Frag3 x[DIM / 32];
#pragma unroll
    for (int q = 0; q < DIM / 32; ++q) x[q] = edge::lds_frag3p(in, 0, q);
#pragma unroll
    for (int q = 0; q < DIM / 32; ++q) {
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t] = mfma_bf16(f.t[t].p[q][2], x[q].p[0], acc[t]);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t] = mfma_bf16(f.t[t].p[q][1], x[q].p[1], acc[t]);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t] = mfma_bf16(f.t[t].p[q][0], x[q].p[2], acc[t]);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t] = mfma_bf16(f.t[t].p[q][1], x[q].p[0], acc[t]);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t] = mfma_bf16(f.t[t].p[q][0], x[q].p[1], acc[t]);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t] = mfma_bf16(f.t[t].p[q][0], x[q].p[0], acc[t]);
    }
}
__device__ __forceinline__ f32x4 mma_strip_p(const char* __restrict__ P, const edge::WFragB1& f) {
    Frag3 x[DIM / 32];
#pragma unroll
    for (int q = 0; q < DIM / 32; ++q) x[q] = edge::lds_frag3p(P, 0, q);
    f32x4 a[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int h = 0; h < DIM / 64; ++h) {
#pragma unroll
        for (int t = 0; t < 2; ++t) a[t] = mfma_bf16(f.p[h + 2 * t][2], x[h + 2 * t].p[0], a[t]);
#pragma unroll
        for (int t = 0; t < 2; ++t) a[t] = mfma_bf16(f.p[h + 2 * t][1], x[h + 2 * t].p[1], a[t]);
#pragma unroll
        for (int t = 0; t < 2; ++t) a[t] = mfma_bf16(f.p[h + 2 * t][0], x[h + 2 * t].p[2], a[t]);
#pragma unroll
        for (int t = 0; t < 2; ++t) a[t] = mfma_bf16(f.p[h + 2 * t][1], x[h + 2 * t].p[0], a[t]);
#pragma unroll
        for (int t = 0; t < 2; ++t) a[t] = mfma_bf16(f.p[h + 2 * t][0], x[h + 2 * t].p[1], a[t]);
#pragma unroll
        for (int t = 0; t < 2; ++t) a[t] = mfma_bf16(f.p[h + 2 * t][0], x[h + 2 * t].p[0], a[t]);
    }
    return a[0] + a[1];
}
__device__ __forceinline__ void mma_planes_t(const char* __restrict__ in, const WFragB<1>& f, f32x4 (&acc)[1]) {
    acc[0] = mma_strip_p(in, f.t[0]);
}

// NWV waves (4: two 16-channel tiles per wave, one wave per SIMD; 8: one tile per wave, two waves per SIMD -- the epilogue of
// one under the MFMAs of the other, as in the backward chain)
template <bool RIDER, int NWV>
__global__ __launch_bounds__(64 * NWV) void node_tail_fwd_bf16_kernel(const float* __restrict__ x2,
                                                                     const float* __restrict__ res_x, int64_t n, TailParams p,
                                                                     float* __restrict__ Z, float* __restrict__ R,
                                                                     float* __restrict__ x_out, PreNext nx,
                                                                     Mlp2Rider rd = Mlp2Rider{}, LocalAgg la = LocalAgg{}) {
    // fp32 tiles: res_x, h0, 7 pre-activation tiles, 3 residual taps (parked, written out once after the chain);
    // piece-plane tiles: three, rotating through the chain
    constexpr int PT = edge::PTILE;
    constexpr int NT = 8 / NWV;                               // 16-channel tiles per wave
    constexpr int RPP = 2 * NWV, NJ = BMN / RPP;              // sweeps: rows per pass, passes
    __shared__ __attribute__((aligned(16))) float lds[12 * SLOT + 3 * PT / 4];
    if constexpr (RIDER) {
        if ((int)blockIdx.x >= rd.n_chain) {
            // 8 waves: the MLP's own 8-wave geometry (both matrices resident as pieces, chunks of up to 5 tiles -- a rider's ~5 tiles
            // in one chunk); 4 waves: two slices per wave, a matrix at a time, 3-tile chunks (no spills)
            constexpr int RMT = NWV == 8 ? 5 : 3;
            static_assert((12 * SLOT + 3 * PT / 4) * 4 >= RMT * edge::MLP2_TILE_B, "rider tiles must fit the chain's LDS");
            const int b = (int)blockIdx.x - rd.n_chain, nr = (int)gridDim.x - rd.n_chain;
            const int base = rd.ntiles / nr, rem = rd.ntiles % nr;
            const int64_t t0 = rd.tile0 + (int64_t)b * base + (b < rem ? b : rem);
            const int cnt = base + (b < rem ? 1 : 0);
            edge::Span sp;
            sp.beg = t0 * 16;
            const int64_t e = (t0 + cnt) * 16;
            sp.end = e < rd.m ? e : rd.m;
            if (sp.beg > sp.end) sp.beg = sp.end;
            const int nch = (cnt + RMT - 1) / RMT;
            sp.cmt = nch > 0 ? (cnt + nch - 1) / nch : 1;
            edge::mlp2_fwd_body<RMT, NWV>(rd.x, rd.set, sp, lds);
            return;
        }
    }
    float* RX = lds;
    float* H0 = lds + SLOT;
    float* ZL = lds + 2 * SLOT;               // [7] z_k tiles (reused by the next layer's head: Zx1, x1, P_b)
    float* TL = lds + 9 * SLOT;               // [3] r1, r2, x_out tiles
    char* P0 = reinterpret_cast<char*>(lds + 12 * SLOT);
    char* P1 = P0 + PT;
    char* P2 = P1 + PT;
    const int64_t row0 = (int64_t)blockIdx.x * BMN;
    const int64_t plane = n * DIM;
    const int lane = threadIdx.x & 63, r16 = lane & 15, kg = lane >> 4;
    const int wc = (threadIdx.x >> 6) * (16 * NT);
    const int sc4 = threadIdx.x & 31, sr = threadIdx.x >> 5;  // sweep coordinates: rows sr + RPP j

    TPROBE(40);
    WFragB<NT> wf;
    load_wfragb(wf, p.W[0], wc);
    if (la.m_ji) {                                            // (workgroup-uniform) the x2 rows are formed here
        int64_t node[NJ];
        bool ok[NJ];
        float4 xin[NJ], rx[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) node[j] = row0 + sr + RPP * j, ok[j] = node[j] < n;
#pragma unroll
        for (int j = 0; j < NJ; ++j) rx[j] = ldg4z(res_x, node[j], n, DIM, sc4);
        local_agg_rows<NJ>(la, node, ok, sc4, xin);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            st_lds4(RX, sr + RPP * j, sc4, rx[j]);
            edge::st_pieces4(P0, sr + RPP * j, sc4, xin[j]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int r = sr + RPP * j;
            const int64_t g = row0 + r;
            st_lds4(RX, r, sc4, ldg4z(res_x, g, n, DIM, sc4));
            edge::st_pieces4(P0, r, sc4, ldg4z(x2, g, n, DIM, sc4));
        }
    }
    __syncthreads();

    // layer k: planes `in` -> planes `dst` = SiLU(W_k in + b_k) (+ add1 + add2); z_k parked; fp32 copies of the result
    // where something reads it as a residual (`dst32`) or the backward wants it (`tap`)
    auto layer = [&](const char* in, char* dst, int k, const float* add1, const float* add2, float* dst32, float* tap,
                     const float* Wnext) {
        TPROBE(4 * k);
        float4 bv[NT];                                         // before the prefetch (in-order vmcnt)
#pragma unroll
        for (int n2 = 0; n2 < NT; ++n2) bv[n2] = *reinterpret_cast<const float4*>(p.b[k] + wc + 16 * n2 + 4 * kg);
        f32x4 acc[NT];
#pragma unroll
        for (int n2 = 0; n2 < NT; ++n2) acc[n2] = f32x4{0.f, 0.f, 0.f, 0.f};
        mma_planes_t(in, wf, acc);                             // lane: row r16, channels wc + 16 n2 + 4 kg + 0..3
        TPROBE(4 * k + 1);
        if (Wnext) load_wfragb(wf, Wnext, wc);
        float* zk = ZL + k * SLOT;
#pragma unroll
        for (int n2 = 0; n2 < NT; ++n2) {
            const int o = r16 * LDT + wc + 16 * n2 + 4 * kg;
            const float4 z = make_float4(acc[n2][0] + bv[n2].x, acc[n2][1] + bv[n2].y, acc[n2][2] + bv[n2].z, acc[n2][3] + bv[n2].w);
            float4 a = make_float4(silu(z.x), silu(z.y), silu(z.z), silu(z.w));
            if (add1) {
                const float4 t = *reinterpret_cast<const float4*>(add1 + o);
                a.x += t.x, a.y += t.y, a.z += t.z, a.w += t.w;
            }
            if (add2) {
                const float4 t = *reinterpret_cast<const float4*>(add2 + o);
                a.x += t.x, a.y += t.y, a.z += t.z, a.w += t.w;
            }
            edge::st_pieces4(dst, r16, (wc + 16 * n2 + 4 * kg) >> 2, a);
            *reinterpret_cast<float4*>(zk + o) = z;
            if (dst32) *reinterpret_cast<float4*>(dst32 + o) = a;
            if (tap) *reinterpret_cast<float4*>(tap + o) = a;
        }
        TPROBE(4 * k + 2);
        __syncthreads();
        TPROBE(4 * k + 3);
    };

    layer(P0, P1, 0, nullptr, nullptr, H0, nullptr, nn(p.W[1]));              // h0 -> P1 (+ fp32 H0)
    layer(P1, P2, 1, nullptr, nullptr, nullptr, nullptr, nn(p.W[2]));         // a1 -> P2
    layer(P2, P0, 2, H0, RX, nullptr, TL, nn(p.W[3]));                        // r1 -> P0   (+ h0 + res_x)
    layer(P0, P1, 3, nullptr, nullptr, nullptr, nullptr, nn(p.W[4]));         // a3 -> P1
    layer(P1, P2, 4, TL, nullptr, nullptr, TL + SLOT, nn(p.W[5]));            // r2 -> P2   (+ r1)
    layer(P2, P1, 5, nullptr, nullptr, nullptr, nullptr, nn(p.W[6]));         // a5 -> P1
    layer(P1, P0, 6, TL + SLOT, nullptr, nullptr, TL + 2 * SLOT, nn(nx.nblk > 0 ? nx.Wx1 : p.W[6]));   // r3 = x_out -> P0

    // park -> memory: Z[7][n][128], R[2][n][128], x_out[n][128]
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int r = sr + RPP * j;
        const int64_t g = row0 + r;
        if (g >= n) continue;
        if (Z) {                                              // backward-only saves: null in inference mode
#pragma unroll
            for (int k = 0; k < 7; ++k) stg4_nt(Z + (int64_t)k * plane, g, DIM, sc4, lds4(ZL + k * SLOT, r, sc4));
            stg4_nt(R, g, DIM, sc4, lds4(TL, r, sc4));
            stg4_nt(R + plane, g, DIM, sc4, lds4(TL + SLOT, r, sc4));
        }
        stg4(x_out, g, DIM, sc4, lds4(TL + 2 * SLOT, r, sc4));
    }
    TPROBE(41);

    // ---- the next layer's head on the x_out tile (its planes are in P0); outputs reuse the z_k parking slots, which the
    // sweep above has already read -> barrier, one GEMM for x1, nblk for the projections, then a second flush
    if (nx.nblk > 0) {
        __syncthreads();
        {
            float4 bv[NT];
#pragma unroll
            for (int n2 = 0; n2 < NT; ++n2) bv[n2] = *reinterpret_cast<const float4*>(nx.bx1 + wc + 16 * n2 + 4 * kg);
            f32x4 acc[NT];
#pragma unroll
            for (int n2 = 0; n2 < NT; ++n2) acc[n2] = f32x4{0.f, 0.f, 0.f, 0.f};
            mma_planes_t(P0, wf, acc);
            load_wfragb(wf, nx.wp[0], wc);
#pragma unroll
            for (int n2 = 0; n2 < NT; ++n2) {
                const int o = r16 * LDT + wc + 16 * n2 + 4 * kg;
                const float4 z = make_float4(acc[n2][0] + bv[n2].x, acc[n2][1] + bv[n2].y, acc[n2][2] + bv[n2].z,
                                             acc[n2][3] + bv[n2].w);
                const float4 a = make_float4(silu(z.x), silu(z.y), silu(z.z), silu(z.w));
                *reinterpret_cast<float4*>(ZL + o) = z;                    // Zx1
                *reinterpret_cast<float4*>(ZL + SLOT + o) = a;             // x1
                edge::st_pieces4(P1, r16, (wc + 16 * n2 + 4 * kg) >> 2, a);
            }
            __syncthreads();
        }
        for (int b = 0; b < nx.nblk; ++b) {
            f32x4 acc[NT];
#pragma unroll
            for (int n2 = 0; n2 < NT; ++n2) acc[n2] = f32x4{0.f, 0.f, 0.f, 0.f};
            mma_planes_t(P1, wf, acc);
            if (b + 1 < nx.nblk) load_wfragb(wf, nx.wp[b + 1], wc);
            float* pb = ZL + (2 + b) * SLOT;
#pragma unroll
            for (int n2 = 0; n2 < NT; ++n2)
                *reinterpret_cast<float4*>(pb + r16 * LDT + wc + 16 * n2 + 4 * kg) =
                    make_float4(acc[n2][0], acc[n2][1], acc[n2][2], acc[n2][3]);
        }
        __syncthreads();
        TPROBE(42);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int r = sr + RPP * j;
            const int64_t g = row0 + r;
            if (g >= n) continue;
            if (nx.Zx1) stg4_nt(nx.Zx1, g, DIM, sc4, lds4(ZL, r, sc4));       // (backward-only save)
            stg4(nx.x1, g, DIM, sc4, lds4(ZL + SLOT, r, sc4));
            for (int b = 0; b < nx.nblk; ++b) stg4(nx.P + (int64_t)b * plane, g, DIM, sc4, lds4(ZL + (2 + b) * SLOT, r, sc4));
        }
        TPROBE(43);
    }
}

// Head branch of every layer in one launch: o3 = mlp_out(x_out), out = W_out . o3 + b_out, att = W . o3
// (layers/global_message_passing.py:46-50).  grid = (ceil(n/16), layers).
constexpr int MAX_HEAD_LAYERS = 16;
struct HeadLayer {
    const float* x_out;       // [n][128] chain output of this layer
    const float* W[3];        // mlp_out matrices (row-major, or fragment images when packed)
    const float* b[3];
    const float* w_out;
    const float* b_out;
    const float* w_att;
    float* Z;                 // this layer's [10][n][128] pre-activation block (slots 7..9 written) or null
    float* out;
    float* att;
};
struct HeadBatch {
    HeadLayer l[MAX_HEAD_LAYERS];
};

// HM 16-row tiles per workgroup share every weight slice (3 x 64 KB per workgroup from L2).  Measured at the QM9 batch
// (1 716 tiles): HM = 1 39.6 us, 2 42.3 us, 4 54.4 us -- neither the L2 weight stream nor occupancy (25 vs 50 KB of LDS:
// no change) bounds it, the three dependent GEMMs per tile do; PAMNET_HEADS_TILES=2|4 selects the wider forms.
template <bool PACKED, int HM>
__global__ __launch_bounds__(WG) void node_heads_fwd_kernel(HeadBatch hb, int64_t n) {
    // Unlike the chains, this launch has several workgroups per CU to hide a store behind, so the pre-activations go to
    // memory straight from the accumulators (after the next weight slice has been requested: vmcnt retires in order,
    // the prefetch does not wait for them).
    constexpr int BM = 16 * HM;
    constexpr int HSLOT = BM * LDT;
    __shared__ __attribute__((aligned(16))) float lds[3 * HSLOT];
    float* X = lds;
    float* A = lds + HSLOT;
    float* B = lds + 2 * HSLOT;
    const HeadLayer& hl = hb.l[blockIdx.y];
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    const int64_t plane = n * DIM;
    const int lane = threadIdx.x & 63, r16 = lane & 15, kg = lane >> 4;
    const int wc = (threadIdx.x >> 6) * 32;
    WFrag wf;
    load_w<PACKED>(wf, hl.W[0], DIM, wc);
    sweep_rows<BM>([&](int r, int c4) { st_lds4(X, r, c4, ldg4z(hl.x_out, row0 + r, n, DIM, c4)); });
    __syncthreads();
    // (operands swapped, gemm_core.h mma_tile_frag_t: a lane holds four consecutive channels of one row -- the pre-activations
    // leave as 16-byte stores, four times fewer than channel by channel; same bits)
    auto layer = [&](const float* in, float* dst, int k, const float* Wnext) {
        const Bias8 bv = load_bias8(hl.b[k], wc);
        f32x4 acc[HM][2];
#pragma unroll
        for (int m = 0; m < HM; ++m) {
            acc[m][0] = acc[m][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            mma_tile_frag_t(in + m * 16 * LDT, wf, acc[m]);
        }
        if (Wnext) load_w<PACKED>(wf, Wnext, DIM, wc);
        float* zg = hl.Z ? hl.Z + (int64_t)(7 + k) * plane : nullptr;
#pragma unroll
        for (int m = 0; m < HM; ++m)
#pragma unroll
            for (int n2 = 0; n2 < 2; ++n2) {
                const int rw = 16 * m + r16, c0 = wc + 16 * n2 + 4 * kg;
                const float4 z = make_float4(acc[m][n2][0] + bv.v[n2].x, acc[m][n2][1] + bv.v[n2].y, acc[m][n2][2] + bv.v[n2].z,
                                             acc[m][n2][3] + bv.v[n2].w);
                *reinterpret_cast<float4*>(dst + rw * LDT + c0) = make_float4(silu(z.x), silu(z.y), silu(z.z), silu(z.w));
                // (a backward-only save, but cached: the head branch's backward is the first kernel of the backward and reads it back)
                if (zg && row0 + rw < n) *reinterpret_cast<float4*>(zg + (row0 + rw) * DIM + c0) = z;
            }
        __syncthreads();
    };
    layer(X, A, 0, nn(hl.W[1]));                  // o1
    layer(A, B, 1, nn(hl.W[2]));                  // o2
    layer(B, A, 2, nullptr);                  // o3 -> A
    // heads: 16 lanes per row, 8 columns each, butterfly over the 16-lane group; 16 rows per pass
#pragma unroll
    for (int m = 0; m < HM; ++m) {
        const int r = 16 * m + (threadIdx.x >> 4), part = threadIdx.x & 15;
        float so = 0.f, sa = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float v = A[r * LDT + part * 8 + c];
            so += v * hl.w_out[part * 8 + c];
            sa += v * hl.w_att[part * 8 + c];
        }
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) {
            so += __shfl_xor(so, o, 64);
            sa += __shfl_xor(sa, o, 64);
        }
        const int64_t g = row0 + r;
        if (part == 0 && g < n) {
            hl.out[g] = so + hl.b_out[0];
            hl.att[g] = sa;
        }
    }
}

// Backward of the chain.  Produces dZ_k for every layer (consumed by the batched weight-gradient kernel), d x2 (gradient
// of the chain input), d res_x, and per-workgroup partial sums for the two head vectors.
// HEADS = false: the head branch was differentiated by node_heads_bwd_kernel (all layers in one launch at the start of
// the backward: it only needs d out / d att, which the loss hands over for every layer at once); `d_out` then carries
// its contribution to d x_out ([n][128]) and the chain starts at layer 6.
// PRE = true (packed weights, HEADS = false): the kernel first runs the backward of the NEXT layer's head on the same
// row tile (what node_pre_bwd_kernel does: d x1 = sum_b dP_b Wp_b + d x1_direct, dz_x1 = d x1 * SiLU'(z_x1),
// d x = dz_x1 Wx1 + d_add) and feeds its d x straight into this chain as d x_out -- the two kernels ran back to back on
// the same tiles with a [n,128] round trip and a launch between them.
struct PreBwd {
    const float* dP;          // [nblk][n][128]
    const float* wp[4];       // transposed-orientation images of the projection blocks
    const float* Wx1;         // transposed-orientation image
    const float* Zx1;         // [n][128]
    const float* dx1_direct;  // [n][128]
    const float* d_add;       // [n][128]
    float* dZx1;              // [n][128] out
    int nblk;
    // Round 6: plane b of dP may be FORMED here instead of read -- the segment sum the engine used to launch ahead of this kernel
    // (the source-side reduction of the global layer's d z; the four of the local layer): gsrc[b] != null: plane b, row i =
    // the sum over q in [gptr[b][i], gptr[b][i + 1]) of gsrc[b][gperm[b] ? gperm[b][q] : q], in that order; written to
    // dP_out + b * plane for the weight-gradient jobs that read the plane later.
    const float* gsrc[4];
    const int32_t* gptr[4];
    const int32_t* gperm[4];
    float* dP_out;
};
// Plane rows `row` (this thread's float4 column) of up to four planes as the segment sums of their rows of gsrc[b], in CSR order,
// formed TOGETHER: one thread's sums are chains of dependent round trips (row range -> permutation entries -> rows), and plane
// by plane, four rows at a time, a local layer's four planes of degree ~2 were ~20 of them in a row ahead of the chain.  Here the
// row ranges of all planes are one request (gather_begin: issued ahead of the tile's other loads), then every step requests U rows
// of every plane at once, with the permutation entries of the NEXT step in flight beside them: 2 + ceil(max degree / U) round
// trips.  The additions of a plane are in CSR order as before (same bits as pamnet_segment_sum_multi_f32).
constexpr int GU = 4;                          // rows in flight per plane and step
struct GatherState {
    int q[4], q1[4];
};
__device__ __forceinline__ void gather_begin(GatherState& gs, const PreBwd& pb, int64_t row, bool in_range) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        gs.q[b] = gs.q1[b] = 0;
        if (b < pb.nblk && pb.gsrc[b] && in_range) gs.q[b] = pb.gptr[b][row], gs.q1[b] = pb.gptr[b][row + 1];
    }
}
__device__ __forceinline__ void gather_finish(GatherState& gs, const PreBwd& pb, int c4, float4 (&s)[4]) {
    int idx[4][GU];
    auto fetch_idx = [&](int (&out)[4][GU]) __attribute__((always_inline)) {
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int qq = gs.q[b] + u;
                out[b][u] = -1;
                if (qq < gs.q1[b]) out[b][u] = pb.gperm[b] ? pb.gperm[b][qq] : qq;
            }
    };
    fetch_idx(idx);
#pragma unroll
    for (int b = 0; b < 4; ++b) s[b] = f4zero();
    while (gs.q[0] < gs.q1[0] || gs.q[1] < gs.q1[1] || gs.q[2] < gs.q1[2] || gs.q[3] < gs.q1[3]) {
        float4 v[4][GU];
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int u = 0; u < GU; ++u)
                if (idx[b][u] >= 0) v[b][u] = ldg4(pb.gsrc[b], idx[b][u], DIM, c4);
#pragma unroll
        for (int b = 0; b < 4; ++b) gs.q[b] += GU;
        int nidx[4][GU];
        fetch_idx(nidx);                                      // (in flight beside the rows)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                if (idx[b][u] >= 0) s[b] = f4add(s[b], v[b][u]);
                idx[b][u] = nidx[b][u];
            }
    }
}

template <bool PACKED, bool HEADS, bool PRE = false>
__global__ __launch_bounds__(TWG) void node_tail_bwd_kernel(const float* __restrict__ d_xout /* may be null */,
                                                            const float* __restrict__ d_out,
                                                            const float* __restrict__ d_att, int64_t n, TailParams p,
                                                            const float* __restrict__ Z, float* __restrict__ dZ,
                                                            float* __restrict__ d_x2, float* __restrict__ d_resx,
                                                            float* __restrict__ head_partial /* [grid][257] */,
                                                            PreBwd pb = PreBwd{}, WBatchS rider = WBatchS{},
                                                            float* __restrict__ rider_partial = nullptr, int n_tiles = 0) {
    static_assert(!PRE || (PACKED && !HEADS), "the fused head backward needs packed weights and deferred heads");
    // All ten pre-activation tiles are fetched up front (one burst of coalesced rows) and every dz_k overwrites its z_k
    // in place; d x2 / d res_x are parked too and the tiles leave as coalesced rows after the chain: like the forward,
    // the per-layer critical path holds no global access except the next weight slice.
    __shared__ __attribute__((aligned(16))) float lds[14 * SLOT + 16 * 256 + 16];
    if constexpr (PRE) {
        // Riders: the chain owns ceil(n/16) workgroups (143 of 256 CUs at the QM9 batch); the workgroups behind them are
        // split-K slots of the PREVIOUS chain's weight gradients (its dZ planes and saved activations are final), one per
        // otherwise idle CU -- the weight-gradient launch of that layer shrinks by what rides here.
        static_assert(sizeof(lds) >= sizeof(float) * WGRAD_LDS_FLOATS, "rider staging must fit the chain's LDS");
        if ((int)blockIdx.x >= n_tiles) {
#ifndef PAMNET_RIDER_4W
            // all eight waves (wave tile 32x64): the bf16x6 inner loop holds the split fragments of two pipeline stages and
            // does not fit the 256 registers of a two-waves-per-SIMD kernel with the 64x64 tile; two waves per SIMD also
            // interleave one wave's splits with the other's MFMAs in hardware
            wgrad_body<8>(rider, rider_partial, (int)blockIdx.x - n_tiles, lds);
#else
            // four of the eight waves, one per SIMD, each with the 64x64 wave tile of the stand-alone kernel (the other
            // four leave: a terminated wave no longer takes part in the workgroup barrier)
            if (threadIdx.x < 256) wgrad_body<4>(rider, rider_partial, (int)blockIdx.x - n_tiles, lds);
#endif
            return;
        }
    }
    float* D0 = lds;                  // dz ping
    float* D1 = lds + SLOT;           // dz pong
    float* K = lds + 2 * SLOT;        // residual gradient kept across a Res block
    float* EX = lds + 3 * SLOT;       // d res_x
    float* ZL = lds + 4 * SLOT;       // [10] z_k, then dz_k
    float* red = lds + 14 * SLOT;     // [16][256] head partials (+16 for d b_out)
    const int64_t row0 = (int64_t)blockIdx.x * BMN;
    const int64_t plane = n * DIM;
    const Frag fr;
    const int c = fr.col();
    const int to = fr.r16 * LDT + fr.wc + 4 * fr.kg;          // (transposed accumulators: row r16, channels wc + 4 kg + 0..3)
    const bool trow = row0 + fr.r16 < n;                      // ... and whether that row exists
    // (a factor, not a select: with `trow ? x : 0` the compiler runs the four SiLU' sequences of a lane one after the other under
    // exec masks -- four dependent exp / rcp chains in a row -- instead of interleaved; the rows past n are zero either way)
    const float tmask = trow ? 1.f : 0.f;

    constexpr int KTOP = HEADS ? 9 : 6;                       // first matrix of the backward chain
    constexpr int NZ = HEADS ? 10 : 7;
    WFrag1 wf;
    if constexpr (PRE) load_wfrag1_img(wf, pb.wp[0]);
    else if constexpr (PACKED) load_wfrag1_img(wf, p.W[KTOP]);
    else load_wfrag1<true>(wf, p.W[KTOP], fr.wc);
    const int sc4 = threadIdx.x & 31, sr = threadIdx.x >> 5;          // sweep coordinates: 512 threads = 16 rows x 32 float4
    const int64_t sg = row0 + sr;
    {
        GatherState gst;
        if constexpr (PRE) gather_begin(gst, pb, sg, sg < n);   // (the row ranges of the planes formed here: first request of all)
#pragma unroll
        for (int k = 0; k < NZ; ++k) st_lds4(ZL + k * SLOT, sr, sc4, ldg4z(Z + (int64_t)k * plane, sg, n, DIM, sc4));
        float4 kx = d_xout ? ldg4z(d_xout, sg, n, DIM, sc4) : f4zero();
        if constexpr (!HEADS) kx = f4add(kx, ldg4z(d_out, sg, n, DIM, sc4));      // + the head branch's d x_out
        if constexpr (PRE) {
            // tiles of the head backward: dP planes -> ZL[7..9] + the (unused) head-partial area, z_x1 -> EX,
            // d x1_direct -> D1, d_add joins K (K = d_add + g_head; the head's d x is added below)
            float* const pl[4] = {ZL + 7 * SLOT, ZL + 8 * SLOT, ZL + 9 * SLOT, red};
            float4 gsum[4];
            gather_finish(gst, pb, sc4, gsum);                 // (rows past n: empty ranges, zeros)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (b >= pb.nblk) break;
                float4 v;
                if (pb.gsrc[b]) {                              // (workgroup-uniform) the plane is formed here, and kept
                    v = gsum[b];
                    if (sg < n) stg4(pb.dP_out + (int64_t)b * plane, sg, DIM, sc4, v);
                } else {
                    v = ldg4z(pb.dP + (int64_t)b * plane, sg, n, DIM, sc4);
                }
                st_lds4(pl[b], sr, sc4, v);
            }
            st_lds4(EX, sr, sc4, ldg4z(pb.Zx1, sg, n, DIM, sc4));
            st_lds4(D1, sr, sc4, ldg4z(pb.dx1_direct, sg, n, DIM, sc4));
            kx = f4add(kx, ldg4z(pb.d_add, sg, n, DIM, sc4));
        }
        st_lds4(K, sr, sc4, kx);
    }
    __syncthreads();

    if constexpr (PRE) {
        float* const pl[4] = {ZL + 7 * SLOT, ZL + 8 * SLOT, ZL + 9 * SLOT, red};
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int b = 0; b < pb.nblk; ++b) {
            acc += mma_strip_t(pl[b], wf);
            load_wfrag1_img(wf, b + 1 < pb.nblk ? pb.wp[b + 1] : pb.Wx1);
        }
        st_f32x4(D0 + to, acc);
        __syncthreads();
        {   // dz_x1 = (d x1) * SiLU'(z_x1): over D1 in place, and parked in `red` for the final coalesced store
            const float4 d1 = f4add(lds4(D0, sr, sc4), lds4(D1, sr, sc4));
            const float4 dzx = f4mul(d1, f4dsilu(lds4(EX, sr, sc4)));
            st_lds4(D1, sr, sc4, dzx);
            st_lds4(red, sr, sc4, dzx);
        }
        __syncthreads();
        const f32x4 a2 = mma_strip_t(D1, wf);
        load_wfrag1_img(wf, p.W[KTOP]);
        st_f32x4(K + to, lds_f32x4(K + to) + a2);              // d x_out = head's d x + d_add + g_head
        __syncthreads();
    }

    auto load_z = [&](int k) {
        f32x4 z;
#pragma unroll
        for (int r = 0; r < 4; ++r) z[r] = ZL[k * SLOT + fr.row(r) * LDT + c];
        return z;
    };

    if constexpr (HEADS) {
    // head-vector partials: sum_rows d_out * o3, sum_rows d_att * o3, sum_rows d_out  (o3 = SiLU(z9))
    {
        float4 so = f4zero(), sa = f4zero();
        float sb = 0.f;
        if (sg < n) {
            const float4 o3 = f4silu(lds4(ZL + 9 * SLOT, sr, sc4));
            const float go = d_out[sg], ga = d_att[sg];
            so = make_float4(go * o3.x, go * o3.y, go * o3.z, go * o3.w);
            sa = make_float4(ga * o3.x, ga * o3.y, ga * o3.z, ga * o3.w);
            sb = go;
        }
        float* mine = red + sr * 256;
        *reinterpret_cast<float4*>(mine + 4 * sc4) = so;
        *reinterpret_cast<float4*>(mine + 128 + 4 * sc4) = sa;
        if (sc4 == 0) red[16 * 256 + sr] = sb;
        __syncthreads();
        if (threadIdx.x < 256) {
            float tot = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) tot += red[q * 256 + threadIdx.x];
            head_partial[(int64_t)blockIdx.x * 257 + threadIdx.x] = tot;
        } else if (threadIdx.x == 256) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += red[16 * 256 + q];
            head_partial[(int64_t)blockIdx.x * 257 + 256] = t;
        }
    }

    // dz9 = (d_out * w_out + d_att * w_att) * SiLU'(z9) -> D0 and in place of z9
    {
        const f32x4 z9 = load_z(9);
        const float wo = p.w_out[c], wa = p.w_att[c];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rw = fr.row(r);
            const int64_t g = row0 + rw;
            const float dz = (g < n) ? (d_out[g] * wo + d_att[g] * wa) * dsilu(z9[r]) : 0.f;
            D0[rw * LDT + c] = dz;
            ZL[9 * SLOT + rw * LDT + c] = dz;
        }
        __syncthreads();
    }
    } else {
        // dz6 = d r3 * SiLU'(z6), d r3 = d x_out (next layer + head branch) already in K
        const f32x4 z6 = lds_f32x4(ZL + 6 * SLOT + to), kv = lds_f32x4(K + to);
        f32x4 dz;
#pragma unroll
        for (int r = 0; r < 4; ++r) dz[r] = kv[r] * dsilu(z6[r]) * tmask;
        st_f32x4(D1 + to, dz);
        st_f32x4(ZL + 6 * SLOT + to, dz);
        __syncthreads();
    }

    // One backward step: v = dz_k * W_k (+ K) ; optionally K <- v, extra <- v ; then dz_{k-1} = v * SiLU'(z_{k-1}).
    // k == 0 ends the chain: v = d x2 (left in dst).
    auto back = [&](const float* in, float* dst, int k, bool add_k, bool keep_k, float* extra) {
        TPROBE(4 * k);
        f32x4 zn = {0.f, 0.f, 0.f, 0.f};
        if (k > 0) zn = lds_f32x4(ZL + (k - 1) * SLOT + to);
        f32x4 v = mma_strip_t(in, wf);
        TPROBE(4 * k + 1);
        if (k > 0) {
            if constexpr (PACKED) load_wfrag1_img(wf, p.W[k - 1]);
            else load_wfrag1<true>(wf, p.W[k - 1], fr.wc);
        }
        if (add_k) v += lds_f32x4(K + to);
        if (keep_k) st_f32x4(K + to, v);
        if (extra) st_f32x4(extra + to, v);
        if (k == 0) {
            st_f32x4(dst + to, v);
        } else {
            f32x4 dz;
#pragma unroll
            for (int r = 0; r < 4; ++r) dz[r] = v[r] * dsilu(zn[r]) * tmask;
            st_f32x4(dst + to, dz);
            st_f32x4(ZL + (k - 1) * SLOT + to, dz);
        }
        TPROBE(4 * k + 2);
        __syncthreads();
        TPROBE(4 * k + 3);
    };

    if constexpr (HEADS) {
        back(D0, D1, 9, false, false, nullptr);    // d o2          -> dz8
        back(D1, D0, 8, false, false, nullptr);    // d o1          -> dz7
        back(D0, D1, 7, true, true, nullptr);      // d r3 = . + d x_out (kept)   -> dz6
    }
    back(D1, D0, 6, false, false, nullptr);        // d a5          -> dz5
    back(D0, D1, 5, true, true, nullptr);          // d r2 = . + d r3 (kept)      -> dz4
    back(D1, D0, 4, false, false, nullptr);        // d a3          -> dz3
    back(D0, D1, 3, true, true, EX);               // d r1 = . + d r2 (kept, = d res_x) -> dz2
    back(D1, D0, 2, false, false, nullptr);        // d a1          -> dz1
    back(D0, D1, 1, true, false, nullptr);         // d h0 = . + d r1             -> dz0
    back(D1, D0, 0, false, false, nullptr);        // d x2 -> D0

    if (sg < n) {
#pragma unroll
        for (int k = 0; k < NZ; ++k) stg4(dZ + (int64_t)k * plane, sg, DIM, sc4, lds4(ZL + k * SLOT, sr, sc4));
        stg4(d_x2, sg, DIM, sc4, lds4(D0, sr, sc4));
        stg4(d_resx, sg, DIM, sc4, lds4(EX, sr, sc4));
        if constexpr (PRE) stg4(pb.dZx1, sg, DIM, sc4, lds4(red, sr, sc4));
    }
}

// ---- the backward chain on the bf16 matrix pipe at fp32 accuracy ("bf16x6") ----------------------------------------------
// node_tail_bwd_kernel<true, false, PRE> (packed weights, deferred heads: the production form of single-round batches) with
// every GEMM as six bf16 piece products, the way node_tail_fwd_bf16_kernel runs the forward: 24 v_mfma_f32_16x16x32_bf16 per
// wave and layer (two waves per SIMD: 768 matrix-pipe cycles) instead of 32 fp32 MFMAs (2 048).  Every GEMM input -- the head's
// d P planes, dz_x1, the dz_k ping / pong -- is written once as three bf16 piece planes (edge_core.h st_pieces4) by the sweep or
// the epilogue that forms it; the weights are kind-1 fragment images in the transposed orientation, the A operand of the MFMA
// (operands swapped, mma_strip_t: a lane holds row r16, channels wc + 4 kg + 0..3).  The fp32 tiles that are not GEMM inputs
// (z_k / dz_k, the kept residual gradient, d res_x, d x2) stay as they were.
__device__ __forceinline__ void st_pieces_acc(char* __restrict__ P, const Frag& fr, const f32x4& v) {
    edge::st_pieces4(P, fr.r16, (fr.wc + 4 * fr.kg) >> 2, make_float4(v[0], v[1], v[2], v[3]));
}

template <bool PRE>
__global__ __launch_bounds__(TWG) void node_tail_bwd_bf16_kernel(const float* __restrict__ d_xout /* may be null */,
                                                                 const float* __restrict__ d_out, int64_t n, TailParams p,
                                                                 const float* __restrict__ Z, float* __restrict__ dZ,
                                                                 float* __restrict__ d_x2, float* __restrict__ d_resx,
                                                                 PreBwd pb = PreBwd{}, WBatchS rider = WBatchS{},
                                                                 float* __restrict__ rider_partial = nullptr, int n_tiles = 0) {
    constexpr int PT = edge::PTILE;
    __shared__ __attribute__((aligned(16))) float lds[12 * SLOT + 4 * PT / 4];
    if constexpr (PRE) {
        static_assert(sizeof(lds) >= sizeof(float) * WGRAD_LDS_FLOATS, "rider staging must fit the chain's LDS");
        if ((int)blockIdx.x >= n_tiles) {                     // riders: see node_tail_bwd_kernel
#ifdef PAMNET_PROBE_SKIP_RIDERS
            return;       // development probe (never in libpamnet_hip.so): WRONG weight gradients, the launch without its riders' time
#endif
#ifndef PAMNET_RIDER_4W
            wgrad_body<8>(rider, rider_partial, (int)blockIdx.x - n_tiles, lds);
#else
            if (threadIdx.x < 256) wgrad_body<4>(rider, rider_partial, (int)blockIdx.x - n_tiles, lds);
#endif
            return;
        }
    }
    float* D0 = lds;                  // fp32: the head's d x1 partial, at the end d x2
    float* D1 = lds + SLOT;           // fp32: d x1_direct
    float* K = lds + 2 * SLOT;        // residual gradient kept across a Res block
    float* EX = lds + 3 * SLOT;       // z_x1, then d res_x
    float* ZL = lds + 4 * SLOT;       // [7] z_k, then dz_k
    float* RED = lds + 11 * SLOT;     // dz_x1, parked for the final coalesced store
    char* PL = reinterpret_cast<char*>(lds + 12 * SLOT);      // [4] piece planes: the d P planes, then dz_x1 | dz ping | dz pong
    const int64_t row0 = (int64_t)blockIdx.x * BMN;
    const int64_t plane = n * DIM;
    const Frag fr;
    const int to = fr.r16 * LDT + fr.wc + 4 * fr.kg;          // (transposed accumulators: row r16, channels wc + 4 kg + 0..3)
    const bool trow = row0 + fr.r16 < n;
    // (a factor, not a select: with `trow ? x : 0` the compiler runs the four SiLU' sequences of a lane one after the other under
    // exec masks -- four dependent exp / rcp chains in a row -- instead of interleaved; the rows past n are zero either way)
    const float tmask = trow ? 1.f : 0.f;

    TPROBE(40);
    edge::WFragB1 wf;
    edge::load_wfragb1<true, 1>(wf, PRE ? pb.wp[0] : p.W[6], 0, fr.wc);
    const int sc4 = threadIdx.x & 31, sr = threadIdx.x >> 5;          // sweep coordinates: 512 threads = 16 rows x 32 float4
    const int64_t sg = row0 + sr;
    {
        GatherState gst;
        if constexpr (PRE) gather_begin(gst, pb, sg, sg < n);
#pragma unroll
        for (int k = 0; k < 7; ++k) st_lds4(ZL + k * SLOT, sr, sc4, ldg4z(Z + (int64_t)k * plane, sg, n, DIM, sc4));
        float4 kx = d_xout ? ldg4z(d_xout, sg, n, DIM, sc4) : f4zero();
        kx = f4add(kx, ldg4z(d_out, sg, n, DIM, sc4));         // + the head branch's d x_out
        if constexpr (PRE) {
            float4 gsum[4];
            gather_finish(gst, pb, sc4, gsum);                 // (rows past n: empty ranges, zeros)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (b >= pb.nblk) break;
                float4 v;
                if (pb.gsrc[b]) {                              // (workgroup-uniform) the plane is formed here, and kept
                    v = gsum[b];
                    if (sg < n) stg4(pb.dP_out + (int64_t)b * plane, sg, DIM, sc4, v);
                } else {
                    v = ldg4z(pb.dP + (int64_t)b * plane, sg, n, DIM, sc4);
                }
                edge::st_pieces4(PL + b * PT, sr, sc4, v);
            }
            st_lds4(EX, sr, sc4, ldg4z(pb.Zx1, sg, n, DIM, sc4));
            st_lds4(D1, sr, sc4, ldg4z(pb.dx1_direct, sg, n, DIM, sc4));
            kx = f4add(kx, ldg4z(pb.d_add, sg, n, DIM, sc4));
        }
        st_lds4(K, sr, sc4, kx);
    }
    __syncthreads();
    TPROBE(41);

    if constexpr (PRE) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int b = 0; b < pb.nblk; ++b) {
            acc += mma_strip_p(PL + b * PT, wf);
            edge::load_wfragb1<true, 1>(wf, b + 1 < pb.nblk ? pb.wp[b + 1] : pb.Wx1, 0, fr.wc);
        }
        st_f32x4(D0 + to, acc);
        __syncthreads();
        {   // dz_x1 = (d x1) * SiLU'(z_x1): as pieces over the first d P plane (read for the last time before the barrier above),
            // and parked in RED for the final coalesced store
            const float4 d1 = f4add(lds4(D0, sr, sc4), lds4(D1, sr, sc4));
            const float4 dzx = f4mul(d1, f4dsilu(lds4(EX, sr, sc4)));
            edge::st_pieces4(PL, sr, sc4, dzx);
            st_lds4(RED, sr, sc4, dzx);
        }
        __syncthreads();
        const f32x4 a2 = mma_strip_p(PL, wf);
        edge::load_wfragb1<true, 1>(wf, p.W[6], 0, fr.wc);
        st_f32x4(K + to, lds_f32x4(K + to) + a2);              // d x_out = head's d x + d_add + g_head
        __syncthreads();
    }
    TPROBE(42);
    char* const PA = PL + PT;
    char* const PB = PL + 2 * PT;
    {
        // dz6 = d r3 * SiLU'(z6), d r3 = d x_out (next layer + head branch) already in K
        const f32x4 z6 = lds_f32x4(ZL + 6 * SLOT + to), kv = lds_f32x4(K + to);
        f32x4 dz;
#pragma unroll
        for (int r = 0; r < 4; ++r) dz[r] = kv[r] * dsilu(z6[r]) * tmask;
        st_pieces_acc(PA, fr, dz);
        st_f32x4(ZL + 6 * SLOT + to, dz);
        __syncthreads();
    }

    // One backward step: v = dz_k * W_k (+ K) ; optionally K <- v, extra <- v ; then dz_{k-1} = v * SiLU'(z_{k-1}) -> planes dst.
    // k == 0 ends the chain: v = d x2 (left in D0).
    auto back = [&](const char* in, char* dst, int k, bool add_k, bool keep_k, float* extra) {
        TPROBE(4 * k);
        f32x4 zn = {0.f, 0.f, 0.f, 0.f};
        if (k > 0) zn = lds_f32x4(ZL + (k - 1) * SLOT + to);
        f32x4 v = mma_strip_p(in, wf);
        TPROBE(4 * k + 1);
        if (k > 0) edge::load_wfragb1<true, 1>(wf, p.W[k - 1], 0, fr.wc);
        if (add_k) v += lds_f32x4(K + to);
        if (keep_k) st_f32x4(K + to, v);
        if (extra) st_f32x4(extra + to, v);
        if (k == 0) {
            st_f32x4(D0 + to, v);
        } else {
            f32x4 dz;
#pragma unroll
            for (int r = 0; r < 4; ++r) dz[r] = v[r] * dsilu(zn[r]) * tmask;
            st_pieces_acc(dst, fr, dz);
            st_f32x4(ZL + (k - 1) * SLOT + to, dz);
        }
        TPROBE(4 * k + 2);
        __syncthreads();
        TPROBE(4 * k + 3);
    };
    back(PA, PB, 6, false, false, nullptr);        // d a5          -> dz5
    back(PB, PA, 5, true, true, nullptr);          // d r2 = . + d r3 (kept)      -> dz4
    back(PA, PB, 4, false, false, nullptr);        // d a3          -> dz3
    back(PB, PA, 3, true, true, EX);               // d r1 = . + d r2 (kept, = d res_x) -> dz2
    back(PA, PB, 2, false, false, nullptr);        // d a1          -> dz1
    back(PB, PA, 1, true, false, nullptr);         // d h0 = . + d r1             -> dz0
    back(PA, nullptr, 0, false, false, nullptr);   // d x2 -> D0

    if (sg < n) {
#pragma unroll
        for (int k = 0; k < 7; ++k) stg4(dZ + (int64_t)k * plane, sg, DIM, sc4, lds4(ZL + k * SLOT, sr, sc4));
        stg4(d_x2, sg, DIM, sc4, lds4(D0, sr, sc4));
        stg4(d_resx, sg, DIM, sc4, lds4(EX, sr, sc4));
        if constexpr (PRE) stg4(pb.dZx1, sg, DIM, sc4, lds4(RED, sr, sc4));
    }
    TPROBE(43);
}

// ---- the backward chain for batches of several rounds (see node_tail_fwd_lean_kernel) -----------------------------------
// The kernel above stages all seven z_k tiles, the residual gradient, d res_x and the head's operands in LDS (135 KB: one
// workgroup per CU).  Every element of those tiles is only ever touched by the lane that owns it in the accumulator layout
// (row 4 kg + r, column 16 wave + r16), so here they are REGISTERS: z_{k-1} is fetched by its lane ahead of layer k's MFMAs,
// dz_k leaves from the accumulators (behind the next weight request), the kept residual gradient and d res_x never leave
// the lane.  LDS holds the two GEMM-input tiles and, with PRE, the head's <= 4 dP planes: 51 KB, two workgroups of 8 waves
// per CU -- one's epilogue under the other's MFMAs.  Packed (transposed) weight images, deferred heads; the same arithmetic
// in the same order as node_tail_bwd_kernel<true, false, PRE>: bitwise the same results.  In-place calls (d_x2 / d_resx
// aliasing dx1_direct / d_add) stay safe: a lane reads its elements before it writes them.
template <bool PRE>
__global__ __launch_bounds__(TWG, 2) void node_tail_bwd_lean_kernel(const float* __restrict__ d_xout /* may be null */,
                                                                   const float* __restrict__ d_out, int64_t n, TailParams p,
                                                                   const float* __restrict__ Z, float* __restrict__ dZ,
                                                                   float* d_x2, float* d_resx, PreBwd pb) {
    __shared__ __attribute__((aligned(16))) float lds[(PRE ? 6 : 2) * SLOT];
    float* D0 = lds;
    float* D1 = lds + SLOT;
    const int64_t row0 = (int64_t)blockIdx.x * BMN;
    const int64_t plane = n * DIM;
    const Frag fr;
    // (operands swapped, mma_strip_t: a lane holds row r16, channels wc + 4 kg + 0..3 -- every access to a plane or an LDS tile is
    // ONE 16-byte request where the channel-per-lane form made four 4-byte ones; same sums, same bits)
    const int to = fr.r16 * LDT + fr.wc + 4 * fr.kg;
    const bool ok = row0 + fr.r16 < n;
    const float okm = ok ? 1.f : 0.f;     // (a factor, not a select around SiLU': see node_tail_bwd_kernel's tmask; rows past n read row 0's z: finite)
    const int64_t off = (ok ? row0 + fr.r16 : 0) * DIM + fr.wc + 4 * fr.kg;   // element offset of the lane's float4 in a plane
    auto ldp = [&](const float* base) __attribute__((always_inline)) { return lds_f32x4(base + off); };   // (a global float4)
    auto stp = [&](float* base, const f32x4& v) __attribute__((always_inline)) { st_f32x4(base + off, v); };
    WFrag1 wf;
    load_wfrag1_img(wf, PRE ? pb.wp[0] : p.W[6]);
    // z_k of a lane's four elements.  Requested TWO layers ahead (a layer is ~1 500 cycles, an HBM round trip ~2 000 under
    // load); the first three right here, so that they travel during the head's GEMMs / the first layer.
    auto ldz = [&](int k) __attribute__((always_inline)) { return ldp(Z + (int64_t)k * plane); };
    const f32x4 z6 = ldz(6), z5 = ldz(5), z4 = ldz(4);
    // d x_out = (next layer's d x) + the head branch's contribution (+ the head backward's d_add and d x below)
    f32x4 kreg;
    {
        f32x4 k = d_xout ? ldp(d_xout) : f32x4{0.f, 0.f, 0.f, 0.f};
        k += ldp(d_out);
        if constexpr (PRE) k += ldp(pb.d_add);
#pragma unroll
        for (int r = 0; r < 4; ++r) kreg[r] = ok ? k[r] : 0.f;
    }
    if constexpr (PRE) {
        float* PL = lds + 2 * SLOT;
        const f32x4 zx = ldp(pb.Zx1);
        f32x4 dxd = ldp(pb.dx1_direct);
#pragma unroll
        for (int r = 0; r < 4; ++r) dxd[r] = ok ? dxd[r] : 0.f;
        const int sc4 = threadIdx.x & 31, sr = threadIdx.x >> 5;
        for (int b = 0; b < pb.nblk; ++b)
            st_lds4(PL + b * SLOT, sr, sc4, ldg4z(pb.dP + (int64_t)b * plane, row0 + sr, n, DIM, sc4));
        __syncthreads();
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int b = 0; b < pb.nblk; ++b) {
            acc += mma_strip_t(PL + b * SLOT, wf);
            load_wfrag1_img(wf, b + 1 < pb.nblk ? pb.wp[b + 1] : pb.Wx1);
        }
        f32x4 dzx;
#pragma unroll
        for (int r = 0; r < 4; ++r) dzx[r] = (acc[r] + dxd[r]) * dsilu(zx[r]) * okm;     // dz_x1 = (d x1) * SiLU'(z_x1)
        st_f32x4(D1 + to, dzx);
        if (ok) stp(pb.dZx1, dzx);
        __syncthreads();
        const f32x4 a2 = mma_strip_t(D1, wf);
        load_wfrag1_img(wf, p.W[6]);
        kreg += a2;                                            // d x_out = d_add + g_head + the head's d x
        __syncthreads();                                       // (every wave is done reading D1)
    }
    {   // dz6 = d r3 * SiLU'(z6)
        f32x4 dz;
#pragma unroll
        for (int r = 0; r < 4; ++r) dz[r] = kreg[r] * dsilu(z6[r]) * okm;
        st_f32x4(D1 + to, dz);
        if (ok) stp(dZ + 6 * plane, dz);
        __syncthreads();
    }
    // One backward step: v = dz_k * W_k (+ kept); optionally kept <- v, d res_x <- v; then dz_{k-1} = v * SiLU'(z_{k-1}).
    // zn = z_{k-1} (requested two steps ago); *pre <- z_{k-3}, requested now
    auto back = [&](const float* in, float* dst, int k, bool add_k, bool keep_k, bool extra, const f32x4& zn, f32x4* pre) {
        if (pre) *pre = ldz(k - 3);
        f32x4 v = mma_strip_t(in, wf);
        if (k > 0) load_wfrag1_img(wf, p.W[k - 1]);
        if (add_k) v += kreg;
        if (keep_k) kreg = v;
        if (extra && ok) stp(d_resx, v);
        if (k == 0) {
            if (ok) stp(d_x2, v);
        } else {
            f32x4 dz;
#pragma unroll
            for (int r = 0; r < 4; ++r) dz[r] = v[r] * dsilu(zn[r]) * okm;
            st_f32x4(dst + to, dz);
            if (ok) stp(dZ + (int64_t)(k - 1) * plane, dz);
        }
        if (k > 0) __syncthreads();
    };
    f32x4 z3, z2, z1, z0;
    const f32x4 none = {0.f, 0.f, 0.f, 0.f};
    back(D1, D0, 6, false, false, false, z5, &z3);      // d a5          -> dz5
    back(D0, D1, 5, true, true, false, z4, &z2);        // d r2 = . + d r3 (kept)      -> dz4
    back(D1, D0, 4, false, false, false, z3, &z1);      // d a3          -> dz3
    back(D0, D1, 3, true, true, true, z2, &z0);         // d r1 = . + d r2 (kept, = d res_x) -> dz2
    back(D1, D0, 2, false, false, false, z1, nullptr);  // d a1          -> dz1
    back(D0, D1, 1, true, false, false, z0, nullptr);   // d h0 = . + d r1             -> dz0
    back(D1, D0, 0, false, false, false, none, nullptr);   // d x2
}

// Backward of the head branch of every layer in one launch (grid = (ceil(n/16), layers)): from d out / d att
//   head partials (d w_out, d w_att, d b_out), dz9, dz8, dz7 (-> dZ3[l][0..2] for the weight gradients) and the branch's
//   contribution to d x_out, g_head[l] = dz7 * W7.
struct HeadBwdLayer {
    const float* d_out;       // [n]
    const float* d_att;       // [n]
    const float* W[3];        // mlp_out matrices 7, 8, 9 (row-major, or transposed-orientation images when packed)
    const float* w_out;
    const float* w_att;
    const float* Z;           // the layer's [10][n][128] pre-activation block (slots 7..9 read)
    float* dZ3;               // [3][n][128]: dz7, dz8, dz9
    float* g_head;            // [n][128]
    float* head_partial;      // [ceil(n/16)][257]
};
struct HeadBwdBatch {
    HeadBwdLayer l[MAX_HEAD_LAYERS];
};

template <bool PACKED>
__global__ __launch_bounds__(TWG) void node_heads_bwd_kernel(HeadBwdBatch hb, int64_t n) {
    __shared__ __attribute__((aligned(16))) float lds[5 * SLOT + 16 * 256 + 16];
    float* D0 = lds;
    float* D1 = lds + SLOT;
    float* ZL = lds + 2 * SLOT;       // [3] z7, z8, z9, then dz7, dz8, dz9
    float* red = lds + 5 * SLOT;
    const HeadBwdLayer& hl = hb.l[blockIdx.y];
    const int64_t row0 = (int64_t)blockIdx.x * BMN;
    const int64_t plane = n * DIM;
    const Frag fr;
    const int c = fr.col();
    WFrag1 wf;
    if constexpr (PACKED) load_wfrag1_img(wf, hl.W[2]);
    else load_wfrag1<true>(wf, hl.W[2], fr.wc);
    const int sc4 = threadIdx.x & 31, sr = threadIdx.x >> 5;
    const int64_t sg = row0 + sr;
#pragma unroll
    for (int k = 0; k < 3; ++k) st_lds4(ZL + k * SLOT, sr, sc4, ldg4z(hl.Z + (int64_t)(7 + k) * plane, sg, n, DIM, sc4));
    __syncthreads();
    auto load_z = [&](int k) {
        f32x4 z;
#pragma unroll
        for (int r = 0; r < 4; ++r) z[r] = ZL[k * SLOT + fr.row(r) * LDT + c];
        return z;
    };
    {
        float4 so = f4zero(), sa = f4zero();
        float sb = 0.f;
        if (sg < n) {
            const float4 o3 = f4silu(lds4(ZL + 2 * SLOT, sr, sc4));
            const float go = hl.d_out[sg], ga = hl.d_att[sg];
            so = make_float4(go * o3.x, go * o3.y, go * o3.z, go * o3.w);
            sa = make_float4(ga * o3.x, ga * o3.y, ga * o3.z, ga * o3.w);
            sb = go;
        }
        float* mine = red + sr * 256;
        *reinterpret_cast<float4*>(mine + 4 * sc4) = so;
        *reinterpret_cast<float4*>(mine + 128 + 4 * sc4) = sa;
        if (sc4 == 0) red[16 * 256 + sr] = sb;
        __syncthreads();
        if (threadIdx.x < 256) {
            float tot = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) tot += red[q * 256 + threadIdx.x];
            hl.head_partial[(int64_t)blockIdx.x * 257 + threadIdx.x] = tot;
        } else if (threadIdx.x == 256) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += red[16 * 256 + q];
            hl.head_partial[(int64_t)blockIdx.x * 257 + 256] = t;
        }
    }
    {
        const f32x4 z9 = load_z(2);
        const float wo = hl.w_out[c], wa = hl.w_att[c];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rw = fr.row(r);
            const int64_t g = row0 + rw;
            const float dz = (g < n) ? (hl.d_out[g] * wo + hl.d_att[g] * wa) * dsilu(z9[r]) : 0.f;
            D0[rw * LDT + c] = dz;
            ZL[2 * SLOT + rw * LDT + c] = dz;
        }
        __syncthreads();
    }
    // v = dz_k * W_k; k > 7: dz_{k-1} = v * SiLU'(z_{k-1}); k == 7: v = the branch's d x_out
    const int to = fr.r16 * LDT + fr.wc + 4 * fr.kg;          // (operands swapped: row r16, channels wc + 4 kg + 0..3; mma_strip_t)
    const bool trow = row0 + fr.r16 < n;
    // (a factor, not a select: with `trow ? x : 0` the compiler runs the four SiLU' sequences of a lane one after the other under
    // exec masks -- four dependent exp / rcp chains in a row -- instead of interleaved; the rows past n are zero either way)
    const float tmask = trow ? 1.f : 0.f;
    auto back = [&](const float* in, float* dst, int k) {
        f32x4 zn = {0.f, 0.f, 0.f, 0.f};
        if (k > 7) zn = lds_f32x4(ZL + (k - 8) * SLOT + to);
        const f32x4 acc = mma_strip_t(in, wf);
        if (k > 7) {
            if constexpr (PACKED) load_wfrag1_img(wf, hl.W[k - 8]);
            else load_wfrag1<true>(wf, hl.W[k - 8], fr.wc);
        }
        if (k == 7) {
            st_f32x4(dst + to, acc);
        } else {
            f32x4 dz;
#pragma unroll
            for (int r = 0; r < 4; ++r) dz[r] = acc[r] * dsilu(zn[r]) * tmask;
            st_f32x4(dst + to, dz);
            st_f32x4(ZL + (k - 8) * SLOT + to, dz);
        }
        __syncthreads();
    };
    back(D0, D1, 9);          // -> dz8
    back(D1, D0, 8);          // -> dz7
    back(D0, D1, 7);          // -> g_head
    if (sg < n) {
#pragma unroll
        for (int k = 0; k < 3; ++k) stg4_nt(hl.dZ3 + (int64_t)k * plane, sg, DIM, sc4, lds4(ZL + k * SLOT, sr, sc4));   // (read by the layers' weight-gradient launches, much later)
        stg4(hl.g_head, sg, DIM, sc4, lds4(D1, sr, sc4));
    }
}

// columns: [0,128) d w_out, [128,256) d w_att, [256] d b_out
// one workgroup per 16 columns: 16 row-slices x 16 columns, fixed-order tree in LDS (deterministic)
__global__ __launch_bounds__(WG) void head_reduce_kernel(const float* __restrict__ partial, int nblocks,
                                                         float* __restrict__ d_wout, float* __restrict__ d_watt,
                                                         float* __restrict__ d_bout) {
    __shared__ float red[16][17];
    const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    float s = 0.f;
    if (c < 257)
        for (int b = sl; b < nblocks; b += 16) s += partial[(int64_t)b * 257 + c];
    red[sl][cl] = s;
    __syncthreads();
    if (sl == 0 && c < 257) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += red[q][cl];
        if (c < 128) d_wout[c] = t;
        else if (c < 256) d_watt[c - 128] = t;
        else d_bout[0] = t;
    }
}

inline TailParams make_tail(const float* const* weights, const float* const* biases, const float* w_out,
                            const float* b_out, const float* w_att, int packed) {
    TailParams p;
    p.packed = packed;
    for (int k = 0; k < 10; ++k) {
        p.W[k] = weights[k];
        p.b[k] = biases ? biases[k] : nullptr;
    }
    p.w_out = w_out;
    p.b_out = b_out;
    p.w_att = w_att;
    return p;
}

struct PackJobs {
    const float* src[192];
    int ld[192];
};
// grid (njobs, 8): block (job, j) writes the 512 float4 fragments of 16-column tile j
__global__ __launch_bounds__(512) void pack_weights_kernel(PackJobs jobs, int transposed, float* __restrict__ images) {
    const float* __restrict__ W = jobs.src[blockIdx.x];
    const int ld = jobs.ld[blockIdx.x];
    const int j = blockIdx.y, q = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = 16 * j + (lane & 15), k0 = 16 * q + 4 * (lane >> 4);
    float4 v;
    if (!transposed) v = *reinterpret_cast<const float4*>(W + (size_t)c * ld + k0);
    else v = make_float4(W[(size_t)k0 * ld + c], W[(size_t)(k0 + 1) * ld + c], W[(size_t)(k0 + 2) * ld + c],
                         W[(size_t)(k0 + 3) * ld + c]);
    reinterpret_cast<float4*>(images + (size_t)blockIdx.x * IMG)[(j * 8 + q) * 64 + lane] = v;
}

}  // namespace

// Fragment-ordered images of n (<= 192) 128x128 matrices (row stride ld[i] floats) into images[i * 16384 ...]; see the
// layout note above.  transposed = 0: images for Y = X W^T (forward), 1: for Y = X W (backward).
extern "C" int pamnet_pack_weights_f32(int64_t n, const float* const* W, const int64_t* ld, int32_t transposed,
                                       float* images, pamnet_stream_t stream) {
    if (n < 0 || n > 192) return PAMNET_EINVAL;
    if (n == 0) return PAMNET_OK;
    if (!W || !ld || !images) return PAMNET_ENULL;
    PackJobs jobs;
    for (int i = 0; i < 192; ++i) {
        const int s = i < n ? i : 0;
        if (!W[s]) return PAMNET_ENULL;
        if (ld[s] < DIM || (ld[s] & 3)) return PAMNET_EINVAL;
        jobs.src[i] = W[s];
        jobs.ld[i] = (int)ld[s];
    }
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)n, 8), dim3(512), 0, as_stream(stream), jobs, (int)transposed,
                       images);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

// One launch for the images of a whole step direction (round 6): kind 0 = the chains' fp32 fragment image above (16 384 floats),
// kind 1 = the edge-level kernels' bf16x3 fragment image (edge_core.h load_wfragb1: 24 576 floats, the pieces a wave would
// have made of its slice itself); offset[i] = where image i starts in `images`, in floats (multiples of 4).
constexpr int PACK_MIXED_MAX = 224;
namespace {
struct PackJobsX {
    const float* src[PACK_MIXED_MAX];
    int ld[PACK_MIXED_MAX];                   // < 0: kind 1 (of row stride -ld)
    uint32_t off4[PACK_MIXED_MAX];            // image offset in float4 units
};
__global__ __launch_bounds__(512) void pack_weights_mixed_kernel(PackJobsX jobs, int transposed, float* __restrict__ images) {
    const float* __restrict__ W = jobs.src[blockIdx.x];
    const int ld = jobs.ld[blockIdx.x];
    float4* __restrict__ out = reinterpret_cast<float4*>(images) + jobs.off4[blockIdx.x];
    const int j = blockIdx.y, q = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (ld > 0) {
        const int c = 16 * j + (lane & 15), k0 = 16 * q + 4 * (lane >> 4);
        float4 v;
        if (!transposed) v = *reinterpret_cast<const float4*>(W + (size_t)c * ld + k0);
        else v = make_float4(W[(size_t)k0 * ld + c], W[(size_t)(k0 + 1) * ld + c], W[(size_t)(k0 + 2) * ld + c],
                             W[(size_t)(k0 + 3) * ld + c]);
        out[(j * 8 + q) * 64 + lane] = v;
    } else if (q < DIM / 32) {
        const Frag3 f = transposed ? edge::wfragb1_q<true>(W, -ld, 16 * j, q, lane) : edge::wfragb1_q<false>(W, -ld, 16 * j, q, lane);
        uint4* o = reinterpret_cast<uint4*>(out) + ((j * (DIM / 32) + q) * 3) * 64 + lane;
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) o[pc * 64] = make_uint4(f.p[pc][0], f.p[pc][1], f.p[pc][2], f.p[pc][3]);
    }
}
}  // namespace

extern "C" int pamnet_pack_weights_mixed_f32(int64_t n, const float* const* W, const int64_t* ld, const int32_t* kind,
                                             const int64_t* offset, int32_t transposed, float* images, pamnet_stream_t stream) {
    if (n < 0 || n > PACK_MIXED_MAX) return PAMNET_EINVAL;
    if (n == 0) return PAMNET_OK;
    if (!W || !ld || !kind || !offset || !images) return PAMNET_ENULL;
    PackJobsX jobs;
    for (int i = 0; i < PACK_MIXED_MAX; ++i) {
        const int s = i < n ? i : 0;
        if (!W[s]) return PAMNET_ENULL;
        if (ld[s] < DIM || (ld[s] & 3) || ld[s] > 0x3fffffff || (kind[s] != 0 && kind[s] != 1)) return PAMNET_EINVAL;
        if (offset[s] < 0 || (offset[s] & 3) || (offset[s] >> 2) > 0xffffffffll) return PAMNET_EINVAL;
        jobs.src[i] = W[s];
        jobs.ld[i] = kind[s] ? -(int)ld[s] : (int)ld[s];
        jobs.off4[i] = (uint32_t)(offset[s] >> 2);
    }
    hipLaunchKernelGGL(pack_weights_mixed_kernel, dim3((unsigned)n, 8), dim3(512), 0, as_stream(stream), jobs, (int)transposed,
                       images);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

// bf16x3 fragment images of n (<= 192) matrices into images[i * 24 576 ...]: kind-1 images of the launch above, one after the
// other (node_tail_fwd_bf16_kernel and the edge-level kernels read the same layout).
constexpr int64_t IMGB_FLOATS = 3 * DIM * DIM / 2;
extern "C" int pamnet_pack_weights_bf16x3(int64_t n, const float* const* W, const int64_t* ld, int32_t transposed,
                                          float* images, pamnet_stream_t stream) {
    if (n < 0 || n > 192) return PAMNET_EINVAL;
    if (n == 0) return PAMNET_OK;
    if (!W || !ld || !images) return PAMNET_ENULL;
    int32_t kind[192];
    int64_t offset[192];
    for (int i = 0; i < n; ++i) kind[i] = 1, offset[i] = i * IMGB_FLOATS;
    return pamnet_pack_weights_mixed_f32(n, W, ld, kind, offset, transposed, images, stream);
}

// PAMNET_CHAIN_WAVES=4|8: the bf16x6 forward chain's workgroup geometry (read once, for A/B timing)
static int chain_waves() {
    static const int v = [] { const char* e = getenv("PAMNET_CHAIN_WAVES"); return (e && atoi(e) == 4) ? 4 : 8; }();
    return v;
}

static int tail_fwd_launch(const float* x2, const float* res_x, int64_t n, const float* const* weights,
                           const float* const* biases, const float* w_out, const float* b_out, const float* w_att, float* Z,
                           float* R, float* x_out, float* out, float* att, const float* next_Wx1, const float* next_bx1,
                           const float* const* next_wp, int64_t next_ldwp, int64_t next_nblk, float* next_Zx1,
                           float* next_x1, float* next_P, int32_t packed, const Mlp2Rider* rider, int rider_wgs,
                           pamnet_stream_t stream, const pamnet_local_agg* agg = nullptr) {
    if (n < 0 || next_nblk < 0 || next_nblk > 4) return PAMNET_EINVAL;
    if (n == 0) return PAMNET_OK;
    if (!x2 || !res_x || !weights || !biases || !w_out || !b_out || !w_att || (Z && !R) || !x_out || (!out != !att))
        return PAMNET_ENULL;
    const bool heads = out != nullptr;        // out = att = null: the head branch is run later by pamnet_node_heads_fwd_f32
    for (int k = 0; k < 10; ++k)
        if (!weights[k] || !biases[k]) return PAMNET_ENULL;
    PreNext nx{};
    nx.nblk = (int)next_nblk;
    if (next_nblk > 0) {
        if (!next_Wx1 || !next_bx1 || !next_wp || !next_x1 || !next_P) return PAMNET_ENULL;   // next_Zx1: optional save
        nx.Wx1 = next_Wx1, nx.bx1 = next_bx1, nx.ldwp = (int)next_ldwp;
        nx.Zx1 = next_Zx1, nx.x1 = next_x1, nx.P = next_P;
        for (int b = 0; b < next_nblk; ++b) {
            if (!next_wp[b]) return PAMNET_ENULL;
            nx.wp[b] = next_wp[b];
        }
    }
    hipStream_t st = as_stream(stream);
    const dim3 grid((unsigned)ceil_div(n, BMN));
    const TailParams tp = make_tail(weights, biases, w_out, b_out, w_att, packed ? 1 : 0);
    if (packed == 2 && heads) return PAMNET_EINVAL;            // bf16x3 images: deferred heads only
    LocalAgg la{};
    if (agg) {
        if (!agg->t_ptr || !agg->l_ptr) return PAMNET_ENULL;
        // the fp32 chain kernels (packed images, deferred heads) form x2 themselves; every other form reads it: the aggregation
        // runs as its own launch first, as it did until round 6
        const bool in_kernel = packed && !heads && (packed == 2 || !(grid.x > LEAN_FROM_TILES && lean_mode() >= 1 && !rider));
        if (in_kernel) {
            la.m_ji = (const float4*)agg->m_ji, la.m_nb = (const float4*)agg->m_nb, la.s = (const float4*)agg->s;
            la.q3 = (const float4*)agg->q3, la.init = (const float4*)agg->init, la.t_ptr = agg->t_ptr, la.t_col = agg->t_col;
            la.l_ptr = agg->l_ptr, la.m_t = (float4*)agg->m_t, la.x2_out = (float4*)const_cast<float*>(x2);
            if (!la.m_ji) la = LocalAgg{};                     // (a batch without local edges: x2 = init, by the launch below)
        }
        if (!la.m_ji) {
            const int rc = pamnet_local_agg_fwd_f32(agg->m_ji, agg->m_nb, agg->s, agg->q3, agg->t_ptr, agg->t_col, agg->l_ptr,
                                                    agg->init, n, agg->m_t, const_cast<float*>(x2), stream);
            if (rc) return rc;
        }
    }
    if (rider) {
        if (!packed || heads) return PAMNET_EINVAL;
        Mlp2Rider rd = *rider;
        rd.n_chain = (int)grid.x;
        if (packed == 2)
            if (chain_waves() == 8)
                hipLaunchKernelGGL((node_tail_fwd_bf16_kernel<true, 8>), dim3(grid.x + (unsigned)rider_wgs), dim3(TWG), 0, st, x2,
                                   res_x, n, tp, Z, R, x_out, nx, rd, la);
            else
                hipLaunchKernelGGL((node_tail_fwd_bf16_kernel<true, 4>), dim3(grid.x + (unsigned)rider_wgs), dim3(WG), 0, st, x2,
                                   res_x, n, tp, Z, R, x_out, nx, rd, la);
        else
            hipLaunchKernelGGL((node_tail_fwd_kernel<true, false, true>), dim3(grid.x + (unsigned)rider_wgs), dim3(WG), 0, st,
                               x2, res_x, n, tp, Z, R, x_out, out, att, nx, rd, la);
    } else if (packed == 2 && chain_waves() == 8) hipLaunchKernelGGL((node_tail_fwd_bf16_kernel<false, 8>), grid, dim3(TWG), 0, st, x2, res_x, n, tp, Z, R, x_out, nx, Mlp2Rider{}, la);
    else if (packed == 2) hipLaunchKernelGGL((node_tail_fwd_bf16_kernel<false, 4>), grid, dim3(WG), 0, st, x2, res_x, n, tp, Z, R, x_out, nx, Mlp2Rider{}, la);
    else if (packed && heads) hipLaunchKernelGGL((node_tail_fwd_kernel<true, true>), grid, dim3(WG), 0, st, x2, res_x, n, tp, Z, R, x_out, out, att, nx);
    else if (packed && grid.x > LEAN_FROM_TILES && lean_mode() >= 1) hipLaunchKernelGGL(node_tail_fwd_lean_kernel, grid, dim3(WG), 0, st, x2, res_x, n, tp, Z, R, x_out, nx);
    else if (packed) hipLaunchKernelGGL((node_tail_fwd_kernel<true, false>), grid, dim3(WG), 0, st, x2, res_x, n, tp, Z, R, x_out, out, att, nx, Mlp2Rider{}, la);
    else if (heads) hipLaunchKernelGGL((node_tail_fwd_kernel<false, true>), grid, dim3(WG), 0, st, x2, res_x, n, tp, Z, R, x_out, out, att, nx);
    else hipLaunchKernelGGL((node_tail_fwd_kernel<false, false>), grid, dim3(WG), 0, st, x2, res_x, n, tp, Z, R, x_out, out, att, nx);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_node_tail_fwd_f32(const float* x2, const float* res_x, int64_t n, const float* const* weights,
                                        const float* const* biases, const float* w_out, const float* b_out,
                                        const float* w_att, float* Z, float* R, float* x_out, float* out, float* att,
                                        const float* next_Wx1, const float* next_bx1, const float* const* next_wp,
                                        int64_t next_ldwp, int64_t next_nblk, float* next_Zx1, float* next_x1,
                                        float* next_P, int32_t packed, pamnet_stream_t stream) {
    return tail_fwd_launch(x2, res_x, n, weights, biases, w_out, b_out, w_att, Z, R, x_out, out, att, next_Wx1, next_bx1,
                           next_wp, next_ldwp, next_nblk, next_Zx1, next_x1, next_P, packed, nullptr, 0, stream);
}

// The same launch (packed weight images, deferred heads: out = att = null) with a rider: row tiles
// [mlp_tile0, mlp_tile0 + mlp_ntiles) (16 rows each) of the two-layer MLP  y = SiLU(W2 SiLU(W1 x + b1) + b2)  over
// mlp_x [mlp_rows, 128] are computed by `rider_wgs` extra workgroups (mlp = {W1, b1, W2, b2}; mlp_out = {z1, z2, y},
// z1 / z2 nullable saves) -- pamnet_mlp2_fwd_f32 on the CUs the chain leaves idle.
extern "C" int pamnet_node_tail_fwd_rider_f32(const float* x2, const float* res_x, int64_t n, const float* const* weights,
                                              const float* const* biases, const float* w_out, const float* b_out,
                                              const float* w_att, float* Z, float* R, float* x_out,
                                              const float* next_Wx1, const float* next_bx1, const float* const* next_wp,
                                              int64_t next_ldwp, int64_t next_nblk, float* next_Zx1, float* next_x1,
                                              float* next_P, const float* mlp_x, int64_t mlp_rows, int64_t mlp_tile0,
                                              int64_t mlp_ntiles, const float* const* mlp, float* const* mlp_out,
                                              int64_t rider_wgs, int32_t packed, pamnet_stream_t stream) {
    if (mlp_rows < 0 || mlp_tile0 < 0 || mlp_ntiles < 0 || rider_wgs < 0 || (packed != 1 && packed != 2)) return PAMNET_EINVAL;
    if (mlp_ntiles == 0 || rider_wgs == 0)
        return tail_fwd_launch(x2, res_x, n, weights, biases, w_out, b_out, w_att, Z, R, x_out, nullptr, nullptr, next_Wx1,
                               next_bx1, next_wp, next_ldwp, next_nblk, next_Zx1, next_x1, next_P, packed, nullptr, 0, stream);
    if (!mlp_x || !mlp || !mlp_out || !mlp[0] || !mlp[1] || !mlp[2] || !mlp[3] || !mlp_out[2]) return PAMNET_ENULL;
    if (n == 0) return PAMNET_EINVAL;                        // no chain to ride on
    Mlp2Rider rd{};
    rd.x = mlp_x, rd.m = mlp_rows;
    rd.set = edge::Mlp2Set{mlp[0], mlp[1], mlp[2], mlp[3], mlp_out[0], mlp_out[1], mlp_out[2]};
    rd.tile0 = (int)mlp_tile0, rd.ntiles = (int)mlp_ntiles;
    if (rider_wgs > mlp_ntiles) rider_wgs = mlp_ntiles;       // never more workgroups than tiles
    return tail_fwd_launch(x2, res_x, n, weights, biases, w_out, b_out, w_att, Z, R, x_out, nullptr, nullptr, next_Wx1,
                           next_bx1, next_wp, next_ldwp, next_nblk, next_Zx1, next_x1, next_P, packed, &rd, (int)rider_wgs, stream);
}

// pamnet_node_tail_fwd_rider_f32 (mlp_ntiles = 0: without riders) whose chain input x2 is FORMED by the launch -- `agg` holds the
// operands of pamnet_local_agg_fwd_f32 (out = x2, which is also written); deferred heads, packed images.
extern "C" int pamnet_node_tail_fwd_agg_f32(float* x2, const float* res_x, int64_t n, const float* const* weights,
                                            const float* const* biases, const float* w_out, const float* b_out,
                                            const float* w_att, float* Z, float* R, float* x_out, const float* next_Wx1,
                                            const float* next_bx1, const float* const* next_wp, int64_t next_ldwp,
                                            int64_t next_nblk, float* next_Zx1, float* next_x1, float* next_P,
                                            const float* mlp_x, int64_t mlp_rows, int64_t mlp_tile0, int64_t mlp_ntiles,
                                            const float* const* mlp, float* const* mlp_out, int64_t rider_wgs, int32_t packed,
                                            const pamnet_local_agg* agg, pamnet_stream_t stream) {
    if (!agg) return PAMNET_ENULL;
    if (mlp_rows < 0 || mlp_tile0 < 0 || mlp_ntiles < 0 || rider_wgs < 0 || (packed != 1 && packed != 2)) return PAMNET_EINVAL;
    if (mlp_ntiles == 0 || rider_wgs == 0)
        return tail_fwd_launch(x2, res_x, n, weights, biases, w_out, b_out, w_att, Z, R, x_out, nullptr, nullptr, next_Wx1,
                               next_bx1, next_wp, next_ldwp, next_nblk, next_Zx1, next_x1, next_P, packed, nullptr, 0, stream, agg);
    if (!mlp_x || !mlp || !mlp_out || !mlp[0] || !mlp[1] || !mlp[2] || !mlp[3] || !mlp_out[2]) return PAMNET_ENULL;
    if (n == 0) return PAMNET_EINVAL;
    Mlp2Rider rd{};
    rd.x = mlp_x, rd.m = mlp_rows;
    rd.set = edge::Mlp2Set{mlp[0], mlp[1], mlp[2], mlp[3], mlp_out[0], mlp_out[1], mlp_out[2]};
    rd.tile0 = (int)mlp_tile0, rd.ntiles = (int)mlp_ntiles;
    if (rider_wgs > mlp_ntiles) rider_wgs = mlp_ntiles;
    return tail_fwd_launch(x2, res_x, n, weights, biases, w_out, b_out, w_att, Z, R, x_out, nullptr, nullptr, next_Wx1,
                           next_bx1, next_wp, next_ldwp, next_nblk, next_Zx1, next_x1, next_P, packed, &rd, (int)rider_wgs, stream,
                           agg);
}

extern "C" int pamnet_node_heads_fwd_f32(int64_t n_layers, const float* const* x_out, const float* const* weights,
                                         const float* const* biases, const float* const* w_out,
                                         const float* const* b_out, const float* const* w_att, float* const* Z,
                                         float* const* out, float* const* att, int64_t n, int32_t packed,
                                         pamnet_stream_t stream) {
    if (n < 0 || n_layers < 0) return PAMNET_EINVAL;
    if (n == 0 || n_layers == 0) return PAMNET_OK;
    if (!x_out || !weights || !biases || !w_out || !b_out || !w_att || !out || !att) return PAMNET_ENULL;
    hipStream_t st = as_stream(stream);
    for (int64_t l0 = 0; l0 < n_layers; l0 += MAX_HEAD_LAYERS) {
        const int nl = (int)(n_layers - l0 < MAX_HEAD_LAYERS ? n_layers - l0 : MAX_HEAD_LAYERS);
        HeadBatch hb{};
        for (int i = 0; i < nl; ++i) {
            const int64_t l = l0 + i;
            HeadLayer& h = hb.l[i];
            h.x_out = x_out[l];
            for (int k = 0; k < 3; ++k) h.W[k] = weights[3 * l + k], h.b[k] = biases[3 * l + k];
            h.w_out = w_out[l], h.b_out = b_out[l], h.w_att = w_att[l];
            h.Z = Z ? Z[l] : nullptr;
            h.out = out[l], h.att = att[l];
            if (!h.x_out || !h.W[0] || !h.W[1] || !h.W[2] || !h.b[0] || !h.b[1] || !h.b[2] || !h.w_out || !h.b_out ||
                !h.w_att || !h.out || !h.att)
                return PAMNET_ENULL;
        }
        // one 16-row tile per workgroup (32- and 64-row tilings were measured and lost at every batch size tried)
        const dim3 grid((unsigned)ceil_div(n, BMN), (unsigned)nl);
        if (packed) hipLaunchKernelGGL((node_heads_fwd_kernel<true, 1>), grid, dim3(WG), 0, st, hb, n);
        else hipLaunchKernelGGL((node_heads_fwd_kernel<false, 1>), grid, dim3(WG), 0, st, hb, n);
        PAMNET_LAUNCH_CHECK();
    }
    return PAMNET_OK;
}

extern "C" int pamnet_node_tail_bwd_f32(const float* d_xout, const float* d_out, const float* d_att, int64_t n,
                                        const float* const* weights, const float* w_out, const float* w_att,
                                        const float* Z, float* dZ, float* d_x2, float* d_resx, float* head_partial,
                                        float* d_wout, float* d_watt, float* d_bout, int32_t packed,
                                        pamnet_stream_t stream) {
    if (n < 0) return PAMNET_EINVAL;
    if (n == 0) return PAMNET_OK;
    if (!d_out || !d_att || !weights || !w_out || !w_att || !Z || !dZ || !d_x2 || !d_resx || !head_partial)
        return PAMNET_ENULL;
    hipStream_t st = as_stream(stream);
    const unsigned grid = (unsigned)ceil_div(n, BMN);
    if (packed)
        hipLaunchKernelGGL((node_tail_bwd_kernel<true, true>), dim3(grid), dim3(TWG), 0, st, d_xout, d_out, d_att, n,
                           make_tail(weights, nullptr, w_out, nullptr, w_att, 1), Z, dZ, d_x2, d_resx, head_partial);
    else
        hipLaunchKernelGGL((node_tail_bwd_kernel<false, true>), dim3(grid), dim3(TWG), 0, st, d_xout, d_out, d_att, n,
                           make_tail(weights, nullptr, w_out, nullptr, w_att, 0), Z, dZ, d_x2, d_resx, head_partial);
    PAMNET_LAUNCH_CHECK();
    if (d_wout || d_watt || d_bout) {      // all null: the caller reduces head_partial itself (pamnet_wgrad_batched_f32)
        if (!d_wout || !d_watt || !d_bout) return PAMNET_ENULL;
        hipLaunchKernelGGL(head_reduce_kernel, dim3(17), dim3(WG), 0, st, head_partial, (int)grid, d_wout, d_watt,
                           d_bout);
        PAMNET_LAUNCH_CHECK();
    }
    return PAMNET_OK;
}

extern "C" int pamnet_node_tail_main_bwd_f32(const float* d_xout, const float* g_head, int64_t n,
                                             const float* const* weights, const float* Z, float* dZ, float* d_x2,
                                             float* d_resx, int32_t packed, pamnet_stream_t stream) {
    if (n < 0) return PAMNET_EINVAL;
    if (n == 0) return PAMNET_OK;
    if (!g_head || !weights || !Z || !dZ || !d_x2 || !d_resx) return PAMNET_ENULL;
    for (int k = 0; k < 7; ++k)
        if (!weights[k]) return PAMNET_ENULL;
    hipStream_t st = as_stream(stream);
    const unsigned grid = (unsigned)ceil_div(n, BMN);
    TailParams tp{};
    for (int k = 0; k < 7; ++k) tp.W[k] = weights[k];
    tp.packed = packed ? 1 : 0;
    if (packed == 2)                                          // bf16x3 (kind-1, transposed) images: the bf16x6 chain
        hipLaunchKernelGGL((node_tail_bwd_bf16_kernel<false>), dim3(grid), dim3(TWG), 0, st, d_xout, g_head, n, tp, Z, dZ, d_x2,
                           d_resx);
    else if (packed && grid > LEAN_FROM_TILES && lean_mode() >= 2)
        hipLaunchKernelGGL((node_tail_bwd_lean_kernel<false>), dim3(grid), dim3(TWG), 0, st, d_xout, g_head, n, tp, Z, dZ, d_x2,
                           d_resx, PreBwd{});
    else if (packed)
        hipLaunchKernelGGL((node_tail_bwd_kernel<true, false>), dim3(grid), dim3(TWG), 0, st, d_xout, g_head,
                           (const float*)nullptr, n, tp, Z, dZ, d_x2, d_resx, (float*)nullptr);
    else
        hipLaunchKernelGGL((node_tail_bwd_kernel<false, false>), dim3(grid), dim3(TWG), 0, st, d_xout, g_head,
                           (const float*)nullptr, n, tp, Z, dZ, d_x2, d_resx, (float*)nullptr);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

/* tail_main_bwd preceded, on the same row tiles, by the backward of the next layer's head (pamnet_node_pre_bwd_f32 with
 * its d x feeding this chain's d x_out; d_xout of the chain is that d x, so there is no d_xout argument).  Packed
 * (transposed-orientation) weight images only. */
static int node_pre_tail_bwd_impl(float* dP, const float* const* gsrc, const int32_t* const* gptr,
                                  const int32_t* const* gperm, const float* dx1_direct, const float* d_add, int64_t n,
                                  const float* Wx1, const float* const* wp, int64_t nblk, const float* Zx1,
                                  float* dZx1, const float* g_head, const float* const* weights,
                                  const float* Z, float* dZ, float* d_x2, float* d_resx,
                                  const void* rider, pamnet_stream_t stream) {
    const bool pieces = nblk >= 0 && (nblk & PAMNET_CHAIN_PIECES) != 0;      // every image a bf16x3 (kind-1) one: the bf16x6 chain
    if (pieces) nblk &= ~(int64_t)PAMNET_CHAIN_PIECES;
    if (n < 0 || nblk < 1 || nblk > 4) return PAMNET_EINVAL;
    if (n == 0) return PAMNET_OK;
    if (!dP || !dx1_direct || !d_add || !Wx1 || !wp || !Zx1 || !dZx1 || !g_head || !weights || !Z || !dZ || !d_x2 || !d_resx)
        return PAMNET_ENULL;
    PreBwd pb{};
    pb.dP = dP, pb.Wx1 = Wx1, pb.Zx1 = Zx1, pb.dx1_direct = dx1_direct, pb.d_add = d_add, pb.dZx1 = dZx1;
    pb.nblk = (int)nblk;
    pb.dP_out = dP;
    bool any_gather = false;
    for (int b = 0; b < nblk; ++b) {
        if (!wp[b]) return PAMNET_ENULL;
        pb.wp[b] = wp[b];
        if (gsrc && gsrc[b]) {
            if (!gptr || !gptr[b]) return PAMNET_ENULL;
            pb.gsrc[b] = gsrc[b], pb.gptr[b] = gptr[b], pb.gperm[b] = gperm ? gperm[b] : nullptr;
            any_gather = true;
        }
    }
    TailParams tp{};
    for (int k = 0; k < 7; ++k) {
        if (!weights[k]) return PAMNET_ENULL;
        tp.W[k] = weights[k];
    }
    tp.packed = 1;
    const int n_tiles = (int)ceil_div(n, BMN);
    WBatchS rb{};
    float* rpart = nullptr;
    int rslots = 0;
    if (rider) {                                              // planned by pamnet_wgrad_rider_plan_f32
        const WgradRider* r = static_cast<const WgradRider*>(rider);
        rb = r->batch, rpart = r->partial, rslots = r->slots;
    }
    const bool lean = !pieces && rslots == 0 && (unsigned)n_tiles > LEAN_FROM_TILES && lean_mode() >= 2;
    if (lean && any_gather) {
        // the lean form reads its planes: form them with the batched segment-sum launch first (what the engine did until round 6)
        float* so[4];
        const float* sa[4];
        const int32_t *sp[4], *sr[4];
        int64_t nj = 0;
        for (int b = 0; b < nblk; ++b)
            if (pb.gsrc[b]) so[nj] = dP + (int64_t)b * n * DIM, sa[nj] = pb.gsrc[b], sp[nj] = pb.gperm[b], sr[nj] = pb.gptr[b], ++nj;
        const int rc = pamnet_segment_sum_multi_f32(nj, so, sa, sp, sr, n, DIM, stream);
        if (rc) return rc;
        for (int b = 0; b < 4; ++b) pb.gsrc[b] = nullptr;
    }
    if (pieces)
        hipLaunchKernelGGL((node_tail_bwd_bf16_kernel<true>), dim3((unsigned)(n_tiles + rslots)), dim3(TWG), 0, as_stream(stream),
                           (const float*)nullptr, g_head, n, tp, Z, dZ, d_x2, d_resx, pb, rb, rpart, n_tiles);
    else if (lean)
        hipLaunchKernelGGL((node_tail_bwd_lean_kernel<true>), dim3((unsigned)n_tiles), dim3(TWG), 0, as_stream(stream),
                           (const float*)nullptr, g_head, n, tp, Z, dZ, d_x2, d_resx, pb);
    else
        hipLaunchKernelGGL((node_tail_bwd_kernel<true, false, true>), dim3((unsigned)(n_tiles + rslots)), dim3(TWG), 0,
                           as_stream(stream), (const float*)nullptr, g_head, (const float*)nullptr, n, tp, Z, dZ, d_x2, d_resx,
                           (float*)nullptr, pb, rb, rpart, n_tiles);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_node_pre_tail_bwd_f32(const float* dP, const float* dx1_direct, const float* d_add, int64_t n,
                                            const float* Wx1, const float* const* wp, int64_t nblk, const float* Zx1,
                                            float* dZx1, const float* g_head, const float* const* weights,
                                            const float* Z, float* dZ, float* d_x2, float* d_resx,
                                            const void* rider, pamnet_stream_t stream) {
    return node_pre_tail_bwd_impl(const_cast<float*>(dP), nullptr, nullptr, nullptr, dx1_direct, d_add, n, Wx1, wp, nblk, Zx1, dZx1,
                                  g_head, weights, Z, dZ, d_x2, d_resx, rider, stream);
}
/* The same with planes of dP formed inside the launch: gather_src[b] != null -> plane b (written to dP + b n 128 as well) is the
 * segment sum of gather_src[b] over (gather_ptr[b], gather_perm[b] nullable) -- the launches of pamnet_segment_sum(_multi)_f32
 * that used to run ahead of this one. */
extern "C" int pamnet_node_pre_tail_bwd_gather_f32(float* dP, const float* const* gather_src,
                                                   const int32_t* const* gather_ptr, const int32_t* const* gather_perm,
                                                   const float* dx1_direct, const float* d_add, int64_t n, const float* Wx1,
                                                   const float* const* wp, int64_t nblk, const float* Zx1, float* dZx1,
                                                   const float* g_head, const float* const* weights, const float* Z,
                                                   float* dZ, float* d_x2, float* d_resx, const void* rider,
                                                   pamnet_stream_t stream) {
    if (!gather_src || !gather_ptr) return PAMNET_ENULL;
    return node_pre_tail_bwd_impl(dP, gather_src, gather_ptr, gather_perm, dx1_direct, d_add, n, Wx1, wp, nblk, Zx1, dZx1, g_head,
                                  weights, Z, dZ, d_x2, d_resx, rider, stream);
}

extern "C" int pamnet_node_heads_bwd_f32(int64_t n_layers, const float* const* d_out, const float* const* d_att,
                                         const float* const* weights, const float* const* w_out,
                                         const float* const* w_att, const float* const* Z, float* const* dZ3,
                                         float* const* g_head, float* const* head_partial, int64_t n, int32_t packed,
                                         pamnet_stream_t stream) {
    if (n < 0 || n_layers < 0) return PAMNET_EINVAL;
    if (n == 0 || n_layers == 0) return PAMNET_OK;
    if (!d_out || !d_att || !weights || !w_out || !w_att || !Z || !dZ3 || !g_head || !head_partial) return PAMNET_ENULL;
    hipStream_t st = as_stream(stream);
    for (int64_t l0 = 0; l0 < n_layers; l0 += MAX_HEAD_LAYERS) {
        const int nl = (int)(n_layers - l0 < MAX_HEAD_LAYERS ? n_layers - l0 : MAX_HEAD_LAYERS);
        HeadBwdBatch hb{};
        for (int i = 0; i < nl; ++i) {
            const int64_t l = l0 + i;
            HeadBwdLayer& h = hb.l[i];
            h.d_out = d_out[l], h.d_att = d_att[l];
            for (int k = 0; k < 3; ++k) h.W[k] = weights[3 * l + k];
            h.w_out = w_out[l], h.w_att = w_att[l], h.Z = Z[l];
            h.dZ3 = dZ3[l], h.g_head = g_head[l], h.head_partial = head_partial[l];
            if (!h.d_out || !h.d_att || !h.W[0] || !h.W[1] || !h.W[2] || !h.w_out || !h.w_att || !h.Z || !h.dZ3 ||
                !h.g_head || !h.head_partial)
                return PAMNET_ENULL;
        }
        const dim3 grid((unsigned)ceil_div(n, BMN), (unsigned)nl);
        if (packed) hipLaunchKernelGGL(node_heads_bwd_kernel<true>, grid, dim3(TWG), 0, st, hb, n);
        else hipLaunchKernelGGL(node_heads_bwd_kernel<false>, grid, dim3(TWG), 0, st, hb, n);
        PAMNET_LAUNCH_CHECK();
    }
    return PAMNET_OK;
}
