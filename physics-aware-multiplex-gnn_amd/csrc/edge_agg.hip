// Edge MLP -> node segment-sum as ONE kernel (dim = 128): the message of layers/global_message_passing.py:52-56 is
// reduced over its target node (PyG add-aggregation, global_message_passing.py:38) while the message tile is still in
// LDS -- the [E,128] message tensor of the reference (and of pamnet_global_edge_fwd_f32 + pamnet_segment_sum_f32)
// never exists in memory.
//
// Work split.  Edges are stored in CSR order of their target node, so a node's messages are consecutive rows.  The
// kernels of edge_chain.hip cut the rows into 16-row tiles regardless of the nodes; here the cuts are NODE-ALIGNED:
//   * workgroup b of G owns the nodes [cut(b), cut(b+1)), cut(k) = the node boundary nearest to row k*E/G
//     (two loads: row_of[k*E/G], ptr[.]), i.e. ~E/G rows give or take half a node degree;
//   * it walks its rows in chunks of 16*MTX rows (full MFMA tiles): e rows -> LDS, two fp32-MFMA GEMMs, epilogue
//     z = W_e e + b + P_i[i] + P_j[j], msg = SiLU(z) * (W_ea e) written back into the LDS tile, then one 32-lane
//     group per node adds the node's rows IN CSR ORDER onto init[node] and stores the 512-byte output row; a node
//     cut by a chunk boundary hands its running sum to the next chunk through LDS (same order).
//   Every output row has exactly one owner and a fixed summation order: no atomics, no carry between workgroups,
//   bitwise identical from run to run and independent of how the batch is cut into workgroups / chunks.
// The backward kernel mirrors it: d z rows are reduced per target node (d P_i) from the LDS tile before the dX GEMMs
// overwrite it; only the reduction by SOURCE node (transposed CSR) remains a separate segment sum.
//
// local_agg: the two chained aggregations of the local layer (layers/local_message_passing.py:49-54),
//   m_t[e] = m_ji[e] + sum_{r in rows(e)} m_nb[idx[r]] * s[r],   x2[i] = x1[i] + sum_{e -> i} q3[e] * m_t[e],
// as one launch (one lane group per edge, node sums through LDS in CSR order).
#include <type_traits>

#include "edge_core.h"

using namespace edge;

#ifdef PAMNET_PHASE_PROBE
// Development aid (tools/agg_probe.py builds a private copy of this file with -DPAMNET_PHASE_PROBE): shader-clock
// timestamps of the middle workgroup -- slots [0, 32): wave 0, [32, 64): wave 4 -- and wall clock at start / end of every
// workgroup.  Never compiled into libpamnet_hip.so.
__device__ long long pamnet_agg_probe[64];
__device__ long long pamnet_agg_probe_wg[2 * 1024];
#define APROBE(i)                                                                                         \
    do {                                                                                                  \
        if (blockIdx.x == gridDim.x / 2 && (threadIdx.x == 0 || threadIdx.x == 256) && (i) < 32)            \
            pamnet_agg_probe[(threadIdx.x >> 8) * 32 + (i)] = clock64();                                    \
    } while (0)
// the same inside a chunk loop, for ONE chosen chunk of the workgroup (-DPAMNET_PROBE_CHUNK=k; default: every chunk, i.e. the
// last one's stamps survive); the loop keeps `aprobe_chunk`
#ifndef PAMNET_PROBE_CHUNK
#define PAMNET_PROBE_CHUNK -1
#endif
#define APROBE_C(i)                                                                     \
    do {                                                                                \
        if (PAMNET_PROBE_CHUNK < 0 || aprobe_chunk == PAMNET_PROBE_CHUNK) APROBE(i);    \
    } while (0)
#define APROBE_CHUNK_DECL int aprobe_chunk = 0
#define APROBE_CHUNK_NEXT ++aprobe_chunk
#define APROBE_WG(slot)                                                                                       \
    do {                                                                                                      \
        if (threadIdx.x == 0 && blockIdx.x < 1024) pamnet_agg_probe_wg[2 * blockIdx.x + slot] = wall_clock64(); \
    } while (0)
extern "C" int pamnet_agg_probe_read(long long* host64, long long* wg, int n) {
    int rc = (int)hipMemcpyFromSymbol(host64, HIP_SYMBOL(pamnet_agg_probe), sizeof(long long) * 64);
    if (rc) return rc;
    return (int)hipMemcpyFromSymbol(wg, HIP_SYMBOL(pamnet_agg_probe_wg), sizeof(long long) * 2 * n);
}
// the ping-pong kernel: five stamps per wave (slots [8 wave, 8 wave + 5)) in iteration PAMNET_PROBE_T of the middle workgroup
#ifndef PAMNET_PROBE_T
#define PAMNET_PROBE_T 20
#endif
#define PPROBE(i, t)                                                                                                    \
    do {                                                                                                                \
        if (blockIdx.x == gridDim.x / 2 && (threadIdx.x & 63) == 0 && (t) == PAMNET_PROBE_T)                            \
            pamnet_agg_probe[8 * (threadIdx.x >> 6) + (i)] = clock64();                                                 \
    } while (0)
#else
#define PPROBE(i, t)
#define APROBE(i)
#define APROBE_C(i)
#define APROBE_CHUNK_DECL
#define APROBE_CHUNK_NEXT
#define APROBE_WG(slot)
#endif

namespace {

constexpr int NMAX = 511;                 // nodes per chunk (their CSR offsets are staged in LDS)

// XCD-aware work order.  The dispatcher deals workgroups to the 8 XCDs round-robin (workgroup b runs on XCD b % 8), and
// every XCD has an L2 of its own: with node range k handed to workgroup k, eight NEIGHBOURING node ranges -- the same
// molecule / complex, whose P_i / P_j rows and CSR slices they all gather -- land on eight different L2s and every
// node-plane row is fetched from the fabric up to eight times (measured at the PDBbind shape: 1.44x the algorithmic bytes
// in the inference instantiation, profiles/r02_edge_agg_pmc.json).  Workgroup b takes range
//   order(b) = (b % 8) * (G / 8) + b / 8  (+ the remainder spread over the first G % 8 XCDs),
// so the workgroups of one XCD own one contiguous eighth of the nodes.  A bijection on [0, G): every range still has
// exactly one owner, results are bitwise unchanged.
__device__ __forceinline__ int xcd_order(int b, int G) {
    constexpr int X = 8;
    const int x = b % X, i = b / X, q = G / X, r = G % X;
    return i + x * q + (x < r ? x : r);
}

// node boundary nearest to row k*m/G (0 for k = 0, n for k = G): monotone in k, so the ranges tile [0, n)
__device__ __forceinline__ int seg_cut(const int32_t* __restrict__ ptr, const int32_t* __restrict__ row_of, int64_t n,
                                       int64_t m, int k, int G) {
    if (k <= 0) return 0;
    if (k >= G || m == 0) return (int)n;
    const int64_t t = (int64_t)k * m / G;
    const int node = row_of[t];
    const int64_t a = ptr[node], b = ptr[node + 1];
    return (t - a <= b - t) ? node : node + 1;
}

// A workgroup walks its rows [rb, re) in chunks of CAP rows (full MFMA tiles; the last one may be short).  A chunk
// touches the nodes [c0, c1]: c0 may have begun in the previous chunk (its running sum arrives in `carry`), c1 may
// continue in the next one (its running sum leaves in `carry`); every node in between is complete.  Workgroup
// boundaries are node-aligned, so nothing is ever carried between workgroups.
struct Chunk {
    int c1;                // last node of the chunk (inclusive)
    int64_t r1;            // rows [r0, r1)
};

__device__ __forceinline__ Chunk plan_chunk(const int32_t* __restrict__ ptr, const int32_t* __restrict__ row_of, int c0,
                                            int64_t r0, int ne, int64_t re, int cap) {
    Chunk ch;
    ch.r1 = r0 + cap < re ? r0 + cap : re;
    ch.c1 = ch.r1 < re ? row_of[ch.r1 - 1] : ne - 1;       // the workgroup's last chunk also takes trailing empty nodes
    if (ch.c1 - c0 + 1 > NMAX) {                            // a long run of nodes without edges
        ch.c1 = c0 + NMAX - 1;
        const int64_t e1 = ptr[ch.c1 + 1];
        if (e1 < ch.r1) ch.r1 = e1;
    }
    return ch;
}

// ---- the same plan on the scalar unit -----------------------------------------------------------------------------------
// plan_chunk's loads are wave-uniform, but behind the first global store of a kernel the compiler can no longer prove ptr /
// row_of unwritten and issues them as VECTOR loads -- and vector memory operations retire in order, loads and stores on one
// counter: each of the plan's two or three dependent loads at the head of a chunk then waited for every store of the previous
// chunk and for the prefetched rows (tools/edge_wgrad_phase_probe.py measured what that costs in the round-5 kernel).  s_load
// has a counter of its own.  Addresses must be wave-uniform (they are: they derive from scalar loads only).
__device__ __forceinline__ void sload2(const int32_t* pa, const int32_t* pb, int& va, int& vb) {
    asm volatile("s_load_dword %0, %2, 0x0\n\ts_load_dword %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(va), "=&s"(vb)
                 : "s"(pa), "s"(pb)
                 : "memory");
}
struct SPlan {
    int c0, c1, rows;      // nodes [c0, c1] (c1 inclusive), rows [r0, r1)
    bool open_end;         // node c1 continues in the next chunk
    int64_t r0, r1;
};
// the chunk that starts at node c0 / row r0 (plan_chunk's rules), and whether its last node goes on behind it
__device__ __forceinline__ SPlan splan(const int32_t* __restrict__ ptr, const int32_t* __restrict__ row_of, int c0, int64_t r0,
                                       int ne, int64_t re, int cap) {
    SPlan p;
    p.c0 = c0, p.r0 = r0, p.c1 = c0 - 1, p.r1 = r0, p.open_end = false;
    if (c0 < ne) {
        int64_t r1 = r0 + cap < re ? r0 + cap : re;
        int last = 0, nxt = 0;
        if (r0 < re) sload2(row_of + r1 - 1, row_of + (r1 < re ? r1 : re - 1), last, nxt);
        int c1 = r1 < re ? last : ne - 1;                       // the workgroup's last chunk also takes trailing empty nodes
        if (c1 - c0 + 1 > NMAX) {                               // a long run of nodes without edges
            c1 = c0 + NMAX - 1;
            int e1 = 0, dummy = 0;
            sload2(ptr + c1 + 1, ptr + c1 + 1, e1, dummy);
            if (e1 < r1) r1 = e1;
            if (r1 < re) sload2(row_of + r1, ptr + c1 + 1, nxt, dummy);
        }
        p.c1 = c1, p.r1 = r1;
        p.open_end = r1 < re && nxt == c1;
    }
    p.rows = (int)(p.r1 - p.r0);
    return p;
}
// ptr[c0 + t] of a chunk's nodes for thread t, clamped to the array (raw: the subtraction of r0 waits for the load)
__device__ __forceinline__ int load_node_ptr(const int32_t* __restrict__ ptr, int c0, int64_t n) {
    int64_t k = (int64_t)c0 + threadIdx.x;
    k = k < 0 ? 0 : (k < n ? k : n);
    return ptr[k];
}

__device__ __forceinline__ void seq_add(float4& s, const float* __restrict__ tile, int q0, int q1, int c4) {
    int q = q0;
    for (; q + 4 <= q1; q += 4) {                                      // loads independent, adds strictly in row order
        const float4 v0 = lds4(tile, q, c4), v1 = lds4(tile, q + 1, c4), v2 = lds4(tile, q + 2, c4),
                     v3 = lds4(tile, q + 3, c4);
        s = f4add(f4add(f4add(f4add(s, v0), v1), v2), v3);
    }
    for (; q < q1; ++q) s = f4add(s, lds4(tile, q, c4));
}

// the same sums with NB loads in flight (same order)
template <int NB>
__device__ __forceinline__ void seq_add_wide(float4& s, const float* __restrict__ tile, int q0, int q1, int c4) {
    int q = q0;
    for (; q + NB <= q1; q += NB) {
        float4 v[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) v[u] = lds4(tile, q + u, c4);
#pragma unroll
        for (int u = 0; u < NB; ++u) s = f4add(s, v[u]);
    }
    seq_add(s, tile, q, q1, c4);
}

// out[node] = (sum of the node's rows of `tile`, in CSR order) + init[node], one 32-lane group per node.
// sptr[k] = ptr[c0 + k] - r0 for k = 0 .. nn (raw: negative = the node began before the chunk, > rows = it goes on).
template <int NGRP, int WIDE = 0>
__device__ __forceinline__ void reduce_nodes(int c0, int nn, int rows, const float* __restrict__ tile,
                                             const int* __restrict__ sptr, const float4* __restrict__ carry_in,
                                             float4* __restrict__ carry_out, const float* __restrict__ init,
                                             float* __restrict__ out) {
    const int grp = threadIdx.x >> 5, c4 = threadIdx.x & 31;
    for (int k = grp; k < nn; k += NGRP) {
        const int b = sptr[k], e = sptr[k + 1];
        // the node's `init` row joins its finished sum: requested here, used behind the row adds (as the sum's first term it
        // was a bare global round trip ahead of every node's adds)
        const float4 iv = init ? ldg4(init, c0 + k, DIM, c4) : f4zero();
        float4 s = b < 0 ? carry_in[c4] : f4zero();                     // (a carry arrives only at k = 0)
        if constexpr (WIDE > 0) seq_add_wide<WIDE>(s, tile, b < 0 ? 0 : b, e > rows ? rows : e, c4);
        else seq_add(s, tile, b < 0 ? 0 : b, e > rows ? rows : e, c4);
        if (e > rows) carry_out[c4] = s;                                // only k = nn - 1 (another lane group than k = 0)
        else stg4(out, c0 + k, DIM, c4, f4add(s, iv));
    }
}

struct GAggFwd {
    const float *e, *We, *bm, *Wea, *Pi, *Pj, *init;
    const int32_t *ptr, *row_of, *col;
    const int32_t* cuts;      // optional [gridDim.x + 1] node cuts (pamnet_seg_cuts_i32): saves the dependent loads of seg_cut
    float *z, *ea, *out;
    int64_t m, n;
    int ld_we, ld_wea;
};

// PRE: the next chunk's e rows travel in registers while this chunk's GEMMs run (multi-chunk workgroups); the 9-tile
// instantiation (one chunk per workgroup at ~128 rows give or take half a node) has no registers to spare for it.
// PL: the e rows wait in LDS as bf16 piece planes, split by the thread that stages them (edge_core.h "piece planes": once
// per workgroup instead of once per wave); a tile's 12 KB slot takes the tile's z accumulators (fp32, 8.4 KB) once its
// GEMMs are done.
template <int MTX, bool PRE, bool SAVE, bool PL = false>
__global__ __launch_bounds__(WG8, 1) void global_edge_agg_fwd_kernel(GAggFwd a) {
    constexpr int FT = 16 * LDT * 4;                        // bytes of an fp32 tile
    __shared__ __attribute__((aligned(16))) char ldsb[PL ? MTX * (PTILE + FT) : 2 * MTX * FT];
    __shared__ int sptr[NMAX + 1];
    __shared__ float4 carry[2][32];                         // running sum of a node that spans two chunks (in / out)
    float* S0 = reinterpret_cast<float*>(ldsb);             // (!PL) e rows, then z
    char* P = ldsb;                                         // (PL)  e pieces per tile, then that tile's z
    float* S1 = reinterpret_cast<float*>(ldsb + (PL ? MTX * PTILE : MTX * FT));
    auto stage_e = [&](int r, int c4_, const float4& v) {
        if constexpr (PL) st_pieces4(P, r, c4_, v);
        else st_lds4(S0, r, c4_, v);
    };
    constexpr int CAP = MTX * 16;
    const float* __restrict__ e = a.e;
    const float* __restrict__ Pi = a.Pi;
    const float* __restrict__ Pj = a.Pj;
    const int32_t* __restrict__ ptr = a.ptr;
    const int32_t* __restrict__ row_of = a.row_of;
    const int32_t* __restrict__ col = a.col;
    float* __restrict__ zs = a.z;
    float* __restrict__ eas = a.ea;
    const int wc = wave_col<8>();
    APROBE_WG(0);
    APROBE(0);
    const BiasSet<1> zero_bias = lane_biases<1>(nullptr, wc);
    const BiasSet<1> bv = lane_biases<1>(a.bm, wc);
    // both weight slices as resident bf16x3 pieces (edge_core.h): the two GEMMs run on the bf16 matrix pipe at fp32
    // accuracy and share one split of every A fragment
    WFragB1 f1, f2;
    load_wfragb1<false>(f1, a.We, a.ld_we, wc);
    load_wfragb1<false>(f2, a.Wea, a.ld_wea, wc);
    APROBE(20);
    const int wg = xcd_order(blockIdx.x, gridDim.x);
    const int nb = a.cuts ? a.cuts[wg] : seg_cut(ptr, row_of, a.n, a.m, wg, gridDim.x);
    const int ne = a.cuts ? a.cuts[wg + 1] : seg_cut(ptr, row_of, a.n, a.m, wg + 1, gridDim.x);
    const int64_t rb = ptr[nb], re = ptr[ne];
    APROBE(21);
    constexpr int RPP = 16, NI = MTX;                       // sweep geometry of 512 threads: 16 rows per pass
    constexpr int SC = 3;                                   // tiles per pipeline stage
    const int c4 = threadIdx.x & 31, rr = threadIdx.x >> 5;
    float4 pre[PRE ? NI : 1];
    if (PRE && rb < re) {
#pragma unroll
        for (int i = 0; i < NI; ++i) pre[i] = ldg4z_nt(e, rb + rr + RPP * i, re, DIM, c4);
    }
    int par = 0;
    // the plan one chunk ahead on the scalar unit (splan above)
    SPlan cur = splan(ptr, row_of, __builtin_amdgcn_readfirstlane(nb), (int64_t)__builtin_amdgcn_readfirstlane((int)rb), ne, re, CAP);
    APROBE_CHUNK_DECL;
    while (cur.c0 < ne) {
        APROBE_C(30);
        const int c0 = cur.c0, c1 = cur.c1, rows = cur.rows;
        const int64_t r0 = cur.r0, r1 = cur.r1;
        const SPlan nxt = splan(ptr, row_of, cur.open_end ? c1 : c1 + 1, r1, ne, re, CAP);
        const int mt = (rows + 15) >> 4;
        const int nn = c1 - c0 + 1;
        // CSR offsets of the chunk's nodes: requested now (raw), parked in LDS behind the staging of the e rows
        const int mypraw = load_node_ptr(ptr, c0, a.n);
        APROBE_C(22);
        if (rows > 0) {
            // node indices of this thread's rows (clamped to the chunk: every load below is unconditional), requested FIRST: the
            // node-plane gathers behind the barrier depend on them, and requested after the staging they were a bare memory
            // round trip per chunk (tools/agg_probe.py, chunk 10 of 26 at the PDBbind shape: 2 900 of 33 800 cycles)
            int ri[NI], ci[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                int64_t g = r0 + rr + RPP * i;
                g = g < r1 ? g : r1 - 1;
                ri[i] = row_of[g], ci[i] = col[g];
            }
            if (PRE) {
#pragma unroll
                for (int i = 0; i < NI; ++i)
                    if (RPP * i < 16 * mt) stage_e(rr + RPP * i, c4, pre[i]);
                // (no branch around the requests: behind the workgroup's last row the guarded load reads that row again and
                // zeroes it, and the requests in flight stay countable -- vector loads and stores retire in order on one counter)
#pragma unroll
                for (int i = 0; i < NI; ++i) pre[i] = ldg4zl_nt(e, r1 + rr + RPP * i, re, DIM, c4);
            } else {
#pragma unroll
                for (int i = 0; i < NI; ++i)
                    if (RPP * i < 16 * mt) stage_e(rr + RPP * i, c4, ldg4z_nt(e, r0 + rr + RPP * i, r1, DIM, c4));
            }
            APROBE_C(23);
            if ((int)threadIdx.x <= nn) sptr[threadIdx.x] = mypraw - (int)r0;
            APROBE_C(1);
            __syncthreads();
            APROBE_C(2);
            // Software pipeline over sub-chunks of SC tiles: stage s runs the two GEMMs of sub-chunk s, then the epilogue of
            // sub-chunk s-1 (SiLU, gate, z / ea stores, message -> LDS).  The epilogue holds no load latency: the node-plane
            // rows P_i[i], P_j[j] of a sub-chunk are requested a stage ahead (right after the previous epilogue has consumed
            // its own), and its global stores are issued last -- vmcnt retires in order, a store ahead of a load would
            // make the load's wait drain it (the epilogue of 48 rows went from 9 100 to 2 400 cycles, tools/agg_probe.py).
            // (Letting the two waves of a SIMD take GEMM and epilogue in opposite order was measured and dropped: an fp32
            // MFMA burst leaves the other wave's VALU work no issue slots -- fp32 matrix and vector peak are the same
            // 157 TFLOP/s on this part -- so the epilogue just ran four times longer.)
            const int nsub = (mt + SC - 1) / SC;
            float4 gpi[SC], gpj[SC];
            auto gather = [&](int sb) {
#pragma unroll
                for (int i = 0; i < SC; ++i) {
                    const int k = sb * SC + i < NI ? sb * SC + i : NI - 1;
                    gpi[i] = ldg4(Pi, ri[k], DIM, c4);
                    gpj[i] = ldg4(Pj, ci[k], DIM, c4);
                }
            };
            gather(0);
            for (int sb = 0; sb <= nsub; ++sb) {
                const bool do_g = sb < nsub, do_e = sb > 0;
                const int smt = do_g ? (mt - sb * SC < SC ? mt - sb * SC : SC) : 0;
                const int goff = sb * SC * 16, eoff = (sb - 1) * SC * 16;
                AccSet<SC, 1> ag, az;
                auto gemm = [&]() {
                    ag.zero();
                    az.zero();
                    if constexpr (PL) mma_p16<SC, false, (PRE ? 1 : 3)>(P + sb * SC * PTILE, f1, az.a[0], f2, ag.a[0], smt);
                    else mma_b16<SC, false, (PRE ? 1 : 3)>(S0 + goff * LDT, f1, az.a[0], f2, ag.a[0], smt);
                };
                auto epi = [&]() {
                    float4 zz[SC], gate[SC];
#pragma unroll
                    for (int i = 0; i < SC; ++i) {
                        const int r = eoff + rr + RPP * i;
                        if (r < 16 * mt) {                     // (the last sub-chunk may hold fewer than SC tiles)
                            const float4 zacc = PL ? lds4(reinterpret_cast<const float*>(P + (r >> 4) * PTILE), r & 15, c4)
                                                   : lds4(S0, r, c4);
                            zz[i] = f4add(f4add(zacc, gpi[i]), gpj[i]);
                            gate[i] = lds4(S1, r, c4);
                            st_lds4(S1, r, c4, f4mul(f4silu(zz[i]), gate[i]));  // the message stays on chip
                            if constexpr (SAVE && PL) {        // (piece-plane form: stores first, see below)
                                if (r0 + r < r1) {
                                    stg4_nt(zs, r0 + r, DIM, c4, zz[i]);
                                    stg4_nt(eas, r0 + r, DIM, c4, gate[i]);
                                }
                            }
                        }
                    }
                    // the next epilogue's node rows: a whole stage of lead.  In the piece-plane form the saves are issued
                    // BEFORE them -- the values die at their store, which keeps the kernel inside 256 registers; the stores
                    // have the whole next GEMM stage to retire before anything waits behind them (vmcnt retires in order).
                    if (sb < nsub) gather(sb);
                    if (SAVE && !PL) {                         // backward-only saves (inference instantiation: none)
#pragma unroll
                        for (int i = 0; i < SC; ++i) {
                            const int64_t g = r0 + eoff + rr + RPP * i;
                            if (g < r1) {
                                stg4_nt(zs, g, DIM, c4, zz[i]);
                                stg4_nt(eas, g, DIM, c4, gate[i]);
                            }
                        }
                    }
                };
                APROBE_C(3 + 4 * sb);
                if (do_g) gemm();
                APROBE_C(4 + 4 * sb);
                // ONE barrier per stage: every wave is done reading the e rows of sub-chunk sb (their slots take the
                // accumulators now), and the accumulators stored in stage sb - 1 are visible to the epilogue below.  The
                // accumulators leave before the epilogue starts: they are not live across it (24 registers).
                __syncthreads();
                if (do_g) {
                    if constexpr (PL) store_set<SC, 1>(az, reinterpret_cast<float*>(P + sb * SC * PTILE), wc, bv, smt, PTILE / 4);
                    else store_set<SC, 1>(az, S0 + goff * LDT, wc, bv, smt);
                    store_set<SC, 1>(ag, S1 + goff * LDT, wc, zero_bias, smt);
                }
                APROBE_C(5 + 4 * sb);
                if (do_e) epi();
                APROBE_C(6 + 4 * sb);
            }
        } else if ((int)threadIdx.x <= nn) {
            sptr[threadIdx.x] = mypraw - (int)r0;
        }
        __syncthreads();
        APROBE_C(28);
        reduce_nodes<16>(c0, nn, rows, S1, sptr, carry[par], carry[par ^ 1], a.init, a.out);
        par ^= 1;
        __syncthreads();
        APROBE_C(29);
        APROBE_WG(1);
        APROBE_CHUNK_NEXT;
        cur = nxt;
    }
}

// ---- round 6: the same op with the two halves of the workgroup in OPPOSITE phases ("ping-pong") --------------------------
// In the kernel above every wave of the workgroup is in the same phase: while the eight waves run the GEMMs the vector
// units idle, and during staging / epilogue / node sums the matrix pipe does (profiles/r05_issue_slots_pdbbind_pmc.txt:
// 45 % parked, 30 % stalled, 3.6 VALU per MFMA; a 112-row chunk took 32 400 cycles of which 14 800 were GEMM).  Here the
// rows of a workgroup are a STREAM of 32-row groups and the two waves of every SIMD are always in different phases:
//
//     iteration t   phase A:  waves 0-3  GEMMs of group t (their 64 columns)   | waves 4-7  vector work
//                   phase B:  waves 0-3  vector work                           | waves 4-7  GEMMs of group t (theirs)
//
// one s_barrier between phases.  The vector work of an iteration is everything else of the pipeline, each item on the
// group whose turn it is:  epilogue of group t-1 (z = acc + P_i + P_j, SiLU, gate, saves; message -> LDS),  staging of
// group t+1 (rows -> bf16 piece planes), the gathers for group t's epilogue and the row loads of group t+3 (requested an
// iteration / two iterations ahead: no load is waited for in the phase that issues it), and -- ONE wave, the "walker" --
// the node sums of group t-2.  Waves 4-7 take the first tile of every group for the vector work, waves 0-2 the second
// (192 threads: three passes of six rows), wave 3 walks.
//
// Node sums without a plan.  The chunked kernel stages CSR offsets per chunk (splan, sptr) and reduces many nodes in
// parallel at the chunk's end.  A stream has no chunk ends: the walker adds the message rows one by one, in CSR order, onto
// a running sum that lives in its registers (two columns per lane), and writes out[node] = sum + init[node] whenever the
// row's node (the row_of value the epilogue threads loaded anyway, passed through LDS) changes -- nodes without rows in
// between get init.  Exactly the additions of reduce_nodes in the same order: every output bit is the chunked kernel's
// (tests/test_hip_edge_agg.py compares the two entry points bit for bit).
//
// LDS: piece planes of 2 groups (staged / multiplied), z accumulators of 2 groups (written / consumed), gate -> message
// tiles of 3 groups (written / finished / summed): 131 KB.  Registers: the resident weight pieces (96) + four accumulators
// + fragments on the GEMM side, the rows / gathers in flight on the vector side.
template <bool SAVE>
__global__ __launch_bounds__(WG8, 1) void global_edge_agg_fwd_pp_kernel(GAggFwd a) {
    constexpr int TG = 2;                                   // 16-row tiles per group
    constexpr int GR = 16 * TG;                             // rows per group
    constexpr int FT = 16 * LDT * 4;                        // bytes of an fp32 tile
    constexpr int TF = 16 * LDT;                            // floats of an fp32 tile
    constexpr int P_BYTES = 2 * TG * PTILE, Z_BYTES = 2 * TG * FT, M_BYTES = 3 * TG * FT;
    __shared__ __attribute__((aligned(16))) char ldsb[P_BYTES + Z_BYTES + M_BYTES];
    __shared__ int rn[3][GR];                               // node of every row of a group (for the walker)
    char* const Pb = ldsb;
    float* const Zb = reinterpret_cast<float*>(ldsb + P_BYTES);
    float* const Mb = reinterpret_cast<float*>(ldsb + P_BYTES + Z_BYTES);
    const float* __restrict__ e = a.e;
    const float* __restrict__ Pi = a.Pi;
    const float* __restrict__ Pj = a.Pj;
    const float* __restrict__ init = a.init;
    const int32_t* __restrict__ ptr = a.ptr;
    const int32_t* __restrict__ row_of = a.row_of;
    const int32_t* __restrict__ col = a.col;
    float* __restrict__ zs = a.z;
    float* __restrict__ eas = a.ea;
    float* __restrict__ out = a.out;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int half = wave >> 2;                             // 0: GEMMs in phase A, vector work in phase B; 1: the reverse
    const bool walker = wave == 3;
    const int wc = wave_col<8>();
    const float bias = lane_bias(a.bm, wc);
    WFragB1 f1, f2;                                         // both weight slices as resident bf16x3 pieces (edge_core.h)
    load_wfragb1<false>(f1, a.We, a.ld_we, wc);
    load_wfragb1<false>(f2, a.Wea, a.ld_wea, wc);
    const int wg = xcd_order(blockIdx.x, gridDim.x);
    const int nb = a.cuts ? a.cuts[wg] : seg_cut(ptr, row_of, a.n, a.m, wg, gridDim.x);
    const int ne = a.cuts ? a.cuts[wg + 1] : seg_cut(ptr, row_of, a.n, a.m, wg + 1, gridDim.x);
    const int64_t rb = ptr[nb], re = ptr[ne];

    // ---- matrix side: both GEMMs of a group on this wave's 16 columns ----------------------------------------------------
    auto gemm = [&](int t) __attribute__((always_inline)) {
#ifdef PP_NO_GEMM
        return;
#endif
        const char* P = Pb + (t & 1) * TG * PTILE;
        Acc<TG> az, ag;
        az.zero();
        ag.zero();
        // (the fragments of the next k-step are requested before this one's products: one wave per SIMD is multiplying, nobody
        // else covers its LDS round trips)
        Frag3 A0 = lds_frag3p(P, 0, 0), A1 = lds_frag3p(P, 1, 0);
#pragma unroll
        for (int q = 0; q < DIM / 32; ++q) {
            Frag3 N0, N1;
            if (q + 1 < DIM / 32) N0 = lds_frag3p(P, 0, q + 1), N1 = lds_frag3p(P, 1, q + 1);
            // the six piece products with i + j <= 2, small ones first (mfma6's order per accumulator); the four accumulators
            // are independent chains
#define PP_PROD(i, j)                                        \
    az.v[0] = mfma_bf16(A0.p[i], f1.p[q][j], az.v[0]);       \
    az.v[1] = mfma_bf16(A1.p[i], f1.p[q][j], az.v[1]);       \
    ag.v[0] = mfma_bf16(A0.p[i], f2.p[q][j], ag.v[0]);       \
    ag.v[1] = mfma_bf16(A1.p[i], f2.p[q][j], ag.v[1]);
            PP_PROD(2, 0) PP_PROD(1, 1) PP_PROD(0, 2) PP_PROD(1, 0) PP_PROD(0, 1) PP_PROD(0, 0)
#undef PP_PROD
            if (q + 1 < DIM / 32) A0 = N0, A1 = N1;
        }
        acc_store<TG>(az, Zb + (t & 1) * TG * TF, wc, bias, TG);
        acc_store<TG>(ag, Mb + (t % 3) * TG * TF, wc, 0.f, TG);
    };
    const int nb_ = nb, ne_ = ne;

    // ---- roles.  The split is made ONCE, at the top: every role is a loop of its own with a straight-line body, so the
    // compiler counts each wave's memory operations exactly (vector memory retires in order on one counter; with the roles as
    // branches inside one loop it has to assume a wave may have taken any of them and waits for everything).  All roles pass
    // the same 1 + 2 (T + 1) barriers.
    if (walker) {
        // ---- the walker: node `cur` is open (if `have`), `s` its running sum, `iv` its init row -----------------------------
        int cur = nb_ - 1;
        bool have = false;
        float2 s = make_float2(0.f, 0.f), iv = make_float2(0.f, 0.f);
        auto emit_and_fill = [&](int upto) __attribute__((always_inline)) {   // close the open node; empty nodes get init
            if (have) *reinterpret_cast<float2*>(out + (int64_t)cur * DIM + 2 * lane) = make_float2(s.x + iv.x, s.y + iv.y);
            for (int k = cur + 1; k < upto; ++k) {
                const float2 i2 = init ? *reinterpret_cast<const float2*>(init + (int64_t)k * DIM + 2 * lane) : make_float2(0.f, 0.f);
                *reinterpret_cast<float2*>(out + (int64_t)k * DIM + 2 * lane) = make_float2(0.f + i2.x, 0.f + i2.y);
            }
        };
        if (rb >= re) {                                     // a range without rows: its nodes get init
            emit_and_fill(ne_);
            return;
        }
        const int T = (int)((re - rb + GR - 1) / GR);
        auto walk = [&](int k) __attribute__((always_inline)) {              // node sums of group k: rows in CSR order
            const int64_t left = re - (rb + (int64_t)GR * k);
            const int rows = left < GR ? (int)left : GR;
            const int slot = k % 3;
            const float* Mk = Mb + slot * TG * TF + 2 * lane;
            const int myn = rn[slot][lane & (GR - 1)];
            // every row of the group requested at once (the walker has the registers the workers spend on rows in flight);
            // then one scalar test per row: rows of the open node -- all but one or two of a complex's group -- cost two adds
            float2 v[GR];
#pragma unroll
            for (int r = 0; r < GR; ++r) v[r] = *reinterpret_cast<const float2*>(Mk + r * LDT);
            // bit r: row r starts another node than row r - 1 (row 0: than the open node)
            const int prevn = __shfl_up(myn, 1);
            const bool first = (lane & (GR - 1)) == 0;
            uint32_t changes = (uint32_t)__builtin_amdgcn_ballot_w64(lane < GR && myn != (first ? cur : prevn));
            if (rows < GR) changes &= (1u << rows) - 1u;
#pragma unroll
            for (int r = 0; r < GR; ++r) {
                if (r < rows) {
                    if (changes & (1u << r)) {
                        const int node = __builtin_amdgcn_readlane(myn, r);
                        emit_and_fill(node);
                        cur = node, have = true, s = make_float2(0.f, 0.f);
                        iv = init ? *reinterpret_cast<const float2*>(init + (int64_t)node * DIM + 2 * lane) : make_float2(0.f, 0.f);
                    }
                    s.x += v[r].x, s.y += v[r].y;
                }
            }
        };
        __syncthreads();
#pragma unroll 1
        for (int t = 0; t <= T; ++t) {
            PPROBE(0, t);
            if (t < T) gemm(t);
            PPROBE(1, t);
            __syncthreads();
            PPROBE(2, t);
            if (t >= 2) walk(t - 2);
            PPROBE(3, t);
            __syncthreads();
            PPROBE(4, t);
        }
        walk(T - 1);
        emit_and_fill(ne_);
        return;
    }
    if (rb >= re) return;
    const int T = (int)((re - rb + GR - 1) / GR);           // groups of this workgroup

    // ---- workers: the vector side of the pipeline on the half's tile of every group ---------------------------------------
    // waves 4-7: rows 0 .. 15 of a group, eight rows per pass, two passes; waves 0-1: rows 16 .. 19 / 22 .. 25 / 28 .. 31 (six
    // rows per pass, three passes); wave 2: rows 20, 21, 26, 27 (two passes)
    const int c4 = tid & 31;
    const int rr = (half ? tid - 256 : tid) >> 5;
    const int trow0 = (half ? 0 : 16) + rr;
    auto worker = [&](auto rpp_c, auto niw_c, auto first_c) __attribute__((always_inline)) {
        constexpr int RPP = decltype(rpp_c)::value, NIW = decltype(niw_c)::value;
        constexpr bool GEMM_FIRST = decltype(first_c)::value;
        float4 pre[2][NIW], gpi[NIW], gpj[NIW];             // rows of groups t+1 / t+2 in flight; the gathers of group t
        int ri[NIW], ci[NIW];
        auto grow = [&](int k, int i) __attribute__((always_inline)) { return rb + (int64_t)GR * k + trow0 + RPP * i; };
        auto load_e = [&](auto slot_c, int k) __attribute__((always_inline)) {
            constexpr int SL = decltype(slot_c)::value;
#pragma unroll
            for (int i = 0; i < NIW; ++i) pre[SL][i] = ldg4zl_nt(e, grow(k, i), re, DIM, c4);
        };
        auto stage = [&](auto slot_c, int k) __attribute__((always_inline)) {
            constexpr int SL = decltype(slot_c)::value;
            char* P = Pb + (k & 1) * TG * PTILE;
#pragma unroll
            for (int i = 0; i < NIW; ++i) st_pieces4(P, trow0 + RPP * i, c4, pre[SL][i]);
        };
        auto load_idx = [&](int k) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < NIW; ++i) {
                int64_t g = grow(k, i);
                g = g < re ? g : re - 1;
                ri[i] = row_of[g], ci[i] = col[g];
            }
        };
        auto gather = [&](int k) __attribute__((always_inline)) {
            const int slot = k % 3;
#pragma unroll
            for (int i = 0; i < NIW; ++i) {
                gpi[i] = ldg4(Pi, ri[i], DIM, c4);
                gpj[i] = ldg4(Pj, ci[i], DIM, c4);
                if (c4 == 0) rn[slot][trow0 + RPP * i] = ri[i];
            }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        // KIND 0: first iteration (no epilogue yet), 1: steady state, 2: behind the last group (epilogue only, row masks)
        auto vec = [&](auto kind_c, auto par_c, int t) __attribute__((always_inline)) {
            constexpr int KIND = decltype(kind_c)::value;
#ifdef PP_VEC_PRIO
            __builtin_amdgcn_s_setprio(PP_VEC_PRIO);
#endif
#ifdef PP_NO_VEC
            return;
#endif
            if constexpr (KIND >= 1) {                      // epilogue of group t - 1
                const int k = t - 1;
                float* Zk = Zb + (k & 1) * TG * TF;
                float* Mk = Mb + (k % 3) * TG * TF;
                // (every LDS operand of the phase requested before the first is used: one wave per SIMD is doing this work --
                // nobody covers its round trips, and what a phase waits for is this wave)
                float4 zacc_[NIW], gate_[NIW];
#pragma unroll
                for (int i = 0; i < NIW; ++i) {
                    zacc_[i] = lds4(Zk, trow0 + RPP * i, c4);
                    gate_[i] = lds4(Mk, trow0 + RPP * i, c4);
                }
#pragma unroll
                for (int i = 0; i < NIW; ++i) {
                    const int r = trow0 + RPP * i;
                    const float4 zacc = zacc_[i];
                    const float4 gate = gate_[i];
                    const float4 zz = f4add(f4add(zacc, gpi[i]), gpj[i]);
#ifdef PP_NO_SILU
                    st_lds4(Mk, r, c4, f4mul(zz, gate));
#else
                    st_lds4(Mk, r, c4, f4mul(f4silu(zz), gate));           // the message stays on chip
#endif
                    if constexpr (SAVE) {
                        // the saves go out at once (the values die here).  Nothing this thread waits for below is younger
                        // than they are except the gathers it requests now and consumes a whole iteration later.
                        const int64_t g = grow(k, i);
                        if (KIND == 1 || g < re) {
                            stg4_nt(zs, g, DIM, c4, zz);
                            stg4_nt(eas, g, DIM, c4, gate);
                        }
                    }
                }
            }
            PPROBE(5, t);
            if constexpr (KIND <= 1) {
                stage(par_c, t + 1);                        // (behind the last group: zeros into a slot nobody reads)
                PPROBE(6, t);
                gather(t);
                load_idx(t + 1);
                load_e(par_c, t + 3);
                PPROBE(7, t);
            }
#ifdef PP_VEC_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
        };
        auto iter = [&](auto kind_c, auto par_c, int t) __attribute__((always_inline)) {
            constexpr int KIND = decltype(kind_c)::value;
            PPROBE(0, t);
            if constexpr (GEMM_FIRST) {
                if constexpr (KIND != 2) gemm(t);
                PPROBE(1, t);
                __syncthreads();
                PPROBE(2, t);
                vec(kind_c, par_c, t);
                PPROBE(3, t);
                __syncthreads();
            } else {
                vec(kind_c, par_c, t);
                PPROBE(1, t);
                __syncthreads();
                PPROBE(2, t);
                if constexpr (KIND != 2) gemm(t);
                PPROBE(3, t);
                __syncthreads();
            }
            PPROBE(4, t);
        };
        using K0 = std::integral_constant<int, 0>;
        using K1 = std::integral_constant<int, 1>;
        using K2 = std::integral_constant<int, 2>;
        load_e(I0{}, 0);                                    // group 0 staged, groups 1 and 2 requested
        load_idx(0);
        stage(I0{}, 0);
        load_e(I1{}, 1);
        load_e(I0{}, 2);
        __syncthreads();
        iter(K0{}, I1{}, 0);
        int t = 1;                                          // (t odd here: the group staged at t, t + 1, has parity 0)
#pragma unroll 1
        for (; t + 1 < T; t += 2) {
            iter(K1{}, I0{}, t);
            iter(K1{}, I1{}, t + 1);
        }
        if (t < T) {
            iter(K1{}, I0{}, t);
            ++t;
        }
        iter(K2{}, I0{}, T);
    };
    using C2 = std::integral_constant<int, 2>;
    using C3 = std::integral_constant<int, 3>;
    using C6 = std::integral_constant<int, 6>;
    using C8 = std::integral_constant<int, 8>;
    if (wave >= 4) worker(C8{}, C2{}, std::false_type{});
    else if (wave < 2) worker(C6{}, C3{}, std::true_type{});
    else worker(C6{}, C2{}, std::true_type{});
}

struct GAggBwd {
    const float *d_agg, *z, *ea, *We, *Wea;
    const int32_t *ptr, *row_of;
    const int32_t* cuts;      // optional node cuts (pamnet_seg_cuts_i32)
    float *dz, *dea, *d_e, *dPi;
    int64_t m, n;
    int ld_we, ld_wea, accumulate;
};

// dm[e] = d_agg[i(e)];  dz = dm * ea * SiLU'(z);  dea = dm * SiLU(z);  d_e (+)= dz W_e + dea W_ea;
// dPi[i] = sum_{e -> i} dz[e]   (the gradient of the target-side node projection, reduced from the LDS tile)
// PRE (multi-chunk workgroups): the z / ea rows of the NEXT chunk are requested while this chunk's tiles are being reduced and
// multiplied -- a workgroup is alone on its CU, so without it the memory phase and the GEMM phase of a chunk alternate and
// the HBM idles during the GEMMs of all 256 workgroups at once.
// PL: the two GEMM operands (d z, d ea) wait in LDS as bf16 piece planes written by the sweep that computes them
// (edge_core.h "piece planes": the split once per workgroup instead of once per wave); d z also as fp32 for the node sums.
template <int MTX, bool PRE = false, bool PL = false>
__global__ __launch_bounds__(WG8, 1) void global_edge_agg_bwd_kernel(GAggBwd a) {
    constexpr int FT = 16 * LDT * 4;                         // bytes of an fp32 tile
    __shared__ __attribute__((aligned(16))) char ldsb[PL ? MTX * (FT + 2 * PTILE) : 2 * MTX * FT];
    __shared__ int sptr[NMAX + 1];
    __shared__ float4 carry[2][32];
    float* S0 = reinterpret_cast<float*>(ldsb);
    float* S1 = S0 + MTX * 16 * LDT;                         // (!PL)
    char* P0 = ldsb + MTX * FT;                              // (PL)
    char* P1 = P0 + MTX * PTILE;
    constexpr int CAP = MTX * 16;
    const float* __restrict__ d_agg = a.d_agg;
    const float* __restrict__ zs = a.z;
    const float* __restrict__ eas = a.ea;
    const int32_t* __restrict__ ptr = a.ptr;
    const int32_t* __restrict__ row_of = a.row_of;
    float* __restrict__ dz = a.dz;
    float* __restrict__ dea = a.dea;
    float* __restrict__ d_e = a.d_e;
    const int accumulate = a.accumulate;
    const int wc = wave_col<8>();
    APROBE_WG(0);
    APROBE(0);
    const BiasSet<1> zero_bias = lane_biases<1>(nullptr, wc);
    WFragB1 f1, f2;                               // bf16x3 pieces of the two transposed weight slices (edge_core.h)
    load_wfragb1<true>(f1, a.We, a.ld_we, wc);
    load_wfragb1<true>(f2, a.Wea, a.ld_wea, wc);
    const int wg = xcd_order(blockIdx.x, gridDim.x);
    const int nb = a.cuts ? a.cuts[wg] : seg_cut(ptr, row_of, a.n, a.m, wg, gridDim.x);
    const int ne = a.cuts ? a.cuts[wg + 1] : seg_cut(ptr, row_of, a.n, a.m, wg + 1, gridDim.x);
    const int64_t re = ptr[ne];
    constexpr int RPP = 16, NI = MTX;
    constexpr bool GATHER_FIRST = MTX <= 3;       // (the larger instantiations have no registers for NI rows of d_agg at once)
    constexpr int BG = 3;                         // row tiles per MFMA group: three independent accumulator chains
    const int c4 = threadIdx.x & 31, rr = threadIdx.x >> 5;
    int par = 0;
    // the plan one chunk ahead on the scalar unit (splan above); the CSR offsets of a chunk's nodes are requested a chunk ahead
    SPlan cur = splan(ptr, row_of, __builtin_amdgcn_readfirstlane(nb), (int64_t)__builtin_amdgcn_readfirstlane(ptr[nb]), ne, re, CAP);
    int mypraw = load_node_ptr(ptr, cur.c0, a.n);
    float4 pz[PRE ? NI : 1], pe[PRE ? NI : 1];
    // AHEAD (round 6): the rows' node indices travel with the next chunk's z / ea rows (requested before the GEMMs), the d_agg rows
    // they select are requested right behind the GEMMs -- as two dependent round trips at the top of a chunk's sweep they were
    // ~1 600 of its ~3 900 cycles (tools/agg_probe.py, QM9 batch: three 48-row chunks per workgroup)
    constexpr bool AHEAD = PRE && GATHER_FIRST;
    int pgi[AHEAD ? NI : 1];
    float4 pdm[AHEAD ? NI : 1];
    auto next_indices = [&](int64_t rbeg, int64_t rend) __attribute__((always_inline)) {
        if constexpr (AHEAD) {
            const int64_t last = rend > rbeg ? rend - 1 : rbeg;      // (a chunk of edgeless nodes: any valid row, never used)
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int64_t g = rbeg + rr + RPP * i;
                pgi[i] = row_of[g < last ? g : last];
            }
        }
    };
    auto next_gather = [&]() __attribute__((always_inline)) {
        if constexpr (AHEAD) {
#pragma unroll
            for (int i = 0; i < NI; ++i) pdm[i] = ldg4(d_agg, pgi[i], DIM, c4);
        }
    };
    if (PRE && cur.r0 < re) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            pz[i] = ldg4z(zs, cur.r0 + rr + RPP * i, re, DIM, c4);
            pe[i] = ldg4z(eas, cur.r0 + rr + RPP * i, re, DIM, c4);
        }
        next_indices(cur.r0, cur.r1);
        next_gather();
    }
    APROBE(13);
    while (cur.c0 < ne) {
        const int c0 = cur.c0, c1 = cur.c1, rows = cur.rows;
        const int64_t r0 = cur.r0, r1 = cur.r1;
        const bool open_end = cur.open_end;
        const SPlan nxt = splan(ptr, row_of, open_end ? c1 : c1 + 1, r1, ne, re, CAP);
        const int mt = (rows + 15) >> 4;
        const int nn = c1 - c0 + 1;
        if ((int)threadIdx.x <= nn) sptr[threadIdx.x] = mypraw - (int)r0;
        mypraw = load_node_ptr(ptr, nxt.c0, a.n);
        APROBE(1);
        if (rows > 0) {
            // (Restructuring this sweep -- operands a row / a group of rows ahead, stores last -- changed nothing: all
            // workgroups run it at the same time and it moves 85 MB through L2 / Infinity Cache in ~8 us: bandwidth, not
            // latency, tools/agg_probe.py.)
            // the rows' node indices, then their d_agg rows: all requested before the first is used (clamped to the chunk, no
            // branch) -- row by row each gather waited for its index, and behind the first row's stores each index load
            // waited for those too (vector loads and stores retire in order on one counter)
            float4 dmv[GATHER_FIRST ? NI : 1];
            if constexpr (AHEAD) {
#pragma unroll
                for (int i = 0; i < NI; ++i) dmv[i] = pdm[i];
            } else if constexpr (GATHER_FIRST) {
                int gi[NI];
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    int64_t g = r0 + rr + RPP * i;
                    gi[i] = row_of[g < r1 ? g : r1 - 1];
                }
#pragma unroll
                for (int i = 0; i < NI; ++i) dmv[i] = ldg4(d_agg, gi[i], DIM, c4);
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                if (RPP * i >= 16 * mt) continue;
                const int r = rr + RPP * i;
                const int64_t g = r0 + r;
                float4 x = f4zero(), y = f4zero();
                if (g < r1) {
                    const float4 dm = GATHER_FIRST ? dmv[GATHER_FIRST ? i : 0] : ldg4(d_agg, row_of[g], DIM, c4);
                    const float4 zz = PRE ? pz[PRE ? i : 0] : ldg4(zs, g, DIM, c4);   // (non-temporal reads of the saves
                    const float4 ee = PRE ? pe[PRE ? i : 0] : ldg4(eas, g, DIM, c4);  //  measured slower: 793 vs 766 us)
                    x = f4mul(f4mul(dm, ee), f4dsilu(zz));
                    y = f4mul(dm, f4silu(zz));
                    stg4(dz, g, DIM, c4, x);
                    stg4(dea, g, DIM, c4, y);
                }
                st_lds4(S0, r, c4, x);
                if constexpr (PL) {
                    st_pieces4(P0, r, c4, x);
                    st_pieces4(P1, r, c4, y);
                } else {
                    st_lds4(S1, r, c4, y);
                }
            }
        }
        if (PRE && r1 < re) {                                  // the next chunk starts at r1: its rows travel during the GEMMs
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                pz[i] = ldg4z(zs, r1 + rr + RPP * i, re, DIM, c4);
                pe[i] = ldg4z(eas, r1 + rr + RPP * i, re, DIM, c4);
            }
            next_indices(r1, nxt.r1);
        }
        APROBE(2);
        __syncthreads();
        APROBE(3);
        reduce_nodes<16>(c0, nn, rows, S0, sptr, carry[par], carry[par ^ 1], nullptr, a.dPi);   // reads S0 only
        par ^= 1;
        APROBE(4);
        if (rows > 0) {
            constexpr bool PREACC = MTX <= 5;                  // (the 8 / 9-tile instantiations have no registers to spare)
            constexpr bool EARLY = PRE && MTX <= 3;            // ahead of the GEMMs where the registers allow
            float4 dacc[PREACC ? NI : 1];                      // accumulate operand of the final sweep
            auto fetch_acc = [&]() {
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    int64_t g = r0 + rr + RPP * i;
                    g = g < r1 ? g : r1 - 1;
                    dacc[PREACC ? i : 0] = ldg4(d_e, g, DIM, c4);
                }
            };
            if (EARLY && accumulate) fetch_acc();
            AccSet<MTX, 1> acc;
            acc.zero();
            if constexpr (PL) {
                mma_p16<MTX, true, BG>(P0, f1, acc.a[0], f1, acc.a[0], mt);
                mma_p16<MTX, true, BG>(P1, f2, acc.a[0], f2, acc.a[0], mt);
            } else {
                mma_b16<MTX, true, BG>(S0, f1, acc.a[0], f1, acc.a[0], mt);
                mma_b16<MTX, true, BG>(S1, f2, acc.a[0], f2, acc.a[0], mt);
            }
            APROBE(5);
            if (AHEAD && r1 < re) next_gather();               // (the indices arrived during the GEMMs)
            // otherwise requested here (the A fragments' registers are free again): the two barriers and the accumulator
            // stores cover the latency
            if (PREACC && !EARLY && accumulate) fetch_acc();
            __syncthreads();
            store_set<MTX, 1>(acc, S0, wc, zero_bias, mt);
            __syncthreads();
            APROBE(6);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int r = rr + RPP * i;
                const int64_t g = r0 + r;
                if (RPP * i < 16 * mt && g < r1) {
                    float4 v = lds4(S0, r, c4);
                    if (accumulate) v = f4add(v, PREACC ? dacc[PREACC ? i : 0] : ldg4(d_e, g, DIM, c4));
                    stg4(d_e, g, DIM, c4, v);
                }
            }
        } else if (AHEAD && r1 < re) {
            next_gather();                                     // (a chunk of edgeless nodes ran no GEMMs to issue it behind)
        }
        APROBE(7);
        __syncthreads();
        APROBE_WG(1);
        cur = nxt;
    }
}

// ================================================================ backward + its two weight gradients, one kernel (round 5)
// The weight gradients of the global edge step, dW_e = dz^T e and dW_ea = dea^T e (E_g rows each), were two jobs of the
// split-K launches of wgrad.hip: four [E_g, 128] streams read again (dz, dea, e twice) that this kernel had on chip a moment
// earlier, plus the d ea rows written for that purpose only -- 1.45 ms of the 8.3 ms PDBbind step.  Here every workgroup
// also forms its rows' share of both products and leaves two 128 x 128 partial tiles (+ the column sums of dz = the bias
// gradient) in the slot format of wgrad_core.h; the fixed-order reduction over the workgroups rides in the next
// weight-gradient launch (pamnet_wgrad_edge_enqueue_f32).  Per layer: + one read of the e rows, - the d ea write, - the
// four operand streams of the weight-gradient jobs.
//
// What it takes on a CU:
//   * the row index is the MFMA k dimension of these products: chunks of 32 rows = one k-step of v_mfma_f32_16x16x32_bf16;
//     wave w owns rows [16 w, 16 w + 16) of both results (2 x 8 accumulator tiles = 64 registers).
//   * operands as "8 consecutive rows of one column" fragments.  The sweep thread owns a float4 of ONE row, so the piece
//     images are written row-major and read through ds_read_b64_tr_b16 (the hardware's 4 x 4 transposing read; semantics
//     checked on the device by tools/probes/tr16_probe.hip).  ONE image per operand serves both reads -- the dX GEMMs'
//     row fragments (ds_read_b128) and the weight gradients' column fragments:
//       image = [piece 3][channel tile 8] subtiles of KSUB bytes, a subtile = [32 rows][16 channels] bf16, row r at
//       position kpos(r) = r with bits 2 and 3 swapped, 32 bytes per row.
//     transposing read (two 32-lane groups): the rows {8 g + j} of two k-groups g sit at 8 distinct positions mod 8 ->
//     256 distinct bytes; ds_read_b128 (the guide's four 16-lane groups: rows {0-3, 12-15} of one k-group with rows {4-11}
//     of the next, i.e. the two 16-byte halves of a row): 8 distinct positions mod 8 per half; the sweep's ds_write_b64
//     (16 lanes = one row, 16 float4 columns = 4 subtiles x 4 x 8 bytes): KSUB = 1024 + 32 puts the four subtiles 8 banks
//     apart.  All three conflict-free by the guide's LDS table.
//   * registers: the resident weight pieces (96) + 64 accumulators do not fit beside the sweeps.  The THIRD piece of both
//     weight slices -- used by one of the six products -- waits in LDS (64 KB, wave-private lane-linear images), the other
//     two stay in registers (64).
// d z, d e, d P_i: same arithmetic in the same order as global_edge_agg_bwd_kernel (bitwise the same values).
constexpr int KSUB = 1024 + 32;
constexpr int KIMG = 3 * 8 * KSUB;            // bytes of one operand's image: 32 rows x 128 channels x 3 pieces (+ padding)
__device__ __forceinline__ int kpos(int r) { return (r & 19) | ((r & 8) >> 1) | ((r & 4) << 1); }
// row r (0..31), float4 column c4 -> 8 bytes in each piece plane
__device__ __forceinline__ void st_kpieces4(char* __restrict__ I, int r, int c4, const float4& v) {
    char* d = I + (c4 >> 2) * KSUB + kpos(r) * 32 + (c4 & 3) * 8;
    uint32_t a0, b0, c0, a1, b1, c1;
    split3(v.x, v.y, a0, b0, c0);
    split3(v.z, v.w, a1, b1, c1);
    *reinterpret_cast<uint2*>(d) = make_uint2(a0, a1);
    *reinterpret_cast<uint2*>(d + 8 * KSUB) = make_uint2(b0, b1);
    *reinterpret_cast<uint2*>(d + 16 * KSUB) = make_uint2(c0, c1);
}
// row fragment (rows 16 m + (l & 15), channels 32 q + 8 (l >> 4) + 0..7): the A operand of the dX GEMMs
__device__ __forceinline__ Frag3 lds_kfrag_row(const char* __restrict__ I, int m, int q) {
    const int lane = threadIdx.x & 63, kg = lane >> 4;
    const char* s = I + (2 * q + (kg >> 1)) * KSUB + (kpos(lane & 15) + 16 * m) * 32 + (kg & 1) * 16;
    Frag3 f;
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) {
        const uint4 u = *reinterpret_cast<const uint4*>(s + pc * 8 * KSUB);
        f.p[pc][0] = u.x, f.p[pc][1] = u.y, f.p[pc][2] = u.z, f.p[pc][3] = u.w;
    }
    return f;
}
typedef short i16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x2 lds_tr16(const char* p) {
    typedef i16x4 __attribute__((address_space(3))) * lds_ptr;
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p)));
}
// column fragment (channel 16 ct + (l & 15), rows 8 (l >> 4) + 0..7): both operands of the weight-gradient products.
// Lane 4 j + c of a 16-lane group hands the transposing read the 8 bytes (row 8 g + j, channels 4 c .. 4 c + 3) and gets
// rows 8 g + 0..3 of its own channel back; rows + 4 sit 8 positions = 256 bytes further.
__device__ __forceinline__ Frag3 lds_kfrag_col(const char* __restrict__ I, int ct) {
    const int lane = threadIdx.x & 63;
    const int r = 8 * (lane >> 4) + ((lane & 15) >> 2);
    const char* s = I + ct * KSUB + kpos(r) * 32 + (lane & 3) * 8;
    Frag3 f;
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) {
        const u32x2 lo = lds_tr16(s + pc * 8 * KSUB), hi = lds_tr16(s + pc * 8 * KSUB + 256);
        f.p[pc][0] = lo[0], f.p[pc][1] = lo[1], f.p[pc][2] = hi[0], f.p[pc][3] = hi[1];
    }
    return f;
}
// a wave's transposed weight slice: pieces 0 and 1 in registers, piece 2 in its own lane-linear LDS image ([q][lane] x 16 B)
struct WFragB2 {
    uint32_t p[DIM / 32][2][4];
};
__device__ __forceinline__ void load_wfragb2_t(WFragB2& f, char* __restrict__ w2, const float* __restrict__ W, int ldw,
                                               int wc) {
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int q = 0; q < DIM / 32; ++q) {
        float v[8];
        const float* wp = W + (size_t)(32 * q + 8 * kg) * ldw + wc + j;
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = wp[(size_t)t * ldw];
        const Frag3 fr = split_frag(v);
#pragma unroll
        for (int t = 0; t < 4; ++t) f.p[q][0][t] = fr.p[0][t], f.p[q][1][t] = fr.p[1][t];
        *reinterpret_cast<uint4*>(w2 + (q * 64 + lane) * 16) = make_uint4(fr.p[2][0], fr.p[2][1], fr.p[2][2], fr.p[2][3]);
    }
}
// acc[0 .. G) += I(tiles 0 .. G) * slice: the six piece products in the order of mfma6 (edge_core.h)
template <int G>
__device__ __forceinline__ void mma_k16(const char* __restrict__ I, const WFragB2& f, const char* __restrict__ w2,
                                        Acc<2>& acc) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int q = 0; q < DIM / 32; ++q) {
        Frag3 a[G];
#pragma unroll
        for (int g = 0; g < G; ++g) a[g] = lds_kfrag_row(I, g, q);
        const uint4 u = *reinterpret_cast<const uint4*>(w2 + (q * 64 + lane) * 16);
        const uint32_t b2[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int g = 0; g < G; ++g) acc.v[g] = mfma_bf16(a[g].p[2], f.p[q][0], acc.v[g]);
#pragma unroll
        for (int g = 0; g < G; ++g) acc.v[g] = mfma_bf16(a[g].p[1], f.p[q][1], acc.v[g]);
#pragma unroll
        for (int g = 0; g < G; ++g) acc.v[g] = mfma_bf16(a[g].p[0], b2, acc.v[g]);
#pragma unroll
        for (int g = 0; g < G; ++g) acc.v[g] = mfma_bf16(a[g].p[1], f.p[q][0], acc.v[g]);
#pragma unroll
        for (int g = 0; g < G; ++g) acc.v[g] = mfma_bf16(a[g].p[0], f.p[q][1], acc.v[g]);
#pragma unroll
        for (int g = 0; g < G; ++g) acc.v[g] = mfma_bf16(a[g].p[0], f.p[q][0], acc.v[g]);
    }
}
// acc += a^T b over the chunk's 32 rows: the six piece products, small ones first
__device__ __forceinline__ void mfma6_col(const Frag3& a, const Frag3& b, f32x4& acc) {
    acc = mfma_bf16(a.p[2], b.p[0], acc);
    acc = mfma_bf16(a.p[1], b.p[1], acc);
    acc = mfma_bf16(a.p[0], b.p[2], acc);
    acc = mfma_bf16(a.p[1], b.p[0], acc);
    acc = mfma_bf16(a.p[0], b.p[1], acc);
    acc = mfma_bf16(a.p[0], b.p[0], acc);
}

constexpr int64_t WSLOT = DIM * DIM + 2 * DIM;      // floats per partial slot (wgrad_core.h: the tile, two bias row parts)

struct GAggBwdW {
    const float *d_agg, *z, *ea, *e, *We, *Wea;
    const int32_t *ptr, *row_of, *cuts;
    float *dz, *d_e, *dPi, *partial;
    int64_t m, n;
    int ld_we, ld_wea, accumulate;
};

template <bool ACC>
__global__ __launch_bounds__(WG8, 1) void global_edge_agg_bwd_wg_kernel(GAggBwdW a) {
    constexpr int MTX = 2, CAP = 32;
    constexpr int FT = 16 * LDT * 4;
    constexpr int W2B = 4 * 64 * 16;                           // a wave's third-piece image of one matrix
    __shared__ __attribute__((aligned(16))) char ldsb[3 * KIMG + MTX * FT + 2 * 8 * W2B];
    __shared__ int sptr[NMAX + 1];
    __shared__ float4 carry[2][32];
    char* Iz = ldsb;                                           // d z, d ea, e: piece images of the chunk
    char* Ia = ldsb + KIMG;
    char* Ie = ldsb + 2 * KIMG;
    float* S0 = reinterpret_cast<float*>(ldsb + 3 * KIMG);     // the d e accumulators on their way to rows
    const float* __restrict__ d_agg = a.d_agg;
    const float* __restrict__ zs = a.z;
    const float* __restrict__ eas = a.ea;
    const float* __restrict__ es = a.e;
    const int32_t* __restrict__ ptr = a.ptr;
    const int32_t* __restrict__ row_of = a.row_of;
    float* __restrict__ dz = a.dz;
    float* __restrict__ d_e = a.d_e;
    const int wv = threadIdx.x >> 6, wc = wave_col<8>();
    char* w2e = ldsb + 3 * KIMG + MTX * FT + wv * W2B;
    char* w2a = w2e + 8 * W2B;
    WFragB2 f1, f2;
    load_wfragb2_t(f1, w2e, a.We, a.ld_we, wc);
    load_wfragb2_t(f2, w2a, a.Wea, a.ld_wea, wc);
    const int wg = xcd_order(blockIdx.x, gridDim.x);
    const int nb = a.cuts ? a.cuts[wg] : seg_cut(ptr, row_of, a.n, a.m, wg, gridDim.x);
    const int ne = a.cuts ? a.cuts[wg + 1] : seg_cut(ptr, row_of, a.n, a.m, wg + 1, gridDim.x);
    const int64_t re = ptr[ne];
    constexpr int RPP = 16, NI = MTX;
    const int c4 = threadIdx.x & 31, rr = threadIdx.x >> 5;
    f32x4 accE[8], accA[8];                                    // rows [16 wv, 16 wv + 16) of dW_e / dW_ea
#pragma unroll
    for (int b = 0; b < 8; ++b) accE[b] = accA[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;                                          // column sums of d z (the bias gradient): this lane's channel, its node rows
    // Memory operations retire in order, loads AND stores on one counter (vmcnt), and the compiler can only wait for "all but
    // the N youngest": wherever N is not a compile-time constant -- a request under a branch, a loop of stores -- it waits for
    // everything.  The first form of this loop (dependent vector loads in the plan, the d_agg gather through row_of[g],
    // requests under `if (row < r1)`) drained the queue five times per chunk, each time behind the previous chunk's stores
    // and this chunk's prefetched rows: a 20 600-cycle loop period for 15 000 cycles of work
    // (tools/edge_wgrad_phase_probe.py).  This form issues the SAME vector-memory instructions in every iteration:
    //   * the plan on the scalar unit (s_load: its own counter), made one chunk ahead;
    //   * every load from a clamped address, every store of a row past the chunk's end to a dummy line (`dump`);
    //   * ptr[c0 + t] of the NEXT chunk's nodes first thing in a chunk, used a whole chunk later;
    //   * the rows (z, ea, e) and the row indices of the next chunk requested before the GEMMs, the gather of its d_agg
    //     rows behind them -- ahead of the final sweep's stores, so that the next sweep's wait for it is not a wait for those;
    //   * the accumulate operand requested ahead of all of them: the final sweep's wait for it covers nothing younger;
    //   * chunks of exactly 32 rows -- all but a workgroup's last -- run a body without row masks (FULL).
    // Addresses: one 32-bit byte offset per row and thread on top of the (scalar) stream bases -- the host side keeps this
    // kernel to n_edges * 512 < 2^32.
    int par = 0;
    float* dump = a.partial + 2 * (int64_t)gridDim.x * WSLOT + 4 * threadIdx.x;   // 8 KB behind the slots: writes nobody reads
    float4 pz[NI], pe[NI], px[NI], pdm[NI];                    // z, ea, e rows and gathered d_agg rows of the next chunk
    int pri[NI];                                               // ... and the target node of each (the gather's index)
    const int rlast = re > 0 ? (int)re - 1 : 0;
    auto at = [](const float* base, uint32_t byte_off) {
        return reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + byte_off);
    };
    auto at_w = [](float* base, uint32_t byte_off) { return reinterpret_cast<float4*>(reinterpret_cast<char*>(base) + byte_off); };
    const uint32_t coff = 16u * c4;
    auto prefetch = [&](int rb) {                              // (clamped: rows past the workgroup's end re-read its last row)
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int g = rb + rr + RPP * i;
            g = g < rlast ? g : rlast;
            const uint32_t off = (uint32_t)g * 512u + coff;
            pz[i] = *at(zs, off);
            pe[i] = *at(eas, off);
            px[i] = *at(es, off);
            pri[i] = row_of[g];
        }
    };
    auto gather = [&]() {
#pragma unroll
        for (int i = 0; i < NI; ++i) pdm[i] = *at(d_agg, (uint32_t)pri[i] * 512u + coff);
    };
    // The plan's loads on the scalar unit.  ptr / row_of are never written by this kernel, but the compiler cannot know --
    // behind the first store of the loop it falls back to vector loads -- hence by hand.  Addresses are wave-uniform
    // (readfirstlane where a value passed through a vector register).
    auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    struct Plan {
        int c0, c1, rows, r0, r1;
    };
    // chunk starting at row r0_ whose node is c0_; returns in `ahead` the node of the row behind the chunk (-1 at the end of
    // the workgroup's rows).  Nodes without edges are never a chunk's business: their d P_i rows are zero-filled up front.
    const int ire = (int)re;
    auto plan_at = [&](int c0_, int r0_, int& ahead) {
        Plan p;
        p.c0 = c0_, p.r0 = r0_, p.c1 = c0_ - 1, p.r1 = r0_;
        ahead = -1;
        if (r0_ < ire) {
            int r1_ = r0_ + CAP < ire ? r0_ + CAP : ire;
            int last = 0, nxt = 0;
            sload2(row_of + r1_ - 1, row_of + (r1_ < ire ? r1_ : ire - 1), last, nxt);   // node of the last row, of the row behind
            int c1_ = last;
            if (c1_ - c0_ + 1 > NMAX) {                        // a long run of nodes without edges inside the chunk: cut there
                c1_ = c0_ + NMAX - 1;
                int e1 = 0, dummy = 0;
                sload2(ptr + c1_ + 1, ptr + c1_ + 1, e1, dummy);
                r1_ = e1;                                      // (>= the rows of node c0_: never empty)
                if (r1_ < ire) sload2(row_of + r1_, ptr + c1_ + 1, nxt, dummy);
            }
            p.c1 = c1_, p.r1 = r1_;
            ahead = r1_ < ire ? nxt : -1;
        }
        p.rows = p.r1 - p.r0;
        return p;
    };
    // d P_i of the workgroup's nodes: zeros first (nodes without edges keep them); acknowledged before the loop stores sums
    for (int64_t k = (int64_t)nb * (DIM / 4) + threadIdx.x; k < (int64_t)ne * (DIM / 4); k += WG8)
        reinterpret_cast<float4*>(a.dPi)[k] = f4zero();
    const int rbeg = uni(ptr[nb]);
    int ahead = -1, first = 0;
    if (rbeg < ire) sload2(row_of + rbeg, row_of + rbeg, first, ahead);
    Plan cur = plan_at(first, rbeg, ahead);
    int myp = load_node_ptr(ptr, cur.c0, a.n);                 // ptr[c0 + t] of the current chunk's nodes (raw)
    APROBE_WG(0);
    APROBE(0);
    if (rbeg < ire) {
        prefetch(cur.r0);
        gather();
    }
    __builtin_amdgcn_s_waitcnt(0x0070);                        // vmcnt(0)  (expcnt 7, lgkmcnt 15 untouched)
    __syncthreads();
    auto body = [&](auto full_c) {
        constexpr bool FULL = decltype(full_c)::value;
        APROBE(1);
        const int c0 = cur.c0, c1 = cur.c1, rows = cur.rows, r0 = cur.r0, r1 = cur.r1;
        const int nn = c1 - c0 + 1;
        // the next chunk starts at the node of the row behind this one: node c1 itself if it goes on
        const bool open_end = ahead == c1;
        const Plan nxt = plan_at(ahead, r1, ahead);
        const int mt = FULL ? 2 : (rows + 15) >> 4;
        if ((int)threadIdx.x <= nn) sptr[threadIdx.x] = myp - r0;
        myp = load_node_ptr(ptr, nxt.c0, a.n);                 // (raw value: subtracting here would wait for the load here)
        APROBE(2);
        // every row of the 32-row k-step is written: rows past the chunk's end as zeros in all three images
        float4 sx[NI], sy[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int r = rr + RPP * i;
            const bool ok = FULL || r < rows;
            st_kpieces4(Ie, r, c4, ok ? px[i] : f4zero());
            sx[i] = ok ? f4mul(f4mul(pdm[i], pe[i]), f4dsilu(pz[i])) : f4zero();     // d z = dm * ea * SiLU'(z)
            sy[i] = ok ? f4mul(pdm[i], f4silu(pz[i])) : f4zero();                    // d ea = dm * SiLU(z)
            st_kpieces4(Iz, r, c4, sx[i]);
            st_kpieces4(Ia, r, c4, sy[i]);
        }
        const uint32_t boff = (uint32_t)(r0 + rr) * 512u + coff;   // byte offset of this thread's first row
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            float4* dst = at_w(dz, boff + 512u * RPP * i);
            if (!FULL && rr + RPP * i >= rows) dst = reinterpret_cast<float4*>(dump);
            *dst = sx[i];
        }
        APROBE(3);
        float4 dacc[NI];
        if constexpr (ACC) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                uint32_t off = boff + 512u * RPP * i;
                if (!FULL && rr + RPP * i >= rows) off = (uint32_t)r0 * 512u + coff;
                dacc[i] = *at(d_e, off);
            }
        }
        prefetch(r1);                                          // the next chunk's rows travel during the GEMMs
        APROBE(4);
        __syncthreads();
        APROBE(5);
        const bool open_begin = sptr[0] < 0;
        // d P_i[node] = the sum of the node's d z rows -- on the matrix pipe: [16 nodes x 32 rows] of 0 / 1 (exact in bf16) times
        // this wave's column fragment of the d z pieces (the operand the weight-gradient k-step below loads anyway), three MFMAs
        // per 16 nodes and wave, every product exact, fp32 accumulation.  (As one serial chain per node on a 32-lane group --
        // reduce_nodes, the form of the plain kernel -- a complex's ~36-row nodes cost 1 800-3 800 cycles of a 16 000-cycle chunk
        // on the critical path, alone on the LDS or not.)  A node that spans chunks hands its running sum on through `carry`
        // (this wave's 16 channels, same wave next chunk).
        {
            const Frag3 az = lds_kfrag_col(Iz, wv);
            const int lane = threadIdx.x & 63, i16 = lane & 15, kg = lane >> 4;
            float* cin = reinterpret_cast<float*>(carry[par]) + wc;
            float* cout = reinterpret_cast<float*>(carry[par ^ 1]) + wc;
            auto node_tile = [&](int nt) {
                const int k = 16 * nt + i16;
                int b = rows, e = rows;                        // (beyond the chunk's nodes: no rows)
                if (k < nn) b = sptr[k], e = sptr[k + 1];
                uint32_t ind[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int q = 8 * kg + 2 * t;
                    ind[t] = ((q >= b && q < e) ? 0x3f80u : 0u) | ((q + 1 >= b && q + 1 < e) ? 0x3f800000u : 0u);
                }
                f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
                d = mfma_bf16(ind, az.p[2], d);
                d = mfma_bf16(ind, az.p[1], d);
                d = mfma_bf16(ind, az.p[0], d);
                bsum += (d[0] + d[1]) + (d[2] + d[3]);         // every row of the chunk belongs to one of its nodes
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k2 = 16 * nt + 4 * kg + r;
                    float v = d[r];
                    if (k2 == 0 && open_begin) v += cin[i16];
                    const bool keep = k2 == nn - 1 && open_end;
                    if (keep) cout[i16] = v;
                    // (one store per lane and r whatever the node count: rows that are not a finished node's go to the dump)
                    float* dst = (k2 < nn && !keep) ? a.dPi + (int64_t)(c0 + k2) * DIM + wc + i16 : dump;
                    *dst = v;
                }
            };
            node_tile(0);
            if (nn > 16) {                                     // (rare: a chunk with more than 16 nodes; the side path drains)
                for (int nt = 1; 16 * nt < nn; ++nt) node_tile(nt);
                __builtin_amdgcn_s_waitcnt(0x0070);
            }
        }
        par ^= 1;
        APROBE(6);
        // the weight gradients' k-step: rows of both results from this wave's 16 channels of d z / d ea against all eight
        // channel tiles of e
        {
            const Frag3 az = lds_kfrag_col(Iz, wv), aa = lds_kfrag_col(Ia, wv);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const Frag3 be = lds_kfrag_col(Ie, b);
                mfma6_col(az, be, accE[b]);
                mfma6_col(aa, be, accA[b]);
            }
        }
        APROBE(7);
        Acc<2> acc;
        acc.zero();
        if (mt == 2) {
            mma_k16<2>(Iz, f1, w2e, acc);
            mma_k16<2>(Ia, f2, w2a, acc);
        } else {
            mma_k16<1>(Iz, f1, w2e, acc);
            mma_k16<1>(Ia, f2, w2a, acc);
        }
        APROBE(8);
        gather();                                              // (the row indices arrived during the GEMMs)
        acc_store<2>(acc, S0, wc, 0.f, mt);                   // (S0 is this stage's alone: no barrier ahead of it)
        __syncthreads();
        APROBE(9);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int r = rr + RPP * i;
            float4 v = lds4(S0, r, c4);
            if constexpr (ACC) v = f4add(v, dacc[i]);
            float4* dst = at_w(d_e, boff + 512u * RPP * i);
            if (!FULL && r >= rows) dst = reinterpret_cast<float4*>(dump);
            *dst = v;
        }
        APROBE(10);
        // (no barrier at the end of a chunk: every wave reached the one above with its GEMMs done -- the images are free for the
        // next sweep --, S0 is next written behind the next sweep's barrier, sptr likewise, `carry` is wave-private)
        APROBE(11);
        cur = nxt;
    };
    while (cur.rows == CAP) body(std::true_type{});
    while (cur.rows > 0) body(std::false_type{});
    __builtin_amdgcn_s_waitcnt(0x0070);
    __syncthreads();                                           // (the last final sweep is done with S0)
    // ---- the workgroup's partial tiles in the slot format of wgrad_core.h: slot blockIdx.x of job 0 (dW_e, with the bias
    // parts), slot gridDim.x + blockIdx.x of job 1 (dW_ea).  Through LDS so that the tiles leave as 512-byte rows.
    float* T = reinterpret_cast<float*>(ldsb);                 // [128][LDT] over the (free) piece images
    const int lane = threadIdx.x & 63, r16 = lane & 15, kg = lane >> 4;
    float* slot0 = a.partial + (int64_t)blockIdx.x * WSLOT;
    float* slot1 = a.partial + ((int64_t)gridDim.x + blockIdx.x) * WSLOT;
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) T[(16 * wv + 4 * kg + r) * LDT + 16 * b + r16] = accE[b][r];
    S0[kg * DIM + wc + r16] = bsum;                            // [4 node-row groups][128]
    __syncthreads();
#pragma unroll
    for (int i = 0; i < DIM / 16; ++i) {
        const int row = rr + 16 * i;
        *reinterpret_cast<float4*>(slot0 + row * DIM + 4 * c4) = *reinterpret_cast<const float4*>(T + row * LDT + 4 * c4);
    }
    if (threadIdx.x < DIM) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) s += S0[k * DIM + threadIdx.x];    // fixed order
        slot0[DIM * DIM + threadIdx.x] = s;
        slot0[DIM * DIM + DIM + threadIdx.x] = 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) T[(16 * wv + 4 * kg + r) * LDT + 16 * b + r16] = accA[b][r];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < DIM / 16; ++i) {
        const int row = rr + 16 * i;
        *reinterpret_cast<float4*>(slot1 + row * DIM + 4 * c4) = *reinterpret_cast<const float4*>(T + row * LDT + 4 * c4);
    }
    APROBE(12);
    APROBE_WG(1);
}

// ------------------------------------------------------------------------------------------------ local aggregation
// Workgroup = 8 lane groups of 32; it owns NPW consecutive nodes and walks their local edges 8 at a time: group k
// computes v = q3[e] * (m_ji[e] + sum_r m_nb[idx[r]] * s[r]) for its edge (rows of e in CSR order), the node owners add
// the round's values in edge order.
constexpr int NPW = 4;
__global__ __launch_bounds__(256) void local_agg_fwd_kernel(const float4* __restrict__ m_ji, const float4* __restrict__ m_nb,
                                                            const float4* __restrict__ s, const float4* __restrict__ q3,
                                                            const int32_t* __restrict__ t_ptr,
                                                            const int32_t* __restrict__ t_col,
                                                            const int32_t* __restrict__ l_ptr,
                                                            const float4* __restrict__ init, float4* __restrict__ m_t,
                                                            float4* __restrict__ out, int64_t n) {
    __shared__ float4 val[8][32];
    const int grp = threadIdx.x >> 5, c = threadIdx.x & 31;
    const int64_t n0 = (int64_t)blockIdx.x * NPW;
    const int64_t n1 = n0 + NPW < n ? n0 + NPW : n;
    const int eb = l_ptr[n0], ee = l_ptr[n1];
    const int64_t node = n0 + grp;                           // groups 0..NPW-1 own a node each
    const bool owner = grp < NPW && node < n1;
    int nb0 = 0, nb1 = 0;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (owner) {
        nb0 = l_ptr[node], nb1 = l_ptr[node + 1];
        if (init) acc = init[node * 32 + c];
    }
    for (int e0 = eb; e0 < ee || e0 == eb; e0 += 8) {
        const int e = e0 + grp;
        if (e < ee) {
            const int t0 = t_ptr[e], t1 = t_ptr[e + 1];
            float4 v = m_ji[(int64_t)e * 32 + c];
            const float4 gate = q3[(int64_t)e * 32 + c];
            int t = t0;
            for (; t + 4 <= t1; t += 4) {
                const int64_t k0 = t_col[t], k1 = t_col[t + 1], k2 = t_col[t + 2], k3 = t_col[t + 3];
                const float4 a0 = m_nb[k0 * 32 + c], a1 = m_nb[k1 * 32 + c], a2 = m_nb[k2 * 32 + c], a3 = m_nb[k3 * 32 + c];
                const float4 b0 = s[(int64_t)t * 32 + c], b1 = s[(int64_t)(t + 1) * 32 + c],
                             b2 = s[(int64_t)(t + 2) * 32 + c], b3 = s[(int64_t)(t + 3) * 32 + c];
                v = pamnet::f4add(v, pamnet::f4mul(a0, b0));
                v = pamnet::f4add(v, pamnet::f4mul(a1, b1));
                v = pamnet::f4add(v, pamnet::f4mul(a2, b2));
                v = pamnet::f4add(v, pamnet::f4mul(a3, b3));
            }
            for (; t < t1; ++t)
                v = pamnet::f4add(v, pamnet::f4mul(m_nb[(int64_t)t_col[t] * 32 + c], s[(int64_t)t * 32 + c]));
            if (m_t) pamnet::st_nt4(m_t + (int64_t)e * 32 + c, v);   // backward-only save (streamed)
            val[grp][c] = pamnet::f4mul(v, gate);
        }
        __syncthreads();
        if (owner) {
            const int lo = nb0 > e0 ? nb0 : e0, hi = nb1 < e0 + 8 ? nb1 : e0 + 8;
            for (int q = lo; q < hi; ++q) acc = pamnet::f4add(acc, val[q - e0][c]);
        }
        __syncthreads();
        if (e0 + 8 >= ee) break;
    }
    if (owner) out[node * 32 + c] = acc;
}

// Backward of local_agg_fwd up to the inputs of the MLP kernels, one launch (blockIdx.y selects the direction):
//   y = 0, one lane group per edge e (target i):  d m_t[e] = d x2[i] * q3[e],  d q3[e] = d x2[i] * m_t[e],
//          d s[r] = m_nb[idx[r]] * d m_t[e] for the rows r of e;
//   y = 1, one lane group per edge e' walking the transposed row list:  d m_nb[e'] = sum_{r: idx[r] = e'} s[r] * d m_t[edge(r)]
//          with d m_t recomputed from d x2 and q3 (no dependency on the y = 0 half), rows in transposed-CSR order.
__global__ __launch_bounds__(256) void local_agg_bwd_kernel(const float4* __restrict__ d_x2, const int32_t* __restrict__ l_row,
                                                            const float4* __restrict__ q3, const float4* __restrict__ m_t,
                                                            const float4* __restrict__ m_nb, const float4* __restrict__ s,
                                                            const int32_t* __restrict__ t_ptr,
                                                            const int32_t* __restrict__ t_col,
                                                            const int32_t* __restrict__ t_row,
                                                            const int32_t* __restrict__ tT_ptr,
                                                            const int32_t* __restrict__ tT_perm,
                                                            const int32_t* __restrict__ tT_edge,
                                                            const int32_t* __restrict__ tT_node, int64_t el,
                                                            float4* __restrict__ d_mt, float4* __restrict__ d_q3,
                                                            float4* __restrict__ d_s, float4* __restrict__ d_mnb) {
    const int c = threadIdx.x & 31;
    const int64_t e = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
    if (e >= el) return;
    if (blockIdx.y == 0) {
        const float4 dx = d_x2[(int64_t)l_row[e] * 32 + c];
        const float4 g = pamnet::f4mul(dx, q3[e * 32 + c]);
        d_mt[e * 32 + c] = g;
        d_q3[e * 32 + c] = pamnet::f4mul(dx, m_t[e * 32 + c]);
        const int t0 = t_ptr[e], t1 = t_ptr[e + 1];
        // four rows per step, the last step predicated (as a row-by-row tail every leftover row was two dependent round trips)
        for (int t = t0; t < t1; t += 4) {
            int64_t k[4];
            float4 a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) k[u] = t + u < t1 ? t_col[t + u] : -1;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (k[u] >= 0) a[u] = m_nb[k[u] * 32 + c];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (k[u] >= 0) d_s[(int64_t)(t + u) * 32 + c] = pamnet::f4mul(a[u], g);
        }
    } else {
        const int q0 = tT_ptr[e], q1 = tT_ptr[e + 1];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        int q = q0;
        if (tT_edge) {
            // tT_edge[q] = t_row[tT_perm[q]] (the target edge of the row that gathers e), tT_node[q] = l_row[tT_edge[q]] (its
            // node), made with the graph (pamnet_triplet_transpose_aux_i32): the three index reads of a term are independent --
            // one level of indirection ahead of the data instead of three (perm -> t_row -> l_row -> d x2); four terms in
            // flight.  Same terms in the same order.
            for (; q < q1; q += 4) {                           // (the last step predicated: no row-by-row tail)
                float4 sv[4], dv[4], qv[4];
                int64_t ta[4], ra[4], nd[4];
                bool ok[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    ok[u] = q + u < q1;
                    if (ok[u]) ta[u] = tT_perm[q + u], ra[u] = tT_edge[q + u], nd[u] = tT_node[q + u];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (ok[u]) sv[u] = s[ta[u] * 32 + c], dv[u] = d_x2[nd[u] * 32 + c], qv[u] = q3[ra[u] * 32 + c];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (ok[u]) v = pamnet::f4add(v, pamnet::f4mul(sv[u], pamnet::f4mul(dv[u], qv[u])));
            }
        } else {
            for (; q + 2 <= q1; q += 2) {
                const int64_t ta = tT_perm[q], tb = tT_perm[q + 1];
                const int64_t ra = t_row[ta], rb = t_row[tb];
                const float4 sa = s[ta * 32 + c], sb = s[tb * 32 + c];
                const float4 ga = pamnet::f4mul(d_x2[(int64_t)l_row[ra] * 32 + c], q3[ra * 32 + c]);
                const float4 gb = pamnet::f4mul(d_x2[(int64_t)l_row[rb] * 32 + c], q3[rb * 32 + c]);
                v = pamnet::f4add(v, pamnet::f4mul(sa, ga));
                v = pamnet::f4add(v, pamnet::f4mul(sb, gb));
            }
            if (q < q1) {
                const int64_t ta = tT_perm[q];
                const int64_t ra = t_row[ta];
                v = pamnet::f4add(v, pamnet::f4mul(s[ta * 32 + c], pamnet::f4mul(d_x2[(int64_t)l_row[ra] * 32 + c], q3[ra * 32 + c])));
            }
        }
        d_mnb[e * 32 + c] = v;
    }
}

inline int pick_mtx(int64_t m, int64_t grid) {
    const int64_t per = ceil_div(m, grid);
    // 9 tiles: one chunk per workgroup when ~E/256 rows give or take half a node degree fit 144 rows; beyond that the
    // workgroups walk several balanced chunks of <= 128 rows
    return per <= 40 ? 3 : (per <= 72 ? 5 : (per <= 132 ? 9 : 8));
}
// PAMNET_AGG_PIECES=0: the multi-chunk kernels without piece planes (every wave splits its own A fragments) -- the form
// before round 4's issue-slot measurement, kept for A/B timing (read once).
inline bool agg_pieces() {
    static bool v = [] { const char* e = getenv("PAMNET_AGG_PIECES"); return !e || atoi(e) != 0; }();
    return v;
}
inline int64_t agg_grid(int64_t m) {
    const int64_t g = ceil_div(m, 16);
    return g < 1 ? 1 : (g > N_CU ? N_CU : g);
}
// Round 6: the forward takes the ping-pong form (global_edge_agg_fwd_pp_kernel) where a workgroup streams enough 32-row groups
// to fill its pipeline.  PAMNET_AGG_PP=0 / 1 forces the chunked / the ping-pong form (A/B runs).
inline bool agg_pp(int64_t n_edges) {
    const char* e = getenv("PAMNET_AGG_PP");            // (per call: a test flips it inside one process)
    if (e && e[0]) return atoi(e) != 0;
    return n_edges >= 256 * 512;
}

}  // namespace

// The ping-pong form of pamnet_global_edge_agg_fwd_f32 (same arguments, same results bit for bit), whatever the size.
extern "C" int pamnet_global_edge_agg_fwd_pp_f32(const float* e, int64_t n_edges, int64_t n_nodes, const float* We,
                                                 int64_t ld_we, const float* bm, const float* Wea, int64_t ld_wea,
                                                 const float* Pi, const float* Pj, const int32_t* ptr, const int32_t* row_of,
                                                 const int32_t* col, const int32_t* cuts, const float* init, float* z, float* ea,
                                                 float* out, pamnet_stream_t stream) {
    if (n_edges < 0 || n_nodes < 0) return PAMNET_EINVAL;
    if ((ld_we == 0) != (ld_wea == 0) || (ld_we != 0 && (ld_we < DIM || ld_wea < DIM))) return PAMNET_EINVAL;   // 0, 0: images
    if (n_nodes == 0) return PAMNET_OK;
    if (!We || !bm || !Wea || !Pi || !Pj || !ptr || !out) return PAMNET_ENULL;      // init, z, ea: optional
    if (n_edges > 0 && (!e || !row_of || !col)) return PAMNET_ENULL;
    if ((z == nullptr) != (ea == nullptr)) return PAMNET_EINVAL;               // both saves or none
    GAggFwd a{e, We, bm, Wea, Pi, Pj, init, ptr, row_of, col, cuts, z, ea, out, n_edges, n_nodes, (int)ld_we, (int)ld_wea};
    const int64_t grid = agg_grid(n_edges);
    hipStream_t st = as_stream(stream);
    if (z) hipLaunchKernelGGL((global_edge_agg_fwd_pp_kernel<true>), dim3((unsigned)grid), dim3(WG8), 0, st, a);
    else hipLaunchKernelGGL((global_edge_agg_fwd_pp_kernel<false>), dim3((unsigned)grid), dim3(WG8), 0, st, a);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

// out[i] = init[i] + sum_{e -> i} SiLU(W_e e + b_m + Pi[i] + Pj[col[e]]) * (W_ea e)   for every node i < n_nodes
// (nodes without edges get init).  z, ea: optional saves for the backward.  ptr [n_nodes + 1] / row_of / col: CSR by target.
extern "C" int pamnet_global_edge_agg_fwd_f32(const float* e, int64_t n_edges, int64_t n_nodes, const float* We,
                                              int64_t ld_we, const float* bm, const float* Wea, int64_t ld_wea,
                                              const float* Pi, const float* Pj, const int32_t* ptr, const int32_t* row_of,
                                              const int32_t* col, const int32_t* cuts, const float* init, float* z, float* ea,
                                              float* out, pamnet_stream_t stream) {
    if (n_edges < 0 || n_nodes < 0) return PAMNET_EINVAL;
    if ((ld_we == 0) != (ld_wea == 0) || (ld_we != 0 && (ld_we < DIM || ld_wea < DIM))) return PAMNET_EINVAL;   // 0, 0: images
    if (n_nodes == 0) return PAMNET_OK;
    if (!We || !bm || !Wea || !Pi || !Pj || !ptr || !out) return PAMNET_ENULL;      // init, z, ea: optional
    if (n_edges > 0 && (!e || !row_of || !col)) return PAMNET_ENULL;
    if ((z == nullptr) != (ea == nullptr)) return PAMNET_EINVAL;               // both saves or none
    if (agg_pp(n_edges))
        return pamnet_global_edge_agg_fwd_pp_f32(e, n_edges, n_nodes, We, ld_we, bm, Wea, ld_wea, Pi, Pj, ptr, row_of, col, cuts,
                                                 init, z, ea, out, stream);
    GAggFwd a{e, We, bm, Wea, Pi, Pj, init, ptr, row_of, col, cuts, z, ea, out, n_edges, n_nodes, (int)ld_we, (int)ld_wea};
    const int64_t grid = agg_grid(n_edges);
    hipStream_t st = as_stream(stream);
    const bool save = z != nullptr;
#define PAMNET_AGG_FWD(MTX, PRE)                                                                                         \
    do {                                                                                                                 \
        if (save) hipLaunchKernelGGL((global_edge_agg_fwd_kernel<MTX, PRE, true>), dim3((unsigned)grid), dim3(WG8), 0, st, a); \
        else hipLaunchKernelGGL((global_edge_agg_fwd_kernel<MTX, PRE, false>), dim3((unsigned)grid), dim3(WG8), 0, st, a);    \
    } while (0)
    if (agg_pieces()) {
        // Piece-plane form, 7-tile chunks with the next chunk's rows in flight, for every size: at the PDBbind shape 381 us
        // in training / 323 in inference against 426 / 377 for the reader-side split; at the QM9 batch (one and a bit
        // chunks per workgroup) 26.0 / 22.6 against 28.9 / 26.8 for the 9-tile single-chunk form; B = 16 .. 64 likewise.
        if (save) hipLaunchKernelGGL((global_edge_agg_fwd_kernel<7, true, true, true>), dim3((unsigned)grid), dim3(WG8), 0, st, a);
        else hipLaunchKernelGGL((global_edge_agg_fwd_kernel<7, true, false, true>), dim3((unsigned)grid), dim3(WG8), 0, st, a);
    } else {
        switch (pick_mtx(n_edges, grid)) {
            case 3: PAMNET_AGG_FWD(3, true); break;
            case 5: PAMNET_AGG_FWD(5, true); break;
            case 9: PAMNET_AGG_FWD(9, false); break;
            default: PAMNET_AGG_FWD(8, true); break;  // (5-tile chunks measured the same in training, 4 % slower in inference)
        }
    }
#undef PAMNET_AGG_FWD
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

// cuts[k] = node boundary nearest to edge row k * n_edges / G for k = 0 .. G, G = the grid of the fused kernels for this
// edge count (*grid_out): the node-aligned work split, computed once per graph instead of by every workgroup of every
// launch (two dependent loads ahead of everything else).  cuts: G + 1 ints (<= 257).
__global__ __launch_bounds__(256) void seg_cuts_kernel(const int32_t* __restrict__ ptr, const int32_t* __restrict__ row_of,
                                                       int64_t n, int64_t m, int G, int32_t* __restrict__ cuts) {
    for (int k = threadIdx.x; k <= G; k += 256) cuts[k] = seg_cut(ptr, row_of, n, m, k, G);
}
extern "C" int pamnet_seg_cuts_i32(const int32_t* ptr, const int32_t* row_of, int64_t n_nodes, int64_t n_edges,
                                   int32_t* cuts, int64_t* grid_out, pamnet_stream_t stream) {
    if (n_nodes < 0 || n_edges < 0) return PAMNET_EINVAL;
    if (!ptr || !cuts || (n_edges > 0 && !row_of)) return PAMNET_ENULL;
    const int64_t grid = agg_grid(n_edges);
    if (grid_out) *grid_out = grid;
    hipLaunchKernelGGL(seg_cuts_kernel, dim3(1), dim3(256), 0, as_stream(stream), ptr, row_of, n_nodes, n_edges, (int)grid, cuts);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

// Backward of the fused global edge step: dz, dea [E,128] written (the weight gradients and the source-side reduction
// read them), d_e written / accumulated, dPi[i] = sum_{e -> i} dz[e] for every node (zero for nodes without edges).
extern "C" int pamnet_global_edge_agg_bwd_f32(const float* d_agg, int64_t n_edges, int64_t n_nodes, const int32_t* ptr,
                                              const int32_t* row_of, const int32_t* cuts, const float* z, const float* ea,
                                              const float* We,
                                              int64_t ld_we, const float* Wea, int64_t ld_wea, float* dz, float* dea,
                                              float* d_e, int32_t accumulate, float* dPi, pamnet_stream_t stream) {
    if (n_edges < 0 || n_nodes < 0) return PAMNET_EINVAL;
    if ((ld_we == 0) != (ld_wea == 0) || (ld_we != 0 && (ld_we < DIM || ld_wea < DIM))) return PAMNET_EINVAL;   // 0, 0: images
    if (n_nodes == 0) return PAMNET_OK;
    if (!d_agg || !ptr || !We || !Wea || !dPi) return PAMNET_ENULL;
    if (n_edges > 0 && (!row_of || !z || !ea || !dz || !dea || !d_e)) return PAMNET_ENULL;
    GAggBwd a{d_agg, z, ea, We, Wea, ptr, row_of, cuts, dz, dea, d_e, dPi, n_edges, n_nodes, (int)ld_we, (int)ld_wea,
              (int)accumulate};
    const int64_t grid = agg_grid(n_edges);
    hipStream_t st = as_stream(stream);
    if (agg_pieces()) {
        // Piece-plane form, 3-tile chunks with the next chunk's rows in flight, for every size: PDBbind shape 490 us against
        // 604 (+ the transposed segment sum: 591 / 704); QM9 batch 34.5 against 42.2 for the 9-tile single-chunk form.
        hipLaunchKernelGGL((global_edge_agg_bwd_kernel<3, true, true>), dim3((unsigned)grid), dim3(WG8), 0, st, a);
    } else {
        switch (pick_mtx(n_edges, grid)) {
            case 3: hipLaunchKernelGGL(global_edge_agg_bwd_kernel<3>, dim3((unsigned)grid), dim3(WG8), 0, st, a); break;
            case 5: hipLaunchKernelGGL(global_edge_agg_bwd_kernel<5>, dim3((unsigned)grid), dim3(WG8), 0, st, a); break;
            case 9: hipLaunchKernelGGL(global_edge_agg_bwd_kernel<9>, dim3((unsigned)grid), dim3(WG8), 0, st, a); break;
            default:  // several chunks per workgroup: 3-tile chunks with the next chunk's rows in flight (712 us at the PDBbind
                      // shape against 803 for 8-tile chunks without, 726 for 4-tile chunks with the prefetch)
                hipLaunchKernelGGL((global_edge_agg_bwd_kernel<3, true>), dim3((unsigned)grid), dim3(WG8), 0, st, a);
                break;
        }
    }
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

// The same backward with the two weight gradients of the step formed in the kernel (global_edge_agg_bwd_wg_kernel): d ea is
// not written; `partial` receives 2 * G slots of 128 * 128 + 256 floats (G = *slots of pamnet_global_edge_agg_wg_floats):
// slots [0, G) = the workgroups' shares of dW_e = dz^T e with the column sums of dz (the bias gradient), [G, 2 G) = those of
// dW_ea = dea^T e.  pamnet_wgrad_edge_enqueue_f32 hands them to a deferred weight-gradient context for the fixed-order sum.
extern "C" int pamnet_global_edge_agg_wg_floats(int64_t n_edges, int64_t* floats, int64_t* slots) {
    if (n_edges < 0 || !floats) return PAMNET_EINVAL;
    const int64_t g = agg_grid(n_edges);
    *floats = 2 * g * WSLOT + 4 * WG8;      // + a dump line behind the slots (stores of rows past a chunk's end)
    if (slots) *slots = g;
    return PAMNET_OK;
}
extern "C" int pamnet_global_edge_agg_bwd_wg_f32(const float* d_agg, int64_t n_edges, int64_t n_nodes, const int32_t* ptr,
                                                 const int32_t* row_of, const int32_t* cuts, const float* z, const float* ea,
                                                 const float* e, const float* We, int64_t ld_we, const float* Wea,
                                                 int64_t ld_wea, float* dz, float* d_e, int32_t accumulate, float* dPi,
                                                 float* partial, pamnet_stream_t stream) {
    if (n_edges < 0 || n_nodes < 0) return PAMNET_EINVAL;
    if (n_nodes == 0) return PAMNET_EINVAL;                     // (the slots must be written: no launch, no partial sums)
    if (ld_we < DIM || ld_wea < DIM) return PAMNET_EINVAL;      // (fp32 matrices only: no fragment images, see load_wfragb2_t)
    if (n_edges >= (int64_t(1) << 23) || n_nodes >= (int64_t(1) << 23)) return PAMNET_EINVAL;   // 32-bit byte offsets per stream
    if (!d_agg || !ptr || !We || !Wea || !dPi || !partial) return PAMNET_ENULL;
    if (n_edges > 0 && (!row_of || !z || !ea || !e || !dz || !d_e)) return PAMNET_ENULL;
    GAggBwdW a{d_agg, z, ea, e, We, Wea, ptr, row_of, cuts, dz, d_e, dPi, partial, n_edges, n_nodes, (int)ld_we, (int)ld_wea,
               (int)accumulate};
    const int64_t grid = agg_grid(n_edges);
    if (accumulate) hipLaunchKernelGGL(global_edge_agg_bwd_wg_kernel<true>, dim3((unsigned)grid), dim3(WG8), 0, as_stream(stream), a);
    else hipLaunchKernelGGL(global_edge_agg_bwd_wg_kernel<false>, dim3((unsigned)grid), dim3(WG8), 0, as_stream(stream), a);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

// m_t[e] = m_ji[e] + sum_{r in [t_ptr[e], t_ptr[e+1])} m_nb[t_col[r]] * s[r]      (m_t: optional save)
// out[i] = init[i] + sum_{e in [l_ptr[i], l_ptr[i+1])} q3[e] * m_t[e]
extern "C" int pamnet_local_agg_fwd_f32(const float* m_ji, const float* m_nb, const float* s, const float* q3,
                                        const int32_t* t_ptr, const int32_t* t_col, const int32_t* l_ptr,
                                        const float* init, int64_t n_nodes, float* m_t, float* out,
                                        pamnet_stream_t stream) {
    if (n_nodes < 0) return PAMNET_EINVAL;
    if (n_nodes == 0) return PAMNET_OK;
    if (!l_ptr || !t_ptr || !out) return PAMNET_ENULL;        // the float inputs may be null when there are no edges
    hipLaunchKernelGGL(local_agg_fwd_kernel, dim3((unsigned)ceil_div(n_nodes, NPW)), dim3(256), 0, as_stream(stream),
                       (const float4*)m_ji, (const float4*)m_nb, (const float4*)s, (const float4*)q3, t_ptr, t_col, l_ptr,
                       (const float4*)init, (float4*)m_t, (float4*)out, n_nodes);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

// out_edge[q] = t_row[tT_perm[q]], out_node[q] = l_row[out_edge[q]] for the n_rows entries of the transposed triplet / pair list:
// the two dependent index reads of pamnet_local_agg_bwd_f32's gather half, done once per graph.
__global__ __launch_bounds__(256) void tt_aux_kernel(const int32_t* __restrict__ tT_perm, const int32_t* __restrict__ t_row,
                                                     const int32_t* __restrict__ l_row, int64_t n, int32_t* __restrict__ oe,
                                                     int32_t* __restrict__ on) {
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (int64_t)gridDim.x * 256) {
        const int e = t_row[tT_perm[q]];
        oe[q] = e;
        on[q] = l_row[e];
    }
}
extern "C" int pamnet_triplet_transpose_aux_i32(const int32_t* tT_perm, const int32_t* t_row, const int32_t* l_row,
                                                int64_t n_rows, int32_t* out_edge, int32_t* out_node, pamnet_stream_t stream) {
    if (n_rows < 0) return PAMNET_EINVAL;
    if (n_rows == 0) return PAMNET_OK;
    if (!tT_perm || !t_row || !l_row || !out_edge || !out_node) return PAMNET_ENULL;
    int64_t grid = ceil_div(n_rows, 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(tt_aux_kernel, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), tT_perm, t_row, l_row, n_rows,
                       out_edge, out_node);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

// Backward of pamnet_local_agg_fwd_f32 (one launch): d_mt[e] = d_x2[l_row[e]] * q3[e], d_q3[e] = d_x2[l_row[e]] * m_t[e],
// d_s[r] = m_nb[t_col[r]] * d_mt[t_row[r]], d_mnb[e'] = sum_{r: t_col[r] = e'} s[r] * d_mt[t_row[r]]  (tT_*: transposed
// CSR of t_col over the edges).
extern "C" int pamnet_local_agg_bwd_f32(const float* d_x2, const int32_t* l_row, const float* q3, const float* m_t,
                                        const float* m_nb, const float* s, const int32_t* t_ptr, const int32_t* t_col,
                                        const int32_t* t_row, const int32_t* tT_ptr, const int32_t* tT_perm,
                                        const int32_t* tT_edge, const int32_t* tT_node, int64_t n_edges, float* d_mt,
                                        float* d_q3, float* d_s, float* d_mnb, pamnet_stream_t stream) {
    if (n_edges < 0) return PAMNET_EINVAL;
    if (n_edges == 0) return PAMNET_OK;
    if (!d_x2 || !l_row || !q3 || !m_t || !m_nb || !t_ptr || !tT_ptr || !d_mt || !d_q3 || !d_mnb) return PAMNET_ENULL;
    if ((tT_edge == nullptr) != (tT_node == nullptr)) return PAMNET_EINVAL;           // both auxiliary lists or none
    hipLaunchKernelGGL(local_agg_bwd_kernel, dim3((unsigned)ceil_div(n_edges, 8), 2), dim3(256), 0, as_stream(stream),
                       (const float4*)d_x2, l_row, (const float4*)q3, (const float4*)m_t, (const float4*)m_nb,
                       (const float4*)s, t_ptr, t_col, t_row, tT_ptr, tT_perm, tT_edge, tT_node, n_edges, (float4*)d_mt, (float4*)d_q3,
                       (float4*)d_s, (float4*)d_mnb);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}
