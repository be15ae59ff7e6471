// Fused edge-level kernels of PAMNet's message passing (dim = 128), forward and backward.
//
// With the message MLPs split algebraically (W[x_i|x_j|e] = W_i x_i + W_j x_j + W_e e) the per-edge work is
//   global (layers/global_message_passing.py:52-56):
//       z = W_e e + b_m + P_i[i] + P_j[j],   m = SiLU(z) * (W_ea e)
//   local  (layers/local_message_passing.py:46-48, 53):
//       z_ji = W_ji,e r + b_ji + P0[i] + P2[j]      m_ji = SiLU(z_ji)
//       z_kj = W_kj,e r + b_kj + P1[i] + P3[j]      m_nb = SiLU(z_kj) * (lin_rbf r)       q3 = lin_rbf_out r
//   triplet/pair MLP (layers/local_message_passing.py:49):  s = SiLU(W2 SiLU(W1 sbf + b1) + b2)
// A kernel keeps a row tile in LDS, runs its 2-4 fp32-MFMA GEMMs on it and applies the gather-add / SiLU / gate epilogue
// while the tile is written out in coalesced 512-byte rows -- the [E,3d] concatenations and the per-edge GEMM outputs
// of the reference never exist in memory.  P_* are the node-level projections (node_chain.hip); edges are sorted by
// target so P_i rows repeat within a tile (L1/L2 hits).
//
// Launch shape (measured with tools/phase_probe.py): the hardware hands workgroups to CUs round-robin
// and two workgroups on one CU share its matrix pipes, so a fixed 64-row tiling ran as long as its unluckiest CU --
// 276 tiles on 256 CUs took twice a tile's time, 514 tiles three times.  Here every launch is ONE balanced wave:
//   * the rows are cut into 16-row MFMA tiles and dealt evenly to <= 256 workgroups (one per CU);
//   * a workgroup is 8 waves, wave w owns output columns [16w, 16w+16): every weight slice a kernel needs (up to four
//     128x16 slices = 128 VGPRs) is loaded ONCE per workgroup and stays in registers;
//   * the workgroup walks its rows in chunks of up to 128 (two-slot kernels) / 96 (three-slot) rows of LDS.
#include <stdlib.h>

#include "common.h"
#include "gemm_core.h"

using namespace pamnet;

#ifdef PAMNET_PHASE_PROBE
// Development aid (tools/phase_probe.py builds a private copy of this file with -DPAMNET_PHASE_PROBE): timestamps of one
// workgroup at phase boundaries and of every workgroup's start / end.  Never compiled into libpamnet_hip.so.
__device__ long long pamnet_probe_buf[32];
__device__ long long pamnet_probe_wg[2 * 4096];     // wall clock (100 MHz) at start / end of every workgroup
#define PROBE(i)                                                                                   \
    do {                                                                                           \
        if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) {                                     \
            pamnet_probe_buf[i] = clock64();                                                       \
            pamnet_probe_buf[16 + i] = wall_clock64(); /* constant 100 MHz */                      \
        }                                                                                          \
    } while (0)
#define PROBE_WG(slot)                                                                                        \
    do {                                                                                                      \
        if (threadIdx.x == 0 && blockIdx.x < 4096) pamnet_probe_wg[2 * blockIdx.x + slot] = wall_clock64();   \
    } while (0)
extern "C" int pamnet_probe_read(long long* host32) {
    return (int)hipMemcpyFromSymbol(host32, HIP_SYMBOL(pamnet_probe_buf), sizeof(long long) * 32);
}
extern "C" int pamnet_probe_read_wg(long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(pamnet_probe_wg), sizeof(long long) * 2 * n);
}
#else
#define PROBE(i)
#define PROBE_WG(slot)
#endif

#include "edge_core.h"

using namespace edge;

namespace {


// -------------------------------------------------------------------------------------------------- global edges
template <int MTX, int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void global_edge_fwd_kernel(const float* __restrict__ e, int64_t m,
                                                              const float* __restrict__ We, int ld_we,
                                                              const float* __restrict__ bm,
                                                              const float* __restrict__ Wea, int ld_wea,
                                                              const float* __restrict__ Pi, const float* __restrict__ Pj,
                                                              const int32_t* __restrict__ row_of,
                                                              const int32_t* __restrict__ col, float* __restrict__ z,
                                                              float* __restrict__ ea, float* __restrict__ msg, int pa,
                                                              int pb, int pc, int cmt) {
    __shared__ __attribute__((aligned(16))) float lds[2 * MTX * 16 * LDT];
    float* S0 = lds;
    float* S1 = lds + MTX * 16 * LDT;
    constexpr int NS = 8 / NW;
    const int wc = wave_col<NW>();
    const BiasSet<NS> zero_bias = lane_biases<NS>(nullptr, wc);
    const BiasSet<NS> bv = lane_biases<NS>(bm, wc);
    WSet<NS> f1, f2;
    load_wset<false>(f1, We, ld_we, wc);
    load_wset<false>(f2, Wea, ld_wea, wc);
    const Span sp = Span::make<NW>(m, pa, pb, pc, cmt);
    // the NEXT chunk's input rows travel in registers while this chunk's GEMMs run (a workgroup with 9 tiles walks them
    // as 5 + 4: without this its second load phase is fully exposed)
    constexpr int RPP = 2 * NW, NI = 16 * MTX / RPP;          // sweep geometry: rows per pass, passes per chunk
    float4 pre[NI];
    {
        const int c4 = threadIdx.x & 31, r0 = threadIdx.x >> 5;
#pragma unroll
        for (int i = 0; i < NI; ++i) pre[i] = ldg4z(e, sp.beg + r0 + RPP * i, sp.end, DIM, c4);
    }
    CHUNK_LOOP(sp) {
        const int mt = chunk_mt(sp, row0);
        {
            const int c4 = threadIdx.x & 31, r0 = threadIdx.x >> 5;
#pragma unroll
            for (int i = 0; i < NI; ++i)
                if (RPP * i < 16 * mt) st_lds4(S0, r0 + RPP * i, c4, pre[i]);
            const int64_t nxt = row0 + (int64_t)sp.cmt * 16;
            if (nxt < sp.end) {
#pragma unroll
                for (int i = 0; i < NI; ++i) pre[i] = ldg4z(e, nxt + r0 + RPP * i, sp.end, DIM, c4);
            }
        }
        __syncthreads();
        AccSet<MTX, NS> au, aa;
        au.zero();
        aa.zero();
        mma_set<MTX, NS>(S0, f1, au, mt);
        mma_set<MTX, NS>(S0, f2, aa, mt);
        __syncthreads();                                   // every wave is done reading the e tile
        store_set<MTX, NS>(au, S0, wc, bv, mt);
        store_set<MTX, NS>(aa, S1, wc, zero_bias, mt);
        __syncthreads();
        sweep<MTX, NW>(mt, [&](int r, int c4) {
            const int64_t g = row0 + r;
            if (g >= sp.end) return;
            const int64_t i = row_of[g], j = col[g];
            const float4 zz = f4add(f4add(lds4(S0, r, c4), ldg4(Pi, i, DIM, c4)), ldg4(Pj, j, DIM, c4));
            const float4 gate = lds4(S1, r, c4);
            if (z) stg4(z, g, DIM, c4, zz);                    // backward-only saves: null in inference mode
            if (ea) stg4(ea, g, DIM, c4, gate);
            stg4(msg, g, DIM, c4, f4mul(f4silu(zz), gate));
        });
        __syncthreads();
    }
}

// dm[e] = d_agg[i(e)];  dz = dm * ea * SiLU'(z);  dea = dm * SiLU(z);  d_e (+)= dz * W_e + dea * W_ea
template <int MTX, int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void global_edge_bwd_kernel(const float* __restrict__ d_agg,
                                                              const int32_t* __restrict__ row_of, int64_t m,
                                                              const float* __restrict__ z, const float* __restrict__ ea,
                                                              const float* __restrict__ We, int ld_we,
                                                              const float* __restrict__ Wea, int ld_wea,
                                                              float* __restrict__ dz, float* __restrict__ dea,
                                                              float* __restrict__ d_e, int accumulate, int pa, int pb, int pc, int cmt) {
    __shared__ __attribute__((aligned(16))) float lds[2 * MTX * 16 * LDT];
    float* S0 = lds;
    float* S1 = lds + MTX * 16 * LDT;
    constexpr int NS = 8 / NW;
    const int wc = wave_col<NW>();
    const BiasSet<NS> zero_bias = lane_biases<NS>(nullptr, wc);
    WSet<NS> f1, f2;
    load_wset<true>(f1, We, ld_we, wc);
    load_wset<true>(f2, Wea, ld_wea, wc);
    const Span sp = Span::make<NW>(m, pa, pb, pc, cmt);
    CHUNK_LOOP(sp) {
        const int mt = chunk_mt(sp, row0);
        sweep<MTX, NW>(mt, [&](int r, int c4) {
            const int64_t g = row0 + r;
            float4 a = f4zero(), b = f4zero();
            if (g < sp.end) {
                const float4 dm = ldg4(d_agg, row_of[g], DIM, c4);
                const float4 zz = ldg4(z, g, DIM, c4);
                a = f4mul(f4mul(dm, ldg4(ea, g, DIM, c4)), f4dsilu(zz));
                b = f4mul(dm, f4silu(zz));
                stg4(dz, g, DIM, c4, a);
                stg4(dea, g, DIM, c4, b);
            }
            st_lds4(S0, r, c4, a);
            st_lds4(S1, r, c4, b);
        });
        __syncthreads();
        AccSet<MTX, NS> acc;
        acc.zero();
        mma_set<MTX, NS>(S0, f1, acc, mt);
        mma_set<MTX, NS>(S1, f2, acc, mt);
        __syncthreads();
        store_set<MTX, NS>(acc, S0, wc, zero_bias, mt);
        __syncthreads();
        sweep<MTX, NW>(mt, [&](int r, int c4) {
            const int64_t g = row0 + r;
            if (g >= sp.end) return;
            float4 v = lds4(S0, r, c4);
            if (accumulate) v = f4add(v, ldg4(d_e, g, DIM, c4));
            stg4(d_e, g, DIM, c4, v);
        });
        __syncthreads();
    }
}

// -------------------------------------------------------------------------------------------------- local edges
struct LocalW {
    const float* W[4];      // ji_e, kj_e, lin_rbf, lin_rbf_out   ([128][*] blocks)
    int ld[4];
    const float* P[4];      // node planes: ji_i, kj_i, ji_j, kj_j  ([N][128] each)
};

// blockIdx.y = 0: the k->j half (z_kj, q2, m_nb);  1: the j->i half (z_ji, m_ji, q3).  The two halves share only the
// input tile, so they run as separate workgroups: two weight slices (64 VGPRs) and one epilogue each instead of four
// slices and two dependent epilogues in a row -- E_l rows give only ~135 workgroups per half, the chip has room.
template <int MTX, int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void local_edge_fwd_kernel(const float* __restrict__ rbf, int64_t m, LocalW w,
                                                             const float* __restrict__ b_ji,
                                                             const float* __restrict__ b_kj,
                                                             const int32_t* __restrict__ row_of,
                                                             const int32_t* __restrict__ col, float* __restrict__ z_ji,
                                                             float* __restrict__ z_kj, float* __restrict__ q2,
                                                             float* __restrict__ q3, float* __restrict__ m_ji,
                                                             float* __restrict__ m_nb, int pa, int pb, int pc, int cmt) {
    constexpr int NS = 8 / NW;
    constexpr bool B16 = NS == 1;
    // the rbf rows: piece planes in the 8-wave geometry (split by the sweep that stages them, edge_core.h), fp32 otherwise
    constexpr int A_B = B16 ? PTILE : 16 * LDT * 4;
    __shared__ __attribute__((aligned(16))) char ldsb[MTX * (A_B + 2 * 16 * LDT * 4)];
    float* S0 = reinterpret_cast<float*>(ldsb);
    char* P = ldsb;
    float* S1 = reinterpret_cast<float*>(ldsb + MTX * A_B);
    float* S2 = S1 + MTX * 16 * LDT;
    const int wc = wave_col<NW>();
    const BiasSet<NS> zero_bias = lane_biases<NS>(nullptr, wc);
    const bool kj = blockIdx.y == 0;
    const BiasSet<NS> bz = lane_biases<NS>(kj ? b_kj : b_ji, wc);
    // slice producing z (with bias), slice producing the gate.  8-wave geometry (one slice per wave): both as resident bf16x3
    // pieces, the two GEMMs share one split of every A fragment (edge_core.h "bf16x6"); the paired 4-wave geometry keeps
    // fp32 MFMAs (four slices as pieces would not fit the registers of two co-resident workgroups).
    WSet<B16 ? 0 : NS> fz, fq;
    WFragB1 bz_, bq_;
    if constexpr (B16) {
        load_wfragb1<false>(bz_, w.W[kj ? 1 : 0], w.ld[kj ? 1 : 0], wc);
        load_wfragb1<false>(bq_, w.W[kj ? 2 : 3], w.ld[kj ? 2 : 3], wc);
    } else {
        load_wset<false>(fz, w.W[kj ? 1 : 0], w.ld[kj ? 1 : 0], wc);
        load_wset<false>(fq, w.W[kj ? 2 : 3], w.ld[kj ? 2 : 3], wc);
    }
    const float* __restrict__ Pi = w.P[kj ? 1 : 0];
    const float* __restrict__ Pj = w.P[kj ? 3 : 2];
    const Span sp = Span::make<NW>(m, pa, pb, pc, cmt);
    CHUNK_LOOP(sp) {
        const int mt = chunk_mt(sp, row0);
        // (8-wave geometry, chunks of <= 3 tiles) the node-plane rows of the epilogue: their indices are requested ahead of the
        // staging sweep, the rows themselves behind it -- in the epilogue sweep they were two dependent round trips behind the
        // GEMMs' barrier (clamped rows: every request is unconditional)
        constexpr bool AHEAD = B16 && MTX <= 3;
        constexpr int NI = AHEAD ? MTX : 1;
        float4 gpi[NI], gpj[NI];
        int ii[NI], jj[NI];
        if constexpr (AHEAD) {
            const int rq = threadIdx.x >> 5;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                int64_t g = row0 + rq + 16 * i;
                g = g < sp.end ? g : sp.end - 1;
                ii[i] = row_of[g], jj[i] = col[g];
            }
        }
        sweep<MTX, NW>(mt, [&](int r, int c4) {
            const float4 v = ldg4z(rbf, row0 + r, sp.end, DIM, c4);
            if constexpr (B16) st_pieces4(P, r, c4, v);
            else st_lds4(S0, r, c4, v);
        });
        if constexpr (AHEAD) {
            const int c4 = threadIdx.x & 31;
#pragma unroll
            for (int i = 0; i < NI; ++i) gpi[i] = ldg4(Pi, ii[i], DIM, c4), gpj[i] = ldg4(Pj, jj[i], DIM, c4);
        }
        __syncthreads();
        if constexpr (B16) {
            AccSet<MTX, NS> accq, accz;
            accq.zero();
            accz.zero();
            mma_p16<MTX, false, 3>(P, bq_, accq.a[0], bz_, accz.a[0], mt);
            store_set<MTX, NS>(accq, S2, wc, zero_bias, mt);            // q2 = lin_rbf r   |  q3 = lin_rbf_out r
            store_set<MTX, NS>(accz, S1, wc, bz, mt);                   // W_kj,e r + b_kj  |  W_ji,e r + b_ji
        } else {
            AccSet<MTX, NS> acc;
            acc.zero();
            mma_set<MTX, NS>(S0, fq, acc, mt);                          // q2 = lin_rbf r   |  q3 = lin_rbf_out r
            store_set<MTX, NS>(acc, S2, wc, zero_bias, mt);
            acc.zero();
            mma_set<MTX, NS>(S0, fz, acc, mt);                          // W_kj,e r + b_kj  |  W_ji,e r + b_ji
            store_set<MTX, NS>(acc, S1, wc, bz, mt);
        }
        __syncthreads();
        // (the pass index is a compile-time constant here: indexing gpi / gpj by r >> 4 made them dynamically indexed registers --
        // 10 -> 30 us)
        auto epilogue = [&](int r, int c4, const float4& pi_, const float4& pj_) __attribute__((always_inline)) {
            const int64_t g = row0 + r;
            const float4 zz = f4add(f4add(lds4(S1, r, c4), pi_), pj_);
            const float4 gate = lds4(S2, r, c4);
            if (kj) {
                if (z_kj) stg4_nt(z_kj, g, DIM, c4, zz);         // z_*, q2: backward-only saves, null in inference mode
                if (q2) stg4_nt(q2, g, DIM, c4, gate);
                stg4(m_nb, g, DIM, c4, f4mul(f4silu(zz), gate));
            } else {
                if (z_ji) stg4_nt(z_ji, g, DIM, c4, zz);
                stg4(m_ji, g, DIM, c4, f4silu(zz));
                stg4(q3, g, DIM, c4, gate);
            }
        };
        if constexpr (AHEAD) {
            const int c4 = threadIdx.x & 31, rq = threadIdx.x >> 5;
#pragma unroll
            for (int i = 0; i < NI; ++i)
                if (16 * i < 16 * mt && row0 + rq + 16 * i < sp.end) epilogue(rq + 16 * i, c4, gpi[i], gpj[i]);
        } else {
            sweep<MTX, NW>(mt, [&](int r, int c4) {
                const int64_t g = row0 + r;
                if (g >= sp.end) return;
                const int64_t i = row_of[g], j = col[g];
                epilogue(r, c4, ldg4(Pi, i, DIM, c4), ldg4(Pj, j, DIM, c4));
            });
        }
        __syncthreads();
    }
}

// dz_ji = d_mji * SiLU'(z_ji);  dz_kj = d_mnb * q2 * SiLU'(z_kj);  dq2 = d_mnb * SiLU(z_kj);  dq3 given.
// d_rbf (+)= dz_ji W_ji,e + dz_kj W_kj,e + dq2 W_lin_rbf + dq3 W_lin_rbf_out
// MTX = tiles per chunk (LDS and register footprint follow it): the local graph is small, 3 is plenty and keeps the four
// weight slices + accumulators free of scratch spills (the 8-tile instantiation needed 172 B/lane)
struct LocalBwdArgs {
    const float *d_mji, *d_mnb, *d_q3;
    int64_t m;
    const float *z_ji, *z_kj, *q2;
    LocalW w;
    float *dz_ji, *dz_kj, *dq2, *d_rbf;
    int accumulate, base, rem, cmt;
};

// workgroup `bid` of the plan (base, rem, cmt); lds: LOCAL_BWD_LDS_B(MTX) bytes
// Round 6: the four dX GEMMs on the bf16 matrix pipe at fp32 accuracy ("bf16x6", edge_core.h) like every other edge-level GEMM --
// on fp32 MFMAs they were 384 matrix instructions of 32 cycles per wave and chunk (the pole of the backward pair at the QM9 batch:
// 10 us of matrix pipe for three tiles), now 288 of 16.  Two passes of two operands: each operand tile is split ONCE into piece
// planes by the sweep that computes it, the two weight slices of a pass are resident as bf16x3 pieces (a row stride of 0: kind-1
// fragment images), the second pass's slices are requested behind the first pass's products.
// IMG: the slices arrive as images (compile-time: the loads sit inside the chunk loop, and with both arms of the loader there
// the compiler parks the fragments in scratch).  IMG = false, fp32 matrices, splits the same pieces: bitwise the same results.
constexpr int LOCAL_BWD_LDS_B(int mtx) { return mtx * (2 * PTILE + 16 * LDT * 4); }
template <int MTX, bool IMG>
__device__ __forceinline__ void local_edge_bwd_body(const LocalBwdArgs& a, const int bid, float* lds) {
    const float* __restrict__ d_mji = a.d_mji;
    const float* __restrict__ d_mnb = a.d_mnb;
    const float* __restrict__ d_q3 = a.d_q3;
    const float* __restrict__ z_ji = a.z_ji;
    const float* __restrict__ z_kj = a.z_kj;
    const float* __restrict__ q2 = a.q2;
    float* __restrict__ dz_ji = a.dz_ji;
    float* __restrict__ dz_kj = a.dz_kj;
    float* __restrict__ dq2 = a.dq2;
    float* __restrict__ d_rbf = a.d_rbf;
    const int accumulate = a.accumulate;
    const LocalW& w = a.w;
    char* P0 = reinterpret_cast<char*>(lds);                   // piece planes of the pass's two operands
    char* P1 = P0 + MTX * PTILE;
    float* S0 = reinterpret_cast<float*>(P1 + MTX * PTILE);    // the accumulators' fp32 tile
    const int wc = wave_col<8>();
    WFragB1 fa, fb;
    const Span sp = Span::make_at<8>(a.m, a.base, a.rem, 0, a.cmt, bid);
    CHUNK_LOOP(sp) {
        const int mt = chunk_mt(sp, row0);
        Acc<MTX> acc;
        acc.zero();
        load_wfragb1<true, IMG ? 1 : 0>(fa, w.W[0], w.ld[0], wc);           // (per chunk: one or two chunks per workgroup)
        load_wfragb1<true, IMG ? 1 : 0>(fb, w.W[1], w.ld[1], wc);
        // pass A: dz_ji -> P0, dz_kj -> P1
        sweep<MTX, 8>(mt, [&](int r, int c4) {
            const int64_t g = row0 + r;
            float4 x = f4zero(), y = f4zero();
            if (g < sp.end) {
                x = f4mul(ldg4(d_mji, g, DIM, c4), f4dsilu(ldg4(z_ji, g, DIM, c4)));
                y = f4mul(f4mul(ldg4(d_mnb, g, DIM, c4), ldg4(q2, g, DIM, c4)), f4dsilu(ldg4(z_kj, g, DIM, c4)));
                stg4(dz_ji, g, DIM, c4, x);
                stg4(dz_kj, g, DIM, c4, y);
            }
            st_pieces4(P0, r, c4, x);
            st_pieces4(P1, r, c4, y);
        });
        __syncthreads();
        mma_p16<MTX, true, 3>(P0, fa, acc, fa, acc, mt);
        mma_p16<MTX, true, 3>(P1, fb, acc, fb, acc, mt);
        load_wfragb1<true, IMG ? 1 : 0>(fa, w.W[2], w.ld[2], wc);
        load_wfragb1<true, IMG ? 1 : 0>(fb, w.W[3], w.ld[3], wc);
        __syncthreads();
        // pass B: dq2 -> P0, dq3 -> P1
        sweep<MTX, 8>(mt, [&](int r, int c4) {
            const int64_t g = row0 + r;
            float4 x = f4zero(), y = f4zero();
            if (g < sp.end) {
                x = f4mul(ldg4(d_mnb, g, DIM, c4), f4silu(ldg4(z_kj, g, DIM, c4)));
                y = ldg4(d_q3, g, DIM, c4);
                stg4(dq2, g, DIM, c4, x);
            }
            st_pieces4(P0, r, c4, x);
            st_pieces4(P1, r, c4, y);
        });
        __syncthreads();
        mma_p16<MTX, true, 3>(P0, fa, acc, fa, acc, mt);
        mma_p16<MTX, true, 3>(P1, fb, acc, fb, acc, mt);
        acc_store<MTX>(acc, S0, wc, 0.f, mt);                  // (S0 is the accumulators' alone: no barrier ahead of it)
        __syncthreads();
        sweep<MTX, 8>(mt, [&](int r, int c4) {
            const int64_t g = row0 + r;
            if (g >= sp.end) return;
            float4 v = lds4(S0, r, c4);
            if (accumulate) v = f4add(v, ldg4(d_rbf, g, DIM, c4));
            stg4(d_rbf, g, DIM, c4, v);
        });
        __syncthreads();
    }
}

template <int MTX, bool IMG>
__global__ __launch_bounds__(WG8) void local_edge_bwd_kernel(LocalBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[LOCAL_BWD_LDS_B(MTX) / 4];
    local_edge_bwd_body<MTX, IMG>(a, (int)blockIdx.x, lds);
}

// -------------------------------------------------------------------------------------------------- 2-layer MLP
// blockIdx.y selects one of up to 8 independent (weights, outputs) sets applied to the same input rows: the triplet/pair
// MLPs of all layers depend only on the basis embedding, so the engine runs them in one launch up front.
struct Mlp2Batch {
    Mlp2Set s[8];
};
template <int MTX, int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void mlp2_fwd_kernel(const float* __restrict__ x, int64_t m, Mlp2Batch batch, int pa,
                                                       int pb, int pc, int cmt) {
    __shared__ __attribute__((aligned(16))) float lds[MTX * MLP2_TILE_B / 4];
    PROBE_WG(0);
    mlp2_fwd_body<MTX, NW>(x, batch.s[blockIdx.y], Span::make<NW>(m, pa, pb, pc, cmt), lds);
    PROBE_WG(1);
}

struct Mlp2BwdArgs {
    const float* dy;
    int64_t m;
    const float *z1, *z2, *W1, *W2;
    float *dz1, *dz2, *dx;
    int accumulate, pa, pb, pc, cmt;
    int img = 0;                              // W1, W2 are bf16x3 fragment images (transposed orientation; 8-wave geometry)
};

template <int MTX, int NW>
__device__ __forceinline__ void mlp2_bwd_body(const Mlp2BwdArgs& a, const int bid, float* lds) {
    const float* __restrict__ dy = a.dy;
    const float* __restrict__ z1 = a.z1;
    const float* __restrict__ z2 = a.z2;
    const float* __restrict__ W1 = a.W1;
    const float* __restrict__ W2 = a.W2;
    float* __restrict__ dz1 = a.dz1;
    float* __restrict__ dz2 = a.dz2;
    float* __restrict__ dx = a.dx;
    const int accumulate = a.accumulate;
    constexpr int NS = 8 / NW;
    constexpr bool B16 = NS == 1;             // 8-wave geometry: the dX GEMMs on the bf16 matrix pipe (edge_core.h "bf16x6"),
                                              // their inputs as piece planes written by the sweeps that compute them
    // B16: P = piece planes (dz2, then dz1), S1 = the accumulators' fp32 tiles;  else S0 / S1 = two fp32 tile arrays
    float* S0 = lds;
    char* P = reinterpret_cast<char*>(lds);
    float* S1 = B16 ? reinterpret_cast<float*>(P + MTX * PTILE) : lds + MTX * 16 * LDT;
    const int wc = wave_col<NW>();
    const BiasSet<NS> zero_bias = lane_biases<NS>(nullptr, wc);
    WSet<B16 ? 0 : NS> f1, f2;
    WFragB1 b1_, b2_;
    if constexpr (B16) {
        load_wfragb1<true>(b2_, W2, a.img ? 0 : DIM, wc);
        load_wfragb1<true>(b1_, W1, a.img ? 0 : DIM, wc);
    } else {
        load_wset<true>(f2, W2, DIM, wc);
        load_wset<true>(f1, W1, DIM, wc);
    }
    const Span sp = Span::make_at<NW>(a.m, a.pa, a.pb, a.pc, a.cmt, bid);
    CHUNK_LOOP(sp) {
        const int mt = chunk_mt(sp, row0);
        // z1 and the accumulate operand are requested together with dy / z2: no global load waits mid-chunk
        constexpr int RPP = 2 * NW, NI = 16 * MTX / RPP;      // sweep geometry: rows per pass, passes per chunk
        float4 z1r[NI], dxr[NI];
        {
            const int c4 = threadIdx.x & 31, r0 = threadIdx.x >> 5;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int64_t g = row0 + r0 + RPP * i;
                float4 a = f4zero();
                z1r[i] = f4zero();
                dxr[i] = f4zero();
                if (RPP * i < 16 * mt && g < sp.end) {
                    a = f4mul(ldg4(dy, g, DIM, c4), f4dsilu(ldg4(z2, g, DIM, c4)));
                    z1r[i] = ldg4(z1, g, DIM, c4);
                    if (accumulate) dxr[i] = ldg4(dx, g, DIM, c4);
                    stg4(dz2, g, DIM, c4, a);
                }
                if (RPP * i < 16 * mt) {
                    if constexpr (B16) st_pieces4(P, r0 + RPP * i, c4, a);
                    else st_lds4(S0, r0 + RPP * i, c4, a);
                }
            }
        }
        __syncthreads();
        AccSet<MTX, NS> acc;
        acc.zero();
        if constexpr (B16) mma_p16<MTX, true, 3>(P, b2_, acc.a[0], b2_, acc.a[0], mt);
        else mma_set<MTX, NS>(S0, f2, acc, mt);
        store_set<MTX, NS>(acc, S1, wc, zero_bias, mt);
        __syncthreads();
        {
            const int c4 = threadIdx.x & 31, r0 = threadIdx.x >> 5;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                if (RPP * i >= 16 * mt) continue;
                const int r = r0 + RPP * i;
                const int64_t g = row0 + r;
                float4 a = f4zero();
                if (g < sp.end) {
                    a = f4mul(lds4(S1, r, c4), f4dsilu(z1r[i]));
                    stg4(dz1, g, DIM, c4, a);
                }
                if constexpr (B16) st_pieces4(P, r, c4, a);             // (the dz2 pieces were last read before the barrier)
                else st_lds4(S1, r, c4, a);
            }
        }
        __syncthreads();
        acc.zero();
        float* const D = B16 ? S1 : S0;       // B16: S1's dX1 tile was consumed by the sweep above; else S0 (dz2) is free
        if constexpr (B16) mma_p16<MTX, true, 3>(P, b1_, acc.a[0], b1_, acc.a[0], mt);
        else mma_set<MTX, NS>(S1, f1, acc, mt);
        store_set<MTX, NS>(acc, D, wc, zero_bias, mt);
        __syncthreads();
        {
            const int c4 = threadIdx.x & 31, r0 = threadIdx.x >> 5;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int r = r0 + RPP * i;
                const int64_t g = row0 + r;
                if (RPP * i < 16 * mt && g < sp.end) stg4(dx, g, DIM, c4, f4add(lds4(D, r, c4), dxr[i]));
            }
        }
        __syncthreads();
    }
}

template <int MTX, int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void mlp2_bwd_kernel(const float* __restrict__ dy, int64_t m,
                                                       const float* __restrict__ z1, const float* __restrict__ z2,
                                                       const float* __restrict__ W1, const float* __restrict__ W2,
                                                       float* __restrict__ dz1, float* __restrict__ dz2,
                                                       float* __restrict__ dx, int accumulate, int pa, int pb, int pc, int cmt) {
    __shared__ __attribute__((aligned(16))) float lds[(NW == 8 ? MTX * MLP2_TILE_B : 2 * MTX * 16 * LDT * 4) / 4];
    mlp2_bwd_body<MTX, NW>(Mlp2BwdArgs{dy, m, z1, z2, W1, W2, dz1, dz2, dx, accumulate, pa, pb, pc, cmt}, (int)blockIdx.x, lds);
}

// The two kernels of a local layer's backward that depend on local_agg_bwd only and not on each other -- the triplet / pair
// MLP (mlp2_bwd) and the local edge stage (local_edge_bwd) -- as ONE launch: workgroups [0, g_mlp) walk the MLP's plan, the
// rest the edge plan, the CUs split between them by their work.  Each alone is one round of <= 256 workgroups whose fixed
// cost (launch boundary + prologue: weights, work split, first rows) is ~9 us of its 20-24 us at the QM9 batch; side by
// side that cost is paid once.  Same bodies, same per-row arithmetic: results are bitwise those of the two launches.
// (cost of a local edge row in triplet / pair MLP rows, for the CU split of the pair: four GEMMs + three sweeps against two +
// three; swept 0.8 .. 6.5 at the QM9 batch, flat from 2.5 to 4: profiles/r06_local_bwd_bf16.txt)
constexpr double PAIR_EDGE_COST = 3.1;
template <int MTM, bool IMG>
__global__ __launch_bounds__(WG8) void local_bwd_pair_kernel(Mlp2BwdArgs ma, LocalBwdArgs la, int g_mlp) {
    constexpr int MTL = 3;
    constexpr int LDS_B = MTM * MLP2_TILE_B > LOCAL_BWD_LDS_B(MTL) ? MTM * MLP2_TILE_B : LOCAL_BWD_LDS_B(MTL);
    __shared__ __attribute__((aligned(16))) float lds[LDS_B / 4];
    if ((int)blockIdx.x < g_mlp) mlp2_bwd_body<MTM, 8>(ma, (int)blockIdx.x, lds);
    else local_edge_bwd_body<MTL, IMG>(la, (int)blockIdx.x - g_mlp, lds);
}

// One balanced wave of workgroups: `per` 16-row tiles each (<= N_CU workgroups), walked in chunks of `cmt` <= cap tiles.
struct Plan {
    unsigned grid;
    int pa, pb, pc, cmt;       // Span::make arguments
    bool paired;
};
// One wave of 8-wave workgroups, one per CU (or per `target_wgs`): pa = base, pb = rem tiles per workgroup.
inline Plan plan8(int64_t rows, int cap, int target_wgs = N_CU) {
    const int64_t tiles16 = ceil_div(rows, 16);
    const int64_t per = ceil_div(tiles16, target_wgs);
    const int64_t grid = ceil_div(tiles16, per);
    const int64_t nchunk = ceil_div(per, cap);
    Plan p;
    p.grid = (unsigned)grid;
    p.pa = (int)(tiles16 / grid);
    p.pb = (int)(tiles16 % grid);
    p.pc = 0;
    p.cmt = (int)ceil_div(per, nchunk);
    p.paired = false;
    return p;
}
// Pairs of 4-wave workgroups: pair c owns pa tiles (pa = ceil(tiles / target pairs)), split pc + (pa - pc).
inline Plan plan4(int64_t rows, int cap, int target_pairs = N_CU) {
    const int64_t tiles16 = ceil_div(rows, 16);
    const int64_t per = ceil_div(tiles16, target_pairs);
    const int64_t pairs = ceil_div(tiles16, per);
    const int64_t hi = (per + 1) / 2;
    Plan p;
    p.pa = (int)per;
    p.pb = (int)pairs;
    p.pc = (int)hi;
    p.grid = (unsigned)(per > hi ? 2 * pairs : pairs);
    p.cmt = (int)ceil_div(hi, ceil_div(hi, cap));
    p.paired = true;
    return p;
}
// Measured at the QM9 B=128 shape (tools/step_profile.py): the paired 4-wave geometry wins where a launch spans several
// rounds of workgroups (the all-layers triplet/pair MLP: 104 vs 120 us) and loses or ties on the single-round kernels
// (17 -> 20 us local edge forward, 25 -> 29 us mlp2 backward), so it is the default only for the former.
// PAMNET_EDGE_WAVES=4 / 8 forces one geometry everywhere (tests cover both).
inline int forced_waves() {
    static const int v = [] { const char* e = getenv("PAMNET_EDGE_WAVES"); return e ? atoi(e) : 0; }();
    return v;
}
inline bool four_waves(bool preferred = false) {
    const int f = forced_waves();
    return f == 4 || (f != 8 && preferred);
}
inline Plan plan(int64_t rows, int cap8, int cap4, int target = N_CU, bool prefer4 = false) {
    return four_waves(prefer4) ? plan4(rows, cap4, target) : plan8(rows, cap8, target);
}

// Instantiations by geometry and chunk size: registers (accumulators, A fragments, unrolled sweeps) and LDS follow the
// template bound, so a launch uses the smallest one that holds its chunks: 8 waves x {3, 5, 8} tiles of 16 rows, or
// 4 waves x {2, 3} (two workgroups per CU; larger 4-wave chunks spill: the sweeps unroll twice as far).
#define PAMNET_EDGE_LAUNCH(KERNEL, PLAN, GRIDY, ...)                                                                   \
    do {                                                                                                               \
        const dim3 grid__((PLAN).grid, GRIDY);                                                                         \
        if ((PLAN).paired) {                                                                                           \
            if ((PLAN).cmt <= 2)                                                                                       \
                hipLaunchKernelGGL((KERNEL<2, 4>), grid__, dim3(256), 0, as_stream(stream), __VA_ARGS__, (PLAN).pa,    \
                                   (PLAN).pb, (PLAN).pc, (PLAN).cmt);                                                  \
            else                                                                                                       \
                hipLaunchKernelGGL((KERNEL<3, 4>), grid__, dim3(256), 0, as_stream(stream), __VA_ARGS__, (PLAN).pa,    \
                                   (PLAN).pb, (PLAN).pc, (PLAN).cmt);                                                  \
        } else if ((PLAN).cmt <= 3) {                                                                                  \
            hipLaunchKernelGGL((KERNEL<3, 8>), grid__, dim3(WG8), 0, as_stream(stream), __VA_ARGS__, (PLAN).pa,        \
                               (PLAN).pb, (PLAN).pc, (PLAN).cmt);                                                      \
        } else if ((PLAN).cmt <= 5) {                                                                                  \
            hipLaunchKernelGGL((KERNEL<5, 8>), grid__, dim3(WG8), 0, as_stream(stream), __VA_ARGS__, (PLAN).pa,        \
                               (PLAN).pb, (PLAN).pc, (PLAN).cmt);                                                      \
        } else {                                                                                                       \
            hipLaunchKernelGGL((KERNEL<8, 8>), grid__, dim3(WG8), 0, as_stream(stream), __VA_ARGS__, (PLAN).pa,        \
                               (PLAN).pb, (PLAN).pc, (PLAN).cmt);                                                      \
        }                                                                                                              \
    } while (0)

// triplet / pair MLP forward and backward: piece planes + one fp32 tile per 16 rows (20.7 KB): <= 7 tiles with 8 waves
#define PAMNET_EDGE_LAUNCH7(KERNEL, PLAN, GRIDY, ...)                                                                      \
    do {                                                                                                               \
        const dim3 grid__((PLAN).grid, GRIDY);                                                                         \
        if ((PLAN).paired) {                                                                                           \
            if ((PLAN).cmt <= 2)                                                                                       \
                hipLaunchKernelGGL((KERNEL<2, 4>), grid__, dim3(256), 0, as_stream(stream), __VA_ARGS__,      \
                                   (PLAN).pa, (PLAN).pb, (PLAN).pc, (PLAN).cmt);                                       \
            else                                                                                                       \
                hipLaunchKernelGGL((KERNEL<3, 4>), grid__, dim3(256), 0, as_stream(stream), __VA_ARGS__,      \
                                   (PLAN).pa, (PLAN).pb, (PLAN).pc, (PLAN).cmt);                                       \
        } else if ((PLAN).cmt <= 3) {                                                                                  \
            hipLaunchKernelGGL((KERNEL<3, 8>), grid__, dim3(WG8), 0, as_stream(stream), __VA_ARGS__,          \
                               (PLAN).pa, (PLAN).pb, (PLAN).pc, (PLAN).cmt);                                           \
        } else if ((PLAN).cmt <= 5) {                                                                                  \
            hipLaunchKernelGGL((KERNEL<5, 8>), grid__, dim3(WG8), 0, as_stream(stream), __VA_ARGS__,          \
                               (PLAN).pa, (PLAN).pb, (PLAN).pc, (PLAN).cmt);                                           \
        } else {                                                                                                       \
            hipLaunchKernelGGL((KERNEL<7, 8>), grid__, dim3(WG8), 0, as_stream(stream), __VA_ARGS__,          \
                               (PLAN).pa, (PLAN).pb, (PLAN).pc, (PLAN).cmt);                                           \
        }                                                                                                              \
    } while (0)

// three-slot kernel (local edge forward): chunks of <= 5 tiles with 8 waves (149 KB... 3 x 5 x 16 rows), <= 2 with 4 waves
#define PAMNET_EDGE_LAUNCH3(KERNEL, PLAN, GRIDY, ...)                                                                  \
    do {                                                                                                               \
        const dim3 grid__((PLAN).grid, GRIDY);                                                                         \
        if ((PLAN).paired) {                                                                                           \
            hipLaunchKernelGGL((KERNEL<2, 4>), grid__, dim3(256), 0, as_stream(stream), __VA_ARGS__, (PLAN).pa,        \
                               (PLAN).pb, (PLAN).pc, (PLAN).cmt);                                                      \
        } else if ((PLAN).cmt <= 3) {                                                                                  \
            hipLaunchKernelGGL((KERNEL<3, 8>), grid__, dim3(WG8), 0, as_stream(stream), __VA_ARGS__, (PLAN).pa,        \
                               (PLAN).pb, (PLAN).pc, (PLAN).cmt);                                                      \
        } else {                                                                                                       \
            hipLaunchKernelGGL((KERNEL<5, 8>), grid__, dim3(WG8), 0, as_stream(stream), __VA_ARGS__, (PLAN).pa,        \
                               (PLAN).pb, (PLAN).pc, (PLAN).cmt);                                                      \
        }                                                                                                              \
    } while (0)

}  // namespace

extern "C" int pamnet_global_edge_fwd_f32(const float* e, int64_t n_edges, const float* We, int64_t ld_we,
                                          const float* bm, const float* Wea, int64_t ld_wea, const float* Pi,
                                          const float* Pj, const int32_t* row_of, const int32_t* col, float* z,
                                          float* ea, float* msg, pamnet_stream_t stream) {
    if (n_edges < 0) return PAMNET_EINVAL;
    if (n_edges == 0) return PAMNET_OK;
    if (!e || !We || !bm || !Wea || !Pi || !Pj || !row_of || !col || !msg) return PAMNET_ENULL;   // z, ea: optional saves
    const Plan p = plan(n_edges, MT2, 3);
    PAMNET_EDGE_LAUNCH(global_edge_fwd_kernel, p, 1, e, n_edges, We, (int)ld_we, bm, Wea, (int)ld_wea, Pi, Pj,
                       row_of, col, z, ea, msg);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_global_edge_bwd_f32(const float* d_agg, const int32_t* row_of, int64_t n_edges, const float* z,
                                          const float* ea, const float* We, int64_t ld_we, const float* Wea,
                                          int64_t ld_wea, float* dz, float* dea, float* d_e, int32_t accumulate,
                                          pamnet_stream_t stream) {
    if (n_edges < 0) return PAMNET_EINVAL;
    if (n_edges == 0) return PAMNET_OK;
    if (!d_agg || !row_of || !z || !ea || !We || !Wea || !dz || !dea || !d_e) return PAMNET_ENULL;
    const Plan p = plan(n_edges, MT2, 3);
    PAMNET_EDGE_LAUNCH(global_edge_bwd_kernel, p, 1, d_agg, row_of, n_edges, z, ea, We, (int)ld_we, Wea,
                       (int)ld_wea, dz, dea, d_e, (int)accumulate);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

// *images: all four row strides are 0 = the slices arrive as fragment images (edge_core.h load_wfragb1; forward, 8-wave geometry)
static int fill_local(LocalW& w, const float* const* Wq, const int64_t* ldq, const float* const* P, bool* images = nullptr) {
    int zeros = 0;
    for (int b = 0; b < 4; ++b) zeros += ldq[b] == 0;
    if (zeros != 0 && (zeros != 4 || !images)) return PAMNET_EINVAL;
    if (images) *images = zeros == 4;
    for (int b = 0; b < 4; ++b) {
        if (!Wq[b]) return PAMNET_ENULL;
        if (zeros == 0 && (ldq[b] < DIM || ldq[b] > (1 << 21))) return PAMNET_EINVAL;   // (32-bit byte offsets of a slice)
        w.W[b] = Wq[b];
        w.ld[b] = (int)ldq[b];
        w.P[b] = P ? P[b] : nullptr;
        if (P && !P[b]) return PAMNET_ENULL;
    }
    return PAMNET_OK;
}

extern "C" int pamnet_local_edge_fwd_f32(const float* rbf, int64_t n_edges, const float* const* Wq, const int64_t* ldq,
                                         const float* b_ji, const float* b_kj, const float* const* P,
                                         const int32_t* row_of, const int32_t* col, float* z_ji, float* z_kj, float* q2,
                                         float* q3, float* m_ji, float* m_nb, pamnet_stream_t stream) {
    if (n_edges < 0) return PAMNET_EINVAL;
    if (n_edges == 0) return PAMNET_OK;
    if (!rbf || !Wq || !ldq || !b_ji || !b_kj || !P || !row_of || !col || !q3 || !m_ji || !m_nb)   // z_ji, z_kj, q2: optional
        return PAMNET_ENULL;
    LocalW w;
    bool images = false;
    int rc = fill_local(w, Wq, ldq, P, &images);
    if (rc) return rc;
    const Plan p = plan(n_edges, 5, 2, N_CU / 2);           // two halves (grid.y) share the CUs
    if (images && p.paired) return PAMNET_EINVAL;           // (the 4-wave geometry multiplies fp32 fragments)
    PAMNET_EDGE_LAUNCH3(local_edge_fwd_kernel, p, 2, rbf, n_edges, w, b_ji, b_kj, row_of, col, z_ji, z_kj, q2,
                       q3, m_ji, m_nb);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_local_edge_bwd_f32(const float* d_mji, const float* d_mnb, const float* d_q3, int64_t n_edges,
                                         const float* z_ji, const float* z_kj, const float* q2, const float* const* Wq,
                                         const int64_t* ldq, float* dz_ji, float* dz_kj, float* dq2, float* d_rbf,
                                         int32_t accumulate, pamnet_stream_t stream) {
    if (n_edges < 0) return PAMNET_EINVAL;
    if (n_edges == 0) return PAMNET_OK;
    if (!d_mji || !d_mnb || !d_q3 || !z_ji || !z_kj || !q2 || !Wq || !ldq || !dz_ji || !dz_kj || !dq2 || !d_rbf)
        return PAMNET_ENULL;
    LocalW w;
    bool images = false;                                    // (all four strides 0: transposed fp32 fragment images, load_wfrag1)
    int rc = fill_local(w, Wq, ldq, nullptr, &images);
    if (rc) return rc;
    constexpr int MTL = 3;
    const Plan p = plan8(n_edges, MTL);                     // four weight matrices: stays one 8-wave workgroup per CU
    const LocalBwdArgs la{d_mji, d_mnb, d_q3, n_edges, z_ji, z_kj, q2, w, dz_ji, dz_kj, dq2, d_rbf, (int)accumulate, p.pa, p.pb, p.cmt};
    if (images) hipLaunchKernelGGL((local_edge_bwd_kernel<MTL, true>), dim3(p.grid), dim3(WG8), 0, as_stream(stream), la);
    else hipLaunchKernelGGL((local_edge_bwd_kernel<MTL, false>), dim3(p.grid), dim3(WG8), 0, as_stream(stream), la);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

// pamnet_mlp2_bwd_f32 + pamnet_local_edge_bwd_f32 (same arguments, same results) as one launch: see local_bwd_pair_kernel.
extern "C" int pamnet_local_bwd_pair_f32(const float* dy, int64_t rows, const float* z1, const float* z2, const float* W1,
                                         const float* W2, float* dz1, float* dz2, float* dx, int32_t accumulate_dx,
                                         const float* d_mji, const float* d_mnb, const float* d_q3, int64_t n_edges,
                                         const float* z_ji, const float* z_kj, const float* q2, const float* const* Wq,
                                         const int64_t* ldq, float* dz_ji, float* dz_kj, float* dq2, float* d_rbf,
                                         int32_t accumulate_rbf, pamnet_stream_t stream) {
    if (rows < 0 || n_edges < 0) return PAMNET_EINVAL;
    const bool mlp_img = (accumulate_dx & PAMNET_WEIGHT_IMAGES) != 0;
    accumulate_dx &= ~PAMNET_WEIGHT_IMAGES;
    if (rows == 0 || n_edges == 0 || four_waves()) {         // nothing to pair (or the 4-wave geometry is forced): two launches
        if (mlp_img) return PAMNET_EINVAL;                   // (images: the paired 8-wave launch only)
        int rc = pamnet_mlp2_bwd_f32(dy, rows, z1, z2, W1, W2, dz1, dz2, dx, accumulate_dx, stream);
        if (rc) return rc;
        return pamnet_local_edge_bwd_f32(d_mji, d_mnb, d_q3, n_edges, z_ji, z_kj, q2, Wq, ldq, dz_ji, dz_kj, dq2, d_rbf,
                                         accumulate_rbf, stream);
    }
    if (!dy || !z1 || !z2 || !W1 || !W2 || !dz1 || !dz2 || !dx) return PAMNET_ENULL;
    if (!d_mji || !d_mnb || !d_q3 || !z_ji || !z_kj || !q2 || !Wq || !ldq || !dz_ji || !dz_kj || !dq2 || !d_rbf)
        return PAMNET_ENULL;
    LocalW w;
    bool local_img = false;
    int rc = fill_local(w, Wq, ldq, nullptr, &local_img);
    if (rc) return rc;
    // CU shares by work (PAIR_EDGE_COST); what decides at the QM9 batch is the whole number of tiles a workgroup of each
    // half walks: 3 edge tiles (90 workgroups) beside 7 MLP tiles (166) -- 21.3 us; 4 beside 6: 24.5; 2 beside 10: 26.1
    const double we = PAIR_EDGE_COST * (double)n_edges, wm = (double)rows;
    int ge = (int)(N_CU * we / (we + wm) + 0.5);
    ge = ge < 8 ? 8 : (ge > N_CU - 8 ? N_CU - 8 : ge);
    const Plan pe = plan8(n_edges, 3, ge);
    const Plan pm = plan8(rows, 7, N_CU - (int)pe.grid);      // (piece planes + one fp32 tile per 16 rows: <= 7 tiles of LDS)
    const Mlp2BwdArgs ma{dy, rows, z1, z2, W1, W2, dz1, dz2, dx, (int)accumulate_dx, pm.pa, pm.pb, pm.pc, pm.cmt, mlp_img ? 1 : 0};
    const LocalBwdArgs la{d_mji, d_mnb, d_q3, n_edges, z_ji, z_kj, q2, w, dz_ji, dz_kj, dq2, d_rbf, (int)accumulate_rbf,
                          pe.pa, pe.pb, pe.cmt};
    const dim3 grid(pm.grid + pe.grid);
    hipStream_t st = as_stream(stream);
#define PAMNET_PAIR_LAUNCH(M)                                                                                       \
    do {                                                                                                            \
        if (local_img) hipLaunchKernelGGL((local_bwd_pair_kernel<M, true>), grid, dim3(WG8), 0, st, ma, la, (int)pm.grid);   \
        else hipLaunchKernelGGL((local_bwd_pair_kernel<M, false>), grid, dim3(WG8), 0, st, ma, la, (int)pm.grid);           \
    } while (0)
    if (pm.cmt <= 3) PAMNET_PAIR_LAUNCH(3);
    else if (pm.cmt <= 5) PAMNET_PAIR_LAUNCH(5);
    else PAMNET_PAIR_LAUNCH(7);
#undef PAMNET_PAIR_LAUNCH
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_mlp2_fwd_f32(const float* x, int64_t rows, const float* W1, const float* b1, const float* W2,
                                   const float* b2, float* z1, float* z2, float* y, pamnet_stream_t stream) {
    if (rows < 0) return PAMNET_EINVAL;
    if (rows == 0) return PAMNET_OK;
    if (!x || !W1 || !b1 || !W2 || !b2 || !y) return PAMNET_ENULL;                  // z1, z2: optional saves
    const Plan p = plan(rows, 7, 3);
    Mlp2Batch b;
    for (int k = 0; k < 8; ++k) b.s[k] = Mlp2Set{W1, b1, W2, b2, z1, z2, y};
    PAMNET_EDGE_LAUNCH7(mlp2_fwd_kernel, p, 1, x, rows, b);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

// nsets <= 8 two-layer MLPs on the same input rows in one launch; params[k] = {W1, b1, W2, b2}, outs[k] = {z1, z2, y}.
extern "C" int pamnet_mlp2_fwd_multi_f32(const float* x, int64_t rows, int64_t nsets, const float* const* params,
                                         float* const* outs, pamnet_stream_t stream) {
    if (rows < 0 || nsets < 1 || nsets > 8) return PAMNET_EINVAL;
    if (rows == 0) return PAMNET_OK;
    if (!x || !params || !outs) return PAMNET_ENULL;
    Mlp2Batch b;
    for (int k = 0; k < 8; ++k) {
        const int s = k < nsets ? k : 0;
        for (int i = 0; i < 4; ++i)
            if (!params[4 * s + i]) return PAMNET_ENULL;
        if (!outs[3 * s + 2]) return PAMNET_ENULL;            // z1, z2 (outs[3s], outs[3s+1]) are optional saves
        b.s[k] = Mlp2Set{params[4 * s], params[4 * s + 1], params[4 * s + 2], params[4 * s + 3],
                         outs[3 * s], outs[3 * s + 1], outs[3 * s + 2]};
    }
    // (Round 3 ran several sets -- several rounds of workgroups -- as paired 4-wave workgroups: 104 vs 120 us at the QM9 batch.
    // With piece planes the 8-wave geometry, whose weight pieces stay resident instead of being re-read and re-split per
    // chunk, is ahead again: PDBbind step 9.06 -> 9.02 ms, QM9 2.184 -> 2.178.)
    const Plan p = plan(rows, 7, 3);
    PAMNET_EDGE_LAUNCH7(mlp2_fwd_kernel, p, (unsigned)nsets, x, rows, b);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_mlp2_bwd_f32(const float* dy, int64_t rows, const float* z1, const float* z2, const float* W1,
                                   const float* W2, float* dz1, float* dz2, float* dx, int32_t accumulate,
                                   pamnet_stream_t stream) {
    if (rows < 0) return PAMNET_EINVAL;
    if (rows == 0) return PAMNET_OK;
    if (!dy || !z1 || !z2 || !W1 || !W2 || !dz1 || !dz2 || !dx) return PAMNET_ENULL;
    const Plan p = plan(rows, 7, 3);
    PAMNET_EDGE_LAUNCH7(mlp2_bwd_kernel, p, 1, dy, rows, z1, z2, W1, W2, dz1, dz2, dx, (int)accumulate);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}
