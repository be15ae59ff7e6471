// Input-embedding layers of PAMNet (dim = 128 outputs, tiny K):
//   edge_attr_rbf_{l,g} = SiLU(Linear(16 -> d)(rbf))                       models.py:185-186, layers/basic.py:19-22
//   edge_attr_sbf{1,2}  = SiLU(Linear(42 -> d)(sbf))  (mlp_sbf1 on pair rows, mlp_sbf2 on triplet rows)  models.py:187-188
//   x = init_linear(x_raw[:, 3:])   (18 -> d, no bias, no activation; PDBbind)                           models.py:119
//
// K is 16 / 18 / 42: the GEMM is thin, the traffic is not (E_g x 128 outputs), so these are bandwidth kernels with the
// arithmetic on the matrix unit (v_mfma_f32_16x16x4_f32, exact fp32):
//   * a workgroup (4 waves) owns 64-row tiles; the input rows are staged in LDS zero-padded to KP = 16*ceil(K/16)
//     columns, the weight slices of a wave (32 output columns, KP/16 float4 per 16-column tile) stay in registers;
//   * the combined triplet/pair row list selects between two weight sets per row (`kind`): the row's A fragment is
//     masked into two operands that accumulate W0- and W1-products into the same tile -- this replaces the reference's
//     two Linear calls plus the index_select / index_copy traffic of a host-side split;
//   * outputs leave as coalesced 512-byte rows through an LDS transpose.
// Backward: z is recomputed (cheaper than storing it); dW = dz^T x runs on the matrix unit with the row index as the
// MFMA k dimension, accumulating across a workgroup's tiles in registers; workgroup partials are reduced in a fixed
// order (deterministic); dx (only needed for the trainable Bessel frequencies, K = 16) is a third small GEMM.
#include "common.h"
#include "gemm_core.h"
#include "type_rows_core.h"

using namespace pamnet;

namespace {

constexpr int DOUT = 128;
constexpr int TR = 64;                 // rows per tile (4 MFMA row tiles)
constexpr int MT = TR / 16;

template <int K>
struct Dims {
    static constexpr int G = (K + 15) / 16;     // 16-wide k groups
    static constexpr int KP = 16 * G;
    static constexpr int LDX = KP + 4;          // LDS row stride of the staged inputs (16-byte aligned rows)
};

// weight slice of a wave: columns [wc, wc+32) as two 16-column tiles, k zero-padded to KP
template <int K>
struct WSlice {
    float4 b[Dims<K>::G][2];
};
template <int K>
__device__ __forceinline__ void load_wslice(WSlice<K>& f, const float* __restrict__ W, int wc) {
    const int lane = threadIdx.x & 63, r16 = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int g = 0; g < Dims<K>::G; ++g)
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2) {
            const float* wp = W + (size_t)(wc + 16 * n2 + r16) * K;
            const int k0 = 16 * g + 4 * kg;
            f.b[g][n2] = make_float4(k0 < K ? wp[k0] : 0.f, k0 + 1 < K ? wp[k0 + 1] : 0.f, k0 + 2 < K ? wp[k0 + 2] : 0.f,
                                     k0 + 3 < K ? wp[k0 + 3] : 0.f);
        }
}

// rows [row0, row0+64) x K floats (contiguous in memory) -> xs[r][k], rows beyond `rows` zero; pad columns untouched
template <int K>
__device__ __forceinline__ void stage_rows(const float* __restrict__ x, int64_t row0, int64_t rows, float* xs) {
    const int64_t base = row0 * K;
    const int64_t lim = rows * K;
    for (int i = threadIdx.x; i < TR * K; i += 256) {
        const int r = i / K, k = i - r * K;
        xs[r * Dims<K>::LDX + k] = (base + i < lim) ? x[base + i] : 0.f;
    }
}
template <int K>
__device__ __forceinline__ void zero_pad(float* xs) {
    constexpr int PADW = Dims<K>::LDX - K;
    for (int i = threadIdx.x; i < TR * PADW; i += 256) {
        const int r = i / PADW, k = K + (i - r * PADW);
        xs[r * Dims<K>::LDX + k] = 0.f;
    }
}

// acc[m][n2] = x_tile * W_kind^T for this wave's 32 columns (no bias)
template <int K, bool TWO>
__device__ __forceinline__ void z_tile(const float* xs, const int* ks, const WSlice<K>& w0, const WSlice<K>& w1,
                                       f32x4 (&acc)[MT][2]) {
    const int lane = threadIdx.x & 63, r16 = lane & 15, kg = lane >> 4;
    acc_zero<MT>(acc);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const bool k1 = TWO && ks[16 * m + r16] != 0;
#pragma unroll
        for (int g = 0; g < Dims<K>::G; ++g) {
            const float4 a = *reinterpret_cast<const float4*>(xs + (16 * m + r16) * Dims<K>::LDX + 16 * g + 4 * kg);
            const float4 a0 = k1 ? f4zero() : a;
#pragma unroll
            for (int n2 = 0; n2 < 2; ++n2) {
                const float4 b = w0.b[g][n2];
                acc[m][n2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b.x, acc[m][n2], 0, 0, 0);
                acc[m][n2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b.y, acc[m][n2], 0, 0, 0);
                acc[m][n2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b.z, acc[m][n2], 0, 0, 0);
                acc[m][n2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b.w, acc[m][n2], 0, 0, 0);
            }
            if (TWO) {
                const float4 a1 = k1 ? a : f4zero();
#pragma unroll
                for (int n2 = 0; n2 < 2; ++n2) {
                    const float4 b = w1.b[g][n2];
                    acc[m][n2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b.x, acc[m][n2], 0, 0, 0);
                    acc[m][n2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b.y, acc[m][n2], 0, 0, 0);
                    acc[m][n2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b.z, acc[m][n2], 0, 0, 0);
                    acc[m][n2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b.w, acc[m][n2], 0, 0, 0);
                }
            }
        }
    }
}

__device__ __forceinline__ float4 ld4_or_zero(const float* p, int c4) {
    return p ? *reinterpret_cast<const float4*>(p + 4 * c4) : f4zero();
}

// envelope u(x) of the Bessel rows (layers/basic.py:36-51, p = 5) -- the same expression as basis.hip's rbf kernels
__device__ __forceinline__ float envelope_f(float x) {
    if (!(x < 1.0f)) return 0.0f;
    const float x2 = x * x, x5 = x2 * x2 * x;
    return 1.0f / x + x5 * (-21.0f + x * (35.0f - 15.0f * x));
}

// One embedding layer as the kernels see it (device-side image of pamnet_embed_job, include/pamnet_hip.h).
struct EJob {
    const float* x;                 // [rows, K] input rows, or null with `dist`
    const float* dist;              // [rows]: K = 16 Bessel rows u(d/c) sin(freq_n d/c) formed while staging (layers/basic.py:74-76)
    const float* freq;              // [16]
    float inv_cutoff;
    int act;
    int64_t rows;
    const int32_t* kind;
    const float *W0, *b0, *W1, *b1;
    float* out;                     // forward
    const float* gout;              // backward
    float* partial;                 // backward: per-workgroup partial gradients
    float* dx;                      // backward, K = 16 without dist: input gradient (nullable)
    float *dW0, *db0, *dW1, *db1, *dfreq;
    int code;                       // Code below
    int nblk;                       // workgroups of this job
};
enum Code { C16 = 0, C16_RBF, C18, C42, C42_TWO, C16_DX };

// `embeddings[x]` (models.py:107,140) rides in the same launches: forward rows table[idx], backward per-type row sums
struct TJob {
    const float* table;             // [n_types, 128]
    const int32_t* idx;             // [n]
    int64_t n;
    int n_types;
    float* out;                     // forward [n, 128]
    const float* g;                 // backward [n, 128]
    float4* partial;                // backward scratch (pamnet_reduce_scratch_bytes)
    float* dtable;                  // backward [n_types, 128]
    int nblk;
};
constexpr int MAXJ = 4;
struct EJobs {
    EJob job[MAXJ];
    TJob types;
    int n;
};

// input rows [row0, row0 + 64) of a job into xs (zero beyond `rows`); RBF: the Bessel rows from the distances, with the
// scaled distance of each row left in ds[] for the backward
// this thread's edge length of tile row threadIdx.x / 4 (rows beyond the job: 0, never used -- validity is the row index)
__device__ __forceinline__ float job_dist(const EJob& jb, int64_t row0) {
    const int64_t g = row0 + (threadIdx.x >> 2);
    return g < jb.rows ? jb.dist[g] : 0.0f;
}

// `have`: dpre holds this thread's edge length, fetched a tile ahead.  Whether a row exists is decided by its index alone,
// so a NaN / negative edge length reaches the envelope and the sine exactly as in the forward (bad geometry is not masked).
template <int K, bool RBF, bool BWD = false>
__device__ __forceinline__ void stage_job_rows(const EJob& jb, int64_t row0, float* xs, float* ds, bool have = false,
                                               float dpre = 0.0f) {
    if (!RBF) {
        stage_rows<K>(jb.x, row0, jb.rows, xs);
        return;
    }
    const int r = threadIdx.x >> 2, n0 = 4 * (threadIdx.x & 3);
    if (!have) dpre = job_dist(jb, row0);                          // (not fetched ahead)
    const bool ok = row0 + r < jb.rows;
    const float xr = ok ? dpre * jb.inv_cutoff : 1.0f;             // u(1) = 0: padded rows contribute nothing
    const float u = envelope_f(xr);
    const float4 f = *reinterpret_cast<const float4*>(jb.freq + n0);
    // the backward recomputes the rows on the hardware sine (gemm_core.h sin_turns); the forward's are sinf's floats
    const float4 sv = BWD ? make_float4(sin_turns(f.x * xr), sin_turns(f.y * xr), sin_turns(f.z * xr), sin_turns(f.w * xr))
                          : make_float4(sinf(f.x * xr), sinf(f.y * xr), sinf(f.z * xr), sinf(f.w * xr));
    *reinterpret_cast<float4*>(xs + r * Dims<K>::LDX + n0) =
        ok ? make_float4(u * sv.x, u * sv.y, u * sv.z, u * sv.w) : f4zero();
    if ((threadIdx.x & 3) == 0) ds[r] = xr;
}

template <int K, bool TWO, bool RBF>
__device__ __forceinline__ void embed_fwd_body(const EJob& jb, int bid, float* xs, float* Ds, int* ks, float* ds) {
    const int64_t rows = jb.rows;
    const int wc = (threadIdx.x >> 6) * 32;
    WSlice<K> w0, w1;
    load_wslice<K>(w0, jb.W0, wc);
    if (TWO) load_wslice<K>(w1, jb.W1, wc);
    const int c4 = threadIdx.x & 31;
    const float4 bias0 = ld4_or_zero(jb.b0, c4), bias1 = TWO ? ld4_or_zero(jb.b1, c4) : f4zero();
    const int act = jb.act;
    float* __restrict__ out = jb.out;
    zero_pad<K>(xs);
    if (threadIdx.x < TR) ks[threadIdx.x] = 0;
    const int64_t ntiles = (rows + TR - 1) / TR;
    for (int64_t tile = bid; tile < ntiles; tile += jb.nblk) {
        const int64_t row0 = tile * TR;
        __syncthreads();
        stage_job_rows<K, RBF>(jb, row0, xs, ds);
        if (TWO && threadIdx.x < TR) ks[threadIdx.x] = (row0 + threadIdx.x < rows) ? jb.kind[row0 + threadIdx.x] : 0;
        __syncthreads();
        f32x4 acc[MT][2];
        z_tile<K, TWO>(xs, ks, w0, w1, acc);
        acc_to_lds<MT>(acc, Ds, wc, load_bias2(nullptr, 0));
        __syncthreads();
        sweep_rows<TR>([&](int r, int c) {
            const int64_t g = row0 + r;
            if (g >= rows) return;
            const float4 z = f4add(lds4(Ds, r, c), (TWO && ks[r]) ? bias1 : bias0);
            stg4(out, g, DOUT, c, act ? f4silu(z) : z);
        });
    }
}

// partial layout per workgroup: [2 kinds][128][K] dW, [2][128] db, [16] dfreq
template <int K>
__host__ __device__ constexpr int partial_floats() { return 2 * DOUT * K + 2 * DOUT + 16; }

// DXM: 0 no input gradient, 1 dx [rows, 16] written, 2 the input rows are Bessel rows: d freq accumulated instead
// (layers/basic.py:76: d/d freq_n of u(x) sin(freq_n x) = u(x) x cos(freq_n x)) -- the [rows, 16] gradient never exists
template <int K, bool TWO, int DXM>
__device__ __forceinline__ void embed_bwd_body(const EJob& jb, int bid, float* xs, float* Ds, int* ks, float* ds) {
    constexpr int G = Dims<K>::G, LDX = Dims<K>::LDX;
    constexpr bool DX = DXM != 0;
    const int64_t rows = jb.rows;
    const float* __restrict__ gout = jb.gout;
    const int act = jb.act;
    const int lane = threadIdx.x & 63, r16 = lane & 15, kg = lane >> 4;
    const int wave = threadIdx.x >> 6, wc = wave * 32;
    WSlice<K> w0, w1;
    load_wslice<K>(w0, jb.W0, wc);
    if (TWO) load_wslice<K>(w1, jb.W1, wc);
    const int c4 = threadIdx.x & 31;
    const float4 bias0 = ld4_or_zero(jb.b0, c4), bias1 = TWO ? ld4_or_zero(jb.b1, c4) : f4zero();
    // dx = dz * W0: B[k = c][j = input k] for this wave's row tile; c runs over all 128 columns (K == 16 only)
    float4 wdx[DX ? DOUT / 16 : 1];
    if (DX) {
#pragma unroll
        for (int q = 0; q < DOUT / 16; ++q) {
            const float* wp = jb.W0 + (size_t)(16 * q + 4 * kg) * K + r16;
            wdx[q] = make_float4(wp[0], wp[K], wp[2 * K], wp[3 * K]);
        }
    }
    const float fn = DXM == 2 ? jb.freq[r16] : 0.f;
    float facc = 0.f;
    // dW accumulators: columns [wc, wc+32) as two 16-row (c) tiles x G k-tiles, per kind
    f32x4 dw0[2][G], dw1[TWO ? 2 : 1][TWO ? G : 1];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            dw0[a][g] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (TWO) dw1[a][g] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    float4 dbs0 = f4zero(), dbs1 = f4zero();              // bias gradients of this thread's float4 column, its rows
    zero_pad<K>(xs);
    if (threadIdx.x < TR) ks[threadIdx.x] = 0;
    const int64_t ntiles = (rows + TR - 1) / TR;
    // The 16-wide single-set layers are the long jobs (one row per global edge: 700 k rows at a PDBbind batch) and their
    // tiles were latency bound -- five barriers and two global-load round trips each, 1 TB/s of a 363 MB gradient stream.
    // The tile's slice of the output gradient (8 x 16 bytes per thread) is fetched one tile ahead, after the sweep that
    // consumed the previous one: the loads fly during the MFMA phases and the next tile's staging.
    constexpr bool PREF = K == 16 && !TWO;
    float4 pre[PREF ? TR / 8 : 1];
    auto prefetch = [&](int64_t t) {
        if constexpr (PREF) {
            const int c4p = threadIdx.x & 31, r0p = threadIdx.x >> 5;
#pragma unroll
            for (int i = 0; i < TR / 8; ++i) {
                const int64_t g = t * TR + r0p + 8 * i;
                pre[i] = (t < ntiles && g < rows) ? ldg4(gout, g, DOUT, c4p) : f4zero();
            }
        }
    };
    prefetch(bid);
    float dnext = 0.0f;
    if constexpr (DXM == 2) dnext = bid < ntiles ? job_dist(jb, (int64_t)bid * TR) : 0.0f;
    for (int64_t tile = bid; tile < ntiles; tile += jb.nblk) {
        const int64_t row0 = tile * TR;
        __syncthreads();
        stage_job_rows<K, DXM == 2, true>(jb, row0, xs, ds, DXM == 2, dnext);
        if constexpr (DXM == 2) dnext = tile + jb.nblk < ntiles ? job_dist(jb, (tile + jb.nblk) * TR) : 0.0f;
        if (TWO && threadIdx.x < TR) ks[threadIdx.x] = (row0 + threadIdx.x < rows) ? jb.kind[row0 + threadIdx.x] : 0;
        __syncthreads();
        {
            f32x4 acc[MT][2];
#ifdef EMBED_PROBE_NO_Z
            acc_zero<MT>(acc);
#else
            z_tile<K, TWO>(xs, ks, w0, w1, acc);
#endif
            acc_to_lds<MT>(acc, Ds, wc, load_bias2(nullptr, 0));
        }
        __syncthreads();
        // dz = g * act'(z), in place; bias gradients
        if constexpr (PREF) {
            const int c = threadIdx.x & 31, r0p = threadIdx.x >> 5;
#pragma unroll
            for (int i = 0; i < TR / 8; ++i) {
                const int r = r0p + 8 * i;
                float4 dz = f4zero();
                if (row0 + r < rows) {
                    const bool k1 = TWO && ks[r];
                    dz = pre[i];
                    if (act) dz = f4mul(dz, f4dsilu(f4add(lds4(Ds, r, c), k1 ? bias1 : bias0)));
                    if (k1) dbs1 = f4add(dbs1, dz);
                    else dbs0 = f4add(dbs0, dz);
                }
                st_lds4(Ds, r, c, dz);
            }
            prefetch(tile + jb.nblk);
        } else {
            // (the tile's eight gradient rows of this thread requested together, then consumed: row by row under `if (g < rows)`
            // each was a round trip of its own -- a one-tile workgroup of the two-set 42-wide layer is the launch at the QM9 batch)
            const int c = threadIdx.x & 31, r0p = threadIdx.x >> 5;
            float4 gr[TR / 8];
#pragma unroll
            for (int i = 0; i < TR / 8; ++i) {
                int64_t g = row0 + r0p + 8 * i;
                g = g < rows ? g : rows - 1;                  // (clamped: unconditional requests; rows > 0 inside the tile loop)
                gr[i] = ldg4(gout, g, DOUT, c);
            }
#pragma unroll
            for (int i = 0; i < TR / 8; ++i) {
                const int r = r0p + 8 * i;
                float4 dz = f4zero();
                if (row0 + r < rows) {
                    const bool k1 = TWO && ks[r];
                    dz = gr[i];
                    if (act) dz = f4mul(dz, f4dsilu(f4add(lds4(Ds, r, c), k1 ? bias1 : bias0)));
                    if (k1) dbs1 = f4add(dbs1, dz);
                    else dbs0 = f4add(dbs0, dz);
                }
                st_lds4(Ds, r, c, dz);
            }
        }
        __syncthreads();
        // dW[c][k] += sum_r dz[r][c] * x[r][k]: MFMA with the row index as the reduction dimension
#ifndef EMBED_PROBE_NO_DW
#pragma unroll 4
        for (int s = 0; s < TR / 4; ++s) {
            const int r = 4 * s + kg;
            const bool k1 = TWO && ks[r] != 0;
            float av[2], bv[G];
#pragma unroll
            for (int a = 0; a < 2; ++a) av[a] = Ds[r * LDT + wc + 16 * a + r16];
#pragma unroll
            for (int g = 0; g < G; ++g) bv[g] = xs[r * LDX + 16 * g + r16];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    dw0[a][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(k1 ? 0.f : av[a], bv[g], dw0[a][g], 0, 0, 0);
                    if (TWO) dw1[a][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(k1 ? av[a] : 0.f, bv[g], dw1[a][g], 0, 0, 0);
                }
        }
#endif
#ifdef EMBED_PROBE_NO_DX
        if (false) {
#else
        if (DX) {
#endif
            // dx[r][k] = sum_c dz[r][c] * W0[c][k]; wave w owns rows [16w, 16w+16) of the tile
            f32x4 ax = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < DOUT / 16; ++q) {
                const float4 a = *reinterpret_cast<const float4*>(Ds + (16 * wave + r16) * LDT + 16 * q + 4 * kg);
                ax = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, wdx[q].x, ax, 0, 0, 0);
                ax = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, wdx[q].y, ax, 0, 0, 0);
                ax = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, wdx[q].z, ax, 0, 0, 0);
                ax = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, wdx[q].w, ax, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (DXM == 1) {
                    const int64_t g = row0 + 16 * wave + 4 * kg + r;
                    if (g < rows) jb.dx[g * K + r16] = ax[r];
                } else {
                    const float xr = ds[16 * wave + 4 * kg + r];
                    facc += ax[r] * envelope_f(xr) * xr * cos_turns(fn * xr);
                }
            }
        }
    }
    // workgroup partial: dW tiles straight from the accumulators (element (c = wc + 16a + 4kg + r, k = 16g + r16)),
    // bias gradients through LDS (8 row groups x 128 columns per kind, summed in order)
    float* p = jb.partial + (int64_t)bid * partial_floats<K>();
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int k = 16 * g + r16;
            if (k < K) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = wc + 16 * a + 4 * kg + r;
                    p[c * K + k] = dw0[a][g][r];
                    if (TWO) p[DOUT * K + c * K + k] = dw1[a][g][r];
                }
            }
        }
    __syncthreads();
    float* red = Ds;                                        // [2][8][128], then [16 lane groups][16]
    const int rg = threadIdx.x >> 5;
    *reinterpret_cast<float4*>(red + rg * DOUT + 4 * c4) = dbs0;
    *reinterpret_cast<float4*>(red + 8 * DOUT + rg * DOUT + 4 * c4) = dbs1;
    if (DXM == 2) red[16 * DOUT + (threadIdx.x >> 4) * 16 + r16] = facc;
    __syncthreads();
    {
        const int kd = threadIdx.x >> 7, c = threadIdx.x & 127;
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += red[kd * 8 * DOUT + q * DOUT + c];
        p[2 * DOUT * K + kd * DOUT + c] = t;
    }
    if (DXM == 2 && threadIdx.x < 16) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += red[16 * DOUT + q * 16 + threadIdx.x];
        p[2 * DOUT * K + 2 * DOUT + threadIdx.x] = t;
    }
}

// 32 output elements x 8 slices of the workgroup partials per block; fixed-order tree over the slices (deterministic).
// Elements: [2][128][K] dW, [2][128] db, [16] dfreq; parts without a destination are skipped.
__device__ __forceinline__ void embed_reduce_body(const EJob& jb, int K, int bx, float (*sm)[33]) {
    const int per = 2 * DOUT * K + 2 * DOUT + 16;
    const int lane = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const int e = bx * 32 + lane;
    float* dst = nullptr;
    if (e < DOUT * K) dst = jb.dW0 + e;
    else if (e < 2 * DOUT * K) dst = jb.dW1 ? jb.dW1 + (e - DOUT * K) : nullptr;
    else if (e < 2 * DOUT * K + DOUT) dst = jb.db0 ? jb.db0 + (e - 2 * DOUT * K) : nullptr;
    else if (e < 2 * DOUT * K + 2 * DOUT) dst = jb.db1 ? jb.db1 + (e - 2 * DOUT * K - DOUT) : nullptr;
    else if (e < per) dst = jb.dfreq ? jb.dfreq + (e - 2 * DOUT * K - 2 * DOUT) : nullptr;
    // (eight partials requested before the first is added: as one load per addition the slice's 35-64 terms were a chain of
    // memory round trips -- 15 us of a QM9 step's critical path; the order of the additions is unchanged)
    float s = 0.f;
    if (dst)
        for (int b = slice; b < jb.nblk; b += 64) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = b + 8 * u < jb.nblk ? jb.partial[(int64_t)(b + 8 * u) * per + e] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (b + 8 * u < jb.nblk) s += v[u];
        }
    sm[slice][lane] = s;
    __syncthreads();
    if (slice == 0 && dst) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += sm[q][lane];
        *dst = t;
    }
}

__host__ __device__ inline int code_k(int code) { return code == C18 ? 18 : (code == C42 || code == C42_TWO ? 42 : 16); }

// forward rows of the type table: one float4 per thread
__device__ __forceinline__ void type_rows_fwd_body(const TJob& t, int bid) {
    const float4* __restrict__ tab = reinterpret_cast<const float4*>(t.table);
    float4* __restrict__ out = reinterpret_cast<float4*>(t.out);
    const int64_t total = t.n * (DOUT / 4);
    for (int64_t i = (int64_t)bid * 256 + threadIdx.x; i < total; i += (int64_t)t.nblk * 256) {
        const int64_t r = i >> 5;
        const int ty = t.idx[r];
        out[i] = (ty >= 0 && ty < t.n_types) ? tab[ty * (DOUT / 4) + (int)(i & 31)] : f4zero();
    }
}

constexpr int XS_FLOATS = TR * Dims<42>::LDX;

// All input embeddings of a forward in one launch: workgroups [0, nblk_0) run job 0, the next nblk_1 job 1, ...; the
// type-table rows take the last ones.  (Separately these are 5-7 launches of 4-18 us each on the critical path before
// the first layer; together they fill the machine once.)
__global__ __launch_bounds__(256, 2) void embed_multi_fwd_kernel(const EJobs J) {
    __shared__ __attribute__((aligned(16))) float xs[XS_FLOATS];
    __shared__ __attribute__((aligned(16))) float Ds[TR * LDT];
    __shared__ int ks[TR];
    __shared__ float ds[TR];
    int bid = blockIdx.x;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
        if (j >= J.n) break;
        const EJob& jb = J.job[j];
        if (bid < jb.nblk) {
            switch (jb.code) {
                case C16: embed_fwd_body<16, false, false>(jb, bid, xs, Ds, ks, ds); break;
                case C16_RBF: embed_fwd_body<16, false, true>(jb, bid, xs, Ds, ks, ds); break;
                case C18: embed_fwd_body<18, false, false>(jb, bid, xs, Ds, ks, ds); break;
                case C42: embed_fwd_body<42, false, false>(jb, bid, xs, Ds, ks, ds); break;
                default: embed_fwd_body<42, true, false>(jb, bid, xs, Ds, ks, ds); break;
            }
            return;
        }
        bid -= jb.nblk;
    }
    if (J.types.nblk) type_rows_fwd_body(J.types, bid);
}

__global__ __launch_bounds__(256, 2) void embed_multi_bwd_kernel(const EJobs J) {
    __shared__ __attribute__((aligned(16))) float xs[XS_FLOATS];
    __shared__ __attribute__((aligned(16))) float Ds[TR * LDT];
    __shared__ int ks[TR];
    __shared__ float ds[TR];
    int bid = blockIdx.x;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
        if (j >= J.n) break;
        const EJob& jb = J.job[j];
        if (bid < jb.nblk) {
            switch (jb.code) {
                case C16: embed_bwd_body<16, false, 0>(jb, bid, xs, Ds, ks, ds); break;
                case C16_DX: embed_bwd_body<16, false, 1>(jb, bid, xs, Ds, ks, ds); break;
                case C16_RBF: embed_bwd_body<16, false, 2>(jb, bid, xs, Ds, ks, ds); break;
                case C18: embed_bwd_body<18, false, 0>(jb, bid, xs, Ds, ks, ds); break;
                case C42: embed_bwd_body<42, false, 0>(jb, bid, xs, Ds, ks, ds); break;
                default: embed_bwd_body<42, true, 0>(jb, bid, xs, Ds, ks, ds); break;
            }
            return;
        }
        bid -= jb.nblk;
    }
    if (J.types.nblk)
        type_rows::grad_body(reinterpret_cast<const float4*>(J.types.g), J.types.idx, J.types.n, J.types.n_types, DOUT / 4,
                             J.types.partial, bid, J.types.nblk, reinterpret_cast<float4*>(Ds));
}

// blockIdx.y = job (the type table's finish is job n): fixed-order sums of the workgroup partials
__global__ __launch_bounds__(256) void embed_multi_reduce_kernel(const EJobs J) {
    __shared__ float sm[8][33];
    const int j = blockIdx.y;
    if (j < J.n) {
        const EJob& jb = J.job[j];
        const int K = code_k(jb.code);
        if ((int)blockIdx.x * 32 >= 2 * DOUT * K + 2 * DOUT + 16) return;
        embed_reduce_body(jb, K, blockIdx.x, sm);
    } else if (J.types.nblk && blockIdx.x == 0) {
        type_rows::finish_body(J.types.partial, J.types.nblk, J.types.n_types * (DOUT / 4),
                               reinterpret_cast<float4*>(J.types.dtable));
    }
}

inline int grid_for(int64_t rows, int cap) {
    const int64_t tiles = (rows + TR - 1) / TR;
    return (int)(tiles < 1 ? 1 : (tiles > cap ? cap : tiles));
}
constexpr int FWD_CAP = 1024, BWD_SLOTS = 512;
#ifndef EMBED_BWD_BIG_CAP
#define EMBED_BWD_BIG_CAP 256                                  // (tools/embed_probe.py builds other values)
#endif
// backward: the workgroups of a multi launch are co-resident up to 512 (two per CU); launch_multi deals them to the jobs by
// COST -- a 64-row tile of the two-set 42-wide layer (the triplet / pair rows) is ~5x the matrix work of a 16-wide tile: with
// "two tiles per workgroup" for every job that job's 138 workgroups were the launch (47-55 us for the QM9 batch, a 10x
// multiple of the bytes' floor) while the 16-wide jobs' workgroups had left long before.  A job alone gets all slots.
inline int bwd_grid_max(int64_t rows) { return grid_for(rows, BWD_SLOTS); }
inline int tile_cost(int code) { return code == C42_TWO ? 5 : (code == C42 ? 3 : (code == C16_RBF || code == C16_DX ? 2 : 1)); }

// host: pamnet_embed_job -> EJob (validated); bwd selects the backward variant and grid
int make_job(const pamnet_embed_job& h, bool bwd, EJob* o) {
    const int64_t K = h.K;
    if (h.rows < 0 || (K != 16 && K != 18 && K != 42)) return PAMNET_EINVAL;
    if (h.dist && (K != 16 || h.kind || h.x || !(h.cutoff > 0.f))) return PAMNET_EINVAL;
    if (h.dx && (h.kind || K != 16 || h.dist)) return PAMNET_EINVAL;   // input gradients only for the plain K = 16 layer
    if (!h.W0 || (h.kind && !h.W1) || (h.W1 && h.rows > 0 && !h.kind)) return PAMNET_ENULL;
    if (h.rows > 0 && !h.x && !h.dist) return PAMNET_ENULL;
    if (h.dist && !h.freq) return PAMNET_ENULL;
    if (!bwd && h.rows > 0 && !h.out) return PAMNET_ENULL;
    if (bwd && (!h.dW0 || !h.partial || (h.rows > 0 && !h.gout) || (h.W1 && !h.dW1) || (h.dist && !h.dfreq))) return PAMNET_ENULL;
    o->x = h.x, o->dist = h.dist, o->freq = h.freq, o->inv_cutoff = h.dist ? 1.0f / h.cutoff : 0.f, o->act = h.act;
    o->rows = h.rows, o->kind = h.kind, o->W0 = h.W0, o->b0 = h.b0, o->W1 = h.W1, o->b1 = h.b1;
    o->out = h.out, o->gout = h.gout, o->partial = h.partial, o->dx = h.dx;
    o->dW0 = h.dW0, o->db0 = h.db0, o->dW1 = h.dW1, o->db1 = h.db1, o->dfreq = h.dfreq;
    o->code = K == 18 ? C18 : (K == 42 ? (h.W1 ? C42_TWO : C42) : (h.dist ? C16_RBF : ((bwd && h.dx) ? C16_DX : C16)));
    o->nblk = bwd ? bwd_grid_max(h.rows) : grid_for(h.rows, FWD_CAP);     // (backward: re-dealt by cost in launch_multi)
    return PAMNET_OK;
}

int make_types(const pamnet_type_rows_job* h, bool bwd, TJob* o) {
    o->nblk = 0;
    if (!h) return PAMNET_OK;
    if (h->n < 0 || h->n_types < 1 || h->n_types > type_rows::TYPE_MAX) return PAMNET_EINVAL;
    if (!bwd && (!h->table || (h->n > 0 && (!h->idx || !h->out)))) return PAMNET_ENULL;
    if (bwd && (!h->dtable || !h->scratch || (h->n > 0 && (!h->idx || !h->g)))) return PAMNET_ENULL;
    o->table = h->table, o->idx = h->idx, o->n = h->n, o->n_types = (int)h->n_types, o->out = h->out, o->g = h->g;
    o->partial = static_cast<float4*>(h->scratch), o->dtable = h->dtable;
    if (bwd) o->nblk = type_rows::blocks_for_rows(h->n);
    else o->nblk = (int)(h->n == 0 ? 0 : (ceil_div(h->n * (DOUT / 4), 256) < 256 ? ceil_div(h->n * (DOUT / 4), 256) : 256));
    return PAMNET_OK;
}

int launch_multi(const pamnet_embed_job* jobs, int32_t n_jobs, const pamnet_type_rows_job* types, bool bwd, hipStream_t st) {
    if (n_jobs < 0 || n_jobs > MAXJ || (n_jobs > 0 && !jobs)) return PAMNET_EINVAL;
    EJobs J;
    J.n = 0;
    int grid = 0;
    for (int j = 0; j < n_jobs; ++j) {
        if (!bwd && jobs[j].rows == 0) continue;                          // nothing to write
        const int rc = make_job(jobs[j], bwd, &J.job[J.n]);
        if (rc != PAMNET_OK) return rc;
        grid += J.job[J.n].nblk;
        ++J.n;
    }
    const int rc = make_types(types, bwd, &J.types);
    if (rc != PAMNET_OK) return rc;
    if (bwd && J.n > 0) {
        // Workgroups of the backward launch.  Launches of a few tiles per workgroup (a QM9 batch: 860 tiles) are dealt by COST:
        // the matrix work of a tile is what a workgroup's time is there, and with "two tiles per workgroup" for every job the
        // two-set 42-wide layer's workgroups were the launch.  Jobs of hundreds of thousands of rows (a PDBbind batch: 10 800
        // tiles of Bessel rows) are bound by their gradient stream -- a tile costs the same whatever its layer, the cost model
        // misdeals them (measured: 283 -> 751 us) -- and keep two tiles per workgroup up to 256.  Never more workgroups than a
        // job's tiles or its scratch (bwd_grid_max).
        int64_t all_tiles = 0, total = 0;
        for (int j = 0; j < J.n; ++j) {
            const int64_t tiles = (J.job[j].rows + TR - 1) / TR;
            all_tiles += tiles;
            total += tiles * tile_cost(J.job[j].code);
        }
        const bool by_cost = all_tiles <= 4 * BWD_SLOTS;
        const int64_t budget = BWD_SLOTS - J.types.nblk > 64 ? BWD_SLOTS - J.types.nblk : 64;
        grid = 0;
        for (int j = 0; j < J.n; ++j) {
            const int64_t tiles = (J.job[j].rows + TR - 1) / TR;
            int64_t want = grid_for((J.job[j].rows + 1) / 2, EMBED_BWD_BIG_CAP);
            if (by_cost) want = total > 0 ? (tiles * tile_cost(J.job[j].code) * budget + total - 1) / total : 1;
            want = want < 1 ? 1 : want;
            J.job[j].nblk = (int)(want < J.job[j].nblk ? want : J.job[j].nblk);
            grid += J.job[j].nblk;
        }
    }
    grid += J.types.nblk;
    if (grid == 0) return PAMNET_OK;
    if (!bwd) {
        hipLaunchKernelGGL(embed_multi_fwd_kernel, dim3(grid), dim3(256), 0, st, J);
        PAMNET_LAUNCH_CHECK();
        return PAMNET_OK;
    }
    hipLaunchKernelGGL(embed_multi_bwd_kernel, dim3(grid), dim3(256), 0, st, J);
    PAMNET_LAUNCH_CHECK();
    int gx = 1;
    for (int j = 0; j < J.n; ++j) {
        const int per = 2 * DOUT * code_k(J.job[j].code) + 2 * DOUT + 16;
        gx = (per + 31) / 32 > gx ? (per + 31) / 32 : gx;
    }
    hipLaunchKernelGGL(embed_multi_reduce_kernel, dim3(gx, J.n + (J.types.nblk ? 1 : 0)), dim3(256), 0, st, J);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

}  // namespace

// scratch floats for the backward of one layer: (largest grid a launch may give the job) * (2*128*K + 2*128 + 16)
extern "C" int pamnet_embed_scratch_floats(int64_t rows, int64_t K, int64_t* floats) {
    if (rows < 0 || K <= 0 || !floats) return PAMNET_EINVAL;
    *floats = (int64_t)bwd_grid_max(rows) * (2 * DOUT * K + 2 * DOUT + 16);
    return PAMNET_OK;
}

// All input embeddings of a forward / backward in one (two) launch(es); see pamnet_embed_job in pamnet_hip.h.
extern "C" int pamnet_embed_multi_fwd_f32(const pamnet_embed_job* jobs, int32_t n_jobs, const pamnet_type_rows_job* types,
                                          pamnet_stream_t stream) {
    return launch_multi(jobs, n_jobs, types, false, as_stream(stream));
}

extern "C" int pamnet_embed_multi_bwd_f32(const pamnet_embed_job* jobs, int32_t n_jobs, const pamnet_type_rows_job* types,
                                          pamnet_stream_t stream) {
    return launch_multi(jobs, n_jobs, types, true, as_stream(stream));
}

extern "C" int pamnet_embed_fwd_f32(const float* x, int64_t rows, int64_t K, const int32_t* kind, const float* W0,
                                    const float* b0, const float* W1, const float* b1, int32_t act, float* out,
                                    pamnet_stream_t stream) {
    if (rows < 0 || (K != 16 && K != 18 && K != 42)) return PAMNET_EINVAL;
    if (rows == 0) return PAMNET_OK;
    if (!x || !W0 || !out || (kind && !W1) || (W1 && !kind)) return PAMNET_ENULL;
    pamnet_embed_job jb = {};
    jb.x = x, jb.rows = rows, jb.K = (int32_t)K, jb.kind = kind, jb.W0 = W0, jb.b0 = b0, jb.W1 = W1, jb.b1 = b1, jb.act = act;
    jb.out = out;
    return launch_multi(&jb, 1, nullptr, false, as_stream(stream));
}

extern "C" int pamnet_embed_bwd_f32(const float* x, int64_t rows, int64_t K, const int32_t* kind, const float* W0,
                                    const float* b0, const float* W1, const float* b1, int32_t act, const float* gout,
                                    float* dW0, float* db0, float* dW1, float* db1, float* dx, float* partial,
                                    pamnet_stream_t stream) {
    if (rows < 0 || (K != 16 && K != 18 && K != 42)) return PAMNET_EINVAL;
    if (rows > 0 && (!x || !gout)) return PAMNET_ENULL;   // rows == 0: gradients are written as zeros
    if (!W0 || !dW0 || !partial || (kind && !W1) || (W1 && (!dW1 || (rows > 0 && !kind)))) return PAMNET_ENULL;
    if (dx && (kind || K != 16)) return PAMNET_EINVAL;     // input gradients only for the single-kind K = 16 (rbf) layer
    pamnet_embed_job jb = {};
    jb.x = x, jb.rows = rows, jb.K = (int32_t)K, jb.kind = kind, jb.W0 = W0, jb.b0 = b0, jb.W1 = W1, jb.b1 = b1, jb.act = act;
    jb.gout = gout, jb.dW0 = dW0, jb.db0 = db0, jb.dW1 = dW1, jb.db1 = db1, jb.dx = dx, jb.partial = partial;
    return launch_multi(&jb, 1, nullptr, true, as_stream(stream));
}
