// Batched weight-gradient GEMMs:  dW[128,128] = dZ[rows,128]^T * A[rows,128],  db[128] = column sums of dZ.
//
// The backward of every Linear in a PAMNet layer needs one of these; the reference launches one GEMM + one reduce per
// Linear (~60 per layer pair, most with only 2-4 K rows: launch-latency bound).  Here all jobs of a layer are ONE
// launch: grid = (split, jobs).  Each workgroup reduces a contiguous chunk of rows into a full 128x128 fp32 tile with
// v_mfma_f32_16x16x4_f32 (the row index is the MFMA k dimension), writes it to a partial buffer, and a second small
// kernel sums the partials in a fixed order -> deterministic, atomics-free.
// A may be given as a pre-activation (a_mode 1: A = SiLU(Z_prev) applied while staging), so activations that are a
// pure SiLU of a saved z are never stored twice.
#include "common.h"
#include "gemm_core.h"

using namespace pamnet;

namespace {

constexpr int MAXJ = 24;
constexpr int RB = 32;            // rows staged per step
constexpr int LDW = 144;          // LDS leading dim: 144 mod 32 = 16 -> conflict-free ds_read_b32 fragment reads

struct WJob {
    const float* dZ;
    const float* A;
    float* dW;
    float* db;        // may be null
    int64_t rows;
    int ld_dz, ld_a, ld_dw, a_mode;
};
struct WBatch {
    WJob job[MAXJ];
};

__host__ __device__ inline int job_split(int64_t rows, int split) {
    const int64_t want = (rows + 511) / 512;
    return (int)(want < 1 ? 1 : (want > split ? split : want));
}

__global__ __launch_bounds__(WG) void wgrad_kernel(WBatch batch, int split, float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float lds[2 * RB * LDW];
    float* Zs = lds;
    float* As = lds + RB * LDW;
    const WJob jb = batch.job[blockIdx.y];
    const int s = blockIdx.x;
    const int js = job_split(jb.rows, split);               // short jobs use (and later sum) fewer partial slots
    if (s >= js) return;
    const int64_t chunk = ((jb.rows + js - 1) / js + RB - 1) / RB * RB;
    const int64_t beg = (int64_t)s * chunk;
    const int64_t end = beg + chunk < jb.rows ? beg + chunk : jb.rows;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r16 = lane & 15, kg = lane >> 4;
    const int i0 = (w >> 1) * 64, j0 = (w & 1) * 64;      // wave tile: dW rows (n) [i0, i0+64), cols (k) [j0, j0+64)
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    double colsum = 0.0;      // threads 0..127: bias gradient of column threadIdx.x (fp64: long, cancelling sums)

    // register double buffer: the next 32-row block is in flight from L2/HBM while the MFMAs chew on the current one
    const int c4 = threadIdx.x & 31, rr = threadIdx.x >> 5;
    float4 zr[RB / 8], ar[RB / 8];
    auto fetch = [&](int64_t r0) {
#pragma unroll
        for (int i = 0; i < RB / 8; ++i) {
            const int64_t g = r0 + rr + 8 * i;
            zr[i] = f4zero();
            ar[i] = f4zero();
            if (g < end) {
                zr[i] = ldg4(jb.dZ, g, jb.ld_dz, c4);
                ar[i] = ldg4(jb.A, g, jb.ld_a, c4);
            }
        }
    };
    if (beg < end) fetch(beg);
    for (int64_t r0 = beg; r0 < end; r0 += RB) {
        // stage dZ[r0:r0+32, :] and A[r0:r0+32, :] (coalesced float4, zero padded)
#pragma unroll
        for (int i = 0; i < RB / 8; ++i) {
            const int r = rr + 8 * i;
            float4 a = ar[i];
            if (jb.a_mode == 1 && r0 + r < end) a = f4silu(a);
            *reinterpret_cast<float4*>(Zs + r * LDW + 4 * c4) = zr[i];
            *reinterpret_cast<float4*>(As + r * LDW + 4 * c4) = a;
        }
        __syncthreads();
        if (r0 + RB < end) fetch(r0 + RB);
        if (jb.db && threadIdx.x < 128) {
#pragma unroll 8
            for (int r = 0; r < RB; ++r) colsum += (double)Zs[r * LDW + threadIdx.x];
        }
#pragma unroll
        for (int st = 0; st < RB / 4; ++st) {
            const int r = 4 * st + kg;
            float za[4], ab[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                za[t] = Zs[r * LDW + i0 + 16 * t + r16];
                ab[t] = As[r * LDW + j0 + 16 * t + r16];
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(za[a], ab[b], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
    }
    // partial[job][split][128][128] (+ bias partial behind it)
    float* out = partial + ((int64_t)blockIdx.y * split + s) * (DIM * DIM + DIM);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                out[(i0 + 16 * a + kg * 4 + r) * DIM + j0 + 16 * b + r16] = acc[a][b][r];
    if (threadIdx.x < 128) out[DIM * DIM + threadIdx.x] = (float)colsum;
}

__global__ __launch_bounds__(WG) void wgrad_reduce_kernel(WBatch batch, int split, const float* __restrict__ partial) {
    const WJob jb = batch.job[blockIdx.y];
    const float* base = partial + (int64_t)blockIdx.y * split * (DIM * DIM + DIM);
    const int t = blockIdx.x * WG + threadIdx.x;                 // 0 .. 128*128 + 128
    const int js = job_split(jb.rows, split);
    if (t < DIM * DIM) {
        float s = 0.f;
        for (int q = 0; q < js; ++q) s += base[(int64_t)q * (DIM * DIM + DIM) + t];
        jb.dW[(int64_t)(t >> 7) * jb.ld_dw + (t & 127)] = s;
    } else if (t < DIM * DIM + DIM && jb.db) {
        double s = 0.0;
        for (int q = 0; q < js; ++q) s += (double)base[(int64_t)q * (DIM * DIM + DIM) + t];
        jb.db[t - DIM * DIM] = (float)s;
    }
}

}  // namespace

// jobs described by parallel host arrays (njobs <= 24).  partial: njobs * split * (128*128 + 128) floats of scratch.
extern "C" int pamnet_wgrad_batched_f32(int64_t njobs, const float* const* dZ, const int64_t* ld_dz,
                                        const float* const* A, const int64_t* ld_a, const int32_t* a_mode,
                                        const int64_t* rows, float* const* dW, const int64_t* ld_dw, float* const* db,
                                        int64_t split, float* partial, pamnet_stream_t stream) {
    if (njobs < 0 || njobs > MAXJ || split < 1 || split > 1024) return PAMNET_EINVAL;
    if (njobs == 0) return PAMNET_OK;
    if (!dZ || !ld_dz || !A || !ld_a || !a_mode || !rows || !dW || !ld_dw || !db || !partial) return PAMNET_ENULL;
    WBatch b;
    for (int j = 0; j < njobs; ++j) {
        if (!dZ[j] || !A[j] || !dW[j]) return PAMNET_ENULL;
        b.job[j] = WJob{dZ[j], A[j], dW[j], db[j], rows[j], (int)ld_dz[j], (int)ld_a[j], (int)ld_dw[j], a_mode[j]};
    }
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(wgrad_kernel, dim3((unsigned)split, (unsigned)njobs), dim3(WG), 0, st, b, (int)split, partial);
    PAMNET_LAUNCH_CHECK();
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((DIM * DIM + DIM + WG - 1) / WG, (unsigned)njobs), dim3(WG), 0, st, b,
                       (int)split, partial);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}
