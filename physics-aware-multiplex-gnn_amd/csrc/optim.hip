// Optimiser tail of the training step as ONE pass over the flat fp32 buffers (train.FlatParams):
//   clip_grad_norm_(max_norm, L2)  ->  Adam(lr, betas, eps, weight_decay, amsgrad=False)  ->  EMA  [-> zero the gradient]
// i.e. main_qm9.py:111-112,116 + utils/ema.py:13-20.  The reference issues ~10 small launches per parameter tensor
// (~3 000 per step at L=6); here every element is read and written once: 5 reads + 4 (5) writes of n floats, HBM-bound.
// The pre-clip gradient norm is read from device memory (no host round trip between the norm reduction and the update).
#include "common.h"

namespace {

struct AdamArgs {
    float lr, beta1, beta2, eps, weight_decay;
    float bias1, bias2_sqrt;          // 1 - beta1^t,  sqrt(1 - beta2^t)
    float ema_decay, max_norm;
    int zero_grad;
};

template <bool EMA>
__global__ __launch_bounds__(256) void adam_ema_kernel(float4* __restrict__ p, float4* __restrict__ g,
                                                       float4* __restrict__ m, float4* __restrict__ v,
                                                       float4* __restrict__ shadow, int64_t n4,
                                                       const float* __restrict__ grad_norm,
                                                       const double* __restrict__ sumsq_partials,
                                                       float* __restrict__ norm_out, AdamArgs a) {
    float clip = 1.0f;
    if (sumsq_partials) {
        // 256 fp64 partial sums of squares (pamnet_sumsq_partials_f32): every workgroup adds them in index order itself
        // (2 KB of L2 hits) -- no finish launch, no cross-workgroup synchronisation
        __shared__ double red[256];
        red[threadIdx.x] = sumsq_partials[threadIdx.x];
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        const float nrm = (float)sqrt(red[0]);
        if (norm_out && blockIdx.x == 0 && threadIdx.x == 0) norm_out[0] = nrm;
        clip = fminf(a.max_norm / (nrm + 1e-6f), 1.0f);
    } else if (grad_norm) {
        clip = fminf(a.max_norm / (grad_norm[0] + 1e-6f), 1.0f);                  // torch clip_grad_norm_
    }
    const float step = a.lr / a.bias1;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 pp = p[i], gg = g[i], mm = m[i], vv = v[i], ss = EMA ? shadow[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        float* pf = &pp.x;
        float* gf = &gg.x;
        float* mf = &mm.x;
        float* vf = &vv.x;
        float* sf = &ss.x;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float gr = gf[c] * clip;
            if (a.weight_decay != 0.0f) gr = fmaf(a.weight_decay, pf[c], gr);
            mf[c] = fmaf(a.beta1, mf[c], (1.0f - a.beta1) * gr);                      // exp_avg.lerp_(grad, 1-beta1)
            vf[c] = fmaf(a.beta2, vf[c], (1.0f - a.beta2) * gr * gr);                 // exp_avg_sq
            const float denom = sqrtf(vf[c]) / a.bias2_sqrt + a.eps;
            pf[c] -= step * (mf[c] / denom);
            if (EMA) sf[c] = fmaf(a.ema_decay, sf[c], (1.0f - a.ema_decay) * pf[c]);
        }
        p[i] = pp;
        m[i] = mm;
        v[i] = vv;
        if (EMA) shadow[i] = ss;
        if (a.zero_grad) g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// n must be a multiple of 4 and the buffers 16-byte aligned (train.FlatParams pads every tensor to 64 floats).
// step_count = t >= 1 of this update.  grad_norm: device scalar holding the pre-clip L2 norm (nullable = no clipping).
int adam_launch(float* p, float* g, float* m, float* v, float* shadow, int64_t n, float lr, float beta1, float beta2,
                float eps, float weight_decay, int64_t step_count, float ema_decay, const float* grad_norm,
                const double* sumsq_partials, float* norm_out, float max_norm, int32_t zero_grad, pamnet_stream_t stream) {
    if (n < 0 || (n & 3) || step_count < 1) return PAMNET_EINVAL;
    if (n == 0) return PAMNET_OK;
    if (!p || !g || !m || !v) return PAMNET_ENULL;
    AdamArgs a;
    a.lr = lr, a.beta1 = beta1, a.beta2 = beta2, a.eps = eps, a.weight_decay = weight_decay;
    a.bias1 = (float)(1.0 - pow((double)beta1, (double)step_count));
    a.bias2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step_count));
    a.ema_decay = ema_decay, a.max_norm = max_norm, a.zero_grad = zero_grad;
    const int64_t n4 = n / 4;
    int64_t blocks = ceil_div(n4, 256);
    if (blocks > 4096) blocks = 4096;
    if (shadow)
        hipLaunchKernelGGL(adam_ema_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), (float4*)p,
                           (float4*)g, (float4*)m, (float4*)v, (float4*)shadow, n4, grad_norm, sumsq_partials, norm_out, a);
    else                                   // no EMA (main_pdbbind.py / main_rna_puzzles.py): 4 reads + 3 (4) writes per element
        hipLaunchKernelGGL(adam_ema_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), (float4*)p,
                           (float4*)g, (float4*)m, (float4*)v, (float4*)nullptr, n4, grad_norm, sumsq_partials, norm_out, a);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}
}  // namespace

extern "C" int pamnet_adam_ema_f32(float* p, float* g, float* m, float* v, float* shadow, int64_t n, float lr,
                                   float beta1, float beta2, float eps, float weight_decay, int64_t step_count,
                                   float ema_decay, const float* grad_norm, float max_norm, int32_t zero_grad,
                                   pamnet_stream_t stream) {
    return adam_launch(p, g, m, v, shadow, n, lr, beta1, beta2, eps, weight_decay, step_count, ema_decay, grad_norm,
                       nullptr, nullptr, max_norm, zero_grad, stream);
}

// Same update with the gradient norm taken from the 256 fp64 partial sums of squares of pamnet_sumsq_partials_f32
// (clip on); norm_out[0] (nullable) receives the pre-clip L2 norm.
extern "C" int pamnet_adam_ema_norm_f32(float* p, float* g, float* m, float* v, float* shadow, int64_t n, float lr,
                                        float beta1, float beta2, float eps, float weight_decay, int64_t step_count,
                                        float ema_decay, const double* sumsq_partials, float* norm_out, float max_norm,
                                        int32_t zero_grad, pamnet_stream_t stream) {
    if (!sumsq_partials) return PAMNET_ENULL;
    return adam_launch(p, g, m, v, shadow, n, lr, beta1, beta2, eps, weight_decay, step_count, ema_decay, nullptr,
                       sumsq_partials, norm_out, max_norm, zero_grad, stream);
}
