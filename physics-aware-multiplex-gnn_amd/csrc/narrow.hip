// C entry points of the narrow-width (dim = 16 / 32 / 64) row kernels; the kernels themselves are in narrow_core.h
// (shared with the narrow-width layer-stack engine, narrow_engine.hip).
#include "narrow_core.h"

extern "C" int pamnet_narrow_blocks(int64_t rows, int64_t* blocks) {
    if (rows < 0) return PAMNET_EINVAL;
    if (!blocks) return PAMNET_ENULL;
    *blocks = 256;                                           // backward kernels: at most one workgroup per CU
    return PAMNET_OK;
}

extern "C" int pamnet_narrow_global_fwd_f32(const float* e, int64_t m, int64_t d, const int32_t* tgt, const int32_t* src,
                                            const float* P, const float* We, int64_t ldwe, const float* bias,
                                            const float* Wea, int64_t ldwea, float* msg, pamnet_stream_t stream) {
    if (m < 0 || !width_ok(d) || ldwe < d || ldwea < d || (ldwe & 3) || (ldwea & 3)) return PAMNET_EINVAL;
    if (m == 0) return PAMNET_OK;
    if (!e || !tgt || !src || !P || !We || !bias || !Wea || !msg) return PAMNET_ENULL;
    hipStream_t st = as_stream(stream);
    const int grid = grid_for(m, fwd_per_cu(d));
#define CALL(DD)                                                                                                     \
    {                                                                                                                \
        const size_t lds = 2 * wimg_bytes(DD);                                                      \
        hipLaunchKernelGGL((nglobal_fwd_kernel<DD>), dim3(grid), dim3(NWG), lds, st, e, m, tgt, src, P, We, (int)ldwe, \
                           bias, Wea, (int)ldwea, msg);                                                              \
    }
    NARROW_DISPATCH(d, CALL)
#undef CALL
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_narrow_global_bwd_f32(const float* e, int64_t m, int64_t d, const int32_t* tgt, const int32_t* src,
                                            const float* P, const float* We, int64_t ldwe, const float* bias,
                                            const float* Wea, int64_t ldwea, const float* dagg, float* dz, float* de,
                                            float* partial, float* dWe, float* dWea, float* db,
                                            pamnet_stream_t stream) {
    if (m <= 0 || !width_ok(d) || ldwe < d || ldwea < d || (ldwe & 3) || (ldwea & 3)) return PAMNET_EINVAL;
    if (!e || !tgt || !src || !P || !We || !bias || !Wea || !dagg || !dz || !de || !partial || !dWe || !dWea || !db)
        return PAMNET_ENULL;
    hipStream_t st = as_stream(stream);
    const int grid = grid_for(m, 1, bwd_waves((int)d));
    const int stride = (int)(2 * d * d + d);
#define CALL(DD)                                                                                                     \
    {                                                                                                                \
        const size_t lds = 4 * wimg_bytes(DD) + bwd_waves(DD) * 16 * (DD + 4) * sizeof(float);      \
        hipError_t e_ = allow_lds(nglobal_bwd_kernel<DD>, lds);                                                      \
        if (e_ != hipSuccess) return (int)e_;                                                                        \
        hipLaunchKernelGGL((nglobal_bwd_kernel<DD>), dim3(grid), dim3(64 * bwd_waves(DD)), lds, st, e, m, tgt, src, P, We, (int)ldwe, \
                           bias, Wea, (int)ldwea, dagg, dz, de, partial, stride, 0);                                    \
    }
    NARROW_DISPATCH(d, CALL)
#undef CALL
    PAMNET_LAUNCH_CHECK();
    // dWe and dWea are separate outputs: two reduce launches over the same partial rows (matrix 0 / matrix 1 + bias)
    const int per = (int)(d * d);
    hipLaunchKernelGGL(narrow_reduce_kernel, dim3((per + 63) / 64), dim3(64, 8), 0, st, partial, grid, stride, 1, (int)d,
                       (int)d, (int)d, 0, dWe, (float*)nullptr);
    PAMNET_LAUNCH_CHECK();
    hipLaunchKernelGGL(narrow_reduce_kernel, dim3((per + (int)d + 63) / 64), dim3(64, 8), 0, st, partial + per, grid,
                       stride, 1, (int)d, (int)d, (int)d, (int)d, dWea, db);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_narrow_mlp2_fwd_f32(const float* x, int64_t m, int64_t d, const float* W1, const float* b1,
                                          const float* W2, const float* b2, int32_t res_x, const float* res, float* y,
                                          pamnet_stream_t stream) {
    if (m < 0 || !width_ok(d)) return PAMNET_EINVAL;
    if (m == 0) return PAMNET_OK;
    if (!x || !W1 || !b1 || !W2 || !b2 || !y) return PAMNET_ENULL;
    hipStream_t st = as_stream(stream);
    const int grid = grid_for(m, fwd_per_cu(d));
#define CALL(DD)                                                                                                  \
    {                                                                                                             \
        const size_t lds = 2 * wimg_bytes(DD) + 4 * 16 * (DD + 4) * sizeof(float);               \
        hipLaunchKernelGGL((nmlp2_fwd_kernel<DD>), dim3(grid), dim3(NWG), lds, st, x, m, W1, b1, W2, b2, (int)res_x, res, y); \
    }
    NARROW_DISPATCH(d, CALL)
#undef CALL
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_narrow_mlp2_bwd_f32(const float* x, int64_t m, int64_t d, const float* W1, const float* b1,
                                          const float* W2, const float* b2, const float* dy, int32_t res_x, float* dx,
                                          float* partial, float* dW /* [2, d, d] */, float* db /* [2, d] */,
                                          pamnet_stream_t stream) {
    if (m <= 0 || !width_ok(d)) return PAMNET_EINVAL;
    if (!x || !W1 || !b1 || !W2 || !b2 || !dy || !partial || !dW || !db) return PAMNET_ENULL;
    hipStream_t st = as_stream(stream);
    const int grid = grid_for(m, 1, bwd_waves((int)d));
    const int stride = (int)(2 * d * d + 2 * d);
#define CALL(DD)                                                                                                       \
    {                                                                                                                  \
        const size_t lds = 4 * wimg_bytes(DD) + bwd_waves(DD) * 16 * (DD + 4) * sizeof(float);        \
        hipError_t e_ = allow_lds(nmlp2_bwd_kernel<DD>, lds);                                                          \
        if (e_ != hipSuccess) return (int)e_;                                                                          \
        hipLaunchKernelGGL((nmlp2_bwd_kernel<DD>), dim3(grid), dim3(64 * bwd_waves(DD)), lds, st, x, m, W1, b1, W2, b2, dy, (int)res_x, dx, \
                           partial, stride, 0);                                                                           \
    }
    NARROW_DISPATCH(d, CALL)
#undef CALL
    PAMNET_LAUNCH_CHECK();
    const int total = (int)(2 * d * d + 2 * d);
    hipLaunchKernelGGL(narrow_reduce_kernel, dim3((total + 63) / 64), dim3(64, 8), 0, st, partial, grid, stride, 2, (int)d,
                       (int)d, (int)d, (int)(2 * d), dW, db);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_narrow_linear_fwd_f32(const float* x, int64_t m, int64_t d, const float* W, int64_t ldw,
                                            const float* b, int32_t act, float* y, int64_t ldy,
                                            pamnet_stream_t stream) {
    if (m < 0 || !width_ok(d) || ldw < d || ldy < d) return PAMNET_EINVAL;
    if (m == 0) return PAMNET_OK;
    if (!x || !W || !y) return PAMNET_ENULL;
    hipStream_t st = as_stream(stream);
    const int grid = grid_for(m, fwd_per_cu(d));
#define CALL(DD)                                                                                                       \
    {                                                                                                                  \
        const size_t lds = wimg_bytes(DD);                                                                         \
        hipLaunchKernelGGL((nlinear_fwd_kernel<DD>), dim3(grid), dim3(NWG), lds, st, x, m, W, (int)ldw, b, (int)act, y, ldy); \
    }
    NARROW_DISPATCH(d, CALL)
#undef CALL
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

/* dx optional; accumulate != 0: dx += ...; dW [d, d] dense, db [d] (null when the layer has no bias) */
extern "C" int pamnet_narrow_linear_bwd_f32(const float* x, int64_t m, int64_t d, const float* W, int64_t ldw,
                                            const float* b, int32_t act, const float* dy, int64_t lddy, float* dx,
                                            int32_t accumulate, float* partial, float* dW, float* db,
                                            pamnet_stream_t stream) {
    if (m <= 0 || !width_ok(d) || ldw < d || lddy < d) return PAMNET_EINVAL;
    if (!x || !W || !dy || !partial || !dW) return PAMNET_ENULL;
    hipStream_t st = as_stream(stream);
    const int grid = grid_for(m, 1, lin_bwd_waves((int)d));
    const int stride = (int)(d * d + d);
#define CALL(DD)                                                                                                        \
    {                                                                                                                   \
        const size_t lds = 2 * wimg_bytes(DD) + lin_bwd_waves(DD) * 16 * (DD + 4) * sizeof(float);         \
        hipLaunchKernelGGL((nlinear_bwd_kernel<DD>), dim3(grid), dim3(64 * lin_bwd_waves(DD)), lds, st, x, m, W, (int)ldw, b, (int)act, dy, lddy, \
                           dx, (int)accumulate, partial, stride);                                                       \
    }
    NARROW_DISPATCH(d, CALL)
#undef CALL
    PAMNET_LAUNCH_CHECK();
    const int total = (int)(d * d + (db ? d : 0));
    hipLaunchKernelGGL(narrow_reduce_kernel, dim3((total + 63) / 64), dim3(64, 8), 0, st, partial, grid, stride, 1, (int)d,
                       (int)d, (int)d, db ? (int)d : 0, dW, db);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_narrow_heads_fwd_f32(const float* o, int64_t m, int64_t d, const float* w_out, const float* b_out,
                                           const float* w_att, float* out, float* att, pamnet_stream_t stream) {
    if (m < 0 || !width_ok(d)) return PAMNET_EINVAL;
    if (m == 0) return PAMNET_OK;
    if (!o || !w_out || !b_out || !w_att || !out || !att) return PAMNET_ENULL;
    hipStream_t st = as_stream(stream);
    const int rpb = 256 / (int)(d / 4);
    const int64_t want = (m + rpb - 1) / rpb;
    const int grid = (int)(want < 1024 ? want : 1024);
#define CALL(DD) hipLaunchKernelGGL((nheads_fwd_kernel<DD>), dim3(grid), dim3(256), 0, st, o, m, w_out, b_out, w_att, out, att);
    NARROW_DISPATCH(d, CALL)
#undef CALL
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

/* dvec [2 d + 1] = [d w_out | d w_att | d b_out]; partial: pamnet_narrow_blocks x (2 d + 1) floats */
extern "C" int pamnet_narrow_heads_bwd_f32(const float* o, int64_t m, int64_t d, const float* w_out, const float* w_att,
                                           const float* g_out, const float* g_att, float* d_o, float* partial,
                                           float* dvec, pamnet_stream_t stream) {
    if (m <= 0 || !width_ok(d)) return PAMNET_EINVAL;
    if (!o || !w_out || !w_att || !g_out || !g_att || !d_o || !partial || !dvec) return PAMNET_ENULL;
    hipStream_t st = as_stream(stream);
    const int rpb = 256 / (int)(d / 4);
    const int64_t want = (m + rpb - 1) / rpb;
    const int grid = (int)(want < 256 ? want : 256);
#define CALL(DD) hipLaunchKernelGGL((nheads_bwd_kernel<DD>), dim3(grid), dim3(256), 0, st, o, m, w_out, w_att, g_out, g_att, d_o, partial);
    NARROW_DISPATCH(d, CALL)
#undef CALL
    PAMNET_LAUNCH_CHECK();
    const int total = (int)(2 * d + 1);
    hipLaunchKernelGGL(narrow_reduce_kernel, dim3((total + 63) / 64), dim3(64, 8), 0, st, partial, grid, total, 0, (int)d,
                       (int)d, (int)d, total, (float*)nullptr, dvec);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_narrow_local_gate_fwd_f32(const float* P, const float* Q, const int32_t* tgt, const int32_t* src,
                                                const float* b_ji, const float* b_kj, int64_t m, int64_t d, float* m_ji,
                                                float* m_nb, pamnet_stream_t stream) {
    if (m < 0 || !width_ok(d)) return PAMNET_EINVAL;
    if (m == 0) return PAMNET_OK;
    if (!P || !Q || !tgt || !src || !b_ji || !b_kj || !m_ji || !m_nb) return PAMNET_ENULL;
    hipStream_t st = as_stream(stream);
    const int64_t want = (m * (d / 4) + 255) / 256;
    const int grid = (int)(want < 4096 ? want : 4096);
#define CALL(DD) hipLaunchKernelGGL((nlocal_gate_fwd_kernel<DD>), dim3(grid), dim3(256), 0, st, P, Q, tgt, src, b_ji, b_kj, m, m_ji, m_nb);
    NARROW_DISPATCH(d, CALL)
#undef CALL
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_narrow_local_gate_bwd_f32(const float* P, const float* Q, const int32_t* tgt, const int32_t* src,
                                                const float* b_ji, const float* b_kj, int64_t m, int64_t d,
                                                const float* g_ji, const float* g_nb, float* dz, float* dQ,
                                                pamnet_stream_t stream) {
    if (m < 0 || !width_ok(d)) return PAMNET_EINVAL;
    if (m == 0) return PAMNET_OK;
    if (!P || !Q || !tgt || !src || !b_ji || !b_kj || !g_ji || !g_nb || !dz || !dQ) return PAMNET_ENULL;
    hipStream_t st = as_stream(stream);
    const int64_t want = (m * (d / 4) + 255) / 256;
    const int grid = (int)(want < 4096 ? want : 4096);
#define CALL(DD) hipLaunchKernelGGL((nlocal_gate_bwd_kernel<DD>), dim3(grid), dim3(256), 0, st, P, Q, tgt, src, b_ji, b_kj, m, g_ji, g_nb, dz, dQ, 1);
    NARROW_DISPATCH(d, CALL)
#undef CALL
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_narrow_embed_fwd_f32(const float* F, int64_t m, int64_t k, int64_t d, const int32_t* kind,
                                           const float* Wa, const float* ba, const float* Wb, const float* bb, float* y,
                                           pamnet_stream_t stream) {
    if (m < 0 || !width_ok(d) || (k != 16 && k != 42)) return PAMNET_EINVAL;
    if (m == 0) return PAMNET_OK;
    if (!F || !Wa || !ba || !y || (kind && (!Wb || !bb))) return PAMNET_ENULL;
    hipStream_t st = as_stream(stream);
    const int grid = grid_for(m, fwd_per_cu(d));
    const bool two = kind != nullptr;
#define CALL(DD)                                                                                                         \
    {                                                                                                                    \
        if (k == 16) {                                                                                                   \
            const size_t lds = (two ? 2 : 1) * (size_t)DD * 16 * sizeof(float);                                          \
            if (two)                                                                                                     \
                hipLaunchKernelGGL((nembed_fwd_kernel<DD, 16, true>), dim3(grid), dim3(NWG), lds, st, F, m, kind, Wa, ba, Wb, \
                                   bb, y, (const float*)nullptr, 0.f);                                                                               \
            else                                                                                                         \
                hipLaunchKernelGGL((nembed_fwd_kernel<DD, 16, false>), dim3(grid), dim3(NWG), lds, st, F, m, kind, Wa, ba,  \
                                   Wb, bb, y, (const float*)nullptr, 0.f);                                                                           \
        } else {                                                                                                         \
            const size_t lds = (two ? 2 : 1) * (size_t)DD * 48 * sizeof(float);                                          \
            if (two)                                                                                                     \
                hipLaunchKernelGGL((nembed_fwd_kernel<DD, 42, true>), dim3(grid), dim3(NWG), lds, st, F, m, kind, Wa, ba, Wb, \
                                   bb, y, (const float*)nullptr, 0.f);                                                                               \
            else                                                                                                         \
                hipLaunchKernelGGL((nembed_fwd_kernel<DD, 42, false>), dim3(grid), dim3(NWG), lds, st, F, m, kind, Wa, ba,  \
                                   Wb, bb, y, (const float*)nullptr, 0.f);                                                                           \
        }                                                                                                                \
    }
    NARROW_DISPATCH(d, CALL)
#undef CALL
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

/* The 16-wide edge embedding (models.py:185-186) on Bessel rows formed inside the kernel (forward only: inference):
 * dist [m], freq [16], cutoff as for pamnet_rbf_fwd_f32; the same floats as pamnet_rbf_fwd_f32 + pamnet_narrow_embed_fwd_f32. */
extern "C" int pamnet_narrow_embed_rbf_fwd_f32(const float* dist, const float* freq, float cutoff, int64_t m, int64_t d,
                                               const float* Wa, const float* ba, float* y, pamnet_stream_t stream) {
    if (m < 0 || !width_ok(d) || !(cutoff > 0.f)) return PAMNET_EINVAL;
    if (m == 0) return PAMNET_OK;
    if (!dist || !freq || !Wa || !ba || !y) return PAMNET_ENULL;
    hipStream_t st = as_stream(stream);
    const int grid = grid_for(m, fwd_per_cu(d));
#define CALL(DD)                                                                                                         \
    {                                                                                                                    \
        const size_t lds = (size_t)DD * 16 * sizeof(float);                                                              \
        hipLaunchKernelGGL((nembed_fwd_kernel<DD, 16, false, true>), dim3(grid), dim3(NWG), lds, st, dist, m,             \
                           (const int32_t*)nullptr, Wa, ba, (const float*)nullptr, (const float*)nullptr, y, freq,       \
                           1.0f / cutoff);                                                                               \
    }
    NARROW_DISPATCH(d, CALL)
#undef CALL
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

/* dW: [sets, d, k] (set 0 = Wa, set 1 = Wb when `kind` is given), db: [sets, d]; df [m, 16] optional (k = 16, one set) */
extern "C" int pamnet_narrow_embed_bwd_f32(const float* F, int64_t m, int64_t k, int64_t d, const int32_t* kind,
                                           const float* Wa, const float* ba, const float* Wb, const float* bb,
                                           const float* dy, float* df, float* partial, float* dW, float* db,
                                           pamnet_stream_t stream) {
    if (m <= 0 || !width_ok(d) || (k != 16 && k != 42)) return PAMNET_EINVAL;
    if (!F || !Wa || !ba || !dy || !partial || !dW || !db || (kind && (!Wb || !bb))) return PAMNET_ENULL;
    if (df && (k != 16 || kind)) return PAMNET_EINVAL;
    hipStream_t st = as_stream(stream);
    const int grid = grid_for(m, 1, bwd_waves((int)d));
    const bool two = kind != nullptr;
    const int kp = (k == 16) ? 16 : 48;
    const int sets = two ? 2 : 1;
    const int stride = (int)(sets * (d * kp + d));
#define CALL(DD)                                                                                                          \
    {                                                                                                                     \
        const size_t scratch = bwd_waves(DD) * 16 * ((kp > DD ? kp : DD) + 4) * sizeof(float);                                        \
        if (k == 16 && df) {                                                                                              \
            const size_t lds = 2 * (size_t)DD * 16 * sizeof(float) + scratch;                                             \
            hipLaunchKernelGGL((nembed_bwd_kernel<DD, 16, false, true>), dim3(grid), dim3(64 * bwd_waves(DD)), lds, st, F, m, kind, Wa, ba, \
                               Wb, bb, dy, df, partial, stride, (const float*)nullptr, 0.f);                                                          \
        } else if (k == 16 && !two) {                                                                                     \
            const size_t lds = (size_t)DD * 16 * sizeof(float) + scratch;                                                 \
            hipLaunchKernelGGL((nembed_bwd_kernel<DD, 16, false, false>), dim3(grid), dim3(64 * bwd_waves(DD)), lds, st, F, m, kind, Wa,    \
                               ba, Wb, bb, dy, df, partial, stride, (const float*)nullptr, 0.f);                                                      \
        } else if (k == 16) {                                                                                             \
            const size_t lds = 2 * (size_t)DD * 16 * sizeof(float) + scratch;                                             \
            hipLaunchKernelGGL((nembed_bwd_kernel<DD, 16, true, false>), dim3(grid), dim3(64 * bwd_waves(DD)), lds, st, F, m, kind, Wa, ba, \
                               Wb, bb, dy, df, partial, stride, (const float*)nullptr, 0.f);                                                          \
        } else if (!two) {                                                                                                \
            const size_t lds = (size_t)DD * 48 * sizeof(float) + scratch;                                                 \
            hipLaunchKernelGGL((nembed_bwd_kernel<DD, 42, false, false>), dim3(grid), dim3(64 * bwd_waves(DD)), lds, st, F, m, kind, Wa,    \
                               ba, Wb, bb, dy, df, partial, stride, (const float*)nullptr, 0.f);                                                      \
        } else {                                                                                                          \
            const size_t lds = 2 * (size_t)DD * 48 * sizeof(float) + scratch;                                             \
            hipLaunchKernelGGL((nembed_bwd_kernel<DD, 42, true, false>), dim3(grid), dim3(64 * bwd_waves(DD)), lds, st, F, m, kind, Wa, ba, \
                               Wb, bb, dy, df, partial, stride, (const float*)nullptr, 0.f);                                                          \
        }                                                                                                                 \
    }
    NARROW_DISPATCH(d, CALL)
#undef CALL
    PAMNET_LAUNCH_CHECK();
    const int total = (int)(sets * (d * kp + d));
    hipLaunchKernelGGL(narrow_reduce_kernel, dim3((total + 63) / 64), dim3(64, 8), 0, st, partial, grid, stride, sets,
                       (int)d, kp, (int)k, (int)(sets * d), dW, db);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

/* Backward of pamnet_narrow_embed_rbf_fwd_f32: dW [d, 16], db_dfreq [d + 16] = the bias gradient followed by the gradient of
 * the 16 Bessel frequencies (layers/basic.py:65-76); partial: blocks x (d * 16 + d + 16) floats.  The [m, 16] rows and their
 * gradient never exist. */
extern "C" int pamnet_narrow_embed_rbf_bwd_f32(const float* dist, const float* freq, float cutoff, int64_t m, int64_t d,
                                               const float* Wa, const float* ba, const float* dy, float* partial, float* dW,
                                               float* db_dfreq, pamnet_stream_t stream) {
    if (m <= 0 || !width_ok(d) || !(cutoff > 0.f)) return PAMNET_EINVAL;
    if (!dist || !freq || !Wa || !ba || !dy || !partial || !dW || !db_dfreq) return PAMNET_ENULL;
    hipStream_t st = as_stream(stream);
    const int grid = grid_for(m, 1, bwd_waves((int)d));
    const int stride = (int)(d * 16 + d + 16);
#define CALL(DD)                                                                                                            \
    {                                                                                                                       \
        const size_t scratch = bwd_waves(DD) * 16 * (DD + 4) * sizeof(float);                                               \
        const size_t lds = 2 * (size_t)DD * 16 * sizeof(float) + scratch;                                                   \
        hipLaunchKernelGGL((nembed_bwd_kernel<DD, 16, false, true, true>), dim3(grid), dim3(64 * bwd_waves(DD)), lds, st,    \
                           dist, m, (const int32_t*)nullptr, Wa, ba, (const float*)nullptr, (const float*)nullptr, dy,      \
                           (float*)nullptr, partial, stride, freq, 1.0f / cutoff);                                          \
    }
    NARROW_DISPATCH(d, CALL)
#undef CALL
    PAMNET_LAUNCH_CHECK();
    const int total = (int)(d * 16 + d + 16);
    hipLaunchKernelGGL(narrow_reduce_kernel, dim3((total + 63) / 64), dim3(64, 8), 0, st, partial, grid, stride, 1, (int)d, 16,
                       16, (int)(d + 16), dW, db_dfreq);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}
