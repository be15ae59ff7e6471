// Layer-stack engine: the whole n_layer x (global, local) message-passing loop of PAMNet.forward (models.py:196-204)
// -- forward or backward -- enqueued by ONE C call.
//
// The reference drives this loop from Python, one small op at a time (~150 launches per layer pair); even with fused
// kernels a Python-level loop leaves the MI355X idle behind the interpreter.  Here the host side is a straight C++
// sequence of ~7 (forward) / ~16 (backward) kernel launches per layer pair on the caller's stream, working out of two
// caller-owned arenas:
//   saved : activations the backward needs (pre-activations, gates, residual taps), one slab per layer
//   temp  : scratch reused by every layer (projections, messages, gradient staging, split-K partials)
// Nothing is allocated, no state is kept, nothing synchronises: the caller sizes the arenas with
// pamnet_stack_workspace and keeps them alive until the backward has been enqueued.
// The shared edge embeddings (e_g, rbf_e, e_sbf feed all layers) get their gradients accumulated in place by the
// backward kernels themselves (accumulate flag), in a fixed layer order -> deterministic.

#include <stdlib.h>

#include <vector>

#include "common.h"

namespace {

constexpr int64_t D = 128;
constexpr int NG = 28;      // pointers per global layer: Wx1 bx1 Wm bm Wea | tail W[10] b[10] w_out b_out w_att
constexpr int NL = 35;      // pointers per local layer : Wx1 bx1 Wji bji Wkj bkj Ws1 bs1 Ws2 bs2 Wlr Wlo | tail ...
constexpr int GT = 5, LT = 12;   // offset of the tail block

struct Graph {
    int64_t n, eg, el, tp;
    const int32_t *g_ptr, *g_row, *g_col, *gT_ptr, *gT_perm;
    const int32_t *l_ptr, *l_row, *l_col, *lT_ptr, *lT_perm;
    const int32_t *t_ptr, *t_row, *t_col, *tT_ptr, *tT_perm;
    const int32_t* cuts;      // nullable: the fused global-edge kernels' work split, made with the graph (pamnet_seg_cuts_i32)
    const int32_t *tT_edge, *tT_node;   // nullable pair: pamnet_triplet_transpose_aux_i32 (the local aggregation's backward gather)
};

inline int64_t al(int64_t x) { return (x + 63) / 64 * 64; }       // 256-byte aligned slabs

// hdz / gh / hp: backward scratch of the layer's head branch (dz7..dz9, its d x_out contribution, head-vector partials),
// filled for all layers by one launch at the start of the backward
struct GlobalSaved { float *Zx1, *z, *ea, *x2, *Z, *R, *xout, *hdz, *gh, *hp, *Pg; };
struct LocalSaved { float *Zx1, *zji, *zkj, *q2, *q3, *mnb, *mt, *s, *z1, *z2, *x2, *Z, *R, *xout, *hdz, *gh, *hp; };
inline int64_t head_partial_floats(const Graph& g) { return al(((g.n + 15) / 16) * 257); }

// Round 6 A/B ("backward by recompute"): PAMNET_EDGE_RECOMPUTE=1 -- the training forward of the fused
// global-edge step saves nothing of edge size (it runs the inference form: no z, no ea), only the two node planes P_i, P_j it
// gathered from; the backward re-runs the forward kernel for z and ea into scratch right ahead of the fused backward kernel.
// Same numbers bit for bit (the same kernel computes them), 2 x E_g x 512 bytes less per layer in the saved arena; whether it
// is faster is what profiles/r06_edge_recompute_ab.txt records (it is not).  Read once.  Only where the backward takes its
// fused weight-gradient form (the scratch it frees holds the recomputed rows).
inline bool edge_recompute_on() {
    static const bool v = [] { const char* e = getenv("PAMNET_EDGE_RECOMPUTE"); return e && atoi(e) != 0; }();
    return v;
}
inline bool edge_wgrad(const Graph& g);
inline bool edge_recompute(const Graph& g) { return edge_recompute_on() && edge_wgrad(g); }
inline int64_t global_saved_floats(const Graph& g) {
    const int64_t edge = edge_recompute(g) ? al(g.n * D) * 2 : al(g.eg * D) * 2;
    return al(g.n * D) * 19 + edge + head_partial_floats(g);
}
inline int64_t local_saved_floats(const Graph& g) {
    return al(g.n * D) * 19 + al(g.el * D) * 6 + al(g.tp * D) * 3 + head_partial_floats(g);
}

inline GlobalSaved carve_global(float* p, const Graph& g) {
    GlobalSaved s;
    const int64_t nd = al(g.n * D), ed = al(g.eg * D);
    s.Zx1 = p; p += nd;
    if (edge_recompute(g)) {
        s.z = s.ea = nullptr;
        s.Pg = p; p += 2 * nd;   // (planes n*D apart)
    } else {
        s.Pg = nullptr;
        s.z = p; p += ed;
        s.ea = p; p += ed;
    }
    s.x2 = p; p += nd;
    s.Z = p; p += 10 * nd;       // NB: planes are n*D apart (not padded): 10*nd >= 10*n*D
    s.R = p; p += 2 * nd;
    s.xout = p; p += nd;
    s.hdz = p; p += 3 * nd;
    s.gh = p; p += nd;
    s.hp = p;
    return s;
}

inline LocalSaved carve_local(float* p, const Graph& g) {
    LocalSaved s;
    const int64_t nd = al(g.n * D), ed = al(g.el * D), td = al(g.tp * D);
    s.Zx1 = p; p += nd;
    s.zji = p; p += ed;
    s.zkj = p; p += ed;
    s.q2 = p; p += ed;
    s.q3 = p; p += ed;
    s.mnb = p; p += ed;
    s.mt = p; p += ed;
    s.s = p; p += td;
    s.z1 = p; p += td;
    s.z2 = p; p += td;
    s.x2 = p; p += nd;
    s.Z = p; p += 10 * nd;
    s.R = p; p += 2 * nd;
    s.xout = p; p += nd;
    s.hdz = p; p += 3 * nd;
    s.gh = p; p += nd;
    s.hp = p;
    return s;
}

// temp arena carving (forward and backward share it; the backward needs more)
struct Temp {
    float *x1, *P, *msg, *mji;                                         // forward
    float *dZ, *dZ2, *dx2, *dresx, *head, *dz, *dea, *dP, *dZx1, *dxa, *dxb;    // backward (global + shared)
    float *dPg, *dZx1g;          // the global layer's own d P (2 planes) / d Zx1: the local layer's stay alive until the pair's
                                 // merged weight-gradient launch
    float *dzji, *dzkj, *dq2, *dmt, *dq3, *dmnb, *ds, *dz1, *dz2;      // backward (local)
    float *partial, *partial2;   // split-K scratch of two consecutive weight-gradient batches (the reduction of one runs
                                 // inside the launch of the next)
    float* rider_partial;        // split-K scratch of the rider batch (10 node-level jobs in a node-chain launch)
    float* dump;                 // PAMNET_EDGE_RECOMPUTE: node output of the recompute launch (discarded)
    float* edge_partial;         // partial tiles of the fused global-edge backward's own weight gradients (2 x <= 256 slots)
    int32_t* cuts;               // node-aligned work split of the fused global-edge kernels (<= 257 ints)
};

constexpr int WJOBS = 24;

// split-K scratch of the largest weight-gradient batch of a layer (slots of 128*128 + 256 floats, <= 256 rows each)
inline int64_t wgrad_floats(const Graph& g) {
    auto slots = [](int64_t rows) { int64_t s = (rows + 127) / 128; return s < 1 ? 1 : (s > 256 ? 256 : s); };   // 128: the smallest chunk a plan may use (wgrad.hip)
    const int64_t glob = 15 * slots(g.n) + 2 * slots(g.eg);
    const int64_t loc = 15 * slots(g.n) + 4 * slots(g.el) + 2 * slots(g.tp);
    return (glob + loc) * (D * D + 2 * D);       // (a layer pair's merged batch holds both layers' own jobs)
}

// the 10 tail jobs of a chain riding in the next chain launch (256-row slots)
inline int64_t rider_floats(const Graph& g) {
    int64_t s = (g.n + 255) / 256;
    s = s < 1 ? 1 : (s > 256 ? 256 : s);
    return 2 * 10 * s * (D * D + 2 * D);         // two chains' riders wait for the pair's merged launch
}

// Round 5: the weight gradients of the global edge step (dW_e = dz^T e_g, dW_ea = dea^T e_g) are formed inside the fused
// backward edge kernel (edge_agg.hip global_edge_agg_bwd_wg_kernel) instead of as two E_g-row jobs of the split-K launches.
// PAMNET_EDGE_WGRAD=0 / 1 forces the old / new route (read once); default: the new one where a workgroup has enough rows to
// amortise its two 64 KB partial tiles.
inline bool edge_wgrad(const Graph& g) {
    static const int v = [] { const char* e = getenv("PAMNET_EDGE_WGRAD"); return e ? atoi(e) : -1; }();
    const bool fits = g.eg > 0 && g.eg < (int64_t(1) << 23) && g.n < (int64_t(1) << 23);   // (32-bit byte offsets in that kernel)
    if (v >= 0) return v != 0 && fits;
    return fits && g.eg >= 256 * 512;
}
inline int64_t edge_partial_floats(const Graph& g) {
    if (!edge_wgrad(g)) return 0;                 // (the plain backward leaves no partial tiles: nothing to reserve)
    int64_t f = 0;
    pamnet_global_edge_agg_wg_floats(g.eg, &f, nullptr);
    return f;
}

inline int64_t temp_floats(const Graph& g) {
    const int64_t nd = al(g.n * D), gd = al(g.eg * D), ld = al(g.el * D), td = al(g.tp * D);
    int64_t t = 0;
    t += nd + 4 * nd + gd + ld;                                        // x1 P msg mji
    t += 10 * nd + 7 * nd + nd + nd + al(((g.n + 15) / 16) * 257) + gd + gd + 4 * nd + nd + nd + nd;
    t += 6 * ld + 3 * td;
    t += 3 * nd;                                                       // dPg (2 planes), dZx1g
    t += 2 * wgrad_floats(g) + rider_floats(g) + edge_partial_floats(g) + 320;
    if (edge_recompute(g)) t += nd;                                    // the recompute launch's node output (discarded)
    return t;
}

inline Temp carve_temp(float* p, const Graph& g) {
    Temp t;
    const int64_t nd = al(g.n * D), gd = al(g.eg * D), ld = al(g.el * D), td = al(g.tp * D);
    t.x1 = p; p += nd;
    t.P = p; p += 4 * nd;
    t.msg = p; p += gd;
    t.mji = p; p += ld;
    t.dZ = p; p += 10 * nd;
    t.dZ2 = p; p += 7 * nd;      // second chain-gradient buffer: a fused head+chain launch writes the next chain's dZ
                                 // while the previous chain's is still waiting for its weight-gradient launch
    t.dx2 = p; p += nd;
    t.dresx = p; p += nd;
    t.head = p; p += al(((g.n + 15) / 16) * 257);
    t.dz = p; p += gd;
    t.dea = p; p += gd;
    t.dP = p; p += 4 * nd;
    t.dZx1 = p; p += nd;
    t.dPg = p; p += 2 * nd;
    t.dZx1g = p; p += nd;
    t.dxa = p; p += nd;
    t.dxb = p; p += nd;
    t.dzji = p; p += ld;
    t.dzkj = p; p += ld;
    t.dq2 = p; p += ld;
    t.dmt = p; p += ld;
    t.dq3 = p; p += ld;
    t.dmnb = p; p += ld;
    t.ds = p; p += td;
    t.dz1 = p; p += td;
    t.dz2 = p; p += td;
    t.partial = p; p += wgrad_floats(g);
    t.partial2 = p; p += wgrad_floats(g);
    t.rider_partial = p; p += rider_floats(g);
    t.edge_partial = p; p += edge_partial_floats(g);
    t.dump = nullptr;
    if (edge_recompute(g)) { t.dump = p; p += nd; }
    t.cuts = reinterpret_cast<int32_t*>(p);
    return t;
}

#define CK(call)                 \
    do {                         \
        int rc__ = (call);       \
        if (rc__) return rc__;   \
    } while (0)

#define HK(call)                                   \
    do {                                           \
        const hipError_t e__ = (call);             \
        if (e__ != hipSuccess) return (int)e__;    \
    } while (0)

// fragment-ordered weight images for the node chains (kind 0: fp32 fragments, or bf16x3 images with `pieces`) and, round 6, for
// the edge-level kernels (kind 1: the bf16x3 fragments a wave would split out of its slice itself, edge_core.h load_wfragb1; in
// their own region `ebase` of the arena) -- all of a direction's images in one launch (pamnet_pack_weights_mixed_f32, <= 224
// matrices per launch)
constexpr int64_t EDGE_IMG = 3 * D * D / 2;
struct PackList {
    static constexpr int CAP = 224;
    const float* src[CAP];
    int64_t ld[CAP], off[CAP];
    int32_t kind[CAP];
    int n = 0;
    float* base;
    float* ebase = nullptr;
    int64_t done = 0, edone = 0;   // chain / edge images handed out so far
    int32_t transposed;
    pamnet_stream_t st;
    int rc = 0;
    int64_t img = D * D;       // floats per chain image: fp32 fragment images, or bf16x3 images (3 pieces x 2 bytes: 1.5 x)
    bool bf16x3 = false;
    PackList(float* b, int32_t t, pamnet_stream_t s, bool pieces = false, float* eb = nullptr)
        : base(b), ebase(eb), transposed(t), st(s), img(pieces ? 3 * D * D / 2 : D * D), bf16x3(pieces) {}
    // returns the image the matrix will occupy
    const float* add(const float* W, int64_t ldw, bool fp32 = false) {      // fp32: an fp32 fragment image whatever the list's kind
        if (n == CAP) flush();
        src[n] = W, ld[n] = ldw, kind[n] = (bf16x3 && !fp32) ? 1 : 0, off[n] = done * img;
        ++n;
        return base + done++ * img;
    }
    const float* add_edge(const float* W, int64_t ldw) {
        if (n == CAP) flush();
        src[n] = W, ld[n] = ldw, kind[n] = 1, off[n] = (ebase - base) + edone * EDGE_IMG;
        ++n;
        return ebase + edone++ * EDGE_IMG;
    }
    void flush() {
        if (n && !rc) rc = pamnet_pack_weights_mixed_f32(n, src, ld, kind, off, transposed, base, st);
        n = 0;
    }
};
// edge-level images per layer pair: forward W_e, W_ea + the local edge step's four slices; backward W_e, W_ea, mlp_sbf's two and
// the local edge step's four (transposed)
constexpr int64_t EDGE_PACK_PER_PAIR = 8;
// PAMNET_EDGE_IMAGES=0: the edge-level kernels split their fp32 slices themselves (the form before round 6).  The images need the
// 8-wave geometry of the local edge kernel (PAMNET_EDGE_WAVES=4 forces the other one: no images then).
inline bool edge_images() {
    static const bool v = [] {
        const char* e = getenv("PAMNET_EDGE_IMAGES");
        const char* w = getenv("PAMNET_EDGE_WAVES");
        return (!e || atoi(e) != 0) && !(w && atoi(w) == 4);
    }();
    return v;
}
constexpr int64_t PACK_PER_PAIR = 28;       // forward: 10 + 5 (global chain + local head) + 10 + 3 (local chain + next global head)
constexpr int64_t PACK_FLOATS_PER_PAIR = PACK_PER_PAIR * (3 * D * D / 2);     // sized for bf16x3 images throughout
// The forward chains of single-round batches (<= 256 row tiles: QM9, RNA) on the bf16 matrix pipe at fp32 accuracy
// (node_tail_fwd_bf16_kernel, bf16x3 weight images): 768 instead of 2 048 matrix-pipe cycles per layer.  Built in round 5 with
// the accumulator four rows of one channel per lane and 27.8 us per launch against the fp32-MFMA form's 26.5 (the epilogue, 12
// four-byte piece stores per lane, grew by what the MFMAs shrank); with the operands swapped (round 6: a lane's four consecutive
// channels leave as three 8-byte piece stores) a 7-layer chain is 23 500 cycles against 27 500 and the QM9 step 1.954 against
// 1.989 ms same box (profiles/r06_chain_bf16.txt); same outputs to 3e-7 (tests/test_hip_fused.py::test_node_tail_fwd_bf16x6).
// PAMNET_CHAIN_BF16=0: the fp32-MFMA chains.  Larger batches run the lean fp32 chain kernels (several workgroups per CU).
// Round 6: the segment sums that feed a fused head + chain backward launch (the source-side sum of the global layer's d z, the
// four sums of the local layer) are formed by that launch's own row tiles (node_tail.hip gather_begin / gather_finish) instead of by launches
// of their own ahead of it: two launches fewer per layer pair on the dependent chain.  PAMNET_FUSE_SEGSUM=0: the separate launches.
inline bool fuse_segsum(const Graph& g) {
    static const bool v = [] { const char* e = getenv("PAMNET_FUSE_SEGSUM"); return !e || atoi(e) != 0; }();
    // (batches of more than 256 row tiles run the LEAN chain kernels -- node_tail.hip LEAN_FROM_TILES --, which read their planes:
    // those keep the tuned stand-alone segment sums)
    return v && (g.n + 15) / 16 <= 256;
}
// Round 6: the local layer's two chained aggregations (pamnet_local_agg_fwd_f32) are formed by the row tiles of the chain launch
// that consumes them (node_tail.hip local_agg_rows; pamnet_node_tail_fwd_agg_f32): one launch fewer per layer on the dependent
// chain.  PAMNET_FUSE_LOCAL_AGG=0: the separate launch.  Small batches only (the lean chain forms read their input).
inline bool fuse_local_agg(const Graph& g) {
    static const bool v = [] { const char* e = getenv("PAMNET_FUSE_LOCAL_AGG"); return !e || atoi(e) != 0; }();
    return v && (g.n + 15) / 16 <= 256;
}
// (row tiles up to which the bf16x6 chains run: they park a tile's state in LDS, one workgroup per CU -- single-round batches;
// PAMNET_CHAIN_BF16_TILES overrides, for measurements)
inline int64_t chain_bf16_tiles() {
    static const int64_t v = [] { const char* e = getenv("PAMNET_CHAIN_BF16_TILES"); return e ? (int64_t)atoll(e) : (int64_t)256; }();
    return v;
}
inline bool chain_bf16() {
    static bool v = [] { const char* e = getenv("PAMNET_CHAIN_BF16"); return !e || atoi(e) != 0; }();
    return v;
}

// weight-gradient job list builder
struct Jobs {
    const float* dZ[WJOBS];
    const float* A[WJOBS];
    float* dW[WJOBS];
    float* db[WJOBS];
    int64_t ld_dz[WJOBS], ld_a[WJOBS], ld_dw[WJOBS], rows[WJOBS];
    int32_t mode[WJOBS];
    int n = 0;
    void add(const float* dz, const float* a, int mode_, int64_t rows_, float* dw, int64_t ld_dw_, float* db_) {
        dZ[n] = dz; A[n] = a; dW[n] = dw; db[n] = db_;
        ld_dz[n] = D; ld_a[n] = D; ld_dw[n] = ld_dw_; rows[n] = rows_; mode[n] = mode_;
        ++n;
    }
};

// Riders: with packed weights the 10 tail jobs of a chain do not go into its layer's weight-gradient launch but ride as
// extra workgroups of the NEXT node-chain launch (which leaves 256 - ceil(n/16) CUs idle); the layer's own launch keeps
// the jobs whose operands that next launch overwrites (dZx1, dP) and the edge-level ones.
constexpr int RIDER_MAX_SLOTS = 256;
inline int plan_rider(Jobs& j, float* partial, const Graph& g, void* rider, int64_t* slots) {
    const int64_t room = RIDER_MAX_SLOTS - (g.n + 15) / 16;   // idle CUs of the chain launch (riders_fit: >= 80)
    return pamnet_wgrad_rider_plan_f32(j.n, j.dZ, j.ld_dz, j.A, j.ld_a, j.mode, j.rows, j.dW, j.ld_dw, j.db, partial, room,
                                       rider, slots);
}
// riders pay when the chain leaves enough CUs idle for slots of a few hundred rows: ceil(n/16) <= 176 workgroups
// how many of a chain's 10 tail jobs ride (the rest stay in the layer's own weight-gradient launch)
constexpr int rider_jobs() { return 10; }      // all ten (fewer riders, the rest in the layer's own launch, measured slower)
inline bool riders_fit(const Graph& g) { return (g.n + 15) / 16 <= RIDER_MAX_SLOTS - 80; }

inline void tail_jobs(Jobs& j, const Graph& g, const float* dZ, const float* hdz, const float* x2, const float* Z,
                      const float* R, const float* xout, float* const* gt /* tail block of the gradient table */,
                      int k0 = 0, int k1 = 10) {
    const int64_t pl = g.n * D;
    const float* src[10] = {x2, Z, Z + pl, R, Z + 3 * pl, R + pl, Z + 5 * pl, xout, Z + 7 * pl, Z + 8 * pl};
    const int mode[10] = {0, 1, 1, 0, 1, 0, 1, 0, 1, 1};
    for (int k = k0; k < k1; ++k)         // dz7..dz9 come from the batched head-branch backward
        j.add(k < 7 ? dZ + k * pl : hdz + (k - 7) * pl, src[k], mode[k], g.n, gt[k], D, gt[10 + k]);
}

// all weight gradients of a layer + the head-vector gradients of its node chain (partials left by node_tail_bwd)
struct HeadGrads {
    const float* partial;       // the chain's head-vector partials (null: none)
    float *d_wout, *d_watt, *d_bout;
};
inline int run_jobs(Jobs& j, float* partial, const Graph& g, const HeadGrads& h, const HeadGrads& h2, void* ctx,
                    pamnet_stream_t st) {
    return pamnet_wgrad_deferred_f32(j.n, j.dZ, j.ld_dz, j.A, j.ld_a, j.mode, j.rows, j.dW, j.ld_dw, j.db, partial, h.partial,
                                     (g.n + 15) / 16, h.d_wout, h.d_watt, h.d_bout, h2.partial, h2.d_wout, h2.d_watt,
                                     h2.d_bout, ctx, st);
}

}  // namespace

// floats of the optional weight-image arena (`wpack`) of pamnet_stack_fwd_f32 / pamnet_stack_bwd_f32
extern "C" int pamnet_stack_pack_floats(int64_t n_layer, int64_t* floats) {
    if (n_layer < 1 || !floats) return PAMNET_EINVAL;
    *floats = n_layer * (PACK_FLOATS_PER_PAIR + EDGE_PACK_PER_PAIR * EDGE_IMG);
    return PAMNET_OK;
}

extern "C" int pamnet_stack_workspace(int64_t n, int64_t eg, int64_t el, int64_t tp, int64_t n_layer,
                                      int64_t* saved_floats, int64_t* temp_floats_out) {
    if (n < 0 || eg < 0 || el < 0 || tp < 0 || n_layer < 1 || !saved_floats || !temp_floats_out) return PAMNET_EINVAL;
    Graph g{};
    g.n = n; g.eg = eg; g.el = el; g.tp = tp;
    *saved_floats = n_layer * (al(global_saved_floats(g)) + al(local_saved_floats(g)));
    *temp_floats_out = temp_floats(g);
    return PAMNET_OK;
}

// layout[0] = floats per layer pair in `saved`; layout[1] / layout[2] = offset of the global / local layer's output
// node features x_out inside a pair's slab (for inspection: x after every layer).
extern "C" int pamnet_stack_layout(int64_t n, int64_t eg, int64_t el, int64_t tp, int64_t* layout) {
    if (n < 0 || eg < 0 || el < 0 || tp < 0 || !layout) return PAMNET_EINVAL;
    Graph g{};
    g.n = n; g.eg = eg; g.el = el; g.tp = tp;
    const int64_t gs = al(global_saved_floats(g)), ls = al(local_saved_floats(g));
    float* base = nullptr;
    layout[0] = gs + ls;
    layout[1] = carve_global(base, g).xout - base;
    layout[2] = gs + (carve_local(base, g).xout - base);
    return PAMNET_OK;
}

// graph_desc: 4 x int64 sizes {n, eg, el, tp}; graph_idx: 18 device pointers in the order of struct Graph (the last three nullable).
static int fill_graph(Graph& g, const int64_t* sizes, const int32_t* const* idx) {
    if (!sizes || !idx) return PAMNET_ENULL;
    g.n = sizes[0]; g.eg = sizes[1]; g.el = sizes[2]; g.tp = sizes[3];
    const int32_t** f = &g.g_ptr;
    for (int k = 0; k < 18; ++k) f[k] = idx[k];
    // PAMNET_TT_AUX=0: ignore the precomputed gather indices of the local aggregation's backward (A/B timing; read once)
    static const bool tt_aux = [] { const char* e = getenv("PAMNET_TT_AUX"); return !e || atoi(e) != 0; }();
    if (!tt_aux || !g.tT_edge || !g.tT_node) g.tT_edge = g.tT_node = nullptr;
    return PAMNET_OK;
}

extern "C" int pamnet_stack_fwd_f32(const int64_t* sizes, const int32_t* const* graph_idx, int64_t n_layer,
                                    const float* x0, const float* e_g, const float* rbf_e, const float* e_sbf,
                                    const float* const* gparams, const float* const* lparams, float* saved, float* temp,
                                    float* outs, float* atts, int32_t save_for_backward, float* wpack,
                                    pamnet_stream_t aux, void* const* aux_events, pamnet_stream_t st) {
    Graph g;
    CK(fill_graph(g, sizes, graph_idx));
    if (n_layer < 1) return PAMNET_EINVAL;
    if (!x0 || !e_g || !rbf_e || !e_sbf || !gparams || !lparams || !saved || !temp || !outs || !atts) return PAMNET_ENULL;
    const Temp t = carve_temp(temp, g);
    const int64_t gs = al(global_saved_floats(g)), ls = al(local_saved_floats(g));
    const float* x = x0;
    // Inference mode (save_for_backward = 0): tensors only the backward reads (pre-activations, gates, residual taps)
    // are not written at all -- about half of the forward's HBM writes.  `sv(p)` = p or null.
    const bool keep = save_for_backward != 0;
    auto sv = [keep](float* p) -> float* { return keep ? p : nullptr; };
    // The triplet/pair MLP s_k = mlp_sbf_k(e_sbf) does not depend on the node features: with an auxiliary stream all
    // n_layer of them are enqueued up front and run beside the node-level kernels of the first layers, which occupy
    // only ceil(n/16) of the 256 CUs.  Event 0 = inputs ready, event 1+k = s_k ready.
    const bool forked = aux && aux_events && g.tp > 0;
    // Riders (packed weights, chains that leave CUs idle): the triplet/pair MLP of layer k >= 1 rides in the two node-chain
    // launches that precede its use -- first half of its row tiles in the local chain of pair k-1, second half in the
    // global chain of pair k; only layer 0's runs as a launch of its own ahead of the loop.
    const bool ride = !forked && wpack != nullptr && g.tp > 0 && riders_fit(g);
    const bool cb0 = ride && chain_bf16() && (g.n + 15) / 16 <= chain_bf16_tiles();      // (= cb below: the bf16x6 chains run)
    const int64_t mlp_tiles = (g.tp + 15) / 16, mlp_half = mlp_tiles / 2;
    const int64_t rider_wgs = RIDER_MAX_SLOTS - (g.n + 15) / 16;
    // Without the fork: the triplet/pair MLPs of up to 8 layers at a time as one launch ahead of the layer loop.
    if (!forked && g.tp > 0 && ride) {
        // layer 0: the first half of its row tiles as a launch of its own, the second half rides in the first chain launch
        const float* const* lp = lparams;
        const LocalSaved q = carve_local(saved + gs, g);
        // (with the bf16x6 chains' 8-wave riders the first chain launch carries all of layer 0's tiles: +6.6 us there against the
        // 14 us of this launch -- profiles/r06_chain_bf16.txt)
        if (mlp_half > 0 && !cb0)
            CK(pamnet_mlp2_fwd_f32(e_sbf, mlp_half * 16, lp[6], lp[7], lp[8], lp[9], sv(q.z1), sv(q.z2), q.s, st));
    } else if (!forked && g.tp > 0) {
        const int64_t n_up = n_layer;
        for (int64_t k0 = 0; k0 < n_up; k0 += 8) {
            const int64_t nk = n_up - k0 < 8 ? n_up - k0 : 8;
            const float* prm[32];
            float* out[24];
            for (int64_t k = 0; k < nk; ++k) {
                const float* const* lp = lparams + (k0 + k) * NL;
                const LocalSaved q = carve_local(saved + (k0 + k) * (gs + ls) + gs, g);
                prm[4 * k] = lp[6], prm[4 * k + 1] = lp[7], prm[4 * k + 2] = lp[8], prm[4 * k + 3] = lp[9];
                out[3 * k] = sv(q.z1), out[3 * k + 1] = sv(q.z2), out[3 * k + 2] = q.s;
            }
            CK(pamnet_mlp2_fwd_multi_f32(e_sbf, g.tp, nk, prm, out, st));
        }
    }
    if (forked) {
        hipStream_t a = as_stream(aux);
        HK(hipEventRecord(reinterpret_cast<hipEvent_t>(aux_events[0]), as_stream(st)));
        HK(hipStreamWaitEvent(a, reinterpret_cast<hipEvent_t>(aux_events[0]), 0));
        for (int64_t k = 0; k < n_layer; ++k) {
            const float* const* lp = lparams + k * NL;
            const LocalSaved q = carve_local(saved + k * (gs + ls) + gs, g);
            CK(pamnet_mlp2_fwd_f32(e_sbf, g.tp, lp[6], lp[7], lp[8], lp[9], sv(q.z1), sv(q.z2), q.s, aux));
            HK(hipEventRecord(reinterpret_cast<hipEvent_t>(aux_events[1 + k]), a));
        }
    }
    // Fragment-ordered weight images for the node chains (optional `wpack` arena): one pack launch for all layers.
    // img[k] = {global chain W[10], local head Wx1 + 4 blocks, local chain W[10], next global head Wx1 + 2 blocks}
    const bool packed = wpack != nullptr;
    struct PairImg {
        const float *gt[10], *lh[5], *lt[10], *nh[3];
    };
    std::vector<PairImg> img_store(packed ? (size_t)n_layer : 0);
    PairImg* img = packed ? img_store.data() : nullptr;
    // bf16x3 images for everything the chains multiply by (matrices 0..6 + the fused heads of the next layers), fp32
    // images for the mlp_out matrices 7..9 (node_heads_fwd_kernel)
    const bool cb = packed && chain_bf16() && (g.n + 15) / 16 <= chain_bf16_tiles();      // (one workgroup per CU: single-round batches)
    // edge-level fragment images (region behind the chain images): W_e, W_ea of the global step, the local edge step's slices
    const bool eimg = packed && edge_images();
    struct EdgeImg {
        const float *we, *wea, *wq[4];
    };
    std::vector<EdgeImg> eimg_store(eimg ? (size_t)n_layer : 0);
    if (packed) {
        PackList pl(wpack, 0, st, cb, wpack + n_layer * PACK_FLOATS_PER_PAIR);
        for (int64_t k = 0; k < n_layer; ++k) {
            const float* const* gp = gparams + k * NG;
            const float* const* lp = lparams + k * NL;
            for (int i = 0; i < 7; ++i) img[k].gt[i] = pl.add(gp[GT + i], D);
            for (int i = 7; i < 10; ++i) img[k].gt[i] = pl.add(gp[GT + i], D, true);
            img[k].lh[0] = pl.add(lp[0], D);
            img[k].lh[1] = pl.add(lp[2], 3 * D), img[k].lh[2] = pl.add(lp[4], 3 * D);
            img[k].lh[3] = pl.add(lp[2] + D, 3 * D), img[k].lh[4] = pl.add(lp[4] + D, 3 * D);
            for (int i = 0; i < 7; ++i) img[k].lt[i] = pl.add(lp[LT + i], D);
            for (int i = 7; i < 10; ++i) img[k].lt[i] = pl.add(lp[LT + i], D, true);
            if (k + 1 < n_layer) {
                const float* const* gn = gparams + (k + 1) * NG;
                img[k].nh[0] = pl.add(gn[0], D);
                img[k].nh[1] = pl.add(gn[2], 3 * D), img[k].nh[2] = pl.add(gn[2] + D, 3 * D);
            }
            if (eimg) {
                EdgeImg& ei = eimg_store[k];
                ei.we = pl.add_edge(gp[2] + 2 * D, 3 * D), ei.wea = pl.add_edge(gp[4], D);
                ei.wq[0] = pl.add_edge(lp[2] + 2 * D, 3 * D), ei.wq[1] = pl.add_edge(lp[4] + 2 * D, 3 * D);
                ei.wq[2] = pl.add_edge(lp[10], D), ei.wq[3] = pl.add_edge(lp[11], D);
            }
        }
        pl.flush();
        CK(pl.rc);
    }
    const int32_t pkc = cb ? 2 : (packed ? 1 : 0);          // what the chain launches are told about their images
    const int32_t pk = packed ? 1 : 0;
    // work split of the fused global-edge kernels: made with the graph (a side-stream launch of graph construction) or here
    const int32_t* cuts = g.cuts;
    if (!cuts) {
        CK(pamnet_seg_cuts_i32(g.g_ptr, g.g_row, g.n, g.eg, t.cuts, nullptr, st));
        cuts = t.cuts;
    }
    for (int64_t k = 0; k < n_layer; ++k) {
        // ---------------- global layer (layers/global_message_passing.py:33-56)
        const float* const* gp = gparams + k * NG;
        const GlobalSaved s = carve_global(saved + k * (gs + ls), g);
        const float* wpg[2] = {gp[2], gp[2] + D};
        // the head of every layer but the first runs inside the preceding layer's node chain (x_out tile still on chip)
        // (recompute A/B: the node planes of a training forward go to the saved arena instead of the scratch)
        float* const Pk = (keep && s.Pg) ? s.Pg : t.P;
        if (k == 0) CK(pamnet_node_pre_fwd_f32(x, g.n, gp[0], gp[1], wpg, 3 * D, 2, sv(s.Zx1), t.x1, Pk, st));
        // message MLP + add-aggregation in one kernel: x2 = x1 + sum_{e -> i} msg_e, the messages never leave the chip
        // (ld = 0: the weight argument is its fragment image)
        if (eimg)
            CK(pamnet_global_edge_agg_fwd_f32(e_g, g.eg, g.n, eimg_store[k].we, 0, gp[3], eimg_store[k].wea, 0, Pk, Pk + g.n * D,
                                              g.g_ptr, g.g_row, g.g_col, cuts, t.x1, sv(s.z), sv(s.ea), s.x2, st));
        else
            CK(pamnet_global_edge_agg_fwd_f32(e_g, g.eg, g.n, gp[2] + 2 * D, 3 * D, gp[3], gp[4], D, Pk, Pk + g.n * D,
                                              g.g_ptr, g.g_row, g.g_col, cuts, t.x1, sv(s.z), sv(s.ea), s.x2, st));
        const float* const* lp = lparams + k * NL;
        const LocalSaved q = carve_local(saved + k * (gs + ls) + gs, g);
        const float* wpl[4] = {lp[2], lp[4], lp[2] + D, lp[4] + D};
        if (ride) {
            const float* mp[4] = {lp[6], lp[7], lp[8], lp[9]};
            float* mo[3] = {sv(q.z1), sv(q.z2), q.s};
            CK(pamnet_node_tail_fwd_rider_f32(s.x2, x, g.n, img[k].gt, gp + GT + 10, gp[GT + 20], gp[GT + 21], gp[GT + 22],
                                              sv(s.Z), sv(s.R), s.xout, img[k].lh[0], lp[1], img[k].lh + 1, 3 * D, 4,
                                              sv(q.Zx1), t.x1, t.P, e_sbf, g.tp, (k == 0 && cb0) ? 0 : mlp_half,
                                              (k == 0 && cb0) ? mlp_tiles : mlp_tiles - mlp_half, mp, mo, rider_wgs, pkc, st));
        } else {
            CK(pamnet_node_tail_fwd_f32(s.x2, x, g.n, packed ? img[k].gt : gp + GT, gp + GT + 10, gp[GT + 20], gp[GT + 21],
                                        gp[GT + 22], sv(s.Z), sv(s.R), s.xout, nullptr, nullptr,
                                        packed ? img[k].lh[0] : lp[0], lp[1], packed ? img[k].lh + 1 : wpl, 3 * D, 4,
                                        sv(q.Zx1), t.x1, t.P, pkc, st));
        }
        x = s.xout;
        // ---------------- local layer (layers/local_message_passing.py:36-66); its head ran in the chain above
        const float* wq[4] = {lp[2] + 2 * D, lp[4] + 2 * D, lp[10], lp[11]};
        const int64_t ldq[4] = {3 * D, 3 * D, D, D};
        const int64_t ld0[4] = {0, 0, 0, 0};
        const float* planes[4] = {t.P, t.P + g.n * D, t.P + 2 * g.n * D, t.P + 3 * g.n * D};
        CK(pamnet_local_edge_fwd_f32(rbf_e, g.el, eimg ? eimg_store[k].wq : wq, eimg ? ld0 : ldq, lp[3], lp[5], planes, g.l_row,
                                     g.l_col, sv(q.zji), sv(q.zkj), sv(q.q2), q.q3, t.mji, q.mnb, st));
        if (forked) HK(hipStreamWaitEvent(as_stream(st), reinterpret_cast<hipEvent_t>(aux_events[1 + k]), 0));
        // both aggregations of the local layer (rows -> edges -> nodes) in one launch; m_t is a backward-only save
        const bool agg_in = packed && fuse_local_agg(g);      // x2 formed by the chain launch's own tiles
        const pamnet_local_agg la{t.mji, q.mnb, q.s, q.q3, t.x1, g.t_ptr, g.t_col, g.l_ptr, sv(q.mt)};
        if (!agg_in)
            CK(pamnet_local_agg_fwd_f32(t.mji, q.mnb, q.s, q.q3, g.t_ptr, g.t_col, g.l_ptr, t.x1, g.n, sv(q.mt), q.x2, st));
        if (k + 1 < n_layer) {
            const float* const* gn = gparams + (k + 1) * NG;
            const GlobalSaved sn = carve_global(saved + (k + 1) * (gs + ls), g);
            const float* wpn[2] = {gn[2], gn[2] + D};
            float* const Pn = (keep && sn.Pg) ? sn.Pg : t.P;         // the next global layer's node planes
            if (ride) {
                const float* const* ln = lparams + (k + 1) * NL;
                const LocalSaved qn = carve_local(saved + (k + 1) * (gs + ls) + gs, g);
                const float* mp[4] = {ln[6], ln[7], ln[8], ln[9]};
                float* mo[3] = {sv(qn.z1), sv(qn.z2), qn.s};
                if (agg_in)
                    CK(pamnet_node_tail_fwd_agg_f32(q.x2, x, g.n, img[k].lt, lp + LT + 10, lp[LT + 20], lp[LT + 21], lp[LT + 22],
                                                    sv(q.Z), sv(q.R), q.xout, img[k].nh[0], gn[1], img[k].nh + 1, 3 * D, 2,
                                                    sv(sn.Zx1), t.x1, Pn, e_sbf, g.tp, 0, mlp_half, mp, mo, rider_wgs, pkc, &la,
                                                    st));
                else
                    CK(pamnet_node_tail_fwd_rider_f32(q.x2, x, g.n, img[k].lt, lp + LT + 10, lp[LT + 20], lp[LT + 21],
                                                      lp[LT + 22], sv(q.Z), sv(q.R), q.xout, img[k].nh[0], gn[1], img[k].nh + 1,
                                                      3 * D, 2, sv(sn.Zx1), t.x1, Pn, e_sbf, g.tp, 0, mlp_half, mp, mo, rider_wgs,
                                                      pkc, st));
            } else if (agg_in) {
                CK(pamnet_node_tail_fwd_agg_f32(q.x2, x, g.n, img[k].lt, lp + LT + 10, lp[LT + 20], lp[LT + 21], lp[LT + 22],
                                                sv(q.Z), sv(q.R), q.xout, img[k].nh[0], gn[1], img[k].nh + 1, 3 * D, 2, sv(sn.Zx1),
                                                t.x1, Pn, nullptr, 0, 0, 0, nullptr, nullptr, 0, pkc, &la, st));
            } else {
                CK(pamnet_node_tail_fwd_f32(q.x2, x, g.n, packed ? img[k].lt : lp + LT, lp + LT + 10, lp[LT + 20], lp[LT + 21],
                                            lp[LT + 22], sv(q.Z), sv(q.R), q.xout, nullptr, nullptr,
                                            packed ? img[k].nh[0] : gn[0], gn[1],
                                            packed ? img[k].nh + 1 : wpn, 3 * D, 2, sv(sn.Zx1), t.x1, Pn, pkc, st));
            }
        } else if (agg_in) {
            CK(pamnet_node_tail_fwd_agg_f32(q.x2, x, g.n, img[k].lt, lp + LT + 10, lp[LT + 20], lp[LT + 21], lp[LT + 22], sv(q.Z),
                                            sv(q.R), q.xout, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, 0, 0,
                                            0, nullptr, nullptr, 0, pkc, &la, st));
        } else {
            CK(pamnet_node_tail_fwd_f32(q.x2, x, g.n, packed ? img[k].lt : lp + LT, lp + LT + 10, lp[LT + 20], lp[LT + 21],
                                        lp[LT + 22], sv(q.Z), sv(q.R), q.xout, nullptr, nullptr, nullptr, nullptr,
                                        nullptr, 0, 0, nullptr, nullptr, nullptr, pkc, st));
        }
        x = q.xout;
    }
    // the head branch (mlp_out + W_out / W) of all 2 n_layer chains in one launch: 2L x ceil(n/16) workgroups
    {
        const int64_t nh = 2 * n_layer;
        std::vector<const float*> hx(nh), hw(3 * nh), hb(3 * nh), hwo(nh), hbo(nh), hwa(nh);
        std::vector<float*> hz(nh), ho(nh), ha(nh);
        for (int64_t k = 0; k < n_layer; ++k) {
            const float* const* gp = gparams + k * NG;
            const float* const* lp = lparams + k * NL;
            const GlobalSaved s = carve_global(saved + k * (gs + ls), g);
            const LocalSaved q = carve_local(saved + k * (gs + ls) + gs, g);
            for (int side = 0; side < 2; ++side) {
                const int64_t l = 2 * k + side;
                const float* const* tp = side ? lp + LT : gp + GT;
                hx[l] = side ? q.xout : s.xout;
                for (int i = 0; i < 3; ++i) {
                    hw[3 * l + i] = packed ? (side ? img[k].lt[7 + i] : img[k].gt[7 + i]) : tp[7 + i];
                    hb[3 * l + i] = tp[10 + 7 + i];
                }
                hwo[l] = tp[20], hbo[l] = tp[21], hwa[l] = tp[22];
                hz[l] = sv(side ? q.Z : s.Z);
                ho[l] = outs + l * g.n, ha[l] = atts + l * g.n;
            }
        }
        CK(pamnet_node_heads_fwd_f32(nh, hx.data(), hw.data(), hb.data(), hwo.data(), hbo.data(), hwa.data(), hz.data(),
                                     ho.data(), ha.data(), g.n, pk, st));
    }
    return PAMNET_OK;
}

// d_outs / d_atts: [2L][n].  ggrads / lgrads: gradient buffers in the same tables as the parameters (written, not
// accumulated).  d_x0 [n,128] written; d_eg, d_rbf, d_sbf written (first layer processed) then accumulated.
extern "C" int pamnet_stack_bwd_f32(const int64_t* sizes, const int32_t* const* graph_idx, int64_t n_layer,
                                    const float* x0, const float* e_g, const float* rbf_e, const float* e_sbf,
                                    const float* const* gparams, const float* const* lparams, const float* saved,
                                    float* temp, const float* d_outs, const float* d_atts, float* const* ggrads,
                                    float* const* lgrads, float* d_x0, float* d_eg, float* d_rbf, float* d_sbf,
                                    float* wpack, void* const* layer_done, pamnet_stream_t st) {
    Graph g;
    CK(fill_graph(g, sizes, graph_idx));
    if (n_layer < 1) return PAMNET_EINVAL;
    if (!x0 || !e_g || !rbf_e || !e_sbf || !gparams || !lparams || !saved || !temp || !d_outs || !d_atts || !ggrads ||
        !lgrads || !d_x0 || !d_eg || !d_rbf || !d_sbf)
        return PAMNET_ENULL;
    const Temp t = carve_temp(temp, g);
    const int64_t gs = al(global_saved_floats(g)), ls = al(local_saved_floats(g));
    // transposed weight images of both node chains of every layer (optional `wpack` arena), one pack launch
    const bool packed = wpack != nullptr;
    struct PairImgT {
        const float *gt[10], *lt[10], *gh[3], *lh[5];      // chains; heads: {projection blocks ..., Wx1}
    };
    std::vector<PairImgT> img_store(packed ? (size_t)n_layer : 0);
    PairImgT* img = packed ? img_store.data() : nullptr;
    // transposed fragment images of W_e, W_ea for the plain global-edge backward (the weight-gradient-forming kernel of large
    // batches keeps its own two-pieces-in-registers loader)
    const bool eimg = packed && edge_images() && !edge_wgrad(g);
    // ... and of the local layer's backward pair: mlp_sbf's W1, W2 and the local edge step's four slices (bf16x3 fragments)
    const bool limg = packed && edge_images() && g.tp > 0 && g.el > 0;      // (the paired launch exists)
    struct EdgeImgT {
        const float *we, *wea, *w1, *w2, *wq[4];
    };
    std::vector<EdgeImgT> eimg_store((eimg || limg) ? (size_t)n_layer : 0);
    // the chain launches of single-round batches on the bf16 matrix pipe (node_tail_bwd_bf16_kernel): bf16x3 images for everything
    // they multiply by; fp32 images for the heads' matrices 7..9 and for the first layer's stand-alone head backward
    static const bool bwd_on = [] { const char* e = getenv("PAMNET_CHAIN_BF16"); return !e || atoi(e) != 2; }();   // (2: forward only)
    const bool cbb = packed && chain_bf16() && bwd_on && (g.n + 15) / 16 <= chain_bf16_tiles();
    const int64_t pcs = cbb ? PAMNET_CHAIN_PIECES : 0;
    if (packed) {
        PackList pl(wpack, 1, st, cbb, wpack + n_layer * PACK_FLOATS_PER_PAIR);
        for (int64_t k = n_layer - 1; k >= 0; --k) {
            const float* const* lp = lparams + k * NL;
            const float* const* gp = gparams + k * NG;
            for (int i = 0; i < 10; ++i) img[k].lt[i] = pl.add(lp[LT + i], D, i >= 7);      // (7..9: the heads' fp32 images)
            img[k].lh[0] = pl.add(lp[2], 3 * D), img[k].lh[1] = pl.add(lp[4], 3 * D);
            img[k].lh[2] = pl.add(lp[2] + D, 3 * D), img[k].lh[3] = pl.add(lp[4] + D, 3 * D);
            img[k].lh[4] = pl.add(lp[0], D);
            for (int i = 0; i < 10; ++i) img[k].gt[i] = pl.add(gp[GT + i], D, i >= 7);
            img[k].gh[0] = pl.add(gp[2], 3 * D, k == 0), img[k].gh[1] = pl.add(gp[2] + D, 3 * D, k == 0);
            img[k].gh[2] = pl.add(gp[0], D, k == 0);
            if (eimg) eimg_store[k].we = pl.add_edge(gp[2] + 2 * D, 3 * D), eimg_store[k].wea = pl.add_edge(gp[4], D);
            if (limg) {
                EdgeImgT& ei = eimg_store[k];
                ei.w1 = pl.add_edge(lp[6], D), ei.w2 = pl.add_edge(lp[8], D);
                ei.wq[0] = pl.add_edge(lp[2] + 2 * D, 3 * D), ei.wq[1] = pl.add_edge(lp[4] + 2 * D, 3 * D);
                ei.wq[2] = pl.add_edge(lp[10], D), ei.wq[3] = pl.add_edge(lp[11], D);
            }
        }
        pl.flush();
        CK(pl.rc);
    }
    const int32_t pk = packed ? 1 : 0;
    // head branch of all 2 n_layer chains first, in one launch: it needs only d out / d att
    {
        const int64_t nh = 2 * n_layer;
        std::vector<const float*> ho(nh), ha(nh), hw(3 * nh), hwo(nh), hwa(nh), hz(nh);
        std::vector<float*> hd(nh), hg(nh), hp(nh);
        for (int64_t k = 0; k < n_layer; ++k) {
            const float* const* gp = gparams + k * NG;
            const float* const* lp = lparams + k * NL;
            const GlobalSaved s = carve_global(const_cast<float*>(saved) + k * (gs + ls), g);
            const LocalSaved q = carve_local(const_cast<float*>(saved) + k * (gs + ls) + gs, g);
            for (int side = 0; side < 2; ++side) {
                const int64_t l = 2 * k + side;
                const float* const* tp = side ? lp + LT : gp + GT;
                ho[l] = d_outs + l * g.n, ha[l] = d_atts + l * g.n;
                for (int i = 0; i < 3; ++i) hw[3 * l + i] = packed ? (side ? img[k].lt[7 + i] : img[k].gt[7 + i]) : tp[7 + i];
                hwo[l] = tp[20], hwa[l] = tp[22];
                hz[l] = side ? q.Z : s.Z;
                hd[l] = side ? q.hdz : s.hdz, hg[l] = side ? q.gh : s.gh, hp[l] = side ? q.hp : s.hp;
            }
        }
        CK(pamnet_node_heads_bwd_f32(nh, ho.data(), ha.data(), hw.data(), hwo.data(), hwa.data(), hz.data(), hd.data(),
                                     hg.data(), hp.data(), g.n, pk, st));
    }
    // Chain backward launches.  With packed weights the backward of a layer's head (node_pre_bwd) is fused into the
    // backward of the chain that produced that layer's input (same row tiles, the head's d x stays on chip): per layer
    // pair  [chain L_k] local edges [head L_k + chain G_k] wgrad L_k  global edges [head G_k + chain L_{k-1}] wgrad G_k.
    // The chain gradients alternate between two buffers because a fused launch writes the next chain's dZ before the
    // previous chain's weight-gradient launch (which also needs that launch's dZx1) has consumed its own.
    const bool fuse = packed;
    // weight-gradient batches: the fixed-order reduction of each batch rides in the next batch's launch
    int64_t ctx_bytes = 0, rider_bytes = 0;
    CK(pamnet_wgrad_ctx_bytes(&ctx_bytes));
    CK(pamnet_wgrad_rider_bytes(&rider_bytes));
    std::vector<char> wctx((size_t)ctx_bytes, 0), rider((size_t)rider_bytes, 0);
    const bool ride = fuse && riders_fit(g);
    const int32_t* cuts = g.cuts;          // work split of the fused global-edge kernels (as in the forward)
    if (!cuts) {
        CK(pamnet_seg_cuts_i32(g.g_ptr, g.g_row, g.n, g.eg, t.cuts, nullptr, st));
        cuts = t.cuts;
    }
    float* parts[2] = {t.partial, t.partial2};
    int pflip = 0;
    Jobs pair_jobs;                       // a pair's local-layer jobs waiting for its merged launch
    HeadGrads pair_head{nullptr, nullptr, nullptr, nullptr};
    int64_t rider_a_slots = 0;            // slots of the local chain's riders (the global chain's start behind them)
    const float* d_xout = nullptr;        // nothing consumes the last layer's node features (models.py:196-224)
    float* dx_bufs[2] = {t.dxa, t.dxb};
    float* dz_bufs[2] = {t.dZ, t.dZ2};
    int flip = 0, zflip = 0;
    float* dz_local = dz_bufs[zflip];     // dZ of the local chain of the current pair
    if (fuse) {
        const LocalSaved ql = carve_local(const_cast<float*>(saved) + (n_layer - 1) * (gs + ls) + gs, g);
        CK(pamnet_node_tail_main_bwd_f32(nullptr, ql.gh, g.n, img[n_layer - 1].lt, ql.Z, dz_local, t.dx2, t.dresx, cbb ? 2 : pk, st));
    }
    for (int64_t k = n_layer - 1; k >= 0; --k) {
        const GlobalSaved s = carve_global(const_cast<float*>(saved) + k * (gs + ls), g);
        const LocalSaved q = carve_local(const_cast<float*>(saved) + k * (gs + ls) + gs, g);
        const int acc = (k != n_layer - 1) ? 1 : 0;
        float* dz_global = nullptr;
        // ================= local layer backward
        {
            const float* const* lp = lparams + k * NL;
            float* const* lg = lgrads + k * NL;
            const float* x_in = s.xout;           // input of the local layer = output of this pair's global layer
            if (!fuse) {
                dz_local = t.dZ;
                CK(pamnet_node_tail_main_bwd_f32(d_xout, q.gh, g.n, lp + LT, q.Z, dz_local, t.dx2, t.dresx, pk, st));
            }
            // d m_t = d x2[i] * q3,  d q3 = d x2[i] * m_t,  d s = m_nb[idx] * d m_t[e],  d m_nb = transposed sum: one launch
            CK(pamnet_local_agg_bwd_f32(t.dx2, g.l_row, q.q3, q.mt, q.mnb, q.s, g.t_ptr, g.t_col, g.t_row, g.tT_ptr,
                                        g.tT_perm, g.tT_edge, g.tT_node, g.el, t.dmt, t.dq3, t.ds, t.dmnb, st));
            // the triplet / pair MLP's backward and the local edge stage's: independent of each other, one launch
            const float* wq[4] = {lp[2] + 2 * D, lp[4] + 2 * D, lp[10], lp[11]};
            const int64_t ldq[4] = {3 * D, 3 * D, D, D};
            const int64_t ld0[4] = {0, 0, 0, 0};
            if (limg)
                CK(pamnet_local_bwd_pair_f32(t.ds, g.tp, q.z1, q.z2, eimg_store[k].w1, eimg_store[k].w2, t.dz1, t.dz2, d_sbf,
                                             acc | PAMNET_WEIGHT_IMAGES, t.dmt, t.dmnb, t.dq3, g.el, q.zji, q.zkj, q.q2,
                                             eimg_store[k].wq, ld0, t.dzji, t.dzkj, t.dq2, d_rbf, acc, st));
            else
                CK(pamnet_local_bwd_pair_f32(t.ds, g.tp, q.z1, q.z2, lp[6], lp[8], t.dz1, t.dz2, d_sbf, acc, t.dmt, t.dmnb,
                                             t.dq3, g.el, q.zji, q.zkj, q.q2, wq, ldq, t.dzji, t.dzkj, t.dq2, d_rbf, acc, st));
            const int64_t pl = g.n * D;
            const float* sa[4] = {t.dzji, t.dzkj, t.dzji, t.dzkj};
            const int32_t* sp[4] = {nullptr, nullptr, g.lT_perm, g.lT_perm};
            const int32_t* sr[4] = {g.l_ptr, g.l_ptr, g.lT_ptr, g.lT_ptr};
            const bool gather_l = fuse && fuse_segsum(g);      // the four sums inside the fused launch below
            if (!gather_l) {
                float* so[4] = {t.dP, t.dP + pl, t.dP + 2 * pl, t.dP + 3 * pl};
                CK(pamnet_segment_sum_multi_f32(4, so, sa, sp, sr, g.n, D, st));
            }
            if (fuse) {
                // head of the local layer + the global chain of this pair
                zflip ^= 1;
                dz_global = dz_bufs[zflip];
                if (ride) {                                   // this layer's chain gradients ride in the launch below
                    Jobs jr;
                    tail_jobs(jr, g, dz_local, q.hdz, q.x2, q.Z, q.R, q.xout, lg + LT, 0, rider_jobs());
                    CK(plan_rider(jr, t.rider_partial, g, rider.data(), &rider_a_slots));
                }
                if (gather_l)
                    CK(pamnet_node_pre_tail_bwd_gather_f32(t.dP, sa, sr, sp, t.dx2, t.dresx, g.n, img[k].lh[4], img[k].lh, 4 | pcs, q.Zx1,
                                                           t.dZx1, s.gh, img[k].gt, s.Z, dz_global, t.dx2, t.dresx,
                                                           ride ? rider.data() : nullptr, st));
                else
                    CK(pamnet_node_pre_tail_bwd_f32(t.dP, t.dx2, t.dresx, g.n, img[k].lh[4], img[k].lh, 4 | pcs, q.Zx1, t.dZx1, s.gh,
                                                    img[k].gt, s.Z, dz_global, t.dx2, t.dresx, ride ? rider.data() : nullptr, st));
                if (ride) CK(pamnet_wgrad_rider_enqueue_f32(wctx.data(), rider.data()));
            } else {
                const float* wpl[4] = {lp[2], lp[4], lp[2] + D, lp[4] + D};
                float* dx = dx_bufs[flip];
                CK(pamnet_node_pre_bwd_f32(t.dP, t.dx2, t.dresx, g.n, lp[0], wpl, 3 * D, 4, q.Zx1, t.dZx1, dx, pk, st));
                d_xout = dx;
                flip ^= 1;
            }
            // With riders a layer pair's own jobs -- 11 of the local layer, 5 of the global one -- are ONE launch behind the
            // pair's second chain launch (k >= 1; the last pair's global layer keeps its ten tail jobs: two launches there):
            // a weight-gradient launch at this batch size is ~14 us of work in ~30 us (prologue, partial stores, the riding
            // reductions), and the local layer's operands (t.dP, t.dZx1, the edge / row gradients) stay untouched until the
            // next pair's local phase -- the global phase writes its own d P / d Zx1 (t.dPg, t.dZx1g).
            const bool merged = ride && k > 0;
            Jobs j;
            tail_jobs(j, g, dz_local, q.hdz, q.x2, q.Z, q.R, q.xout, lg + LT, ride ? rider_jobs() : 0, 10);
            j.add(t.dZx1, x_in, 0, g.n, lg[0], D, lg[1]);
            j.add(t.dP, q.Zx1, 1, g.n, lg[2], 3 * D, nullptr);
            j.add(t.dP + pl, q.Zx1, 1, g.n, lg[4], 3 * D, nullptr);
            j.add(t.dP + 2 * pl, q.Zx1, 1, g.n, lg[2] + D, 3 * D, nullptr);
            j.add(t.dP + 3 * pl, q.Zx1, 1, g.n, lg[4] + D, 3 * D, nullptr);
            j.add(t.dzji, rbf_e, 0, g.el, lg[2] + 2 * D, 3 * D, lg[3]);
            j.add(t.dzkj, rbf_e, 0, g.el, lg[4] + 2 * D, 3 * D, lg[5]);
            j.add(t.dq2, rbf_e, 0, g.el, lg[10], D, nullptr);
            j.add(t.dq3, rbf_e, 0, g.el, lg[11], D, nullptr);
            j.add(t.dz2, q.z1, 1, g.tp, lg[8], D, lg[9]);
            j.add(t.dz1, e_sbf, 0, g.tp, lg[6], D, lg[7]);
            const HeadGrads hl{q.hp, lg[LT + 20], lg[LT + 22], lg[LT + 21]};
            if (!merged) {
                CK(run_jobs(j, parts[pflip], g, hl, HeadGrads{nullptr, nullptr, nullptr, nullptr}, wctx.data(), st));
                pflip ^= 1;
                // that launch also reduced the weight gradients of the previous pair's global layer: pair k+1 is complete
                if (k + 1 < n_layer && layer_done && layer_done[k + 1]) {
                    HK(hipEventRecord(reinterpret_cast<hipEvent_t>(layer_done[k + 1]), as_stream(st)));
                }
            } else {
                pair_jobs = j;
                pair_head = hl;
            }
        }
        // ================= global layer backward
        {
            const float* const* gp = gparams + k * NG;
            float* const* gg = ggrads + k * NG;
            const float* x_in = (k == 0) ? x0 : carve_local(const_cast<float*>(saved) + (k - 1) * (gs + ls) + gs, g).xout;
            if (!fuse) {
                dz_global = t.dZ;
                CK(pamnet_node_tail_main_bwd_f32(d_xout, s.gh, g.n, gp + GT, s.Z, dz_global, t.dx2, t.dresx, pk, st));
            }
            // d z, d ea, d e and the target-side reduction d P_i in one kernel; the source-side one walks the transposed CSR
            const int64_t pl = g.n * D;
            const bool ewg = edge_wgrad(g);
            if (ewg) {
                // ... and the step's own weight gradients: partial tiles per workgroup, summed by the next weight-gradient launch
                int64_t efloats = 0, eslots = 0;
                CK(pamnet_global_edge_agg_wg_floats(g.eg, &efloats, &eslots));
                const float *zk = s.z, *eak = s.ea;
                if (s.Pg) {
                    // recompute A/B: z and ea of this layer once more, by the kernel that made them in the forward, into scratch
                    // the fused backward does not use (the forward's message buffer, the d ea rows it no longer writes)
                    CK(pamnet_global_edge_agg_fwd_f32(e_g, g.eg, g.n, gp[2] + 2 * D, 3 * D, gp[3], gp[4], D, s.Pg, s.Pg + g.n * D,
                                                      g.g_ptr, g.g_row, g.g_col, cuts, nullptr, t.msg, t.dea, t.dump, st));
                    zk = t.msg, eak = t.dea;
                }
                CK(pamnet_global_edge_agg_bwd_wg_f32(t.dx2, g.eg, g.n, g.g_ptr, g.g_row, cuts, zk, eak, e_g, gp[2] + 2 * D, 3 * D,
                                                     gp[4], D, t.dz, d_eg, acc, t.dPg, t.edge_partial, st));
                CK(pamnet_wgrad_edge_enqueue_f32(wctx.data(), eslots, gg[2] + 2 * D, 3 * D, gg[3], gg[4], D, t.edge_partial));
            } else {
                if (eimg)
                    CK(pamnet_global_edge_agg_bwd_f32(t.dx2, g.eg, g.n, g.g_ptr, g.g_row, cuts, s.z, s.ea, eimg_store[k].we, 0,
                                                      eimg_store[k].wea, 0, t.dz, t.dea, d_eg, acc, t.dPg, st));
                else
                    CK(pamnet_global_edge_agg_bwd_f32(t.dx2, g.eg, g.n, g.g_ptr, g.g_row, cuts, s.z, s.ea, gp[2] + 2 * D, 3 * D,
                                                      gp[4], D, t.dz, t.dea, d_eg, acc, t.dPg, st));
            }
            const bool gather_g = fuse && k > 0 && fuse_segsum(g);   // the source-side sum inside the fused launch below
            if (!gather_g)
                CK(pamnet_segment_sum_f32(t.dPg + pl, nullptr, t.dz, nullptr, nullptr, nullptr, g.gT_perm, g.gT_ptr, g.n, D, st));
            if (fuse && k > 0) {
                // head of the global layer + the local chain of the previous pair
                const LocalSaved qp = carve_local(const_cast<float*>(saved) + (k - 1) * (gs + ls) + gs, g);
                zflip ^= 1;
                dz_local = dz_bufs[zflip];
                if (ride) {
                    // (its slots lie right behind the local chain's riders: both wait for the pair's merged launch)
                    Jobs jr;
                    tail_jobs(jr, g, dz_global, s.hdz, s.x2, s.Z, s.R, s.xout, gg + GT, 0, rider_jobs());
                    CK(plan_rider(jr, t.rider_partial + rider_a_slots * (D * D + 2 * D), g, rider.data(), nullptr));
                }
                if (gather_g) {
                    const float* ga[2] = {nullptr, t.dz};
                    const int32_t* gr[2] = {nullptr, g.gT_ptr};
                    const int32_t* gq[2] = {nullptr, g.gT_perm};
                    CK(pamnet_node_pre_tail_bwd_gather_f32(t.dPg, ga, gr, gq, t.dx2, t.dresx, g.n, img[k].gh[2], img[k].gh, 2 | pcs, s.Zx1,
                                                           t.dZx1g, qp.gh, img[k - 1].lt, qp.Z, dz_local, t.dx2, t.dresx,
                                                           ride ? rider.data() : nullptr, st));
                } else {
                    CK(pamnet_node_pre_tail_bwd_f32(t.dPg, t.dx2, t.dresx, g.n, img[k].gh[2], img[k].gh, 2 | pcs, s.Zx1, t.dZx1g, qp.gh,
                                                    img[k - 1].lt, qp.Z, dz_local, t.dx2, t.dresx,
                                                    ride ? rider.data() : nullptr, st));
                }
                if (ride) CK(pamnet_wgrad_rider_enqueue_f32(wctx.data(), rider.data()));
            } else {
                const float* wpg[2] = {gp[2], gp[2] + D};
                float* dx = (k == 0) ? d_x0 : dx_bufs[flip];
                CK(pamnet_node_pre_bwd_f32(t.dPg, t.dx2, t.dresx, g.n, packed ? img[k].gh[2] : gp[0],
                                           packed ? img[k].gh : wpg, 3 * D, 2, s.Zx1, t.dZx1g, dx, pk, st));
                d_xout = dx;
                flip ^= 1;
            }
            const bool merged = ride && k > 0;
            Jobs j;
            if (merged) j = pair_jobs;                        // the local layer's own jobs, parked above
            tail_jobs(j, g, dz_global, s.hdz, s.x2, s.Z, s.R, s.xout, gg + GT, (ride && k > 0) ? rider_jobs() : 0, 10);
            j.add(t.dZx1g, x_in, 0, g.n, gg[0], D, gg[1]);
            j.add(t.dPg, s.Zx1, 1, g.n, gg[2], 3 * D, nullptr);
            j.add(t.dPg + pl, s.Zx1, 1, g.n, gg[2] + D, 3 * D, nullptr);
            if (!ewg) {
                j.add(t.dz, e_g, 0, g.eg, gg[2] + 2 * D, 3 * D, gg[3]);
                j.add(t.dea, e_g, 0, g.eg, gg[4], D, nullptr);
            }
            const HeadGrads hg{s.hp, gg[GT + 20], gg[GT + 22], gg[GT + 21]};
            CK(run_jobs(j, parts[pflip], g, hg, merged ? pair_head : HeadGrads{nullptr, nullptr, nullptr, nullptr}, wctx.data(),
                        st));
            pflip ^= 1;
            // (merged: that launch also reduced the previous pair's merged batch: pair k+1 is complete)
            if (merged && k + 1 < n_layer && layer_done && layer_done[k + 1]) {
                HK(hipEventRecord(reinterpret_cast<hipEvent_t>(layer_done[k + 1]), as_stream(st)));
            }
        }
    }
    CK(pamnet_wgrad_flush_f32(wctx.data(), st));
    if (layer_done && layer_done[0]) {
        HK(hipEventRecord(reinterpret_cast<hipEvent_t>(layer_done[0]), as_stream(st)));
    }
    return PAMNET_OK;
}
