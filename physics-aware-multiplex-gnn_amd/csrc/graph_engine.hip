// Graph-construction engine: the whole parameter-independent front of PAMNet.forward (models.py:62-98, 104-177 --
// remove_self_loops / radius / knn / cutoff masks / SparseTensor CSR / triplets + pairs / angles -- and the spherical
// basis of layers/basic.py:107-116) enqueued by ONE C call.
//
// The reference drives these steps from Python through four third-party wheels and reads data-dependent sizes back to
// the host in between (boolean masks, repeat_interleave).  pamnet_amd/graph.py restates the sequence as ~30 kernel
// launches, each with its own buffer allocation; at the RNA batch that orchestration alone was 0.36 ms of a 1.28 ms
// step and the step was bound by the host (profiles/r03_host_phases_rna_before.txt).  When the sizes of a batch are known
// to the host (pamnet_amd.store: per-graph counts taken once per dataset) nothing has to come back from the device, so
// the sequence is a straight line: this file is that line in C++ -- one caller-owned int32 arena carved by a bump
// allocator, the same extern "C" kernels entry points as the Python path in the same order (bit-identical results:
// tests/test_graph_engine.py), capped fills and clamped pointers, and the deferred size / validity check in the flag word.
//
// pamnet_graph_plan and pamnet_graph_build_i32 run the SAME function (once without launching, to lay the arena out):
// the layout cannot drift from the launches.  Nothing is allocated, nothing synchronises, no state is kept.
#include "common.h"

namespace {

constexpr int64_t RAD_W = 42;                 // num_spherical * num_radial of the compiled basis

__global__ __launch_bounds__(256) void clamp_copy_kernel(const int32_t* __restrict__ in, int32_t* __restrict__ out, int64_t n,
                                                         int32_t cap) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i] < cap ? in[i] : cap;
}

// pos = x[:, :3] (models.py:120,141); sign = where(pos.x > 40, -1, +1) (models.py:124; nullable)
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ x, int64_t width, int64_t n,
                                                         float* __restrict__ pos, float* __restrict__ sign) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float px = x[i * width], py = x[i * width + 1], pz = x[i * width + 2];
    pos[3 * i] = px, pos[3 * i + 1] = py, pos[3 * i + 2] = pz;
    if (sign) sign[i] = px > 40.0f ? -1.0f : 1.0f;
}

inline unsigned blocks(int64_t n) { return (unsigned)((n + 255) / 256 > 0 ? (n + 255) / 256 : 1); }

struct Run {
    const pamnet_graph_desc& d;
    int32_t* base;            // arena (null while planning)
    float* sbf;               // [tp, 42] (nullable)
    pamnet_stream_t stream;
    bool dry;                 // plan only
    int64_t off = 0;
    int64_t* lay;             // layout table (PAMNET_GRAPH_FIELDS entries)

    int64_t take(int64_t n) {
        const int64_t o = off;
        off += (n + 3) / 4 * 4;               // 16-byte aligned slices
        return o;
    }
    int32_t* I(int64_t o) const { return base + o; }
    float* F(int64_t o) const { return reinterpret_cast<float*>(base + o); }
    void field(int f, int64_t o) { lay[f] = o; }
};

#define GO(call)                         \
    do {                                 \
        if (!r.dry) {                    \
            const int rc__ = (call);     \
            if (rc__) return rc__;       \
        }                                \
    } while (0)
#define GOK(...)                                               \
    do {                                                       \
        if (!r.dry) {                                          \
            hipLaunchKernelGGL(__VA_ARGS__);                   \
            const hipError_t e__ = hipGetLastError();          \
            if (e__ != hipSuccess) return (int)e__;            \
        }                                                      \
    } while (0)

// counts [n] -> exclusive scan [n + 1]
int scan(Run& r, int64_t counts, int64_t n, int64_t& out) {
    out = r.take(n + 1);
    const int64_t tmp = r.take((n + 4095) / 4096 + 1);
    GO(pamnet_exclusive_scan_i32(r.I(counts), r.I(out), n, r.I(tmp), r.stream));
    return PAMNET_OK;
}

// stable counting sort of keys [m] over `rows` -> ptr [rows + 1], perm [m]  (ptr / perm may be pre-assigned: pass >= 0).
// `cursor`: rows + 2 counters inside the batch's zero-filled block (no fill launch per sort).
int csr(Run& r, int64_t keys, int64_t m, int64_t rows, int64_t cursor, int64_t& ptr, int64_t& perm) {
    if (ptr < 0) ptr = r.take(rows + 1);
    if (perm < 0) perm = r.take(m);
    const int64_t perm_tmp = r.take(m), tmp = r.take((rows + 4095) / 4096 + 1);
    GO(pamnet_csr_from_keys_z_i32(r.I(keys), m, rows, r.I(ptr), r.I(perm), r.I(cursor), r.I(perm_tmp), r.I(tmp), r.stream));
    return PAMNET_OK;
}

int clamp(Run& r, int64_t in, int64_t n, int64_t cap, int64_t& out) {
    out = r.take(n);
    GOK(clamp_copy_kernel, dim3(blocks(n)), dim3(256), 0, as_stream(r.stream), r.I(in), r.I(out), n, (int32_t)cap);
    return PAMNET_OK;
}

int run(Run& r) {
    const pamnet_graph_desc& d = r.d;
    const int64_t n = d.n, ng = d.n_graphs, eg = d.eg, el = d.el, tp = d.tp;
    const bool grad = d.need_grad != 0;
    const int wt = d.with_triplets ? 1 : 0;
    for (int f = 0; f < PAMNET_GRAPH_FIELDS; ++f) r.lay[f] = -1;

    // ---- the reference's index tensors as int32 + the per-graph node pointer + the validity / self-loop flags
    const int64_t node_graph = r.take(n), types = r.take(n), gf = r.take(ng + 7);
    const int64_t src0 = r.take(d.n_bonds), dst0 = r.take(d.n_bonds);
    r.field(PAMNET_GF_NODE_GRAPH, node_graph);
    r.field(PAMNET_GF_GPTR, gf);
    r.field(PAMNET_GF_FLAG, gf + ng + 1);
    r.field(PAMNET_GF_LOOPS, gf + ng + 2);
    const bool has_types = d.schema != PAMNET_SCHEMA_PDBBIND;
    if (has_types) r.field(PAMNET_GF_TYPES, types);
    GO(pamnet_ingest_indices_i32(d.batch, d.batch_kind, n, ng, has_types ? d.types : nullptr, has_types ? d.types_kind : 0,
                                 d.types_stride > 0 ? d.types_stride : 1, d.n_types > 0 ? d.n_types : 1, d.edge_src, d.edge_dst,
                                 d.edge_kind, d.n_bonds, r.I(node_graph), r.I(gf), r.I(types), r.I(src0), r.I(dst0),
                                 r.stream));
    const int32_t* flag = r.dry ? nullptr : r.I(gf + ng + 1);
    const int32_t* loops = r.dry ? nullptr : r.I(gf + ng + 2);
    const int32_t* gptr = r.dry ? nullptr : r.I(gf);

    // ---- positions (QM9: given; PDBbind / RNA: the first three columns of x) and the PDBbind pooling signs
    const float* pos = d.pos;
    if (d.schema != PAMNET_SCHEMA_QM9) {
        const int64_t p = r.take(3 * n);
        int64_t sg = -1;
        if (d.schema == PAMNET_SCHEMA_PDBBIND) sg = r.take(n);
        r.field(PAMNET_GF_POS, p);
        if (sg >= 0) r.field(PAMNET_GF_SIGN, sg);
        GOK(split_rows_kernel, dim3(blocks(n)), dim3(256), 0, as_stream(r.stream), d.rows, d.rows_width, n, r.F(p),
            sg >= 0 ? r.F(sg) : nullptr);
        pos = r.dry ? nullptr : r.F(p);
    }

    // ---- every array a capped fill writes: one zero-filled block, so that with sizes that turn out too large the
    // unwritten tails hold valid indices (row 0) and finite values until the deferred check raises
    const int64_t z0 = r.off;
    const int64_t g_row = r.take(eg), g_col = r.take(eg), g_dist = r.take(eg);
    const int64_t t_row = r.take(tp), t_col = r.take(tp), t_angle = r.take(tp), t_kind = r.take(tp);
    int64_t l_row = -1, l_col = -1, l_dist = -1;
    int64_t kq_g = -1, kn_l = -1, kd_l = -1, kq_l = -1, gn2 = -1, gd2 = -1;       // RNA intermediates
    if (d.schema != PAMNET_SCHEMA_QM9) l_row = r.take(el), l_col = r.take(el), l_dist = r.take(el);
    if (d.schema == PAMNET_SCHEMA_RNA) {
        kn_l = r.take(el), kd_l = r.take(el), kq_l = r.take(el);      // local cut of the kNN table by query + its query ids
        if (!d.aggregate_at_query) gn2 = r.take(eg), gd2 = r.take(eg), kq_g = r.take(eg);
    }
    // counters of the counting sorts (rows + 2 each): bond CSR / local transposition, transposed global, transposed local
    const int64_t cur_a = r.take(n + 2), cur_b = r.take(n + 2), cur_c = r.take(n + 2);
    // the transposed triplet / pair rows: written by a capped structural fill, so the tail stays valid with wrong sizes
    const int64_t tT_perm_z = grad ? r.take(tp) : -1;
    const int64_t z1 = r.off;
    if (!r.dry) {
        const hipError_t e = hipMemsetAsync(r.I(z0), 0, sizeof(int32_t) * (size_t)(z1 - z0), as_stream(r.stream));
        if (e != hipSuccess) return (int)e;
    }

    int64_t g_ptr = -1, l_ptr = -1, t_ptr_raw = -1, t_ptr = -1;
    int64_t gT_ptr = -1, gT_perm = -1, lT_ptr = -1, lT_perm = -1;
    // slot 0: global edges, 1: local edges, 2: triplet + pair rows (the bit order graph.raise_for_flag names them by)
    const int32_t* chk_ptr[3] = {nullptr, nullptr, nullptr};
    int64_t chk_val[3] = {0, 0, 0};
    auto check = [&](int slot, int64_t raw_ptr, int64_t rows, int64_t expected) {
        chk_ptr[slot] = r.dry ? nullptr : r.I(raw_ptr + rows);
        chk_val[slot] = expected;
    };
    // triplet / pair row pointer of the local graph (models.py:68-98): count -> scan -> clamp
    int64_t t_cnt = -1;                                       // triplets per bond (its pairs follow them in its rows)
    auto triplet_ptr = [&](int64_t lp, int64_t lsrc, int64_t ldst) -> int {
        const int64_t tc = r.take(el), tpc = r.take(el);
        t_cnt = tc;
        GO(pamnet_triplet_count_i32(r.I(lp), r.I(lsrc), r.I(ldst), el, wt, r.I(tc), r.I(tpc), r.stream));
        int rc = scan(r, tpc, el, t_ptr_raw);
        if (rc) return rc;
        check(2, t_ptr_raw, el, tp);
        return clamp(r, t_ptr_raw, el + 1, tp, t_ptr);
    };

    int64_t tT_ptr = -1, tT_perm = -1;
    const int64_t mol_totals = gf + ng + 3;                   // the ingest launch's four spare zeroed words
    const bool mol = d.schema == PAMNET_SCHEMA_QM9 && d.mol_local != 0;
    if (mol) {
        // molecule-local builder (graph_mol.hip): two launches write every index / geometry array of the batch
        l_row = r.take(el), l_col = r.take(el), l_dist = r.take(el);
        l_ptr = r.take(n + 1), g_ptr = r.take(n + 1), t_ptr = r.take(el + 1);
        if (grad) {
            gT_ptr = g_ptr, gT_perm = r.take(eg);
            lT_ptr = r.take(n + 1), lT_perm = r.take(el);
            tT_ptr = r.take(el + 1), tT_perm = tT_perm_z;
        }
        const int64_t mol_tot = r.take(4 * ng);
        GO(pamnet_mol_graph_count_i32(pos, gptr, n, ng, r.I(src0), r.I(dst0), d.n_bonds, d.cutoff_g, wt, r.I(mol_tot),
                                      r.I(mol_totals), r.stream));
        pamnet_mol_graph_out o{};
        if (!r.dry) {
            o.g_ptr = r.I(g_ptr), o.g_row = r.I(g_row), o.g_col = r.I(g_col), o.g_dist = r.F(g_dist);
            o.l_ptr = r.I(l_ptr), o.l_row = r.I(l_row), o.l_col = r.I(l_col), o.l_dist = r.F(l_dist);
            o.t_ptr = r.I(t_ptr), o.t_row = r.I(t_row), o.t_col = r.I(t_col), o.t_angle = r.F(t_angle), o.t_kind = r.I(t_kind);
            if (grad) {
                o.gT_perm = r.I(gT_perm), o.lT_ptr = r.I(lT_ptr), o.lT_perm = r.I(lT_perm);
                o.tT_ptr = r.I(tT_ptr), o.tT_perm = r.I(tT_perm);
            }
        }
        GO(pamnet_mol_graph_fill_i32(pos, gptr, n, ng, r.I(src0), r.I(dst0), d.n_bonds, d.cutoff_g, wt, grad ? 1 : 0,
                                     r.I(mol_tot), eg, tp, &o, r.stream));
        // device-side totals against the host's sizes; a molecule outside the builder's limits counts no bonds
        chk_ptr[0] = r.dry ? nullptr : r.I(mol_totals), chk_val[0] = eg;
        chk_ptr[1] = r.dry ? nullptr : r.I(mol_totals + 3), chk_val[1] = el;
        chk_ptr[2] = r.dry ? nullptr : r.I(mol_totals + 1), chk_val[2] = tp;
    } else if (d.schema == PAMNET_SCHEMA_QM9) {
        // bond list in CSR order of its targets + bond lengths (j, i = edge_index; models.py:64-65); no self loops assumed
        // (the ingest launch noted them; a store strips them at ingestion)
        l_row = r.take(el), l_col = r.take(el), l_dist = r.take(el);
        int64_t perm = -1;
        int rc = csr(r, dst0, d.n_bonds, n, cur_a, l_ptr, perm);
        if (rc) return rc;
        GO(pamnet_gather2_i32(r.I(perm), r.I(src0), r.I(dst0), d.n_bonds, r.I(l_col), r.I(l_row), pos, r.F(l_dist), r.stream));
        if ((rc = triplet_ptr(l_ptr, l_col, l_row))) return rc;
        // global graph: radius search at cutoff_g (models.py:110), symmetric, rows = query
        const int64_t cnt = r.take(n);
        int64_t raw = -1;
        GO(pamnet_radius_count_i32(pos, r.I(node_graph), gptr, n, ng, d.cutoff_g, d.max_neighbors, r.I(cnt), const_cast<int32_t*>(flag), r.stream));
        if ((rc = scan(r, cnt, n, raw))) return rc;
        check(0, raw, n, eg);
        check(1, l_ptr, n, el);
        if ((rc = clamp(r, raw, n + 1, eg, g_ptr))) return rc;
        GO(pamnet_radius_fill_i32(pos, r.I(node_graph), gptr, n, ng, d.cutoff_g, d.max_neighbors, r.I(g_ptr), r.I(g_col), r.F(g_dist),
                                  r.I(g_row), eg, r.stream));
    } else if (d.schema == PAMNET_SCHEMA_PDBBIND) {
        // global = radius graph at cutoff_g, local = the same at cutoff_l (= global edges with dist <= cutoff_l,
        // models.py:128-134): both degree sequences come from count passes over the positions
        const int64_t cg = r.take(n), cl = r.take(n);
        int64_t raw_g = -1, raw_l = -1;
        GO(pamnet_radius_count_i32(pos, r.I(node_graph), gptr, n, ng, d.cutoff_g, d.max_neighbors, r.I(cg), const_cast<int32_t*>(flag), r.stream));
        int rc = scan(r, cg, n, raw_g);
        if (rc) return rc;
        GO(pamnet_radius_count_i32(pos, r.I(node_graph), gptr, n, ng, d.cutoff_l, 0, r.I(cl), nullptr, r.stream));
        if ((rc = scan(r, cl, n, raw_l))) return rc;
        check(0, raw_g, n, eg);
        check(1, raw_l, n, el);
        if ((rc = clamp(r, raw_g, n + 1, eg, g_ptr))) return rc;
        if ((rc = clamp(r, raw_l, n + 1, el, l_ptr))) return rc;
        GO(pamnet_radius_fill_i32(pos, r.I(node_graph), gptr, n, ng, d.cutoff_g, d.max_neighbors, r.I(g_ptr), r.I(g_col), r.F(g_dist),
                                  r.I(g_row), eg, r.stream));
        GO(pamnet_csr_filter_fill_i32(r.I(g_ptr), r.I(g_col), r.F(g_dist), n, d.cutoff_l, r.I(l_ptr), r.I(l_col), r.F(l_dist),
                                      el, r.stream));
        GO(pamnet_expand_rows_i32(r.I(l_ptr), n, r.I(l_row), el, r.stream));
        if ((rc = triplet_ptr(l_ptr, l_col, l_row))) return rc;
    } else {
        // kNN table by query (self dropped), cut at cutoff_g / cutoff_l (models.py:143-156)
        const int64_t k = d.knn_k;
        const int64_t kn = r.take(n * k), kd = r.take(n * k);
        const int64_t ca = r.take(n), cb = r.take(n);
        // one search, the per-query sizes of both cuts counted on the way (rows are ascending in distance)
        GO(pamnet_knn_cut_i32(pos, r.I(node_graph), gptr, n, (int32_t)k, d.cutoff_g, d.cutoff_l, r.I(kn), r.F(kd), r.I(ca), r.I(cb),
                              r.stream));
        const int64_t raw_a = r.take(n + 1), raw_b = r.take(n + 1), pa = r.take(n + 1), pb = r.take(n + 1);
        const int64_t stmp = r.take((n + 4095) / 4096 + 1);
        GO(pamnet_exclusive_scan_pair_i32(r.I(ca), r.I(raw_a), r.I(cb), r.I(raw_b), n, r.I(stmp), r.stream));
        check(0, raw_a, n, eg);
        check(1, raw_b, n, el);
        // stored by query: (pa, gq_n, gq_d, query ids) and (pb, kn_l, kd_l, kq_l), both cuts in one pass
        const int64_t gq_n = d.aggregate_at_query ? g_col : gn2, gq_d = d.aggregate_at_query ? g_dist : gd2;
        const int64_t gq_q = d.aggregate_at_query ? g_row : kq_g;
        GO(pamnet_knn_cut_fill_i32(r.I(kn), r.F(kd), n, (int32_t)k, d.cutoff_g, r.I(raw_a), eg, r.I(gq_n), r.F(gq_d), r.I(gq_q),
                                   r.I(pa), d.cutoff_l, r.I(raw_b), el, r.I(kn_l), r.F(kd_l), r.I(kq_l), r.I(pb), r.stream));
        int rc = PAMNET_OK;
        if (d.aggregate_at_query) {                   // flow = target_to_source: the global layer aggregates at the query
            g_ptr = pa;
        } else {                                      // aggregate at the neighbour: re-store the list by neighbour
            int64_t perm = -1;
            if ((rc = csr(r, gq_n, eg, n, cur_b, g_ptr, perm))) return rc;
            int64_t inv = -1;
            if (grad) inv = r.take(eg);
            GO(pamnet_transpose_gather_i32(r.I(perm), r.I(kq_g), r.F(gq_d), eg, r.I(g_col), r.F(g_dist),
                                           grad ? r.I(inv) : nullptr, r.stream));
            if (grad) gT_ptr = pa, gT_perm = inv;     // the inverse transposition: rows = queries
        }
        if (!d.aggregate_at_query) {
            GO(pamnet_expand_rows_i32(r.I(g_ptr), n, r.I(g_row), eg, r.stream));
        }
        // the local layer always aggregates at the neighbour (models.py:153-156: j = query, i = neighbour)
        int64_t perm = -1;
        if ((rc = csr(r, kn_l, el, n, cur_a, l_ptr, perm))) return rc;
        int64_t inv = -1;
        if (grad) inv = r.take(el);
        GO(pamnet_transpose_gather_i32(r.I(perm), r.I(kq_l), r.F(kd_l), el, r.I(l_col), r.F(l_dist), grad ? r.I(inv) : nullptr,
                                       r.stream));
        if (grad) lT_ptr = pb, lT_perm = inv;
        GO(pamnet_expand_rows_i32(r.I(l_ptr), n, r.I(l_row), el, r.stream));
        if ((rc = triplet_ptr(l_ptr, l_col, l_row))) return rc;
    }

    // ---- triplets / pairs + angles, rows grouped by target edge (models.py:68-98, 165-177)
    if (!mol) {
        GO(pamnet_triplet_fill_f32(pos, r.I(l_ptr), r.I(l_col), r.I(l_row), el, wt, r.I(t_ptr), r.I(t_col), r.I(t_row),
                                   r.F(t_angle), r.I(t_kind), tp, r.stream));
    }
    // ---- the sizes the host assumed against the device-side counts + the input-validity flag: one launch
    GO(pamnet_check_sizes_i32(3, chk_ptr, chk_val, nullptr, d.schema == PAMNET_SCHEMA_QM9 ? loops : nullptr,
                              const_cast<int32_t*>(flag), r.stream));

    // ---- index structures of the backward gathers (transposed CSRs)
    if (grad && !mol) {
        if (d.schema != PAMNET_SCHEMA_RNA) {          // radius graph: its own pointer + the reverse-edge index
            gT_ptr = g_ptr;
            gT_perm = r.take(eg);
            GO(pamnet_reverse_edges_i32(r.I(g_ptr), r.I(g_row), r.I(g_col), eg, r.I(gT_perm), nullptr, r.stream));
        } else if (d.aggregate_at_query) {            // kNN list stored by query: counting sort of the neighbour column
            int rc = csr(r, g_col, eg, n, cur_b, gT_ptr, gT_perm);
            if (rc) return rc;
        }
        if (d.schema == PAMNET_SCHEMA_PDBBIND) {
            lT_ptr = l_ptr;
            lT_perm = r.take(el);
            GO(pamnet_reverse_edges_i32(r.I(l_ptr), r.I(l_row), r.I(l_col), el, r.I(lT_perm), nullptr, r.stream));
        } else if (d.schema == PAMNET_SCHEMA_QM9) {   // user-supplied bonds: counting sort of the source column
            int rc = csr(r, l_col, el, n, cur_c, lT_ptr, lT_perm);
            if (rc) return rc;
        }
        // transposed triplet / pair rows from the structure of the local graph (no sort over the T + P rows)
        const int64_t cntT = r.take(el);
        GO(pamnet_triplet_transpose_count_i32(r.I(l_ptr), r.I(l_col), r.I(l_row), r.I(lT_ptr), r.I(lT_perm), el, wt, r.I(cntT),
                                              r.stream));
        int64_t rawT = -1;
        int rc = scan(r, cntT, el, rawT);
        if (rc) return rc;
        if ((rc = clamp(r, rawT, el + 1, tp, tT_ptr))) return rc;
        tT_perm = tT_perm_z;
        GO(pamnet_triplet_transpose_fill_i32(r.I(l_ptr), r.I(l_col), r.I(l_row), r.I(lT_ptr), r.I(lT_perm), el, wt, r.I(t_ptr),
                                             r.I(t_cnt), r.I(tT_ptr), r.I(tT_perm), tp, r.stream));
    }

    // the local aggregation's backward gathers through tT_perm -> t_row -> l_row: both hops once per graph
    // (need_grad == 2: the consumer is the narrow-width engine, which reads neither -- no launch, no 2 tp ints)
    int64_t tT_edge = -1, tT_node = -1;
    if (d.need_grad == 1 && tT_perm >= 0 && tp > 0) {
        tT_edge = r.take(tp), tT_node = r.take(tp);
        GO(pamnet_triplet_transpose_aux_i32(r.I(tT_perm), r.I(t_row), r.I(l_row), tp, r.I(tT_edge), r.I(tT_node), r.stream));
    }

    // ---- spherical basis on the combined rows (layers/basic.py:107-116)
    const int64_t rad = r.take(el * RAD_W);
    if (r.sbf) {
        GO(pamnet_sbf_radial_f32(r.F(l_dist), d.cutoff_l, el, r.F(rad), r.stream));
        GO(pamnet_sbf_combine_f32(r.F(rad), r.I(t_col), r.F(t_angle), tp, r.sbf, r.stream));
    }
    // ---- node-aligned work split of the fused global-edge kernels (csrc/edge_agg.hip), once per graph
    const int64_t cuts = r.take(260);
    GO(pamnet_seg_cuts_i32(r.I(g_ptr), r.I(g_row), n, eg, r.I(cuts), nullptr, r.stream));

    r.field(PAMNET_GF_G_PTR, g_ptr), r.field(PAMNET_GF_G_ROW, g_row), r.field(PAMNET_GF_G_COL, g_col);
    r.field(PAMNET_GF_G_DIST, g_dist), r.field(PAMNET_GF_GT_PTR, gT_ptr), r.field(PAMNET_GF_GT_PERM, gT_perm);
    r.field(PAMNET_GF_L_PTR, l_ptr), r.field(PAMNET_GF_L_ROW, l_row), r.field(PAMNET_GF_L_COL, l_col);
    r.field(PAMNET_GF_L_DIST, l_dist), r.field(PAMNET_GF_LT_PTR, lT_ptr), r.field(PAMNET_GF_LT_PERM, lT_perm);
    r.field(PAMNET_GF_T_PTR, t_ptr), r.field(PAMNET_GF_T_ROW, t_row), r.field(PAMNET_GF_T_COL, t_col);
    r.field(PAMNET_GF_T_ANGLE, t_angle), r.field(PAMNET_GF_T_KIND, t_kind);
    r.field(PAMNET_GF_TT_PTR, tT_ptr), r.field(PAMNET_GF_TT_PERM, tT_perm);
    r.field(PAMNET_GF_CUTS, cuts);
    r.field(PAMNET_GF_TT_EDGE, tT_edge), r.field(PAMNET_GF_TT_NODE, tT_node);
    return PAMNET_OK;
}

int validate(const pamnet_graph_desc* d) {
    if (!d) return PAMNET_ENULL;
    if (d->schema < PAMNET_SCHEMA_QM9 || d->schema > PAMNET_SCHEMA_RNA) return PAMNET_EINVAL;
    // the straight-line form has no empty-list cases: such batches take the step-by-step entry points
    if (d->n < 1 || d->n_graphs < 1 || d->eg < 1 || d->el < 1 || d->tp < 1 || d->n_bonds < 0) return PAMNET_EINVAL;
    if (d->n >= ((int64_t)1 << 31) / 64 || d->eg >= (int64_t)1 << 31 || d->tp >= (int64_t)1 << 31) return PAMNET_EINVAL;
    if (d->schema == PAMNET_SCHEMA_QM9 && (d->n_bonds != d->el || d->n_types < 1)) return PAMNET_EINVAL;
    if (d->schema != PAMNET_SCHEMA_QM9 && (d->n_bonds != 0 || d->rows_width < 3)) return PAMNET_EINVAL;
    if (d->schema == PAMNET_SCHEMA_PDBBIND && !(d->cutoff_l <= d->cutoff_g)) return PAMNET_EINVAL;
    if (d->schema == PAMNET_SCHEMA_RNA && (d->knn_k < 1 || d->knn_k > 64 || d->n_types < 1)) return PAMNET_EINVAL;
    return PAMNET_OK;
}

}  // namespace

extern "C" int pamnet_graph_plan(const pamnet_graph_desc* desc, int64_t* layout, int64_t* arena_ints) {
    const int rc = validate(desc);
    if (rc) return rc;
    if (!layout || !arena_ints) return PAMNET_ENULL;
    Run r{*desc, nullptr, nullptr, nullptr, true, 0, layout};
    const int rr = run(r);
    *arena_ints = r.off + 64;
    return rr;
}

extern "C" int pamnet_graph_build_i32(const pamnet_graph_desc* desc, int32_t* arena, float* sbf, pamnet_stream_t stream) {
    const int rc = validate(desc);
    if (rc) return rc;
    if (!arena || !desc->batch) return PAMNET_ENULL;
    if (desc->schema == PAMNET_SCHEMA_QM9 && (!desc->pos || !desc->types || !desc->edge_src || !desc->edge_dst)) return PAMNET_ENULL;
    if (desc->schema != PAMNET_SCHEMA_QM9 && !desc->rows) return PAMNET_ENULL;
    if (desc->schema == PAMNET_SCHEMA_RNA && !desc->types) return PAMNET_ENULL;
    int64_t layout[PAMNET_GRAPH_FIELDS];
    Run r{*desc, arena, sbf, stream, false, 0, layout};
    return run(r);
}
