/*
 * pamnet_hip.h -- C ABI of libpamnet_hip.so: PAMNet's multiplex message-passing hot path as gfx950 (MI355X) kernels.
 *
 * The reference (XieResearchGroup/Physics-aware-Multiplex-GNN) has no FFI of its own: its hot path reaches native code
 * through four third-party wheels (torch_scatter, torch_sparse, torch_cluster, torch_geometric) and torch ATen.  Each
 * entry point below replaces one of those native call sites; the reference file:line it stands in for is cited.
 *
 * Conventions (every function):
 *   - arguments are raw DEVICE pointers, 64-bit sizes and a hipStream_t (passed as void*); no torch types;
 *   - the caller owns every device buffer; the library allocates no device memory, keeps no mutable state between calls
 *     and never synchronises (the pamnet_stack_* engine calls build small host-side pointer tables on the stack / heap per
 *     call).  The only process-wide data are five read-once developer switches taken from the environment on first use
 *     (PAMNET_EDGE_WAVES, PAMNET_EDGE_IMAGES, PAMNET_CHAIN_BF16, PAMNET_SMALL_FORMS, PAMNET_AGG_PIECES, PAMNET_CHAIN_LEAN: kernel-variant selection for measurements; C++11 static
 *     initialisation, thread-safe, constant afterwards);
 *   - work is enqueued on `stream`; return value 0 = OK, >0 = hipError_t of the failed launch, <0 = argument error
 *     (PAMNET_EINVAL: bad size / unsupported width; PAMNET_ENULL: required pointer is null);
 *   - float tensors are fp32 row-major [rows, d]; index tensors are int32; CSR pointers have rows+1 entries;
 *   - re-entrant and thread-safe: nothing mutable is shared between calls.  Segment reductions are sorted-CSR, atomics-free, deterministic.
 */
#ifndef PAMNET_HIP_H
#define PAMNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PAMNET_OK 0
#define PAMNET_EINVAL (-1)
#define PAMNET_ENULL (-2)
/* flag bit of pamnet_local_bwd_pair_f32's accumulate_dx (bit 0 = accumulate): W1 / W2 are fragment images, see
 * pamnet_pack_weights_mixed_f32 */
#define PAMNET_WEIGHT_IMAGES 2
/* flag bit of `nblk` of pamnet_node_pre_tail_bwd(_gather)_f32: every weight image of the call is a bf16x3 (kind-1, transposed)
 * one, and the launch runs its GEMMs as bf16x6 piece products (single-round batches; pamnet_node_tail_main_bwd_f32 takes
 * packed == 2 for the same) */
#define PAMNET_CHAIN_PIECES 16

typedef void* pamnet_stream_t; /* hipStream_t */

/* Library / ABI version (bumped on any signature change).  pamnet_abi_version() returns the PAMNET_ABI_VERSION the library
 * was built against; a binding compares it with this header's (pamnet_amd/lib.py load(): a stale .so fails loudly). */
#define PAMNET_ABI_VERSION 14
int pamnet_abi_version(void);

/* ------------------------------------------------------------------------------------------------------------------
 * Segment reduction / gather  (torch_scatter.scatter(src, index, dim=0, dim_size, reduce='add'):
 *   layers/local_message_passing.py:50,54,107,111;  PyG MessagePassing aggregate: layers/global_message_passing.py:38;
 *   global_add_pool / global_mean_pool: models.py:216,219,221,351;  row gathers x[i], m[idx]: local...:46,49)
 *
 * out[r, :] = (init ? init[r, :] : 0) + sum_{q = ptr[r] .. ptr[r+1]-1}  A[ia(k), :] * (B ? B[ib(k), :] : 1),
 *             k = perm ? perm[q] : q,   ia(k) = ia ? ia[k] : k,   ib(k) = ib ? ib[k] : k.
 * With ia=ib=perm=B=init=NULL this is exactly scatter(src=A, index=sorted segment ids) -- the scatter-add roofline
 * kernel.  The same entry serves every backward (gather^T) through a transposed CSR (`perm`).
 * d must be a multiple of 4.  `out` may not alias A or B.
 * ------------------------------------------------------------------------------------------------------------------ */
int pamnet_segment_sum_f32(float* out, const float* init, const float* A, const int32_t* ia, const float* B,
                           const int32_t* ib, const int32_t* perm, const int32_t* ptr, int64_t rows, int64_t d,
                           pamnet_stream_t stream);

/* out[k, :] = A[ia ? ia[k] : k, :] * (B ? B[ib ? ib[k] : k, :] : 1)   for k < m.   (x[i], x[j], m_neighbor[idx]) */
int pamnet_gather_mul_f32(float* out, const float* A, const int32_t* ia, const float* B, const int32_t* ib,
                          int64_t m, int64_t d, pamnet_stream_t stream);

/* Batched forms used by the layer backward (d = 128): up to 4 plain segment sums over the same number of rows in one
 * launch (host arrays of device pointers; perm[j] nullable), and one gather feeding two products
 * (out1 = A[ia]*B1, out2 = A[ia]*B2: d m_t and d q3 of layers/local_message_passing.py:53). */
int pamnet_segment_sum_multi_f32(int64_t njobs, float* const* out, const float* const* A, const int32_t* const* perm,
                                 const int32_t* const* ptr, int64_t rows, int64_t d, pamnet_stream_t stream);
int pamnet_gather_mul2_f32(float* out1, float* out2, const float* A, const int32_t* ia, const float* B1,
                           const float* B2, int64_t m, int64_t d, pamnet_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Index plumbing for graph construction (torch_sparse.SparseTensor CSR build: models.py:71-73, 267-269)
 * ------------------------------------------------------------------------------------------------------------------ */
/* out[0]=0, out[i+1]=sum_{j<=i} in[j]  (n inputs -> n+1 outputs).  `tmp` needs ceil(n/4096)+1 ints. */
int pamnet_exclusive_scan_i32(const int32_t* in, int32_t* out, int64_t n, int32_t* tmp, pamnet_stream_t stream);
/* two arrays of the same length, one launch when they are short */
int pamnet_exclusive_scan_pair_i32(const int32_t* in_a, int32_t* out_a, const int32_t* in_b, int32_t* out_b, int64_t n,
                                   int32_t* tmp, pamnet_stream_t stream);

/* flag[0] = 1 when the index inputs of a batch are out of range (the reference would raise an IndexError): node_graph not
 * sorted / not in [0, n_graphs), a type (float, element i at types[i * type_stride]; nullable) not in [0, n_types), an
 * edge endpoint (src / dst, nullable with n_edges = 0) not in [0, n).  One launch; flag is zeroed by the call. */
int pamnet_validate_inputs_i32(const int32_t* node_graph, int64_t n, int64_t n_graphs, const float* types,
                               int64_t type_stride, int64_t n_types, const int32_t* src, const int32_t* dst,
                               int64_t n_edges, int32_t* flag, pamnet_stream_t stream);
/* Stable counting sort of m keys in [0, rows): ptr[rows+1] (CSR) and perm[m] with keys[perm[q]] non-decreasing and
 * perm ascending inside a row.  Scratch: `cursor` rows + 1 ints (the rows' counters plus one flag: an already
 * non-decreasing key sequence takes the identity-permutation path), `perm_tmp` m ints, `tmp` ceil(rows/4096)+1 ints (the
 * scan's chunk sums).  Deterministic. */
int pamnet_csr_from_keys_i32(const int32_t* keys, int64_t m, int64_t rows, int32_t* ptr, int32_t* perm,
                             int32_t* cursor, int32_t* perm_tmp, int32_t* tmp, pamnet_stream_t stream);
/* The same with `cursor` already zero-filled by the caller (no fill launch of its own). */
int pamnet_csr_from_keys_z_i32(const int32_t* keys, int64_t m, int64_t rows, int32_t* ptr, int32_t* perm,
                               int32_t* cursor, int32_t* perm_tmp, int32_t* tmp, pamnet_stream_t stream);

/* row_of[q] = r for q in [ptr[r], ptr[r+1])  (repeat_interleave of row ids, models.py:76-77, 88-89).
 * `cap` = entries the output holds (the *_fill entry points take one too): nothing is written at or beyond it.  With sizes
 * read back from the device cap = ptr[rows] and never binds; it makes the zero-host-sync path (sizes assumed by the host,
 * verified later: pamnet_check_sizes_i32) memory-safe when the assumption is wrong. */
int pamnet_expand_rows_i32(const int32_t* ptr, int64_t rows, int32_t* row_of, int64_t cap, pamnet_stream_t stream);

/* CSR row filter: keep entries with nbr >= 0 and dist <= cut  (the cutoff masks, models.py:131-134, 147-156).
 * count -> (caller scans) -> fill. */
int pamnet_csr_filter_count_i32(const int32_t* ptr_in, const int32_t* nbr, const float* dist, int64_t rows, float cut,
                                int32_t* count, pamnet_stream_t stream);
int pamnet_csr_filter_fill_i32(const int32_t* ptr_in, const int32_t* nbr, const float* dist, int64_t rows, float cut,
                               const int32_t* ptr_out, int32_t* nbr_out, float* dist_out, int64_t cap,
                               pamnet_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Neighbour search (torch_cluster.radius / knn: models.py:110,128,143,301), self loops removed (models.py:63).
 * `gptr[b..b+1]` = node range of graph b (batch sorted).  Two-pass: count -> (caller scans) -> fill.
 * radius: neighbours j != i of the same graph with ||pos_i - pos_j|| <= r, ascending j.  Symmetric by construction --
 * unless max_neighbors binds.  max_neighbors (0 = unlimited; models.py:110,128 pass 1000, :301 passes 500): a query keeps
 * the first max_neighbors points of its graph within r in ascending index order, itself included in that count (the search
 * returns the query; remove_self_loops follows, models.py:63); the count pass ORs 64 into *cap_flag (nullable) when a row
 * was truncated -- the graph is then not symmetric and the caller must build general transposes.
 * n_graphs (0 = unknown) only selects the launch shape: one thread per node for molecule-sized graphs, one wavefront per
 * node from ~100 nodes per graph on; the output is the same.
 * ------------------------------------------------------------------------------------------------------------------ */
int pamnet_radius_count_i32(const float* pos, const int32_t* node_graph, const int32_t* gptr, int64_t n,
                            int64_t n_graphs, float r, int64_t max_neighbors, int32_t* count, int32_t* cap_flag,
                            pamnet_stream_t stream);
int pamnet_radius_fill_i32(const float* pos, const int32_t* node_graph, const int32_t* gptr, int64_t n,
                           int64_t n_graphs, float r, int64_t max_neighbors, const int32_t* ptr, int32_t* nbr, float* dist,
                           int32_t* row_of /* nullable: the query node of every entry */, int64_t cap,
                           pamnet_stream_t stream);

/* knn: for every query node its k nearest nodes of the same graph (itself included, as torch_cluster.knn does),
 * ordered by (distance, index); then the self entry is dropped and entries with dist > cutoff are masked out:
 * nbr[i*k + s] = neighbour index or -1, dist[i*k + s] = distance.  (models.py:143-150) */
int pamnet_knn_i32(const float* pos, const int32_t* node_graph, const int32_t* gptr, int64_t n, int32_t k,
                   float cutoff, int32_t* nbr, float* dist, pamnet_stream_t stream);
/* The kNN table (no cutoff: every entry but the self entry kept) together with what its two cuts keep per query -- cnt_a[i] /
 * cnt_b[i] = entries of row i with dist <= cut_a / cut_b (models.py:147-156: the global and the local graph of the RNA path
 * are cuts of one kNN search) -- and, after the caller's scans (pamnet_exclusive_scan_pair_i32), both cut lists written in
 * one pass: kept entries of row i at raw[i] + rank with the query id beside them (capped at cap), pointers clamped to cap
 * into ptr_a / ptr_b [n + 1].  The same arrays as pamnet_csr_filter_count/fill_i32 + pamnet_expand_rows_i32 per cut. */
int pamnet_knn_cut_i32(const float* pos, const int32_t* node_graph, const int32_t* gptr, int64_t n, int32_t k, float cut_a,
                       float cut_b, int32_t* nbr, float* dist, int32_t* cnt_a, int32_t* cnt_b, pamnet_stream_t stream);
/* total[0] (int64, zero on entry) = triplet + pair rows (models.py:68-98) of the graph the entries within `cut` of a kNN table
 * define (query -> neighbour, aggregated at the neighbour), before that graph is built: the kNN path's sizes in ONE host
 * read-back.  indeg: [n] int32 scratch, zero on entry. */
int pamnet_knn_tp_total_i64(const int32_t* nbr, const float* dist, int64_t n, int32_t k, float cut, int32_t with_triplets,
                            int32_t* indeg, int64_t* total, pamnet_stream_t stream);
int pamnet_knn_cut_fill_i32(const int32_t* nbr, const float* dist, int64_t n, int32_t k, float cut_a, const int32_t* raw_a,
                            int64_t cap_a, int32_t* nbr_a, float* dist_a, int32_t* row_a, int32_t* ptr_a, float cut_b,
                            const int32_t* raw_b, int64_t cap_b, int32_t* nbr_b, float* dist_b, int32_t* row_b, int32_t* ptr_b,
                            pamnet_stream_t stream);

/* Edge distances  dist[e] = ||pos[a[e]] - pos[b[e]]||   (PAMNet.get_edge_info, models.py:62-66) */
int pamnet_edge_dist_f32(const float* pos, const int32_t* a, const int32_t* b, int64_t m, float* dist,
                         pamnet_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Triplet / pair enumeration + angles  (PAMNet.indices, models.py:68-98, 263-281; angles models.py:165-177)
 * Local edges e = (src[e] -> dst[e]) are stored CSR by dst (lptr[n+1]).  For every edge e=(j->i) the combined row list
 * [tp_ptr[e], tp_ptr[e+1]) holds
 *   its triplets: every e'=(k->j), k != i       kind 0, tp_idx = e' (= idx_kj), tp_edge = e (= idx_ji), angle2
 *   its pairs:    every e'=(j'->i) incl. e'=e    kind 1, tp_idx = e' (= idx_jj_pair), tp_edge = e (= idx_ji_pair), angle1
 * i.e. the reference's cat(idx_kj, idx_jj_pair) / cat(idx_ji, idx_ji_pair) (local_message_passing.py:38-39) regrouped
 * by target edge.  with_triplets = 0 enumerates pairs only (PAMNet_s).  count -> (caller scans tpcount) -> fill.
 * ------------------------------------------------------------------------------------------------------------------ */
int pamnet_triplet_count_i32(const int32_t* lptr, const int32_t* src, const int32_t* dst, int64_t n_edges,
                             int32_t with_triplets, int32_t* tcount, int32_t* tpcount, pamnet_stream_t stream);
int pamnet_triplet_fill_f32(const float* pos, const int32_t* lptr, const int32_t* src, const int32_t* dst,
                            int64_t n_edges, int32_t with_triplets, const int32_t* tp_ptr, int32_t* tp_idx,
                            int32_t* tp_edge, float* tp_angle, int32_t* tp_kind, int64_t cap, pamnet_stream_t stream);
/* Transposed triplet / pair row list (for every source bond the rows that gather it, ascending: the (ptr, perm) that
 * pamnet_csr_from_keys_i32 returns for keys = tp_idx over n_edges rows) from the graph's structure instead of a counting sort
 * over the rows: count -> caller scans -> fill.  lt_ptr [n + 1] / lt_perm [n_edges]: the transposed bond list (bonds by source
 * atom, any order inside a row); tp_ptr / tcount as pamnet_triplet_count_i32 / the scan produced them; tt_ptr = the scanned
 * counts; cap = entries tt_perm holds.  No self loops. */
int pamnet_triplet_transpose_count_i32(const int32_t* lptr, const int32_t* src, const int32_t* dst, const int32_t* lt_ptr,
                                       const int32_t* lt_perm, int64_t n_edges, int32_t with_triplets, int32_t* count,
                                       pamnet_stream_t stream);
int pamnet_triplet_transpose_fill_i32(const int32_t* lptr, const int32_t* src, const int32_t* dst, const int32_t* lt_ptr,
                                      const int32_t* lt_perm, int64_t n_edges, int32_t with_triplets, const int32_t* tp_ptr,
                                      const int32_t* tcount, const int32_t* tt_ptr, int32_t* tt_perm, int64_t cap,
                                      pamnet_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Zero-host-sync graph construction (models.py:62-98,104-157 read their data-dependent sizes back to the
 * host through boolean masks / repeat_interleave).
 * check_sizes: up to 4 device-side totals (actual[k][0], e.g. the last element of a CSR pointer) against the values the
 *   host assumed; bit (2 << k) of flag[0] is set on a mismatch, bit 32 when *all_kept (a device bool, nullable) is false
 *   or *self_loops (int32, nullable) is non-zero.
 *   flag is NOT zeroed (pamnet_validate_inputs_i32 owns bit 1 of the same word).
 * collate: batch of n_graphs graphs sel[k] of a dataset kept resident as concatenated arrays (prefix sums src_nptr /
 *   src_eptr, bonds with graph-local endpoints): node features [n_out, x_width], positions (nullable), int32 batch
 *   vector, bonds with batch-level endpoints, the graphs' targets.  out_nptr / out_eptr: the batch's prefix sums (device,
 *   n_graphs + 1).
 * ------------------------------------------------------------------------------------------------------------------ */
/* Transposed CSR of a SYMMETRIC graph stored by target with ascending columns (radius graphs): rev[e] = position of the
 * reverse edge of e; (ptr, rev) equals what pamnet_csr_from_keys_i32(col) returns as (ptr, perm).  flag (nullable):
 * bit 64 is set if an edge has no reverse. */
int pamnet_reverse_edges_i32(const int32_t* ptr, const int32_t* row_of, const int32_t* col, int64_t m, int32_t* rev,
                             int32_t* flag, pamnet_stream_t stream);
/* out[k] (device int64) = *src[k], read as int32 (kind 0), bool / uint8 (1) or int64 (2); n <= 8.  The data-dependent sizes
 * of a batch in one launch, ahead of the one device->host copy graph construction makes. */
int pamnet_gather_scalars_i64(int64_t n, const void* const* src /* host array of device ptrs */,
                              const int32_t* kind /* host */, int64_t* out, pamnet_stream_t stream);
int pamnet_check_sizes_i32(int64_t n_checks, const int32_t* const* actual /* host array of device ptrs */,
                           const int64_t* expected /* host */, const void* all_kept, const int32_t* self_loops,
                           int32_t* flag, pamnet_stream_t stream);
/* The reference's index tensors (`x`, `edge_index`, `batch` of models.py:104-110; `j, i = edge_index` models.py:64) in
 * ONE launch: each is read as int64 (kind 1, what PyG hands over), int32 (2) or fp32 (3; kind 0 = no `x`) and written as
 * int32 -- node_graph [n], types [n] (element i read at x[i * x_stride]), src / dst [n_edges].  gptr_flag holds
 * n_graphs + 7 ints, zeroed by the call (the last four are spare zeroed words for the caller: the molecule-local builder's
 * batch totals live there), gptr_flag[0 .. n_graphs] = first node of every graph (the CSR pointer of the
 * sorted batch vector), gptr_flag[n_graphs + 1] = 1 when an index is out of range (batch unsorted / not in [0, n_graphs),
 * type not in [0, n_types), edge endpoint not in [0, n): the reference raises IndexError; offending entries are written
 * as 0 so that later kernels stay in bounds), gptr_flag[n_graphs + 2] = 1 when the edge list has a self loop
 * (remove_self_loops, models.py:63, would then not be a no-op). */
int pamnet_ingest_indices_i32(const void* batch, int32_t batch_kind, int64_t n, int64_t n_graphs, const void* x,
                              int32_t x_kind, int64_t x_stride, int64_t n_types, const void* edge_src,
                              const void* edge_dst, int32_t edge_kind, int64_t n_edges, int32_t* node_graph,
                              int32_t* gptr_flag, int32_t* types, int32_t* src, int32_t* dst, pamnet_stream_t stream);
/* out_a[q] = a[perm[q]], out_b[q] = b[perm[q]]: the bond list in CSR order of its targets (models.py:71-73); with
 * dist (nullable; then pos [n,3] is required) also dist[q] = ||pos[out_b[q]] - pos[out_a[q]]|| (models.py:65). */
int pamnet_gather2_i32(const int32_t* perm, const int32_t* a, const int32_t* b, int64_t m, int32_t* out_a,
                       int32_t* out_b, const float* pos, float* dist, pamnet_stream_t stream);
/* An edge list stored by query node, re-stored by neighbour (models.py:147-156 aggregate the kNN edges at the neighbour):
 * with perm = the stable counting sort of the neighbour column, out_q[e'] = q[perm[e']], out_dist[e'] = dist[perm[e']],
 * and inv[perm[e']] = e' (nullable) -- the positions of the query-ordered edges in the new list, i.e. together with the
 * query-ordered CSR pointer the transposed CSR the backward gathers d x[j] with. */
int pamnet_transpose_gather_i32(const int32_t* perm, const int32_t* q, const float* dist, int64_t m, int32_t* out_q,
                                float* out_dist, int32_t* inv, pamnet_stream_t stream);
int pamnet_collate_f32(int64_t n_graphs, const int32_t* sel, const int32_t* out_nptr, const int32_t* out_eptr,
                       const int32_t* src_nptr, const int32_t* src_eptr, const float* x, int64_t x_width,
                       const float* pos, const int32_t* esrc, const int32_t* edst, int64_t n_out, int64_t e_out,
                       float* out_x, float* out_pos, int32_t* out_batch, int32_t* out_esrc, int32_t* out_edst,
                       const float* y /* [graphs of the dataset] targets, nullable */, float* out_y /* [n_graphs] */,
                       pamnet_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Graph-construction engine: everything of PAMNet.forward that does not depend on the parameters -- index ingestion,
 * remove_self_loops, radius / knn graphs, cutoff masks, the SparseTensor CSRs, triplets / pairs and their angles
 * (models.py:62-98, 104-177), the transposed index lists of the backward and the spherical basis
 * (layers/basic.py:107-116) -- enqueued by ONE call (csrc/graph_engine.hip) for a batch whose data-dependent sizes the host
 * already knows (eg, el, tp: per-graph constants, pamnet_amd/store.py).  Same kernels, same order, same results as the
 * step-by-step entry points above; fills are capped by the sizes, and the device-side counts are compared with them by
 * pamnet_check_sizes_i32 into the flag word (field PAMNET_GF_FLAG), to be read whenever the host next synchronises.
 * `arena`: caller-owned int32 buffer of *arena_ints entries (pamnet_graph_plan); layout[f] = offset (in ints) of field f
 * inside it, or -1 when the batch has no such array (float fields are stored in place: reinterpret).  Batches with an
 * empty node / edge / row list are rejected (PAMNET_EINVAL): they take the step-by-step entry points.
 * ------------------------------------------------------------------------------------------------------------------ */
#define PAMNET_SCHEMA_QM9 0     /* x = atom types, pos, bond list (models.py:104-113) */
#define PAMNET_SCHEMA_PDBBIND 1 /* rows = xyz + features; radius graphs at both cutoffs (models.py:115-136) */
#define PAMNET_SCHEMA_RNA 2     /* rows = xyz + type; kNN graph cut at both cutoffs (models.py:138-157) */
typedef struct pamnet_graph_desc {
    int64_t n, n_graphs, n_bonds; /* nodes, graphs, directed bonds (QM9; 0 otherwise) */
    int64_t eg, el, tp;           /* sizes the host assumes: global edges, local edges, triplet + pair rows */
    const void* batch;            /* [n] graph id per node, sorted; int64 / int32 / fp32 by batch_kind (1 / 2 / 3) */
    const void* types;            /* QM9 / RNA: atom type of node i at types[i * types_stride], kind as above */
    int64_t types_stride, n_types;
    const float* pos;             /* QM9: [n, 3] */
    const float* rows;            /* PDBbind / RNA: [n, rows_width] fp32, xyz first */
    int64_t rows_width;
    const void* edge_src;         /* QM9: edge_index[0], edge_index[1] ([n_bonds] each), kind edge_kind */
    const void* edge_dst;
    int32_t schema, batch_kind, types_kind, edge_kind;
    int32_t with_triplets;        /* 0: pairs only (PAMNet_s) */
    int32_t need_grad;            /* 1: also build the transposed index lists the backward gathers with, and the two index hops of
                                     the dim-128 engine's local aggregation backward (TT_EDGE / TT_NODE); 2: the lists only */
    int32_t aggregate_at_query;   /* RNA: flow == 'target_to_source' (the global layer aggregates at the kNN query) */
    int32_t knn_k;                /* RNA: neighbours per query (models.py:143: 50) */
    float cutoff_l, cutoff_g;
    int32_t max_neighbors;        /* radius searches: torch_cluster's max_num_neighbors (models.py:110,128: 1000; :301: 500;
                                     0 = unlimited).  A batch in which it binds is flagged (bit 64 of the flag word): the one-call
                                     graph assumes symmetric radius graphs */
    int32_t mol_local;            /* QM9: 1 = the molecule-local builder (pamnet_mol_graph_*: the caller vouches for <= 64 atoms,
                                     <= 256 directed bonds per molecule and bonds grouped by molecule in batch order; a batch
                                     that is not so is flagged as a local-edge size mismatch) */
} pamnet_graph_desc;
enum {
    PAMNET_GF_NODE_GRAPH = 0, /* int32 [n] */
    PAMNET_GF_GPTR,           /* int32 [n_graphs + 1] first node of every graph */
    PAMNET_GF_FLAG,           /* int32 [1]: bit 1 invalid index inputs, bits 2 / 4 / 8 size mismatch (eg / el / tp), 32 self loops, 64 neighbour cap binds */
    PAMNET_GF_LOOPS,          /* int32 [1] */
    PAMNET_GF_TYPES,          /* int32 [n] */
    PAMNET_GF_POS,            /* fp32 [n, 3] (PDBbind / RNA; QM9 uses the caller's) */
    PAMNET_GF_SIGN,           /* fp32 [n] pooling sign (PDBbind, models.py:124) */
    PAMNET_GF_G_PTR, PAMNET_GF_G_ROW, PAMNET_GF_G_COL, PAMNET_GF_G_DIST,   /* global graph, CSR by aggregation target */
    PAMNET_GF_GT_PTR, PAMNET_GF_GT_PERM,                                     /* its transposed CSR (need_grad) */
    PAMNET_GF_L_PTR, PAMNET_GF_L_ROW, PAMNET_GF_L_COL, PAMNET_GF_L_DIST,   /* local graph */
    PAMNET_GF_LT_PTR, PAMNET_GF_LT_PERM,
    PAMNET_GF_T_PTR, PAMNET_GF_T_ROW, PAMNET_GF_T_COL, PAMNET_GF_T_ANGLE, PAMNET_GF_T_KIND,   /* triplet + pair rows */
    PAMNET_GF_TT_PTR, PAMNET_GF_TT_PERM,
    PAMNET_GF_CUTS,           /* int32 [<= 257] node-aligned work split of the fused global-edge kernels */
    PAMNET_GF_TT_EDGE, PAMNET_GF_TT_NODE,   /* int32 [tp] each: pamnet_triplet_transpose_aux_i32 (need_grad) */
    PAMNET_GRAPH_FIELDS
};
/* Molecule-local graph construction for the QM9 schema (positions + bond list given; models.py:62-98, 104-118, 165-177) --
 * bond CSR by target + bond lengths, triplet / pair rows + angles, radius graph at cutoff_g, and (need_grad) the transposed
 * index lists of the backward -- one wavefront per molecule, two launches instead of the ~14 of the step-by-step entry points
 * above, bit-identical arrays.  For batches of small molecules: <= 64 atoms and <= 256 directed bonds per molecule; bonds
 * grouped by molecule in batch order (torch_geometric collation, pamnet_collate_f32) with both ends in one molecule; no
 * self loops.  gptr [n_graphs + 1], src / dst int32 [n_bonds] as pamnet_ingest_indices_i32 returns them.
 *   count: mol_tot [n_graphs, 4] = (global edges, triplet + pair rows, first bond, violation bits) per molecule;
 *          totals [4] (zeroed by the caller) += (global edges, triplet + pair rows, violation bits (OR), bonds) over the
 *          molecules without a violation (bits: 1 atoms, 2 bonds, 4 a bond leaving its molecule / bonds not grouped);
 *   fill:  every array of `out`; eg_cap / tp_cap = what the buffers hold (writes beyond are dropped, pointers clamped).
 * Array sizes: *_ptr over nodes [n + 1] (g, l, lT) or over bonds [n_bonds + 1] (t, tT); g_* / gT_perm [eg_cap];
 * l_* / lT_perm [n_bonds]; t_* / tT_perm [tp_cap].  gT_ptr = g_ptr (a radius graph is symmetric: the transposed list is the
 * reverse-edge index).  The transposed lists are written only with need_grad. */
typedef struct pamnet_mol_graph_out {
    int32_t *g_ptr, *g_row, *g_col;
    float* g_dist;
    int32_t* gT_perm;
    int32_t *l_ptr, *l_row, *l_col;
    float* l_dist;
    int32_t *lT_ptr, *lT_perm;
    int32_t *t_ptr, *t_row, *t_col;
    float* t_angle;
    int32_t* t_kind;
    int32_t *tT_ptr, *tT_perm;
} pamnet_mol_graph_out;
int pamnet_mol_graph_count_i32(const float* pos, const int32_t* gptr, int64_t n, int64_t n_graphs, const int32_t* src,
                               const int32_t* dst, int64_t n_bonds, float cutoff_g, int32_t with_triplets, int32_t* mol_tot,
                               int32_t* totals, pamnet_stream_t stream);
int pamnet_mol_graph_fill_i32(const float* pos, const int32_t* gptr, int64_t n, int64_t n_graphs, const int32_t* src,
                              const int32_t* dst, int64_t n_bonds, float cutoff_g, int32_t with_triplets, int32_t need_grad,
                              const int32_t* mol_tot, int64_t eg_cap, int64_t tp_cap, const pamnet_mol_graph_out* out,
                              pamnet_stream_t stream);

int pamnet_graph_plan(const pamnet_graph_desc* desc, int64_t* layout /* [PAMNET_GRAPH_FIELDS] */, int64_t* arena_ints);
int pamnet_graph_build_i32(const pamnet_graph_desc* desc, int32_t* arena, float* sbf /* [tp, 42], nullable */,
                           pamnet_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Basis functions  (layers/basic.py:36-51 Envelope, :59-76 BesselBasisLayer, :79-116 SphericalBasisLayer; utils/sbf.py)
 * ------------------------------------------------------------------------------------------------------------------ */
/* rbf[e, n] = env(d_e/c) * sin(freq[n] * d_e/c),  n < 16, envelope exponent p = 5 */
int pamnet_rbf_fwd_f32(const float* dist, const float* freq, float cutoff, int64_t m, float* rbf,
                       pamnet_stream_t stream);
/* dfreq[n] = sum_e grad[e, n] * env(x_e) * x_e * cos(freq[n] x_e)   (freq is trainable: basic.py:65-72).
 * `partial` needs 16*2048 floats of scratch. */
int pamnet_rbf_bwd_f32(const float* dist, const float* freq, float cutoff, int64_t m, const float* grad,
                       float* dfreq, float* partial, pamnet_stream_t stream);
/* radial table rad[e, l*6+n] = env(x) * N_ln * j_l(z_ln x), x = d_e/c, evaluated in fp64 and rounded once to fp32 */
int pamnet_sbf_radial_f32(const float* dist, float cutoff, int64_t m, float* rad, pamnet_stream_t stream);
/* sbf[t, l*6+n] = rad[idx[t], l*6+n] * Y_l0(angle[t]) */
int pamnet_sbf_combine_f32(const float* rad, const int32_t* idx, const float* angle, int64_t m, float* sbf,
                           pamnet_stream_t stream);
/* The same basis layers for any PAMNet(config, num_spherical, num_radial, envelope_exponent) (models.py:22; the entries
 * above are specialised for the default (7, 6, 5) every script of the reference uses): num_spherical <= 16, num_radial <= 64,
 * envelope exponent p >= 1 (layers/basic.py:36-51).  zeros [ns * nr] (float32: utils/sbf.py:15-26 stores them so) and norm
 * [ns * nr] (float64: N_ln = 1 / sqrt(0.5 j_{l+1}(z_ln)^2), utils/sbf.py:43-49) are device tables the caller computes once
 * per model.  rad / sbf rows have ns * nr entries, index l * nr + n. */
int pamnet_rbf_fwd_env_f32(const float* dist, const float* freq, float cutoff, int32_t envelope_exponent, int64_t m,
                           float* rbf, pamnet_stream_t stream);
int pamnet_rbf_bwd_env_f32(const float* dist, const float* freq, float cutoff, int32_t envelope_exponent, int64_t m,
                           const float* grad, float* dfreq, float* partial, pamnet_stream_t stream);
int pamnet_sbf_radial_tab_f32(const float* dist, float cutoff, int64_t m, int32_t num_spherical, int32_t num_radial,
                              int32_t envelope_exponent, const float* zeros, const double* norm, float* rad,
                              pamnet_stream_t stream);
int pamnet_sbf_combine_tab_f32(const float* rad, const int32_t* idx, const float* angle, int64_t m, int32_t num_spherical,
                               int32_t num_radial, float* sbf, pamnet_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Attention fusion + per-graph pooling  (models.py:206-224)
 * outs/atts: [2L, n] rows ordered (global_0, local_0, global_1, local_1, ...).
 *   node_out[n] = sign[n] * sum_l sum_c outs[l,c,n] * softmax_c(leaky_relu(atts[l,c,n], 0.2))
 *   graph_out[b] = (mean ? 1/|b| : 1) * sum_{n in b} node_out[n];  sign may be NULL (=1).
 * ------------------------------------------------------------------------------------------------------------------ */
int pamnet_fuse_pool_fwd_f32(const float* outs, const float* atts, int64_t n_layer, int64_t n, const float* sign,
                             const int32_t* gptr, int64_t n_graphs, int32_t mean, float* node_out, float* graph_out,
                             pamnet_stream_t stream);
int pamnet_fuse_pool_bwd_f32(const float* outs, const float* atts, int64_t n_layer, int64_t n, const float* sign,
                             const int32_t* node_graph, const int32_t* gptr, int32_t mean, const float* grad_graph,
                             float* grad_outs, float* grad_atts, pamnet_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Fused dense chains (dim = 128 only), fp32 MFMA.  Weight pointers are HOST arrays of DEVICE pointers to row-major
 * [out=128][in] blocks (the layout of an nn.Linear weight; row stride 128 unless an ld is given).
 *
 * node_tail: the 10-Linear update stack + heads shared by both layer kinds
 *   (layers/global_message_passing.py:39-50, layers/local_message_passing.py:55-66; Res: layers/basic.py:25-33):
 *   weights/biases order: mlp_x2, res1.0, res1.1, res2.0, res2.1, res3.0, res3.1, mlp_out.0, mlp_out.1, mlp_out.2.
 *   Saves Z[10][n][128] (pre-activations) and R[2][n][128] (r1, r2) for the backward.
 *   next_* (next_nblk = 0: none): the node_pre of the FOLLOWING layer applied to x_out while its tile is still on chip
 *   (same outputs as pamnet_node_pre_fwd_f32 on x_out) -- the layer loop then needs one node-level launch per layer.
 * node_pre: x1 = SiLU(mlp_x1 x) and the node-level halves P = x1 * Wp_b^T (b < nblk <= 4) of the split message MLPs
 *   (mlp_m / mlp_m_ji / mlp_m_kj on [x_i | x_j | e]: layers/global_message_passing.py:52-56, local...:46-48).
 * wgrad_batched: dW_j = dZ_j^T * A_j (A_j optionally SiLU'd on load), db_j = colsum(dZ_j) for up to 24 jobs, in two
 *   launches (split-K partial tiles, then one fixed-order reduction).  The reduction launch can also finish a node_tail
 *   backward: pass that call's head_partial / workgroup count (= ceil(n/16)) and the three head-gradient outputs here and
 *   give node_tail_bwd null d_wout/d_watt/d_bout (then it launches no reduction of its own).
 * ------------------------------------------------------------------------------------------------------------------ */
int pamnet_node_tail_fwd_f32(const float* x2, const float* res_x, int64_t n, const float* const* weights,
                             const float* const* biases, const float* w_out, const float* b_out, const float* w_att,
                             float* Z, float* R, float* x_out, float* out, float* att, const float* next_Wx1,
                             const float* next_bx1, const float* const* next_wp, int64_t next_ldwp, int64_t next_nblk,
                             float* next_Zx1, float* next_x1, float* next_P, int32_t packed, pamnet_stream_t stream);
/* pamnet_node_tail_fwd_f32 (packed weight images, deferred heads) with a RIDER: row tiles [mlp_tile0, mlp_tile0 + mlp_ntiles)
 * (16 rows each) of the two-layer MLP of pamnet_mlp2_fwd_f32 over mlp_x [mlp_rows,128] run as `rider_wgs` extra workgroups
 * of the launch -- the chain occupies only ceil(n/16) of the 256 CUs, the triplet/pair MLPs of the next layers do not
 * depend on the node features.  mlp = {W1, b1, W2, b2}; mlp_out = {z1, z2, y} (z1 / z2 nullable saves). */
int pamnet_node_tail_fwd_rider_f32(const float* x2, const float* res_x, int64_t n, const float* const* weights,
                                   const float* const* biases, const float* w_out, const float* b_out,
                                   const float* w_att, float* Z, float* R, float* x_out, const float* next_Wx1,
                                   const float* next_bx1, const float* const* next_wp, int64_t next_ldwp,
                                   int64_t next_nblk, float* next_Zx1, float* next_x1, float* next_P, const float* mlp_x,
                                   int64_t mlp_rows, int64_t mlp_tile0, int64_t mlp_ntiles, const float* const* mlp,
                                   float* const* mlp_out, int64_t rider_wgs, int32_t packed /* 1 or 2 */,
                                   pamnet_stream_t stream);
/* out = att = null in pamnet_node_tail_fwd_f32 leaves the head branch of the chain (mlp_out, W_out, W:
 * layers/global_message_passing.py:46-50) to this call, which runs it for n_layers chains in one launch (nothing
 * downstream of a layer depends on its heads): per layer l  x_out[l] [n,128] -> out[l] [n], att[l] [n], and slots 7..9
 * of that layer's pre-activation block Z[l] ([10][n][128]; Z or Z[l] null: not saved).  weights / biases: 3 per layer
 * (mlp_out), fragment images when `packed`. */
int pamnet_node_heads_fwd_f32(int64_t n_layers, const float* const* x_out, const float* const* weights,
                              const float* const* biases, const float* const* w_out, const float* const* b_out,
                              const float* const* w_att, float* const* Z, float* const* out, float* const* att, int64_t n,
                              int32_t packed, pamnet_stream_t stream);
int pamnet_node_tail_bwd_f32(const float* d_xout, const float* d_out, const float* d_att, int64_t n,
                             const float* const* weights, const float* w_out, const float* w_att, const float* Z,
                             float* dZ, float* d_x2, float* d_resx, float* head_partial, float* d_wout, float* d_watt,
                             float* d_bout, int32_t packed, pamnet_stream_t stream);
/* Backward counterparts of the deferred head branch.  heads_bwd (all layers, one launch; needs only d out / d att):
 * per layer l the head-vector partials (head_partial[l]: ceil(n/16) x 257 floats, reduced by pamnet_wgrad_batched_f32),
 * dZ3[l] = [dz7, dz8, dz9] ([3][n][128]) and g_head[l] = the branch's contribution to d x_out ([n][128]); weights: the
 * mlp_out matrices 7, 8, 9 of every layer (transposed-orientation images when `packed`).  tail_main_bwd: the rest of
 * the chain (layers 6..0) from d x_out = d_xout (may be null) + g_head; writes dZ slots 0..6. */
int pamnet_node_heads_bwd_f32(int64_t n_layers, const float* const* d_out, const float* const* d_att,
                              const float* const* weights, const float* const* w_out, const float* const* w_att,
                              const float* const* Z, float* const* dZ3, float* const* g_head, float* const* head_partial,
                              int64_t n, int32_t packed, pamnet_stream_t stream);
int pamnet_node_tail_main_bwd_f32(const float* d_xout, const float* g_head, int64_t n, const float* const* weights,
                                  const float* Z, float* dZ, float* d_x2, float* d_resx, int32_t packed,
                                  pamnet_stream_t stream);
/* pamnet_node_pre_bwd_f32 (backward of a layer's head: dP [nblk][n][128], d x1_direct, d_add -> dZx1 and d x) fused
 * with pamnet_node_tail_main_bwd_f32 of the chain that produced that layer's input: the head's d x becomes the chain's
 * d x_out on chip.  All weights are transposed-orientation images (pamnet_pack_weights_f32; with nblk | PAMNET_CHAIN_PIECES:
 * bf16x3 images, the chain on the bf16 matrix pipe at fp32 accuracy). */
int pamnet_node_pre_tail_bwd_f32(const float* dP, const float* dx1_direct, const float* d_add, int64_t n,
                                 const float* Wx1, const float* const* wp, int64_t nblk, const float* Zx1, float* dZx1,
                                 const float* g_head, const float* const* weights, const float* Z, float* dZ,
                                 float* d_x2, float* d_resx, const void* rider /* host, nullable */,
                                 pamnet_stream_t stream);
/* The operands of pamnet_local_agg_fwd_f32 as one argument (round 6): pamnet_node_tail_fwd_agg_f32 is
 * pamnet_node_tail_fwd_rider_f32 (mlp_ntiles = 0: without riders) whose chain input x2 is formed by the launch's own row tiles --
 * the two chained aggregations of layers/local_message_passing.py:49-54, bit for bit pamnet_local_agg_fwd_f32's rows -- and
 * written to x2 (and m_t) as well; chain forms that read their input run the aggregation as a launch of its own first. */
typedef struct pamnet_local_agg {
    const float *m_ji, *m_nb, *s, *q3, *init; /* init nullable = 0 */
    const int32_t *t_ptr, *t_col, *l_ptr;
    float* m_t;                               /* nullable: backward-only save */
} pamnet_local_agg;
int pamnet_node_tail_fwd_agg_f32(float* x2, const float* res_x, int64_t n, const float* const* weights,
                                 const float* const* biases, const float* w_out, const float* b_out, const float* w_att,
                                 float* Z, float* R, float* x_out, const float* next_Wx1, const float* next_bx1,
                                 const float* const* next_wp, int64_t next_ldwp, int64_t next_nblk, float* next_Zx1,
                                 float* next_x1, float* next_P, const float* mlp_x, int64_t mlp_rows, int64_t mlp_tile0,
                                 int64_t mlp_ntiles, const float* const* mlp, float* const* mlp_out, int64_t rider_wgs,
                                 int32_t packed, const pamnet_local_agg* agg, pamnet_stream_t stream);
/* The same with planes of dP formed inside the launch (round 6): gather_src[b] != null -> plane b, row i = the sum of
 * gather_src[b][gather_perm[b] ? gather_perm[b][q] : q] over q in [gather_ptr[b][i], gather_ptr[b][i+1]) in that order, used and
 * also written to dP + b n 128 -- the pamnet_segment_sum(_multi)_f32 launches that otherwise run ahead of this one. */
int pamnet_node_pre_tail_bwd_gather_f32(float* dP, const float* const* gather_src, const int32_t* const* gather_ptr,
                                        const int32_t* const* gather_perm, const float* dx1_direct, const float* d_add,
                                        int64_t n, const float* Wx1, const float* const* wp, int64_t nblk, const float* Zx1,
                                        float* dZx1, const float* g_head, const float* const* weights, const float* Z,
                                        float* dZ, float* d_x2, float* d_resx, const void* rider /* host, nullable */,
                                        pamnet_stream_t stream);
/* Fragment-ordered weight images for the node chains: n (<= 192) 128x128 matrices (row stride ld[i]) -> images[i*16384..],
 * transposed = 0 for the forward (Y = X W^T), 1 for the backward (Y = X W).  With packed != 0 the `weights` (and, in the
 * forward, next_Wx1 / next_wp) arguments of node_tail_fwd / node_tail_bwd, and Wx1 / wp of node_pre_bwd (transposed
 * images), are such images: every weight-slice request of
 * a wave is then one contiguous 1 KB read instead of 16 half-used cache lines of the row-major matrix. */
int pamnet_pack_weights_f32(int64_t n, const float* const* W, const int64_t* ld, int32_t transposed, float* images,
                            pamnet_stream_t stream);
/* The same matrices as bf16x3 fragment images (images[i*24576..], 96 KB each): every weight split exactly into three bf16
 * pieces, laid out as operand fragments of v_mfma_f32_16x16x32_bf16 (the kind-1 images of pamnet_pack_weights_mixed_f32 below,
 * one after the other).  With packed == 2, pamnet_node_tail_fwd_f32 (deferred
 * heads: out = att = null) and pamnet_node_tail_fwd_rider_f32 take such images for weights[0..6], next_Wx1 and next_wp and
 * run the chain on the bf16 matrix pipe at fp32 accuracy (six piece products per product: csrc/gemm_core.h "bf16x6"). */
int pamnet_pack_weights_bf16x3(int64_t n, const float* const* W, const int64_t* ld, int32_t transposed, float* images,
                               pamnet_stream_t stream);
/* Round 6 (ABI 14): all weight images of a step direction in one launch, n <= 224.  kind[i] = 0: the chains' fp32 fragment
 * image of pamnet_pack_weights_f32 (16 384 floats); kind[i] = 1: the EDGE-LEVEL kernels' bf16x3 fragment image (24 576 floats):
 * uint4 image[((tile * 4 + q) * 3 + piece) * 64 + lane] = the exact bf16 pieces lane `lane` of the wave that owns output
 * columns [16 tile, 16 tile + 16) holds for k-step q (csrc/edge_core.h load_wfragb1) -- what every workgroup of
 * pamnet_global_edge_agg_fwd_f32 / _fwd_pp_f32 / _bwd_f32 and pamnet_local_edge_fwd_f32 otherwise makes of its slices itself
 * (64-128 loads and ~300 vector instructions per lane ahead of the first row: 2 us of a 30 us launch at the QM9 batch).  Those
 * four entry points take such an image in place of a weight matrix when its row stride argument is 0 (all of a call's strides
 * zero or none; pamnet_local_edge_fwd_f32: 8-wave geometry only); the results are bitwise those of the fp32 matrices.
 * offset[i]: where image i starts in `images`, in floats (a multiple of 4).  transposed as in pamnet_pack_weights_f32
 * (0: Y = X W^T, the forward entries; 1: Y = X W, pamnet_global_edge_agg_bwd_f32). */
int pamnet_pack_weights_mixed_f32(int64_t n, const float* const* W, const int64_t* ld, const int32_t* kind,
                                  const int64_t* offset, int32_t transposed, float* images, pamnet_stream_t stream);
int pamnet_node_pre_fwd_f32(const float* x, int64_t n, const float* Wx1, const float* bx1, const float* const* wp,
                            int64_t ldwp, int64_t nblk, float* Zx1, float* x1, float* P, pamnet_stream_t stream);
int pamnet_node_pre_bwd_f32(const float* dP, const float* dx1_direct, const float* d_add, int64_t n, const float* Wx1,
                            const float* const* wp, int64_t ldwp, int64_t nblk, const float* Zx1, float* dZx1,
                            float* dx, int32_t packed, pamnet_stream_t stream);
int pamnet_wgrad_scratch_floats(int64_t njobs, const int64_t* rows, int64_t* floats);
int pamnet_wgrad_batched_f32(int64_t njobs, const float* const* dZ, const int64_t* ld_dz, const float* const* A,
                             const int64_t* ld_a, const int32_t* a_mode, const int64_t* rows, float* const* dW,
                             const int64_t* ld_dw, float* const* db, float* partial, const float* head_partial,
                             int64_t head_blocks, float* d_wout, float* d_watt, float* d_bout, pamnet_stream_t stream);
/* Deferred form for a sequence of batches (the layers of one backward pass): the fixed-order reduction of batch i runs
 * inside the launch of batch i+1's split-K pass; pamnet_wgrad_flush_f32 reduces the last one.  ctx: caller-owned HOST
 * memory of *bytes (pamnet_wgrad_ctx_bytes) bytes, zeroed before the first call -- the library keeps no state of its
 * own.  Consecutive calls must pass different `partial` buffers. */
int pamnet_wgrad_ctx_bytes(int64_t* bytes /* host */);
int pamnet_wgrad_deferred_f32(int64_t njobs, const float* const* dZ, const int64_t* ld_dz, const float* const* A,
                              const int64_t* ld_a, const int32_t* a_mode, const int64_t* rows, float* const* dW,
                              const int64_t* ld_dw, float* const* db, float* partial, const float* head_partial,
                              int64_t head_blocks, float* d_wout, float* d_watt, float* d_bout,
                              const float* head2_partial /* nullable: a second chain's head-vector partials (same head_blocks):
                                                            a layer pair's merged batch carries the local and the global chain's */,
                              float* d_wout2, float* d_watt2, float* d_bout2, void* ctx /* host */,
                              pamnet_stream_t stream);
int pamnet_wgrad_flush_f32(void* ctx /* host */, pamnet_stream_t stream);
/* Riders: weight-gradient slots as extra workgroups of a node-chain backward launch (the chain owns ceil(n/16) workgroups,
 * 143 of the 256 CUs at the QM9 batch; the riders take the idle CUs and the layer's own weight-gradient launch shrinks).
 *   pamnet_wgrad_rider_plan_f32   : lay a batch (<= 12 jobs) out over <= max_slots slots; the plan goes to `rider`
 *                                   (caller-owned HOST memory of pamnet_wgrad_rider_bytes bytes); *slots_out = slots used
 *   pamnet_node_pre_tail_bwd_f32  : takes the plan (`rider` argument) and appends the slots to its grid
 *   pamnet_wgrad_rider_enqueue_f32: registers the batch with a deferred context so that the next pamnet_wgrad_deferred_f32
 *                                   launch (or the flush) reduces its slots in the usual fixed order.  A second rider batch
 *                                   before that launch is appended to the first (<= 24 jobs together); its `partial` must
 *                                   start right behind the first one's slots in the same buffer. */
int pamnet_wgrad_rider_bytes(int64_t* bytes);
int pamnet_wgrad_rider_plan_f32(int64_t njobs, const float* const* dZ, const int64_t* ld_dz, const float* const* A,
                                const int64_t* ld_a, const int32_t* a_mode, const int64_t* rows, float* const* dW,
                                const int64_t* ld_dw, float* const* db, float* partial, int64_t max_slots,
                                void* rider /* host */, int64_t* slots_out);
int pamnet_wgrad_rider_enqueue_f32(void* ctx /* host */, const void* rider /* host */);

/* ------------------------------------------------------------------------------------------------------------------
 * Fused edge-level kernels (dim = 128), fp32 MFMA.  P planes are node_pre outputs ([N][128] each).
 * global edge (layers/global_message_passing.py:52-56 + PyG propagate gathers):
 *     z = W_e e + b_m + Pi[row_of] + Pj[col];  ea = W_ea e;  msg = SiLU(z) * ea          (z, ea saved for backward)
 *   bwd: dm = d_agg[row_of]; dz = dm*ea*SiLU'(z); dea = dm*SiLU(z); d_e (+)= dz W_e + dea W_ea
 * local edge (layers/local_message_passing.py:46-48,53):  Wq = {W_ji[:,2d:], W_kj[:,2d:], lin_rbf, lin_rbf_out},
 *     P = {ji_i, kj_i, ji_j, kj_j}:  z_ji, z_kj, q2 = lin_rbf r, q3 = lin_rbf_out r, m_ji = SiLU(z_ji),
 *     m_nb = SiLU(z_kj) * q2
 * mlp2 (layers/local_message_passing.py:49 `mlp_sbf`): y = SiLU(W2 SiLU(W1 x + b1) + b2); z1, z2 saved.
 * `accumulate` != 0: the input-gradient output is added to (edge embeddings are shared by all layers).
 * ------------------------------------------------------------------------------------------------------------------ */
int pamnet_global_edge_fwd_f32(const float* e, int64_t n_edges, const float* We, int64_t ld_we, const float* bm,
                               const float* Wea, int64_t ld_wea, const float* Pi, const float* Pj,
                               const int32_t* row_of, const int32_t* col, float* z, float* ea, float* msg,
                               pamnet_stream_t stream);
int pamnet_global_edge_bwd_f32(const float* d_agg, const int32_t* row_of, int64_t n_edges, const float* z,
                               const float* ea, const float* We, int64_t ld_we, const float* Wea, int64_t ld_wea,
                               float* dz, float* dea, float* d_e, int32_t accumulate, pamnet_stream_t stream);
int pamnet_local_edge_fwd_f32(const float* rbf, int64_t n_edges, const float* const* Wq, const int64_t* ldq,
                              const float* b_ji, const float* b_kj, const float* const* P, const int32_t* row_of,
                              const int32_t* col, float* z_ji, float* z_kj, float* q2, float* q3, float* m_ji,
                              float* m_nb, pamnet_stream_t stream);
int pamnet_local_edge_bwd_f32(const float* d_mji, const float* d_mnb, const float* d_q3, int64_t n_edges,
                              const float* z_ji, const float* z_kj, const float* q2, const float* const* Wq,
                              const int64_t* ldq, float* dz_ji, float* dz_kj, float* dq2, float* d_rbf,
                              int32_t accumulate, pamnet_stream_t stream);
/* Edge MLP -> node segment-sum as ONE kernel (csrc/edge_agg.hip; layers/global_message_passing.py:38,52-56): the
 * message tile is reduced over its target node from LDS, the [E,128] message tensor is never written.
 *   fwd: out[i] = init[i] + sum_{e -> i} SiLU(W_e e + b_m + Pi[i] + Pj[col[e]]) * (W_ea e)  for all i < n_nodes
 *        (init nullable = 0; z / ea [E,128]: optional saves for the backward; ptr [n_nodes+1], row_of, col: CSR by target)
 *   bwd: dz, dea, d_e as pamnet_global_edge_bwd_f32, plus dPi[i] = sum_{e -> i} dz[e] (the target-side reduction).
 * Node-aligned work split: every output row has one owner and a fixed CSR summation order (no atomics, no carries;
 * results do not depend on the launch geometry).
 * pamnet_local_agg_fwd_f32 (layers/local_message_passing.py:49-54, one launch for both aggregations):
 *   m_t[e] = m_ji[e] + sum_{r in [t_ptr[e], t_ptr[e+1])} m_nb[t_col[r]] * s[r]     (m_t nullable: backward-only save)
 *   out[i] = init[i] + sum_{e in [l_ptr[i], l_ptr[i+1])} q3[e] * m_t[e] */
int pamnet_global_edge_agg_fwd_f32(const float* e, int64_t n_edges, int64_t n_nodes, const float* We, int64_t ld_we,
                                   const float* bm, const float* Wea, int64_t ld_wea, const float* Pi, const float* Pj,
                                   const int32_t* ptr, const int32_t* row_of, const int32_t* col,
                                   const int32_t* cuts /* nullable: pamnet_seg_cuts_i32 */, const float* init, float* z,
                                   float* ea, float* out, pamnet_stream_t stream);
/* The same operator with the two halves of every workgroup in opposite phases (matrix pipe beside vector units: the form
 * pamnet_global_edge_agg_fwd_f32 takes by itself from 131 072 edges; PAMNET_AGG_PP=0/1 forces either).  Same arguments, the
 * same results bit for bit, whatever the size. */
int pamnet_global_edge_agg_fwd_pp_f32(const float* e, int64_t n_edges, int64_t n_nodes, const float* We, int64_t ld_we,
                                      const float* bm, const float* Wea, int64_t ld_wea, const float* Pi, const float* Pj,
                                      const int32_t* ptr, const int32_t* row_of, const int32_t* col,
                                      const int32_t* cuts /* nullable */, const float* init, float* z, float* ea, float* out,
                                      pamnet_stream_t stream);
/* cuts[0 .. G] = the node-aligned work split of the fused kernels for this graph (node boundary nearest to edge row
 * k * n_edges / G), G = *grid_out <= 256: computed once per graph, handed to every pamnet_global_edge_agg_fwd_f32 launch
 * (without it every workgroup derives its cuts itself: two dependent loads ahead of everything else). */
int pamnet_seg_cuts_i32(const int32_t* ptr, const int32_t* row_of, int64_t n_nodes, int64_t n_edges, int32_t* cuts,
                        int64_t* grid_out, pamnet_stream_t stream);
int pamnet_global_edge_agg_bwd_f32(const float* d_agg, int64_t n_edges, int64_t n_nodes, const int32_t* ptr,
                                   const int32_t* row_of, const int32_t* cuts /* nullable */, const float* z,
                                   const float* ea, const float* We,
                                   int64_t ld_we, const float* Wea, int64_t ld_wea, float* dz, float* dea, float* d_e,
                                   int32_t accumulate, float* dPi, pamnet_stream_t stream);
/* The same backward WITH the step's two weight gradients (layers/global_message_passing.py:52-56: the gradients of the
 * e-block of mlp_m and of W_edge_attr): d ea is not written; every workgroup forms its rows' share of dW_e = dz^T e,
 * dW_ea = dea^T e and of the bias gradient (column sums of dz) and leaves them in `partial`: 2 * G slots of 128 * 128 + 256
 * floats (pamnet_global_edge_agg_wg_floats: *floats, *slots = G), slots [0, G) = dW_e shares + bias parts, [G, 2 G) = dW_ea
 * shares.  pamnet_wgrad_edge_enqueue_f32 registers them with a deferred weight-gradient context: the next
 * pamnet_wgrad_deferred_f32 launch (or pamnet_wgrad_flush_f32) sums them in slot order into dW_e / db / dW_ea.
 * dz, d_e, dPi: bitwise the values of pamnet_global_edge_agg_bwd_f32. */
int pamnet_global_edge_agg_wg_floats(int64_t n_edges, int64_t* floats /* host */, int64_t* slots /* host, nullable */);
int pamnet_global_edge_agg_bwd_wg_f32(const float* d_agg, int64_t n_edges, int64_t n_nodes, const int32_t* ptr,
                                      const int32_t* row_of, const int32_t* cuts /* nullable */, const float* z,
                                      const float* ea, const float* e, const float* We, int64_t ld_we, const float* Wea,
                                      int64_t ld_wea, float* dz, float* d_e, int32_t accumulate, float* dPi, float* partial,
                                      pamnet_stream_t stream);
int pamnet_wgrad_edge_enqueue_f32(void* ctx /* host */, int64_t slots, float* dW_e, int64_t ld_e, float* db /* nullable */,
                                  float* dW_ea, int64_t ld_ea, const float* partial);
int pamnet_local_agg_fwd_f32(const float* m_ji, const float* m_nb, const float* s, const float* q3,
                             const int32_t* t_ptr, const int32_t* t_col, const int32_t* l_ptr, const float* init,
                             int64_t n_nodes, float* m_t, float* out, pamnet_stream_t stream);
/* backward of pamnet_local_agg_fwd_f32, one launch: d_mt[e] = d_x2[l_row[e]] * q3[e];  d_q3[e] = d_x2[l_row[e]] * m_t[e];
 * d_s[r] = m_nb[t_col[r]] * d_mt[t_row[r]];  d_mnb[e'] = sum_{r: t_col[r] = e'} s[r] * d_mt[t_row[r]]
 * (tT_ptr / tT_perm: transposed CSR of t_col over the local edges). */
int pamnet_local_agg_bwd_f32(const float* d_x2, const int32_t* l_row, const float* q3, const float* m_t,
                             const float* m_nb, const float* s, const int32_t* t_ptr, const int32_t* t_col,
                             const int32_t* t_row, const int32_t* tT_ptr, const int32_t* tT_perm,
                             const int32_t* tT_edge /* nullable */, const int32_t* tT_node /* nullable */, int64_t n_edges,
                             float* d_mt, float* d_q3, float* d_s, float* d_mnb, pamnet_stream_t stream);
/* tT_edge[q] = t_row[tT_perm[q]], tT_node[q] = l_row[tT_edge[q]] for the n_rows entries of the transposed triplet / pair list:
 * made once per graph, they turn the gather half of pamnet_local_agg_bwd_f32 from three dependent index reads per term into
 * one level of independent ones (both or neither may be given). */
int pamnet_triplet_transpose_aux_i32(const int32_t* tT_perm, const int32_t* t_row, const int32_t* l_row, int64_t n_rows,
                                     int32_t* out_edge, int32_t* out_node, pamnet_stream_t stream);
int pamnet_mlp2_fwd_f32(const float* x, int64_t rows, const float* W1, const float* b1, const float* W2,
                        const float* b2, float* z1, float* z2, float* y, pamnet_stream_t stream);
/* nsets <= 8 such MLPs on the same input rows in one launch: params[4k..4k+3] = {W1, b1, W2, b2} of set k,
 * outs[3k..3k+2] = {z1, z2, y} (host arrays of device pointers).  The per-layer mlp_sbf of all layers take the same input. */
int pamnet_mlp2_fwd_multi_f32(const float* x, int64_t rows, int64_t nsets, const float* const* params,
                              float* const* outs, pamnet_stream_t stream);
int pamnet_mlp2_bwd_f32(const float* dy, int64_t rows, const float* z1, const float* z2, const float* W1,
                        const float* W2, float* dz1, float* dz2, float* dx, int32_t accumulate,
                        pamnet_stream_t stream);
/* pamnet_mlp2_bwd_f32 + pamnet_local_edge_bwd_f32 -- both depend on pamnet_local_agg_bwd_f32 only, not on each other -- as ONE
 * launch, the CUs split between the two plans by their work (layers/local_message_passing.py:46-53 backward).  Arguments
 * and results are those of the two calls.  Round 6 (ABI 14): accumulate_dx | PAMNET_WEIGHT_IMAGES: W1, W2 are transposed kind-1
 * images of pamnet_pack_weights_mixed_f32; ldq[0..3] all 0 (here and in pamnet_local_edge_bwd_f32): Wq[0..3] are transposed
 * kind-1 images too (the four dX GEMMs of the local edge stage run on the bf16 matrix pipe at fp32 accuracy, like the MLP's).
 * Same bits as with the matrices. */
int pamnet_local_bwd_pair_f32(const float* dy, int64_t rows, const float* z1, const float* z2, const float* W1,
                              const float* W2, float* dz1, float* dz2, float* dx, int32_t accumulate_dx,
                              const float* d_mji, const float* d_mnb, const float* d_q3, int64_t n_edges,
                              const float* z_ji, const float* z_kj, const float* q2, const float* const* Wq,
                              const int64_t* ldq, float* dz_ji, float* dz_kj, float* dq2, float* d_rbf,
                              int32_t accumulate_rbf, pamnet_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Input-embedding layers (128 outputs, K = 16 / 18 / 42 inputs):
 *   out[r, :] = act( W_{kind[r]} x[r, :] + b_{kind[r]} ),  act = SiLU when `act` != 0, identity otherwise.
 * Replaces mlp_rbf_g / mlp_rbf_l (K=16), mlp_sbf1 / mlp_sbf2 (K=42; `kind[r]` = 0 selects (W0,b0), 1 selects (W1,b1) on
 * the combined triplet/pair row list) and init_linear (K=18, no bias, act=0): models.py:119,185-188, layers/basic.py:19-22.
 * W*: [128, K] row-major (out, in); b* nullable; kind nullable (then only W0/b0 are used).
 * Backward writes dW0/db0 (and dW1/db1 when kind != null) = sums over rows, reduced in a fixed order through `partial`
 * (pamnet_embed_scratch_floats floats); dx [rows, K] (nullable) is only available for kind == null, K == 16.
 * (One-job forms of pamnet_embed_multi_*_f32 below.)
 * ------------------------------------------------------------------------------------------------------------------ */
int pamnet_embed_scratch_floats(int64_t rows, int64_t K, int64_t* floats);
int pamnet_embed_fwd_f32(const float* x, int64_t rows, int64_t K, const int32_t* kind, const float* W0,
                         const float* b0, const float* W1, const float* b1, int32_t act, float* out,
                         pamnet_stream_t stream);
int pamnet_embed_bwd_f32(const float* x, int64_t rows, int64_t K, const int32_t* kind, const float* W0,
                         const float* b0, const float* W1, const float* b1, int32_t act, const float* gout,
                         float* dW0, float* db0, float* dW1, float* db1, float* dx, float* partial,
                         pamnet_stream_t stream);

/* All input embeddings of a PAMNet forward (models.py:107/119/140, 185-188) in ONE launch, their backward in TWO (main +
 * fixed-order reduce): up to 4 embedding layers plus the rows of the atom-type table.
 * A job with `dist` != null is a K = 16 layer whose input rows are the Bessel basis of those distances
 * (layers/basic.py:59-76: u(d/c) sin(freq_n d/c), u = the p = 5 envelope): the rows are formed while staging, so neither
 * the [rows, 16] basis nor -- in the backward, where d freq is accumulated in the same pass -- its gradient ever exists.
 * Forward reads x|dist, kind, W*, b*, writes out.  Backward reads the same plus gout and OVERWRITES dW0/db0 (dW1/db1
 * with `kind`), dfreq (with `dist`), dx (nullable; plain K = 16 layer only) through `partial`
 * (pamnet_embed_scratch_floats(rows, K) floats per job).  Unused pointers are null. */
typedef struct pamnet_embed_job {
    const float* x;          /* [rows, K] input rows, or null with `dist` */
    const float* dist;       /* [rows] distances, or null */
    const float* freq;       /* [16] Bessel frequencies (with `dist`) */
    float cutoff;            /* c (with `dist`) */
    int32_t K;               /* 16 | 18 | 42 */
    int32_t act;             /* != 0: SiLU */
    int64_t rows;
    const int32_t* kind;     /* [rows] 0 -> (W0, b0), 1 -> (W1, b1); nullable */
    const float *W0, *b0, *W1, *b1;
    float* out;              /* forward: [rows, 128] */
    const float* gout;       /* backward: d out [rows, 128] */
    float *dW0, *db0, *dW1, *db1, *dfreq, *dx;
    float* partial;
} pamnet_embed_job;
/* rows of the atom-type table: forward out[r, :] = table[idx[r], :] (zeros for idx outside [0, n_types)); backward
 * dtable[t, :] = sum_{r: idx[r] = t} g[r, :] through `scratch` (pamnet_reduce_scratch_bytes).  Width 128. */
typedef struct pamnet_type_rows_job {
    const float* table;      /* [n_types, 128] */
    const int32_t* idx;      /* [n] */
    int64_t n;
    int64_t n_types;         /* <= 8 */
    float* out;              /* forward [n, 128] */
    const float* g;          /* backward [n, 128] */
    void* scratch;           /* backward */
    float* dtable;           /* backward [n_types, 128] */
} pamnet_type_rows_job;
int pamnet_embed_multi_fwd_f32(const pamnet_embed_job* jobs, int32_t n_jobs, const pamnet_type_rows_job* types,
                               pamnet_stream_t stream);
int pamnet_embed_multi_bwd_f32(const pamnet_embed_job* jobs, int32_t n_jobs, const pamnet_type_rows_job* types,
                               pamnet_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Layer-stack engine: the n_layer x (global, local) loop of PAMNet.forward (models.py:196-204) in ONE call per
 * direction (dim = 128).  Host-side C++ enqueues ~10 (fwd) / ~20 (bwd) fused launches per layer pair on `stream`.
 *   sizes      : {n, e_g, e_l, tp}
 *   graph_idx  : 18 device index arrays {g_ptr, g_row, g_col, gT_ptr, gT_perm, l_ptr, l_row, l_col, lT_ptr, lT_perm,
 *                tp_ptr, tp_row, tp_col, tpT_ptr, tpT_perm, cuts, tpT_edge, tpT_node}  (the *T_* entries are only read by
 *                the backward; cuts: nullable -- the work split of pamnet_seg_cuts_i32 made with the graph, else computed per
 *                call; tpT_edge / tpT_node: nullable pair -- pamnet_triplet_transpose_aux_i32; the narrow-width engine reads
 *                the first 15)
 *   gparams    : n_layer x 28 device pointers  {mlp_x1.W, .b, mlp_m.W [128,384], .b, W_edge_attr.W, tail W[10], b[10],
 *                W_out.weight, W_out.bias, W}
 *   lparams    : n_layer x 35 device pointers  {mlp_x1.W, .b, mlp_m_ji.W, .b, mlp_m_kj.W, .b, mlp_sbf.0.W, .b,
 *                mlp_sbf.1.W, .b, lin_rbf.W, lin_rbf_out.W, tail ...}
 *   saved/temp : caller-owned arenas sized by pamnet_stack_workspace (floats); `saved` must survive until the backward
 *   outs/atts  : [2*n_layer, n] rows ordered (global_0, local_0, global_1, ...)
 * wpack (nullable, both directions): scratch of pamnet_stack_pack_floats(n_layer) floats; when given, the call first
 * re-packs the node chains' weight matrices into fragment-ordered images (pamnet_pack_weights_f32, one launch) and the
 * chains read those.  Contents need not survive the call.
 * Forward, save_for_backward = 0 (inference): tensors only the backward reads are not written (the single kernels take
 * null for those outputs: z / ea, z_ji / z_kj / q2, z1 / z2, Z / R, Zx1); `saved` still holds the forward's own
 * intermediates and the per-layer node features.
 * Forward, optional fork: `aux_stream` + `aux_events` (n_layer + 1 hipEvent_t handles, both nullable): the per-layer
 * triplet/pair MLPs (independent of the node features) are enqueued on aux_stream up front and joined by event where
 * each layer consumes them; event 0 marks the inputs ready on `stream`.
 * Backward: ggrads / lgrads are gradient buffers laid out like the parameter tables (written, not accumulated);
 * d_x0, d_eg, d_rbf, d_sbf are written.  `layer_done` (nullable): n_layer hipEvent_t handles; event k is recorded on
 * `stream` once every gradient of layer pair k (global_layer.k, local_layer.k) has been enqueued -- the backward runs
 * k = n_layer-1 .. 0, so a data-parallel caller can start reducing the last layers' gradients on another stream while
 * the earlier layers are still being differentiated.
 * ------------------------------------------------------------------------------------------------------------------ */
int pamnet_stack_workspace(int64_t n, int64_t eg, int64_t el, int64_t tp, int64_t n_layer, int64_t* saved_floats,
                           int64_t* temp_floats_out);
int pamnet_stack_pack_floats(int64_t n_layer, int64_t* floats);
int pamnet_stack_layout(int64_t n, int64_t eg, int64_t el, int64_t tp, int64_t* layout);
int pamnet_stack_fwd_f32(const int64_t* sizes, const int32_t* const* graph_idx, int64_t n_layer, const float* x0,
                         const float* e_g, const float* rbf_e, const float* e_sbf, const float* const* gparams,
                         const float* const* lparams, float* saved, float* temp, float* outs, float* atts,
                         int32_t save_for_backward, float* wpack, pamnet_stream_t aux_stream, void* const* aux_events,
                         pamnet_stream_t stream);
int pamnet_stack_bwd_f32(const int64_t* sizes, const int32_t* const* graph_idx, int64_t n_layer, const float* x0,
                         const float* e_g, const float* rbf_e, const float* e_sbf, const float* const* gparams,
                         const float* const* lparams, const float* saved, float* temp, const float* d_outs,
                         const float* d_atts, float* const* ggrads, float* const* lgrads, float* d_x0, float* d_eg,
                         float* d_rbf, float* d_sbf, float* wpack, void* const* layer_done, pamnet_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Small whole-buffer reductions of the training step (csrc/reduce.hip), fixed summation order.
 *   pamnet_sumsq_partials_f32 : partials[0:256] (fp64) = sums of squares of 256 contiguous slices of g[0:n]; the L2 norm
 *                               (clip_grad_norm_, main_qm9.py:111) is finished inside pamnet_adam_ema_norm_f32
 *   pamnet_l1_loss_f32        : loss[0] = mean|out - y|; d_out[i] = grad_scale * sign(out[i] - y[i]) / n  (main_qm9.py:108)
 *   pamnet_mse_loss_f32       : loss[0] = mean (out - y)^2; d_out[i] = grad_scale * 2 (out[i] - y[i]) / n  (main_pdbbind.py:93)
 *   pamnet_smooth_l1_loss_f32 : loss[0] = mean h(out - y), h(d) = d^2/2 if |d| < 1 else |d| - 1/2 (beta = 1);
 *                               d_out[i] = grad_scale * clamp(out[i] - y[i], -1, 1) / n       (main_rna_puzzles.py:92)
 *                               (all three: one launch, d_out nullable)
 *   pamnet_type_rows_grad_f32 : out[t,:] = sum_{r: idx[r] = t} g[r,:], t < n_types <= 8   (gradient of embeddings[x],
 *                               models.py:107,140); scratch: pamnet_reduce_scratch_bytes bytes of device memory
 * ------------------------------------------------------------------------------------------------------------------ */
int pamnet_reduce_scratch_bytes(int64_t* bytes);
/* dst[0:n] = src[0:n], 16 bytes per lane, non-temporal (n % 4 == 0): the memory-system calibration beside bench.py's roofline */
int pamnet_stream_copy_f32(const float* src, float* dst, int64_t n, pamnet_stream_t stream);
int pamnet_sumsq_partials_f32(const float* g, int64_t n, double* partials, pamnet_stream_t stream);
int pamnet_l1_loss_f32(const float* out, const float* y, int64_t n, float grad_scale, float* loss, float* d_out,
                       pamnet_stream_t stream);
int pamnet_mse_loss_f32(const float* out, const float* y, int64_t n, float grad_scale, float* loss, float* d_out,
                        pamnet_stream_t stream);
int pamnet_smooth_l1_loss_f32(const float* out, const float* y, int64_t n, float grad_scale, float* loss, float* d_out,
                              pamnet_stream_t stream);
int pamnet_type_rows_grad_f32(const float* g, const int32_t* idx, int64_t n, int64_t n_types, int64_t d, void* scratch,
                              float* out, pamnet_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Optimiser tail of the reference loop on flat fp32 buffers, one pass (main_qm9.py:111-112,116; utils/ema.py:13-20):
 *   g *= min(1, max_norm / (*grad_norm + 1e-6))                      clip_grad_norm_ (grad_norm: device scalar, nullable)
 *   Adam(lr, betas, eps, weight_decay, amsgrad=False), update number `step_count` >= 1
 *   shadow = ema_decay * shadow + (1 - ema_decay) * p_new            (shadow nullable: no EMA -- the loops of
 *                                                                     main_pdbbind.py:88-95, main_rna_puzzles.py:86-93)
 *   g = 0 when zero_grad != 0 (the next step's zero_grad, main_qm9.py:105)
 * n % 4 == 0; buffers 16-byte aligned.
 * ------------------------------------------------------------------------------------------------------------------ */
int pamnet_adam_ema_f32(float* p, float* g, float* m, float* v, float* shadow, int64_t n, float lr, float beta1,
                        float beta2, float eps, float weight_decay, int64_t step_count, float ema_decay,
                        const float* grad_norm, float max_norm, int32_t zero_grad, pamnet_stream_t stream);
/* same update, gradient norm = sqrt(sum of the 256 fp64 partials of pamnet_sumsq_partials_f32) added inside the kernel;
 * norm_out[0] (nullable) receives the pre-clip norm */
int pamnet_adam_ema_norm_f32(float* p, float* g, float* m, float* v, float* shadow, int64_t n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int64_t step_count, float ema_decay,
                             const double* sumsq_partials, float* norm_out, float max_norm, int32_t zero_grad,
                             pamnet_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Narrow widths (d = 16 / 32 / 64: the reference's RNA configurations, inference_rna_puzzles.py:29-30,
 * main_rna_puzzles.py:52-53).  Row-wise kernels, one wavefront per 16-row tile; backward kernels recompute the forward
 * pre-activations and return weight gradients reduced in fixed order.  `partial` scratch: *blocks (from
 * pamnet_narrow_blocks(rows, &blocks)) rows of the stride given per call.
 * ------------------------------------------------------------------------------------------------------------------ */
int pamnet_narrow_blocks(int64_t rows, int64_t* blocks /* host */);

/* Global message (layers/global_message_passing.py:52-53, W_m split into node / edge blocks):
 *   msg[q] = SiLU(P[tgt[q], :d] + P[src[q], d:] + e[q] We^T + b) * (e[q] Wea^T);  P is [N, 2d], We/Wea [d, d] with row
 *   strides ldwe / ldwea (multiples of 4).  The caller segment-sums msg over tgt. */
int pamnet_narrow_global_fwd_f32(const float* e, int64_t m, int64_t d, const int32_t* tgt, const int32_t* src,
                                 const float* P, const float* We, int64_t ldwe, const float* bias, const float* Wea,
                                 int64_t ldwea, float* msg, pamnet_stream_t stream);
/* dagg [N, d] = gradient of the aggregated message (d msg[q] = dagg[tgt[q]]).  Outputs: dz [m, d] (the caller
 * segment-sums it over tgt and over src into dP), de [m, d], dWe/dWea [d, d] dense, db [d].
 * partial: pamnet_narrow_blocks(m) x (2 d^2 + d) floats. */
int pamnet_narrow_global_bwd_f32(const float* e, int64_t m, int64_t d, const int32_t* tgt, const int32_t* src,
                                 const float* P, const float* We, int64_t ldwe, const float* bias, const float* Wea,
                                 int64_t ldwea, const float* dagg, float* dz, float* de, float* partial, float* dWe,
                                 float* dWea, float* db, pamnet_stream_t stream);

/* y = SiLU(W2 SiLU(W1 x + b1) + b2) on [m, d] rows (mlp_sbf, layers/local_message_passing.py:24,49); dense [d, d]. */
/* res_x != 0 adds x (the Res block of layers/basic.py:25-33); `res` (optional, [m, d]) adds one more residual row. */
int pamnet_narrow_mlp2_fwd_f32(const float* x, int64_t m, int64_t d, const float* W1, const float* b1, const float* W2,
                               const float* b2, int32_t res_x, const float* res, float* y, pamnet_stream_t stream);
/* dx optional (null: not needed; with res_x it includes the + dy of the skip); dW [2, d, d], db [2, d];
 * partial: blocks x (2 d^2 + 2 d) floats. */
int pamnet_narrow_mlp2_bwd_f32(const float* x, int64_t m, int64_t d, const float* W1, const float* b1, const float* W2,
                               const float* b2, const float* dy, int32_t res_x, float* dx, float* partial, float* dW,
                               float* db, pamnet_stream_t stream);

/* One dense layer y = act(x W^T + b) (layers/basic.py:19-22): W is a [d, d] block with row stride ldw (a column slice of
 * the [d, 3d] message weights included), b optional, act 1 = SiLU / 0 = identity, y with row stride ldy (blocks of a
 * [m, n d] projection).  Backward: dy with row stride lddy; dx optional, accumulate != 0 adds into dx; dW [d, d], db [d]
 * (null without bias); partial: blocks x (d^2 + d) floats. */
int pamnet_narrow_linear_fwd_f32(const float* x, int64_t m, int64_t d, const float* W, int64_t ldw, const float* b,
                                 int32_t act, float* y, int64_t ldy, pamnet_stream_t stream);
int pamnet_narrow_linear_bwd_f32(const float* x, int64_t m, int64_t d, const float* W, int64_t ldw, const float* b,
                                 int32_t act, const float* dy, int64_t lddy, float* dx, int32_t accumulate,
                                 float* partial, float* dW, float* db, pamnet_stream_t stream);

/* Layer heads (layers/global_message_passing.py:47-50): out[n] = o[n] . w_out + b_out, att[n] = o[n] . w_att.
 * Backward: d_o [m, d] and dvec [2 d + 1] = [d w_out | d w_att | d b_out]; partial: blocks x (2 d + 1) floats. */
int pamnet_narrow_heads_fwd_f32(const float* o, int64_t m, int64_t d, const float* w_out, const float* b_out,
                                const float* w_att, float* out, float* att, pamnet_stream_t stream);
int pamnet_narrow_heads_bwd_f32(const float* o, int64_t m, int64_t d, const float* w_out, const float* w_att,
                                const float* g_out, const float* g_att, float* d_o, float* partial, float* dvec,
                                pamnet_stream_t stream);

/* Local-edge gates (layers/local_message_passing.py:46-48 with the split message weights): for local edge q = (src -> tgt)
 *   m_ji[q] = SiLU(P[tgt, 0:d] + P[src, 2d:3d] + Q[q, 0:d] + b_ji)
 *   m_nb[q] = SiLU(P[tgt, d:2d] + P[src, 3d:4d] + Q[q, d:2d] + b_kj) * Q[q, 2d:3d]
 * P [N, 4d], Q [m, 4d].  Backward: dz [m, 2d] (segment-summed by the caller into dP, column-summed into the biases)
 * and dQ [m, 4d] (last block zero). */
int pamnet_narrow_local_gate_fwd_f32(const float* P, const float* Q, const int32_t* tgt, const int32_t* src,
                                     const float* b_ji, const float* b_kj, int64_t m, int64_t d, float* m_ji,
                                     float* m_nb, pamnet_stream_t stream);
int pamnet_narrow_local_gate_bwd_f32(const float* P, const float* Q, const int32_t* tgt, const int32_t* src,
                                     const float* b_ji, const float* b_kj, int64_t m, int64_t d, const float* g_ji,
                                     const float* g_nb, float* dz, float* dQ, pamnet_stream_t stream);

/* Edge-embedding MLPs (models.py:185-188): y = SiLU(W f + b), f [m, k], k = 16 or 42, W [d, k] dense.  With `kind`
 * [m] rows of kind 0 use (Wa, ba) and rows of kind != 0 use (Wb, bb); kind null: one set. */
int pamnet_narrow_embed_fwd_f32(const float* F, int64_t m, int64_t k, int64_t d, const int32_t* kind, const float* Wa,
                                const float* ba, const float* Wb, const float* bb, float* y, pamnet_stream_t stream);
/* dW [sets, d, k], db [sets, d]; df [m, 16] optional (k = 16, one set: the Bessel frequencies are trainable).
 * partial: blocks x sets (d * kp + d) floats, kp = 16 or 48. */
int pamnet_narrow_embed_bwd_f32(const float* F, int64_t m, int64_t k, int64_t d, const int32_t* kind, const float* Wa,
                                const float* ba, const float* Wb, const float* bb, const float* dy, float* df,
                                float* partial, float* dW, float* db, pamnet_stream_t stream);
/* Forward of the 16-wide embedding on Bessel rows formed inside the kernel (BesselBasisLayer, layers/basic.py:59-76, fused
 * into models.py:185-186): dist [m], freq [16], cutoff as for pamnet_rbf_fwd_f32.  The same floats as
 * pamnet_rbf_fwd_f32 followed by pamnet_narrow_embed_fwd_f32 (k = 16); the [m, 16] basis tensor never exists. */
int pamnet_narrow_embed_rbf_fwd_f32(const float* dist, const float* freq, float cutoff, int64_t m, int64_t d,
                                    const float* Wa, const float* ba, float* y, pamnet_stream_t stream);
/* ... and its backward: dW [d, 16]; db_dfreq [d + 16] = the bias gradient followed by the gradient of the 16 frequencies;
 * partial: blocks x (d * 16 + d + 16) floats (blocks = pamnet_narrow_blocks).  Neither the rows nor their gradient exist. */
int pamnet_narrow_embed_rbf_bwd_f32(const float* dist, const float* freq, float cutoff, int64_t m, int64_t d,
                                    const float* Wa, const float* ba, const float* dy, float* partial, float* dW,
                                    float* db_dfreq, pamnet_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Narrow-width layer-stack engine (csrc/narrow_engine.hip): the n_layer x (global, local) loop of PAMNet.forward
 * (models.py:196-204) at d = 16 / 32 / 64 in ONE call per direction -- the counterpart of pamnet_stack_*_f32.
 * sizes, graph_idx, gparams / lparams (n_layer x 28 / 35 device pointers; mlp_m etc. are [d, 3d]), ggrads / lgrads,
 * outs / atts ([2 n_layer, n]), saved / temp (caller-owned arenas sized by pamnet_narrow_stack_workspace; `saved` must
 * survive until the backward) and layer_done (nullable hipEvent_t handles) exactly as documented for pamnet_stack_*_f32.
 * The node-side chains of a layer are single launches (pre: mlp_x1 + projections; tail: mlp_x2 .. heads); weight
 * gradients are written (not accumulated) in fixed summation order; d_x0, d_eg, d_rbf, d_sbf are written.
 * The backward needs n, e_g, e_l, tp > 0.  pamnet_narrow_stack_layout: layout[0] = floats per layer pair in `saved`,
 * layout[1] / [2] = offset of the global / local layer's node output [n, d] within a pair.
 * ------------------------------------------------------------------------------------------------------------------ */
int pamnet_narrow_stack_workspace(int64_t n, int64_t eg, int64_t el, int64_t tp, int64_t n_layer, int64_t d,
                                  int64_t* saved_floats, int64_t* temp_floats);
int pamnet_narrow_stack_layout(int64_t n, int64_t eg, int64_t el, int64_t tp, int64_t d, int64_t* layout);
int pamnet_narrow_stack_fwd_f32(const int64_t* sizes, const int32_t* const* graph_idx, int64_t n_layer, int64_t d,
                                const float* x0, const float* e_g, const float* rbf_e, const float* e_sbf,
                                const float* const* gparams, const float* const* lparams, float* saved, float* temp,
                                float* outs, float* atts, pamnet_stream_t stream);
int pamnet_narrow_stack_bwd_f32(const int64_t* sizes, const int32_t* const* graph_idx, int64_t n_layer, int64_t d,
                                const float* x0, const float* e_g, const float* rbf_e, const float* e_sbf,
                                const float* const* gparams, const float* const* lparams, const float* saved, float* temp,
                                const float* d_outs, const float* d_atts, float* const* ggrads, float* const* lgrads,
                                float* d_x0, float* d_eg, float* d_rbf, float* d_sbf, void* const* layer_done,
                                pamnet_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Dense layers of any width (csrc/dense.hip): Sequential(Linear, SiLU) of layers/basic.py:19-22 and the bare F.linear
 * projections, for hidden sizes above 128 (models.py:25) and input widths no engine is built for (models.py:187-188 with
 * a non-default num_spherical * num_radial; models.py:119 at such a dim).  fp32-accurate GEMMs on the bf16 matrix pipe
 * (three exact bf16 pieces per operand, six products); no transposed copies, no library GEMM.
 *   fwd:  Z = X W^T + bias (X [n][k] row stride ldx, W [m][k] row stride ldw, bias [m] nullable);
 *         Y [n][m] = act ? SiLU(Z) : Z;  Z [n][m] nullable (the backward of an activated layer needs it).
 *   bwd:  dZ = act ? G * SiLU'(Z) : G (G, Z [n][m]; formed while staging, never stored);
 *         dX [n][k] (nullable) = dZ W;  dW [m][k] (nullable) = dZ^T X;  db [m] (nullable, with dW) = column sums of dZ.
 *         The row sum of dW is split over the grid and reduced in a fixed order (deterministic); `partial`:
 *         pamnet_dense_scratch_floats(n, k, m) floats of caller-owned scratch, required with dW.
 * ------------------------------------------------------------------------------------------------------------------ */
int pamnet_dense_fwd_f32(const float* X, int64_t ldx, const float* W, int64_t ldw, const float* bias, int64_t n,
                         int64_t k, int64_t m, int32_t act, float* Z, float* Y, pamnet_stream_t stream);
int pamnet_dense_scratch_floats(int64_t n, int64_t k, int64_t m, int64_t* floats);
int pamnet_dense_bwd_f32(const float* G, const float* Z, const float* X, int64_t ldx, const float* W, int64_t ldw,
                         int64_t n, int64_t k, int64_t m, int32_t act, float* dX, float* dW, float* db, float* partial,
                         pamnet_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PAMNET_HIP_H */
