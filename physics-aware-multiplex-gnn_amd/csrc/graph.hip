// Device-side graph construction for PAMNet.forward (models.py:62-98, 104-157):
//   * exclusive scan + stable counting sort into CSR (stands in for torch_sparse.SparseTensor, models.py:71-73)
//   * batched fixed-radius and k-NN neighbour search (stands in for torch_cluster.radius / knn, models.py:110,128,143)
//   * CSR row filter by distance (the `dist <= cutoff` masks, models.py:131-134, 147-156)
//   * triplet (k->j->i) / pair (j,j'->i) enumeration with their angles (models.py:68-98, 165-177)
// Integer work: bit-exact against the oracle.  Everything is two-pass (count -> caller scans -> fill), atomics are
// used only for histogram counts and slot claiming whose result is re-sorted, so outputs are deterministic.
#include <stdlib.h>

#include "common.h"
#include "geom_core.h"

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_CHUNK = SCAN_THREADS * SCAN_ITEMS;   // 4096

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int u = __shfl_up(v, o, 64);
        if (lane >= o) v += u;
    }
    return v;
}

// inclusive scan of one 4096-element chunk; out[i+1] = chunk-local inclusive sum, sums[b] = chunk total
__global__ __launch_bounds__(SCAN_THREADS) void scan_chunk_kernel(const int32_t* __restrict__ in,
                                                                  int32_t* __restrict__ out, int64_t n,
                                                                  int32_t* __restrict__ sums) {
    __shared__ int wave_tot[SCAN_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)tid * SCAN_ITEMS;
    int v[SCAN_ITEMS];
    int run = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        const int64_t g = base + i;
        run += (g < n) ? in[g] : 0;
        v[i] = run;
    }
    const int incl = wave_incl_scan(run, lane);
    if (lane == 63) wave_tot[w] = incl;
    __syncthreads();
    int off = incl - run;
    for (int k = 0; k < w; ++k) off += wave_tot[k];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        const int64_t g = base + i;
        if (g < n) out[g + 1] = v[i] + off;
    }
    if (tid == SCAN_THREADS - 1) sums[blockIdx.x] = off + run;
}

// single block: exclusive scan of the chunk totals in place
__global__ __launch_bounds__(SCAN_THREADS) void scan_sums_kernel(int32_t* __restrict__ sums, int64_t nb) {
    __shared__ int wave_tot[SCAN_THREADS / 64];
    __shared__ int carry_s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < nb; base += SCAN_THREADS) {
        const int64_t g = base + tid;
        const int x = (g < nb) ? sums[g] : 0;
        const int incl = wave_incl_scan(x, lane);
        if (lane == 63) wave_tot[w] = incl;
        __syncthreads();
        int off = carry_s;
        for (int k = 0; k < w; ++k) off += wave_tot[k];
        if (g < nb) sums[g] = off + incl - x;
        __syncthreads();
        if (tid == SCAN_THREADS - 1) carry_s = off + incl;
        __syncthreads();
    }
}

// RAW: `sums` still holds the chunk totals (no scan_sums launch): the 256 outputs of a workgroup lie in one chunk, whose
// offset it adds up itself -- a few hundred values at most (the caller takes the three-launch form beyond that).
template <bool RAW>
__global__ __launch_bounds__(256) void scan_add_kernel(int32_t* __restrict__ out, int64_t n,
                                                       const int32_t* __restrict__ sums) {
    __shared__ int part[4];
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int add;
    if (RAW) {
        const int chunk = (int)(((int64_t)blockIdx.x * 256) / SCAN_CHUNK);
        int v = 0;
        for (int c = threadIdx.x; c < chunk; c += 256) v += sums[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
        __syncthreads();
        add = part[0] + part[1] + part[2] + part[3];
    } else {
        add = g < n ? sums[g / SCAN_CHUNK] : 0;
    }
    if (g == 0) out[0] = 0;
    if (g < n) out[g + 1] += add;
}

// Runs of equal keys inside a wavefront (the common case: keys emitted row by row) are folded into one atomic by the
// run's first lane; `head_of` / `run_len` describe the run a lane belongs to.
struct Run { int head; int len; };
__device__ __forceinline__ Run wave_run(int key, bool valid, int lane) {
    const int prev = __shfl_up(key, 1, 64);
    const bool head = valid && (lane == 0 || prev != key);
    const unsigned long long heads = __ballot(head);
    const unsigned long long live = __ballot(valid);
    const unsigned long long below = heads & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
    Run r;
    r.head = 63 - __builtin_clzll(below | 1ull);
    const unsigned long long above = (lane == 63) ? 0ull : (heads >> (lane + 1));
    const int nlive = __builtin_popcountll(live);            // valid lanes are a prefix of the wavefront
    const int next = above ? (lane + 1 + __builtin_ctzll(above)) : nlive;
    r.len = next - lane;                                     // meaningful on head lanes only
    return r;
}

// count[key] += multiplicity; *unsorted = 1 if the key sequence ever decreases (then the sort below is needed)
// Keys outside [0, rows) are skipped here and in claim_kernel (no out-of-bounds write); the Python side validates the
// index inputs on the device and raises IndexError with the batch's size round trip (graph._input_flag).
// `arrival` (nullable, m ints): the entry's place among its row's entries in the order the atomics landed -- what lets the
// claim below hand out slots WITHOUT a second round of atomics (867 k of them per pass at the RNA batch: 45 us).
__global__ __launch_bounds__(256) void hist_kernel(const int32_t* __restrict__ keys, int64_t m, int64_t rows,
                                                   int32_t* __restrict__ count, int32_t* __restrict__ unsorted,
                                                   int32_t* __restrict__ arrival) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool valid = k < m;
    const int key = valid ? keys[k] : -1;
    const Run r = wave_run(key, valid, lane);
    int base = 0;
    if (valid && r.head == lane && (uint64_t)key < (uint64_t)rows) base = atomicAdd(&count[key], r.len);
    if (arrival) {
        base = __shfl(base, r.head, 64);
        if (valid) arrival[k] = base + (lane - r.head);
    }
    if (valid && k + 1 < m && keys[k + 1] < key) *unsorted = 1;
}

// The same for long key lists over few enough rows that ONE WORKGROUP HOLDS EVERY COUNTER IN LDS (rows * 4 B <= 128 KB): a
// workgroup counts a contiguous chunk of keys with LDS atomics (whose return value is the entry's arrival inside the
// chunk), adds its non-zero counters to the global ones with one returning atomic per DISTINCT key of the chunk, and hands
// every entry arrival = that base + its place in the chunk.  The keys of a chunk are the neighbours of ~80 consecutive
// nodes, a few hundred distinct ones: 867 k global atomics on 17.7 k addresses become ~100 k at the RNA batch (45 -> ~12 us).
// Any order inside a row will do (sort_rows_kernel ranks the entries), so the tables are bit for bit the plain kernel's.
constexpr int HIST_LDS_THREADS = 1024;
constexpr int64_t HIST_LDS_MAX_ROWS = 32768;                  // 128 KB of counters
constexpr int64_t HIST_LDS_MIN_KEYS = 131072;
constexpr int HIST_LDS_PER_THREAD = 4;                        // keys per thread: chunks of 4096
__global__ __launch_bounds__(HIST_LDS_THREADS) void hist_lds_kernel(const int32_t* __restrict__ keys, int64_t m, int rows,
                                                                    int32_t* __restrict__ count,
                                                                    int32_t* __restrict__ unsorted,
                                                                    int32_t* __restrict__ arrival) {
    extern __shared__ int32_t hcnt[];
    const int lane = threadIdx.x & 63;
    for (int r = threadIdx.x; r < rows; r += HIST_LDS_THREADS) hcnt[r] = 0;
    __syncthreads();
    const int64_t k0 = (int64_t)blockIdx.x * (HIST_LDS_THREADS * HIST_LDS_PER_THREAD) + threadIdx.x;
    int key[HIST_LDS_PER_THREAD], nextk[HIST_LDS_PER_THREAD], place[HIST_LDS_PER_THREAD];
#pragma unroll
    for (int i = 0; i < HIST_LDS_PER_THREAD; ++i) {
        const int64_t k = k0 + (int64_t)i * HIST_LDS_THREADS;
        key[i] = k < m ? keys[k] : -1;
        nextk[i] = k + 1 < m ? keys[k + 1] : 0x7fffffff;
    }
    bool dec = false;
#pragma unroll
    for (int i = 0; i < HIST_LDS_PER_THREAD; ++i) {
        const int64_t k = k0 + (int64_t)i * HIST_LDS_THREADS;
        const bool valid = k < m;
        const Run r = wave_run(key[i], valid, lane);
        int base = 0;
        if (valid && r.head == lane && (unsigned)key[i] < (unsigned)rows) base = atomicAdd(&hcnt[key[i]], r.len);
        base = __shfl(base, r.head, 64);
        place[i] = base + (lane - r.head);
        dec |= valid && nextk[i] < key[i];
    }
    if (dec) *unsorted = 1;
    __syncthreads();
    for (int r = threadIdx.x; r < rows; r += HIST_LDS_THREADS) {
        const int c = hcnt[r];
        if (c > 0) hcnt[r] = atomicAdd(&count[r], c);
    }
    if (!arrival) return;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < HIST_LDS_PER_THREAD; ++i) {
        const int64_t k = k0 + (int64_t)i * HIST_LDS_THREADS;
        if (k < m) arrival[k] = ((unsigned)key[i] < (unsigned)rows ? hcnt[key[i]] : 0) + place[i];
    }
}

// sorted key sequence: the stable permutation is the identity.  Otherwise every run claims a block of slots.
__global__ __launch_bounds__(256) void claim_kernel(const int32_t* __restrict__ keys, int64_t m, int64_t rows,
                                                    const int32_t* __restrict__ ptr, int32_t* __restrict__ cursor,
                                                    int32_t* __restrict__ perm_tmp,
                                                    int32_t* __restrict__ perm, const int32_t* __restrict__ unsorted) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (*unsorted == 0) {
        if (k < m) perm[k] = (int32_t)k;
        return;
    }
    // the histogram pass noted every entry's arrival order inside its row (in `perm`, which is free until the ranking pass
    // writes it): slot = the row's first slot + that (any order inside a row will do, sort_rows_kernel ranks the entries)
    if (k >= m) return;
    const int key = keys[k];
    if ((uint64_t)key < (uint64_t)rows) perm_tmp[ptr[key] + perm[k]] = (int32_t)k;
}

// one thread per claimed entry: its rank among the (distinct) entries of its row -> ascending perm inside each row.
// Cost sum(len^2) L2-resident loads; rows here are short (node degrees), long rows (e.g. 5 embedding rows) stay parallel.
__global__ __launch_bounds__(256) void sort_rows_kernel(const int32_t* __restrict__ keys,
                                                        const int32_t* __restrict__ ptr,
                                                        const int32_t* __restrict__ perm_tmp,
                                                        int32_t* __restrict__ perm, int64_t m, int64_t rows,
                                                        const int32_t* __restrict__ unsorted) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (*unsorted == 0 || q >= m) return;
    const int v = perm_tmp[q];
    if ((uint64_t)v >= (uint64_t)m) return;                  // slot never claimed (only with out-of-range keys)
    const int r = keys[v];
    if ((uint64_t)r >= (uint64_t)rows) return;
    const int beg = ptr[r], end = ptr[r + 1];
    int rank = 0;
    int t = beg;
    for (; t + 8 <= end; t += 8) {                           // eight independent loads in flight
        int a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = perm_tmp[t + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) rank += (a[u] < v) ? 1 : 0;
    }
    for (; t < end; ++t) rank += (perm_tmp[t] < v) ? 1 : 0;
    perm[beg + rank] = v;
}

// ---- small inputs: ONE single-workgroup launch ------------------------------------------------------------------------
// At molecule-batch sizes (a few thousand rows, tens of thousands of keys) the multi-launch forms above are 3 / 9 launches
// of up to ~130 workgroups each, issued on the input pipeline's side stream while the training step runs: every one of
// them takes CUs away from main-stream kernels that were balanced for exactly 256 (measured: 125 us of a 2.50 ms step,
// tools/noprefetch_bound.py).  One workgroup does the whole scan / the whole stable counting sort on one CU instead;
// results are identical (same stable order), so the large-input forms remain the reference in the tests.
constexpr int SMALL_THREADS = 1024;
constexpr int SMALL_SCAN_MAX = 24576;         // elements (runs of <= 24 per thread; beyond, the strided run reads lose to three launches: 29 us vs 12 at 65 k)
constexpr int SMALL_CSR_ROWS = 12288;         // rows (counters live in LDS)
constexpr int SMALL_CSR_KEYS = 1 << 13;      // beyond this the ranking pass of ONE workgroup (sum of squared row lengths, L2 loads)
                                              // takes longer than the nine small launches: 65 us against ~36 at 33 k keys

// block-wide exclusive scan step: returns the exclusive prefix of x over the workgroup and its total (wt: 16 ints of LDS)
__device__ __forceinline__ int block_excl_scan(int x, int* wt, int& total) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int incl = wave_incl_scan(x, lane);
    __syncthreads();                                          // wt may still be read from the previous call
    if (lane == 63) wt[w] = incl;
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < SMALL_THREADS / 64; ++k) {
        const int v = wt[k];
        off += (k < w) ? v : 0;
        tot += v;
    }
    total = tot;
    return off + incl - x;
}

// Every thread owns a contiguous run of ceil(n / 1024) elements: one pass to sum the run, ONE workgroup scan over the
// 1 024 run totals, one pass to write the prefixes (the second read of the run comes out of the L1 / L2).  (The earlier
// form scanned 1 024 elements per iteration with a dependent global load and two barriers in each: 15 us at 17.7 k.)
__global__ __launch_bounds__(SMALL_THREADS) void scan_small_kernel(const int32_t* __restrict__ in, int32_t* __restrict__ out,
                                                                   int n, const int32_t* __restrict__ in2 = nullptr,
                                                                   int32_t* __restrict__ out2 = nullptr) {
    __shared__ int wt[SMALL_THREADS / 64];
    if (blockIdx.x == 1) in = in2, out = out2;                // second array of a pair launch (same length)
    const int run = (n + SMALL_THREADS - 1) / SMALL_THREADS;
    const int beg = min((int)threadIdx.x * run, n), end = min(beg + run, n);
    int sum = 0;
    int k = beg;
    for (; k + 8 <= end; k += 8) {                            // eight independent loads in flight
        int a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = in[k + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) sum += a[u];
    }
    for (; k < end; ++k) sum += in[k];
    int total;
    int acc = block_excl_scan(sum, wt, total);
    if (threadIdx.x == 0) out[0] = 0;
    k = beg;
    for (; k + 8 <= end; k += 8) {
        int a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = in[k + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc += a[u];
            out[k + u + 1] = acc;
        }
    }
    for (; k < end; ++k) {
        acc += in[k];
        out[k + 1] = acc;
    }
}

// ptr / perm of pamnet_csr_from_keys_i32 for m <= SMALL_CSR_KEYS keys over rows <= SMALL_CSR_ROWS rows
__global__ __launch_bounds__(SMALL_THREADS) void csr_small_kernel(const int32_t* __restrict__ keys, int m, int rows,
                                                                  int32_t* __restrict__ ptr, int32_t* __restrict__ perm,
                                                                  int32_t* __restrict__ perm_tmp) {
    __shared__ int cnt[SMALL_CSR_ROWS];                       // histogram, then the rows' write cursors
    __shared__ int wt[SMALL_THREADS / 64];
    __shared__ int unsorted;
    const int tid = threadIdx.x;
    for (int r = tid; r < rows; r += SMALL_THREADS) cnt[r] = 0;
    if (tid == 0) unsorted = 0;
    __syncthreads();
    bool dec = false;
    for (int k = tid; k < m; k += SMALL_THREADS) {
        const int key = keys[k];
        if ((unsigned)key < (unsigned)rows) atomicAdd(&cnt[key], 1);
        if (k + 1 < m && keys[k + 1] < key) dec = true;
    }
    if (dec) unsorted = 1;
    __syncthreads();
    // exclusive scan of the histogram -> ptr (global) and the cursors (in place)
    int carry = 0;
    if (tid == 0) ptr[0] = 0;
    for (int base = 0; base < rows; base += SMALL_THREADS) {
        const int r = base + tid;
        const int x = r < rows ? cnt[r] : 0;
        int total;
        const int ex = block_excl_scan(x, wt, total);
        if (r < rows) {
            cnt[r] = carry + ex;
            ptr[r + 1] = carry + ex + x;
        }
        carry += total;
    }
    __syncthreads();
    if (!unsorted) {                                          // sorted keys: the stable permutation is the identity
        for (int k = tid; k < m; k += SMALL_THREADS) perm[k] = k;
        return;
    }
    for (int k = tid; k < m; k += SMALL_THREADS) {            // claim a slot (any order inside a row) ...
        const int key = keys[k];
        if ((unsigned)key < (unsigned)rows) perm_tmp[atomicAdd(&cnt[key], 1)] = k;
    }
    __threadfence_block();
    __syncthreads();
    for (int q = tid; q < carry; q += SMALL_THREADS) {        // ... then rank every entry inside its row: ascending indices
        const int v = perm_tmp[q];
        const int r = keys[v];
        const int beg = ptr[r], end = ptr[r + 1];
        int rank = 0;
        int t = beg;
        for (; t + 8 <= end; t += 8) {                        // eight independent loads in flight
            int a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = perm_tmp[t + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) rank += (a[u] < v) ? 1 : 0;
        }
        for (; t < end; ++t) rank += (perm_tmp[t] < v) ? 1 : 0;
        perm[beg + rank] = v;
    }
}


// max_nb > 0: torch_cluster.radius(..., max_num_neighbors) (models.py:110,128: 1000; :301: 500): a query keeps the first
// max_nb points of its graph within r in ascending index order -- ITSELF INCLUDED in that count (the search returns the
// query too; remove_self_loops comes after, models.py:63) -- and a truncated row raises CAP_BIT in *cap_flag.  A capped
// graph is no longer symmetric; the host then takes the general (counting-sort) transposes.  (Order of a capped search:
// torch_cluster 1.5.x's CUDA kernel walks the candidates by index and stops at the cap; that library is not in the
// reference tree -- parity unpinned, as SURVEY 8c records for every third-party boundary.)
constexpr int CAP_BIT = 64;

template <bool FILL>
__global__ __launch_bounds__(256) void radius_kernel(const float* __restrict__ pos,
                                                     const int32_t* __restrict__ node_graph,
                                                     const int32_t* __restrict__ gptr, int64_t n, float r,
                                                     int32_t* __restrict__ count, const int32_t* __restrict__ ptr,
                                                     int32_t* __restrict__ nbr, float* __restrict__ dist, int64_t cap,
                                                     int32_t* __restrict__ row_of, int max_nb,
                                                     int32_t* __restrict__ cap_flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int g = node_graph[i];
    const int beg = gptr[g], end = gptr[g + 1];
    int c = 0, seen = 0;
    int64_t w = FILL ? ptr[i] : 0;
    for (int j = beg; j < end; ++j) {
        const float d = j == i ? 0.f : dist3(pos, i, j);
        if (j != i && !(d <= r)) continue;
        if (max_nb > 0 && seen >= max_nb) {                    // this hit and everything behind it is cut off
            if (!FILL && cap_flag) atomicOr(cap_flag, CAP_BIT);
            break;
        }
        ++seen;
        if (j == i) continue;
        if (FILL) {
            if (w < cap) {                                 // cap: see pamnet_radius_fill_i32
                nbr[w] = j;
                dist[w] = d;
                if (row_of) row_of[w] = (int32_t)i;
            }
            ++w;
        }
        ++c;
    }
    if (!FILL) count[i] = c;
}

// The same search with one WAVEFRONT per node, for graphs of hundreds of nodes (PDBbind complexes: ~600 atoms): the lanes
// take 64 consecutive candidates, vote, and the survivors keep ascending order through the prefix population count of
// the ballot -- identical output.  (One thread per node walks the whole graph alone: 137 + 169 us per batch of 19 000
// atoms with only 75 workgroups in flight.)
template <bool FILL>
__global__ __launch_bounds__(256) void radius_wave_kernel(const float* __restrict__ pos,
                                                          const int32_t* __restrict__ node_graph,
                                                          const int32_t* __restrict__ gptr, int64_t n, float r,
                                                          int32_t* __restrict__ count, const int32_t* __restrict__ ptr,
                                                          int32_t* __restrict__ nbr, float* __restrict__ dist,
                                                          int64_t cap, int32_t* __restrict__ row_of, int max_nb,
                                                          int32_t* __restrict__ cap_flag) {
    const int lane = threadIdx.x & 63;
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (i >= n) return;
    const int g = node_graph[i];
    const int beg = gptr[g], end = gptr[g + 1];
    int c = 0, seen = 0;
    const int64_t w0 = FILL ? ptr[i] : 0;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int j0 = beg; j0 < end; j0 += 64) {
        const int j = j0 + lane;
        float d = 0.f;
        bool keep = false;
        if (j < end && j != i) {
            d = dist3(pos, i, j);
            keep = d <= r;
        }
        if (max_nb > 0) {                                      // (wave-uniform branch; hits = kept candidates + the query)
            const unsigned long long hits = __ballot(keep || j == i);
            const bool allowed = seen + __builtin_popcountll(hits & below) < max_nb;
            seen += __builtin_popcountll(hits);
            if (seen > max_nb) {                               // wave-uniform
                if (!FILL && cap_flag && lane == 0) atomicOr(cap_flag, CAP_BIT);
                keep = keep && allowed;
            }
        }
        const unsigned long long votes = __ballot(keep);
        if (FILL && keep) {
            const int64_t w = w0 + c + __builtin_popcountll(votes & below);
            if (w < cap) {
                nbr[w] = j;
                dist[w] = d;
                if (row_of) row_of[w] = (int32_t)i;
            }
        }
        c += __builtin_popcountll(votes);
        if (max_nb > 0 && seen >= max_nb && FILL) break;       // nothing behind the cap is written (the count pass keeps
    }                                                          // scanning: a later hit is what raises the flag)
    if (!FILL && lane == 0) count[i] = c;
}

// graphs this large on average take the wavefront-per-node form (QM9 molecules: ~18 atoms -> one thread per node)
constexpr int64_t RADIUS_WAVE_MIN_NODES = 96;

__device__ __forceinline__ bool pair_less(float da, int ja, float db, int jb) {
    return (da < db) || (da == db && ja < jb);
}

// ---- k nearest neighbours (torch_cluster.knn stand-in, models.py:143) ---------------------------------------------------
// One wave per query, candidates = the nodes of the query's graph (self included, as the reference's knn returns it).
// Result: the K smallest (squared distance, index) pairs in ascending order.
//
// Selection path (graphs up to 64 * KNN_CACHE nodes): every lane caches the squared distances of its candidates in
// registers, the K-th smallest distance is found by bisection on the float bit pattern (non-negative floats order like
// unsigned integers; each probe is one compare + ballot count per cached value, no cross-lane traffic), the selected
// candidates (ties broken by index = scan order) are compacted through LDS and sorted by a 64-lane bitonic network.
// Streaming path (larger graphs): running sorted top-K across lanes with serial insertion.
constexpr int KNN_CACHE = 64;

__device__ __forceinline__ float sqdist(const float* __restrict__ pos, float xi, float yi, float zi, int64_t j) {
    const float dx = xi - pos[3 * j], dy = yi - pos[3 * j + 1], dz = zi - pos[3 * j + 2];
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

__device__ __forceinline__ void knn_stream(const float* __restrict__ pos, float xi, float yi, float zi, int beg, int end,
                                           int K, int lane, float& bd, int& bj) {
    bd = INFINITY;
    bj = 0x7fffffff;
    float kd = INFINITY;            // current K-th best
    int kj = 0x7fffffff;
    for (int base = beg; base < end; base += 64) {
        const int j = base + lane;
        const bool valid = j < end;
        const float d = valid ? sqdist(pos, xi, yi, zi, j) : INFINITY;
        unsigned long long mask = __ballot(valid && pair_less(d, j, kd, kj));
        while (mask) {
            const int src = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const float cd = __shfl(d, src, 64);
            const int cj = __shfl(j, src, 64);
            if (!pair_less(cd, cj, kd, kj)) continue;                        // wave-uniform
            const bool lt = (lane < K) && pair_less(bd, bj, cd, cj);
            const int p = __popcll(__ballot(lt));
            const float ud = __shfl_up(bd, 1, 64);
            const int uj = __shfl_up(bj, 1, 64);
            if (lane > p) { bd = ud; bj = uj; }
            else if (lane == p) { bd = cd; bj = cj; }
            kd = __shfl(bd, K - 1, 64);
            kj = __shfl(bj, K - 1, 64);
        }
    }
}

// 64-lane bitonic sort by (distance, index); padding (inf, INT_MAX) sinks to the end
__device__ __forceinline__ void sort64_pairs(float& bd, int& bj, int lane) {
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int st = k >> 1; st > 0; st >>= 1) {
            const float od = __shfl_xor(bd, st, 64);
            const int oj = __shfl_xor(bj, st, 64);
            const bool keep_min = (((lane & k) == 0) == ((lane & st) == 0));
            const bool take = keep_min ? pair_less(od, oj, bd, bj) : pair_less(bd, bj, od, oj);
            if (take) { bd = od; bj = oj; }
        }
    }
}

// K smallest (distance bits, index) pairs out of NC values per lane (padding = 0xffffffff), values in scan order
// (slot-major, lane-minor = ascending candidate index): bisection on the bit pattern for the K-th smallest distance,
// ties by scan order, compaction into sd/sj[64], then a 64-lane bitonic sort.  G = slots per guarded group
// (`slots` = number of slots that can hold values).
template <int NC, int G, bool EXPLICIT_J>
__device__ __forceinline__ void knn_topk(const unsigned (&dv)[NC], const int (&jv)[EXPLICIT_J ? NC : 1], int j0, int slots,
                                         int kk, int lane, float* sd, int* sj, float& bd, int& bj) {
    // Counting stays on the vector unit (per-lane counters, padding never counts); the scalar unit is shared by the
    // four SIMDs of a CU and a ballot + s_bcnt per value made it the bottleneck of the first version (353 us).
    auto wave_count = [&](unsigned bound, bool inclusive) -> int {
        int c_lane = 0;
#pragma unroll
        for (int c0 = 0; c0 < NC; c0 += G) {
            if (c0 < slots) {
#pragma unroll
                for (int c = c0; c < c0 + G; ++c) c_lane += (inclusive ? (dv[c] <= bound) : (dv[c] < bound)) ? 1 : 0;
            }
        }
        int tot = 0;                                         // lane counters are <= 64: seven bit planes
#pragma unroll
        for (int b = 0; b < 7; ++b)
            if ((1 << b) <= NC) tot += __popcll(__ballot((c_lane >> b) & 1)) << b;
        return tot;
    };
    unsigned mn = 0xffffffffu, mx = 0u;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (dv[c] != 0xffffffffu) {
            mn = (dv[c] != 0u && dv[c] < mn) ? dv[c] : mn;       // smallest NON-ZERO distance (see below)
            mx = dv[c] > mx ? dv[c] : mx;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned a = (unsigned)__shfl_xor((int)mn, o, 64), b = (unsigned)__shfl_xor((int)mx, o, 64);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
    }
    // smallest T with #(d <= T) >= kk, searched inside the wave's distance bits.  The query itself is a candidate at
    // distance 0: a search interval that starts at bit pattern 0 spends its first ~6 probes crossing the exponent range
    // below the nearest real neighbour, so it starts at the smallest non-zero distance unless the zeros alone reach kk.
    unsigned lo = (mn != 0xffffffffu && wave_count(0u, true) < kk) ? mn : 0u, hi = mx;
    while (lo < hi) {
        const unsigned mid = lo + ((hi - lo) >> 1);
        if (wave_count(mid, true) >= kk) hi = mid; else lo = mid + 1;
    }
    const unsigned T = lo;
    const int need_ties = kk - wave_count(T, false);
    // compact the selected pairs in scan order: all d < T, then the first need_ties with d == T
    sd[lane] = INFINITY;
    sj[lane] = 0x7fffffff;
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int out = 0, ties = 0;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (c < slots) {
            const unsigned long long tie_m = __ballot(dv[c] == T);
            const int my_tie = ties + __popcll(tie_m & below);
            const bool sel = (dv[c] < T) || (dv[c] == T && my_tie < need_ties);
            const unsigned long long sel_m = __ballot(sel);
            if (sel) {
                const int slot = out + __popcll(sel_m & below);
                sd[slot] = __uint_as_float(dv[c]);
                sj[slot] = EXPLICIT_J ? jv[EXPLICIT_J ? c : 0] : (j0 + c * 64);
            }
            out += __popcll(sel_m);
            ties += __popcll(tie_m);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    bd = sd[lane];
    bj = sj[lane];
    sort64_pairs(bd, bj, lane);
}

// Selection path.  Bisection over all cached candidates costs two vector instructions per candidate and probe, so it
// only runs until the upper end of the search interval admits at most KNN_SURV candidates (about five probes: the
// count grows like d^3); those survivors are compacted into LDS, four per lane, and the remaining ~20 probes, the tie
// handling and the sort work on them alone.  Massive ties (interval closed with more than KNN_SURV survivors) take
// the same code over the whole cache.
constexpr int KNN_SURV = 256;
// Development aid (tools/knn_probe.py compiles a private copy with -DKNN_PHASE_CUT=n): the selection returns after phase n
// with a value that keeps the phase's work alive.  Never defined in libpamnet_hip.so.
#ifdef KNN_PHASE_CUT
#define KNN_CUT(n, v)                                     \
    do {                                                  \
        if (KNN_PHASE_CUT == (n)) {                       \
            bd = __uint_as_float(v), bj = (int)(v);       \
            return;                                       \
        }                                                 \
    } while (0)
#else
#define KNN_CUT(n, v)
#endif

__device__ __forceinline__ void knn_select(const float* __restrict__ pos, float xi, float yi, float zi, int beg, int end,
                                           int K, int lane, float* sd, int* sj, unsigned* vd, int* vj, float& bd,
                                           int& bj) {
    const int iters = (end - beg + 63) >> 6;
    const int kk = (end - beg < K) ? (end - beg) : K;       // entries that exist
    unsigned dc[KNN_CACHE];
    unsigned mn = 0xffffffffu, mx = 0u;
    int zeros = 0;
    // Eight slots at a time: their 24 coordinate loads are requested together (no lane guard: candidates behind the graph's end
    // read its last atom and are masked afterwards).  Slot by slot behind a lane guard each slot's three loads were a round
    // trip of their own -- a graph's coordinates (46 KB) do not stay in the 32 KB L1, so mostly to the L2 -- and the pass was
    // 35 dependent round trips per query: 30.7 of the kernel's 83 us (tools/knn_probe.py).
#pragma unroll
    for (int c0 = 0; c0 < KNN_CACHE; c0 += 8) {
        if (c0 < iters) {                                   // (wave-uniform)
            float px[8], py[8], pz[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = beg + (c0 + u) * 64 + lane;
                const int64_t jc = j < end ? j : end - 1;
                // one 12-byte load per candidate: the wave's 64 candidates are 768 contiguous bytes, touched once instead of
                // once per coordinate
                const float3 pj = *reinterpret_cast<const float3*>(pos + 3 * jc);
                px[u] = pj.x, py[u] = pj.y, pz[u] = pj.z;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = c0 + u;
                const int j = beg + c * 64 + lane;
                const float dx = xi - px[u], dy = yi - py[u], dz = zi - pz[u];
                const unsigned d = __float_as_uint(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
                const bool valid = j < end;                 // (sqdist's arithmetic, same bits)
                dc[c] = valid ? d : 0xffffffffu;
                zeros += (valid && d == 0u) ? 1 : 0;
                mn = (valid && d != 0u && d < mn) ? d : mn;          // smallest NON-ZERO distance (knn_topk: why)
                mx = (valid && d > mx) ? d : mx;
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) dc[c0 + u] = 0xffffffffu;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned a = (unsigned)__shfl_xor((int)mn, o, 64), b = (unsigned)__shfl_xor((int)mx, o, 64);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
        zeros += __shfl_xor(zeros, o, 64);
    }
    unsigned lo = (mn != 0xffffffffu && zeros < kk) ? mn : 0u, hi = mx;
    int cnt_hi = end - beg;                                 // #(d <= hi), always >= kk
    KNN_CUT(1, mx);                                         // (development probe: the distance pass alone)
    while (lo < hi && cnt_hi > KNN_SURV) {
        const unsigned mid = lo + ((hi - lo) >> 1);
        int c_lane = 0;
#pragma unroll
        for (int c0 = 0; c0 < KNN_CACHE; c0 += 8) {
            if (c0 < iters) {
#pragma unroll
                for (int c = c0; c < c0 + 8; ++c) c_lane += (dc[c] <= mid) ? 1 : 0;
            }
        }
        int tot = 0;
#pragma unroll
        for (int b = 0; b < 7; ++b) tot += __popcll(__ballot((c_lane >> b) & 1)) << b;
        if (tot >= kk) { hi = mid; cnt_hi = tot; } else lo = mid + 1;
    }
    KNN_CUT(2, hi);                                         // (+ the first-stage bisection)
    if (cnt_hi > KNN_SURV) {                                // wave-uniform; more than KNN_SURV candidates tie at lo
        const int none[1] = {0};
        knn_topk<KNN_CACHE, 8, false>(dc, none, beg + lane, iters, kk, lane, sd, sj, bd, bj);
        return;
    }
    // survivors d <= hi, in scan order
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int nsurv = 0;
#pragma unroll
    for (int c = 0; c < KNN_CACHE; ++c) {
        if (c < iters) {
            const bool sel = dc[c] <= hi;
            const unsigned long long m = __ballot(sel);
            if (m) {                                        // (wave-uniform: the survivors of a query sit in a few slots)
                if (sel) {
                    const int slot = nsurv + __popcll(m & below);
                    vd[slot] = dc[c];
                    vj[slot] = beg + c * 64 + lane;
                }
                nsurv += __popcll(m);
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    KNN_CUT(3, (unsigned)nsurv);                            // (+ the compaction of the survivors)
    unsigned sv[KNN_SURV / 64];
    int svj[KNN_SURV / 64];
#pragma unroll
    for (int t = 0; t < KNN_SURV / 64; ++t) {
        const int slot = t * 64 + lane;
        sv[t] = slot < nsurv ? vd[slot] : 0xffffffffu;
        svj[t] = slot < nsurv ? vj[slot] : 0x7fffffff;
    }
    // Second stage, on the survivors alone (four per lane: a probe is 8 compares + 3 ballots): the bisection goes on until at
    // most 64 candidates lie at or below the upper end -- typically three or four more probes, where closing the interval
    // completely takes ~25 -- and those are sorted by the 64-lane network; the first kk of them are the answer (ties at the
    // K-th distance fall to the lower index through the sort's (distance, index) order, as in knn_topk).
    constexpr int NS = KNN_SURV / 64;
    while (lo < hi && cnt_hi > 64) {
        const unsigned mid = lo + ((hi - lo) >> 1);
        int c_lane = 0;
#pragma unroll
        for (int t = 0; t < NS; ++t) c_lane += (sv[t] <= mid) ? 1 : 0;
        int tot = 0;
#pragma unroll
        for (int b = 0; b < 3; ++b) tot += __popcll(__ballot((c_lane >> b) & 1)) << b;
        if (tot >= kk) { hi = mid; cnt_hi = tot; } else lo = mid + 1;
    }
    KNN_CUT(4, hi);                                          // (+ the second-stage bisection)
    if (cnt_hi > 64) {                                       // more than 64 candidates tie at lo: the general selection
        knn_topk<NS, NS, true>(sv, svj, 0, NS, kk, lane, sd, sj, bd, bj);
        return;
    }
    sd[lane] = INFINITY;
    sj[lane] = 0x7fffffff;
    int out = 0;
#pragma unroll
    for (int t = 0; t < NS; ++t) {
        const bool sel = sv[t] <= hi;
        const unsigned long long m = __ballot(sel);
        if (sel) {
            const int slot = out + __popcll(m & below);
            sd[slot] = __uint_as_float(sv[t]);
            sj[slot] = svj[t];
        }
        out += __popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    bd = sd[lane];
    bj = sj[lane];
    KNN_CUT(5, (unsigned)bj);                                // (+ the second compaction)
    sort64_pairs(bd, bj, lane);
}

__global__ __launch_bounds__(256) void knn_kernel(const float* __restrict__ pos, const int32_t* __restrict__ node_graph,
                                                  const int32_t* __restrict__ gptr, int64_t n, int K, float cutoff,
                                                  int32_t* __restrict__ nbr, float* __restrict__ dist,
                                                  float cut_a = 0.f, float cut_b = 0.f,
                                                  int32_t* __restrict__ cnt_a = nullptr,
                                                  int32_t* __restrict__ cnt_b = nullptr) {
    __shared__ float sd[4][64];
    __shared__ int sj[4][64];
    __shared__ unsigned vd[4][KNN_SURV];
    __shared__ int vj[4][KNN_SURV];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (i >= n) return;
    const int g = node_graph[i];
    const int beg = gptr[g], end = gptr[g + 1];
    const float xi = pos[3 * i], yi = pos[3 * i + 1], zi = pos[3 * i + 2];
    float bd;
    int bj;
    if (end - beg <= 64 * KNN_CACHE) knn_select(pos, xi, yi, zi, beg, end, K, lane, sd[w], sj[w], vd[w], vj[w], bd, bj);
    else knn_stream(pos, xi, yi, zi, beg, end, K, lane, bd, bj);
    const float d = __fsqrt_rn(bd);
    const bool keep = lane < K && (bj != 0x7fffffff) && (bj != (int)i) && (d <= cutoff);
    if (lane < K) {
        nbr[i * K + lane] = keep ? bj : -1;
        dist[i * K + lane] = d;
    }
    if (cnt_a) {                                              // entries a cut of this row at cut_a / cut_b keeps (csr_filter_kernel)
        const int ca = __popcll(__ballot(keep && d <= cut_a)), cb = __popcll(__ballot(keep && d <= cut_b));
        if (lane == 0) cnt_a[i] = ca, cnt_b[i] = cb;
    }
}

// Both cuts of a kNN table (rows of K entries, -1 = dropped) written in one pass, a wavefront per query: the kept entries of
// row i go to slots raw[i] + rank (capped), with the query id beside them, and the clamped pointers are written on the way --
// what two pamnet_csr_filter_fill_i32, two pointer clamps, a stride pointer and two pamnet_expand_rows_i32 launches did.
__global__ __launch_bounds__(256) void knn_cut_fill_kernel(const int32_t* __restrict__ kn, const float* __restrict__ kd, int64_t n,
                                                           int K, float cut_a, const int32_t* __restrict__ raw_a, int64_t cap_a,
                                                           int32_t* __restrict__ nbr_a, float* __restrict__ dist_a,
                                                           int32_t* __restrict__ row_a, int32_t* __restrict__ ptr_a, float cut_b,
                                                           const int32_t* __restrict__ raw_b, int64_t cap_b,
                                                           int32_t* __restrict__ nbr_b, float* __restrict__ dist_b,
                                                           int32_t* __restrict__ row_b, int32_t* __restrict__ ptr_b) {
    const int lane = threadIdx.x & 63;
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (i >= n) return;
    int j = -1;
    float d = 0.f;
    if (lane < K) j = kn[i * K + lane], d = kd[i * K + lane];
    const unsigned long long below = (1ull << lane) - 1ull;
    {
        const bool keep = j >= 0 && d <= cut_a;
        const int64_t w = (int64_t)raw_a[i] + __popcll(__ballot(keep) & below);
        if (keep && w < cap_a) nbr_a[w] = j, dist_a[w] = d, row_a[w] = (int32_t)i;
    }
    {
        const bool keep = j >= 0 && d <= cut_b;
        const int64_t w = (int64_t)raw_b[i] + __popcll(__ballot(keep) & below);
        if (keep && w < cap_b) nbr_b[w] = j, dist_b[w] = d, row_b[w] = (int32_t)i;
    }
    if (lane == 0) {
        ptr_a[i] = (int32_t)(raw_a[i] < cap_a ? raw_a[i] : cap_a);
        ptr_b[i] = (int32_t)(raw_b[i] < cap_b ? raw_b[i] : cap_b);
        if (i == n - 1) {
            ptr_a[n] = (int32_t)(raw_a[n] < cap_a ? raw_a[n] : cap_a);
            ptr_b[n] = (int32_t)(raw_b[n] < cap_b ? raw_b[n] : cap_b);
        }
    }
}

// One wavefront per input row: the lanes read 64 consecutive entries (coalesced), vote, and the survivors keep their
// order through the prefix population count of the ballot.  (One thread per row walked its 50 entries alone, 200 bytes
// apart from its neighbour lane: 23 us per pass over the 17 700 x 50 RNA table.)
template <bool FILL>
__global__ __launch_bounds__(256) void csr_filter_kernel(const int32_t* __restrict__ ptr_in,
                                                         const int32_t* __restrict__ nbr,
                                                         const float* __restrict__ dist, int64_t rows, float cut,
                                                         int32_t* __restrict__ count,
                                                         const int32_t* __restrict__ ptr_out,
                                                         int32_t* __restrict__ nbr_out, float* __restrict__ dist_out,
                                                         int64_t cap) {
    const int lane = threadIdx.x & 63;
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= rows) return;
    const int beg = ptr_in[r], end = ptr_in[r + 1];
    int c = 0;
    const int64_t w0 = FILL ? ptr_out[r] : 0;
    for (int q0 = beg; q0 < end; q0 += 64) {
        const int q = q0 + lane;
        int j = -1;
        float d = 0.f;
        if (q < end) {
            j = nbr[q];
            d = dist[q];
        }
        const bool keep = q < end && j >= 0 && d <= cut;
        const unsigned long long votes = __ballot(keep);
        if (FILL && keep) {
            const int64_t w = w0 + c + __builtin_popcountll(votes & ((1ull << lane) - 1ull));
            if (w < cap) {
                nbr_out[w] = j;
                dist_out[w] = d;
            }
        }
        c += __builtin_popcountll(votes);
    }
    if (!FILL && lane == 0) count[r] = c;
}

__global__ __launch_bounds__(256) void edge_dist_kernel(const float* __restrict__ pos, const int32_t* __restrict__ a,
                                                        const int32_t* __restrict__ b, int64_t m,
                                                        float* __restrict__ dist) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < m) dist[e] = dist3(pos, a[e], b[e]);
}

__global__ __launch_bounds__(256) void expand_rows_kernel(const int32_t* __restrict__ ptr, int64_t rows,
                                                          int32_t* __restrict__ row_of, int64_t cap) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const int64_t e = ptr[r + 1] < cap ? ptr[r + 1] : cap;
    for (int64_t q = ptr[r]; q < e; ++q) row_of[q] = (int32_t)r;
}

__global__ __launch_bounds__(256) void triplet_count_kernel(const int32_t* __restrict__ lptr,
                                                            const int32_t* __restrict__ src,
                                                            const int32_t* __restrict__ dst, int64_t n_edges,
                                                            int with_triplets, int32_t* __restrict__ tcount,
                                                            int32_t* __restrict__ tpcount) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_edges) return;
    const int j = src[e], i = dst[e];
    int t = 0;
    if (with_triplets)
        for (int q = lptr[j]; q < lptr[j + 1]; ++q) t += (src[q] != i) ? 1 : 0;  // edges k->j, k != i
    tcount[e] = t;
    tpcount[e] = t + (lptr[i + 1] - lptr[i]);                                     // + edges j'->i, incl. e itself
}


// rows of edge e: [tp_ptr[e], tp_ptr[e+1]) = its triplets (kind 0) followed by its pairs (kind 1)
__global__ __launch_bounds__(256) void triplet_fill_kernel(const float* __restrict__ pos,
                                                           const int32_t* __restrict__ lptr,
                                                           const int32_t* __restrict__ src,
                                                           const int32_t* __restrict__ dst, int64_t n_edges,
                                                           int with_triplets, const int32_t* __restrict__ tp_ptr,
                                                           int32_t* __restrict__ tp_idx, int32_t* __restrict__ tp_edge,
                                                           float* __restrict__ tp_angle, int32_t* __restrict__ tp_kind,
                                                           int64_t cap) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_edges) return;
    const int j = src[e], i = dst[e];
    const float pix = pos[3 * (int64_t)i], piy = pos[3 * (int64_t)i + 1], piz = pos[3 * (int64_t)i + 2];
    const float pjx = pos[3 * (int64_t)j], pjy = pos[3 * (int64_t)j + 1], pjz = pos[3 * (int64_t)j + 2];
    int64_t w = tp_ptr[e];
    if (with_triplets) {
        // triplets: pos_ji = p_j - p_i, pos_kj = p_k - p_j                        (models.py:165)
        for (int q = lptr[j]; q < lptr[j + 1]; ++q) {
            const int k = src[q];
            if (k == i) continue;
            if (w >= cap) break;
            const float pkx = pos[3 * (int64_t)k], pky = pos[3 * (int64_t)k + 1], pkz = pos[3 * (int64_t)k + 2];
            tp_idx[w] = q;
            tp_edge[w] = (int32_t)e;
            tp_kind[w] = 0;
            tp_angle[w] = angle3(pjx - pix, pjy - piy, pjz - piz, pkx - pjx, pky - pjy, pkz - pjz);
            ++w;
        }
    }
    // pairs: idx_i_pair = j, idx_j1_pair = i, idx_j2_pair = j'  ->  a = p_i - p_j, b = p_j' - p_i  (models.py:171-177)
    for (int q = lptr[i]; q < lptr[i + 1]; ++q) {
        const int j2 = src[q];
        if (w >= cap) break;
        const float px = pos[3 * (int64_t)j2], py = pos[3 * (int64_t)j2 + 1], pz = pos[3 * (int64_t)j2 + 2];
        tp_idx[w] = q;
        tp_edge[w] = (int32_t)e;
        tp_kind[w] = 1;
        tp_angle[w] = angle3(pix - pjx, piy - pjy, piz - pjz, px - pix, py - piy, pz - piz);
        ++w;
    }
}

// ---- transposed triplet / pair row list without a sort -----------------------------------------------------------------
// For source bond q = (k -> j) the rows that gather it (what pamnet_csr_from_keys_i32 returns for keys = tp_idx) are: one
// triplet row of every bond e leaving j towards an atom other than k (row = e's first row + q's rank among e's triplets),
// and one pair row of every bond e arriving at j (row = e's first pair row + q's place in j's list); rows grow with e, so
// an entry's place is the number of such bonds with a smaller id.  A thread per bond and a few loads each, instead of a
// counting sort over the T + P rows (hist + claim + sort: 55 us of atomics at the RNA batch's 669 k rows).
// lt_ptr / lt_perm: the transposed bond list (bonds by source atom; any order inside a row).
template <bool FILL>
__global__ __launch_bounds__(256) void triplet_transpose_kernel(const int32_t* __restrict__ lptr, const int32_t* __restrict__ src,
                                                                const int32_t* __restrict__ dst,
                                                                const int32_t* __restrict__ lt_ptr,
                                                                const int32_t* __restrict__ lt_perm, int64_t n_edges,
                                                                int with_triplets, int32_t* __restrict__ count,
                                                                const int32_t* __restrict__ tp_ptr,
                                                                const int32_t* __restrict__ tcount,
                                                                const int32_t* __restrict__ tt_ptr,
                                                                int32_t* __restrict__ tt_perm, int64_t cap) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_edges) return;
    const int k = src[q], j = dst[q];
    const int bb = lptr[j], be = lptr[j + 1];
    const int ab = lt_ptr[j], ae = with_triplets ? lt_ptr[j + 1] : ab;
    if (!FILL) {
        int c = be - bb;
        for (int a = ab; a < ae; ++a) c += (dst[lt_perm[a]] != k) ? 1 : 0;
        count[q] = c;
        return;
    }
    const int64_t w0 = tt_ptr[q];
    constexpr int CAP = 16;
    if (ae - ab <= CAP && be - bb <= CAP) {
        // Usual degrees: both lists of atom j in registers first -- every load below is independent of the others (the
        // loops further down chase dst[lt_perm[.]] inside two nested loops: ~100 dependent loads per thread, 54 us at the
        // RNA batch with one wave per SIMD).
        int ea[CAP], ia[CAP], ta[CAP], sb[CAP], tb[CAP];
        const int na = ae - ab, nb = be - bb;
#pragma unroll
        for (int u = 0; u < CAP; ++u) ea[u] = u < na ? lt_perm[ab + u] : 0x7fffffff;
#pragma unroll
        for (int u = 0; u < CAP; ++u) {
            ia[u] = u < na ? dst[ea[u]] : k;                  // (padding counts as "towards k": not a triplet)
            ta[u] = u < na ? tp_ptr[ea[u]] : 0;
            sb[u] = u < nb ? src[bb + u] : 0;
            tb[u] = u < nb ? tp_ptr[bb + u] + tcount[bb + u] : 0;
        }
        const int qb = (int)q - bb;
#pragma unroll
        for (int u = 0; u < CAP; ++u) {                       // triplet rows: bonds e = (j -> i), i != k
            if (u < na && ia[u] != k) {
                const int e = ea[u];
                int pos = e - bb < 0 ? 0 : (e - bb < nb ? e - bb : nb), rank = 0;
#pragma unroll
                for (int v = 0; v < CAP; ++v) {
                    pos += (ia[v] != k && ea[v] < e) ? 1 : 0;
                    rank += (v < qb && sb[v] != ia[u]) ? 1 : 0;
                }
                const int64_t row = (int64_t)ta[u] + rank;
                if (w0 + pos < cap) tt_perm[w0 + pos] = (int32_t)(row < cap ? row : cap - 1);
            }
        }
#pragma unroll
        for (int u = 0; u < CAP; ++u) {                       // pair rows: bonds e = (j' -> j)
            if (u < nb) {
                const int e = bb + u;
                int pos = u;
#pragma unroll
                for (int v = 0; v < CAP; ++v) pos += (ia[v] != k && ea[v] < e) ? 1 : 0;
                const int64_t row = (int64_t)tb[u] + qb;
                if (w0 + pos < cap) tt_perm[w0 + pos] = (int32_t)(row < cap ? row : cap - 1);
            }
        }
        return;
    }
    for (int a = ab; a < ae; ++a) {                           // triplet rows: bonds e = (j -> i), i != k
        const int e = lt_perm[a];
        const int i = dst[e];
        if (i == k) continue;
        int pos = e - bb < 0 ? 0 : (e - bb < be - bb ? e - bb : be - bb);        // bonds arriving at j with a smaller id
        for (int a2 = ab; a2 < ae; ++a2) {
            const int e2 = lt_perm[a2];
            pos += (e2 < e && dst[e2] != k) ? 1 : 0;
        }
        int rank = 0;                                         // q's place among e's triplets: bonds (k' -> j), k' != i
        for (int q2 = bb; q2 < (int)q; ++q2) rank += (src[q2] != i) ? 1 : 0;
        const int64_t row = tp_ptr[e] + rank;               // (clamped: sizes that turn out wrong must stay inside the arrays)
        if (w0 + pos < cap) tt_perm[w0 + pos] = (int32_t)(row < cap ? row : cap - 1);
    }
    for (int e = bb; e < be; ++e) {                           // pair rows: bonds e = (j' -> j)
        int pos = e - bb;
        for (int a2 = ab; a2 < ae; ++a2) {
            const int e2 = lt_perm[a2];
            pos += (e2 < e && dst[e2] != k) ? 1 : 0;
        }
        const int64_t row = (int64_t)tp_ptr[e] + tcount[e] + ((int)q - bb);
        if (w0 + pos < cap) tt_perm[w0 + pos] = (int32_t)(row < cap ? row : cap - 1);
    }
}

inline unsigned blocks_for(int64_t n, int per = 256) { return (unsigned)(n > 0 ? ceil_div(n, per) : 1); }
// PAMNET_SMALL_FORMS=0: always the multi-launch scan / counting sort (the tests compare the two)
inline bool small_forms() {
    static bool v = [] { const char* e = getenv("PAMNET_SMALL_FORMS"); return !e || atoi(e) != 0; }();
    return v;
}

}  // namespace

extern "C" int pamnet_exclusive_scan_i32(const int32_t* in, int32_t* out, int64_t n, int32_t* tmp,
                                         pamnet_stream_t stream) {
    if (n < 0) return PAMNET_EINVAL;
    if (!out || (n > 0 && (!in || !tmp))) return PAMNET_ENULL;
    hipStream_t st = as_stream(stream);
    if (n == 0) {
        hipError_t e = hipMemsetAsync(out, 0, sizeof(int32_t), st);
        return (int)e;
    }
    if (n <= SMALL_SCAN_MAX && small_forms()) {
        hipLaunchKernelGGL(scan_small_kernel, dim3(1), dim3(SMALL_THREADS), 0, st, in, out, (int)n, (const int32_t*)nullptr,
                           (int32_t*)nullptr);
        PAMNET_LAUNCH_CHECK();
        return PAMNET_OK;
    }
    const int64_t nb = ceil_div(n, SCAN_CHUNK);
    hipLaunchKernelGGL(scan_chunk_kernel, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, st, in, out, n, tmp);
    PAMNET_LAUNCH_CHECK();
    if (nb <= 1024) {                                         // two launches: every add workgroup sums its chunk's offset itself
        hipLaunchKernelGGL((scan_add_kernel<true>), dim3(blocks_for(n)), dim3(256), 0, st, out, n, tmp);
        PAMNET_LAUNCH_CHECK();
        return PAMNET_OK;
    }
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(SCAN_THREADS), 0, st, tmp, nb);
    PAMNET_LAUNCH_CHECK();
    hipLaunchKernelGGL((scan_add_kernel<false>), dim3(blocks_for(n)), dim3(256), 0, st, out, n, tmp);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

// PAMNET_HIST_LDS=0 keeps the plain kernel (A/B runs).  More than 64 KB of dynamic LDS has to be asked for -- per launch, not
// once per process: the attribute belongs to the device the call runs on (a process may drive several).
static bool hist_lds_enabled() {
    static const bool on = [] {
        const char* e = getenv("PAMNET_HIST_LDS");
        return !(e && e[0] == '0');
    }();
    return on;
}
static bool hist_lds_ready(size_t bytes) {
    if (!hist_lds_enabled()) return false;
    if (bytes <= 64 * 1024) return true;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(hist_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)bytes) == hipSuccess)
        return true;
    // a device without that much LDS: the caller takes the plain kernel.  The refused call leaves HIP's sticky "last error"
    // set, and the launch check behind the plain kernel would report it as that launch's failure -- clear it here.
    (void)hipGetLastError();
    return false;
}

static int csr_from_keys(const int32_t* keys, int64_t m, int64_t rows, int32_t* ptr, int32_t* perm, int32_t* cursor,
                         int32_t* perm_tmp, int32_t* tmp, bool cursor_is_zero, pamnet_stream_t stream) {
    if (m < 0 || rows <= 0) return PAMNET_EINVAL;
    if (!ptr || !cursor || !tmp || (m > 0 && (!keys || !perm || !perm_tmp))) return PAMNET_ENULL;
    hipStream_t st = as_stream(stream);
    if (m > 0 && m <= SMALL_CSR_KEYS && rows <= SMALL_CSR_ROWS && small_forms()) {
        hipLaunchKernelGGL(csr_small_kernel, dim3(1), dim3(SMALL_THREADS), 0, st, keys, (int)m, (int)rows, ptr, perm, perm_tmp);
        PAMNET_LAUNCH_CHECK();
        return PAMNET_OK;
    }
    int32_t* unsorted = cursor + rows;                       // the spare int behind the counters: one memset for both
    if (!cursor_is_zero) {
        hipError_t e = hipMemsetAsync(cursor, 0, sizeof(int32_t) * (rows + 1), st);
        if (e != hipSuccess) return (int)e;
    }
    if (m > 0) {
        if (m >= HIST_LDS_MIN_KEYS && rows <= HIST_LDS_MAX_ROWS && hist_lds_ready(sizeof(int32_t) * (size_t)rows)) {
            const int64_t per = (int64_t)HIST_LDS_THREADS * HIST_LDS_PER_THREAD;
            hipLaunchKernelGGL(hist_lds_kernel, dim3((unsigned)((m + per - 1) / per)), dim3(HIST_LDS_THREADS),
                               sizeof(int32_t) * (size_t)rows, st, keys, m, (int)rows, cursor, unsorted, perm);
        } else {
            hipLaunchKernelGGL(hist_kernel, dim3(blocks_for(m)), dim3(256), 0, st, keys, m, rows, cursor, unsorted, perm);
        }
        PAMNET_LAUNCH_CHECK();
    }
    int rc = pamnet_exclusive_scan_i32(cursor, ptr, rows, tmp, stream);
    if (rc) return rc;
    if (m == 0) return PAMNET_OK;
    hipLaunchKernelGGL(claim_kernel, dim3(blocks_for(m)), dim3(256), 0, st, keys, m, rows, ptr, cursor, perm_tmp, perm,
                       unsorted);
    PAMNET_LAUNCH_CHECK();
    hipLaunchKernelGGL(sort_rows_kernel, dim3(blocks_for(m)), dim3(256), 0, st, keys, ptr, perm_tmp, perm, m, rows, unsorted);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

// two arrays of the same length in one launch (the two cuts of a kNN table); longer arrays: one after the other
extern "C" int pamnet_exclusive_scan_pair_i32(const int32_t* in_a, int32_t* out_a, const int32_t* in_b, int32_t* out_b,
                                              int64_t n, int32_t* tmp, pamnet_stream_t stream) {
    if (n < 1) return PAMNET_EINVAL;
    if (!in_a || !out_a || !in_b || !out_b || !tmp) return PAMNET_ENULL;
    if (n <= SMALL_SCAN_MAX && small_forms()) {
        hipLaunchKernelGGL(scan_small_kernel, dim3(2), dim3(SMALL_THREADS), 0, as_stream(stream), in_a, out_a, (int)n, in_b,
                           out_b);
        PAMNET_LAUNCH_CHECK();
        return PAMNET_OK;
    }
    const int rc = pamnet_exclusive_scan_i32(in_a, out_a, n, tmp, stream);
    return rc ? rc : pamnet_exclusive_scan_i32(in_b, out_b, n, tmp, stream);
}

extern "C" int pamnet_csr_from_keys_i32(const int32_t* keys, int64_t m, int64_t rows, int32_t* ptr, int32_t* perm,
                                        int32_t* cursor, int32_t* perm_tmp, int32_t* tmp, pamnet_stream_t stream) {
    return csr_from_keys(keys, m, rows, ptr, perm, cursor, perm_tmp, tmp, false, stream);
}

// The same with `cursor` (rows + 1 ints) already zero-filled by the caller -- e.g. as part of one larger fill: no memset
// launch of its own (the graph-construction engine zero-fills every counter array of a batch together).
extern "C" int pamnet_csr_from_keys_z_i32(const int32_t* keys, int64_t m, int64_t rows, int32_t* ptr, int32_t* perm,
                                          int32_t* cursor, int32_t* perm_tmp, int32_t* tmp, pamnet_stream_t stream) {
    return csr_from_keys(keys, m, rows, ptr, perm, cursor, perm_tmp, tmp, true, stream);
}

extern "C" int pamnet_expand_rows_i32(const int32_t* ptr, int64_t rows, int32_t* row_of, int64_t cap,
                                      pamnet_stream_t stream) {
    if (rows < 0 || cap < 0) return PAMNET_EINVAL;
    if (rows == 0) return PAMNET_OK;
    if (!ptr || !row_of) return PAMNET_ENULL;
    hipLaunchKernelGGL(expand_rows_kernel, dim3(blocks_for(rows)), dim3(256), 0, as_stream(stream), ptr, rows, row_of, cap);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_radius_count_i32(const float* pos, const int32_t* node_graph, const int32_t* gptr, int64_t n,
                                       int64_t n_graphs, float r, int64_t max_neighbors, int32_t* count,
                                       int32_t* cap_flag, pamnet_stream_t stream) {
    if (n < 0 || n_graphs < 0 || max_neighbors < 0 || max_neighbors > 0x7fffffff) return PAMNET_EINVAL;
    if (n == 0) return PAMNET_OK;
    if (!pos || !node_graph || !gptr || !count) return PAMNET_ENULL;
    const int mx = (int)max_neighbors;
    if (n_graphs > 0 && n >= RADIUS_WAVE_MIN_NODES * n_graphs)
        hipLaunchKernelGGL((radius_wave_kernel<false>), dim3(blocks_for(n, 4)), dim3(256), 0, as_stream(stream), pos,
                           node_graph, gptr, n, r, count, (const int32_t*)nullptr, (int32_t*)nullptr, (float*)nullptr,
                           (int64_t)0, (int32_t*)nullptr, mx, cap_flag);
    else
        hipLaunchKernelGGL((radius_kernel<false>), dim3(blocks_for(n)), dim3(256), 0, as_stream(stream), pos, node_graph,
                           gptr, n, r, count, (const int32_t*)nullptr, (int32_t*)nullptr, (float*)nullptr, (int64_t)0,
                           (int32_t*)nullptr, mx, cap_flag);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_radius_fill_i32(const float* pos, const int32_t* node_graph, const int32_t* gptr, int64_t n,
                                      int64_t n_graphs, float r, int64_t max_neighbors, const int32_t* ptr, int32_t* nbr,
                                      float* dist, int32_t* row_of, int64_t cap, pamnet_stream_t stream) {
    if (n < 0 || cap < 0 || n_graphs < 0 || max_neighbors < 0 || max_neighbors > 0x7fffffff) return PAMNET_EINVAL;
    if (n == 0) return PAMNET_OK;
    if (!pos || !node_graph || !gptr || !ptr || !nbr || !dist) return PAMNET_ENULL;
    const int mx = (int)max_neighbors;
    if (n_graphs > 0 && n >= RADIUS_WAVE_MIN_NODES * n_graphs)
        hipLaunchKernelGGL((radius_wave_kernel<true>), dim3(blocks_for(n, 4)), dim3(256), 0, as_stream(stream), pos,
                           node_graph, gptr, n, r, (int32_t*)nullptr, ptr, nbr, dist, cap, row_of, mx, (int32_t*)nullptr);
    else
        hipLaunchKernelGGL((radius_kernel<true>), dim3(blocks_for(n)), dim3(256), 0, as_stream(stream), pos, node_graph,
                           gptr, n, r, (int32_t*)nullptr, ptr, nbr, dist, cap, row_of, mx, (int32_t*)nullptr);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_knn_i32(const float* pos, const int32_t* node_graph, const int32_t* gptr, int64_t n, int32_t k,
                              float cutoff, int32_t* nbr, float* dist, pamnet_stream_t stream) {
    if (n < 0 || k < 1 || k > 64) return PAMNET_EINVAL;
    if (n == 0) return PAMNET_OK;
    if (!pos || !node_graph || !gptr || !nbr || !dist) return PAMNET_ENULL;
    hipLaunchKernelGGL(knn_kernel, dim3(blocks_for(n, 4)), dim3(256), 0, as_stream(stream), pos, node_graph, gptr, n,
                       (int)k, cutoff, nbr, dist, 0.f, 0.f, (int32_t*)nullptr, (int32_t*)nullptr);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_knn_cut_i32(const float* pos, const int32_t* node_graph, const int32_t* gptr, int64_t n, int32_t k,
                                  float cut_a, float cut_b, int32_t* nbr, float* dist, int32_t* cnt_a, int32_t* cnt_b,
                                  pamnet_stream_t stream) {
    if (n < 0 || k < 1 || k > 64) return PAMNET_EINVAL;
    if (n == 0) return PAMNET_OK;
    if (!pos || !node_graph || !gptr || !nbr || !dist || !cnt_a || !cnt_b) return PAMNET_ENULL;
    hipLaunchKernelGGL(knn_kernel, dim3(blocks_for(n, 4)), dim3(256), 0, as_stream(stream), pos, node_graph, gptr, n,
                       (int)k, INFINITY, nbr, dist, cut_a, cut_b, cnt_a, cnt_b);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

// ---- the triplet + pair row total of the graph the LOCAL cut of a kNN table defines, before that graph exists (round 6) ----
// Edges: query j -> neighbour i for the table entries of j within `cut` (models.py:153-156: j = query, i = neighbour; the local
// layer aggregates at i).  Rows of edge (j -> i), models.py:68-98 as triplet_count_kernel counts them on the finished graph:
//     triplets  #{k -> j, k != i} = indeg(j) - [i -> j is an edge]        pairs  #{j' -> i} = indeg(i)
// Two launches ahead of the host's ONE size read-back of the kNN path (the second read-back, of the scanned row counts, goes):
// a histogram of in-degrees (integer atomics: a count is a count whatever the order), then per table entry its rows, summed with
// integer atomics into total[0].  One wavefront per query, as the fill kernel.
__global__ __launch_bounds__(256) void knn_indeg_kernel(const int32_t* __restrict__ kn, const float* __restrict__ kd, int64_t n,
                                                        int K, float cut, int32_t* __restrict__ indeg) {
    const int lane = threadIdx.x & 63;
    const int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (j >= n || lane >= K) return;
    const int i = kn[j * K + lane];
    if (i >= 0 && kd[j * K + lane] <= cut) atomicAdd(indeg + i, 1);
}
__global__ __launch_bounds__(256) void knn_tp_total_kernel(const int32_t* __restrict__ kn, const float* __restrict__ kd, int64_t n,
                                                           int K, float cut, int with_triplets,
                                                           const int32_t* __restrict__ indeg,
                                                           unsigned long long* __restrict__ total) {
    const int lane = threadIdx.x & 63;
    const int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (j >= n) return;
    int i = -1;
    bool keep = false;
    if (lane < K) {
        i = kn[j * K + lane];
        keep = i >= 0 && kd[j * K + lane] <= cut;
    }
    long long mine = keep ? indeg[i] : 0;                                 // pairs
    if (with_triplets) {
        // is i -> j an edge, i.e. j within the cut of i's own list?  The wave reads the list of one kept neighbour at a time (one
        // coalesced row per step, all steps independent) instead of every lane walking a list of its own
        unsigned long long todo = __ballot(keep);
        int mutual = 0;
        while (todo) {
            const int t = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int it = __builtin_amdgcn_readlane(i, t);
            const bool hit = lane < K && kn[(int64_t)it * K + lane] == (int32_t)j && kd[(int64_t)it * K + lane] <= cut;
            if (__ballot(hit) != 0 && lane == t) mutual = 1;
        }
        if (keep) mine += indeg[j] - mutual;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
    if (lane == 0 && mine) atomicAdd(total, (unsigned long long)mine);
}
// indeg: [n] int32 scratch, total: [1] int64 -- both zero on entry
extern "C" int pamnet_knn_tp_total_i64(const int32_t* nbr, const float* dist, int64_t n, int32_t k, float cut,
                                       int32_t with_triplets, int32_t* indeg, int64_t* total, pamnet_stream_t stream) {
    if (n < 0 || k < 1 || k > 64) return PAMNET_EINVAL;
    if (n == 0) return PAMNET_OK;
    if (!nbr || !dist || !indeg || !total) return PAMNET_ENULL;
    hipLaunchKernelGGL(knn_indeg_kernel, dim3(blocks_for(n, 4)), dim3(256), 0, as_stream(stream), nbr, dist, n, (int)k, cut, indeg);
    PAMNET_LAUNCH_CHECK();
    hipLaunchKernelGGL(knn_tp_total_kernel, dim3(blocks_for(n, 4)), dim3(256), 0, as_stream(stream), nbr, dist, n, (int)k, cut,
                       (int)with_triplets, indeg, reinterpret_cast<unsigned long long*>(total));
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_knn_cut_fill_i32(const int32_t* nbr, const float* dist, int64_t n, int32_t k, float cut_a,
                                       const int32_t* raw_a, int64_t cap_a, int32_t* nbr_a, float* dist_a, int32_t* row_a,
                                       int32_t* ptr_a, float cut_b, const int32_t* raw_b, int64_t cap_b, int32_t* nbr_b,
                                       float* dist_b, int32_t* row_b, int32_t* ptr_b, pamnet_stream_t stream) {
    if (n < 1 || k < 1 || k > 64 || cap_a < 0 || cap_b < 0) return PAMNET_EINVAL;
    if (!nbr || !dist || !raw_a || !raw_b || !ptr_a || !ptr_b) return PAMNET_ENULL;
    if ((cap_a > 0 && (!nbr_a || !dist_a || !row_a)) || (cap_b > 0 && (!nbr_b || !dist_b || !row_b))) return PAMNET_ENULL;
    hipLaunchKernelGGL(knn_cut_fill_kernel, dim3(blocks_for(n, 4)), dim3(256), 0, as_stream(stream), nbr, dist, n, (int)k, cut_a,
                       raw_a, cap_a, nbr_a, dist_a, row_a, ptr_a, cut_b, raw_b, cap_b, nbr_b, dist_b, row_b, ptr_b);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_csr_filter_count_i32(const int32_t* ptr_in, const int32_t* nbr, const float* dist, int64_t rows,
                                           float cut, int32_t* count, pamnet_stream_t stream) {
    if (rows < 0) return PAMNET_EINVAL;
    if (rows == 0) return PAMNET_OK;
    if (!ptr_in || !nbr || !dist || !count) return PAMNET_ENULL;
    hipLaunchKernelGGL((csr_filter_kernel<false>), dim3(blocks_for(rows, 4)), dim3(256), 0, as_stream(stream), ptr_in, nbr,
                       dist, rows, cut, count, (const int32_t*)nullptr, (int32_t*)nullptr, (float*)nullptr, (int64_t)0);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_csr_filter_fill_i32(const int32_t* ptr_in, const int32_t* nbr, const float* dist, int64_t rows,
                                          float cut, const int32_t* ptr_out, int32_t* nbr_out, float* dist_out,
                                          int64_t cap, pamnet_stream_t stream) {
    if (rows < 0 || cap < 0) return PAMNET_EINVAL;
    if (rows == 0) return PAMNET_OK;
    if (!ptr_in || !nbr || !dist || !ptr_out || !nbr_out || !dist_out) return PAMNET_ENULL;
    hipLaunchKernelGGL((csr_filter_kernel<true>), dim3(blocks_for(rows, 4)), dim3(256), 0, as_stream(stream), ptr_in, nbr,
                       dist, rows, cut, (int32_t*)nullptr, ptr_out, nbr_out, dist_out, cap);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_edge_dist_f32(const float* pos, const int32_t* a, const int32_t* b, int64_t m, float* dist,
                                    pamnet_stream_t stream) {
    if (m < 0) return PAMNET_EINVAL;
    if (m == 0) return PAMNET_OK;
    if (!pos || !a || !b || !dist) return PAMNET_ENULL;
    hipLaunchKernelGGL(edge_dist_kernel, dim3(blocks_for(m)), dim3(256), 0, as_stream(stream), pos, a, b, m, dist);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_triplet_count_i32(const int32_t* lptr, const int32_t* src, const int32_t* dst, int64_t n_edges,
                                        int32_t with_triplets, int32_t* tcount, int32_t* tpcount,
                                        pamnet_stream_t stream) {
    if (n_edges < 0) return PAMNET_EINVAL;
    if (n_edges == 0) return PAMNET_OK;
    if (!lptr || !src || !dst || !tcount || !tpcount) return PAMNET_ENULL;
    hipLaunchKernelGGL(triplet_count_kernel, dim3(blocks_for(n_edges)), dim3(256), 0, as_stream(stream), lptr, src, dst,
                       n_edges, (int)with_triplets, tcount, tpcount);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_triplet_transpose_count_i32(const int32_t* lptr, const int32_t* src, const int32_t* dst,
                                                  const int32_t* lt_ptr, const int32_t* lt_perm, int64_t n_edges,
                                                  int32_t with_triplets, int32_t* count, pamnet_stream_t stream) {
    if (n_edges < 0) return PAMNET_EINVAL;
    if (n_edges == 0) return PAMNET_OK;
    if (!lptr || !src || !dst || !lt_ptr || !lt_perm || !count) return PAMNET_ENULL;
    hipLaunchKernelGGL((triplet_transpose_kernel<false>), dim3(blocks_for(n_edges)), dim3(256), 0, as_stream(stream), lptr, src,
                       dst, lt_ptr, lt_perm, n_edges, (int)with_triplets, count, (const int32_t*)nullptr,
                       (const int32_t*)nullptr, (const int32_t*)nullptr, (int32_t*)nullptr, (int64_t)0);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_triplet_transpose_fill_i32(const int32_t* lptr, const int32_t* src, const int32_t* dst,
                                                 const int32_t* lt_ptr, const int32_t* lt_perm, int64_t n_edges,
                                                 int32_t with_triplets, const int32_t* tp_ptr, const int32_t* tcount,
                                                 const int32_t* tt_ptr, int32_t* tt_perm, int64_t cap,
                                                 pamnet_stream_t stream) {
    if (n_edges < 0 || cap < 0) return PAMNET_EINVAL;
    if (n_edges == 0 || cap == 0) return PAMNET_OK;
    if (!lptr || !src || !dst || !lt_ptr || !lt_perm || !tp_ptr || !tcount || !tt_ptr || !tt_perm) return PAMNET_ENULL;
    hipLaunchKernelGGL((triplet_transpose_kernel<true>), dim3(blocks_for(n_edges)), dim3(256), 0, as_stream(stream), lptr, src,
                       dst, lt_ptr, lt_perm, n_edges, (int)with_triplets, (int32_t*)nullptr, tp_ptr, tcount, tt_ptr, tt_perm,
                       cap);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_triplet_fill_f32(const float* pos, const int32_t* lptr, const int32_t* src, const int32_t* dst,
                                       int64_t n_edges, int32_t with_triplets, const int32_t* tp_ptr, int32_t* tp_idx,
                                       int32_t* tp_edge, float* tp_angle, int32_t* tp_kind, int64_t cap,
                                       pamnet_stream_t stream) {
    if (n_edges < 0 || cap < 0) return PAMNET_EINVAL;
    if (n_edges == 0) return PAMNET_OK;
    if (!pos || !lptr || !src || !dst || !tp_ptr || !tp_idx || !tp_edge || !tp_angle || !tp_kind) return PAMNET_ENULL;
    hipLaunchKernelGGL(triplet_fill_kernel, dim3(blocks_for(n_edges)), dim3(256), 0, as_stream(stream), pos, lptr, src,
                       dst, n_edges, (int)with_triplets, tp_ptr, tp_idx, tp_edge, tp_angle, tp_kind, cap);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}


// ---- transposed CSR of a symmetric graph = the reverse-edge index --------------------------------------------------------
// Rows = aggregation targets i, columns j ascending inside a row (radius graphs: pamnet_radius_fill_i32).  The transposed
// CSR the backward needs (for every source j the positions q with col[q] = j, ascending q: what pamnet_csr_from_keys_i32
// returns for keys = col) has, for a symmetric graph, the same pointer array, and its k-th entry of row j -- j's k-th
// neighbour i_k -- is the position of the edge with column j in row i_k: the reverse edge.  One bisection per edge replaces
// a nine-launch counting sort over E keys.  flag[0] |= 64 if some edge has no reverse (not a symmetric graph).
namespace {
__global__ __launch_bounds__(256) void reverse_edges_kernel(const int32_t* __restrict__ ptr, const int32_t* __restrict__ row_of,
                                                            const int32_t* __restrict__ col, int64_t m,
                                                            int32_t* __restrict__ rev, int32_t* __restrict__ flag) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m) return;
    const int i = row_of[e], j = col[e];
    int lo = ptr[j], hi = ptr[j + 1];                        // find i in row j
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (col[mid] < i) lo = mid + 1;
        else hi = mid;
    }
    const bool found = lo < ptr[j + 1] && col[lo] == i;
    rev[e] = found ? lo : (int32_t)e;
    if (!found && flag) atomicOr(flag, 64);
}
}  // namespace

extern "C" int pamnet_reverse_edges_i32(const int32_t* ptr, const int32_t* row_of, const int32_t* col, int64_t m,
                                        int32_t* rev, int32_t* flag, pamnet_stream_t stream) {
    if (m < 0) return PAMNET_EINVAL;
    if (m == 0) return PAMNET_OK;
    if (!ptr || !row_of || !col || !rev) return PAMNET_ENULL;
    hipLaunchKernelGGL(reverse_edges_kernel, dim3(blocks_for(m)), dim3(256), 0, as_stream(stream), ptr, row_of, col, m, rev, flag);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

// ---- data-dependent sizes to the host in one launch + one copy --------------------------------------------------------------
// out[k] = the k-th scalar (int32 or bool, by `kind`): graph construction reads 3-4 device scalars back per batch; as torch
// expressions that was four casts, a stack and the copy.
namespace {
struct ScalarGather {
    const void* src[8];
    int kind[8];              // 0: int32, 1: bool / uint8, 2: int64
    int n;
};
__global__ void gather_scalars_kernel(ScalarGather g, int64_t* __restrict__ out) {
    const int k = threadIdx.x;
    if (k >= g.n) return;
    int64_t v;
    if (g.kind[k] == 0) v = *static_cast<const int32_t*>(g.src[k]);
    else if (g.kind[k] == 1) v = *static_cast<const uint8_t*>(g.src[k]);
    else v = *static_cast<const int64_t*>(g.src[k]);
    out[k] = v;
}
}  // namespace

extern "C" int pamnet_gather_scalars_i64(int64_t n, const void* const* src, const int32_t* kind, int64_t* out,
                                         pamnet_stream_t stream) {
    if (n < 1 || n > 8) return PAMNET_EINVAL;
    if (!src || !kind || !out) return PAMNET_ENULL;
    ScalarGather g;
    g.n = (int)n;
    for (int k = 0; k < g.n; ++k) {
        if (!src[k] || kind[k] < 0 || kind[k] > 2) return PAMNET_EINVAL;
        g.src[k] = src[k];
        g.kind[k] = kind[k];
    }
    hipLaunchKernelGGL(gather_scalars_kernel, dim3(1), dim3(8), 0, as_stream(stream), g, out);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

// ---- deferred size check (zero-host-sync graph construction) -----------------------------------------------------------
// A batch collated from a resident dataset carries its data-dependent sizes (global edges, triplet / pair rows) as host
// integers summed from per-graph counts taken once per dataset (pamnet_amd/store.py), so graph construction needs no
// host round trip.  The counts are still recomputed on the device every batch; this launch ORs bit (2 << k) into flag[0]
// when the k-th recomputed total differs from what the host assumed, and bit 1 << 5 when *all_kept is false (a bond
// list with self loops, models.py:63).  The host reads the flag whenever it next synchronises for its own reasons.
namespace {
struct SizeChecks {
    const int32_t* actual[4];
    int64_t expected[4];
    int n;
};
__global__ void check_sizes_kernel(SizeChecks c, const bool* __restrict__ all_kept, const int32_t* __restrict__ loops,
                                   int32_t* __restrict__ flag) {
    int bad = 0;
    for (int k = 0; k < c.n; ++k)
        if ((int64_t)c.actual[k][0] != c.expected[k]) bad |= 2 << k;
    if (all_kept && !all_kept[0]) bad |= 1 << 5;
    if (loops && loops[0]) bad |= 1 << 5;
    if (bad) atomicOr(flag, bad);
}

// ---- device-side batch collation -----------------------------------------------------------------------------------------
// Batch b of a resident dataset: graph k of the batch is dataset graph sel[k]; its nodes / bonds are copied from the
// dataset's concatenated arrays to the batch's (bonds are stored with graph-local endpoints and get the batch offset of
// their graph).  One launch: thread t handles output node t and output bond t; the graph of an output row is found by
// bisection in the batch's node / bond prefix sums (<= 11 probes for 1 024 graphs).
struct Collate {
    const int32_t *sel, *out_nptr, *out_eptr;     // [B], [B+1], [B+1]  (device)
    const int32_t *src_nptr, *src_eptr;           // dataset prefix sums [M+1]
    const float *x, *pos;                         // dataset node features [Ntot, xw], positions [Ntot, 3] (nullable)
    const int32_t *esrc, *edst;                   // dataset bonds, graph-local endpoints [Etot] (nullable)
    const float* y;                               // dataset targets [M] (nullable)
    float* oy;                                    // [B]
    float *ox, *opos;
    int32_t *obatch, *oesrc, *oedst;
    int64_t B, n_out, e_out, xw;
};
__device__ __forceinline__ int graph_of(const int32_t* __restrict__ ptr, int64_t B, int64_t t) {
    int lo = 0, hi = (int)B;                      // ptr[lo] <= t < ptr[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (ptr[mid] <= t) lo = mid;
        else hi = mid;
    }
    return lo;
}
__global__ __launch_bounds__(256) void collate_kernel(Collate c) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c.y && t < c.B) c.oy[t] = c.y[c.sel[t]];
    if (t < c.n_out) {
        const int k = graph_of(c.out_nptr, c.B, t);
        const int64_t s = c.src_nptr[c.sel[k]] + (t - c.out_nptr[k]);
        for (int64_t q = 0; q < c.xw; ++q) c.ox[t * c.xw + q] = c.x[s * c.xw + q];
        if (c.pos) {
            c.opos[3 * t] = c.pos[3 * s], c.opos[3 * t + 1] = c.pos[3 * s + 1], c.opos[3 * t + 2] = c.pos[3 * s + 2];
        }
        c.obatch[t] = k;
    }
    if (t < c.e_out) {
        const int k = graph_of(c.out_eptr, c.B, t);
        const int64_t s = c.src_eptr[c.sel[k]] + (t - c.out_eptr[k]);
        const int off = c.out_nptr[k];
        c.oesrc[t] = c.esrc[s] + off;
        c.oedst[t] = c.edst[s] + off;
    }
}
}  // namespace

extern "C" int pamnet_check_sizes_i32(int64_t n_checks, const int32_t* const* actual, const int64_t* expected,
                                      const void* all_kept, const int32_t* self_loops, int32_t* flag,
                                      pamnet_stream_t stream) {
    if (n_checks < 0 || n_checks > 4) return PAMNET_EINVAL;
    if (!flag || (n_checks > 0 && (!actual || !expected))) return PAMNET_ENULL;
    SizeChecks c;
    c.n = (int)n_checks;
    for (int k = 0; k < c.n; ++k) {
        if (!actual[k]) return PAMNET_ENULL;
        c.actual[k] = actual[k];
        c.expected[k] = expected[k];
    }
    hipLaunchKernelGGL(check_sizes_kernel, dim3(1), dim3(1), 0, as_stream(stream), c, static_cast<const bool*>(all_kept), self_loops,
                       flag);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_collate_f32(int64_t n_graphs, const int32_t* sel, const int32_t* out_nptr, const int32_t* out_eptr,
                                  const int32_t* src_nptr, const int32_t* src_eptr, const float* x, int64_t x_width,
                                  const float* pos, const int32_t* esrc, const int32_t* edst, int64_t n_out,
                                  int64_t e_out, float* out_x, float* out_pos, int32_t* out_batch, int32_t* out_esrc,
                                  int32_t* out_edst, const float* y, float* out_y, pamnet_stream_t stream) {
    if (n_graphs < 0 || n_out < 0 || e_out < 0 || x_width < 1) return PAMNET_EINVAL;
    if (n_out == 0 && e_out == 0) return PAMNET_OK;
    if (!sel || !out_nptr || !src_nptr || !x || !out_x || !out_batch || (pos && !out_pos) || (y && !out_y)) return PAMNET_ENULL;
    if (e_out > 0 && (!out_eptr || !src_eptr || !esrc || !edst || !out_esrc || !out_edst)) return PAMNET_ENULL;
    Collate c{sel, out_nptr, out_eptr, src_nptr, src_eptr, x, pos, esrc, edst, y, out_y, out_x, out_pos, out_batch, out_esrc,
              out_edst, n_graphs, n_out, e_out, x_width};
    int64_t total = n_out > e_out ? n_out : e_out;
    total = total > n_graphs ? total : n_graphs;
    hipLaunchKernelGGL(collate_kernel, dim3(blocks_for(total)), dim3(256), 0, as_stream(stream), c);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

// ---- input validation ---------------------------------------------------------------------------------------------------
// One launch instead of a dozen tiny tensor ops per batch: flag[0] = 1 when `node_graph` is not sorted / not in
// [0, n_graphs), an atom type (float, read with `type_stride`) is not in [0, n_types), or an edge endpoint is not in [0, n).
namespace {
__global__ __launch_bounds__(256) void validate_inputs_kernel(const int32_t* __restrict__ node_graph, int64_t n,
                                                              int64_t n_graphs, const float* __restrict__ types,
                                                              int64_t type_stride, int64_t n_types,
                                                              const int32_t* __restrict__ src,
                                                              const int32_t* __restrict__ dst, int64_t n_edges,
                                                              int32_t* __restrict__ flag) {
    const int64_t total = n > n_edges ? n : n_edges;
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < n) {
            const int g = node_graph[i];
            bad |= g < 0 || g >= n_graphs || (i > 0 && node_graph[i - 1] > g);
            if (types) {
                const float t = types[i * type_stride];
                bad |= !(t >= 0.f && t < (float)n_types);
            }
        }
        if (i < n_edges) {
            const int a = src[i], b = dst[i];
            bad |= a < 0 || a >= n || b < 0 || b >= n;
        }
    }
    if (bad) flag[0] = 1;
}
}  // namespace

// ---- the calling convention's index tensors in one launch ---------------------------------------------------------------
// `x`, `edge_index`, `batch` of the reference (models.py:104-110) arrive as int64 (PyG) -- or int32 / fp32; the kernels here
// index with int32.  One launch converts all three, builds the per-graph node pointer from the (sorted) batch vector,
// validates every index and notes self loops: the work of seven tensor ops (three casts, !=, all, a counting sort of the
// batch vector, a validation launch) that each cost a 5-7 us launch on the input pipeline's stream.
namespace {
enum { KIND_NONE = 0, KIND_I64 = 1, KIND_I32 = 2, KIND_F32 = 3 };
__device__ __forceinline__ int64_t load_index(const void* p, int kind, int64_t i, bool& bad) {
    if (kind == KIND_I64) return static_cast<const int64_t*>(p)[i];
    if (kind == KIND_I32) return static_cast<const int32_t*>(p)[i];
    const float v = static_cast<const float*>(p)[i];
    if (!(v >= 0.f && v < 2147483648.f)) { bad = true; return -1; }
    return (int64_t)v;                                        // .to(int32) of a non-negative float truncates
}
struct Ingest {
    const void *batch, *x, *esrc, *edst;
    int batch_kind, x_kind, edge_kind;
    int64_t x_stride, n, n_graphs, n_types, n_edges;
    int32_t *node_graph, *gptr, *types, *src, *dst, *flag;   // gptr [n_graphs + 1] and flag [2] arrive zeroed
};
__global__ __launch_bounds__(256) void ingest_kernel(Ingest c) {
    const int64_t total = c.n > c.n_edges ? c.n : c.n_edges;
    bool bad = false, loop = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < c.n) {
            const int64_t g = load_index(c.batch, c.batch_kind, i, bad);
            const int64_t p = i > 0 ? load_index(c.batch, c.batch_kind, i - 1, bad) : -1;
            const bool ok = g >= 0 && g < c.n_graphs && p <= g && p >= -1;
            bad |= !ok;
            c.node_graph[i] = ok ? (int32_t)g : 0;
            if (ok) {
                for (int64_t q = p + 1; q <= g; ++q) c.gptr[q] = (int32_t)i;          // first node of graphs p+1 .. g
                if (i == c.n - 1)
                    for (int64_t q = g + 1; q <= c.n_graphs; ++q) c.gptr[q] = (int32_t)c.n;
            }
            if (c.x_kind != KIND_NONE) {
                const int64_t t = load_index(c.x, c.x_kind, i * c.x_stride, bad);
                const bool tok = t >= 0 && t < c.n_types;
                bad |= !tok;
                c.types[i] = tok ? (int32_t)t : 0;
            }
        }
        if (i < c.n_edges) {
            const int64_t a = load_index(c.esrc, c.edge_kind, i, bad), b = load_index(c.edst, c.edge_kind, i, bad);
            const bool eok = a >= 0 && a < c.n && b >= 0 && b < c.n;
            bad |= !eok;
            loop |= eok && a == b;
            c.src[i] = eok ? (int32_t)a : 0;
            c.dst[i] = eok ? (int32_t)b : 0;
        }
    }
    if (bad) c.flag[0] = 1;
    if (loop) c.flag[1] = 1;
}

__global__ __launch_bounds__(256) void gather2_kernel(const int32_t* __restrict__ perm, const int32_t* __restrict__ a,
                                                      const int32_t* __restrict__ b, int64_t m, int32_t* __restrict__ oa,
                                                      int32_t* __restrict__ ob, const float* __restrict__ pos,
                                                      float* __restrict__ dist) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= m) return;
    const int p = perm[q];
    const bool ok = (uint64_t)p < (uint64_t)m;
    const int va = ok ? a[p] : 0, vb = ok ? b[p] : 0;
    oa[q] = va;
    ob[q] = vb;
    if (dist) dist[q] = dist3(pos, vb, va);                   // ||pos_i - pos_j||, i = b (target), j = a (models.py:65)
}

// transposed edge list from the counting sort's permutation: slot e' takes the query node and the length of original edge
// perm[e'], and the inverse permutation is noted on the way (it IS the transposed CSR of the transposed list: see
// graph.InverseTranspose)
__global__ __launch_bounds__(256) void transpose_gather_kernel(const int32_t* __restrict__ perm,
                                                               const int32_t* __restrict__ q,
                                                               const float* __restrict__ dist, int64_t m,
                                                               int32_t* __restrict__ out_q, float* __restrict__ out_dist,
                                                               int32_t* __restrict__ inv) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    const int p = perm[t];
    const bool ok = (uint64_t)p < (uint64_t)m;
    out_q[t] = ok ? q[p] : 0;
    out_dist[t] = ok ? dist[p] : 0.f;
    if (ok && inv) inv[p] = (int32_t)t;
}
}  // namespace

extern "C" int pamnet_transpose_gather_i32(const int32_t* perm, const int32_t* q, const float* dist, int64_t m,
                                           int32_t* out_q, float* out_dist, int32_t* inv, pamnet_stream_t stream) {
    if (m < 0) return PAMNET_EINVAL;
    if (m == 0) return PAMNET_OK;
    if (!perm || !q || !dist || !out_q || !out_dist) return PAMNET_ENULL;
    hipLaunchKernelGGL(transpose_gather_kernel, dim3(blocks_for(m)), dim3(256), 0, as_stream(stream), perm, q, dist, m,
                       out_q, out_dist, inv);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_ingest_indices_i32(const void* batch, int32_t batch_kind, int64_t n, int64_t n_graphs, const void* x,
                                         int32_t x_kind, int64_t x_stride, int64_t n_types, const void* edge_src,
                                         const void* edge_dst, int32_t edge_kind, int64_t n_edges, int32_t* node_graph,
                                         int32_t* gptr_flag, int32_t* types, int32_t* src, int32_t* dst,
                                         pamnet_stream_t stream) {
    auto kind_ok = [](int k) { return k == KIND_I64 || k == KIND_I32 || k == KIND_F32; };
    if (n < 0 || n_graphs < 0 || n_edges < 0 || n >= (int64_t)1 << 31) return PAMNET_EINVAL;
    if (n > 0 && !kind_ok(batch_kind)) return PAMNET_EINVAL;
    if (x_kind != KIND_NONE && (!kind_ok(x_kind) || x_stride < 1 || n_types < 1)) return PAMNET_EINVAL;
    if (n_edges > 0 && !kind_ok(edge_kind)) return PAMNET_EINVAL;
    if (!gptr_flag || (n > 0 && (!batch || !node_graph)) || (n > 0 && x_kind != KIND_NONE && (!x || !types)) ||
        (n_edges > 0 && (!edge_src || !edge_dst || !src || !dst)))
        return PAMNET_ENULL;
    hipStream_t st = as_stream(stream);
    const hipError_t e = hipMemsetAsync(gptr_flag, 0, sizeof(int32_t) * (n_graphs + 7), st);     // (+ 4 spare words: header)
    if (e != hipSuccess) return (int)e;
    const int64_t total = n > n_edges ? n : n_edges;
    if (total == 0) return PAMNET_OK;
    int64_t blocks = ceil_div(total, 256);
    if (blocks > 1024) blocks = 1024;
    Ingest c{batch, x, edge_src, edge_dst, (int)batch_kind, n > 0 ? (int)x_kind : KIND_NONE, (int)edge_kind, x_stride, n,
             n_graphs, n_types, n_edges, node_graph, gptr_flag, types, src, dst, gptr_flag + n_graphs + 1};
    hipLaunchKernelGGL(ingest_kernel, dim3((unsigned)blocks), dim3(256), 0, st, c);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_gather2_i32(const int32_t* perm, const int32_t* a, const int32_t* b, int64_t m, int32_t* out_a,
                                  int32_t* out_b, const float* pos, float* dist, pamnet_stream_t stream) {
    if (m < 0) return PAMNET_EINVAL;
    if (m == 0) return PAMNET_OK;
    if (!perm || !a || !b || !out_a || !out_b || (dist && !pos)) return PAMNET_ENULL;
    hipLaunchKernelGGL(gather2_kernel, dim3(blocks_for(m)), dim3(256), 0, as_stream(stream), perm, a, b, m, out_a, out_b,
                       pos, dist);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_validate_inputs_i32(const int32_t* node_graph, int64_t n, int64_t n_graphs, const float* types,
                                          int64_t type_stride, int64_t n_types, const int32_t* src, const int32_t* dst,
                                          int64_t n_edges, int32_t* flag, pamnet_stream_t stream) {
    if (n < 0 || n_graphs < 0 || n_edges < 0) return PAMNET_EINVAL;
    if (!flag || (n > 0 && !node_graph) || (n_edges > 0 && (!src || !dst))) return PAMNET_ENULL;
    hipStream_t st = as_stream(stream);
    const hipError_t e = hipMemsetAsync(flag, 0, sizeof(int32_t), st);
    if (e != hipSuccess) return (int)e;
    const int64_t total = n > n_edges ? n : n_edges;
    if (total == 0) return PAMNET_OK;
    int64_t blocks = ceil_div(total, 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(validate_inputs_kernel, dim3((unsigned)blocks), dim3(256), 0, st, node_graph, n, n_graphs, types,
                       type_stride, n_types, src, dst, n_edges, flag);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}
