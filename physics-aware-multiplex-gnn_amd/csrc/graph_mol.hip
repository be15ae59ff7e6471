// Molecule-local graph construction: the whole index / geometry front of PAMNet.forward for the QM9 schema (positions
// + a bond list given: models.py:62-98 triplets / pairs, :104-118 bond lengths + radius graph, :165-177 angles, and the
// transposed index lists of the backward) for batches of SMALL molecules, one wavefront per molecule, two launches.
//
// Every one of these structures is molecule-local -- a radius graph never leaves its molecule, a triplet lives on the
// bonds of one molecule -- and only the offsets of a molecule's slices depend on the rest of the batch.  The step-by-step
// path (graph.hip) nevertheless treats the batch as one flat problem: ~14 launches of scans, counting sorts and per-node /
// per-edge kernels, several of them single-workgroup, 0.2 ms of side-stream time per 128-molecule batch that cost the
// training step 0.11 ms through displaced workgroups of the model's full-chip launches (profiles/r03_step_timeline.txt).
// Here a wavefront holds its molecule in registers and LDS (<= 64 atoms: a lane per atom, adjacency one 64-bit mask per atom;
// <= 256 directed bonds: four per lane),
//   count launch: degrees, triplet / pair rows, radius-graph edges per molecule -> mol_tot [B, 4] and the batch totals;
//   fill launch:  slice offsets = sums over the preceding molecules' totals, then every array, in the SAME order as the
//                 step-by-step path (tests/test_graph_engine.py, tests/test_hip_kernels.py: bit-identical).
// Requirements, checked on the device (a violation is reported in totals[2]; callers then take the step-by-step path):
// bonds grouped by molecule in batch order (what torch_geometric's collation and pamnet_collate_f32 produce), both ends
// of a bond in the same molecule, no molecule over the caps below.  Self loops must have been stripped (the ingest launch
// notes them).
#include "common.h"
#include "geom_core.h"

namespace {

constexpr int MOL_ATOMS = 64, MOL_BONDS = 256, MOL_CHUNKS = MOL_BONDS / 64;

struct MolIn {
    const float* pos;
    const int32_t *gptr, *src, *dst;
    int64_t n, n_graphs, n_bonds;
    float cutoff_g;
    int with_triplets, need_grad;
};

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ int wave_incl(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(v, o, 64);
        if (lane >= o) v += u;
    }
    return v;
}

__device__ __forceinline__ float lane_value(float v, int l) {       // l wave-uniform: one v_readlane, no LDS round trip
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// First bond whose target is >= a0 (lanes 0-31) / >= a1 (lanes 32-63): the bonds are grouped by molecule, so the target's
// molecule does not decrease along the list.  Each half-wave probes 32 split points of its interval per round (three
// dependent loads for 32 k bonds; a lane-serial bisection made 2 x 13 of them, 13 us of a 17 us launch).
__device__ __forceinline__ void bond_range(const int32_t* __restrict__ dst, int m, int a0, int a1, int lane, int& b0, int& b1) {
    const int a = lane < 32 ? a0 : a1, t = lane & 31;
    int lo = 0, hi = m;
    for (int round = 0; round < 8; ++round) {                 // 32^7 > 2^31: terminates for any m
        const bool open = lo < hi;
        if (!__ballot(open)) break;
        const int step = open ? (hi - lo + 31) / 32 : 1;
        const int p = lo + (t + 1) * step - 1;
        const bool probe = open && p < hi;
        const bool less = probe && dst[p] < a;
        const unsigned long long v = __ballot(less);
        const int c = __popc((unsigned)(lane < 32 ? v : (v >> 32)));
        if (open) {
            const int nlo = lo + c * step, nhi = lo + (c + 1) * step - 1;
            lo = nlo;
            hi = nhi < hi ? nhi : hi;
        }
    }
    b0 = __builtin_amdgcn_readlane(lo, 0);
    b1 = __builtin_amdgcn_readlane(lo, 32);
}

template <bool FILL>
__global__ __launch_bounds__(64) void mol_graph_kernel(MolIn in, int32_t* __restrict__ mol_tot,
                                                       int32_t* __restrict__ totals, pamnet_mol_graph_out out,
                                                       int64_t eg_cap, int64_t tp_cap) {
    __shared__ float px[MOL_ATOMS], py[MOL_ATOMS], pz[MOL_ATOMS];
    __shared__ int l_src[MOL_BONDS], l_dst[MOL_BONDS];        // bonds in CSR order of their targets (stable)
    __shared__ int lt[MOL_BONDS];                             // ... listed by source atom (the transposed list)
    __shared__ int lptr[MOL_ATOMS + 1], ltptr[MOL_ATOMS + 1];
    __shared__ int tptr[MOL_BONDS + 1], tcnt[MOL_BONDS];
    __shared__ int gex[MOL_ATOMS];
    __shared__ unsigned long long adj[MOL_ATOMS];
    const int g = blockIdx.x, lane = threadIdx.x;
    const int a0 = in.gptr[g], a1 = in.gptr[g + 1], na = a1 - a0;
    const bool last = g == (int)in.n_graphs - 1;
    int b0, b1;
    int se = 0, st = 0;
    if constexpr (FILL) {
        // the count launch left every molecule's first bond and totals: this molecule's slices start at the sums over the
        // preceding molecules (all of these loads are independent of each other: one round trip)
        b0 = mol_tot[4 * g + 2];
        b1 = last ? (int)in.n_bonds : mol_tot[4 * g + 6];
        for (int q = lane; q < g; q += 64) se += mol_tot[4 * q], st += mol_tot[4 * q + 1];
    } else {
        bond_range(in.dst, (int)in.n_bonds, a0, a1, lane, b0, b1);
    }
    const int nb = b1 - b0;
    int flags = 0;
    if (na < 0 || na > MOL_ATOMS) flags |= 1;
    if (nb < 0 || nb > MOL_BONDS) flags |= 2;
    if (flags) {                                              // wave-uniform
        if (!FILL && lane == 0) {
            mol_tot[4 * g] = mol_tot[4 * g + 1] = 0;
            mol_tot[4 * g + 2] = b0;
            mol_tot[4 * g + 3] = flags;
            atomicOr(&totals[2], flags);
        }
        return;
    }
    float xi = 0.f, yi = 0.f, zi = 0.f;
    if (lane < na) {
        const int64_t a = a0 + lane;
        xi = in.pos[3 * a], yi = in.pos[3 * a + 1], zi = in.pos[3 * a + 2];
        px[lane] = xi, py[lane] = yi, pz[lane] = zi;
    }
    const int nchunk = (nb + 63) >> 6;
    int bs[MOL_CHUNKS], bd[MOL_CHUNKS];                       // bond c * 64 + lane in input order (-1: none)
    bool stray = false;
#pragma unroll
    for (int c = 0; c < MOL_CHUNKS; ++c) {
        const int k = c * 64 + lane;
        bs[c] = bd[c] = -1;
        if (k < nb) {
            int s = in.src[b0 + k] - a0, d = in.dst[b0 + k] - a0;
            if (s < 0 || s >= na || d < 0 || d >= na) stray = true, s = d = 0;
            bs[c] = s, bd[c] = d;
        }
    }
    if (__ballot(stray)) flags |= 4;

    // ---- bonds in CSR order of their targets, stable (as the counting sort of pamnet_csr_from_keys_i32): for atom i the
    // bonds with target i are a ballot; a bond's slot = bonds of earlier atoms + earlier such bonds.  No memory traffic.
    {
        int run = 0, mine = 0;
        for (int i = 0; i < na; ++i) {
            if (lane == i) mine = run;
#pragma unroll
            for (int c = 0; c < MOL_CHUNKS; ++c) {
                if (c < nchunk) {
                    const bool hit = bd[c] == i;
                    const unsigned long long m = __ballot(hit);
                    if (hit) {
                        const int w = run + __popcll(m & ((1ull << lane) - 1ull));
                        l_src[w] = bs[c], l_dst[w] = i;
                    }
                    run += __popcll(m);
                }
            }
        }
        if (lane < na) lptr[lane] = mine;
        if (lane == 0) lptr[na] = nb;
    }
    __syncthreads();
    int ss[MOL_CHUNKS], sd[MOL_CHUNKS];                       // bond c * 64 + lane in CSR order
#pragma unroll
    for (int c = 0; c < MOL_CHUNKS; ++c) {
        const int k = c * 64 + lane;
        ss[c] = k < nb ? l_src[k] : -1;
        sd[c] = k < nb ? l_dst[k] : -1;
    }
    // ---- the same bonds listed by source atom (stable): the transposed bond list, also the index of the triplet transposition
    {
        int run = 0, mine = 0;
        for (int j = 0; j < na; ++j) {
            if (lane == j) mine = run;
#pragma unroll
            for (int c = 0; c < MOL_CHUNKS; ++c) {
                if (c < nchunk) {
                    const bool hit = ss[c] == j;
                    const unsigned long long m = __ballot(hit);
                    if (hit) lt[run + __popcll(m & ((1ull << lane) - 1ull))] = c * 64 + lane;
                    run += __popcll(m);
                }
            }
        }
        if (lane < na) ltptr[lane] = mine;
        if (lane == 0) ltptr[na] = nb;
    }

    // ---- triplet + pair rows per bond e = (j -> i): triplets = bonds (k -> j), k != i; pairs = bonds (j' -> i) incl. e
    int carry = 0;
#pragma unroll
    for (int c = 0; c < MOL_CHUNKS; ++c) {
        if (c < nchunk) {
            const int k = c * 64 + lane;
            int t = 0, cnt = 0;
            if (k < nb) {
                const int j = ss[c], i = sd[c];
                if (in.with_triplets)
                    for (int q = lptr[j]; q < lptr[j + 1]; ++q) t += (l_src[q] != i) ? 1 : 0;
                cnt = t + lptr[i + 1] - lptr[i];
                tcnt[k] = t;
            }
            const int inc = wave_incl(cnt, lane);
            if (k < nb) tptr[k] = carry + inc - cnt;
            carry += __shfl(inc, 63, 64);
        }
    }
    const int tp_m = carry;

    // ---- radius graph inside the molecule: adjacency masks + degrees
    unsigned long long mask = 0ull;
    for (int j = 0; j < na; ++j) {
        const float d = dist3_xyz(xi, yi, zi, lane_value(xi, j), lane_value(yi, j), lane_value(zi, j));
        if (j != lane && lane < na && d <= in.cutoff_g) mask |= 1ull << j;
    }
    const int deg = __popcll(mask);
    const int ginc = wave_incl(deg, lane);
    const int eg_m = __shfl(ginc, 63, 64);

    if (!FILL) {
        if (lane == 0) {
            mol_tot[4 * g] = flags ? 0 : eg_m;
            mol_tot[4 * g + 1] = flags ? 0 : tp_m;
            mol_tot[4 * g + 2] = b0;
            mol_tot[4 * g + 3] = flags;
            if (flags) {
                atomicOr(&totals[2], flags);
            } else {
                atomicAdd(&totals[0], eg_m);
                atomicAdd(&totals[1], tp_m);
                atomicAdd(&totals[3], nb);
            }
        }
        return;
    }
    if (flags) return;                                        // (reported by the count launch)
    adj[lane] = mask;
    gex[lane] = ginc - deg;

    const int64_t eoff = wave_sum(se), toff = wave_sum(st);
    __syncthreads();

    // ---- local (bond) graph: pointer, endpoints, lengths; rows of every bond: its triplets (kind 0), then its pairs (kind 1)
    if (lane < na) out.l_ptr[a0 + lane] = b0 + lptr[lane];
    if (last && lane == 0) {
        out.l_ptr[in.n] = (int32_t)in.n_bonds;
        out.g_ptr[in.n] = (int32_t)(eoff + eg_m < eg_cap ? eoff + eg_m : eg_cap);
        out.t_ptr[in.n_bonds] = (int32_t)(toff + tp_m < tp_cap ? toff + tp_m : tp_cap);
    }
#pragma unroll
    for (int c = 0; c < MOL_CHUNKS; ++c) {
        const int k = c * 64 + lane;
        if (k >= nb) continue;
        const int j = ss[c], i = sd[c];
        const float pix = px[i], piy = py[i], piz = pz[i], pjx = px[j], pjy = py[j], pjz = pz[j];
        out.l_row[b0 + k] = a0 + i;
        out.l_col[b0 + k] = a0 + j;
        out.l_dist[b0 + k] = dist3_xyz(pix, piy, piz, pjx, pjy, pjz);
        int64_t w = toff + tptr[k];
        out.t_ptr[b0 + k] = (int32_t)(w < tp_cap ? w : tp_cap);
        if (in.with_triplets) {
            for (int q = lptr[j]; q < lptr[j + 1]; ++q) {       // pamnet_triplet_fill_f32, same expressions
                const int kk = l_src[q];
                if (kk == i) continue;
                if (w < tp_cap) {
                    out.t_col[w] = b0 + q;
                    out.t_row[w] = b0 + k;
                    out.t_kind[w] = 0;
                    out.t_angle[w] = angle3(pjx - pix, pjy - piy, pjz - piz, px[kk] - pjx, py[kk] - pjy, pz[kk] - pjz);
                }
                ++w;
            }
        }
        for (int q = lptr[i]; q < lptr[i + 1]; ++q) {
            const int j2 = l_src[q];
            if (w < tp_cap) {
                out.t_col[w] = b0 + q;
                out.t_row[w] = b0 + k;
                out.t_kind[w] = 1;
                out.t_angle[w] = angle3(pix - pjx, piy - pjy, piz - pjz, px[j2] - pix, py[j2] - piy, pz[j2] - piz);
            }
            ++w;
        }
    }

    // ---- radius graph: rows = query atom, columns ascending; the reverse-edge index is the transposed list
    if (lane < na) {
        const int64_t w0 = eoff + gex[lane];
        out.g_ptr[a0 + lane] = (int32_t)(w0 < eg_cap ? w0 : eg_cap);
        unsigned long long m = mask;
        int64_t w = w0;
        while (m) {
            const int j = __ffsll((long long)m) - 1;
            m &= m - 1;
            if (w < eg_cap) {
                out.g_row[w] = a0 + lane;
                out.g_col[w] = a0 + j;
                out.g_dist[w] = dist3_xyz(xi, yi, zi, px[j], py[j], pz[j]);
                if (in.need_grad) {                             // (clamped: wrong sizes must stay inside the arrays)
                    const int64_t rv = eoff + gex[j] + __popcll(adj[j] & ((1ull << lane) - 1ull));
                    out.gT_perm[w] = (int32_t)(rv < eg_cap ? rv : eg_cap - 1);
                }
            }
            ++w;
        }
    }
    if (!in.need_grad) return;

    // ---- transposed bond list (what the counting sort of the source column returns)
    if (lane < na) out.lT_ptr[a0 + lane] = b0 + ltptr[lane];
    if (last && lane == 0) {
        out.lT_ptr[in.n] = (int32_t)in.n_bonds;
        out.tT_ptr[in.n_bonds] = (int32_t)(toff + tp_m < tp_cap ? toff + tp_m : tp_cap);
    }
    for (int k = lane; k < nb; k += 64) out.lT_perm[b0 + k] = b0 + lt[k];

    // ---- transposed triplet / pair rows: for source bond q = (k -> j) the rows that gather it, ascending -- its triplet rows
    // belong to the bonds e leaving j towards an atom other than k (row = e's first row + q's rank among e's triplets), its
    // pair rows to the bonds e arriving at j (row = e's first pair row + q's place in j's list): two ascending lists of
    // bonds, merged.  (A counting sort of the row list walks it twice per bond: 20 us for a 250-row molecule.)
    carry = 0;
#pragma unroll
    for (int c = 0; c < MOL_CHUNKS; ++c) {
        if (c >= nchunk) continue;
        const int q = c * 64 + lane;
        const bool on = q < nb;
        const int kq = on ? ss[c] : 0, jq = on ? sd[c] : 0;
        const int ab = ltptr[jq], ae = in.with_triplets ? ltptr[jq + 1] : ab, bb = lptr[jq], be = lptr[jq + 1];
        int cnt = 0;
        if (on) {
            for (int a = ab; a < ae; ++a) cnt += (l_dst[lt[a]] != kq) ? 1 : 0;
            cnt += be - bb;
        }
        const int inc = wave_incl(cnt, lane);
        if (on) {
            int64_t w2 = toff + carry + inc - cnt;
            out.tT_ptr[b0 + q] = (int32_t)(w2 < tp_cap ? w2 : tp_cap);
            int a = ab, b = bb;
            while (true) {
                while (a < ae && l_dst[lt[a]] == kq) ++a;     // (k -> j -> k is no triplet)
                const int ea = a < ae ? lt[a] : 0x7fffffff, eb = b < be ? b : 0x7fffffff;
                if (ea == 0x7fffffff && eb == 0x7fffffff) break;
                int row;
                if (ea < eb) {
                    const int ie = l_dst[ea];
                    int rank = 0;
                    for (int q2 = bb; q2 < q; ++q2) rank += (l_src[q2] != ie) ? 1 : 0;
                    row = tptr[ea] + rank;
                    ++a;
                } else {
                    row = tptr[eb] + tcnt[eb] + (q - bb);
                    ++b;
                }
                if (w2 < tp_cap) out.tT_perm[w2] = (int32_t)(toff + row < tp_cap ? toff + row : tp_cap - 1);
                ++w2;
            }
        }
        carry += __shfl(inc, 63, 64);
    }
}

int check_in(const float* pos, const int32_t* gptr, int64_t n, int64_t n_graphs, const int32_t* src, const int32_t* dst,
             int64_t n_bonds, float cutoff_g) {
    if (n < 0 || n_graphs < 1 || n_bonds < 0 || n >= ((int64_t)1 << 31) || n_bonds >= ((int64_t)1 << 30) || !(cutoff_g >= 0.f))
        return PAMNET_EINVAL;
    if (!pos || !gptr || (n_bonds > 0 && (!src || !dst))) return PAMNET_ENULL;
    return PAMNET_OK;
}

}  // namespace

extern "C" int pamnet_mol_graph_count_i32(const float* pos, const int32_t* gptr, int64_t n, int64_t n_graphs,
                                          const int32_t* src, const int32_t* dst, int64_t n_bonds, float cutoff_g,
                                          int32_t with_triplets, int32_t* mol_tot, int32_t* totals,
                                          pamnet_stream_t stream) {
    const int rc = check_in(pos, gptr, n, n_graphs, src, dst, n_bonds, cutoff_g);
    if (rc) return rc;
    if (!mol_tot || !totals) return PAMNET_ENULL;
    const MolIn in{pos, gptr, src, dst, n, n_graphs, n_bonds, cutoff_g, with_triplets ? 1 : 0, 0};
    const pamnet_mol_graph_out none{};
    hipLaunchKernelGGL((mol_graph_kernel<false>), dim3((unsigned)n_graphs), dim3(64), 0, as_stream(stream), in, mol_tot, totals,
                       none, (int64_t)0, (int64_t)0);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}

extern "C" int pamnet_mol_graph_fill_i32(const float* pos, const int32_t* gptr, int64_t n, int64_t n_graphs,
                                         const int32_t* src, const int32_t* dst, int64_t n_bonds, float cutoff_g,
                                         int32_t with_triplets, int32_t need_grad, const int32_t* mol_tot, int64_t eg_cap,
                                         int64_t tp_cap, const pamnet_mol_graph_out* out, pamnet_stream_t stream) {
    const int rc = check_in(pos, gptr, n, n_graphs, src, dst, n_bonds, cutoff_g);
    if (rc) return rc;
    if (!mol_tot || !out) return PAMNET_ENULL;
    if (eg_cap < 0 || tp_cap < 0) return PAMNET_EINVAL;
    const pamnet_mol_graph_out& o = *out;
    if (!o.g_ptr || !o.l_ptr || !o.t_ptr) return PAMNET_ENULL;
    if (eg_cap > 0 && (!o.g_row || !o.g_col || !o.g_dist || (need_grad && !o.gT_perm))) return PAMNET_ENULL;
    if (n_bonds > 0 && (!o.l_row || !o.l_col || !o.l_dist || (need_grad && !o.lT_perm))) return PAMNET_ENULL;
    if (tp_cap > 0 && (!o.t_row || !o.t_col || !o.t_angle || !o.t_kind || (need_grad && !o.tT_perm))) return PAMNET_ENULL;
    if (need_grad && (!o.lT_ptr || !o.tT_ptr)) return PAMNET_ENULL;
    const MolIn in{pos, gptr, src, dst, n, n_graphs, n_bonds, cutoff_g, with_triplets ? 1 : 0, need_grad ? 1 : 0};
    hipLaunchKernelGGL((mol_graph_kernel<true>), dim3((unsigned)n_graphs), dim3(64), 0, as_stream(stream), in,
                       const_cast<int32_t*>(mol_tot), (int32_t*)nullptr, o, eg_cap, tp_cap);
    PAMNET_LAUNCH_CHECK();
    return PAMNET_OK;
}
