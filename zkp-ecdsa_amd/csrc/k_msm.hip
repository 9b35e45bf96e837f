// Cross-proof batched check of the verifier's Tom-256 relations (SURVEY.md section 8 row f: "next").
//
// verifySignatureList's boolean is "every relation of the proof holds" (src/exp/exp.ts:267-346, exp/pointAdd.ts:215-255,
// commit/mult.ts:158-173, commit/equality.ts:103-114, proofGK/gk.ts:223-259); the engine already folds the relations of
// ONE proof into sums  sum_i s_i P_i = identity  with independent 128-bit multipliers per relation (k_verify.hip).  The
// multipliers of different proofs are independent too, so the sum over ALL proofs of a chunk is the identity iff
// (up to 2^-128) every proof's sums are.  One multi-scalar multiplication over the chunk's ~15 M live terms with the
// bucket method costs 16 (8 for 128-bit scalars) mixed additions per term instead of the 65 + 7 (33 + 7) of a per-proof
// windowed Straus sum.  If the chunk's total is the identity every proof passed its Tom relations; if not, the caller
// falls back to the per-proof sums to find out which ones failed -- the verdicts are the same either way, only the cost
// of a chunk that contains a bad proof doubles.
//
// Pipeline per chunk (one HIP stream, two host round trips: the live count and the verdict):
//   k_msm_pack      live terms -> 128-byte AoS niels entries (the bucket sums gather them)
//   k_msm_compact   ids of the live terms + their sixteen 16-bit digits, one atomic per workgroup
//   per window w:   rocprim::radix_sort_pairs (digit -> term id), k_msm_bounds first / last position of every digit
//   k_msm_bucket    thread (window, digit): sum of its terms (8 modmuls per term); buckets far above the average
//                   (k_msm_bucket_big / _big2) are summed by slices over many workgroups
//   k_msm_reduce1/2/3  sum_d d * B_d per window by two levels of running sums, times 2^(16 w)
//   k_msm_coef, one fixed-base commitment, k_msm_final: windows + fixed-base part == identity ?
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include "engine.h"

// Two shapes of the same pass, chosen per context (zk_ctx_set_verify_groups): 8 groups x 16-bit windows (16 windows) or 64 groups x
// 13-bit windows (20 windows).  Either way a window has 2^19 (group, digit) buckets and the sort keys are 19 bits wide, so every
// buffer keeps its size; the finer shape costs 25 % more bucket additions and four more sorts on every chunk and makes a forged
// proof cost a 64th of its chunk instead of an eighth.  C is a template parameter: the digit cuts and loop bounds stay constants.
#define MSM_KEY_BITS 19
#define MSM_NBG (1u << MSM_KEY_BITS)   // (group, digit) values per window
#define MSM_ENTRY_WORDS 32
template <int C>
struct MsmShape {
    static constexpr uint32_t c = C, nb = 1u << C, nw = (256 + C - 1) / C, gbits = MSM_KEY_BITS - C, g = 1u << gbits;
    static constexpr uint32_t nwg = nw * g;        // (window, group) pairs: the reductions treat each as a window of its own
    static constexpr uint32_t l1 = nb / 64;        // level-1 ranges of 64 buckets per (window, group)
    static constexpr uint32_t l2 = l1 / 32;        // level-2 ranges of 32 level-1 ranges
    static_assert(nw <= MSM_NW_MAX && g <= MSM_G_MAX && l2 >= 1, "shape");
};

// term id space: [0, n0) slot_terms, [n0, n0 + n1) gk_terms, [n0 + n1, n0 + n1 + n2) misc_terms
struct MsmDims {
    uint32_t n0, n1, n2;        // capacities (ids)
    uint32_t g0, g1, g2;        // group strides (terms k of group g sit at k * stride + g)
    uint32_t l0, l1, l2;        // live groups of this chunk
    uint32_t nq, gsz;           // gk groups per proof; proofs per MSM group
};
ZK_DEV const VTerms& msm_list(const VWork& V, const MsmDims& D, uint32_t id, uint32_t& idx, bool& live, uint32_t& proof) {
    if (id < D.n0) {
        uint32_t g = id % D.g0;
        idx = id, live = g < D.l0, proof = g / VK;
        return V.slot_terms;
    }
    if (id < D.n0 + D.n1) {
        idx = id - D.n0;
        uint32_t g = idx % D.g1;
        live = g < D.l1, proof = g / D.nq;
        return V.gk_terms;
    }
    idx = id - D.n0 - D.n1;
    proof = idx % D.g2, live = proof < D.l2;
    return V.misc_terms;
}
__global__ void __launch_bounds__(256) k_msm_pack(VWork V, MsmDims D, uint32_t* aos) {
    uint32_t id = gtid();
    if (id >= D.n0 + D.n1 + D.n2) return;
    uint32_t idx, proof;
    bool live;
    const VTerms& L = msm_list(V, D, id, idx, live, proof);
    if (!live || fe_is_zero(soa_ld<ModQ, 1>(L.sc, idx))) return;
    Ft2 x = soa_ld<ModT, 2>(L.nx, idx), y = soa_ld<ModT, 2>(L.ny, idx), dt = soa_ld<ModT, 2>(L.ndt, idx);
    uint32_t w[28];
#pragma unroll
    for (int l = 0; l < 9; l++) w[l] = x.l[l], w[9 + l] = y.l[l], w[18 + l] = dt.l[l];
    w[27] = 0;
    uint4* q = (uint4*)(aos + (size_t)id * MSM_ENTRY_WORDS);
#pragma unroll
    for (int i = 0; i < 7; i++) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
// Compaction of the live terms (scalar != 0): ids[pos] = term id and, for every window, keys[w * cap + pos] = group << 16 | its
// 16-bit digit (0 included: digit 0 is simply a bucket nobody sums).  One atomic per workgroup.
template <int C>
__global__ void __launch_bounds__(256) k_msm_compact(VWork V, MsmDims D, uint32_t cap, uint32_t* keys, uint32_t* ids, uint32_t* counter) {
    typedef MsmShape<C> S;
    __shared__ uint32_t wave_cnt[4], block_base;
    uint32_t id = gtid();
    uint32_t w8[8], proof = 0;
    bool act = false;
    if (id < D.n0 + D.n1 + D.n2) {
        uint32_t idx;
        bool live;
        const VTerms& L = msm_list(V, D, id, idx, live, proof);
        if (live) {
            Fe<ModQ, 1> sc = soa_ld<ModQ, 1>(L.sc, idx);
            act = !fe_is_zero(sc);
            words_from_limbs<8>(w8, sc.l);
        }
    }
    uint64_t mask = __ballot(act);
    uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
    if (lane == 0) wave_cnt[wv] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        block_base = tot ? atomicAdd(counter, tot) : 0;
    }
    __syncthreads();
    if (!act) return;
    uint32_t pos = block_base + below;
    for (uint32_t k = 0; k < wv; k++) pos += wave_cnt[k];
    ids[pos] = id;
    const uint32_t grp = (proof / D.gsz) << S::c;
#pragma unroll
    for (int w = 0; w < (int)S::nw; w++) {   // bits [C w, C (w + 1)) of the 256-bit scalar: word index and shift are constants
        const int bit = C * w, k = bit >> 5, sh = bit & 31;
        uint32_t d = w8[k] >> sh;
        if (sh + C > 32 && k + 1 < 8) d |= w8[k + 1] << (32 - sh);
        keys[(size_t)w * cap + pos] = grp | (d & (S::nb - 1));
    }
}
__global__ void __launch_bounds__(256) k_msm_bounds(const uint32_t* __restrict__ keys, uint32_t n, uint32_t* start, uint32_t* end) {
    uint32_t i = gtid();
    if (i >= n) return;
    uint32_t k = keys[i];
    if (i == 0 || keys[i - 1] != k) start[k] = i;
    if (i + 1 == n || keys[i + 1] != k) end[k] = i + 1;
}
ZK_DEV TomNiels msm_ld(const uint32_t* e) {
    const uint4* q = (const uint4*)e;
    uint32_t w[28];
#pragma unroll
    for (int i = 0; i < 7; i++) {
        uint4 v = q[i];
        w[4 * i] = v.x, w[4 * i + 1] = v.y, w[4 * i + 2] = v.z, w[4 * i + 3] = v.w;
    }
    TomNiels n;
#pragma unroll
    for (int l = 0; l < 9; l++) n.x.l[l] = w[l], n.y.l[l] = w[9 + l], n.dt.l[l] = w[18 + l];
    return n;
}
ZK_DEV void msm_st(uint32_t* p, const TomPt& a) {
#pragma unroll
    for (int l = 0; l < 9; l++) p[l] = a.x.l[l], p[9 + l] = a.y.l[l], p[18 + l] = a.t.l[l], p[27 + l] = a.z.l[l];
}
ZK_DEV TomPt msm_ldp(const uint32_t* p) {
    TomPt a;
#pragma unroll
    for (int l = 0; l < 9; l++) a.x.l[l] = p[l], a.y.l[l] = p[9 + l], a.t.l[l] = p[18 + l], a.z.l[l] = p[27 + l];
    return a;
}
// buckets[(w * 65536 + d) * 36]: extended point.  vals: sorted term ids of window w at vals + w * cap.
// Digits are not always uniform (scalars with a structured part put thousands of terms into one bucket): a bucket with more
// than `big` terms (8 x the window's average + 64) is left to k_msm_bucket_big, one workgroup per such bucket, so that no lane
// walks a long list alone.
#define MSM_BIG_MAX 4096u
// A lane sums one bucket, so a wave takes as long as its largest bucket: with Poisson-sized buckets (mean 27 in the low windows, 10
// in the high ones) a wave of 64 neighbouring digits waits for a bucket 1.5-1.8x the mean.  The buckets of all windows are therefore
// ordered by size first (k_msm_sizes + ONE 8-bit radix pass over 8.4 M (key, id) pairs), largest first: the lanes of a wave get
// buckets of the same size, the empty ones end up together at the end.
template <int C>
__global__ void __launch_bounds__(256) k_msm_sizes(const uint32_t* __restrict__ start, const uint32_t* __restrict__ end, uint32_t* key, uint32_t* id) {
    uint32_t wd = gtid();   // w * MSM_NBG + (group << C | digit)
    uint32_t n = (wd & (MsmShape<C>::nb - 1)) != 0 ? end[wd] - start[wd] : 0;
    key[wd] = 255u - (n < 255u ? n : 255u), id[wd] = wd;
}
template <int C>
__global__ void __launch_bounds__(256) k_msm_bucket(const uint32_t* __restrict__ aos, const uint32_t* __restrict__ vals, uint32_t cap,
                                                    const uint32_t* __restrict__ start, const uint32_t* __restrict__ end, const uint32_t* __restrict__ order,
                                                    uint32_t* buckets, uint32_t* big_cnt, uint32_t* big_list, uint32_t big) {
    uint32_t wd = order[gtid()], w = wd / MSM_NBG, d = wd % MSM_NBG;   // d = group << C | digit
    uint32_t s = start[wd], e = end[wd];
    TomPt acc = tom_identity();
    const bool nz = (d & (MsmShape<C>::nb - 1)) != 0;
    if (nz && e - s > big) {
        uint32_t pos = atomicAdd(big_cnt, 1u);
        if (pos < MSM_BIG_MAX) big_list[pos] = wd, e = s;  // handled by k_msm_bucket_big (beyond the list: here after all)
    }
    if (nz && e > s) {
        const uint32_t* v = vals + (size_t)w * cap;
        TomNiels nx = msm_ld(aos + (size_t)v[s] * MSM_ENTRY_WORDS);
        acc = tom_from_niels(nx);
#pragma unroll 1
        for (uint32_t i = s + 1; i < e; i++) {
            nx = msm_ld(aos + (size_t)v[i] * MSM_ENTRY_WORDS);
            acc = tom_add_niels(acc, nx);
        }
    }
    msm_st(buckets + (size_t)wd * 36, acc);
}
// Oversized buckets (sums of a few 208-bit products put ~6 terms per proof into digits 1..3 of window 13): block (j, b)
// sums slice j (MSM_SLICE terms, strided over the grid's x extent) of big bucket b into part[b * MSM_NSLICE + j];
// k_msm_bucket_big2 adds a bucket's partial sums.
#define MSM_SLICE 2048u
#define MSM_NSLICE 128u
ZK_DEV TomPt msm_block_sum(TomPt acc, uint32_t* sh) {  // all 256 lanes: tree over the block, result valid in lane 0
    uint32_t t = threadIdx.x;
    for (uint32_t o = 128; o >= 1; o >>= 1) {
        __syncthreads();
        if (t >= o && t < 2 * o) msm_st(sh + (size_t)(t - o) * 36, acc);
        __syncthreads();
        if (t < o) acc = tom_add(acc, msm_ldp(sh + (size_t)t * 36));
    }
    __syncthreads();
    return acc;
}
__global__ void __launch_bounds__(256) k_msm_bucket_big(const uint32_t* __restrict__ aos, const uint32_t* __restrict__ vals, uint32_t cap,
                                                        const uint32_t* __restrict__ start, const uint32_t* __restrict__ end,
                                                        const uint32_t* __restrict__ big_cnt, const uint32_t* __restrict__ big_list, uint32_t* part) {
    __shared__ uint32_t sh[128 * 36];
    uint32_t n = *big_cnt < MSM_BIG_MAX ? *big_cnt : MSM_BIG_MAX;
    for (uint32_t b = blockIdx.y; b < n; b += gridDim.y) {
        uint32_t wd = big_list[b], w = wd / MSM_NBG;
        uint32_t s = start[wd], e = end[wd], t = threadIdx.x;
        const uint32_t* v = vals + (size_t)w * cap;
        TomPt acc = tom_identity();
        bool first = true;
        for (uint32_t base = s + blockIdx.x * MSM_SLICE; base < e; base += MSM_NSLICE * MSM_SLICE) {
            uint32_t lim = base + MSM_SLICE < e ? base + MSM_SLICE : e;
#pragma unroll 1
            for (uint32_t i = base + t; i < lim; i += 256) {
                TomNiels nx = msm_ld(aos + (size_t)v[i] * MSM_ENTRY_WORDS);
                acc = first ? tom_from_niels(nx) : tom_add_niels(acc, nx);
                first = false;
            }
        }
        acc = msm_block_sum(acc, sh);
        if (t == 0) msm_st(part + ((size_t)b * MSM_NSLICE + blockIdx.x) * 36, acc);
    }
}
__global__ void __launch_bounds__(256) k_msm_bucket_big2(const uint32_t* __restrict__ big_cnt, const uint32_t* __restrict__ big_list, const uint32_t* __restrict__ part, uint32_t* buckets) {
    __shared__ uint32_t sh[128 * 36];
    uint32_t n = *big_cnt < MSM_BIG_MAX ? *big_cnt : MSM_BIG_MAX;
    for (uint32_t b = blockIdx.x; b < n; b += gridDim.x) {
        TomPt acc = threadIdx.x < MSM_NSLICE ? msm_ldp(part + ((size_t)b * MSM_NSLICE + threadIdx.x) * 36) : tom_identity();
        acc = msm_block_sum(acc, sh);
        if (threadIdx.x == 0) msm_st(buckets + (size_t)big_list[b] * 36, acc);
    }
}
// level 1: 64 buckets per thread.  F1 = sum_j j * B_{64 r + j}, G1 = sum_j B_{64 r + j}
template <int C>
__global__ void __launch_bounds__(256) k_msm_reduce1(const uint32_t* __restrict__ buckets, uint32_t* F1, uint32_t* G1) {
    typedef MsmShape<C> S;
    uint32_t r = blockIdx.x * blockDim.x + threadIdx.x, w = blockIdx.y;  // r < l1; w = window * groups + group
    if (r >= S::l1) return;
    const uint32_t* b = buckets + ((size_t)w * S::nb + 64 * r) * 36;
    TomPt run = tom_identity(), acc = tom_identity();
#pragma unroll 1
    for (int j = 63; j >= 1; j--) {
        run = tom_add(run, msm_ldp(b + 36 * j));
        acc = tom_add(acc, run);
    }
    run = tom_add(run, msm_ldp(b));
    msm_st(F1 + ((size_t)w * S::l1 + r) * 36, acc);
    msm_st(G1 + ((size_t)w * S::l1 + r) * 36, run);
}
// level 2: 32 level-1 ranges per thread.  F2 = sum_j j * G1_{32 s + j}, G2 = sum_j G1_{32 s + j}, H2 = sum_j F1_{32 s + j}
template <int C>
__global__ void __launch_bounds__(64) k_msm_reduce2(const uint32_t* __restrict__ F1, const uint32_t* __restrict__ G1, uint32_t* F2, uint32_t* G2, uint32_t* H2) {
    typedef MsmShape<C> S;
    uint32_t t = gtid();
    if (t >= S::nwg * S::l2) return;
    uint32_t w = t / S::l2, s = t % S::l2;
    const uint32_t* g = G1 + ((size_t)w * S::l1 + 32 * s) * 36;
    const uint32_t* f = F1 + ((size_t)w * S::l1 + 32 * s) * 36;
    TomPt run = tom_identity(), acc = tom_identity(), h = tom_identity();
#pragma unroll 1
    for (int j = 31; j >= 1; j--) {
        run = tom_add(run, msm_ldp(g + 36 * j));
        acc = tom_add(acc, run);
        h = tom_add(h, msm_ldp(f + 36 * j));
    }
    run = tom_add(run, msm_ldp(g));
    h = tom_add(h, msm_ldp(f));
    msm_st(F2 + (size_t)t * 36, acc), msm_st(G2 + (size_t)t * 36, run), msm_st(H2 + (size_t)t * 36, h);
}
// level 3: one thread per window.  sum_d d B_d = H + 64 (GF2 + 32 FG2);  result times 2^(16 w)
template <int C>
__global__ void __launch_bounds__(64) k_msm_reduce3(const uint32_t* __restrict__ F2, const uint32_t* __restrict__ G2, const uint32_t* __restrict__ H2, uint32_t* Tw) {
    typedef MsmShape<C> S;
    uint32_t w = gtid();   // window * groups + group
    if (w >= S::nwg) return;
    TomPt run = tom_identity(), fg = tom_identity(), gf = tom_identity(), gh = tom_identity();
#pragma unroll 1
    for (int s = (int)S::l2 - 1; s >= 0; s--) {
        size_t o = ((size_t)w * S::l2 + s) * 36;
        if (s >= 1) {
            run = tom_add(run, msm_ldp(G2 + o));
            fg = tom_add(fg, run);
        }
        gf = tom_add(gf, msm_ldp(F2 + o));
        gh = tom_add(gh, msm_ldp(H2 + o));
    }
    for (int i = 0; i < 5; i++) fg = tom_dbl(fg);
    TomPt t = tom_add(gf, fg);
    for (int i = 0; i < 6; i++) t = tom_dbl(t);
    t = tom_add(t, gh);
#pragma unroll 1
    for (uint32_t i = 0; i < S::c * (w / S::g); i++) t = tom_dbl(t);
    msm_st(Tw + (size_t)w * 36, t);
}
// coefficient sums of the fixed bases over a group's proofs (block g): list C slots p * 4n + {0, 1} hold (mg, mh), (eg, eh)
__global__ void __launch_bounds__(256) k_msm_coef(Workspace W, uint32_t count, uint32_t gsz, TomList one) {
    __shared__ uint32_t sh[2][9][256];
    uint32_t t = threadIdx.x, g = blockIdx.x;
    uint32_t p0 = g * gsz, p1 = p0 + gsz < count ? p0 + gsz : count;
    Fe<ModQ, 1> gg = fe_zero<ModQ>(), h = fe_zero<ModQ>();
    for (uint32_t p = p0 + t; p < p1; p += 256) {
        uint32_t lc = p * 4 * W.n;
        gg = fe_add_mod(gg, fe_add_mod(soa_ld<ModQ, 1>(W.lc.v, lc), soa_ld<ModQ, 1>(W.lc.v, lc + 1)));
        h = fe_add_mod(h, fe_add_mod(soa_ld<ModQ, 1>(W.lc.r, lc), soa_ld<ModQ, 1>(W.lc.r, lc + 1)));
    }
    for (int l = 0; l < 9; l++) sh[0][l][t] = gg.l[l], sh[1][l][t] = h.l[l];
    __syncthreads();
    for (uint32_t o = 128; o >= 1; o >>= 1) {
        if (t < o) {
            Fe<ModQ, 1> a, b, c2, d2;
            for (int l = 0; l < 9; l++) a.l[l] = sh[0][l][t], b.l[l] = sh[0][l][t + o], c2.l[l] = sh[1][l][t], d2.l[l] = sh[1][l][t + o];
            a = fe_add_mod(a, b), c2 = fe_add_mod(c2, d2);
            for (int l = 0; l < 9; l++) sh[0][l][t] = a.l[l], sh[1][l][t] = c2.l[l];
        }
        __syncthreads();
    }
    if (t == 0) {
        Fe<ModQ, 1> a, c2;
        for (int l = 0; l < 9; l++) a.l[l] = sh[0][l][0], c2.l[l] = sh[1][l][0];
        soa_st(one.v, g, a), soa_st(one.r, g, c2);
    }
}
template <int C>
__global__ void k_msm_final(const uint32_t* __restrict__ Tw, TomList one, uint32_t* flag) {
    typedef MsmShape<C> S;
    uint32_t g = gtid();
    if (g >= S::g) return;
    Ft2 x = soa_ld<ModT, 2>(one.proj.x, g), y = soa_ld<ModT, 2>(one.proj.y, g), z = soa_ld<ModT, 2>(one.proj.z, g);
    TomPt t;
    t.x = x * z, t.y = y * z, t.t = x * y, t.z = z * z;  // (X : Y : Z) -> extended
    for (uint32_t w = 0; w < S::nw; w++) t = tom_add(t, msm_ldp(Tw + ((size_t)w * S::g + g) * 36));
    bool id = fe_is_zero(t.x) && fe_eq(t.y, t.z) && !fe_is_zero(t.z);
    flag[g] = id ? 1u : 0u;
}

size_t msm_workspace_bytes(uint32_t cap) {
    size_t tmp = 0;
    rocprim::radix_sort_pairs(nullptr, tmp, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, cap, 0, MSM_KEY_BITS);
    size_t tmp2 = 0;   // the bucket ordering pass
    rocprim::radix_sort_pairs(nullptr, tmp2, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, MSM_NW_MAX * MSM_NBG, 0, 8);
    return tmp > tmp2 ? tmp : tmp2;
}
// returns through host_flags[groups] (after a stream synchronisation): 1 = the Tom total of that group of proofs is the identity
template <int C>
static hipError_t run_msm_t(hipStream_t s, const DevParams& P, const Workspace& W, const VWork& V, uint32_t count, uint32_t nq, const MsmBuf& M, uint32_t* host_flags, uint32_t* gsz_out,
                            hipEvent_t ev0, hipEvent_t ev1) {
    typedef MsmShape<C> S;
    MsmDims D;
    D.g0 = V.C * VK, D.g1 = V.C * nq, D.g2 = V.C;
    D.n0 = D.g0 * V_SLOT_TERMS, D.n1 = D.g1 * 8, D.n2 = D.g2 * 3;
    D.l0 = count * VK, D.l1 = count * nq, D.l2 = count;
    D.nq = nq, D.gsz = (count + S::g - 1) / S::g;
    *gsz_out = D.gsz;
    const uint32_t total = D.n0 + D.n1 + D.n2;
    const bool dbg = getenv("ZK_MSM_DEBUG") != nullptr;  // phase timings on stderr (adds stream synchronisations)
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now();
    hipLaunchKernelGGL(k_msm_pack, dim3((total + 255) / 256), dim3(256), 0, s, V, D, M.aos);
    hipMemsetAsync(M.start, 0, sizeof(uint32_t) * S::nw * MSM_NBG, s);
    hipMemsetAsync(M.end, 0, sizeof(uint32_t) * S::nw * MSM_NBG, s);
    hipMemsetAsync(M.counters, 0, sizeof(uint32_t) * 32, s);
    hipLaunchKernelGGL(k_msm_compact<C>, dim3((total + 255) / 256), dim3(256), 0, s, V, D, M.cap, M.keys_all, M.vals_in, M.counters);
    launch_words_to_host(s, M.host, M.counters, 1);
    hipError_t e0 = hipStreamSynchronize(s);
    if (e0 != hipSuccess) return e0;
    const uint32_t n = M.host[0];
    const uint32_t nmax = n;
    if (dbg) fprintf(stderr, "msm: %u live terms, pack+compact %.2f ms\n", n, now() - t0), t0 = now();
    for (uint32_t w = 0; w < S::nw && n; w++) {
        size_t tmp = M.sort_tmp_bytes;
        hipError_t e = rocprim::radix_sort_pairs(M.sort_tmp, tmp, M.keys_all + (size_t)w * M.cap, M.keys_out, M.vals_in, M.vals_out + (size_t)w * M.cap, n, 0,
                                                 MSM_KEY_BITS, s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_msm_bounds, dim3((n + 255) / 256), dim3(256), 0, s, M.keys_out, n, M.start + (size_t)w * MSM_NBG, M.end + (size_t)w * MSM_NBG);
    }
    if (dbg) {
        hipStreamSynchronize(s);
        fprintf(stderr, "msm: %u sorts + bounds %.2f ms\n", S::nw, now() - t0), t0 = now();
    }
    hipMemsetAsync(M.counters + 32, 0, 4, s);
    {
        hipLaunchKernelGGL(k_msm_sizes<C>, dim3(S::nw * MSM_NBG / 256), dim3(256), 0, s, M.start, M.end, M.ord_key, M.ord_id);
        size_t tmp = M.sort_tmp_bytes;
        hipError_t e = rocprim::radix_sort_pairs(M.sort_tmp, tmp, M.ord_key, M.ord_key2, M.ord_id, M.ord_id2, S::nw * MSM_NBG, 0, 8, s);
        if (e != hipSuccess) return e;
    }
    if (ev0) hipEventRecord(ev0, s);   // the bucket sums alone (bench.py: roofline.others)
    hipLaunchKernelGGL(k_msm_bucket<C>, dim3(S::nw * MSM_NBG / 256), dim3(256), 0, s, M.aos, M.vals_out, M.cap, M.start, M.end, M.ord_id2, M.buckets, M.counters + 32,
                       M.big_list, 8 * ((nmax + MSM_NBG - 1) / MSM_NBG) + 64);
    if (ev1) hipEventRecord(ev1, s);
    hipLaunchKernelGGL(k_msm_bucket_big, dim3(MSM_NSLICE, 32), dim3(256), 0, s, M.aos, M.vals_out, M.cap, M.start, M.end, M.counters + 32, M.big_list, M.big_part);
    hipLaunchKernelGGL(k_msm_bucket_big2, dim3(64), dim3(256), 0, s, M.counters + 32, M.big_list, M.big_part, M.buckets);
    constexpr uint32_t bt = S::l1 < 256 ? S::l1 : 256;
    hipLaunchKernelGGL(k_msm_reduce1<C>, dim3(S::l1 / bt, S::nwg), dim3(bt), 0, s, M.buckets, M.F1, M.G1);
    hipLaunchKernelGGL(k_msm_reduce2<C>, dim3((S::nwg * S::l2 + 63) / 64), dim3(64), 0, s, M.F1, M.G1, M.F2, M.G2, M.H2);
    hipLaunchKernelGGL(k_msm_reduce3<C>, dim3((S::nwg + 63) / 64), dim3(64), 0, s, M.F2, M.G2, M.H2, M.Tw);
    hipLaunchKernelGGL(k_msm_coef, dim3(S::g), dim3(256), 0, s, W, count, D.gsz, M.one);
    launch_tom_commit(s, P, M.one, S::g, 1, 1);
    hipLaunchKernelGGL(k_msm_final<C>, dim3(1), dim3(64), 0, s, M.Tw, M.one, M.flag);
    launch_words_to_host(s, M.host + 8, M.flag, S::g);
    hipError_t e = hipStreamSynchronize(s);
    for (uint32_t g = 0; g < S::g; g++) host_flags[g] = M.host[8 + g];
    if (dbg) fprintf(stderr, "msm tail (bucket .. final) %.2f ms\n", now() - t0);
    return e;
}
hipError_t run_msm(hipStream_t s, const DevParams& P, const Workspace& W, const VWork& V, uint32_t count, uint32_t nq, const MsmBuf& M, uint32_t groups, uint32_t* host_flags,
                   uint32_t* gsz_out, hipEvent_t ev0, hipEvent_t ev1) {
    return groups == 64 ? run_msm_t<13>(s, P, W, V, count, nq, M, host_flags, gsz_out, ev0, ev1) : run_msm_t<16>(s, P, W, V, count, nq, M, host_flags, gsz_out, ev0, ev1);
}
