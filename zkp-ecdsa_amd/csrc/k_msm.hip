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
// Pipeline per chunk (one HIP stream, ONE host round trip: the verdicts):
//   (the terms' 128-byte niels entries are written by the kernels that parse the proofs, k_verify.hip: vt_st; rounds 2-4 copied them here, k_msm_pack)
//   grouping of the (window, group, digit) keys of all live terms, hand-written (round 5; rounds 2-4 ran 16 rocprim::radix_sort_pairs + 96 bounds
//   kernels per chunk, 166 launches and 12 ms per 65 536 proofs): the key space is fixed (19 bits per window), so two counting passes do:
//     k_msm_hist      per workgroup (a contiguous range of term ids) and window: LDS histogram of the keys' top 9 bits ("bins"); digit 0 is dropped
//     k_msm_binscan1/2  where every (window, bin, workgroup) run starts
//     k_msm_scatter   the same walk again: (key, term id) pairs into their bins, positions from LDS cursors (no global atomics)
//     k_msm_binsort   one workgroup per (window, bin): counting sort by the key's low 10 bits in LDS; writes the sorted term ids, first / last position
//                     of every (group, digit) value, and its 1024 buckets ordered by size (what k_msm_bucket's waves want)
//   k_msm_bucket    thread (window, digit): sum of its terms (8 modmuls per term); buckets far above the average
//                   (k_msm_bucket_big / _big2) are summed by slices over many workgroups
//   k_msm_red1 / _redk x 3 / _red_last  sum_d d * B_d per (window, group) by four levels of running sums over sixteen entries each, times 2^(16 w)
//   k_msm_coef, one fixed-base commitment, k_msm_final: windows + fixed-base part == identity ?
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "engine.h"
#include "coop_dev.h"

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
    static_assert(nw <= MSM_NW_MAX && g <= MSM_G_MAX, "shape");
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
// ---------------------------------------------------------------- grouping of the keys (counting sort in two passes; no library sort)
// A term's key in window w is  group << C | digit_w  (MSM_KEY_BITS = 19 bits); a term whose digit is 0 in a window takes no part in it (bucket 0 is
// never summed).  Pass A partitions the (key, term id) pairs of every window by the key's top MSM_HB bits (a "bin"), pass B sorts every bin by
// the remaining bits.  Both passes count in LDS and place by LDS cursors; between them only prefix sums travel.
#define MSM_HB 9
#define MSM_NBIN (1u << MSM_HB)                 // bins per window
#define MSM_LB (MSM_KEY_BITS - MSM_HB)          // low key bits, sorted by pass B
#define MSM_NLOW (1u << MSM_LB)                 // keys per bin
#define MSM_SORT_G 512u                         // workgroups of pass A: each owns a contiguous range of term ids
#define MSM_SORT_T 1024u                        // their threads
// the scalar of term `id` as eight 32-bit words if the term takes part in the chunk's sum (live group, scalar != 0)
ZK_DEV bool msm_term_words(const VWork& V, const MsmDims& D, uint32_t id, uint32_t w8[8], uint32_t& proof) {
    if (id >= D.n0 + D.n1 + D.n2) return false;
    uint32_t idx;
    bool live;
    const VTerms& L = msm_list(V, D, id, idx, live, proof);
    if (!live) return false;
    const Fe<ModQ, 1> sc = soa_ld<ModQ, 1>(L.sc, idx);
    if (fe_is_zero(sc)) return false;
    words_from_limbs<8>(w8, sc.l);
    return true;
}
// Pass A walks the terms GROUP BY GROUP: the key's top bits are the group of the term's proof, so a workgroup that only sees terms of one group scatters
// into MSM_NBIN / groups bins per window instead of MSM_NBIN -- 1 024 runs instead of 8 192 at 8 groups, each growing eight times as fast, so that a run's
// cache line is complete before the L2 has to give it up (with id-ordered ranges k_msm_scatter took 3.6 ms per 14 M-term chunk on partial-line writes).
// The live terms of group q, proofs [p0, p0 + np): every k of the three term lists over the group's contiguous slot / membership-group / proof range
// ("virtual id" v < np * MSM_TERMS_PER_PROOF); reads stay coalesced (consecutive v = consecutive g of one k).
ZK_DEV bool msm_group_term(const VWork& V, const MsmDims& D, uint32_t p0, uint32_t np, uint32_t v, uint32_t& id, uint32_t w8[8]) {
    const uint32_t na = np * VK, sec0 = V_SLOT_TERMS * na, nb = np * D.nq, sec1 = 8 * nb;
    const VTerms* L;
    uint32_t idx;
    if (v < sec0) {
        idx = (v / na) * D.g0 + p0 * VK + v % na, id = idx, L = &V.slot_terms;
    } else if (v < sec0 + sec1) {
        const uint32_t u = v - sec0;
        idx = (u / nb) * D.g1 + p0 * D.nq + u % nb, id = D.n0 + idx, L = &V.gk_terms;
    } else {
        const uint32_t u = v - sec0 - sec1;
        if (u >= 3 * np) return false;
        idx = (u / np) * D.g2 + p0 + u % np, id = D.n0 + D.n1 + idx, L = &V.misc_terms;
    }
    const Fe<ModQ, 1> sc = soa_ld<ModQ, 1>(L->sc, idx);
    if (fe_is_zero(sc)) return false;
    words_from_limbs<8>(w8, sc.l);
    return true;
}
// workgroup -> (group q, its proofs [p0, p0 + np), its virtual ids [lo, hi))
template <int C>
ZK_DEV void msm_wg_range(const MsmDims& D, uint32_t& q, uint32_t& p0, uint32_t& np, uint32_t& lo, uint32_t& hi) {
    constexpr uint32_t per_group = MSM_SORT_G / MsmShape<C>::g;   // workgroups per group
    q = blockIdx.x / per_group;
    p0 = q * D.gsz;
    const uint32_t count = D.l2;
    np = p0 >= count ? 0 : (p0 + D.gsz <= count ? D.gsz : count - p0);
    const uint32_t total = np * (V_SLOT_TERMS * VK + 8 * D.nq + 3);
    const uint32_t per = ((total + per_group - 1) / per_group + MSM_SORT_T - 1) / MSM_SORT_T * MSM_SORT_T;
    lo = (blockIdx.x % per_group) * per;
    hi = lo + per < total ? lo + per : total;
    if (lo > hi) lo = hi;
}
template <int C>
ZK_DEV uint32_t msm_digit(const uint32_t w8[8], int w) {   // bits [C w, C (w + 1)) of the 256-bit scalar (w is a constant after unrolling)
    const int bit = C * w, k = bit >> 5, sh = bit & 31;
    uint32_t d = w8[k] >> sh;
    if (sh + C > 32 && k + 1 < 8) d |= w8[k + 1] << (32 - sh);
    return d & (MsmShape<C>::nb - 1);
}
// Pass A, counting.  bin_cnt[g][w * MSM_NBIN + bin]: pairs of workgroup g's terms in bin `bin` of window w.  live_cnt[g]: its live terms.
template <int C>
__global__ void __launch_bounds__(MSM_SORT_T) k_msm_hist(VWork V, MsmDims D, uint32_t* bin_cnt, uint32_t* live_cnt) {
    typedef MsmShape<C> S;
    __shared__ uint32_t h[S::nw * MSM_NBIN];
    __shared__ uint32_t nlive;
    for (uint32_t i = threadIdx.x; i < S::nw * MSM_NBIN; i += MSM_SORT_T) h[i] = 0;
    if (threadIdx.x == 0) nlive = 0;
    __syncthreads();
    uint32_t q, p0, np, lo, hi;
    msm_wg_range<C>(D, q, p0, np, lo, hi);
    const uint32_t grp = q << S::c;
    uint32_t mine = 0;
    for (uint32_t v = lo + threadIdx.x; v < hi; v += MSM_SORT_T) {
        uint32_t w8[8], id;
        if (!msm_group_term(V, D, p0, np, v, id, w8)) continue;
        mine++;
#pragma unroll
        for (int w = 0; w < (int)S::nw; w++) {
            const uint32_t d = msm_digit<C>(w8, w);
            if (d) atomicAdd(&h[w * MSM_NBIN + ((grp | d) >> MSM_LB)], 1u);
        }
    }
    if (mine) atomicAdd(&nlive, mine);
    __syncthreads();
    uint32_t* out = bin_cnt + (size_t)blockIdx.x * (S::nw * MSM_NBIN);
    for (uint32_t i = threadIdx.x; i < S::nw * MSM_NBIN; i += MSM_SORT_T) out[i] = h[i];
    if (threadIdx.x == 0) live_cnt[blockIdx.x] = nlive;
}
// bin_off[wb][g] = pairs of (window, bin) wb that the workgroups before g hold; bin_tot[wb] = all of them.  One workgroup of MSM_SORT_G threads per wb.
__global__ void __launch_bounds__(MSM_SORT_G) k_msm_binscan1(const uint32_t* __restrict__ bin_cnt, uint32_t nwb, uint32_t* bin_off, uint32_t* bin_tot) {
    __shared__ uint32_t sh[17];
    const uint32_t wb = blockIdx.x, g = threadIdx.x;
    const uint32_t v = bin_cnt[(size_t)g * nwb + wb];
    uint32_t tot;
    const uint32_t e = block_excl_scan(v, sh, tot);
    bin_off[(size_t)wb * MSM_SORT_G + g] = e;
    if (g == 0) bin_tot[wb] = tot;
}
// bin_start[w][0 .. MSM_NBIN]: where the bins of window w start in its pair arrays (entry MSM_NBIN = the window's pairs).  One workgroup of MSM_NBIN
// threads per window; workgroup 0 also adds up the live terms (counters[0]: the bucket kernel's "oversized" threshold and the host's statistics).
__global__ void __launch_bounds__(MSM_NBIN) k_msm_binscan2(const uint32_t* __restrict__ bin_tot, uint32_t* bin_start, const uint32_t* __restrict__ live_cnt, uint32_t* counters) {
    __shared__ uint32_t sh[17];
    const uint32_t w = blockIdx.x, b = threadIdx.x;
    uint32_t tot;
    const uint32_t e = block_excl_scan(bin_tot[w * MSM_NBIN + b], sh, tot);
    bin_start[w * (MSM_NBIN + 1) + b] = e;
    if (b == 0) bin_start[w * (MSM_NBIN + 1) + MSM_NBIN] = tot;
    if (w == 0) {
        uint32_t n;
        (void)block_excl_scan(b < MSM_SORT_G ? live_cnt[b] : 0, sh, n);   // (MSM_SORT_G == MSM_NBIN threads)
        if (b == 0) counters[0] = n;
    }
}
static_assert(MSM_SORT_G == MSM_NBIN, "k_msm_binscan2 sums live_cnt with one thread per workgroup of pass A");
static_assert(MSM_SORT_G % MSM_G_MAX == 0, "pass A gives every group the same number of workgroups");
// Pass A, placing: the walk of k_msm_hist again; pair (key, id) of window w goes to position bin_start[w][bin] + bin_off[w, bin][g] + (its rank among this
// workgroup's pairs of that bin, in whatever order the LDS cursor hands out: a bucket's sum does not depend on the order of its terms).  One 8-byte store per pair.
template <int C>
__global__ void __launch_bounds__(MSM_SORT_T) k_msm_scatter(VWork V, MsmDims D, uint32_t cap, const uint32_t* __restrict__ bin_start,
                                                            const uint32_t* __restrict__ bin_off, uint2* pairs) {
    typedef MsmShape<C> S;
    __shared__ uint32_t cur[S::nw * MSM_NBIN];
    for (uint32_t i = threadIdx.x; i < S::nw * MSM_NBIN; i += MSM_SORT_T)
        cur[i] = bin_start[(i / MSM_NBIN) * (MSM_NBIN + 1) + i % MSM_NBIN] + bin_off[(size_t)i * MSM_SORT_G + blockIdx.x];
    __syncthreads();
    uint32_t q, p0, np, lo, hi;
    msm_wg_range<C>(D, q, p0, np, lo, hi);
    const uint32_t grp = q << S::c;
    for (uint32_t v = lo + threadIdx.x; v < hi; v += MSM_SORT_T) {
        uint32_t w8[8], id;
        if (!msm_group_term(V, D, p0, np, v, id, w8)) continue;
#pragma unroll
        for (int w = 0; w < (int)S::nw; w++) {
            const uint32_t d = msm_digit<C>(w8, w);
            if (d) {
                const uint32_t key = grp | d;
                const uint32_t pos = atomicAdd(&cur[w * MSM_NBIN + (key >> MSM_LB)], 1u);
                pairs[(size_t)w * cap + pos] = make_uint2(key, id);
            }
        }
    }
}
// Pass B: workgroup (bin, w) sorts its pairs by the low MSM_LB key bits.  vals[w][...]: term ids grouped by key; start / end [w * MSM_NBG + key]: the
// group's positions (digit 0 and absent keys: empty); order[w * MSM_NBG + (bin << MSM_LB) + r] (ZKATTEST_MSM_ORDER=local only): this bin's MSM_NLOW buckets,
// largest first; size_cnt[t][w, bin]: how many of them have size key t = 255 - min(size, 255) (the chunk-wide order: k_msm_sizescan, k_msm_order).
// One workgroup of 1 024 threads per CU (it asks for most of the CU's LDS): a bin's ids -- 27 k in the low windows -- are placed in LDS and leave as
// coalesced runs; 256 resident workgroups per XCD scattering single dwords over 110 KB each kept no line in the L2 until it was full (2.7 ms per chunk).
#define MSM_STAGE_MAX 36000u   // pairs of a bin that are staged in LDS (144 000 bytes); a larger bin scatters straight to memory
__global__ void __launch_bounds__(MSM_NLOW) k_msm_binsort(const uint2* __restrict__ pairs, uint32_t cap, const uint32_t* __restrict__ bin_start,
                                                          uint32_t* vals, uint32_t* start, uint32_t* end, uint32_t* order, uint32_t* size_cnt /* [256][windows * bins] or nullptr */) {
    static_assert(MSM_NLOW == 1024, "one key per thread");
    extern __shared__ uint32_t stage[];   // [MSM_STAGE_MAX]
    __shared__ uint32_t h[MSM_NLOW], hs[256], sh[17];
    const uint32_t bin = blockIdx.x, w = blockIdx.y, t = threadIdx.x;
    const uint32_t b0 = bin_start[w * (MSM_NBIN + 1) + bin], b1 = bin_start[w * (MSM_NBIN + 1) + bin + 1], n = b1 - b0;
    h[t] = 0;
    if (t < 256) hs[t] = 0;
    __syncthreads();
    const uint2* pr = pairs + (size_t)w * cap;
    for (uint32_t i = b0 + t; i < b1; i += MSM_NLOW) atomicAdd(&h[pr[i].x & (MSM_NLOW - 1)], 1u);
    __syncthreads();
    const uint32_t c = h[t], sk = 255u - (c < 255u ? c : 255u);
    atomicAdd(&hs[sk], 1u);
    uint32_t tot;
    const uint32_t e = block_excl_scan(c, sh, tot);
    const size_t wd0 = (size_t)w * MSM_NBG + ((size_t)bin << MSM_LB);
    h[t] = e;   // the key's cursor, relative to b0
    start[wd0 + t] = b0 + e, end[wd0 + t] = b0 + e + c;
    const uint32_t hv = t < 256 ? hs[t] : 0;   // buckets of this bin whose size key is t (the barriers of the scan above lie between the atomics on hs and this read)
    if (size_cnt && t < 256) size_cnt[(size_t)t * (gridDim.y * MSM_NBIN) + blockIdx.y * MSM_NBIN + bin] = hv;
    uint32_t tot2;
    const uint32_t he = block_excl_scan(hv, sh, tot2);
    if (t < 256) hs[t] = he;
    __syncthreads();
    if (!size_cnt) order[wd0 + atomicAdd(&hs[sk], 1u)] = (uint32_t)(wd0 + t);
    uint32_t* vo = vals + (size_t)w * cap + b0;
    if (n <= MSM_STAGE_MAX) {
        for (uint32_t i = b0 + t; i < b1; i += MSM_NLOW) {
            const uint2 p = pr[i];
            stage[atomicAdd(&h[p.x & (MSM_NLOW - 1)], 1u)] = p.y;
        }
        __syncthreads();
        for (uint32_t i = t; i < n; i += MSM_NLOW) vo[i] = stage[i];
    } else {
        for (uint32_t i = b0 + t; i < b1; i += MSM_NLOW) {
            const uint2 p = pr[i];
            vo[atomicAdd(&h[p.x & (MSM_NLOW - 1)], 1u)] = p.y;
        }
    }
}
// The same order over ALL buckets of the chunk (ZKATTEST_MSM_ORDER=global): size key t = 255 - min(size, 255) first, then (window, bin), then whatever order
// the cursors hand out.  k_msm_sizescan: workgroup t turns size_cnt[t][*] into exclusive offsets in place and leaves the key's total in size_tot[t];
// k_msm_order: workgroup (window, bin) places its MSM_NLOW buckets.
__global__ void __launch_bounds__(1024) k_msm_sizescan(uint32_t* size_cnt, uint32_t nwb, uint32_t* size_tot) {
    __shared__ uint32_t sh[17];
    uint32_t* row = size_cnt + (size_t)blockIdx.x * nwb;
    const uint32_t per = (nwb + 1023) / 1024, lo = threadIdx.x * per, hi = lo + per < nwb ? lo + per : nwb;
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += row[i];
    uint32_t tot;
    uint32_t e = block_excl_scan(sum, sh, tot);
    for (uint32_t i = lo; i < hi; i++) {
        const uint32_t v = row[i];
        row[i] = e, e += v;
    }
    if (threadIdx.x == 0) size_tot[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(256) k_msm_order(const uint32_t* __restrict__ start, const uint32_t* __restrict__ end, const uint32_t* __restrict__ size_off,
                                                   const uint32_t* __restrict__ size_tot, uint32_t* order) {
    constexpr uint32_t PER = MSM_NLOW / 256;
    __shared__ uint32_t cur[256], sh[17];
    const uint32_t bin = blockIdx.x, w = blockIdx.y, t = threadIdx.x, nwb = gridDim.y * MSM_NBIN;
    uint32_t tot;
    const uint32_t base = block_excl_scan(size_tot[t], sh, tot);
    cur[t] = base + size_off[(size_t)t * nwb + w * MSM_NBIN + bin];
    __syncthreads();
    const size_t wd0 = (size_t)w * MSM_NBG + ((size_t)bin << MSM_LB);
#pragma unroll
    for (uint32_t j = 0; j < PER; j++) {
        const size_t wd = wd0 + PER * t + j;
        const uint32_t c = end[wd] - start[wd];
        order[atomicAdd(&cur[255u - (c < 255u ? c : 255u)], 1u)] = (uint32_t)wd;
    }
}
// ZK_MSM_CHECK=1 (tests): is the grouping exact?  Every position of a window's id list must hold a live term whose key owns that position, no term
// twice (bitmap), as many positions as pairs were counted.  err[0]: violations, err[1 + w]: positions seen per window.
template <int C>
__global__ void __launch_bounds__(256) k_msm_check(VWork V, MsmDims D, uint32_t cap, const uint32_t* __restrict__ bin_start, const uint32_t* __restrict__ vals,
                                                   const uint32_t* __restrict__ start, const uint32_t* __restrict__ end, uint32_t* seen, uint32_t* err) {
    typedef MsmShape<C> S;
    const uint32_t w = blockIdx.y, pos = gtid();
    if (pos >= bin_start[w * (MSM_NBIN + 1) + MSM_NBIN]) return;
    atomicAdd(&err[1 + w], 1u);
    const uint32_t id = vals[(size_t)w * cap + pos];
    uint32_t w8[8], proof;
    if (!msm_term_words(V, D, id, w8, proof)) {
        atomicAdd(&err[0], 1u);
        return;
    }
    uint32_t d = 0;
#pragma unroll
    for (int k = 0; k < (int)S::nw; k++)
        if (k == (int)w) d = msm_digit<C>(w8, k);
    const uint32_t key = ((proof / D.gsz) << S::c) | d;
    const size_t wd = (size_t)w * MSM_NBG + key;
    if (!d || pos < start[wd] || pos >= end[wd]) atomicAdd(&err[0], 1u);
    const size_t bit = (size_t)w * cap + id;
    if (atomicOr(&seen[bit >> 5], 1u << (bit & 31)) & (1u << (bit & 31))) atomicAdd(&err[0], 1u);
}
template <int C>
__global__ void __launch_bounds__(256) k_msm_check_count(VWork V, MsmDims D, uint32_t* err) {   // err[64 + w]: pairs window w must hold
    typedef MsmShape<C> S;
    uint32_t w8[8], proof;
    if (!msm_term_words(V, D, gtid(), w8, proof)) return;
#pragma unroll
    for (int w = 0; w < (int)S::nw; w++)
        if (msm_digit<C>(w8, w)) atomicAdd(&err[64 + w], 1u);
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
template <int C>
__global__ void __launch_bounds__(256) k_msm_bucket(const uint32_t* __restrict__ aos, const uint32_t* __restrict__ vals, uint32_t cap,
                                                    const uint32_t* __restrict__ start, const uint32_t* __restrict__ end, const uint32_t* __restrict__ order,
                                                    uint32_t* buckets, uint32_t* big_cnt, uint32_t* big_list, const uint32_t* __restrict__ nlive) {
    const uint32_t big = 8 * ((*nlive + MSM_NBG - 1) / MSM_NBG) + 64;   // oversized: 8 x a window's average + 64
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
// sum_d d * B_d per (window, group) in FOUR levels of sixteen (until round 5: 64 / 32 / 32, 361 dependent point operations in a row -- 3.3 ms per chunk during
// which the GPU did next to nothing else; now 4 x 31 + 15):
//   level 1: a thread takes 16 buckets:  F = sum_j j B_j (running sums), G = sum_j B_j
//   level l: a thread takes 16 (the last level: what is left) entries of level l - 1.  Role 0 (blockIdx.y): F' = sum_j j G_j, G' = sum_j G_j; role k >= 1: the plain
//            sum of the k-th carried array -- the F of level l - 1, and the plain sums carried so far.  All roles of a level run in ONE launch, so a level is 31
//            operations long whatever it carries.
//   last:    sum_d d B_d = P_0 + 16 (P_1 + 16 (P_2 + 16 F_4)),  P_0 = sum of all F_1, P_1 = sum of all F_2, P_2 = sum of all F_3;  times 2^(C w)
#define MSM_RED_R 16u
struct MsmRedLevel {
    const uint32_t* Gin;
    const uint32_t* Pin[3];   // carried arrays (Pin[0] = F of the level below), nin entries per (window, group) each
    uint32_t *Fout, *Gout, *Pout[3];
    uint32_t nin, r, nwg;     // entries per (window, group) on the way in; entries a thread takes; (window, group) pairs
};
template <int C>
__global__ void __launch_bounds__(256) k_msm_red1(const uint32_t* __restrict__ buckets, uint32_t* F1, uint32_t* G1) {
    typedef MsmShape<C> S;
    constexpr uint32_t n1 = S::nb / MSM_RED_R;
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x, w = blockIdx.y;   // w = window * groups + group
    if (r >= n1) return;
    const uint32_t* b = buckets + ((size_t)w * S::nb + MSM_RED_R * r) * 36;
    TomPt run = tom_identity(), acc = tom_identity();
#pragma unroll 1
    for (int j = MSM_RED_R - 1; j >= 1; j--) {
        run = tom_add(run, msm_ldp(b + 36 * j));
        acc = tom_add(acc, run);
    }
    run = tom_add(run, msm_ldp(b));
    msm_st(F1 + ((size_t)w * n1 + r) * 36, acc);
    msm_st(G1 + ((size_t)w * n1 + r) * 36, run);
}
__global__ void __launch_bounds__(64) k_msm_redk(MsmRedLevel L) {
    const uint32_t nout = L.nin / L.r, t = gtid(), role = blockIdx.y;
    if (t >= L.nwg * nout) return;
    const size_t base = ((size_t)(t / nout) * L.nin + (size_t)(t % nout) * L.r) * 36;
    if (role == 0) {
        const uint32_t* g = L.Gin + base;
        TomPt run = tom_identity(), acc = tom_identity();
#pragma unroll 1
        for (int j = (int)L.r - 1; j >= 1; j--) {
            run = tom_add(run, msm_ldp(g + 36 * j));
            acc = tom_add(acc, run);
        }
        run = tom_add(run, msm_ldp(g));
        msm_st(L.Fout + (size_t)t * 36, acc), msm_st(L.Gout + (size_t)t * 36, run);
    } else {
        const uint32_t* p = L.Pin[role - 1] + base;
        TomPt h = msm_ldp(p);
#pragma unroll 1
        for (uint32_t j = 1; j < L.r; j++) h = tom_add(h, msm_ldp(p + 36 * j));
        msm_st(L.Pout[role - 1] + (size_t)t * 36, h);
    }
}
// one thread per (window, group): F4 and the three carried sums -> sum_d d B_d, times 2^(C w)
template <int C>
__global__ void __launch_bounds__(64) k_msm_red_last(const uint32_t* __restrict__ F4, const uint32_t* __restrict__ P2, const uint32_t* __restrict__ P1, const uint32_t* __restrict__ P0,
                                                    uint32_t* Tw) {
    typedef MsmShape<C> S;
    const uint32_t w = gtid();   // window * groups + group
    if (w >= S::nwg) return;
    TomPt t = msm_ldp(F4 + (size_t)w * 36);
    const uint32_t* carried[3] = {P2, P1, P0};
#pragma unroll 1
    for (int k = 0; k < 3; k++) {
        for (int i = 0; i < 4; i++) t = tom_dbl(t);   // x 16
        t = tom_add(t, msm_ldp(carried[k] + (size_t)w * 36));
    }
#pragma unroll 1
    for (uint32_t i = 0; i < S::c * (w / S::g); i++) t = tom_dbl(t);
    msm_st(Tw + (size_t)w * 36, t);
}
// The same on a cooperating wave per (window, group) (coop.h): up to 16 w doublings in a row are what the last chunk of a call waits for -- 0.57 us per doubling
// against 3.85 us in one lane (profiles/r06_coop_microbench.txt).
template <int C>
__global__ void __launch_bounds__(64) k_msm_red_last_co(const uint32_t* __restrict__ F4, const uint32_t* __restrict__ P2, const uint32_t* __restrict__ P1, const uint32_t* __restrict__ P0,
                                                       uint32_t* Tw) {
    typedef MsmShape<C> S;
    const uint32_t w = blockIdx.x;   // window * groups + group
    const CoU32 mj = co_limbs(ModT::mod);
    CoTom t;
    t.v = co_load_aos<ModT, 2, 4>(F4 + (size_t)w * 36);
    const uint32_t* carried[3] = {P2, P1, P0};
#pragma unroll 1
    for (int k = 0; k < 3; k++) {
        CoTom c;
        c.v = co_load_aos<ModT, 2, 4>(carried[k] + (size_t)w * 36);
#pragma unroll 1
        for (int i = 0; i < 4; i++) t = co_tom_dbl(t, mj);   // x 16
        t = co_tom_add(t, c, mj);
    }
#pragma unroll 1
    for (uint32_t i = 0; i < S::c * (w / S::g); i++) t = co_tom_dbl(t, mj);
    co_store_aos<ModT, 2, 4>(Tw + (size_t)w * 36, t.v);
}
// entries of MsmBuf::red per (window, group): level 1 (F, G), level 2 (F, G, P), level 3 (F, G, 2 P), level 4 (F, G, 3 P)
template <int C>
struct MsmRedPlan {
    static constexpr uint32_t n1 = MsmShape<C>::nb / MSM_RED_R, n2 = n1 / MSM_RED_R, n3 = n2 / MSM_RED_R;   // 4096 / 256 / 16 at C = 16, 512 / 32 / 2 at C = 13
    static_assert(n3 >= 1 && n3 <= MSM_RED_R, "four levels");
    static constexpr size_t per_wg = 2 * (size_t)n1 + 3 * n2 + 4 * n3 + 5;
};
size_t msm_red_words(uint32_t groups) {
    return groups == 64 ? MsmRedPlan<13>::per_wg * MsmShape<13>::nwg * 36 : MsmRedPlan<16>::per_wg * MsmShape<16>::nwg * 36;
}
template <int C>
static void launch_msm_reduce(hipStream_t s, const MsmBuf& M) {
    typedef MsmShape<C> S;
    typedef MsmRedPlan<C> R;
    const size_t W = (size_t)S::nwg * 36;
    uint32_t* p = M.red;
    auto take = [&](size_t n) { uint32_t* q = p; p += n * W; return q; };
    uint32_t *F1 = take(R::n1), *G1 = take(R::n1);
    uint32_t *F2 = take(R::n2), *G2 = take(R::n2), *P2a = take(R::n2);
    uint32_t *F3 = take(R::n3), *G3 = take(R::n3), *P3a = take(R::n3), *P3b = take(R::n3);
    uint32_t *F4 = take(1), *G4 = take(1), *P4a = take(1), *P4b = take(1), *P4c = take(1);
    constexpr uint32_t bt = R::n1 < 256 ? R::n1 : 256;
    hipLaunchKernelGGL(k_msm_red1<C>, dim3(R::n1 / bt, S::nwg), dim3(bt), 0, s, M.buckets, F1, G1);
    MsmRedLevel L2{G1, {F1, nullptr, nullptr}, F2, G2, {P2a, nullptr, nullptr}, R::n1, MSM_RED_R, S::nwg};
    hipLaunchKernelGGL(k_msm_redk, dim3((S::nwg * R::n2 + 63) / 64, 2), dim3(64), 0, s, L2);
    MsmRedLevel L3{G2, {F2, P2a, nullptr}, F3, G3, {P3a, P3b, nullptr}, R::n2, MSM_RED_R, S::nwg};
    hipLaunchKernelGGL(k_msm_redk, dim3((S::nwg * R::n3 + 63) / 64, 3), dim3(64), 0, s, L3);
    MsmRedLevel L4{G3, {F3, P3a, P3b}, F4, G4, {P4a, P4b, P4c}, R::n3, R::n3, S::nwg};
    hipLaunchKernelGGL(k_msm_redk, dim3((S::nwg + 63) / 64, 4), dim3(64), 0, s, L4);
    if (zk_one_lane_chains()) hipLaunchKernelGGL(k_msm_red_last<C>, dim3((S::nwg + 63) / 64), dim3(64), 0, s, F4, P4a, P4b, P4c, M.Tw);
    else hipLaunchKernelGGL(k_msm_red_last_co<C>, dim3(S::nwg), dim3(64), 0, s, F4, P4a, P4b, P4c, M.Tw);
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

// bytes of the grouping passes' scratch (MsmBuf::sort_tmp): bin_cnt and bin_off [MSM_SORT_G][windows * bins], bin_tot, bin_start, live_cnt
size_t msm_workspace_bytes(uint32_t) {
    const size_t nwb = (size_t)MSM_NW_MAX * MSM_NBIN;
    return 4 * (2 * nwb * MSM_SORT_G + nwb + (size_t)MSM_NW_MAX * (MSM_NBIN + 1) + MSM_SORT_G + 256 * nwb + 256) + 1024;   // + size_cnt, size_tot
}
// Enqueues the pass on s and returns; once s has drained to this point M.host holds the live-term count and the groups' verdicts (msm_read_flags): 1 = the Tom
// total of that group of proofs is the identity.  No host round trip in here: the caller enqueues the passes of several chunks before it waits for the first.
template <int C>
static hipError_t run_msm_t(hipStream_t s, const DevParams& P, const Workspace& W, const VWork& V, uint32_t count, uint32_t nq, const MsmBuf& M, uint32_t* gsz_out,
                            hipEvent_t ev0, hipEvent_t ev1, hipEvent_t ev2, hipEvent_t ev3) {
    typedef MsmShape<C> S;
    MsmDims D;
    D.g0 = V.C * VK, D.g1 = V.C * nq, D.g2 = V.C;
    D.n0 = D.g0 * V_SLOT_TERMS, D.n1 = D.g1 * 8, D.n2 = D.g2 * 3;
    D.l0 = count * VK, D.l1 = count * nq, D.l2 = count;
    D.nq = nq, D.gsz = (count + S::g - 1) / S::g;
    *gsz_out = D.gsz;
    const uint32_t total = D.n0 + D.n1 + D.n2;
    // scratch of the grouping passes
    const uint32_t nwb = S::nw * MSM_NBIN;
    uint32_t* bin_cnt = (uint32_t*)M.sort_tmp;
    uint32_t* bin_off = bin_cnt + (size_t)MSM_SORT_G * nwb;
    uint32_t* bin_tot = bin_off + (size_t)MSM_SORT_G * nwb;
    uint32_t* bin_start = bin_tot + nwb;
    uint32_t* live_cnt = bin_start + (size_t)S::nw * (MSM_NBIN + 1);
    hipMemsetAsync(M.counters, 0, sizeof(uint32_t) * 64, s);
    if (ev2) hipEventRecord(ev2, s);   // the grouping of the keys alone (bench.py: verify.roofline.non_arithmetic)
    hipLaunchKernelGGL(k_msm_hist<C>, dim3(MSM_SORT_G), dim3(MSM_SORT_T), 0, s, V, D, bin_cnt, live_cnt);
    hipLaunchKernelGGL(k_msm_binscan1, dim3(nwb), dim3(MSM_SORT_G), 0, s, bin_cnt, nwb, bin_off, bin_tot);
    hipLaunchKernelGGL(k_msm_binscan2, dim3(S::nw), dim3(MSM_NBIN), 0, s, bin_tot, bin_start, live_cnt, M.counters);
    hipLaunchKernelGGL(k_msm_scatter<C>, dim3(MSM_SORT_G), dim3(MSM_SORT_T), 0, s, V, D, M.cap, bin_start, bin_off, M.pairs);
    // order of the buckets for k_msm_bucket's lanes: by size over the whole chunk (within each bin only, two launches fewer, cost 8 ms of bucket sums per
    // step: profiles/r05_ab_variants.txt (1))
    uint32_t* size_cnt = (uint32_t*)(live_cnt + MSM_SORT_G);   // [256][nwb], then size_tot[256]
    uint32_t* size_tot = size_cnt + (size_t)256 * nwb;
    (void)hipFuncSetAttribute((const void*)k_msm_binsort, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(MSM_STAGE_MAX * 4));   // (per device: a pool calls this on each)
    hipLaunchKernelGGL(k_msm_binsort, dim3(MSM_NBIN, S::nw), dim3(MSM_NLOW), MSM_STAGE_MAX * 4, s, M.pairs, M.cap, bin_start, M.vals_out, M.start, M.end, M.ord_id,
                       size_cnt);
    hipLaunchKernelGGL(k_msm_sizescan, dim3(256), dim3(1024), 0, s, size_cnt, nwb, size_tot);
    hipLaunchKernelGGL(k_msm_order, dim3(MSM_NBIN, S::nw), dim3(256), 0, s, M.start, M.end, size_cnt, size_tot, M.ord_id);
    if (ev3) hipEventRecord(ev3, s);
    if (getenv("ZK_MSM_CHECK")) {   // tests: the grouping against first principles (k_msm_check)
        uint32_t *seen = nullptr, *err = nullptr;
        const size_t seen_words = ((size_t)S::nw * M.cap + 31) / 32;
        if (hipMalloc(&seen, seen_words * 4) != hipSuccess || hipMalloc(&err, 512) != hipSuccess) return hipErrorOutOfMemory;
        hipMemsetAsync(seen, 0, seen_words * 4, s), hipMemsetAsync(err, 0, 512, s);
        hipLaunchKernelGGL(k_msm_check<C>, dim3((M.cap + 255) / 256, S::nw), dim3(256), 0, s, V, D, M.cap, bin_start, M.vals_out, M.start, M.end, seen, err);
        hipLaunchKernelGGL(k_msm_check_count<C>, dim3((total + 255) / 256), dim3(256), 0, s, V, D, err);
        uint32_t h[128];
        hipMemcpyAsync(h, err, 512, hipMemcpyDeviceToHost, s);
        hipError_t ec = hipStreamSynchronize(s);
        hipFree(seen), hipFree(err);
        if (ec != hipSuccess) return ec;
        bool bad = h[0] != 0;
        for (uint32_t w = 0; w < S::nw; w++) bad = bad || h[1 + w] != h[64 + w];
        if (bad) {
            fprintf(stderr, "ZK_MSM_CHECK: grouping wrong: %u violations; pairs per window (listed / expected):", h[0]);
            for (uint32_t w = 0; w < S::nw; w++) fprintf(stderr, " %u/%u", h[1 + w], h[64 + w]);
            fprintf(stderr, "\n");
            return hipErrorAssert;
        }
    }
    // the groups' fixed-base parts need nothing from the buckets: before the long kernel, not in the chain of small ones behind it
    hipLaunchKernelGGL(k_msm_coef, dim3(S::g), dim3(256), 0, s, W, count, D.gsz, M.one);
    launch_tom_commit(s, P, M.one, S::g, 1, 1);
    if (ev0) hipEventRecord(ev0, s);   // the bucket sums alone (bench.py: roofline.others)
    hipLaunchKernelGGL(k_msm_bucket<C>, dim3(S::nw * MSM_NBG / 256), dim3(256), 0, s, M.aos, M.vals_out, M.cap, M.start, M.end, M.ord_id, M.buckets, M.counters + 32,
                       M.big_list, M.counters);
    if (ev1) hipEventRecord(ev1, s);
    hipLaunchKernelGGL(k_msm_bucket_big, dim3(MSM_NSLICE, 32), dim3(256), 0, s, M.aos, M.vals_out, M.cap, M.start, M.end, M.counters + 32, M.big_list, M.big_part);
    hipLaunchKernelGGL(k_msm_bucket_big2, dim3(64), dim3(256), 0, s, M.counters + 32, M.big_list, M.big_part, M.buckets);
    launch_msm_reduce<C>(s, M);
    hipLaunchKernelGGL(k_msm_final<C>, dim3(1), dim3(64), 0, s, M.Tw, M.one, M.flag);
    launch_words_to_host(s, M.host, M.counters, 1);   // live terms of the pass (statistics: zk_test_counter 2)
    launch_words_to_host(s, M.host + 8, M.flag, S::g);
    return hipSuccess;
}
hipError_t run_msm(hipStream_t s, const DevParams& P, const Workspace& W, const VWork& V, uint32_t count, uint32_t nq, const MsmBuf& M, uint32_t groups,
                   uint32_t* gsz_out, hipEvent_t ev0, hipEvent_t ev1, hipEvent_t ev2, hipEvent_t ev3) {
    return groups == 64 ? run_msm_t<13>(s, P, W, V, count, nq, M, gsz_out, ev0, ev1, ev2, ev3) : run_msm_t<16>(s, P, W, V, count, nq, M, gsz_out, ev0, ev1, ev2, ev3);
}
void msm_read_flags(const MsmBuf& M, uint32_t groups, uint32_t* host_flags) {
    for (uint32_t g = 0; g < groups; g++) host_flags[g] = M.host[8 + g];
}
