// Cross-proof batched check of the verifier's P-256 relation (the P-256 half; k_msm.hip is the Tom-256 half).
//
// verifyExp's P-256 side (src/exp/exp.ts:270-276, 305-317; multiN.evaluate(), exp.ts:348) is, per proof, ONE sum
//     SR * R + SH * h_NIST + SL * Clambda + sum_j rho_j * (-A_j)  ==  identity            (k_v_slot_terms / k_v_proof_terms fold it with 128-bit rho_j)
// and the rho_j of different proofs are independent, so the sum over a GROUP of proofs is the identity iff (up to 2^-128) every proof's is.  The 21 terms per
// proof with short scalars -- 20 x (rho_j < 2^128, -A_j) and (SL < 2^133, Clambda) -- go through the bucket method: 10 (13) mixed additions per term instead of
// the 33 windows x (4 doublings + 1 addition) / 5 of k_v_p256_straus.  SR * R keeps its per-proof window table (k_v_exp_points needs that table anyway: 65
// additions, no doubling) and is summed over the group by a tree; the h_NIST parts become ONE comb walk per group on the sum of the SH.
// If a group's total is not the identity the caller runs the per-proof sums (k_v_p256_tables / _straus / _total) to find out which proofs are bad.
//
// Shape: C-bit digits over the low nw * C >= 134 scalar bits; C = 13 with 8 groups per chunk (8 192 buckets per (window, group), about ten terms each at
// 32 768 proofs), C = 10 with 64 groups.  The top window of the rho_j is 11 (8) bits wide -- buckets there are twice (four times) as full -- and the windows
// above bit 128 see only SL >> 128, a handful of values: those buckets are "big" and get a workgroup each.
//
// Part 1 (pmsm_prepare; closes stage 1 of the chunk on its own stream -- latency-bound, 0.8 ms per 32 768 proofs):
//   k_pm_pack     term -> 64-byte affine entry + its digits (uint16, window-major)
//   k_pm_group    workgroup (window, group): counting sort of the group's terms by digit in LDS -> id lists, bucket bounds, buckets ordered by size
//   k_pm_rpart    SR * R per proof through its table, summed per 64 proofs; the SH of the same proofs summed mod n
// Part 2 (pmsm_sums; an auxiliary stream forked at the start of stage 2: arithmetic beside the Tom-256 pass's grouping kernels, which are not; measured
// placements in profiles/r05_p256_pass_ab.txt):
//   k_pm_bucket   thread per bucket, in size order: complete mixed additions (RCB 2016 algorithm 5); k_pm_big: a workgroup per oversized bucket
//   k_pm_reduce   workgroup (window, group): sum_d d * B_d by per-thread running sums and a tree of (F, G) segments
//   k_pm_final    per group: windows (Horner, C doublings each) + R parts + (sum SH) * h_NIST == identity ?
#include "rtab.h"
#include "coop_dev.h"
#include "ktab.h"

#define PM_TERMS (VK + 1)
#define PM_SCALAR_BITS 134u   // SL is the sum of at most 20 values below 2^128; k_pm_pack checks the bound on every scalar
#define PM_PT_WORDS 28
#define PM_BIG 64u            // terms above which a bucket is summed by a workgroup
#define PM_BIG_MAX 2048u
template <int C>
struct PmShape {
    static constexpr uint32_t c = C, nb = 1u << C, nw = (PM_SCALAR_BITS + C - 1) / C;
    static_assert(nw <= PM_NW_MAX, "windows");
};

ZK_DEV void pm_st(uint32_t* p, const P256Pt& a) {
    uint4* q = (uint4*)p;
    uint32_t w[28];
#pragma unroll
    for (int l = 0; l < 9; l++) w[l] = a.x.l[l], w[9 + l] = a.y.l[l], w[18 + l] = a.z.l[l];
    w[27] = 0;
#pragma unroll
    for (int i = 0; i < 7; i++) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
// points in LDS: limb-major ([27][n]), one column per thread
ZK_DEV void pm_sh_st(uint32_t* sh, uint32_t n, uint32_t i, const P256Pt& a) {
#pragma unroll
    for (int l = 0; l < 9; l++) sh[l * n + i] = a.x.l[l], sh[(9 + l) * n + i] = a.y.l[l], sh[(18 + l) * n + i] = a.z.l[l];
}
ZK_DEV P256Pt pm_sh_ld(const uint32_t* sh, uint32_t n, uint32_t i) {
    P256Pt a;
#pragma unroll
    for (int l = 0; l < 9; l++) a.x.l[l] = sh[l * n + i], a.y.l[l] = sh[(9 + l) * n + i], a.z.l[l] = sh[(18 + l) * n + i];
    return a;
}
// sum over the first n threads of the workgroup (n a power of two, <= blockDim.x), valid in thread 0.  sh: 27 * n / 2 words.
ZK_DEV P256Pt pm_block_sum(P256Pt acc, uint32_t n, uint32_t* sh) {
    const uint32_t t = threadIdx.x;
#pragma unroll 1
    for (uint32_t o = n / 2; o >= 1; o >>= 1) {
        __syncthreads();
        if (t >= o && t < 2 * o) pm_sh_st(sh, n / 2, t - o, acc);
        __syncthreads();
        if (t < o) acc = p256_add(acc, pm_sh_ld(sh, n / 2, t));
    }
    __syncthreads();
    return acc;
}
ZK_DEV bool pm_good(const VWork& V, uint32_t p) { return V.st[p] == ZK_OK && !(V.okflags[p] & 8); }

// term t < count * VK: A term t (= proof * VK + j), id = t; then Clambda of proof t - count * VK, id = C * VK + proof
template <int C>
__global__ void __launch_bounds__(256) k_pm_pack(VWork V, uint32_t count, PMsmBuf M) {
    typedef PmShape<C> S;
    const uint32_t t = gtid();
    if (t >= count * PM_TERMS) return;
    const bool is_cl = t >= count * VK;
    const uint32_t idx = is_cl ? t - count * VK : t, id = is_cl ? V.C * VK + idx : idx;
    uint32_t kw[8];
    words_from_limbs<8>(kw, soa_ld<ModN, 1>(is_cl ? V.pSL : V.pa_sc, idx).l);
    // a scalar beyond the windows would be cut off silently: the group then fails and its proofs get the per-proof sums
    static_assert(S::nw * C > 128 && S::nw * C < 160, "the bound below looks at word 4");
    const bool over = (kw[4] >> (S::nw * C - 128)) != 0 || (kw[5] | kw[6] | kw[7]) != 0;
    if (over) atomicOr(M.counters + 1, 1u);
    uint16_t* dg = M.dig + id;
#pragma unroll
    for (int w = 0; w < (int)S::nw; w++) {
        const int bit = C * w, k = bit >> 5, sh = bit & 31;
        uint32_t d = kw[k] >> sh;
        if (sh + C > 32 && k + 1 < 8) d |= kw[k + 1] << (32 - sh);
        dg[(size_t)w * M.ncap] = (uint16_t)(d & (S::nb - 1));
    }
    const Fq2 x = soa_ld<ModQ, 2>(is_cl ? V.clx : V.pa_x, idx), y = soa_ld<ModQ, 2>(is_cl ? V.cly : V.pa_y, idx);
    st_ktab(M.aos + (size_t)id * 16, fe_canon(x), fe_canon(y));
}
// id of the v-th term of the group whose proofs are [p0, p0 + np)
ZK_DEV uint32_t pm_group_id(const VWork& V, uint32_t p0, uint32_t np, uint32_t v) { return v < np * VK ? p0 * VK + v : V.C * VK + p0 + (v - np * VK); }

// Workgroup (w, g).  start / end [(w * G + g) * nb + d]: the bucket's segment of vals + (w * G + g) * gcap; order[(w * G + g) * nb + r]: its buckets, largest first.
template <int C>
__global__ void __launch_bounds__(1024) k_pm_group(VWork V, uint32_t count, uint32_t gsz, PMsmBuf M) {
    typedef PmShape<C> S;
    constexpr uint32_t PER = S::nb / 1024 ? S::nb / 1024 : 1;
    static_assert(S::nb >= 1024, "a bucket per thread at least");
    __shared__ uint32_t h[S::nb], hs[256], sh[17];
    const uint32_t w = blockIdx.x, g = blockIdx.y, t = threadIdx.x, G = gridDim.y;
    const uint32_t p0 = g * gsz, np = p0 >= count ? 0 : (p0 + gsz <= count ? gsz : count - p0), nt = np * PM_TERMS;
    for (uint32_t i = t; i < S::nb; i += 1024) h[i] = 0;
    if (t < 256) hs[t] = 0;
    __syncthreads();
    const uint16_t* dg = M.dig + (size_t)w * M.ncap;
    for (uint32_t v = t; v < nt; v += 1024) {
        const uint32_t d = dg[pm_group_id(V, p0, np, v)];
        if (d) atomicAdd(&h[d], 1u);
    }
    __syncthreads();
    uint32_t cnt[PER], sum = 0;
#pragma unroll
    for (uint32_t j = 0; j < PER; j++) cnt[j] = h[t * PER + j], sum += cnt[j];
    uint32_t tot;
    uint32_t e = block_excl_scan(sum, sh, tot);
    const size_t base = ((size_t)w * G + g) * S::nb;
#pragma unroll
    for (uint32_t j = 0; j < PER; j++) {
        const uint32_t b = t * PER + j;
        M.start[base + b] = e, M.end[base + b] = e + cnt[j];
        h[b] = e;   // the bucket's cursor
        e += cnt[j];
        atomicAdd(&hs[255u - (cnt[j] < 255u ? cnt[j] : 255u)], 1u);
    }
    __syncthreads();
    uint32_t tot2;
    const uint32_t hv = t < 256 ? hs[t] : 0;
    const uint32_t he = block_excl_scan(hv, sh, tot2);
    if (t < 256) hs[t] = he;
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < PER; j++) M.order[base + atomicAdd(&hs[255u - (cnt[j] < 255u ? cnt[j] : 255u)], 1u)] = t * PER + j;
    uint32_t* vo = M.vals + ((size_t)w * G + g) * M.gcap;
    for (uint32_t v = t; v < nt; v += 1024) {
        const uint32_t id = pm_group_id(V, p0, np, v), d = dg[id];
        if (d) vo[atomicAdd(&h[d], 1u)] = id;
    }
}
// thread (w * G + g, r): the r-th largest bucket of (w, g)
template <int C>
__global__ void __launch_bounds__(256) k_pm_bucket(PMsmBuf M) {
    typedef PmShape<C> S;
    const uint32_t tid = gtid(), wg = tid / S::nb;
    const uint32_t b = M.order[tid];
    const size_t bi = (size_t)wg * S::nb + b;
    uint32_t s = M.start[bi], e = M.end[bi];
    if (e - s > PM_BIG) {
        const uint32_t pos = atomicAdd(M.counters + 2, 1u);
        if (pos < PM_BIG_MAX) {
            M.big_list[pos] = (uint32_t)bi;
            return;   // k_pm_big writes the bucket
        }
    }
    P256Pt acc = p256_identity();
    if (e > s) {
        const uint32_t* v = M.vals + (size_t)wg * M.gcap;
        acc = p256_from_affine(ld_ktab(M.aos + (size_t)v[s] * 16));
#pragma unroll 1
        for (uint32_t i = s + 1; i < e; i++) acc = p256_add_mixed(acc, ld_ktab(M.aos + (size_t)v[i] * 16));
    }
    pm_st(M.buckets + bi * PM_PT_WORDS, acc);
}
template <int C>
__global__ void __launch_bounds__(256) k_pm_big(PMsmBuf M) {
    typedef PmShape<C> S;
    __shared__ uint32_t sh[27 * 128];
    const uint32_t n = M.counters[2] < PM_BIG_MAX ? M.counters[2] : PM_BIG_MAX, t = threadIdx.x;
    for (uint32_t k = blockIdx.x; k < n; k += gridDim.x) {
        const uint32_t bi = M.big_list[k], wg = bi / S::nb;
        const uint32_t s = M.start[bi], e = M.end[bi];
        const uint32_t* v = M.vals + (size_t)wg * M.gcap;
        P256Pt acc = p256_identity();
#pragma unroll 1
        for (uint32_t i = s + t; i < e; i += 256) acc = p256_add_mixed(acc, ld_ktab(M.aos + (size_t)v[i] * 16));
        acc = pm_block_sum(acc, 256, sh);
        if (t == 0) pm_st(M.buckets + (size_t)bi * PM_PT_WORDS, acc);
    }
}
// Workgroup (w * G + g), 256 threads: thread t owns buckets [t * PER, (t + 1) * PER): F = sum_j j * B_j, G = sum_j B_j (running sums), then a tree over
// the threads: two neighbouring segments L, R of m buckets each join as F = F_L + F_R + m * G_R (log2 m doublings), G = G_L + G_R.
template <int C>
__global__ void __launch_bounds__(256) k_pm_reduce(PMsmBuf M) {
    typedef PmShape<C> S;
    constexpr uint32_t PER = S::nb / 256;
    __shared__ uint32_t sh[27 * 128];   // (F and G take turns: 13.8 KB fit on a CU beside a workgroup of the Tom-256 pass's k_msm_binsort, 27.6 would not)
    const uint32_t wg = blockIdx.x, t = threadIdx.x;
    const uint32_t* b = M.buckets + ((size_t)wg * S::nb + (size_t)t * PER) * PM_PT_WORDS;
    P256Pt run = p256_identity(), acc = p256_identity();
#pragma unroll 1
    for (int j = (int)PER - 1; j >= 1; j--) {
        run = p256_add(run, ld_rtab(b + PM_PT_WORDS * j));
        acc = p256_add(acc, run);
    }
    run = p256_add(run, ld_rtab(b));
    uint32_t logm = 0;
    while ((1u << logm) < PER) logm++;
#pragma unroll 1
    for (uint32_t o = 1; o < 256; o <<= 1, logm++) {   // segments of o threads: the right one of each pair hands (F, G) to the left one
        const bool right = (t & (2 * o - 1)) == o, left = (t & (2 * o - 1)) == 0;
        __syncthreads();
        if (right) pm_sh_st(sh, 128, t / (2 * o), run);
        __syncthreads();
        P256Pt gr = p256_identity();
        if (left) gr = pm_sh_ld(sh, 128, t / (2 * o));
        __syncthreads();
        if (right) pm_sh_st(sh, 128, t / (2 * o), acc);
        __syncthreads();
        if (left) {
            run = p256_add(run, gr);
#pragma unroll 1
            for (uint32_t k = 0; k < logm; k++) gr = p256_dbl(gr);
            acc = p256_add(p256_add(acc, pm_sh_ld(sh, 128, t / (2 * o))), gr);
        }
    }
    if (t == 0) pm_st(M.Tw + (size_t)wg * PM_PT_WORDS, acc);
}
// Workgroup (b, g): proofs g * gsz + 64 b + t.  rpart[g * PB + b]: sum of SR * R; shpart[(g * PB + b) * 9 ..]: sum of SH mod n.
__global__ void __launch_bounds__(64) k_pm_rpart(Workspace W, VWork V, uint32_t count, uint32_t gsz, PMsmBuf M) {
    __shared__ uint32_t sh[27 * 32], shs[9][64];
    const uint32_t g = blockIdx.y, t = threadIdx.x, PB = gridDim.x;
    const uint32_t k = blockIdx.x * 64 + t, p = g * gsz + k;
    P256Pt acc = p256_identity();
    Fe<ModN, 1> shv = fe_zero<ModN>();
    if (k < gsz && p < count && pm_good(V, p)) {
        uint32_t kw[8];
        words_from_limbs<8>(kw, soa_ld<ModN, 1>(V.pSR, p).l);
        acc = p256_rtab_mul(W.rtab + (size_t)p * rtab_words(RTAB_VERIFY_BITS), kw, RTAB_VERIFY_BITS);
        shv = soa_ld<ModN, 1>(V.pSH, p);
    }
    acc = pm_block_sum(acc, 64, sh);
#pragma unroll
    for (int l = 0; l < 9; l++) shs[l][t] = shv.l[l];
    __syncthreads();
    for (uint32_t o = 32; o >= 1; o >>= 1) {
        if (t < o) {
            Fe<ModN, 1> a, c2;
#pragma unroll
            for (int l = 0; l < 9; l++) a.l[l] = shs[l][t], c2.l[l] = shs[l][t + o];
            a = fe_add_mod(a, c2);
#pragma unroll
            for (int l = 0; l < 9; l++) shs[l][t] = a.l[l];
        }
        __syncthreads();
    }
    if (t == 0) {
        pm_st(M.rpart + ((size_t)g * PB + blockIdx.x) * PM_PT_WORDS, acc);
#pragma unroll
        for (int l = 0; l < 9; l++) M.shpart[((size_t)g * PB + blockIdx.x) * 9 + l] = shs[l][0];
    }
}
// Workgroup g, 128 threads: wave 0 adds up the group's R parts and SH and walks h_NIST's comb; wave 1 joins the windows (Horner); thread 0 checks.
template <int C>
__global__ void __launch_bounds__(128) k_pm_final(DevParams P, uint32_t PB, uint32_t G, PMsmBuf M, bool one_lane) {
    typedef PmShape<C> S;
    __shared__ uint32_t sh[27 * 32], shs[9][64], shw[27];
    const uint32_t g = blockIdx.x, t = threadIdx.x;
    P256Pt acc = p256_identity();
    Fe<ModN, 1> shv = fe_zero<ModN>();
    if (t < 64) {
#pragma unroll 1
        for (uint32_t b = t; b < PB; b += 64) {
            acc = p256_add(acc, ld_rtab(M.rpart + ((size_t)g * PB + b) * PM_PT_WORDS));
            Fe<ModN, 1> v;
#pragma unroll
            for (int l = 0; l < 9; l++) v.l[l] = M.shpart[((size_t)g * PB + b) * 9 + l];
            shv = fe_add_mod(shv, v);
        }
#pragma unroll
        for (int l = 0; l < 9; l++) shs[l][t] = shv.l[l];
    }
    if (one_lane) {
        if (t == 64) {   // round 5's form (A/B only): one lane walks the windows
            P256Pt hw = ld_rtab(M.Tw + ((size_t)(S::nw - 1) * G + g) * PM_PT_WORDS);
#pragma unroll 1
            for (int w = (int)S::nw - 2; w >= 0; w--) {
#pragma unroll 1
                for (int k = 0; k < C; k++) hw = p256_dbl(hw);
                hw = p256_add(hw, ld_rtab(M.Tw + ((size_t)w * G + g) * PM_PT_WORDS));
            }
            pm_sh_st(shw, 1, 0, hw);
        }
    } else if (t >= 64) {   // wave 1, cooperating (coop.h: one limb per lane, X, Y, Z in rows): (nw - 1) x (C doublings + 1 addition) in a row
        const CoU32 mj = co_limbs(ModQ::mod);
        CoP256 hw;
        hw.v = co_load_aos<ModQ, 8, 3>(M.Tw + ((size_t)(S::nw - 1) * G + g) * PM_PT_WORDS);
#pragma unroll 1
        for (int w = (int)S::nw - 2; w >= 0; w--) {
            CoP256 tw;
            tw.v = co_load_aos<ModQ, 8, 3>(M.Tw + ((size_t)w * G + g) * PM_PT_WORDS);
#pragma unroll 1
            for (int k = 0; k < C; k++) hw = co_p256_dbl(hw, mj);
            hw = co_p256_add(hw, tw, mj);
        }
        co_store_aos<ModQ, 8, 3>(shw, hw.v);   // the layout pm_sh_ld(shw, 1, 0) reads
    }
    // (the tree's barriers are reached by all 128 threads; only the first 64 hold anything)
    acc = pm_block_sum(acc, 64, sh);
    for (uint32_t o = 32; o >= 1; o >>= 1) {
        if (t < o) {
            Fe<ModN, 1> a, c2;
#pragma unroll
            for (int l = 0; l < 9; l++) a.l[l] = shs[l][t], c2.l[l] = shs[l][t + o];
            a = fe_add_mod(a, c2);
#pragma unroll
            for (int l = 0; l < 9; l++) shs[l][t] = a.l[l];
        }
        __syncthreads();
    }
    if (t == 0) {
        Fe<ModN, 1> s;
#pragma unroll
        for (int l = 0; l < 9; l++) s.l[l] = shs[l][0];
        uint32_t kw[8];
        words_from_limbs<8>(kw, s.l);
#pragma unroll 1
        for (int w = 0; w < PFIX_NWIN; w++) {
            const uint32_t d = kw[0] & (PFIX_WIN_SIZE - 1);
            shr256<PFIX_WIN_BITS>(kw);
            const P256Pt sum = p256_add_mixed(acc, ld_pfix(P.pfix_H + (size_t)PFIX_ENTRY_WORDS * (w * PFIX_WIN_SIZE + (d ? d : 1))));
            acc = p256_select(d != 0, sum, acc);
        }
        acc = p256_add(acc, pm_sh_ld(shw, 1, 0));
        const bool id = fe_is_zero(fe_reduce(acc.x)) && fe_is_zero(fe_reduce(acc.z)) && !fe_is_zero(fe_reduce(acc.y));   // weier.ts:117-119
        M.flag[g] = id && M.counters[1] == 0 ? 1u : 0u;
    }
}
// every group passed: the per-proof verdict k_v_final reads
__global__ void __launch_bounds__(256) k_pm_all_ok(VWork V, uint32_t count) {
    const uint32_t p = gtid();
    if (p < count) V.p256_ok[p] = pm_good(V, p) ? 1u : 0u;
}
void launch_pm_all_ok(hipStream_t s, const VWork& V, uint32_t count) {
    if (count) hipLaunchKernelGGL(k_pm_all_ok, dim3((count + 255) / 256), dim3(256), 0, s, V, count);
}

size_t pmsm_carve(PMsmBuf* M, uint8_t* base, uint32_t Ccap, uint32_t groups) {
    const size_t nw = groups == 64 ? PmShape<10>::nw : PmShape<13>::nw, nb = groups == 64 ? PmShape<10>::nb : PmShape<13>::nb;
    const size_t ncap = (size_t)Ccap * PM_TERMS, gcap_all = ((size_t)Ccap + groups) * PM_TERMS, nwgb = nw * groups * nb;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        uint8_t* p = base ? base + off : nullptr;
        off += (bytes + 255) & ~(size_t)255;
        return p;
    };
    PMsmBuf m{};
    m.ncap = (uint32_t)ncap;
    m.aos = (uint32_t*)take(ncap * 64);
    m.dig = (uint16_t*)take(nw * ncap * 2);
    m.vals = (uint32_t*)take(nw * gcap_all * 4);
    m.start = (uint32_t*)take(nwgb * 4), m.end = (uint32_t*)take(nwgb * 4), m.order = (uint32_t*)take(nwgb * 4);
    m.buckets = (uint32_t*)take(nwgb * PM_PT_WORDS * 4);
    m.big_list = (uint32_t*)take(PM_BIG_MAX * 4);
    m.Tw = (uint32_t*)take(nw * groups * PM_PT_WORDS * 4);
    const size_t parts = (size_t)Ccap / 64 + 2 * groups;
    m.rpart = (uint32_t*)take(parts * PM_PT_WORDS * 4), m.shpart = (uint32_t*)take(parts * 9 * 4);
    m.counters = (uint32_t*)take(256), m.flag = (uint32_t*)take(256);
    if (M) *M = m;
    return off;
}
// part 1 (stage 1 of the chunk, on its stream): everything that needs no arithmetic throughput -- entries and digits, the counting sort, the R parts
template <int C>
static void pmsm_prepare_t(hipStream_t s, const Workspace& W, const VWork& V, uint32_t count, uint32_t G, PMsmBuf M) {
    typedef PmShape<C> S;
    const uint32_t gsz = (count + G - 1) / G, PB = (gsz + 63) / 64;
    M.gcap = gsz * PM_TERMS;
    hipMemsetAsync(M.counters, 0, 256, s);
    hipLaunchKernelGGL(k_pm_pack<C>, dim3((count * PM_TERMS + 255) / 256), dim3(256), 0, s, V, count, M);
    hipLaunchKernelGGL(k_pm_group<C>, dim3(S::nw, G), dim3(1024), 0, s, V, count, gsz, M);
    hipLaunchKernelGGL(k_pm_rpart, dim3(PB, G), dim3(64), 0, s, W, V, count, gsz, M);
}
// part 2 (stage 2, beside the Tom-256 pass): the bucket sums and their reduction
template <int C>
static void pmsm_sums_t(hipStream_t s, const DevParams& P, uint32_t count, uint32_t G, PMsmBuf M, uint32_t* host_flags_pinned) {
    typedef PmShape<C> S;
    const uint32_t gsz = (count + G - 1) / G, PB = (gsz + 63) / 64;
    M.gcap = gsz * PM_TERMS;
    hipLaunchKernelGGL(k_pm_bucket<C>, dim3(S::nw * G * S::nb / 256), dim3(256), 0, s, M);
    hipLaunchKernelGGL(k_pm_big<C>, dim3(256), dim3(256), 0, s, M);
    hipLaunchKernelGGL(k_pm_reduce<C>, dim3(S::nw * G), dim3(256), 0, s, M);
    hipLaunchKernelGGL(k_pm_final<C>, dim3(G), dim3(128), 0, s, P, PB, G, M, zk_one_lane_chains());
    launch_words_to_host(s, host_flags_pinned, M.flag, G);
}
void pmsm_prepare(hipStream_t s, const Workspace& W, const VWork& V, uint32_t count, const PMsmBuf& M, uint32_t groups) {
    if (groups == 64) pmsm_prepare_t<10>(s, W, V, count, groups, M);
    else pmsm_prepare_t<13>(s, W, V, count, groups, M);
}
// host_flags_pinned[g] (page-locked) = 1 once s has drained: the P-256 total of group g (proofs [g * gsz, (g + 1) * gsz), gsz = ceil(count / groups) as in
// run_msm) is the identity.  No host round trip of its own.
void pmsm_sums(hipStream_t s, const DevParams& P, uint32_t count, const PMsmBuf& M, uint32_t groups, uint32_t* host_flags_pinned) {
    if (groups == 64) pmsm_sums_t<10>(s, P, count, groups, M, host_flags_pinned);
    else pmsm_sums_t<13>(s, P, count, groups, M, host_flags_pinned);
}
