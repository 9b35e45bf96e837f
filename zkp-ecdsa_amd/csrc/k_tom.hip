// Tom-256 kernels: Pedersen commitments by fixed-base comb, batch normalisation to affine, and the few point
// additions the PointAdd sub-protocol needs.
//
// Reference: PedersenParams.commit (src/commit/pedersen.ts:53-58) = h.dblmul(r, g, v), i.e. the Straus/Shamir
// window-4 double-and-add of src/curves/group.ts:97-132 (256 dbl + 160 add = 4064 modmuls).  Only the AFFINE
// result is observable (hash input / proof bytes), so the engine evaluates v*g + r*h as additions of
// precomputed multiples (8 modmuls each, no doublings): 2 * ceil(256/W) additions for a W-bit comb (256 modmuls at W = 16,
// 176 at W = 24).
//
// proveMult's variable-base products C4 = x*Cy and A4_2 = kx*Cy (src/commit/mult.ts:103,114) are also commitments
// with KNOWN openings (x*y, x*ry), so they go through the same kernel (see k_scalar.hip).
#include "engine.h"
#include "comb_digits.h"
#include "coop_dev.h"   // commitments of a call of a few proofs on cooperating waves (k_tom_commit_co)

ZK_DEV TomNiels ld_niels(const uint32_t* e) {
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

// An entry is loaded one window ahead and crosses the loop back edge.  At its point of USE its limbs are pinned as 32-bit
// values (see limbs_mont_mul: otherwise some reach the multiplier as 64-bit pieces of the dwordx4 loads).  Not at the load:
// the asm operand would make the wave wait for the gather it has just issued.
ZK_DEV void niels_pin(TomNiels& n) {
#if ZK_PIN_LIMBS32
#pragma unroll
    for (int l = 0; l < 9; l++) asm("" : "+v"(n.x.l[l]), "+v"(n.y.l[l]), "+v"(n.dt.l[l]));
#endif
}

// unpaired / paired commitment slots of list B (see k_tom_commit_pairs)
__device__ const uint8_t LB_SINGLE_K[16] = {0, 1, 2, 6, 7, 8, 11, 13, 14, 17, 19, 20, 23, 25, 26, 29};
__device__ const uint8_t LB_PAIR_K0[9] = {9, 15, 21, 27, 30, 32, 3, 4, 5};
__device__ const uint8_t LB_PAIR_K1[9] = {10, 16, 22, 28, 31, 33, 12, 18, 24};
template <bool SGN>
struct NielsSel;  // table entry as used by the addition: as loaded (unsigned combs) or conditionally negated (signed combs)
template <>
struct NielsSel<false> {
    typedef TomNiels T;
    static ZK_DEV T sel(const TomNiels& q, bool) { return q; }
};
template <>
struct NielsSel<true> {
    typedef TomNielsT<4> T;
    static ZK_DEV T sel(const TomNiels& q, bool neg) { return tom_niels_neg_sel(q, neg); }
};
template <int OCC, bool SGN>
__global__ void __launch_bounds__(256, OCC) k_tom_commit(const uint32_t* __restrict__ tab_g, const uint32_t* __restrict__ tab_h, TomList L,
                                                         uint32_t count, uint32_t per_group, uint32_t slots_per_group, uint32_t kstride,
                                                         uint32_t bits, uint32_t nwin, uint32_t lb_singles) {
    uint32_t c = gtid();
    if (c >= count) return;
    // kstride == 0: slot = group * slots_per_group + member;  kstride != 0 (list B): per_group = items, slot = k * kstride + item
    // (lb_singles: k runs over the unpaired slots of list B only, see k_tom_commit_pairs)
    uint32_t kk = c / per_group;
    if (lb_singles) kk = LB_SINGLE_K[kk];
    uint32_t slot = kstride ? kk * kstride + (c % per_group) : kk * slots_per_group + (c % per_group);
    CombDigits dgv, dgr;
    dgv.init(bits), dgr.init(bits);
    {
        Fe<ModQ, 1> v = soa_ld<ModQ, 1>(L.v, slot), r = soa_ld<ModQ, 1>(L.r, slot);
        words_from_limbs<8>(dgv.w, v.l);
        words_from_limbs<8>(dgr.w, r.l);
    }
    const uint32_t ent = tom_win_entries(bits);
    TomPt acc = tom_identity();
    // software pipeline, one addition deep: the h-entry of window w is gathered during the g-addition of window w, the
    // g-entry of window w+1 during the h-addition (one entry in flight, one in use: 54 VGPRs instead of 108)
    // A g-window whose digit is 0 in EVERY lane of the wave is skipped (entry 0 is the identity): list B is item-fastest, so a wave
    // holds one slot of 64 items, and slot 6 -- C4 of pi8, the commitment to i7 * i8 = 1 (mult.ts:103 with x = i7, y = 1/i7) -- has
    // v in {0, 1} for all of them: 10 of its 11 g-additions go.  Unsigned combs only (a signed zero digit is not entry 0).
    uint32_t dv, dr;
    bool sv, sr, g_live = true;
    dgv.next(dv, sv);
    TomNiels ng = ld_niels(tab_g + (size_t)TOM_ENTRY_WORDS * dv), nh;
#pragma unroll 1
    for (uint32_t w = 0; w < nwin; w++) {
        size_t base = (size_t)w * ent;
        dgr.next(dr, sr);
        nh = ld_niels(tab_h + (size_t)TOM_ENTRY_WORDS * (base + dr));
        if (g_live) {
            niels_pin(ng);
            typename NielsSel<SGN>::T cg = NielsSel<SGN>::sel(ng, sv);
            acc = w == 0 ? tom_from_niels(cg) : tom_add_niels(acc, cg);                  // first step: identity + entry
        }
        if (w + 1 < nwin) {
            dgv.next(dv, sv);
            g_live = SGN || ZK_UNIFORM_CF || __ballot(dv != 0) != 0;   // wave-uniform; the uniform build (curve.h: ZK_UNIFORM_CF) never skips: the digits are secret
            if (g_live) ng = ld_niels(tab_g + (size_t)TOM_ENTRY_WORDS * (base + ent + dv));
        }
        niels_pin(nh);
        {
            typename NielsSel<SGN>::T ch = NielsSel<SGN>::sel(nh, sr);
            acc = w + 1 == nwin ? tom_add_niels_last(acc, ch) : tom_add_niels(acc, ch);  // last step: nobody reads T
        }
    }
    soa_st(L.proj.x, slot, acc.x);
    soa_st(L.proj.y, slot, acc.y);
    soa_st(L.proj.z, slot, acc.z);
}
// ---- list B of provePointAdd (34 commitments per zero-bit repetition, slot = k * kstride + item).  Nine pairs commit
// to the SAME value under two blinding factors: A_z / A_4_1 of each proveMult (k_z, mult.ts:110-113), A_1 / A_2 of each
// proveEquality (k, equality.ts:62-64), and -- round 2 -- C_z / C_4 of pi10, pi11, pi13: C_4 = x * C_y opens to x * y
// (mult.ts:103), which is the z the proof is about, i.e. the value of C10, C11, C13 (pointAdd.ts:133-136: i10 = i8 i9,
// i11 = i10 i10, i13 = i10 i12).  v*g is accumulated once per pair and both h-parts continue from it: 3 * nwin additions
// instead of 4 * nwin.  With the skipped g-windows of pi8's C4 (k_tom_commit): 639 instead of 748 additions per item.
// Units 0..15 are the unpaired slots.
#define LB_UNITS_SINGLE 16
#define LB_UNITS_PAIR 9
// acc += sum_w tab[w][digit_w(words)], gathers pipelined one window ahead.  FIRST: acc is the identity (the first entry is
// taken as is); LAST: the result is final (no T coordinate).
template <bool FIRST, bool LAST, bool SGN>
ZK_DEV TomPt tom_comb_acc(TomPt acc, const uint32_t* __restrict__ tab, uint32_t* words, uint32_t bits, uint32_t nwin) {
    CombDigits dg;
    dg.init(bits);
#pragma unroll
    for (int i = 0; i < 8; i++) dg.w[i] = words[i];
    const uint32_t ent = tom_win_entries(bits);
    uint32_t d;
    bool sg, nsg = false;
    dg.next(d, sg);
    TomNiels nx = ld_niels(tab + (size_t)TOM_ENTRY_WORDS * d);
#pragma unroll 1
    for (uint32_t w = 0; w < nwin; w++) {
        niels_pin(nx);
        typename NielsSel<SGN>::T cur = NielsSel<SGN>::sel(nx, sg);
        if (w + 1 < nwin) {
            dg.next(d, nsg);
            nx = ld_niels(tab + (size_t)TOM_ENTRY_WORDS * ((size_t)(w + 1) * ent + d));
        }
        sg = nsg;
        if (FIRST && w == 0) acc = tom_from_niels(cur);
        else if (LAST && w + 1 == nwin) acc = tom_add_niels_last(acc, cur);
        else acc = tom_add_niels(acc, cur);
    }
    return acc;
}
template <bool SGN>
__global__ void __launch_bounds__(256, 2) k_tom_commit_pairs(const uint32_t* __restrict__ tab_g, const uint32_t* __restrict__ tab_h, TomList L,
                                                             uint32_t items, uint32_t kstride, uint32_t bits, uint32_t nwin) {
    uint32_t c = gtid();
    if (c >= items * LB_UNITS_PAIR) return;
    uint32_t slot0 = LB_PAIR_K0[c / items] * kstride + c % items, slot1 = LB_PAIR_K1[c / items] * kstride + c % items;
    uint32_t w8[8];
    words_from_limbs<8>(w8, soa_ld<ModQ, 1>(L.v, slot0).l);
    TomPt G = tom_comb_acc<true, false, SGN>(tom_identity(), tab_g, w8, bits, nwin);
    words_from_limbs<8>(w8, soa_ld<ModQ, 1>(L.r, slot0).l);
    TomPt A = tom_comb_acc<false, true, SGN>(G, tab_h, w8, bits, nwin);
    soa_st(L.proj.x, slot0, A.x), soa_st(L.proj.y, slot0, A.y), soa_st(L.proj.z, slot0, A.z);
    words_from_limbs<8>(w8, soa_ld<ModQ, 1>(L.r, slot1).l);
    A = tom_comb_acc<false, true, SGN>(G, tab_h, w8, bits, nwin);
    soa_st(L.proj.x, slot1, A.x), soa_st(L.proj.y, slot1, A.y), soa_st(L.proj.z, slot1, A.z);
}
// ---- the same commitments for a SMALL launch, four lanes each (engine.h: ZK_WIDE_MAX_UNITS).  The comb tables hold every window multiple, so lane `part`
// takes windows [part * per, part * per + per) of v's g-part and r's h-part -- 3 + 3 table additions at 24 bits instead of 11 + 11 in a row -- and the four
// partial points are added through the wave's cross-lane moves (two complete extended additions).  Same group element, hence the same affine
// coordinates once the list is normalised.  Unsigned combs only (a signed comb's zero digit is not the identity entry; those widths keep one lane).
ZK_DEV TomPt tom_shfl_xor(const TomPt& a, int m) {
    TomPt r;
#pragma unroll
    for (int l = 0; l < NLIMB; l++) {
        r.x.l[l] = (uint32_t)__shfl_xor((int)a.x.l[l], m), r.y.l[l] = (uint32_t)__shfl_xor((int)a.y.l[l], m);
        r.t.l[l] = (uint32_t)__shfl_xor((int)a.t.l[l], m), r.z.l[l] = (uint32_t)__shfl_xor((int)a.z.l[l], m);
    }
    return r;
}
// acc + sum over windows [w0, w0 + per) of tab[w][digit_w(k)]; windows past the last one (and what the scalar has no bits for) add entry 0 of window 0, the identity
ZK_DEV TomPt tom_comb_range(TomPt acc, const uint32_t* __restrict__ tab, const uint32_t kw[8], uint32_t bits, uint32_t nwin, uint32_t w0, uint32_t per) {
    CombDigits dg;
    dg.init(bits);
#pragma unroll
    for (int i = 0; i < 8; i++) dg.w[i] = kw[i];
    const uint32_t ent = tom_win_entries(bits);
    uint32_t d;
    bool sg;
#pragma unroll 1
    for (uint32_t w = 0; w < w0; w++) dg.next(d, sg);
#pragma unroll 1
    for (uint32_t j = 0; j < per; j++) {
        const uint32_t w = w0 + j;
        dg.next(d, sg);
        const bool in = w < nwin;
        TomNiels e = ld_niels(tab + (size_t)TOM_ENTRY_WORDS * (in ? (size_t)w * ent + d : 0));
        niels_pin(e);
        acc = tom_add_niels(acc, e);
    }
    return acc;
}
__global__ void __launch_bounds__(256) k_tom_commit_wide(const uint32_t* __restrict__ tab_g, const uint32_t* __restrict__ tab_h, TomList L, uint32_t count, uint32_t per_group,
                                                         uint32_t slots_per_group, uint32_t kstride, uint32_t bits, uint32_t nwin) {
    const uint32_t tt = gtid();
    const bool live = tt < count * 4;
    const uint32_t c = live ? tt >> 2 : count - 1, part = tt & 3;   // dead lanes of the last wave mirror the last commitment: the cross-lane moves need every lane of a group
    const uint32_t kk = c / per_group;
    const uint32_t slot = kstride ? kk * kstride + (c % per_group) : kk * slots_per_group + (c % per_group);
    uint32_t vw[8], rw[8];
    words_from_limbs<8>(vw, soa_ld<ModQ, 1>(L.v, slot).l);
    words_from_limbs<8>(rw, soa_ld<ModQ, 1>(L.r, slot).l);
    const uint32_t per = (nwin + 3) / 4;
    TomPt acc = tom_comb_range(tom_identity(), tab_g, vw, bits, nwin, part * per, per);
    acc = tom_comb_range(acc, tab_h, rw, bits, nwin, part * per, per);
    acc = tom_add(acc, tom_shfl_xor(acc, 1));
    acc = tom_add(acc, tom_shfl_xor(acc, 2));
    if (!live || part) return;
    soa_st(L.proj.x, slot, acc.x);
    soa_st(L.proj.y, slot, acc.y);
    soa_st(L.proj.z, slot, acc.z);
}
// ---- and for a call of a few proofs: one commitment per workgroup of four cooperating waves (coop.h).  Wave `part` adds the table entries of its quarter of the
// windows (g-part, then h-part) at three passes an addition -- 0.9 us instead of one lane's 4.5 -- and the four partial points meet in LDS.  The same group
// element again, hence the same bytes.  A zero digit's entry is the identity and is added like any other (no digit-dependent control flow here at all).
ZK_DEV CoTom co_tom_comb_range(CoTom acc, const uint32_t* __restrict__ tab, const uint32_t kw[8], uint32_t bits, uint32_t nwin, uint32_t w0, uint32_t per, const CoU32& mj) {
    CombDigits dg;
    dg.init(bits);
#pragma unroll
    for (int i = 0; i < 8; i++) dg.w[i] = kw[i];
    const uint32_t ent = tom_win_entries(bits);
    uint32_t d;
    bool sg;
#pragma unroll 1
    for (uint32_t w = 0; w < w0; w++) dg.next(d, sg);
#pragma unroll 1
    for (uint32_t w = w0; w < w0 + per && w < nwin; w++) {
        dg.next(d, sg);
        CoFe<ModT, 2> e = co_load_aos<ModT, 2, 3>(tab + (size_t)TOM_ENTRY_WORDS * ((size_t)w * ent + d));   // rows x, y, d'T
        if (co_row_index() == 3) e.v = co_limbs(ModT::one);                                              // Z = 1
        acc = co_tom_add_tab(acc, e, false, mj);
    }
    return acc;
}
__global__ void __launch_bounds__(256) k_tom_commit_co(const uint32_t* __restrict__ tab_g, const uint32_t* __restrict__ tab_h, TomList L, uint32_t count, uint32_t per_group,
                                                       uint32_t slots_per_group, uint32_t kstride, uint32_t bits, uint32_t nwin) {
    __shared__ uint32_t partial[3][64];
    const uint32_t c = blockIdx.x, part = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t kk = c / per_group;
    const uint32_t slot = kstride ? kk * kstride + (c % per_group) : kk * slots_per_group + (c % per_group);
    const CoU32 mj = co_limbs(ModT::mod);
    uint32_t vw[8], rw[8];
    words_from_limbs<8>(vw, soa_ld<ModQ, 1>(L.v, slot).l);
    words_from_limbs<8>(rw, soa_ld<ModQ, 1>(L.r, slot).l);
    const uint32_t per = (nwin + 3) / 4;
    CoTom acc = co_tom_comb_range(co_tom_identity(), tab_g, vw, bits, nwin, part * per, per, mj);
    acc = co_tom_comb_range(acc, tab_h, rw, bits, nwin, part * per, per, mj);
    if (part) partial[part - 1][lane] = acc.v.v;
    __syncthreads();
    if (part) return;
#pragma unroll 1
    for (uint32_t k = 0; k < 3; k++) {
        CoTom o;
        o.v.v = partial[k][lane];
        acc = co_tom_add(acc, o, mj);
    }
    co_store_soa(acc.v, slot, L.proj.x, L.proj.y, Soa{nullptr, 0}, L.proj.z);   // rows X, Y, T, Z
}
static bool tom_co(const DevParams& P, uint32_t count) { return !tom_signed(P.tom_bits) && (uint64_t)count * 4 <= ZK_COOP_MAX_CHAINS && !zk_one_lane_chains(); }
static void launch_tom_commit_co(hipStream_t s, const DevParams& P, const TomList& L, uint32_t count, uint32_t per_group, uint32_t slots_per_group, uint32_t kstride) {
    g_coop_chains.fetch_add((uint64_t)count * 4, std::memory_order_relaxed);
    hipLaunchKernelGGL(k_tom_commit_co, dim3(count), dim3(256), 0, s, P.tom_tab_g, P.tom_tab_h, L, count, per_group, slots_per_group, kstride, P.tom_bits, tom_nwin(P.tom_bits));
}
static bool tom_wide(const DevParams& P, uint32_t count) { return !tom_signed(P.tom_bits) && count <= ZK_WIDE_MAX_UNITS; }
void launch_tom_commit_listb(hipStream_t s, const DevParams& P, const TomList& L, uint32_t items, uint32_t kstride) {
    if (!items) return;
    if (tom_co(P, items * LB_COMMITS)) {
        launch_tom_commit_co(s, P, L, items * LB_COMMITS, items, 0u, kstride);
        return;
    }
    if (tom_wide(P, items * LB_COMMITS)) {   // all 34 slots of every item as independent commitments (the pairs' shared v * g is recomputed: the GPU is idle anyway)
        hipLaunchKernelGGL(k_tom_commit_wide, dim3((items * LB_COMMITS * 4 + 255) / 256), dim3(256), 0, s, P.tom_tab_g, P.tom_tab_h, L, items * LB_COMMITS, items, 0u, kstride,
                           P.tom_bits, tom_nwin(P.tom_bits));
        return;
    }
    uint32_t nwin = tom_nwin(P.tom_bits);
    uint32_t n1 = items * LB_UNITS_SINGLE, n2 = items * LB_UNITS_PAIR;
    if (tom_signed(P.tom_bits)) {
        hipLaunchKernelGGL((k_tom_commit<2, true>), dim3((n1 + 255) / 256), dim3(256), 0, s, P.tom_tab_g, P.tom_tab_h, L, n1, items, 0u, kstride, P.tom_bits, nwin, 1u);
        hipLaunchKernelGGL(k_tom_commit_pairs<true>, dim3((n2 + 255) / 256), dim3(256), 0, s, P.tom_tab_g, P.tom_tab_h, L, items, kstride, P.tom_bits, nwin);
    } else {
        hipLaunchKernelGGL((k_tom_commit<2, false>), dim3((n1 + 255) / 256), dim3(256), 0, s, P.tom_tab_g, P.tom_tab_h, L, n1, items, 0u, kstride, P.tom_bits, nwin, 1u);
        hipLaunchKernelGGL(k_tom_commit_pairs<false>, dim3((n2 + 255) / 256), dim3(256), 0, s, P.tom_tab_g, P.tom_tab_h, L, items, kstride, P.tom_bits, nwin);
    }
}
// k_tom_commit over a compacted list of slots (see launch_tom_commit_list): the grid covers the largest possible list, lanes past *count_dev leave at once
template <bool SGN>
__global__ void __launch_bounds__(256, 2) k_tom_commit_list(const uint32_t* __restrict__ tab_g, const uint32_t* __restrict__ tab_h, TomList L, const uint32_t* __restrict__ list,
                                                            const uint32_t* __restrict__ count_dev, uint32_t bits, uint32_t nwin) {
    const uint32_t c = gtid();
    if (c >= *count_dev) return;
    const uint32_t slot = list[c];
    uint32_t vw[8], rw[8];
    words_from_limbs<8>(vw, soa_ld<ModQ, 1>(L.v, slot).l);
    words_from_limbs<8>(rw, soa_ld<ModQ, 1>(L.r, slot).l);
    TomPt G = tom_comb_acc<true, false, SGN>(tom_identity(), tab_g, vw, bits, nwin);
    TomPt A = tom_comb_acc<false, true, SGN>(G, tab_h, rw, bits, nwin);
    soa_st(L.proj.x, slot, A.x), soa_st(L.proj.y, slot, A.y), soa_st(L.proj.z, slot, A.z);
}
void launch_tom_commit_list(hipStream_t s, const DevParams& P, const TomList& L, const uint32_t* list, const uint32_t* count_dev, uint32_t max_count) {
    if (!max_count) return;
    dim3 g((max_count + 255) / 256), b(256);
    if (tom_signed(P.tom_bits)) hipLaunchKernelGGL(k_tom_commit_list<true>, g, b, 0, s, P.tom_tab_g, P.tom_tab_h, L, list, count_dev, P.tom_bits, tom_nwin(P.tom_bits));
    else hipLaunchKernelGGL(k_tom_commit_list<false>, g, b, 0, s, P.tom_tab_g, P.tom_tab_h, L, list, count_dev, P.tom_bits, tom_nwin(P.tom_bits));
}
void launch_tom_commit(hipStream_t s, const DevParams& P, const TomList& L, uint32_t count, uint32_t per_group, uint32_t slots_per_group, uint32_t kstride) {
    if (!count) return;
    if (tom_co(P, count)) {
        launch_tom_commit_co(s, P, L, count, per_group, slots_per_group, kstride);
        return;
    }
    if (tom_wide(P, count)) {
        hipLaunchKernelGGL(k_tom_commit_wide, dim3((count * 4 + 255) / 256), dim3(256), 0, s, P.tom_tab_g, P.tom_tab_h, L, count, per_group, slots_per_group, kstride, P.tom_bits,
                           tom_nwin(P.tom_bits));
        return;
    }
    dim3 g((count + 255) / 256), b(256);
    if (tom_signed(P.tom_bits)) hipLaunchKernelGGL((k_tom_commit<2, true>), g, b, 0, s, P.tom_tab_g, P.tom_tab_h, L, count, per_group, slots_per_group, kstride, P.tom_bits, tom_nwin(P.tom_bits), 0u);
    else hipLaunchKernelGGL((k_tom_commit<2, false>), g, b, 0, s, P.tom_tab_g, P.tom_tab_h, L, count, per_group, slots_per_group, kstride, P.tom_bits, tom_nwin(P.tom_bits), 0u);
}

// Batch normalisation: (X:Y:Z) on the a=1 image -> affine (x, y) of the ORIGINAL curve, plain canonical limbs
// (edwards.ts:184-193 toAffine).  Montgomery's trick: each thread owns `per` elements (strided by the thread
// count), the 256 threads of a workgroup share ONE Fermat inversion (block_inverse, engine.h).  Prefix products are parked in
// the ax output array.
// Element c of the pass maps to list slot (c / per_group) * slots_per_group + first + c % per_group.
ZK_DEV uint32_t norm_slot(uint32_t c, uint32_t first, uint32_t per_group, uint32_t slots_per_group, uint32_t kstride) {
    return kstride ? (first + c / per_group) * kstride + (c % per_group) : (c / per_group) * slots_per_group + first + c % per_group;
}
// -DZK_NORM_PIPELINE=1 software-pipelines both passes by hand (the loads of element j + 1 -- j - 1 in the backward pass -- are issued before
// the products of element j; without it the stores to the output arrays, which may alias the inputs as far as the compiler knows, keep
// load -> multiply -> store -> load in program order).  Measured, same box (profiles/r04_ab_variants.txt): tom_normalize 10.90 -> 10.68 ms,
// p256_normalize 3.89 -> 3.77 ms per step, 78 -> 99 VGPRs, no change in proofs/s: the passes are not waiting for those loads.  A thread's
// products form ONE dependent chain (acc = acc * z), two or three in the backward pass, at two to four waves per SIMD -- the multiplier
// issues a dependent v_mad_u64_u32 every 8 cycles at two chains per SIMD against 5.7 at eight (profiles/r03_valu_peak_microbench.txt) -- and
// every workgroup waits once for thread 0's Fermat inversion (~380 dependent products).  More, thinner threads (per <= 32) lose: 12.5 ms,
// four times as many inversions.  Default off.
#ifndef ZK_NORM_PIPELINE
#define ZK_NORM_PIPELINE 0
#endif
__global__ void __launch_bounds__(256) k_tom_normalize(TomList L, uint32_t count, uint32_t nthreads, uint32_t per, uint32_t first,
                                                       uint32_t per_group, uint32_t slots_per_group, uint32_t kstride) {
    __shared__ uint32_t lds[2 * 256 * NLIMB];
    uint32_t t = gtid();   // threads beyond nthreads own no element but take part in the workgroup's inversion
    Ft2 acc = fe_one_mont<ModT>().as<2>();
    // elements of this thread: c = t + j * nthreads < count, j < per
    uint32_t mine = 0;
    if (t < nthreads && t < count) mine = std::min<uint32_t>(per, (count - t + nthreads - 1) / nthreads);
#if ZK_NORM_PIPELINE
    {
        Ft2 zn = fe_one_mont<ModT>().as<2>();
        uint32_t en = 0;
        if (mine) en = norm_slot(t, first, per_group, slots_per_group, kstride), zn = soa_ld<ModT, 2>(L.proj.z, en);
        for (uint32_t j = 0; j < mine; j++) {
            const uint32_t e = en;
            const Ft2 z = zn;
            if (j + 1 < mine) en = norm_slot(t + (j + 1) * nthreads, first, per_group, slots_per_group, kstride), zn = soa_ld<ModT, 2>(L.proj.z, en);
            soa_st(L.ax, e, acc);  // prefix product before element e
            acc = acc * z;
        }
    }
#else
    for (uint32_t j = 0; j < mine; j++) {
        uint32_t e = norm_slot(t + j * nthreads, first, per_group, slots_per_group, kstride);
        soa_st(L.ax, e, acc);  // prefix product before element e
        acc = acc * soa_ld<ModT, 2>(L.proj.z, e);
    }
#endif
    // The running inverse is kept in the PLAIN domain (one extra product per thread): a Montgomery product of a plain and a
    // Montgomery operand is plain, so 1/z_e, the next running inverse, x and y come out plain without the two
    // from-Montgomery products per point (5 products per point in this pass instead of 7).
    Fe<ModT, 1> one = fe_zero<ModT>();
    one.l[0] = 1;
    Ft2 inv = block_inverse<ModT>(acc, lds) * one;
    if (!mine) return;
    const auto sinv = fe_const<ModT, 1>(TOM_SINV_M);
#if ZK_NORM_PIPELINE
    uint32_t en = norm_slot(t + (mine - 1) * nthreads, first, per_group, slots_per_group, kstride);
    Ft2 zn = soa_ld<ModT, 2>(L.proj.z, en), pn = soa_ld<ModT, 2>(L.ax, en), xn = soa_ld<ModT, 2>(L.proj.x, en), yn = soa_ld<ModT, 2>(L.proj.y, en);
    for (int j = (int)mine - 1; j >= 0; j--) {
        const uint32_t e = en;
        const Ft2 z = zn, pre = pn, px = xn, py = yn;
        if (j > 0) {
            en = norm_slot(t + (uint32_t)(j - 1) * nthreads, first, per_group, slots_per_group, kstride);
            zn = soa_ld<ModT, 2>(L.proj.z, en), pn = soa_ld<ModT, 2>(L.ax, en), xn = soa_ld<ModT, 2>(L.proj.x, en), yn = soa_ld<ModT, 2>(L.proj.y, en);
        }
        Ft2 zi = inv * pre;
        inv = inv * z;
        Ft2 x = px * (zi * sinv);
        Ft2 y = py * zi;
        soa_st(L.ax, e, fe_canon(x));
        soa_st(L.ay, e, fe_canon(y));
    }
#else
    for (int j = (int)mine - 1; j >= 0; j--) {
        uint32_t e = norm_slot(t + (uint32_t)j * nthreads, first, per_group, slots_per_group, kstride);
        Ft2 z = soa_ld<ModT, 2>(L.proj.z, e);
        Ft2 zi = inv * soa_ld<ModT, 2>(L.ax, e);
        inv = inv * z;
        Ft2 x = soa_ld<ModT, 2>(L.proj.x, e) * (zi * sinv);
        Ft2 y = soa_ld<ModT, 2>(L.proj.y, e) * zi;
        soa_st(L.ax, e, fe_canon(x));
        soa_st(L.ay, e, fe_canon(y));
    }
#endif
}
// A launch of a few thousand points (every normaliser of a small call) is a chain of latencies: prefix products, eight scan rounds with barriers, the workgroup's
// inversion, the backward pass.  One thread per point with its OWN divsteps inversion (38 us whatever the number of lanes, tools/coop_bench.hip) is shorter: the same
// affine values -- an inverse is an inverse -- in canonical limbs, hence the same bytes.  Above ZK_NORM_EACH_MAX points the shared inversion wins on throughput.
#ifndef ZK_NORM_EACH_MAX
#define ZK_NORM_EACH_MAX 16384u
#endif
__global__ void __launch_bounds__(256) k_tom_normalize_each(TomList L, uint32_t count, uint32_t first, uint32_t per_group, uint32_t slots_per_group, uint32_t kstride) {
    const uint32_t c = gtid();
    if (c >= count) return;
    const uint32_t e = norm_slot(c, first, per_group, slots_per_group, kstride);
    Fe<ModT, 1> one = fe_zero<ModT>();
    one.l[0] = 1;
    const Ft2 zi = fe_inv<ModT, true>(soa_ld<ModT, 2>(L.proj.z, e)) * one;   // plain 1 / z: the products below come out plain (see k_tom_normalize)
    const Ft2 x = soa_ld<ModT, 2>(L.proj.x, e) * (zi * fe_const<ModT, 1>(TOM_SINV_M));
    const Ft2 y = soa_ld<ModT, 2>(L.proj.y, e) * zi;
    soa_st(L.ax, e, fe_canon(x));
    soa_st(L.ay, e, fe_canon(y));
}
void launch_tom_normalize(hipStream_t s, const TomList& L, uint32_t count, uint32_t first, uint32_t per_group, uint32_t slots_per_group, uint32_t kstride) {
    if (!count) return;
    if (count <= ZK_NORM_EACH_MAX && !zk_one_lane_chains()) {
        hipLaunchKernelGGL(k_tom_normalize_each, dim3((count + 255) / 256), dim3(256), 0, s, L, count, first, per_group, slots_per_group, kstride);
        return;
    }
    uint32_t per = count / ZK_NORM_MIN_THREADS;
    if (per < 4) per = 4;
    if (per > ZK_NORM_PER_MAX) per = ZK_NORM_PER_MAX;
    uint32_t nthreads = (count + per - 1) / per;
    hipLaunchKernelGGL(k_tom_normalize, dim3((nthreads + 255) / 256), dim3(256), 0, s, L, count, nthreads, per, first, per_group, slots_per_group, kstride);
}

// original-curve affine plain -> a=1 image extended Montgomery (no validation: engine-produced points)
ZK_DEV TomPt tom_from_affine_plain(const Fe<ModT, 1>& xp, const Fe<ModT, 1>& yp) {
    TomPt r;
    Ft2 x = fe_to_mont(xp);
    r.y = fe_to_mont(yp);
    r.x = x * fe_const<ModT, 1>(TOM_S_M);
    r.t = r.x * r.y;
    r.z = fe_one_mont<ModT>().as<2>();
    return r;
}
ZK_DEV TomPt ld_aff(const TomList& L, uint32_t slot) { return tom_from_affine_plain(soa_ld<ModT, 1>(L.ax, slot), soa_ld<ModT, 1>(L.ay, slot)); }

// The five derived commitments of provePointAdd whose affine encodings enter Fiat-Shamir hashes
// (src/exp/pointAdd.ts:137-160): C7 = C2 - C1, C9 = C5 - C4, C12 = C1 - C3, Cint_x = C3 + C1 + C2, Cint_y = C6 + C4
// with C1 = T1x, C2 = pkX, C3 = Tx_i, C4 = T1y, C5 = pkY, C6 = Ty_i.  Inputs are affine (lists A and B);
// outputs go to slots 34..38 of list B (projective) and are normalised with the rest of the list.
__global__ void __launch_bounds__(256) k_padd_derived(Workspace W, uint32_t items) {
    uint32_t t = gtid();
    if (t >= items * 5) return;
    uint32_t item = t % items, k = t / items;
    uint32_t proof = W.item_proof[item], rep = W.item_rep[item];
    uint32_t la = proof * (2 + 2 * W.sec);
    uint32_t lb0 = lbi(W, item, 0), lb1 = lbi(W, item, 1);
    TomPt r;
    if (k == 0) {        // C7 = pkX - T1x
        r = tom_add(ld_aff(W.la, la + 0), tom_neg(ld_aff(W.lb, lb0)));
    } else if (k == 1) { // C9 = pkY - T1y
        r = tom_add(ld_aff(W.la, la + 1), tom_neg(ld_aff(W.lb, lb1)));
    } else if (k == 2) { // C12 = T1x - Tx_i
        r = tom_add(ld_aff(W.lb, lb0), tom_neg(ld_aff(W.la, la + 2 + 2 * rep)));
    } else if (k == 3) { // Cint_x = Tx_i + T1x + pkX
        r = tom_add(tom_add(ld_aff(W.la, la + 2 + 2 * rep), ld_aff(W.lb, lb0)), ld_aff(W.la, la + 0));
    } else {             // Cint_y = Ty_i + T1y
        r = tom_add(ld_aff(W.la, la + 3 + 2 * rep), ld_aff(W.lb, lb1));
    }
    uint32_t slot = lbi(W, item, LB_COMMITS + k);
    soa_st(W.lb.proj.x, slot, r.x);
    soa_st(W.lb.proj.y, slot, r.y);
    soa_st(W.lb.proj.z, slot, r.z);
}
void launch_padd_derived(hipStream_t s, const Workspace& W, uint32_t items) {
    if (!items) return;
    hipLaunchKernelGGL(k_padd_derived, dim3((items * 5 + 255) / 256), dim3(256), 0, s, W, items);
}
