// P-256 kernels of the prover: ECDSA front end, the multiplications by R of the Exp commit phase, T1.
//
// Reference call sites (src/zkpAttestList.ts:104-145, src/exp/exp.ts:144-156,186-193): every scalar multiplication
// there is the window-4 Point.mul of src/curves/group.ts:133-152 (4448 modmuls).  Only affine results are
// observable, so the engine uses: 20-bit fixed-base combs for G and h_NIST (13 mixed complete additions), per-KEY tables of the
// ring for every multiple of the signer's public key (ktab.h: 33 mixed additions; R = u1 G + u2 pk turns alpha * R into
// (alpha u1) * G + (alpha u2) * pk), and -- for proofs whose `which` does not name their own key, and for rings above 2^16 keys -- a
// per-proof signed-digit table of R (rtab.h), shared by the sec + 1 multiplications by R of one proof (43 complete additions each).
// All additions are the complete RCB formulas the reference uses.
#include "rtab.h"   // engine.h (and with it ktab.h), the projective table entries of rtab.h
#include "coop_dev.h"   // the table sums of a call of a few proofs on cooperating waves (k_front_co, k_exp_commit_kt_co)

// k * B for a fixed base with a PFIX_WIN_BITS-bit comb table; k given as 8 little-endian words (clobbered)
ZK_DEV P256Pt p256_fixed_mul(const uint32_t* __restrict__ tab, uint32_t kw[8]) {
    P256Pt acc;
    {   // first window: identity + entry = the entry
        uint32_t d = kw[0] & (PFIX_WIN_SIZE - 1);
        shr256<PFIX_WIN_BITS>(kw);
        acc = p256_select(d != 0, p256_from_affine(ld_pfix(tab + (size_t)PFIX_ENTRY_WORDS * d)), p256_identity());
    }
#pragma unroll 1
    for (int w = 1; w < PFIX_NWIN; w++) {
        uint32_t d = kw[0] & (PFIX_WIN_SIZE - 1);
        shr256<PFIX_WIN_BITS>(kw);
        ZK_ADD_IF(d != 0, acc, p256_add_mixed(acc, ld_pfix(tab + (size_t)PFIX_ENTRY_WORDS * (w * PFIX_WIN_SIZE + (d ? d : 1)))));   // a zero digit (2^-20) idles its lane (rtab.h: ZK_UNIFORM_CF)
    }
    return acc;
}
// acc + k * B: the comb's additions go straight onto a running point (one complete addition less than summing two results)
ZK_DEV P256Pt p256_fixed_mul_acc(P256Pt acc, const uint32_t* __restrict__ tab, uint32_t kw[8]) {
#pragma unroll 1
    for (int w = 0; w < PFIX_NWIN; w++) {
        uint32_t d = kw[0] & (PFIX_WIN_SIZE - 1);
        shr256<PFIX_WIN_BITS>(kw);
        ZK_ADD_IF(d != 0, acc, p256_add_mixed(acc, ld_pfix(tab + (size_t)PFIX_ENTRY_WORDS * (w * PFIX_WIN_SIZE + (d ? d : 1)))));   // a zero digit (2^-20) idles its lane (rtab.h: ZK_UNIFORM_CF)
    }
    return acc;
}
ZK_DEV void st_proj(const Soa3& a, uint32_t e, const P256Pt& p) {
    soa_st(a.x, e, p.x), soa_st(a.y, e, p.y), soa_st(a.z, e, p.z);
}
ZK_DEV P256Pt ld_proj(const Soa3& a, uint32_t e) {
    P256Pt p;
    p.x = soa_ld<ModQ, 8>(a.x, e), p.y = soa_ld<ModQ, 8>(a.y, e), p.z = soa_ld<ModQ, 8>(a.z, e);
    return p;
}

// ---------------------------------------------------------------- front end (zkpAttestList.ts:112-136)
// Three kernels, one thread per proof each, so that no live set exceeds 256 registers:
//   k_front          key check, the scalars u1, u2, s1, z1 mod n (two Fermat inversions)
//   k_front_table    1..8 times pk and the signed 4-bit digits of u2; u1*G and Q = z1*G by the fixed-base comb
//   k_front_walk     u2*pk by 65 windows of 4 doublings + one addition (260 + 72 point operations instead of the 256 + 256 of a
//                    bit-serial ladder), R = u1*G + u2*pk, R affine
// Between the kernels the values live in the proof's R-table area, which k_rtab_* only writes afterwards: entries 0..7 = d * pk,
// entry 8 = u1*G, words of entry 9: u2's digits (65 bytes), entry 10: u1, z1.
#define FRONT_NW 65   // signed 4-bit digits of a 256-bit scalar
ZK_DEV uint32_t* front_area(const Workspace& W, uint32_t p) { return W.rtab + (size_t)p * rtab_words(RTAB_PROVE_BITS); }
static_assert(FRONT_NW <= 4 * RTAB_ENTRY_WORDS && 2 * NLIMB <= RTAB_ENTRY_WORDS, "front-end scratch inside R-table entries");
__global__ void __launch_bounds__(64, 2) k_front(DevParams P, Workspace W, ChunkIn in) {
    uint32_t p = gtid();
    if (p >= in.count) return;
    uint32_t xw[8], yw[8], zw[8], rw[8], sw[8];
    load_be32(in.pk + 64 * (size_t)p, xw);
    load_be32(in.pk + 64 * (size_t)p + 32, yw);
    load_be32(in.msg + 32 * (size_t)p, zw);
    load_be32(in.sig + 64 * (size_t)p, rw);
    load_be32(in.sig + 64 * (size_t)p + 32, sw);
    int32_t status = ZK_OK;
    // values[index] must exist (gk.ts:162: `undefined.k` is a TypeError).  proveMembership runs last (zkpAttestList.ts:141-142), so an
    // invalid key (below) and 'T[i] is at infinity' (k_front_walk, the normaliser) replace this status; indices in [n_keys, N) name the
    // padding, i.e. keys[0] (gk.ts:75-86)
    if (in.which[p] >= W.N) status = ZK_E_ARG;
    // deserializePoint (weier.ts:74-89): isOnGroup works mod p, coordinates are not range-checked
    Fe<ModQ, 1> pkx = fe_from_words256_reduce<ModQ>(xw), pky = fe_from_words256_reduce<ModQ>(yw);
    P256Aff pk;
    pk.x = fe_to_mont(pkx), pk.y = fe_to_mont(pky);
    if (!p256_on_curve(pk)) status = ZK_E_POINT_NOT_IN_GROUP;
    soa_st(W.pkx, p, pkx), soa_st(W.pky, p, pky);
    soa_st(W.pkxm, p, pk.x), soa_st(W.pkym, p, pk.y);
    // scalars mod n (zkpAttestList.ts:119-127,133-135); invMod(0) = 0
    Fn2 z = fe_to_mont(fe_from_words256_reduce<ModN>(zw));
    Fn2 r = fe_to_mont(fe_from_words256_reduce<ModN>(rw));
    Fn2 s = fe_to_mont(fe_from_words256_reduce<ModN>(sw));
    // one Fermat inversion for both (Montgomery's trick): 1 / (s r), times r and times s.  invMod(0) = 0 (the reference's convention) is kept per
    // operand: a zero operand is replaced by 1 inside the product and its inverse forced to 0 afterwards
    const bool s0 = fe_is_zero(s), r0 = fe_is_zero(r);
    const Fn2 one = fe_one_mont<ModN>().as<2>();
    const Fn2 sn = fe_select(s0, one, s), rn = fe_select(r0, one, r);
    const Fn2 inv_sr = fe_inv<ModN>(sn * rn);
    const Fn2 zero2 = fe_zero<ModN>().as<2>();
    Fn2 sinv = fe_select(s0, zero2, Fn2(inv_sr * rn)), rinv = fe_select(r0, zero2, Fn2(inv_sr * sn));
    Fe<ModN, 1> u1 = fe_from_mont(sinv * z), u2 = fe_from_mont(sinv * r);
    Fe<ModN, 1> s1 = fe_from_mont(rinv * s), z1 = fe_from_mont(rinv * z);
    soa_st(W.s1, p, s1);
    // key-table path (k_ktab.hip): the ring value this proof names is its own key's x and that key has a table; + / - by the root
    uint32_t use = 0;
    if (W.ktab && status == ZK_OK && W.ktab_ok[in.which[p]]) {
        Fe<ModQ, 1> rv = soa_ld<ModQ, 1>(W.ring, in.which[p]);
        bool same = true;
#pragma unroll
        for (int l = 0; l < NLIMB; l++) same = same && rv.l[l] == pkx.l[l];
        if (same) {
            P256Aff b0 = ld_ktab(W.ktab + (size_t)in.which[p] * KTAB_KEY_WORDS);   // window 0, slot 0: the base point
            use = fe_eq(pk.y, b0.y) ? 1 : 2;
        }
    }
    W.kt_use[p] = (uint8_t)use, W.kt_key[p] = in.which[p];
    W.r_zero[p] = r0 ? 1 : 0;   // r = 0 mod n: s1 = z1 = 0 and the relation T1 + pk = T_i of k_t1 does not hold -- k_scan gives the proof the reference's exception
    soa_st(W.u1m, p, fe_canon(sinv * z)), soa_st(W.u2m, p, fe_canon(sinv * r));
    uint32_t* area = front_area(W, p);
    {
        uint8_t* dig = (uint8_t*)(area + 9 * RTAB_ENTRY_WORDS);
        uint32_t u2w[8], carry = 0;
        words_from_limbs<8>(u2w, u2.l);
#pragma unroll 1
        for (uint32_t w = 0; w < FRONT_NW; w++) {
            uint32_t d = (u2w[0] & 15) + carry;
            shr256<4>(u2w);
            bool neg = d > 8;
            carry = neg ? 1 : 0;
            if (neg) d = 16 - d;
            dig[w] = (uint8_t)(d | (neg ? 0x80u : 0u));
        }
        uint32_t* sc = area + 10 * RTAB_ENTRY_WORDS;
#pragma unroll
        for (int l = 0; l < NLIMB; l++) sc[l] = u1.l[l], sc[NLIMB + l] = z1.l[l];
    }
    W.st[p] = status;
}
// Small chunks: proofs on the key-table path get their three table sums -- u1 * G and u2 * pk for R, z1 * G for Q -- from EIGHT lanes each (k_front_wide:
// four per sum, a quarter of the windows per lane, two cross-lane additions) instead of 13 + 33 + 13 additions in a row in one lane; k_front_table and
// k_front_walk then only see the proofs that are off that path (skip_kt).
__global__ void __launch_bounds__(256) k_front_wide(DevParams P, Workspace W, uint32_t count) {
    const uint32_t tt = gtid();
    const bool live = tt < count * 8;
    const uint32_t p = live ? tt >> 3 : count - 1, part = tt & 3, which = live ? (tt >> 2) & 1 : 0;   // dead lanes of the last wave mirror a live sum: the cross-lane moves need every lane of a group
    const uint32_t use = W.kt_use[p];
    if (!use) return;   // (all eight lanes of the proof leave together)
    const uint32_t* sc = front_area(W, p) + 10 * RTAB_ENTRY_WORDS;
    Fe<ModN, 1> k;
#pragma unroll
    for (int l = 0; l < NLIMB; l++) k.l[l] = sc[(which ? NLIMB : 0) + l];   // u1 (R's sum) or z1 (Q's)
    uint32_t kw[8];
    words_from_limbs<8>(kw, k.l);
    constexpr uint32_t gper = (PFIX_NWIN + 3) / 4, kper = (KTAB_NWIN + 3) / 4;
    P256Pt acc = p256_fixed_mul_range(p256_identity(), P.pfix_G, kw, part * gper, gper);
    if (!which) {
        words_from_limbs<8>(kw, fe_from_mont(soa_ld<ModN, 1>(W.u2m, p).as<2>()).l);
        acc = p256_ktab_mul_range(acc, W.ktab + (size_t)W.kt_key[p] * KTAB_KEY_WORDS, kw, use == 2, part * kper, kper);
    }
    acc = p256_quad_sum(acc);
    if (!live || part) return;
    if (which) {
        st_proj(W.Q, p, acc);
        return;
    }
    // R affine (output + base of the per-proof table), as in k_front_walk
    Fq2 rz = fe_reduce(acc.z);
    if (fe_is_zero(rz) && (W.st[p] == ZK_OK || W.st[p] == ZK_E_ARG)) W.st[p] = ZK_E_T_INF;
    Fq2 zi = fe_inv<ModQ>(rz);
    Fq2 rx = acc.x * zi, ry = acc.y * zi;
    soa_st(W.Rxm, p, rx), soa_st(W.Rym, p, ry);
    soa_st(W.Rx, p, fe_from_mont(rx)), soa_st(W.Ry, p, fe_from_mont(ry));
}
__global__ void __launch_bounds__(64, 2) k_front_table(DevParams P, Workspace W, uint32_t count, uint32_t skip_kt) {
    uint32_t p = gtid();
    if (p >= count) return;
    if (skip_kt && W.kt_use[p]) return;
    uint32_t* area = front_area(W, p);
    if (!W.kt_use[p]) {
        P256Aff pk;
        pk.x = soa_ld<ModQ, 2>(W.pkxm, p), pk.y = soa_ld<ModQ, 2>(W.pkym, p);
        P256Pt base = p256_from_affine(pk), m = base;
        st_rtab(area, m);
        m = p256_dbl(base);
        st_rtab(area + RTAB_ENTRY_WORDS, m);
#pragma unroll 1
        for (uint32_t d = 2; d < 8; d++) {
            m = p256_add(m, base);
            st_rtab(area + d * RTAB_ENTRY_WORDS, m);
        }
    }
    const uint32_t* sc = area + 10 * RTAB_ENTRY_WORDS;
    Fe<ModN, 1> u1, z1;
#pragma unroll
    for (int l = 0; l < NLIMB; l++) u1.l[l] = sc[l], z1.l[l] = sc[NLIMB + l];
    uint32_t kw[8];
    words_from_limbs<8>(kw, u1.l);
    st_rtab(area + 8 * RTAB_ENTRY_WORDS, p256_fixed_mul(P.pfix_G, kw));
    words_from_limbs<8>(kw, z1.l);
    st_proj(W.Q, p, p256_fixed_mul(P.pfix_G, kw));
}
__global__ void __launch_bounds__(64, 2) k_front_walk(Workspace W, uint32_t count, uint32_t skip_kt) {
    uint32_t p = gtid();
    if (p >= count) return;
    if (skip_kt && W.kt_use[p]) return;
    const uint32_t* area = front_area(W, p);
    const uint8_t* dig = (const uint8_t*)(area + 9 * RTAB_ENTRY_WORDS);
    P256Pt R;
    const uint32_t use = W.kt_use[p];
    if (use) {   // u1 * G + u2 * pk with u2 * pk through the key's table: 33 gathered entries instead of the doubling chain
        uint32_t kw[8];
        words_from_limbs<8>(kw, fe_from_mont(soa_ld<ModN, 1>(W.u2m, p).as<2>()).l);
        R = p256_ktab_mul_acc(ld_rtab(area + 8 * RTAB_ENTRY_WORDS), W.ktab + (size_t)W.kt_key[p] * KTAB_KEY_WORDS, kw, use == 2);
    } else {
        P256Pt acc = p256_identity();
#pragma unroll 1
        for (int w = FRONT_NW - 1; w >= 0; w--) {
#pragma unroll 1
            for (int i = 0; i < 4; i++) acc = p256_dbl(acc);
            uint32_t db = dig[w], d = db & 15;
            P256Pt e = ld_rtab(area + (d ? d - 1 : 0) * RTAB_ENTRY_WORDS);
            e.y = fe_select((db & 0x80u) != 0, fq8_neg(e.y), e.y);
            P256Pt s = p256_add(acc, e);
            acc = p256_select(d != 0, s, acc);
        }
        R = p256_add(ld_rtab(area + 8 * RTAB_ENTRY_WORDS), acc);
    }
    // R affine (output + base of the per-proof table).  R = identity makes every T_i the identity: exp.ts:151.
    Fq2 rz = fe_reduce(R.z);
    if (fe_is_zero(rz) && (W.st[p] == ZK_OK || W.st[p] == ZK_E_ARG)) W.st[p] = ZK_E_T_INF;
    Fq2 zi = fe_inv<ModQ>(rz);
    Fq2 rx = R.x * zi, ry = R.y * zi;
    soa_st(W.Rxm, p, rx), soa_st(W.Rym, p, ry);
    soa_st(W.Rx, p, fe_from_mont(rx)), soa_st(W.Ry, p, fe_from_mont(ry));
}
void launch_front_co(hipStream_t s, const DevParams& P, const Workspace& W, uint32_t count);
void launch_front(hipStream_t s, const DevParams& P, const Workspace& W, const ChunkIn& in) {
    hipLaunchKernelGGL(k_front, dim3((in.count + 63) / 64), dim3(64), 0, s, P, W, in);
    const uint32_t wide = W.ktab && in.count <= ZK_WIDE_MAX_UNITS / 8 ? 1u : 0u;   // a small chunk: eight lanes per proof on the key-table path
    if (wide && (uint64_t)in.count * 8 <= ZK_COOP_MAX_CHAINS && !zk_one_lane_chains()) launch_front_co(s, P, W, in.count);   // below: the same sums on cooperating waves
    else if (wide) hipLaunchKernelGGL(k_front_wide, dim3((in.count * 8 + 255) / 256), dim3(256), 0, s, P, W, in.count);
    hipLaunchKernelGGL(k_front_table, dim3((in.count + 63) / 64), dim3(64), 0, s, P, W, in.count, wide);
    hipLaunchKernelGGL(k_front_walk, dim3((in.count + 63) / 64), dim3(64), 0, s, W, in.count, wide);
}

// ---------------------------------------------------------------- per-proof table of R (layout and use: rtab.h)
__global__ void __launch_bounds__(64) k_rtab_base(Workspace W, uint32_t count, uint32_t bits, const uint8_t* __restrict__ skip) {
    uint32_t p = gtid();
    if (p >= count || (skip && skip[p])) return;
    P256Aff r;
    r.x = soa_ld<ModQ, 2>(W.Rxm, p), r.y = soa_ld<ModQ, 2>(W.Rym, p);
    P256Pt b0 = p256_from_affine(r);
    if (W.st[p] == ZK_E_T_INF) b0 = p256_identity();
    // 2^(bits w) R for every window: one chain of 256 doublings per proof -- the longest chain a single verification waits for -- in Jacobian
    // coordinates (8 products per doubling instead of the complete formula's 13; R has odd prime order or is the identity: curve.h, p256_jdbl).
    // k_rtab_fill converts the bases to the homogeneous form the additions take.
    P256Jac b = p256_jac_from(b0);
    const uint32_t nwin = rtab_nwin(bits);
#pragma unroll 1
    for (uint32_t w = 0; w < nwin; w++) {
        const uint32_t e = p * nwin + w;
        soa_st(W.rbase.x, e, b.x), soa_st(W.rbase.y, e, b.y), soa_st(W.rbase.z, e, b.z);
#pragma unroll 1
        for (uint32_t i = 0; i < bits; i++) b = p256_jdbl(b);
    }
}
__global__ void __launch_bounds__(256) k_rtab_fill(Workspace W, uint32_t count, uint32_t bits, const uint8_t* __restrict__ skip) {
    uint32_t t = gtid();
    const uint32_t nwin = rtab_nwin(bits), ent = rtab_entries(bits);
    if (t >= count * nwin || (skip && skip[t / nwin])) return;
    P256Jac bj;
    bj.x = soa_ld<ModQ, 34>(W.rbase.x, t), bj.y = soa_ld<ModQ, 34>(W.rbase.y, t), bj.z = soa_ld<ModQ, 10>(W.rbase.z, t);
    P256Pt b = p256_from_jac(bj);
    uint32_t* e = W.rtab + (size_t)(t / nwin) * rtab_words(bits) + (size_t)(t % nwin) * ent * RTAB_ENTRY_WORDS;
    st_rtab(e, p256_identity());
    st_rtab(e + RTAB_ENTRY_WORDS, b);
    P256Pt acc = b;
#pragma unroll 1
    for (uint32_t d = 2; d < ent; d++) {
        acc = p256_add(acc, b);
        st_rtab(e + d * RTAB_ENTRY_WORDS, acc);
    }
}
void launch_rtab(hipStream_t s, const Workspace& W, uint32_t count, uint32_t bits, const uint8_t* skip) {
    if (count <= ZK_COOP_MAX_CHAINS / 8 && !zk_one_lane_chains()) launch_rtab_base_co(s, W, count, bits, skip);   // few proofs: the 256 doublings on a cooperating wave per proof (k_coop.hip)
    else hipLaunchKernelGGL(k_rtab_base, dim3((count + 63) / 64), dim3(64), 0, s, W, count, bits, skip);
    hipLaunchKernelGGL(k_rtab_fill, dim3((count * rtab_nwin(bits) + 255) / 256), dim3(256), 0, s, W, count, bits, skip);
}

// ---------------------------------------------------------------- Exp commit phase (exp.ts:144-149) and comS1
// item j < sec of proof p:  T = alpha_j * R,  A = T + r_j * h_NIST     (draws 3+4j, 3+4j+1, both mod n)
// item j = sec          :  comS1 = s1 * R + r0 * h_NIST                 (zkpAttestList.ts:138, draw 0)
// Proofs with a per-proof table of R: k_exp_commit.  Proofs on the key-table path (W.kt_use != 0): k_exp_commit_kt, alpha * R =
// (alpha u1) * G + (alpha u2) * pk with R = u1 G + u2 pk (zkpAttestList.ts:119-131), both through tables.  Two kernels so that neither
// carries the other's registers.
ZK_DEV void exp_scalars(const Workspace& W, uint32_t p, uint32_t j, uint32_t aw[8], uint32_t bw[8], bool want_a, bool want_b) {
    if (want_a) {
        Fe<ModN, 1> a = j < W.sec ? rng_draw<ModN>(W.rng, p, 3 + 4 * j) : soa_ld<ModN, 1>(W.s1, p);
        words_from_limbs<8>(aw, a.l);
    }
    if (want_b) {
        Fe<ModN, 1> b = rng_draw<ModN>(W.rng, p, j < W.sec ? 3 + 4 * j + 1 : 0);
        words_from_limbs<8>(bw, b.l);
    }
}
__global__ void __launch_bounds__(256) k_exp_commit(DevParams P, Workspace W, uint32_t count) {
    uint32_t t = gtid();
    uint32_t per = W.sec + 1;
    if (t >= count * per) return;
    uint32_t p = t / per, j = t % per;
    if (W.kt_use[p]) return;
    uint32_t aw[8], bw[8];
    exp_scalars(W, p, j, aw, bw, true, true);
    P256Pt T = p256_rtab_mul(W.rtab + (size_t)p * rtab_words(RTAB_PROVE_BITS), aw, RTAB_PROVE_BITS);
    P256Pt U = p256_fixed_mul(P.pfix_H, bw);
    P256Pt A = p256_add(T, U);
    st_proj(W.Tproj, t, T);
    st_proj(W.Aproj, t, A);
}
__global__ void __launch_bounds__(256) k_exp_commit_kt(DevParams P, Workspace W, uint32_t count) {
    uint32_t t = gtid();
    uint32_t per = W.sec + 1;
    if (t >= count * per) return;
    uint32_t p = t / per, j = t % per;
    const uint32_t use = W.kt_use[p];
    if (!use) return;
    uint32_t aw[8], bw[8], gw[8], kw[8];
    exp_scalars(W, p, j, aw, bw, true, true);
    Fe<ModN, 1> al;
    limbs_from_words<8>(al.l, aw);
    words_from_limbs<8>(gw, fe_canon(al.as<2>() * soa_ld<ModN, 1>(W.u1m, p).as<2>()).l);   // plain x Montgomery = plain
    words_from_limbs<8>(kw, fe_canon(al.as<2>() * soa_ld<ModN, 1>(W.u2m, p).as<2>()).l);
    P256Pt T = p256_ktab_mul_acc(p256_fixed_mul(P.pfix_G, gw), W.ktab + (size_t)W.kt_key[p] * KTAB_KEY_WORDS, kw, use == 2);
    st_proj(W.Tproj, t, T);
    st_proj(W.Aproj, t, p256_fixed_mul_acc(T, P.pfix_H, bw));   // A = T + r * h, the comb's additions straight onto T
}
// The same for a small chunk, FOUR lanes per (proof, repetition): each takes a quarter of the windows of G's comb, of the key's table and of h's comb
// (4 + 9 + 4 gathered additions in a row instead of 13 + 33 + 13), the partial sums meet through the wave's cross-lane moves.  Same group elements,
// hence the same affine coordinates and the same bytes.
__global__ void __launch_bounds__(256) k_exp_commit_kt_wide(DevParams P, Workspace W, uint32_t count) {
    const uint32_t tt = gtid(), per = W.sec + 1;
    const bool live = tt < count * per * 4;
    const uint32_t t = live ? tt >> 2 : count * per - 1, part = tt & 3;   // dead lanes of the last wave mirror the last sum
    const uint32_t p = t / per, j = t % per;
    const uint32_t use = W.kt_use[p];
    if (!use) return;   // (the four lanes of a sum leave together)
    uint32_t aw[8], bw[8], gw[8], kw[8];
    exp_scalars(W, p, j, aw, bw, true, true);
    Fe<ModN, 1> al;
    limbs_from_words<8>(al.l, aw);
    words_from_limbs<8>(gw, fe_canon(al.as<2>() * soa_ld<ModN, 1>(W.u1m, p).as<2>()).l);   // plain x Montgomery = plain
    words_from_limbs<8>(kw, fe_canon(al.as<2>() * soa_ld<ModN, 1>(W.u2m, p).as<2>()).l);
    constexpr uint32_t gper = (PFIX_NWIN + 3) / 4, kper = (KTAB_NWIN + 3) / 4;
    P256Pt T = p256_fixed_mul_range(p256_identity(), P.pfix_G, gw, part * gper, gper);
    T = p256_ktab_mul_range(T, W.ktab + (size_t)W.kt_key[p] * KTAB_KEY_WORDS, kw, use == 2, part * kper, kper);
    P256Pt U = p256_fixed_mul_range(p256_identity(), P.pfix_H, bw, part * gper, gper);
    T = p256_quad_sum(T), U = p256_quad_sum(U);
    if (!live || part) return;
    st_proj(W.Tproj, t, T);
    st_proj(W.Aproj, t, p256_add(T, U));
}
// ---------------------------------------------------------------- a call of a few proofs: the table sums on cooperating waves (coop.h)
// One sum = one workgroup of four waves: wave q adds the entries of a quarter of the comb's windows (and of the key table's) at 1.2 us an addition instead of
// one lane's 5.4-8, the four partial sums meet in LDS.  Same group elements as the one-lane kernels, hence the same affine coordinates and the same bytes
// (tests/test_gpu_prove.py: one-lane against cooperative chains, byte for byte).  ZK_UNIFORM_CF: a zero digit's addition is computed and discarded here too.
ZK_DEV CoP256 co_add_if(bool cond, const CoP256& acc, const CoFe<ModQ, 8>& ent, const CoU32& mj) {
    CoP256 e;
    e.v = ent;
#if ZK_UNIFORM_CF
    const CoP256 s = co_p256_add(acc, e, mj);
    CoP256 r;
    r.v = co_pick((uint32_t)cond, s.v, acc.v);
    return r;
#else
    return cond ? co_p256_add(acc, e, mj) : acc;
#endif
}
// acc + (windows [w0, w0 + per) of k) * B through B's comb (rtab.h: p256_fixed_mul_range)
ZK_DEV CoP256 co_fixed_mul_range(CoP256 acc, const uint32_t* __restrict__ tab, uint32_t kw[8], uint32_t w0, uint32_t per, const CoU32& mj) {
#pragma unroll 1
    for (uint32_t w = 0; w < w0; w++) shr256<PFIX_WIN_BITS>(kw);
#pragma unroll 1
    for (uint32_t w = w0; w < w0 + per && w < PFIX_NWIN; w++) {
        const uint32_t d = kw[0] & (PFIX_WIN_SIZE - 1);
        shr256<PFIX_WIN_BITS>(kw);
        acc = co_add_if(d != 0, acc, co_load_pfix(tab + (size_t)PFIX_ENTRY_WORDS * (w * PFIX_WIN_SIZE + (d ? d : 1))), mj);
    }
    return acc;
}
// ... and through a ring key's table (ktab.h: p256_ktab_mul_range)
ZK_DEV CoP256 co_ktab_mul_range(CoP256 acc, const uint32_t* __restrict__ kt, const uint32_t kw[8], bool neg, uint32_t w0, uint32_t per, const CoU32& mj) {
    KeyDigits kd;
    kd.init();
#pragma unroll
    for (int i = 0; i < 8; i++) kd.w[i] = kw[i];
    uint32_t d;
    bool dn;
#pragma unroll 1
    for (uint32_t w = 0; w < w0; w++) kd.next(d, dn);
#pragma unroll 1
    for (uint32_t w = w0; w < w0 + per && w < KTAB_NWIN; w++) {
        kd.next(d, dn);
        acc = co_add_if(d != 0, acc, co_load_ktab(kt + ((size_t)w * KTAB_ENT + (d ? d - 1 : 0)) * KTAB_ENTRY_WORDS, neg != dn), mj);
    }
    return acc;
}
// the sum of the four waves' points: waves 1..3 park theirs in LDS, wave 0 returns the total (the others return their own)
ZK_DEV CoP256 co_wg4_sum(CoP256 acc, uint32_t (*part)[64], uint32_t q, const CoU32& mj) {
    const uint32_t lane = threadIdx.x & 63u;
    __syncthreads();   // (the buffer may still be read from the sum before)
    if (q) part[q - 1][lane] = acc.v.v;
    __syncthreads();
    if (q) return acc;
#pragma unroll 1
    for (uint32_t k = 0; k < 3; k++) {
        CoP256 o;
        o.v.v = part[k][lane];
        acc = co_p256_add(acc, o, mj);
    }
    return acc;
}
// k_exp_commit_kt_wide's sums: workgroup t = (proof, repetition)
__global__ void __launch_bounds__(256) k_exp_commit_kt_co(DevParams P, Workspace W, uint32_t count) {
    __shared__ uint32_t part[3][64];
    const uint32_t t = blockIdx.x, per = W.sec + 1, q = threadIdx.x >> 6;
    const uint32_t p = t / per, j = t % per;
    const uint32_t use = W.kt_use[p];
    if (!use) return;   // (uniform for the workgroup)
    const CoU32 mj = co_limbs(ModQ::mod);
    uint32_t aw[8], bw[8], gw[8], kw[8];
    exp_scalars(W, p, j, aw, bw, true, true);
    Fe<ModN, 1> al;
    limbs_from_words<8>(al.l, aw);
    words_from_limbs<8>(gw, fe_canon(al.as<2>() * soa_ld<ModN, 1>(W.u1m, p).as<2>()).l);   // plain x Montgomery = plain
    words_from_limbs<8>(kw, fe_canon(al.as<2>() * soa_ld<ModN, 1>(W.u2m, p).as<2>()).l);
    constexpr uint32_t gper = (PFIX_NWIN + 3) / 4, kper = (KTAB_NWIN + 3) / 4;
    CoP256 T = co_fixed_mul_range(co_p256_identity(), P.pfix_G, gw, q * gper, gper, mj);
    T = co_ktab_mul_range(T, W.ktab + (size_t)W.kt_key[p] * KTAB_KEY_WORDS, kw, use == 2, q * kper, kper, mj);
    CoP256 U = co_fixed_mul_range(co_p256_identity(), P.pfix_H, bw, q * gper, gper, mj);
    T = co_wg4_sum(T, part, q, mj);
    U = co_wg4_sum(U, part, q, mj);
    if (q) return;
    co_store_soa(T.v, t, W.Tproj.x, W.Tproj.y, W.Tproj.z, Soa{nullptr, 0});
    co_store_soa(co_p256_add(T, U, mj).v, t, W.Aproj.x, W.Aproj.y, W.Aproj.z, Soa{nullptr, 0});
}
// k_front_wide's sums: workgroup = proof, eight waves -- 0..3 R = u1 G + u2 pk, 4..7 Q = z1 G; lane 0 then takes R to affine form as k_front_wide does
__global__ void __launch_bounds__(512) k_front_co(DevParams P, Workspace W, uint32_t count) {
    __shared__ uint32_t part[2][3][64];
    __shared__ uint32_t rsum[64];
    const uint32_t p = blockIdx.x, q = threadIdx.x >> 6, which = q >> 2, pt = q & 3, lane = threadIdx.x & 63u;
    const uint32_t use = W.kt_use[p];
    if (!use) return;   // (uniform for the workgroup)
    const CoU32 mj = co_limbs(ModQ::mod);
    const uint32_t* sc = front_area(W, p) + 10 * RTAB_ENTRY_WORDS;
    Fe<ModN, 1> k;
#pragma unroll
    for (int l = 0; l < NLIMB; l++) k.l[l] = sc[(which ? NLIMB : 0) + l];   // u1 (R's sum) or z1 (Q's)
    uint32_t kw[8];
    words_from_limbs<8>(kw, k.l);
    constexpr uint32_t gper = (PFIX_NWIN + 3) / 4, kper = (KTAB_NWIN + 3) / 4;
    CoP256 acc = co_fixed_mul_range(co_p256_identity(), P.pfix_G, kw, pt * gper, gper, mj);
    if (!which) {
        words_from_limbs<8>(kw, fe_from_mont(soa_ld<ModN, 1>(W.u2m, p).as<2>()).l);
        acc = co_ktab_mul_range(acc, W.ktab + (size_t)W.kt_key[p] * KTAB_KEY_WORDS, kw, use == 2, pt * kper, kper, mj);
    }
    if (pt) part[which][pt - 1][lane] = acc.v.v;
    __syncthreads();
    if (pt) return;
#pragma unroll 1
    for (uint32_t i = 0; i < 3; i++) {
        CoP256 o;
        o.v.v = part[which][i][lane];
        acc = co_p256_add(acc, o, mj);
    }
    if (which) {
        co_store_soa(acc.v, p, W.Q.x, W.Q.y, W.Q.z, Soa{nullptr, 0});
        return;
    }
    rsum[lane] = co_normalize(acc.v).v;   // wave 0 only from here on: rows X, Y, Z of R
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    if (lane) return;
    Fe<ModQ, 8> X, Y, Z;
#pragma unroll
    for (int l = 0; l < NLIMB; l++) X.l[l] = rsum[l], Y.l[l] = rsum[16 + l], Z.l[l] = rsum[32 + l];
    // R affine (output + base of the per-proof table), as in k_front_walk
    Fq2 rz = fe_reduce(Z);
    if (fe_is_zero(rz) && (W.st[p] == ZK_OK || W.st[p] == ZK_E_ARG)) W.st[p] = ZK_E_T_INF;
    Fq2 zi = fe_inv<ModQ>(rz);
    Fq2 rx = X * zi, ry = Y * zi;
    soa_st(W.Rxm, p, rx), soa_st(W.Rym, p, ry);
    soa_st(W.Rx, p, fe_from_mont(rx)), soa_st(W.Ry, p, fe_from_mont(ry));
}
void launch_front_co(hipStream_t s, const DevParams& P, const Workspace& W, uint32_t count) {
    g_coop_chains.fetch_add((uint64_t)count * 8, std::memory_order_relaxed);
    hipLaunchKernelGGL(k_front_co, dim3(count), dim3(512), 0, s, P, W, count);
}
void launch_exp_commit(hipStream_t s, const DevParams& P, const Workspace& W, uint32_t count) {
    uint32_t n = count * (W.sec + 1);
    if (W.ktab) {
        if ((uint64_t)n * 4 <= ZK_COOP_MAX_CHAINS && !zk_one_lane_chains()) {
            g_coop_chains.fetch_add((uint64_t)n * 4, std::memory_order_relaxed);
            hipLaunchKernelGGL(k_exp_commit_kt_co, dim3(n), dim3(256), 0, s, P, W, count);
        } else if (n <= ZK_WIDE_MAX_UNITS) hipLaunchKernelGGL(k_exp_commit_kt_wide, dim3((n * 4 + 255) / 256), dim3(256), 0, s, P, W, count);
        else hipLaunchKernelGGL(k_exp_commit_kt, dim3((n + 255) / 256), dim3(256), 0, s, P, W, count);
    }
    hipLaunchKernelGGL(k_exp_commit, dim3((n + 255) / 256), dim3(256), 0, s, P, W, count);
}

// ---------------------------------------------------------------- batch normalisation (weier.ts:231-243)
// Z = 0 (identity) sets st[owner] = err_code (if no earlier error) and yields (0, 0).
#ifndef ZK_NORM_PIPELINE
#define ZK_NORM_PIPELINE 0   // 1: loads of the next element issued before the products of the current one; measured, no gain: see k_tom_normalize (k_tom.hip)
#endif
__global__ void __launch_bounds__(256) k_p256_normalize(Soa3 proj, uint32_t count, uint32_t nthreads, uint32_t per, Soa ox, Soa oy,
                                                        int32_t* st, uint32_t per_proof, int32_t err_code, const uint32_t* owner) {
    __shared__ uint32_t lds[2 * 256 * NLIMB];
    uint32_t t = gtid();   // threads beyond nthreads own no element but take part in the workgroup's inversion
    Fq2 acc = fe_one_mont<ModQ>().as<2>();
    uint32_t mine = 0;
    if (t < nthreads && t < count) mine = std::min<uint32_t>(per, (count - t + nthreads - 1) / nthreads);
    {
#if ZK_NORM_PIPELINE
        Fe<ModQ, 8> zn = fe_one_mont<ModQ>().as<8>();
        if (mine) zn = soa_ld<ModQ, 8>(proj.z, t);
#endif
        for (uint32_t j = 0; j < mine; j++) {
            uint32_t e = t + j * nthreads;
#if ZK_NORM_PIPELINE
            Fq2 z = fe_reduce(zn);
            if (j + 1 < mine) zn = soa_ld<ModQ, 8>(proj.z, e + nthreads);
#else
            Fq2 z = fe_reduce(soa_ld<ModQ, 8>(proj.z, e));
#endif
            bool zero = fe_is_zero(z);
            if (zero) {
                uint32_t o = owner ? owner[e] : e / per_proof;
                // The prover's T list holds s1 * R as entry `sec` of a proof (k_exp_commit: comS1 = s1 R + r0 h).  The reference never takes its affine
                // form (commit() adds the two products, weier.ts), so s1 = 0 -- r = 0 mod n, invMod(0) = 0 -- is no 'T[i] is at infinity' there.
                const bool counts = !(err_code == ZK_ST_T_INF_LATE && e % per_proof == per_proof - 1);
                if (err_code && counts && atomicCAS(&st[o], ZK_OK, err_code) == ZK_E_ARG) atomicCAS(&st[o], ZK_E_ARG, err_code);   // `which` out of range is found last in the reference
                z = fe_one_mont<ModQ>().as<2>();
            }
            soa_st(ox, e, acc);
            soa_st(oy, e, z);  // sanitised z
            acc = acc * z;
        }
    }
    // running inverse in the plain domain (see k_tom_normalize): x and y come out plain, no from-Montgomery products
    Fe<ModQ, 1> one = fe_zero<ModQ>();
    one.l[0] = 1;
    Fq2 inv = block_inverse<ModQ>(acc, lds) * one;
    if (!mine) return;
#if ZK_NORM_PIPELINE
    uint32_t en = t + (mine - 1) * nthreads;
    Fq2 zn = soa_ld<ModQ, 2>(oy, en), pn = soa_ld<ModQ, 2>(ox, en);
    Fe<ModQ, 8> xn = soa_ld<ModQ, 8>(proj.x, en), yn = soa_ld<ModQ, 8>(proj.y, en);
    for (int j = (int)mine - 1; j >= 0; j--) {
        const uint32_t e = en;
        const Fq2 z = zn, pre = pn;
        const Fe<ModQ, 8> px = xn, py = yn;
        if (j > 0) {
            en = e - nthreads;
            zn = soa_ld<ModQ, 2>(oy, en), pn = soa_ld<ModQ, 2>(ox, en), xn = soa_ld<ModQ, 8>(proj.x, en), yn = soa_ld<ModQ, 8>(proj.y, en);
        }
        Fq2 zi = inv * pre;
        inv = inv * z;
        Fq2 x = px * zi;
        Fq2 y = py * zi;
        soa_st(ox, e, fe_canon(x));
        soa_st(oy, e, fe_canon(y));
    }
#else
    for (int j = (int)mine - 1; j >= 0; j--) {
        uint32_t e = t + (uint32_t)j * nthreads;
        Fq2 z = soa_ld<ModQ, 2>(oy, e);
        Fq2 zi = inv * soa_ld<ModQ, 2>(ox, e);
        inv = inv * z;
        Fq2 x = soa_ld<ModQ, 8>(proj.x, e) * zi;
        Fq2 y = soa_ld<ModQ, 8>(proj.y, e) * zi;
        soa_st(ox, e, fe_canon(x));
        soa_st(oy, e, fe_canon(y));
    }
#endif
}
// one thread per point with its own divsteps inversion: the small launches (k_tom.hip: k_tom_normalize_each has the reasoning); same statuses, same canonical values
#ifndef ZK_NORM_EACH_MAX
#define ZK_NORM_EACH_MAX 16384u
#endif
__global__ void __launch_bounds__(256) k_p256_normalize_each(Soa3 proj, uint32_t count, Soa ox, Soa oy, int32_t* st, uint32_t per_proof, int32_t err_code, const uint32_t* owner) {
    const uint32_t e = gtid();
    if (e >= count) return;
    Fq2 z = fe_reduce(soa_ld<ModQ, 8>(proj.z, e));
    if (fe_is_zero(z)) {   // the identity: status as in k_p256_normalize, coordinates (0, 0) (x = X / 1, y = Y / 1 with X = 0 ... the batch form yields the same: Z sanitised to 1)
        const uint32_t o = owner ? owner[e] : e / per_proof;
        const bool counts = !(err_code == ZK_ST_T_INF_LATE && e % per_proof == per_proof - 1);
        if (err_code && counts && atomicCAS(&st[o], ZK_OK, err_code) == ZK_E_ARG) atomicCAS(&st[o], ZK_E_ARG, err_code);
        z = fe_one_mont<ModQ>().as<2>();
    }
    Fe<ModQ, 1> one = fe_zero<ModQ>();
    one.l[0] = 1;
    const Fq2 zi = fe_inv<ModQ, true>(z) * one;   // (every lane its own point: field.h, LOCKSTEP)
    const Fq2 x = soa_ld<ModQ, 8>(proj.x, e) * zi, y = soa_ld<ModQ, 8>(proj.y, e) * zi;
    soa_st(ox, e, fe_canon(x));
    soa_st(oy, e, fe_canon(y));
}
void launch_p256_normalize(hipStream_t s, const Soa3& proj, uint32_t count, const Soa& ox, const Soa& oy, int32_t* st, uint32_t per_proof,
                           int32_t err_code, const uint32_t* owner) {
    if (!count) return;
    if (count <= ZK_NORM_EACH_MAX && !zk_one_lane_chains()) {
        hipLaunchKernelGGL(k_p256_normalize_each, dim3((count + 255) / 256), dim3(256), 0, s, proj, count, ox, oy, st, per_proof, err_code, owner);
        return;
    }
    uint32_t per = count / ZK_NORM_MIN_THREADS;
    if (per < 4) per = 4;
    if (per > ZK_NORM_PER_MAX) per = ZK_NORM_PER_MAX;
    uint32_t nthreads = (count + per - 1) / per;
    hipLaunchKernelGGL(k_p256_normalize, dim3((nthreads + 255) / 256), dim3(256), 0, s, proj, count, nthreads, per, ox, oy, st, per_proof, err_code, owner);
}

// ---------------------------------------------------------------- T1 = (alpha_i - s1) * R + Q  (exp.ts:186-191)
// (r = 0 mod n is the one input for which the identity below fails -- invMod(0) = 0 makes s1 = z1 = 0, T1 = T_i, and provePointAdd throws
// "Points don't add up!" at the first zero-bit repetition, pointAdd.ts:104-106: k_scan gives such a proof ZK_E_POINTS_DONT_ADD and no items.)
// alpha_i * R is T_i, already in Tproj (k_exp_commit), and s1 * R = Q + pk identically (s1 = s/r, R = (z/s) G + (r/s) pk,
// Q = (z/r) G: the relation the whole proof is about), so T1 = T_i - pk: one mixed addition instead of a second
// 256-bit scalar multiplication per zero-bit repetition.  Same group element, hence the same affine bytes; T1 = identity
// (T_i = pk) is still caught by the normaliser ('T1 is at infinity', exp.ts:193).
__global__ void __launch_bounds__(256) k_t1(Workspace W, uint32_t items) {
    uint32_t it = gtid();
    if (it >= items) return;
    uint32_t p = W.item_proof[it], i = W.item_rep[it];
    P256Aff npk;
    npk.x = soa_ld<ModQ, 2>(W.pkxm, p);
    npk.y = fe_reduce(fe_neg(soa_ld<ModQ, 2>(W.pkym, p)));
    st_proj(W.T1proj, it, p256_add_mixed(ld_proj(W.Tproj, p * (W.sec + 1) + i), npk));
}
void launch_t1(hipStream_t s, const Workspace& W, uint32_t items) {
    if (!items) return;
    hipLaunchKernelGGL(k_t1, dim3((items + 255) / 256), dim3(256), 0, s, W, items);
}

// ---------------------------------------------------------------- unit-test hook: k*G / k*h_NIST
__global__ void __launch_bounds__(64) k_test_pfix(const uint32_t* tab, uint64_t count, const uint8_t* k_be, uint8_t* out) {
    uint32_t t = gtid();
    if (t >= count) return;
    uint32_t kw[8];
    load_be32(k_be + 32 * (size_t)t, kw);
    Fe<ModN, 1> k = fe_from_words256_reduce<ModN>(kw);
    words_from_limbs<8>(kw, k.l);
    P256Pt r = p256_fixed_mul(tab, kw);
    Fq2 z = fe_reduce(r.z);
    Fq2 zi = fe_inv<ModQ>(z);
    uint32_t xw[8], yw[8];
    words_from_limbs<8>(xw, fe_from_mont(r.x * zi).l);
    words_from_limbs<8>(yw, fe_from_mont(r.y * zi).l);
    if (fe_is_zero(z)) {
        for (int i = 0; i < 8; i++) xw[i] = 0, yw[i] = 0;
    }
    store_be<8>(out + 64 * (size_t)t, xw);
    store_be<8>(out + 64 * (size_t)t + 32, yw);
}
void launch_test_pfix(hipStream_t s, const uint32_t* tab, uint64_t count, const uint8_t* k_be, uint8_t* out) {
    hipLaunchKernelGGL(k_test_pfix, dim3((count + 63) / 64), dim3(64), 0, s, tab, count, k_be, out);
}
