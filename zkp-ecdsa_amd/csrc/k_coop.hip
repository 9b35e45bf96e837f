// The verifier's and the table builder's DEPENDENT CHAINS on cooperating waves (coop.h: one field element per 16-lane row, four per wave) -- used where a call
// holds so few chains that their LENGTH is what the caller waits for: one proof per call is the reference's only call shape (src/zkpAttestList.ts:147-184,
// timed by bench/zkpAttestList.bench.ts:56-62).  A cooperating wave needs ~9x the issue slots of a lane for the same chain, so launches with tens of
// thousands of chains keep the one-lane kernels (k_verify.hip, k_p256.hip); the launch wrappers choose by the number of chains.
//   k_v_straus_co      : k_v_straus, one wave per (slot / group, part): windows top down, 4 doublings + the part's table additions        (group.ts:133-152)
//   k_v_p256_straus_co : k_v_p256_straus, one wave per partial sum of a proof's P-256 relation                                           (exp.ts:267-311)
//   k_rtab_base_co     : k_rtab_base, the 256 doublings of a proof's table of R, Jacobian with Z^2 kept beside Z (nine products in three passes)
// Same values as the one-lane kernels: the same formulas modulo the field prime; every consumer canonicalises or compares projectively.
#include "coop_dev.h"
#include "rtab.h"

std::atomic<uint64_t> g_coop_chains{0};
#define VW_ENT 8
#define VW_NW256 65
#define VW_NW128 33
#define VP_NW 33
#define VP_NW_CL 35

__global__ void __launch_bounds__(64) k_v_straus_co(VTerms L, uint32_t ngroups, uint32_t ng_stride, uint32_t n256, uint32_t n128, Soa4 out, const uint32_t* __restrict__ perm,
                                                    const uint32_t* __restrict__ cnt, uint32_t tsplit, uint32_t ostride, uint32_t ystride) {
    const uint32_t lane = blockIdx.x;   // the lane of k_v_straus this wave stands for
    const uint32_t yo = blockIdx.y * ystride;
    const uint32_t gi = lane / tsplit, part = lane % tsplit;
    const uint32_t g = perm ? perm[gi] : gi;
    const bool full = !perm || gi < cnt[0];
    const uint32_t nt = n256 + n128;
    const uint32_t klo = full ? part : nt - 2, kstep = full ? tsplit : 1;
    const bool idle = !full && part != 0;
    const CoU32 mj = co_limbs(ModT::mod);
    CoTom acc = co_tom_identity();
#pragma unroll 1
    for (int w = idle ? -1 : (full && n256 > part ? VW_NW256 - 1 : VW_NW128 - 1); w >= 0; w--) {
        const uint32_t kmax = w >= VW_NW128 ? n256 : nt;
        // the first term's digit and entry are on their way while the wave doubles
        uint32_t idx = klo * ng_stride + g + yo;
        uint32_t db = klo < kmax ? L.dig[(size_t)w * L.cap + idx] : 0u;
        CoFe<ModT, 2> ent = co_load_aos<ModT, 2, 4>(L.tab + ((size_t)idx * VW_ENT + ((db & 15) ? (db & 15) - 1 : 0)) * 36);
#pragma unroll 1
        for (int i = 0; i < 4; i++) acc = co_tom_dbl(acc, mj);
#pragma unroll 1
        for (uint32_t k = klo; k < kmax; k += kstep) {
            if (k != klo) {
                idx = k * ng_stride + g + yo;
                db = L.dig[(size_t)w * L.cap + idx];
                ent = co_load_aos<ModT, 2, 4>(L.tab + ((size_t)idx * VW_ENT + ((db & 15) ? (db & 15) - 1 : 0)) * 36);
            }
            if (db & 15) acc = co_tom_add_tab(acc, ent, (db & 0x80u) != 0, mj);   // wave-uniform: one term at a time
        }
    }
    co_store_soa(acc.v, g * ostride + part + yo, out.x, out.y, out.t, out.z);
}
void launch_v_straus_co(hipStream_t s, const VTerms& L, uint32_t ngroups, uint32_t ng_stride, uint32_t n256, uint32_t n128, const Soa4& out, const uint32_t* perm,
                        const uint32_t* cnt, uint32_t tsplit, uint32_t ostride, uint32_t ny, uint32_t ystride) {
    g_coop_chains.fetch_add((uint64_t)ngroups * tsplit * ny, std::memory_order_relaxed);
    hipLaunchKernelGGL(k_v_straus_co, dim3(ngroups * tsplit, ny), dim3(64), 0, s, L, ngroups, ng_stride, n256, n128, out, perm, cnt, tsplit, ostride, ystride);
}

// wave t = (p, q), q < parts: the window walk over A term p * VK + q (per = 1: small batches only); q == parts - 1: SL * Clambda
__global__ void __launch_bounds__(64) k_v_p256_straus_co(VWork V, uint32_t count) {
    const uint32_t t = blockIdx.x, parts = VK + 1;
    const uint32_t p = t / parts, q = t % parts;
    const bool is_cl = q == parts - 1;
    const uint32_t cap = is_cl ? V.C : V.C * VK;
    const uint32_t idx = is_cl ? p : p * VK + q;
    const uint8_t* dig = is_cl ? V.cl_dig : V.pa_dig;
    const uint32_t* tab = is_cl ? V.cl_tab : V.pa_tab;
    const CoU32 mj = co_limbs(ModQ::mod);
    const CoU32 s8 = co_sub_const<ModQ, 8>();
    CoP256 acc = co_p256_identity();
#pragma unroll 1
    for (int w = (is_cl ? VP_NW_CL : VP_NW) - 1; w >= 0; w--) {
        const uint32_t db = dig[(size_t)w * cap + idx], d = db & 15;
        CoP256 e;
        e.v = co_load_aos<ModQ, 8, 3>(tab + ((size_t)idx * 8 + (d ? d - 1 : 0)) * RTAB_ENTRY_WORDS);
#pragma unroll 1
        for (int i = 0; i < 4; i++) acc = co_p256_dbl(acc, mj);
        if (d) {
            if ((db & 0x80u) && co_row_index() == 1) e.v.v = co_carry(s8 - e.v.v);   // -Y = 8 q - Y <= 8 q (rtab.h: fq8_neg)
            acc = co_p256_add(acc, e, mj);
        }
    }
    co_store_soa(acc.v, t, V.pacc.x, V.pacc.y, V.pacc.z, Soa{nullptr, 0});
}
void launch_v_p256_straus_co(hipStream_t s, const VWork& V, uint32_t count) {
    g_coop_chains.fetch_add((uint64_t)count * (VK + 1), std::memory_order_relaxed);
    if (count) hipLaunchKernelGGL(k_v_p256_straus_co, dim3(count * (VK + 1)), dim3(64), 0, s, V, count);
}

// T = alpha R / T1 = z R + Q of a sampled repetition (exp.ts:267,299,311; k_verify.hip: k_v_exp_points) for a call of a few proofs: one workgroup per slot, its four
// waves walk a quarter of the 65 windows of R's table each -- the table holds every 2^(4w) R, so there are no doublings: 17 additions in a row at 1.2 us instead
// of one lane's 5.4 -- and meet in LDS (three more, then Q).  The first slot whose point is the identity lowers V.exp_jz[p] (k_v_exp_status).
__global__ void __launch_bounds__(256) k_v_exp_points_co(Workspace W, VWork V, uint32_t count, const uint8_t* proofs, const uint64_t* off, uint64_t first) {
    __shared__ uint32_t part[3][64];
    const uint32_t t = blockIdx.x, p = t / VK, q = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t iv = V.idx[t], i = iv & 255, bit = iv >> 8;
    const bool good = V.st[p] == ZK_OK && !(V.okflags[p] & 8);
    const CoU32 mj = co_limbs(ModQ::mod);
    const CoU32 s8 = co_sub_const<ModQ, 8>();
    CoP256 acc = co_p256_identity();
    if (good) {   // uniform for the workgroup
        const uint8_t* rep = proofs + off[first + p] + rep_offset(V.hbits + 4 * p, i);
        uint32_t kw[8];
        {
            uint32_t w[8];
            load_be32(rep + 208, w);
            words_from_limbs<8>(kw, fe_from_words256_reduce<ModN>(w).l);   // the scalar as verifyExp reads it (k_verify.hip: ld_scalar_n)
        }
        const uint32_t* tab = W.rtab + (size_t)p * rtab_words(RTAB_VERIFY_BITS);
        constexpr uint32_t bits = RTAB_VERIFY_BITS, nwin = (257 + bits - 1) / bits, ent = (1u << (bits - 1)) + 1, half = 1u << (bits - 1), mask = (1u << bits) - 1, per = (nwin + 3) / 4;   // engine.h: rtab_nwin, rtab_entries
        uint32_t carry = 0;
#pragma unroll 1
        for (uint32_t w = 0; w < q * per; w++) {   // signed recoding of the windows below this wave's (rtab.h: p256_rtab_mul_range)
            const uint32_t d = (kw[0] & mask) + carry;
            shr256_var(kw, bits);
            carry = d > half ? 1 : 0;
        }
#pragma unroll 1
        for (uint32_t w = q * per; w < (q + 1) * per && w < nwin; w++) {
            uint32_t d = (kw[0] & mask) + carry;
            shr256_var(kw, bits);
            const bool neg = d > half;
            carry = neg ? 1 : 0;
            if (neg) d = (1u << bits) - d;
            if (!d) continue;   // entry 0 is the identity
            CoP256 e;
            e.v = co_load_aos<ModQ, 8, 3>(tab + (size_t)RTAB_ENTRY_WORDS * (w * ent + d));
            if (neg && co_row_index() == 1) e.v.v = co_carry(s8 - e.v.v);   // -Y = 8 q - Y <= 8 q (rtab.h: fq8_neg)
            acc = co_p256_add(acc, e, mj);
        }
    }
    if (q) part[q - 1][lane] = acc.v.v;
    __syncthreads();
    if (q) return;
#pragma unroll 1
    for (uint32_t k = 0; k < 3; k++) {
        CoP256 o;
        o.v.v = part[k][lane];
        acc = co_p256_add(acc, o, mj);
    }
    if (good) {
        if (!bit) {
            CoP256 qq;
            qq.v = co_load_soa<ModQ, 8>(p, W.Q.x, W.Q.y, W.Q.z, Soa{nullptr, 0});
            acc = co_p256_add(acc, qq, mj);
        }
        // 'T is at infinity' / 'T1 is at infinity' (exp.ts:274,312): Z = 0 or q (a product's value is below 2 q); the first such slot counts
        const uint32_t z = co_normalize(acc.v).v;
        const uint32_t is0 = (uint32_t)(__ballot(z == 0u) >> 32) & 0xffffu, isq = (uint32_t)(__ballot(z == mj) >> 32) & 0xffffu;
        if (lane == 0 && (is0 == 0xffffu || isq == 0xffffu)) atomicMin(V.exp_jz + p, t % VK);
    } else {   // keep later kernels on defined data: G
        const uint32_t row = co_row_index();
        acc.v.v = row == 0 ? co_limbs(P256_GX_M) : row == 1 ? co_limbs(P256_GY_M) : row == 2 ? co_limbs(ModQ::one) : 0u;
    }
    co_store_soa(acc.v, t, W.Tproj.x, W.Tproj.y, W.Tproj.z, Soa{nullptr, 0});
}
void launch_v_exp_points_co(hipStream_t s, const Workspace& W, const VWork& V, uint32_t count, const uint8_t* proofs, const uint64_t* off, uint64_t first) {
    g_coop_chains.fetch_add((uint64_t)count * VK * 4, std::memory_order_relaxed);
    if (count) hipLaunchKernelGGL(k_v_exp_points_co, dim3(count * VK), dim3(256), 0, s, W, V, count, proofs, off, first);
}

// one wave per proof: W.rbase entry p * nwin + w <- 2^(bits w) R in Jacobian coordinates (k_rtab_fill takes them to the homogeneous form)
__global__ void __launch_bounds__(64) k_rtab_base_co(Workspace W, uint32_t count, uint32_t bits, const uint8_t* __restrict__ skip) {
    const uint32_t p = blockIdx.x;
    if (skip && skip[p]) return;
    const CoU32 mj = co_limbs(ModQ::mod);
    CoP256J b;
    if (W.st[p] == ZK_E_T_INF) {
        b.v.v = co_row_index() == 1 ? co_limbs(ModQ::one) : 0u;                                  // (0 : 1 : 0 : 0)
    } else {
        const uint32_t row = __lane_id() >> 4, j = __lane_id() & 15u;
        const uint32_t* srcp = row == 0 ? W.Rxm.p : W.Rym.p;
        const uint32_t srcs = row == 0 ? W.Rxm.stride : W.Rym.stride;
        const CoU32 one = co_limbs(ModQ::one);
        b.v.v = j >= NLIMB ? 0u : row < 2 ? srcp[(size_t)j * srcs + p] : one;                // (x : y : 1 : 1)
    }
    const uint32_t nwin = rtab_nwin(bits);
#pragma unroll 1
    for (uint32_t w = 0; w < nwin; w++) {
        co_store_soa(b.v, p * nwin + w, W.rbase.x, W.rbase.y, W.rbase.z, Soa{nullptr, 0});
#pragma unroll 1
        for (uint32_t i = 0; i < bits; i++) b = co_p256_jdbl(b, mj);
    }
}
void launch_rtab_base_co(hipStream_t s, const Workspace& W, uint32_t count, uint32_t bits, const uint8_t* skip) {
    g_coop_chains.fetch_add(count, std::memory_order_relaxed);
    if (count) hipLaunchKernelGGL(k_rtab_base_co, dim3(count), dim3(64), 0, s, W, count, bits, skip);
}
