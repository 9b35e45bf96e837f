// Fiat-Shamir hashing and the RNG prepass.
// hashPoints (src/curves/group.ts:221-233): SHA-256 over the concatenated affine uncompressed encodings
// (P-256: 04 || x(32) || y(32), weier.ts:244-255; Tom-256: 04 || x(33) || y(33), edwards.ts:194-203),
// challenge = first 10 bytes of the digest as a big-endian integer.
#include "engine.h"

ZK_DEV void absorb_tom(ShaStream& s, const Soa& ax, const Soa& ay, uint32_t slot) {
    uint32_t w[9];
    s.put_byte(4);
    words_from_limbs<9>(w, soa_ld<ModT, 1>(ax, slot).l);
    s.put_be<33>(w);
    words_from_limbs<9>(w, soa_ld<ModT, 1>(ay, slot).l);
    s.put_be<33>(w);
}
ZK_DEV void absorb_tom_limbs(ShaStream& s, const uint32_t* xy18) {
    uint32_t w[9];
    s.put_byte(4);
    words_from_limbs<9>(w, xy18);
    s.put_be<33>(w);
    words_from_limbs<9>(w, xy18 + 9);
    s.put_be<33>(w);
}
ZK_DEV void absorb_p256(ShaStream& s, const Soa& ax, const Soa& ay, uint32_t e) {
    uint32_t w[8];
    s.put_byte(4);
    words_from_limbs<8>(w, soa_ld<ModQ, 1>(ax, e).l);
    s.put_be<32>(w);
    words_from_limbs<8>(w, soa_ld<ModQ, 1>(ay, e).l);
    s.put_be<32>(w);
}
// digest -> 80-bit challenge as 3 little-endian words (+ a zero word)
ZK_DEV void challenge_words(const uint32_t h[8], uint32_t c[4]) {
    c[0] = (h[1] << 16) | (h[2] >> 16);
    c[1] = (h[0] << 16) | (h[1] >> 16);
    c[2] = h[0] >> 16;
    c[3] = 0;
}

// ---------------------------------------------------------------- RNG prepass
// `fill` (seed mode only): the fills are also written out as an explicit stream, [proof][block][32 bytes], and every
// later kernel of the chunk reads its draws from there (mode 1) instead of hashing seed || k again.
// The scan runs in two stages: blocks [0, 3 + 4 sec) before the Exp commit phase, and -- once the challenge has fixed the
// number z of zero bits -- the blocks the rest of the proof can reach, [3 + 4 sec, 3 + 4 sec + 40 z + 5 n + RNG_MAX_EXC).
__global__ void __launch_bounds__(256) k_rng_prepass(Workspace W, uint32_t count, uint32_t blk0, uint32_t span, uint32_t stride, uint32_t* fill, int by_zcnt) {
    uint32_t t = gtid();
    if (t >= count * span) return;
    uint32_t p = t / span, blk = blk0 + t % span;
    if (by_zcnt && blk >= 3 + 4 * W.sec + 40 * W.zcnt[p] + 5 * W.n + RNG_MAX_EXC) return;
    uint32_t w[8];
    rng_block(W.rng, p, blk, w);
    if (fill) {
        uint4* o = (uint4*)(fill + ((size_t)p * stride + blk) * 8);
        o[0] = make_uint4(bswap32(w[7]), bswap32(w[6]), bswap32(w[5]), bswap32(w[4]));
        o[1] = make_uint4(bswap32(w[3]), bswap32(w[2]), bswap32(w[1]), bswap32(w[0]));
    }
    if (w[7] != 0xffffffffu) return;
    uint32_t fl = (words_geq<8>(w, ModN::mod32) ? 1u : 0u) | (words_geq<8>(w, ModQ::mod32) ? 2u : 0u);
    if (!fl) return;
    uint32_t i = atomicAdd(&W.rng.exc_cnt[p], 1u);
    if (i < RNG_MAX_EXC) {
        W.rng.exc_idx[p * RNG_MAX_EXC + i] = blk;
        W.rng.exc_flags[p * RNG_MAX_EXC + i] = fl;
    }
}
void launch_rng_prepass(hipStream_t s, const Workspace& W, uint32_t count, uint32_t blk0, uint32_t blk1, uint32_t stride, uint32_t* fill, bool by_zcnt) {
    if (blk0 == 0) hipMemsetAsync(W.rng.exc_cnt, 0, sizeof(uint32_t) * count, s);
    uint64_t n = (uint64_t)count * (blk1 - blk0);
    if (!n) return;
    hipLaunchKernelGGL(k_rng_prepass, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, W, count, blk0, blk1 - blk0, stride, fill, by_zcnt ? 1 : 0);
}

// ---------------------------------------------------------------- Exp challenge (exp.ts:158-165)
// c = H(Px, Py, A_0, Tx_0, Ty_0, ..., A_{sec-1}, Tx_{sec-1}, Ty_{sec-1})
__global__ void __launch_bounds__(64) k_exp_challenge(Workspace W, uint32_t count) {
    __shared__ uint32_t lds[16 * 64];
    uint32_t p = gtid();
    bool live = p < count;
    if (!live) p = count - 1;  // keep the wave's structure uniform
    ShaStream s;
    s.init(lds, threadIdx.x, 64);
    uint32_t la = p * (2 + 2 * W.sec), ea = p * (W.sec + 1);
    absorb_tom(s, W.la.ax, W.la.ay, la + 0);
    absorb_tom(s, W.la.ax, W.la.ay, la + 1);
    for (uint32_t i = 0; i < W.sec; i++) {
        absorb_p256(s, W.Ax, W.Ay, ea + i);
        absorb_tom(s, W.la.ax, W.la.ay, la + 2 + 2 * i);
        absorb_tom(s, W.la.ax, W.la.ay, la + 3 + 2 * i);
    }
    uint32_t h[8], c[4];
    s.finish(h);
    challenge_words(h, c);
    if (live) {
#pragma unroll
        for (int i = 0; i < 4; i++) W.chal[4 * p + i] = c[i];
    }
}
// The same digest in three kernels (prover chunks of any size: the buffers are borrowed from list B, api.hip carve).  One lane per proof hashing 16 KB is a chain of 251
// compressions, and of a compression's ~1 650 instructions a third is the message schedule and a fifth the byte-wise absorption of 33-byte coordinates --
// neither depends on the chaining value.  k_exph_msg (one lane per point) writes the padded message, k_exph_sched (one lane per (block, proof)) expands every
// block to its 64 words W_i + K_i, k_exph_rounds (one lane per proof) runs the 64 rounds per block and nothing else: 1.33 -> 0.7 ms for one proof (round 4).
// Round 5: the schedule is stored proof-fastest ([block][uint4 i][proof]), so the 64 lanes of a rounds wave read 1 KB runs, and chunks of any size take this
// path -- a 22 016-proof chunk is 344 such waves, one per SIMD, i.e. it takes what ONE proof takes instead of 1.8 ms (k_exp_challenge stays for larger chunks).
template <int NW>
ZK_DEV void exph_put_coord(uint8_t* o, const uint32_t* w) {   // big-endian, (NW == 9 ? 33 : 32) bytes
    constexpr int NB = NW == 9 ? 33 : 32;
#pragma unroll
    for (int i = 0; i < NB; i++) o[i] = (uint8_t)(w[(NB - 1 - i) >> 2] >> (8 * ((NB - 1 - i) & 3)));
}
__global__ void __launch_bounds__(256) k_exph_msg(Workspace W, uint32_t count) {
    const uint32_t ne = 2 + 3 * W.sec + 1, t = gtid();
    if (t >= count * ne) return;
    const uint32_t p = t / ne, e = t % ne, nblk = exph_blocks(W.sec);
    uint8_t* m = W.exph_msg + (size_t)p * nblk * 64;
    const uint32_t la = p * (2 + 2 * W.sec), ea = p * (W.sec + 1);
    if (e == ne - 1) {
        exph_put_padding(m, W.sec);
        return;
    }
    const uint32_t off = exph_elem_offset(e);
    uint32_t slot;
    bool p256 = false;
    if (e < 2) slot = la + e;
    else {
        const uint32_t j = (e - 2) / 3, k = (e - 2) % 3;
        p256 = k == 0, slot = p256 ? ea + j : la + 2 + 2 * j + (k - 1);
    }
    m[off] = 4;
    if (p256) {
        uint32_t w[8];
        words_from_limbs<8>(w, soa_ld<ModQ, 1>(W.Ax, slot).l);
        exph_put_coord<8>(m + off + 1, w);
        words_from_limbs<8>(w, soa_ld<ModQ, 1>(W.Ay, slot).l);
        exph_put_coord<8>(m + off + 33, w);
    } else {
        uint32_t w[9];
        words_from_limbs<9>(w, soa_ld<ModT, 1>(W.la.ax, slot).l);
        exph_put_coord<9>(m + off + 1, w);
        words_from_limbs<9>(w, soa_ld<ModT, 1>(W.la.ay, slot).l);
        exph_put_coord<9>(m + off + 34, w);
    }
}
__global__ void __launch_bounds__(256) k_exph_sched(Workspace W, uint32_t count, uint32_t nblk) {
    const uint32_t t = gtid();
    if (t >= count * nblk) return;
    const uint32_t b = t / count, p = t % count;   // consecutive lanes: consecutive proofs, the same block
    const uint4* src = (const uint4*)(W.exph_msg + ((size_t)p * nblk + b) * 64);
    uint32_t w[64];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint4 v = src[i];
        w[4 * i] = bswap32(v.x), w[4 * i + 1] = bswap32(v.y), w[4 * i + 2] = bswap32(v.z), w[4 * i + 3] = bswap32(v.w);
    }
#pragma unroll
    for (int i = 16; i < 64; i++) {
        const uint32_t w15 = w[i - 15], w2 = w[i - 2];
        w[i] = w[i - 16] + zk_xor3(rotr32(w15, 7), rotr32(w15, 18), w15 >> 3) + w[i - 7] + zk_xor3(rotr32(w2, 17), rotr32(w2, 19), w2 >> 10);
    }
    uint4* dst = (uint4*)W.exph_wk + (size_t)b * 16 * count + p;
#pragma unroll
    for (int i = 0; i < 16; i++) dst[(size_t)i * count] = make_uint4(w[4 * i] + SHA_K[4 * i], w[4 * i + 1] + SHA_K[4 * i + 1], w[4 * i + 2] + SHA_K[4 * i + 2], w[4 * i + 3] + SHA_K[4 * i + 3]);
}
template <bool VAR>   // VAR: message p has nblk_of[p] <= nblk blocks
__global__ void __launch_bounds__(64) k_exph_rounds(Workspace W, uint32_t count, uint32_t* chal, uint32_t nblk, uint32_t ostride, const uint8_t* __restrict__ nblk_of) {
    const uint32_t p = gtid();
    if (p >= count) return;
    const uint32_t mine = VAR ? nblk_of[p] : nblk;   // VAR: every lane runs nblk blocks and keeps the chaining value after its own last one (see k_exph_rounds2)
    uint32_t keep[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const uint4* wk = (const uint4*)W.exph_wk + p;   // word group i of block b: wk[(16 b + i) * count]
    uint32_t h[8];
    sha256_iv(h);
    uint4 nx[16];
#pragma unroll
    for (int i = 0; i < 16; i++) nx[i] = wk[(size_t)i * count];
#pragma unroll 1
    for (uint32_t b = 0; b < nblk; b++) {
        uint32_t w[64];
#pragma unroll
        for (int i = 0; i < 16; i++) w[4 * i] = nx[i].x, w[4 * i + 1] = nx[i].y, w[4 * i + 2] = nx[i].z, w[4 * i + 3] = nx[i].w;
        if (b + 1 < nblk) {   // the next block's words travel while this block's rounds run
#pragma unroll
            for (int i = 0; i < 16; i++) nx[i] = wk[((size_t)16 * (b + 1) + i) * count];
        }
        uint32_t a = h[0], bb = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
        for (int i = 0; i < 64; i++) {
            const uint32_t t1 = hh + zk_xor3(rotr32(e, 6), rotr32(e, 11), rotr32(e, 25)) + zk_bfi(e, f, g) + w[i];
            const uint32_t t2 = zk_xor3(rotr32(a, 2), rotr32(a, 13), rotr32(a, 22)) + zk_maj(a, bb, c);
            hh = g, g = f, f = e, e = d + t1, d = c, c = bb, bb = a, a = t1 + t2;
        }
        h[0] += a, h[1] += bb, h[2] += c, h[3] += d, h[4] += e, h[5] += f, h[6] += g, h[7] += hh;
        if (VAR && b + 1 == mine) {
#pragma unroll
            for (int i = 0; i < 8; i++) keep[i] = h[i];
        }
    }
    if (VAR) {
#pragma unroll
        for (int i = 0; i < 8; i++) h[i] = keep[i];
    }
    uint32_t cw[4];
    challenge_words(h, cw);
    for (uint32_t i = 0; i < ostride; i++) chal[ostride * p + i] = cw[i];
}
// The rounds on a PAIR of lanes per proof.  One wave issues one vector instruction every ~4.6 cycles whatever it is, so a lone chain is as long as its instruction
// count: 14 per round in one lane.  Lane E holds (e, f, g, h), lane A holds (a, b, c, d); with per-lane rotation amounts the SAME three v_alignbit + xor3 give
// Sigma1(e) on E and Sigma0(a) on A, and Maj(a, b, c) = Ch(~(a ^ b), b, c) lets one bitop3 + one bfi give Ch on E and Maj on A.  E adds h and W_i + K_i, the
// lanes swap T1 for d through a DPP move, and each has its new first word: 11 instructions per round.  (E's schedule words are real, A reads a zero cell.)
static bool getenv_exph_one_lane() { return zk_one_lane_chains(); }   // ZKATTEST_ONE_LANE_CHAINS: the A/B switch of the cooperative kernels covers this one too
__device__ const uint4 g_exph_zero16 = {0, 0, 0, 0};
template <bool VAR>
__global__ void __launch_bounds__(64) k_exph_rounds2(Workspace W, uint32_t count, uint32_t* chal, uint32_t nblk, uint32_t ostride, const uint8_t* __restrict__ nblk_of) {
    const uint32_t t = gtid(), p0 = t >> 1;
    const bool live = p0 < count;
    const uint32_t p = live ? p0 : count - 1;   // (a dead pair mirrors the last proof: the DPP moves need both lanes of a pair)
    const bool isA = t & 1;
    // VAR: message p ends after nblk_of[p] <= nblk blocks.  Every pair still runs the launch's nblk blocks (a loop count in a scalar register keeps the loads of the
    // next block in front of this block's rounds) and keeps the chaining value it had after its own last block.
    const uint32_t mine = VAR ? nblk_of[p] : nblk;   // (both lanes of a pair read the same count)
    uint32_t k0 = 0, k1 = 0, k2 = 0, k3 = 0;
    const uint32_t m = isA ? 0xffffffffu : 0u, nm = ~m;
    const uint32_t s1 = isA ? 2 : 6, s2 = isA ? 13 : 11, s3 = isA ? 22 : 25;
    // word group i of block b: wk[(16 b + i) * count + p] on E, the zero cell on A
    const uint4* wk = isA ? &g_exph_zero16 : (const uint4*)W.exph_wk + p;
    const size_t step = isA ? 0 : count;
    uint32_t iv[8];
    sha256_iv(iv);
    uint32_t hc0 = isA ? iv[0] : iv[4], hc1 = isA ? iv[1] : iv[5], hc2 = isA ? iv[2] : iv[6], hc3 = isA ? iv[3] : iv[7];
    uint4 nx[16];
#pragma unroll
    for (int i = 0; i < 16; i++) nx[i] = wk[(size_t)i * step];
#pragma unroll 1
    for (uint32_t b = 0; b < nblk; b++) {
        uint32_t w[64];
#pragma unroll
        for (int i = 0; i < 16; i++) w[4 * i] = nx[i].x, w[4 * i + 1] = nx[i].y, w[4 * i + 2] = nx[i].z, w[4 * i + 3] = nx[i].w;
        if (b + 1 < nblk) {   // the next block's words travel while this block's rounds run
#pragma unroll
            for (int i = 0; i < 16; i++) nx[i] = wk[((size_t)16 * (b + 1) + i) * step];
        }
        uint32_t r0 = hc0, r1 = hc1, r2 = hc2, r3 = hc3;
        uint32_t pre = w[0] + (r3 & nm);                               // E: h + W_0 + K_0; A: 0
#pragma unroll
        for (int i = 0; i < 64; i++) {
            const uint32_t sg = zk_xor3(__builtin_amdgcn_alignbit(r0, r0, s1), __builtin_amdgcn_alignbit(r0, r0, s2), __builtin_amdgcn_alignbit(r0, r0, s3));
            const uint32_t x = __builtin_amdgcn_bitop3_b32(r0, r1, m, 0xd2);   // m ? ~(r0 ^ r1) : r0 -- E: e; A: ~(a ^ b)   (truth table over a = 0xf0, b = 0xcc, c = 0xaa)
            const uint32_t ch = zk_bfi(x, r1, r2);                     // E: Ch(e, f, g); A: Maj(a, b, c)
            const uint32_t z = sg + ch + pre;                          // E: T1 = h + Sigma1 + Ch + W_i + K_i; A: T2 = Sigma0 + Maj
            const uint32_t y = isA ? r3 : z;                           // what the other lane needs: E's T1, A's d
            // the next round's h + W + K (the new h is this round's g) needs nothing of this round's result: its two instructions fill the two wait states a DPP
            // read of y needs after y was written, where the compiler would otherwise put s_nops
            if (i < 63) pre = w[i + 1] + (r2 & nm);
            const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)y, 0xB1, 0xf, 0xf, true);   // quad_perm [1, 0, 3, 2]: the pair's other lane
            r3 = r2, r2 = r1, r1 = r0, r0 = z + o;                     // E: e' = d + T1; A: a' = T2 + T1
        }
        hc0 += r0, hc1 += r1, hc2 += r2, hc3 += r3;
        if (VAR && b + 1 == mine) k0 = hc0, k1 = hc1, k2 = hc2, k3 = hc3;
    }
    if (VAR) hc0 = k0, hc1 = k1, hc2 = k2, hc3 = k3;
    // the challenge is cut out of digest words 0..2: lane A has them
    if (!live || !isA) return;
    const uint32_t h[8] = {hc0, hc1, hc2, hc3, 0, 0, 0, 0};
    uint32_t cw[4];
    challenge_words(h, cw);
    for (uint32_t i = 0; i < ostride; i++) chal[ostride * p + i] = cw[i];
}
// schedule per block, rounds per message: nblk blocks of 64 bytes per message at W.exph_msg, the challenge's ostride (3 or 4) words to chal[ostride p ..]
void launch_sha_msgs(hipStream_t s, const Workspace& W, uint32_t count, uint32_t* chal, uint32_t nblk, uint32_t ostride, const uint8_t* nblk_of) {
    hipLaunchKernelGGL(k_exph_sched, dim3((count * nblk + 255) / 256), dim3(256), 0, s, W, count, nblk);
    // up to one wave per SIMD the chain's length is the cost: two lanes per proof; beyond that the lanes are, and the one-lane form has fewer of them
    const bool two = count <= 32768 && !getenv_exph_one_lane();
    const dim3 grid(((two ? 2 : 1) * count + 63) / 64);
    if (two && nblk_of) hipLaunchKernelGGL(k_exph_rounds2<true>, grid, dim3(64), 0, s, W, count, chal, nblk, ostride, nblk_of);
    else if (two) hipLaunchKernelGGL(k_exph_rounds2<false>, grid, dim3(64), 0, s, W, count, chal, nblk, ostride, nblk_of);
    else if (nblk_of) hipLaunchKernelGGL(k_exph_rounds<true>, grid, dim3(64), 0, s, W, count, chal, nblk, ostride, nblk_of);
    else hipLaunchKernelGGL(k_exph_rounds<false>, grid, dim3(64), 0, s, W, count, chal, nblk, ostride, nblk_of);
}
void launch_exph_hash(hipStream_t s, const Workspace& W, uint32_t count, uint32_t* chal) {
    launch_sha_msgs(s, W, count, chal, (2 * 67 + W.sec * (65 + 2 * 67) + 9 + 63) / 64, 4);
}
void launch_exp_challenge(hipStream_t s, const Workspace& W, uint32_t count) {
    const bool small = count <= W.exph_cap && W.exph_wk, big = !small && count <= W.exph_big_cap && W.exph_big_wk;
    if (small || big) {
        Workspace Wb = W;
        if (big) Wb.exph_msg = W.exph_big_msg, Wb.exph_wk = W.exph_big_wk;   // list B's memory: free until stage 2 of this chunk
        const uint32_t ne = 2 + 3 * W.sec + 1;
        hipLaunchKernelGGL(k_exph_msg, dim3((count * ne + 255) / 256), dim3(256), 0, s, Wb, count);
        launch_exph_hash(s, Wb, count, W.chal);
        return;
    }
    hipLaunchKernelGGL(k_exp_challenge, dim3((count + 63) / 64), dim3(64), 0, s, W, count);
}

// ---------------------------------------------------------------- PointAdd sub-proof challenges
// thread t = h * items + item, h: 0 pi8, 1 pi10, 2 pi11, 3 pi13 (mult.ts:115-116), 4 pix, 5 piy (equality.ts:69)
__global__ void __launch_bounds__(256) k_padd_hash(DevParams P, Workspace W, uint32_t items) {
    __shared__ uint32_t lds[16 * 256];
    uint32_t t = gtid();
    bool live = t < items * 6;
    if (!live) t = items * 6 - 1;
    uint32_t h = t / items, item = t % items;
    ShaStream s;
    s.init(lds, threadIdx.x, 256);
    const Soa &ax = W.lb.ax, &ay = W.lb.ay;
    if (h < 4) {
        uint32_t cx, cy, cz, first;
        if (h == 0) cx = 34, cy = 2, cz = 0xffffffffu, first = 6;       // C7, C8, C14 = g
        else if (h == 1) cx = 2, cy = 35, cz = 3, first = 12;           // C8, C9, C10
        else if (h == 2) cx = 3, cy = 3, cz = 4, first = 18;            // C10, C10, C11
        else cx = 3, cy = 36, cz = 5, first = 24;                       // C10, C12, C13
        absorb_tom(s, ax, ay, lbi(W, item, cx));
        absorb_tom(s, ax, ay, lbi(W, item, cy));
        if (cz == 0xffffffffu) absorb_tom_limbs(s, P.tom_g_aff);
        else absorb_tom(s, ax, ay, lbi(W, item, cz));
        for (uint32_t k = 0; k < 6; k++) absorb_tom(s, ax, ay, lbi(W, item, first + k));
    } else {
        uint32_t c1 = h == 4 ? 4 : 5, c2 = h == 4 ? 37 : 38, a1 = h == 4 ? 30 : 32;
        absorb_tom(s, ax, ay, lbi(W, item, c1));
        absorb_tom(s, ax, ay, lbi(W, item, c2));
        absorb_tom(s, ax, ay, lbi(W, item, a1));
        absorb_tom(s, ax, ay, lbi(W, item, a1 + 1));
    }
    uint32_t dg[8], c[4];
    s.finish(dg);
    challenge_words(dg, c);
    if (live) {
        uint32_t* o = W.padd_c + ((size_t)item * 6 + h) * 3;
        o[0] = c[0], o[1] = c[1], o[2] = c[2];
    }
}
void launch_padd_hash(hipStream_t s, const DevParams& P, const Workspace& W, uint32_t items) {
    if (!items) return;
    hipLaunchKernelGGL(k_padd_hash, dim3((items * 6 + 255) / 256), dim3(256), 0, s, P, W, items);
}

// ---------------------------------------------------------------- GK challenge x = H(cl || ca || cb || cd) (gk.ts:178-180)
__global__ void __launch_bounds__(64) k_gk_hash(Workspace W, uint32_t count, const uint8_t* __restrict__ msg) {
    __shared__ uint32_t lds[16 * 64];
    uint32_t p = gtid();
    bool live = p < count;
    if (!live) p = count - 1;
    ShaStream s;
    s.init(lds, threadIdx.x, 64);
    for (uint32_t k = 0; k < 4 * W.n; k++) absorb_tom(s, W.lc.ax, W.lc.ay, p * 4 * W.n + k);
    if (W.hardened) {   // the statement: which ring, which message, which R, which committed key (gk.ts:178 TODO)
        sha_put_gk_statement_head(s, W.ring_digest, msg + 32 * (size_t)p);
        absorb_p256(s, W.Rx, W.Ry, p);
        absorb_tom(s, W.la.ax, W.la.ay, p * (2 + 2 * W.sec));
    }
    uint32_t h[8], c[4];
    s.finish(h);
    challenge_words(h, c);
    if (live) W.gk_x[3 * p] = c[0], W.gk_x[3 * p + 1] = c[1], W.gk_x[3 * p + 2] = c[2];
}
// The same digest for a call of a few proofs through the Exp challenge's three kernels (its buffers are free in stage 2): one lane per point writes the message
// (4 n points of 67 bytes; hardened: the statement behind them), one lane per block expands the schedule, a pair of lanes per proof runs the rounds --
// 68 blocks at 1.5 us instead of 2.5, and nobody absorbs 4 KB byte by byte in one lane.
__global__ void __launch_bounds__(256) k_gk_msg(Workspace W, uint32_t count, const uint8_t* __restrict__ msg, uint32_t nblk) {
    const uint32_t ne = 4 * W.n + 1, t = gtid();
    if (t >= count * ne) return;
    const uint32_t p = t / ne, e = t % ne;
    uint8_t* m = W.exph_msg + (size_t)p * nblk * 64;
    uint32_t w[9];
    if (e < 4 * W.n) {
        uint8_t* o = m + 67 * e;
        o[0] = 4;
        words_from_limbs<9>(w, soa_ld<ModT, 1>(W.lc.ax, p * 4 * W.n + e).l);
        exph_put_coord<9>(o + 1, w);
        words_from_limbs<9>(w, soa_ld<ModT, 1>(W.lc.ay, p * 4 * W.n + e).l);
        exph_put_coord<9>(o + 34, w);
        return;
    }
    uint32_t len = 67 * 4 * W.n;
    if (W.hardened) {   // the statement: which ring, which message, which R, which committed key (sha256.h: sha_put_gk_statement_head, then R and keyXcom as points)
        const char tag[] = "ZKAttest-GK-statement-v1";
        for (int i = 0; i < 24; i++) m[len + i] = (uint8_t)tag[i];
        len += 24;
        for (int i = 0; i < 8; i++) {
            const uint32_t v = W.ring_digest[i];
            m[len + 4 * i] = (uint8_t)(v >> 24), m[len + 4 * i + 1] = (uint8_t)(v >> 16), m[len + 4 * i + 2] = (uint8_t)(v >> 8), m[len + 4 * i + 3] = (uint8_t)v;
        }
        len += 32;
        for (int i = 0; i < 32; i++) m[len + i] = msg[32 * (size_t)p + i];
        len += 32;
        m[len] = 4;
        words_from_limbs<8>(w, soa_ld<ModQ, 1>(W.Rx, p).l);
        exph_put_coord<8>(m + len + 1, w);
        words_from_limbs<8>(w, soa_ld<ModQ, 1>(W.Ry, p).l);
        exph_put_coord<8>(m + len + 33, w);
        len += 65;
        m[len] = 4;
        words_from_limbs<9>(w, soa_ld<ModT, 1>(W.la.ax, p * (2 + 2 * W.sec)).l);
        exph_put_coord<9>(m + len + 1, w);
        words_from_limbs<9>(w, soa_ld<ModT, 1>(W.la.ay, p * (2 + 2 * W.sec)).l);
        exph_put_coord<9>(m + len + 34, w);
        len += 67;
    }
    m[len] = 0x80;   // padding: 0x80, zeros, the bit length in eight bytes
    for (uint32_t i = len + 1; i < nblk * 64 - 8; i++) m[i] = 0;
    const uint64_t bits = (uint64_t)len * 8;
    for (int i = 0; i < 8; i++) m[nblk * 64 - 1 - i] = (uint8_t)(bits >> (8 * i));
}
void launch_gk_hash(hipStream_t s, const Workspace& W, uint32_t count, const uint8_t* msg) {
    if (count <= W.exph_cap && W.exph_wk && count <= 512 && !getenv_exph_one_lane()) {
        const uint32_t len = 67 * 4 * W.n + (W.hardened ? 24 + 32 + 32 + 65 + 67 : 0), nblk = (len + 9 + 63) / 64;
        if (nblk <= (2 * 67 + W.sec * (65 + 2 * 67) + 9 + 63) / 64) {   // (the buffers were carved for the Exp challenge's blocks)
            hipLaunchKernelGGL(k_gk_msg, dim3((count * (4 * W.n + 1) + 255) / 256), dim3(256), 0, s, W, count, msg, nblk);
            launch_sha_msgs(s, W, count, W.gk_x, nblk, 3);
            return;
        }
    }
    hipLaunchKernelGGL(k_gk_hash, dim3((count + 63) / 64), dim3(64), 0, s, W, count, msg);
}

// ---------------------------------------------------------------- digest of the padded ring (hardened mode)
// SHA-256("ZKAttest-ring-v1" || be64(N) || leaf_0 || leaf_1 || ...), leaf_i = SHA-256 of ring entries [256 i, 256 i + 256) as
// 32-byte big-endian integers: one lane per leaf, then one lane over the leaf digests.
__global__ void __launch_bounds__(64) k_ring_leaves(Soa ring, uint64_t N, uint32_t nleaves, uint32_t* leaf_words) {
    __shared__ uint32_t lds[16 * 64];
    uint32_t t = gtid();
    bool live = t < nleaves;
    if (!live) t = nleaves - 1;
    ShaStream s;
    s.init(lds, threadIdx.x, 64);
    uint64_t e0 = (uint64_t)t * 256, e1 = e0 + 256 < N ? e0 + 256 : N;
    // all lanes of a wave must absorb the same structure: a short last leaf only occurs when it is the only leaf (N < 256)
    // or N is not a multiple of 256, which a padded ring (power of two) excludes for N >= 256
    for (uint64_t e = e0; e < e1; e++) {
        uint32_t w[8];
        words_from_limbs<8>(w, soa_ld<ModQ, 1>(ring, (uint32_t)e).l);
        s.put_be<32>(w);
    }
    uint32_t h[8];
    s.finish(h);
    if (live)
        for (int i = 0; i < 8; i++) leaf_words[8 * t + i] = h[i];
}
__global__ void __launch_bounds__(64) k_ring_root(uint64_t N, uint32_t nleaves, const uint32_t* __restrict__ leaf_words, uint32_t* digest8) {
    __shared__ uint32_t lds[16 * 64];
    ShaStream s;
    s.init(lds, threadIdx.x, 64);
    if (threadIdx.x) return;
    const char tag[] = "ZKAttest-ring-v1";
    for (int i = 0; i < 16; i++) s.put_byte((uint8_t)tag[i]);
    for (int i = 7; i >= 0; i--) s.put_byte((uint32_t)(N >> (8 * i)));
    for (uint32_t l = 0; l < nleaves * 8; l++) {
        uint32_t w = leaf_words[l];
        s.put_byte(w >> 24), s.put_byte(w >> 16), s.put_byte(w >> 8), s.put_byte(w);
    }
    uint32_t h[8];
    s.finish(h);
    for (int i = 0; i < 8; i++) digest8[i] = h[i];
}
void launch_ring_digest(hipStream_t s, const Soa& ring, uint64_t N, uint32_t* leaf_words, uint32_t* digest8) {
    uint32_t nleaves = (uint32_t)((N + 255) / 256);
    hipLaunchKernelGGL(k_ring_leaves, dim3((nleaves + 63) / 64), dim3(64), 0, s, ring, N, nleaves, leaf_words);
    hipLaunchKernelGGL(k_ring_root, dim3(1), dim3(64), 0, s, N, nleaves, leaf_words, digest8);
}

// ---------------------------------------------------------------- unit-test hooks
__global__ void __launch_bounds__(64) k_test_sha256(uint64_t count, uint64_t len, const uint8_t* msgs, uint8_t* out) {
    __shared__ uint32_t lds[16 * 64];
    uint64_t t = gtid();
    bool live = t < count;
    if (!live) t = count - 1;
    ShaStream s;
    s.init(lds, threadIdx.x, 64);
    for (uint64_t i = 0; i < len; i++) s.put_byte(msgs[t * len + i]);
    uint32_t h[8];
    s.finish(h);
    if (live)
        for (int i = 0; i < 8; i++) ((uint32_t*)out)[t * 8 + i] = bswap32(h[i]);
}
void launch_test_sha256(hipStream_t s, uint64_t count, uint64_t len, const uint8_t* d_msgs, uint8_t* d_out) {
    hipLaunchKernelGGL(k_test_sha256, dim3((uint32_t)((count + 63) / 64)), dim3(64), 0, s, count, len, d_msgs, d_out);
}
__global__ void k_test_rng(RngCtx g, uint64_t B, uint32_t first_k, uint32_t n_k, uint8_t* out) {
    uint64_t t = gtid();
    if (t >= B * n_k) return;
    uint32_t p = (uint32_t)(t / n_k), k = first_k + (uint32_t)(t % n_k);
    uint32_t w[8];
    rng_draw_words(g, p, k, w);
    store_be<8>(out + 32 * t, w);
}
void launch_test_rng(hipStream_t s, const RngCtx& g, uint64_t B, uint32_t first_k, uint32_t n_k, uint8_t* d_out) {
    uint64_t n = B * n_k;
    hipLaunchKernelGGL(k_test_rng, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, g, B, first_k, n_k, d_out);
}
