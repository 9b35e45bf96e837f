// Verifier's ring fold on the matrix pipe (the one deviation from "no MFMA" on this path).
//
// verifyMembership (src/proofGK/gk.ts:239-250) computes total = sum_i key_i prod_j f_{j,i_j}(x): N (n + 1) modular
// multiplications per proof, N = ring size.  In the ratio form of k_gk.hip the 8 low index bits of a block of 256 keys
// contribute  T[p][b] = sum_{i < 256} c_{p,i} * key_{b,i}  -- an exact INTEGER matrix product
//     C [proofs x 256]  x  K [256 x blocks]        (entries: 256-bit integers, K shared by every proof)
// followed by ONE modular reduction per (proof, block).  k_v_gk_block does it on the VALU: 256 x 81 v_mad_u64_u32 per entry of
// T (0.64 of the multiplier peak at ring 2^20, 63 % of that verifier's time).  Here both factors are split into 33 BALANCED
// base-256 digits (x = sum_u x_u 256^u, x_u in [-128, 127]), so that
//     T = sum_d 256^d  D_d,     D_d[p][b] = sum_{u + v = d}  sum_i  c_{p,i}[u] * key_{b,i}[v]
// and every D_d is a sum of int8 matrix products with exact int32 accumulation: |digit product| <= 2^14, 256 values of i and at
// most 33 (u, v) pairs per diagonal stay below 2^27.1.  v_mfma_i32_16x16x64_i8 does 16 384 of those multiply-adds per
// instruction.  The diagonals (64 accumulators: diagonal 64, the product of the two carry digits, rides on 63 scaled by 256) are recombined with carries into the same 18-limb integer the VALU kernel forms and go through
// the same redc_wide: the result is bit-identical (exact integer arithmetic), so verdicts and statuses cannot change.
//
// Tiling: one wave = 16 proofs x 16 blocks; accumulators 64 diagonals x 4 registers, all of the 256 AGPRs; per 64 values of i the wave keeps the 33
// digit planes of its block tile in registers (132) and streams the 33 planes of its proof tile (one 1 KB fragment each, issued
// two ahead): 33 x 33 matrix instructions per 66 KB of operands.  One wave per SIMD (~430 registers).  Operands are stored in
// fragment order (what a lane needs is 16 contiguous bytes, a wave reads 1 KB runs), both sides use the same assignment of i to
// (lane group, byte), so the result does not depend on the hardware's internal K order.
// Placement: workgroup w runs on XCD w % 8; an XCD works through rectangles of 16 proof tiles x 8 block tiles (one workgroup of
// 4 block tiles per CU), whose operands (3.2 MB) stay in its L2.
#include "engine.h"

typedef int v4i __attribute__((ext_vector_type(4)));
#define GKM_ND 33               // balanced base-256 digits of a 256-bit integer
#define GKM_NDIAG (2 * GKM_ND - 2)   // 64 accumulators = 256 registers: the product of the two top digits (0 or 1 each: the carry out of digit 31)
                                    // joins diagonal 63 scaled by 256, as (16 a) x (16 b) -- see gkm_mfma_row
#define GKM_FRAG 1024           // bytes of one operand fragment: 64 lanes x 16
#define GKM_TILE_BYTES (4 * GKM_ND * GKM_FRAG)   // one tile of 16 rows: 4 chunks of 64 i x 33 digits

// 8 little-endian words -> 33 balanced digits
ZK_DEV void gkm_digits(const uint32_t w[8], int8_t d[GKM_ND]) {
    uint32_t carry = 0;
#pragma unroll
    for (int u = 0; u < 32; u++) {
        uint32_t t = ((w[u >> 2] >> (8 * (u & 3))) & 255u) + carry;
        carry = t >= 128u;
        d[u] = (int8_t)(carry ? (int)t - 256 : (int)t);
    }
    d[32] = (int8_t)carry;
}
// fragment address of digit u of row r (0..15 of tile `tile`), index i (0..255)
ZK_DEV size_t gkm_addr(uint32_t tile, uint32_t r, uint32_t i, uint32_t u) {
    const uint32_t kc = i >> 6, g = (i >> 4) & 3, j = i & 15;
    return ((size_t)(tile * 4 + kc) * GKM_ND + u) * GKM_FRAG + (size_t)((g << 4) | r) * 16 + j;
}
// ring -> block-side fragments, once per ring (zk_ctx_set_ring): key(block, i), block = tile * 16 + r
__global__ void __launch_bounds__(256) k_gkm_ring_digits(Soa ring, uint32_t nblocks, int8_t* frag) {
    uint32_t t = gtid();
    if (t >= nblocks * 256) return;
    uint32_t block = t >> 8, i = t & 255;
    uint32_t w[8];
    words_from_limbs<8>(w, soa_ld<ModQ, 1>(ring, block * 256 + i).l);
    int8_t d[GKM_ND];
    gkm_digits(w, d);
#pragma unroll
    for (int u = 0; u < GKM_ND; u++) frag[gkm_addr(block >> 4, block & 15, i, u)] = d[u];
}
// proof-side fragments: the coefficients c_i of k_v_gk_csub (same values), as digits
__global__ void __launch_bounds__(256) k_gkm_coef_digits(VWork V, uint32_t count, int8_t* frag) {
    uint32_t t = gtid();
    if (t >= count * 256) return;
    uint32_t p = t >> 8, i = t & 255;
    Fe<ModQ, 2> acc = fe_one_mont<ModQ>().as<2>();
    bool zero = false;
    for (uint32_t j = 0; j < 8; j++) {
        bool swap = V.gk_swap[j * V.C + p] != 0, set = (i >> j) & 1;
        if (swap) zero = zero || !set;
        else if (set) acc = acc * soa_ld<ModQ, 2>(V.gk_f, j * V.C + p);
    }
    Fe<ModQ, 1> c = zero ? fe_zero<ModQ>() : fe_canon(acc);
    uint32_t w[8];
    words_from_limbs<8>(w, c.l);
    int8_t d[GKM_ND];
    gkm_digits(w, d);
#pragma unroll
    for (int u = 0; u < GKM_ND; u++) frag[gkm_addr(p >> 4, p & 15, i, u)] = d[u];
}

ZK_DEV v4i gkm_ld(const int8_t* base, uint32_t lane) { return *(const v4i*)(base + (size_t)lane * 16); }

// digit u of A against all digits of B
ZK_DEV void gkm_mfma_row(int u, const v4i& a, const v4i (&b)[GKM_ND], v4i (&acc)[GKM_NDIAG]) {
#pragma unroll
    for (int v = 0; v < GKM_ND; v++) {
        if (u + v < GKM_NDIAG) acc[u + v] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b[v], acc[u + v], 0, 0, 0);
        else acc[GKM_NDIAG - 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a << 4, b[v] << 4, acc[GKM_NDIAG - 1], 0, 0, 0);   // u = v = 32: bytes 0 / 1 -> 16
    }
}
// the same against the digits [V0, V0 + NV) of B only, held in b[0 .. NV): one half of the planes (GKM_DOUBLE_BUFFER)
template <int V0, int NV>
ZK_DEV void gkm_mfma_row_part(int u, const v4i& a, const v4i (&b)[NV], v4i (&acc)[GKM_NDIAG]) {
#pragma unroll
    for (int v = V0; v < V0 + NV; v++) {
        if (u + v < GKM_NDIAG) acc[u + v] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b[v - V0], acc[u + v], 0, 0, 0);
        else acc[GKM_NDIAG - 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a << 4, b[v - V0] << 4, acc[GKM_NDIAG - 1], 0, 0, 0);
    }
}
// EXPERIMENT, default off (round 5's review item 6; profiles/r06_ab_variants.txt (10)).  One chunk of 64 values of i with the B planes in two halves (17 + 16): while the
// matrix pipe works through one half against all 33 planes of A, the other half's registers are free and take the NEXT chunk's planes, so that the 33 loads that head every
// chunk no longer stand in front of its first matrix instruction.  The price decides it: A is streamed TWICE per chunk, one 1 KB plane per 17 matrix instructions instead of
// per 33 -- 15 bytes per cycle and CU from the L2, about 9 TB/s chip-wide -- and the kernels become operand-bound: gk_fold 15.6 -> 21.6 ms per step, v_gk_total 6.4 -> 10.0 ms,
// verification at ring 2^20 409 -> 330 k/s (same box, twice).  Same results (the matrix-pipe tests pass on it); not used.
#ifndef GKM_DOUBLE_BUFFER
#define GKM_DOUBLE_BUFFER 0
#endif
#define GKM_H0 17
#define GKM_H1 (GKM_ND - GKM_H0)
ZK_DEV void gkm_chunk_halves(const int8_t* ak, const int8_t* bnext, uint32_t lane, v4i (&b0)[GKM_H0], v4i (&b1)[GKM_H1], v4i (&acc)[GKM_NDIAG]) {
    {
        v4i a0 = gkm_ld(ak, lane), a1 = gkm_ld(ak + GKM_FRAG, lane), a2;
#pragma unroll
        for (int u = 0; u < GKM_ND; u++) {
            if (u + 2 < GKM_ND) a2 = gkm_ld(ak + (size_t)(u + 2) * GKM_FRAG, lane);
            gkm_mfma_row_part<0, GKM_H0>(u, a0, b0, acc);
            a0 = a1, a1 = a2;
        }
    }
    if (bnext) {
#pragma unroll
        for (int v = 0; v < GKM_H0; v++) b0[v] = gkm_ld(bnext + (size_t)v * GKM_FRAG, lane);
    }
    {
        v4i a0 = gkm_ld(ak, lane), a1 = gkm_ld(ak + GKM_FRAG, lane), a2;
#pragma unroll
        for (int u = 0; u < GKM_ND; u++) {
            if (u + 2 < GKM_ND) a2 = gkm_ld(ak + (size_t)(u + 2) * GKM_FRAG, lane);
            gkm_mfma_row_part<GKM_H0, GKM_H1>(u, a0, b1, acc);
            a0 = a1, a1 = a2;
        }
    }
    if (bnext) {
#pragma unroll
        for (int v = 0; v < GKM_H1; v++) b1[v] = gkm_ld(bnext + (size_t)(GKM_H0 + v) * GKM_FRAG, lane);
    }
}

// 64 diagonal sums -> the 18-limb radix-2^30 integer sum_d D_d 256^d (non-negative: it IS sum_i c_i key_i)
ZK_DEV void gkm_recombine(const int32_t D[GKM_NDIAG], uint32_t t30[18]) {
    uint8_t by[72];
    int64_t c = 0;
#pragma unroll
    for (int d = 0; d < GKM_NDIAG; d++) {
        c += D[d];
        by[d] = (uint8_t)(c & 255);
        c >>= 8;   // arithmetic: the running value may be negative in between
    }
#pragma unroll
    for (int d = GKM_NDIAG; d < 72; d++) {
        by[d] = (uint8_t)(c & 255);
        c >>= 8;
    }
#pragma unroll
    for (int k = 0; k < 18; k++) {   // bits [30k, 30k + 30)
        const int bit = 30 * k, b0 = bit >> 3, sh = bit & 7;
        uint64_t v = 0;
#pragma unroll
        for (int q = 0; q < 5; q++)
            if (b0 + q < 72) v |= (uint64_t)by[b0 + q] << (8 * q);
        t30[k] = (uint32_t)(v >> sh) & LIMB_MASK;
    }
}

// the epilogue of the prover's matrix-core kernel at ONE wave per SIMD: accumulator components (proofs of a lane) two at a time in lock step, so that
// the 64-bit multiply-adds of independent carry chains overlap (tools/valu_peak.hip: 15.8 cycles per dependent v_mad_u64_u32 with one
// chain, 8.1 with two; four at a time spill)
template <int NJ>
ZK_DEV void gkm_recombine_n(const int32_t (&D)[NJ][GKM_NDIAG], uint32_t (&t30)[NJ][18]) {   // gkm_recombine, NJ rows in lock step, limbs packed on the fly
    int64_t c[NJ] = {};
    uint64_t buf[NJ] = {};
    int nb = 0, k = 0;
#pragma unroll
    for (int d = 0; d < 72; d++) {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            if (d < GKM_NDIAG) c[j] += D[j][d];
            buf[j] |= (uint64_t)(c[j] & 255) << nb;
            c[j] >>= 8;   // arithmetic: the running value may be negative in between
        }
        nb += 8;
        if (nb >= 30 && k < 18) {
#pragma unroll
            for (int j = 0; j < NJ; j++) t30[j][k] = (uint32_t)buf[j] & LIMB_MASK, buf[j] >>= 30;
            k++, nb -= 30;
        }
    }
#pragma unroll
    for (int kk = 0; kk < 18; kk++)
        if (kk >= k) {
#pragma unroll
            for (int j = 0; j < NJ; j++) t30[j][kk] = (uint32_t)buf[j] & LIMB_MASK, buf[j] >>= 30;
        }
}
template <int NJ>
ZK_DEV void redc_wide_n(const uint32_t (&T)[NJ][18], Fe<ModQ, 2> (&r)[NJ]) {   // redc_wide (engine.h), NJ values in lock step
    uint64_t a[NJ] = {};
    uint32_t m[NJ][NLIMB];
#pragma unroll
    for (int k = 0; k < NLIMB; k++) {
#pragma unroll
        for (int j = 0; j < NJ; j++) a[j] += T[j][k];
#pragma unroll
        for (int i = 0; i < k; i++)
#pragma unroll
            for (int j = 0; j < NJ; j++) a[j] = mad64(m[j][i], ModQ::mod[k - i], a[j]);
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            m[j][k] = ((uint32_t)a[j] * ModQ::n0) & LIMB_MASK;
            a[j] = mad64(m[j][k], ModQ::mod[0], a[j]);
            a[j] >>= LIMB_BITS;
        }
    }
#pragma unroll
    for (int k = NLIMB; k < 2 * NLIMB; k++) {
#pragma unroll
        for (int j = 0; j < NJ; j++) a[j] += T[j][k];
#pragma unroll
        for (int i = k - (NLIMB - 1); i < NLIMB; i++)
#pragma unroll
            for (int j = 0; j < NJ; j++) a[j] = mad64(m[j][i], ModQ::mod[k - i], a[j]);
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            if (k < 2 * NLIMB - 1) r[j].l[k - NLIMB] = (uint32_t)a[j] & LIMB_MASK, a[j] >>= LIMB_BITS;
            else r[j].l[NLIMB - 1] = (uint32_t)a[j];
        }
    }
}
__global__ void __launch_bounds__(256, 1) k_v_gk_block_mfma(uint32_t count, const int8_t* __restrict__ afrag, const int8_t* __restrict__ bfrag, uint32_t nblocks, uint32_t nkc, Soa res) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t tiles_p = (count + 15) >> 4, tiles_b = nblocks >> 4;
    // rectangle walk, XCD-aware (see the header)
    const uint32_t g = blockIdx.x, xcd = g & 7, slot = g >> 3;
    const uint32_t rects_b = (tiles_b + 7) >> 3;
    const uint32_t rect = (slot >> 5) * 8 + xcd, in = slot & 31;
    const uint32_t rp = rect / rects_b, rb = rect % rects_b;
    const uint32_t tile_p = rp * 16 + (in & 15), tile_b = rb * 8 + (in >> 4) * 4 + wave;
    if (tile_p >= tiles_p || tile_b >= tiles_b) return;
    v4i acc[GKM_NDIAG];
#pragma unroll
    for (int d = 0; d < GKM_NDIAG; d++) acc[d] = (v4i){0, 0, 0, 0};
    const int8_t* ap = afrag + (size_t)tile_p * GKM_TILE_BYTES;
    const int8_t* bp = bfrag + (size_t)tile_b * GKM_TILE_BYTES;
#if GKM_DOUBLE_BUFFER
    {
        v4i b0[GKM_H0], b1[GKM_H1];
#pragma unroll
        for (int v = 0; v < GKM_H0; v++) b0[v] = gkm_ld(bp + (size_t)v * GKM_FRAG, lane);
#pragma unroll
        for (int v = 0; v < GKM_H1; v++) b1[v] = gkm_ld(bp + (size_t)(GKM_H0 + v) * GKM_FRAG, lane);
#pragma unroll 1
        for (uint32_t kc = 0; kc < nkc; kc++)
            gkm_chunk_halves(ap + (size_t)kc * GKM_ND * GKM_FRAG, kc + 1 < nkc ? bp + (size_t)(kc + 1) * GKM_ND * GKM_FRAG : nullptr, lane, b0, b1, acc);
    }
#else
#pragma unroll 1
    for (uint32_t kc = 0; kc < nkc; kc++) {
        const int8_t* ak = ap + (size_t)kc * GKM_ND * GKM_FRAG;
        const int8_t* bk = bp + (size_t)kc * GKM_ND * GKM_FRAG;
        v4i b[GKM_ND];
#pragma unroll
        for (int v = 0; v < GKM_ND; v++) b[v] = gkm_ld(bk + (size_t)v * GKM_FRAG, lane);
        v4i a0 = gkm_ld(ak, lane), a1 = gkm_ld(ak + GKM_FRAG, lane), a2;
#pragma unroll
        for (int u = 0; u < GKM_ND; u++) {
            if (u + 2 < GKM_ND) a2 = gkm_ld(ak + (size_t)(u + 2) * GKM_FRAG, lane);   // two fragments ahead
            gkm_mfma_row(u, a0, b, acc);
            a0 = a1, a1 = a2;
        }
    }
#endif
    // D layout: lane l, register r <-> proof 4 (l >> 4) + r, block l & 15; two proofs at a time in lock step (see gkm_recombine_n)
    const uint32_t pb = tile_p * 16 + 4 * (lane >> 4), block = tile_b * 16 + (lane & 15);
#pragma unroll 1
    for (int h = 0; h < 2; h++) {
        int32_t D[2][GKM_NDIAG];
#pragma unroll
        for (int d = 0; d < GKM_NDIAG; d++) D[0][d] = h ? acc[d].z : acc[d].x, D[1][d] = h ? acc[d].w : acc[d].y;
        uint32_t t30[2][18];
        gkm_recombine_n<2>(D, t30);
        Fe<ModQ, 2> r2[2];
        redc_wide_n<2>(t30, r2);
#pragma unroll
        for (int j = 0; j < 2; j++)
            if (pb + 2 * h + j < count) soa_st(res, (pb + 2 * h + j) * nblocks + block, fe_canon(r2[j]));
    }
}

size_t gkm_ring_frag_bytes(uint64_t N) { return (size_t)(N >> 12) * GKM_TILE_BYTES; }   // N / 256 blocks, 16 per tile
size_t gkm_coef_frag_bytes(uint32_t C) { return (size_t)((C + 15) >> 4) * GKM_TILE_BYTES; }
void launch_gkm_ring_digits(hipStream_t s, const Soa& ring, uint32_t nblocks, int8_t* frag) {
    hipLaunchKernelGGL(k_gkm_ring_digits, dim3(nblocks), dim3(256), 0, s, ring, nblocks, frag);
}
void launch_v_gk_block_mfma(hipStream_t s, const VWork& V, const int8_t* ring_frag, uint32_t nblocks, uint32_t count, int8_t* coef_frag, const Soa& res) {
    hipLaunchKernelGGL(k_gkm_coef_digits, dim3(count), dim3(256), 0, s, V, count, coef_frag);
    const uint32_t tiles_p = (count + 15) >> 4, tiles_b = nblocks >> 4;
    const uint32_t rects = ((tiles_p + 15) >> 4) * ((tiles_b + 7) >> 3);
    const uint32_t nwg = ((rects + 7) / 8) * 8 * 32;   // 32 workgroups per rectangle, rectangles dealt to the 8 XCDs
    hipLaunchKernelGGL(k_v_gk_block_mfma, dim3(nwg), dim3(256), 0, s, count, coef_frag, ring_frag, nblocks, 4u, res);
}

// ---------------------------------------------------------------- prover: the table path's big coefficient classes
// k_gk_block (k_gk.hip) computes, per (proof, block), coefficient k of the block's polynomial as sum over the subsets S of size 8 - k of
// a_S * D_S(l_low)[block]: 255 multiply-accumulates of 256-bit integers, one reduction per coefficient.  D_S(l) is table E: for proofs
// with the SAME l_low it is a shared factor, so a tile of 16 such proofs x 16 blocks is the matrix product
//     A [16 proofs x |class|]  x  D_l [|class| x 16 blocks]        per coefficient class,
// done here for the classes k = 2..6 (28, 56, 70, 56, 28 subsets: 238 of the 255 products) exactly like the verifier's fold: balanced
// base-256 digits, exact int32 diagonals, recombination, the same redc_wide -- bit-identical coefficients.  The classes are padded to
// chunks of 64 subsets (6 chunks: 28 | 56 | 64 + 6 | 56 | 28, zero padding on both sides); the small classes (1, 8, 8 subsets) and the
// x^8 term stay in k_gk_block.  Proofs are already sorted by l_low (k_gk_sort); a group of n proofs owns ceil(n / 16) tiles.
#define GKP_CHUNKS 6
__device__ const uint8_t GKP_BASE[GKP_CHUNKS] = {9, 37, 93, 157, 163, 219};   // first rank of the chunk (GK_RT.kstart: 0, 1, 9, 37, 93, 163, 219, 247, 255)
__device__ const uint8_t GKP_CNT[GKP_CHUNKS] = {28, 56, 64, 6, 56, 28};
#define GKP_TILE_BYTES (GKP_CHUNKS * GKM_ND * GKM_FRAG)
// rank (9..246) -> chunk and position inside it
ZK_DEV void gkp_place(uint32_t rank, uint32_t& c, uint32_t& kpos) {
    c = rank < 37 ? 0 : rank < 93 ? 1 : rank < 157 ? 2 : rank < 163 ? 3 : rank < 219 ? 4 : 5;
    kpos = rank - GKP_BASE[c];
}
ZK_DEV size_t gkp_addr(size_t tile, uint32_t r, uint32_t c, uint32_t kpos, uint32_t u) {
    return ((tile * GKP_CHUNKS + c) * GKM_ND + u) * GKM_FRAG + (size_t)((((kpos >> 4) & 3) << 4) | r) * 16 + (kpos & 15);
}
// table E (29-bit limbs, [l_low][rank][limb][block]) -> digit fragments [l_low][block tile][chunk][digit]; the buffer is zeroed first
__global__ void __launch_bounds__(256) k_gkm_etab_digits(const uint32_t* __restrict__ E, uint32_t nblocks, int8_t* edig) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;   // (l_low, rank - 9, block), block fastest
    uint64_t total = (uint64_t)256 * 238 * nblocks;
    if (t >= total) return;
    uint32_t block = (uint32_t)(t % nblocks);
    uint32_t lr = (uint32_t)(t / nblocks), rank = 9 + lr % 238, llow = lr / 238;
    uint32_t l29[9], l30[9], w[8];
#pragma unroll
    for (int l = 0; l < 9; l++) l29[l] = E[(((size_t)llow * 256 + rank) * 9 + l) * nblocks + block];
    limbs_repack<29, 30, 9, 9>(l30, l29);
    words_from_limbs<8>(w, l30);
    int8_t d[GKM_ND];
    gkm_digits(w, d);
    uint32_t c, kpos;
    gkp_place(rank, c, kpos);
    const size_t tile = (size_t)llow * (nblocks >> 4) + (block >> 4);
#pragma unroll
    for (int u = 0; u < GKM_ND; u++) edig[gkp_addr(tile, block & 15, c, kpos, u)] = d[u];
}
// a_S of the chunk's proofs (k_gk_asub: 29-bit limbs) -> digit fragments, rows in sorted order; the buffer is zeroed first
__global__ void __launch_bounds__(256) k_gkm_asub_digits(ChunkIn in, const uint32_t* __restrict__ asub, const uint32_t* __restrict__ order, const uint32_t* __restrict__ goff,
                                                         const uint32_t* __restrict__ toff, int8_t* adig) {
    uint32_t t = gtid();
    if (t >= in.count * 238) return;
    uint32_t pos = t / 238, rank = 9 + t % 238;
    uint32_t p = order[pos], g = in.which[p] & 255, idx = pos - goff[g];
    uint32_t l29[9], l30[9], w[8];
#pragma unroll
    for (int l = 0; l < 9; l++) l29[l] = asub[((size_t)p * 256 + rank) * 9 + l];
    limbs_repack<29, 30, 9, 9>(l30, l29);
    words_from_limbs<8>(w, l30);
    int8_t d[GKM_ND];
    gkm_digits(w, d);
    uint32_t c, kpos;
    gkp_place(rank, c, kpos);
    const size_t tile = toff[g] + (idx >> 4);
#pragma unroll
    for (int u = 0; u < GKM_ND; u++) adig[gkp_addr(tile, idx & 15, c, kpos, u)] = d[u];
}
// one wave: tile t (16 proofs of one l_low group) x 16 blocks x one coefficient class; workgroup = 4 consecutive block tiles of one proof tile
__global__ void __launch_bounds__(256, 1) k_gk_block_mfma(Workspace W, ChunkIn in, const int8_t* __restrict__ adig, const int8_t* __restrict__ edig,
                                                         const uint32_t* __restrict__ order, const uint32_t* __restrict__ goff, const uint32_t* __restrict__ toff,
                                                         uint32_t nblocks, uint32_t tiles_max, Soa res) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t tiles_b = nblocks >> 4, wgb = (tiles_b + 3) >> 2;
    // XCD-aware placement: workgroup i runs on XCD i % 8; XCD x takes the x-th contiguous eighth of the (block quad, proof tile) list, proof
    // tile fastest, so that the tiles of one l_low group -- they share the B operand, 200 KB per block tile -- meet in one L2
    const uint32_t nwork = tiles_max * wgb, seg = (nwork + 7) >> 3;
    const uint32_t widx = (blockIdx.x & 7) * seg + (blockIdx.x >> 3);
    if (widx >= nwork || (blockIdx.x >> 3) >= seg) return;
    const uint32_t t = widx % tiles_max, tile_b = (widx / tiles_max) * 4 + wave;
    if (t >= toff[256] || tile_b >= tiles_b) return;
    uint32_t lo = 0, hi = 256;   // the group of tile t: largest g with toff[g] <= t (empty groups have toff[g] == toff[g + 1])
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (toff[mid] <= t) lo = mid;
        else hi = mid;
    }
    const uint32_t g = lo, g0 = goff[g], gcnt = goff[g + 1] - g0, row0 = (t - toff[g]) * 16;
    const int8_t* ap = adig + (size_t)t * GKP_TILE_BYTES;
    const int8_t* bp = edig + ((size_t)g * tiles_b + tile_b) * GKP_TILE_BYTES;
    // blockIdx.y = coefficient class: chunks 0 | 1 | 2 + 3 | 4 | 5 -> coefficients 2, 3, 4, 5, 6 (coefficient 4 spans 64 + 6 subsets)
    const uint32_t cls = blockIdx.y, c0 = cls < 3 ? cls : cls + 1, c1 = cls == 2 ? 4 : c0 + 1, k = cls + 2;
    v4i acc[GKM_NDIAG];
#pragma unroll
    for (int d = 0; d < GKM_NDIAG; d++) acc[d] = (v4i){0, 0, 0, 0};
#if GKM_DOUBLE_BUFFER
    {
        const int8_t* bk0 = bp + (size_t)c0 * GKM_ND * GKM_FRAG;
        v4i b0[GKM_H0], b1[GKM_H1];
#pragma unroll
        for (int v = 0; v < GKM_H0; v++) b0[v] = gkm_ld(bk0 + (size_t)v * GKM_FRAG, lane);
#pragma unroll
        for (int v = 0; v < GKM_H1; v++) b1[v] = gkm_ld(bk0 + (size_t)(GKM_H0 + v) * GKM_FRAG, lane);
#pragma unroll 1
        for (uint32_t c = c0; c < c1; c++)
            gkm_chunk_halves(ap + (size_t)c * GKM_ND * GKM_FRAG, c + 1 < c1 ? bp + (size_t)(c + 1) * GKM_ND * GKM_FRAG : nullptr, lane, b0, b1, acc);
    }
#else
#pragma unroll 1
    for (uint32_t c = c0; c < c1; c++) {
        const int8_t* ak = ap + (size_t)c * GKM_ND * GKM_FRAG;
        const int8_t* bk = bp + (size_t)c * GKM_ND * GKM_FRAG;
        v4i b[GKM_ND];
#pragma unroll
        for (int v = 0; v < GKM_ND; v++) b[v] = gkm_ld(bk + (size_t)v * GKM_FRAG, lane);
        v4i a0 = gkm_ld(ak, lane), a1 = gkm_ld(ak + GKM_FRAG, lane), a2;
#pragma unroll
        for (int u = 0; u < GKM_ND; u++) {
            if (u + 2 < GKM_ND) a2 = gkm_ld(ak + (size_t)(u + 2) * GKM_FRAG, lane);
            gkm_mfma_row(u, a0, b, acc);
            a0 = a1, a1 = a2;
        }
    }
#endif
    // D layout: lane l, register r <-> row 4 (l >> 4) + r of the tile, block l & 15
    const uint32_t rowb = row0 + 4 * (lane >> 4), block = tile_b * 16 + (lane & 15);
    // epilogue at ONE wave per SIMD, where nothing hides a dependent chain: the lane's rows two at a time in lock step (tools/valu_peak.hip:
    // 15.8 cycles per dependent v_mad_u64_u32 with one chain, 8.1 with two; four at a time spill).  3.74 -> 2.73 ms per 16 384 proofs.
#pragma unroll 1
    for (int h = 0; h < 2; h++) {   // rows 2h, 2h + 1 of the lane
        int32_t D[2][GKM_NDIAG];
#pragma unroll
        for (int d = 0; d < GKM_NDIAG; d++) D[0][d] = h ? acc[d].z : acc[d].x, D[1][d] = h ? acc[d].w : acc[d].y;
        uint32_t t30[2][18];
        gkm_recombine_n<2>(D, t30);
        Fe<ModQ, 2> r2[2];
        redc_wide_n<2>(t30, r2);
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const uint32_t row = rowb + 2 * h + j;
            if (row < gcnt) soa_st(res, (k * W.C + order[g0 + row]) * nblocks + block, fe_canon(r2[j]));
        }
    }
}
size_t gkm_etab_frag_bytes(uint64_t N) { return (size_t)256 * (N >> 12) * GKP_TILE_BYTES; }
size_t gkm_asub_frag_bytes(uint32_t C) { return (size_t)((C >> 4) + 256 + 1) * GKP_TILE_BYTES; }
void launch_gkm_etab_digits(hipStream_t s, const uint32_t* E, uint32_t nblocks, int8_t* edig) {
    hipMemsetAsync(edig, 0, gkm_etab_frag_bytes((uint64_t)nblocks << 8), s);
    uint64_t total = (uint64_t)256 * 238 * nblocks;
    hipLaunchKernelGGL(k_gkm_etab_digits, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, s, E, nblocks, edig);
}
void launch_gk_block_mfma(hipStream_t s, const Workspace& W, const ChunkIn& in, uint32_t nblocks, const Soa& res) {
    const uint32_t tiles_max = (in.count >> 4) + 256;   // ceil(n_g / 16) summed over 256 groups
    hipMemsetAsync(W.gk_adig, 0, (size_t)tiles_max * GKP_TILE_BYTES, s);
    hipLaunchKernelGGL(k_gkm_asub_digits, dim3((in.count * 238 + 255) / 256), dim3(256), 0, s, in, W.gk_asub, W.gk_order, W.gk_goff, W.gk_toff, W.gk_adig);
    const uint32_t wgb = ((nblocks >> 4) + 3) >> 2;
    hipLaunchKernelGGL(k_gk_block_mfma, dim3(((tiles_max * wgb + 7) >> 3) << 3, 5), dim3(256), 0, s, W, in, W.gk_adig, W.gk_edig, W.gk_order, W.gk_goff, W.gk_toff, nblocks, tiles_max, res);
}
