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
// instruction.  The 65 diagonals are recombined with carries into the same 18-limb integer the VALU kernel forms and go through
// the same redc_wide: the result is bit-identical (exact integer arithmetic), so verdicts and statuses cannot change.
//
// Tiling: one wave = 16 proofs x 16 blocks; accumulators 65 diagonals x 4 registers; per 64 values of i the wave keeps the 33
// digit planes of its block tile in registers (132) and streams the 33 planes of its proof tile (one 1 KB fragment each, issued
// two ahead): 33 x 33 matrix instructions per 66 KB of operands.  One wave per SIMD (~430 registers).  Operands are stored in
// fragment order (what a lane needs is 16 contiguous bytes, a wave reads 1 KB runs), both sides use the same assignment of i to
// (lane group, byte), so the result does not depend on the hardware's internal K order.
// Placement: workgroup w runs on XCD w % 8; an XCD works through rectangles of 16 proof tiles x 8 block tiles (one workgroup of
// 4 block tiles per CU), whose operands (3.2 MB) stay in its L2.
#include "engine.h"

typedef int v4i __attribute__((ext_vector_type(4)));
#define GKM_ND 33               // balanced base-256 digits of a 256-bit integer
#define GKM_NDIAG (2 * GKM_ND - 1)
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

// 65 diagonal sums -> the 18-limb radix-2^30 integer sum_d D_d 256^d (non-negative: it IS sum_i c_i key_i)
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

__global__ void __launch_bounds__(256, 1) k_v_gk_block_mfma(uint32_t count, const int8_t* __restrict__ afrag, const int8_t* __restrict__ bfrag, uint32_t nblocks, Soa res) {
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
#pragma unroll 1
    for (uint32_t kc = 0; kc < 4; kc++) {
        const int8_t* ak = ap + (size_t)kc * GKM_ND * GKM_FRAG;
        const int8_t* bk = bp + (size_t)kc * GKM_ND * GKM_FRAG;
        v4i b[GKM_ND];
#pragma unroll
        for (int v = 0; v < GKM_ND; v++) b[v] = gkm_ld(bk + (size_t)v * GKM_FRAG, lane);
        v4i a0 = gkm_ld(ak, lane), a1 = gkm_ld(ak + GKM_FRAG, lane), a2;
#pragma unroll
        for (int u = 0; u < GKM_ND; u++) {
            if (u + 2 < GKM_ND) a2 = gkm_ld(ak + (size_t)(u + 2) * GKM_FRAG, lane);   // two fragments ahead
#pragma unroll
            for (int v = 0; v < GKM_ND; v++) acc[u + v] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b[v], acc[u + v], 0, 0, 0);
            a0 = a1, a1 = a2;
        }
    }
    // D layout: lane l, register r <-> proof 4 (l >> 4) + r, block l & 15
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
        const uint32_t p = tile_p * 16 + 4 * (lane >> 4) + r, block = tile_b * 16 + (lane & 15);
        int32_t D[GKM_NDIAG];
#pragma unroll
        for (int d = 0; d < GKM_NDIAG; d++) D[d] = r == 0 ? acc[d].x : r == 1 ? acc[d].y : r == 2 ? acc[d].z : acc[d].w;
        uint32_t t30[18];
        gkm_recombine(D, t30);
        if (p < count) soa_st(res, p * nblocks + block, fe_canon(redc_wide(t30)));
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
    hipLaunchKernelGGL(k_v_gk_block_mfma, dim3(nwg), dim3(256), 0, s, count, coef_frag, ring_frag, nblocks, res);
}
