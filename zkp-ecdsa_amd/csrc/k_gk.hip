// Groth-Kohlweiss prover polynomial, low 8 index bits as ONE multilinear step over a per-ring table.
//
// Reference: proveMembership (src/proofGK/gk.ts:141-171) evaluates d(w) = sum_i (v_l - v_i) p_i(w) at n points and
// interpolates; the engine computes the coefficients of P(x) = sum_i key_i p_i(x) directly,
// p_i(x) = prod_j f_{j,i_j}(x), f_{j,1} = l_j x + a_j, f_{j,0} = (1 - l_j) x - a_j.  Folding index bit j combines a
// pair (ev, od) into  a_j (od - ev) + x sel_j,  sel_j = od if l_j else ev.  Expanding the 8 low bits of a block of 256
// consecutive ring elements at once gives
//     P_block(x) = sum_{S subset of bits 0..7}  (prod_{j in S} a_j)  x^(8-|S|)  D_S(l),
//     D_S(l) = "difference over the bits in S, selection by l over the others" of the block's 256 keys,
// and D_S(l) depends only on the RING and on l's 8 low bits: it is precomputed once per ring for all 256 values of
// l_low (table E, 256 x the ring: 604 MB at N = 2^16 -- HBM is there to be used).  A proof then needs, per block,
// 255 multiply-accumulates a_S * D_S grouped by |S| and only 8 modular reductions, instead of 502 full modular
// multiplications for the same 8 fold levels; the remaining n - 8 levels (N/256 polynomials) go through the LDS
// fold of k_scalar.hip.  Exact integer arithmetic mod q: the coefficients, hence the proof bytes, are unchanged.
//
// Arithmetic: both factors are repacked to 9 limbs of 29 bits, so a column of one product is < 9 * 2^58 and SEVEN
// products can be accumulated in the seventeen 64-bit column sums before a carry pass (radix 2^30 leaves room for one).
// a_S is wave-uniform (one proof per workgroup) and is read through scalar loads: v_mad_u64_u32 takes it from SGPRs.
//
// Locality: the slice of E a proof reads (its l_low) is 2.36 MB per 65 536 ring elements.  Proofs are counting-sorted
// by l_low and the sorted work list is cut into 8 contiguous segments, one per XCD (workgroup w runs on XCD w mod 8),
// so each XCD's 4 MB L2 holds the slice its resident workgroups share and HBM sees every slice about once per chunk.
#include "engine.h"

#define GKB_BITS 8
#define GKB_SIZE 256

struct GkRankTab {
    uint8_t subset[256];   // rank -> S, ordered by |S| descending (coefficient k = 8 - |S| ascending), then by value
    uint8_t rank[256];     // S -> rank
    uint16_t kstart[10];   // ranks of coefficient k: [kstart[k], kstart[k+1])
};
constexpr GkRankTab gk_make_rank_tab() {
    GkRankTab t{};
    int r = 0;
    for (int k = 0; k <= 8; k++) {
        t.kstart[k] = (uint16_t)r;
        for (int s = 0; s < 256; s++) {
            int pc = 0;
            for (int b = 0; b < 8; b++) pc += (s >> b) & 1;
            if (pc == 8 - k) t.subset[r] = (uint8_t)s, t.rank[s] = (uint8_t)r, r++;
        }
    }
    t.kstart[9] = (uint16_t)r;
    return t;
}
__device__ const GkRankTab GK_RT = gk_make_rank_tab();

// ---------------------------------------------------------------- per-ring table E (zk_ctx_set_ring)
// E[((l_low * 256 + rank(S)) * 9 + limb) * nblocks + block] = limb of D_S(l_low) for that block, canonical, 29-bit limbs.
__global__ void __launch_bounds__(256) k_gk_etab(Soa ring, uint32_t nblocks, uint32_t* E) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t total = (uint64_t)GKB_SIZE * GKB_SIZE * nblocks;
    if (t >= total) return;
    uint32_t block = (uint32_t)(t % nblocks);
    uint32_t lr = (uint32_t)(t / nblocks);
    uint32_t rank = lr & 255, llow = lr >> 8;
    uint32_t S = GK_RT.subset[rank];
    uint32_t fixed = llow & ~S;
    uint32_t pcS = __popc(S);
    Fe<ModQ, 1> acc = fe_zero<ModQ>();
    for (uint32_t T = S;; T = (T - 1) & S) {  // all subsets T of S: element index = l's bits outside S, T inside
        Fe<ModQ, 1> v = soa_ld<ModQ, 1>(ring, block * GKB_SIZE + (fixed | T));
        bool neg = (pcS - __popc(T)) & 1;        // one factor -1 for every bit of S that is 0 in the index (od - ev)
        acc = neg ? fe_sub_mod(acc, v) : fe_add_mod(acc, v);
        if (T == 0) break;
    }
    uint32_t o[9];
    limbs_repack<30, 29, 9, 9>(o, acc.l);
#pragma unroll
    for (int l = 0; l < 9; l++) E[((size_t)lr * 9 + l) * nblocks + block] = o[l];
}
void launch_gk_etab(hipStream_t s, const Soa& ring, uint32_t nblocks, uint32_t* E) {
    uint64_t total = (uint64_t)GKB_SIZE * GKB_SIZE * nblocks;
    hipLaunchKernelGGL(k_gk_etab, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, s, ring, nblocks, E);
}
size_t gk_etab_words(uint64_t N) { return (size_t)GKB_SIZE * GKB_SIZE * 9 * (N / GKB_SIZE); }

// ---------------------------------------------------------------- per chunk: sort by l_low, a_S
// order[pos] = proof, goff[g] = first sorted position of l_low group g (goff[256] = count).  One workgroup of 320 threads (a
// 16-wave workgroup can starve behind the other lane's kernels, see k_scan).
// toff (may be nullptr): toff[g] = first 16-proof tile of group g in the matrix-core path (k_gk_mfma.hip), toff[256] = tiles in all
__global__ void __launch_bounds__(320) k_gk_sort(ChunkIn in, uint32_t* order, uint32_t* goff, uint32_t* toff) {
    __shared__ uint32_t hist[GKB_SIZE], offs[GKB_SIZE + 1];
    uint32_t t = threadIdx.x;
    if (t < GKB_SIZE) hist[t] = 0;
    __syncthreads();
    for (uint32_t p = t; p < in.count; p += blockDim.x) atomicAdd(&hist[in.which[p] & 255], 1u);
    __syncthreads();
    if (t == 0) {
        uint32_t a = 0, tl = 0;
        for (int g = 0; g < GKB_SIZE; g++) {
            offs[g] = a, a += hist[g];
            if (toff) toff[g] = tl, tl += (hist[g] + 15) >> 4;
        }
        offs[GKB_SIZE] = a;
        if (toff) toff[GKB_SIZE] = tl;
    }
    __syncthreads();
    if (t <= GKB_SIZE) goff[t] = offs[t];
    __syncthreads();
    // stable order inside a group is not needed: every proof writes to its own output slot
    for (uint32_t p = t; p < in.count; p += blockDim.x) order[atomicAdd(&offs[in.which[p] & 255], 1u)] = p;
}
// asub[(p * 256 + rank(S)) * 9 + limb] = prod_{j in S} a_j in Montgomery form (value a_S * 2^270 mod q), 29-bit limbs
__global__ void __launch_bounds__(256) k_gk_asub(Workspace W, uint32_t count, Soa am, uint32_t* asub) {
    uint32_t t = gtid();
    if (t >= count * GKB_SIZE) return;
    uint32_t p = t >> 8, rank = t & 255;
    uint32_t S = GK_RT.subset[rank];
    Fe<ModQ, 2> acc = fe_one_mont<ModQ>().as<2>();
    for (uint32_t j = 0; j < GKB_BITS; j++)
        if ((S >> j) & 1) acc = acc * soa_ld<ModQ, 2>(am, j * W.C + p);
    Fe<ModQ, 1> c = fe_canon(acc);
    uint32_t o[9];
    limbs_repack<30, 29, 9, 9>(o, c.l);
#pragma unroll
    for (int l = 0; l < 9; l++) asub[(size_t)t * 9 + l] = o[l];
}

// ---------------------------------------------------------------- the block kernel
// One lane = one block of 256 ring elements of one proof.  UNI: a workgroup's 256 lanes are 256 blocks of ONE proof
// (nblocks a multiple of 256, i.e. n >= 16); otherwise consecutive lanes walk (sorted proof, block) pairs.
// Output: polynomial of the block (9 coefficients, canonical) in the tile format of k_gk_finish:
// res[(k * C + p) * nblocks + block].
struct GkCols {
    uint64_t c[18];
};
ZK_DEV void gkc_zero(GkCols& a) {
#pragma unroll
    for (int i = 0; i < 18; i++) a.c[i] = 0;
}
ZK_DEV void gkc_carry(GkCols& a) {  // keep the value, bring every column below 2^29 (the top one absorbs)
#pragma unroll
    for (int i = 0; i < 17; i++) {
        a.c[i + 1] += a.c[i] >> 29;
        a.c[i] &= (1u << 29) - 1;
    }
}
ZK_DEV void gkc_mac(GkCols& a, const uint32_t x[9], const uint32_t y[9]) {
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
        for (int j = 0; j < 9; j++) a.c[i + j] = mad64(x[i], y[j], a.c[i + j]);
}
template <bool UNI>
__global__ void __launch_bounds__(256) k_gk_block(Workspace W, ChunkIn in, const uint32_t* __restrict__ E, const uint32_t* __restrict__ asub,
                                                  const uint32_t* __restrict__ order, const uint32_t* __restrict__ goff, uint32_t nblocks, uint32_t nwg, Soa res,
                                                  uint32_t kmask) {   // bit k set: coefficient k is computed here (the others on the matrix cores)
    // XCD-aware placement: workgroup w runs on XCD w % 8; XCD x takes the x-th contiguous eighth of the sorted work
    uint32_t w = blockIdx.x, seg = (nwg + 7) / 8;
    uint32_t widx = (w & 7) * seg + (w >> 3);
    if (widx >= nwg) return;
    uint32_t p, block;
    if (UNI) {
        // unit u = (sorted proof, group of 256 blocks); inside an l_low group the units are block-group-major so that
        // concurrently running workgroups share one 2.36 MB sub-slice of E
        uint32_t nbg = nblocks >> 8;
        uint32_t lo = 0, hi = GKB_SIZE;  // largest g with goff[g] * nbg <= widx
        while (hi - lo > 1) {
            uint32_t mid = (lo + hi) >> 1;
            if (goff[mid] * nbg <= widx) lo = mid;
            else hi = mid;
        }
        uint32_t g0 = goff[lo], cnt = goff[lo + 1] - g0;
        uint32_t r = widx - g0 * nbg;
        p = order[g0 + r % cnt];
        block = (r / cnt) * 256 + threadIdx.x;
    } else {
        uint32_t slot = widx * 256 + threadIdx.x;
        if (slot >= in.count * nblocks) return;
        p = order[slot / nblocks];
        block = slot % nblocks;
    }
    uint32_t which = in.which[p];
    uint32_t llow = which & 255;
    const uint32_t* a = asub + (size_t)p * GKB_SIZE * 9;
    const uint32_t* e = E + (size_t)llow * GKB_SIZE * 9 * nblocks + block;
#pragma unroll 1
    for (uint32_t k = 0; k < 8; k++) {
        if (!((kmask >> k) & 1)) continue;
        GkCols acc;
        gkc_zero(acc);
        uint32_t r0 = GK_RT.kstart[k], r1 = GK_RT.kstart[k + 1];
#pragma unroll 1
        for (uint32_t rb = r0; rb < r1; rb += 7) {
            uint32_t re = rb + 7 < r1 ? rb + 7 : r1;
#pragma unroll 1
            for (uint32_t r = rb; r < re; r++) {
                uint32_t x[9], y[9];
#pragma unroll
                for (int l = 0; l < 9; l++) x[l] = a[r * 9 + l], y[l] = e[((size_t)r * 9 + l) * nblocks];
                gkc_mac(acc, x, y);
            }
            gkc_carry(acc);
        }
        // 18 limbs of 29 bits -> 18 limbs of 30 bits -> Montgomery reduction (sum < 70 * 2^512 < q * 2^270)
        uint32_t t29[18], t30[18];
#pragma unroll
        for (int i = 0; i < 18; i++) t29[i] = (uint32_t)acc.c[i];
        limbs_repack<29, 30, 18, 18>(t30, t29);
        Fe<ModQ, 1> c = fe_canon(redc_wide(t30));
        soa_st(res, (k * W.C + p) * nblocks + block, c);
    }
    // x^8 coefficient: D_{} = the key selected by l_low
    soa_st(res, (8 * W.C + p) * nblocks + block, soa_ld<ModQ, 1>(W.ring, block * GKB_SIZE + llow));
}
void launch_gk_block_stage(hipStream_t s, const Workspace& W, const ChunkIn& in, const Soa& am, const Soa& res) {
    uint32_t nblocks = W.N >> GKB_BITS;
    const bool mm = W.gk_edig != nullptr && W.n >= GKM_MINN;   // coefficients 2..6 (238 of the 255 products) on the matrix cores
    hipLaunchKernelGGL(k_gk_sort, dim3(1), dim3(320), 0, s, in, W.gk_order, W.gk_goff, mm ? W.gk_toff : nullptr);
    hipLaunchKernelGGL(k_gk_asub, dim3(in.count), dim3(256), 0, s, W, in.count, am, W.gk_asub);
    bool uni = (nblocks & 255) == 0;
    uint32_t nwg = uni ? in.count * (nblocks >> 8) : (uint32_t)(((uint64_t)in.count * nblocks + 255) / 256);
    uint32_t grid = ((nwg + 7) / 8) * 8;
    const uint32_t kmask = mm ? 0x83u : 0xffu;   // with the matrix path: coefficients 0, 1, 7 (1 + 8 + 8 products) and x^8 stay here
    if (uni) hipLaunchKernelGGL(k_gk_block<true>, dim3(grid), dim3(256), 0, s, W, in, W.gk_etab, W.gk_asub, W.gk_order, W.gk_goff, nblocks, nwg, res, kmask);
    else hipLaunchKernelGGL(k_gk_block<false>, dim3(grid), dim3(256), 0, s, W, in, W.gk_etab, W.gk_asub, W.gk_order, W.gk_goff, nblocks, nwg, res, kmask);
    if (mm) launch_gk_block_mfma(s, W, in, nblocks, res);
}

// ---------------------------------------------------------------- verifier: total = sum_i key_i prod_j f_{j,i_j}(x)  (gk.ts:239-250)
// In the ratio form of k_verify.hip -- (x - f_j) ev + f_j od = g_j (ev + rho_j od), the g_j applied once at the end --
// the 8 low index bits of a block contribute  sum_{i < 256} c_i key_{block,i},  c_i = prod_{j < 8, bit j of i set} rho_j
// (a level with g_j = 0 keeps only its odd branch: factor 1 for a set bit, 0 for a clear one).  The coefficients are the
// same for every block of a proof, and the keys of one index i across blocks are the rank-255 (S = {}) row of slice
// l_low = i of table E, i.e. the ring transposed to [i][limb][block]: 256 multiply-accumulates and ONE reduction per
// block instead of 255 modular multiplications, coalesced across lanes, coefficients in SGPRs.
__global__ void __launch_bounds__(256) k_v_gk_csub(VWork V, uint32_t count, uint32_t* csub) {
    uint32_t t = gtid();
    if (t >= count * GKB_SIZE) return;
    uint32_t p = t >> 8, i = t & 255;
    Fe<ModQ, 2> acc = fe_one_mont<ModQ>().as<2>();
    bool zero = false;
    for (uint32_t j = 0; j < GKB_BITS; j++) {
        bool swap = V.gk_swap[j * V.C + p] != 0, set = (i >> j) & 1;
        if (swap) zero = zero || !set;
        else if (set) acc = acc * soa_ld<ModQ, 2>(V.gk_f, j * V.C + p);
    }
    Fe<ModQ, 1> c = zero ? fe_zero<ModQ>() : fe_canon(acc);
    uint32_t o[9];
    limbs_repack<30, 29, 9, 9>(o, c.l);
#pragma unroll
    for (int l = 0; l < 9; l++) csub[(size_t)t * 9 + l] = o[l];
}
template <bool UNI>
__global__ void __launch_bounds__(256) k_v_gk_block(uint32_t count, const uint32_t* __restrict__ E, const uint32_t* __restrict__ csub, uint32_t nblocks, Soa res) {
    uint32_t p, block;
    if (UNI) {
        uint32_t nbg = nblocks >> 8;
        p = blockIdx.x / nbg;
        block = (blockIdx.x % nbg) * 256 + threadIdx.x;
    } else {
        uint32_t slot = blockIdx.x * 256 + threadIdx.x;
        if (slot >= count * nblocks) return;
        p = slot / nblocks, block = slot % nblocks;
    }
    const uint32_t* a = csub + (size_t)p * GKB_SIZE * 9;
    const uint32_t* e = E + ((size_t)255 * 9) * nblocks + block;  // rank 255 = S {} of slice i: key[block * 256 + i]
    const size_t slice = (size_t)GKB_SIZE * 9 * nblocks;
    GkCols acc;
    gkc_zero(acc);
#pragma unroll 1
    for (uint32_t ib = 0; ib < GKB_SIZE; ib += 7) {
        uint32_t ie = ib + 7 < GKB_SIZE ? ib + 7 : GKB_SIZE;
#pragma unroll 1
        for (uint32_t i = ib; i < ie; i++) {
            uint32_t x[9], y[9];
#pragma unroll
            for (int l = 0; l < 9; l++) x[l] = a[i * 9 + l], y[l] = e[i * slice + (size_t)l * nblocks];
            gkc_mac(acc, x, y);
        }
        gkc_carry(acc);
    }
    uint32_t t29[18], t30[18];
#pragma unroll
    for (int i = 0; i < 18; i++) t29[i] = (uint32_t)acc.c[i];
    limbs_repack<29, 30, 18, 18>(t30, t29);
    soa_st(res, p * nblocks + block, fe_canon(redc_wide(t30)));
}
void launch_v_gk_block_stage(hipStream_t s, const VWork& V, const uint32_t* E, uint32_t nblocks, uint32_t count, uint32_t* csub, const Soa& res) {
    hipLaunchKernelGGL(k_v_gk_csub, dim3(count), dim3(256), 0, s, V, count, csub);
    if ((nblocks & 255) == 0) hipLaunchKernelGGL(k_v_gk_block<true>, dim3(count * (nblocks >> 8)), dim3(256), 0, s, count, E, csub, nblocks, res);
    else hipLaunchKernelGGL(k_v_gk_block<false>, dim3((uint32_t)(((uint64_t)count * nblocks + 255) / 256)), dim3(256), 0, s, count, E, csub, nblocks, res);
}
