// Internal header of libzkattest_hip.so: workspace layout, SoA helpers and the launch wrappers each .hip
// translation unit exports.  Nothing here is part of the C ABI (include/zkattest.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <cstdlib>
#include "../../include/zkattest.h"
#include "curve.h"
#include "rng.h"

// ------------------------------------------------------------------ SoA views (limb l of element e at p[l*stride+e])
struct Soa {
    uint32_t* p;
    uint32_t stride;
};
struct Soa3 {
    Soa x, y, z;
};
template <class M, int K = 2>
ZK_DEV Fe<M, K> soa_ld(const Soa& a, uint32_t e) {
    Fe<M, K> r;
#pragma unroll
    for (int l = 0; l < NLIMB; l++) r.l[l] = a.p[(size_t)l * a.stride + e];
    return r;
}
template <class M, int K>
ZK_DEV void soa_st(const Soa& a, uint32_t e, const Fe<M, K>& v) {
#pragma unroll
    for (int l = 0; l < NLIMB; l++) a.p[(size_t)l * a.stride + e] = v.l[l];
}

// ------------------------------------------------------------------ fixed-base tables
// Tom: W-bit comb windows, W chosen at run time (zk_ctx_set_comb_bits, 8..24): ceil(256/W) windows x 2^W digits,
// entry = niels (x, y, d'*x*y) Montgomery limbs in a 128-byte line (27 words used), so one gather touches exactly
// one L2/HBM line.  Measured per 104 M commitments (MI355X): W = 8/11/13/16 with 112-byte entries 361/278/232/201 ms;
// with 128-byte entries W = 16/20/22/24: 194/170/157/147 ms (tables 0.27/3.5/12.9/47 GB for the two bases).
#define TOM_ENTRY_WORDS 32
#define TOM_DEFAULT_BITS 16
#define TOM_MAX_BITS 26
// widths above 24 use SIGNED digits in (-2^(W-1), 2^(W-1)]: half the entries per window (a negative digit negates the
// entry on load) at the price of one spare bit: 25 = the 11 windows of 24 bits in 23.6 GB, 26 = 10 windows in 86 GB
__host__ __device__ static inline bool tom_signed(uint32_t bits) { return bits > 24; }
__host__ __device__ static inline uint32_t tom_nwin(uint32_t bits) { return ((tom_signed(bits) ? 257 : 256) + bits - 1) / bits; }
__host__ __device__ static inline uint32_t tom_win_entries(uint32_t bits) { return tom_signed(bits) ? (1u << (bits - 1)) + 1 : 1u << bits; }
static inline size_t tom_tab_words(uint32_t bits) { return (size_t)tom_nwin(bits) * tom_win_entries(bits) * TOM_ENTRY_WORDS; }
// P-256 fixed bases (G, h_NIST): PFIX_WIN_BITS-bit comb windows (default 20: 13 windows, 1.1 GB per base), entry = affine (x, y) Montgomery limbs,
// 20 words (80 B); digit 0 unused.
// batch normalisers: a thread walks `per` points strided by the thread count, a workgroup shares one Fermat inversion
// (block_inverse).  Few fat threads beat many thin ones when other lanes' kernels share the GPU (same-box A/B, proofs/s at
// per <= 32 / 64 / 128 / 256 with 524288 / 262144 / 131072 / 131072 threads at least: 292 / 297 / 300 / 292 k against 296 k for one
// inversion per thread of 256 points).
#ifndef ZK_NORM_PER_MAX
#define ZK_NORM_PER_MAX 128
#endif
#ifndef ZK_NORM_MIN_THREADS
#define ZK_NORM_MIN_THREADS 131072
#endif
#ifndef PFIX_WIN_BITS
#define PFIX_WIN_BITS 20
#endif
#define PFIX_NWIN ((256 + PFIX_WIN_BITS - 1) / PFIX_WIN_BITS)
#define PFIX_WIN_SIZE (1u << PFIX_WIN_BITS)
#define PFIX_ENTRY_WORDS 20
#define PFIX_TAB_WORDS ((size_t)PFIX_NWIN * PFIX_WIN_SIZE * PFIX_ENTRY_WORDS)
#include "ktab.h"   // layout of the per-key tables of the ring (KTAB_*) and the multiplication through them
size_t ktab_temp_bytes(uint64_t N, uint32_t slab_keys);
void launch_ktab_build(hipStream_t s, const Soa& ring, uint64_t N, uint32_t* ktab, uint8_t* ok, void* temp, uint32_t slab_keys);
// per-proof table of R (rtab.h): signed `bits`-bit comb, ceil(257/bits) windows x (2^(bits-1) + 1) entries of 28 words
#define RTAB_ENTRY_WORDS 28
#define RTAB_PROVE_BITS 6
#define RTAB_VERIFY_BITS 4
#define RTAB_MAX_NWIN 65
__host__ __device__ static inline uint32_t rtab_nwin(uint32_t bits) { return (257 + bits - 1) / bits; }
__host__ __device__ static inline uint32_t rtab_entries(uint32_t bits) { return (1u << (bits - 1)) + 1; }
__host__ __device__ static inline uint32_t rtab_words(uint32_t bits) { return rtab_nwin(bits) * rtab_entries(bits) * RTAB_ENTRY_WORDS; }

// ------------------------------------------------------------------ ZKA1 layout constants (bytes)
#define ZK_PB 32
#define ZK_TB 36
#define ZK_FIXED (ZK_HDR + 2 * 64 + 2 * 72)  // header, R, comS1, keyXcom, keyYcom = 304
#define ZK_REP_HEAD (64 + 72 + 72 + 4 * 32)    // 336
#define ZK_MULT_SZ (6 * 72 + 7 * 32)            // 656
#define ZK_EQ_SZ (2 * 72 + 3 * 32)              // 240
#define ZK_PADD_SZ (4 * 72 + 4 * ZK_MULT_SZ + 2 * ZK_EQ_SZ)  // 3392
#include "wire.h"   // struct Wire: the two byte layouts of a proof (ZKA1 / ZKA1P)
#define ZK_ST_T_INF_LATE 103  // internal: T_i = identity found by normalisation (resolved to a public status in k_scan)
#define ZK_MAXN 32

// list A: per proof 2 + 2*sec Tom commitments (pkX, pkY, Tx_i, Ty_i);  list B: per zero-bit rep 39 points
#define LB_SLOTS 39
#define LB_COMMITS 34

struct TomList {   // a list of Pedersen commitments to compute: (v, r) -> projective -> affine (original curve)
    Soa v, r;      // plain canonical scalars mod q
    Soa3 proj;     // a=1 image, Montgomery (X, Y, Z)
    Soa ax, ay;    // affine, original curve, plain canonical
    uint32_t cap;
};

struct DevParams {            // device-resident, built by zk_ctx_set_params
    uint32_t* tom_tab_g;      // tom_tab_words(tom_bits)
    uint32_t* tom_tab_h;
    uint32_t tom_bits;        // comb width of the two tables
    uint32_t* pfix_G;         // PFIX_TAB_WORDS
    uint32_t* pfix_H;
    uint32_t tom_g_aff[18];   // original-curve affine plain limbs of g (x, y) -- C_14 in pointAdd.ts:144
    uint32_t sec;
};

struct Workspace {
    uint32_t C;        // proofs per chunk
    uint32_t sec, n;   // secLevel, log2 ring
    uint32_t items_cap;
    // per proof
    int32_t* st;                 // [C] status
    Soa pkx, pky;                // plain canonical mod q (affine pk)
    Soa pkxm, pkym;              // Montgomery
    Soa Rxm, Rym;                // R affine Montgomery
    Soa Rx, Ry;                  // R affine plain
    Soa3 Q;                      // projective Montgomery
    Soa s1;                      // plain mod n
    // per-key tables (k_ktab.hip): multiples of every ring key, owned by the context; nullptr for rings above 2^KTAB_MAXN keys
    const uint32_t* ktab;        // [N][KTAB_NWIN][KTAB_ENT][16] affine Montgomery (x, y) as 2 x 8 words; slot d - 1 holds d * 2^(8 w) * key
    const uint8_t* ktab_ok;      // [N] 1: the ring value is the x-coordinate of a curve point and has a table
    uint32_t* kt_key;            // [C] ring index of the proof's key (in.which)
    uint8_t* kt_use;             // [C] 0: per-proof table of R (rtab.h); 1 / 2: pk = + / - the key table's base point
    Soa u1m, u2m;                // u1, u2 mod n in Montgomery form (proofs on the key-table path multiply their nonces into them)
    uint32_t* rtab;              // [C][rtab_words(bits)], sized for RTAB_PROVE_BITS
    Soa3 rbase;                  // [C*RTAB_MAX_NWIN] window bases 2^(bits w) R, JACOBIAN (X, Y < 34 q, Z < 10 q: k_rtab_base / k_rtab_fill)
    uint32_t* chal;              // [C][4] challenge words (80 bits in words 0..2)
    uint32_t* zcnt;              // [C]
    uint32_t* item_base;         // [C+1]
    uint64_t* out_base;          // [C+1] byte offsets inside this chunk's output region
    // exp commit: index proof*(sec+1)+j ; j = sec is comS1
    Soa3 Tproj, Aproj;
    Soa Tx, Ty, Ax, Ay;          // affine plain
    // items
    uint32_t* item_proof;        // [items_cap]
    uint32_t* item_rep;
    uint32_t* item_rank;
    Soa3 T1proj;
    Soa T1x, T1y;
    uint32_t* padd_c;            // [items_cap][6][3] challenges of pi8, pi10, pi11, pix, pi13, piy
    TomList la, lb, lc;
    // GK
    uint32_t* gk_x;              // [C][3]
    Soa gk_coef;                 // [(n+1)*C] final polynomial coefficients, index k*C + proof
    uint32_t gk_group;           // proofs per fold pass
    uint32_t* rng_fill;          // [C][nblk][8] the chunk's RNG fills as a stream (seed mode), see k_rng_prepass
    uint8_t* r_zero;             // [C] r = 0 mod n (k_front)
    uint8_t* exph_msg;           // [exph_cap][blocks * 64] the padded message of the Exp challenge (k_hash.hip: k_exph_*)
    uint32_t* exph_wk;           // [blocks][16][count] x uint4: its expanded schedule W_i + K_i, proof-fastest (the lanes of k_exph_rounds are consecutive proofs)
    uint32_t exph_cap;           // proofs these two hold: min(C, EXPH_MAXP)
    uint8_t* exph_big_msg;       // the same two for a prover chunk of any size, borrowed from list B (idle until stage 2): api.hip carve
    uint32_t* exph_big_wk;
    uint32_t exph_big_cap;       // proofs they hold (C, or 0 where list B is too small)
    uint32_t* gk_bufA;           // ping-pong level buffers
    uint32_t* gk_bufB;
    // block-transform path of the ring fold (k_gk.hip), used when the ring has a table E (9 <= n <= GK_ETAB_MAXN)
    const uint32_t* gk_etab;     // per-ring table, owned by the context (nullptr: plain fold)
    const int8_t* gk_kdig;       // the ring as int8 digit fragments for the verifier's matrix-pipe fold (k_gk_mfma.hip; nullptr: VALU fold)
    const int8_t* gk_edig;       // table E's coefficient classes 2..6 as int8 digit fragments (prover's matrix-pipe path; nullptr: VALU only)
    int8_t* gk_adig;             // [tiles][6 chunks][33 digits][1 KB] digit fragments of the chunk's a_S products, proofs sorted by l_low
    uint32_t* gk_toff;           // [257] first 16-proof tile of every l_low group
    uint32_t* gk_asub;           // [C][256][9] products of the a_j over the subsets of the 8 low bits
    uint32_t* gk_order;          // [C] proofs sorted by the 8 low bits of their ring index
    uint32_t* gk_goff;           // [257] group offsets of that order
    Soa ring;                    // [N] plain canonical limbs (shared, owned by ctx)
    uint32_t N;
    uint32_t hardened;           // zk_ctx_set_mode: the GK challenge also hashes the statement (ring digest, msgHash, R, keyXcom)
    const uint32_t* ring_digest; // [8] SHA-256 words of the padded ring (k_hash.hip: launch_ring_digest), owned by ctx
    Wire wire;                   // layout the prover's writers emit (zk_ctx_set_wire)
    RngCtx rng;
};

// The batched Tom-256 check sums the relations of contiguous groups of a chunk's proofs separately (same pass: the group index
// rides on top of the digit in the 19-bit sort key): a forged proof only sends ITS group to the per-proof sums.  8 groups with
// 16-bit windows or 64 groups with 13-bit windows (zk_ctx_set_verify_groups; k_msm.hip).
#define MSM_G_MAX 64
#define MSM_NW_MAX 20
// Table sums of a SMALL chunk run "wide": several lanes per sum, a range of windows each, partial sums added through the wave's cross-lane moves (rtab.h).  Up to
// this many sums per launch -- four lanes each fill one residency of the GPU at two to three register-heavy waves per SIMD; beyond it the one-lane kernels,
// whose lanes all work, are faster (a compile-time constant; same-box A/B and the size sweep in profiles/r05_ab_variants.txt (2)).
#ifndef ZK_WIDE_MAX_UNITS
#define ZK_WIDE_MAX_UNITS 32768u
#endif
#define V_SAMPLE_FILLS 256  // verifier: one-byte fills of the sampler hashed up front per proof (k_verify.hip: k_v_sample_fills; the walk needs ~106 of them)
#define EXPH_MAXP 256       // a verifier chunk of at most this many proofs hashes its Exp challenge on an auxiliary stream (api_verify.hip: small one-chunk calls)
#define V_SLOT_SPLIT 4      // slot accumulators per checked repetition (k_v_straus: a slot's terms over up to 4 lanes)
// per-proof sums of at most V_WIDE_MAXP proofs: ONE term per lane (a slot's 36 terms over 36 lanes, a membership group's 8 over 8), folded by
// k_v_acc_tree, the five sums side by side on the lane's auxiliary streams -- the chain of a lane is what a small batch waits for
#ifndef V_WIDE_MAXP
#define V_WIDE_MAXP 256
#endif
// chunks of at most V_SIDE_MAXP proofs leave the GPU mostly idle: the challenge hashes and the membership total run beside the P-256 front end, the P-256
// sums (one term per lane) beside the Tom-256 sums, on the lane's auxiliary streams
#ifndef V_SIDE_MAXP
#define V_SIDE_MAXP 8192
#endif
#define V_WIDE_GK 8         // terms (= lanes) per membership group
#define V_AUX_STREAMS 4
#define V_RECHECK 0x100u    // group flag: per-proof sums were computed, low byte = lanes per slot
#define V_FOLDED 0x200u     // ... and folded to ONE accumulator per proof for the slots and one for the membership groups (k_v_acc_tree; calls of a few proofs)
struct VGroupFlags {        // per group of a chunk: 1 = passed the batched check, V_RECHECK | tsplit otherwise (kernel argument of k_v_final)
    uint32_t v[MSM_G_MAX];
};
// ------------------------------------------------------------------ verifier workspace
#define VK 20             // reps checked by verifySignatureList (zkpAttestList.ts:177)
#define V_SLOT_TERMS 36   // 10 x 256-bit + 26 x 128-bit terms per checked rep
struct Soa4 {
    Soa x, y, z, t;
};
#define VT_ENTRY_WORDS 32 // a term's point: 128-byte entry (x', y, d'x'y: 27 limbs, Montgomery) -- what the bucket sums of k_msm.hip gather, written once by the kernels
                          // that parse the proofs (until round 5 they wrote three SoA arrays and k_msm_pack copied every live term: 1.6 ms per 32 768 proofs)
struct VTerms {           // terms of the "sum s_i P_i = identity" checks: niels point on the a=1 image + plain scalar
    uint32_t* pts;        // [cap][VT_ENTRY_WORDS]; the three lists of a workspace are consecutive, so a term's id in k_msm.hip is its entry number from slot_terms.pts on
    Soa sc;
    uint32_t* tab;        // window tables, AoS: multiples 1..8 of term idx, 36 words (X, Y, d'T, Z) each, at tab[(idx*8 + e)*36]
    uint8_t* dig;         // signed 4-bit digits of the scalars: dig[w * cap + idx] = |d| (0..8) | sign << 7, w = 0..64
    uint32_t cap;         // terms the arrays were carved for
};
struct VWork {
    uint32_t C, sec, n;
    uint32_t hardened;           // as in Workspace
    const uint32_t* ring_digest;
    int32_t* st;          // [C] structural status (deserialisation)
    int32_t* exp_st;      // [C] exceptions of verifyExp (k_v_exp_status; k_v_final folds exp_jm in)
    uint32_t* exp_jm;     // [C] first sampled slot whose response type differs from the recomputed challenge bit, VK: none (k_v_sample_check)
    uint32_t* exp_jz;     // [C] the sampled slot exp_st's identity was found at, VK: none
    uint32_t* okflags;    // [C] bit 3: GKProof length mismatch (verifyMembership returns false)
    uint32_t* zcnt;       // [C]
    uint32_t* hbits;      // [C][4] challenge bits of the header (layout)
    uint32_t* chal;       // [C][4] recomputed Exp challenge
    uint32_t* gkx;        // [C][3]
    uint32_t* idx;        // [C][VK] checked rep index | bit << 8
    uint32_t* t1_act;     // [2 C VK] list-A slots whose commitment T1x / T1y the verifier must compute (zero-bit sampled repetitions of good proofs), compacted
    uint32_t* t1_cnt;     // [1] their number (k_v_t1_scalars)
    uint8_t* vbytes;      // [C][1536] first byte of the verifier-RNG fills (k_v_sample_fills)
    uint32_t* vc;         // [C*VK][6][3] sub-proof challenges
    uint8_t* ph_msg;      // [min(C, V_PH_MAXP) * VK * 6][V_PH_BLOCKS * 64] the padded messages of those challenges in a call of a few proofs (k_v_padd_msg) ...
    uint32_t* ph_wk;      // ... their expanded schedules (k_hash.hip: launch_sha_msgs) ...
    uint8_t* ph_nblk;     // ... and how many blocks each has (10, 5, or 1 for a slot that has no PointAdd proof)
    TomList vd;           // [C*VK*5] derived commitments (proj + affine)
    Soa gk_f, gk_g;       // [n*C] rho_j = f_j/g_j and the level's scale factor g_j (Montgomery); see k_v_gk_fg
    uint32_t* gk_swap;    // [n*C] 1 where g_j = 0 (x = f_j): the level keeps the odd branch, scale f_j
    Soa gk_total;         // [C]
    uint32_t* gk_csub;    // [C][256][9] coefficients of the 8 low index bits (k_gk.hip, used when the ring has table E)
    VTerms slot_terms, gk_terms, misc_terms;
    uint8_t* slot_class;  // [C*VK] 1 = zero-bit slot of a good proof (36 live terms), 0 = only the two 128-bit terms 34, 35
    uint32_t* slot_perm;  // [C*VK] per re-checked proof range: local slot ids, class-1 slots from the front, the others from the back
    uint32_t* slot_cnt;   // [2 * (MSM_G_MAX + 1)] how many of each, per range
    Soa4 slot_acc, gk_acc, misc_acc;
    Soa4 wide_acc;        // [min(C, V_WIDE_MAXP) * (VK * 36 + nq * 8)] one accumulator per TERM of the slots, then of the membership groups (small batches)
    Soa sSg, sSh, sSkx, sSky, sSR, sSH, sSL;   // per slot partial sums
    Soa pSR, pSH, pSL;                         // per proof (mod n)
    Soa pa_x, pa_y, pa_sc;                     // P-256 A terms [C*VK]
    uint32_t* pa_tab;                          // [C*VK][8][28] multiples 1..8 of every A term (k_v_p256_straus)
    uint8_t* pa_dig;                           // [33][C*VK] signed 4-bit digits of the randomisers
    Soa3 pacc;                                 // [max(5 C, 22 min(C, V_SIDE_MAXP))] partial sums of k_v_p256_straus (5 lanes per proof, or 21 in a small chunk + the table walks' sum)
    Soa clx, cly;                              // Clambda (Montgomery affine)
    uint32_t* cl_tab;                          // [C][8][28] multiples 1..8 of Clambda
    uint8_t* cl_dig;                           // [35][C] signed 4-bit digits of SL
    uint32_t* p256_ok;                         // [C] the P-256 relation holds (k_v_p256_total)
};
// ZKA1P -> ZKA1 for `count` proofs from proof `first` on: uoff[0 .. count] = their offsets in `out`, starting at `base` (k_verify.hip)
void launch_offsets_monotonic(hipStream_t s, const uint64_t* d_off, uint64_t B, uint32_t* d_bad /* |= 1 if off[i] > off[i + 1] or off[i] is not 4-byte aligned */);
void launch_v_unpack(hipStream_t s, uint32_t sec, uint32_t count, const uint8_t* packed, const uint64_t* poff, uint64_t first, uint64_t base, uint8_t* out, uint64_t* uoff);
void launch_v_header_validate(hipStream_t s, const VWork& V, uint32_t count, const uint8_t* proofs, const uint64_t* off, uint64_t first);
void launch_v_front_r(hipStream_t s, const Workspace& W, const VWork& V, uint32_t count, const uint8_t* proofs, const uint64_t* off, uint64_t first);   // R (and W.st)
void launch_v_front_q(hipStream_t s, const DevParams& P, const Workspace& W, const VWork& V, uint32_t count, const uint8_t* proofs, const uint64_t* off, const uint8_t* msg, uint64_t first);   // Q, Clambda
// parts: 1 = the Exp challenge, 2 = the Groth-Kohlweiss challenge, 3 = both (one lane per proof each); the Exp challenge of a small chunk alternatively
// through the three-kernel path (message from the proof bytes, then launch_exph_hash into V.chal)
void launch_v_challenges(hipStream_t s, const VWork& V, uint32_t count, const uint8_t* proofs, const uint64_t* off, const uint8_t* msg, uint64_t first, uint32_t parts);
void launch_v_exp_challenge_small(hipStream_t s, const Workspace& W, const VWork& V, uint32_t count, const uint8_t* proofs, const uint64_t* off, uint64_t first);
void launch_v_sample(hipStream_t s, const VWork& V, uint32_t count, const uint8_t* vseeds, uint64_t first);
void launch_v_sample_check(hipStream_t s, const VWork& V, uint32_t count);
void launch_v_exp_status(hipStream_t s, const Workspace& W, const VWork& V, uint32_t count, bool have_jm);
#define V_PH_MAXP 64      // proofs per call whose PointAdd challenges go through the message / schedule / two-lane rounds kernels
#define V_PH_BLOCKS 10    // 9 points of 67 bytes + padding
// k_hash.hip: `count` messages of up to nblk 64-byte blocks at W.exph_msg (message p at p * nblk * 64; nblk_of[p] blocks if given) -> challenge words chal[ostride p ..]
void launch_sha_msgs(hipStream_t s, const Workspace& W, uint32_t count, uint32_t* chal, uint32_t nblk, uint32_t ostride, const uint8_t* nblk_of = nullptr);
void launch_exph_hash(hipStream_t s, const Workspace& W, uint32_t count, uint32_t* chal);   // k_hash.hip: schedule per block, rounds per proof -> chal[4 p ..]
void launch_v_exp_points(hipStream_t s, const Workspace& W, const VWork& V, uint32_t count, const uint8_t* proofs, const uint64_t* off, uint64_t first, uint32_t split = 1);   // split: 1, or 4 lanes per checked repetition (small chunks)
void launch_v_t1_scalars(hipStream_t s, const Workspace& W, const VWork& V, uint32_t count, const uint8_t* proofs, const uint64_t* off, uint64_t first);
void launch_v_derived(hipStream_t s, const Workspace& W, const VWork& V, uint32_t count, const uint8_t* proofs, const uint64_t* off, uint64_t first);
void launch_v_padd_hash(hipStream_t s, const DevParams& P, const Workspace& W, const VWork& V, uint32_t count, const uint8_t* proofs, const uint64_t* off, uint64_t first);
void launch_v_gk_total(hipStream_t s, const VWork& V, const Soa& ring, const uint32_t* etab, const int8_t* kdig, uint32_t count, uint32_t N, const uint8_t* proofs, const uint64_t* off, uint64_t first, const Soa& res, const Soa& res2);
// k_gk_mfma.hip: the 8 low index bits of the verifier's ring fold as int8 matrix products (rings of at least 2^12 keys)
#define GKM_MINN 12
size_t gkm_ring_frag_bytes(uint64_t N);
size_t gkm_coef_frag_bytes(uint32_t C);
void launch_gkm_ring_digits(hipStream_t s, const Soa& ring, uint32_t nblocks, int8_t* frag);
void launch_v_gk_block_mfma(hipStream_t s, const VWork& V, const int8_t* ring_frag, uint32_t nblocks, uint32_t count, int8_t* coef_frag, const Soa& res);
// prover: coefficients 2..6 of a block's polynomial (238 of its 255 products a_S * D_S) as int8 matrix products
struct ChunkIn;
size_t gkm_etab_frag_bytes(uint64_t N);
size_t gkm_asub_frag_bytes(uint32_t C);
void launch_gkm_etab_digits(hipStream_t s, const uint32_t* E, uint32_t nblocks, int8_t* edig);
void launch_gk_block_mfma(hipStream_t s, const Workspace& W, const ChunkIn& in, uint32_t nblocks, const Soa& res);
void launch_v_slot_points(hipStream_t s, const VWork& V, uint32_t count, const uint8_t* proofs, const uint64_t* off, uint64_t first);
void launch_v_slot_terms(hipStream_t s, const Workspace& W, const VWork& V, uint32_t count, const uint8_t* proofs, const uint64_t* off, const uint8_t* vseeds, uint64_t first);
void launch_v_proof_points(hipStream_t s, const VWork& V, uint32_t count, const uint8_t* proofs, const uint64_t* off, uint64_t first, uint32_t which);
void launch_v_proof_terms(hipStream_t s, const Workspace& W, const VWork& V, uint32_t count, const uint8_t* proofs, const uint64_t* off, const uint8_t* vseeds, uint64_t first);
void launch_v_proof_sums(hipStream_t s, const Workspace& W, const VWork& V, uint32_t count);
void launch_v_straus(hipStream_t s, const VTerms& L, uint32_t ngroups, uint32_t ng_stride, uint32_t n256, uint32_t n128, const Soa4& out,
                     const uint32_t* perm, const uint32_t* cnt, uint32_t tsplit = 1, uint32_t ostride = 1, uint32_t ny = 1, uint32_t ystride = 0);
void launch_v_acc_tree(hipStream_t s, const Soa4& src, uint32_t nouter, uint32_t width, const Soa4& dst, uint32_t dstride, uint32_t fill);
// k_coop.hip: the same sums on cooperating waves (one wave per chain), chosen by the launch wrappers when a launch has at most ZK_COOP_MAX_CHAINS chains
#ifndef ZK_COOP_MAX_CHAINS
#define ZK_COOP_MAX_CHAINS 16384u
#endif
// ZKATTEST_ONE_LANE_CHAINS (any value): every dependent chain stays in one lane (the kernels of round 5) -- the A/B switch behind profiles/r06_ab_variants.txt
extern std::atomic<uint64_t> g_coop_chains;   // k_coop.hip: chains handed to cooperating waves by this process so far (zk_test_counter 4; tests only)
static inline bool zk_one_lane_chains() {
    static const bool v = getenv("ZKATTEST_ONE_LANE_CHAINS") != nullptr;
    return v;
}
void launch_v_straus_co(hipStream_t s, const VTerms& L, uint32_t ngroups, uint32_t ng_stride, uint32_t n256, uint32_t n128, const Soa4& out, const uint32_t* perm,
                        const uint32_t* cnt, uint32_t tsplit, uint32_t ostride, uint32_t ny, uint32_t ystride);
void launch_v_p256_straus_co(hipStream_t s, const VWork& V, uint32_t count);
void launch_v_exp_points_co(hipStream_t s, const Workspace& W, const VWork& V, uint32_t count, const uint8_t* proofs, const uint64_t* off, uint64_t first);
void launch_rtab_base_co(hipStream_t s, const Workspace& W, uint32_t count, uint32_t bits, const uint8_t* skip);
void launch_v_slot_perm(hipStream_t s, const uint8_t* slot_class, uint32_t nslots, uint32_t* perm, uint32_t* cnt);
void launch_v_p256_straus(hipStream_t s, const VWork& V, uint32_t count, uint32_t per);   // per = A terms per lane: 5, or 1 for small batches
void launch_v_p256_total(hipStream_t s, const DevParams& P, const Workspace& W, const VWork& V, uint32_t count, uint32_t per);
void launch_v_p256_total_fixed(hipStream_t s, const DevParams& P, const Workspace& W, const VWork& V, uint32_t count);
void launch_v_p256_total_sum(hipStream_t s, const DevParams& P, const Workspace& W, const VWork& V, uint32_t count);
void launch_v_final(hipStream_t s, const Workspace& W, const VWork& V, uint32_t count, uint8_t* ok, int32_t* status, uint64_t first, const VGroupFlags& gf, uint32_t gsz);

// chunk inputs (device pointers, already offset to the chunk's first proof)
struct ChunkIn {
    const uint8_t* msg;
    const uint8_t* sig;
    const uint8_t* pk;
    const uint32_t* which;
    uint32_t count;   // proofs in this chunk
};

// buffers of the batched Tom-256 check (k_msm.hip), one set per verifier lane
struct MsmBuf {
    uint32_t cap;          // term ids
    uint32_t* aos;         // [cap][32] niels entries of the live terms
    uint2* pairs;          // [windows][cap] (key = group << C | digit, term id) of the live terms' non-zero digits, partitioned by bin (k_msm_scatter)
    uint32_t* vals_out;    // [windows][cap] term ids grouped by key (k_msm_binsort)
    uint32_t *start, *end; // [windows][2^19] segment of every (group, digit) value
    uint32_t* ord_id;      // [windows * 2^19] buckets in the order k_msm_bucket's lanes take them: by size within each bin
    uint32_t *counters, *flag;   // counters[0]: live terms, counters[32]: oversized buckets
    uint32_t* big_list;    // [4096] window * 65536 + digit of the oversized buckets
    uint32_t* big_part;    // [4096][128][36] partial sums of their slices
    uint32_t* buckets;     // [windows][2^19][36]
    uint32_t* red;         // the levels of the bucket reduction (k_msm.hip: launch_msm_reduce), msm_red_words(groups) words
    uint32_t* Tw;          // [windows * groups][36] sum_d d * B_d, times 2^(C w)
    void* sort_tmp;
    size_t sort_tmp_bytes;
    TomList one;           // [groups] the groups' fixed-base commitments
    uint32_t* host;        // page-locked host words (live-term count, group verdicts): a pageable destination makes the runtime
                           // wait for every stream of the device, i.e. for the other lanes' kernels
};
// buffers of the cross-proof P-256 pass (k_pmsm.hip)
#define PM_NW_MAX 14
struct PMsmBuf {
    uint32_t ncap, gcap;       // term ids (C * 21); ids per (window, group) list
    uint32_t* aos;             // [ncap][16] affine entries (ktab.h format)
    uint16_t* dig;             // [windows][ncap] digits
    uint32_t* vals;            // [windows * groups][gcap] term ids grouped by digit
    uint32_t *start, *end;     // [windows * groups * 2^C] a bucket's segment of its (window, group) list
    uint32_t* order;           // [windows * groups * 2^C] buckets of a (window, group), largest first
    uint32_t* buckets;         // [windows * groups * 2^C][28]
    uint32_t* big_list;        // oversized buckets
    uint32_t* Tw;              // [windows * groups][28]
    uint32_t *rpart, *shpart;  // per 64 proofs of a group: sum of SR * R, sum of SH
    uint32_t *counters, *flag; // counters[1]: a scalar beyond the windows, counters[2]: oversized buckets; flag[g]
};
// ------------------------------------------------------------------ launch wrappers (one per kernel family)
// k_pmsm.hip: carve (base == nullptr: size only); the pass in two parts (no host round trip of its own: host_flags_pinned[g] is valid once the stream of
// pmsm_sums has drained)
size_t pmsm_carve(PMsmBuf* M, uint8_t* base, uint32_t Ccap, uint32_t groups);
void pmsm_prepare(hipStream_t s, const Workspace& W, const VWork& V, uint32_t count, const PMsmBuf& M, uint32_t groups);
void pmsm_sums(hipStream_t s, const DevParams& P, uint32_t count, const PMsmBuf& M, uint32_t groups, uint32_t* host_flags_pinned);
void launch_pm_all_ok(hipStream_t s, const VWork& V, uint32_t count);
// k_msm.hip
size_t msm_workspace_bytes(uint32_t cap);
size_t msm_red_words(uint32_t groups);
// run_msm enqueues the pass on s (no host round trip); once s has drained that far, msm_read_flags gives host_flags[g] = 1: the Tom-256 total of group g (proofs
// [g * gsz, (g + 1) * gsz) of the chunk) is the identity; M.flag holds the same on the device, M.host[0] the live terms of the pass
hipError_t run_msm(hipStream_t s, const DevParams& P, const Workspace& W, const VWork& V, uint32_t count, uint32_t nq, const MsmBuf& M, uint32_t groups /*8 or 64*/,
                   uint32_t* gsz, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr /* recorded around k_msm_bucket */,
                   hipEvent_t ev2 = nullptr, hipEvent_t ev3 = nullptr /* ... around the grouping of the keys */);
void msm_read_flags(const MsmBuf& M, uint32_t groups, uint32_t* host_flags /*[groups]*/);
// group size of one k_gk_finish pass over `ntiles` polynomials of T+1 coefficients: <= 64, dynamic LDS below 60 KB
static inline size_t gk_finish_lds(uint32_t T, uint32_t g) { return sizeof(uint32_t) * 9 * ((size_t)g * (T + 1) + (size_t)(g / 2) * (T + 2)); }
static inline uint32_t gk_finish_gsz(uint32_t T, uint32_t ntiles) {
    uint32_t g = ntiles < 64 ? ntiles : 64;
    while (g > 2 && gk_finish_lds(T, g) > 60 * 1024) g >>= 1;
    return g;
}
// k_gk.hip
#define GK_ETAB_MINN 9
#define GK_ETAB_MAXN 20
size_t gk_etab_words(uint64_t N);
void launch_gk_etab(hipStream_t s, const Soa& ring, uint32_t nblocks, uint32_t* E);
void launch_gk_block_stage(hipStream_t s, const Workspace& W, const ChunkIn& in, const Soa& am, const Soa& res);
struct VWork;
void launch_v_gk_block_stage(hipStream_t s, const VWork& V, const uint32_t* E, uint32_t nblocks, uint32_t count, uint32_t* csub, const Soa& res);
// k_tables.hip
void launch_build_tom_table(hipStream_t s, const uint32_t* aff_xy_words /*18 words on device*/, uint32_t bits, uint32_t* tab, uint32_t* scratch, int32_t* ok);
size_t tom_table_scratch_words(uint32_t bits);
void launch_build_pfix_table(hipStream_t s, const uint32_t* aff_xy_words /*16 words on device, or nullptr for G*/, uint32_t* tab, uint32_t* scratch, int32_t* ok);
size_t pfix_table_scratch_words();
// k_tom.hip
void launch_tom_commit(hipStream_t s, const DevParams& P, const TomList& L, uint32_t count, uint32_t per_group, uint32_t slots_per_group, uint32_t kstride = 0);
// the commitments of the listed slots only (`list[i]`, i < *count_dev <= max_count: the verifier's T1x / T1y exist for zero-bit repetitions only)
void launch_tom_commit_list(hipStream_t s, const DevParams& P, const TomList& L, const uint32_t* list, const uint32_t* count_dev, uint32_t max_count);
void launch_tom_commit_listb(hipStream_t s, const DevParams& P, const TomList& L, uint32_t items, uint32_t kstride);  // the 34 commitments of every PointAdd item
void launch_tom_normalize(hipStream_t s, const TomList& L, uint32_t count, uint32_t first, uint32_t per_group, uint32_t slots_per_group, uint32_t kstride = 0);
void launch_padd_derived(hipStream_t s, const Workspace& W, uint32_t items);
// k_p256.hip
void launch_front(hipStream_t s, const DevParams& P, const Workspace& W, const ChunkIn& in);
void launch_rtab(hipStream_t s, const Workspace& W, uint32_t count, uint32_t bits, const uint8_t* skip = nullptr /*[count] != 0: no table for that proof*/);
void launch_exp_commit(hipStream_t s, const DevParams& P, const Workspace& W, uint32_t count);
void launch_p256_normalize(hipStream_t s, const Soa3& proj, uint32_t count, const Soa& ox, const Soa& oy, int32_t* st, uint32_t per_proof, int32_t err_code, const uint32_t* owner /*nullable: item->proof*/);
void launch_t1(hipStream_t s, const Workspace& W, uint32_t items);
void launch_test_pfix(hipStream_t s, const uint32_t* tab, uint64_t count, const uint8_t* k_be, uint8_t* out);
// k_hash.hip
void launch_rng_prepass(hipStream_t s, const Workspace& W, uint32_t count, uint32_t blk0, uint32_t blk1, uint32_t stride, uint32_t* fill, bool by_zcnt);
void launch_exp_challenge(hipStream_t s, const Workspace& W, uint32_t count);
void launch_padd_hash(hipStream_t s, const DevParams& P, const Workspace& W, uint32_t items);
void launch_gk_hash(hipStream_t s, const Workspace& W, uint32_t count, const uint8_t* msg);
void launch_ring_digest(hipStream_t s, const Soa& ring, uint64_t N, uint32_t* leaf_words /*[8 * ceil(N/256)]*/, uint32_t* digest8);
void launch_test_sha256(hipStream_t s, uint64_t count, uint64_t len, const uint8_t* d_msgs, uint8_t* d_out);
void launch_test_rng(hipStream_t s, const RngCtx& g, uint64_t B, uint32_t first_k, uint32_t n_k, uint8_t* d_out);
// k_scalar.hip
void launch_lista_scalars(hipStream_t s, const Workspace& W, uint32_t count);
void launch_scan(hipStream_t s, const Workspace& W, uint32_t count, uint64_t cursor, uint64_t out_cap, uint64_t* d_out_off, int32_t* d_status_out,
                 uint32_t* d_totals /*[4]: items, overflow, bytes lo, bytes hi*/, uint64_t first_proof);
void launch_words_to_host(hipStream_t s, void* dst_pinned, const void* src_dev, size_t nwords);   // read-back without the DMA engine
void launch_status_out(hipStream_t s, const Workspace& W, uint32_t count, int32_t* d_status_out, uint64_t first_proof);
void launch_items(hipStream_t s, const Workspace& W, uint32_t count);
void launch_padd_scalars(hipStream_t s, const DevParams& P, const Workspace& W, uint32_t items);
void launch_padd_respond(hipStream_t s, const Workspace& W, uint32_t items, uint8_t* out);
void launch_write_fixed(hipStream_t s, const Workspace& W, uint32_t count, uint8_t* out);
void launch_write_padd_points(hipStream_t s, const Workspace& W, uint32_t items, uint8_t* out);
void launch_gk_scalars_fold(hipStream_t s, const Workspace& W, const ChunkIn& in, const Soa& am /*[n*C] a_j, Montgomery*/);
void launch_gk_cd_scalars(hipStream_t s, const Workspace& W, uint32_t count);
void launch_gk_respond(hipStream_t s, const Workspace& W, const ChunkIn& in, uint8_t* out);
void launch_test_field(hipStream_t s, int which, int op, uint64_t count, const uint8_t* a, const uint8_t* b, uint8_t* out);
void launch_ring_load(hipStream_t s, const uint8_t* d_keys_be32, uint64_t nkeys, uint64_t N, const Soa& ring);
void launch_keys_to_ints(hipStream_t s, const uint8_t* d_pk, uint64_t count, uint8_t* d_out, int32_t* d_st);
void launch_bytes_to_scalars(hipStream_t s, const uint8_t* d_be32, uint64_t count, const Soa& out);
void launch_affine_to_bytes(hipStream_t s, const Soa& ax, const Soa& ay, uint64_t count, int tom, uint8_t* d_out);

// Montgomery's trick across a workgroup of 256 threads: every thread brings `acc`, the product of its own elements (Montgomery
// domain, non-zero), and receives 1/acc.  Inclusive prefix and suffix products by a doubling scan through LDS (8 rounds, 2 products
// each), ONE Fermat inversion per workgroup (thread 0), 1/acc_t = (1/total) * prefix_{t-1} * suffix_{t+1}.  The batch normalisers
// use it so that the number of points per inversion (256 x per) no longer dictates how few threads a launch has.
// lds: 2 * 256 * NLIMB words, limb-major.  Every thread of the workgroup must call it.
#if !defined(ZK_HOST_BUILD)
template <class M>
__device__ inline Fe<M, 2> block_inverse(const Fe<M, 2>& acc, uint32_t* lds) {
    const uint32_t lt = threadIdx.x;
    uint32_t *P = lds, *S = lds + 256 * NLIMB;
    auto put = [&](uint32_t* a, uint32_t i, const Fe<M, 2>& v) {
#pragma unroll
        for (int l = 0; l < NLIMB; l++) a[l * 256 + i] = v.l[l];
    };
    auto get = [&](const uint32_t* a, uint32_t i) {
        Fe<M, 2> v;
#pragma unroll
        for (int l = 0; l < NLIMB; l++) v.l[l] = a[l * 256 + i];
        return v;
    };
    Fe<M, 2> pre = acc, suf = acc;
#pragma unroll 1
    for (uint32_t d = 1; d < 256; d <<= 1) {
        put(P, lt, pre), put(S, lt, suf);
        __syncthreads();
        if (lt >= d) pre = get(P, lt - d) * pre;
        if (lt + d < 256) suf = get(S, lt + d) * suf;
        __syncthreads();
    }
    put(P, lt, pre), put(S, lt, suf);
    __syncthreads();
    Fe<M, 2> total = get(P, 255);
    Fe<M, 2> left = lt > 0 ? get(P, lt - 1) : fe_one_mont<M>().template as<2>();
    Fe<M, 2> right = lt < 255 ? get(S, lt + 1) : fe_one_mont<M>().template as<2>();
    __syncthreads();
    if (lt == 0) put(P, 0, fe_inv<M>(total));
    __syncthreads();
    return (get(P, 0) * left) * right;
}
#endif

// bit repacking between limb widths (all indices and shifts are compile-time constants after unrolling)
template <int IN_BITS, int OUT_BITS, int NIN, int NOUT>
ZK_DEV void limbs_repack(uint32_t* out, const uint32_t* in) {
    uint64_t buf = 0;
    int nb = 0, oi = 0;
#pragma unroll
    for (int i = 0; i < NIN; i++) {
        buf |= (uint64_t)in[i] << nb;
        nb += IN_BITS;
        if (nb >= OUT_BITS && oi < NOUT) {
            out[oi++] = (uint32_t)buf & ((1u << OUT_BITS) - 1);
            buf >>= OUT_BITS;
            nb -= OUT_BITS;
        }
        if (nb >= OUT_BITS && oi < NOUT) {
            out[oi++] = (uint32_t)buf & ((1u << OUT_BITS) - 1);
            buf >>= OUT_BITS;
            nb -= OUT_BITS;
        }
    }
#pragma unroll
    for (int k = 0; k < NOUT; k++)
        if (k >= oi) {
            out[k] = (uint32_t)buf & ((1u << OUT_BITS) - 1);
            buf >>= OUT_BITS;
        }
}
// Montgomery reduction of an 18-limb radix-2^30 integer T < q * 2^270: returns T / 2^270 mod q, < 2q.
ZK_DEV Fe<ModQ, 2> redc_wide(const uint32_t T[2 * NLIMB]) {
    uint64_t acc = 0;
    uint32_t m[NLIMB];
    Fe<ModQ, 2> r;
#pragma unroll
    for (int k = 0; k < NLIMB; k++) {
        acc += T[k];
#pragma unroll
        for (int i = 0; i < k; i++) acc = mad64(m[i], ModQ::mod[k - i], acc);
        m[k] = ((uint32_t)acc * ModQ::n0) & LIMB_MASK;
        acc = mad64(m[k], ModQ::mod[0], acc);
        acc >>= LIMB_BITS;
    }
#pragma unroll
    for (int k = NLIMB; k < 2 * NLIMB; k++) {
        acc += T[k];
#pragma unroll
        for (int i = k - (NLIMB - 1); i < NLIMB; i++) acc = mad64(m[i], ModQ::mod[k - i], acc);
        if (k < 2 * NLIMB - 1) {
            r.l[k - NLIMB] = (uint32_t)acc & LIMB_MASK;
            acc >>= LIMB_BITS;
        } else {
            r.l[NLIMB - 1] = (uint32_t)acc;
        }
    }
    return r;
}

// list B is item-fastest: slot k of item i lives at k * items_cap + i (coalesced for every per-item kernel)
ZK_DEV uint32_t lbi(const Workspace& W, uint32_t item, uint32_t k) { return k * W.items_cap + item; }

// ------------------------------------------------------------------ small device helpers shared by TUs
ZK_DEV uint32_t gtid() { return blockIdx.x * blockDim.x + threadIdx.x; }
// exclusive prefix sum over the workgroup (blockDim.x a multiple of 64, at most 1024); tot = the workgroup's total.  sh: 17 words of LDS.
ZK_DEV uint32_t block_excl_scan(uint32_t v, uint32_t* sh, uint32_t& tot) {
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o, 64);
        if (lane >= (uint32_t)o) inc += t;
    }
    if (lane == 63) sh[wv] = inc;
    __syncthreads();
    if (wv == 0) {
        const uint32_t sv = lane < nwv ? sh[lane] : 0;
        uint32_t si = sv;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            const uint32_t t = __shfl_up(si, o, 64);
            if (lane >= (uint32_t)o) si += t;
        }
        if (lane < nwv) sh[lane] = si - sv;
        if (lane == nwv - 1) sh[16] = si;
    }
    __syncthreads();
    const uint32_t r = sh[wv] + inc - v;
    tot = sh[16];
    __syncthreads();
    return r;
}
// big-endian load of a 32-byte integer into 8 little-endian words
ZK_DEV void load_be32(const uint8_t* p, uint32_t w[8]) {
    const uint32_t* q = (const uint32_t*)p;
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = bswap32(q[7 - i]);
}
// store NW little-endian words as a big-endian byte string (4-byte aligned destination)
template <int NW>
ZK_DEV void store_be(uint8_t* p, const uint32_t w[NW]) {
    uint32_t* q = (uint32_t*)p;
#pragma unroll
    for (int i = 0; i < NW; i++) q[i] = bswap32(w[NW - 1 - i]);
}
template <class M>
ZK_DEV void store_scalar_be(uint8_t* p, const Fe<M, 1>& a) {  // canonical plain -> 32 bytes
    uint32_t w[8];
    words_from_limbs<8>(w, a.l);
    store_be<8>(p, w);
}
ZK_DEV void store_tomcoord_be(uint8_t* p, const Fe<ModT, 1>& a) {  // canonical plain -> 36 bytes
    uint32_t w[9];
    words_from_limbs<9>(w, a.l);
    store_be<9>(p, w);
}
// shift a 256-bit little-endian word array right by SH bits
template <int SH>
ZK_DEV void shr256(uint32_t w[8]) {
#pragma unroll
    for (int i = 0; i < 7; i++) w[i] = (w[i] >> SH) | (w[i + 1] << (32 - SH));
    w[7] >>= SH;
}
// offset of rep i inside a proof, given the challenge (bit = 1 -> short response)
ZK_DEV uint32_t zeros_below(const uint32_t* chal, uint32_t i) {  // number of 0 bits among challenge bits [0, i)
    uint32_t ones = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        uint32_t lo = 32 * w;
        if (i > lo) {
            uint32_t nb = i - lo >= 32 ? 32 : i - lo;
            uint32_t mask = nb == 32 ? 0xffffffffu : ((1u << nb) - 1);
            ones += __popc(chal[w] & mask);
        }
    }
    return i - ones;
}
// Exp challenge of a small chunk in three kernels (k_hash.hip: k_exph_*): the padded message of hashPoints(Px, Py, A_0, Tx_0, Ty_0, ...) -- two Tom
// points (67 bytes each), then per repetition a P-256 point (65) and two Tom points -- and its length in 64-byte blocks
ZK_DEV uint32_t exph_msg_bytes(uint32_t sec) { return 2 * 67 + sec * (65 + 2 * 67); }
ZK_DEV uint32_t exph_blocks(uint32_t sec) { return (exph_msg_bytes(sec) + 9 + 63) / 64; }
ZK_DEV uint32_t exph_elem_offset(uint32_t e) { return e < 2 ? 67 * e : 134 + 199 * ((e - 2) / 3) + ((e - 2) % 3 == 0 ? 0 : (e - 2) % 3 == 1 ? 65 : 132); }
ZK_DEV void exph_put_padding(uint8_t* m, uint32_t sec) {   // 0x80, zeros, the bit length in eight bytes
    const uint32_t len = exph_msg_bytes(sec), end = exph_blocks(sec) * 64;
    m[len] = 0x80;
    for (uint32_t i = len + 1; i < end - 8; i++) m[i] = 0;
    const uint64_t bits = (uint64_t)len * 8;
    for (int i = 0; i < 8; i++) m[end - 1 - i] = (uint8_t)(bits >> (8 * i));
}
ZK_DEV uint64_t rep_offset(const uint32_t* chal, uint32_t i) { return ZK_FIXED + (uint64_t)ZK_REP_HEAD * i + (uint64_t)ZK_PADD_SZ * zeros_below(chal, i); }
ZK_DEV uint64_t rep_offset_w(const Wire& w, const uint32_t* chal, uint32_t i) { return w.fixed + (uint64_t)w.rep_head * i + (uint64_t)w.padd * zeros_below(chal, i); }
// Two Tom points (x0, y0, x1, y1: canonical plain) as 4 x 33 big-endian bytes = 33 dwords at a 4-byte aligned address (ZKA1P).  Byte k of a
// coordinate's string: k = 0 the top byte (bits 256..263), then the eight words big-endian; all shifts are constants after unrolling.
ZK_DEV uint32_t tomc_byte(const uint32_t w[9], int k) { return k == 0 ? (w[8] & 0xffu) : ((w[8 - ((k - 1) / 4 + 1)] >> (8 * (3 - (k - 1) % 4))) & 0xffu); }
ZK_DEV void store_tom_pair_packed(uint8_t* p4, const Fe<ModT, 1>& x0, const Fe<ModT, 1>& y0, const Fe<ModT, 1>& x1, const Fe<ModT, 1>& y1) {
    uint32_t w[4][9];
    words_from_limbs<9>(w[0], x0.l), words_from_limbs<9>(w[1], y0.l), words_from_limbs<9>(w[2], x1.l), words_from_limbs<9>(w[3], y1.l);
    uint32_t* q = (uint32_t*)p4;
#pragma unroll
    for (int d = 0; d < 33; d++) {
        uint32_t v = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int pos = 4 * d + b;
            v |= tomc_byte(w[pos / 33], pos % 33) << (8 * b);
        }
        q[d] = v;
    }
}
