/*
 * zkattest.h -- C ABI of libzkattest_hip.so, the MI355X (gfx950) engine behind the reference's
 * proveSignatureList / verifySignatureList / generateParamsList API.
 *
 * The reference (cloudflare/zkp-ecdsa, TypeScript) has no FFI layer; its boundary for this path is the ESM
 * surface of src/index.ts:17-19.  Each entry point below names the reference interface it replaces.  A thin
 * N-API addon or the Python mirror in zkp-ecdsa_amd/ binds these symbols.
 *
 * Conventions: caller allocates every buffer; the library never frees caller memory; a context is not
 * re-entrant (one in-flight batch per ctx); several contexts (one per GPU / process) may coexist.  All
 * integers inside byte buffers are big-endian, like the reference's toBytes (src/bignum/big.ts:121-134).
 *
 * Byte formats
 *   P-256 point   64 B  x(32) || y(32)                       (affine; src/curves/weier.ts:244-255 without 0x04)
 *   Tom-256 point 72 B  x(36) || y(36)  zero-padded from the reference's 33-byte coordinates so that every
 *                                         field is 4-byte aligned (src/curves/edwards.ts:194-203)
 *   scalar        32 B
 *   proof         "ZKA1" layout, the binary equivalent of SignatureProofList (src/zkpAttestList.ts:27-60):
 *     header 32 B : "ZKA1" | total_len u32 | secLevel u32 | n = log2(padded ring) u32 | challenge bits (16 B)
 *     R (P), comS1 (P), keyXcom (T), keyYcom (T)
 *     expProof[secLevel], each: A (P), Tx (T), Ty (T), then
 *        challenge bit 1: alpha, beta1 (mod n), beta2, beta3 (mod q)                     (src/exp/exp.ts:171-185)
 *        challenge bit 0: z, z2 (mod n), r1, r2 (mod q), PointAddProof                   (src/exp/exp.ts:186-226)
 *     PointAddProof: C_8, C_10, C_11, C_13 (T), pi_8, pi_10, pi_11, pi_13 (MultProof), pi_x, pi_y (EqualityProof)
 *        MultProof: C_4, A_x, A_y, A_z, A_4_1, A_4_2 (T), t_x, t_y, t_z, t_rx, t_ry, t_rz, t_r4   (src/commit/mult.ts:26-52)
 *        EqualityProof: A_1, A_2 (T), t_x, t_r1, t_r2                                              (src/commit/equality.ts:27-40)
 *     GKProof: cl[n], ca[n], cb[n], cd[n] (T), f[n], za[n], zb[n], zd                              (src/proofGK/gk.ts:31-58)
 *
 * Randomness contract (replaces crypto.getRandomValues in src/bignum/big.ts:171-181): the k-th 32-byte fill
 * of proof b is SHA-256(seed_b || be64(k)) (ZK_RNG_SEED) or block k of a caller-supplied stream
 * (ZK_RNG_STREAM); fills are consumed in the reference's draw order.
 */
#ifndef ZKATTEST_H
#define ZKATTEST_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct zk_ctx zk_ctx;
typedef int32_t zk_status;

/* call-level and per-proof status codes; the per-proof ones mirror the reference's thrown errors */
enum {
    ZK_OK = 0,
    ZK_E_POINT_NOT_IN_GROUP = 1, /* 'point not in group'            src/curves/weier.ts:83 */
    ZK_E_INVALID_KEY = 2,        /* 'invalid public key'            src/zkpAttestList.ts:117 */
    ZK_E_T_INF = 3,              /* 'T[i] is at infinity'           src/exp/exp.ts:151 */
    ZK_E_T1_INF = 4,             /* 'T1 is at infinity'             src/exp/exp.ts:193 */
    ZK_E_PADD_INF = 5,           /* 'P/Q/R is at infinity'          src/exp/pointAdd.ts:117-125 (no input reaches it) */
    ZK_E_POINTS_DONT_ADD = 6,    /* "Points don't add up!"          src/exp/pointAdd.ts:105 (a signature with r = 0 mod n: invMod(0) = 0) */
    ZK_E_R_INF = 7,              /* 'R is at infinity'              src/zkpAttestList.ts:159 */
    ZK_E_PARAMS_NOT_FOUND = 8,   /* 'params not found'              src/exp/exp.ts:270,302 */
    ZK_E_SECLEVEL = 9,           /* 'security level not achieved'   src/exp/exp.ts:244 */
    ZK_E_BAD_ENCODING = 10,      /* deserialisation failures        src/curves/weier.ts:87,258, edwards.ts:84,207 */
    ZK_E_RNG_EXHAUSTED = 11,     /* stream too short / too many rejected fills */
    ZK_E_BUFFER = 12,            /* output buffer too small, or context not configured */
    ZK_E_INTERPOLATION = 13,     /* 'incorrect interpolation'       src/proofGK/interpolate.ts:65 */
    ZK_E_ARG = 14,               /* invalid argument */
    ZK_E_DEVICE = 15             /* HIP runtime failure (zk_last_error has the text) */
};

enum { ZK_RNG_SEED = 0, ZK_RNG_STREAM = 1 };
typedef struct {
    int32_t mode;           /* ZK_RNG_SEED: data = B x 32-byte seeds; ZK_RNG_STREAM: data = B x stride_blocks x 32 bytes */
    const uint8_t *data;
    uint64_t stride_blocks; /* ZK_RNG_STREAM only */
} zk_rng;

/* Creates an engine bound to one GPU.  Owns the HIP streams, the fixed-base tables and the workspace.  (SURVEY.md section 8(b)
 * sketched zk_ctx_create(device_ids, n_dev); here one zk_ctx is one GPU and zk_pool_create below takes the device list.) */
zk_status zk_ctx_create(int device_id, zk_ctx **out);   /* *out = NULL on failure (never a half-built context) */
void zk_ctx_destroy(zk_ctx *ctx);
const char *zk_strerror(zk_status s);
const char *zk_last_error(const zk_ctx *ctx);           /* ctx = NULL: why this thread's last zk_ctx_create failed */

/* Replaces SystemParametersList as produced by generateParamsList (src/zkpAttestList.ts:88-92,62-78):
 * NistGroup.h, ProofGroup.g, ProofGroup.h and SecLevel.  Builds the fixed-base tables for g, h, G, h_NIST. */
zk_status zk_ctx_set_params(zk_ctx *ctx, const uint8_t nist_h[64], const uint8_t tom_g[72], const uint8_t tom_h[72], uint32_t sec_level);

/* Replaces the `keys: bigint[]` argument (src/zkpAttestList.ts:110,150): n_keys big-endian 32-byte scalars.
 * Pads to the next power of two with copies of keys[0] (src/proofGK/gk.ts:75-86).  The _device variant takes a
 * pointer already resident on this context's GPU (e.g. the target of an RCCL broadcast).
 * n_keys >= 2.  A ONE-key ring is refused with ZK_E_ARG: the reference cannot prove over it either -- the padded length is 1, n = 0, and
 * proveMembership's interpolate([], []) evaluates `-x[0] % m` with x[0] undefined (src/proofGK/interpolate.ts:40), a TypeError
 * ("Cannot mix BigInt and other types") for every `which` -- so no proof over such a ring exists to be verified; the facade throws that
 * TypeError.  `which` (zk_prove_batch) indexes the PADDED ring: [0, n_keys) are the caller's keys, [n_keys, N) the padding copies of keys[0]
 * (the owner of keys[0] may name any of them; the index bits enter the proof), and which >= N is per-proof status ZK_E_ARG, the
 * reference's `values[index].k` on undefined (gk.ts:162; proveMembership runs last, so an invalid key and 'T[i] is at infinity' are
 * reported in its place when they apply too; 'T1 is at infinity' -- a planted nonce and a bad index in one call -- is not). */
zk_status zk_ctx_set_ring(zk_ctx *ctx, const uint8_t *keys_be32, uint64_t n_keys);
zk_status zk_ctx_set_ring_device(zk_ctx *ctx, const void *d_keys_be32, uint64_t n_keys);

/* keyToInt (src/zkpAttestList.ts:94-102) for a whole key set: n_keys public keys as 64-byte affine (x, y) big-endian
 * (the WebCrypto 'raw' export without its 0x04 prefix) -> n_keys 32-byte big-endian ring entries (the x coordinate,
 * reduced mod p like toAffine does).  per_key_status[i] = ZK_E_POINT_NOT_IN_GROUP where deserializePoint would throw
 * (src/curves/weier.ts:74-89: curve equation mod p, coordinates not range-checked); such entries are written as zero.
 * Host pointers; the return value is ZK_OK even if some keys are bad (check the statuses). */
zk_status zk_keys_to_ints(zk_ctx *ctx, uint64_t n_keys, const uint8_t *pk_xy64, uint8_t *keys_be32, int32_t *per_key_status);

/* Proofs processed per pipeline pass (workspace grows linearly with it: about 0.9 MB per proof at secLevel 80).
 * Default 4096, maximum 2^18. */
zk_status zk_ctx_set_chunk(zk_ctx *ctx, uint32_t proofs_per_chunk);

/* Pipeline lanes (1..4, default 2; ZKATTEST_LANES): consecutive chunks rotate over that many HIP streams and workspaces, so the
 * low-occupancy per-proof kernels, the host's waits and (host-pointer calls) the output phases of one chunk overlap the heavy
 * kernels of the others; 1 = strictly serial kernels (what bench.py uses for its per-kernel roofline pass).  Every lane holds
 * a workspace of `chunk` proofs. */
zk_status zk_ctx_set_lanes(zk_ctx *ctx, uint32_t lanes);
/* (The lane-scheduling experiments of round 5 -- a "heavy queue", stream priorities, CU masks, staggered starts: none beat this default -- are recorded in
 * profiles/r05_overlap.txt; their switches are gone from the library.) */
/* Width W (8..26 bits, default 16) of the fixed-base comb tables of the Tom-256 bases g and h: a commitment
 * v*g + r*h (PedersenParams.commit, src/commit/pedersen.ts:53-58) costs 2*ceil(256/W) table additions, the tables
 * take 2 * ceil(256/W) * 2^W * 128 bytes of HBM (0.27 GB at 16, 3.5 GB at 20, 47 GB at 24); 25 and 26 select SIGNED
 * digits (half the entries per window): 25 = the 22 additions of 24 bits in 23.6 GB, 26 = 20 additions in 86 GB.  They are rebuilt by the
 * next zk_ctx_set_params, which must follow.  The proof bytes do not depend on W.  The environment variable
 * ZKATTEST_COMB_BITS sets the default of new contexts. */
zk_status zk_ctx_set_comb_bits(zk_ctx *ctx, uint32_t bits);

/* zk_prove_batch on a page-locked `out`: 1 (default) = the first chunks of the L lanes hold chunk/L, 2*chunk/L, ... proofs, so
 * the lanes run out of phase and their output phases (PCIe transfers) interleave; 0 = uniform chunks; n >= 2 = n rising first
 * chunks (chunk/n, 2*chunk/n, ...) whatever the number of lanes.  The proof bytes do not depend on it. */
zk_status zk_ctx_set_host_taper(zk_ctx *ctx, uint32_t on);
/* The prover runs a chunk's PointAdd phase (src/exp/pointAdd.ts:92-163: 80 % of the proof bytes) in slices of `proofs`
 * consecutive proofs; with a page-locked `out` every slice is followed by the DMA of the proofs it completed.  0 (default) =
 * 4096 for page-locked output (512 / 1024 in the small chunks of short calls -- one chunk of at most 2048 / 8192 proofs, or chunks
 * of at most 2048 / 4096 in a call of at most 32768 -- whose output's transfer is a large part of their time), no slicing otherwise; at least 64.  The proof bytes do not depend on it. */
zk_status zk_ctx_set_slice(zk_ctx *ctx, uint32_t proofs);

/* Verifier strategy for the Tom-256 relations: in a chunk of at least min_chunk proofs (default 256) the relations of ALL
 * proofs are checked with one bucket-method multi-scalar sum (independent 128-bit multipliers per relation and per proof);
 * a chunk is cut into 8 (or 64: zk_ctx_set_verify_groups) contiguous groups of proofs whose sums come out of the same pass, and only
 * the groups whose sum is not the identity -- some proof of theirs is bad -- go through the per-proof sums to tell which.  ok[] and the
 * statuses are the same either way; a forged proof costs the per-proof sums of its group, and the chunk-wide sum has a fixed
 * cost of a few milliseconds, hence the threshold.  0 = never, 1 = always.  (ZKATTEST_VERIFY_BATCH)
 * Chunks of at least 8192 proofs (ZKATTEST_P256_BATCH, read by zk_ctx_create; 0 = never) that take this path sum their P-256 relations the same way, per group,
 * beside the Tom-256 sums; if a group's P-256 total is not the identity the chunk's P-256 relations are checked per proof (zk_test_counter 3 counts the
 * proofs settled by the pass). */
zk_status zk_ctx_set_batch_verify(zk_ctx *ctx, uint32_t min_chunk);
/* Groups per chunk of that check: 8 (default; 16-bit windows) or 64 (13-bit windows: 25 % more bucket additions and four more sorts on
 * every chunk -- about 8 % of the device-resident verification rate at the headline shape, nothing where the PCIe link is the
 * bound -- for a forged proof that costs a 64th of its chunk instead of an eighth).  Verdicts and statuses do not depend on it.
 * (ZKATTEST_VERIFY_GROUPS) */
zk_status zk_ctx_set_verify_groups(zk_ctx *ctx, uint32_t groups);

/* Per-key tables (default on; rings of up to 65 536 keys): zk_ctx_set_ring stores 128 multiples x 33 byte positions of EVERY ring key
 * (264 KB per key, 17.7 GB at 2^16 keys, built in ~0.1 s), so that the prover's u2 * publicKey (src/zkpAttestList.ts:129-131) and the
 * alpha_i * R of proveExp (src/exp/exp.ts:144-149, as (alpha_i u1) * G + (alpha_i u2) * publicKey) are sums of 33 gathered entries instead of
 * a doubling chain and a per-proof table of R.  A proof whose `which` names a ring value that is not its own key's x-coordinate takes
 * the per-proof path; the bytes are the same either way.  0 = per-proof tables for every proof.  Call before zk_ctx_set_ring.
 * (ZKATTEST_KEYTAB) */
zk_status zk_ctx_set_key_tables(zk_ctx *ctx, uint32_t on);

/* The ring fold's 8 low index bits on the matrix cores (v_mfma_i32_16x16x64_i8; rings of at least 4096 keys), 1 (default), or as
 * 64-bit multiply-adds on the vector ALU, 0.  Verifier: verifyMembership's total (src/proofGK/gk.ts:239-250), every block of 256 keys
 * as int8 matrix products (ZKATTEST_GK_MFMA).  Prover: the table path's coefficient classes 2..6 of proveMembership's polynomial
 * (gk.ts:141-171), per group of proofs with equal low index bits; its operand table (0.83 GB at 2^16 keys, 13 GB at 2^20) is built
 * by zk_ctx_set_ring unless ZKATTEST_GK_MFMA_PROVE=0.  Exact integer arithmetic either way: same proofs, totals and verdicts.
 * Takes effect with the next prove / verify call.
 * Every zk_ctx_set_* above returns ZK_E_ARG while streamed jobs are queued on the context (zk_prove_submit below): a queued job keeps
 * the lanes, chunk size, workspaces and mode it was planned with. */
zk_status zk_ctx_set_ring_fold(zk_ctx *ctx, uint32_t matrix_pipe);

/* Wire layout of the proofs (default ZK_WIRE_ZKA1).  ZK_WIRE_ZKA1P is ZKA1 with every Tom-256 coordinate in the reference's own 33 bytes
 * instead of 36 (src/curves/edwards.ts:194-203) and the magic "ZK1P": every run of Tom points in the layout holds an even number of points, so
 * all other fields stay 4-byte aligned; 5.3 % fewer bytes (160.0 instead of 169.0 KB per proof at secLevel 80, n = 16), which is what the
 * host-pointer entry points are bound by (PCIe).  The prover's writers emit the chosen layout directly (every entry point, device pointers
 * included); the verifier expands packed proofs on the device, chunk by chunk, before its unchanged kernels read them (every entry point; a
 * device-pointer call reads one offset per chunk back first).  Verdicts and statuses are those of the ZKA1 form of the same proofs.
 * zk_proof_pack / zk_proof_unpack convert single proofs on the host; the JSON converters accept both layouts and write ZKA1. */
enum { ZK_WIRE_ZKA1 = 0, ZK_WIRE_ZKA1P = 1 };
zk_status zk_ctx_set_wire(zk_ctx *ctx, uint32_t wire);
/* host-side conversions of ONE proof; *out_len is always set to the required size, ZK_E_BUFFER when out_cap is too small, ZK_E_BAD_ENCODING
 * when the input is not a structurally complete proof of its layout (header, length, challenge bits) */
zk_status zk_proof_pack(const uint8_t *zka1, uint64_t len, uint8_t *out, uint64_t out_cap, uint64_t *out_len);
zk_status zk_proof_unpack(const uint8_t *zka1p, uint64_t len, uint8_t *out, uint64_t out_cap, uint64_t *out_len);

/* Upper bound of one proof's size, in the context's wire layout, for the current params/ring. */
uint64_t zk_proof_max_size(const zk_ctx *ctx);

/* Page-locked host memory for the big buffers of the host-pointer entry points (`out` of zk_prove_batch, `proofs` of
 * zk_verify_batch: ~169 KB per proof at secLevel 80).  With such a buffer the proof bytes move by DMA chunk by chunk while
 * the neighbouring chunks are being proved / verified; with ordinary (pageable) memory the runtime stages one blocking copy
 * through its own bounce buffers (measured 8.6 GB/s, several times the proving time).  Memory that the caller page-locked
 * itself (hipHostMalloc, hipHostRegister) is recognised too.  NULL when the allocation fails. */
void *zk_host_alloc(size_t bytes);
void zk_host_free(void *p);

/* Replaces B calls of proveSignatureList(params, msgHash, sigBytes, publicKey, which, keys)
 * (src/zkpAttestList.ts:104-145).  pk_xy is the WebCrypto 'raw' export without its 0x04 prefix.
 * out receives the proofs back to back; out_off[b]..out_off[b+1] delimits proof b (empty when
 * per_proof_status[b] != 0).  Host pointers; `out` from zk_host_alloc is filled by overlapped DMA.  `out` may also be memory of
 * this context's GPU (hipMalloc): the proofs then stay in HBM and only the inputs cross the link. */
zk_status zk_prove_batch(zk_ctx *ctx, uint64_t B, const uint8_t *msg_hash /*Bx32*/, const uint8_t *sig /*Bx64*/,
                         const uint8_t *pk_xy /*Bx64*/, const uint32_t *which /*B*/, const zk_rng *rng,
                         uint8_t *out, uint64_t out_cap, uint64_t *out_off /*B+1*/, int32_t *per_proof_status /*B*/);

/* Same, every pointer (including rng->data) resident in this GPU's HBM; nothing crosses PCIe.  Synchronous. */
zk_status zk_prove_batch_device(zk_ctx *ctx, uint64_t B, const void *d_msg_hash, const void *d_sig, const void *d_pk_xy,
                                const void *d_which, const zk_rng *rng_device, void *d_out, uint64_t out_cap,
                                void *d_out_off /*u64[B+1]*/, void *d_per_proof_status /*i32[B]*/);

/* Replaces B calls of verifySignatureList(params, msgHash, keys, proof) (src/zkpAttestList.ts:147-184):
 * ok[b] = 1 iff the reference verifier returns true; per_proof_status[b] != 0 mirrors a thrown error
 * ('params not found', 'T is at infinity', deserialisation failures ...), in which case ok[b] = 0.
 * The reference verifier is randomised: it checks a random 20-subset of the secLevel reps (src/exp/exp.ts:95-109,
 * 261-264).  verifier_seeds (B x 32 bytes, may be NULL) fixes that choice under the same contract as the prover's
 * RNG: verifier fill k = SHA-256(seed_b || be64(k)), randomScalar() consumes a 32-byte fill, rnd(small) the first
 * byte of a fill; fills are consumed in the reference's order (2n+1 randomScalar draws of verifyMembership, then
 * generateIndices).  The random multipliers of Relation.drain (src/curves/multimult.ts:168-173) do not influence the
 * boolean; the engine draws its own 128-bit ones (from the same seed).  The seeds must be unpredictable to whoever made
 * the proofs and independent across proofs.  NULL: the engine draws fresh OS randomness for the call (what the
 * reference does with crypto.getRandomValues); pass seeds only to reproduce a run.
 * proofs must be packed back to back (proof_off[0] = 0, 4-byte aligned).  Host pointers; `proofs` from zk_host_alloc is
 * read by overlapped DMA.
 * Non-canonical encodings: a P-256 coordinate in [p, 2^256) is accepted where deserializePoint accepts it (the curve equation is
 * checked mod p, src/curves/weier.ts:74-89) and enters the Fiat-Shamir hashes REDUCED, as toBytes -> toAffine does
 * (src/curves/weier.ts:231-255).  Tom-256 coordinates must be canonical (< the field prime, zero padding bytes): ZKA1 is this
 * engine's format and is stricter here than the reference's JSON reader, which reduces out-of-range values silently
 * (src/curves/edwards.ts:204-209); such a proof gets ZK_E_BAD_ENCODING.
 * Status codes are exact (tests/test_gpu_mutants.py: seeded mutants, engine == oracle): ZK_E_BAD_ENCODING for anything that does not
 * deserialise -- magic, lengths, a header n above 63, challenge bits set above secLevel, any point of the announced structure off its
 * curve, also inside a GKProof of the wrong length --; ok = 0 with status 0 where the reference returns false (membership, a GKProof
 * whose length is not the ring's, gk.ts:208-218, a failed relation); otherwise the exception verifyExp throws FIRST when it walks
 * the 20 sampled repetitions in order (exp.ts:265-346): ZK_E_PARAMS_NOT_FOUND, ZK_E_T_INF or ZK_E_T1_INF.  ZK_E_R_INF cannot occur (R
 * travels in affine form).  One restriction: a batch is homogeneous in secLevel -- a proof whose header announces another repetition
 * count than the context's gets ZK_E_BAD_ENCODING, where the reference would verify it with its own count (exp.ts:243-260); a context
 * with secLevel < 20 refuses the call with ZK_E_SECLEVEL. */
zk_status zk_verify_batch(zk_ctx *ctx, uint64_t B, const uint8_t *msg_hash /*Bx32*/, const uint8_t *proofs,
                          const uint64_t *proof_off /*B+1*/, const uint8_t *verifier_seeds /*Bx32 or NULL*/,
                          uint8_t *ok /*B*/, int32_t *per_proof_status /*B*/);
zk_status zk_verify_batch_device(zk_ctx *ctx, uint64_t B, const void *d_msg_hash, const void *d_proofs,
                                 const void *d_proof_off, const void *d_verifier_seeds, void *d_ok, void *d_per_proof_status);

/* ---- two (or more) batches in flight on one context.  zk_prove_batch / zk_verify_batch are synchronous: each call pays its own head
 * (no byte of a chunk exists before its stage 1 is over) and its own tail (the copies of the last slices, with nothing left to
 * hide them).  The submit / wait pair splits a call so that the pipeline keeps running ACROSS calls: submit stages the inputs and
 * queues the job (the verifier's proof bytes start crossing PCIe at once, right behind the bytes of the job before), wait drives the
 * job's chunks -- enqueueing stage 1 of the NEXT job's first chunks while this job's last chunks are in their output phase -- and
 * hands the results back.  Steady state:  submit(0); submit(1); wait(0); submit(2); wait(1); submit(3); wait(2); ...
 * Rules: one thread per context; waits in submission order; at most 4 jobs queued; jobs of one kind (prove or verify) at a time;
 * `out` / `proofs` page-locked (zk_host_alloc; ZK_E_ARG otherwise); every pointer of a job -- inputs included -- stays valid and
 * untouched until its wait returns; chunk, lanes, parameters and ring do not change while jobs are queued; the synchronous calls
 * return ZK_E_ARG while jobs are queued.  Bytes, statuses and verdicts are those of the synchronous calls.  zk_ctx_destroy abandons
 * queued jobs.  (The reference proves one signature per call on one thread, src/zkpAttestList.ts:104-145: no counterpart.) */
typedef struct zk_job zk_job;
zk_status zk_prove_submit(zk_ctx *ctx, uint64_t B, const uint8_t *msg_hash, const uint8_t *sig, const uint8_t *pk_xy, const uint32_t *which,
                          const zk_rng *rng, uint8_t *out, uint64_t out_cap, uint64_t *out_off /*B+1*/, int32_t *per_proof_status /*B*/, zk_job **job);
/* the same job with every buffer already in HBM (zk_prove_batch_device's arguments; rng->data is a device pointer): nothing crosses
 * the link, consecutive batches keep the lanes' pipelines full.  The buffers must be complete when the call is made. */
zk_status zk_prove_submit_device(zk_ctx *ctx, uint64_t B, const uint8_t *d_msg_hash, const uint8_t *d_sig, const uint8_t *d_pk_xy, const uint32_t *d_which,
                                 const zk_rng *rng, uint8_t *d_out, uint64_t out_cap, uint64_t *d_out_off /*B+1*/, int32_t *d_per_proof_status /*B*/,
                                 zk_job **job);
zk_status zk_prove_wait(zk_ctx *ctx, zk_job *job);   /* the job is released whatever the result */
zk_status zk_verify_submit(zk_ctx *ctx, uint64_t B, const uint8_t *msg_hash, const uint8_t *proofs, const uint64_t *proof_off /*B+1*/,
                           const uint8_t *verifier_seeds /*Bx32 or NULL*/, uint8_t *ok /*B*/, int32_t *per_proof_status /*B*/, zk_job **job);
zk_status zk_verify_wait(zk_ctx *ctx, zk_job *job);

/* ---- several GPUs of one node behind one handle (the reference is single-threaded,
 * src/zkpAttestList.ts:104-145 proves one signature per call).  Proofs are independent given (params, ring): a batch is split
 * into contiguous shards, shard i = proofs [i*B/G, (i+1)*B/G) on device_ids[i], one host thread per device, no exchange while
 * proving or verifying.  zk_pool_set_ring uploads the ring once and broadcasts it device-to-device (RCCL ncclBroadcast over
 * xGMI; hipMemcpyPeer when RCCL is not usable -- zk_pool_ring_transport tells which: "rccl", "peer-copy", "single");
 * fixed-base tables and the per-ring table are rebuilt locally on every device, concurrently.  zk_pool_ctx(i) gives the
 * per-device context for the settings above (chunk, lanes, comb width: before zk_pool_set_params) and for zk_last_error.
 * A pool call is not re-entrant; the listed devices may repeat (several contexts on one GPU). */
typedef struct zk_pool zk_pool;
zk_status zk_pool_create(const int *device_ids, int n_dev, zk_pool **out);   /* *out = NULL on failure (never a half-built pool) */
void zk_pool_destroy(zk_pool *pool);
int zk_pool_size(const zk_pool *pool);
zk_ctx *zk_pool_ctx(zk_pool *pool, int i);
const char *zk_pool_last_error(const zk_pool *pool);                        /* pool = NULL: why this thread's last zk_pool_create failed;
                                                                             * after a zk_pool_set_ring whose transport is not "rccl": why RCCL was not used */
/* Host side of a shard.  Every shard of a pool call runs on its own host thread, bound to the CPUs next to its device (sysfs
 * local_cpulist of the device's PCI address; ZKATTEST_POOL_AFFINITY=0 switches the binding off).  zk_pool_host_alloc returns a
 * page-locked buffer for `out` of zk_pool_prove_batch / `proofs` of zk_pool_verify_batch whose per-shard regions
 * [i * R, (i+1) * R), R = (bytes / G) & ~255, were first touched on the NUMA node of device i, so that the ~40 GB/s of proof
 * bytes per device stay on that device's socket.  Free with zk_pool_host_free.  zk_pool_numa_node: -1 = unknown. */
void *zk_pool_host_alloc(zk_pool *pool, size_t bytes);
void zk_pool_host_free(void *p);
int zk_pool_numa_node(const zk_pool *pool, int i);
/* wall milliseconds every shard's host thread spent in its part of the last pool call (set_params, set_ring, prove, verify);
 * returns the number of shards */
int zk_pool_shard_ms(const zk_pool *pool, float *ms, int cap);
const char *zk_pool_ring_transport(const zk_pool *pool);
/* the file the nccl* entry points were taken from ("" while RCCL was never loaded).  zk_pool_set_ring looks for librccl in this order: the path in
 * ZKATTEST_RCCL_LIB (if it cannot be loaded RCCL is not used at all); a librccl the process has already mapped (under PyTorch: the wheel's own, bound to
 * the HIP runtime this library is then running on as well); the system's librccl.so.1.  ZKATTEST_NO_RCCL=1 goes straight to peer copies. */
const char *zk_pool_rccl_library(const zk_pool *pool);
void zk_pool_shard(const zk_pool *pool, uint64_t B, int i, uint64_t *first, uint64_t *count);
zk_status zk_pool_set_params(zk_pool *pool, const uint8_t nist_h[64], const uint8_t tom_g[72], const uint8_t tom_h[72], uint32_t sec_level);
zk_status zk_pool_set_ring(zk_pool *pool, const uint8_t *keys_be32, uint64_t n_keys);
/* zk_prove_batch over all devices.  Shard i writes its proofs back to back from out + i * ((out_cap / G) & ~255): proof b lies
 * at out_off[b] .. out_off[b] + out_len[b] (its ZKA1 header carries the same length); there are gaps between shards, none
 * inside one.  ZK_E_BUFFER when a shard does not fit its region.  `out` from zk_host_alloc is filled by overlapped DMA. */
zk_status zk_pool_prove_batch(zk_pool *pool, uint64_t B, const uint8_t *msg_hash, const uint8_t *sig, const uint8_t *pk_xy, const uint32_t *which,
                              const zk_rng *rng, uint8_t *out, uint64_t out_cap, uint64_t *out_off /*B*/, uint64_t *out_len /*B*/,
                              int32_t *per_proof_status /*B*/);
/* The same with the proofs left in HBM: shard i writes its proofs back to back from d_out[i], a buffer of out_cap[i] bytes on
 * device_ids[i] (zk_pool_device_alloc / zk_pool_device_free); out_off[b] is relative to the proof's own shard buffer.  Only the 160 bytes of
 * inputs per proof cross PCIe: through this entry point a multi-GPU run measures the GPUs and their host threads without the node's host
 * memory (eight shards of zk_pool_prove_batch emit ~55 GB/s of page-locked writes each).  zk_prove_batch accepts a
 * device-resident `out` the same way for one context. */
zk_status zk_pool_prove_batch_device(zk_pool *pool, uint64_t B, const uint8_t *msg_hash, const uint8_t *sig, const uint8_t *pk_xy, const uint32_t *which,
                                     const zk_rng *rng, void *const *d_out /*G*/, const uint64_t *out_cap /*G*/, uint64_t *out_off /*B*/, uint64_t *out_len /*B*/,
                                     int32_t *per_proof_status /*B*/);
void *zk_pool_device_alloc(zk_pool *pool, int i, size_t bytes);   /* hipMalloc on device_ids[i]; the calling thread's current device is kept */
void zk_pool_device_free(zk_pool *pool, int i, void *p);
/* zk_verify_batch over all devices.  Inside every shard the proofs must lie back to back in index order (true for the output
 * of zk_pool_prove_batch and for any fully packed buffer), every shard starting 4-byte aligned; ZK_E_ARG otherwise. */
zk_status zk_pool_verify_batch(zk_pool *pool, uint64_t B, const uint8_t *msg_hash, const uint8_t *proofs, const uint64_t *proof_off /*B*/,
                               const uint64_t *proof_len /*B*/, const uint8_t *verifier_seeds /*Bx32 or NULL*/, uint8_t *ok /*B*/,
                               int32_t *per_proof_status /*B*/);

/* The streamed form of the two pool calls (see "two batches in flight"): every device's shard goes through zk_prove_submit / zk_prove_wait on
 * its own context, so a node keeps two or three batches in flight per GPU.  Same rules per device (waits in submission order, `out` /
 * `proofs` page-locked -- zk_pool_host_alloc --, every pointer valid until the wait returns); (out_off, out_len) are filled by the wait. */
typedef struct zk_pool_job zk_pool_job;
zk_status zk_pool_prove_submit(zk_pool *pool, uint64_t B, const uint8_t *msg_hash, const uint8_t *sig, const uint8_t *pk_xy, const uint32_t *which, const zk_rng *rng,
                               uint8_t *out, uint64_t out_cap, uint64_t *out_off /*B*/, uint64_t *out_len /*B*/, int32_t *per_proof_status /*B*/, zk_pool_job **job);
zk_status zk_pool_prove_wait(zk_pool *pool, zk_pool_job *job);
zk_status zk_pool_verify_submit(zk_pool *pool, uint64_t B, const uint8_t *msg_hash, const uint8_t *proofs, const uint64_t *proof_off /*B*/, const uint64_t *proof_len /*B*/,
                                const uint8_t *verifier_seeds /*Bx32 or NULL*/, uint8_t *ok /*B*/, int32_t *per_proof_status /*B*/, zk_pool_job **job);
zk_status zk_pool_verify_wait(zk_pool *pool, zk_pool_job *job);

/* ---- hardened mode: the two protocol hardenings the reference leaves as TODOs, as an explicit opt-in.  NOT byte-compatible
 * with the reference: a proof made in one mode only verifies in that mode.  Default: ZK_MODE_REFERENCE (byte parity).
 *   (1) src/commit/pedersen.ts:62 "we must generate h without using scalar mult": zk_hardened_h derives NistGroup.h and
 *       ProofGroup.h from SHA-256 by try-and-increment (specification in csrc/h2c_host.cpp), so that nobody knows their
 *       discrete logarithms; pass them to zk_ctx_set_params like any other parameters.  Host-only, no context.
 *   (2) src/proofGK/gk.ts:178 "we should hash in the statement": with zk_ctx_set_mode(ctx, ZK_MODE_HARDENED) the challenge of
 *       the membership proof is SHA-256(cl || ca || cb || cd || "ZKAttest-GK-statement-v1" || ring digest || msgHash || R ||
 *       keyXcom)[0..10) instead of the hash of the commitments alone; ring digest = zk_ring_digest: SHA-256("ZKAttest-ring-v1" ||
 *       be64(N) || SHA-256 of every 256 consecutive entries of the padded ring).  Prover and verifier must use the same mode. */
enum { ZK_MODE_REFERENCE = 0, ZK_MODE_HARDENED = 1 };
zk_status zk_ctx_set_mode(zk_ctx *ctx, uint32_t mode);

/* Zeroes the witness-derived device memory of the context: the prover lanes' workspaces (per proof: the RNG stream -- 116 KB --, the nonces, s1 = s / r, the
 * blinders of every commitment) and the staging copy of the inputs of the host-pointer calls (signatures, seeds).  The library does this by itself in
 * zk_ctx_destroy and when a prove call fails; a successful call leaves the memory to be overwritten by the next one, and a host that wants it gone earlier
 * calls this (cost: one memset of the workspaces, ~4 ms per lane at 22 016 proofs per chunk).  No streamed job may be in flight.  The reference has no
 * counterpart (src/zkpAttestList.ts:104-145 leaves its BigInts to the garbage collector). */
zk_status zk_ctx_wipe(zk_ctx *ctx);
zk_status zk_ring_digest(zk_ctx *ctx, uint8_t digest[32]);
zk_status zk_hardened_h(const uint8_t *tag, uint64_t tag_len, uint8_t nist_h[64], uint8_t tom_h[72]);

/* Seeded synthetic workload generator: fills device or host buffers with a ring of
 * n_keys uniform scalars, and B valid ECDSA P-256 signatures whose public keys' x-coordinates are planted at
 * ring[which_b], which_b = b mod n_keys.  Mirrors oracle/zkattest_ref.py synth_* byte for byte.  Host pointers. */
zk_status zk_synth_workload(zk_ctx *ctx, uint64_t seed, uint64_t n_keys, uint64_t B, uint8_t *ring_be32 /*n_keys x 32*/,
                            uint8_t *msg_hash, uint8_t *sig, uint8_t *pk_xy, uint32_t *which, uint8_t *rng_seeds /*Bx32*/);
/* Derives the synthetic SystemParametersList of seed S (h = k*g with k = SHA-256 tags). */
zk_status zk_synth_params(zk_ctx *ctx, uint64_t seed, uint8_t nist_h[64], uint8_t tom_g[72], uint8_t tom_h[72]);

/* JSON wire format of SignatureProofList (writeJson/readJson, src/serde.ts:21-36, over the typedjson decorators of
 * src/zkpAttestList.ts:27-35, exp/exp.ts:26-40, exp/pointAdd.ts:28-38, commit/mult.ts:26-40, commit/equality.ts:27-33,
 * proofGK/gk.ts:31-40, curves/{weier.ts:92-101, edwards.ts:89-98, group.ts:155-161}, bignum/big.ts:230-248).
 * Host-only conversions between one ZKA1 proof and its JSON text; no context and no GPU needed.  *out_len is always
 * set to the required size; ZK_E_BUFFER if out_cap is too small (call once with out = NULL to size).  from_json is
 * order-tolerant and ignores "__type"/unknown members; like JSON.parse it keeps the LAST of duplicated members, decodes the string
 * escapes (\uXXXX included) and rejects text that is not JSON (stray tokens, raw control characters, unknown escapes); it checks
 * structure, group names and hex syntax/width --
 * curve membership is checked where the reference checks it semantically, in zk_verify_batch's validation. */
zk_status zk_proof_to_json(const uint8_t *proof, uint64_t proof_len, char *out, uint64_t out_cap, uint64_t *out_len);
zk_status zk_proof_from_json(const char *json, uint64_t json_len, uint8_t *out, uint64_t out_cap, uint64_t *out_len);
/* The same for whole batches on `threads` host threads (0 = one per hardware thread; ZKATTEST_JSON_THREADS): the reference's bench
 * times toJson / fromJson per proof (bench/zkpAttestList.bench.ts:63-68); at ~600 KB of text per proof a batch of the GPU's size
 * needs every core.  Item i is proofs[proof_off[i] .. proof_off[i+1]) (texts[text_off[i] .. text_off[i+1])); the results lie back
 * to back in `out`, delimited by the n + 1 offsets written; an item that does not convert gets its status and an empty result.
 * ZK_E_BUFFER when `out` is too small: the offsets are complete anyway (the last one is the size to come back with). */
zk_status zk_proofs_to_json_batch(uint64_t n, const uint8_t *proofs, const uint64_t *proof_off /*n+1*/, char *out, uint64_t out_cap,
                                  uint64_t *text_off /*n+1*/, int32_t *per_proof_status /*n*/, uint32_t threads);
zk_status zk_proofs_from_json_batch(uint64_t n, const char *texts, const uint64_t *text_off /*n+1*/, uint8_t *out, uint64_t out_cap,
                                    uint64_t *proof_off /*n+1*/, int32_t *per_proof_status /*n*/, uint32_t threads);

/* Timing of the last prove/verify call from HIP events around every kernel family: per family the accumulated GPU milliseconds (names[i] are static
 * strings; a name that starts with '+' is a part of another family), and *total_ms = their SUM ('+' parts excluded).  With one lane and a large chunk the
 * families run one after the other and the sum is the GPU time of the call; with several lanes, and in small calls that fork work onto side streams, the
 * families overlap and the sum exceeds the time that passed -- zk_last_wall_ms gives that: earliest start to latest end over the same events.
 * The events sit between the kernels of a stream and cost a call of a few proofs 0.15-0.45 ms (two per family and chunk): by default (ZK_TIMING_AUTO) only
 * blocking calls of more than 8 192 proofs record them, and after a smaller call zk_last_timing reports no families (returns 0, *total_ms = 0);
 * zk_ctx_set_timing(ctx, ZK_TIMING_ON) records them in every blocking call, ZK_TIMING_OFF in none.  Streamed jobs (zk_prove_submit ...) never record them. */
#define ZK_TIMING_OFF 0
#define ZK_TIMING_ON 1
#define ZK_TIMING_AUTO 2
zk_status zk_ctx_set_timing(zk_ctx *ctx, int mode);
uint32_t zk_last_timing(const zk_ctx *ctx, float *total_ms, const char **names, float *ms, uint32_t cap);
float zk_last_wall_ms(const zk_ctx *ctx);

/* Diagnostic: the rate (GB/s) a page-locked copy of `bytes` (at least 1 MiB) reaches on the copy stream of pipeline lane `lane` (0..3), device
 * to host and host to device, measured with HIP events.  ~57 GB/s is what the link carries on MI355X boxes; ~27 GB/s says that the runtime of this
 * PROCESS serves copies with shader blits instead of an SDMA engine, and every host-pointer call will run at about 0.6 of
 * its usual rate whatever the buffers are. */
zk_status zk_ctx_copy_probe(zk_ctx *ctx, uint32_t lane, size_t bytes, int numa_node /* >= 0: the host buffer is bound to that node; -1: wherever
                            the runtime puts it */, float *d2h_gbps, float *h2d_gbps);

/* Unit-test hooks (tests/ call these through the C ABI to compare single primitives with the oracle).
 * zk_pool_test_locality: NUMA node and local CPUs of a PCI address as the pool reads them from sysfs (ZKATTEST_SYSFS_ROOT). */
int zk_pool_test_locality(const char *pci_bus_id, int *numa_node, int *cpus, int cap);
/* work counters: 0 = proofs that went through the verifier's per-proof sums since the context was created (fallback of the batched check);
 * 1 = proofs of the last chunk on lane 0 whose scalar multiplications by the signer's key went through the per-key tables;
 * 2 = live terms that went through the verifier's batched Tom-256 check (bucket pass) since the context was created;
 * 3 = proofs whose P-256 relation was accepted by the cross-proof P-256 pass (chunks of at least ZKATTEST_P256_BATCH proofs, default 8192): every GROUP
 *     without a failing proof counts (a failing group, not its chunk, goes through the per-proof sums);
 * 4 = dependent chains of small calls handed to cooperating waves so far (k_coop.hip: Straus sums, the table of R), process-wide */
uint64_t zk_test_counter(const zk_ctx *ctx, int which);
/*
 * which_field: 0 = F_q (p256.p), 1 = Z_n, 2 = F_t;  op: 0 mul, 1 add, 2 sub, 3 inverse, 4 a*b - a - b (fused double subtraction), 5 (a + b)^2 (dedicated squaring).  count x 40-byte BE operands. */
zk_status zk_test_field_op(zk_ctx *ctx, int which_field, int op, uint64_t count, const uint8_t *a_be40, const uint8_t *b_be40, uint8_t *out_be40);
/* out[i] = v[i]*g + r[i]*h on Tom-256 (72-byte affine), through the fixed-base comb kernel */
zk_status zk_test_tom_commit(zk_ctx *ctx, uint64_t count, const uint8_t *v_be32, const uint8_t *r_be32, uint8_t *out_xy72);
/* out[i] = k[i]*G (base_sel 0) or k[i]*h_NIST (base_sel 1) on P-256 (64-byte affine; all-zero for the identity) */
zk_status zk_test_p256_fixed_mul(zk_ctx *ctx, int base_sel, uint64_t count, const uint8_t *k_be32, uint8_t *out_xy64);
/* digest[i] = SHA-256(msg[i]) for count messages of identical length len */
zk_status zk_test_sha256(zk_ctx *ctx, uint64_t count, uint64_t len, const uint8_t *msgs, uint8_t *digests32);
/* logical RNG draw k of each proof (after rejection mapping), 32 bytes BE */
zk_status zk_test_rng_draws(zk_ctx *ctx, uint64_t B, const zk_rng *rng, uint32_t first_k, uint32_t n_k, uint8_t *out /*B x n_k x 32*/);

#ifdef __cplusplus
}
#endif
#endif
