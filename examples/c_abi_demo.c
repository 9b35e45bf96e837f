/* Minimal C program over the C ABI (include/zkattest.h): synthetic workload -> proveSignatureList for a small batch ->
 * JSON round trip of the first proof -> verifySignatureList.  No Python, no torch: the boundary a cgo / N-API / JNI
 * binding would sit on.
 *
 *   gcc -O2 -Iinclude examples/c_abi_demo.c -o /tmp/zk_demo -Lzkp-ecdsa_amd/lib -lzkattest_hip -Wl,-rpath,$PWD/zkp-ecdsa_amd/lib
 *   /tmp/zk_demo [n_proofs] [n_keys]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "zkattest.h"

#define CHECK(call)                                                                         \
    do {                                                                                    \
        zk_status st_ = (call);                                                             \
        if (st_ != ZK_OK) {                                                                 \
            fprintf(stderr, "%s -> %s (%s)\n", #call, zk_strerror(st_), zk_last_error(ctx)); \
            return 1;                                                                       \
        }                                                                                   \
    } while (0)

int main(int argc, char **argv) {
    uint64_t B = argc > 1 ? strtoull(argv[1], 0, 10) : 4, nkeys = argc > 2 ? strtoull(argv[2], 0, 10) : 16;
    zk_ctx *ctx = NULL;
    CHECK(zk_ctx_create(0, &ctx));

    /* SystemParametersList (src/zkpAttestList.ts:63-92): deterministic synthetic parameters for the demo */
    uint8_t nist_h[64], tom_g[72], tom_h[72];
    CHECK(zk_synth_params(ctx, 1, nist_h, tom_g, tom_h));
    CHECK(zk_ctx_set_params(ctx, nist_h, tom_g, tom_h, 80));

    /* ring of keys + B signed messages whose keys are in the ring */
    uint8_t *ring = malloc(32 * nkeys), *msg = malloc(32 * B), *sig = malloc(64 * B), *pk = malloc(64 * B), *seeds = malloc(32 * B);
    uint32_t *which = malloc(4 * B);
    CHECK(zk_synth_workload(ctx, 1, nkeys, B, ring, msg, sig, pk, which, seeds));
    CHECK(zk_ctx_set_ring(ctx, ring, nkeys));

    /* proveSignatureList x B (seeds: in production 32 fresh random bytes per proof) */
    uint64_t cap = zk_proof_max_size(ctx) * B;
    uint8_t *proofs = malloc(cap);
    uint64_t *off = malloc(8 * (B + 1));
    int32_t *status = malloc(4 * B);
    zk_rng rng = {ZK_RNG_SEED, seeds, 0};
    CHECK(zk_prove_batch(ctx, B, msg, sig, pk, which, &rng, proofs, cap, off, status));
    for (uint64_t b = 0; b < B; b++)
        if (status[b] != ZK_OK) fprintf(stderr, "proof %llu: %s\n", (unsigned long long)b, zk_strerror((zk_status)status[b]));
    printf("%llu proofs, %llu bytes, first proof %llu bytes\n", (unsigned long long)B, (unsigned long long)off[B], (unsigned long long)(off[1] - off[0]));

    /* writeJson -> readJson (src/serde.ts) */
    uint64_t jlen = 0, blen = 0;
    zk_proof_to_json(proofs, off[1] - off[0], NULL, 0, &jlen);
    char *json = malloc(jlen);
    CHECK(zk_proof_to_json(proofs, off[1] - off[0], json, jlen, &jlen));
    uint8_t *back = malloc(off[1] - off[0]);
    CHECK(zk_proof_from_json(json, jlen, back, off[1] - off[0], &blen));
    printf("JSON %llu bytes, round trip %s\n", (unsigned long long)jlen, blen == off[1] - off[0] && !memcmp(back, proofs, blen) ? "identical" : "DIFFERENT");

    /* verifySignatureList x B (NULL seeds: the engine draws OS randomness) */
    uint8_t *ok = malloc(B);
    CHECK(zk_verify_batch(ctx, B, msg, proofs, off, NULL, ok, status));
    uint64_t good = 0;
    for (uint64_t b = 0; b < B; b++) good += ok[b];
    printf("verified %llu of %llu\n", (unsigned long long)good, (unsigned long long)B);
    proofs[off[1] - 1] ^= 1; /* forge the last byte (zd) of proof 0 */
    CHECK(zk_verify_batch(ctx, B, msg, proofs, off, NULL, ok, status));
    printf("after tampering proof 0: ok[0] = %d\n", ok[0]);
    zk_ctx_destroy(ctx);
    return good == B && ok[0] == 0 ? 0 : 2;
}
