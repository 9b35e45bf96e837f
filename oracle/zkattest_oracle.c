/*
 * ORACLE (test infrastructure, NOT product code) -- C restatement of cloudflare/zkp-ecdsa's
 * proveSignatureList / verifySignatureList path.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product (libzkattest_hip.so) never does.
 *
 * It follows the reference's algorithms (window-4 mul/dblmul, RCB complete formulas on P-256,
 * Hisil et al. unified formulas on Tom-256, one inversion per serialised point, the 2*N*n ring
 * loop, Lagrange interpolation, Bos-Coster) so that it can serve as the timed CPU baseline
 * ("port").  File:line citations are into /root/reference/src.
 *
 * PARITY STATUS: "parity unpinned" at proof level -- the TypeScript reference cannot be run in
 * the build container and holds no golden proofs.  This restatement is pinned to the Python
 * restatement (oracle/zkattest_ref.py), byte for byte, through tests/golden/ and to the
 * reference's KATs / public vectors (see that file's header).
 *
 * RNG contract: the k-th 32-byte fill of a proof is SHA-256(seed || be64(k)), or block k of an
 * explicit stream (rng_mode 1).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

typedef uint64_t u64;
typedef unsigned __int128 u128;

/* ------------------------------------------------------------------ status codes (mirror include/zkattest.h) */
enum {
    ZK_OK = 0,
    ZK_E_POINT_NOT_IN_GROUP = 1, /* weier.ts:83 'point not in group' */
    ZK_E_INVALID_KEY = 2,        /* zkpAttestList.ts:117 */
    ZK_E_T_INF = 3,              /* exp.ts:151 'T[i] is at infinity' */
    ZK_E_T1_INF = 4,             /* exp.ts:193 */
    ZK_E_PADD_INF = 5,           /* pointAdd.ts:117-125 */
    ZK_E_POINTS_DONT_ADD = 6,    /* pointAdd.ts:105 */
    ZK_E_R_INF = 7,              /* zkpAttestList.ts:159 */
    ZK_E_PARAMS_NOT_FOUND = 8,   /* exp.ts:270,302 */
    ZK_E_SECLEVEL = 9,           /* exp.ts:244 */
    ZK_E_BAD_ENCODING = 10,      /* deserialisation failures */
    ZK_E_RNG_EXHAUSTED = 11,
    ZK_E_BUFFER = 12,
    ZK_E_INTERPOLATION = 13,     /* interpolate.ts:65 */
    ZK_E_ARG = 14,               /* a TypeError of the JavaScript runtime (`which` past the padded ring, gk.ts:162) */
};

/* ------------------------------------------------------------------ SHA-256 (FIPS 180-4) */
static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
typedef struct {
    uint32_t h[8];
    uint8_t buf[64];
    u64 len;
} sha256_t;
#define ROR(x, n) (((x) >> (n)) | ((x) << (32 - (n))))
static void sha256_block(uint32_t *h, const uint8_t *p) {
    uint32_t w[64], a, b, c, d, e, f, g, hh;
    for (int i = 0; i < 16; i++) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = ROR(w[i - 15], 7) ^ ROR(w[i - 15], 18) ^ (w[i - 15] >> 3);
        uint32_t s1 = ROR(w[i - 2], 17) ^ ROR(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; i++) {
        uint32_t S1 = ROR(e, 6) ^ ROR(e, 11) ^ ROR(e, 25), ch = (e & f) ^ (~e & g);
        uint32_t t1 = hh + S1 + ch + K256[i] + w[i];
        uint32_t S0 = ROR(a, 2) ^ ROR(a, 13) ^ ROR(a, 22), mj = (a & b) ^ (a & c) ^ (b & c);
        uint32_t t2 = S0 + mj;
        hh = g, g = f, f = e, e = d + t1, d = c, c = b, b = a, a = t1 + t2;
    }
    h[0] += a, h[1] += b, h[2] += c, h[3] += d, h[4] += e, h[5] += f, h[6] += g, h[7] += hh;
}
static void sha256_init(sha256_t *s) {
    static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    memcpy(s->h, iv, sizeof iv);
    s->len = 0;
}
static void sha256_update(sha256_t *s, const uint8_t *p, size_t n) {
    size_t fill = s->len & 63;
    s->len += n;
    if (fill) {
        size_t take = 64 - fill < n ? 64 - fill : n;
        memcpy(s->buf + fill, p, take);
        p += take, n -= take, fill += take;
        if (fill < 64) return;
        sha256_block(s->h, s->buf);
    }
    while (n >= 64) sha256_block(s->h, p), p += 64, n -= 64;
    if (n) memcpy(s->buf, p, n);
}
static void sha256_final(sha256_t *s, uint8_t out[32]) {
    u64 bits = s->len * 8;
    uint8_t pad[72] = {0x80};
    size_t fill = s->len & 63, padlen = (fill < 56 ? 56 : 120) - fill;
    for (int i = 0; i < 8; i++) pad[padlen + i] = (uint8_t)(bits >> (56 - 8 * i));
    sha256_update(s, pad, padlen + 8);
    for (int i = 0; i < 8; i++) out[4 * i] = s->h[i] >> 24, out[4 * i + 1] = s->h[i] >> 16, out[4 * i + 2] = s->h[i] >> 8, out[4 * i + 3] = s->h[i];
}
void zko_sha256(const uint8_t *p, u64 n, uint8_t out[32]) {
    sha256_t s;
    sha256_init(&s);
    sha256_update(&s, p, n);
    sha256_final(&s, out);
}

/* ------------------------------------------------------------------ big numbers: 5 x 64-bit limbs, little-endian */
#define NL 5
typedef struct {
    u64 v[NL];
} fe;
typedef struct {
    fe m;      /* modulus */
    int nl;    /* limbs used: 4 (256-bit moduli) or 5 (Tom field) */
    u64 n0;    /* -m^-1 mod 2^64 */
    fe r2;     /* R^2 mod m, R = 2^(64*nl) */
    fe one;    /* R mod m */
    fe m2;     /* m - 2 (Fermat exponent) */
} mctx;

static int fe_is_zero(const fe *a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3] | a->v[4]) == 0; }
static int fe_cmp(const fe *a, const fe *b) {
    for (int i = NL - 1; i >= 0; i--) {
        if (a->v[i] < b->v[i]) return -1;
        if (a->v[i] > b->v[i]) return 1;
    }
    return 0;
}
static int fe_eq(const fe *a, const fe *b) { return fe_cmp(a, b) == 0; }
static u64 fe_add_raw(fe *r, const fe *a, const fe *b) {
    u128 c = 0;
    for (int i = 0; i < NL; i++) c += (u128)a->v[i] + b->v[i], r->v[i] = (u64)c, c >>= 64;
    return (u64)c;
}
static u64 fe_sub_raw(fe *r, const fe *a, const fe *b) {
    u64 br = 0;
    for (int i = 0; i < NL; i++) {
        u128 d = (u128)a->v[i] - b->v[i] - br;
        r->v[i] = (u64)d, br = (u64)(d >> 64) & 1;
    }
    return br;
}
static void fe_from_be(fe *r, const uint8_t *b, int len) { /* big.ts:161-168 */
    memset(r, 0, sizeof *r);
    for (int i = 0; i < len; i++) {
        int bi = len - 1 - i;
        r->v[bi / 8] |= (u64)b[i] << (8 * (bi % 8));
    }
}
static void fe_to_be(const fe *a, uint8_t *b, int len) { /* big.ts:121-134 */
    for (int i = 0; i < len; i++) {
        int bi = len - 1 - i;
        b[i] = bi / 8 < NL ? (uint8_t)(a->v[bi / 8] >> (8 * (bi % 8))) : 0;
    }
}
static void fe_set_u64(fe *r, u64 x) {
    memset(r, 0, sizeof *r);
    r->v[0] = x;
}
/* r = (a + b) mod m, inputs < m */
static void mod_add(const mctx *c, fe *r, const fe *a, const fe *b) {
    fe t, u;
    u64 cy = fe_add_raw(&t, a, b);
    u64 br = fe_sub_raw(&u, &t, &c->m);
    *r = (cy || !br) ? u : t;
}
static void mod_sub(const mctx *c, fe *r, const fe *a, const fe *b) {
    fe t;
    if (fe_sub_raw(&t, a, b)) fe_add_raw(&t, &t, &c->m);
    *r = t;
}
static void mod_neg(const mctx *c, fe *r, const fe *a) {
    if (fe_is_zero(a)) *r = *a;
    else fe_sub_raw(r, &c->m, a);
}
/* Montgomery product r = a*b/R mod m (CIOS) */
static void mont_mul(const mctx *c, fe *r, const fe *a, const fe *b) {
    const int nl = c->nl;
    u64 t[NL + 2] = {0};
    for (int i = 0; i < nl; i++) {
        u128 cy = 0;
        for (int j = 0; j < nl; j++) cy += (u128)a->v[i] * b->v[j] + t[j], t[j] = (u64)cy, cy >>= 64;
        cy += t[nl], t[nl] = (u64)cy, t[nl + 1] = (u64)(cy >> 64);
        u64 mm = t[0] * c->n0;
        cy = (u128)mm * c->m.v[0] + t[0], cy >>= 64;
        for (int j = 1; j < nl; j++) cy += (u128)mm * c->m.v[j] + t[j], t[j - 1] = (u64)cy, cy >>= 64;
        cy += t[nl], t[nl - 1] = (u64)cy, t[nl] = t[nl + 1] + (u64)(cy >> 64);
    }
    fe x, y;
    memset(&x, 0, sizeof x);
    for (int j = 0; j < nl; j++) x.v[j] = t[j];
    u64 br = 0;
    memset(&y, 0, sizeof y);
    for (int j = 0; j < nl; j++) {
        u128 d = (u128)x.v[j] - c->m.v[j] - br;
        y.v[j] = (u64)d, br = (u64)(d >> 64) & 1;
    }
    *r = (t[nl] || !br) ? y : x;
}
static void to_mont(const mctx *c, fe *r, const fe *a) { mont_mul(c, r, a, &c->r2); }
static void from_mont(const mctx *c, fe *r, const fe *a) {
    fe one;
    fe_set_u64(&one, 1);
    mont_mul(c, r, a, &one);
}
/* plain-domain product (both operands and result non-Montgomery) */
static void mod_mul(const mctx *c, fe *r, const fe *a, const fe *b) {
    fe t;
    mont_mul(c, &t, a, &c->r2);
    mont_mul(c, r, &t, b);
}
/* Montgomery-domain inverse by Fermat; inv(0) = 0 like the reference's invMod (big.ts:113-119, App. C item 9) */
static void mont_inv(const mctx *c, fe *r, const fe *a) {
    fe acc = c->one, base = *a;
    int nbits = 64 * c->nl;
    for (int i = 0; i < nbits; i++) {
        if ((c->m2.v[i / 64] >> (i % 64)) & 1) mont_mul(c, &acc, &acc, &base);
        mont_mul(c, &base, &base, &base);
    }
    *r = acc;
}
static void mod_inv(const mctx *c, fe *r, const fe *a) { /* plain domain */
    fe t;
    to_mont(c, &t, a);
    mont_inv(c, &t, &t);
    from_mont(c, r, &t);
}
/* reduce an arbitrary 320-bit value mod m (used for 256-bit inputs vs 256-bit moduli: at most a few subtractions) */
static void mod_reduce(const mctx *c, fe *r, const fe *a) {
    fe t = *a, u;
    while (fe_cmp(&t, &c->m) >= 0) fe_sub_raw(&u, &t, &c->m), t = u;
    *r = t;
}
static void mctx_init(mctx *c, const char *hex, int nl) {
    memset(c, 0, sizeof *c);
    c->nl = nl;
    int len = (int)strlen(hex);
    for (int i = 0; i < len; i++) {
        char ch = hex[len - 1 - i];
        u64 d = ch <= '9' ? ch - '0' : (ch | 32) - 'a' + 10;
        c->m.v[i / 16] |= d << (4 * (i % 16));
    }
    u64 inv = 1;
    for (int i = 0; i < 6; i++) inv *= 2 - c->m.v[0] * inv;
    c->n0 = (u64)0 - inv;
    /* R mod m by repeated doubling of 1, then R^2 by 64*nl more doublings */
    fe x;
    fe_set_u64(&x, 1);
    for (int i = 0; i < 64 * nl; i++) mod_add(c, &x, &x, &x);
    c->one = x;
    for (int i = 0; i < 64 * nl; i++) mod_add(c, &x, &x, &x);
    c->r2 = x;
    fe two;
    fe_set_u64(&two, 2);
    fe_sub_raw(&c->m2, &c->m, &two);
}

/* ------------------------------------------------------------------ groups (instances.ts:22-54) */
static mctx FP, FN, FT; /* P-256 field (= Tom scalar field q), P-256 order n, Tom field t */
#define FQ FP
static fe P256_B, P256_GX, P256_GY;          /* Montgomery mod p */
static fe TOM_A, TOM_D, TOM_GX, TOM_GY;      /* Montgomery mod t */
static pthread_once_t g_once = PTHREAD_ONCE_INIT;
static void fe_from_hex_mont(const mctx *c, fe *r, const char *hex) {
    mctx tmp;
    memset(&tmp, 0, sizeof tmp);
    int len = (int)strlen(hex);
    fe x;
    memset(&x, 0, sizeof x);
    for (int i = 0; i < len; i++) {
        char ch = hex[len - 1 - i];
        u64 d = ch <= '9' ? ch - '0' : (ch | 32) - 'a' + 10;
        x.v[i / 16] |= d << (4 * (i % 16));
    }
    to_mont(c, r, &x);
}
static void init_consts(void) {
    mctx_init(&FP, "ffffffff00000001000000000000000000000000ffffffffffffffffffffffff", 4);
    mctx_init(&FN, "ffffffff00000000ffffffffffffffffbce6faada7179e84f3b9cac2fc632551", 4);
    mctx_init(&FT, "3fffffffc000000040000000000000002ae382c7957cc4ff9713c3d82bc47d3af", 5);
    fe_from_hex_mont(&FP, &P256_B, "5ac635d8aa3a93e7b3ebbd55769886bc651d06b0cc53b0f63bce3c3e27d2604b");
    fe_from_hex_mont(&FP, &P256_GX, "6b17d1f2e12c4247f8bce6e563a440f277037d812deb33a0f4a13945d898c296");
    fe_from_hex_mont(&FP, &P256_GY, "4fe342e2fe1a7f9b8ee7eb4a7c0f9e162bce33576b315ececbb6406837bf51f5");
    fe_from_hex_mont(&FT, &TOM_A, "1abce3fd8e1d7a21252515332a512e09d4249bd5b1ec35e316c02254fe8cedf5d");
    fe_from_hex_mont(&FT, &TOM_D, "051781d9823abde00ec99295ba542c8b1401874bcbeb9e9c861174c7bca6a02aa");
    fe_from_hex_mont(&FT, &TOM_GX, "7907055d0a7d4abc3eafdc25d431d9659fbe007ee2d8ddc4e906206ea9ba4fdb");
    fe_from_hex_mont(&FT, &TOM_GY, "be231cb9f9bf18319c9f081141559b0a33dddccd2221f0464a9cd57081b01a01");
}
static void ensure_init(void) { pthread_once(&g_once, init_consts); }

/* ---- P-256, projective (X:Y:Z), Montgomery coordinates ---- */
typedef struct {
    fe x, y, z;
} ppt;
#define PM(r, a, b) mont_mul(&FP, r, a, b)
#define PA(r, a, b) mod_add(&FP, r, a, b)
#define PS(r, a, b) mod_sub(&FP, r, a, b)
static void p_identity(ppt *r) { /* weier.ts:50-52 */
    memset(r, 0, sizeof *r);
    r->y = FP.one;
}
static int p_is_identity(const ppt *a) { return fe_is_zero(&a->x) && !fe_is_zero(&a->y) && fe_is_zero(&a->z); } /* weier.ts:117-119 */
static void p_dbl(ppt *r, const ppt *a) { /* weier.ts:133-175 */
    fe t0, t1, t2, t3, x3, y3, z3;
    const fe *x = &a->x, *y = &a->y, *z = &a->z, *b = &P256_B;
    PM(&t0, x, x); PM(&t1, y, y); PM(&t2, z, z); PM(&t3, x, y); PA(&t3, &t3, &t3);
    PM(&z3, x, z); PA(&z3, &z3, &z3); PM(&y3, b, &t2); PS(&y3, &y3, &z3); PA(&x3, &y3, &y3);
    PA(&y3, &x3, &y3); PS(&x3, &t1, &y3); PA(&y3, &t1, &y3); PM(&y3, &x3, &y3); PM(&x3, &x3, &t3);
    PA(&t3, &t2, &t2); PA(&t2, &t2, &t3); PM(&z3, b, &z3); PS(&z3, &z3, &t2); PS(&z3, &z3, &t0);
    PA(&t3, &z3, &z3); PA(&z3, &z3, &t3); PA(&t3, &t0, &t0); PA(&t0, &t3, &t0); PS(&t0, &t0, &t2);
    PM(&t0, &t0, &z3); PA(&y3, &y3, &t0); PM(&t0, y, z); PA(&t0, &t0, &t0); PM(&z3, &t0, &z3);
    PS(&x3, &x3, &z3); PM(&z3, &t0, &t1); PA(&z3, &z3, &z3); PA(&z3, &z3, &z3);
    r->x = x3, r->y = y3, r->z = z3;
}
static void p_add(ppt *r, const ppt *p, const ppt *q) { /* weier.ts:176-230 */
    fe t0, t1, t2, t3, t4, x3, y3, z3;
    const fe *x1 = &p->x, *y1 = &p->y, *z1 = &p->z, *x2 = &q->x, *y2 = &q->y, *z2 = &q->z, *b = &P256_B;
    PM(&t0, x1, x2); PM(&t1, y1, y2); PM(&t2, z1, z2); PA(&t3, x1, y1); PA(&t4, x2, y2);
    PM(&t3, &t3, &t4); PA(&t4, &t0, &t1); PS(&t3, &t3, &t4); PA(&t4, y1, z1); PA(&x3, y2, z2);
    PM(&t4, &t4, &x3); PA(&x3, &t1, &t2); PS(&t4, &t4, &x3); PA(&x3, x1, z1); PA(&y3, x2, z2);
    PM(&x3, &x3, &y3); PA(&y3, &t0, &t2); PS(&y3, &x3, &y3); PM(&z3, b, &t2); PS(&x3, &y3, &z3);
    PA(&z3, &x3, &x3); PA(&x3, &x3, &z3); PS(&z3, &t1, &x3); PA(&x3, &t1, &x3); PM(&y3, b, &y3);
    PA(&t1, &t2, &t2); PA(&t2, &t1, &t2); PS(&y3, &y3, &t2); PS(&y3, &y3, &t0); PA(&t1, &y3, &y3);
    PA(&y3, &t1, &y3); PA(&t1, &t0, &t0); PA(&t0, &t1, &t0); PS(&t0, &t0, &t2); PM(&t1, &t4, &y3);
    PM(&t2, &t0, &y3); PM(&y3, &x3, &z3); PA(&y3, &y3, &t2); PM(&x3, &t3, &x3); PS(&x3, &x3, &t1);
    PM(&z3, &t4, &z3); PM(&t1, &t3, &t0); PA(&z3, &z3, &t1);
    r->x = x3, r->y = y3, r->z = z3;
}
static void p_neg(ppt *r, const ppt *a) {
    *r = *a;
    mod_neg(&FP, &r->y, &a->y);
}
static int p_eq(const ppt *a, const ppt *b) { /* weier.ts:120-128 */
    fe l, r_;
    PM(&l, &a->x, &b->z); PM(&r_, &b->x, &a->z);
    if (!fe_eq(&l, &r_)) return 0;
    PM(&l, &a->y, &b->z); PM(&r_, &b->y, &a->z);
    return fe_eq(&l, &r_);
}
/* weier.ts:231-243: returns 0 for identity; normalises in place */
static int p_to_affine(ppt *a) {
    if (p_is_identity(a)) {
        a->y = FP.one;
        return 0;
    }
    fe zi;
    mont_inv(&FP, &zi, &a->z);
    PM(&a->x, &a->x, &zi); PM(&a->y, &a->y, &zi);
    a->z = FP.one;
    return 1;
}
static int p_on_curve(const ppt *a) { /* weier.ts:56-70 with a = -3 */
    fe y2z, x3, z2, axz2, bz3, t, three;
    PM(&t, &a->y, &a->y); PM(&y2z, &t, &a->z);
    PM(&t, &a->x, &a->x); PM(&x3, &t, &a->x);
    PM(&z2, &a->z, &a->z);
    PM(&t, &a->x, &z2);
    PA(&three, &t, &t); PA(&axz2, &three, &t); /* 3*x*z^2 */
    PM(&t, &z2, &a->z); PM(&bz3, &P256_B, &t);
    PS(&t, &x3, &axz2); PA(&t, &t, &bz3);
    return fe_eq(&t, &y2z);
}
static int hexdigit(const fe *k, int i) { return (int)((k->v[i / 16] >> (4 * (i % 16))) & 15); }
/* group.ts:133-152: window-4, MSB first; leading zero digits only double the identity, so iterating
   all 64 digits gives the same point as iterating k.toString(16). */
static void p_mul(ppt *r, const ppt *p, const fe *k) {
    ppt tab[16], q;
    p_identity(&tab[0]);
    for (int i = 1; i < 16; i++) p_add(&tab[i], &tab[i - 1], p);
    p_identity(&q);
    int top = 63;
    while (top > 0 && hexdigit(k, top) == 0) top--;
    for (int i = top; i >= 0; i--) {
        p_dbl(&q, &q); p_dbl(&q, &q); p_dbl(&q, &q); p_dbl(&q, &q);
        p_add(&q, &q, &tab[hexdigit(k, i)]);
    }
    *r = q;
}
static void p_dblmul(ppt *r, const ppt *p1, const fe *k1, const ppt *p2, const fe *k2) { /* group.ts:97-132 */
    ppt t1[16], t2[16], q;
    p_identity(&t1[0]); p_identity(&t2[0]);
    for (int i = 1; i < 16; i++) p_add(&t1[i], &t1[i - 1], p1), p_add(&t2[i], &t2[i - 1], p2);
    p_identity(&q);
    int top = 63;
    while (top > 0 && hexdigit(k1, top) == 0 && hexdigit(k2, top) == 0) top--;
    for (int i = top; i >= 0; i--) {
        p_dbl(&q, &q); p_dbl(&q, &q); p_dbl(&q, &q); p_dbl(&q, &q);
        p_add(&q, &q, &t1[hexdigit(k1, i)]);
        p_add(&q, &q, &t2[hexdigit(k2, i)]);
    }
    *r = q;
}
/* affine big-endian 64 B (x||y) <-> point */
static int p_from_xy(ppt *r, const uint8_t *xy) { /* weier.ts:74-89 */
    fe x, y;
    fe_from_be(&x, xy, 32); fe_from_be(&y, xy + 32, 32);
    if (fe_cmp(&x, &FP.m) >= 0 || fe_cmp(&y, &FP.m) >= 0) {
        /* the reference does not range-check Weierstrass coordinates; isOnGroup works mod p.  Reduce. */
        mod_reduce(&FP, &x, &x); mod_reduce(&FP, &y, &y);
    }
    to_mont(&FP, &r->x, &x); to_mont(&FP, &r->y, &y);
    r->z = FP.one;
    return p_on_curve(r);
}
static void p_affine_bytes(ppt *a, uint8_t *out64) { /* caller guarantees non-identity */
    fe x, y;
    p_to_affine(a);
    from_mont(&FP, &x, &a->x); from_mont(&FP, &y, &a->y);
    fe_to_be(&x, out64, 32); fe_to_be(&y, out64 + 32, 32);
}

/* ---- Tom-256, extended (X,Y,T,Z), Montgomery coordinates ---- */
typedef struct {
    fe x, y, t, z;
} tpt;
#define TM(r, a, b) mont_mul(&FT, r, a, b)
#define TA(r, a, b) mod_add(&FT, r, a, b)
#define TS(r, a, b) mod_sub(&FT, r, a, b)
static void t_identity(tpt *r) { /* edwards.ts:46-48 */
    memset(r, 0, sizeof *r);
    r->y = FT.one, r->z = FT.one;
}
static void t_dbl(tpt *r, const tpt *p) { /* edwards.ts:141-160 */
    fe A, B, C, D, E, G, F, H, xy;
    TM(&A, &p->x, &p->x); TM(&B, &p->y, &p->y); TM(&C, &p->z, &p->z); TA(&C, &C, &C);
    TM(&D, &TOM_A, &A); TA(&xy, &p->x, &p->y); TM(&E, &xy, &xy); TS(&E, &E, &A); TS(&E, &E, &B);
    TA(&G, &D, &B); TS(&F, &G, &C); TS(&H, &D, &B);
    TM(&r->x, &E, &F); TM(&r->y, &G, &H); TM(&r->t, &E, &H); TM(&r->z, &F, &G);
}
static void t_add(tpt *r, const tpt *p, const tpt *q) { /* edwards.ts:161-183 */
    fe A, B, C, D, E, F, G, H, e1, e2, aA;
    TM(&A, &p->x, &q->x); TM(&B, &p->y, &q->y); TM(&C, &TOM_D, &p->t); TM(&C, &C, &q->t); TM(&D, &p->z, &q->z);
    TA(&e1, &p->x, &p->y); TA(&e2, &q->x, &q->y); TM(&E, &e1, &e2); TS(&E, &E, &A); TS(&E, &E, &B);
    TS(&F, &D, &C); TA(&G, &D, &C); TM(&aA, &TOM_A, &A); TS(&H, &B, &aA);
    TM(&r->x, &E, &F); TM(&r->y, &G, &H); TM(&r->t, &E, &H); TM(&r->z, &F, &G);
}
static void t_neg(tpt *r, const tpt *a) {
    *r = *a;
    mod_neg(&FT, &r->x, &a->x); mod_neg(&FT, &r->t, &a->t);
}
static void t_sub(tpt *r, const tpt *a, const tpt *b) {
    tpt nb;
    t_neg(&nb, b);
    t_add(r, a, &nb);
}
static int t_eq(const tpt *a, const tpt *b) { /* edwards.ts:126-135 */
    fe l, r_;
    TM(&l, &a->x, &b->z); TM(&r_, &b->x, &a->z);
    if (!fe_eq(&l, &r_)) return 0;
    TM(&l, &a->y, &b->z); TM(&r_, &b->y, &a->z);
    return fe_eq(&l, &r_);
}
static int t_is_identity(const tpt *a) { /* edwards.ts:117-125 */
    return fe_is_zero(&a->x) && !fe_is_zero(&a->y) && fe_is_zero(&a->t) && !fe_is_zero(&a->z) && fe_eq(&a->y, &a->z);
}
static void t_to_affine(tpt *a) { /* edwards.ts:184-193 */
    fe zi;
    mont_inv(&FT, &zi, &a->z);
    TM(&a->x, &a->x, &zi); TM(&a->y, &a->y, &zi); TM(&a->t, &a->x, &a->y);
    a->z = FT.one;
}
static int t_on_curve(const tpt *a) { /* edwards.ts:52-65 */
    fe x2, y2, t2, z2, l0, r0, l1, r1;
    TM(&x2, &a->x, &a->x); TM(&y2, &a->y, &a->y); TM(&t2, &a->t, &a->t); TM(&z2, &a->z, &a->z);
    TM(&l0, &TOM_A, &x2); TA(&l0, &l0, &y2); TM(&r0, &TOM_D, &t2); TA(&r0, &r0, &z2);
    TM(&l1, &a->x, &a->y); TM(&r1, &a->z, &a->t);
    return fe_eq(&l0, &r0) && fe_eq(&l1, &r1);
}
static void t_mul(tpt *r, const tpt *p, const fe *k) { /* group.ts:133-152 */
    tpt tab[16], q;
    t_identity(&tab[0]);
    for (int i = 1; i < 16; i++) t_add(&tab[i], &tab[i - 1], p);
    t_identity(&q);
    int top = 63;
    while (top > 0 && hexdigit(k, top) == 0) top--;
    for (int i = top; i >= 0; i--) {
        t_dbl(&q, &q); t_dbl(&q, &q); t_dbl(&q, &q); t_dbl(&q, &q);
        t_add(&q, &q, &tab[hexdigit(k, i)]);
    }
    *r = q;
}
static void t_dblmul(tpt *r, const tpt *p1, const fe *k1, const tpt *p2, const fe *k2) { /* group.ts:97-132 */
    tpt t1[16], t2[16], q;
    t_identity(&t1[0]); t_identity(&t2[0]);
    for (int i = 1; i < 16; i++) t_add(&t1[i], &t1[i - 1], p1), t_add(&t2[i], &t2[i - 1], p2);
    t_identity(&q);
    int top = 63;
    while (top > 0 && hexdigit(k1, top) == 0 && hexdigit(k2, top) == 0) top--;
    for (int i = top; i >= 0; i--) {
        t_dbl(&q, &q); t_dbl(&q, &q); t_dbl(&q, &q); t_dbl(&q, &q);
        t_add(&q, &q, &t1[hexdigit(k1, i)]);
        t_add(&q, &q, &t2[hexdigit(k2, i)]);
    }
    *r = q;
}
#define TB 36 /* ZKA1 Tom coordinate width */
static int t_from_xy(tpt *r, const uint8_t *xy) { /* edwards.ts:70-86 (36-byte zero-padded coordinates) */
    fe x, y;
    fe_from_be(&x, xy, TB); fe_from_be(&y, xy + TB, TB);
    if (fe_cmp(&x, &FT.m) >= 0 || fe_cmp(&y, &FT.m) >= 0) return 0;
    to_mont(&FT, &r->x, &x); to_mont(&FT, &r->y, &y);
    TM(&r->t, &r->x, &r->y);
    r->z = FT.one;
    return t_on_curve(r);
}
static void t_affine_xy(tpt *a, fe *x, fe *y) {
    t_to_affine(a);
    from_mont(&FT, x, &a->x); from_mont(&FT, y, &a->y);
}
static void t_affine_bytes(tpt *a, uint8_t *out72) {
    fe x, y;
    t_affine_xy(a, &x, &y);
    fe_to_be(&x, out72, TB); fe_to_be(&y, out72 + TB, TB);
}

/* ------------------------------------------------------------------ hashPoints (group.ts:221-233) */
typedef struct {
    sha256_t s;
} hp_t;
static void hp_init(hp_t *h) { sha256_init(&h->s); }
static void hp_p(hp_t *h, ppt *a) { /* weier.ts:244-255 */
    uint8_t b[65];
    if (!p_to_affine(a)) {
        b[0] = 0;
        sha256_update(&h->s, b, 1);
        return;
    }
    b[0] = 4;
    p_affine_bytes(a, b + 1);
    sha256_update(&h->s, b, 65);
}
static void hp_t_(hp_t *h, tpt *a) { /* edwards.ts:194-203: 33-byte coordinates */
    uint8_t b[67];
    fe x, y;
    t_affine_xy(a, &x, &y);
    b[0] = 4;
    fe_to_be(&x, b + 1, 33); fe_to_be(&y, b + 34, 33);
    sha256_update(&h->s, b, 67);
}
static void hp_final(hp_t *h, fe *c) {
    uint8_t d[32];
    sha256_final(&h->s, d);
    fe_from_be(c, d, 10);
}

/* ------------------------------------------------------------------ RNG (big.ts:171-181 under the contract) */
typedef struct {
    int mode;            /* 0 = seed, 1 = explicit stream */
    const uint8_t *data; /* 32-byte seed, or blocks */
    u64 nblocks;         /* stream mode */
    u64 k;
    int err;
} rng_t;
static void rng_fill32(rng_t *g, uint8_t out[32]) {
    if (g->mode == 0) {
        uint8_t m[40];
        memcpy(m, g->data, 32);
        for (int i = 0; i < 8; i++) m[32 + i] = (uint8_t)(g->k >> (56 - 8 * i));
        zko_sha256(m, 40, out);
    } else if (g->k < g->nblocks) {
        memcpy(out, g->data + 32 * g->k, 32);
    } else {
        memset(out, 0, 32);
        g->err = 1;
    }
    g->k++;
}
static void rnd_mod(rng_t *g, const mctx *c, fe *r) {
    uint8_t b[32];
    for (;;) {
        rng_fill32(g, b);
        fe_from_be(r, b, 32);
        if (fe_cmp(r, &c->m) < 0 || g->err) return;
    }
}

/* ------------------------------------------------------------------ Pedersen (pedersen.ts) */
typedef struct {
    tpt g, h;
} tparams;
typedef struct {
    tpt p;
    fe r; /* plain scalar mod q */
} tcom;
static void tom_commit(const tparams *pp, rng_t *g, tcom *out, const fe *v) { /* pedersen.ts:53-58 */
    rnd_mod(g, &FQ, &out->r);
    t_dblmul(&out->p, &pp->h, &out->r, &pp->g, v);
}

/* ------------------------------------------------------------------ ZKA1 writer helpers */
typedef struct {
    uint8_t *p;
    u64 cap, off;
    int err;
} wr_t;
static uint8_t *wr_take(wr_t *w, u64 n) {
    static uint8_t sink[128];
    if (w->off + n > w->cap) {
        w->err = 1;
        return sink;
    }
    uint8_t *r = w->p + w->off;
    w->off += n;
    return r;
}
static void wr_sc(wr_t *w, const fe *s) { fe_to_be(s, wr_take(w, 32), 32); }
static void wr_tp(wr_t *w, tpt *a) { t_affine_bytes(a, wr_take(w, 2 * TB)); }
static void wr_pp(wr_t *w, ppt *a) { p_affine_bytes(a, wr_take(w, 64)); }

/* ------------------------------------------------------------------ equality.ts:60-78 */
static void prove_equality(const tparams *pp, rng_t *g, wr_t *w, const fe *x, tcom *C1, tcom *C2) {
    fe k, c, t, tx, tr1, tr2;
    tcom A1, A2;
    rnd_mod(g, &FQ, &k);
    tom_commit(pp, g, &A1, &k);
    tom_commit(pp, g, &A2, &k);
    hp_t h;
    hp_init(&h);
    hp_t_(&h, &C1->p); hp_t_(&h, &C2->p); hp_t_(&h, &A1.p); hp_t_(&h, &A2.p);
    hp_final(&h, &c);
    mod_mul(&FQ, &t, &c, x); mod_sub(&FQ, &tx, &k, &t);
    mod_mul(&FQ, &t, &c, &C1->r); mod_sub(&FQ, &tr1, &A1.r, &t);
    mod_mul(&FQ, &t, &c, &C2->r); mod_sub(&FQ, &tr2, &A2.r, &t);
    wr_tp(w, &A1.p); wr_tp(w, &A2.p); wr_sc(w, &tx); wr_sc(w, &tr1); wr_sc(w, &tr2);
}
/* ------------------------------------------------------------------ mult.ts:93-131 */
static void prove_mult(const tparams *pp, rng_t *g, wr_t *w, const fe *x, const fe *y, const fe *z, tcom *Cx, tcom *Cy, tcom *Cz) {
    fe r4, kx, ky, kz, c, t, tx, ty, tz, trx, try_, trz, tr4;
    tpt C4, A42;
    tcom Ax, Ay, Az, A41;
    t_mul(&C4, &Cy->p, x);
    mod_mul(&FQ, &r4, &Cy->r, x);
    rnd_mod(g, &FQ, &kx); rnd_mod(g, &FQ, &ky); rnd_mod(g, &FQ, &kz);
    tom_commit(pp, g, &Ax, &kx); tom_commit(pp, g, &Ay, &ky); tom_commit(pp, g, &Az, &kz); tom_commit(pp, g, &A41, &kz);
    t_mul(&A42, &Cy->p, &kx);
    hp_t h;
    hp_init(&h);
    hp_t_(&h, &Cx->p); hp_t_(&h, &Cy->p); hp_t_(&h, &Cz->p); hp_t_(&h, &C4);
    hp_t_(&h, &Ax.p); hp_t_(&h, &Ay.p); hp_t_(&h, &Az.p); hp_t_(&h, &A41.p); hp_t_(&h, &A42);
    hp_final(&h, &c);
    mod_mul(&FQ, &t, &c, x); mod_sub(&FQ, &tx, &kx, &t);
    mod_mul(&FQ, &t, &c, y); mod_sub(&FQ, &ty, &ky, &t);
    mod_mul(&FQ, &t, &c, z); mod_sub(&FQ, &tz, &kz, &t);
    mod_mul(&FQ, &t, &c, &Cx->r); mod_sub(&FQ, &trx, &Ax.r, &t);
    mod_mul(&FQ, &t, &c, &Cy->r); mod_sub(&FQ, &try_, &Ay.r, &t);
    mod_mul(&FQ, &t, &c, &Cz->r); mod_sub(&FQ, &trz, &Az.r, &t);
    mod_mul(&FQ, &t, &c, &r4); mod_sub(&FQ, &tr4, &A41.r, &t);
    wr_tp(w, &C4); wr_tp(w, &Ax.p); wr_tp(w, &Ay.p); wr_tp(w, &Az.p); wr_tp(w, &A41.p); wr_tp(w, &A42);
    wr_sc(w, &tx); wr_sc(w, &ty); wr_sc(w, &tz); wr_sc(w, &trx); wr_sc(w, &try_); wr_sc(w, &trz); wr_sc(w, &tr4);
}
static void tcom_sub(tcom *r, const tcom *a, const tcom *b) { /* pedersen.ts:33-35 */
    t_sub(&r->p, &a->p, &b->p);
    mod_sub(&FQ, &r->r, &a->r, &b->r);
}
/* ------------------------------------------------------------------ pointAdd.ts:92-163 */
static int prove_point_add(const tparams *pp, rng_t *g, wr_t *w, ppt *P, ppt *Q, ppt *R, tcom *PX, tcom *PY, tcom *QX, tcom *QY, tcom *RX, tcom *RY) {
    ppt s;
    p_add(&s, P, Q);
    if (!p_eq(&s, R)) return ZK_E_POINTS_DONT_ADD;
    if (!p_to_affine(P) || !p_to_affine(Q) || !p_to_affine(R)) return ZK_E_PADD_INF;
    fe x1, y1, x2, y2, x3, i7, i8, i9, i10, i11, i12, i13, one;
    from_mont(&FP, &x1, &P->x); from_mont(&FP, &y1, &P->y); from_mont(&FP, &x2, &Q->x); from_mont(&FP, &y2, &Q->y); from_mont(&FP, &x3, &R->x);
    /* coordinates are < p = q, so they are already scalars of the proof group */
    mod_sub(&FQ, &i7, &x2, &x1); mod_inv(&FQ, &i8, &i7); mod_sub(&FQ, &i9, &y2, &y1);
    mod_mul(&FQ, &i10, &i8, &i9); mod_mul(&FQ, &i11, &i10, &i10); mod_sub(&FQ, &i12, &x1, &x3); mod_mul(&FQ, &i13, &i10, &i12);
    tcom *C1 = PX, *C2 = QX, *C3 = RX, *C4 = PY, *C5 = QY, *C6 = RY, C7, C8, C9, C10, C11, C12, C13, C14, Cint;
    tcom_sub(&C7, C2, C1);
    tom_commit(pp, g, &C8, &i8);
    tcom_sub(&C9, C5, C4);
    tom_commit(pp, g, &C10, &i10);
    tom_commit(pp, g, &C11, &i11);
    tcom_sub(&C12, C1, C3);
    tom_commit(pp, g, &C13, &i13);
    C14.p = pp->g;
    fe_set_u64(&C14.r, 0);
    fe_set_u64(&one, 1);
    wr_tp(w, &C8.p); wr_tp(w, &C10.p); wr_tp(w, &C11.p); wr_tp(w, &C13.p);
    /* layout order: pi8, pi10, pi11, pi13, pix, piy; draw order: pi8, pi10, pi11, pix, pi13, piy */
    u64 off_pix = w->off + 4 * (6 * 2 * TB + 7 * 32);
    prove_mult(pp, g, w, &i7, &i8, &one, &C7, &C8, &C14);
    prove_mult(pp, g, w, &i8, &i9, &i10, &C8, &C9, &C10);
    prove_mult(pp, g, w, &i10, &i10, &i11, &C10, &C10, &C11);
    t_add(&Cint.p, &C3->p, &C1->p); t_add(&Cint.p, &Cint.p, &C2->p);
    mod_add(&FQ, &Cint.r, &C3->r, &C1->r); mod_add(&FQ, &Cint.r, &Cint.r, &C2->r);
    wr_t wx = *w;
    wx.off = off_pix;
    prove_equality(pp, g, &wx, &i11, &C11, &Cint);
    prove_mult(pp, g, w, &i10, &i12, &i13, &C10, &C12, &C13);
    t_add(&Cint.p, &C6->p, &C4->p);
    mod_add(&FQ, &Cint.r, &C6->r, &C4->r);
    if (w->off != off_pix) w->err |= (w->cap >= off_pix); /* only an overflowing writer can disagree */
    w->off = wx.off;
    w->err |= wx.err;
    prove_equality(pp, g, w, &i13, &C13, &Cint);
    return ZK_OK;
}

/* ------------------------------------------------------------------ interpolate.ts:27-70 (plain domain mod q) */
static int interpolate_q(int n, const fe *y, fe *coeff) {
    /* x = [0..n-1].  The reference's intermediate s[] are signed JS remainders; values agree mod q. */
    fe s[65], xi, phi, ff, b, t, jj;
    if (n == 0) return ZK_OK;
    for (int i = 0; i <= n; i++) fe_set_u64(&s[i], 0);
    for (int i = 0; i < n; i++) fe_set_u64(&coeff[i], 0);
    fe_set_u64(&s[n], 1);
    fe_set_u64(&s[n - 1], 0); /* -x[0] = 0 */
    for (int i = 1; i < n; i++) {
        fe_set_u64(&xi, (u64)i);
        for (int j = n - i - 1; j < n - 1; j++) {
            mod_mul(&FQ, &t, &xi, &s[j + 1]);
            mod_sub(&FQ, &s[j], &s[j], &t);
        }
        mod_sub(&FQ, &s[n - 1], &s[n - 1], &xi);
    }
    for (int i = 0; i < n; i++) {
        fe_set_u64(&xi, (u64)i);
        fe_set_u64(&phi, 0);
        for (int j = n; j >= 1; j--) {
            fe_set_u64(&jj, (u64)j);
            mod_mul(&FQ, &phi, &xi, &phi);
            mod_mul(&FQ, &t, &jj, &s[j]);
            mod_add(&FQ, &phi, &phi, &t);
        }
        mod_inv(&FQ, &ff, &phi);
        fe_set_u64(&b, 1);
        for (int j = n - 1; j >= 0; j--) {
            mod_mul(&FQ, &t, &b, &ff);
            mod_mul(&FQ, &t, &t, &y[i]);
            mod_add(&FQ, &coeff[j], &coeff[j], &t);
            mod_mul(&FQ, &b, &xi, &b);
            mod_add(&FQ, &b, &b, &s[j]);
        }
    }
    for (int i = 0; i < n; i++) { /* self-check, interpolate.ts:63-67 */
        fe ret;
        fe_set_u64(&ret, 0);
        fe_set_u64(&xi, (u64)i);
        for (int k = n - 1; k >= 0; k--) {
            mod_mul(&FQ, &ret, &xi, &ret);
            mod_add(&FQ, &ret, &ret, &coeff[k]);
        }
        if (!fe_eq(&ret, &y[i])) return ZK_E_INTERPOLATION;
    }
    return ZK_OK;
}
/* ------------------------------------------------------------------ gk.ts:94-195 */
static void fe_pow_small(fe *r, const fe *x, int e) { /* expMod(x, e, q), big.ts:44-59 */
    fe acc, b = *x;
    fe_set_u64(&acc, 1);
    while (e > 0) {
        if (e & 1) mod_mul(&FQ, &acc, &acc, &b);
        mod_mul(&FQ, &b, &b, &b);
        e >>= 1;
    }
    *r = acc;
}
static int prove_membership(const tparams *pp, rng_t *g, wr_t *w, const tcom *com, u64 index, const fe *values_m, u64 N, int n, fe *scratch_p) {
    fe eli[64], ri[64], ai[64], si[64], ti[64], rho[64], dv[64], di[64], t, u;
    tpt cl[64], ca[64], cb[64], cd[64];
    for (int i = 0; i < n; i++) fe_set_u64(&eli[i], (index >> i) & 1);
    for (int i = 0; i < n; i++) {
        rnd_mod(g, &FQ, &ri[i]); rnd_mod(g, &FQ, &ai[i]); rnd_mod(g, &FQ, &si[i]); rnd_mod(g, &FQ, &ti[i]); rnd_mod(g, &FQ, &rho[i]);
    }
    for (int i = 0; i < n; i++) { /* gk.ts:88-92: g.dblmul(val, h, blinder) */
        t_dblmul(&cl[i], &pp->g, &eli[i], &pp->h, &ri[i]);
        t_dblmul(&ca[i], &pp->g, &ai[i], &pp->h, &si[i]);
        mod_mul(&FQ, &t, &eli[i], &ai[i]);
        t_dblmul(&cb[i], &pp->g, &t, &pp->h, &ti[i]);
    }
    fe *p = scratch_p;
    for (int wi = 0; wi < n; wi++) { /* gk.ts:141-171 */
        fe wv, f0[64], f1[64], ratio[64], prod, dval, one;
        fe_set_u64(&wv, (u64)wi);
        fe_set_u64(&one, 1);
        for (int j = 0; j < n; j++) {
            mod_sub(&FQ, &t, &one, &eli[j]); mod_mul(&FQ, &t, &t, &wv); mod_sub(&FQ, &f0[j], &t, &ai[j]);
            mod_mul(&FQ, &t, &eli[j], &wv); mod_add(&FQ, &f1[j], &t, &ai[j]);
            mod_inv(&FQ, &t, &f0[j]); mod_mul(&FQ, &ratio[j], &f1[j], &t);
        }
        fe_set_u64(&prod, 1);
        for (int j = 0; j < n; j++) mod_mul(&FQ, &prod, &prod, &f0[j]);
        p[0] = prod;
        u64 len = 1;
        /* keep ratio in Montgomery form so the N-element doubling costs one mont_mul per element */
        for (int i = 0; i < n; i++) {
            fe rm;
            to_mont(&FQ, &rm, &ratio[i]);
            for (u64 j = 0; j < len; j++) mont_mul(&FQ, &p[len + j], &rm, &p[j]);
            len <<= 1;
        }
        fe_set_u64(&dval, 0);
        fe vlm = values_m[index];
        for (u64 i = 0; i < N; i++) {
            fe dm;
            mod_sub(&FQ, &dm, &vlm, &values_m[i]);
            mont_mul(&FQ, &u, &dm, &p[i]);
            mod_add(&FQ, &dval, &dval, &u);
        }
        dv[wi] = dval;
    }
    int rc = interpolate_q(n, dv, di);
    if (rc) return rc;
    for (int i = 0; i < n; i++) t_dblmul(&cd[i], &pp->g, &di[i], &pp->h, &rho[i]);
    hp_t h;
    hp_init(&h);
    for (int i = 0; i < n; i++) hp_t_(&h, &cl[i]);
    for (int i = 0; i < n; i++) hp_t_(&h, &ca[i]);
    for (int i = 0; i < n; i++) hp_t_(&h, &cb[i]);
    for (int i = 0; i < n; i++) hp_t_(&h, &cd[i]);
    fe x, f[64], za[64], zb[64], zd, xp;
    hp_final(&h, &x);
    fe_pow_small(&xp, &x, n);
    mod_mul(&FQ, &zd, &com->r, &xp);
    for (int i = 0; i < n; i++) {
        mod_mul(&FQ, &t, &eli[i], &x); mod_add(&FQ, &f[i], &t, &ai[i]);
        mod_mul(&FQ, &t, &ri[i], &x); mod_add(&FQ, &za[i], &t, &si[i]);
        mod_sub(&FQ, &t, &x, &f[i]); mod_mul(&FQ, &t, &ri[i], &t); mod_add(&FQ, &zb[i], &t, &ti[i]);
    }
    for (int i = 0; i < n; i++) {
        fe_pow_small(&xp, &x, i);
        mod_mul(&FQ, &t, &rho[i], &xp);
        mod_sub(&FQ, &zd, &zd, &t);
    }
    for (int i = 0; i < n; i++) wr_tp(w, &cl[i]);
    for (int i = 0; i < n; i++) wr_tp(w, &ca[i]);
    for (int i = 0; i < n; i++) wr_tp(w, &cb[i]);
    for (int i = 0; i < n; i++) wr_tp(w, &cd[i]);
    for (int i = 0; i < n; i++) wr_sc(w, &f[i]);
    for (int i = 0; i < n; i++) wr_sc(w, &za[i]);
    for (int i = 0; i < n; i++) wr_sc(w, &zb[i]);
    wr_sc(w, &zd);
    return ZK_OK;
}

/* ------------------------------------------------------------------ context */
typedef struct {
    ppt nist_h;
    tparams tom;
    uint32_t sec;
    fe *ring;    /* padded, plain scalars mod q */
    fe *ring_m;  /* same, Montgomery form */
    u64 N;       /* padded length */
    int n;       /* log2 N */
} zko_ctx;

zko_ctx *zko_ctx_create(void) {
    ensure_init();
    return (zko_ctx *)calloc(1, sizeof(zko_ctx));
}
void zko_ctx_destroy(zko_ctx *c) {
    if (c) free(c->ring), free(c->ring_m), free(c);
}
int zko_ctx_set_params(zko_ctx *c, const uint8_t nist_h[64], const uint8_t tom_g[72], const uint8_t tom_h[72], uint32_t sec) {
    if (sec == 0 || sec > 128) return ZK_E_SECLEVEL;
    if (!p_from_xy(&c->nist_h, nist_h)) return ZK_E_POINT_NOT_IN_GROUP;
    if (!t_from_xy(&c->tom.g, tom_g) || !t_from_xy(&c->tom.h, tom_h)) return ZK_E_POINT_NOT_IN_GROUP;
    c->sec = sec;
    return ZK_OK;
}
int zko_ctx_set_ring(zko_ctx *c, const uint8_t *keys_be32, u64 nkeys) { /* gk.ts:75-86 */
    if (nkeys == 0) return ZK_E_BUFFER;
    int n = 0;
    while (((u64)1 << n) < nkeys) n++;
    u64 N = (u64)1 << n;
    free(c->ring); free(c->ring_m);
    c->ring = (fe *)malloc(sizeof(fe) * N);
    c->ring_m = (fe *)malloc(sizeof(fe) * N);
    for (u64 i = 0; i < N; i++) {
        fe v;
        fe_from_be(&v, keys_be32 + 32 * (i < nkeys ? i : 0), 32);
        mod_reduce(&FQ, &c->ring[i], &v); /* Scalar ctor reduces, group.ts:164-167 */
        to_mont(&FQ, &c->ring_m[i], &c->ring[i]);
    }
    c->N = N, c->n = n;
    return ZK_OK;
}

/* ------------------------------------------------------------------ zkpAttestList.ts:104-145 + exp.ts:126-231 */
#define MAXSEC 128
static int prove_one(const zko_ctx *c, const uint8_t *msg_hash, const uint8_t *sig, const uint8_t *pk_xy, uint32_t which, rng_t *g, wr_t *w, fe *scratch_p) {
    ppt pk, G, R, Q, t1p, t2p;
    if (!p_from_xy(&pk, pk_xy)) return ZK_E_POINT_NOT_IN_GROUP;
    fe z, r, s, sinv, u1, u2, rinv, s1, z1, pkx, pky;
    fe_from_be(&z, msg_hash, 32); /* truncateToN is a no-op for 32-byte hashes (zkpAttestList.ts:80-86) */
    fe_from_be(&r, sig, 32); fe_from_be(&s, sig + 32, 32);
    /* BigInt arithmetic: reduce operands first (posMod of products is what matters) */
    fe zr, rr, sr;
    mod_reduce(&FN, &zr, &z); mod_reduce(&FN, &rr, &r); mod_reduce(&FN, &sr, &s);
    mod_inv(&FN, &sinv, &sr); mod_mul(&FN, &u1, &sinv, &zr); mod_mul(&FN, &u2, &sinv, &rr);
    G.x = P256_GX, G.y = P256_GY, G.z = FP.one;
    p_mul(&t1p, &G, &u1); p_mul(&t2p, &pk, &u2); p_add(&R, &t1p, &t2p);
    mod_inv(&FN, &rinv, &rr); mod_mul(&FN, &s1, &rinv, &sr); mod_mul(&FN, &z1, &rinv, &zr);
    p_mul(&Q, &G, &z1);
    /* comS1 = paramsSigExp.commit(s1): h.dblmul(r, g=R, v=s1)  (pedersen.ts:53-58) */
    fe comS1_r;
    ppt comS1;
    rnd_mod(g, &FN, &comS1_r);
    p_dblmul(&comS1, &c->nist_h, &comS1_r, &R, &s1);
    from_mont(&FP, &pkx, &pk.x); from_mont(&FP, &pky, &pk.y);
    tcom pkX, pkY;
    tom_commit(&c->tom, g, &pkX, &pkx);
    tom_commit(&c->tom, g, &pkY, &pky);

    const int sec = (int)c->sec;
    static __thread fe alpha[MAXSEC], rr_[MAXSEC];
    static __thread ppt T[MAXSEC], A[MAXSEC];
    static __thread tcom Tx[MAXSEC], Ty[MAXSEC];
    for (int i = 0; i < sec; i++) { /* exp.ts:144-156 */
        rnd_mod(g, &FN, &alpha[i]); rnd_mod(g, &FN, &rr_[i]);
        p_mul(&T[i], &R, &alpha[i]);
        ppt hr;
        p_mul(&hr, &c->nist_h, &rr_[i]);
        p_add(&A[i], &T[i], &hr);
        if (!p_to_affine(&T[i])) return ZK_E_T_INF;
        fe tx, ty;
        from_mont(&FP, &tx, &T[i].x); from_mont(&FP, &ty, &T[i].y);
        tom_commit(&c->tom, g, &Tx[i], &tx);
        tom_commit(&c->tom, g, &Ty[i], &ty);
    }
    hp_t h;
    hp_init(&h);
    hp_t_(&h, &pkX.p); hp_t_(&h, &pkY.p);
    for (int i = 0; i < sec; i++) hp_p(&h, &A[i]), hp_t_(&h, &Tx[i].p), hp_t_(&h, &Ty[i].p);
    fe challenge;
    hp_final(&h, &challenge);

    /* header */
    uint8_t *hdr = wr_take(w, 32);
    u64 start = w->off - 32;
    wr_pp(w, &R); wr_pp(w, &comS1); wr_tp(w, &pkX.p); wr_tp(w, &pkY.p);
    for (int i = 0; i < sec; i++) { /* exp.ts:168-229, bit i = (challenge >> i) & 1 */
        int bit = (int)((challenge.v[i / 64] >> (i % 64)) & 1);
        wr_pp(w, &A[i]); wr_tp(w, &Tx[i].p); wr_tp(w, &Ty[i].p);
        if (bit) {
            wr_sc(w, &alpha[i]); wr_sc(w, &rr_[i]); wr_sc(w, &Tx[i].r); wr_sc(w, &Ty[i].r);
        } else {
            fe zz, z2;
            mod_sub(&FN, &zz, &alpha[i], &s1);
            ppt T1;
            p_mul(&T1, &R, &zz);
            p_add(&T1, &T1, &Q);
            if (!p_to_affine(&T1)) return ZK_E_T1_INF;
            fe x, y;
            from_mont(&FP, &x, &T1.x); from_mont(&FP, &y, &T1.y);
            tcom T1x, T1y;
            tom_commit(&c->tom, g, &T1x, &x);
            tom_commit(&c->tom, g, &T1y, &y);
            mod_sub(&FN, &z2, &rr_[i], &comS1_r);
            wr_sc(w, &zz); wr_sc(w, &z2); wr_sc(w, &T1x.r); wr_sc(w, &T1y.r);
            ppt Pc = pk, Tc = T[i];
            int rc = prove_point_add(&c->tom, g, w, &T1, &Pc, &Tc, &T1x, &T1y, &pkX, &pkY, &Tx[i], &Ty[i]);
            if (rc) return rc;
        }
    }
    /* gk.ts:162 reads values[index].k with index = which: past the padded ring that is `undefined.k`, a TypeError -- thrown by
     * proveMembership, i.e. after proveExp and its exceptions (zkpAttestList.ts:141-142).  (A one-key ring never gets that far: n = 0 and
     * interpolate([], []) evaluates -x[0] % m with x[0] undefined, interpolate.ts:40, a TypeError for every `which`.) */
    if ((u64)which >= c->N) return ZK_E_ARG;
    int rc = prove_membership(&c->tom, g, w, &pkX, which, c->ring_m, c->N, c->n, scratch_p);
    if (rc) return rc;
    if (g->err) return ZK_E_RNG_EXHAUSTED;
    if (w->err) return ZK_E_BUFFER;
    u64 total = w->off - start;
    memcpy(hdr, "ZKA1", 4);
    hdr[4] = total >> 24, hdr[5] = total >> 16, hdr[6] = total >> 8, hdr[7] = total;
    hdr[8] = 0, hdr[9] = 0, hdr[10] = 0, hdr[11] = (uint8_t)sec;
    hdr[12] = 0, hdr[13] = 0, hdr[14] = 0, hdr[15] = (uint8_t)c->n;
    fe_to_be(&challenge, hdr + 16, 16);
    if (sec < 128) { /* keep only sec bits */
        fe m = challenge;
        for (int i = sec; i < 128; i++) m.v[i / 64] &= ~((u64)1 << (i % 64));
        m.v[2] = m.v[3] = m.v[4] = 0;
        fe_to_be(&m, hdr + 16, 16);
    }
    return ZK_OK;
}

u64 zko_max_proof_size(const zko_ctx *c) {
    u64 mult = 6 * 2 * TB + 7 * 32, eq = 2 * 2 * TB + 3 * 32, padd = 4 * 2 * TB + 4 * mult + 2 * eq;
    u64 rep = 64 + 2 * 2 * TB + 4 * 32 + padd;
    return 32 + 128 + 4 * TB + c->sec * rep + (u64)c->n * (4 * 2 * TB + 3 * 32) + 32;
}

typedef struct {
    const zko_ctx *c;
    u64 B, lo, hi;
    const uint8_t *msg, *sig, *pk, *rng_data;
    const uint32_t *which;
    int rng_mode;
    u64 rng_stride_blocks;
    uint8_t *out;
    u64 slot;
    u64 *sizes;
    int32_t *status;
} job_t;
static void *prove_worker(void *arg) {
    job_t *j = (job_t *)arg;
    fe *scratch = (fe *)malloc(sizeof(fe) * (j->c->N ? j->c->N : 1));
    for (u64 b = j->lo; b < j->hi; b++) {
        rng_t g = {j->rng_mode, j->rng_mode == 0 ? j->rng_data + 32 * b : j->rng_data + 32 * j->rng_stride_blocks * b, j->rng_stride_blocks, 0, 0};
        wr_t w = {j->out + j->slot * b, j->slot, 0, 0};
        int rc = prove_one(j->c, j->msg + 32 * b, j->sig + 64 * b, j->pk + 64 * b, j->which[b], &g, &w, scratch);
        if (g.err) rc = ZK_E_RNG_EXHAUSTED; /* a short stream is a caller error, whatever it broke downstream */
        j->status[b] = rc;
        j->sizes[b] = rc ? 0 : w.off;
    }
    free(scratch);
    return NULL;
}
/* Proofs are written at out + b*slot (slot >= zko_max_proof_size); sizes[b] receives the byte length. */
int zko_prove_batch(const zko_ctx *c, u64 B, const uint8_t *msg, const uint8_t *sig, const uint8_t *pk, const uint32_t *which,
                    int rng_mode, const uint8_t *rng_data, u64 rng_stride_blocks, uint8_t *out, u64 slot, u64 *sizes, int32_t *status, int nthreads) {
    if (!c->ring || !c->sec) return ZK_E_BUFFER;
    if (nthreads < 1) nthreads = 1;
    if ((u64)nthreads > B) nthreads = (int)(B ? B : 1);
    pthread_t th[256];
    job_t jobs[256];
    if (nthreads > 256) nthreads = 256;
    for (int t = 0; t < nthreads; t++) {
        job_t j = {c, B, B * t / nthreads, B * (t + 1) / nthreads, msg, sig, pk, rng_data, which, rng_mode, rng_stride_blocks, out, slot, sizes, status};
        jobs[t] = j;
        if (nthreads == 1) prove_worker(&jobs[t]);
        else pthread_create(&th[t], NULL, prove_worker, &jobs[t]);
    }
    if (nthreads > 1)
        for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    return ZK_OK;
}

/* ------------------------------------------------------------------ verifier: multimult.ts */
typedef struct {
    int is_tom;
    union {
        ppt p;
        tpt t;
    } pt;
    fe s; /* plain scalar (mod n for P-256, mod q for Tom) */
} pair_t;
typedef struct {
    int is_tom;
    const mctx *order;
    pair_t *pairs;
    int len, cap;
    int known[8], nknown;
    rng_t *vr;
} mm_t;
static void mm_init(mm_t *m, int is_tom, rng_t *vr) {
    memset(m, 0, sizeof *m);
    m->is_tom = is_tom, m->order = is_tom ? &FQ : &FN, m->vr = vr;
    m->cap = 1024, m->pairs = (pair_t *)malloc(sizeof(pair_t) * m->cap);
}
static void mm_free(mm_t *m) { free(m->pairs); }
static int pt_eq(int is_tom, const void *a, const void *b) { return is_tom ? t_eq((const tpt *)a, (const tpt *)b) : p_eq((const ppt *)a, (const ppt *)b); }
static void mm_push(mm_t *m, const void *pt, const fe *s) {
    if (m->len == m->cap) m->cap *= 2, m->pairs = (pair_t *)realloc(m->pairs, sizeof(pair_t) * m->cap);
    pair_t *p = &m->pairs[m->len++];
    p->is_tom = m->is_tom;
    if (m->is_tom) p->pt.t = *(const tpt *)pt;
    else p->pt.p = *(const ppt *)pt;
    p->s = *s;
}
static void mm_add_known(mm_t *m, const void *pt) { /* multimult.ts:42-48 */
    for (int i = 0; i < m->nknown; i++)
        if (pt_eq(m->is_tom, pt, &m->pairs[m->known[i]].pt)) return;
    fe z;
    fe_set_u64(&z, 0);
    mm_push(m, pt, &z);
    m->known[m->nknown++] = m->len - 1;
}
static void mm_insert(mm_t *m, const void *pt, const fe *s) { /* multimult.ts:50-59 */
    for (int i = 0; i < m->nknown; i++)
        if (pt_eq(m->is_tom, pt, &m->pairs[m->known[i]].pt)) {
            mod_add(m->order, &m->pairs[m->known[i]].s, &m->pairs[m->known[i]].s, s);
            return;
        }
    mm_push(m, pt, s);
}
static void heap_swap(pair_t *a, pair_t *b) {
    pair_t t = *a;
    *a = *b, *b = t;
}
static void bubbleup(pair_t *arr, int index) { /* multimult.ts:113-124 */
    while (index > 1) {
        int parent = index / 2;
        if (fe_cmp(&arr[parent - 1].s, &arr[index - 1].s) < 0) heap_swap(&arr[parent - 1], &arr[index - 1]), index = parent;
        else return;
    }
}
static void pushdown(pair_t *arr, int len, int parent) { /* multimult.ts:126-145 */
    for (;;) {
        int son = 2 * parent, daughter = son + 1, child = son;
        if (son > len) return;
        if (daughter <= len && fe_cmp(&arr[daughter - 1].s, &arr[son - 1].s) > 0) child = daughter;
        if (fe_cmp(&arr[parent - 1].s, &arr[child - 1].s) < 0) heap_swap(&arr[parent - 1], &arr[child - 1]), parent = child;
        else return;
    }
}
static int pair_mul_is_identity(const pair_t *a) {
    if (a->is_tom) {
        tpt r;
        t_mul(&r, &a->pt.t, &a->s);
        return t_is_identity(&r);
    }
    ppt r;
    p_mul(&r, &a->pt.p, &a->s);
    return p_is_identity(&r);
}
static int mm_evaluate_is_identity(mm_t *m) { /* multimult.ts:61-89 */
    if (m->len == 0) return 1;
    if (m->len == 1) return pair_mul_is_identity(&m->pairs[0]);
    for (int i = 0; i < m->len; i++) bubbleup(m->pairs, i + 1);
    for (;;) {
        if (m->len == 1) return pair_mul_is_identity(&m->pairs[0]);
        heap_swap(&m->pairs[0], &m->pairs[m->len - 1]);
        pair_t a = m->pairs[--m->len];
        pushdown(m->pairs, m->len, 1);
        pair_t *b = &m->pairs[0];
        if (fe_is_zero(&b->s)) return pair_mul_is_identity(&a);
        pair_t cc = a;
        mod_sub(m->order, &cc.s, &a.s, &b->s);
        if (m->is_tom) t_add(&b->pt.t, &b->pt.t, &a.pt.t);
        else p_add(&b->pt.p, &b->pt.p, &a.pt.p);
        if (!fe_is_zero(&cc.s)) {
            if (m->len == m->cap) m->cap *= 2, m->pairs = (pair_t *)realloc(m->pairs, sizeof(pair_t) * m->cap);
            m->pairs[m->len++] = cc;
            bubbleup(m->pairs, m->len);
        }
    }
}
/* Relation (multimult.ts:147-174): collect, then drain with one fresh randomiser */
typedef struct {
    const void *pts[8];
    fe sc[8];
    int n;
} rel_t;
static void rel_ins(rel_t *r, const void *pt, const fe *s) { r->pts[r->n] = pt, r->sc[r->n] = *s, r->n++; }
static void rel_drain(rel_t *r, mm_t *m) {
    fe rz, t;
    rnd_mod(m->vr, m->order, &rz);
    for (int i = 0; i < r->n; i++) {
        mod_mul(m->order, &t, &r->sc[i], &rz);
        mm_insert(m, r->pts[i], &t);
    }
    r->n = 0;
}

/* ------------------------------------------------------------------ ZKA1 reader */
typedef struct {
    const uint8_t *p;
    u64 len, off;
    int err;
} rd_t;
static const uint8_t *rd_take(rd_t *r, u64 n) {
    static const uint8_t zeros[128];
    if (r->off + n > r->len) {
        r->err = ZK_E_BAD_ENCODING;
        return zeros;
    }
    const uint8_t *q = r->p + r->off;
    r->off += n;
    return q;
}
static void rd_pp(rd_t *r, ppt *a) {
    fe x, y;
    const uint8_t *b = rd_take(r, 64);
    fe_from_be(&x, b, 32); fe_from_be(&y, b + 32, 32);
    (void)x; (void)y;
    if (!p_from_xy(a, b)) r->err = ZK_E_BAD_ENCODING; /* weier.ts:256-260 (no range check in the reference) */
}
static void rd_tp(rd_t *r, tpt *a) {
    if (!t_from_xy(a, rd_take(r, 2 * TB))) r->err = ZK_E_BAD_ENCODING; /* edwards.ts:204-209 */
}
static void rd_sc(rd_t *r, const mctx *order, fe *s) {
    fe v;
    fe_from_be(&v, rd_take(r, 32), 32);
    mod_reduce(order, s, &v); /* Scalar ctor reduces */
}
typedef struct {
    tpt C4, Ax, Ay, Az, A41, A42;
    fe tx, ty, tz, trx, try_, trz, tr4;
} multp_t;
typedef struct {
    tpt A1, A2;
    fe tx, tr1, tr2;
} eqp_t;
typedef struct {
    tpt C8, C10, C11, C13;
    multp_t pi8, pi10, pi11, pi13;
    eqp_t pix, piy;
} paddp_t;
static void rd_mult(rd_t *r, multp_t *m) {
    rd_tp(r, &m->C4); rd_tp(r, &m->Ax); rd_tp(r, &m->Ay); rd_tp(r, &m->Az); rd_tp(r, &m->A41); rd_tp(r, &m->A42);
    rd_sc(r, &FQ, &m->tx); rd_sc(r, &FQ, &m->ty); rd_sc(r, &FQ, &m->tz); rd_sc(r, &FQ, &m->trx); rd_sc(r, &FQ, &m->try_); rd_sc(r, &FQ, &m->trz); rd_sc(r, &FQ, &m->tr4);
}
static void rd_eq(rd_t *r, eqp_t *e) {
    rd_tp(r, &e->A1); rd_tp(r, &e->A2);
    rd_sc(r, &FQ, &e->tx); rd_sc(r, &FQ, &e->tr1); rd_sc(r, &FQ, &e->tr2);
}
static void hash4(tpt *a, tpt *b, tpt *c, tpt *d, fe *out) {
    hp_t h;
    hp_init(&h);
    hp_t_(&h, a); hp_t_(&h, b); hp_t_(&h, c); hp_t_(&h, d);
    hp_final(&h, out);
}
static void aggregate_equality(const tparams *pp, tpt *C1, tpt *C2, eqp_t *pi, mm_t *multi) { /* equality.ts:94-116 */
    fe cc, one;
    hash4(C1, C2, &pi->A1, &pi->A2, &cc);
    fe_set_u64(&one, 1);
    tpt nA1, nA2;
    t_neg(&nA1, &pi->A1); t_neg(&nA2, &pi->A2);
    rel_t r1 = {0}, r2 = {0};
    rel_ins(&r1, &pp->g, &pi->tx); rel_ins(&r1, &pp->h, &pi->tr1); rel_ins(&r1, C1, &cc); rel_ins(&r1, &nA1, &one);
    rel_ins(&r2, &pp->g, &pi->tx); rel_ins(&r2, &pp->h, &pi->tr2); rel_ins(&r2, C2, &cc); rel_ins(&r2, &nA2, &one);
    rel_drain(&r1, multi); rel_drain(&r2, multi);
}
static void aggregate_mult(const tparams *pp, tpt *Cx, tpt *Cy, tpt *Cz, multp_t *pi, mm_t *multi) { /* mult.ts:148-175 */
    fe cc, one;
    hp_t h;
    hp_init(&h);
    hp_t_(&h, Cx); hp_t_(&h, Cy); hp_t_(&h, Cz); hp_t_(&h, &pi->C4); hp_t_(&h, &pi->Ax); hp_t_(&h, &pi->Ay); hp_t_(&h, &pi->Az); hp_t_(&h, &pi->A41); hp_t_(&h, &pi->A42);
    hp_final(&h, &cc);
    fe_set_u64(&one, 1);
    tpt nAx, nAy, nAz, nA41, nA42;
    t_neg(&nAx, &pi->Ax); t_neg(&nAy, &pi->Ay); t_neg(&nAz, &pi->Az); t_neg(&nA41, &pi->A41); t_neg(&nA42, &pi->A42);
    rel_t a = {0}, b = {0}, c = {0}, d = {0}, e = {0};
    rel_ins(&a, &pp->g, &pi->tx); rel_ins(&a, &pp->h, &pi->trx); rel_ins(&a, Cx, &cc); rel_ins(&a, &nAx, &one);
    rel_ins(&b, &pp->g, &pi->ty); rel_ins(&b, &pp->h, &pi->try_); rel_ins(&b, Cy, &cc); rel_ins(&b, &nAy, &one);
    rel_ins(&c, &pp->g, &pi->tz); rel_ins(&c, &pp->h, &pi->trz); rel_ins(&c, Cz, &cc); rel_ins(&c, &nAz, &one);
    rel_ins(&d, &pp->g, &pi->tz); rel_ins(&d, &pp->h, &pi->tr4); rel_ins(&d, &pi->C4, &cc); rel_ins(&d, &nA41, &one);
    rel_ins(&e, Cy, &pi->tx); rel_ins(&e, &pi->C4, &cc); rel_ins(&e, &nA42, &one);
    rel_drain(&a, multi); rel_drain(&b, multi); rel_drain(&c, multi); rel_drain(&d, multi); rel_drain(&e, multi);
}
static void aggregate_point_add(const tparams *pp, tpt *PX, tpt *PY, tpt *QX, tpt *QY, tpt *RX, tpt *RY, paddp_t *pi, mm_t *multi) { /* pointAdd.ts:199-259 */
    tpt *C1 = PX, *C2 = QX, *C3 = RX, *C4 = PY, *C5 = QY, *C6 = RY, C7, C9, C12, C14 = pp->g, Cint;
    t_sub(&C7, C2, C1); t_sub(&C9, C5, C4); t_sub(&C12, C1, C3);
    aggregate_mult(pp, &C7, &pi->C8, &C14, &pi->pi8, multi);
    aggregate_mult(pp, &pi->C8, &C9, &pi->C10, &pi->pi10, multi);
    aggregate_mult(pp, &pi->C10, &pi->C10, &pi->C11, &pi->pi11, multi);
    t_add(&Cint, C3, C1); t_add(&Cint, &Cint, C2);
    aggregate_equality(pp, &pi->C11, &Cint, &pi->pix, multi);
    aggregate_mult(pp, &pi->C10, &C12, &pi->C13, &pi->pi13, multi);
    t_add(&Cint, C4, C6);
    aggregate_equality(pp, &pi->C13, &Cint, &pi->piy, multi);
}

/* gk.ts:197-262; the ring loop is the reference's N*n product form */
static int verify_membership(const zko_ctx *c, tpt *com, int n, tpt *cl, tpt *ca, tpt *cb, tpt *cd, fe *f, fe *za, fe *zb, fe *zd, rng_t *vr) {
    if (n != c->n) return 0; /* gk.ts:208-218 */
    const tparams *pp = &c->tom;
    mm_t multi;
    mm_init(&multi, 1, vr);
    hp_t h;
    hp_init(&h);
    for (int i = 0; i < n; i++) hp_t_(&h, &cl[i]);
    for (int i = 0; i < n; i++) hp_t_(&h, &ca[i]);
    for (int i = 0; i < n; i++) hp_t_(&h, &cb[i]);
    for (int i = 0; i < n; i++) hp_t_(&h, &cd[i]);
    fe x, one, t, u;
    hp_final(&h, &x);
    fe_set_u64(&one, 1);
    mm_add_known(&multi, &pp->g); mm_add_known(&multi, &pp->h);
    for (int i = 0; i < n; i++) {
        rel_t r0 = {0}, r1 = {0};
        fe nf, nza, nzb, xmf;
        mod_neg(&FQ, &nf, &f[i]); mod_neg(&FQ, &nza, &za[i]); mod_neg(&FQ, &nzb, &zb[i]); mod_sub(&FQ, &xmf, &x, &f[i]);
        rel_ins(&r0, &cl[i], &x); rel_ins(&r0, &ca[i], &one); rel_ins(&r0, &pp->g, &nf); rel_ins(&r0, &pp->h, &nza);
        rel_drain(&r0, &multi);
        rel_ins(&r1, &cl[i], &xmf); rel_ins(&r1, &cb[i], &one); rel_ins(&r1, &pp->h, &nzb);
        rel_drain(&r1, &multi);
    }
    fe fm[64], gm[64], total;
    for (int j = 0; j < n; j++) {
        to_mont(&FQ, &fm[j], &f[j]);
        mod_sub(&FQ, &t, &x, &f[j]);
        to_mont(&FQ, &gm[j], &t);
    }
    fe_set_u64(&total, 0);
    for (u64 i = 0; i < c->N; i++) { /* gk.ts:239-250 */
        fe pix = c->ring[i]; /* plain * mont-factors stays plain */
        for (int j = 0; j < n; j++) mont_mul(&FQ, &pix, &pix, (i >> j) & 1 ? &fm[j] : &gm[j]);
        mod_add(&FQ, &total, &total, &pix);
    }
    rel_t rf = {0};
    fe xs[65], negs[64], ntot, nzd;
    tpt comc = *com;
    /* relFinal has n+3 terms: drain it in pieces sharing one randomiser */
    fe rz;
    rnd_mod(vr, &FQ, &rz);
    for (int i = 0; i <= n; i++) fe_pow_small(&xs[i], &x, i);
    for (int i = 0; i < n; i++) {
        mod_neg(&FQ, &negs[i], &xs[i]);
        mod_mul(&FQ, &t, &negs[i], &rz);
        mm_insert(&multi, &cd[i], &t);
    }
    mod_mul(&FQ, &t, &xs[n], &rz); mm_insert(&multi, &comc, &t);
    mod_neg(&FQ, &ntot, &total); mod_mul(&FQ, &t, &ntot, &rz); mm_insert(&multi, &pp->g, &t);
    mod_neg(&FQ, &nzd, zd); mod_mul(&FQ, &u, &nzd, &rz); mm_insert(&multi, &pp->h, &u);
    (void)rf;
    int ok = mm_evaluate_is_identity(&multi);
    mm_free(&multi);
    return ok;
}

static int verify_one(const zko_ctx *c, const uint8_t *msg_hash, const uint8_t *proof, u64 plen, rng_t *vr, uint8_t *ok_out) {
    *ok_out = 0;
    rd_t rd = {proof, plen, 0, 0};
    const uint8_t *hdr = rd_take(&rd, 32);
    if (rd.err || memcmp(hdr, "ZKA1", 4)) return ZK_E_BAD_ENCODING;
    u64 total = (u64)hdr[4] << 24 | (u64)hdr[5] << 16 | (u64)hdr[6] << 8 | hdr[7];
    int sec = hdr[11], n = hdr[15];
    if (total != plen || hdr[8] || hdr[9] || hdr[10] || hdr[12] || hdr[13] || hdr[14] || n > 63 || sec > MAXSEC) return ZK_E_BAD_ENCODING;
    fe bits;
    fe_from_be(&bits, hdr + 16, 16);
    for (int i = sec; i < 128; i++) /* ZKA1 (include/zkattest.h): the challenge-bit field is zero above secLevel */
        if ((bits.v[i / 64] >> (i % 64)) & 1) return ZK_E_BAD_ENCODING;
    ppt R, comS1, G, Q;
    tpt kx, ky;
    rd_pp(&rd, &R); rd_pp(&rd, &comS1); rd_tp(&rd, &kx); rd_tp(&rd, &ky);
    if (rd.err) return rd.err;
    /* parse reps */
    static __thread ppt A[MAXSEC];
    static __thread tpt Tx[MAXSEC], Ty[MAXSEC];
    static __thread fe s0[MAXSEC], s1_[MAXSEC], s2[MAXSEC], s3[MAXSEC];
    static __thread paddp_t *padd[MAXSEC];
    static __thread paddp_t paddbuf[MAXSEC];
    for (int i = 0; i < sec; i++) {
        int bit = (int)((bits.v[i / 64] >> (i % 64)) & 1);
        rd_pp(&rd, &A[i]); rd_tp(&rd, &Tx[i]); rd_tp(&rd, &Ty[i]);
        rd_sc(&rd, &FN, &s0[i]); rd_sc(&rd, &FN, &s1_[i]); rd_sc(&rd, &FQ, &s2[i]); rd_sc(&rd, &FQ, &s3[i]);
        padd[i] = NULL;
        if (!bit) {
            paddp_t *pa = &paddbuf[i];
            rd_tp(&rd, &pa->C8); rd_tp(&rd, &pa->C10); rd_tp(&rd, &pa->C11); rd_tp(&rd, &pa->C13);
            rd_mult(&rd, &pa->pi8); rd_mult(&rd, &pa->pi10); rd_mult(&rd, &pa->pi11); rd_mult(&rd, &pa->pi13);
            rd_eq(&rd, &pa->pix); rd_eq(&rd, &pa->piy);
            padd[i] = pa;
        }
        if (rd.err) return rd.err;
    }
    tpt cl[64], ca[64], cb[64], cd[64];
    fe f[64], za[64], zb[64], zd;
    for (int i = 0; i < n; i++) rd_tp(&rd, &cl[i]);
    for (int i = 0; i < n; i++) rd_tp(&rd, &ca[i]);
    for (int i = 0; i < n; i++) rd_tp(&rd, &cb[i]);
    for (int i = 0; i < n; i++) rd_tp(&rd, &cd[i]);
    for (int i = 0; i < n; i++) rd_sc(&rd, &FQ, &f[i]);
    for (int i = 0; i < n; i++) rd_sc(&rd, &FQ, &za[i]);
    for (int i = 0; i < n; i++) rd_sc(&rd, &FQ, &zb[i]);
    rd_sc(&rd, &FQ, &zd);
    if (rd.err) return rd.err;
    if (rd.off != plen) return ZK_E_BAD_ENCODING;

    /* zkpAttestList.ts:153-164 */
    fe z, zr, rx, rinv, z1;
    fe_from_be(&z, msg_hash, 32);
    mod_reduce(&FN, &zr, &z);
    if (!p_to_affine(&R)) return ZK_E_R_INF;
    from_mont(&FP, &rx, &R.x);
    mod_reduce(&FN, &rx, &rx);
    mod_inv(&FN, &rinv, &rx); mod_mul(&FN, &z1, &rinv, &zr);
    G.x = P256_GX, G.y = P256_GY, G.z = FP.one;
    p_mul(&Q, &G, &z1);
    if (!verify_membership(c, &kx, n, cl, ca, cb, cd, f, za, zb, &zd, vr)) return ZK_OK; /* false */

    /* exp.ts:233-349 with secparam = 20 (zkpAttestList.ts:177) */
    const int secparam = 20;
    if (secparam > sec) return ZK_E_SECLEVEL;
    mm_t multiW, multiN;
    mm_init(&multiW, 1, vr); mm_init(&multiN, 0, vr);
    mm_add_known(&multiW, &c->tom.g); mm_add_known(&multiW, &c->tom.h);
    mm_add_known(&multiN, &R); mm_add_known(&multiN, &c->nist_h); mm_add_known(&multiN, &comS1);
    hp_t h;
    hp_init(&h);
    hp_t_(&h, &kx); hp_t_(&h, &ky);
    for (int i = 0; i < sec; i++) hp_p(&h, &A[i]), hp_t_(&h, &Tx[i]), hp_t_(&h, &Ty[i]);
    fe challenge;
    hp_final(&h, &challenge);
    int indices[MAXSEC];
    for (int i = 0; i < sec; i++) indices[i] = i;
    for (int i = 0; i < sec - 2; i++) { /* exp.ts:95-109 */
        u64 range = (u64)(sec - i), v;
        uint8_t b[32];
        do {
            rng_fill32(vr, b);
            v = b[0];
        } while (v >= range); /* rnd(range): 1-byte rejection sampling */
        int j = i + (int)v, k = indices[i];
        indices[i] = indices[j], indices[j] = k;
    }
    int rc = ZK_OK, result = 1;
    fe one;
    fe_set_u64(&one, 1);
    for (int jj = 0; jj < secparam && result; jj++) {
        int i = indices[jj];
        int bit = (int)((challenge.v[i / 64] >> (i % 64)) & 1);
        ppt nA;
        p_neg(&nA, &A[i]);
        if (bit) {
            if (padd[i]) { rc = ZK_E_PARAMS_NOT_FOUND; break; }
            ppt T;
            p_mul(&T, &R, &s0[i]);
            rel_t ra = {0};
            rel_ins(&ra, &T, &one); rel_ins(&ra, &c->nist_h, &s1_[i]); rel_ins(&ra, &nA, &one);
            rel_drain(&ra, &multiN);
            if (!p_to_affine(&T)) { rc = ZK_E_T_INF; break; }
            fe sx, sy;
            from_mont(&FP, &sx, &T.x); from_mont(&FP, &sy, &T.y);
            tpt nTx, nTy;
            t_neg(&nTx, &Tx[i]); t_neg(&nTy, &Ty[i]);
            rel_t rx_ = {0}, ry_ = {0};
            rel_ins(&rx_, &c->tom.g, &sx); rel_ins(&rx_, &c->tom.h, &s2[i]); rel_ins(&rx_, &nTx, &one);
            rel_ins(&ry_, &c->tom.g, &sy); rel_ins(&ry_, &c->tom.h, &s3[i]); rel_ins(&ry_, &nTy, &one);
            rel_drain(&rx_, &multiW); rel_drain(&ry_, &multiW);
        } else {
            if (!padd[i]) { rc = ZK_E_PARAMS_NOT_FOUND; break; }
            ppt T1;
            p_mul(&T1, &R, &s0[i]);
            rel_t ra = {0};
            rel_ins(&ra, &T1, &one); rel_ins(&ra, &comS1, &one); rel_ins(&ra, &nA, &one); rel_ins(&ra, &c->nist_h, &s1_[i]);
            rel_drain(&ra, &multiN);
            p_add(&T1, &T1, &Q);
            if (!p_to_affine(&T1)) { rc = ZK_E_T1_INF; break; }
            fe sx, sy;
            from_mont(&FP, &sx, &T1.x); from_mont(&FP, &sy, &T1.y);
            tpt T1x, T1y;
            t_dblmul(&T1x, &c->tom.g, &sx, &c->tom.h, &s2[i]);
            t_dblmul(&T1y, &c->tom.g, &sy, &c->tom.h, &s3[i]);
            aggregate_point_add(&c->tom, &T1x, &T1y, &kx, &ky, &Tx[i], &Ty[i], padd[i], &multiW);
        }
    }
    if (rc == ZK_OK && result) result = mm_evaluate_is_identity(&multiW) && mm_evaluate_is_identity(&multiN);
    mm_free(&multiW); mm_free(&multiN);
    if (rc) return rc;
    *ok_out = (uint8_t)result;
    return ZK_OK;
}

typedef struct {
    const zko_ctx *c;
    u64 lo, hi;
    const uint8_t *msg, *proofs, *vseeds;
    const u64 *off;
    uint8_t *ok;
    int32_t *status;
} vjob_t;
static void *verify_worker(void *arg) {
    vjob_t *j = (vjob_t *)arg;
    for (u64 b = j->lo; b < j->hi; b++) {
        uint8_t seed[32];
        if (j->vseeds) memcpy(seed, j->vseeds + 32 * b, 32);
        else {
            uint8_t tag[16] = "zko-verifier";
            memcpy(tag + 12, &b, 4);
            zko_sha256(tag, 16, seed);
        }
        rng_t vr = {0, seed, 0, 0, 0}; /* verifier-RNG contract: fill k = SHA-256(seed || be64(k)) */
        j->status[b] = verify_one(j->c, j->msg + 32 * b, j->proofs + j->off[b], j->off[b + 1] - j->off[b], &vr, &j->ok[b]);
    }
    return NULL;
}
int zko_verify_batch(const zko_ctx *c, u64 B, const uint8_t *msg, const uint8_t *proofs, const u64 *off, const uint8_t *vseeds, uint8_t *ok, int32_t *status, int nthreads) {
    if (!c->ring || !c->sec) return ZK_E_BUFFER;
    if (nthreads < 1) nthreads = 1;
    if ((u64)nthreads > B) nthreads = (int)(B ? B : 1);
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256];
    vjob_t jobs[256];
    for (int t = 0; t < nthreads; t++) {
        vjob_t j = {c, B * t / nthreads, B * (t + 1) / nthreads, msg, proofs, vseeds, off, ok, status};
        jobs[t] = j;
        if (nthreads == 1) verify_worker(&jobs[t]);
        else pthread_create(&th[t], NULL, verify_worker, &jobs[t]);
    }
    if (nthreads > 1)
        for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    return ZK_OK;
}

/* ------------------------------------------------------------------ unit-test hooks (ctypes) */
/* k*G or k*P on P-256: in/out affine big-endian; returns 0 if the result is the identity */
int zko_p256_mul(const uint8_t k_be[32], const uint8_t *p_xy /* NULL = generator */, uint8_t out_xy[64]) {
    ensure_init();
    ppt P, r;
    fe k;
    fe_from_be(&k, k_be, 32);
    if (p_xy) {
        if (!p_from_xy(&P, p_xy)) return -1;
    } else P.x = P256_GX, P.y = P256_GY, P.z = FP.one;
    p_mul(&r, &P, &k);
    if (!p_to_affine(&r)) return 0;
    p_affine_bytes(&r, out_xy);
    return 1;
}
/* v*g + r*h on Tom-256 with the context's parameters (gk.ts:88-92 argument order; same point as pedersen.ts:56) */
int zko_tom_commit(const zko_ctx *c, const uint8_t v_be[32], const uint8_t r_be[32], uint8_t out_xy[72]) {
    fe v, r;
    tpt o;
    fe_from_be(&v, v_be, 32); fe_from_be(&r, r_be, 32);
    mod_reduce(&FQ, &v, &v); mod_reduce(&FQ, &r, &r);
    t_dblmul(&o, &c->tom.g, &v, &c->tom.h, &r);
    t_affine_bytes(&o, out_xy);
    return 1;
}
/* k*P on Tom-256; p_xy NULL = generator */
int zko_tom_mul(const uint8_t k_be[32], const uint8_t *p_xy, uint8_t out_xy[72]) {
    ensure_init();
    tpt P, r;
    fe k;
    fe_from_be(&k, k_be, 32);
    if (p_xy) {
        if (!t_from_xy(&P, p_xy)) return -1;
    } else {
        P.x = TOM_GX, P.y = TOM_GY, P.z = FT.one;
        TM(&P.t, &P.x, &P.y);
    }
    t_mul(&r, &P, &k);
    t_affine_bytes(&r, out_xy);
    return 1;
}
/* field ops for kernel unit tests: which = 0 (F_q = p256.p), 1 (Z_n), 2 (F_t); op = 0 mul, 1 add, 2 sub, 3 inv; 40-byte BE operands */
int zko_field_op(int which, int op, const uint8_t a_be[40], const uint8_t b_be[40], uint8_t out_be[40]) {
    ensure_init();
    const mctx *c = which == 0 ? &FP : which == 1 ? &FN : &FT;
    fe a, b, r;
    fe_from_be(&a, a_be, 40); fe_from_be(&b, b_be, 40);
    if (fe_cmp(&a, &c->m) >= 0 || fe_cmp(&b, &c->m) >= 0) return -1;
    switch (op) {
    case 0: mod_mul(c, &r, &a, &b); break;
    case 1: mod_add(c, &r, &a, &b); break;
    case 2: mod_sub(c, &r, &a, &b); break;
    default: mod_inv(c, &r, &a); break;
    }
    fe_to_be(&r, out_be, 40);
    return 0;
}
