// ZKA1 <-> JSON wire format of SignatureProofList (host-only code, no kernels).
//
// Reference: writeJson/readJson (src/serde.ts:21-36) run typedjson 1.8.0 over the decorated classes.  The decorators
// pin the member names and order and the scalar/coordinate encoding:
//   SignatureProofList {R, comS1, keyXcom, keyYcom, expProof[], membershipProof}      src/zkpAttestList.ts:27-34
//   ExpProof {A, Tx, Ty, [alpha, beta1, beta2, beta3] | [z, z2, proof, r1, r2]}         src/exp/exp.ts:26-40 (optional members)
//   PointAddProof {C_8, C_10, C_11, C_13, pi_8, pi_10, pi_11, pi_13, pi_x, pi_y}        src/exp/pointAdd.ts:28-38
//   MultProof {C_4, A_x, A_y, A_z, A_4_1, A_4_2, t_x, t_y, t_z, t_rx, t_ry, t_rz, t_r4} src/commit/mult.ts:26-40
//   EqualityProof {A_1, A_2, t_x, t_r1, t_r2}                                           src/commit/equality.ts:27-33
//   GKProof {cl[], ca[], cb[], cd[], f[], za[], zb[], zd}                               src/proofGK/gk.ts:31-40
//   point  {group: {name}, x, y}  (toAffine before serialisation)   src/curves/weier.ts:92-101, edwards.ts:89-98, group.ts:21
//   scalar {group: {name}, k}     (reduce before serialisation)     src/curves/group.ts:155-161
//   bigint "0x" + lowercase hex without leading zeros               src/bignum/big.ts:230-239
// typedjson itself is not in /root/reference (package.json:64-66), so the byte-level JSON shape is UNPINNED
// (SURVEY.md section 8c): the emitter follows typedjson's documented behaviour -- members in declaration order,
// undefined optional members omitted, and a trailing "__type" hint on values whose runtime class differs from the
// declared one (Group -> WeierstrassGroup / TEdwards, Group.Point -> WeierstrassPoint / TEdwardsPoint).  The parser is
// deliberately tolerant: member order is free, "__type" and unknown members are ignored, so real typedjson output
// parses whatever the hint policy of the installed version is.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/zkattest.h"

namespace {
const char* G_P = "{\"name\":\"p256\",\"__type\":\"WeierstrassGroup\"}";
const char* G_T = "{\"name\":\"tomEdwards256\",\"__type\":\"TEdwards\"}";

void hex_of(const uint8_t* be, int n, std::string& o) {   // n <= 36
    static const char* d = "0123456789abcdef";
    char tmp[2 * 36 + 4];
    char* q = tmp;
    *q++ = '"', *q++ = '0', *q++ = 'x';
    int i = 0;
    while (i < n && be[i] == 0) i++;
    if (i == n) *q++ = '0';
    else {
        if (be[i] >> 4) *q++ = d[be[i] >> 4];
        *q++ = d[be[i] & 15];
        for (i++; i < n; i++) *q++ = d[be[i] >> 4], *q++ = d[be[i] & 15];
    }
    *q++ = '"';
    o.append(tmp, (size_t)(q - tmp));
}
struct Rd {
    const uint8_t* p;
    uint64_t len, off;
    bool ok;
    const uint8_t* take(uint64_t n) {
        static const uint8_t z[128] = {0};
        if (off + n > len) {
            ok = false;
            return z;
        }
        const uint8_t* r = p + off;
        off += n;
        return r;
    }
};
void pt_p(Rd& r, std::string& o) {
    const uint8_t* b = r.take(64);
    o += "{\"group\":", o += G_P, o += ",\"x\":", hex_of(b, 32, o), o += ",\"y\":", hex_of(b + 32, 32, o), o += ",\"__type\":\"WeierstrassPoint\"}";
}
void pt_t(Rd& r, std::string& o) {
    const uint8_t* b = r.take(72);
    o += "{\"group\":", o += G_T, o += ",\"x\":", hex_of(b, 36, o), o += ",\"y\":", hex_of(b + 36, 36, o), o += ",\"__type\":\"TEdwardsPoint\"}";
}
void sc(Rd& r, bool tom, std::string& o) {
    const uint8_t* b = r.take(32);
    o += "{\"group\":", o += tom ? G_T : G_P, o += ",\"k\":", hex_of(b, 32, o), o += "}";
}
void key(std::string& o, const char* k, bool first = false) {
    if (!first) o += ',';
    o += '"', o += k, o += "\":";
}
void mult(Rd& r, std::string& o) {
    static const char* P[6] = {"C_4", "A_x", "A_y", "A_z", "A_4_1", "A_4_2"};
    static const char* S[7] = {"t_x", "t_y", "t_z", "t_rx", "t_ry", "t_rz", "t_r4"};
    o += '{';
    for (int i = 0; i < 6; i++) key(o, P[i], i == 0), pt_t(r, o);
    for (int i = 0; i < 7; i++) key(o, S[i]), sc(r, true, o);
    o += '}';
}
void eq(Rd& r, std::string& o) {
    o += '{';
    key(o, "A_1", true), pt_t(r, o), key(o, "A_2"), pt_t(r, o);
    key(o, "t_x"), sc(r, true, o), key(o, "t_r1"), sc(r, true, o), key(o, "t_r2"), sc(r, true, o);
    o += '}';
}

// ---------------------------------------------------------------- tolerant JSON reader
// One pass over the text into a flat node array (no allocation per value: a proof at secLevel 80 is ~36 000 values in 596 KB of text);
// strings and keys are spans of the input, only strings with escapes -- none in honest output -- are decoded into a side list.
struct Node {
    enum { NUL, STR, OBJ, ARR, OTHER };
    uint8_t t = NUL;
    uint32_t ks = 0, kl = 0;      // key span (members of an object), or an index into Doc::dec if kesc
    uint32_t s = 0, l = 0;        // STR: value span, or an index into Doc::dec if sesc
    bool kesc = false, sesc = false;
    uint32_t child = 0, next = 0, n = 0;   // first child / next sibling (0 = none; node 0 is the root), number of children
};
struct Doc {
    const char* base = nullptr;
    std::vector<Node> nd;
    std::vector<std::string> dec;
    // key and string accessors
    bool key_is(const Node& m, const char* k, size_t kl) const {
        if (m.kesc) return dec[m.ks] == std::string(k, kl);
        return m.kl == kl && memcmp(base + m.ks, k, kl) == 0;
    }
    void str(const Node& v, const char*& p, size_t& l) const {
        if (v.sesc) p = dec[v.s].data(), l = dec[v.s].size();
        else p = base + v.s, l = v.l;
    }
    const Node* get(const Node* v, const char* k) const {   // first member named k, like the map lookup of a JSON parser
        if (!v || v->t != Node::OBJ) return nullptr;
        size_t kl = strlen(k);
        for (uint32_t c = v->child; c; c = nd[c].next)
            if (key_is(nd[c], k, kl)) return &nd[c];
        return nullptr;
    }
};
struct Parser {
    const char* p;
    const char* e;
    Doc& d;
    bool ok = true;
    void ws() {
        while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++;
    }
    // a string: span [s, s + l) of the input, or (escapes present) decoded like the earlier reader did: the character after a
    // backslash is taken literally
    bool str(uint32_t& s, uint32_t& l, bool& esc) {
        if (p >= e || *p != '"') return ok = false;
        p++;
        const char* q = (const char*)memchr(p, '"', (size_t)(e - p));   // the common case: no escape before the closing quote
        if (!q) return ok = false;
        if (const char* bs = (const char*)memchr(p, '\\', (size_t)(q - p))) q = bs;
        if (*q == '"') {
            s = (uint32_t)(p - d.base), l = (uint32_t)(q - p), esc = false;
            p = q + 1;
            return true;
        }
        std::string out(p, q);
        p = q;
        while (p < e && *p != '"') {
            if (*p == '\\') {
                if (++p >= e) return ok = false;
            }
            out += *p++;
        }
        if (p >= e) return ok = false;
        p++;
        s = (uint32_t)d.dec.size(), l = 0, esc = true;
        d.dec.push_back(std::move(out));
        return true;
    }
    // parses one value into node `at` (already allocated)
    bool val(uint32_t at, int depth = 0) {
        if (depth > 16) return ok = false;
        ws();
        if (p >= e) return ok = false;
        if (*p == '"') {
            d.nd[at].t = Node::STR;
            uint32_t s, l;
            bool esc;
            if (!str(s, l, esc)) return false;
            d.nd[at].s = s, d.nd[at].l = l, d.nd[at].sesc = esc;
            return true;
        }
        if (*p == '{' || *p == '[') {
            const bool obj = *p == '{';
            d.nd[at].t = obj ? Node::OBJ : Node::ARR;
            p++, ws();
            if (p < e && *p == (obj ? '}' : ']')) return p++, true;
            uint32_t last = 0, cnt = 0;
            for (;;) {
                uint32_t c = (uint32_t)d.nd.size();
                d.nd.emplace_back();
                if (obj) {
                    ws();
                    uint32_t s, l;
                    bool esc;
                    if (!str(s, l, esc)) return false;
                    d.nd[c].ks = s, d.nd[c].kl = l, d.nd[c].kesc = esc;
                    ws();
                    if (p >= e || *p != ':') return ok = false;
                    p++;
                }
                if (last) d.nd[last].next = c;
                else d.nd[at].child = c;
                last = c, cnt++;
                if (!val(c, depth + 1)) return false;
                ws();
                if (p < e && *p == ',') {
                    p++;
                    continue;
                }
                if (p < e && *p == (obj ? '}' : ']')) {
                    d.nd[at].n = cnt;
                    return p++, true;
                }
                return ok = false;
            }
        }
        d.nd[at].t = Node::OTHER;  // numbers, true/false/null: skipped
        while (p < e && *p != ',' && *p != '}' && *p != ']') p++;
        return true;
    }
};
struct Wr {
    const Doc& d;
    std::vector<uint8_t> b;
    bool ok = true;
    explicit Wr(const Doc& doc) : d(doc) {}
    // "0x.." -> nbytes big-endian (serdeBigInt.deserializer, big.ts:240-248; negative values are not valid here)
    void hex(const Node* v, int nbytes) {
        size_t at = b.size();
        b.resize(at + nbytes, 0);
        const char* s = nullptr;
        size_t sl = 0;
        if (v && v->t == Node::STR) d.str(*v, s, sl);
        if (!s || sl < 3 || s[0] != '0' || (s[1] != 'x' && s[1] != 'X')) {
            ok = false;
            return;
        }
        size_t nd = sl - 2, lead = 2;
        while (nd > 1 && s[lead] == '0') lead++, nd--;  // BigInt('0x000a') is valid
        if (nd > (size_t)2 * nbytes) {
            ok = false;
            return;
        }
        // two digits per byte from the least significant end, through a nibble table (0xff = not a hex digit)
        static const struct Lut {
            uint8_t v[256];
            Lut() {
                memset(v, 0xff, sizeof v);
                for (int c = '0'; c <= '9'; c++) v[c] = (uint8_t)(c - '0');
                for (int c = 'a'; c <= 'f'; c++) v[c] = (uint8_t)(c - 'a' + 10), v[c - 32] = (uint8_t)(c - 'a' + 10);
            }
        } lut;
        const uint8_t* q = (const uint8_t*)s + sl;   // one past the last digit
        uint8_t* o = b.data() + at + nbytes;           // one past the last byte
        uint8_t bad = 0;
        size_t i = nd;
        for (; i >= 2; i -= 2) {
            uint8_t lo = lut.v[q[-1]], hi = lut.v[q[-2]];
            q -= 2;
            bad |= lo | hi;
            *--o = (uint8_t)(hi << 4 | (lo & 15));
        }
        if (i) {
            uint8_t lo = lut.v[q[-1]];
            bad |= lo;
            *--o = (uint8_t)(lo & 15);
        }
        if (bad & 0x80) ok = false;   // 0xff entries only
    }
    bool group_is(const Node* v, const char* name) {
        const Node* n = d.get(d.get(v, "group"), "name");
        if (!n || n->t != Node::STR) return false;   // instances.ts:58-78: unknown group names are rejected
        const char* s;
        size_t sl;
        d.str(*n, s, sl);
        return sl == strlen(name) && memcmp(s, name, sl) == 0;
    }
    void pt(const Node* v, bool tom) {
        if (!v || v->t != Node::OBJ || !group_is(v, tom ? "tomEdwards256" : "p256")) ok = false;
        hex(d.get(v, "x"), tom ? 36 : 32);
        hex(d.get(v, "y"), tom ? 36 : 32);
    }
    void sc(const Node* v, bool tom) {
        if (!v || v->t != Node::OBJ || !group_is(v, tom ? "tomEdwards256" : "p256")) ok = false;
        hex(d.get(v, "k"), 32);
    }
    void mult(const Node* v) {
        static const char* P[6] = {"C_4", "A_x", "A_y", "A_z", "A_4_1", "A_4_2"};
        static const char* S[7] = {"t_x", "t_y", "t_z", "t_rx", "t_ry", "t_rz", "t_r4"};
        if (!v || v->t != Node::OBJ) ok = false;
        for (auto k : P) pt(d.get(v, k), true);
        for (auto k : S) sc(d.get(v, k), true);
    }
    void eq(const Node* v) {
        if (!v || v->t != Node::OBJ) ok = false;
        pt(d.get(v, "A_1"), true), pt(d.get(v, "A_2"), true);
        sc(d.get(v, "t_x"), true), sc(d.get(v, "t_r1"), true), sc(d.get(v, "t_r2"), true);
    }
};
}  // namespace

// No C++ exception may cross the C ABI: an allocation failure on a hostile input is reported as ZK_E_BUFFER.
#define ZK_JSON_MAX_TEXT ((uint64_t)64 << 20)   // a SignatureProofList at secLevel 128, n = 64 is below 4 MB of JSON
static zk_status proof_to_json_impl(const uint8_t* proof, uint64_t len, char* out, uint64_t cap, uint64_t* out_len) {
    if (!proof || !out_len || len < 32 || memcmp(proof, "ZKA1", 4)) return ZK_E_BAD_ENCODING;
    uint32_t total = (uint32_t)proof[4] << 24 | proof[5] << 16 | proof[6] << 8 | proof[7];
    uint32_t sec = (uint32_t)proof[8] << 24 | proof[9] << 16 | proof[10] << 8 | proof[11];
    uint32_t n = (uint32_t)proof[12] << 24 | proof[13] << 16 | proof[14] << 8 | proof[15];
    if (total != len || sec > 128 || n > 64) return ZK_E_BAD_ENCODING;
    Rd r{proof, len, 32, true};
    std::string o;
    o.reserve(800000);
    o += '{';
    key(o, "R", true), pt_p(r, o), key(o, "comS1"), pt_p(r, o), key(o, "keyXcom"), pt_t(r, o), key(o, "keyYcom"), pt_t(r, o);
    key(o, "expProof"), o += '[';
    for (uint32_t i = 0; i < sec; i++) {
        int bi = 16 + 15 - (int)(i >> 3);
        bool bit = (proof[bi] >> (i & 7)) & 1;
        if (i) o += ',';
        o += '{';
        key(o, "A", true), pt_p(r, o), key(o, "Tx"), pt_t(r, o), key(o, "Ty"), pt_t(r, o);
        if (bit) {
            key(o, "alpha"), sc(r, false, o), key(o, "beta1"), sc(r, false, o), key(o, "beta2"), sc(r, true, o), key(o, "beta3"), sc(r, true, o);
        } else {
            std::string z, z2, r1, r2;
            sc(r, false, z), sc(r, false, z2), sc(r, true, r1), sc(r, true, r2);
            key(o, "z"), o += z, key(o, "z2"), o += z2;
            key(o, "proof"), o += '{';
            static const char* C[4] = {"C_8", "C_10", "C_11", "C_13"};
            for (int k = 0; k < 4; k++) key(o, C[k], k == 0), pt_t(r, o);
            static const char* M[4] = {"pi_8", "pi_10", "pi_11", "pi_13"};
            for (int k = 0; k < 4; k++) key(o, M[k]), mult(r, o);
            key(o, "pi_x"), eq(r, o), key(o, "pi_y"), eq(r, o);
            o += '}';
            key(o, "r1"), o += r1, key(o, "r2"), o += r2;
        }
        o += '}';
    }
    o += ']';
    key(o, "membershipProof"), o += '{';
    static const char* PA[4] = {"cl", "ca", "cb", "cd"};
    for (int k = 0; k < 4; k++) {
        key(o, PA[k], k == 0), o += '[';
        for (uint32_t i = 0; i < n; i++) {
            if (i) o += ',';
            pt_t(r, o);
        }
        o += ']';
    }
    static const char* SA[3] = {"f", "za", "zb"};
    for (int k = 0; k < 3; k++) {
        key(o, SA[k]), o += '[';
        for (uint32_t i = 0; i < n; i++) {
            if (i) o += ',';
            sc(r, true, o);
        }
        o += ']';
    }
    key(o, "zd"), sc(r, true, o);
    o += "}}";
    if (!r.ok || r.off != len) return ZK_E_BAD_ENCODING;
    *out_len = o.size();
    if (!out || cap < o.size()) return ZK_E_BUFFER;
    memcpy(out, o.data(), o.size());
    return ZK_OK;
}

extern "C" zk_status zk_proof_to_json(const uint8_t* proof, uint64_t len, char* out, uint64_t cap, uint64_t* out_len) {
    try {
        return proof_to_json_impl(proof, len, out, cap, out_len);
    } catch (...) {
        return ZK_E_BUFFER;
    }
}

static zk_status proof_from_json_impl(const char* json, uint64_t len, uint8_t* out, uint64_t cap, uint64_t* out_len) {
    if (!json || !out_len) return ZK_E_ARG;
    if (len > ZK_JSON_MAX_TEXT) return ZK_E_BAD_ENCODING;
    Doc d;
    d.base = json;
    d.nd.reserve((size_t)(len / 14) + 16);
    d.nd.emplace_back();
    Parser ps{json, json + len, d};
    if (!ps.val(0) || d.nd[0].t != Node::OBJ) return ZK_E_BAD_ENCODING;
    ps.ws();
    if (ps.p != ps.e) return ZK_E_BAD_ENCODING;
    const Node* root = &d.nd[0];
    const Node* ex = d.get(root, "expProof");
    const Node* gk = d.get(root, "membershipProof");
    if (!ex || ex->t != Node::ARR || !gk || gk->t != Node::OBJ || ex->n > 128) return ZK_E_BAD_ENCODING;
    Wr w(d);
    w.b.reserve((size_t)(len / 3) + 64);
    w.b.resize(32, 0);
    w.pt(d.get(root, "R"), false), w.pt(d.get(root, "comS1"), false), w.pt(d.get(root, "keyXcom"), true), w.pt(d.get(root, "keyYcom"), true);
    uint32_t sec = ex->n;
    uint8_t bits[16] = {0};
    uint32_t i = 0;
    for (uint32_t c = ex->child; c; c = d.nd[c].next, i++) {
        const Node* e = &d.nd[c];
        if (e->t != Node::OBJ) return ZK_E_BAD_ENCODING;
        w.pt(d.get(e, "A"), false), w.pt(d.get(e, "Tx"), true), w.pt(d.get(e, "Ty"), true);
        const Node* alpha = d.get(e, "alpha");
        if (alpha) {  // response1 (exp.ts:30-34)
            bits[15 - (i >> 3)] |= (uint8_t)(1u << (i & 7));
            w.sc(alpha, false), w.sc(d.get(e, "beta1"), false), w.sc(d.get(e, "beta2"), true), w.sc(d.get(e, "beta3"), true);
        } else {      // response0 (exp.ts:35-40)
            w.sc(d.get(e, "z"), false), w.sc(d.get(e, "z2"), false), w.sc(d.get(e, "r1"), true), w.sc(d.get(e, "r2"), true);
            const Node* pa = d.get(e, "proof");
            if (!pa || pa->t != Node::OBJ) return ZK_E_BAD_ENCODING;
            for (auto k : {"C_8", "C_10", "C_11", "C_13"}) w.pt(d.get(pa, k), true);
            for (auto k : {"pi_8", "pi_10", "pi_11", "pi_13"}) w.mult(d.get(pa, k));
            w.eq(d.get(pa, "pi_x")), w.eq(d.get(pa, "pi_y"));
        }
    }
    const Node* cl = d.get(gk, "cl");
    if (!cl || cl->t != Node::ARR || cl->n > 64) return ZK_E_BAD_ENCODING;
    uint32_t n = cl->n;
    for (auto k : {"cl", "ca", "cb", "cd"}) {
        const Node* a = d.get(gk, k);
        if (!a || a->t != Node::ARR || a->n != n) return ZK_E_BAD_ENCODING;
        for (uint32_t c = a->child; c; c = d.nd[c].next) w.pt(&d.nd[c], true);
    }
    for (auto k : {"f", "za", "zb"}) {
        const Node* a = d.get(gk, k);
        if (!a || a->t != Node::ARR || a->n != n) return ZK_E_BAD_ENCODING;
        for (uint32_t c = a->child; c; c = d.nd[c].next) w.sc(&d.nd[c], true);
    }
    w.sc(d.get(gk, "zd"), true);
    if (!w.ok) return ZK_E_BAD_ENCODING;
    uint32_t total = (uint32_t)w.b.size();
    memcpy(w.b.data(), "ZKA1", 4);
    uint32_t hv[3] = {total, sec, n};
    for (int k = 0; k < 3; k++)
        for (int j = 0; j < 4; j++) w.b[4 + 4 * k + j] = (uint8_t)(hv[k] >> (24 - 8 * j));
    memcpy(w.b.data() + 16, bits, 16);
    *out_len = total;
    if (!out || cap < total) return ZK_E_BUFFER;
    memcpy(out, w.b.data(), total);
    return ZK_OK;
}
extern "C" zk_status zk_proof_from_json(const char* json, uint64_t len, uint8_t* out, uint64_t cap, uint64_t* out_len) {
    try {
        return proof_from_json_impl(json, len, out, cap, out_len);
    } catch (...) {
        return ZK_E_BUFFER;
    }
}
