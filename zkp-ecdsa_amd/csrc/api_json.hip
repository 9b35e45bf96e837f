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
// (SURVEY.md section 8d): the emitter follows typedjson's documented behaviour -- members in declaration order,
// undefined optional members omitted, and a trailing "__type" hint on values whose runtime class differs from the
// declared one (Group -> WeierstrassGroup / TEdwards, Group.Point -> WeierstrassPoint / TEdwardsPoint).  The parser is
// deliberately tolerant: member order is free, "__type" and unknown members are ignored, so real typedjson output
// parses whatever the hint policy of the installed version is.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/zkattest.h"
#include "wire.h"

namespace {
// ---------------------------------------------------------------- writer
// The text is produced by ONE traversal of the ZKA1 bytes, instantiated for two sinks: MaxSink adds up an upper bound of the
// text length (literals exactly, every hex field at its full width), BufSink writes into a buffer of at least that size without
// any further check.  Literals carry their length (no strlen per append: a proof is ~10 000 points and scalars), hex digits come
// out 16 bytes per step through a byte shuffle where the CPU has one.
#define LIT(s) s, sizeof(s) - 1
#define G_P_TXT "{\"name\":\"p256\",\"__type\":\"WeierstrassGroup\"}"
#define G_T_TXT "{\"name\":\"tomEdwards256\",\"__type\":\"TEdwards\"}"
struct MaxSink {
    uint64_t n = 0;
    void lit(const char*, size_t l) { n += l; }
    void hex(const uint8_t*, int nbytes) { n += 4 + 2 * (size_t)nbytes; }
};
struct LenSink {   // the exact length: hex fields without their leading zeros
    uint64_t n = 0;
    void lit(const char*, size_t l) { n += l; }
    void hex(const uint8_t* be, int nbytes) {
        int i = 0;
        while (i < nbytes && be[i] == 0) i++;
        n += 4 + (i == nbytes ? 1 : 2 * (size_t)(nbytes - i) - ((be[i] >> 4) ? 0 : 1));
    }
};
#if defined(__x86_64__)
__attribute__((target("ssse3"))) static inline void hex16_ssse3(const uint8_t* src, char* dst) {   // 16 bytes -> 32 lowercase digits
    typedef char v16 __attribute__((vector_size(16)));
    typedef unsigned char u16v __attribute__((vector_size(16)));
    u16v in;
    memcpy(&in, src, 16);
    const u16v lut = {'0', '1', '2', '3', '4', '5', '6', '7', '8', '9', 'a', 'b', 'c', 'd', 'e', 'f'};
    u16v hi = (in >> 4) & (unsigned char)15, lo = in & (unsigned char)15;
    u16v dh = (u16v)__builtin_ia32_pshufb128((v16)lut, (v16)hi), dl = (u16v)__builtin_ia32_pshufb128((v16)lut, (v16)lo);
    u16v a = __builtin_shufflevector(dh, dl, 0, 16, 1, 17, 2, 18, 3, 19, 4, 20, 5, 21, 6, 22, 7, 23);
    u16v b = __builtin_shufflevector(dh, dl, 8, 24, 9, 25, 10, 26, 11, 27, 12, 28, 13, 29, 14, 30, 15, 31);
    memcpy(dst, &a, 16), memcpy(dst + 16, &b, 16);
}
static const bool g_have_ssse3 = __builtin_cpu_supports("ssse3");
#endif
struct BufSink {
    char* q;
    void lit(const char* s, size_t l) {
        memcpy(q, s, l);
        q += l;
    }
    void hex(const uint8_t* be, int n) {   // "0x" + lowercase hex without leading zeros (big.ts:230-239); n <= 36
        static const char* d = "0123456789abcdef";
        *q++ = '"', *q++ = '0', *q++ = 'x';
        int i = 0;
        while (i < n && be[i] == 0) i++;
        if (i == n) *q++ = '0';
        else {
            if (be[i] >> 4) *q++ = d[be[i] >> 4];
            *q++ = d[be[i] & 15];
            i++;
#if defined(__x86_64__)
            if (g_have_ssse3)
                for (; i + 16 <= n; i += 16) hex16_ssse3(be + i, q), q += 32;
#endif
            for (; i < n; i++) *q++ = d[be[i] >> 4], *q++ = d[be[i] & 15];
        }
        *q++ = '"';
    }
};
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
template <class S>
void pt_p(Rd& r, S& o) {
    const uint8_t* b = r.take(64);
    o.lit(LIT("{\"group\":" G_P_TXT ",\"x\":")), o.hex(b, 32), o.lit(LIT(",\"y\":")), o.hex(b + 32, 32), o.lit(LIT(",\"__type\":\"WeierstrassPoint\"}"));
}
template <class S>
void pt_t(Rd& r, S& o) {
    const uint8_t* b = r.take(72);
    o.lit(LIT("{\"group\":" G_T_TXT ",\"x\":")), o.hex(b, 36), o.lit(LIT(",\"y\":")), o.hex(b + 36, 36), o.lit(LIT(",\"__type\":\"TEdwardsPoint\"}"));
}
template <class S>
void sc(Rd& r, bool tom, S& o) {
    const uint8_t* b = r.take(32);
    if (tom) o.lit(LIT("{\"group\":" G_T_TXT ",\"k\":"));
    else o.lit(LIT("{\"group\":" G_P_TXT ",\"k\":"));
    o.hex(b, 32), o.lit(LIT("}"));
}
template <class S>
void key(S& o, const char* k, bool first = false) {
    if (!first) o.lit(LIT(","));
    o.lit(LIT("\"")), o.lit(k, strlen(k)), o.lit(LIT("\":"));
}
template <class S>
void mult(Rd& r, S& o) {
    static const char* P[6] = {"C_4", "A_x", "A_y", "A_z", "A_4_1", "A_4_2"};
    static const char* T[7] = {"t_x", "t_y", "t_z", "t_rx", "t_ry", "t_rz", "t_r4"};
    o.lit(LIT("{"));
    for (int i = 0; i < 6; i++) key(o, P[i], i == 0), pt_t(r, o);
    for (int i = 0; i < 7; i++) key(o, T[i]), sc(r, true, o);
    o.lit(LIT("}"));
}
template <class S>
void eq(Rd& r, S& o) {
    o.lit(LIT("{"));
    key(o, "A_1", true), pt_t(r, o), key(o, "A_2"), pt_t(r, o);
    key(o, "t_x"), sc(r, true, o), key(o, "t_r1"), sc(r, true, o), key(o, "t_r2"), sc(r, true, o);
    o.lit(LIT("}"));
}
// the whole SignatureProofList; returns false when the bytes are not a well-formed ZKA1 proof
template <class S>
bool proof_text(const uint8_t* proof, uint64_t len, S& o) {
    if (!proof || len < 32 || memcmp(proof, "ZKA1", 4)) return false;
    uint32_t total = (uint32_t)proof[4] << 24 | proof[5] << 16 | proof[6] << 8 | proof[7];
    uint32_t sec = (uint32_t)proof[8] << 24 | proof[9] << 16 | proof[10] << 8 | proof[11];
    uint32_t n = (uint32_t)proof[12] << 24 | proof[13] << 16 | proof[14] << 8 | proof[15];
    if (total != len || sec > 128 || n > 64) return false;
    Rd r{proof, len, 32, true};
    o.lit(LIT("{"));
    key(o, "R", true), pt_p(r, o), key(o, "comS1"), pt_p(r, o), key(o, "keyXcom"), pt_t(r, o), key(o, "keyYcom"), pt_t(r, o);
    key(o, "expProof"), o.lit(LIT("["));
    for (uint32_t i = 0; i < sec; i++) {
        int bi = 16 + 15 - (int)(i >> 3);
        bool bit = (proof[bi] >> (i & 7)) & 1;
        if (i) o.lit(LIT(","));
        o.lit(LIT("{"));
        key(o, "A", true), pt_p(r, o), key(o, "Tx"), pt_t(r, o), key(o, "Ty"), pt_t(r, o);
        if (bit) {
            key(o, "alpha"), sc(r, false, o), key(o, "beta1"), sc(r, false, o), key(o, "beta2"), sc(r, true, o), key(o, "beta3"), sc(r, true, o);
        } else {
            // ZKA1 stores z, z2, r1, r2 ahead of the PointAddProof; the JSON member order is z, z2, proof, r1, r2 (exp.ts:35-40)
            key(o, "z"), sc(r, false, o), key(o, "z2"), sc(r, false, o);
            Rd rr{proof, len, r.off, true};          // r1, r2: emitted after the proof
            r.take(64);
            key(o, "proof"), o.lit(LIT("{"));
            static const char* C[4] = {"C_8", "C_10", "C_11", "C_13"};
            for (int k = 0; k < 4; k++) key(o, C[k], k == 0), pt_t(r, o);
            static const char* M[4] = {"pi_8", "pi_10", "pi_11", "pi_13"};
            for (int k = 0; k < 4; k++) key(o, M[k]), mult(r, o);
            key(o, "pi_x"), eq(r, o), key(o, "pi_y"), eq(r, o);
            o.lit(LIT("}"));
            key(o, "r1"), sc(rr, true, o), key(o, "r2"), sc(rr, true, o);
            if (!rr.ok) r.ok = false;
        }
        o.lit(LIT("}"));
    }
    o.lit(LIT("]"));
    key(o, "membershipProof"), o.lit(LIT("{"));
    static const char* PA[4] = {"cl", "ca", "cb", "cd"};
    for (int k = 0; k < 4; k++) {
        key(o, PA[k], k == 0), o.lit(LIT("["));
        for (uint32_t i = 0; i < n; i++) {
            if (i) o.lit(LIT(","));
            pt_t(r, o);
        }
        o.lit(LIT("]"));
    }
    static const char* SA[3] = {"f", "za", "zb"};
    for (int k = 0; k < 3; k++) {
        key(o, SA[k]), o.lit(LIT("["));
        for (uint32_t i = 0; i < n; i++) {
            if (i) o.lit(LIT(","));
            sc(r, true, o);
        }
        o.lit(LIT("]"));
    }
    key(o, "zd"), sc(r, true, o);
    o.lit(LIT("}}"));
    return r.ok && r.off == len;
}
// upper bound of a proof's JSON length from its header alone (every hex field at full width); 0 = malformed
uint64_t proof_text_bound(const uint8_t* proof, uint64_t len) {
    MaxSink m;
    return proof_text(proof, len, m) ? m.n : 0;
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
    void reset(const char* b) { base = b, nd.clear(), dec.clear(); }   // keeps the capacity: one node array per THREAD, not per proof
    // key and string accessors
    bool key_is(const Node& m, const char* k, size_t kl) const {
        if (m.kesc) return dec[m.ks] == std::string(k, kl);
        return m.kl == kl && memcmp(base + m.ks, k, kl) == 0;
    }
    void str(const Node& v, const char*& p, size_t& l) const {
        if (v.sesc) p = dec[v.s].data(), l = dec[v.s].size();
        else p = base + v.s, l = v.l;
    }
    const Node* get(const Node* v, const char* k) const {   // LAST member named k: what JSON.parse keeps of a duplicated key
        if (!v || v->t != Node::OBJ) return nullptr;
        size_t kl = strlen(k);
        const Node* hit = nullptr;
        for (uint32_t c = v->child; c; c = nd[c].next)
            if (key_is(nd[c], k, kl)) hit = &nd[c];
        return hit;
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
    // a string: span [s, s + l) of the input, or (escapes present) decoded as JSON.parse decodes it: the eight short escapes and
    // \uXXXX (UTF-8 out, surrogate pairs joined); anything else after a backslash, and raw control characters, are syntax errors
    static int hex4(const char* q) {
        int v = 0;
        for (int i = 0; i < 4; i++) {
            char ch = q[i];
            int d = ch >= '0' && ch <= '9' ? ch - '0' : ch >= 'a' && ch <= 'f' ? ch - 'a' + 10 : ch >= 'A' && ch <= 'F' ? ch - 'A' + 10 : -1;
            if (d < 0) return -1;
            v = v << 4 | d;
        }
        return v;
    }
    static void utf8(std::string& o, uint32_t cp) {
        if (cp < 0x80) o += (char)cp;
        else if (cp < 0x800) o += (char)(0xc0 | cp >> 6), o += (char)(0x80 | (cp & 63));
        else if (cp < 0x10000) o += (char)(0xe0 | cp >> 12), o += (char)(0x80 | (cp >> 6 & 63)), o += (char)(0x80 | (cp & 63));
        else o += (char)(0xf0 | cp >> 18), o += (char)(0x80 | (cp >> 12 & 63)), o += (char)(0x80 | (cp >> 6 & 63)), o += (char)(0x80 | (cp & 63));
    }
    static bool has_ctl(const char* a, const char* b) {   // any byte below 0x20?  eight at a time
        for (; b - a >= 8; a += 8) {
            uint64_t v;
            memcpy(&v, a, 8);
            // a byte x is < 0x20 iff its top three bits are clear: (x - 0x20) borrows into bit 7 while x's own bit 7 is clear
            if ((v - 0x2020202020202020ull) & ~v & 0x8080808080808080ull) return true;
        }
        for (; a < b; a++)
            if ((unsigned char)*a < 0x20) return true;
        return false;
    }
    bool str(uint32_t& s, uint32_t& l, bool& esc) {
        if (p >= e || *p != '"') return ok = false;
        p++;
        const char* q = (const char*)memchr(p, '"', (size_t)(e - p));   // the common case: no escape before the closing quote
        if (!q) return ok = false;
        if (const char* bs = (const char*)memchr(p, '\\', (size_t)(q - p))) q = bs;
        if (has_ctl(p, q)) return ok = false;
        if (*q == '"') {
            s = (uint32_t)(p - d.base), l = (uint32_t)(q - p), esc = false;
            p = q + 1;
            return true;
        }
        std::string out(p, q);
        p = q;
        while (p < e && *p != '"') {
            if ((unsigned char)*p < 0x20) return ok = false;
            if (*p != '\\') {
                out += *p++;
                continue;
            }
            if (++p >= e) return ok = false;
            switch (*p++) {
            case '"': out += '"'; break;
            case '\\': out += '\\'; break;
            case '/': out += '/'; break;
            case 'b': out += '\b'; break;
            case 'f': out += '\f'; break;
            case 'n': out += '\n'; break;
            case 'r': out += '\r'; break;
            case 't': out += '\t'; break;
            case 'u': {
                if (e - p < 4) return ok = false;
                int hi = hex4(p);
                if (hi < 0) return ok = false;
                p += 4;
                uint32_t cp = (uint32_t)hi;
                if (hi >= 0xd800 && hi < 0xdc00 && e - p >= 6 && p[0] == '\\' && p[1] == 'u') {   // a surrogate pair
                    int lo = hex4(p + 2);
                    if (lo >= 0xdc00 && lo < 0xe000) cp = 0x10000 + (((uint32_t)hi - 0xd800) << 10) + ((uint32_t)lo - 0xdc00), p += 6;
                }
                utf8(out, cp);   // a lone surrogate comes out as its own 3-byte form (never equal to an ASCII name or digit)
                break;
            }
            default: return ok = false;
            }
        }
        if (p >= e) return ok = false;
        p++;
        s = (uint32_t)d.dec.size(), l = 0, esc = true;
        d.dec.push_back(std::move(out));
        return true;
    }
    // true / false / null / a number of the JSON grammar: -?(0|[1-9][0-9]*)(\.[0-9]+)?([eE][+-]?[0-9]+)?
    bool scalar_token() {
        auto word = [&](const char* w, size_t n) { return (size_t)(e - p) >= n && memcmp(p, w, n) == 0 ? (p += n, true) : false; };
        if (word("true", 4) || word("false", 5) || word("null", 4)) return true;
        const char* q = p;
        auto digits = [&] {
            const char* a = q;
            while (q < e && *q >= '0' && *q <= '9') q++;
            return q > a;
        };
        if (q < e && *q == '-') q++;
        if (q < e && *q == '0') q++;
        else if (!digits()) return false;
        if (q < e && *q == '.') {
            q++;
            if (!digits()) return false;
        }
        if (q < e && (*q == 'e' || *q == 'E')) {
            q++;
            if (q < e && (*q == '+' || *q == '-')) q++;
            if (!digits()) return false;
        }
        p = q;
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
        d.nd[at].t = Node::OTHER;  // numbers, true / false / null: valid JSON, but never a member this format reads
        if (!scalar_token()) return ok = false;
        return true;
    }
};
struct Wr {
    const Doc& d;
    std::vector<uint8_t>& b;
    bool ok = true;
    Wr(const Doc& doc, std::vector<uint8_t>& buf) : d(doc), b(buf) {}
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

// ---------------------------------------------------------------- fast path of the reader
// Text in exactly the writer's shape (member order of the decorators, "__type" hints, no white space: what writeJson produces here
// and, per the decorators, in the reference) is matched literal by literal and decoded field by field, without building the node
// tree -- one pass at hex-decoding speed.  ANY deviation (another member order, white space, missing hints, escapes, duplicated
// members ...) abandons the attempt and the tolerant reader above decides; the fast path never rejects a text by itself.
struct Cur {
    const char* p;
    const char* e;
    bool lit(const char* s, size_t l) {
        if ((size_t)(e - p) < l || memcmp(p, s, l)) return false;
        p += l;
        return true;
    }
    bool peek(const char* s, size_t l) const { return (size_t)(e - p) >= l && memcmp(p, s, l) == 0; }
    // "0x<1 .. 2*nbytes hex digits>" -> nbytes big-endian
    bool hex(uint8_t* dst, int nbytes) {
        if (e - p < 5 || p[0] != '"' || p[1] != '0' || p[2] != 'x') return false;
        const char* q = p + 3;
        const char* end = (const char*)memchr(q, '"', (size_t)(e - q) < 80 ? (size_t)(e - q) : 80);
        if (!end || end == q || end - q > 2 * nbytes) return false;
        static const struct Lut {
            uint8_t v[256];
            Lut() {
                memset(v, 0xff, sizeof v);
                for (int c = '0'; c <= '9'; c++) v[c] = (uint8_t)(c - '0');
                for (int c = 'a'; c <= 'f'; c++) v[c] = (uint8_t)(c - 'a' + 10), v[c - 32] = (uint8_t)(c - 'a' + 10);
            }
        } lut;
        memset(dst, 0, (size_t)nbytes);
        const uint8_t* r = (const uint8_t*)end;
        uint8_t* o = dst + nbytes;
        uint8_t bad = 0;
        size_t nd = (size_t)(end - q);
        for (; nd >= 2; nd -= 2) {
            uint8_t lo = lut.v[r[-1]], hi = lut.v[r[-2]];
            r -= 2, bad |= lo | hi;
            *--o = (uint8_t)(hi << 4 | (lo & 15));
        }
        if (nd) {
            uint8_t lo = lut.v[r[-1]];
            bad |= lo;
            *--o = (uint8_t)(lo & 15);
        }
        if (bad & 0x80) return false;
        p = end + 1;
        return true;
    }
};
struct FastRd {
    Cur c;
    std::vector<uint8_t>& b;
    uint8_t* put(size_t n) {
        size_t at = b.size();
        b.resize(at + n);
        return b.data() + at;
    }
    bool pt_p() {
        uint8_t* d = put(64);
        return c.lit(LIT("{\"group\":" G_P_TXT ",\"x\":")) && c.hex(d, 32) && c.lit(LIT(",\"y\":")) && c.hex(d + 32, 32) && c.lit(LIT(",\"__type\":\"WeierstrassPoint\"}"));
    }
    bool pt_t() {
        uint8_t* d = put(72);
        return c.lit(LIT("{\"group\":" G_T_TXT ",\"x\":")) && c.hex(d, 36) && c.lit(LIT(",\"y\":")) && c.hex(d + 36, 36) && c.lit(LIT(",\"__type\":\"TEdwardsPoint\"}"));
    }
    bool sc_at(uint8_t* d, bool tom) {
        return (tom ? c.lit(LIT("{\"group\":" G_T_TXT ",\"k\":")) : c.lit(LIT("{\"group\":" G_P_TXT ",\"k\":"))) && c.hex(d, 32) && c.lit(LIT("}"));
    }
    bool sc(bool tom) { return sc_at(put(32), tom); }
    bool key(const char* k, bool first = false) { return (first || c.lit(LIT(","))) && c.lit(LIT("\"")) && c.lit(k, strlen(k)) && c.lit(LIT("\":")); }
    bool mult() {
        static const char* P[6] = {"C_4", "A_x", "A_y", "A_z", "A_4_1", "A_4_2"};
        static const char* T[7] = {"t_x", "t_y", "t_z", "t_rx", "t_ry", "t_rz", "t_r4"};
        if (!c.lit(LIT("{"))) return false;
        for (int i = 0; i < 6; i++)
            if (!key(P[i], i == 0) || !pt_t()) return false;
        for (int i = 0; i < 7; i++)
            if (!key(T[i]) || !sc(true)) return false;
        return c.lit(LIT("}"));
    }
    bool eq() {
        return c.lit(LIT("{")) && key("A_1", true) && pt_t() && key("A_2") && pt_t() && key("t_x") && sc(true) && key("t_r1") && sc(true) && key("t_r2") && sc(true) &&
               c.lit(LIT("}"));
    }
    bool run() {
        b.clear();
        b.resize(32, 0);
        if (!(c.lit(LIT("{")) && key("R", true) && pt_p() && key("comS1") && pt_p() && key("keyXcom") && pt_t() && key("keyYcom") && pt_t() && key("expProof") &&
              c.lit(LIT("["))))
            return false;
        uint32_t sec = 0;
        uint8_t bits[16] = {0};
        if (!c.peek(LIT("]"))) {
            for (;;) {
                if (sec >= 128) return false;
                if (!(c.lit(LIT("{")) && key("A", true) && pt_p() && key("Tx") && pt_t() && key("Ty") && pt_t())) return false;
                if (c.peek(LIT(",\"alpha\":"))) {
                    bits[15 - (sec >> 3)] |= (uint8_t)(1u << (sec & 7));
                    if (!(key("alpha") && sc(false) && key("beta1") && sc(false) && key("beta2") && sc(true) && key("beta3") && sc(true))) return false;
                } else {
                    if (!(key("z") && sc(false) && key("z2") && sc(false))) return false;
                    const size_t r12 = b.size();   // ZKA1 keeps r1, r2 ahead of the PointAddProof; the text has them behind it
                    put(64);
                    static const char* C4[4] = {"C_8", "C_10", "C_11", "C_13"};
                    static const char* M4[4] = {"pi_8", "pi_10", "pi_11", "pi_13"};
                    if (!(key("proof") && c.lit(LIT("{")))) return false;
                    for (int k = 0; k < 4; k++)
                        if (!key(C4[k], k == 0) || !pt_t()) return false;
                    for (int k = 0; k < 4; k++)
                        if (!key(M4[k]) || !mult()) return false;
                    if (!(key("pi_x") && eq() && key("pi_y") && eq() && c.lit(LIT("}")))) return false;
                    if (!(key("r1") && sc_at(b.data() + r12, true) && key("r2") && sc_at(b.data() + r12 + 32, true))) return false;
                }
                if (!c.lit(LIT("}"))) return false;
                sec++;
                if (c.lit(LIT(","))) continue;
                break;
            }
        }
        if (!(c.lit(LIT("]")) && key("membershipProof") && c.lit(LIT("{")))) return false;
        static const char* PA[4] = {"cl", "ca", "cb", "cd"};
        static const char* SA[3] = {"f", "za", "zb"};
        uint32_t n = 0;
        for (int k = 0; k < 7; k++) {
            if (!(key(k < 4 ? PA[k] : SA[k - 4], k == 0) && c.lit(LIT("[")))) return false;
            uint32_t cnt = 0;
            if (!c.peek(LIT("]"))) {
                for (;;) {
                    if (cnt >= 64) return false;
                    if (!(k < 4 ? pt_t() : sc(true))) return false;
                    cnt++;
                    if (c.lit(LIT(","))) continue;
                    break;
                }
            }
            if (!c.lit(LIT("]"))) return false;
            if (k == 0) n = cnt;
            else if (cnt != n) return false;
        }
        if (!(key("zd") && sc(true) && c.lit(LIT("}}")) && c.p == c.e)) return false;
        const uint32_t total = (uint32_t)b.size();
        memcpy(b.data(), "ZKA1", 4);
        const uint32_t hv[3] = {total, sec, n};
        for (int k = 0; k < 3; k++)
            for (int j = 0; j < 4; j++) b[4 + 4 * k + j] = (uint8_t)(hv[k] >> (24 - 8 * j));
        memcpy(b.data() + 16, bits, 16);
        return true;
    }
};

// No C++ exception may cross the C ABI: an allocation failure on a hostile input is reported as ZK_E_BUFFER.
#define ZK_JSON_MAX_TEXT ((uint64_t)64 << 20)   // a SignatureProofList at secLevel 128, n = 64 is below 4 MB of JSON
static zk_status proof_to_json_impl(const uint8_t* proof, uint64_t len, char* out, uint64_t cap, uint64_t* out_len) {
    if (!proof || !out_len) return ZK_E_BAD_ENCODING;
    const uint64_t bound = proof_text_bound(proof, len);
    if (!bound) return ZK_E_BAD_ENCODING;
    std::vector<char> tmp;
    char* dst = out;
    if (!out || cap < bound) {   // the exact length is only known after the conversion (hex fields drop their leading zeros)
        tmp.resize(bound);
        dst = tmp.data();
    }
    BufSink b{dst};
    if (!proof_text(proof, len, b)) return ZK_E_BAD_ENCODING;
    const uint64_t n = (uint64_t)(b.q - dst);
    *out_len = n;
    if (!out || cap < n) return ZK_E_BUFFER;
    if (dst != out) memcpy(out, dst, n);
    return ZK_OK;
}

// ---- ZKA1 <-> ZKA1P on the host (one proof): the same fields in the same order, Tom-256 coordinates 36 <-> 33 bytes (include/zkattest.h)
static zk_status convert_wire(const uint8_t* in, uint64_t len, bool to_packed, uint8_t* out, uint64_t cap, uint64_t* out_len) {
    if (!in || !out_len) return ZK_E_ARG;
    const Wire src = wire_make(!to_packed), dst = wire_make(to_packed);
    auto be32 = [](const uint8_t* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; };
    if (len < ZK_HDR || memcmp(in, to_packed ? "ZKA1" : "ZK1P", 4) != 0) return ZK_E_BAD_ENCODING;
    const uint32_t total = be32(in + 4), sec = be32(in + 8), n = be32(in + 12);
    if (total != len || sec > ZK_MAXSEC || n > 63) return ZK_E_BAD_ENCODING;
    uint32_t z = 0;
    auto bit = [&](uint32_t i) { return (in[31 - i / 8] >> (i % 8)) & 1u; };
    for (uint32_t i = 0; i < 128; i++) {
        if (i >= sec && bit(i)) return ZK_E_BAD_ENCODING;
        if (i < sec && !bit(i)) z++;
    }
    if (len != wire_proof_size(src, sec, n, z)) return ZK_E_BAD_ENCODING;
    const uint64_t need = wire_proof_size(dst, sec, n, z);
    *out_len = need;
    if (!out || cap < need) return ZK_E_BUFFER;
    const uint8_t* p = in;
    uint8_t* q = out;
    bool bad = false;
    auto plain = [&](size_t nb) { memcpy(q, p, nb), p += nb, q += nb; };
    auto coords = [&](uint32_t cnt) {
        for (uint32_t k = 0; k < cnt; k++) {
            if (to_packed) {
                bad = bad || p[0] || p[1] || p[2];   // a value of 2^264 or more has no 33-byte form (and is no coordinate)
                memcpy(q, p + 3, 33), p += 36, q += 33;
            } else {
                q[0] = q[1] = q[2] = 0;
                memcpy(q + 3, p, 33), p += 33, q += 36;
            }
        }
    };
    plain(ZK_HDR + 128), coords(4);
    for (uint32_t i = 0; i < sec; i++) {
        plain(64), coords(4), plain(128);
        if (!bit(i)) {
            coords(8);
            for (int m = 0; m < 4; m++) coords(12), plain(224);
            for (int e = 0; e < 2; e++) coords(4), plain(96);
        }
    }
    coords(8 * n), plain(32 * (3 * (size_t)n + 1));
    if (bad || (uint64_t)(p - in) != len || (uint64_t)(q - out) != need) return ZK_E_BAD_ENCODING;
    memcpy(out, to_packed ? "ZK1P" : "ZKA1", 4);
    out[4] = (uint8_t)(need >> 24), out[5] = (uint8_t)(need >> 16), out[6] = (uint8_t)(need >> 8), out[7] = (uint8_t)need;
    return ZK_OK;
}
extern "C" zk_status zk_proof_pack(const uint8_t* zka1, uint64_t len, uint8_t* out, uint64_t cap, uint64_t* out_len) { return convert_wire(zka1, len, true, out, cap, out_len); }
extern "C" zk_status zk_proof_unpack(const uint8_t* zka1p, uint64_t len, uint8_t* out, uint64_t cap, uint64_t* out_len) { return convert_wire(zka1p, len, false, out, cap, out_len); }

extern "C" zk_status zk_proof_to_json(const uint8_t* proof, uint64_t len, char* out, uint64_t cap, uint64_t* out_len) {
    try {
        if (proof && len >= 4 && memcmp(proof, "ZK1P", 4) == 0) {   // the packed layout: expanded first, the text is the same
            static thread_local std::vector<uint8_t> tmp;
            uint64_t need = 0;
            zk_status zs = convert_wire(proof, len, false, nullptr, 0, &need);
            if (zs != ZK_E_BUFFER) return zs ? zs : ZK_E_BAD_ENCODING;
            tmp.resize(need);
            if ((zs = convert_wire(proof, len, false, tmp.data(), need, &need))) return zs;
            return proof_to_json_impl(tmp.data(), need, out, cap, out_len);
        }
        return proof_to_json_impl(proof, len, out, cap, out_len);
    } catch (...) {
        return ZK_E_BUFFER;
    }
}

static zk_status proof_from_json_impl(const char* json, uint64_t len, uint8_t* out, uint64_t cap, uint64_t* out_len) {
    if (!json || !out_len) return ZK_E_ARG;
    if (len > ZK_JSON_MAX_TEXT) return ZK_E_BAD_ENCODING;
    // per-thread scratch, reused from proof to proof: a fresh 1.8 MB node array per proof costs more in page faults (and, across
    // threads, in contention on the address space) than the parse itself
    static thread_local Doc d;
    static thread_local std::vector<uint8_t> wbuf;
    if (!getenv("ZKATTEST_JSON_NO_FAST")) {   // the writer's own shape: one pass, no tree (falls through on any deviation)
        wbuf.reserve((size_t)(len / 3) + 64);
        FastRd f{Cur{json, json + len}, wbuf};
        if (f.run()) {
            *out_len = wbuf.size();
            if (!out || cap < wbuf.size()) return ZK_E_BUFFER;
            memcpy(out, wbuf.data(), wbuf.size());
            return ZK_OK;
        }
    }
    d.reset(json);
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
    Wr w(d, wbuf);
    w.b.clear();
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

// ---------------------------------------------------------------- whole batches, on several host threads
// bench/zkpAttestList.bench.ts:63-68 times toJson / fromJson per proof; a GPU that makes 300 000 proofs per second needs the
// converters at batch scale ("so the ~10 GB/batch host conversion is not the bottleneck").  Proofs are
// converted in blocks: every thread converts whole proofs into its own scratch, a prefix sum places them, the threads copy them out.
#include <sched.h>
#include <atomic>
#include <cstdio>
#include <thread>
// CPUs this process may really use: the affinity mask, capped by the cgroup v2 CPU quota (a container with 16 CPUs' worth of quota on a
// 256-thread host reports hardware_concurrency() = 256; 256 threads on 16 CPUs' worth of time convert more slowly than 16)
static uint32_t usable_cpus() {
    uint32_t n = std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) {
        int c = CPU_COUNT(&set);
        if (c > 0 && (uint32_t)c < n) n = (uint32_t)c;
    }
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64] = {0};
        unsigned long long period = 0;
        if (fscanf(f, "%63s %llu", q, &period) == 2 && strcmp(q, "max") != 0 && period) {
            unsigned long long quota = strtoull(q, nullptr, 10), c = quota / period;
            if (c >= 1 && c < n) n = (uint32_t)c;
        }
        fclose(f);
    }
    return n ? n : 1;
}
static uint32_t json_threads(uint32_t want, uint64_t n) {
    uint32_t hw = usable_cpus();
    if (const char* e = getenv("ZKATTEST_JSON_THREADS")) hw = (uint32_t)atoi(e);
    uint32_t t = want ? want : (hw ? hw : 1);
    if (t > 256) t = 256;
    if (t > n) t = (uint32_t)(n ? n : 1);
    return t;
}
template <class F>
static void run_threads(uint32_t T, F f) {
    if (T <= 1) return f(0);
    // std::thread's constructor throws std::system_error when the process may not start another thread (pid / cgroup limits): a
    // joinable thread destroyed by the unwinding would terminate the process before the entry point's catch (...) could answer, so the
    // threads already running are kept and the parts that got no thread run here
    std::vector<std::thread> th;
    th.reserve(T);
    uint32_t started = 0;
    try {
        for (; started < T; started++) th.emplace_back(f, started);
    } catch (...) {
    }
    for (uint32_t t = started; t < T; t++) f(t);
    for (auto& x : th) x.join();
}
// `conv(i, scratch)` converts item i into scratch (resized by it) and returns its status; items are placed back to back in out.
template <class Conv>
static zk_status batch_convert(uint64_t n, uint8_t* out, uint64_t cap, uint64_t* out_off, int32_t* status, uint32_t threads, Conv conv) {
    const uint32_t T = json_threads(threads, n);
    const uint64_t BLOCK = (uint64_t)T * 16;
    std::vector<std::vector<uint8_t>> tmp(BLOCK);
    uint64_t cursor = 0;
    bool overflow = false;
    out_off[0] = 0;
    for (uint64_t b0 = 0; b0 < n; b0 += BLOCK) {
        const uint64_t b1 = b0 + BLOCK < n ? b0 + BLOCK : n;
        std::atomic<uint64_t> next(b0);
        run_threads(T, [&](uint32_t) {
            for (;;) {
                uint64_t i = next.fetch_add(1);
                if (i >= b1) break;
                auto& v = tmp[i - b0];
                v.clear();
                int32_t st = ZK_E_BUFFER;
                try {
                    st = conv(i, v);
                } catch (...) {
                    st = ZK_E_BUFFER;
                }
                if (st) v.clear();
                status[i] = st;
            }
        });
        for (uint64_t i = b0; i < b1; i++) cursor += tmp[i - b0].size(), out_off[i + 1] = cursor;
        if (cursor > cap || !out) overflow = true;
        if (!overflow) {
            next = b0;
            run_threads(T, [&](uint32_t) {
                for (;;) {
                    uint64_t i = next.fetch_add(1);
                    if (i >= b1) break;
                    const auto& v = tmp[i - b0];
                    if (!v.empty()) memcpy(out + out_off[i], v.data(), v.size());
                }
            });
        }
    }
    return overflow ? ZK_E_BUFFER : ZK_OK;   // out_off[] is complete either way: out_off[n] is the size to come back with
}
// proof_text for either layout: a ZKA1P proof is expanded into a per-thread scratch first (the text does not know the layout)
template <class Sink>
static bool proof_text_any(const uint8_t* proof, uint64_t len, Sink& sink) {
    if (len >= 4 && memcmp(proof, "ZK1P", 4) == 0) {
        static thread_local std::vector<uint8_t> tmp;
        uint64_t need = 0;
        if (convert_wire(proof, len, false, nullptr, 0, &need) != ZK_E_BUFFER) return false;
        tmp.resize(need);
        if (convert_wire(proof, len, false, tmp.data(), need, &need) != ZK_OK) return false;
        return proof_text(tmp.data(), need, sink);
    }
    return proof_text(proof, len, sink);
}
// to JSON: the exact length of every text first (a traversal that only counts), a prefix sum, then every thread writes its proofs'
// texts straight into their final place -- no scratch copies.
extern "C" zk_status zk_proofs_to_json_batch(uint64_t n, const uint8_t* proofs, const uint64_t* proof_off, char* out, uint64_t out_cap, uint64_t* text_off,
                                             int32_t* per_proof_status, uint32_t threads) {
    if (!text_off || !per_proof_status || (n && (!proofs || !proof_off))) return ZK_E_ARG;
    for (uint64_t i = 0; i < n; i++)
        if (proof_off[i + 1] < proof_off[i]) return ZK_E_ARG;
    try {
        const uint32_t T = json_threads(threads, n);
        std::atomic<uint64_t> next(0);
        run_threads(T, [&](uint32_t) {
            for (;;) {
                uint64_t i0 = next.fetch_add(8);
                if (i0 >= n) break;
                for (uint64_t i = i0; i < i0 + 8 && i < n; i++) {
                    LenSink m;
                    const bool good = proof_text_any(proofs + proof_off[i], proof_off[i + 1] - proof_off[i], m);
                    per_proof_status[i] = good ? ZK_OK : ZK_E_BAD_ENCODING;
                    text_off[i + 1] = good ? m.n : 0;   // lengths for now
                }
            }
        });
        text_off[0] = 0;
        for (uint64_t i = 0; i < n; i++) text_off[i + 1] += text_off[i];
        if (!out || text_off[n] > out_cap) return n && text_off[n] ? ZK_E_BUFFER : ZK_OK;
        next = 0;
        run_threads(T, [&](uint32_t) {
            for (;;) {
                uint64_t i0 = next.fetch_add(8);
                if (i0 >= n) break;
                for (uint64_t i = i0; i < i0 + 8 && i < n; i++) {
                    if (per_proof_status[i]) continue;
                    BufSink b{out + text_off[i]};
                    proof_text_any(proofs + proof_off[i], proof_off[i + 1] - proof_off[i], b);
                }
            }
        });
        return ZK_OK;
    } catch (...) {
        return ZK_E_BUFFER;
    }
}
extern "C" zk_status zk_proofs_from_json_batch(uint64_t n, const char* texts, const uint64_t* text_off, uint8_t* out, uint64_t out_cap, uint64_t* proof_off,
                                               int32_t* per_proof_status, uint32_t threads) {
    if (!proof_off || !per_proof_status || (n && (!texts || !text_off))) return ZK_E_ARG;
    for (uint64_t i = 0; i < n; i++)
        if (text_off[i + 1] < text_off[i]) return ZK_E_ARG;
    try {
        return batch_convert(n, out, out_cap, proof_off, per_proof_status, threads, [&](uint64_t i, std::vector<uint8_t>& v) -> int32_t {
            const uint64_t len = text_off[i + 1] - text_off[i];
            v.resize((size_t)(len / 3) + 64);   // every proof byte costs at least two hex digits plus punctuation
            uint64_t got = 0;
            zk_status st = proof_from_json_impl(texts + text_off[i], len, v.data(), v.size(), &got);
            if (st == ZK_E_BUFFER && got > v.size()) {
                v.resize((size_t)got);
                st = proof_from_json_impl(texts + text_off[i], len, v.data(), v.size(), &got);
            }
            if (st) return st;
            v.resize((size_t)got);
            return ZK_OK;
        });
    } catch (...) {
        return ZK_E_BUFFER;
    }
}
