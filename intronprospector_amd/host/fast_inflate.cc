// fast_inflate.cc -- see fast_inflate.hh.
#include "fast_inflate.hh"

#include <cstring>

namespace ipxhost {
namespace {

constexpr int LL_BITS = 11, D_BITS = 8, PRE_BITS = 7, MAX_LEN = 15;
constexpr uint32_t LL_CAP = (1u << LL_BITS) + 288 * 16, D_CAP = (1u << D_BITS) + 32 * 128;

// table entry: bits 0..7 number of bits of the codeword (a pointer: index bits of the second-level table),
// 8..11 extra bits, 12..15 kind, 16..31 literal / base value / offset of the second-level table
enum : uint32_t { K_LITERAL = 0, K_BASE = 1, K_END = 2, K_SUBTABLE = 3, K_INVALID = 15 };
constexpr uint32_t entry(uint32_t kind, uint32_t value, uint32_t extra) { return (value << 16) | (kind << 12) | (extra << 8); }
constexpr uint32_t INVALID = entry(K_INVALID, 0, 0);
inline uint32_t kind_of(uint32_t e) { return (e >> 12) & 0xfu; }

const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073,
                                4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

inline uint32_t litlen_entry(uint32_t sym) {
    if (sym < 256) return entry(K_LITERAL, sym, 0);
    if (sym == 256) return entry(K_END, 0, 0);
    if (sym < 286) return entry(K_BASE, LEN_BASE[sym - 257], LEN_EXTRA[sym - 257]);
    return INVALID;                                     // 286, 287: have codes in the fixed tree, never occur
}
inline uint32_t dist_entry(uint32_t sym) { return sym < 30 ? entry(K_BASE, DIST_BASE[sym], DIST_EXTRA[sym]) : INVALID; }
inline uint32_t pre_entry(uint32_t sym) { return entry(K_LITERAL, sym, 0); }

inline uint32_t reverse_bits(uint32_t code, int len) {
    uint32_t r = 0;
    for (int i = 0; i < len; ++i) { r = (r << 1) | (code & 1u); code >>= 1; }
    return r;
}

// canonical Huffman decoding table from code lengths (RFC 1951 3.2.2); false if the lengths over-subscribe the
// code space.  Incomplete codes are accepted: the unused patterns decode as INVALID.
template <typename EntryFn>
bool build_table(const uint8_t *lens, int n_sym, int primary, uint32_t *table, uint32_t cap, EntryFn sym_entry) {
    int count[MAX_LEN + 1] = {0};
    for (int s = 0; s < n_sym; ++s) count[lens[s]]++;
    count[0] = 0;
    int left = 1;
    for (int l = 1; l <= MAX_LEN; ++l) {
        left <<= 1;
        left -= count[l];
        if (left < 0) return false;
    }
    uint32_t next[MAX_LEN + 2];
    uint32_t code = 0;
    for (int l = 1; l <= MAX_LEN; ++l) {
        code = (code + (uint32_t)count[l - 1]) << 1;
        next[l] = code;
    }
    const uint32_t psize = 1u << primary, pmask = psize - 1;
    for (uint32_t i = 0; i < psize; ++i) table[i] = INVALID;
    // first pass: short codes go straight in; long ones tell their prefix how deep its second level must be
    uint8_t deepest[1u << LL_BITS];
    bool any_long = false;
    uint32_t codes[288];
    for (int s = 0; s < n_sym; ++s) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t rev = reverse_bits(next[l]++, l);
        codes[s] = rev;
        if (l <= primary) {
            const uint32_t e = sym_entry((uint32_t)s) | (uint32_t)l;
            for (uint32_t i = rev; i < psize; i += 1u << l) table[i] = e;
        } else {
            if (!any_long) { memset(deepest, 0, psize); any_long = true; }
            uint8_t &d = deepest[rev & pmask];
            if (l > d) d = (uint8_t)l;
        }
    }
    if (!any_long) return true;
    uint32_t used = psize;
    for (uint32_t p = 0; p < psize; ++p) {
        if (!deepest[p]) continue;
        const uint32_t sub_bits = (uint32_t)deepest[p] - (uint32_t)primary, size = 1u << sub_bits;
        if (used + size > cap) return false;
        for (uint32_t i = 0; i < size; ++i) table[used + i] = INVALID;
        table[p] = entry(K_SUBTABLE, used, 0) | sub_bits;
        used += size;
    }
    for (int s = 0; s < n_sym; ++s) {
        const int l = lens[s];
        if (l <= primary) continue;
        const uint32_t rev = codes[s], ptr = table[rev & pmask];
        const uint32_t base = ptr >> 16, size = 1u << (ptr & 0xffu);
        const uint32_t e = sym_entry((uint32_t)s) | (uint32_t)l;
        for (uint32_t i = rev >> primary; i < size; i += 1u << (l - primary)) table[base + i] = e;
    }
    return true;
}

struct Tables {
    uint32_t ll[LL_CAP];
    uint32_t d[D_CAP];
};

struct FixedTables : Tables {
    FixedTables() {
        uint8_t lens[288];
        for (int i = 0; i < 144; ++i) lens[i] = 8;
        for (int i = 144; i < 256; ++i) lens[i] = 9;
        for (int i = 256; i < 280; ++i) lens[i] = 7;
        for (int i = 280; i < 288; ++i) lens[i] = 8;
        (void)build_table(lens, 288, LL_BITS, ll, LL_CAP, litlen_entry);
        uint8_t dl[32];
        for (int i = 0; i < 32; ++i) dl[i] = 5;
        (void)build_table(dl, 32, D_BITS, d, D_CAP, dist_entry);
    }
};

inline uint64_t load64(const uint8_t *p) {
    uint64_t v;
    memcpy(&v, p, 8);
    return v;                                           // little-endian hosts only (x86-64), like the rest of host/
}

struct Stream {
    uint64_t bitbuf = 0;
    uint32_t bitcnt = 0;                                // bits of bitbuf that are accounted for (more may be set: see refill_fast)
    const uint8_t *ip, *in_end;
    uint8_t *op, *out, *out_end;
    size_t overrun = 0;                                 // zero bytes fed after the input ran out

    // at least n <= 56 bits in the buffer, byte by byte, zeros beyond the input's end
    inline void need(uint32_t n) {
        while (bitcnt < n) {
            if (ip < in_end) bitbuf |= (uint64_t)*ip++ << bitcnt;
            else ++overrun;
            bitcnt += 8;
        }
    }
    // 56..63 bits with one load (ip + 8 <= in_end).  The load may set bits above bitcnt; they are the bits of
    // the bytes that follow, so a later refill ORs the same values onto them
    inline void refill_fast() {
        bitbuf |= load64(ip) << bitcnt;
        ip += (63 - bitcnt) >> 3;
        bitcnt |= 56;
    }
    inline uint32_t peek(uint32_t n) const { return (uint32_t)(bitbuf & ((1ull << n) - 1)); }
    inline void drop(uint32_t n) { bitbuf >>= n; bitcnt -= n; }
    inline uint32_t take(uint32_t n) { const uint32_t v = peek(n); drop(n); return v; }
};

inline uint32_t lookup(const uint32_t *t, uint64_t bits, int primary) {
    uint32_t e = t[bits & ((1u << primary) - 1)];
    if (kind_of(e) == K_SUBTABLE) e = t[(e >> 16) + ((uint32_t)(bits >> primary) & ((1u << (e & 0xffu)) - 1))];
    return e;
}

// symbols of one block.  0: end of block; 1 (FAST only): too close to the end of input or output, go on carefully;
// -1: malformed.  The stream's state lives in locals while the loop runs: the byte stores to the output may
// alias anything, and the compiler would otherwise reload the bit buffer from the Stream after each of them.
template <bool FAST>
int decode_block(Stream &s, const Tables &t) {
    uint64_t bitbuf = s.bitbuf;
    uint32_t bitcnt = s.bitcnt;
    const uint8_t *ip = s.ip;
    uint8_t *op = s.op;
    const uint8_t *const in_end = s.in_end;
    uint8_t *const out = s.out, *const out_end = s.out_end;
    size_t overrun = s.overrun;
    const uint32_t *const ll = t.ll, *const dt = t.d;
    int result;
#define IPX_DROP(n) do { bitbuf >>= (n); bitcnt -= (n); } while (0)
#define IPX_LOOKUP(e, tab, PB)                                                                                       \
    do {                                                                                                             \
        (e) = (tab)[bitbuf & ((1u << (PB)) - 1)];                                                                    \
        if (kind_of(e) == K_SUBTABLE) (e) = (tab)[((e) >> 16) + ((uint32_t)(bitbuf >> (PB)) & ((1u << ((e) & 0xffu)) - 1))]; \
    } while (0)
    for (;;) {
        if (FAST) {
            if (in_end - ip < 16 || out_end - op < 258 + 16) { result = 1; break; }
            bitbuf |= load64(ip) << bitcnt;             // see Stream::refill_fast
            ip += (63 - bitcnt) >> 3;
            bitcnt |= 56;
        } else {
            while (bitcnt < 48) {
                if (ip < in_end) bitbuf |= (uint64_t)*ip++ << bitcnt;
                else ++overrun;
                bitcnt += 8;
            }
        }
        uint32_t e;
        IPX_LOOKUP(e, ll, LL_BITS);
        if (kind_of(e) == K_LITERAL) {
            if (!FAST && op >= out_end) { result = -1; break; }
            IPX_DROP(e & 0xffu);
            *op++ = (uint8_t)(e >> 16);
            if (!FAST) continue;
            // up to two more literals from the bits at hand (a length / distance pair may need 48)
            IPX_LOOKUP(e, ll, LL_BITS);
            if (kind_of(e) != K_LITERAL) continue;
            IPX_DROP(e & 0xffu);
            *op++ = (uint8_t)(e >> 16);
            IPX_LOOKUP(e, ll, LL_BITS);
            if (kind_of(e) != K_LITERAL || bitcnt < 15) continue;
            IPX_DROP(e & 0xffu);
            *op++ = (uint8_t)(e >> 16);
            continue;
        }
        if (kind_of(e) == K_END) {
            IPX_DROP(e & 0xffu);
            result = 0;
            break;
        }
        if (kind_of(e) != K_BASE) { result = -1; break; }
        IPX_DROP(e & 0xffu);
        const uint32_t lx = (e >> 8) & 0xfu;
        const uint32_t len = (e >> 16) + (uint32_t)(bitbuf & ((1ull << lx) - 1));
        IPX_DROP(lx);
        uint32_t d;
        IPX_LOOKUP(d, dt, D_BITS);
        if (kind_of(d) != K_BASE) { result = -1; break; }
        IPX_DROP(d & 0xffu);
        const uint32_t dx = (d >> 8) & 0xfu;
        const uint32_t dist = (d >> 16) + (uint32_t)(bitbuf & ((1ull << dx) - 1));
        IPX_DROP(dx);
        if ((size_t)dist > (size_t)(op - out)) { result = -1; break; }
        const uint8_t *src = op - dist;
        uint8_t *end = op + len;
        if (FAST) {                                    // 16 bytes of room beyond `end`
            if (dist >= 8) {
                uint8_t *o = op;
                do { memcpy(o, src, 8); o += 8; src += 8; } while (o < end);
            } else if (dist == 1) {
                memset(op, *src, len);
            } else {
                uint8_t *o = op;
                do { *o++ = *src++; } while (o < end);
            }
            op = end;
        } else {
            if ((size_t)len > (size_t)(out_end - op)) { result = -1; break; }
            uint8_t *o = op;
            do { *o++ = *src++; } while (o < end);
            op = end;
        }
    }
#undef IPX_DROP
#undef IPX_LOOKUP
    s.bitbuf = bitbuf;
    s.bitcnt = bitcnt;
    s.ip = ip;
    s.op = op;
    s.overrun = overrun;
    return result;
}

}  // namespace

long fast_inflate(const uint8_t *in, size_t in_n, uint8_t *out, size_t out_cap) {
    static const FixedTables fixed;
    Stream s;
    s.ip = in;
    s.in_end = in + in_n;
    s.op = s.out = out;
    s.out_end = out + out_cap;
    Tables dyn;
    for (;;) {
        s.need(3);
        const uint32_t final_block = s.take(1), type = s.take(2);
        const Tables *t = nullptr;
        if (type == 0) {
            // stored: back to the byte boundary, LEN, ~LEN, the bytes
            if (s.overrun) return -1;
            s.drop(s.bitcnt & 7u);
            s.ip -= s.bitcnt >> 3;
            s.bitbuf = 0;
            s.bitcnt = 0;
            if (s.in_end - s.ip < 4) return -1;
            const uint32_t len = s.ip[0] | (s.ip[1] << 8), nlen = s.ip[2] | (s.ip[3] << 8);
            s.ip += 4;
            if ((len ^ nlen) != 0xffffu) return -1;
            if ((size_t)(s.in_end - s.ip) < len || (size_t)(s.out_end - s.op) < len) return -1;
            memcpy(s.op, s.ip, len);
            s.ip += len;
            s.op += len;
        } else if (type == 1) {
            t = &fixed;
        } else if (type == 2) {
            s.need(14);
            const uint32_t hlit = s.take(5) + 257, hdist = s.take(5) + 1, hclen = s.take(4) + 4;
            if (hlit > 286 || hdist > 30) return -1;
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t pre_lens[19] = {0};
            for (uint32_t i = 0; i < hclen; ++i) {
                s.need(3);
                pre_lens[order[i]] = (uint8_t)s.take(3);
            }
            uint32_t pre[1u << PRE_BITS];
            if (!build_table(pre_lens, 19, PRE_BITS, pre, 1u << PRE_BITS, pre_entry)) return -1;
            uint8_t lens[288 + 32];
            uint32_t i = 0;
            const uint32_t total = hlit + hdist;
            while (i < total) {
                s.need(PRE_BITS + 7);
                const uint32_t e = pre[s.peek(PRE_BITS)];
                if (kind_of(e) != K_LITERAL) return -1;
                s.drop(e & 0xffu);
                const uint32_t sym = e >> 16;
                if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
                uint32_t rep, val = 0;
                if (sym == 16) {
                    if (i == 0) return -1;
                    val = lens[i - 1];
                    rep = 3 + s.take(2);
                } else if (sym == 17) {
                    rep = 3 + s.take(3);
                } else {
                    rep = 11 + s.take(7);
                }
                if (i + rep > total) return -1;
                memset(lens + i, (int)val, rep);
                i += rep;
            }
            if (lens[256] == 0) return -1;                       // no end-of-block code
            if (!build_table(lens, (int)hlit, LL_BITS, dyn.ll, LL_CAP, litlen_entry)) return -1;
            if (!build_table(lens + hlit, (int)hdist, D_BITS, dyn.d, D_CAP, dist_entry)) return -1;
            t = &dyn;
        } else {
            return -1;
        }
        if (t) {
            int r = decode_block<true>(s, *t);
            if (r == 1) r = decode_block<false>(s, *t);
            if (r != 0) return -1;
        }
        if (final_block) break;
    }
    if (s.overrun && (size_t)s.bitcnt < 8 * s.overrun) return -1;    // bits were taken from beyond the input's end
    return (long)(s.op - s.out);
}

namespace {
struct CrcTables {
    uint32_t t[16][256];
    CrcTables() {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int k = 1; k < 16; ++k) t[k][i] = (t[k - 1][i] >> 8) ^ t[0][t[k - 1][i] & 0xffu];
    }
};
}  // namespace

uint32_t fast_crc32(uint32_t crc, const uint8_t *p, size_t n) {
    static const CrcTables T;
    uint32_t c = ~crc;
    while (n >= 16) {
        uint32_t w[4];
        memcpy(w, p, 16);
        const uint32_t a = w[0] ^ c, b = w[1], d = w[2], e = w[3];
        c = T.t[0][e >> 24] ^ T.t[1][(e >> 16) & 0xffu] ^ T.t[2][(e >> 8) & 0xffu] ^ T.t[3][e & 0xffu] ^
            T.t[4][d >> 24] ^ T.t[5][(d >> 16) & 0xffu] ^ T.t[6][(d >> 8) & 0xffu] ^ T.t[7][d & 0xffu] ^
            T.t[8][b >> 24] ^ T.t[9][(b >> 16) & 0xffu] ^ T.t[10][(b >> 8) & 0xffu] ^ T.t[11][b & 0xffu] ^
            T.t[12][a >> 24] ^ T.t[13][(a >> 16) & 0xffu] ^ T.t[14][(a >> 8) & 0xffu] ^ T.t[15][a & 0xffu];
        p += 16;
        n -= 16;
    }
    while (n--) c = T.t[0][(c ^ *p++) & 0xffu] ^ (c >> 8);
    return ~c;
}

}  // namespace ipxhost
