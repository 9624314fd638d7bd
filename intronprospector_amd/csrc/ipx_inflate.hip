// ipx_inflate.hip -- BGZF blocks inflated on the device, one lane per block (EXPERIMENT).
//
// A BGZF file is tens of thousands of independent DEFLATE streams of at most 64 KiB each (the reference reads its
// BAM through htslib's bgzf.c, which inflates them one after the other; src/junctions_extractor.cc:620-634 sits on
// top of that).  Inflating is what bounds the command line (6 of its 7 CPU-seconds on a 10 M-record file), and
// the blocks are independent, so every lane of the device can take one: a serial decoder per lane, Huffman tables
// in a private slice of global memory (they stay in L2), no cooperation between lanes.  RFC 1951 throughout:
//   * codes of up to 10 bits (literal/length) or 8 bits (distance) through a direct table, longer ones through
//     the canonical count/offset walk bit by bit (rare symbols).  (Tables of 9 / 7 bits in LDS, one column per lane,
//     were slower in round 2: 328 against 222 ms on a 10 M-record BAM -- too many 10-bit codes fell to the walk.  Round 4
//     tried the full 10 / 8-bit tables in LDS, interleaved over the lanes, a wave per workgroup and all 160 KB of a CU:
//     255 against 132 ms on the final kernel -- with one wave per CU only 256 of the file's 563 waves run at a time.)
//   * A wave's turn takes as long as its slowest lane.  Round 4 (169 -> 132 ms, 12.9 -> 16.5 GB/s of inflated bytes): the
//     walk BEHIND the direct table is arithmetic on counts held in a register and one load (LongCodes) where it was up to
//     fifteen dependent loads; the input word behind the one just taken is asked for ahead of its use (Bits::ahead); a
//     match copies unaligned 8-byte words, its last bytes from ONE more load, a run (distance < 8) from a register that is
//     stored whole and rotated; literals are gathered into aligned 8-byte stores (OutBytes); the blocks are dealt to the
//     lanes longest first (`order`).  What a turn still waits for is the table entry of every symbol -- a line of its own
//     per lane, from a 73 MB working set -- and, most turns, some lane's symbol look-up behind the table.
//   * every loop is bounded by the block's output size or a constant, every index is checked: a damaged stream
//     ends in a status word, not in a fault or a wave that never finishes.
// status[b]: 0 = inflated to exactly isize[b] bytes, otherwise an error code (1 bad block type / stored header,
// 2 bad code lengths, 3 invalid code, 4 distance before the start, 5 output overrun, 6 input overrun, 7 wrong size).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ipx_common.h"
#include <cstdlib>

namespace {

constexpr uint32_t LLB = 10, DB = 8;
constexpr uint32_t LANE_WORDS = 1024;                 // private table space per lane: 4 KiB
// byte offsets inside a lane's slice
constexpr uint32_t O_LL = 0, O_SLL = 2560, O_SD = 3136, O_LENS = 3200, O_CNT = 3520;   // u16 ll[1024], (512 bytes that held the distance table until it moved to LDS), u16 sorted_ll[288], u16 sorted_d[32], u8 lens[320], u16 cnt_ll[16], u16 cnt_d[16]

// (the length and distance codes' bases and extra-bit counts are computed, RFC 1951 3.2.5: see the symbol loop)
__constant__ uint8_t C_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct Bits {
    const uint8_t *base;          // the whole compressed buffer (4-byte aligned)
    uint64_t pos, end;            // next byte to fetch / one past the block's last byte
    uint64_t buf;
    uint32_t cnt;
    uint32_t past;                // zero bytes supplied beyond `end`
    // The word BEHIND the one fill() last took is asked for at once and kept in `ahead` (round 4): the next fill() finds it
    // there instead of waiting a round trip for it -- a symbol's turn was two dependent round trips, the input word and the
    // table entry its bits select.  (The buffers end in spare bytes: a word behind a block's last is readable.)
    uint32_t ahead;
    uint64_t ahead_at;            // the aligned address `ahead` was loaded from (~0: nothing loaded yet)
    // at least 33 bits (zeros beyond the end)
    __device__ __forceinline__ void fill() {
        while (cnt <= 32) {
            if (pos >= end) {                                           // nothing left: four zero bytes, not a load
                cnt += 32;
                past += 4;
                pos += 4;
                continue;
            }
            // the aligned word around pos, cut to the bytes of this block
            const uint64_t a = pos & ~3ull;
            uint32_t w = a == ahead_at ? ahead : *reinterpret_cast<const uint32_t *>(base + a);
            ahead = *reinterpret_cast<const uint32_t *>(base + a + 4);
            ahead_at = a + 4;
            const uint32_t skip = (uint32_t)(pos - a);                  // bytes of the word that lie before pos
            w >>= 8 * skip;
            uint32_t have = 4 - skip;                                   // bytes now in w
            if (pos + have > end) {
                const uint32_t keep = pos < end ? (uint32_t)(end - pos) : 0u;
                w &= keep >= 4 ? 0xffffffffu : ((1u << (8 * keep)) - 1u);
                past += have - keep;
            }
            buf |= (uint64_t)w << cnt;
            cnt += 8 * have;
            pos += have;
        }
    }
    __device__ __forceinline__ uint32_t peek(uint32_t n) const { return (uint32_t)(buf & ((1ull << n) - 1)); }
    __device__ __forceinline__ void drop(uint32_t n) { buf >>= n; cnt -= n; }
    __device__ __forceinline__ uint32_t take(uint32_t n) { const uint32_t v = peek(n); drop(n); return v; }
};

__device__ __forceinline__ uint32_t rev_bits(uint32_t code, uint32_t len) { return __brev(code) >> (32 - len); }

// a lane's direct tables: a slice of global memory (the LDS form that was tried: see the header)
struct TabGlobal {
    uint16_t *p;
    __device__ __forceinline__ uint16_t get(uint32_t i) const { return p[i]; }
    __device__ __forceinline__ void set(uint32_t i, uint16_t v) const { p[i] = v; }
};

// the distance table (and the code-length code's, which is built in its space) in LDS, a column per lane: what a match
// waits for is then one trip to memory less (round 4)
struct TabLds {
    uint16_t *p;                  // the lane's first entry
    uint32_t stride;              // lanes per wave
    __device__ __forceinline__ uint16_t get(uint32_t i) const { return p[i * stride]; }
    __device__ __forceinline__ void set(uint32_t i, uint16_t v) const { p[i * stride] = v; }
};

// what the canonical walk needs of a code to go on BEHIND the direct table (round 4): the state of RFC 1951 3.2.2's count /
// offset walk after `primary` bits -- no code that short matched, the table said so -- and the number of codes of every
// greater length, nine bits each, in one register.  The walk behind the table is then at most five (literal / length) or
// seven (distance) turns of arithmetic and ONE load, where it used to be up to fifteen dependent loads of cnt[] from the
// lane's slice of global memory: and a wave's turn takes as long as its slowest lane -- with a few symbols in a hundred
// longer than ten bits, some lane of the 64 was in that walk nearly every turn (~3 us per symbol).
struct LongCodes {
    unsigned long long counts;      // cnt[primary + 1 + k] at bits 9 k .. 9 k + 8
    uint32_t first, index;          // `first` and `index` of the walk after length `primary`
};

template <class Tab>
__device__ bool build(const uint8_t *lens, uint32_t n_sym, uint32_t primary, const Tab tab, uint16_t *sorted, uint16_t *cnt,
                      LongCodes *lc = nullptr) {
    for (uint32_t l = 0; l < 16; ++l) cnt[l] = 0;
    for (uint32_t s = 0; s < n_sym; ++s) cnt[lens[s]]++;
    cnt[0] = 0;
    int left = 1;
    for (uint32_t l = 1; l < 16; ++l) {
        left <<= 1;
        left -= (int)cnt[l];
        if (left < 0) return false;
    }
    uint32_t offs[16], next[16];
    offs[1] = 0;
    uint32_t code = 0;
    next[0] = 0;
    for (uint32_t l = 1; l < 16; ++l) {
        if (l > 1) offs[l] = offs[l - 1] + cnt[l - 1];
        code = (code + (l > 1 ? cnt[l - 1] : 0u)) << 1;
        next[l] = code;
    }
    if (lc) {
        uint32_t first = 0, index = 0;
        for (uint32_t l = 1; l <= primary; ++l) { index += cnt[l]; first = (first + cnt[l]) << 1; }
        unsigned long long packed = 0;
        for (uint32_t l = primary + 1, k = 0; l < 16; ++l, ++k) packed |= (unsigned long long)cnt[l] << (9u * k);      // (cnt <= 288 < 512; k <= 6 for primary >= 8)
        lc->counts = packed; lc->first = first; lc->index = index;
    }
    const uint32_t size = 1u << primary;
    for (uint32_t i = 0; i < size; ++i) tab.set(i, 0xffffu);
    for (uint32_t s = 0; s < n_sym; ++s) {
        const uint32_t l = lens[s];
        if (!l) continue;
        sorted[offs[l]++] = (uint16_t)s;
        const uint32_t c = next[l]++;
        if (l <= primary) {
            const uint32_t r = rev_bits(c, l);
            const uint16_t e = (uint16_t)(s | (l << 12));
            for (uint32_t i = r; i < size; i += 1u << l) tab.set(i, e);
        }
    }
    return true;
}

// a symbol: through the direct table, or bit by bit (RFC 1951 3.2.2's canonical order); 0xffffffff: no such code
// (lc != nullptr: the walk goes on behind the table, see LongCodes; primary >= 8)
// (decode_entry: the table entry `e` of the next `primary` bits was fetched by the caller -- ahead of time, see the symbol loop)
template <class Tab>
__device__ __forceinline__ uint32_t decode_entry(Bits &b, const uint32_t e, uint32_t primary, const uint16_t *sorted, const uint16_t *cnt,
                                                 const LongCodes *lc);
template <class Tab>
__device__ __forceinline__ uint32_t decode(Bits &b, const Tab tab, uint32_t primary, const uint16_t *sorted, const uint16_t *cnt,
                                           const LongCodes *lc = nullptr) {
    return decode_entry<Tab>(b, tab.get(b.peek(primary)), primary, sorted, cnt, lc);
}
template <class Tab>
__device__ __forceinline__ uint32_t decode_entry(Bits &b, const uint32_t e, uint32_t primary, const uint16_t *sorted, const uint16_t *cnt,
                                                 const LongCodes *lc) {
    if (e != 0xffffu) {
        b.drop(e >> 12);
        return e & 0x1ffu;
    }
    if (lc) {
        uint32_t code = rev_bits(b.peek(primary), primary) << 1, first = lc->first, index = lc->index;
        uint64_t bits = b.buf >> primary;
        unsigned long long cc = lc->counts;
        for (uint32_t l = primary + 1; l < 16; ++l, cc >>= 9) {
            code |= (uint32_t)(bits & 1u);
            bits >>= 1;
            const uint32_t c = (uint32_t)cc & 0x1ffu;
            if (code - first < c) {
                b.drop(l);
                return sorted[index + (code - first)];
            }
            index += c;
            first = (first + c) << 1;
            code <<= 1;
        }
        return 0xffffffffu;
    }
    uint32_t code = 0, first = 0, index = 0;
    uint64_t bits = b.buf;
    for (uint32_t l = 1; l < 16; ++l) {
        code |= (uint32_t)(bits & 1u);
        bits >>= 1;
        const uint32_t c = cnt[l];
        if (code - first < c) {
            b.drop(l);
            return sorted[index + (code - first)];
        }
        index += c;
        first = (first + c) << 1;
        code <<= 1;
    }
    return 0xffffffffu;
}

// The output of a lane, literals gathered into aligned 8-byte stores.  `word` mirrors the aligned 8-byte word the next
// byte falls into, from the first byte of it that is this lane's (w_own: 0 except in the word the block begins or a
// match ended in) up to w_pos; a completed word that is wholly the lane's goes out in one store, anything else byte by
// byte.  flush() makes memory current (a match is about to read it); after_copy() starts afresh behind bytes that a
// match wrote to memory directly.
struct __attribute__((packed, aligned(1))) U64 { unsigned long long v; };      // an 8-byte word at any address

struct OutBytes {
    uint8_t *o;
    uint32_t n;                   // bytes produced so far (in memory or in `word`)
    unsigned long long word;
    uint32_t w_own, w_done, w_pos;         // w_done: bytes of the word already in memory
    __device__ __forceinline__ void start(uint8_t *base) {
        o = base; n = 0; word = 0;
        w_own = w_done = w_pos = (uint32_t)((uintptr_t)base & 7u);
    }
    __device__ __forceinline__ void put(uint32_t byte) {
        word |= (unsigned long long)byte << (8u * w_pos);
        ++w_pos; ++n;
        if (w_pos == 8u) {
            uint8_t *w = o + n - 8u;                                       // the word's first byte (8-byte aligned)
            if (w_own == 0u) *reinterpret_cast<unsigned long long *>(w) = word;
            else for (uint32_t k = w_done; k < 8u; ++k) w[k] = (uint8_t)(word >> (8u * k));
            word = 0; w_own = w_done = w_pos = 0u;
        }
    }
    __device__ __forceinline__ void flush() {
        uint8_t *w = o + n - w_pos;
        for (uint32_t k = w_done; k < w_pos; ++k) w[k] = (uint8_t)(word >> (8u * k));
        w_done = w_pos;
    }
    __device__ __forceinline__ void after_copy(uint32_t len) {           // `len` bytes were written to memory behind a flush()
        n += len;
        word = 0;
        w_own = w_done = w_pos = (uint32_t)((uintptr_t)(o + n) & 7u);
    }
};

// in_end == nullptr: block b's payload ends where block b + 1's begins (in_off has n_blocks + 1 entries)
// order: the block a lane takes (null: its own number)
// (no second launch bound: `__launch_bounds__(64, 3)` -- what the kernel's 162 registers give anyway -- crashes this
// compiler's post-RA scheduler, ROCm 7.2.0 clang 22; the launcher asks the runtime how many waves fit)
__global__ __launch_bounds__(64) void k_bgzf_inflate(const uint8_t *__restrict__ comp, const uint64_t *__restrict__ in_off,
                                                     const uint64_t *__restrict__ in_end,
                                                     const uint32_t *__restrict__ isize, const uint64_t *__restrict__ out_off,
                                                     uint64_t n_blocks, uint8_t *__restrict__ out, uint32_t *__restrict__ tables,
                                                     uint32_t *__restrict__ status, const uint32_t *__restrict__ order,
                                                     uint32_t lanes_per_wave) {
    extern __shared__ uint16_t lds_d[];                                  // [256][lanes_per_wave]
    if (threadIdx.x >= lanes_per_wave) return;
    const uint64_t gid = (uint64_t)blockIdx.x * lanes_per_wave + threadIdx.x;
    if (gid >= n_blocks) return;
    const uint64_t blk = order ? order[gid] : gid;
    uint8_t *slice = reinterpret_cast<uint8_t *>(tables + blk * LANE_WORDS);
    const TabGlobal ll{reinterpret_cast<uint16_t *>(slice + O_LL)};
    const TabLds dt{lds_d + threadIdx.x, lanes_per_wave};
    uint16_t *sll = reinterpret_cast<uint16_t *>(slice + O_SLL), *sd = reinterpret_cast<uint16_t *>(slice + O_SD);
    uint8_t *lens = slice + O_LENS;
    uint16_t *cll = reinterpret_cast<uint16_t *>(slice + O_CNT), *cd = cll + 16;
    Bits b{comp, in_off[blk], in_end ? in_end[blk] : in_off[blk + 1], 0, 0, 0, 0, ~0ull};
    LongCodes lc_ll{0, 0, 0}, lc_d{0, 0, 0};
    OutBytes ob;
    ob.start(out + out_off[blk]);
    uint8_t *const o = ob.o;
    const uint32_t cap = isize[blk];
    uint32_t err = 0;
    // A match of up to 32 bytes whose source lies wholly before its destination is copied in two halves (round 4): its
    // source words are asked for when it is decoded, its bytes are stored when the NEXT match is decoded (or at the end)
    // -- the symbols between do not wait for the copy's trip to memory.  Memory sees the lane's loads and stores in program
    // order, so a later match that reads these bytes finds them: the stores below are issued before its loads.
    uint32_t p_len = 0, p_at = 0;
    unsigned long long p0 = 0, p1 = 0, p2 = 0, p3 = 0;
    auto put_word = [&](uint32_t k, unsigned long long w) {
        const uint32_t rem = p_len - 8u * k;                              // bytes of the match from this word on
        uint8_t *d = o + p_at + 8u * k;
        if (rem >= 8u) reinterpret_cast<U64 *>(d)->v = w;
        else for (uint32_t j = 0; j < rem; ++j) d[j] = (uint8_t)(w >> (8u * j));
    };
    auto retire = [&]() {
        if (p_len) {
            put_word(0, p0);
            if (p_len > 8u) put_word(1, p1);
            if (p_len > 16u) put_word(2, p2);
            if (p_len > 24u) put_word(3, p3);
            p_len = 0;
        }
    };
    for (uint32_t guard = 0; guard < 70000 && !err; ++guard) {            // a block of n bytes has at most n / 1 DEFLATE blocks worth looking at
        b.fill();
        const uint32_t final_block = b.take(1), type = b.take(2);
        if (type == 0) {
            b.drop(b.cnt & 7u);
            b.fill();
            const uint32_t len = b.take(16);
            b.fill();
            const uint32_t nlen = b.take(16);
            if ((len ^ nlen) != 0xffffu) { err = 1; break; }
            if (ob.n + len > cap) { err = 5; break; }
            for (uint32_t k = 0; k < len; ++k) {
                b.fill();
                ob.put(b.take(8));
            }
        } else if (type == 1 || type == 2) {
            if (type == 1) {
                for (uint32_t i = 0; i < 144; ++i) lens[i] = 8;
                for (uint32_t i = 144; i < 256; ++i) lens[i] = 9;
                for (uint32_t i = 256; i < 280; ++i) lens[i] = 7;
                for (uint32_t i = 280; i < 288; ++i) lens[i] = 8;
                (void)build(lens, 288, LLB, ll, sll, cll, &lc_ll);
                for (uint32_t i = 0; i < 32; ++i) lens[i] = 5;
                (void)build(lens, 32, DB, dt, sd, cd, &lc_d);
            } else {
                b.fill();
                const uint32_t hlit = b.take(5) + 257, hdist = b.take(5) + 1, hclen = b.take(4) + 4;
                if (hlit > 286 || hdist > 30) { err = 2; break; }
                uint8_t *pre = lens + 300;                               // 19 code-length code lengths, behind the others
                for (uint32_t i = 0; i < 19; ++i) pre[i] = 0;
                for (uint32_t i = 0; i < hclen; ++i) {
                    b.fill();
                    pre[C_ORDER[i]] = (uint8_t)b.take(3);
                }
                // the code-length code: 7 bits at most, through the distance table's space
                if (!build(pre, 19, 7, dt, sd, cd)) { err = 2; break; }
                const uint32_t total = hlit + hdist;
                uint32_t i = 0;
                while (i < total && !err) {
                    b.fill();
                    const uint32_t sym = decode(b, dt, 7, sd, cd);
                    if (sym > 18) { err = 3; break; }
                    if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
                    uint32_t rep, val = 0;
                    if (sym == 16) {
                        if (i == 0) { err = 2; break; }
                        val = lens[i - 1];
                        rep = 3 + b.take(2);
                    } else if (sym == 17) {
                        rep = 3 + b.take(3);
                    } else {
                        rep = 11 + b.take(7);
                    }
                    if (i + rep > total) { err = 2; break; }
                    for (uint32_t k = 0; k < rep; ++k) lens[i++] = (uint8_t)val;
                }
                if (err) break;
                if (lens[256] == 0) { err = 2; break; }
                // the distance lengths follow the literal/length ones in lens[]: build that table first, it moves nothing
                if (!build(lens + hlit, hdist, DB, dt, sd, cd, &lc_d)) { err = 2; break; }
                if (!build(lens, hlit, LLB, ll, sll, cll, &lc_ll)) { err = 2; break; }
            }
            // symbols.  A lane's turn is a chain of dependent round trips to memory, and the wave's turn is its slowest lane's
            // -- in practice a match's, some lane of 64 has one every turn: table entry of the length code, base and extra
            // bits of that length (tables), entry of the distance code, its base and extra bits, the bytes to copy: five
            // trips, ~5 us.  Round 4, second half: lengths and distances come from their codes by arithmetic (RFC 1951
            // 3.2.5's tables are geometric: no load), and the entry of the NEXT symbol is asked for as soon as the bits of
            // this one are known to be spent -- before the copy, whose source bytes then travel at the same time: a match
            // is the distance entry and the copy's bytes behind one another, a literal one entry.
            bool eob = false;
            b.fill();
            uint32_t e_next = ll.get(b.peek(LLB));
            for (uint32_t sguard = 0; sguard < 70000; ++sguard) {        // every symbol but the last writes a byte
                const uint32_t sym = decode_entry<TabGlobal>(b, e_next, LLB, sll, cll, &lc_ll);
                if (sym < 256) {
                    if (ob.n >= cap) { err = 5; break; }
                    b.fill();
                    e_next = ll.get(b.peek(LLB));
                    ob.put(sym);
                    continue;
                }
                if (sym == 256) { eob = true; break; }
                if (sym > 285) { err = 3; break; }
                // 257..264: 3..10; then four codes per extra bit count; 285: 258
                const uint32_t li = sym - 257u;
                const uint32_t lx = (li < 8u || li == 28u) ? 0u : (li >> 2) - 1u;
                const uint32_t lbase = li < 8u ? 3u + li : (li == 28u ? 258u : 3u + ((4u + (li & 3u)) << lx));
                const uint32_t len = lbase + b.take(lx);
                b.fill();
                const uint32_t ds = decode(b, dt, DB, sd, cd, &lc_d);
                if (ds > 29) { err = 3; break; }
                // 0..3: 1..4; then two codes per extra bit count
                const uint32_t dx = ds < 4u ? 0u : (ds >> 1) - 1u;
                const uint32_t dbase = ds < 4u ? 1u + ds : 1u + ((2u + (ds & 1u)) << dx);
                const uint32_t dist = dbase + b.take(dx);
                if (dist > ob.n) { err = 4; break; }
                if (ob.n + len > cap) { err = 5; break; }
                b.fill();
                e_next = ll.get(b.peek(LLB));                            // on its way while the match is copied
                retire();                                                // the match before this one: its bytes go to memory now
                ob.flush();                                              // the match reads what was produced so far from memory
                if (len <= 32u && dist >= len) {
                    // (all four words: what lies behind the source's last byte is not used -- and readable, the output
                    // buffer ends in spare bytes)
                    const U64 *src = reinterpret_cast<const U64 *>(o + ob.n - dist);
                    p0 = src[0].v; p1 = src[1].v; p2 = src[2].v; p3 = src[3].v;
                    p_len = len; p_at = ob.n;
                    ob.after_copy(len);
                    continue;
                }
                // 8-byte words (unaligned: the hardware takes them) where a word's source ends before its destination begins;
                // 32 bytes a turn, all four loads before the four stores, where that holds for the whole piece.  (Byte by
                // byte every load waits for the store before it: ~1 us per byte; and the wave waits for its slowest lane.)
                uint32_t k = 0, at = ob.n;
                if (dist >= 32) {
                    for (; k + 32 <= len; k += 32, at += 32) {
                        const U64 *src = reinterpret_cast<const U64 *>(o + at - dist);
                        const unsigned long long w0 = src[0].v, w1 = src[1].v, w2 = src[2].v, w3 = src[3].v;
                        U64 *dst = reinterpret_cast<U64 *>(o + at);
                        dst[0].v = w0; dst[1].v = w1; dst[2].v = w2; dst[3].v = w3;
                    }
                }
                if (dist >= 8) {
                    for (; k + 8 <= len; k += 8, at += 8)
                        reinterpret_cast<U64 *>(o + at)->v = reinterpret_cast<const U64 *>(o + at - dist)->v;
                    // the last bytes: ONE more word loaded, its first bytes stored (byte by byte from memory every load waited
                    // for the store before it -- up to seven round trips behind every match)
                    if (k < len) {
                        const unsigned long long w = reinterpret_cast<const U64 *>(o + at - dist)->v;
                        for (uint32_t j = 0; k < len; ++k, ++at, ++j) o[at] = (uint8_t)(w >> (8u * j));
                    }
                } else {
                    // source and destination overlap: the last `dist` bytes repeat.  They are loaded once (the word's other
                    // bytes are whatever lies behind -- the output buffers end in spare bytes) and dealt out from the register.
                    // (A run -- distance 1, a few hundred bytes -- byte by byte was a few hundred turns of a loop that 63 other
                    // lanes waited for: the repeating bytes are spread over an 8-byte word that is stored whole and then turned
                    // on by 8 mod dist places: two shifts, no memory.)
                    unsigned long long w = reinterpret_cast<const U64 *>(o + at - dist)->v;
                    const uint32_t db = 8u * dist;                       // 8 .. 56
                    w &= (1ull << db) - 1ull;
                    w |= w << db;                                          // 2 dist bytes
                    if (2u * db < 64u) w |= w << (2u * db);                // 4 dist
                    if (4u * db < 64u) w |= w << (4u * db);                // 8 dist (dist == 1)
                    const uint32_t sh = 8u * (8u % dist);
                    for (; k + 8 <= len; k += 8, at += 8) {
                        reinterpret_cast<U64 *>(o + at)->v = w;
                        if (sh) w = (w >> sh) | (w << (db - sh));
                    }
                    for (uint32_t j = 0; k < len; ++k, ++at, ++j) o[at] = (uint8_t)(w >> (8u * j));
                }
                ob.after_copy(len);
            }
            if (!err && !eob) err = 8;                                   // the guard ran out before an end-of-block symbol
        } else {
            err = 1;
        }
        if (final_block || err) break;
    }
    retire();
    ob.flush();
    if (!err && b.past * 8 > b.cnt) err = 6;                            // bits were taken from beyond the block's end
    if (!err && ob.n != cap) err = 7;
    status[blk] = err;
}

// CRC-32 (the gzip polynomial, reflected) of every inflated block against the value in its trailer: one lane per block,
// four bytes per step through slicing tables in LDS.  A block whose sum differs gets 0x100 OR-ed into its status.
__global__ __launch_bounds__(64) void k_bgzf_crc(const uint8_t *__restrict__ out, const uint64_t *__restrict__ out_off,
                                                 const uint32_t *__restrict__ isize, const uint32_t *__restrict__ crc_expected,
                                                 uint64_t n_blocks, uint32_t *__restrict__ status, uint32_t lanes_per_wave) {
    __shared__ uint32_t T[4][256];
    for (uint32_t i = threadIdx.x; i < 256; i += 64) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (0xedb88320u ^ (c >> 1)) : (c >> 1);
        T[0][i] = c;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 256; i += 64) {
        uint32_t c = T[0][i];
        for (uint32_t t = 1; t < 4; ++t) { c = T[0][c & 0xffu] ^ (c >> 8); T[t][i] = c; }
    }
    __syncthreads();
    if (threadIdx.x >= lanes_per_wave) return;              // (after the barriers above; as few lanes a wave as k_bgzf_inflate, for its reasons)
    const uint64_t blk = (uint64_t)blockIdx.x * lanes_per_wave + threadIdx.x;
    if (blk >= n_blocks) return;
    const uint8_t *p = out + out_off[blk];
    const uint32_t n = isize[blk];
    uint32_t c = 0xffffffffu, i = 0;
    auto word = [&](uint32_t w) {
        c ^= w;
        c = T[3][c & 0xffu] ^ T[2][(c >> 8) & 0xffu] ^ T[1][(c >> 16) & 0xffu] ^ T[0][c >> 24];
    };
    for (; i < n && ((uintptr_t)(p + i) & 15u); ++i) c = T[0][(c ^ p[i]) & 0xffu] ^ (c >> 8);
    // sixteen bytes a turn, the next sixteen asked for before this turn's look-ups (round 4: a 4-byte load per turn, waited
    // for at once, made the sum a trip to memory per word -- 14 ms for 2.2 GB; the inflated stream ends in spare bytes)
    if (i + 16 <= n) {
        uint4 nxt = *reinterpret_cast<const uint4 *>(p + i);
        for (; i + 16 <= n; i += 16) {
            const uint4 cur = nxt;
            nxt = *reinterpret_cast<const uint4 *>(p + i + 16);
            word(cur.x); word(cur.y); word(cur.z); word(cur.w);
        }
    }
    for (; i + 4 <= n; i += 4) word(*reinterpret_cast<const uint32_t *>(p + i));
    for (; i < n; ++i) c = T[0][(c ^ p[i]) & 0xffu] ^ (c >> 8);
    if ((c ^ 0xffffffffu) != crc_expected[blk]) status[blk] |= 0x100u;
}

// Records of the BAM stream per block, on the premise that every block begins at a record boundary (htslib's writers
// flush the block before a record that does not fit it): the chain of block_size words from the block's first byte --
// from `records_begin` in the block that holds it: the header ends there -- must end exactly at the block's end.
// count[b] = its records; a block whose chain does not fit sets *unaligned (the caller then reads the file the host's
// way, which follows the chain serially).  Blocks before records_begin count nothing.
__global__ __launch_bounds__(64) void k_bam_count_blocks(const uint8_t *__restrict__ bytes, const uint64_t *__restrict__ out_off,
                                                         const uint32_t *__restrict__ isize, uint64_t n_blocks,
                                                         uint64_t records_begin, uint32_t *__restrict__ count,
                                                         uint64_t *__restrict__ span_begin, uint32_t *__restrict__ unaligned) {
    const uint64_t blk = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (blk > n_blocks) return;
    if (blk == n_blocks) { span_begin[blk] = out_off[blk - 1] + isize[blk - 1]; return; }
    const uint64_t b0 = out_off[blk], end = b0 + isize[blk];
    uint64_t off = b0 < records_begin ? records_begin : b0;
    span_begin[blk] = off;              // the spans tile [records_begin, end of the stream): blocks of the header are empty spans
    uint32_t k = 0;
    while (off + 4 <= end) {
        const uint32_t *w = reinterpret_cast<const uint32_t *>(bytes + (off & ~3ull));
        const uint32_t bs = __builtin_amdgcn_alignbyte(w[1], w[0], (uint32_t)off & 3u);
        if (bs < 32 || off + 4 + (uint64_t)bs > end) break;
        ++k;
        off += 4 + (uint64_t)bs;
    }
    if (off < end && end > records_begin) atomicOr(unaligned, 1u);       // the header's last block may end before records_begin
    count[blk] = k;
}

// span_first[b] = records before block b (64-bit, as ipx_launch_bam_decode takes them) from the 32-bit prefix
__global__ __launch_bounds__(256) void k_span_first(const uint32_t *__restrict__ prefix, const uint32_t *__restrict__ total,
                                                    uint64_t n_blocks, uint64_t *__restrict__ span_first) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n_blocks) span_first[i] = prefix[i];
    else if (i == n_blocks) span_first[i] = *total;
}

}  // namespace

size_t ipx_bgzf_tables_bytes(uint64_t n_blocks) { return (size_t)4 * LANE_WORDS * (n_blocks ? n_blocks : 1); }

static void launch_inflate(hipStream_t s, const uint8_t *comp, const uint64_t *in_begin, const uint64_t *in_end, const uint32_t *isize,
                           const uint64_t *out_off, uint64_t n_blocks, uint8_t *out, uint32_t *tables, uint32_t *status,
                           const uint32_t *order) {
    // Lanes per wave: as few as still let EVERY wave be resident at once (three per SIMD by the kernel's registers).  A wave's turn lasts as long as its slowest lane's, so fewer lanes per wave wait less for each other and
    // more waves overlap their trips to memory -- until the waves no longer fit the chip and run in rounds (measured on
    // 36 316 blocks: 64 lanes 101 ms, 32: 91, 16: 81, 12: 74 = 3027 waves on 3072 places; 8: a second round).
    // (the kernel's dynamic LDS is lpw * 512 bytes: the occupancy is asked for with the LDS of the lpw in question -- 32 KB a
    // wave at 64 lanes is five waves a CU, not the twelve that 8 KB allow -- advisor, round 4)
    static const uint32_t n_cus = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        return (uint32_t)cus;
    }();
    auto places_at = [&](uint32_t l) {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_bgzf_inflate, 64, (size_t)l * 512) != hipSuccess || per_cu <= 0) per_cu = 4;
        (void)hipGetLastError();
        return n_cus * (uint32_t)per_cu;
    };
    const char *e = getenv("IPX_INFLATE_LPW");                            // (experiments)
    uint32_t lpw = 64;
    if (e) {
        lpw = (uint32_t)atoi(e);
    } else {
        // the fewest lanes per wave whose waves all fit the chip at once (tried at the sizes that matter: the answer is
        // cached per size by the runtime's occupancy query being cheap, a few microseconds each)
        static const uint32_t cand[] = {8, 12, 16, 24, 32, 48, 64};
        for (uint32_t l : cand) {
            const uint32_t pl = places_at(l);
            if (ipx_div_up(n_blocks, l) <= pl - pl / 64) { lpw = l; break; }
        }
    }
    if (lpw < 1) lpw = 1;
    if (lpw > 64) lpw = 64;
    const dim3 grid((uint32_t)ipx_div_up(n_blocks, lpw));
    hipLaunchKernelGGL(k_bgzf_inflate, grid, dim3(64), (size_t)lpw * 512, s, comp, in_begin, in_end, isize, out_off, n_blocks, out, tables, status, order, lpw);
}

hipError_t ipx_launch_bgzf_inflate(hipStream_t s, const uint8_t *comp, const uint64_t *in_begin, const uint64_t *in_end,
                                   const uint32_t *isize, const uint64_t *out_off, const uint32_t *crc, uint64_t n_blocks,
                                   uint8_t *out, uint32_t *tables, uint32_t *status, const uint32_t *order) {
    launch_inflate(s, comp, in_begin, in_end, isize, out_off, n_blocks, out, tables, status, order);
    if (crc) {
        // (lanes per wave as in launch_inflate: the sum's kernel fits eight waves on a SIMD, six are aimed at)
        static const uint32_t crc_places = [] {
            int dev = 0, cus = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
            return (uint32_t)cus * 24u;
        }();
        uint32_t lpw = (uint32_t)ipx_div_up(n_blocks, crc_places);
        if (lpw < 1) lpw = 1;
        if (lpw > 64) lpw = 64;
        hipLaunchKernelGGL(k_bgzf_crc, dim3((uint32_t)ipx_div_up(n_blocks, lpw)), dim3(64), 0, s, out, out_off, isize, crc, n_blocks, status, lpw);
    }
    return hipGetLastError();
}

hipError_t ipx_launch_bam_count_blocks(hipStream_t s, const uint8_t *bytes, const uint64_t *out_off, const uint32_t *isize,
                                       uint64_t n_blocks, uint64_t records_begin, uint32_t *count, uint32_t *prefix, uint32_t *sums,
                                       uint32_t *total, uint64_t *span_begin, uint64_t *span_first, uint32_t *unaligned) {
    hipLaunchKernelGGL(k_bam_count_blocks, dim3((uint32_t)ipx_div_up(n_blocks + 1, 64)), dim3(64), 0, s, bytes, out_off, isize, n_blocks,
                       records_begin, count, span_begin, unaligned);
    hipError_t e = ipx_launch_scan_u32(s, count, prefix, n_blocks, sums, total);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_span_first, dim3((uint32_t)ipx_div_up(n_blocks + 1, 256)), dim3(256), 0, s, prefix, total, n_blocks, span_first);
    return hipGetLastError();
}

extern "C" int ipx_bgzf_inflate(int32_t device, const uint8_t *comp, uint64_t n_bytes, const uint64_t *in_off,
                                const uint32_t *isize, const uint64_t *out_off, uint64_t n_blocks, uint8_t *out,
                                uint64_t out_bytes, uint32_t *status, float *kernel_ms) {
    if (!comp || !in_off || !isize || !out_off || !out || !status || n_blocks == 0) return IPX_EINVAL;
    for (uint64_t b = 0; b < n_blocks; ++b)
        // (no sum that could wrap; output ranges ascending and apart, so that no two lanes write the same bytes)
        if (in_off[b] > in_off[b + 1] || in_off[b + 1] > n_bytes || isize[b] > 65536 || out_off[b] > out_bytes ||
            isize[b] > out_bytes - out_off[b] || (b > 0 && out_off[b] < out_off[b - 1] + isize[b - 1]))
            return IPX_EINVAL;
    if (hipSetDevice(device) != hipSuccess) return IPX_ENODEVICE;
    uint8_t *d_comp = nullptr, *d_out = nullptr;
    uint64_t *d_in_off = nullptr, *d_out_off = nullptr;
    uint32_t *d_isize = nullptr, *d_tables = nullptr, *d_status = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = IPX_OK;
    auto ok = [&](hipError_t e) { if (e != hipSuccess && rc == IPX_OK) rc = IPX_EDEVICE; return e == hipSuccess; };
    // (spare bytes behind both buffers, as ipx_decode_bgzf has them: the bit reader asks for the word behind the one in use,
    // a short match loads four whole words of its source)
    ok(hipMalloc(&d_comp, n_bytes + 16)) && ok(hipMalloc(&d_out, out_bytes + 64)) && ok(hipMalloc(&d_in_off, 8 * (n_blocks + 1))) &&
        ok(hipMalloc(&d_out_off, 8 * n_blocks)) && ok(hipMalloc(&d_isize, 4 * n_blocks)) &&
        ok(hipMalloc(&d_tables, 4ull * LANE_WORDS * n_blocks)) && ok(hipMalloc(&d_status, 4 * n_blocks)) &&
        ok(hipEventCreate(&e0)) && ok(hipEventCreate(&e1));
    if (rc == IPX_OK) {
        ok(hipMemcpy(d_comp, comp, n_bytes, hipMemcpyHostToDevice));
        ok(hipMemset(d_comp + n_bytes, 0, 16));
        ok(hipMemcpy(d_in_off, in_off, 8 * (n_blocks + 1), hipMemcpyHostToDevice));
        ok(hipMemcpy(d_out_off, out_off, 8 * n_blocks, hipMemcpyHostToDevice));
        ok(hipMemcpy(d_isize, isize, 4 * n_blocks, hipMemcpyHostToDevice));
    }
    if (rc == IPX_OK) {
        ok(hipEventRecord(e0, nullptr));
        launch_inflate(nullptr, d_comp, d_in_off, (const uint64_t *)nullptr, d_isize, d_out_off, n_blocks, d_out, d_tables, d_status,
                       (const uint32_t *)nullptr);
        ok(hipGetLastError());
        ok(hipEventRecord(e1, nullptr));
        ok(hipEventSynchronize(e1));
        if (rc == IPX_OK && kernel_ms) ok(hipEventElapsedTime(kernel_ms, e0, e1));
        ok(hipMemcpy(out, d_out, out_bytes, hipMemcpyDeviceToHost));
        ok(hipMemcpy(status, d_status, 4 * n_blocks, hipMemcpyDeviceToHost));
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(d_comp); (void)hipFree(d_out); (void)hipFree(d_in_off); (void)hipFree(d_out_off);
    (void)hipFree(d_isize); (void)hipFree(d_tables); (void)hipFree(d_status);
    return rc;
}
