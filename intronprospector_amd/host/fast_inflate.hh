// fast_inflate.hh -- raw DEFLATE (RFC 1951) for BGZF blocks.
//
// Stands where htslib's bgzf.c calls inflate() on one BGZF block (the reference reads its BAM through
// sam_read1, src/junctions_extractor.cc:620-634): the whole compressed block and the size of its output are known
// before the first bit is read, so the decoder needs no streaming state, no window copy and no output checks
// outside a short tail -- which is where zlib 1.2.11's byte-at-a-time bit buffer spends the time: inflating
// was 7 of the command line's 8 CPU-seconds on a 10 M-record BAM.
//   * 64-bit bit buffer, refilled with one unaligned 8-byte load per symbol pair;
//   * literal/length table of 2^11 entries and distance table of 2^8 entries with second-level tables behind
//     them for longer codes; an entry carries the base value and the number of extra bits, so a length or a
//     distance is one look-up, one mask, one add;
//   * matches are copied eight bytes at a time where source and destination do not overlap closer than that.
// Anything malformed (over-subscribed codes, a distance before the start of the output, output or input
// running over) returns -1; the caller (host/bgzf_reader.cc) then lets zlib look at the block, so what the
// user is told about a bad file does not change.
#ifndef IPX_HOST_FAST_INFLATE_HH
#define IPX_HOST_FAST_INFLATE_HH
#include <cstddef>
#include <cstdint>

namespace ipxhost {

// in[0, in_n): a complete raw DEFLATE stream (every block up to and including the final one); out: room for
// out_cap bytes.  Returns the number of bytes written, or -1.
long fast_inflate(const uint8_t *in, size_t in_n, uint8_t *out, size_t out_cap);

// CRC-32 (IEEE 802.3, the gzip one) of buf[0, n), continuing from crc (0 to start): sixteen table look-ups per
// sixteen bytes instead of zlib 1.2.11's four per four
uint32_t fast_crc32(uint32_t crc, const uint8_t *buf, size_t n);

}  // namespace ipxhost
#endif
