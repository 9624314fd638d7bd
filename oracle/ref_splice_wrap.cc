/* ref_splice_wrap.cc -- C entry points over the REFERENCE's own splice_juncs.cc.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is ours; the reference source it wraps
 * is compiled where it lies (/root/reference/src/splice_juncs.cc) by
 * oracle/Makefile into oracle/_ref/libref_splice.so (git-ignored).  It is the
 * only part of the reference's hot path that builds from its own sources with
 * nothing but a C++ compiler: every other file includes htslib headers, and
 * htslib is not in this image ("Oracle").
 */
#include "splice_juncs.hh"
#include <stdexcept>

extern "C" int ref_is_canonical(const char *s) { return is_canonical(std::string(s)) ? 1 : 0; }

extern "C" int ref_junction_filter_check(int filter, const char *s) {
    return junction_filter_check((JunctionFilter)filter, std::string(s)) ? 1 : 0;
}

extern "C" int ref_junction_filter_parse(const char *spec) {
    try {
        return (int)junction_filter_parse(std::string(spec));
    } catch (const std::invalid_argument &) {
        return -1;
    }
}
