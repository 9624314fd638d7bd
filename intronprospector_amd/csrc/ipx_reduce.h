// ipx_reduce.h -- buffers and launchers of the segmented reduce (ipx_reduce.hip).
#pragma once
#include "ipx_common.h"

struct SortedItems {
    const uint4 *items;           // records sorted by key
    uint64_t n;
};

struct ReduceBuffers {
    // per 512-record chunk, exclusive prefix of segment heads
    uint32_t *chunk_j, *chunk_b;
    uint32_t n_j, n_b;            // junction keys / (key, off16) bins before filtering
    // per junction key (n_j)
    uint32_t *jrank, *jstart, *jend, *jmotif, *amin, *amax, *jvalid, *jpos, *jfirstbin;
    uint8_t *jstrand;
    unsigned long long *first, *creator, *counts;
    // per bin (n_b)
    uint32_t *bin_off, *bin_count, *bin_flag, *bin_pos;
    uint32_t *scan_sums;          // >= div_up(max(n_j, n_b), 4096) + 1
    uint32_t *totals;             // [0] valid junctions, [1] non-empty bins
    // outputs (device), sized n_j / n_b
    uint32_t *out_target, *out_intron_start, *out_intron_end, *out_anchor_start, *out_anchor_end;
    uint8_t *out_strand;
    uint16_t *out_motif;
    uint64_t *out_read_counts, *out_creator_order, *out_hist_begin;
    uint32_t *out_ijunc;
    uint16_t *out_hist_value;
    uint32_t *out_hist_count;
    uint4 *rank_a, *rank_b;       // creation-order sort for ijunc (n_j)
};

uint64_t ipx_reduce_chunks(uint64_t n);
// *unsorted (device): |= 1 when two neighbouring items are out of order
hipError_t ipx_launch_count_heads(hipStream_t s, const SortedItems &si, uint32_t *chunk_j, uint32_t *chunk_b, uint32_t *unsorted);
hipError_t ipx_launch_reduce(hipStream_t s, const SortedItems &si, const uint4 *pays, const uint32_t *cnts, const KeyLayout &kl,
                             const ReduceBuffers &rb, const GenomeDev &genome, int32_t sj_filter,
                             int32_t skip_missing, const uint32_t *tid_of_rank, Counters *counters);
hipError_t ipx_launch_emit_ijunc(hipStream_t s, uint32_t n_valid, const uint4 *sorted_rank_items, uint32_t *ijunc);

// ijunc without a sort: the creation-order keys (rank_items[o].x/.y, row o) differ on few bits; squeezed to
// those bits they index a bitmap, and a key's rank is the number of set bits below its own.
// runs: up to 4 (lo, width) runs of varying bits, least significant first, `bits` wide in all (<= 28).
struct RankBits {
    uint32_t n_runs, bits;
    uint32_t lo[4], width[4];
};
// workspace: 2 * 2^bits / 32 + 2^bits / 32 / 4096 + 2 words
size_t ipx_rank_bitmap_workspace_bytes(uint32_t bits);
hipError_t ipx_launch_rank_by_bitmap(hipStream_t s, uint32_t n_valid, const uint4 *rank_items, const RankBits &rb,
                                     uint32_t *workspace, uint32_t *ijunc);
hipError_t ipx_launch_find_genome_error(hipStream_t s, uint32_t n_j, const unsigned long long *first,
                                        const uint32_t *jmotif, const uint32_t *jrank,
                                        const uint32_t *jstart, const uint32_t *jend,
                                        unsigned long long err_seq, uint32_t *detail4 /* rank,start,end,found */);
