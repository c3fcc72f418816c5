// kernels.h -- launchers of the gfx950 kernels (definitions in kernels.hip). All asynchronous on `stream`.
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"
#include "bam_core.h"

namespace rgx {

constexpr uint32_t kSegBytes = 16384;   // arena segment walked by one lane during record-boundary discovery
constexpr uint32_t kSparseRecordBytes = 2048;   // mean record size above which k_decode_seg reads records in place instead of staging segments
constexpr uint32_t kSegCpSlots = 64;     // checkpoints per segment: offset of every 8th record (a segment starts at most 457 records)
constexpr uint64_t kChainEnd = ~0ull;
constexpr uint64_t kChainUnknown = ~0ull - 1;   // exit of a segment in which the guess found nothing: no claim at all (not "the chain ended")   // "the record chain ended before this point" (truncated/corrupt stream)

// ---- a1: BGZF inflate (one lane per member) -----------------------------------------------------------
// member m writes its bytes at arena + (members[m].upos - upos_bias)
// len_scratch: inflate_scratch_bytes(n_members) bytes of device memory (code-length scratch of the block headers)
size_t inflate_scratch_bytes(uint32_t n_members);
// an error of a kernel launch (refused configuration) or of a launch's set-up since this host thread last asked; hipSuccess = none (clears it)
hipError_t pending_launch_error();
// Arrival gate of the overlapped upload (round 4): ONE k_inflate_coop launch covers the whole range while the file is still on the bus; a wave
// starts once the upload chunk holding the last byte its members need has landed -- flags[k] == epoch, written by a 4-byte copy queued on the
// copy stream right behind chunk k (api_front.cpp stage_upload).  flags = nullptr: no gate.
struct InflateGate {
    const uint32_t *flags = nullptr;   // device memory, one word per upload chunk
    uint32_t epoch = 0;                // this call's value (the words keep earlier calls' values: smaller)
    uint32_t n_chunks = 0;
    uint64_t lo = 0, chunk_bytes = 1;  // chunk k = bytes [lo + k * chunk_bytes, lo + (k + 1) * chunk_bytes) of the file (the last one to the range's end)
    // Early tail (round 4): the launch's waves are counted as they finish, by part of the member list -- part j = the waves from part_start[j - 1]
    // (workgroup index; part 0 starts at 0) up to part_start[j] -- so that the pipeline's stream can frame and decode the front parts of the arena
    // (launch_wait_done) while the waves of the later parts still run.  done = null: nobody counts.
    uint32_t *done = nullptr;          // device memory, kGateParts words, zeroed by the caller in front of the launch
    uint32_t part_start[7] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
};
constexpr uint32_t kGateParts = 8;
constexpr uint32_t kInflateSortGroup = 1024;   // k_inflate_coop: the lanes of a wave take their members from one group of that many consecutive ones (k_member_sort)
// one lane on `stream` waits until done[0] reaches `expected` (a wave count); gives up after ~2 s and sets *timed_out
void launch_wait_done(const uint32_t *done, uint32_t expected, uint32_t *timed_out, hipStream_t stream);
constexpr uint32_t kStatusEarly = 76;   // status[kStatusEarly..+1]: the same pair for members below ignore_below (only written when that is > 0)
void launch_inflate(const uint8_t *comp, const Member *members, uint32_t n_members, uint8_t *arena, uint64_t upos_bias, uint32_t *len_scratch,
                    uint32_t *status /* [0]=first bad member (min), [1]=its status */, hipStream_t stream, uint32_t ignore_below = 0 /* failures of members below this index are not reported */,
                    uint32_t index_bias = 0 /* added to the launch's member index in status[] and in the ignore_below test: a range launched in pieces */,
                    bool piece = false /* one of several concurrent launches over a range (own kernel symbol, same code) */,
                    int form = 0 /* 0 = chosen by member count / REGTOOLS_AMD_INFLATE; 1 = k_inflate, 2 = k_inflate_wave, 3 = k_inflate_ring */,
                    uint8_t *bad_flags = nullptr /* optional, zeroed by the caller: [index in the caller's range] = 1 for every member that did not inflate */,
                    int plan = 1 /* inflate_plan_for */,
                    bool check_layout = false /* a caller's own member list (stage entry point): k_inflate_coop only runs when the list's layout suits it */,
                    InflateGate gate = InflateGate() /* k_inflate_coop only: waves wait for their upload chunk */);
// whether launch_inflate would pick k_inflate_coop for a range of this size (the gate needs it)
bool inflate_takes_coop(uint32_t n_members);
void launch_gate_set(uint32_t *flag, uint32_t epoch, hipStream_t stream);      // flags[k] = epoch, in stream order behind chunk k's copy
// Which options suit a payload, from how well the file compresses (round 4, 50 M-read files / 10 M long reads on one box, ms, all byte-equal to zlib;
// tools/lab/forms_r4.sh, profiles/r04_inflate_forms.txt):
//   inflated / compressed > 32  (long reads: run-length copies)        k_inflate_coop, windowed bit reader, one symbol per trip, lanes in file order:
//                                                                      117.5-118.7 (round 3's choice -- plain reader -- 120.7-122.2; pairs + sorted lanes 122; k_inflate 131)
//   <= 32                       (the bench payload: 21;                 k_inflate_coop, windowed bit reader, literal pairs, lanes sorted by compressed length
//                                random bases + binned qualities: 3.6)  per 1024 members: bench payload 14.5-15.2 (k_inflate 19.9-21.3); random bases 44.9-47.5
//                                                                      (k_inflate with four literals per trip, round 3's choice for them: 53.1-54.1)
// bit 0 = literal pairs + sorted lanes.  (k_inflate stays selectable: REGTOOLS_AMD_INFLATE=lane, form 1 / 5 of the stage entry point.)
// Second half of round 4 (k_inflate_coop's mode word, launch_inflate): runs at distances of 16 bytes or less are written from registers, 64 bytes a
// trip, for every class (long reads 121.2 -> 79.8 ms: half their trips were such runs growing 1, 2, 4 ... bytes a trip, each with a partial-chunk
// store); up to 32 x also a literal pair and the match behind it in one trip, bits counted exactly (bench payload 2,030 -> 1,463 trips per
// member, 15.2 -> 14.1-15.0 ms: the launch follows its memory requests, not its trips).
inline int inflate_plan_for(uint64_t compressed_bytes, uint64_t inflated_bytes) {
    return compressed_bytes * 32 > inflated_bytes ? 1 : 0;
}

// ---- a1 (container): BGZF member discovery on the device --------------------------------------------------
// The member chain (bgzf.c:525: next = this + BSIZE + 1) is serial on a CPU (one dependent cache miss per member).
// Here: (1) every byte offset is tested against check_header's predicate (bgzf.c:348-355) -> sorted candidate list
// (two passes: count per 4 KiB tile, scan, fill); (2) each candidate links to the candidate at off+BSIZE+1 (binary
// search); (3) reachability from offset 0 by pointer doubling (log2 n rounds) keeps exactly the members the serial walk
// would visit -- false positives inside compressed data are unreachable and drop out; (4) compaction + scan of ISIZE.
constexpr uint32_t kMagicTile = 4096;
void launch_magic_count(const uint8_t *bam, uint64_t len, uint32_t n_tiles, uint32_t *tile_cnt, hipStream_t stream);
void launch_magic_fill(const uint8_t *bam, uint64_t len, uint32_t n_tiles, const uint32_t *tile_base, uint64_t *cand, hipStream_t stream);
// next[i] = index of the candidate at cand[i] + BSIZE + 1, or n when the chain ends there; isize[i] = ISIZE footer
void launch_member_link(const uint8_t *bam, uint64_t len, const uint64_t *cand, uint32_t n, uint32_t *next, uint32_t *isize,
                        uint32_t *reach, uint64_t root2 /* second chain root: compressed offset of a seek target, ~0 = none */, hipStream_t stream);
void launch_member_jump(uint32_t n, const uint32_t *next_in, uint32_t *next_out, uint32_t *reach, hipStream_t stream);
// members[rank] for reachable candidates (rank = exclusive scan of reach); upos filled later by launch_member_upos
void launch_member_compact(const uint8_t *bam, uint64_t len, const uint64_t *cand, const uint32_t *isize, const uint32_t *reach, const uint32_t *rank,
                           uint32_t n, Member *members, uint32_t *isize_compact, hipStream_t stream);
// 64-bit exclusive scan of isize over the compacted members (single workgroup; n is ~1e5) -> Member::upos, total
void launch_member_upos(Member *members, const uint32_t *isize_compact, const uint32_t *n_members /*device*/, uint64_t *total, hipStream_t stream);
// answers host questions about the member list without copying it: q_coff[k] -> index of the member whose file offset is
// q_coff[k]-18+18 (exact match) or n; also the first index >= from[k] whose isize is 0 or > 65536 (stream stop rule)
void launch_member_query(const Member *members, const uint32_t *n_members /*device*/, const uint64_t *q_coff, uint32_t n_q, uint32_t *q_index,
                         uint64_t *q_upos, hipStream_t stream);
// *stop = min(*stop, first index >= *from whose ISIZE is 0 or > 65536)
// repair of files whose ISIZE footers do not say what the members inflate to (the reference never reads ISIZE, bgzf.c:292-316):
// launch_inflate_probe puts member m into slots + m * 64 KiB and its true length (~0 = does not inflate) into sizes[m];
// launch_member_fix makes those lengths the members' sizes before the offsets are scanned.
void launch_inflate_probe(const uint8_t *comp, const Member *members, uint32_t n_members, uint8_t *slots, uint32_t *len_scratch, uint32_t *sizes,
                          hipStream_t stream);
void launch_member_fix(Member *members, uint32_t *isize_compact, uint32_t max_members, const uint32_t *n_members /*device*/, const uint32_t *fix,
                       hipStream_t stream);
void launch_member_stop(const Member *members, uint32_t max_members, const uint32_t *n_members /*device*/, const uint32_t *from /*device*/, uint32_t *stop,
                        hipStream_t stream);

// ---- a2: record framing ---------------------------------------------------------------------------------
// Which bytes of the arena hold the record stream.  One chain (whole-file runs, shards): segment s covers arena [pos0 + s*kSegBytes,
// +kSegBytes) clipped to lim, and only segment 0 starts at an exact offset.  Region queries: ONE CHAIN PER CHUNK the reference's iterator
// reads (hts_itr_next, hts.c:1924-1965: a seek to the chunk's begin, then records as long as the position in front of the next one is
// below the chunk's end -- the first record after the seek is read whatever its position): chunk c owns segments [seg_base, next
// chunk's seg_base), its first segment starts exactly at a, its chain is followed up to b, and dlim is where the bytes a reader that
// started at a can get end (the next empty or unreadable member).  The chunk table lives in HBM; null = the one chain.
struct SegChunk { uint64_t a, b, dlim; uint32_t seg_base, pad; };
// REGTOOLS_AMD_DECODE="lane|wave[,seg_bytes]" (tests): long-record files through the workgroup-per-segment decode the lane form replaced, and either
// segment size (16384 / 131072) forced onto any file -- the extraction tests run the long-record path on short-read files with it.
struct DecodeKnobs { bool wave_form = false; int seg_bytes = 0; };
const DecodeKnobs &decode_knobs();
struct SegGeom {
    uint64_t pos0, lim;
    uint64_t data_end;             // end of the inflated bytes (k_decode_seg's staging window may reach past a chunk's end)
    const SegChunk *chunks; uint32_t n_chunks;
    uint32_t seg_bytes;            // kSegBytes, or kSegBytesLong for files of long records (the checkpoints and k_decode_seg only exist for kSegBytes)
    uint32_t lite_walk;            // round 4: the chain walk reads a record's block_size only (one request per record instead of seven dwords over
                                   // two lines); bam_read1's other acceptance tests (sam.c:421-423) are then made by the decode pass on the head it
                                   // reads anyway (ExtractCfg::insane_out), and a record that fails them sends the call through the full walk
};
constexpr uint32_t kSegBytesLong = 131072;     // a lane's first-record search reads, on average, half a record of sequence and quality: with records of
                                               // kilobytes (long reads) 16 KiB segments spend their time there (config 5: k_seg_walk 11.9 ms of a 47 ms tail)
constexpr uint32_t kLongRecordBytes = 2048;    // mean record size (estimated on the host from the file's first records) from which the long segments are used
// seg_start[s] = guessed (or, for a chain's first segment, exact) first record start >= segment begin; seg_exit[s] = first record
// start >= segment end reached by the chain from seg_start[s]; seg_cnt[s] = records that start inside the segment.
void launch_seg_walk(const uint8_t *arena, SegGeom g, uint32_t n_seg, int32_t n_ref,
                     uint64_t *seg_start, uint64_t *seg_exit, uint32_t *seg_cnt, uint16_t *seg_cp /* n_seg * kSegCpSlots */, hipStream_t stream,
                     uint32_t s_begin = 0 /* only segments [s_begin, n_seg): the guesses of a range can be made as soon as its bytes are inflated */);
// One verification sweep: segment s re-walks from seg_exit_in[s-1] when that differs from seg_start_in[s].
// *changed is incremented when anything changed. Reads *_in, writes *_out (all segments).
void launch_seg_verify(const uint8_t *arena, SegGeom g, uint32_t n_seg,
                       const uint64_t *seg_start_in, const uint64_t *seg_exit_in, const uint32_t *seg_cnt_in,
                       uint64_t *seg_start_out, uint64_t *seg_exit_out, uint32_t *seg_cnt_out,
                       uint32_t *status /* [0] leftmost disagreeing segment, [1] leftmost chain end; both preset to ~0u */, uint16_t *seg_cp, hipStream_t stream);
void launch_seg_truncate(SegGeom g, uint32_t n_seg, uint32_t last, uint64_t *seg_start, uint64_t *seg_exit, uint32_t *seg_cnt, hipStream_t stream);

// ---- a2/a3/a5/a6: SoA decode + per-read event count ---------------------------------------------------------
struct ReadSoA {
    int32_t  *tid, *pos;
    uint32_t *flag_nc;     // flag << 16 | n_cigar
    uint64_t *cig_off;     // arena offset of the CIGAR array
    uint8_t  *strand;      // per-read strand char from the tag (XS mode) or the flag rule (RF/FR)
    uint32_t *n_ev;        // junction events this read contributes (after region filter and junction_qc)
    uint64_t *rec_off;     // optional (may be null): arena offset of the record's block_size word (-b barcodes re-read the aux block)
};
struct ExtractCfg {
    int32_t  n_ref;
    int32_t  strandness;           // 0 tag, 1 RF, 2 FR, 3 intron-motif
    uint8_t  tag0, tag1;
    uint32_t min_anchor, min_intron, max_intron;
    int32_t  region_tid;           // -2 = whole file
    int32_t  region_beg, region_end;
    uint32_t long_threshold;       // reads with more CIGAR ops than this go to the wave-per-read kernel
    // intron-motif strand rule (a FASTA was given): file bytes + one descriptor per BAM contig, both in HBM
    const uint8_t *fa_data;
    const struct FaContig *fa_tab;
    uint32_t *fa_missing;          // [0] set to 1+tid when a junction lies on a contig the FASTA does not have
    // the iterator's end rule (hts_itr_next, hts.c:1946-1950: the first record read whose tid is not the region's or whose pos is not below
    // its end finishes the iteration, whatever follows).  stop_out (null = off): [0] = smallest index of such a record (preset ~0),
    // [1] = 1 + largest index of a record that passed the overlap test (preset 0); records at or behind stop_index are not iterated.
    uint32_t *stop_out;
    uint32_t stop_index;
    uint32_t *insane_out;          // null = the framing made bam_read1's acceptance test itself; else [0] is set when a decoded record fails it (SegGeom::lite_walk)
    // -s XS: the smallest index of an iterated read with an N operation whose strand tag lies behind an aux field of unknown type (preset ~0; null =
    // off): the reference's bam_aux_get abort()s on it (sam.c:1233-1252) when the read's first junction asks for its strand (junctions_extractor.cc:283-286)
    uint32_t *abort_out;
    uint8_t  bc0, bc1;             // -b: the barcode tag (0 = no -b).  set_junction_barcode asks for it on every read with more than one CIGAR operation, before the
                                   // CIGAR is looked at (junctions_extractor.cc:393-395): a field of unknown type in front of it (or anywhere, without it) is abort_out's too
    // `identify -s XS` (one extraction for every window): such reads are COUNTED here (null = off) and marked in bit 7 of their row's strand byte; upstream
    // abort()s in the first window -- in the order of the variants -- that reads one of them (launch_collect_odd_aux, cse_api.cpp)
    uint32_t *odd_count;
};
// the reads the decode kernels marked (ExtractCfg::odd_count): out[3 k ..] = tid, pos, bam_endpos of the k-th found (any order); *count = how many (may exceed cap)
void launch_collect_odd_aux(const uint8_t *arena, ReadSoA soa, uint32_t n_rec, uint32_t cap, uint32_t *count, int32_t *out, hipStream_t stream);

// one wave per framing segment, segment bytes staged through LDS (replaces launch_seg_fill + launch_decode on the hot path)
// seg_iter[s] = records of segment s that pass the region filter; seg_long[s] = its reads for the wave-per-read kernel
void launch_decode_seg(const uint8_t *arena, SegGeom g, uint32_t n_seg, const uint64_t *seg_start, const uint32_t *seg_base,
                       const uint32_t *seg_cnt, ExtractCfg cfg, ReadSoA soa, uint32_t *seg_iter, uint32_t *seg_long, const uint16_t *seg_cp,
                       bool staged /* segment bytes through LDS (short records) or read in place (long records) */, hipStream_t stream,
                       uint32_t s_begin = 0 /* only segments [s_begin, n_seg): the early tail decodes the file's front part first */);
void launch_long_fill(uint32_t n_seg, const uint32_t *seg_base, const uint32_t *seg_cnt, const uint32_t *seg_long_base, ExtractCfg cfg, ReadSoA soa,
                      uint32_t *long_list, hipStream_t stream);

// ---- a4: CIGAR scan + junction emit ---------------------------------------------------------------------------
struct EventSoA {
    uint32_t *tid, *start, *ilen_cls;   // key words: tid | start | (end-start) << 2 | strand class
    uint32_t *ts, *te;                  // thick_start / thick_end of this read's instance
    uint32_t *rpos, *rend;              // optional (may be null): the supporting read's pos / bam_endpos (window join of `identify`)
    uint8_t  *strand;
    uint32_t *read;                     // optional (may be null): index of the supporting read (-b barcodes)
};
void launch_emit_short(const uint8_t *arena, uint32_t n_rec, ExtractCfg cfg, ReadSoA soa, const uint32_t *ev_base,
                       EventSoA ev, hipStream_t stream, uint32_t row_begin = 0 /* rows [row_begin, n_rec) */,
                       const uint32_t *part_totals = nullptr, uint32_t n_parts = 0 /* device words added to every ev_base (early tail: ev_base counts per part) */,
                       uint32_t slot_cap = 0xffffffffu /* events at or behind this slot are counted, not written */);
void launch_emit_long(const uint8_t *arena, const uint32_t *long_list, uint32_t n_long,
                      ExtractCfg cfg, ReadSoA soa, const uint32_t *ev_base, EventSoA ev, hipStream_t stream);

// ---- primitives --------------------------------------------------------------------------------------------------
// exclusive scan of n uint32 (out may alias in); *total (device) receives the sum. tmp needs scan_tmp_words(n) words.
size_t scan_tmp_words(uint32_t n);
void launch_scan_u32(const uint32_t *in, uint32_t *out, uint32_t n, uint32_t *total, uint32_t *tmp, hipStream_t stream);

// One stable LSD radix pass on a permutation: digit(i) = (word[perm_in ? perm_in[i] : i] >> shift) & (2^bits-1), bits<=8.
size_t radix_tmp_words(uint32_t n);
void launch_radix_pass(const uint32_t *word, uint32_t shift, uint32_t bits, const uint32_t *perm_in, uint32_t *perm_out,
                       uint32_t n, uint32_t *tmp, hipStream_t stream);
// The same pass with the keys travelling along: position i's key is key_in[i] (streamed, no gather through the permutation), and
// key_out receives the keys in the new order.  perm_in null = identity.  A multi-pass sort on one word gathers that word once.
void launch_radix_pass_keyed(const uint32_t *key_in, uint32_t *key_out, uint32_t shift, uint32_t bits, const uint32_t *perm_in, uint32_t *perm_out,
                             uint32_t n, uint32_t *tmp, hipStream_t stream);

// ---- a7: segmented reduce ------------------------------------------------------------------------------------------
struct UniqueSoA {
    uint32_t *tid, *start, *end, *ts_min, *te_max, *count, *first_seen, *last_seen, *name_rank;
    uint8_t  *strand;
};
// Partial rows (round 4, k_preagg): one row per distinct key of a tile of consecutive events; reduce_events sorts and reduces these instead of the events.
struct PartialSoA { uint32_t *tid, *start, *ilen_cls, *ts, *te, *count, *first, *last; };
void launch_preagg(EventSoA ev, uint32_t n, PartialSoA p, uint32_t *p_total /* device, zeroed by the caller: rows appended */, hipStream_t stream);
// u.count / te_max / last_seen pre-filled with 0, u.ts_min / first_seen with 0xffffffff
void launch_reduce_partials(PartialSoA p, const uint32_t *perm, const uint32_t *head, const uint32_t *seg_excl, uint32_t n, UniqueSoA u, hipStream_t stream);
void launch_reduce_finish_partials(const uint8_t *ev_strand, uint32_t n_unique, UniqueSoA u, uint32_t *first_flag /* one word per EVENT, zeroed */, hipStream_t stream);
// head[i] = 1 where sorted position i starts a new key
void launch_heads(EventSoA ev, const uint32_t *perm, uint32_t n, uint32_t *head, hipStream_t stream);
// seg_excl = exclusive scan of head. ts_min must be pre-filled with 0xffffffff, te_max with 0.
void launch_reduce(EventSoA ev, const uint32_t *perm, const uint32_t *head, const uint32_t *seg_excl, uint32_t n, UniqueSoA u,
                   uint32_t *head_pos, hipStream_t stream);
// count / last_seen / strand per row; first_flag[first_seen[row]] = 1 (first_flag zeroed by the caller, n entries)
void launch_reduce_finish(EventSoA ev, const uint32_t *perm, uint32_t n, uint32_t n_unique, const uint32_t *head_pos, UniqueSoA u,
                          uint32_t *first_flag, hipStream_t stream);
// name_rank[row] = 1 + exclusive_scan(first_flag)[first_seen[row]]
void launch_name_rank(uint32_t n_unique, const uint32_t *flag_scan, UniqueSoA u, hipStream_t stream);
void launch_gather_u32(uint32_t n, const uint32_t *table, const uint32_t *idx, uint32_t *out, hipStream_t stream);
void launch_fill_u32(uint32_t *p, uint32_t v, size_t n, hipStream_t stream);
// ten u32 columns of n rows each, in `order`: tid,start,end,ts,te,count,name_rank,first_seen,last_seen,strand
void launch_rows_out(UniqueSoA u, const uint32_t *order, uint32_t n, uint32_t *out, hipStream_t stream);
// The result table's host block, written on the device so that ONE copy fills every column (api_ctx.cpp table_alloc): m = padded row count;
// u64 name_index[m], first_seen[m], last_seen[m]; u32 tid[m], start[m], end[m], thick_start[m], thick_end[m], read_count[m]; u8 strand[m], left_ok[m], right_ok[m]
RGX_HD size_t table_block_rows(uint64_t n) { return ((size_t)n + 1 + 15) & ~(size_t)15; }
RGX_HD size_t table_block_bytes(uint64_t n) { return table_block_rows(n) * (8 * 3 + 4 * 6 + 3); }
void launch_rows_table(UniqueSoA u, const uint32_t *order, uint32_t n, uint32_t min_anchor, uint8_t *out, hipStream_t stream);


// ---- -b: barcode counts per junction (junctions_extractor.cc:362-374, :204-217; barcode_kernels.hip) -------------------------------
// per event: where its read's barcode string lies in the arena (off = ~0 -> the literal "?"), the 64-bit grouping hash, the junction's output row
struct BarcodeEv { uint64_t *off; uint32_t *len, *h_lo, *h_hi, *row; };
// ev_urow[e] = unique row of event e (from the group-by's sorted order); must run before head / seg_excl are reused
void launch_event_urow(const uint32_t *sorted, const uint32_t *head, const uint32_t *seg_excl, uint32_t n, uint32_t *ev_urow, hipStream_t stream);
void launch_inverse_perm(const uint32_t *perm, uint32_t n, uint32_t *inv, hipStream_t stream);
// flags[0] = 1 when some read's tag is present but not a string
void launch_bc_event_keys(const uint8_t *arena, uint32_t n_events, const uint32_t *ev_read, const uint64_t *rec_off, const uint32_t *ev_urow,
                          const uint32_t *urow_pos, uint8_t t0, uint8_t t1, BarcodeEv b, uint32_t *flags, hipStream_t stream);
// head[i] = sorted position i starts a new (row, barcode); equal hashes with different bytes set flags[1]
void launch_bc_heads(const uint8_t *arena, BarcodeEv b, const uint32_t *perm, uint32_t n, uint32_t *head, uint32_t *flags, hipStream_t stream);
// one output row per head: pair_row, pair_first (earliest event), pair_pos (sorted position of the head), pair_off / pair_len (the string)
void launch_bc_pairs(BarcodeEv b, const uint32_t *perm, const uint32_t *head, const uint32_t *seg_excl, uint32_t n, uint32_t *pair_row,
                     uint32_t *pair_first, uint32_t *pair_pos, uint64_t *pair_off, uint32_t *pair_len, hipStream_t stream);
void launch_bc_counts(uint32_t n_pairs, uint32_t n, const uint32_t *pair_pos, uint32_t *pair_count, hipStream_t stream);
// text[str_begin[k] .. +pair_len[k]) = the k-th pair's barcode
void launch_bc_gather(const uint8_t *arena, uint32_t n_pairs, const uint64_t *pair_off, const uint32_t *pair_len, const uint32_t *str_begin,
                      uint8_t *text, hipStream_t stream);

// ---- a9-a11: `cis-splice-effects identify` interval kernels (cse_kernels.hip; logic in cse_core.h) --------------------------
struct GtfView;
struct VariantOpts;
}  // namespace rgx
#include "cse_core.h"
namespace rgx {
// count pass (fill = false): count[i], ces[i], cee[i]; fill pass: hit_tx[base[i]+k], hit_ad[2*(base[i]+k)] = annotation, +1 = distance
void launch_variant_scan(bool fill, GtfView g, uint32_t n, const int32_t *chrom, const uint32_t *pos0, VariantOpts o, uint32_t *count, const uint32_t *base,
                         uint32_t *ces, uint32_t *cee, uint32_t *hit_tx, uint32_t *hit_ad, unsigned long long *visits /* count pass: += exon records visited */,
                         hipStream_t stream, uint32_t *last_score = nullptr /* count pass: upstream's variant.score behind the walk (cse_core.h) */);
// count pass: count[i], flags[i] = known_donor | known_acceptor<<1 | known_junction<<2; fill pass: items (kind,a,b) in visitation order
void launch_junction_scan(bool fill, GtfView g, uint32_t n, const int32_t *chrom, const uint32_t *js, const uint32_t *je, const uint8_t *strand, uint32_t *count,
                          const uint32_t *base, uint32_t *flags, uint32_t *item_kind, uint32_t *item_a, uint32_t *item_b, unsigned long long *visits,
                          uint32_t *visit_each /* wave form: exon records per junction */, hipStream_t stream);
void launch_max_span(EventSoA ev, uint32_t n, uint32_t *out /* zeroed by the caller */, hipStream_t stream);
constexpr uint32_t kWinSlices = 4;      // workgroups per variant window (k_window_pairs); count / base arrays hold n_win x kWinSlices entries
void launch_window_pairs(bool fill, EventSoA ev, uint32_t n_events, uint32_t n_win, const int32_t *w_tid, const int32_t *w_beg, const int32_t *w_end,
                         const uint32_t *max_span, uint32_t *w_lo, uint32_t *w_hi /* the windows' candidate ranges: written by the count pass (fill = false), read by the fill pass */,
                         uint32_t *count, const uint32_t *base, uint32_t *pair_ev, uint32_t *pair_win, hipStream_t stream);
// associate: (window, junction) pairs; junction arrays are bucketed by contig, chrom_off has n_chrom + 1 entries
void launch_assoc_pairs(bool fill, uint32_t n_win, const int32_t *w_chrom, const uint32_t *w_ces, const uint32_t *w_cee, const uint32_t *chrom_off,
                        const uint32_t *j_start, const uint32_t *j_end, uint32_t *count, const uint32_t *base, uint32_t *pair_j, uint32_t *pair_win, hipStream_t stream);
void launch_pair_gather(EventSoA ev, const uint32_t *pair_ev, const uint32_t *pair_win, uint32_t n, EventSoA out, hipStream_t stream);

// ---- multi-GPU table merge on the device (merge_kernels.hip) ----------------------------------------------------------------
struct MergeSoA { uint32_t *tid, *start, *end, *ts, *te, *count, *cls, *first, *shard; uint32_t *strand; };
struct MergeUnique { uint32_t *tid, *start, *end, *ts, *te, *count, *first, *last_shard, *strand; };
void launch_merge_unpack(const uint32_t *rows, uint32_t stride_rows, uint32_t n_parts, const uint32_t *part_rows, const uint32_t *part_base, MergeSoA m, hipStream_t st);
void launch_cols_to_packed(const uint32_t *cols /* launch_rows_out block */, uint32_t n, uint32_t *out, hipStream_t st);
void launch_merge_heads(MergeSoA m, const uint32_t *sorted, uint32_t n, uint32_t *head, hipStream_t st);
void launch_merge_reduce(MergeSoA m, const uint32_t *sorted, const uint32_t *head, const uint32_t *seg_excl, uint32_t n, MergeUnique u /* ts preset to ~0, rest 0 */, hipStream_t st);
void launch_merge_rank(const uint32_t *by_first, uint32_t n, uint32_t *name_rank, hipStream_t st);
void launch_merge_pack(MergeUnique u, const uint32_t *order, const uint32_t *name_rank, uint32_t n, uint32_t *out, hipStream_t st);
// the merged rows in the result table's host block layout (see launch_rows_table)
void launch_merge_table(MergeUnique u, const uint32_t *order, const uint32_t *name_rank, uint32_t n, uint32_t min_anchor, uint8_t *out, hipStream_t st);

}  // namespace rgx
