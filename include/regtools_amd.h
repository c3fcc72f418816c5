/* include/regtools_amd.h -- C ABI of libregtools_amd.so, the MI355X (gfx950) drop-in for the
 * `regtools junctions extract` hot path.
 *
 * The reference has no FFI layer; its seam is the public interface of class JunctionsExtractor
 * (/root/reference/src/junctions/junctions_extractor.h:183-247).  Each entry point below names the
 * reference interface it replaces.  Pure C: plain pointers and sizes, no C++ or torch types.
 * INTEGRATION.md shows the binding a regtools maintainer would add.
 *
 * Error model (replaces `throw std::runtime_error(msg)` caught in junctions_main.cc:51-57): every call
 * returns 0 on success; otherwise nonzero with the reference's own message text in `err`.
 * There is NO CPU fallback: without a HIP device (or if the gfx950 code object cannot be loaded)
 * rgx_ctx_create fails loudly with RGX_ERR_NO_DEVICE.
 */
#ifndef REGTOOLS_AMD_H
#define REGTOOLS_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RGX_OK             0
#define RGX_ERR_OPEN       1  /* "Unable to open BAM/SAM file.\n\n"                      junctions_extractor.cc:505 */
#define RGX_ERR_INDEX      2  /* "Unable to open BAM/SAM index. Make sure ...\n\n"       junctions_extractor.cc:510 */
#define RGX_ERR_REGION     3  /* "Unable to iterate to region within BAM.\n\n"           junctions_extractor.cc:521 */
#define RGX_ERR_NO_DEVICE  4  /* no usable HIP device / code object: the product never falls back to a CPU */
#define RGX_ERR_DEVICE     5  /* HIP runtime error */
#define RGX_ERR_FORMAT     6  /* malformed BGZF/BAM beyond what the reference tolerates silently */
#define RGX_ERR_ARG        7
#define RGX_ERR_FASTA      8  /* "Unable to extract FASTA sequence for position ...\n\n" junctions_extractor.cc:553 */
#define RGX_ERR_ABORT      9  /* the reference abort()s on this input (`junctions extract -s XS`, and `identify -s XS` in the first variant's window that reads it: an
                               * aux field of unknown type in front of the strand tag of a spliced read, sam.c:1233-1252 skip_aux; a VCF record whose FORMAT
                               * names a Flag, vcf.c:1638-1639); the tool then prints what htslib printed and calls abort() itself, a library caller gets this code */
#define RGX_ERR_EXIT      10  /* the reference's LIBRARY ends the process with exit(1) on this input, past the tool's own error handling (a VCF sample with more
                               * fields than FORMAT has keys, vcf.c:1610-1614): err is all it prints; the tool prints it and exits with 1 */

typedef struct rgx_ctx rgx_ctx;   /* one per process+device: HIP stream(s) and a reusable HBM workspace */

/* Parameters of one extraction.
 * Replaces the JunctionsExtractor constructors (junctions_extractor.h:185-205) and the option parser
 * (junctions_extractor.cc:42-122).  rgx_extract_params_default fills the default-ctor values. */
typedef struct {
    const char *region;        /* "." = whole file (h:196); "chr:beg-end" as sam_itr_querys parses it */
    int32_t     strandness;    /* 0 XS tag, 1 RF, 2 FR, 3 intron-motif (cc:71-84) */
    char        strand_tag[2]; /* "XS" (h:192) */
    uint32_t    min_anchor;    /* 8   (h:186)  -a */
    uint32_t    min_intron;    /* 70  (h:187)  -m */
    uint32_t    max_intron;    /* 500000 (h:188) -M */
    const char *fasta_path;    /* NULL = "NA"; required by strandness 3 (cc:105-110) */
    /* shard of the BGZF member list handled by this call (multi-GPU): members are cut into
     * n_shards contiguous ranges balanced by compressed bytes; records belong to the shard their first
     * byte is in.  0/1 = everything. */
    int32_t     shard;
    int32_t     n_shards;
    /* -b (cc:82-84): also count, per junction, the cell barcodes of its supporting reads (set_junction_barcode, cc:362-374).
     * Needs the whole file in one shard. */
    int32_t     barcodes;      /* 0 */
    char        barcode_tag[2];/* "CB" (h:192, h:204) */
} rgx_extract_params;

void rgx_extract_params_default(rgx_extract_params *p);

/* Result table, structure-of-arrays, rows in the reference's output order
 * (compare_junctions, junctions_extractor.h:117-140).  Replaces vector<Junction> from
 * JunctionsExtractor::get_all_junctions (cc:238-246); `left_ok && right_ok` is the filter
 * print_all_junctions applies (cc:267).  Owned by the library; release with rgx_table_free. */
typedef struct {
    int32_t    n_ref;
    char     **ref_name;        /* header->target_name */
    uint32_t  *ref_len;
    uint64_t   n;               /* rows */
    int32_t   *tid;
    uint32_t  *start, *end;     /* Junction::start/end (BED::start/end) */
    uint32_t  *thick_start, *thick_end;
    uint32_t  *read_count;
    uint64_t  *name_index;      /* k of "JUNC%08d" (get_new_junction_name, cc:152-157) */
    char      *strand;
    uint8_t   *left_ok, *right_ok;
    /* statistics (not part of the reference interface) */
    uint64_t   n_records;       /* alignments iterated in this shard */
    uint64_t   n_events;        /* junction events that passed junction_qc */
    uint64_t   inflated_bytes, compressed_bytes, n_members;
    double     ms_total, ms_inflate, ms_records, ms_scan, ms_reduce; /* wall + HIP-event stage times */
    /* partial (per-shard) tables also carry what a cross-shard merge needs */
    uint64_t  *first_seen;      /* event order of the first read of each row (shard-local) */
    uint64_t  *last_seen;       /* event order of the last read (its strand is the row's strand) */
    uint64_t   framing_sweeps;  /* statistics: verification sweeps of the speculative record framing (1 = every guess was right) */
    /* -b: Junction::barcodes (junctions_extractor.h:58) of every row, flattened.  Row i owns entries [bc_row_begin[i], bc_row_begin[i+1]),
     * listed in the order Junction::print_barcodes (h:99-111) writes them, i.e. the iteration order of the std::unordered_map the
     * reference keeps; entry k is the string bc_text[bc_str_begin[k] .. bc_str_begin[k+1]) seen bc_count[k] times.  All NULL unless
     * rgx_extract_params.barcodes was set. */
    uint64_t  *bc_row_begin;    /* n + 1 */
    uint32_t  *bc_count;
    uint64_t  *bc_str_begin;    /* entries + 1 */
    char      *bc_text;
    uint32_t  *bc_insert_rank;  /* entry k was the bc_insert_rank[k]-th distinct barcode its junction saw (0-based): a binding that refills a
                                 * Junction::barcodes inserts a row's entries in THIS order and the container iterates them in the listed order */
    double     ms_barcodes;     /* statistics: the barcode group-by (device + host ordering) */
    /* per-shard tables: nonzero when the record stream stopped inside this shard for a reason that ends iteration upstream (a member
     * that does not inflate, an empty member, an unreadable record) instead of reaching the shard's upper cut.  The merges ignore the
     * shards behind such a one: a later shard is a seek past the damage, a sequential reader never gets there. */
    uint64_t   stream_ended;
    double     ms_inflate_launch;   /* statistics: the whole-range DEFLATE launch alone, HIP events on the stream it ran on (0 = the range went up as several
                                     * launches).  Host input: the arrival-gated launch -- it spans the upload, its waves wait for their chunk; ms_inflate
                                     * is the pipeline stream's stage time, which with the early tail no longer covers that launch */
} rgx_junction_table;

int  rgx_ctx_create(int device, rgx_ctx **out, char *err, size_t errlen);
void rgx_ctx_destroy(rgx_ctx *ctx);
/* Statistics: the DEFLATE launch's time depends on where the arena's pages lie (12.8 / 13.9 / 15.0 ms for the same launch into ten arenas of one
 * process), so a context tries a few allocations on its first large call and keeps the fastest (REGTOOLS_AMD_ARENA=n, 0 = off; never in a one-shot
 * context).  ms[0] = the call's own arena, ms[1..] = the challengers; returns how many were timed by the last calibration (0 = none yet). */
int  rgx_ctx_arena_trials(const rgx_ctx *ctx, float *ms, int cap);

/* Replaces JunctionsExtractor::identify_junctions_from_BAM + get_all_junctions (cc:500-535, 238-246):
 * reads <bam_path> and its index (.csi before .bai, plain or BGZF-compressed: hts.c:2031-2042) from disk, uploads, runs the device pipeline, returns the table. */
int  rgx_extract(rgx_ctx *ctx, const char *bam_path, const rgx_extract_params *p,
                 rgx_junction_table **out, char *err, size_t errlen);

/* Same, input already in host memory (file bytes of the .bam and of its .bai or .csi) -- the configuration SURVEY.md 8(d) times:
 * "file bytes in host memory" -> "sorted junction table in host memory".  The file is uploaded in chunks on a copy stream while the
 * BGZF members of the chunks that have arrived are already being inflated; for that overlap the bytes should sit in page-locked memory
 * (rgx_host_alloc, or any hipHostMalloc / pinned allocation) -- pageable memory works, the copies then go through the driver's staging
 * buffer at a fraction of the link rate.  Replaces the file read of junctions_extractor.cc:503-525 (hts_open / sam_itr_querys). */
void *rgx_host_alloc(size_t bytes);      /* page-locked host memory (NULL when there is none to be had) */
void  rgx_host_free(void *p);
int  rgx_extract_mem(rgx_ctx *ctx, const void *bam, size_t bam_len, const void *bai, size_t bai_len,
                     const rgx_extract_params *p, rgx_junction_table **out, char *err, size_t errlen);

/* Several files in flight on one device (round 6).  Replaces the loop a cohort run makes around `regtools junctions extract` -- one process per BAM, one
 * after the other (junctions_main.cc:45-59).  A pipeline owns `depth` contexts on `device`; rgx_extract_submit hands file k to context k mod depth and
 * returns at once with a ticket, rgx_extract_wait returns that file's table (or its error) -- each file is one ordinary rgx_extract_mem call, so the
 * tables are those of sequential calls, while file k+1's upload and inflate run under file k's tail.  `bam` / `bai` must stay readable until the
 * file's rgx_extract_wait returns (page-locked memory: rgx_host_alloc); the parameter struct and its strings are copied by submit.  Tickets may be
 * waited for in any order, each once.  rgx_pipeline_destroy runs what is still queued to its end and frees tables nobody waited for.
 * Hardware queues: the HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES queues (4 unless the environment says otherwise WHEN HIP STARTS).
 * A pipeline works on four; with 16 or more the files' DEFLATE launches go out at once instead of in turns and a file costs ~7 % less;
 * depth > 2 is refused below 4 x depth. */
typedef struct rgx_pipeline rgx_pipeline;
int  rgx_pipeline_create(int device, int depth /* 1..8, 2 = one file's tail under the next one's upload */, rgx_pipeline **out, char *err, size_t errlen);
int  rgx_pipeline_depth(const rgx_pipeline *pl);
/* The context file `ticket` runs on.  After rgx_extract_wait its rows are still in that context's HBM (rgx_last_table_pack_device) until the file `depth`
 * tickets later is submitted: a rank of a multi-GPU job merges file k with the other ranks' there while file k+1 is already going up. */
rgx_ctx *rgx_pipeline_ctx(const rgx_pipeline *pl, uint64_t ticket);
int  rgx_extract_submit(rgx_pipeline *pl, const void *bam, size_t bam_len, const void *bai, size_t bai_len, const rgx_extract_params *p,
                        uint64_t *ticket, char *err, size_t errlen);
int  rgx_extract_wait(rgx_pipeline *pl, uint64_t ticket, rgx_junction_table **out, char *err, size_t errlen);
void rgx_pipeline_destroy(rgx_pipeline *pl);

/* Same, with the .bam bytes ALREADY RESIDENT IN HBM at d_bam (the measured configuration: bench.py, or a
 * pipeline that DMA'd the file straight to the device).  No host copy of the BAM is needed: even the BGZF
 * member chain (BSIZE at +16 of every member, bgzf.c:525) is discovered on the device.  Only the (small) .bai is
 * read on the host.  d_bam must be readable up to bam_len + 8 bytes. */
int  rgx_extract_device(rgx_ctx *ctx, const void *d_bam, size_t bam_len,
                        const void *bai, size_t bai_len, const rgx_extract_params *p,
                        rgx_junction_table **out, char *err, size_t errlen);

/* The same over several GPUs of one node from ONE host process: shard g of n_devices -- a contiguous BGZF member range
 * cut at record starts the index lists -- runs the whole pipeline on devices[g] from its own host thread; the shards' unique rows
 * (48 bytes each, packed in HBM on the shard's own stream) are gathered to devices[0] with ncclSend / ncclRecv in one group (RCCL over xGMI;
 * librccl.so.1 is loaded at run time; peer access is switched on per device pair) and merged there (rgx_table_merge_device).  Where RCCL cannot
 * be loaded, initialised or reports an error, the same rows travel as hipMemcpyPeerAsync copies and rgx_multi_exchange_kind() says so.  Shard order is file order: the table is the single-GPU table whatever n_devices is.  A device
 * may be listed more than once: those shards take turns on it and the exchange is a device copy (how a one-GPU box tests this path).
 * Replaces the call junctions_extract() makes into JunctionsExtractor (junctions_main.cc:45-59) on a multi-GPU node.  -b works across shards
 * (rgx_table_merge_barcodes).  The per-device contexts (workspace, streams) are created on first use and kept for the life of the process;
 * concurrent calls take turns. */
int  rgx_extract_multi(const int *devices, int n_devices, const char *bam_path, const rgx_extract_params *p,
                       rgx_junction_table **out, char *err, size_t errlen);
int  rgx_extract_multi_mem(const int *devices, int n_devices, const void *bam, size_t bam_len, const void *bai, size_t bai_len,
                           const rgx_extract_params *p, rgx_junction_table **out, char *err, size_t errlen);
/* How the last rgx_extract_multi* call of this process moved the shards' rows to the first device: "rccl grouped send/recv, N ranks, ...",
 * "hipMemcpyPeerAsync ... (RCCL not used: why)", "device copies (a device is listed more than once)", "none (one shard)".  The reference has
 * no counterpart (junctions_main.cc:45-59 is one process on one core); callers that report a multi-GPU run quote it (bench.py multi_gpu.exchange). */
const char *rgx_multi_exchange_kind(void);

void rgx_table_free(rgx_junction_table *t);

/* Merge per-shard tables (shard order = file order) into the final table: sum counts, min/max thick
 * bounds, earliest first_seen names the row, latest last_seen gives the strand; rows are renamed and
 * re-sorted.  This is the host-side half of the multi-GPU path; the device-side exchange is an
 * all-gather of the packed rows (see rgx_table_pack / rgx_table_unpack). */
int  rgx_table_merge(const rgx_junction_table *const *parts, int n_parts, uint32_t min_anchor,
                     rgx_junction_table **out, char *err, size_t errlen);

/* Fixed-width row packing for the RCCL all-gather: 48 bytes per row. */
#define RGX_PACKED_ROW_BYTES 48
size_t rgx_table_pack(const rgx_junction_table *t, void *dst, size_t dst_cap); /* returns bytes needed */
int    rgx_table_unpack(const void *src, size_t n_rows, const rgx_junction_table *names_from,
                        rgx_junction_table **out);
/* The barcode lists of a table as one byte block, and back onto a table of the same rows (rgx_table_unpack): what the ranks of the
 * one-process-per-GPU driver exchange next to the packed rows so that -b works there too (junctions_extractor.cc:204-217).  pack returns the bytes
 * needed (0 = no barcode lists) and writes when dst_cap suffices; unpack validates every offset. */
size_t rgx_table_pack_barcodes(const rgx_junction_table *t, void *dst, size_t dst_cap);
int    rgx_table_unpack_barcodes(rgx_junction_table *t, const void *src, size_t len);
/* -b across shards: fills merged->bc_* from the shards' tables (every one extracted with barcodes = 1; shard order = file order).  A junction's
 * barcodes are the shards' lists one after the other in first-seen order, equal strings summed, handed to the container the reference keeps
 * (junctions_extractor.cc:204-217, h:99-111).  rgx_table_merge and rgx_extract_multi call it themselves. */
int    rgx_table_merge_barcodes(const rgx_junction_table *const *parts, int n_parts, rgx_junction_table *merged, char *err, size_t errlen);
/* rgx_table_pack without the host: t must be the result of the LAST rgx_extract / rgx_extract_mem / rgx_extract_device call on this
 * context (its rows are then still in HBM; anything else is RGX_ERR_ARG and the caller packs on the host).  Writes t->n packed rows
 * to device memory at d_dst -- the all-gather input of the multi-GPU path. */
int    rgx_last_table_pack_device(rgx_ctx *ctx, const rgx_junction_table *t, void *d_dst, uint64_t cap_rows, char *err, size_t errlen);
/* The same merge with the gathered rows STILL IN HBM (what an RCCL all-gather leaves there): shard g's packed rows start at
 * d_rows + g * stride_rows * 48 bytes and there are part_rows[g] of them.  Sort, reduce, naming and output order run on the
 * device; one copy of the final rows comes back.  first_seen/last_seen of the result are the merge's own order words. */
int    rgx_table_merge_device(rgx_ctx *ctx, const void *d_rows, uint64_t stride_rows, const uint64_t *part_rows, int n_parts,
                              uint32_t min_anchor, const rgx_junction_table *names_from, rgx_junction_table **out,
                              char *err, size_t errlen);

/* Replaces Junction::print / print_all_junctions (junctions_extractor.h:90-98, cc:249-280): BED12 text.
 * only_anchored != 0 keeps rows with both anchors (the `junctions extract` output).  Returns the number
 * of bytes written, or the size needed when buf is NULL. */
size_t rgx_table_format_bed12(const rgx_junction_table *t, int only_anchored, char *buf, size_t cap);

/* Replaces Junction::print_barcodes as print_all_junctions calls it (h:99-111, cc:272-273): one line "<distinct>\t<bc>:<count>,...\n"
 * per printed row.  Same buffer protocol as rgx_table_format_bed12.  A table without barcodes gives "0\t\n" lines (what
 * `cis-splice-effects identify -b` writes: its extractor never collects any, identifier.cc:288, :239-241). */
size_t rgx_table_format_barcodes(const rgx_junction_table *t, int only_anchored, char *buf, size_t cap);

/* Library/build identification: "regtools_amd <version> gfx950". */
const char *rgx_version(void);

/* Stage-level kernel entry points on caller-provided DEVICE buffers (used by tests/bench to measure the
 * dominant kernel in isolation; all asynchronous on `stream`, a hipStream_t passed as void*). */
typedef struct { uint64_t cpos, upos; uint32_t clen, isize; } rgx_member;   /* == rgx::Member */
int  rgx_k_inflate(const void *d_comp, const rgx_member *d_members, uint32_t n_members,
                   void *d_arena, uint32_t *d_status, void *stream);
/* The same with the form of the decoder named: 0 = the pipeline's choice (one member per WAVE, whole member in LDS, up to 2048 members -- a
 * member in ~1.5 ms; above that one member per LANE -- ~8 ms per launch whatever its size, 196,608 members at a time; REGTOOLS_AMD_INFLATE=
 * lane|wave|ring|coop overrides), 1 = lane (k_inflate), 2 = wave (k_inflate_wave), 3 = lane with an LDS window and whole-line output
 * (k_inflate_ring; needs 16 readable bytes in front of d_arena), 4 = lane with the long matches copied by the wave (k_inflate_coop: what the
 * pipeline runs above 2048 members; the members of a launch must lie in the arena in the order of the list, a group of 1024 of them within
 * 4 GiB -- a list that does not is refused by forms 0 and 4: d_status[0] = the first offending member, d_status[1] = 10, nothing written),
 * 5 = lane taking up to four literals per trip (k_inflate<.., 4>). */
int  rgx_k_inflate_form(int form, const void *d_comp, const rgx_member *d_members, uint32_t n_members,
                        void *d_arena, uint32_t *d_status, void *stream);

/* =====================================================================================================
 * `cis-splice-effects identify`.
 * Replaces CisSpliceEffectsIdentifier::identify() + annotate_junctions()
 * (src/cis-splice-effects/cis_splice_effects_identifier.cc:222-312) and the interval cores it drives:
 * VariantsAnnotator::annotate_record_with_transcripts (src/variants/variants_annotator.cc:455-518),
 * the per-variant JunctionsExtractor re-runs (identifier.cc:288-299) and
 * JunctionsAnnotator::annotate_junction_with_gtf (src/junctions/junctions_annotator.cc:344-363).
 * ===================================================================================================== */
typedef struct {
    const char *vcf_path, *bam_path, *fasta_path, *gtf_path;   /* the four positional arguments (identifier.cc:193-198) */
    const char *out_tsv;       /* -o  annotated junctions; NULL = stdout */
    const char *out_vcf;       /* -v  splice-relevant variants (VCF); NULL = not written */
    const char *out_bed;       /* -j  junctions BED12; NULL = not written */
    uint32_t    window;        /* -w  [0 = the cis-effect window between neighbouring exons] */
    uint32_t    intronic_min;  /* -i  [2] */
    uint32_t    exonic_min;    /* -e  [3] */
    int32_t     all_intronic;  /* -I */
    int32_t     all_exonic;    /* -E */
    int32_t     skip_single;   /* 1 unless -S */
    int32_t     strandness;    /* -s  0 XS, 1 RF, 2 FR, 3 intron-motif */
    char        strand_tag[2]; /* -t */
    uint32_t    min_anchor;    /* -a  (also the minimum intron length here: ctor quirk junctions_extractor.h:200) */
    uint32_t    min_intron;    /* -m  accepted and ignored, as upstream */
    uint32_t    max_intron;    /* -M */
    int32_t     override_motif;/* -C */
    const char *bed_path;      /* rgx_associate only: junctions BED12 (second positional of `cis-splice-effects associate`) */
    int32_t     echo;          /* 1: write to stderr what upstream writes while it works -- "exonic_min_distance_ is 3" (variants_annotator.h:151), and per
                                * splice-relevant variant "Variant <chrom> <start> <end> <score>" + "Variant region is <region>" (identifier.cc:265-277,
                                * associator.cc:243-257); the tool sets it, a library caller normally does not [0] */
} rgx_identify_params;

typedef struct {
    uint64_t n_variants, n_relevant, n_windows, n_pairs, n_window_rows, n_junctions;
    uint64_t n_records, n_events;
    uint64_t exon_visits_variants;   /* E_v summed: exon records of all candidate transcripts (SURVEY 8d algorithmic bytes) */
    uint64_t exon_visits_junctions;  /* E_j summed */
    double   ms_total, ms_gtf, ms_variants, ms_extract, ms_join, ms_annotate, ms_output;
    double   ms_k_variant_scan, ms_k_junction_scan, ms_k_window_pairs;   /* the interval kernels alone (HIP events, both passes of each) */
} rgx_identify_stats;

void rgx_identify_params_default(rgx_identify_params *p);   /* CisSpliceEffectsIdentifier ctor, identifier.h:101-117 */

/* Whole command: reads the four files, runs the interval kernels and the extraction on the device, writes -o/-v/-j.
 * Error texts and the exit-code mapping are the reference's (nonzero return == exit 1). */
int  rgx_identify(rgx_ctx *ctx, const rgx_identify_params *p, rgx_identify_stats *stats, char *err, size_t errlen);
/* The same over several devices: the BAM's extraction is sharded over `devices` the way rgx_extract_multi shards it (contiguous member
 * ranges cut at record starts from the index), the junction events are gathered onto devices[0] in file order with device-to-device copies, and the
 * join, the annotation and the outputs run there.  Contexts come from the process-wide cache rgx_extract_multi uses; a device may be listed more than
 * once.  Outputs are byte-identical to rgx_identify's whatever the list. */
int  rgx_identify_multi(const int *devices, int n_devices, const rgx_identify_params *p, rgx_identify_stats *stats, char *err, size_t errlen);

/* SURVEY 8(f) rows f2/f3 -- the three sibling commands over the same kernels.
 * `cis-splice-effects associate` (CisSpliceEffectsAssociator::associate, cis_splice_effects_associator.cc:234-276): the junctions come
 * from p->bed_path (BED12, e.g. the output of `junctions extract`) instead of a BAM; bam_path/strandness/strand_tag are ignored. */
int  rgx_associate(rgx_ctx *ctx, const rgx_identify_params *p, rgx_identify_stats *stats, char *err, size_t errlen);
/* `variants annotate` (VariantsAnnotator::annotate_vcf, variants_annotator.cc:541-550): EVERY record of p->vcf_path written to
 * p->out_vcf (NULL = stdout) with genes= transcripts= distances= annotations= appended to INFO ("NA" when not splice relevant).
 * Uses vcf_path, gtf_path, out_vcf, intronic_min, exonic_min, all_intronic, all_exonic, skip_single. */
int  rgx_variants_annotate(rgx_ctx *ctx, const rgx_identify_params *p, rgx_identify_stats *stats, char *err, size_t errlen);
/* `junctions annotate` (junctions_main.cc:62-93): BED12 rows -> annotated TSV (out_path NULL = stdout); *n_rows = rows written.
 * bedtools' reader semantics are kept: leading #/track/browser lines are skipped, a later one (or a blank line) ends the input,
 * a malformed line ends the run with the reference's message after the rows before it were written. */
int  rgx_junctions_annotate(rgx_ctx *ctx, const char *bed_path, const char *fasta_path, const char *gtf_path, const char *out_path,
                            uint64_t *n_rows, char *err, size_t errlen);
/* The same with options.  Bit 0 = -S (junctions_annotator.cc:392-393, consumed at :131 / :231): single-exon transcripts take part in the scan (they
 * can make a junction's donor or acceptor known, never skip anything); the reference's one unchecked read on this path (exons[i + 1] behind a
 * transcript's last exon, with or without -S) is "no match" here, as in the default mode.  Bit 1 = echo: "position = <region>" on stderr for each of
 * a junction's two FASTA look-ups, as get_reference_sequence writes it (junctions_annotator.cc:366-370); the tool sets it. */
#define RGX_ANNOTATE_SINGLE_EXON 1
#define RGX_ANNOTATE_ECHO        2
int  rgx_junctions_annotate_opts(rgx_ctx *ctx, const char *bed_path, const char *fasta_path, const char *gtf_path, const char *out_path,
                                 int options, uint64_t *n_rows, char *err, size_t errlen);

/* Stage entry points over a loaded annotation (flat exon/transcript/bin arrays in HBM). */
typedef struct rgx_gtf rgx_gtf;
int  rgx_gtf_load(rgx_ctx *ctx, const char *gtf_path, rgx_gtf **out, char *err, size_t errlen);   /* GtfParser::load, gtf_parser.cc:257-263 */
void rgx_gtf_free(rgx_gtf *g);
int  rgx_gtf_info(const rgx_gtf *g, uint32_t *n_transcripts, uint32_t *n_exons, uint32_t *n_chroms);
int  rgx_gtf_transcript_bin(const rgx_gtf *g, const char *transcript_id, uint32_t *bin);           /* GtfParser::bin_from_transcript */

/* a10: one row per variant (chrom name + 0-based pos).  hit_off has n+1 entries; hits are in the reference's
 * visitation order; annotation codes 1 exonic, 2 intronic, 3 splicing_exonic, 4 splicing_intronic. */
typedef struct {
    uint64_t n; uint32_t *cis_start, *cis_end; uint32_t *hit_off; uint32_t *hit_transcript, *hit_annotation, *hit_distance;
} rgx_variant_hits;
int  rgx_variant_windows(rgx_ctx *ctx, const rgx_gtf *g, uint64_t n, const char *const *chrom, const uint32_t *pos0, uint32_t intronic_min,
                         uint32_t exonic_min, int all_intronic, int all_exonic, int skip_single, rgx_variant_hits **out, char *err, size_t errlen);
void rgx_variant_hits_free(rgx_variant_hits *h);
const char *rgx_gtf_transcript_id(const rgx_gtf *g, uint32_t t);

/* a11: one row per junction (chrom, start, end = Junction.end + 1, strand).  flags bit0 known_donor, bit1 known_acceptor,
 * bit2 known_junction; counts are of UNIQUE skipped elements; transcripts in id order, offsets n+1. */
typedef struct {
    uint64_t n; uint32_t *flags, *n_acceptors_skipped, *n_exons_skipped, *n_donors_skipped; uint32_t *tx_off, *tx;
} rgx_junction_annot;
int  rgx_annotate_junctions(rgx_ctx *ctx, const rgx_gtf *g, uint64_t n, const char *const *chrom, const uint32_t *start, const uint32_t *end1,
                            const char *strand, rgx_junction_annot **out, char *err, size_t errlen);
void rgx_junction_annot_free(rgx_junction_annot *a);

/* a9: the window join on its own.  For every window w (contig name, 0-based half-open [beg, end) as sam_itr_querys leaves a region)
 * the junction table that `junctions extract -r` over that window would produce with the parameters in p (p->region is ignored):
 * only reads with pos < end && endpos > beg count, so read_count / thick bounds / names are window-restricted.  The BAM is inflated
 * and scanned ONCE for all windows.  Rows come window-major in input order, inside a window in get_all_junctions order; name_index
 * restarts at 1 in every window.  A contig that is not in the BAM header is an error (RGX_ERR_REGION), as upstream. */
typedef struct {
    uint64_t n;
    uint32_t *window, *start, *end, *thick_start, *thick_end, *read_count, *name_index;
    char *strand;
} rgx_window_rows;
int  rgx_window_join(rgx_ctx *ctx, const char *bam_path, const rgx_extract_params *p, uint64_t n_windows, const char *const *chrom,
                     const int32_t *beg, const int32_t *end, rgx_window_rows **out, char *err, size_t errlen);
void rgx_window_rows_free(rgx_window_rows *r);

#ifdef __cplusplus
}
#endif
#endif
