// host_io.h -- host-side container parsing around the device pipeline: BGZF member chain, BAI, BAM header,
// region strings.  No zlib anywhere: bytes are only *located* here, never inflated.
// file:line citations are relative to /root/reference/src/utils/htslib.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "common.h"

namespace rgx {

struct HostMember { uint64_t coff; uint32_t blen; uint32_t isize; };

// bgzf.c:348-355 check_header + :525 block_length. Walks the BSIZE chain from offset 0 and stops at the first
// malformed member (the reference would fail to read that block, ending iteration).
void walk_members(const uint8_t *bam, size_t len, std::vector<HostMember> &out);

// Host threads worth starting for a short parallel phase: the hardware's count, capped by the CPU quota of the control group the process runs
// in (a container with cpu.max = 16 CPUs on a 256-thread host is throttled for most of a period when 48 threads run at once) and by `cap`.
unsigned usable_threads(unsigned cap);

// The member list of a WELL-FORMED file, found by several host threads at once (the upload of the file to the device runs meanwhile):
// chain k starts at the first BGZF header at or after byte k * len / K and walks the BSIZE chain (bgzf.c:525), every thread several chains in
// turn (memory-level parallelism without more threads); every walk must end exactly where the next one started and the last one at the end of
// the file, every member must carry the header of bgzf.c:348-355, a
// BSIZE of at least 26 and an ISIZE of at most 64 KiB.  Anything else -- false = the caller falls back to the device's member discovery,
// which knows what the reference does with damaged files.  Members come out as the device kernels build them (k_member_compact).
bool scan_members_parallel(const uint8_t *bam, size_t len, int threads, std::vector<Member> &out, uint64_t &total_inflated);

// hts.c:1517-1567 (BAI loader), :1092 META_BIN, :1721-1731 HTS_IDX_START
struct BaiInfo {
    int32_t  n_ref = 0;
    bool     have_start = false;
    uint64_t start_voff = 0;
    uint64_t n_no_coor = 0;
    bool     have_nocoor = false;    // the LAST reference has the pseudo-bin: region "*" starts at its end offset (hts.c:1733-1741)
    uint64_t nocoor_voff = 0;
    std::vector<uint64_t> anchors;   // sorted unique virtual offsets that are record starts (linear index + chunk begins)
};
bool parse_bai(const uint8_t *bai, size_t len, BaiInfo &out, bool collect_anchors = true);
// for each target virtual offset: the smallest record-start offset listed in the index that is >= target
// (UINT64_MAX when none); one linear pass, no sorting.
void bai_first_anchor_ge(const uint8_t *bai, size_t len, const uint64_t *targets, int n, uint64_t *out);
// The chunks the reference's iterator over [beg, end) of reference tid reads, in order (hts_itr_query, hts.c:1733-1800): virtual-offset
// pairs, sorted, merged exactly as upstream merges them.  false = the index cannot answer (tid beyond it, damaged); an empty list =
// an iterator that returns nothing.
struct VChunk { uint64_t u, v; };
bool region_chunks(const uint8_t *bai, size_t len, int32_t tid, int32_t beg, int32_t end, std::vector<VChunk> &out);
// [first chunk begin, largest chunk end) of region_chunks (tests; false = nothing to read)
bool bai_region_span(const uint8_t *bai, size_t len, int32_t tid, int32_t beg, int32_t end, uint64_t &lo, uint64_t &hi, bool &usable);
constexpr uint32_t kCsiBin = 0xfffffffeu;   // bin number normalize_index gives the real bins of a converted .csi
constexpr uint32_t kCsiMeta = 0xffffffffu;  // the pseudo-bin in the block of real bin numbers behind such an image
// The BAM header from the head of the file, inflated on the host (the product's own decoder): region queries need the contig names
// BEFORE the device launch to turn the region into a member range.  false = not available this way (the device path will say why).
struct BamHeader;
bool host_bam_header(const uint8_t *bam_head, size_t len, BamHeader &h, size_t *consumed = nullptr /* compressed bytes of the members the header spans */,
                     uint32_t *mean_record_bytes = nullptr /* of the first records behind the header (up to two more members are inflated for it); 0 = none seen */);

// hts.c:2009-2042 index file name resolution: "<fn>.csi", "<fn minus extension>.csi", then the same two for ".bai".
// returns 0 found, 1 none
int find_index(const std::string &bam_path, std::string &out);

// "<fn>.tbi", then "<fn minus extension>.tbi" (tbx_index_load -> hts_idx_getfn)
bool find_tbi(const std::string &path, std::string &out);

bool read_file(const std::string &path, std::vector<uint8_t> &out);

// gzip / BGZF bytes -> plain bytes, with the product's own decoder compiled for the host (no zlib).  "" on success.
std::string gunzip_all(const uint8_t *d, size_t n, std::string &out);

// hts_idx_load_local (hts.c:1569-1618) reads an index through bgzf_open, so .bai and .csi may both be BGZF-compressed, and a .csi
// (any min_shift / depth) is as good as a .bai.  This path only needs what parse_bai / bai_first_anchor_ge read -- the pseudo-bin with
// the first offset, every chunk begin and lower bound (record starts), n_no_coor -- so a CSI is rewritten as an equivalent BAI image.
// out/out_len point into `in` (already a plain BAI) or into `storage`.  false = not an index this path understands.
// read_file + normalize_index: `out` holds a plain BAI image afterwards.
bool read_index(const std::string &path, std::vector<uint8_t> &out);
bool normalize_index(const uint8_t *in, size_t n, std::vector<uint8_t> &storage, const uint8_t *&out, size_t &out_len);

// The bytes of a (large) input file without a private copy: a read-only mapping for regular files -- the upload to the device reads
// straight from the page cache -- and a plain read for everything else (pipes, /dev/stdin).
struct FileBytes {
    const uint8_t *p = nullptr; size_t n = 0;
    bool mapped = false;
    std::vector<uint8_t> own;
    FileBytes() = default;
    FileBytes(const FileBytes &) = delete;
    FileBytes &operator=(const FileBytes &) = delete;
    ~FileBytes();
    void release();
    void release_later();      // the unmapping on the process's background thread (worker_pool.h Reaper): the caller's call ends without it
    // populate: fault the whole mapping in at once (text files that several threads are about to scan)
    bool open(const std::string &path, bool populate = false);
    const uint8_t *data() const { return p; }
    size_t size() const { return n; }
};

// sam.c:114-223 bam_hdr_read on the first bytes of the inflated stream.
// returns 0 ok, 1 need more bytes (need set), 2 bad magic
struct BamHeader { std::vector<std::string> names; std::vector<uint32_t> lens; uint64_t end = 0; };
int parse_bam_header(const uint8_t *d, uint64_t have, BamHeader &h, uint64_t &need);

// hts.c:1834-1922 hts_parse_decimal / hts_parse_reg / hts_itr_querys (+ sam.c:262-277 bam_name2id, last duplicate wins).
// returns false when the reference's iterator would be NULL.
// say: hts_parse_decimal's two warnings go to stderr (one call per query upstream: the caller that stands for it passes true)
bool parse_region(const BamHeader &h, const char *reg, int32_t &tid, int32_t &beg, int32_t &end, bool say = false);
std::string bam_open_notes(const uint8_t *bam, size_t len, const char *bam_path, const char *index_path);

}  // namespace rgx
