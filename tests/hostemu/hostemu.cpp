// tests/hostemu/hostemu.cpp -- TEST INFRASTRUCTURE ONLY.
// Compiles the per-lane device cores (inflate_core.h, bam_core.h) for the host so their logic can be
// unit-tested on a machine without a GPU.  Never linked into the product library.
#include "../../regtools_amd/csrc/inflate_core.h"

extern "C" int emu_inflate(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t cap, uint32_t *out_len) {
    rgx::HostTab T;
    return rgx::inflate_raw(in, in_len, out, cap, out_len, T);
}

// ---- the per-alignment cores of bam_core.h / cse_core.h, as the kernels call them ---------------------------------------------
#include "../../regtools_amd/csrc/bam_core.h"
#include "../../regtools_amd/csrc/cse_core.h"

struct emu_candidate { uint32_t start, end, thick_start, thick_end; };

extern "C" int emu_cigar_walk(int32_t pos, const uint32_t *cigar, uint32_t n_cigar, emu_candidate *out, int max_out) {
    int n = 0;
    rgx::cigar_walk(pos, reinterpret_cast<const uint8_t *>(cigar), n_cigar, [&](uint32_t s, uint32_t e, uint32_t ts, uint32_t te) {
        if (n < max_out) out[n] = emu_candidate{s, e, ts, te};
        ++n;
    });
    return n;
}
extern "C" int emu_strand_from_flag(uint32_t flag, int strandness) { return rgx::strand_from_flag(flag, strandness); }
extern "C" int emu_strand_from_tag(const uint8_t *aux, uint32_t len, uint8_t t0, uint8_t t1) { return rgx::strand_from_tag(aux, aux + len, t0, t1); }
extern "C" uint32_t emu_ucsc_bin(uint32_t start, uint32_t end) { return rgx::ucsc_bin(start, end); }
extern "C" int32_t emu_rec_endpos(const uint32_t *cigar, uint32_t n_cigar, uint32_t flag, int32_t pos) {
    return rgx::rec_endpos(reinterpret_cast<const uint8_t *>(cigar), n_cigar, flag, pos);
}

// ---- index normalisation (host_io.cpp): BAI / CSI / BGZF-compressed -> what the pipeline reads from it ---------------------------------
#include "../../regtools_amd/csrc/host_io.h"
// out[0..4] = n_ref, have_start, start_voff, n_no_coor, n_anchors; got[k] = first listed record start >= targets[k].  0 = not an index
extern "C" int emu_index_summary(const uint8_t *idx, size_t n, uint64_t *out, const uint64_t *targets, int nt, uint64_t *got) {
    std::vector<uint8_t> image; const uint8_t *d; size_t len;
    if (!rgx::normalize_index(idx, n, image, d, len)) return 0;
    rgx::BaiInfo bi;
    if (!rgx::parse_bai(d, len, bi, true)) return 0;
    out[0] = (uint64_t)bi.n_ref; out[1] = bi.have_start; out[2] = bi.start_voff; out[3] = bi.n_no_coor; out[4] = bi.anchors.size();
    if (nt) rgx::bai_first_anchor_ge(d, len, targets, nt, got);
    return 1;
}

// ---- the REAL std::unordered_map<std::string,int> of this image's libstdc++: what oracle.c's restated container is pinned against -----
#include <string>
#include <unordered_map>
// insert keys[0..n) the way Junction::barcodes receives them (a repeated key increments); order[k] = index of the first occurrence of the
// k-th key in iteration order, counts[k] its count, buckets = final bucket_count; returns the number of distinct keys
extern "C" size_t emu_umap_order(const char *const *keys, size_t n, size_t *order, int *counts, size_t *buckets) {
    std::unordered_map<std::string, int> m, first;
    for (size_t i = 0; i < n; ++i) {
        auto it = m.find(keys[i]);
        if (it != m.end()) { std::unordered_map<std::string, int> c = m; c[it->first]++; m = c; }      // the reference's copy-then-bump (cc:207-210)
        else { std::unordered_map<std::string, int> c = m; c.insert(std::pair<std::string, int>(keys[i], 1)); m = c; first[keys[i]] = (int)i; }
    }
    size_t k = 0;
    for (auto it = m.begin(); it != m.end(); ++it, ++k) { order[k] = (size_t)first[it->first]; counts[k] = it->second; }
    *buckets = m.bucket_count();
    return k;
}

// region -> [lo, hi) virtual offsets from a BAI (host_io.cpp bai_region_span); returns 1 span, 0 nothing to read, -1 not usable
extern "C" int emu_region_span(const uint8_t *bai, size_t n, int32_t tid, int32_t beg, int32_t end, uint64_t *lo, uint64_t *hi) {
    std::vector<uint8_t> image; const uint8_t *d; size_t len;
    if (!rgx::normalize_index(bai, n, image, d, len)) return -2;          // as prepare_events sees it: .bai, .csi or BGZF-compressed
    bool usable = false;
    if (rgx::bai_region_span(d, len, tid, beg, end, *lo, *hi, usable)) return 1;
    return usable ? 0 : -1;
}
// the BAM header through the host decoder (host_io.cpp host_bam_header): number of contigs, or -1
extern "C" int emu_host_header(const uint8_t *bam, size_t n, char *first_name, size_t cap) {
    rgx::BamHeader h;
    if (!rgx::host_bam_header(bam, n, h)) return -1;
    if (!h.names.empty() && cap) { strncpy(first_name, h.names[0].c_str(), cap - 1); first_name[cap - 1] = 0; }
    return (int)h.names.size();
}
