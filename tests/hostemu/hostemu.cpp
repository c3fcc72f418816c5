// tests/hostemu/hostemu.cpp -- TEST INFRASTRUCTURE ONLY.
// Compiles the per-lane device cores (inflate_core.h, bam_core.h) for the host so their logic can be
// unit-tested on a machine without a GPU.  Never linked into the product library.
#include "../../regtools_amd/csrc/inflate_core.h"

extern "C" int emu_inflate(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t cap, uint32_t *out_len) {
    rgx::HostTab T;
    return rgx::inflate_raw(in, in_len, out, cap, out_len, T);
}
extern "C" int emu_inflate_lits(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t cap, uint32_t *out_len) {
    rgx::HostTab T;
    return rgx::inflate_raw<4>(in, in_len, out, cap, out_len, T);
}

// the round-2 decoder (inflate_ring.h: per-lane LDS window, cooperative line flush) with a one-lane wave.  The member is inflated into
// a private buffer at destination phase `phase` (address % 128) with guard bytes around it: returns -100 / -101 if a byte in front of /
// behind the member's [0, cap) was written.
#include "../../regtools_amd/csrc/inflate_ring.h"
#include <vector>
extern "C" int emu_inflate_ring(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t cap, uint32_t *out_len, uint32_t phase) {
    std::vector<uint8_t> ibuf((size_t)in_len + 64 + 32, 0);
    memcpy(ibuf.data() + 32, in, in_len);
    std::vector<uint8_t> obuf((size_t)cap + 1024, 0xA5);
    uint8_t *dst = (uint8_t *)(((uintptr_t)obuf.data() + 127) & ~(uintptr_t)127) + 256 + (phase & 127u);
    rgx::HostTab T; rgx::HostRing R; rgx::HostCoop C;
    memset(&R, 0xCC, sizeof R);
    const int st = rgx::inflate_ring(ibuf.data() + 32, in_len, dst, cap, out_len, T, R, C, true);
    for (uint8_t *q = obuf.data(); q < dst; ++q) if (*q != 0xA5) return -100;
    for (uint8_t *q = dst + cap; q < obuf.data() + obuf.size(); ++q) if (*q != 0xA5) return -101;
    memcpy(out, dst, *out_len <= cap ? *out_len : cap);
    return st;
}

// the round-3 decoder (inflate_coop.h: long matches handed to the wave) with a one-lane wave; same guard bytes, every destination phase
#include "../../regtools_amd/csrc/inflate_coop.h"
extern "C" int emu_inflate_coop(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t cap, uint32_t *out_len, uint32_t phase, int pairs) {
    // (the payload at every phase of a 16-byte block too)
    std::vector<uint8_t> ibuf((size_t)in_len + 64 + 64, 0xEE);
    uint8_t *src = (uint8_t *)(((uintptr_t)ibuf.data() + 15) & ~(uintptr_t)15) + 16 + ((phase * 7u) & 15u);
    memcpy(src, in, in_len);
    memset(src + in_len, 0, 24);
    std::vector<uint8_t> obuf((size_t)cap + 1024, 0xA5);
    uint8_t *dst = (uint8_t *)(((uintptr_t)obuf.data() + 127) & ~(uintptr_t)127) + 256 + (phase & 127u);
    rgx::HostTab T; rgx::HostCopy C;
    // bit 0 of `pairs`: a literal and the symbol behind it per trip; bit 1 selects the windowed bit reader; bit 2 (with bit 0): two literals and a
    // match behind them per trip, bits counted exactly (the decoder's mode bit 1); bit 3: runs written from registers (mode bit 2)
    const uint32_t mode = (uint32_t)(pairs & 1) | ((pairs & 4) ? 2u : 0u) | ((pairs & 8) ? 4u : 0u);
    const int st = (pairs & 2) ? rgx::inflate_coop<rgx::BitReaderWin>(src, in_len, dst, cap, out_len, T, C, true, mode)
                               : rgx::inflate_coop<rgx::BitReader>(src, in_len, dst, cap, out_len, T, C, true, mode);
    for (uint8_t *q = obuf.data(); q < dst; ++q) if (*q != 0xA5) return -100;
    for (uint8_t *q = dst + cap; q < obuf.data() + obuf.size(); ++q) if (*q != 0xA5) return -101;
    memcpy(out, dst, *out_len <= cap ? *out_len : cap);
    return st;
}

// the small-input decoder (inflate_wave.h: one member per wave, whole member in LDS) with the host's one-thread wave
#include "../../regtools_amd/csrc/inflate_wave.h"
extern "C" int emu_inflate_wave(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t cap, uint32_t *out_len, uint32_t phase) {
    std::vector<uint8_t> ibuf((size_t)in_len + 64, 0);
    memcpy(ibuf.data(), in, in_len);
    std::vector<uint8_t> obuf((size_t)cap + 1024, 0xA5);
    uint8_t *dst = (uint8_t *)(((uintptr_t)obuf.data() + 127) & ~(uintptr_t)127) + 256 + (phase & 127u);
    static rgx::WaveShared S;
    memset(&S, 0xCC, sizeof S);
    rgx::HostWave W;
    const int st = rgx::inflate_wave(W, S, ibuf.data(), in_len, dst, cap, out_len);
    for (uint8_t *q = obuf.data(); q < dst; ++q) if (*q != 0xA5) return -100;
    for (uint8_t *q = dst + cap; q < obuf.data() + obuf.size(); ++q) if (*q != 0xA5) return -101;
    memcpy(out, dst, *out_len <= cap ? *out_len : cap);
    return st;
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

// the host's parallel BGZF member scan (host_io.cpp scan_members_parallel: what rgx_extract_mem's overlapped upload launches from): number of
// members and total inflated size, or -1 when the file is not one the scan vouches for; members[k] = {cpos, upos, clen, isize} as 4 x u64
extern "C" long emu_scan_members(const uint8_t *bam, size_t n, int threads, uint64_t *members, size_t cap, uint64_t *total) {
    std::vector<rgx::Member> m;
    if (!rgx::scan_members_parallel(bam, n, threads, m, *total)) return -1;
    for (size_t k = 0; k < m.size() && k < cap; ++k) { members[4 * k] = m[k].cpos; members[4 * k + 1] = m[k].upos; members[4 * k + 2] = m[k].clen; members[4 * k + 3] = m[k].isize; }
    return (long)m.size();
}

// the annotated-VCF writer (cse_host.cpp write_annotated_vcf_records over vcf_rewrite.cpp) with "NA" for every record: what
// `variants annotate` writes when no transcript is near any variant.  Returns 0, 1 = load error (message in err), 2 = writer error.
#include "../../regtools_amd/csrc/cse_host.h"
// returns 0, or how the reference's process ends: 1 = through the tool's own error path (message, status 1); 4 / 5 = htslib ends it while the header is read,
// with exit(1) / abort(); 2 / 3 = the same at a record (what was written in front of it is in the file)
extern "C" int emu_vcf_rewrite(const char *in_path, const char *out_path, char *err, size_t errlen) {
    rgx::VcfText vcf;
    std::string e = vcf.load(in_path);
    if (!e.empty()) { snprintf(err, errlen, "%s", e.c_str()); return vcf.death == 2 ? 5 : vcf.death == 1 ? 4 : 1; }
    FILE *f = fopen(out_path, "w");
    if (!f) return 2;
    std::vector<size_t> todo(vcf.recs.size());
    for (size_t i = 0; i < todo.size(); ++i) todo[i] = i;
    e = rgx::write_annotated_vcf_records(f, vcf, todo, [](size_t) { return rgx::VcfAnnot{nullptr, nullptr, nullptr, nullptr}; });
    fclose(f);
    if (!e.empty()) { snprintf(err, errlen, "%s", e.c_str()); return vcf.fatal_aborts ? 3 : 2; }
    return 0;
}

// GtfModel::load on the host (it has no device part): every table, as text, for tests/test_hostemu.py
extern "C" int emu_gtf_dump(const char *gtf_path, const char *out_path, char *err, size_t errlen) {
    rgx::GtfModel m;
    const std::string e = m.load(gtf_path);
    if (!e.empty()) { snprintf(err, errlen, "%s", e.c_str()); return 1; }
    FILE *f = fopen(out_path, "w");
    if (!f) return 2;
    for (size_t i = 0; i < m.chroms.size(); ++i) fprintf(f, "chrom %zu %s\n", i, m.chroms[i].c_str());
    for (size_t t = 0; t < m.tx_id.size(); ++t) {
        fprintf(f, "tx %s %s %s %d %c %u", m.tx_id[t].c_str(), m.tx_gene_name[t].c_str(), m.tx_gene_id[t].c_str(), m.tx_chrom[t], m.tx_strand[t], m.tx_bin[t]);
        for (uint32_t q = 0; q < m.tx_n_exons[t]; ++q) fprintf(f, " %u-%u", m.es[m.tx_exon_off[t] + q], m.ee[m.tx_exon_off[t] + q]);
        fputc('\n', f);
    }
    for (size_t i = 0; i < m.bin_key.size(); ++i) fprintf(f, "bin %llu %u %u\n", (unsigned long long)(m.bin_key[i] >> 32), (unsigned)(m.bin_key[i] & 0xffffffffu), m.bin_tx[i]);
    {   // the direct index must describe the table above exactly
        bool ok = m.bin_start.size() == m.chroms.size() * (size_t)m.bin_stride + 1 && m.bin_start.back() == m.bin_key.size() && m.bin_start[0] == 0;
        for (size_t k = 0; ok && k + 1 < m.bin_start.size(); ++k) {
            if (m.bin_start[k] > m.bin_start[k + 1]) ok = false;
            for (uint32_t j = m.bin_start[k]; ok && j < m.bin_start[k + 1]; ++j)
                if (m.bin_key[j] != ((uint64_t)(k / m.bin_stride) << 32 | (uint64_t)(k % m.bin_stride))) ok = false;
        }
        fprintf(f, "bin_start %s\n", ok ? "ok" : "BAD");
    }
    fclose(f);
    return 0;
}

// worker_pool.h / bigvec.h on their own: parallel_sort against std::sort, the pool's every-task-once contract, huge-page vectors
#include <stdexcept>
#include "../../regtools_amd/csrc/worker_pool.h"
#include "../../regtools_amd/csrc/bigvec.h"
extern "C" int emu_pool_selftest(uint32_t seed, uint32_t n, uint32_t threads) {
    rgx::WorkerPool pool(threads);
    uint64_t x = seed * 0x9e3779b97f4a7c15ull + 1;
    auto rnd = [&] { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    rgx::BigVec<uint64_t> a(n), b;
    for (auto &v : a) v = rnd() % (n / 3 + 2);                     // plenty of equal keys
    b.assign(a.begin(), a.end());
    std::sort(b.begin(), b.end());
    rgx::parallel_sort(pool, a.begin(), a.end(), [](uint64_t p, uint64_t q) { return p < q; });
    if (a.size() != b.size() || !std::equal(a.begin(), a.end(), b.begin())) return 1;
    // every task exactly once, whatever the task count
    for (size_t tasks : {(size_t)0, (size_t)1, (size_t)2, (size_t)threads, (size_t)threads * 7 + 3}) {
        std::vector<std::atomic<int>> hit(tasks);
        for (auto &h : hit) h = 0;
        pool.run(tasks, [&](size_t k) { hit[k].fetch_add(1); });
        for (auto &h : hit) if (h.load() != 1) return 2;
    }
    // a task that throws: the phase still runs every other task once, the first exception comes out of run(), the pool stays usable
    {
        const size_t tasks = (size_t)threads * 5 + 2;
        std::vector<std::atomic<int>> hit(tasks);
        for (auto &h : hit) h = 0;
        bool thrown = false;
        try { pool.run(tasks, [&](size_t k) { hit[k].fetch_add(1); if (k == tasks / 2) throw std::runtime_error("task"); }); }
        catch (const std::runtime_error &) { thrown = true; }
        if (!thrown) return 6;
        for (auto &h : hit) if (h.load() != 1) return 7;
        std::atomic<int> sum{0};
        pool.run(tasks, [&](size_t) { sum.fetch_add(1); });
        if (sum.load() != (int)tasks) return 8;
        // a run() from inside a task does its work on the calling thread instead of queueing behind itself
        std::atomic<int> inner{0};
        pool.run(4, [&](size_t) { pool.run(3, [&](size_t) { inner.fetch_add(1); }); });
        if (inner.load() != 12) return 9;
    }
    // a block above the huge-page threshold: aligned, writable end to end, survives growth
    rgx::BigVec<uint32_t> big;
    big.resize((3u << 20) / 4 + 5);
    if (((uintptr_t)big.data() & ((2u << 20) - 1)) != 0) return 3;
    for (size_t i = 0; i < big.size(); ++i) big[i] = (uint32_t)i;
    big.resize((9u << 20) / 4 + 1);
    for (size_t i = 0; i < (3u << 20) / 4 + 5; ++i) if (big[i] != (uint32_t)i) return 4;
    big.back() = 7;
    rgx::BigVec<std::string> names(1000);
    for (auto &s : names) if (!s.empty()) return 5;                 // strings ARE constructed (only trivial types are left untouched)
    return 0;
}

// (lab) wall time of GtfModel::load alone, in ms: what the call costs its caller including the teardown of its temporaries
#include <chrono>
extern "C" double emu_gtf_load_ms(const char *gtf_path) {
    const auto t0 = std::chrono::steady_clock::now();
    rgx::GtfModel *m = new rgx::GtfModel();
    const std::string e = m->load(gtf_path);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    delete m;
    return e.empty() ? ms : -1.0;
}
