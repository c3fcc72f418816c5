// host_io.cpp -- see host_io.h.
#include "host_io.h"
#include "worker_pool.h"

#include <thread>
#include "inflate_core.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <ctype.h>
#include <limits.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <unordered_map>

namespace rgx {

static inline uint16_t h16(const uint8_t *p) { return (uint16_t)(p[0] | p[1] << 8); }
static inline uint32_t h32(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
static inline uint64_t h64(const uint8_t *p) { return (uint64_t)h32(p) | (uint64_t)h32(p + 4) << 32; }

void walk_members(const uint8_t *bam, size_t len, std::vector<HostMember> &out) {
    out.clear();
    size_t off = 0;
    while (off < len) {
        if (len - off < 18) break;
        const uint8_t *h = bam + off;
        bool ok = h[0] == 31 && h[1] == 139 && h[2] == 8 && (h[3] & 4) && h16(h + 10) == 6 && h[12] == 'B' && h[13] == 'C' && h16(h + 14) == 2;
        if (!ok) break;
        size_t blen = (size_t)h16(h + 16) + 1;
        if (blen < 26 || off + blen > len) break;
        out.push_back({(uint64_t)off, (uint32_t)blen, h32(h + blen - 4)});
        off += blen;
    }
}

unsigned usable_threads(unsigned cap) {
    static const unsigned limit = [] {
        unsigned hw = std::thread::hardware_concurrency();
        if (!hw) hw = 4;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                 // cgroup v2: "<quota> <period>" or "max <period>"
            long long q = 0, per = 0;
            if (fscanf(f, "%lld %lld", &q, &per) == 2 && q > 0 && per > 0) hw = std::min<unsigned>(hw, (unsigned)std::max<long long>(1, (q + per - 1) / per));
            fclose(f);
        }
        // one process per GPU (torchrun): the node's cores are shared by the ranks on it
        for (const char *name : {"LOCAL_WORLD_SIZE", "WORLD_SIZE"})
            if (const char *e = getenv(name)) { const int w = atoi(e); if (w > 1) { hw = std::max(1u, hw / (unsigned)w); break; } }
        return hw;
    }();
    return std::max(1u, std::min(limit, cap));
}

static inline bool host_magic_at(const uint8_t *h) {
    return h[0] == 31 && h[1] == 139 && h[2] == 8 && (h[3] & 4) && h16(h + 10) == 6 && h[12] == 'B' && h[13] == 'C' && h16(h + 14) == 2;
}

bool scan_members_parallel(const uint8_t *bam, size_t len, int threads, std::vector<Member> &out, uint64_t &total_inflated) {
    out.clear(); total_inflated = 0;
    if (len < 28 || !host_magic_at(bam)) return false;
    int T = threads < 1 ? 1 : threads;
    // The walk is a chain of dependent cache misses (BSIZE of one member names the next): every thread walks several chains in turn, so that a
    // core keeps as many misses in flight as a dozen threads would -- the ranks of a multi-GPU job share the host's cores.
    constexpr int kChains = 8;
    size_t n_seg = (size_t)T * kChains;
    if (n_seg > len / (1u << 19) + 1) n_seg = len / (1u << 19) + 1;           // not worth a chain per few members
    if ((size_t)T > n_seg) T = (int)n_seg;
    std::vector<size_t> start(n_seg + 1, len);
    start[0] = 0;
    for (size_t t = 1; t < n_seg; ++t) {
        size_t p = std::max(start[t - 1] + 1, (size_t)((double)len * (double)t / (double)n_seg));
        size_t found = len;
        while (p + 18 <= len) {
            const uint8_t *q = (const uint8_t *)memchr(bam + p, 31, len - 18 - p + 1);
            if (!q) break;
            if (host_magic_at(q)) { found = (size_t)(q - bam); break; }
            p = (size_t)(q - bam) + 1;
        }
        start[t] = found;
    }
    // (the chains' lists live across calls -- whoever holds the pool's lock below owns them: 4 MB of fresh vectors were a thousand page faults per call)
    static std::mutex pool_mu;
    static WorkerPool *pool = nullptr;
    std::unique_lock<std::mutex> pool_lock(pool_mu, std::try_to_lock);       // (a second caller at the same time -- shards of different files -- walks on its own thread)
    static std::vector<std::vector<Member>> part_keep;
    std::vector<std::vector<Member>> part_own;
    std::vector<std::vector<Member>> &part = pool_lock.owns_lock() ? part_keep : part_own;
    if (part.size() < n_seg) part.resize(n_seg);
    for (size_t k = 0; k < n_seg; ++k) part[k].clear();
    std::vector<char> ok(n_seg, 0);
    auto walk = [&](int t) {
        const size_t s0 = n_seg * (size_t)t / (size_t)T, s1 = n_seg * ((size_t)t + 1) / (size_t)T;
        std::vector<size_t> off(s1 - s0);
        std::vector<char> live(s1 - s0, 1);
        for (size_t k = s0; k < s1; ++k) { off[k - s0] = start[k]; part[k].reserve((start[k + 1] - start[k]) / 2048 + 16); if (start[k] >= start[k + 1]) { live[k - s0] = 0; ok[k] = start[k] == start[k + 1]; } }
        for (size_t alive = 1; alive;) {
            alive = 0;
            for (size_t k = s0; k < s1; ++k) {
                if (!live[k - s0]) continue;
                const size_t o = off[k - s0], end = start[k + 1];
                if (len - o < 18 || !host_magic_at(bam + o)) { live[k - s0] = 0; continue; }
                const size_t blen = (size_t)h16(bam + o + 16) + 1;
                if (blen < 26 || o + blen > len) { live[k - s0] = 0; continue; }
                Member m; m.cpos = o + 18; m.upos = 0; m.isize = h32(bam + o + blen - 4);
                if (m.isize > kBgzfMaxBlock) { live[k - s0] = 0; continue; }
                m.clen = (uint32_t)(blen - 18);
                if (m.cpos + m.clen + 8 > len) m.clen = len > m.cpos + 8 ? (uint32_t)(len - 8 - m.cpos) : 0;      // as k_member_compact
                part[k].push_back(m);
                const size_t nx = o + blen;
                off[k - s0] = nx;
                if (nx >= end) { live[k - s0] = 0; ok[k] = nx == end; }
                else { __builtin_prefetch(bam + nx + 16); ++alive; }
            }
        }
    };
    // (round 4: the walkers are a pool that outlives the call -- starting two dozen threads was half of the scan's 1.8 ms, and the inflate launch waits
    //  for this list -- and the lists of the chains are put together by the same threads)
    if (pool_lock.owns_lock() && T > 1 && (!pool || pool->threads() < (size_t)T)) { delete pool; pool = new WorkerPool((size_t)T); }
    auto run = [&](size_t n_tasks, const std::function<void(size_t)> &f) {
        if (pool_lock.owns_lock() && pool && T > 1) pool->run(n_tasks, f);
        else for (size_t k = 0; k < n_tasks; ++k) f(k);
    };
    run((size_t)T, [&](size_t t) { walk((int)t); });
    std::vector<size_t> first(n_seg + 1, 0);
    std::vector<uint64_t> up0(n_seg + 1, 0);
    for (size_t t = 0; t < n_seg; ++t) { if (!ok[t]) return false; first[t + 1] = first[t] + part[t].size(); }
    const size_t n = first[n_seg];
    if (n == 0 || n >= 0xfffffff0u) return false;
    out.resize(n);
    run(n_seg, [&](size_t t) { uint64_t u = 0; for (const Member &m : part[t]) u += m.isize; up0[t + 1] = u; });
    for (size_t t = 0; t < n_seg; ++t) up0[t + 1] += up0[t];
    run(n_seg, [&](size_t t) {
        uint64_t up = up0[t]; Member *dst = out.data() + first[t];
        for (Member m : part[t]) { m.upos = up; up += m.isize; *dst++ = m; }
    });
    total_inflated = up0[n_seg];
    return true;
}

bool parse_bai(const uint8_t *d, size_t len, BaiInfo &bi, bool collect_anchors) {
    bi = BaiInfo();
    if (len < 8 || memcmp(d, "BAI\1", 4)) return false;
    size_t p = 4;
    bi.n_ref = (int32_t)h32(d + p); p += 4;
    bi.start_voff = UINT64_MAX;
    for (int32_t r = 0; r < bi.n_ref; ++r) {
        if (p + 4 > len) return false;
        int32_t n_bin = (int32_t)h32(d + p); p += 4;
        for (int32_t b = 0; b < n_bin; ++b) {
            if (p + 8 > len) return false;
            uint32_t bin = h32(d + p); int32_t n_chunk = (int32_t)h32(d + p + 4); p += 8;
            if (n_chunk < 0 || p + (size_t)n_chunk * 16 > len) return false;
            if (bin == 37450) {                                  // pseudo-bin: chunk 0 = (first offset, last offset)
                if (n_chunk > 0) { uint64_t u = h64(d + p); bi.have_start = true; if (u < bi.start_voff) bi.start_voff = u; }
                if (n_chunk > 0 && r == bi.n_ref - 1) { bi.have_nocoor = true; bi.nocoor_voff = h64(d + p + 8); }
            } else if (collect_anchors) {
                for (int32_t c = 0; c < n_chunk; ++c) bi.anchors.push_back(h64(d + p + (size_t)c * 16));
            }
            p += (size_t)n_chunk * 16;
        }
        if (p + 4 > len) return false;
        int32_t n_intv = (int32_t)h32(d + p); p += 4;
        if (n_intv < 0 || p + (size_t)n_intv * 8 > len) return false;
        if (collect_anchors) for (int32_t i = 0; i < n_intv; ++i) { uint64_t v = h64(d + p + (size_t)i * 8); if (v) bi.anchors.push_back(v); }
        p += (size_t)n_intv * 8;
    }
    bi.n_no_coor = (p + 8 <= len) ? h64(d + p) : 0;
    std::sort(bi.anchors.begin(), bi.anchors.end());
    bi.anchors.erase(std::unique(bi.anchors.begin(), bi.anchors.end()), bi.anchors.end());
    return true;
}

void bai_first_anchor_ge(const uint8_t *d, size_t len, const uint64_t *targets, int n, uint64_t *out) {
    for (int k = 0; k < n; ++k) out[k] = UINT64_MAX;
    if (len < 8 || memcmp(d, "BAI\1", 4)) return;
    auto see = [&](uint64_t v) { for (int k = 0; k < n; ++k) if (v >= targets[k] && v < out[k]) out[k] = v; };
    size_t p = 4;
    int32_t n_ref = (int32_t)h32(d + p); p += 4;
    for (int32_t r = 0; r < n_ref; ++r) {
        if (p + 4 > len) return;
        int32_t n_bin = (int32_t)h32(d + p); p += 4;
        for (int32_t b = 0; b < n_bin; ++b) {
            if (p + 8 > len) return;
            uint32_t bin = h32(d + p); int32_t n_chunk = (int32_t)h32(d + p + 4); p += 8;
            if (n_chunk < 0 || p + (size_t)n_chunk * 16 > len) return;
            if (bin != 37450) for (int32_t c = 0; c < n_chunk; ++c) see(h64(d + p + (size_t)c * 16));
            p += (size_t)n_chunk * 16;
        }
        if (p + 4 > len) return;
        int32_t n_intv = (int32_t)h32(d + p); p += 4;
        if (n_intv < 0 || p + (size_t)n_intv * 8 > len) return;
        for (int32_t i = 0; i < n_intv; ++i) { uint64_t v = h64(d + p + (size_t)i * 8); if (v) see(v); }
        p += (size_t)n_intv * 8;
    }
}

// hts_itr_query (hts.c:1733-1800) for one reference of a BAI image (or of an image converted from a .csi: normalize_index leaves the real
// bin numbers, the bins' loffs and the geometry behind it): min_off = loff of the nearest existing bin at / left of / above beg's
// finest-level bin (:1771-1783; a BAI's loffs come from its zero-filled linear index, update_loff :1330-1350 with the load-time fill
// :1543-1547), the chunks of the region's bins of every level (reg2bins :1690-1706) that end behind min_off, sorted, contained chunks
// dropped, overlaps cut, chunks that meet inside one compressed block joined (:1787-1797).
bool region_chunks(const uint8_t *d, size_t len, int32_t tid, int32_t beg, int32_t end, std::vector<VChunk> &out) {
    out.clear();
    if (len < 8 || memcmp(d, "BAI\1", 4) || tid < 0) return false;
    int32_t min_shift = 14, depth = 5;
    size_t body_end = len, x = 0;
    const bool ext = len >= 8 + 24 && !memcmp(d + len - 4, "RGXC", 4);
    if (ext) {
        const size_t ext_end = len - 20;
        min_shift = (int32_t)h32(d + ext_end); depth = (int32_t)h32(d + ext_end + 4);
        const uint64_t ext_off = h64(d + ext_end + 8);
        if (ext_off > ext_end || min_shift < 0 || min_shift > 31 || depth < 0 || depth > 12) return false;
        body_end = (size_t)ext_off; x = (size_t)ext_off;
    }
    size_t p = 4;
    const int32_t n_ref = (int32_t)h32(d + p); p += 4;
    if (tid >= n_ref) return false;
    auto first_of = [](int l) { return (uint64_t)((((uint64_t)1 << (3 * l)) - 1) / 7); };
    const uint64_t n_bins = first_of(depth + 1);
    for (int32_t r = 0; r <= tid; ++r) {
        if (p + 4 > body_end) return false;
        const int32_t n_bin = (int32_t)h32(d + p); p += 4;
        if (n_bin < 0 || (ext && x + (size_t)n_bin * 12 > len - 20)) return false;
        struct Bin { uint64_t loff; size_t chunks; int32_t n; };
        std::unordered_map<uint64_t, Bin> bins;
        if (r == tid) bins.reserve((size_t)n_bin * 2);
        for (int32_t b = 0; b < n_bin; ++b) {
            if (p + 8 > body_end) return false;
            uint64_t bin = h32(d + p); const int32_t n_chunk = (int32_t)h32(d + p + 4); p += 8;
            if (n_chunk < 0 || p + (size_t)n_chunk * 16 > body_end) return false;
            if (r == tid) {
                uint64_t loff = 0;
                if (ext) { const uint32_t real = h32(d + x + (size_t)b * 12); bin = real == kCsiMeta ? n_bins + 1 : real; loff = h64(d + x + (size_t)b * 12 + 4); }
                if (bin < n_bins) bins[bin] = Bin{loff, p, n_chunk};          // (the pseudo-bin n_bins + 1 holds no alignments)
            }
            p += (size_t)n_chunk * 16;
        }
        x += (size_t)n_bin * 12;
        if (p + 4 > body_end) return false;
        const int32_t n_intv = (int32_t)h32(d + p); p += 4;
        if (n_intv < 0 || p + (size_t)n_intv * 8 > body_end) return false;
        if (r != tid) { p += (size_t)n_intv * 8; continue; }
        if (!ext) {
            std::vector<uint64_t> lin((size_t)n_intv);
            for (int32_t i = 0; i < n_intv; ++i) { lin[(size_t)i] = h64(d + p + (size_t)i * 8); if (i > 0 && !lin[(size_t)i]) lin[(size_t)i] = lin[(size_t)i - 1]; }
            for (auto &kv : bins) {
                int l = 0;
                while (l < depth && kv.first >= first_of(l + 1)) ++l;
                const uint64_t bot = (kv.first - first_of(l)) << (3 * (depth - l));
                kv.second.loff = bot < (uint64_t)n_intv ? lin[(size_t)bot] : 0;
            }
        }
        if (beg < 0) beg = 0;
        if (end <= beg) return true;                                             // reg2bins: no bins, an iterator that returns nothing
        uint64_t min_off = 0;
        {
            uint64_t bin = first_of(depth) + ((uint64_t)(uint32_t)beg >> min_shift);
            bool found = false;
            do {
                auto it = bins.find(bin);
                if (it != bins.end()) { min_off = it->second.loff; found = true; break; }
                if (!bin) break;
                const uint64_t parent = (bin - 1) >> 3, first = (parent << 3) + 1;
                bin = bin > first ? bin - 1 : parent;
            } while (bin);
            if (!found) { auto it = bins.find(0); if (it != bins.end()) min_off = it->second.loff; }
        }
        int64_t e = end;
        const int top = min_shift + 3 * depth;
        if (top < 62 && e >= ((int64_t)1 << top)) e = (int64_t)1 << top;
        const uint64_t b0 = (uint64_t)beg, e0 = (uint64_t)(e - 1);
        for (auto &kv : bins) {
            int l = 0;
            while (l < depth && kv.first >= first_of(l + 1)) ++l;
            const int sh = min_shift + 3 * (depth - l);
            const uint64_t k = kv.first - first_of(l);
            if (k < (b0 >> sh) || k > (e0 >> sh)) continue;
            for (int32_t c = 0; c < kv.second.n; ++c) {
                const uint64_t u = h64(d + kv.second.chunks + (size_t)c * 16), v = h64(d + kv.second.chunks + (size_t)c * 16 + 8);
                if (v > min_off) out.push_back(VChunk{u, v});
            }
        }
        if (out.empty()) return true;
        std::sort(out.begin(), out.end(), [](const VChunk &a, const VChunk &b) { return a.u != b.u ? a.u < b.u : a.v < b.v; });
        size_t l = 0;
        for (size_t i = 1; i < out.size(); ++i) if (out[l].v < out[i].v) out[++l] = out[i];
        out.resize(l + 1);
        for (size_t i = 1; i < out.size(); ++i) if (out[i - 1].v >= out[i].u) out[i - 1].v = out[i].u;
        l = 0;
        for (size_t i = 1; i < out.size(); ++i) { if (out[l].v >> 16 == out[i].u >> 16) out[l].v = out[i].v; else out[++l] = out[i]; }
        out.resize(l + 1);
        return true;
    }
    return false;
}

bool bai_region_span(const uint8_t *d, size_t len, int32_t tid, int32_t beg, int32_t end, uint64_t &lo, uint64_t &hi, bool &usable) {
    std::vector<VChunk> ch;
    usable = region_chunks(d, len, tid, beg, end, ch);
    if (!usable || ch.empty()) return false;
    lo = ch.front().u; hi = ch.front().v;
    for (const VChunk &c : ch) hi = std::max(hi, c.v);
    return true;
}

bool host_bam_header(const uint8_t *d, size_t n, BamHeader &h, size_t *consumed, uint32_t *mean_record_bytes) {
    std::string plain;
    size_t off = 0;
    if (mean_record_bytes) *mean_record_bytes = 0;
    bool have_header = false;
    int extra = 0;
    auto estimate = [&]() -> bool {                                   // records that lie whole in what is inflated so far; true = enough of them
        uint64_t o = h.end, sum = 0; uint32_t cnt = 0;
        while (o + 4 <= plain.size()) {
            const uint32_t bl = h32((const uint8_t *)plain.data() + o);
            if (bl < 32 || bl > (1u << 27) || o + 4 + bl > plain.size()) break;
            sum += 4 + (uint64_t)bl; ++cnt; o += 4 + (uint64_t)bl;
        }
        if (mean_record_bytes && cnt) *mean_record_bytes = (uint32_t)(sum / cnt);
        return cnt >= 4;
    };
    for (int members = 0; off + 18 <= n && members < 4096; ++members) {
        if (d[off] != 31 || d[off + 1] != 139 || d[off + 2] != 8 || !(d[off + 3] & 4) || d[off + 10] != 6 || d[off + 11] != 0 || d[off + 12] != 'B' || d[off + 13] != 'C') return have_header;
        const size_t bl = (size_t)h16(d + off + 16) + 1;
        if (bl < 26 || off + bl + 16 > n) return have_header;                    // (the decoder may look 16 bytes past a payload)
        const uint32_t isz = h32(d + off + bl - 4);
        if (isz == 0 || isz > 65536) return have_header;                         // an empty member ends the header read upstream: let the device path judge
        const size_t base = plain.size();
        plain.resize(base + isz + 64);
        HostTab T; uint32_t got = 0;
        if (inflate_raw(d + off + 18, (uint32_t)(bl - 26), (uint8_t *)&plain[base], isz, &got, T) != INF_OK || got != isz) { plain.resize(base); return have_header; }
        plain.resize(base + isz);
        off += bl;
        if (have_header) {                                                 // (only here for the record-size estimate)
            if (estimate() || ++extra >= 2) return true;
            continue;
        }
        uint64_t need = 0;
        const int r = parse_bam_header((const uint8_t *)plain.data(), plain.size(), h, need);
        if (r == 0) {
            if (consumed) *consumed = off;
            if (!mean_record_bytes || estimate()) return true;
            have_header = true;                                            // a few more bytes of the stream, to see some records
            continue;
        }
        if (r == 2) return false;
    }
    return have_header;
}

static bool readable(const std::string &p) { FILE *f = fopen(p.c_str(), "rb"); if (!f) return false; fclose(f); return true; }

static bool idx_name(const std::string &fn, const char *ext, std::string &out) {
    std::string a = fn + ext;
    if (readable(a)) { out = a; return true; }
    size_t i = fn.size();
    for (i = fn.size(); i-- > 1;) if (fn[i] == '.') break;     // i in [1, len-1]; falls to 0 when no '.'
    if (fn.size() < 2) i = 0;
    std::string b = fn.substr(0, i) + ext;
    if (readable(b)) { out = b; return true; }
    return false;
}

bool find_tbi(const std::string &path, std::string &out) { return idx_name(path, ".tbi", out); }       // hts_idx_getfn, as for the BAM indexes

// What htslib says when a BAM and its index are opened (stderr, default verbosity): the file does not end in the empty BGZF member
// (bam_hdr_read -> bgzf_check_EOF, sam.c:122-127, bgzf.c:835-846); the index is older than the file (hts_idx_load2, hts.c:2046-2054: whole seconds).
std::string bam_open_notes(const uint8_t *bam, size_t len, const char *bam_path, const char *index_path) {
    static const uint8_t kEof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    std::string s;
    if (bam) {
        if (len < 28) s += "[W::bam_hdr_read] bgzf_check_EOF: Invalid argument\n";            // (perror: the seek to 28 bytes before the end fails)
        else if (memcmp(bam + len - 28, kEof, 28)) s += "[W::bam_hdr_read] EOF marker is absent. The input is probably truncated.\n";
    }
    struct stat sb, si;
    if (bam_path && index_path && !stat(bam_path, &sb) && !stat(index_path, &si) && si.st_mtime < sb.st_mtime)
        s += std::string("Warning: The index file is older than the data file: ") + index_path + "\n";
    return s;
}

int find_index(const std::string &bam_path, std::string &out) {      // hts_idx_load: <fn>.csi, <stem>.csi, <fn>.bai, <stem>.bai (hts.c:2031-2042)
    if (idx_name(bam_path, ".csi", out)) return 0;
    if (idx_name(bam_path, ".bai", out)) return 0;
    return 1;
}

bool read_file(const std::string &path, std::vector<uint8_t> &out) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (n < 0) { fclose(f); return false; }
    out.resize((size_t)n);
    bool ok = n == 0 || fread(out.data(), 1, (size_t)n, f) == (size_t)n;
    fclose(f);
    return ok;
}

std::string gunzip_all(const uint8_t *d, size_t n, std::string &out) {
    size_t off = 0;
    out.clear();
    while (off + 18 <= n && d[off] == 0x1f && d[off + 1] == 0x8b) {
        if (d[off + 2] != 8) return "regtools_amd: unsupported compression method in gzip input\n\n";
        const uint8_t flg = d[off + 3];
        size_t q = off + 10;
        size_t member_len = 0;                                   // known for BGZF members (BC subfield)
        if (flg & 4) {
            if (q + 2 > n) break;
            const size_t xlen = d[q] | d[q + 1] << 8; q += 2;
            for (size_t x = q; x + 4 <= q + xlen && x + 4 <= n;) {
                const size_t sl = d[x + 2] | d[x + 3] << 8;
                if (d[x] == 'B' && d[x + 1] == 'C' && sl == 2 && x + 6 <= n) member_len = (size_t)(d[x + 4] | d[x + 5] << 8) + 1;
                x += 4 + sl;
            }
            q += xlen;
        }
        if (flg & 8) { while (q < n && d[q]) ++q; ++q; }        // FNAME
        if (flg & 16) { while (q < n && d[q]) ++q; ++q; }       // FCOMMENT
        if (flg & 2) q += 2;                                     // FHCRC
        if (q + 8 > n) return "regtools_amd: truncated gzip input\n\n";
        if (member_len && (off + member_len < q + 8 || off + member_len > n)) return "regtools_amd: corrupt BGZF member in compressed input\n\n";
        const size_t in_len = member_len ? off + member_len - 8 - q : n - 8 - q;
        if (in_len > 0xfffffff0u) return "regtools_amd: gzip member too large for this path (use bgzip)\n\n";
        size_t cap = member_len ? 65536 : std::max<size_t>(in_len * 4, 1 << 16);
        uint32_t out_len = 0, used = 0;
        for (;;) {
            if (cap > 0xfffffff0u) return "regtools_amd: gzip member inflates to more than 4 GiB (use bgzip)\n\n";
            const size_t base = out.size();
            out.resize(base + cap + 64);
            HostTab T;
            std::vector<uint8_t> padded;                          // the decoder prefetches up to 16 bytes past the payload
            const uint8_t *src = d + q;
            if (q + in_len + 16 > n) { padded.assign(d + q, d + q + in_len); padded.resize(in_len + 32, 0); src = padded.data(); }
            const int st = inflate_raw(src, (uint32_t)in_len, (uint8_t *)&out[base], (uint32_t)cap, &out_len, T, &used);
            if (st == INF_OK) { out.resize(base + out_len); break; }
            out.resize(base);
            // DEFLATE cannot expand more than 1032:1 (a 258-byte match per 2 bits): a stream that still overflows is corrupt, not big
            if (st == INF_OUT_OVERFLOW && !member_len && cap < in_len * 1032 + 65536) { cap *= 2; continue; }
            return "regtools_amd: corrupt compressed input\n\n";
        }
        off = member_len ? off + member_len : q + used + 8;      // + CRC32, ISIZE
    }
    return "";
}

bool normalize_index(const uint8_t *in, size_t n, std::vector<uint8_t> &storage, const uint8_t *&out, size_t &out_len) {
    std::string plain;
    const uint8_t *d = in; size_t len = n;
    if (n >= 2 && in[0] == 0x1f && in[1] == 0x8b) {
        if (!gunzip_all(in, n, plain).empty()) return false;
        d = (const uint8_t *)plain.data(); len = plain.size();
    }
    if (len >= 8 && !memcmp(d, "BAI\1", 4)) {
        if (d == in) { out = in; out_len = n; return true; }
        storage.assign(d, d + len); out = storage.data(); out_len = storage.size();
        return true;
    }
    if (len < 20 || memcmp(d, "CSI\1", 4)) return false;
    const int32_t min_shift = (int32_t)h32(d + 4), depth = (int32_t)h32(d + 8), l_aux = (int32_t)h32(d + 12);
    // `samtools index -c` defaults (min_shift 14, depth 5) are the BAI's own geometry: the bins keep their numbers and the level-5 bins'
    // lower bounds (the linear-index entry of their 16 KiB window, hts.c:1330-1350) stand in for the linear index, so region queries
    // find their member range as with a .bai.  Any other geometry: the bins are renumbered kCsiBin (a deeper geometry has real bins
    // numbered like a BAI's pseudo-bin) and their numbers and loffs follow the image in a block of their own.
    const bool bai_geometry = min_shift == 14 && depth == 5;
    if (depth < 0 || depth > 12 || l_aux < 0 || 16 + (size_t)l_aux + 4 > len) return false;
    const uint32_t meta_bin = (uint32_t)((((uint64_t)1 << (3 * depth + 3)) - 1) / 7 + 1);   // META_BIN: n_bins + 1 (hts.c:1277)
    size_t p = 16 + (size_t)l_aux;
    const int32_t n_ref = (int32_t)h32(d + p); p += 4;
    if (n_ref < 0) return false;
    std::vector<uint8_t> &o = storage;
    o.clear();
    std::vector<uint32_t> ext_bins; std::vector<uint64_t> ext_loffs; std::vector<int32_t> ext_counts;
    auto w32 = [&](uint32_t v) { for (int k = 0; k < 4; ++k) o.push_back((uint8_t)(v >> (8 * k))); };
    auto w64 = [&](uint64_t v) { for (int k = 0; k < 8; ++k) o.push_back((uint8_t)(v >> (8 * k))); };
    o.insert(o.end(), {'B', 'A', 'I', 1});
    w32((uint32_t)n_ref);
    for (int32_t r = 0; r < n_ref; ++r) {
        if (p + 4 > len) return false;
        const int32_t n_bin = (int32_t)h32(d + p); p += 4;
        if (n_bin < 0) return false;
        w32((uint32_t)n_bin);
        std::vector<uint64_t> loffs;
        if (bai_geometry) loffs.reserve(64);
        for (int32_t b = 0; b < n_bin; ++b) {
            if (p + 16 > len) return false;
            const uint32_t bin = h32(d + p); const uint64_t loff = h64(d + p + 4); const int32_t n_chunk = (int32_t)h32(d + p + 12); p += 16;
            if (n_chunk < 0 || p + (size_t)n_chunk * 16 > len) return false;
            w32(bin == meta_bin ? 37450u : bai_geometry ? bin : kCsiBin);                      // other geometries: an id no BAI has
            w32((uint32_t)n_chunk);
            o.insert(o.end(), d + p, d + p + (size_t)n_chunk * 16);
            if (bai_geometry) {
                if (bin >= 4681 && bin < 37449 && loff) { const size_t w = bin - 4681; if (loffs.size() <= w) loffs.resize(w + 1, 0); loffs[w] = loff; }
            } else if (bin != meta_bin && loff) loffs.push_back(loff);
            ext_bins.push_back(bin == meta_bin ? kCsiMeta : bin); ext_loffs.push_back(loff);
            p += (size_t)n_chunk * 16;
        }
        ext_counts.push_back(n_bin);
        w32((uint32_t)loffs.size());                                                           // "linear index": the bins' lower bounds
        for (uint64_t v : loffs) w64(v);
    }
    w64(p + 8 <= len ? h64(d + p) : 0);
    {
        // the real bin numbers and the bins' loff, in the order of the image's bins: what a region query needs (region_chunks; also for the
        // default geometry: the loff a .csi stores for a bin is not always what a BAI loader would derive from a linear index);
        // layout: per reference n_bin x (u32 bin, u64 loff), then i32 min_shift, i32 depth, u64 offset of this block, "RGXC"
        const size_t ext_off = o.size();
        size_t k = 0;
        for (int32_t r = 0; r < n_ref; ++r) for (int32_t b = 0; b < ext_counts[(size_t)r]; ++b, ++k) { w32(ext_bins[k]); w64(ext_loffs[k]); }
        w32((uint32_t)min_shift); w32((uint32_t)depth); w64(ext_off);
        o.insert(o.end(), {'R', 'G', 'X', 'C'});
    }
    out = o.data(); out_len = o.size();
    return true;
}

bool read_index(const std::string &path, std::vector<uint8_t> &out) {
    std::vector<uint8_t> raw, image;
    if (!read_file(path, raw)) return false;
    const uint8_t *d; size_t n;
    if (!normalize_index(raw.data(), raw.size(), image, d, n)) return false;
    if (d == raw.data()) out.swap(raw); else out.swap(image);
    return true;
}

FileBytes::~FileBytes() { release(); }
void FileBytes::release_later() {
    if (mapped && p && n) { uint8_t *q = const_cast<uint8_t *>(p); const size_t len = n; Reaper::get().later([q, len] { munmap(q, len); }); p = nullptr; n = 0; mapped = false; }
    release();
}
void FileBytes::release() { if (mapped && p && n) munmap(const_cast<uint8_t *>(p), n); p = nullptr; n = 0; mapped = false; std::vector<uint8_t>().swap(own); }

bool FileBytes::open(const std::string &path, bool populate) {
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
        void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE | (populate ? MAP_POPULATE : 0), fd, 0);
        if (m != MAP_FAILED) {
            ::close(fd);
            (void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
            p = static_cast<const uint8_t *>(m); n = (size_t)st.st_size; mapped = true;
            return true;
        }
    }
    ::close(fd);
    if (!read_file(path, own)) return false;
    p = own.data(); n = own.size();
    return true;
}

int parse_bam_header(const uint8_t *d, uint64_t have, BamHeader &h, uint64_t &need) {
    h = BamHeader();
    if (have < 12) { need = 12; return 1; }
    if (memcmp(d, "BAM\1", 4)) return 2;
    uint64_t q = 8 + (uint64_t)h32(d + 4);
    if (q + 4 > have) { need = q + 4; return 1; }
    int32_t n_ref = (int32_t)h32(d + q); q += 4;
    for (int32_t i = 0; i < n_ref; ++i) {
        if (q + 4 > have) { need = q + 4 + 64; return 1; }
        uint32_t ln = h32(d + q);
        if (q + 4 + (uint64_t)ln + 4 > have) { need = q + 8 + ln + 64; return 1; }
        // target names are NUL-terminated C strings upstream
        std::string name((const char *)d + q + 4, ln);
        size_t z = name.find('\0'); if (z != std::string::npos) name.resize(z);
        h.names.push_back(name);
        h.lens.push_back(h32(d + q + 4 + ln));
        q += 4 + (uint64_t)ln + 4;
    }
    h.end = q;
    return 0;
}

// (say: the two lines hts_parse_decimal writes to stderr at htslib's default verbosity, hts.c:1865-1872)
static long long parse_decimal(const char *str, const char **end, bool say) {
    long long n = 0; int decimals = 0, e = 0, lost = 0; char sign = '+', esign = '+';
    while (isspace((unsigned char)*str)) str++;
    const char *s = str;
    if (*s == '+' || *s == '-') sign = *s++;
    while (*s) {
        if (isdigit((unsigned char)*s)) n = 10 * n + (*s++ - '0');
        else if (*s == ',') s++;
        else break;
    }
    if (*s == '.') { s++; while (isdigit((unsigned char)*s)) { decimals++; n = 10 * n + (*s++ - '0'); } }
    if (*s == 'E' || *s == 'e') {
        s++;
        if (*s == '+' || *s == '-') esign = *s++;
        while (isdigit((unsigned char)*s)) e = 10 * e + (*s++ - '0');
        if (esign == '-') e = -e;
    }
    e -= decimals;
    while (e > 0) { n *= 10; e--; }
    while (e < 0) { lost += (int)(n % 10); n /= 10; e++; }
    if (say && lost > 0) fprintf(stderr, "[W::hts_parse_decimal] discarding fractional part of %.*s\n", (int)(s - str), str);
    if (end) *end = s;
    else if (say && *s) fprintf(stderr, "[W::hts_parse_decimal] ignoring unknown characters after %.*s[%s]\n", (int)(s - str), str, s);
    return sign == '+' ? n : -n;
}

static int name2id(const BamHeader &h, const std::string &name) {
    int id = -1;
    for (size_t i = 0; i < h.names.size(); ++i) if (h.names[i] == name) id = (int)i;
    return id;
}

bool parse_region(const BamHeader &h, const char *reg, int32_t &tid, int32_t &beg, int32_t &end, bool say) {
    const char *colon = strrchr(reg, ':');
    bool parsed = false;
    if (!colon) { beg = 0; end = INT_MAX; parsed = true; colon = reg + strlen(reg); }
    else {
        const char *hy;
        beg = (int32_t)(parse_decimal(colon + 1, &hy, say) - 1);
        if (beg < 0) beg = 0;
        if (*hy == '\0') { end = INT_MAX; parsed = true; }
        else if (*hy == '-') { end = (int32_t)parse_decimal(hy + 1, nullptr, say); parsed = true; }
        if (parsed && beg >= end) parsed = false;
    }
    if (parsed) tid = name2id(h, std::string(reg, (size_t)(colon - reg)));
    else { tid = name2id(h, reg); beg = 0; end = INT_MAX; }
    return tid >= 0;
}

}  // namespace rgx
