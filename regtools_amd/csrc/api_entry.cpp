// api_entry.cpp -- the extract entry points of the C ABI, the kernels' test entry points, packing / unpacking / merging of tables (host and device).
#include "api_internal.h"

extern "C" int rgx_extract_device(rgx_ctx *ctx, const void *d_bam, size_t bam_len, const void *bai, size_t bai_len,
                                  const rgx_extract_params *p, rgx_junction_table **out, char *err, size_t errlen) {
    if (!ctx || !d_bam || !out) return fail(err, errlen, RGX_ERR_ARG, "regtools_amd: bad arguments\n");
    return run_pipeline(ctx, (const uint8_t *)d_bam, nullptr, bam_len, (const uint8_t *)bai, bai_len, p, out, err, errlen);
}

extern "C" int rgx_extract_mem(rgx_ctx *ctx, const void *bam, size_t bam_len, const void *bai, size_t bai_len, const rgx_extract_params *p,
                               rgx_junction_table **out, char *err, size_t errlen) {
    if (!ctx || !bam || !out) return fail(err, errlen, RGX_ERR_ARG, "regtools_amd: bad arguments\n");
    fputs(bam_open_notes((const uint8_t *)bam, bam_len, nullptr, nullptr).c_str(), stderr);         // (a file without its EOF member: bam_hdr_read says so)
    return run_pipeline(ctx, nullptr, (const uint8_t *)bam, bam_len, (const uint8_t *)bai, bai_len, p, out, err, errlen);
}

// rgx_extract_multi's shards: the same call with the member list the caller scanned once (multi.cpp)
int rgx_extract_mem_scanned(rgx_ctx *ctx, const void *bam, size_t bam_len, const void *bai, size_t bai_len, const rgx_extract_params *p,
                            const std::vector<rgx::Member> *members, uint64_t total_inflated, rgx_junction_table **out, char *err, size_t errlen) {
    if (!ctx || !bam || !out) return fail(err, errlen, RGX_ERR_ARG, "regtools_amd: bad arguments\n");
    SharedMembers sm{members, total_inflated};
    return run_pipeline(ctx, nullptr, (const uint8_t *)bam, bam_len, (const uint8_t *)bai, bai_len, p, out, err, errlen, members && !members->empty() ? &sm :
        nullptr);
}

extern "C" void *rgx_host_alloc(size_t bytes) {
    void *p = nullptr;
    return hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? p : nullptr;
}
extern "C" void rgx_host_free(void *p) { if (p) (void)hipHostFree(p); }

extern "C" int rgx_extract(rgx_ctx *ctx, const char *bam_path, const rgx_extract_params *p, rgx_junction_table **out, char *err, size_t errlen) {
    if (!ctx || !bam_path || !out) return fail(err, errlen, RGX_ERR_ARG, "regtools_amd: bad arguments\n");
    FileBytes bam; std::vector<uint8_t> bai;
    // (hts_open_format's line in front of regtools' own, hts.c:408-411)
    if (!bam.open(bam_path)) return fail(err, errlen, RGX_ERR_OPEN, "[E::hts_open_format] fail to open file '%s'\n%s", bam_path, kMsgOpen);
    std::string idx;
    int r = find_index(bam_path, idx);
    fputs(bam_open_notes(bam.data(), bam.size(), r == 0 ? bam_path : nullptr, r == 0 ? idx.c_str() : nullptr).c_str(), stderr);
    if (r != 0 || !read_index(idx, bai)) return fail(err, errlen, RGX_ERR_INDEX, "%s", kMsgIndex);
    return run_pipeline(ctx, nullptr, bam.data(), bam.size(), bai.data(), bai.size(), p, out, err, errlen);
}

extern "C" int rgx_k_inflate(const void *d_comp, const rgx_member *d_members, uint32_t n_members, void *d_arena, uint32_t *d_status, void *stream) {
    return rgx_k_inflate_form(0, d_comp, d_members, n_members, d_arena, d_status, stream);
}

extern "C" int rgx_k_inflate_form(int form, const void *d_comp, const rgx_member *d_members, uint32_t n_members, void *d_arena, uint32_t *d_status,
    void *stream) {
    if (form < 0 || form > 5) return RGX_ERR_ARG;                    // 5 = k_inflate with up to four literals per trip
    static_assert(sizeof(rgx_member) == sizeof(Member), "rgx_member layout");
    // stage entry point: the code-length scratch is a process-lifetime buffer grown on demand
    static void *scratch = nullptr; static size_t scratch_cap = 0;
    const size_t need = inflate_scratch_bytes(n_members);
    if (need > scratch_cap) {
        if (scratch) (void)hipFree(scratch);
        if (hipMalloc(&scratch, need) != hipSuccess) { scratch = nullptr; scratch_cap = 0; return RGX_ERR_DEVICE; }
        scratch_cap = need;
    }
    launch_inflate((const uint8_t *)d_comp, (const Member *)d_members, n_members, (uint8_t *)d_arena, 0, (uint32_t *)scratch, d_status, (hipStream_t)stream,
        0, 0, false, form, nullptr, 1, true);
    return hipGetLastError() == hipSuccess ? RGX_OK : RGX_ERR_DEVICE;
}

// ---- multi-shard merge (host half of SURVEY 8e) -----------------------------------------------------------------------------------
extern "C" size_t rgx_table_pack(const rgx_junction_table *t, void *dst, size_t dst_cap) {
    const size_t need = (size_t)t->n * RGX_PACKED_ROW_BYTES;
    if (!dst || dst_cap < need) return need;
    uint8_t *q = (uint8_t *)dst;
    for (uint64_t i = 0; i < t->n; ++i, q += RGX_PACKED_ROW_BYTES) {
        uint32_t w[12] = {(uint32_t)t->tid[i], t->start[i], t->end[i], t->thick_start[i], t->thick_end[i], t->read_count[i],
                          (uint32_t)t->first_seen[i], (uint32_t)(t->first_seen[i] >> 32), (uint32_t)t->last_seen[i], (uint32_t)(t->last_seen[i] >> 32),
                          (uint32_t)(uint8_t)t->strand[i], (uint32_t)t->name_index[i]};
        memcpy(q, w, sizeof w);
    }
    return need;
}

extern "C" int rgx_table_unpack(const void *src, size_t n_rows, const rgx_junction_table *names_from, rgx_junction_table **out) {
    BamHeader h;
    for (int32_t i = 0; i < names_from->n_ref; ++i) { h.names.push_back(names_from->ref_name[i]); h.lens.push_back(names_from->ref_len[i]); }
    rgx_junction_table *t = table_alloc(h, n_rows);
    if (!t) return RGX_ERR_ARG;
    const uint8_t *q = (const uint8_t *)src;
    for (size_t i = 0; i < n_rows; ++i, q += RGX_PACKED_ROW_BYTES) {
        uint32_t w[12]; memcpy(w, q, sizeof w);
        t->tid[i] = (int32_t)w[0]; t->start[i] = w[1]; t->end[i] = w[2]; t->thick_start[i] = w[3]; t->thick_end[i] = w[4]; t->read_count[i] = w[5];
        t->first_seen[i] = (uint64_t)w[6] | (uint64_t)w[7] << 32; t->last_seen[i] = (uint64_t)w[8] | (uint64_t)w[9] << 32; t->strand[i] = (char)w[10];
        t->name_index[i] = w[11];
    }
    *out = t;
    return RGX_OK;
}

// The barcode lists of a table as one byte block (the wire format of the one-process-per-GPU driver, next to the 48-byte rows):
// u64 rows, u64 entries, u64 text bytes, then bc_row_begin (rows + 1 u64), bc_str_begin (entries + 1 u64), bc_count and bc_insert_rank
// (entries u32 each), the text.  Returns the bytes needed; writes when dst_cap suffices.  0 = the table carries no barcode lists.
extern "C" size_t rgx_table_pack_barcodes(const rgx_junction_table *t, void *dst, size_t dst_cap) {
    if (!t || !t->bc_row_begin) return 0;
    const uint64_t n = t->n, E = t->bc_row_begin[n], T = t->bc_str_begin ? t->bc_str_begin[E] : 0;
    const size_t need = 24 + (size_t)(n + 1) * 8 + (size_t)(E + 1) * 8 + (size_t)E * 8 + (size_t)T;
    if (!dst || dst_cap < need) return need;
    uint8_t *q = (uint8_t *)dst;
    const uint64_t head[3] = {n, E, T};
    memcpy(q, head, 24); q += 24;
    memcpy(q, t->bc_row_begin, (size_t)(n + 1) * 8); q += (size_t)(n + 1) * 8;
    memcpy(q, t->bc_str_begin, (size_t)(E + 1) * 8); q += (size_t)(E + 1) * 8;
    memcpy(q, t->bc_count, (size_t)E * 4); q += (size_t)E * 4;
    memcpy(q, t->bc_insert_rank, (size_t)E * 4); q += (size_t)E * 4;
    memcpy(q, t->bc_text, (size_t)T);
    return need;
}

// ... and back, onto a table of the same rows (rgx_table_unpack of the shard's packed rows).  Every offset is checked: the block crossed a wire.
extern "C" int rgx_table_unpack_barcodes(rgx_junction_table *t, const void *src, size_t len) {
    if (!t || !src || len < 24) return RGX_ERR_ARG;
    const uint8_t *q = (const uint8_t *)src;
    uint64_t head[3]; memcpy(head, q, 24); q += 24;
    const uint64_t n = head[0], E = head[1], T = head[2];
    if (n != t->n || E > (len >> 3) || T > len) return RGX_ERR_ARG;
    const size_t need = 24 + (size_t)(n + 1) * 8 + (size_t)(E + 1) * 8 + (size_t)E * 8 + (size_t)T;
    if (len < need) return RGX_ERR_ARG;
    uint64_t *row_begin = (uint64_t *)calloc((size_t)n + 1, 8), *str_begin = (uint64_t *)calloc((size_t)E + 1, 8);
    uint32_t *count = (uint32_t *)calloc((size_t)E + 1, 4), *rank = (uint32_t *)calloc((size_t)E + 1, 4);
    char *text = (char *)malloc((size_t)T + 1);
    bool ok = row_begin && str_begin && count && rank && text;
    if (ok) {
        memcpy(row_begin, q, (size_t)(n + 1) * 8); q += (size_t)(n + 1) * 8;
        memcpy(str_begin, q, (size_t)(E + 1) * 8); q += (size_t)(E + 1) * 8;
        memcpy(count, q, (size_t)E * 4); q += (size_t)E * 4;
        memcpy(rank, q, (size_t)E * 4); q += (size_t)E * 4;
        memcpy(text, q, (size_t)T);
        ok = row_begin[0] == 0 && row_begin[n] == E && str_begin[0] == 0 && str_begin[E] == T;
        for (uint64_t i = 0; ok && i < n; ++i) ok = row_begin[i] <= row_begin[i + 1];
        for (uint64_t k = 0; ok && k < E; ++k) ok = str_begin[k] <= str_begin[k + 1];
        // a row's ranks are a permutation of 0 .. (its entries - 1): rgx_table_merge_barcodes indexes by them
        for (uint64_t i = 0; ok && i < n; ++i) {
            const uint64_t b = row_begin[i], e = row_begin[i + 1];
            std::vector<uint8_t> seen((size_t)(e - b), 0);
            for (uint64_t k = b; ok && k < e; ++k) { ok = rank[k] < e - b && !seen[rank[k]]; if (ok) seen[rank[k]] = 1; }
        }
    }
    if (!ok) { free(row_begin); free(str_begin); free(count); free(rank); free(text); return RGX_ERR_ARG; }
    free(t->bc_row_begin); free(t->bc_count); free(t->bc_str_begin); free(t->bc_text); free(t->bc_insert_rank);
    t->bc_row_begin = row_begin; t->bc_str_begin = str_begin; t->bc_count = count; t->bc_insert_rank = rank; t->bc_text = text;
    return RGX_OK;
}

// -b across shards.  A junction's barcode map (junctions_extractor.h:58, cc:204-217) only depends on the sequence in which DISTINCT barcodes
// first reach it (a repeat bumps a count, it never moves a node): shard order is file order and bc_insert_rank keeps the order inside a shard,
// so the merged junction's sequence is the shards' sequences one after the other minus the barcodes already seen -- fed, as in barcode_rows, to
// the container the reference keeps, whose iteration order is the order print_barcodes writes (h:99-111).
extern "C" int rgx_table_merge_barcodes(const rgx_junction_table *const *parts, int n_parts, rgx_junction_table *t, char *err, size_t errlen) {
    if (!parts || n_parts <= 0 || !t) return fail(err, errlen, RGX_ERR_ARG, "regtools_amd: nothing to merge\n");
    for (int g = 0; g < n_parts; ++g) if (!parts[g] || !parts[g]->bc_row_begin) return fail(err, errlen, RGX_ERR_ARG,
        "regtools_amd: shard %d carries no barcode counts\n", g);
    auto cls = [](char c) { return c == '+' ? 0u : c == '-' ? 1u : 2u; };
    struct Key { int32_t tid; uint32_t start, end, cls; bool operator==(const Key &o) const { return tid == o.tid && start == o.start && end == o.end &&
        cls == o.cls; } };
    struct KeyHash {
        size_t operator()(const Key &k) const {
            const uint64_t h = (uint64_t)(uint32_t)k.tid * 0x9e3779b97f4a7c15ull ^ ((uint64_t)k.start << 32 | k.end) * 0xc2b2ae3d27d4eb4full ^ k.cls;
            return (size_t)(h ^ h >> 29);
        }
    };
    std::unordered_map<Key, uint64_t, KeyHash> row_of;
    row_of.reserve((size_t)t->n * 2 + 16);
    for (uint64_t i = 0; i < t->n; ++i) row_of[Key{t->tid[i], t->start[i], t->end[i], cls(t->strand[i])}] = i;
    struct Ent { const char *s; uint32_t len; uint32_t count; };
    std::vector<std::vector<Ent>> per_row((size_t)t->n);
    // a junction of a single-cell library carries thousands of barcodes, times the shards: rows that grow past a few entries get an index
    // (barcode bytes -> entry) instead of the linear scan the common few-barcode rows keep
    typedef std::pair<const char *, uint32_t> Sv;
    struct SvHash {
        size_t operator()(const Sv &k) const {
            uint64_t h = 1469598103934665603ull;
            for (uint32_t i = 0; i < k.second; ++i) h = (h ^ (uint8_t)k.first[i]) * 1099511628211ull;
            return (size_t)h;
        }
    };
    struct SvEq { bool operator()(const Sv &a, const Sv &b) const { return a.second == b.second && !memcmp(a.first, b.first, a.second); } };
    typedef std::unordered_map<std::pair<const char *, uint32_t>, uint32_t, SvHash, SvEq> RowIndex;
    std::unordered_map<uint64_t, RowIndex> row_index;
    constexpr size_t kScanRows = 16;
    std::vector<uint64_t> order;
    for (int g = 0; g < n_parts; ++g) {
        const rgx_junction_table *p = parts[g];
        for (uint64_t i = 0; i < p->n; ++i) {
            auto it = row_of.find(Key{p->tid[i], p->start[i], p->end[i], cls(p->strand[i])});
            if (it == row_of.end()) return fail(err, errlen, RGX_ERR_ARG,
                "regtools_amd: a shard row is missing from the merged table (shard %d row %llu: tid %d %u-%u '%c'; merged rows %llu)\n", g,
                (unsigned long long)i, p->tid[i], p->start[i], p->end[i], p->strand[i], (unsigned long long)t->n);
            std::vector<Ent> &dst = per_row[(size_t)it->second];
            const uint64_t b = p->bc_row_begin[i], e = p->bc_row_begin[i + 1];
            order.assign(e - b, 0);
            for (uint64_t k = b; k < e; ++k) order[p->bc_insert_rank[k]] = k;                     // the shard's first-seen order
            for (uint64_t k : order) {
                const char *str = p->bc_text + p->bc_str_begin[k];
                const uint32_t len = (uint32_t)(p->bc_str_begin[k + 1] - p->bc_str_begin[k]);
                bool found = false;
                if (dst.size() > kScanRows) {
                    RowIndex &ix = row_index[it->second];
                    // (the row just outgrew the scan)
                    if (ix.empty()) for (uint32_t q = 0; q < dst.size(); ++q) ix.emplace(std::make_pair(dst[q].s, dst[q].len), q);
                    auto f = ix.find(std::make_pair(str, len));
                    if (f != ix.end()) { dst[f->second].count += p->bc_count[k]; found = true; }
                    else ix.emplace(std::make_pair(str, len), (uint32_t)dst.size());
                } else for (Ent &x : dst) if (x.len == len && !memcmp(x.s, str, len)) { x.count += p->bc_count[k]; found = true; break; }
                if (!found) dst.push_back(Ent{str, len, p->bc_count[k]});
            }
        }
        if (p->stream_ended) break;          // upstream reads nothing behind the point where the record stream ended
    }
    size_t n_pairs = 0, text_len = 0;
    for (auto &v : per_row) { n_pairs += v.size(); for (auto &x : v) text_len += x.len; }
    free(t->bc_row_begin); free(t->bc_count); free(t->bc_str_begin); free(t->bc_text); free(t->bc_insert_rank);
    t->bc_row_begin = (uint64_t *)calloc((size_t)t->n + 1, 8);
    t->bc_count = (uint32_t *)calloc(n_pairs + 1, 4);
    t->bc_str_begin = (uint64_t *)calloc(n_pairs + 1, 8);
    t->bc_text = (char *)malloc(text_len + 1);
    t->bc_insert_rank = (uint32_t *)calloc(n_pairs + 1, 4);
    uint64_t o = 0, pos = 0;
    for (uint64_t r = 0; r < t->n; ++r) {
        t->bc_row_begin[r] = o;
        const std::vector<Ent> &v = per_row[(size_t)r];
        std::unordered_map<std::string, int> m;                                                    // the reference's container
        for (size_t k = 0; k < v.size(); ++k) m.insert(std::pair<std::string, int>(std::string(v[k].s, v[k].len), (int)k));
        for (auto it = m.begin(); it != m.end(); ++it, ++o) {
            const Ent &x = v[(size_t)it->second];
            t->bc_count[o] = x.count; t->bc_insert_rank[o] = (uint32_t)it->second; t->bc_str_begin[o] = pos;
            memcpy(t->bc_text + pos, x.s, x.len); pos += x.len;
        }
    }
    t->bc_row_begin[t->n] = o; t->bc_str_begin[n_pairs] = pos;
    return RGX_OK;
}

extern "C" int rgx_table_merge(const rgx_junction_table *const *parts, int n_parts, uint32_t min_anchor, rgx_junction_table **out, char *err, size_t errlen) {
    if (n_parts <= 0 || !parts || !parts[0]) return fail(err, errlen, RGX_ERR_ARG, "regtools_amd: nothing to merge\n");
    struct Row { int32_t tid; uint32_t start, end, ts, te, cnt; uint64_t first, last; char strand; };
    auto cls = [](char c) { return c == '+' ? 0 : c == '-' ? 1 : 2; };
    const bool trace = getenv("REGTOOLS_AMD_TRACE") != nullptr; double tl = now_ms();
    auto mark = [&](const char *w) { if (trace) { double t = now_ms(); fprintf(stderr, "[rgx trace] merge %-14s %8.3f ms\n", w, t - tl); tl = t; } };
    std::vector<Row> rows;
    for (int g = 0; g < n_parts; ++g) {
        const rgx_junction_table *t = parts[g];
        for (uint64_t i = 0; i < t->n; ++i)
            rows.push_back({t->tid[i], t->start[i], t->end[i], t->thick_start[i], t->thick_end[i], t->read_count[i],
                            (uint64_t)g << 40 | t->first_seen[i], (uint64_t)g << 40 | t->last_seen[i], t->strand[i]});
        if (t->stream_ended) break;          // the record stream ended inside this shard: upstream reads nothing behind that point
    }
    mark("collect");
    std::stable_sort(rows.begin(), rows.end(), [&](const Row &a, const Row &b) {
        if (a.tid != b.tid) return a.tid < b.tid;
        if (a.start != b.start) return a.start < b.start;
        if (a.end != b.end) return a.end < b.end;
        return cls(a.strand) < cls(b.strand);
    });
    mark("key sort");
    std::vector<Row> uq;
    for (const Row &r : rows) {
        if (!uq.empty() && uq.back().tid == r.tid && uq.back().start == r.start && uq.back().end == r.end && cls(uq.back().strand) == cls(r.strand)) {
            Row &m = uq.back();
            m.cnt += r.cnt; m.ts = std::min(m.ts, r.ts); m.te = std::max(m.te, r.te);
            if (r.first < m.first) m.first = r.first;
            if (r.last > m.last) { m.last = r.last; m.strand = r.strand; }
        } else uq.push_back(r);
    }
    mark("reduce");
    std::vector<size_t> by_first(uq.size());
    for (size_t i = 0; i < uq.size(); ++i) by_first[i] = i;
    std::sort(by_first.begin(), by_first.end(), [&](size_t a, size_t b) { return uq[a].first < uq[b].first; });
    BamHeader h;
    for (int32_t i = 0; i < parts[0]->n_ref; ++i) { h.names.push_back(parts[0]->ref_name[i]); h.lens.push_back(parts[0]->ref_len[i]); }
    rgx_junction_table *t = table_alloc(h, uq.size());
    if (!t) return fail(err, errlen, RGX_ERR_ARG, "regtools_amd: no memory for the result table\n");
    for (size_t k = 0; k < by_first.size(); ++k) {
        const Row &r = uq[by_first[k]];
        const size_t i = by_first[k];
        t->tid[i] = r.tid; t->start[i] = r.start; t->end[i] = r.end; t->thick_start[i] = r.ts; t->thick_end[i] = r.te; t->read_count[i] = r.cnt;
        t->name_index[i] = k + 1; t->strand[i] = r.strand; t->first_seen[i] = r.first; t->last_seen[i] = r.last;
        t->left_ok[i] = (uint32_t)(r.start - r.ts) >= min_anchor; t->right_ok[i] = (uint32_t)(r.te - r.end) >= min_anchor;
    }
    mark("name+fill");
    host_sort_rows(t);
    mark("order sort");
    for (int g = 0; g < n_parts; ++g) {
        t->n_records += parts[g]->n_records; t->n_events += parts[g]->n_events; t->inflated_bytes += parts[g]->inflated_bytes;
        t->compressed_bytes = parts[g]->compressed_bytes; t->n_members += parts[g]->n_members;
    }
    bool all_bc = true;
    for (int g = 0; g < n_parts; ++g) if (!parts[g]->bc_row_begin) all_bc = false;
    if (all_bc) { const int rc = rgx_table_merge_barcodes(parts, n_parts, t, err, errlen); if (rc != RGX_OK) { rgx_table_free(t); return rc; } }
    *out = t;
    return RGX_OK;
}

// Peer access between two devices, both directions, once per pair and process (round 4): without it hipMemcpyPeer* between two GPUs is a bounce
// through host memory instead of a copy over xGMI.  Returns whether the pair is peer-accessible (a copy still works when it is not).
bool rgx_enable_peer(int a, int b) {
    if (a == b) return true;
    static std::mutex mu; static std::map<std::pair<int, int>, bool> done;
    std::lock_guard<std::mutex> lk(mu);
    const std::pair<int, int> key{std::min(a, b), std::max(a, b)};
    auto it = done.find(key);
    if (it != done.end()) return it->second;
    int prev = 0; (void)hipGetDevice(&prev);
    bool ok = true;
    for (int dir = 0; dir < 2; ++dir) {
        const int self = dir ? b : a, peer = dir ? a : b;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, self, peer) != hipSuccess || !can) { ok = false; continue; }
        if (hipSetDevice(self) != hipSuccess) { ok = false; continue; }
        const hipError_t e = hipDeviceEnablePeerAccess(peer, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) ok = false;
    }
    (void)hipSetDevice(prev);
    (void)hipGetLastError();                                        // ("already enabled" is not an error of anybody's launch)
    done[key] = ok;
    return ok;
}

// the rows of the context's last extraction, packed (48 bytes per row) into a buffer of the CONTEXT on its own stream, no host wait: *done is
// recorded behind the kernel, for the exchange stream of rgx_extract_multi to wait on (multi.cpp; not part of the C ABI)
int rgx_last_table_pack_async(rgx_ctx *c, const rgx_junction_table *t, void **d_packed, hipEvent_t *done, char *err, size_t errlen) {
    if (!c || !t || !d_packed || !done) return fail(err, errlen, RGX_ERR_ARG, "regtools_amd: bad arguments\n");
    if (!c->last_rows_valid || t->n != c->last_rows || t->n_records != c->last_records || t->n_events != c->last_events || t->inflated_bytes != c->last_bytes)
        return fail(err, errlen, RGX_ERR_ARG, "regtools_amd: the table is not the result of the last extraction on this context\n");
    HIP_ENTER(c->device);
    DevBuf &b = c->buf("rows_packed");
    HIP_TRY(b.ensure((size_t)std::max<uint64_t>(1, t->n) * RGX_PACKED_ROW_BYTES));
    if (!c->ev_packed) HIP_TRY(hipEventCreateWithFlags(&c->ev_packed, hipEventDisableTiming));
    if (t->n) launch_cols_to_packed(c->buf("rows_out").as<uint32_t>(), (uint32_t)t->n, b.as<uint32_t>(), c->stream);
    HIP_TRY(hipEventRecord(c->ev_packed, c->stream));
    *d_packed = b.p; *done = c->ev_packed;
    return RGX_OK;
}

// the rows of the context's last extraction, packed for the all-gather without leaving HBM
extern "C" int rgx_last_table_pack_device(rgx_ctx *c, const rgx_junction_table *t, void *d_dst, uint64_t cap_rows, char *err, size_t errlen) {
    if (!c || !t) return fail(err, errlen, RGX_ERR_ARG, "regtools_amd: bad arguments\n");
    if (!c->last_rows_valid || t->n != c->last_rows || t->n_records != c->last_records || t->n_events != c->last_events || t->inflated_bytes != c->last_bytes)
        return fail(err, errlen, RGX_ERR_ARG, "regtools_amd: the table is not the result of the last extraction on this context\n");
    if (!t->n) return RGX_OK;
    if (!d_dst || cap_rows < t->n) return fail(err, errlen, RGX_ERR_ARG, "regtools_amd: destination holds %llu rows, %llu needed\n",
        (unsigned long long)cap_rows, (unsigned long long)t->n);
    HIP_ENTER(c->device);
    launch_cols_to_packed(c->buf("rows_out").as<uint32_t>(), (uint32_t)t->n, (uint32_t *)d_dst, c->stream);
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RGX_OK;
}

// ---- multi-shard merge on the device (the gathered packed rows never leave HBM until the merged table is final) ----------------------
extern "C" int rgx_table_merge_device(rgx_ctx *c, const void *d_rows, uint64_t stride_rows, const uint64_t *part_rows, int n_parts, uint32_t min_anchor,
                                      const rgx_junction_table *names_from, rgx_junction_table **out, char *err, size_t errlen) {
    if (!c || !d_rows || !part_rows || n_parts <= 0 || n_parts > 255 || !names_from || !out) return fail(err, errlen, RGX_ERR_ARG,
        "regtools_amd: bad arguments\n");
    *out = nullptr;
    HIP_ENTER(c->device);
    hipStream_t st = c->stream;
    const bool trace = getenv("REGTOOLS_AMD_TRACE") != nullptr;
    double t_last = now_ms();
    auto mark = [&](const char *what) { if (trace) { (void)hipStreamSynchronize(st); double t = now_ms(); fprintf(stderr,
        "[rgx trace] merge: %-24s +%8.3f ms\n", what, t - t_last); t_last = t; } };
    std::vector<uint32_t> h_rows((size_t)n_parts), h_base((size_t)n_parts);
    uint64_t total = 0;
    for (int g = 0; g < n_parts; ++g) {
        if (part_rows[g] > stride_rows || part_rows[g] >= (1u << 24)) return fail(err, errlen, RGX_ERR_ARG,
            "regtools_amd: shard table too large for the device merge\n");
        h_rows[(size_t)g] = (uint32_t)part_rows[g]; h_base[(size_t)g] = (uint32_t)total; total += part_rows[g];
    }
    if (total >= (1ull << 31) || stride_rows * (uint64_t)n_parts >= (1ull << 32)) return fail(err, errlen, RGX_ERR_ARG,
        "regtools_amd: too many rows for the device merge\n");
    BamHeader h;
    for (int32_t i = 0; i < names_from->n_ref; ++i) { h.names.push_back(names_from->ref_name[i]); h.lens.push_back(names_from->ref_len[i]); }
    const uint32_t N = (uint32_t)total;
    if (!N) { *out = table_alloc(h, 0); return *out ? RGX_OK : RGX_ERR_DEVICE; }
    std::vector<uint32_t> rank_of_tid;
    chrom_string_ranks(h, rank_of_tid);
    uint32_t rk = 0; for (uint32_t r : rank_of_tid) rk = std::max(rk, r);

    DevBuf &b = c->buf("merge"), &sc = c->buf("scalars");
    HIP_TRY(sc.ensure(512));
    const size_t Nn = N, P = (size_t)n_parts, R = rank_of_tid.size();
    const size_t tmp_words = radix_tmp_words(N) + scan_tmp_words(N) + 64;
    HIP_TRY(b.ensure((Nn * (10 + 2 + 2 + 9 + 4 + 13) + 64 + 2 * P + R + tmp_words) * 4 + 1024));
    uint32_t *w = b.as<uint32_t>();
    MergeSoA m; m.tid = w; w += Nn; m.start = w; w += Nn; m.end = w; w += Nn; m.ts = w; w += Nn; m.te = w; w += Nn; m.count = w; w += Nn;
    m.cls = w; w += Nn; m.first = w; w += Nn; m.shard = w; w += Nn; m.strand = w; w += Nn;
    uint32_t *perm[2] = {w, w + Nn}; w += 2 * Nn;
    uint32_t *head = w; w += Nn; uint32_t *seg = w; w += Nn;
    MergeUnique u; u.tid = w; w += Nn; u.start = w; w += Nn; u.end = w; w += Nn; u.ts = w; w += Nn; u.te = w; w += Nn; u.count = w; w += Nn;
    u.first = w; w += Nn; u.last_shard = w; w += Nn; u.strand = w; w += Nn;
    uint32_t *name_rank = w; w += Nn; uint32_t *crank = w; w += Nn; uint32_t *uperm[2] = {w, w + Nn}; w += 2 * Nn;
    uint32_t *packed = w; w += Nn * 13 + 64;          // the merged table's device image (51 bytes per row + padding)
    uint32_t *d_rows_n = w; w += P; uint32_t *d_base = w; w += P; uint32_t *d_rank = w; w += R;
    uint32_t *tmp = w;
    uint32_t *d_total = sc.as<uint32_t>() + 70;
    HIP_TRY(hipMemcpyAsync(d_rows_n, h_rows.data(), P * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_base, h_base.data(), P * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_rank, rank_of_tid.data(), R * 4, hipMemcpyHostToDevice, st));
    launch_merge_unpack((const uint32_t *)d_rows, (uint32_t)stride_rows, (uint32_t)n_parts, d_rows_n, d_base, m, st);
    // stable LSD radix sort by (tid, start, end, class); rows of one key end up in shard order
    int pc = -1;
    auto sort_word = [&](const uint32_t *word, uint32_t nbits, uint32_t n, uint32_t **pp, int &cur) {
        for (uint32_t sh = 0; sh < nbits; sh += 8) {
            const uint32_t bits = std::min<uint32_t>(8, nbits - sh);
            const int nxt = cur < 0 ? 0 : cur ^ 1;
            launch_radix_pass(word, sh, bits, cur < 0 ? nullptr : pp[cur], pp[nxt], n, tmp, st);
            cur = nxt;
        }
    };
    sort_word(m.cls, 2, N, perm, pc);
    sort_word(m.end, 32, N, perm, pc);
    sort_word(m.start, 32, N, perm, pc);
    sort_word(m.tid, std::max<uint32_t>(1, bitlen((uint32_t)std::max<int32_t>(1, names_from->n_ref))), N, perm, pc);
    const uint32_t *sorted = perm[pc];
    mark("unpack + key sort");
    launch_merge_heads(m, sorted, N, head, st);
    launch_scan_u32(head, seg, N, d_total, tmp, st);
    uint32_t U = 0;
    HIP_TRY(hipMemcpyAsync(&U, d_total, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    launch_fill_u32(u.ts, 0xffffffffu, U, st); launch_fill_u32(u.first, 0xffffffffu, U, st);
    launch_fill_u32(u.te, 0u, U, st); launch_fill_u32(u.count, 0u, U, st); launch_fill_u32(u.last_shard, 0u, U, st);
    launch_merge_reduce(m, sorted, head, seg, N, u, st);
    // first-seen naming: rank by (first shard that has the key, the row's rank inside that shard)
    int upc = -1;
    sort_word(u.first, 32, U, uperm, upc);
    launch_merge_rank(uperm[upc], U, name_rank, st);
    // output order (junctions_extractor.h:117-140): chrom string rank, thick_start, thick_end, name
    launch_gather_u32(U, d_rank, u.tid, crank, st);
    upc = -1;
    sort_word(name_rank, std::max<uint32_t>(1, bitlen(U)), U, uperm, upc);
    sort_word(u.te, 32, U, uperm, upc);
    sort_word(u.ts, 32, U, uperm, upc);
    sort_word(crank, std::max<uint32_t>(1, bitlen(rk)), U, uperm, upc);
    launch_merge_table(u, uperm[upc], name_rank, U, min_anchor, (uint8_t *)packed, st);      // the packed area doubles as the table's device image
    mark("reduce + name + order");
    rgx_junction_table *t = table_alloc(h, U, /*zero=*/false, /*pinned=*/true);
    if (!t) { (void)hipStreamSynchronize(st); return fail(err, errlen, RGX_ERR_DEVICE, "regtools_amd: no memory for the result table\n"); }
    {
        hipError_t e_ = hipMemcpyAsync(((TableBox *)t)->block, packed, table_block_bytes(U), hipMemcpyDeviceToHost, st);
        if (e_ == hipSuccess) e_ = hipStreamSynchronize(st);
        if (e_ != hipSuccess) { rgx_table_free(t); return fail(err, errlen, RGX_ERR_DEVICE, "HIP error %s copying the merged table\n", hipGetErrorString(e_)); }
    }
    mark("rows to host table");
    *out = t;
    return RGX_OK;
}

