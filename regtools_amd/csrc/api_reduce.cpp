// api_reduce.cpp -- junction events -> the table: group-by of equal keys, first-seen naming, output order, the -b barcode lists, the finished
// rows.
#include "api_internal.h"

int reduce_events(rgx_ctx *c, EventSoA ev, uint32_t n_events, uint32_t group_bits, uint32_t ilen_bits, const uint32_t *rank_of_group_host,
                         uint32_t n_groups, HostRows &R, char *err, size_t errlen, bool view_only, RowMap *row_map, TableSink *sink, bool allow_preagg) {
    hipStream_t st = c->stream;
    uint32_t *d_sc = c->buf("scalars").as<uint32_t>();
    uint32_t *h_sc = (uint32_t *)c->pinned;
    DevBuf &b_sort = c->buf("sort"), &b_uni = c->buf("unique");
    c->last_rows_valid = false;            // the "rows_out" block is about to be overwritten
    uint32_t n_unique = 0;
    UniqueSoA u; memset(&u, 0, sizeof u);
    uint32_t *perm[2] = {nullptr, nullptr};
    uint32_t *final_perm = nullptr;
    uint32_t *chrom_rank_rows = nullptr;
    R = HostRows();
    if (n_events) {
        // Round 4: equal keys are grouped per tile of consecutive events first (k_preagg); what is sorted and reduced are the tiles' partial
        // rows.  Callers that need every event's row (the -b pass: row_map) keep the event form.
        const bool preagg = !row_map && allow_preagg;
        PartialSoA pr; memset(&pr, 0, sizeof pr);
        uint32_t *ev_flag = nullptr;           // preagg: one word per EVENT (first-seen flags, then their scan)
        EventSoA sev = ev;                     // what is sorted: the events, or the partial rows
        uint32_t n_s = n_events;
        if (preagg) {
            DevBuf &b_par = c->buf("partials");
            const size_t Ev = n_events;
            HIP_TRY(b_par.ensure(Ev * 4 * 9 + scan_tmp_words(n_events) * 4 + 512));
            uint32_t *q = b_par.as<uint32_t>();
            pr.tid = q; q += Ev; pr.start = q; q += Ev; pr.ilen_cls = q; q += Ev; pr.ts = q; q += Ev; pr.te = q; q += Ev;
            pr.count = q; q += Ev; pr.first = q; q += Ev; pr.last = q; q += Ev; ev_flag = q;
            HIP_TRY(hipMemsetAsync(d_sc + 7, 0, 4, st));
            launch_preagg(ev, n_events, pr, d_sc + 7, st);
            HIP_TRY(hipMemcpyAsync(h_sc + 7, d_sc + 7, 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            n_s = h_sc[7];
            memset(&sev, 0, sizeof sev);
            sev.tid = pr.tid; sev.start = pr.start; sev.ilen_cls = pr.ilen_cls; sev.ts = pr.ts; sev.te = pr.te;
        }
        const size_t E = n_s;
        const size_t rtmp = radix_tmp_words(n_s) + scan_tmp_words(n_s) + 64;
        HIP_TRY(b_sort.ensure(E * 4 * 6 + rtmp * 4 + 256));
        uint32_t *q = b_sort.as<uint32_t>();
        perm[0] = q; q += E; perm[1] = q; q += E;
        uint32_t *key[2]; key[0] = q; q += E; key[1] = q; q += E;    // the word being sorted on, carried along with the permutation
        uint32_t *head = q; q += E; uint32_t *seg_excl = q; q += E;
        uint32_t *tmp = q;
        int pc = -1;  // current permutation buffer (-1 = identity)
        // each word is gathered through the current permutation ONCE, then its 8-bit passes stream (key, permutation) pairs: with
        // 10^8 events the per-pass gathers of the plain form miss every cache (29 -> 12 ms on the long-read workload)
        auto sort_word = [&](const uint32_t *word, uint32_t nbits) {
            const uint32_t *kin = word;
            int kc = 0;
            if (pc >= 0) { launch_gather_u32(n_s, word, perm[pc], key[0], st); kin = key[0]; kc = 1; }
            for (uint32_t sh = 0; sh < nbits; sh += 8) {
                const uint32_t bits = std::min<uint32_t>(8, nbits - sh);
                const int nxt = pc < 0 ? 0 : pc ^ 1;
                launch_radix_pass_keyed(kin, key[kc], sh, bits, pc < 0 ? nullptr : perm[pc], perm[nxt], n_s, tmp, st);
                kin = key[kc]; kc ^= 1;
                pc = nxt;
            }
        };
        sort_word(sev.ilen_cls, ilen_bits);
        sort_word(sev.start, 32);
        sort_word(sev.tid, group_bits);
        const uint32_t *sorted = perm[pc];
        launch_heads(sev, sorted, n_s, head, st);
        launch_scan_u32(head, seg_excl, n_s, d_sc + 6, tmp, st);
        HIP_TRY(hipMemcpyAsync(h_sc + 6, d_sc + 6, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        n_unique = h_sc[6];

        const size_t U = n_unique;
        const size_t utmp = radix_tmp_words(n_unique) + 64;
        HIP_TRY(b_uni.ensure(U * 4 * 13 + U + utmp * 4 + 256));
        uint32_t *w = b_uni.as<uint32_t>();
        u.tid = w; w += U; u.start = w; w += U; u.end = w; w += U; u.ts_min = w; w += U; u.te_max = w; w += U; u.count = w; w += U;
        u.first_seen = w; w += U; u.last_seen = w; w += U; u.name_rank = w; w += U;
        uint32_t *head_pos = w; w += U; chrom_rank_rows = w; w += U;
        uint32_t *uperm[2]; uperm[0] = w; w += U; uperm[1] = w; w += U;
        uint32_t *utmp_p = w; w += utmp;
        u.strand = (uint8_t *)w;
        launch_fill_u32(u.ts_min, 0xffffffffu, U, st);
        launch_fill_u32(u.te_max, 0u, U, st);
        if (preagg) {
            launch_fill_u32(u.count, 0u, U, st);
            launch_fill_u32(u.first_seen, 0xffffffffu, U, st);
            launch_fill_u32(u.last_seen, 0u, U, st);
            launch_reduce_partials(pr, sorted, head, seg_excl, n_s, u, st);
            // first-seen naming (junctions_extractor.cc:152-157): rank of the key's first event among all keys -- flags over the EVENTS
            HIP_TRY(hipMemsetAsync(ev_flag, 0, (size_t)n_events * 4, st));
            launch_reduce_finish_partials(ev.strand, n_unique, u, ev_flag, st);
            launch_scan_u32(ev_flag, ev_flag, n_events, nullptr, ev_flag + n_events, st);
            launch_name_rank(n_unique, ev_flag, u, st);
        } else {
            launch_reduce(ev, sorted, head, seg_excl, n_events, u, head_pos, st);
            if (row_map) {
                DevBuf &b_map = c->buf("row_map");
                HIP_TRY(b_map.ensure((E + U) * 4 + 256));
                row_map->ev_urow = b_map.as<uint32_t>(); row_map->urow_pos = row_map->ev_urow + E;
                launch_event_urow(sorted, head, seg_excl, n_events, row_map->ev_urow, st);
            }
            // first-seen naming (junctions_extractor.cc:152-157): rank of the key's first event among all keys
            uint32_t *first_flag = head;       // reuse: head/seg_excl are dead after launch_reduce
            HIP_TRY(hipMemsetAsync(first_flag, 0, E * 4, st));
            launch_reduce_finish(ev, sorted, n_events, n_unique, head_pos, u, first_flag, st);
            launch_scan_u32(first_flag, seg_excl, n_events, nullptr, tmp, st);
            launch_name_rank(n_unique, seg_excl, u, st);
        }

        // output order (junctions_extractor.h:117-140): rank of the group (chrom string order), thick_start, thick_end, name
        uint32_t rk = 0;
        for (uint32_t i = 0; i < n_groups; ++i) rk = std::max(rk, rank_of_group_host[i]);
        DevBuf &b_rank = c->buf("rank");
        HIP_TRY(b_rank.ensure((size_t)n_groups * 4 + 64));
        // (outlives the asynchronous copy: every call ends with a sync of the stream)
        c->rank_stage.assign(rank_of_group_host, rank_of_group_host + n_groups);
        HIP_TRY(hipMemcpyAsync(b_rank.p, c->rank_stage.data(), (size_t)n_groups * 4, hipMemcpyHostToDevice, st));
        launch_gather_u32(n_unique, b_rank.as<uint32_t>(), u.tid, chrom_rank_rows, st);
        int upc = -1;
        auto usort = [&](const uint32_t *word, uint32_t nbits) {
            for (uint32_t sh = 0; sh < nbits; sh += 8) {
                const uint32_t bits = std::min<uint32_t>(8, nbits - sh);
                const int nxt = upc < 0 ? 0 : upc ^ 1;
                launch_radix_pass(word, sh, bits, upc < 0 ? nullptr : uperm[upc], uperm[nxt], n_unique, utmp_p, st);
                upc = nxt;
            }
        };
        usort(u.name_rank, std::max<uint32_t>(1, bitlen(n_unique)));
        usort(u.te_max, 32);
        usort(u.ts_min, 32);
        usort(chrom_rank_rows, std::max<uint32_t>(1, bitlen(rk)));
        final_perm = uperm[upc];
        if (row_map) launch_inverse_perm(final_perm, n_unique, row_map->urow_pos, st);
        if (!n_unique) HIP_TRY(hipStreamSynchronize(st));
    }

    if (n_unique) {
        // rows in final order: gathered on the device into one block of ten columns, ONE copy into pinned memory
        const size_t U = n_unique;
        DevBuf &b_out = c->buf("rows_out");
        HIP_TRY(b_out.ensure(U * 40 + 256));
        launch_rows_out(u, final_perm, n_unique, b_out.as<uint32_t>(), st);
        if (sink) {
            DevBuf &b_tab = c->buf("table_dev");
            const size_t bytes = table_block_bytes(U);
            HIP_TRY(b_tab.ensure(bytes + 256));
            launch_rows_table(u, final_perm, n_unique, sink->min_anchor, b_tab.as<uint8_t>(), st);
            // page-locking a block costs ~10 ms, a copy into pageable memory ~2 ms more than one into page-locked memory: the first table of a
            // context (a one-shot process has no second) is pageable, the loop that runs step after step gets its recycled page-locked block
            rgx_junction_table *t = table_alloc(*sink->hdr, U, /*zero=*/false, /*pinned=*/c->tables_made++ > 0);
            if (!t) { (void)hipStreamSynchronize(st); return fail(err, errlen, RGX_ERR_DEVICE, "regtools_amd: no memory for the result table\n"); }
            hipError_t e_ = hipMemcpyAsync(((TableBox *)t)->block, b_tab.p, bytes, hipMemcpyDeviceToHost, st);
            if (e_ == hipSuccess) e_ = hipStreamSynchronize(st);
            if (e_ != hipSuccess) { rgx_table_free(t); return fail(err, errlen, RGX_ERR_DEVICE, "HIP error %s copying the result table\n",
                hipGetErrorString(e_)); }
            sink->table = t; R.n = U;
            return RGX_OK;
        }
        if (U * 40 > c->pinned_rows_cap) {
            if (c->pinned_rows) (void)hipHostFree(c->pinned_rows);
            c->pinned_rows = nullptr; c->pinned_rows_cap = 0;
            const size_t want = U * 40 + U * 5 + 4096;
            HIP_TRY(hipHostMalloc(&c->pinned_rows, want, hipHostMallocDefault));
            c->pinned_rows_cap = want;
        }
        HIP_TRY(hipMemcpyAsync(c->pinned_rows, b_out.p, U * 40, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        const uint32_t *hp = (const uint32_t *)c->pinned_rows;
        R.cols = hp; R.n = U;
        if (view_only) return RGX_OK;
        auto col = [&](size_t k, std::vector<uint32_t> &dst) { dst.assign(hp + k * U, hp + (k + 1) * U); };
        col(0, R.group); col(1, R.start); col(2, R.end); col(3, R.ts); col(4, R.te); col(5, R.count); col(6, R.name_rank); col(7, R.first_seen); col(8,
            R.last_seen);
        R.strand.resize(U);
        for (size_t i = 0; i < U; ++i) R.strand[i] = (uint8_t)hp[9 * U + i];
        R.n = U;
    }
    return RGX_OK;
}

// ---- -b: barcode counts per junction ---------------------------------------------------------------------------------------------
// Second group-by, on (output row, barcode of the supporting read) -- barcode_kernels.hip.  The device returns one entry per distinct
// (junction, barcode) with its count and first event; the host puts each junction's distinct barcodes, in first-seen order, into the
// container the reference keeps them in (std::unordered_map<std::string,int>, junctions_extractor.h:58) and reads back its iteration
// order -- the order print_barcodes (h:99-111) writes.  Copies of that map (cc:202, :208, :214, :235) keep node order, bucket count and
// rehash state, so one map fed in first-seen order walks through the same states as the reference's per-read copies.
int barcode_rows(rgx_ctx *c, const Prep &P, const RowMap &rm, const rgx_extract_params *p, rgx_junction_table *t, char *err, size_t errlen) {
    hipStream_t st = c->stream;
    const double t0 = now_ms();
    const size_t E = P.n_events, U = t->n;
    t->bc_row_begin = (uint64_t *)calloc(U + 1, 8);
    if (!E) { t->bc_count = (uint32_t *)calloc(1, 4); t->bc_str_begin = (uint64_t *)calloc(1, 8); t->bc_text = (char *)calloc(1, 1);
        t->bc_insert_rank = (uint32_t *)calloc(1, 4); return RGX_OK; }
    uint32_t *d_sc = c->buf("scalars").as<uint32_t>();
    uint32_t *h_sc = (uint32_t *)c->pinned;
    DevBuf &b_bc = c->buf("barcodes");
    const size_t rtmp = radix_tmp_words((uint32_t)E) + scan_tmp_words((uint32_t)E) + 64;
    HIP_TRY(b_bc.ensure(E * (8 + 4 * 4 + 4 * 4 + 8 + 4 * 5) + rtmp * 4 + 512));
    uint8_t *q = b_bc.as<uint8_t>();
    BarcodeEv b;
    b.off = (uint64_t *)q; q += E * 8;
    uint64_t *pair_off = (uint64_t *)q; q += E * 8;
    b.len = (uint32_t *)q; q += E * 4; b.h_lo = (uint32_t *)q; q += E * 4; b.h_hi = (uint32_t *)q; q += E * 4; b.row = (uint32_t *)q; q += E * 4;
    uint32_t *perm[2]; perm[0] = (uint32_t *)q; q += E * 4; perm[1] = (uint32_t *)q; q += E * 4;
    uint32_t *head = (uint32_t *)q; q += E * 4; uint32_t *seg_excl = (uint32_t *)q; q += E * 4;
    uint32_t *pair_row = (uint32_t *)q; q += E * 4; uint32_t *pair_first = (uint32_t *)q; q += E * 4; uint32_t *pair_pos = (uint32_t *)q; q += E * 4;
    uint32_t *pair_len = (uint32_t *)q; q += E * 4; uint32_t *pair_count = (uint32_t *)q; q += E * 4;
    uint32_t *tmp = (uint32_t *)q;
    uint32_t *flags = d_sc + 72;
    HIP_TRY(hipMemsetAsync(flags, 0, 8, st));
    launch_bc_event_keys(P.arena, (uint32_t)E, P.ev.read, P.soa.rec_off, rm.ev_urow, rm.urow_pos, (uint8_t)p->barcode_tag[0], (uint8_t)p->barcode_tag[1], b,
        flags, st);
    int pc = -1;
    auto sort_word = [&](const uint32_t *word, uint32_t nbits) {
        for (uint32_t sh = 0; sh < nbits; sh += 8) {
            const int nxt = pc < 0 ? 0 : pc ^ 1;
            launch_radix_pass(word, sh, std::min<uint32_t>(8, nbits - sh), pc < 0 ? nullptr : perm[pc], perm[nxt], (uint32_t)E, tmp, st);
            pc = nxt;
        }
    };
    sort_word(b.h_lo, 32); sort_word(b.h_hi, 32);
    sort_word(b.row, std::max<uint32_t>(1, bitlen((uint32_t)std::max<size_t>(U, 1) - 1)));
    const uint32_t *sorted = perm[pc];
    launch_bc_heads(P.arena, b, sorted, (uint32_t)E, head, flags, st);
    launch_scan_u32(head, seg_excl, (uint32_t)E, d_sc + 74, tmp, st);
    HIP_TRY(hipMemcpyAsync(h_sc + 72, d_sc + 72, 12, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (h_sc[72]) return fail(err, errlen, RGX_ERR_FORMAT,
        "regtools_amd: the %c%c tag of an alignment is not a string (the reference dies on such input)\n\n", p->barcode_tag[0], p->barcode_tag[1]);
    if (h_sc[73]) return fail(err, errlen, RGX_ERR_DEVICE, "regtools_amd: two different barcodes of one junction share a 64-bit hash; not handled\n\n");
    const uint32_t n_pairs = h_sc[74];
    launch_bc_pairs(b, sorted, head, seg_excl, (uint32_t)E, pair_row, pair_first, pair_pos, pair_off, pair_len, st);
    launch_bc_counts(n_pairs, (uint32_t)E, pair_pos, pair_count, st);
    uint32_t *str_begin = head;                       // head / seg_excl are dead after launch_bc_pairs
    launch_scan_u32(pair_len, str_begin, n_pairs, d_sc + 75, tmp, st);
    HIP_TRY(hipMemcpyAsync(h_sc + 75, d_sc + 75, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const size_t text_len = h_sc[75];
    DevBuf &b_txt = c->buf("barcode_text");
    HIP_TRY(b_txt.ensure(text_len + 256));
    launch_bc_gather(P.arena, n_pairs, pair_off, pair_len, str_begin, b_txt.as<uint8_t>(), st);
    std::vector<uint32_t> h_row(n_pairs), h_first(n_pairs), h_count(n_pairs), h_begin(n_pairs), h_len(n_pairs);
    std::vector<char> h_text(text_len + 1);
    HIP_TRY(hipMemcpyAsync(h_row.data(), pair_row, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(h_first.data(), pair_first, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(h_count.data(), pair_count, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(h_begin.data(), str_begin, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(h_len.data(), pair_len, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, st));
    if (text_len) HIP_TRY(hipMemcpyAsync(h_text.data(), b_txt.p, text_len, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));

    // host: container order per junction.  Entries arrive sorted by row (then hash): each row's run is contiguous.
    t->bc_count = (uint32_t *)calloc((size_t)n_pairs + 1, 4);
    t->bc_str_begin = (uint64_t *)calloc((size_t)n_pairs + 1, 8);
    t->bc_text = (char *)malloc(text_len + 1);
    t->bc_insert_rank = (uint32_t *)calloc((size_t)n_pairs + 1, 4);
    std::vector<uint32_t> rank_of(n_pairs);
    std::vector<uint32_t> run_begin(U + 1, 0);
    for (uint32_t k = 0; k < n_pairs; ++k) run_begin[h_row[k] + 1]++;
    for (size_t r = 0; r < U; ++r) run_begin[r + 1] += run_begin[r];
    for (size_t r = 0; r <= U; ++r) t->bc_row_begin[r] = run_begin[r];
    const unsigned n_thr = (unsigned)std::max<size_t>(1, std::min<size_t>(16, U / 256));
    std::vector<std::thread> pool;
    auto work = [&](size_t r0, size_t r1) {
        std::vector<uint32_t> idx;
        for (size_t r = r0; r < r1; ++r) {
            const uint32_t k0 = run_begin[r], k1 = run_begin[r + 1];
            idx.resize(k1 - k0);
            for (uint32_t k = k0; k < k1; ++k) idx[k - k0] = k;
            std::sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) { return h_first[x] < h_first[y]; });      // first-seen order
            for (uint32_t q = 0; q < idx.size(); ++q) rank_of[idx[q]] = q;
            std::unordered_map<std::string, int> m;                                                               // the reference's container
            for (uint32_t k : idx) m.insert(std::pair<std::string, int>(std::string(h_text.data() + h_begin[k], h_len[k]), (int)k));
            uint32_t o = k0;
            for (auto it = m.begin(); it != m.end(); ++it, ++o) { t->bc_count[o] = h_count[(uint32_t)it->second]; t->bc_str_begin[o] = (uint64_t)it->second;
                /* entry id for now */ t->bc_insert_rank[o] = rank_of[(uint32_t)it->second]; }
        }
    };
    for (unsigned w = 0; w < n_thr; ++w) pool.emplace_back(work, U * w / n_thr, U * (w + 1) / n_thr);
    for (auto &th : pool) th.join();
    // lay the strings out in output order
    uint64_t pos = 0;
    for (uint32_t o = 0; o < n_pairs; ++o) {
        const uint32_t k = (uint32_t)t->bc_str_begin[o];
        t->bc_str_begin[o] = pos;
        memcpy(t->bc_text + pos, h_text.data() + h_begin[k], h_len[k]);
        pos += h_len[k];
    }
    t->bc_str_begin[n_pairs] = pos;
    t->ms_barcodes = now_ms() - t0;
    return RGX_OK;
}

void chrom_string_ranks(const BamHeader &hdr, std::vector<uint32_t> &rank_of_tid) {
    const size_t n = hdr.names.size();
    std::vector<uint32_t> order(n);
    rank_of_tid.assign(n ? n : 1, 0);
    for (size_t i = 0; i < n; ++i) order[i] = (uint32_t)i;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return hdr.names[a] < hdr.names[b]; });
    uint32_t rk = 0;
    for (size_t i = 0; i < n; ++i) {
        if (i > 0 && hdr.names[order[i]] != hdr.names[order[i - 1]]) ++rk;
        rank_of_tid[order[i]] = rk;
    }
}

int run_pipeline(rgx_ctx *c, const uint8_t *d_bam_in, const uint8_t *h_bam, size_t bam_len, const uint8_t *bai, size_t bai_len,
                        const rgx_extract_params *p, rgx_junction_table **out, char *err, size_t errlen, const SharedMembers *shared) {
    *out = nullptr;
    c->last_rows_valid = false;
    Prep P;
    int rc = prepare_events(c, d_bam_in, h_bam, bam_len, bai, bai_len, p, false, P, err, errlen, nullptr, true, false, shared);
    if (rc != RGX_OK) return rc;
    hipStream_t st = c->stream;
    const int32_t n_ref = (int32_t)P.hdr.names.size();
    std::vector<uint32_t> rank_of_tid;
    chrom_string_ranks(P.hdr, rank_of_tid);
    HostRows R;
    RowMap rm;
    TableSink sink; sink.hdr = &P.hdr; sink.min_anchor = p->min_anchor;
    rc = reduce_events(c, P.ev, P.n_events, std::max<uint32_t>(1, bitlen((uint32_t)std::max(n_ref - 1, 0))), std::min<uint32_t>(32, bitlen(p->max_intron) + 2),
                       rank_of_tid.data(), (uint32_t)std::max(n_ref, 1), R, err, errlen, /*view_only=*/true, p->barcodes ? &rm : nullptr, &sink);
    if (rc != RGX_OK) return rc;
    c->last_rows = R.n; c->last_records = P.n_iterated; c->last_events = P.n_events; c->last_bytes = P.total; c->last_rows_valid = true;
    HIP_TRY(hipEventRecord(c->ev[6], st));
    rgx_junction_table *t = sink.table ? sink.table : table_alloc(P.hdr, 0);
    if (!t) return fail(err, errlen, RGX_ERR_DEVICE, "regtools_amd: no memory for the result table\n");
    if (R.n >= 100000000u) host_sort_rows(t);   // names wider than 8 digits compare as strings upstream
    if (p->barcodes) {
        if (R.n >= 100000000u) { rgx_table_free(t); return fail(err, errlen, RGX_ERR_ARG, "regtools_amd: -b with 10^8 or more junctions is not supported\n"); }
        rc = barcode_rows(c, P, rm, p, t, err, errlen);
        if (rc != RGX_OK) { rgx_table_free(t); return rc; }
    }
    t->n_records = P.n_iterated;
    t->n_events = P.n_events; t->inflated_bytes = P.total; t->compressed_bytes = bam_len; t->n_members = P.n_range; t->framing_sweeps = P.framing_sweeps;
        t->stream_ended = P.stream_ended ? 1 : 0;
    float ms = 0;
    HIP_TRY(hipEventSynchronize(c->ev[6]));
    (void)hipEventElapsedTime(&ms, c->ev[0], c->ev[1]); t->ms_inflate = ms;
    t->ms_inflate_launch = 0;
    if (c->launch_timed && hipEventSynchronize(c->ev_launch[1]) == hipSuccess && hipEventElapsedTime(&ms, c->ev_launch[0],
        c->ev_launch[1]) == hipSuccess) t->ms_inflate_launch = ms;
    (void)hipEventElapsedTime(&ms, c->ev[2], c->ev[4]); t->ms_records = ms;
    (void)hipEventElapsedTime(&ms, c->ev[4], c->ev[5]); t->ms_scan = ms;
    (void)hipEventElapsedTime(&ms, c->ev[5], c->ev[6]); t->ms_reduce = ms;
    t->ms_total = now_ms() - P.t_begin;
    // (nothing reads the call's arena any more: one that lost its place to a challenger goes now)
    if (c->arena_retired) { c->arena_retired->release(); delete c->arena_retired; c->arena_retired = nullptr; }
    if (getenv("REGTOOLS_AMD_TRACE")) {
        fprintf(stderr, "[rgx trace] total %.3f ms; device buffers grown so far: %llu allocations, %.1f MB, %.3f ms\n", t->ms_total,
            (unsigned long long)g_alloc_stats.calls,
                (double)g_alloc_stats.bytes / 1e6, g_alloc_stats.ms);
    }
    *out = t;
    return RGX_OK;
}

