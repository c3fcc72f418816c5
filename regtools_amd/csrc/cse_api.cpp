// cse_api.cpp -- `cis-splice-effects identify / associate`, `variants annotate`, `junctions annotate` behind the C ABI.
#include "api_internal.h"

// `cis-splice-effects identify` on the device (rgx_ctx, DevBuf, prepare_events / reduce_events: api_internal.h).  Host code here parses text and assembles
// strings; every interval computation is a kernel.
#include <set>
#include <tuple>

#include "cse_host.h"

struct rgx_gtf {
    rgx_ctx *ctx = nullptr;
    GtfModel m;
    void *dev = nullptr;     // one allocation holding all flat arrays
    GtfView view{};
};

extern "C" void rgx_identify_params_default(rgx_identify_params *p) {
    memset(p, 0, sizeof *p);
    p->intronic_min = 2; p->exonic_min = 3; p->skip_single = 1; p->strandness = -1; p->strand_tag[0] = 'X'; p->strand_tag[1] = 'S';
    p->min_anchor = 8; p->min_intron = 70; p->max_intron = 500000;
}

// the host stages of a call share one pool of threads (set by the call: identify_run, the annotate commands)
static thread_local WorkerPool *tl_pool = nullptr;

// pooled = the tables go into a buffer of the CONTEXT (valid until its next call that loads an annotation: identify / associate / the annotate commands,
// which use the annotation inside the call) through one page-locked staging block and ONE copy -- eight synchronous copies out of pageable vectors into a
// fresh hipMalloc were 8 ms of config 4's `identify`, on its critical path behind the GTF thread.  rgx_gtf_load's annotation outlives the call: its own block.
static int gtf_upload(rgx_ctx *c, rgx_gtf *g, char *err, size_t errlen, bool pooled = false) {
    const GtfModel &m = g->m;
    const size_t T = m.tx_id.size(), E = m.es.size(), B = m.bin_key.size(), S = m.bin_start.size();
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_strand = 0, o_off = al(T), o_n = o_off + al(T * 4), o_es = o_n + al(T * 4), o_ee = o_es + al(E * 4), o_bk = o_ee + al(E * 4),
                 o_bt = o_bk + al(B * 8), o_bs = o_bt + al(B * 4), total = o_bs + al(S * 4) + 256;
    HIP_ENTER(c->device);
    uint8_t *d = nullptr;
    struct Piece { size_t off; const void *src; size_t bytes; };
    const Piece pieces[8] = {{o_strand, m.tx_strand.data(), T}, {o_off, m.tx_exon_off.data(), T * 4}, {o_n, m.tx_n_exons.data(), T * 4}, {o_es, m.es.data(),
        E * 4},
                             {o_ee, m.ee.data(), E * 4}, {o_bk, m.bin_key.data(), B * 8}, {o_bt, m.bin_tx.data(), B * 4}, {o_bs, m.bin_start.data(), S * 4}};
    if (pooled) {
        DevBuf &b = c->buf("gtf_tables");
        HIP_TRY(b.ensure(total));
        d = b.as<uint8_t>();
        g->dev = nullptr;                                          // (the context's: rgx_gtf_free leaves it alone)
        if (total > c->pinned_rows_cap) {
            if (c->pinned_rows) (void)hipHostFree(c->pinned_rows);
            c->pinned_rows = nullptr; c->pinned_rows_cap = 0;
            HIP_TRY(hipHostMalloc(&c->pinned_rows, total + total / 4, hipHostMallocDefault));
            c->pinned_rows_cap = total + total / 4;
        }
        uint8_t *stage = (uint8_t *)c->pinned_rows;
        // the pieces into the staging block, the large ones in slices on the stage's threads
        struct Slice { uint8_t *dst; const uint8_t *src; size_t n; };
        std::vector<Slice> sl;
        for (const Piece &q : pieces) for (size_t o = 0; o < q.bytes; o += (size_t)1 << 20) sl.push_back({stage + q.off + o, (const uint8_t *)q.src + o,
            std::min<size_t>((size_t)1 << 20, q.bytes - o)});
        if (tl_pool && sl.size() > 1) tl_pool->run(sl.size(), [&](size_t k) { memcpy(sl[k].dst, sl[k].src, sl[k].n); });
        else for (const Slice &x : sl) memcpy(x.dst, x.src, x.n);
        HIP_TRY(hipMemcpyAsync(d, stage, total - 256, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    } else {
        HIP_TRY(hipMalloc(&g->dev, total));
        d = (uint8_t *)g->dev;
        for (const Piece &q : pieces) if (q.bytes) HIP_TRY(hipMemcpy(d + q.off, q.src, q.bytes, hipMemcpyHostToDevice));
    }
    g->view.tx_strand = d + o_strand; g->view.tx_exon_off = (const uint32_t *)(d + o_off); g->view.tx_n_exons = (const uint32_t *)(d + o_n);
    g->view.es = (const uint32_t *)(d + o_es); g->view.ee = (const uint32_t *)(d + o_ee);
    g->view.bin_key = (const uint64_t *)(d + o_bk); g->view.bin_tx = (const uint32_t *)(d + o_bt); g->view.n_bin = (uint32_t)B;
    if (S) { g->view.bin_start = (const uint32_t *)(d + o_bs); g->view.bin_stride = m.bin_stride; }
    else { g->view.bin_start = nullptr; g->view.bin_stride = 0; }
    g->view.keep_single = 0;
    return RGX_OK;
}

extern "C" int rgx_gtf_load(rgx_ctx *ctx, const char *gtf_path, rgx_gtf **out, char *err, size_t errlen) {
    *out = nullptr;
    if (!ctx || !gtf_path) return fail(err, errlen, RGX_ERR_ARG, "regtools_amd: bad arguments\n");
    rgx_gtf *g = new rgx_gtf();
    g->ctx = ctx;
    std::string e = g->m.load(gtf_path);
    if (!e.empty()) { delete g; return fail(err, errlen, RGX_ERR_FORMAT, "%s", e.c_str()); }
    g->m.release_load_scratch();                              // (this annotation lives as long as its caller keeps it)
    int rc = gtf_upload(ctx, g, err, errlen);
    if (rc != RGX_OK) { rgx_gtf_free(g); return rc; }
    *out = g;
    return RGX_OK;
}
extern "C" void rgx_gtf_free(rgx_gtf *g) { if (!g) return; if (g->dev) (void)hipFree(g->dev); delete g; }
extern "C" int rgx_gtf_info(const rgx_gtf *g, uint32_t *n_tx, uint32_t *n_exons, uint32_t *n_chroms) {
    if (n_tx) *n_tx = (uint32_t)g->m.tx_id.size();
    if (n_exons) *n_exons = (uint32_t)g->m.es.size();
    if (n_chroms) *n_chroms = (uint32_t)g->m.chroms.size();
    return RGX_OK;
}
extern "C" int rgx_gtf_transcript_bin(const rgx_gtf *g, const char *transcript_id, uint32_t *bin) {
    auto it = std::lower_bound(g->m.tx_id.begin(), g->m.tx_id.end(), std::string(transcript_id));
    if (it == g->m.tx_id.end() || *it != transcript_id) return RGX_ERR_ARG;
    *bin = g->m.tx_bin[(size_t)(it - g->m.tx_id.begin())];
    return RGX_OK;
}
extern "C" const char *rgx_gtf_transcript_id(const rgx_gtf *g, uint32_t t) { return t < g->m.tx_id.size() ? g->m.tx_id[t].c_str() : ""; }

// f(t) for t in [0, T): on the pool the running `identify` call keeps for all its host stages (six of them start a dozen threads each otherwise:
// 2-3 ms of a 70 ms call), or on threads of its own where no such call is running
// (tl_pool: declared in front of gtf_upload)
static void run_tasks(size_t T, const std::function<void(size_t)> &f) {
    if (T <= 1) { for (size_t t = 0; t < T; ++t) f(t); return; }
    if (tl_pool) { tl_pool->run(T, f); return; }
    std::vector<std::thread> th;
    for (size_t t = 1; t < T; ++t) th.emplace_back(f, t);
    f(0);
    for (auto &x : th) x.join();
}

// ---- a10 ----------------------------------------------------------------------------------------------------------------
struct VariantHitsHost { std::vector<uint32_t> ces, cee, off, tx, ann, dist, last; };      // last: upstream's variant.score behind the walk (0xffffffff = "-1")

static int variant_windows(rgx_ctx *c, const rgx_gtf *g, const std::vector<int32_t> &chrom, const std::vector<uint32_t> &pos0, const VariantOpts &o,
                           VariantHitsHost &H, char *err, size_t errlen, uint64_t *exon_visits = nullptr) {
    const uint32_t n = (uint32_t)chrom.size();
    H = VariantHitsHost();
    H.off.assign((size_t)n + 1, 0);
    if (!n) return RGX_OK;
    hipStream_t st = c->stream;
    HIP_ENTER(c->device);
    DevBuf &b = c->buf("cse_variants"), &sc = c->buf("scalars");
    HIP_TRY(sc.ensure(512));
    const size_t N = n;
    HIP_TRY(b.ensure(N * 4 * 7 + scan_tmp_words(n) * 4 + 256));
    uint32_t *w = b.as<uint32_t>();
    int32_t *d_chrom = (int32_t *)w; w += N; uint32_t *d_pos = w; w += N; uint32_t *d_cnt = w; w += N; uint32_t *d_base = w; w += N;
    uint32_t *d_ces = w; w += N; uint32_t *d_cee = w; w += N; uint32_t *d_last = w; w += N; uint32_t *d_tmp = w;
    uint32_t *d_total = sc.as<uint32_t>() + 60;
    HIP_TRY(hipMemcpyAsync(d_chrom, chrom.data(), N * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_pos, pos0.data(), N * 4, hipMemcpyHostToDevice, st));
    unsigned long long *d_visits = (unsigned long long *)(sc.as<uint32_t>() + 64), h_visits = 0;
    HIP_TRY(hipMemsetAsync(d_visits, 0, 8, st));
    ktime_begin(c, 0);
    launch_variant_scan(false, g->view, n, d_chrom, d_pos, o, d_cnt, nullptr, d_ces, d_cee, nullptr, nullptr, d_visits, st, d_last);
    ktime_end(c);
    launch_scan_u32(d_cnt, d_base, n, d_total, d_tmp, st);
    uint32_t total = 0;
    HIP_TRY(hipMemcpyAsync(&total, d_total, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&h_visits, d_visits, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (exon_visits) *exon_visits = h_visits;
    H.ces.resize(N); H.cee.resize(N); H.last.resize(N);
    HIP_TRY(hipMemcpy(H.last.data(), d_last, N * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(H.ces.data(), d_ces, N * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(H.cee.data(), d_cee, N * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(H.off.data(), d_base, N * 4, hipMemcpyDeviceToHost));
    H.off[N] = total;
    if (total) {
        DevBuf &bh = c->buf("cse_variant_hits");
        HIP_TRY(bh.ensure((size_t)total * 12 + 256));
        uint32_t *d_tx = bh.as<uint32_t>(), *d_ad = d_tx + total;
        ktime_begin(c, 0);
        launch_variant_scan(true, g->view, n, d_chrom, d_pos, o, d_cnt, d_base, d_ces, d_cee, d_tx, d_ad, nullptr, st);
        ktime_end(c);
        std::vector<uint32_t> ad((size_t)total * 2);
        H.tx.resize(total);
        HIP_TRY(hipMemcpyAsync(H.tx.data(), d_tx, (size_t)total * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(ad.data(), d_ad, (size_t)total * 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        H.ann.resize(total); H.dist.resize(total);
        for (size_t k = 0; k < total; ++k) { H.ann[k] = ad[2 * k]; H.dist[k] = ad[2 * k + 1]; }
    }
    return RGX_OK;
}

extern "C" int rgx_variant_windows(rgx_ctx *ctx, const rgx_gtf *g, uint64_t n, const char *const *chrom, const uint32_t *pos0, uint32_t intronic_min,
                                   uint32_t exonic_min, int all_intronic, int all_exonic, int skip_single, rgx_variant_hits **out, char *err, size_t errlen) {
    *out = nullptr;
    std::vector<int32_t> ci((size_t)n); std::vector<uint32_t> ps(pos0, pos0 + n);
    for (uint64_t i = 0; i < n; ++i) ci[(size_t)i] = g->m.chrom_of(chrom[i]);
    VariantOpts o{intronic_min, exonic_min, all_intronic, all_exonic, skip_single};
    VariantHitsHost H;
    int rc = variant_windows(ctx, g, ci, ps, o, H, err, errlen);
    if (rc != RGX_OK) return rc;
    rgx_variant_hits *r = (rgx_variant_hits *)calloc(1, sizeof *r);
    auto dup = [](const std::vector<uint32_t> &v) { uint32_t *p = (uint32_t *)malloc((v.size() + 1) * 4); if (!v.empty()) memcpy(p, v.data(), v.size() * 4);
        return p; };
    r->n = n; r->cis_start = dup(H.ces); r->cis_end = dup(H.cee); r->hit_off = dup(H.off); r->hit_transcript = dup(H.tx); r->hit_annotation = dup(H.ann);
        r->hit_distance = dup(H.dist);
    *out = r;
    return RGX_OK;
}
extern "C" void rgx_variant_hits_free(rgx_variant_hits *h) {
    if (!h) return;
    free(h->cis_start); free(h->cis_end); free(h->hit_off); free(h->hit_transcript); free(h->hit_annotation); free(h->hit_distance); free(h);
}

// ---- a11 ----------------------------------------------------------------------------------------------------------------
struct JunctionAnnotHost { std::vector<uint32_t> flags, n_acc, n_exo, n_don, tx_off, tx; };

static int annotate_junctions(rgx_ctx *c, const rgx_gtf *g, const std::vector<int32_t> &chrom, const std::vector<uint32_t> &js, const std::vector<uint32_t> &je,
                              const std::vector<uint8_t> &strand, JunctionAnnotHost &A, char *err, size_t errlen, uint64_t *exon_visits = nullptr,
                                  bool keep_single = false) {
    const uint32_t n = (uint32_t)chrom.size();
    GtfView view = g->view; view.keep_single = keep_single ? 1u : 0u;          // (`junctions annotate -S` only)
    A = JunctionAnnotHost();
    A.tx_off.assign((size_t)n + 1, 0);
    if (!n) return RGX_OK;
    hipStream_t st = c->stream;
    HIP_ENTER(c->device);
    DevBuf &b = c->buf("cse_junctions"), &sc = c->buf("scalars");
    HIP_TRY(sc.ensure(512));
    const size_t N = n;
    HIP_TRY(b.ensure(N * 4 * 7 + N + scan_tmp_words(n) * 4 + 512));
    uint32_t *w = b.as<uint32_t>();
    int32_t *d_chrom = (int32_t *)w; w += N; uint32_t *d_js = w; w += N; uint32_t *d_je = w; w += N; uint32_t *d_cnt = w; w += N; uint32_t *d_base = w; w += N;
    uint32_t *d_flags = w; w += N; uint32_t *d_visit_each = w; w += N; uint32_t *d_tmp = w; w += scan_tmp_words(n) + 8; uint8_t *d_strand = (uint8_t *)w;
    uint32_t *d_total = sc.as<uint32_t>() + 61;
    HIP_TRY(hipMemcpyAsync(d_chrom, chrom.data(), N * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_js, js.data(), N * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_je, je.data(), N * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_strand, strand.data(), N, hipMemcpyHostToDevice, st));
    unsigned long long *d_visits = (unsigned long long *)(sc.as<uint32_t>() + 66), h_visits = 0;
    HIP_TRY(hipMemsetAsync(d_visits, 0, 8, st));
    ktime_begin(c, 1);
    launch_junction_scan(false, view, n, d_chrom, d_js, d_je, d_strand, d_cnt, nullptr, d_flags, nullptr, nullptr, nullptr, d_visits, d_visit_each, st);
    ktime_end(c);
    launch_scan_u32(d_cnt, d_base, n, d_total, d_tmp, st);
    uint32_t total = 0;
    HIP_TRY(hipMemcpyAsync(&total, d_total, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&h_visits, d_visits, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    std::vector<uint32_t> off((size_t)n + 1), kind(total), ia(total), ib(total);
    A.flags.resize(N);
    HIP_TRY(hipMemcpy(A.flags.data(), d_flags, N * 4, hipMemcpyDeviceToHost));
    // SURVEY 8d's E_j: the lane form adds into one counter, the wave form writes one count per junction
    if (exon_visits) {
        HIP_TRY(hipMemcpy(off.data(), d_visit_each, N * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < N; ++i) h_visits += off[i];
        *exon_visits = h_visits;
    }
    HIP_TRY(hipMemcpy(off.data(), d_base, N * 4, hipMemcpyDeviceToHost));
    off[N] = total;
    if (total) {
        DevBuf &bi = c->buf("cse_junction_items");
        HIP_TRY(bi.ensure((size_t)total * 12 + 256));
        uint32_t *d_k = bi.as<uint32_t>(), *d_a = d_k + total, *d_b = d_a + total;
        ktime_begin(c, 1);
        launch_junction_scan(true, view, n, d_chrom, d_js, d_je, d_strand, d_cnt, d_base, d_flags, d_k, d_a, d_b, nullptr, nullptr, st);
        ktime_end(c);
        HIP_TRY(hipMemcpyAsync(kind.data(), d_k, (size_t)total * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(ia.data(), d_a, (size_t)total * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(ib.data(), d_b, (size_t)total * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    // the reference keeps sets (junctions_annotator.h:41-45): unique skipped elements by coordinate, transcripts by id -- here small
    // vectors, sorted and made unique, ranges of junctions on the host's threads (std::set per junction: 20 of config 4's 280 ms)
    A.n_acc.resize(N); A.n_exo.resize(N); A.n_don.resize(N);
    const size_t T = N < 4096 ? 1 : usable_threads(16);
    std::vector<std::vector<uint32_t>> tx_part(T);
    std::vector<uint32_t> tx_cnt(N);
    auto work = [&](size_t t) {
        std::vector<uint32_t> acc, don, txs; std::vector<uint64_t> exo;
        auto uniq = [](auto &v) { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); return (uint32_t)v.size(); };
        for (size_t i = N * t / T; i < N * (t + 1) / T; ++i) {
            acc.clear(); don.clear(); txs.clear(); exo.clear();
            for (uint32_t k = off[i]; k < off[i + 1]; ++k) {
                if (kind[k] == ITEM_TX) txs.push_back(ia[k]);
                else if (kind[k] == ITEM_EXON) exo.push_back((uint64_t)ia[k] << 32 | ib[k]);
                else if (kind[k] == ITEM_DONOR) don.push_back(ia[k]);
                else acc.push_back(ia[k]);
            }
            A.n_acc[i] = uniq(acc); A.n_exo[i] = uniq(exo); A.n_don[i] = uniq(don);
            tx_cnt[i] = uniq(txs);                                    // transcript indices ascend with transcript ids
            tx_part[t].insert(tx_part[t].end(), txs.begin(), txs.end());
        }
    };
    run_tasks(T, work);
    for (size_t i = 0; i < N; ++i) A.tx_off[i + 1] = A.tx_off[i] + tx_cnt[i];
    A.tx.reserve(A.tx_off[N]);
    for (size_t t = 0; t < T; ++t) A.tx.insert(A.tx.end(), tx_part[t].begin(), tx_part[t].end());
    return RGX_OK;
}

extern "C" int rgx_annotate_junctions(rgx_ctx *ctx, const rgx_gtf *g, uint64_t n, const char *const *chrom, const uint32_t *start, const uint32_t *end1,
                                      const char *strand, rgx_junction_annot **out, char *err, size_t errlen) {
    *out = nullptr;
    std::vector<int32_t> ci((size_t)n); std::vector<uint32_t> js(start, start + n), je(end1, end1 + n); std::vector<uint8_t> sd((size_t)n);
    for (uint64_t i = 0; i < n; ++i) { ci[(size_t)i] = g->m.chrom_of(chrom[i]); sd[(size_t)i] = (uint8_t)strand[i]; }
    JunctionAnnotHost A;
    int rc = annotate_junctions(ctx, g, ci, js, je, sd, A, err, errlen);
    if (rc != RGX_OK) return rc;
    rgx_junction_annot *r = (rgx_junction_annot *)calloc(1, sizeof *r);
    auto dup = [](const std::vector<uint32_t> &v) { uint32_t *p = (uint32_t *)malloc((v.size() + 1) * 4); if (!v.empty()) memcpy(p, v.data(), v.size() * 4);
        return p; };
    r->n = n; r->flags = dup(A.flags); r->n_acceptors_skipped = dup(A.n_acc); r->n_exons_skipped = dup(A.n_exo); r->n_donors_skipped = dup(A.n_don);
    r->tx_off = dup(A.tx_off); r->tx = dup(A.tx);
    *out = r;
    return RGX_OK;
}
extern "C" void rgx_junction_annot_free(rgx_junction_annot *a) {
    if (!a) return;
    free(a->flags); free(a->n_acceptors_skipped); free(a->n_exons_skipped); free(a->n_donors_skipped); free(a->tx_off); free(a->tx); free(a);
}

// ---- a9: window join -------------------------------------------------------------------------------------------------------
// rows of every window in the order get_all_junctions would give for that window's extraction (thick_start, thick_end, name)
static int window_join(rgx_ctx *c, const Prep &P, const std::vector<int32_t> &w_tid, const std::vector<int32_t> &w_beg, const std::vector<int32_t> &w_end,
                       uint32_t ilen_bits, HostRows &R, uint64_t &n_pairs, char *err, size_t errlen) {
    R = HostRows(); n_pairs = 0;
    const uint32_t W = (uint32_t)w_tid.size();
    if (!W || !P.n_events) return RGX_OK;
    hipStream_t st = c->stream;
    DevBuf &b = c->buf("cse_windows"), &sc = c->buf("scalars");
    const size_t Wn = W;
    const size_t Sn = Wn * kWinSlices;                               // count / base: one entry per (window, slice)
    HIP_TRY(b.ensure(Wn * 4 * 5 + Sn * 4 * 2 + scan_tmp_words((uint32_t)Sn) * 4 + 256));
    uint32_t *w = b.as<uint32_t>();
    int32_t *d_tid = (int32_t *)w; w += Wn; int32_t *d_beg = (int32_t *)w; w += Wn; int32_t *d_end = (int32_t *)w; w += Wn;
    uint32_t *d_lo = w; w += Wn; uint32_t *d_hi = w; w += Wn; uint32_t *d_cnt = w; w += Sn; uint32_t *d_base = w; w += Sn; uint32_t *d_tmp = w;
    uint32_t *d_span = sc.as<uint32_t>() + 62, *d_total = sc.as<uint32_t>() + 63;
    HIP_TRY(hipMemcpyAsync(d_tid, w_tid.data(), Wn * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_beg, w_beg.data(), Wn * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_end, w_end.data(), Wn * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(d_span, 0, 4, st));
    launch_max_span(P.ev, P.n_events, d_span, st);
    ktime_begin(c, 2);
    launch_window_pairs(false, P.ev, P.n_events, W, d_tid, d_beg, d_end, d_span, d_lo, d_hi, d_cnt, nullptr, nullptr, nullptr, st);
    ktime_end(c);
    // The (window, event) pairs are materialised in batches of whole windows: a VCF dense in splice-region variants of highly
    // expressed genes multiplies events by windows, and neither a 32-bit pair count nor HBM should be the limit of that.
    std::vector<uint32_t> h_cnt(Wn);
    {
        std::vector<uint32_t> h_slices(Sn);
        HIP_TRY(hipMemcpyAsync(h_slices.data(), d_cnt, Sn * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (size_t k = 0; k < Wn; ++k) { uint64_t c = 0; for (uint32_t q = 0; q < kWinSlices; ++q) c += h_slices[k * kWinSlices + q];
            h_cnt[k] = c > 0xffffffffull ? 0xffffffffu : (uint32_t)c; }
    }
    static const uint64_t kPairBatch = getenv("REGTOOLS_AMD_PAIR_BATCH") ? strtoull(getenv("REGTOOLS_AMD_PAIR_BATCH"), nullptr, 10) : (1ull << 26);
    for (uint32_t w0 = 0; w0 < W;) {
        uint64_t total64 = h_cnt[w0];
        uint32_t w1 = w0 + 1;
        while (w1 < W && total64 + h_cnt[w1] <= kPairBatch) total64 += h_cnt[w1++];
        if (total64 >= (1ull << 31)) return fail(err, errlen, RGX_ERR_ARG,
            "regtools_amd: one variant window holds %llu junction-supporting reads; more than the join handles\n", (unsigned long long)total64);
        const uint32_t nw = w1 - w0, total = (uint32_t)total64;
        n_pairs += total;
        if (total) {
            launch_scan_u32(d_cnt + (size_t)w0 * kWinSlices, d_base + (size_t)w0 * kWinSlices, nw * kWinSlices, d_total, d_tmp, st);
            DevBuf &bp = c->buf("cse_pairs");
            const size_t Pn = total;
            HIP_TRY(bp.ensure(Pn * 4 * 7 + Pn + 256));
            uint32_t *q = bp.as<uint32_t>();
            uint32_t *pair_ev = q; q += Pn; uint32_t *pair_win = q; q += Pn;
            EventSoA pe; memset(&pe, 0, sizeof pe);
            pe.tid = q; q += Pn; pe.start = q; q += Pn; pe.ilen_cls = q; q += Pn; pe.ts = q; q += Pn; pe.te = q; q += Pn; pe.strand = (uint8_t *)q;
            ktime_begin(c, 2);
            launch_window_pairs(true, P.ev, P.n_events, nw, d_tid + w0, d_beg + w0, d_end + w0, d_span, d_lo + w0, d_hi + w0,
                d_cnt + (size_t)w0 * kWinSlices, d_base + (size_t)w0 * kWinSlices, pair_ev, pair_win, st);
            ktime_end(c);
            launch_pair_gather(P.ev, pair_ev, pair_win, total, pe, st);
            std::vector<uint32_t> ident(nw);
            for (uint32_t i = 0; i < nw; ++i) ident[i] = i;
            HostRows B;
            int rc = reduce_events(c, pe, total, std::max<uint32_t>(1, bitlen(nw - 1)), ilen_bits, ident.data(), nw, B, err, errlen, false, nullptr, nullptr,
                /*allow_preagg=*/false);
            if (rc != RGX_OK) return rc;
            // rows of the batch, window indices made absolute; name ranks stay batch-local (callers only use them inside a window)
            for (uint32_t &g : B.group) g += w0;
            auto app = [](std::vector<uint32_t> &dst, const std::vector<uint32_t> &src) { dst.insert(dst.end(), src.begin(), src.end()); };
            app(R.group, B.group); app(R.start, B.start); app(R.end, B.end); app(R.ts, B.ts); app(R.te, B.te); app(R.count, B.count);
            app(R.name_rank, B.name_rank); app(R.first_seen, B.first_seen); app(R.last_seen, B.last_seen);
            R.strand.insert(R.strand.end(), B.strand.begin(), B.strand.end());
            R.n += B.n;
        }
        w0 = w1;
    }
    return RGX_OK;
}

// `identify` on a file whose record stream ENDED somewhere (a member that does not inflate, an unreadable record): upstream reads every variant's window
// through
// the index on its own (identifier.cc:288-290) -- also the windows BEHIND the damage, which one pass over the file never reaches.  Here, for such a file only:
// one
// region extraction per window from the file's bytes in HBM (what `junctions extract -r` makes of a damaged file: the iterator's chunks are seeks of their
// own), the
// windows' events put together as window_join's pairs are, the same group-by behind them.  *w_abort (SIZE_MAX = none): the first window that reads a read
// bam_aux_get
// abort()s on (Prep::odd_aux); the windows behind it are not read.
static int window_join_by_seeks(rgx_ctx *c, const uint8_t *d_file, size_t bam_len, const uint8_t *bai, size_t bai_len, const rgx_extract_params &ep0,
                                const std::vector<std::string> &w_region, uint32_t ilen_bits, HostRows &R, uint64_t &n_pairs, size_t &w_abort, char *err,
                                    size_t errlen) {
    R = HostRows(); n_pairs = 0; w_abort = SIZE_MAX;
    const size_t W = w_region.size();
    hipStream_t st = c->stream;
    std::vector<uint32_t> h_col[6];                               // window (batch-local), start, ilen_cls, ts, te; [5] unused
    std::vector<uint8_t> h_strand;
    size_t w0 = 0;
    auto flush = [&](size_t w1) -> int {
        const size_t total = h_col[0].size(), nw = w1 - w0;
        if (total && nw) {
            if (total >= (1ull << 31)) return fail(err, errlen, RGX_ERR_ARG,
                "regtools_amd: %zu junction-supporting reads in one batch of windows; more than the join handles\n", total);
            DevBuf &bp = c->buf("cse_pairs");
            HIP_TRY(bp.ensure(total * 4 * 7 + total + 256));
            uint32_t *q = bp.as<uint32_t>() + 2 * total;          // (window_join's layout: the pair lists' place stays empty)
            EventSoA pe; memset(&pe, 0, sizeof pe);
            pe.tid = q; q += total; pe.start = q; q += total; pe.ilen_cls = q; q += total; pe.ts = q; q += total; pe.te = q; q += total;
                pe.strand = (uint8_t *)q;
            uint32_t *dst[5] = {pe.tid, pe.start, pe.ilen_cls, pe.ts, pe.te};
            for (int k = 0; k < 5; ++k) HIP_TRY(hipMemcpyAsync(dst[k], h_col[k].data(), total * 4, hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(pe.strand, h_strand.data(), total, hipMemcpyHostToDevice, st));
            HIP_TRY(hipStreamSynchronize(st));
            std::vector<uint32_t> ident(nw);
            for (size_t i = 0; i < nw; ++i) ident[i] = (uint32_t)i;
            HostRows B;
            const int rc = reduce_events(c, pe, (uint32_t)total, std::max<uint32_t>(1, bitlen((uint32_t)nw - 1)), ilen_bits, ident.data(), (uint32_t)nw, B,
                err, errlen, false,
                nullptr, nullptr, /*allow_preagg=*/false);
            if (rc != RGX_OK) return rc;
            for (uint32_t &g : B.group) g += (uint32_t)w0;
            auto app = [](std::vector<uint32_t> &d, const std::vector<uint32_t> &s2) { d.insert(d.end(), s2.begin(), s2.end()); };
            app(R.group, B.group); app(R.start, B.start); app(R.end, B.end); app(R.ts, B.ts); app(R.te, B.te); app(R.count, B.count);
            app(R.name_rank, B.name_rank); app(R.first_seen, B.first_seen); app(R.last_seen, B.last_seen);
            R.strand.insert(R.strand.end(), B.strand.begin(), B.strand.end());
            R.n += B.n;
            n_pairs += total;
        }
        for (auto &v : h_col) v.clear();
        h_strand.clear();
        w0 = w1;
        return RGX_OK;
    };
    for (size_t w = 0; w < W; ++w) {
        rgx_extract_params q = ep0;
        q.region = w_region[w].c_str(); q.shard = 0; q.n_shards = 1;
        Prep Pw;
        const int rc = prepare_events(c, d_file, nullptr, bam_len, bai, bai_len, &q, true, Pw, err, errlen);
        if (rc != RGX_OK) return rc;
        if (!Pw.odd_aux.empty()) { w_abort = w; break; }
        const size_t n = Pw.n_events;
        if (n) {
            const size_t at = h_col[0].size();
            for (int k = 0; k < 5; ++k) h_col[k].resize(at + n);
            h_strand.resize(at + n);
            const uint32_t *src[5] = {nullptr, Pw.ev.start, Pw.ev.ilen_cls, Pw.ev.ts, Pw.ev.te};
            for (int k = 1; k < 5; ++k) HIP_TRY(hipMemcpyAsync(h_col[k].data() + at, src[k], n * 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(h_strand.data() + at, Pw.ev.strand, n, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            std::fill(h_col[0].begin() + (ptrdiff_t)at, h_col[0].end(), (uint32_t)(w - w0));
        }
        if (h_col[0].size() >= (1u << 22)) { const int rc2 = flush(w + 1); if (rc2 != RGX_OK) return rc2; }
    }
    return flush(w_abort == SIZE_MAX ? W : w_abort);
}

extern "C" int rgx_window_join(rgx_ctx *c, const char *bam_path, const rgx_extract_params *p, uint64_t n_windows, const char *const *chrom, const int32_t *beg,
                               const int32_t *end, rgx_window_rows **out, char *err, size_t errlen) {
    if (!c || !bam_path || !p || !out || (n_windows && (!chrom || !beg || !end))) return fail(err, errlen, RGX_ERR_ARG, "regtools_amd: bad arguments\n");
    *out = nullptr;
    FileBytes bam; std::vector<uint8_t> bai;
    if (!bam.open(bam_path)) return fail(err, errlen, RGX_ERR_OPEN, "%s", kMsgOpen);
    std::string idx;
    if (find_index(bam_path, idx) != 0 || !read_index(idx, bai)) return fail(err, errlen, RGX_ERR_INDEX, "%s", kMsgIndex);
    rgx_extract_params ep = *p;
    ep.region = "."; ep.shard = 0; ep.n_shards = 1;
    Prep P;
    int rc = prepare_events(c, nullptr, bam.data(), bam.size(), bai.data(), bai.size(), &ep, true, P, err, errlen);
    if (rc != RGX_OK) return rc;
    std::vector<int32_t> w_tid((size_t)n_windows), w_beg(beg, beg + n_windows), w_end(end, end + n_windows);
    for (uint64_t w = 0; w < n_windows; ++w) {
        int32_t tid = -1;
        for (size_t t = 0; t < P.hdr.names.size(); ++t) if (P.hdr.names[t] == chrom[w]) { tid = (int32_t)t; break; }
        if (tid < 0 || w_end[(size_t)w] < w_beg[(size_t)w]) return fail(err, errlen, RGX_ERR_REGION, "%s", kMsgRegion);
        w_tid[(size_t)w] = tid;
    }
    HostRows R; uint64_t n_pairs = 0;
    rc = window_join(c, P, w_tid, w_beg, w_end, std::min<uint32_t>(32, bitlen(ep.max_intron) + 2), R, n_pairs, err, errlen);
    if (rc != RGX_OK) return rc;
    rgx_window_rows *r = (rgx_window_rows *)calloc(1, sizeof *r);
    const size_t n = R.n;
    r->n = n;
    auto col = [&](const std::vector<uint32_t> &v) { uint32_t *q = (uint32_t *)malloc((n + 1) * 4); if (n) memcpy(q, v.data(), n * 4); return q; };
    r->window = col(R.group); r->start = col(R.start); r->end = col(R.end); r->thick_start = col(R.ts); r->thick_end = col(R.te); r->read_count = col(R.count);
    r->name_index = (uint32_t *)malloc((n + 1) * 4); r->strand = (char *)malloc(n + 1);
    for (size_t i = 0; i < n; ++i) r->strand[i] = (char)R.strand[i];
    // names restart in every window: rank of the row's (global, window-major) first-seen rank among the rows of its window
    for (size_t lo = 0; lo < n;) {
        size_t hi = lo; while (hi < n && R.group[hi] == R.group[lo]) ++hi;
        std::vector<std::pair<uint32_t, size_t>> order;
        for (size_t i = lo; i < hi; ++i) order.push_back({R.name_rank[i], i});
        std::sort(order.begin(), order.end());
        for (size_t k = 0; k < order.size(); ++k) r->name_index[order[k].second] = (uint32_t)k + 1;
        lo = hi;
    }
    *out = r;
    return RGX_OK;
}
extern "C" void rgx_window_rows_free(rgx_window_rows *r) {
    if (!r) return;
    free(r->window); free(r->start); free(r->end); free(r->thick_start); free(r->thick_end); free(r->read_count); free(r->name_index); free(r->strand); free(r);
}

// ---- shared stages of the four commands ------------------------------------------------------------------------------------
struct VStr { std::string genes, transcripts, distances, annotations; };
struct VariantStage {
    VcfText vcf;
    VariantHitsHost H;
    // comma strings in visitation order (variants_annotator.cc:479-506), one entry per splice relevant record (the others are written as "NA" x4)
    std::vector<VStr> vstr;
    std::vector<uint32_t> vstr_of;     // record -> its entry of vstr, UINT32_MAX when not splice relevant
    std::vector<size_t> relevant;      // indices into vcf.recs
};

// a10 for a whole VCF (V.vcf loaded by the caller, usually on a side thread): every record against the annotation on the device, strings on the host
static int variant_scan_stage(rgx_ctx *c, const rgx_gtf *g, const VariantOpts &vo, VariantStage &V, uint64_t *exon_visits, char *err, size_t errlen) {
    const size_t n = V.vcf.recs.size();
    std::vector<int32_t> vchrom(n); std::vector<uint32_t> vpos(n);
    {   // (records of a VCF come contig by contig: one table lookup per run of equal names)
        const std::string *last = nullptr; int32_t last_c = -1;
        for (size_t i = 0; i < n; ++i) {
            const std::string &cn = V.vcf.recs[i].chrom;
            if (!last || *last != cn) { last = &cn; last_c = g->m.chrom_of(cn); }
            vchrom[i] = last_c; vpos[i] = V.vcf.recs[i].pos0;
        }
    }
    int rc = variant_windows(c, g, vchrom, vpos, vo, V.H, err, errlen, exon_visits);
    if (rc != RGX_OK) return rc;
    static const char *kAnn[] = {"non_splice_region", "exonic", "intronic", "splicing_exonic", "splicing_intronic"};
    // strings only for the splice relevant records (a few per cent of a VCF), built by threads over ranges of them
    V.relevant.clear();
    for (size_t i = 0; i < n; ++i) if (V.H.off[i + 1] != V.H.off[i]) V.relevant.push_back(i);
    const size_t R = V.relevant.size();
    V.vstr.assign(R, VStr());
    V.vstr_of.assign(n, UINT32_MAX);
    for (size_t r = 0; r < R; ++r) V.vstr_of[V.relevant[r]] = (uint32_t)r;
    auto build = [&](size_t r0, size_t r1) {
        std::vector<const std::string *> seen;
        for (size_t r = r0; r < r1; ++r) {
            const size_t i = V.relevant[r];
            VStr &s = V.vstr[r];
            seen.clear();
            for (uint32_t k = V.H.off[i]; k < V.H.off[i + 1]; ++k) {
                const uint32_t t = V.H.tx[k];
                const std::string &gn = g->m.tx_gene_name[t];
                bool dup = false; for (auto *x : seen) if (*x == gn) dup = true;
                if (!dup) { if (!seen.empty()) s.genes += ","; s.genes += gn; seen.push_back(&gn); }
                if (k != V.H.off[i]) { s.transcripts += ","; s.distances += ","; s.annotations += ","; }
                s.transcripts += g->m.tx_id[t]; s.distances += std::to_string(V.H.dist[k]); s.annotations += kAnn[V.H.ann[k]];
            }
        }
    };
    const size_t nt = R < 4096 ? 1 : std::min<size_t>(usable_threads(16), 16);
    run_tasks(nt, [&](size_t k) { build(R * k / nt, R * (k + 1) / nt); });
    return RGX_OK;
}

// -v / `variants annotate -o`: what htslib writes for bcf_hdr_append x4 + bcf_hdr_write, then per record bcf_update_info_string x4 +
// bcf_write (variants_annotator.cc:130-154, 521-533) -- every record goes through BCF's typed form and back (vcf_rewrite.h).
// all_records = false writes only the splice relevant ones (identifier.cc:278-280), true every one (annotator.cc:545-548).
static int write_annotated_vcf(const char *path, const VariantStage &V, bool all_records, char *err, size_t errlen, bool print_notes = true) {
    FILE *fv = path ? fopen(path, "w") : stdout;
    if (!fv) return fail(err, errlen, RGX_ERR_OPEN, "Unable to open output VCF file.\n\n");
    if (fv != stdout) setvbuf(fv, nullptr, _IOFBF, 1 << 22);
    const VcfText &vcf = V.vcf;
    std::vector<size_t> todo;
    const size_t R = vcf.recs.size();
    for (size_t ri = 0; ri < R; ++ri) if (all_records || V.H.off[ri + 1] != V.H.off[ri]) todo.push_back(ri);
    const std::string e = write_annotated_vcf_records(fv, vcf, todo, [&](size_t ri) -> VcfAnnot {
        if (V.H.off[ri + 1] == V.H.off[ri]) return VcfAnnot{nullptr, nullptr, nullptr, nullptr};
        const VStr &s = V.vstr[V.vstr_of[ri]];
        return VcfAnnot{&s.genes, &s.transcripts, &s.distances, &s.annotations};
    }, print_notes);
    if (fv != stdout) fclose(fv);
    // (the record the reference's process ends in -- exit(1), or abort() -- is the one behind the last one written)
    if (!e.empty()) return fail(err, errlen, e != vcf.fatal ? RGX_ERR_OPEN : vcf.fatal_aborts ? RGX_ERR_ABORT : RGX_ERR_EXIT, "%s\n", e.c_str());
    return RGX_OK;
}

static const char *kJunctionHeader = "chrom\tstart\tend\tname\tscore\tstrand\tsplice_site\tacceptors_skipped\texons_skipped\tdonors_skipped\t"
                                     "anchor\tknown_donor\tknown_acceptor\tknown_junction\tgene_names\tgene_ids\ttranscripts";

// get_splice_site (junctions_annotator.cc:94-114); je = AnnotatedJunction.end
static int splice_site(const Fasta &fa, const std::string &chrom, uint32_t js, uint32_t je, const std::string &strand, std::string &site, char *err,
    size_t errlen) {
    std::string s1, s2;
    if (!fa.fetch(chrom, (int64_t)js + 1, (int64_t)js + 2, s1)) return fail(err, errlen, RGX_ERR_FASTA,
        "Unable to extract FASTA sequence for position %s:%u-%u\n\n", chrom.c_str(), js + 1, js + 2);
    if (!fa.fetch(chrom, (int64_t)je - 2, (int64_t)je - 1, s2)) return fail(err, errlen, RGX_ERR_FASTA,
        "Unable to extract FASTA sequence for position %s:%u-%u\n\n", chrom.c_str(), je - 2, je - 1);
    site = strand == "-" ? rev_comp(s2) + "-" + rev_comp(s1) : s1 + "-" + s2;
    return RGX_OK;
}

// what get_reference_sequence writes to stderr for a junction's two look-ups (junctions_annotator.cc:366-370), the second only if the first one was read
static void append_positions(std::string &o, const std::string &chrom, uint32_t js, uint32_t je, bool both = true) {
    o += "position = "; o += chrom; o += ':'; o += std::to_string(js + 1); o += '-'; o += std::to_string(js + 2); o += '\n';
    if (both) { o += "position = "; o += chrom; o += ':'; o += std::to_string(je - 2); o += '-'; o += std::to_string(je - 1); o += '\n'; }
}

// AnnotatedJunction::print (junctions_annotator.h:84-126) up to the transcripts column, row i of an annotate_junctions() result
// (text is appended to strings with to_chars: 66 k rows through fprintf into memory streams were 8.5 ms on 16 threads, a std::set of string pairs per row
// among them)
static inline void put_u(std::string &o, uint64_t v) { char b[24]; auto r = std::to_chars(b, b + sizeof b, v); o.append(b, (size_t)(r.ptr - b)); }
static inline void put_i(std::string &o, int64_t v) { char b[24]; auto r = std::to_chars(b, b + sizeof b, v); o.append(b, (size_t)(r.ptr - b)); }
static void append_junction_row(std::string &o, const rgx_gtf *g, const JunctionAnnotHost &A, size_t i, const std::string &chrom, uint32_t js, uint32_t je,
    const std::string &name,
                                const std::string &score, const std::string &strand, const std::string &site) {
    const uint32_t f = A.flags[i];
    const bool kd = f & 1, ka = f & 2, kj = f & 4;
    const char *anchor = kj ? "DA" : kd ? (ka ? "NDA" : "D") : ka ? "A" : "N";          // annotate_anchor :295-308
    o += chrom; o += '\t'; put_u(o, js); o += '\t'; put_u(o, je); o += '\t'; o += name; o += '\t'; o += score; o += '\t'; o += strand; o += '\t'; o += site;
        o += '\t';
    put_u(o, A.n_acc[i]); o += '\t'; put_u(o, A.n_exo[i]); o += '\t'; put_u(o, A.n_don[i]); o += '\t'; o += anchor;
    o += kd ? "\t1" : "\t0"; o += ka ? "\t1" : "\t0"; o += kj ? "\t1" : "\t0";
    if (A.tx_off[i + 1] > A.tx_off[i]) {
        // set< vector<string> > of (gene name, gene id): lexicographic, unique
        std::vector<std::pair<const std::string *, const std::string *>> genes;
        for (uint32_t k = A.tx_off[i]; k < A.tx_off[i + 1]; ++k) genes.push_back({&g->m.tx_gene_name[A.tx[k]], &g->m.tx_gene_id[A.tx[k]]});
        auto less = [](const std::pair<const std::string *, const std::string *> &x, const std::pair<const std::string *, const std::string *> &y) {
            const int c = x.first->compare(*y.first); return c < 0 || (c == 0 && *x.second < *y.second); };
        std::sort(genes.begin(), genes.end(), less);
        genes.erase(std::unique(genes.begin(), genes.end(), [](const auto &x, const auto &y) { return *x.first == *y.first && *x.second == *y.second; }),
            genes.end());
        o += '\t'; for (size_t k = 0; k < genes.size(); ++k) { if (k) o += ','; o += *genes[k].first; }
        o += '\t'; for (size_t k = 0; k < genes.size(); ++k) { if (k) o += ','; o += *genes[k].second; }
        o += '\t';
        for (uint32_t k = A.tx_off[i]; k < A.tx_off[i + 1]; ++k) { if (k != A.tx_off[i]) o += ','; o += g->m.tx_id[A.tx[k]]; }
    } else o += "\tNA\tNA\tNA";
}
static void print_junction_row(FILE *fo, const rgx_gtf *g, const JunctionAnnotHost &A, size_t i, const std::string &chrom, uint32_t js, uint32_t je,
    const std::string &name,
                               const std::string &score, const std::string &strand, const std::string &site) {
    std::string o;
    append_junction_row(o, g, A, i, chrom, js, je, name, score, strand, site);
    fwrite(o.data(), 1, o.size(), fo);
}

// unique_junctions_ / junction_to_variant_ (identifier.cc:292-299, associator.cc:266-270): first-inserted row wins per (chrom, start, end)
// Upstream: std::map<Junction-by-(chrom string, start, end), ...> filled with insert (the first row of a key stays) and, per junction, a
// std::set of variants ordered by (chrom string, start, end).  Here: the candidate rows as flat records keyed by the RANK of the contig name
// in string order, sorted once -- (key, arrival order) for the rows, (key, variant) for the links -- which visits keys, first rows and
// variants in exactly the order those containers iterate.  (The containers themselves, keyed on strings, were 25 of identify's 280 ms.)
struct JEntry { uint32_t ts, te, count; std::string strand, color; int nblocks; };
struct JTable {
    struct Cand { uint32_t crank, js, jend, order; uint32_t vrank, vpos, src; };     // src: where the caller finds the row's fields
    std::vector<std::string> chrom_name, vchrom_name;        // by rank
    std::vector<Cand> cand;                                  // every (row, variant) link in arrival order
    // after finish(): one entry per junction in map order, its variants in set order
    struct Row { uint32_t crank, js, jend; JEntry e; uint32_t v0, v1; };
    std::vector<Row> rows;
    std::vector<std::pair<uint32_t, uint32_t>> vars;         // (variant contig rank, pos0)
    void add(uint32_t crank, uint32_t js, uint32_t jend, uint32_t vrank, uint32_t vpos, uint32_t src) { cand.push_back(Cand{crank, js, jend,
        (uint32_t)cand.size(), vrank, vpos, src}); }
    template <class F> void finish(F entry_of /* src -> JEntry, asked once per junction */) {
        std::vector<uint32_t> idx(cand.size());
        for (uint32_t i = 0; i < idx.size(); ++i) idx[i] = i;
        WorkerPool own(tl_pool || cand.size() < (1u << 14) ? 1 : usable_threads(16));
        WorkerPool &pool = tl_pool ? *tl_pool : own;
        parallel_sort(pool, idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) {
            const Cand &x = cand[a], &y = cand[b];
            if (x.crank != y.crank) return x.crank < y.crank;
            if (x.js != y.js) return x.js < y.js;
            if (x.jend != y.jend) return x.jend < y.jend;
            return x.order < y.order;
        });
        rows.clear(); vars.clear();
        for (size_t i = 0; i < idx.size();) {
            const Cand &f = cand[idx[i]];                       // the first arrival of this key: its fields stay (map::insert)
            size_t j = i;
            const uint32_t v0 = (uint32_t)vars.size();
            while (j < idx.size() && cand[idx[j]].crank == f.crank && cand[idx[j]].js == f.js &&
                cand[idx[j]].jend == f.jend) { vars.push_back({cand[idx[j]].vrank, cand[idx[j]].vpos}); ++j; }
            std::sort(vars.begin() + v0, vars.end());
            vars.erase(std::unique(vars.begin() + v0, vars.end()), vars.end());
            rows.push_back(Row{f.crank, f.js, f.jend, entry_of(f.src), v0, (uint32_t)vars.size()});
            i = j;
        }
    }
    size_t size() const { return rows.size(); }
};
typedef JTable JMap;

// ranks of names in string order (equal names share a rank)
static void string_ranks(const std::vector<std::string> &names, std::vector<uint32_t> &rank_of, std::vector<std::string> &name_of_rank) {
    std::vector<uint32_t> order(names.size());
    for (uint32_t i = 0; i < order.size(); ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return names[a] < names[b]; });
    rank_of.assign(names.size(), 0); name_of_rank.clear();
    for (size_t k = 0; k < order.size(); ++k) {
        if (k == 0 || names[order[k]] != names[order[k - 1]]) name_of_rank.push_back(names[order[k]]);
        rank_of[order[k]] = (uint32_t)name_of_rank.size() - 1;
    }
}

// a11 + outputs (annotate_junctions identifier.cc:222-246 / associator.cc:182-203)
static int write_junction_outputs(rgx_ctx *c, const rgx_gtf *g, const char *fasta_path, const JMap &uj, const char *out_tsv, const char *out_bed,
    uint64_t *exon_visits,
                                  double *ms_annotate, char *err, size_t errlen, bool echo = false) {
    const double t0 = now_ms();
    struct Teardown { double t = 0; const char *what; ~Teardown() { if (t > 0) fprintf(stderr, "[rgx trace] %s +%8.3f ms\n", what, now_ms() - t);
        } } teardown{0, "outputs: locals released"};
    Fasta *fap = host_fasta(c, fasta_path);
    if (!fap) return fail(err, errlen, RGX_ERR_FASTA, "Unable to open FASTA file.\n\n");
    const Fasta &fa = *fap;
    std::vector<int32_t> jc; std::vector<uint32_t> jjs, jje; std::vector<uint8_t> jst;
    jc.reserve(uj.size()); jjs.reserve(uj.size()); jje.reserve(uj.size()); jst.reserve(uj.size());
    {
        std::vector<int32_t> gtf_chrom(uj.chrom_name.size());
        for (size_t k = 0; k < gtf_chrom.size(); ++k) gtf_chrom[k] = g->m.chrom_of(uj.chrom_name[k]);
        for (const JTable::Row &r : uj.rows) {
            jc.push_back(gtf_chrom[r.crank]); jjs.push_back(r.js); jje.push_back(r.jend + 1);
            jst.push_back(r.e.strand.size() == 1 ? (uint8_t)r.e.strand[0] : (uint8_t)'?');
        }
    }
    JunctionAnnotHost A;
    int rc = annotate_junctions(c, g, jc, jjs, jje, jst, A, err, errlen, exon_visits);
    if (rc != RGX_OK) return rc;
    if (ms_annotate) *ms_annotate += now_ms() - t0;
    const bool trace = getenv("REGTOOLS_AMD_TRACE") != nullptr;
    double t_last = now_ms();
    auto lap = [&](const char *what) { if (trace) { const double t = now_ms(); fprintf(stderr, "[rgx trace] outputs: %-18s +%8.3f ms\n", what, t - t_last);
        t_last = t; } };
    // get_splice_site for every junction up front, on several host threads: two 2-base reads at random places of a multi-GB FASTA mapping are
    // two page faults per junction (0.13 s of config 4's 0.47 s when done row by row in the print loop).  The FIRST junction that fails, in
    // output order, ends the run with its message after the rows before it were written -- as the row-by-row loop did.
    std::vector<std::string> sites(uj.size());
    size_t first_bad = SIZE_MAX; char bad_msg[512] = {0};
    const std::vector<JTable::Row> &rows = uj.rows;
    {
        const size_t n = rows.size();
        const size_t T = n < 2048 ? 1 : usable_threads(16);
        std::vector<size_t> bad(T, SIZE_MAX); std::vector<std::string> msg(T);
        auto work = [&](size_t t) {
            for (size_t k = n * t / T; k < n * (t + 1) / T; ++k) {
                const JTable::Row &r = rows[k];
                char e2[512] = {0};
                if (splice_site(fa, uj.chrom_name[r.crank], r.js, r.jend + 1, r.e.strand, sites[k], e2, sizeof e2) != RGX_OK) { bad[t] = k; msg[t] = e2;
                    return; }
            }
        };
        run_tasks(T, work);
        for (size_t t = 0; t < T; ++t) if (bad[t] < first_bad) { first_bad = bad[t]; snprintf(bad_msg, sizeof bad_msg, "%s", msg[t].c_str()); }
    }
    lap("splice sites");
    if (echo) {                                                    // (in output order; the junction whose look-up fails is the last one heard of)
        std::string s;
        const size_t upto = std::min(rows.size(), first_bad);
        s.reserve(upto * 64);
        for (size_t k = 0; k < upto; ++k) append_positions(s, uj.chrom_name[rows[k].crank], rows[k].js, rows[k].jend + 1);
        if (first_bad != SIZE_MAX) {
            const JTable::Row &r = rows[first_bad];
            std::string tmp;
            append_positions(s, uj.chrom_name[r.crank], r.js, r.jend + 1, fa.fetch(uj.chrom_name[r.crank], (int64_t)r.js + 1, (int64_t)r.js + 2, tmp));
        }
        fwrite(s.data(), 1, s.size(), stderr);
    }
    FILE *fo = out_tsv ? fopen(out_tsv, "w") : stdout;
    if (!fo) return fail(err, errlen, RGX_ERR_OPEN, "Unable to open %s", out_tsv);
    FILE *fj = out_bed ? fopen(out_bed, "w") : nullptr;
    if (fo != stdout) setvbuf(fo, nullptr, _IOFBF, 1 << 22);
    if (fj) setvbuf(fj, nullptr, _IOFBF, 1 << 22);
    fprintf(fo, "%s\tvariant_info\n", kJunctionHeader);
    // the rows are formatted by several threads, each into its own memory stream, and written out in order (66 k rows: 80 ms in one thread)
    // (round 6: four chunks per thread, handed out as threads come free -- rows that name many transcripts made equal shares take 1.6 to 5.4 ms -- and
    //  a writer thread that puts the chunks out in order while the later ones are still being formatted)
    const size_t n_rows = std::min(rows.size(), first_bad);
    const size_t T = n_rows < 4096 ? 1 : 4 * usable_threads(16);
    struct Chunk { std::string tsv, bed; bool ok = true; std::atomic<int> ready{0}; };
    std::vector<Chunk> chunks(T);
    const bool want_bed = fj != nullptr;
    std::vector<double> task_ms(T, 0);
    auto format = [&](size_t t) {
        Chunk &ck = chunks[t];
        const double t_task = trace ? now_ms() : 0;
        struct Stamp { double &slot; double t0; bool on; ~Stamp() { if (on) slot = now_ms() - t0; } } stamp{task_ms[t], t_task, trace};
        try {
            const size_t i0 = n_rows * t / T, i1 = n_rows * (t + 1) / T;
            ck.tsv.reserve((i1 - i0) * 224);
            if (want_bed) ck.bed.reserve((i1 - i0) * 96);
            std::string name, score;
            for (size_t i = i0; i < i1; ++i) {
                // (a row names a handful of transcripts anywhere in three 8 MB string tables: their lines are asked for a few rows ahead --
                //  the formatting was bound by those misses, not by the text)
                if (i + 6 < i1) for (uint32_t k = A.tx_off[i + 6]; k < A.tx_off[i + 7]; ++k) {
                    const uint32_t t2 = A.tx[k];
                    __builtin_prefetch(&g->m.tx_gene_name[t2]); __builtin_prefetch(&g->m.tx_gene_id[t2]); __builtin_prefetch(&g->m.tx_id[t2]);
                }
                const JTable::Row &r = rows[i];
                const std::string &chrom = uj.chrom_name[r.crank];
                const uint32_t js = r.js, jend = r.jend, je = jend + 1;
                const JEntry &e = r.e;
                { char nb[32]; snprintf(nb, sizeof nb, "JUNC%08zu", i + 1); name = nb; }
                if (want_bed) {
                    std::string &b = ck.bed;
                    b += chrom; b += '\t'; put_u(b, e.ts); b += '\t'; put_u(b, e.te); b += '\t'; b += name; b += '\t'; put_u(b, e.count); b += '\t';
                        b += e.strand; b += '\t';
                    put_u(b, e.ts); b += '\t'; put_u(b, e.te); b += '\t'; b += e.color; b += '\t'; put_i(b, e.nblocks); b += '\t';
                    put_u(b, (uint32_t)(js - e.ts)); b += ','; put_u(b, (uint32_t)(e.te - jend)); b += "\t0,"; put_u(b, (uint32_t)(jend - e.ts)); b += '\n';
                }
                score.clear(); put_u(score, e.count);
                append_junction_row(ck.tsv, g, A, i, chrom, js, je, name, score, e.strand, sites[i]);
                ck.tsv += '\t';
                for (uint32_t k = r.v0; k < r.v1; ++k) {                                // variant_set_to_string
                    if (k != r.v0) ck.tsv += ',';
                    ck.tsv += uj.vchrom_name[uj.vars[k].first]; ck.tsv += ':'; put_i(ck.tsv, (int)uj.vars[k].second); ck.tsv += '-'; put_i(ck.tsv,
                        (int)(uj.vars[k].second + 1));
                }
                ck.tsv += '\n';
            }
        } catch (const std::bad_alloc &) { ck.ok = false; }
        ck.ready.store(1, std::memory_order_release);
    };
    std::atomic<bool> mem_bad{false};
    std::thread writer;
    if (T > 1) writer = std::thread([&] {
        for (Chunk &ck : chunks) {
            while (!ck.ready.load(std::memory_order_acquire)) std::this_thread::sleep_for(std::chrono::microseconds(50));
            if (!ck.ok) { mem_bad.store(true); return; }            // (what was written stays; the call fails)
            if (!ck.tsv.empty()) fwrite(ck.tsv.data(), 1, ck.tsv.size(), fo);
            if (fj && !ck.bed.empty()) fwrite(ck.bed.data(), 1, ck.bed.size(), fj);
            std::string().swap(ck.tsv); std::string().swap(ck.bed);
        }
    });
    struct JoinWriter { std::thread &t; ~JoinWriter() { if (t.joinable()) t.join(); } } join_writer{writer};
    run_tasks(T, format);
    if (trace) { double lo = 1e9, hi = 0, sum = 0; for (double v : task_ms) { lo = std::min(lo, v); hi = std::max(hi, v); sum += v; } fprintf(stderr,
        "[rgx trace] outputs: %zu format tasks: min %.3f avg %.3f max %.3f ms\n", T, lo, sum / (double)T, hi); }
    lap("rows formatted");
    bool mem_ok = true;
    if (writer.joinable()) { writer.join(); mem_ok = !mem_bad.load(); }
    else {
        for (const Chunk &ck : chunks) if (!ck.ok) mem_ok = false;
        if (mem_ok) for (Chunk &ck : chunks) { if (!ck.tsv.empty()) fwrite(ck.tsv.data(), 1, ck.tsv.size(), fo); if (fj &&
            !ck.bed.empty()) fwrite(ck.bed.data(), 1, ck.bed.size(), fj); }
    }
    if (!mem_ok) { if (fo != stdout) fclose(fo); if (fj) fclose(fj); return fail(err, errlen, RGX_ERR_OPEN, "regtools_amd: no memory for the output rows\n"); }
    if (first_bad != SIZE_MAX) { if (fo != stdout) fclose(fo); if (fj) fclose(fj); return fail(err, errlen, RGX_ERR_FASTA, "%s", bad_msg); }
    if (fo != stdout) fclose(fo);
    if (fj) fclose(fj);
    lap("rows");
    if (trace) teardown.t = now_ms();
    return RGX_OK;
}

// ---- `cis-splice-effects identify` ---------------------------------------------------------------------------------------------
// multi.cpp: the process-wide context of the nth listing of a device (rgx_extract_multi's cache)
rgx_ctx *rgx_multi_context(int device, int nth, char *err, size_t errlen, int *rc);

// SURVEY 8e for `identify`: the extraction is what is worth sharding (29 of config 4's 36 ms of device work) -- shard g of the BAM is inflated, framed
// and scanned on device g exactly as rgx_extract_multi's shards are (contiguous member ranges cut at record starts from the index, one host scan of
// the members for all), and the junction EVENTS (32 B each; config 4: 7.5 M = 230 MB) are gathered onto the first device in shard order = file order,
// where the windows are joined as on one device.  The gather is device-to-device copies (hipMemcpyPeerAsync: xGMI between two GPUs, a plain copy when
// a device is listed twice) -- no reduction is involved, so no collective.  A shard whose record stream ended for a reason that ends iteration upstream
// ends the event list (the shards behind it are dropped, as the table merge drops them).
static int prepare_events_sharded(const std::vector<rgx_ctx *> &cs, const uint8_t *bam, size_t bam_len, const uint8_t *bai, size_t bai_len,
    const rgx_extract_params *ep,
                                  Prep &P, char *err, size_t errlen) {
    const int n = (int)cs.size();
    std::vector<Member> members; uint64_t total_inflated = 0;
    if (bam_len < ((size_t)8 << 20) || !scan_members_parallel(bam, bam_len, (int)usable_threads(24), members, total_inflated))
        // a file the host scan does not vouch for: one device, its own member discovery
        return prepare_events(cs[0], nullptr, bam, bam_len, bai, bai_len, ep, true, P, err, errlen);
    SharedMembers sm{&members, total_inflated};
    std::vector<Prep> parts((size_t)n);
    std::vector<int> rcs((size_t)n, RGX_OK);
    std::vector<std::string> errs((size_t)n, std::string(512, '\0'));
    auto run = [&](int g) {
        rgx_extract_params q = *ep;
        q.shard = g; q.n_shards = n;
        rcs[(size_t)g] = prepare_events(cs[(size_t)g], nullptr, bam, bam_len, bai, bai_len, &q, true, parts[(size_t)g], &errs[(size_t)g][0], 512, nullptr,
            true, false, &sm);
    };
    bool distinct = true;
    for (int a = 0; a < n; ++a) for (int b = a + 1; b < n; ++b) if (cs[(size_t)a]->device == cs[(size_t)b]->device) distinct = false;
    if (distinct) {
        std::vector<std::thread> pool;
        for (int g = 1; g < n; ++g) pool.emplace_back(run, g);
        run(0);
        for (auto &t : pool) t.join();
    } else for (int g = 0; g < n; ++g) run(g);                   // (shards that share a device take turns on it)
    for (int g = 0; g < n; ++g) if (rcs[(size_t)g] != RGX_OK) return fail(err, errlen, rcs[(size_t)g], "%s", errs[(size_t)g].c_str());
    int used = n;
    uint64_t N = 0, iterated = 0, rec = 0;
    for (int g = 0; g < n; ++g) {
        N += parts[(size_t)g].n_events; iterated += parts[(size_t)g].n_iterated; rec += parts[(size_t)g].n_rec;
        if (parts[(size_t)g].stream_ended) { used = g + 1; break; }
    }
    if (N >= 0xfffffff0ull) return fail(err, errlen, RGX_ERR_ARG, "regtools_amd: %llu junction events; more than the join handles\n", (unsigned long long)N);
    rgx_ctx *c0 = cs[0];
    HIP_TRY(hipSetDevice(c0->device));
    DevBuf &ball = c0->buf("cse_events_all");
    const size_t Nn = (size_t)N, stride = (Nn * 4 + 255) & ~(size_t)255;
    HIP_TRY(ball.ensure(stride * 8 + 256));
    uint8_t *q = ball.as<uint8_t>();
    EventSoA all; memset(&all, 0, sizeof all);
    all.tid = (uint32_t *)q; all.start = (uint32_t *)(q + stride); all.ilen_cls = (uint32_t *)(q + 2 * stride); all.ts = (uint32_t *)(q + 3 * stride);
    all.te = (uint32_t *)(q + 4 * stride); all.rpos = (uint32_t *)(q + 5 * stride); all.rend = (uint32_t *)(q + 6 * stride); all.strand = q + 7 * stride;
    size_t off = 0;
    for (int g = 0; g < used; ++g) {
        const Prep &pg = parts[(size_t)g];
        const size_t k = pg.n_events;
        if (!k) continue;
        const int dg = cs[(size_t)g]->device;
        HIP_TRY(hipSetDevice(dg));
        HIP_TRY(hipStreamSynchronize(cs[(size_t)g]->stream));      // (the shard's events are complete)
        (void)rgx_enable_peer(c0->device, dg);                       // (xGMI instead of a bounce through the host; a copy works either way)
        HIP_TRY(hipSetDevice(c0->device));
#define RGX_GATHER(F, BYTES) HIP_TRY(hipMemcpyPeerAsync((uint8_t *)all.F + off * (BYTES), c0->device, pg.ev.F, dg, k * (BYTES), c0->stream))
        RGX_GATHER(tid, 4); RGX_GATHER(start, 4); RGX_GATHER(ilen_cls, 4); RGX_GATHER(ts, 4); RGX_GATHER(te, 4); RGX_GATHER(rpos, 4); RGX_GATHER(rend, 4);
            RGX_GATHER(strand, 1);
#undef RGX_GATHER
        off += k;
    }
    HIP_TRY(hipStreamSynchronize(c0->stream));
    P = Prep();
    P.hdr = parts[0].hdr; P.ev = all; P.n_events = (uint32_t)N; P.n_iterated = iterated; P.n_rec = (uint32_t)std::min<uint64_t>(rec, 0xffffffffull);
    P.stream_ended = used < n || parts[(size_t)used - 1].stream_ended;
    for (int g = 0; g < used; ++g) P.odd_aux.insert(P.odd_aux.end(), parts[(size_t)g].odd_aux.begin(), parts[(size_t)g].odd_aux.end());
    return RGX_OK;
}

static int identify_run(rgx_ctx *c, const std::vector<rgx_ctx *> *shards, const rgx_identify_params *p, rgx_identify_stats *stats, char *err, size_t errlen);
extern "C" int rgx_identify(rgx_ctx *c, const rgx_identify_params *p, rgx_identify_stats *stats, char *err, size_t errlen) { return identify_run(c, nullptr,
    p, stats, err, errlen); }

// `identify` with the extraction sharded over the listed devices (prepare_events_sharded); everything behind it on the first one.  A device may be
// listed more than once (its shards take turns on it).  The outputs do not depend on the list.
extern "C" int rgx_identify_multi(const int *devices, int n_devices, const rgx_identify_params *p, rgx_identify_stats *stats, char *err, size_t errlen) {
    if (!devices || n_devices <= 0 || n_devices > 255 || !p) return fail(err, errlen, RGX_ERR_ARG, "regtools_amd: bad arguments\n");
    static std::mutex call_mu;                                    // the cached contexts are shared by every call of the process: calls take turns
    std::lock_guard<std::mutex> lock(call_mu);
    std::vector<rgx_ctx *> cs((size_t)n_devices);
    std::map<int, int> seen;
    for (int g = 0; g < n_devices; ++g) {
        int rc = RGX_OK;
        cs[(size_t)g] = rgx_multi_context(devices[g], seen[devices[g]]++, err, errlen, &rc);
        if (rc != RGX_OK) return rc;
    }
    return identify_run(cs[0], &cs, p, stats, err, errlen);
}

static int identify_run(rgx_ctx *c, const std::vector<rgx_ctx *> *shards, const rgx_identify_params *p, rgx_identify_stats *stats, char *err, size_t errlen) {
    if (!c || !p) return fail(err, errlen, RGX_ERR_ARG, "regtools_amd: bad arguments\n");
    if (!p->vcf_path || !p->bam_path || !p->fasta_path || !p->gtf_path) return fail(err, errlen, RGX_ERR_ARG, "Error parsing inputs!(2)\n\n");
    if (p->strandness < 0 || p->strandness > 3) return fail(err, errlen, RGX_ERR_ARG, "Please supply strand specificity with '-s' option!\n\n");
    struct Teardown { double t = 0; const char *what; ~Teardown() { if (t > 0) fprintf(stderr, "[rgx trace] %s +%8.3f ms\n", what, now_ms() - t);
        } } teardown{0, "identify: locals released"};
    rgx_identify_stats S; memset(&S, 0, sizeof S);
    ktime_collect(c); c->kms[0] = c->kms[1] = c->kms[2] = 0;
    const double t0 = now_ms();
    double tl = t0;
    auto lap = [&](double &slot) { const double t = now_ms(); slot += t - tl; tl = t; };

    // The three inputs are independent until the variant scan, so they are read at the same time: the GTF and the VCF text on two host
    // threads (parsing only, no device calls), the BAM on this one, which owns the device.  What the reference reports when several of them
    // are unusable is decided afterwards, in its order: GtfParser::load (identifier.cc:258-259), the FASTA, the VCF, and the BAM only when a
    // variant is splice relevant (it opens the BAM per such variant, :288-290) -- the extraction below is done ahead of knowing that.
    rgx_gtf *g = new rgx_gtf();
    g->ctx = c;
    struct GtfGuard { rgx_gtf *g; ~GtfGuard() { rgx_gtf_free(g); } } guard{g};
    std::unique_ptr<VariantStage> V_holder(new VariantStage());      // (on the heap: a finished call hands its teardown to the background thread)
    VariantStage &V = *V_holder;
    std::string gtf_err, vcf_err;
    double gtf_thread_ms = 0, vcf_thread_ms = 0;
    std::thread t_gtf([&] { const double t = now_ms(); try { gtf_err = g->m.load(p->gtf_path);
        } catch (const std::exception &e) { gtf_err = std::string("regtools_amd: ") + e.what() + "\n"; } gtf_thread_ms = now_ms() - t; });
    std::thread t_vcf([&] { const double t = now_ms(); try { vcf_err = V.vcf.load(p->vcf_path, /*annotating=*/p->out_vcf != nullptr);
        } catch (const std::exception &e) { vcf_err = std::string("regtools_amd: ") + e.what() + "\n"; } vcf_thread_ms = now_ms() - t; });
    struct Joiner { std::thread &a, &b; ~Joiner() { if (a.joinable()) a.join(); if (b.joinable()) b.join(); } } joiner{t_gtf, t_vcf};

    // ---- identify: the extraction, ONCE for all windows (the reference re-opens the BAM per variant: identifier.cc:288-290) ----
    FileBytes bam; std::vector<uint8_t> bai;
    // what htslib says when the BAM and its index are opened (no EOF member, an index older than the file): upstream opens both once per splice-relevant
    // variant, behind the variant's echo (identifier.cc:288-289)
    std::string bam_notes;
    Prep P;
    int rc_bam = RGX_OK;
    char err_bam[512]; err_bam[0] = 0;
    rgx_extract_params ep; rgx_extract_params_default(&ep);
    if (!p->bed_path) {
        ep.region = "."; ep.strandness = p->strandness; ep.strand_tag[0] = p->strand_tag[0]; ep.strand_tag[1] = p->strand_tag[1];
        ep.min_anchor = p->min_anchor; ep.min_intron = p->min_anchor /* ctor quirk junctions_extractor.h:200 */; ep.max_intron = p->max_intron;
        ep.fasta_path = (p->override_motif || p->strandness == 3) ? p->fasta_path : nullptr;   // ref_to_pass (identifier.cc:282-287)
        std::string idx;
        if (!bam.open(p->bam_path)) rc_bam = fail(err_bam, sizeof err_bam, RGX_ERR_OPEN, "%s", kMsgOpen);
        else if (find_index(p->bam_path, idx) != 0) { bam_notes = bam_open_notes(bam.data(), bam.size(), nullptr, nullptr);
            rc_bam = fail(err_bam, sizeof err_bam, RGX_ERR_INDEX, "%s", kMsgIndex); }
        else if (bam_notes = bam_open_notes(bam.data(), bam.size(), p->bam_path, idx.c_str()), !read_index(idx, bai)) rc_bam = fail(err_bam, sizeof err_bam,
            RGX_ERR_INDEX, "%s", kMsgIndex);
        else if (shards && shards->size() > 1) rc_bam = prepare_events_sharded(*shards, bam.data(), bam.size(), bai.data(), bai.size(), &ep, P, err_bam,
            sizeof err_bam);
        else rc_bam = prepare_events(c, nullptr, bam.data(), bam.size(), bai.data(), bai.size(), &ep, true, P, err_bam, sizeof err_bam);
        lap(S.ms_extract);
    }

    // (the host stages behind this point share one pool of threads; started here, where this thread would wait for the GTF otherwise)
    WorkerPool stage_pool(usable_threads(16));
    struct PoolScope { WorkerPool *prev; PoolScope(WorkerPool *p) : prev(tl_pool) { tl_pool = p; } ~PoolScope() { tl_pool = prev; } } pool_scope{&stage_pool};
    t_gtf.join();
    if (!gtf_err.empty()) return fail(err, errlen, RGX_ERR_FORMAT, "%s", gtf_err.c_str());
    // (the annotator's constructor prints the member before it assigns it: always the default, variants_annotator.h:141-152)
    if (p->echo) fputs("exonic_min_distance_ is 3\n", stderr);
    int rc = gtf_upload(c, g, err, errlen, /*pooled=*/true);
    if (rc != RGX_OK) return rc;
    lap(S.ms_gtf);                                              // (what of the GTF was still to do when the extraction was done)
    if (!host_fasta(c, p->fasta_path)) return fail(err, errlen, RGX_ERR_FASTA, "Unable to open FASTA file.\n\n");

    // a10: every variant against the annotation
    t_vcf.join();
    if (!vcf_err.empty()) return fail(err, errlen, V.vcf.death == 2 ? RGX_ERR_ABORT : V.vcf.death == 1 ? RGX_ERR_EXIT : RGX_ERR_OPEN, "%s", vcf_err.c_str());
    if (p->echo) fputs("\n", stderr);                                // (identifier.cc:265, associator.cc:243)
    if (getenv("REGTOOLS_AMD_TRACE")) fprintf(stderr, "[rgx trace] inputs: gtf thread %8.3f ms, vcf thread %8.3f ms, extraction %8.3f ms (side by side)\n",
        gtf_thread_ms, vcf_thread_ms, S.ms_extract);
    VariantOpts vo{p->intronic_min, p->exonic_min, p->all_intronic, p->all_exonic, p->skip_single};
    rc = variant_scan_stage(c, g, vo, V, &S.exon_visits_variants, err, errlen);
    if (rc != RGX_OK) return rc;
    const VcfText &vcf = V.vcf; const VariantHitsHost &H = V.H; const std::vector<size_t> &relevant = V.relevant;
    S.n_variants = vcf.recs.size(); S.n_relevant = relevant.size();
    lap(S.ms_variants);
    // the annotated VCF is written on a side thread while the windows are joined and the junctions annotated (it only reads V); it is complete, or
    // its error is the call's, before any junction output is opened
    int rc_vcf = RGX_OK;
    char err_vcf[512]; err_vcf[0] = 0;
    double vcf_ms = 0;
    std::thread t_vcfout;
    if (p->out_vcf) t_vcfout = std::thread([&] { const double t = now_ms(); rc_vcf = write_annotated_vcf(p->out_vcf, V, false, err_vcf, sizeof err_vcf,
        /*print_notes=*/false); vcf_ms = now_ms() - t; });
    struct JoinOne { std::thread &t; ~JoinOne() { if (t.joinable()) t.join(); } } join_vcfout{t_vcfout};

    // p->echo: what upstream writes to stderr for every splice-relevant variant, in file order, before it looks at the alignments of its window
    // (identifier.cc:275-277, associator.cc:255-257): "Variant " + BED's operator<< (chrom, start, end, score, strand, each followed by a tab;
    // bedFile.h:183-194; the score is what the annotation walk left there: H.last) and the window as the region string
    // What reading the records says (vcf.notes: a name the header does not declare, ...) comes out here as well, a record's lines in front of its
    // "Variant" lines: upstream reads, annotates and echoes one record after the other (identifier.cc:267-277).
    auto echo_variants = [&](size_t upto) {
        const bool all = upto >= relevant.size();
        if (!p->echo) {                                                    // (a library caller: the records' lines, and the BAM's once)
            vcf.flush_notes(all ? SIZE_MAX : relevant[upto - 1] + 1);
            if (upto && !relevant.empty()) fputs(bam_notes.c_str(), stderr);
            return;
        }
        std::string s;
        s.reserve(std::min(upto, relevant.size()) * 72);
        for (size_t w = 0; w < upto && w < relevant.size(); ++w) {
            const size_t i = relevant[w];
            if (vcf.notes_printed < vcf.notes.size() && vcf.notes[vcf.notes_printed].first <= i) {
                fwrite(s.data(), 1, s.size(), stderr); s.clear();
                vcf.flush_notes(i + 1);
            }
            const uint32_t start = vcf.recs[i].pos0, end = start + 1;
            const uint32_t rs = p->window ? (uint32_t)(start - p->window) : H.ces[i], re = p->window ? (uint32_t)(end + p->window) : H.cee[i];
            s += "Variant "; s += vcf.recs[i].chrom; s += '\t'; put_u(s, start); s += '\t'; put_u(s, end); s += '\t';
            if (H.last[i] == 0xffffffffu) s += "-1"; else put_u(s, H.last[i]);
            s += "\t\t\nVariant region is "; s += vcf.recs[i].chrom; s += ':'; put_u(s, rs); s += '-'; put_u(s, re); s += "\n\n";
            s += bam_notes;
        }
        fwrite(s.data(), 1, s.size(), stderr);
        if (all) vcf.flush_notes(SIZE_MAX);
    };
    // the record the reference's process ends in, once everything in front of it is echoed (its variants have had their windows read by then)
    auto died_reading_the_vcf = [&]() -> int {
        if (vcf.fatal.empty()) return RGX_OK;
        return fail(err, errlen, vcf.fatal_aborts ? RGX_ERR_ABORT : RGX_ERR_EXIT, "%s\n", vcf.fatal.c_str());
    };
    JMap uj;
    if (p->bed_path) {
        echo_variants(relevant.size());
        if (int rc_died = died_reading_the_vcf()) return rc_died;
        // ---- associate: junctions from a BED12 (associator.cc:206-276) ----
        BedJunctions B;
        { std::string e = B.load(p->bed_path); if (!e.empty()) return fail(err, errlen, RGX_ERR_FORMAT, "%s", e.c_str()); }
        // bucket by contig, file order kept inside a bucket
        std::unordered_map<std::string, int32_t> cidx; std::vector<std::string> cname;
        std::vector<int32_t> jch(B.n());
        for (size_t i = 0; i < B.n(); ++i) { auto it = cidx.find(B.chrom[i]); if (it == cidx.end()) { it = cidx.emplace(B.chrom[i],
            (int32_t)cname.size()).first; cname.push_back(B.chrom[i]); } jch[i] = it->second; }
        std::vector<uint32_t> chrom_off(cname.size() + 1, 0), order(B.n()), js(B.n()), je(B.n());
        for (size_t i = 0; i < B.n(); ++i) chrom_off[(size_t)jch[i] + 1]++;
        for (size_t k = 0; k < cname.size(); ++k) chrom_off[k + 1] += chrom_off[k];
        { std::vector<uint32_t> fill(chrom_off.begin(), chrom_off.end() - 1);
          // Junction.end = line.end - 1 (:221)
          for (size_t i = 0; i < B.n(); ++i) { const uint32_t q = fill[(size_t)jch[i]]++; order[q] = (uint32_t)i; js[q] = B.start[i]; je[q] = B.end[i] - 1; } }
        const uint32_t W = (uint32_t)relevant.size(), J = (uint32_t)B.n();
        S.n_windows = W; S.n_events = J;
        std::vector<uint32_t> pj, pw;
        if (W && J) {
            std::vector<int32_t> wch(W); std::vector<uint32_t> wces(W), wcee(W);
            for (uint32_t w = 0; w < W; ++w) { const size_t vi = relevant[w]; auto it = cidx.find(vcf.recs[vi].chrom); wch[w] = it == cidx.end() ? -1 :
                it->second; wces[w] = H.ces[vi]; wcee[w] = H.cee[vi]; }
            hipStream_t st = c->stream;
            HIP_ENTER(c->device);
            DevBuf &b = c->buf("cse_assoc"), &sc = c->buf("scalars");
            HIP_TRY(sc.ensure(512));
            const size_t Wn = W, Jn = J, Cn = chrom_off.size();
            HIP_TRY(b.ensure((Wn * 5 + Jn * 2 + Cn + scan_tmp_words(W)) * 4 + 512));
            uint32_t *w = b.as<uint32_t>();
            int32_t *d_wch = (int32_t *)w; w += Wn; uint32_t *d_ces = w; w += Wn; uint32_t *d_cee = w; w += Wn; uint32_t *d_cnt = w; w += Wn;
                uint32_t *d_base = w; w += Wn;
            uint32_t *d_js = w; w += Jn; uint32_t *d_je = w; w += Jn; uint32_t *d_off = w; w += Cn; uint32_t *d_tmp = w;
            uint32_t *d_total = sc.as<uint32_t>() + 68;
            HIP_TRY(hipMemcpyAsync(d_wch, wch.data(), Wn * 4, hipMemcpyHostToDevice, st)); HIP_TRY(hipMemcpyAsync(d_ces, wces.data(), Wn * 4,
                hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(d_cee, wcee.data(), Wn * 4, hipMemcpyHostToDevice, st)); HIP_TRY(hipMemcpyAsync(d_js, js.data(), Jn * 4,
                hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(d_je, je.data(), Jn * 4, hipMemcpyHostToDevice, st)); HIP_TRY(hipMemcpyAsync(d_off, chrom_off.data(), Cn * 4,
                hipMemcpyHostToDevice, st));
            launch_assoc_pairs(false, W, d_wch, d_ces, d_cee, d_off, d_js, d_je, d_cnt, nullptr, nullptr, nullptr, st);
            launch_scan_u32(d_cnt, d_base, W, d_total, d_tmp, st);
            uint32_t total = 0;
            std::vector<uint32_t> h_cnt(Wn);
            HIP_TRY(hipMemcpyAsync(h_cnt.data(), d_cnt, Wn * 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(&total, d_total, 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            { uint64_t t64 = 0; for (uint32_t x : h_cnt) t64 += x;                   // the device scan is 32 bits wide
              if (t64 >= (1ull << 31)) return fail(err, errlen, RGX_ERR_ARG, "regtools_amd: %llu (variant, junction) pairs; more than the join handles\n",
                  (unsigned long long)t64); }
            if (total) {
                DevBuf &bp = c->buf("cse_pairs");
                HIP_TRY(bp.ensure((size_t)total * 8 + 256));
                uint32_t *d_pj = bp.as<uint32_t>(), *d_pw = d_pj + total;
                launch_assoc_pairs(true, W, d_wch, d_ces, d_cee, d_off, d_js, d_je, d_cnt, d_base, d_pj, d_pw, st);
                pj.resize(total); pw.resize(total);
                HIP_TRY(hipMemcpyAsync(pj.data(), d_pj, (size_t)total * 4, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipMemcpyAsync(pw.data(), d_pw, (size_t)total * 4, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
            }
            S.n_pairs = total;
        }
        S.n_window_rows = pj.size();
        {
            std::vector<uint32_t> crank_of, vrank_of;
            string_ranks(cname, crank_of, uj.chrom_name);
            std::vector<std::string> vnames; std::unordered_map<std::string, uint32_t> vidx; std::vector<uint32_t> v_name(relevant.size());
            for (size_t w = 0; w < relevant.size(); ++w) { const std::string &cn = vcf.recs[relevant[w]].chrom; auto it = vidx.find(cn);
                if (it == vidx.end()) { it = vidx.emplace(cn, (uint32_t)vnames.size()).first; vnames.push_back(cn); } v_name[w] = it->second; }
            string_ranks(vnames, vrank_of, uj.vchrom_name);
            uj.cand.reserve(pj.size());
            for (size_t r = 0; r < pj.size(); ++r) {
                const size_t vi = relevant[pw[r]], bi = order[pj[r]];
                uj.add(crank_of[(size_t)jch[bi]], B.start[bi], B.end[bi] - 1, vrank_of[v_name[pw[r]]], vcf.recs[vi].pos0, (uint32_t)bi);
            }
            uj.finish([&](uint32_t bi) { return JEntry{B.ts[bi], B.te[bi], (uint32_t)atoi(B.score[bi].c_str()), B.strand[bi], B.color[bi], B.nblocks[bi]}; });
        }
        S.n_junctions = uj.size();
        lap(S.ms_join);
    } else {
        // (upstream opens the BAM for the first such variant)
        if (!relevant.empty() && rc_bam != RGX_OK) { echo_variants(1); return fail(err, errlen, rc_bam, "%s", err_bam); }
        if (relevant.empty()) P = Prep();                       // (no variant asks for the BAM: upstream never opens it)
        S.n_records = P.n_iterated; S.n_events = P.n_events;

        // windows (identifier.cc:270-274): "chrom:start-end" built with uint32 arithmetic, then parsed as sam_itr_querys would
        const bool jtrace = getenv("REGTOOLS_AMD_TRACE") != nullptr;
        double jt = now_ms();
        std::vector<int32_t> w_tid, w_beg, w_end;
        // a file whose record stream ended (damage): every window is read through the index on its own, as upstream reads it (window_join_by_seeks)
        const bool by_seeks = P.stream_ended && !relevant.empty();
        std::vector<std::string> w_region;
        HostRows R;
        BaiInfo bi; (void)parse_bai(bai.data(), bai.size(), bi, false);
        {
            // (every window's region string goes through the region parser, as upstream; ranges of them on several threads, the first one that
            //  does not parse -- in file order -- aborts the run)
            const size_t W = relevant.size();
            w_tid.resize(W); w_beg.resize(W); w_end.resize(W);
            if (by_seeks) w_region.resize(W);
            WorkerPool own(tl_pool || W < 4096 ? 1 : usable_threads(16));
            WorkerPool &pool = tl_pool && W >= 4096 ? *tl_pool : own;
            const size_t nt = pool.threads();
            std::vector<size_t> bad(nt, SIZE_MAX);
            pool.run(nt, [&](size_t t) {
                for (size_t w = W * t / nt; w < W * (t + 1) / nt; ++w) {
                    const size_t i = relevant[w];
                    const uint32_t start = vcf.recs[i].pos0, end = start + 1;
                    const uint32_t rs = p->window ? (uint32_t)(start - p->window) : H.ces[i], re = p->window ? (uint32_t)(end + p->window) : H.cee[i];
                    const std::string region = vcf.recs[i].chrom + ":" + std::to_string(rs) + "-" + std::to_string(re);
                    int32_t tid, beg, en;
                    if (!parse_region(P.hdr, region.c_str(), tid, beg, en) || tid >= bi.n_ref || en < beg) { bad[t] = w; return; }
                    w_tid[w] = tid; w_beg[w] = beg; w_end[w] = en;
                    if (by_seeks) w_region[w] = region;
                }
            });
            size_t w_bad = SIZE_MAX;
            // (a thread stops at its first: the first thread's is the file's first)
            for (size_t t = 0; t < nt && w_bad == SIZE_MAX; ++t) w_bad = bad[t];
            // -s XS: a read with an N operation whose strand tag lies behind an aux field of unknown type ends the process in the first window that READS it
            // (tid, pos < end, bam_endpos > beg: hts.c:1946-1957) -- bam_aux_get abort()s, sam.c:1233-1252, nothing printed -- behind that variant's echo
            if (by_seeks) {
                const uint8_t *d_file = P.d_file;
                // (a sharded extraction left no whole copy of the file in HBM: once more, unsharded)
                Prep P0;
                if (!d_file) {
                    const int rc0 = prepare_events(c, nullptr, bam.data(), bam.size(), bai.data(), bai.size(), &ep, true, P0, err, errlen);
                    if (rc0 != RGX_OK) { echo_variants(1); return rc0; }
                    d_file = P0.d_file;
                }
                w_region.resize(std::min(W, w_bad));
                size_t w_abort = SIZE_MAX;
                const int rcj = window_join_by_seeks(c, d_file, bam.size(), bai.data(), bai.size(), ep, w_region, std::min<uint32_t>(32,
                    bitlen(p->max_intron) + 2), R, S.n_pairs,
                    w_abort, err, errlen);
                if (rcj != RGX_OK) { echo_variants(1); return rcj; }
                if (w_abort != SIZE_MAX) {
                    echo_variants(w_abort + 1);
                    return fail(err, errlen, RGX_ERR_ABORT,
                        "regtools_amd: a read in the window of the variant at %s:%u has an auxiliary field of unknown type in front of its strand "
                        "tag: the reference abort()s there\n", vcf.recs[relevant[w_abort]].chrom.c_str(), vcf.recs[relevant[w_abort]].pos0 + 1);
                }
            } else if (!P.odd_aux.empty())
                for (size_t w = 0; w < std::min(W, w_bad); ++w)
                    for (const Prep::OddAux &o : P.odd_aux)
                        if (o.tid == w_tid[w] && o.pos < w_end[w] && o.end > w_beg[w]) {
                            echo_variants(w + 1);
                            return fail(err, errlen, RGX_ERR_ABORT,
                                "regtools_amd: a read at %s:%d has an auxiliary field of unknown type in front of its strand tag: the reference "
                                "abort()s in this variant's window\n", vcf.recs[relevant[w]].chrom.c_str(), o.pos + 1);
                        }
            // aborts the run
            if (w_bad != SIZE_MAX) { echo_variants(w_bad + 1); return fail(err, errlen, RGX_ERR_REGION, "%s", kMsgRegion); }
        }
        echo_variants(relevant.size());
        if (int rc_died = died_reading_the_vcf()) return rc_died;
        S.n_windows = w_tid.size();
        auto jlap = [&](const char *what) { if (jtrace) { const double t = now_ms(); fprintf(stderr, "[rgx trace] join: %-22s +%8.3f ms\n", what, t - jt);
            jt = t; } };
        jlap("window regions");
        if (!by_seeks) rc = window_join(c, P, w_tid, w_beg, w_end, std::min<uint32_t>(32, bitlen(p->max_intron) + 2), R, S.n_pairs, err, errlen);
        if (rc != RGX_OK) return rc;
        S.n_window_rows = R.n;
        jlap("window_join");
        {
            std::vector<uint32_t> crank_of, vrank_of;
            string_ranks(P.hdr.names, crank_of, uj.chrom_name);
            // (a window's contig is its variant's: the variant's name ranks like the window's contig)
            std::vector<std::string> vnames; std::unordered_map<std::string, uint32_t> vidx; std::vector<uint32_t> v_name(relevant.size());
            {   // (records of a VCF come contig by contig: one table lookup per run of equal names)
                const std::string *last = nullptr; uint32_t last_idx = 0;
                for (size_t w = 0; w < relevant.size(); ++w) {
                    const std::string &cn = vcf.recs[relevant[w]].chrom;
                    if (!last || *last != cn) { auto it = vidx.find(cn); if (it == vidx.end()) { it = vidx.emplace(cn, (uint32_t)vnames.size()).first;
                        vnames.push_back(cn); } last = &cn; last_idx = it->second; }
                    v_name[w] = last_idx;
                }
            }
            string_ranks(vnames, vrank_of, uj.vchrom_name);
            jlap("keep: name ranks");
            uj.cand.reserve(R.n);
            for (size_t r = 0; r < R.n; ++r) {
                const size_t vi = relevant[R.group[r]];
                const uint32_t ces = H.ces[vi], cee = H.cee[vi];
                const uint32_t js = R.start[r], je = R.end[r];
                if (!((js >= ces && js <= cee) || (je <= cee && je >= ces))) continue;           // identifier.cc:292-294
                uj.add(crank_of[(size_t)w_tid[R.group[r]]], js, je, vrank_of[v_name[R.group[r]]], vcf.recs[vi].pos0, (uint32_t)r);
            }
            jlap("keep: filter");
            uj.finish([&](uint32_t r) { return JEntry{R.ts[r], R.te[r], R.count[r], std::string(1, (char)R.strand[r]), "255,0,0", 2}; });
        }
        S.n_junctions = uj.size();
        jlap("keep: first-insert map");
        lap(S.ms_join);
    }

    if (t_vcfout.joinable()) t_vcfout.join();
    if (rc_vcf != RGX_OK) return fail(err, errlen, rc_vcf, "%s", err_vcf);
    if (getenv("REGTOOLS_AMD_TRACE")) fprintf(stderr, "[rgx trace] outputs: annotated VCF (side thread) %8.3f ms\n", vcf_ms);
    lap(S.ms_output);                                           // (what of the VCF was still being written when the join was done)
    rc = write_junction_outputs(c, g, p->fasta_path, uj, p->out_tsv, p->out_bed, &S.exon_visits_junctions, &S.ms_annotate, err, errlen, p->echo != 0);
    if (rc != RGX_OK) return rc;
    { const double t = now_ms(); S.ms_output += t - tl - S.ms_annotate; tl = t; }
    ktime_collect(c); S.ms_k_variant_scan = c->kms[0]; S.ms_k_junction_scan = c->kms[1]; S.ms_k_window_pairs = c->kms[2];
    S.ms_total = now_ms() - t0;
    if (stats) *stats = S;
    // The outputs are written.  What is left is teardown -- unmapping the BAM (8.5 ms for 533 MB: page tables), the annotation's tables and its
    // block of HBM (4 ms), the VCF's lines and strings (3 ms): 15 of config 4's 75 ms.  Round 4: the process's background thread does it
    // (worker_pool.h Reaper; rgx_ctx_destroy and the process's exit wait for it), the call returns.
    if (getenv("REGTOOLS_AMD_TRACE")) { fprintf(stderr, "[rgx trace] identify: total %8.3f ms\n", S.ms_total); teardown.t = now_ms(); }
    {
        VariantStage *vs = V_holder.release();
        rgx_gtf *gg = guard.g; guard.g = nullptr;
        // (the junction table and the index image go the same way, and FIRST: freed by this thread, their blocks' munmap waited for the address-space
        //  lock behind the background thread's unmapping of the 533 MB BAM -- 5-7 ms of the caller's time between "total" and the return, round 6)
        JMap *ujh = new JMap(std::move(uj));
        std::vector<uint8_t> *baih = new std::vector<uint8_t>(std::move(bai));
        Reaper::get().later([ujh, baih] { delete ujh; delete baih; });
        bam.release_later();
        // (host memory only: the annotation's device tables are the context's, gtf_upload(pooled) -- no HIP call on that thread)
        Reaper::get().later([vs, gg] { delete vs; rgx_gtf_free(gg); });
    }
    return RGX_OK;
}

// ---- `cis-splice-effects associate` (cis_splice_effects_associator.cc:234-276): identify with p->bed_path set -------------------
extern "C" int rgx_associate(rgx_ctx *c, const rgx_identify_params *p, rgx_identify_stats *stats, char *err, size_t errlen) {
    if (!c || !p || !p->bed_path) return fail(err, errlen, RGX_ERR_ARG, "Error parsing inputs!(2)\n\n");
    rgx_identify_params q = *p;
    q.bam_path = p->bed_path;          // only checked for presence on this branch
    q.strandness = 0;
    return rgx_identify(c, &q, stats, err, errlen);
}

// ---- `variants annotate` (variants_annotator.cc:541-550) ---------------------------------------------------------------------------
extern "C" int rgx_variants_annotate(rgx_ctx *c, const rgx_identify_params *p, rgx_identify_stats *stats, char *err, size_t errlen) {
    if (!c || !p || !p->vcf_path || !p->gtf_path) return fail(err, errlen, RGX_ERR_ARG, "Error parsing inputs!(2)\n\n");
    rgx_identify_stats S; memset(&S, 0, sizeof S);
    ktime_collect(c); c->kms[0] = c->kms[1] = c->kms[2] = 0;
    const double t0 = now_ms();
    // the VCF is read on a side thread while this one parses the GTF (as `identify` reads its inputs side by side); the GTF's error comes first
    rgx_gtf *g = new rgx_gtf();
    g->ctx = c;
    struct GtfGuard { rgx_gtf *g; ~GtfGuard() { rgx_gtf_free(g); } } guard{g};
    VariantStage V;
    std::string vcf_err;
    std::thread t_vcf([&] { try { vcf_err = V.vcf.load(p->vcf_path);
        } catch (const std::exception &e) { vcf_err = std::string("regtools_amd: ") + e.what() + "\n"; } });
    struct JoinOne { std::thread &t; ~JoinOne() { if (t.joinable()) t.join(); } } join_vcf{t_vcf};
    {
        std::string e;
        try { e = g->m.load(p->gtf_path); } catch (const std::exception &x) { e = std::string("regtools_amd: ") + x.what() + "\n"; }
        if (!e.empty()) return fail(err, errlen, RGX_ERR_FORMAT, "%s", e.c_str());
    }
    int rc = gtf_upload(c, g, err, errlen, /*pooled=*/true);
    if (rc != RGX_OK) return rc;
    S.ms_gtf = now_ms() - t0;
    VariantOpts vo{p->intronic_min, p->exonic_min, p->all_intronic, p->all_exonic, p->skip_single};
    t_vcf.join();
    if (!vcf_err.empty()) return fail(err, errlen, V.vcf.death == 2 ? RGX_ERR_ABORT : V.vcf.death == 1 ? RGX_ERR_EXIT : RGX_ERR_OPEN, "%s", vcf_err.c_str());
    rc = variant_scan_stage(c, g, vo, V, &S.exon_visits_variants, err, errlen);
    if (rc != RGX_OK) return rc;
    S.n_variants = V.vcf.recs.size(); S.n_relevant = V.relevant.size();
    S.ms_variants = now_ms() - t0 - S.ms_gtf;
    rc = write_annotated_vcf(p->out_vcf, V, true, err, errlen);
    if (rc != RGX_OK) return rc;
    ktime_collect(c); S.ms_k_variant_scan = c->kms[0]; S.ms_k_junction_scan = c->kms[1]; S.ms_k_window_pairs = c->kms[2];
    S.ms_total = now_ms() - t0; S.ms_output = S.ms_total - S.ms_gtf - S.ms_variants;
    if (stats) *stats = S;
    return RGX_OK;
}

// ---- `junctions annotate` (junctions_main.cc:62-93) --------------------------------------------------------------------------------
extern "C" int rgx_junctions_annotate(rgx_ctx *c, const char *bed_path, const char *fasta_path, const char *gtf_path, const char *out_path, uint64_t *n_rows,
                                      char *err, size_t errlen) {
    return rgx_junctions_annotate_opts(c, bed_path, fasta_path, gtf_path, out_path, 0, n_rows, err, errlen);
}
// include_single_exon: -S (junctions_annotator.cc:392-393: skip_single_exon_genes_ = false)
extern "C" int rgx_junctions_annotate_opts(rgx_ctx *c, const char *bed_path, const char *fasta_path, const char *gtf_path, const char *out_path,
    int options,
                                           uint64_t *n_rows, char *err, size_t errlen) {
    const int include_single_exon = options & RGX_ANNOTATE_SINGLE_EXON;
    const bool echo = (options & RGX_ANNOTATE_ECHO) != 0;
    if (!c || !bed_path || !fasta_path || !gtf_path) return fail(err, errlen, RGX_ERR_ARG, "Error parsing inputs!(2)\n\n");
    rgx_gtf *g = nullptr;
    int rc = rgx_gtf_load(c, gtf_path, &g, err, errlen);
    if (rc != RGX_OK) return rc;
    struct GtfGuard { rgx_gtf *g; ~GtfGuard() { rgx_gtf_free(g); } } guard{g};
    FILE *fo = out_path ? fopen(out_path, "w") : stdout;
    if (!fo) return fail(err, errlen, RGX_ERR_OPEN, "Unable to open %s", out_path);
    fprintf(fo, "%s\n", kJunctionHeader);
    BedJunctions B;
    const std::string bed_err = B.load(bed_path);                 // rows before a bad line are still annotated and printed, as upstream
    const Fasta *fap = host_fasta(c, fasta_path);
    const bool have_fa = fap != nullptr;
    const size_t n = B.n();
    std::vector<int32_t> jc(n); std::vector<uint8_t> jst(n);
    for (size_t i = 0; i < n; ++i) { jc[i] = g->m.chrom_of(B.chrom[i]); jst[i] = B.strand[i].size() == 1 ? (uint8_t)B.strand[i][0] : (uint8_t)'?'; }
    JunctionAnnotHost A;
    rc = annotate_junctions(c, g, jc, B.start, B.end, jst, A, err, errlen, nullptr, include_single_exon != 0);
    size_t done = 0;
    for (size_t i = 0; i < n && rc == RGX_OK; ++i) {
        std::string site;
        std::string said;
        if (!have_fa) { rc = fail(err, errlen, RGX_ERR_FASTA, "Unable to extract FASTA sequence for position %s:%u-%u\n\n", B.chrom[i].c_str(),
            B.start[i] + 1, B.start[i] + 2); if (echo) { append_positions(said, B.chrom[i], B.start[i], B.end[i], false); fputs(said.c_str(), stderr);
                } break; }
        rc = splice_site(*fap, B.chrom[i], B.start[i], B.end[i], B.strand[i], site, err, errlen);
        if (echo) {
            std::string tmp;
            append_positions(said, B.chrom[i], B.start[i], B.end[i], rc == RGX_OK || fap->fetch(B.chrom[i], (int64_t)B.start[i] + 1, (int64_t)B.start[i] + 2,
                tmp));
            fputs(said.c_str(), stderr);
        }
        if (rc != RGX_OK) break;
        print_junction_row(fo, g, A, i, B.chrom[i], B.start[i], B.end[i], B.name[i], B.score[i], B.strand[i], site);
        fputc('\n', fo);
        ++done;
    }
    if (fo != stdout) fclose(fo);
    if (n_rows) *n_rows = done;
    if (rc == RGX_OK && !bed_err.empty()) return fail(err, errlen, RGX_ERR_FORMAT, "%s", bed_err.c_str());
    return rc;
}
