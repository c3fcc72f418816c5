// api_ctx.cpp -- contexts and their streams, result tables (one block per table, a small cache of released blocks) and their BED12 / barcode text.
#include "api_internal.h"

AllocStats g_alloc_stats;                                          // (REGTOOLS_AMD_TRACE: what growing the device buffers cost a call)

// The FASTA at `path`, mapped (cse_host.h).  A context keeps the last one: what a call's 10^5 two-base lookups cost is mostly page-table work --
// faulting the pages in (sixteen per fault) and, dearer, taking two million entries down again when the mapping goes (11 ms of config 4's
// `identify`) -- and a caller that runs one sample after the other against the same genome pays both once.  The file is recognised by device,
// inode, size and modification time; anything else is a new file.  nullptr = it cannot be opened.
rgx::Fasta *host_fasta(rgx_ctx *c, const char *path) {
    struct stat st;
    if (!path || stat(path, &st) != 0) return nullptr;
    const uint64_t key[4] = {(uint64_t)st.st_dev, (uint64_t)st.st_ino, (uint64_t)st.st_size,
        (uint64_t)st.st_mtim.tv_sec * 1000000000ull + (uint64_t)st.st_mtim.tv_nsec};
    if (c->host_fasta && c->host_fasta_path == path && !memcmp(key, c->host_fasta_key, sizeof key)) return c->host_fasta;
    delete c->host_fasta; c->host_fasta = nullptr;
    rgx::Fasta *f = new rgx::Fasta();
    if (!f->load(path)) { delete f; return nullptr; }
    c->host_fasta = f; c->host_fasta_path = path; memcpy(c->host_fasta_key, key, sizeof key);
    return f;
}

void ktime_begin(rgx_ctx *c, int slot) {
    hipEvent_t e[2];
    for (auto &x : e) { if (!c->kfree.empty()) { x = c->kfree.back(); c->kfree.pop_back(); } else if (hipEventCreate(&x) != hipSuccess) return; }
    (void)hipEventRecord(e[0], c->stream);
    c->kpend.push_back({e[0], e[1], slot});
}
void ktime_end(rgx_ctx *c) { if (!c->kpend.empty()) (void)hipEventRecord(c->kpend.back().b, c->stream); }
void ktime_collect(rgx_ctx *c) {
    for (auto &k : c->kpend) {
        float ms = 0;
        if (hipEventSynchronize(k.b) == hipSuccess && hipEventElapsedTime(&ms, k.a, k.b) == hipSuccess) c->kms[k.slot] += ms;
        c->kfree.push_back(k.a); c->kfree.push_back(k.b);
    }
    c->kpend.clear();
}

extern "C" const char *rgx_version(void) { return "regtools_amd 0.1 gfx950"; }

extern "C" void rgx_extract_params_default(rgx_extract_params *p) {
    memset(p, 0, sizeof *p);
    p->region = "."; p->strandness = -1; p->strand_tag[0] = 'X'; p->strand_tag[1] = 'S';
    p->min_anchor = 8; p->min_intron = 70; p->max_intron = 500000; p->fasta_path = nullptr; p->shard = 0; p->n_shards = 1;
    p->barcodes = 0; p->barcode_tag[0] = 'C'; p->barcode_tag[1] = 'B';
}

extern "C" int rgx_ctx_create(int device, rgx_ctx **out, char *err, size_t errlen) {
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(err, errlen, RGX_ERR_NO_DEVICE, "regtools_amd: no HIP device visible; this library has no CPU fallback\n");
    if (device < 0 || device >= n) return fail(err, errlen, RGX_ERR_NO_DEVICE, "regtools_amd: device %d out of range (%d visible)\n", device, n);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (!strstr(prop.gcnArchName, "gfx950"))
        return fail(err, errlen, RGX_ERR_NO_DEVICE, "regtools_amd: device %d is %s; the kernels are built for gfx950 only\n", device, prop.gcnArchName);
    rgx_ctx *c = new rgx_ctx();
    c->device = device;
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    { const char *e = getenv("REGTOOLS_AMD_ONE_SHOT"); c->one_shot = e && strcmp(e, "0") != 0; }
    c->buf("arena").piece = arena_knobs().piece;                     // (DevBuf::map_pieces: what the DEFLATE launch writes into)
    for (auto &e : c->ev) HIP_TRY(hipEventCreate(&e));
    for (auto &e : c->ev_launch) HIP_TRY(hipEventCreate(&e));
    for (auto &e : c->ev_trial) HIP_TRY(hipEventCreate(&e));
    HIP_TRY(hipHostMalloc(&c->pinned, 4096, hipHostMallocDefault));
    c->pinned_cap = 4096;
    *out = c;
    return RGX_OK;
}

// the copy stream and the side streams of the overlapped upload: made when a call first takes that path (a one-shot process that reads a
// small file never pays for them; eight stream creations are ~100 ms of a cold start)
hipError_t ensure_upload_streams(rgx_ctx *c) {
    if (c->copy_stream || c->ev_ready) return hipSuccess;
    // A process that makes one call (bin/regtools-amd: REGTOOLS_AMD_ONE_SHOT, set by its main()) does without streams of its own: creating
    // the copy stream and two side streams costs 24-30 ms (8-10 ms per hardware queue), the overlap they buy -- upload under inflate, three inflate
    // launches side by side -- 5 ms of a call: the file goes up in one piece on the context's stream, one inflate launch follows it.
    if (c->one_shot) return hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming);
    hipError_t e = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking);
    if (e != hipSuccess) return e;
    // (the runtime maps the streams of one priority onto four hardware queues)
    // two side streams + the pipeline's own (measured and not kept, round 4: a third one of the greatest priority for the early tail's launch, +0.8 ms;
    // four to seven pieces on streams of other priorities, no gain: DESIGN.md 4.4)
    for (int k = 0; k < kSideStreams; ++k) {
        if ((e = hipStreamCreateWithPriority(&c->side[k], hipStreamNonBlocking, 0)) != hipSuccess) return e;
        if ((e = hipEventCreateWithFlags(&c->ev_side[k], hipEventDisableTiming)) != hipSuccess) return e;
    }
    return hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming);
}

// (multi.cpp: a context made for a device that a device list names a second time -- shards taking turns on one GPU, a test configuration -- does without the
// trials:
//  several contexts of one device would each hold a second arena at the same time)
void rgx_ctx_no_arena_trials(rgx_ctx *c) { if (c) c->arena_calibrated_bytes = UINT64_MAX; }
// pipeline.cpp: the contexts of one pipeline take the host link in turns
void *rgx_link_turn_create(int depth) {
    LinkTurn *l = new LinkTurn;
    // (what the environment held when HIP started is what the runtime uses; a value set later is only a wrong guess about it, and either way is correct)
    const char *q = getenv("GPU_MAX_HW_QUEUES");
    // (eight queues per file in flight: three files on sixteen queues with two launches at once took 90-93 ms per file -- some of their twelve-odd streams
    //  share a queue again; in turns they take 20.4-22.2, and on thirty-two queues 19.1-19.5 at once)
    l->chip.width = q && atoi(q) >= 8 * std::max(2, depth) ? 2 : 1;
    return l;
}
void rgx_link_turn_destroy(void *l) { delete (LinkTurn *)l; }
void rgx_ctx_set_link(rgx_ctx *c, void *l) { if (c) c->link = (LinkTurn *)l; }

extern "C" int rgx_ctx_arena_trials(const rgx_ctx *c, float *ms, int cap) {
    if (!c) return 0;
    for (int k = 0; k < c->arena_trials && k < cap; ++k) ms[k] = c->arena_trial_ms[k];
    return c->arena_trials;
}

extern "C" void rgx_ctx_destroy(rgx_ctx *c) {
    if (!c) return;
    Reaper::get().drain();                                  // (deferred teardown of finished calls may still hold memory of this device)
    (void)hipSetDevice(c->device);
    for (auto &kv : c->bufs) kv.second.release();
    if (c->arena_retired) { c->arena_retired->release(); delete c->arena_retired; }
    for (auto &e : c->ev) if (e) (void)hipEventDestroy(e);
    for (auto &e : c->ev_launch) if (e) (void)hipEventDestroy(e);
    for (auto &e : c->ev_trial) if (e) (void)hipEventDestroy(e);
    for (auto &e : c->chunk_ev) if (e) (void)hipEventDestroy(e);
    ktime_collect(c);
    for (auto &e : c->kfree) (void)hipEventDestroy(e);
    for (auto &e : c->ev_side) if (e) (void)hipEventDestroy(e);
    if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
    if (c->ev_packed) (void)hipEventDestroy(c->ev_packed);
    for (auto &q : c->side) if (q) (void)hipStreamDestroy(q);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    delete c->fasta;
    delete c->host_fasta;
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->pinned_rows) (void)hipHostFree(c->pinned_rows);
    if (c->pinned_members) (void)hipHostFree(c->pinned_members);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// ---- table plumbing --------------------------------------------------------------------------------------
static std::mutex g_block_mu;
static std::vector<CachedBlock> g_blocks;                        // released blocks, at most kBlockCacheEntries / kBlockCacheBytes
static const size_t kBlockCacheEntries = 6, kBlockCacheBytes = (size_t)1 << 30;

// pinned = page-locked (hipHostMalloc): the device pipelines copy the finished columns straight into the block
void *block_take(size_t need, size_t &cap, bool pinned) {
    {
        std::lock_guard<std::mutex> lk(g_block_mu);
        size_t best = g_blocks.size();
        for (size_t i = 0; i < g_blocks.size(); ++i)
            if (g_blocks[i].pinned == pinned && g_blocks[i].cap >= need && g_blocks[i].cap <= need * 2 + (1 << 20) &&
                (best == g_blocks.size() || g_blocks[i].cap < g_blocks[best].cap)) best = i;
        if (best != g_blocks.size()) { void *p = g_blocks[best].p; cap = g_blocks[best].cap; g_blocks.erase(g_blocks.begin() + (long)best); return p; }
    }
    cap = need;
    if (!pinned) return malloc(need);
    void *p = nullptr;
    if (hipHostMalloc(&p, need, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
void block_give(void *p, size_t cap, bool pinned) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(g_block_mu);
        size_t held = 0;
        for (auto &b : g_blocks) held += b.cap;
        if (cap >= (1 << 16) && g_blocks.size() < kBlockCacheEntries && held + cap <= kBlockCacheBytes) { g_blocks.push_back(CachedBlock{p, cap, pinned});
            return; }
    }
    if (pinned) (void)hipHostFree(p); else free(p);
}

// zero = the caller does not write every column of every row
rgx_junction_table *table_alloc(const BamHeader &h, uint64_t n, bool zero, bool pinned) {
    TableBox *box = (TableBox *)calloc(1, sizeof *box);
    rgx_junction_table *t = &box->t;
    t->n_ref = (int32_t)h.names.size();
    t->ref_name = (char **)calloc(h.names.size() + 1, sizeof(char *));
    t->ref_len = (uint32_t *)calloc(h.names.size() + 1, sizeof(uint32_t));
    for (size_t i = 0; i < h.names.size(); ++i) { t->ref_name[i] = strdup(h.names[i].c_str()); t->ref_len[i] = h.lens[i]; }
    t->n = n;
    const size_t m = table_block_rows(n);                         // every column starts 16-byte aligned; the layout launch_rows_table writes
    const size_t need = table_block_bytes(n);
    box->pinned = pinned;
    box->block = block_take(need, box->block_cap, pinned);
    if (!box->block && pinned) { box->pinned = false; box->block = block_take(need, box->block_cap, false); }
    if (!box->block) {                                           // no memory for the rows: no table (callers report RGX_ERR_DEVICE / RGX_ERR_ARG)
        for (int32_t i = 0; i < t->n_ref; ++i) free(t->ref_name[i]);
        free(t->ref_name); free(t->ref_len); free(box);
        return nullptr;
    }
    if (zero) memset(box->block, 0, need);
    uint8_t *q = (uint8_t *)box->block;
    t->name_index = (uint64_t *)q; q += m * 8; t->first_seen = (uint64_t *)q; q += m * 8; t->last_seen = (uint64_t *)q; q += m * 8;
    t->tid = (int32_t *)q; q += m * 4; t->start = (uint32_t *)q; q += m * 4; t->end = (uint32_t *)q; q += m * 4;
    t->thick_start = (uint32_t *)q; q += m * 4; t->thick_end = (uint32_t *)q; q += m * 4; t->read_count = (uint32_t *)q; q += m * 4;
    t->strand = (char *)q; q += m; t->left_ok = q; q += m; t->right_ok = q;
    return t;
}

extern "C" void rgx_table_free(rgx_junction_table *t) {
    if (!t) return;
    TableBox *box = (TableBox *)t;                                   // t is the first member
    if (t->ref_name) for (int32_t i = 0; i < t->n_ref; ++i) free(t->ref_name[i]);
    free(t->ref_name); free(t->ref_len);
    block_give(box->block, box->block_cap, box->pinned);
    free(t->bc_row_begin); free(t->bc_count); free(t->bc_str_begin); free(t->bc_text); free(t->bc_insert_rank);
    free(box);
}

// compare_junctions (junctions_extractor.h:117-140): chrom string, thick_start, thick_end, name string
void host_sort_rows(rgx_junction_table *t) {
    // compare_junctions (junctions_extractor.h:117-140): chrom string, thick_start, thick_end, name string.  Names are "JUNC%08d":
    // below 10^8 the string order is the numeric order; beyond, the longer decimal strings are compared as text.
    std::vector<uint32_t> crank((size_t)std::max(t->n_ref, 1), 0);
    {
        std::vector<int32_t> order((size_t)t->n_ref);
        for (int32_t i = 0; i < t->n_ref; ++i) order[(size_t)i] = i;
        std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return strcmp(t->ref_name[a], t->ref_name[b]) < 0; });
        uint32_t rk = 0;
        for (int32_t i = 0; i < t->n_ref; ++i) { if (i > 0 && strcmp(t->ref_name[order[(size_t)i]], t->ref_name[order[(size_t)i - 1]]) != 0) ++rk;
            crank[(size_t)order[(size_t)i]] = rk; }
    }
    auto name_less = [](uint64_t a, uint64_t b) {
        if (a < 100000000ull && b < 100000000ull) return a < b;
        char na[32], nb[32];
        snprintf(na, sizeof na, "%08llu", (unsigned long long)a); snprintf(nb, sizeof nb, "%08llu", (unsigned long long)b);
        return strcmp(na, nb) < 0;
    };
    std::vector<uint64_t> idx(t->n);
    for (uint64_t i = 0; i < t->n; ++i) idx[i] = i;
    std::sort(idx.begin(), idx.end(), [&](uint64_t a, uint64_t b) {
        const uint32_t ca = crank[(size_t)t->tid[a]], cb = crank[(size_t)t->tid[b]];
        if (ca != cb) return ca < cb;
        if (t->thick_start[a] != t->thick_start[b]) return t->thick_start[a] < t->thick_start[b];
        if (t->thick_end[a] != t->thick_end[b]) return t->thick_end[a] < t->thick_end[b];
        return name_less(t->name_index[a], t->name_index[b]);
    });
    auto permute = [&](auto *col) {
        typedef typename std::remove_reference<decltype(col[0])>::type T;
        std::vector<T> tmp(t->n);
        for (uint64_t i = 0; i < t->n; ++i) tmp[i] = col[idx[i]];
        memcpy(col, tmp.data(), sizeof(T) * t->n);
    };
    permute(t->tid); permute(t->start); permute(t->end); permute(t->thick_start); permute(t->thick_end); permute(t->read_count);
    permute(t->name_index); permute(t->strand); permute(t->left_ok); permute(t->right_ok); permute(t->first_seen); permute(t->last_seen);
}

// Junction::print (junctions_extractor.h:90-98) for rows [r0, r1): appended to `out`.  The name is copied as it is: the BAM header puts no
// limit on its length.
void format_bed12_rows(const rgx_junction_table *t, int only_anchored, uint64_t r0, uint64_t r1, std::string &out) {
    char tail[256];                                        // everything behind the contig name: ten bounded numeric fields
    for (uint64_t i = r0; i < r1; ++i) {
        if (only_anchored && !(t->left_ok[i] && t->right_ok[i])) continue;
        const char *name = t->ref_name[t->tid[i]];
        const int n = snprintf(tail, sizeof tail, "\t%u\t%u\tJUNC%08llu\t%u\t%c\t%u\t%u\t255,0,0\t2\t%u,%u\t0,%u\n",
                               t->thick_start[i], t->thick_end[i], (unsigned long long)t->name_index[i], t->read_count[i], t->strand[i],
                               t->thick_start[i], t->thick_end[i], (uint32_t)(t->start[i] - t->thick_start[i]),
                               (uint32_t)(t->thick_end[i] - t->end[i]), (uint32_t)(t->end[i] - t->thick_start[i]));
        out.append(name); out.append(tail, (size_t)n);
    }
}

extern "C" size_t rgx_table_format_bed12(const rgx_junction_table *t, int only_anchored, char *buf, size_t cap) {
    // text work of ~170 ns per row on one core: ranges of rows on the host's cores (55 -> 6 ms for 300 k rows)
    const unsigned n_thr = t->n >= 20000 ? std::max(1u, std::min<unsigned>(usable_threads(16), (unsigned)(t->n / 8192))) : 1u;
    std::vector<std::string> part(n_thr);
    if (n_thr == 1) format_bed12_rows(t, only_anchored, 0, t->n, part[0]);
    else {
        std::vector<std::thread> pool;
        for (unsigned w = 0; w < n_thr; ++w)
            pool.emplace_back([&, w] { part[w].reserve((size_t)(t->n / n_thr + 1) * 96); format_bed12_rows(t, only_anchored, t->n * w / n_thr,
                t->n * (w + 1) / n_thr, part[w]); });
        for (auto &th : pool) th.join();
    }
    size_t need = 0;
    for (const std::string &q : part) need += q.size();
    if (buf && need <= cap) { size_t o = 0; for (const std::string &q : part) { memcpy(buf + o, q.data(), q.size()); o += q.size(); } }
    return need;
}

extern "C" size_t rgx_table_format_barcodes(const rgx_junction_table *t, int only_anchored, char *buf, size_t cap) {
    size_t need = 0;
    auto put = [&](const char *s, size_t n) { if (buf && need + n <= cap) memcpy(buf + need, s, n); need += n; };
    char num[32];
    for (uint64_t i = 0; i < t->n; ++i) {
        if (only_anchored && !(t->left_ok[i] && t->right_ok[i])) continue;
        const uint64_t b = t->bc_row_begin ? t->bc_row_begin[i] : 0, e = t->bc_row_begin ? t->bc_row_begin[i + 1] : 0;
        put(num, (size_t)snprintf(num, sizeof num, "%llu\t", (unsigned long long)(e - b)));     // Junction::print_barcodes (h:103-110)
        for (uint64_t k = b; k < e; ++k) {
            if (k != b) put(",", 1);
            put(t->bc_text + t->bc_str_begin[k], (size_t)(t->bc_str_begin[k + 1] - t->bc_str_begin[k]));
            put(num, (size_t)snprintf(num, sizeof num, ":%u", t->bc_count[k]));
        }
        put("\n", 1);
    }
    return need;
}

// ---- the pipeline ------------------------------------------------------------------------------------------------
