// pipeline.cpp -- several files in flight on ONE device: rgx_pipeline_create / rgx_extract_submit / rgx_extract_wait (include/regtools_amd.h).
//
// Replaces the loop a cohort run makes around `regtools junctions extract` -- one process, one BAM, one after the other
// (/root/reference/src/junctions/junctions_main.cc:45-59).  One call of rgx_extract_mem leaves the device idle while a file's first chunks cross the
// link and leaves the link idle during the file's tail (link 10 ms + a late member's chain 8-9 ms + tail 3.4 ms for ~20 ms of device
// work).  A pipeline owns `depth` contexts on the device, each with its own streams and workspace and a host thread that runs the ordinary call on it;
// file k goes to context k mod depth, so file k+1's upload and arrival-gated inflate run under file k's tail.  Nothing else changes: every file is one
// rgx_extract_mem call, so N interleaved files are N sequential calls byte for byte (tests/test_gpu_pipeline.py), a damaged file in the middle included.
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/regtools_amd.h"

// api_ctx.cpp: the contexts of a pipeline take the host link in turns (a call's upload starts when the call before it has its file on the device)
void *rgx_link_turn_create(int depth);
void rgx_link_turn_destroy(void *l);
void rgx_ctx_set_link(rgx_ctx *c, void *l);

namespace {

struct Job {
    uint64_t ticket = 0;
    const void *bam = nullptr, *bai = nullptr; size_t bam_len = 0, bai_len = 0;
    rgx_extract_params params{};
    std::string region, fasta;                       // the params' strings, owned (the caller's may go away behind submit)
    rgx_junction_table *table = nullptr;
    int rc = RGX_OK; std::string err;
    bool done = false, claimed = false;
};

struct Lane {                                        // one context + the thread that runs calls on it
    rgx_ctx *ctx = nullptr;
    std::thread th;
    std::deque<std::shared_ptr<Job>> q;
};

}  // namespace

struct rgx_pipeline {
    std::mutex mu;
    std::condition_variable work, finished;
    std::vector<Lane> lanes;
    std::deque<std::shared_ptr<Job>> open;           // submitted and not yet waited for, in ticket order
    uint64_t next_ticket = 1;
    bool stopping = false;
    void *link = nullptr;
};

static void lane_loop(rgx_pipeline *pl, size_t k) {
    Lane &ln = pl->lanes[k];
    for (;;) {
        std::shared_ptr<Job> j;
        {
            std::unique_lock<std::mutex> lock(pl->mu);
            pl->work.wait(lock, [&] { return pl->stopping || !ln.q.empty(); });
            if (ln.q.empty()) return;                // (stopping, and nothing left to run)
            j = ln.q.front(); ln.q.pop_front();
        }
        char err[512] = {0};
        static const bool trace = getenv("REGTOOLS_AMD_TRACE") != nullptr;
        const auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double t_in = now();
        const int rc = rgx_extract_mem(ln.ctx, j->bam, j->bam_len, j->bai, j->bai_len, &j->params, &j->table, err, sizeof err);
        if (trace) fprintf(stderr, "[rgx trace] pipeline: ticket %llu on context %zu from %.3f to %.3f ms (%.3f)\n", (unsigned long long)j->ticket, k,
            fmod(t_in, 1e5), fmod(now(), 1e5), now() - t_in);
        {
            std::lock_guard<std::mutex> lock(pl->mu);
            j->rc = rc; j->err = err; j->done = true;
        }
        pl->finished.notify_all();
    }
}

extern "C" int rgx_pipeline_create(int device, int depth, rgx_pipeline **out, char *err, size_t errlen) {
    if (!out || depth < 1 || depth > 8) { if (err && errlen) snprintf(err, errlen, "regtools_amd: a pipeline holds 1 to 8 files in flight\n");
        return RGX_ERR_ARG; }
    // Every context has four streams, and the runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 unless the environment says otherwise,
    // read when HIP starts).  Two contexts share queues pairwise and work; with three and more, a file's arrival-gated waves and the one-lane kernels that
    // release them end up behind one another in ONE hardware queue and every such wave waits out its 2 s time-out (measured: 130-240 ms per file instead
    // of 21).  A hardware queue per stream is what was seen to be safe: four files in flight on eight queues still timed out on a payload whose launches
    // are long (profiles/r06_sustained_pipeline_ab.txt).
    if (depth > 2) {
        const char *q = getenv("GPU_MAX_HW_QUEUES");
        if (!q || atoi(q) < 4 * depth) {
            if (err && errlen) snprintf(err, errlen,
                "regtools_amd: %d files in flight need GPU_MAX_HW_QUEUES=%d or more in the environment before HIP starts\n", depth, 4 * depth);
            return RGX_ERR_ARG;
        }
    }
    std::unique_ptr<rgx_pipeline> pl(new rgx_pipeline);
    pl->lanes.resize((size_t)depth);
    for (int k = 0; k < depth; ++k) {
        const int rc = rgx_ctx_create(device, &pl->lanes[(size_t)k].ctx, err, errlen);
        if (rc != RGX_OK) { for (Lane &ln : pl->lanes) if (ln.ctx) rgx_ctx_destroy(ln.ctx); return rc; }
    }
    if (depth > 1) { pl->link = rgx_link_turn_create(depth); for (Lane &ln : pl->lanes) rgx_ctx_set_link(ln.ctx, pl->link); }
    for (size_t k = 0; k < pl->lanes.size(); ++k) pl->lanes[k].th = std::thread(lane_loop, pl.get(), k);
    *out = pl.release();
    return RGX_OK;
}

extern "C" int rgx_pipeline_depth(const rgx_pipeline *pl) { return pl ? (int)pl->lanes.size() : 0; }

// the context file `ticket` runs on: its rows stay in that context's HBM until the file `depth` tickets later starts there (rgx_last_table_pack_device,
// rgx_table_merge_device: how a rank merges file k with the other ranks' while file k+1 is already on its way up)
extern "C" rgx_ctx *rgx_pipeline_ctx(const rgx_pipeline *pl, uint64_t ticket) {
    return pl && ticket ? pl->lanes[(size_t)((ticket - 1) % pl->lanes.size())].ctx : nullptr;
}

extern "C" int rgx_extract_submit(rgx_pipeline *pl, const void *bam, size_t bam_len, const void *bai, size_t bai_len, const rgx_extract_params *p,
                                  uint64_t *ticket, char *err, size_t errlen) {
    if (!pl || !p || !ticket) { if (err && errlen) snprintf(err, errlen, "regtools_amd: rgx_extract_submit needs a pipeline, parameters and a ticket\n");
        return RGX_ERR_ARG; }
    std::shared_ptr<Job> j(new Job);
    j->bam = bam; j->bam_len = bam_len; j->bai = bai; j->bai_len = bai_len; j->params = *p;
    if (p->region) { j->region = p->region; j->params.region = j->region.c_str(); }
    if (p->fasta_path) { j->fasta = p->fasta_path; j->params.fasta_path = j->fasta.c_str(); }
    {
        std::lock_guard<std::mutex> lock(pl->mu);
        if (pl->stopping) return RGX_ERR_ARG;
        j->ticket = pl->next_ticket++;
        // file k -> context k mod depth: which context a file meets does not depend on timing
        pl->lanes[(size_t)((j->ticket - 1) % pl->lanes.size())].q.push_back(j);
        pl->open.push_back(j);
        *ticket = j->ticket;
    }
    pl->work.notify_all();
    return RGX_OK;
}

extern "C" int rgx_extract_wait(rgx_pipeline *pl, uint64_t ticket, rgx_junction_table **out, char *err, size_t errlen) {
    if (!pl || !out) return RGX_ERR_ARG;
    *out = nullptr;
    std::shared_ptr<Job> j;
    {
        std::unique_lock<std::mutex> lock(pl->mu);
        for (auto &o : pl->open) if (o->ticket == ticket && !o->claimed) { j = o; break; }
        if (!j) { if (err && errlen) snprintf(err, errlen, "regtools_amd: no file in flight under ticket %llu\n", (unsigned long long)ticket);
            return RGX_ERR_ARG; }
        j->claimed = true;
        pl->finished.wait(lock, [&] { return j->done; });
        for (auto it = pl->open.begin(); it != pl->open.end(); ++it) if (it->get() == j.get()) { pl->open.erase(it); break; }
    }
    if (j->rc != RGX_OK && err && errlen) snprintf(err, errlen, "%s", j->err.c_str());
    *out = j->table;
    return j->rc;
}

extern "C" void rgx_pipeline_destroy(rgx_pipeline *pl) {
    if (!pl) return;
    { std::lock_guard<std::mutex> lock(pl->mu); pl->stopping = true; }
    pl->work.notify_all();
    // (files still queued are run to their end: their buffers were promised to the pipeline)
    for (Lane &ln : pl->lanes) if (ln.th.joinable()) ln.th.join();
    for (auto &j : pl->open) if (j->table) rgx_table_free(j->table);     // results nobody waited for
    for (Lane &ln : pl->lanes) rgx_ctx_destroy(ln.ctx);
    if (pl->link) rgx_link_turn_destroy(pl->link);
    delete pl;
}
