// api_front.cpp -- the front of one call (EventsRun, api_internal.h): the file on its way to HBM, the BGZF member list, the DEFLATE launch.
#include "api_internal.h"

int EventsRun::run() {
    if (!p || p->strandness < 0 || p->strandness > 3) return fail(err, errlen, RGX_ERR_ARG, "Please supply strandness mode with '-s' option!\n\n");
    if (p->strandness == 3 && !p->fasta_path) return fail(err, errlen, RGX_ERR_ARG, "Strandness mode 'intron-motif' requires a fasta file!\n\n");
    HIP_ENTER(c->device);
    st = c->stream;
    copy_q = c->copy_stream ? c->copy_stream : c->stream;     // where the file's upload goes (a one-shot context: its only stream)
    t_begin = now_ms();
    trace = getenv("REGTOOLS_AMD_TRACE") != nullptr;
    t_last = t_begin;

    if (bam_len < 28) return fail(err, errlen, RGX_ERR_OPEN, "%s", kMsgOpen);
    // (the arena the last call's data lay in, after it lost its place)
    if (c->arena_retired) { c->arena_retired->release(); delete c->arena_retired; c->arena_retired = nullptr; }
    { const int rc = stage_upload(); if (rc != kGoOn) return rc; }
    { const int rc = stage_members(); if (rc != kGoOn) return rc; }
    { const int rc = stage_range_and_inflate(); if (rc != kGoOn) return rc; }
    { const int rc = stage_footers_and_header(); if (rc != kGoOn) return rc; }
    { const int rc = stage_bounds_and_chains(); if (rc != kGoOn) return rc; }
    { const int rc = stage_framing(); if (rc != kGoOn) return rc; }
    { const int rc = stage_decode(); if (rc != kGoOn) return rc; }
    const int rc_emit = stage_emit();
    if (rc_emit == RGX_OK) { const int rc = calibrate_arena(); if (rc != RGX_OK) return rc; }
    P.d_file = rc_emit == RGX_OK && p->n_shards <= 1 && (!p->region || !strcmp(p->region, ".")) ? d_bam : nullptr;      // (every byte of the file went up)
    return rc_emit;
}

// Arena placement trials (rgx_ctx above) -- OPT-IN (REGTOOLS_AMD_ARENA=5) since round 6.  On a context's first call with an arena of 2 GiB and more
// (and again when a later one is a quarter larger), once the call's own work is enqueued: the same whole-range launch, plain, into the call's arena and into
// fresh allocations ONE AT A TIME (a warm-up launch, then the median of three timed with HIP events); a challenger that beats the incumbent's median by 1.5 %
// becomes the context's arena (rgx_ctx_arena_trials reports the times).  The call's data stays where it is -- when a challenger wins, the old arena is
// retired and released by the next call.  At most ONE arena's worth of extra memory is ever held, free memory is asked for again before every candidate, and
// anything that goes wrong inside a trial leaves the incumbent in place: the caller's result is complete before the first trial starts.
static int arena_challengers() { return arena_knobs().trials; }
int EventsRun::calibrate_arena() {
    if (!arena_challengers() || c->one_shot || d_true_sizes || chunked || split_B || P.stream_ended || !n_range || n_range <= 2048 ||
        total < ((uint64_t)2 << 30)) return RGX_OK;
    if (c->arena_calibrated_bytes == UINT64_MAX || (c->arena_calibrated_bytes &&
        total + 256 <= c->arena_calibrated_bytes + c->arena_calibrated_bytes / 4)) return RGX_OK;
    // (one calibration at a time per DEVICE: the arenas of different devices have nothing to do with one another)
    static std::mutex trial_mu[16];
    std::lock_guard<std::mutex> trial_lock(trial_mu[(unsigned)c->device % 16u]);
    if (!inflate_takes_coop(n_range) || h_sc[0] != 0xffffffffu) return RGX_OK;
    DevBuf &b_arena = c->buf("arena"), &b_lens = c->buf("inflate_scratch");
    if (!b_arena.p || b_arena.cap < total + 256) return RGX_OK;
    auto room_for_one = [&] {                                  // (a challenger AND what the call -- or a co-tenant of the device -- may still allocate)
        size_t free_b = 0, total_b = 0;
        return hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b >= b_arena.cap + b_arena.cap / 8 + ((size_t)16 << 30);
    };
    const int plan = inflate_plan_for(bam_len, total_all);
    uint32_t *d_dummy = d_sc + 100;                            // (the trial launches' verdicts: not looked at -- the call's own launch gave the verdict)
    auto time_into = [&](uint8_t *arena_p, float &ms) -> hipError_t {
        float t[3] = {0, 0, 0};
        for (int k = 0; k < 4; ++k) {                          // (the first launch into a fresh allocation also pays for its pages: not timed)
            hipError_t e = hipMemsetAsync(d_dummy, 0xff, 8, st);
            if (e != hipSuccess) return e;
            if ((e = hipEventRecord(c->ev_trial[0], st)) != hipSuccess) return e;
            launch_inflate(d_bam, d_members + m_lo, n_range, arena_p, upos_lo, b_lens.as<uint32_t>(), d_dummy, st, 0, 0, false, 0, nullptr, plan);
            if ((e = hipEventRecord(c->ev_trial[1], st)) != hipSuccess) return e;
            if ((e = hipEventSynchronize(c->ev_trial[1])) != hipSuccess) return e;
            if (k && (e = hipEventElapsedTime(&t[k - 1], c->ev_trial[0], c->ev_trial[1])) != hipSuccess) return e;
        }
        std::sort(t, t + 3);
        ms = t[1];
        return hipSuccess;
    };
    c->arena_trials = 0;
    float best_ms = 0;
    // (the call's own arena: the same bytes written once more, behind everything that read them)
    if (time_into(b_arena.as<uint8_t>(), best_ms) != hipSuccess) { (void)hipGetLastError(); return RGX_OK; }
    c->arena_trial_ms[c->arena_trials++] = best_ms;
    DevBuf best;                                               // the fastest challenger so far (empty: the incumbent leads)
    for (int k = 0; k < arena_challengers(); ++k) {
        if (best.p) break;                                     // (a winner is kept at once: never two challengers' memory at a time)
        if (!room_for_one()) break;
        // (what makes one placement faster than another is not known -- DESIGN 5.5 -- so the challengers are not of one kind)
        // (a hipMalloc block never won one: 15.6-16.3 ms beside 12.7-13.6)
        static const size_t kLadder[] = {(size_t)1 << 30, (size_t)256 << 20, (size_t)512 << 20, (size_t)128 << 20, (size_t)1 << 30, (size_t)64 << 20,
            (size_t)512 << 20};
        DevBuf cand; cand.piece = b_arena.piece ? kLadder[k % 7] : 0;
        if (cand.ensure(b_arena.cap) != hipSuccess) { (void)hipGetLastError(); break; }
        float ms = 0;
        if (time_into(cand.as<uint8_t>(), ms) != hipSuccess) { (void)hipGetLastError(); cand.release(); break; }
        if (c->arena_trials < 8) c->arena_trial_ms[c->arena_trials++] = ms;
        if (ms < best_ms * 0.985f) { best = cand; best_ms = ms; } else cand.release();
    }
    if (best.p) {
        // the call's data lies in the old arena and the caller may still read it (P.arena): it is retired, not released
        if (c->arena_retired) { c->arena_retired->release(); delete c->arena_retired; }
        c->arena_retired = new DevBuf(b_arena);
        b_arena = best;
        b_arena.piece = arena_knobs().piece;                   // (a later regrow is made of the configured pieces, not of the winner's ladder size)
    }
    c->arena_calibrated_bytes = b_arena.cap;
    if (trace) {
        fprintf(stderr, "[rgx trace] arena placement: call's arena %.3f ms", c->arena_trial_ms[0]);
        for (int k = 1; k < c->arena_trials; ++k) fprintf(stderr, ", %.3f", c->arena_trial_ms[k]);
        fprintf(stderr, " -> %s\n", best.p ? "a challenger kept" : "kept");
    }
    mark("arena placement trial");
    return RGX_OK;
}

int EventsRun::stage_upload() {
    // -- index: ~1 ms of host parsing for a 5 MB .bai, done on a second host thread while this one feeds the device the member scan --
    bai_thread = std::thread([this] {
        bai_ok = bai && normalize_index(bai, bai_len, index_image, bai, bai_len) && parse_bai(bai, bai_len, bi, /*collect_anchors=*/false);
    });
    // -- upload ----------------------------------------------------------------------------------------------------------
    // Host input: the file goes up in chunks on the copy stream from a helper thread (a pageable source makes hipMemcpyAsync block), while
    // this thread finds the members on the host (scan_members_parallel) -- the inflate of chunk k's members then runs while chunk k+1 is
    // still on the bus (SURVEY 8d times the path from file bytes in host memory).  A file the host scan does not vouch for waits for the
    // whole upload and takes the device's member discovery, as does device input.
    d_bam = d_bam_in;
    // the host scan's member list (overlap only; the context keeps its pages: a fresh 4 MB is a thousand page faults per call)
    std::vector<Member> &hm = c->hm_scratch;
    hm.clear();
    if (!d_bam) {
        DevBuf &b = c->buf("bam");
        HIP_TRY(b.ensure(bam_len + 64));
        d_bam = b.as<uint8_t>();
        mark("file buffer in HBM");
        if (c->link) { c->wire_hold.take(&c->link->wire); mark("the link is ours"); }
        const size_t overlap_min = overlap_knobs().min_bytes;
        if (allow_overlap && !d_true_sizes && bam_len >= overlap_min) {
            HIP_TRY(ensure_upload_streams(c));
            if (c->copy_stream) copy_q = c->copy_stream;
            mark("upload streams");
            // A shard of a file whose members the caller scanned: only the bytes this shard reads go up -- the header's members and the
            // range between its two cuts (the same cuts as below, from the index) -- N shards then move the file once, not N times.
            size_t up_lo = 0, up_hi = bam_len, hdr_hi = 0;
            if (shared && p->n_shards > 1 && p->shard >= 0 && p->shard < p->n_shards) {
                if (bai_thread.joinable()) bai_thread.join();
                BamHeader hh; size_t hb = 0;
                if (bai_ok && host_bam_header(h_bam, std::min<size_t>(bam_len, (size_t)8 << 20), hh, &hb)) {
                    const bool rest0 = p->region && !strcmp(p->region, "*"), whole0 = rest0 || !strcmp(p->region ? p->region : ".", ".");
                    uint64_t sv = 0;
                    if (rest0 && bi.have_nocoor) sv = bi.nocoor_voff; else if (whole0 && !rest0 && bi.have_start) sv = bi.start_voff;
                    uint64_t tgt[2], got[2];
                    for (int k = 0; k < 2; ++k) tgt[k] = std::max<uint64_t>((uint64_t)((double)bam_len * (p->shard + k) / p->n_shards) << 16, sv ? sv : 1);
                    bai_first_anchor_ge(bai, bai_len, tgt, 2, got);
                    if (p->shard > 0 && got[0] != UINT64_MAX) up_lo = std::min<size_t>(bam_len, (size_t)(got[0] >> 16));
                    else if (p->shard > 0) up_lo = bam_len;
                    if (p->shard + 1 < p->n_shards && got[1] != UINT64_MAX) up_hi = std::min<size_t>(bam_len, (size_t)(got[1] >> 16) + 2 * kBgzfMaxBlock + 64);
                    if (up_hi < up_lo) up_hi = up_lo;
                    // (the header's members, and at least the four the device-side header read starts with)
                    const std::vector<Member> &sm = *shared->members;
                    if (!sm.empty()) { const Member &m4 = sm[std::min<size_t>(sm.size(), 4) - 1]; hb = std::max<size_t>(hb, (size_t)m4.cpos + m4.clen + 8); }
                    hdr_hi = std::min(up_lo, hb + 64);
                    up_lo &= ~(size_t)4095;
                    if (up_lo < hdr_hi) { up_lo = 0; hdr_hi = 0; }
                }
            }
            // Round 4: ONE inflate launch for the whole range, its waves gated by the arrival of their upload chunk (kernels.h InflateGate):
            // the file goes up in kGateChunks equal chunks, a 4-byte copy of this call's epoch into the chunk's flag word queued right behind each.
            // The members of the early chunks start ~0.6 ms into the upload; only those of the last chunk pay the lane-serial floor behind it
            // (round 3: three launches, each ~10 ms for a third of the members, the last one started when the last third had arrived).
            // A one-shot context, or one whose gate once gave an unclean verdict: round 3's pieces.
            const unsigned gate_chunks = overlap_knobs().chunks;
            gated = !c->gate_distrust && !c->one_shot && up_hi - up_lo >= std::min(overlap_min, (size_t)8 << 20) &&
                up_hi - up_lo >= 2 * 4096 * (size_t)gate_chunks;
            if (gated) {
                // ... for payloads whose inflate is of the upload's order (measured: bench payload 27.5 -> 26.5 ms, random bases + qualities 98.2 ->
                // 93.7); run-length payloads (long reads: 1 GB of file, 65 GB inflated, five rounds of waves) lose 5-6 ms of 158 to it and keep
                // the pieces.  The class comes from the file's first members (BSIZE / ISIZE of up to 64 of them: what the host scan will find).
                uint64_t cb = 0, ub = 0; size_t o = 0;
                for (int k = 0; k < 64 && o + 28 <= bam_len; ++k) {
                    if (!(h_bam[o] == 0x1f && h_bam[o + 1] == 0x8b && h_bam[o + 12] == 'B' && h_bam[o + 13] == 'C')) break;
                    const size_t bl = (size_t)(h_bam[o + 16] | h_bam[o + 17] << 8) + 1;
                    if (bl < 26 || o + bl > bam_len) break;
                    uint32_t isz; memcpy(&isz, h_bam + o + bl - 4, 4);
                    cb += bl; ub += isz; o += bl;
                }
                if (cb && !(inflate_plan_for(cb, ub) & 1)) gated = false;
            }
            // three pieces, each its own launch on its own hardware queue (two side streams + the pipeline's own; a launch takes ~8 ms however
            // small -- one lane per member).  Equal thirds measured best: 31.6 ms per step against 32.4-33.3 ms for pieces that shrink towards
            // the end, and 30.9-31.5 ms for four to six equal pieces on side streams of other priorities (= other queue pools), 32.6 for seven
            // (tools/lab/pieces.sh): the concurrent launches share the chip, finer pieces do not end sooner.
            std::vector<unsigned> cuts = {33, 67};
            if (c->one_shot) cuts.clear();                            // (one stream: pieces would only take turns on it)
            if (gated) {
                gate_chunk = (((up_hi - up_lo) + gate_chunks - 1) / gate_chunks + 4095) & ~(size_t)4095;
                for (size_t e = up_lo + gate_chunk; e < up_hi; e += gate_chunk) up.end.push_back(e);
                DevBuf &bg = c->buf("gate_flags");
                if (!bg.p) { HIP_TRY(bg.ensure(4 * 64)); HIP_TRY(hipMemset(bg.p, 0, 4 * 64)); c->gate_epoch = 0; }
                ++c->gate_epoch;
            } else
            for (unsigned pc : cuts) {
                const size_t e = (up_lo + (size_t)((double)(up_hi - up_lo) * pc / 100.0) + 4095) & ~(size_t)4095;
                if (e < up_hi && e > up_lo && (up.end.empty() || e > up.end.back())) up.end.push_back(e);
            }
            up.end.push_back(up_hi);
            up.lo = up_lo; up.hi = up_hi; up.hdr_hi = hdr_hi;
            while (c->chunk_ev.size() < up.end.size()) { hipEvent_t e; HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming)); c->chunk_ev.push_back(e); }
            uint8_t *dst = b.as<uint8_t>();
            up.copy_stream = copy_q;
            uint32_t *gate_flags = gated ? c->buf("gate_flags").as<uint32_t>() : nullptr;
            const uint32_t gate_epoch = c->gate_epoch;
            hipStream_t gate_q = c->side[0] ? c->side[0] : copy_q;
            up.th = std::thread([this, dst, hdr_hi, up_lo, gate_q, gate_flags, gate_epoch] {
                if (hipSetDevice(c->device) != hipSuccess) { up.err = 1; up.recorded = (uint32_t)up.end.size(); return; }
                if (hdr_hi && hipMemcpyAsync(dst, h_bam, hdr_hi, hipMemcpyHostToDevice, copy_q) != hipSuccess) up.err = 1;
                size_t o = up_lo;
                for (size_t j = 0; j < up.end.size(); ++j) {
                    if ((up.end[j] > o && hipMemcpyAsync(dst + o, h_bam + o, up.end[j] - o, hipMemcpyHostToDevice, copy_q) != hipSuccess) ||
                        hipEventRecord(c->chunk_ev[j], copy_q) != hipSuccess) up.err = 1;
                    // (the flag's one-lane kernel goes to a side stream behind the chunk's event: on the copy stream itself it sat between two
                    //  copies, ~30 us of an idle bus per chunk)
                    if (gate_flags) {
                        if (gate_q != copy_q && hipStreamWaitEvent(gate_q, c->chunk_ev[j], 0) != hipSuccess) up.err = 1;
                        launch_gate_set(gate_flags + j, gate_epoch, gate_q);
                        // (a refused launch is only in THIS thread's hipGetLastError: unread, the flag would never be set and every gated wave
                        //  would wait out its time-out)
                        if (hipGetLastError() != hipSuccess) up.err = 1;
                    }
                    o = up.end[j];
                    up.recorded.store((uint32_t)j + 1, std::memory_order_release);
                }
                // the file is on the device: the next call's upload may start.  A host function behind the last copy, not a wait in this thread -- a thread
                // blocked in hipEventSynchronize kept the call's own launches from being enqueued until the upload was over (round 6: the gated launch
                // went out 8.8 ms late).
                if (c->link && hipLaunchHostFunc(copy_q, [](void *h) { ((TurnHold *)h)->give(); }, &c->wire_hold) != hipSuccess) { (void)hipGetLastError();
                    c->wire_hold.give(); }
            });
            if (shared) { hm = *shared->members; hm_total = shared->total_inflated; overlap = !hm.empty(); }
            else overlap = scan_members_parallel(h_bam, bam_len, (int)usable_threads(24), hm, hm_total);
            mark("host member scan");
            if (!overlap) {       // not a file the host vouches for: everything on the device, after the last chunk
                up.th.join();
                if (up.err) return fail(err, errlen, RGX_ERR_DEVICE, "regtools_amd: upload failed\n");
                if (up_lo || up_hi < bam_len) {                 // (only a range went up: the rest before the device looks at the file)
                    if (up_lo > hdr_hi) HIP_TRY(hipMemcpyAsync(dst + hdr_hi, h_bam + hdr_hi, up_lo - hdr_hi, hipMemcpyHostToDevice, copy_q));
                    if (up_hi < bam_len) HIP_TRY(hipMemcpyAsync(dst + up_hi, h_bam + up_hi, bam_len - up_hi, hipMemcpyHostToDevice, copy_q));
                    HIP_TRY(hipEventRecord(c->chunk_ev[up.end.size() - 1], copy_q));
                }
                HIP_TRY(hipStreamWaitEvent(st, c->chunk_ev[up.end.size() - 1], 0));
            }
        } else HIP_TRY(hipMemcpyAsync(b.p, h_bam, bam_len, hipMemcpyHostToDevice, st));
    }
    DevBuf &b_scalars = c->buf("scalars");
    HIP_TRY(b_scalars.ensure(512));
    // u32 scalars: [0]=first bad member [1]=its status [2]=changed [3]=n_rec [4]=n_events [5]=n_long [6]=n_unique [8..9]=n_iterated(u64)
    //              [12..13]=header inflate status [16]=n_cand [17]=n_members [18]=stop [20..21]=total inflated (u64)
    //              [24..26]=q_index [32..37]=q_upos (u64 x3) [40..45]=q_coff (u64 x3)
    d_sc = b_scalars.as<uint32_t>();
    h_sc = (uint32_t *)c->pinned;
    HIP_TRY(hipMemsetAsync(d_sc, 0, 512, st));
    HIP_TRY(hipMemsetAsync(d_sc, 0xff, 4, st));
    HIP_TRY(hipMemsetAsync(d_sc + 12, 0xff, 4, st));
    HIP_TRY(hipMemsetAsync(d_sc + 18, 0xff, 4, st));
    HIP_TRY(hipMemsetAsync(d_sc + kStatusEarly, 0xff, 4, st));

    return kGoOn;
}

int EventsRun::stage_members() {
    // -- BGZF member discovery on the device (replaces the serial BSIZE walk, bgzf.c:421-546) -------------------------------
    std::vector<Member> &hm = c->hm_scratch;
    DevBuf &b_members = c->buf("members"), &b_disc = c->buf("discover");
    if (overlap) {
        // the member list came from the host scan: what the discovery kernels would have left in HBM
        // (in page-locked host memory, read in place by the kernels -- 24 bytes per member, once: an upload would queue behind the file's
        // chunks on the copy engine, measured 4 ms)
        n_cand = (uint32_t)hm.size();
        const size_t need = ((size_t)n_cand + 1) * sizeof(Member);
        if (need > c->pinned_members_cap) {
            if (c->pinned_members) (void)hipHostFree(c->pinned_members);
            c->pinned_members = nullptr; c->pinned_members_cap = 0;
            HIP_TRY(hipHostMalloc(&c->pinned_members, need + need / 4, hipHostMallocDefault));
            c->pinned_members_cap = need + need / 4;
        }
        memcpy(c->pinned_members, hm.data(), (size_t)n_cand * sizeof(Member));
        h_sc[17] = n_cand; memcpy(h_sc + 20, &hm_total, 8);
        HIP_TRY(hipMemcpyAsync(d_sc + 17, h_sc + 17, 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d_sc + 20, h_sc + 20, 8, hipMemcpyHostToDevice, st));
    } else {
    const uint32_t n_tiles = (uint32_t)((bam_len + kMagicTile - 1) / kMagicTile);
    HIP_TRY(b_disc.ensure((size_t)n_tiles * 4 + scan_tmp_words(n_tiles) * 4 + 256));
    uint32_t *tile_cnt = b_disc.as<uint32_t>(), *tile_tmp = tile_cnt + n_tiles;
    launch_magic_count(d_bam, bam_len, n_tiles, tile_cnt, st);
    launch_scan_u32(tile_cnt, tile_cnt, n_tiles, d_sc + 16, tile_tmp, st);
    HIP_TRY(hipMemcpyAsync(h_sc + 16, d_sc + 16, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    n_cand = h_sc[16];
    if (n_cand == 0) return fail(err, errlen, RGX_ERR_OPEN, "%s", kMsgOpen);
    DevBuf &b_cand = c->buf("cand");
    {
        const size_t N = n_cand;
        HIP_TRY(b_cand.ensure(N * 8 + N * 4 * 6 + scan_tmp_words(n_cand) * 4 + 256));
        HIP_TRY(b_members.ensure((N + 1) * sizeof(Member)));
    }
    cand = b_cand.as<uint64_t>();
    nx[0] = (uint32_t *)(cand + n_cand); nx[1] = (uint32_t *)(cand + n_cand) + n_cand;
    c_isize = nx[1] + n_cand; c_reach = c_isize + n_cand; c_rank = c_reach + n_cand; c_isz2 = c_rank + n_cand; c_tmp = c_isz2 + n_cand;
    launch_magic_fill(d_bam, bam_len, n_tiles, tile_cnt, cand, st);
    }
    d_members = overlap ? (Member *)c->pinned_members : b_members.as<Member>();
    from_members = overlap ? hipMemcpyHostToHost : hipMemcpyDeviceToHost;
    if (!overlap) chain(UINT64_MAX);
    if (bai_thread.joinable()) bai_thread.join();
    if (!bai_ok) return (void)hipStreamSynchronize(st), fail(err, errlen, RGX_ERR_INDEX, "%s", kMsgIndex);
    mark("parse_bai");
    // "." = every record from the first one on; "*" = every record behind the last reference's reads (hts_itr_querys, hts.c:1901-1904:
    // HTS_IDX_START / HTS_IDX_NOCOOR; both read to the end of the file without a predicate)
    const bool rest = p->region && !strcmp(p->region, "*");
    whole = rest || !strcmp(p->region ? p->region : ".", ".");
    // where the record stream starts (hts.c:1721-1741)
    seek = false; seek_voff = 0;
    if (rest) {
        if (bi.have_nocoor) { seek_voff = bi.nocoor_voff; seek = seek_voff != 0; }
        else if (!bi.n_no_coor) return (void)hipStreamSynchronize(st), fail(err, errlen, RGX_ERR_REGION, "%s", kMsgRegion);
    } else if (whole) {
        if (bi.have_start) { seek_voff = bi.start_voff; seek = seek_voff != 0; }
        else if (!bi.n_no_coor) return (void)hipStreamSynchronize(st), fail(err, errlen, RGX_ERR_REGION, "%s", kMsgRegion);
    }
    // shard cut points: virtual offsets the BAI lists (every chunk begin / linear-index entry is a record start), so
    // no shard ever guesses its first record.  A record belongs to the shard in which its first byte lies.
    cut_lo = seek ? seek_voff : 0; cut_hi = UINT64_MAX;          // 0 = "right after the header"
    if (p->n_shards > 1) {
        if (p->shard < 0 || p->shard >= p->n_shards) return (void)hipStreamSynchronize(st), fail(err, errlen, RGX_ERR_ARG, "regtools_amd: shard %d of %d\n",
            p->shard, p->n_shards);
        uint64_t tgt[2], got[2];
        for (int k = 0; k < 2; ++k) {
            const int g = p->shard + k;
            tgt[k] = std::max<uint64_t>((uint64_t)((double)bam_len * g / p->n_shards) << 16, seek ? seek_voff : 1);
        }
        bai_first_anchor_ge(bai, bai_len, tgt, 2, got);
        if (p->shard > 0) cut_lo = got[0];
        if (p->shard + 1 < p->n_shards) cut_hi = got[1];
        if (cut_hi < cut_lo) cut_hi = cut_lo;
        mark("shard cuts");
    }

    // region queries: the reference's iterator reads the CHUNKS the index lists for the region's bins, one seek each, and ends at the first
    // record it reads that lies on another contig or at / behind the region's end (hts_itr_query / hts_itr_next, hts.c:1733-1800, :1924-1965).
    // The chunk list is computed here, from the index; the members between the first chunk's begin and the last one's end are inflated,
    // every chunk becomes its own record chain (SegGeom) and k_decode_seg applies the end rule.  Needs the contig names before the
    // launch: the header is inflated on the host from the head of the file.  A header that cannot be read that way leaves the range
    // alone: the whole file is read and filtered by overlap (such a header is not readable upstream either).
    chunked = false;
    geom_chunked_hint = !whole;                    // (region queries keep the checked path: their chunk table wants the members' verdicts)
    if (!whole && p->region) {
        const size_t head_len = std::min<size_t>(bam_len, (size_t)8 << 20);
        std::vector<uint8_t> head_copy;
        const uint8_t *head = h_bam;
        if (!head) { head_copy.resize(head_len); HIP_TRY(hipMemcpy(head_copy.data(), d_bam, head_len, hipMemcpyDeviceToHost)); head = head_copy.data(); }
        BamHeader hh;
        int32_t tid = -1, beg = 0, end = 0;
        // (the one parse of a call that talks: what sam_itr_querys -> hts_parse_decimal says about the region's numbers, once per query)
        if (host_bam_header(head, head_len, hh) && parse_region(hh, p->region, tid, beg, end,
            /*say=*/p->n_shards <= 1 || p->shard == 0) && tid < bi.n_ref && end >= beg &&
            region_chunks(bai, bai_len, tid, beg, end, chunks)) {
            chunked = true;
            if (p->n_shards > 1) {
                // a region query over several shards: the iterator's chunk list is dealt out in order, in runs of about equal compressed size;
                // shard g follows its run as an iterator of its own, and the one that meets the record that ends the iteration says so
                // (stream_ended): the merge ignores the shards behind it, as it does behind damage
                std::vector<uint64_t> before(chunks.size() + 1, 0);
                for (size_t k = 0; k < chunks.size(); ++k) before[k + 1] = before[k] + std::max<uint64_t>(1, (chunks[k].v >> 16) - (chunks[k].u >> 16));
                const uint64_t W = std::max<uint64_t>(1, before[chunks.size()]);
                std::vector<VChunk> mine;
                for (size_t k = 0; k < chunks.size(); ++k)
                    if ((int)std::min<uint64_t>((uint64_t)p->n_shards - 1, before[k] * (uint64_t)p->n_shards / W) == p->shard) mine.push_back(chunks[k]);
                chunks.swap(mine);
                cut_lo = seek ? seek_voff : 0; cut_hi = UINT64_MAX;     // (the byte cuts above were for a whole-file read)
            }
            if (!chunks.empty()) {
                // like the iterator's bgzf_seek: reading starts at the first chunk whatever the state of the members in front of it
                uint64_t hi = 0;
                for (const VChunk &ch : chunks) hi = std::max(hi, ch.v);
                cut_lo = chunks.front().u; cut_hi = std::max(hi, cut_lo); seek = true; seek_voff = cut_lo;
            // no bin of the region holds a record: nothing to inflate
            } else if (bi.have_start && bi.start_voff) { cut_lo = cut_hi = bi.start_voff; seek = true; seek_voff = cut_lo; }
        }
        mark("region chunks");
    }

    empty_stream = false;
    auto query = [&]() -> hipError_t {
        uint64_t q[3] = {seek ? (seek_voff >> 16) : 0, cut_lo >> 16, cut_hi == UINT64_MAX ? UINT64_MAX - 64 : (cut_hi >> 16)};
        if (overlap) {
            // the member list is on the host (scan_members_parallel): what k_member_query / k_member_stop would answer, without a round trip
            const uint32_t nm = (uint32_t)hm.size();
            uint64_t q_up[3];
            for (int k = 0; k < 3; ++k) {
                uint32_t lo_ = 0, hi_ = nm;
                while (lo_ < hi_) { const uint32_t mid = lo_ + (hi_ - lo_) / 2; if (hm[mid].cpos < q[k] + 18) lo_ = mid + 1; else hi_ = mid; }
                const bool hit = lo_ < nm && hm[lo_].cpos == q[k] + 18;
                h_sc[24 + k] = hit ? lo_ : nm; q_up[k] = hit ? hm[lo_].upos : ~0ull;
            }
            memcpy(h_sc + 32, q_up, sizeof q_up);
            uint32_t stop_ = 0xffffffffu;
            for (uint32_t i = h_sc[24]; i < nm; ++i) if (hm[i].isize == 0 || hm[i].isize > kBgzfMaxBlock) { stop_ = i; break; }
            h_sc[18] = stop_; h_sc[17] = nm; memcpy(h_sc + 20, &hm_total, 8);
            return hipSuccess;
        }
        memcpy(h_sc + 40, q, sizeof q);
        hipError_t e = hipMemcpyAsync(d_sc + 40, h_sc + 40, sizeof q, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return e;
        e = hipMemsetAsync(d_sc + 18, 0xff, 4, st);
        if (e != hipSuccess) return e;
        launch_member_query(d_members, d_sc + 17, (const uint64_t *)(d_sc + 40), 3, d_sc + 24, (uint64_t *)(d_sc + 32), st);
        // the stream ends at the first empty (or oversized = corrupt) member at/after the first one read (bgzf.c:548-578)
        launch_member_stop(d_members, n_cand, d_sc + 17, d_sc + 24, d_sc + 18, st);
        e = hipMemcpyAsync(h_sc, d_sc, 256, hipMemcpyDeviceToHost, st);
        if (e != hipSuccess) return e;
        return hipStreamSynchronize(st);
    };
    HIP_TRY(query());
    if (overlap && seek && (seek_voff >> 16) != 0 && h_sc[24] >= h_sc[17]) {
        // the index points at something that is no member of this (well-formed) file: the device's discovery decides what that means
        up.th.join();
        HIP_TRY(complete_upload());
        HIP_TRY(hipStreamSynchronize(copy_q));
        HIP_TRY(hipStreamSynchronize(st));
        const int rc2 = prepare_events(c, d_bam, nullptr, bam_len, bai, bai_len, p, want_read_span, P, err, errlen, nullptr, false, region_to_file_end);
        P.t_begin = t_begin;
        return rc2;
    }
    if (seek && (seek_voff >> 16) != 0 && h_sc[24] >= h_sc[17]) {
        // the seek target is no member of the chain from offset 0: something in front of it is broken.  bgzf_seek (hts_itr_next, hts.c:1935)
        // goes there regardless -- take it as a second chain root.  (Only damaged files get here.)
        chain(seek_voff >> 16);
        HIP_TRY(query());
        mark("second chain root");
        if (h_sc[24] >= h_sc[17]) {
            // there is no BGZF member at the seek target at all (a truncated file, an index that belongs to another file): the
            // reference's read after bgzf_seek fails and the iterator returns nothing.  Keep the head of the file for the header only.
            chain(UINT64_MAX);
            seek = false; cut_lo = 0; cut_hi = 1; empty_stream = true;
            HIP_TRY(query());
        }
    }
    n_members_all = h_sc[17];
    if (n_members_all == 0) return fail(err, errlen, RGX_ERR_OPEN, "%s", kMsgOpen);     // offset 0 is not a BGZF member
    memcpy(&total_all, h_sc + 20, 8);
    first_member = seek ? h_sc[24] : 0;                                    // == n_members_all when the seek target is no member
    stop = std::min(h_sc[18], n_members_all);
    memcpy(q_upos, h_sc + 32, sizeof q_upos);
    mark("member discovery (2 syncs)");

    return kGoOn;
}

int EventsRun::stage_range_and_inflate() {
    // -- member range of this call ---------------------------------------------------------------------------------------------
    std::vector<Member> &hm = c->hm_scratch;
    DevBuf &b_arena = c->buf("arena");
    m_lo = cut_lo ? h_sc[25] : 0;
    if (m_lo <= 4) m_lo = 0;        // keep the file head (BAM header) in the same launch: a lone lane needs milliseconds per member
    m_hi = stop;                                                  // exclusive
    if (cut_hi != UINT64_MAX) {
        const uint32_t mh = h_sc[26];
        const uint32_t hi_m = (mh < n_members_all && (cut_hi & 0xffff)) ? mh + 1 : mh;
        // a region's chunks: each is a seek of its own, so an empty member between two of them ends nothing (the chunks' own limits do
        // that, below); two members more than the index asks for, for the records of a stale index that run past their chunk's end
        if (chunked) m_hi = region_to_file_end ? n_members_all : (uint32_t)std::min<uint64_t>(n_members_all, (uint64_t)hi_m + 2);
        else m_hi = std::min(stop, hi_m);
    }
    if (m_lo > m_hi) m_lo = m_hi;
    if (overlap && (up.lo || up.hi < bam_len) && m_hi > m_lo) {
        // only a byte range of the file went up (a shard of a shared scan): it must hold every member of this call's range
        const uint64_t need_lo = hm[m_lo].cpos - 18, need_hi = hm[m_hi - 1].cpos + hm[m_hi - 1].clen + 16;
        if (need_lo < up.lo || need_hi > up.hi) {
            mark("shard range does not cover its members: whole-file upload");
            up.th.join();
            HIP_TRY(hipStreamSynchronize(copy_q));
            HIP_TRY(hipStreamSynchronize(st));
            const int rc2 = prepare_events(c, d_bam_in, h_bam, bam_len, bai, bai_len, p, want_read_span, P, err, errlen, d_true_sizes, allow_overlap,
                region_to_file_end, nullptr);
            P.t_begin = t_begin;
            return rc2;
        }
    }
    // arena offsets of the range ends
    uint64_t upos_hi = 0;
    upos_lo = 0;
    HIP_TRY(upos_of(m_lo, upos_lo));
    HIP_TRY(upos_of(m_hi, upos_hi));
    total = upos_hi - upos_lo;
    n_range = m_hi - m_lo;
    HIP_TRY(b_arena.ensure(total + 256));
    HIP_TRY(hipEventRecord(c->ev[0], st));
    DevBuf &b_lens = c->buf("inflate_scratch");
    HIP_TRY(b_lens.ensure(inflate_scratch_bytes(std::max<uint32_t>(n_range, 64))));
    // with a seek, the members in front of its target are only inflated for the header's sake (same launch): their failures end nothing
    const uint32_t ignore_below = (seek && first_member < n_members_all && first_member > m_lo) ? first_member - m_lo : 0;
    d_bad = nullptr;                                 // region queries: which members of the range did not inflate (every chunk has its own end of stream)
    if (chunked && !chunks.empty() && n_range) {
        DevBuf &b_bad = c->buf("bad_members");
        HIP_TRY(b_bad.ensure((size_t)n_range + 64));
        d_bad = b_bad.as<uint8_t>();
        HIP_TRY(hipMemsetAsync(d_bad, 0, n_range, st));
    }
    const int pairs = inflate_plan_for(bam_len, total_all);      // (the whole file's ratio: a range of it is the same kind of payload)
    // (early tail: the second of two gated launches is still running on a side stream; whoever reads its part of the arena, or the launch's
    //  verdict, first makes the pipeline's stream wait for it)
    c->launch_timed = false;
    auto timed_launch = [&](hipStream_t q, bool piece, InflateGate gate) {      // the call's whole-range launch, with its own pair of events on its own stream
        if (c->link) { c->chip_hold.take(&c->link->chip); mark("the chip's DEFLATE turn is ours"); }
        (void)hipEventRecord(c->ev_launch[0], q);
        launch_inflate(d_bam, d_members + m_lo, n_range, b_arena.as<uint8_t>(), upos_lo, b_lens.as<uint32_t>(), d_sc, q, ignore_below, 0, piece, 0, d_bad,
            pairs, false, gate);
        (void)hipEventRecord(c->ev_launch[1], q);
        if (c->link && hipLaunchHostFunc(q, [](void *h) { ((TurnHold *)h)->give(); }, &c->chip_hold) != hipSuccess) { (void)hipGetLastError();
            c->chip_hold.give(); }
        c->launch_timed = true;
    };
    if (!overlap) timed_launch(st, false, InflateGate());
    else {
        // one launch per upload chunk, on the side streams: the members whose bytes (plus the decoder's 16-byte look-ahead) have arrived with
        // chunk j start as soon as its event fires, next to the launches of the chunks before it
        HIP_TRY(b_lens.ensure(inflate_scratch_bytes(std::max<uint32_t>(n_range, 64)) + up.end.size() * inflate_scratch_bytes(64)));
        HIP_TRY(hipEventRecord(c->ev_ready, st));
        for (auto &q : c->side) if (q) HIP_TRY(hipStreamWaitEvent(q, c->ev_ready, 0));
        uint32_t g_lo = m_lo; size_t scratch_off = 0; unsigned used_side = 0;
        if (gated) {
            // one launch on the pipeline's stream, now: its waves wait for their chunk's flag themselves (k_inflate_coop; a range the wave form
            // takes -- a few thousand members -- is one launch behind the last chunk)
            if (inflate_takes_coop(n_range)) {
                InflateGate gate;
                gate.flags = c->buf("gate_flags").as<uint32_t>(); gate.epoch = c->gate_epoch; gate.n_chunks = (uint32_t)up.end.size(); gate.lo = up.lo;
                    gate.chunk_bytes = gate_chunk;
                // Round 4, second half ("early tail"): the launch goes to a side stream and counts its finished waves per PART of the member list
                // (parts cut where upload chunks end, at multiples of the lane-sorting group); the pipeline's stream waits for part after part
                // (launch_wait_done) and frames, verifies and decodes the part of the arena behind it while the waves of the later parts still
                // run -- what is left behind the launch's end is the last part's framing and decode, not the whole file's.  (A member's own
                // chain puts the end of the launch 6-9 ms behind the last chunk's arrival, whatever the chip does meanwhile.)
                // REGTOOLS_AMD_EARLY_TAIL="6,9,12,14" = the cuts in sixteenths of the upload (up to seven; the default since round 5: 23.5 ms per step where
                // "8,12,14" gives 24.5 -- the tail
                // under the launch is the critical path from the first part on, so it starts earlier and in smaller parts;
                // profiles/r05_step_early_tail_four_cuts_ab.txt), "0" = off.
                static const std::vector<unsigned> env_cuts = [] {
                    std::vector<unsigned> v; const char *e = getenv("REGTOOLS_AMD_EARLY_TAIL");
                    unsigned x7[7] = {0, 0, 0, 0, 0, 0, 0};
                    const int n = sscanf(e ? e : "6,9,12,14", "%u,%u,%u,%u,%u,%u,%u", &x7[0], &x7[1], &x7[2], &x7[3], &x7[4], &x7[5], &x7[6]);
                    for (unsigned x : x7) if ((int)v.size() < n && x > 0 && x < 16 && (v.empty() || x > v.back())) v.push_back(x);
                    return v;
                }();
                const uint32_t early_min = overlap_knobs().early_min;
                if (!env_cuts.empty() && !c->early_distrust && c->side[1] && up.end.size() >= 8 && !d_bad && !d_true_sizes) {
                    const uint32_t align = kInflateSortGroup;          // (a wave's members all come from one group of that many)
                    for (unsigned cut : env_cuts) {
                        const size_t kA = std::max<size_t>(1, up.end.size() * (size_t)cut / 16);
                        const uint64_t lim_b = up.end[kA - 1];
                        uint32_t lo = m_lo, hi = m_hi;             // first member that reads bytes behind chunk kA - 1 (k_inflate_coop's own rule)
                        while (lo < hi) { const uint32_t mid = lo + (hi - lo) / 2; if (hm[mid].cpos + hm[mid].clen + 24 <= lim_b) lo = mid + 1; else hi = mid; }
                        uint32_t k = (lo - m_lo) / align * align;     // members of the range in front of the cut
                        const uint32_t prev = early_parts.empty() ? 0u : early_parts.back().members;
                        if (k >= prev + early_min && n_range - k >= early_min) early_parts.push_back(EarlyPart{k, k / 64, hm[m_lo + k].upos - upos_lo});
                    }
                }
                if (!early_parts.empty()) {
                    DevBuf &bd = c->buf("gate_done");
                    HIP_TRY(bd.ensure(64));
                    uint32_t *d_done = bd.as<uint32_t>();
                    hipStream_t q = c->side[1];
                    HIP_TRY(hipMemsetAsync(d_done, 0, 32, q));
                    gate.done = d_done;
                    for (size_t j = 0; j < early_parts.size(); ++j) gate.part_start[j] = early_parts[j].waves;
                    timed_launch(q, /*piece=*/true, gate);
                    HIP_TRY(hipEventRecord(c->ev_side[1], q));
                    split_ev = c->ev_side[1];
                    split_B = true;
                } else timed_launch(st, /*piece=*/true, gate);
            } else {
                while (up.recorded.load(std::memory_order_acquire) < up.end.size()) std::this_thread::yield();
                HIP_TRY(hipStreamWaitEvent(st, c->chunk_ev[up.end.size() - 1], 0));
                timed_launch(st, false, InflateGate());
            }
            g_lo = m_hi;
        }
        for (size_t j = 0; j < up.end.size() && g_lo < m_hi; ++j) {
            uint32_t g_hi = m_hi;
            if (j + 1 < up.end.size()) {     // first member of [g_lo, m_hi) that needs bytes beyond this chunk
                const uint64_t lim_b = up.end[j];
                uint32_t lo = g_lo, hi = m_hi;
                while (lo < hi) { const uint32_t mid = lo + (hi - lo) / 2; if (hm[mid].cpos + hm[mid].clen + 16 <= lim_b) lo = mid + 1; else hi = mid; }
                g_hi = lo;
            }
            if (g_hi == g_lo) continue;
            // (an event must have been recorded before a stream can wait on it)
            while (up.recorded.load(std::memory_order_acquire) <= j) std::this_thread::yield();
            if (up.err) return fail(err, errlen, RGX_ERR_DEVICE, "regtools_amd: upload failed\n");
            // the pipeline's own stream is idle until the inflate is done: it takes every third piece (the runtime maps streams onto four
            // hardware queues round-robin; a third side stream would share its queue with the second: profiles/r02_overlap_timeline.txt)
            const bool own = j + 1 == up.end.size() || j >= (size_t)kSideStreams || !c->side[j];
            hipStream_t q = own ? st : c->side[j];
            if (!own) used_side |= 1u << j;
            HIP_TRY(hipStreamWaitEvent(q, c->chunk_ev[j], 0));
            launch_inflate(d_bam, d_members + g_lo, g_hi - g_lo, b_arena.as<uint8_t>(), upos_lo, (uint32_t *)(b_lens.as<uint8_t>() + scratch_off), d_sc, q,
                ignore_below, g_lo - m_lo, /*piece=*/true, 0, d_bad, pairs);
            scratch_off += inflate_scratch_bytes(g_hi - g_lo);
            g_lo = g_hi;
        }
        up.th.join();
        // (gated waves give up after ~2 s)
        if (up.err) { (void)hipStreamSynchronize(st); return fail(err, errlen, RGX_ERR_DEVICE, "regtools_amd: upload failed\n"); }
        for (unsigned k = 0; k < (unsigned)kSideStreams; ++k) if (used_side >> k & 1) { HIP_TRY(hipEventRecord(c->ev_side[k], c->side[k]));
            HIP_TRY(hipStreamWaitEvent(st, c->ev_side[k], 0)); }
        HIP_TRY(hipStreamWaitEvent(st, c->chunk_ev[up.end.size() - 1], 0));      // (later stages read the file too: barcodes, header)
    }
    HIP_TRY(hipEventRecord(c->ev[1], st));
    mark(gated && overlap && inflate_takes_coop(n_range) ? "launch inflate (gated)" : "launch inflate");

    return kGoOn;
}

