// api_records.cpp -- the back of one call (EventsRun, api_internal.h): footers and BAM header, record framing, decode into the SoA, junction events.
#include "api_internal.h"

int EventsRun::stage_footers_and_header() {
    // -- files whose ISIZE footers lie ------------------------------------------------------------------------------------------
    // The arena was laid out from the footers; the reference never reads them (inflate_block, bgzf.c:292-316: a block is as long as
    // zlib says, at most 64 KiB).  When a member inflates to another length than its footer claims, or the member that ends the
    // stream is not the plain empty block it claims to be, every member is inflated once into its own 64 KiB slot to learn the true
    // lengths and the pipeline starts over with those.  Costs two extra inflate passes; only malformed files ever pay them.
    // Host input whose members the host scan vouched for: no round trip here.  The header comes from the host's own inflate of the file's head,
    // the stages behind the inflate are enqueued on the assumption that every member inflates to its footer's length (what a well-formed file
    // does), and the launch's verdict is read with the framing's counts: anything else starts over on the device-resident path below.
    DevBuf &b_arena = c->buf("arena"), &b_hdr = c->buf("hdr_arena"), &b_lens = c->buf("inflate_scratch");
    BamHeader hdr_host;
    mean_rec = 0;                                    // mean size of the file's first records (0 = unknown: 16 KiB segments)
    spec = overlap && !d_true_sizes && !geom_chunked_hint && host_bam_header(h_bam, std::min<size_t>(bam_len, (size_t)8 << 20), hdr_host, nullptr, &mean_rec);
    if (spec) { h_sc[0] = h_sc[1] = 0xffffffffu; h_sc[kStatusEarly] = h_sc[kStatusEarly + 1] = 0xffffffffu; }
    else {
        HIP_TRY(join_B());
        HIP_TRY(hipMemcpyAsync(h_sc, d_sc, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(h_sc + kStatusEarly, d_sc + kStatusEarly, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    if (!d_true_sizes) {
        auto size_trouble = [&](uint32_t k) { return h_sc[k] != 0xffffffffu &&
            (h_sc[k + 1] == 12u /* INF_SIZE_MISMATCH */ || h_sc[k + 1] == 10u /* INF_OUT_OVERFLOW */); };
        bool lies = size_trouble(0) || size_trouble(kStatusEarly);
        if (!lies && stop < n_members_all) {
            Member ms; uint8_t two[2] = {0, 0};
            HIP_TRY(member_at(stop, ms));
            if (ms.isize == 0 && ms.clen >= 2) {
                if (h_bam && ms.cpos + 2 <= bam_len) memcpy(two, h_bam + ms.cpos, 2);
                else { HIP_TRY(hipMemcpyAsync(two, d_bam + ms.cpos, 2, hipMemcpyDeviceToHost, st)); HIP_TRY(hipStreamSynchronize(st)); }
            }
            // fine: an empty block (03 00, the EOF marker) or a member cut off by the end of the file
            lies = !(ms.isize == 0 && two[0] == 3 && two[1] == 0) && ms.isize != 0xffffffffu;
        }
        if (lies) {
            mark("footer mismatch: probing");
            DevBuf &b_slots = c->buf("probe_slots"), &b_sizes = c->buf("probe_sizes");
            HIP_TRY(b_slots.ensure((size_t)n_members_all * kBgzfMaxBlock + 256));
            HIP_TRY(b_sizes.ensure((size_t)n_members_all * 4 + 64));
            HIP_TRY(b_lens.ensure(inflate_scratch_bytes(std::max<uint32_t>(n_members_all, 64))));
            launch_inflate_probe(d_bam, d_members, n_members_all, b_slots.as<uint8_t>(), b_lens.as<uint32_t>(), b_sizes.as<uint32_t>(), st);
            HIP_TRY(hipStreamSynchronize(st));
            b_slots.release();                                  // 64 KiB per member: not kept
            const int rc2 = prepare_events(c, d_bam_in, h_bam, bam_len, bai, bai_len, p, want_read_span, P, err, errlen, b_sizes.as<uint32_t>(), true,
                region_to_file_end);
            P.t_begin = t_begin;
            return rc2;
        }
    }

    // -- header (sam.c:114-223): it sits at the start of the arena when the range starts at member 0; otherwise the head of
    //    the file is inflated into its own small arena -----------------------------------------------------------------------
    if (spec) hdr = hdr_host;
    else {
        const uint32_t h_early = h_sc[kStatusEarly];            // read back right after the launch finished (below the footer check)
        uint32_t n_h = std::min<uint32_t>(n_members_all, 4);
        for (;;) {
            const uint8_t *src; uint64_t have;
            uint32_t bad_h = 0xffffffffu;
            std::vector<Member> hmem(n_h);
            HIP_TRY(hipMemcpyAsync(hmem.data(), d_members, (size_t)n_h * sizeof(Member), from_members, st));
            HIP_TRY(hipStreamSynchronize(st));
            uint32_t used = 0;
            for (; used < n_h; ++used) if (hmem[used].isize == 0 || hmem[used].isize > kBgzfMaxBlock) break;   // the header read stops there
            if (!used) return fail(err, errlen, RGX_ERR_REGION, "%s", kMsgRegion);
            have = hmem[used - 1].upos + hmem[used - 1].isize;
            if (m_lo == 0 && used <= n_range) src = b_arena.as<uint8_t>();
            else {
                HIP_TRY(b_hdr.ensure(have + 256));
                HIP_TRY(hipMemsetAsync(d_sc + 12, 0xff, 4, st));
                launch_inflate(d_bam, d_members, used, b_hdr.as<uint8_t>(), 0, b_lens.as<uint32_t>(), d_sc + 12, st);
                src = b_hdr.as<uint8_t>();
            }
            std::vector<uint8_t> hbuf(have);
            HIP_TRY(hipMemcpyAsync(hbuf.data(), src, have, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(h_sc, d_sc, 64, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            bad_h = (src == b_arena.as<uint8_t>()) ? std::min(h_sc[0], h_early) : h_sc[12];     // (members in front of a seek target report apart)
            if (bad_h != 0xffffffffu && bad_h < used) have = hmem[bad_h].upos;          // a corrupt member ends the header read
            uint64_t need = 0;
            int r = parse_bam_header(hbuf.data(), have, hdr, need);
            if (r == 0) {
                // how long the file's records are, from the first ones behind the header in the bytes at hand (the host's estimate on the spec path):
                // files of long records are framed in long segments
                uint64_t o = hdr.end, sum = 0; uint32_t cnt = 0;
                while (o + 4 <= have) {
                    uint32_t bl; memcpy(&bl, hbuf.data() + o, 4);
                    if (bl < 32 || bl > (1u << 27) || o + 4 + bl > have) break;
                    sum += 4 + (uint64_t)bl; ++cnt; o += 4 + (uint64_t)bl;
                }
                if (cnt >= 4) mean_rec = (uint32_t)(sum / cnt);
                break;
            }
            if (r == 2 || used < n_h || n_h == n_members_all || (bad_h != 0xffffffffu && bad_h < used))
                return fail(err, errlen, RGX_ERR_REGION, "%s", kMsgRegion);   // sam_hdr_read == NULL (cc:519-522)
            n_h = std::min(n_members_all, n_h * 4);
        }
    }
    n_ref = (int32_t)hdr.names.size();
    mark("header (sync: inflate done)");

    return kGoOn;
}

int EventsRun::stage_bounds_and_chains() {
    // -- stream bounds inside the arena -------------------------------------------------------------------------------------
    lim = total;
    if (h_sc[0] != 0xffffffffu) {       // a member of the range failed to inflate: the stream ends where it starts
        uint64_t u = 0;
        HIP_TRY(upos_of(m_lo + h_sc[0], u));
        lim = u - upos_lo;
    }
    auto arena_of = [&](uint64_t voff, uint32_t idx, uint64_t upos) -> uint64_t {   // virtual offset -> arena offset (idx/upos from the query)
        if (idx >= n_members_all || idx < m_lo || idx >= m_hi) return total;
        return std::min<uint64_t>(total, upos - upos_lo + (voff & 0xffff));
    };
    if (cut_lo) pos0 = arena_of(cut_lo, h_sc[25], q_upos[1]);
    else pos0 = hdr.end;                 // no seek: records start right after the header (range starts at member 0)
    // Did the stream stop for a reason that ends iteration upstream, rather than at this shard's upper cut?  (A later shard is a seek past
    // that point; the merge drops the shards behind one that ended, so that a damaged file gives the same table whatever the shard count.)
    chain_ended = false;
    P.stream_ended = empty_stream;
    if (cut_hi != UINT64_MAX) {
        const uint64_t cut_lim = arena_of(cut_hi, h_sc[26], q_upos[2]);
        const uint32_t mh = h_sc[26];
        const uint32_t hi_wanted = (mh < n_members_all && (cut_hi & 0xffff)) ? mh + 1 : mh;
        // (a region's chunks are seeks of their own: what lies between two of them ends nothing, and a chunk whose reader does run into such a
        //  member reports it through its chain, below)
        if (!chunked && h_sc[0] != 0xffffffffu && lim < cut_lim) P.stream_ended = true;        // a member in front of the cut does not inflate
        // an empty / unusable member in front of the cut (bgzf.c:548-578)
        if (!chunked && stop < std::min(hi_wanted, n_members_all)) P.stream_ended = true;
        lim = std::min(lim, cut_lim);
    } else if (h_sc[0] != 0xffffffffu) P.stream_ended = true;
    if (pos0 > lim) pos0 = lim;
    if (empty_stream) lim = pos0;            // the seek target does not exist: no record is read

    memset(&cfg, 0, sizeof cfg);
    cfg.n_ref = n_ref; cfg.strandness = p->strandness; cfg.tag0 = (uint8_t)p->strand_tag[0]; cfg.tag1 = (uint8_t)p->strand_tag[1];
    // (`junctions extract` only: identify's per-window extractions upstream meet such a read only inside a window -- DESIGN 8)
    if ((p->strandness == 0 || p->barcodes) && !want_read_span) { cfg.abort_out = d_sc + 96; HIP_TRY(hipMemsetAsync(d_sc + 96, 0xff, 4, st)); }
    if (p->barcodes && !want_read_span) { cfg.bc0 = (uint8_t)p->barcode_tag[0]; cfg.bc1 = (uint8_t)p->barcode_tag[1]; }
    // (identify: the same reads counted and marked; which of them a window reads is known when the windows are, cse_api.cpp)
    if (p->strandness == 0 && want_read_span) { cfg.odd_count = d_sc + 97; HIP_TRY(hipMemsetAsync(d_sc + 97, 0, 4, st)); }
    P.odd_aux.clear();
    cfg.min_anchor = p->min_anchor; cfg.min_intron = p->min_intron; cfg.max_intron = p->max_intron;
    cfg.region_tid = -2; cfg.long_threshold = 16;
    if (!whole) {
        int32_t tid, beg, end;
        if (!parse_region(hdr, p->region, tid, beg, end) || tid >= bi.n_ref || end < beg)
            return fail(err, errlen, RGX_ERR_REGION, "%s", kMsgRegion);
        cfg.region_tid = tid; cfg.region_beg = beg; cfg.region_end = end;
    }

    // -- intron-motif strand rule: FASTA bytes + one descriptor per BAM contig in HBM (junctions_extractor.cc:345-359) ------
    if (p->fasta_path) {
        if (!c->fasta || c->fasta_path != p->fasta_path) {
            delete c->fasta; c->fasta = new Fasta(); c->fasta_path.clear();
            if (!c->fasta->load(p->fasta_path)) { delete c->fasta; c->fasta = nullptr; return fail(err, errlen, RGX_ERR_FASTA,
                "Unable to open FASTA file.\n\n"); }
            DevBuf &bf = c->buf("fasta");
            HIP_TRY(bf.ensure(c->fasta->size + 256));
            HIP_TRY(hipMemcpy(bf.p, c->fasta->data, c->fasta->size, hipMemcpyHostToDevice));
            c->fasta_path = p->fasta_path;
        }
        std::vector<FaContig> tab((size_t)std::max(n_ref, 1));
        memset(tab.data(), 0, tab.size() * sizeof(FaContig));
        for (int32_t t = 0; t < n_ref; ++t)
            for (const Fasta::Seq &s : c->fasta->seqs)
                if (s.name == hdr.names[(size_t)t]) {
                    // the kernels index fa[offset + p / line_blen * line_len + p % line_blen] without a bounds check: a descriptor that does
                    // not fit the file (stale or damaged .fai) makes the contig absent -- a junction there then fails the call like a contig
                    // the FASTA does not have (junctions_extractor.cc:553), instead of reading HBM out of bounds
                    const bool sane = s.offset >= 0 && s.len >= 0 && s.line_blen > 0 && s.line_len >= s.line_blen &&
                                      (s.len == 0 ||
                                          (uint64_t)s.offset + (uint64_t)((s.len - 1) / s.line_blen) * (uint64_t)s.line_len +
                                                  (uint64_t)((s.len - 1) % s.line_blen) < (uint64_t)c->fasta->size);
                    if (!sane) continue;
                    tab[(size_t)t].offset = s.offset; tab[(size_t)t].len = s.len; tab[(size_t)t].line_blen = s.line_blen;
                        tab[(size_t)t].line_len = s.line_len; tab[(size_t)t].present = 1;
                }
        DevBuf &bt = c->buf("fasta_tab");
        HIP_TRY(bt.ensure(tab.size() * sizeof(FaContig) + 64));
        HIP_TRY(hipMemcpy(bt.p, tab.data(), tab.size() * sizeof(FaContig), hipMemcpyHostToDevice));
        cfg.fa_data = c->buf("fasta").as<uint8_t>(); cfg.fa_tab = bt.as<FaContig>(); cfg.fa_missing = d_sc + 64;
    }

    // -- region queries: one record chain per chunk of the iterator -----------------------------------------------------------------
    // chunk c = virtual offsets [u, v): a seek to u (bgzf_seek: the member at u >> 16, the offset inside it clipped to its length; no such
    // member = the read fails and the iteration is over), then records while the position in front of the next one is below v.
    memset(&geom, 0, sizeof geom);
    const int env_seg = decode_knobs().seg_bytes;                   // (tests) 16384 or 131072
    seg_bytes = env_seg == (int)kSegBytes || env_seg == (int)kSegBytesLong ? (uint32_t)env_seg : (mean_rec >= kLongRecordBytes ? kSegBytesLong : kSegBytes);
    geom.seg_bytes = seg_bytes;
    if (chunked && chunks.empty()) lim = pos0;                // an iterator without chunks returns nothing
    if (chunked && !chunks.empty() && !empty_stream) {
        std::vector<Member> rm(n_range);
        std::vector<uint8_t> bad(n_range, 0);
        if (n_range) {
            HIP_TRY(hipMemcpyAsync(rm.data(), d_members + m_lo, (size_t)n_range * sizeof(Member), from_members, st));
            HIP_TRY(hipMemcpyAsync(bad.data(), d_bad, n_range, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
        }
        // bgzf.c:548-578: an empty block reads as the end of the file
        for (uint32_t k = 0; k < n_range; ++k) if (rm[k].isize == 0 || rm[k].isize > kBgzfMaxBlock) bad[k] = 1;
        std::vector<uint32_t> next_bad((size_t)n_range + 1, n_range);
        for (uint32_t k = n_range; k-- > 0;) next_bad[k] = bad[k] ? k : next_bad[k + 1];
        auto usize = [&](uint32_t k) { return rm[k].isize <= kBgzfMaxBlock ? rm[k].isize : 0u; };
        auto lower = [&](uint64_t cfile) {                    // first member of the range at or behind file offset cfile
            uint32_t lo_ = 0, hi_ = n_range;
            while (lo_ < hi_) { const uint32_t mid = lo_ + (hi_ - lo_) / 2; if (rm[mid].cpos - 18 < cfile) lo_ = mid + 1; else hi_ = mid; }
            return lo_;
        };
        uint32_t seg_total = 0;
        for (const VChunk &ch : chunks) {
            const uint32_t k = lower(ch.u >> 16);
            // the seek lands on no member: upstream's next read fails, nothing behind it is read
            if (k >= n_range || rm[k].cpos - 18 != (ch.u >> 16)) break;
            SegChunk sc; memset(&sc, 0, sizeof sc);
            sc.a = rm[k].upos - upos_lo + std::min<uint64_t>(ch.u & 0xffff, usize(k));
            const uint32_t kv = lower(ch.v >> 16);
            if (kv >= n_range) sc.b = total;
            else sc.b = rm[kv].upos - upos_lo + (rm[kv].cpos - 18 == (ch.v >> 16) ? std::min<uint64_t>(ch.v & 0xffff, usize(kv)) : 0);
            if (sc.b <= sc.a) sc.b = sc.a + 1;                              // the first record behind a seek is read whatever the chunk's end says
            const uint32_t nb = next_bad[k];
            sc.dlim = nb < n_range ? rm[nb].upos - upos_lo : total;
            sc.seg_base = seg_total;
            const uint64_t ns = (sc.b - sc.a + seg_bytes - 1) / seg_bytes;
            if (seg_total + ns > 0x7fffffffull) return fail(err, errlen, RGX_ERR_FORMAT, "regtools_amd: region too large\n");
            seg_total += (uint32_t)ns;
            seg_chunks.push_back(sc);
        }
        if (seg_chunks.empty()) lim = pos0;
        else {
            DevBuf &b_ch = c->buf("seg_chunks");
            HIP_TRY(b_ch.ensure(seg_chunks.size() * sizeof(SegChunk) + 64));
            HIP_TRY(hipMemcpyAsync(b_ch.p, seg_chunks.data(), seg_chunks.size() * sizeof(SegChunk), hipMemcpyHostToDevice, st));
            geom.chunks = b_ch.as<SegChunk>(); geom.n_chunks = (uint32_t)seg_chunks.size();
        }
        mark("chunk table");
    }

    return kGoOn;
}

int EventsRun::stage_framing() {
    // -- record framing ------------------------------------------------------------------------------------------------
    arena = c->buf("arena").as<uint8_t>();
    span = lim - pos0;
    n_seg = (uint32_t)((span + seg_bytes - 1) / seg_bytes);
    geom.pos0 = pos0; geom.lim = lim; geom.data_end = lim; geom.seg_bytes = seg_bytes;
    const int env_lite = 1;
    lite_walk = env_lite && !c->walk_strict;
    geom.lite_walk = lite_walk ? 1u : 0u;
    if (geom.chunks) {
        span = 0;
        for (const SegChunk &sc : seg_chunks) span += sc.b - sc.a;
        const SegChunk &lastc = seg_chunks.back();
        n_seg = lastc.seg_base + (uint32_t)((lastc.b - lastc.a + seg_bytes - 1) / seg_bytes);
        geom.data_end = total;
    }
    n_rec = 0;
    DevBuf &b_seg = c->buf("seg"), &b_tmp = c->buf("tmp");
    HIP_TRY(hipEventRecord(c->ev[2], st));
    // Early tail (round 4): while the side stream's launch still inflates the members of the last upload chunks, the segments that lie wholly
    // in front of their part of the arena (one member's margin: a walk only ever reads the 36 bytes behind its segment, a guess that
    // reads further is only a guess) are framed, verified and decoded -- exact for the same reason the whole chain is: segment 0 starts at
    // an exact offset.  Plain whole-file calls on 16 KiB segments only; anything unusual in the prefix (the chain ends there, sweeps beyond
    // the usual one) drops back to the one-pass order.
    sA = 0;
    emit_parts_ok = false; emit_parts = 0; emit_rows = 0; ev_lay = 0;
    memset(&ev_e, 0, sizeof ev_e);
    soa_cap = 0;
    memset(&soa, 0, sizeof soa);
    ev_base = nullptr; long_list = nullptr;
    if (n_seg) {
        const size_t per = (size_t)n_seg;
        HIP_TRY(b_seg.ensure(per * (8 + 8 + 4) * 2 + per * 4 + per * 12 + 64));
        HIP_TRY(c->buf("seg_cp").ensure(per * kSegCpSlots * 2 + 64));
        seg_cp = c->buf("seg_cp").as<uint16_t>();
        uint8_t *q = b_seg.as<uint8_t>();
        for (int k = 0; k < 2; ++k) { seg_start[k] = (uint64_t *)q; q += per * 8; seg_exit[k] = (uint64_t *)q; q += per * 8; }
        for (int k = 0; k < 2; ++k) { seg_cnt[k] = (uint32_t *)q; q += per * 4; }
        seg_base = (uint32_t *)q; q += per * 4;
        seg_iter_e = (uint32_t *)q; q += per * 4; seg_long_e = (uint32_t *)q; q += per * 4; seg_long_base_e = (uint32_t *)q;
        HIP_TRY(b_tmp.ensure(scan_tmp_words(n_seg) * 4 + 64));
        bool early = split_B && spec && !geom.chunks && seg_bytes == kSegBytes && cut_hi == UINT64_MAX && !empty_stream && lim == total;
        const bool env_early_emit = true;
        const bool small_ok = overlap_knobs().early_small;
        uint32_t waves_done = 0;
        for (size_t j = 0; early && j < early_parts.size(); ++j) {
            const EarlyPart &ep = early_parts[j];
            if (ep.upos <= pos0 + 2 * (uint64_t)kBgzfMaxBlock) continue;
            uint32_t sJ = (uint32_t)std::min<uint64_t>(n_seg, (ep.upos - kBgzfMaxBlock - pos0) / seg_bytes);
            if (small_ok ? (sJ < sA + 2 || n_seg - sJ < 1) : (sJ < sA + 1024 || n_seg - sJ < 64)) continue;
            // the stream waits until every wave of parts 0..j has finished (the counters of the parts are waited for in turn)
            HIP_TRY(hipMemsetAsync(d_sc + 83, 0, 4, st));
            for (size_t i = 0; i <= j; ++i) {
                const uint32_t w_end = early_parts[i].waves, w_beg = i ? early_parts[i - 1].waves : 0u;
                if (w_end > waves_done) { launch_wait_done(c->buf("gate_done").as<uint32_t>() + i, w_end - w_beg, d_sc + 83, st); waves_done = w_end; }
            }
            HIP_TRY(hipMemcpyAsync(h_sc + 83, d_sc + 83, 4, hipMemcpyDeviceToHost, st));      // (read behind the framing's first wait for the stream)
            if (trace) fprintf(stderr, "[rgx trace] early tail: part %zu: members < %u, segments [%u, %u) of %u\n", j, ep.members, sA, sJ, n_seg);
            bool ended_J = false;
            const uint32_t sweeps0 = P.framing_sweeps;
            const int rcJ = frame(sJ, sA, ended_J);
            if (rcJ != -1) return rcJ;
            const uint32_t n_rec_J = h_sc[3];
            // Where the last record that STARTS in the prefix ends = the verified chain's exit from its last segment.  The margin between the prefix
            // and the part of the arena still being inflated is one member; a record that reaches past it (a CIGAR of tens of thousands of
            // operations, a read of tens of kilobases) would be decoded from bytes that may not be there yet: such a file takes the one-pass order.
            uint64_t exit_J = 0;
            HIP_TRY(hipMemcpyAsync(h_sc + 92, seg_exit[cur] + (sJ - 1), 8, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            memcpy(&exit_J, h_sc + 92, 8);
            const bool reaches_J = exit_J > ep.upos;
            if (reaches_J && trace) fprintf(stderr,
                "[rgx trace] early tail: a record of the prefix ends at %llu, behind the inflated part (%llu): one pass\n", (unsigned long long)exit_J,
                (unsigned long long)ep.upos);
            const bool slow_J = P.framing_sweeps - sweeps0 > 2;
            P.framing_sweeps = sweeps0;                               // (the sweeps over everything, below, are the call's count)
            const uint64_t span_J = (uint64_t)sJ * seg_bytes;
            if (h_sc[83]) { c->early_distrust = true; if (trace) fprintf(stderr,
                "[rgx trace] early tail: a wait for a part's waves timed out, this context no longer uses it\n"); }
            if (h_sc[83] || ended_J || slow_J || reaches_J || !n_rec_J || span_J / n_rec_J > kSparseRecordBytes ||
                // not the plain case: one pass over everything below
                (sA && n_rec_J > soa_cap)) { early = false; sA = 0; emit_parts_ok = false; emit_parts = 0; emit_rows = 0; break; }
            if (!sA) {
                // rows for the whole file, estimated from the first part (+ 1/8); when the estimate turns out short the decode is simply made again below
                HIP_TRY(soa_layout((size_t)((double)n_rec_J * ((double)n_seg / sJ) * 1.125) + 65536));
                cfg.insane_out = nullptr;
                if (lite_walk) { cfg.insane_out = d_sc + 82; HIP_TRY(hipMemsetAsync(d_sc + 82, 0, 4, st)); h_sc[82] = 0; }
            }
            launch_decode_seg(arena, geom, sJ, seg_start[cur], seg_base, seg_cnt[cur], cfg, soa, seg_iter_e, seg_long_e, seg_cp, /*staged=*/true, st, sA);
            // ... and its junction events emitted, into the events block the context's last call left (no count is known yet, so nothing can be
            // sized: a first call, or a block that turns out too small, emits everything at the end as before).  ev_base counts from the part's
            // first row; the totals of the parts stay on the device, k_emit_short adds those in front of its part.
            if (!sA) {
                DevBuf &b_ev0 = c->buf("events");
                ev_lay = b_ev0.cap > 256 ? (b_ev0.cap - 256) / 33 : 0;
                emit_parts_ok = env_early_emit && ev_lay >= 4096;
                if (emit_parts_ok) { HIP_TRY(b_tmp.ensure(scan_tmp_words((uint32_t)std::min<size_t>(soa_cap, 0xffffffffu)) * 4 + 64));
                    ev_e = ev_layout(b_ev0.as<uint8_t>(), ev_lay); }
            }
            if (emit_parts_ok && emit_parts < kGateParts - 1) {
                launch_scan_u32(soa.n_ev + emit_rows, ev_base + emit_rows, n_rec_J - emit_rows, d_sc + 84 + emit_parts, b_tmp.as<uint32_t>(), st);
                launch_emit_short(arena, n_rec_J, cfg, soa, ev_base, ev_e, st, emit_rows, d_sc + 84, emit_parts, (uint32_t)std::min<size_t>(ev_lay,
                    0xffffffffu));
                emit_rows = n_rec_J; ++emit_parts;
            }
            sA = sJ;
            mark("early tail: part framed and decoded");
        }
        HIP_TRY(join_B());
        const int rcF = frame(n_seg, sA, chain_ended);
        if (rcF != -1) return rcF;
        n_rec = h_sc[3];
    } else HIP_TRY(join_B());
    if (spec && !n_seg) {                                     // (no framing, no sync yet: the inflate's verdict is still due)
        HIP_TRY(hipMemcpyAsync(h_sc, d_sc, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(h_sc + kStatusEarly, d_sc + kStatusEarly, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (h_sc[0] != 0xffffffffu || h_sc[kStatusEarly] != 0xffffffffu) {
            HIP_TRY(complete_upload());
            HIP_TRY(hipStreamSynchronize(copy_q));
            const int rc2 = prepare_events(c, d_bam, nullptr, bam_len, bai, bai_len, p, want_read_span, P, err, errlen, nullptr, false, region_to_file_end);
            P.t_begin = t_begin;
            return rc2;
        }
    }
    if (chain_ended) P.stream_ended = true;
    if (chain_ended && geom.chunks && m_hi < n_members_all && !region_to_file_end) {
        // a chunk's chain stopped -- possibly only because a record runs past the members the index asked for (an index that does not
        // describe this file): once more with everything up to the end of the file inflated
        mark("region: chain ended, re-reading to the end of the file");
        if (overlap) { HIP_TRY(complete_upload()); HIP_TRY(hipStreamSynchronize(copy_q)); }
        const int rc2 = prepare_events(c, d_bam, nullptr, bam_len, bai, bai_len, p, want_read_span, P, err, errlen, d_true_sizes, false, true);
        P.t_begin = t_begin;
        return rc2;
    }
    HIP_TRY(hipEventRecord(c->ev[3], st));
    mark("framing (sync)");

    return kGoOn;
}

int EventsRun::stage_decode() {
    // -- decode + count -----------------------------------------------------------------------------------------------------
    DevBuf &b_tmp = c->buf("tmp");
    n_events = 0; n_long = 0;
    n_iterated = 0;
    if (n_rec) {
        const size_t R = n_rec;
        // (early tail: the prefix is decoded already, into columns laid out for an estimate of the row count; when that was short, or the chain
        //  ended after all, everything is decoded again)
        uint32_t s_from = sA;
        if (sA && (R > soa_cap || chain_ended)) s_from = 0;
        if (!s_from) { HIP_TRY(soa_layout(R)); emit_parts_ok = false; }
        HIP_TRY(b_tmp.ensure(scan_tmp_words(n_rec) * 4 + 64));
        // per-segment outputs (no hot atomics): reuse the spare segment arrays as seg_iter / seg_long
        uint32_t *seg_iter = sA ? seg_iter_e : seg_cnt[cur ^ 1], *seg_long = sA ? seg_long_e : (uint32_t *)seg_start[cur ^ 1], *seg_long_base = sA ?
            seg_long_base_e : (uint32_t *)seg_exit[cur ^ 1];
        // the iterator's end rule only where the iterator's chunks are followed (hts_itr_next, hts.c:1946-1950): the first pass finds the
        // first record that ends the iteration and the last one that passed the overlap test; when one of those lies behind the other
        // (records out of order -- no indexer writes such a file) the pass is repeated with the stop in place
        if (geom.chunks) {
            cfg.stop_out = d_sc + 80; cfg.stop_index = 0xffffffffu;
            HIP_TRY(hipMemsetAsync(d_sc + 80, 0xff, 4, st));
            HIP_TRY(hipMemsetAsync(d_sc + 81, 0, 4, st));
        }
        if (!s_from) {
            cfg.insane_out = nullptr;
            if (lite_walk) { cfg.insane_out = d_sc + 82; HIP_TRY(hipMemsetAsync(d_sc + 82, 0, 4, st)); h_sc[82] = 0; }
        }
        for (int pass = 0; pass < 2; ++pass) {
            launch_decode_seg(arena, geom, n_seg, seg_start[cur], seg_base, seg_cnt[cur], cfg, soa, seg_iter, seg_long, seg_cp,
                              /*staged=*/span / n_rec <= kSparseRecordBytes, st, s_from);
            if (emit_parts_ok && s_from) {
                // the last part's events, emitted like the others'; the event total = the parts' totals
                launch_scan_u32(soa.n_ev + emit_rows, ev_base + emit_rows, n_rec - emit_rows, d_sc + 84 + emit_parts, b_tmp.as<uint32_t>(), st);
                launch_emit_short(arena, n_rec, cfg, soa, ev_base, ev_e, st, emit_rows, d_sc + 84, emit_parts, (uint32_t)std::min<size_t>(ev_lay, 0xffffffffu));
                HIP_TRY(hipMemcpyAsync(h_sc + 84, d_sc + 84, 4 * kGateParts, hipMemcpyDeviceToHost, st));
            } else launch_scan_u32(soa.n_ev, ev_base, n_rec, d_sc + 4, b_tmp.as<uint32_t>(), st);
            launch_scan_u32(seg_iter, seg_iter, n_seg, d_sc + 8, b_tmp.as<uint32_t>(), st);
            launch_scan_u32(seg_long, seg_long_base, n_seg, d_sc + 5, b_tmp.as<uint32_t>(), st);
            HIP_TRY(hipMemcpyAsync(h_sc + 4, d_sc + 4, 24, hipMemcpyDeviceToHost, st));
            if (cfg.stop_out) HIP_TRY(hipMemcpyAsync(h_sc + 80, d_sc + 80, 8, hipMemcpyDeviceToHost, st));
            if (cfg.insane_out) HIP_TRY(hipMemcpyAsync(h_sc + 82, d_sc + 82, 4, hipMemcpyDeviceToHost, st));
            if (cfg.abort_out) HIP_TRY(hipMemcpyAsync(h_sc + 96, d_sc + 96, 4, hipMemcpyDeviceToHost, st));
            if (cfg.odd_count) HIP_TRY(hipMemcpyAsync(h_sc + 97, d_sc + 97, 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            if (cfg.insane_out && h_sc[82]) {
                // a record the reference's reader would not have accepted (sam.c:421-423) lies on the chain the block_size walk followed:
                // the whole call again with the framing making the full test (damaged files only)
                mark("decode: a record fails bam_read1's test, starting over with the full walk");
                if (overlap) { HIP_TRY(complete_upload()); HIP_TRY(hipStreamSynchronize(copy_q)); }
                c->walk_strict = true;
                const int rc2 = prepare_events(c, d_bam, nullptr, bam_len, bai, bai_len, p, want_read_span, P, err, errlen, d_true_sizes, false,
                    region_to_file_end);
                c->walk_strict = false;
                P.t_begin = t_begin;
                return rc2;
            }
            if (pass || !cfg.stop_out || h_sc[80] == 0xffffffffu || h_sc[81] <= h_sc[80] + 1) break;
            cfg.stop_index = h_sc[80];
        }
        // an iterated read (in front of the record that ends the iteration) whose strand tag upstream cannot get at
        if (cfg.abort_out && h_sc[96] != 0xffffffffu && h_sc[96] < (cfg.stop_out ? cfg.stop_index : 0xffffffffu))
            return fail(err, errlen, RGX_ERR_ABORT,
                "regtools_amd: record %u has an auxiliary field of unknown type in front of its strand or barcode tag: the reference abort()s here\n",
                h_sc[96]);
        n_events = h_sc[4]; n_long = h_sc[5];
        if (emit_parts_ok && s_from) {
            uint64_t tot = 0;
            for (uint32_t k = 0; k <= emit_parts && k < kGateParts; ++k) tot += h_sc[84 + k];
            if (tot > ev_lay || tot > 0xffffffffull || n_long) {
                // the recycled block was too small after all (or wave-per-read rows want their global slots): everything once more, the plain way
                emit_parts_ok = false;
                launch_scan_u32(soa.n_ev, ev_base, n_rec, d_sc + 4, b_tmp.as<uint32_t>(), st);
                HIP_TRY(hipMemcpyAsync(h_sc + 4, d_sc + 4, 4, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                n_events = h_sc[4];
            } else n_events = (uint32_t)tot;
        }
        n_iterated = h_sc[8];
        if (cfg.odd_count && h_sc[97]) {
            // damaged files only: the marked rows' (tid, pos, end) come to the host; what was counted (a second decode pass counts again) bounds the list
            const uint32_t cap = h_sc[97];
            DevBuf &b_odd = c->buf("odd_aux");
            HIP_TRY(b_odd.ensure((size_t)cap * 12 + 16));
            HIP_TRY(hipMemsetAsync(d_sc + 97, 0, 4, st));
            launch_collect_odd_aux(arena, soa, n_rec, cap, d_sc + 97, b_odd.as<int32_t>(), st);
            HIP_TRY(hipMemcpyAsync(h_sc + 97, d_sc + 97, 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            const uint32_t n_odd = std::min(h_sc[97], cap);
            std::vector<int32_t> rows((size_t)n_odd * 3);
            if (n_odd) { HIP_TRY(hipMemcpyAsync(rows.data(), b_odd.p, (size_t)n_odd * 12, hipMemcpyDeviceToHost, st)); HIP_TRY(hipStreamSynchronize(st)); }
            for (uint32_t k = 0; k < n_odd; ++k) P.odd_aux.push_back(Prep::OddAux{rows[3 * (size_t)k], rows[3 * (size_t)k + 1], rows[3 * (size_t)k + 2]});
        }
        // this shard read the record that ends the iteration (hts.c:1946-1950)
        if (geom.chunks && p->n_shards > 1 && h_sc[80] != 0xffffffffu) P.stream_ended = true;
        if (n_long) launch_long_fill(n_seg, seg_base, seg_cnt[cur], seg_long_base, cfg, soa, long_list, st);
    }
    HIP_TRY(hipEventRecord(c->ev[4], st));
    mark("decode+count (sync)");

    return kGoOn;
}

int EventsRun::stage_emit() {
    // -- emit -----------------------------------------------------------------------------------------------------------------
    DevBuf &b_ev = c->buf("events");
    EventSoA ev; memset(&ev, 0, sizeof ev);
    if (n_events) {
        const size_t E = n_events;
        if (emit_parts_ok && n_rec) ev = ev_e;                  // (early tail: every part's rows are out already)
        else {
            HIP_TRY(b_ev.ensure(E * (4 * 8 + 1) + 256));
            ev = ev_layout(b_ev.as<uint8_t>(), E);
            launch_emit_short(arena, n_rec, cfg, soa, ev_base, ev, st);
            launch_emit_long(arena, long_list, n_long, cfg, soa, ev_base, ev, st);
        }
        if (cfg.fa_data) {
            HIP_TRY(hipMemcpyAsync(h_sc + 64, d_sc + 64, 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            // cc:553
            if (h_sc[64]) return fail(err, errlen, RGX_ERR_FASTA, "Unable to extract FASTA sequence for position %s\n\n",
                hdr.names[(size_t)(h_sc[64] - 1)].c_str());
        }
    }
    HIP_TRY(hipEventRecord(c->ev[5], st));

    P.hdr = hdr; P.arena = arena; P.soa = soa; P.ev = ev; P.n_rec = n_rec; P.n_events = n_events; P.n_range = n_range;
    P.n_iterated = n_iterated; P.total = total; P.t_begin = t_begin;
    return RGX_OK;
}

int prepare_events(rgx_ctx *c, const uint8_t *d_bam_in, const uint8_t *h_bam, size_t bam_len, const uint8_t *bai, size_t bai_len,
                          const rgx_extract_params *p, bool want_read_span, Prep &P, char *err, size_t errlen, const uint32_t *d_true_sizes,
                          bool allow_overlap, bool region_to_file_end, const SharedMembers *shared) {
    EventsRun r{c, d_bam_in, h_bam, bam_len, bai, bai_len, p, want_read_span, P, err, errlen, d_true_sizes, allow_overlap, region_to_file_end, shared};
    return r.run();
}

