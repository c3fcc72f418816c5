// cse_kernels.hip -- gfx950 kernels of the `cis-splice-effects identify` interval work.
//   a10  k_variant_scan   one lane per variant: 7-level bin walk over the flat GTF, exon scan, splice window (min/max)
//   a11  k_junction_scan  one lane per junction: same walk, order-dependent known-donor/acceptor flags, skipped elements
//   a9   k_window_pairs   one wave per variant window: binary-search the event range, keep events whose READ overlaps it
// All three are gather-dominated integer kernels (HBM/L2 latency bound); counts are produced first, offsets by the shared
// scan, then a fill pass writes variable-length results -- deterministic, no atomics on the data path.
#include "kernels.h"

#include "cse_core.h"

#include <stdlib.h>
#include <string.h>

namespace rgx {

__device__ __forceinline__ uint32_t lane_id2() { return threadIdx.x & 63u; }

// exon records of the candidate transcripts (SURVEY 8d's E_v / E_j), one atomic per wave
__device__ __forceinline__ void wave_add_visits(uint32_t v, unsigned long long *visits) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if (visits && (threadIdx.x & 63) == 0 && v) atomicAdd(visits, (unsigned long long)v);
}

template <bool FILL>
__global__ void k_variant_scan(GtfView g, uint32_t n, const int32_t *__restrict__ chrom, const uint32_t *__restrict__ pos0, VariantOpts o,
                               uint32_t *count, const uint32_t *__restrict__ base, uint32_t *ces, uint32_t *cee, uint32_t *hit_tx, uint32_t *hit_ad,
                               unsigned long long *visits, uint32_t *last_score) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t a, b, k = 0, ev = 0, last = 0xffffffffu;
    if (i < n) {
        const uint32_t off = FILL ? base[i] : 0u;
        variant_scan(g, chrom[i], pos0[i], o, a, b, ev, [&](uint32_t t, uint32_t ann, uint32_t dist) {
            if (FILL) { hit_tx[off + k] = t; hit_ad[2 * (size_t)(off + k)] = ann; hit_ad[2 * (size_t)(off + k) + 1] = dist; }
            ++k;
        }, &last);
        if (!FILL) { count[i] = k; ces[i] = a; cee[i] = b; if (last_score) last_score[i] = last; }
    }
    if (!FILL) wave_add_visits(ev, visits);
}

template <bool FILL>
__global__ void k_junction_scan(GtfView g, uint32_t n, const int32_t *__restrict__ chrom, const uint32_t *__restrict__ js, const uint32_t *__restrict__ je,
                                const uint8_t *__restrict__ strand, uint32_t *count, const uint32_t *__restrict__ base, uint32_t *flags, uint32_t *item_kind,
                                uint32_t *item_a, uint32_t *item_b, unsigned long long *visits) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t k = 0, ev = 0;
    if (i < n) {
        JunctionFlags f;
        const uint32_t off = FILL ? base[i] : 0u;
        junction_scan(g, chrom[i], js[i], je[i], (char)strand[i], f, ev, [&](uint32_t kind, uint32_t a, uint32_t b) {
            if (FILL) { item_kind[off + k] = kind; item_a[off + k] = a; item_b[off + k] = b; }
            ++k;
        });
        if (!FILL) { count[i] = k; flags[i] = f.known_donor | f.known_acceptor << 1 | f.known_junction << 2; }
    }
    if (!FILL) wave_add_visits(ev, visits);
}

// a11, one WAVE per junction.  The lane form above walks ~450 exon records per junction (config 4) on one lane: a chain of dependent gathers
// a few hundred long, on 1,000 waves.  Here the junction's candidate transcripts -- seven contiguous ranges of the (contig, bin) table, found by
// seven lanes at once -- are dealt to the lanes, one transcript each, sixty-four per turn: a lane's chain is one transcript's exons.
// What made upstream's loop serial is kept by two wave scans per turn: a transcript is listed when the donor / acceptor / junction flags of ALL
// transcripts visited so far (itself included) are not all clear (an inclusive OR scan in visiting order, carried from turn to
// turn), and the items land in visiting order (an exclusive sum scan of the lanes' item counts).  A transcript's own flags and skipped elements
// do not depend on the flags before it (junction_vs_transcript only reads them for its return value).
__device__ __forceinline__ uint32_t wave_scan_or(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(v, d, 64); if (lane >= (uint32_t)d) v |= o; }
    return v;
}
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(v, d, 64); if (lane >= (uint32_t)d) v += o; }
    return v;
}
template <bool FILL>
__global__ __launch_bounds__(256) void k_junction_scan_wave(GtfView g, uint32_t n, const int32_t *__restrict__ chrom, const uint32_t *__restrict__ js_, const uint32_t *__restrict__ je_,
                                     const uint8_t *__restrict__ strand_, uint32_t *count, const uint32_t *__restrict__ base, uint32_t *flags, uint32_t *item_kind,
                                     uint32_t *item_a, uint32_t *item_b, uint32_t *visit_each /* exon records of junction i's candidates (66,000 waves adding to
                                     one counter took 0.57 of the pass's 0.82 ms) */) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;                                                     // (wave-uniform)
    const int32_t ch = chrom[i];
    const uint32_t js = js_[i], je = je_[i];
    const char strand = (char)strand_[i];
    uint32_t acc = 0, n_items = 0, ev = 0;                                  // wave-uniform: flags so far (bit 0 donor, 1 acceptor, 2 junction), items so far
    if (ch >= 0 && (strand == '+' || strand == '-')) {                      // '?' matches no transcript (:322-323)
        // the seven bin ranges (junction_scan, cse_core.h), one level per lane
        uint32_t lo = 0, cnt = 0;
        if (lane < 7) {
            const uint32_t sb = (js >> 14) >> (3 * lane), eb = ((uint32_t)(je - 1) >> 14) >> (3 * lane);
            if (sb <= eb) {
                const uint32_t off = bin_offset((int)lane);
                uint32_t hi;
                bin_range(g, (uint32_t)ch, sb + off, eb + off, lo, hi);
                cnt = hi - lo;
            }
        }
        uint32_t l_lo[7], l_end[7], total = 0;                              // level l holds candidates [l_end[l-1], l_end[l])
#pragma unroll
        for (int l = 0; l < 7; ++l) { l_lo[l] = __shfl(lo, l, 64); total += __shfl(cnt, l, 64); l_end[l] = total; }
        const uint32_t out0 = FILL ? base[i] : 0u;
        for (uint32_t c0 = 0; c0 < total; c0 += 64) {
            const uint32_t c = c0 + lane;
            bool mine = c < total;
            uint32_t t = 0, nex = 0;
            if (mine) {
                uint32_t j = 0, before = 0;
#pragma unroll
                for (int l = 6; l >= 0; --l) if (c < l_end[l]) { before = l ? l_end[l - 1] : 0u; j = l_lo[l] + (c - before); }
                t = g.bin_tx[j];
                mine = (char)g.tx_strand[t] == strand;
                if (mine) { nex = g.tx_n_exons[t]; ev += nex; }
            }
            const uint32_t *es = g.es + (mine ? g.tx_exon_off[t] : 0u), *ee = g.ee + (mine ? g.tx_exon_off[t] : 0u);
            // pass A: the transcript's own flags, its item count, whether its loop ran to the end (an early return lists nothing)
            JunctionFlags f{0, 0, 0};
            uint32_t k = 0;
            bool reached = false;
            if (mine && (nex > 1 || g.keep_single)) {        // (single-exon transcripts: `junctions annotate -S` only)
                const bool outside = strand == '+' ? (es[0] > je || ee[nex - 1] < js) : (ee[0] < js || es[nex - 1] > je);
                if (!outside) { reached = true; (void)junction_vs_transcript(strand, es, ee, nex, js, je, f, [&](uint32_t, uint32_t, uint32_t) { ++k; }, g.keep_single != 0); }
            }
            const uint32_t own = f.known_donor | f.known_acceptor << 1 | f.known_junction << 2;
            const uint32_t seen = wave_scan_or(own, lane) | acc;           // flags of everything visited up to and including this transcript
            const bool listed = reached && seen != 0;
            k += listed ? 1u : 0u;
            const uint32_t incl = wave_scan_add(k, lane);
            if (FILL && k) {
                uint32_t w = out0 + n_items + incl - k;
                if (k > (listed ? 1u : 0u)) {
                    JunctionFlags f2{0, 0, 0};
                    (void)junction_vs_transcript(strand, es, ee, nex, js, je, f2, [&](uint32_t kind, uint32_t a, uint32_t b) { item_kind[w] = kind; item_a[w] = a; item_b[w] = b; ++w; }, g.keep_single != 0);
                }
                if (listed) { item_kind[w] = ITEM_TX; item_a[w] = t; item_b[w] = 0u; }
            }
            acc = __shfl(seen, 63, 64);
            n_items += __shfl(incl, 63, 64);
        }
    }
    if (!FILL) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) ev += __shfl_xor(ev, d, 64);
        if (lane == 0) { count[i] = n_items; flags[i] = acc; if (visit_each) visit_each[i] = ev; }
    }
}

// longest reference span of a read that supports an event: bounds how far before a window its overlapping reads can start
__global__ __launch_bounds__(256) void k_max_span(EventSoA ev, uint32_t n, uint32_t *out) {
    __shared__ uint32_t s_max[4];
    uint32_t m = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) { const uint32_t d = ev.rend[i] - ev.rpos[i]; m = m > d ? m : d; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = __shfl_xor(m, d, 64); m = m > o ? m : o; }
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) { m = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3])); atomicMax(out, m); }
}

// first event index whose (tid, read pos) >= (t, p); events are in file order = (tid, pos) order of the sorted BAM
__device__ __forceinline__ uint32_t ev_lower_bound(const EventSoA &ev, uint32_t n, uint32_t t, uint32_t p) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        const uint32_t mt = ev.tid[mid], mp = ev.rpos[mid];
        if (mt < t || (mt == t && (int32_t)mp < (int32_t)p)) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Window w = reads with tid == t, pos < end, endpos > beg (hts.c:1946-1957).  Pairs (window, event) are produced window-major
// and in event (file) order inside a window, so the later stable group-by sees every window's reads in file order.
// A window's candidate range -- events whose read starts in [beg - longest read span, end) -- is found once per window (k_window_ranges, the fill pass reuses it) and cut into kWinSlices slices,
// one workgroup of four waves each: a window over a highly expressed gene has orders of magnitude more candidates than the median one.  count / base are indexed
// [window * kWinSlices + slice]; a window's pairs are its slices' pairs in slice order.
// the candidate range of every window, one lane each (two binary searches over the events: ~50 dependent loads, done once per window here
// instead of once per workgroup of the pair kernel)
__global__ void k_window_ranges(EventSoA ev, uint32_t n_events, uint32_t n_win, const int32_t *__restrict__ w_tid, const int32_t *__restrict__ w_beg,
                                const int32_t *__restrict__ w_end, const uint32_t *__restrict__ max_span, uint32_t *w_lo, uint32_t *w_hi) {
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_win) return;
    const int32_t t = w_tid[w], beg = w_beg[w], end = w_end[w];
    const uint32_t span = *max_span;
    const int32_t lo_pos = beg > (int32_t)span ? beg - (int32_t)span : 0;
    w_lo[w] = ev_lower_bound(ev, n_events, (uint32_t)t, (uint32_t)lo_pos);
    w_hi[w] = ev_lower_bound(ev, n_events, (uint32_t)t, (uint32_t)end);
}

// Round 5: two kernels.  Most windows have a handful of candidates (config 4: 19 k windows, 2.6 M pairs, a median window a few dozen events), a few -- over
// a highly expressed gene -- hundreds of thousands.  A WAVE takes a small window whole (its four slices in turn: one launch of n_win / 4 workgroups, no
// barrier; the four-workgroups-per-window form was 77 k workgroups of one step each -- the launch WAS its workgroups: 0.016 of the HBM line), the
// windows above kBigWindow candidates keep the workgroup-per-slice form, a fixed grid striding over the windows and skipping the small ones.
constexpr uint32_t kBigWindow = 4096;
template <bool FILL>
__global__ __launch_bounds__(256) void k_window_pairs_small(EventSoA ev, uint32_t n_win, const int32_t *__restrict__ w_beg, const uint32_t *__restrict__ w_lo,
                                                            const uint32_t *__restrict__ w_hi, uint32_t *count, const uint32_t *__restrict__ base, uint32_t *pair_ev,
                                                            uint32_t *pair_win) {
    const uint32_t lane = threadIdx.x & 63u, w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_win) return;
    const uint32_t lo = w_lo[w], hi = w_hi[w];
    const uint64_t len = hi - lo;
    if (len > kBigWindow) return;                                             // (k_window_pairs' window)
    const int32_t beg = w_beg[w];
    for (uint32_t sl = 0; sl < kWinSlices; ++sl) {
        const uint32_t a = lo + (uint32_t)(len * sl / kWinSlices), b = lo + (uint32_t)(len * (sl + 1) / kWinSlices);
        uint32_t out = FILL ? base[(size_t)w * kWinSlices + sl] : 0u, total = 0;
        for (uint32_t e0 = a; e0 < b; e0 += 64) {                             // (wave-uniform trip count)
            const uint32_t e = e0 + lane;
            const bool keep = e < b && (int32_t)ev.rend[e] > beg;             // pos < end holds for the whole range
            const uint64_t m = __ballot(keep);
            if (FILL && keep) { const uint32_t k = out + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)); pair_ev[k] = e; pair_win[k] = w; }
            out += (uint32_t)__popcll(m); total += (uint32_t)__popcll(m);
        }
        if (!FILL && lane == 0) count[(size_t)w * kWinSlices + sl] = total;
    }
}

// (sixteen waves per workgroup: a window over the file's most expressed gene holds hundreds of thousands of candidates -- config 4: 600 k -- and the
//  launch is as long as that window's four slices: 117 us with four waves each, a quarter of that with sixteen)
constexpr uint32_t kBigBlock = 1024;
template <bool FILL>
__global__ __launch_bounds__(kBigBlock) void k_window_pairs(EventSoA ev, uint32_t n_win, const int32_t *__restrict__ w_beg, const uint32_t *__restrict__ w_lo,
                                                            const uint32_t *__restrict__ w_hi, uint32_t *count, const uint32_t *__restrict__ base, uint32_t *pair_ev,
                                                            uint32_t *pair_win) {
    __shared__ uint32_t wave_cnt[kBigBlock / 64];
    const uint32_t sl = blockIdx.y, lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    for (uint32_t w = blockIdx.x; w < n_win; w += gridDim.x) {                // (everything below is uniform over the workgroup)
        const uint32_t lo = w_lo[w], hi = w_hi[w];
        const uint64_t len = hi - lo;
        if (len <= kBigWindow) continue;                                      // (k_window_pairs_small's window)
        const int32_t beg = w_beg[w];
        const uint32_t a = lo + (uint32_t)(len * sl / kWinSlices), b = lo + (uint32_t)(len * (sl + 1) / kWinSlices);
        uint32_t out = FILL ? base[(size_t)w * kWinSlices + sl] : 0u, total = 0;
        for (uint32_t e0 = a; e0 < b; e0 += kBigBlock) {                      // (block-uniform trip count)
            const uint32_t e = e0 + threadIdx.x;
            const bool keep = e < b && (int32_t)ev.rend[e] > beg;             // pos < end holds for the whole range
            const uint64_t m = __ballot(keep);
            if (lane == 0) wave_cnt[wv] = (uint32_t)__popcll(m);
            __syncthreads();
            uint32_t before = 0, all = 0;
#pragma unroll
            for (uint32_t k = 0; k < kBigBlock / 64; ++k) { const uint32_t ck = wave_cnt[k]; before += k < wv ? ck : 0u; all += ck; }
            if (FILL && keep) {
                const uint32_t k = out + before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                pair_ev[k] = e; pair_win[k] = w;
            }
            out += all; total += all;
            __syncthreads();
        }
        if (!FILL && threadIdx.x == 0) count[(size_t)w * kWinSlices + sl] = total;
    }
}

// `cis-splice-effects associate` (cis_splice_effects_associator.cc:261-272): window w keeps every junction of its contig whose start
// or end lies inside [ces, cee].  Junctions are bucketed by contig (file order kept inside a bucket), pairs come out window-major
// and in file order inside a window -- the insertion order of the reference's set<Junction>.
template <bool FILL>
__global__ __launch_bounds__(64) void k_assoc_pairs(uint32_t n_win, const int32_t *__restrict__ w_chrom, const uint32_t *__restrict__ w_ces,
                                                    const uint32_t *__restrict__ w_cee, const uint32_t *__restrict__ chrom_off,
                                                    const uint32_t *__restrict__ j_start, const uint32_t *__restrict__ j_end, uint32_t *count,
                                                    const uint32_t *__restrict__ base, uint32_t *pair_j, uint32_t *pair_win) {
    const uint32_t w = blockIdx.x, lane = threadIdx.x;
    if (w >= n_win) return;
    const int32_t ch = w_chrom[w];
    uint32_t total = 0, out = FILL ? base[w] : 0u;
    if (ch >= 0) {
        const uint32_t ces = w_ces[w], cee = w_cee[w], lo = chrom_off[ch], hi = chrom_off[ch + 1];
        for (uint32_t j0 = lo; j0 < hi; j0 += 64) {
            const uint32_t j = j0 + lane;
            bool keep = false;
            if (j < hi) { const uint32_t s = j_start[j], e = j_end[j]; keep = (s >= ces && s <= cee) || (e <= cee && e >= ces); }
            const uint64_t m = __ballot(keep);
            if (FILL && keep) { const uint32_t k = out + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)); pair_j[k] = j; pair_win[k] = w; }
            out += (uint32_t)__popcll(m); total += (uint32_t)__popcll(m);
        }
    }
    if (!FILL && lane == 0) count[w] = total;
}

// the events of the pairs, re-keyed by window (group word = window index)
__global__ void k_pair_gather(EventSoA ev, const uint32_t *__restrict__ pair_ev, const uint32_t *__restrict__ pair_win, uint32_t n, EventSoA out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t e = pair_ev[i];
    out.tid[i] = pair_win[i]; out.start[i] = ev.start[e]; out.ilen_cls[i] = ev.ilen_cls[e]; out.ts[i] = ev.ts[e]; out.te[i] = ev.te[e]; out.strand[i] = ev.strand[e];
}

void launch_variant_scan(bool fill, GtfView g, uint32_t n, const int32_t *chrom, const uint32_t *pos0, VariantOpts o, uint32_t *count, const uint32_t *base,
                         uint32_t *ces, uint32_t *cee, uint32_t *hit_tx, uint32_t *hit_ad, unsigned long long *visits, hipStream_t stream, uint32_t *last_score) {
    if (!n) return;
    if (fill) hipLaunchKernelGGL(k_variant_scan<true>, dim3((n + 127) / 128), dim3(128), 0, stream, g, n, chrom, pos0, o, count, base, ces, cee, hit_tx, hit_ad, visits, last_score);
    else hipLaunchKernelGGL(k_variant_scan<false>, dim3((n + 127) / 128), dim3(128), 0, stream, g, n, chrom, pos0, o, count, base, ces, cee, hit_tx, hit_ad, visits, last_score);
}
void launch_junction_scan(bool fill, GtfView g, uint32_t n, const int32_t *chrom, const uint32_t *js, const uint32_t *je, const uint8_t *strand, uint32_t *count,
                          const uint32_t *base, uint32_t *flags, uint32_t *item_kind, uint32_t *item_a, uint32_t *item_b, unsigned long long *visits, uint32_t *visit_each,
                          hipStream_t stream) {
    if (!n) return;
    static const bool lane_form = [] { const char *e = getenv("REGTOOLS_AMD_JSCAN"); return e && !strcmp(e, "lane"); }();   // (tests: the one-lane-per-junction form, against which the wave form is checked)
    if (!lane_form) {
        if (fill) hipLaunchKernelGGL(k_junction_scan_wave<true>, dim3((n + 3) / 4), dim3(256), 0, stream, g, n, chrom, js, je, strand, count, base, flags, item_kind, item_a, item_b, visit_each);
        else hipLaunchKernelGGL(k_junction_scan_wave<false>, dim3((n + 3) / 4), dim3(256), 0, stream, g, n, chrom, js, je, strand, count, base, flags, item_kind, item_a, item_b, visit_each);
        return;
    }
    if (visit_each && !fill) (void)hipMemsetAsync(visit_each, 0, (size_t)n * 4, stream);            // (the lane form counts into *visits)
    if (fill) hipLaunchKernelGGL(k_junction_scan<true>, dim3((n + 127) / 128), dim3(128), 0, stream, g, n, chrom, js, je, strand, count, base, flags, item_kind, item_a, item_b, visits);
    else hipLaunchKernelGGL(k_junction_scan<false>, dim3((n + 127) / 128), dim3(128), 0, stream, g, n, chrom, js, je, strand, count, base, flags, item_kind, item_a, item_b, visits);
}
void launch_max_span(EventSoA ev, uint32_t n, uint32_t *out, hipStream_t stream) {
    if (!n) return;
    uint32_t blocks = (n + 255) / 256; if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_max_span, dim3(blocks), dim3(256), 0, stream, ev, n, out);
}
void launch_window_pairs(bool fill, EventSoA ev, uint32_t n_events, uint32_t n_win, const int32_t *w_tid, const int32_t *w_beg, const int32_t *w_end,
                         const uint32_t *max_span, uint32_t *w_lo, uint32_t *w_hi, uint32_t *count, const uint32_t *base, uint32_t *pair_ev, uint32_t *pair_win,
                         hipStream_t stream) {
    if (!n_win) return;
    const dim3 g_small((n_win + 3) / 4), g_big(std::min<uint32_t>(n_win, 256), kWinSlices);
    if (fill) {
        hipLaunchKernelGGL(k_window_pairs_small<true>, g_small, dim3(256), 0, stream, ev, n_win, w_beg, w_lo, w_hi, count, base, pair_ev, pair_win);
        hipLaunchKernelGGL(k_window_pairs<true>, g_big, dim3(kBigBlock), 0, stream, ev, n_win, w_beg, w_lo, w_hi, count, base, pair_ev, pair_win);
    } else {
        hipLaunchKernelGGL(k_window_ranges, dim3((n_win + 63) / 64), dim3(64), 0, stream, ev, n_events, n_win, w_tid, w_beg, w_end, max_span, w_lo, w_hi);
        hipLaunchKernelGGL(k_window_pairs_small<false>, g_small, dim3(256), 0, stream, ev, n_win, w_beg, w_lo, w_hi, count, base, pair_ev, pair_win);
        hipLaunchKernelGGL(k_window_pairs<false>, g_big, dim3(kBigBlock), 0, stream, ev, n_win, w_beg, w_lo, w_hi, count, base, pair_ev, pair_win);
    }
}
void launch_assoc_pairs(bool fill, uint32_t n_win, const int32_t *w_chrom, const uint32_t *w_ces, const uint32_t *w_cee, const uint32_t *chrom_off,
                        const uint32_t *j_start, const uint32_t *j_end, uint32_t *count, const uint32_t *base, uint32_t *pair_j, uint32_t *pair_win, hipStream_t stream) {
    if (!n_win) return;
    if (fill) hipLaunchKernelGGL(k_assoc_pairs<true>, dim3(n_win), dim3(64), 0, stream, n_win, w_chrom, w_ces, w_cee, chrom_off, j_start, j_end, count, base, pair_j, pair_win);
    else hipLaunchKernelGGL(k_assoc_pairs<false>, dim3(n_win), dim3(64), 0, stream, n_win, w_chrom, w_ces, w_cee, chrom_off, j_start, j_end, count, base, pair_j, pair_win);
}
void launch_pair_gather(EventSoA ev, const uint32_t *pair_ev, const uint32_t *pair_win, uint32_t n, EventSoA out, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(k_pair_gather, dim3((n + 255) / 256), dim3(256), 0, stream, ev, pair_ev, pair_win, n, out);
}

}  // namespace rgx
