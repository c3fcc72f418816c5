// tools/lab/req_model.cpp -- LAB TOOL (host, g++): where k_inflate_coop's memory requests come from, per BGZF member.
// Re-runs inflate_coop.h's trip rule on the host (a literal + the symbol behind it per trip, copies <= 64 bytes by the lane, longer ones by
// the wave) and counts, per member: symbol-list indices (how many hot symbols a lane needs in LDS), the scattered 16-byte lane requests by
// kind (stage stores, lane-copy loads, coop head / tail loads, bit-stream window loads), the chunks a trip completes, and what an LDS line
// stage of W bytes would turn into whole 64-byte line stores or serve as a copy source.
//   g++ -O2 -std=c++17 -o tools/lab/bin/req_model tools/lab/req_model.cpp && tools/lab/bin/req_model FILE.bam [max_members]
#include "../../regtools_amd/csrc/inflate_coop.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace rgx;

struct Stats {
    uint64_t trips = 0, lits = 0, matches = 0, headers = 0, out = 0;
    uint64_t ll_lookups = 0, ll_idx_hist[289] = {0};
    uint64_t lane_copy_trips = 0, lane_copy_loads = 0, coop_trips = 0, coop_chunks = 0, coop_heads = 0, coop_tails = 0;
    uint64_t chunk_done = 0, done_hist[8] = {0};        // 16-byte chunks the register stage completes (= stage stores today), per trip
    uint64_t flushes = 0;                                // flush_partial stores (copy source inside the unflushed chunk)
    uint64_t lane_src_within[4] = {0};                   // lane-copy trips whose whole source lies within the last 64 / 128 / 256 / 512 bytes
    uint64_t coop_src_within[4] = {0};
    uint64_t lines_touched_by_stage = 0;                 // distinct 64-byte lines that receive at least one stage chunk
    uint64_t lines_all_stage = 0;                        // ... of which all four chunks come from the stage (a line stage stores them as ONE request)
    uint64_t bit_refills = 0, bit_window_loads = 0;
    uint64_t short_dist_hist[6] = {0};
    uint64_t cold138_lit = 0, cold138_other = 0, cold160_lit = 0, cold160_other = 0;   // lookups in the list's cold end (138 / 160 hot entries): literals, and the rest (an extra trip each)                   // lane copies by distance: <=16, <=64, <=128, <=256, <=1024, more
};

static int run(const uint8_t *in, uint32_t in_len, std::vector<uint8_t> &outv, Stats &S) {
    HostTab T; BitReaderWin br; br.init(in, in_len);
    OutStage St; uint8_t *out = outv.data(); St.init(out, 65536);
    uint32_t o = 0, last = 0; int status = INF_OK; bool in_symbols = false, done = false;
    uint32_t pend_len = 0, pend_dist = 0; Code LL, DD;
    for (int k = 0; k < 16; ++k) { LL.c[k] = 0; DD.c[k] = 0; }
    std::vector<uint8_t> chunk_src(65536 / 16 + 8, 0);   // per 16-byte chunk: 1 = completed by the stage, 2 = written by the wave (coop body)
    const uint8_t *last_win = nullptr;
    for (;;) {
        ++S.trips;
        const bool copying = pend_len != 0;
        uint32_t n = 0, head = 0, nb = 0, tail = 0; bool coop = false;
        uint32_t done_chunks = 0;
        auto account = [&](uint32_t o0, uint32_t nbytes) {          // bytes [o0, o0 + nbytes) pass through the register stage
            const uint32_t c0 = o0 >> 4, c1 = (o0 + nbytes) >> 4;   // chunks completed: those whose last byte is written
            for (uint32_t c = c0; c < c1; ++c) { chunk_src[c] = 1; ++done_chunks; }
        };
        if (copying) {
            n = std::min(pend_len, kCoopCopyMax); if (pend_dist < n) n = pend_dist;
            coop = n > kLaneCopyMax;
            if (pend_dist < n + 16 && (o & 15u)) ++S.flushes;
            const int wi = pend_dist + 0 <= 64 ? 0 : pend_dist <= 128 ? 1 : pend_dist <= 256 ? 2 : pend_dist <= 512 ? 3 : 4;
            if (!coop) {
                ++S.lane_copy_trips; S.lane_copy_loads += (n + 15) / 16;
                for (int w = wi; w < 4; ++w) ++S.lane_src_within[w];
                ++S.short_dist_hist[pend_dist <= 16 ? 0 : pend_dist <= 64 ? 1 : pend_dist <= 128 ? 2 : pend_dist <= 256 ? 3 : pend_dist <= 1024 ? 4 : 5];
            } else {
                const uint32_t k = o & 15u;
                head = k ? 16u - k : 0u; nb = (n - head) >> 4; tail = (n - head) & 15u;
                ++S.coop_trips; S.coop_chunks += nb; if (head) ++S.coop_heads; if (tail) ++S.coop_tails;
                for (int w = wi; w < 4; ++w) ++S.coop_src_within[w];
            }
        }
        uint32_t lit = 256, lit2 = 256, new_len = 0, new_dist = 0;
        if (pend_len == n && !done) do {
            if (in_symbols) {
                uint32_t v = rev15(br.peek(15)); uint32_t l; uint32_t idx = code_lookup(LL, v, l);
                if (l == 0 || idx >= 288) return INF_BAD_CODE;
                ++S.ll_lookups; ++S.ll_idx_hist[idx];
                uint32_t sym = T.get_ll_sym(idx); br.drop(l);
                if (idx >= 138) ++(sym < 256 ? S.cold138_lit : S.cold138_other);
                if (idx >= 160) ++(sym < 256 ? S.cold160_lit : S.cold160_other);
                if (sym < 256 && br.cnt >= 48) {
                    lit = sym;
                    v = rev15(br.peek(15)); idx = code_lookup(LL, v, l);
                    if (l == 0 || idx >= 288) return INF_BAD_CODE;
                    ++S.ll_lookups; ++S.ll_idx_hist[idx];
                    sym = T.get_ll_sym(idx); br.drop(l);
                    if (idx >= 138) ++(sym < 256 ? S.cold138_lit : S.cold138_other);
                    if (idx >= 160) ++(sym < 256 ? S.cold160_lit : S.cold160_other);
                    if (sym < 256) { lit2 = sym; break; }
                } else if (sym < 256) { lit = sym; break; }
                if (sym == 256) { in_symbols = false; if (last) done = true; }
                else {
                    const uint32_t c = sym - 257;
                    if (c > 28) return INF_BAD_CODE;
                    if (c < 8) new_len = 3 + c; else if (c == 28) new_len = 258; else { const uint32_t e = (c >> 2) - 1; new_len = ((4 + (c & 3)) << e) + 3 + br.bits(e); }
                    const uint32_t dv = rev15(br.peek(15)); uint32_t dl; const uint32_t didx = code_lookup(DD, dv, dl);
                    if (dl == 0 || didx >= 32) return INF_BAD_CODE;
                    const uint32_t dsym = T.get_d_sym(didx); br.drop(dl);
                    if (dsym < 4) new_dist = 1 + dsym; else { const uint32_t e = (dsym >> 1) - 1; new_dist = ((2 + (dsym & 1)) << e) + 1 + br.bits(e); }
                }
            } else if (!copying) {
                ++S.headers;
                const int r = block_header(br, T, LL, DD, in, in_len, out, o, 65536, last, status, St);
                if (status != INF_OK) return status;
                if (r) in_symbols = true; else if (last) done = true;
            }
        } while (0);
        if (br.cnt < 48) { ++S.bit_refills; br.refill(); }
        { const uint8_t *wp = br.p - br.off; if (wp != last_win) { ++S.bit_window_loads; last_win = wp; } }
        if (copying) {
            for (uint32_t k = 0; k < n; ++k) out[o + k] = out[o + k - pend_dist];
            if (!coop) account(o, n);
            else { account(o, head); for (uint32_t j = 0; j < nb; ++j) chunk_src[((o + head) >> 4) + j] = 2; }
            o += n; pend_len -= n; if (n == pend_dist) pend_dist += pend_dist;
        }
        if (lit < 256) { out[o] = (uint8_t)lit; account(o, 1); ++o; ++S.lits; }
        if (lit2 < 256) { out[o] = (uint8_t)lit2; account(o, 1); ++o; ++S.lits; }
        if (new_len) { pend_len = new_len; pend_dist = new_dist; ++S.matches; }
        S.chunk_done += done_chunks; ++S.done_hist[std::min(done_chunks, 7u)];
        if (done && pend_len == 0) break;
    }
    S.out = o;
    for (uint32_t l = 0; l * 64 < o; ++l) {
        int st = 0; for (int c = 0; c < 4; ++c) st += chunk_src[l * 4 + c] == 1;
        if (st) ++S.lines_touched_by_stage;
        if (st == 4) ++S.lines_all_stage;
    }
    return INF_OK;
}

int main(int argc, char **argv) {
    if (argc < 2) return 1;
    const size_t max_members = argc > 2 ? (size_t)atol(argv[2]) : 4000;
    FILE *f = fopen(argv[1], "rb"); if (!f) return 1;
    fseek(f, 0, SEEK_END); size_t len = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> bam(len + 64); if (fread(bam.data(), 1, len, f) != len) return 1; fclose(f);
    Stats S; size_t members = 0; std::vector<uint8_t> out(65536 + 64);
    uint64_t max_idx_per_member_hist[289] = {0};
    for (size_t off = 0; off + 28 <= len && members < max_members;) {
        const uint32_t bsize = (uint32_t)bam[off + 16] | (uint32_t)bam[off + 17] << 8;
        Stats M;
        const int st = run(bam.data() + off + 18, bsize + 1 - 18, out, M);
        if (st != INF_OK) { fprintf(stderr, "member at %zu: status %d\n", off, st); return 2; }
        off += bsize + 1;
        if (!M.out) continue;
        ++members;
        const uint64_t *a = &M.trips; uint64_t *b = &S.trips;
        for (size_t k = 0; k < sizeof(Stats) / 8; ++k) b[k] += a[k];
        int mx = 0; for (int i = 0; i < 289; ++i) if (M.ll_idx_hist[i]) mx = i;
        ++max_idx_per_member_hist[mx];
    }
    const double M = (double)members;
    printf("members %.0f\n", M);
    printf("per member: trips %.0f literals %.0f matches %.0f headers %.2f out %.0f\n", S.trips / M, S.lits / M, S.matches / M, S.headers / M, S.out / M);
    printf("lane-copy trips %.0f (16-byte source loads %.0f)  coop trips %.0f (body chunks %.0f, heads %.0f, tails %.0f)\n", S.lane_copy_trips / M, S.lane_copy_loads / M, S.coop_trips / M, S.coop_chunks / M, S.coop_heads / M, S.coop_tails / M);
    printf("stage chunk stores %.0f  flush_partial stores %.0f   chunks completed per trip 0..7+:", S.chunk_done / M, S.flushes / M);
    for (int k = 0; k < 8; ++k) printf(" %.1f", S.done_hist[k] / M); printf("\n");
    printf("64-byte lines with a stage chunk %.0f (all four from the stage: %.0f) of %.0f lines\n", S.lines_touched_by_stage / M, S.lines_all_stage / M, S.out / M / 64);
    printf("lane copies whose source lies within the last 64/128/256/512 bytes: %.0f %.0f %.0f %.0f   coop: %.0f %.0f %.0f %.0f\n", S.lane_src_within[0] / M, S.lane_src_within[1] / M, S.lane_src_within[2] / M, S.lane_src_within[3] / M,
           S.coop_src_within[0] / M, S.coop_src_within[1] / M, S.coop_src_within[2] / M, S.coop_src_within[3] / M);
    printf("lane copies by distance <=16 <=64 <=128 <=256 <=1024 more:"); for (int k = 0; k < 6; ++k) printf(" %.0f", S.short_dist_hist[k] / M); printf("\n");
    printf("bit-stream refills %.0f, 16-byte window loads %.0f\n", S.bit_refills / M, S.bit_window_loads / M);
    uint64_t cum = 0; printf("LL list index coverage: ");
    for (int i = 0; i < 289; ++i) { cum += S.ll_idx_hist[i]; if (i == 31 || i == 47 || i == 55 || i == 63 || i == 79 || i == 95 || i == 111 || i == 127 || i == 159 || i == 287) printf(" <=%d: %.4f%%", i, 100.0 * cum / S.ll_lookups); }
    printf("\nlookups beyond index 55 / 95 / 159 per member: ");
    { uint64_t b55 = 0, b95 = 0, b159 = 0; for (int i = 0; i < 289; ++i) { if (i > 55) b55 += S.ll_idx_hist[i]; if (i > 95) b95 += S.ll_idx_hist[i]; if (i > 159) b159 += S.ll_idx_hist[i]; } printf("%.1f %.1f %.1f\n", b55 / M, b95 / M, b159 / M); }
    printf("cold lookups per member, 138 hot: %.0f literals + %.0f others (an extra trip each); 160 hot: %.0f + %.0f\n", S.cold138_lit / M, S.cold138_other / M, S.cold160_lit / M, S.cold160_other / M);
    printf("members by highest list index used: "); { uint64_t c = 0; for (int i = 0; i < 289; ++i) { c += max_idx_per_member_hist[i]; if (i == 55 || i == 95 || i == 127 || i == 159 || i == 287) printf(" <=%d: %.1f%%", i, 100.0 * c / M); } } printf("\n");
    const double scattered = (double)S.chunk_done + S.flushes + S.lane_copy_loads + S.coop_heads + S.coop_tails + S.bit_window_loads * 1.23;
    printf("scattered 16-byte lane requests per member today ~ %.0f (stage %.0f, lane-copy loads %.0f, coop head+tail loads %.0f, bit stream %.0f); coop groups %.0f x (load + store)\n",
           scattered / M, (S.chunk_done + S.flushes) / M, S.lane_copy_loads / M, (S.coop_heads + S.coop_tails) / M, S.bit_window_loads * 1.23 / M, S.coop_trips / M);
    return 0;
}
