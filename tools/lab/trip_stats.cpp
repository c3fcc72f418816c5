// tools/lab/trip_stats.cpp -- LAB TOOL (host, g++): what one lane per member costs in symbol trips on a given BAM.
// Re-runs the product's symbol loop (inflate_core.h inflate_raw's control flow, same trip rule: one copy batch OR one symbol per trip)
// with counters instead of the output stage, per BGZF member; then groups members 64 per wave in file order and in sorted order.
//   g++ -O2 -std=c++17 -o tools/lab/bin/trip_stats tools/lab/trip_stats.cpp && tools/lab/bin/trip_stats FILE.bam [batch]
#include "../../regtools_amd/csrc/inflate_core.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace rgx;

struct Stats { uint32_t trips = 0, lits = 0, matches = 0, copy_trips = 0, headers = 0, out = 0, clen = 0; uint64_t match_bytes = 0; uint32_t len_hist[10] = {0}; uint32_t dist_hist[10] = {0}; };

static int run(const uint8_t *in, uint32_t in_len, std::vector<uint8_t> &outv, uint32_t batch, Stats &S) {
    HostTab T; BitReader br; br.init(in, in_len);
    OutStage St; uint8_t *out = outv.data(); St.init(out, 65536);
    uint32_t o = 0, last = 0; int status = INF_OK; bool in_symbols = false, done = false;
    uint32_t pend_len = 0, pend_dist = 0; Code LL, DD;
    for (int k = 0; k < 16; ++k) { LL.c[k] = 0; DD.c[k] = 0; }
    for (;;) {
        ++S.trips;
        const bool copying = pend_len != 0;
        uint32_t n = 0;
        if (copying) { n = std::min(pend_len, batch); if (pend_dist < n) n = pend_dist; ++S.copy_trips; }
        uint32_t lit = 256, new_len = 0, new_dist = 0;
        if (pend_len == n && !done) {
            if (in_symbols) {
                const uint32_t v = rev15(br.peek(15)); uint32_t l; const uint32_t idx = code_lookup(LL, v, l);
                if (l == 0 || idx >= 288) return INF_BAD_CODE;
                const uint32_t sym = T.get_ll_sym(idx); br.drop(l);
                if (sym < 256) lit = sym;
                else if (sym == 256) { in_symbols = false; if (last) done = true; }
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
        }
        br.ensure(48);
        if (copying) { for (uint32_t k = 0; k < n; ++k) out[o + k] = out[o + k - pend_dist]; o += n; pend_len -= n; if (n == pend_dist) pend_dist += pend_dist; }
        if (lit < 256) { out[o++] = (uint8_t)lit; ++S.lits; }
        else if (new_len) {
            pend_len = new_len; pend_dist = new_dist; ++S.matches; S.match_bytes += new_len;
            int lb = 0; while ((8u << lb) < new_len && lb < 9) ++lb; ++S.len_hist[lb];
            int db = 0; while ((4u << (2 * db)) < new_dist && db < 9) ++db; ++S.dist_hist[db];
        }
        if (done && pend_len == 0) break;
    }
    S.out = o;
    return INF_OK;
}

int main(int argc, char **argv) {
    if (argc < 2) return 1;
    const uint32_t batch = argc > 2 ? (uint32_t)atoi(argv[2]) : 128;
    FILE *f = fopen(argv[1], "rb"); if (!f) return 1;
    fseek(f, 0, SEEK_END); size_t len = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> bam(len + 64); if (fread(bam.data(), 1, len, f) != len) return 1; fclose(f);
    std::vector<Stats> all; std::vector<uint8_t> out(65536 + 64);
    for (size_t off = 0; off + 28 <= len;) {
        const uint32_t bsize = (uint32_t)bam[off + 16] | (uint32_t)bam[off + 17] << 8;
        Stats S; S.clen = bsize + 1 - 26;
        const int st = run(bam.data() + off + 18, bsize + 1 - 18, out, batch, S);
        if (st != INF_OK) { fprintf(stderr, "member at %zu: status %d\n", off, st); return 2; }
        if (S.out) all.push_back(S);
        off += bsize + 1;
    }
    Stats tot; uint64_t trips = 0;
    for (auto &s : all) { trips += s.trips; tot.lits += s.lits; tot.matches += s.matches; tot.copy_trips += s.copy_trips; tot.headers += s.headers; tot.match_bytes += s.match_bytes; tot.out += s.out;
        for (int k = 0; k < 10; ++k) { tot.len_hist[k] += s.len_hist[k]; tot.dist_hist[k] += s.dist_hist[k]; } }
    const double M = (double)all.size();
    printf("members %zu  batch %u\nper member: trips %.0f  literals %.0f  matches %.0f  copy trips %.0f  headers %.2f  match bytes %.0f (%.1f per match)  out %.0f\n", all.size(), batch,
           trips / M, tot.lits / M, tot.matches / M, tot.copy_trips / M, tot.headers / M, tot.match_bytes / M, (double)tot.match_bytes / std::max(1u, tot.matches), tot.out / M);
    printf("match length histogram (<=8,16,32,64,128,256,..):"); for (int k = 0; k < 10; ++k) printf(" %.1f%%", 100.0 * tot.len_hist[k] / std::max(1u, tot.matches)); printf("\n");
    printf("match distance histogram (<=4,16,64,256,1k,4k,16k,64k):"); for (int k = 0; k < 10; ++k) printf(" %.1f%%", 100.0 * tot.dist_hist[k] / std::max(1u, tot.matches)); printf("\n");
    auto wave_cost = [&](const std::vector<Stats> &v) { uint64_t c = 0; for (size_t i = 0; i < v.size(); i += 64) { uint32_t mx = 0; for (size_t j = i; j < std::min(v.size(), i + 64); ++j) mx = std::max(mx, v[j].trips); c += mx; } return c; };
    const size_t waves = (all.size() + 63) / 64;
    printf("wave trips (max over 64 lanes), file order: %.0f per wave; mean lane trips %.0f -> lane utilisation %.1f%%\n", (double)wave_cost(all) / waves, trips / M, 100.0 * trips / 64.0 / wave_cost(all) * (all.size() % 64 ? 1 : 1));
    std::vector<Stats> s2 = all; std::sort(s2.begin(), s2.end(), [](const Stats &a, const Stats &b) { return a.clen < b.clen; });
    printf("  sorted by compressed length: %.0f per wave (utilisation %.1f%%)\n", (double)wave_cost(s2) / waves, 100.0 * trips / 64.0 / wave_cost(s2));
    std::sort(s2.begin(), s2.end(), [](const Stats &a, const Stats &b) { return a.trips < b.trips; });
    printf("  sorted by trips (oracle):    %.0f per wave (utilisation %.1f%%)\n", (double)wave_cost(s2) / waves, 100.0 * trips / 64.0 / wave_cost(s2));
    uint32_t mn = ~0u, mx = 0; for (auto &s : all) { mn = std::min(mn, s.trips); mx = std::max(mx, s.trips); }
    printf("trips per member: min %u max %u\n", mn, mx);
    return 0;
}
