// tools/lab/trip_modes.cpp -- LAB TOOL (host, g++): trips per BGZF member of inflate_coop.h's decoder under its `mode` bits, checked against the
// members' own lengths (the one-lane host wave).   g++ -O2 -std=c++17 -o tools/lab/bin/trip_modes tools/lab/trip_modes.cpp && tools/lab/bin/trip_modes FILE.bam [max_members]
#define RGX_HOST_TRIPS
#include "../../regtools_amd/csrc/inflate_coop.h"

#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace rgx;

int main(int argc, char **argv) {
    if (argc < 2) return 1;
    const size_t max_members = argc > 2 ? (size_t)atol(argv[2]) : 1000;
    FILE *f = fopen(argv[1], "rb"); if (!f) return 1;
    fseek(f, 0, SEEK_END); size_t len = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> bam(len + 64); if (fread(bam.data(), 1, len, f) != len) return 1; fclose(f);
    std::vector<uint8_t> ref(65536 + 64), out(65536 + 64);
    std::vector<std::vector<uint8_t>> refs;
    for (uint32_t mode : {0u, 1u, 3u, 4u, 5u, 7u}) {
        size_t members = 0; uint64_t trips = 0;
        for (size_t off = 0; off + 28 <= len && members < max_members;) {
            const uint32_t bsize = (uint32_t)bam[off + 16] | (uint32_t)bam[off + 17] << 8;
            const uint32_t isize = (uint32_t)bam[off + bsize - 3] | (uint32_t)bam[off + bsize - 2] << 8 | (uint32_t)bam[off + bsize - 1] << 16 | (uint32_t)bam[off + bsize] << 24;
            HostTab T; HostCopy C; uint32_t n = 0;
            rgx_host_trips = 0;
            const int st = inflate_coop<BitReaderWin>(bam.data() + off + 18, bsize + 1 - 26, out.data(), isize, &n, T, C, true, mode);
            if (st != INF_OK || n != isize) { fprintf(stderr, "mode %u member at %zu: status %d, %u of %u bytes\n", mode, off, st, n, isize); return 2; }
            if (mode == 0) refs.emplace_back(out.begin(), out.begin() + isize);
            else if (memcmp(refs[members + (isize ? 0 : 0)].data(), out.data(), isize)) { fprintf(stderr, "mode %u member at %zu: bytes differ from mode 0\n", mode, off); return 3; }
            off += bsize + 1;
            if (!isize) { if (mode == 0) refs.pop_back(); continue; }
            ++members; trips += rgx_host_trips;
        }
        printf("mode %u: %zu members, %.1f trips per member\n", mode, members, (double)trips / members);
    }
    return 0;
}
