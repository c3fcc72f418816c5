// lab: 133,000 two-byte reads at random places of a 2 GiB page-cached file through a private mapping, 16 threads -- the time of the reads and of
// the munmap afterwards, with and without MADV_RANDOM (does it switch fault-around off?), and the same reads as pread()s
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const char *path = argv[1];
    int fd = open(path, O_RDONLY);
    struct stat st; fstat(fd, &st);
    const size_t n = (size_t)st.st_size;
    std::mt19937_64 rng(7);
    std::vector<size_t> pos(133000);
    for (auto &p : pos) p = rng() % (n - 8);
    for (int mode = 0; mode < 3; ++mode)
        for (int rep = 0; rep < 2; ++rep) {
            char *m = nullptr;
            if (mode < 2) { m = (char *)mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0); if (mode == 1) madvise(m, n, MADV_RANDOM); }
            double t0 = now();
            std::vector<unsigned> sums(16, 0);
            std::vector<std::thread> th;
            for (int k = 0; k < 16; ++k) th.emplace_back([&, k] {
                unsigned s = 0; char b[8];
                for (size_t i = pos.size() * k / 16; i < pos.size() * (k + 1) / 16; ++i) {
                    if (mode < 2) s += (unsigned char)m[pos[i]] + (unsigned char)m[pos[i] + 1];
                    else { if (pread(fd, b, 4, (off_t)pos[i]) == 4) s += (unsigned char)b[0] + (unsigned char)b[1]; }
                }
                sums[k] = s;
            });
            for (auto &t : th) t.join();
            double t1 = now();
            if (m) munmap(m, n);
            double t2 = now();
            printf("%s: reads %.2f ms, munmap %.2f ms\n", mode == 0 ? "mmap" : mode == 1 ? "mmap + MADV_RANDOM" : "pread 4 B", t1 - t0, t2 - t1);
        }
    return 0;
}
