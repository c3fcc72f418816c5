// lab: the floor of a threaded pass over a page-cached 190 MB text file on the GPU box's host: mmap (4 KiB file pages, fault-around) against
// pread into an advised anonymous buffer, T threads each touching their own range
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const char *path = argv[1];
    const int T = argc > 2 ? atoi(argv[2]) : 32;
    int fd = open(path, O_RDONLY);
    struct stat st; fstat(fd, &st);
    const size_t n = (size_t)st.st_size;
    for (int mode = 0; mode < 3; ++mode)
        for (int rep = 0; rep < 2; ++rep) {
            double t0 = now();
            std::vector<size_t> lines(T, 0);
            char *p = nullptr; size_t maplen = 0;
            if (mode == 0) { p = (char *)mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0); maplen = n; if (getenv("SEQ")) madvise(p, n, MADV_SEQUENTIAL); }
            else {
                maplen = (n + (4u << 20)) & ~(size_t)((2u << 20) - 1);
                char *raw = (char *)mmap(nullptr, maplen + (2u << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
                p = (char *)(((uintptr_t)raw + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1));
                if (mode == 2) madvise(p, maplen, MADV_HUGEPAGE);
            }
            double t1 = now();
            std::vector<std::thread> th;
            for (int k = 0; k < T; ++k) th.emplace_back([&, k] {
                size_t a = n * k / T, b = n * (k + 1) / T;
                if (mode) { size_t o = a; while (o < b) { ssize_t r = pread(fd, p + o, b - o, (off_t)o); if (r <= 0) break; o += (size_t)r; } }
                size_t c = 0;
                for (const char *q = p + a; q < p + b;) { const char *nl = (const char *)memchr(q, '\n', (size_t)(p + b - q)); if (!nl) break; ++c; q = nl + 1; }
                lines[k] = c;
            });
            for (auto &t : th) t.join();
            double t2 = now();
            size_t total = 0; for (size_t c : lines) total += c;
            printf("mode %d (%s) T=%d: map %.2f ms, threads %.2f ms, %zu lines\n", mode, mode == 0 ? "mmap file" : mode == 1 ? "pread -> 4K anon" : "pread -> huge anon", T, t1 - t0, t2 - t1, total);
            if (mode == 0) munmap(p, maplen);
        }
    return 0;
}
