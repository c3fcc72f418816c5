// lab: what first-touching fresh memory costs on the GPU box's host (serial, threads, transparent huge pages)
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t n = 64u << 20;
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            double t0 = now();
            char *p = (char *)mmap(nullptr, n + (2u << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            char *q = (char *)(((uintptr_t)p + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1));
            if (mode & 1) madvise(q, n, MADV_HUGEPAGE);
            if (mode & 2) {
                std::vector<std::thread> th;
                for (int k = 0; k < 16; ++k) th.emplace_back([=] { memset(q + n / 16 * k, 0, n / 16); });
                for (auto &t : th) t.join();
            } else memset(q, 0, n);
            double t1 = now();
            munmap(p, n + (2u << 20));
            printf("mode %d (%s, %s): %.2f ms for 64 MiB\n", mode, mode & 1 ? "MADV_HUGEPAGE" : "4K pages", mode & 2 ? "16 threads" : "serial", t1 - t0);
        }
    }
    return 0;
}
