// bigvec.h -- std::vector for the host side's large, written-once tables.
//
// What a fresh multi-megabyte std::vector costs on the GPU box's host is not the bytes but the page faults: 64 MiB first touched through 4 KiB
// pages takes 11 ms on one thread, 4.5 ms when the range is madvise(MADV_HUGEPAGE)'d (the box runs transparent huge pages in `madvise` mode),
// 0.8 ms when sixteen threads touch their own slices of such a range (tools/lab/hostmem/touch.cpp).  So: blocks of 2 MiB and more are mapped
// 2 MiB-aligned and advised, and elements are default-initialised -- a resize() of a vector of integers touches nothing, the threads that
// fill it take the faults, each on its own pages.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <sys/mman.h>

#include <new>
#include <utility>
#include <vector>

namespace rgx {

template <class T>
struct HugeAlloc {
    using value_type = T;
    static constexpr size_t kHuge = (size_t)2 << 20;
    HugeAlloc() = default;
    template <class U> HugeAlloc(const HugeAlloc<U> &) {}
    static size_t mapped(size_t bytes) { return (bytes + kHuge - 1) & ~(kHuge - 1); }
    T *allocate(size_t n) {
        const size_t bytes = n * sizeof(T);
        if (bytes < kHuge) { void *p = malloc(bytes ? bytes : 1); if (!p) throw std::bad_alloc(); return (T *)p; }
        const size_t len = mapped(bytes);
        char *raw = (char *)mmap(nullptr, len + kHuge, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (raw == (char *)MAP_FAILED) throw std::bad_alloc();
        char *p = (char *)(((uintptr_t)raw + kHuge - 1) & ~(uintptr_t)(kHuge - 1));
        if (p != raw) munmap(raw, (size_t)(p - raw));
        const size_t tail = (size_t)(raw + len + kHuge - (p + len));
        if (tail) munmap(p + len, tail);
        (void)madvise(p, len, MADV_HUGEPAGE);
        return (T *)p;
    }
    void deallocate(T *p, size_t n) {
        const size_t bytes = n * sizeof(T);
        if (bytes < kHuge) free(p); else munmap((void *)p, mapped(bytes));
    }
    template <class U> void construct(U *p) { ::new ((void *)p) U; }                                   // default-, not value-initialised
    template <class U, class... A> void construct(U *p, A &&...a) { ::new ((void *)p) U(std::forward<A>(a)...); }
    template <class U> bool operator==(const HugeAlloc<U> &) const { return true; }
    template <class U> bool operator!=(const HugeAlloc<U> &) const { return false; }
};
template <class T> using BigVec = std::vector<T, HugeAlloc<T>>;

}  // namespace rgx
