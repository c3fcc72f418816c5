#!/usr/bin/env python3
"""Source patches for tools/lab/build.sh.  Every patch must apply exactly once."""
import sys

def sub(s, old, new, count=1):
    assert s.count(old) == count, (old, s.count(old))
    return s.replace(old, new)

def no_copy_loads(s):
    s = sub(s, "            v0 = ld128(s);\n", "            v0 = u32x4{(uint32_t)(uintptr_t)s, 0, 0, 0};\n")
    for k in range(1, 8):
        s = sub(s, "v%d = ld128(s + %d);" % (k, 16 * k), "v%d = v0;" % k)
    return s

def no_stores(s):        # the output stage never writes (a condition the compiler cannot fold keeps the code alive)
    return sub(s, "    RGX_HD void store_chunk(uint32_t o, uint64_t l, uint64_t h) const {\n", "    RGX_HD void store_chunk(uint32_t o, uint64_t l, uint64_t h) const {\n        if (cap != 0xfffffff0u) return;\n")

def hot_loads(s):      # every copy load hits the same (cached) place: what do the loads cost when they never miss?
    return sub(s, "            const uint8_t *s = out + o - pend_dist;\n", "            const uint8_t *s = out + (pend_dist & 15u);\n")

def batch(nb):
    def f(s):
        return sub(s, "constexpr uint32_t kCopyBatch = 128;", "constexpr uint32_t kCopyBatch = %d;" % nb)
    return f

def occ16(s):
    return s
occ16.kernels2 = [("constexpr uint32_t kHotSyms = 160;", "constexpr uint32_t kHotSyms = 96;"), ("amdgpu_waves_per_eu(3, 3)", "amdgpu_waves_per_eu(4, 4)")]

def cond_wait(s):       # the explicit vmcnt(0) only on trips that issued copy loads
    return sub(s, "        __builtin_amdgcn_s_waitcnt(0x0F70);                         // vmcnt(0), expcnt/lgkmcnt untouched", "        if (copying) __builtin_amdgcn_s_waitcnt(0x0F70);")

def lds_bytes(nbytes):
    def f(s):
        return s
    f.kernels = ("constexpr uint32_t kInflateLdsBytes = kLdsDwordsPerLane * 64 * 4;", "constexpr uint32_t kInflateLdsBytes = %d;" % nbytes)
    return f

VARIANTS = {
    "base": [],
    "noload": [no_copy_loads],
    "hotload": [hot_loads],
    "nostore": [no_stores],
    "decode_only": [no_copy_loads, no_stores],
    "batch64": [batch(64)],
    "batch64_noload": [batch(64), no_copy_loads],
    "batch64_hotload": [batch(64), hot_loads],
}

if __name__ == "__main__":
    if sys.argv[1] == "--list":
        print(" ".join(VARIANTS))
    else:
        v, d = sys.argv[1], sys.argv[2]
        p = d + "/inflate_core.h"
        s = open(p).read()
        for f in VARIANTS[v]:
            s = f(s)
        open(p, "w").write(s)
        pk = d + "/kernels.hip"
        k = open(pk).read()
        for f in VARIANTS[v]:
            if hasattr(f, "kernels"):
                k = sub(k, f.kernels[0], f.kernels[1])
            for a, b in getattr(f, "kernels2", []):
                k = sub(k, a, b)
        open(pk, "w").write(k)
