#!/usr/bin/env python3
"""Source patches for tools/lab/build_coop.sh: ablations of round 3's k_inflate_coop (wrong output, same control flow).  Every patch must apply exactly once."""
import sys


def sub(s, old, new, count=1):
    assert s.count(old) == count, (old, s.count(old))
    return s.replace(old, new)


def P(fname, old, new):
    return (fname, old, new)


NO_STAGE_STORES = P("inflate_core.h", "    RGX_HD void store_chunk(uint32_t o, uint64_t l, uint64_t h) const {\n", "    RGX_HD void store_chunk(uint32_t o, uint64_t l, uint64_t h) const {\n        if (cap != 0xfffffff0u) return;\n")
NO_COOP_STORES = P("kernels.hip", "    __device__ __forceinline__ void stores() {\n", "    __device__ __forceinline__ void stores() {\n        if (lane != 0xfffffff0u) { t0 = t1 = t2 = t3 = 0xffffffffu; return; }\n")
NO_LANE_LOADS = [P("inflate_coop.h", "                v0 = ld128(s);\n", "                v0 = u32x4{(uint32_t)(uintptr_t)s, 0, 0, 0};\n"),
                 P("inflate_coop.h", "if (n > 16) v1 = ld128(s + 16);", "if (n > 16) v1 = v0;"), P("inflate_coop.h", "if (n > 32) v2 = ld128(s + 32);", "if (n > 32) v2 = v0;"),
                 P("inflate_coop.h", "if (n > 48) v3 = ld128(s + 48);", "if (n > 48) v3 = v0;"),
                 P("inflate_coop.h", "if (head) v0 = ld128(s);", "if (head) v0 = u32x4{(uint32_t)(uintptr_t)s, 0, 0, 0};"),
                 P("inflate_coop.h", "if (tail) v1 = ld128(s + head + 16 * nb);", "if (tail) v1 = u32x4{(uint32_t)(uintptr_t)s, 1, 0, 0};")]
NO_COOP_LOADS = P("kernels.hip", "D = ld128(wave_base + Tg - (pp & 0xffffu));", "D = u32x4{Tg, pp, 0, 0};")
HOT_BITS = P("inflate_core.h", "        next = ld64(p);\n    }\n    RGX_HD void ensure", "        next = ld64(in + ((p - in) & 63));\n    }\n    RGX_HD void ensure")     # the bit stream's words always from the member's first line
# every store goes to one hot place per lane (the member's first chunk / the wave's first line): the instructions and their acknowledgements stay, the misses go
DUMMY_STAGE = P("inflate_core.h", "            *(u32x4 *)(out + cb) = v;                                      // 16-byte aligned by construction", "            *(u32x4 *)(out - a + (cb & 0)) = v;")
DUMMY_COOP = P("kernels.hip", "        if (t0 != 0xffffffffu) *(u32x4 *)(wave_base + t0) = d0;\n        if (t1 != 0xffffffffu) *(u32x4 *)(wave_base + t1) = d1;\n        if (t2 != 0xffffffffu) *(u32x4 *)(wave_base + t2) = d2;\n        if (t3 != 0xffffffffu) *(u32x4 *)(wave_base + t3) = d3;\n",
               "        uint8_t *dummy = (uint8_t *)((uintptr_t)wave_base & ~(uintptr_t)15) + 16 * lane;\n        if (t0 != 0xffffffffu) *(u32x4 *)dummy = d0;\n        if (t1 != 0xffffffffu) *(u32x4 *)dummy = d1;\n        if (t2 != 0xffffffffu) *(u32x4 *)dummy = d2;\n        if (t3 != 0xffffffffu) *(u32x4 *)dummy = d3;\n")
# loads kept alive without the stores: everything loaded is folded into one word that is stored once, at the end
SINK = [P("inflate_coop.h", "    u32x4 v0 = {0, 0, 0, 0}, v1 = v0, v2 = v0, v3 = v0;", "    uint32_t sink = 0;\n    u32x4 v0 = {0, 0, 0, 0}, v1 = v0, v2 = v0, v3 = v0;"),
        P("inflate_coop.h", "            o += n; pend_len -= n;\n", "            sink ^= v0[0] ^ v0[3] ^ v1[1] ^ v2[2] ^ v3[3] ^ (n > 16 ? v1[0] : 0) ^ (n > 32 ? v2[0] : 0) ^ (n > 48 ? v3[0] : 0);\n            o += n; pend_len -= n;\n"),
        P("inflate_coop.h", "    if (active) S.flush_partial(o);                                 // the tail chunk", "    if (active && sink == 0x12345u) out[0] = 1;\n    if (active) S.flush_partial(o);                                 // the tail chunk"),
        P("kernels.hip", "    __device__ __forceinline__ void stores() {\n", "    __device__ __forceinline__ void stores() {\n        if (lane != 0xfffffff0u) { const uint32_t x = (t0 != 0xffffffffu ? d0[0] : 0) ^ (t1 != 0xffffffffu ? d1[0] : 0) ^ (t2 != 0xffffffffu ? d2[0] : 0) ^ (t3 != 0xffffffffu ? d3[0] : 0); if (x == 0x12345u) *(uint32_t *)wave_base = x; t0 = t1 = t2 = t3 = 0xffffffffu; return; }\n")]
HOT_LANE_LOADS = P("inflate_coop.h", "            const uint8_t *s = out + o - pend_dist;\n", "            const uint8_t *s = out + (pend_dist & 15u);\n")
# 16-byte look-ahead window for the bit stream (round 1's experiment, tools/lab/windowed_bitstream.patch, on today's kernel): a load per
# 8 consumed bytes instead of one per trip
BITWIN = [P("inflate_core.h", "    uint64_t next;         // the 8 bytes at p, loaded one refill ahead of their use\n    RGX_HD void init(const uint8_t *i, uint32_t n) { in = i; in_len = n; p = i; buf = 0; cnt = 0; next = ld64(p); }",
            """    uint64_t w0, w1; uint32_t off;
    RGX_HD void load_window() {
        const uint8_t *wp = (size_t)(p - in) > (size_t)in_len ? in + in_len : p;
        off = (uint32_t)(p - wp);
        const u32x4 v = ld128(wp);
        w0 = (uint64_t)v[0] | (uint64_t)v[1] << 32; w1 = (uint64_t)v[2] | (uint64_t)v[3] << 32;
    }
    RGX_HD void init(const uint8_t *i, uint32_t n) { in = i; in_len = n; p = i; buf = 0; cnt = 0; load_window(); }"""),
          P("inflate_core.h", """        buf |= next << cnt;
        p += (63u - cnt) >> 3;
        cnt |= 56u;""", """        const uint32_t sh = 8 * off;
        const uint64_t next = (sh >= 64 ? 0 : w0 >> sh) | (sh == 0 ? 0 : w1 << ((64 - sh) & 63));
        buf |= next << cnt;
        const uint32_t adv = (63u - cnt) >> 3;
        p += adv; off += adv;
        cnt |= 56u;"""),
          P("inflate_core.h", """        if ((size_t)(p - in) > (size_t)in_len + 8) p = in + in_len + 8;
        next = ld64(p);""", """        if ((size_t)(p - in) > (size_t)in_len + 8) { off -= (uint32_t)((size_t)(p - in) - ((size_t)in_len + 8)); p = in + in_len + 8; }
        if (off > 8) load_window();"""),
          P("inflate_core.h", "    RGX_HD void restart_at(const uint8_t *q) { p = q; buf = 0; cnt = 0; next = ld64(p); }", "    RGX_HD void restart_at(const uint8_t *q) { p = q; buf = 0; cnt = 0; load_window(); }"),
          P("inflate_coop.h", "br.buf = 0; br.cnt = 0; br.next = 0;", "br.buf = 0; br.cnt = 0; br.w0 = 0; br.w1 = 0; br.off = 0;"),
          P("inflate_ring.h", "br.buf = 0; br.cnt = 0; br.next = 0;", "br.buf = 0; br.cnt = 0; br.w0 = 0; br.w1 = 0; br.off = 0;")]
NT_STAGE = P("inflate_core.h", "            *(u32x4 *)(out + cb) = v;                                      // 16-byte aligned by construction", "            __builtin_nontemporal_store(v, (u32x4 *)(out + cb));")
NT_COOP = P("kernels.hip", "        if (t0 != 0xffffffffu) *(u32x4 *)(wave_base + t0) = d0;\n        if (t1 != 0xffffffffu) *(u32x4 *)(wave_base + t1) = d1;\n        if (t2 != 0xffffffffu) *(u32x4 *)(wave_base + t2) = d2;\n        if (t3 != 0xffffffffu) *(u32x4 *)(wave_base + t3) = d3;\n",
            "        if (t0 != 0xffffffffu) __builtin_nontemporal_store(d0, (u32x4 *)(wave_base + t0));\n        if (t1 != 0xffffffffu) __builtin_nontemporal_store(d1, (u32x4 *)(wave_base + t1));\n        if (t2 != 0xffffffffu) __builtin_nontemporal_store(d2, (u32x4 *)(wave_base + t2));\n        if (t3 != 0xffffffffu) __builtin_nontemporal_store(d3, (u32x4 *)(wave_base + t3));\n")
NT_BITS = P("inflate_core.h", "        next = ld64(p);\n    }\n    RGX_HD void ensure", "        next = __builtin_nontemporal_load((const u64_unaligned *)p);\n    }\n    RGX_HD void ensure")
LANE_ONLY = P("inflate_coop.h", "constexpr uint32_t kLaneCopyMax = 64;", "constexpr uint32_t kLaneCopyMax = 64; static_assert(true, \"lab\");")

VARIANTS = {
    "base": [],
    "nostore": [NO_STAGE_STORES, NO_COOP_STORES],
    "nocoopstore": [NO_COOP_STORES],
    "nostagestore": [NO_STAGE_STORES],
    "noload": NO_LANE_LOADS + [NO_COOP_LOADS],
    "nocoopload": [NO_COOP_LOADS],
    "decode_only": NO_LANE_LOADS + [NO_COOP_LOADS, NO_STAGE_STORES, NO_COOP_STORES],
    "dummystore": [DUMMY_STAGE, DUMMY_COOP],
    "dummystage": [DUMMY_STAGE],
    "lane48": [P("inflate_coop.h", "constexpr uint32_t kLaneCopyMax = 64;", "constexpr uint32_t kLaneCopyMax = 48;")],
    "lane32": [P("inflate_coop.h", "constexpr uint32_t kLaneCopyMax = 64;", "constexpr uint32_t kLaneCopyMax = 32;")],
    "loadsonly": SINK + [NO_STAGE_STORES],
    "hotlaneload": [HOT_LANE_LOADS],
    "bitwin": BITWIN,
    "bitwin_loadsonly": BITWIN + SINK + [NO_STAGE_STORES],
    # ("onesym" -- one symbol per trip -- is a run-time choice now: REGTOOLS_AMD_INFLATE_PAIRS=0 with the base build)
    "ntstage": [NT_STAGE],
    "ntcoop": [NT_COOP],
    "ntall": [NT_STAGE, NT_COOP],
    "ntbits": [NT_BITS],
    "bitwin_noload": BITWIN + NO_LANE_LOADS + [NO_COOP_LOADS],
    "hotbits": [HOT_BITS],
    "hotbits_nostore": [HOT_BITS, NO_STAGE_STORES, NO_COOP_STORES],
}

if __name__ == "__main__":
    if sys.argv[1] == "--list":
        print(" ".join(VARIANTS))
    else:
        v, d = sys.argv[1], sys.argv[2]
        for fname, old, new in VARIANTS[v]:
            p = d + "/" + fname
            text = sub(open(p).read(), old, new)
            open(p, "w").write(text)
