// tests/hostemu/hostemu.cpp -- TEST INFRASTRUCTURE ONLY.
// Compiles the per-lane device cores (inflate_core.h, bam_core.h) for the host so their logic can be
// unit-tested on a machine without a GPU.  Never linked into the product library.
#include "../../regtools_amd/csrc/inflate_core.h"

extern "C" int emu_inflate(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t cap, uint32_t *out_len) {
    rgx::HostTab T;
    return rgx::inflate_raw(in, in_len, out, cap, out_len, T);
}
