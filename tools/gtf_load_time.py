#!/usr/bin/env python3
"""Host-only: time GtfModel::load (the GTF thread of `identify`) on the config-4 annotation, with its trace laps."""
import ctypes
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from regtools_amd import synth

d = tempfile.mkdtemp(prefix="rgx_gtf_", dir="/tmp")
a = synth.annotation(os.path.join(d, "c4"), int(sys.argv[1]) if len(sys.argv) > 1 else 62500, 1000, seed=4, fasta=False)
lib = ctypes.CDLL(os.path.join(ROOT, "tests", "hostemu", "libhostemu.so"))
lib.emu_gtf_dump.argtypes = [ctypes.c_char_p] * 3 + [ctypes.c_size_t]
err = ctypes.create_string_buffer(256)
os.environ["REGTOOLS_AMD_TRACE"] = "1"
for i in range(4):
    lib.emu_gtf_dump(a["gtf"].encode(), b"/nonexistent/x", err, 256)
    sys.stderr.write("--\n")
