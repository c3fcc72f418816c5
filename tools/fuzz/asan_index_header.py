import ctypes, os, random, sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import bamio, csi_common
from regtools_amd import synth
emu = ctypes.CDLL('/tmp/libhostemu_asan.so')
emu.emu_index_summary.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]
emu.emu_region_span.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
emu.emu_host_header.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
rng = random.Random(1)
bam, bai, _ = synth.generate(3000, shape="fuzz", seed=2)
forms = [bai, csi_common.csi_bytes(bai), csi_common.csi_bytes(bai, compress=False), b"".join(bamio.bgzf_member(bai[i:i+0xff00]) for i in range(0,len(bai),0xff00))+bamio.EOF_MARKER]
out=(ctypes.c_uint64*5)(); t=(ctypes.c_uint64*2)(1<<16, 1<<30); got=(ctypes.c_uint64*2)(); lo=ctypes.c_uint64(); hi=ctypes.c_uint64(); name=ctypes.create_string_buffer(64)
n=0
for it in range(6000):
    f = bytearray(rng.choice(forms))
    mode = rng.random()
    if mode < 0.4:
        for _ in range(rng.choice([1,1,2,8])):
            f[rng.randrange(len(f))] = rng.randrange(256)
    elif mode < 0.7:
        f = f[:rng.randrange(len(f)+1)]
    elif mode < 0.85:
        k = rng.randrange(len(f)); f[k:k+4] = bytes([255,255,255,rng.choice([0x7f,0xff])])
    b = bytes(f)
    # exact-size heap copy so that ASan sees any over-read
    buf = ctypes.create_string_buffer(b, len(b)) if b else ctypes.create_string_buffer(1)
    emu.emu_index_summary(buf, len(b), out, t, 2, got)
    emu.emu_region_span(buf, len(b), rng.randrange(-1,6), rng.randrange(0,700000), rng.randrange(0,700000), ctypes.byref(lo), ctypes.byref(hi))
    n+=1
for it in range(1500):
    f = bytearray(bam[:rng.randrange(20, 70000)])
    if rng.random()<0.6:
        for _ in range(rng.choice([1,2,16])): f[rng.randrange(len(f))] = rng.randrange(256)
    b=bytes(f); buf = ctypes.create_string_buffer(b, len(b))
    emu.emu_host_header(buf, len(b), name, 64)
print("ok", n)
