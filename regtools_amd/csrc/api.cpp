// api.cpp -- C ABI (include/regtools_amd.h) and the host orchestration of the device pipeline.
//
//   members (host BSIZE walk) -> [K] inflate -> header/BAI -> [K] segment chains + verify -> [K] fill offsets
//   -> [K] decode SoA + count -> scan -> [K] emit events -> 8 radix passes -> [K] heads/reduce/name -> 12 radix
//   passes (output order) -> D2H of the unique rows.
// There is no CPU fallback anywhere in this file: every byte of BAM payload is touched on the device only.
#include "../../include/regtools_amd.h"

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <charconv>
#include <chrono>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "cse_host.h"
#include "host_io.h"
#include "worker_pool.h"
#include <sys/stat.h>
#include "kernels.h"

using namespace rgx;

bool rgx_enable_peer(int a, int b);         // (below; also multi.cpp)

namespace {

int fail(char *err, size_t errlen, int code, const char *fmt, ...) {
    if (err && errlen) { va_list ap; va_start(ap, fmt); vsnprintf(err, errlen, fmt, ap); va_end(ap); }
    return code;
}

// Every checked HIP call also answers for the kernel launches queued since the last one (round 4): a launch whose configuration is refused
// returns its error from hipLaunchKernel, which the launch_* wrappers do not look at -- it stays with the thread until hipGetLastError reads
// it.  The pipeline synchronises (HIP_TRY(hipStreamSynchronize)) before it reads anything a kernel wrote, so a refused launch is an
// RGX_ERR_DEVICE at the next such point instead of an untouched buffer read as data.
#define HIP_TRY(expr)                                                                                       \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) return fail(err, errlen, RGX_ERR_DEVICE, "HIP error %s at %s:%d (%s)\n", hipGetErrorString(e_), __FILE__, __LINE__, #expr); \
        e_ = rgx::pending_launch_error();                                                                   \
        if (e_ != hipSuccess) return fail(err, errlen, RGX_ERR_DEVICE, "HIP error %s from a kernel launch before %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// First HIP call of an entry point: errors other code left with this host thread (another library's polling of an event, an ignored return
// of a clean-up call) are not this call's launches' -- drop them, then select the device.
#define HIP_ENTER(dev)                                                                                      \
    do { (void)rgx::pending_launch_error(); HIP_TRY(hipSetDevice(dev)); } while (0)

const char *kMsgOpen = "Unable to open BAM/SAM file.\n\n";
const char *kMsgIndex = "Unable to open BAM/SAM index. Make sure alignments are indexed\n\n";
const char *kMsgRegion = "Unable to iterate to region within BAM.\n\n";

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

uint32_t bitlen(uint32_t v) { uint32_t b = 0; while (v) { ++b; v >>= 1; } return b; }

// growable device buffer that survives across calls (workspace reuse: no hipMalloc in steady state)
// Every buffer starts kFront bytes into its allocation: k_inflate_ring's far copies load their source from up to 15 bytes in front of it
// (inflate_ring.h, kArenaFrontPad), and for the first member of an arena that is in front of the buffer.
struct AllocStats { double ms = 0; uint64_t calls = 0, bytes = 0; };
static AllocStats g_alloc_stats;                                   // (REGTOOLS_AMD_TRACE: what growing the device buffers cost a call)
// REGTOOLS_AMD_ARENA="trials[,piece_MiB]": how many fresh arenas a context's first large call times its DEFLATE launch into (default 0 = none: the trials are
// OPT-IN since round 6 -- on the driver's box five of them bought 14.5 -> 14.0 ms for 0.3 s and an arena's worth of transient memory; calibrate_arena),
// and the size of the pieces the arena's device memory is created in (default 512, 0 = one hipMalloc block; DevBuf::map_pieces).
struct ArenaKnobs { int trials = 0; size_t piece = (size_t)512 << 20; };
static const ArenaKnobs &arena_knobs() {
    static const ArenaKnobs k = [] {
        ArenaKnobs v;
        if (const char *e = getenv("REGTOOLS_AMD_ARENA")) {
            long long a = -1, b = -1;
            const int n = sscanf(e, "%lld,%lld", &a, &b);
            if (n >= 1 && a >= 0) v.trials = (int)std::min<long long>(a, 7);
            if (n >= 2 && b >= 0) v.piece = b == 0 ? 0 : (size_t)std::min<long long>(std::max<long long>(b, 2), 16384) << 20;
        }
        return v;
    }();
    return k;
}
struct DevBuf {
    static constexpr size_t kFront = 256;
    void *p = nullptr; size_t cap = 0;
    size_t piece = 0;               // asked for by the owner (the arena): memory created in pieces of this size and mapped side by side, see map_pieces (0: one hipMalloc block)
    size_t mapped = 0;              // bytes of the reserved address range the pieces are mapped into (0: a hipMalloc block)
    std::vector<size_t> piece_len;  // the mappings inside that range, in address order (each is unmapped on its own)
    // The arena's form (round 5).  The DEFLATE launch writes 169,000 streams 64 KB apart at once, and what it costs depends on the memory under them: 13.9-15.9 ms
    // into one hipMalloc block of 11 GB, 12.3-12.6 ms into the same bytes created as pieces of 1 GiB (hipMemCreate) and mapped side by side into one reserved address range --
    // whatever the order of the pieces (profiles/r05_inflate_arena_pieces.txt: forty pieces, 110 subsets, 12.30-12.37 ms).  Pieces of 2 MiB: 16.0 ms; 32 MiB: 12.8-13.7;
    // 256 MiB: 12.4-13.7.  A runtime that refuses any of the calls leaves the buffer to hipMalloc.
    static void unmap_range(void *base, const std::vector<size_t> &lens, size_t reserved) {
        static const bool trace = getenv("REGTOOLS_AMD_TRACE") != nullptr;
        size_t at = 0;
        for (size_t n : lens) {                                     // (one mapping at a time: the form HIP's own tests use; a refusal would leak the piece silently)
            const hipError_t e = hipMemUnmap((uint8_t *)base + at, n);
            if (e != hipSuccess && trace) fprintf(stderr, "[rgx trace] hipMemUnmap of %zu bytes at +%zu: %s\n", n, at, hipGetErrorString(e));
            if (e != hipSuccess) (void)hipGetLastError();
            at += n;
        }
        const hipError_t e = hipMemAddressFree(base, reserved);
        if (e != hipSuccess) { if (trace) fprintf(stderr, "[rgx trace] hipMemAddressFree of %zu bytes: %s\n", reserved, hipGetErrorString(e)); (void)hipGetLastError(); }
    }
    hipError_t map_pieces(size_t bytes, void **out) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
        size_t gran = 0;                                            // what this runtime wants sizes and addresses to be multiples of (2 MiB on ROCm 7.2)
        if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || !gran) { (void)hipGetLastError(); gran = (size_t)2 << 20; }
        const size_t total = (bytes + gran - 1) / gran * gran, each = std::max(gran, piece / gran * gran);
        void *base = nullptr;
        if ((e = hipMemAddressReserve(&base, total, 0, nullptr, 0)) != hipSuccess) return e;
        std::vector<size_t> lens;
        size_t done = 0;
        while (done < total) {
            const size_t n = std::min(each, total - done);
            hipMemGenericAllocationHandle_t h;
            if ((e = hipMemCreate(&h, n, &prop, 0)) != hipSuccess) break;
            e = hipMemMap((uint8_t *)base + done, n, 0, h, 0);
            (void)hipMemRelease(h);                                 // (the mapping keeps the memory; an unmapped, released piece is gone)
            if (e != hipSuccess) break;
            lens.push_back(n);
            done += n;
        }
        if (e == hipSuccess) {
            hipMemAccessDesc ad = {}; ad.location.type = hipMemLocationTypeDevice; ad.location.id = dev; ad.flags = hipMemAccessFlagsProtReadWrite;
            e = hipMemSetAccess(base, total, &ad, 1);
        }
        if (e != hipSuccess) { unmap_range(base, lens, total); return e; }
        *out = base; mapped = total; piece_len.swap(lens);
        return hipSuccess;
    }
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        static const bool trace = getenv("REGTOOLS_AMD_TRACE") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        release();
        // (growth slack, so that a context which sees files of slowly growing size does not reallocate per call; a one-shot process gets what it asks
        //  for: device memory is cleared when it is handed out, 15.6 GB cost such a process 60-360 ms -- 1.7 GB of that was slack)
        static const bool no_slack = [] { const char *e = getenv("REGTOOLS_AMD_ONE_SHOT"); return e && strcmp(e, "0") != 0; }();      // (first use: after main() said so)
        size_t want = bytes + (no_slack ? 0 : bytes / 8) + 256;
        void *raw = nullptr;
        hipError_t e = hipErrorNotSupported;
        if (piece && want + kFront >= piece) { e = map_pieces(want + kFront, &raw); if (e != hipSuccess) { (void)hipGetLastError(); raw = nullptr; mapped = 0; piece_len.clear(); } }
        if (e != hipSuccess) e = hipMalloc(&raw, want + kFront);
        if (e == hipSuccess) { p = (uint8_t *)raw + kFront; cap = want; }
        if (trace) {                                                 // (shards of a multi-device call grow their buffers on their own threads)
            static std::mutex mu; std::lock_guard<std::mutex> lock(mu);
            g_alloc_stats.ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); ++g_alloc_stats.calls; g_alloc_stats.bytes += want;
        }
        return e;
    }
    void release() {
        if (p && mapped) { (void)hipDeviceSynchronize();    /* (what hipFree does by itself: nothing in flight may still touch the range) */
                           unmap_range((uint8_t *)p - kFront, piece_len, mapped); }
        else if (p) (void)hipFree((uint8_t *)p - kFront);
        p = nullptr; cap = 0; mapped = 0; piece_len.clear();
    }
    template <class T> T *as() const { return (T *)p; }
};

}  // namespace

constexpr int kSideStreams = 2;
// REGTOOLS_AMD_OVERLAP="min_bytes[,chunks[,early_min_members]]" (tests): the thresholds of the overlapped upload, so that files of test size take the path
// the 533 MB bench file takes -- from how many bytes a host buffer goes up in chunks behind ONE arrival-gated inflate launch (default 8 MiB), in how many
// chunks (16), and from how many members an early-tail part is worth cutting (4096; giving it also lifts the part's minimum of 1,024 segments).
struct OverlapKnobs { size_t min_bytes = (size_t)8 << 20; unsigned chunks = 16; uint32_t early_min = 4096; bool early_small = false; };
static const OverlapKnobs &overlap_knobs() {
    static const OverlapKnobs k = [] {
        OverlapKnobs v;
        if (const char *e = getenv("REGTOOLS_AMD_OVERLAP")) {
            long long a = -1, b = -1, c2 = -1;
            const int n = sscanf(e, "%lld,%lld,%lld", &a, &b, &c2);
            if (n >= 1 && a >= 0) v.min_bytes = (size_t)a;
            if (n >= 2 && b >= 2) v.chunks = (unsigned)std::min<long long>(b, 64);
            if (n >= 3 && c2 >= 1) { v.early_min = (uint32_t)c2; v.early_small = true; }
        }
        return v;
    }();
    return k;
}
// Two things the contexts of one pipeline (pipeline.cpp) take in turns, first come first served:
//   wire -- the host link: a call's upload starts when the call before it has ITS file on the device (two uploads at once halve the link between them);
//   chip -- the DEFLATE launch: a call's launch is enqueued when the launch before it has finished.  One launch is 2,647 of the chip's 3,072 wave slots and all
//           of its LDS; two at once leave the first file's framing / decode / sort kernels nowhere to run until the second file's waves drain (measured: both
//           files of a pair end together, 43 ms for the two).  One after the other, a file's tail runs in the slots its successor's launch leaves free.
struct Turn {
    std::mutex mu; std::condition_variable cv; uint64_t next = 0, serving = 0;
};
struct LinkTurn { Turn wire, chip; };
// one context's hold on a turn: taken by the call's host thread, given back from a host function on the stream when the copy / the launch is over
// (or by the end of the call, whichever comes first).  Lives in the context: a stream may still owe the give when a failed call has returned.
struct TurnHold {
    Turn *t = nullptr; std::atomic<bool> held{false};
    void take(Turn *x) {
        if (!x || held.load()) return;
        std::unique_lock<std::mutex> lk(x->mu);
        const uint64_t mine = x->next++;
        x->cv.wait(lk, [&] { return x->serving == mine; });
        t = x; held.store(true);
    }
    void give() { if (held.exchange(false)) { { std::lock_guard<std::mutex> lk(t->mu); ++t->serving; } t->cv.notify_all(); } }
};

struct rgx_ctx {
    int device = 0;
    LinkTurn *link = nullptr;                          // (not owned; nullptr = a context on its own)
    TurnHold wire_hold, chip_hold;
    hipStream_t stream = nullptr;
    // host input (rgx_extract_mem / rgx_extract): the file goes up in chunks on its own stream while the members that have arrived are
    // being inflated on the side streams (prepare_events)
    hipStream_t copy_stream = nullptr, side[kSideStreams] = {};
    bool gate_distrust = false, early_distrust = false;   // a gated launch whose verdict was not clean / an early-tail wait that timed out on this context: not tried again (a stream layout in which
                                                       // the waiting waves and the kernels that release them share a hardware queue would cost every call its 2 s time-out)
    bool walk_strict = false;             // set around the re-run of a call whose block_size-only framing met a record bam_read1 refuses (prepare_events)
    bool one_shot = false;                             // REGTOOLS_AMD_ONE_SHOT at creation: no streams besides `stream` (ensure_upload_streams)
    std::vector<hipEvent_t> chunk_ev;
    uint32_t gate_epoch = 0;                           // arrival gate of the overlapped upload (kernels.h InflateGate): this context's call counter
    hipEvent_t ev_ready = nullptr, ev_side[kSideStreams] = {}, ev_packed = nullptr;
    hipEvent_t ev[8] = {};
    // Arena placement (round 5): the DEFLATE launch's time depends on where the arena's pages lie -- 12.8 / 13.9 / 15.0 ms for the same launch into
    // ten arenas of one process, stable per arena -- so a context that is not one-shot tries a few on its first large call and keeps the fastest.
    uint64_t arena_calibrated_bytes = 0;                 // the size the kept arena was chosen at (0 = not yet)
    DevBuf *arena_retired = nullptr;                     // the arena a call's data lies in after it lost to a challenger: released by the next call
    hipEvent_t ev_trial[2] = {};
    float arena_trial_ms[8] = {}; int arena_trials = 0;  // (statistics: the candidates' times of the last calibration, [0] = the arena the call ran on)
    hipEvent_t ev_launch[2] = {}; bool launch_timed = false;   // around the call's whole-range DEFLATE launch, on the stream it runs on (host input: the arrival-gated launch, which spans the upload)
    std::map<std::string, DevBuf> bufs;
    void *pinned = nullptr; size_t pinned_cap = 0;     // small pinned staging for scalar readbacks
    std::vector<Member> hm_scratch;
    void *pinned_members = nullptr; size_t pinned_members_cap = 0;      // the host scan's member list: kernels read it in place (grow-only)
    void *pinned_rows = nullptr; size_t pinned_rows_cap = 0;
    uint64_t last_rows = 0, last_records = 0, last_events = 0, last_bytes = 0; bool last_rows_valid = false;      // rows of the last rgx_extract* call, still in the "rows_out" block in HBM   // grow-only pinned staging for whole result tables (device merge)
    // HIP-event timing of single kernels inside a stage (the interval kernels of `identify`: roofline figures need the kernel's own
    // duration, not the stage's wall time): event pairs wait in kpend until the call's end, kms[slot] accumulates
    struct KPend { hipEvent_t a, b; int slot; };
    std::vector<KPend> kpend; std::vector<hipEvent_t> kfree; double kms[3] = {0, 0, 0};
    uint64_t tables_made = 0;
    std::vector<uint32_t> rank_stage;                  // host copy of a group-rank table while its upload is in flight
    std::string fasta_path;                            // FASTA currently resident in the "fasta" buffer
    rgx::Fasta *fasta = nullptr;
    // The genome the output stages look splice sites up in (host_fasta below): its mapping stays with the context from call to call
    rgx::Fasta *host_fasta = nullptr; std::string host_fasta_path; uint64_t host_fasta_key[4] = {0, 0, 0, 0};
    DevBuf &buf(const char *name) { return bufs[name]; }
};

// The FASTA at `path`, mapped (cse_host.h).  A context keeps the last one: what a call's 10^5 two-base lookups cost is mostly page-table work --
// faulting the pages in (sixteen per fault) and, dearer, taking two million entries down again when the mapping goes (11 ms of config 4's
// `identify`) -- and a caller that runs one sample after the other against the same genome pays both once.  The file is recognised by device,
// inode, size and modification time; anything else is a new file.  nullptr = it cannot be opened.
static rgx::Fasta *host_fasta(rgx_ctx *c, const char *path) {
    struct stat st;
    if (!path || stat(path, &st) != 0) return nullptr;
    const uint64_t key[4] = {(uint64_t)st.st_dev, (uint64_t)st.st_ino, (uint64_t)st.st_size, (uint64_t)st.st_mtim.tv_sec * 1000000000ull + (uint64_t)st.st_mtim.tv_nsec};
    if (c->host_fasta && c->host_fasta_path == path && !memcmp(key, c->host_fasta_key, sizeof key)) return c->host_fasta;
    delete c->host_fasta; c->host_fasta = nullptr;
    rgx::Fasta *f = new rgx::Fasta();
    if (!f->load(path)) { delete f; return nullptr; }
    c->host_fasta = f; c->host_fasta_path = path; memcpy(c->host_fasta_key, key, sizeof key);
    return f;
}

static void ktime_begin(rgx_ctx *c, int slot) {
    hipEvent_t e[2];
    for (auto &x : e) { if (!c->kfree.empty()) { x = c->kfree.back(); c->kfree.pop_back(); } else if (hipEventCreate(&x) != hipSuccess) return; }
    (void)hipEventRecord(e[0], c->stream);
    c->kpend.push_back({e[0], e[1], slot});
}
static void ktime_end(rgx_ctx *c) { if (!c->kpend.empty()) (void)hipEventRecord(c->kpend.back().b, c->stream); }
static void ktime_collect(rgx_ctx *c) {
    for (auto &k : c->kpend) {
        float ms = 0;
        if (hipEventSynchronize(k.b) == hipSuccess && hipEventElapsedTime(&ms, k.a, k.b) == hipSuccess) c->kms[k.slot] += ms;
        c->kfree.push_back(k.a); c->kfree.push_back(k.b);
    }
    c->kpend.clear();
}

extern "C" const char *rgx_version(void) { return "regtools_amd 0.1 gfx950"; }

extern "C" void rgx_extract_params_default(rgx_extract_params *p) {
    memset(p, 0, sizeof *p);
    p->region = "."; p->strandness = -1; p->strand_tag[0] = 'X'; p->strand_tag[1] = 'S';
    p->min_anchor = 8; p->min_intron = 70; p->max_intron = 500000; p->fasta_path = nullptr; p->shard = 0; p->n_shards = 1;
    p->barcodes = 0; p->barcode_tag[0] = 'C'; p->barcode_tag[1] = 'B';
}

extern "C" int rgx_ctx_create(int device, rgx_ctx **out, char *err, size_t errlen) {
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(err, errlen, RGX_ERR_NO_DEVICE, "regtools_amd: no HIP device visible; this library has no CPU fallback\n");
    if (device < 0 || device >= n) return fail(err, errlen, RGX_ERR_NO_DEVICE, "regtools_amd: device %d out of range (%d visible)\n", device, n);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (!strstr(prop.gcnArchName, "gfx950"))
        return fail(err, errlen, RGX_ERR_NO_DEVICE, "regtools_amd: device %d is %s; the kernels are built for gfx950 only\n", device, prop.gcnArchName);
    rgx_ctx *c = new rgx_ctx();
    c->device = device;
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    { const char *e = getenv("REGTOOLS_AMD_ONE_SHOT"); c->one_shot = e && strcmp(e, "0") != 0; }
    c->buf("arena").piece = arena_knobs().piece;                     // (DevBuf::map_pieces: what the DEFLATE launch writes into)
    for (auto &e : c->ev) HIP_TRY(hipEventCreate(&e));
    for (auto &e : c->ev_launch) HIP_TRY(hipEventCreate(&e));
    for (auto &e : c->ev_trial) HIP_TRY(hipEventCreate(&e));
    HIP_TRY(hipHostMalloc(&c->pinned, 4096, hipHostMallocDefault));
    c->pinned_cap = 4096;
    *out = c;
    return RGX_OK;
}

// the copy stream and the side streams of the overlapped upload: made when a call first takes that path (a one-shot process that reads a
// small file never pays for them; eight stream creations are ~100 ms of a cold start)
static hipError_t ensure_upload_streams(rgx_ctx *c) {
    if (c->copy_stream || c->ev_ready) return hipSuccess;
    // A process that makes one call (bin/regtools-amd: REGTOOLS_AMD_ONE_SHOT, set by its main()) does without streams of its own: creating
    // the copy stream and two side streams costs 24-30 ms (8-10 ms per hardware queue), the overlap they buy -- upload under inflate, three inflate
    // launches side by side -- 5 ms of a call: the file goes up in one piece on the context's stream, one inflate launch follows it.
    if (c->one_shot) return hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming);
    hipError_t e = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking);
    if (e != hipSuccess) return e;
    // (the runtime maps the streams of one priority onto four hardware queues)
    // two side streams + the pipeline's own (measured and not kept, round 4: a third one of the greatest priority for the early tail's launch, +0.8 ms;
    // four to seven pieces on streams of other priorities, no gain: DESIGN.md 4.4)
    for (int k = 0; k < kSideStreams; ++k) {
        if ((e = hipStreamCreateWithPriority(&c->side[k], hipStreamNonBlocking, 0)) != hipSuccess) return e;
        if ((e = hipEventCreateWithFlags(&c->ev_side[k], hipEventDisableTiming)) != hipSuccess) return e;
    }
    return hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming);
}

// (multi.cpp: a context made for a device that a device list names a second time -- shards taking turns on one GPU, a test configuration -- does without the trials:
//  several contexts of one device would each hold a second arena at the same time)
void rgx_ctx_no_arena_trials(rgx_ctx *c) { if (c) c->arena_calibrated_bytes = UINT64_MAX; }
// pipeline.cpp: the contexts of one pipeline take the host link in turns
void *rgx_link_turn_create() { return new LinkTurn; }
void rgx_link_turn_destroy(void *l) { delete (LinkTurn *)l; }
void rgx_ctx_set_link(rgx_ctx *c, void *l) { if (c) c->link = (LinkTurn *)l; }

extern "C" int rgx_ctx_arena_trials(const rgx_ctx *c, float *ms, int cap) {
    if (!c) return 0;
    for (int k = 0; k < c->arena_trials && k < cap; ++k) ms[k] = c->arena_trial_ms[k];
    return c->arena_trials;
}

extern "C" void rgx_ctx_destroy(rgx_ctx *c) {
    if (!c) return;
    Reaper::get().drain();                                  // (deferred teardown of finished calls may still hold memory of this device)
    (void)hipSetDevice(c->device);
    for (auto &kv : c->bufs) kv.second.release();
    if (c->arena_retired) { c->arena_retired->release(); delete c->arena_retired; }
    for (auto &e : c->ev) if (e) (void)hipEventDestroy(e);
    for (auto &e : c->ev_launch) if (e) (void)hipEventDestroy(e);
    for (auto &e : c->ev_trial) if (e) (void)hipEventDestroy(e);
    for (auto &e : c->chunk_ev) if (e) (void)hipEventDestroy(e);
    ktime_collect(c);
    for (auto &e : c->kfree) (void)hipEventDestroy(e);
    for (auto &e : c->ev_side) if (e) (void)hipEventDestroy(e);
    if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
    if (c->ev_packed) (void)hipEventDestroy(c->ev_packed);
    for (auto &q : c->side) if (q) (void)hipStreamDestroy(q);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    delete c->fasta;
    delete c->host_fasta;
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->pinned_rows) (void)hipHostFree(c->pinned_rows);
    if (c->pinned_members) (void)hipHostFree(c->pinned_members);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// ---- table plumbing --------------------------------------------------------------------------------------
// Result tables: all row columns of a table live in ONE block, and released blocks are kept (a few, bounded) for the next table.  A
// pipeline that runs step after step -- bench.py, a multi-GPU job merging every step -- then writes its rows into pages that are already
// mapped instead of paying mmap + first-touch faults + munmap for ~50 bytes per row each time (measured: 9 of 14 ms of an 8-shard merge).
struct TableBox { rgx_junction_table t; void *block; size_t block_cap; bool pinned; };
struct CachedBlock { void *p; size_t cap; bool pinned; };
static std::mutex g_block_mu;
static std::vector<CachedBlock> g_blocks;                        // released blocks, at most kBlockCacheEntries / kBlockCacheBytes
static const size_t kBlockCacheEntries = 6, kBlockCacheBytes = (size_t)1 << 30;

// pinned = page-locked (hipHostMalloc): the device pipelines copy the finished columns straight into the block
static void *block_take(size_t need, size_t &cap, bool pinned) {
    {
        std::lock_guard<std::mutex> lk(g_block_mu);
        size_t best = g_blocks.size();
        for (size_t i = 0; i < g_blocks.size(); ++i)
            if (g_blocks[i].pinned == pinned && g_blocks[i].cap >= need && g_blocks[i].cap <= need * 2 + (1 << 20) &&
                (best == g_blocks.size() || g_blocks[i].cap < g_blocks[best].cap)) best = i;
        if (best != g_blocks.size()) { void *p = g_blocks[best].p; cap = g_blocks[best].cap; g_blocks.erase(g_blocks.begin() + (long)best); return p; }
    }
    cap = need;
    if (!pinned) return malloc(need);
    void *p = nullptr;
    if (hipHostMalloc(&p, need, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
static void block_give(void *p, size_t cap, bool pinned) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(g_block_mu);
        size_t held = 0;
        for (auto &b : g_blocks) held += b.cap;
        if (cap >= (1 << 16) && g_blocks.size() < kBlockCacheEntries && held + cap <= kBlockCacheBytes) { g_blocks.push_back(CachedBlock{p, cap, pinned}); return; }
    }
    if (pinned) (void)hipHostFree(p); else free(p);
}

// zero = the caller does not write every column of every row
static rgx_junction_table *table_alloc(const BamHeader &h, uint64_t n, bool zero = true, bool pinned = false) {
    TableBox *box = (TableBox *)calloc(1, sizeof *box);
    rgx_junction_table *t = &box->t;
    t->n_ref = (int32_t)h.names.size();
    t->ref_name = (char **)calloc(h.names.size() + 1, sizeof(char *));
    t->ref_len = (uint32_t *)calloc(h.names.size() + 1, sizeof(uint32_t));
    for (size_t i = 0; i < h.names.size(); ++i) { t->ref_name[i] = strdup(h.names[i].c_str()); t->ref_len[i] = h.lens[i]; }
    t->n = n;
    const size_t m = table_block_rows(n);                         // every column starts 16-byte aligned; the layout launch_rows_table writes
    const size_t need = table_block_bytes(n);
    box->pinned = pinned;
    box->block = block_take(need, box->block_cap, pinned);
    if (!box->block && pinned) { box->pinned = false; box->block = block_take(need, box->block_cap, false); }
    if (!box->block) {                                           // no memory for the rows: no table (callers report RGX_ERR_DEVICE / RGX_ERR_ARG)
        for (int32_t i = 0; i < t->n_ref; ++i) free(t->ref_name[i]);
        free(t->ref_name); free(t->ref_len); free(box);
        return nullptr;
    }
    if (zero) memset(box->block, 0, need);
    uint8_t *q = (uint8_t *)box->block;
    t->name_index = (uint64_t *)q; q += m * 8; t->first_seen = (uint64_t *)q; q += m * 8; t->last_seen = (uint64_t *)q; q += m * 8;
    t->tid = (int32_t *)q; q += m * 4; t->start = (uint32_t *)q; q += m * 4; t->end = (uint32_t *)q; q += m * 4;
    t->thick_start = (uint32_t *)q; q += m * 4; t->thick_end = (uint32_t *)q; q += m * 4; t->read_count = (uint32_t *)q; q += m * 4;
    t->strand = (char *)q; q += m; t->left_ok = q; q += m; t->right_ok = q;
    return t;
}

extern "C" void rgx_table_free(rgx_junction_table *t) {
    if (!t) return;
    TableBox *box = (TableBox *)t;                                   // t is the first member
    if (t->ref_name) for (int32_t i = 0; i < t->n_ref; ++i) free(t->ref_name[i]);
    free(t->ref_name); free(t->ref_len);
    block_give(box->block, box->block_cap, box->pinned);
    free(t->bc_row_begin); free(t->bc_count); free(t->bc_str_begin); free(t->bc_text); free(t->bc_insert_rank);
    free(box);
}

// compare_junctions (junctions_extractor.h:117-140): chrom string, thick_start, thick_end, name string
static void host_sort_rows(rgx_junction_table *t) {
    // compare_junctions (junctions_extractor.h:117-140): chrom string, thick_start, thick_end, name string.  Names are "JUNC%08d":
    // below 10^8 the string order is the numeric order; beyond, the longer decimal strings are compared as text.
    std::vector<uint32_t> crank((size_t)std::max(t->n_ref, 1), 0);
    {
        std::vector<int32_t> order((size_t)t->n_ref);
        for (int32_t i = 0; i < t->n_ref; ++i) order[(size_t)i] = i;
        std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return strcmp(t->ref_name[a], t->ref_name[b]) < 0; });
        uint32_t rk = 0;
        for (int32_t i = 0; i < t->n_ref; ++i) { if (i > 0 && strcmp(t->ref_name[order[(size_t)i]], t->ref_name[order[(size_t)i - 1]]) != 0) ++rk; crank[(size_t)order[(size_t)i]] = rk; }
    }
    auto name_less = [](uint64_t a, uint64_t b) {
        if (a < 100000000ull && b < 100000000ull) return a < b;
        char na[32], nb[32];
        snprintf(na, sizeof na, "%08llu", (unsigned long long)a); snprintf(nb, sizeof nb, "%08llu", (unsigned long long)b);
        return strcmp(na, nb) < 0;
    };
    std::vector<uint64_t> idx(t->n);
    for (uint64_t i = 0; i < t->n; ++i) idx[i] = i;
    std::sort(idx.begin(), idx.end(), [&](uint64_t a, uint64_t b) {
        const uint32_t ca = crank[(size_t)t->tid[a]], cb = crank[(size_t)t->tid[b]];
        if (ca != cb) return ca < cb;
        if (t->thick_start[a] != t->thick_start[b]) return t->thick_start[a] < t->thick_start[b];
        if (t->thick_end[a] != t->thick_end[b]) return t->thick_end[a] < t->thick_end[b];
        return name_less(t->name_index[a], t->name_index[b]);
    });
    auto permute = [&](auto *col) {
        typedef typename std::remove_reference<decltype(col[0])>::type T;
        std::vector<T> tmp(t->n);
        for (uint64_t i = 0; i < t->n; ++i) tmp[i] = col[idx[i]];
        memcpy(col, tmp.data(), sizeof(T) * t->n);
    };
    permute(t->tid); permute(t->start); permute(t->end); permute(t->thick_start); permute(t->thick_end); permute(t->read_count);
    permute(t->name_index); permute(t->strand); permute(t->left_ok); permute(t->right_ok); permute(t->first_seen); permute(t->last_seen);
}

// Junction::print (junctions_extractor.h:90-98) for rows [r0, r1): appended to `out`.  The name is copied as it is: the BAM header puts no
// limit on its length.
static void format_bed12_rows(const rgx_junction_table *t, int only_anchored, uint64_t r0, uint64_t r1, std::string &out) {
    char tail[256];                                        // everything behind the contig name: ten bounded numeric fields
    for (uint64_t i = r0; i < r1; ++i) {
        if (only_anchored && !(t->left_ok[i] && t->right_ok[i])) continue;
        const char *name = t->ref_name[t->tid[i]];
        const int n = snprintf(tail, sizeof tail, "\t%u\t%u\tJUNC%08llu\t%u\t%c\t%u\t%u\t255,0,0\t2\t%u,%u\t0,%u\n",
                               t->thick_start[i], t->thick_end[i], (unsigned long long)t->name_index[i], t->read_count[i], t->strand[i],
                               t->thick_start[i], t->thick_end[i], (uint32_t)(t->start[i] - t->thick_start[i]),
                               (uint32_t)(t->thick_end[i] - t->end[i]), (uint32_t)(t->end[i] - t->thick_start[i]));
        out.append(name); out.append(tail, (size_t)n);
    }
}

extern "C" size_t rgx_table_format_bed12(const rgx_junction_table *t, int only_anchored, char *buf, size_t cap) {
    // text work of ~170 ns per row on one core: ranges of rows on the host's cores (55 -> 6 ms for 300 k rows)
    const unsigned n_thr = t->n >= 20000 ? std::max(1u, std::min<unsigned>(usable_threads(16), (unsigned)(t->n / 8192))) : 1u;
    std::vector<std::string> part(n_thr);
    if (n_thr == 1) format_bed12_rows(t, only_anchored, 0, t->n, part[0]);
    else {
        std::vector<std::thread> pool;
        for (unsigned w = 0; w < n_thr; ++w)
            pool.emplace_back([&, w] { part[w].reserve((size_t)(t->n / n_thr + 1) * 96); format_bed12_rows(t, only_anchored, t->n * w / n_thr, t->n * (w + 1) / n_thr, part[w]); });
        for (auto &th : pool) th.join();
    }
    size_t need = 0;
    for (const std::string &q : part) need += q.size();
    if (buf && need <= cap) { size_t o = 0; for (const std::string &q : part) { memcpy(buf + o, q.data(), q.size()); o += q.size(); } }
    return need;
}

extern "C" size_t rgx_table_format_barcodes(const rgx_junction_table *t, int only_anchored, char *buf, size_t cap) {
    size_t need = 0;
    auto put = [&](const char *s, size_t n) { if (buf && need + n <= cap) memcpy(buf + need, s, n); need += n; };
    char num[32];
    for (uint64_t i = 0; i < t->n; ++i) {
        if (only_anchored && !(t->left_ok[i] && t->right_ok[i])) continue;
        const uint64_t b = t->bc_row_begin ? t->bc_row_begin[i] : 0, e = t->bc_row_begin ? t->bc_row_begin[i + 1] : 0;
        put(num, (size_t)snprintf(num, sizeof num, "%llu\t", (unsigned long long)(e - b)));     // Junction::print_barcodes (h:103-110)
        for (uint64_t k = b; k < e; ++k) {
            if (k != b) put(",", 1);
            put(t->bc_text + t->bc_str_begin[k], (size_t)(t->bc_str_begin[k + 1] - t->bc_str_begin[k]));
            put(num, (size_t)snprintf(num, sizeof num, ":%u", t->bc_count[k]));
        }
        put("\n", 1);
    }
    return need;
}

// ---- the pipeline ------------------------------------------------------------------------------------------------
// The member list of a file scanned ONCE by a caller that runs several shards of it (rgx_extract_multi): every shard then uploads only the
// header's members and its own byte range instead of the whole file, and none repeats the scan.
struct SharedMembers { const std::vector<Member> *members; uint64_t total_inflated; };

// Everything the later stages need from the front half of the pipeline (file bytes -> junction events in file order).
struct Prep {
    BamHeader hdr;
    const uint8_t *arena = nullptr;
    ReadSoA soa{};
    EventSoA ev{};
    uint32_t n_rec = 0, n_events = 0, n_range = 0;
    uint64_t n_iterated = 0, total = 0;
    uint32_t framing_sweeps = 0;
    bool stream_ended = false;     // the record stream stopped for a reason that ends iteration upstream (not: it reached this shard's upper cut)
    double t_begin = 0;
};

static int prepare_events(rgx_ctx *c, const uint8_t *d_bam_in, const uint8_t *h_bam, size_t bam_len, const uint8_t *bai, size_t bai_len,
                          const rgx_extract_params *p, bool want_read_span, Prep &P, char *err, size_t errlen, const uint32_t *d_true_sizes = nullptr,
                          bool allow_overlap = true, bool region_to_file_end = false, const SharedMembers *shared = nullptr);

constexpr int kGoOn = -1;                                  // a stage of EventsRun: nothing to report, the next one

// One call of the front half of the pipeline: file bytes -> junction events in file order.  The stages run in the order of
// run(); each returns kGoOn, or the call's result (an error, or the result of the call starting over on another path: a file whose footers
// lie, a record the lite walk must not accept, a shard whose range was not uploaded).  What one stage leaves for the next are the members below.
struct EventsRun {
    // -- the call's arguments (prepare_events) --
    rgx_ctx *c; const uint8_t *d_bam_in; const uint8_t *h_bam; size_t bam_len; const uint8_t *bai; size_t bai_len;
    const rgx_extract_params *p; bool want_read_span; Prep &P; char *err; size_t errlen; const uint32_t *d_true_sizes;
    bool allow_overlap, region_to_file_end; const SharedMembers *shared;
    // -- what the stages leave for one another --
    hipStream_t st = nullptr, copy_q = nullptr;               // the pipeline's stream; where the file's upload goes
    double t_begin = 0, t_last = 0; bool trace = false;
    void mark(const char *what) { if (trace) { double t = now_ms(); fprintf(stderr, "[rgx trace] %-28s +%8.3f ms  (at %8.3f)%s\n", what, t - t_last, t - t_begin, c->link ? (" clock " + std::to_string(fmod(t, 1e5))).c_str() : ""); t_last = t; } }
    // stage_upload: the index (parsed on a second host thread), the file on its way to HBM, the host's member scan
    BaiInfo bi; bool bai_ok = false;
    std::vector<uint8_t> index_image;                        // a .csi (or a compressed index) rewritten as a plain BAI image
    std::thread bai_thread;
    const uint8_t *d_bam = nullptr;
    struct Upload {
        std::thread th; std::atomic<uint32_t> recorded{0}; std::atomic<int> err{0};
        std::vector<size_t> end;                            // end[j] = bytes [lo, end[j]) resident once chunk event j has fired
        size_t lo = 0, hi = 0, hdr_hi = 0;                  // the byte range that goes up (a shard's, + the header's [0, hdr_hi); the whole file otherwise)
        hipStream_t copy_stream = nullptr;
        // every way out of this function: the helper has enqueued its copies and the DMA out of the caller's buffer is over (the caller
        // may free or reuse that buffer as soon as the call returns)
        ~Upload() { if (th.joinable()) th.join(); if (copy_stream) (void)hipStreamSynchronize(copy_stream); }
    } up;
    bool overlap = false, gated = false;
    size_t gate_chunk = 0;
    uint64_t hm_total = 0;
    uint32_t *d_sc = nullptr, *h_sc = nullptr;               // the call's scalars in HBM and their pinned host mirror
    // before anything looks at the file through the device (the fallbacks of damaged files): the bytes a shard did not send
    hipError_t complete_upload() {
        if (!h_bam || !(up.lo || (up.hi && up.hi < bam_len))) return hipSuccess;
        if (up.th.joinable()) up.th.join();
        hipError_t e = hipStreamSynchronize(copy_q);
        uint8_t *dst = c->buf("bam").as<uint8_t>();
        if (e == hipSuccess && up.lo > up.hdr_hi) e = hipMemcpy(dst + up.hdr_hi, h_bam + up.hdr_hi, up.lo - up.hdr_hi, hipMemcpyHostToDevice);
        if (e == hipSuccess && up.hi < bam_len) e = hipMemcpy(dst + up.hi, h_bam + up.hi, bam_len - up.hi, hipMemcpyHostToDevice);
        up.lo = 0; up.hi = bam_len; up.hdr_hi = 0;
        return e;
    }
    // stage_members: the member list (device discovery, or the host scan's), the record stream's start, cuts and chunks
    uint32_t n_cand = 0;
    uint64_t *cand = nullptr;
    uint32_t *nx[2] = {nullptr, nullptr}, *c_isize = nullptr, *c_reach = nullptr, *c_rank = nullptr, *c_isz2 = nullptr, *c_tmp = nullptr;
    Member *d_members = nullptr; hipMemcpyKind from_members = hipMemcpyDeviceToHost;
    // the members = the candidates that chain up from offset 0 (and, second try below, from the offset a seek lands on)
    void chain(uint64_t root2) {
        launch_member_link(d_bam, bam_len, cand, n_cand, nx[0], c_isize, c_reach, root2, st);
        int cur = 0;
        for (uint32_t span = 1; span < n_cand; span <<= 1) { launch_member_jump(n_cand, nx[cur], nx[cur ^ 1], c_reach, st); cur ^= 1; }
        launch_member_jump(n_cand, nx[cur], nx[cur ^ 1], c_reach, st);
        launch_scan_u32(c_reach, c_rank, n_cand, d_sc + 17, c_tmp, st);
        launch_member_compact(d_bam, bam_len, cand, c_isize, c_reach, c_rank, n_cand, d_members, c_isz2, st);
        if (d_true_sizes) launch_member_fix(d_members, c_isz2, n_cand, d_sc + 17, d_true_sizes, st);     // second run: lengths from the probe, not the footers
        launch_member_upos(d_members, c_isz2, d_sc + 17, (uint64_t *)(d_sc + 20), st);
    }
    bool whole = false, seek = false, chunked = false, geom_chunked_hint = false, empty_stream = false;
    uint64_t seek_voff = 0, cut_lo = 0, cut_hi = UINT64_MAX, total_all = 0, q_upos[3] = {0, 0, 0};
    std::vector<VChunk> chunks;
    uint32_t n_members_all = 0, first_member = 0, stop = 0;
    // stage_range_and_inflate: this call's member range, its arena, the launch (or launches) that fill it
    uint32_t m_lo = 0, m_hi = 0, n_range = 0;
    uint64_t upos_lo = 0, total = 0;
    uint8_t *d_bad = nullptr;
    // one member of the list (the host scan's list is host memory; the device's is read on the pipeline's stream -- never through the null stream, which
    // would wait for whatever any other stream of the process has in flight)
    hipError_t member_at(uint32_t k, Member &m) {
        if (from_members == hipMemcpyHostToHost) { memcpy(&m, d_members + k, sizeof m); return hipSuccess; }
        hipError_t e = hipMemcpyAsync(&m, d_members + k, sizeof m, from_members, st);
        return e == hipSuccess ? hipStreamSynchronize(st) : e;
    }
    hipError_t upos_of(uint32_t k, uint64_t &out_v) {
        if (k >= n_members_all) { out_v = total_all; return hipSuccess; }
        Member m;
        hipError_t e = member_at(k, m);
        out_v = m.upos;
        return e;
    }
    struct EarlyPart { uint32_t members, waves; uint64_t upos; };       // a part ends in front of member `members` of the range = workgroup `waves` = arena offset `upos`
    std::vector<EarlyPart> early_parts;
    bool split_B = false; hipEvent_t split_ev = nullptr;     // early tail: the gated launch still runs on a side stream; whoever reads its part of the arena waits for it
    hipError_t join_B() {
        if (!split_B) return hipSuccess;
        split_B = false;
        return hipStreamWaitEvent(st, split_ev, 0);
    }
    // stage_footers_and_header / stage_bounds_and_chains
    bool spec = false; uint32_t mean_rec = 0;
    BamHeader hdr; int32_t n_ref = 0;
    uint64_t lim = 0, pos0 = 0; bool chain_ended = false;
    ExtractCfg cfg; SegGeom geom; uint32_t seg_bytes = 0;
    std::vector<SegChunk> seg_chunks;
    // stage_framing (+ early tail) / stage_decode / stage_emit
    const uint8_t *arena = nullptr; uint64_t span = 0; uint32_t n_seg = 0, n_rec = 0; bool lite_walk = false;
    uint64_t *seg_start[2] = {nullptr, nullptr}, *seg_exit[2] = {nullptr, nullptr};
    uint32_t *seg_cnt[2] = {nullptr, nullptr}, *seg_base = nullptr;
    uint32_t *seg_iter_e = nullptr, *seg_long_e = nullptr, *seg_long_base_e = nullptr;      // early tail: per-segment outputs of the decode that the second framing must not overwrite
    uint16_t *seg_cp = nullptr;
    int cur = 0;
    // One framing: the walk of segments [walk_from, n_s), then verification sweeps over [0, n_s) until the chain agrees.  Returns -1 to go on,
    // anything else is the call's result (a restart on another path has run, or an error).  `ended` = the chain ends inside [0, n_s).
    int frame(uint32_t n_s, uint32_t walk_from, bool &ended) {
        DevBuf &b_tmp = c->buf("tmp");
        launch_seg_walk(arena, geom, n_s, n_ref, seg_start[cur], seg_exit[cur], seg_cnt[cur], seg_cp, st, walk_from);
        // d_sc[10]: leftmost disagreeing segment, d_sc[11]: leftmost chain end, d_sc[3]: record total
        for (int iter = 0;; ++iter) {
            HIP_TRY(hipMemsetAsync(d_sc + 10, 0xff, 8, st));
            launch_seg_verify(arena, geom, n_s, seg_start[cur], seg_exit[cur], seg_cnt[cur], seg_start[cur ^ 1], seg_exit[cur ^ 1],
                              seg_cnt[cur ^ 1], d_sc + 10, seg_cp, st);
            cur ^= 1;
            launch_scan_u32(seg_cnt[cur], seg_base, n_s, d_sc + 3, b_tmp.as<uint32_t>(), st);
            HIP_TRY(hipMemcpyAsync(h_sc + 3, d_sc + 3, 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(h_sc + 10, d_sc + 10, 8, hipMemcpyDeviceToHost, st));
            if (spec && iter == 0) {
                HIP_TRY(hipMemcpyAsync(h_sc, d_sc, 8, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipMemcpyAsync(h_sc + kStatusEarly, d_sc + kStatusEarly, 8, hipMemcpyDeviceToHost, st));
            }
            HIP_TRY(hipStreamSynchronize(st));
            if (spec && iter == 0 && (h_sc[0] != 0xffffffffu || h_sc[kStatusEarly] != 0xffffffffu)) {
                // some member did not inflate to its footer's length: nothing enqueued since is worth anything
                mark("inflate verdict: not clean, starting over device-resident");
                if (gated) { c->gate_distrust = true; if (trace) fprintf(stderr, "[rgx trace] arrival gate: verdict not clean, this context no longer uses it\n"); }
                HIP_TRY(join_B());
                HIP_TRY(complete_upload());
                HIP_TRY(hipStreamSynchronize(copy_q));
                HIP_TRY(hipStreamSynchronize(st));
                const int rc2 = prepare_events(c, d_bam, nullptr, bam_len, bai, bai_len, p, want_read_span, P, err, errlen, nullptr, false, region_to_file_end);
                P.t_begin = t_begin;
                return rc2;
            }
            ++P.framing_sweeps;
            if (h_sc[11] != 0xffffffffu) ended = true;           // some segment's chain ends: an unreadable / cut-off record (sam.c:421-423)
            // the chain ends inside the exact prefix (or everything is exact): nothing starts after that segment -- with one chain the
            // end already spread to the right by itself; the chains of later chunks would not know
            if (h_sc[11] != 0xffffffffu && (h_sc[11] < h_sc[10] || (h_sc[10] == 0xffffffffu && geom.chunks))) {
                launch_seg_truncate(geom, n_s, h_sc[11], seg_start[cur], seg_exit[cur], seg_cnt[cur], st);
                launch_scan_u32(seg_cnt[cur], seg_base, n_s, d_sc + 3, b_tmp.as<uint32_t>(), st);
                HIP_TRY(hipMemcpyAsync(h_sc + 3, d_sc + 3, 4, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                break;
            }
            if (h_sc[10] == 0xffffffffu) break;
            if (iter > 1 << 20) return fail(err, errlen, RGX_ERR_FORMAT, "regtools_amd: record framing did not converge\n");
        }
        return -1;
    }
    uint32_t sA = 0;                                          // early tail: segments [0, sA) are framed, verified and decoded
    bool emit_parts_ok = false; uint32_t emit_parts = 0, emit_rows = 0; size_t ev_lay = 0;      // early tail: rows [0, emit_rows) have their events out, in emit_parts parts
    EventSoA ev_e;
    EventSoA ev_layout(uint8_t *q, size_t E) {
        EventSoA v; memset(&v, 0, sizeof v);
        v.tid = (uint32_t *)q; q += E * 4; v.start = (uint32_t *)q; q += E * 4; v.ilen_cls = (uint32_t *)q; q += E * 4;
        v.ts = (uint32_t *)q; q += E * 4; v.te = (uint32_t *)q; q += E * 4;
        if (want_read_span) { v.rpos = (uint32_t *)q; q += E * 4; v.rend = (uint32_t *)q; q += E * 4; }
        if (p->barcodes) { v.read = (uint32_t *)q; q += E * 4; }
        v.strand = q;
        return v;
    }
    size_t soa_cap = 0;                                       // rows the SoA columns are laid out for (early tail: an estimate made from the prefix)
    ReadSoA soa;
    uint32_t *ev_base = nullptr, *long_list = nullptr;
    hipError_t soa_layout(size_t R) {
        DevBuf &b_soa = c->buf("soa");
        hipError_t e_ = b_soa.ensure(R * (4 + 4 + 4 + 8 + 1 + 4 + 4 + 4 + (p->barcodes ? 8 : 0)) + 256);
        if (e_ != hipSuccess) return e_;
        uint8_t *q = b_soa.as<uint8_t>();
        soa.cig_off = (uint64_t *)q; q += R * 8;
        if (p->barcodes) { soa.rec_off = (uint64_t *)q; q += R * 8; }
        soa.tid = (int32_t *)q; q += R * 4; soa.pos = (int32_t *)q; q += R * 4; soa.flag_nc = (uint32_t *)q; q += R * 4;
        soa.n_ev = (uint32_t *)q; q += R * 4; ev_base = (uint32_t *)q; q += R * 4; long_list = (uint32_t *)q; q += R * 4;
        soa.strand = q;
        soa_cap = R;
        return hipSuccess;
    }
    uint32_t n_events = 0, n_long = 0; uint64_t n_iterated = 0;
    // every way out while the side stream's launch may still run (an error in the prefix's framing, say: the next call on this context must not meet
    // it) and while the index thread runs; `up` joins its helper and waits for the DMA out of the caller's buffer itself
    ~EventsRun() {
        if (split_B && split_ev) (void)hipEventSynchronize(split_ev);
        if (bai_thread.joinable()) bai_thread.join();
        if (c->link && !d_bam_in) { c->wire_hold.give(); c->chip_hold.give(); }      // (a call that ended early: the other contexts must not wait for it)
    }
    int run();
    int calibrate_arena();
    int stage_upload();
    int stage_members();
    int stage_range_and_inflate();
    int stage_footers_and_header();
    int stage_bounds_and_chains();
    int stage_framing();
    int stage_decode();
    int stage_emit();
};

int EventsRun::run() {
    if (!p || p->strandness < 0 || p->strandness > 3) return fail(err, errlen, RGX_ERR_ARG, "Please supply strandness mode with '-s' option!\n\n");
    if (p->strandness == 3 && !p->fasta_path) return fail(err, errlen, RGX_ERR_ARG, "Strandness mode 'intron-motif' requires a fasta file!\n\n");
    HIP_ENTER(c->device);
    st = c->stream;
    copy_q = c->copy_stream ? c->copy_stream : c->stream;     // where the file's upload goes (a one-shot context: its only stream)
    t_begin = now_ms();
    trace = getenv("REGTOOLS_AMD_TRACE") != nullptr;
    t_last = t_begin;

    if (bam_len < 28) return fail(err, errlen, RGX_ERR_OPEN, "%s", kMsgOpen);
    if (c->arena_retired) { c->arena_retired->release(); delete c->arena_retired; c->arena_retired = nullptr; }      // (the arena the last call's data lay in, after it lost its place)
    { const int rc = stage_upload(); if (rc != kGoOn) return rc; }
    { const int rc = stage_members(); if (rc != kGoOn) return rc; }
    { const int rc = stage_range_and_inflate(); if (rc != kGoOn) return rc; }
    { const int rc = stage_footers_and_header(); if (rc != kGoOn) return rc; }
    { const int rc = stage_bounds_and_chains(); if (rc != kGoOn) return rc; }
    { const int rc = stage_framing(); if (rc != kGoOn) return rc; }
    { const int rc = stage_decode(); if (rc != kGoOn) return rc; }
    const int rc_emit = stage_emit();
    if (rc_emit == RGX_OK) { const int rc = calibrate_arena(); if (rc != RGX_OK) return rc; }
    return rc_emit;
}

// Arena placement trials (rgx_ctx above) -- OPT-IN (REGTOOLS_AMD_ARENA=5) since round 6.  On a context's first call with an arena of 2 GiB and more
// (and again when a later one is a quarter larger), once the call's own work is enqueued: the same whole-range launch, plain, into the call's arena and into
// fresh allocations ONE AT A TIME (a warm-up launch, then the median of three timed with HIP events); a challenger that beats the incumbent's median by 1.5 %
// becomes the context's arena (rgx_ctx_arena_trials reports the times).  The call's data stays where it is -- when a challenger wins, the old arena is
// retired and released by the next call.  At most ONE arena's worth of extra memory is ever held, free memory is asked for again before every candidate, and
// anything that goes wrong inside a trial leaves the incumbent in place: the caller's result is complete before the first trial starts.
static int arena_challengers() { return arena_knobs().trials; }
int EventsRun::calibrate_arena() {
    if (!arena_challengers() || c->one_shot || d_true_sizes || chunked || split_B || P.stream_ended || !n_range || n_range <= 2048 || total < ((uint64_t)2 << 30)) return RGX_OK;
    if (c->arena_calibrated_bytes == UINT64_MAX || (c->arena_calibrated_bytes && total + 256 <= c->arena_calibrated_bytes + c->arena_calibrated_bytes / 4)) return RGX_OK;
    static std::mutex trial_mu[16];                            // (one calibration at a time per DEVICE: the arenas of different devices have nothing to do with one another)
    std::lock_guard<std::mutex> trial_lock(trial_mu[(unsigned)c->device % 16u]);
    if (!inflate_takes_coop(n_range) || h_sc[0] != 0xffffffffu) return RGX_OK;
    DevBuf &b_arena = c->buf("arena"), &b_lens = c->buf("inflate_scratch");
    if (!b_arena.p || b_arena.cap < total + 256) return RGX_OK;
    auto room_for_one = [&] {                                  // (a challenger AND what the call -- or a co-tenant of the device -- may still allocate)
        size_t free_b = 0, total_b = 0;
        return hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b >= b_arena.cap + b_arena.cap / 8 + ((size_t)16 << 30);
    };
    const int plan = inflate_plan_for(bam_len, total_all);
    uint32_t *d_dummy = d_sc + 100;                            // (the trial launches' verdicts: not looked at -- the call's own launch gave the verdict)
    auto time_into = [&](uint8_t *arena_p, float &ms) -> hipError_t {
        float t[3] = {0, 0, 0};
        for (int k = 0; k < 4; ++k) {                          // (the first launch into a fresh allocation also pays for its pages: not timed)
            hipError_t e = hipMemsetAsync(d_dummy, 0xff, 8, st);
            if (e != hipSuccess) return e;
            if ((e = hipEventRecord(c->ev_trial[0], st)) != hipSuccess) return e;
            launch_inflate(d_bam, d_members + m_lo, n_range, arena_p, upos_lo, b_lens.as<uint32_t>(), d_dummy, st, 0, 0, false, 0, nullptr, plan);
            if ((e = hipEventRecord(c->ev_trial[1], st)) != hipSuccess) return e;
            if ((e = hipEventSynchronize(c->ev_trial[1])) != hipSuccess) return e;
            if (k && (e = hipEventElapsedTime(&t[k - 1], c->ev_trial[0], c->ev_trial[1])) != hipSuccess) return e;
        }
        std::sort(t, t + 3);
        ms = t[1];
        return hipSuccess;
    };
    c->arena_trials = 0;
    float best_ms = 0;
    if (time_into(b_arena.as<uint8_t>(), best_ms) != hipSuccess) { (void)hipGetLastError(); return RGX_OK; }      // (the call's own arena: the same bytes written once more, behind everything that read them)
    c->arena_trial_ms[c->arena_trials++] = best_ms;
    DevBuf best;                                               // the fastest challenger so far (empty: the incumbent leads)
    for (int k = 0; k < arena_challengers(); ++k) {
        if (best.p) break;                                     // (a winner is kept at once: never two challengers' memory at a time)
        if (!room_for_one()) break;
        // (what makes one placement faster than another is not known -- DESIGN 5.5 -- so the challengers are not of one kind)
        static const size_t kLadder[] = {(size_t)1 << 30, (size_t)256 << 20, (size_t)512 << 20, (size_t)128 << 20, (size_t)1 << 30, (size_t)64 << 20, (size_t)512 << 20};     // (a hipMalloc block never won one: 15.6-16.3 ms beside 12.7-13.6)
        DevBuf cand; cand.piece = b_arena.piece ? kLadder[k % 7] : 0;
        if (cand.ensure(b_arena.cap) != hipSuccess) { (void)hipGetLastError(); break; }
        float ms = 0;
        if (time_into(cand.as<uint8_t>(), ms) != hipSuccess) { (void)hipGetLastError(); cand.release(); break; }
        if (c->arena_trials < 8) c->arena_trial_ms[c->arena_trials++] = ms;
        if (ms < best_ms * 0.985f) { best = cand; best_ms = ms; } else cand.release();
    }
    if (best.p) {
        // the call's data lies in the old arena and the caller may still read it (P.arena): it is retired, not released
        if (c->arena_retired) { c->arena_retired->release(); delete c->arena_retired; }
        c->arena_retired = new DevBuf(b_arena);
        b_arena = best;
        b_arena.piece = arena_knobs().piece;                   // (a later regrow is made of the configured pieces, not of the winner's ladder size)
    }
    c->arena_calibrated_bytes = b_arena.cap;
    if (trace) {
        fprintf(stderr, "[rgx trace] arena placement: call's arena %.3f ms", c->arena_trial_ms[0]);
        for (int k = 1; k < c->arena_trials; ++k) fprintf(stderr, ", %.3f", c->arena_trial_ms[k]);
        fprintf(stderr, " -> %s\n", best.p ? "a challenger kept" : "kept");
    }
    mark("arena placement trial");
    return RGX_OK;
}

int EventsRun::stage_upload() {
    // -- index: ~1 ms of host parsing for a 5 MB .bai, done on a second host thread while this one feeds the device the member scan --
    bai_thread = std::thread([this] {
        bai_ok = bai && normalize_index(bai, bai_len, index_image, bai, bai_len) && parse_bai(bai, bai_len, bi, /*collect_anchors=*/false);
    });
    // -- upload ----------------------------------------------------------------------------------------------------------
    // Host input: the file goes up in chunks on the copy stream from a helper thread (a pageable source makes hipMemcpyAsync block), while
    // this thread finds the members on the host (scan_members_parallel) -- the inflate of chunk k's members then runs while chunk k+1 is
    // still on the bus (SURVEY 8d times the path from file bytes in host memory).  A file the host scan does not vouch for waits for the
    // whole upload and takes the device's member discovery, as does device input.
    d_bam = d_bam_in;
    std::vector<Member> &hm = c->hm_scratch;                 // the host scan's member list (overlap only; the context keeps its pages: a fresh 4 MB is a thousand page faults per call)
    hm.clear();
    if (!d_bam) {
        DevBuf &b = c->buf("bam");
        HIP_TRY(b.ensure(bam_len + 64));
        d_bam = b.as<uint8_t>();
        mark("file buffer in HBM");
        if (c->link) { c->wire_hold.take(&c->link->wire); mark("the link is ours"); }
        const size_t overlap_min = overlap_knobs().min_bytes;
        if (allow_overlap && !d_true_sizes && bam_len >= overlap_min) {
            HIP_TRY(ensure_upload_streams(c));
            if (c->copy_stream) copy_q = c->copy_stream;
            mark("upload streams");
            // A shard of a file whose members the caller scanned: only the bytes this shard reads go up -- the header's members and the
            // range between its two cuts (the same cuts as below, from the index) -- N shards then move the file once, not N times.
            size_t up_lo = 0, up_hi = bam_len, hdr_hi = 0;
            if (shared && p->n_shards > 1 && p->shard >= 0 && p->shard < p->n_shards) {
                if (bai_thread.joinable()) bai_thread.join();
                BamHeader hh; size_t hb = 0;
                if (bai_ok && host_bam_header(h_bam, std::min<size_t>(bam_len, (size_t)8 << 20), hh, &hb)) {
                    const bool rest0 = p->region && !strcmp(p->region, "*"), whole0 = rest0 || !strcmp(p->region ? p->region : ".", ".");
                    uint64_t sv = 0;
                    if (rest0 && bi.have_nocoor) sv = bi.nocoor_voff; else if (whole0 && !rest0 && bi.have_start) sv = bi.start_voff;
                    uint64_t tgt[2], got[2];
                    for (int k = 0; k < 2; ++k) tgt[k] = std::max<uint64_t>((uint64_t)((double)bam_len * (p->shard + k) / p->n_shards) << 16, sv ? sv : 1);
                    bai_first_anchor_ge(bai, bai_len, tgt, 2, got);
                    if (p->shard > 0 && got[0] != UINT64_MAX) up_lo = std::min<size_t>(bam_len, (size_t)(got[0] >> 16));
                    else if (p->shard > 0) up_lo = bam_len;
                    if (p->shard + 1 < p->n_shards && got[1] != UINT64_MAX) up_hi = std::min<size_t>(bam_len, (size_t)(got[1] >> 16) + 2 * kBgzfMaxBlock + 64);
                    if (up_hi < up_lo) up_hi = up_lo;
                    // (the header's members, and at least the four the device-side header read starts with)
                    const std::vector<Member> &sm = *shared->members;
                    if (!sm.empty()) { const Member &m4 = sm[std::min<size_t>(sm.size(), 4) - 1]; hb = std::max<size_t>(hb, (size_t)m4.cpos + m4.clen + 8); }
                    hdr_hi = std::min(up_lo, hb + 64);
                    up_lo &= ~(size_t)4095;
                    if (up_lo < hdr_hi) { up_lo = 0; hdr_hi = 0; }
                }
            }
            // Round 4: ONE inflate launch for the whole range, its waves gated by the arrival of their upload chunk (kernels.h InflateGate):
            // the file goes up in kGateChunks equal chunks, a 4-byte copy of this call's epoch into the chunk's flag word queued right behind each.
            // The members of the early chunks start ~0.6 ms into the upload; only those of the last chunk pay the lane-serial floor behind it
            // (round 3: three launches, each ~10 ms for a third of the members, the last one started when the last third had arrived).
            // A one-shot context, or one whose gate once gave an unclean verdict: round 3's pieces.
            const unsigned gate_chunks = overlap_knobs().chunks;
            gated = !c->gate_distrust && !c->one_shot && up_hi - up_lo >= std::min(overlap_min, (size_t)8 << 20) && up_hi - up_lo >= 2 * 4096 * (size_t)gate_chunks;
            if (gated) {
                // ... for payloads whose inflate is of the upload's order (measured: bench payload 27.5 -> 26.5 ms, random bases + qualities 98.2 ->
                // 93.7); run-length payloads (long reads: 1 GB of file, 65 GB inflated, five rounds of waves) lose 5-6 ms of 158 to it and keep
                // the pieces.  The class comes from the file's first members (BSIZE / ISIZE of up to 64 of them: what the host scan will find).
                uint64_t cb = 0, ub = 0; size_t o = 0;
                for (int k = 0; k < 64 && o + 28 <= bam_len; ++k) {
                    if (!(h_bam[o] == 0x1f && h_bam[o + 1] == 0x8b && h_bam[o + 12] == 'B' && h_bam[o + 13] == 'C')) break;
                    const size_t bl = (size_t)(h_bam[o + 16] | h_bam[o + 17] << 8) + 1;
                    if (bl < 26 || o + bl > bam_len) break;
                    uint32_t isz; memcpy(&isz, h_bam + o + bl - 4, 4);
                    cb += bl; ub += isz; o += bl;
                }
                if (cb && !(inflate_plan_for(cb, ub) & 1)) gated = false;
            }
            // three pieces, each its own launch on its own hardware queue (two side streams + the pipeline's own; a launch takes ~8 ms however
            // small -- one lane per member).  Equal thirds measured best: 31.6 ms per step against 32.4-33.3 ms for pieces that shrink towards
            // the end, and 30.9-31.5 ms for four to six equal pieces on side streams of other priorities (= other queue pools), 32.6 for seven
            // (tools/lab/pieces.sh): the concurrent launches share the chip, finer pieces do not end sooner.
            std::vector<unsigned> cuts = {33, 67};
            if (c->one_shot) cuts.clear();                            // (one stream: pieces would only take turns on it)
            if (gated) {
                gate_chunk = (((up_hi - up_lo) + gate_chunks - 1) / gate_chunks + 4095) & ~(size_t)4095;
                for (size_t e = up_lo + gate_chunk; e < up_hi; e += gate_chunk) up.end.push_back(e);
                DevBuf &bg = c->buf("gate_flags");
                if (!bg.p) { HIP_TRY(bg.ensure(4 * 64)); HIP_TRY(hipMemset(bg.p, 0, 4 * 64)); c->gate_epoch = 0; }
                ++c->gate_epoch;
            } else
            for (unsigned pc : cuts) {
                const size_t e = (up_lo + (size_t)((double)(up_hi - up_lo) * pc / 100.0) + 4095) & ~(size_t)4095;
                if (e < up_hi && e > up_lo && (up.end.empty() || e > up.end.back())) up.end.push_back(e);
            }
            up.end.push_back(up_hi);
            up.lo = up_lo; up.hi = up_hi; up.hdr_hi = hdr_hi;
            while (c->chunk_ev.size() < up.end.size()) { hipEvent_t e; HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming)); c->chunk_ev.push_back(e); }
            uint8_t *dst = b.as<uint8_t>();
            up.copy_stream = copy_q;
            uint32_t *gate_flags = gated ? c->buf("gate_flags").as<uint32_t>() : nullptr;
            const uint32_t gate_epoch = c->gate_epoch;
            hipStream_t gate_q = c->side[0] ? c->side[0] : copy_q;
            up.th = std::thread([this, dst, hdr_hi, up_lo, gate_q, gate_flags, gate_epoch] {
                if (hipSetDevice(c->device) != hipSuccess) { up.err = 1; up.recorded = (uint32_t)up.end.size(); return; }
                if (hdr_hi && hipMemcpyAsync(dst, h_bam, hdr_hi, hipMemcpyHostToDevice, copy_q) != hipSuccess) up.err = 1;
                size_t o = up_lo;
                for (size_t j = 0; j < up.end.size(); ++j) {
                    if ((up.end[j] > o && hipMemcpyAsync(dst + o, h_bam + o, up.end[j] - o, hipMemcpyHostToDevice, copy_q) != hipSuccess) ||
                        hipEventRecord(c->chunk_ev[j], copy_q) != hipSuccess) up.err = 1;
                    // (the flag's one-lane kernel goes to a side stream behind the chunk's event: on the copy stream itself it sat between two
                    //  copies, ~30 us of an idle bus per chunk)
                    if (gate_flags) {
                        if (gate_q != copy_q && hipStreamWaitEvent(gate_q, c->chunk_ev[j], 0) != hipSuccess) up.err = 1;
                        launch_gate_set(gate_flags + j, gate_epoch, gate_q);
                        // (a refused launch is only in THIS thread's hipGetLastError: unread, the flag would never be set and every gated wave
                        //  would wait out its time-out)
                        if (hipGetLastError() != hipSuccess) up.err = 1;
                    }
                    o = up.end[j];
                    up.recorded.store((uint32_t)j + 1, std::memory_order_release);
                }
                // the file is on the device: the next call's upload may start.  A host function behind the last copy, not a wait in this thread -- a thread
                // blocked in hipEventSynchronize kept the call's own launches from being enqueued until the upload was over (round 6: the gated launch
                // went out 8.8 ms late).
                if (c->link && hipLaunchHostFunc(copy_q, [](void *h) { ((TurnHold *)h)->give(); }, &c->wire_hold) != hipSuccess) { (void)hipGetLastError(); c->wire_hold.give(); }
            });
            if (shared) { hm = *shared->members; hm_total = shared->total_inflated; overlap = !hm.empty(); }
            else overlap = scan_members_parallel(h_bam, bam_len, (int)usable_threads(24), hm, hm_total);
            mark("host member scan");
            if (!overlap) {       // not a file the host vouches for: everything on the device, after the last chunk
                up.th.join();
                if (up.err) return fail(err, errlen, RGX_ERR_DEVICE, "regtools_amd: upload failed\n");
                if (up_lo || up_hi < bam_len) {                 // (only a range went up: the rest before the device looks at the file)
                    if (up_lo > hdr_hi) HIP_TRY(hipMemcpyAsync(dst + hdr_hi, h_bam + hdr_hi, up_lo - hdr_hi, hipMemcpyHostToDevice, copy_q));
                    if (up_hi < bam_len) HIP_TRY(hipMemcpyAsync(dst + up_hi, h_bam + up_hi, bam_len - up_hi, hipMemcpyHostToDevice, copy_q));
                    HIP_TRY(hipEventRecord(c->chunk_ev[up.end.size() - 1], copy_q));
                }
                HIP_TRY(hipStreamWaitEvent(st, c->chunk_ev[up.end.size() - 1], 0));
            }
        } else HIP_TRY(hipMemcpyAsync(b.p, h_bam, bam_len, hipMemcpyHostToDevice, st));
    }
    DevBuf &b_scalars = c->buf("scalars");
    HIP_TRY(b_scalars.ensure(512));
    // u32 scalars: [0]=first bad member [1]=its status [2]=changed [3]=n_rec [4]=n_events [5]=n_long [6]=n_unique [8..9]=n_iterated(u64)
    //              [12..13]=header inflate status [16]=n_cand [17]=n_members [18]=stop [20..21]=total inflated (u64)
    //              [24..26]=q_index [32..37]=q_upos (u64 x3) [40..45]=q_coff (u64 x3)
    d_sc = b_scalars.as<uint32_t>();
    h_sc = (uint32_t *)c->pinned;
    HIP_TRY(hipMemsetAsync(d_sc, 0, 512, st));
    HIP_TRY(hipMemsetAsync(d_sc, 0xff, 4, st));
    HIP_TRY(hipMemsetAsync(d_sc + 12, 0xff, 4, st));
    HIP_TRY(hipMemsetAsync(d_sc + 18, 0xff, 4, st));
    HIP_TRY(hipMemsetAsync(d_sc + kStatusEarly, 0xff, 4, st));

    return kGoOn;
}

int EventsRun::stage_members() {
    // -- BGZF member discovery on the device (replaces the serial BSIZE walk, bgzf.c:421-546) -------------------------------
    std::vector<Member> &hm = c->hm_scratch;
    DevBuf &b_members = c->buf("members"), &b_disc = c->buf("discover");
    if (overlap) {
        // the member list came from the host scan: what the discovery kernels would have left in HBM
        // (in page-locked host memory, read in place by the kernels -- 24 bytes per member, once: an upload would queue behind the file's
        // chunks on the copy engine, measured 4 ms)
        n_cand = (uint32_t)hm.size();
        const size_t need = ((size_t)n_cand + 1) * sizeof(Member);
        if (need > c->pinned_members_cap) {
            if (c->pinned_members) (void)hipHostFree(c->pinned_members);
            c->pinned_members = nullptr; c->pinned_members_cap = 0;
            HIP_TRY(hipHostMalloc(&c->pinned_members, need + need / 4, hipHostMallocDefault));
            c->pinned_members_cap = need + need / 4;
        }
        memcpy(c->pinned_members, hm.data(), (size_t)n_cand * sizeof(Member));
        h_sc[17] = n_cand; memcpy(h_sc + 20, &hm_total, 8);
        HIP_TRY(hipMemcpyAsync(d_sc + 17, h_sc + 17, 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d_sc + 20, h_sc + 20, 8, hipMemcpyHostToDevice, st));
    } else {
    const uint32_t n_tiles = (uint32_t)((bam_len + kMagicTile - 1) / kMagicTile);
    HIP_TRY(b_disc.ensure((size_t)n_tiles * 4 + scan_tmp_words(n_tiles) * 4 + 256));
    uint32_t *tile_cnt = b_disc.as<uint32_t>(), *tile_tmp = tile_cnt + n_tiles;
    launch_magic_count(d_bam, bam_len, n_tiles, tile_cnt, st);
    launch_scan_u32(tile_cnt, tile_cnt, n_tiles, d_sc + 16, tile_tmp, st);
    HIP_TRY(hipMemcpyAsync(h_sc + 16, d_sc + 16, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    n_cand = h_sc[16];
    if (n_cand == 0) return fail(err, errlen, RGX_ERR_OPEN, "%s", kMsgOpen);
    DevBuf &b_cand = c->buf("cand");
    {
        const size_t N = n_cand;
        HIP_TRY(b_cand.ensure(N * 8 + N * 4 * 6 + scan_tmp_words(n_cand) * 4 + 256));
        HIP_TRY(b_members.ensure((N + 1) * sizeof(Member)));
    }
    cand = b_cand.as<uint64_t>();
    nx[0] = (uint32_t *)(cand + n_cand); nx[1] = (uint32_t *)(cand + n_cand) + n_cand;
    c_isize = nx[1] + n_cand; c_reach = c_isize + n_cand; c_rank = c_reach + n_cand; c_isz2 = c_rank + n_cand; c_tmp = c_isz2 + n_cand;
    launch_magic_fill(d_bam, bam_len, n_tiles, tile_cnt, cand, st);
    }
    d_members = overlap ? (Member *)c->pinned_members : b_members.as<Member>();
    from_members = overlap ? hipMemcpyHostToHost : hipMemcpyDeviceToHost;
    if (!overlap) chain(UINT64_MAX);
    if (bai_thread.joinable()) bai_thread.join();
    if (!bai_ok) return (void)hipStreamSynchronize(st), fail(err, errlen, RGX_ERR_INDEX, "%s", kMsgIndex);
    mark("parse_bai");
    // "." = every record from the first one on; "*" = every record behind the last reference's reads (hts_itr_querys, hts.c:1901-1904:
    // HTS_IDX_START / HTS_IDX_NOCOOR; both read to the end of the file without a predicate)
    const bool rest = p->region && !strcmp(p->region, "*");
    whole = rest || !strcmp(p->region ? p->region : ".", ".");
    // where the record stream starts (hts.c:1721-1741)
    seek = false; seek_voff = 0;
    if (rest) {
        if (bi.have_nocoor) { seek_voff = bi.nocoor_voff; seek = seek_voff != 0; }
        else if (!bi.n_no_coor) return (void)hipStreamSynchronize(st), fail(err, errlen, RGX_ERR_REGION, "%s", kMsgRegion);
    } else if (whole) {
        if (bi.have_start) { seek_voff = bi.start_voff; seek = seek_voff != 0; }
        else if (!bi.n_no_coor) return (void)hipStreamSynchronize(st), fail(err, errlen, RGX_ERR_REGION, "%s", kMsgRegion);
    }
    // shard cut points: virtual offsets the BAI lists (every chunk begin / linear-index entry is a record start), so
    // no shard ever guesses its first record.  A record belongs to the shard in which its first byte lies.
    cut_lo = seek ? seek_voff : 0; cut_hi = UINT64_MAX;          // 0 = "right after the header"
    if (p->n_shards > 1) {
        if (p->shard < 0 || p->shard >= p->n_shards) return (void)hipStreamSynchronize(st), fail(err, errlen, RGX_ERR_ARG, "regtools_amd: shard %d of %d\n", p->shard, p->n_shards);
        uint64_t tgt[2], got[2];
        for (int k = 0; k < 2; ++k) {
            const int g = p->shard + k;
            tgt[k] = std::max<uint64_t>((uint64_t)((double)bam_len * g / p->n_shards) << 16, seek ? seek_voff : 1);
        }
        bai_first_anchor_ge(bai, bai_len, tgt, 2, got);
        if (p->shard > 0) cut_lo = got[0];
        if (p->shard + 1 < p->n_shards) cut_hi = got[1];
        if (cut_hi < cut_lo) cut_hi = cut_lo;
        mark("shard cuts");
    }

    // region queries: the reference's iterator reads the CHUNKS the index lists for the region's bins, one seek each, and ends at the first
    // record it reads that lies on another contig or at / behind the region's end (hts_itr_query / hts_itr_next, hts.c:1733-1800, :1924-1965).
    // The chunk list is computed here, from the index; the members between the first chunk's begin and the last one's end are inflated,
    // every chunk becomes its own record chain (SegGeom) and k_decode_seg applies the end rule.  Needs the contig names before the
    // launch: the header is inflated on the host from the head of the file.  A header that cannot be read that way leaves the range
    // alone: the whole file is read and filtered by overlap (such a header is not readable upstream either).
    chunked = false;
    geom_chunked_hint = !whole;                    // (region queries keep the checked path: their chunk table wants the members' verdicts)
    if (!whole && p->region) {
        const size_t head_len = std::min<size_t>(bam_len, (size_t)8 << 20);
        std::vector<uint8_t> head_copy;
        const uint8_t *head = h_bam;
        if (!head) { head_copy.resize(head_len); HIP_TRY(hipMemcpy(head_copy.data(), d_bam, head_len, hipMemcpyDeviceToHost)); head = head_copy.data(); }
        BamHeader hh;
        int32_t tid = -1, beg = 0, end = 0;
        if (host_bam_header(head, head_len, hh) && parse_region(hh, p->region, tid, beg, end) && tid < bi.n_ref && end >= beg &&
            region_chunks(bai, bai_len, tid, beg, end, chunks)) {
            chunked = true;
            if (p->n_shards > 1) {
                // a region query over several shards: the iterator's chunk list is dealt out in order, in runs of about equal compressed size;
                // shard g follows its run as an iterator of its own, and the one that meets the record that ends the iteration says so
                // (stream_ended): the merge ignores the shards behind it, as it does behind damage
                std::vector<uint64_t> before(chunks.size() + 1, 0);
                for (size_t k = 0; k < chunks.size(); ++k) before[k + 1] = before[k] + std::max<uint64_t>(1, (chunks[k].v >> 16) - (chunks[k].u >> 16));
                const uint64_t W = std::max<uint64_t>(1, before[chunks.size()]);
                std::vector<VChunk> mine;
                for (size_t k = 0; k < chunks.size(); ++k)
                    if ((int)std::min<uint64_t>((uint64_t)p->n_shards - 1, before[k] * (uint64_t)p->n_shards / W) == p->shard) mine.push_back(chunks[k]);
                chunks.swap(mine);
                cut_lo = seek ? seek_voff : 0; cut_hi = UINT64_MAX;     // (the byte cuts above were for a whole-file read)
            }
            if (!chunks.empty()) {
                // like the iterator's bgzf_seek: reading starts at the first chunk whatever the state of the members in front of it
                uint64_t hi = 0;
                for (const VChunk &ch : chunks) hi = std::max(hi, ch.v);
                cut_lo = chunks.front().u; cut_hi = std::max(hi, cut_lo); seek = true; seek_voff = cut_lo;
            } else if (bi.have_start && bi.start_voff) { cut_lo = cut_hi = bi.start_voff; seek = true; seek_voff = cut_lo; }     // no bin of the region holds a record: nothing to inflate
        }
        mark("region chunks");
    }

    empty_stream = false;
    auto query = [&]() -> hipError_t {
        uint64_t q[3] = {seek ? (seek_voff >> 16) : 0, cut_lo >> 16, cut_hi == UINT64_MAX ? UINT64_MAX - 64 : (cut_hi >> 16)};
        if (overlap) {
            // the member list is on the host (scan_members_parallel): what k_member_query / k_member_stop would answer, without a round trip
            const uint32_t nm = (uint32_t)hm.size();
            uint64_t q_up[3];
            for (int k = 0; k < 3; ++k) {
                uint32_t lo_ = 0, hi_ = nm;
                while (lo_ < hi_) { const uint32_t mid = lo_ + (hi_ - lo_) / 2; if (hm[mid].cpos < q[k] + 18) lo_ = mid + 1; else hi_ = mid; }
                const bool hit = lo_ < nm && hm[lo_].cpos == q[k] + 18;
                h_sc[24 + k] = hit ? lo_ : nm; q_up[k] = hit ? hm[lo_].upos : ~0ull;
            }
            memcpy(h_sc + 32, q_up, sizeof q_up);
            uint32_t stop_ = 0xffffffffu;
            for (uint32_t i = h_sc[24]; i < nm; ++i) if (hm[i].isize == 0 || hm[i].isize > kBgzfMaxBlock) { stop_ = i; break; }
            h_sc[18] = stop_; h_sc[17] = nm; memcpy(h_sc + 20, &hm_total, 8);
            return hipSuccess;
        }
        memcpy(h_sc + 40, q, sizeof q);
        hipError_t e = hipMemcpyAsync(d_sc + 40, h_sc + 40, sizeof q, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return e;
        e = hipMemsetAsync(d_sc + 18, 0xff, 4, st);
        if (e != hipSuccess) return e;
        launch_member_query(d_members, d_sc + 17, (const uint64_t *)(d_sc + 40), 3, d_sc + 24, (uint64_t *)(d_sc + 32), st);
        // the stream ends at the first empty (or oversized = corrupt) member at/after the first one read (bgzf.c:548-578)
        launch_member_stop(d_members, n_cand, d_sc + 17, d_sc + 24, d_sc + 18, st);
        e = hipMemcpyAsync(h_sc, d_sc, 256, hipMemcpyDeviceToHost, st);
        if (e != hipSuccess) return e;
        return hipStreamSynchronize(st);
    };
    HIP_TRY(query());
    if (overlap && seek && (seek_voff >> 16) != 0 && h_sc[24] >= h_sc[17]) {
        // the index points at something that is no member of this (well-formed) file: the device's discovery decides what that means
        up.th.join();
        HIP_TRY(complete_upload());
        HIP_TRY(hipStreamSynchronize(copy_q));
        HIP_TRY(hipStreamSynchronize(st));
        const int rc2 = prepare_events(c, d_bam, nullptr, bam_len, bai, bai_len, p, want_read_span, P, err, errlen, nullptr, false, region_to_file_end);
        P.t_begin = t_begin;
        return rc2;
    }
    if (seek && (seek_voff >> 16) != 0 && h_sc[24] >= h_sc[17]) {
        // the seek target is no member of the chain from offset 0: something in front of it is broken.  bgzf_seek (hts_itr_next, hts.c:1935)
        // goes there regardless -- take it as a second chain root.  (Only damaged files get here.)
        chain(seek_voff >> 16);
        HIP_TRY(query());
        mark("second chain root");
        if (h_sc[24] >= h_sc[17]) {
            // there is no BGZF member at the seek target at all (a truncated file, an index that belongs to another file): the
            // reference's read after bgzf_seek fails and the iterator returns nothing.  Keep the head of the file for the header only.
            chain(UINT64_MAX);
            seek = false; cut_lo = 0; cut_hi = 1; empty_stream = true;
            HIP_TRY(query());
        }
    }
    n_members_all = h_sc[17];
    if (n_members_all == 0) return fail(err, errlen, RGX_ERR_OPEN, "%s", kMsgOpen);     // offset 0 is not a BGZF member
    memcpy(&total_all, h_sc + 20, 8);
    first_member = seek ? h_sc[24] : 0;                                    // == n_members_all when the seek target is no member
    stop = std::min(h_sc[18], n_members_all);
    memcpy(q_upos, h_sc + 32, sizeof q_upos);
    mark("member discovery (2 syncs)");

    return kGoOn;
}

int EventsRun::stage_range_and_inflate() {
    // -- member range of this call ---------------------------------------------------------------------------------------------
    std::vector<Member> &hm = c->hm_scratch;
    DevBuf &b_arena = c->buf("arena");
    m_lo = cut_lo ? h_sc[25] : 0;
    if (m_lo <= 4) m_lo = 0;        // keep the file head (BAM header) in the same launch: a lone lane needs milliseconds per member
    m_hi = stop;                                                  // exclusive
    if (cut_hi != UINT64_MAX) {
        const uint32_t mh = h_sc[26];
        const uint32_t hi_m = (mh < n_members_all && (cut_hi & 0xffff)) ? mh + 1 : mh;
        // a region's chunks: each is a seek of its own, so an empty member between two of them ends nothing (the chunks' own limits do
        // that, below); two members more than the index asks for, for the records of a stale index that run past their chunk's end
        if (chunked) m_hi = region_to_file_end ? n_members_all : (uint32_t)std::min<uint64_t>(n_members_all, (uint64_t)hi_m + 2);
        else m_hi = std::min(stop, hi_m);
    }
    if (m_lo > m_hi) m_lo = m_hi;
    if (overlap && (up.lo || up.hi < bam_len) && m_hi > m_lo) {
        // only a byte range of the file went up (a shard of a shared scan): it must hold every member of this call's range
        const uint64_t need_lo = hm[m_lo].cpos - 18, need_hi = hm[m_hi - 1].cpos + hm[m_hi - 1].clen + 16;
        if (need_lo < up.lo || need_hi > up.hi) {
            mark("shard range does not cover its members: whole-file upload");
            up.th.join();
            HIP_TRY(hipStreamSynchronize(copy_q));
            HIP_TRY(hipStreamSynchronize(st));
            const int rc2 = prepare_events(c, d_bam_in, h_bam, bam_len, bai, bai_len, p, want_read_span, P, err, errlen, d_true_sizes, allow_overlap, region_to_file_end, nullptr);
            P.t_begin = t_begin;
            return rc2;
        }
    }
    // arena offsets of the range ends
    uint64_t upos_hi = 0;
    upos_lo = 0;
    HIP_TRY(upos_of(m_lo, upos_lo));
    HIP_TRY(upos_of(m_hi, upos_hi));
    total = upos_hi - upos_lo;
    n_range = m_hi - m_lo;
    HIP_TRY(b_arena.ensure(total + 256));
    HIP_TRY(hipEventRecord(c->ev[0], st));
    DevBuf &b_lens = c->buf("inflate_scratch");
    HIP_TRY(b_lens.ensure(inflate_scratch_bytes(std::max<uint32_t>(n_range, 64))));
    // with a seek, the members in front of its target are only inflated for the header's sake (same launch): their failures end nothing
    const uint32_t ignore_below = (seek && first_member < n_members_all && first_member > m_lo) ? first_member - m_lo : 0;
    d_bad = nullptr;                                 // region queries: which members of the range did not inflate (every chunk has its own end of stream)
    if (chunked && !chunks.empty() && n_range) {
        DevBuf &b_bad = c->buf("bad_members");
        HIP_TRY(b_bad.ensure((size_t)n_range + 64));
        d_bad = b_bad.as<uint8_t>();
        HIP_TRY(hipMemsetAsync(d_bad, 0, n_range, st));
    }
    const int pairs = inflate_plan_for(bam_len, total_all);      // (the whole file's ratio: a range of it is the same kind of payload)
    // (early tail: the second of two gated launches is still running on a side stream; whoever reads its part of the arena, or the launch's
    //  verdict, first makes the pipeline's stream wait for it)
    c->launch_timed = false;
    auto timed_launch = [&](hipStream_t q, bool piece, InflateGate gate) {      // the call's whole-range launch, with its own pair of events on its own stream
        if (c->link) { c->chip_hold.take(&c->link->chip); mark("the chip's DEFLATE turn is ours"); }
        (void)hipEventRecord(c->ev_launch[0], q);
        launch_inflate(d_bam, d_members + m_lo, n_range, b_arena.as<uint8_t>(), upos_lo, b_lens.as<uint32_t>(), d_sc, q, ignore_below, 0, piece, 0, d_bad, pairs, false, gate);
        (void)hipEventRecord(c->ev_launch[1], q);
        if (c->link && hipLaunchHostFunc(q, [](void *h) { ((TurnHold *)h)->give(); }, &c->chip_hold) != hipSuccess) { (void)hipGetLastError(); c->chip_hold.give(); }
        c->launch_timed = true;
    };
    if (!overlap) timed_launch(st, false, InflateGate());
    else {
        // one launch per upload chunk, on the side streams: the members whose bytes (plus the decoder's 16-byte look-ahead) have arrived with
        // chunk j start as soon as its event fires, next to the launches of the chunks before it
        HIP_TRY(b_lens.ensure(inflate_scratch_bytes(std::max<uint32_t>(n_range, 64)) + up.end.size() * inflate_scratch_bytes(64)));
        HIP_TRY(hipEventRecord(c->ev_ready, st));
        for (auto &q : c->side) if (q) HIP_TRY(hipStreamWaitEvent(q, c->ev_ready, 0));
        uint32_t g_lo = m_lo; size_t scratch_off = 0; unsigned used_side = 0;
        if (gated) {
            // one launch on the pipeline's stream, now: its waves wait for their chunk's flag themselves (k_inflate_coop; a range the wave form
            // takes -- a few thousand members -- is one launch behind the last chunk)
            if (inflate_takes_coop(n_range)) {
                InflateGate gate;
                gate.flags = c->buf("gate_flags").as<uint32_t>(); gate.epoch = c->gate_epoch; gate.n_chunks = (uint32_t)up.end.size(); gate.lo = up.lo; gate.chunk_bytes = gate_chunk;
                // Round 4, second half ("early tail"): the launch goes to a side stream and counts its finished waves per PART of the member list
                // (parts cut where upload chunks end, at multiples of the lane-sorting group); the pipeline's stream waits for part after part
                // (launch_wait_done) and frames, verifies and decodes the part of the arena behind it while the waves of the later parts still
                // run -- what is left behind the launch's end is the last part's framing and decode, not the whole file's.  (A member's own
                // chain puts the end of the launch 6-9 ms behind the last chunk's arrival, whatever the chip does meanwhile.)
                // REGTOOLS_AMD_EARLY_TAIL="6,9,12,14" = the cuts in sixteenths of the upload (up to seven; the default since round 5: 23.5 ms per step where "8,12,14" gives 24.5 -- the tail
                // under the launch is the critical path from the first part on, so it starts earlier and in smaller parts; profiles/r05_step_early_tail_four_cuts_ab.txt), "0" = off.
                static const std::vector<unsigned> env_cuts = [] {
                    std::vector<unsigned> v; const char *e = getenv("REGTOOLS_AMD_EARLY_TAIL");
                    unsigned x7[7] = {0, 0, 0, 0, 0, 0, 0};
                    const int n = sscanf(e ? e : "6,9,12,14", "%u,%u,%u,%u,%u,%u,%u", &x7[0], &x7[1], &x7[2], &x7[3], &x7[4], &x7[5], &x7[6]);
                    for (unsigned x : x7) if ((int)v.size() < n && x > 0 && x < 16 && (v.empty() || x > v.back())) v.push_back(x);
                    return v;
                }();
                const uint32_t early_min = overlap_knobs().early_min;
                if (!env_cuts.empty() && !c->early_distrust && c->side[1] && up.end.size() >= 8 && !d_bad && !d_true_sizes) {
                    const uint32_t align = kInflateSortGroup;          // (a wave's members all come from one group of that many)
                    for (unsigned cut : env_cuts) {
                        const size_t kA = std::max<size_t>(1, up.end.size() * (size_t)cut / 16);
                        const uint64_t lim_b = up.end[kA - 1];
                        uint32_t lo = m_lo, hi = m_hi;             // first member that reads bytes behind chunk kA - 1 (k_inflate_coop's own rule)
                        while (lo < hi) { const uint32_t mid = lo + (hi - lo) / 2; if (hm[mid].cpos + hm[mid].clen + 24 <= lim_b) lo = mid + 1; else hi = mid; }
                        uint32_t k = (lo - m_lo) / align * align;     // members of the range in front of the cut
                        const uint32_t prev = early_parts.empty() ? 0u : early_parts.back().members;
                        if (k >= prev + early_min && n_range - k >= early_min) early_parts.push_back(EarlyPart{k, k / 64, hm[m_lo + k].upos - upos_lo});
                    }
                }
                if (!early_parts.empty()) {
                    DevBuf &bd = c->buf("gate_done");
                    HIP_TRY(bd.ensure(64));
                    uint32_t *d_done = bd.as<uint32_t>();
                    hipStream_t q = c->side[1];
                    HIP_TRY(hipMemsetAsync(d_done, 0, 32, q));
                    gate.done = d_done;
                    for (size_t j = 0; j < early_parts.size(); ++j) gate.part_start[j] = early_parts[j].waves;
                    timed_launch(q, /*piece=*/true, gate);
                    HIP_TRY(hipEventRecord(c->ev_side[1], q));
                    split_ev = c->ev_side[1];
                    split_B = true;
                } else timed_launch(st, /*piece=*/true, gate);
            } else {
                while (up.recorded.load(std::memory_order_acquire) < up.end.size()) std::this_thread::yield();
                HIP_TRY(hipStreamWaitEvent(st, c->chunk_ev[up.end.size() - 1], 0));
                timed_launch(st, false, InflateGate());
            }
            g_lo = m_hi;
        }
        for (size_t j = 0; j < up.end.size() && g_lo < m_hi; ++j) {
            uint32_t g_hi = m_hi;
            if (j + 1 < up.end.size()) {     // first member of [g_lo, m_hi) that needs bytes beyond this chunk
                const uint64_t lim_b = up.end[j];
                uint32_t lo = g_lo, hi = m_hi;
                while (lo < hi) { const uint32_t mid = lo + (hi - lo) / 2; if (hm[mid].cpos + hm[mid].clen + 16 <= lim_b) lo = mid + 1; else hi = mid; }
                g_hi = lo;
            }
            if (g_hi == g_lo) continue;
            while (up.recorded.load(std::memory_order_acquire) <= j) std::this_thread::yield();      // (an event must have been recorded before a stream can wait on it)
            if (up.err) return fail(err, errlen, RGX_ERR_DEVICE, "regtools_amd: upload failed\n");
            // the pipeline's own stream is idle until the inflate is done: it takes every third piece (the runtime maps streams onto four
            // hardware queues round-robin; a third side stream would share its queue with the second: profiles/r02_overlap_timeline.txt)
            const bool own = j + 1 == up.end.size() || j >= (size_t)kSideStreams || !c->side[j];
            hipStream_t q = own ? st : c->side[j];
            if (!own) used_side |= 1u << j;
            HIP_TRY(hipStreamWaitEvent(q, c->chunk_ev[j], 0));
            launch_inflate(d_bam, d_members + g_lo, g_hi - g_lo, b_arena.as<uint8_t>(), upos_lo, (uint32_t *)(b_lens.as<uint8_t>() + scratch_off), d_sc, q, ignore_below, g_lo - m_lo, /*piece=*/true, 0, d_bad, pairs);
            scratch_off += inflate_scratch_bytes(g_hi - g_lo);
            g_lo = g_hi;
        }
        up.th.join();
        if (up.err) { (void)hipStreamSynchronize(st); return fail(err, errlen, RGX_ERR_DEVICE, "regtools_amd: upload failed\n"); }      // (gated waves give up after ~2 s)
        for (unsigned k = 0; k < (unsigned)kSideStreams; ++k) if (used_side >> k & 1) { HIP_TRY(hipEventRecord(c->ev_side[k], c->side[k])); HIP_TRY(hipStreamWaitEvent(st, c->ev_side[k], 0)); }
        HIP_TRY(hipStreamWaitEvent(st, c->chunk_ev[up.end.size() - 1], 0));      // (later stages read the file too: barcodes, header)
    }
    HIP_TRY(hipEventRecord(c->ev[1], st));
    mark(gated && overlap && inflate_takes_coop(n_range) ? "launch inflate (gated)" : "launch inflate");

    return kGoOn;
}

int EventsRun::stage_footers_and_header() {
    // -- files whose ISIZE footers lie ------------------------------------------------------------------------------------------
    // The arena was laid out from the footers; the reference never reads them (inflate_block, bgzf.c:292-316: a block is as long as
    // zlib says, at most 64 KiB).  When a member inflates to another length than its footer claims, or the member that ends the
    // stream is not the plain empty block it claims to be, every member is inflated once into its own 64 KiB slot to learn the true
    // lengths and the pipeline starts over with those.  Costs two extra inflate passes; only malformed files ever pay them.
    // Host input whose members the host scan vouched for: no round trip here.  The header comes from the host's own inflate of the file's head,
    // the stages behind the inflate are enqueued on the assumption that every member inflates to its footer's length (what a well-formed file
    // does), and the launch's verdict is read with the framing's counts: anything else starts over on the device-resident path below.
    DevBuf &b_arena = c->buf("arena"), &b_hdr = c->buf("hdr_arena"), &b_lens = c->buf("inflate_scratch");
    BamHeader hdr_host;
    mean_rec = 0;                                    // mean size of the file's first records (0 = unknown: 16 KiB segments)
    spec = overlap && !d_true_sizes && !geom_chunked_hint && host_bam_header(h_bam, std::min<size_t>(bam_len, (size_t)8 << 20), hdr_host, nullptr, &mean_rec);
    if (spec) { h_sc[0] = h_sc[1] = 0xffffffffu; h_sc[kStatusEarly] = h_sc[kStatusEarly + 1] = 0xffffffffu; }
    else {
        HIP_TRY(join_B());
        HIP_TRY(hipMemcpyAsync(h_sc, d_sc, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(h_sc + kStatusEarly, d_sc + kStatusEarly, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    if (!d_true_sizes) {
        auto size_trouble = [&](uint32_t k) { return h_sc[k] != 0xffffffffu && (h_sc[k + 1] == 12u /* INF_SIZE_MISMATCH */ || h_sc[k + 1] == 10u /* INF_OUT_OVERFLOW */); };
        bool lies = size_trouble(0) || size_trouble(kStatusEarly);
        if (!lies && stop < n_members_all) {
            Member ms; uint8_t two[2] = {0, 0};
            HIP_TRY(member_at(stop, ms));
            if (ms.isize == 0 && ms.clen >= 2) {
                if (h_bam && ms.cpos + 2 <= bam_len) memcpy(two, h_bam + ms.cpos, 2);
                else { HIP_TRY(hipMemcpyAsync(two, d_bam + ms.cpos, 2, hipMemcpyDeviceToHost, st)); HIP_TRY(hipStreamSynchronize(st)); }
            }
            // fine: an empty block (03 00, the EOF marker) or a member cut off by the end of the file
            lies = !(ms.isize == 0 && two[0] == 3 && two[1] == 0) && ms.isize != 0xffffffffu;
        }
        if (lies) {
            mark("footer mismatch: probing");
            DevBuf &b_slots = c->buf("probe_slots"), &b_sizes = c->buf("probe_sizes");
            HIP_TRY(b_slots.ensure((size_t)n_members_all * kBgzfMaxBlock + 256));
            HIP_TRY(b_sizes.ensure((size_t)n_members_all * 4 + 64));
            HIP_TRY(b_lens.ensure(inflate_scratch_bytes(std::max<uint32_t>(n_members_all, 64))));
            launch_inflate_probe(d_bam, d_members, n_members_all, b_slots.as<uint8_t>(), b_lens.as<uint32_t>(), b_sizes.as<uint32_t>(), st);
            HIP_TRY(hipStreamSynchronize(st));
            b_slots.release();                                  // 64 KiB per member: not kept
            const int rc2 = prepare_events(c, d_bam_in, h_bam, bam_len, bai, bai_len, p, want_read_span, P, err, errlen, b_sizes.as<uint32_t>(), true, region_to_file_end);
            P.t_begin = t_begin;
            return rc2;
        }
    }

    // -- header (sam.c:114-223): it sits at the start of the arena when the range starts at member 0; otherwise the head of
    //    the file is inflated into its own small arena -----------------------------------------------------------------------
    if (spec) hdr = hdr_host;
    else {
        const uint32_t h_early = h_sc[kStatusEarly];            // read back right after the launch finished (below the footer check)
        uint32_t n_h = std::min<uint32_t>(n_members_all, 4);
        for (;;) {
            const uint8_t *src; uint64_t have;
            uint32_t bad_h = 0xffffffffu;
            std::vector<Member> hmem(n_h);
            HIP_TRY(hipMemcpyAsync(hmem.data(), d_members, (size_t)n_h * sizeof(Member), from_members, st));
            HIP_TRY(hipStreamSynchronize(st));
            uint32_t used = 0;
            for (; used < n_h; ++used) if (hmem[used].isize == 0 || hmem[used].isize > kBgzfMaxBlock) break;   // the header read stops there
            if (!used) return fail(err, errlen, RGX_ERR_REGION, "%s", kMsgRegion);
            have = hmem[used - 1].upos + hmem[used - 1].isize;
            if (m_lo == 0 && used <= n_range) src = b_arena.as<uint8_t>();
            else {
                HIP_TRY(b_hdr.ensure(have + 256));
                HIP_TRY(hipMemsetAsync(d_sc + 12, 0xff, 4, st));
                launch_inflate(d_bam, d_members, used, b_hdr.as<uint8_t>(), 0, b_lens.as<uint32_t>(), d_sc + 12, st);
                src = b_hdr.as<uint8_t>();
            }
            std::vector<uint8_t> hbuf(have);
            HIP_TRY(hipMemcpyAsync(hbuf.data(), src, have, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(h_sc, d_sc, 64, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            bad_h = (src == b_arena.as<uint8_t>()) ? std::min(h_sc[0], h_early) : h_sc[12];     // (members in front of a seek target report apart)
            if (bad_h != 0xffffffffu && bad_h < used) have = hmem[bad_h].upos;          // a corrupt member ends the header read
            uint64_t need = 0;
            int r = parse_bam_header(hbuf.data(), have, hdr, need);
            if (r == 0) {
                // how long the file's records are, from the first ones behind the header in the bytes at hand (the host's estimate on the spec path):
                // files of long records are framed in long segments
                uint64_t o = hdr.end, sum = 0; uint32_t cnt = 0;
                while (o + 4 <= have) {
                    uint32_t bl; memcpy(&bl, hbuf.data() + o, 4);
                    if (bl < 32 || bl > (1u << 27) || o + 4 + bl > have) break;
                    sum += 4 + (uint64_t)bl; ++cnt; o += 4 + (uint64_t)bl;
                }
                if (cnt >= 4) mean_rec = (uint32_t)(sum / cnt);
                break;
            }
            if (r == 2 || used < n_h || n_h == n_members_all || (bad_h != 0xffffffffu && bad_h < used))
                return fail(err, errlen, RGX_ERR_REGION, "%s", kMsgRegion);   // sam_hdr_read == NULL (cc:519-522)
            n_h = std::min(n_members_all, n_h * 4);
        }
    }
    n_ref = (int32_t)hdr.names.size();
    mark("header (sync: inflate done)");

    return kGoOn;
}

int EventsRun::stage_bounds_and_chains() {
    // -- stream bounds inside the arena -------------------------------------------------------------------------------------
    lim = total;
    if (h_sc[0] != 0xffffffffu) {       // a member of the range failed to inflate: the stream ends where it starts
        uint64_t u = 0;
        HIP_TRY(upos_of(m_lo + h_sc[0], u));
        lim = u - upos_lo;
    }
    auto arena_of = [&](uint64_t voff, uint32_t idx, uint64_t upos) -> uint64_t {   // virtual offset -> arena offset (idx/upos from the query)
        if (idx >= n_members_all || idx < m_lo || idx >= m_hi) return total;
        return std::min<uint64_t>(total, upos - upos_lo + (voff & 0xffff));
    };
    if (cut_lo) pos0 = arena_of(cut_lo, h_sc[25], q_upos[1]);
    else pos0 = hdr.end;                 // no seek: records start right after the header (range starts at member 0)
    // Did the stream stop for a reason that ends iteration upstream, rather than at this shard's upper cut?  (A later shard is a seek past
    // that point; the merge drops the shards behind one that ended, so that a damaged file gives the same table whatever the shard count.)
    chain_ended = false;
    P.stream_ended = empty_stream;
    if (cut_hi != UINT64_MAX) {
        const uint64_t cut_lim = arena_of(cut_hi, h_sc[26], q_upos[2]);
        const uint32_t mh = h_sc[26];
        const uint32_t hi_wanted = (mh < n_members_all && (cut_hi & 0xffff)) ? mh + 1 : mh;
        // (a region's chunks are seeks of their own: what lies between two of them ends nothing, and a chunk whose reader does run into such a
        //  member reports it through its chain, below)
        if (!chunked && h_sc[0] != 0xffffffffu && lim < cut_lim) P.stream_ended = true;        // a member in front of the cut does not inflate
        if (!chunked && stop < std::min(hi_wanted, n_members_all)) P.stream_ended = true;      // an empty / unusable member in front of the cut (bgzf.c:548-578)
        lim = std::min(lim, cut_lim);
    } else if (h_sc[0] != 0xffffffffu) P.stream_ended = true;
    if (pos0 > lim) pos0 = lim;
    if (empty_stream) lim = pos0;            // the seek target does not exist: no record is read

    memset(&cfg, 0, sizeof cfg);
    cfg.n_ref = n_ref; cfg.strandness = p->strandness; cfg.tag0 = (uint8_t)p->strand_tag[0]; cfg.tag1 = (uint8_t)p->strand_tag[1];
    cfg.min_anchor = p->min_anchor; cfg.min_intron = p->min_intron; cfg.max_intron = p->max_intron;
    cfg.region_tid = -2; cfg.long_threshold = 16;
    if (!whole) {
        int32_t tid, beg, end;
        if (!parse_region(hdr, p->region, tid, beg, end) || tid >= bi.n_ref || end < beg)
            return fail(err, errlen, RGX_ERR_REGION, "%s", kMsgRegion);
        cfg.region_tid = tid; cfg.region_beg = beg; cfg.region_end = end;
    }

    // -- intron-motif strand rule: FASTA bytes + one descriptor per BAM contig in HBM (junctions_extractor.cc:345-359) ------
    if (p->fasta_path) {
        if (!c->fasta || c->fasta_path != p->fasta_path) {
            delete c->fasta; c->fasta = new Fasta(); c->fasta_path.clear();
            if (!c->fasta->load(p->fasta_path)) { delete c->fasta; c->fasta = nullptr; return fail(err, errlen, RGX_ERR_FASTA, "Unable to open FASTA file.\n\n"); }
            DevBuf &bf = c->buf("fasta");
            HIP_TRY(bf.ensure(c->fasta->size + 256));
            HIP_TRY(hipMemcpy(bf.p, c->fasta->data, c->fasta->size, hipMemcpyHostToDevice));
            c->fasta_path = p->fasta_path;
        }
        std::vector<FaContig> tab((size_t)std::max(n_ref, 1));
        memset(tab.data(), 0, tab.size() * sizeof(FaContig));
        for (int32_t t = 0; t < n_ref; ++t)
            for (const Fasta::Seq &s : c->fasta->seqs)
                if (s.name == hdr.names[(size_t)t]) {
                    // the kernels index fa[offset + p / line_blen * line_len + p % line_blen] without a bounds check: a descriptor that does
                    // not fit the file (stale or damaged .fai) makes the contig absent -- a junction there then fails the call like a contig
                    // the FASTA does not have (junctions_extractor.cc:553), instead of reading HBM out of bounds
                    const bool sane = s.offset >= 0 && s.len >= 0 && s.line_blen > 0 && s.line_len >= s.line_blen &&
                                      (s.len == 0 || (uint64_t)s.offset + (uint64_t)((s.len - 1) / s.line_blen) * (uint64_t)s.line_len + (uint64_t)((s.len - 1) % s.line_blen) < (uint64_t)c->fasta->size);
                    if (!sane) continue;
                    tab[(size_t)t].offset = s.offset; tab[(size_t)t].len = s.len; tab[(size_t)t].line_blen = s.line_blen; tab[(size_t)t].line_len = s.line_len; tab[(size_t)t].present = 1;
                }
        DevBuf &bt = c->buf("fasta_tab");
        HIP_TRY(bt.ensure(tab.size() * sizeof(FaContig) + 64));
        HIP_TRY(hipMemcpy(bt.p, tab.data(), tab.size() * sizeof(FaContig), hipMemcpyHostToDevice));
        cfg.fa_data = c->buf("fasta").as<uint8_t>(); cfg.fa_tab = bt.as<FaContig>(); cfg.fa_missing = d_sc + 64;
    }

    // -- region queries: one record chain per chunk of the iterator -----------------------------------------------------------------
    // chunk c = virtual offsets [u, v): a seek to u (bgzf_seek: the member at u >> 16, the offset inside it clipped to its length; no such
    // member = the read fails and the iteration is over), then records while the position in front of the next one is below v.
    memset(&geom, 0, sizeof geom);
    const int env_seg = decode_knobs().seg_bytes;                   // (tests) 16384 or 131072
    seg_bytes = env_seg == (int)kSegBytes || env_seg == (int)kSegBytesLong ? (uint32_t)env_seg : (mean_rec >= kLongRecordBytes ? kSegBytesLong : kSegBytes);
    geom.seg_bytes = seg_bytes;
    if (chunked && chunks.empty()) lim = pos0;                // an iterator without chunks returns nothing
    if (chunked && !chunks.empty() && !empty_stream) {
        std::vector<Member> rm(n_range);
        std::vector<uint8_t> bad(n_range, 0);
        if (n_range) {
            HIP_TRY(hipMemcpyAsync(rm.data(), d_members + m_lo, (size_t)n_range * sizeof(Member), from_members, st));
            HIP_TRY(hipMemcpyAsync(bad.data(), d_bad, n_range, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
        }
        for (uint32_t k = 0; k < n_range; ++k) if (rm[k].isize == 0 || rm[k].isize > kBgzfMaxBlock) bad[k] = 1;     // bgzf.c:548-578: an empty block reads as the end of the file
        std::vector<uint32_t> next_bad((size_t)n_range + 1, n_range);
        for (uint32_t k = n_range; k-- > 0;) next_bad[k] = bad[k] ? k : next_bad[k + 1];
        auto usize = [&](uint32_t k) { return rm[k].isize <= kBgzfMaxBlock ? rm[k].isize : 0u; };
        auto lower = [&](uint64_t cfile) {                    // first member of the range at or behind file offset cfile
            uint32_t lo_ = 0, hi_ = n_range;
            while (lo_ < hi_) { const uint32_t mid = lo_ + (hi_ - lo_) / 2; if (rm[mid].cpos - 18 < cfile) lo_ = mid + 1; else hi_ = mid; }
            return lo_;
        };
        uint32_t seg_total = 0;
        for (const VChunk &ch : chunks) {
            const uint32_t k = lower(ch.u >> 16);
            if (k >= n_range || rm[k].cpos - 18 != (ch.u >> 16)) break;        // the seek lands on no member: upstream's next read fails, nothing behind it is read
            SegChunk sc; memset(&sc, 0, sizeof sc);
            sc.a = rm[k].upos - upos_lo + std::min<uint64_t>(ch.u & 0xffff, usize(k));
            const uint32_t kv = lower(ch.v >> 16);
            if (kv >= n_range) sc.b = total;
            else sc.b = rm[kv].upos - upos_lo + (rm[kv].cpos - 18 == (ch.v >> 16) ? std::min<uint64_t>(ch.v & 0xffff, usize(kv)) : 0);
            if (sc.b <= sc.a) sc.b = sc.a + 1;                              // the first record behind a seek is read whatever the chunk's end says
            const uint32_t nb = next_bad[k];
            sc.dlim = nb < n_range ? rm[nb].upos - upos_lo : total;
            sc.seg_base = seg_total;
            const uint64_t ns = (sc.b - sc.a + seg_bytes - 1) / seg_bytes;
            if (seg_total + ns > 0x7fffffffull) return fail(err, errlen, RGX_ERR_FORMAT, "regtools_amd: region too large\n");
            seg_total += (uint32_t)ns;
            seg_chunks.push_back(sc);
        }
        if (seg_chunks.empty()) lim = pos0;
        else {
            DevBuf &b_ch = c->buf("seg_chunks");
            HIP_TRY(b_ch.ensure(seg_chunks.size() * sizeof(SegChunk) + 64));
            HIP_TRY(hipMemcpyAsync(b_ch.p, seg_chunks.data(), seg_chunks.size() * sizeof(SegChunk), hipMemcpyHostToDevice, st));
            geom.chunks = b_ch.as<SegChunk>(); geom.n_chunks = (uint32_t)seg_chunks.size();
        }
        mark("chunk table");
    }

    return kGoOn;
}

int EventsRun::stage_framing() {
    // -- record framing ------------------------------------------------------------------------------------------------
    arena = c->buf("arena").as<uint8_t>();
    span = lim - pos0;
    n_seg = (uint32_t)((span + seg_bytes - 1) / seg_bytes);
    geom.pos0 = pos0; geom.lim = lim; geom.data_end = lim; geom.seg_bytes = seg_bytes;
    const int env_lite = 1;
    lite_walk = env_lite && !c->walk_strict;
    geom.lite_walk = lite_walk ? 1u : 0u;
    if (geom.chunks) {
        span = 0;
        for (const SegChunk &sc : seg_chunks) span += sc.b - sc.a;
        const SegChunk &lastc = seg_chunks.back();
        n_seg = lastc.seg_base + (uint32_t)((lastc.b - lastc.a + seg_bytes - 1) / seg_bytes);
        geom.data_end = total;
    }
    n_rec = 0;
    DevBuf &b_seg = c->buf("seg"), &b_tmp = c->buf("tmp");
    HIP_TRY(hipEventRecord(c->ev[2], st));
    // Early tail (round 4): while the side stream's launch still inflates the members of the last upload chunks, the segments that lie wholly
    // in front of their part of the arena (one member's margin: a walk only ever reads the 36 bytes behind its segment, a guess that
    // reads further is only a guess) are framed, verified and decoded -- exact for the same reason the whole chain is: segment 0 starts at
    // an exact offset.  Plain whole-file calls on 16 KiB segments only; anything unusual in the prefix (the chain ends there, sweeps beyond
    // the usual one) drops back to the one-pass order.
    sA = 0;
    emit_parts_ok = false; emit_parts = 0; emit_rows = 0; ev_lay = 0;
    memset(&ev_e, 0, sizeof ev_e);
    soa_cap = 0;
    memset(&soa, 0, sizeof soa);
    ev_base = nullptr; long_list = nullptr;
    if (n_seg) {
        const size_t per = (size_t)n_seg;
        HIP_TRY(b_seg.ensure(per * (8 + 8 + 4) * 2 + per * 4 + per * 12 + 64));
        HIP_TRY(c->buf("seg_cp").ensure(per * kSegCpSlots * 2 + 64));
        seg_cp = c->buf("seg_cp").as<uint16_t>();
        uint8_t *q = b_seg.as<uint8_t>();
        for (int k = 0; k < 2; ++k) { seg_start[k] = (uint64_t *)q; q += per * 8; seg_exit[k] = (uint64_t *)q; q += per * 8; }
        for (int k = 0; k < 2; ++k) { seg_cnt[k] = (uint32_t *)q; q += per * 4; }
        seg_base = (uint32_t *)q; q += per * 4;
        seg_iter_e = (uint32_t *)q; q += per * 4; seg_long_e = (uint32_t *)q; q += per * 4; seg_long_base_e = (uint32_t *)q;
        HIP_TRY(b_tmp.ensure(scan_tmp_words(n_seg) * 4 + 64));
        bool early = split_B && spec && !geom.chunks && seg_bytes == kSegBytes && cut_hi == UINT64_MAX && !empty_stream && lim == total;
        const bool env_early_emit = true;
        const bool small_ok = overlap_knobs().early_small;
        uint32_t waves_done = 0;
        for (size_t j = 0; early && j < early_parts.size(); ++j) {
            const EarlyPart &ep = early_parts[j];
            if (ep.upos <= pos0 + 2 * (uint64_t)kBgzfMaxBlock) continue;
            uint32_t sJ = (uint32_t)std::min<uint64_t>(n_seg, (ep.upos - kBgzfMaxBlock - pos0) / seg_bytes);
            if (small_ok ? (sJ < sA + 2 || n_seg - sJ < 1) : (sJ < sA + 1024 || n_seg - sJ < 64)) continue;
            // the stream waits until every wave of parts 0..j has finished (the counters of the parts are waited for in turn)
            HIP_TRY(hipMemsetAsync(d_sc + 83, 0, 4, st));
            for (size_t i = 0; i <= j; ++i) {
                const uint32_t w_end = early_parts[i].waves, w_beg = i ? early_parts[i - 1].waves : 0u;
                if (w_end > waves_done) { launch_wait_done(c->buf("gate_done").as<uint32_t>() + i, w_end - w_beg, d_sc + 83, st); waves_done = w_end; }
            }
            HIP_TRY(hipMemcpyAsync(h_sc + 83, d_sc + 83, 4, hipMemcpyDeviceToHost, st));      // (read behind the framing's first wait for the stream)
            if (trace) fprintf(stderr, "[rgx trace] early tail: part %zu: members < %u, segments [%u, %u) of %u\n", j, ep.members, sA, sJ, n_seg);
            bool ended_J = false;
            const uint32_t sweeps0 = P.framing_sweeps;
            const int rcJ = frame(sJ, sA, ended_J);
            if (rcJ != -1) return rcJ;
            const uint32_t n_rec_J = h_sc[3];
            // Where the last record that STARTS in the prefix ends = the verified chain's exit from its last segment.  The margin between the prefix
            // and the part of the arena still being inflated is one member; a record that reaches past it (a CIGAR of tens of thousands of
            // operations, a read of tens of kilobases) would be decoded from bytes that may not be there yet: such a file takes the one-pass order.
            uint64_t exit_J = 0;
            HIP_TRY(hipMemcpyAsync(h_sc + 92, seg_exit[cur] + (sJ - 1), 8, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            memcpy(&exit_J, h_sc + 92, 8);
            const bool reaches_J = exit_J > ep.upos;
            if (reaches_J && trace) fprintf(stderr, "[rgx trace] early tail: a record of the prefix ends at %llu, behind the inflated part (%llu): one pass\n", (unsigned long long)exit_J, (unsigned long long)ep.upos);
            const bool slow_J = P.framing_sweeps - sweeps0 > 2;
            P.framing_sweeps = sweeps0;                               // (the sweeps over everything, below, are the call's count)
            const uint64_t span_J = (uint64_t)sJ * seg_bytes;
            if (h_sc[83]) { c->early_distrust = true; if (trace) fprintf(stderr, "[rgx trace] early tail: a wait for a part's waves timed out, this context no longer uses it\n"); }
            if (h_sc[83] || ended_J || slow_J || reaches_J || !n_rec_J || span_J / n_rec_J > kSparseRecordBytes ||
                (sA && n_rec_J > soa_cap)) { early = false; sA = 0; emit_parts_ok = false; emit_parts = 0; emit_rows = 0; break; }     // not the plain case: one pass over everything below
            if (!sA) {
                // rows for the whole file, estimated from the first part (+ 1/8); when the estimate turns out short the decode is simply made again below
                HIP_TRY(soa_layout((size_t)((double)n_rec_J * ((double)n_seg / sJ) * 1.125) + 65536));
                cfg.insane_out = nullptr;
                if (lite_walk) { cfg.insane_out = d_sc + 82; HIP_TRY(hipMemsetAsync(d_sc + 82, 0, 4, st)); h_sc[82] = 0; }
            }
            launch_decode_seg(arena, geom, sJ, seg_start[cur], seg_base, seg_cnt[cur], cfg, soa, seg_iter_e, seg_long_e, seg_cp, /*staged=*/true, st, sA);
            // ... and its junction events emitted, into the events block the context's last call left (no count is known yet, so nothing can be
            // sized: a first call, or a block that turns out too small, emits everything at the end as before).  ev_base counts from the part's
            // first row; the totals of the parts stay on the device, k_emit_short adds those in front of its part.
            if (!sA) {
                DevBuf &b_ev0 = c->buf("events");
                ev_lay = b_ev0.cap > 256 ? (b_ev0.cap - 256) / 33 : 0;
                emit_parts_ok = env_early_emit && ev_lay >= 4096;
                if (emit_parts_ok) { HIP_TRY(b_tmp.ensure(scan_tmp_words((uint32_t)std::min<size_t>(soa_cap, 0xffffffffu)) * 4 + 64)); ev_e = ev_layout(b_ev0.as<uint8_t>(), ev_lay); }
            }
            if (emit_parts_ok && emit_parts < kGateParts - 1) {
                launch_scan_u32(soa.n_ev + emit_rows, ev_base + emit_rows, n_rec_J - emit_rows, d_sc + 84 + emit_parts, b_tmp.as<uint32_t>(), st);
                launch_emit_short(arena, n_rec_J, cfg, soa, ev_base, ev_e, st, emit_rows, d_sc + 84, emit_parts, (uint32_t)std::min<size_t>(ev_lay, 0xffffffffu));
                emit_rows = n_rec_J; ++emit_parts;
            }
            sA = sJ;
            mark("early tail: part framed and decoded");
        }
        HIP_TRY(join_B());
        const int rcF = frame(n_seg, sA, chain_ended);
        if (rcF != -1) return rcF;
        n_rec = h_sc[3];
    } else HIP_TRY(join_B());
    if (spec && !n_seg) {                                     // (no framing, no sync yet: the inflate's verdict is still due)
        HIP_TRY(hipMemcpyAsync(h_sc, d_sc, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(h_sc + kStatusEarly, d_sc + kStatusEarly, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (h_sc[0] != 0xffffffffu || h_sc[kStatusEarly] != 0xffffffffu) {
            HIP_TRY(complete_upload());
            HIP_TRY(hipStreamSynchronize(copy_q));
            const int rc2 = prepare_events(c, d_bam, nullptr, bam_len, bai, bai_len, p, want_read_span, P, err, errlen, nullptr, false, region_to_file_end);
            P.t_begin = t_begin;
            return rc2;
        }
    }
    if (chain_ended) P.stream_ended = true;
    if (chain_ended && geom.chunks && m_hi < n_members_all && !region_to_file_end) {
        // a chunk's chain stopped -- possibly only because a record runs past the members the index asked for (an index that does not
        // describe this file): once more with everything up to the end of the file inflated
        mark("region: chain ended, re-reading to the end of the file");
        if (overlap) { HIP_TRY(complete_upload()); HIP_TRY(hipStreamSynchronize(copy_q)); }
        const int rc2 = prepare_events(c, d_bam, nullptr, bam_len, bai, bai_len, p, want_read_span, P, err, errlen, d_true_sizes, false, true);
        P.t_begin = t_begin;
        return rc2;
    }
    HIP_TRY(hipEventRecord(c->ev[3], st));
    mark("framing (sync)");

    return kGoOn;
}

int EventsRun::stage_decode() {
    // -- decode + count -----------------------------------------------------------------------------------------------------
    DevBuf &b_tmp = c->buf("tmp");
    n_events = 0; n_long = 0;
    n_iterated = 0;
    if (n_rec) {
        const size_t R = n_rec;
        // (early tail: the prefix is decoded already, into columns laid out for an estimate of the row count; when that was short, or the chain
        //  ended after all, everything is decoded again)
        uint32_t s_from = sA;
        if (sA && (R > soa_cap || chain_ended)) s_from = 0;
        if (!s_from) { HIP_TRY(soa_layout(R)); emit_parts_ok = false; }
        HIP_TRY(b_tmp.ensure(scan_tmp_words(n_rec) * 4 + 64));
        // per-segment outputs (no hot atomics): reuse the spare segment arrays as seg_iter / seg_long
        uint32_t *seg_iter = sA ? seg_iter_e : seg_cnt[cur ^ 1], *seg_long = sA ? seg_long_e : (uint32_t *)seg_start[cur ^ 1], *seg_long_base = sA ? seg_long_base_e : (uint32_t *)seg_exit[cur ^ 1];
        // the iterator's end rule only where the iterator's chunks are followed (hts_itr_next, hts.c:1946-1950): the first pass finds the
        // first record that ends the iteration and the last one that passed the overlap test; when one of those lies behind the other
        // (records out of order -- no indexer writes such a file) the pass is repeated with the stop in place
        if (geom.chunks) {
            cfg.stop_out = d_sc + 80; cfg.stop_index = 0xffffffffu;
            HIP_TRY(hipMemsetAsync(d_sc + 80, 0xff, 4, st));
            HIP_TRY(hipMemsetAsync(d_sc + 81, 0, 4, st));
        }
        if (!s_from) {
            cfg.insane_out = nullptr;
            if (lite_walk) { cfg.insane_out = d_sc + 82; HIP_TRY(hipMemsetAsync(d_sc + 82, 0, 4, st)); h_sc[82] = 0; }
        }
        for (int pass = 0; pass < 2; ++pass) {
            launch_decode_seg(arena, geom, n_seg, seg_start[cur], seg_base, seg_cnt[cur], cfg, soa, seg_iter, seg_long, seg_cp,
                              /*staged=*/span / n_rec <= kSparseRecordBytes, st, s_from);
            if (emit_parts_ok && s_from) {
                // the last part's events, emitted like the others'; the event total = the parts' totals
                launch_scan_u32(soa.n_ev + emit_rows, ev_base + emit_rows, n_rec - emit_rows, d_sc + 84 + emit_parts, b_tmp.as<uint32_t>(), st);
                launch_emit_short(arena, n_rec, cfg, soa, ev_base, ev_e, st, emit_rows, d_sc + 84, emit_parts, (uint32_t)std::min<size_t>(ev_lay, 0xffffffffu));
                HIP_TRY(hipMemcpyAsync(h_sc + 84, d_sc + 84, 4 * kGateParts, hipMemcpyDeviceToHost, st));
            } else launch_scan_u32(soa.n_ev, ev_base, n_rec, d_sc + 4, b_tmp.as<uint32_t>(), st);
            launch_scan_u32(seg_iter, seg_iter, n_seg, d_sc + 8, b_tmp.as<uint32_t>(), st);
            launch_scan_u32(seg_long, seg_long_base, n_seg, d_sc + 5, b_tmp.as<uint32_t>(), st);
            HIP_TRY(hipMemcpyAsync(h_sc + 4, d_sc + 4, 24, hipMemcpyDeviceToHost, st));
            if (cfg.stop_out) HIP_TRY(hipMemcpyAsync(h_sc + 80, d_sc + 80, 8, hipMemcpyDeviceToHost, st));
            if (cfg.insane_out) HIP_TRY(hipMemcpyAsync(h_sc + 82, d_sc + 82, 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            if (cfg.insane_out && h_sc[82]) {
                // a record the reference's reader would not have accepted (sam.c:421-423) lies on the chain the block_size walk followed:
                // the whole call again with the framing making the full test (damaged files only)
                mark("decode: a record fails bam_read1's test, starting over with the full walk");
                if (overlap) { HIP_TRY(complete_upload()); HIP_TRY(hipStreamSynchronize(copy_q)); }
                c->walk_strict = true;
                const int rc2 = prepare_events(c, d_bam, nullptr, bam_len, bai, bai_len, p, want_read_span, P, err, errlen, d_true_sizes, false, region_to_file_end);
                c->walk_strict = false;
                P.t_begin = t_begin;
                return rc2;
            }
            if (pass || !cfg.stop_out || h_sc[80] == 0xffffffffu || h_sc[81] <= h_sc[80] + 1) break;
            cfg.stop_index = h_sc[80];
        }
        n_events = h_sc[4]; n_long = h_sc[5];
        if (emit_parts_ok && s_from) {
            uint64_t tot = 0;
            for (uint32_t k = 0; k <= emit_parts && k < kGateParts; ++k) tot += h_sc[84 + k];
            if (tot > ev_lay || tot > 0xffffffffull || n_long) {
                // the recycled block was too small after all (or wave-per-read rows want their global slots): everything once more, the plain way
                emit_parts_ok = false;
                launch_scan_u32(soa.n_ev, ev_base, n_rec, d_sc + 4, b_tmp.as<uint32_t>(), st);
                HIP_TRY(hipMemcpyAsync(h_sc + 4, d_sc + 4, 4, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                n_events = h_sc[4];
            } else n_events = (uint32_t)tot;
        }
        n_iterated = h_sc[8];
        if (geom.chunks && p->n_shards > 1 && h_sc[80] != 0xffffffffu) P.stream_ended = true;      // this shard read the record that ends the iteration (hts.c:1946-1950)
        if (n_long) launch_long_fill(n_seg, seg_base, seg_cnt[cur], seg_long_base, cfg, soa, long_list, st);
    }
    HIP_TRY(hipEventRecord(c->ev[4], st));
    mark("decode+count (sync)");

    return kGoOn;
}

int EventsRun::stage_emit() {
    // -- emit -----------------------------------------------------------------------------------------------------------------
    DevBuf &b_ev = c->buf("events");
    EventSoA ev; memset(&ev, 0, sizeof ev);
    if (n_events) {
        const size_t E = n_events;
        if (emit_parts_ok && n_rec) ev = ev_e;                  // (early tail: every part's rows are out already)
        else {
            HIP_TRY(b_ev.ensure(E * (4 * 8 + 1) + 256));
            ev = ev_layout(b_ev.as<uint8_t>(), E);
            launch_emit_short(arena, n_rec, cfg, soa, ev_base, ev, st);
            launch_emit_long(arena, long_list, n_long, cfg, soa, ev_base, ev, st);
        }
        if (cfg.fa_data) {
            HIP_TRY(hipMemcpyAsync(h_sc + 64, d_sc + 64, 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            if (h_sc[64]) return fail(err, errlen, RGX_ERR_FASTA, "Unable to extract FASTA sequence for position %s\n\n", hdr.names[(size_t)(h_sc[64] - 1)].c_str());   // cc:553
        }
    }
    HIP_TRY(hipEventRecord(c->ev[5], st));

    P.hdr = hdr; P.arena = arena; P.soa = soa; P.ev = ev; P.n_rec = n_rec; P.n_events = n_events; P.n_range = n_range;
    P.n_iterated = n_iterated; P.total = total; P.t_begin = t_begin;
    return RGX_OK;
}

static int prepare_events(rgx_ctx *c, const uint8_t *d_bam_in, const uint8_t *h_bam, size_t bam_len, const uint8_t *bai, size_t bai_len,
                          const rgx_extract_params *p, bool want_read_span, Prep &P, char *err, size_t errlen, const uint32_t *d_true_sizes,
                          bool allow_overlap, bool region_to_file_end, const SharedMembers *shared) {
    EventsRun r{c, d_bam_in, h_bam, bam_len, bai, bai_len, p, want_read_span, P, err, errlen, d_true_sizes, allow_overlap, region_to_file_end, shared};
    return r.run();
}

// Group-by of junction events + output order, generic over the leading key word `ev.tid` (the contig for
// `junctions extract`, the window for `cis-splice-effects identify`): stable radix sort on (group, start, len*4+class),
// segmented reduce, first-seen naming, then the order sort (rank of group, thick_start, thick_end, name).
struct HostRows {
    std::vector<uint32_t> group, start, end, ts, te, count, name_rank, first_seen, last_seen;
    std::vector<uint8_t> strand;
    size_t n = 0;
    // the same rows as ten u32 columns of n entries in the context's pinned staging block (valid until the next call on the context);
    // filled instead of the vectors when the caller asks for the view only
    const uint32_t *cols = nullptr;
};

// where each event ended up: its unique row, and each unique row's position in the output order (device arrays; the -b pass keys on them)
struct RowMap { uint32_t *ev_urow = nullptr, *urow_pos = nullptr; };
// ask reduce_events for the finished result table: columns written on the device in the host block's layout, one copy, no host loop
struct TableSink { const BamHeader *hdr = nullptr; uint32_t min_anchor = 0; rgx_junction_table *table = nullptr; };

static int reduce_events(rgx_ctx *c, EventSoA ev, uint32_t n_events, uint32_t group_bits, uint32_t ilen_bits, const uint32_t *rank_of_group_host,
                         uint32_t n_groups, HostRows &R, char *err, size_t errlen, bool view_only = false, RowMap *row_map = nullptr, TableSink *sink = nullptr,
                         bool allow_preagg = true /* identify's window pairs (a few million, one small sort) measured 0.3-0.4 ms slower with it */) {
    hipStream_t st = c->stream;
    uint32_t *d_sc = c->buf("scalars").as<uint32_t>();
    uint32_t *h_sc = (uint32_t *)c->pinned;
    DevBuf &b_sort = c->buf("sort"), &b_uni = c->buf("unique");
    c->last_rows_valid = false;            // the "rows_out" block is about to be overwritten
    uint32_t n_unique = 0;
    UniqueSoA u; memset(&u, 0, sizeof u);
    uint32_t *perm[2] = {nullptr, nullptr};
    uint32_t *final_perm = nullptr;
    uint32_t *chrom_rank_rows = nullptr;
    R = HostRows();
    if (n_events) {
        // Round 4: equal keys are grouped per tile of consecutive events first (k_preagg); what is sorted and reduced are the tiles' partial
        // rows.  Callers that need every event's row (the -b pass: row_map) keep the event form.
        const bool preagg = !row_map && allow_preagg;
        PartialSoA pr; memset(&pr, 0, sizeof pr);
        uint32_t *ev_flag = nullptr;           // preagg: one word per EVENT (first-seen flags, then their scan)
        EventSoA sev = ev;                     // what is sorted: the events, or the partial rows
        uint32_t n_s = n_events;
        if (preagg) {
            DevBuf &b_par = c->buf("partials");
            const size_t Ev = n_events;
            HIP_TRY(b_par.ensure(Ev * 4 * 9 + scan_tmp_words(n_events) * 4 + 512));
            uint32_t *q = b_par.as<uint32_t>();
            pr.tid = q; q += Ev; pr.start = q; q += Ev; pr.ilen_cls = q; q += Ev; pr.ts = q; q += Ev; pr.te = q; q += Ev;
            pr.count = q; q += Ev; pr.first = q; q += Ev; pr.last = q; q += Ev; ev_flag = q;
            HIP_TRY(hipMemsetAsync(d_sc + 7, 0, 4, st));
            launch_preagg(ev, n_events, pr, d_sc + 7, st);
            HIP_TRY(hipMemcpyAsync(h_sc + 7, d_sc + 7, 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            n_s = h_sc[7];
            memset(&sev, 0, sizeof sev);
            sev.tid = pr.tid; sev.start = pr.start; sev.ilen_cls = pr.ilen_cls; sev.ts = pr.ts; sev.te = pr.te;
        }
        const size_t E = n_s;
        const size_t rtmp = radix_tmp_words(n_s) + scan_tmp_words(n_s) + 64;
        HIP_TRY(b_sort.ensure(E * 4 * 6 + rtmp * 4 + 256));
        uint32_t *q = b_sort.as<uint32_t>();
        perm[0] = q; q += E; perm[1] = q; q += E;
        uint32_t *key[2]; key[0] = q; q += E; key[1] = q; q += E;    // the word being sorted on, carried along with the permutation
        uint32_t *head = q; q += E; uint32_t *seg_excl = q; q += E;
        uint32_t *tmp = q;
        int pc = -1;  // current permutation buffer (-1 = identity)
        // each word is gathered through the current permutation ONCE, then its 8-bit passes stream (key, permutation) pairs: with
        // 10^8 events the per-pass gathers of the plain form miss every cache (29 -> 12 ms on the long-read workload)
        auto sort_word = [&](const uint32_t *word, uint32_t nbits) {
            const uint32_t *kin = word;
            int kc = 0;
            if (pc >= 0) { launch_gather_u32(n_s, word, perm[pc], key[0], st); kin = key[0]; kc = 1; }
            for (uint32_t sh = 0; sh < nbits; sh += 8) {
                const uint32_t bits = std::min<uint32_t>(8, nbits - sh);
                const int nxt = pc < 0 ? 0 : pc ^ 1;
                launch_radix_pass_keyed(kin, key[kc], sh, bits, pc < 0 ? nullptr : perm[pc], perm[nxt], n_s, tmp, st);
                kin = key[kc]; kc ^= 1;
                pc = nxt;
            }
        };
        sort_word(sev.ilen_cls, ilen_bits);
        sort_word(sev.start, 32);
        sort_word(sev.tid, group_bits);
        const uint32_t *sorted = perm[pc];
        launch_heads(sev, sorted, n_s, head, st);
        launch_scan_u32(head, seg_excl, n_s, d_sc + 6, tmp, st);
        HIP_TRY(hipMemcpyAsync(h_sc + 6, d_sc + 6, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        n_unique = h_sc[6];

        const size_t U = n_unique;
        const size_t utmp = radix_tmp_words(n_unique) + 64;
        HIP_TRY(b_uni.ensure(U * 4 * 13 + U + utmp * 4 + 256));
        uint32_t *w = b_uni.as<uint32_t>();
        u.tid = w; w += U; u.start = w; w += U; u.end = w; w += U; u.ts_min = w; w += U; u.te_max = w; w += U; u.count = w; w += U;
        u.first_seen = w; w += U; u.last_seen = w; w += U; u.name_rank = w; w += U;
        uint32_t *head_pos = w; w += U; chrom_rank_rows = w; w += U;
        uint32_t *uperm[2]; uperm[0] = w; w += U; uperm[1] = w; w += U;
        uint32_t *utmp_p = w; w += utmp;
        u.strand = (uint8_t *)w;
        launch_fill_u32(u.ts_min, 0xffffffffu, U, st);
        launch_fill_u32(u.te_max, 0u, U, st);
        if (preagg) {
            launch_fill_u32(u.count, 0u, U, st);
            launch_fill_u32(u.first_seen, 0xffffffffu, U, st);
            launch_fill_u32(u.last_seen, 0u, U, st);
            launch_reduce_partials(pr, sorted, head, seg_excl, n_s, u, st);
            // first-seen naming (junctions_extractor.cc:152-157): rank of the key's first event among all keys -- flags over the EVENTS
            HIP_TRY(hipMemsetAsync(ev_flag, 0, (size_t)n_events * 4, st));
            launch_reduce_finish_partials(ev.strand, n_unique, u, ev_flag, st);
            launch_scan_u32(ev_flag, ev_flag, n_events, nullptr, ev_flag + n_events, st);
            launch_name_rank(n_unique, ev_flag, u, st);
        } else {
            launch_reduce(ev, sorted, head, seg_excl, n_events, u, head_pos, st);
            if (row_map) {
                DevBuf &b_map = c->buf("row_map");
                HIP_TRY(b_map.ensure((E + U) * 4 + 256));
                row_map->ev_urow = b_map.as<uint32_t>(); row_map->urow_pos = row_map->ev_urow + E;
                launch_event_urow(sorted, head, seg_excl, n_events, row_map->ev_urow, st);
            }
            // first-seen naming (junctions_extractor.cc:152-157): rank of the key's first event among all keys
            uint32_t *first_flag = head;       // reuse: head/seg_excl are dead after launch_reduce
            HIP_TRY(hipMemsetAsync(first_flag, 0, E * 4, st));
            launch_reduce_finish(ev, sorted, n_events, n_unique, head_pos, u, first_flag, st);
            launch_scan_u32(first_flag, seg_excl, n_events, nullptr, tmp, st);
            launch_name_rank(n_unique, seg_excl, u, st);
        }

        // output order (junctions_extractor.h:117-140): rank of the group (chrom string order), thick_start, thick_end, name
        uint32_t rk = 0;
        for (uint32_t i = 0; i < n_groups; ++i) rk = std::max(rk, rank_of_group_host[i]);
        DevBuf &b_rank = c->buf("rank");
        HIP_TRY(b_rank.ensure((size_t)n_groups * 4 + 64));
        c->rank_stage.assign(rank_of_group_host, rank_of_group_host + n_groups);      // (outlives the asynchronous copy: every call ends with a sync of the stream)
        HIP_TRY(hipMemcpyAsync(b_rank.p, c->rank_stage.data(), (size_t)n_groups * 4, hipMemcpyHostToDevice, st));
        launch_gather_u32(n_unique, b_rank.as<uint32_t>(), u.tid, chrom_rank_rows, st);
        int upc = -1;
        auto usort = [&](const uint32_t *word, uint32_t nbits) {
            for (uint32_t sh = 0; sh < nbits; sh += 8) {
                const uint32_t bits = std::min<uint32_t>(8, nbits - sh);
                const int nxt = upc < 0 ? 0 : upc ^ 1;
                launch_radix_pass(word, sh, bits, upc < 0 ? nullptr : uperm[upc], uperm[nxt], n_unique, utmp_p, st);
                upc = nxt;
            }
        };
        usort(u.name_rank, std::max<uint32_t>(1, bitlen(n_unique)));
        usort(u.te_max, 32);
        usort(u.ts_min, 32);
        usort(chrom_rank_rows, std::max<uint32_t>(1, bitlen(rk)));
        final_perm = uperm[upc];
        if (row_map) launch_inverse_perm(final_perm, n_unique, row_map->urow_pos, st);
        if (!n_unique) HIP_TRY(hipStreamSynchronize(st));
    }

    if (n_unique) {
        // rows in final order: gathered on the device into one block of ten columns, ONE copy into pinned memory
        const size_t U = n_unique;
        DevBuf &b_out = c->buf("rows_out");
        HIP_TRY(b_out.ensure(U * 40 + 256));
        launch_rows_out(u, final_perm, n_unique, b_out.as<uint32_t>(), st);
        if (sink) {
            DevBuf &b_tab = c->buf("table_dev");
            const size_t bytes = table_block_bytes(U);
            HIP_TRY(b_tab.ensure(bytes + 256));
            launch_rows_table(u, final_perm, n_unique, sink->min_anchor, b_tab.as<uint8_t>(), st);
            // page-locking a block costs ~10 ms, a copy into pageable memory ~2 ms more than one into page-locked memory: the first table of a
            // context (a one-shot process has no second) is pageable, the loop that runs step after step gets its recycled page-locked block
            rgx_junction_table *t = table_alloc(*sink->hdr, U, /*zero=*/false, /*pinned=*/c->tables_made++ > 0);
            if (!t) { (void)hipStreamSynchronize(st); return fail(err, errlen, RGX_ERR_DEVICE, "regtools_amd: no memory for the result table\n"); }
            hipError_t e_ = hipMemcpyAsync(((TableBox *)t)->block, b_tab.p, bytes, hipMemcpyDeviceToHost, st);
            if (e_ == hipSuccess) e_ = hipStreamSynchronize(st);
            if (e_ != hipSuccess) { rgx_table_free(t); return fail(err, errlen, RGX_ERR_DEVICE, "HIP error %s copying the result table\n", hipGetErrorString(e_)); }
            sink->table = t; R.n = U;
            return RGX_OK;
        }
        if (U * 40 > c->pinned_rows_cap) {
            if (c->pinned_rows) (void)hipHostFree(c->pinned_rows);
            c->pinned_rows = nullptr; c->pinned_rows_cap = 0;
            const size_t want = U * 40 + U * 5 + 4096;
            HIP_TRY(hipHostMalloc(&c->pinned_rows, want, hipHostMallocDefault));
            c->pinned_rows_cap = want;
        }
        HIP_TRY(hipMemcpyAsync(c->pinned_rows, b_out.p, U * 40, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        const uint32_t *hp = (const uint32_t *)c->pinned_rows;
        R.cols = hp; R.n = U;
        if (view_only) return RGX_OK;
        auto col = [&](size_t k, std::vector<uint32_t> &dst) { dst.assign(hp + k * U, hp + (k + 1) * U); };
        col(0, R.group); col(1, R.start); col(2, R.end); col(3, R.ts); col(4, R.te); col(5, R.count); col(6, R.name_rank); col(7, R.first_seen); col(8, R.last_seen);
        R.strand.resize(U);
        for (size_t i = 0; i < U; ++i) R.strand[i] = (uint8_t)hp[9 * U + i];
        R.n = U;
    }
    return RGX_OK;
}

// ---- -b: barcode counts per junction ---------------------------------------------------------------------------------------------
// Second group-by, on (output row, barcode of the supporting read) -- barcode_kernels.hip.  The device returns one entry per distinct
// (junction, barcode) with its count and first event; the host puts each junction's distinct barcodes, in first-seen order, into the
// container the reference keeps them in (std::unordered_map<std::string,int>, junctions_extractor.h:58) and reads back its iteration
// order -- the order print_barcodes (h:99-111) writes.  Copies of that map (cc:202, :208, :214, :235) keep node order, bucket count and
// rehash state, so one map fed in first-seen order walks through the same states as the reference's per-read copies.
static int barcode_rows(rgx_ctx *c, const Prep &P, const RowMap &rm, const rgx_extract_params *p, rgx_junction_table *t, char *err, size_t errlen) {
    hipStream_t st = c->stream;
    const double t0 = now_ms();
    const size_t E = P.n_events, U = t->n;
    t->bc_row_begin = (uint64_t *)calloc(U + 1, 8);
    if (!E) { t->bc_count = (uint32_t *)calloc(1, 4); t->bc_str_begin = (uint64_t *)calloc(1, 8); t->bc_text = (char *)calloc(1, 1); t->bc_insert_rank = (uint32_t *)calloc(1, 4); return RGX_OK; }
    uint32_t *d_sc = c->buf("scalars").as<uint32_t>();
    uint32_t *h_sc = (uint32_t *)c->pinned;
    DevBuf &b_bc = c->buf("barcodes");
    const size_t rtmp = radix_tmp_words((uint32_t)E) + scan_tmp_words((uint32_t)E) + 64;
    HIP_TRY(b_bc.ensure(E * (8 + 4 * 4 + 4 * 4 + 8 + 4 * 5) + rtmp * 4 + 512));
    uint8_t *q = b_bc.as<uint8_t>();
    BarcodeEv b;
    b.off = (uint64_t *)q; q += E * 8;
    uint64_t *pair_off = (uint64_t *)q; q += E * 8;
    b.len = (uint32_t *)q; q += E * 4; b.h_lo = (uint32_t *)q; q += E * 4; b.h_hi = (uint32_t *)q; q += E * 4; b.row = (uint32_t *)q; q += E * 4;
    uint32_t *perm[2]; perm[0] = (uint32_t *)q; q += E * 4; perm[1] = (uint32_t *)q; q += E * 4;
    uint32_t *head = (uint32_t *)q; q += E * 4; uint32_t *seg_excl = (uint32_t *)q; q += E * 4;
    uint32_t *pair_row = (uint32_t *)q; q += E * 4; uint32_t *pair_first = (uint32_t *)q; q += E * 4; uint32_t *pair_pos = (uint32_t *)q; q += E * 4;
    uint32_t *pair_len = (uint32_t *)q; q += E * 4; uint32_t *pair_count = (uint32_t *)q; q += E * 4;
    uint32_t *tmp = (uint32_t *)q;
    uint32_t *flags = d_sc + 72;
    HIP_TRY(hipMemsetAsync(flags, 0, 8, st));
    launch_bc_event_keys(P.arena, (uint32_t)E, P.ev.read, P.soa.rec_off, rm.ev_urow, rm.urow_pos, (uint8_t)p->barcode_tag[0], (uint8_t)p->barcode_tag[1], b, flags, st);
    int pc = -1;
    auto sort_word = [&](const uint32_t *word, uint32_t nbits) {
        for (uint32_t sh = 0; sh < nbits; sh += 8) {
            const int nxt = pc < 0 ? 0 : pc ^ 1;
            launch_radix_pass(word, sh, std::min<uint32_t>(8, nbits - sh), pc < 0 ? nullptr : perm[pc], perm[nxt], (uint32_t)E, tmp, st);
            pc = nxt;
        }
    };
    sort_word(b.h_lo, 32); sort_word(b.h_hi, 32);
    sort_word(b.row, std::max<uint32_t>(1, bitlen((uint32_t)std::max<size_t>(U, 1) - 1)));
    const uint32_t *sorted = perm[pc];
    launch_bc_heads(P.arena, b, sorted, (uint32_t)E, head, flags, st);
    launch_scan_u32(head, seg_excl, (uint32_t)E, d_sc + 74, tmp, st);
    HIP_TRY(hipMemcpyAsync(h_sc + 72, d_sc + 72, 12, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (h_sc[72]) return fail(err, errlen, RGX_ERR_FORMAT, "regtools_amd: the %c%c tag of an alignment is not a string (the reference dies on such input)\n\n", p->barcode_tag[0], p->barcode_tag[1]);
    if (h_sc[73]) return fail(err, errlen, RGX_ERR_DEVICE, "regtools_amd: two different barcodes of one junction share a 64-bit hash; not handled\n\n");
    const uint32_t n_pairs = h_sc[74];
    launch_bc_pairs(b, sorted, head, seg_excl, (uint32_t)E, pair_row, pair_first, pair_pos, pair_off, pair_len, st);
    launch_bc_counts(n_pairs, (uint32_t)E, pair_pos, pair_count, st);
    uint32_t *str_begin = head;                       // head / seg_excl are dead after launch_bc_pairs
    launch_scan_u32(pair_len, str_begin, n_pairs, d_sc + 75, tmp, st);
    HIP_TRY(hipMemcpyAsync(h_sc + 75, d_sc + 75, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const size_t text_len = h_sc[75];
    DevBuf &b_txt = c->buf("barcode_text");
    HIP_TRY(b_txt.ensure(text_len + 256));
    launch_bc_gather(P.arena, n_pairs, pair_off, pair_len, str_begin, b_txt.as<uint8_t>(), st);
    std::vector<uint32_t> h_row(n_pairs), h_first(n_pairs), h_count(n_pairs), h_begin(n_pairs), h_len(n_pairs);
    std::vector<char> h_text(text_len + 1);
    HIP_TRY(hipMemcpyAsync(h_row.data(), pair_row, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(h_first.data(), pair_first, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(h_count.data(), pair_count, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(h_begin.data(), str_begin, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(h_len.data(), pair_len, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, st));
    if (text_len) HIP_TRY(hipMemcpyAsync(h_text.data(), b_txt.p, text_len, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));

    // host: container order per junction.  Entries arrive sorted by row (then hash): each row's run is contiguous.
    t->bc_count = (uint32_t *)calloc((size_t)n_pairs + 1, 4);
    t->bc_str_begin = (uint64_t *)calloc((size_t)n_pairs + 1, 8);
    t->bc_text = (char *)malloc(text_len + 1);
    t->bc_insert_rank = (uint32_t *)calloc((size_t)n_pairs + 1, 4);
    std::vector<uint32_t> rank_of(n_pairs);
    std::vector<uint32_t> run_begin(U + 1, 0);
    for (uint32_t k = 0; k < n_pairs; ++k) run_begin[h_row[k] + 1]++;
    for (size_t r = 0; r < U; ++r) run_begin[r + 1] += run_begin[r];
    for (size_t r = 0; r <= U; ++r) t->bc_row_begin[r] = run_begin[r];
    const unsigned n_thr = (unsigned)std::max<size_t>(1, std::min<size_t>(16, U / 256));
    std::vector<std::thread> pool;
    auto work = [&](size_t r0, size_t r1) {
        std::vector<uint32_t> idx;
        for (size_t r = r0; r < r1; ++r) {
            const uint32_t k0 = run_begin[r], k1 = run_begin[r + 1];
            idx.resize(k1 - k0);
            for (uint32_t k = k0; k < k1; ++k) idx[k - k0] = k;
            std::sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) { return h_first[x] < h_first[y]; });      // first-seen order
            for (uint32_t q = 0; q < idx.size(); ++q) rank_of[idx[q]] = q;
            std::unordered_map<std::string, int> m;                                                               // the reference's container
            for (uint32_t k : idx) m.insert(std::pair<std::string, int>(std::string(h_text.data() + h_begin[k], h_len[k]), (int)k));
            uint32_t o = k0;
            for (auto it = m.begin(); it != m.end(); ++it, ++o) { t->bc_count[o] = h_count[(uint32_t)it->second]; t->bc_str_begin[o] = (uint64_t)it->second; /* entry id for now */ t->bc_insert_rank[o] = rank_of[(uint32_t)it->second]; }
        }
    };
    for (unsigned w = 0; w < n_thr; ++w) pool.emplace_back(work, U * w / n_thr, U * (w + 1) / n_thr);
    for (auto &th : pool) th.join();
    // lay the strings out in output order
    uint64_t pos = 0;
    for (uint32_t o = 0; o < n_pairs; ++o) {
        const uint32_t k = (uint32_t)t->bc_str_begin[o];
        t->bc_str_begin[o] = pos;
        memcpy(t->bc_text + pos, h_text.data() + h_begin[k], h_len[k]);
        pos += h_len[k];
    }
    t->bc_str_begin[n_pairs] = pos;
    t->ms_barcodes = now_ms() - t0;
    return RGX_OK;
}

static void chrom_string_ranks(const BamHeader &hdr, std::vector<uint32_t> &rank_of_tid) {
    const size_t n = hdr.names.size();
    std::vector<uint32_t> order(n);
    rank_of_tid.assign(n ? n : 1, 0);
    for (size_t i = 0; i < n; ++i) order[i] = (uint32_t)i;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return hdr.names[a] < hdr.names[b]; });
    uint32_t rk = 0;
    for (size_t i = 0; i < n; ++i) {
        if (i > 0 && hdr.names[order[i]] != hdr.names[order[i - 1]]) ++rk;
        rank_of_tid[order[i]] = rk;
    }
}

static int run_pipeline(rgx_ctx *c, const uint8_t *d_bam_in, const uint8_t *h_bam, size_t bam_len, const uint8_t *bai, size_t bai_len,
                        const rgx_extract_params *p, rgx_junction_table **out, char *err, size_t errlen, const SharedMembers *shared = nullptr) {
    *out = nullptr;
    c->last_rows_valid = false;
    Prep P;
    int rc = prepare_events(c, d_bam_in, h_bam, bam_len, bai, bai_len, p, false, P, err, errlen, nullptr, true, false, shared);
    if (rc != RGX_OK) return rc;
    hipStream_t st = c->stream;
    const int32_t n_ref = (int32_t)P.hdr.names.size();
    std::vector<uint32_t> rank_of_tid;
    chrom_string_ranks(P.hdr, rank_of_tid);
    HostRows R;
    RowMap rm;
    TableSink sink; sink.hdr = &P.hdr; sink.min_anchor = p->min_anchor;
    rc = reduce_events(c, P.ev, P.n_events, std::max<uint32_t>(1, bitlen((uint32_t)std::max(n_ref - 1, 0))), std::min<uint32_t>(32, bitlen(p->max_intron) + 2),
                       rank_of_tid.data(), (uint32_t)std::max(n_ref, 1), R, err, errlen, /*view_only=*/true, p->barcodes ? &rm : nullptr, &sink);
    if (rc != RGX_OK) return rc;
    c->last_rows = R.n; c->last_records = P.n_iterated; c->last_events = P.n_events; c->last_bytes = P.total; c->last_rows_valid = true;
    HIP_TRY(hipEventRecord(c->ev[6], st));
    rgx_junction_table *t = sink.table ? sink.table : table_alloc(P.hdr, 0);
    if (!t) return fail(err, errlen, RGX_ERR_DEVICE, "regtools_amd: no memory for the result table\n");
    if (R.n >= 100000000u) host_sort_rows(t);   // names wider than 8 digits compare as strings upstream
    if (p->barcodes) {
        if (R.n >= 100000000u) { rgx_table_free(t); return fail(err, errlen, RGX_ERR_ARG, "regtools_amd: -b with 10^8 or more junctions is not supported\n"); }
        rc = barcode_rows(c, P, rm, p, t, err, errlen);
        if (rc != RGX_OK) { rgx_table_free(t); return rc; }
    }
    t->n_records = P.n_iterated;
    t->n_events = P.n_events; t->inflated_bytes = P.total; t->compressed_bytes = bam_len; t->n_members = P.n_range; t->framing_sweeps = P.framing_sweeps; t->stream_ended = P.stream_ended ? 1 : 0;
    float ms = 0;
    HIP_TRY(hipEventSynchronize(c->ev[6]));
    (void)hipEventElapsedTime(&ms, c->ev[0], c->ev[1]); t->ms_inflate = ms;
    t->ms_inflate_launch = 0;
    if (c->launch_timed && hipEventSynchronize(c->ev_launch[1]) == hipSuccess && hipEventElapsedTime(&ms, c->ev_launch[0], c->ev_launch[1]) == hipSuccess) t->ms_inflate_launch = ms;
    (void)hipEventElapsedTime(&ms, c->ev[2], c->ev[4]); t->ms_records = ms;
    (void)hipEventElapsedTime(&ms, c->ev[4], c->ev[5]); t->ms_scan = ms;
    (void)hipEventElapsedTime(&ms, c->ev[5], c->ev[6]); t->ms_reduce = ms;
    t->ms_total = now_ms() - P.t_begin;
    // (nothing reads the call's arena any more: one that lost its place to a challenger goes now)
    if (c->arena_retired) { c->arena_retired->release(); delete c->arena_retired; c->arena_retired = nullptr; }
    if (getenv("REGTOOLS_AMD_TRACE")) {
        fprintf(stderr, "[rgx trace] total %.3f ms; device buffers grown so far: %llu allocations, %.1f MB, %.3f ms\n", t->ms_total, (unsigned long long)g_alloc_stats.calls,
                (double)g_alloc_stats.bytes / 1e6, g_alloc_stats.ms);
    }
    *out = t;
    return RGX_OK;
}

extern "C" int rgx_extract_device(rgx_ctx *ctx, const void *d_bam, size_t bam_len, const void *bai, size_t bai_len,
                                  const rgx_extract_params *p, rgx_junction_table **out, char *err, size_t errlen) {
    if (!ctx || !d_bam || !out) return fail(err, errlen, RGX_ERR_ARG, "regtools_amd: bad arguments\n");
    return run_pipeline(ctx, (const uint8_t *)d_bam, nullptr, bam_len, (const uint8_t *)bai, bai_len, p, out, err, errlen);
}

extern "C" int rgx_extract_mem(rgx_ctx *ctx, const void *bam, size_t bam_len, const void *bai, size_t bai_len, const rgx_extract_params *p,
                               rgx_junction_table **out, char *err, size_t errlen) {
    if (!ctx || !bam || !out) return fail(err, errlen, RGX_ERR_ARG, "regtools_amd: bad arguments\n");
    return run_pipeline(ctx, nullptr, (const uint8_t *)bam, bam_len, (const uint8_t *)bai, bai_len, p, out, err, errlen);
}

// rgx_extract_multi's shards: the same call with the member list the caller scanned once (multi.cpp)
int rgx_extract_mem_scanned(rgx_ctx *ctx, const void *bam, size_t bam_len, const void *bai, size_t bai_len, const rgx_extract_params *p,
                            const std::vector<rgx::Member> *members, uint64_t total_inflated, rgx_junction_table **out, char *err, size_t errlen) {
    if (!ctx || !bam || !out) return fail(err, errlen, RGX_ERR_ARG, "regtools_amd: bad arguments\n");
    SharedMembers sm{members, total_inflated};
    return run_pipeline(ctx, nullptr, (const uint8_t *)bam, bam_len, (const uint8_t *)bai, bai_len, p, out, err, errlen, members && !members->empty() ? &sm : nullptr);
}

extern "C" void *rgx_host_alloc(size_t bytes) {
    void *p = nullptr;
    return hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? p : nullptr;
}
extern "C" void rgx_host_free(void *p) { if (p) (void)hipHostFree(p); }

extern "C" int rgx_extract(rgx_ctx *ctx, const char *bam_path, const rgx_extract_params *p, rgx_junction_table **out, char *err, size_t errlen) {
    if (!ctx || !bam_path || !out) return fail(err, errlen, RGX_ERR_ARG, "regtools_amd: bad arguments\n");
    FileBytes bam; std::vector<uint8_t> bai;
    if (!bam.open(bam_path)) return fail(err, errlen, RGX_ERR_OPEN, "%s", kMsgOpen);
    std::string idx;
    int r = find_index(bam_path, idx);
    if (r != 0 || !read_index(idx, bai)) return fail(err, errlen, RGX_ERR_INDEX, "%s", kMsgIndex);
    return run_pipeline(ctx, nullptr, bam.data(), bam.size(), bai.data(), bai.size(), p, out, err, errlen);
}

extern "C" int rgx_k_inflate(const void *d_comp, const rgx_member *d_members, uint32_t n_members, void *d_arena, uint32_t *d_status, void *stream) {
    return rgx_k_inflate_form(0, d_comp, d_members, n_members, d_arena, d_status, stream);
}

extern "C" int rgx_k_inflate_form(int form, const void *d_comp, const rgx_member *d_members, uint32_t n_members, void *d_arena, uint32_t *d_status, void *stream) {
    if (form < 0 || form > 5) return RGX_ERR_ARG;                    // 5 = k_inflate with up to four literals per trip
    static_assert(sizeof(rgx_member) == sizeof(Member), "rgx_member layout");
    // stage entry point: the code-length scratch is a process-lifetime buffer grown on demand
    static void *scratch = nullptr; static size_t scratch_cap = 0;
    const size_t need = inflate_scratch_bytes(n_members);
    if (need > scratch_cap) {
        if (scratch) (void)hipFree(scratch);
        if (hipMalloc(&scratch, need) != hipSuccess) { scratch = nullptr; scratch_cap = 0; return RGX_ERR_DEVICE; }
        scratch_cap = need;
    }
    launch_inflate((const uint8_t *)d_comp, (const Member *)d_members, n_members, (uint8_t *)d_arena, 0, (uint32_t *)scratch, d_status, (hipStream_t)stream, 0, 0, false, form, nullptr, 1, true);
    return hipGetLastError() == hipSuccess ? RGX_OK : RGX_ERR_DEVICE;
}

// ---- multi-shard merge (host half of SURVEY 8e) -----------------------------------------------------------------------------------
extern "C" size_t rgx_table_pack(const rgx_junction_table *t, void *dst, size_t dst_cap) {
    const size_t need = (size_t)t->n * RGX_PACKED_ROW_BYTES;
    if (!dst || dst_cap < need) return need;
    uint8_t *q = (uint8_t *)dst;
    for (uint64_t i = 0; i < t->n; ++i, q += RGX_PACKED_ROW_BYTES) {
        uint32_t w[12] = {(uint32_t)t->tid[i], t->start[i], t->end[i], t->thick_start[i], t->thick_end[i], t->read_count[i],
                          (uint32_t)t->first_seen[i], (uint32_t)(t->first_seen[i] >> 32), (uint32_t)t->last_seen[i], (uint32_t)(t->last_seen[i] >> 32),
                          (uint32_t)(uint8_t)t->strand[i], (uint32_t)t->name_index[i]};
        memcpy(q, w, sizeof w);
    }
    return need;
}

extern "C" int rgx_table_unpack(const void *src, size_t n_rows, const rgx_junction_table *names_from, rgx_junction_table **out) {
    BamHeader h;
    for (int32_t i = 0; i < names_from->n_ref; ++i) { h.names.push_back(names_from->ref_name[i]); h.lens.push_back(names_from->ref_len[i]); }
    rgx_junction_table *t = table_alloc(h, n_rows);
    if (!t) return RGX_ERR_ARG;
    const uint8_t *q = (const uint8_t *)src;
    for (size_t i = 0; i < n_rows; ++i, q += RGX_PACKED_ROW_BYTES) {
        uint32_t w[12]; memcpy(w, q, sizeof w);
        t->tid[i] = (int32_t)w[0]; t->start[i] = w[1]; t->end[i] = w[2]; t->thick_start[i] = w[3]; t->thick_end[i] = w[4]; t->read_count[i] = w[5];
        t->first_seen[i] = (uint64_t)w[6] | (uint64_t)w[7] << 32; t->last_seen[i] = (uint64_t)w[8] | (uint64_t)w[9] << 32; t->strand[i] = (char)w[10];
        t->name_index[i] = w[11];
    }
    *out = t;
    return RGX_OK;
}

// The barcode lists of a table as one byte block (the wire format of the one-process-per-GPU driver, next to the 48-byte rows):
// u64 rows, u64 entries, u64 text bytes, then bc_row_begin (rows + 1 u64), bc_str_begin (entries + 1 u64), bc_count and bc_insert_rank
// (entries u32 each), the text.  Returns the bytes needed; writes when dst_cap suffices.  0 = the table carries no barcode lists.
extern "C" size_t rgx_table_pack_barcodes(const rgx_junction_table *t, void *dst, size_t dst_cap) {
    if (!t || !t->bc_row_begin) return 0;
    const uint64_t n = t->n, E = t->bc_row_begin[n], T = t->bc_str_begin ? t->bc_str_begin[E] : 0;
    const size_t need = 24 + (size_t)(n + 1) * 8 + (size_t)(E + 1) * 8 + (size_t)E * 8 + (size_t)T;
    if (!dst || dst_cap < need) return need;
    uint8_t *q = (uint8_t *)dst;
    const uint64_t head[3] = {n, E, T};
    memcpy(q, head, 24); q += 24;
    memcpy(q, t->bc_row_begin, (size_t)(n + 1) * 8); q += (size_t)(n + 1) * 8;
    memcpy(q, t->bc_str_begin, (size_t)(E + 1) * 8); q += (size_t)(E + 1) * 8;
    memcpy(q, t->bc_count, (size_t)E * 4); q += (size_t)E * 4;
    memcpy(q, t->bc_insert_rank, (size_t)E * 4); q += (size_t)E * 4;
    memcpy(q, t->bc_text, (size_t)T);
    return need;
}

// ... and back, onto a table of the same rows (rgx_table_unpack of the shard's packed rows).  Every offset is checked: the block crossed a wire.
extern "C" int rgx_table_unpack_barcodes(rgx_junction_table *t, const void *src, size_t len) {
    if (!t || !src || len < 24) return RGX_ERR_ARG;
    const uint8_t *q = (const uint8_t *)src;
    uint64_t head[3]; memcpy(head, q, 24); q += 24;
    const uint64_t n = head[0], E = head[1], T = head[2];
    if (n != t->n || E > (len >> 3) || T > len) return RGX_ERR_ARG;
    const size_t need = 24 + (size_t)(n + 1) * 8 + (size_t)(E + 1) * 8 + (size_t)E * 8 + (size_t)T;
    if (len < need) return RGX_ERR_ARG;
    uint64_t *row_begin = (uint64_t *)calloc((size_t)n + 1, 8), *str_begin = (uint64_t *)calloc((size_t)E + 1, 8);
    uint32_t *count = (uint32_t *)calloc((size_t)E + 1, 4), *rank = (uint32_t *)calloc((size_t)E + 1, 4);
    char *text = (char *)malloc((size_t)T + 1);
    bool ok = row_begin && str_begin && count && rank && text;
    if (ok) {
        memcpy(row_begin, q, (size_t)(n + 1) * 8); q += (size_t)(n + 1) * 8;
        memcpy(str_begin, q, (size_t)(E + 1) * 8); q += (size_t)(E + 1) * 8;
        memcpy(count, q, (size_t)E * 4); q += (size_t)E * 4;
        memcpy(rank, q, (size_t)E * 4); q += (size_t)E * 4;
        memcpy(text, q, (size_t)T);
        ok = row_begin[0] == 0 && row_begin[n] == E && str_begin[0] == 0 && str_begin[E] == T;
        for (uint64_t i = 0; ok && i < n; ++i) ok = row_begin[i] <= row_begin[i + 1];
        for (uint64_t k = 0; ok && k < E; ++k) ok = str_begin[k] <= str_begin[k + 1];
        // a row's ranks are a permutation of 0 .. (its entries - 1): rgx_table_merge_barcodes indexes by them
        for (uint64_t i = 0; ok && i < n; ++i) {
            const uint64_t b = row_begin[i], e = row_begin[i + 1];
            std::vector<uint8_t> seen((size_t)(e - b), 0);
            for (uint64_t k = b; ok && k < e; ++k) { ok = rank[k] < e - b && !seen[rank[k]]; if (ok) seen[rank[k]] = 1; }
        }
    }
    if (!ok) { free(row_begin); free(str_begin); free(count); free(rank); free(text); return RGX_ERR_ARG; }
    free(t->bc_row_begin); free(t->bc_count); free(t->bc_str_begin); free(t->bc_text); free(t->bc_insert_rank);
    t->bc_row_begin = row_begin; t->bc_str_begin = str_begin; t->bc_count = count; t->bc_insert_rank = rank; t->bc_text = text;
    return RGX_OK;
}

// -b across shards.  A junction's barcode map (junctions_extractor.h:58, cc:204-217) only depends on the sequence in which DISTINCT barcodes
// first reach it (a repeat bumps a count, it never moves a node): shard order is file order and bc_insert_rank keeps the order inside a shard,
// so the merged junction's sequence is the shards' sequences one after the other minus the barcodes already seen -- fed, as in barcode_rows, to
// the container the reference keeps, whose iteration order is the order print_barcodes writes (h:99-111).
extern "C" int rgx_table_merge_barcodes(const rgx_junction_table *const *parts, int n_parts, rgx_junction_table *t, char *err, size_t errlen) {
    if (!parts || n_parts <= 0 || !t) return fail(err, errlen, RGX_ERR_ARG, "regtools_amd: nothing to merge\n");
    for (int g = 0; g < n_parts; ++g) if (!parts[g] || !parts[g]->bc_row_begin) return fail(err, errlen, RGX_ERR_ARG, "regtools_amd: shard %d carries no barcode counts\n", g);
    auto cls = [](char c) { return c == '+' ? 0u : c == '-' ? 1u : 2u; };
    struct Key { int32_t tid; uint32_t start, end, cls; bool operator==(const Key &o) const { return tid == o.tid && start == o.start && end == o.end && cls == o.cls; } };
    struct KeyHash { size_t operator()(const Key &k) const { uint64_t h = (uint64_t)(uint32_t)k.tid * 0x9e3779b97f4a7c15ull ^ ((uint64_t)k.start << 32 | k.end) * 0xc2b2ae3d27d4eb4full ^ k.cls; return (size_t)(h ^ h >> 29); } };
    std::unordered_map<Key, uint64_t, KeyHash> row_of;
    row_of.reserve((size_t)t->n * 2 + 16);
    for (uint64_t i = 0; i < t->n; ++i) row_of[Key{t->tid[i], t->start[i], t->end[i], cls(t->strand[i])}] = i;
    struct Ent { const char *s; uint32_t len; uint32_t count; };
    std::vector<std::vector<Ent>> per_row((size_t)t->n);
    // a junction of a single-cell library carries thousands of barcodes, times the shards: rows that grow past a few entries get an index
    // (barcode bytes -> entry) instead of the linear scan the common few-barcode rows keep
    struct SvHash { size_t operator()(const std::pair<const char *, uint32_t> &k) const { uint64_t h = 1469598103934665603ull; for (uint32_t i = 0; i < k.second; ++i) h = (h ^ (uint8_t)k.first[i]) * 1099511628211ull; return (size_t)h; } };
    struct SvEq { bool operator()(const std::pair<const char *, uint32_t> &a, const std::pair<const char *, uint32_t> &b) const { return a.second == b.second && !memcmp(a.first, b.first, a.second); } };
    typedef std::unordered_map<std::pair<const char *, uint32_t>, uint32_t, SvHash, SvEq> RowIndex;
    std::unordered_map<uint64_t, RowIndex> row_index;
    constexpr size_t kScanRows = 16;
    std::vector<uint64_t> order;
    for (int g = 0; g < n_parts; ++g) {
        const rgx_junction_table *p = parts[g];
        for (uint64_t i = 0; i < p->n; ++i) {
            auto it = row_of.find(Key{p->tid[i], p->start[i], p->end[i], cls(p->strand[i])});
            if (it == row_of.end()) return fail(err, errlen, RGX_ERR_ARG, "regtools_amd: a shard row is missing from the merged table (shard %d row %llu: tid %d %u-%u '%c'; merged rows %llu)\n", g, (unsigned long long)i, p->tid[i], p->start[i], p->end[i], p->strand[i], (unsigned long long)t->n);
            std::vector<Ent> &dst = per_row[(size_t)it->second];
            const uint64_t b = p->bc_row_begin[i], e = p->bc_row_begin[i + 1];
            order.assign(e - b, 0);
            for (uint64_t k = b; k < e; ++k) order[p->bc_insert_rank[k]] = k;                     // the shard's first-seen order
            for (uint64_t k : order) {
                const char *str = p->bc_text + p->bc_str_begin[k];
                const uint32_t len = (uint32_t)(p->bc_str_begin[k + 1] - p->bc_str_begin[k]);
                bool found = false;
                if (dst.size() > kScanRows) {
                    RowIndex &ix = row_index[it->second];
                    if (ix.empty()) for (uint32_t q = 0; q < dst.size(); ++q) ix.emplace(std::make_pair(dst[q].s, dst[q].len), q);     // (the row just outgrew the scan)
                    auto f = ix.find(std::make_pair(str, len));
                    if (f != ix.end()) { dst[f->second].count += p->bc_count[k]; found = true; }
                    else ix.emplace(std::make_pair(str, len), (uint32_t)dst.size());
                } else for (Ent &x : dst) if (x.len == len && !memcmp(x.s, str, len)) { x.count += p->bc_count[k]; found = true; break; }
                if (!found) dst.push_back(Ent{str, len, p->bc_count[k]});
            }
        }
        if (p->stream_ended) break;          // upstream reads nothing behind the point where the record stream ended
    }
    size_t n_pairs = 0, text_len = 0;
    for (auto &v : per_row) { n_pairs += v.size(); for (auto &x : v) text_len += x.len; }
    free(t->bc_row_begin); free(t->bc_count); free(t->bc_str_begin); free(t->bc_text); free(t->bc_insert_rank);
    t->bc_row_begin = (uint64_t *)calloc((size_t)t->n + 1, 8);
    t->bc_count = (uint32_t *)calloc(n_pairs + 1, 4);
    t->bc_str_begin = (uint64_t *)calloc(n_pairs + 1, 8);
    t->bc_text = (char *)malloc(text_len + 1);
    t->bc_insert_rank = (uint32_t *)calloc(n_pairs + 1, 4);
    uint64_t o = 0, pos = 0;
    for (uint64_t r = 0; r < t->n; ++r) {
        t->bc_row_begin[r] = o;
        const std::vector<Ent> &v = per_row[(size_t)r];
        std::unordered_map<std::string, int> m;                                                    // the reference's container
        for (size_t k = 0; k < v.size(); ++k) m.insert(std::pair<std::string, int>(std::string(v[k].s, v[k].len), (int)k));
        for (auto it = m.begin(); it != m.end(); ++it, ++o) {
            const Ent &x = v[(size_t)it->second];
            t->bc_count[o] = x.count; t->bc_insert_rank[o] = (uint32_t)it->second; t->bc_str_begin[o] = pos;
            memcpy(t->bc_text + pos, x.s, x.len); pos += x.len;
        }
    }
    t->bc_row_begin[t->n] = o; t->bc_str_begin[n_pairs] = pos;
    return RGX_OK;
}

extern "C" int rgx_table_merge(const rgx_junction_table *const *parts, int n_parts, uint32_t min_anchor, rgx_junction_table **out, char *err, size_t errlen) {
    if (n_parts <= 0 || !parts || !parts[0]) return fail(err, errlen, RGX_ERR_ARG, "regtools_amd: nothing to merge\n");
    struct Row { int32_t tid; uint32_t start, end, ts, te, cnt; uint64_t first, last; char strand; };
    auto cls = [](char c) { return c == '+' ? 0 : c == '-' ? 1 : 2; };
    const bool trace = getenv("REGTOOLS_AMD_TRACE") != nullptr; double tl = now_ms();
    auto mark = [&](const char *w) { if (trace) { double t = now_ms(); fprintf(stderr, "[rgx trace] merge %-14s %8.3f ms\n", w, t - tl); tl = t; } };
    std::vector<Row> rows;
    for (int g = 0; g < n_parts; ++g) {
        const rgx_junction_table *t = parts[g];
        for (uint64_t i = 0; i < t->n; ++i)
            rows.push_back({t->tid[i], t->start[i], t->end[i], t->thick_start[i], t->thick_end[i], t->read_count[i],
                            (uint64_t)g << 40 | t->first_seen[i], (uint64_t)g << 40 | t->last_seen[i], t->strand[i]});
        if (t->stream_ended) break;          // the record stream ended inside this shard: upstream reads nothing behind that point
    }
    mark("collect");
    std::stable_sort(rows.begin(), rows.end(), [&](const Row &a, const Row &b) {
        if (a.tid != b.tid) return a.tid < b.tid;
        if (a.start != b.start) return a.start < b.start;
        if (a.end != b.end) return a.end < b.end;
        return cls(a.strand) < cls(b.strand);
    });
    mark("key sort");
    std::vector<Row> uq;
    for (const Row &r : rows) {
        if (!uq.empty() && uq.back().tid == r.tid && uq.back().start == r.start && uq.back().end == r.end && cls(uq.back().strand) == cls(r.strand)) {
            Row &m = uq.back();
            m.cnt += r.cnt; m.ts = std::min(m.ts, r.ts); m.te = std::max(m.te, r.te);
            if (r.first < m.first) m.first = r.first;
            if (r.last > m.last) { m.last = r.last; m.strand = r.strand; }
        } else uq.push_back(r);
    }
    mark("reduce");
    std::vector<size_t> by_first(uq.size());
    for (size_t i = 0; i < uq.size(); ++i) by_first[i] = i;
    std::sort(by_first.begin(), by_first.end(), [&](size_t a, size_t b) { return uq[a].first < uq[b].first; });
    BamHeader h;
    for (int32_t i = 0; i < parts[0]->n_ref; ++i) { h.names.push_back(parts[0]->ref_name[i]); h.lens.push_back(parts[0]->ref_len[i]); }
    rgx_junction_table *t = table_alloc(h, uq.size());
    if (!t) return fail(err, errlen, RGX_ERR_ARG, "regtools_amd: no memory for the result table\n");
    for (size_t k = 0; k < by_first.size(); ++k) {
        const Row &r = uq[by_first[k]];
        const size_t i = by_first[k];
        t->tid[i] = r.tid; t->start[i] = r.start; t->end[i] = r.end; t->thick_start[i] = r.ts; t->thick_end[i] = r.te; t->read_count[i] = r.cnt;
        t->name_index[i] = k + 1; t->strand[i] = r.strand; t->first_seen[i] = r.first; t->last_seen[i] = r.last;
        t->left_ok[i] = (uint32_t)(r.start - r.ts) >= min_anchor; t->right_ok[i] = (uint32_t)(r.te - r.end) >= min_anchor;
    }
    mark("name+fill");
    host_sort_rows(t);
    mark("order sort");
    for (int g = 0; g < n_parts; ++g) {
        t->n_records += parts[g]->n_records; t->n_events += parts[g]->n_events; t->inflated_bytes += parts[g]->inflated_bytes;
        t->compressed_bytes = parts[g]->compressed_bytes; t->n_members += parts[g]->n_members;
    }
    bool all_bc = true;
    for (int g = 0; g < n_parts; ++g) if (!parts[g]->bc_row_begin) all_bc = false;
    if (all_bc) { const int rc = rgx_table_merge_barcodes(parts, n_parts, t, err, errlen); if (rc != RGX_OK) { rgx_table_free(t); return rc; } }
    *out = t;
    return RGX_OK;
}

// Peer access between two devices, both directions, once per pair and process (round 4): without it hipMemcpyPeer* between two GPUs is a bounce
// through host memory instead of a copy over xGMI.  Returns whether the pair is peer-accessible (a copy still works when it is not).
bool rgx_enable_peer(int a, int b) {
    if (a == b) return true;
    static std::mutex mu; static std::map<std::pair<int, int>, bool> done;
    std::lock_guard<std::mutex> lk(mu);
    const std::pair<int, int> key{std::min(a, b), std::max(a, b)};
    auto it = done.find(key);
    if (it != done.end()) return it->second;
    int prev = 0; (void)hipGetDevice(&prev);
    bool ok = true;
    for (int dir = 0; dir < 2; ++dir) {
        const int self = dir ? b : a, peer = dir ? a : b;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, self, peer) != hipSuccess || !can) { ok = false; continue; }
        if (hipSetDevice(self) != hipSuccess) { ok = false; continue; }
        const hipError_t e = hipDeviceEnablePeerAccess(peer, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) ok = false;
    }
    (void)hipSetDevice(prev);
    (void)hipGetLastError();                                        // ("already enabled" is not an error of anybody's launch)
    done[key] = ok;
    return ok;
}

// the rows of the context's last extraction, packed (48 bytes per row) into a buffer of the CONTEXT on its own stream, no host wait: *done is
// recorded behind the kernel, for the exchange stream of rgx_extract_multi to wait on (multi.cpp; not part of the C ABI)
int rgx_last_table_pack_async(rgx_ctx *c, const rgx_junction_table *t, void **d_packed, hipEvent_t *done, char *err, size_t errlen) {
    if (!c || !t || !d_packed || !done) return fail(err, errlen, RGX_ERR_ARG, "regtools_amd: bad arguments\n");
    if (!c->last_rows_valid || t->n != c->last_rows || t->n_records != c->last_records || t->n_events != c->last_events || t->inflated_bytes != c->last_bytes)
        return fail(err, errlen, RGX_ERR_ARG, "regtools_amd: the table is not the result of the last extraction on this context\n");
    HIP_ENTER(c->device);
    DevBuf &b = c->buf("rows_packed");
    HIP_TRY(b.ensure((size_t)std::max<uint64_t>(1, t->n) * RGX_PACKED_ROW_BYTES));
    if (!c->ev_packed) HIP_TRY(hipEventCreateWithFlags(&c->ev_packed, hipEventDisableTiming));
    if (t->n) launch_cols_to_packed(c->buf("rows_out").as<uint32_t>(), (uint32_t)t->n, b.as<uint32_t>(), c->stream);
    HIP_TRY(hipEventRecord(c->ev_packed, c->stream));
    *d_packed = b.p; *done = c->ev_packed;
    return RGX_OK;
}

// the rows of the context's last extraction, packed for the all-gather without leaving HBM
extern "C" int rgx_last_table_pack_device(rgx_ctx *c, const rgx_junction_table *t, void *d_dst, uint64_t cap_rows, char *err, size_t errlen) {
    if (!c || !t) return fail(err, errlen, RGX_ERR_ARG, "regtools_amd: bad arguments\n");
    if (!c->last_rows_valid || t->n != c->last_rows || t->n_records != c->last_records || t->n_events != c->last_events || t->inflated_bytes != c->last_bytes)
        return fail(err, errlen, RGX_ERR_ARG, "regtools_amd: the table is not the result of the last extraction on this context\n");
    if (!t->n) return RGX_OK;
    if (!d_dst || cap_rows < t->n) return fail(err, errlen, RGX_ERR_ARG, "regtools_amd: destination holds %llu rows, %llu needed\n", (unsigned long long)cap_rows, (unsigned long long)t->n);
    HIP_ENTER(c->device);
    launch_cols_to_packed(c->buf("rows_out").as<uint32_t>(), (uint32_t)t->n, (uint32_t *)d_dst, c->stream);
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RGX_OK;
}

// ---- multi-shard merge on the device (the gathered packed rows never leave HBM until the merged table is final) ----------------------
extern "C" int rgx_table_merge_device(rgx_ctx *c, const void *d_rows, uint64_t stride_rows, const uint64_t *part_rows, int n_parts, uint32_t min_anchor,
                                      const rgx_junction_table *names_from, rgx_junction_table **out, char *err, size_t errlen) {
    if (!c || !d_rows || !part_rows || n_parts <= 0 || n_parts > 255 || !names_from || !out) return fail(err, errlen, RGX_ERR_ARG, "regtools_amd: bad arguments\n");
    *out = nullptr;
    HIP_ENTER(c->device);
    hipStream_t st = c->stream;
    const bool trace = getenv("REGTOOLS_AMD_TRACE") != nullptr;
    double t_last = now_ms();
    auto mark = [&](const char *what) { if (trace) { (void)hipStreamSynchronize(st); double t = now_ms(); fprintf(stderr, "[rgx trace] merge: %-24s +%8.3f ms\n", what, t - t_last); t_last = t; } };
    std::vector<uint32_t> h_rows((size_t)n_parts), h_base((size_t)n_parts);
    uint64_t total = 0;
    for (int g = 0; g < n_parts; ++g) {
        if (part_rows[g] > stride_rows || part_rows[g] >= (1u << 24)) return fail(err, errlen, RGX_ERR_ARG, "regtools_amd: shard table too large for the device merge\n");
        h_rows[(size_t)g] = (uint32_t)part_rows[g]; h_base[(size_t)g] = (uint32_t)total; total += part_rows[g];
    }
    if (total >= (1ull << 31) || stride_rows * (uint64_t)n_parts >= (1ull << 32)) return fail(err, errlen, RGX_ERR_ARG, "regtools_amd: too many rows for the device merge\n");
    BamHeader h;
    for (int32_t i = 0; i < names_from->n_ref; ++i) { h.names.push_back(names_from->ref_name[i]); h.lens.push_back(names_from->ref_len[i]); }
    const uint32_t N = (uint32_t)total;
    if (!N) { *out = table_alloc(h, 0); return *out ? RGX_OK : RGX_ERR_DEVICE; }
    std::vector<uint32_t> rank_of_tid;
    chrom_string_ranks(h, rank_of_tid);
    uint32_t rk = 0; for (uint32_t r : rank_of_tid) rk = std::max(rk, r);

    DevBuf &b = c->buf("merge"), &sc = c->buf("scalars");
    HIP_TRY(sc.ensure(512));
    const size_t Nn = N, P = (size_t)n_parts, R = rank_of_tid.size();
    const size_t tmp_words = radix_tmp_words(N) + scan_tmp_words(N) + 64;
    HIP_TRY(b.ensure((Nn * (10 + 2 + 2 + 9 + 4 + 13) + 64 + 2 * P + R + tmp_words) * 4 + 1024));
    uint32_t *w = b.as<uint32_t>();
    MergeSoA m; m.tid = w; w += Nn; m.start = w; w += Nn; m.end = w; w += Nn; m.ts = w; w += Nn; m.te = w; w += Nn; m.count = w; w += Nn;
    m.cls = w; w += Nn; m.first = w; w += Nn; m.shard = w; w += Nn; m.strand = w; w += Nn;
    uint32_t *perm[2] = {w, w + Nn}; w += 2 * Nn;
    uint32_t *head = w; w += Nn; uint32_t *seg = w; w += Nn;
    MergeUnique u; u.tid = w; w += Nn; u.start = w; w += Nn; u.end = w; w += Nn; u.ts = w; w += Nn; u.te = w; w += Nn; u.count = w; w += Nn;
    u.first = w; w += Nn; u.last_shard = w; w += Nn; u.strand = w; w += Nn;
    uint32_t *name_rank = w; w += Nn; uint32_t *crank = w; w += Nn; uint32_t *uperm[2] = {w, w + Nn}; w += 2 * Nn;
    uint32_t *packed = w; w += Nn * 13 + 64;          // the merged table's device image (51 bytes per row + padding)
    uint32_t *d_rows_n = w; w += P; uint32_t *d_base = w; w += P; uint32_t *d_rank = w; w += R;
    uint32_t *tmp = w;
    uint32_t *d_total = sc.as<uint32_t>() + 70;
    HIP_TRY(hipMemcpyAsync(d_rows_n, h_rows.data(), P * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_base, h_base.data(), P * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_rank, rank_of_tid.data(), R * 4, hipMemcpyHostToDevice, st));
    launch_merge_unpack((const uint32_t *)d_rows, (uint32_t)stride_rows, (uint32_t)n_parts, d_rows_n, d_base, m, st);
    // stable LSD radix sort by (tid, start, end, class); rows of one key end up in shard order
    int pc = -1;
    auto sort_word = [&](const uint32_t *word, uint32_t nbits, uint32_t n, uint32_t **pp, int &cur) {
        for (uint32_t sh = 0; sh < nbits; sh += 8) {
            const uint32_t bits = std::min<uint32_t>(8, nbits - sh);
            const int nxt = cur < 0 ? 0 : cur ^ 1;
            launch_radix_pass(word, sh, bits, cur < 0 ? nullptr : pp[cur], pp[nxt], n, tmp, st);
            cur = nxt;
        }
    };
    sort_word(m.cls, 2, N, perm, pc);
    sort_word(m.end, 32, N, perm, pc);
    sort_word(m.start, 32, N, perm, pc);
    sort_word(m.tid, std::max<uint32_t>(1, bitlen((uint32_t)std::max<int32_t>(1, names_from->n_ref))), N, perm, pc);
    const uint32_t *sorted = perm[pc];
    mark("unpack + key sort");
    launch_merge_heads(m, sorted, N, head, st);
    launch_scan_u32(head, seg, N, d_total, tmp, st);
    uint32_t U = 0;
    HIP_TRY(hipMemcpyAsync(&U, d_total, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    launch_fill_u32(u.ts, 0xffffffffu, U, st); launch_fill_u32(u.first, 0xffffffffu, U, st);
    launch_fill_u32(u.te, 0u, U, st); launch_fill_u32(u.count, 0u, U, st); launch_fill_u32(u.last_shard, 0u, U, st);
    launch_merge_reduce(m, sorted, head, seg, N, u, st);
    // first-seen naming: rank by (first shard that has the key, the row's rank inside that shard)
    int upc = -1;
    sort_word(u.first, 32, U, uperm, upc);
    launch_merge_rank(uperm[upc], U, name_rank, st);
    // output order (junctions_extractor.h:117-140): chrom string rank, thick_start, thick_end, name
    launch_gather_u32(U, d_rank, u.tid, crank, st);
    upc = -1;
    sort_word(name_rank, std::max<uint32_t>(1, bitlen(U)), U, uperm, upc);
    sort_word(u.te, 32, U, uperm, upc);
    sort_word(u.ts, 32, U, uperm, upc);
    sort_word(crank, std::max<uint32_t>(1, bitlen(rk)), U, uperm, upc);
    launch_merge_table(u, uperm[upc], name_rank, U, min_anchor, (uint8_t *)packed, st);      // the packed area doubles as the table's device image
    mark("reduce + name + order");
    rgx_junction_table *t = table_alloc(h, U, /*zero=*/false, /*pinned=*/true);
    if (!t) { (void)hipStreamSynchronize(st); return fail(err, errlen, RGX_ERR_DEVICE, "regtools_amd: no memory for the result table\n"); }
    {
        hipError_t e_ = hipMemcpyAsync(((TableBox *)t)->block, packed, table_block_bytes(U), hipMemcpyDeviceToHost, st);
        if (e_ == hipSuccess) e_ = hipStreamSynchronize(st);
        if (e_ != hipSuccess) { rgx_table_free(t); return fail(err, errlen, RGX_ERR_DEVICE, "HIP error %s copying the merged table\n", hipGetErrorString(e_)); }
    }
    mark("rows to host table");
    *out = t;
    return RGX_OK;
}

#include "cse_api.inc"
