// api_internal.h -- what the translation units behind the C ABI (include/regtools_amd.h) share: the context, its device buffers, one call's state (EventsRun)
// and the functions one unit calls in another.  Round 6: api.cpp (2,600 lines, with cse_api.inc included into it) split along its stages --
//   api_ctx.cpp      contexts, streams, result tables and their text
//   api_front.cpp    EventsRun: upload, member list, the DEFLATE launch (+ the opt-in arena placement trials)
//   api_records.cpp  EventsRun: footers and header, record framing, decode, emit
//   api_reduce.cpp   group-by, output order, barcodes, the finished table
//   api_entry.cpp    the extract entry points, packing and merging of tables
//   cse_api.cpp      identify / associate / annotate (rows a9-a12, f2, f3)
#pragma once
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

inline int fail(char *err, size_t errlen, int code, const char *fmt, ...) {
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
        if (e_ != hipSuccess)                                                                               \
            return fail(err, errlen, RGX_ERR_DEVICE, "HIP error %s from a kernel launch before %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// First HIP call of an entry point: errors other code left with this host thread (another library's polling of an event, an ignored return
// of a clean-up call) are not this call's launches' -- drop them, then select the device.
#define HIP_ENTER(dev)                                                                                      \
    do { (void)rgx::pending_launch_error(); HIP_TRY(hipSetDevice(dev)); } while (0)

static const char *const kMsgOpen = "Unable to open BAM/SAM file.\n\n";
static const char *const kMsgIndex = "Unable to open BAM/SAM index. Make sure alignments are indexed\n\n";
static const char *const kMsgRegion = "Unable to iterate to region within BAM.\n\n";

inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

inline uint32_t bitlen(uint32_t v) { uint32_t b = 0; while (v) { ++b; v >>= 1; } return b; }

// growable device buffer that survives across calls (workspace reuse: no hipMalloc in steady state)
// Every buffer starts kFront bytes into its allocation: k_inflate_ring's far copies load their source from up to 15 bytes in front of it
// (inflate_ring.h, kArenaFrontPad), and for the first member of an arena that is in front of the buffer.
struct AllocStats { double ms = 0; uint64_t calls = 0, bytes = 0; };
extern AllocStats g_alloc_stats;                                   // (REGTOOLS_AMD_TRACE: what growing the device buffers cost a call)
// REGTOOLS_AMD_ARENA="trials[,piece_MiB]": how many fresh arenas a context's first large call times its DEFLATE launch into (default 0 = none: the trials are
// OPT-IN since round 6 -- on the driver's box five of them bought 14.5 -> 14.0 ms for 0.3 s and an arena's worth of transient memory; calibrate_arena),
// and the size of the pieces the arena's device memory is created in (default 512, 0 = one hipMalloc block; DevBuf::map_pieces).
struct ArenaKnobs { int trials = 0; size_t piece = (size_t)512 << 20; };
inline const ArenaKnobs &arena_knobs() {
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
    // asked for by the owner (the arena): memory created in pieces of this size and mapped side by side, see map_pieces (0: one hipMalloc block)
    size_t piece = 0;
    size_t mapped = 0;              // bytes of the reserved address range the pieces are mapped into, which may be longer than they are (0: a hipMalloc block)
    std::vector<size_t> piece_len;  // the mappings inside that range, in address order (each is unmapped on its own)
    int range_dev = -1;             // the device the range's pieces were created on
    // The arena's form (round 5).  The DEFLATE launch writes 169,000 streams 64 KB apart at once, and what it costs depends on the memory under
    // them: 13.9-15.9 ms
    // into one hipMalloc block of 11 GB, 12.3-12.6 ms into the same bytes created as pieces of 1 GiB (hipMemCreate) and mapped side by side into one reserved
    // address range --
    // whatever the order of the pieces (profiles/r05_inflate_arena_pieces.txt: forty pieces, 110 subsets, 12.30-12.37 ms).  Pieces of 2 MiB: 16.0 ms; 32 MiB:
    // 12.8-13.7;
    // 256 MiB: 12.4-13.7.  A runtime that refuses any of the calls leaves the buffer to hipMalloc.
    // Reserved address ranges are KEPT for the next arena, never handed back (round 6, fifth session): hipMemAddressFree of a range whose pieces had all
    // been unmapped -- behind a hipDeviceSynchronize -- died of a null pointer inside the runtime (SIGSEGV at address 0x68 in libamdhip64, ROCm 7.0.2) once
    // in ~150 calls when ranges were reserved, mapped, unmapped and freed in quick succession (the arena placement trials:
    // tests/test_gpu_parity.py::test_arena_placement_trials_leave_the_results_alone failed 3-5 % of its runs; backtraces in
    // profiles/r06_s5_addressfree_segv.txt).  A range holds no memory once its pieces are unmapped, only addresses; map_pieces takes the smallest kept
    // range of its device that is large enough before it reserves a new one.
    struct KeptRange { void *base; size_t len; int dev; };
    struct KeptRanges { std::mutex mu; std::vector<KeptRange> v; };
    // (never destroyed: contexts are released by static destructors too)
    static KeptRanges &kept_ranges() { static KeptRanges *k = new KeptRanges(); return *k; }
    static void *take_range(int dev, size_t total, size_t *reserved) {           // (a range goes back to the device whose pieces it held)
        KeptRanges &k = kept_ranges();
        std::lock_guard<std::mutex> lock(k.mu);
        size_t best = SIZE_MAX;
        for (size_t i = 0; i < k.v.size(); ++i) if (k.v[i].dev == dev && k.v[i].len >= total && (best == SIZE_MAX || k.v[i].len < k.v[best].len)) best = i;
        if (best == SIZE_MAX) return nullptr;
        void *base = k.v[best].base; *reserved = k.v[best].len;
        k.v.erase(k.v.begin() + (long)best);
        return base;
    }
    static void unmap_range(void *base, const std::vector<size_t> &lens, size_t reserved, int dev) {
        static const bool trace = getenv("REGTOOLS_AMD_TRACE") != nullptr;
        size_t at = 0;
        // (one mapping at a time: the form HIP's own tests use; a refusal would leak the piece silently)
        for (size_t n : lens) {
            const hipError_t e = hipMemUnmap((uint8_t *)base + at, n);
            if (e != hipSuccess && trace) fprintf(stderr, "[rgx trace] hipMemUnmap of %zu bytes at +%zu: %s\n", n, at, hipGetErrorString(e));
            if (e != hipSuccess) (void)hipGetLastError();
            at += n;
        }
        KeptRanges &k = kept_ranges();
        std::lock_guard<std::mutex> lock(k.mu);
        k.v.push_back(KeptRange{base, reserved, dev});
    }
    hipError_t map_pieces(size_t bytes, void **out) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
        size_t gran = 0;                                            // what this runtime wants sizes and addresses to be multiples of (2 MiB on ROCm 7.2)
        if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || !gran) { (void)hipGetLastError();
            gran = (size_t)2 << 20; }
        const size_t total = (bytes + gran - 1) / gran * gran, each = std::max(gran, piece / gran * gran);
        size_t reserved = total;
        void *base = take_range(dev, total, &reserved);
        if (!base) { reserved = total; if ((e = hipMemAddressReserve(&base, total, 0, nullptr, 0)) != hipSuccess) return e; }
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
        if (e != hipSuccess) { unmap_range(base, lens, reserved, dev); return e; }
        *out = base; mapped = reserved; range_dev = dev; piece_len.swap(lens);
        return hipSuccess;
    }
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        static const bool trace = getenv("REGTOOLS_AMD_TRACE") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        release();
        // (growth slack, so that a context which sees files of slowly growing size does not reallocate per call; a one-shot process gets what it asks
        //  for: device memory is cleared when it is handed out, 15.6 GB cost such a process 60-360 ms -- 1.7 GB of that was slack)
        // (first use: after main() said so)
        static const bool no_slack = [] { const char *e = getenv("REGTOOLS_AMD_ONE_SHOT"); return e && strcmp(e, "0") != 0; }();
        size_t want = bytes + (no_slack ? 0 : bytes / 8) + 256;
        void *raw = nullptr;
        hipError_t e = hipErrorNotSupported;
        if (piece && want + kFront >= piece) { e = map_pieces(want + kFront, &raw); if (e != hipSuccess) { (void)hipGetLastError(); raw = nullptr;
            mapped = 0; piece_len.clear(); } }
        if (e != hipSuccess) e = hipMalloc(&raw, want + kFront);
        if (e == hipSuccess) { p = (uint8_t *)raw + kFront; cap = want; }
        if (trace) {                                                 // (shards of a multi-device call grow their buffers on their own threads)
            static std::mutex mu; std::lock_guard<std::mutex> lock(mu);
            g_alloc_stats.ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); ++g_alloc_stats.calls;
                g_alloc_stats.bytes += want;
        }
        return e;
    }
    void release() {
        if (p && mapped) { (void)hipDeviceSynchronize();    /* (what hipFree does by itself: nothing in flight may still touch the range) */
                           unmap_range((uint8_t *)p - kFront, piece_len, mapped, range_dev); }
        else if (p) (void)hipFree((uint8_t *)p - kFront);
        p = nullptr; cap = 0; mapped = 0; piece_len.clear();
    }
    template <class T> T *as() const { return (T *)p; }
};

constexpr int kSideStreams = 2;
// REGTOOLS_AMD_OVERLAP="min_bytes[,chunks[,early_min_members]]" (tests): the thresholds of the overlapped upload, so that files of test size take the path
// the 533 MB bench file takes -- from how many bytes a host buffer goes up in chunks behind ONE arrival-gated inflate launch (default 8 MiB), in how many
// chunks (16), and from how many members an early-tail part is worth cutting (4096; giving it also lifts the part's minimum of 1,024 segments).
struct OverlapKnobs { size_t min_bytes = (size_t)8 << 20; unsigned chunks = 16; uint32_t early_min = 4096; bool early_small = false; };
inline const OverlapKnobs &overlap_knobs() {
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
//           That is what happens on the runtime's default FOUR hardware queues, which the eight streams of two contexts share pairwise (a kernel behind
//           another stream's launch in the same queue waits for it).  With a hardware queue per stream (GPU_MAX_HW_QUEUES >= 16 when HIP starts) the
//           launches go out at once and the files' kernels interleave: 20.2-21.5 ms per file against 20.8-23.0 in turns and 22.1-22.8 on four queues
//           (profiles/r06_pipeline_hw_queues_ab.txt) -- the chip turn is then TWO wide (rgx_link_turn_create: eight queues per file in flight; three files
//           on sixteen queues with two launches at once took 90-93 ms per file, on thirty-two 19.1-19.5).
struct Turn {
    std::mutex mu; std::condition_variable cv; uint64_t next = 0, serving = 0;
    uint64_t width = 1;                                // how many may hold it at once, admitted in the order they asked
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
        x->cv.wait(lk, [&] { return mine < x->serving + x->width; });
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
    // a gated launch whose verdict was not clean / an early-tail wait that timed out on this context: not tried again (a stream layout in which
    bool gate_distrust = false, early_distrust = false;
                                                       // the waiting waves and the kernels that release them share a hardware queue would cost every call its
                                                       // 2 s time-out)
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
    // around the call's whole-range DEFLATE launch, on the stream it runs on (host input: the arrival-gated launch, which spans the upload)
    hipEvent_t ev_launch[2] = {}; bool launch_timed = false;
    std::map<std::string, DevBuf> bufs;
    void *pinned = nullptr; size_t pinned_cap = 0;     // small pinned staging for scalar readbacks
    std::vector<Member> hm_scratch;
    void *pinned_members = nullptr; size_t pinned_members_cap = 0;      // the host scan's member list: kernels read it in place (grow-only)
    void *pinned_rows = nullptr; size_t pinned_rows_cap = 0;
    // rows of the last rgx_extract* call, still in the "rows_out" block in HBM   // grow-only pinned staging for whole result tables (device merge)
    uint64_t last_rows = 0, last_records = 0, last_events = 0, last_bytes = 0; bool last_rows_valid = false;
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


// ---- functions one unit calls in another -------------------------------------------------------------------------------------------------
// api_ctx.cpp
rgx::Fasta *host_fasta(rgx_ctx *c, const char *path);
void ktime_begin(rgx_ctx *c, int slot);
void ktime_end(rgx_ctx *c);
void ktime_collect(rgx_ctx *c);
hipError_t ensure_upload_streams(rgx_ctx *c);
rgx_junction_table *table_alloc(const BamHeader &h, uint64_t n, bool zero = true, bool pinned = false);
void host_sort_rows(rgx_junction_table *t);
void format_bed12_rows(const rgx_junction_table *t, int only_anchored, uint64_t r0, uint64_t r1, std::string &out);
void *block_take(size_t need, size_t &cap, bool pinned);
void block_give(void *p, size_t cap, bool pinned);
// Result tables: all row columns of a table live in ONE block, and released blocks are kept (a few, bounded) for the next table.  A
// pipeline that runs step after step -- bench.py, a multi-GPU job merging every step -- then writes its rows into pages that are already
// mapped instead of paying mmap + first-touch faults + munmap for ~50 bytes per row each time (measured: 9 of 14 ms of an 8-shard merge).
struct TableBox { rgx_junction_table t; void *block; size_t block_cap; bool pinned; };
struct CachedBlock { void *p; size_t cap; bool pinned; };
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
    // identify -s XS: the reads with an N operation whose strand tag lies behind an aux field of unknown type -- bam_aux_get abort()s on them (sam.c:1233-1252)
    // in the first window that reads one (ExtractCfg::odd_count; empty on every file that is not damaged)
    struct OddAux { int32_t tid, pos, end; };
    std::vector<OddAux> odd_aux;
    // the file's bytes in HBM as this call left them (the context's "bam" block, or the caller's device buffer): whole only for an unsharded call.  identify
    // on a
    // damaged file reads every window through the index on its own from here (cse_api.cpp window_join_by_seeks)
    const uint8_t *d_file = nullptr;
};

int prepare_events(rgx_ctx *c, const uint8_t *d_bam_in, const uint8_t *h_bam, size_t bam_len, const uint8_t *bai, size_t bai_len,
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
    void mark(const char *what) { if (trace) { double t = now_ms(); fprintf(stderr, "[rgx trace] %-28s +%8.3f ms  (at %8.3f)%s\n", what, t - t_last,
        t - t_begin, c->link ? (" clock " + std::to_string(fmod(t, 1e5))).c_str() : ""); t_last = t; } }
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
    // a part ends in front of member `members` of the range = workgroup `waves` = arena offset `upos`
    struct EarlyPart { uint32_t members, waves; uint64_t upos; };
    std::vector<EarlyPart> early_parts;
    // early tail: the gated launch still runs on a side stream; whoever reads its part of the arena waits for it
    bool split_B = false; hipEvent_t split_ev = nullptr;
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
    // early tail: per-segment outputs of the decode that the second framing must not overwrite
    uint32_t *seg_iter_e = nullptr, *seg_long_e = nullptr, *seg_long_base_e = nullptr;
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
                if (gated) { c->gate_distrust = true; if (trace) fprintf(stderr,
                    "[rgx trace] arrival gate: verdict not clean, this context no longer uses it\n"); }
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
    // early tail: rows [0, emit_rows) have their events out, in emit_parts parts
    bool emit_parts_ok = false; uint32_t emit_parts = 0, emit_rows = 0; size_t ev_lay = 0;
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

// api_reduce.cpp
int reduce_events(rgx_ctx *c, EventSoA ev, uint32_t n_events, uint32_t group_bits, uint32_t ilen_bits, const uint32_t *rank_of_group_host,
                  uint32_t n_groups, HostRows &R, char *err, size_t errlen, bool view_only = false, RowMap *row_map = nullptr, TableSink *sink = nullptr,
                  bool allow_preagg = true /* identify's window pairs (a few million, one small sort) measured 0.3-0.4 ms slower with it */);
int barcode_rows(rgx_ctx *c, const Prep &P, const RowMap &rm, const rgx_extract_params *p, rgx_junction_table *t, char *err, size_t errlen);
void chrom_string_ranks(const BamHeader &hdr, std::vector<uint32_t> &rank_of_tid);
int run_pipeline(rgx_ctx *c, const uint8_t *d_bam_in, const uint8_t *h_bam, size_t bam_len, const uint8_t *bai, size_t bai_len,
                 const rgx_extract_params *p, rgx_junction_table **out, char *err, size_t errlen, const SharedMembers *shared = nullptr);
// api_entry.cpp
bool rgx_enable_peer(int a, int b);
int rgx_last_table_pack_async(rgx_ctx *c, const rgx_junction_table *t, void **d_packed, hipEvent_t *done, char *err, size_t errlen);
