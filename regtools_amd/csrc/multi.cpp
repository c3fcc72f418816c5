// multi.cpp -- `junctions extract` over several GPUs of one node from ONE host process (BASELINE.json north_star:
// "BAMs shard by BGZF block / coordinate window across the GPUs of one node with a final RCCL reduce of per-junction counts").
//
// Replaces, for a multi-GPU node, the single call junctions_extract() makes into JunctionsExtractor::identify_junctions_from_BAM
// (/root/reference/src/junctions/junctions_main.cc:45-59).  One host thread, one context and one stream per device:
//   thread g:  shard g of n (contiguous BGZF member range cut at record starts the index lists, api_front.cpp stage_members) ->
//              the whole single-GPU pipeline on device g -> its unique rows packed in HBM (48 bytes per row)
//   exchange:  ONE grouped gather of the padded row blocks to the first device over xGMI (ncclSend / ncclRecv in one group; RCCL is loaded
//              at run time: librccl.so.1 is the only thing this library needs from it, and a process that also runs PyTorch must not end
//              up with two RCCL copies bound at link time).  Communicators, exchange buffers and streams are made once per device list
//              and kept for the life of the process, like the contexts.
//   host:      the file's BGZF members are found ONCE (scan_members_parallel) and every shard uploads only the header's members and its
//              own byte range (api_front.cpp stage_upload): N shards move the file over PCIe once, not N times.
//   merge:     on the first device, rgx_table_merge_device: radix sort by key, sum / min / max, first-seen naming by (shard, rank),
//              strand of the last shard that saw the key, output order.  Shard order = file order, so the table is the single-GPU table.
// The same device may be listed more than once (the shards then run one after the other on it and the exchange is a device copy): that
// is how a one-GPU box exercises every line here except the collective itself.
#include "../../include/regtools_amd.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <set>
#include <thread>
#include <vector>

#include "host_io.h"

// api_entry.cpp (not part of the C ABI): the rows of a context's last extraction packed on ITS stream, an event behind the kernel; peer access per pair
int rgx_last_table_pack_async(rgx_ctx *c, const rgx_junction_table *t, void **d_packed, hipEvent_t *done, char *err, size_t errlen);
bool rgx_enable_peer(int a, int b);
// api_entry.cpp: rgx_extract_mem with the member list the caller scanned (not part of the C ABI)
int rgx_extract_mem_scanned(rgx_ctx *ctx, const void *bam, size_t bam_len, const void *bai, size_t bai_len, const rgx_extract_params *p,
                            const std::vector<rgx::Member> *members, uint64_t total_inflated, rgx_junction_table **out, char *err, size_t errlen);

using namespace rgx;

void rgx_ctx_no_arena_trials(rgx_ctx *c);            // api_ctx.cpp: no arena placement trials for this context

namespace {

int failm(char *err, size_t errlen, int code, const char *fmt, ...) {
    if (err && errlen) { va_list ap; va_start(ap, fmt); vsnprintf(err, errlen, fmt, ap); va_end(ap); }
    return code;
}

// ---- RCCL, bound at run time ------------------------------------------------------------------------------------------------------
// (rccl.h: ncclResult_t is an int with ncclSuccess == 0; ncclComm_t an opaque pointer; ncclUint8 == 1 in ncclDataType_t)
typedef void *nccl_comm;
struct Rccl {
    void *so = nullptr;
    int (*CommInitAll)(nccl_comm *, int, const int *) = nullptr;
    int (*CommDestroy)(nccl_comm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, nccl_comm, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool load(char *err, size_t errlen) {
        if (so) return true;
        for (const char *name : {"librccl.so.1", "librccl.so"}) { so = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (so) break; }
        if (!so) { failm(err, errlen, RGX_ERR_DEVICE, "regtools_amd: cannot load RCCL (librccl.so.1): %s\n", dlerror()); return false; }
        CommInitAll = (decltype(CommInitAll))dlsym(so, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(so, "ncclCommDestroy");
        GroupStart = (decltype(GroupStart))dlsym(so, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(so, "ncclGroupEnd");
        AllGather = (decltype(AllGather))dlsym(so, "ncclAllGather");
        Send = (decltype(Send))dlsym(so, "ncclSend");
        Recv = (decltype(Recv))dlsym(so, "ncclRecv");
        GetErrorString = (decltype(GetErrorString))dlsym(so, "ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !AllGather) {
            failm(err, errlen, RGX_ERR_DEVICE, "regtools_amd: librccl.so.1 lacks a symbol this path needs\n");
            return false;
        }
        return true;
    }
};
Rccl g_rccl;
const int kNcclUint8 = 1;
char g_exchange_kind[160] = "none";      // how the last rgx_extract_multi call moved its rows (rgx_multi_exchange_kind)

struct Shard {
    int device = 0, nth = 0;                 // nth: which of the listings of this device
    rgx_ctx *ctx = nullptr;                  // (owned by the process-wide cache)
    rgx_junction_table *table = nullptr;
    int rc = RGX_OK;
    char err[512] = {0};
    double ms_extract = 0;
    void *d_packed = nullptr;                // the shard's rows, 48 bytes each, in a buffer of its context (packed on the shard's thread and stream)
    hipEvent_t ev_packed = nullptr;          // ... recorded behind the pack kernel: the exchange waits for it on the device, no host wait per shard
};

// Contexts are kept for the life of the process, one per (device, how many times the device is listed): a second call finds its HBM
// workspace, streams and page-locked staging where the first left them (a context's first extraction allocates ~13 GB for a 50 M-read shard).
std::mutex g_ctx_mu;
std::map<std::pair<int, int>, rgx_ctx *> g_ctx;
rgx_ctx *context_for(int device, int nth, char *err, size_t errlen, int &rc) {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    auto it = g_ctx.find({device, nth});
    if (it != g_ctx.end()) { rc = RGX_OK; return it->second; }
    rgx_ctx *c = nullptr;
    rc = rgx_ctx_create(device, &c, err, errlen);
    if (rc == RGX_OK) { g_ctx[{device, nth}] = c; if (nth > 0) rgx_ctx_no_arena_trials(c); }
    return c;
}

}  // namespace
// (for rgx_identify_multi, cse_api.cpp: the same cache)
rgx_ctx *rgx_multi_context(int device, int nth, char *err, size_t errlen, int *rc) { int r = RGX_OK; rgx_ctx *c = context_for(device, nth, err, errlen, r);
    if (rc) *rc = r; return c; }
namespace {

// What the exchange needs besides the contexts, per device LIST: RCCL communicators (ncclCommInitAll on an 8-GPU node takes hundreds of
// milliseconds -- once, not per call), one stream and one send buffer per entry, the receive block on the first device.  Buffers grow,
// nothing is handed back before the process ends (rgx_extract_multi calls take turns: call_mu).
struct Exchange {
    std::vector<int> devices;
    std::vector<nccl_comm> comms;            // empty while the list repeats a device (no collective then) or when RCCL could not be set up
    std::string rccl_error;                  // why comms is empty on a list of distinct devices
    std::vector<char> peer;                  // [g]: peer access between devices[0] and devices[g] is on (xGMI copies; else they go through the host)
    std::vector<hipStream_t> streams;
    std::vector<void *> d_send; std::vector<size_t> send_cap;
    void *d_recv = nullptr; size_t recv_cap = 0;
};
std::map<std::vector<int>, Exchange> g_exchange;

int exchange_for(const int *devices, int n, bool distinct, size_t block, const std::vector<char> &need_send, Exchange *&out, char *err, size_t errlen) {
    std::vector<int> key(devices, devices + n);
    Exchange &x = g_exchange[key];
    if (x.devices.empty()) {
        x.devices = key; x.streams.assign((size_t)n, nullptr); x.d_send.assign((size_t)n, nullptr); x.send_cap.assign((size_t)n, 0);
        for (int g = 0; g < n; ++g)
            if (hipSetDevice(devices[g]) != hipSuccess || hipStreamCreateWithFlags(&x.streams[(size_t)g], hipStreamNonBlocking) != hipSuccess)
                return failm(err, errlen, RGX_ERR_DEVICE, "regtools_amd: no stream for the row exchange on device %d\n", devices[g]);
        x.peer.assign((size_t)n, 1);
        if (distinct && n > 1) {
            // peer access first (RCCL's own transports and the fall-back copies both want it), then the communicators.  RCCL that cannot be
            // loaded or initialised is not the end of the call: the gather then is peer copies (reported: rgx_multi_exchange_kind)
            for (int g = 1; g < n; ++g) x.peer[(size_t)g] = rgx_enable_peer(devices[0], devices[g]) ? 1 : 0;
            char why[256] = {0};
            if (!g_rccl.load(why, sizeof why)) x.rccl_error = why;
            else if (!g_rccl.Send || !g_rccl.Recv) x.rccl_error = "this librccl has no ncclSend / ncclRecv";
            else {
                x.comms.assign((size_t)n, nullptr);
                const int r = g_rccl.CommInitAll(x.comms.data(), n, devices);
                if (r != 0) { x.comms.clear(); x.rccl_error = std::string("ncclCommInitAll failed: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) :
                    "?"); }
            }
            while (!x.rccl_error.empty() && (x.rccl_error.back() == '\n' || x.rccl_error.back() == ' ')) x.rccl_error.pop_back();
        }
    }
    for (int g = 0; g < n; ++g) {
        if (!need_send[(size_t)g] || x.send_cap[(size_t)g] >= block) continue;
        if (hipSetDevice(devices[g]) != hipSuccess) return failm(err, errlen, RGX_ERR_DEVICE, "regtools_amd: device %d\n", devices[g]);
        if (x.d_send[(size_t)g]) (void)hipFree(x.d_send[(size_t)g]);
        x.d_send[(size_t)g] = nullptr; x.send_cap[(size_t)g] = 0;
        if (hipMalloc(&x.d_send[(size_t)g], block + block / 4 + 4096) != hipSuccess) return failm(err, errlen, RGX_ERR_DEVICE,
            "regtools_amd: no memory for the row exchange on device %d\n", devices[g]);
        x.send_cap[(size_t)g] = block + block / 4 + 4096;
    }
    if (x.recv_cap < block * (size_t)n) {
        if (hipSetDevice(devices[0]) != hipSuccess) return failm(err, errlen, RGX_ERR_DEVICE, "regtools_amd: device %d\n", devices[0]);
        if (x.d_recv) (void)hipFree(x.d_recv);
        x.d_recv = nullptr; x.recv_cap = 0;
        const size_t want = (block + block / 4 + 4096) * (size_t)n;
        if (hipMalloc(&x.d_recv, want) != hipSuccess) return failm(err, errlen, RGX_ERR_DEVICE,
            "regtools_amd: no memory for the row exchange on device %d\n", devices[0]);
        x.recv_cap = want;
    }
    out = &x;
    return RGX_OK;
}

void release(std::vector<Shard> &S) {
    for (Shard &s : S) if (s.table) rgx_table_free(s.table);
}

double now_ms() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }

}  // namespace

extern "C" int rgx_extract_multi_mem(const int *devices, int n_devices, const void *bam, size_t bam_len, const void *bai, size_t bai_len,
                                     const rgx_extract_params *p, rgx_junction_table **out, char *err, size_t errlen) {
    if (!devices || n_devices <= 0 || n_devices > 255 || !bam || !out || !p) return failm(err, errlen, RGX_ERR_ARG, "regtools_amd: bad arguments\n");
    // the contexts are shared by every call of this process: calls take turns
    static std::mutex call_mu;
    std::lock_guard<std::mutex> call_lock(call_mu);
    if (p->n_shards > 1) return failm(err, errlen, RGX_ERR_ARG, "regtools_amd: rgx_extract_multi shards the file itself\n");
    *out = nullptr;
    const int n = n_devices;
    std::vector<Shard> S((size_t)n);
    struct Guard { std::vector<Shard> &s; ~Guard() { release(s); } } guard{S};
    const bool distinct = std::set<int>(devices, devices + n).size() == (size_t)n;
    { std::map<int, int> seen; for (int g = 0; g < n; ++g) { S[(size_t)g].device = devices[g]; S[(size_t)g].nth = seen[devices[g]]++; } }

    // -- the file's members, found once for all shards (a file the host scan does not vouch for: every shard takes the device's member
    //    discovery on the whole file, as a single-GPU call would) -----------------------------------------------------------------------
    const double t_begin = now_ms();
    std::vector<Member> members; uint64_t total_inflated = 0;
    if (n > 1 && bam_len >= ((size_t)8 << 20) && !scan_members_parallel((const uint8_t *)bam, bam_len, (int)usable_threads(24), members,
        total_inflated)) members.clear();
    const double t_scan = now_ms();
    // -- one extraction per shard: a thread per device (shards that share a device take turns on it) -------------------------------------
    auto extract = [&](int g) {
        Shard &s = S[(size_t)g];
        const double t0 = now_ms();
        s.ctx = context_for(s.device, s.nth, s.err, sizeof s.err, s.rc);
        if (s.rc != RGX_OK) return;
        rgx_extract_params q = *p;
        q.shard = g; q.n_shards = n;
        s.rc = rgx_extract_mem_scanned(s.ctx, bam, bam_len, bai, bai_len, &q, &members, total_inflated, &s.table, s.err, sizeof s.err);
        // the shard's rows are still in HBM: packed for the exchange right here, on the shard's own thread and stream (round 3 packed the shards
        // one after the other on the calling thread, a host wait each).  A shard whose rows are not its context's last any more (several
        // listings of one device share nothing, so this does not happen today) is packed from the host table below.
        if (s.rc == RGX_OK && n > 1 && s.table->n) {
            char e2[256];
            if (rgx_last_table_pack_async(s.ctx, s.table, &s.d_packed, &s.ev_packed, e2, sizeof e2) != RGX_OK) { s.d_packed = nullptr; s.ev_packed = nullptr; }
        }
        s.ms_extract = now_ms() - t0;
    };
    if (distinct) {
        std::vector<std::thread> pool;
        for (int g = 1; g < n; ++g) pool.emplace_back(extract, g);
        extract(0);
        for (auto &t : pool) t.join();
    } else for (int g = 0; g < n; ++g) extract(g);
    for (int g = 0; g < n; ++g) if (S[(size_t)g].rc != RGX_OK) return failm(err, errlen, S[(size_t)g].rc, "%s", S[(size_t)g].err);
    // (REGTOOLS_AMD_RCCL=selftest: a one-device list goes through pack, ncclCommInitAll / ncclAllGather of one rank and the device merge
    // as well -- the collective's call sequence on the real library where only one GPU is visible; REGTOOLS_AMD_RCCL=off: the peer-copy fall-back on purpose)
    const char *rccl_env = getenv("REGTOOLS_AMD_RCCL");
    const bool selftest = rccl_env && !strcmp(rccl_env, "selftest");
    if (n == 1 && !selftest) { snprintf(g_exchange_kind, sizeof g_exchange_kind, "none (one shard)"); *out = S[0].table; S[0].table = nullptr; return RGX_OK; }
    if (n == 1) {
        // one rank through ncclCommInitAll / ncclAllGather on the real library (no peer to send to)
        if (!g_rccl.load(err, errlen)) return RGX_ERR_DEVICE;
        nccl_comm comm = nullptr; void *d_a = nullptr, *d_b = nullptr; hipStream_t st1 = nullptr;
        const size_t blk = std::max<size_t>(1, (size_t)S[0].table->n) * RGX_PACKED_ROW_BYTES;
        int r = g_rccl.CommInitAll(&comm, 1, devices);
        bool ok = r == 0 && hipSetDevice(devices[0]) == hipSuccess && hipMalloc(&d_a, blk) == hipSuccess && hipMalloc(&d_b, blk) == hipSuccess &&
                  hipStreamCreateWithFlags(&st1, hipStreamNonBlocking) == hipSuccess;
        if (ok && S[0].table->n) ok = rgx_last_table_pack_device(S[0].ctx, S[0].table, d_a, S[0].table->n, err, errlen) == RGX_OK;
        if (ok) { r = g_rccl.GroupStart(); if (r == 0) r = g_rccl.AllGather(d_a, d_b, blk, kNcclUint8, comm, st1); const int r2 = g_rccl.GroupEnd();
            ok = r == 0 && r2 == 0 && hipStreamSynchronize(st1) == hipSuccess; }
        rgx_junction_table *m1 = nullptr;
        uint64_t rows1 = S[0].table->n;
        int rc1 = ok ? rgx_table_merge_device(S[0].ctx, d_b, std::max<uint64_t>(1, rows1), &rows1, 1, p->min_anchor, S[0].table, &m1, err, errlen) :
            RGX_ERR_DEVICE;
        if (comm) g_rccl.CommDestroy(comm);
        if (d_a) (void)hipFree(d_a);
        if (d_b) (void)hipFree(d_b);
        if (st1) (void)hipStreamDestroy(st1);
        if (rc1 != RGX_OK) return ok ? rc1 : failm(err, errlen, RGX_ERR_DEVICE, "regtools_amd: RCCL self test failed\n");
        const rgx_junction_table *t = S[0].table;
        m1->n_records = t->n_records; m1->n_events = t->n_events; m1->inflated_bytes = t->inflated_bytes; m1->n_members = t->n_members;
            m1->compressed_bytes = bam_len;
        m1->stream_ended = t->stream_ended; m1->framing_sweeps = t->framing_sweeps;
        if (p->barcodes) { const rgx_junction_table *parts1[1] = {t}; rc1 = rgx_table_merge_barcodes(parts1, 1, m1, err, errlen);
            if (rc1 != RGX_OK) { rgx_table_free(m1); return rc1; } }
        *out = m1;
        return RGX_OK;
    }

    // -- what every shard sends: its packed rows where they lie (d_packed, ready when ev_packed fires), or -- no device copy of them left --
    //    a block uploaded from the host table ---------------------------------------------------------------------------------------------
    const double t_extract = now_ms();
    uint64_t stride = 1;
    std::vector<uint64_t> part_rows((size_t)n);
    std::vector<char> need_send((size_t)n, 0);
    for (int g = 0; g < n; ++g) {
        part_rows[(size_t)g] = S[(size_t)g].table->n; stride = std::max<uint64_t>(stride, S[(size_t)g].table->n);
        need_send[(size_t)g] = S[(size_t)g].table->n && !S[(size_t)g].d_packed;
    }
    const size_t block = (size_t)stride * RGX_PACKED_ROW_BYTES;
    Exchange *X = nullptr;
    { const int rcx = exchange_for(devices, n, distinct, block, need_send, X, err, errlen); if (rcx != RGX_OK) return rcx; }
    std::vector<const void *> src((size_t)n, nullptr);
    for (int g = 0; g < n; ++g) {
        Shard &s = S[(size_t)g];
        if (!s.table->n) continue;
        if (hipSetDevice(s.device) != hipSuccess) return failm(err, errlen, RGX_ERR_DEVICE, "regtools_amd: device %d\n", s.device);
        if (s.d_packed) {
            // the exchange stream of this shard waits for the pack ON THE DEVICE
            if (hipStreamWaitEvent(X->streams[(size_t)g], s.ev_packed, 0) != hipSuccess) return failm(err, errlen, RGX_ERR_DEVICE,
                "regtools_amd: no wait on the rows of device %d\n", s.device);
            src[(size_t)g] = s.d_packed;
        } else {
            std::vector<uint8_t> h((size_t)s.table->n * RGX_PACKED_ROW_BYTES);
            rgx_table_pack(s.table, h.data(), h.size());
            if (hipMemcpy(X->d_send[(size_t)g], h.data(), h.size(), hipMemcpyHostToDevice) != hipSuccess) return failm(err, errlen, RGX_ERR_DEVICE,
                "regtools_amd: row upload failed\n");
            src[(size_t)g] = X->d_send[(size_t)g];
        }
    }

    // -- the one exchange of the job: every shard's rows to the first device (shard g's at d_recv + g x block; it sends exactly its rows) ---------
    auto bytes_of = [&](int g) { return (size_t)part_rows[(size_t)g] * RGX_PACKED_ROW_BYTES; };
    bool by_rccl = false;
    std::string rccl_note = distinct ? X->rccl_error : std::string();
    // (REGTOOLS_AMD_RCCL=off: the fall-back on purpose -- how a multi-GPU box tests it)
    const bool rccl_off = getenv("REGTOOLS_AMD_RCCL") && !strcmp(getenv("REGTOOLS_AMD_RCCL"), "off");
    if (distinct && rccl_off) rccl_note = "REGTOOLS_AMD_RCCL=off is set";
    if (distinct && !X->comms.empty() && !rccl_off) {
        // a gather: rank g sends its rows, rank 0 receives n - 1 blocks, all in one group (its own rows are a device copy on its stream)
        int r = g_rccl.GroupStart();
        for (int g = 1; g < n && r == 0; ++g) {
            if (!bytes_of(g)) continue;
            if (hipSetDevice(devices[g]) != hipSuccess) { r = -1; break; }
            r = g_rccl.Send(src[(size_t)g], bytes_of(g), kNcclUint8, 0, X->comms[(size_t)g], X->streams[(size_t)g]);
        }
        if (r == 0 && hipSetDevice(devices[0]) != hipSuccess) r = -1;
        for (int g = 1; g < n && r == 0; ++g) if (bytes_of(g)) r = g_rccl.Recv((uint8_t *)X->d_recv + (size_t)g * block, bytes_of(g), kNcclUint8, g,
            X->comms[0], X->streams[0]);
        const int r2 = g_rccl.GroupEnd();
        by_rccl = r == 0 && r2 == 0;
        if (by_rccl) {
            for (int g = 0; g < n && by_rccl; ++g)
                if (hipSetDevice(devices[g]) != hipSuccess || hipStreamSynchronize(X->streams[(size_t)g]) != hipSuccess) by_rccl = false;
            if (!by_rccl) rccl_note = "the grouped ncclSend / ncclRecv did not complete";
        } else rccl_note = std::string("the grouped ncclSend / ncclRecv failed: ") + (r < 0 ? "device selection" : g_rccl.GetErrorString ?
            g_rccl.GetErrorString(r > 0 ? r : r2) : "?");
        (void)hipGetLastError();
    }
    if (hipSetDevice(devices[0]) != hipSuccess) return failm(err, errlen, RGX_ERR_DEVICE, "regtools_amd: device %d\n", devices[0]);
    // rows that did not travel by RCCL: copies on the first device's exchange stream, each behind its shard's pack (peer copies between distinct
    // devices -- the fall-back when RCCL is missing or failed, said so in rgx_multi_exchange_kind --, device copies when a device is listed twice)
    for (int g = 0; g < n; ++g) {
        if (!bytes_of(g) || (by_rccl && g > 0)) continue;
        const Shard &s = S[(size_t)g];
        if (s.d_packed && hipStreamWaitEvent(X->streams[0], s.ev_packed, 0) != hipSuccess) return failm(err, errlen, RGX_ERR_DEVICE,
            "regtools_amd: no wait on the rows of device %d\n", s.device);
        const hipError_t e = s.device == devices[0] ? hipMemcpyAsync((uint8_t *)X->d_recv + (size_t)g * block, src[(size_t)g], bytes_of(g),
            hipMemcpyDeviceToDevice, X->streams[0])
                                                    : hipMemcpyPeerAsync((uint8_t *)X->d_recv + (size_t)g * block, devices[0], src[(size_t)g], s.device,
                                                        bytes_of(g), X->streams[0]);
        if (e != hipSuccess) return failm(err, errlen, RGX_ERR_DEVICE, "regtools_amd: row copy from device %d failed: %s\n", s.device, hipGetErrorString(e));
    }
    // (the merge runs on the context's own stream: the rows must be there before it is enqueued)
    if (hipStreamSynchronize(X->streams[0]) != hipSuccess) return failm(err, errlen, RGX_ERR_DEVICE, "regtools_amd: the row copies did not complete\n");
    {
        bool all_peer = true; for (int g = 1; g < n; ++g) if (!X->peer[(size_t)g]) all_peer = false;
        if (!distinct) snprintf(g_exchange_kind, sizeof g_exchange_kind, "device copies (a device is listed more than once)");
        else if (by_rccl) snprintf(g_exchange_kind, sizeof g_exchange_kind, "rccl grouped send/recv, %d ranks%s", n, all_peer ? ", peer access on" :
            ", peer access NOT available");
        else snprintf(g_exchange_kind, sizeof g_exchange_kind, "hipMemcpyPeerAsync%s (RCCL not used: %.90s)", all_peer ? " over peer access" :
            " WITHOUT peer access", rccl_note.c_str());
    }
    const double t_exchange = now_ms();

    // -- merge on the first device.  A shard whose record stream ENDED (a member that does not inflate, an unreadable record) hides the
    //    shards behind it: a sequential reader never gets there (api_entry.cpp, rgx_table_merge) ---------------------------------------------------
    std::vector<uint64_t> merge_rows = part_rows;
    for (int g = 0; g < n; ++g) if (S[(size_t)g].table->stream_ended) { for (int k = g + 1; k < n; ++k) merge_rows[(size_t)k] = 0; break; }
    rgx_junction_table *m = nullptr;
    int rc = rgx_table_merge_device(S[0].ctx, X->d_recv, stride, merge_rows.data(), n, p->min_anchor, S[0].table, &m, err, errlen);
    if (rc != RGX_OK) return rc;
    const double t_merge = now_ms();
    if (getenv("REGTOOLS_AMD_TRACE")) {
        fprintf(stderr, "[rgx trace] multi: %d shards (%s), member scan %.3f ms, shards %.3f ms (", n, g_exchange_kind,
                t_scan - t_begin, t_extract - t_scan);
        for (int g = 0; g < n; ++g) fprintf(stderr, "%s%.1f", g ? " " : "", S[(size_t)g].ms_extract);
        fprintf(stderr, "), pack + exchange %.3f ms, merge %.3f ms\n", t_exchange - t_extract, t_merge - t_exchange);
    }
    for (int g = 0; g < n; ++g) {
        const rgx_junction_table *t = S[(size_t)g].table;
        m->n_records += t->n_records; m->n_events += t->n_events; m->inflated_bytes += t->inflated_bytes; m->n_members += t->n_members;
        m->ms_inflate = std::max(m->ms_inflate, t->ms_inflate); m->ms_records = std::max(m->ms_records, t->ms_records);
        m->ms_scan = std::max(m->ms_scan, t->ms_scan); m->ms_reduce = std::max(m->ms_reduce, t->ms_reduce); m->ms_total = std::max(m->ms_total, t->ms_total);
        m->framing_sweeps = std::max(m->framing_sweeps, t->framing_sweeps);
        if (t->stream_ended) { m->stream_ended = 1; break; }
    }
    m->compressed_bytes = bam_len;
    if (p->barcodes) {          // -b: the shards' per-junction barcode lists, one after the other in file order (api_entry.cpp rgx_table_merge_barcodes)
        std::vector<const rgx_junction_table *> parts((size_t)n);
        for (int g = 0; g < n; ++g) parts[(size_t)g] = S[(size_t)g].table;
        rc = rgx_table_merge_barcodes(parts.data(), n, m, err, errlen);
        if (rc != RGX_OK) { rgx_table_free(m); return rc; }
    }
    *out = m;
    return RGX_OK;
}

// how the last rgx_extract_multi / rgx_extract_multi_mem call of this process moved the shards' rows to the first device
extern "C" const char *rgx_multi_exchange_kind(void) { return g_exchange_kind; }

extern "C" int rgx_extract_multi(const int *devices, int n_devices, const char *bam_path, const rgx_extract_params *p, rgx_junction_table **out,
                                 char *err, size_t errlen) {
    if (!bam_path || !out) return failm(err, errlen, RGX_ERR_ARG, "regtools_amd: bad arguments\n");
    FileBytes bam; std::vector<uint8_t> bai;
    if (!bam.open(bam_path)) return failm(err, errlen, RGX_ERR_OPEN, "[E::hts_open_format] fail to open file '%s'\nUnable to open BAM/SAM file.\n\n", bam_path);
    std::string idx;
    const bool have_index = find_index(bam_path, idx) == 0;
    fputs(bam_open_notes(bam.data(), bam.size(), have_index ? bam_path : nullptr, have_index ? idx.c_str() : nullptr).c_str(), stderr);
    if (!have_index || !read_index(idx, bai)) return failm(err, errlen, RGX_ERR_INDEX,
        "Unable to open BAM/SAM index. Make sure alignments are indexed\n\n");
    return rgx_extract_multi_mem(devices, n_devices, bam.data(), bam.size(), bai.data(), bai.size(), p, out, err, errlen);
}
