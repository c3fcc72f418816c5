// barcode_kernels.hip -- `junctions extract -b`: how often each cell barcode (the CB:Z tag) supports each junction
// (junctions_extractor.cc:362-374 set_junction_barcode, :204-217 the per-junction unordered_map<string,int>, h:99-111 print_barcodes).
// The reference copies a hash map per supporting read; here the junction group-by has already run, so the barcodes are a second
// group-by on (output row, barcode): every event gets its read's barcode location + a 64-bit hash, one stable radix sort brings equal
// (row, hash) together in file order, equal hashes are verified byte for byte, and one row per distinct (junction, barcode) comes back
// with its count and its first event -- the host only has to put each junction's distinct barcodes into the reference's container
// order.  Integer / byte work bounded by HBM; the strings stay in the inflated arena until the final gather.
#include "kernels.h"
#include "bam_core.h"

namespace rgx {

__global__ void k_event_urow(const uint32_t *__restrict__ sorted, const uint32_t *__restrict__ head, const uint32_t *__restrict__ seg_excl, uint32_t n,
                             uint32_t *ev_urow) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ev_urow[sorted[i]] = seg_excl[i] + head[i] - 1;
}
__global__ void k_inverse_perm(const uint32_t *__restrict__ perm, uint32_t n, uint32_t *inv) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) inv[perm[i]] = i;
}

__global__ void k_bc_event_keys(const uint8_t *__restrict__ arena, uint32_t n_events, const uint32_t *__restrict__ ev_read, const uint64_t *__restrict__ rec_off,
                                const uint32_t *__restrict__ ev_urow, const uint32_t *__restrict__ urow_pos, uint8_t t0, uint8_t t1, BarcodeEv b, uint32_t *flags) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_events) return;
    const uint64_t o = rec_off[ev_read[e]];
    RecHead h; rec_head(arena + o, h);
    const uint8_t *data = arena + o + 36;
    const int64_t aux_off = (int64_t)h.l_qname + 4 * (int64_t)h.n_cigar + (((int64_t)h.l_qseq + 1) >> 1) + h.l_qseq;
    const int64_t l_data = (int64_t)h.block_len - 32;
    uint32_t val = 0, len = 0;
    const int r = aux_find_string(data + aux_off, data + l_data, t0, t1, &val, &len);
    if (r < 0) flags[0] = 1;
    uint32_t lo, hi;
    if (r > 0) {
        const uint8_t *sp = data + aux_off + val;
        barcode_hash(sp, len, &lo, &hi);
        b.off[e] = (uint64_t)(sp - arena); b.len[e] = len;
    } else {
        const uint8_t q = '?';                       // no tag: the key is the one-character string "?" (cc:370) -- the same key as a literal CB:Z:?
        barcode_hash(&q, 1, &lo, &hi);
        b.off[e] = ~0ull; b.len[e] = 1;
    }
    b.h_lo[e] = lo; b.h_hi[e] = hi;
    b.row[e] = urow_pos[ev_urow[e]];
}

__device__ __forceinline__ uint8_t bc_byte(const uint8_t *arena, uint64_t off, uint32_t k) { return off == ~0ull ? (uint8_t)'?' : arena[off + k]; }

__global__ void k_bc_heads(const uint8_t *__restrict__ arena, BarcodeEv b, const uint32_t *__restrict__ perm, uint32_t n, uint32_t *head, uint32_t *flags) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t hd = 1;
    if (i > 0) {
        const uint32_t x = perm[i], y = perm[i - 1];
        hd = b.row[x] != b.row[y] || b.h_lo[x] != b.h_lo[y] || b.h_hi[x] != b.h_hi[y];
        if (!hd) {                                   // same junction, same hash: the strings must be the same bytes (transitive along the run)
            bool same = b.len[x] == b.len[y];
            for (uint32_t k = 0; same && k < b.len[x]; ++k) same = bc_byte(arena, b.off[x], k) == bc_byte(arena, b.off[y], k);
            if (!same) flags[1] = 1;
        }
    }
    head[i] = hd;
}

__global__ void k_bc_pairs(BarcodeEv b, const uint32_t *__restrict__ perm, const uint32_t *__restrict__ head, const uint32_t *__restrict__ seg_excl, uint32_t n,
                           uint32_t *pair_row, uint32_t *pair_first, uint32_t *pair_pos, uint64_t *pair_off, uint32_t *pair_len) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !head[i]) return;
    const uint32_t k = seg_excl[i], e = perm[i];     // stable sort: the head of a run is its earliest event
    pair_row[k] = b.row[e]; pair_first[k] = e; pair_pos[k] = i; pair_off[k] = b.off[e]; pair_len[k] = b.len[e];
}
__global__ void k_bc_counts(uint32_t n_pairs, uint32_t n, const uint32_t *__restrict__ pair_pos, uint32_t *pair_count) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_pairs) pair_count[k] = (k + 1 < n_pairs ? pair_pos[k + 1] : n) - pair_pos[k];
}
__global__ void k_bc_gather(const uint8_t *__restrict__ arena, uint32_t n_pairs, const uint64_t *__restrict__ pair_off, const uint32_t *__restrict__ pair_len,
                            const uint32_t *__restrict__ str_begin, uint8_t *text) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_pairs) return;
    const uint64_t off = pair_off[k];
    uint8_t *dst = text + str_begin[k];
    for (uint32_t q = 0; q < pair_len[k]; ++q) dst[q] = bc_byte(arena, off, q);
}

static inline dim3 grid1(uint32_t n) { return dim3((n + 255) / 256); }
void launch_event_urow(const uint32_t *sorted, const uint32_t *head, const uint32_t *seg_excl, uint32_t n, uint32_t *ev_urow, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(k_event_urow, grid1(n), dim3(256), 0, stream, sorted, head, seg_excl, n, ev_urow);
}
void launch_inverse_perm(const uint32_t *perm, uint32_t n, uint32_t *inv, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(k_inverse_perm, grid1(n), dim3(256), 0, stream, perm, n, inv);
}
void launch_bc_event_keys(const uint8_t *arena, uint32_t n_events, const uint32_t *ev_read, const uint64_t *rec_off, const uint32_t *ev_urow,
                          const uint32_t *urow_pos, uint8_t t0, uint8_t t1, BarcodeEv b, uint32_t *flags, hipStream_t stream) {
    if (n_events) hipLaunchKernelGGL(k_bc_event_keys, grid1(n_events), dim3(256), 0, stream, arena, n_events, ev_read, rec_off, ev_urow, urow_pos, t0, t1, b, flags);
}
void launch_bc_heads(const uint8_t *arena, BarcodeEv b, const uint32_t *perm, uint32_t n, uint32_t *head, uint32_t *flags, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(k_bc_heads, grid1(n), dim3(256), 0, stream, arena, b, perm, n, head, flags);
}
void launch_bc_pairs(BarcodeEv b, const uint32_t *perm, const uint32_t *head, const uint32_t *seg_excl, uint32_t n, uint32_t *pair_row,
                     uint32_t *pair_first, uint32_t *pair_pos, uint64_t *pair_off, uint32_t *pair_len, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(k_bc_pairs, grid1(n), dim3(256), 0, stream, b, perm, head, seg_excl, n, pair_row, pair_first, pair_pos, pair_off, pair_len);
}
void launch_bc_counts(uint32_t n_pairs, uint32_t n, const uint32_t *pair_pos, uint32_t *pair_count, hipStream_t stream) {
    if (n_pairs) hipLaunchKernelGGL(k_bc_counts, grid1(n_pairs), dim3(256), 0, stream, n_pairs, n, pair_pos, pair_count);
}
void launch_bc_gather(const uint8_t *arena, uint32_t n_pairs, const uint64_t *pair_off, const uint32_t *pair_len, const uint32_t *str_begin,
                      uint8_t *text, hipStream_t stream) {
    if (n_pairs) hipLaunchKernelGGL(k_bc_gather, grid1(n_pairs), dim3(256), 0, stream, arena, n_pairs, pair_off, pair_len, str_begin, text);
}

}  // namespace rgx
