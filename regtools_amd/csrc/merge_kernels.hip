// merge_kernels.hip -- device half of the multi-GPU table merge: the packed 48-byte rows of all shards, as the ONE
// all-gather left them in HBM, are grouped by (tid, start, end, strand class) and reduced exactly like the host merge
// (rgx_table_merge): sum of counts, min/max thick bounds, the name of the earliest first-seen, the strand of the last shard
// that saw the key; then the first-seen naming and the output order.  Integer work bounded by HBM; a few dozen small launches.
#include "kernels.h"

namespace rgx {

// packed row (rgx_table_pack): tid,start,end,ts,te,count,first_lo,first_hi,last_lo,last_hi,strand,name_index
__global__ void k_merge_unpack(const uint32_t *__restrict__ rows, uint32_t stride_rows, uint32_t n_parts, const uint32_t *__restrict__ part_rows,
                               const uint32_t *__restrict__ part_base, MergeSoA m) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t g = r / stride_rows, j = r % stride_rows;
    if (g >= n_parts || j >= part_rows[g]) return;
    const uint32_t *w = rows + (size_t)r * 12;
    const uint32_t o = part_base[g] + j;
    const uint32_t strand = w[10] & 0xff;
    m.tid[o] = w[0]; m.start[o] = w[1]; m.end[o] = w[2]; m.ts[o] = w[3]; m.te[o] = w[4]; m.count[o] = w[5];
    m.cls[o] = strand == '+' ? 0u : strand == '-' ? 1u : 2u;
    m.strand[o] = strand;
    m.first[o] = g << 24 | ((w[11] - 1u) & 0xffffffu);         // (shard, rank of the row's first read inside the shard)
    m.shard[o] = g;
}

// the ten-column block a pipeline run leaves in HBM (launch_rows_out) -> packed 48-byte rows, written straight into the all-gather input
__global__ void k_cols_to_packed(const uint32_t *__restrict__ cols, uint32_t n, uint32_t *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t N = n;
    uint32_t *w = out + (size_t)i * 12;
    w[0] = cols[i]; w[1] = cols[N + i]; w[2] = cols[2 * N + i]; w[3] = cols[3 * N + i]; w[4] = cols[4 * N + i]; w[5] = cols[5 * N + i];
    w[6] = cols[7 * N + i]; w[7] = 0; w[8] = cols[8 * N + i]; w[9] = 0; w[10] = cols[9 * N + i]; w[11] = cols[6 * N + i];
}

__device__ __forceinline__ bool merge_same(const MergeSoA &m, uint32_t a, uint32_t b) {
    return m.tid[a] == m.tid[b] && m.start[a] == m.start[b] && m.end[a] == m.end[b] && m.cls[a] == m.cls[b];
}

__global__ void k_merge_heads(MergeSoA m, const uint32_t *__restrict__ sorted, uint32_t n, uint32_t *head) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    head[i] = (i == 0 || !merge_same(m, sorted[i], sorted[i - 1])) ? 1u : 0u;
}

// seg[i] = inclusive count of heads up to i, minus one = index of the unique row
__global__ void k_merge_reduce(MergeSoA m, const uint32_t *__restrict__ sorted, const uint32_t *__restrict__ head, const uint32_t *__restrict__ seg_excl,
                               uint32_t n, MergeUnique u) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t e = sorted[i], s = seg_excl[i] + head[i] - 1u;
    if (head[i]) { u.tid[s] = m.tid[e]; u.start[s] = m.start[e]; u.end[s] = m.end[e]; }
    atomicAdd(&u.count[s], m.count[e]);
    atomicMin(&u.ts[s], m.ts[e]);
    atomicMax(&u.te[s], m.te[e]);
    atomicMin(&u.first[s], m.first[e]);
    atomicMax(&u.last_shard[s], m.shard[e]);
}

__global__ void k_merge_strand(MergeSoA m, const uint32_t *__restrict__ sorted, const uint32_t *__restrict__ head, const uint32_t *__restrict__ seg_excl,
                               uint32_t n, MergeUnique u) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t e = sorted[i], s = seg_excl[i] + head[i] - 1u;
    if (m.shard[e] == u.last_shard[s]) u.strand[s] = m.strand[e];       // one row per (key, shard): a single writer
}

__global__ void k_merge_rank(const uint32_t *__restrict__ by_first, uint32_t n, uint32_t *name_rank) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) name_rank[by_first[i]] = i;
}

// rows in final order, packed again (name_index = rank + 1; first/last = the merge's own order words)
__global__ void k_merge_pack(MergeUnique u, const uint32_t *__restrict__ order, const uint32_t *__restrict__ name_rank, uint32_t n, uint32_t *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = order[i];
    uint32_t *w = out + (size_t)i * 12;
    w[0] = u.tid[s]; w[1] = u.start[s]; w[2] = u.end[s]; w[3] = u.ts[s]; w[4] = u.te[s]; w[5] = u.count[s];
    w[6] = u.first[s]; w[7] = 0; w[8] = u.last_shard[s]; w[9] = 0; w[10] = u.strand[s]; w[11] = name_rank[s] + 1u;
}

// the same rows in the result table's host block layout (kernels.h table_block_rows), so that one copy fills the host table
__global__ void k_merge_table(MergeUnique u, const uint32_t *__restrict__ order, const uint32_t *__restrict__ name_rank, uint32_t n, uint32_t min_anchor, uint8_t *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = order[i];
    const size_t m = table_block_rows(n);
    const uint32_t start = u.start[s], end = u.end[s], ts = u.ts[s], te = u.te[s];
    uint64_t *q8 = (uint64_t *)out;
    q8[i] = (uint64_t)name_rank[s] + 1u; q8[m + i] = u.first[s]; q8[2 * m + i] = u.last_shard[s];
    uint32_t *q4 = (uint32_t *)(out + m * 24);
    q4[i] = u.tid[s]; q4[m + i] = start; q4[2 * m + i] = end; q4[3 * m + i] = ts; q4[4 * m + i] = te; q4[5 * m + i] = u.count[s];
    uint8_t *q1 = out + m * 48;
    q1[i] = (uint8_t)u.strand[s]; q1[m + i] = (uint32_t)(start - ts) >= min_anchor; q1[2 * m + i] = (uint32_t)(te - end) >= min_anchor;
}

static inline dim3 grid_for(uint32_t n) { return dim3((n + 255) / 256); }

void launch_merge_unpack(const uint32_t *rows, uint32_t stride_rows, uint32_t n_parts, const uint32_t *part_rows, const uint32_t *part_base, MergeSoA m, hipStream_t st) {
    const uint64_t total = (uint64_t)stride_rows * n_parts;
    if (!total) return;
    hipLaunchKernelGGL(k_merge_unpack, grid_for((uint32_t)total), dim3(256), 0, st, rows, stride_rows, n_parts, part_rows, part_base, m);
}
void launch_cols_to_packed(const uint32_t *cols, uint32_t n, uint32_t *out, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_cols_to_packed, grid_for(n), dim3(256), 0, st, cols, n, out);
}
void launch_merge_heads(MergeSoA m, const uint32_t *sorted, uint32_t n, uint32_t *head, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_merge_heads, grid_for(n), dim3(256), 0, st, m, sorted, n, head);
}
void launch_merge_reduce(MergeSoA m, const uint32_t *sorted, const uint32_t *head, const uint32_t *seg_excl, uint32_t n, MergeUnique u, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_merge_reduce, grid_for(n), dim3(256), 0, st, m, sorted, head, seg_excl, n, u);
    hipLaunchKernelGGL(k_merge_strand, grid_for(n), dim3(256), 0, st, m, sorted, head, seg_excl, n, u);
}
void launch_merge_rank(const uint32_t *by_first, uint32_t n, uint32_t *name_rank, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_merge_rank, grid_for(n), dim3(256), 0, st, by_first, n, name_rank);
}
void launch_merge_pack(MergeUnique u, const uint32_t *order, const uint32_t *name_rank, uint32_t n, uint32_t *out, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_merge_pack, grid_for(n), dim3(256), 0, st, u, order, name_rank, n, out);
}

void launch_merge_table(MergeUnique u, const uint32_t *order, const uint32_t *name_rank, uint32_t n, uint32_t min_anchor, uint8_t *out, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_merge_table, grid_for(n), dim3(256), 0, st, u, order, name_rank, n, min_anchor, out);
}

}  // namespace rgx
