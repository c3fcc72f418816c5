"""Multi-GPU exchange of SURVEY.md 8e: every rank extracts its own shard (contiguous BGZF member range or its
own coordinate slice), the per-rank unique-junction tables -- 48 bytes per row, at most a few hundred thousand
rows -- are exchanged with ONE all-gather (RCCL over xGMI when the backend is "nccl"; gloo on CPU in the unit
tests) and merged: sum counts, min/max thick bounds, earliest first_seen names the row, latest last_seen gives
the strand.  Shard order = rank order = file order, so the result does not depend on the number of GPUs.
No data-path collective exists anywhere else: alignments never leave the GPU that inflated them.
"""
import ctypes as C

from . import _ffi

ROW = 48  # RGX_PACKED_ROW_BYTES


class MergedTable(object):
    def __init__(self, ptr):
        self._table = ptr

    @property
    def table(self):
        return self._table

    @property
    def n(self):
        return self._table.contents.n

    def bed12(self, only_anchored=True):
        return _ffi.format_bed12(self._table, only_anchored)

    def barcodes_text(self, only_anchored=True):
        """Junction::print_barcodes per printed row (junctions_extractor.h:99-111) -- present when the shards were extracted with -b and merged by
        rgx_table_merge / rgx_extract_multi (rgx_table_merge_barcodes)."""
        lib = _ffi.lib()
        n = lib.rgx_table_format_barcodes(self._table, 1 if only_anchored else 0, None, 0)
        buf = C.create_string_buffer(n + 1)
        lib.rgx_table_format_barcodes(self._table, 1 if only_anchored else 0, buf, n)
        return buf.raw[:n]

    def __del__(self):
        try:
            if self._table:
                _ffi.lib().rgx_table_free(self._table)
                self._table = None
        except Exception:
            pass


def pack_table(table_ptr):
    lib = _ffi.lib()
    n = table_ptr.contents.n
    buf = (C.c_uint8 * max(1, n * ROW))()
    lib.rgx_table_pack(table_ptr, buf, n * ROW)
    return bytes(buf)[: n * ROW], n


def pack_barcodes(table_ptr):
    """the table's -b lists as bytes (rgx_table_pack_barcodes); b"" when it carries none"""
    lib = _ffi.lib()
    n = lib.rgx_table_pack_barcodes(table_ptr, None, 0)
    if not n:
        return b""
    buf = (C.c_uint8 * n)()
    lib.rgx_table_pack_barcodes(table_ptr, buf, n)
    return bytes(buf)


def _unpack_parts(parts, names_from, ended=None, barcodes=None):
    lib = _ffi.lib()
    ptrs = (C.POINTER(_ffi.JunctionTable) * len(parts))()
    for i, (b, n) in enumerate(parts):
        t = C.POINTER(_ffi.JunctionTable)()
        raw = (C.c_uint8 * max(1, len(b))).from_buffer_copy(b if len(b) else b"\0")
        lib.rgx_table_unpack(raw, n, names_from, C.byref(t))
        if ended is not None and ended[i]:
            t.contents.stream_ended = 1
        if barcodes is not None:
            bc = barcodes[i]
            if lib.rgx_table_unpack_barcodes(t, (C.c_uint8 * max(1, len(bc))).from_buffer_copy(bc if len(bc) else b"\0"), len(bc)) != 0:
                for j in range(i):
                    lib.rgx_table_free(ptrs[j])
                lib.rgx_table_free(t)
                raise RuntimeError("regtools_amd: shard %d's barcode block does not fit its rows" % i)
        ptrs[i] = t
    return ptrs


def merge_packed(parts, names_from, min_anchor, ended=None, barcodes=None):
    """parts: list of (bytes, n_rows) in shard order; names_from: any JunctionTable* carrying the contig table; ended: per shard, the
    table's stream_ended flag (the shards behind the first one that ended are ignored, as a sequential reader never gets there);
    barcodes: per shard, its pack_barcodes() bytes -- the merged table then carries the merged -b lists (rgx_table_merge does it)."""
    lib = _ffi.lib()
    ptrs = _unpack_parts(parts, names_from, ended, barcodes)
    out = C.POINTER(_ffi.JunctionTable)()
    err = C.create_string_buffer(256)
    rc = lib.rgx_table_merge(ptrs, len(parts), min_anchor, C.byref(out), err, len(err))
    for i in range(len(parts)):
        lib.rgx_table_free(ptrs[i])
    if rc:
        raise RuntimeError(err.value.decode())
    return MergedTable(out)


def merge_device(ctx, d_rows_ptr, stride_rows, part_rows, names_from, min_anchor):
    """Merge packed shard tables that are already in HBM (rgx_table_merge_device): shard g's rows at d_rows_ptr + g*stride_rows*48."""
    lib = _ffi.lib()
    out = C.POINTER(_ffi.JunctionTable)()
    err = C.create_string_buffer(256)
    sizes = (C.c_uint64 * len(part_rows))(*part_rows)
    rc = lib.rgx_table_merge_device(ctx._h, C.c_void_p(d_rows_ptr), stride_rows, sizes, len(part_rows), min_anchor, names_from, C.byref(out), err, len(err))
    if rc:
        raise RuntimeError(err.value.decode())
    return MergedTable(out)


def gather_and_merge(je_or_table, min_anchor=8, group=None, ctx=None):
    """All-gather the packed per-rank tables and merge them; every rank returns the same MergedTable.
    With the nccl (RCCL) backend and a device context the gathered rows stay in HBM and are merged there."""
    import torch
    import torch.distributed as dist

    table = je_or_table.table if hasattr(je_or_table, "table") else je_or_table
    if ctx is None:
        ctx = getattr(je_or_table, "_ctx", None)
    world = dist.get_world_size(group)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    if dev == "cuda" and ctx is not None:
        return _gather_and_merge_device(table, ctx, min_anchor, group, world)
    payload, n = pack_table(table)
    bc = pack_barcodes(table)                     # -b: the rank's barcode lists travel behind its rows, in the same collective
    sizes = [torch.zeros(3, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([n, int(table.contents.stream_ended), len(bc)], dtype=torch.int64, device=dev), group=group)
    ended = [bool(int(s[1].item())) for s in sizes]
    bc_sizes = [int(s[2].item()) for s in sizes]
    sizes = [int(s[0].item()) for s in sizes]
    cap = max(1, max(sz * ROW + b for sz, b in zip(sizes, bc_sizes)))
    local = torch.zeros(cap, dtype=torch.uint8, device=dev)
    if n or bc:
        local[: n * ROW + len(bc)].copy_(torch.frombuffer(bytearray(payload + bc), dtype=torch.uint8))
    gathered = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(gathered, local, group=group)          # the one collective of the whole job
    parts, bcs = [], []
    for r in range(world):
        raw = gathered[r][: sizes[r] * ROW + bc_sizes[r]].cpu().numpy().tobytes()
        parts.append((raw[: sizes[r] * ROW], sizes[r]))
        bcs.append(raw[sizes[r] * ROW:])
    return merge_packed(parts, table, min_anchor, ended, bcs if all(bc_sizes) else None)


_pinned = {}


def _gather_and_merge_device(table, ctx, min_anchor, group, world):
    """RCCL path: rows are packed straight into pinned memory, all-gathered into one HBM buffer and merged there."""
    import torch
    import torch.distributed as dist

    lib = _ffi.lib()
    n = int(table.contents.n)
    sizes_t = torch.zeros(2 * world, dtype=torch.int64, device="cuda")
    dist.all_gather_into_tensor(sizes_t, torch.tensor([n, int(table.contents.stream_ended)], dtype=torch.int64, device="cuda"), group=group)
    flat = [int(x) for x in sizes_t.tolist()]
    sizes = flat[0::2]
    merge_sizes = list(sizes)
    for r in range(world):                       # the shards behind one whose record stream ended contribute nothing
        if flat[2 * r + 1]:
            merge_sizes[r + 1:] = [0] * (world - r - 1)
            break
    stride = max(1, max(sizes))
    cap = stride * ROW
    local = torch.empty(cap, dtype=torch.uint8, device="cuda")
    # the rows of the extraction that produced `table` are still in HBM: pack them there (no host round trip) ...
    err = C.create_string_buffer(256)
    rc = lib.rgx_last_table_pack_device(ctx._h, table, C.c_void_p(local.data_ptr()), stride, err, len(err))
    if rc != 0:
        # ... unless the context has run something else since: then from the host copy
        host = _pinned.get("rows")
        if host is None or host.numel() < cap:
            host = _pinned["rows"] = torch.empty(cap + cap // 8, dtype=torch.uint8, pin_memory=True)
        if n:
            lib.rgx_table_pack(table, C.c_void_p(host.data_ptr()), n * ROW)
        local[: n * ROW].copy_(host[: n * ROW], non_blocking=True)
    big = torch.empty(cap * world, dtype=torch.uint8, device="cuda")
    dist.all_gather_into_tensor(big, local, group=group)      # the one data collective of the whole job; the rows stay in HBM
    torch.cuda.current_stream().synchronize()
    merged = merge_device(ctx, big.data_ptr(), stride, merge_sizes, table, min_anchor)
    bc = pack_barcodes(table)
    if bc:
        # -b: the barcode lists are host data (rgx_junction_table.bc_*): a second, small all-gather of bytes, then the host-side merge onto the
        # device-merged rows (rgx_table_merge_barcodes)
        bsz = torch.zeros(world, dtype=torch.int64, device="cuda")
        dist.all_gather_into_tensor(bsz, torch.tensor([len(bc)], dtype=torch.int64, device="cuda"), group=group)
        bsz = [int(x) for x in bsz.tolist()]
        bcap = max(bsz)
        mine = torch.zeros(bcap, dtype=torch.uint8, device="cuda")
        mine[: len(bc)].copy_(torch.frombuffer(bytearray(bc), dtype=torch.uint8))
        allbc = torch.empty(bcap * world, dtype=torch.uint8, device="cuda")
        dist.all_gather_into_tensor(allbc, mine, group=group)
        rows_host = big.cpu().numpy().tobytes()
        bc_host = allbc.cpu().numpy().tobytes()
        parts = [(rows_host[r * cap: r * cap + merge_sizes[r] * ROW], merge_sizes[r]) for r in range(world)]
        bcs = [bc_host[r * bcap: r * bcap + bsz[r]] if merge_sizes[r] == sizes[r] else None for r in range(world)]
        keep = [r for r in range(world) if bcs[r] is not None]
        ptrs = _unpack_parts([parts[r] for r in keep], table, None, [bcs[r] for r in keep])
        err = C.create_string_buffer(256)
        rc = lib.rgx_table_merge_barcodes(ptrs, len(keep), merged.table, err, len(err))
        for i in range(len(keep)):
            lib.rgx_table_free(ptrs[i])
        if rc:
            raise RuntimeError(err.value.decode())
    return merged
