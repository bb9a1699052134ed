"""Multi-GPU SpMM: 1-D row partition of A, dense B replicated by one RCCL exchange.

The reference is single-GPU; this is the north_star's multi-GPU
design. Output rows are independent, so A is cut into contiguous row ranges with
(nearly) equal nnz, every rank keeps its rows' CSR (rebased), the FULL dense B and
its own rows of C. The only exchange is getting B onto every rank:

    exchange_dense(B_shard)   each rank owns K/world rows of B  -> all_gather
    broadcast_dense(B, src)   one rank owns B                   -> broadcast

No reduction, no halo. One process per GPU; backend "nccl" is RCCL on ROCm and runs
over xGMI. Collectives are issued in chunks of < 2^31 elements.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import graphs
from . import spmm as _spmm

_MAX_ELEMS = (1 << 31) - 1024


def shard_csr(rowptr, colind, val, cut, rank):
    """Rows [cut[rank], cut[rank+1]) of a host (numpy) CSR, rowptr rebased to 0."""
    r0, r1 = int(cut[rank]), int(cut[rank + 1])
    p0, p1 = int(rowptr[r0]), int(rowptr[r1])
    local_ptr = (rowptr[r0:r1 + 1] - rowptr[r0]).astype(np.int32)
    local_val = None if val is None else np.ascontiguousarray(val[p0:p1])
    return local_ptr, np.ascontiguousarray(colind[p0:p1]), local_val, (r0, r1)


def partition_rows(rowptr, world):
    """nnz-balanced cut points via the C ABI (gespmm_row_partition)."""
    return graphs.row_partition(rowptr, world)


def exchange_dense(B_shard, row_counts=None, group=None):
    """all_gather of B row-shards into the full K x N matrix on every rank (every xGMI link busy, no root).
    row_counts[r] = rows owned by rank r (equal shards if None)."""
    world = dist.get_world_size(group)
    N = B_shard.shape[1]
    if row_counts is None:
        row_counts = [B_shard.shape[0]] * world
    K = int(sum(row_counts))
    full = torch.empty((K, N), dtype=B_shard.dtype, device=B_shard.device)
    _gather_rows(full, B_shard.contiguous(), [int(c) for c in row_counts], group)
    return full


def broadcast_dense(B, K, N, src=0, group=None, device=None):
    """B (K x N) lives on rank `src`; returns the replicated copy on every rank."""
    if dist.get_rank(group) == src:
        full = B.contiguous()
    else:
        full = torch.empty((K, N), dtype=torch.float32, device=device)
    flat = full.view(-1)
    for a in range(0, flat.numel(), _MAX_ELEMS):
        dist.broadcast(flat[a:a + _MAX_ELEMS], src=src, group=group)
    return full


def local_spmm(local_rowptr, local_colind, local_val, B_full, variant=-1, out=None):
    """This rank's rows of C = A @ B (device tensors)."""
    if local_val is None:
        return _spmm.csr_spmm_no_edge_value(local_rowptr, local_colind, B_full, variant=variant, out=out)
    return _spmm.csr_spmm(local_rowptr, local_colind, local_val, B_full, variant=variant, out=out)


def _gather_rows(full, shard, row_counts, group=None):
    """full[K x w] <- every rank's rows (rank r owns row_counts[r] rows starting at sum(row_counts[:r])).
    Equal shards: all_gather in row chunks of < 2^31 elements through a staging block (the collective wants one
    contiguous [world, rows, w] output; the rows of one rank are contiguous in `full`, the ranks are not adjacent
    chunk by chunk). Ragged shards: one broadcast per owner and chunk."""
    world = dist.get_world_size(group)
    w = shard.shape[1]
    offs = np.concatenate([[0], np.cumsum(row_counts)]).astype(np.int64)
    if all(c == row_counts[0] for c in row_counts):
        kr = int(row_counts[0])
        if world * kr * w <= _MAX_ELEMS:
            dist.all_gather_into_tensor(full, shard, group=group)
            return
        rc = max(1, _MAX_ELEMS // max(world * w, 1))
        stage = torch.empty((world * min(rc, kr), w), dtype=shard.dtype, device=shard.device)
        for a in range(0, kr, rc):
            n = min(rc, kr - a)
            st = stage if world * n == stage.shape[0] else torch.empty((world * n, w), dtype=shard.dtype, device=shard.device)
            dist.all_gather_into_tensor(st, shard[a:a + n].contiguous(), group=group)  # rank r's rows at st[r*n : (r+1)*n]
            for r in range(world):
                full[int(offs[r]) + a:int(offs[r]) + a + n].copy_(st[r * n:(r + 1) * n])
        return
    rows_per_chunk = max(1, _MAX_ELEMS // max(w, 1))
    me = dist.get_rank(group)
    for r in range(world):
        for a in range(int(offs[r]), int(offs[r + 1]), rows_per_chunk):
            b = min(a + rows_per_chunk, int(offs[r + 1]))
            view = full[a:b]
            if me == r:
                view.copy_(shard[a - int(offs[r]):b - int(offs[r])])
            dist.broadcast(view, src=dist.get_global_rank(group, r) if group is not None else r, group=group)


class PanelPipeline:
    """C_loc = A_loc @ B with B owned as row shards and the exchange INSIDE the product: B travels in column panels,
    the all-gather of panel p+1 runs on a second stream while panel p is multiplied (the RMAT-26 exchange is >= 64 ms
    over 7 xGMI links against ~20 ms of kernel per GPU, so the pipeline is exchange-bound and the multiply is hidden).
    Panels are separate contiguous K x w (B) and M_loc x w (C) matrices; C's panels are returned in order.

    ``product(rowptr, colind, val, Bpanel, out)`` defaults to the HIP path; the CPU test-suite injects its checker."""

    def __init__(self, rowptr, colind, val, K, row_counts, panel_widths, device, variant=-1, group=None, product=None):
        self.rowptr, self.colind, self.val = rowptr, colind, val
        self.K, self.row_counts, self.widths, self.group = int(K), [int(c) for c in row_counts], list(panel_widths), group
        self.device = torch.device(device)
        self.M = rowptr.numel() - 1
        self.cuda = self.device.type == "cuda"
        wmax = max(self.widths)
        self.flat = [torch.empty(self.K * wmax, dtype=torch.float32, device=self.device) for _ in range(2)]
        self.C = [torch.empty((self.M, w), dtype=torch.float32, device=self.device) for w in self.widths]
        self.comm = torch.cuda.Stream(self.device) if self.cuda else None
        if product is None:
            plans = {}

            def product(rowptr, colind, val, Bp, out):
                w = Bp.shape[1]
                if w not in plans:
                    plans[w] = _spmm.SpmmPlan(rowptr, colind, self.K, w, variant=variant, values=val, reorder=False)
                if val is None:
                    return _spmm.csr_spmm_no_edge_value(rowptr, colind, Bp, variant=variant, out=out, plan=plans[w])
                return _spmm.csr_spmm(rowptr, colind, val, Bp, variant=variant, out=out, plan=plans[w])

        self.product = product

    def _panel_buffer(self, p):
        return self.flat[p % 2][:self.K * self.widths[p]].view(self.K, self.widths[p])

    def run(self, shard_panels):
        P = len(self.widths)
        if not self.cuda:  # no streams on the host: same data flow, serial
            for p in range(P):
                buf = self._panel_buffer(p)
                _gather_rows(buf, shard_panels[p], self.row_counts, self.group)
                self.product(self.rowptr, self.colind, self.val, buf, self.C[p])
            return self.C
        cur = torch.cuda.current_stream(self.device)
        arrived = [torch.cuda.Event() for _ in range(P)]
        consumed = [torch.cuda.Event() for _ in range(P)]
        self.comm.wait_stream(cur)  # the shards were produced on the current stream
        with torch.cuda.stream(self.comm):
            _gather_rows(self._panel_buffer(0), shard_panels[0], self.row_counts, self.group)
            arrived[0].record(self.comm)
        for p in range(P):
            if p + 1 < P:
                if p >= 1:
                    self.comm.wait_event(consumed[p - 1])  # buffer (p+1) % 2 was read by the product of panel p-1
                with torch.cuda.stream(self.comm):
                    _gather_rows(self._panel_buffer(p + 1), shard_panels[p + 1], self.row_counts, self.group)
                    arrived[p + 1].record(self.comm)
            cur.wait_event(arrived[p])
            self.product(self.rowptr, self.colind, self.val, self._panel_buffer(p), self.C[p])
            consumed[p].record(cur)
        return self.C
