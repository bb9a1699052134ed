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
    """all_gather of B row-shards into the full K x N matrix on every rank.
    row_counts[r] = rows owned by rank r (equal shards if None)."""
    world = dist.get_world_size(group)
    N = B_shard.shape[1]
    if row_counts is None:
        row_counts = [B_shard.shape[0]] * world
    K = int(sum(row_counts))
    full = torch.empty((K, N), dtype=B_shard.dtype, device=B_shard.device)
    offs = np.concatenate([[0], np.cumsum(row_counts)])
    equal = all(c == row_counts[0] for c in row_counts)
    if equal and B_shard.numel() <= _MAX_ELEMS:
        dist.all_gather_into_tensor(full, B_shard.contiguous(), group=group)
        return full
    # ragged or very large: broadcast each owner's shard in row chunks
    rows_per_chunk = max(1, _MAX_ELEMS // max(N, 1))
    for r in range(world):
        for a in range(int(offs[r]), int(offs[r + 1]), rows_per_chunk):
            b = min(a + rows_per_chunk, int(offs[r + 1]))
            view = full[a:b]
            if dist.get_rank(group) == r:
                view.copy_(B_shard[a - int(offs[r]):b - int(offs[r])])
            dist.broadcast(view, src=dist.get_global_rank(group, r) if group is not None else r, group=group)
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
