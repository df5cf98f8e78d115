"""Scenario sharding across the GPUs of one node and the result gather.

A plan never spans GPUs: scenarios are independent, so rank r of W takes one
contiguous block of scenarios and the only communication is one all-gather of
a small fixed-size record per scenario (RCCL over xGMI on GPUs; the same code
runs over gloo on CPU tensors, which is how it is tested without GPUs).

Record layout (bytes, little endian; the library's mp_pack_results writes the
same on the device):
  [0,8) best_cost f64 | [8,10) layers_done u16 | [10] status i8 | [11] st_len u8
  | st_path i16[n_t] | path_idx i16[n_s] | zero padding to a multiple of 8
"""
import ctypes as C

import torch
import torch.distributed as dist


def shard_range(n_total, rank, world):
    """Contiguous block [lo, hi) of rank: sizes differ by at most one."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def record_width(n_s, n_t):
    """Bytes per scenario record."""
    return (12 + 2 * n_t + 2 * n_s + 7) & ~7


def pack_results(path_idx, status, layers_done, best_cost, st_path, st_len, out=None):
    """One record per scenario, uint8 [B, record_width].  CUDA tensors are packed by
    the library's kernel (one launch), CPU tensors with torch."""
    B, n_s = path_idx.shape
    n_t = st_path.shape[1]
    width = record_width(n_s, n_t)
    if out is None:
        out = torch.empty((B, width), dtype=torch.uint8, device=path_idx.device)
    if path_idx.is_cuda:
        from . import _lib
        from .batch import _p, _stream
        # the kernel reads int32 / float64 through raw pointers
        args = [t.to(torch.float64 if t is best_cost else torch.int32).contiguous()
                for t in (path_idx, status, layers_done, best_cost, st_path, st_len)]
        # (pinned host memory is mapped into the device's address space: the kernel can
        # store the records straight into it, which saves a separate device-to-host copy)
        if not ((out.is_cuda or out.is_pinned()) and out.dtype == torch.uint8 and out.is_contiguous()
                and out.shape == (B, width)):
            raise ValueError("pack_results: out must be a contiguous uint8 tensor [%d, %d] in device or pinned host "
                             "memory" % (B, width))
        _lib.check(_lib.lib().mp_pack_results(B, n_s, n_t, *[_p(t) for t in args], _p(out), _stream()),
                   "mp_pack_results")
        return out
    out.zero_()
    out[:, 0:8] = best_cost.contiguous().view(torch.uint8).view(B, 8)
    out[:, 8:10] = layers_done.to(torch.int16).contiguous().view(torch.uint8).view(B, 2)
    out[:, 10] = status.to(torch.int8).view(torch.uint8)
    out[:, 11] = st_len.to(torch.uint8)
    out[:, 12:12 + 2 * n_t] = st_path.to(torch.int16).contiguous().view(torch.uint8).view(B, 2 * n_t)
    out[:, 12 + 2 * n_t:12 + 2 * n_t + 2 * n_s] = path_idx.to(torch.int16).contiguous().view(torch.uint8).view(B, 2 * n_s)
    return out


def unpack_results(rec, n_s, n_t):
    """Records -> dict of int32 / float64 tensors (path_idx, status, layers_done,
    best_cost, st_path, st_len)."""
    rec = rec.contiguous()
    B = rec.shape[0]
    path = rec[:, 12 + 2 * n_t:12 + 2 * n_t + 2 * n_s].contiguous().view(torch.int16).view(B, n_s).to(torch.int32)
    return dict(path_idx=path,
                status=rec[:, 10].contiguous().view(torch.int8).to(torch.int32),
                layers_done=rec[:, 8:10].contiguous().view(torch.int16).view(B).to(torch.int32) & 0xFFFF,
                best_cost=rec[:, 0:8].contiguous().view(torch.float64).view(B),
                st_path=rec[:, 12:12 + 2 * n_t].contiguous().view(torch.int16).view(B, n_t).to(torch.int32),
                st_len=rec[:, 11].to(torch.int32))


def gather_results(rec, out=None, group=None):
    """All ranks receive every rank's records in rank order.  Equal shard sizes
    use one all_gather_into_tensor (a single direct exchange: the payload is about a
    hundred bytes per scenario); ragged shards are padded to the largest."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return rec
    n = torch.tensor([rec.shape[0]], dtype=torch.int64, device=rec.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s) for s in sizes]
    if len(set(sizes)) == 1:
        if out is None:
            out = torch.empty((world * sizes[0], rec.shape[1]), dtype=rec.dtype, device=rec.device)
        dist.all_gather_into_tensor(out, rec.contiguous(), group=group)
        return out
    m = max(sizes)
    padded = torch.zeros((m, rec.shape[1]), dtype=rec.dtype, device=rec.device)
    padded[:rec.shape[0]] = rec
    buf = torch.empty((world * m, rec.shape[1]), dtype=rec.dtype, device=rec.device)
    dist.all_gather_into_tensor(buf, padded, group=group)
    return torch.cat([buf[r * m:r * m + sizes[r]] for r in range(world)])


def gather_results_same_size(rec, out, group=None, async_op=False):
    """Hot-loop form: shard sizes known to be equal, output preallocated.  With
    async_op the collective runs on the process group's own stream and the returned
    handle is waited on later, so the next batch's kernels overlap the exchange."""
    return dist.all_gather_into_tensor(out, rec, group=group, async_op=async_op)
