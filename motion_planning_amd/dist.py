"""Scenario sharding across the GPUs of one node and the result gather.

A plan never spans GPUs: scenarios are independent, so rank r of W takes one
contiguous block of scenarios and the only communication is one all-gather of
a small fixed-size record per scenario (RCCL over xGMI on GPUs; the same code
runs over gloo on CPU tensors, which is how it is tested without GPUs).
"""
import torch
import torch.distributed as dist


def shard_range(n_total, rank, world):
    """Contiguous block [lo, hi) of rank: sizes differ by at most one."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def record_width(n_s, n_t):
    """int32 words per scenario: path_idx[n_s] status layers_done best_cost(2)
    st_path[n_t] st_len."""
    return n_s + 2 + 2 + n_t + 1


def pack_results(path_idx, status, layers_done, best_cost, st_path, st_len, out=None):
    """One int32 row per scenario (best_cost travels as its two 32-bit halves)."""
    B, n_s = path_idx.shape
    n_t = st_path.shape[1]
    if out is None:
        out = torch.empty((B, record_width(n_s, n_t)), dtype=torch.int32, device=path_idx.device)
    out[:, :n_s] = path_idx
    out[:, n_s] = status
    out[:, n_s + 1] = layers_done
    out[:, n_s + 2:n_s + 4] = best_cost.contiguous().view(torch.int32).view(B, 2)
    out[:, n_s + 4:n_s + 4 + n_t] = st_path
    out[:, n_s + 4 + n_t] = st_len
    return out


def unpack_results(rec, n_s, n_t):
    rec = rec.contiguous()
    return dict(path_idx=rec[:, :n_s], status=rec[:, n_s], layers_done=rec[:, n_s + 1],
                best_cost=rec[:, n_s + 2:n_s + 4].contiguous().view(torch.float64).view(-1),
                st_path=rec[:, n_s + 4:n_s + 4 + n_t], st_len=rec[:, n_s + 4 + n_t])


def gather_results(rec, out=None, group=None):
    """All ranks receive every rank's records in rank order.  Equal shard sizes
    use one all_gather_into_tensor (a single direct exchange: the payload is a
    few hundred bytes per scenario); ragged shards are padded to the largest."""
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


def gather_results_same_size(rec, out, group=None):
    """Hot-loop form: shard sizes known to be equal, output preallocated."""
    dist.all_gather_into_tensor(out, rec, group=group)
    return out
