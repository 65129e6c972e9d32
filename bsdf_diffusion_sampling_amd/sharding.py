"""Multi-GPU sharding of shading-query batches (new design: the reference has no
multi-GPU code at all).

Every query is independent, so a batch shards embarrassingly: rank r of W owns the
contiguous range ``shard_range(N, r, W)``; weights (<= 90 KB) are replicated; there is NO
collective on the data path.  The Philox counter is ``offset + global query index``, so a
sharded run draws exactly the samples a single-GPU run would.  The only communication
is the optional final concatenation of ``(wo[3], pdf)`` = 16 B/query:

  * ``gather_to_root``: one ``torch.distributed.gather`` — lands on root's 7 xGMI links
    in parallel (the cheap choice when one consumer wants the whole wavefront);
  * ``all_gather``: ring all-gather (every rank gets everything; per-link bound).

Both run on whatever process group is current: ``nccl`` (= RCCL over xGMI) on the GPU
node, ``gloo`` in the CPU tests.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) of rank's shard; sizes differ by at most one, earlier ranks larger."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_sizes(n: int, world: int) -> List[int]:
    return [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]


def bucket_by_material(material_id: torch.Tensor, n_materials: int):
    """Config 4 (mixed-material batches): stable sort of queries by material id.
    Returns (perm, counts): ``perm`` gathers queries into contiguous per-material runs,
    ``counts[m]`` is the run length — one kernel launch per non-empty run.
    CUDA tensors with <= 64 materials go through the native stable counting sort
    (``bsdfd_bucket_by_material``, csrc/bucket.hip: 0.3 ms for 16 Mi ids vs 1.7 ms for torch.argsort);
    CPU tensors (the gloo tests) use torch."""
    if material_id.is_cuda and n_materials <= 64:
        import ctypes as C
        from . import _lib
        ids = material_id.contiguous()
        n = ids.shape[0]
        L = _lib.lib()
        perm = torch.empty(n, dtype=torch.int64, device=ids.device)
        counts = torch.empty(n_materials, dtype=torch.int64, device=ids.device)
        ws = torch.empty(max(int(L.bsdfd_bucket_workspace_bytes(n, n_materials)), 1), dtype=torch.uint8, device=ids.device)
        with torch.cuda.device(ids.device):
            _lib.check(L.bsdfd_bucket_by_material(C.c_void_p(ids.data_ptr()), n, n_materials, C.c_void_p(perm.data_ptr()),
                                                  C.c_void_p(counts.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(),
                                                  C.c_void_p(torch.cuda.current_stream(ids.device).cuda_stream)))
        return perm, counts
    perm = torch.argsort(material_id, stable=True)
    counts = torch.bincount(material_id, minlength=n_materials)
    return perm, counts


def pack_result(wo: torch.Tensor, pdf: torch.Tensor) -> torch.Tensor:
    """[n,3] + [n] -> [n,4] (16 B/query), the unit that crosses xGMI."""
    return torch.cat([wo, pdf[:, None]], dim=1).contiguous()


def gather_to_root(local: torch.Tensor, n_total: int, root: int = 0, group=None) -> Optional[torch.Tensor]:
    """Concatenate the ranks' [n_r, C] shards on ``root`` in rank order (None elsewhere)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = shard_sizes(n_total, world)
    if local.shape[0] != sizes[rank]:
        raise ValueError(f"rank {rank}: shard has {local.shape[0]} rows, expected {sizes[rank]}")
    cols = local.shape[1]
    if len(set(sizes)) == 1:
        outs = [torch.empty((sizes[0], cols), dtype=local.dtype, device=local.device) for _ in range(world)] \
            if rank == root else None
        dist.gather(local.contiguous(), outs, dst=root, group=group)
        return torch.cat(outs, 0) if rank == root else None
    # ragged: pad to the largest shard (sizes differ by at most one row)
    m = max(sizes)
    padded = torch.zeros((m, cols), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    outs = [torch.empty_like(padded) for _ in range(world)] if rank == root else None
    dist.gather(padded, outs, dst=root, group=group)
    if rank != root:
        return None
    return torch.cat([o[:s] for o, s in zip(outs, sizes)], 0)


def all_gather(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = shard_sizes(n_total, world)
    if local.shape[0] != sizes[rank]:
        raise ValueError(f"rank {rank}: shard has {local.shape[0]} rows, expected {sizes[rank]}")
    m, cols = max(sizes), local.shape[1]
    padded = torch.zeros((m, cols), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    out = torch.empty((world * m, cols), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    if len(set(sizes)) == 1:
        return out
    return torch.cat([out[r * m: r * m + s] for r, s in enumerate(sizes)], 0)


class ShardedPlugin:
    """Run a plugin's ``sample_t`` / ``pdf_t`` on this rank's shard of a global batch."""

    def __init__(self, plugin, group=None):
        self.plugin = plugin
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def local_range(self, n_total: int) -> Tuple[int, int]:
        return shard_range(n_total, self.rank, self.world)

    def sample_local(self, wi_local: torch.Tensor, n_total: int, seed: int, x0_local=None):
        lo, _ = self.local_range(n_total)
        return self.plugin.sample_t(wi_local, x0=x0_local, seed=seed, offset=lo)

    def sample_gathered(self, wi_local: torch.Tensor, n_total: int, seed: int, root: int = 0):
        wo, pdf = self.sample_local(wi_local, n_total, seed)
        if self.world == 1:
            return pack_result(wo, pdf)
        return gather_to_root(pack_result(wo, pdf), n_total, root=root, group=self.group)
