"""Multi-GPU layout of the matching path: image pairs are independent, so the pair list is cut into
contiguous per-rank blocks (the reference slices its file list by rank the same way,
homodataset/HomoDataset.py:40-45) and every rank matches its block on its own GPU.  There is NO
data-path collective; torch.distributed (RCCL on GPUs, gloo in CPU tests) is used only to collect the
small per-pair summaries / timings at the end."""
from typing import Callable, List, Sequence, Tuple


def shard_bounds(num_items: int, world_size: int, rank: int) -> Tuple[int, int]:
    """[begin, end) of `rank`'s contiguous block; blocks differ by at most one item and cover the list."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError(f'bad rank {rank} / world_size {world_size}')
    base, extra = divmod(num_items, world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def run_sharded(pairs: Sequence, match_fn: Callable, batch: int = 1, group=None) -> List:
    """Runs match_fn(list_of_pairs) -> list_of_summaries over this rank's block in batches and returns the
    summaries of ALL pairs in list order on every rank (gathered as python objects; they are small)."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    lo, hi = shard_bounds(len(pairs), world, rank)
    local = []
    for s in range(lo, hi, batch):
        out = match_fn(list(pairs[s:min(s + batch, hi)]))
        if len(out) != min(s + batch, hi) - s:
            raise RuntimeError('match_fn must return one summary per pair')
        local.extend(out)
    if world == 1:
        return local
    parts = [None] * world
    dist.all_gather_object(parts, local, group=group)
    return [x for part in parts for x in part]
