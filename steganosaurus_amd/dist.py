"""Multi-GPU plumbing of the batched path: independent images are sharded over ranks
(one process per GPU), the shared bin list is broadcast once, results/timings are reduced.
There is no collective on the data path.  Works with backend "nccl" (= RCCL over xGMI on
MI355X nodes) and with "gloo" (CPU tests)."""
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard(n_total: int, rank: int, world_size: int):
    """Contiguous block of images [lo, hi) owned by `rank` (remainder spread over the first ranks)."""
    base, rem = divmod(n_total, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_bins(bins_u8: torch.Tensor, src: int = 0, bit_index: torch.Tensor = None):
    """Broadcast the (n_bits, 8) uint8 view of the tfft_bin list computed on `src` (8 B x n_bits) and, when the
    list was put in address order (tfft_bins_sort), the int64 bit index that goes with it (tfft_set_bit_index)."""
    _, ws = world()
    if ws > 1:
        dist.broadcast(bins_u8, src=src)
        if bit_index is not None:
            dist.broadcast(bit_index, src=src)
    return bins_u8 if bit_index is None else (bins_u8, bit_index)


def max_over_ranks(seconds: float, device=None) -> float:
    _, ws = world()
    if ws == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_bits(raw_u8: torch.Tensor, dst: int = 0):
    """Gather every rank's extracted bit matrix (n_images_rank, n_bits) on `dst` (equal shard sizes)."""
    rank, ws = world()
    if ws == 1:
        return [raw_u8]
    out = [torch.empty_like(raw_u8) for _ in range(ws)] if rank == dst else None
    dist.gather(raw_u8, out, dst=dst)
    return out
