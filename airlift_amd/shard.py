"""Read sharding across the GPUs of one node.

Fragments are independent (the reference's kt_for over n_frag, map.c:592), so rank r of R maps the contiguous range
[floor(r*N/R), floor((r+1)*N/R)) of every batch and the full index is replicated per GPU.  The only exchange step is one
all-gather of {n_records, n_bytes} (16 bytes per rank; RCCL over xGMI when the backend is "nccl") whose exclusive prefix
gives each rank the offset of its block in the merged output -- output order then equals input order like the
reference's serial writer (map.c:601-644)."""
import torch


def frag_range(n_frag, rank, world):
    return (rank * n_frag) // world, ((rank + 1) * n_frag) // world


def output_offsets(n_records, n_bytes, rank, world, device=None, dist=None):
    """Returns (record_offset, byte_offset, total_records, total_bytes) for this rank."""
    if world == 1 or dist is None:
        return 0, 0, int(n_records), int(n_bytes)
    mine = torch.tensor([int(n_records), int(n_bytes)], dtype=torch.int64, device=device)
    allr = torch.empty(2 * world, dtype=torch.int64, device=device)
    if hasattr(dist, "all_gather_into_tensor") and (device is not None and torch.device(device).type == "cuda"):
        dist.all_gather_into_tensor(allr, mine)
    else:
        parts = [torch.empty(2, dtype=torch.int64, device=device) for _ in range(world)]
        dist.all_gather(parts, mine)
        allr = torch.cat(parts)
    allr = allr.view(world, 2).cpu()
    excl = torch.cumsum(allr, 0) - allr
    return int(excl[rank, 0]), int(excl[rank, 1]), int(allr[:, 0].sum()), int(allr[:, 1].sum())
