"""Multi-GPU sharding of the batched path: one process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The path shards by independent units (30 s chunks): no collective in
the data path.  Only two exchanges exist:
  * load time:  rank 0 builds its model (the packed weight blob is then in its HBM) and `broadcast_blob_dev` sends that
                allocation to every other rank's HBM over RCCL — no host image, no second upload (`broadcast_blob`, the
                host-image form, stays for the gloo seam of the CPU tests);
  * per batch:  `gather_results` brings the fixed-size result records (ids, length, score,
                no_speech_prob) to rank 0, in rank order, so the output order equals the
                serial reference's order.
PyTorch is plumbing here (device memory for the collective); the engine itself never sees a
torch type (the blob pointer crosses the C ABI as void*).
"""
from typing import List, Optional, Sequence, Tuple

import numpy as np


def partition(n_items: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous block partition of n_items over `world` ranks: rank r gets [lo, hi).
    Blocks differ by at most one item and preserve order (results concatenate without a sort)."""
    base, rem = divmod(n_items, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


def _dist():
    import torch.distributed as dist
    return dist


def broadcast_blob(blob: Optional[np.ndarray], rank: int, local_rank: int):
    """rank 0 passes the packed blob (uint8 ndarray); returns a uint8 torch tensor holding the blob on
    this rank's device (GPU for nccl, CPU for gloo)."""
    import torch
    dist = _dist()
    on_gpu = dist.get_backend() == "nccl"
    dev = torch.device("cuda", local_rank) if on_gpu else torch.device("cpu")
    size = torch.tensor([blob.shape[0] if rank == 0 else 0], dtype=torch.int64, device=dev)
    dist.broadcast(size, src=0)
    n = int(size.item())
    if rank == 0:
        t = torch.from_numpy(blob).to(dev)
    else:
        t = torch.empty(n, dtype=torch.uint8, device=dev)
    dist.broadcast(t, src=0)
    return t


class _DeviceBytes:
    """`n` bytes of device memory at `ptr` as a __cuda_array_interface__ object (torch.as_tensor wraps it without a copy)"""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def broadcast_blob_dev(blob_dev: Optional[Tuple[int, int]], rank: int, local_rank: int):
    """Round 6: the broadcast WITHOUT a host bounce.  Rank 0 has already built its model (the packed blob lives in ITS HBM:
    `Whisper.blob()` = fw_model_blob's pointer and size) and passes (ptr, nbytes); RCCL reads straight from that allocation
    (wrapped as a tensor, no copy) and every other rank receives into a fresh device buffer.  Returns (tensor, nbytes): on
    rank 0 the zero-copy view of the model's own blob, elsewhere the received buffer (keep it alive as long as the model
    built on it).  nccl only — the CPU seam (gloo) keeps `broadcast_blob` on the host image."""
    import torch
    dist = _dist()
    assert dist.get_backend() == "nccl", "broadcast_blob_dev needs device memory on every rank (backend nccl = RCCL)"
    dev = torch.device("cuda", local_rank)
    size = torch.tensor([blob_dev[1] if rank == 0 else 0], dtype=torch.int64, device=dev)
    dist.broadcast(size, src=0)
    n = int(size.item())
    if rank == 0:
        t = torch.as_tensor(_DeviceBytes(blob_dev[0], n), device=dev)
        assert t.data_ptr() == blob_dev[0], "the wrapped blob must alias the model's allocation (no copy)"
    else:
        t = torch.empty(n, dtype=torch.uint8, device=dev)
    dist.broadcast(t, src=0)
    return t, n


RECORD_EXTRA = 5   # int32 words besides the ids: length + two float64


def encode_records(results, max_len: int) -> np.ndarray:
    """fixed-size int32 record per chunk: [len, ids[max_len], score (float64, 2 words), no_speech_prob (float64,
    2 words)] — float64 so that rank 0 sees bit-for-bit what a serial run computes (avg_logprob is a Python float)"""
    rec = np.zeros((len(results), max_len + RECORD_EXTRA), dtype=np.int32)
    for i, r in enumerate(results):
        ids = r.sequences_ids[0][:max_len]
        rec[i, 0] = len(ids)
        rec[i, 1:1 + len(ids)] = ids
        rec[i, max_len + 1:max_len + 3] = np.array([r.scores[0] if r.scores else 0.0], dtype=np.float64).view(np.int32)
        rec[i, max_len + 3:max_len + 5] = np.array([r.no_speech_prob], dtype=np.float64).view(np.int32)
    return rec


def decode_records(rec: np.ndarray, max_len: int):
    out = []
    for row in rec:
        n = int(row[0])
        row = np.ascontiguousarray(row)
        out.append((row[1:1 + n].tolist(), float(row[max_len + 1:max_len + 3].view(np.float64)[0]),
                    float(row[max_len + 3:max_len + 5].view(np.float64)[0])))
    return out


def gather_results(results, max_len: int, rank: int, world: int, local_rank: int, counts: Sequence[int] = None):
    """Gather per-chunk result records to rank 0 (rank order).  Returns the decoded list on rank 0,
    None elsewhere.  counts: chunks per rank (defaults to len(results) everywhere)."""
    import torch
    dist = _dist()
    on_gpu = dist.get_backend() == "nccl"
    dev = torch.device("cuda", local_rank) if on_gpu else torch.device("cpu")
    n_max = max(counts) if counts is not None else len(results)
    rec = np.zeros((n_max, max_len + RECORD_EXTRA), dtype=np.int32)
    if len(results):
        rec[:len(results)] = encode_records(results, max_len)
    t = torch.from_numpy(rec).to(dev)
    bufs = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, bufs, dst=0)
    if rank != 0:
        return None
    out = []
    for r in range(world):
        n = counts[r] if counts is not None else len(results)
        out.extend(decode_records(bufs[r].cpu().numpy()[:n], max_len))
    return out
