"""Multi-GPU exchange steps of the hot path (one process per GPU, torch.distributed; "nccl" = RCCL over xGMI).

The path shards by reads; only two exchanges exist:
  1. k-mer counting — surviving k-mers go to the rank that owns their key (all-to-all-v), each rank counts
     its disjoint key partition, the kept keys are all-gathered so that every rank holds the replicated set;
  2. reference index — the (k-mer id, reference id) entries of each rank's accepted reads are all-gathered
     in rank order (= global read order) so that every rank can query the replicated k-mer->reads index.
Everything here is plumbing around tensors that the HIP kernels produce/consume; it works on CPU tensors
with the gloo backend too, which is how the CPU test suite covers it.
"""
from __future__ import annotations
import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_initialized() else 0


def _host_staged() -> bool:
    """gloo moves tensors through host memory: device tensors are staged explicitly (functional tests of the multi-rank
    path on fewer GPUs than ranks use gloo; production is nccl = RCCL, device to device)."""
    return dist.is_initialized() and dist.get_backend() == "gloo"


def owner_of(kmers: torch.Tensor, w: int) -> torch.Tensor:
    """Key partition (any fixed function of the key works): k-mer mod a prime, mod world."""
    return kmers.remainder(1000003).remainder(w)


def exchange_kmers(kmers: torch.Tensor) -> torch.Tensor:
    """all-to-all-v of surviving k-mers by owner rank."""
    w = world()
    if w == 1:
        return kmers
    dev = kmers.device
    if _host_staged():
        kmers = kmers.cpu()
    dest = owner_of(kmers, w)
    order = torch.argsort(dest, stable=True)
    send = kmers[order].contiguous()
    scnt = torch.bincount(dest, minlength=w)
    rcnt = torch.empty_like(scnt)
    dist.all_to_all_single(rcnt, scnt)
    recv = torch.empty(int(rcnt.sum().item()), dtype=kmers.dtype, device=kmers.device)
    dist.all_to_all_single(recv, send, rcnt.tolist(), scnt.tolist())
    return recv.to(dev)


def all_gather_v(t: torch.Tensor) -> list:
    """Variable-length all-gather (rank order).  Returns the list of per-rank tensors."""
    w = world()
    if w == 1:
        return [t]
    dev = t.device
    if _host_staged():
        t = t.cpu()
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(w)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    pad = torch.zeros(mx, dtype=t.dtype, device=t.device)
    pad[:t.numel()] = t
    bufs = [torch.empty_like(pad) for _ in range(w)]
    dist.all_gather(bufs, pad)
    return [b[:s].to(dev) for b, s in zip(bufs, sizes)]


def all_reduce_sum_ints(*vals):
    w = world()
    if w == 1:
        return list(vals)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor(list(vals), dtype=torch.int64, device=dev)
    dist.all_reduce(t)
    return [int(x) for x in t.tolist()]


def exclusive_prefix(val: int, device) -> tuple:
    """(sum over lower ranks, total) of one integer per rank."""
    w = world()
    if w == 1:
        return 0, val
    parts = all_gather_v(torch.tensor([val], dtype=torch.int64, device=device))
    vals = [int(p.item()) for p in parts]
    return sum(vals[:rank()]), sum(vals)
