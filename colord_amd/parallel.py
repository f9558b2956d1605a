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
    """Variable-length all-gather (rank order): the list of per-rank tensors, each at its own size — no padding to the largest
    shard (one broadcast per source rank into a buffer of exactly its size)."""
    w = world()
    if w == 1:
        return [t]
    dev = t.device
    if _host_staged():
        t = t.cpu()
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(w)]
    dist.all_gather(sizes, n)
    out = []
    for r, s_ in enumerate(int(x.item()) for x in sizes):
        buf = t.contiguous() if r == rank() else torch.empty(s_, dtype=t.dtype, device=t.device)
        if s_:
            dist.broadcast(buf, src=r)
        out.append(buf.to(dev))
    return out


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


# ---------------------------------------------------------------------------------------------------------------------
# cl_exchange: the collectives a multi-GPU cl_compressor (csrc/stream.hip) calls back into.  The library says WHAT is
# exchanged (k-mers by key-range owner, kept keys, reference reads, index entries); this class moves the bytes with
# torch.distributed — backend "nccl" is RCCL over xGMI, device to device; "gloo" stages through host memory and exists for
# functional tests with fewer GPUs than ranks (and for the CPU suite, where the "device" pointers are host pointers).
# ---------------------------------------------------------------------------------------------------------------------
import ctypes as _C
import numpy as _np
from . import _native as _N


class _DevPtr:
    """A raw device pointer as a __cuda_array_interface__ object (zero-copy view for torch.as_tensor)."""
    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class TorchExchange:
    def __init__(self, device=None):
        self.rank, self.world = rank(), world()
        self.device = device
        self.host_mem = device is None or torch.device(device).type == "cpu"     # CPU suite: pointers are host pointers
        self.err = None
        self.bytes_moved = 0                                                       # payload bytes this rank received (diagnostic)
        # one record per callback: (phase, collective, seconds, bytes received from the other ranks).  `phase` is set by the driver before
        # the library call that makes the exchange ("kmers": cl_compressor_count_finish, "refs": cl_compressor_refs_finish), so that a
        # multi-GPU run can say where its time between the GPUs went
        self.phase = ""
        self.log = []
        self._cbs = (_N._EXCH_GATHER_HOST(self._gather_host), _N._EXCH_A2AV(self._all_to_all_v), _N._EXCH_GATHERV(self._all_gather_v))
        self.c_struct = _N.Exchange(None, self.rank, self.world, *self._cbs)

    # -- pointer -> tensor views --------------------------------------------------------------------------------------
    def _view(self, ptr, nbytes):
        if not nbytes:
            return torch.empty(0, dtype=torch.uint8, device="cpu" if self.host_mem else self.device)
        if self.host_mem:
            return torch.from_numpy(_np.ctypeslib.as_array(_C.cast(ptr, _C.POINTER(_C.c_uint8)), (int(nbytes),)))
        return torch.as_tensor(_DevPtr(int(ptr), int(nbytes)), device=self.device)

    @staticmethod
    def _widen(t, *counts):
        """View byte tensors / byte counts in the widest element type that divides them all (fewer, larger elements)."""
        for dt, w in ((torch.int64, 8), (torch.int32, 4)):
            if all(c % w == 0 for cs in counts for c in cs) and all(x.numel() % w == 0 and x.data_ptr() % w == 0 for x in t):
                return [x.view(dt) for x in t], [[c // w for c in cs] for cs in counts]
        return list(t), [list(cs) for cs in counts]

    def _guard(self, fn, op=""):
        import time as _t
        t0, b0 = _t.perf_counter(), self.bytes_moved
        try:
            fn()
            self.log.append((self.phase, op, _t.perf_counter() - t0, self.bytes_moved - b0))
            return 0
        except Exception as e:                      # surfaces as CL_E_HIP from the library call; the caller re-raises self.err
            self.err = e
            return _N.CL_E_HIP

    # -- callbacks ----------------------------------------------------------------------------------------------------
    def _gather_host(self, user, vals, n, out):
        def run():
            mine = torch.from_numpy(_np.ctypeslib.as_array(vals, (int(n),)).view(_np.int64).copy())
            dev = "cpu" if (self.host_mem or _host_staged()) else self.device
            parts = [torch.empty(int(n), dtype=torch.int64, device=dev) for _ in range(self.world)]
            dist.all_gather(parts, mine.to(dev))
            res = torch.cat(parts).cpu().numpy().view(_np.uint64)
            _np.ctypeslib.as_array(out, (int(n) * self.world,))[:] = res
        return self._guard(run, "all_gather_host")

    def _all_to_all_v(self, user, d_send, h_send, d_recv, h_recv):
        def run():
            sb = [int(h_send[i]) for i in range(self.world)]
            rb = [int(h_recv[i]) for i in range(self.world)]
            send, recv = self._view(d_send, sum(sb)), self._view(d_recv, sum(rb))
            (send, recv), (sc, rc) = self._widen([send, recv], sb, rb)
            if self.host_mem or not _host_staged():
                dist.all_to_all_single(recv, send, rc, sc)
            else:
                tmp = torch.empty(recv.shape, dtype=recv.dtype)
                dist.all_to_all_single(tmp, send.cpu(), rc, sc)
                recv.copy_(tmp)
            if not self.host_mem:
                torch.cuda.synchronize()
            self.bytes_moved += sum(rb) - rb[self.rank]
        return self._guard(run, "all_to_all_v")

    def _all_gather_v(self, user, d_send, send_bytes, d_recv, h_recv):
        def run():
            rb = [int(h_recv[i]) for i in range(self.world)]
            assert int(send_bytes) == rb[self.rank], "all_gather_v: send size differs from the announced size"
            recv = self._view(d_recv, sum(rb))
            offs = [sum(rb[:r]) for r in range(self.world)]
            staged = (not self.host_mem) and _host_staged()
            if rb[self.rank]:
                recv[offs[self.rank]:offs[self.rank] + rb[self.rank]].copy_(self._view(d_send, rb[self.rank]))
            if not self.host_mem and not staged and dist.get_backend() == "nccl":
                # RCCL: ONE grouped call, every rank's share straight into its slice of the receive buffer (uneven sizes: the process
                # group issues the per-source broadcasts inside one ncclGroup), then one synchronisation for the library's own streams
                dist.all_gather([recv[offs[r]:offs[r] + rb[r]] for r in range(self.world)], recv[offs[self.rank]:offs[self.rank] + rb[self.rank]])
                torch.cuda.synchronize()
            else:
                for r in range(self.world):                      # gloo (functional tests, CPU suite): one broadcast per source rank
                    if not rb[r]:
                        continue
                    (v,), _ = self._widen([recv[offs[r]:offs[r] + rb[r]]])
                    if staged:
                        h = v.cpu()
                        dist.broadcast(h, src=r)
                        if r != self.rank:
                            v.copy_(h)
                    else:
                        dist.broadcast(v, src=r)
                if not self.host_mem:
                    torch.cuda.synchronize()
            self.bytes_moved += sum(rb) - rb[self.rank]
        return self._guard(run, "all_gather_v")

    def summary(self):
        """Seconds, bytes received and calls per (phase, collective) since the log was last cleared."""
        out = {}
        for ph, op, s_, b in self.log:
            e = out.setdefault(f"{ph or 'other'}.{op}", {"seconds": 0.0, "bytes_received": 0, "calls": 0})
            e["seconds"] += s_; e["bytes_received"] += int(b); e["calls"] += 1
        for e in out.values():
            e["seconds"] = round(e["seconds"], 4)
        return out


def gather_to_root(t: torch.Tensor, root: int = 0):
    """Variable-length gather of byte tensors to `root` (compressed parts go to the rank that writes the archive).
    Point-to-point sends (nccl: RCCL send/recv over xGMI); returns the list of per-rank tensors on root, None elsewhere."""
    w, r = world(), rank()
    if w == 1:
        return [t]
    staged = _host_staged()
    n = torch.tensor([t.numel()], dtype=torch.int64, device="cpu" if staged else t.device)
    sizes = [torch.zeros_like(n) for _ in range(w)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    if r == root:
        out = []
        for src in range(w):
            if src == root:
                out.append(t)
                continue
            buf = torch.empty(sizes[src], dtype=t.dtype, device="cpu" if staged else t.device)
            if sizes[src]:
                dist.recv(buf, src=src)
            out.append(buf.to(t.device))
        return out
    if t.numel():
        dist.send(t.cpu() if staged else t, dst=root)
    return None
