"""One-frame-per-GPU data parallelism for the rasterizer hot path.

GGRt's intended (but dead, reference ``train_ggrt_stable.py:322-328``) multi-GPU mode is plain data
parallelism: ``batch_size = 1`` per process, "use distributed parallel on multiple GPUs to train
multiple target views per batch" (reference ``ggrt/base/trainer.py:115-117``).  Frames are independent, so
nothing of the rasterizer's data path is exchanged; the single collective per iteration is the mean
all-reduce of the parameter gradients of the encoder + pose network.  On MI355X that is one RCCL
all-reduce over xGMI on a flat fp32 buffer (backend "nccl" IS RCCL on ROCm); on CPU tests it is gloo.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Sequence

import torch
import torch.distributed as dist


def init_from_env(expected_world: int | None = None, backend: str | None = None, force: bool = False):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run contract) and, for
    world > 1 (or when `force`d: a one-rank group, so that the RCCL path itself can be executed on a 1-GPU box),
    creates the process group (RCCL on GPU, gloo on CPU).  Returns (rank, world, local_rank)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if expected_world is not None and expected_world != world:
        if world == 1 and expected_world > 1:
            raise RuntimeError(f"--gpus {expected_world} needs a torch.distributed.run launch (WORLD_SIZE={world})")
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            n_dev = torch.cuda.device_count()
            if n_dev == 1 and local >= 1:
                local = 0   # per-rank device visibility (HIP_VISIBLE_DEVICES set per rank): the one visible device is index 0
            elif n_dev and local >= n_dev:
                # folding ranks onto shared GPUs silently would turn a launch mistake into a wrong scaling figure
                # (the reference fails the same way: `cuda:{local_rank}`, ggrt/base/trainer.py)
                raise RuntimeError(f"LOCAL_RANK {local} but only {n_dev} GPUs are visible to this rank: launch one rank "
                                   "per visible GPU (or give every rank exactly one device)")
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device(f"cuda:{local}")
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def world_size() -> int:
    return dist.get_world_size() if dist.is_initialized() else 1


def barrier():
    if dist.is_initialized():
        dist.barrier()


def shutdown():
    if dist.is_initialized():
        dist.destroy_process_group()


def shard_frames(num_frames: int, rank: int, world: int) -> List[int]:
    """Frame indices owned by `rank`: frame f goes to rank f % world (an 8-frame iteration on 8 GPUs
    gives one frame each — BASELINE.json config 5)."""
    return [f for f in range(num_frames) if f % world == rank]


def allreduce_mean_(flat: torch.Tensor) -> torch.Tensor:
    """In-place mean all-reduce of a flat gradient buffer (one large message: xGMI rings are per-link
    bound, so one 260 MB collective beats many small ones)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(dist.get_world_size())
    return flat


class ChunkedMeanAllReduce:
    """Mean all-reduce of a flat buffer issued as `chunks` asynchronous collectives of equal size (the last one takes
    the remainder) — the bucketed form a trainer overlaps with its backward: chunk k can be issued as soon as the
    gradients it covers exist, RCCL runs the chunks in order on its own stream, and the consumer waits chunk by
    chunk.  RCCL averages inside the collective (`ReduceOp.AVG`); gloo has no AVG: sum, then scale on wait."""

    def __init__(self, chunks: int = 1):
        self.chunks = max(1, int(chunks))
        self.pending: list = []

    def issue(self, flat: torch.Tensor) -> None:
        if not dist.is_initialized():
            return
        use_avg = dist.get_backend() == "nccl"
        n = flat.numel()
        if n == 0:   # (an empty buffer still takes part in the collective sequence: every rank issues the same calls)
            work = dist.all_reduce(flat, op=dist.ReduceOp.AVG if use_avg else dist.ReduceOp.SUM, async_op=True)
            self.pending.append((work, None))
            return
        step = (n + self.chunks - 1) // self.chunks
        for lo in range(0, n, step):
            part = flat[lo:lo + step]
            work = dist.all_reduce(part, op=dist.ReduceOp.AVG if use_avg else dist.ReduceOp.SUM, async_op=True)
            self.pending.append((work, None if use_avg else part))

    def wait(self) -> None:
        world = world_size()
        for work, part in self.pending:
            work.wait()
            if part is not None and world > 1:
                part.div_(world)
        self.pending = []


def allreduce_gradients(params: Iterable[torch.nn.Parameter]) -> None:
    """Flatten → one all-reduce → scatter back, for the modules that stay on stock PyTorch
    (encoder, pose network) and for the camera-pose gradients the rasterizer produces."""
    ps: Sequence[torch.nn.Parameter] = [p for p in params if p.requires_grad]
    if not ps or not dist.is_initialized() or dist.get_world_size() == 1:
        return
    # the SAME layout on every rank: a parameter that got no gradient on this rank (unused for this frame) counts
    # as zeros — filtering on `p.grad is not None` would give ranks different message sizes (hang / mixed grads)
    for p in ps:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    flat = torch.cat([p.grad.reshape(-1) for p in ps])
    allreduce_mean_(flat)
    o = 0
    for p in ps:
        n = p.grad.numel()
        p.grad.copy_(flat[o:o + n].view_as(p.grad))
        o += n


def device_identity(device) -> dict:
    """What tells physical GPUs apart: PCI domain:bus:device where torch exposes it, else the device UUID / name."""
    pr = torch.cuda.get_device_properties(device)
    ident = None
    if all(hasattr(pr, a) for a in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
        ident = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
    elif getattr(pr, "uuid", None) is not None:
        ident = str(pr.uuid)
    return {"index": torch.device(device).index, "pci": ident, "name": pr.name}


def rank_device_report(device, shared_ok: bool = False) -> list:
    """All ranks' device identities, gathered on every rank; raises if two ranks of the job sit on the same physical
    device (unless `shared_ok`: the 1-GPU dry run of the N > 1 control flow).  One process per GPU is what the frame
    sharding assumes (reference ggrt/base/trainer.py:115-117: `device = cuda:{local_rank}`)."""
    mine = dict(rank=dist.get_rank() if dist.is_initialized() else 0, **device_identity(device))
    if not dist.is_initialized():
        return [mine]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, mine)
    ids = [o["pci"] if o["pci"] is not None else f"index{o['index']}" for o in out]
    if not shared_ok and len(set(ids)) != len(ids):
        raise RuntimeError(f"two ranks share one physical GPU: {out}")
    return out


def max_over_ranks(value: float, device) -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
