"""Data-parallel glue: one process per GPU, graphs sharded by contiguous ``ptr`` ranges,
RCCL (``torch.distributed`` backend "nccl" on ROCm) for the two exchange steps the
reference has:

1. the training-loss metric, ``self.log(..., sync_dist=True)`` at
   ``/root/reference/gnnepcsaft/train/models.py:195-201``: here one all-reduce(sum) of
   ``[sum(ape), count]`` -> exact global MAPE (identical to the reference's mean of per-rank
   means when shards are equal-sized);
2. the DDP gradient all-reduce Lightning inserts (``train.py:142-145``): here ONE flat f32
   buffer per step (2-28 MB for the reference's model sizes -- latency-bound on xGMI, so a
   single collective instead of 25 MB DDP buckets).

BatchNorm statistics stay per rank, as in the reference (no SyncBatchNorm anywhere).
There is no data-path collective: forward kernels never communicate.
"""

from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist

from .data.synthetic import GraphData, split_graphs


def exchange_active() -> bool:
    """True when the exchange steps must issue their collectives: more than one rank -- or ONE rank with
    ``GNNSAFT_FORCE_COLLECTIVES=1`` (read at every call), which sends every collective of the data-parallel path
    through the backend anyway.  That is how a one-GPU box runs communicator creation, the asynchronous loss
    all-reduce, the flat gradient all-reduce and the segment-wise exchange on real RCCL
    (tests/test_gpu_nccl_single_rank.py); an all-reduce over one rank leaves the buffer as it is."""
    if not dist.is_initialized():
        return False
    return dist.get_world_size() > 1 or os.environ.get("GNNSAFT_FORCE_COLLECTIVES") == "1"


def init_from_env(backend: Optional[str] = None) -> tuple:
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run contract)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    forced = os.environ.get("GNNSAFT_FORCE_COLLECTIVES") == "1"
    if (world > 1 or forced) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard(data: GraphData, rank: int, world: int) -> GraphData:
    """This rank's contiguous range of graphs (what DDP + DistributedSampler give the reference)."""
    return split_graphs(data, world, rank)


def global_mape(loss3: torch.Tensor) -> torch.Tensor:
    """``loss3 = [local mape, local sum(ape), local count]`` -> global MAPE over all ranks."""
    if not exchange_active():
        return loss3[0]  # one rank: the kernel's own mean is the answer, no extra launches
    parts = loss3[1:3].clone()
    dist.all_reduce(parts, op=dist.ReduceOp.SUM)
    return parts[0] / parts[1]


class PendingMape:
    """Handle of an in-flight ``[sum(ape), count]`` all-reduce; ``result()`` waits for it."""

    def __init__(self, parts: torch.Tensor, work):
        self.parts = parts
        self.work = work

    def result(self) -> torch.Tensor:
        if self.work is not None:
            self.work.wait()
            self.work = None
        return self.parts[0] / self.parts[1]


def global_mape_async(loss3: torch.Tensor) -> PendingMape:
    """As ``global_mape`` but returns at once: the 8-byte collective is latency-bound, and the loss is only
    a logged metric (``sync_dist=True``), so the next step's kernels need not queue behind it."""
    parts = loss3[1:3].clone()
    work = None
    if exchange_active():
        work = dist.all_reduce(parts, op=dist.ReduceOp.SUM, async_op=True)
    return PendingMape(parts, work)


def allreduce_flat_sum(flat: torch.Tensor) -> float:
    """SUM all-reduce of one flat gradient buffer, in place; returns the factor that turns the sum into the
    mean (1 / world) so that the caller can fold it into its next kernel (the fused optimizers' ``grad_scale``)
    instead of spending a launch on the division."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if exchange_active():
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return 1.0 / world


def common_gradient_buffer(params: Iterable[torch.nn.Parameter]) -> Optional[torch.Tensor]:
    """The one flat tensor every ``.grad`` is a view of (what ``gnnsaft_backward`` leaves behind), or None."""
    base = None
    for p in params:
        g = p.grad
        b = None if g is None else getattr(g, "_base", None)
        if b is None or b.dim() != 1 or not g.is_contiguous():
            return None
        if base is None:
            base = b
        elif b.data_ptr() != base.data_ptr():
            return None
    return base


class OverlappedGradientExchange:
    """The gradient exchange of SURVEY.md section 5, overlapped with the backward: ``gnnsaft_backward`` completes the
    flat gradient buffer segment by segment (readout, layer L-1 .. 0, embeddings) and records an event per segment;
    every segment is SUM-all-reduced on a separate stream as soon as its event has fired, under the kernels that
    are still computing the earlier layers' gradients.  ``launch`` is called right after ``loss.backward()`` (the
    backward is enqueued, not finished), ``wait`` before the optimizer step.  A handful of collectives of 0.3-6 MB
    instead of one of 2-28 MB: latency-bound either way on xGMI, but hidden behind ~1.5 ms of backward."""

    def __init__(self, model, device=None):
        self.model = model
        self.segments = model.gradient_segments()
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.stream = torch.cuda.Stream(dev)
        self.events = [torch.cuda.Event() for _ in self.segments]
        for ev in self.events:
            ev.record(torch.cuda.current_stream(dev))     # creates the handles gnnsaft_backward records on
        model.gradient_segment_events = self.events
        self._works: List = []

    def close(self) -> None:
        self.model.gradient_segment_events = None

    def launch(self, flat: torch.Tensor) -> float:
        """All-reduce every segment of ``flat`` behind its completion event; returns 1 / world (the factor the fused
        optimizer folds into its kernel)."""
        world = dist.get_world_size() if dist.is_initialized() else 1
        if exchange_active():
            with torch.cuda.stream(self.stream):
                for (a, b), ev in zip(self.segments, self.events):
                    self.stream.wait_event(ev)
                    if b > a:
                        self._works.append(dist.all_reduce(flat[a:b], op=dist.ReduceOp.SUM, async_op=True))
        return 1.0 / world

    def wait(self) -> None:
        for w in self._works:
            w.wait()
        self._works.clear()
        torch.cuda.current_stream(self.stream.device).wait_stream(self.stream)


class FlatGradientAllReduce:
    """Averages gradients across ranks with a single collective over one flat buffer."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(total, dtype=ref.dtype, device=ref.device)

    def __call__(self) -> None:
        world = dist.get_world_size() if dist.is_initialized() else 1
        shared = common_gradient_buffer(self.params)
        if shared is not None:     # gradients already live in one buffer: reduce it where it is
            if exchange_active():
                dist.all_reduce(shared, op=dist.ReduceOp.SUM)
                if world > 1:
                    shared.div_(world)
            return
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is not None:
                self.flat[off:off + n].copy_(p.grad.reshape(-1))
            else:
                self.flat[off:off + n].zero_()
            off += n
        if exchange_active():
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            if world > 1:
                self.flat.div_(world)
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                p.grad = torch.empty_like(p)
            p.grad.copy_(self.flat[off:off + n].view_as(p))
            off += n
