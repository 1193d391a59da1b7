"""Training-step host loop around the hot path: what Lightning's ``Trainer.fit``
does with ``PNApcsaftL`` in ``/root/reference/gnnepcsaft/train/train.py:142-185`` -- ``max_steps`` optimizer steps,
scheduler stepped per batch (``models.py:181-188``), ``train_mape`` logged every ``log_every_steps`` with
``sync_dist=True`` (mean over ranks, ``models.py:195-201``), a checkpoint every ``checkpoint_every_steps``
(``train.py:86-107``) -- without Lightning or Ray (both absent from the image; their callbacks, loggers and the
``feos``-based validation are out of scope).

One step = ``gnnsaft_forward`` (tape) -> ``gnnsaft_mape`` / ``gnnsaft_mape_backward`` -> ``gnnsaft_backward``
(every gradient into ONE flat buffer) -> [data-parallel: one RCCL all-reduce(SUM) of that buffer; the 1/world
average is folded into the optimizer kernel] -> ``gnnsaft_adamw_step`` / ``gnnsaft_sgd_step`` on the flat parameter
buffer.  No host synchronisation inside a step; the loss is read back only when it is logged.
"""

from __future__ import annotations

import os
from typing import Callable, Iterable, List, Optional, Tuple

import torch
import torch.distributed as dist

from ..parallel import OverlappedGradientExchange, allreduce_flat_sum, common_gradient_buffer, exchange_active
from .checkpoint import lightning_checkpoint, resume, save_checkpoint
from .models import PNApcsaftL, _cfg


def _world() -> int:
    return dist.get_world_size() if dist.is_initialized() else 1


def allreduce_gradients(optimizer) -> None:
    """Data-parallel exchange of one step: SUM all-reduce of the flat gradient buffer, averaging deferred to the
    optimizer kernel (``grad_scale``).  One collective of 2-28 MB for the reference's model sizes."""
    flat = optimizer._flat_grad()      # backward's own buffer (zero copy) or a gathered copy of the p.grad tensors
    optimizer.use_reduced_gradient(flat, allreduce_flat_sum(flat))   # step() consumes THIS buffer, not p.grad again


def broadcast_training_state(lit: PNApcsaftL, optimizer, scheduler, step: int, src: int = 0) -> int:
    """What DDP does when Lightning wraps the module (train.py:142-145): every replica starts from rank ``src``'s
    parameters and buffers -- plus, here, its optimizer / scheduler state and step counter, so that a resume on
    one rank or a per-rank seed cannot leave silently diverging replicas.  One flat parameter buffer, one flat
    buffer per optimizer state, one packed buffer for the BatchNorm statistics."""
    if not exchange_active():   # (one rank: nothing to agree on -- unless the collectives are forced through the backend)
        return step
    dist.broadcast(optimizer.flat_parameters(), src)          # every trainable parameter, one collective
    owned = {id(p) for p in optimizer._params}
    for p in lit.parameters():                                 # frozen parameters live outside the flat buffer
        if id(p) not in owned:
            dist.broadcast(p.data, src)
    bufs = [b for b in lit.buffers()]
    if bufs:
        packed = torch.cat([b.detach().reshape(-1).to(torch.float64) for b in bufs])
        dist.broadcast(packed, src)
        off = 0
        with torch.no_grad():
            for b in bufs:
                b.copy_(packed[off:off + b.numel()].view(b.shape).to(b.dtype))
                off += b.numel()
    meta = [None]
    if dist.get_rank() == src:
        meta = [dict(step=int(step), steps=int(optimizer._steps), keys=sorted(optimizer._flat_state),
                     sched=scheduler.state_dict(), lr=[g["lr"] for g in optimizer.param_groups])]
    dist.broadcast_object_list(meta, src)
    m = meta[0]
    for key in m["keys"]:
        dist.broadcast(optimizer._state_buffer(key), src)
    if dist.get_rank() != src:
        optimizer._steps = m["steps"]
        for p in optimizer._params:
            if m["steps"]:
                optimizer.state[p]["step"] = torch.tensor(float(m["steps"]))
        scheduler.load_state_dict(m["sched"])
        for g, lr in zip(optimizer.param_groups, m["lr"]):
            g["lr"] = lr
    return int(m["step"])


def training_loop(lit: PNApcsaftL, batches: Iterable, max_steps: Optional[int] = None, *,
                  log_every_steps: Optional[int] = None, checkpoint_every_steps: Optional[int] = None,
                  workdir: Optional[str] = None, resume_from=None,
                  on_log: Optional[Callable[[int, float, float], None]] = None,
                  overlap_gradient_exchange: bool = False) -> List[Tuple[int, float]]:
    """Runs ``max_steps`` (default ``config.num_train_steps``) training steps over ``batches`` (any re-iterable of
    device-resident PyG-like batches with ``para``; iterated again when exhausted, i.e. epochs) and returns the
    logged ``(step, train_mape)`` pairs.  Rank 0 writes Lightning-dialect checkpoints under
    ``workdir/train/checkpoints`` when asked to."""
    cfg = lit.config
    if max_steps is None:
        max_steps = int(_cfg(cfg, "num_train_steps"))
    if log_every_steps is None:
        log_every_steps = int(_get(cfg, "log_every_steps", 50))
    if checkpoint_every_steps is None:
        checkpoint_every_steps = int(_get(cfg, "checkpoint_every_steps", 0))
    conf = lit.configure_optimizers()
    opt, sched = conf["optimizer"], conf["lr_scheduler"]["scheduler"]
    step = 0
    if resume_from is not None:
        step = resume(resume_from, opt, sched)
    lit.train()
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = _world()
    step = broadcast_training_state(lit, opt, sched, step)
    # world > 1, opt-in: all-reduce every finished gradient segment under the rest of the backward (needs the
    # gradients in backward's own flat buffer, i.e. no frozen parameters; otherwise the single flat all-reduce)
    exchange = OverlappedGradientExchange(lit.model) if (overlap_gradient_exchange and exchange_active()) else None
    history: List[Tuple[int, float]] = []
    epoch = 0
    while step < max_steps:
        seen = 0
        for batch in batches:
            if step >= max_steps:
                break
            seen += 1
            opt.zero_grad(set_to_none=True)
            loss = lit.training_step(batch, seen - 1)
            loss.backward()
            flat = common_gradient_buffer(opt._params) if exchange is not None else None
            if flat is not None and flat.numel() >= opt._total:
                opt.use_reduced_gradient(flat, exchange.launch(flat))
                exchange.wait()
            else:
                allreduce_gradients(opt)
            opt.step()
            sched.step()
            step += 1
            if log_every_steps and step % log_every_steps == 0:
                logged = loss.detach().clone()
                if exchange_active():   # sync_dist=True: mean of the per-rank means
                    dist.all_reduce(logged, op=dist.ReduceOp.SUM)
                    if world > 1:
                        logged /= world
                value = float(logged)   # the only host sync of the loop
                flags = lit.model.input_error_flags()      # rides on that sync: clamped / dropped indices are not
                if flags & 16:                              # fatal: BatchNorm statistics of the readout incomplete
                    raise RuntimeError(f"GNNSAFT_FLAG_BARRIER_TIMEOUT (bit 16 of {flags:#x}) by step {step}: a grid "
                                       "barrier of the fused readout (forward or backward) gave up waiting -- its "
                                       "workgroups were not co-resident (GPU shared with another process, CU mask or "
                                       "partition mode).  The affected outputs / gradients were poisoned with NaN; "
                                       "set model.fused_readout = False on such a device")
                if flags:                                   # silent (the reference's Embedding / scatter would fault)
                    raise ValueError(f"training batches raised GNNSAFT_FLAG_* bits {flags:#x} by step {step} (1 bad "
                                     "edge index, 2 categorical index outside its vocabulary, 4 batch vector not "
                                     "sorted / out of range, 8 in-degree >= 32 with the degree-folded update)")
                history.append((step, value))
                if on_log is not None and rank == 0:
                    on_log(step, value, float(opt.param_groups[0]["lr"]))
            if checkpoint_every_steps and workdir and step % checkpoint_every_steps == 0 and rank == 0:
                save_checkpoint(lightning_checkpoint(lit, opt, sched, step, epoch),
                                os.path.join(workdir, "train", "checkpoints", f"step={step}.ckpt"))
        if seen == 0:
            raise ValueError("training_loop got an empty batch iterable")
        epoch += 1
    if exchange is not None:
        exchange.close()
    return history


class GraphedTrainingStep:
    """One training step on a FIXED batch -- ``training_step`` (gnnsaft_forward with tape + MAPE), ``backward``
    (gnnsaft_mape_backward + gnnsaft_backward on two streams), fused AdamW step -- captured once in a hipGraph and
    replayed: ~140 launches and ~1.5 ms of host work per step become one graph launch.  Every replay trains on the
    SAME batch object (same tensors; their contents may be overwritten in place between replays), so this serves
    fixed-shape training and measurement, not a loader of ragged batches.

    The optimizer must be a ``FusedAdamW`` (it is switched to ``capturable``: learning rate and bias corrections are
    read from device memory that ``prepare_replay()`` rewrites with a one-thread launch in front of every replay --
    kernel arguments, so replays may queue up behind a busy GPU without reading a later step's values);
    ``scheduler.step()`` runs
    after every replay, on the host, as Lightning's ``interval="step"`` does.  ``warmup`` eager steps run first, on a
    side stream, as torch's whole-network capture recipe requires -- they are real training steps."""

    def __init__(self, lit: PNApcsaftL, optimizer, batch, scheduler=None, warmup: int = 3, choose: str = "graph",
                 trial_steps: int = 6):
        """``choose="auto"``: after the capture, ``trial_steps`` replays and ``trial_steps`` eager steps are timed (all of
        them real training steps) and ``__call__`` uses the faster form from then on (``self.mode``, ``self.trial_ms``):
        the eager step overlaps its two backward streams, a replayed graph realises only part of that overlap, so on
        large batches the eager step can be the faster one."""
        from .optim import FusedAdamW
        if not isinstance(optimizer, FusedAdamW):
            raise TypeError("GraphedTrainingStep needs this package's FusedAdamW (optimizer='adam')")
        self.lit, self.opt, self.sched, self.batch = lit, optimizer, scheduler, batch
        optimizer.capturable = True
        dev = next(lit.parameters()).device
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(max(1, int(warmup))):
                self._eager()
        cur.wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)      # the captured backward then owns one static flat gradient buffer
        with torch.cuda.graph(self.graph):
            self.loss = lit.training_step(batch)
            self.loss.backward()
            optimizer.step()                       # records copy + kernel; counts nothing (prepare_replay does)
        self.mode, self.trial_ms = "graph", None
        if choose == "auto":
            import time

            def timed(fn):
                fn()                               # (untimed: the first eager step behind a capture re-allocates its tape)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(max(1, int(trial_steps))):
                    fn()
                torch.cuda.synchronize(dev)
                return (time.perf_counter() - t0) / max(1, int(trial_steps)) * 1e3

            g_ms, e_ms = timed(self._replay), timed(self._eager)
            self.trial_ms = {"graph": g_ms, "eager": e_ms}
            self.mode = "graph" if g_ms <= e_ms else "eager"
        elif choose != "graph":
            raise ValueError("choose must be 'graph' or 'auto'")

    def _eager(self):
        self.opt.zero_grad(set_to_none=True)
        loss = self.lit.training_step(self.batch)
        loss.backward()
        self.opt.step()
        if self.sched is not None:
            self.sched.step()
        return loss

    def _replay(self) -> torch.Tensor:
        self.opt.prepare_replay()
        self.lit.model.bump_dropout_step()     # (readout dropout: fresh masks per replay; no-op without dropout)
        self.graph.replay()
        if self.sched is not None:
            self.sched.step()
        return self.loss

    def __call__(self) -> torch.Tensor:
        """One optimizer step -- a replay of the captured graph (its static loss tensor), or, when ``choose="auto"``
        found it faster, the eager step."""
        return self._replay() if self.mode == "graph" else self._eager()


def _get(cfg, name, default):
    try:
        return _cfg(cfg, name)
    except (KeyError, AttributeError):
        return default
