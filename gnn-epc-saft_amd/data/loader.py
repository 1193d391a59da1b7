"""PyG-free batch loader for the training loop: what ``torch_geometric.loader.
DataLoader(train_dataset, batch_size, shuffle=True)`` gives the reference (``/root/reference/gnnepcsaft/train/
train.py:74-79``) for the fields the hot path reads -- ``x, edge_index, edge_attr, batch, ptr, para,
num_graphs`` -- delivered device-resident.

The dataset (a list of small per-molecule graphs) is flattened ONCE into packed int64 / float32 arrays with
per-graph offsets; a batch is then three gathers of contiguous row ranges (no Python loop over tensors, no
per-graph ``torch.cat``), staged through pinned host buffers and copied on a side HIP stream so that the copy of
batch k+1 overlaps the training step of batch k.  DataLoader worker processes (``train.py:77``) are not needed:
collation is a few vectorised index operations per batch.
"""

from __future__ import annotations

import queue
import threading
from typing import Iterator, List, Optional, Sequence

import numpy as np
import torch

from .synthetic import GraphData


class PackedGraphs:
    """All graphs of a dataset in five flat tensors + offsets (the on-host analogue of PyG's ``(data, slices)``
    ``InMemoryDataset`` storage, ``graphdataset.py:48``)."""

    def __init__(self, graphs: Sequence):
        if len(graphs) == 0:
            raise ValueError("empty dataset")
        self.num_graphs = len(graphs)
        n = torch.tensor([int(g.x.shape[0]) for g in graphs], dtype=torch.int64)
        e = torch.tensor([int(g.edge_index.shape[1]) for g in graphs], dtype=torch.int64)
        self.node_ptr = torch.cat([torch.zeros(1, dtype=torch.int64), n.cumsum(0)])
        self.edge_ptr = torch.cat([torch.zeros(1, dtype=torch.int64), e.cumsum(0)])
        self.x = torch.cat([g.x.to(torch.int64) for g in graphs])
        self.edge_index = torch.cat([g.edge_index.to(torch.int64) for g in graphs], dim=1)     # graph-local node ids
        self.edge_attr = torch.cat([g.edge_attr.to(torch.int64) for g in graphs])
        paras = [g.para for g in graphs]
        if any(p is None for p in paras):
            self.para, self.para_width = None, 0
        else:
            self.para_width = int(paras[0].numel())
            self.para = torch.stack([p.reshape(-1).to(torch.float32) for p in paras])

    @staticmethod
    def _ranges(ptr: np.ndarray, ids: np.ndarray):
        """Indices of the concatenated ranges [ptr[i], ptr[i+1]) for i in ids, and the start of each in the output."""
        lens = ptr[ids + 1] - ptr[ids]
        out_ptr = np.zeros(ids.size + 1, dtype=np.int64)
        np.cumsum(lens, out=out_ptr[1:])
        owner = np.repeat(np.arange(ids.size, dtype=np.int64), lens)                   # batch-local graph of each row
        idx = np.arange(int(out_ptr[-1]), dtype=np.int64) - out_ptr[owner] + ptr[ids][owner]
        return idx, owner, out_ptr

    def collate(self, ids: torch.Tensor) -> GraphData:
        """Batch of the graphs ``ids`` (in that order), identical to ``synthetic.collate([graphs[i] for i in ids])``.
        The index arithmetic runs in numpy on views of the packed tensors: a handful of vectorised single-thread ops
        (~0.2 ms for 512 molecules).  The same few torch CPU ops fan out over every host core for 1e4-element
        arrays: measured 23 ms per batch on the 256-core GPU host, ten training steps' worth."""
        ids_np = np.asarray(ids, dtype=np.int64).reshape(-1)
        nidx, nowner, nptr = self._ranges(self.node_ptr.numpy(), ids_np)
        eidx, eowner, _ = self._ranges(self.edge_ptr.numpy(), ids_np)
        edge_index = self.edge_index.numpy()[:, eidx] + nptr[eowner]       # re-base graph-local node ids
        para = None if self.para is None else torch.from_numpy(self.para.numpy()[ids_np].reshape(-1))
        return GraphData(torch.from_numpy(self.x.numpy()[nidx]), torch.from_numpy(edge_index),
                         torch.from_numpy(self.edge_attr.numpy()[eidx]), torch.from_numpy(nowner),
                         torch.from_numpy(nptr), para, int(ids_np.size))


class GraphLoader:
    """Iterable of device-resident batches.  ``for batch in loader`` yields ``GraphData`` on ``device``; every epoch
    reshuffles with ``seed + epoch`` (``shuffle=True``) like PyG's loader does through its sampler."""

    def __init__(self, graphs, batch_size: int, shuffle: bool = False, drop_last: bool = False,
                 device: Optional[torch.device] = None, seed: int = 0, rank: int = 0, world_size: int = 1,
                 cache_on_device: bool = False, structure_for=None, host_threads: Optional[int] = 8,
                 prefetch: int = 2):
        self.packed = graphs if isinstance(graphs, PackedGraphs) else PackedGraphs(graphs)
        # Collation and staging are a handful of tiny CPU ops per batch.  torch defaults to one intra-op thread per
        # host core (128 on the MI355X hosts) while a one-GPU job owns a 16-CPU share: the idle OpenMP workers spin on
        # the same CPUs as the thread that feeds the GPU -- measured 15 ms per 512-graph batch against 0.6-1.0 ms with
        # <= 8 threads, and a 2.1 ms training step stretched to 27 ms.  `host_threads=None` leaves torch's setting alone.
        if host_threads is not None and torch.get_num_threads() > host_threads:
            torch.set_num_threads(int(host_threads))
        if batch_size < 1:
            raise ValueError("batch_size must be positive")
        self.batch_size, self.shuffle, self.drop_last = int(batch_size), bool(shuffle), bool(drop_last)
        self.device = None if device is None else torch.device(device)
        self.seed, self.epoch = int(seed), 0
        self.rank, self.world_size = int(rank), int(world_size)     # DistributedSampler semantics: strided shards
        self._copy_stream = None
        # cache_on_device (only without shuffling: the batch list is then the same every epoch): keep the collated
        # batches resident in HBM; with structure_for=<PNAPCSAFT> also their CSR / degree tiles
        # (model.build_structure), so that from the second epoch on a step starts at the embeddings.
        if cache_on_device and shuffle:
            raise ValueError("cache_on_device needs a fixed batch list (shuffle=False)")
        self.cache_on_device, self.structure_for = bool(cache_on_device), structure_for
        self._cache: Optional[List[GraphData]] = None
        # pinned staging: three reusable slots (batch k+1 is staged while batch k computes; a slot is overwritten only
        # after its own copy has completed).  `tensor.pin_memory()` per batch would allocate page-locked memory for
        # six tensors every step: measured 25 ms per 512-graph batch against a 2.5 ms training step.
        # `prefetch` batches are collated, staged and put on the copy stream by a background thread while the caller
        # trains (collation is ~0.8 ms of host work per 512-graph batch, staging ~0.2 ms, and a training step at these
        # sizes is bound by the host: 1.6 ms).  numpy's gathers and the HIP calls release the GIL, the training step
        # spends its time inside one C call.  0 = collate in the caller's thread, one batch ahead.
        self.prefetch = max(0, int(prefetch))
        self._slots = [dict(buffers={}, ready=None) for _ in range(self.prefetch + 2 if self.prefetch else 3)]
        self._slot_i = 0

    def __len__(self) -> int:
        per_rank = (self.packed.num_graphs + self.world_size - 1) // self.world_size
        return per_rank // self.batch_size if self.drop_last else (per_rank + self.batch_size - 1) // self.batch_size

    def _order(self) -> torch.Tensor:
        n = self.packed.num_graphs
        if self.shuffle:
            order = torch.randperm(n, generator=torch.Generator().manual_seed(self.seed + self.epoch))
        else:
            order = torch.arange(n)
        if self.world_size > 1:      # pad to a multiple of the world size by wrapping around, then stride
            pad = (-n) % self.world_size
            if pad:
                order = torch.cat([order, order[:pad]])
            order = order[self.rank::self.world_size]
        return order

    def _batches(self) -> List[torch.Tensor]:
        order = self._order()
        chunks = list(order.split(self.batch_size))
        if self.drop_last and chunks and chunks[-1].numel() < self.batch_size:
            chunks.pop()
        return chunks

    def _stage(self, host: GraphData):
        """Pinned staging + asynchronous copy on the side stream; returns (device batch, ready event)."""
        dev = self.device
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(dev)
        slot = self._slots[self._slot_i % len(self._slots)]
        self._slot_i += 1
        if slot["ready"] is not None:
            slot["ready"].synchronize()          # the copy that last used these buffers (three batches ago) is done

        def pin(name, t):
            if t is None:
                return None
            buf = slot["buffers"].get(name)
            if buf is None or buf.dtype != t.dtype or buf.numel() < t.numel():
                buf = torch.empty(max(int(t.numel() * 1.25), 1), dtype=t.dtype).pin_memory()
                slot["buffers"][name] = buf
            view = buf[:t.numel()].view(t.shape)
            view.copy_(t)
            return view

        pinned = GraphData(pin("x", host.x), pin("edge_index", host.edge_index), pin("edge_attr", host.edge_attr),
                           pin("batch", host.batch), pin("ptr", host.ptr), pin("para", host.para), host.num_graphs)
        with torch.cuda.stream(self._copy_stream):
            out = pinned.to(dev, non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(self._copy_stream)
        slot["ready"] = ready
        return out, ready, pinned

    def __iter__(self) -> Iterator[GraphData]:
        if self._cache is not None:
            self.epoch += 1
            yield from self._cache
            return
        if self.cache_on_device:
            if self.device is None or self.device.type != "cuda":
                raise ValueError("cache_on_device needs a HIP device")
            built = []
            for ids in self._batches():
                b = self.packed.collate(ids).to(self.device)
                if self.structure_for is not None:
                    b.gnnsaft_structure = self.structure_for.build_structure(b)
                built.append(b)
            self._cache = built
            self.epoch += 1
            yield from built
            return
        chunks = self._batches()
        self.epoch += 1
        if self.device is None or self.device.type != "cuda":
            for ids in chunks:
                b = self.packed.collate(ids)
                yield b if self.device is None else b.to(self.device)
            return
        yield from self._device_batches(iter(chunks))

    def forever(self) -> Iterator[GraphData]:
        """Endless stream of device batches, epoch after epoch (reshuffled like successive ``iter(loader)`` calls), with
        the prefetch running ACROSS epoch boundaries: a 2 000-graph set is four batches per epoch, and restarting the
        producer per epoch would leave every fourth step waiting for its batch."""
        if self.device is None or self.device.type != "cuda" or self.cache_on_device:
            while True:
                yield from self
            return

        def chunks():
            while True:
                cs = self._batches()
                self.epoch += 1
                yield from cs

        yield from self._device_batches(chunks())

    def _device_batches(self, chunks) -> Iterator[GraphData]:
        if self.prefetch == 0:
            nxt = next(chunks, None)
            nxt = self._stage(self.packed.collate(nxt)) if nxt is not None else None
            while nxt is not None:
                cur = nxt
                ids = next(chunks, None)
                nxt = self._stage(self.packed.collate(ids)) if ids is not None else None   # overlaps the step on `cur`
                yield self._hand_over(cur)
            return
        # background producer: at most `prefetch` staged batches wait in the queue, one more is being built, one is with
        # the consumer -- the slot ring (prefetch + 2) never hands out buffers whose copy is still running
        q: "queue.Queue" = queue.Queue(maxsize=self.prefetch)
        stop = threading.Event()
        dev = self.device

        def produce():
            try:
                torch.cuda.set_device(dev)
                for ids in chunks:
                    if stop.is_set():
                        return
                    item = self._stage(self.packed.collate(ids))
                    while not stop.is_set():
                        try:
                            q.put(item, timeout=0.05)
                            break
                        except queue.Full:
                            continue
                q.put(None)
            except BaseException as exc:  # noqa: BLE001  (handed to the consumer)
                q.put(exc)

        worker = threading.Thread(target=produce, name="gnnsaft-loader", daemon=True)
        worker.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                yield self._hand_over(item)
        finally:
            stop.set()
            while worker.is_alive():       # unblock a producer waiting on a full queue, then let it finish
                try:
                    q.get_nowait()
                except queue.Empty:
                    pass
                worker.join(timeout=0.05)

    def _hand_over(self, staged) -> GraphData:
        """The consumer's stream waits for the batch's copy; the allocator learns who uses the tensors."""
        batch, ready, _keep = staged
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ready)
        for t in (batch.x, batch.edge_index, batch.edge_attr, batch.batch, batch.ptr, batch.para):
            if t is not None:
                t.record_stream(cur)
        return batch
