"""MI355X-native ``PNAPCSAFT``: same constructor, ``forward(data)`` signature,
duck-typed PyG ``Data`` / ``Batch`` input and ``state_dict`` keys as
``/root/reference/gnnepcsaft/train/models.py:48-135``, but the arithmetic runs
in hand-written gfx950 kernels behind the C ABI of ``include/gnnsaft.h``.

The sub-modules below are parameter containers only (they reproduce the
attribute names PyG / ogb give their parameters -- SURVEY.md Appendix C -- so
reference checkpoints load with ``load_state_dict``); none of them computes
anything.  There is no CPU or PyTorch-eager fallback: tensors that are not on
a HIP device raise.  float32 modules run the batched pipeline (``gnnsaft_forward``,
train and eval, with backward) or -- eval mode, small batches -- the per-graph
fused kernel (``gnnsaft_graph_forward``); float64 modules (``.to(torch.float64)``,
what ``evaluate_ensemble.py:67-77`` and ``demo/utils.py:23-27`` do) run that
per-graph kernel in double precision, eval mode only.
"""

from __future__ import annotations

import ctypes
import dataclasses
import math
import os
from typing import List, Optional, Sequence, Tuple

import torch
from torch import nn

from .. import _native
from .._native import ModelDesc, aux_for, check, lib

# int32 words of the per-module flag buffer: word 0 = the sticky GNNSAFT_FLAG_* word, the rest = persistent barrier
# state of the cooperative structure chain (gnnsaft_model_desc.persistent_sync_words; zero between calls)
_ERR_WORDS = 16

ATOM_FEATURE_DIMS = (119, 5, 12, 12, 10, 6, 6, 2, 2)  # ogb >= 1.3
BOND_FEATURE_DIMS = (5, 6, 2)
TOWERS = 2  # models.py:76


@dataclasses.dataclass
class PnaconvsParams:
    """models.py:28-37"""
    propagation_depth: int
    pre_layers: int
    post_layers: int
    deg: torch.Tensor
    dropout: float = 0.0
    skip_connections: bool = False
    self_loops: bool = False


@dataclasses.dataclass
class ReadoutMLPParams:
    """models.py:40-45"""
    num_mlp_layers: int
    num_para: int
    dropout: float = 0.0


# --------------------------------------------------------------------------
# parameter containers (names only)
# --------------------------------------------------------------------------
class _ParamsOnly(nn.Module):
    def forward(self, *args, **kwargs):  # pragma: no cover - never part of the compute path
        raise RuntimeError(f"{type(self).__name__} only stores parameters; the arithmetic lives in libgnnsaft.so")


class _CategoricalTables(_ParamsOnly):
    def __init__(self, list_name: str, dims: Sequence[int], width: int):
        super().__init__()
        tables = nn.ModuleList()
        for d in dims:
            table = nn.Embedding(int(d), width)
            nn.init.xavier_uniform_(table.weight.data)  # ogb initialisation
            tables.append(table)
        setattr(self, list_name, tables)
        self._list_name = list_name

    def tables(self) -> List[nn.Embedding]:
        return list(getattr(self, self._list_name))


class _DegreeStatistics(_ParamsOnly):
    def __init__(self, deg: torch.Tensor):
        super().__init__()
        hist = deg.detach().to("cpu", torch.float)
        total = int(hist.sum())
        if total <= 0:
            raise ValueError("deg histogram must contain at least one node")
        bins = torch.arange(hist.numel())
        self.register_buffer("avg_deg_lin", torch.full((1,), float((bins * hist).sum()) / total))
        self.register_buffer("avg_deg_log", torch.full((1,), float(((bins + 1).log() * hist).sum()) / total))


def _tower_stack(n_in: int, width: int, depth: int) -> nn.Sequential:
    layers: List[nn.Module] = [nn.Linear(n_in, width)]
    for _ in range(depth - 1):
        layers.append(nn.ReLU())
        layers.append(nn.Linear(width, width))
    return nn.Sequential(*layers)


class _PNAConvWeights(_ParamsOnly):
    """Parameters of PyG ``PNAConv`` as built at models.py:69-80."""

    def __init__(self, hidden: int, deg: torch.Tensor, pre_layers: int, post_layers: int):
        super().__init__()
        self.aggr_module = _DegreeStatistics(deg)
        self.edge_encoder = nn.Linear(hidden, hidden)
        self.pre_nns = nn.ModuleList([_tower_stack(3 * hidden, hidden, pre_layers) for _ in range(TOWERS)])
        self.post_nns = nn.ModuleList(
            [_tower_stack(13 * hidden, hidden // TOWERS, post_layers) for _ in range(TOWERS)])
        self.lin = nn.Linear(hidden, hidden)


class _NodeBatchNormWeights(_ParamsOnly):
    """PyG ``BatchNorm`` keeps its BatchNorm1d under the attribute ``module``."""

    def __init__(self, hidden: int):
        super().__init__()
        self.module = nn.BatchNorm1d(hidden)


class PNAPCSAFT(nn.Module):
    """Graph neural network predicting ePC-SAFT parameters (models.py:48-135), MI355X kernels."""

    def __init__(self, hidden_dim: int, pna_params: PnaconvsParams, mlp_params: ReadoutMLPParams,
                 atom_feature_dims: Sequence[int] = ATOM_FEATURE_DIMS,
                 bond_feature_dims: Sequence[int] = BOND_FEATURE_DIMS):
        super().__init__()
        if hidden_dim % 32 != 0 or hidden_dim < 32:
            raise ValueError("hidden_dim must be a positive multiple of 32 (reference envelope: 64, 128, 256)")
        self.hidden_dim = int(hidden_dim)
        self.pna_params = pna_params
        self.mlp_params = mlp_params
        # registration order as in the reference (models.py:62-66: convs, batch_norms, node_embed, edge_embed, mlp):
        # it fixes ``parameters()`` order, hence the positional indices of torch optimizer state_dicts
        self.convs = nn.ModuleList()
        self.batch_norms = nn.ModuleList()
        self.node_embed = _CategoricalTables("atom_embedding_list", atom_feature_dims, hidden_dim)
        self.edge_embed = _CategoricalTables("bond_embedding_list", bond_feature_dims, hidden_dim)
        for _ in range(pna_params.propagation_depth):
            self.convs.append(_PNAConvWeights(hidden_dim, pna_params.deg, pna_params.pre_layers,
                                              pna_params.post_layers))
            self.batch_norms.append(_NodeBatchNormWeights(hidden_dim))
        h = hidden_dim
        self.mlp = nn.Sequential()
        for _ in range(mlp_params.num_mlp_layers):
            for mod in (nn.Linear(h, h), nn.BatchNorm1d(h), nn.ReLU(), nn.Dropout(p=mlp_params.dropout)):
                self.mlp.append(mod)
        tail: List[nn.Module] = []
        for n_in, n_out in ((h, h // 2), (h // 2, h // 4)):
            tail += [nn.Linear(n_in, n_out), nn.BatchNorm1d(n_out), nn.ReLU(), nn.Dropout(p=mlp_params.dropout)]
        tail.append(nn.Linear(h // 4, mlp_params.num_para))
        self.mlp.append(nn.Sequential(*tail))
        self._workspace: Optional[torch.Tensor] = None
        self._err_flag: Optional[torch.Tensor] = None
        self._retired_flags: list = []
        self._loss_buf: Optional[torch.Tensor] = None
        # Degree-folded update GEMM (K = 5F instead of 13F): exact for in-degrees (self-loop included) below
        # gnnsaft_degree_buckets() = 32, which covers molecular graphs.  Decided here, on the host, from the
        # training-set degree histogram the constructor is given anyway: if that histogram reaches the bucket limit
        # the forward uses the scalers-on-load kernel (K = 13F) instead.  A batch that exceeds the histogram it was
        # built for raises GNNSAFT_FLAG_BAD_DEGREE = 8 (input_error_flags(); training_loop checks it).
        max_deg = int(pna_params.deg.numel()) - 1 + int(bool(pna_params.self_loops))
        self.fold_degree_scalers = max_deg < int(lib.gnnsaft_degree_buckets())
        # Also fold the message's destination term W_dst x_dst (a per-node constant under mean/min/max, invisible
        # to std) into those weights: removes half of the message GEMM and a quarter of K4's reads.  Used when
        # fold_degree_scalers is on, pre_layers == 1 and hidden_dim % 64 == 0; otherwise ignored.
        self.fold_dst_term = True
        # Optional side stream for the structure chain (gnnsaft_aux).  Off by default: measured on MI355X the
        # hipGraph replay of the step is 0.490 ms with it and 0.491 ms without (C2), and the graph executor does
        # not reliably run the two branches concurrently (profiles/r01_c2_graph_replay_timeline.txt).
        self.use_side_stream = os.environ.get("GNNSAFT_SIDE_STREAM", "0") == "1"
        # Batch structure (CSR by destination, degree plan) built by cooperating workgroups of the forward's FIRST
        # launch beside the embedding work (csrc/elementwise.hip: k0_chain_body: one grid barrier, a look-back scan,
        # a ticket) instead of four dependent launches behind it.  Measured on MI355X (C2): 21 launches instead of 25,
        # head of the step 43 us instead of 57 us.  Needs this module's persistent flag buffer (_flag_buffer).
        self.fused_structure_chain = os.environ.get("GNNSAFT_K0_FUSED", "1") == "1"
        # gnnsaft_backward can run weight / bias gradients, edge-class sums and the edge-table chain on a side stream
        # (forked from and joined into the current stream inside the call).  True / False / None = decide per batch:
        # the ~10 event records + waits per layer cost the host more than the overlap gives the GPU on small batches
        # (measured, MI355X: C5 stand-in, 10 k nodes x 64 channels: 1.82 ms one stream, 2.3 ms two; C2, 20 k x 128:
        # 1.97 ms one stream, 1.82 ms two), so the side stream is used from ~2 M node-channels up.
        env = os.environ.get("GNNSAFT_BACKWARD_SIDE_STREAM", "auto")
        self.backward_side_stream = None if env == "auto" else env == "1"
        # backward fast path: set .grad to views of the one flat gradient buffer when every .grad is None
        self.direct_grads = True
        self._profile = None  # gnnsaft_profile* (bench.py attaches one to time kernels with HIP events)
        # add-pool -> readout MLP (train-mode BatchNorm across the batch through a grid barrier) -> MAPE in ONE launch
        # (csrc/readout.hip) for up to 16 384 graphs; False restores the nine per-op launches
        self.fused_readout = True
        # train-mode node BatchNorm: "pool" (default) = combine + apply launches, but the LAST layer's normalisation +
        # ReLU + residual is applied by the pooling kernel while it loads y (no x_L round trip); True = every layer's
        # applied on load by the next layer's message GEMM (measured slower: the GEMM's operand path is its
        # bottleneck); False = combine + apply everywhere.  All three give the same bits.
        self.fused_batchnorm = "pool"
        self._debug_barrier_extra = 0   # tests only: make the fused readout's grid barriers time out
        self._dropout_step = None       # device int64: mixed into the dropout key by kernels replayed from a hipGraph
        # Per-graph fused kernel (csrc/graph_eval.hip: one workgroup per molecule, whole network in one launch):
        # always for float64 modules; for float32 in eval mode without autograd when the input has at most this many
        # graphs and nodes.  Measured on MI355X (tools/single_molecule_latency.py, default model H=64 L=6): one
        # molecule of <= 8 atoms 120-150 us vs 210 us in the ~50-launch batched pipeline; beyond one 8-row tile the
        # single CU re-streams the layer's weights per tile and the batched pipeline wins again.  0 switches it off.
        self.graph_kernel_max_graphs = 1
        self.graph_kernel_max_nodes = 8
        self._eval_pack = None      # (key, packed weights): BatchNorm-folded, transposed; rebuilt when weights change
        self._pack_generation = 0   # bumped by whoever rewrites parameters behind torch's version counters
        self._graph_ws: Optional[torch.Tensor] = None
        self.gradient_segment_events = None   # L + 2 torch.cuda.Event: recorded by gnnsaft_backward per finished segment
        self._last_flat_grad: Optional[torch.Tensor] = None

    # ------------------------------------------------------------------ host glue
    def _weight_tensors(self) -> List[torch.Tensor]:
        """Canonical weight table order (csrc/forward.hip header).  Walks the module tree through the
        ``_modules`` / ``_parameters`` / ``_buffers`` dicts: ``nn.Module.__getattr__`` costs ~1 us per dot, which
        made this walk (89 tensors at L = 3) the largest single item of the per-call Python time.  Nothing is
        cached, so replaced parameters / submodules are always seen."""
        out: List[torch.Tensor] = []
        mods = self._modules

        def lin(m):       # nn.Linear
            p = m._parameters
            out.append(p["weight"])
            out.append(p["bias"])

        def bn(m):        # nn.BatchNorm1d
            if not (m.affine and m.track_running_stats):
                raise NotImplementedError("BatchNorm1d must be affine with running statistics (reference default)")
            p, b = m._parameters, m._buffers
            out.extend((p["weight"], p["bias"], b["running_mean"], b["running_var"], b["num_batches_tracked"]))

        for enc in (mods["node_embed"], mods["edge_embed"]):
            for t in enc._modules[enc._list_name]._modules.values():
                out.append(t._parameters["weight"])
        for conv, norm in zip(mods["convs"]._modules.values(), mods["batch_norms"]._modules.values()):
            cm = conv._modules
            out.append(cm["aggr_module"]._buffers["avg_deg_log"])
            lin(cm["edge_encoder"])
            for stacks in (cm["pre_nns"], cm["post_nns"]):
                for stack in stacks._modules.values():
                    for layer in stack._modules.values():
                        if isinstance(layer, nn.Linear):
                            lin(layer)
            lin(cm["lin"])
            bn(norm._modules["module"])
        seq = list(mods["mlp"]._modules.values())
        m = self.mlp_params.num_mlp_layers
        for i in range(m):
            lin(seq[4 * i])
            bn(seq[4 * i + 1])
        tail = list(seq[4 * m]._modules.values())
        lin(tail[0])
        bn(tail[1])
        lin(tail[4])
        bn(tail[5])
        lin(tail[8])
        return out

    def _model_desc(self) -> ModelDesc:
        d = ModelDesc()
        d.hidden = self.hidden_dim
        d.num_layers = len(self.convs)
        d.pre_layers = self.pna_params.pre_layers
        d.post_layers = self.pna_params.post_layers
        d.num_mlp_layers = self.mlp_params.num_mlp_layers
        d.num_para = self.mlp_params.num_para
        d.skip_connections = int(bool(self.pna_params.skip_connections))
        d.self_loops = int(bool(self.pna_params.self_loops))
        d.training = int(self.training)
        atom, bond = self.node_embed.tables(), self.edge_embed.tables()
        d.num_atom_cols, d.num_bond_cols = len(atom), len(bond)
        for k, t in enumerate(atom):
            d.atom_dims[k] = t.num_embeddings
        for k, t in enumerate(bond):
            d.bond_dims[k] = t.num_embeddings
        bn0 = self.batch_norms[0].module if len(self.batch_norms) else self.mlp[4 * d.num_mlp_layers][1]
        d.bn_eps = bn0.eps
        d.bn_eps_f64 = bn0.eps
        d.bn_momentum = 0.1 if bn0.momentum is None else bn0.momentum
        d.fold_degree_scalers = int(self.fold_degree_scalers)
        d.fold_dst_term = int(self.fold_dst_term)
        d.unfused_readout = int(not self.fused_readout)
        d.debug_barrier_extra = int(self._debug_barrier_extra)
        d.unfused_bn_apply = {"pool": 0, True: 2, False: 1}[self.fused_batchnorm]
        # Dropout of the readout MLP (models.py:88,95,99; config.dropout_rate through train/utils.py:66-70): a fresh
        # Philox key per training forward, drawn from torch's CPU generator (torch.manual_seed makes runs repeatable);
        # the backward regenerates the masks from the key kept in the tape's descriptor
        d.readout_dropout = float(self.mlp_params.dropout) if self.training else 0.0
        if d.readout_dropout > 0.0:
            if not 0.0 <= d.readout_dropout < 1.0:
                raise ValueError("dropout must be in [0, 1)")
            d.dropout_seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
            if torch.cuda.is_current_stream_capturing():
                # the seed is a kernel argument, frozen in the graph: the kernels add a DEVICE word to it when they
                # run, which bump_dropout_step() advances in front of every replay (GraphedTrainingStep does)
                if self._dropout_step is None:
                    raise RuntimeError("readout dropout under hipGraph capture: run one eager training forward first "
                                       "(it allocates the device word the replays' masks are keyed by)")
                d.dropout_step = self._dropout_step.data_ptr()
            elif self._dropout_step is None or self._dropout_step.device != torch.device("cuda", torch.cuda.current_device()):
                self._dropout_step = torch.zeros(1, dtype=torch.int64, device="cuda")
        return d

    def _check_mode(self, x: torch.Tensor) -> None:
        if not x.is_cuda:
            raise RuntimeError("PNAPCSAFT (MI355X build) has no CPU path: move the module and the batch to a HIP "
                               "device (`.to('cuda')`)")
        if self.training and self.pna_params.dropout > 0:
            raise NotImplementedError("dropout on the node states (PnaconvsParams.dropout) is not implemented: the "
                                      "reference's factory never sets it (train/utils.py:57-63 leaves the default "
                                      "0.0); config.dropout_rate reaches the readout MLP only, which is implemented")
        if self.training and self.mlp_params.dropout > 0 and not self.fused_readout:
            raise NotImplementedError("readout dropout runs in the one-launch readout only: leave fused_readout on")

    def _trainable(self, weights: List[torch.Tensor]) -> List[torch.Tensor]:
        """Parameters of the weight table that require a gradient (every parameter of the module is in the table;
        ``self.parameters()`` walks the module tree through ~90 generator frames, 0.25 ms per call)."""
        return [t for t in weights if isinstance(t, nn.Parameter) and t.requires_grad]

    def _needs_grad(self) -> bool:
        if not torch.is_grad_enabled():
            return False
        first = next(iter(self._modules["node_embed"].parameters()), None)   # the usual case: everything trains
        if first is not None and first.requires_grad:
            return True
        return any(isinstance(t, nn.Parameter) and t.requires_grad for t in self._weight_tensors())

    def _launch(self, data, target: Optional[torch.Tensor], tape: bool, weights: Optional[List[torch.Tensor]] = None):
        """One gnnsaft_forward call.  Returns (pred, loss3, ctx) where ctx holds what gnnsaft_backward needs
        when ``tape`` is set (the workspace doubles as the tape and is then private to this call)."""
        x = data.x
        edge_index = data.edge_index
        edge_attr = data.edge_attr
        batch = getattr(data, "batch", None)
        self._check_mode(x)
        dev = x.device
        if x.dtype != torch.int64 or edge_index.dtype != torch.int64 or edge_attr.dtype != torch.int64:
            raise TypeError("x, edge_index and edge_attr must be int64 categorical indices (data/graph.py:28-37)")
        x = x.contiguous()
        edge_index = edge_index.contiguous()
        edge_attr = edge_attr.contiguous()
        n, e = int(x.shape[0]), int(edge_index.shape[1])
        if batch is None:
            g = 1
        else:
            batch = batch.contiguous()
            g = getattr(data, "num_graphs", None)
            if g is None:
                g = int(batch[-1]) + 1  # sorted by PyG collate; the reference syncs here too (batch.max())
            g = int(g)
        desc = self._model_desc()
        if tape:
            if desc.hidden % 64 != 0 or desc.num_para > 8:
                raise NotImplementedError("backward needs hidden_dim % 64 == 0 and num_para <= 8 (the reference's "
                                          "envelope: 64 / 128 / 256, 3 or 5); run other shapes under torch.no_grad()")
            if not self.fold_degree_scalers:
                raise NotImplementedError("backward is implemented for the degree-folded update only (in-degrees "
                                          "below gnnsaft_degree_buckets() = 32); fold_degree_scalers is off for "
                                          "this model, run it under torch.no_grad()")
            desc.save_tape, desc.fold_degree_scalers, desc.fold_dst_term = 1, 1, 0
        if x.shape[1] != desc.num_atom_cols or edge_attr.shape[1] != desc.num_bond_cols:
            raise ValueError("x / edge_attr column counts do not match the embedding tables")
        if self.training and (n < 2 or g < 2):
            raise ValueError("Expected more than 1 value per channel when training")  # torch BatchNorm1d
        if weights is None:
            weights = self._weight_tensors()
        fdtype = weights[0].dtype
        for t in weights:
            if t.device != dev or not t.is_contiguous():
                raise RuntimeError("all parameters and buffers must be contiguous and on the input's device")
            if t.dtype != torch.int64 and (t.dtype != fdtype or fdtype not in (torch.float32, torch.float64)):
                raise NotImplementedError("parameters and buffers must be uniformly float32 or float64 "
                                          "(module.to(torch.float32) / .to(torch.float64))")
        if fdtype == torch.float64 or (not tape and not self.training and target is None and
                                       self._profile is None and 0 < g <= self.graph_kernel_max_graphs and
                                       n <= self.graph_kernel_max_nodes and
                                       desc.hidden <= 256 and
                                       getattr(data, "gnnsaft_structure", None) is None):
            if fdtype == torch.float64 and (tape or self.training or target is not None):
                raise NotImplementedError("float64 modules run the eval-mode forward only (model.eval() under "
                                          "torch.no_grad(), as evaluate_ensemble.py:67-77 / demo/utils.py:141-152 "
                                          "use them); train in float32")
            return self._launch_graph(desc, weights, fdtype, x, edge_index, edge_attr, batch, n, e, g), None, None
        nw = len(weights)
        if nw != lib.gnnsaft_num_weights(ctypes.byref(desc)):
            raise RuntimeError("internal error: weight table length mismatch")
        wtab = (ctypes.c_void_p * nw)(*[t.data_ptr() for t in weights])

        need = lib.gnnsaft_forward_workspace_bytes(ctypes.byref(desc), n, e, g)
        if need == 0:
            raise _native.GnnsaftError("configuration outside the supported shape envelope")
        if tape:
            ws = torch.empty(need + 256, dtype=torch.uint8, device=dev)
        else:
            if self._workspace is None or self._workspace.device != dev or self._workspace.numel() < need + 256:
                self._workspace = torch.empty(int(need * 1.25) + 256, dtype=torch.uint8, device=dev)
            ws = self._workspace
        self._flag_buffer(dev, n)
        # (the words behind the flag word: persistent barrier state + fill cursors of the cooperative structure chain)
        desc.persistent_sync_words = self._err_flag.numel() - 1 if self.fused_structure_chain else 0
        out = torch.empty((g, desc.num_para), dtype=torch.float32, device=dev)
        tgt_ptr, loss_ptr, loss = None, None, None
        if target is not None:
            target = target.reshape(-1, desc.num_para).to(torch.float32).contiguous()
            if target.shape[0] != g or target.device != dev:
                raise ValueError("target must be [num_graphs, num_para] on the input's device")
            loss = torch.empty(3, dtype=torch.float32, device=dev)
            tgt_ptr, loss_ptr = target.data_ptr(), loss.data_ptr()
        stream = torch.cuda.current_stream(dev).cuda_stream
        ws_ptr = (ws.data_ptr() + 255) // 256 * 256
        ws_bytes = ws.numel() - (ws_ptr - ws.data_ptr())
        structure = getattr(data, "gnnsaft_structure", None)      # cached by build_structure() / GraphLoader
        struct_ptr = None
        if structure is not None:
            if structure.device != dev or structure.dtype != torch.uint8 or not structure.is_contiguous() or \
                    structure.numel() != lib.gnnsaft_structure_bytes(ctypes.byref(desc), n, e, g):
                raise ValueError("data.gnnsaft_structure does not belong to this batch / model configuration")
            struct_ptr = structure.data_ptr()
        with torch.cuda.device(dev):
            aux = aux_for(dev.index if dev.index is not None else torch.cuda.current_device()) \
                if self.use_side_stream else None
            rc = lib.gnnsaft_forward(ctypes.byref(desc), wtab, nw, x.data_ptr(), edge_index.data_ptr() if e else None,
                                     edge_attr.data_ptr() if e else None,
                                     None if batch is None else batch.data_ptr(), n, e, g, tgt_ptr, out.data_ptr(),
                                     loss_ptr, self._err_flag.data_ptr(), ws_ptr, ws_bytes, struct_ptr, self._profile,
                                     aux, stream)
        check(rc, "gnnsaft_forward")
        ctx = None
        if tape:
            ctx = dict(desc=desc, weights=weights, wtab=wtab, x=x, batch=batch, n=n, e=e, g=g, ws=ws, ws_ptr=ws_ptr,
                       ws_bytes=ws_bytes, dev=dev)
        return out, loss, ctx

    def invalidate_eval_pack(self) -> None:
        """Forget the packed inference weights.  Needed only after parameters were rewritten behind torch's version
        counters (raw kernels, ``p.data`` tricks); in-place torch ops, ``load_state_dict``, ``.to()``, ``train()`` /
        ``eval()`` switches and this package's fused optimizers are noticed without it."""
        self._pack_generation += 1
        self._eval_pack = None

    def train(self, mode: bool = True):
        self._eval_pack = None      # weights move while training: the pack is rebuilt at the next inference call
        return super().train(mode)

    def _launch_graph(self, desc, weights, fdtype, x, edge_index, edge_attr, batch, n, e, g) -> torch.Tensor:
        """gnnsaft_graph_forward: eval-mode forward, one workgroup per graph, float32 or float64."""
        dev = x.device
        code = _native.DTYPE_F64 if fdtype == torch.float64 else _native.DTYPE_F32
        if desc.hidden > 256:
            raise NotImplementedError("the per-graph kernel (float64 modules) supports hidden_dim up to 256")
        key = (self._pack_generation, code, dev, tuple((t.data_ptr(), t._version) for t in weights))
        stream = torch.cuda.current_stream(dev).cuda_stream
        nw = len(weights)
        with torch.cuda.device(dev):
            if self._eval_pack is None or self._eval_pack[0] != key:
                nbytes = lib.gnnsaft_eval_pack_bytes(ctypes.byref(desc), code)
                if nbytes == 0:
                    raise _native.GnnsaftError("configuration outside the per-graph kernel's shape envelope")
                pack = torch.empty(nbytes + 16, dtype=torch.uint8, device=dev)
                wtab = (ctypes.c_void_p * nw)(*[t.data_ptr() for t in weights])
                pp = (pack.data_ptr() + 15) // 16 * 16
                check(lib.gnnsaft_eval_pack(ctypes.byref(desc), wtab, nw, code, pp, nbytes, stream), "gnnsaft_eval_pack")
                self._eval_pack = (key, pack, pp)
            pp = self._eval_pack[2]
            need = lib.gnnsaft_graph_forward_workspace_bytes(ctypes.byref(desc), code, n, e, g)
            if need == 0:
                raise _native.GnnsaftError("configuration outside the per-graph kernel's shape envelope")
            if self._graph_ws is None or self._graph_ws.device != dev or self._graph_ws.numel() < need + 256:
                self._graph_ws = torch.empty(int(need * 1.25) + 256, dtype=torch.uint8, device=dev)
            ws = self._graph_ws
            ws_ptr = (ws.data_ptr() + 255) // 256 * 256
            self._flag_buffer(dev)
            out = torch.empty((g, desc.num_para), dtype=fdtype, device=dev)
            check(lib.gnnsaft_graph_forward(ctypes.byref(desc), code, pp, x.data_ptr(),
                                            edge_index.data_ptr() if e else None, edge_attr.data_ptr() if e else None,
                                            None if batch is None else batch.data_ptr(), n, e, g, out.data_ptr(),
                                            self._err_flag.data_ptr(), ws_ptr, ws.numel() - (ws_ptr - ws.data_ptr()),
                                            stream), "gnnsaft_graph_forward")
        return out

    def flat_layout(self):
        """``(params, offsets, total)``: the slice of ONE flat f32 buffer each trainable tensor of the weight table
        owns (weight-table order, 64-float aligned).  ``gnnsaft_backward`` writes its gradients in this layout; the
        fused optimizers (train/optim.py) and the gradient all-reduce (parallel.py) adopt it, so that a training
        step touches one gradient buffer and one parameter buffer instead of ~50 tensors."""
        params, offsets, off = [], [], 0
        for t in self._weight_tensors():
            if t.dtype == torch.float32 and isinstance(t, nn.Parameter):
                params.append(t)
                offsets.append(off)
                off += (t.numel() + 63) // 64 * 64
        return params, offsets, off

    def _backward(self, ctx, grad_out: torch.Tensor):
        """gnnsaft_backward: gradients of every float parameter of the weight table (None for buffers)."""
        desc, weights, dev = ctx["desc"], ctx["weights"], ctx["dev"]
        nw = len(weights)
        # one flat buffer, gradients are views into it (one allocation; also the layout a flat all-reduce wants);
        # zero-filled so that the alignment gaps hold no garbage for whoever consumes the buffer whole
        f32 = torch.float32
        sizes = [t.numel() if (t.dtype == f32 and isinstance(t, nn.Parameter)) else 0 for t in weights]
        total = 0
        for s_ in sizes:
            total += (s_ + 63) // 64 * 64
        flat = torch.zeros(total, dtype=f32, device=dev)
        base = flat.data_ptr()
        grads, ptrs, off = [], [], 0
        for t, sz in zip(weights, sizes):
            if sz:
                # one op per view (slice + view would be two; ~70 views per step)
                grads.append(flat.as_strided(t.shape, t.stride(), off))
                ptrs.append(base + 4 * off)
                off += (sz + 63) // 64 * 64
            else:
                grads.append(None)
                ptrs.append(None)
        wtab = ctx["wtab"]
        gtab = (ctypes.c_void_p * nw)(*ptrs)
        need = lib.gnnsaft_backward_scratch_bytes(ctypes.byref(desc), ctx["n"], ctx["e"], ctx["g"])
        scratch = torch.empty(need + 256, dtype=torch.uint8, device=dev)
        sp = (scratch.data_ptr() + 255) // 256 * 256
        grad_out = grad_out.to(torch.float32).contiguous()
        stream = torch.cuda.current_stream(dev).cuda_stream
        events = None
        if self.gradient_segment_events is not None:   # set by parallel.OverlappedGradientExchange
            evs = self.gradient_segment_events
            if len(evs) != desc.num_layers + 2:
                raise ValueError("gradient_segment_events must hold num_layers + 2 events")
            events = (ctypes.c_void_p * len(evs))(*[e.cuda_event for e in evs])
        with torch.cuda.device(dev):
            two = self.backward_side_stream
            if two is None:
                two = ctx["n"] * desc.hidden >= 2_000_000
            aux = aux_for(dev.index if dev.index is not None else torch.cuda.current_device()) if two else None
            rc = lib.gnnsaft_backward(ctypes.byref(desc), wtab, gtab, nw, ctx["x"].data_ptr(),
                                      None if ctx["batch"] is None else ctx["batch"].data_ptr(), ctx["n"], ctx["e"],
                                      ctx["g"], grad_out.data_ptr(), ctx["ws_ptr"], ctx["ws_bytes"], sp,
                                      scratch.numel() - (sp - scratch.data_ptr()), events,
                                      None if self._err_flag is None else self._err_flag.data_ptr(), aux, stream)
        check(rc, "gnnsaft_backward")
        self._last_flat_grad = flat
        return grads

    def gradient_segments(self):
        """``[(start, end)]`` float offsets into the flat gradient buffer (``flat_layout``), in the order
        ``gnnsaft_backward`` completes them: readout, layer L-1 .. layer 0, embeddings."""
        params, offsets, total = self.flat_layout()
        n_embed = len(self.node_embed.tables()) + len(self.edge_embed.tables())
        per_layer = []
        idx = n_embed
        for conv in self.convs:
            k = sum(1 for _ in conv.parameters()) + 2            # + BatchNorm weight, bias
            per_layer.append((idx, idx + k))
            idx += k
        bounds = offsets + [total]
        segs = [(bounds[idx], total)]                               # readout
        for a, b in reversed(per_layer):
            segs.append((bounds[a], bounds[b]))
        segs.append((0, bounds[n_embed]))                           # embeddings
        return segs

    @torch.no_grad()
    def build_structure(self, data) -> torch.Tensor:
        """The batch STRUCTURE of ``data`` (CSR by destination, graph offsets, degree tiles: what depends on
        ``edge_index`` / ``edge_attr`` / ``batch`` only) as an opaque device tensor.  Attach it to a batch that will
        be seen again (``data.gnnsaft_structure = model.build_structure(data)``; ``GraphLoader(cache_structure=
        True)`` does so): every later forward / training step over that batch skips the CSR build.  Valid for this
        model's ``hidden_dim`` / ``self_loops`` and this batch only."""
        x, edge_index, edge_attr = data.x, data.edge_index.contiguous(), data.edge_attr.contiguous()
        batch = getattr(data, "batch", None)
        self._check_mode(x)
        dev = x.device
        n, e = int(x.shape[0]), int(edge_index.shape[1])
        g = 1 if batch is None else int(getattr(data, "num_graphs", None) or int(batch[-1]) + 1)
        desc = self._model_desc()
        need = lib.gnnsaft_forward_workspace_bytes(ctypes.byref(desc), n, e, g)
        nbytes = lib.gnnsaft_structure_bytes(ctypes.byref(desc), n, e, g)
        if need == 0 or nbytes == 0:
            raise _native.GnnsaftError("configuration outside the supported shape envelope")
        ws = torch.empty(need + 256, dtype=torch.uint8, device=dev)
        ws_ptr = (ws.data_ptr() + 255) // 256 * 256
        blob = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self._flag_buffer(dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            check(lib.gnnsaft_structure_build(ctypes.byref(desc), edge_index.data_ptr() if e else None,
                                              edge_attr.data_ptr() if e else None,
                                              None if batch is None else batch.contiguous().data_ptr(), n, e, g,
                                              blob.data_ptr(), self._err_flag.data_ptr(), ws_ptr,
                                              ws.numel() - (ws_ptr - ws.data_ptr()), stream),
                  "gnnsaft_structure_build")
        return blob

    def run(self, data, target: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        """Forward (+ MAPE loss when ``target`` [G,P] is given) WITHOUT autograd.  Returns
        ``(pred [G,P], loss3)`` with ``loss3 = [mape, sum(ape), numel]`` on device."""
        if self._needs_grad():
            raise RuntimeError("run() is the no-grad entry point; call the module (forward) to build a graph, or "
                               "wrap the call in torch.no_grad()")
        out, loss, _ = self._launch(data, target, tape=False)
        return out, loss

    def forward(self, data) -> torch.Tensor:
        """models.py:105-135.  ``data``: anything with ``x``, ``edge_index``, ``edge_attr`` and optionally
        ``batch`` / ``num_graphs`` attributes (PyG ``Data`` / ``Batch``)."""
        if torch.is_grad_enabled():
            weights = self._weight_tensors()
            params = self._trainable(weights)
            if params:
                return _PNAForwardFunction.apply(self, data, weights, *params)
        return self._launch(data, None, tape=False)[0]

    def bump_dropout_step(self) -> None:
        """In front of every replay of a hipGraph that holds a training forward with readout dropout (on the replay's
        stream): the kernels mix this device word into their Philox key, so that every replay draws fresh masks and
        its backward regenerates exactly those."""
        if self._dropout_step is not None:
            self._dropout_step.add_(1)

    def input_error_flags(self) -> int:
        """Synchronises, returns and clears the OR of the GNNSAFT_FLAG_* bits raised by the forwards since the
        last call (out-of-range indices are clamped, never dereferenced).  The flag word is not reset per forward:
        that would cost a launch at the head of every step for a word that is all but always zero."""
        if self._err_flag is None:
            return 0
        flags = 0
        for buf in [self._err_flag] + self._retired_flags:
            f = int(buf[0].item())
            if f:
                buf[0:1].zero_()   # only the sticky flag word: the library keeps the words behind it zero itself
            flags |= f
        return flags

    def _flag_buffer(self, dev, num_nodes: int = 0) -> torch.Tensor:
        """The module's int32 flag buffer on ``dev``: word 0 = the sticky GNNSAFT_FLAG_* word (cleared when read), the
        words behind it = state the cooperative structure chain keeps between calls (barrier words, one fill cursor
        per node; zero between calls).  Grown by replacement; a replaced buffer stays alive -- a captured hipGraph
        may still write its flag word -- and is read by ``input_error_flags`` too."""
        need = _ERR_WORDS + (num_nodes if self.fused_structure_chain else 0)
        buf = self._err_flag
        if buf is None or buf.device != dev or buf.numel() < need:
            if buf is not None and buf.device == dev:
                self._retired_flags.append(buf)
            self._err_flag = torch.zeros(need + need // 4, dtype=torch.int32, device=dev)
        return self._err_flag

    def _apply(self, fn, *args, **kwargs):
        self._workspace = None
        self._graph_ws = None
        self._eval_pack = None
        return super()._apply(fn, *args, **kwargs)


class _PNAForwardFunction(torch.autograd.Function):
    """Autograd node of the whole network: forward = gnnsaft_forward (tape kept), backward = gnnsaft_backward."""

    @staticmethod
    def forward(ctx, module, data, weights, *params):
        out, _, tape = module._launch(data, None, tape=True, weights=weights)
        ctx.module, ctx.tape, ctx.params = module, tape, params
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        grads = ctx.module._backward(ctx.tape, grad_out)
        by_id = {id(w): g for w, g in zip(ctx.tape["weights"], grads)}
        ctx.tape = None  # the tape (workspace) can be freed now
        out = [by_id.get(id(p)) for p in ctx.params]
        if ctx.module.direct_grads and all(p.grad is None and not p._backward_hooks for p in ctx.params):
            # Every .grad is unset (zero_grad(set_to_none=True), torch's and Lightning's default): hand the views of
            # the flat gradient buffer over as they are.  Returning them to autograd instead would make
            # AccumulateGrad clone each one (~50 copy launches) and scatter them over separate allocations.
            for p, g in zip(ctx.params, out):
                p.grad = g
            return (None, None, None) + (None,) * len(out)
        return (None, None, None) + tuple(out)   # accumulate into existing .grad the autograd way


class _MapeFunction(torch.autograd.Function):
    """torchmetrics MAPE (models.py:194) with its gradient: gnnsaft_mape / gnnsaft_mape_backward."""

    @staticmethod
    def forward(ctx, pred, target):
        pred = pred.contiguous()
        target = target.reshape(pred.shape).to(torch.float32).contiguous()
        out3 = torch.empty(3, dtype=torch.float32, device=pred.device)
        stream = torch.cuda.current_stream(pred.device).cuda_stream
        check(lib.gnnsaft_mape(pred.data_ptr(), target.data_ptr(), pred.numel(), out3.data_ptr(), stream),
              "gnnsaft_mape")
        ctx.save_for_backward(pred, target)
        return out3[0]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dloss):
        pred, target = ctx.saved_tensors
        dpred = torch.empty_like(pred)
        dloss = dloss.to(torch.float32).contiguous()
        stream = torch.cuda.current_stream(pred.device).cuda_stream
        check(lib.gnnsaft_mape_backward(pred.data_ptr(), target.data_ptr(), pred.shape[0], pred.shape[1],
                                        dloss.data_ptr(), dpred.data_ptr(), stream), "gnnsaft_mape_backward")
        return dpred, None


def mape_loss(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """mean(|pred - target| / max(|target|, 1.17e-6)) on the device, differentiable w.r.t. ``pred``."""
    return _MapeFunction.apply(pred, target)


class PNApcsaftL(nn.Module):
    """Host-side mirror of the LightningModule at models.py:138-251, limited to the
    hot path: ``forward`` (:155-160) and the MAPE loss of ``training_step``
    (:191-202).  Lightning itself is absent from the image; validation needs
    ``feos`` and is out of scope (component 2)."""

    def __init__(self, pna_params: PnaconvsParams, mlp_params: ReadoutMLPParams, config):
        super().__init__()
        self.config = config
        self.model = PNAPCSAFT(_cfg(config, "hidden_dim"), pna_params=pna_params, mlp_params=mlp_params)

    def forward(self, data) -> torch.Tensor:
        return self.model(data)

    def training_step(self, graphs, batch_idx=None) -> torch.Tensor:
        """target = graphs.para.view(-1, num_para); returns mean |pred - target| / max(|target|, 1.17e-6)."""
        target = graphs.para.view(-1, _cfg(self.config, "num_para"))
        if self.model._needs_grad():   # builds the autograd graph: loss.backward() runs gnnsaft_backward
            return mape_loss(self.model(graphs), target)
        _, loss = self.model.run(graphs, target=target)
        return loss[0]

    def training_step_parts(self, graphs) -> torch.Tensor:
        """[mape, sum of absolute percentage errors, element count] -- the pair that is all-reduced
        across ranks (``sync_dist=True`` at models.py:195-201)."""
        return self.model.run(graphs, target=graphs.para.view(-1, _cfg(self.config, "num_para")))[1]

    def configure_optimizers(self):
        """models.py:162-188."""
        from .optim import FusedAdamW, FusedSGD
        opt_name = _cfg(self.config, "optimizer")
        # Parameters in ``self.parameters()`` order -- the order the reference hands to torch.optim, which indexes
        # its state_dict by position -- each with the flat-buffer offset gnnsaft_backward writes its gradient at.
        table_params, table_offsets, total = self.model.flat_layout()
        offset_of = {id(p): off for p, off in zip(table_params, table_offsets)}
        params = list(self.parameters())
        if any(not p.requires_grad for p in params):      # frozen tensors: let the optimizer lay out what is left
            params, layout = [p for p in params if p.requires_grad], None
        else:
            layout = ([offset_of[id(p)] for p in params], total)
        if opt_name == "adam":
            opt = FusedAdamW(params, lr=_cfg(self.config, "learning_rate"),
                             weight_decay=_cfg(self.config, "weight_decay"), amsgrad=True, eps=1e-5, layout=layout)
        elif opt_name == "sgd":
            opt = FusedSGD(params, lr=_cfg(self.config, "learning_rate"), momentum=_cfg(self.config, "momentum"),
                           weight_decay=_cfg(self.config, "weight_decay"), nesterov=True, layout=layout)
        else:
            raise ValueError(f"Unsupported optimizer: {opt_name}.")
        opt.on_parameters_rewritten = self.model.invalidate_eval_pack   # fused steps bypass torch's version counters
        sched = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(opt, _cfg(self.config, "warmup_steps"))
        return {"optimizer": opt, "lr_scheduler": {"scheduler": sched, "interval": "step", "frequency": 1}}

    def validation_step(self, graphs, batch_idx=None):
        raise NotImplementedError("validation scores predictions with feos on the CPU (models.py:204-248); "
                                  "out of scope for the MI355X hot path")


def _cfg(config, name: str):
    """ml_collections.ConfigDict, dict or attribute bag."""
    if isinstance(config, dict):
        return config[name]
    return getattr(config, name)
