"""ctypes binding of ``lib/libgnnsaft.so`` (C ABI declared in ``include/gnnsaft.h``).

There is deliberately NO fallback: if the shared library is missing or a symbol
cannot be resolved, importing this module raises.  The library is built
in-tree by ``__graft_entry__.build()`` / ``make -C gnn-epc-saft_amd/csrc``.
"""

from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int32, c_int64, c_size_t, c_void_p

# torch FIRST: its wheel bundles its own libamdhip64 / libhsa-runtime64.  Loaded after torch, libgnnsaft.so binds
# to that already-resident HIP runtime (same soname) and shares torch's streams, events and allocations.  Loaded
# before torch, the dynamic linker would pull /opt/rocm's runtime in for this library and torch would then add its
# own: two HIP runtimes in one process (measured on the GPU box: hipStreamCreate in the second one fails with
# "no ROCm-capable device is detected", and stream handles would cross runtimes).
import torch  # noqa: F401  (import order matters, see above)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GNNSAFT_LIB") or os.path.join(_HERE, "lib", "libgnnsaft.so")  # override: A/B runs of two builds

MAX_TABLES = 16
ABI_VERSION = 5


class ModelDesc(ctypes.Structure):
    """``gnnsaft_model_desc`` (include/gnnsaft.h)."""

    _fields_ = [
        ("hidden", c_int32), ("num_layers", c_int32), ("pre_layers", c_int32), ("post_layers", c_int32),
        ("num_mlp_layers", c_int32), ("num_para", c_int32), ("skip_connections", c_int32),
        ("self_loops", c_int32), ("training", c_int32), ("num_atom_cols", c_int32), ("num_bond_cols", c_int32),
        ("atom_dims", c_int32 * MAX_TABLES), ("bond_dims", c_int32 * MAX_TABLES),
        ("bn_eps", c_float), ("bn_momentum", c_float), ("fold_degree_scalers", c_int32),
        ("fold_dst_term", c_int32), ("save_tape", c_int32), ("unfused_readout", c_int32), ("bn_eps_f64", ctypes.c_double),
        ("debug_barrier_extra", c_int32), ("readout_dropout", c_float), ("dropout_seed", ctypes.c_uint64),
        ("unfused_bn_apply", c_int32), ("persistent_sync_words", c_int32),
        ("dropout_step", c_void_p),
    ]


class WorkspaceMap(ctypes.Structure):
    """``gnnsaft_workspace_map`` (include/gnnsaft.h)."""

    _fields_ = [(n, c_size_t) for n in (
        "rowptr", "src", "dst", "combo", "log_amp", "log_att", "graph_ptr", "x_embed", "x_final", "pq", "agg", "u",
        "y", "rtab", "pooled", "total", "ro", "x_stride", "bnstat", "y_stride", "ry", "rstat")]


P = c_void_p
# name -> (restype, argtypes); one entry per symbol include/gnnsaft.h declares
SIGNATURES = {
    "gnnsaft_abi_version": (c_int32, []),
    "gnnsaft_error_string": (c_char_p, [c_int32]),
    "gnnsaft_csr_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "gnnsaft_csr_build": (c_int32, [P, P, c_int64, c_int64, c_int32, POINTER(c_int32), c_int32, P, P, P, P, P, P, P, P,
                                    c_size_t, P]),
    "gnnsaft_batch_to_ptr": (c_int32, [P, c_int64, c_int64, P, P, P]),
    "gnnsaft_embed_sum": (c_int32, [P, c_int64, c_int32, POINTER(c_void_p), POINTER(c_int32), c_int32, P, P, P]),
    "gnnsaft_bond_combo_embed": (c_int32, [c_int32, POINTER(c_void_p), POINTER(c_int32), c_int32, P, P]),
    "gnnsaft_linear": (c_int32, [P, c_int64, c_int32, P, c_int64, P, P, c_int64, c_int64, c_int32, c_int32, P, P,
                                 c_int32, P, c_int64, P, P]),
    "gnnsaft_bn_rows_per_group": (c_int32, []),
    "gnnsaft_pna_node_terms": (c_int32, [P, c_int64, c_int32, P, P, P, P]),
    "gnnsaft_pna_edge_table": (c_int32, [P, c_int32, c_int32, P, P, P, P, P, P, P, P, P]),
    "gnnsaft_pna_edge_mlp": (c_int32, [P, P, P, c_int64, c_int32, P, P, P, P, P, P, P, P]),
    "gnnsaft_pna_edge_preact": (c_int32, [P, P, P, c_int64, c_int32, P, P, P, P]),
    "gnnsaft_pna_aggregate": (c_int32, [P, P, P, c_int64, c_int32, P, P, P, P, P]),
    "gnnsaft_pna_update": (c_int32, [P, P, P, P, P, c_int64, c_int32, P, P, P, P, P, P]),
    "gnnsaft_degree_buckets": (c_int32, []),
    "gnnsaft_degree_tiles_capacity": (c_int64, [c_int64, c_int32]),
    "gnnsaft_degree_scratch_ints": (c_size_t, [c_int64]),
    "gnnsaft_degree_tiles": (c_int32, [P, c_int64, c_int32, P, P, P, P, P, P]),
    "gnnsaft_pna_fold_post_weights": (c_int32, [P, P, P, P, c_int32, P, P]),
    "gnnsaft_pna_update_folded": (c_int32, [P, P, P, P, P, c_int64, c_int32, P, P, P, P, P]),
    "gnnsaft_debug_linear_tile": (c_int32, [P, c_int64, P, c_int64, P, P, c_int64, c_int64, c_int32, c_int32, P,
                                            c_int32, P]),
    "gnnsaft_w3_image_bytes": (c_size_t, [c_int64, c_int64]),
    "gnnsaft_w3_pack": (c_int32, [P, c_int64, c_int32, c_int32, P, P]),
    "gnnsaft_debug_linear_w3": (c_int32, [P, c_int64, P, P, P, c_int64, c_int64, c_int32, c_int32, P, c_int32, P]),
    "gnnsaft_sum_rows_by_class": (c_int32, [P, c_int32, P, c_int64, c_int64, c_int32, P, c_int64, P, c_size_t, c_int32, P]),
    "gnnsaft_pna_update_folded_ar": (c_int32, [P, P, P, P, P, c_int64, c_int32, P, P, P, P, P]),
    "gnnsaft_debug_linear_ar": (c_int32, [P, c_int64, P, P, P, c_int64, c_int64, c_int32, c_int32, c_int32, P]),
    "gnnsaft_debug_ar_stamps": (c_int32, [P]),
    "gnnsaft_debug_linear_w3s": (c_int32, [P, c_int64, P, P, P, c_int64, c_int64, c_int32, c_int32, P, c_int32, P]),
    "gnnsaft_pna_update_agg": (c_int32, [P, P, P, c_int32, P, P, P, P, P, P, c_int64, c_int32, P, P, P, P, P]),
    "gnnsaft_debug_update_agg_stamps": (c_int32, [P]),
    "gnnsaft_bn_finalize": (c_int32, [P, c_int64, c_int32, P, P, P, P, P, c_float, c_float, c_int32, P, P, P]),
    "gnnsaft_bn_train_scratch_bytes": (c_size_t, [c_int64, c_int32]),
    "gnnsaft_bn_train_apply": (c_int32, [P, P, c_int64, c_int32, P, P, P, P, P, c_float, c_float, P, P, P, P, c_size_t,
                                         P]),
    "gnnsaft_pna_fold_post_weights_multi": (c_int32, [c_int32, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
                                                      POINTER(c_void_p), POINTER(c_void_p), P, P, c_int32, P, c_int64,
                                                      P]),
    "gnnsaft_pna_src_terms": (c_int32, [P, c_int64, c_int32, P, P, P, P]),
    "gnnsaft_pna_aggregate_src": (c_int32, [P, P, P, c_int64, c_int32, P, P, P, P]),
    "gnnsaft_bn_relu_residual": (c_int32, [P, P, P, P, P, c_int64, c_int32, P]),
    "gnnsaft_add_pool": (c_int32, [P, P, c_int64, c_int64, c_int32, P, P]),
    "gnnsaft_mape": (c_int32, [P, P, c_int64, P, P]),
    "gnnsaft_num_weights": (c_int32, [POINTER(ModelDesc)]),
    "gnnsaft_readout_resident_workgroups": (c_int32, [c_int32, c_int32]),
    "gnnsaft_forward_workspace_bytes": (c_size_t, [POINTER(ModelDesc), c_int64, c_int64, c_int64]),
    "gnnsaft_forward_workspace_map": (c_int32, [POINTER(ModelDesc), c_int64, c_int64, c_int64, POINTER(WorkspaceMap)]),
    "gnnsaft_forward": (c_int32, [POINTER(ModelDesc), POINTER(c_void_p), c_int32, P, P, P, P, c_int64, c_int64,
                                  c_int64, P, P, P, P, P, c_size_t, P, P, P, P]),
    "gnnsaft_structure_bytes": (c_size_t, [POINTER(ModelDesc), c_int64, c_int64, c_int64]),
    "gnnsaft_structure_build": (c_int32, [POINTER(ModelDesc), P, P, P, c_int64, c_int64, c_int64, P, P, P, c_size_t,
                                          P]),
    "gnnsaft_adamw_step": (c_int32, [P, P, P, P, P, c_int64, c_float, c_float, c_float, c_float, c_float, c_int64,
                                     c_float, P]),
    "gnnsaft_adamw_args_floats": (c_int32, []),
    "gnnsaft_adamw_args": (c_int32, [c_float, c_float, c_float, c_float, c_float, c_int64, c_float, P, P]),
    "gnnsaft_adamw_step_dev": (c_int32, [P, P, P, P, P, c_int64, P, P]),
    "gnnsaft_sgd_step": (c_int32, [P, P, P, c_int64, c_float, c_float, c_float, c_int32, c_float, P]),
    "gnnsaft_aux_create": (c_int32, [POINTER(c_void_p)]),
    "gnnsaft_aux_destroy": (None, [P]),
    "gnnsaft_backward_scratch_bytes": (c_size_t, [POINTER(ModelDesc), c_int64, c_int64, c_int64]),
    "gnnsaft_backward": (c_int32, [POINTER(ModelDesc), POINTER(c_void_p), POINTER(c_void_p), c_int32, P, P, c_int64,
                                   c_int64, c_int64, P, P, c_size_t, P, c_size_t, POINTER(c_void_p), P, P, P]),
    "gnnsaft_mape_backward": (c_int32, [P, P, c_int64, c_int32, P, P, P]),
    "gnnsaft_wgrad_scratch_bytes": (c_size_t, [c_int64, c_int32, c_int32]),
    "gnnsaft_debug_linear_wgrad": (c_int32, [P, c_int64, P, c_int64, c_int64, c_int32, c_int32, P, c_int64, P, c_size_t,
                                             c_int32, c_int32, c_int64, P]),
    "gnnsaft_linear_wgrad": (c_int32, [P, c_int64, P, c_int64, c_int32, c_int64, c_int32, c_int32, P, c_int64, c_int32,
                                       P, P, c_size_t, P]),
    "gnnsaft_eval_pack_bytes": (c_size_t, [POINTER(ModelDesc), c_int32]),
    "gnnsaft_eval_pack": (c_int32, [POINTER(ModelDesc), POINTER(c_void_p), c_int32, c_int32, P, c_size_t, P]),
    "gnnsaft_graph_forward_workspace_bytes": (c_size_t, [POINTER(ModelDesc), c_int32, c_int64, c_int64, c_int64]),
    "gnnsaft_graph_forward": (c_int32, [POINTER(ModelDesc), c_int32, P, P, P, P, P, c_int64, c_int64, c_int64, P, P, P,
                                        c_size_t, P]),
    "gnnsaft_profile_create": (c_int32, [c_int32, ctypes.c_uint32, POINTER(c_void_p)]),
    "gnnsaft_profile_destroy": (None, [P]),
    "gnnsaft_profile_reset": (c_int32, [P]),
    "gnnsaft_profile_summary": (c_int32, [P, ctypes.c_uint32, POINTER(c_int32), POINTER(c_float)]),
}

PROF_AGGREGATE, PROF_UPDATE, PROF_NODE_TERMS, PROF_LIN, PROF_UPDATE_AGG = 1, 2, 4, 8, 16
DTYPE_F32, DTYPE_F64 = 0, 1


def _load() -> ctypes.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the MI355X library has not been built. Run "
            f"`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C gnn-epc-saft_amd/csrc`). "
            f"This package has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.gnnsaft_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI version {lib.gnnsaft_abi_version()} != {ABI_VERSION}; rebuild the library")
    return lib


lib = _load()


class GnnsaftError(RuntimeError):
    pass


def check(code: int, what: str) -> None:
    if code != 0:
        msg = lib.gnnsaft_error_string(code)
        raise GnnsaftError(f"{what} failed: {msg.decode() if msg else code} (code {code})")


_AUX = {}


def aux_for(device_index: int):
    """gnnsaft_aux handle (side stream + fork/join events) of a device; created on first use, with that device
    current, and kept for the life of the process.  One per device: the boundary contract is one forward at a
    time per process per device ("Threading"), and the handle must exist before a hipGraph
    capture starts (stream creation is not capturable) -- the warm-up forward every capture needs creates it."""
    key = int(device_index)
    h = _AUX.get(key)
    if h is None:
        out = c_void_p()
        check(lib.gnnsaft_aux_create(ctypes.byref(out)), "gnnsaft_aux_create")
        h = _AUX[key] = out
    return h
