"""gnn-epc-saft_amd: MI355X-native PNAPCSAFT forward + MAPE loss.

Import name ``gnn_epc_saft_amd`` (see the shim module at the repository root;
the directory name carries a hyphen).  Importing loads ``lib/libgnnsaft.so``
and raises if it is missing -- there is no CPU fallback.
"""

from . import _native  # noqa: F401  (fails loudly when the HIP library is absent)
from .data.loader import GraphLoader, PackedGraphs  # noqa: F401
from .data.synthetic import GraphData, collate, make_synthetic_batch  # noqa: F401
from .train.models import PNAPCSAFT, PNApcsaftL, PnaconvsParams, ReadoutMLPParams  # noqa: F401
from .train.checkpoint import load_checkpoint, save_checkpoint  # noqa: F401
from .train.loop import GraphedTrainingStep, training_loop  # noqa: F401
from .train.optim import FusedAdamW, FusedSGD  # noqa: F401
from .train.utils import calc_deg, create_model  # noqa: F401

__all__ = ["PNAPCSAFT", "PNApcsaftL", "PnaconvsParams", "ReadoutMLPParams", "create_model", "calc_deg",
           "GraphData", "GraphLoader", "PackedGraphs", "collate", "make_synthetic_batch", "training_loop", "GraphedTrainingStep", "FusedAdamW", "FusedSGD",
           "load_checkpoint", "save_checkpoint"]
