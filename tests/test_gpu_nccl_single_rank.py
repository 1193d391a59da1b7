"""The RCCL backend on a one-GPU box: one rank under ``torch.distributed.run`` with the collectives of the data-parallel
path forced through "nccl" (tests/nccl_single_rank_worker.py).  No scaling number can come out of this; what it
establishes is that communicator creation, the asynchronous loss all-reduce behind hipGraph replays, the training-state
broadcast, the flat 28 MB gradient all-reduce and the segment-wise overlapped exchange execute on real RCCL beside
the default forward (cooperative structure chain on) and change no bit.  Replaces, for this box, what Lightning DDP
does for the reference (/root/reference/gnnepcsaft/train/train.py:142-156; sync_dist at models.py:195-201)."""

import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_one_rank_on_the_rccl_backend_runs_every_collective_and_changes_no_bit():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PYTHONPATH=ROOT, GNNSAFT_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("GNNSAFT_K0_FUSED", None)      # the default forward: cooperative structure chain
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "nccl_single_rank_worker.py")]
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    print(p.stdout[-1500:])
    assert p.returncode == 0 and "NCCL_SINGLE_OK" in p.stdout, p.stdout[-4000:]
    assert "flat gradient all-reduce on RCCL" in p.stdout
