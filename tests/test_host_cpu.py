"""CPU tests of the host side: C-ABI library loads and exports every declared symbol, the module's
state_dict matches the reference key map, factory / degree histogram / synthetic
data / sharding logic, and the two-rank gloo rehearsal of the N > 1 path."""

import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_symbol_declared_in_the_header():
    from gnn_epc_saft_amd import _native
    header = open(os.path.join(ROOT, "include", "gnnsaft.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(gnnsaft_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 30
    lib = ctypes.CDLL(_native.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/gnnsaft.h but not exported"
        assert name in _native.SIGNATURES, f"{name} has no ctypes signature in _native.py"
    assert set(_native.SIGNATURES) <= declared
    # ... and NOTHING else: the library is built with -fvisibility=hidden + csrc/exports.map, so the dynamic symbol
    # table is exactly the C ABI (no mangled launchers / kernel handles another HIP library could interpose)
    import subprocess
    nm = subprocess.run(["nm", "-D", "--defined-only", _native.LIB_PATH], check=True, capture_output=True, text=True)
    exported = {line.split()[-1] for line in nm.stdout.splitlines() if line.strip()}
    assert exported == declared, sorted(exported ^ declared)[:10]
    assert _native.lib.gnnsaft_abi_version() == _native.ABI_VERSION == 5
    assert _native.lib.gnnsaft_error_string(-2) == b"workspace too small"


def test_host_only_sizing_functions():
    from gnn_epc_saft_amd import _native
    import gnn_epc_saft_amd as G
    deg = torch.tensor([0, 5, 6, 3, 1])
    m = G.PNAPCSAFT(128, G.PnaconvsParams(3, 1, 1, deg, skip_connections=True, self_loops=True),
                    G.ReadoutMLPParams(1, 3))
    desc = m._model_desc()
    assert _native.lib.gnnsaft_num_weights(ctypes.byref(desc)) == len(m._weight_tensors())
    small = _native.lib.gnnsaft_forward_workspace_bytes(ctypes.byref(desc), 100, 200, 5)
    big = _native.lib.gnnsaft_forward_workspace_bytes(ctypes.byref(desc), 20480, 40960, 1024)
    assert 0 < small < big < 2 ** 31
    wmap = _native.WorkspaceMap()
    assert _native.lib.gnnsaft_forward_workspace_map(ctypes.byref(desc), 20480, 40960, 1024, ctypes.byref(wmap)) == 0
    assert wmap.total == big and wmap.agg % 256 == 0
    desc.hidden = 100  # not a multiple of 32: outside the envelope
    assert _native.lib.gnnsaft_forward_workspace_bytes(ctypes.byref(desc), 100, 200, 5) == 0
    for mp, pp, mm in ((2, 3, 0), (1, 1, 2)):
        m2 = G.PNAPCSAFT(64, G.PnaconvsParams(2, mp, pp, deg), G.ReadoutMLPParams(mm, 5))
        d2 = m2._model_desc()
        assert _native.lib.gnnsaft_num_weights(ctypes.byref(d2)) == len(m2._weight_tensors())


def expected_keys(hidden, depth, pre, post, mlp):
    keys = [f"node_embed.atom_embedding_list.{k}.weight" for k in range(9)]
    keys += [f"edge_embed.bond_embedding_list.{k}.weight" for k in range(3)]
    for l in range(depth):
        c = f"convs.{l}"
        keys += [f"{c}.aggr_module.avg_deg_lin", f"{c}.aggr_module.avg_deg_log", f"{c}.edge_encoder.weight",
                 f"{c}.edge_encoder.bias"]
        for t in range(2):
            for j in range(pre):
                keys += [f"{c}.pre_nns.{t}.{2 * j}.weight", f"{c}.pre_nns.{t}.{2 * j}.bias"]
        for t in range(2):
            for j in range(post):
                keys += [f"{c}.post_nns.{t}.{2 * j}.weight", f"{c}.post_nns.{t}.{2 * j}.bias"]
        keys += [f"{c}.lin.weight", f"{c}.lin.bias"]
    bn = ("weight", "bias", "running_mean", "running_var", "num_batches_tracked")
    for l in range(depth):
        keys += [f"batch_norms.{l}.module.{s}" for s in bn]
    for i in range(mlp):
        keys += [f"mlp.{4 * i}.weight", f"mlp.{4 * i}.bias"] + [f"mlp.{4 * i + 1}.{s}" for s in bn]
    f = f"mlp.{4 * mlp}"
    keys += [f"{f}.0.weight", f"{f}.0.bias"] + [f"{f}.1.{s}" for s in bn]
    keys += [f"{f}.4.weight", f"{f}.4.bias"] + [f"{f}.5.{s}" for s in bn]
    keys += [f"{f}.8.weight", f"{f}.8.bias"]
    return keys


@pytest.mark.parametrize("cfg", [(64, 6, 1, 1, 1, 5), (128, 2, 2, 3, 0, 3), (32, 1, 3, 2, 2, 5)])
def test_state_dict_key_map_matches_reference_and_oracle(cfg):
    import gnn_epc_saft_amd as G
    from oracle.pna_torch import OracleMlpParams, OraclePNAPCSAFT, OraclePnaParams
    hidden, depth, pre, post, mlp, p = cfg
    deg = torch.tensor([0, 5, 6, 3, 1])
    hip = G.PNAPCSAFT(hidden, G.PnaconvsParams(depth, pre, post, deg), G.ReadoutMLPParams(mlp, p))
    ora = OraclePNAPCSAFT(hidden, OraclePnaParams(depth, pre, post, deg), OracleMlpParams(mlp, p))
    want = expected_keys(hidden, depth, pre, post, mlp)
    assert sorted(hip.state_dict().keys()) == sorted(want)
    assert sorted(ora.state_dict().keys()) == sorted(want)
    for k, v in ora.state_dict().items():
        assert hip.state_dict()[k].shape == v.shape, k
    sd = hip.state_dict()
    assert sd["convs.0.pre_nns.0.0.weight"].shape == (hidden, 3 * hidden)
    assert sd["convs.0.post_nns.1.0.weight"].shape == (hidden // 2, 13 * hidden)
    assert sd[f"mlp.{4 * mlp}.8.weight"].shape == (p, hidden // 4)
    # a Lightning checkpoint prefixes the same keys with "model."
    lit = G.PNApcsaftL(G.PnaconvsParams(depth, pre, post, deg), G.ReadoutMLPParams(mlp, p),
                       dict(hidden_dim=hidden, num_para=p))
    assert sorted(lit.state_dict().keys()) == sorted("model." + k for k in want)
    # PyG's buffers: avg_deg_log = sum(log(bin + 1) * deg) / sum(deg)
    ref = float((torch.log(torch.arange(5.0) + 1) * deg).sum() / deg.sum())
    assert abs(float(sd["convs.0.aggr_module.avg_deg_log"]) - ref) < 1e-6
    assert torch.equal(sd["convs.0.aggr_module.avg_deg_log"], ora.state_dict()["convs.0.aggr_module.avg_deg_log"])


def test_create_model_and_calc_deg_contract():
    import gnn_epc_saft_amd as G
    from gnn_epc_saft_amd.data.synthetic import degree_histogram, ethanol_all_atom, ethanol_heavy, make_synthetic_batch
    graphs = [ethanol_heavy(), ethanol_all_atom()]
    deg = G.calc_deg(graphs)
    assert deg.tolist() == [0, 8, 2, 0, 2]       # raw in-degrees, no self-loops: 8 leaves, CH2 + OH, two sp3 carbons
    assert torch.equal(deg, degree_histogram(graphs))
    with pytest.raises(NotImplementedError):
        G.calc_deg("ramirez", "/tmp")
    cfg = dict(propagation_depth=6, hidden_dim=64, pre_layers=1, post_layers=1, num_mlp_layers=1, num_para=5,
               skip_connections=True, add_self_loops=True, dropout_rate=0.0, model="PNA")
    m = G.create_model(cfg, deg)                   # configs/default.py shape
    assert isinstance(m, G.PNAPCSAFT) and sum(p.numel() for p in m.parameters()) == 537829 - 0
    cfg["model"] = "PNAL"
    assert isinstance(G.create_model(cfg, deg), G.PNApcsaftL)
    cfg["model"] = "GCN"
    with pytest.raises(ValueError):
        G.create_model(cfg, deg)
    lit = G.create_model(dict(cfg, model="PNAL", optimizer="adam", learning_rate=1e-3, weight_decay=1e-2,
                              warmup_steps=100, momentum=0.9), deg)
    with pytest.raises(RuntimeError, match="HIP device"):   # fused optimizers live on the GPU: no CPU fallback
        lit.configure_optimizers()
    with pytest.raises(ValueError, match="Unsupported optimizer"):
        G.create_model(dict(cfg, model="PNAL", optimizer="lion", learning_rate=1e-3, weight_decay=1e-2,
                            warmup_steps=100, momentum=0.9), deg).configure_optimizers()
    b = make_synthetic_batch(50, 3)
    assert b.x.shape[1] == 9 and b.edge_attr.shape[1] == 3 and b.edge_index.shape[1] % 2 == 0
    assert bool((b.edge_index[0, 0::2] == b.edge_index[1, 1::2]).all())       # (i,j),(j,i) per bond
    assert int(torch.bincount(b.edge_index[1]).max()) <= 4                    # valence cap
    assert 12 <= b.x.shape[0] / 50 <= 28 and b.ptr[-1] == b.x.shape[0]


def test_module_fails_loudly_without_a_device():
    import gnn_epc_saft_amd as G
    from gnn_epc_saft_amd.data.synthetic import ethanol_heavy
    m = G.PNAPCSAFT(64, G.PnaconvsParams(1, 1, 1, torch.tensor([0, 2, 1])), G.ReadoutMLPParams(0, 3)).eval()
    with pytest.raises(RuntimeError, match="no CPU path"):
        with torch.no_grad():
            m(ethanol_heavy())
    with pytest.raises(ValueError):
        G.PNAPCSAFT(100, G.PnaconvsParams(1, 1, 1, torch.tensor([0, 2, 1])), G.ReadoutMLPParams(0, 3))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gnn-epc-saft_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in re.sub(r"#.*|//.*", "", text).replace("oracle/", ""), os.path.join(dirpath, f)


def test_split_graphs_partitions_a_batch():
    from gnn_epc_saft_amd.data.synthetic import make_synthetic_batch, split_graphs
    b = make_synthetic_batch(37, 9)
    parts = [split_graphs(b, 4, r) for r in range(4)]
    assert sum(p.num_graphs for p in parts) == 37
    assert sum(p.x.shape[0] for p in parts) == b.x.shape[0]
    assert sum(p.edge_index.shape[1] for p in parts) == b.edge_index.shape[1]
    for p in parts:
        assert int(p.edge_index.max()) < p.x.shape[0] and int(p.batch.max()) == p.num_graphs - 1
        assert p.para.numel() == 3 * p.num_graphs and int(p.ptr[-1]) == p.x.shape[0]


def test_two_rank_gloo_rehearsal():
    """world_size = 2 on the CPU with gloo: graph sharding, the [sum(ape), count] all-reduce and the flat
    gradient all-reduce (the collectives bench.py / a training loop issue over RCCL)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", PYTHONPATH=ROOT)
    worker = os.path.join(ROOT, "tests", "gloo_worker.py")
    procs = [subprocess.Popen([sys.executable, worker], env=dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2"),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    assert "GLOO_OK" in outs[0] and "GLOO_OK" in outs[1]


def test_checkpoint_dialects_round_trip(tmp_path):
    """Both checkpoint dialects of the reference (legacy ``model_state_dict`` of train/utils.py:109-119 and the
    Lightning ``state_dict`` with ``model.`` keys, demo/utils.py:42-50) load into either module type."""
    import gnn_epc_saft_amd as G
    from gnn_epc_saft_amd.train import checkpoint as C
    deg = torch.tensor([0, 5, 3, 2, 1])
    cfg = dict(hidden_dim=64, num_para=3)
    mk = lambda: G.PNAPCSAFT(64, G.PnaconvsParams(2, 1, 2, deg), G.ReadoutMLPParams(1, 3))
    torch.manual_seed(3)
    src = mk()
    for b in src.buffers():
        if b.dtype == torch.float32:
            b.add_(torch.rand_like(b))
    lit = G.PNApcsaftL(src.pna_params, src.mlp_params, cfg)
    lit.model.load_state_dict(src.state_dict())
    legacy, light = str(tmp_path / "legacy.pt"), str(tmp_path / "epoch=0-step=10.ckpt")
    C.save_checkpoint(C.legacy_checkpoint(src, None, step=10), legacy)
    C.save_checkpoint(C.lightning_checkpoint(lit, None, None, global_step=10), light)
    raw = torch.load(light, weights_only=True)
    assert all(k.startswith("model.") for k in raw["state_dict"]) and raw["global_step"] == 10
    assert set(torch.load(legacy, weights_only=True)) == {"model_state_dict", "optimizer_state_dict",
                                                           "scaler_state_dict", "step"}
    for path in (legacy, light):
        for target in (mk(), G.PNApcsaftL(src.pna_params, src.mlp_params, cfg)):
            ck = C.load_checkpoint(target, path)
            got = (target.model if isinstance(target, G.PNApcsaftL) else target).state_dict()
            for k, v in src.state_dict().items():
                assert torch.equal(got[k], v), k
            assert C.resume(ck) == 10
    # a float64 checkpoint (evaluate_ensemble.py:68 casts the model to double) loads into the float32 module
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in src.state_dict().items()}
    t = mk()
    C.load_checkpoint(t, {"model_state_dict": sd64})
    assert all(torch.equal(t.state_dict()[k], v) for k, v in src.state_dict().items())
    with pytest.raises(ValueError):
        C.load_checkpoint(mk(), {"model_state_dict": {"x": 3}})


def _graph_list(num, seed):
    from gnn_epc_saft_amd.data.synthetic import GraphData, make_synthetic_batch
    b = make_synthetic_batch(num, seed, num_para=3)
    out = []
    for g in range(b.num_graphs):
        n0, n1 = int(b.ptr[g]), int(b.ptr[g + 1])
        em = (b.edge_index[1] >= n0) & (b.edge_index[1] < n1)
        out.append(GraphData(b.x[n0:n1], b.edge_index[:, em] - n0, b.edge_attr[em], para=b.para.view(-1, 3)[g]))
    return out


def test_packed_loader_collates_like_pyg_and_shards_like_distributed_sampler():
    """GraphLoader = PyG DataLoader for the fields the path reads (train.py:74-79): vectorised collation equals the
    per-graph concatenation, epochs reshuffle deterministically, ranks get disjoint strided shards."""
    from gnn_epc_saft_amd.data.loader import GraphLoader, PackedGraphs
    from gnn_epc_saft_amd.data.synthetic import GraphData, collate
    graphs = _graph_list(37, 5)
    graphs.append(GraphData(graphs[0].x[:1], torch.zeros((2, 0), dtype=torch.int64),
                            torch.zeros((0, 3), dtype=torch.int64), para=graphs[0].para))   # 1 node, 0 edges
    pk = PackedGraphs(graphs)
    ids = torch.tensor([5, 37, 0, 36, 7, 7, 12])
    got, want = pk.collate(ids), collate([graphs[i] for i in ids.tolist()])
    for f in ("x", "edge_index", "edge_attr", "batch", "ptr", "para"):
        assert torch.equal(getattr(got, f), getattr(want, f)), f
    assert got.num_graphs == 7
    ld = GraphLoader(pk, 8, shuffle=True, seed=3)
    assert len(ld) == 5
    e1 = [b.para.view(-1, 3) for b in ld]
    e2 = [b.para.view(-1, 3) for b in ld]
    assert sum(p.shape[0] for p in e1) == 38 and not torch.equal(torch.cat(e1), torch.cat(e2))   # reshuffled
    again = GraphLoader(pk, 8, shuffle=True, seed=3)
    assert torch.equal(torch.cat(e1), torch.cat([b.para.view(-1, 3) for b in again]))              # same seed, same epoch
    assert len(GraphLoader(pk, 8, drop_last=True)) == 4
    # two ranks: 19 graphs each, together every graph exactly once (38 is even: no wrap-around padding)
    parts = [torch.cat([b.para.view(-1, 3) for b in GraphLoader(pk, 8, seed=1, rank=r, world_size=2)])
             for r in range(2)]
    assert parts[0].shape[0] == parts[1].shape[0] == 19
    both = torch.cat(parts)
    ref = torch.stack([g.para for g in graphs])
    assert torch.equal(both[both[:, 0].argsort()][:, 0], ref[ref[:, 0].argsort()][:, 0])


def test_hand_counted_waits_of_k_gemm_ar_hold_on_the_emitted_isa():
    """csrc/gemm_ar.hip issues its A loads as inline asm and waits for them with ONE counted s_waitcnt vmcnt per stage:
    nothing but that count stands between a load and the first use of its registers, and the compiler is free to move
    or copy registers around it (a build with the probe's extra branches did exactly that: copies of registers whose
    loads were still in flight, caught on the GPU as a wrong result).  tools/check_ar_isa.py compiles the file for gfx950
    (no GPU needed) and replays prologue + two loop trips against a model of the in-order vector-memory counter: no
    instruction may touch a register a load still owns, no vmcnt(0) inside the loop, every stage issues its LDS-DMA
    pieces before its four loads."""
    import shutil
    import subprocess
    import sys
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if shutil.which(hipcc) is None and not os.path.exists(hipcc):
        pytest.skip("no hipcc in this environment")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_ar_isa.py")], capture_output=True, text=True,
                       timeout=600)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count("clean") == 7   # five plain tile shapes + the two-tower update at hidden 128 / 256
    # ... and the checker has teeth: the probe build (stamps behind two extra branches) is the build on which a GPU
    # parity case failed -- it must be reported
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_ar_isa.py"), "--define", "GS_AR_STAMPS"],
                        capture_output=True, text=True, timeout=600)
    assert r2.returncode != 0 and "while a load owns them" in r2.stdout, r2.stdout[-2000:]
