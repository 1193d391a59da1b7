"""Degree-folded PNAConv update (degree tiles + per-degree weights + K = 5F GEMM) against the
oracle's update stage, and the GEMM tile configurations against an f64 product."""

import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import mini4, rel_err  # noqa: E402
from test_gpu_stages import BOND_DIMS, DEV, K, conv_pieces, synth  # noqa: E402


@pytest.mark.parametrize("case", ["mini4", "synth300"])
@pytest.mark.parametrize("hidden", [64, 128, 256])
def test_degree_tiles_partition_nodes_by_degree(case, hidden):
    d = mini4() if case == "mini4" else synth(300, 17)
    n = d.x.shape[0]
    k = K()
    rowptr, *_ = k.csr_build(d.edge_index.to(DEV), d.edge_attr.to(DEV), n, BOND_DIMS, True)
    perm, tiles, num_tiles, hist3, err = k.degree_tiles(rowptr, hidden)
    assert int(err.item()) == 0
    deg = (rowptr[1:] - rowptr[:-1]).cpu().long()
    buckets = 32
    assert torch.equal(hist3[:buckets].cpu().long(), torch.bincount(deg, minlength=buckets))
    perm = perm.cpu().long()
    assert torch.equal(torch.sort(perm).values, torch.arange(n))          # a permutation
    order = torch.argsort(deg, stable=True)
    assert torch.equal(perm, order)                                       # ascending node id inside a degree
    nt = int(num_tiles.item())
    tiles = tiles.cpu().long()[:nt]
    covered = torch.zeros(n, dtype=torch.bool)
    for dgr, start, count, _ in tiles.tolist():
        assert count >= 1
        rows = perm[start:start + count]
        assert bool((deg[rows] == dgr).all())                            # one degree per tile
        assert not bool(covered[rows].any())
        covered[rows] = True
    assert bool(covered.all())


def test_degree_overflow_is_flagged():
    # a hub with 40 in-edges exceeds the 32 degree buckets
    n = 41
    src = torch.arange(1, n)
    ei = torch.stack([src, torch.zeros(n - 1, dtype=torch.int64)])
    ea = torch.zeros((n - 1, 3), dtype=torch.int64)
    rowptr, *_ = K().csr_build(ei.to(DEV), ea.to(DEV), n, BOND_DIMS, True)
    *_, err = K().degree_tiles(rowptr, 64)
    assert int(err.item()) & 8


@pytest.mark.parametrize("hidden", [64, 128, 256])
@pytest.mark.parametrize("loops", [True, False])
def test_folded_update_matches_oracle(hidden, loops):
    d = synth(40, hidden + 5)
    conv, x, btabs, ei, edge_emb = conv_pieces(hidden, 1, 1, d, loops, seed=4)
    n = x.shape[0]
    stages = {}
    with torch.no_grad():
        conv.double()(x.double(), ei, edge_emb.double(), stages)
        conv.float()
    k = K()
    g = lambda t: t.detach().float().contiguous().to(DEV)
    rowptr, src, dst, combo, la, lt, err = k.csr_build(d.edge_index.to(DEV), d.edge_attr.to(DEV), n, BOND_DIMS, loops)
    perm, tiles, num_tiles, hist3, err = k.degree_tiles(rowptr, hidden)
    agg = g(stages["agg"])
    q0, q1 = conv.post_nns[0], conv.post_nns[1]
    u = k.pna_update_folded(g(x), agg, perm, tiles, num_tiles, hist3, g(conv.aggr_module.avg_deg_log), g(q0[0].weight),
                            g(q0[0].bias), g(q1[0].weight), g(q1[0].bias))
    assert not torch.isnan(u).any()
    assert rel_err(u.cpu(), stages["post"]) < 3e-6
    # and the unfolded (scalers-on-load) kernel agrees with it
    u2 = k.pna_update(g(x), agg, la, lt, g(conv.aggr_module.avg_deg_log), g(q0[0].weight), g(q0[0].bias),
                      g(q1[0].weight), g(q1[0].bias))
    assert rel_err(u2.cpu(), stages["post"]) < 6e-6  # K = 13F f32 chain (3328 terms at H=256)


@pytest.mark.parametrize("hidden", [64, 128, 256])
@pytest.mark.parametrize("loops", [True, False])
def test_destination_term_fold_matches_oracle_conv(hidden, loops):
    """Whole conv from x: source-term GEMM -> K4 over m~ -> update with W_dst folded into W_eff(d),
    against the oracle's PNAConv.  The std threshold may flip on a handful of elements (the variance is
    taken over m~ instead of m = P_i + m~: same value, different rounding), hence the quantile form."""
    d = synth(40, hidden + 9)
    conv, x, btabs, ei, edge_emb = conv_pieces(hidden, 1, 1, d, loops, seed=6)
    n = x.shape[0]
    stages = {}
    with torch.no_grad():
        conv.double()(x.double(), ei, edge_emb.double(), stages)
        conv.float()
    k = K()
    g = lambda t: t.detach().float().contiguous().to(DEV)
    rowptr, src, dst, combo, la, lt, err = k.csr_build(d.edge_index.to(DEV), d.edge_attr.to(DEV), n, BOND_DIMS, loops)
    perm, tiles, num_tiles, hist3, err = k.degree_tiles(rowptr, hidden)
    p0, p1, q0, q1 = conv.pre_nns[0], conv.pre_nns[1], conv.post_nns[0], conv.post_nns[1]
    cemb = k.bond_combo_embed([g(t) for t in btabs])
    rtab = k.pna_edge_table(cemb, g(conv.edge_encoder.weight), g(conv.edge_encoder.bias), g(p0[0].weight),
                            g(p0[0].bias), g(p1[0].weight), g(p1[0].bias))
    q = k.pna_src_terms(g(x), g(p0[0].weight), g(p1[0].weight))
    agg_src = k.pna_aggregate_src(rowptr, src, combo, hidden, q, rtab)
    u = k.pna_update_folded_dst(g(x), agg_src, perm, tiles, num_tiles, hist3, g(conv.aggr_module.avg_deg_log),
                                g(q0[0].weight), g(q0[0].bias), g(q1[0].weight), g(q1[0].bias), g(p0[0].weight),
                                g(p1[0].weight))
    assert not torch.isnan(u).any()
    want = stages["post"]
    err = (u.cpu().double() - want).abs() / float(want.abs().max())
    assert float(torch.quantile(err.flatten()[:: max(1, err.numel() // 500000)], 0.999)) < 6e-6
    assert float(err.max()) < 2e-3
    # std is identical up to flips; mean/min/max differ from the oracle's by the per-node shift P_i
    f = hidden
    std, std64 = agg_src.cpu().double()[..., 3 * f:], stages["agg"][..., 3 * f:]
    assert float(((std - std64).abs() > 1e-5 * float(std64.max())).float().mean()) < 1e-3


@pytest.mark.parametrize("mode", ["split-bf16 (x6)", "f32"])
@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("m,n_out,k", [(1000, 128, 128), (257, 96, 64), (77, 5, 16), (300, 64, 1280), (130, 128, 36)])
def test_every_gemm_tile_configuration(cfg, m, n_out, k, mode):
    """Every tile shape in both arithmetic modes of k_gemm_f32 against the f64 product: the f32 matrix-core form
    (v_mfma_f32_32x32x2_f32: an f32 fma chain) and the split form (every operand = hi + mid + lo in bf16, exactly; six
    v_mfma_f32_32x32x16_bf16 per k16 step, f32 accumulation) under the SAME bar -- and, printed, next to each other:
    the split form must be no further from f64 than 1.5x the f32 chain (measured: closer)."""
    torch.manual_seed(cfg * 7 + m)
    # operands with a wide dynamic range (row scales over three decades) and signs: what the splits must carry exactly
    a = torch.randn(m, k) * torch.logspace(-1.5, 1.5, m).view(-1, 1)
    w, b = torch.randn(n_out, k) / math.sqrt(k), torch.randn(n_out)
    ref = a.double() @ w.double().t() + b.double()
    row_scale = (a.double().abs() @ w.double().abs().t() + b.double().abs())      # sum |a||w|: the error's own scale
    run = lambda c: K().linear(a.to(DEV), w.to(DEV), b.to(DEV), tile_config=c).cpu().double()
    offset = {"split-bf16 (x6)": 16, "f32": 32}[mode]
    out = run(cfg + offset)      # per call: no global tuning state
    other = run(cfg + (32 if mode.startswith("split") else 16))
    err = float(((out - ref).abs() / row_scale).max())
    err_other = float(((other - ref).abs() / row_scale).max())
    print(f"cfg {cfg} [{m},{k}]x[{k},{n_out}] {mode}: max |err| / sum|a||w| = {err:.2e} (the other mode: {err_other:.2e})")
    assert rel_err(out, ref) < 2e-6
    assert err < 4e-7
    if mode.startswith("split"):
        assert err <= 1.5 * err_other + 1e-8


@pytest.mark.parametrize("hidden,graphs,loops", [(128, 96, True), (256, 40, True), (128, 33, False), (256, 300, True)])
def test_fused_aggregate_update_equals_the_two_launches(hidden, graphs, loops):
    """gnnsaft_pna_update_agg (csrc/update_agg.hip: the PNA aggregation done by the producer waves of the update GEMM,
    the aggregates never in HBM) against the two launches it replaces -- gnnsaft_pna_aggregate_src then the folded
    update on its output -- on the same operands, and both against the f64 evaluation of the same algebra.  Same
    per-element reduction arithmetic (sums of m - m_first, PyG's std clamp and mask); the f32 accumulation walks the k
    stages slab-major instead of aggregator-major: equal to rounding (1e-6 of the scale; a std entry within rounding
    of PyG's 1e-5 threshold may mask differently, as between any two evaluations).  Batches with in-degrees up to 13
    (beyond the 4 gathers a lane keeps in flight), 1-node graphs, with and without self-loops."""
    from gnn_epc_saft_amd.data.synthetic import GraphData, collate, make_synthetic_batch
    torch.manual_seed(hidden + graphs)
    base = make_synthetic_batch(graphs, 7 + graphs, num_para=3)
    n0 = base.x.shape[0]
    # a hub with 12 in-edges and two isolated single-node graphs appended
    hub = GraphData(base.x[:14], torch.stack([torch.arange(1, 13), torch.zeros(12, dtype=torch.long)]),
                    base.edge_attr[:12])
    lone = GraphData(base.x[:1], base.edge_index[:, :0], base.edge_attr[:0])
    parts = []
    ptr = base.ptr.tolist()
    for gi in range(graphs):
        lo, hi = ptr[gi], ptr[gi + 1]
        m = (base.edge_index[0] >= lo) & (base.edge_index[0] < hi)
        parts.append(GraphData(base.x[lo:hi], base.edge_index[:, m] - lo, base.edge_attr[m]))
    d = collate(parts + [hub, lone, lone])
    n = d.x.shape[0]
    assert n == n0 + 16
    k = K()
    x = torch.randn(n, hidden)
    rowptr, src, dst, combo, la, lt, err = k.csr_build(d.edge_index.to(DEV), d.edge_attr.to(DEV), n, BOND_DIMS, loops)
    perm, tiles, num_tiles, hist3, err2 = k.degree_tiles(rowptr, hidden)
    q = torch.randn(n, 2 * hidden) * 0.7
    rtab = torch.randn(60, 2 * hidden) * 0.5
    w_post = [torch.randn(hidden // 2, 13 * hidden) / math.sqrt(13 * hidden) for _ in range(2)]
    b_post = [torch.randn(hidden // 2) for _ in range(2)]
    avg = torch.tensor([1.1])
    g = lambda t: t.float().to(DEV)
    agg = k.pna_aggregate_src(rowptr, src, combo, hidden, g(q), g(rtab))
    two = k.pna_update_folded(g(x), agg, perm, tiles, num_tiles, hist3, g(avg), g(w_post[0]), g(b_post[0]), g(w_post[1]),
                              g(b_post[1])).cpu().double()
    one = k.pna_update_agg(g(x), g(q), g(rtab), rowptr, src, combo, perm, tiles, num_tiles, hist3, g(avg), g(w_post[0]),
                           g(b_post[0]), g(w_post[1]), g(b_post[1])).cpu().double()
    assert int(err.item()) == 0 and not torch.isnan(one).any()
    # f64 evaluation of the same algebra from the kernel's own f32 aggregates (the aggregation itself is covered by
    # test_gpu_stages.py::test_message_and_aggregate): cat[x, A, A amp, A att] W^T + b per tower
    deg = (rowptr[1:] - rowptr[:-1]).cpu().double()
    amp = torch.log(deg + 1) / float(avg)
    att = float(avg) / torch.log(deg.clamp(min=1) + 1)
    a64 = agg.cpu().double()
    want = torch.empty(n, hidden, dtype=torch.float64)
    for t in range(2):
        z = torch.cat([x.double(), a64[:, t], a64[:, t] * amp[:, None], a64[:, t] * att[:, None]], 1)
        want[:, t * hidden // 2:(t + 1) * hidden // 2] = z @ w_post[t].double().t() + b_post[t].double()
    scale = float(want.abs().max())
    e_one, e_two = float((one - want).abs().max()) / scale, float((two - want).abs().max()) / scale
    diff = (one - two).abs() / scale
    print(f"H={hidden} {n} nodes, max in-degree {int(deg.max())}: fused vs f64 {e_one:.2e}, two launches vs f64 {e_two:.2e}, "
          f"fused vs two launches max {float(diff.max()):.2e}, rows differing by > 1e-6: {int((diff.max(1).values > 1e-6).sum())}")
    assert e_one <= max(2 * e_two, 2e-6)
    # rows whose std mask flipped between the two evaluations of the SAME f32 arithmetic should not exist at all:
    # the reductions are the same instruction sequence on the same operands
    assert float(diff.max()) <= 2e-6


@pytest.mark.parametrize("hidden,graphs,loops", [(128, 96, True), (256, 40, True), (128, 33, False), (256, 300, True), (128, 1024, True)])
def test_ar_update_equals_the_folded_update(hidden, graphs, loops):
    """gnnsaft_pna_update_folded_ar (csrc/gemm_ar.hip: the degree-folded update with BOTH towers in one workgroup per
    degree tile, the A operand [x | A_t] in registers, what gnnsaft_forward launches below 64 k nodes) against
    gnnsaft_pna_update_folded (k_gemm_f32<PostFoldA, X6>) on the same operands and both against the f64 evaluation of
    cat[x, A, A amp, A att] W^T + b: same k order inside a 32-k stage and the same six products -- equal to the order of
    the k16 steps.  Partial degree tiles, a hub of 12 in-edges, isolated nodes."""
    from gnn_epc_saft_amd.data.synthetic import GraphData, collate, make_synthetic_batch
    torch.manual_seed(hidden + graphs)
    base = make_synthetic_batch(graphs, 7 + graphs, num_para=3)
    hub = GraphData(base.x[:14], torch.stack([torch.arange(1, 13), torch.zeros(12, dtype=torch.long)]), base.edge_attr[:12])
    lone = GraphData(base.x[:1], base.edge_index[:, :0], base.edge_attr[:0])
    parts = []
    ptr = base.ptr.tolist()
    for gi in range(graphs):
        lo, hi = ptr[gi], ptr[gi + 1]
        m = (base.edge_index[0] >= lo) & (base.edge_index[0] < hi)
        parts.append(GraphData(base.x[lo:hi], base.edge_index[:, m] - lo, base.edge_attr[m]))
    d = collate(parts + [hub, lone, lone])
    n = d.x.shape[0]
    k = K()
    x = torch.randn(n, hidden)
    rowptr, src, dst, combo, la, lt, err = k.csr_build(d.edge_index.to(DEV), d.edge_attr.to(DEV), n, BOND_DIMS, loops)
    perm, tiles, num_tiles, hist3, err2 = k.degree_tiles(rowptr, hidden)
    agg = torch.randn(n, 2, 4 * hidden) * 0.7
    w_post = [torch.randn(hidden // 2, 13 * hidden) / math.sqrt(13 * hidden) for _ in range(2)]
    b_post = [torch.randn(hidden // 2) for _ in range(2)]
    avg = torch.tensor([1.1])
    g = lambda t: t.float().to(DEV)
    two = k.pna_update_folded(g(x), g(agg), perm, tiles, num_tiles, hist3, g(avg), g(w_post[0]), g(b_post[0]), g(w_post[1]),
                              g(b_post[1])).cpu().double()
    one = k.pna_update_folded_ar(g(x), g(agg), perm, tiles, num_tiles, hist3, g(avg), g(w_post[0]), g(b_post[0]),
                                 g(w_post[1]), g(b_post[1])).cpu().double()
    assert int(err.item()) == 0 and not torch.isnan(one).any()
    deg = (rowptr[1:] - rowptr[:-1]).cpu().double()
    amp = torch.log(deg + 1) / float(avg)
    att = float(avg) / torch.log(deg.clamp(min=1) + 1)
    want = torch.empty(n, hidden, dtype=torch.float64)
    for t in range(2):
        z = torch.cat([x.double(), agg[:, t].double(), agg[:, t].double() * amp[:, None], agg[:, t].double() * att[:, None]], 1)
        want[:, t * hidden // 2:(t + 1) * hidden // 2] = z @ w_post[t].double().t() + b_post[t].double()
    scale = float(want.abs().max())
    e_one, e_two = float((one - want).abs().max()) / scale, float((two - want).abs().max()) / scale
    print(f"H={hidden} {n} nodes: AR update vs f64 {e_one:.2e}, X6 update vs f64 {e_two:.2e}, AR vs X6 max "
          f"{float((one - two).abs().max()) / scale:.2e}")
    assert e_one <= max(2 * e_two, 2e-6)
    assert float((one - two).abs().max()) <= 2e-6 * scale


AR_TILES = ["128x128", "128x64", "64x128", "64x64", "256x128"]


@pytest.mark.parametrize("cfg", range(len(AR_TILES)))
@pytest.mark.parametrize("m,n_out,k", [(1000, 128, 128), (257, 96, 64), (300, 64, 1280), (130, 256, 32), (4099, 512, 256)])
def test_every_ar_gemm_tile_configuration(cfg, m, n_out, k):
    """k_gemm_ar (csrc/gemm_ar.hip: the A operand loaded and split by the lane that feeds it to the matrix core, hand-
    issued loads with counted waits, weights through an LDS ring of image stages) in every tile shape against the f64
    product under the bar of test_every_gemm_tile_configuration (4e-7 of sum |a||w|) and against k_gemm_w3 on the
    same image: same k order, same six products -- equal up to the order of the k16 steps.  K of 1, 2, 8 and 40
    stages: the three-stage loop body, its one- and two-stage tails and the requests past the end."""
    torch.manual_seed(cfg * 7 + m)
    a = torch.randn(m, k) * torch.logspace(-1.5, 1.5, m).view(-1, 1)
    w, b = torch.randn(n_out, k) / math.sqrt(k), torch.randn(n_out)
    ref = a.double() @ w.double().t() + b.double()
    row_scale = (a.double().abs() @ w.double().abs().t() + b.double().abs())
    ad, wd, bd = a.to(DEV), w.to(DEV), b.to(DEV)
    img = K().w3_pack(wd)
    out = K().linear_ar(ad, img, n_out, bd, cfg).cpu().double()
    w3 = K().linear_w3(ad, img, n_out, bd, 0).cpu().double()
    f32 = K().linear(ad, wd, bd, tile_config=3 + 32).cpu().double()
    err, err32 = (float(((o - ref).abs() / row_scale).max()) for o in (out, f32))
    print(f"ar {AR_TILES[cfg]} [{m},{k}]x[{k},{n_out}]: max |err| / sum|a||w| = {err:.2e} (f32 chain {err32:.2e})")
    assert not torch.isnan(out).any()
    assert rel_err(out, ref) < 2e-6
    assert err < 4e-7 and err <= 1.5 * err32 + 1e-8
    assert float((out - w3).abs().max()) <= 4e-7 * float(row_scale.max())


@pytest.mark.parametrize("kind", ["x6 in-kernel split", "f32 matrix cores", "w3 images"])
def test_non_finite_operands_poison_only_their_own_rows(kind):
    """what a +-inf / NaN operand does in the GEMMs, pinned.  Every arithmetic mode returns non-finite
    values in exactly the rows that hold a non-finite A element (and nowhere else); the split-bf16 modes return NaN
    there (inf = hi, mid = inf - inf = NaN: csrc/x6.hpp), the f32 fma chain +-inf or NaN."""
    torch.manual_seed(3)
    m, k, n_out = 300, 128, 128
    a = torch.randn(m, k)
    a[7, 5] = float("inf")
    a[100, 64] = float("-inf")
    a[201, 127] = float("nan")
    w, b = torch.randn(n_out, k) / math.sqrt(k), torch.randn(n_out)
    ad, wd, bd = a.to(DEV), w.to(DEV), b.to(DEV)
    if kind.startswith("w3"):
        out = K().linear_w3(ad, K().w3_pack(wd), n_out, bd, 0).cpu()
    else:
        out = K().linear(ad, wd, bd, tile_config=3 + (16 if kind.startswith("x6") else 32)).cpu()
    bad = ~torch.isfinite(out).all(dim=1)
    assert sorted(torch.nonzero(bad).flatten().tolist()) == [7, 100, 201]
    assert not torch.isfinite(out[[7, 100, 201]]).any()
    if not kind.startswith("f32"):
        assert torch.isnan(out[[7, 100, 201]]).all()
    clean = torch.ones(m, dtype=torch.bool)
    clean[[7, 100, 201]] = False
    ref = a[clean].double() @ w.double().t() + b.double()
    assert rel_err(out[clean].double(), ref) < 2e-6


W3_TILES = ["128x128", "128x256", "64x128", "64x64", "128x64", "64x256", "128x128 double-buffered", "128x256 double-buffered",
            "wave-specialised 128x128", "wave-specialised 64x128"]


@pytest.mark.parametrize("cfg", range(len(W3_TILES)))
@pytest.mark.parametrize("m,n_out,k", [(1000, 128, 128), (257, 96, 64), (300, 64, 1280), (130, 256, 32), (4099, 512, 256)])
def test_every_w3_gemm_tile_configuration(cfg, m, n_out, k):
    """k_gemm_w3 (pre-split weight images, direct-to-LDS weight copies, 32-k stages) in every tile shape against the
    f64 product under the bar of test_every_gemm_tile_configuration (4e-7 of sum |a||w|), next to the f32 fma chain of
    the f32 matrix cores and to k_gemm_f32<X6>; with BatchNorm partials (STATS) against the f64 column statistics."""
    torch.manual_seed(cfg * 7 + m)
    a = torch.randn(m, k) * torch.logspace(-1.5, 1.5, m).view(-1, 1)
    w, b = torch.randn(n_out, k) / math.sqrt(k), torch.randn(n_out)
    ref = a.double() @ w.double().t() + b.double()
    row_scale = (a.double().abs() @ w.double().abs().t() + b.double().abs())
    ad, wd, bd = a.to(DEV), w.to(DEV), b.to(DEV)
    img = K().w3_pack(wd)
    spec = cfg >= 8      # gemm_w3s.hip: consumer / producer waves
    out, stats = K().linear_w3(ad, img, n_out, bd, cfg - 8 if spec else cfg, want_stats=True, specialised=spec)
    out = out.cpu().double()
    f32 = K().linear(ad, wd, bd, tile_config=3 + 32).cpu().double()
    x6 = K().linear(ad, wd, bd, tile_config=3 + 16).cpu().double()
    err, err32, err6 = (float(((o - ref).abs() / row_scale).max()) for o in (out, f32, x6))
    print(f"w3 {W3_TILES[cfg]} [{m},{k}]x[{k},{n_out}]: max |err| / sum|a||w| = {err:.2e} (f32 chain {err32:.2e}, "
          f"in-kernel split {err6:.2e})")
    assert not torch.isnan(out).any()
    assert rel_err(out, ref) < 2e-6
    assert err < 4e-7 and err <= 1.5 * err32 + 1e-8
    # the same k order and the same six products as the in-kernel split: equal up to the order of the k16 steps
    assert float((out - x6).abs().max()) <= 4e-7 * float(row_scale.max())
    # BatchNorm partials: (mean, M2) per 64-row group and column
    rpg = K().bn_rows_per_group()
    groups = (m + rpg - 1) // rpg
    st = stats.cpu().double()
    for gi in (0, groups - 1):
        rows = ref[gi * rpg:(gi + 1) * rpg]
        mean, m2 = rows.mean(0), ((rows - rows.mean(0)) ** 2).sum(0)
        assert float((st[gi, 0] - mean).abs().max()) < 2e-6 * float(row_scale.max())
        assert float((st[gi, 1] - m2).abs().max()) < 2e-5 * float(m2.abs().max() + 1e-30)


@pytest.mark.parametrize("rows,ch", [(1000, 128), (63, 64), (20480, 256), (2, 32), (777, 16)])
def test_fused_batchnorm_train_apply(rows, ch):
    torch.manual_seed(rows + ch)
    a = torch.randn(rows, ch) * 2 + 3
    bn = torch.nn.BatchNorm1d(ch).double().train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_()
        bn.running_mean.normal_()
        bn.running_var.uniform_(0.5, 2)
    res = torch.randn(rows, ch)
    g = lambda t: t.detach().float().to(DEV)
    rm, rv, nbt = g(bn.running_mean), g(bn.running_var), torch.tensor(5, device=DEV)
    y, stats = K().linear(a.to(DEV), torch.eye(ch).to(DEV), None, want_stats=True)
    out = K().bn_train_apply(stats, y, g(bn.weight), g(bn.bias), rm, rv, nbt, 0.1, 1e-5, res.to(DEV)).cpu()
    ref = torch.relu(bn(a.double())) + res.double()
    assert rel_err(out, ref) < 2e-6
    assert rel_err(rm.cpu(), bn.running_mean) < 1e-6 and rel_err(rv.cpu(), bn.running_var) < 1e-6
    assert int(nbt.item()) == 6
    out = K().bn_train_apply(stats, y, g(bn.weight), g(bn.bias), None, None, None, 0.1, 1e-5, None).cpu()
    assert rel_err(out, torch.relu(bn(a.double()))) < 2e-6


def test_forward_folded_equals_unfolded():
    import copy

    import gnn_epc_saft_amd as G
    from gnn_epc_saft_amd.data.synthetic import degree_histogram, make_synthetic_batch
    data = make_synthetic_batch(64, 5)
    torch.manual_seed(0)
    m = G.PNAPCSAFT(128, G.PnaconvsParams(3, 1, 2, degree_histogram(data), skip_connections=True, self_loops=True),
                    G.ReadoutMLPParams(1, 3)).to(DEV).eval()
    with torch.no_grad():
        m.fold_degree_scalers = True
        a = m(data.to(DEV))
        m.fold_degree_scalers = False
        b = m(data.to(DEV))
    assert m.input_error_flags() == 0
    assert rel_err(a, b) < 2e-6
