#!/usr/bin/env python3
"""Headline benchmark: molecular graphs/s for PNAPCSAFT forward + MAPE loss (train-mode
BatchNorm, no backward) on synthetic molecular graphs, one process per MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path (gnnsaft_forward: CSR build, embeddings, L PNA layers,
add-pool, readout MLP, MAPE) over one HBM-resident batch, plus -- for N > 1 -- the RCCL
all-reduce of [sum(ape), count] that the reference performs for its `sync_dist=True` loss
metric.  Weak scaling: every rank owns its own G graphs; there is no data-path collective.
Rank 0 prints ONE JSON line.

`--config 2` (default) is BASELINE.json configs[1], the configuration the metric is quoted on; the N = 1 line
also carries a `c3` block (configs[2], beyond the 256 MiB Infinity Cache: the HBM-honest roofline figures), and every
N > 1 line a `c4` block: BASELINE.json configs[3] -- 8192 graphs PER RANK on the H=256 / L=5 model, forward + loss with
the loss all-reduce, and a full training step with the 28 MB flat gradient all-reduce timed on its own (single
collective and segment-overlapped).  Every line carries `input_error_flags` (GNNSAFT_FLAG_* bits raised by any of its
forwards, incl. BARRIER_TIMEOUT); a non-zero value fails the run.
`--config 3` runs configs[2] as the headline; `--config 5` is the C5 stand-in of SURVEY.md 8(d) (2 000 synthetic
graphs, batch 512, default model) where a step is a full TRAINING step.
"""

from __future__ import annotations

import argparse
import csv
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # BASELINE.json configs[1] (the configuration the metric is quoted on) and configs[2]
    2: dict(graphs=1024, hidden=128, depth=3, num_para=3, name="C2"),
    3: dict(graphs=8192, hidden=256, depth=5, num_para=3, name="C3"),
    # configs[4] stand-in: default model of configs/default.py:35-45, batch 512 of a 2 000-graph set
    5: dict(graphs=512, hidden=64, depth=6, num_para=5, name="C5", dataset=2000),
}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3   # MI355X_MICROARCH.md: dense f32 MFMA
MFMA_BF16_PEAK_TF = 2500.0 # MI355X_MICROARCH.md: dense bf16 MFMA (the pipe the split-bf16 GEMMs run on)
GEMM_X6 = os.environ.get("GNNSAFT_GEMM_X6", "1") != "0"   # csrc/gemm.hip: six bf16 MFMAs per f32-equivalent product
K4_KERNEL = "k_pna_aggregate<2"    # (rocprofv3 prints k_pna_aggregate<2, false> / <2, true>) the kernel gnnsaft_forward launches for pre_layers == 1 (kFusedQ source)
PROFILE_TAG = "r04"        # profiles/<tag>_* files are the rocprofv3 evidence of THIS round's kernels


def k4_algorithmic_bytes(n: int, e_prime: int, hidden: int) -> int:
    """each [T*F] message row read once, int64 destination ids as
    delivered, the four aggregates written once."""
    return 8 * hidden * (e_prime + 4 * n) + 8 * e_prime


def gemm_reference_flops(n: int, e_prime: int, hidden: int) -> float:
    """reference formulation, per layer, pre = post = 1."""
    return (14.0 * e_prime + 28.0 * n) * hidden * hidden


def cpu_baseline(cfg, data, deg, budget_s: float):
    """The oracle (CPU restatement, PyG-equivalent op sequence) timed on the host cores: a quick probe picks the
    best of the thread counts {16, 32, 64, 128} that the host has, then 3 warm-ups and >= 10 timed passes there."""
    from oracle.pna_torch import OracleMlpParams, OraclePNAPCSAFT, OraclePnaParams, training_loss
    torch.manual_seed(0)
    p = cfg["num_para"]
    model = OraclePNAPCSAFT(cfg["hidden"], OraclePnaParams(cfg["depth"], 1, 1, deg, skip_connections=True,
                                                           self_loops=True), OracleMlpParams(1, p)).train()
    host = os.cpu_count() or 1
    before = torch.get_num_threads()
    candidates = sorted({t for t in (16, 32, 64, 128) if t <= host} or {host})
    probe = {}
    try:
        with torch.no_grad():
            for t in candidates:
                torch.set_num_threads(t)
                training_loss(model, data, p)
                best = float("inf")
                for _ in range(2):
                    t0 = time.perf_counter()
                    training_loss(model, data, p)
                    best = min(best, time.perf_counter() - t0)
                probe[t] = best
            threads = min(probe, key=probe.get)
            torch.set_num_threads(threads)
            for _ in range(3):
                training_loss(model, data, p)
            times = []
            t_end = time.perf_counter() + budget_s
            while len(times) < 10 or (time.perf_counter() < t_end and len(times) < 50):
                t0 = time.perf_counter()
                training_loss(model, data, p)
                times.append(time.perf_counter() - t0)
    finally:
        torch.set_num_threads(before)
    times.sort()
    med = times[len(times) // 2]
    return {"value": data.num_graphs / med, "unit": "graphs/s", "cores": threads, "kind": "port",
            "sample": f"{len(times)} timed forward+loss passes (median; min {data.num_graphs / times[-1]:.0f}, max "
                      f"{data.num_graphs / times[0]:.0f} graphs/s) after 3 warm-ups of the oracle (CPU restatement, "
                      f"PyG-equivalent op sequence, torch {torch.__version__}) over the same {data.num_graphs}-graph "
                      f"batch, {threads} threads = best of a 2-pass probe over {candidates} on {host} host CPUs "
                      f"(probe s/pass: {', '.join(f'{t}: {v:.2f}' for t, v in probe.items())})"}


def profile_file(name: str):
    path = os.path.join(ROOT, "profiles", name)
    return path if os.path.exists(path) else None


def committed_k4_evidence(cfg_name: str):
    """(traffic bytes per launch or None, note, rocprofv3 avg launch ms or None) from the files under profiles/.
    A traffic file recorded for another kernel than the one timed here is refused."""
    traffic, note, rocprof_ms = None, None, None
    tpath = profile_file(f"{PROFILE_TAG}_k4_hbm_traffic_{cfg_name}.json")
    if tpath is None:
        note = f"no profiles/{PROFILE_TAG}_k4_hbm_traffic_{cfg_name}.json"
    else:
        with open(tpath) as fh:
            rec = json.load(fh)
        if K4_KERNEL in str(rec.get("kernel", "")):
            traffic = rec.get("hbm_bytes_per_launch")
            note = f"profiles/{os.path.basename(tpath)} ({rec.get('kernel')}, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)"
        else:
            note = f"refused {os.path.basename(tpath)}: recorded for {rec.get('kernel')!r}, timed kernel is {K4_KERNEL}"
    spath = profile_file(f"{PROFILE_TAG}_{cfg_name.lower()}_kernel_stats.csv")
    if spath is not None:  # rocprofv3 --kernel-trace --stats of this command, committed under profiles/
        with open(spath) as fh:
            for row in csv.DictReader(fh):
                if K4_KERNEL in row["Name"]:
                    rocprof_ms = float(row["AverageNs"]) * 1e-6
    return traffic, note, rocprof_ms


class Workload:
    """Model + device-resident batch of one configuration."""

    def __init__(self, cfg, dev, rank: int, config_id: int):
        import gnn_epc_saft_amd as G
        from gnn_epc_saft_amd.data.synthetic import degree_histogram, make_synthetic_batch
        self.cfg, self.dev = cfg, dev
        p = cfg["num_para"]
        # synthetic workload: every rank its own G graphs (weak scaling), same model everywhere
        self.data = make_synthetic_batch(cfg["graphs"], 1234 + config_id + 1000 * rank, num_para=p)
        self.deg = degree_histogram(make_synthetic_batch(cfg["graphs"], 1234 + config_id, num_para=p))
        torch.manual_seed(0)
        self.model = G.PNApcsaftL(
            G.PnaconvsParams(cfg["depth"], 1, 1, self.deg, skip_connections=True, self_loops=True),
            G.ReadoutMLPParams(1, p),
            dict(hidden_dim=cfg["hidden"], num_para=p, optimizer="adam", learning_rate=1e-3, weight_decay=1e-2,
                 warmup_steps=100, momentum=0.9)).to(dev).train()
        self.ddev = self.data.to(dev)
        self.n, self.e = int(self.data.x.shape[0]), int(self.data.edge_index.shape[1])
        self.e_prime = self.e + self.n

    def describe(self):
        c = self.cfg
        return (f"{c['name']}: {c['graphs']} synthetic molecular graphs per GPU (|V|~U[12,28], |E|~2|V|, 9 int64 node / "
                f"3 int64 edge categorical features), PNAPCSAFT H={c['hidden']} L={c['depth']} pre=post=1 mlp=1 "
                f"P={c['num_para']} skip+self-loops")


def instrumented_kernel_times(wl: Workload, steps: int, stream, barrier, fused: bool = False):
    """K eager steps with HIP events recorded ON THE LAUNCH STREAM around every K4 / GEMM launch (gnnsaft_profile_*):
    {kernel: (launches, avg ms)} and the wall time of the instrumented repeat.  Asking for the aggregation kernel's
    events makes the forward launch it on its own (k_pna_aggregate + the update GEMM: the two launches the taped
    training forward always issues); ``fused``: no aggregation events -- the no-grad forward's own schedule, in which
    the update kernel aggregates inside its operand path (k_update_agg_w3s) and `update` times THAT launch."""
    from gnn_epc_saft_amd import _native
    mask = _native.PROF_UPDATE | _native.PROF_NODE_TERMS | _native.PROF_LIN | _native.PROF_UPDATE_AGG
    if not fused:
        mask |= _native.PROF_AGGREGATE
    handle = ctypes.c_void_p()
    _native.check(_native.lib.gnnsaft_profile_create(steps * wl.cfg["depth"] * 4, mask, ctypes.byref(handle)),
                  "gnnsaft_profile_create")
    wl.model.model._profile = handle
    try:
        with torch.no_grad(), torch.cuda.stream(stream):
            barrier()
            t1 = time.perf_counter()
            for _ in range(steps):
                wl.model.training_step_parts(wl.ddev)
            barrier()
            elapsed = time.perf_counter() - t1
    finally:
        wl.model.model._profile = None

    def kernel_ms(bit):
        cnt, tot = ctypes.c_int32(), ctypes.c_float()
        _native.check(_native.lib.gnnsaft_profile_summary(handle, bit, ctypes.byref(cnt), ctypes.byref(tot)),
                      "gnnsaft_profile_summary")
        return cnt.value, (tot.value / cnt.value if cnt.value else float("nan"))

    out = {"k4": kernel_ms(_native.PROF_AGGREGATE), "update": kernel_ms(_native.PROF_UPDATE),
           "node_terms": kernel_ms(_native.PROF_NODE_TERMS), "lin": kernel_ms(_native.PROF_LIN),
           "update_agg": kernel_ms(_native.PROF_UPDATE_AGG)}
    _native.lib.gnnsaft_profile_destroy(handle)
    return out, elapsed


def roofline_blocks(wl: Workload, times, event_overhead_ms=None, fused_times=None):
    """`roofline` (K4, HBM) and `roofline_gemm` (f32 MFMA) objects from the event-timed launches."""
    cfg, n, ep, h = wl.cfg, wl.n, wl.e_prime, wl.cfg["hidden"]
    k4_n, k4_ms = times["k4"]
    k4_bytes = k4_algorithmic_bytes(n, ep, h)
    k4_gbs = k4_bytes / (k4_ms * 1e-3) / 1e9
    traffic, traffic_note, rocprof_ms = committed_k4_evidence(cfg["name"])
    ws_mb = (8 * h * n + 32 * h * n) / 1e6   # q [N,2H] gathered + agg [N,2,4H] written, f32
    roof = {
        "kernel": f"{K4_KERNEL}, STREAM_OUT> (K4 segmented mean|min|max|std, messages gathered from q[src] + rtab[class]; "
                  "streaming stores when the aggregates exceed the Infinity Cache)",
        "bound": "hbm", "achieved": k4_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": k4_gbs / HBM_PEAK_GBS,
        "traffic": traffic, "traffic_source": traffic_note,
        "algorithmic_bytes_per_launch": k4_bytes, "avg_launch_ms": k4_ms, "launches_timed": k4_n,
        "achieved_physical": (traffic / (k4_ms * 1e-3) / 1e9) if traffic else None,   # PMC bytes / event time
        "frac_physical": (traffic / (k4_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
        "rocprofv3_avg_launch_ms": rocprof_ms,   # from profiles/ (kernel time without the event overhead)
        "event_pair_ms_around_empty_kernel": event_overhead_ms,
        "working_set_mb": ws_mb,
        "launched_by": ("the taped (training) forward and this instrumented repeat of the timed steps; the timed no-grad "
                        "forward itself fuses the aggregation into the update GEMM's operand path (k_update_agg_w3s, "
                        "`roofline_gemm.fused_update_agg`: from 64 k nodes up and hidden % 128 == 0), so that the "
                        "aggregates never reach HBM"
                        if (fused_times is not None and fused_times["update_agg"][0] > 0) else
                        "every forward of this workload (timed no-grad steps, taped training forward, this instrumented "
                        "repeat): below 64 k nodes the two launches are faster than the fused aggregation + update"),
        "how": "HIP events on the launch stream around every K4 launch of an instrumented repeat of the timed steps; "
               "`achieved` = SURVEY 8(d) algorithmic bytes / event time, `achieved_physical` = the PMC-counted HBM "
               "bytes of `traffic` / the same time.  " +
               (f"Working set {ws_mb:.0f} MB < 256 MiB Infinity Cache and the same batch is replayed: part of this "
                "rate is L3, not HBM -- the `c3` block is the beyond-L3 figure."
                if ws_mb < 256 else f"Working set {ws_mb:.0f} MB > 256 MiB Infinity Cache: HBM-bound in earnest."),
    }
    executed = {"node_terms": 4.0, "update": 10.0, "lin": 2.0}     # N H^2 FLOP units per layer and kernel
    per_kernel, tot_ms, tot_flop = {}, 0.0, 0.0
    for k, units in executed.items():
        ms = times[k][1]
        flop = units * n * h * h
        per_kernel[k] = {"avg_ms": ms, "executed_tflops": flop / (ms * 1e-3) / 1e12,
                         "frac_of_f32_mfma_peak": flop / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TF}
        tot_ms += ms
        tot_flop += flop
    ex_tf = tot_flop / (tot_ms * 1e-3) / 1e12
    if GEMM_X6:
        # every f32 operand is split exactly into three bf16 numbers while it is staged; six of the nine cross products
        # are issued as v_mfma_f32_32x32x16_bf16 with f32 accumulation: 6 bf16-MFMA FLOPs per f32-equivalent FLOP
        for v in per_kernel.values():
            v["f32_equivalent_vs_f32_mfma_peak"] = v.pop("frac_of_f32_mfma_peak")   # a speed ratio, not a roofline fraction
            v["bf16_mfma_tflops_issued"] = 6.0 * v["executed_tflops"]
            v["frac_of_bf16_mfma_peak"] = 6.0 * v["executed_tflops"] / MFMA_BF16_PEAK_TF
        gemm = {
            "kernels": "k_gemm_f32<..., X6>: source terms (PlainA), degree-folded update (PostFoldA), lin (+BN partials); "
                       "per layer; split-bf16 arithmetic at f32 accuracy (hi + mid + lo, six bf16 MFMAs per product)",
            "bound": "mfma", "achieved": 6.0 * ex_tf, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
            "frac": 6.0 * ex_tf / MFMA_BF16_PEAK_TF,
            "flops_basis": "bf16 MFMA FLOPs ISSUED = 6 x the executed f32-equivalent FLOPs (16 N H^2 per layer: source "
                           "terms 4, degree-folded update 10, lin 2), against the dense bf16 MFMA peak",
            "f32_equivalent_tflops": ex_tf, "f32_equivalent_vs_f32_mfma_peak": ex_tf / MFMA_F32_PEAK_TF,
            "per_kernel": per_kernel,
            "speedup_vs_reference_formulation": gemm_reference_flops(n, ep, h) / tot_flop,
            "reference_formulation_note": "SURVEY 8(d) counts (14E'+28N)H^2 per layer for the reference's edge-level "
                                          "GEMMs; the restructured path issues 16 N H^2 f32-equivalent -- the ratio is an "
                                          "algorithmic saving, not a fraction of peak",
        }
        if fused_times is not None and fused_times["update_agg"][0] > 0:
            ms = fused_times["update_agg"][1]
            two = times["k4"][1] + times["update"][1]
            flop = 10.0 * n * h * h
            gemm["fused_update_agg"] = {
                "kernel": "k_update_agg_w3s: aggregation (k_pna_aggregate's reduction, by the producer waves) + "
                          "degree-folded update GEMM in one launch -- what the timed no-grad forward runs per layer",
                "avg_ms": ms, "launches_timed": fused_times["update_agg"][0],
                "replaces_ms": two, "replaces": "k_pna_aggregate + update GEMM launched separately (above)",
                "bf16_mfma_tflops_issued": 6.0 * flop / (ms * 1e-3) / 1e12,
                "frac_of_bf16_mfma_peak": 6.0 * flop / (ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TF,
                "hbm_bytes_not_moved": 2 * 32 * h * n,   # the aggregates [N,2,4H] f32: written once, read once
            }
        return roof, gemm
    gemm = {
        "kernels": "k_gemm_f32: source terms (PlainA), degree-folded update (PostFoldA), lin (+BN partials); per layer",
        "bound": "mfma", "achieved": ex_tf, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
        "frac": ex_tf / MFMA_F32_PEAK_TF,
        "flops_basis": "EXECUTED f32 MFMA FLOPs: 16 N H^2 per layer (source terms 4, degree-folded update 10, lin 2)",
        "per_kernel": per_kernel,
        "speedup_vs_reference_formulation": gemm_reference_flops(n, ep, h) / tot_flop,
        "reference_formulation_note": "SURVEY 8(d) counts (14E'+28N)H^2 per layer for the reference's edge-level GEMMs; "
                                      "the restructured path issues 16 N H^2 -- the ratio is an algorithmic saving, not a "
                                      "fraction of peak",
    }
    return roof, gemm


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--graph", type=int, default=1, help="replay the step from a captured hipGraph (1) or launch eagerly (0)")
    ap.add_argument("--train-steps", type=int, default=20, help="extra forward+backward steps timed (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-c3", action="store_true", help="skip the secondary C3 roofline block of the N = 1 line")
    ap.add_argument("--no-c4", action="store_true", help="skip the C4 block (8192 graphs per rank, H=256 L=5) of an N > 1 line")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    args = ap.parse_args()

    from gnn_epc_saft_amd import parallel

    # GNNSAFT_BENCH_REHEARSAL=1: multi-rank rehearsal on a ONE-GPU box (gloo backend, every rank on cuda:0);
    # exercises the N > 1 control flow only, its numbers mean nothing.
    rehearsal = os.environ.get("GNNSAFT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        # the ranks SHARE one GPU: kernels whose workgroups wait for each other (the cooperative structure chain of
        # the forward's first launch) lose whole scheduling quanta there (measured 4x on the c4 forward); one process
        # per GPU -- the deployment, and every real N > 1 run -- is unaffected
        os.environ.setdefault("GNNSAFT_K0_FUSED", "0")
    rank, local_rank, world = parallel.init_from_env("gloo" if rehearsal else "nccl")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    dev = torch.device("cuda", 0 if rehearsal else local_rank)
    torch.cuda.set_device(dev)
    if args.config == 5:
        return bench_training_loop(args, dev, rank, world)
    cfg = CONFIGS[args.config]
    wl = Workload(cfg, dev, rank, args.config)
    model, ddev = wl.model, wl.ddev

    import torch.distributed as dist

    pending = []

    def exchange(parts):
        if world == 1:
            return parallel.global_mape(parts)
        # N > 1: RCCL all-reduce(sum) of [sum(ape), count], asynchronous (a logged metric, as sync_dist=True);
        # every handle is waited for before the closing barrier of the timed region
        pending.append(parallel.global_mape_async(parts))
        return pending[-1]

    def step():
        return exchange(model.training_step_parts(ddev))       # [mape, sum(ape), count] on device

    def drain():
        if not pending:
            return None
        for h in pending[:-1]:
            if h.work is not None:
                h.work.wait()
        out = pending[-1].result()
        pending.clear()
        return out

    def barrier():
        last = drain()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        return last

    stream = torch.cuda.Stream(dev)
    # The captured graph holds the forward + loss kernels only; for N > 1 the (tiny) loss exchange is issued
    # right behind every replay, outside the graph (RCCL inside a captured graph is not exercised here).
    use_graph = bool(args.graph)
    graph, parts_static, graph_note = None, None, None
    with torch.no_grad():
        with torch.cuda.stream(stream):
            for _ in range(max(args.warmup, 1) if use_graph else args.warmup):
                loss = step()
            barrier()
        if use_graph:
            # captured on its own stream, thread-local error mode: a HIP call of another thread (e.g. the process
            # group's watchdog) must not invalidate it; if the capture is refused anyway, the timed steps fall back
            # to eager launches on the untouched `stream` and the JSON line says so
            try:
                cap_stream = torch.cuda.Stream(dev)
                cap_stream.wait_stream(stream)
                g_obj = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_obj, stream=cap_stream, capture_error_mode="thread_local"):
                    parts_static = model.training_step_parts(ddev)
                stream.wait_stream(cap_stream)
                with torch.cuda.stream(stream):
                    g_obj.replay()
                torch.cuda.synchronize(dev)
                graph = g_obj
            except Exception as exc:  # noqa: BLE001
                graph, graph_note = None, f"capture failed: {type(exc).__name__}: {exc}"[:200]
                try:
                    torch.cuda.synchronize(dev)
                except Exception:  # noqa: BLE001
                    pass
    with torch.no_grad(), torch.cuda.stream(stream):

        def replay_step():
            graph.replay()
            return exchange(parts_static)

        run = replay_step if graph is not None else step

        # ---- timed region 1: K steps, nothing else on the stream -> `value`
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = run()
        last = barrier()
        elapsed = time.perf_counter() - t0
        final_loss = last if last is not None else loss

        # ---- the same K steps launched eagerly (one C call per step), for the record
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step()
        barrier()
        elapsed_eager = time.perf_counter() - t0

    # ---- timed region 2: the same K steps launched eagerly with HIP events recorded on the
    # launch stream around the K4 / GEMM launches -> per-kernel durations for the roofline
    times, elapsed_instr = instrumented_kernel_times(wl, args.steps, stream, barrier)
    fused_times, _ = instrumented_kernel_times(wl, args.steps, stream, barrier, fused=True)

    # ---- secondary measurement: a training step WITH backward (gnnsaft_backward), eager; for N > 1 followed by
    # the single flat RCCL all-reduce of the gradients (what DDP does for the reference, train.py:142-145)
    def measure_train_step():
        from gnn_epc_saft_amd.train.loop import allreduce_gradients
        conf = model.configure_optimizers()        # fused AdamW(amsgrad) + CosineAnnealingWarmRestarts
        opt, sched = conf["optimizer"], conf["lr_scheduler"]["scheduler"]

        def train_step():
            opt.zero_grad(set_to_none=True)
            loss_t = model.training_step(ddev)      # gnnsaft_forward (tape) + MAPE
            loss_t.backward()                       # gnnsaft_mape_backward + gnnsaft_backward -> one flat buffer
            allreduce_gradients(opt)                # N > 1: one RCCL all-reduce(SUM); mean folded into the step
            opt.step()                              # gnnsaft_adamw_step on the flat parameter buffer
            sched.step()
            return loss_t

        with torch.cuda.stream(stream):
            for _ in range(3):
                train_step()
            barrier()
            t2 = time.perf_counter()
            for _ in range(args.train_steps):
                train_step()
            barrier()
            el = time.perf_counter() - t2
        tt = torch.tensor([el], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt[0])
        out = {"what": "forward + MAPE + backward (all parameter gradients)" +
                       (" + flat gradient all-reduce (RCCL)" if world > 1 else "") +
                       " + fused AdamW(amsgrad) step + LR schedule step, eager",
               "steps": args.train_steps, "ms_per_step": el / args.train_steps * 1e3,
               "graphs_per_s": cfg["graphs"] * world * args.train_steps / el}
        if world == 1:
            # the same step captured in ONE hipGraph (train/loop.py::GraphedTrainingStep): the eager figure above is
            # bound by the host (~140 launches, ~1.5 ms of enqueue per step at C2), this one by the GPU
            try:
                import gnn_epc_saft_amd as G
                with torch.cuda.stream(stream):
                    graphed = G.GraphedTrainingStep(model, opt, ddev, scheduler=sched, warmup=2, choose="auto")
                    picked, graphed.mode = graphed.mode, "graph"     # time the replay itself ...
                    for _ in range(3):
                        graphed()
                    barrier()
                    t3 = time.perf_counter()
                    for _ in range(args.train_steps):
                        graphed()
                    barrier()
                    eg = time.perf_counter() - t3
                    graphed.mode = picked                            # ... and report what choose="auto" settled on
                out["hipgraph"] = {"what": "the same step replayed from one captured hipGraph (fixed batch)",
                                   "steps": args.train_steps, "ms_per_step": eg / args.train_steps * 1e3,
                                   "graphs_per_s": cfg["graphs"] * args.train_steps / eg,
                                   "auto_choice": {"mode": picked, "trial_ms": graphed.trial_ms,
                                                   "what": "GraphedTrainingStep(choose='auto') times replay against "
                                                           "eager at construction and steps with the faster one"}}
            except Exception as exc:  # noqa: BLE001
                out["hipgraph"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        return out

    train = None
    if args.train_steps > 0:
        if world > 1:
            train = measure_train_step()      # collectives inside: a failure must stay loud on every rank
        else:
            try:                              # secondary measurement: never lose the headline line over it
                train = measure_train_step()
            except Exception as exc:  # noqa: BLE001
                train = {"error": f"{type(exc).__name__}: {exc}"[:300]}

    # What a HIP event pair measures around a launch that does (almost) nothing (~6 us: the launch's fixed cost plus
    # ~2.4 us of dispatch latency / event handling).  Reported beside the K4 time to explain the gap between the
    # event-timed and the rocprofv3 kernel duration; the roofline `achieved` keeps the raw event time.
    event_overhead_ms = None
    if rank == 0 and world == 1:
        from gnn_epc_saft_amd import kernels as _k
        tiny_p, tiny_t = torch.ones(1, 3, device=dev), torch.ones(1, 3, device=dev)
        pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(64)]
        with torch.cuda.stream(stream):
            for _ in range(8):
                _k.mape(tiny_p, tiny_t)
            for a, b in pairs:
                a.record(stream)
                _k.mape(tiny_p, tiny_t)
                b.record(stream)
        torch.cuda.synchronize(dev)
        gaps = sorted(a.elapsed_time(b) for a, b in pairs)
        event_overhead_ms = gaps[len(gaps) // 2]

    # ---- secondary block (N = 1 line of the default config only): BASELINE.json configs[2], whose K4 working set
    # (q 336 MB + agg 1.3 GB) is far beyond the Infinity Cache -- the HBM figure that cannot be an L3 artefact
    c3_block = None
    if rank == 0 and world == 1 and args.config == 2 and not args.no_c3:
        try:
            c3_block = measure_c3(dev, stream, event_overhead_ms, train_steps=5 if args.train_steps > 0 else 0)
        except Exception as exc:  # noqa: BLE001
            c3_block = {"error": f"{type(exc).__name__}: {exc}"[:300]}

    # ---- N > 1: BASELINE.json configs[3] (65 536 graphs = 8 x 8192 on the H=256 / L=5 model): every rank takes part
    flag_words = {"c2": model.model.input_error_flags()}
    c4_block = None
    if world > 1 and args.config == 2 and not args.no_c4:
        c4_block, flag_words["c4"] = measure_c4(dev, rank, world, stream)
    if c3_block is not None:
        flag_words["c3"] = int(c3_block.pop("_flags", 0))

    t = torch.tensor([elapsed, elapsed_instr, elapsed_eager], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, elapsed_instr, elapsed_eager = float(t[0]), float(t[1]), float(t[2])
    flags = reduce_flags(flag_words, dev, world)

    if rank == 0:
        total_graphs = cfg["graphs"] * world * args.steps
        roof, gemm = roofline_blocks(wl, times, event_overhead_ms, fused_times)
        out = {
            "metric": "molecular graphs/sec (forward+loss)",
            "value": total_graphs / elapsed,
            "unit": "graphs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": wl.describe() + ", train-mode BatchNorm forward + MAPE loss, no backward",
                "graphs_per_gpu": cfg["graphs"], "nodes": wl.n, "edges": wl.e, "edges_with_self_loops": wl.e_prime,
                "launch": "hipGraph replay" if graph is not None else "eager (one C call per step)" +
                          (f" [{graph_note}]" if graph_note else ""),
                "loss_exchange": "RCCL all-reduce of [sum(ape), count]" if world > 1 else "none (1 GPU)",
            },
            "roofline": roof,
            "roofline_gemm": gemm,
            "eager_ms_per_step": elapsed_eager / args.steps * 1e3,
            "instrumented_ms_per_step": elapsed_instr / args.steps * 1e3,
            "final_loss": float(final_loss),
            "train_step": train,
        }
        if c3_block is not None:
            out["c3"] = c3_block
        if c4_block is not None:
            out["c4"] = c4_block
        out["input_error_flags"] = flags["any"]
        out["input_error_flags_by_block"] = flags["by_block"]
        if not args.no_cpu_baseline and world == 1:   # timed on rank 0 at N = 1 only
            try:
                out["cpu_baseline"] = cpu_baseline(cfg, wl.data, wl.deg, args.cpu_budget)
            except Exception as exc:  # noqa: BLE001
                out["cpu_baseline"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if flags["any"]:
        raise SystemExit(f"bench.py: GNNSAFT_FLAG_* bits raised during the run: {flags} "
                         "(16 = BARRIER_TIMEOUT of the fused readout: results invalid)")


def reduce_flags(words, dev, world):
    """OR of the per-block flag words over all ranks (bit by bit through a MAX all-reduce: RCCL has no bitwise OR)."""
    import torch.distributed as dist
    names = sorted(words)
    bits = torch.tensor([[(int(words[n]) >> b) & 1 for b in range(8)] for n in names], dtype=torch.int32, device=dev)
    if world > 1:
        dist.all_reduce(bits, op=dist.ReduceOp.MAX)
    by_block = {n: int(sum(int(v) << b for b, v in enumerate(row))) for n, row in zip(names, bits.tolist())}
    total = 0
    for v in by_block.values():
        total |= v
    return {"any": total, "by_block": by_block}


def measure_c4(dev, rank, world, stream, steps: int = 10, warmup: int = 3, train_steps: int = 5):
    """BASELINE.json configs[3] on `world` ranks: 8192 synthetic graphs PER RANK (65 536 at 8 ranks), H=256 / L=5 model.
    (a) forward + MAPE loss with the [sum(ape), count] all-reduce: whole-job graphs/s from the max-over-ranks time of
    `steps` eager steps; (b) the full training step (forward with tape, backward, gradient exchange, fused AdamW,
    scheduler) in both exchange modes the loop offers -- ONE flat all-reduce of the whole gradient buffer (28 MB), and
    the segment-wise exchange overlapped with the backward -- with the collective itself bracketed by HIP events on the
    compute stream: `allreduce_ms` (single collective: the stream waits for RCCL between the two events) resp.
    `exposed_exchange_ms` (overlapped: from the end of the backward to the last segment's arrival).  Replaces what
    Lightning DDP does for the reference (train/train.py:142-156, sync_dist at models.py:195-201)."""
    import torch.distributed as dist

    from gnn_epc_saft_amd import parallel
    from gnn_epc_saft_amd.train.loop import broadcast_training_state
    wl = Workload(CONFIGS[3], dev, rank, 3)
    model, ddev = wl.model, wl.ddev

    def sync():
        dist.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(seconds):
        tt = torch.tensor([seconds], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt[0])

    # ---- (a) forward + loss
    with torch.no_grad(), torch.cuda.stream(stream):
        pend = []
        for _ in range(warmup):
            pend.append(parallel.global_mape_async(model.training_step_parts(ddev)))
        loss = pend[-1].result()
        sync()
        t0 = time.perf_counter()
        pend = []
        for _ in range(steps):
            pend.append(parallel.global_mape_async(model.training_step_parts(ddev)))
        for h in pend:
            loss = h.result()
        sync()
        fwd = max_over_ranks(time.perf_counter() - t0)
    block = {
        "workload": wl.describe() + f", x {world} ranks = {wl.cfg['graphs'] * world} graphs per step (BASELINE.json "
                                    "configs[3]); per-rank BatchNorm statistics, as the reference (no SyncBatchNorm)",
        "forward_loss": {"steps": steps, "warmup": warmup, "ms_per_step": fwd / steps * 1e3,
                         "graphs_per_s": wl.cfg["graphs"] * world * steps / fwd, "global_mape": float(loss),
                         "exchange": "all-reduce(SUM) of [sum(ape), count] per step, asynchronous"},
    }

    # ---- (b) training step, both exchange modes
    conf = model.configure_optimizers()
    opt, sched = conf["optimizer"], conf["lr_scheduler"]["scheduler"]
    broadcast_training_state(model, opt, sched, 0)
    grad_bytes = int(opt._total) * 4
    exchange = parallel.OverlappedGradientExchange(model.model, dev)

    def step_single(ev):
        opt.zero_grad(set_to_none=True)
        model.training_step(ddev).backward()
        flat = opt._flat_grad()                       # gnnsaft_backward's own buffer: zero copy
        ev[0].record()
        scale = parallel.allreduce_flat_sum(flat)     # ONE collective over the whole gradient
        ev[1].record()
        opt.use_reduced_gradient(flat, scale)
        opt.step()
        sched.step()

    def step_overlap(ev):
        opt.zero_grad(set_to_none=True)
        model.training_step(ddev).backward()          # records an event per finished gradient segment
        flat = parallel.common_gradient_buffer(opt._params)
        ev[0].record()                                # end of the backward on the compute stream
        opt.use_reduced_gradient(flat, exchange.launch(flat))
        exchange.wait()                               # compute stream waits for the exchange stream
        ev[1].record()
        opt.step()
        sched.step()

    train = {"gradient_bytes": grad_bytes}
    try:
        for name, fn in (("single_collective", step_single), ("segment_overlapped", step_overlap)):
            with torch.cuda.stream(stream):
                scratch = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                for _ in range(2):
                    fn(scratch)
                sync()
                pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                         for _ in range(train_steps)]
                t0 = time.perf_counter()
                for ev in pairs:
                    fn(ev)
                sync()
                el = max_over_ranks(time.perf_counter() - t0)
            ex_ms = sorted(a.elapsed_time(b) for a, b in pairs)[len(pairs) // 2]
            ex_ms = max_over_ranks(ex_ms)
            entry = {"steps": train_steps, "ms_per_step": el / train_steps * 1e3,
                     "graphs_per_s": wl.cfg["graphs"] * world * train_steps / el}
            if name == "single_collective":
                entry["allreduce_ms"] = ex_ms
                entry["allreduce_bus_gb_per_s"] = 2.0 * (world - 1) / world * grad_bytes / (ex_ms * 1e-3) / 1e9
                entry["how"] = ("HIP events on the compute stream around dist.all_reduce(flat, SUM): the stream waits "
                                "for the collective between them (median over the steps, max over ranks)")
            else:
                entry["exposed_exchange_ms"] = ex_ms
                entry["segments"] = len(exchange.segments)
                entry["how"] = ("readout / layer L-1..0 / embedding segments all-reduced on a second stream behind "
                                "gnnsaft_backward's per-segment events; events from the end of the backward to the "
                                "arrival of the last segment (what the overlap leaves exposed)")
            train[name] = entry
    finally:
        exchange.close()
    block["train_step"] = train
    flags = model.model.input_error_flags()
    del wl, model, ddev, opt, conf
    torch.cuda.empty_cache()
    return (block if rank == 0 else None), flags


def measure_c3(dev, stream, event_overhead_ms, steps: int = 10, warmup: int = 3, train_steps: int = 5):
    """10 eager steps of BASELINE.json configs[2] on this GPU: graphs/s plus the K4 / GEMM rooflines."""
    wl = Workload(CONFIGS[3], dev, 0, 3)

    def barrier():
        torch.cuda.synchronize(dev)

    with torch.no_grad(), torch.cuda.stream(stream):
        for _ in range(warmup):
            wl.model.training_step_parts(wl.ddev)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            parts = wl.model.training_step_parts(wl.ddev)
        barrier()
        elapsed = time.perf_counter() - t0
    times, _ = instrumented_kernel_times(wl, steps, stream, barrier)
    fused_times, _ = instrumented_kernel_times(wl, steps, stream, barrier, fused=True)
    roof, gemm = roofline_blocks(wl, times, event_overhead_ms, fused_times)
    out = {"workload": wl.describe() + ", train-mode BatchNorm forward + MAPE loss, eager", "steps": steps,
           "warmup": warmup, "ms_per_step": elapsed / steps * 1e3, "graphs_per_s": wl.cfg["graphs"] * steps / elapsed,
           "nodes": wl.n, "edges_with_self_loops": wl.e_prime, "final_loss": float(parts[0]),
           "roofline": roof, "roofline_gemm": gemm}
    if train_steps > 0:
        # the training step on the same batch: forward (tape) + MAPE + backward + fused AdamW + schedule, eager
        conf = wl.model.configure_optimizers()
        opt, sched = conf["optimizer"], conf["lr_scheduler"]["scheduler"]

        def train_step():
            opt.zero_grad(set_to_none=True)
            wl.model.training_step(wl.ddev).backward()
            opt.step()
            sched.step()

        with torch.cuda.stream(stream):
            for _ in range(2):
                train_step()
            barrier()
            t1 = time.perf_counter()
            for _ in range(train_steps):
                train_step()
            barrier()
            el = time.perf_counter() - t1
        out["train_step"] = {"what": "forward + MAPE + backward (all parameter gradients) + fused AdamW(amsgrad) step "
                                     "+ LR schedule step, eager", "steps": train_steps,
                             "ms_per_step": el / train_steps * 1e3,
                             "graphs_per_s": wl.cfg["graphs"] * train_steps / el,
                             "vs_forward_loss": (el / train_steps) / (elapsed / steps)}
        del opt, sched, conf
    out["_flags"] = wl.model.model.input_error_flags()
    del wl
    torch.cuda.empty_cache()
    return out


def bench_training_loop(args, dev, rank: int, world: int) -> None:
    """--config 5: the C5 stand-in of SURVEY.md 8(d) -- BASELINE.json configs[4] needs the DVC/GCS dataset, which
    cannot be fetched -- 2 000 synthetic graphs, batch 512 (configs/default.py:20), default model H=64 L=6 P=5
    (configs/default.py:35-45), AdamW(amsgrad) + CosineAnnealingWarmRestarts, shuffled epochs through GraphLoader
    with strided rank shards.  A step = forward (tape) + MAPE + backward + [flat gradient all-reduce] + fused
    optimizer step + scheduler step over one batch."""
    import torch.distributed as dist

    import gnn_epc_saft_amd as G
    from gnn_epc_saft_amd.data.loader import GraphLoader
    from gnn_epc_saft_amd.data.synthetic import degree_histogram, synthetic_dataset
    from gnn_epc_saft_amd.train.loop import allreduce_gradients, broadcast_training_state
    cfg = CONFIGS[5]
    graphs = synthetic_dataset(cfg["dataset"], 1234 + 5, num_para=cfg["num_para"])
    config = dict(propagation_depth=cfg["depth"], hidden_dim=cfg["hidden"], pre_layers=1, post_layers=1,
                  num_mlp_layers=1, num_para=cfg["num_para"], skip_connections=True, add_self_loops=True,
                  dropout_rate=0.0, model="PNAL", optimizer="adam", learning_rate=1e-3, weight_decay=1e-2,
                  warmup_steps=100, momentum=0.9)
    torch.manual_seed(0)
    lit = G.create_model(config, degree_histogram(graphs)).to(dev).train()
    loader = GraphLoader(graphs, cfg["graphs"], shuffle=True, device=dev, seed=0, rank=rank, world_size=world)
    conf = lit.configure_optimizers()
    opt, sched = conf["optimizer"], conf["lr_scheduler"]["scheduler"]
    broadcast_training_state(lit, opt, sched, 0)

    it = loader.forever()     # epochs back to back, batches prefetched by the loader's background thread
    seen = [0]

    def train_step():
        batch = next(it)
        seen[0] += batch.num_graphs
        opt.zero_grad(set_to_none=True)
        loss = lit.training_step(batch)
        loss.backward()
        allreduce_gradients(opt)
        opt.step()
        sched.step()
        return loss

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    first = None
    for i in range(max(args.warmup, 1)):
        loss = train_step()
        if i == 0:
            first = float(loss)
    barrier()
    seen[0] = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = train_step()
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed, float(seen[0])], dtype=torch.float64, device=dev)
    if world > 1:
        tm = t[:1].clone()
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        ts = t[1:].clone()
        dist.all_reduce(ts, op=dist.ReduceOp.SUM)
        elapsed, total = float(tm[0]), float(ts[0])
    else:
        total = float(seen[0])
    flags = reduce_flags({"c5": lit.model.input_error_flags()}, dev, world)["any"]
    if rank == 0:
        print(json.dumps({
            "metric": "molecular graphs/sec (training step: forward+loss+backward+optimizer)",
            "value": total / elapsed, "unit": "graphs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "train_steps_per_s": args.steps / elapsed,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C5 stand-in (BASELINE.json configs[4]; the ThermoML-derived dataset is a DVC/GCS "
                                   f"pointer that cannot be fetched): {cfg['dataset']} synthetic molecular graphs, batch "
                                   f"{cfg['graphs']} (last batch of an epoch smaller), shuffled epochs, PNAPCSAFT H=64 L=6 "
                                   "pre=post=1 mlp=1 P=5 skip+self-loops (configs/default.py), AdamW(amsgrad, eps 1e-5, "
                                   "wd 1e-2) + CosineAnnealingWarmRestarts(100), eager",
                       "dataset_graphs": cfg["dataset"], "batch_size": cfg["graphs"],
                       "gradient_exchange": "one flat RCCL all-reduce(SUM) per step" if world > 1 else "none (1 GPU)"},
            "first_loss": first, "final_loss": float(loss), "input_error_flags": flags,
        }))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if flags:
        raise SystemExit(f"bench.py: GNNSAFT_FLAG_* bits {flags:#x} raised during the run")


if __name__ == "__main__":
    main()
