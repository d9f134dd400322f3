#!/usr/bin/env python3
"""bench.py -- throughput of the hot path on MI355X, with roofline and CPU-baseline legs.

A "step" is one pass of the hot path over one batch of synthetic counts that are already resident
in HBM: per-sample constants -> beta-binomial emissions for every (exon, sample) cell -> Viterbi per
(sample, chromosome) chain -> call table (+ the per-sample dispersion fit when --fit).  Workload:
BASELINE.json configs[2] geometry, 200 000 exons x 1024 samples per GPU (weak scaling: N GPUs hold
N x 1024 samples, configs[3] at N=8).  Metric: exons*samples/s (BASELINE.json.metric).

Contract: `python bench.py --gpus N --steps K --warmup W`; for N>1 launched with
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (one rank per GPU,
RCCL).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
FP64_VALU_PEAK_TFLOPS = 78.6   # FP64 vector peak (no MFMA applies to this path)
ALGO_BYTES_PER_CELL = 9        # read test 4 B + read reference 4 B + write state 1 B


def pmc_traffic(kernel, n_launch):
    """HBM bytes per launch of `kernel` (n_launch launches per step) from the committed rocprofv3 PMC passes (profiles/*_pmc_traffic.json,
    written by tools/profile_to_json.py): FETCH_SIZE x2 (gfx950 tallies the 128-byte requests of a coalesced
    stream at 64 B, MI355X_MICROARCH.md) + WRITE_SIZE, KB -> bytes.  None if no profile is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    if kernel not in d:
        return None, None
    k = d[kernel]   # per-step total of the profiled run, re-divided by this run's launches per step
    per_step = k["hbm_bytes_per_launch"] * k["launches_profiled"] / k.get("steps_profiled", 3)
    return per_step / n_launch, os.path.basename(files[-1])


def pmc_valu(kernel, cells_per_step):
    """VALU figures of `kernel` from the committed rocprofv3 PMC passes (profiles/*_pmc_SQ.csv, *_pmc_GRBM_GUI_ACTIVE.csv,
    summarised per launch by tools/pmc_summary.py): wave-instructions per cell and the fraction of SIMD issue cycles spent
    on them, SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 XCDs) / 1024 SIMDs.  None if no profile is committed."""
    import csv, glob
    sq = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_SQ.csv")))
    gr = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_GRBM_GUI_ACTIVE.csv")))
    if not sq or not gr:
        return None
    def load(f):
        return {(r["kernel"], r["counter"]): (float(r["mean_per_launch"]), int(r["launches"])) for r in csv.DictReader(open(f))}
    a, b = load(sq[-1]), load(gr[-1])
    try:
        insts, n = a[(kernel, "SQ_INSTS_VALU")]
        active = a[(kernel, "SQ_ACTIVE_INST_VALU")][0]
        gui = b[(kernel, "GRBM_GUI_ACTIVE")][0]
    except KeyError:
        return None
    steps = 3   # tools/profile_round.sh: --steps 2 --warmup 1
    return {"valu_wave_instructions_per_cell": insts * n / steps / (cells_per_step / 64.0),
            "valu_busy": active * 4.0 / (gui / 8.0) / 1024.0,
            "source": [os.path.basename(sq[-1]), os.path.basename(gr[-1])]}


def cpu_baseline(test_h, ref_h, p, phi, chrom_off, start, end, fit, allcores=False, test_all=None, p_all=None, phi_all=None):
    """The CPU checker's libm flavour (bit-identical to the reference's compiled special functions)
    timed on one host core over a bounded sample of the same workload."""
    from oracle import edoracle as eo

    eo.build()
    n_s = test_h.shape[1]
    t_emit = t_vit = 0.0
    for s in range(n_s):
        t0 = time.perf_counter()
        ll, _ = eo.get_loglike_matrix(phi[s], p[s], test_h[:, s] + ref_h[:, s], test_h[:, s], 1.0, eo.LIBM)
        t1 = time.perf_counter()
        eo.callcnvs(ll, chrom_off, start, end)
        t2 = time.perf_counter()
        t_emit += t1 - t0
        t_vit += t2 - t1
    cells = test_h.shape[0] * n_s
    t_fit = 0.0
    n_fit = 0
    if fit:
        # stand-in for aod::betabin (not in the reference tree): Nelder-Mead on the same likelihood
        n_fit = min(n_s, 4)
        t0 = time.perf_counter()
        for s in range(n_fit):
            eo.fit_nm(test_h[:, s], ref_h[:, s])
        t_fit = (time.perf_counter() - t0) * (n_s / n_fit)   # extrapolated linearly to the n_s samples
    extra = {}
    if allcores:
        # the same work, one sample per process on every host core (the reference is single-threaded: reported for
        # completeness)
        from oracle import cpu_bench
        ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        v, wall, nproc = cpu_bench.all_cores(test_all[0], test_all[1], p_all, phi_all, chrom_off, start, end, fit, ncores)
        extra = {"all_cores": {"value": v, "unit": "exons*samples/s", "cores": nproc, "wall_s": wall,
                               "sample": "%d samples x %d exons, one process per core, same work per sample" % (nproc, test_h.shape[0])}}
    return {**extra, "value": cells / (t_emit + t_vit + t_fit), "unit": "exons*samples/s", "cores": 1, "kind": "port",
            "sample": "%d samples x %d exons of the same synthetic batch: emissions + Viterbi + call table with the "
                      "oracle's libm flavour (bit-identical to the reference's compiled lnbeta), single thread%s"
                      % (n_s, test_h.shape[0],
                         "; dispersion fit = Nelder-Mead stand-in for aod::betabin timed on %d samples and "
                         "extrapolated linearly" % n_fit if fit else ""),
            "emissions_s": t_emit, "viterbi_s": t_vit, "fit_s": t_fit}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--exons", type=int, default=200_000)
    ap.add_argument("--samples", type=int, default=1024, help="samples per GPU")
    ap.add_argument("--chroms", type=int, default=24)
    ap.add_argument("--depth", type=float, default=100.0, help="median reads per exon and sample of the synthetic counts (SURVEY.md 8d: 100)")
    ap.add_argument("--fit", type=int, default=1, help="1 (default): the step includes the per-sample dispersion fit (configs[2]); 0: phi given (configs[1] style)")
    ap.add_argument("--fused", type=int, default=0, help="1: emissions + Viterbi as one kernel (csrc/edfused.inc)")
    ap.add_argument("--keep-loglik", type=int, default=1, help="fused mode: 0 = do not materialise the likelihood matrix")
    ap.add_argument("--phi-bins", type=int, default=1, help="> 1: the depth-binned dispersion model (phi.bins, csrc/edbins.inc); "
                    "an optional mode, not the headline configuration")
    ap.add_argument("--cov", type=int, default=0, help="> 0: the mean model with that many per-exon covariates (csrc/edcov.inc); "
                    "an optional mode, not the headline configuration")
    ap.add_argument("--cpu-all-cores", type=int, default=1, help="1: also time the CPU baseline with one sample per host core "
                    "(process-level parallelism; reported inside cpu_baseline.all_cores)")
    ap.add_argument("--cpu-samples", type=int, default=12, help="columns timed on the host for cpu_baseline (0 = skip)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # ED_BENCH_BACKEND=gloo + ED_BENCH_SHARE_GPU=1 let the N>1 code path be exercised on a 1-GPU box (functional
    # check only: every rank then uses GPU 0 and the call-table gather goes through host memory)
    backend = os.environ.get("ED_BENCH_BACKEND", "nccl")
    if os.environ.get("ED_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    cdev = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")   # device of the collectives
    if args.gpus != world:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N>1)"
                  % (args.gpus, world), file=sys.stderr)
        if args.gpus > 1 and world == 1:
            sys.exit(2)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    import exomedepth_amd as ed
    from exomedepth_amd import _build, dist as eddist
    if not os.path.exists(_build.LIB) and rank == 0:   # never-built tree: compile the HIP library (there is no other path)
        _build.build()
    if world > 1:
        dist.barrier()
    from exomedepth_amd import synth

    E, S, C = args.exons, args.samples, args.chroms
    chrom_off, start, end = synth.exon_design(E, C, seed=20250620)
    torch.manual_seed(20250620 + 3 + rank)
    test, ref, p, phi = synth.counts_torch(chrom_off, S, dev, seed=20250620 + 3 + 1000 * rank, mean_depth=args.depth)
    torch.cuda.synchronize()

    plan = ed.Plan(chrom_off, start, end, 1e-4, 50000.0, device=local_rank)
    batch = ed.Batch(plan, S)
    batch.enable_timing(True)
    batch.set_fused(bool(args.fused))
    batch.keep_loglik(bool(args.keep_loglik))
    stream = torch.cuda.current_stream().cuda_stream
    phi_fit = torch.empty(S, dtype=torch.float64, device=dev)
    p_fit = torch.empty(S, dtype=torch.float64, device=dev)
    phib_fit = torch.empty((max(args.phi_bins, 1), S), dtype=torch.float64, device=dev)
    Xcov = (torch.rand((E, max(args.cov, 1)), dtype=torch.float64, device=dev) - 0.5) * 0.4 if args.cov > 0 else None
    beta_fit = torch.empty((max(args.cov, 0) + 1, S), dtype=torch.float64, device=dev)
    edges_fit = torch.empty((max(args.phi_bins, 1) + 1, S), dtype=torch.float64, device=dev)

    def step():
        if args.cov > 0:
            batch.fit_cov(test, ref, Xcov, beta_fit, phi_fit, stream=stream)
            batch.run_cov(test, ref, Xcov, beta_fit, phi_fit, 1.0, stream=stream)
        elif args.phi_bins > 1:
            batch.fit_bins(test, ref, args.phi_bins, phib_fit, edges_fit, p_fit, stream=stream)
            batch.run_bins(test, ref, args.phi_bins, phib_fit, edges_fit, p_fit, 1.0, stream=stream)
        elif args.fit:
            batch.fit(test, ref, phi_fit, p_fit, stream=stream)
            batch.run(test, ref, phi_fit, p_fit, 1.0, stream=stream)
        else:
            batch.run(test, ref, phi, p, 1.0, stream=stream)

    def finish():
        """final gather of the compact call tables (the path's only collective)"""
        calls = batch.calls()
        if world > 1:
            t = eddist.calls_to_tensor(calls, cdev)
            g = eddist.gather_call_tables(t, rank * S)
            return int(g.shape[0]) if g is not None else 0
        return len(calls)

    for _ in range(args.warmup):
        step()
    finish()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    stage_acc = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        ms = batch.stage_ms()   # reads the HIP events of this step (synchronises the stream)
        for k, v in ms.items():
            stage_acc[k] = stage_acc.get(k, 0.0) + v
    n_calls = finish()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    cells_per_step = E * S * world
    value = cells_per_step * args.steps / elapsed
    stage_ms = {k: v / args.steps for k, v in stage_acc.items()}

    if rank == 0:
        t_emit = stage_ms["emissions"] * 1e-3
        achieved = ALGO_BYTES_PER_CELL * E * S / t_emit / 1e9 if t_emit > 0 else 0.0
        n_launch = max(1, batch.n_emit_launches)   # one emission launch per overlap group of chromosomes
        traffic, traffic_src = pmc_traffic("k_emit_viterbi" if args.fused else "k_emit_batch", n_launch)
        out = {
            "metric": "exons*samples/s through betabinom emissions + Viterbi" + (" + dispersion fit" if args.fit else ""),
            "value": value, "unit": "exons*samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[2] geometry: %d exons x %d samples per GPU, %d chromosomes, "
                                   "phi %s, transition.probability 1e-4, expected.CNV.length 5e4"
                                   % (E, S, C, "fitted on device" if args.fit else "given per sample (fixed)"),
                       "exons": E, "samples_per_gpu": S, "samples_total": S * world, "fit": bool(args.fit), "fused": bool(args.fused), "phi_bins": args.phi_bins, "covariates": args.cov,
                       "parallelism": "samples sharded, %d rank(s); call tables gathered over RCCL" % world},
            "roofline": {"bound": "hbm", "kernel": "k_emit_viterbi" if args.fused else "k_emit_batch", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_cell": ALGO_BYTES_PER_CELL, "launches_per_step": n_launch,
                         "kernel_ms": stage_ms["emissions"] / n_launch, "kernel_ms_per_step": stage_ms["emissions"],
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_CELL * E * S / n_launch,
                         "kernel_cells_per_s": (E * S / t_emit if t_emit else 0.0),
                         "algorithmic_bytes_per_cell_with_likelihood_matrix": 33,
                         "valu": pmc_valu("k_emit_viterbi" if args.fused else "k_emit_batch", float(E) * S),
                         "note": "FP64-VALU-bound kernel (no MFMA applies; SURVEY.md 0.5): the HBM roofline is the formal "
                                 "denominator. rocprofv3 PMC (profiles/r01_k_pmc_SQ.csv, r01_k_pmc_GRBM_GUI_ACTIVE.csv): 1253 VALU "
                                 "instructions per cell at 85% VALU-busy (the kernel alone runs at 8.8 ms per step; the Viterbi kernels sharing the SIMDs cost it 1.2 ms, DESIGN.md 4.2). traffic exceeds the 9 B/cell figure because the "
                                 "kernel materialises the [E][3][S] f64 likelihood matrix (the reference's S4 `likelihood` "
                                 "slot: 24 B/cell written once, read once by the Viterbi; 33 B/cell algorithmic in that form, "
                                 "SURVEY.md 8d) and gathers tabulated terms (L2-resident tables; the misses are counted with "
                                 "the guide's x2 on FETCH_SIZE, which over-counts 64-byte lines; see profiles/README.md)"},
            "stage_ms": stage_ms, "n_calls": n_calls,
        }
        if world == 1 and args.cpu_samples > 0:
            k = min(args.cpu_samples, S)
            ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            ka = min(ncores, S) if args.cpu_all_cores else 0
            out["cpu_baseline"] = cpu_baseline(test[:, :k].cpu().numpy(), ref[:, :k].cpu().numpy(),
                                               p[:k].cpu().numpy(), phi[:k].cpu().numpy(), chrom_off, start, end, bool(args.fit),
                                               allcores=ka > 0,
                                               test_all=(test[:, :ka].cpu().numpy(), ref[:, :ka].cpu().numpy()) if ka else None,
                                               p_all=p[:ka].cpu().numpy() if ka else None, phi_all=phi[:ka].cpu().numpy() if ka else None)
            out["speedup_vs_cpu_1core"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    batch.close()
    plan.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
