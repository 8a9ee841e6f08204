#!/usr/bin/env python3
"""bench.py -- site x pop-pair Fst+dxy evaluations per second on MI355X.

A "step" is one pass of the hot path (K1 allele counts + census, offsets scan, K2 pair statistics
with fused segment window sums, stitch) over one batch of synthetic genotypes that is already
resident in HBM.  Workload at N=1: BASELINE.json configs[1] (1 M sites x 200 diploid samples,
10 populations = 45 pairs, SNP windows -w 100,100).  For N>1 every rank holds its own 1 M-site
shard of one genome (weak scaling); SNP windows couple the shards through one exchange step
per pass: an all-gather of the per-pair included-site totals (RCCL), from which each rank derives
the global window phase of its shard.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (sites per GPU, samples, pops, mode, size, step, missing)
    "cfg2": dict(sites=1_000_000, samples=200, pops=10, mode="pairs", size=100, step=100, miss=0.0,
                 label="synthetic 1M sites x 200 diploid samples, 10 pops (45 pairs), SNP windows 100,100"),
    "cfg2m": dict(sites=1_000_000, samples=200, pops=10, mode="pairs", size=100, step=100, miss=0.02,
                 label="cfg2 with 2% missing calls (diagnostic: every chunk takes the general walk)"),
    "cfg2nw": dict(sites=1_000_000, samples=200, pops=10, mode="pairs", size=0, step=0, miss=0.0,
                  label="cfg2 without windows (totals only; diagnostic)"),
    "cfg3s": dict(sites=2_000_000, samples=1000, pops=100, mode="global", size=20, step=10, miss=0.0,
                  label="synthetic 2M-site shard x 1000 samples, 100 pops, Fst-against-all, windows 20,10"),
    "cfg4s": dict(sites=6_250_000, samples=2000, pops=20, mode="pairs", size=100, step=100, miss=0.0,
                  label="synthetic 6.25M-site shard (1/8 of 50M) x 2000 samples, 20 pops (190 pairs), windows 100,100"),
    "cfg4m": dict(sites=6_250_000, samples=2000, pops=20, mode="pairs", size=100, step=100, miss=0.02,
                  label="cfg4s with 2% missing calls (diagnostic: every chunk takes the general walk)"),
    "cfg5s": dict(sites=2_500_000, samples=5000, pops=500, mode="gram", size=100, step=100, miss=0.15, bp=10_000,
                  label="synthetic 2.5M-site shard (1/8 of 20M) x 5000 samples, 500 pops (124,750 pairs), 15% missing, "
                        "Gram path (FP64 MFMA) over 10 kb physical windows cut from the shard's positions (one site per 100 bp)"),
    "cfg5s_w1000": dict(sites=2_500_000, samples=5000, pops=500, mode="gram", size=1000, step=1000, miss=0.15,
                        label="cfg5s with 1000-site windows (diagnostic: prologue / epilogue amortised over 10x the sites)"),
}
HBM_PEAK_GBS = 8000.0
F64_MFMA_PEAK_TFLOPS = 78.6  # MI355X FP64 matrix peak  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def host_cores():
    """Host cores this process may use: its affinity mask, capped by the pool's share (OMP_NUM_THREADS is 16 per GPU on
    the GPU box, where os.cpu_count() reports the whole machine)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        n = min(n, int(os.environ.get("OMP_NUM_THREADS", "16")))
    except ValueError:
        n = min(n, 16)
    return max(1, n)


def _cpu_pass(O, cfg, codes, sp, nf, pairs):
    """One pass of the CPU oracle over a range of sites: counts from allele codes, per-site statistics, windows."""
    alt, nm = O.counts_from_codes(codes, sp, cfg["pops"])
    if cfg["mode"] == "gram":
        pi, pj = pairs
        num, den, dxy, inc = O.pair_site_stats(alt, nm, nf, pi, pj)
        starts = np.arange(0, len(codes), cfg["size"])
        m = inc == 1
        for a in (num, den, dxy):
            np.add.reduceat(np.where(m, a, 0.0), starts, axis=0)
    elif cfg["mode"] == "pairs":
        pi, pj = pairs
        num, den, dxy, inc = O.pair_site_stats(alt, nm, nf, pi, pj)
        for q in range(len(pi)):
            O.snp_windows(num, den, dxy, inc, q, cfg["size"], cfg["step"])
    else:
        use = np.arange(cfg["pops"], dtype=np.int32)
        sizes = np.full(cfg["pops"], cfg["samples"] // cfg["pops"], np.int64)
        kept, num, den = O.global_site_stats(alt, nm, use, sizes)
        O.global_windows(kept, num, den, cfg["size"], cfg["step"])


def cpu_baseline(e, cfg, sample_sites):
    """The CPU oracle (closed-form restatement in C, oracle/evo_oracle.c) on the first sample_sites sites of the same
    workload, on one host core and then on every host core this process may use (site ranges in threads: the oracle's
    C functions run without the GIL).  Returns (evaluations/s on all cores, cores, seconds, evaluations/s on one core,
    seconds)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    S = sample_sites
    codes = e.unpack_codes(e.download_tiles(0, S), S)
    sp = (np.arange(cfg["samples"]) % cfg["pops"]).astype(np.int32)
    nf = e.pop_chroms()
    if cfg["mode"] == "global":
        pairs, units = None, cfg["pops"]
    else:
        pi, pj = np.triu_indices(cfg["pops"], 1)
        pairs, units = (pi.astype(np.int32), pj.astype(np.int32)), len(pi)
    evals = float(S) * units
    t0 = time.perf_counter()
    _cpu_pass(O, cfg, codes, sp, nf, pairs)
    dt1 = time.perf_counter() - t0
    cores = host_cores()
    # site ranges cut on multiples of the window step so every thread runs whole windows
    per = max(cfg["size"] or 1, -(-S // cores))
    per = -(-per // max(1, cfg["size"] or 1)) * max(1, cfg["size"] or 1)
    ranges = [(a, min(S, a + per)) for a in range(0, S, per)]
    with ThreadPoolExecutor(max_workers=cores) as ex:
        t0 = time.perf_counter()
        list(ex.map(lambda r: _cpu_pass(O, cfg, codes[r[0]:r[1]], sp, nf, pairs), ranges))
        dtn = time.perf_counter() - t0
    return evals / dtn, cores, dtn, evals / dt1, dt1


def timed_reps(run_steps, first=None, min_total=0.05, agree=None):
    """Repeats a timed region (run_steps() -> seconds of exactly --steps steps, fenced on both sides) until at least
    min_total seconds have been timed and at least three times (once if one region alone takes a second); `agree` makes
    the ranks of a multi-GPU run take the same number.  Returns the list of durations."""
    dts = [run_steps() if first is None else first]
    n = 1 if dts[0] >= 1.0 else max(3, int(np.ceil(min_total / max(dts[0], 1e-9))))
    n = min(n, 400)
    if agree is not None:
        n = agree(n)
    while len(dts) < n:
        dts.append(run_steps())
    return dts


PASS_NAMES = {1: "k1_count_kernel + scan_kernel + walk + stitch_kernel (separate kernels)", 2: "stream_pairs_kernel + stitch_kernel",
              3: "fused_pairs_kernel (lock-step) + stitch_kernel"}


def hbm_leg(device, steps=10, tune=None):
    """The pairs pass on one GPU's share of BASELINE configs[3] (6.25 M sites x 2000 samples, 190 pairs): 6.25 GB of tiles
    per step, far beyond every cache.  Returns the roofline object of its dominant kernel (command: bench.py --config cfg4s)."""
    from evo_amd import EvoFst
    cfg = CONFIGS["cfg4s"]
    S, N, P = cfg["sites"], cfg["samples"], cfg["pops"]
    e = EvoFst(device)
    if tune:
        tune(e)
    e.set_layout((np.arange(N) % P).astype(np.int32), P)
    e.set_pairs()
    e.set_windows(cfg["size"], cfg["step"])
    e.reserve(S)
    e.synth(S, seed=20261110, missing_rate=cfg["miss"])
    # two resident batches on two streams, as `--config cfg4s` itself runs: one pass's tail (its last chunks, the stitch)
    # overlaps the next one's start
    import torch
    e2 = EvoFst(device)
    if tune:
        tune(e2)
    e2.set_layout((np.arange(N) % P).astype(np.int32), P)
    e2.set_pairs()
    e2.set_windows(cfg["size"], cfg["step"])
    e2.reserve(S)
    e2.synth(S, seed=20261110, missing_rate=cfg["miss"], first_site_global=S)
    sts = [torch.cuda.Stream(), torch.cuda.Stream()]
    es = [e, e2]
    for x, st_ in zip(es, sts):
        x.set_stream(st_.cuda_stream)
    for i in range(4):
        es[i & 1].run()
    for x in es:
        x.sync()
    turn = [0]

    def run_steps():
        for x in es:
            x.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            es[turn[0] & 1].run()
            turn[0] += 1
        for x in es:
            x.sync()
        return time.perf_counter() - t0
    dts = np.array(timed_reps(run_steps)) / steps
    dt = float(np.median(dts))
    e.set_profiling(True)
    km = np.zeros((5, 5))
    for i in range(len(km)):
        e.run(); e.sync()
        for k in range(5):
            try:
                km[i, k] = e.kernel_ms(k)
            except Exception:
                km[i, k] = 0.0
    k = km.mean(axis=0)
    which = e.last_pairs_pass()
    cb = e.geometry().count_bytes
    g = np.gcd(cfg["size"], cfg["step"])
    rows_bytes = (S / g) * 190 * 28.0
    if which == 1:
        cbw = 1 if cb == 2 and not cfg.get("miss") else cb     # clean chunks leave one byte per count (K1Params::compact)
        dom, kernel = 0, "k1_count_kernel"
        dom_bytes = S * (N / 2.0) + S * P * cbw + S / 64.0 * (2 * P + 3) * 8.0
        kernels = {"k1_count_kernel": float(k[0]), "scan_kernel": float(k[1]),
                   "walk(clean_walk_kernel + fused_pairs_kernel on K1 counts)": float(k[2]), "stitch_kernel": float(k[3])}
    else:
        # one kernel from the tiles to the segment rows: 2 bits per haplotype read, one 28-byte row per segment and pair written
        dom, kernel = 4, "stream_pairs_kernel" if which == 2 else "fused_pairs_kernel"
        dom_bytes = S * (N / 2.0) + rows_bytes
        kernels = {kernel: float(k[4]), "stitch_kernel": float(k[3])}
        if k[2] > 0:
            kernels["walk of the chunks the stream pass marked"] = float(k[2])
    e.close()
    e2.close()
    ach = dom_bytes / (k[dom] * 1e-3) / 1e9 if k[dom] > 0 else 0.0
    return {"workload": cfg["label"], "pairs_pass": PASS_NAMES.get(which, "?"), "ms_per_step": dt * 1e3,
            "ms_per_step_min": float(dts.min()) * 1e3, "ms_per_step_max": float(dts.max()) * 1e3, "timed_reps": len(dts), "steps": steps,
            "site_pairs_per_s": S * 190 / dt, "bound": "hbm",
            "kernel": kernel, "algorithmic_bytes": dom_bytes, "achieved": ach,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "pipeline_GBps": S * (N / 2.0 + 4.0) / dt / 1e9, "pipeline_frac": S * (N / 2.0 + 4.0) / dt / 1e9 / HBM_PEAK_GBS,
            "kernels_ms": kernels, "resident_batches": 2, "streams": 2,
            "command": "python bench.py --config cfg4s"}


def scale_hbm_leg(local_rank, rank, world, tune, steps=10):
    """N > 1: the multi-rank pairs pass (counts -> included-site totals over RCCL -> base offsets -> walk -> carries) on
    BASELINE configs[3]'s per-GPU share, 6.25 M sites per rank -- north_star's 8-GPU pairs configuration.  Every rank
    calls this; returns the object for the JSON line (the slowest rank's time per repetition, median over repetitions)."""
    import torch
    import torch.distributed as dist
    from evo_amd import EvoFst, shard
    cfg = CONFIGS["cfg4s"]
    S, N, P = cfg["sites"], cfg["samples"], cfg["pops"]
    e = tune(EvoFst(local_rank))
    e.set_layout((np.arange(N) % P).astype(np.int32), P)
    e.set_pairs()
    e.set_windows(cfg["size"], cfg["step"])
    st = torch.cuda.Stream()
    e.set_stream(st.cuda_stream)
    e.reserve(S)
    e.synth(S, seed=20261110, missing_rate=cfg["miss"], first_site_global=rank * S)
    e.sync()
    msg = shard.PairsPassMessage(e.n_pairs, world, rank, "cuda")
    st.wait_stream(torch.cuda.current_stream())

    def one_pass():
        with torch.cuda.stream(st):
            e.run_counts()
            e.copy_included_totals(msg.totals_ptr())
            msg.gather(dist)
            e.set_base_offsets_gathered(msg.recv_ptr(), msg.row, rank)
            e.run_stats()
            e.copy_boundary(msg.carry_ptr())

    def fence():
        e.sync()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    def run_steps():
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            one_pass()
        with torch.cuda.stream(st):
            msg.gather(dist)                       # the carries of the last pass
        fence()
        return time.perf_counter() - t0

    def agree(n):
        t = torch.tensor([n], dtype=torch.int64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return int(t.item())
    for _ in range(2):
        one_pass()
    dts = timed_reps(run_steps, agree=agree)
    t = torch.tensor(dts, dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dts = np.array([float(x) for x in t.cpu()]) / steps
    dt = float(np.median(dts))
    e.close()
    return {"workload": cfg["label"] + " per rank", "rccl_ranks": dist.get_world_size(), "scaling": "weak", "sites_per_gpu": S,
            "pairs_pass": "k1_count_kernel -> all-gather of included-site totals (RCCL) -> scan_kernel + walk + stitch_kernel",
            "ms_per_step": dt * 1e3, "ms_per_step_min": float(dts.min()) * 1e3, "ms_per_step_max": float(dts.max()) * 1e3,
            "timed_reps": len(dts), "steps": steps, "value": S * 190.0 * world / dt, "unit": "site*pair/s",
            "pipeline_GBps": S * (N / 2.0 + 4.0) * world / dt / 1e9,
            "pipeline_frac_of_n_gpus": S * (N / 2.0 + 4.0) / dt / 1e9 / HBM_PEAK_GBS}


def spawn_ranks(n):
    """python bench.py --gpus N without torch.distributed.run: N children, one per GPU, rendezvous on 127.0.0.1."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = rc or p.wait()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--cpu-sites", type=int, default=-1, help="sites of the CPU baseline sample (0 = skip)")
    ap.add_argument("--dist-path", action="store_true", help="rehearsal: take the multi-rank step even with one rank")
    ap.add_argument("--rendezvous-check", action="store_true",
                    help="no GPU work: the ranks only meet (gloo, CPU) and rank 0 prints how many there are -- tests the launch path")
    ap.add_argument("--batches", type=int, default=0,
                    help="resident synthetic batches the timed loop rotates over (0 = enough to exceed the 256 MiB Infinity Cache)")
    ap.add_argument("--gram-denominator", type=int, default=0, choices=(0, 2),
                    help="evofst_set_option gram_denominator: 0 the library's choice, 2 always the two FP64 streams")
    ap.add_argument("--gram-bp", type=int, default=0,
                    help="experiments only: another physical window length for the Gram configuration (the label says so)")
    ap.add_argument("--gram-gb", type=float, default=8.0,
                    help="Gram path: windows per evofst_run_gram call, as GB of its four result arrays held at a time")
    ap.add_argument("--streams", type=int, default=3,
                    help="HIP streams the resident batches are spread over (1: the steps run one after the other; more: "
                         "independent batches overlap, as a run that keeps several batches in flight would)")
    ap.add_argument("--no-hbm-leg", action="store_true", help="skip the short cfg4-shard run behind roofline_hbm")
    ap.add_argument("--sites", type=int, default=0, help="experiments only: another number of sites per GPU (the label says so)")
    ap.add_argument("--pairs-pass", type=int, default=0, choices=(0, 1, 2, 3),
                    help="evofst_set_option pairs_pass: 0 the library's choice (the stream pass), 1 counts + scan + walk as "
                         "separate kernels, 2 stream pass, 3 the lock-step one-kernel pass")
    ap.add_argument("--chunk-tiles", type=int, default=0, help="evofst_set_option chunk_tiles (0: the library's choice)")
    ap.add_argument("--stream-blocks", type=int, default=0, help="evofst_set_option stream_blocks (0: what the device holds)")
    ap.add_argument("--stream-sets", type=int, default=0, help="evofst_set_option stream_sets (0: the library's choice)")
    ap.add_argument("--stream-buffers", type=int, default=0, help="evofst_set_option stream_buffers (0: the library's choice)")
    ap.add_argument("--stream-waves", default="0,0", help="evofst_set_option stream_producers,stream_consumers (0: the library's choice)")
    args = ap.parse_args()
    cfg = dict(CONFIGS[args.config])
    if args.sites > 0:
        cfg["sites"] = args.sites
        cfg["label"] += " [EXPERIMENT: %d sites per GPU]" % args.sites

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: start one child per GPU before this process touches torch.cuda or HIP, wait for all of them and
        # pass rank 0's JSON line through (rank 0 is the only child that writes to stdout)
        sys.exit(spawn_ranks(args.gpus))

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d (launch with torch.distributed.run --nproc-per-node %d, "
                         "or without a launcher: bench.py starts the ranks itself)" % (args.gpus, world, args.gpus))
    if args.rendezvous_check:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([rank + 1], dtype=torch.int64)
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"rendezvous_check": True, "n_gpus": args.gpus, "ranks": dist.get_world_size(), "rank_sum": int(t.item())}))
        dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback of the product path)")
    torch.cuda.set_device(local_rank)
    multi = world > 1 or args.dist_path
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # RCCL prints its version banner on stdout when the communicator is created: send it to stderr, stdout carries
        # the one JSON line only
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            if dist.get_world_size() != args.gpus:
                raise SystemExit("bench.py: process group has %d ranks, --gpus says %d" % (dist.get_world_size(), args.gpus))
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    from evo_amd import EvoFst
    def tune(x):
        x.set_option("pairs_pass", args.pairs_pass)
        if args.gram_denominator:
            x.set_option("gram_denominator", args.gram_denominator)
        x.set_option("chunk_tiles", args.chunk_tiles)
        x.set_option("stream_blocks", args.stream_blocks)
        x.set_option("stream_buffers", args.stream_buffers)
        x.set_option("stream_sets", args.stream_sets)
        x.set_option("stream_producers", int(args.stream_waves.split(",")[0]))
        x.set_option("stream_consumers", int(args.stream_waves.split(",")[1]))
        return x

    e = tune(EvoFst(local_rank))
    S, N, P = cfg["sites"], cfg["samples"], cfg["pops"]
    sp = (np.arange(N) % P).astype(np.int32)
    e.set_layout(sp, P)
    if cfg["mode"] == "pairs":
        e.set_pairs()
        units_per_site = e.n_pairs
    elif cfg["mode"] == "gram":
        e.set_gram()
        units_per_site = e.n_pairs
    else:
        e.set_against_all(np.arange(P, dtype=np.int32), np.full(P, N // P, np.int64))
        units_per_site = P
    if cfg["mode"] == "gram":
        if args.gram_bp > 0 and cfg.get("bp"):
            cfg = dict(cfg, bp=args.gram_bp, label=cfg["label"] + " [EXPERIMENT: %d bp windows]" % args.gram_bp)
        if cfg.get("bp"):
            # physical windows for configs[4].  The windows are cut from the positions that travel with the shard
            # (synthetic: one site per 100 bp, evofst_synth), a window = the sites of one block of cfg["bp"] bases
            # (fixedWindowStart .. fixedWindowStart + physicalWindowSize, process_vcf_fst.cpp:589-629)
            e.reserve(S)
            e.synth(S, seed=20261110, missing_rate=cfg["miss"], first_site_global=rank * S)
            pos = e.positions().astype(np.int64)
            blk = (pos - 1) // cfg["bp"]
            ws = np.concatenate(([0], np.flatnonzero(np.diff(blk)) + 1, [S])).astype(np.int64)
        else:
            ws = np.arange(0, S + cfg["size"], cfg["size"], dtype=np.int64).clip(max=S)
        e.set_site_windows(ws)
        n_win = len(ws) - 1
        gram_batch = max(1, min(n_win, int(args.gram_gb * 1e9 // (e.n_pairs * 32))))   # <= that many GB of results resident
    else:
        e.set_windows(cfg["size"], cfg["step"])
    e.reserve(S)
    e.synth(S, seed=20261110, missing_rate=cfg["miss"], first_site_global=rank * S)
    e.sync()
    # The timed loop rotates over several resident batches (contexts with different sites of the same genome) so that no
    # step finds its input in the 256 MiB Infinity Cache: configs[1]'s 112 MB batch alone would stay there and its
    # "HBM" rate would be a cache rate.  One stream for all of them: the steps run one after the other.
    tiles_bytes = e.tiles_bytes(S)
    n_batches = args.batches if args.batches > 0 else int(min(8, max(1, -(-640 * 2**20 // tiles_bytes))))
    if args.batches <= 0 and args.streams > 1:
        n_batches = max(n_batches, 2)          # two batches at least, so that one's counting overlaps the other's walk
    if cfg["mode"] == "gram" or (multi and cfg["mode"] != "pairs"):
        n_batches = 1
    es = [e]
    if n_batches > 1:
        streams = [torch.cuda.Stream() for _ in range(max(1, min(args.streams, n_batches)))]
        shared = streams[0]
        e.set_stream(shared.cuda_stream)
        for b in range(1, n_batches):
            eb = tune(EvoFst(local_rank))
            eb.set_layout(sp, P)
            if cfg["mode"] == "pairs":
                eb.set_pairs()
            else:
                eb.set_against_all(np.arange(P, dtype=np.int32), np.full(P, N // P, np.int64))
            eb.set_windows(cfg["size"], cfg["step"])
            eb.set_stream(streams[b % len(streams)].cuda_stream)
            eb.reserve(S)
            eb.synth(S, seed=20261110, missing_rate=cfg["miss"], first_site_global=(rank * n_batches + b) * S)
            eb.sync()
            es.append(eb)
    turn = [0]
    # No event records inside the timed region: eight hipEventRecord per step cost 0.030 ms of a 0.107 ms step on
    # cfg2 (measured with EVOFST_BENCH_EVENTS=1).  The per-kernel durations come from the same launches repeated
    # right after the timed region (below), on the library's stream; profiles/ holds the rocprofv3 view of them.
    e.set_profiling(os.environ.get("EVOFST_BENCH_EVENTS") is not None)
    from evo_amd import shard
    ag = shard.torch_all_gather(device="cuda") if multi else None
    if multi and cfg["mode"] == "pairs":
        # a pass stays on one stream: kernels, device-to-device copies and the all-gather (RCCL; every rank issues the
        # passes, hence the collectives, in the same order).  The resident batches take the streams in turn, as in the
        # one-rank run, so one batch's exchange and statistics overlap the next one's counting.
        if n_batches == 1:
            streams = [torch.cuda.Stream()]
            e.set_stream(streams[0].cuda_stream)
        pass_streams = [streams[b % len(streams)] for b in range(n_batches)]
        nq = e.n_pairs
        # one message per rank and pass: [included-site totals of this pass (nq x i64) | carries of the pass before
        # (nq x 4 f64, bit patterns)].  The totals are needed before K2; the carries only by the host after the
        # pass, so they ride with the batch's next pass and one more gather per batch after the last pass delivers
        # the rest.
        msgs = [shard.PairsPassMessage(nq, world, rank, "cuda") for _ in range(n_batches)]
        msg = msgs[0]
        for st_ in set(pass_streams):
            st_.wait_stream(torch.cuda.current_stream())     # the message tensors were filled on torch's current stream

    e_prof = [False]

    gram_ms, popsums_ms = [], []

    def step():
        if cfg["mode"] == "gram":
            # every rank works on its own shard; windows are cut on shard boundaries, so there is no exchange
            e.run_counts()
            for w0 in range(0, n_win, gram_batch):
                e.run_gram(w0, min(gram_batch, n_win - w0))
                if e_prof[0]:
                    e.sync(); gram_ms.append(e.kernel_ms(2)); popsums_ms.append(e.kernel_ms(3))
        elif not multi:
            es[turn[0] % n_batches].run()
            turn[0] += 1
        elif cfg["mode"] == "global":
            # {used, kept} totals and the last SIZE kept sites of every shard (all-gathers over RCCL)
            shard.compute_sharded_global(e, ag, rank, cfg["size"], cfg["step"])
        else:
            # the two window-boundary exchanges of a pass (evo_amd/shard.py): included-site totals after K1
            # (all-gather over RCCL) -> base offsets; carries behind the last complete segment after K2, which
            # travel in the next pass's message (one collective per pass instead of two).
            # Nothing in this step waits on the host; what the gathered carries are used for (completing the
            # windows that straddle the left edge, evo_amd.shard.carry_in) is host bookkeeping after the pass.
            b = turn[0] % n_batches
            turn[0] += 1
            eb_, mb_ = es[b], msgs[b]
            with torch.cuda.stream(pass_streams[b]):
                eb_.run_counts()
                eb_.copy_included_totals(mb_.totals_ptr())
                mb_.gather(dist)
                eb_.set_base_offsets_gathered(mb_.recv_ptr(), mb_.row, rank)     # sum of the earlier ranks' totals
                eb_.run_stats()
                eb_.copy_boundary(mb_.carry_ptr())

    def flush():
        # the carries of the last pass (multi-rank pairs mode): one more gather, inside the timed region
        if multi and cfg["mode"] == "pairs":
            for b in range(n_batches):
                with torch.cuda.stream(pass_streams[b]):
                    msgs[b].gather(dist)

    def fence():
        for x in es:
            x.sync()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    kms = np.zeros((args.steps, 5))

    def run_steps():
        # EXACTLY --steps steps between two fences (barrier + synchronize)
        fence()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step()
        flush()
        fence()
        return time.perf_counter() - t0

    def agree(n):
        if not multi:
            return n
        t = torch.tensor([n], dtype=torch.int64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return int(t.item())
    # A region of 20 cfg2 steps is 1.5 ms long, too short to be stable: the region is repeated until 50 ms have been
    # timed, every repetition is --steps steps, and the line carries the median (min and max beside it).
    dts = timed_reps(run_steps, agree=agree)
    # per-kernel durations from HIP events on the library's stream (last step; then a few more
    # profiled steps outside the timed region to average)
    reps = min(args.steps, 20) if cfg["mode"] != "gram" else 1
    e.set_profiling(True)
    e_prof[0] = True
    for x in es:
        x.set_profiling(True)
    for i in range(reps):
        cur = es[turn[0] % n_batches] if cfg["mode"] != "gram" and (not multi or cfg["mode"] == "pairs") else e
        step()
        cur.sync()
        for k in range(5):
            try:
                kms[i, k] = cur.kernel_ms(k)
            except Exception:
                kms[i, k] = 0.0
    kavg = kms[:reps].mean(axis=0)
    # A short dominant kernel (cfg2's K1: 37 us) is measured a second way: an event pair around ONE launch adds about 3 us
    # of its own (the events above say 40 us where rocprofv3's timestamps say 36.6), so K1 is also launched back to
    # back, four rounds over the resident batches on one stream, between one pair of events -- the gaps between the
    # launches count, the events' own time does not.  The batches then run a whole pass again (their results are read below).
    k1_b2b_ms = 0.0
    if (not multi and cfg["mode"] == "pairs" and n_batches > 1 and e.last_pairs_pass() == 1 and kavg[0] > 0
            and kavg[0] < 0.5 and kavg[0] >= kavg[2]):
        s0 = streams[0]
        torch.cuda.synchronize()
        for x in es:
            x.set_profiling(False); x.set_stream(s0.cuda_stream)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        rounds = 4
        with torch.cuda.stream(s0):
            for x in es:
                x.run_counts()                      # (warm: the first launch after a change of stream)
            ev0.record(s0)
            for i in range(rounds * n_batches):
                es[i % n_batches].run_counts()
            ev1.record(s0)
        ev1.synchronize()
        k1_b2b_ms = ev0.elapsed_time(ev1) / (rounds * n_batches)
        for b_, x in enumerate(es):
            x.set_stream(streams[b_ % len(streams)].cuda_stream)
            x.set_profiling(True)
            x.run()
        for x in es:
            x.sync()
    flush()          # every rank: the carries of the last profiled pass, for the exchange check below
    fence()
    if multi:
        t = torch.tensor(dts, dtype=torch.float64, device="cuda")     # per repetition: the slowest rank's time
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dts = [float(x) for x in t.cpu()]
    dt = float(np.median(dts))
    ms_per_step = dt / args.steps * 1e3
    evals = float(S) * units_per_site * world
    value = evals / (dt / args.steps)

    # roofline of the dominant kernel (algorithmic bytes)
    cb = e.geometry().count_bytes
    k1_bytes = S * (N / 2.0) + S * P * cb          # 2 bits per haplotype read + packed counts written
    if cfg["mode"] == "gram":
        k2_bytes = 0.0
    elif cfg["mode"] == "pairs":
        g = np.gcd(cfg["size"], cfg["step"]) if cfg["size"] else 10**9
        # data without missing calls: every chunk is clean and holds one byte per count (K1Params::compact)
        cbw = 1 if cb == 2 and not cfg.get("miss") and e.n_pairs <= 256 else cb
        k1_bytes += S * P * (cbw - cb) + S / 64.0 * (2 * P + 2) * 8.0         # zero / fixed / irregular / skip masks written
        k2_bytes = S * P * cbw + S / 64.0 * (2 * P + 2) * 8.0 + (S / g) * e.n_pairs * 28.0   # counts + masks read, segment rows written
    else:
        k2_bytes = S * P * cb + S * 8.0 + (S / cfg["step"]) * P * 16.0
    names = ["k1_count_kernel", "scan_kernel",
             {"pairs": "walk(clean_walk_kernel + fused_pairs_kernel on K1 counts)", "gram": "gram_kernel"}.get(
                 cfg["mode"], "global_tile_kernel" if e.last_global_walk() == 2 else "global_window_kernel"), "stitch_kernel",
             "fused_pairs_kernel"]
    if cfg["mode"] == "gram":
        kavg[2] = float(np.sum(gram_ms))          # the gram_kernel launches of every batch of one step
        kavg[3] = float(np.sum(popsums_ms))       # and the launches of the per-population window sums before them
        names[3] = "gram_popsums_kernel"
    fused = cfg["mode"] == "pairs" and not multi and kavg[4] > 0
    if fused:
        names[4] = "stream_pairs_kernel" if e.last_pairs_pass() == 2 else "fused_pairs_kernel"
        walk_ms = kavg[2] if e.last_pairs_pass() == 2 else 0.0       # the general walk over the chunks the stream pass marked
        names[2] = "walk of the chunks the stream pass marked (fused_pairs_kernel on spilled counts)"
        # one kernel from the tiles to the segment rows: 2 bits per haplotype read, one 28-byte row per segment and pair written
        dom = 4
        kavg[0] = kavg[1] = 0.0                   # stale events of a K1 launch behind a getter, if any
        kavg[2] = walk_ms
        dom_bytes = S * (N / 2.0) + (S / g) * e.n_pairs * 28.0
    else:
        dom = 0 if kavg[0] >= kavg[2] else 2
        dom_bytes = k1_bytes if dom == 0 else k2_bytes
    dom_ms = k1_b2b_ms if dom == 0 and k1_b2b_ms > 0 else kavg[dom]
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    # HBM bytes per launch of that kernel from the committed rocprofv3 PMC passes (FETCH_SIZE x 2 on gfx950 +
    # WRITE_SIZE; profiles/README.md says how they were collected) -- null for a configuration not profiled
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r03_traffic.json")) as f:
            tr = json.load(f).get(args.config, {}).get(names[dom].split("(")[0])
        if tr:
            traffic = tr["hbm_read_bytes"] + tr["hbm_write_bytes"]
    except (OSError, ValueError, KeyError):
        traffic = None
    pipeline_bytes = S * (N / 2.0 + 4.0)      # packed genotypes + position per site
    roofline = {"bound": "hbm", "kernel": names[dom], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "pipeline_GBps": pipeline_bytes * world / (dt / args.steps) / 1e9,
                "kernels_ms": {names[k]: float(kavg[k]) for k in range(5) if kavg[k] > 0},
                "k1_GBps": k1_bytes / (kavg[0] * 1e-3) / 1e9 if kavg[0] > 0 else 0.0,
                "algorithmic_bytes": dom_bytes, "kernel_ms": float(dom_ms),
                "kernel_ms_how": ("%d launches back to back on one stream between one pair of HIP events" % (4 * n_batches)) if dom == 0 and k1_b2b_ms > 0
                                 else "a pair of HIP events around every launch"}

    if cfg["mode"] == "gram" and kavg[2] > 0:
        # SURVEY 8d counts 2 S flops per pair and contraction.  Of the contractions behind a window's sums only the two of
        # the Hudson denominator (p q^T, q p^T) need doubles and run on the FP64 matrix cores; the integer ones (c c^T,
        # the fixed-with-missing term, the site filter's exclusions) run on the int8 cores (exact in 32 bits).  "frac" is
        # the FP64 MFMA rate on the flops of those two over the pairs (no padding of P to 64, no diagonal blocks);
        # issued_* counts the padded 64 x 64 blocks on or above the diagonal.  What bounds the kernel at 100-site windows
        # is the table of results it writes (hbm_write_GBps), not the matrix cores.
        Pp = 64 * ((P + 63) // 64)
        nblk = Pp // 64
        n_f64 = 2                                # SURVEY 8d's algorithm: the two contractions of the denominator
        # what ran: one FP64 contraction (centred frequencies, u u^T) + one more int8 one where the library takes that form
        ran_f64 = e.last_gram_streams() or 2
        n_i8 = (6 if cfg["miss"] > 0 else 4) + (1 if ran_f64 == 1 and cfg["miss"] > 0 else 0)
        t_s = kavg[2] * 1e-3
        useful = 2.0 * S * e.n_pairs * n_f64
        issued = 2.0 * S * (64 * 64) * (nblk * (nblk + 1) / 2) * n_f64
        # 8-byte entries wherever the integer contractions run on the int8 cores (16-bit packed counts, <= 127 chromosomes a
        # population; GramParams::fmt), 16-byte ones otherwise.  (Upper bound: waves below the diagonal write nothing.)
        compact = ran_f64 == 1 or (cb == 2 and 2 * (N // P + 1) <= 127)
        written = float(n_win) * (nblk * (nblk + 1) / 2) * 4096 * (16.0 if compact else 32.0)
        ach_u = useful / t_s / 1e12
        roofline.update({"bound": "mfma", "kernel": "gram_kernel", "achieved": ach_u, "peak": F64_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": ach_u / F64_MFMA_PEAK_TFLOPS, "algorithmic_flops": useful,
                         "issued_flops": issued, "issued_TFLOPs": issued / t_s / 1e12,
                         "issued_frac": issued / t_s / 1e12 / F64_MFMA_PEAK_TFLOPS,
                         "streams": {"f64_mfma": ran_f64, "int8_mfma": n_i8},
                         "executed_f64_TFLOPs": useful / n_f64 * ran_f64 / t_s / 1e12,
                         "executed_f64_frac": useful / n_f64 * ran_f64 / t_s / 1e12 / F64_MFMA_PEAK_TFLOPS,
                         "issued_f64_frac": issued / n_f64 * ran_f64 / t_s / 1e12 / F64_MFMA_PEAK_TFLOPS,
                         "all_contractions_as_f64_TFLOPs": 2.0 * S * e.n_pairs * (5 if cfg["miss"] > 0 else 3) / t_s / 1e12,
                         "launches": len(gram_ms), "launch_ms": float(np.mean(gram_ms)), "windows_per_launch": gram_batch,
                         "hbm_write_bytes": written, "hbm_write_GBps": written / t_s / 1e9,
                         "hbm_write_frac": written / t_s / 1e9 / HBM_PEAK_GBS, "traffic": None})
    out = {
        "metric": "site x pop-pair Fst+dxy evaluations/sec",
        "value": value, "unit": "site*pair/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "timed_reps": len(dts), "ms_per_step_min": min(dts) / args.steps * 1e3,
        "ms_per_step_max": max(dts) / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": cfg["label"], "sites_per_gpu": S, "samples": N, "pops": P,
                   "units_per_site": units_per_site, "sharding": "contiguous site shards, one per GPU",
                   "resident_batches": n_batches, "streams": min(args.streams, n_batches) if n_batches > 1 else 1,
                   "rccl_ranks": dist.get_world_size() if multi else 1},
        "roofline": roofline,
    }
    if rank == 0 and world == 1 and args.cpu_sites != 0:
        # the oracle does ~3e7 evaluations/s on one core: 6e8 evaluations are ~20 s; the whole cfg2 workload is less
        cs = args.cpu_sites if args.cpu_sites > 0 else min(S, max(2_000, int(6e8 / units_per_site)))
        v, cores, secs, v1, secs1 = cpu_baseline(e, cfg, cs)
        out["cpu_baseline"] = {"value": v, "unit": "site*pair/s", "cores": cores, "kind": "port",
                               "one_core": {"value": v1, "cores": 1, "seconds": secs1},
                               "sample": "first %d sites of the same workload, oracle/evo_oracle.c closed-form restatement on "
                                         "allele codes (counts + per-site stats + windows): %.1f s on 1 core, %.1f s with the "
                                         "sites cut into %d ranges on %d threads (all host cores of this process's share)"
                                         % (cs, secs1, secs, cores, cores)}
    if multi and cfg["mode"] == "pairs" and rank == 0:
        # sanity of the device-side exchange: the gathered carries equal what the host getter returns (reported, and
        # said loudly on stderr, rather than raised: the timing above is valid either way)
        try:
            own = msg.gathered_carries()[rank].cpu().numpy()
            ok = bool(np.array_equal(own, e.boundary(), equal_nan=True))
        except Exception as ex:  # noqa: BLE001
            ok = False
            print("bench.py: exchange check could not run: %r" % (ex,), file=sys.stderr)
        out["config"]["exchange_check"] = ok
        if not ok:
            print("bench.py: device-side boundary copy differs from evofst_get_boundary", file=sys.stderr)
    if rank == 0 and world == 1 and not args.no_hbm_leg and args.config == "cfg2" and args.sites == 0:
        # cfg2 is small; the same pass on one GPU's share of configs[3] (6.25 GB of tiles per step) is the HBM measurement
        for x in es[1:]:
            x.close()
        del es[1:]
        out["roofline_hbm"] = hbm_leg(local_rank, tune=tune)
    if multi and not args.no_hbm_leg and args.config == "cfg2" and args.sites == 0:
        # the same multi-rank pass on north_star's 8-GPU pairs configuration (6.25 M sites per rank): the curve that matters
        for x in es[1:]:
            x.close()
        del es[1:]
        sh = scale_hbm_leg(local_rank, rank, world, tune)
        if rank == 0:
            out["scale_hbm"] = sh
    if rank == 0:
        print(json.dumps(out))
    for x in es[1:]:
        x.close()
    e.close()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
