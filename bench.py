#!/usr/bin/env python3
"""bench.py -- polished Mbp/s of the GoldPolish per-batch k-mer correction path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched
under torch.distributed.run with one rank per GPU.  A step = one pass of the whole hot path
(filter build -> ntEdit cascade + guard -> mask -> Sealer -> upper) over the rank's batches,
with reads + contigs already resident in HBM; at N > 1 the step ends with the gather of the
polished sequences on rank 0 (RCCL).  W untimed warm-up steps, then exactly K timed steps
between barrier + synchronize; max over ranks; rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1]): one synthetic 10 Mbp contig (41 % GC, 1 % draft error)
+ 30x ONT-like reads (6 % error), k = 32 only, bsize = 1, per GPU (weak scaling: every rank
polishes its own 10 Mbp contig).  `--workload goldenpath` instead cuts the same 10 Mbp into
25 kbp contigs (the regime GoldPolish's fixed 10 MiB / 512 KiB filters are sized for) and runs
the full k cascade; its figures are reported under "aux" by default.
"""
import argparse
import json
import os
import pathlib
import sys
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
ATOMIC_CEILING_GPS = 27.0  # scattered device-scope atomics per ns, measured (tools/probes/atomic_probe.hip)


def make_workload(kind, seed):
    from goldpolish_amd import synth
    if kind == "config2":
        return synth.make_workload([10_000_000], 30, seed=seed), [32]
    if kind == "goldenpath":
        return synth.make_workload([25_000] * 400, 30, seed=seed), [32, 28, 24, 20]
    if kind == "config3":   # BASELINE.json configs[2]: 100 Mbp assembly, 40x reads, full cascade
        return synth.make_workload([25_000] * 4000, 40, seed=seed), [32, 28, 24, 20]
    if kind == "tiny":
        return synth.make_workload([200_000], 30, seed=seed), [32]
    raise SystemExit(f"unknown workload {kind}")


def algorithmic_bytes_per_bp(n_k, hash_num, coverage):
    """stage A C*(1 + |K|*2H), stage B scan 6*|K|, mask+upper 2 B/bp."""
    return coverage * (1 + n_k * 2 * hash_num) + 6 * n_k + 2


def pmc_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes
    (FETCH_SIZE and WRITE_SIZE collected in separate runs of this same command, see
    tools/make_profiles.sh; FETCH_SIZE calibrated 1:1 on classify_kernel's 16 B/lane stream)."""
    files = sorted((ROOT / "profiles").glob("r*_config2_pmc_traffic.json"))
    if not files:
        return None, None
    t = json.loads(files[-1].read_text()).get("gp::" + kernel)
    if not t or "FETCH_SIZE" not in t or "WRITE_SIZE" not in t:
        return None, None
    per_launch = (t["FETCH_SIZE"]["per_launch_KB"] + t["WRITE_SIZE"]["per_launch_KB"]) * 1024
    return int(per_launch), f"profiles/{files[-1].name}"


def gather_to_rank0(lens, payload, rank, world):
    """Variable-length gather of the polished records on rank 0 (the path's only exchange).
    `lens` (int64) and `payload` (uint8) are tensors on the backend's device: HBM under
    nccl/RCCL, host memory under gloo.  Rank 0 returns [(lens, payload)] per rank."""
    import torch
    import torch.distributed as dist
    meta = torch.tensor([payload.numel(), lens.numel()], dtype=torch.int64, device=lens.device)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    if rank == 0:
        out = [(lens, payload)]
        reqs = []
        for r in range(1, world):
            nb, nl = int(metas[r][0]), int(metas[r][1])
            b = torch.empty(nb, dtype=torch.uint8, device=lens.device)
            l = torch.empty(nl, dtype=torch.int64, device=lens.device)
            reqs += [dist.irecv(l, src=r), dist.irecv(b, src=r)]
            out.append((l, b))
        for q in reqs:
            q.wait()
        return out
    reqs = [dist.isend(lens, dst=0), dist.isend(payload, dst=0)]
    for q in reqs:
        q.wait()
    return None


def split_records(lens, payload):
    """(lens, payload) tensors -> list of bytes, one per record (b"" for dropped ones)."""
    l = lens.cpu().numpy()
    b = payload.cpu().numpy().tobytes()
    at, out = 0, []
    for n in l:
        out.append(b[at:at + int(n)])
        at += int(n)
    return out


def cpu_baseline(seed):
    """The oracle (a scalar C port of the reference path) timed on this box's host cores, on a
    bounded sample of the same kind of workload."""
    sys.path.insert(0, str(ROOT / "tests"))
    import subprocess
    import oracle_lib
    from goldpolish_amd import synth
    so = ROOT / "oracle" / "libgpo.so"
    if not so.exists():
        subprocess.check_call(["make", "-C", str(ROOT / "oracle"), "libgpo.so"])
    O = oracle_lib.Oracle(str(so))
    sample_bp = 10_000_000      # the whole configs[1] contig: about 20 s of one host core
    w = synth.make_workload([sample_bp], 30, seed=seed)
    params = O.batch_params(k_values=(32,))
    t = time.perf_counter()
    O.polish_batch(w.names, [w.contig(0)], [w.reads_of(0)], [int(w.kmer_threshold[0])], params)
    dt = time.perf_counter() - t
    return {
        "value": round(w.total_bp / dt / 1e6, 4), "unit": "Mbp/s", "cores": 1, "kind": "port",
        "sample": f"one {sample_bp} bp contig + 30x reads, k=32, whole path, {dt:.1f} s of the "
                  "single-threaded C oracle",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="config2")
    ap.add_argument("--no-aux", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl",
                    help="gloo rehearses the multi-rank control flow on a box with fewer GPUs than ranks")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the polishing core has no CPU fallback")
    device_index = local_rank % torch.cuda.device_count()   # == local_rank on a full node
    torch.cuda.set_device(device_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend)
    on_gpu = args.backend == "nccl"
    from goldpolish_amd import core

    def run(kind, steps, warmup, want_roofline):
        w, ks = make_workload(kind, seed=2 + rank)
        params = core.default_params(k_values=ks)
        ctx = core.Context(device_index)
        job = core.Job.from_workload(ctx, params, w)

        def step():
            job.polish(sync=True)
            # the polished records, packed back to back in one HBM buffer: what rank 0 ends a
            # step with at every N (at N > 1 the other ranks' buffers arrive next to it)
            lens, _ = job.lengths()
            payload = torch.empty(max(int(lens.sum()), 1), dtype=torch.uint8, device="cuda")
            job.sequences_into_device(payload.data_ptr(), payload.numel())
            if world > 1:
                # device-to-device: packed on the GPU, sent from HBM over xGMI
                lens_t = torch.from_numpy(lens.astype(np.int64)).cuda()
                if not on_gpu:   # rehearsal: the same exchange over host memory
                    lens_t, payload = lens_t.cpu(), payload.cpu()
                gather_to_rank0(lens_t, payload[:int(lens.sum())], rank, world)

        for _ in range(warmup):
            step()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        dt = time.perf_counter() - t0
        if dist:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        res = {"bp_per_rank": w.total_bp, "dt": dt, "steps": steps, "counters": job.counters(),
               "coverage": float(w.read_off[-1]) / max(w.total_bp, 1), "n_k": len(ks),
               "n_contigs": w.n_contigs}
        if want_roofline:
            kt = {}
            for name in ("cbf_first_pass", "cbf_round", "cbf_local", "ntedit_scan", "ntedit_walk", "mask", "sealer_bfs", "upper"):
                ms, n, units = job.kernel_time(name)
                kt[name] = {"ms": ms, "launches": n, "units": units}
            res["kernels"] = kt
        job.close()
        if want_roofline and rank == 0 and world == 1:
            # the same step with the hand-over included: host buffers -> pinned staging -> HBM
            # (gp_job_create), the path, and the polished bytes copied back to host memory.
            # Reported beside `value`, never as it.
            t0 = time.perf_counter()
            job2 = core.Job.from_workload(ctx, params, w)
            t1 = time.perf_counter()
            job2.polish(sync=True)
            job2.sequences()
            t2 = time.perf_counter()
            job2.close()
            res["pcie_inclusive"] = {"value": round(w.total_bp / (t2 - t0) / 1e6, 3), "unit": "Mbp/s",
                                     "upload_ms": round((t1 - t0) * 1e3, 2),
                                     "path_and_download_ms": round((t2 - t1) * 1e3, 2),
                                     "note": "one cold run: job creation (pinned staging + device "
                                             "allocations + PCIe upload) + path + download"}
        ctx.close()
        return res

    main_res = run(args.workload, args.steps, args.warmup, True)
    total_bp = main_res["bp_per_rank"] * world
    value = total_bp * args.steps / main_res["dt"] / 1e6
    line = {
        "metric": "polished Mbp/sec (whole node), k={32,28,24,20}, 1/2/4/8 MI355X",
        "value": round(value, 3), "unit": "Mbp/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(main_res["dt"] / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
        "data": "synthetic",
        "config": {
            "workload": ("BASELINE.json configs[1]: one synthetic 10 Mbp contig + 30x ONT-error "
                         "reads per GPU, k=32 only, bsize=1" if args.workload == "config2"
                         else args.workload),
            "contigs_per_gpu": main_res["n_contigs"], "bp_per_gpu": main_res["bp_per_rank"],
            "read_coverage": round(main_res["coverage"], 2), "k": "32" if main_res["n_k"] == 1 else "32,28,24,20",
            "parallelism": f"batches sharded over {world} GPU(s), gather to rank 0",
        },
    }
    if rank == 0:
        # roofline of the dominant kernel: stage A's streaming pass (cbf_first_pass).  One
        # launch covers `units` (read base, k) pairs; SURVEY 8d prices each at 1 B of sequence
        # + 2H B of counter traffic.
        k = main_res["kernels"]
        dom = max(k, key=lambda n: k[n]["ms"])
        fp = k["cbf_first_pass"]
        H = 4
        if fp["launches"]:
            bytes_per_launch = (1 + 2 * H) * fp["units"] / fp["launches"]
            avg_s = fp["ms"] / fp["launches"] / 1e3
            achieved = bytes_per_launch / avg_s / 1e9
        else:
            bytes_per_launch, achieved = 0, 0.0
        traffic, traffic_src = pmc_traffic("cbf_first_pass")
        line["roofline"] = {
            "bound": "hbm", "kernel": "cbf_first_pass", "achieved": round(achieved, 2),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": traffic, "traffic_source": traffic_src,
            "traffic_GBps": (round(traffic / avg_s / 1e9, 1) if traffic and fp["launches"] else None),
            "alg_bytes_per_launch": int(bytes_per_launch),
            "avg_launch_ms": round(fp["ms"] / max(fp["launches"], 1), 4),
            "largest_by_time": dom,
            "whole_path_alg_GBps": round(
                algorithmic_bytes_per_bp(main_res["n_k"], H, main_res["coverage"]) * value * 1e6 / 1e9, 2),
            "kernel_ms_per_run": {n: round(v["ms"], 3) for n, v in k.items()},
        }
        # what actually bounds the filter build: scattered device-scope atomics, which this
        # chip retires at 27 G/s whatever their width, scope or target (tools/probes/
        # atomic_probe.hip, measured; 64 M atomicMin over 2^16..2^24 slots)
        resv = main_res["counters"].get("a_reservations", 0)
        span_ms = k["cbf_round"]["ms"]
        if resv and span_ms:
            rate = resv / (span_ms / 1e3) / 1e9
            line["roofline"]["atomics"] = {
                "reservation_atomics_per_step": resv, "stage_a_ms": round(span_ms, 3),
                "achieved_G_per_s": round(rate, 2), "measured_ceiling_G_per_s": ATOMIC_CEILING_GPS,
                "frac": round(rate / ATOMIC_CEILING_GPS, 3)}
        line["counters"] = main_res["counters"]
        if "pcie_inclusive" in main_res:
            line["pcie_inclusive"] = main_res["pcie_inclusive"]
    if world == 1 and not args.no_aux and args.workload == "config2":
        aux = run("goldenpath", max(1, min(args.steps, 3)), 1, True)
        if rank == 0:
            line["aux"] = {
                "workload": "10 Mbp as 400 x 25 kbp contigs + 30x reads, k=32,28,24,20 "
                            "(filters in their design regime: edits and gap fills happen)",
                "value": round(aux["bp_per_rank"] * aux["steps"] / aux["dt"] / 1e6, 3),
                "unit": "Mbp/s", "ms_per_step": round(aux["dt"] / aux["steps"] * 1e3, 2),
                "counters": aux["counters"],
                "kernel_ms_per_run": {n: round(v["ms"], 3) for n, v in aux["kernels"].items()},
            }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(seed=2)
    if rank == 0:
        print(json.dumps(line))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
