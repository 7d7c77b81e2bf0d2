#!/usr/bin/env python3
"""bench.py -- polished Mbp/s of the GoldPolish per-batch k-mer correction path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched
under torch.distributed.run with one rank per GPU.  A step = one pass of the whole hot path
(filter build -> ntEdit cascade + guard -> mask -> Sealer -> upper) over the rank's batches,
measured as SURVEY.md 8d defines the metric: from reads + contigs resident in HOST memory to the
polished bytes in HOST memory at rank 0.  The steps stream through gp_pipeline_*: while step i is
polished, the inputs of step i + 1 are uploaded (pinned staging, DMA on a second stream), so every
timed step contains one full upload (H2D over PCIe), the path, and the download (D2H) of the
polished bytes; at N > 1 it ends with the gather of the polished sequences on rank 0 (RCCL, from
HBM) and their copy to host memory there.  W untimed warm-up steps, then exactly K timed steps
between barrier + synchronize; max over ranks; rank 0 prints ONE JSON line.  `hbm_resident`
reports, beside it, the same path with the inputs already in HBM and the output left there
(rounds 1-2's headline).

Workload at N = 1 = the configuration BASELINE.json's metric is quoted on (full k cascade
{32,28,24,20}) that fits one GPU: configs[2], a synthetic 100 Mbp assembly + 40x ONT-like reads
(6 % error), bsize = 1.  The assembly is 4000 contigs of 25 kbp: the regime GoldPolish's fixed
10 MiB / 512 KiB filters are sized for (the reference's own fixture,
tests/goldrush_test_golden_path.fa, holds contigs of 2-43 kbp); substitutions, indels and gap
fills really happen there.  At N > 1 the workload is BASELINE.json configs[3] scaled to the node:
ONE assembly of N x 375 Mbp (3 Gbp at N = 8) in contigs of 5-100 kbp + 30x reads, its batches
dealt to the ranks by gp_partition_batches, every rank generating and polishing
its share (weak scaling: 375 Mbp per GPU).  `aux` reports configs[1] beside the N = 1 line (one
10 Mbp contig, k = 32 only: its filters saturate, so it measures the filter build alone).
--workload picks another one (see WORKLOADS).
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
RANDOM_RMW_CEILING_GPS = 20.7  # random 1-byte load + store pairs per ns over 4 GiB (19.9 over 16 GiB, 24.2 over 1 GiB), measured (tools/probes/random_rmw_probe.hip)
METRIC = "polished Mbp/sec (whole node), k={32,28,24,20}, 1/2/4/8 MI355X"

WORKLOADS = {
    # name: (contig lengths, coverage, k list, description)
    "config3": ([25_000] * 4000, 40, [32, 28, 24, 20],
                "BASELINE.json configs[2]: synthetic 100 Mbp assembly (4000 x 25 kbp contigs, one per "
                "batch) + 40x ONT-error reads per GPU, k=32,28,24,20, bsize=1"),
    "config2": ([10_000_000], 30, [32],
                "BASELINE.json configs[1]: one synthetic 10 Mbp contig + 30x ONT-error reads per GPU, "
                "k=32 only, bsize=1"),
    "goldenpath": ([25_000] * 400, 30, [32, 28, 24, 20],
                   "10 Mbp as 400 x 25 kbp contigs + 30x reads, k=32,28,24,20"),
    "tiny": ([200_000], 30, [32], "one 200 kbp contig + 30x reads, k=32"),
    # one rank's share of BASELINE.json configs[3] (3 Gbp over 8 GPUs) in GoldPolish's normal
    # batch regime: more than 2^32 read positions and 60 000 (batch, k) pairs in one job
    "config4share": ([25_000] * 15000, 30, [32, 28, 24, 20],
                     "a rank's share of BASELINE.json configs[3]: 375 Mbp (15000 x 25 kbp contigs, one per "
                     "batch) + 30x ONT-error reads, k=32,28,24,20, bsize=1"),
}


def fixture_like_lengths(total_bp, seed):
    """contig lengths with the shape of the reference's fixture (tests/goldrush_test_golden_path.fa:
    152 contigs, 2 kbp .. 43 kbp, median 7 kbp): log-normal, median 7 kbp, sigma 0.75, clipped"""
    rng = np.random.default_rng(seed)
    out, got = [], 0
    while got < total_bp:
        n = int(np.clip(rng.lognormal(np.log(7000.0), 0.75), 2000, 45000))
        out.append(n)
        got += n
    return out


def node_share(world, rank, seed=4):
    """BASELINE.json configs[3] scaled to the node: ONE assembly of world x 15000 contigs (log-normal
    around 25 kbp, 5-100 kbp: 375 Mbp per GPU, 3 Gbp at 8) + 30x reads, its batches (bsize = 1) dealt
    to the ranks by gp_partition_batches with cost = contig bases + mapped read bases.
    A contig's bases and reads depend on its index in the assembly only, so every rank generates
    exactly its own share of the same assembly, whatever the number of ranks."""
    from goldpolish_amd import core, synth
    rng = np.random.default_rng(seed)
    per_gpu = int(os.environ.get("GP_BENCH_NODE_CONTIGS", "15000"))   # (a rehearsal on one GPU takes fewer)
    lens = np.clip(rng.lognormal(np.log(22_000.0), 0.5, size=per_gpu * world), 5_000, 100_000).astype(np.int64)
    cov = 30
    cost = (lens * (1 + cov)).astype(np.uint64)
    rank_of, load = core.partition_batches(cost, world)
    mine = np.nonzero(np.asarray(rank_of) == rank)[0]
    w = synth.make_workload_chunked([int(lens[i]) for i in mine], cov, seed=seed, chunk=1,
                                    seeds=[seed * 1000003 + int(i) for i in mine])
    desc = (f"BASELINE.json configs[3] scaled to the node: one synthetic assembly of {int(lens.sum())} bp "
            f"({len(lens)} contigs of 5-100 kbp, one per batch) + 30x ONT-error reads, k=32,28,24,20, batches "
            f"dealt to {world} ranks by gp_partition_batches (cost = contig + read bases; loads "
            f"{int(min(load))}..{int(max(load))})")
    return w, [32, 28, 24, 20], desc, {"assembly_bp": int(lens.sum()), "assembly_contigs": int(len(lens)),
                                       "rank_contigs": int(len(mine))}


def survey_lengths(total_bp, mean_bp, seed):
    """contig lengths ~ log-normal with the given mean (sigma 0.6), clipped to [mean/8, 6 mean], until total_bp"""
    rng = np.random.default_rng(seed)
    sigma = 0.6
    mu = np.log(mean_bp) - sigma ** 2 / 2
    out, got = [], 0
    while got < total_bp:
        n = int(np.clip(rng.lognormal(mu, sigma), mean_bp / 8, 6 * mean_bp))
        n = min(n, total_bp - got) if total_bp - got > mean_bp / 8 else n
        out.append(n)
        got += n
    return out


def make_workload(kind, seed):
    from goldpolish_amd import synth
    if kind == "survey3":   # SURVEY 8d config 3 in its own shape: the long-contig regime (filters fill up)
        lens, cov, ks = survey_lengths(100_000_000, 2_000_000, 3), 40, [32, 28, 24, 20]
        desc = ("SURVEY 8d config 3's shape: 100 Mbp in %d contigs ~ log-normal(mean 2 Mbp) + 40x reads, k=32,28,24,20, "
                "bsize=1 (every batch saturates its 10 MiB / 512 KiB filters: the grid-wide schedule)" % len(lens))
        return synth.make_workload_chunked(lens, cov, seed=seed, chunk=1), ks, desc
    if kind == "survey4share":   # a rank's share of SURVEY 8d config 4: N50 10-20 Mbp
        lens, cov, ks = survey_lengths(375_000_000, 12_000_000, 4), 30, [32, 28, 24, 20]
        desc = ("a rank's share of SURVEY 8d config 4: 375 Mbp in %d contigs ~ log-normal(mean 12 Mbp) + 30x reads, "
                "k=32,28,24,20, bsize=1" % len(lens))
        return synth.make_workload_chunked(lens, cov, seed=seed, chunk=1), ks, desc
    if kind == "ragged":   # 20 Mbp of fixture-shaped contigs: thousands of short, ragged batches
        lens, cov, ks = fixture_like_lengths(20_000_000, seed), 40, [32, 28, 24, 20]
        desc = "20 Mbp in contigs shaped like the reference's fixture (2-45 kbp, median 7 kbp) + 40x reads, full cascade"
    elif kind in WORKLOADS:
        lens, cov, ks, desc = WORKLOADS[kind]
    else:
        raise SystemExit(f"unknown workload {kind}")
    if len(lens) >= 200:
        w = synth.make_workload_chunked(lens, cov, seed=seed, chunk=50)
    else:
        w = synth.make_workload(lens, cov, seed=seed)
    return w, ks, desc


def algorithmic_bytes_per_bp(n_k, hash_num, coverage):
    """stage A C*(1 + |K|*2H), stage B scan 6*|K|, mask+upper 2 B/bp."""
    return coverage * (1 + n_k * 2 * hash_num) + 6 * n_k + 2


def pmc_traffic(workload, kernel):
    """HBM-side bytes per launch of `kernel` from the COMMITTED rocprofv3 PMC passes of this
    workload (FETCH_SIZE and WRITE_SIZE need separate runs of the command, tools/make_profiles.sh;
    nothing here is measured by this process).  Newest round first."""
    for f in sorted((ROOT / "profiles").glob(f"r*_{workload}_pmc_traffic.json"), reverse=True):
        t = json.loads(f.read_text()).get("gp::" + kernel)
        if t and "FETCH_SIZE" in t and "WRITE_SIZE" in t:
            per_launch = (t["FETCH_SIZE"]["per_launch_KB"] + t["WRITE_SIZE"]["per_launch_KB"]) * 1024
            return int(per_launch), f"profiles/{f.name}"
    return None, None


def pmc_traffic_census(workload):
    """the same for the three census kernels in front of a cbf_local launch, summed per cbf_local launch.
    census_mark and census_pack read wide, streaming: their FETCH_SIZE is the guide's x2 case, which
    tools/collect_profiles.py writes as FETCH_SIZE_corrected; that is what is summed where it is there."""
    for f in sorted((ROOT / "profiles").glob(f"r*_{workload}_pmc_traffic.json"), reverse=True):
        t = json.loads(f.read_text())
        loc = t.get("gp::cbf_local")
        if not loc or "FETCH_SIZE" not in loc:
            continue
        kb = 0.0
        for name in ("gp::census_scatter", "gp::census_mark", "gp::census_pack"):
            e = t.get(name, {})
            kb += e.get("FETCH_SIZE_corrected", e.get("FETCH_SIZE", {})).get("sum_KB", 0.0)
            kb += e.get("WRITE_SIZE", {}).get("sum_KB", 0.0)
        return int(kb * 1024 / max(loc["FETCH_SIZE"]["launches"], 1))
    return None


def gather_to_rank0(lens, payload, rank, world):
    """Variable-length gather of the polished records on rank 0 (the path's only exchange).
    `lens` (int64) and `payload` (uint8) are tensors on the backend's device: HBM under
    nccl/RCCL, host memory under gloo.  Rank 0 returns [(lens, payload)] per rank."""
    import torch
    import torch.distributed as dist
    meta = torch.tensor([payload.numel(), lens.numel()], dtype=torch.int64, device=lens.device)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    # (nothing of zero length is sent or received: a rank with no batches, or whose records were all
    # dropped, takes part in the all-gather of sizes only -- both sides know that from `metas`)
    if rank == 0:
        out = [(lens, payload)]
        reqs = []
        for r in range(1, world):
            nb, nl = int(metas[r][0]), int(metas[r][1])
            b = torch.empty(nb, dtype=torch.uint8, device=lens.device)
            l = torch.empty(nl, dtype=torch.int64, device=lens.device)
            if nl:
                reqs.append(dist.irecv(l, src=r))
            if nb:
                reqs.append(dist.irecv(b, src=r))
            out.append((l, b))
        for q in reqs:
            q.wait()
        return out
    reqs = []
    if lens.numel():
        reqs.append(dist.isend(lens, dst=0))
    if payload.numel():
        reqs.append(dist.isend(payload, dst=0))
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


def input_order(rank_of, per_rank_records):
    """The gathered records back in the assembly's order (the ordered collection of
    scripts/goldpolish-reaper:51-73): batch b went to rank rank_of[b], and every rank polishes and
    sends its batches in ascending batch order, so rank r's i-th record is the i-th batch with
    rank_of == r.  per_rank_records[r] = list of records of rank r."""
    nxt = [0] * len(per_rank_records)
    out = []
    for r in rank_of:
        r = int(r)
        out.append(per_rank_records[r][nxt[r]])
        nxt[r] += 1
    assert all(n == len(x) for n, x in zip(nxt, per_rank_records)), "records and partition do not match"
    return out


def host_cores():
    """(hardware threads this process may use, physical cores of the box, hardware threads of the box)"""
    seen = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    phys = set()
    try:
        pid = cid = None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                pid = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                cid = ln.split(":")[1].strip()
            elif not ln.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    return seen, (len(phys) or None), (os.cpu_count() or seen)


def cpu_baseline(w, ks, desc, gpu_out=None, sample=None, threads=None, label="a GPU's share of the host"):
    """The oracle (a C port of the reference path, oracle/) timed on this box's host cores on
    a bounded sample of the SAME workload, batches in parallel -- one OpenMP task per batch, each
    batch serial, which is how the reference spreads batches over threads
    (goldpolish_targeted_bfs.cpp:189-246, scripts/goldpolish:496-514).  The reference itself
    cannot be built on this box (btllib, boost); its per-batch process / Python / file overhead is
    NOT in this number.  `sample`: batch indices (default: the first 30 per thread); `threads`:
    default a GPU's share of the node, 16 (GP_BENCH_CPU_THREADS overrides)."""
    sys.path.insert(0, str(ROOT / "tests"))
    import subprocess
    import oracle_lib
    so = ROOT / "oracle" / "libgpo.so"
    if not so.exists():
        subprocess.check_call(["make", "-C", str(ROOT / "oracle"), "libgpo.so"])
    O = oracle_lib.Oracle(str(so))
    # A GPU box hands one GPU's job 16 host cores (the node shows all of them)
    seen, physical, logical = host_cores()
    cores = threads or int(os.environ.get("GP_BENCH_CPU_THREADS", "0")) or min(seen, 16)
    # about 15 s: a 25 kbp batch of the full cascade costs one core about half a second, a
    # single long contig (one batch: no batch parallelism to be had) is taken whole
    if sample is None:
        n = w.n_contigs if w.n_contigs <= 4 else min(w.n_contigs, 30 * cores)
        sample = list(range(n))
    n = len(sample)
    names = [w.names[i] for i in sample]
    seqs = [w.contig(i) for i in sample]
    reads = [w.reads_of(i) for i in sample]
    thr = [int(w.kmer_threshold[i]) for i in sample]
    params = O.batch_params(k_values=tuple(ks))
    t = time.perf_counter()
    outs, used = O.polish_batches(names, seqs, reads, thr, list(range(n + 1)), params, n_threads=cores)
    dt = time.perf_counter() - t
    bp = sum(len(s) for s in seqs)
    parity = None
    if gpu_out is not None:
        # the oracle's output for the sample is there anyway: the device path's polished records
        # for the same batches, from the timed job, must be the same bytes (parity at the bench's
        # full size, not only in tests/)
        bad = [i for j, i in enumerate(sample) if gpu_out[i] != outs[j]]
        if bad:
            raise SystemExit(f"bench.py: {len(bad)} of the {n} sampled batches differ from the oracle (first: {w.names[bad[0]]})")
        parity = {"batches_compared_with_the_oracle": n, "identical": True}
    which = (f"the first {n}" if sample == list(range(n)) else f"{n} (the shortest that fit the time bound)")
    return {
        **({"parity_sample": parity} if parity else {}),
        "value": round(bp / dt / 1e6, 4), "unit": "Mbp/s", "cores": int(min(used, n)), "kind": "port",
        "threads_used": int(min(used, n)), "threads_are": label,
        "host": {"hardware_threads_visible": seen, "hardware_threads_of_the_box": logical, "physical_cores_of_the_box": physical},
        "sample": f"{which} of {w.n_contigs} batches of this workload ({bp} bp + their reads), whole path, "
                  f"k={','.join(map(str, ks))}, {dt:.1f} s: the C oracle, one OpenMP task per batch on "
                  f"{used} threads",
    }


def cpu_sample_by_cost(w, cores, sec_per_mbp, budget_s):
    """batch indices for a CPU sample of about budget_s seconds on `cores` threads: the shortest
    batches first, each thread given about budget_s of work (a long contig is one serial batch:
    the sample's wall time is its longest batch)"""
    lens = np.diff(np.asarray(w.contig_off, dtype=np.int64))
    order = np.argsort(lens, kind="stable")
    pick, load = [], 0.0
    for i in order:
        cost = float(lens[i]) / 1e6 * sec_per_mbp
        if pick and (cost > budget_s or load + cost > budget_s * cores):
            break
        pick.append(int(i))
        load += cost
    return sorted(pick)


def read_selection_cost(w):
    """What read ranking (gp_rank_reads: truncated phred descending, id ascending, the first
    size_t(len * s / 1e4)) would add to a step of this workload: timed here, on one host core, over every
    contig's reads with synthetic ids and phreds.  It is outside the timed region (the workload is born
    selected); SURVEY 8d's region starts at "reads + mappings resident", so the figure is given."""
    from goldpolish_amd import core
    rng = np.random.default_rng(1)
    crb = np.asarray(w.contig_read_begin, dtype=np.int64)
    roff = np.asarray(w.read_off, dtype=np.int64)
    lens_c = np.diff(np.asarray(w.contig_off, dtype=np.int64))
    n_c = min(w.n_contigs, 400)
    t = time.perf_counter()
    n_reads = 0
    for c in range(n_c):
        a, b = int(crb[c]), int(crb[c + 1])
        ids = [b"r%08d" % i for i in range(a, b)]
        ph = rng.uniform(7, 20, b - a)
        core.rank_reads(ids, ph, np.diff(roff[a:b + 1]), int(lens_c[c]), 40.0)
        n_reads += b - a
    dt = time.perf_counter() - t
    scale = w.n_contigs / max(n_c, 1)
    return {"in_timed_region": False, "gp_rank_reads_ms_per_step_one_core": round(dt * 1e3 * scale, 1),
            "measured_on": f"{n_c} of {w.n_contigs} contigs ({n_reads} reads) through the Python binding, scaled"}


def dry_run(world):
    """Per rank of `bench.py --gpus world`: contigs, bases, and the bytes it will hold -- host pageable
    (the generated workload), host pinned (staging chunks, result buffers, rank 0's gather buffers) and
    device (up to four sets of inputs in flight, one set of work buffers) -- from the assembly's lengths alone:
    no GPU, no reads generated.  The device model is checked against what one rank's share measured
    on one MI355X (137 GiB for 375 Mbp in 15000 batches at 30x: DESIGN 6)."""
    from goldpolish_amd import core
    seed, cov, n_k = 4, 30, 4
    rng = np.random.default_rng(seed)
    per_gpu = int(os.environ.get("GP_BENCH_NODE_CONTIGS", "15000"))
    lens = np.clip(rng.lognormal(np.log(22_000.0), 0.5, size=per_gpu * world), 5_000, 100_000).astype(np.int64)
    cost = (lens * (1 + cov)).astype(np.uint64)
    rank_of, load = core.partition_batches(cost, world)
    rank_of = np.asarray(rank_of)
    ranks = []
    for r in range(world):
        mine = lens[rank_of == r]
        bp, nb = int(mine.sum()), int(mine.size)
        reads = bp * cov + bp * cov // 15000            # + one separator per read of about 15 kb
        inputs = reads + bp
        dev = (4 * inputs                                # up to four sets' inputs on the device (run_stream's DEPTH: uploading, uploaded, built, polishing)
               + nb * n_k * (512 << 10)                  # Bloom filters, results of stage A
               + min(nb * n_k, 512) * (10 << 20)         # counting-filter arena
               + (16 << 30) + (1 << 30)                  # census record slots, settled tables
               + 150 * bp                                # stage B/C per-base state (two sequence slots of 1.5 x, node lists, bitmaps, 16-byte site records, Sealer's gap tables): fitted to 59 GiB for configs[2]
               + int(7.3 * 2**30))                       # Sealer arenas (read-length contigs)
        pinned = 2 * 2 * (64 << 20) + 4 * int(bp * 1.02) + (int(1.1 * bp * (world - 1)) if r == 0 and world > 1 else 0)
        ranks.append({"rank": r, "batches": nb, "contig_bp": bp, "host_pageable_bytes": inputs + 16 * (reads // 15000 + nb),
                      "host_pinned_bytes": pinned, "device_bytes": dev})
    return {"dry_run": True, "n_gpus": world, "assembly_bp": int(lens.sum()), "batches": int(lens.size),
            "partition_load_min_max": [int(min(load)), int(max(load))], "ranks": ranks,
            "host_total_GiB": round(sum(x["host_pageable_bytes"] + x["host_pinned_bytes"] for x in ranks) / 2**30, 1),
            "device_GiB_per_rank_max": round(max(x["device_bytes"] for x in ranks) / 2**30, 1),
            "notes": "generation forks a pool of workers per rank (each holds one chunk of contigs); HBM per MI355X: 288 GB"}


class _DevView:
    """a device pointer as an object torch.as_tensor can wrap without copying"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="auto",
                    help="auto: config3 (BASELINE configs[2]) at N = 1, the node's share of configs[3] at N > 1")
    ap.add_argument("--resident-steps", type=int, default=3,
                    help="steps of the HBM-resident extra at N = 1 (0: skip it)")
    ap.add_argument("--jobs-per-gpu", type=int, default=1,
                    help="HBM-resident extra only: jobs the batches are cut into and kept in flight (gp_jobs_polish)")
    ap.add_argument("--no-aux", action="store_true")
    ap.add_argument("--aux-steps", type=int, default=3,
                    help="timed steps of the long-contig block (aux_survey3: SURVEY 8d config 3's own shape, about 6 s a step; 0: skip it)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU, no generation of reads: print what every rank of --gpus N would hold in host "
                         "(pageable + pinned) and device memory, then exit")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cache", default=os.environ.get("GP_BENCH_CACHE", ""),
                    help="directory of generated workloads (.npz): loaded when there, written when not -- a run "
                         "under rocprofv3 must not fork its generator's workers")
    ap.add_argument("--prepare-only", action="store_true", help="generate (and cache) the workloads, then exit")
    ap.add_argument("--backend", default="nccl",
                    help="gloo rehearses the multi-rank control flow on a box with fewer GPUs than ranks")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    kind = args.workload if args.workload != "auto" else ("config3" if world == 1 else "node")

    # Synthetic inputs first: the generator forks worker processes, which must happen before
    # anything in this process touches the GPU.
    from goldpolish_amd import synth

    def cached(name, make):
        f = pathlib.Path(args.cache) / f"{name}.npz" if args.cache else None
        if f and f.exists():
            w = synth.load_workload(f)
            return (w,) + tuple(json.loads((f.with_suffix(".json")).read_text()))
        out = make()
        if f:
            f.parent.mkdir(parents=True, exist_ok=True)
            synth.save_workload(out[0], f)
            f.with_suffix(".json").write_text(json.dumps(list(out[1:])))
        return out

    extra_cfg = {}
    if kind == "node":
        w_main, ks_main, desc_main, extra_cfg = cached(f"node_{world}_{rank}", lambda: node_share(world, rank))
    else:
        w_main, ks_main, desc_main = cached(f"{kind}_{2 + rank}", lambda: make_workload(kind, seed=2 + rank))
    if args.dry_run:
        print(json.dumps(dry_run(args.gpus)))
        return
    want_aux = world == 1 and rank == 0 and not args.no_aux and kind == "config3"
    aux_in = cached("config2_2", lambda: make_workload("config2", seed=2)) if want_aux else None
    # the regimes SURVEY 8d describes beside the headline's: its own long-contig shape of config 3, and
    # contigs shaped like the reference's fixture (thousands of short, ragged batches)
    aux_ragged_in = cached("ragged_2", lambda: make_workload("ragged", seed=2)) if want_aux else None
    aux_survey_in = cached("survey3_2", lambda: make_workload("survey3", seed=2)) if want_aux and args.aux_steps > 0 else None
    if args.prepare_only:
        return

    from goldpolish_amd import core   # before torch brings the HIP runtime up: core asks it for eight hardware queues
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

    KERNELS = ("cbf_first_pass", "cbf_round", "cbf_census", "cbf_ckc", "cbf_local", "ntedit_scan", "ntedit_walk", "mask",
               "sealer_bfs", "upper")

    def word_sum(a):
        """a cheap checksum of host bytes: the sum of its 64-bit words (and of the odd tail)"""
        n8 = a.size // 8 * 8
        return (int(a[:n8].view(np.uint64).sum(dtype=np.uint64)) + int(a[n8:].sum(dtype=np.uint64))) & (2**64 - 1)

    def run_stream(w, ks, steps, warmup):
        """SURVEY 8d's timed region: host memory to host memory, the steps streamed through
        gp_pipeline_* (upload of step i + 1 beside the polishing of step i).  Every step is the
        whole workload: one upload, the path, one download (and at N > 1 the gather)."""
        params = core.default_params(k_values=ks)
        bs, keep = core.batch_set_of_workload(w)
        # sets in the pipeline: one uploading, one being polished and -- for the sets gp_pipeline splits (those with
        # long contigs) -- one more in the filter build and a second one in its back half
        DEPTH = 2 if os.environ.get("GP_PIPELINE_OVERLAP") == "0" else int(os.environ.get("GP_BENCH_DEPTH", "4"))
        pipe = core.Pipeline(device_index, params, depth=DEPTH)
        sums, times = [], {"upload_ms": [], "polish_ms": [], "download_ms": []}
        last = {}
        pinned = {}

        def step(tag, timed):
            r = pipe.collect(copy=False)
            # keep the stream full: the next set is handed over before anything else, so that its upload
            # runs beside the filter build and the back half of the sets before it
            pipe.submit(bs, keep, tag + DEPTH)
            data = r["bytes"]
            if world > 1:
                # polished records leave from HBM (the pipeline's packed copy), over xGMI; rank 0
                # ends the step with every rank's bytes in host memory
                n = r["total_bytes"]
                lens_t = torch.from_numpy(r["lengths"].astype(np.int64))
                if on_gpu:
                    payload = torch.as_tensor(_DevView(r["device_ptr"], max(n, 1)), device="cuda")[:n]
                    lens_t = lens_t.cuda()
                else:   # rehearsal: the same exchange over host memory
                    payload = torch.from_numpy(data)
                got = gather_to_rank0(lens_t, payload, rank, world)
                if rank == 0:
                    # ... in pinned host buffers kept from step to step (a pageable .cpu() of 7 x 375 MB
                    # would cost more than the gather itself)
                    host = []
                    for i, (l, b) in enumerate(got[1:]):
                        if on_gpu:
                            hb = pinned.get(i)
                            if hb is None or hb.numel() < b.numel():
                                hb = pinned[i] = torch.empty(int(b.numel() * 1.1) + 4096, dtype=torch.uint8, pin_memory=True)
                            hb[:b.numel()].copy_(b, non_blocking=True)
                            host.append((l.cpu(), hb[:b.numel()]))
                        else:
                            host.append((l, b))
                    if on_gpu:
                        torch.cuda.synchronize()
                    last["gathered"] = (sum(int(l.numel()) for l, _ in host) + r["lengths"].size,
                                        sum(int(b.numel()) for _, b in host) + n)
            sums.append((r["total_bytes"], word_sum(data)))
            if timed:
                for k in times:
                    times[k].append(r[k])
            last["r"] = r

        for i in range(DEPTH):
            pipe.submit(bs, keep, i)
        for i in range(warmup):
            step(i, False)
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(warmup + i, True)
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        dt = time.perf_counter() - t0
        if dist:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        if len(set(sums)) != 1:
            raise SystemExit(f"bench.py: the polished output changed between steps: {sorted(set(sums))}")
        r = last["r"]
        kt = {}
        for name in KERNELS:
            ms, n, units = pipe.kernel_time(name)
            kt[name] = {"ms": ms, "launches": n, "units": units}
        res = {"bp_per_rank": w.total_bp, "dt": dt, "steps": steps, "counters": r["counters"],
               "coverage": float(w.read_off[-1]) / max(w.total_bp, 1), "n_k": len(ks),
               "n_contigs": w.n_contigs, "device_bytes": r["device_bytes_in_use"], "n_jobs": 1,
               "kernels": kt,
               "output_checksum": {"bytes": sums[0][0], "word_sum": sums[0][1], "steps_compared": len(sums)},
               "stream": {k: round(float(np.mean(v)), 2) for k, v in times.items()},
               "input_bytes": int(w.read_off[-1]) + int(w.contig_off[-1]),
               "gathered": last.get("gathered")}
        res["polished"] = core.split_polished(r) if rank == 0 and world == 1 else None
        # the sets still in flight (uploaded and built beside the last timed steps) are drained untimed
        for _ in range(DEPTH):
            pipe.collect(copy=False)
        pipe.close()
        return res

    def run(w, ks, steps, warmup, n_jobs):
        """the same path with the inputs already resident in HBM and the output left there (rounds 1-2's
        timed region), the rank's batches cut into n_jobs jobs of consecutive contigs"""
        params = core.default_params(k_values=ks)
        n_jobs = max(1, min(n_jobs, w.n_contigs))
        # shrinking jobs (4 : 3 : 2 : 1 ...): only the last job's back half is not hidden behind a
        # filter build, so the last job is the smallest; each build still outlasts the back half before it
        weights = np.arange(n_jobs, 0, -1, dtype=np.float64)
        cuts = [0] + [int(round(w.n_contigs * x)) for x in np.cumsum(weights) / weights.sum()]
        cuts[-1] = w.n_contigs
        for i in range(1, n_jobs):     # never an empty job, whatever the rounding did
            cuts[i] = min(max(cuts[i], cuts[i - 1] + 1), w.n_contigs - (n_jobs - i))
        ctxs = [core.Context(device_index) for _ in range(n_jobs)]
        jobs = [core.Job(ctxs[i], params, **core.slice_workload(w, cuts[i], cuts[i + 1])) for i in range(n_jobs)]

        sums = []

        def step():
            core.polish_jobs(jobs)
            # the polished records, packed back to back in one HBM buffer in input order
            lens = [j.lengths()[0] for j in jobs]
            sizes = [int(l.sum()) for l in lens]
            payload = torch.empty(max(sum(sizes), 1), dtype=torch.uint8, device="cuda")
            at = 0
            for j, n in zip(jobs, sizes):
                if n:
                    j.sequences_into_device(payload.data_ptr() + at, n)
                at += n
            # every step must give the same bytes: a cheap checksum of the packed records on the GPU
            sums.append((sum(sizes), int(payload[:sum(sizes)].sum(dtype=torch.int64).item())))

        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if len(set(sums)) != 1:
            raise SystemExit(f"bench.py: the polished output changed between steps: {sorted(set(sums))}")
        counters = {}
        for j in jobs:
            for name, v in j.counters().items():
                counters[name] = counters.get(name, 0) + v
        res = {"bp_per_rank": w.total_bp, "dt": dt, "steps": steps, "counters": counters,
               "coverage": float(w.read_off[-1]) / max(w.total_bp, 1), "n_k": len(ks),
               "n_contigs": w.n_contigs, "device_bytes": core.device_bytes_in_use(), "n_jobs": n_jobs,
               "output_checksum": {"bytes": sums[0][0], "byte_sum": sums[0][1], "steps_compared": len(sums)}}

        def kernel_times():
            kt = {n: {"ms": 0.0, "launches": 0, "units": 0} for n in KERNELS}
            for j in jobs:
                for name in KERNELS:
                    ms, n, units = j.kernel_time(name)
                    kt[name]["ms"] += ms
                    kt[name]["launches"] += n
                    kt[name]["units"] += units
            return kt
        res["kernels"] = kernel_times()          # the last timed step, kernels of several jobs side by side
        res["polished"] = None
        if n_jobs > 1:
            # every job once more, ONE AT A TIME: a launch's own duration with the chip to itself
            for j in jobs:
                j.polish(sync=True)
            res["kernels_alone"] = kernel_times()
        for j in jobs:
            j.close()
        for c in ctxs:
            c.close()
        return res

    def roofline(res, workload):
        """SURVEY.md 8d's streaming model for the stage-A kernel the step spends most time in:
        one launch covers `units` (read base, k) pairs, each priced at 1 B of sequence + H counter
        reads + H counter writes.  HIP events on the library's stream give the launch time."""
        k = res.get("kernels_alone") or res["kernels"]
        H = 4
        name = max(("cbf_local", "cbf_first_pass"), key=lambda n: k[n]["ms"])
        kk = k[name]
        out = {"bound": "hbm", "kernel": name, "peak": HBM_PEAK_GBS, "unit": "GB/s"}
        # With the census in front of it, cbf_local does its launch's units together with the three
        # census kernels (one timer, "cbf_census"): the launch is priced over both, so that `frac`
        # never credits cbf_local with work the census did.
        census_ms = k.get("cbf_census", {}).get("ms", 0.0) if name == "cbf_local" else 0.0
        if kk["launches"]:
            bytes_per_launch = (1 + 2 * H) * kk["units"] / kk["launches"]
            avg_s = (kk["ms"] + census_ms) / kk["launches"] / 1e3
            achieved = bytes_per_launch / avg_s / 1e9
        else:
            bytes_per_launch, avg_s, achieved = 0, 0.0, 0.0
        if census_ms:
            out["priced_over"] = ("cbf_local's launch plus the census kernels in front of it (census_scatter, census_mark, "
                                  "census_pack: %.3f ms per cbf_local launch)" % (census_ms / max(kk["launches"], 1)))
        traffic, src = pmc_traffic(workload, name)
        if census_ms and traffic:
            out["traffic_census_kernels"] = pmc_traffic_census(workload)   # per cbf_local launch, same source
        out.update({
            "achieved": round(achieved, 2), "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": traffic,
            "traffic_source": (src + " (committed rocprofv3 PMC passes of this workload; not collected by this run)"
                               if src else None),
            "traffic_GBps": (round(traffic / (kk["ms"] / kk["launches"] / 1e3) / 1e9, 1) if traffic and kk["launches"] and kk["ms"] else None),
            "alg_bytes_per_launch": int(bytes_per_launch),
            "avg_launch_ms": round(kk["ms"] / max(kk["launches"], 1), 4),
            "largest_by_time": max(k, key=lambda n: k[n]["ms"]),
            "whole_path_alg_GBps": round(
                algorithmic_bytes_per_bp(res["n_k"], H, res["coverage"]) * res["bp_per_rank"] * res["steps"]
                / res["dt"] / 1e9, 2),
            "kernel_ms_per_run": {n: round(v["ms"], 3) for n, v in k.items()},
        })
        if "kernels_alone" in res:
            ks_ = res["kernels"][name]
            out["measured_on"] = (f"each of the rank's {res['n_jobs']} jobs polished once more, one at a time, after the "
                                  "timed steps: a launch's duration with the chip to itself (in the timed steps "
                                  "kernels of different jobs and stages run side by side)")
            out["avg_launch_ms_in_step"] = round(ks_["ms"] / max(ks_["launches"], 1), 4)
            out["kernel_ms_in_step"] = {n: round(v["ms"], 3) for n, v in res["kernels"].items()}
        c = res["counters"]
        if k.get("cbf_census", {}).get("ms"):
            # stage A for the short batches = census of first touches + replay of the candidates
            sa_ms = k["cbf_census"]["ms"] + k["cbf_local"]["ms"]
            out["stage_a"] = {"census_ms": round(k["cbf_census"]["ms"], 3), "replay_ms": round(k["cbf_local"]["ms"], 3),
                              "operations": c["a_ops"], "replayed": c.get("a_replayed", 0),
                              "replayed_found_settled": c.get("a_settled_hits", 0),
                              "candidates_settled_by_their_kmers_count": c.get("a_counted", 0),
                              "alg_GBps": round((1 + 2 * H) * k["cbf_local"]["units"] / (sa_ms / 1e3) / 1e9, 2)}
        lv = k.get("cbf_census", {})
        if lv.get("ms") and not k["cbf_local"]["launches"]:
            # the census LEVEL BY LEVEL (long batches whose filters stay unsaturated; no cbf_local behind it): its
            # timer's units are the (read base, k) pairs it decides, priced like every stage-A kernel here
            lv_bytes = (1 + 2 * H) * lv["units"]
            out["census_level_by_level"] = {
                "ms": round(lv["ms"], 3), "operations_decided": int(lv["units"]),
                "alg_GBps": round(lv_bytes / (lv["ms"] / 1e3) / 1e9, 2),
                "frac": round(lv_bytes / (lv["ms"] / 1e3) / 1e9 / HBM_PEAK_GBS, 5),
                "note": "thr - 1 passes of census_mark_seg / census_pack over the surviving records; in gp_pipeline it runs "
                        "beside the back halves of two other sets (about 1.7 x its time alone)"}
        if name == "cbf_local" and kk["ms"]:
            # what bounds this kernel: random memory transactions, one line each -- a look-up in the
            # table of settled k-mers per replayed operation, and for those not found there H counter
            # loads (plus stores and filter look-ups, not counted here); the probe gives the chip's rate
            # for random 1-byte load + store pairs
            replayed = c.get("a_replayed") or c["a_ops"]
            hits = c.get("a_settled_hits", 0)
            touches = (replayed - hits) * H + (replayed if hits else 0)
            rate = touches / (kk["ms"] / 1e3) / 1e9
            out["random_access"] = {"random_accesses_per_step": touches,
                                    "counted": "settled-table look-ups + counter loads of the operations not found there "
                                               "(their stores and filter look-ups come on top)",
                                    "achieved_G_per_s": round(rate, 2),
                                    "measured_chip_rates_G_per_s": {"random_1_byte_load_store_pairs_over_4_GiB": RANDOM_RMW_CEILING_GPS,
                                                                    "random_loads_from_memory": "55-65"}}
        resv = c.get("a_reservations", 0)
        span_ms = k["cbf_round"]["ms"]
        if resv and span_ms:
            # the grid-wide schedule is bound by scattered device-scope atomics (27 G/s on this
            # chip whatever their width, scope or target: tools/probes/atomic_probe.hip)
            rate = resv / (span_ms / 1e3) / 1e9
            out["atomics"] = {"reservation_atomics_per_step": resv, "stage_a_ms": round(span_ms, 3),
                              "achieved_G_per_s": round(rate, 2), "measured_ceiling_G_per_s": ATOMIC_CEILING_GPS,
                              "frac": round(rate / ATOMIC_CEILING_GPS, 3)}
        return out

    main_res = run_stream(w_main, ks_main, args.steps, args.warmup)
    total_bp = main_res["bp_per_rank"] * world
    if dist:
        t = torch.tensor([main_res["bp_per_rank"]], dtype=torch.int64, device="cuda" if on_gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        total_bp = int(t.item())
    value = total_bp * args.steps / main_res["dt"] / 1e6
    line = {
        "metric": METRIC,
        "value": round(value, 3), "unit": "Mbp/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(main_res["dt"] / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
        "data": "synthetic",
        "config": {
            "workload": desc_main,
            "timed_region": "SURVEY 8d: reads + contigs resident in host memory -> polished bytes in host memory at "
                            "rank 0, steady state: every step holds one upload (H2D, beside the polishing of the step "
                            "before), the path, the download (D2H)" + (" and the gather on rank 0" if world > 1 else "") +
                            ".  NOT inside: read ranking and selection (gp_rank_reads / gp_job_create_from_files, "
                            "serve_batch's goldpolish_targeted_bfs.cpp:96-123) -- the synthetic reads are generated per "
                            "contig, already selected; what it would add per step is config.read_selection",
            "contigs_per_gpu": main_res["n_contigs"], "bp_per_gpu": main_res["bp_per_rank"], "bp_total": total_bp,
            "read_coverage": round(main_res["coverage"], 2), "k": ",".join(map(str, ks_main)),
            "parallelism": f"batches sharded over {world} GPU(s), one process per GPU; gp_pipeline (upload | polish | "
                           "collect) per GPU" + ((", gather of polished records to rank 0 (" + ("RCCL" if on_gpu else "gloo rehearsal") + ")") if world > 1 else ""),
            "device_GiB_in_use": round(main_res["device_bytes"] / 2**30, 1),
            **extra_cfg,
        },
    }
    if rank == 0:
        line["roofline"] = roofline(main_res, kind)
        line["counters"] = main_res["counters"]
        line["output_checksum"] = main_res["output_checksum"]   # identical in every warm-up and timed step
        st = main_res["stream"]
        line["stream"] = {
            **st, "input_bytes_per_step": main_res["input_bytes"],
            "upload_GBps": round(main_res["input_bytes"] / max(st["upload_ms"], 1e-9) / 1e6, 2),
            "note": "per step, means over the timed steps: upload_ms = gp_job_create of the step (pinned staging + "
                    "H2D), hidden behind the polishing of the step before when it is shorter than polish_ms + "
                    "download_ms; ms_per_step is what the caller sees between two collects",
        }
        if main_res.get("gathered"):
            line["stream"]["gathered_records_and_bytes_at_rank0"] = main_res["gathered"]
    if world == 1 and rank == 0 and args.resident_steps > 0:
        # beside it: the inputs already in HBM, the output left there (what rounds 1-2 reported as `value`)
        rs = run(w_main, ks_main, args.resident_steps, 1, max(1, args.jobs_per_gpu))
        line["hbm_resident"] = {
            "value": round(rs["bp_per_rank"] * rs["steps"] / rs["dt"] / 1e6, 3), "unit": "Mbp/s",
            "ms_per_step": round(rs["dt"] / rs["steps"] * 1e3, 3), "steps": rs["steps"],
            "note": "the same path and workload with reads + contigs already resident in HBM and the polished "
                    "records left packed in HBM: no PCIe in the timed region",
            "same_output_bytes": rs["output_checksum"]["bytes"] == main_res["output_checksum"]["bytes"],
        }
    def aux_block(w_aux, ks_aux, desc_aux, name, steps, cpu_sample, note="", warmup=1):
        """one more workload through the same timed region, with a roofline and a CPU sample of its own"""
        core.device_pool_trim()   # the blocks of the workload before are of another shape
        aux = run_stream(w_aux, ks_aux, steps, warmup)
        blk = {
            "workload": desc_aux + note,
            "value": round(aux["bp_per_rank"] * aux["steps"] / aux["dt"] / 1e6, 3),
            "unit": "Mbp/s", "ms_per_step": round(aux["dt"] / aux["steps"] * 1e3, 2), "steps": aux["steps"], "warmup": warmup,
            "stream": aux["stream"], "roofline": roofline(aux, name), "counters": aux["counters"],
            "device_GiB_in_use": round(aux["device_bytes"] / 2**30, 1),
        }
        if cpu_sample is not None and not args.no_cpu_baseline:
            blk["cpu_baseline"] = cpu_baseline(w_aux, ks_aux, desc_aux, aux.get("polished"), sample=cpu_sample)
        return blk

    if want_aux:
        w_aux, ks_aux, desc_aux = aux_in
        line["aux"] = aux_block(w_aux, ks_aux, desc_aux, "config2", max(1, min(args.steps, 5)), None,
                                " (its one pair of 10 MiB / 512 KiB filters saturates, as the reference's would: "
                                "ntEdit and Sealer find nothing to do, the step is the filter build)")
        del w_aux, aux_in
        threads = int(os.environ.get("GP_BENCH_CPU_THREADS", "0")) or min(host_cores()[0], 16)
        w_aux, ks_aux, desc_aux = aux_ragged_in
        line["aux_ragged"] = aux_block(w_aux, ks_aux, desc_aux, "ragged", max(1, min(args.steps, 5)),
                                       list(range(min(w_aux.n_contigs, 60 * threads))))
        del w_aux, aux_ragged_in
        if aux_survey_in:
            w_aux, ks_aux, desc_aux = aux_survey_in
            # the oracle needs about 28 s per Mbp of such a contig on one core (17 s for a 0.6 Mbp batch)
            line["aux_survey3"] = aux_block(w_aux, ks_aux, desc_aux, "survey3", args.aux_steps,
                                            cpu_sample_by_cost(w_aux, threads, 28.0, 25.0),
                                            " -- the regime SURVEY 8d describes for configs 3 and 4; the fixed 10 MiB / "
                                            "512 KiB filters (src/goldpolish_targeted_bfs.cpp:270-272) saturate on such contigs",
                                            # gp_pipeline keeps four such sets going (upload | filter build | two back
                                            # halves) and the first one allocates 137 GiB: timed from the third collect on
                                            warmup=3)
            del w_aux, aux_survey_in
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(w_main, ks_main, desc_main, main_res.get("polished"))
        seen, physical, logical = host_cores()
        if seen > line["cpu_baseline"]["threads_used"] and w_main.n_contigs > 4:
            # once more on every hardware thread the process may use: the whole node's CPU figure, so that
            # the 8-GPU ratio needs no extrapolation from a 16-thread share
            n_all = min(w_main.n_contigs, 2 * seen)   # (3072 batches took 256 threads 131 s: 0.59 Mbp/s, less than 16 threads' 1.32)
            line["cpu_baseline_all_cores"] = cpu_baseline(w_main, ks_main, desc_main, main_res.get("polished"),
                                                          sample=list(range(n_all)), threads=seen,
                                                          label="every hardware thread the process may use")
        line["config"]["read_selection"] = read_selection_cost(w_main)
    if rank == 0:
        print(json.dumps(line))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
