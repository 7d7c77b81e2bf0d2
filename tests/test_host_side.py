"""CPU tests of the host side: the synthetic workload generator, the multi-rank gather of
polished sequences (gloo, world_size 2) and the oracle under AddressSanitizer."""
import os
import pathlib
import subprocess
import sys

import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]


def test_workload_generator_is_seeded_and_ranked():
    from goldpolish_amd import synth
    a = synth.make_workload([4000, 2500], 20, seed=9)
    b = synth.make_workload([4000, 2500], 20, seed=9)
    assert np.array_equal(a.read_buf, b.read_buf) and np.array_equal(a.contig_buf, b.contig_buf)
    assert a.n_contigs == 2 and a.contig_read_begin[0] == 0
    # per-contig cap size_t(len * 40 / 10000)
    for c in range(2):
        n = int(a.contig_read_begin[c + 1] - a.contig_read_begin[c])
        assert n <= int(len(a.contig(c)) * 40 / 10000)
    assert set(np.unique(a.read_buf)) <= set(b"ACGT")


_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[3])
import torch.distributed as dist
import bench
rank, world = int(sys.argv[1]), int(sys.argv[2])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[4], RANK=str(rank), WORLD_SIZE=str(world))
dist.init_process_group("gloo", rank=rank, world_size=world)
import numpy as np, torch
mine = [bytes([65 + rank]) * (10 + 7 * rank), b"", b"ACGT" * (3 + rank)]   # middle record dropped
lens = torch.tensor([len(x) for x in mine], dtype=torch.int64)
payload = torch.from_numpy(np.frombuffer(b"".join(mine), dtype=np.uint8).copy())
out = bench.gather_to_rank0(lens, payload, rank, world)
if rank == 0:
    assert len(out) == world
    for r in range(world):
        exp = [bytes([65 + r]) * (10 + 7 * r), b"", b"ACGT" * (3 + r)]
        got = bench.split_records(*out[r])
        assert got == exp, (r, got, exp)
    print("GATHER_OK")
else:
    assert out is None
dist.barrier()
dist.destroy_process_group()
'''


def test_gather_of_polished_segments_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    port = str(29500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", str(ROOT), port],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "GATHER_OK" in outs[0]


_PART_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[3])
import torch.distributed as dist
import bench
from goldpolish_amd import core
rank, world, scenario = int(sys.argv[1]), int(sys.argv[2]), sys.argv[5]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[4], RANK=str(rank), WORLD_SIZE=str(world))
dist.init_process_group("gloo", rank=rank, world_size=world)
import numpy as np, torch
# one fixed assembly of single-contig batches; every rank computes the same partition by itself
rng = np.random.default_rng(11)
n_batches = {"even": 23, "few": 5, "drop": 41, "skew": 30}[scenario]
lens = rng.integers(40, 400, size=n_batches)
if scenario == "skew":
    lens[3] = 40000          # one batch outweighs the rest: its rank gets nothing else
contigs = [bytes(rng.choice(list(b"acgt"), size=int(n)).astype(np.uint8)) for n in lens]
cost = (lens * 31).astype(np.uint64)
rank_of, load = core.partition_batches(cost, world)
def polished_of(b):          # stand-in for the GPU path
    if scenario == "drop" and rank_of[b] == world // 2:
        return b""           # every record of one rank dropped: its payload is empty
    return contigs[b].upper() if b % 7 else b""
mine = [b for b in range(len(contigs)) if rank_of[b] == rank]          # ascending: the order a rank polishes in
polished = [polished_of(b) for b in mine]
lens_t = torch.tensor([len(x) for x in polished], dtype=torch.int64)
payload = torch.from_numpy(np.frombuffer(b"".join(polished) or b"\0", dtype=np.uint8).copy())[:sum(len(x) for x in polished)]
out = bench.gather_to_rank0(lens_t, payload, rank, world)
if rank == 0:
    per_rank = [bench.split_records(*out[r]) for r in range(world)]
    whole = bench.input_order(rank_of, per_rank)
    assert whole == [polished_of(b) for b in range(len(contigs))]
    assert sorted(int(x) for x in load) == sorted(int(cost[rank_of == r].sum()) for r in range(world))
    counts = [int(x) for x in np.bincount(rank_of, minlength=world)]
    if scenario == "few":
        assert counts.count(0) == world - n_batches        # ranks with no batch at all
    if scenario == "skew":
        assert counts[int(rank_of[3])] == 1
    print("PARTITION_GATHER_OK", counts)
dist.barrier()
dist.destroy_process_group()
'''


def _run_ranks(tmp_path, world, scenario, port0):
    script = tmp_path / "part_worker.py"
    script.write_text(_PART_WORKER)
    port = str(port0 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), str(ROOT), port, scenario],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = [p.communicate(timeout=400)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "PARTITION_GATHER_OK" in outs[0]


def test_partition_polish_order_gather_gloo_world2(tmp_path):
    """the multi-GPU flow end to end on CPU ranks: gp_partition_batches (same answer on every rank)
    -> each rank's batches in ascending order -> variable-length gather -> input order at rank 0"""
    _run_ranks(tmp_path, 2, "even", 31500)


@pytest.mark.parametrize("scenario", ["few", "drop", "skew"])
def test_partition_polish_order_gather_gloo_world8(tmp_path, scenario):
    """the same with the eight ranks of a node: fewer batches than ranks (three ranks hold nothing and
    send nothing), a rank whose records were all dropped (empty payload), one batch that outweighs the
    rest (uneven shares)"""
    _run_ranks(tmp_path, 8, scenario, {"few": 33500, "drop": 35500, "skew": 37500}[scenario])


def test_bench_dry_run_reports_every_rank_of_a_node():
    """`bench.py --gpus 8 --dry-run`: what the driver's first 8-GPU run will hold per rank, without a GPU"""
    import json
    out = subprocess.check_output([sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--dry-run"], text=True,
                                  env=dict(os.environ, GP_BENCH_NODE_CONTIGS="15000"))
    d = json.loads(out.strip().splitlines()[-1])
    assert d["n_gpus"] == 8 and len(d["ranks"]) == 8 and d["batches"] == 120000
    assert 2.9e9 < d["assembly_bp"] < 3.1e9
    assert all(r["device_bytes"] < 200 * 2**30 for r in d["ranks"])         # inside 288 GB of HBM with room to spare
    assert d["ranks"][0]["host_pinned_bytes"] > d["ranks"][1]["host_pinned_bytes"]   # rank 0 holds the gather buffers
    lo, hi = d["partition_load_min_max"]
    assert hi - lo < 0.001 * hi


_ASAN = r'''
import sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import oracle_lib
from goldpolish_amd import synth
O = oracle_lib.Oracle(sys.argv[1] + "/oracle/libgpo_asan.so")
w = synth.make_workload([3000, 1200], 25, seed=77, draft_err=0.03)
for c in range(w.n_contigs):
    outs, st, _ = O.polish_batch([w.names[c]], [w.contig(c)], [w.reads_of(c)], [int(w.kmer_threshold[c])])
    assert outs[0]
print("ASAN_OK")
'''


def test_oracle_under_address_sanitizer(tmp_path):
    """GPU sanitizers are not available on the pool; the CPU oracle at least runs clean"""
    subprocess.check_call(["make", "-C", str(ROOT / "oracle"), "libgpo_asan.so"],
                          stdout=subprocess.DEVNULL)
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    script = tmp_path / "asan.py"
    script.write_text(_ASAN)
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1")
    r = subprocess.run([sys.executable, str(script), str(ROOT)], env=env, text=True,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0 and "ASAN_OK" in r.stdout, r.stdout[-3000:]


def test_batches_are_dealt_longest_first_to_the_least_loaded_rank():
    """the partition every rank computes for itself -- deterministic, complete, and
    within the list-scheduling bound"""
    from goldpolish_amd import core
    rng = np.random.default_rng(5)
    for n_batches, n_ranks in [(0, 4), (1, 8), (7, 8), (300, 8), (300, 3), (64, 1)]:
        cost = rng.lognormal(14, 1.0, size=n_batches).astype(np.uint64)
        rank_of, load = core.partition_batches(cost, n_ranks)
        again, _ = core.partition_batches(cost, n_ranks)
        assert np.array_equal(rank_of, again)
        assert rank_of.size == n_batches and (rank_of < n_ranks).all()
        assert [int(cost[rank_of == r].sum()) for r in range(n_ranks)] == [int(x) for x in load]
        if n_batches:
            # greedy list scheduling: nobody ends later than the mean load plus one batch
            assert float(load.max()) <= float(cost.sum()) / n_ranks + float(cost.max()) * (1 - 1 / n_ranks) + 1
    # ties: lower batch index first, lower rank first
    rank_of, load = core.partition_batches([5, 5, 5, 5, 1], 2)
    assert list(rank_of) == [0, 1, 0, 1, 0] and list(load) == [11, 10]
    with pytest.raises(core.GpError):
        core.partition_batches([1, 2], 0)
