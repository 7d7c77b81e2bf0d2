"""The per-tool programs of SURVEY 8b.2 (examples/gp_tools.cpp: goldpolish-targeted-bfs, ntedit-gr,
sealer-gr as the reference's scripts call them) driven the way scripts/goldpolish-polish-batch
drives the reference's: the filter server's FIFO protocol, one ntedit-gr per k, the mask, sealer-gr,
upper-casing -- file to file, a process per step -- must give the bytes the in-process path gives
for the same batch."""
import os
import pathlib
import subprocess
import time

import numpy as np
import pytest

from goldpolish_amd import synth

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parents[1]
BIN = ROOT / "goldpolish_amd" / "bin"


def test_tool_chain_equals_the_in_process_path(tmp_path, oracle):
    from goldpolish_amd import core
    rng = np.random.default_rng(97)
    asm, fq, paf = tmp_path / "asm.fa", tmp_path / "reads.fq", tmp_path / "asm.paf"
    rows, n_read, names = [], 0, []
    with open(asm, "wb") as fa, open(fq, "wb") as fr:
        for ci, n in enumerate([26000, 14000, 60]):       # the 60 bp contig is dropped by ntEdit
            truth = synth.random_genome(n, rng)
            draft = synth.mutate(truth, 0.01, rng)
            name = b"ctg%d" % ci
            names.append(name)
            fa.write(b">" + name + (b" some comment" if ci == 1 else b"") + b"\n" + draft.tobytes() + b"\n")
            got = 0
            while got < 30 * n:
                L = int(np.clip(rng.lognormal(np.log(5000.0), 0.5), 50, n))
                s0 = int(rng.integers(0, n - L + 1))
                frag = truth[s0:s0 + L]
                rd = synth.mutate(synth.revcomp(frag) if rng.random() < 0.5 else frag, 0.05, rng)
                rid = b"read%05d" % n_read
                n_read += 1
                q = bytes(rng.integers(33 + 5, 33 + 30, size=rd.size, dtype=np.uint8))
                fr.write(b"@" + rid + b"\n" + rd.tobytes() + b"\n+\n" + q + b"\n")
                rows.append(b"\t".join([rid, b"%d" % rd.size, b"0", b"%d" % rd.size, b"+", name, b"%d" % draft.size,
                                        b"%d" % s0, b"%d" % min(s0 + L, draft.size), b"%d" % L, b"%d" % L, b"60"]))
                got += L
    paf.write_bytes(b"\n".join(rows) + b"\n")
    ks = [32, 28, 24, 20]
    p = core.default_params()
    with core.SeqIndex(asm) as ti, core.SeqIndex(fq) as ri:
        ti.save(tmp_path / "asm.fa.index")
        ri.save(tmp_path / "reads.fq.index")
    # (both sides work from the SAVED indexes, as the reference's server does: a saved index holds the
    # average phred with six digits)
    with core.SeqIndex(asm, tmp_path / "asm.fa.index") as ti, core.SeqIndex(fq, tmp_path / "reads.fq.index") as ri:
        # ---- the in-process path: one batch holding the three contigs
        with core.Context(0) as ctx, core.Mappings(paf, ti, mx_max_mapped_per_10kbp=150.0) as m:
            with core.Job.from_files(ctx, p, ti, ri, m, [names], 40.0) as job:
                job.polish()
                job.write_fasta(tmp_path / "want.fa")
                want_filters = [job.get_filter(0, ki).copy() for ki in range(4)]

    # ---- the filter server, spoken to as scripts/goldpolish does (:313-333, goldpolish-polish-batch)
    env = dict(os.environ)
    srv = subprocess.Popen([str(BIN / "goldpolish-targeted-bfs"), str(asm), str(tmp_path / "asm.fa.index"), str(paf), str(fq),
                            str(tmp_path / "reads.fq.index"), "150", "40", "4"] + [str(k) for k in ks],
                           cwd=tmp_path, env=env, stderr=subprocess.PIPE)
    t0 = time.time()
    while not ((tmp_path / "batch_name_input").exists() and (tmp_path / "batch_target_ids_input_ready").exists()):
        assert srv.poll() is None, srv.stderr.read()
        assert time.time() - t0 < 120
        time.sleep(0.05)
    (tmp_path / "batch_name_input").open("w").write("b0\n")
    assert (tmp_path / "batch_target_ids_input_ready").open().read().strip() == "1"
    with (tmp_path / "b0-target_ids_input").open("w") as f:
        f.write("".join(n.decode() + "\n" for n in names))
    assert (tmp_path / "b0-bfs_ready").open().read().strip() == "1"
    (tmp_path / "batch_name_input").open("w").write("x\n")
    assert srv.wait(timeout=60) == 0
    assert not (tmp_path / "batch_name_input").exists() and not (tmp_path / "b0-bfs_ready").exists()
    for ki, k in enumerate(ks):
        kk, hh, data = core.filter_load(tmp_path / f"b0-k{k}.bf")
        assert (kk, hh) == (k, 4) and np.array_equal(data, want_filters[ki])

    # ---- scripts/goldpolish-ntedit:20-40: one ntedit-gr per k, each on the output of the one before
    cur = asm
    for k in ks:
        prefix = tmp_path / f"step.k{k}"
        subprocess.check_call([str(BIN / "ntedit-gr"), "-f", str(cur), "-r", str(tmp_path / f"b0-k{k}.bf"), "-d5", "-i5", "-m1",
                               "-X0.5", "-Y0.5", "-b", str(prefix), "-t4", "-a1"], cwd=tmp_path)
        cur = pathlib.Path(str(prefix) + "_edited.fa")
    # ---- goldpolish-mask -s -k32 (the oracle's restatement of the script), sealer-gr, goldpolish-to-upper
    recs = cur.read_bytes().split(b"\n")[:-1]
    with open(tmp_path / "masked.fa", "wb") as f:
        for h, s in zip(recs[0::2], recs[1::2]):
            f.write(h + b"\n" + oracle.mask(s, ks[0]) + b"\n")
    r = subprocess.run([str(BIN / "sealer-gr"), "-v", "-S", str(tmp_path / "masked.fa"), "-o", str(tmp_path / "out"), "-L32", "-j4",
                        "-P10", "-B1000", "--lower"] + [f"-k{k}" for k in ks] +
                       [f"--input-bloom={tmp_path / f'b0-k{k}.bf'}" for k in ks], cwd=tmp_path, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    assert "Gaps closed = " in r.stderr and "No start/goal kmer: " in r.stderr
    got = (tmp_path / "out_scaffold.fa").read_bytes().split(b"\n")[:-1]
    got = b"".join(h + b"\n" + s.upper() + b"\n" for h, s in zip(got[0::2], got[1::2]))
    want = (tmp_path / "want.fa").read_bytes()
    assert want.count(b">") == 2 and got == want
    # bad arguments end with a message and status 1, as the reference's tools do
    r = subprocess.run([str(BIN / "ntedit-gr"), "-f", str(asm)], stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1 and "usage" in r.stderr
