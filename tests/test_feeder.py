"""SeqIndex / AllMappings / serve_batch's read selection in libgpcore (host side)
against the Python restatement in tests/feeder_restatement.py, on files written here."""
import numpy as np
import pytest

import feeder_restatement as R
from goldpolish_amd import synth


def _core():
    from goldpolish_amd import core
    return core


def write_reads_fastq(path, rng, n=60, dup=True):
    recs = []
    for i in range(n):
        L = int(rng.integers(30, 400))
        seq = synth.random_genome(L, rng).tobytes()
        qual = bytes(rng.integers(33 + 2, 33 + 41, size=L, dtype=np.uint8))
        header = b"@read%d" % i
        if i % 3 == 0:
            header += b" comment with spaces"
        if i % 5 == 0:
            header += b"\ttabbed"
        if i % 7 == 0:
            header = b"@read%d\ttab first then space" % i
        recs.append((header, seq, qual))
    if dup:
        recs.append((b"@read3 again", b"ACGT" * 5, b"I" * 20))       # duplicate id: first record wins
    recs.append((b"@single", b"A", b"#"))                            # one quality character
    with open(path, "wb") as f:
        for h, s, q in recs:
            f.write(h + b"\n" + s + b"\n+\n" + q + b"\n")
    return recs


def write_targets_fasta(path, rng, lens):
    recs = []
    with open(path, "wb") as f:
        for i, L in enumerate(lens):
            seq = synth.random_genome(L, rng).tobytes()
            h = b">ctg%d" % i + (b" len=%d extra words" % L if i % 2 else b"")
            f.write(h + b"\n" + seq + b"\n")
            recs.append((h, seq))
    return recs


def test_index_build_save_load(tmp_path):
    core = _core()
    rng = np.random.default_rng(1)
    fq = tmp_path / "reads.fq"
    recs = write_reads_fastq(fq, rng)
    want, order = R.build_index(fq)
    with core.SeqIndex(fq) as ix:
        got = ix.entries()
        assert [e[0] for e in got] == order
        for sid, start, ln, ph in got:
            assert (start, ln) == want[sid][:2]
            assert ph == want[sid][2], sid                       # same double, not just close
        assert ix.find(b"read3")[1] == len(recs[3][1]) and ix.find(b"nope") is None
        assert ix.read(b"read10") == recs[10][1] and ix.read(b"single") == b"A"
        ix.save(tmp_path / "reads.fq.index")
    lines = (tmp_path / "reads.fq.index").read_bytes().split(b"\n")[:-1]
    assert len(lines) == len(order)
    with core.SeqIndex(fq, tmp_path / "reads.fq.index") as ix2:
        for sid, start, ln, ph in ix2.entries():
            assert (start, ln) == want[sid][:2] and ph == R.saved_phred(want[sid][2])
        assert ix2.read(b"read10") == recs[10][1]
    fa = tmp_path / "targets.fa"
    trecs = write_targets_fasta(fa, rng, [500, 1200, 90])
    want, order = R.build_index(fa)
    with core.SeqIndex(fa) as ix:
        assert [(e[0], e[1], e[2], e[3]) for e in ix.entries()] == [(s,) + want[s] for s in order]
        assert ix.read(b"ctg1") == trecs[1][1]
    with pytest.raises(core.GpError):
        core.SeqIndex(tmp_path / "missing.fa")


def _targets(tmp_path, rng, lens):
    fa = tmp_path / "targets.fa"
    write_targets_fasta(fa, rng, lens)
    return fa, R.build_index(fa)[0]


def test_mappings_paf_and_sam(tmp_path):
    core = _core()
    rng = np.random.default_rng(2)
    fa, tix = _targets(tmp_path, rng, [500, 1200, 90])
    paf = tmp_path / "m.paf"
    rows = []
    for i in range(80):
        t = b"ctg%d" % int(rng.integers(0, 5))                    # ctg3, ctg4 are not indexed
        r = b"read%d" % int(rng.integers(0, 30))
        rows.append(b"\t".join([r, b"100", b"0", b"100", b"+", t, b"500", b"0", b"100", b"90", b"100", b"60"]))
    rows.insert(5, b"@a header-like line")
    rows.insert(9, b"lonely_read\t100")                           # short line: inherits the target
    rows.insert(20, b"")                                           # empty line
    paf.write_bytes(b"\n".join(rows) + b"\n")
    sam = tmp_path / "m.sam"
    srows = [b"@HD\tVN:1.6", b"@SQ\tSN:ctg0\tLN:500"]
    for i in range(60):
        t = b"ctg%d" % int(rng.integers(0, 4))
        r = b"read%d" % int(rng.integers(0, 25))
        srows.append(b"\t".join([r, b"0", t, b"1", b"60", b"100M", b"*", b"0", b"0", b"ACGT", b"IIII"]))
    sam.write_bytes(b"\n".join(srows) + b"\n")
    with core.SeqIndex(fa) as ix:
        for path in (paf, sam):
            want = R.load_mappings(path, tix)
            with core.Mappings(path, ix) as m:
                for t in (b"ctg0", b"ctg1", b"ctg2", b"ctg3", b"nope"):
                    assert m.get(t) == want.get(t, []), (path.name, t)
        with pytest.raises(core.GpError):
            core.Mappings(tmp_path / "x.bam", ix)


def test_mappings_ntlink_threshold_search(tmp_path):
    core = _core()
    rng = np.random.default_rng(3)
    fa, tix = _targets(tmp_path, rng, [300, 2500, 10000, 120])
    tsv = tmp_path / "reads.fq.k32.w100.z1000.verbose_mapping.tsv"
    rows = []
    for i in range(900):
        t = int(rng.choice([0, 1, 2, 3, 7], p=[0.1, 0.3, 0.4, 0.1, 0.1]))
        rows.append(b"read%d ctg%d %d" % (int(rng.integers(0, 400)), t, int(rng.integers(0, 45))))
    tsv.write_bytes(b"\n".join(rows) + b"\n")
    with core.SeqIndex(fa) as ix:
        for per10k in (100.0, 40.0, 7.5, 1.0):
            for mx_min, mx_max in ((1, 30), (3, 12)):
                want = R.load_mappings(tsv, tix, mx_min, mx_max, per10k)
                with core.Mappings(tsv, ix, mx_min, mx_max, per10k) as m:
                    for t in (b"ctg0", b"ctg1", b"ctg2", b"ctg3", b"ctg7"):
                        assert m.get(t) == want.get(t, []), (per10k, mx_min, t)
        with pytest.raises(core.GpError):
            core.Mappings(tsv, ix, 5, 5, 10.0)


@pytest.mark.gpu
def test_job_from_files_equals_job_from_arrays(tmp_path):
    """targets + reads + PAF on disk -> the same filters and polished contigs as handing the
    selected reads over in arrays (selection done by the Python restatement)."""
    core = _core()
    w = synth.make_workload([6000, 9000, 250, 4000], 25, seed=77)
    fa, fq, paf = tmp_path / "t.fa", tmp_path / "r.fq", tmp_path / "m.paf"
    rng = np.random.default_rng(5)
    with open(fa, "wb") as f:
        for c in range(w.n_contigs):
            f.write(b">" + w.names[c].encode() + (b" some comment" if c % 2 else b"") + b"\n" + w.contig(c) + b"\n")
    rows, names = [], []
    with open(fq, "wb") as f:
        for c in range(w.n_contigs):
            for j, r in enumerate(w.reads_of(c)):
                rid = b"r%d_%d" % (c, j)
                q = bytes(rng.integers(33 + 5, 33 + 30, size=len(r), dtype=np.uint8))
                f.write(b"@" + rid + b" x\n" + r + b"\n+\n" + q + b"\n")
                rows.append(b"\t".join([rid, b"1", b"0", b"1", b"+", w.names[c].encode(), b"1", b"0", b"1", b"1", b"1", b"60"]))
    rng.shuffle(rows)
    paf.write_bytes(b"\n".join(rows) + b"\n")
    tix, _ = R.build_index(fa)
    rix, _ = R.build_index(fq)
    maps = R.load_mappings(paf, tix)
    s = 40.0
    batches = [[w.names[0], w.names[1]], [w.names[2]], [w.names[3]]]
    # arrays built from the restatement's selection
    reads_bytes = open(fq, "rb").read()
    sel, thr = [], []
    for c in range(w.n_contigs):
        chosen, t = R.select_reads(len(w.contig(c)), maps.get(w.names[c].encode(), []), rix, s)
        sel.append([reads_bytes[rix[r][0]:rix[r][0] + rix[r][1]] for r in chosen])
        thr.append(t)
    flat = [r for rs in sel for r in rs]
    roff = np.concatenate([[0], np.cumsum([len(r) for r in flat])]).astype(np.uint64)
    crb = np.concatenate([[0], np.cumsum([len(rs) for rs in sel])]).astype(np.uint32)
    p = core.default_params(k_values=[32, 24])
    with core.Context(0) as ctx:
        with core.Job(ctx, p, contig_buf=w.contig_buf, contig_off=w.contig_off, contig_read_begin=crb,
                      kmer_threshold=np.array(thr, dtype=np.uint32),
                      read_buf=np.frombuffer(b"".join(flat), dtype=np.uint8), read_off=roff,
                      batch_contig_begin=[0, 2, 3, 4],
                      name_len=[len(w.names[c]) + (13 if c % 2 else 0) for c in range(w.n_contigs)]) as a:
            a.polish()
            want = a.sequences()
            want_f = [a.get_filter(b, 0).tobytes() for b in range(3)]
        with core.SeqIndex(fa) as ti, core.SeqIndex(fq) as ri, core.Mappings(paf, ti) as m:
            with core.Job.from_files(ctx, p, ti, ri, m, batches, s) as b:
                b.polish()
                assert b.sequences() == want
                assert [b.get_filter(i, 0).tobytes() for i in range(3)] == want_f
                # FASTA out: the target file's header lines, input order, dropped records left out
                b.write_fasta(tmp_path / "out.fa")
                exp = b"".join(b">" + w.names[c].encode() + (b" some comment" if c % 2 else b"") + b"\n" + want[c] + b"\n"
                               for c in range(w.n_contigs) if want[c] is not None)
                assert (tmp_path / "out.fa").read_bytes() == exp
    assert any(x is not None and x != w.contig(i) for i, x in enumerate(want))   # something was polished


def test_filter_file_round_trip(tmp_path):
    """the build's own filter container (GPBF1): what save()/load of a .bf does for the reference"""
    core = _core()
    rng = np.random.default_rng(9)
    data = rng.integers(0, 256, size=4099, dtype=np.uint8)
    core.filter_save(tmp_path / "b-k24.bf", 24, 4, data)
    k, h, got = core.filter_load(tmp_path / "b-k24.bf")
    assert (k, h) == (24, 4) and np.array_equal(got, data)
    assert (tmp_path / "b-k24.bf").read_bytes().startswith(b"GPBF1 24 4 4099\n")
    (tmp_path / "other.bf").write_bytes(b"[BTLKmerBloomFilter_v2]\nbytes = 4\n")
    with pytest.raises(core.GpError):
        core.filter_load(tmp_path / "other.bf")


def test_filter_file_btl_style_header(tmp_path):
    """a file laid out the way SURVEY 8a-A7 describes btllib's (hand-written header: that layout is
    UNPINNED here): keys in any order, unknown keys skipped, payload length checked"""
    core = _core()
    rng = np.random.default_rng(10)
    data = rng.integers(0, 256, size=1024, dtype=np.uint8)
    hdr = b'[BTLKmerBloomFilter_v6]\nhash_fn = "ntHash_v1"\nk = 28\nbytes = 1024\nhash_num = 3\n[HeaderEnd]\n'
    (tmp_path / "x.bf").write_bytes(hdr + data.tobytes())
    k, h, got = core.filter_load(tmp_path / "x.bf")
    assert (k, h) == (28, 3) and np.array_equal(got, data)
    for bad in (hdr + data.tobytes()[:-1],                                   # short payload
                hdr.replace(b"k = 28\n", b"") + data.tobytes(),              # a plain BloomFilter: no k
                hdr.replace(b"[HeaderEnd]\n", b"") + data.tobytes(),
                b"[BTLCountingBloomFilter_v6]\n" + hdr.split(b"\n", 1)[1] + data.tobytes()):
        (tmp_path / "bad.bf").write_bytes(bad)
        with pytest.raises(core.GpError):
            core.filter_load(tmp_path / "bad.bf")


def test_filter_file_written_under_the_btllib_header(tmp_path):
    """gp_filter_save_btllib: the published layout (UNPINNED), read back by gp_filter_load; and a
    filter file that cannot be written whole is an error that leaves no file behind"""
    import resource
    core = _core()
    rng = np.random.default_rng(11)
    data = rng.integers(0, 256, size=70000, dtype=np.uint8)
    core.filter_save(tmp_path / "7-k32.bf", 32, 4, data, btllib_header=True)
    raw = (tmp_path / "7-k32.bf").read_bytes()
    hdr = b'[BTLKmerBloomFilter_v2]\nbytes = 70000\nhash_fn = "ntHash_v2"\nhash_num = 4\nk = 32\n[HeaderEnd]\n'
    assert raw == hdr + data.tobytes()
    k, h, got = core.filter_load(tmp_path / "7-k32.bf")
    assert (k, h) == (32, 4) and np.array_equal(got, data)
    soft, hard = resource.getrlimit(resource.RLIMIT_FSIZE)
    try:
        resource.setrlimit(resource.RLIMIT_FSIZE, (10000, hard))   # write() fails with EFBIG
        for btl in (False, True):
            with pytest.raises(core.GpError) as e:
                core.filter_save(tmp_path / "full.bf", 32, 4, data, btllib_header=btl)
            assert e.value.code == -10 and not (tmp_path / "full.bf").exists()
    finally:
        resource.setrlimit(resource.RLIMIT_FSIZE, (soft, hard))


@pytest.mark.gpu
def test_cpp_host_program_over_the_c_abi(tmp_path):
    """examples/gp_polish_files.cpp (C++ above include/gpcore.h: index, mappings, batches of -b
    contigs, polish, FASTA out) writes what the same calls made through the ctypes binding write;
    jobs cut by -j are appended in input order"""
    import pathlib
    import subprocess
    core = _core()
    exe = pathlib.Path(core.LIB_PATH).parent / "gp_polish_files"
    if not exe.exists():
        subprocess.check_call(["make", "-C", str(pathlib.Path(core.LIB_PATH).parent / "csrc")])
    w = synth.make_workload([5000, 7000, 250, 3000, 4500], 20, seed=78)
    fa, fq, paf = tmp_path / "t.fa", tmp_path / "r.fq", tmp_path / "m.paf"
    rng = np.random.default_rng(6)
    with open(fa, "wb") as f:
        for c in range(w.n_contigs):
            f.write(b">" + w.names[c].encode() + (b" a comment" if c % 2 else b"") + b"\n" + w.contig(c) + b"\n")
    rows = []
    with open(fq, "wb") as f:
        for c in range(w.n_contigs):
            for j, r in enumerate(w.reads_of(c)):
                rid = b"r%d_%d" % (c, j)
                q = bytes(rng.integers(33 + 5, 33 + 30, size=len(r), dtype=np.uint8))
                f.write(b"@" + rid + b"\n" + r + b"\n+\n" + q + b"\n")
                rows.append(b"\t".join([rid, b"1", b"0", b"1", b"+", w.names[c].encode(), b"1", b"0", b"1", b"1", b"1", b"60"]))
    paf.write_bytes(b"\n".join(rows) + b"\n")
    p = core.default_params(k_values=[32, 24])
    batches = [[w.names[0], w.names[1]], [w.names[2], w.names[3]], [w.names[4]]]   # -b 2
    with core.Context(0) as ctx, core.SeqIndex(fa) as ti, core.SeqIndex(fq) as ri, core.Mappings(paf, ti) as m:
        with core.Job.from_files(ctx, p, ti, ri, m, batches, 40.0) as job:
            job.polish()
            job.write_fasta(tmp_path / "py.fa")
    # -j 0: every batch its own job, appended; -d 0,0 / 0,0,0: the several-devices path (a host thread
    # and a context each, batches dealt by gp_partition_batches, part files merged back into input
    # order) with every "device" being GPU 0
    for extra in ([], ["-j", "0"], ["-d", "0,0"], ["-d", "0,0,0", "-j", "0"]):
        out = tmp_path / "cpp.fa"
        r = subprocess.run([str(exe), "-k", "32", "-k", "24", "-b", "2", *extra, str(fa), str(fq), str(paf), str(out)],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        assert out.read_bytes() == (tmp_path / "py.fa").read_bytes(), extra
        assert "Gaps closed = " in r.stderr and "No start/goal kmer: " in r.stderr
        assert not list(tmp_path.glob("cpp.fa.part*"))
    r = subprocess.run([str(exe), str(fa), str(fq), str(tmp_path / "missing.paf"), str(tmp_path / "x.fa")],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "loading the mappings failed" in r.stderr
