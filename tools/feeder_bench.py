"""Host-side rates of the feeder: index build, index load, mapping load (dev tool)."""
import pathlib, sys, tempfile, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np
from goldpolish_amd import core, synth

rng = np.random.default_rng(1)
d = pathlib.Path(tempfile.mkdtemp(prefix="gpfeed"))
n, L = 20000, 8000
g = synth.random_genome(L * 64, rng).tobytes()
q = bytes(rng.integers(40, 70, size=L, dtype=np.uint8))
with open(d / "r.fq", "wb") as f:
    for i in range(n):
        o = (i * 7919) % (len(g) - L)
        f.write(b"@read%d\n" % i + g[o:o + L] + b"\n+\n" + q + b"\n")
with open(d / "t.fa", "wb") as f:
    for i in range(400):
        f.write(b">ctg%d\n" % i + g[:25000] + b"\n")
with open(d / "m.paf", "wb") as f:
    for i in range(n):
        f.write(b"read%d\t1\t0\t1\t+\tctg%d\t1\t0\t1\t1\t1\t60\n" % (i, i % 400))
size = (d / "r.fq").stat().st_size
t = time.perf_counter(); ix = core.SeqIndex(d / "r.fq"); dt = time.perf_counter() - t
print(f"index build: {size/1e6:.0f} MB FASTQ, {n} reads in {dt:.2f} s = {size/dt/1e6:.0f} MB/s")
ix.save(d / "r.fq.index")
t = time.perf_counter(); ix2 = core.SeqIndex(d / "r.fq", d / "r.fq.index"); dt = time.perf_counter() - t
print(f"index load: {n} records in {dt*1e3:.1f} ms")
ti = core.SeqIndex(d / "t.fa")
t = time.perf_counter(); m = core.Mappings(d / "m.paf", ti); dt = time.perf_counter() - t
print(f"PAF load: {n} rows in {dt*1e3:.1f} ms")
