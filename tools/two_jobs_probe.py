"""Does running two jobs (half the batches each) from two host threads overlap the memory-bound
filter build of one with the latency-bound ntEdit walk of the other?  Dev probe."""
import pathlib, sys, threading, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np
from goldpolish_amd import core, synth


def slice_workload(w, c0, c1):
    r0, r1 = int(w.contig_read_begin[c0]), int(w.contig_read_begin[c1])
    return dict(contig_buf=w.contig_buf[int(w.contig_off[c0]):int(w.contig_off[c1])],
                contig_off=w.contig_off[c0:c1 + 1] - w.contig_off[c0],
                contig_read_begin=w.contig_read_begin[c0:c1 + 1] - w.contig_read_begin[c0],
                kmer_threshold=w.kmer_threshold[c0:c1],
                read_buf=w.read_buf[int(w.read_off[r0]):int(w.read_off[r1])],
                read_off=w.read_off[r0:r1 + 1] - w.read_off[r0],
                name_len=[len(n) for n in w.names[c0:c1]])


n_contigs = int(sys.argv[1]) if len(sys.argv) > 1 else 400
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 2
w = synth.make_workload([25_000] * n_contigs, 30, seed=2)
p = core.default_params()
ctxs = [core.Context(0) for _ in range(parts)]
cuts = [n_contigs * i // parts for i in range(parts + 1)]
jobs = [core.Job(ctxs[i], p, **slice_workload(w, cuts[i], cuts[i + 1])) for i in range(parts)]
whole_ctx = core.Context(0)
whole = core.Job.from_workload(whole_ctx, p, w)
for rep in range(3):
    t = time.perf_counter(); whole.polish(); t_one = time.perf_counter() - t
    t = time.perf_counter()
    th = [threading.Thread(target=j.polish) for j in jobs]
    for x in th: x.start()
    for x in th: x.join()
    t_two = time.perf_counter() - t
    print(f"one job {t_one*1e3:.0f} ms = {w.total_bp/t_one/1e6:.1f} Mbp/s; {parts} jobs side by side {t_two*1e3:.0f} ms = {w.total_bp/t_two/1e6:.1f} Mbp/s", flush=True)
a = whole.sequences(); b = [s for j in jobs for s in j.sequences()]
print("same output:", a == b)
