import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1] / "tests"))
import numpy as np
from goldpolish_amd import core, synth
import oracle_lib
from test_gpu_stage_a import _custom_workload
from test_gpu_stage_b import _filters_blob
O = oracle_lib.Oracle("oracle/libgpo.so")
rng = np.random.default_rng(17)
g = synth.random_genome(4000, rng).tobytes()
p = core.default_params()
hap_b = bytearray(g); hap_b[1000] = ord("A") if g[1000] != ord("A") else ord("C"); hap_b = bytes(hap_b)
reads = [g] * 8 + [hap_b] * 8
ks = list(p.k[:p.n_k])
_, bfs = O.fill_bfs(reads, ks, 5, p.cbf_bytes, p.bf_bytes, p.hash_num)
def lower(s, a, b): return s[:a] + s[a:b].lower() + s[b:]
seqs = [lower(g, 990, 1010), lower(g, 2000, 2030), g[:1500] + b"N" * 25 + g[1525:], lower(g, 10, 20), g[:700] + g[600:700] + g[600:]]
with core.Context(0) as ctx:
    for i, s in enumerate(seqs):
        w = _custom_workload([s], [[]], [5])
        job = core.Job(ctx, p, contig_buf=w.contig_buf, contig_off=w.contig_off, contig_read_begin=w.contig_read_begin,
                       kmer_threshold=w.kmer_threshold, read_buf=w.read_buf, read_off=w.read_off)
        job.set_filters(np.full(p.n_k * p.bf_bytes, 255, np.uint8)); job.ntedit_pass(0)
        job.set_filters(_filters_blob([bfs])); job.sealer()
        got = job.sequences()[0]; cnt = job.counters()
        want, st = O.sealer(s, ks, bfs, p.hash_num)
        print(i, "EQ" if got == want else "DIFF", len(s), len(got), len(want), {k: v for k, v in cnt.items() if k.startswith("c_")},
              "oracle gaps", st.gaps_found, "closed", st.gaps_closed, "uniq", st.unique_path, "multi", st.multiple_paths, "tests", st.vertex_tests)
        if got != want:
            n = min(len(got), len(want))
            d = [j for j in range(n) if got[j] != want[j]]
            print("   ndiff", len(d), "first", d[:5]); j = d[0] if d else n
            print("   in  ", s[max(0, j - 30):j + 30]); print("   got ", got[max(0, j - 30):j + 30]); print("   want", want[max(0, j - 30):j + 30])
        job.close()
