"""The oracle's side of tools/fuzz_parity.py under AddressSanitizer (CPU only; dev tool):
  make -C oracle libgpo_asan.so
  LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 \
      python tools/fuzz_oracle_asan.py <first_seed> <end_seed>"""
import sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "tools"))
import numpy as np
import fuzz_parity, oracle_lib
O = oracle_lib.Oracle(str(ROOT / "oracle" / "libgpo_asan.so"))
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    w, ks, (cbf, bf), bcb, extra = fuzz_parity.random_case(seed)
    rng = np.random.default_rng(seed + 77)
    nt = dict(max_insertions=int(rng.choice([0, 1, 3, 5])), max_deletions=int(rng.choice([0, 2, 5, 10])),
              jump=int(rng.choice([1, 2, 3, 5])), missing_ratio=extra["missing_ratio"], edit_ratio=extra["edit_ratio"],
              soft_mask=int(rng.integers(0, 2)), min_contig_len=int(rng.choice([100, 50, 500])))
    se = dict(max_paths=extra["sealer_max_paths"], max_branches=extra["sealer_max_branches"],
              max_gap_len=int(rng.choice([30, 800])), max_cost=int(rng.choice([300, 100000])))
    print("seed", seed, "lens", [int(w.contig_off[i+1]-w.contig_off[i]) for i in range(w.n_contigs)], ks, cbf, bf, nt, se, flush=True)
    contig = lambda c: bytes(w.contig_buf[int(w.contig_off[c]):int(w.contig_off[c + 1])])
    # whole-batch driver
    op = O.batch_params(k_values=tuple(ks), cbf_bytes=cbf, bf_bytes=bf, ntedit_x=extra["missing_ratio"], ntedit_y=extra["edit_ratio"],
                        sealer_P=extra["sealer_max_paths"], sealer_B=extra["sealer_max_branches"])
    for b in range(len(bcb) - 1):
        cs = list(range(bcb[b], bcb[b + 1]))
        if cs:
            O.polish_batch([w.names[c] for c in cs], [contig(c) for c in cs], [w.reads_of(c) for c in cs], [int(w.kmer_threshold[c]) for c in cs], op, want_bfs=True)
    print("  batch ok", flush=True)
    for b in range(len(bcb) - 1):
        cs = list(range(bcb[b], bcb[b + 1]))
        if not cs: continue
        cb = [np.zeros(cbf, np.uint8) for _ in ks]; bb = [np.zeros(bf, np.uint8) for _ in ks]
        for c in cs:
            if w.reads_of(c):
                O.fill_bfs(w.reads_of(c), ks, int(w.kmer_threshold[c]), cbf, bf, 4, cb, bb)
        for c in cs:
            cur = contig(c)
            for ki, k in enumerate(ks):
                if cur is None or len(cur) < nt["min_contig_len"]:
                    cur = None; break
                q = O.ntedit_params(k, 4, max_insertions=nt["max_insertions"], max_deletions=nt["max_deletions"], missing_ratio=nt["missing_ratio"],
                                    edit_ratio=nt["edit_ratio"], jump=nt["jump"], mask=nt["soft_mask"], min_contig_len=nt["min_contig_len"])
                cur, _ = O.ntedit(cur, bb[ki], k, 4, q)
            if cur is None: continue
            m = O.mask(cur, ks[0])
            O.sealer(m, ks, bb, 4, **se)
    print("  staged ok", flush=True)
