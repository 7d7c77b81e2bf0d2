import os, sys, time
sys.path.insert(0, '/root/repo')
from goldpolish_amd import synth
lens = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1000000,1300000,600000,2000000,200000,30000").split(",")]
w = synth.make_workload_chunked(lens, 40, seed=3, chunk=1)
from goldpolish_amd import core
p = core.default_params()
os.environ["GP_WALK_SEG"] = sys.argv[2] if len(sys.argv) > 2 else "65536"
with core.Context(0) as ctx, core.Job.from_workload(ctx, p, w) as job:
    job.polish()
    os.environ["GP_WALK_SEG_DEBUG"] = "1"; os.environ["GP_WALK_PROFILE"] = "1"
    t = time.time(); job.polish(); print("polish", time.time() - t, {n: round(job.kernel_time(n)[0], 1) for n in ("ntedit_scan", "ntedit_walk", "sealer_bfs")}, flush=True)
