import os, sys, time
sys.path.insert(0, '/root/repo')
from goldpolish_amd import synth
w = synth.make_workload_chunked([1000000,1300000,600000], 40, seed=3, chunk=1)
from goldpolish_amd import core
p = core.default_params()
for env in ({"GP_WALK_SEG": "0"}, {"GP_WALK_SEG": "65536"}):
    os.environ.update(env)
    print("==", env, flush=True)
    with core.Context(0) as ctx, core.Job.from_workload(ctx, p, w) as job:
        job.polish()
        os.environ["GP_WALK_SEG_DEBUG"] = "1"
        t = time.time(); job.polish(); print("polish", time.time() - t, {n: round(job.kernel_time(n)[0], 1) for n in ("ntedit_scan", "ntedit_walk", "sealer_bfs")}, flush=True)
        os.environ.pop("GP_WALK_SEG_DEBUG")
