"""bench.py's configs[3] throughput leg alone (B frames per launch): python3 tests/tools/bench_tracking_batch.py [B] [steps] [lanes] [pinned 0|1] [one host thread per lane 0|1] [overlap 0|1]"""
import importlib.util, json, os, sys
# FT_BENCH_CPUS="0,1": the process is pinned to those CPUs before anything of HIP or the library is loaded (the library then picks the
# sleeping form of its host waits: option blocking_sync = 2)
if os.environ.get("FT_BENCH_CPUS"):
    os.sched_setaffinity(0, {int(c) for c in os.environ["FT_BENCH_CPUS"].split(",")})
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
from fasttrack_amd import orb
ctx = orb.Context(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 2
pinned = (int(sys.argv[4]) if len(sys.argv) > 4 else 1) != 0
one = (int(sys.argv[5]) if len(sys.argv) > 5 else 0) != 0
ovl = (int(sys.argv[6]) if len(sys.argv) > 6 else 0) != 0
out = bench.tracking_batch_leg(orb, ctx, B=B, steps=steps, in_flight=lanes, pinned=pinned, one_thread_per_lane=one, overlap=ovl)
out = {k: v for k, v in out.items() if not k.startswith("_")}
print(json.dumps(out))

