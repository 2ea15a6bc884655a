"""bench.py's configs[3] throughput leg alone (B frames per launch): python3 tests/tools/bench_tracking_batch.py [B] [steps]"""
import importlib.util, json, os, sys
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
out = bench.tracking_batch_leg(orb, ctx, B=B, steps=steps, in_flight=lanes)
out = {k: v for k, v in out.items() if not k.startswith("_")}
print(json.dumps(out))

