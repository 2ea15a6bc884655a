"""bench.py's configs[3] tracking leg alone (for `rocprofv3 --kernel-trace --stats -- python3 tools/profile_tracking_leg.py`)."""
import importlib.util, json, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
from fasttrack_amd import orb
ctx = orb.Context(0)
out = bench.tracking_leg(orb, ctx, frames=int(sys.argv[1]) if len(sys.argv) > 1 else 16, warmup=2, cpu=False)
print(json.dumps({th: {"fps": v["value"], "lib": v["inside_the_library"]} for th, v in out["by_th"].items()}))
