"""Search over the assignment of the extractors' streams to lanes (FT_LANE_MAP, ft_host.h): runs bench.py once per candidate
on the GPU box and prints frames/s.
usage: python tools/lane_search.py random N [SEED] [LANES]      N random maps (plus a few hand-made ones)
       python tools/lane_search.py climb N [SEED] [LANES] "MAP"  hill climbing from MAP: N evaluations of 1-2 entry mutations"""
import json, os, random, subprocess, sys

root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode, n = sys.argv[1], int(sys.argv[2])
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
lanes = int(sys.argv[4]) if len(sys.argv) > 4 else 8
rng = random.Random(seed)
STEPS = os.environ.get("LS_STEPS", "64")
EXTRA = os.environ.get("LS_BENCH_ARGS", "").split()  # e.g. "--workload stereo_752x480_nf1200" or "--mosaic 10"


def run(m):
    env = dict(os.environ, FT_LANE_MAP=m, GPU_MAX_HW_QUEUES=str(lanes + 2))
    try:
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-host-in", "--no-workloads", "--steps", STEPS] + EXTRA,
                             env=env, capture_output=True, text=True, timeout=120).stdout
        return json.loads(out.strip().splitlines()[-1])["value"]
    except Exception:  # noqa: BLE001
        return 0.0


if mode == "random":
    cands = ["own", "1 2 3 4 6 7 7 6 5 4 3 2 1 0 7 6", "0 1 2 3 4 5 6 7", "0 1 2 3 4 5 6 7 0 5 6 7 4 1 2 3"]
    while len(cands) < n:
        cands.append(" ".join(str(rng.randrange(lanes)) for _ in range(16)))
    res = []
    for m in cands[:n]:
        v = run(m)
        res.append((v, m))
        print(f"{v:9.0f}  {m}", flush=True)
else:
    cur = [int(x) for x in sys.argv[5].split()]
    curV = (run(" ".join(map(str, cur))) + run(" ".join(map(str, cur)))) / 2
    print(f"start {curV:9.0f}  {' '.join(map(str, cur))}", flush=True)
    res = [(curV, " ".join(map(str, cur)))]
    seen = {tuple(cur)}
    used = 2
    while used < n:
        c = list(cur)
        for _ in range(rng.choice((1, 1, 2))):
            c[rng.randrange(len(c))] = rng.randrange(lanes)
        if tuple(c) in seen:
            continue
        seen.add(tuple(c))
        m = " ".join(map(str, c))
        v = run(m)
        used += 1
        tag = ""
        if v > curV * 1.004:  # confirm before moving
            v2 = run(m)
            used += 1
            v = (v + v2) / 2
            if v > curV * 1.003:
                cur, curV, tag = c, v, "  <- accepted"
        res.append((v, m))
        print(f"{v:9.0f}  {m}{tag}", flush=True)
res.sort(reverse=True)
print("best:")
for v, m in res[:6]:
    print(f"{v:9.0f}  {m}")
