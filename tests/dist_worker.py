"""Worker of tests/test_shard_gloo.py: the N>1 path of bench.py (sharding, barrier, max-over-ranks time,
whole-job aggregate) with the oracle standing in for the per-rank compute, so it runs without a GPU."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from fasttrack_amd import shard, synth  # noqa: E402
from oracle import binding as ob  # noqa: E402


def main():
    rank, local_rank, world = shard.env()
    dist = shard.init(rank, world)
    seeds = shard.stream_seeds(rank, 2)
    pairs = [synth.make_stereo_pair(160, 120, s) for s in seeds]
    intr = synth.intrinsics(160, 120)
    shard.barrier(dist)
    t0 = time.perf_counter()
    kps = 0
    for L, R in pairs:
        exL, exR = ob.Extractor(300, 1.2, 4), ob.Extractor(300, 1.2, 4)
        kL, dL, _ = exL.extract(L)
        kR, dR, _ = exR.extract(R)
        ob.stereo_match(exL, exR, kL, kR, dL, dR, intr["mbf"], intr["mb"])
        kps += len(kL) + len(kR)
    time.sleep(0.05 * (rank + 1))  # uneven ranks: the job time must be the slowest rank's
    elapsed = time.perf_counter() - t0
    shard.barrier(dist)
    job_time = shard.reduce_max(dist, elapsed)
    frames, total_kps = shard.reduce_sum(dist, [len(pairs), kps])
    all_seeds = shard.gather_ints(dist, seeds, world)
    if rank == 0:
        print(json.dumps({"world": world, "job_time": job_time, "my_time": elapsed, "frames": frames, "kps": total_kps,
                          "seeds": all_seeds, "value": frames / job_time}))
    shard.finish(dist)


if __name__ == "__main__":
    main()
