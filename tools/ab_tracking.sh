#!/bin/bash
# A/B of the tracking leg (bench.py workloads.tracking_512x512_nf2000) between the shipped library and fasttrack_amd/ab_prev/
# (tools/ab_build.sh prev ""): frames/s at th 7 / 15 and the track_local_map part, three alternating rounds.
for r in 1 2 3; do for v in new prev; do L=""; [ $v = prev ] && L=fasttrack_amd/ab_prev/libfasttrack_amd.so; FT_LIB=$L python bench.py --no-cpu-baseline --no-host-in --steps 4 --warmup 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['workloads']['tracking_512x512_nf2000']['by_th']; print('$v', round(t['7']['value']), round(t['15']['value']), t['7']['ms_per_frame_by_part']['track_local_map'], t['15']['ms_per_frame_by_part']['track_local_map'])"; done; done
