for r in 1 2 3; do for v in base head24 head32; do
  if [ $v = base ]; then unset FT_LIB; else export FT_LIB=$PWD/fasttrack_amd/ab_$v/libfasttrack_amd.so; fi
  python3 tests/tools/bench_tracking_batch.py 128 12 3 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['by_th']['7']['value']), round(d['by_th']['15']['value']))"
done; done | sort
