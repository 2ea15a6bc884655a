#!/bin/bash
# Dumps the gfx950 ISA of one kernel of a .hip file (comment lines removed) and prints its instruction-class counts.
# usage: tools/isa_dump.sh <file.hip> <mangled-name-prefix> <out.s>
REPO=$(cd "$(dirname "$0")/.." && pwd)
SRC=$REPO/fasttrack_amd/csrc/$1; PFX=$2; OUT=${3:-/tmp/isa/kernel.s}
mkdir -p "$(dirname "$OUT")" /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -fno-fast-math -I$REPO/include -S --cuda-device-only -o /tmp/isa/_full.s "$SRC" 2>&1 | grep -E "error"
python3 - "$PFX" "$OUT" <<'PY'
import re, collections, sys
pfx, out = sys.argv[1], sys.argv[2]
lines = open('/tmp/isa/_full.s').read().split('\n')
start = [i for i, l in enumerate(lines) if l.startswith(pfx)][0]
end = [i for i, l in enumerate(lines) if l.startswith('.Lfunc_end') and i > start][0]
body = [l for l in lines[start:end] if not l.strip().startswith(';')]
open(out, 'w').write('\n'.join(body))
c = collections.Counter(re.match(r'\s+(\w+)', l).group(1).split('_')[0] for l in body if re.match(r'\s+\w+', l))
slow = collections.Counter(re.match(r'\s+(\w+)', l).group(1) for l in body if re.match(r'\s+(v_mul_lo_u32|v_mul_hi_u32|v_mad_u64_u32|v_mad_i64_i32|flat_\w+)', l))
print(len(body), dict(c), dict(slow))
PY
