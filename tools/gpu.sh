#!/bin/bash
# Builds the library and the oracle here (the built .so files travel with the snapshot), then runs a command on the GPU box.
# usage: tools/gpu.sh <timeout-seconds> '<command>'
set -e
cd "$(dirname "$0")/.."
make -C fasttrack_amd/csrc -j8 2>&1 | grep -E "error|warning:" || true
make -C oracle 2>&1 | grep -E "error" || true
python - <<'PY'
import sys; sys.path.insert(0, '.')
from fasttrack_amd import _capi
L = _capi.lib()
missing = [s for s in _capi.declared_symbols() if not hasattr(L, s)]
assert not missing, missing
PY
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
