#!/bin/bash
# Builds the product library (via __graft_entry__.build) and the phase-stamp diagnostic build side by side.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
( cd "$ROOT/gconv-adapter_amd" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared \
    -DGCONV_TILE_STAMPS -Wno-unused-value -Wno-pass-failed -o lib/libgconv_adapter_stamps.so csrc/*.hip ) &
( cd "$ROOT" && python -c "import __graft_entry__ as g; g.build()" )
wait
ls -la "$ROOT/gconv-adapter_amd/lib"
