#!/bin/bash
# Builds the product library and the phase-stamp diagnostic build (objects cached per file under
# gconv-adapter_amd/build/, compiled in parallel).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
python -c "import __graft_entry__ as g; g.build()"
python -c "import __graft_entry__ as g, os; g.build_library(os.path.join(g.PKG, 'lib', 'libgconv_adapter_stamps.so'), defines=('-DGCONV_TILE_STAMPS',), tag='_stamps')"
ls -la "$ROOT/gconv-adapter_amd/lib"
