#!/bin/bash
# build the library in-tree and check that the package imports (from any working directory)
set -e -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
make -C "$ROOT/gnn-epc-saft_amd/csrc" -j8 "$@" 2>&1 | grep -v "^/opt/rocm/bin/hipcc\|^make: \(Entering\|Leaving\)" || true
cd "$ROOT" && python -c "import gnn_epc_saft_amd; print('library ok')"
