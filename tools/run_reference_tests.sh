#!/usr/bin/env bash
# Boundary (b1) check: run the REFERENCE'S OWN test-suite against this package.
#
# Build container only (/root/reference does not exist on the GPU box).  The reference's tests are copied to a scratch
# directory at run time — they are never committed or shipped — and run with
#   PYTHONPATH = whisper-finetune_amd  (the drop-in `whisper_finetune` package)
#              : oracle/stubs          (restatement of the un-vendored minLoRA dependency the tests import)
#              : oracle/stubs          (… and of `muon`, which tests/test_optimizer.py importorskips)
# Expected: 101 passed, 5 skipped (the 5 need CUDA + a downloaded whisper-tiny checkpoint).
set -euo pipefail
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
REF="${WFT_REFERENCE_ROOT:-/root/reference}"
if [ ! -d "$REF/tests" ]; then
  echo "reference tests not found under $REF (this check only runs in the build container)" >&2
  exit 2
fi
SCRATCH="$(mktemp -d /tmp/wft_reftests.XXXXXX)"
trap 'rm -rf "$SCRATCH"' EXIT
cp -r "$REF/tests" "$SCRATCH/tests"
cd "$SCRATCH"
PYTHONPATH="$REPO/whisper-finetune_amd:$REPO/oracle/stubs" PYTHONDONTWRITEBYTECODE=1 \
  python -m pytest tests -q -p no:cacheprovider -rs "$@"
