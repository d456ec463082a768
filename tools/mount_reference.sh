#!/bin/bash
# Puts a scratch copy of the reference's python package where a gpurun box can see it (the box has no /root/reference):
# _refmount/ is git-ignored (never committed, removed again after the run) but travels with the gpurun snapshot.
#   bash tools/mount_reference.sh            # in the build container
#   gpurun -- 'RECHUB_REFERENCE=$GRAFT_REPO_ROOT/_refmount python -m pytest -m gpu tests/test_integration_patch.py -v'
#   bash tools/mount_reference.sh --remove
set -euo pipefail
cd "$(dirname "$0")/.."
if [ "${1:-}" = "--remove" ]; then rm -rf _refmount; echo "removed _refmount"; exit 0; fi
rm -rf _refmount && mkdir -p _refmount
cp -r /root/reference/torch_rechub _refmount/torch_rechub
find _refmount -name __pycache__ -type d -exec rm -rf {} +
echo "mounted $(du -sh _refmount | cut -f1) at _refmount (git-ignored)"
