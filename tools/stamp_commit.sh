#!/bin/bash
# Run HERE (where .git exists) right before a gpurun call that collects profiles: records HEAD (and whether the tree is dirty) in
# build/COMMIT, which travels with the snapshot and ends up in every profiles/rNN_pmc_*.json (tools/build_identity.py).
mkdir -p build
c=$(git rev-parse --short HEAD)
git diff --quiet HEAD -- deepsignal_amd/csrc include || c="$c+dirty"
echo "$c" > build/COMMIT
cat build/COMMIT
