#!/bin/bash
# Build libmixq_hip.so for gfx950 (what __graft_entry__.build() runs): one object per source, in parallel.
set -e
cd "$(dirname "$0")/.."
make -C mixq_amd/csrc -j"$(nproc)" "$@"
echo "built mixq_amd/libmixq_hip.so"
