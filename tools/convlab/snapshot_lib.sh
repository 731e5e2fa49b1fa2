#!/bin/bash
# build libdeva_hip.so of a commit into tools/convlab/libdeva_<tag>.so:  tools/convlab/snapshot_lib.sh <commit> <tag>
set -e
cd "$(dirname "$0")/../.."
rm -rf /tmp/snap_$2 && mkdir -p /tmp/snap_$2
git archive $1 tracking-anything-with-deva_amd/csrc include | tar -x -C /tmp/snap_$2
make -s -C /tmp/snap_$2/tracking-anything-with-deva_amd/csrc -j8 OUT=/tmp/snap_$2/lib.so >/dev/null
cp /tmp/snap_$2/lib.so tools/convlab/libdeva_$2.so
ls -la tools/convlab/libdeva_$2.so
