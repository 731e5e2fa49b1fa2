#!/bin/bash
# build the WORKING TREE's library with the measurement switches (make PROBES=1) into tools/convlab/libconv_probes.so
set -e
cd "$(dirname "$0")/../.."
rm -rf /tmp/probes_build && mkdir -p /tmp/probes_build/tracking-anything-with-deva_amd
cp -r tracking-anything-with-deva_amd/csrc /tmp/probes_build/tracking-anything-with-deva_amd/csrc
cp -r include /tmp/probes_build/include
rm -f /tmp/probes_build/tracking-anything-with-deva_amd/csrc/*.o
make -s -C /tmp/probes_build/tracking-anything-with-deva_amd/csrc -j8 PROBES=1 OUT=/tmp/probes_build/lib.so >/dev/null
cp /tmp/probes_build/lib.so tools/convlab/libconv_probes.so
ls -la tools/convlab/libconv_probes.so
