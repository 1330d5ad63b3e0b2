#!/bin/bash
# development: time the direct-tile up-convolution with one phase removed at a time (results are wrong in those builds)
set -e
cd "$(dirname "$0")/../geometric_aware_dense_matching_amd/csrc"
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math"
for v in "" "-DUT_SKIP_A" "-DUT_SKIP_B" "-DUT_SKIP_C" "-DUT_SKIP_A -DUT_SKIP_B" "-DUT_SKIP_B -DUT_SKIP_C"; do
  /opt/rocm/bin/hipcc $FL $v -c gdm_upconv.hip -o gdm_upconv.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libgdm_hip.so *.o
  echo "== variant [$v]"
  (cd ../.. && python tools/bench_upconv.py 2>&1 | grep "form")
done
