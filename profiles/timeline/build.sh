#!/bin/bash
# Diagnostic (stamped) build of the library -> profiles/bin/libpicstep_timeline.so (git-ignored; travels with gpurun)
set -e
cd "$(dirname "$0")/../.."
mkdir -p profiles/bin
hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -std=c++17 -Wno-unused-result -Wno-unused-value \
  -Iinclude -Ioptimal-control-1d-electrostatic-plasma_amd/csrc -o profiles/bin/libpicstep_timeline.so profiles/timeline/picstep_timeline.hip
ls -la profiles/bin/libpicstep_timeline.so
