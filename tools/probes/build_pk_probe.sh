#!/bin/bash
# builds tools/probes/libpkprobe.so (gfx950)
cd "$(dirname "$0")" && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -ffp-contract=off pk_probe.hip -o libpkprobe.so
