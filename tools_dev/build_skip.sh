#!/bin/bash
# build container: the library with -DD3M_DEV_SKIP (kernels named in $D3M_SKIP are not launched) as tools_dev/lib_skip.so
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -fvisibility=hidden -DD3M_DEV_SKIP \
    deep3dmap_amd/csrc/d3m_raster.hip -o tools_dev/lib_skip.so
