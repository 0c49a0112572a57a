#!/bin/bash
# build variants of the library with different -D flags into tools_dev/lib_<tag>.so (run in the build container)
cd /root/repo/deep3dmap_amd/csrc
i=0
for flags in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -fvisibility=hidden $flags d3m_raster.hip -o /root/repo/tools_dev/lib_v$i.so &
  i=$((i+1))
done
wait
ls -la /root/repo/tools_dev/*.so
