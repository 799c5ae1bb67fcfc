#!/bin/bash
# Builds ablation variants of the C8 kernels (c8.hip): lsps_amd/liblsps_hip_<name>.so (select with LSPS_HIP_LIB).
# Usage: tools/build_abl_c8.sh C8_ABL_NOEPI C8_ABL_NODMA ...
set -e
cd "$(dirname "$0")/../lsps_amd/csrc"
make -j8 > /dev/null
for v in "$@"; do
  name=$(echo "$v" | tr 'A-Z' 'a-z' | sed 's/c8_abl_//; s/[^a-z0-9]/_/g')
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $(for d in $(echo $v | tr ',' ' '); do echo -n "-D$d "; done) -c c8.hip -o /tmp/c8_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC igemm.o wino4.o chwn.o /tmp/c8_$name.o norm_act.o loss.o mlp_adam.o data.o -o ../liblsps_hip_$name.so
  echo built liblsps_hip_$name.so
done
