#!/bin/bash
# Builds ablation variants of the F(4x4,3x3) kernel: lsps_amd/liblsps_hip_<name>.so (select with LSPS_HIP_LIB).
set -e
cd "$(dirname "$0")/../lsps_amd/csrc"
make -j8 > /dev/null
for v in "$@"; do
  name=$(echo "$v" | tr 'A-Z' 'a-z' | sed 's/w4_abl_//; s/[^a-z0-9]/_/g')
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize $(for d in $(echo $v | tr ',' ' '); do echo -n "-D$d "; done) -c wino4.hip -o /tmp/wino4_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC igemm.o /tmp/wino4_$name.o chwn.o c8.o x3.o norm_act.o loss.o mlp_adam.o data.o -o ../liblsps_hip_$name.so
  echo built liblsps_hip_$name.so
done
