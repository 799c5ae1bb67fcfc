#!/bin/bash
# On the GPU box: tools/run_abl.sh <timing script> <lib suffix> ... (the libraries built by tools/build_abl.sh); '' = the default library
REPO=$(cd "$(dirname "$0")/.." && pwd)
T=$1; shift
python $REPO/tools/$T
for v in "$@"; do LSPS_HIP_LIB=$REPO/lsps_amd/liblsps_hip_$v.so python $REPO/tools/$T; done
