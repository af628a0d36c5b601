#!/bin/bash
# A second build of libdfd_hip.so with one source recompiled under extra -D flags (kernel timing experiments):
#   bash scripts/build_variant.sh <name> <source.hip> -DFOO=1 ...   -> deepfakedetection_amd/_variants/libdfd_hip_<name>.so
# use it with DFD_LIB_PATH=deepfakedetection_amd/_variants/libdfd_hip_<name>.so
set -e
NAME=$1; SRC=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG=$ROOT/deepfakedetection_amd
mkdir -p $PKG/_variants
python -m deepfakedetection_amd.build > /dev/null
OBJ=$PKG/_variants/${NAME}_$(basename $SRC .hip).o
hipcc -O3 -fPIC -std=c++17 -ffp-contract=off --offload-arch=gfx950 "$@" -I$ROOT/include -c $PKG/csrc/$SRC -o $OBJ
OBJS=$(ls $PKG/csrc/build/*.o | grep -v "/$(basename $SRC .hip).o")
hipcc -shared -fPIC --offload-arch=gfx950 $OBJS $OBJ -o $PKG/_variants/libdfd_hip_$NAME.so
echo $PKG/_variants/libdfd_hip_$NAME.so
