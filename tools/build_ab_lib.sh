#!/bin/bash
# tools/build_ab_lib.sh <name> [extra hipcc flags, e.g. -DH2GCN_PLAIN_STORES] -- build a variant of libh2gcn_hip.so into
# build/ab/lib_<name>.so (what tools/ab_*.sh interleave; select one with H2GCN_HIP_LIBRARY=$PWD/build/ab/lib_<name>.so).
#   tools/build_ab_lib.sh base                      # the tree as it is
#   tools/build_ab_lib.sh plainstore -DH2GCN_PLAIN_STORES
#   git stash; tools/build_ab_lib.sh before; git stash pop      # an older state of the sources
set -eu
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/build/ab"
cd "$ROOT/h2gcn_amd/csrc"
SRCS=$(sed -n 's/^SRCS *:= *//p' Makefile)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -I"$ROOT/include" --offload-arch=gfx950 -Wno-unused-function -shared "$@" -o "$ROOT/build/ab/lib_$NAME.so" $SRCS
ls -la "$ROOT/build/ab/lib_$NAME.so"
