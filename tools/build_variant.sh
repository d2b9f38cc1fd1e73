#!/bin/bash
# build_variant.sh <name> <extra -D flags...>  ->  yuki_amd/libyuki_hip_<name>.so
set -e
NAME=$1; shift
cd "$(dirname "$0")/../yuki_amd/csrc"
make -s -j3 OUT=../libyuki_hip_$NAME.so BUILD=build_$NAME EXTRA="$*"
