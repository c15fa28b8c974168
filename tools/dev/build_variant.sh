#!/bin/bash
# Developer experiments: build libseirhip with extra -D flags into tools/dev/variants/libseirhip_<tag>.so
#   bash tools/dev/build_variant.sh <tag> [-DNAME=VALUE ...]
set -e
tag=$1; shift
root=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p "$root/tools/dev/variants"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-pass-failed "$@" \
    -o "$root/tools/dev/variants/libseirhip_${tag}.so" "$root/covid19uk_amd/csrc/seir_hip.hip"
echo "built variant $tag: $*"
