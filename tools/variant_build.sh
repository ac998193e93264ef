#!/bin/bash
# builds build/variants/lib<name>.so with extra -D flags (experiments only; the product library is liodom_amd/lib/libliodom_hip.so)
# usage: tools/variant_build.sh name "-DFOO=1 -DBAR=2"
set -e
cd $(dirname $0)/..
mkdir -p build/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value $2 -o build/variants/lib$1.so liodom_amd/csrc/liodom_hip.hip
