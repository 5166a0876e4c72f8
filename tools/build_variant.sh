#!/bin/bash
# Builds baby-vision-curriculum_amd/libbvc_hip_<name>.so: ONE source recompiled with extra flags (or from an alternative
# file), linked with the product objects of the last build().  For same-box A/B runs: BVC_LIB_PATH=<that .so> python tools/...
# usage: tools/build_variant.sh <name> <source.hip> [extra hipcc flags...]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
P=$R/baby-vision-curriculum_amd
name=$1; src=$2; shift 2
python -c "import sys; sys.path.insert(0, '$R'); import __graft_entry__ as g; g.build()" > /dev/null
base=$(basename "$src" .hip)
mkdir -p /tmp/bvc_variant
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$P/csrc "$@" -c "$src" -o /tmp/bvc_variant/${base}_$name.o
objs=""
for o in $P/build/*.o; do
  if [ "$(basename $o .o)" = "$base" ]; then objs="$objs /tmp/bvc_variant/${base}_$name.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $P/libbvc_hip_$name.so
echo "built $P/libbvc_hip_$name.so"
