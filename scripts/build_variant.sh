#!/bin/bash
# Build a second copy of libchore_hip.so with one source compiled differently, for same-box A/B runs through CHORE_HIP_LIB:
#   scripts/build_variant.sh <name> <source.hip> [extra hipcc flags ...]
#   e.g.  scripts/build_variant.sh stamps query_fwd.hip -DCHORE_QUERY_STAMPS=1      (-> scripts/query_stamps.py)
#         git show HEAD~1:chore_amd/csrc/train_bwd.hip > /tmp/train_bwd.hip && scripts/build_variant.sh old /tmp/train_bwd.hip
# The library lands in chore_amd/csrc/build_ab/libchore_hip_<name>.so (git-ignored; travels with gpurun).  The other objects
# are the ones of the last regular build (chore_amd/csrc/build/).
set -e
name=$1; src=$2; shift 2
cs=$(cd "$(dirname "$0")/../chore_amd/csrc" && pwd)
mkdir -p $cs/build_ab
[ -f "$src" ] || src=$cs/$src
base=$(basename $src)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -Wno-pass-failed "$@" \
    -I $cs/../../include -I $cs -c $src -o $cs/build_ab/${base}_$name.o
objs=$(ls $cs/build/*.hip.o | grep -v "/$base.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $cs/build_ab/libchore_hip_$name.so $objs $cs/build_ab/${base}_$name.o
echo $cs/build_ab/libchore_hip_$name.so
