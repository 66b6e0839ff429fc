#!/bin/bash
# A/B builds: tools/build_variant.sh <name> "<extra hipcc flags>" <file.hip>...
# recompiles the listed csrc/*.hip files with the extra flags and links them with the other objects of build/
# (python __graft_entry__.py first) into esrganplus_amd/lib_<name>.so (git-ignored; load it with ESR_LIB_PATH).
set -e
NAME=$1; FLAGS=$2; shift 2
ROOT=$(cd $(dirname $0)/.. && pwd)
mkdir -p $ROOT/build/var_$NAME
OBJS=""
for f in $ROOT/esrganplus_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  if [[ " $* " == *" $b.hip "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -c $f -o $ROOT/build/var_$NAME/$b.o &
    OBJS="$OBJS $ROOT/build/var_$NAME/$b.o"
  else
    OBJS="$OBJS $ROOT/build/$b.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/esrganplus_amd/lib_$NAME.so $OBJS
echo built esrganplus_amd/lib_$NAME.so
