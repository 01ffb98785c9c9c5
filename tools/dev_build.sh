#!/bin/bash
# Developer build of the library with extra defines for the ARM kernel only (fast: arm_rollout.hip and capi.hip are compiled
# with the flags, the other objects come from mjmpc_amd/_build/):  tools/dev_build.sh NAME -DFLAG ...  -> tools/_build/libmjmpc_NAME.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/_build/_dev_$name
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-pass-failed -I mjmpc_amd/csrc"
/opt/rocm/bin/hipcc $F ${ARM_SCHED--mllvm -amdgpu-sched-strategy=iterative-ilp} "$@" -c mjmpc_amd/csrc/arm_rollout.hip -o tools/_build/_dev_$name/arm_rollout.o &
/opt/rocm/bin/hipcc $F "$@" -c mjmpc_amd/csrc/capi.hip -o tools/_build/_dev_$name/capi.o &
wait
objs=$(ls mjmpc_amd/_build/*.o | grep -v -e arm_rollout.o -e capi.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs tools/_build/_dev_$name/arm_rollout.o tools/_build/_dev_$name/capi.o -o tools/_build/libmjmpc_$name.so
echo tools/_build/libmjmpc_$name.so
