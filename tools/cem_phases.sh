#!/bin/bash
# Phase timing of the CEM selection launch by early exits: builds tools/_build/libcemstop{1..4}.so (update.hip compiled with
# -DKTH_STOP=n, linked with the product's other objects) -  bash tools/cem_phases.sh --build  here, then on the GPU box
# bash tools/cem_phases.sh [P]  prints the "select + moments" line of tools/cem_time.py per variant
# (1: keys loaded + shared bits, 2: + radix passes, 3: + tie cut and elite list, 4: + provisional centre, product: everything).
cd "$(dirname "$0")/.."
if [ "$1" = "--build" ]; then
  python -m mjmpc_amd.build || exit 1
  mkdir -p tools/_build
  for n in 1 2 3 4; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-pass-failed -I mjmpc_amd/csrc \
      -DKTH_STOP=$n -c mjmpc_amd/csrc/update.hip -o tools/_build/update_stop$n.o || exit 1
    objs=$(ls mjmpc_amd/_build/*.o | grep -v /update.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs tools/_build/update_stop$n.o -o tools/_build/libcemstop$n.so || exit 1
    rm tools/_build/update_stop$n.o
  done
  exit 0
fi
P=${1:-16384}
for n in 1 2 3 4; do
  echo -n "stop $n: "; MJMPC_AMD_LIB=$PWD/tools/_build/libcemstop$n.so python tools/cem_time.py $P 2>/dev/null | grep "select + moments"
done
echo -n "product: "; python tools/cem_time.py $P 2>/dev/null | grep "select + moments"
