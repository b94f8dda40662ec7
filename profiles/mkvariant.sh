#!/bin/bash
# usage: bash profiles/mkvariant.sh <name> <translation unit, e.g. env_cr> "<extra hipcc flags>"
# Builds neorl-industrial-gym_amd/libnig_<name>.so = the current objects with ONE translation unit recompiled
# with extra flags (-D switches of an experiment), for same-box A/B runs with profiles/ab.sh.
set -e
name=$1; tu=$2; extra=$3
P=neorl-industrial-gym_amd
python -c "import importlib; importlib.import_module('neorl-industrial-gym_amd._build').build()"
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -std=c++17 -w $extra -c -o /tmp/${tu}_$name.o $P/csrc/$tu.hip
objs=$(ls $P/csrc/_obj/*.o | grep -v "/$tu.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libnig_$name.so $objs /tmp/${tu}_$name.o -ldl
echo built $P/libnig_$name.so
