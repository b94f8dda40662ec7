#!/bin/bash
# usage: bash profiles/mkvariant.sh <name> "<translation units, e.g. env_cr or 'env_cr env_pg nig_api'>" "<extra hipcc flags>"
# Builds neorl-industrial-gym_amd/libnig_<name>.so = the current objects with the named translation units recompiled
# with extra flags (-D switches of an experiment or of a test-only variant), for same-box A/B runs with profiles/ab.sh
# and for tests that load a variant through NIG_LIB_PATH.  libnig.so itself is never touched.
set -e
name=$1; tus=$2; extra=$3
P=neorl-industrial-gym_amd
python -c "import importlib; importlib.import_module('neorl-industrial-gym_amd._build').build()"
objs=$(ls $P/csrc/_obj/*.o)
pids=()
for tu in $tus; do
  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -std=c++17 -w $extra -c -o /tmp/${tu}_$name.o $P/csrc/$tu.hip &
  pids+=($!)
  objs=$(echo "$objs" | grep -v "/$tu.o")
  objs="$objs"$'\n'"/tmp/${tu}_$name.o"     # (one object per line: the grep above works line by line)
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libnig_$name.so $objs -ldl
echo built $P/libnig_$name.so
