#!/bin/bash
# Experimental build of the library: scratch/build_variant.sh NAME [extra hipcc flags...] -> scratch/libs/NAME.so
# (selected at run time with BA_LIB=scratch/libs/NAME.so by the scratch/ab_*.py scripts)
set -e
name=$1; shift
cd "$(dirname "$0")/../3d_reconstruction_amd/csrc"
out=../../scratch/libs/$name.so
tmp=$(mktemp -d)
for f in ba_engine ba_lm ba_chol ba_init ba_subscenes ba_pairs ba_exchange ba_gauge ba_graph; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -fPIC -Wno-unused-value -I../../include "$@" -c -o $tmp/$f.o $f.hip &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out $tmp/*.o
rm -rf $tmp
echo built $out
