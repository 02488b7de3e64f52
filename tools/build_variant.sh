#!/bin/bash
# Alternative build of the library for in-process A/B timing (tools/ab_conv.py --libs):
#   tools/build_variant.sh NAME "-DFN2_X2_ORDER=1"   ->  flownet2-tf_amd/lib/variants/NAME.so
set -e
cd "$(dirname "$0")/../flownet2-tf_amd/csrc"
name=$1; shift
mkdir -p build/var_$name ../lib/variants
for f in $(ls *.hip | sed "s/\.hip$//"); do
  hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 -Wno-unused-function $@ -c $f.hip -o build/var_$name/$f.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/$name.so build/var_$name/*.o
echo built ../lib/variants/$name.so
