#!/bin/bash
# A second build of the library with other flags for csrc/valley_mfma.hip (its objects otherwise: topo_descriptors_amd/build/ of the last
# full build), for A/B runs inside one GPU session (tools/ubench/vm_ab.sh; lab_libs/ is git-ignored but travels with gpurun).
# usage: tools/ubench/lab_build.sh <tag> <extra flags...>  -> /root/repo/lab_libs/libtopo_<tag>.so
set -e
tag=$1; shift
cd /root/repo/topo_descriptors_amd
mkdir -p ../lab_libs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-variable "$@" -c csrc/valley_mfma.hip -o /tmp/vm_$tag.o
objs="build/sx.o build/gauss.o $(for g in $(seq 0 15); do echo build/disc_wave_group$g.o; done) build/disc_pair.o build/disc.o build/disc_wave.o build/disc_big.o build/valley.o build/valley_fft.o build/capi.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lab_libs/libtopo_$tag.so $objs /tmp/vm_$tag.o -L/opt/rocm/lib -lrccl -lhipfft -pthread
echo built $tag
