#!/bin/bash
# usage: build_variant_online.sh NAME "extra flags for lws_online.hip"
set -e
cd /root/repo/lws_amd/csrc
mkdir -p ../variants
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off"
hipcc $F $2 -c lws_online.hip -o /tmp/von_$1.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/lib_$1.so lws_capi.o lws_generic.o lws_systolic.o lws_systolic_wide.o lws_stft.o lws_host.o lwslib_compat.o /tmp/von_$1.o
echo built $1
