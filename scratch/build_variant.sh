#!/bin/bash
# usage: build_variant.sh NAME "extra flags for lws_systolic.hip"
set -e
cd /root/repo/lws_amd/csrc
mkdir -p ../variants
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function"
hipcc $F -fno-slp-vectorize $2 -c lws_systolic.hip -o /tmp/var_$1.o
[ -f lws_capi.o ] || make >/dev/null
hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/lib_$1.so lws_capi.o lws_generic.o lwslib_compat.o /tmp/var_$1.o
echo built $1
