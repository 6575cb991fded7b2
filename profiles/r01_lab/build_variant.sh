#!/bin/bash
# usage: build_variant.sh NAME "extra flags for lws_systolic.hip"   -> lws_amd/variants/lib_NAME.so
set -e
cd /root/repo/lws_amd/csrc
mkdir -p ../variants
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-int-to-pointer-cast"
hipcc $F -fno-slp-vectorize $2 -c lws_systolic.hip -o /tmp/var_$1.o 2>/dev/null
make >/dev/null 2>&1
hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/lib_$1.so lws_capi.o lws_generic.o lws_online.o lws_stft.o lws_host.o lws_systolic_wide.o lwslib_compat.o /tmp/var_$1.o
echo built $1
