# Builds every native artefact in-tree:
#   pathtracer_amd/libptamd.so   HIP kernels + the C-ABI of include/pt_api.h      (hipcc, gfx950)
#   pathtracer_amd/libpthost.so  host library of include/pt_host.h                (g++)
#   pathtracer_amd/pathTracer    command line host, same usage as the reference   (g++)
#   oracle/libptoracle.so        CPU restatement of the hot path (test checker)   (gcc)
#   oracle/_ref/libptref.so      the reference's own sources + harness            (g++, only if /root/reference exists)
HIPCC   ?= /opt/rocm/bin/hipcc
CXX     ?= g++
CC      ?= gcc
ROOT    := $(dir $(abspath $(lastword $(MAKEFILE_LIST))))
INC     := -I$(ROOT)include

HOST_SRC := $(wildcard pathtracer_amd/host/*.cpp)
HOST_LIB_SRC := $(filter-out pathtracer_amd/host/main.cpp,$(HOST_SRC))
HOST_HDR := $(wildcard pathtracer_amd/host/*.hpp) $(wildcard include/*.h)

# -ffp-contract=off: host float math must round like the reference's (x86-64, no FMA contraction)
HOST_FLAGS := -std=c++17 -O2 -fPIC -pthread -ffp-contract=off -Wall -Wno-unused-function $(INC) -Ipathtracer_amd/host

HIP_SRC  := $(wildcard pathtracer_amd/csrc/*.hip)
HIP_HDR  := $(wildcard pathtracer_amd/csrc/*.h) $(wildcard pathtracer_amd/csrc/*.hpp) $(wildcard include/*.h)
HIP_FLAGS := --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math -Wall $(INC)

all: host device cli oracle ref

host: pathtracer_amd/libpthost.so
device: pathtracer_amd/libptamd.so
cli: pathtracer_amd/pathTracer
oracle: oracle/libptoracle.so

pathtracer_amd/libpthost.so: $(HOST_LIB_SRC) $(HOST_HDR)
	$(CXX) $(HOST_FLAGS) -shared -o $@ $(HOST_LIB_SRC)

pathtracer_amd/libptamd.so: $(HIP_SRC) $(HIP_HDR)
	$(HIPCC) $(HIP_FLAGS) -shared -o $@ $(HIP_SRC)

pathtracer_amd/pathTracer: pathtracer_amd/host/main.cpp pathtracer_amd/libpthost.so pathtracer_amd/libptamd.so
	$(CXX) $(HOST_FLAGS) -o $@ pathtracer_amd/host/main.cpp -Lpathtracer_amd -lpthost -lptamd -L/opt/rocm/lib -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,/opt/rocm/lib -Wl,--allow-shlib-undefined

oracle/libptoracle.so: oracle/pt_oracle.c oracle/pt_oracle.h include/pt_api.h
	$(CC) -std=c11 -O2 -fPIC -fopenmp -ffp-contract=off -fno-fast-math -Wall $(INC) -shared -o $@ oracle/pt_oracle.c -lm

ref:
	@if [ -d /root/reference/src ]; then $(MAKE) -C oracle -f Makefile.ref; else echo "no /root/reference: keeping prebuilt oracle/_ref"; fi

clean:
	rm -f pathtracer_amd/libpthost.so pathtracer_amd/libptamd.so pathtracer_amd/pathTracer oracle/libptoracle.so
	rm -rf oracle/_ref

.PHONY: all host device cli oracle ref clean
