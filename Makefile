# Top-level build.  `make all` = everything __graft_entry__.build() needs:
#   ntsm_amd/libntsm_hip.so   HIP kernels + C ABI (include/ntsm_hip.h), gfx950
#   ntsm_amd/libntsm_synth.so synthetic workload generator (host + device fills)
#   build/ntsmCount           host CLI (C++), links libntsm_hip.so
#   build/ntsm_synth          generator CLI
#   build/ntsm_host_test      host-logic test driver (no GPU calls)
#   oracle/                   CPU checker (+ oracle/_ref when /root/reference is present)
HIPCC    ?= /opt/rocm/bin/hipcc
CXX      ?= g++
ARCH     ?= gfx950
CXXFLAGS ?= -O3 -std=c++17 -Wall -Wextra -fPIC
HIPFLAGS ?= -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -Wall -Wextra -Wno-unused-parameter
CSRC     := ntsm_amd/csrc

all: oracle_all build/ntsm_synth

oracle_all:
	$(MAKE) -C oracle all

build/ntsm_synth: tools/ntsm_synth.cpp $(CSRC)/synth_host.cpp $(CSRC)/synth.h include/ntsm_synth.h
	@mkdir -p build
	$(CXX) $(CXXFLAGS) tools/ntsm_synth.cpp $(CSRC)/synth_host.cpp -o $@ -lz

clean:
	rm -rf build ntsm_amd/*.so
	$(MAKE) -C oracle clean
.PHONY: all oracle_all clean
