# Top-level build.  `make all` = everything __graft_entry__.build() needs:
#   ntsm_amd/libntsm_hip.so   HIP kernels + C ABI (include/ntsm_hip.h), gfx950
#   ntsm_amd/libntsm_synth.so synthetic workload generator (host + device fills)
#   build/ntsmCount           host CLI (C++), links libntsm_hip.so
#   build/ntsm_synth          generator CLI
#   build/ntsm_host_test      host-logic test driver (no GPU calls)
#   ntsm_amd/libntsm_eval_hip.so, build/ntsmEval   all-pairs scoring of ntsmEval (HIP library + CLI mirror)
#   oracle/                   CPU checker (+ oracle/_ref when /root/reference is present)
HIPCC    ?= /opt/rocm/bin/hipcc
CXX      ?= g++
ARCH     ?= gfx950
CXXFLAGS ?= -O3 -std=c++17 -Wall -Wextra -fPIC
HIPFLAGS ?= -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -Wall -Wextra -Wno-unused-parameter
CSRC     := ntsm_amd/csrc

HOST     := $(CSRC)/host
HOSTSRC  := $(HOST)/seq_reader.cpp $(HOST)/site_set.cpp $(HOST)/report.cpp $(HOST)/parallel_fastq.cpp \
            $(HOST)/inflate.cpp $(HOST)/inflate_spec.cpp $(HOST)/gz_stream.cpp $(HOST)/gz_parallel.cpp $(HOST)/crc32_fast.cpp $(HOST)/pack2.cpp
HOSTHDR  := $(wildcard $(HOST)/*.hpp) include/ntsm_host.h include/ntsm_hip.h

all: oracle_all build/ntsm_synth build/gather_bench ntsm_amd/libntsm_hip.so ntsm_amd/libntsm_synth.so ntsm_amd/libntsm_host.so build/ntsmCount ntsm_amd/libntsm_eval_hip.so build/ntsmEval

# host-only pieces (reader, site loader, report formatting): no HIP dependency
ntsm_amd/libntsm_host.so: $(HOSTSRC) $(HOST)/early_ingest.cpp $(HOST)/host_capi.cpp $(HOSTHDR)
	$(CXX) $(CXXFLAGS) -shared -o $@ $(HOSTSRC) $(HOST)/early_ingest.cpp $(HOST)/host_capi.cpp -lz -pthread

build/ntsmCount: $(HOSTSRC) $(HOST)/fingerprint.cpp $(HOST)/early_ingest.cpp $(HOST)/ntsm_count_main.cpp $(HOSTHDR) ntsm_amd/libntsm_hip.so
	@mkdir -p build
	$(CXX) $(CXXFLAGS) -o $@ $(HOSTSRC) $(HOST)/fingerprint.cpp $(HOST)/early_ingest.cpp $(HOST)/ntsm_count_main.cpp \
	    -Lntsm_amd -lntsm_hip -lz -pthread -Wl,-rpath,'$$ORIGIN/../ntsm_amd' -Wl,-rpath,/opt/rocm/lib

ntsm_amd/libntsm_hip.so: $(CSRC)/ntsm_hip.hip $(CSRC)/ntsm_device.h include/ntsm_hip.h
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(CSRC)/ntsm_hip.hip -ldl

# the same ABI with the tabulated k = 19 kernel compiled in (ntsm_set_kernel(ctx, 3)): a measured negative result kept
# buildable and tested (tests/test_gpu_parity.py::test_tabulated_kernel_paths loads it through NTSM_HIP_LIB), not shipped
tab: ntsm_amd/libntsm_hip_tab.so
ntsm_amd/libntsm_hip_tab.so: $(CSRC)/ntsm_hip.hip $(CSRC)/ntsm_tab_kernel.inc $(CSRC)/ntsm_device.h include/ntsm_hip.h
	$(HIPCC) $(HIPFLAGS) -DNTSM_WITH_TAB -shared -o $@ $(CSRC)/ntsm_hip.hip -ldl

# ntsmEval all-pairs scoring: own library, own CLI
ntsm_amd/libntsm_eval_hip.so: $(CSRC)/ntsm_eval.hip include/ntsm_eval_hip.h
	$(HIPCC) $(HIPFLAGS) -ffp-contract=off -shared -o $@ $(CSRC)/ntsm_eval.hip

build/ntsmEval: $(HOST)/ntsm_eval_main.cpp include/ntsm_eval_hip.h ntsm_amd/libntsm_eval_hip.so
	@mkdir -p build
	$(CXX) $(CXXFLAGS) -ffp-contract=off -o $@ $(HOST)/ntsm_eval_main.cpp -Lntsm_amd -lntsm_eval_hip \
	    -Wl,-rpath,'$$ORIGIN/../ntsm_amd' -Wl,-rpath,/opt/rocm/lib

# ablation builds of the tabulated kernel for tools/ab_libs.sh (never shipped: wrong counts by construction)
ablation: $(CSRC)/ntsm_hip.hip $(CSRC)/ntsm_tab_kernel.inc $(CSRC)/ntsm_device.h include/ntsm_hip.h
	for a in $(or $(ABL),1 2 4 5 7 8); do $(HIPCC) $(HIPFLAGS) -DNTSM_WITH_TAB -DNTSM_ABLATION -DNTSM_TAB_ABL=$$a -shared -o ntsm_amd/libntsm_hip_abl$$a.so $(CSRC)/ntsm_hip.hip -ldl & done; wait

ntsm_amd/libntsm_synth.so: $(CSRC)/synth_dev.hip $(CSRC)/synth_host.cpp $(CSRC)/synth.h include/ntsm_synth.h
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(CSRC)/synth_dev.hip $(CSRC)/synth_host.cpp -lz

oracle_all:
	$(MAKE) -C oracle all

build/ntsm_synth: tools/ntsm_synth.cpp $(CSRC)/synth_host.cpp $(CSRC)/synth.h include/ntsm_synth.h
	@mkdir -p build
	$(CXX) $(CXXFLAGS) tools/ntsm_synth.cpp $(CSRC)/synth_host.cpp -o $@ -lz

build/gather_bench: tools/gather_bench.hip
	@mkdir -p build
	$(HIPCC) -O3 --offload-arch=$(ARCH) tools/gather_bench.hip -o $@

clean:
	rm -rf build ntsm_amd/*.so
	$(MAKE) -C oracle clean
.PHONY: all oracle_all clean ablation tab
