# Top-level build.  `make all` = everything __graft_entry__.build() needs:
#   ntsm_amd/libntsm_hip.so   HIP kernels + C ABI (include/ntsm_hip.h), gfx950: csrc/kernels_*.hip + tables / runtime / rccl_bind / capi .cpp
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

all: oracle_all build/ntsm_synth build/gather_bench build/ntsm_feed_bench build/ubench/inflate_wave ntsm_amd/libntsm_hip.so ntsm_amd/libntsm_synth.so ntsm_amd/libntsm_host.so build/ntsmCount ntsm_amd/libntsm_eval_hip.so build/ntsmEval ref_gpu_binding

# host-only pieces (reader, site loader, report formatting): no HIP dependency
ntsm_amd/libntsm_host.so: $(HOSTSRC) $(HOST)/early_ingest.cpp $(HOST)/host_capi.cpp $(HOSTHDR)
	$(CXX) $(CXXFLAGS) -shared -o $@ $(HOSTSRC) $(HOST)/early_ingest.cpp $(HOST)/host_capi.cpp -lz -pthread

build/ntsmCount: $(HOSTSRC) $(HOST)/fingerprint.cpp $(HOST)/early_ingest.cpp $(HOST)/ntsm_count_main.cpp $(HOSTHDR) ntsm_amd/libntsm_hip.so
	@mkdir -p build
	$(CXX) $(CXXFLAGS) -o $@ $(HOSTSRC) $(HOST)/fingerprint.cpp $(HOST)/early_ingest.cpp $(HOST)/ntsm_count_main.cpp \
	    -Lntsm_amd -lntsm_hip -lz -pthread -Wl,-rpath,'$$ORIGIN/../ntsm_amd' -Wl,-rpath,/opt/rocm/lib

# libntsm_hip.so = two device translation units (the kernels + their launchers) and four host-only ones
HIPLIB_DEV  := $(CSRC)/kernels_generic.hip $(CSRC)/kernels_mz.hip $(CSRC)/kernels_run.hip
HIPLIB_HOST := $(CSRC)/tables.cpp $(CSRC)/runtime.cpp $(CSRC)/rccl_bind.cpp $(CSRC)/capi.cpp
HIPLIB_HDR  := $(CSRC)/ntsm_internal.h $(CSRC)/kernels_common.h $(CSRC)/ntsm_hooks.h $(CSRC)/ntsm_device.h include/ntsm_hip.h
HIPLIB_TAB  := $(CSRC)/ntsm_tab_kernel.inc $(CSRC)/ntsm_tab_launch.inc $(CSRC)/ntsm_tab_runtime.inc $(CSRC)/ntsm_tab_tables.inc
# $(call hiplib,output,extra flags,object dir): every source to its own object (in parallel; stale objects removed first and every
# compile's exit status collected, so a failed translation unit fails the recipe instead of linking an old object), then one link
define hiplib
	@mkdir -p $(3) $(dir $(1))
	rm -f $(3)/*.o
	pids=""; for f in $(HIPLIB_DEV) $(HIPLIB_HOST); do $(HIPCC) $(HIPFLAGS) -fvisibility=hidden $(2) -c $$f -o $(3)/$$(basename $$f).o & pids="$$pids $$!"; done; \
	rc=0; for p in $$pids; do wait $$p || rc=1; done; exit $$rc
	$(HIPCC) $(HIPFLAGS) -shared -o $(1) $(foreach f,$(HIPLIB_DEV) $(HIPLIB_HOST),$(3)/$(notdir $(f)).o) -ldl
endef

ntsm_amd/libntsm_hip.so: $(HIPLIB_DEV) $(HIPLIB_HOST) $(HIPLIB_HDR)
	$(call hiplib,$@,,build/obj)

# Experiment and negative-result builds of the same ABI live under build/lib/, NOT in the package directory: the package ships
# libntsm_hip.so, libntsm_host.so, libntsm_synth.so and libntsm_eval_hip.so only.  NTSM_HIP_LIB=<bare name> finds them there
# (ntsm_amd/capi.py).
# the same ABI with the tabulated k = 19 kernel compiled in (ntsm_set_kernel(ctx, 3)): a measured negative result kept
# buildable and tested (tests/test_gpu_parity.py::test_tabulated_kernel_paths loads it through NTSM_HIP_LIB), not shipped
tab: build/lib/libntsm_hip_tab.so
build/lib/libntsm_hip_tab.so: $(HIPLIB_DEV) $(HIPLIB_HOST) $(HIPLIB_HDR) $(HIPLIB_TAB)
	$(call hiplib,$@,-DNTSM_WITH_TAB,build/obj_tab)

# design-study build: the two-level form with 12-mer minimizers instead of 14-mers, i.e. a
# minimizer Bloom in front of the one-level kernel's own blocks -- tools/mid_study.sh measures it on the 2.5 M-key set
m12: build/lib/libntsm_hip_m12.so
build/lib/libntsm_hip_m12.so: $(HIPLIB_DEV) $(HIPLIB_HOST) $(HIPLIB_HDR)
	$(call hiplib,$@,-DNTSM_TWO_M=12,build/obj_m12)

# experiment builds of the same ABI: make xlib XNAME=run96 XFLAGS="-DNTSM_RUN_C=96 ..." -> build/lib/libntsm_hip_run96.so (NTSM_HIP_LIB selects it)
xlib:
	$(call hiplib,build/lib/libntsm_hip_$(XNAME).so,$(XFLAGS),build/obj_$(XNAME))

# ntsmEval all-pairs scoring: own library, own CLI
ntsm_amd/libntsm_eval_hip.so: $(CSRC)/ntsm_eval.hip include/ntsm_eval_hip.h
	$(HIPCC) $(HIPFLAGS) -ffp-contract=off -shared -o $@ $(CSRC)/ntsm_eval.hip

build/ntsmEval: $(HOST)/ntsm_eval_main.cpp include/ntsm_eval_hip.h ntsm_amd/libntsm_eval_hip.so
	@mkdir -p build
	$(CXX) $(CXXFLAGS) -ffp-contract=off -o $@ $(HOST)/ntsm_eval_main.cpp -Lntsm_amd -lntsm_eval_hip \
	    -Wl,-rpath,'$$ORIGIN/../ntsm_amd' -Wl,-rpath,/opt/rocm/lib

# ablation builds (never shipped: wrong counts by construction).  `make ablation`: the default kernels with the switches of
# ntsm_ablation.inc (NTSM_DEBUG_KERNEL ...) -> build/lib/libntsm_hip_abl.so; `make ablation ABL="1 2 4"`: the tabulated
# kernel's compile-time ablations as well -> build/lib/libntsm_hip_abl<N>.so (tools/ablate.sh drives both)
ablation: $(HIPLIB_DEV) $(HIPLIB_HOST) $(HIPLIB_HDR) $(HIPLIB_TAB) $(CSRC)/ntsm_ablation.inc
	$(call hiplib,build/lib/libntsm_hip_abl.so,-DNTSM_ABLATION $(ABLFLAGS),build/obj_abl)
	for a in $(ABL); do $(MAKE) --no-print-directory abl_tab A=$$a; done
abl_tab:
	$(call hiplib,build/lib/libntsm_hip_abl$(A).so,-DNTSM_WITH_TAB -DNTSM_ABLATION -DNTSM_TAB_ABL=$(A),build/obj_abl$(A))

ntsm_amd/libntsm_synth.so: $(CSRC)/synth_dev.hip $(CSRC)/synth_host.cpp $(CSRC)/synth.h include/ntsm_synth.h
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(CSRC)/synth_dev.hip $(CSRC)/synth_host.cpp -lz

oracle_all:
	$(MAKE) -C oracle all
# INTEGRATION.md section 2 compiled around the unmodified reference class and linked against the product library (test
# infrastructure: oracle/ref_gpu_binding.cpp -> oracle/_ref/ref_gpu_ntsmCount; only where /root/reference exists)
ref_gpu_binding: ntsm_amd/libntsm_hip.so
	$(MAKE) -C oracle refgpu

build/ntsm_synth: tools/ntsm_synth.cpp $(CSRC)/synth_host.cpp $(CSRC)/synth.h include/ntsm_synth.h
	@mkdir -p build
	$(CXX) $(CXXFLAGS) tools/ntsm_synth.cpp $(CSRC)/synth_host.cpp -o $@ -lz

# the host-fed path on a PCIe roofline (bench.py other_configs.feed)
build/ntsm_feed_bench: tools/feed_bench.cpp $(HOST)/pack2.cpp $(HOSTHDR) include/ntsm_synth.h ntsm_amd/libntsm_hip.so ntsm_amd/libntsm_host.so ntsm_amd/libntsm_synth.so
	@mkdir -p build
	$(HIPCC) -O3 -std=c++17 -Wall -Wextra -o $@ tools/feed_bench.cpp $(HOST)/pack2.cpp -Lntsm_amd -lntsm_hip -lntsm_host -lntsm_synth -pthread \
	    -Wl,-rpath,'$$ORIGIN/../ntsm_amd' -Wl,-rpath,/opt/rocm/lib

# gate for a device-side inflate of FASTQ .gz: one wave per chunk of a deflate stream
build/ubench/inflate_wave: tools/ubench/inflate_wave.hip $(CSRC)/synth_host.cpp $(CSRC)/synth.h include/ntsm_synth.h
	@mkdir -p build/ubench
	$(HIPCC) -O3 -std=c++17 --offload-arch=$(ARCH) -Wall -Wextra -o $@ tools/ubench/inflate_wave.hip $(CSRC)/synth_host.cpp -lz -pthread

build/gather_bench: tools/gather_bench.hip
	@mkdir -p build
	$(HIPCC) -O3 --offload-arch=$(ARCH) tools/gather_bench.hip -o $@

clean:
	rm -rf build ntsm_amd/*.so
	$(MAKE) -C oracle clean
.PHONY: all oracle_all ref_gpu_binding clean ablation abl_tab tab m12 xlib
