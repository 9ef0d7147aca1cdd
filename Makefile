# Top-level build.  Everything is built in-tree (build/ and *.so are git-ignored but travel with gpurun).
#
#   make            product: libpagraph_hip.so (HIP kernels + C ABI, gfx950) and the `pagraph` executable
#   make harness    test-only programs under tests/harness/ (these link the ORACLE; never shipped)
#   make oracle     oracle/libpag_oracle.so (+ oracle/_ref/* when /root/reference is present)
HIPCC    ?= /opt/rocm/bin/hipcc
CXX      ?= g++
CC       ?= gcc
ARCH     ?= gfx950
CXXFLAGS := -O2 -std=c++17 -Wall -Wextra -fPIC -Iinclude -Ialigngraph2_amd/csrc/host
ifdef WALK_PROF
PROF_FLAGS := -DPAG_WALK_PROF
endif
WALK_WINDOW ?= tiny
ifeq ($(WALK_WINDOW),small)
PROF_FLAGS += -DPAG_WALK_SMALL_WINDOW
endif
ifeq ($(WALK_WINDOW),tiny)
PROF_FLAGS += -DPAG_WALK_TINY_WINDOW
endif
ifdef WALK_EU
PROF_FLAGS += -DPAG_WALK_WAVES_PER_EU=$(WALK_EU)
endif
HIPFLAGS := $(PROF_FLAGS) -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -Iinclude -Ialigngraph2_amd/csrc/hip -ffp-contract=off -Wall -Wno-unused-value

HOST_DIR := aligngraph2_amd/csrc/host
HIP_DIR  := aligngraph2_amd/csrc/hip
B        := build

HOST_SRCS := $(wildcard $(HOST_DIR)/*.cpp)
HOST_LIB_SRCS := $(filter-out %_main.cpp,$(HOST_SRCS))
HOST_OBJS := $(patsubst $(HOST_DIR)/%.cpp,$(B)/host/%.o,$(HOST_LIB_SRCS))
HIP_SRCS  := $(wildcard $(HIP_DIR)/*.hip)
HIP_HDRS  := $(wildcard $(HIP_DIR)/*.h) $(wildcard $(HIP_DIR)/*.hpp) include/pagraph_hip.h

.PHONY: all product harness oracle clean sort_variants
all: product harness oracle

product: aligngraph2_amd/libpagraph_hip.so aligngraph2_amd/bin/pagraph aligngraph2_amd/bin/kmer_counter aligngraph2_amd/bin/pre_process aligngraph2_amd/bin/pa_cns aligngraph2_amd/bin/paf2aln aligngraph2_amd/libpagraph_host.so

$(B)/host/%.o: $(HOST_DIR)/%.cpp $(wildcard $(HOST_DIR)/*.hpp) include/pagraph_hip.h
	@mkdir -p $(B)/host
	$(CXX) $(CXXFLAGS) -c $< -o $@

$(B)/libpagh_host.a: $(HOST_OBJS)
	@rm -f $@
	ar rcs $@ $^

aligngraph2_amd/libpagraph_hip.so: $(HIP_SRCS) $(HIP_HDRS)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(HIP_SRCS)

aligngraph2_amd/bin/pagraph: $(HOST_DIR)/pagraph_main.cpp $(B)/libpagh_host.a aligngraph2_amd/libpagraph_hip.so
	@mkdir -p aligngraph2_amd/bin
	$(CXX) $(CXXFLAGS) -o $@ $< $(B)/libpagh_host.a -Laligngraph2_amd -lpagraph_hip -Wl,-rpath,'$$ORIGIN/..' -pthread

aligngraph2_amd/bin/kmer_counter: $(HOST_DIR)/kmer_counter_main.cpp $(B)/libpagh_host.a aligngraph2_amd/libpagraph_hip.so
	@mkdir -p aligngraph2_amd/bin
	$(CXX) $(CXXFLAGS) -o $@ $< $(B)/libpagh_host.a -Laligngraph2_amd -lpagraph_hip -Wl,-rpath,'$$ORIGIN/..' -pthread

# host-only tool (no device work in the reference either)
aligngraph2_amd/bin/pre_process: $(HOST_DIR)/pre_process_main.cpp $(HOST_DIR)/line_index.hpp
	@mkdir -p aligngraph2_amd/bin
	$(CXX) $(CXXFLAGS) -o $@ $< -pthread

# the consensus step after pagraph: parsing / slicing on the host, the per-part graphs on host threads or on the device (csrc/hip/k_cns.hip;
# libpagraph_hip.so is loaded at run time, by the device backend only)
aligngraph2_amd/bin/pa_cns: $(HOST_DIR)/pa_cns_main.cpp $(B)/host/seq_db.o $(wildcard $(HOST_DIR)/*.hpp) $(HIP_DIR)/cns_graph.hpp
	@mkdir -p aligngraph2_amd/bin
	$(CXX) $(CXXFLAGS) -o $@ $< $(B)/host/seq_db.o -ldl -pthread

aligngraph2_amd/bin/paf2aln: $(HOST_DIR)/paf2aln_main.cpp
	@mkdir -p aligngraph2_amd/bin
	$(CXX) $(CXXFLAGS) -o $@ $< -pthread

aligngraph2_amd/libpagraph_host.so: $(B)/libpagh_host.a aligngraph2_amd/libpagraph_hip.so
	$(CXX) -shared -o $@ -Wl,--whole-archive $(B)/libpagh_host.a -Wl,--no-whole-archive -Laligngraph2_amd -lpagraph_hip -Wl,-rpath,'$$ORIGIN' -pthread

# ---- test-only -------------------------------------------------------------------------------
oracle:
	$(MAKE) -C oracle all

$(B)/pag_oracle.o: oracle/pag_oracle.c oracle/pag_oracle.h include/pagraph_hip.h
	@mkdir -p $(B)
	$(CC) -O2 -std=c99 -Wall -Wextra -fPIC -Iinclude -c $< -o $@

# (the host restatement of the preparation stage is test infrastructure too: the product prepares on the device)
$(B)/graph_input.o: tests/harness/graph_input.cpp tests/harness/graph_input.hpp $(wildcard $(HOST_DIR)/*.hpp)
	@mkdir -p $(B)
	$(CXX) $(CXXFLAGS) -Itests/harness -c $< -o $@

HOST_NOHIP_OBJS := $(filter-out $(B)/host/hip_backend.o $(B)/host/traverse_api.o,$(HOST_OBJS)) $(B)/graph_input.o
HARNESS := tests/harness/bin/oracle_graph_dump tests/harness/bin/libpagh_test.so tests/harness/bin/pagraph_oracle \
           tests/harness/bin/seg_kernels_test tests/harness/bin/sort_bench tests/harness/bin/libpagh_walk_test.so \
           tests/harness/bin/libpagh_stitch_test.so
harness: $(HARNESS)

# kernel-level check of K3/K4 against a sequential restatement (needs a GPU to run)
tests/harness/bin/seg_kernels_test: tests/harness/seg_kernels_test.hip $(HIP_DIR)/k34_segments.hip $(HIP_HDRS)
	@mkdir -p tests/harness/bin
	$(HIPCC) $(HIPFLAGS) -fno-PIC -o $@ $<

# timing + correctness aid for the radix sort (needs a GPU to run)
tests/harness/bin/sort_bench: tests/harness/sort_bench.hip $(HIP_DIR)/k2_sort.hip $(HIP_DIR)/util.hip $(HIP_HDRS)
	@mkdir -p tests/harness/bin
	$(HIPCC) $(HIPFLAGS) -fno-PIC -o $@ $<

# measurement variants of the scatter kernel (DESIGN §6 "the sort's bound"): the least a decoupled look-back would add to a tile
# (its digit row published, the rows of N predecessor tiles read back) and the payload staged as two u32 planes
tests/harness/bin/sort_bench_lb%: tests/harness/sort_bench.hip $(HIP_DIR)/k2_sort.hip $(HIP_DIR)/util.hip $(HIP_HDRS)
	@mkdir -p tests/harness/bin
	$(HIPCC) $(HIPFLAGS) -fno-PIC -DSORT_LOOKBACK_PROBE=$* -o $@ $<
tests/harness/bin/sort_bench_2plane: tests/harness/sort_bench.hip $(HIP_DIR)/k2_sort.hip $(HIP_DIR)/util.hip $(HIP_HDRS)
	@mkdir -p tests/harness/bin
	$(HIPCC) $(HIPFLAGS) -fno-PIC -DSORT_TWO_PLANE -o $@ $<
sort_variants: tests/harness/bin/sort_bench tests/harness/bin/sort_bench_lb4 tests/harness/bin/sort_bench_lb16 tests/harness/bin/sort_bench_lb64 tests/harness/bin/sort_bench_2plane

tests/harness/bin/oracle_graph_dump: tests/harness/oracle_graph_dump.cpp $(HOST_NOHIP_OBJS) $(B)/pag_oracle.o
	@mkdir -p tests/harness/bin
	$(CXX) $(CXXFLAGS) -Ioracle -Itests/harness -o $@ $< $(HOST_NOHIP_OBJS) $(B)/pag_oracle.o -lm -pthread

tests/harness/bin/libpagh_test.so: tests/harness/pagh_test.cpp $(HOST_NOHIP_OBJS)
	@mkdir -p tests/harness/bin
	$(CXX) $(CXXFLAGS) -Itests/harness -shared -o $@ $< $(HOST_NOHIP_OBJS) -pthread

# host restatement of the reference's traversal: test infrastructure, linked into harness programs only
$(B)/host_walk.o: tests/harness/host_walk.cpp tests/harness/host_walk.hpp $(wildcard $(HOST_DIR)/*.hpp)
	@mkdir -p $(B)
	$(CXX) $(CXXFLAGS) -Itests/harness -c $< -o $@

tests/harness/bin/pagraph_oracle: tests/harness/pagraph_oracle.cpp $(HOST_NOHIP_OBJS) $(B)/pag_oracle.o $(B)/host_walk.o
	@mkdir -p tests/harness/bin
	$(CXX) $(CXXFLAGS) -Ioracle -Itests/harness -o $@ $< $(HOST_NOHIP_OBJS) $(B)/pag_oracle.o $(B)/host_walk.o -lm -pthread

# device graph exported + host walk (needs the HIP library at run time): cross-check of the device walkers
tests/harness/bin/libpagh_walk_test.so: tests/harness/pagh_walk_test.cpp $(HOST_NOHIP_OBJS) $(B)/host_walk.o aligngraph2_amd/libpagraph_hip.so
	@mkdir -p tests/harness/bin
	$(CXX) $(CXXFLAGS) -Itests/harness -shared -o $@ $< $(HOST_NOHIP_OBJS) $(B)/host_walk.o -Laligngraph2_amd -lpagraph_hip -Wl,-rpath,'$$ORIGIN/../../../aligngraph2_amd' -pthread

# the host bookkeeping of walks cut into pieces (walk_stitch.hpp: plain C++, no device code) for CPU unit tests
tests/harness/bin/libpagh_stitch_test.so: tests/harness/stitch_test.cpp $(HIP_DIR)/walk_stitch.hpp
	@mkdir -p tests/harness/bin
	$(CXX) $(CXXFLAGS) -I$(HIP_DIR) -shared -o $@ $<

clean:
	rm -rf $(B) aligngraph2_amd/libpagraph_hip.so aligngraph2_amd/bin tests/harness/bin
	$(MAKE) -C oracle clean
