# Builds the product library (HIP, gfx950 only) and the test oracle.  No cmake: hipcc and g++ directly.
#   make            -> chad_tsdf_amd/lib/libchad_hip.so  +  oracle/_build/libchad_oracle.so
# -ffp-contract=off on host AND device: voxel traversal must round like the reference's strict-IEEE CPU build.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
HIPFLAGS ?= --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-result
CSRC := chad_tsdf_amd/csrc
LIB := chad_tsdf_amd/lib/libchad_hip.so
HDRS := $(wildcard $(CSRC)/*.hpp) include/chad_hip.h include/chad/tsdf.hpp

all: $(LIB) oracle

$(LIB): $(CSRC)/chad_hip.hip $(CSRC)/tsdf_shim.cpp $(HDRS)
	@mkdir -p chad_tsdf_amd/lib
	$(HIPCC) $(HIPFLAGS) -shared -Iinclude -o $@ $(CSRC)/chad_hip.hip -x hip $(CSRC)/tsdf_shim.cpp -ldl

oracle:
	$(MAKE) -C oracle -s

# C++ demo in the shape of the reference's src/chad/main.cpp (links the drop-in class)
demo: $(LIB) examples/sphere_demo.cpp
	$(HIPCC) -O2 -std=c++17 -Iinclude -o examples/sphere_demo examples/sphere_demo.cpp -Lchad_tsdf_amd/lib -lchad_hip -Wl,-rpath,'$$ORIGIN/../chad_tsdf_amd/lib'

asm: $(CSRC)/chad_hip.hip $(HDRS)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -Iinclude --cuda-device-only -S -o build/chad_hip.s $(CSRC)/chad_hip.hip -Rpass-analysis=kernel-resource-usage 2> build/resource_usage.txt || true

# CPU sanitizer configuration: the host-side C++ that runs without a GPU -- the oracle (oracle/chad_oracle.cpp) and
# the DAG fold of Submap::finalize (chad_tsdf_amd/csrc/dag.hpp through tests/cpp/dag_check.cpp), the planning of the sparse multi-GPU exchange (exchange_plan.hpp through tests/cpp/plan_check.cpp), the range bookkeeping of the DAG pool (range_pool.hpp through tests/cpp/pool_check.cpp) -- built with
# -fsanitize=address,undefined and driven by the CPU test suite (GPU AddressSanitizer is not available on this pool).
ASAN_RT := $(shell g++ -print-file-name=libasan.so):$(shell g++ -print-file-name=libubsan.so)
asan:
	$(MAKE) -C oracle -s asan
	CHAD_SANITIZE=1 LD_PRELOAD=$(ASAN_RT) ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
		python3 -m pytest tests/test_oracle_golden.py tests/test_oracle_survey_pins.py tests/test_dag_host.py tests/test_exchange_plan_host.py tests/test_range_pool_host.py tests/test_mc_table.py -x -q -p no:cacheprovider

clean:
	rm -rf chad_tsdf_amd/lib/*.so build examples/sphere_demo
	$(MAKE) -C oracle clean
.PHONY: all oracle demo asm asan clean
