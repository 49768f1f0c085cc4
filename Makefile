# Builds the product libraries in-tree (they travel to the GPU box with the snapshot):
#   gpufluidsimulation_amd/libbimocq_hip.so   C-ABI: gpu_* operators + fl_* runtime   (HIP, gfx950)
#   gpufluidsimulation_amd/libbimocq_host.so  C++ host solver (advance/outputResult) + bq_solver_* C API
# and, for tests only, the CPU oracle (oracle/Makefile).
#
# -ffp-contract=off is part of the numerics contract (bit parity with the oracle), not a tuning knob.
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
PKG      := gpufluidsimulation_amd
CSRC     := $(PKG)/csrc
OBJDIR   := build/obj
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -Iinclude
RCCL_LIB ?= -ldl    # RCCL itself is dlopen'ed by bq_halo.hip on first multi-GPU use
# make HAVE_OPENVDB=1: outputResult additionally writes real .vdb files (needs OpenVDB; absent in this image)
ifeq ($(HAVE_OPENVDB),1)
VDB_FLAGS := -DHAVE_OPENVDB
VDB_LIBS  := -lopenvdb -ltbb
endif

KERNEL_SRCS := $(sort $(wildcard $(CSRC)/*.hip))
KERNEL_OBJS := $(patsubst $(CSRC)/%.hip,$(OBJDIR)/%.o,$(KERNEL_SRCS))
HOST_SRCS   := $(wildcard $(CSRC)/host/*.cpp)
HOST_OBJS   := $(patsubst $(CSRC)/host/%.cpp,$(OBJDIR)/host_%.o,$(HOST_SRCS))

all: $(PKG)/libbimocq_hip.so $(PKG)/libbimocq_host.so oracle example

KERNEL_HDRS := $(wildcard $(CSRC)/*.h) $(wildcard $(CSRC)/*.inc)

$(OBJDIR)/%.o: $(CSRC)/%.hip $(KERNEL_HDRS) include/bimocq_gpu.h
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

# the gather kernels a second time with one-fma lerps (bq_device.hip.h: inline namespace bq::fast, entry points *_fast)
$(OBJDIR)/bq_advect_fast.o: $(CSRC)/bq_advect.hip $(KERNEL_HDRS) include/bimocq_gpu.h
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -DBQ_FAST_LERP -c $< -o $@

$(PKG)/libbimocq_hip.so: $(KERNEL_OBJS) $(OBJDIR)/bq_advect_fast.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(KERNEL_OBJS) $(OBJDIR)/bq_advect_fast.o $(RCCL_LIB)

$(OBJDIR)/host_%.o: $(CSRC)/host/%.cpp $(wildcard $(CSRC)/host/*.hpp) include/bimocq_gpu.h include/bimocq_solver.h
	@mkdir -p $(OBJDIR)
	g++ -O2 -std=c++17 -fPIC -pthread -Wall -Wextra -Iinclude $(VDB_FLAGS) -c $< -o $@

# the host solver only speaks the C-ABI: it links against libbimocq_hip.so by name ($$ORIGIN rpath)
$(PKG)/libbimocq_host.so: $(HOST_OBJS) $(PKG)/libbimocq_hip.so
	g++ -shared -fPIC -pthread -o $@ $(HOST_OBJS) -L$(PKG) -lbimocq_hip $(VDB_LIBS) -Wl,-rpath,'$$ORIGIN'

# the reference's driver loop (main.cpp:137-159) on this library
example: build/bimocq3d build/bimocq3d_ranks
build/bimocq3d: examples/bimocq3d_main.cpp $(PKG)/libbimocq_host.so
	g++ -O2 -std=c++17 -pthread -Iinclude -I$(CSRC)/host $< -o $@ -L$(PKG) -lbimocq_host -lbimocq_hip -Wl,-rpath,'$$ORIGIN/../$(PKG)'

# the same loop as one rank of an N-GPU run, C++ only (RANK / WORLD_SIZE from the environment, ncclUniqueId through a file)
build/bimocq3d_ranks: examples/bimocq3d_ranks.cpp $(PKG)/libbimocq_host.so
	g++ -O2 -std=c++17 -pthread -Iinclude -I$(CSRC)/host $< -o $@ -L$(PKG) -lbimocq_host -lbimocq_hip -Wl,-rpath,'$$ORIGIN/../$(PKG)'

oracle:
	$(MAKE) -s -C oracle

# AddressSanitizer + UndefinedBehaviorSanitizer on the CPU build.  The product's C++ host layer
# (csrc/host/*.cpp: ghost-plane validity tracking, wall-sheet boxes, plane windows -- index arithmetic throughout), the
# test-only CPU stand-in of the C-ABI and the oracle are rebuilt with -fsanitize=address,undefined and the CPU tests that
# drive them run under it: the host state machine, the multi-rank slab logic over gloo, the oracle's known answers and
# fixtures.  libasan has to be the first library of the interpreter, hence LD_PRELOAD; leak checking is off (CPython and
# torch never free everything).  GPU code cannot run under ASan on this pool (no xnack).  Summary: profiles/r04_sanitize_summary.txt
SAN_TESTS ?= tests/test_host_logic_cpu.py tests/test_slab_multirank.py tests/test_oracle_kat.py tests/test_oracle_mgcg.py tests/test_golden.py
sanitize:
	BQ_SANITIZE=1 LD_PRELOAD="$$(gcc -print-file-name=libasan.so) $$(gcc -print-file-name=libubsan.so)" \
	ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
	python -m pytest $(SAN_TESTS) -x -q -m "not gpu" -p no:cacheprovider

clean:
	rm -rf build $(PKG)/*.so oracle/_build oracle/_build_san tests/_build tests/_build_san

.PHONY: all oracle clean example sanitize
