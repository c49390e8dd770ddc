# Plain-make build of the product library for hosts without Python (the driver and the tests use dv-pari_amd/build.py, which
# runs the same hipcc lines one after the other):   make -j8        -> dv-pari_amd/libdvpari_hip.so
#                                                    make cli        -> examples/dvp_prove_cli (g++, public header only)
#                                                    make oracle     -> the CPU oracle (test infrastructure, oracle/Makefile)
HIPCC  ?= /opt/rocm/bin/hipcc
ARCH   ?= gfx950
CSRC   := dv-pari_amd/csrc
SRCS   := capi.cpp cache.cpp tree_io.cpp ecfft.hip msm.hip codec.hip fr_ops.hip prove.hip setup.hip
OBJS   := $(addprefix $(CSRC)/,$(addsuffix .o,$(basename $(SRCS))))
HDRS   := $(wildcard $(CSRC)/*.h $(CSRC)/*.cuh include/*.h)
LIB    := dv-pari_amd/libdvpari_hip.so
FLAGS  := -O3 --offload-arch=$(ARCH) -std=c++17 -fPIC -Wno-pass-failed -Wno-int-to-pointer-cast

all: $(LIB)
$(CSRC)/%.o: $(CSRC)/%.hip $(HDRS)
	$(HIPCC) $(FLAGS) -c -x hip $< -o $@
$(CSRC)/%.o: $(CSRC)/%.cpp $(HDRS)
	$(HIPCC) $(FLAGS) -c -x hip $< -o $@
$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)
cli: $(LIB)
	g++ -O2 -std=c++17 -Iinclude examples/dvp_prove_cli.cpp -Ldv-pari_amd -ldvpari_hip -Wl,-rpath,$(abspath dv-pari_amd) -pthread -o examples/dvp_prove_cli
oracle:
	$(MAKE) -C oracle
clean:
	rm -f $(OBJS) $(LIB) examples/dvp_prove_cli
.PHONY: all cli oracle clean

# Host-side parsers under AddressSanitizer + UBSan (CPU build only: GPU entries are stand-ins, tools/asan/stubs.cpp):
#   make asan        -> builds and runs tools/asan/fuzz with ASAN_ITERS cases (default 20000); see tools/README.md
ASAN_ITERS ?= 20000
asan:
	g++ -std=c++17 -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=all -D__HIP_PLATFORM_AMD__ \
	  -I/opt/rocm/include $(CSRC)/cache.cpp $(CSRC)/tree_io.cpp tools/asan/stubs.cpp tools/asan/fuzz.cpp -pthread -o tools/asan/fuzz
	ASAN_OPTIONS=detect_leaks=1:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 ./tools/asan/fuzz $(ASAN_ITERS)
.PHONY: asan
