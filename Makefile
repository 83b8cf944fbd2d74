# Builds the gfx950 shared library.  hipcc cross-compiles without a GPU.  Three translation units (the fp32 conv-shaped
# kernels, the split-operand kernels and everything else) compile in parallel: `make -j3`.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
LIB    = bde2vid_amd/libbde2vid.so
SRCS   = bde2vid_amd/csrc/bde_api.hip bde2vid_amd/csrc/conv_tu.hip bde2vid_amd/csrc/sb_tu.hip
OBJS   = $(SRCS:bde2vid_amd/csrc/%.hip=build/%.o)
# (per-unit header dependencies from the compiler: a header edit rebuilds only the units that include it)
FLAGS  = -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -Wall -Wno-unused-function -fvisibility=hidden -DBDE_BUILD -MMD -MP

all:
	@$(MAKE) --no-print-directory -j3 $(LIB)

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

build/%.o: bde2vid_amd/csrc/%.hip include/bde2vid.h
	@mkdir -p build
	$(HIPCC) $(FLAGS) -c -o $@ $<

-include $(OBJS:.o=.d)

clean:
	rm -rf build $(LIB)
