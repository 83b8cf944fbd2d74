# Builds the gfx950 shared library and the C oracle pieces.  hipcc cross-compiles without a GPU.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
LIB    = bde2vid_amd/libbde2vid.so
SRC    = bde2vid_amd/csrc/bde_api.hip
HDR    = $(wildcard bde2vid_amd/csrc/*.h) include/bde2vid.h

all: $(LIB)

$(LIB): $(SRC) $(HDR)
	$(HIPCC) -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -shared -Wall -Wno-unused-function \
	    -fvisibility=hidden -DBDE_BUILD -o $@ $(SRC)

clean:
	rm -f $(LIB)
