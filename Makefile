# Build libhdpgpc_hip.so (gfx950) and the CPU oracle helpers.  `python -c "import __graft_entry__ as g; g.build()"` runs this.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
SRC    = hdpgpc_amd/csrc/hgp_kernels.hip
HDR    = hdpgpc_amd/csrc/tile_f64.hpp include/hdpgpc_hip.h
LIB    = hdpgpc_amd/lib/libhdpgpc_hip.so

all: $(LIB)

$(LIB): $(SRC) $(HDR)
	mkdir -p hdpgpc_amd/lib
	$(HIPCC) -O3 --offload-arch=$(ARCH) -mllvm -pragma-unroll-threshold=1048576 -shared -fPIC -Wno-unused-result -o $@ $(SRC)

# diagnostic build with in-kernel cycle stamps (tools/stamps.py); never used by the product path
stamps: $(SRC) $(HDR)
	$(HIPCC) -O3 --offload-arch=$(ARCH) -mllvm -pragma-unroll-threshold=1048576 -DHGP_STAMPS -shared -fPIC -Wno-unused-result -o hdpgpc_amd/lib/libhdpgpc_hip_stamps.so $(SRC)

clean:
	rm -f $(LIB)
