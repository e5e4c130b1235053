# Build libhdpgpc_hip.so (gfx950).  `python -c "import __graft_entry__ as g; g.build()"` runs this.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC   = hdpgpc_amd/csrc
SRCS   = $(CSRC)/hgp_kernels.hip $(CSRC)/hgp_pairs.hip $(CSRC)/hgp_pairs_acc.hip $(CSRC)/hgp_matlik.hip $(CSRC)/hgp_matlik_coop.hip $(CSRC)/hgp_assign.hip $(CSRC)/hgp_warp.hip $(CSRC)/hgp_chain.hip
HDR    = $(CSRC)/tile_f64.hpp $(CSRC)/hgp_internal.hpp include/hdpgpc_hip.h
OBJDIR = build/obj
OBJS   = $(patsubst $(CSRC)/%.hip,$(OBJDIR)/%.o,$(SRCS))
LIB    = hdpgpc_amd/lib/libhdpgpc_hip.so
FLAGS  = -O3 --offload-arch=$(ARCH) -mllvm -pragma-unroll-threshold=1048576 -fPIC -Wno-unused-result

all: $(LIB)

$(OBJDIR)/%.o: $(CSRC)/%.hip $(HDR)
	mkdir -p $(OBJDIR)
	$(HIPCC) $(FLAGS) -c -o $@ $<

$(LIB): $(OBJS)
	mkdir -p hdpgpc_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

# diagnostic build with in-kernel cycle stamps (tools/stamps.py); never used by the product path
stamps: $(SRCS) $(HDR)
	mkdir -p hdpgpc_amd/lib/ab
	$(HIPCC) $(FLAGS) -DHGP_STAMPS -shared -o hdpgpc_amd/lib/ab/libhgp_stamps.so $(SRCS)

# diagnostic builds for the cooperative-factor race (tile_f64.hpp, coop_factor): a delay injected in front of the
# right-hand-side row update, with the pre-round-2 round-robin dealing (races) and with the owner dealing (immune)
raceprobe: $(SRCS) $(HDR)
	mkdir -p build/probe
	$(HIPCC) $(FLAGS) -DHGP_RACE_PROBE_DELAY -DHGP_RACE_PROBE_ROUNDROBIN -shared -o build/probe/libhgp_race_old.so $(SRCS)
	$(HIPCC) $(FLAGS) -DHGP_RACE_PROBE_DELAY -shared -o build/probe/libhgp_race_new.so $(SRCS)

# every workgroup barrier followed by a pseudo-random per-wave delay: run the GPU tests with HGP_LIB pointing at it
racestress: $(SRCS) $(HDR)
	mkdir -p build/probe
	$(HIPCC) $(FLAGS) -DHGP_RACE_STRESS -shared -o hdpgpc_amd/lib/ab/libhgp_race_stress.so $(SRCS)

# in-situ knock-out builds of k_pairs (results wrong by construction, only the time matters; tools/knockout_time.py):
# one component replaced by a stub each - what the component costs INSIDE the kernel, not in isolation
KNOCKOUTS = NOEXP NODIAG NORHS NOFACTOR NOFILL NOAF SHARED_M
knockouts: $(LIB)
	mkdir -p build/probe
	for v in $(KNOCKOUTS); do \
	  $(HIPCC) $(FLAGS) -DHGP_EXP_$$v -c -o build/probe/hgp_pairs_$$v.o $(CSRC)/hgp_pairs.hip && \
	  $(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o build/probe/libhgp_exp_$$v.so $(filter-out $(OBJDIR)/hgp_pairs.o,$(OBJS)) build/probe/hgp_pairs_$$v.o || exit 1; \
	done

clean:
	rm -rf $(LIB) $(OBJDIR)
