# Top-level build: everything is compiled in-tree so the built libraries travel with the source.
#
#   make lib      multigrid_amd/libmgrt.so      gfx950 HIP kernels + C ABI (hipcc cross-compiles without a GPU)
#   make host     multigrid_amd/host/multigrid2D, multigrid3D   C++ drop-in classes + REPL over the C ABI
#   make oracle   oracle/libmg_oracle.so        CPU restatement (test infrastructure)
#   make ref      oracle/_ref/libmg_ref.so      reference kernels for x86-64 (only where /root/reference exists)

HIPCC    ?= hipcc
ARCH     ?= gfx950
# -ffp-contract=off: results must match the reference bit for bit; a fused multiply-add would not.
HIPFLAGS  = --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -fPIC -Iinclude -Wall -Wno-unused-function
CSRC      = multigrid_amd/csrc
LIB       = multigrid_amd/libmgrt.so
LIB_SRCS  = $(CSRC)/mgrt_runtime.hip $(CSRC)/mgrt_kernels.hip $(CSRC)/mgrt_smooth_fused.hip $(CSRC)/mgrt_smooth_pair.hip $(CSRC)/mgrt_smooth2d.hip $(CSRC)/mgrt_bottom.hip $(CSRC)/mgrt_plan.hip $(CSRC)/mgrt_slab.hip
LIB_OBJS  = $(LIB_SRCS:.hip=.o)

all: lib oracle host

lib: $(LIB)

$(CSRC)/%.o: $(CSRC)/%.hip $(CSRC)/mgrt_internal.h $(CSRC)/mgrt_device.h include/mgrt.h
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

$(LIB): $(LIB_OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -o $@ $(LIB_OBJS) -ldl

# diagnostic build with in-kernel time stamps in the fused sweep (tools/stamps.py); not the product
stamps:
	$(HIPCC) $(HIPFLAGS) -I$(CSRC) -shared -o multigrid_amd/libmgrt_stamps.so $(filter-out $(CSRC)/mgrt_smooth_fused.hip,$(LIB_SRCS)) tools/diag/mgrt_stamps.hip -ldl

oracle:
	$(MAKE) -C oracle

ref:
	$(MAKE) -C oracle ref

host: lib
	@if [ -f multigrid_amd/host/Makefile ]; then $(MAKE) -C multigrid_amd/host; fi

clean:
	rm -f $(LIB_OBJS) $(LIB)
	$(MAKE) -C oracle clean

.PHONY: all lib stamps oracle ref host clean
