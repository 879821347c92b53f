# Builds the C-ABI shared library of the hot path for gfx950 (MI355X).  hipcc cross-compiles
# without a GPU.  `make` -> nerf-ca_amd/lib/libnerfca_hip.so
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := nerf-ca_amd/csrc
OUT   := nerf-ca_amd/lib/libnerfca_hip.so
SRCS  := $(CSRC)/nca_api.hip $(CSRC)/nca_kernels_f32.hip $(CSRC)/nca_kernels_bf16.hip $(CSRC)/nca_kernels_loss.hip $(CSRC)/nca_kernels_wide.hip
HDRS  := include/nerfca_hip.h $(CSRC)/nca_layout.hpp $(CSRC)/nca_kernels.hpp $(CSRC)/nca_rng.hpp
OBJS  := $(SRCS:.hip=.o)
FLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function

all: $(OUT)

# headers only some translation units include
$(CSRC)/nca_kernels_f32.o: $(CSRC)/nca_f32_layer.inc
$(CSRC)/nca_kernels_wide.o $(CSRC)/nca_api.o: $(CSRC)/nca_wide.hpp
$(CSRC)/nca_api.o: $(CSRC)/nca_api_wide.inc

%.o: %.hip $(HDRS)
	$(HIPCC) $(FLAGS) $(EXTRA) -c $< -o $@

$(OUT): $(OBJS)
	@mkdir -p $(dir $@)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(OBJS) -o $@

clean:
	rm -f $(OBJS) $(OUT)

.PHONY: all clean
