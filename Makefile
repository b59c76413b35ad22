# Build the MI355X (gfx950) shared library of the Voice100 hot path and the CPU oracle helpers.
#   make            -> voice100_amd/libvoice100_hip.so  (+ oracle/_build/libconv_ref.so)
# No torch, no cmake: plain hipcc; the .so travels to the GPU box with the snapshot.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := voice100_amd/csrc
OBJD  := build/obj
SRCS  := $(wildcard $(CSRC)/*.hip)
OBJS  := $(patsubst $(CSRC)/%.hip,$(OBJD)/%.o,$(SRCS))
LIB   := voice100_amd/libvoice100_hip.so
# the depthwise window walk is one fully unrolled body of up to 8*83 FMAs: lift clang's cap on '#pragma unroll'
# -fno-slp-vectorize: hipcc otherwise pairs adjacent fp32 FMAs into v_pk_fma_f32, which is no faster on gfx950 and costs
# aligned register pairs + moves (the fused depthwise backward went from 215 to 207 VGPRs and lost 330 packed ops)
HIPFLAGS := -O3 --offload-arch=$(ARCH) -fPIC -std=c++17 -Iinclude -I$(CSRC) -Wno-unused-result -mllvm -pragma-unroll-threshold=1000000 -fno-slp-vectorize $(EXTRA)

all: $(LIB) oracle

$(OBJD)/%.o: $(CSRC)/%.hip $(wildcard $(CSRC)/*.h)
	@mkdir -p $(OBJD)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf build $(LIB)
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
