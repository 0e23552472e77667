# Builds the product library (HIP, gfx950) and the test oracle.  No autotools needed; the
# autotools files under build-aux/ wrap the same rules for trees that use them.
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
HIPFLAGS ?= -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-function
CSRC     := hashmergejoin_amd/csrc
OBJS     := $(CSRC)/radix.o $(CSRC)/probe.o $(CSRC)/gen.o $(CSRC)/api.o
LIB      := hashmergejoin_amd/libhmj_hip.so

all: $(LIB) oracle

$(CSRC)/%.o: $(CSRC)/%.hip $(CSRC)/hmj_dev.h $(CSRC)/hmj_launch.h include/hmj.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

oracle:
	$(MAKE) -C oracle all

examples: $(LIB)
	$(MAKE) -C examples

clean:
	rm -f $(OBJS) $(LIB)
	$(MAKE) -C oracle clean

.PHONY: all oracle examples clean
