# Builds the product library (HIP, gfx950) and the test oracle with plain make.  configure.ac / Makefile.am at
# the repo root wrap the same rules for autotools trees (unexercised here: no autoreconf in the image).
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
HIPFLAGS ?= -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-function
CSRC     := hashmergejoin_amd/csrc
OBJS     := $(CSRC)/radix.o $(CSRC)/probe.o $(CSRC)/gen.o $(CSRC)/gtable.o $(CSRC)/api.o $(CSRC)/exchange.o
LIB      := hashmergejoin_amd/libhmj_hip.so

all: $(LIB) oracle cpptest tests/cpp/strgen_bench examples/hashjoin_bench_hip examples/exchange_join

$(CSRC)/%.o: $(CSRC)/%.hip $(CSRC)/hmj_dev.h $(CSRC)/hmj_launch.h $(CSRC)/hmj_ctx.h include/hmj.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -ldl

oracle:
	$(MAKE) -C oracle all

# C++ drop-in test: host-only g++ against the C ABI (+ the oracle as checker)
cpptest: tests/cpp/test_dropin
tests/cpp/test_dropin: tests/cpp/test_dropin.cc include/hashmergejoin_hip.hpp include/hmj.h oracle/strgen_restated.h $(LIB) oracle
	g++ -std=c++11 -O2 -Wall -Iinclude -Ioracle $< -o $@ -Lhashmergejoin_amd -lhmj_hip -Loracle -lhmj_oracle \
	  -Wl,-rpath,'$$ORIGIN/../../hashmergejoin_amd' -Wl,-rpath,'$$ORIGIN/../../oracle' -Wl,-rpath,/opt/rocm/lib -pthread

# BASELINE configs[0] through the drop-in (test infrastructure: the strgen generator restated in oracle/, the product header)
tests/cpp/strgen_bench: tests/cpp/strgen_bench.cc include/hashmergejoin_hip.hpp include/hmj.h oracle/strgen_restated.h $(LIB)
	g++ -std=c++11 -O2 -Wall -Iinclude -Ioracle $< -o $@ -Lhashmergejoin_amd -lhmj_hip \
	  -Wl,-rpath,'$$ORIGIN/../../hashmergejoin_amd' -Wl,-rpath,/opt/rocm/lib -pthread

examples/hashjoin_bench_hip: examples/hashjoin_bench_hip.cc include/hashmergejoin_hip.hpp include/hmj.h $(LIB)
	g++ -std=c++11 -O2 -Wall -Iinclude $< -o $@ -Lhashmergejoin_amd -lhmj_hip \
	  -Wl,-rpath,'$$ORIGIN/../hashmergejoin_amd' -Wl,-rpath,/opt/rocm/lib -pthread

examples/exchange_join: examples/exchange_join.cc include/hmj.h $(LIB)
	g++ -std=c++11 -O2 -Wall -Iinclude $< -o $@ -Lhashmergejoin_amd -lhmj_hip -L/opt/rocm/lib -lamdhip64 \
	  -Wl,-rpath,'$$ORIGIN/../hashmergejoin_amd' -Wl,-rpath,/opt/rocm/lib -pthread

clean:
	rm -f $(OBJS) $(LIB) tests/cpp/test_dropin tests/cpp/strgen_bench
	$(MAKE) -C oracle clean

.PHONY: all oracle cpptest clean
