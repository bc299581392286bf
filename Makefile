# Builds the MI355X library, the plain-C driver and the CPU oracle (the same commands __graft_entry__.build() runs).
HIPCC ?= hipcc
PKG := extendedrtirtmodeling.jl_amd
LIB := $(PKG)/libertirt.so

all: $(LIB) tools/erm_cli oracle/liberm_oracle.so

$(LIB): $(PKG)/csrc/ertirt.hip $(PKG)/csrc/erm_kernels.hpp $(PKG)/csrc/erm_rng.hpp $(PKG)/csrc/erm_layout.hpp $(PKG)/csrc/erm_geometry.hpp include/ertirt.h
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -I include $(PKG)/csrc/ertirt.hip -o $@ $(ERM_HIPCC_FLAGS)

tools/erm_cli: tools/erm_cli.c include/ertirt.h $(LIB)
	gcc -O2 -Wall -I include tools/erm_cli.c -L $(PKG) -lertirt -lm -Wl,-rpath,'$$ORIGIN/../$(PKG)' -o $@

oracle/liberm_oracle.so: oracle/erm_oracle.c oracle/orc_rng.h
	$(MAKE) -C oracle -s

clean:
	rm -f $(LIB) tools/erm_cli oracle/liberm_oracle.so

.PHONY: all clean
