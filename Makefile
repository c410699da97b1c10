# Build everything in-tree.  `make` = scene generator + oracle (CPU) + HIP library (gfx950).
#   libbrmi.so        HIP kernels + C ABI (include/brmi.h)           -> basicrenderer_amd/lib/
#   libbrmi_scene.so  procedural scene generator (host only)          -> basicrenderer_amd/lib/
#   libbrmi_compose.so  RCCL composition of the row-band partition (include/brmi_compose.h) -> basicrenderer_amd/lib/
#   liboracle.so      scalar CPU restatement of the reference shaders -> oracle/_build/   (tests only)
ROCM      ?= /opt/rocm
HIPCC     ?= $(ROCM)/bin/hipcc
CXX       ?= g++
LIBDIR    := basicrenderer_amd/lib
ORCDIR    := oracle/_build

# Strict IEEE arithmetic everywhere: no FMA contraction, no fast-math, correctly rounded div/sqrt.
# -fno-slp-vectorize: the SLP vectoriser pairs scalar fp32 operations into v_pk_* instructions; on these kernels the v_movs that build the
# register pairs cost more issue slots than the packed operations save (k_gbuffer 104 -> 93 us, k_shade 237 -> 233 us; same bits).
EXTRA     ?=
HIPFLAGS  := $(EXTRA) --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fno-slp-vectorize \
             -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -Iinclude -Wall
ORCFLAGS  := -O2 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fopenmp -Iinclude -Wall
SCNFLAGS  := -O2 -std=c++17 -fPIC -shared -Iinclude -Wall -fopenmp

HIP_SRCS  := $(wildcard basicrenderer_amd/csrc/*.hip)
HIP_HDRS  := $(wildcard basicrenderer_amd/csrc/*.h) $(wildcard include/*.h)
ORC_SRCS  := $(wildcard oracle/*.cpp)
ORC_HDRS  := $(wildcard oracle/*.h) $(wildcard include/*.h)

all: scene oracle hip compose host_example

scene: $(LIBDIR)/libbrmi_scene.so
oracle: $(ORCDIR)/liboracle.so
hip: $(LIBDIR)/libbrmi.so
compose: $(LIBDIR)/libbrmi_compose.so

SCN_SRCS  := basicrenderer_amd/csrc/scene/scene_gen.cpp basicrenderer_amd/csrc/scene/lod_builder.cpp
$(LIBDIR)/libbrmi_scene.so: $(SCN_SRCS) include/brmi_scene.h include/brmi_types.h
	@mkdir -p $(LIBDIR)
	$(CXX) $(SCNFLAGS) $(SCN_SRCS) -o $@

$(ORCDIR)/liboracle.so: $(ORC_SRCS) $(ORC_HDRS)
	@mkdir -p $(ORCDIR)
	$(CXX) $(ORCFLAGS) $(ORC_SRCS) -o $@

# one object per translation unit (build/, git-ignored): an edit recompiles its own file only, and `make -j` compiles them side by side
OBJDIR    := build/hip
HIP_OBJS  := $(patsubst basicrenderer_amd/csrc/%.hip,$(OBJDIR)/%.o,$(HIP_SRCS))
HIPCFLAGS := $(filter-out -shared,$(HIPFLAGS))
$(OBJDIR)/%.o: basicrenderer_amd/csrc/%.hip $(HIP_HDRS) Makefile
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPCFLAGS) -c $< -o $@
$(LIBDIR)/libbrmi.so: $(HIP_OBJS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC $(HIP_OBJS) -o $@

# multi-GPU composition: the only library that links RCCL (include/brmi_compose.h)
$(LIBDIR)/libbrmi_compose.so: basicrenderer_amd/csrc/compose/brmi_compose.hip include/brmi_compose.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(EXTRA) --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude -I$(ROCM)/include -Wall $< -L$(ROCM)/lib -lrccl -o $@

# C++ host driving the passes through basicrenderer_amd/host/brmi_passes.hpp (links both libraries)
host_example: $(LIBDIR)/brmi_host_frame
$(LIBDIR)/brmi_host_frame: examples/host_frame.cpp basicrenderer_amd/host/brmi_passes.hpp $(LIBDIR)/libbrmi.so $(LIBDIR)/libbrmi_scene.so
	$(HIPCC) -O2 -std=c++17 -Iinclude examples/host_frame.cpp -L$(LIBDIR) -lbrmi -lbrmi_scene -Wl,-rpath,'$$ORIGIN' -o $@

# tools/valu_issue_probe.hip: what a SIMD issues per cycle (profiles/r03_valu_issue_probe.txt is its output on an MI355X)
probe: build/valu_issue_probe
build/valu_issue_probe: tools/valu_issue_probe.hip
	@mkdir -p build
	$(HIPCC) --offload-arch=gfx950 -O2 $< -o $@

clean:
	rm -rf $(LIBDIR) $(ORCDIR) build

.PHONY: all scene oracle hip compose host_example probe clean
