#!/bin/bash
# Repro of a compiler crash worked around in csrc/ipx_inflate.hip (VERDICT round 4 asked for it to be kept).
# ROCm 7.2.0, clang 22, --offload-arch=gfx950: giving k_bgzf_inflate a second launch bound -- `__launch_bounds__(64, 3)`,
# i.e. exactly the three waves per SIMD its 162 registers allow anyway -- crashes the PostRA Machine Instruction Scheduler
# (llvm::ScheduleDAGInstrs::addVRegUseDeps, a segmentation fault while it builds the scheduling graph of that kernel).
# With `__launch_bounds__(64)` the file compiles; the launcher asks hipOccupancyMaxActiveBlocksPerMultiprocessor instead.
# usage (from the repository root; no GPU needed): bash profiles/experiments/r05_compiler_crash_launch_bounds.sh
set -u
D=$(mktemp -d)
sed 's/__global__ __launch_bounds__(64) void k_bgzf_inflate(/__global__ __launch_bounds__(64, 3) void k_bgzf_inflate(/' \
    intronprospector_amd/csrc/ipx_inflate.hip > $D/ipx_inflate_lb.hip
grep -c "__launch_bounds__(64, 3) void k_bgzf_inflate" $D/ipx_inflate_lb.hip
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Iintronprospector_amd/csrc -c $D/ipx_inflate_lb.hip -o $D/x.o 2>&1 \
    | grep -E "Running pass|addVRegUseDeps|buildSchedGraph|PLEASE submit" | head -8
echo "hipcc exit status: ${PIPESTATUS[0]}"
rm -rf $D
