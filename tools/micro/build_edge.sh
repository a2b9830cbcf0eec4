#!/bin/bash
# builds the edge-stack harness: plain and with phase stamps
D=$(dirname "$0")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -Wno-unused-value $EXTRA "$D/edge_stack.hip" -o "$D/edge_stack" -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|VGPRs:|Spill|ScratchSize|AGPRs" 
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -Wno-unused-value -DSTAMPS $EXTRA "$D/edge_stack.hip" -o "$D/edge_stack_st" 2>&1 | grep -E "error" -A3
