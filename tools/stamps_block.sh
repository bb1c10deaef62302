#!/bin/bash
# phase stamps of the RTI kernels (diagnostic build path, ALORE_NMPC_STAMPS=1): bench.py --lanes <encoded> --batch B
# usage: tools/stamps_block.sh "<lanes list>" "<batch list>"
for L in $1; do for B in $2; do
echo "=== lanes=$L batch=$B"
ALORE_NMPC_STAMPS=1 timeout 300 python bench.py --lanes $L --batch $B --no-cpu-baseline --no-extras --no-graph --steps 20 --warmup 2 2>&1 | grep -v "^{" | grep -v amdgpu.ids
done; done
