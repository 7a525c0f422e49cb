#!/bin/bash
# Register / LDS / spill report of every gfx950 kernel in a HIP source (no GPU needed):
#   tools/kernel_resources.sh stormphrax_amd/csrc/spx_kernels.hip [extra hipcc flags]
set -e
src=$1; shift
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I"$(dirname "$src")" --cuda-device-only -save-temps=obj -c "$src" -o "$tmp/k.o" "$@" 2>/dev/null
s=$(ls "$tmp"/*.s | head -1)
awk '/^[ \t]*\.amdhsa_kernel /{name=$2} /\.amdhsa_next_free_vgpr/{v=$2} /\.amdhsa_group_segment_fixed_size/{l=$2} /\.amdhsa_private_segment_fixed_size/{p=$2} /^[ \t]*\.end_amdhsa_kernel/{printf "%-110s vgpr %4s lds %6s scratch %5s\n", name, v, l, p}' "$s" | sed 's/_ZN3spx[0-9]*//'
rm -rf "$tmp"
