#!/bin/bash
# usage: tools/isa.sh <source.hip under e-d3dgs_amd/csrc> <mangled-name regex>...   -- device ISA of the file into /tmp/<name>.s and, per
# kernel matching a regex, its loop waits / registers / LDS (what the r2 work on LDS-DMA waits was steered by)
src=$1; shift
cd /root/repo/e-d3dgs_amd/csrc || exit 1
extra=""; [ "$src" = preprocess.hip ] && extra="-ffp-contract=off"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -munsafe-fp-atomics $extra -S --cuda-device-only -o /tmp/${src%.hip}.s $src 2>&1 | grep -v "warning: argument unused" | head -20
for re in "$@"; do
  for sym in $(grep -oE "^_Z[A-Za-z0-9_]*:" /tmp/${src%.hip}.s | tr -d : | grep -E "$re"); do
    awk "/^$sym:/,/s_endpgm/" /tmp/${src%.hip}.s > /tmp/k_$sym.s
    echo "== $sym ($(wc -l < /tmp/k_$sym.s) lines) mfma $(grep -c v_mfma /tmp/k_$sym.s) vmcnt-waits at: $(grep -n 's_waitcnt vmcnt' /tmp/k_$sym.s | cut -d: -f1 | tr '\n' ' ') barriers at: $(grep -n s_barrier /tmp/k_$sym.s | cut -d: -f1 | tr '\n' ' ')"
    grep -A40 "amdhsa_kernel $sym\$" /tmp/${src%.hip}.s | grep -E "group_segment_fixed|next_free_vgpr|private_segment_fixed" | tr -d '\t' | tr '\n' ' '; echo
  done
done
