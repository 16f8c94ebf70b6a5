#!/bin/bash
# Registers / scratch (spill) bytes per kernel, from hipcc's resource-usage remarks (no GPU needed).
#   tools/kernel_resources.sh fa_fwd_mfma fa_bwd_dkdv_mfma fa_bwd_dq_mfma
cd "$(dirname "$0")/../flashattention-pytorch_amd/csrc" || exit 1
for f in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -munsafe-fp-atomics -c "$f.hip" -o /dev/null \
      -Rpass-analysis=kernel-resource-usage 2>&1 |
    awk '/Function Name:/ {name=$5} /VGPRs:/ {v=$4} /ScratchSize/ {print name, "vgpr=" v, "scratch=" $5}' |
    sed 's/_ZN2fa[0-9]*//; s/EEvPK.*ff[i]* / /; s/INS_//; s/_tagE/ /' | sort -u
done
