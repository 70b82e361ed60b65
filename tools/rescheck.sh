#!/bin/bash
# usage: rescheck.sh TX HT [extra flags]  -> resource usage of the fp16x3 kernels of that object
TX=$1; HT=$2; shift 2
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}/stribor_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize -mllvm -amdgpu-atomic-optimizer-strategy=None -DSX_F16X3 -DSX_TX=$TX -DSX_HT=$HT "$@" -Rpass-analysis=kernel-resource-usage -c sx_flow_inst.hip -o /tmp/rescheck_$$.o 2>&1 | grep -E "Function Name|VGPRs:|AGPRs|ScratchSize" | paste - - - - | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g; s/.\/sx_flow_kernel.h:[0-9]*:1: remark: //g; s/Function Name: _ZN8sx_f16x317flow_fused_kernelILi1E//; s/EEv5dprogNS_10flow_kargsE//'
rm -f /tmp/rescheck_$$.o
