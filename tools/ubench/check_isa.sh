#!/bin/bash
# Counts, per microbenchmark kernel, the VALU opcodes inside its timed loop: the body must hold CH = 16 of the instruction the row
# is named after and nothing else from the VALU.  Runs without a GPU:  bash tools/ubench/check_isa.sh
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S valu_rates.hip -o /tmp/valu_rates.s 2>/dev/null
awk '/^_Z[0-9]+k_[a-z0-9_]+Pjj:/{name=$1; inloop=0} /^.LBB[0-9_]+:/{if(name!="") inloop=1} inloop && /^\tv_/{c[name" "$1]++} /s_cbranch_scc[01]/{inloop=0} /s_endpgm/{name=""} END{for(k in c) print k, c[k]}' /tmp/valu_rates.s | sort
