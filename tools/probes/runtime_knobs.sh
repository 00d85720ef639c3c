#!/bin/bash
# Same-box sweep of HIP-runtime environment switches over the recorded B = 8 and B = 1 tiled calls (stream replay): which of them
# move the GPU-side cost of a dependent launch.  usage: bash tools/probes/runtime_knobs.sh "VAR=val VAR2=val ..."  (each run alone, baseline between)
KNOBS=${1:-"HIP_FORCE_DEV_KERNARG=0 ROC_USE_FGS_KERNARG=0 ROC_USE_FGS_KERNARG=1 DEBUG_HIP_KERNARG_COPY_OPT=0 DEBUG_HIP_KERNARG_COPY_OPT=1 DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0 DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1 GPU_STREAMOPS_CP_WAIT=1 ROC_SYSTEM_SCOPE_SIGNAL=0 AMD_OPT_FLUSH=0 ROC_AQL_QUEUE_SIZE=65536"}
CONFIGS=${CONFIGS:-8x1,1x1}
run() { echo "== $1"; env $1 timeout 300 python tools/probes/half_batches.py --configs $CONFIGS 2>&1 | grep "x B"; }
run "GN_BASELINE=1"
for k in $KNOBS; do run "$k"; done
run "GN_BASELINE=2"
