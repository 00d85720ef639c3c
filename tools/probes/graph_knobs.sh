#!/bin/bash
# hipGraph replay against stream replay of the recorded calls, same box: one / two streams inside the program, and the runtime's graph switches
CONFIGS=${CONFIGS:-1x1,8x1}
run() { echo "== $1 $2"; env $1 timeout 300 python tools/probes/half_batches.py --configs $CONFIGS $2 2>&1 | grep "x B"; }
run "GN_X=0" ""
run "GN_X=0" "--graph"
run "GN_TWO_STREAMS=0" ""
run "GN_TWO_STREAMS=0" "--graph"
for k in DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_HIP_FORCE_GRAPH_QUEUES=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=2 DEBUG_HIP_FORCE_GRAPH_QUEUES=8 DEBUG_HIP_GRAPH_BATCH_SIZE=1 DEBUG_HIP_GRAPH_BATCH_SIZE=1024 DEBUG_HIP_DYNAMIC_QUEUES=0; do run "$k" "--graph"; done
run "GN_X=1" ""
