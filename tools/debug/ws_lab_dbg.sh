#!/bin/bash
cd $GRAFT_REPO_ROOT
L=tools/ubench/bin/pw_lab
O=gpurun_out/ws_dbg3.txt
: > $O
for d in 0 2 7 8 3 5; do echo "--- BNHIP_WS_DBG=$d" >> $O; BNHIP_WS_DBG=$d timeout 60 $L b13/expand 12 6 >> $O 2>&1; done
