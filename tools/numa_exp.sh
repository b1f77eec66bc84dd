#!/bin/bash
# does it matter WHEN the submitting thread is bound? (a) the whole process on the GPU's node from the start (taskset), (b) the
# API call after the HIP runtime has initialised (the default), (c) unbound.   tools/numa_exp.sh ["<bench flags>"]
cd "$(dirname "$0")/.."
FLAGS=${1:---config=lse_small}
NODE=$(python -c "
import cunvsm_amd as ca
print(ca.bind_host_thread(0))" 2>/dev/null | tail -1)
CPUS=$(cat /sys/devices/system/node/node$NODE/cpulist)
echo "GPU on NUMA node $NODE, cpus $CPUS"
echo "== (a) taskset from the start"; NVSM_BIND_HOST=0 taskset -c $CPUS tools/lse_modes.sh 4 "$FLAGS"
echo "== (b) nvsm_bind_host_thread"; tools/lse_modes.sh 4 "$FLAGS"
echo "== (c) unbound"; NVSM_BIND_HOST=0 tools/lse_modes.sh 4 "$FLAGS"
