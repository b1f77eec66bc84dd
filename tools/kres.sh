#!/bin/bash
# compact per-kernel register / LDS / occupancy table of one .hip file: tools/kres.sh cunvsm_amd/csrc/update.hip [grep pattern] [extra hipcc flags]
f=$1; pat=${2:-.}; shift 2 || true
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage "$@" -c $f -o /tmp/kres_$$.o 2>&1 |
  grep -E "Function Name|VGPRs:|AGPRs|SGPRs:|Occupancy|LDS Size|ScratchSize" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' |
  awk '/Function Name/{if(n)print n" |"s; n=$3; s=""; next}{gsub(/^ +/,""); s=s" "$0";"}END{print n" |"s}' | c++filt | grep -E "$pat"
rm -f /tmp/kres_$$.o
