# the experiment switches are read by the experiments build only (tuning.h): make -C cunvsm_amd/csrc dbg
export CUNVSM_AMD_LIB=${CUNVSM_AMD_LIB:-$(cd "$(dirname "$0")/.." && pwd)/cunvsm_amd/libcunvsm_amd_dbg.so}
cd "$(dirname "$0")/.."
for round in 1 2; do for v in ${EPWS:-0 17 18 20 25}; do
  r=$(NVSM_LOSS_EPW=$v python bench.py --steps 60 --repeats 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])")
  echo "[epw=$v] $r"
done; done
