# the experiment switches are read by the experiments build only (tuning.h): make -C cunvsm_amd/csrc dbg
export CUNVSM_AMD_LIB=${CUNVSM_AMD_LIB:-$(cd "$(dirname "$0")/.." && pwd)/cunvsm_amd/libcunvsm_amd_dbg.so}
cd "$(dirname "$0")/.."
run() { r=$(env $1 python bench.py --steps 200 --warmup 20 --repeats 3 --batch 6400 --no-cpu-baseline --no-extra-legs --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"); echo "[$1] $r"; }
for round in 1 2; do
for v in "NVSM_DT_SLABS=128" "NVSM_X=0" "NVSM_ROWS_TPW=1" "NVSM_ROWS_TPW=1 NVSM_LAZY_MIN_MB=0" "NVSM_ROWS_TPW=1 NVSM_LAZY_MIN_MB=0 NVSM_ENTRY_WALK_MIN_DOCS=0" "NVSM_ROWS_TPW=1 NVSM_ENTRY_WALK_MIN_DOCS=0" "NVSM_ROWS_TPW=1 NVSM_LAZY_MIN_MB=0 NVSM_DT_SLABS=12" "NVSM_ROWS_TPW=1 NVSM_LAZY_MIN_MB=0 NVSM_DT_SLABS=50"; do run "$v"; done; done
