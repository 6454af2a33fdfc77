# throughput against frames per launch, workgroups per chain and steps in flight (one GPU call):
#   bash tools/parts_sweep.sh [frames:parts:overlap ...]
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it), or set it to the repo root}"
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/parts
[ $# -gt 0 ] || set -- 10000:16:2 10000:1:2 10000:2:2 10000:16:3 10000:1:3
for cfg in "$@"; do
  IFS=: read F P O <<< "$cfg"
  python3 bench.py --cpu-frames 0 --frames $F --parts $P --overlap $O > gpurun_out/parts/f${F}_p${P}_o$O.json 2> gpurun_out/parts/f${F}_p${P}_o$O.err
  python3 -c "
import json
r=json.load(open('gpurun_out/parts/f${F}_p${P}_o$O.json'))
print('frames $F parts $P overlap $O', round(r['value']), round(r['ms_per_step'],2), [round(x,1) for x in r['stages_ms']['chain_mcycles_mean_max']])"
done
