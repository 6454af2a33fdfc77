set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/parts
for cfg in "16 2" "1 2" "1 3" "2 2" "2 3" "16 3" "1 4"; do
  set -- $cfg
  python3 bench.py --cpu-frames 0 --parts $1 --overlap $2 > gpurun_out/parts/p$1_o$2.json 2> gpurun_out/parts/p$1_o$2.err
  python3 -c "
import json,sys
r=json.load(open('gpurun_out/parts/p$1_o$2.json'))
print('parts $1 overlap $2', round(r['value']), round(r['ms_per_step'],2), [round(x,1) for x in r['stages_ms']['chain_mcycles_mean_max']])"
done
