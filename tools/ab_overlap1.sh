# same-box A/B of library variants with ONE step in flight, then two:  bash tools/ab_overlap1.sh name1 name2 ...
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it), or set it to the repo root}"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for o in 1 2; do for round in 1 2; do for n in "$@"; do
MVMC_LIB_PATH=$PWD/multiview_motion_capture_amd/lib/libmvmc_$n.so python3 bench.py --cpu-frames 0 --overlap $o 2>/dev/null > gpurun_out/ab1_$n.json
python3 -c "
import json;r=json.load(open('gpurun_out/ab1_$n.json'));print('$n overlap $o', round(r['value']), round(r['ms_per_step'],2), [round(x,1) for x in r['stages_ms']['chain_mcycles_mean_max']])"
done; done; done
