: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it), or set it to the repo root}"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=multiview_motion_capture_amd/lib
python3 tools/lib_diff.py $L/libmvmc_base.so $L/libmvmc_hip.so 2048 8 8 | tail -5
for round in 1 2; do for n in "$@"; do
MVMC_LIB_PATH=$PWD/$L/libmvmc_$n.so python3 bench.py --cpu-frames 0 --views 8 --people 8 --frames 8192 --steps 3 --warmup 1 2>/dev/null > gpurun_out/ab5_$n.json
python3 -c "
import json;r=json.load(open('gpurun_out/ab5_$n.json'));s=r['stages_ms'];sh=s['chain_cycle_shares'];mc=s['chain_mcycles_mean_max'];print('$n config5', round(r['value']), round(r['ms_per_step'],2), 'ALS %.1f IK %.1f mean %.1f max %.1f'%(sh['als']*mc[0], sh['ik']*mc[0], mc[0], mc[1]))"
done; done
