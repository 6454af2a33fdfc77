: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it), or set it to the repo root}"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for seed in 1 2 3 4 5; do
python3 bench.py --no-other-configs --cpu-frames 0 --sustain 0 --steps 30 --warmup 2 --seed $seed --occlusion 0.02 --spurious 0.05 2>/dev/null > gpurun_out/seed_$seed.json || { echo "seed $seed FAILED"; exit 1; }
python3 -c "
import json;r=json.load(open('gpurun_out/seed_$seed.json'));print('seed $seed', round(r['value']), r['tracker_events_per_step'], r['accuracy']['frames_with_all_people_tracked'], round(r['accuracy']['joint_error_vs_ground_truth_cm']['median'],3))"
done
