# Config 5 geometry (C8 P8) with occlusions and false detections over several seeds: no void word, nothing repaired (the ninth / tenth
# tracklets stay in the launch), throughput and accuracy per seed.   bash tools/seed_soak_c5.sh   (inside one GPU call)
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it), or set it to the repo root}"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for spec in "11 0.05 0.2" "12 0.05 0.2" "13 0.1 0.3" "14 0.02 0.5" "15 0.1 0.0" "20260104 0.05 0.2"; do
set -- $spec
python3 bench.py --no-other-configs --cpu-frames 0 --sustain 0 --views 8 --people 8 --frames 8192 --steps 6 --warmup 1 --seed $1 --occlusion $2 --spurious $3 2>gpurun_out/soak_c5_$1.err > gpurun_out/soak_c5_$1.json || { echo "seed $1 FAILED"; tail -2 gpurun_out/soak_c5_$1.err; continue; }
python3 -c "
import json;r=json.loads(open('gpurun_out/soak_c5_$1.json').read().strip().splitlines()[-1]);e=r['tracker_events_per_step'];print('seed $1 occlusion $2 spurious $3:', round(r['value']), 'frames/s; births', e['births'], 'deaths', e['deaths'], 'repaired', e['chains_repaired_per_step'], 'word', e['capacity_word'], '; all people tracked in', round(r['accuracy']['frames_with_all_people_tracked'],3), 'of the frames, median joint error', round(r['accuracy']['joint_error_vs_ground_truth_cm']['median'],2), 'cm')"
done
