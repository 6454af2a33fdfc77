for n in base pair; do
  for fr in 64 1024; do
  MVMC_LIB_PATH=$PWD/multiview_motion_capture_amd/lib/libmvmc_$n.so timeout -k 10 200 python bench.py --cpu-frames 0 --no-other-configs --sustain 0 --frames $fr --overlap 1 --steps 10 2>/dev/null > gpurun_out/ab_$n.json || exit 1
  python - "$n" $fr <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_%s.json" % sys.argv[1]))
s = d["stages_ms"]; sh = s.get("chain_cycle_shares", {}); mc = s.get("chain_mcycles_mean_max", [0, 0])
print("%-8s F=%s %8.0f frames/s  %.3f ms  ALS %.2f Mcyc  IK %.2f Mcyc  chain mean %.2f max %.2f" % (
    sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], sh.get("als", 0) * mc[0], sh.get("ik", 0) * mc[0], mc[0], mc[1]), flush=True)
PY
  done
done
