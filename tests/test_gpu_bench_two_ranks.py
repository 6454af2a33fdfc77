"""GPU: the line the first `--gpus N` run of bench.py prints.  Two ranks share the one GPU of the test box (host-staged gloo
collective: the rehearsal of the N > 1 call path); the headline command at N > 1 must also time BASELINE config 5 -- the configuration
BASELINE defines by its scaling curve (200 k frames, C8 P8 over 8 GPUs = 25,008 frames per GPU; the sequential pass it shards:
motion_capture.py:1062-1116) -- and every line must say what its collective was."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _check_collective(c, world):
    assert c["backend"] == "gloo" and c["world_size"] == world and c["messages_gathered"] == world
    assert c["all_gathers_per_step"] == 1 and c["message_bytes"] > 0
    assert c["gather_ms"]["p50"] > 0 and c["gather_ms"]["max"] >= c["gather_ms"]["p50"]


def test_two_ranks_print_config_4_and_config_5_with_their_collectives():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--backend", "gloo", "--steps", "2",
                        "--warmup", "1", "--sustain", "0", "--cpu-frames", "0"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["views"] == 5 and line["config"]["people"] == 4 and line["config"]["frames_per_gpu"] == 10000
    assert len(line["per_rank_ms_per_step"]) == 2
    _check_collective(line["collective"], 2)
    assert line["collective"]["chains_stitched"] == 2 * 625
    # ONE scene, cut into two contiguous ranges (round 6: --walk continuous, the same seed on both ranks): the four people are carried
    # across every chain boundary inside the shards AND across the boundary between the shards, and remain four identities
    c = line["collective"]
    assert line["config"]["walk"] == "continuous" and line["scaling"] == "weak" and c["shard_boundaries"] == 1
    assert c["identities_carried_across_shards"] > 0
    assert c["identities_carried_within_shard"] >= 0.95 * 2 * 624 * 4
    assert c["global_identities"] < 0.05 * (2 * 625 * 4)          # (a scene per chain, as in rounds 1 - 5, would give 5,000)
    gt = line["accuracy"]["stitch_vs_ground_truth_rank0"]
    assert gt["chain_boundaries_checked"] == 624 and gt["identities_carried"] >= 0.95 * gt["people_on_both_sides"] > 0
    assert gt["carried_to_the_right_person"] >= 0.99 * gt["identities_carried"]
    h = line["als_iterations"]
    assert h["graphs"] == 10000 and 1 <= h["min"] <= h["p50"] <= h["p90"] <= h["max"] <= 1000
    assert sum(h["histogram"].values()) == h["graphs"]
    # the boundary's host buffers inside the bracket (pinned host memory -> device before a step, its tables back after it)
    hio = line["host_io"]
    assert hio["h2d_bytes_per_step"] == 10000 * 5 * 4 * 25 * 3 * 4 + 10000 * 5 * 4 and hio["d2h_bytes_per_step"] > 0 and 0 < hio["value"]
    oc = line["other_configs"]
    assert len(oc) == 1
    c5 = oc[0]
    assert c5["n_gpus"] == 2 and c5["config"]["views"] == 8 and c5["config"]["people"] == 8 and c5["config"]["frames_per_gpu"] == 25008
    assert c5["config"]["seed"] == 20260104 and c5["steps"] >= 3
    _check_collective(c5["collective"], 2)
    assert c5["collective"]["chains_stitched"] == 2 * 1563
    assert c5["collective"]["identities_carried_across_shards"] > 0 and c5["config"]["walk"] == "continuous"
    assert c5["als_iterations"]["graphs"] == 25008
    assert c5["value"] > 0 and c5["tracker_events_per_step"]["capacity_word"] == 0
    print(f"two ranks on one GPU (gloo): config 4 {line['value'] / 1e3:.0f} k frames/s, config 5 {c5['value'] / 1e3:.0f} k frames/s; gather p50 "
          f"{line['collective']['gather_ms']['p50']:.2f} / {c5['collective']['gather_ms']['p50']:.2f} ms, messages of "
          f"{line['collective']['message_bytes'] / 1e6:.1f} / {c5['collective']['message_bytes'] / 1e6:.1f} MB")


def test_strong_scaling_form_splits_one_total_over_the_ranks():
    """`--frames-total` (BASELINE config 5 is defined by a strong-scaling curve: 200 k frames over 1 / 2 / 4 / 8 GPUs): the total is cut
    into contiguous ranges of one scene, the line says `scaling: strong`, and `value` counts the total once."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--backend", "gloo", "--views", "8",
                        "--people", "8", "--frames-total", "4096", "--seed", "20260104", "--steps", "2", "--warmup", "1", "--sustain", "0",
                        "--cpu-frames", "0", "--host-io", "0", "--no-other-configs"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["scaling"] == "strong" and line["n_gpus"] == 2
    assert line["config"]["frames_per_gpu"] == 2048 and line["config"]["frames_total"] == 4096
    assert abs(line["value"] - 4096 * line["steps"] / (line["ms_per_step"] * line["steps"] / 1e3)) < 1e-6 * line["value"]
    assert line["collective"]["chains_stitched"] == 256 and line["collective"]["identities_carried_across_shards"] > 0
