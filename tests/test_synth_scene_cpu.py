"""The benchmark's synthetic scene: one smooth walk that any number of ranks cut into contiguous ranges."""
import hashlib

import numpy as np

from multiview_motion_capture_amd import synth


def test_segments_of_one_seed_tile_one_scene():
    F, P = 320, 4
    segs = [synth.generate(F, 5, P, 20260103, walk="scene", segment=s) for s in range(3)]
    for a in segs[1:]:
        assert np.array_equal(a["K"], segs[0]["K"]) and np.array_equal(a["Rt"], segs[0]["Rt"])      # one calibration
    step = max(np.abs(np.diff(s["gt_joints"], axis=0)).max() for s in segs)                          # the largest move inside a segment
    for a, b in zip(segs, segs[1:]):
        jump = np.abs(b["gt_joints"][0] - a["gt_joints"][-1]).max()
        assert jump <= step, (jump, step)                                                           # the boundary is an ordinary frame step
        bone = lambda s, f: np.linalg.norm(s["gt_joints"][f, :, 1] - s["gt_joints"][f, :, 0], axis=-1)
        assert np.allclose(bone(a, 0), bone(b, 7))                                                  # the same people (bone lengths)
    # a segment is a function of (seed, segment, length) only: whoever generates it gets the same frames
    again = synth.generate(F, 5, P, 20260103, walk="scene", segment=2)
    assert np.array_equal(again["kps25"], segs[2]["kps25"]) and np.array_equal(again["gt_order"], segs[2]["gt_order"])
    # different noise per segment (not a repeated shard)
    assert not np.array_equal(segs[0]["kps25"][:8], segs[1]["kps25"][:8])


def test_the_scene_stays_in_front_of_the_cameras_over_a_long_sequence():
    root, ang = synth.scene_walk(25008, 8, 20260104, segment=7)          # frames 175,056 .. 200,064 of BASELINE config 5's sequence
    assert np.abs(root[..., :2] - root[..., :2].mean(axis=0)).max() < 1.5                            # metres around home
    assert 0.25 < ang.std() < 0.35                                                                   # the chain heads' pose distribution (0.3 rad)
    d = synth.generate(64, 8, 8, 20260104, walk="scene", segment=7)
    assert (d["counts"] == 8).all() and np.isfinite(d["kps25"]).all()


def test_the_default_generator_is_unchanged():
    """Fixtures recorded from the reference on synthetic inputs (tests/golden) depend on generate()'s default stream."""
    d = synth.generate(64, 5, 4, 20260103, chain_len=16)
    assert hashlib.sha256(d["kps25"].tobytes()).hexdigest()[:16] == "c1cd9cbcd422420c"
