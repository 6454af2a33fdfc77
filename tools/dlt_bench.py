"""Config 2 of BASELINE.json (triangulation only, C5 P1 J25): the ingest + DLT kernels at 10 k and 2 M frames.
HBM roofline of the pair: 1,500 B read (12 C P J, f32 keypoints) + 400 B written (16 P J) per frame; the kernels
here keep the reference's f64 (the ingested 17-joint tensor is f64), so the bytes actually moved are also printed."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiview_motion_capture_amd import device as dev, synth  # noqa: E402

C, P = 5, 1
base = synth.generate(10000, C, P, 20260101)
Pm = torch.from_numpy(base["P"]).cuda()
for F in (10000, 2000000):
    reps = F // 10000
    kps = torch.from_numpy(base["kps25"]).cuda().repeat(reps, 1, 1, 1, 1).contiguous()      # (F,5,1,25,3) f32
    cnt = torch.from_numpy(base["counts"]).cuda().repeat(reps, 1).contiguous()
    mem = (torch.arange(F, device="cuda", dtype=torch.int32)[:, None] * C + torch.arange(C, device="cuda", dtype=torch.int32)[None])
    mem = mem.contiguous()                                                              # one cluster per frame: the 5 views
    for rep in range(3):
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        k17, c17 = dev.ingest(kps, cnt)
        e[1].record()
        pts = dev.dlt(k17, Pm, mem)
        e[2].record()
        torch.cuda.synchronize()
    m3 = mem.view(F, 1, C)
    for rep in range(3):
        f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        f0.record()
        fused = dev.ingest_dlt(kps, cnt, Pm, m3)
        f1.record()
        torch.cuda.synchronize()
    assert torch.equal(fused.view(-1), pts.view(-1))
    t_f = f0.elapsed_time(f1)
    print("F=%8d  one pass (mvmc_ingest_dlt) %.3f ms -> %.1f M frames/s, algorithmic %.1f GB/s (%.1f %% of 8 TB/s)" %
          (F, t_f, F / t_f / 1e3, F * (12 * C * P * 25 + 16 * P * 25) / t_f / 1e6, F * (12 * C * P * 25 + 16 * P * 25) / t_f / 1e6 / 80))
    t_in, t_dlt = e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])
    alg = F * (12 * C * P * 25 + 16 * P * 25)
    moved_dlt = F * (C * 17 * 3 * 8 + 17 * 4 * 8 + C * 4)
    print("F=%8d  ingest %.3f ms  DLT %.3f ms  -> %.1f M frames/s (ingest + DLT), algorithmic %.1f GB/s (%.1f %% of 8 TB/s); "
          "DLT kernel alone moves %.1f GB/s of f64" % (F, t_in, t_dlt, F / (t_in + t_dlt) / 1e3, alg / (t_in + t_dlt) / 1e6,
                                                     alg / (t_in + t_dlt) / 1e6 / 80, moved_dlt / t_dlt / 1e6))
    del kps, k17, pts
