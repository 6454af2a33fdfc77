"""Host restatement of the multi-GPU glue (pack / stitch; include/mvmc.h: mvmc_pack_tracks, mvmc_stitch_chains).

TEST INFRASTRUCTURE -- NOT PRODUCT CODE (see oracle/oracle_np.py header).  The reference has no counterpart (its tracker is one
sequential pass, motion_capture.py:1062-1116); this file is the independent NumPy / SciPy statement of the same message format and
the same stitching rule (Hungarian assignment on the mean joint distance per chain boundary, identities propagated along matches),
used (a) to check the device kernels and (b) by the world-size-2 gloo test on the CPU, which has no GPU to run them.
"""
from __future__ import annotations

import numpy as np
from scipy.optimize import linear_sum_assignment

BOUND_WORDS, ROW_WORDS, HDR, LOCAL_HDR = 56, 128, 8, 8


def message_words(b_cap, t_max, row_cap, id_cap=16):
    return HDR + b_cap + b_cap * 2 * t_max * BOUND_WORDS + row_cap * ROW_WORDS + LOCAL_HDR + b_cap * (t_max + id_cap)


def pack_np(params, joints, meta, n_tracks, next_id, chain_len, b_cap, row_cap, t_msg=None, void_words=None, max_dist=0.5, id_cap=16):
    """-> (words,) int32.  The tables have params.shape[1] slots per frame, the message t_msg (default: the same).  The ``local``
    section holds the shard's own stitch: one sequential pass over ITS chain boundaries (the rule of stitch_np below)."""
    F, TT = params.shape[:2]
    T = TT if t_msg is None else int(t_msg)
    L = chain_len
    B = F // L
    msg = np.zeros(message_words(b_cap, T, row_cap, id_cap), dtype=np.int32)
    nt = np.clip(n_tracks, 0, min(T, TT))
    total = int(nt.sum())
    msg[:8] = [B, L, T, min(total, row_cap), total, row_cap, F, 0]
    msg[HDR:HDR + B] = next_id[:B]
    o_b = HDR + b_cap
    o_r = o_b + b_cap * 2 * T * BOUND_WORDS
    o_l = o_r + row_cap * ROW_WORDS
    bounds = msg[o_b:o_r].reshape(b_cap, 2, T, BOUND_WORDS)
    jf = joints.reshape(F, TT, 54).astype(np.float32)
    for b in range(B):
        for side, f in ((0, b * L), (1, b * L + L - 1)):
            for s in range(T):
                live = s < nt[f]
                bounds[b, side, s, 0] = meta[f, s, 0] if live else -1
                bounds[b, side, s, 1:55] = (jf[f, s] if live else np.full(54, np.nan, np.float32)).view(np.int32)
    rows = msg[o_r:o_l].reshape(row_cap, ROW_WORDS)
    r = 0
    pf = params.astype(np.float32)
    for f in range(F):
        for s in range(nt[f]):
            if r >= row_cap:
                break
            rows[r, 0], rows[r, 1] = f, s
            rows[r, 2:6] = meta[f, s]
            rows[r, 6:60] = jf[f, s].view(np.int32)
            rows[r, 60:128] = pf[f, s].view(np.int32)
            r += 1
    # ---- the shard's own stitch ----
    lh = msg[o_l:o_l + LOCAL_HDR]
    lmatch = msg[o_l + LOCAL_HDR:o_l + LOCAL_HDR + b_cap * T].reshape(b_cap, T)
    lgid = msg[o_l + LOCAL_HDR + b_cap * T:].reshape(b_cap, id_cap)
    lmatch[:B] = -1
    lgid[:B] = -1
    n_roots = pairs = err = 0
    for b in range(B):
        n_ids = int(next_id[b])
        if n_ids > id_cap:
            err |= 1
        n_ids = min(n_ids, id_cap)
        if b > 0:
            pids, ids0 = bounds[b - 1, 1, :, 0], bounds[b, 0, :, 0]
            pj = bounds[b - 1, 1, :, 1:55].copy().view(np.float32).reshape(T, 18, 3)
            j0 = bounds[b, 0, :, 1:55].copy().view(np.float32).reshape(T, 18, 3)
            ip, inn = np.nonzero(pids >= 0)[0], np.nonzero(ids0 >= 0)[0]
            for i, j in match_boundary(pj[ip], j0[inn], max_dist):
                lmatch[b, inn[j]] = ip[i]
                pairs += 1
                if ids0[inn[j]] < id_cap and pids[ip[i]] < id_cap:
                    lgid[b, ids0[inn[j]]] = lgid[b - 1, pids[ip[i]]]
        for l in range(n_ids):
            if lgid[b, l] < 0:
                lgid[b, l] = n_roots
                n_roots += 1
    vw = 0
    if void_words is not None:
        for i, w in enumerate(np.asarray(void_words).ravel()):
            vw |= (1 << i) if w else 0
    lh[:4] = [n_roots, pairs, err, vw]
    return msg


def match_boundary(joints_prev, joints_next, max_dist):
    """joints_* (n,18,3) float32 of the live tracklets -> list of (i_prev, i_next)."""
    if len(joints_prev) == 0 or len(joints_next) == 0:
        return []
    a = joints_prev.astype(np.float64)[:, None]
    b = joints_next.astype(np.float64)[None]
    cost = np.sqrt(((a - b) ** 2).sum(axis=-1)).sum(axis=-1) / 18.0
    cost = np.where(np.isfinite(cost), cost, 1e30)      # a tracklet with a non-finite joint matches nobody (SciPy would raise)
    r, c = linear_sum_assignment(cost)
    return [(int(i), int(j)) for i, j in zip(r, c) if cost[i, j] <= max_dist]


def stitch_np(messages, b_cap, t_max, row_cap, max_dist=0.5, id_cap=16):
    """messages (world, words) int32 -> dict(gid (world*b_cap, id_cap), match (world*b_cap, T), info (4)).
    ONE sequential pass over all chain boundaries of all shards, from the bounds sections alone -- the definition the device's two-level
    scheme (shard-local stitch in the message + the boundaries between shards) has to reproduce.  Of the messages' local sections only
    the error and void words are read (they are part of info[2])."""
    world = messages.shape[0]
    T = t_max
    o_b = HDR + b_cap
    o_r = o_b + b_cap * 2 * T * BOUND_WORDS
    o_l = o_r + row_cap * ROW_WORDS
    chains = []   # (n_ids, first table ids, first joints, last ids, last joints)
    flag = 0
    for r in range(world):
        m = messages[r].view(np.int32) if isinstance(messages[r], np.ndarray) else np.asarray(messages[r]).view(np.int32)
        h = m[:8]
        if h[4] > h[5] or h[0] > b_cap or h[2] != T:
            flag |= 1
        if h[0] > 0 and m.size >= o_l + LOCAL_HDR:
            flag |= int(m[o_l + 2]) | (4 if m[o_l + 3] else 0)
        bounds = m[o_b:o_r].reshape(b_cap, 2, T, BOUND_WORDS)
        for b in range(int(h[0])):
            ids = bounds[b, :, :, 0]
            jo = bounds[b, :, :, 1:55].copy().view(np.float32).reshape(2, T, 18, 3)
            chains.append((int(m[HDR + b]), ids[0].copy(), jo[0], ids[1].copy(), jo[1]))
    Btot = len(chains)
    cap = world * b_cap
    gid = -np.ones((cap, id_cap), dtype=np.int32)
    match = -np.ones((cap, T), dtype=np.int32)
    next_gid = pairs = 0
    for g, (n_ids, ids0, j0, ids1, j1) in enumerate(chains):
        if n_ids > id_cap:
            flag |= 1
        n_ids = min(n_ids, id_cap)
        if g > 0:
            _, _, _, pids, pj = chains[g - 1]
            ip = np.nonzero(pids >= 0)[0]
            inn = np.nonzero(ids0 >= 0)[0]
            for i, j in match_boundary(pj[ip], j0[inn], max_dist):
                match[g, inn[j]] = ip[i]
                pairs += 1
                if ids0[inn[j]] < id_cap and pids[ip[i]] < id_cap:
                    gid[g, ids0[inn[j]]] = gid[g - 1, pids[ip[i]]]
        for l in range(n_ids):
            if gid[g, l] < 0:
                gid[g, l] = next_gid
                next_gid += 1
    return dict(gid=gid, match=match, info=np.array([Btot, next_gid, flag, pairs], dtype=np.int32))
