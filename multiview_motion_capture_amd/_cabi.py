"""ctypes binding of the C ABI in include/mvmc.h (libmvmc_hip.so, gfx950).

The product path has no CPU fallback: if the HIP library is missing or a call
fails, this module raises.  Build the library with ``python -c "import
__graft_entry__ as g; g.build()"`` or ``make -C multiview_motion_capture_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

_LIB_PATH = os.environ.get("MVMC_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib",
                                                           "libmvmc_hip.so")
_lib = None

MVMC_OK = 0
MVMC_ABI = 6    # MVMC_ABI_VERSION of include/mvmc.h that the argument types in load() describe
MVMC_F32, MVMC_F64 = 0, 1
N_PARAM = 68
MAX_NODES = 80
IK_SCRATCH_DOUBLES = 7680
IK_FD_WORK_DOUBLES = 40960
IK_STEP_OUT_DOUBLES = 240
BOUND_WORDS, ROW_WORDS = 56, 128

# every symbol declared in include/mvmc.h
SYMBOLS = (
    "mvmc_abi_version", "mvmc_status_string", "mvmc_als_seed_table", "mvmc_ingest", "mvmc_fmats",
    "mvmc_affinity", "mvmc_als_associate", "mvmc_closure_labels", "mvmc_cluster_members", "mvmc_dlt", "mvmc_triangulate_postopt", "mvmc_fk", "mvmc_ik_solve",
    "mvmc_fmats_from_projections", "mvmc_st_affinity", "mvmc_track_assign", "mvmc_track_commit", "mvmc_debug_eigh",
    "mvmc_debug_trstep", "mvmc_ik_solve_stages", "mvmc_chain_run", "mvmc_svt_associate", "mvmc_debug_ik_solve_fd", "mvmc_debug_ik_model_step", "mvmc_ingest_dlt", "mvmc_ingest_dlt_f32", "mvmc_pack_message_words", "mvmc_pack_work_words", "mvmc_stitch_work_words", "mvmc_pack_tracks", "mvmc_stitch_chains",
)


class MvmcSkeleton(C.Structure):
    """mvmcSkeleton of include/mvmc.h."""
    _fields_ = [("bone_dirs", (C.c_double * 3) * 18), ("parents", C.c_int32 * 18),
                ("side_map", C.c_int32 * 18), ("n_side", C.c_int32), ("ref_side_lens", C.c_double * 18)]


class MvmcChainBuffers(C.Structure):
    """mvmcChainBuffers of include/mvmc.h (field order matters)."""
    _INTS = ("n_chains", "chain_len", "n_views", "p_max", "t_max", "k_max", "v_max", "max_nfev_cold", "max_nfev_warm",
             "n_inits", "seed_len", "n_parts", "force_big", "hand_over")
    _PTRS = ("kps17", "counts", "Pmats", "Fmats", "F2", "seed_table", "params", "joints", "meta", "n_tracks", "next_id",
             "n_dead", "slot_src", "S_sp", "W_st", "group_counts", "labels_sp", "labels_st", "n_clusters_sp", "n_clusters_st",
             "iters_sp", "iters_st", "members", "n_members", "cold", "init", "status", "n_new", "ik_params", "ik_joints", "ik_info",
             "ik_scratch", "out_params", "out_joints", "out_meta", "out_n_tracks", "out_info", "out_als_iters", "flags", "out_phase_cycles")
    _fields_ = [(n, C.c_int32) for n in _INTS] + [(n, C.c_void_p) for n in _PTRS]


class MvmcError(RuntimeError):
    pass


def lib_path() -> str:
    return _LIB_PATH


def build_info():
    """lib/BUILD_INFO.json of the shipped library (kernel-source hash, compiler, host, time), or None for an MVMC_LIB_PATH library."""
    if os.environ.get("MVMC_LIB_PATH"):
        return None
    from . import _buildinfo
    return _buildinfo.read()


def load():
    """Load libmvmc_hip.so (once) and declare the argument types."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise MvmcError(
            f"HIP extension not built: {_LIB_PATH} is missing. There is no CPU fallback; "
            "run __graft_entry__.build() (or make -C multiview_motion_capture_amd/csrc).")
    if not os.environ.get("MVMC_LIB_PATH"):
        # the shipped library must be a build of THIS tree's kernel sources (lib/BUILD_INFO.json, written by csrc/Makefile): a stale
        # .so -- built before an edit, or carried to the GPU box from another checkout -- would be measured and tested in place of the
        # code under review.  (A library named through MVMC_LIB_PATH is a deliberate A/B partner and is not checked.)
        from . import _buildinfo
        # (a deployment without the kernel sources beside the library has nothing to compare with: the check is for source trees)
        info, want = _buildinfo.read(), (_buildinfo.sources_sha() if os.path.isdir(_buildinfo.CSRC) else None)
        if want is not None and (info is None or info.get("kernel_sources_sha") != want):
            have = "no BUILD_INFO.json beside it" if info is None else f"built from kernel sources {info.get('kernel_sources_sha')}"
            raise MvmcError(f"{_LIB_PATH} is stale: {have}, the tree's are {want}.  Rebuild it: python -c \"import __graft_entry__ as g; "
                            "g.build()\" (or make -C multiview_motion_capture_amd/csrc)")
    lib = C.CDLL(_LIB_PATH)
    # The argument types declared below are those of ABI version MVMC_ABI: a library of another version (an old build loaded through
    # MVMC_LIB_PATH for an A/B run, a stale .so) would receive shifted pointers and ints -- device faults or silent garbage.  Refuse it.
    try:
        lib.mvmc_abi_version.restype = C.c_int
        have = int(lib.mvmc_abi_version())
    except AttributeError:
        have = -1
    if have != MVMC_ABI:
        raise MvmcError(f"{_LIB_PATH} has ABI version {have}, this package binds version {MVMC_ABI} (include/mvmc.h): rebuild it "
                        "(make -C multiview_motion_capture_amd/csrc) or load a build of the same ABI version")
    vp, i32, f64 = C.c_void_p, C.c_int, C.c_double
    SK = C.POINTER(MvmcSkeleton)
    argtypes = {
        "mvmc_status_string": [C.c_int],
        "mvmc_als_seed_table": [vp, i32],
        "mvmc_ingest": [vp, i32, i32, i32, i32, i32, vp, f64, i32, f64, vp, vp, vp],
        "mvmc_fmats": [vp, vp, i32, vp, vp],
        "mvmc_affinity": [vp, vp, vp, i32, i32, i32, vp, vp, vp],
        "mvmc_als_associate": [vp, i32, vp, i32, i32, i32, i32, vp, i32, vp, vp, vp, vp, vp, vp],
        "mvmc_closure_labels": [vp, vp, i32, i32, vp, vp, vp, vp],
        "mvmc_cluster_members": [vp, vp, i32, i32, i32, i32, i32, vp, vp, vp],
        "mvmc_dlt": [vp, vp, vp, i32, i32, i32, i32, i32, f64, vp, vp],
        "mvmc_ingest_dlt": [vp, i32, i32, i32, i32, i32, vp, f64, i32, f64, vp, vp, i32, i32, f64, vp, vp, vp],
        "mvmc_ingest_dlt_f32": [vp, i32, i32, i32, i32, vp, f64, i32, f64, vp, vp, i32, i32, f64, vp, vp, vp],
        "mvmc_triangulate_postopt": [vp, vp, vp, i32, i32, i32, i32, i32, vp, vp],
        "mvmc_fk": [SK, vp, i32, vp, vp, vp],
        "mvmc_ik_solve": [SK, vp, vp, vp, i32, i32, i32, i32, vp, vp, i32, i32, vp, vp, vp, vp, vp],
        "mvmc_fmats_from_projections": [vp, i32, vp, vp],
        "mvmc_st_affinity": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f64, vp, vp, vp, vp],
        "mvmc_track_assign": [vp] * 8 + [i32] * 6 + [vp] * 7,
        "mvmc_track_commit": [vp] * 4 + [i32] * 4 + [vp] * 9,
        "mvmc_debug_eigh": [vp, vp, i32, i32, vp, vp, vp, vp, vp],
        "mvmc_debug_trstep": [vp, vp, i32, i32, i32, f64, f64, vp, vp, vp, vp],
        "mvmc_chain_run": [SK, C.POINTER(MvmcChainBuffers), vp],
        "mvmc_ik_solve_stages": [SK, vp, vp, vp, vp, i32, i32, i32, i32, vp, i32, i32, vp, vp, vp, vp, vp],
        "mvmc_debug_ik_solve_fd": [SK, vp, vp, vp, i32, i32, i32, i32, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp],
        "mvmc_debug_ik_model_step": [SK, vp, vp, vp, i32, i32, i32, i32, vp, i32, vp, vp, vp, vp, vp],
        "mvmc_pack_message_words": [i32, i32, i32, i32],
        "mvmc_pack_work_words": [i32, i32, i32],
        "mvmc_stitch_work_words": [i32, i32, i32],
        "mvmc_pack_tracks": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, f64, vp, i32, vp, vp, vp],
        "mvmc_stitch_chains": [vp, C.c_longlong, i32, i32, i32, i32, i32, f64, i32, vp, vp, vp, vp, vp],
    }
    restypes = {"mvmc_status_string": C.c_char_p, "mvmc_pack_message_words": C.c_longlong, "mvmc_pack_work_words": C.c_longlong,
                "mvmc_stitch_work_words": C.c_longlong}
    # A build of another revision loaded through MVMC_LIB_PATH for a same-box A/B comparison (tools/lib_diff.py, tools/*_ab.sh) may
    # lack entry points that were added since WITHIN the same ABI version (checked above); only then is a missing symbol skipped -- the
    # shipped library must export every one.
    ab = bool(os.environ.get("MVMC_LIB_PATH"))
    for name in SYMBOLS:
        if ab and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        if name in argtypes:
            fn.argtypes = argtypes[name]
        fn.restype = restypes.get(name, C.c_int)
    _lib = lib
    return lib


def check(status: int, what: str):
    """Status -> exception (the reference raises Python exceptions, motion_capture.py:112)."""
    if status == MVMC_OK:
        return
    msg = load().mvmc_status_string(status).decode()
    if status == 1:
        raise ValueError(f"{what}: {msg}")
    raise MvmcError(f"{what}: {msg} (status {status})")
