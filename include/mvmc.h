/*
 * mvmc.h -- C ABI of the MI355X (gfx950) hot path of multi-view motion capture:
 * cross-view association -> multi-view DLT triangulation -> temporal IK.
 *
 * The reference (khanhha/multiview_motion_capture) is pure Python and has no
 * FFI of its own; the boundary it offers is a set of Python callables
 * (SURVEY.md section 8b).  Each entry point below is the batched device form
 * of one of those callables and cites the reference function it replaces
 * (file:line relative to the reference checkout).  The Python mirror in
 * multiview_motion_capture_amd/ binds them with ctypes and keeps the
 * reference's names and signatures.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter name ends in
 *     `_host`; the caller owns all buffers, nothing is allocated or freed and
 *     nothing synchronises inside a call (safe for hipGraph capture);
 *   - `stream` is a hipStream_t (NULL = default stream);
 *   - return value: MVMC_OK or an MVMC_ERR_* code, never an exception;
 *   - tensors are dense row-major; shapes are written (d0,d1,...).
 *   - a "pose" is 17 COCO joints x (x, y, score) in float64; poses of a batch
 *     live in one array `kps17` of shape (F, C, P, 17, 3); pose index
 *     q = (f*C + c)*P + p, so the camera of pose q is (q / P) % C.
 */
#ifndef MVMC_H
#define MVMC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVMC_ABI_VERSION 6

enum {
    MVMC_OK = 0,
    MVMC_ERR_ARG = 1,         /* bad shape / null pointer / unsupported size */
    MVMC_ERR_LAUNCH = 2,      /* hipLaunch / runtime error */
    MVMC_ERR_UNSUPPORTED = 3  /* size outside the compiled kernel variants */
};

enum { MVMC_F32 = 0, MVMC_F64 = 1 };

#define MVMC_N_COCO 17
#define MVMC_N_SKEL 18
#define MVMC_N_SIDE 11
#define MVMC_N_PARAM 68   /* root(3) + euler(18*3) + side bone lengths(11) */
#define MVMC_MAX_NODES 80 /* max graph nodes (2-D poses + tracklets) per frame: C8 P8 + 16 tracklets */

typedef void* mvmcStream_t; /* hipStream_t */

/* Skeleton constants (inverse_kinematics.py:94-117 Skeleton, :120-173 load_skeleton).
 * Host struct, passed by pointer and copied into the kernel arguments. */
typedef struct {
    double bone_dirs[MVMC_N_SKEL][3]; /* ref_bone_dirs */
    int32_t parents[MVMC_N_SKEL];     /* joint_parents (-1 for the root) */
    int32_t side_map[MVMC_N_SKEL];    /* ref_side_to_full_bone_lens_map */
    int32_t n_side;                   /* number of side lengths (11; 18 = identity map of the old pickle schema) */
    double ref_side_lens[MVMC_N_SKEL]; /* ref_side_bone_lens (first n_side entries): cold-start lengths */
} mvmcSkeleton;

int mvmc_abi_version(void);
const char* mvmc_status_string(int status);

/* Host helper: first `count` doubles of numpy.random.RandomState(0).rand()
 * (MT19937, init_genrand(0), 53-bit doubles).  match_als seeds its factor with
 * RandomState(0).rand(n, r) (mv_association.py:271) == table[i*r + j]. */
int mvmc_als_seed_table(double* out_host, int count);

/* IN-1 + IN-2: OpenPose-25 -> COCO-17 gather (pose_def.py:262-270,
 * motion_capture.py:980-983) and filter_bad_pose(0.01, 4, 5)
 * (motion_capture.py:1023-1043) with per-view compaction (order kept).
 *   kps        (F,C,P,J_in,3) f32|f64, J_in = 25 (gathered) or 17 (copied)
 *   counts_in  (F,C) people per view, or NULL = P everywhere
 *   kps17      (F,C,P,17,3) f64 out (slots >= counts_out are zero)
 *   counts_out (F,C) people kept per view */
int mvmc_ingest(const void* kps, int dtype, int n_frames, int n_views, int p_max, int n_joints_in,
                const int32_t* counts_in, double min_score, int min_valid, double min_bb_size,
                double* kps17, int32_t* counts_out, mvmcStream_t stream);

/* AS-1: calc_pairwise_f_mats (mv_math_util.py:267-285).  K (C,3,3), Rt (C,3,4) f64 -> F (C,C,3,3) f32. */
int mvmc_fmats(const double* K, const double* Rt, int n_views, float* F, mvmcStream_t stream);

/* AS-2 + AS-3: geometry_affinity (mv_math_util.py:320-351) with projected_distance (:288-317).
 * Graph nodes of frame f = its poses in (view, person) order; n_f = sum_c counts[f,c].
 *   D, S  (F,N,N) f32 out with N = C*P (only [0:n_f,0:n_f] is meaningful, rest 0); either may be NULL
 * Both are the reference's bit for bit: the float32 statistics in NumPy's pairwise order, and the sigmoid's exp as NumPy's float32 exp
 * (AVX2 / AVX-512F form: Cody-Waite reduction, P5 / Q2, fused steps -- not the correctly-rounded function; np_exp_f32 in csrc/mvmc_common.h). */
int mvmc_affinity(const double* kps17, const int32_t* counts, const float* Fmats, int n_frames, int n_views,
                  int p_max, float* D, float* S, mvmcStream_t stream);

/* AS-4 + AS-5 + AS-6: match_als (mv_association.py:222-318), transform_closure (:99-121) and the
 * cluster rule of parse_match_result (motion_capture.py:417-425).
 *   W            (F,N,N) f32|f64 affinity in compact node order (leading dimension N)
 *   group_counts (F,G) nodes per group (views; for match_spatial_time group 0 = tracklets)
 *   g_max        upper bound of any group's node count (rank <= min(n, 2*g_max) selects the kernel variant)
 *   seed_table   first seed_len doubles of RandomState(0).rand() (mvmc_als_seed_table)
 *   x_bin, match_mat  (F,N,N) u8 out, may be NULL
 *   labels       (F,N) i32 out: cluster ordinal of each node, -1 = in no cluster of >= 2 members
 *   n_clusters   (F) i32 out (kept columns); iters (F) i32 out (ALS iterations run) */
int mvmc_als_associate(const void* W, int w_dtype, const int32_t* group_counts, int n_frames, int n_groups,
                       int n_max, int g_max, const double* seed_table, int seed_len, uint8_t* x_bin, uint8_t* match_mat,
                       int32_t* labels, int32_t* n_clusters, int32_t* iters, mvmcStream_t stream);

/* AS-5 + AS-6 alone: transform_closure (mv_association.py:99-121) and the cluster rule of
 * parse_match_result (motion_capture.py:419-425) on a caller-supplied binary matrix.
 *   x_bin (F,N,N) u8, n_nodes (F) i32 -> match_mat (F,N,N) u8 (may be NULL), labels (F,N), n_clusters (F) */
int mvmc_closure_labels(const uint8_t* x_bin, const int32_t* n_nodes, int n_frames, int n_max, uint8_t* match_mat,
                        int32_t* labels, int32_t* n_clusters, mvmcStream_t stream);

/* match_svt (mv_association.py:321-411): ADMM with singular-value thresholding and the doubly-stochastic block projection
 * (myproj2dpam, :15-60); the matcher match_multiview_poses can select instead of match_als.  pSelect = 1 (the reference's default).
 *   S            (F,N,N) f32|f64 affinity in compact node order (leading dimension N <= 64); not modified (the reference zeroes
 *                the diagonal of its argument; the symmetrised, zero-diagonal copy lives on the device here)
 *   group_counts (F,G) nodes per group, G <= 16; g_max upper bound of any group's node count (<= 16)
 *   alpha, lambda, mu, tol, max_iter, dual_stochastic: the reference's keyword arguments (0.1, 50, 64, 5e-4, 20, 1)
 *   work         (F,3,N,N) f64 device workspace (Y and the two projection states)
 *   x_bin        (F,N,N) u8 out: (X + X^T) / 2 > 0.5;  x_out (F,N,N) f64 out or NULL: X itself
 *   iters        (F) i32 out: SVD-equivalent decompositions run (info['iter'] + 1 when the tolerance was met, else max_iter)
 * Arithmetic is float64 for either input type.  Follow with mvmc_closure_labels for match_mat / cluster labels. */
int mvmc_svt_associate(const void* S, int s_dtype, const int32_t* group_counts, int n_frames, int n_groups, int n_max,
                       int g_max, double alpha, double lambda, double mu, double tol, int max_iter, int dual_stochastic,
                       double* work, uint8_t* x_bin, double* x_out, int32_t* iters, mvmcStream_t stream);

/* Turns labels into member lists: members (F,K,V) pose index q (ascending node order, -1 padded),
 * n_members (F,K).  Clusters beyond K or members beyond V are dropped (n_members still counts them). */
int mvmc_cluster_members(const int32_t* labels, const int32_t* counts, int n_frames, int n_views, int p_max,
                         int k_max, int v_max, int32_t* members, int32_t* n_members, mvmcStream_t stream);

/* TR-1 + TR-2: triangulate_point_groups_from_multiple_views_linear(post_optimize=False)
 * (mv_math_util.py:152-187, :215-240).  One problem = one member list.
 *   kps     (n_poses,J,3) f64 (J = n_joints; 17 on the batched path, 18 with the synthetic mid-spine);
 *   Pmats (C,3,4) f64; members (B,V) pose indices (-1 = unused slot)
 *   out     (B,J,4) f64: x, y, z, mean score of the views used; problems with < 1 member give NaN */
int mvmc_dlt(const double* kps, const double* Pmats, const int32_t* members, int n_problems, int v_max,
             int n_views, int p_max, int n_joints, double min_score, double* out, mvmcStream_t stream);

/* IN-1/IN-2 + TR-1/TR-2 in ONE pass over the raw keypoints (BASELINE config 2, triangulation only): mvmc_ingest followed by mvmc_dlt
 * on the 17 COCO joints, without the 17-joint tensor ever going to memory (a wave per frame keeps it in LDS): 12 C P J bytes in and
 * 17 x 32 bytes per cluster out per frame instead of the two-kernel form's extra 2 x 408 C P bytes.  Same results bit for bit.
 *   kps, dtype, n_joints_in, counts_in, ingest_min_score, min_valid, min_bb_size: as mvmc_ingest
 *   members (F,K,V) i32: the clusters of every frame as pose indices in mvmc_ingest's OUTPUT numbering, (f C + c) P + slot, all of
 *           frame f (-1 = unused); out (F,K,17,4) f64 as mvmc_dlt; counts_out (F,C) i32 or NULL: people per view after the filter */
int mvmc_ingest_dlt(const void* kps, int dtype, int n_frames, int n_views, int p_max, int n_joints_in, const int32_t* counts_in,
                    double ingest_min_score, int min_valid, double min_bb_size, const double* Pmats, const int32_t* members,
                    int k_max, int v_max, double min_score, double* out, int32_t* counts_out, mvmcStream_t stream);

/* The same pass with float32 keypoints in AND float32 points out -- SURVEY 8(d)'s I/O for config 2: 12 C P J bytes read and
 * 16 P J bytes written per frame, each point ONE 16-byte store.  The arithmetic is mvmc_ingest_dlt's (float64; the reference
 * triangulates in float64, mv_math_util.py:152-187,215-240); out (F,K,17,4) f32 = that result rounded once at the store.
 * MVMC_ERR_UNSUPPORTED for shapes the pipelined kernel does not hold (k_max * 17 > 192, n_views * p_max > 128, p_max > 16, ...):
 * use mvmc_ingest_dlt there. */
int mvmc_ingest_dlt_f32(const float* kps, int n_frames, int n_views, int p_max, int n_joints_in, const int32_t* counts_in,
                        double ingest_min_score, int min_valid, double min_bb_size, const double* Pmats, const int32_t* members,
                        int k_max, int v_max, double min_score, float* out, int32_t* counts_out, mvmcStream_t stream);

/* TR-2, post_optimize=True (mv_math_util.py:189-210): scipy least_squares(max_nfev = 2) on the unsigned
 * residual |proj - obs| * score with eps = 1e-6, i.e. one trust-region trial step from the DLT points, kept
 * only if it lowers the cost.  pts (B,J,4) is the output of mvmc_dlt, updated in place (x, y, z only). */
int mvmc_triangulate_postopt(const double* kps, const double* Pmats, const int32_t* members, int n_problems,
                             int v_max, int n_views, int p_max, int n_joints, double* pts, mvmcStream_t stream);

/* FK-1 + FK-2: foward_kinematics (inverse_kinematics.py:176-199) with Quaternions.from_euler /
 * transforms (Quaternions.py:449-462, :335-366).
 *   params (B,3+54+n_side) f64 = root, euler(18,3), bone lengths
 *   joints (B,18,3) out; G (B,18,4,4) out or NULL */
int mvmc_fk(const mvmcSkeleton* skel_host, const double* params, int n_problems, double* joints, double* G,
            mvmcStream_t stream);

/* IK-1..IK-4: PoseSolver.solve (inverse_kinematics.py:351-433) = two trust-region-reflective
 * least-squares stages (solve_pose_reproj :202-238, solve_pose_bone_lens_reproj :241-277; SciPy
 * least_squares defaults, max_nfev evaluations each).
 *   members      (B,V) pose indices into kps17 (-1 = none; v_max = V <= 64, every member is used)
 *   init_params  (B,68) warm-start parameters, ignored where cold[b] != 0
 *   cold         (B) u8: 1 = cold start (root = midpoint of the post-optimised DLT hips, zero angles,
 *                reference lengths, max_nfev_cold),
 *                0 = warm (max_nfev_warm); NULL = all cold
 *   params_out   (B,68); joints_out (B,18,3); info_out (B,8) f64 =
 *                {cost1, nfev1, status1, cost2, nfev2, status2, njev1+njev2, number of trust-region models that
 *                took the eigensolver fallback (no clean split between range and null space)} or NULL
 *   scratch      (B, MVMC_IK_SCRATCH_DOUBLES) f64 device workspace: the Householder vectors of each trust-region model
 *                and the eigenvectors of the (rare) eigensolver fallback; contents undefined before and after */
#define MVMC_IK_SCRATCH_DOUBLES 7680
int mvmc_ik_solve(const mvmcSkeleton* skel_host, const double* kps17, const double* Pmats,
                  const int32_t* members, int n_problems, int v_max, int n_views, int p_max,
                  const double* init_params, const uint8_t* cold, int max_nfev_cold, int max_nfev_warm,
                  double* params_out, double* joints_out, double* info_out, double* scratch, mvmcStream_t stream);

/* ---- temporal layer: match_spatial_time + tracker, batched over independent chains (sub-sequences) ---- */

/* AS-8 helper: get_fundamental_matrix(P_i, P_j) (mv_math_util.py:57-77) for every view pair.
 * Pmats (C,3,4) f64 -> F2 (C,C,3,3) f64 with x_j^T F2[i][j] x_i = 0. */
int mvmc_fmats_from_projections(const double* Pmats, int n_views, double* F2, mvmcStream_t stream);

/* AS-7/8/9: node graph of match_spatial_time (motion_capture.py:651-756) for one frame of every chain.
 * Nodes of chain b = its n_tracks[b] live tracklets (last FK joints), then the 2-D poses of frame
 * frame_idx[b] by view.  2D-2D different views: calc_epipolar_error (mv_math_util.py:80-115, score product
 * > 0.1); 2D-3D: reprojection_error (motion_capture.py:403-414); same view / 3D-3D: NaN -> nanmax + 1;
 * S = 1/(1+exp(5 (D-15)/30)) clamped (S < 1e-3 -> 0).  Chains with n_tracks == 0 get an empty graph (they
 * take the match_spatial path).
 *   track_joints (B,T,18,3) f64; min_score = 0.1 in the reference (motion_capture.py:696,714,725);
 *   W (B,NS,NS) f64 out, NS = T + C*P; D (B,NS,NS) raw distances (NaN kept) out or NULL;
 *   group_counts (B,C+1) i32 out = {n_tracks, people per view} */
int mvmc_st_affinity(const double* kps17, const int32_t* counts, const int32_t* frame_idx,
                     const double* track_joints, const int32_t* n_tracks, const double* Pmats, const double* F2,
                     int n_chains, int n_views, int p_max, int t_max, double min_score, double* W, double* D,
                     int32_t* group_counts, mvmcStream_t stream);

/* TK-1, first half (motion_capture.py:763-808 and :618-626): cluster labels -> IK problems of the frame.
 * Chains without tracklets read labels_sp (B,C*P) (match_spatial, every member kept); the others read
 * labels_st (B,T+C*P) (tracklet-anchored clusters, one pose per view, first wins).
 *   members (B,T+K,V) pose indices; cold (B,T+K) u8; init_params (B,T+K,68);
 *   status (B,T) i32: 0 unmatched (dies), 1 one view (kept, not updated), 2 updated; n_new (B) new tracklets;
 *   overflow (B) i32 in/out or NULL: bit 0 is OR-ed in where a cluster (more than k_max new ones) or a member (more than v_max
 *   views) was dropped for lack of room -- the reference has no such caps; k_max >= C*P / 2 and v_max >= C*P rule both out */
int mvmc_track_assign(const int32_t* labels_sp, const int32_t* ncl_sp, const int32_t* labels_st,
                      const int32_t* ncl_st, const int32_t* counts, const int32_t* frame_idx,
                      const int32_t* n_tracks, const double* track_params, int n_chains, int n_views, int p_max,
                      int t_max, int k_max, int v_max, int32_t* members, uint8_t* cold, double* init_params,
                      int32_t* status, int32_t* n_new, int32_t* overflow, mvmcStream_t stream);

/* TK-1, second half (MvTracklet.update / mark_missed / __init__, motion_capture.py:352-391, :924-963):
 * applies the frame's IK results to the tracklet table (order kept, survivors first, new ones appended).
 *   meta (B,T,4) i32 = {id, state (1 tentative, 2 confirmed), hits, length}; n_inits = 3 in the reference;
 *   slot_src (B,T) i32 out or NULL: IK problem slot each table entry was solved in this frame (-1 = not solved);
 *   overflow (B) i32 in/out or NULL: bit 1 is OR-ed in where a new tracklet did not fit the table of t_max slots */
int mvmc_track_commit(const int32_t* status, const int32_t* n_new, const double* ik_params, const double* ik_joints,
                      int n_chains, int t_max, int k_max, int n_inits, double* track_params, double* track_joints,
                      int32_t* meta, int32_t* n_tracks, int32_t* next_id, int32_t* n_dead, int32_t* slot_src,
                      int32_t* overflow, mvmcStream_t stream);

/* ---- multi-GPU glue.  No counterpart in the reference: its tracker is one sequential pass
 * (motion_capture.py:1062-1116).  A sequence is cut into chains that cold-start; contiguous chain ranges go to the GPUs; one
 * all-gather of the packed messages below; then the identities are stitched across ALL chain boundaries on every rank. ---- */

/* Message of one shard, in 4-byte words (n_chains_cap = chains the message has room for, >= the shard's own):
 *   [0,8)   header i32 {n_chains, chain_len, t_max, rows written, rows wanted, row_cap, n_frames, 0}; rows wanted > row_cap = overflow
 *   ids     (n_chains_cap) i32: number of local identities of each chain (the tracker's next_id)
 *   bounds  (n_chains_cap, 2, t_max, MVMC_BOUND_WORDS): tracklet table of the chain's first / last frame:
 *           {local id i32 or -1, 18 x 3 joints f32 (NaN when empty), pad}
 *   rows    (row_cap, MVMC_ROW_WORDS): one row per LIVE tracklet and frame, in frame order:
 *           {frame, slot, id, state, hits, length} i32, 54 joints f32, 68 parameters f32
 *   local   the shard's OWN stitch, made before the gather so that no rank repeats another's work:
 *           [0,8) i32 {identities that start in this shard, pairs matched at its inner chain boundaries, error word (bit 0: a chain
 *                 with more than id_cap ids, bit 1: an assignment did not terminate), void word (bit i = void_words[i] != 0), 0 ...}
 *           lmatch (n_chains_cap, t_max) i32: slot of the previous chain's last frame matched to slot s of this chain's first (-1;
 *                 the shard's first chain: -1, its boundary belongs to the gathered pass)
 *           lgid   (n_chains_cap, id_cap) i32: shard-local identity of (chain, local id), numbered in chain order; -1 = no such id */
#define MVMC_BOUND_WORDS 56
#define MVMC_ROW_WORDS 128
long long mvmc_pack_message_words(int n_chains_cap, int t_max, int row_cap, int id_cap);   /* -1 on bad arguments */
long long mvmc_pack_work_words(int n_frames, int chain_len, int id_cap);                  /* words of mvmc_pack_tracks' workspace */

/* Packs mvmc_chain_run's per-frame outputs (out_params (F,t_tables,68), out_joints (F,t_tables,18,3), out_meta (F,t_tables,4),
 * out_n_tracks (F)) and next_id (n_chains) into `message` (mvmc_pack_message_words words), then stitches the shard's own chain
 * boundaries (chain b-1 | b: optimal assignment -- Kuhn-Munkres -- of the last frame's tracklets to the first frame's on the mean
 * joint distance, pairs farther than max_dist metres dropped) and numbers its identities locally (the `local` section).
 *   t_tables  slots per frame of the tables; t_max <= 16 slots of the message: the SAME on every rank (a rank whose repair tier
 *             widened its tables still packs t_max slots; every frame's live tracklets must fit: more are cut and show as an
 *             overflow of rows wanted in no header -- callers pass t_max >= the widest table any rank can have)
 *   void_words (n_void_words <= 32) u32 device words or NULL: non-zero words (mvmc_chain_run's flags[B .. B + 3): hand-over time-out,
 *             graph too large, capacity exceeded) are recorded in the message, and every rank's stitch reports them (info[2] bit 2):
 *             a void step is seen by all ranks without any of them reading device memory before the collective
 *   work      mvmc_pack_work_words i32 device workspace */
int mvmc_pack_tracks(const double* out_params, const double* out_joints, const int32_t* out_meta, const int32_t* out_n_tracks,
                     const int32_t* next_id, int n_frames, int chain_len, int t_tables, int t_max, int n_chains_cap, int row_cap,
                     int id_cap, double max_dist, const uint32_t* void_words, int n_void_words, int32_t* work, void* message,
                     mvmcStream_t stream);

/* Stitches the chains of `world` gathered messages (rank order = sequence order; message r at messages + r * message_words words).
 * The inner boundaries of every shard arrive stitched (the `local` sections); here only the world - 1 boundaries between shards are
 * matched (same rule), local identities become global ones (roots numbered in chain order), and the tables are written -- the result
 * of one pass over all chain boundaries, at a cost per rank that does not grow with the number of chains of the OTHER ranks beyond
 * writing their rows of the tables.
 *   gid    (n_chains_total_cap, id_cap) i32 out: global identity of (chain, local id), -1 where the chain has fewer identities
 *   match  (n_chains_total_cap, t_max) i32 out: slot of the previous chain's last frame matched to slot s of this chain's first, -1
 *   info   (4) i32 out: {chains, global identities, error word, pairs}; error word bit 0 = a message overflowed or a chain has more
 *          than id_cap ids, bit 1 = an assignment did not terminate (cannot happen with finite costs; a tracklet with a non-finite
 *          joint is given a cost beyond any max_dist, i.e. it matches nobody), bit 2 = some shard's void word is set (its compute
 *          step was void).  Non-zero = the result is void
 *   work   mvmc_stitch_work_words i32 device workspace;  n_chains_total_cap >= world * n_chains_cap; id_cap as packed */
long long mvmc_stitch_work_words(int world, int t_max, int id_cap);
int mvmc_stitch_chains(const void* messages, long long message_words, int world, int n_chains_cap, int t_max, int row_cap,
                       int id_cap, double max_dist, int n_chains_total_cap, int32_t* gid, int32_t* match, int32_t* info,
                       int32_t* work, mvmcStream_t stream);

/* ---- diagnostics (used by the tests; not part of the hot path's call surface) ---- */

/* The IK kernel's symmetric eigensolver on caller-supplied matrices: A (B,n,n) symmetric PSD, g (B,n),
 * 3 <= n <= 50.  lam (B,n) ascending with the numerically-null cluster (lam <= 1e-13 lam_max) set to 0;
 * Vt (B,n,n) rows = eigenvectors (rows 0..k0-1 of the null cluster are zero); k0 (B) = size of the null
 * cluster; phase_cycles (B,5) or NULL = shader cycles of {tridiagonalisation, multisection, twisted
 * factorisation, re-orthogonalisation, back-transformation}.  g is unused (kept for ABI stability). */
int mvmc_debug_eigh(const double* A, const double* g, int n_problems, int n, double* lam, double* Vt, int32_t* k0,
                    double* phase_cycles, mvmcStream_t stream);

/* One trust-region step of the IK solver computed in the Krylov tridiagonal basis (no eigendecomposition), on a
 * caller-supplied least-squares model: J = B (n_problems,m,n row-major), residual image r (n_problems,m), so
 * g = B^T r and M = B^T B.  3 <= m,n <= 50.  step (n_problems,n); out4 (n_problems,4) = {alpha, predicted
 * reduction, |step| incl. absorber, size of the leading (range) block, or -1 where the IK kernel would fall back
 * to the eigensolver (then alpha = -1, step = 0)}.  phase_cycles (n_problems,4) or NULL = shader cycles of the
 * tridiagonalisation's {reflector + publish, matrix-vector + exchange, rank-2 update} phases and their total. */
int mvmc_debug_trstep(const double* B, const double* r, int n_problems, int m, int n, double Delta, double alpha0,
                      double* step, double* out4, double* phase_cycles, mvmcStream_t stream);

/* ONE trust-region model and ONE trial step of the production IK solver from a caller-given iterate -- the unit in which the
 * reference's recorded iterates are compared (tests/golden/ik_trf_traces.npz: x_k, Delta_k, alpha_k of SciPy's trf_no_bounds,
 * trf.py:466-500, as called from inverse_kinematics.py:236,274).  The device functions are the solver's own.
 *   params (B,68): the point x_k (stage 0 optimises the first 57 and keeps params[57:] as lengths; stage 1 all 68)
 *   Delta, alpha0 (B): trust radius and the Levenberg-Marquardt parameter handed to the sub-problem (solve_lsq_trust_region's
 *                initial_alpha)
 *   out (B, MVMC_IK_STEP_OUT_DOUBLES) f64:
 *     [0] cost at x   [1] |g|_inf   [2] alpha   [3] predicted reduction   [4] |step| as the solver accounts for it (= Delta whenever
 *     the sub-problem is rank deficient: the share the reference spends on numerically-null directions is modelled, not taken)
 *     [5] cost at the trial point   [6] path: 0 stopped by the gradient test, 1 Krylov step, 2 eigenbasis fallback; + 4 where the
 *     Euler-space model was used instead of the reduced coordinates; -1 = fewer than two views   [7] rows of the leading block
 *     [8, 76) g = J^T f by parameter (0 where the column is structurally zero)   [80, 148) step   [160, 228) trial point
 *   scratch as mvmc_ik_solve. */
#define MVMC_IK_STEP_OUT_DOUBLES 240
int mvmc_debug_ik_model_step(const mvmcSkeleton* skel_host, const double* kps17, const double* Pmats, const int32_t* members,
                             int n_problems, int v_max, int n_views, int p_max, const double* params, int stage,
                             const double* Delta, const double* alpha0, double* out, double* scratch, mvmcStream_t stream);

/* The stages of PoseSolver.solve one at a time, and their 3-D-target variants:
 *   stage_mask 1 = solve_pose_reproj (x = root, euler; inverse_kinematics.py:202-238),
 *              2 = solve_pose_bone_lens_reproj (x = root, euler, side lengths; :241-277), 3 = both in sequence;
 *   targets3d  NULL: reprojection residual on kps17 / members as in mvmc_ik_solve;
 *              (B,18,4) f64 = x, y, z, weight per observation row (COCO-17 + mid-spine): solve_pose (:280-307) and
 *              solve_pose_bone_lens (:310-336), residual (joint - target) * weight; kps17, Pmats, members are unused.
 * Every problem starts from init_params (B,68) with max_nfev evaluations per stage (least_squares(max_nfev=n_max_iter)).
 * Outputs and scratch as in mvmc_ik_solve; info of a stage that was not run is 0. */
int mvmc_ik_solve_stages(const mvmcSkeleton* skel_host, const double* kps17, const double* Pmats,
                         const int32_t* members, const double* targets3d, int n_problems, int v_max, int n_views,
                         int p_max, const double* init_params, int stage_mask, int max_nfev,
                         double* params_out, double* joints_out, double* info_out, double* scratch,
                         mvmcStream_t stream);

/* TRF-faithful IK (diagnostic, never on the hot path): PoseSolver.solve (inverse_kinematics.py:380-433) with the reference's own
 * numerical method -- scipy.optimize.least_squares defaults as called at inverse_kinematics.py:236,274: 2-point finite-difference
 * Jacobians (step sqrt(eps) sign(x) max(1,|x|)), an SVD of J and solve_lsq_trust_region -- instead of the production solver's analytic
 * Jacobian and Krylov step; the cold start triangulates with post_optimize=True through the same least_squares restatement
 * (mv_math_util.py:189-210).  Used to measure the band in which independent implementations of the reference's algorithm land.
 * Arguments as mvmc_ik_solve, plus stage_mask (1 = solve_pose_reproj only, 2 = solve_pose_bone_lens_reproj only, 3 = both);
 *   work  (B, MVMC_IK_FD_WORK_DOUBLES) f64 device workspace.  v_max <= 8. */
#define MVMC_IK_FD_WORK_DOUBLES 40960
int mvmc_debug_ik_solve_fd(const mvmcSkeleton* skel_host, const double* kps17, const double* Pmats, const int32_t* members,
                           int n_problems, int v_max, int n_views, int p_max, const double* init_params, const uint8_t* cold,
                           int max_nfev_cold, int max_nfev_warm, int stage_mask, double* params_out, double* joints_out,
                           double* info_out, double* work, mvmcStream_t stream);

/* The whole temporal hot path of a shard in ONE launch: a persistent 256-thread workgroup per chain runs
 * graph (match_spatial where the chain has no tracklet, match_spatial_time otherwise) -> ALS -> assignment -> IK ->
 * commit for the chain_len frames of its chain -- MvTracker.update_4d (motion_capture.py:873-963) per chain, the
 * same device code as mvmc_affinity / mvmc_st_affinity / mvmc_als_associate / mvmc_track_assign / mvmc_ik_solve /
 * mvmc_track_commit issued frame by frame, and the same results.  Chain b owns frames [b chain_len, (b+1) chain_len).
 * All pointers are device memory owned by the caller (N = n_views p_max, NS = t_max + N, NP = t_max + k_max,
 * B = n_chains, F = B chain_len).  Two LDS layouts (MVMC_ERR_UNSUPPORTED outside them: use the per-stage entry points), both with
 * n_views <= 16, p_max <= 8, NP <= 64, v_max <= 64:
 *   small  N <= 40, NS <= 48, max(t_max, p_max) <= 8 (its association variants hold rank 16) (configs 1-4; 128 VGPRs, four
 *          workgroups per CU; launches of at most two workgroups per CU run the 256-VGPR build of the same kernel); every frame's
 *          actual graph must have <= 24 nodes without tracklets and <= 32 with them, and the frame <= 24 poses in clusters (the IK
 *          phase's view pool) -- checked on the device;
 *   big    N <= 64, NS <= 80, t_max <= 16 (config 5, C8 P8; one 512-thread workgroup per CU; also taken with force_big): graphs of
 *          up to 72 nodes and rank 16 on the fast association variant, up to 80 nodes and rank 32 (sixteen live tracklets) on the
 *          generic one inside the same workgroup: every graph of those sizes fits.
 * Capacities the reference does not have (motion_capture.py:417-446, :763-808 accept any cluster size and any number of clusters and
 * tracklets): with k_max >= N / 2 (a new tracklet needs two poses) and v_max >= min(N, 64) (clusters are disjoint sets of the
 * frame's poses; the view blocks of a frame's IK problems share one pool, so a cluster may be as large as the frame) neither a
 * cluster nor a member is ever dropped; what remains is t_max (tracklet slots, tied to the rank the association variants hold)
 * and the graph sizes above.  A chain that exceeds one of them is reported in its void word, flags[B + 4 + b], and its results are
 * void; tracker.repair_chains re-runs such chains through the per-stage entry points with wider tables. */
typedef struct mvmcChainBuffers {
    int32_t n_chains, chain_len, n_views, p_max, t_max, k_max, v_max, max_nfev_cold, max_nfev_warm, n_inits, seed_len;
    int32_t n_parts;            /* workgroups per chain: 1 = one persistent workgroup per chain; p > 1 (dividing chain_len) =
                                   p workgroups running consecutive frame ranges of the chain one after the other, so that
                                   the hardware dispatcher balances the load over the CUs */
    int32_t force_big;          /* != 0: the big layout even where the small one would do (tests) */
    int32_t hand_over;          /* how a chain's workgroups follow one another (n_parts > 1).  0 = by block index (part * n_chains +
                                   chain; a part waits for its chain's flag: relies on workgroups being dispatched in block order,
                                   bounded wait, loud time-out); 1 = ready queue (a workgroup draws a ticket when it starts and takes
                                   the chain that has been ready longest: no assumption about dispatch order); 2 = the mapping of 0
                                   indexed by a ticket drawn when the workgroup starts instead of by the block index (a part's
                                   predecessor holds a lower ticket, so it has started: no assumption about dispatch order, the speed
                                   of 0; what the Python layer uses).  Same results bit for bit */
    /* inputs */
    const double* kps17;        /* (F,C,P,17,3) after mvmc_ingest */
    const int32_t* counts;      /* (F,C) */
    const double* Pmats;        /* (C,3,4) */
    const float* Fmats;         /* (C,C,3,3) from mvmc_fmats */
    const double* F2;           /* (C,C,3,3) from mvmc_fmats_from_projections */
    const double* seed_table;   /* mvmc_als_seed_table, on the device */
    /* tracker state, in/out (zero-initialised for fresh chains) */
    double* params;             /* (B,T,68) */
    double* joints;             /* (B,T,18,3) */
    int32_t* meta;              /* (B,T,4) id, state, hits, length */
    int32_t* n_tracks;          /* (B) */
    int32_t* next_id;           /* (B) */
    int32_t* n_dead;            /* (B) */
    int32_t* slot_src;          /* (B,T) */
    /* workspaces, contents undefined before and after */
    float* S_sp;                /* (B,N,N) */
    double* W_st;               /* (B,NS,NS) */
    int32_t* group_counts;      /* (B,C+1) */
    int32_t* labels_sp;         /* (B,N) */
    int32_t* labels_st;         /* (B,NS) */
    int32_t* n_clusters_sp;     /* (B) */
    int32_t* n_clusters_st;     /* (B) */
    int32_t* iters_sp;          /* (B) */
    int32_t* iters_st;          /* (B) */
    int32_t* members;           /* (B,NP,V); row s holds n_members[s] entries, the rest of the row is undefined */
    int32_t* n_members;         /* (B,NP) */
    uint8_t* cold;              /* (B,NP) */
    double* init;               /* (B,NP,68) */
    int32_t* status;            /* (B,T) */
    int32_t* n_new;             /* (B) */
    double* ik_params;          /* (B,NP,68) */
    double* ik_joints;          /* (B,NP,18,3) */
    double* ik_info;            /* (B,NP,8) */
    double* ik_scratch;         /* (B,8,MVMC_IK_SCRATCH_DOUBLES) */
    /* per-frame outputs: the tracklet table after every frame */
    double* out_params;         /* (F,T,68) */
    double* out_joints;         /* (F,T,18,3) */
    int32_t* out_meta;          /* (F,T,4) */
    int32_t* out_n_tracks;      /* (F) */
    double* out_info;           /* (F,NP,8) IK info rows of the frame's problems, or NULL */
    int32_t* out_als_iters;     /* (F) ALS iterations of the frame's graph, or NULL */
    uint32_t* flags;            /* (B (n_parts + 1) + 8) u32, zeroed by the call: [0,B) hand-over flags of the chains; afterwards
                                   flags[B] != 0 = a workgroup timed out waiting for its predecessor, flags[B + 1] != 0 = a graph was
                                   too large for the kernel's ALS variant, flags[B + 2] != 0 = a capacity was exceeded in some chain
                                   (bit 0: a cluster, a member or a view block dropped, bit 1: more than t_max tracklets),
                                   flags[B + 4 + b] = the void word of chain b (bits 0, 1 as before, bit 2 = graph too large, bit 3 =
                                   internal: a meeting of two IK waves timed out, builds with -DMVMC_WITH_IK_PAIR only): a
                                   non-zero word voids the chain's results (all chains' after a time-out); from 2 B + 4 on: the
                                   ticket counter (hand_over 1, 2), the ready queue's tail and ring (hand_over == 1) */
    double* out_phase_cycles;   /* (B,8) diagnostic: shader cycles of each chain by phase {graph, ALS, assignment, IK, commit,
                                   outputs, whole chain, 0}, or NULL */
} mvmcChainBuffers;
int mvmc_chain_run(const mvmcSkeleton* skel_host, const mvmcChainBuffers* buffers, mvmcStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MVMC_H */
