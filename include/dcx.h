/*
 * dcx.h — C ABI of libdcx.so, the MI355X (gfx950) implementation of DiffCo's
 * score(+gradient) hot path.
 *
 * The reference (ucsdarclab/diffco) has no FFI: its boundary for this path is a set of
 * Python call signatures.  Each entry point below names the reference code it replaces
 * (paths under /root/reference).  Signatures use only plain pointers and sizes — no torch
 * types — so any host (ctypes, cgo, JNI, a C++ planner) can bind them; see INTEGRATION.md.
 *
 * Conventions
 *   - All arrays are dense, row-major, fp32.  "dev" = device pointer on the model's GPU,
 *     "host|dev" = either (copied with hipMemcpyDefault at create time).
 *   - The caller owns every q/score/grad/jac buffer.  The library owns only the model's
 *     device copy of the support rows and FK parameters (freed by dcx_model_destroy).
 *   - Every launch is enqueued on the caller's HIP stream (`stream` = hipStream_t, NULL =
 *     default stream) and does NOT synchronise.
 *   - Return value: 0 = DCX_OK, otherwise an error code; text via dcx_last_error()
 *     (thread-local).  Nothing throws or exits across the ABI.
 *   - A model handle's numerical state is immutable after creation: concurrent calls on distinct
 *     streams are legal (small-batch launches keep one internal scratch buffer per stream, allocated on
 *     the stream's first small-batch call and freed only by dcx_model_destroy).  A HIP graph captured
 *     from these calls holds that buffer: keep the model alive while the graph exists, and replay the
 *     graph on the stream it was captured on (or at least never concurrently with other small-batch
 *     calls of the same model on the capture stream — they share the buffer's arrival counters).
 */
#ifndef DCX_H
#define DCX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DCX_VERSION 109 /* 0.1.8: the optimisers' collision term for several classes under per-class margins (dcx_score_hinge_grad_mc, dcx_traj_adam_run_mc, dcx_traj_adam_step_mc), dcx_dh_frames / dcx_euler_frames (utils.DH2mat / euler2mat); 0.1.7: dcx_escape_adam (the escape loop of scripts/escape.py on the stream); 0.1.6: dcx_debug_clock_probe; the matrix-core forms are a build option (dcx_debug_set("mfma" / "xm", 1) -> DCX_ERR_UNSUPPORTED without them); owner-polls words tagged per launch, give-up reported by the model's next launch; 0.1.5: dcx_model_create_ex / dcx_model_update (rows packed on the device); 0.1.4: dcx_train_perceptron_ex (per-call flags); 0.1.3: dcx_fk_desc carries the DCX_FK_TREE section; 32 control points, D <= 96 */

/* ---- status codes ---------------------------------------------------------------- */
#define DCX_OK 0
#define DCX_ERR_INVALID 1     /* bad argument (NULL, negative size, inconsistent shapes) */
#define DCX_ERR_UNSUPPORTED 2 /* shape/kind outside the compiled set (e.g. D > DCX_MAX_D) */
#define DCX_ERR_HIP 3         /* a HIP runtime call failed (message has hipGetErrorString) */
#define DCX_ERR_NO_DEVICE 4   /* no gfx950 device visible */

/* ---- pairwise kernel functions (diffco/kernel.py) -------------------------------- */
/* K(x, s) as a function of d2 = ||x - s||^2, r = sqrt(d2); kparams = {p0, p1}.        */
#define DCX_K_RQ 0   /* RQKernel       kernel.py:12-29   (1 + p0/p1 * d2)^(-p1); p0=gamma, p1=p */
#define DCX_K_POLY 1 /* Polyharmonic   kernel.py:59-79   p0=k, p1=eps: k odd r^k/eps; k even r^k log r/eps (0 at r=0) */
#define DCX_K_MQ 2   /* MultiQuadratic kernel.py:45-57   sqrt(d2/p0^2 + 1); p0=eps */

/* ---- forward-kinematics transforms (diffco/model.py == diffco/robot_fkine.py) ---- */
#define DCX_FK_NONE 0   /* transform=None: features are the configuration itself (D = dof)      */
#define DCX_FK_PLANAR 1 /* RevolutePlanarRobot.fkine      model.py:40-48                        */
#define DCX_FK_DH 2     /* DH chains: Baxter L/R model.py:225-241,283-299; BaxterDual 366-383;
                           PandaFK 430-453 (robot_fkine.py:428-444); DualPandaFK 486-502;
                           link transform = utils.DH2mat utils.py:66-75                        */
#define DCX_FK_SE2 3    /* RigidPlanarBody.fkine          model.py:90-93 (utils.rot_2d)         */
#define DCX_FK_SE3 4    /* RigidBody.fkine                model.py:156-159 (utils.euler2mat = Rz Ry Rx) */
#define DCX_FK_TREE 5   /* URDF kinematic tree, flattened into root-to-leaf chains: link-origin features of
                           RobotDiffCo.tensorized_fkine_single_robot collision_checkers.py:385-393 over
                           URDFRobot.compute_forward_kinematics_all_links urdf_interface.py:516-553 and
                           RigidBody.forward_kinematics collision_interfaces/rigid_body.py:82-140      */

/* joint motions of DCX_FK_TREE (rigid_body.py:100-126); the joint variable is v = t_scale * q + t_offset */
#define DCX_J_FIXED 0     /* no motion (fixed joints that were not merged into their successor)        */
#define DCX_J_REV_X 1     /* x_rot(v)  spatial_vector_algebra.py x_rot                                  */
#define DCX_J_REV_Y 2     /* y_rot(v)                                                                    */
#define DCX_J_REV_Z 3     /* z_rot(v)                                                                    */
#define DCX_J_PRISMATIC 4 /* translation t_axis * v along the joint's own (post-fixed-transform) axes   */

#define DCX_MAX_JOINTS 16 /* per chain */
#define DCX_MAX_CHAINS 2
#define DCX_MAX_POINTS 32
#define DCX_MAX_DOF 32
#define DCX_MAX_D 96  /* feature width n_points * point_dim the fused kernels are compiled for */
#define DCX_MAX_C 8   /* weight columns (classes) the fused kernels are compiled for           */
#define DCX_MAX_TREE_CHAINS 16 /* root-to-leaf paths of a DCX_FK_TREE                           */
#define DCX_MAX_TREE_BASES 4   /* distinct base transforms among them (robots side by side)     */
#define DCX_MAX_TREE_JOINTS 64 /* joints summed over all paths (shared prefixes count per path) */

/*
 * Plain-data description of one `transform(q[dof]) -> control points [n_points, point_dim]`.
 * Output feature k*point_dim + j is coordinate j of control point k, i.e. the row-major
 * flattening the reference kernels apply (kernel.py:20-21, 77).
 */
typedef struct dcx_fk_desc {
    int32_t kind;      /* DCX_FK_*                                                          */
    int32_t dof;       /* width of a configuration row                                      */
    int32_t n_points;  /* m (for DCX_FK_NONE: dof)                                          */
    int32_t point_dim; /* d = 2 or 3 (for DCX_FK_NONE: 1)                                   */

    /* DCX_FK_PLANAR: phi_i = sum_{j<=i} q_j; p_i = sum_{j<=i} l_j (cos phi_j, sin phi_j)   */
    float link_length[DCX_MAX_DOF];

    /* DCX_FK_DH: n_chains serial chains.  Joint i of chain c reads q[joint_q[c][i]], adds
     * theta0, and applies T <- T * DH(theta, a, d, sin_alpha, cos_alpha) starting from
     * base[c] (row-major 3x4 [R|t]).  Control point k is base-frame position of
     * pt_off[k] expressed in frame pt_frame[k] (0-based joint index, i.e. after joint
     * pt_frame[k]'s transform) of chain pt_chain[k].                                      */
    int32_t n_chains;
    int32_t chain_len[DCX_MAX_CHAINS];
    int32_t joint_q[DCX_MAX_CHAINS][DCX_MAX_JOINTS];
    float a[DCX_MAX_CHAINS][DCX_MAX_JOINTS];
    float d[DCX_MAX_CHAINS][DCX_MAX_JOINTS];
    float sin_alpha[DCX_MAX_CHAINS][DCX_MAX_JOINTS];
    float cos_alpha[DCX_MAX_CHAINS][DCX_MAX_JOINTS];
    float theta0[DCX_MAX_CHAINS][DCX_MAX_JOINTS];
    float base[DCX_MAX_CHAINS][12];
    int32_t pt_chain[DCX_MAX_POINTS];
    int32_t pt_frame[DCX_MAX_POINTS];
    float pt_off[DCX_MAX_POINTS][3];

    /* DCX_FK_SE2: q = (x, y, theta); p_k = R(theta) keypoints[k][0:2] + (x, y)
     * DCX_FK_SE3: q = (x, y, z, roll, pitch, yaw); p_k = Rz(yaw) Ry(pitch) Rx(roll) keypoints[k] + (x,y,z) */
    float keypoints[DCX_MAX_POINTS][3];

    /* DCX_FK_TREE: t_n_chains serial chains stored back to back in the t_* joint arrays (chain c owns the
     * next t_chain_len[c] entries).  A chain starts from t_base[c] (row-major 3x4) and every joint applies
     *     T <- T * [t_fixed | as 3x4] * Motion(t_type, t_scale * q[t_q] + t_offset)
     * i.e. the joint's <origin> followed by its motion, as rigid_body.py:100-126 composes them.  A URDF
     * tree becomes one chain per leaf; joints on a shared prefix are simply repeated (the library merges
     * identical prefixes back into one node when it compiles the description, so they are computed once).  Mimic joints use the driver's t_q with the mimic multiplier/offset
     * (rigid_body.py:93-94); an axis of -1 is a negative t_scale (rigid_body.py:103-108).
     * Control point k is pt_off[k] in the frame after joint pt_frame[k] (index within its chain) of chain
     * pt_chain[k].  t_coord_major != 0 lays the features out as [3][n_points] (feature j*n_points + k), the
     * torch.stack(..., dim=-1) layout of collision_checkers.py:390; 0 keeps [n_points][3].               */
    int32_t t_n_chains;
    int32_t t_coord_major;
    int32_t t_chain_len[DCX_MAX_TREE_CHAINS];
    float t_base[DCX_MAX_TREE_CHAINS][12];
    int32_t t_type[DCX_MAX_TREE_JOINTS];
    int32_t t_q[DCX_MAX_TREE_JOINTS];
    float t_scale[DCX_MAX_TREE_JOINTS];
    float t_offset[DCX_MAX_TREE_JOINTS];
    float t_fixed[DCX_MAX_TREE_JOINTS][12];
    float t_axis[DCX_MAX_TREE_JOINTS][3];
} dcx_fk_desc;

typedef struct dcx_model dcx_model; /* opaque; changed only by dcx_model_update */

/* ---- library ----------------------------------------------------------------------- */
int dcx_version(void);
const char* dcx_last_error(void);
int dcx_device_count(void);

/* Developer knobs for tests and A/B tools: override one of the launch-geometry rules (they change how the work
 * is split over blocks, never what is computed).  name: "ys" (support super-chunks per tile), "nw" (waves per
 * block), "min_rows" (supports per wave slice), "split_finish_kernel" (1 = finish split launches with a second
 * launch), "inlaunch_tiles", "jac_per_class" (1 = one launch per class in dcx_score_jac), "mfma" (0 = never use the
 * MFMA contraction, 1 = use it wherever it is compiled), "xf" (0 = the sweep in its direct form everywhere, 1 / rule =
 * the expanded form wherever it is compiled and the model qualifies: Polyharmonic(1), or RQKernel(p = 2) behind an FK
 * transform with gamma * max |s - centroid|^2 <= 32; rows of <= 37 floats; the two forms agree to ~1e-6; 2 = also for RQ
 * models outside that rule: measurements only),
 * "traj_fused" (0 = dcx_traj_adam_run as two launches per iteration), "jac_one_sweep" (0 = dcx_score_jac never takes the
 * one-sweep kernel, 1 = whenever it is compiled and the batch is beyond the one-launch-per-all-classes regime),
 * "train_grid" (see dcx_train_perceptron), "fkk" (which FK walk a DH arm takes: 2 / rule = the step table, 1 = the FK
 * program through scalar loads, 0 = the FK program from its LDS copy; bit-identical results), "jt_waves" (0 = the chain and
 * J^T phases of a DH arm on one wave instead of several; bit-identical), "traj_ys" (workgroups per path of the persistent
 * trajectory kernel: 1 = one, k = k; rule = what fills the chip, at most 8), "traj_across" (1 = a path's workgroups dealt
 * across XCDs instead of onto one: a measurement), "owner_poll" (the split launch's hand-over: 0 = arrival counters, last
 * block to arrive finishes; 1 / rule = block y = 0 owns its tile and polls its peers' (value, tag) words - bit-identical;
 * the rule takes it when every block of the launch is resident at once), "hess_ys" (blocks per tile of dcx_score_hess; 1 =
 * never split the supports), "hess_form" (dcx_score_hess: 0 = one lane per (configuration, direction) sweeps the supports, 1 = the
 * moments form - one lane per configuration sweeps gradient, coefficient sum and the symmetric D x D matrix, the direction lanes
 * read M dx - wherever it is compiled (D <= 16); rule = from B = 1024; same Hessian to fp32 round-off), "xm" (1 = the expanded form takes its distance GEMM from the matrix cores, bf16x3 split
 * operands, where compiled: one class, Polyharmonic(1), even D <= 16; agrees with the VALU form to ~1e-6, measured slower),
 * "solve_threads" (dcx_solve's workgroup size: 256 or 512; rule = 256 up to 736 unknowns; same pivots, same arithmetic),
 * "qt" (small batches of a one-class D = 12 / 24 model as tiles of 16 configurations that sweep all the rows from an LDS copy
 * instead of the split launch: 0 = never, 1 = wherever it is compiled and fits with >= 4 waves; rule = at most 16
 * configurations per CU and room for all 16 waves, unless another knob asks for a particular form of the launch).
 * "skew" (how a 16-wave block's rows are dealt to its four wave groups: 0 = equal slices, w0 | w1 << 10 | w2 << 20 = the per-mille
 * shares of groups 0 - 2, rule = 480 / 320 / 150 / 50: a SIMD issues oldest-first and a block's last waves would finish alone; the tile of 16 configurations
 * deals its row slices the same way), "skew8" (the same for 8-wave blocks: the per-mille share of waves 0 - 3, rule = 600, 0 = equal).
 * value < 0 restores the rule.  The initial values come from the DCX_YS / DCX_NW / DCX_XF / ... environment variables,
 * read once at library load; no launch calls getenv.   */
int dcx_debug_set(const char* name, int64_t value);
/* Measurement aid (0.1.6): one wave on `stream` samples the shader clock counter (s_memtime) and the fixed-rate wall clock
 * (s_memrealtime; its rate in kHz -> *wall_clock_khz, may be NULL) when it starts and again after `wall_ticks` of the latter:
 * out4 [4] uint64 on the device = {clock0, clock1, wall0, wall1}.  (clock1 - clock0) / (wall1 - wall0) x rate = the clock the
 * shaders had meanwhile: launched beside a loop of sweeps on another stream it tells what clock the sweep ran at (under its
 * fp32-VALU load the part holds ~1.45 GHz, not its nominal 2.4: profiles/r05_clock_under_load.txt; bench.py `roofline.clock`). */
int dcx_debug_clock_probe(int device, uint64_t* out4, uint64_t wall_ticks, int32_t* wall_clock_khz, void* stream);

/* ---- model = inference state of a kernel perceptron ------------------------------- */
/* Replaces the state DiffCo.score/poly_score read: support_transformed[S,m,d] + gains[S]
 * or rbf_nodes[S] (kernel_perceptrons.py:41-53, 143-196, 282-283); old API support_fkine[S,D]
 * + rbf_nodes[S,C] (deprecated/MultiDiffCo.py:125-154).
 *   fk            NULL or kind DCX_FK_NONE => D must equal the configuration width.
 *   support_feat  [S, D]  transformed supports (already through the FK), host|dev
 *   weights       [S, C]  gains / rbf_nodes, host|dev; rows whose C weights are all zero
 *                         are dropped (they contribute nothing; this is what
 *                         max_num_supports padding produces, kernel_perceptrons.py:159-196)
 */
int dcx_model_create(dcx_model** out, int device, const dcx_fk_desc* fk, int kernel_kind,
                     const float* kparams, const float* support_feat, const float* weights,
                     int64_t S, int32_t D, int32_t C);
/* The same on a stream, with room to grow (round 4).  When support_feat and weights are DEVICE pointers the rows are packed
 * by one kernel on `stream` (dropping zero rows, folding the kernel's constants, building the centred copy) and 16 bytes
 * come back - no bulk copy through the host; `stream` is synchronised for that read-back (the number of kept rows decides
 * the launch geometry of every later call).  Host pointers take the host path.  capacity >= S reserves storage so that
 * dcx_model_update can refill the model in place (0 = exactly S).  Results are bit-identical to dcx_model_create's. */
int dcx_model_create_ex(dcx_model** out, int device, const dcx_fk_desc* fk, int kernel_kind,
                        const float* kparams, const float* support_feat, const float* weights,
                        int64_t S, int32_t D, int32_t C, int64_t capacity, void* stream);
/* New supports / weights for an existing model (same transform, kernel, D, C): what DiffCo.train / fit_poly / update do to
 * the state every round of the reference's active-learning loop (collision_checkers.py:220-252).  Reuses the model's
 * storage when S <= its capacity (else reallocates after a device synchronisation) and every per-stream scratch buffer.
 * The model must not be in use by launches on OTHER streams or threads while it is updated; launches enqueued earlier on
 * `stream` are ordered before the refill, later ones see the new rows.                                               */
int dcx_model_update(dcx_model* m, const float* support_feat, const float* weights, int64_t S, void* stream);
void dcx_model_destroy(dcx_model* m);
/* any out pointer may be NULL; S_active = supports kept after dropping all-zero rows */
int dcx_model_info(const dcx_model* m, int64_t* S_active, int32_t* D, int32_t* C, int32_t* dof,
                   int32_t* device);

/* ---- the hot path ------------------------------------------------------------------ */
/* score[b, c] = sum_j K(T(q_b), support_j) * weights[j, c]
 * Replaces DiffCo.score/score_original kernel_perceptrons.py:359-370, DiffCo.poly_score :309-319,
 * MultiDiffCo.score / rbf_score deprecated/MultiDiffCo.py:118-123,156-169, DiffCoBeta.rbf_score
 * deprecated/DiffCoBeta.py:173-181.   q [B, dof] dev -> score [B, C] dev.                 */
int dcx_score(const dcx_model* m, const float* q, int64_t B, float* score, void* stream);

/* score as above and grad[b, :] = d( sum_c upstream[b,c] * score[b,c] ) / d q_b, in the same
 * pass (the reference obtains it by autograd: optim.py:101, 211-216).  upstream [B, C] dev or
 * NULL (= all ones).  score may be NULL.  grad [B, dof] dev.                              */
int dcx_score_grad(const dcx_model* m, const float* q, int64_t B, const float* upstream,
                   float* score, float* grad, void* stream);

/* score and the full Jacobian jac[b, c, :] = d score[b,c] / d q_b  ([B, C, dof] dev)
 * (torch.autograd.functional.jacobian callers, optim.py:211-216).                          */
int dcx_score_jac(const dcx_model* m, const float* q, int64_t B, float* score, float* jac,
                  void* stream);

/* Second derivatives: hess[b, i, k] = d2( sum_c upstream[b,c] * score[b,c] ) / d q_b[i] d q_b[k]   ([B, dof, dof] dev),
 * analytic (forward-mode tangents through the FK, the sweep and the reverse FK sweep: hess_kernel.hip) — what the
 * reference obtains by a double backward through dist_est for trust-constr's constraint Hessian (optim.py:380-391,
 * torch.autograd.functional.hessian).  upstream [B, C] dev or NULL (= all ones); grad [B, dof] dev or NULL receives
 * the gradient of the same function.  A configuration that sits exactly on a support (|x - s| = 0) takes no
 * contribution from that support for Polyharmonic kernels (the kernel is not twice differentiable there).
 * Every transform kind is accepted (a tree whose frames do not fit the LDS as (value, tangent) pairs keeps them in
 * stream-ordered global scratch); a batch of a few hundred points splits the supports across blocks like
 * dcx_score_grad does.  DCX_ERR_UNSUPPORTED only if one feature row in duals exceeds a block's LDS.                */
int dcx_score_hess(const dcx_model* m, const float* q, int64_t B, const float* upstream, float* grad, float* hess,
                   void* stream);

/* score as above (C == 1 models only) and the gradient of the hinge penalty the optimisers build on it:
 *   grad[b, :] = weight * 1[score_b - margin > 0] * d score_b / d q_b
 * i.e. d/dq of  weight * clamp(dist_est(q) - safety_margin, min=0).sum()  (optim.py:88-89, 97-101) in the
 * same pass that produces the score.  score may be NULL.                                                */
int dcx_score_hinge_grad(const dcx_model* m, const float* q, int64_t B, float margin, float weight,
                         float* score, float* grad, void* stream);
/* The same for ANY class count under per-class margins - the collision term the reference's Adam loops build on a
 * MultiDiffCo score (optim.py:88-89 with options['safety_margin'] a [C] tensor: scripts/2d_trajopt.py:94-102,
 * scripts/active.py:35, 65):
 *   grad[b, :] = d/dq_b of  weight * sum_c clamp(score[b, c] - margin[c], min=0)
 *              = weight * sum_c 1[score[b, c] - margin[c] > 0] * d score[b, c] / d q_b
 * margin: C HOST floats (they travel as kernel arguments).  score [B, C] dev out; with C > 1 it must not be NULL: the
 * class scores are swept first (the hinge's upstream is only known once every support has been seen), then the gradient
 * with that upstream - two launches, no synchronisation.  C == 1 is dcx_score_hinge_grad's single launch.          */
int dcx_score_hinge_grad_mc(const dcx_model* m, const float* q, int64_t B, const float* margin, float weight,
                            float* score, float* grad, void* stream);

/* ---- fused Adam trajectory step (caller of the path) ------------------------------- */
/* Batched restatement of the loop body of adam_traj_optimize (optim.py:86-127): R independent paths of W
 * waypoints are advanced by ONE Adam step on
 *   loss = w_diff * sum |cp[w+1]-cp[w]|^2 + w_collision * sum clamp(score - margin, 0)
 *        + w_max_move * sum_{segments, points} clamp(|cp[w+1]-cp[w]|^2 - max_speed^2, 0)
 *        + w_joint_limit * sum (clamp(lo - q, 0) + clamp(q - hi, 0))
 * with the endpoints' gradient zeroed, plus the reference's bookkeeping (lowest-loss path, best valid path,
 * per-path stop when constraint <= valid_tol and |grad| < grad_tol).  All pointers are device memory owned by
 * the caller; every array is dense fp32 unless noted.                                                    */
typedef struct dcx_traj_state {
    int32_t n_paths, n_waypoints;      /* R, W (W <= 1024)                                              */
    float* path;                       /* [R, W, dof]  in/out                                            */
    float* adam_m;                     /* [R, W, dof]  first moment  (zero before step 1)                */
    float* adam_v;                     /* [R, W, dof]  second moment (zero before step 1)                */
    const float* limits;               /* [dof, 2]     joint limits (lo, hi)                             */
    const float* col_score;            /* [R*W] ([R*W, C] for the _mc entry points) dist_est(path) of THIS step */
    const float* col_grad;             /* [R*W, dof]   hinge gradient of this step (dcx_score_hinge_grad)*/
    float* stats;                      /* [R, 8] out: loss, objective, constraint, |grad|, collision, max_move, joint_limit, 0 */
    float* lowest_loss;                /* [R] in/out (+inf before step 1)                                */
    float* lowest_obj;                 /* [R] out: objective at the lowest-loss step                     */
    float* lowest_path;                /* [R, W, dof] out: path AFTER the lowest-loss step               */
    float* best_valid_obj;             /* [R] in/out (+inf before step 1)                                */
    float* best_valid_path;            /* [R, W, dof] out: path AFTER the best valid step                */
    int32_t* done;                     /* [R] in/out: 1 = path stopped (frozen); 0 before step 1         */
    int32_t* steps;                    /* [R] in/out: steps actually taken                               */
} dcx_traj_state;

typedef struct dcx_traj_opts {
    float lr, beta1, beta2, eps;       /* torch.optim.Adam defaults: beta 0.9 / 0.999, eps 1e-8          */
    float w_diff, w_collision, w_max_move, w_joint_limit; /* 1, 10, 10, 10 in the reference (optim.py:19-22) */
    float safety_margin, max_speed;
    float valid_tol, grad_tol;         /* 1e-2, 1e-4 (optim.py:114, 126)                                  */
} dcx_traj_opts;

/* one Adam step; `step` is 1-based (bias correction).  col_score / col_grad must already hold this step's
 * collision term for the CURRENT path.                                                                   */
int dcx_traj_adam_step(int device, const dcx_fk_desc* fk, const dcx_traj_state* st, const dcx_traj_opts* opt,
                       int32_t step, void* stream);
/* n_iters iterations of {dcx_score_hinge_grad(model, path) -> dcx_traj_adam_step} enqueued back to back from
 * native code, starting at 1-based step `first_step`; `model` must be a C == 1 model whose transform is the
 * robot's FK.  st->col_score / st->col_grad are used as scratch ([R*W], [R*W, dof], non-const here).      */
int dcx_traj_adam_run(const dcx_model* model, const dcx_traj_state* st, const dcx_traj_opts* opt,
                      int32_t first_step, int32_t n_iters, void* stream);
/* The same loop on a model with ANY class count C (a MultiDiffCo score) under per-class margins:
 *   collision term = sum over waypoints and classes of clamp(score[w, c] - margin[c], 0)    (optim.py:88-89 with a [C] margin)
 * margin: C HOST floats, or NULL = opt->safety_margin for every class.  st->col_score is [R*W, C] here (scratch).
 * One persistent launch per <= 192 iterations where the two-sweep form of the trajectory kernel is compiled (D <= 24,
 * RQKernel(p = 2) / Polyharmonic(1), W <= 64), else three launches per iteration (class scores, hinge-gradient sweep,
 * step); the same arithmetic either way.  dcx_traj_adam_step_mc is the step half on its own: col_score [R*W, C] and
 * col_grad [R*W, dof] must hold dcx_score_hinge_grad_mc's outputs for the current path.                            */
int dcx_traj_adam_run_mc(const dcx_model* model, const dcx_traj_state* st, const dcx_traj_opts* opt, const float* margin,
                         int32_t first_step, int32_t n_iters, void* stream);
int dcx_traj_adam_step_mc(int device, const dcx_fk_desc* fk, const dcx_traj_state* st, const dcx_traj_opts* opt,
                          const float* margin, int32_t C, int32_t step, void* stream);

/* ---- escape from collision (caller of the path; SURVEY.md §8f-2 names it beside the trajectory step) -- */
/* Batched restatement of OptimSampler.optim_escape (scripts/escape.py:19-38): Adam on the configurations themselves,
 *   for step in range(n_steps):
 *       excess = sum(dist_est(q) - safety_margin)          # escape.py:26 (no clamp: the loop stops instead)
 *       if excess <= 0: break                              # :27-28
 *       [record q]; q <- Adam(q, d excess / d q); q <- post_transform(q)       # :29-36
 * entirely on the caller's stream: per step one fused score + gradient sweep (the gradient of sum_c score_c: the
 * margin is a constant) and one update launch; nothing is read back, the loop's decisions stay on the device.
 *   joint != 0: ONE loop for the whole batch, as the reference runs it - `excess` is the sum over all B configurations
 *               and all classes, and everybody stops together;  steps [1, 2]
 *   joint == 0: B independent loops advanced together - a configuration stops (and is left where it is) as soon as its
 *               own sum over the classes is <= 0;  steps [B, 2]
 * steps (int32, device, out): per loop (evaluations of dist_est = the reference's second return value, escape.py:38;
 *   Adam steps taken - one fewer when the loop stopped on its last evaluation).
 * history (device, out, may be NULL): [n_slots, B, dof] with n_slots = (record_freq > 0 ? ceil(n_steps / record_freq) : 0)
 *   + 1: slot s / record_freq holds q BEFORE the update of step s when record_freq divides s (escape.py:29-30), and the
 *   slot behind a loop's last record its final configuration (:37); later slots are not written.  q [B, dof] is advanced
 *   in place and holds the final configurations either way.
 * margin [C] device (NULL = 0).  wrap_mask bit i: coordinate i is wrapped to [-pi, pi) after every step - all ones below
 *   dof for post_transform = utils.wrap2pi (utils.py:51-52), bit 2 for utils.se2_wrap2pi (:54-55), 0 for None.
 * work: dcx_escape_work_bytes(model, B) bytes of device memory, the caller's (Adam moments, score and gradient of the
 *   current step, the list of the loops still running); initialised here.  No allocation.
 * compact_every = 0: no synchronisation either - every step sweeps all B configurations, stopped loops included (they are
 *   left alone by the update), and the call can be captured in a HIP graph.
 * compact_every = k > 0 (joint == 0): after every k-th step the loops that stopped are taken out of the sweep's batch.  The
 *   call then SYNCHRONISES the stream there (it reads how many loops are left to size the next launches) and returns as
 *   soon as none is; not capturable.  A configuration's arithmetic does not depend on its place in the batch; the launch
 *   geometry follows the batch size, so results agree with compact_every = 0 to fp32 rounding of the sums, not bit for bit. */
typedef struct dcx_escape_opts {
    float lr, beta1, beta2, eps;       /* torch.optim.Adam: lr 5e-2 (escape.py:12), betas 0.9 / 0.999, eps 1e-8        */
    int32_t n_steps;                   /* N_WAYPOINTS (escape.py:10): the most checks a loop makes                      */
    int32_t record_freq;               /* escape.py:13; 0 or None = final configuration only                            */
    int32_t joint;
    int32_t compact_every;             /* 0 = never; k > 0 (independent loops only): see above                          */
    uint64_t wrap_mask;
} dcx_escape_opts;
size_t dcx_escape_work_bytes(const dcx_model* m, int64_t B);
int dcx_escape_adam(const dcx_model* m, float* q, int64_t B, const float* margin, const dcx_escape_opts* opt, void* work,
                    size_t work_bytes, float* history, int32_t* steps, void* stream);

/* ---- kernel-perceptron trainer (producer of the path's state) ----------------------- */
/* DiffCo.train_perceptron kernel_perceptrons.py:98-137 and MultiDiffCo.train_perceptron
 * deprecated/MultiDiffCo.py:50-83 as one persistent launch: worst-margin search, lazily filled kernel rows,
 * margin fix / support retirement, until convergence or max_iteration.
 *   feats [N, D] dev   transformed samples          y [N, C] dev   labels (normally -1 / +1; any other value
 *                 is used as given: margin y*h, target beta^((1+y)/2)*y, like the reference's expressions)
 *   gains, hypothesis [N, C] dev in/out (zeros for a cold start, previous state for a jump start)
 *   kernel_matrix [N, N] dev in/out: zeros = "row not computed yet"; row i AND column i are filled when sample i
 *                 is first selected (K[i, :] = K[:, i] = k(x_i, X), kernel_perceptrons.py:117-119), so the sub-block
 *                 over the kept supports is complete even for a sample that was never selected itself
 *   info [2] dev out: iterations used; 1 if converged, 0 if not, -1 if the multi-workgroup form gave up on a
 *                 grid barrier (another kernel kept its workgroups from running for seconds)
 * For one label column and N <= 131072 a register-resident kernel, which keeps a label as its sign, is used: one
 * workgroup up to N = 4096, beyond that N / 1024 workgroups on as many CUs (a cooperative launch, one grid-wide barrier
 * per iteration; the same argmin sequence, bit for bit).  Whether the labels really are -1 / +1 is checked by those
 * kernels themselves, on the device (any other label: the generic loop, which uses y as given) - like every entry
 * point of this library the call does not synchronise the caller's stream, and its one-workgroup form can be captured
 * in a HIP graph (a stream that is being captured never takes the multi-workgroup form).  After info[1] == -1 gains,
 * hypothesis and kernel_matrix have been partly updated in place: re-initialise them before running again, e.g. with
 * dcx_train_perceptron_ex(..., DCX_TRAIN_ONE_WORKGROUP, ...), which cannot give up.  Debug knob "train_grid": 0 = one
 * workgroup only, 1 = several from N = 2048, 2 = the generic one-workgroup kernel whatever the labels.            */
int dcx_train_perceptron(int device, int kernel_kind, const float* kparams, float beta, const float* feats, int64_t N,
                         int32_t D, const float* y, int32_t C, float* gains, float* hypothesis, float* kernel_matrix,
                         int32_t max_iteration, int32_t* info, void* stream);
/* the same with per-call flags (thread-safe, unlike the process-wide knob): DCX_TRAIN_ONE_WORKGROUP = never the
 * multi-workgroup form for this call                                                                          */
#define DCX_TRAIN_ONE_WORKGROUP 1
int dcx_train_perceptron_ex(int device, int kernel_kind, const float* kparams, float beta, const float* feats, int64_t N,
                            int32_t D, const float* y, int32_t C, float* gains, float* hypothesis, float* kernel_matrix,
                            int32_t max_iteration, int32_t* info, int32_t flags, void* stream);

/* ---- pieces of the path exposed on their own ---------------------------------------- */
/* X[b] = T(q_b): model.*.fkine (see DCX_FK_*).  q [B, dof] dev -> X [B, n_points*point_dim] dev */
int dcx_fkine(int device, const dcx_fk_desc* fk, const float* q, int64_t B, float* X, void* stream);
/* gq[b] = J_T(q_b)^T gX[b]  (autograd of fkine).  gX [B, D] dev -> gq [B, dof] dev          */
int dcx_fkine_vjp(int device, const dcx_fk_desc* fk, const float* q, const float* gX, int64_t B,
                  float* gq, void* stream);
/* Link transforms on their own - the two helpers user-written robot classes build their FK from (model.py:230, 437):
 *   dcx_dh_frames     utils.DH2mat utils.py:66-75: T[b, i] = Rz(theta) Tz(d_i) Tx(a_i) Rx(alpha_i), theta = q[b, i]
 *                     rows (c, -s ca, s sa, a c), (s, c ca, -c sa, a s), (0, sa, ca, d), (0, 0, 0, 1)
 *                     q [B, dof] dev; a, d, sin_alpha, cos_alpha [dof] dev -> T [B, dof, 4, 4] dev
 *   dcx_euler_frames  utils.euler2mat utils.py:15-38: R[b] = Rz(yaw) Ry(pitch) Rx(roll), phi[b] = (roll, pitch, yaw)
 *                     phi [B, 3] dev -> R [B, 3, 3] dev
 * and their autograd with respect to the angles: gq[b, i] = sum_rc gT[b, i, r, c] dT[b, i, r, c] / d q[b, i];
 * gphi[b, k] = sum_rc gR[b, r, c] dR[b, r, c] / d phi[b, k].  HBM-bound (64 B / 36 B written per frame).           */
int dcx_dh_frames(int device, const float* q, int64_t B, int32_t dof, const float* a, const float* d, const float* sin_alpha,
                  const float* cos_alpha, float* T, void* stream);
int dcx_dh_frames_vjp(int device, const float* q, int64_t B, int32_t dof, const float* a, const float* sin_alpha,
                      const float* cos_alpha, const float* gT, float* gq, void* stream);
int dcx_euler_frames(int device, const float* phi, int64_t B, float* R, void* stream);
int dcx_euler_frames_vjp(int device, const float* phi, const float* gR, int64_t B, float* gphi, void* stream);
/* K[b, j] = K(x_b, s_j): KernelFunc.__call__ kernel.py:17-29, 49-57, 73-79 (used by the trainer's
 * row fill kernel_perceptrons.py:117-119 and fit_poly :271-287).  x [B, D], s [S, D] dev -> K [B, S] dev */
int dcx_kernel_matrix(int device, int kernel_kind, const float* kparams, const float* x, int64_t B,
                      const float* s, int64_t S, int32_t D, float* K, void* stream);

/* x with A x = B: the S x S system fit_poly ends in (torch.linalg.solve at kernel_perceptrons.py:283, deprecated/DiffCo.py:162,
 * deprecated/MultiDiffCo.py:151 - add `reg` to A's diagonal first).  LU with partial pivoting in fp64, ONE launch.
 * A [n, n], B [n, nrhs] -> X [n, nrhs], all row-major fp32 on the device; 1 <= n <= DCX_SOLVE_MAX_N, 1 <= nrhs <= 64.
 * work: dcx_solve_work_bytes(n, nrhs) bytes of device memory, the caller's (no allocation, no synchronisation here; the
 * stream may be under capture).  info [2] int32 on the device: info[0] = 0, or k > 0 when the k-th pivot is exactly zero
 * (LAPACK's info; X is then not a solution), or -1 when a grid barrier gave up (another kernel held the CUs for seconds):
 * call again with DCX_SOLVE_ONE_WORKGROUP in flags.  */
#define DCX_SOLVE_MAX_N 4096
#define DCX_SOLVE_ONE_WORKGROUP 1
size_t dcx_solve_work_bytes(int64_t n, int64_t nrhs);
int dcx_solve(int device, const float* A, const float* B, int64_t n, int64_t nrhs, float* X, void* work, size_t work_bytes,
              int32_t* info, int32_t flags, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DCX_H */
