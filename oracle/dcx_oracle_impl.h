/*
 * dcx_oracle_impl.h — body of the CPU oracle, included twice by dcx_oracle.c with
 *   REAL = float  / FN(x) = x##_f32     and     REAL = double / FN(x) = x##_f64.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/README.md): a plain-C restatement of the reference's
 * algorithm for the score(+grad) path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may call it.  The shipped path (diffco_amd/, libdcx.so) never does.
 *
 * Each function cites the reference lines (under /root/reference) it restates.
 */

/* ---- 3x4 rigid transforms [R|t], row-major ----------------------------------------- */
static void FN(m34_mul)(const REAL* A, const REAL* B, REAL* C) {
    /* torch.bmm(tmp_tf, tfs[:, i]) of two homogeneous transforms — model.py:237, 442 */
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 4; ++c) {
            REAL s = (c == 3) ? A[r * 4 + 3] : (REAL)0;
            for (int k = 0; k < 3; ++k) s += A[r * 4 + k] * B[k * 4 + c];
            C[r * 4 + c] = s;
        }
    }
}

static void FN(dh_link)(REAL th, REAL a, REAL d, REAL sa, REAL ca, REAL* T) {
    /* utils.DH2mat — utils.py:66-75 */
    REAL c = MATH(cos)(th), s = MATH(sin)(th);
    T[0] = c;  T[1] = -s * ca; T[2] = s * sa;   T[3] = a * c;
    T[4] = s;  T[5] = c * ca;  T[6] = -c * sa;  T[7] = a * s;
    T[8] = 0;  T[9] = sa;      T[10] = ca;      T[11] = d;
}

static void FN(euler_zyx)(REAL roll, REAL pitch, REAL yaw, REAL* Rm, REAL* dRoll, REAL* dPitch, REAL* dYaw) {
    /* utils.euler2mat — utils.py:15-38: R = rz @ ry @ rx with phi = (roll, pitch, yaw) */
    REAL sx = MATH(sin)(roll), cx = MATH(cos)(roll);
    REAL sy = MATH(sin)(pitch), cy = MATH(cos)(pitch);
    REAL sz = MATH(sin)(yaw), cz = MATH(cos)(yaw);
    REAL rx[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx}, drx[9] = {0, 0, 0, 0, -sx, -cx, 0, cx, -sx};
    REAL ry[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy}, dry[9] = {-sy, 0, cy, 0, 0, 0, -cy, 0, -sy};
    REAL rz[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1}, drz[9] = {-sz, -cz, 0, cz, -sz, 0, 0, 0, 0};
    const REAL* zs[4] = {rz, rz, rz, drz};
    const REAL* ys[4] = {ry, ry, dry, ry};
    const REAL* xs[4] = {rx, drx, rx, rx};
    REAL* outs[4] = {Rm, dRoll, dPitch, dYaw};
    for (int v = 0; v < 4; ++v) {
        if (!outs[v]) continue;
        REAL zy[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                REAL s = 0;
                for (int k = 0; k < 3; ++k) s += zs[v][i * 3 + k] * ys[v][k * 3 + j];
                zy[i * 3 + j] = s;
            }
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                REAL s = 0;
                for (int k = 0; k < 3; ++k) s += zy[i * 3 + k] * xs[v][k * 3 + j];
                outs[v][i * 3 + j] = s;
            }
    }
}

/* joint_pose of one URDF joint — rigid_body.py:100-126: rot = fixed_rotation @ {x,y,z}_rot(v) and
 * trans = fixed_translation for revolute/continuous (spatial_vector_algebra.py x_rot/y_rot/z_rot),
 * rot = fixed_rotation and trans = fixed_translation + fixed_rotation @ (axis * v) for prismatic; a joint
 * without a q entry keeps joint_pose = [fixed_rotation | fixed_translation] (rigid_body.py:150-176).
 * dA (optional) = d joint_pose / d v. */
static void FN(tree_joint)(const dcx_fk_desc* fk, int j, REAL v, REAL* A, REAL* dA) {
    const float* F = fk->t_fixed[j];
    REAL M[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}, dM[12] = {0};
    REAL c = MATH(cos)(v), s = MATH(sin)(v);
    switch (fk->t_type[j]) {
    case DCX_J_REV_X: M[5] = c; M[6] = -s; M[9] = s; M[10] = c; dM[5] = -s; dM[6] = -c; dM[9] = c; dM[10] = -s; break;
    case DCX_J_REV_Y: M[0] = c; M[2] = s; M[8] = -s; M[10] = c; dM[0] = -s; dM[2] = c; dM[8] = -c; dM[10] = -s; break;
    case DCX_J_REV_Z: M[0] = c; M[1] = -s; M[4] = s; M[5] = c; dM[0] = -s; dM[1] = -c; dM[4] = c; dM[5] = -s; break;
    case DCX_J_PRISMATIC:
        for (int r = 0; r < 3; ++r) { M[r * 4 + 3] = (REAL)fk->t_axis[j][r] * v; dM[r * 4 + 3] = (REAL)fk->t_axis[j][r]; }
        break;
    default: break;
    }
    REAL Fr[12];
    for (int e = 0; e < 12; ++e) Fr[e] = (REAL)F[e];
    FN(m34_mul)(Fr, M, A);
    if (dA) {
        /* derivative of a product with a constant left factor: the homogeneous "1" of M does not vary */
        for (int r = 0; r < 3; ++r)
            for (int cc = 0; cc < 4; ++cc) {
                REAL acc = 0;
                for (int k = 0; k < 3; ++k) acc += Fr[r * 4 + k] * dM[k * 4 + cc];
                dA[r * 4 + cc] = acc;
            }
    }
}

static int FN(tree_slot)(const dcx_fk_desc* fk, int k, int r) {
    return fk->t_coord_major ? r * fk->n_points + k : 3 * k + r;
}

/* ---- forward kinematics of ONE configuration ---------------------------------------- */
/* X: [n_points*point_dim].  frames (optional, DH only): [n_chains][chain_len+1][12], entry 0 = base,
 * entry i+1 = cumulative transform after joint i. */
static void FN(fk_one)(const dcx_fk_desc* fk, const REAL* q, REAL* X, REAL* frames) {
    switch (fk->kind) {
    case DCX_FK_NONE:
        for (int i = 0; i < fk->dof; ++i) X[i] = q[i];
        break;
    case DCX_FK_PLANAR: {
        /* RevolutePlanarRobot.fkine — model.py:40-48: cumsum(q), cumsum(l*cos), cumsum(l*sin) */
        REAL phi = 0, x = 0, y = 0;
        for (int i = 0; i < fk->dof; ++i) {
            phi += q[i];
            x += (REAL)fk->link_length[i] * MATH(cos)(phi);
            y += (REAL)fk->link_length[i] * MATH(sin)(phi);
            X[2 * i] = x;
            X[2 * i + 1] = y;
        }
    } break;
    case DCX_FK_DH: {
        /* BaxterLeftArmFK.fkine model.py:225-241; BaxterDualArmFK.fkine :366-383 (bases);
         * PandaFK.fkine :430-453 (extra finger points = last frame @ offset); DualPandaFK :486-502 */
        for (int c = 0; c < fk->n_chains; ++c) {
            REAL T[12], A[12], N[12];
            for (int e = 0; e < 12; ++e) T[e] = (REAL)fk->base[c][e];
            if (frames) for (int e = 0; e < 12; ++e) frames[(c * (DCX_MAX_JOINTS + 1)) * 12 + e] = T[e];
            for (int i = 0; i < fk->chain_len[c]; ++i) {
                REAL th = q[fk->joint_q[c][i]] + (REAL)fk->theta0[c][i]; /* model.py:229 */
                FN(dh_link)(th, (REAL)fk->a[c][i], (REAL)fk->d[c][i], (REAL)fk->sin_alpha[c][i],
                            (REAL)fk->cos_alpha[c][i], A);
                FN(m34_mul)(T, A, N);
                for (int e = 0; e < 12; ++e) T[e] = N[e];
                if (frames)
                    for (int e = 0; e < 12; ++e) frames[(c * (DCX_MAX_JOINTS + 1) + i + 1) * 12 + e] = T[e];
                for (int k = 0; k < fk->n_points; ++k) {
                    if (fk->pt_chain[k] != c || fk->pt_frame[k] != i) continue;
                    for (int r = 0; r < 3; ++r)
                        X[3 * k + r] = T[r * 4 + 3] + T[r * 4 + 0] * (REAL)fk->pt_off[k][0] +
                                       T[r * 4 + 1] * (REAL)fk->pt_off[k][1] + T[r * 4 + 2] * (REAL)fk->pt_off[k][2];
                }
            }
        }
    } break;
    case DCX_FK_SE2: {
        /* RigidPlanarBody.fkine — model.py:90-93 with utils.rot_2d utils.py:40-48 */
        REAL c = MATH(cos)(q[2]), s = MATH(sin)(q[2]);
        for (int k = 0; k < fk->n_points; ++k) {
            REAL kx = (REAL)fk->keypoints[k][0], ky = (REAL)fk->keypoints[k][1];
            X[2 * k] = c * kx - s * ky + q[0];
            X[2 * k + 1] = s * kx + c * ky + q[1];
        }
    } break;
    case DCX_FK_SE3: {
        /* RigidBody.fkine — model.py:156-159 */
        REAL Rm[9];
        FN(euler_zyx)(q[3], q[4], q[5], Rm, 0, 0, 0);
        for (int k = 0; k < fk->n_points; ++k)
            for (int r = 0; r < 3; ++r)
                X[3 * k + r] = Rm[r * 3 + 0] * (REAL)fk->keypoints[k][0] + Rm[r * 3 + 1] * (REAL)fk->keypoints[k][1] +
                               Rm[r * 3 + 2] * (REAL)fk->keypoints[k][2] + q[r];
    } break;
    case DCX_FK_TREE: {
        /* RigidBody.forward_kinematics rigid_body.py:82-140 unrolled along each root-to-leaf path; features =
         * link-origin translations, collision_checkers.py:385-393 */
        int j0 = 0;
        for (int c = 0; c < fk->t_n_chains; ++c) {
            REAL T[12], A[12], N[12];
            for (int e = 0; e < 12; ++e) T[e] = (REAL)fk->t_base[c][e];
            for (int i = 0; i < fk->t_chain_len[c]; ++i) {
                int j = j0 + i;
                REAL v = (fk->t_type[j] == DCX_J_FIXED) ? (REAL)0
                                                        : (REAL)fk->t_scale[j] * q[fk->t_q[j]] + (REAL)fk->t_offset[j];
                FN(tree_joint)(fk, j, v, A, 0);
                FN(m34_mul)(T, A, N);
                for (int e = 0; e < 12; ++e) T[e] = N[e];
                for (int k = 0; k < fk->n_points; ++k) {
                    if (fk->pt_chain[k] != c || fk->pt_frame[k] != i) continue;
                    for (int r = 0; r < 3; ++r)
                        X[FN(tree_slot)(fk, k, r)] = T[r * 4 + 3] + T[r * 4 + 0] * (REAL)fk->pt_off[k][0] +
                                                     T[r * 4 + 1] * (REAL)fk->pt_off[k][1] +
                                                     T[r * 4 + 2] * (REAL)fk->pt_off[k][2];
                }
            }
            j0 += fk->t_chain_len[c];
        }
    } break;
    default: break;
    }
}

/* gq = J^T gX for ONE configuration: what autograd of the fkine graph returns (optim.py:101).
 * Analytic forms of SURVEY.md §8a-G: revolute DH joint i moves point p by z_{i-1} x (p - o_{i-1}). */
static void FN(fk_vjp_one)(const dcx_fk_desc* fk, const REAL* q, const REAL* gX, REAL* gq) {
    for (int i = 0; i < fk->dof; ++i) gq[i] = 0;
    switch (fk->kind) {
    case DCX_FK_NONE:
        for (int i = 0; i < fk->dof; ++i) gq[i] = gX[i];
        break;
    case DCX_FK_PLANAR: {
        /* d(x_k, y_k)/dq_i = sum_{j=i..k} l_j (-sin phi_j, cos phi_j) */
        REAL phi[DCX_MAX_DOF], acc = 0;
        for (int i = 0; i < fk->dof; ++i) { acc += q[i]; phi[i] = acc; }
        for (int i = 0; i < fk->dof; ++i)
            for (int k = i; k < fk->dof; ++k)
                for (int j = i; j <= k; ++j)
                    gq[i] += (REAL)fk->link_length[j] * (-MATH(sin)(phi[j]) * gX[2 * k] + MATH(cos)(phi[j]) * gX[2 * k + 1]);
    } break;
    case DCX_FK_DH: {
        REAL X[DCX_MAX_POINTS * 3];
        REAL frames[DCX_MAX_CHAINS * (DCX_MAX_JOINTS + 1) * 12];
        FN(fk_one)(fk, q, X, frames);
        for (int k = 0; k < fk->n_points; ++k) {
            int c = fk->pt_chain[k];
            for (int i = 0; i <= fk->pt_frame[k]; ++i) {
                const REAL* P = &frames[(c * (DCX_MAX_JOINTS + 1) + i) * 12]; /* frame before joint i */
                REAL z[3] = {P[2], P[6], P[10]}, o[3] = {P[3], P[7], P[11]};
                REAL r[3] = {X[3 * k] - o[0], X[3 * k + 1] - o[1], X[3 * k + 2] - o[2]};
                REAL v[3] = {z[1] * r[2] - z[2] * r[1], z[2] * r[0] - z[0] * r[2], z[0] * r[1] - z[1] * r[0]};
                gq[fk->joint_q[c][i]] += v[0] * gX[3 * k] + v[1] * gX[3 * k + 1] + v[2] * gX[3 * k + 2];
            }
        }
    } break;
    case DCX_FK_TREE: {
        /* product rule, one joint at a time: dT/dv_m = A_0 .. A_{m-1} (dA_m/dv) A_{m+1} .. ; the chain rule
         * through v = scale*q + offset adds the factor scale.  O(n^2) per chain, deliberately naive. */
        int j0 = 0;
        for (int c = 0; c < fk->t_n_chains; ++c) {
            int n = fk->t_chain_len[c];
            for (int m = 0; m < n; ++m) {
                if (fk->t_type[j0 + m] == DCX_J_FIXED) continue;
                REAL T[12], A[12], dA[12], N[12];
                for (int e = 0; e < 12; ++e) T[e] = (REAL)fk->t_base[c][e];
                for (int i = 0; i < n; ++i) {
                    int j = j0 + i;
                    REAL v = (fk->t_type[j] == DCX_J_FIXED) ? (REAL)0
                                                            : (REAL)fk->t_scale[j] * q[fk->t_q[j]] + (REAL)fk->t_offset[j];
                    FN(tree_joint)(fk, j, v, A, dA);
                    if (i == m) {
                        /* T (3x4, implicit last row 0 0 0 1) times dA (last row 0 0 0 0) */
                        for (int r = 0; r < 3; ++r)
                            for (int cc = 0; cc < 4; ++cc) {
                                REAL acc = 0;
                                for (int k = 0; k < 3; ++k) acc += T[r * 4 + k] * dA[k * 4 + cc];
                                N[r * 4 + cc] = acc;
                            }
                    } else if (i > m) {
                        /* derivative transform [dR | dt] (homogeneous row 0) times a rigid transform [R | t; 0 0 0 1]:
                         * column 3 = dR t + dt */
                        for (int r = 0; r < 3; ++r)
                            for (int cc = 0; cc < 4; ++cc) {
                                REAL acc = (cc == 3) ? T[r * 4 + 3] : (REAL)0;
                                for (int k = 0; k < 3; ++k) acc += T[r * 4 + k] * A[k * 4 + cc];
                                N[r * 4 + cc] = acc;
                            }
                    } else {
                        FN(m34_mul)(T, A, N);
                    }
                    for (int e = 0; e < 12; ++e) T[e] = N[e];
                    if (i < m) continue;
                    for (int k = 0; k < fk->n_points; ++k) {
                        if (fk->pt_chain[k] != c || fk->pt_frame[k] != i) continue;
                        for (int r = 0; r < 3; ++r) {
                            REAL dp = T[r * 4 + 3] + T[r * 4 + 0] * (REAL)fk->pt_off[k][0] +
                                      T[r * 4 + 1] * (REAL)fk->pt_off[k][1] + T[r * 4 + 2] * (REAL)fk->pt_off[k][2];
                            gq[fk->t_q[j0 + m]] += (REAL)fk->t_scale[j0 + m] * dp * gX[FN(tree_slot)(fk, k, r)];
                        }
                    }
                }
            }
            j0 += n;
        }
    } break;
    case DCX_FK_SE2: {
        REAL c = MATH(cos)(q[2]), s = MATH(sin)(q[2]);
        for (int k = 0; k < fk->n_points; ++k) {
            REAL kx = (REAL)fk->keypoints[k][0], ky = (REAL)fk->keypoints[k][1];
            gq[0] += gX[2 * k];
            gq[1] += gX[2 * k + 1];
            gq[2] += gX[2 * k] * (-s * kx - c * ky) + gX[2 * k + 1] * (c * kx - s * ky);
        }
    } break;
    case DCX_FK_SE3: {
        REAL dR[3][9];
        FN(euler_zyx)(q[3], q[4], q[5], 0, dR[0], dR[1], dR[2]);
        for (int k = 0; k < fk->n_points; ++k) {
            for (int r = 0; r < 3; ++r) gq[r] += gX[3 * k + r];
            for (int a = 0; a < 3; ++a)
                for (int r = 0; r < 3; ++r)
                    gq[3 + a] += gX[3 * k + r] * (dR[a][r * 3 + 0] * (REAL)fk->keypoints[k][0] +
                                                  dR[a][r * 3 + 1] * (REAL)fk->keypoints[k][1] +
                                                  dR[a][r * 3 + 2] * (REAL)fk->keypoints[k][2]);
        }
    } break;
    default: break;
    }
}

/* ---- pairwise kernel: value and g such that dK/dx = g * (x - s) --------------------- */
static inline void FN(kernel_eval)(int kind, const REAL* kp, REAL d2, REAL* val, REAL* g) {
    switch (kind) {
    case DCX_K_RQ: {
        /* RQKernel.__call__ — kernel.py:24-25: 1/(1+gamma/p*d2)**p */
        REAL t = 1 + kp[0] / kp[1] * d2;
        *val = MATH(pow)(t, -kp[1]);
        *g = -2 * kp[0] * MATH(pow)(t, -kp[1] - 1);
    } break;
    case DCX_K_POLY: {
        /* Polyharmonic — kernel.py:60-79; r=0: value 0 (NaN->0, :65) and zero sub-gradient (cdist backward) */
        int k = (int)kp[0];
        REAL eps = kp[1], r = MATH(sqrt)(d2);
        if (r == 0) { *val = 0; *g = 0; break; }
        if (k % 2 == 0) {
            REAL lg = MATH(log)(r), rk2 = MATH(pow)(r, (REAL)(k - 2));
            *val = rk2 * r * r * lg / eps;
            *g = rk2 * (k * lg + 1) / eps;
        } else if (k == 1) {
            *val = r / eps;
            *g = 1 / (eps * r);
        } else {
            REAL rk2 = MATH(pow)(r, (REAL)(k - 2));
            *val = rk2 * r * r / eps;
            *g = k * rk2 / eps;
        }
    } break;
    case DCX_K_MQ: {
        /* MultiQuadratic — kernel.py:54: sqrt(sum(diff**2)/eps**2 + 1) */
        REAL v = MATH(sqrt)(d2 / (kp[0] * kp[0]) + 1);
        *val = v;
        *g = 1 / (kp[0] * kp[0] * v);
    } break;
    default: *val = 0; *g = 0;
    }
}

/* ---- exported ---------------------------------------------------------------------- */
void FN(orc_fkine)(const dcx_fk_desc* fk, const REAL* q, int64_t B, REAL* X) {
    const int D = fk->n_points * fk->point_dim;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) FN(fk_one)(fk, q + b * fk->dof, X + b * D, 0);
}

void FN(orc_fkine_vjp)(const dcx_fk_desc* fk, const REAL* q, const REAL* gX, int64_t B, REAL* gq) {
    const int D = fk->n_points * fk->point_dim;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) FN(fk_vjp_one)(fk, q + b * fk->dof, gX + b * D, gq + b * fk->dof);
}

void FN(orc_kernel_matrix)(int kind, const REAL* kp, const REAL* x, int64_t B, const REAL* s, int64_t S, int D, REAL* K) {
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b)
        for (int64_t j = 0; j < S; ++j) {
            REAL d2 = 0, g;
            for (int k = 0; k < D; ++k) { REAL dl = x[b * D + k] - s[j * D + k]; d2 += dl * dl; }
            FN(kernel_eval)(kind, kp, d2, &K[b * S + j], &g);
        }
}

/*
 * score[b,c] = sum_j K(T(q_b), sup_j) W[j,c]                 (kernel_perceptrons.py:319, 369;
 *                                                             deprecated/MultiDiffCo.py:169)
 * grad[b,:]  = d(sum_c up[b,c] score[b,c]) / d q_b           (autograd in the reference)
 * jac[b,c,:] = d score[b,c] / d q_b
 * upstream NULL = ones; score / grad / jac may be NULL.
 */
void FN(orc_score_grad)(const dcx_fk_desc* fk, int kind, const REAL* kp, const REAL* sup, const REAL* W, int64_t S,
                        int D, int C, const REAL* q, int64_t B, const REAL* upstream, REAL* score, REAL* grad,
                        REAL* jac) {
    const int dof = fk->dof;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        REAL X[DCX_MAX_D], gX[DCX_MAX_D], sc[DCX_MAX_C];
        REAL gXc[DCX_MAX_C][DCX_MAX_D];
        FN(fk_one)(fk, q + b * dof, X, 0);
        for (int k = 0; k < D; ++k) gX[k] = 0;
        for (int c = 0; c < C; ++c) {
            sc[c] = 0;
            if (jac) for (int k = 0; k < D; ++k) gXc[c][k] = 0;
        }
        for (int64_t j = 0; j < S; ++j) {
            REAL dl[DCX_MAX_D], d2 = 0, val, g;
            for (int k = 0; k < D; ++k) { dl[k] = X[k] - sup[j * D + k]; d2 += dl[k] * dl[k]; }
            FN(kernel_eval)(kind, kp, d2, &val, &g);
            REAL wbar = 0;
            for (int c = 0; c < C; ++c) {
                sc[c] += val * W[j * C + c];
                wbar += (upstream ? upstream[b * C + c] : (REAL)1) * W[j * C + c];
                if (jac) for (int k = 0; k < D; ++k) gXc[c][k] += W[j * C + c] * g * dl[k];
            }
            for (int k = 0; k < D; ++k) gX[k] += wbar * g * dl[k];
        }
        if (score) for (int c = 0; c < C; ++c) score[b * C + c] = sc[c];
        if (grad) FN(fk_vjp_one)(fk, q + b * dof, gX, grad + b * dof);
        if (jac) for (int c = 0; c < C; ++c) FN(fk_vjp_one)(fk, q + b * dof, gXc[c], jac + (b * C + c) * dof);
    }
}
