// fk_device.h — forward kinematics and its transpose-Jacobian product on the device.
//
// One lane = one configuration.  Everything that is indexed at run time lives in LDS in
// "column" layout (element e of lane l at base[e*64 + l]: conflict-free, no scratch):
//   q row    : sQ[l*dof + i]             (staged coalesced from HBM by the caller)
//   features : sX[k*64 + l]              (control-point coordinates, D = n_points*point_dim)
//   frames   : sF[(6*j + e)*64 + l]      (DH: axis z_{j-1} (e=0..2) and origin o_{j-1} (e=3..5)
//                                          of the frame joint j rotates in; planar: cos/sin phi_j)
// The FK description is read through the constant address space, so every parameter load
// is a scalar (s_load) broadcast: all lanes run the same chain.
//
// Reference semantics restated here (paths under /root/reference/diffco):
//   utils.DH2mat utils.py:66-75; BaxterLeftArmFK.fkine model.py:225-241; BaxterDualArmFK.fkine
//   model.py:366-383; PandaFK.fkine model.py:430-453 (robot_fkine.py:428-444);
//   DualPandaFK.fkine model.py:486-502; RevolutePlanarRobot.fkine model.py:40-48;
//   RigidPlanarBody.fkine model.py:90-93; RigidBody.fkine model.py:156-159 (utils.euler2mat
//   utils.py:15-38).  The vjp is the analytic gradient of SURVEY.md §8a-G.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dcx.h"

namespace dcx {

typedef const __attribute__((address_space(4))) dcx_fk_desc* fk_cptr;

__device__ __forceinline__ fk_cptr as_const(const dcx_fk_desc* p) { return (fk_cptr)(uintptr_t)p; }

// LDS floats per lane the FK needs for its frames.
__host__ __device__ inline int fk_frame_floats(const dcx_fk_desc& fk) {
    if (fk.kind == DCX_FK_DH) {
        int j = 0;
        for (int c = 0; c < fk.n_chains; ++c) j += fk.chain_len[c];
        return 6 * j;
    }
    if (fk.kind == DCX_FK_PLANAR) return 2 * fk.dof;
    return 0;
}

// ---- forward: q (LDS row) -> X (LDS column), frames (LDS column) ------------------------
__device__ inline void fk_forward(fk_cptr fk, const float* sQrow, float* sXcol, float* sFcol) {
    const int kind = fk->kind;
    if (kind == DCX_FK_NONE) {
        const int dof = fk->dof;
        for (int i = 0; i < dof; ++i) sXcol[i * 64] = sQrow[i];
    } else if (kind == DCX_FK_PLANAR) {
        const int dof = fk->dof;
        float phi = 0.f, x = 0.f, y = 0.f;
        for (int i = 0; i < dof; ++i) {
            phi += sQrow[i];
            float s, c;
            sincosf(phi, &s, &c);
            const float l = fk->link_length[i];
            x = fmaf(l, c, x);
            y = fmaf(l, s, y);
            sXcol[(2 * i) * 64] = x;
            sXcol[(2 * i + 1) * 64] = y;
            sFcol[(2 * i) * 64] = c;
            sFcol[(2 * i + 1) * 64] = s;
        }
    } else if (kind == DCX_FK_DH) {
        const int n_pts = fk->n_points;
        int jbase = 0;
        for (int ch = 0; ch < fk->n_chains; ++ch) {
            float r00 = fk->base[ch][0], r01 = fk->base[ch][1], r02 = fk->base[ch][2], t0 = fk->base[ch][3];
            float r10 = fk->base[ch][4], r11 = fk->base[ch][5], r12 = fk->base[ch][6], t1 = fk->base[ch][7];
            float r20 = fk->base[ch][8], r21 = fk->base[ch][9], r22 = fk->base[ch][10], t2 = fk->base[ch][11];
            const int len = fk->chain_len[ch];
            for (int i = 0; i < len; ++i) {
                float* f = sFcol + (6 * (jbase + i)) * 64;  // frame joint i rotates in
                f[0] = r02; f[64] = r12; f[128] = r22; f[192] = t0; f[256] = t1; f[320] = t2;
                const float th = sQrow[fk->joint_q[ch][i]] + fk->theta0[ch][i];
                float s, c;
                sincosf(th, &s, &c);
                const float a = fk->a[ch][i], d = fk->d[ch][i];
                const float sa = fk->sin_alpha[ch][i], ca = fk->cos_alpha[ch][i];
                // T <- T * [[c, -s ca,  s sa, a c], [s, c ca, -c sa, a s], [0, sa, ca, d]]
                const float m01 = -s * ca, m02 = s * sa, m11 = c * ca, m12 = -c * sa;
                const float ac = a * c, as = a * s;
                t0 = fmaf(r00, ac, fmaf(r01, as, fmaf(r02, d, t0)));
                t1 = fmaf(r10, ac, fmaf(r11, as, fmaf(r12, d, t1)));
                t2 = fmaf(r20, ac, fmaf(r21, as, fmaf(r22, d, t2)));
                const float n00 = fmaf(r00, c, r01 * s), n10 = fmaf(r10, c, r11 * s), n20 = fmaf(r20, c, r21 * s);
                const float n01 = fmaf(r00, m01, fmaf(r01, m11, r02 * sa));
                const float n11 = fmaf(r10, m01, fmaf(r11, m11, r12 * sa));
                const float n21 = fmaf(r20, m01, fmaf(r21, m11, r22 * sa));
                const float n02 = fmaf(r00, m02, fmaf(r01, m12, r02 * ca));
                const float n12 = fmaf(r10, m02, fmaf(r11, m12, r12 * ca));
                const float n22 = fmaf(r20, m02, fmaf(r21, m12, r22 * ca));
                r00 = n00; r01 = n01; r02 = n02; r10 = n10; r11 = n11; r12 = n12; r20 = n20; r21 = n21; r22 = n22;
                for (int k = 0; k < n_pts; ++k) {
                    if (fk->pt_chain[k] != ch || fk->pt_frame[k] != i) continue;
                    const float ox = fk->pt_off[k][0], oy = fk->pt_off[k][1], oz = fk->pt_off[k][2];
                    sXcol[(3 * k) * 64] = fmaf(r00, ox, fmaf(r01, oy, fmaf(r02, oz, t0)));
                    sXcol[(3 * k + 1) * 64] = fmaf(r10, ox, fmaf(r11, oy, fmaf(r12, oz, t1)));
                    sXcol[(3 * k + 2) * 64] = fmaf(r20, ox, fmaf(r21, oy, fmaf(r22, oz, t2)));
                }
            }
            jbase += len;
        }
    } else if (kind == DCX_FK_SE2) {
        const float x = sQrow[0], y = sQrow[1];
        float s, c;
        sincosf(sQrow[2], &s, &c);
        const int n_pts = fk->n_points;
        for (int k = 0; k < n_pts; ++k) {
            const float kx = fk->keypoints[k][0], ky = fk->keypoints[k][1];
            sXcol[(2 * k) * 64] = fmaf(c, kx, fmaf(-s, ky, x));
            sXcol[(2 * k + 1) * 64] = fmaf(s, kx, fmaf(c, ky, y));
        }
    } else if (kind == DCX_FK_SE3) {
        float sx, cx, sy, cy, sz, cz;
        sincosf(sQrow[3], &sx, &cx);
        sincosf(sQrow[4], &sy, &cy);
        sincosf(sQrow[5], &sz, &cz);
        // R = Rz(yaw) Ry(pitch) Rx(roll)
        const float r00 = cz * cy, r01 = cz * sy * sx - sz * cx, r02 = cz * sy * cx + sz * sx;
        const float r10 = sz * cy, r11 = sz * sy * sx + cz * cx, r12 = sz * sy * cx - cz * sx;
        const float r20 = -sy, r21 = cy * sx, r22 = cy * cx;
        const float x = sQrow[0], y = sQrow[1], z = sQrow[2];
        const int n_pts = fk->n_points;
        for (int k = 0; k < n_pts; ++k) {
            const float kx = fk->keypoints[k][0], ky = fk->keypoints[k][1], kz = fk->keypoints[k][2];
            sXcol[(3 * k) * 64] = fmaf(r00, kx, fmaf(r01, ky, fmaf(r02, kz, x)));
            sXcol[(3 * k + 1) * 64] = fmaf(r10, kx, fmaf(r11, ky, fmaf(r12, kz, y)));
            sXcol[(3 * k + 2) * 64] = fmaf(r20, kx, fmaf(r21, ky, fmaf(r22, kz, z)));
        }
    }
}

// ---- vjp: gq (LDS row, dof floats) = J(q)^T gX, using X and the frames the forward left ---
__device__ inline void fk_vjp(fk_cptr fk, const float* sQrow, const float* sXcol, const float* sFcol,
                              const float* sGcol, float* gqRow) {
    const int kind = fk->kind;
    const int dof = fk->dof;
    if (kind == DCX_FK_NONE) {
        for (int i = 0; i < dof; ++i) gqRow[i] = sGcol[i * 64];
        return;
    }
    if (kind == DCX_FK_PLANAR) {
        // gq_i = sum_{j>=i} l_j (-sin phi_j * GX_j + cos phi_j * GY_j),  GX_j = sum_{k>=j} gx_k
        float GX = 0.f, GY = 0.f, acc = 0.f;
        for (int j = dof - 1; j >= 0; --j) {
            GX += sGcol[(2 * j) * 64];
            GY += sGcol[(2 * j + 1) * 64];
            const float c = sFcol[(2 * j) * 64], s = sFcol[(2 * j + 1) * 64];
            acc = fmaf(fk->link_length[j], fmaf(c, GY, -s * GX), acc);
            gqRow[j] = acc;
        }
        return;
    }
    for (int i = 0; i < dof; ++i) gqRow[i] = 0.f;
    if (kind == DCX_FK_DH) {
        const int n_pts = fk->n_points;
        int jbase = 0;
        for (int ch = 0; ch < fk->n_chains; ++ch) {
            const int len = fk->chain_len[ch];
            // suffix sums over the points hanging at or beyond joint i:
            //   A = sum p_k x g_k,  G = sum g_k;   gq_i = z_{i-1} . (A - o_{i-1} x G)
            float A0 = 0.f, A1 = 0.f, A2 = 0.f, G0 = 0.f, G1 = 0.f, G2 = 0.f;
            for (int i = len - 1; i >= 0; --i) {
                for (int k = 0; k < n_pts; ++k) {
                    if (fk->pt_chain[k] != ch || fk->pt_frame[k] != i) continue;
                    const float p0 = sXcol[(3 * k) * 64], p1 = sXcol[(3 * k + 1) * 64], p2 = sXcol[(3 * k + 2) * 64];
                    const float g0 = sGcol[(3 * k) * 64], g1 = sGcol[(3 * k + 1) * 64], g2 = sGcol[(3 * k + 2) * 64];
                    A0 += p1 * g2 - p2 * g1;
                    A1 += p2 * g0 - p0 * g2;
                    A2 += p0 * g1 - p1 * g0;
                    G0 += g0; G1 += g1; G2 += g2;
                }
                const float* f = sFcol + (6 * (jbase + i)) * 64;
                const float z0 = f[0], z1 = f[64], z2 = f[128], o0 = f[192], o1 = f[256], o2 = f[320];
                const float v0 = A0 - (o1 * G2 - o2 * G1);
                const float v1 = A1 - (o2 * G0 - o0 * G2);
                const float v2 = A2 - (o0 * G1 - o1 * G0);
                gqRow[fk->joint_q[ch][i]] += z0 * v0 + z1 * v1 + z2 * v2;
            }
            jbase += len;
        }
    } else if (kind == DCX_FK_SE2) {
        float s, c;
        sincosf(sQrow[2], &s, &c);
        float gx = 0.f, gy = 0.f, gt = 0.f;
        const int n_pts = fk->n_points;
        for (int k = 0; k < n_pts; ++k) {
            const float kx = fk->keypoints[k][0], ky = fk->keypoints[k][1];
            const float a = sGcol[(2 * k) * 64], b = sGcol[(2 * k + 1) * 64];
            gx += a;
            gy += b;
            gt += a * (-s * kx - c * ky) + b * (c * kx - s * ky);
        }
        gqRow[0] = gx; gqRow[1] = gy; gqRow[2] = gt;
    } else if (kind == DCX_FK_SE3) {
        float sx, cx, sy, cy, sz, cz;
        sincosf(sQrow[3], &sx, &cx);
        sincosf(sQrow[4], &sy, &cy);
        sincosf(sQrow[5], &sz, &cz);
        // M = sum_k g_k k_k^T (3x3); d/dangle = <dR/dangle, M>
        float m00 = 0, m01 = 0, m02 = 0, m10 = 0, m11 = 0, m12 = 0, m20 = 0, m21 = 0, m22 = 0;
        float g0s = 0, g1s = 0, g2s = 0;
        const int n_pts = fk->n_points;
        for (int k = 0; k < n_pts; ++k) {
            const float kx = fk->keypoints[k][0], ky = fk->keypoints[k][1], kz = fk->keypoints[k][2];
            const float g0 = sGcol[(3 * k) * 64], g1 = sGcol[(3 * k + 1) * 64], g2 = sGcol[(3 * k + 2) * 64];
            g0s += g0; g1s += g1; g2s += g2;
            m00 += g0 * kx; m01 += g0 * ky; m02 += g0 * kz;
            m10 += g1 * kx; m11 += g1 * ky; m12 += g1 * kz;
            m20 += g2 * kx; m21 += g2 * ky; m22 += g2 * kz;
        }
        // dR/droll  (d/dx of Rx): columns 1,2 change
        const float a01 = cz * sy * cx + sz * sx, a02 = -cz * sy * sx + sz * cx;
        const float a11 = sz * sy * cx - cz * sx, a12 = -sz * sy * sx - cz * cx;
        const float a21 = cy * cx, a22 = -cy * sx;
        // dR/dpitch
        const float b00 = -cz * sy, b01 = cz * cy * sx, b02 = cz * cy * cx;
        const float b10 = -sz * sy, b11 = sz * cy * sx, b12 = sz * cy * cx;
        const float b20 = -cy, b21 = -sy * sx, b22 = -sy * cx;
        // dR/dyaw
        const float c00 = -sz * cy, c01 = -sz * sy * sx - cz * cx, c02 = -sz * sy * cx + cz * sx;
        const float c10 = cz * cy, c11 = cz * sy * sx - sz * cx, c12 = cz * sy * cx + sz * sx;
        gqRow[0] = g0s; gqRow[1] = g1s; gqRow[2] = g2s;
        gqRow[3] = a01 * m01 + a02 * m02 + a11 * m11 + a12 * m12 + a21 * m21 + a22 * m22;
        gqRow[4] = b00 * m00 + b01 * m01 + b02 * m02 + b10 * m10 + b11 * m11 + b12 * m12 + b20 * m20 + b21 * m21 + b22 * m22;
        gqRow[5] = c00 * m00 + c01 * m01 + c02 * m02 + c10 * m10 + c11 * m11 + c12 * m12;
    }
}

}  // namespace dcx
