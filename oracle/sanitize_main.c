/*
 * sanitize_main.c — exercises every oracle entry point on small synthetic inputs so that the C restatement
 * can be run under AddressSanitizer / UndefinedBehaviorSanitizer (`make -C oracle sanitize`).
 * Test infrastructure only.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../include/dcx.h"

void orc_fkine_f32(const dcx_fk_desc*, const float*, int64_t, float*);
void orc_fkine_vjp_f32(const dcx_fk_desc*, const float*, const float*, int64_t, float*);
void orc_kernel_matrix_f32(int, const float*, const float*, int64_t, const float*, int64_t, int, float*);
void orc_score_grad_f32(const dcx_fk_desc*, int, const float*, const float*, const float*, int64_t, int, int, const float*,
                        int64_t, const float*, float*, float*, float*);
void orc_fkine_f64(const dcx_fk_desc*, const double*, int64_t, double*);
void orc_score_grad_f64(const dcx_fk_desc*, int, const double*, const double*, const double*, int64_t, int, int,
                        const double*, int64_t, const double*, double*, double*, double*);

static float frand(unsigned* s) { *s = *s * 1664525u + 1013904223u; return (float)(*s >> 8) / 16777216.0f * 2.0f - 1.0f; }

static void dh_chain(dcx_fk_desc* d) {  /* a Baxter-like 7-joint chain, 4 control points + 1 offset point */
    memset(d, 0, sizeof(*d));
    d->kind = DCX_FK_DH; d->dof = 7; d->n_points = 5; d->point_dim = 3; d->n_chains = 1; d->chain_len[0] = 7;
    const float a[7] = {0.069f, 0, 0.069f, 0, 0.01f, 0, 0}, dd[7] = {0.27f, 0, 0.364f, 0, 0.374f, 0, 0.387f};
    for (int i = 0; i < 7; ++i) {
        d->joint_q[0][i] = i; d->a[0][i] = a[i]; d->d[0][i] = dd[i];
        d->sin_alpha[0][i] = (i == 6) ? 0.f : ((i & 1) ? 1.f : -1.f); d->cos_alpha[0][i] = (i == 6) ? 1.f : 0.f;
    }
    d->base[0][0] = d->base[0][5] = d->base[0][10] = 1.f;
    const int frames[5] = {0, 2, 4, 6, 6};
    for (int k = 0; k < 5; ++k) { d->pt_chain[k] = 0; d->pt_frame[k] = frames[k]; }
    d->pt_off[4][1] = 0.1f;
}

int main(void) {
    unsigned seed = 7;
    enum { B = 37, S = 53, C = 3 };
    dcx_fk_desc descs[5];
    dh_chain(&descs[0]);
    memset(&descs[1], 0, sizeof(dcx_fk_desc)); descs[1].kind = DCX_FK_PLANAR; descs[1].dof = 5; descs[1].n_points = 5; descs[1].point_dim = 2;
    for (int i = 0; i < 5; ++i) descs[1].link_length[i] = 0.3f + 0.1f * i;
    memset(&descs[2], 0, sizeof(dcx_fk_desc)); descs[2].kind = DCX_FK_SE2; descs[2].dof = 3; descs[2].n_points = 4; descs[2].point_dim = 2;
    memset(&descs[3], 0, sizeof(dcx_fk_desc)); descs[3].kind = DCX_FK_SE3; descs[3].dof = 6; descs[3].n_points = 8; descs[3].point_dim = 3;
    for (int k = 0; k < 8; ++k) for (int j = 0; j < 3; ++j) { descs[2].keypoints[k % 4][j % 2] = frand(&seed); descs[3].keypoints[k][j] = frand(&seed); }
    memset(&descs[4], 0, sizeof(dcx_fk_desc)); descs[4].kind = DCX_FK_NONE; descs[4].dof = 6; descs[4].n_points = 6; descs[4].point_dim = 1;
    const float kparams[6][2] = {{10, 2}, {3, 3}, {1, 1}, {3, 2}, {2, 1}, {0.5f, 0}};
    const int kinds[6] = {DCX_K_RQ, DCX_K_RQ, DCX_K_POLY, DCX_K_POLY, DCX_K_POLY, DCX_K_MQ};
    double checksum = 0;
    for (int r = 0; r < 5; ++r) {
        const dcx_fk_desc* fk = &descs[r];
        const int dof = fk->dof, D = fk->n_points * fk->point_dim;
        float* q = malloc(sizeof(float) * B * dof), *sq = malloc(sizeof(float) * S * dof);
        float* X = malloc(sizeof(float) * B * D), *sup = malloc(sizeof(float) * S * D), *W = malloc(sizeof(float) * S * C);
        float* up = malloc(sizeof(float) * B * C), *score = malloc(sizeof(float) * B * C), *grad = malloc(sizeof(float) * B * dof);
        float* jac = malloc(sizeof(float) * B * C * dof), *K = malloc(sizeof(float) * B * S), *gq = malloc(sizeof(float) * B * dof);
        for (int i = 0; i < B * dof; ++i) q[i] = 2.f * frand(&seed);
        for (int i = 0; i < S * dof; ++i) sq[i] = 2.f * frand(&seed);
        for (int i = 0; i < S * C; ++i) W[i] = frand(&seed);
        for (int i = 0; i < B * C; ++i) up[i] = frand(&seed);
        memcpy(sq, q, sizeof(float) * dof);  /* one coincident pair (r = 0) */
        orc_fkine_f32(fk, q, B, X);
        orc_fkine_f32(fk, sq, S, sup);
        orc_fkine_vjp_f32(fk, q, X, B, gq);
        for (int k = 0; k < 6; ++k) {
            orc_kernel_matrix_f32(kinds[k], kparams[k], X, B, sup, S, D, K);
            orc_score_grad_f32(fk, kinds[k], kparams[k], sup, W, S, D, C, q, B, up, score, grad, jac);
            orc_score_grad_f32(fk, kinds[k], kparams[k], sup, W, S, D, C, q, B, NULL, score, grad, NULL);
            for (int i = 0; i < B * C; ++i) checksum += score[i];
            for (int i = 0; i < B * dof; ++i) checksum += grad[i] + gq[i];
            for (int i = 0; i < B * S; ++i) checksum += K[i];
        }
        /* fp64 referee on the same data */
        double* qd = malloc(sizeof(double) * B * dof), *supd = malloc(sizeof(double) * S * D), *Wd = malloc(sizeof(double) * S * C);
        double* sd = malloc(sizeof(double) * B * C), *gd = malloc(sizeof(double) * B * dof);
        for (int i = 0; i < B * dof; ++i) qd[i] = q[i];
        for (int i = 0; i < S * D; ++i) supd[i] = sup[i];
        for (int i = 0; i < S * C; ++i) Wd[i] = W[i];
        const double kp64[2] = {1, 1};
        orc_score_grad_f64(fk, DCX_K_POLY, kp64, supd, Wd, S, D, C, qd, B, NULL, sd, gd, NULL);
        for (int i = 0; i < B * C; ++i) checksum += sd[i];
        free(q); free(sq); free(X); free(sup); free(W); free(up); free(score); free(grad); free(jac); free(K); free(gq);
        free(qd); free(supd); free(Wd); free(sd); free(gd);
    }
    if (!isfinite(checksum)) { printf("non-finite checksum\n"); return 1; }
    printf("oracle sanitize run ok (checksum %.6e)\n", checksum);
    return 0;
}
