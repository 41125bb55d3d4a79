// fk_device.h — forward kinematics and its transpose-Jacobian product on the device.
//
// One lane = one configuration.  Everything that is indexed at run time lives in LDS in
// "column" layout (element e of lane l at base[e*64 + l]: conflict-free, no scratch):
//   q row    : sQ[l*dof + i]     (staged coalesced from HBM by the caller)
//   features : sX[k*64 + l]      (control-point coordinates, D = n_points*point_dim)
//   frames   : sF[e*64 + l]      (DH: sin/cos of every joint angle, then each chain's final rotation
//                                 (9 floats) - what the reverse sweep of fk_vjp needs; planar: cos/sin phi_j)
//
// The public description (dcx_fk_desc, include/dcx.h) is compiled on the host into an FkProg: joints in
// execution order, each carrying its parameters and the [begin, end) range of the control points attached to
// its frame.  The program is staged into LDS once per block and read from there with addresses that depend on
// loop counters only - a chain walk has no dependent "load an index, then load through it" steps and no
// per-point search.  Measured on MI355X (profiles/r01_phase_timing.txt): with the description read field by
// field and a point search per joint, the FK of one wave took 13k cycles and its vjp 14k (~11 us of a 45 us
// small-batch launch).
//
// The sin/cos of all joints are independent of each other, so the block's waves share them (wave w takes
// joints w, w+nw, ...); the chain composition (pure FMA) then runs on wave 0.
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
#include <string.h>
#include <type_traits>
#include "../../include/dcx.h"

namespace dcx {

constexpr int kMaxProgJoints = DCX_MAX_CHAINS * DCX_MAX_JOINTS;
// base transforms the program carries: one per DH chain, one per DISTINCT base of a tree's chains
constexpr int kMaxProgChains = DCX_MAX_TREE_BASES > DCX_MAX_CHAINS ? DCX_MAX_TREE_BASES : DCX_MAX_CHAINS;

struct FkProgJoint {  // DCX_FK_DH, 12 dwords
    int32_t q_index;
    float theta0, a, d, sin_alpha, cos_alpha;
    int32_t pt_begin, pt_end;  // control points attached to this joint's frame: points[pt_begin .. pt_end)
    // a copy of points[pt_begin] (p0_k = its out_k with the kPointBare flag; -1: the frame carries no control point): the
    // scalar-load walks (fk_*_dh_k) get a joint and its usual single point from one record
    int32_t p0_k;
    float p0x, p0y, p0z;
};
// DCX_FK_TREE node, 28 dwords:  T <- T_parent * [F] * Motion.  The root-to-leaf chains of the public description
// are merged back into a tree on the host (a joint repeated on several chains becomes ONE node), nodes are stored in
// depth-first order, and the host has conjugated x/y revolute axes onto z (a signed permutation, exact in fp32), so
// the device knows one rotation: Rz(v).
enum { TJ_FIXED = 0, TJ_REV = 1, TJ_PRISM = 2 };
struct FkProgTreeJoint {
    // integers first (one run of LDS reads feeds every wave-uniform decision of a node)
    int32_t type, q_index;
    int32_t slot, slot_owner;  // frames slot of (sin v, cos v) or (v, -); owner = first node that uses the slot
    int32_t pt_begin, pt_end;
    int32_t start;             // where this node's parent frame comes from: -1 = the previous node (registers),
                               // -2 - b = base[b] (a root), s >= 0 = the frame parked in branch slot s
    int32_t park;              // >= 0: this node's frame is parked in that branch slot (it has further children)
    int32_t leaf;              // >= 0: no node continues from this one in registers; its rotation goes to leaf slot
    int32_t pad0;
    float scale, offset;       // joint variable v = scale * q[q_index] + offset
    float ax, ay, az, pad1;    // prismatic direction
    float F[12];               // row-major 3x4 constant transform applied before the motion
};
struct FkProgPoint {  // 4 dwords
    float ox, oy, oz;
    int32_t out_k;  // feature slot: coordinate r goes to X[out_k + r * out_stride]; DH: | kPointBare
};
// DCX_FK_DH: the control point IS the frame origin (offset 0: every Baxter point, Panda's five frame points).  The walks
// then skip R*o and o x l - the same values (a zero offset adds +-0), 9 + 6 operations fewer per point on the lone wave.
constexpr int32_t kPointBare = 1 << 30;
struct FkProg {
    int32_t kind, dof, n_points, point_dim;
    int32_t n_chains, n_joints, out_stride, n_dwords;  // n_dwords: how much of this struct the kind uses
    int32_t f_leaf, f_park, f_adj, n_branch;           // DCX_FK_TREE: frames offsets (floats per lane), see TreePlan
    int32_t chain_begin[kMaxProgChains], chain_end[kMaxProgChains];  // joint ranges
    float base[kMaxProgChains][12];
    FkProgPoint points[DCX_MAX_POINTS];
    float link_length[DCX_MAX_DOF];
    float keypoints[DCX_MAX_POINTS][4];
    union {  // last member: only the used prefix is staged into LDS
        FkProgJoint joints[kMaxProgJoints];
        FkProgTreeJoint tj[DCX_MAX_TREE_JOINTS];
    };
};
constexpr int kFkProgHeadDwords = (int)(offsetof(FkProg, joints) / 4);

// joints a description executes (DH / TREE), 0 otherwise
__host__ __device__ inline int fk_joint_count(const dcx_fk_desc& fk) {
    int j = 0;
    if (fk.kind == DCX_FK_DH)
        for (int c = 0; c < fk.n_chains; ++c) j += fk.chain_len[c];
    if (fk.kind == DCX_FK_TREE)
        for (int c = 0; c < fk.t_n_chains; ++c) j += fk.t_chain_len[c];
    return j;
}
// LDS floats the staged program of this description occupies
__host__ __device__ inline int fk_prog_floats(const dcx_fk_desc& fk) {
    const int per = fk.kind == DCX_FK_DH ? (int)(sizeof(FkProgJoint) / 4) : fk.kind == DCX_FK_TREE ? (int)(sizeof(FkProgTreeJoint) / 4) : 0;
    return (kFkProgHeadDwords + per * fk_joint_count(fk) + 3) & ~3;
}

// DCX_FK_TREE host planner: merge the chains into a tree, assign frames slots.
//   trig slots  : nodes that read the same joint variable (mimic joints with equal multiplier/offset) share (sin, cos)
//   leaf slots  : 9 floats, the rotation of a node nothing continues from (start of a reverse sweep segment)
//   branch slots: 12 floats, the frame of a node with further children (forward: parked [R|t]; reverse: the adjoint
//                 the later children add up, in a second bank of 12)
// frames layout per lane: [2 * n_slots trig][9 * n_leaves][12 * n_branch parked frames][12 * n_branch adjoint sums]
struct TreePlan {
    int n_nodes, n_slots, n_leaves, n_branch, n_bases;
    int base_chain[DCX_MAX_TREE_BASES];             // a chain that carries distinct base b
    int base_of_chain[DCX_MAX_TREE_CHAINS];
    int parent[DCX_MAX_TREE_JOINTS];                // node index, or -1 - b for a root on base b
    int src[DCX_MAX_TREE_JOINTS];                   // index into the description's t_* arrays
    int node_of[DCX_MAX_TREE_JOINTS];               // description joint (flat index) -> node
    int slot[DCX_MAX_TREE_JOINTS], owner[DCX_MAX_TREE_JOINTS];
    int park[DCX_MAX_TREE_JOINTS], leaf[DCX_MAX_TREE_JOINTS], start[DCX_MAX_TREE_JOINTS];
};
inline bool tree_same_joint(const dcx_fk_desc& fk, int a, int b) {
    return fk.t_type[a] == fk.t_type[b] && fk.t_q[a] == fk.t_q[b] && fk.t_scale[a] == fk.t_scale[b] &&
           fk.t_offset[a] == fk.t_offset[b] && memcmp(fk.t_fixed[a], fk.t_fixed[b], sizeof(fk.t_fixed[a])) == 0 &&
           memcmp(fk.t_axis[a], fk.t_axis[b], sizeof(fk.t_axis[a])) == 0;
}
inline void plan_tree(const dcx_fk_desc& fk, TreePlan& tp) {
    tp.n_nodes = tp.n_slots = tp.n_leaves = tp.n_branch = tp.n_bases = 0;
    for (int c = 0; c < fk.t_n_chains; ++c) {
        int b = -1;
        for (int k = 0; k < tp.n_bases && b < 0; ++k)
            if (memcmp(fk.t_base[c], fk.t_base[tp.base_chain[k]], sizeof(fk.t_base[0])) == 0) b = k;
        if (b < 0 && tp.n_bases < DCX_MAX_TREE_BASES) {
            b = tp.n_bases++;
            tp.base_chain[b] = c;
        }
        tp.base_of_chain[c] = b < 0 ? 0 : b;  // more distinct bases than DCX_MAX_TREE_BASES is rejected by check_fk
    }
    int flat = 0;
    for (int c = 0; c < fk.t_n_chains; ++c) {
        int cur = -1 - tp.base_of_chain[c];  // root on this chain's base
        for (int i = 0; i < fk.t_chain_len[c]; ++i, ++flat) {
            int found = -1;
            for (int n = 0; n < tp.n_nodes && found < 0; ++n) {
                if (tp.parent[n] == cur && tree_same_joint(fk, tp.src[n], flat)) found = n;
            }
            if (found < 0) {
                found = tp.n_nodes++;
                tp.parent[found] = cur;
                tp.src[found] = flat;
            }
            tp.node_of[flat] = found;
            cur = found;
        }
    }
    for (int n = 0; n < tp.n_nodes; ++n) {
        tp.park[n] = tp.leaf[n] = -1;
        // trig slot
        const int a = tp.src[n];
        tp.slot[n] = -1;
        tp.owner[n] = 0;
        if (fk.t_type[a] != DCX_J_FIXED) {
            const bool prism = fk.t_type[a] == DCX_J_PRISMATIC;
            for (int m = 0; m < n && tp.slot[n] < 0; ++m) {
                const int b = tp.src[m];
                if (fk.t_type[b] != DCX_J_FIXED && (fk.t_type[b] == DCX_J_PRISMATIC) == prism && fk.t_q[b] == fk.t_q[a] &&
                    fk.t_scale[b] == fk.t_scale[a] && fk.t_offset[b] == fk.t_offset[a])
                    tp.slot[n] = tp.slot[m];
            }
            if (tp.slot[n] < 0) {
                tp.slot[n] = tp.n_slots++;
                tp.owner[n] = 1;
            }
        }
    }
    for (int n = 0; n < tp.n_nodes; ++n) {
        if (tp.parent[n] < 0) {
            tp.start[n] = -2 - (-1 - tp.parent[n]);  // root: base index b encoded as -2 - b
        } else if (tp.parent[n] == n - 1) {
            tp.start[n] = -1;
        } else {
            const int par = tp.parent[n];
            if (tp.park[par] < 0) tp.park[par] = tp.n_branch++;
            tp.start[n] = tp.park[par];
        }
        if (n + 1 == tp.n_nodes || tp.parent[n + 1] != n) tp.leaf[n] = tp.n_leaves++;
    }
}

// host: DCX_FK_TREE -> program.  Revolute joints about x / y are rewritten as rotations about z:
//   Rx(v) = P Rz(v) P^T,  P = [e_y e_z e_x];   Ry(v) = P' Rz(v) P'^T,  P' = P P = [e_z e_x e_y]
// with the permutation folded into the neighbouring constants (F_j <- P_parent^T F_j P_j, control-point offsets
// o <- P_j^T o).  Permuting entries does not round, so the tree computes exactly the values the x/y forms would.
inline void build_tree_prog(const dcx_fk_desc& fk, FkProg& p) {
    TreePlan tp;
    plan_tree(fk, tp);
    p.n_chains = fk.t_n_chains;
    p.out_stride = fk.t_coord_major ? fk.n_points : 1;
    p.n_joints = tp.n_nodes;
    p.n_dwords = (kFkProgHeadDwords + (int)(sizeof(FkProgTreeJoint) / 4) * tp.n_nodes + 3) & ~3;  // stage the merged nodes only
    p.f_leaf = 2 * tp.n_slots;
    p.f_park = p.f_leaf + 9 * tp.n_leaves;
    p.f_adj = p.f_park + 12 * tp.n_branch;
    p.n_branch = tp.n_branch;
    for (int b = 0; b < tp.n_bases; ++b)
        for (int e = 0; e < 12; ++e) p.base[b][e] = fk.t_base[tp.base_chain[b]][e];
    static const int perm_of[3][3] = {{1, 2, 0}, {2, 0, 1}, {0, 1, 2}};  // column c of P is e_{perm[c]}: REV_X, REV_Y, identity
    auto perm = [&](int node) -> const int* {
        if (node < 0) return perm_of[2];
        const int t = fk.t_type[tp.src[node]];
        return t == DCX_J_REV_X ? perm_of[0] : t == DCX_J_REV_Y ? perm_of[1] : perm_of[2];
    };
    int np = 0, flat_of_chain[DCX_MAX_TREE_CHAINS + 1];
    flat_of_chain[0] = 0;
    for (int c = 0; c < fk.t_n_chains; ++c) flat_of_chain[c + 1] = flat_of_chain[c] + fk.t_chain_len[c];
    for (int n = 0; n < tp.n_nodes; ++n) {
        FkProgTreeJoint& J = p.tj[n];
        const int a = tp.src[n], t = fk.t_type[a];
        const int *prev = perm(tp.parent[n]), *cur = perm(n);
        J.type = (t == DCX_J_FIXED) ? TJ_FIXED : (t == DCX_J_PRISMATIC) ? TJ_PRISM : TJ_REV;
        J.q_index = fk.t_q[a];
        J.scale = fk.t_scale[a];
        J.offset = fk.t_offset[a];
        J.slot = tp.slot[n] < 0 ? 0 : tp.slot[n];
        J.slot_owner = tp.owner[n];
        J.ax = fk.t_axis[a][0];
        J.ay = fk.t_axis[a][1];
        J.az = fk.t_axis[a][2];
        J.start = tp.start[n];
        J.park = tp.park[n];
        J.leaf = tp.leaf[n];
        // (P_prev^T F P_cur): row r of the result is row prev[r] of F; column k of its rotation is column cur[k]
        for (int r = 0; r < 3; ++r) {
            for (int k = 0; k < 3; ++k) J.F[r * 4 + k] = fk.t_fixed[a][prev[r] * 4 + cur[k]];
            J.F[r * 4 + 3] = fk.t_fixed[a][prev[r] * 4 + 3];
        }
        J.pt_begin = np;
        for (int k = 0; k < fk.n_points; ++k) {
            const int f = flat_of_chain[fk.pt_chain[k]] + fk.pt_frame[k];
            if (tp.node_of[f] != n) continue;
            const float* o = fk.pt_off[k];
            p.points[np].ox = o[cur[0]];  // P_cur^T o
            p.points[np].oy = o[cur[1]];
            p.points[np].oz = o[cur[2]];
            p.points[np].out_k = fk.t_coord_major ? k : 3 * k;
            ++np;
        }
        J.pt_end = np;
    }
}

// host: compile the public description into the device program
inline void build_fk_prog(const dcx_fk_desc& fk, FkProg& p) {
    memset(&p, 0, sizeof(p));
    p.kind = fk.kind;
    p.dof = fk.dof;
    p.n_points = fk.n_points;
    p.point_dim = fk.point_dim;
    p.out_stride = 1;
    p.n_dwords = fk_prog_floats(fk);
    for (int i = 0; i < DCX_MAX_DOF; ++i) p.link_length[i] = fk.link_length[i];
    for (int k = 0; k < DCX_MAX_POINTS; ++k)
        for (int j = 0; j < 3; ++j) p.keypoints[k][j] = fk.keypoints[k][j];
    if (fk.kind == DCX_FK_TREE) {
        build_tree_prog(fk, p);
        return;
    }
    if (fk.kind != DCX_FK_DH) return;
    p.n_chains = fk.n_chains;
    int nj = 0, np = 0;
    for (int c = 0; c < fk.n_chains; ++c) {
        p.chain_begin[c] = nj;
        for (int e = 0; e < 12; ++e) p.base[c][e] = fk.base[c][e];
        for (int i = 0; i < fk.chain_len[c]; ++i, ++nj) {
            FkProgJoint& J = p.joints[nj];
            J.q_index = fk.joint_q[c][i];
            J.theta0 = fk.theta0[c][i];
            J.a = fk.a[c][i];
            J.d = fk.d[c][i];
            J.sin_alpha = fk.sin_alpha[c][i];
            J.cos_alpha = fk.cos_alpha[c][i];
            J.pt_begin = np;
            for (int k = 0; k < fk.n_points; ++k) {
                if (fk.pt_chain[k] != c || fk.pt_frame[k] != i) continue;
                p.points[np].ox = fk.pt_off[k][0];
                p.points[np].oy = fk.pt_off[k][1];
                p.points[np].oz = fk.pt_off[k][2];
                const bool bare = fk.pt_off[k][0] == 0.0f && fk.pt_off[k][1] == 0.0f && fk.pt_off[k][2] == 0.0f;
                p.points[np].out_k = 3 * k | (bare ? kPointBare : 0);
                ++np;
            }
            J.pt_end = np;
            J.p0_k = -1;
            J.p0x = J.p0y = J.p0z = 0.0f;
            if (J.pt_end > J.pt_begin) {
                J.p0_k = p.points[J.pt_begin].out_k;
                J.p0x = p.points[J.pt_begin].ox;
                J.p0y = p.points[J.pt_begin].oy;
                J.p0z = p.points[J.pt_begin].oz;
            }
        }
        p.chain_end[c] = nj;
    }
    p.n_joints = nj;
}


// ---- DCX_FK_DH as a STEP TABLE (round 3; the fused kernels' default walk for DH arms, ScoreArgs::fkk == 2) --------------
// The walks above interpret FkProg: per joint a record fetched through the scalar cache (fk_*_dh_k) or LDS -> VGPR ->
// v_readfirstlane (fk_forward_chain / fk_vjp), loop bounds for the joint's points, flags.  On the lone wave that runs the
// chain and J^T of a block every one of those is a dependent round trip: a scalar load and an LDS read share lgkmcnt and
// scalar loads return out of order, so each joint's `s_waitcnt lgkmcnt(0)` waited for the NEXT joint's record as well
// (profiles/r02_phase_fkk.txt: 480-560 cycles per joint for 30 FMAs, 560-920 in the reverse sweep).
// Here the host flattens the robot into at most 32 STEPS of one shape - "compose one DH joint, then place at most one
// control point" - 12 dwords each:
//   * a joint with several control points becomes the joint plus identity steps (theta = a = d = alpha = 0, exact in fp32)
//     that carry the further points, ordered so that the reverse sweep adds the points of a frame in the old order;
//   * everything that steers control flow is a KERNEL ARGUMENT (step counts per chain, bit masks "step has a point",
//     "point sits at the frame origin", "step is a real joint"): SGPRs from the first instruction on, tested with
//     s_bitcmp - no load, no v_readfirstlane;
//   * everything else (DH constants, point offset, feature column, q index) is read from the LDS copy of the table at a
//     wave-uniform address straight into VGPRs, where the VALU wants its operands, one step AHEAD of its use; with no
//     scalar load in flight the LDS counter is in order and the waits are exact.
// Same expressions in the same order as the walks above: bit-identical results (tests/test_gpu_parity.py).
constexpr int kMaxDhSteps = 32;
struct alignas(16) DhStep {  // 12 dwords
    float a, d, sa, ca;
    float ox, oy, oz;
    int32_t meta;     // feature column of the point's x (bits 0-7) | q index (8-15) | feature column of step j-1's point (16-23)
    float theta0;
    int32_t pad[3];
};
struct alignas(16) DhProg {
    int32_t n_chains, n_steps, end0, n_dwords;  // chain 0 = steps [0, end0), chain 1 = [end0, n_steps)
    uint32_t pt_mask, bare_mask, real_mask;
    int32_t n_pt;                               // steps that carry a control point ("point steps", numbered in step order)
    float base[DCX_MAX_CHAINS][12];
    int32_t ps_col[kMaxDhSteps];                // feature column of point step ps (the several-wave J^T, dh2_vjp_waves)
    DhStep steps[kMaxDhSteps];
};
struct DhArgs {  // the part of the table that travels as kernel arguments
    const DhProg* prog;  // device copy; null: this model has no step table (not DCX_FK_DH, or more than kMaxDhSteps steps)
    int32_t n_dwords, n_steps, end0, n_chains;
    uint32_t pt, bare, real;
    int32_t n_pt;
    int32_t shared_q;    // two chains that read a common q entry (their J^T updates of it keep chain order)
};
inline int dh_step_count(const dcx_fk_desc& fk) {
    if (fk.kind != DCX_FK_DH) return 0;
    int steps = 0;
    for (int c = 0; c < fk.n_chains; ++c)
        for (int i = 0; i < fk.chain_len[c]; ++i) {
            int pts = 0;
            for (int k = 0; k < fk.n_points; ++k) pts += (fk.pt_chain[k] == c && fk.pt_frame[k] == i);
            steps += pts > 1 ? pts : 1;
        }
    return steps;
}
// host: description -> step table; false when the robot does not fit (the fused kernels then keep the FkProg walks)
inline bool build_dh_prog(const dcx_fk_desc& fk, DhProg& p) {
    memset(&p, 0, sizeof(p));
    if (fk.kind != DCX_FK_DH || dh_step_count(fk) > kMaxDhSteps || fk.n_points * 3 > 255 || fk.dof > 255) return false;
    p.n_chains = fk.n_chains;
    int ns = 0;
    for (int c = 0; c < fk.n_chains; ++c) {
        for (int e = 0; e < 12; ++e) p.base[c][e] = fk.base[c][e];
        for (int i = 0; i < fk.chain_len[c]; ++i) {
            int pts[DCX_MAX_POINTS], np = 0;
            for (int k = 0; k < fk.n_points; ++k)
                if (fk.pt_chain[k] == c && fk.pt_frame[k] == i) pts[np++] = k;
            // the joint carries the frame's LAST point, identity steps the earlier ones in descending order: the reverse
            // sweep (steps in descending order) then meets them as points[pt_begin], points[pt_begin + 1], ... like fk_vjp
            const int n_here = np > 1 ? np : 1;
            for (int u = 0; u < n_here; ++u, ++ns) {
                DhStep& st = p.steps[ns];
                const bool real = (u == 0);
                st.a = real ? fk.a[c][i] : 0.0f;
                st.d = real ? fk.d[c][i] : 0.0f;
                st.sa = real ? fk.sin_alpha[c][i] : 0.0f;
                st.ca = real ? fk.cos_alpha[c][i] : 1.0f;
                st.theta0 = real ? fk.theta0[c][i] : 0.0f;
                int col = 0;
                if (np > 0) {
                    const int k = pts[np - 1 - u];
                    st.ox = fk.pt_off[k][0];
                    st.oy = fk.pt_off[k][1];
                    st.oz = fk.pt_off[k][2];
                    col = 3 * k;
                    p.ps_col[p.n_pt++] = col;
                    p.pt_mask |= 1u << ns;
                    if (st.ox == 0.0f && st.oy == 0.0f && st.oz == 0.0f) p.bare_mask |= 1u << ns;
                }
                if (real) p.real_mask |= 1u << ns;
                st.meta = col | (fk.joint_q[c][i] << 8);
            }
        }
        if (c == 0) p.end0 = ns;
    }
    p.n_steps = ns;
    if (fk.n_chains == 1) p.end0 = ns;
    for (int j = 1; j < ns; ++j) p.steps[j].meta |= (p.steps[j - 1].meta & 0xff) << 16;
    p.n_dwords = (int)(offsetof(DhProg, steps) / 4) + 12 * ns;
    return true;
}
inline DhArgs dh_args_of(const DhProg& p, const DhProg* dev) {
    DhArgs a;
    a.prog = dev;
    a.n_dwords = p.n_dwords;
    a.n_steps = p.n_steps;
    a.end0 = p.end0;
    a.n_chains = p.n_chains;
    a.pt = p.pt_mask;
    a.bare = p.bare_mask;
    a.real = p.real_mask;
    a.n_pt = p.n_pt;
    a.shared_q = 0;
    for (int i = 0; i < p.end0; ++i)
        for (int j = p.end0; j < p.n_steps; ++j)
            if (((p.real_mask >> i) & 1u) && ((p.real_mask >> j) & 1u) && ((p.steps[i].meta >> 8) & 0xff) == ((p.steps[j].meta >> 8) & 0xff))
                a.shared_q = 1;
    return a;
}

// LDS floats per lane the FK needs for its frames.
inline int fk_frame_floats(const dcx_fk_desc& fk) {
    if (fk.kind == DCX_FK_DH) {
        int j = 0;
        for (int c = 0; c < fk.n_chains; ++c) j += fk.chain_len[c];
        const int steps = dh_step_count(fk);  // the step-table walks keep (sin, cos) per step
        return 2 * (steps > j ? steps : j) + 9 * fk.n_chains;
    }
    if (fk.kind == DCX_FK_PLANAR) return 2 * fk.dof;
    if (fk.kind == DCX_FK_TREE) {
        TreePlan tp;
        plan_tree(fk, tp);
        return 2 * tp.n_slots + 9 * tp.n_leaves + 24 * tp.n_branch;
    }
    return 0;
}

typedef const __attribute__((address_space(3))) FkProg* fk_cptr;

// Developer-only per-joint stamps inside the FK phases (build with -DDCX_TIMING; see score_kernel.h DCX_TS)
#ifdef DCX_TIMING
static __shared__ unsigned long long* dcx_fk_ts;
#define DCX_FK_TS(slot, col)                                                                     \
    do {                                                                                         \
        if (dcx_fk_ts && (threadIdx.x & 63) == 0) dcx_fk_ts[(slot) * 16 + (col)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define DCX_FK_TS(slot, col) do { } while (0)
#endif

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

// all threads of the block copy the program global -> LDS (coalesced); caller synchronises
// n_known > 0: the program's size travels as a kernel argument (no dependent load in front of the copy)
__device__ __forceinline__ fk_cptr stage_fk_prog(const FkProg* g, float* lds, int tid, int nthreads, int n_known = 0) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(g);
    uint32_t* dst = reinterpret_cast<uint32_t*>(lds);
    const int n = n_known > 0 ? n_known : rfl(g->n_dwords);
    for (int i = tid; i < n; i += nthreads) dst[i] = src[i];
#ifdef DCX_TIMING
    if (tid == 0) dcx_fk_ts = nullptr;  // the fused kernel points it at its stamp buffer afterwards
#endif
    return (fk_cptr)(uintptr_t)(uint32_t)(uintptr_t)lds;  // generic -> LDS: the low 32 bits are the LDS offset
}

// sin and cos of one angle: Cody-Waite reduction by pi/2 (three fp32 terms, exact products through fma)
// + the classic degree-7 / degree-8 minimax polynomials on [-pi/4, pi/4].  Max abs error 9e-8 for
// |x| <= 1e4 rad (tools/: validated against float64), i.e. libm-class accuracy, in ~25 VALU instructions and
// a handful of registers.  (ocml's sincosf carries a Payne-Hanek path whose register footprint alone would
// cap the fused kernel at 4 waves per SIMD.)
__device__ __forceinline__ void sincos_f32(float x, float* sn, float* cs) {
    const float k = rintf(x * 0.63661977236758134308f);
    float r = fmaf(k, -1.5707963705062866f, x);
    r = fmaf(k, 4.371138828673793e-08f, r);
    r = fmaf(k, 1.7763568394002505e-15f, r);
    const float r2 = r * r;
    const float sp = fmaf(r * r2, fmaf(r2, fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), r);
    const float cp = fmaf(r2 * r2, fmaf(r2, fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f),
                          fmaf(-0.5f, r2, 1.0f));
    const int q = (int)k & 3;
    const float s0 = (q & 1) ? cp : sp;
    const float c0 = (q & 1) ? sp : cp;
    *sn = (q & 2) ? -s0 : s0;
    *cs = ((q + 1) & 2) ? -c0 : c0;
}

__device__ __forceinline__ void sincos_t(float x, float* sn, float* cs) { sincos_f32(x, sn, cs); }
// ---- scalar types of the FK walks ---------------------------------------------------------------------
// The chain composition and the reverse sweep below are templates over their scalar: `float` in every launch of the
// path, and `Dual` (value + one tangent) in dcx_score_hess, where forward-mode differentiation of the SAME code
// yields the second derivatives of the transform (hess_kernel.hip).  With T = float every expression is the one the
// functions held before they were templates: fma3 is fmaf.
struct Dual {
    float v, d;
    Dual() = default;
    __device__ __forceinline__ Dual(float a) : v(a), d(0.0f) {}
    __device__ __forceinline__ Dual(float a, float b) : v(a), d(b) {}
};
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return Dual(a.v + b.v, a.d + b.d); }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return Dual(a.v - b.v, a.d - b.d); }
__device__ __forceinline__ Dual operator-(Dual a) { return Dual(-a.v, -a.d); }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { return Dual(a.v * b.v, fmaf(a.v, b.d, a.d * b.v)); }
__device__ __forceinline__ Dual operator*(Dual a, float b) { return Dual(a.v * b, a.d * b); }
__device__ __forceinline__ Dual operator*(float a, Dual b) { return Dual(a * b.v, a * b.d); }
__device__ __forceinline__ Dual& operator+=(Dual& a, Dual b) { a.v += b.v; a.d += b.d; return a; }
__device__ __forceinline__ float dual_v(float a) { return a; }
__device__ __forceinline__ float dual_v(Dual a) { return a.v; }
__device__ __forceinline__ float fma3(float a, float b, float c) { return fmaf(a, b, c); }
template <class A, class B, class C>
__device__ __forceinline__ Dual fma3(A a, B b, C c) {
    float d = 0.0f;
    bool first = true;
    if constexpr (__is_same(C, Dual)) { d = c.d; first = false; }
    if constexpr (__is_same(A, Dual)) { d = first ? a.d * dual_v(b) : fmaf(a.d, dual_v(b), d); first = false; }
    if constexpr (__is_same(B, Dual)) { d = first ? dual_v(a) * b.d : fmaf(dual_v(a), b.d, d); }
    return Dual(fmaf(dual_v(a), dual_v(b), dual_v(c)), d);
}

__device__ __forceinline__ void sincos_t(Dual x, Dual* sn, Dual* cs) {
    float s0, c0;
    sincos_f32(x.v, &s0, &c0);
    *sn = Dual(s0, c0 * x.d);
    *cs = Dual(c0, -s0 * x.d);
}

// ---- DCX_FK_TREE chain composition and reverse sweep ---------------------------------------------------
// Deliberately NOT inlined: inside the fused kernel their register demand would otherwise be planned into every
// robot's prologue / epilogue (the DH headline kernel went from 12 to 108 bytes of scratch per lane, doubling its
// HBM writes, when these bodies were inlined).
template <class T>
__device__ __attribute__((noinline)) void fk_tree_chain(fk_cptr fk, T* sXcol, T* sFcol) {
    // T_j = T_parent(j) * F_j * Motion_j over the tree in depth-first order (reference: RigidBody.forward_kinematics,
    // collision_interfaces/rigid_body.py:82-140); features = frame origins (+ constant offsets for links behind fixed
    // joints), collision_checkers.py:386-393
    const int stride = rfl(fk->out_stride) * 64, njt = rfl(fk->n_joints);
    const int f_leaf = rfl(fk->f_leaf), f_park = rfl(fk->f_park);
    T r00 = 1.f, r01 = 0.f, r02 = 0.f, r10 = 0.f, r11 = 1.f, r12 = 0.f, r20 = 0.f, r21 = 0.f, r22 = 1.f;
    T t0 = 0.f, t1 = 0.f, t2 = 0.f;
    for (int j = 0; j < njt; ++j) {
        const int start = rfl(fk->tj[j].start);
        if (start <= -2) {  // a root: the frame is base[b]
            const auto* Bm = fk->base[-2 - start];
            r00 = Bm[0]; r01 = Bm[1]; r02 = Bm[2]; t0 = Bm[3];
            r10 = Bm[4]; r11 = Bm[5]; r12 = Bm[6]; t1 = Bm[7];
            r20 = Bm[8]; r21 = Bm[9]; r22 = Bm[10]; t2 = Bm[11];
        } else if (start >= 0) {  // a later child of a branch node: its parent's frame was parked
            const T* pk = sFcol + (f_park + 12 * start) * 64;
            r00 = pk[0]; r01 = pk[64]; r02 = pk[128]; t0 = pk[192];
            r10 = pk[256]; r11 = pk[320]; r12 = pk[384]; t1 = pk[448];
            r20 = pk[512]; r21 = pk[576]; r22 = pk[640]; t2 = pk[704];
        }
        const auto* F = fk->tj[j].F;
        // N = T * F
        const float f00 = F[0], f01 = F[1], f02 = F[2], f03 = F[3], f10 = F[4], f11 = F[5], f12 = F[6], f13 = F[7];
        const float f20 = F[8], f21 = F[9], f22 = F[10], f23 = F[11];
        t0 = fma3(r00, f03, fma3(r01, f13, fma3(r02, f23, t0)));
        t1 = fma3(r10, f03, fma3(r11, f13, fma3(r12, f23, t1)));
        t2 = fma3(r20, f03, fma3(r21, f13, fma3(r22, f23, t2)));
        T n00 = fma3(r00, f00, fma3(r01, f10, r02 * f20)), n01 = fma3(r00, f01, fma3(r01, f11, r02 * f21)),
              n02 = fma3(r00, f02, fma3(r01, f12, r02 * f22));
        T n10 = fma3(r10, f00, fma3(r11, f10, r12 * f20)), n11 = fma3(r10, f01, fma3(r11, f11, r12 * f21)),
              n12 = fma3(r10, f02, fma3(r11, f12, r12 * f22));
        T n20 = fma3(r20, f00, fma3(r21, f10, r22 * f20)), n21 = fma3(r20, f01, fma3(r21, f11, r22 * f21)),
              n22 = fma3(r20, f02, fma3(r21, f12, r22 * f22));
        const int type = rfl(fk->tj[j].type), slot = rfl(fk->tj[j].slot);
        if (type == TJ_REV) {
            // R <- N * Rz(v): columns 0, 1 rotate
            const T s = sFcol[(2 * slot) * 64], c = sFcol[(2 * slot + 1) * 64];
            r00 = fma3(n00, c, n01 * s); r01 = fma3(n01, c, -n00 * s); r02 = n02;
            r10 = fma3(n10, c, n11 * s); r11 = fma3(n11, c, -n10 * s); r12 = n12;
            r20 = fma3(n20, c, n21 * s); r21 = fma3(n21, c, -n20 * s); r22 = n22;
        } else {
            r00 = n00; r01 = n01; r02 = n02; r10 = n10; r11 = n11; r12 = n12; r20 = n20; r21 = n21; r22 = n22;
            if (type == TJ_PRISM) {
                const T v = sFcol[(2 * slot) * 64];
                const T dx = fk->tj[j].ax * v, dy = fk->tj[j].ay * v, dz = fk->tj[j].az * v;
                t0 = fma3(r00, dx, fma3(r01, dy, fma3(r02, dz, t0)));
                t1 = fma3(r10, dx, fma3(r11, dy, fma3(r12, dz, t1)));
                t2 = fma3(r20, dx, fma3(r21, dy, fma3(r22, dz, t2)));
            }
        }
        const int pb = rfl(fk->tj[j].pt_begin), pe = rfl(fk->tj[j].pt_end);
        for (int p = pb; p < pe; ++p) {
            const float ox = fk->points[p].ox, oy = fk->points[p].oy, oz = fk->points[p].oz;
            T* out = sXcol + rfl(fk->points[p].out_k) * 64;
            out[0] = fma3(r00, ox, fma3(r01, oy, fma3(r02, oz, t0)));
            out[stride] = fma3(r10, ox, fma3(r11, oy, fma3(r12, oz, t1)));
            out[2 * stride] = fma3(r20, ox, fma3(r21, oy, fma3(r22, oz, t2)));
        }
        const int park = rfl(fk->tj[j].park), leaf = rfl(fk->tj[j].leaf);
        if (park >= 0) {  // further children start from this frame
            T* pk = sFcol + (f_park + 12 * park) * 64;
            pk[0] = r00; pk[64] = r01; pk[128] = r02; pk[192] = t0;
            pk[256] = r10; pk[320] = r11; pk[384] = r12; pk[448] = t1;
            pk[512] = r20; pk[576] = r21; pk[640] = r22; pk[704] = t2;
        }
        if (leaf >= 0) {  // nothing continues from here in registers: the reverse sweep restarts from this rotation
            T* fr = sFcol + (f_leaf + 9 * leaf) * 64;
            fr[0] = r00; fr[64] = r01; fr[128] = r02; fr[192] = r10; fr[256] = r11; fr[320] = r12;
            fr[384] = r20; fr[448] = r21; fr[512] = r22;
        }
    }
}

#ifndef DCX_VJP_MATRIX_ADJOINT
template <class T>
__device__ __attribute__((noinline)) void fk_tree_vjp(fk_cptr fk, const T* sFcol, const T* sGcol, T* gqRow) {
    const int dof = rfl(fk->dof);
    // Reverse sweep of a wrench through T_j = T_parent(j) F_j M_j(v_j) over the tree (the DH branch of fk_vjp explains the
    // form): force f and moment n about the origin of the current node's frame, in that frame's coordinates; joints
    // driven by the same q (mimic) simply accumulate.  With N = T_parent F_j:
    //   points of node j (offset o, upstream g):  l = R_j^T g ;  f += l ;  n += o x l
    //   revolute : T_j = N Rz(v)         ->  dL/dv = n.z ;  f <- Rz f, n <- Rz n ;  R_N = R_j Rz^T
    //   prismatic: t_j = t_N + R_N a v   ->  dL/dv = f . a ;  n <- n + (a v) x f
    //   through F: f_parent = F_R f ;  n_parent = F_R n + F_t x f_parent ;  R_parent = R_N F_R^T
    // A branch node's later children leave their wrench (6 floats, in the branch node's coordinates) in its adjoint slot.
    for (int i = 0; i < dof; ++i) gqRow[i] = 0.f;  // the sweep reads frames, not q
    const int stride = rfl(fk->out_stride) * 64, njt = rfl(fk->n_joints);
    const int f_leaf = rfl(fk->f_leaf), f_adj = rfl(fk->f_adj), n_branch = rfl(fk->n_branch);
    T* sFw = const_cast<T*>(sFcol);  // the adjoint sums of branch nodes live in the frames area (12 floats apart, 6 used)
    for (int b = 0; b < n_branch; ++b)
        for (int e = 0; e < 6; ++e) sFw[(f_adj + 12 * b + e) * 64] = 0.f;
    T r00 = 1.f, r01 = 0.f, r02 = 0.f, r10 = 0.f, r11 = 1.f, r12 = 0.f, r20 = 0.f, r21 = 0.f, r22 = 1.f;
    T f0 = 0.f, f1 = 0.f, f2 = 0.f, n0 = 0.f, n1 = 0.f, n2 = 0.f;
    // nodes in reverse depth-first order: every child has been processed before its parent
    for (int j = njt - 1; j >= 0; --j) {
        const int leaf = rfl(fk->tj[j].leaf), park = rfl(fk->tj[j].park);
        if (leaf >= 0) {  // no child handed its state over in registers: restart from this node's own rotation
            const T* fr = sFcol + (f_leaf + 9 * leaf) * 64;
            r00 = fr[0]; r01 = fr[64]; r02 = fr[128]; r10 = fr[192]; r11 = fr[256]; r12 = fr[320];
            r20 = fr[384]; r21 = fr[448]; r22 = fr[512];
            f0 = f1 = f2 = 0.f;
            n0 = n1 = n2 = 0.f;
        }
        if (park >= 0) {  // plus what the children that started from the parked frame sent back
            const T* ad = sFcol + (f_adj + 12 * park) * 64;
            f0 += ad[0]; f1 += ad[64]; f2 += ad[128];
            n0 += ad[192]; n1 += ad[256]; n2 += ad[320];
        }
        const int pb = rfl(fk->tj[j].pt_begin), pe = rfl(fk->tj[j].pt_end);
        for (int p = pb; p < pe; ++p) {
            const T* gin = sGcol + rfl(fk->points[p].out_k) * 64;
            const T g0 = gin[0], g1 = gin[stride], g2 = gin[2 * stride];
            const float ox = fk->points[p].ox, oy = fk->points[p].oy, oz = fk->points[p].oz;
            const T l0 = fma3(r00, g0, fma3(r10, g1, r20 * g2));
            const T l1 = fma3(r01, g0, fma3(r11, g1, r21 * g2));
            const T l2 = fma3(r02, g0, fma3(r12, g1, r22 * g2));
            f0 += l0; f1 += l1; f2 += l2;
            n0 = fma3(oy, l2, fma3(-oz, l1, n0));
            n1 = fma3(oz, l0, fma3(-ox, l2, n1));
            n2 = fma3(ox, l1, fma3(-oy, l0, n2));
        }
        const int type = rfl(fk->tj[j].type), slot = rfl(fk->tj[j].slot);
        if (type == TJ_REV) {
            const T s = sFcol[(2 * slot) * 64], c = sFcol[(2 * slot + 1) * 64];
            gqRow[rfl(fk->tj[j].q_index)] += fk->tj[j].scale * n2;
            // into N's coordinates: Rz(v) on f and n; R_N = R_j Rz^T (columns 0, 1 rotate back)
            const T g0 = fma3(c, f0, -(s * f1)), g1 = fma3(s, f0, c * f1);
            const T m0 = fma3(c, n0, -(s * n1)), m1 = fma3(s, n0, c * n1);
            f0 = g0; f1 = g1; n0 = m0; n1 = m1;
            const T p00 = fma3(r00, c, -r01 * s), p01 = fma3(r01, c, r00 * s);
            const T p10 = fma3(r10, c, -r11 * s), p11 = fma3(r11, c, r10 * s);
            const T p20 = fma3(r20, c, -r21 * s), p21 = fma3(r21, c, r20 * s);
            r00 = p00; r01 = p01; r10 = p10; r11 = p11; r20 = p20; r21 = p21;
        } else if (type == TJ_PRISM) {
            const T v = sFcol[(2 * slot) * 64];
            const float ax = fk->tj[j].ax, ay = fk->tj[j].ay, az = fk->tj[j].az;
            gqRow[rfl(fk->tj[j].q_index)] += fk->tj[j].scale * fma3(ax, f0, fma3(ay, f1, az * f2));
            const T dx = ax * v, dy = ay * v, dz = az * v;
            n0 = fma3(dy, f2, fma3(-dz, f1, n0));
            n1 = fma3(dz, f0, fma3(-dx, f2, n1));
            n2 = fma3(dx, f1, fma3(-dy, f0, n2));
        }
        // back through the constant transform F
        const auto* F = fk->tj[j].F;
        const float f00 = F[0], f01 = F[1], f02 = F[2], f03 = F[3], f10 = F[4], f11 = F[5], f12 = F[6], f13 = F[7];
        const float f20 = F[8], f21 = F[9], f22 = F[10], f23 = F[11];
        const T h0 = fma3(f00, f0, fma3(f01, f1, f02 * f2)), h1 = fma3(f10, f0, fma3(f11, f1, f12 * f2)),
                    h2 = fma3(f20, f0, fma3(f21, f1, f22 * f2));
        const T k0 = fma3(f00, n0, fma3(f01, n1, f02 * n2)), k1 = fma3(f10, n0, fma3(f11, n1, f12 * n2)),
                    k2 = fma3(f20, n0, fma3(f21, n1, f22 * n2));
        f0 = h0; f1 = h1; f2 = h2;
        n0 = fma3(f13, h2, fma3(-f23, h1, k0));
        n1 = fma3(f23, h0, fma3(-f03, h2, k1));
        n2 = fma3(f03, h1, fma3(-f13, h0, k2));
        const T q00 = fma3(r00, f00, fma3(r01, f01, r02 * f02)), q01 = fma3(r00, f10, fma3(r01, f11, r02 * f12)),
                    q02 = fma3(r00, f20, fma3(r01, f21, r02 * f22));
        const T q10 = fma3(r10, f00, fma3(r11, f01, r12 * f02)), q11 = fma3(r10, f10, fma3(r11, f11, r12 * f12)),
                    q12 = fma3(r10, f20, fma3(r11, f21, r12 * f22));
        const T q20 = fma3(r20, f00, fma3(r21, f01, r22 * f02)), q21 = fma3(r20, f10, fma3(r21, f11, r22 * f12)),
                    q22 = fma3(r20, f20, fma3(r21, f21, r22 * f22));
        r00 = q00; r01 = q01; r02 = q02; r10 = q10; r11 = q11; r12 = q12; r20 = q20; r21 = q21; r22 = q22;
        // (R, f, n) now refer to the parent frame.  A child that started from a parked frame adds its wrench to the
        // parent's sum; a child that continued in registers just carries on; a root's parent wrench is dropped.
        const int start = rfl(fk->tj[j].start);
        if (start >= 0) {
            T* ad = sFw + (f_adj + 12 * start) * 64;
            ad[0] += f0; ad[64] += f1; ad[128] += f2;
            ad[192] += n0; ad[256] += n1; ad[320] += n2;
        }
    }
}
#else
template <class T>
__device__ __attribute__((noinline)) void fk_tree_vjp(fk_cptr fk, const T* sFcol, const T* sGcol, T* gqRow) {
    const int dof = rfl(fk->dof);
    // Reverse-mode sweep through T_j = T_parent(j) F_j M_j(v_j) over the tree; joints driven by the same q (mimic)
    // simply accumulate.  With N = T_parent F_j:
    //   revolute : R_j = R_N Rz(v), t_j = t_N   ->  dL/dv = <R_N^T GR, dRz/dv>,  G_RN = GR Rz^T
    //   prismatic: R_j = R_N, t_j = t_N + R_N a v -> dL/dv = Gt . (R_N a),        G_RN = GR + Gt (a v)^T
    //   through F: G_R(j-1) = G_RN F_R^T + Gt F_t^T,  R_(j-1) = R_N F_R^T,  Gt unchanged
    for (int i = 0; i < dof; ++i) gqRow[i] = 0.f;  // the sweep reads frames, not q
    const int stride = rfl(fk->out_stride) * 64, njt = rfl(fk->n_joints);
    const int f_leaf = rfl(fk->f_leaf), f_adj = rfl(fk->f_adj), n_branch = rfl(fk->n_branch);
    T* sFw = const_cast<T*>(sFcol);  // the adjoint sums of branch nodes live in the frames area
    for (int e = 0; e < 12 * n_branch; ++e) sFw[(f_adj + e) * 64] = 0.f;
    T r00 = 1.f, r01 = 0.f, r02 = 0.f, r10 = 0.f, r11 = 1.f, r12 = 0.f, r20 = 0.f, r21 = 0.f, r22 = 1.f;
    T G00 = 0.f, G01 = 0.f, G02 = 0.f, G10 = 0.f, G11 = 0.f, G12 = 0.f, G20 = 0.f, G21 = 0.f, G22 = 0.f;
    T T0 = 0.f, T1 = 0.f, T2 = 0.f;
    // nodes in reverse depth-first order: every child has been processed before its parent
    for (int j = njt - 1; j >= 0; --j) {
        const int leaf = rfl(fk->tj[j].leaf), park = rfl(fk->tj[j].park);
        if (leaf >= 0) {  // no child handed its state over in registers: restart from this node's own rotation
            const T* fr = sFcol + (f_leaf + 9 * leaf) * 64;
            r00 = fr[0]; r01 = fr[64]; r02 = fr[128]; r10 = fr[192]; r11 = fr[256]; r12 = fr[320];
            r20 = fr[384]; r21 = fr[448]; r22 = fr[512];
            G00 = G01 = G02 = G10 = G11 = G12 = G20 = G21 = G22 = 0.f;
            T0 = T1 = T2 = 0.f;
        }
        if (park >= 0) {  // plus what the children that started from the parked frame sent back
            const T* ad = sFcol + (f_adj + 12 * park) * 64;
            G00 += ad[0]; G01 += ad[64]; G02 += ad[128]; T0 += ad[192];
            G10 += ad[256]; G11 += ad[320]; G12 += ad[384]; T1 += ad[448];
            G20 += ad[512]; G21 += ad[576]; G22 += ad[640]; T2 += ad[704];
        }
        const int pb = rfl(fk->tj[j].pt_begin), pe = rfl(fk->tj[j].pt_end);
        for (int p = pb; p < pe; ++p) {
            const T* gin = sGcol + rfl(fk->points[p].out_k) * 64;
            const T g0 = gin[0], g1 = gin[stride], g2 = gin[2 * stride];
            const float ox = fk->points[p].ox, oy = fk->points[p].oy, oz = fk->points[p].oz;
            T0 += g0; T1 += g1; T2 += g2;
            G00 = fma3(g0, ox, G00); G01 = fma3(g0, oy, G01); G02 = fma3(g0, oz, G02);
            G10 = fma3(g1, ox, G10); G11 = fma3(g1, oy, G11); G12 = fma3(g1, oz, G12);
            G20 = fma3(g2, ox, G20); G21 = fma3(g2, oy, G21); G22 = fma3(g2, oz, G22);
        }
        const int type = rfl(fk->tj[j].type), slot = rfl(fk->tj[j].slot);
        if (type == TJ_REV) {
            const T s = sFcol[(2 * slot) * 64], c = sFcol[(2 * slot + 1) * 64];
            // R_N = R_j Rz^T: columns 0, 1 rotate back
            const T p00 = fma3(r00, c, -r01 * s), p01 = fma3(r01, c, r00 * s);
            const T p10 = fma3(r10, c, -r11 * s), p11 = fma3(r11, c, r10 * s);
            const T p20 = fma3(r20, c, -r21 * s), p21 = fma3(r21, c, r20 * s);
            // A = R_N^T GR, rows 0 and 1, columns 0 and 1 (dRz/dv = [[-s, -c, 0], [c, -s, 0], [0, 0, 0]])
            const T A00 = fma3(p00, G00, fma3(p10, G10, p20 * G20)), A01 = fma3(p00, G01, fma3(p10, G11, p20 * G21));
            const T A10 = fma3(p01, G00, fma3(p11, G10, p21 * G20)), A11 = fma3(p01, G01, fma3(p11, G11, p21 * G21));
            const T dv = (c * A10 - s * A11) - (s * A00 + c * A01);
            gqRow[rfl(fk->tj[j].q_index)] += fk->tj[j].scale * dv;
            // G_RN = GR Rz^T
            const T h00 = fma3(G00, c, -G01 * s), h01 = fma3(G01, c, G00 * s);
            const T h10 = fma3(G10, c, -G11 * s), h11 = fma3(G11, c, G10 * s);
            const T h20 = fma3(G20, c, -G21 * s), h21 = fma3(G21, c, G20 * s);
            G00 = h00; G01 = h01; G10 = h10; G11 = h11; G20 = h20; G21 = h21;
            r00 = p00; r01 = p01; r10 = p10; r11 = p11; r20 = p20; r21 = p21;
        } else if (type == TJ_PRISM) {
            const T v = sFcol[(2 * slot) * 64];
            const float ax = fk->tj[j].ax, ay = fk->tj[j].ay, az = fk->tj[j].az;
            const T w0 = fma3(r00, ax, fma3(r01, ay, r02 * az)), w1 = fma3(r10, ax, fma3(r11, ay, r12 * az)),
                        w2 = fma3(r20, ax, fma3(r21, ay, r22 * az));
            gqRow[rfl(fk->tj[j].q_index)] += fk->tj[j].scale * fma3(T0, w0, fma3(T1, w1, T2 * w2));
            const T dx = ax * v, dy = ay * v, dz = az * v;
            G00 = fma3(T0, dx, G00); G01 = fma3(T0, dy, G01); G02 = fma3(T0, dz, G02);
            G10 = fma3(T1, dx, G10); G11 = fma3(T1, dy, G11); G12 = fma3(T1, dz, G12);
            G20 = fma3(T2, dx, G20); G21 = fma3(T2, dy, G21); G22 = fma3(T2, dz, G22);
        }
        // back through the constant transform F
        const auto* F = fk->tj[j].F;
        const float f00 = F[0], f01 = F[1], f02 = F[2], f03 = F[3], f10 = F[4], f11 = F[5], f12 = F[6], f13 = F[7];
        const float f20 = F[8], f21 = F[9], f22 = F[10], f23 = F[11];
        const T n00 = fma3(G00, f00, fma3(G01, f01, fma3(G02, f02, T0 * f03)));
        const T n01 = fma3(G00, f10, fma3(G01, f11, fma3(G02, f12, T0 * f13)));
        const T n02 = fma3(G00, f20, fma3(G01, f21, fma3(G02, f22, T0 * f23)));
        const T n10 = fma3(G10, f00, fma3(G11, f01, fma3(G12, f02, T1 * f03)));
        const T n11 = fma3(G10, f10, fma3(G11, f11, fma3(G12, f12, T1 * f13)));
        const T n12 = fma3(G10, f20, fma3(G11, f21, fma3(G12, f22, T1 * f23)));
        const T n20 = fma3(G20, f00, fma3(G21, f01, fma3(G22, f02, T2 * f03)));
        const T n21 = fma3(G20, f10, fma3(G21, f11, fma3(G22, f12, T2 * f13)));
        const T n22 = fma3(G20, f20, fma3(G21, f21, fma3(G22, f22, T2 * f23)));
        G00 = n00; G01 = n01; G02 = n02; G10 = n10; G11 = n11; G12 = n12; G20 = n20; G21 = n21; G22 = n22;
        const T q00 = fma3(r00, f00, fma3(r01, f01, r02 * f02)), q01 = fma3(r00, f10, fma3(r01, f11, r02 * f12)),
                    q02 = fma3(r00, f20, fma3(r01, f21, r02 * f22));
        const T q10 = fma3(r10, f00, fma3(r11, f01, r12 * f02)), q11 = fma3(r10, f10, fma3(r11, f11, r12 * f12)),
                    q12 = fma3(r10, f20, fma3(r11, f21, r12 * f22));
        const T q20 = fma3(r20, f00, fma3(r21, f01, r22 * f02)), q21 = fma3(r20, f10, fma3(r21, f11, r22 * f12)),
                    q22 = fma3(r20, f20, fma3(r21, f21, r22 * f22));
        r00 = q00; r01 = q01; r02 = q02; r10 = q10; r11 = q11; r12 = q12; r20 = q20; r21 = q21; r22 = q22;
        // (R, GR, Gt) now refer to the parent frame.  A child that started from a parked frame adds its adjoint to the
        // parent's sum; a child that continued in registers just carries on; a root's parent adjoint is dropped.
        const int start = rfl(fk->tj[j].start);
        if (start >= 0) {
            T* ad = sFw + (f_adj + 12 * start) * 64;
            ad[0] += G00; ad[64] += G01; ad[128] += G02; ad[192] += T0;
            ad[256] += G10; ad[320] += G11; ad[384] += G12; ad[448] += T1;
            ad[512] += G20; ad[576] += G21; ad[640] += G22; ad[704] += T2;
        }
    }
}

#endif

// ---- DCX_FK_DH walks that read the program with SCALAR loads -------------------------------------------------------------
// fk_forward_chain / fk_vjp interpret the program from its LDS copy: every wave-uniform field arrives in a VGPR and goes
// through v_readfirstlane before it can steer a branch or an address, and a lone wave pays ~10 cycles for each of those
// instructions (profiles/r02_vjp_ab.txt: 680-960 cycles per joint for ~45 FMAs).  Here the same program is read from global
// memory through the constant address space: one s_load per joint record lands in SGPRs, ready to be VALU operands, loop
// bounds and LDS offsets; the record of the NEXT joint is requested before the current one is composed, so the scalar
// cache / L2 latency hides behind a joint's arithmetic.  The arithmetic is the LDS walks', expression for expression:
// results are bit-identical (tests/test_gpu_parity.py).
typedef const __attribute__((address_space(4))) FkProg* fk_kptr;
struct FkJointRec {
    int32_t q_index, pt_begin, pt_end, p0_k;
    float a, d, sa, ca, p0x, p0y, p0z;
};
__device__ __forceinline__ FkJointRec fk_load_joint(fk_kptr gk, int j) {
    FkJointRec r;
    r.q_index = gk->joints[j].q_index;
    r.a = gk->joints[j].a;
    r.d = gk->joints[j].d;
    r.sa = gk->joints[j].sin_alpha;
    r.ca = gk->joints[j].cos_alpha;
    r.pt_begin = gk->joints[j].pt_begin;
    r.pt_end = gk->joints[j].pt_end;
    r.p0_k = gk->joints[j].p0_k;
    r.p0x = gk->joints[j].p0x;
    r.p0y = gk->joints[j].p0y;
    r.p0z = gk->joints[j].p0z;
    return r;
}

template <class T>
__device__ inline void fk_forward_chain_dh_k(fk_kptr gk, T* sXcol, T* sFcol) {
    const int nch = gk->n_chains, njt = gk->n_joints;
    for (int ch = 0; ch < nch; ++ch) {
        T r00 = gk->base[ch][0], r01 = gk->base[ch][1], r02 = gk->base[ch][2], t0 = gk->base[ch][3];
        T r10 = gk->base[ch][4], r11 = gk->base[ch][5], r12 = gk->base[ch][6], t1 = gk->base[ch][7];
        T r20 = gk->base[ch][8], r21 = gk->base[ch][9], r22 = gk->base[ch][10], t2 = gk->base[ch][11];
        const int jb = gk->chain_begin[ch], je = gk->chain_end[ch];
        FkJointRec nx = fk_load_joint(gk, jb);
        for (int j = jb; j < je; ++j) {
            const FkJointRec cu = nx;
            nx = fk_load_joint(gk, j + 1 < je ? j + 1 : j);
            const T s = sFcol[(2 * j) * 64], c = sFcol[(2 * j + 1) * 64];
            const float a = cu.a, d = cu.d, sa = cu.sa, ca = cu.ca;
            const T n00 = fma3(r00, c, r01 * s), n10 = fma3(r10, c, r11 * s), n20 = fma3(r20, c, r21 * s);
            const T u0 = fma3(r01, c, -(r00 * s)), u1 = fma3(r11, c, -(r10 * s)), u2 = fma3(r21, c, -(r20 * s));
            t0 = fma3(n00, a, fma3(r02, d, t0));
            t1 = fma3(n10, a, fma3(r12, d, t1));
            t2 = fma3(n20, a, fma3(r22, d, t2));
            const T n01 = fma3(u0, ca, r02 * sa), n11 = fma3(u1, ca, r12 * sa), n21 = fma3(u2, ca, r22 * sa);
            const T n02 = fma3(r02, ca, -(u0 * sa)), n12 = fma3(r12, ca, -(u1 * sa)), n22 = fma3(r22, ca, -(u2 * sa));
            r00 = n00; r01 = n01; r02 = n02; r10 = n10; r11 = n11; r12 = n12; r20 = n20; r21 = n21; r22 = n22;
            if (cu.p0_k >= 0) {
                T* out = sXcol + (cu.p0_k & (kPointBare - 1)) * 64;
                if (cu.p0_k & kPointBare) {
                    out[0] = t0; out[64] = t1; out[128] = t2;
                } else {
                    out[0] = fma3(r00, cu.p0x, fma3(r01, cu.p0y, fma3(r02, cu.p0z, t0)));
                    out[64] = fma3(r10, cu.p0x, fma3(r11, cu.p0y, fma3(r12, cu.p0z, t1)));
                    out[128] = fma3(r20, cu.p0x, fma3(r21, cu.p0y, fma3(r22, cu.p0z, t2)));
                }
                for (int p = cu.pt_begin + 1; p < cu.pt_end; ++p) {  // further points of this frame (Panda's fingers)
                    const int ok = gk->points[p].out_k;
                    const float ox = gk->points[p].ox, oy = gk->points[p].oy, oz = gk->points[p].oz;
                    T* o2 = sXcol + (ok & (kPointBare - 1)) * 64;
                    o2[0] = fma3(r00, ox, fma3(r01, oy, fma3(r02, oz, t0)));
                    o2[64] = fma3(r10, ox, fma3(r11, oy, fma3(r12, oz, t1)));
                    o2[128] = fma3(r20, ox, fma3(r21, oy, fma3(r22, oz, t2)));
                }
            }
            DCX_FK_TS(7 + (j < 8 ? j : 8), 0);
        }
        T* fr = sFcol + (2 * njt + 9 * ch) * 64;  // final rotation of this chain (for the reverse sweep)
        fr[0] = r00; fr[64] = r01; fr[128] = r02; fr[192] = r10; fr[256] = r11; fr[320] = r12;
        fr[384] = r20; fr[448] = r21; fr[512] = r22;
    }
}

template <class T>
__device__ inline void fk_vjp_dh_k(fk_kptr gk, const T* sFcol, const T* sGcol, T* gqRow) {
    const int dof = gk->dof;
    for (int i = 0; i < dof; ++i) gqRow[i] = 0.f;
    const int nch = gk->n_chains, njt = gk->n_joints;
    for (int ch = 0; ch < nch; ++ch) {
        const int jb = gk->chain_begin[ch], je = gk->chain_end[ch];
        FkJointRec nx = fk_load_joint(gk, je - 1);
        const T* fr = sFcol + (2 * njt + 9 * ch) * 64;
        T r00 = fr[0], r01 = fr[64], r02 = fr[128], r10 = fr[192], r11 = fr[256], r12 = fr[320];
        T r20 = fr[384], r21 = fr[448], r22 = fr[512];
        T f0 = 0.f, f1 = 0.f, f2 = 0.f, n0 = 0.f, n1 = 0.f, n2 = 0.f;
        for (int j = je - 1; j >= jb; --j) {
            const FkJointRec cu = nx;
            nx = fk_load_joint(gk, j > jb ? j - 1 : j);
            // same order as fk_vjp: points[pt_begin], then the further points of the frame
            for (int p = cu.pt_begin; p < cu.pt_end; ++p) {
                const bool first = (p == cu.pt_begin);
                const int ok = first ? cu.p0_k : gk->points[p].out_k;
                const T* gin = sGcol + (ok & (kPointBare - 1)) * 64;
                const T g0 = gin[0], g1 = gin[64], g2 = gin[128];
                const T l0 = fma3(r00, g0, fma3(r10, g1, r20 * g2));
                const T l1 = fma3(r01, g0, fma3(r11, g1, r21 * g2));
                const T l2 = fma3(r02, g0, fma3(r12, g1, r22 * g2));
                f0 += l0; f1 += l1; f2 += l2;
                if (ok & kPointBare) continue;
                const float ox = first ? cu.p0x : gk->points[p].ox, oy = first ? cu.p0y : gk->points[p].oy,
                            oz = first ? cu.p0z : gk->points[p].oz;
                n0 = fma3(oy, l2, fma3(-oz, l1, n0));
                n1 = fma3(oz, l0, fma3(-ox, l2, n1));
                n2 = fma3(ox, l1, fma3(-oy, l0, n2));
            }
            const T s = sFcol[(2 * j) * 64], c = sFcol[(2 * j + 1) * 64];
            const float a = cu.a, d = cu.d, sa = cu.sa, ca = cu.ca;
            const T fy = fma3(ca, f1, -(sa * f2)), fz = fma3(sa, f1, ca * f2);
            const T mx = fma3(-d, fy, n0);
            const T my = fma3(d, f0, fma3(-a, fz, fma3(ca, n1, -(sa * n2))));
            const T mz = fma3(a, fy, fma3(sa, n1, ca * n2));
            gqRow[cu.q_index] += mz;
            if (j > jb) {
                f1 = fma3(s, f0, c * fy);
                f0 = fma3(c, f0, -(s * fy));
                f2 = fz;
                n0 = fma3(c, mx, -(s * my));
                n1 = fma3(s, mx, c * my);
                n2 = mz;
                const T u0 = fma3(ca, r01, -(sa * r02)), u1 = fma3(ca, r11, -(sa * r12)), u2 = fma3(ca, r21, -(sa * r22));
                r02 = fma3(sa, r01, ca * r02); r12 = fma3(sa, r11, ca * r12); r22 = fma3(sa, r21, ca * r22);
                r01 = fma3(s, r00, c * u0); r11 = fma3(s, r10, c * u1); r21 = fma3(s, r20, c * u2);
                r00 = fma3(c, r00, -(s * u0)); r10 = fma3(c, r10, -(s * u1)); r20 = fma3(c, r20, -(s * u2));
            }
            DCX_FK_TS(7 + (j < 8 ? j : 8), 1);
        }
    }
}

// ---- forward, phase A (every wave of the block): sin/cos of the joint angles -> frames ---------------
// wave w of nw takes joints w, w+nw, ...   Caller synchronises the block afterwards.
template <class T>
__device__ inline void fk_forward_trig(fk_cptr fk, const T* sQrow, T* sFcol, int wave, int nw) {
    const int kind = rfl(fk->kind);
    if (kind == DCX_FK_DH) {
        const int nj = rfl(fk->n_joints);
        for (int j = wave; j < nj; j += nw) {
            const T th = sQrow[rfl(fk->joints[j].q_index)] + fk->joints[j].theta0;
            T s, c;
            sincos_t(th, &s, &c);
            sFcol[(2 * j) * 64] = s;
            sFcol[(2 * j + 1) * 64] = c;
        }
    } else if (kind == DCX_FK_TREE) {
        const int nj = rfl(fk->n_joints);
        for (int j = wave; j < nj; j += nw) {
            const int type = rfl(fk->tj[j].type);
            if (type == TJ_FIXED || !rfl(fk->tj[j].slot_owner)) continue;
            const T v = fma3(fk->tj[j].scale, sQrow[rfl(fk->tj[j].q_index)], fk->tj[j].offset);
            const int slot = rfl(fk->tj[j].slot);
            if (type == TJ_REV) {
                T s, c;
                sincos_t(v, &s, &c);
                sFcol[(2 * slot) * 64] = s;
                sFcol[(2 * slot + 1) * 64] = c;
            } else {
                sFcol[(2 * slot) * 64] = v;  // prismatic: the reverse sweep needs v after q has been overwritten
            }
        }
    } else if (kind == DCX_FK_PLANAR) {
        const int dof = rfl(fk->dof);
        for (int j = wave; j < dof; j += nw) {
            T phi = 0.f;
            for (int i = 0; i <= j; ++i) phi += sQrow[i];  // same left-to-right sum as cumsum
            T s, c;
            sincos_t(phi, &s, &c);
            sFcol[(2 * j) * 64] = c;
            sFcol[(2 * j + 1) * 64] = s;
        }
    }
}

// ---- forward, phase B (one wave): compose the chain -> X (LDS column) --------------------------------
template <class T>
__device__ inline void fk_forward_chain(fk_cptr fk, const T* sQrow, T* sXcol, T* sFcol) {
    const int kind = rfl(fk->kind);
    if (kind == DCX_FK_NONE) {
        const int dof = rfl(fk->dof);
        for (int i = 0; i < dof; ++i) sXcol[i * 64] = sQrow[i];
    } else if (kind == DCX_FK_PLANAR) {
        const int dof = rfl(fk->dof);
        T x = 0.f, y = 0.f;
        for (int i = 0; i < dof; ++i) {
            const float l = fk->link_length[i];
            x = fma3(l, sFcol[(2 * i) * 64], x);
            y = fma3(l, sFcol[(2 * i + 1) * 64], y);
            sXcol[(2 * i) * 64] = x;
            sXcol[(2 * i + 1) * 64] = y;
        }
    } else if (kind == DCX_FK_DH) {
        const int nch = rfl(fk->n_chains), njt = rfl(fk->n_joints);
        for (int ch = 0; ch < nch; ++ch) {
            T r00 = fk->base[ch][0], r01 = fk->base[ch][1], r02 = fk->base[ch][2], t0 = fk->base[ch][3];
            T r10 = fk->base[ch][4], r11 = fk->base[ch][5], r12 = fk->base[ch][6], t1 = fk->base[ch][7];
            T r20 = fk->base[ch][8], r21 = fk->base[ch][9], r22 = fk->base[ch][10], t2 = fk->base[ch][11];
            const int jb = rfl(fk->chain_begin[ch]), je = rfl(fk->chain_end[ch]);
            for (int j = jb; j < je; ++j) {
                const T s = sFcol[(2 * j) * 64], c = sFcol[(2 * j + 1) * 64];
                const float a = fk->joints[j].a, d = fk->joints[j].d;
                const float sa = fk->joints[j].sin_alpha, ca = fk->joints[j].cos_alpha;
#ifndef DCX_FK_DH_MATRIX
                // T <- T * Rz(theta) * Trans(a, 0, d) * Rx(alpha)   (utils.DH2mat, factor by factor: columns 0, 1 of R mix by
                // theta, the origin moves by a along the new column 0 and by d along column 2, columns 1, 2 mix by alpha):
                // 30 operations per joint where the product with the assembled 3x4 block takes 39 - the chain is a lone
                // wave's phase, bound by its instruction count.  -DDCX_FK_DH_MATRIX restores the block form.
                const T n00 = fma3(r00, c, r01 * s), n10 = fma3(r10, c, r11 * s), n20 = fma3(r20, c, r21 * s);
                const T u0 = fma3(r01, c, -(r00 * s)), u1 = fma3(r11, c, -(r10 * s)), u2 = fma3(r21, c, -(r20 * s));
                t0 = fma3(n00, a, fma3(r02, d, t0));
                t1 = fma3(n10, a, fma3(r12, d, t1));
                t2 = fma3(n20, a, fma3(r22, d, t2));
                const T n01 = fma3(u0, ca, r02 * sa), n11 = fma3(u1, ca, r12 * sa), n21 = fma3(u2, ca, r22 * sa);
                const T n02 = fma3(r02, ca, -(u0 * sa)), n12 = fma3(r12, ca, -(u1 * sa)), n22 = fma3(r22, ca, -(u2 * sa));
#else
                // T <- T * [[c, -s ca,  s sa, a c], [s, c ca, -c sa, a s], [0, sa, ca, d]]   (utils.DH2mat)
                const T m01 = -s * ca, m02 = s * sa, m11 = c * ca, m12 = -c * sa;
                const T ac = a * c, as = a * s;
                t0 = fma3(r00, ac, fma3(r01, as, fma3(r02, d, t0)));
                t1 = fma3(r10, ac, fma3(r11, as, fma3(r12, d, t1)));
                t2 = fma3(r20, ac, fma3(r21, as, fma3(r22, d, t2)));
                const T n00 = fma3(r00, c, r01 * s), n10 = fma3(r10, c, r11 * s), n20 = fma3(r20, c, r21 * s);
                const T n01 = fma3(r00, m01, fma3(r01, m11, r02 * sa));
                const T n11 = fma3(r10, m01, fma3(r11, m11, r12 * sa));
                const T n21 = fma3(r20, m01, fma3(r21, m11, r22 * sa));
                const T n02 = fma3(r00, m02, fma3(r01, m12, r02 * ca));
                const T n12 = fma3(r10, m02, fma3(r11, m12, r12 * ca));
                const T n22 = fma3(r20, m02, fma3(r21, m12, r22 * ca));
#endif
                r00 = n00; r01 = n01; r02 = n02; r10 = n10; r11 = n11; r12 = n12; r20 = n20; r21 = n21; r22 = n22;
                const int pb = rfl(fk->joints[j].pt_begin), pe = rfl(fk->joints[j].pt_end);
                for (int p = pb; p < pe; ++p) {
                    const int ok = rfl(fk->points[p].out_k);
                    T* out = sXcol + (ok & (kPointBare - 1)) * 64;
                    if (ok & kPointBare) {
                        out[0] = t0; out[64] = t1; out[128] = t2;
                        continue;
                    }
                    const float ox = fk->points[p].ox, oy = fk->points[p].oy, oz = fk->points[p].oz;
                    out[0] = fma3(r00, ox, fma3(r01, oy, fma3(r02, oz, t0)));
                    out[64] = fma3(r10, ox, fma3(r11, oy, fma3(r12, oz, t1)));
                    out[128] = fma3(r20, ox, fma3(r21, oy, fma3(r22, oz, t2)));
                }
                DCX_FK_TS(7 + (j < 8 ? j : 8), 0);
            }
            T* fr = sFcol + (2 * njt + 9 * ch) * 64;  // final rotation of this chain (for the reverse sweep)
            fr[0] = r00; fr[64] = r01; fr[128] = r02; fr[192] = r10; fr[256] = r11; fr[320] = r12;
            fr[384] = r20; fr[448] = r21; fr[512] = r22;
        }
    } else if (kind == DCX_FK_TREE) {
        fk_tree_chain(fk, sXcol, sFcol);
    } else if (kind == DCX_FK_SE2) {
        const T x = sQrow[0], y = sQrow[1];
        T s, c;
        sincos_t(sQrow[2], &s, &c);
        const int n_pts = rfl(fk->n_points);
        for (int k = 0; k < n_pts; ++k) {
            const float kx = fk->keypoints[k][0], ky = fk->keypoints[k][1];
            sXcol[(2 * k) * 64] = fma3(c, kx, fma3(-s, ky, x));
            sXcol[(2 * k + 1) * 64] = fma3(s, kx, fma3(c, ky, y));
        }
    } else if (kind == DCX_FK_SE3) {
        T sx, cx, sy, cy, sz, cz;
        sincos_t(sQrow[3], &sx, &cx);
        sincos_t(sQrow[4], &sy, &cy);
        sincos_t(sQrow[5], &sz, &cz);
        // R = Rz(yaw) Ry(pitch) Rx(roll)
        const T r00 = cz * cy, r01 = cz * sy * sx - sz * cx, r02 = cz * sy * cx + sz * sx;
        const T r10 = sz * cy, r11 = sz * sy * sx + cz * cx, r12 = sz * sy * cx - cz * sx;
        const T r20 = -sy, r21 = cy * sx, r22 = cy * cx;
        const T x = sQrow[0], y = sQrow[1], z = sQrow[2];
        const int n_pts = rfl(fk->n_points);
        for (int k = 0; k < n_pts; ++k) {
            const float kx = fk->keypoints[k][0], ky = fk->keypoints[k][1], kz = fk->keypoints[k][2];
            sXcol[(3 * k) * 64] = fma3(r00, kx, fma3(r01, ky, fma3(r02, kz, x)));
            sXcol[(3 * k + 1) * 64] = fma3(r10, kx, fma3(r11, ky, fma3(r12, kz, y)));
            sXcol[(3 * k + 2) * 64] = fma3(r20, kx, fma3(r21, ky, fma3(r22, kz, z)));
        }
    }
}

// ---- vjp (one wave): gq (LDS row, dof floats) = J(q)^T gX, using the frames the forward left ------------
// NOTE: gqRow may alias sQrow (callers build the gradient row in place of the q row), so every branch must
// finish reading sQrow before its first write to gqRow.
template <class T>
__device__ inline void fk_vjp(fk_cptr fk, const T* sQrow, const T* sFcol, const T* sGcol, T* gqRow) {
    const int kind = rfl(fk->kind);
    const int dof = rfl(fk->dof);
    if (kind == DCX_FK_NONE) {
        for (int i = 0; i < dof; ++i) gqRow[i] = sGcol[i * 64];
        return;
    }
    if (kind == DCX_FK_PLANAR) {
        // gq_i = sum_{j>=i} l_j (-sin phi_j * GX_j + cos phi_j * GY_j),  GX_j = sum_{k>=j} gx_k
        T GX = 0.f, GY = 0.f, acc = 0.f;
        for (int j = dof - 1; j >= 0; --j) {
            GX += sGcol[(2 * j) * 64];
            GY += sGcol[(2 * j + 1) * 64];
            const T c = sFcol[(2 * j) * 64], s = sFcol[(2 * j + 1) * 64];
            acc = fma3(fk->link_length[j], fma3(c, GY, -s * GX), acc);
            gqRow[j] = acc;
        }
        return;
    }
    if (kind == DCX_FK_DH) {
        for (int i = 0; i < dof; ++i) gqRow[i] = 0.f;  // DH reads frames, not q
        const int nch = rfl(fk->n_chains), njt = rfl(fk->n_joints);
#ifndef DCX_VJP_MATRIX_ADJOINT
        // Reverse sweep of a WRENCH (force f, moment n about the frame origin) expressed in the coordinates of the current
        // frame, from the tip of the chain to its base.  With A_j = Rz(theta_j) Trans(a, 0, d) Rx(alpha):
        //   points of frame j (offset o, upstream g):  l = R_j^T g ;  f += l ;  n += o x l
        //   through Rx(alpha) and the translation:     f1 = Rx f ;  n1 = Rx n + (a, 0, d) x f1
        //   dL/dtheta_j = n1.z                         (the moment about frame j-1's z axis; Rz(theta) leaves z alone)
        //   through Rz(theta):                         f <- Rz f1 ;  n <- Rz n1 ;  R_{j-1} = R_j Rx^T Rz^T (recomputed)
        // Six adjoint components instead of the twelve (GR 3x3, Gt) of the matrix chain rule, and the joint derivative
        // costs nothing: ~44 instead of ~100 operations per joint (the lone wave's J^T phase of a block).
        // Like the matrix form - and unlike the world-frame z x (p - o), which differences accumulated positions - the
        // STRUCTURAL zeros come out exactly: they live in the DH constants (Baxter's last joint, a = 0 with the control
        // point on the joint axis: n stays 0 and n1.z = sa*0 + ca*0 + 0*f1.y = 0), so an optimiser such as Adam is not
        // handed 1e-8 of round-off to normalise into full-size steps.  -DDCX_VJP_MATRIX_ADJOINT restores the matrix form.
        for (int ch = 0; ch < nch; ++ch) {
            const T* fr = sFcol + (2 * njt + 9 * ch) * 64;
            T r00 = fr[0], r01 = fr[64], r02 = fr[128], r10 = fr[192], r11 = fr[256], r12 = fr[320];
            T r20 = fr[384], r21 = fr[448], r22 = fr[512];
            T f0 = 0.f, f1 = 0.f, f2 = 0.f, n0 = 0.f, n1 = 0.f, n2 = 0.f;
            const int jb = rfl(fk->chain_begin[ch]), je = rfl(fk->chain_end[ch]);
            for (int j = je - 1; j >= jb; --j) {
                const int pb = rfl(fk->joints[j].pt_begin), pe = rfl(fk->joints[j].pt_end);
                for (int p = pb; p < pe; ++p) {
                    const int ok = rfl(fk->points[p].out_k);
                    const T* gin = sGcol + (ok & (kPointBare - 1)) * 64;
                    const T g0 = gin[0], g1 = gin[64], g2 = gin[128];
                    const T l0 = fma3(r00, g0, fma3(r10, g1, r20 * g2));
                    const T l1 = fma3(r01, g0, fma3(r11, g1, r21 * g2));
                    const T l2 = fma3(r02, g0, fma3(r12, g1, r22 * g2));
                    f0 += l0; f1 += l1; f2 += l2;
                    if (ok & kPointBare) continue;  // o = 0: no moment about the frame origin
                    const float ox = fk->points[p].ox, oy = fk->points[p].oy, oz = fk->points[p].oz;
                    n0 = fma3(oy, l2, fma3(-oz, l1, n0));
                    n1 = fma3(oz, l0, fma3(-ox, l2, n1));
                    n2 = fma3(ox, l1, fma3(-oy, l0, n2));
                }
                const T s = sFcol[(2 * j) * 64], c = sFcol[(2 * j + 1) * 64];
                const float a = fk->joints[j].a, d = fk->joints[j].d;
                const float sa = fk->joints[j].sin_alpha, ca = fk->joints[j].cos_alpha;
                // Rx(alpha), then the moment arm (a, 0, d)
                const T fy = fma3(ca, f1, -(sa * f2)), fz = fma3(sa, f1, ca * f2);
                const T mx = fma3(-d, fy, n0);
                const T my = fma3(d, f0, fma3(-a, fz, fma3(ca, n1, -(sa * n2))));
                const T mz = fma3(a, fy, fma3(sa, n1, ca * n2));
                gqRow[rfl(fk->joints[j].q_index)] += mz;
                if (j > jb) {
                    // Rz(theta)
                    f1 = fma3(s, f0, c * fy);
                    f0 = fma3(c, f0, -(s * fy));
                    f2 = fz;
                    n0 = fma3(c, mx, -(s * my));
                    n1 = fma3(s, mx, c * my);
                    n2 = mz;
                    // R_{j-1} = R_j Rx(alpha)^T Rz(theta)^T : columns 1, 2 mix by alpha, then columns 0, 1 by theta
                    const T u0 = fma3(ca, r01, -(sa * r02)), u1 = fma3(ca, r11, -(sa * r12)), u2 = fma3(ca, r21, -(sa * r22));
                    r02 = fma3(sa, r01, ca * r02); r12 = fma3(sa, r11, ca * r12); r22 = fma3(sa, r21, ca * r22);
                    r01 = fma3(s, r00, c * u0); r11 = fma3(s, r10, c * u1); r21 = fma3(s, r20, c * u2);
                    r00 = fma3(c, r00, -(s * u0)); r10 = fma3(c, r10, -(s * u1)); r20 = fma3(c, r20, -(s * u2));
                }
                DCX_FK_TS(7 + (j < 8 ? j : 8), 1);
            }
        }
#else
        // Reverse-mode sweep through T_j = T_{j-1} A_j(theta_j), exactly the chain rule autograd applies to
        // the reference's bmm chain.  Unlike the geometric form z x (p - o) it reproduces STRUCTURAL zeros
        // exactly (e.g. Baxter's last joint, a = 0 and the control point on the joint axis): an optimiser
        // such as Adam would otherwise amplify 1e-8 round-off on such a joint into full-size steps.
        //   GR, Gt : adjoints of the current frame's rotation / translation
        //   dL/dtheta_j = <R_{j-1}^T GR, dA_R/dtheta> + <R_{j-1}^T Gt, da_t/dtheta>
        //   GR <- GR A_R^T + Gt a_t^T ;  Gt unchanged ;  R_{j-1} = R_j A_R^T (recomputed, not stored)
        for (int ch = 0; ch < nch; ++ch) {
            const T* fr = sFcol + (2 * njt + 9 * ch) * 64;
            T r00 = fr[0], r01 = fr[64], r02 = fr[128], r10 = fr[192], r11 = fr[256], r12 = fr[320];
            T r20 = fr[384], r21 = fr[448], r22 = fr[512];
            T G00 = 0.f, G01 = 0.f, G02 = 0.f, G10 = 0.f, G11 = 0.f, G12 = 0.f, G20 = 0.f, G21 = 0.f, G22 = 0.f;
            T T0 = 0.f, T1 = 0.f, T2 = 0.f;
            const int jb = rfl(fk->chain_begin[ch]), je = rfl(fk->chain_end[ch]);
            for (int j = je - 1; j >= jb; --j) {
                const int pb = rfl(fk->joints[j].pt_begin), pe = rfl(fk->joints[j].pt_end);
                for (int p = pb; p < pe; ++p) {
                    const T* gin = sGcol + (rfl(fk->points[p].out_k) & (kPointBare - 1)) * 64;
                    const T g0 = gin[0], g1 = gin[64], g2 = gin[128];
                    const float ox = fk->points[p].ox, oy = fk->points[p].oy, oz = fk->points[p].oz;
                    T0 += g0; T1 += g1; T2 += g2;
                    G00 = fma3(g0, ox, G00); G01 = fma3(g0, oy, G01); G02 = fma3(g0, oz, G02);
                    G10 = fma3(g1, ox, G10); G11 = fma3(g1, oy, G11); G12 = fma3(g1, oz, G12);
                    G20 = fma3(g2, ox, G20); G21 = fma3(g2, oy, G21); G22 = fma3(g2, oz, G22);
                }
                const T s = sFcol[(2 * j) * 64], c = sFcol[(2 * j + 1) * 64];
                const float a = fk->joints[j].a, d = fk->joints[j].d;
                const float sa = fk->joints[j].sin_alpha, ca = fk->joints[j].cos_alpha;
                // A_R = [[c, -s ca, s sa], [s, c ca, -c sa], [0, sa, ca]],  a_t = (a c, a s, d)
                const T a00 = c, a01 = -s * ca, a02 = s * sa, a10 = s, a11 = c * ca, a12 = -c * sa;
                const T at0 = a * c, at1 = a * s;
                // R_{i-1} = R_i A_R^T
                const T p00 = fma3(r00, a00, fma3(r01, a01, r02 * a02)), p01 = fma3(r00, a10, fma3(r01, a11, r02 * a12)),
                            p02 = fma3(r01, sa, r02 * ca);
                const T p10 = fma3(r10, a00, fma3(r11, a01, r12 * a02)), p11 = fma3(r10, a10, fma3(r11, a11, r12 * a12)),
                            p12 = fma3(r11, sa, r12 * ca);
                const T p20 = fma3(r20, a00, fma3(r21, a01, r22 * a02)), p21 = fma3(r20, a10, fma3(r21, a11, r22 * a12)),
                            p22 = fma3(r21, sa, r22 * ca);
                // M = R_{i-1}^T GR (rows 0,1 only: dA_R/dtheta has a zero third row), u = R_{i-1}^T Gt
                const T M00 = fma3(p00, G00, fma3(p10, G10, p20 * G20)), M01 = fma3(p00, G01, fma3(p10, G11, p20 * G21)),
                            M02 = fma3(p00, G02, fma3(p10, G12, p20 * G22));
                const T M10 = fma3(p01, G00, fma3(p11, G10, p21 * G20)), M11 = fma3(p01, G01, fma3(p11, G11, p21 * G21)),
                            M12 = fma3(p01, G02, fma3(p11, G12, p21 * G22));
                const T u0 = fma3(p00, T0, fma3(p10, T1, p20 * T2)), u1 = fma3(p01, T0, fma3(p11, T1, p21 * T2));
                // dA_R/dtheta = [[-s, -c ca, c sa], [c, -s ca, s sa], [0,0,0]] = [[-a10, -a11, -a12], [a00, a01, a02], 0]
                // da_t/dtheta = (-a s, a c, 0)
                const T dth = (M10 * a00 + M11 * a01 + M12 * a02) - (M00 * a10 + M01 * a11 + M02 * a12)
                                  + (u1 * at0 - u0 * at1);
                gqRow[rfl(fk->joints[j].q_index)] += dth;
                // GR <- GR A_R^T + Gt a_t^T
                const T n00 = fma3(G00, a00, fma3(G01, a01, fma3(G02, a02, T0 * at0)));
                const T n01 = fma3(G00, a10, fma3(G01, a11, fma3(G02, a12, T0 * at1)));
                const T n02 = fma3(G01, sa, fma3(G02, ca, T0 * d));
                const T n10 = fma3(G10, a00, fma3(G11, a01, fma3(G12, a02, T1 * at0)));
                const T n11 = fma3(G10, a10, fma3(G11, a11, fma3(G12, a12, T1 * at1)));
                const T n12 = fma3(G11, sa, fma3(G12, ca, T1 * d));
                const T n20 = fma3(G20, a00, fma3(G21, a01, fma3(G22, a02, T2 * at0)));
                const T n21 = fma3(G20, a10, fma3(G21, a11, fma3(G22, a12, T2 * at1)));
                const T n22 = fma3(G21, sa, fma3(G22, ca, T2 * d));
                G00 = n00; G01 = n01; G02 = n02; G10 = n10; G11 = n11; G12 = n12; G20 = n20; G21 = n21; G22 = n22;
                r00 = p00; r01 = p01; r02 = p02; r10 = p10; r11 = p11; r12 = p12; r20 = p20; r21 = p21; r22 = p22;
                DCX_FK_TS(7 + (j < 8 ? j : 8), 1);
            }
        }
#endif
    } else if (kind == DCX_FK_TREE) {
        fk_tree_vjp(fk, sFcol, sGcol, gqRow);
    } else if (kind == DCX_FK_SE2) {
        T s, c;
        sincos_t(sQrow[2], &s, &c);
        T gx = 0.f, gy = 0.f, gt = 0.f;
        const int n_pts = rfl(fk->n_points);
        for (int k = 0; k < n_pts; ++k) {
            const float kx = fk->keypoints[k][0], ky = fk->keypoints[k][1];
            const T a = sGcol[(2 * k) * 64], b = sGcol[(2 * k + 1) * 64];
            gx += a;
            gy += b;
            gt += a * (-s * kx - c * ky) + b * (c * kx - s * ky);
        }
        gqRow[0] = gx; gqRow[1] = gy; gqRow[2] = gt;
    } else if (kind == DCX_FK_SE3) {
        T sx, cx, sy, cy, sz, cz;
        sincos_t(sQrow[3], &sx, &cx);
        sincos_t(sQrow[4], &sy, &cy);
        sincos_t(sQrow[5], &sz, &cz);
        // M = sum_k g_k k_k^T (3x3); d/dangle = <dR/dangle, M>
        T m00 = 0, m01 = 0, m02 = 0, m10 = 0, m11 = 0, m12 = 0, m20 = 0, m21 = 0, m22 = 0;
        T g0s = 0, g1s = 0, g2s = 0;
        const int n_pts = rfl(fk->n_points);
        for (int k = 0; k < n_pts; ++k) {
            const float kx = fk->keypoints[k][0], ky = fk->keypoints[k][1], kz = fk->keypoints[k][2];
            const T g0 = sGcol[(3 * k) * 64], g1 = sGcol[(3 * k + 1) * 64], g2 = sGcol[(3 * k + 2) * 64];
            g0s += g0; g1s += g1; g2s += g2;
            m00 += g0 * kx; m01 += g0 * ky; m02 += g0 * kz;
            m10 += g1 * kx; m11 += g1 * ky; m12 += g1 * kz;
            m20 += g2 * kx; m21 += g2 * ky; m22 += g2 * kz;
        }
        // dR/droll  (d/dx of Rx): columns 1,2 change
        const T a01 = cz * sy * cx + sz * sx, a02 = -cz * sy * sx + sz * cx;
        const T a11 = sz * sy * cx - cz * sx, a12 = -sz * sy * sx - cz * cx;
        const T a21 = cy * cx, a22 = -cy * sx;
        // dR/dpitch
        const T b00 = -cz * sy, b01 = cz * cy * sx, b02 = cz * cy * cx;
        const T b10 = -sz * sy, b11 = sz * cy * sx, b12 = sz * cy * cx;
        const T b20 = -cy, b21 = -sy * sx, b22 = -sy * cx;
        // dR/dyaw
        const T c00 = -sz * cy, c01 = -sz * sy * sx - cz * cx, c02 = -sz * sy * cx + cz * sx;
        const T c10 = cz * cy, c11 = cz * sy * sx - sz * cx, c12 = cz * sy * cx + sz * sx;
        gqRow[0] = g0s; gqRow[1] = g1s; gqRow[2] = g2s;
        gqRow[3] = a01 * m01 + a02 * m02 + a11 * m11 + a12 * m12 + a21 * m21 + a22 * m22;
        gqRow[4] = b00 * m00 + b01 * m01 + b02 * m02 + b10 * m10 + b11 * m11 + b12 * m12 + b20 * m20 + b21 * m21 + b22 * m22;
        gqRow[5] = c00 * m00 + c01 * m01 + c02 * m02 + c10 * m10 + c11 * m11 + c12 * m12;
    }
}

// ---- the step-table walks (device) ------------------------------------------------------------------------------------
typedef const __attribute__((address_space(3))) DhProg* dh_cptr;
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) f4v* lds_f4;

// all threads of the block copy the used part of the table global -> LDS, 16 bytes per thread; caller synchronises
__device__ __forceinline__ dh_cptr stage_dh_prog(const DhArgs& c, float* lds, int tid, int nthreads) {
    const f4v* src = reinterpret_cast<const f4v*>(c.prog);
    f4v* dst = reinterpret_cast<f4v*>(lds);
    const int n = c.n_dwords >> 2;
    for (int i = tid; i < n; i += nthreads) dst[i] = src[i];
    return (dh_cptr)(uintptr_t)(uint32_t)(uintptr_t)lds;
}
__device__ __forceinline__ bool dh_bit(uint32_t mask, int j) { return (mask >> j) & 1u; }

// sin / cos per step (every wave of the block: wave w takes steps w, w + nw, ...); identity steps get (0, 1)
__device__ inline void dh2_trig(dh_cptr p, const DhArgs& c, const float* sQrow, float* sFcol, int wave, int nw) {
    for (int j = wave; j < c.n_steps; j += nw) {
        float s = 0.0f, co = 1.0f;
        if (dh_bit(c.real, j)) {
            const int qi = (p->steps[j].meta >> 8) & 0xff;
            sincos_f32(sQrow[qi] + p->steps[j].theta0, &s, &co);
        }
        sFcol[(2 * j) * 64] = s;
        sFcol[(2 * j + 1) * 64] = co;
    }
}

// chain composition (one wave): T <- T * Rz(theta) * Trans(a, 0, d) * Rx(alpha) per step, the step's control point out
__device__ inline void dh2_chain(dh_cptr p, const DhArgs& c, float* sXcol, float* sFcol) {
    int jb = 0;
    for (int ch = 0; ch < c.n_chains; ++ch) {
        const int je = ch == 0 ? c.end0 : c.n_steps;
        const f4v b0 = *(lds_f4)&p->base[ch][0], b1 = *(lds_f4)&p->base[ch][4], b2 = *(lds_f4)&p->base[ch][8];
        float r00 = b0.x, r01 = b0.y, r02 = b0.z, t0 = b0.w;
        float r10 = b1.x, r11 = b1.y, r12 = b1.z, t1 = b1.w;
        float r20 = b2.x, r21 = b2.y, r22 = b2.z, t2 = b2.w;
        f4v nk = *(lds_f4)&p->steps[jb].a, no = *(lds_f4)&p->steps[jb].ox;
        float ns = sFcol[(2 * jb) * 64], nc = sFcol[(2 * jb + 1) * 64];
        for (int j = jb; j < je; ++j) {
            const f4v k = nk, o = no;
            const float s = ns, co = nc;
            const int jn = j + 1 < je ? j + 1 : j;  // the next step's operands are requested before this one is composed
            nk = *(lds_f4)&p->steps[jn].a;
            no = *(lds_f4)&p->steps[jn].ox;
            ns = sFcol[(2 * jn) * 64];
            nc = sFcol[(2 * jn + 1) * 64];
            const float a = k.x, d = k.y, sa = k.z, ca = k.w;
            const float n00 = fmaf(r00, co, r01 * s), n10 = fmaf(r10, co, r11 * s), n20 = fmaf(r20, co, r21 * s);
            const float u0 = fmaf(r01, co, -(r00 * s)), u1 = fmaf(r11, co, -(r10 * s)), u2 = fmaf(r21, co, -(r20 * s));
            t0 = fmaf(n00, a, fmaf(r02, d, t0));
            t1 = fmaf(n10, a, fmaf(r12, d, t1));
            t2 = fmaf(n20, a, fmaf(r22, d, t2));
            const float n01 = fmaf(u0, ca, r02 * sa), n11 = fmaf(u1, ca, r12 * sa), n21 = fmaf(u2, ca, r22 * sa);
            const float n02 = fmaf(r02, ca, -(u0 * sa)), n12 = fmaf(r12, ca, -(u1 * sa)), n22 = fmaf(r22, ca, -(u2 * sa));
            r00 = n00; r01 = n01; r02 = n02; r10 = n10; r11 = n11; r12 = n12; r20 = n20; r21 = n21; r22 = n22;
            if (dh_bit(c.pt, j)) {
                float* out = sXcol + (__float_as_int(o.w) & 0xff) * 64;
                if (dh_bit(c.bare, j)) {
                    out[0] = t0; out[64] = t1; out[128] = t2;
                } else {
                    out[0] = fmaf(r00, o.x, fmaf(r01, o.y, fmaf(r02, o.z, t0)));
                    out[64] = fmaf(r10, o.x, fmaf(r11, o.y, fmaf(r12, o.z, t1)));
                    out[128] = fmaf(r20, o.x, fmaf(r21, o.y, fmaf(r22, o.z, t2)));
                }
            }
            DCX_FK_TS(7 + (j < 8 ? j : 8), 0);
        }
        float* fr = sFcol + (2 * c.n_steps + 9 * ch) * 64;  // final rotation of this chain (for the reverse sweep)
        fr[0] = r00; fr[64] = r01; fr[128] = r02; fr[192] = r10; fr[256] = r11; fr[320] = r12;
        fr[384] = r20; fr[448] = r21; fr[512] = r22;
        jb = je;
    }
}

// J^T (one wave): the reverse sweep of a wrench of fk_vjp (see there), step by step from the tip of each chain
__device__ inline void dh2_vjp(dh_cptr p, const DhArgs& c, const float* sFcol, const float* sGcol, float* gqRow, int dof) {
    for (int i = 0; i < dof; ++i) gqRow[i] = 0.f;
    int jb = 0;
    for (int ch = 0; ch < c.n_chains; ++ch) {
        const int je = ch == 0 ? c.end0 : c.n_steps;
        const float* fr = sFcol + (2 * c.n_steps + 9 * ch) * 64;
        float r00 = fr[0], r01 = fr[64], r02 = fr[128], r10 = fr[192], r11 = fr[256], r12 = fr[320];
        float r20 = fr[384], r21 = fr[448], r22 = fr[512];
        float f0 = 0.f, f1 = 0.f, f2 = 0.f, n0 = 0.f, n1 = 0.f, n2 = 0.f;
        const int jl = je - 1;
        f4v nk = *(lds_f4)&p->steps[jl].a, no = *(lds_f4)&p->steps[jl].ox;
        float ns = sFcol[(2 * jl) * 64], nc = sFcol[(2 * jl + 1) * 64];
        float ng0 = 0.f, ng1 = 0.f, ng2 = 0.f;
        if (dh_bit(c.pt, jl)) {
            const float* gin = sGcol + (__float_as_int(no.w) & 0xff) * 64;
            ng0 = gin[0]; ng1 = gin[64]; ng2 = gin[128];
        }
        for (int j = jl; j >= jb; --j) {
            const f4v k = nk, o = no;
            const float s = ns, co = nc, g0 = ng0, g1 = ng1, g2 = ng2;
            const int meta = __float_as_int(o.w);
            if (j > jb) {  // step j - 1's operands, its point's upstream gradient included, are requested now
                const int jn = j - 1;
                nk = *(lds_f4)&p->steps[jn].a;
                no = *(lds_f4)&p->steps[jn].ox;
                ns = sFcol[(2 * jn) * 64];
                nc = sFcol[(2 * jn + 1) * 64];
                if (dh_bit(c.pt, jn)) {
                    const float* gin = sGcol + ((meta >> 16) & 0xff) * 64;
                    ng0 = gin[0]; ng1 = gin[64]; ng2 = gin[128];
                }
            }
            if (dh_bit(c.pt, j)) {
                const float l0 = fmaf(r00, g0, fmaf(r10, g1, r20 * g2));
                const float l1 = fmaf(r01, g0, fmaf(r11, g1, r21 * g2));
                const float l2 = fmaf(r02, g0, fmaf(r12, g1, r22 * g2));
                f0 += l0; f1 += l1; f2 += l2;
                if (!dh_bit(c.bare, j)) {
                    n0 = fmaf(o.y, l2, fmaf(-o.z, l1, n0));
                    n1 = fmaf(o.z, l0, fmaf(-o.x, l2, n1));
                    n2 = fmaf(o.x, l1, fmaf(-o.y, l0, n2));
                }
            }
            const float a = k.x, d = k.y, sa = k.z, ca = k.w;
            const float fy = fmaf(ca, f1, -(sa * f2)), fz = fmaf(sa, f1, ca * f2);
            const float mx = fmaf(-d, fy, n0);
            const float my = fmaf(d, f0, fmaf(-a, fz, fmaf(ca, n1, -(sa * n2))));
            const float mz = fmaf(a, fy, fmaf(sa, n1, ca * n2));
            if (dh_bit(c.real, j)) gqRow[(meta >> 8) & 0xff] += mz;
            if (j > jb) {
                f1 = fmaf(s, f0, co * fy);
                f0 = fmaf(co, f0, -(s * fy));
                f2 = fz;
                n0 = fmaf(co, mx, -(s * my));
                n1 = fmaf(s, mx, co * my);
                n2 = mz;
                const float u0 = fmaf(ca, r01, -(sa * r02)), u1 = fmaf(ca, r11, -(sa * r12)), u2 = fmaf(ca, r21, -(sa * r22));
                r02 = fmaf(sa, r01, ca * r02); r12 = fmaf(sa, r11, ca * r12); r22 = fmaf(sa, r21, ca * r22);
                r01 = fmaf(s, r00, co * u0); r11 = fmaf(s, r10, co * u1); r21 = fmaf(s, r20, co * u2);
                r00 = fmaf(co, r00, -(s * u0)); r10 = fmaf(co, r10, -(s * u1)); r20 = fmaf(co, r20, -(s * u2));
            }
            DCX_FK_TS(7 + (j < 8 ? j : 8), 1);
        }
        jb = je;
    }
}

// ---- the step-table walks on SEVERAL waves (round 3) --------------------------------------------------------------------
// A lone wave issues one instruction per ~5.5 cycles however simple (independent v_fma_f32; ~10 when dependent; an LDS
// round trip 66, a taken branch ~30: tools/lone_wave_ubench.hip, profiles/r03_lone_wave.txt), so the chain and J^T phases
// are priced by the instructions ONE wave executes, not by flops.  Under T <- T * A the rows of [R | t] never mix: the
// chain is split by rows over two waves - role 0 carries rows 0 and 1 as PACKED pairs (v_pk_fma_f32: two rows for the
// issue slots of one), role 1 carries row 2 - ten arithmetic instructions per step and wave instead of thirty.  Same
// expressions per entry as dh2_chain (v_pk_fma_f32 is fmaf per half): bit-identical.  The walks themselves follow below,
// unrolled per chain.
template <class V>
__device__ __forceinline__ V dh_splat(float x) {
    if constexpr (__is_same(V, float)) return x;
    else return V{x, x};
}
// Scalar results are made opaque to the optimiser where they are produced: the SLP vectoriser otherwise pairs the row-2 /
// wrench arithmetic into v_pk_* with a v_mov per operand half (+8 instructions per step on the wave that can least afford
// them), while the rest of the translation unit - the C > 1 sweeps - gains from it (profiles/r03_dev_d_bench.txt).
__device__ __forceinline__ float dh_opaque(float x) {
    asm("" : "+v"(x));
    return x;
}
template <class V>
__device__ __forceinline__ V dh_fma(V a, V b, V c) {
    if constexpr (__is_same(V, float)) return dh_opaque(fmaf(a, b, c));
    else return __builtin_elementwise_fma(a, b, c);
}
// ---- the same walks UNROLLED per chain (chains of at most kDhUnroll steps: every DH robot the reference ships) ----------
// The loops above still spend three instructions of bookkeeping (register rotation of the prefetched operands, table
// address arithmetic, loop control) for every one of arithmetic.  With the step index a compile-time constant the table
// reads become immediate offsets from the chain's first record, the masks are tested with s_bitcmp on an immediate bit,
// and nothing rotates.  Steps past the chain's end are skipped by one scalar compare each.  A robot's chains are
// independent of each other, so they run side by side on different waves.  Same expressions, same order: bit-identical
// to the loops.
constexpr int kDhUnroll = 10;
template <int I, int N, class F>
__device__ __forceinline__ void dh_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        dh_static_for<I + 1, N>(f);
    }
}
__device__ __forceinline__ bool dh2_unrollable(const DhArgs& c) {
    return c.n_chains <= 2 && c.end0 <= kDhUnroll && c.n_steps - c.end0 <= kDhUnroll;
}
struct DhChain {   // one chain's view of the table: steps [jb, jb + n), masks shifted so that bit i is step jb + i
    int ch, jb, n, ps0;
    uint32_t pt, bare, real;
};
__device__ __forceinline__ DhChain dh_chain_of(const DhArgs& c, int ch) {
    DhChain h;
    h.ch = ch;
    h.jb = ch == 0 ? 0 : c.end0;
    h.n = (ch == 0 ? c.end0 : c.n_steps) - h.jb;
    h.pt = c.pt >> h.jb;
    h.bare = c.bare >> h.jb;
    h.real = c.real >> h.jb;
    h.ps0 = __builtin_popcount(c.pt & ((1u << h.jb) - 1u));  // point steps of the chains in front (jb < 32)
    return h;
}

// one step's operands, requested one step ahead of their use, unconditionally (reads past the chain's end stay inside the
// staged table's LDS allocation and inside the frames: the values are never used)
struct DhOps {
    f4v k, o;
    float s, co;
};
template <bool WITH_O>
__device__ __forceinline__ DhOps dh_load_ops(const __attribute__((address_space(3))) DhStep* st, const float* sc, int i) {
    DhOps r;
    r.k = *(lds_f4)&st[i].a;
    if constexpr (WITH_O) r.o = *(lds_f4)&st[i].ox;
    r.s = sc[(2 * i) * 64];
    r.co = sc[(2 * i + 1) * 64];
    return r;
}

template <int ROLE>
__device__ __forceinline__ void dh2_chain_rows_u(dh_cptr p, const DhArgs& c, const DhChain& h, float* sXcol, float* sFcol) {
    using V = typename std::conditional<ROLE == 0, f2v, float>::type;
    const f4v b0 = *(lds_f4)&p->base[h.ch][0], b1 = *(lds_f4)&p->base[h.ch][4], b2 = *(lds_f4)&p->base[h.ch][8];
    V r0, r1, r2, t;
    if constexpr (ROLE == 0) { r0 = V{b0.x, b1.x}; r1 = V{b0.y, b1.y}; r2 = V{b0.z, b1.z}; t = V{b0.w, b1.w}; }
    else { r0 = b2.x; r1 = b2.y; r2 = b2.z; t = b2.w; }
    const auto* st = &p->steps[h.jb];
    const float* sc = sFcol + (2 * h.jb) * 64;
    DhOps nx = dh_load_ops<true>(st, sc, 0);
    dh_static_for<0, kDhUnroll>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        const DhOps cu = nx;
        if constexpr (j + 1 < kDhUnroll) nx = dh_load_ops<true>(st, sc, j + 1);
        if (j < h.n) {
            const V s = dh_splat<V>(cu.s), co = dh_splat<V>(cu.co);
            const V a = dh_splat<V>(cu.k.x), d = dh_splat<V>(cu.k.y), sa = dh_splat<V>(cu.k.z), ca = dh_splat<V>(cu.k.w);
            const V n0 = dh_fma<V>(r0, co, r1 * s);
            const V u = dh_fma<V>(r1, co, -(r0 * s));
            t = dh_fma<V>(n0, a, dh_fma<V>(r2, d, t));
            const V n1 = dh_fma<V>(u, ca, r2 * sa);
            const V n2 = dh_fma<V>(r2, ca, -(u * sa));
            r0 = n0; r1 = n1; r2 = n2;
            if (dh_bit(h.pt, j)) {
                V pt = t;
                if (!dh_bit(h.bare, j))
                    pt = dh_fma<V>(r0, dh_splat<V>(cu.o.x), dh_fma<V>(r1, dh_splat<V>(cu.o.y), dh_fma<V>(r2, dh_splat<V>(cu.o.z), t)));
                float* out = sXcol + (__float_as_int(cu.o.w) & 0xff) * 64;
                if constexpr (ROLE == 0) { out[0] = pt.x; out[64] = pt.y; }
                else out[128] = pt;
            }
        }
    });
    float* fr = sFcol + (2 * c.n_steps + 9 * h.ch) * 64;
    if constexpr (ROLE == 0) {
        fr[0] = r0.x; fr[64] = r1.x; fr[128] = r2.x; fr[192] = r0.y; fr[256] = r1.y; fr[320] = r2.y;
    } else {
        fr[384] = r0; fr[448] = r1; fr[512] = r2;
    }
}
// wave w < 2 * n_chains composes rows (role w & 1) of chain w >> 1; caller: block barrier afterwards
__device__ __forceinline__ void dh2_chain_rows_sel(dh_cptr p, const DhArgs& c, float* sXcol, float* sFcol, int w) {
    if (w >= 2 * c.n_chains) return;
    const DhChain h = dh_chain_of(c, w >> 1);
    if (w & 1) dh2_chain_rows_u<1>(p, c, h, sXcol, sFcol);
    else dh2_chain_rows_u<0>(p, c, h, sXcol, sFcol);
}

// phase R1 of dh2_vjp_waves, unrolled (steps in descending order)
template <int ROLE>
__device__ __forceinline__ void dh2_vjp_r1_u(dh_cptr p, const DhArgs& c, const DhChain& h, const float* sFcol, float* scr, int psw = 12) {
    using V = typename std::conditional<ROLE == 0, f2v, float>::type;
    const float* fr = sFcol + (2 * c.n_steps + 9 * h.ch) * 64;
    V r0, r1, r2;
    if constexpr (ROLE == 0) { r0 = V{fr[0], fr[192]}; r1 = V{fr[64], fr[256]}; r2 = V{fr[128], fr[320]}; }
    else { r0 = fr[384]; r1 = fr[448]; r2 = fr[512]; }
    const auto* st = &p->steps[h.jb];
    const float* sc = sFcol + (2 * h.jb) * 64;
    DhOps nx = dh_load_ops<false>(st, sc, kDhUnroll - 1);
    dh_static_for<0, kDhUnroll>([&](auto ic) __attribute__((always_inline)) {
        constexpr int j = kDhUnroll - 1 - decltype(ic)::value;
        const DhOps cu = nx;
        if constexpr (j > 0) nx = dh_load_ops<false>(st, sc, j - 1);
        if (j < h.n) {
            if (dh_bit(h.pt, j)) {
                const int ps = h.ps0 + __builtin_popcount(h.pt & ((1u << j) - 1u));
                float* o = scr + (psw * ps) * 64;
                if constexpr (ROLE == 0) {
                    o[0] = r0.x; o[64] = r1.x; o[128] = r2.x; o[192] = r0.y; o[256] = r1.y; o[320] = r2.y;
                } else {
                    o[384] = r0; o[448] = r1; o[512] = r2;
                }
            }
            if constexpr (j > 0) {
                const V s = dh_splat<V>(cu.s), co = dh_splat<V>(cu.co);
                const V sa = dh_splat<V>(cu.k.z), ca = dh_splat<V>(cu.k.w);
                const V u = dh_fma<V>(ca, r1, -(sa * r2));
                r2 = dh_fma<V>(sa, r1, ca * r2);
                r1 = dh_fma<V>(s, r0, co * u);
                r0 = dh_fma<V>(co, r0, -(s * u));
            }
        }
    });
}
__device__ __forceinline__ void dh2_vjp_r1_sel(dh_cptr p, const DhArgs& c, const float* sFcol, float* scr, int w, int psw = 12) {
    if (w < 0 || w >= 2 * c.n_chains) return;
    const DhChain h = dh_chain_of(c, w >> 1);
    if (w & 1) dh2_vjp_r1_u<1>(p, c, h, sFcol, scr, psw);
    else dh2_vjp_r1_u<0>(p, c, h, sFcol, scr, psw);
}

// phase R2 of dh2_vjp_waves, unrolled: one chain's wrench recurrence on one wave (gqRow zeroed by the caller)
__device__ __forceinline__ void dh2_vjp_r2_u(dh_cptr p, const DhArgs& c, const DhChain& h, const float* sFcol, const float* scr,
                                             float* gqRow, int psw = 12, int loff = 9) {
    float f0 = 0.f, f1 = 0.f, f2 = 0.f, n0 = 0.f, n1 = 0.f, n2 = 0.f;
    const auto* st = &p->steps[h.jb];
    const float* sc = sFcol + (2 * h.jb) * 64;
    DhOps nx = dh_load_ops<true>(st, sc, kDhUnroll - 1);
    dh_static_for<0, kDhUnroll>([&](auto ic) __attribute__((always_inline)) {
        constexpr int j = kDhUnroll - 1 - decltype(ic)::value;
        const DhOps cu = nx;
        if constexpr (j > 0) nx = dh_load_ops<true>(st, sc, j - 1);
        if (j < h.n) {
            const float s = cu.s, co = cu.co;
            const int meta = __float_as_int(cu.o.w);
            if (dh_bit(h.pt, j)) {
                const int ps = h.ps0 + __builtin_popcount(h.pt & ((1u << j) - 1u));
                const float* l = scr + (psw * ps + loff) * 64;
                const float l0 = l[0], l1 = l[64], l2 = l[128];
                f0 = dh_opaque(f0 + l0); f1 = dh_opaque(f1 + l1); f2 = dh_opaque(f2 + l2);
                if (!dh_bit(h.bare, j)) {
                    n0 = dh_opaque(fmaf(cu.o.y, l2, fmaf(-cu.o.z, l1, n0)));
                    n1 = dh_opaque(fmaf(cu.o.z, l0, fmaf(-cu.o.x, l2, n1)));
                    n2 = dh_opaque(fmaf(cu.o.x, l1, fmaf(-cu.o.y, l0, n2)));
                }
            }
            const float a = cu.k.x, d = cu.k.y, sa = cu.k.z, ca = cu.k.w;
            const f2v y1 = {f1, n1}, y2 = {f2, n2}, sa2 = {sa, sa}, ca2 = {ca, ca};
            const f2v yy = __builtin_elementwise_fma(ca2, y1, -(sa2 * y2));
            const f2v zz = __builtin_elementwise_fma(sa2, y1, ca2 * y2);
            const float fy = yy.x, fz = zz.x;
            const float mx = dh_opaque(fmaf(-d, fy, n0));
            const float my = dh_opaque(fmaf(d, f0, fmaf(-a, fz, yy.y)));
            const float mz = dh_opaque(fmaf(a, fy, zz.y));
            if (dh_bit(h.real, j)) gqRow[(meta >> 8) & 0xff] += mz;
            if constexpr (j > 0) {
                const f2v x0 = {f0, mx}, x1 = {fy, my}, s2 = {s, s}, c2 = {co, co};
                const f2v w1 = __builtin_elementwise_fma(s2, x0, c2 * x1);
                const f2v w0 = __builtin_elementwise_fma(c2, x0, -(s2 * x1));
                f0 = w0.x; n0 = w0.y; f1 = w1.x; n1 = w1.y;
                f2 = fz; n2 = mz;
            }
        }
    });
}

// J^T on several waves, phase by phase (the caller places block barriers between the phases; every expression is
// dh2_vjp's, so the result is bit-identical to it):
//   R1  (two waves per chain: dh2_vjp_r1_sel): the rotations R_j are recomputed from each chain's tip backwards, row-split
//        like the chain (R_{j-1} = R_j Rx^T Rz^T row by row), and the rows of every POINT step's R_j are left in `scr`.
//        Needs only the frames, not the gradient: a split launch runs it beside the arrival count of its hand-over;
//   R1b (wave ps, ps + nw, ...: dh2_vjp_r1b): l = R_j^T (scale * g) of point step ps - the nine FMAs per point that need
//        the rotation; wave 0 also clears the gradient row;
//   R2  (one wave per chain: dh2_vjp_r2_sel): the wrench recurrence itself, which now only reads l: ~30 instructions per step.
// scr: 12 columns per point step ([9] rotation, row-major, [3] l), per lane.  sGtot: the UNSCALED feature gradient columns;
// `scale` is this lane's factor (what the one-wave epilogue multiplies in while staging G).
__device__ __forceinline__ void dh2_vjp_r1b(dh_cptr p, const DhArgs& c, const float* sGtot, float scale, float* scr, float* gqRow,
                                            int dof, int wave, int nw, int psw = 12, int loff = 9) {
    if (wave == 0)
        for (int i = 0; i < dof; ++i) gqRow[i] = 0.f;
    for (int ps = wave; ps < c.n_pt; ps += nw) {
        const float* R = scr + (psw * ps) * 64;
        const float* gin = sGtot + p->ps_col[ps] * 64;
        const float g0 = gin[0] * scale, g1 = gin[64] * scale, g2 = gin[128] * scale;
        const float r00 = R[0], r01 = R[64], r02 = R[128], r10 = R[192], r11 = R[256], r12 = R[320];
        const float r20 = R[384], r21 = R[448], r22 = R[512];
        float* l = scr + (psw * ps + loff) * 64;
        l[0] = fmaf(r00, g0, fmaf(r10, g1, r20 * g2));
        l[64] = fmaf(r01, g0, fmaf(r11, g1, r21 * g2));
        l[128] = fmaf(r02, g0, fmaf(r12, g1, r22 * g2));
    }
}
// wave 0 takes chain 0, wave 1 chain 1 - unless the chains drive a common q entry (c.shared_q: their updates of that
// entry must then happen in chain order, on one wave)
__device__ __forceinline__ void dh2_vjp_r2_sel(dh_cptr p, const DhArgs& c, const float* sFcol, const float* scr, float* gqRow, int wave,
                                               int psw = 12, int loff = 9) {
    if (c.n_chains == 1 || c.shared_q) {
        if (wave != 0) return;
        for (int ch = 0; ch < c.n_chains; ++ch) dh2_vjp_r2_u(p, c, dh_chain_of(c, ch), sFcol, scr, gqRow, psw, loff);
    } else if (wave >= 0 && wave < 2) {
        dh2_vjp_r2_u(p, c, dh_chain_of(c, wave), sFcol, scr, gqRow, psw, loff);
    }
}

// ---- which walk a launch uses ------------------------------------------------------------------------------------------
// fkk = 0: FkProg interpreted from its LDS copy (every transform kind);  1: DH arms, FkProg through scalar loads;
// 2: DH arms, the step table (default where the model has one).  One staged program per launch.
struct FkWalk {
    int fkk;
    const FkProg* g;
    fk_cptr fk;   // fkk != 2
    dh_cptr dh;   // fkk == 2
};
__device__ __forceinline__ FkWalk fk_stage_sel(int fkk, const FkProg* g, int fk_dwords, const DhArgs& c, float* lds, int tid,
                                               int nthreads) {
    FkWalk w;
    w.fkk = fkk;
    w.g = g;
    w.fk = nullptr;
    w.dh = nullptr;
    if (fkk == 2) w.dh = stage_dh_prog(c, lds, tid, nthreads);
    else w.fk = stage_fk_prog(g, lds, tid, nthreads, fk_dwords);
    return w;
}
__device__ __forceinline__ bool fk_is_tree(const FkWalk& w) { return w.fkk == 2 ? false : rfl(w.fk->kind) == DCX_FK_TREE; }
__device__ __forceinline__ void fk_trig_sel(const FkWalk& w, const DhArgs& c, const float* sQrow, float* sFcol, int wave, int nw) {
    if (w.fkk == 2) dh2_trig(w.dh, c, sQrow, sFcol, wave, nw);
    else fk_forward_trig(w.fk, sQrow, sFcol, wave, nw);
}
__device__ __forceinline__ void fk_chain_sel(const FkWalk& w, const DhArgs& c, const float* sQrow, float* sXcol, float* sFcol) {
    if (w.fkk == 2) dh2_chain(w.dh, c, sXcol, sFcol);
    else if (w.fkk == 1) fk_forward_chain_dh_k((fk_kptr)(uintptr_t)w.g, sXcol, sFcol);
    else fk_forward_chain(w.fk, sQrow, sXcol, sFcol);
}
__device__ __forceinline__ void fk_vjp_sel(const FkWalk& w, const DhArgs& c, const float* sQrow, const float* sFcol,
                                           const float* sGcol, float* gqRow, int dof) {
    if (w.fkk == 2) dh2_vjp(w.dh, c, sFcol, sGcol, gqRow, dof);
    else if (w.fkk == 1) fk_vjp_dh_k((fk_kptr)(uintptr_t)w.g, sFcol, sGcol, gqRow);
    else fk_vjp(w.fk, sQrow, sFcol, sGcol, gqRow);
}

}  // namespace dcx
