/*
 * oracle/reacher_ref.c  --  TEST ORACLE, NOT PRODUCT CODE.
 *
 * Plain-C, FP64, deliberately simple restatement of the physics half of mjmpc's hot path
 * for reacher_7dof-v0.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library; nothing under mjmpc_amd/ does.
 *
 * PARITY UNPINNED: the arithmetic restated here lives in closed-source MuJoCo 2.0
 * (mujoco-py >=2.0,<2.1, reference setup/environment.yml:19, setup.py:26), which is not in
 * /root/reference and cannot be installed here; the reference's tests hold no golden vector
 * for it.  What follows restates MuJoCo's *published* algorithm (Computation chapter of the
 * MuJoCo documentation; Todorov 2014 "Convex and analytically-invertible dynamics with
 * contacts and constraints") and is anchored on the reference call sites:
 *
 *   model          mjmpc/envs/assets/xml/sawyer.xml:2-6,11-59,101-109
 *   env.step       mjmpc/envs/basic/reacher_env.py:29-39   (do_simulation x frame_skip, reward)
 *   get_obs        mjmpc/envs/basic/reacher_env.py:41-47
 *   set_env_state  mjmpc/envs/basic/reacher_env.py:87-99
 *   rollout loop   mjmpc/envs/gym_env_wrapper.py:89-156
 *
 * The formulation is chosen to be INDEPENDENT of the HIP kernel's: the mass matrix is built
 * from body Jacobians (not CRBA), bias forces from an inertial-frame Newton-Euler pass, all
 * linear solves are dense Cholesky, and the constraint problem is minimised by Newton with an
 * exact piecewise-quadratic line search.
 *
 * One mj_step  (MuJoCo: mj_forward, then mj_Euler):
 *   1. kinematics (body frames, site)            4. passive (-damping*v) + motor (gear*clip(ctrl))
 *   2. M(q) + armature                           5. limit / sphere-plane rows, soft-constraint solve
 *   3. bias c(q,v)                               6. semi-implicit Euler with implicit joint damping
 *
 * Rounds 2-5 widened the same file to every model the tree engine runs (trees, slide / ball / free joints, springs, the fluid model,
 * friction cones - pyramidal and, round 5, ELLIPTIC with impratio -, friction loss, equalities, tendons, boxes, cylinders, static
 * geoms, joint margin / ref, geom gap, direct solref, mj_checkPos / Vel / Acc -> mj_resetData).  All of it is [EXT] restatement and
 * UNPINNED in the same way; where MuJoCo's own routine could not be restated to the rounding (capsule-box, box-box, sphere /
 * capsule against cylinder) the function's comment says "a scheme of its own" and DESIGN.md 2 says what that means.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXB 48            /* bodies incl. world */
#define MAXV 32
#define MAXQ (MAXV + 8)    /* qpos: a ball joint takes 4 entries for 3 dofs, a free joint 7 for 6 */
#define MAXG 96
#define MAXS 32            /* collision spheres (a colliding capsule is its two end spheres, a box its eight corners) */
#define MAXE 8             /* equality constraints */
#define MAXT 8             /* fixed tendons */
#define MAXTJ 4            /* joints per tendon */
#define MAXC (3 * MAXV + 4 * MAXS + 4 * 4 * 16 + 6 * MAXE + 2 * MAXT)    /* (a geom pair: up to four contacts of four rows) */
#define MJ_MINVAL 1e-15    /* MuJoCo mjMINVAL */

#define HEADER_LEN 80
#define BODY_STRIDE 56
#define GEOM_STRIDE 33
#define ACT_STRIDE 14
#define PAIR_STRIDE 12
#define EQ_STRIDE 28
#define TENDON_STRIDE (8 + 2 * MAXTJ + 7)
#define MAXP 16            /* geom-geom collision pairs */
/* joint kinds (mjmpc_amd/models/raw.py) */
#define JHINGE 1
#define JSLIDE 2
#define JBALL 3
#define OR_PI 3.14159265358979323846
#define JFREE 4
/* constraint row kinds of the primal problem (MuJoCo mj_constraintUpdate) */
#define ROW_UNI 0          /* limits, contacts: cost 1/2 D min(0, r)^2 */
#define ROW_EQ 1           /* equality: 1/2 D r^2 */
#define ROW_FRIC 2         /* friction loss f: Huber - 1/2 D r^2 inside |r| < R f, linear f |r| - 1/2 R f^2 outside */
#define ROW_ELL 3          /* the normal row of an ELLIPTIC-cone contact (round 5); the two rows behind it are its tangents: */
#define ROW_ELLT 4         /* ... which the normal row's entry handles (cone_eval) */

typedef struct {
    int nbody, nv, nq, nu;
    double timestep, gravity[3];
    int frame_skip;
    double solref[2], solimp[5];
    int cone;                                /* 0: pyramidal friction cones (MuJoCo's default), 1: elliptic (MJCF option cone) */
    double impratio;                         /* elliptic cones: R of the friction rows = R of the normal row / impratio */
    /* tree */
    int parent[MAXB];
    double bpos[MAXB][3], bR0[MAXB][9], bquat0[MAXB][4];      /* fixed offset / rotation in the parent frame */
    int dofid[MAXB];                         /* address of the body's first dof, -1: welded */
    int qadr[MAXB];                          /* address of the joint's first qpos entry */
    int jtype[MAXB];                         /* 1 hinge, 2 slide, 3 ball, 4 free */
    double jaxis[MAXB][3], jpos[MAXB][3];    /* axis and anchor in the body frame */
    int dof_body[MAXV];
    double range[MAXV][2], jmargin[MAXV];    /* limits of hinge / slide joints: the row exists while dist < jmargin (MJCF joint margin) */
    int limited[MAXV];
    int ball_limited[MAXB];                  /* a ball joint's limit: the rotation angle stays below ball_range (MJCF range="0 max") */
    double ball_range[MAXB];
    double damping[MAXV], armature[MAXV], stiffness[MAXV], springref[MAXV], frictionloss[MAXV];
    int dof_type[MAXV];                      /* 1 rotation about xaxis through xanchor, 2 translation along xaxis */
    int dof_qadr[MAXV];                      /* hinge / slide dofs: their qpos entry (-1 for ball / free dofs) */
    double solref_f[2], solimp_f[5];         /* friction-loss rows (the model's set; every dof may carry its own) */
    double dof_solref_l[MAXV][2], dof_solimp_l[MAXV][5], dof_solref_f[MAXV][2], dof_solimp_f[MAXV][5];
    /* inertial (inertiafromgeom) */
    double mass[MAXB], ipos[MAXB][3], inertia[MAXB][9];  /* tensor about the COM, body frame */
    /* motors */
    int act_dof[MAXV], act_tendon[MAXV];     /* act_tendon: the actuator pulls on fixed tendon act_dof[a] */
    double gear[MAXV], ctrl_lo[MAXV], ctrl_hi[MAXV];
    /* MuJoCo actuator with gaintype fixed / biastype affine (mj_fwdActuation [EXT]): scalar force = gain ctrl + bias[0] +
     * bias[1] length + bias[2] velocity, length = gear q; a motor is gain 1, <position kp> gain kp / bias (0, -kp, 0),
     * <velocity kv> gain kv / bias (0, 0, -kv) */
    double act_gain[MAXV], act_bias[MAXV][3];
    int ctrllimited[MAXV], forcelimited[MAXV];
    double forcerange[MAXV][2];              /* mj_fwdActuation clamps the scalar force when forcelimited */
    /* site + target */
    int site_body;
    double site_pos[3], target_default[3];
    /* contact: one plane (world) versus collision spheres */
    int has_plane, nsphere;
    double plane_pos[3], plane_n[3], plane_margin, plane_gap;
    int sph_body[MAXS];
    double sph_pos[MAXS][3], sph_r[MAXS], sph_margin[MAXS], sph_gap[MAXS];    /* (geom gap: MuJoCo's includemargin = margin - gap) */
    double sph_mu[MAXS], sph_axis[MAXS][3];   /* friction (0: frictionless row) and capsule axis in the body frame */
    /* round 5: 1 = a box's corner (mjc_PlaneBox: corners above the box centre are skipped, at most four contacts per box),
     * 2 = one of a cylinder's four candidate points on the plane (mjc_PlaneCylinder); sph_k = its index in its geom's group,
     * sph_ctr = the geom's centre in the body frame; cylinders: sph_axis = the axis, sph_r = radius, sph_hh = half height */
    int sph_kind[MAXS], sph_k[MAXS];
    double sph_ctr[MAXS][3], sph_hh[MAXS];
    double sph_solref[MAXS][2], sph_solimp[MAXS][5];   /* the contact's solver parameters (mj_contactParam: geom x plane) */
    /* geom-geom pairs: two segments (a sphere is a segment of length 0) with radii, on two bodies */
    int npair, pair_body[MAXP][2];
    double pair_a[MAXP][2][3], pair_d[MAXP][2][3], pair_r[MAXP][2], pair_margin[MAXP], pair_mu[MAXP];
    double pair_solref[MAXP][2], pair_solimp[MAXP][5];
    int pair_box[MAXP];                      /* -1: two segments; e: geom e of the pair is a BOX (the other a sphere) */
    double pair_R[MAXP][9], pair_half[MAXP][3];   /* that box: orientation in its body's frame, half sizes (pair_a = centre) */
    double pair_R2[MAXP][9], pair_half2[MAXP][3]; /* pair_box = 2 (round 5): BOTH geoms are boxes - geom 0's in pair_R / pair_half, geom 1's here */
    int pair_cyl[MAXP];                      /* -1, or e (round 5): geom e of the pair is a CYLINDER (pair_a / pair_d: its axis, end to end; */
    double pair_cyl_r[MAXP];                 /* ... its radius here, pair_r = 0), the other geom a sphere or a capsule */
    /* equality constraints (MJCF <equality>): connect / weld (two bodies, 0 = world) and joint (two dofs, -1 = none) */
    int neq, eq_type[MAXE], eq_o1[MAXE], eq_o2[MAXE];
    double eq_anchor[MAXE][2][3], eq_relR[MAXE][9], eq_poly[MAXE][5], eq_solref[MAXE][2], eq_solimp[MAXE][5];
    /* fixed tendons: length = sum coef q over hinge / slide dofs; limit rows */
    int ntendon, tn_n[MAXT], tn_dof[MAXT][MAXTJ], tn_limited[MAXT];
    double tn_coef[MAXT][MAXTJ], tn_range[MAXT][2], tn_margin[MAXT], tn_invweight0[MAXT];
    double tn_solref[MAXT][2], tn_solimp[MAXT][5];
    /* TASK 2 (in-hand reorientation): object axis in the site body's frame, direction it should point in */
    double site_axis[3], target_dir[3];
    /* joint-limit rows may carry their own solver parameters (MJCF solreflimit / solimplimit) */
    double solref_l[2], solimp_l[5];
    /* medium: MuJoCo's inertia-box fluid model; principal frame / equivalent box of every body */
    double density, viscosity;
    double iR[MAXB][9], ibox[MAXB][3];
    /* task: 0 reach (reacher_env.py), 1 forward progress (swimmer.py / half_cheetah.py) */
    int task, obs_skip;
    double ctrl_cost;
    /* constants computed at qpos0 (MuJoCo mj_setConst) */
    double dof_invweight0[MAXV], body_invweight0[MAXB], body_invweight0r[MAXB];     /* (translational, rotational) */
    double qpos0[MAXQ];
    /* statistics */
    long newton_iters, newton_calls, newton_fail;
    long resets;                             /* mj_resetData calls of mj_checkPos / mj_checkVel / mj_checkAcc (or_step) */
} OrModel;

/* ---------------------------------------------------------------- small linear algebra */
static void cross3(const double *a, const double *b, double *c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
static double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void matvec3(const double *R, const double *x, double *y) {
    for (int i = 0; i < 3; i++) y[i] = R[3 * i] * x[0] + R[3 * i + 1] * x[1] + R[3 * i + 2] * x[2];
}
static void matmul3(const double *A, const double *B, double *C) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += A[3 * i + k] * B[3 * k + j];
            C[3 * i + j] = s;
        }
}
static void quat2mat(const double *q, double *R) {
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    double w = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z);     R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y);     R[7] = 2 * (y * z + w * x);     R[8] = 1 - 2 * (x * x + y * y);
}
/* rotation by angle about a unit axis (Rodrigues) */
static void axisangle2mat(const double *u, double ang, double *R) {
    double c = cos(ang), s = sin(ang), t = 1 - c;
    R[0] = c + t * u[0] * u[0];        R[1] = t * u[0] * u[1] - s * u[2]; R[2] = t * u[0] * u[2] + s * u[1];
    R[3] = t * u[0] * u[1] + s * u[2]; R[4] = c + t * u[1] * u[1];        R[5] = t * u[1] * u[2] - s * u[0];
    R[6] = t * u[0] * u[2] - s * u[1]; R[7] = t * u[1] * u[2] + s * u[0]; R[8] = c + t * u[2] * u[2];
}
/* quaternion product (w, x, y, z) and MuJoCo mju_quatIntegrate: q <- q * exp(1/2 w h), w in the local frame */
static void quatmul(const double *a, const double *b, double *c) {
    c[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    c[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    c[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    c[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}
static void quat_integrate(double *q, const double *w, double h) {
    double n = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]), ang = n * h;
    if (n < MJ_MINVAL) return;
    double sn = sin(0.5 * ang) / n, r[4] = {cos(0.5 * ang), w[0] * sn, w[1] * sn, w[2] * sn}, t[4];
    quatmul(q, r, t);
    double nn = sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2] + t[3] * t[3]);
    for (int i = 0; i < 4; i++) q[i] = t[i] / nn;
}
/* unit quaternion (w, x, y, z) of a rotation matrix, w >= 0 */
static void mat2quat(const double *R, double *q) {
    double tr = R[0] + R[4] + R[8];
    if (tr > 0) {
        double sq = sqrt(tr + 1.0) * 2;
        q[0] = 0.25 * sq; q[1] = (R[7] - R[5]) / sq; q[2] = (R[2] - R[6]) / sq; q[3] = (R[3] - R[1]) / sq;
    } else if (R[0] > R[4] && R[0] > R[8]) {
        double sq = sqrt(1.0 + R[0] - R[4] - R[8]) * 2;
        q[0] = (R[7] - R[5]) / sq; q[1] = 0.25 * sq; q[2] = (R[1] + R[3]) / sq; q[3] = (R[2] + R[6]) / sq;
    } else if (R[4] > R[8]) {
        double sq = sqrt(1.0 + R[4] - R[0] - R[8]) * 2;
        q[0] = (R[2] - R[6]) / sq; q[1] = (R[1] + R[3]) / sq; q[2] = 0.25 * sq; q[3] = (R[5] + R[7]) / sq;
    } else {
        double sq = sqrt(1.0 + R[8] - R[0] - R[4]) * 2;
        q[0] = (R[3] - R[1]) / sq; q[1] = (R[2] + R[6]) / sq; q[2] = (R[5] + R[7]) / sq; q[3] = 0.25 * sq;
    }
    if (q[0] < 0) for (int i = 0; i < 4; i++) q[i] = -q[i];
}
/* dense Cholesky A = L L^T (lower, in place); returns 0 on success */
static int chol(double *A, int n) {
    for (int j = 0; j < n; j++) {
        double d = A[j * n + j];
        for (int k = 0; k < j; k++) d -= A[j * n + k] * A[j * n + k];
        if (d <= 0) return 1;
        d = sqrt(d);
        A[j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = A[i * n + j];
            for (int k = 0; k < j; k++) s -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = s / d;
        }
    }
    return 0;
}
static void chol_solve(const double *L, int n, double *x) {
    for (int i = 0; i < n; i++) {
        double s = x[i];
        for (int k = 0; k < i; k++) s -= L[i * n + k] * x[k];
        x[i] = s / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = x[i];
        for (int k = i + 1; k < n; k++) s -= L[k * n + i] * x[k];
        x[i] = s / L[i * n + i];
    }
}

/* ---------------------------------------------------------------- kinematics */
typedef struct {
    double xpos[MAXB][3], xmat[MAXB][9];    /* body frame origin / orientation in the world */
    double xipos[MAXB][3];                  /* body COM in the world */
    double xaxis[MAXV][3], xanchor[MAXV][3];
} Kin;

/* MuJoCo mj_kinematics: xpos / xquat = parent o (body pos, quat), then the body's joint: a hinge or ball turns the frame
 * about its anchor (jnt pos, fixed in both frames), a slide moves it along its axis, a free joint replaces it by
 * qpos[0:3], qpos[3:7].  Per dof: world axis and anchor (ball / free rotations: the body's own x, y, z axes). */
static void kinematics(const OrModel *m, const double *q, Kin *k) {
    memset(k->xpos[0], 0, sizeof(k->xpos[0]));
    static const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    memcpy(k->xmat[0], I3, sizeof(I3));
    memset(k->xipos[0], 0, sizeof(k->xipos[0]));
    for (int b = 1; b < m->nbody; b++) {
        int p = m->parent[b];
        double t[3], R[9];
        matvec3(k->xmat[p], m->bpos[b], t);
        for (int i = 0; i < 3; i++) k->xpos[b][i] = k->xpos[p][i] + t[i];
        matmul3(k->xmat[p], m->bR0[b], R);
        int j = m->dofid[b], jt = j >= 0 ? m->jtype[b] : 0;
        if (jt == JSLIDE) {                     /* slide: the frame moves along its axis, no rotation */
            memcpy(k->xmat[b], R, sizeof(R));
            matvec3(R, m->jaxis[b], k->xaxis[j]);
            for (int i = 0; i < 3; i++) k->xpos[b][i] += k->xaxis[j][i] * (q[m->qadr[b]] - m->qpos0[m->qadr[b]]);
            memcpy(k->xanchor[j], k->xpos[b], sizeof(double) * 3);
        } else if (jt == JHINGE || jt == JBALL) {
            double E[9], anchor[3];
            matvec3(R, m->jpos[b], t);
            for (int i = 0; i < 3; i++) anchor[i] = k->xpos[b][i] + t[i];
            if (jt == JHINGE) axisangle2mat(m->jaxis[b], q[m->qadr[b]] - m->qpos0[m->qadr[b]], E);
            else quat2mat(q + m->qadr[b], E);
            matmul3(R, E, k->xmat[b]);
            matvec3(k->xmat[b], m->jpos[b], t);
            for (int i = 0; i < 3; i++) k->xpos[b][i] = anchor[i] - t[i];
            if (jt == JHINGE) {
                matvec3(k->xmat[b], m->jaxis[b], k->xaxis[j]);
                memcpy(k->xanchor[j], anchor, sizeof(anchor));
            } else {
                for (int c = 0; c < 3; c++) {
                    for (int i = 0; i < 3; i++) k->xaxis[j + c][i] = k->xmat[b][3 * i + c];
                    memcpy(k->xanchor[j + c], anchor, sizeof(anchor));
                }
            }
        } else if (jt == JFREE) {
            const double *qq = q + m->qadr[b];
            memcpy(k->xpos[b], qq, sizeof(double) * 3);
            quat2mat(qq + 3, k->xmat[b]);
            for (int c = 0; c < 3; c++) {
                for (int i = 0; i < 3; i++) {
                    k->xaxis[j + c][i] = i == c ? 1.0 : 0.0;                    /* translations: world axes */
                    k->xaxis[j + 3 + c][i] = k->xmat[b][3 * i + c];             /* rotations: body axes */
                }
                memcpy(k->xanchor[j + c], k->xpos[b], sizeof(double) * 3);
                memcpy(k->xanchor[j + 3 + c], k->xpos[b], sizeof(double) * 3);
            }
        } else {
            memcpy(k->xmat[b], R, sizeof(R));
        }
        matvec3(k->xmat[b], m->ipos[b], t);
        for (int i = 0; i < 3; i++) k->xipos[b][i] = k->xpos[b][i] + t[i];
    }
}

/* is dof j an ancestor-or-self joint of body b? */
static int dof_affects(const OrModel *m, int j, int b) {
    int jb = m->dof_body[j];
    while (b > 0) {
        if (b == jb) return 1;
        b = m->parent[b];
    }
    return 0;
}

/* translational Jacobian of a world point rigidly attached to body b (3 x nv, row-major) and
 * rotational Jacobian of body b */
static void jacobian(const OrModel *m, const Kin *k, int b, const double *point, double *Jp, double *Jr) {
    int nv = m->nv;
    for (int j = 0; j < nv; j++) {
        double col[3] = {0, 0, 0}, ax[3] = {0, 0, 0};
        if (dof_affects(m, j, b)) {
            double r[3] = {point[0] - k->xanchor[j][0], point[1] - k->xanchor[j][1], point[2] - k->xanchor[j][2]};
            if (m->dof_type[j] == 2) {
                memcpy(col, k->xaxis[j], sizeof(col));
            } else {
                cross3(k->xaxis[j], r, col);
                memcpy(ax, k->xaxis[j], sizeof(ax));
            }
        }
        for (int i = 0; i < 3; i++) {
            Jp[i * nv + j] = col[i];
            if (Jr) Jr[i * nv + j] = ax[i];
        }
    }
}

/* world-frame inertia tensor of body b about its COM */
static void world_inertia(const OrModel *m, const Kin *k, int b, double *Iw) {
    double T[9], Rt[9];
    matmul3(k->xmat[b], m->inertia[b], T);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Rt[3 * i + j] = k->xmat[b][3 * j + i];
    matmul3(T, Rt, Iw);
}

/* M(q) = sum_b m_b Jp^T Jp + Jr^T Iw Jr  + diag(armature)   (nv x nv, row-major).  A body's Jacobians have non-zero
 * columns only for the dofs on its path to the root: the products run over that list (the FLOP-counting build,
 * OR_DENSE_JACOBIANS, keeps the full nv x nv loops - same values, the skipped terms are exact zeros - so that the counted
 * figure stays the dense formulation's of the earlier rounds) */
static void mass_matrix(const OrModel *m, const Kin *k, double *M) {
    int nv = m->nv;
    memset(M, 0, sizeof(double) * nv * nv);
    for (int b = 1; b < m->nbody; b++) {
        if (m->mass[b] <= 0) continue;
        double Jp[3 * MAXV], Jr[3 * MAXV], Iw[9], IJ[3 * MAXV];
        int idx[MAXV], n = 0;
        jacobian(m, k, b, k->xipos[b], Jp, Jr);
        world_inertia(m, k, b, Iw);
        for (int j = 0; j < nv; j++) {
#ifndef OR_DENSE_JACOBIANS
            if (!dof_affects(m, j, b)) continue;
#endif
            idx[n++] = j;
        }
        for (int i = 0; i < 3; i++)
            for (int jj = 0; jj < n; jj++) {
                int j = idx[jj];
                double s = 0;
                for (int c = 0; c < 3; c++) s += Iw[3 * i + c] * Jr[c * nv + j];
                IJ[i * nv + j] = s;
            }
        for (int ii = 0; ii < n; ii++)
            for (int jj = 0; jj < n; jj++) {
                int i = idx[ii], j = idx[jj];
                double s = 0;
                for (int c = 0; c < 3; c++) s += m->mass[b] * Jp[c * nv + i] * Jp[c * nv + j] + Jr[c * nv + i] * IJ[c * nv + j];
                M[i * nv + j] += s;
            }
    }
    for (int j = 0; j < nv; j++) M[j * nv + j] += m->armature[j];
}

/* Recursive Newton-Euler in the inertial frame: tau = M(q) qacc + c(q, qvel) (+ gravity).
 * MuJoCo mj_rne; with qacc = NULL it returns the bias force qfrc_bias.
 * A body's joint turns it about an ANCHOR that is fixed in the parent too: the anchor's acceleration follows from the
 * parent's motion, the body origin's from the body's own.  The axes of a ball joint (and of the rotations of a free
 * joint) move with the body: sum_k d/dt(axis_k) v_k = w_body x (w_body - w_parent) = w_parent x sum_k axis_k v_k, which
 * is what MuJoCo's mj_comVel does by taking all three cdof_dot with the velocity BEFORE the joint. */
static void rne(const OrModel *m, const Kin *k, const double *v, const double *a, double *tau) {
    double w[MAXB][3], al[MAXB][3], oacc[MAXB][3];    /* ang. vel, ang. acc, origin acceleration */
    double F[MAXB][3], N[MAXB][3];                    /* net force at COM, net moment about body origin */
    memset(w[0], 0, 24);
    memset(al[0], 0, 24);
    for (int i = 0; i < 3; i++) oacc[0][i] = -m->gravity[i];
    for (int b = 1; b < m->nbody; b++) {
        int p = m->parent[b], j = m->dofid[b], jt = j >= 0 ? m->jtype[b] : 0;
        double r[3], t1[3], t2[3], t3[3], anc[3];
        /* the point the body hangs on: its joint's anchor (hinge, ball), else its origin */
        if (jt == JHINGE || jt == JBALL) memcpy(anc, k->xanchor[j], sizeof(anc));
        else memcpy(anc, k->xpos[b], sizeof(anc));
        for (int i = 0; i < 3; i++) r[i] = anc[i] - k->xpos[p][i];
        cross3(al[p], r, t1);
        cross3(w[p], r, t2);
        cross3(w[p], t2, t3);
        double aacc[3];
        for (int i = 0; i < 3; i++) {
            aacc[i] = oacc[p][i] + t1[i] + t3[i];
            w[b][i] = w[p][i];
            al[b][i] = al[p][i];
        }
        int nd = jt == JFREE ? 6 : (jt == JBALL ? 3 : (jt ? 1 : 0));
        for (int d = 0; d < nd; d++) {
            int jd = j + d;
            double wxa[3];
            cross3(w[p], k->xaxis[jd], wxa);
            if (m->dof_type[jd] == 2) {     /* translation: Coriolis 2 w x (axis v) and axis qacc on the anchor */
                for (int i = 0; i < 3; i++) aacc[i] += 2 * wxa[i] * v[jd] + (a ? k->xaxis[jd][i] * a[jd] : 0.0);
            } else {
                for (int i = 0; i < 3; i++) {
                    w[b][i] += k->xaxis[jd][i] * v[jd];
                    al[b][i] += wxa[i] * v[jd] + (a ? k->xaxis[jd][i] * a[jd] : 0.0);
                }
            }
        }
        /* body origin from the anchor, with the body's own rotation */
        for (int i = 0; i < 3; i++) r[i] = k->xpos[b][i] - anc[i];
        cross3(al[b], r, t1);
        cross3(w[b], r, t2);
        cross3(w[b], t2, t3);
        for (int i = 0; i < 3; i++) oacc[b][i] = aacc[i] + t1[i] + t3[i];
        /* COM acceleration */
        double d[3], cacc[3], Iw[9], Iwv[3], Ial[3], wIw[3];
        for (int i = 0; i < 3; i++) d[i] = k->xipos[b][i] - k->xpos[b][i];
        cross3(al[b], d, t1);
        cross3(w[b], d, t2);
        cross3(w[b], t2, t3);
        for (int i = 0; i < 3; i++) cacc[i] = oacc[b][i] + t1[i] + t3[i];
        world_inertia(m, k, b, Iw);
        matvec3(Iw, w[b], Iwv);
        matvec3(Iw, al[b], Ial);
        cross3(w[b], Iwv, wIw);
        double dxF[3];
        for (int i = 0; i < 3; i++) F[b][i] = m->mass[b] * cacc[i];
        cross3(d, F[b], dxF);
        for (int i = 0; i < 3; i++) N[b][i] = Ial[i] + wIw[i] + dxF[i];
    }
    for (int b = m->nbody - 1; b >= 1; b--) {
        int p = m->parent[b], j = m->dofid[b], jt = j >= 0 ? m->jtype[b] : 0;
        int nd = jt == JFREE ? 6 : (jt == JBALL ? 3 : (jt ? 1 : 0));
        for (int d = 0; d < nd; d++) {
            int jd = j + d;
            if (m->dof_type[jd] == 2) {
                tau[jd] = dot3(k->xaxis[jd], F[b]);
            } else {                        /* moment about the anchor = moment about the origin + (origin - anchor) x F */
                double r[3], rxF[3];
                for (int i = 0; i < 3; i++) r[i] = k->xpos[b][i] - k->xanchor[jd][i];
                cross3(r, F[b], rxF);
                tau[jd] = dot3(k->xaxis[jd], N[b]) + dot3(k->xaxis[jd], rxF);
            }
        }
        if (p > 0) {
            double r[3], rxF[3];
            for (int i = 0; i < 3; i++) r[i] = k->xpos[b][i] - k->xpos[p][i];
            cross3(r, F[b], rxF);
            for (int i = 0; i < 3; i++) {
                F[p][i] += F[b][i];
                N[p][i] += N[b][i] + rxF[i];
            }
        }
    }
}

/* ---------------------------------------------------------------- model compile */
/* MuJoCo compiler, inertiafromgeom="true": geom mass = density * volume; sphere 2/5 m r^2;
 * capsule = cylinder + two hemispheres (MuJoCo user_objects: mjCGeom::SetInertia). */
static void geom_inertia(int type, double r, const double *a, const double *b_, const double *quat, double density,
                         double cap, double *mass, double *pos, double *I) {
    const double PI = 3.14159265358979323846;
    memset(I, 0, sizeof(double) * 9);
    if (type == 3) {            /* box: centre a, half sizes b_, orientation quat (body frame) */
        double R[9], Ib[3], T[9] = {0};
        *mass = density * 8.0 * b_[0] * b_[1] * b_[2];
        memcpy(pos, a, 24);
        for (int i = 0; i < 3; i++) {
            int j1 = (i + 1) % 3, j2 = (i + 2) % 3;
            Ib[i] = (*mass) / 3.0 * (b_[j1] * b_[j1] + b_[j2] * b_[j2]);
        }
        quat2mat(quat, R);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) T[3 * i + j] = R[3 * i + j] * Ib[j];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                double sacc = 0;
                for (int c = 0; c < 3; c++) sacc += T[3 * i + c] * R[3 * j + c];
                I[3 * i + j] = sacc;
            }
    } else if (type == 4) {     /* cylinder between a and b_ (flat ends): m r^2 / 2 about the axis, m (3 r^2 + h^2) / 12 across */
        double u[3] = {b_[0] - a[0], b_[1] - a[1], b_[2] - a[2]};
        double h = sqrt(dot3(u, u));
        for (int i = 0; i < 3; i++) { u[i] /= h; pos[i] = 0.5 * (a[i] + b_[i]); }
        *mass = density * PI * r * r * h;
        double Iax = 0.5 * (*mass) * r * r, Iperp = (*mass) * (3 * r * r + h * h) / 12;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) I[3 * i + j] = (i == j ? Iperp : 0.0) + (Iax - Iperp) * u[i] * u[j];
    } else if (type == 1) {
        *mass = density * 4.0 / 3.0 * PI * r * r * r;
        memcpy(pos, a, 24);
        double i = 0.4 * (*mass) * r * r;
        I[0] = I[4] = I[8] = i;
    } else {
        double u[3] = {b_[0] - a[0], b_[1] - a[1], b_[2] - a[2]};
        double len = sqrt(dot3(u, u));
        for (int i = 0; i < 3; i++) { u[i] /= len; pos[i] = 0.5 * (a[i] + b_[i]); }
        double h = len;     /* cylinder height = 2 * half-length */
        /* cap: the end caps' volume in units of pi r^3 - 1 in MuJoCo 2.0 (which the reference pins; gym's published
         * body masses say so to nine digits), 4/3 from MuJoCo 2.1.2 on; the flat model carries it */
        *mass = density * (PI * r * r * h + cap * PI * r * r * r);
        double ms = (*mass) * 4 * r / (4 * r + 3 * h), mc = (*mass) - ms;
        double Iperp = mc * (3 * r * r + h * h) / 12 + 0.4 * ms * r * r + ms * h * (3 * r + 2 * h) / 8;
        double Iax = mc * r * r / 2 + 0.4 * ms * r * r;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) I[3 * i + j] = (i == j ? Iperp : 0.0) + (Iax - Iperp) * u[i] * u[j];
    }
}

static void set_const(OrModel *m);
/* MuJoCo mj_contactParam [EXT]: solver parameters of a contact between two geoms, each {solref[2], solimp[5], solmix,
 * priority} (entries 24..32 of a geom record; the plane's in the header): the higher priority wins; equal priorities:
 * solref / solimp averaged with weights solmix_a : solmix_b */
static void mix_solver(const double *a, const double *b, double *solref, double *solimp) {
    double w;
    if (a[8] != b[8]) w = a[8] > b[8] ? 1.0 : 0.0;
    else if (a[7] >= MJ_MINVAL && b[7] >= MJ_MINVAL) w = a[7] / (a[7] + b[7]);
    else if (a[7] < MJ_MINVAL && b[7] < MJ_MINVAL) w = 0.5;
    else w = a[7] < MJ_MINVAL ? 0.0 : 1.0;
    /* (priorities differ: the winner's set as it is, w is 0 or 1.)  Equal priorities: solref blended when both are in the
     * standard format, else - direct stiffness / damping, negative - the element-wise minimum (mj_contactParam [EXT]) */
    if (a[8] != b[8] || (a[0] > 0 && b[0] > 0)) { for (int i = 0; i < 2; i++) solref[i] = w * a[i] + (1 - w) * b[i]; }
    else for (int i = 0; i < 2; i++) solref[i] = a[i] < b[i] ? a[i] : b[i];
    for (int i = 0; i < 5; i++) solimp[i] = w * a[2 + i] + (1 - w) * b[2 + i];
}

OrModel *or_model_compile(const double *f, int n) {
    OrModel *m = (OrModel *)calloc(1, sizeof(OrModel));
    int nb = (int)f[0], ng = (int)f[1], nu = (int)f[2], np_ = (int)f[38], ne = (int)f[53], nt = (int)f[54];
    if (n != HEADER_LEN + nb * BODY_STRIDE + ng * GEOM_STRIDE + nu * ACT_STRIDE + np_ * PAIR_STRIDE + ne * EQ_STRIDE +
                 nt * TENDON_STRIDE ||
        nb + 1 > MAXB || ng > MAXG || np_ > MAXP || ne > MAXE || nt > MAXT) {
        free(m);
        return NULL;
    }
    m->nbody = nb + 1;
    m->nu = nu;
    m->timestep = f[3];
    memcpy(m->gravity, f + 4, 24);
    m->frame_skip = (int)f[7];
    memcpy(m->solref, f + 8, 16);
    memcpy(m->solimp, f + 10, 40);
    m->site_body = (int)f[15] + 1;
    memcpy(m->site_pos, f + 16, 24);
    memcpy(m->target_default, f + 19, 24);
    m->has_plane = (int)f[22];
    memcpy(m->plane_pos, f + 23, 24);
    memcpy(m->plane_n, f + 26, 24);
    m->plane_margin = f[29];
    m->plane_gap = f[72];
    m->cone = (int)f[73];
    m->impratio = f[74] > 0 ? f[74] : 1.0;
    m->density = f[30];
    m->viscosity = f[31];
    m->task = (int)f[32];
    m->ctrl_cost = f[33];
    m->obs_skip = (int)f[34];
    double plane_mu = f[35];
    int plane_condim = (int)f[36];
    memcpy(m->solref_l, f + 40, 16);
    memcpy(m->solimp_l, f + 42, 40);
    memcpy(m->site_axis, f + 47, 24);
    memcpy(m->target_dir, f + 50, 24);
    memcpy(m->solref_f, f + 56, 16);
    memcpy(m->solimp_f, f + 58, 40);
    m->dofid[0] = -1;
    int nv = 0, nq = 0;
    int has_inertial[MAXB] = {0};
    for (int b = 1; b <= nb; b++) {
        const double *r = f + HEADER_LEN + (b - 1) * BODY_STRIDE;
        m->parent[b] = (int)r[0] + 1;
        memcpy(m->bpos[b], r + 1, 24);
        quat2mat(r + 4, m->bR0[b]);
        {
            double qn = sqrt(r[4] * r[4] + r[5] * r[5] + r[6] * r[6] + r[7] * r[7]);
            for (int i = 0; i < 4; i++) m->bquat0[b][i] = r[4 + i] / qn;
        }
        m->dofid[b] = -1;
        if (r[8] != 0) {
            int jt = (int)r[8], nd = jt == JFREE ? 6 : (jt == JBALL ? 3 : 1);
            if (nv + nd > MAXV || (jt == JFREE && m->parent[b] != 0)) { free(m); return NULL; }
            m->dofid[b] = nv;
            m->qadr[b] = nq;
            m->jtype[b] = jt;
            double nrm = sqrt(dot3(r + 9, r + 9));
            for (int i = 0; i < 3; i++) m->jaxis[b][i] = nrm > 0 ? r[9 + i] / nrm : 0.0;
            memcpy(m->jpos[b], r + 19, 24);
            for (int d = 0; d < nd; d++) {
                int j = nv + d;
                m->dof_type[j] = (jt == JSLIDE || (jt == JFREE && d < 3)) ? 2 : 1;
                m->dof_body[j] = b;
                m->dof_qadr[j] = nd == 1 ? nq : -1;
                m->damping[j] = r[15];
                m->armature[j] = r[16];
                m->frictionloss[j] = r[22];
                memcpy(m->dof_solref_l[j], r + 40, 16);
                memcpy(m->dof_solimp_l[j], r + 42, 40);
                memcpy(m->dof_solref_f[j], r + 47, 16);
                memcpy(m->dof_solimp_f[j], r + 49, 40);
            }
            if (jt == JBALL) {
                m->ball_limited[b] = (int)r[14];
                m->ball_range[b] = r[12] > r[13] ? r[12] : r[13];
            }
            if (nd == 1) {
                m->stiffness[nv] = r[17];
                m->springref[nv] = r[18];
                m->range[nv][0] = r[12];
                m->range[nv][1] = r[13];
                m->limited[nv] = (int)r[14];
                m->jmargin[nv] = r[54];
                m->qpos0[nq] = r[55];           /* MJCF joint ref: the kinematics turn / move by qpos - qpos0 */
            }
            /* qpos0: zero (hinge, slide), the identity (ball), the body's own pose (free) */
            if (jt == JBALL) m->qpos0[nq] = 1.0;
            if (jt == JFREE) {
                memcpy(m->qpos0 + nq, m->bpos[b], 24);
                memcpy(m->qpos0 + nq + 3, m->bquat0[b], 32);
            }
            nv += nd;
            nq += jt == JFREE ? 7 : (jt == JBALL ? 4 : 1);
        }
        if (r[23] != 0) {               /* explicit <inertial>: mass, centre, full tensor in the body frame's axes */
            has_inertial[b] = 1;
            m->mass[b] = r[24];
            memcpy(m->ipos[b], r + 25, 24);
            memcpy(m->inertia[b], r + 28, 72);
        }
    }
    m->nv = nv;
    m->nq = nq;
    /* inertiafromgeom */
    double gm[MAXG], gp[MAXG][3], gI[MAXG][9];
    int gb[MAXG];
    const double *g0 = f + HEADER_LEN + nb * BODY_STRIDE;
    for (int g = 0; g < ng; g++) {
        const double *r = g0 + g * GEOM_STRIDE;
        gb[g] = (int)r[0] + 1;              /* 0: a static geom of the world body */
        geom_inertia((int)r[1], r[2], r + 3, r + 6, r + 14, r[9], f[37], &gm[g], gp[g], gI[g]);
        /* colliding geoms: a sphere, a capsule = its two end spheres, "to" end first (mjc_PlaneCapsule tests
         * pos + axis * halflength, then pos - axis * halflength, and aligns the contact frame with the axis), a box = its
         * eight corners (mjc_PlaneBox; MuJoCo keeps at most four of them, this restatement all that are within the margin).
         * Contact friction / condim = max over the two geoms (MuJoCo mj_contactParam, equal priorities). */
        int gt = (int)r[1];
        int ends = (r[10] != 0 && gb[g] > 0) ? (gt == 1 ? 1 : (gt == 2 ? 2 : (gt == 3 ? 8 : 4))) : 0;
        double Rb[9];
        if (gt == 3) quat2mat(r + 14, Rb);
        for (int e = 0; e < ends; e++) {
            if (m->nsphere >= MAXS) { free(m); return NULL; }
            int s = m->nsphere++;
            m->sph_body[s] = gb[g];
            m->sph_k[s] = e;
            if (gt == 3) {
                double c[3] = {(e & 1 ? 1 : -1) * r[6], (e & 2 ? 1 : -1) * r[7], (e & 4 ? 1 : -1) * r[8]}, t[3];
                matvec3(Rb, c, t);
                for (int i = 0; i < 3; i++) m->sph_pos[s][i] = r[3 + i] + t[i];
                m->sph_r[s] = 0.0;
                m->sph_kind[s] = 1;
                memcpy(m->sph_ctr[s], r + 3, 24);
            } else if (gt == 4) {
                double u[3] = {r[6] - r[3], r[7] - r[4], r[8] - r[5]}, len = sqrt(dot3(u, u));
                for (int i = 0; i < 3; i++) {
                    m->sph_axis[s][i] = u[i] / len;
                    m->sph_ctr[s][i] = 0.5 * (r[3 + i] + r[6 + i]);
                    m->sph_pos[s][i] = m->sph_ctr[s][i];
                }
                m->sph_r[s] = r[2];
                m->sph_hh[s] = 0.5 * len;
                m->sph_kind[s] = 2;
            } else {
                memcpy(m->sph_pos[s], (gt == 2 && e == 0) ? r + 6 : r + 3, 24);
                m->sph_r[s] = r[2];
            }
            m->sph_margin[s] = r[11];
            m->sph_gap[s] = r[18];
            double mu = r[12] > plane_mu ? r[12] : plane_mu;
            int condim = (int)r[13] > plane_condim ? (int)r[13] : plane_condim;
            m->sph_mu[s] = condim >= 3 ? mu : 0.0;
            mix_solver(r + 24, f + 63, m->sph_solref[s], m->sph_solimp[s]);
            if (gt == 2) {
                double u[3] = {r[6] - r[3], r[7] - r[4], r[8] - r[5]}, len = sqrt(dot3(u, u));
                for (int i = 0; i < 3; i++) m->sph_axis[s][i] = u[i] / len;
            }
        }
    }
    for (int b = 1; b <= nb; b++) {
        if (has_inertial[b]) continue;
        double mass = 0, com[3] = {0, 0, 0};
        for (int g = 0; g < ng; g++)
            if (gb[g] == b) {
                mass += gm[g];
                for (int i = 0; i < 3; i++) com[i] += gm[g] * gp[g][i];
            }
        m->mass[b] = mass;
        if (mass > 0)
            for (int i = 0; i < 3; i++) com[i] /= mass;
        memcpy(m->ipos[b], com, 24);
        for (int g = 0; g < ng; g++)
            if (gb[g] == b) {
                double d[3] = {gp[g][0] - com[0], gp[g][1] - com[1], gp[g][2] - com[2]};
                double dd = dot3(d, d);
                for (int i = 0; i < 3; i++)
                    for (int j = 0; j < 3; j++)
                        m->inertia[b][3 * i + j] += gI[g][3 * i + j] + gm[g] * ((i == j ? dd : 0.0) - d[i] * d[j]);
            }
    }
    const double *a0 = g0 + ng * GEOM_STRIDE;
    for (int a = 0; a < nu; a++) {
        m->act_dof[a] = (int)a0[a * ACT_STRIDE];
        m->gear[a] = a0[a * ACT_STRIDE + 1];
        m->ctrl_lo[a] = a0[a * ACT_STRIDE + 2];
        m->ctrl_hi[a] = a0[a * ACT_STRIDE + 3];
        m->act_tendon[a] = (int)a0[a * ACT_STRIDE + 5];
        m->act_gain[a] = a0[a * ACT_STRIDE + 6];
        for (int i = 0; i < 3; i++) m->act_bias[a][i] = a0[a * ACT_STRIDE + 7 + i];
        m->ctrllimited[a] = a0[a * ACT_STRIDE + 10] != 0.0;
        m->forcelimited[a] = a0[a * ACT_STRIDE + 11] != 0.0;
        m->forcerange[a][0] = a0[a * ACT_STRIDE + 12];
        m->forcerange[a][1] = a0[a * ACT_STRIDE + 13];
    }
    /* geom-geom pairs: spheres and capsules are segments (from, to - from) with a radius; ONE geom of a pair may be a box
     * (against a sphere); friction / condim / margin of a contact = the larger of the two geoms' (MuJoCo mj_contactParam
     * with equal priorities) */
    const double *p0 = a0 + nu * ACT_STRIDE;
    m->npair = np_;
    for (int k = 0; k < np_; k++) {
        /* (condim, friction, margin, solref, solimp of the pair's contacts arrive resolved: mj_contactParam's mixing of the
         * two geoms' values, or what the model's <pair> element says - RawModel.pair_contact) */
        const double *pr = p0 + k * PAIR_STRIDE;
        m->pair_box[k] = -1;
        m->pair_cyl[k] = -1;
        for (int e = 0; e < 2; e++) {
            const double *r = g0 + (int)pr[e] * GEOM_STRIDE;
            m->pair_body[k][e] = (int)r[0] + 1;
            memcpy(m->pair_a[k][e], r + 3, 24);
            for (int i = 0; i < 3; i++) m->pair_d[k][e][i] = (int)r[1] == 2 ? r[6 + i] - r[3 + i] : 0.0;
            m->pair_r[k][e] = (int)r[1] == 3 ? 0.0 : r[2];
            if ((int)r[1] == 4) {               /* a cylinder: against a sphere or a capsule (round 5; a scheme of its own, cyl_point) */
                if (m->pair_cyl[k] >= 0) { free(m); return NULL; }      /* (not against another cylinder) */
                m->pair_cyl[k] = e;
                m->pair_cyl_r[k] = r[2];
                m->pair_r[k][e] = 0.0;
                for (int i = 0; i < 3; i++) m->pair_d[k][e][i] = r[6 + i] - r[3 + i];
            }
            if ((int)r[1] == 3) {
                if (m->pair_box[k] >= 0) {          /* the second box of a box-box pair */
                    m->pair_box[k] = 2;
                    quat2mat(r + 14, m->pair_R2[k]);
                    memcpy(m->pair_half2[k], r + 6, 24);
                } else {
                    m->pair_box[k] = e;
                    quat2mat(r + 14, m->pair_R[k]);
                    memcpy(m->pair_half[k], r + 6, 24);
                }
            }
        }
        m->pair_margin[k] = pr[4];
        m->pair_mu[k] = (int)pr[2] >= 3 ? pr[3] : 0.0;
        memcpy(m->pair_solref[k], pr + 5, 16);
        memcpy(m->pair_solimp[k], pr + 7, 40);
        if (m->pair_cyl[k] >= 0 && m->pair_box[k] >= 0) { free(m); return NULL; }       /* (nor against a box) */
    }
    const double *e0 = p0 + np_ * PAIR_STRIDE;
    m->neq = ne;
    for (int k = 0; k < ne; k++) {
        const double *r = e0 + k * EQ_STRIDE;
        m->eq_type[k] = (int)r[0];
        if (m->eq_type[k] < 1 || m->eq_type[k] > 3) { free(m); return NULL; }
        m->eq_o1[k] = (int)r[1] + (m->eq_type[k] == 3 ? 0 : 1);        /* bodies: 0 = world; dofs: -1 = none */
        m->eq_o2[k] = (int)r[2] + (m->eq_type[k] == 3 ? 0 : 1);
        memcpy(m->eq_anchor[k][0], r + 3, 24);
        memcpy(m->eq_poly[k], r + 6, 40);
        memcpy(m->eq_solref[k], r + 11, 16);
        memcpy(m->eq_solimp[k], r + 13, 40);
    }
    const double *t0 = e0 + ne * EQ_STRIDE;
    m->ntendon = nt;
    for (int k = 0; k < nt; k++) {
        const double *r = t0 + k * TENDON_STRIDE;
        m->tn_n[k] = (int)r[0];
        m->tn_limited[k] = (int)r[1];
        m->tn_range[k][0] = r[2];
        m->tn_range[k][1] = r[3];
        m->tn_margin[k] = r[4];
        memcpy(m->tn_solref[k], r + 8 + 2 * MAXTJ, 16);
        memcpy(m->tn_solimp[k], r + 10 + 2 * MAXTJ, 40);
        for (int i = 0; i < m->tn_n[k]; i++) {
            m->tn_dof[k][i] = (int)r[8 + 2 * i];
            m->tn_coef[k][i] = r[9 + 2 * i];
        }
    }
    set_const(m);
    return m;
}

void or_model_free(OrModel *m) { free(m); }

/* MuJoCo mj_setConst (set0): at qpos0, dof_invweight0[j] = (M^-1)_jj for hinge dofs,
 * body_invweight0[b] (translational) = trace(Jcom M^-1 Jcom^T) / 3, world = 0. */
/* eigen-decomposition of a symmetric 3x3 (cyclic Jacobi): A = V diag(w) V^T, columns of V = axes */
static void eig3(const double *A, double *w, double *V) {
    double a[9];
    memcpy(a, A, sizeof(a));
    static const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    memcpy(V, I3, sizeof(I3));
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = fabs(a[1]) + fabs(a[2]) + fabs(a[5]);
        if (off <= 1e-300) break;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                double apq = a[3 * p + q];
                if (apq == 0) continue;
                double th = (a[3 * q + q] - a[3 * p + p]) / (2 * apq);
                double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1));
                double c = 1 / sqrt(t * t + 1), sn = t * c;
                for (int k = 0; k < 3; k++) {       /* A <- A G */
                    double akp = a[3 * k + p], akq = a[3 * k + q];
                    a[3 * k + p] = c * akp - sn * akq;
                    a[3 * k + q] = sn * akp + c * akq;
                }
                for (int k = 0; k < 3; k++) {       /* A <- G^T A */
                    double apk = a[3 * p + k], aqk = a[3 * q + k];
                    a[3 * p + k] = c * apk - sn * aqk;
                    a[3 * q + k] = sn * apk + c * aqk;
                }
                for (int k = 0; k < 3; k++) {
                    double vkp = V[3 * k + p], vkq = V[3 * k + q];
                    V[3 * k + p] = c * vkp - sn * vkq;
                    V[3 * k + q] = sn * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < 3; i++) w[i] = a[4 * i];
}

static void clamp_solimp(double *si) {         /* MuJoCo getsolparam: mjMINIMP = 1e-4, mjMAXIMP = 0.9999 */
    for (int i = 0; i < 2; i++) { if (si[i] < 1e-4) si[i] = 1e-4; if (si[i] > 0.9999) si[i] = 0.9999; }
    if (si[2] < 0) si[2] = 0;
    if (si[3] < 1e-4) si[3] = 1e-4;
    if (si[3] > 0.9999) si[3] = 0.9999;
    if (si[4] < 1) si[4] = 1;
}

/* inertial frame of body b: principal axes of its inertia tensor (the body frame when it is diagonal there) and the box
 * of equal inertia, MuJoCo mj_passive: box_i = sqrt(6 (I_j + I_k - I_i) / m); MuJoCo evaluates the box from the current
 * body_mass / body_inertia, so run-time edits (dynamics randomization) refresh it */
static void body_box(OrModel *m, int b) {
    {
        double w[3], tr = m->inertia[b][0] + m->inertia[b][4] + m->inertia[b][8];
        double off = fabs(m->inertia[b][1]) + fabs(m->inertia[b][2]) + fabs(m->inertia[b][5]);
        static const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        if (off <= 1e-12 * tr) {
            memcpy(m->iR[b], I3, sizeof(I3));
            for (int i = 0; i < 3; i++) w[i] = m->inertia[b][4 * i];
        } else {
            eig3(m->inertia[b], w, m->iR[b]);
        }
        for (int i = 0; i < 3; i++) {
            double x = w[(i + 1) % 3] + w[(i + 2) % 3] - w[i];
            if (x < MJ_MINVAL) x = MJ_MINVAL;
            m->ibox[b][i] = m->mass[b] > MJ_MINVAL ? sqrt(x / m->mass[b] * 6.0) : 0.0;
        }
    }
}

static void set_const(OrModel *m) {
    int nv = m->nv;
    clamp_solimp(m->solimp);
    clamp_solimp(m->solimp_l);
    clamp_solimp(m->solimp_f);
    for (int e = 0; e < m->neq; e++) clamp_solimp(m->eq_solimp[e]);
    for (int j = 0; j < m->nv; j++) { clamp_solimp(m->dof_solimp_l[j]); clamp_solimp(m->dof_solimp_f[j]); }
    for (int s = 0; s < m->nsphere; s++) clamp_solimp(m->sph_solimp[s]);
    for (int k = 0; k < m->npair; k++) clamp_solimp(m->pair_solimp[k]);
    for (int t = 0; t < m->ntendon; t++) clamp_solimp(m->tn_solimp[t]);
    /* inertial frames: principal axes of every body's inertia tensor (body frame when it is diagonal there) and
     * the box of equal inertia, MuJoCo mj_passive: box_i = sqrt(6 (I_j + I_k - I_i) / m) */
    for (int b = 1; b < m->nbody; b++) body_box(m, b);
    double M[MAXV * MAXV];
    Kin k;
    kinematics(m, m->qpos0, &k);
    mass_matrix(m, &k, M);
    chol(M, nv);
    for (int j = 0; j < nv; j++) {
        double e[MAXV] = {0};
        e[j] = 1;
        chol_solve(M, nv, e);
        m->dof_invweight0[j] = e[j];
    }
    /* MuJoCo averages the entries of a ball joint, and of the translations and of the rotations of a free joint */
    for (int b = 1; b < m->nbody; b++) {
        int j = m->dofid[b];
        if (j < 0 || (m->jtype[b] != JBALL && m->jtype[b] != JFREE)) continue;
        for (int g = 0; g < (m->jtype[b] == JFREE ? 2 : 1); g++) {
            double avg = (m->dof_invweight0[j + 3 * g] + m->dof_invweight0[j + 3 * g + 1] + m->dof_invweight0[j + 3 * g + 2]) / 3;
            for (int c = 0; c < 3; c++) m->dof_invweight0[j + 3 * g + c] = avg;
        }
    }
    m->body_invweight0[0] = m->body_invweight0r[0] = 0;
    for (int b = 1; b < m->nbody; b++) {
        double Jp[3 * MAXV], Jr[3 * MAXV], tr = 0, trr = 0;
        jacobian(m, &k, b, k.xipos[b], Jp, Jr);
        for (int i = 0; i < 3; i++) {
            double x[MAXV];
            memcpy(x, Jp + i * nv, sizeof(double) * nv);
            chol_solve(M, nv, x);
            for (int j = 0; j < nv; j++) tr += Jp[i * nv + j] * x[j];
            memcpy(x, Jr + i * nv, sizeof(double) * nv);
            chol_solve(M, nv, x);
            for (int j = 0; j < nv; j++) trr += Jr[i * nv + j] * x[j];
        }
        m->body_invweight0[b] = tr / 3;
        m->body_invweight0r[b] = trr / 3;
    }
    for (int t = 0; t < m->ntendon; t++) {       /* tendon_invweight0 = J M0^-1 J' */
        double x[MAXV] = {0}, y[MAXV], acc = 0;
        for (int i = 0; i < m->tn_n[t]; i++) x[m->tn_dof[t][i]] += m->tn_coef[t][i];
        memcpy(y, x, sizeof(x));
        chol_solve(M, nv, y);
        for (int j = 0; j < nv; j++) acc += x[j] * y[j];
        m->tn_invweight0[t] = acc;
    }
    /* equality constraints at qpos0 (MuJoCo's compiler): a connect's anchor as seen from body 2; a weld's shared point is
     * body 2's origin (seen from body 1) and its relative orientation R1' R2 */
    for (int e = 0; e < m->neq; e++) {
        if (m->eq_type[e] == 3) continue;
        int b1 = m->eq_o1[e], b2 = m->eq_o2[e];
        double w1[3], d[3];
        if (m->eq_type[e] == 2) {
            for (int i = 0; i < 3; i++) d[i] = k.xpos[b2][i] - k.xpos[b1][i];
            for (int i = 0; i < 3; i++)
                m->eq_anchor[e][0][i] = k.xmat[b1][i] * d[0] + k.xmat[b1][3 + i] * d[1] + k.xmat[b1][6 + i] * d[2];
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) {
                    double sacc = 0;
                    for (int c = 0; c < 3; c++) sacc += k.xmat[b1][3 * c + i] * k.xmat[b2][3 * c + j];
                    m->eq_relR[e][3 * i + j] = sacc;
                }
        }
        matvec3(k.xmat[b1], m->eq_anchor[e][0], w1);
        for (int i = 0; i < 3; i++) d[i] = k.xpos[b1][i] + w1[i] - k.xpos[b2][i];
        for (int i = 0; i < 3; i++)
            m->eq_anchor[e][1][i] = k.xmat[b2][i] * d[0] + k.xmat[b2][3 + i] * d[1] + k.xmat[b2][6 + i] * d[2];
    }
}

/* ---------------------------------------------------------------- soft constraints */
/* MuJoCo mj_makeImpedance / getimpedance / mj_referenceConstraint for one scalar row.
 * pos: signed distance, margin: activation margin; r = pos - margin. */
static void row_params_set(const OrModel *m, const double *solref, const double *solimp, double pos, double margin,
                           double diagApprox, double jv, double *D, double *aref) {
    double dmin = solimp[0], dmax = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
    double r = pos - margin, imp;
    if (dmin == dmax || width <= MJ_MINVAL) {
        imp = 0.5 * (dmin + dmax);
    } else {
        double x = fabs(r) / width;
        if (x >= 1) imp = dmax;
        else if (x <= 0) imp = dmin;
        else {
            double y;
            if (power == 1) y = x;
            else if (x <= mid) y = pow(x, power) / pow(mid, power - 1);
            else y = 1 - pow(1 - x, power) / pow(1 - mid, power - 1);
            imp = dmin + y * (dmax - dmin);
        }
    }
    double R = (1 - imp) / imp * diagApprox;
    if (R < MJ_MINVAL) R = MJ_MINVAL;
    *D = 1 / R;
    /* standard solref = (timeconst, dampratio); refsafe: timeconst >= 2*timestep.  Both entries negative (round 5): the DIRECT
     * format (-stiffness, -damping) of mj_makeImpedance [EXT] - k = -solref[0] / dmax^2, b = -solref[1] / dmax, no refsafe */
    double tc = solref[0], dr = solref[1], b, kk;
    if (tc > 0) {
        if (tc < 2 * m->timestep) tc = 2 * m->timestep;
        b = 2 / (dmax * tc);
        kk = 1 / (dmax * dmax * tc * tc * dr * dr);
    } else {
        kk = -solref[0] / (dmax * dmax);
        b = -solref[1] / dmax;
    }
    *aref = -b * jv - kk * imp * r;
}
/* minimise  1/2 (a - a_s)^T M (a - a_s) + sum_i s_i(J_i a - aref_i)
 * (MuJoCo's primal problem, mj_constraintUpdate): per row kind the cost s(r) is
 *   ROW_UNI   limits, frictionless and pyramidal contacts   1/2 D min(0, r)^2
 *   ROW_EQ    equality                                      1/2 D r^2
 *   ROW_FRIC  friction loss f                                1/2 D r^2 for |r| < R f, else f |r| - 1/2 R f^2   (R = 1 / D)
 * - convex and piecewise quadratic; the Newton solver of mj_fwdConstraint converges to this unique minimiser up to its
 * 1e-8 tolerance.  Here: Newton + exact line search, run to machine precision.  fs = M a_s.  Returns qacc in a and row
 * forces -s'(r) in force. */
static double row_slope(int kind, double D, double fl, double r, double *curv) {     /* s'(r), and s''(r) in curv */
    if (kind == ROW_EQ) { *curv = D; return D * r; }
    if (kind == ROW_FRIC) {
        if (D * r <= -fl) { *curv = 0; return -fl; }
        if (D * r >= fl) { *curv = 0; return fl; }
        *curv = D;
        return D * r;
    }
    if (r < 0) { *curv = D; return D * r; }
    *curv = 0;
    return 0;
}
/* One ELLIPTIC-cone contact of condim 3 (MuJoCo mj_constraintUpdate / PrimalUpdateConstraint, mjCNSTR_CONTACT_ELLIPTIC
 * [EXT]): residuals r = (normal, tangent 1, tangent 2), D0 = 1 / R of the normal row, Dt of the friction rows, friction
 * coefficient fr.  With mu = fr sqrt(D0 / Dt) (MuJoCo's contact.mu, the regularised friction) and U = (mu r0, fr r1, fr r2),
 * N = U0, T = |(U1, U2)|, the weighted norm D0 r0^2 + Dt (r1^2 + r2^2) = D0 / mu^2 (N^2 + T^2) is isotropic in U and the
 * cost is D0 / (2 mu^2) times the squared distance of U from the cone N >= mu T:
 *   top zone     N >= mu T                 0                                   (the dual cone: no force)
 *   bottom zone  mu N + T <= 0             1/2 (D0 r0^2 + Dt (r1^2 + r2^2))    (the polar cone: all three rows quadratic)
 *   middle zone  else                      1/2 Dm (N - mu T)^2,  Dm = D0 / (mu^2 (1 + mu^2))
 * Returns the zone (0 / 1 / 2), the gradient in grad[3] and - when H is given - the Hessian in H[9] (both w.r.t. r). */
static int cone_eval(double D0, double Dt, double fr, const double *r, double *grad, double *H, double *cost) {
    double mu = fr * sqrt(D0 / Dt);
    double N = mu * r[0], U1 = fr * r[1], U2 = fr * r[2], T = sqrt(U1 * U1 + U2 * U2);
    if (H) memset(H, 0, 72);
    if (N >= mu * T || (T <= 0 && N >= 0)) {
        grad[0] = grad[1] = grad[2] = 0;
        if (cost) *cost = 0;
        return 0;
    }
    if (mu * N + T <= 0 || (T <= 0 && N < 0)) {
        grad[0] = D0 * r[0]; grad[1] = Dt * r[1]; grad[2] = Dt * r[2];
        if (H) { H[0] = D0; H[4] = Dt; H[8] = Dt; }
        if (cost) *cost = 0.5 * (D0 * r[0] * r[0] + Dt * (r[1] * r[1] + r[2] * r[2]));
        return 1;
    }
    double Dm = D0 / (mu * mu * (1 + mu * mu)), NmT = N - mu * T;
    double u[2] = {U1 / T, U2 / T};
    grad[0] = Dm * NmT * mu;
    grad[1] = -Dm * NmT * mu * fr * u[0];
    grad[2] = -Dm * NmT * mu * fr * u[1];
    if (cost) *cost = 0.5 * Dm * NmT * NmT;
    if (H) {
        H[0] = Dm * mu * mu;
        for (int k = 0; k < 2; k++) {
            H[1 + k] = H[3 * (1 + k)] = -Dm * mu * mu * fr * u[k];
            for (int l = 0; l < 2; l++)
                H[3 * (1 + k) + 1 + l] = Dm * mu * mu * fr * fr * u[k] * u[l] - Dm * NmT * mu * fr * fr * ((k == l ? 1.0 : 0.0) - u[k] * u[l]) / T;
        }
    }
    return 2;
}
/* phi'(alpha) and phi''(alpha) of the rows' cost along jar + alpha jd (all kinds, evaluated directly) */
static void rows_dphi(int nc, const double *D, const int *kind, const double *floss, const double *jar, const double *jd, double alpha,
                      double *d1, double *d2) {
    double s1 = 0, s2 = 0;
    for (int c = 0; c < nc; c++) {
        if (kind[c] == ROW_ELLT) continue;
        if (kind[c] == ROW_ELL) {
            double r[3], g[3], H[9];
            for (int k = 0; k < 3; k++) r[k] = jar[c + k] + alpha * jd[c + k];
            cone_eval(D[c], D[c + 1], floss[c], r, g, H, NULL);
            for (int k = 0; k < 3; k++) {
                s1 += g[k] * jd[c + k];
                for (int l = 0; l < 3; l++) s2 += H[3 * k + l] * jd[c + k] * jd[c + l];
            }
            continue;
        }
        double curv, sl = row_slope(kind[c], D[c], floss[c], jar[c] + alpha * jd[c], &curv);
        s1 += sl * jd[c];
        s2 += curv * jd[c] * jd[c];
    }
    *d1 = s1;
    *d2 = s2;
}
static void solve_rows(OrModel *m, int nv, const double *M, const double *fs, int nc,
                       double (*J)[MAXV], const double *aref, const double *D, const int *kind, const double *floss,
                       double *a, double *force) {
    double L[MAXV * MAXV];
    memcpy(L, M, sizeof(double) * nv * nv);
    chol(L, nv);
    memcpy(a, fs, sizeof(double) * nv);
    chol_solve(L, nv, a);
    double scale = 1;
    for (int i = 0; i < nv; i++) if (fabs(fs[i]) > scale) scale = fabs(fs[i]);
    m->newton_calls++;
#ifdef OR_NO_POLISH        /* the FLOP-counting build tallies the algorithm, not the oracle's extra polishing step */
    int it, polished = 1;
#else
    int it, polished = 0;
#endif
    int any_cone = 0;
    for (it = 0; it < 100 && nc > 0; it++) {
        double jar[MAXC], g[MAXV], H[MAXV * MAXV], d[MAXV];
        for (int i = 0; i < nv; i++) {
            double s = -fs[i];
            for (int j = 0; j < nv; j++) s += M[i * nv + j] * a[j];
            g[i] = s;
        }
        memcpy(H, M, sizeof(double) * nv * nv);
        for (int c = 0; c < nc; c++) {
            double s = -aref[c];
            for (int j = 0; j < nv; j++) s += J[c][j] * a[j];
            jar[c] = s;
            if (kind[c] == ROW_ELL || kind[c] == ROW_ELLT) {
                any_cone = 1;               /* (handled below, once the three residuals of the contact are known) */
            } else if (kind[c] == ROW_UNI) {           /* (the arithmetic of the earlier rounds, kept as it was) */
                if (s < 0) {
                    for (int i = 0; i < nv; i++) {
                        g[i] += D[c] * s * J[c][i];
                        for (int j = 0; j < nv; j++) H[i * nv + j] += D[c] * J[c][i] * J[c][j];
                    }
                }
            } else {
                double curv, sl = row_slope(kind[c], D[c], floss[c], s, &curv);
                for (int i = 0; i < nv; i++) {
                    if (J[c][i] == 0) continue;
                    g[i] += sl * J[c][i];
                    if (curv != 0)
                        for (int j = 0; j < nv; j++) H[i * nv + j] += curv * J[c][i] * J[c][j];
                }
            }
        }
        for (int c0 = 0; c0 < nc && any_cone; c0++) {
            if (kind[c0] != ROW_ELL) continue;
            double gr[3], Hc[9];
            cone_eval(D[c0], D[c0 + 1], floss[c0], jar + c0, gr, Hc, NULL);
            for (int kk = 0; kk < 3; kk++)
                for (int i = 0; i < nv; i++) {
                    if (J[c0 + kk][i] == 0) continue;
                    g[i] += gr[kk] * J[c0 + kk][i];
                    for (int ll = 0; ll < 3; ll++) {
                        if (Hc[3 * kk + ll] == 0) continue;
                        for (int j = 0; j < nv; j++) H[i * nv + j] += Hc[3 * kk + ll] * J[c0 + kk][i] * J[c0 + ll][j];
                    }
                }
        }
        double gn = 0;
        for (int i = 0; i < nv; i++) if (fabs(g[i]) > gn) gn = fabs(g[i]);
        /* converged: one more Newton step with the final active set lands on its exact minimiser (the problem is
         * piecewise quadratic), so that the oracle's own stopping tolerance stays out of parity comparisons */
        if (gn <= 1e-11 * scale) {
            if (polished) break;
            polished = 1;
        }
        chol(H, nv);
        for (int i = 0; i < nv; i++) d[i] = -g[i];
        chol_solve(H, nv, d);
        /* exact line search: phi'(alpha) = p0 + alpha p1 + sum_c s_c'(jar_c + alpha jd_c) jd_c, piecewise linear and
         * increasing: c0 + alpha c1 on the current piece; every break point carries the change of (c0, c1) across it */
        double p0 = 0, p1 = 0, bp[2 * MAXC], dc0[2 * MAXC], dc1[2 * MAXC];
        int nbp = 0;
        for (int i = 0; i < nv; i++) {
            double Md = 0, Ma = -fs[i];
            for (int j = 0; j < nv; j++) { Md += M[i * nv + j] * d[j]; Ma += M[i * nv + j] * a[j]; }
            p0 += d[i] * Ma;
            p1 += d[i] * Md;
        }
        if (any_cone) {
            /* elliptic cones: phi' is increasing and continuous but not piecewise linear - its root by Newton's iteration
             * kept inside a bracket (bisection when a step leaves it), to the resolution of the arithmetic */
            double jd[MAXC], d1, d2;
            for (int c = 0; c < nc; c++) {
                double s = 0;
                for (int j = 0; j < nv; j++) s += J[c][j] * d[j];
                jd[c] = s;
            }
            rows_dphi(nc, D, kind, floss, jar, jd, 0.0, &d1, &d2);
            double f0 = p0 + d1;
            if (!(f0 < 0)) break;                   /* (no descent left: at the minimiser to rounding) */
            double lo = 0, hi = -1, alpha = 1;
            for (int ls = 0; ls < 60; ls++) {
                rows_dphi(nc, D, kind, floss, jar, jd, alpha, &d1, &d2);
                double f = p0 + alpha * p1 + d1, fp = p1 + d2;
                if (fabs(f) <= 1e-15 * fabs(f0)) break;
                if (f < 0) lo = alpha; else hi = alpha;
                double an = alpha - f / fp;
                if (!(fp > 0) || !(an > lo) || (hi > 0 && !(an < hi))) an = hi > 0 ? 0.5 * (lo + hi) : 2 * alpha;
                if (hi > 0 && hi - lo <= 1e-16 * hi) break;
                alpha = an;
            }
            for (int i = 0; i < nv; i++) a[i] += alpha * d[i];
            continue;
        }
        double c0 = p0, c1 = p1;
        for (int c = 0; c < nc; c++) {
            double s = 0;
            for (int j = 0; j < nv; j++) s += J[c][j] * d[j];
            if (kind[c] == ROW_UNI) {
                if (jar[c] < 0 || (jar[c] == 0 && s < 0)) { c0 += D[c] * s * jar[c]; c1 += D[c] * s * s; }
                if (s != 0) {
                    double al = -jar[c] / s;
                    if (al > 0) {
                        /* row c toggles at al: it was active iff jar < 0 (or jar == 0 moving down) */
                        double sgn = jar[c] < 0 ? -1.0 : 1.0;
                        bp[nbp] = al; dc0[nbp] = sgn * D[c] * s * jar[c]; dc1[nbp] = sgn * D[c] * s * s; nbp++;
                    }
                }
            } else if (kind[c] == ROW_EQ) {
                c0 += D[c] * s * jar[c];
                c1 += D[c] * s * s;
            } else {
                /* friction loss: zones r <= -R f (slope -f), |r| < R f (D r), r >= R f (+f); in zone z the row adds
                 * (z0[z], z1[z]) to (c0, c1) */
                double Rf = floss[c] / D[c];
                double z0[3] = {-floss[c] * s, D[c] * s * jar[c], floss[c] * s}, z1[3] = {0, D[c] * s * s, 0};
                int z = jar[c] <= -Rf ? 0 : (jar[c] >= Rf ? 2 : 1);
                if (s > 0 && jar[c] == -Rf) z = 1;          /* on a border, moving inwards */
                if (s < 0 && jar[c] == Rf) z = 1;
                c0 += z0[z];
                c1 += z1[z];
                if (s != 0) {
                    int dir = s > 0 ? 1 : -1;
                    for (int zz = z; zz + dir >= 0 && zz + dir <= 2; zz += dir) {
                        double border = (dir > 0 ? (zz == 0 ? -Rf : Rf) : (zz == 2 ? Rf : -Rf));
                        double al = (border - jar[c]) / s;
                        if (al > 0) {
                            bp[nbp] = al; dc0[nbp] = z0[zz + dir] - z0[zz]; dc1[nbp] = z1[zz + dir] - z1[zz]; nbp++;
                        }
                    }
                }
            }
        }
        for (int i = 1; i < nbp; i++)       /* insertion sort of break points */
            for (int j = i; j > 0 && bp[j] < bp[j - 1]; j--) {
                double t = bp[j]; bp[j] = bp[j - 1]; bp[j - 1] = t;
                t = dc0[j]; dc0[j] = dc0[j - 1]; dc0[j - 1] = t;
                t = dc1[j]; dc1[j] = dc1[j - 1]; dc1[j - 1] = t;
            }
        if (!(c1 > 0)) break;               /* a zero step (the polishing iteration from an exact minimiser): done */
        double alpha = -c0 / c1;
        for (int i = 0; i < nbp; i++) {
            if (alpha <= bp[i]) break;
            c0 += dc0[i];
            c1 += dc1[i];
            alpha = -c0 / c1;
            if (alpha < bp[i]) alpha = bp[i];
        }
        for (int i = 0; i < nv; i++) a[i] += alpha * d[i];
    }
    m->newton_iters += it;
    if (it >= 100) m->newton_fail++;
    for (int c = 0; c < nc; c++) {
        double s = -aref[c], curv;
        for (int j = 0; j < nv; j++) s += J[c][j] * a[j];
        if (kind[c] == ROW_UNI) force[c] = s < 0 ? -D[c] * s : 0.0;
        else if (kind[c] == ROW_ELL || kind[c] == ROW_ELLT) force[c] = s;       /* (the residual for now: see below) */
        else force[c] = -row_slope(kind[c], D[c], floss[c], s, &curv);
    }
    for (int c = 0; c < nc; c++)
        if (kind[c] == ROW_ELL) {
            double r[3] = {force[c], force[c + 1], force[c + 2]}, gr[3];
            cone_eval(D[c], D[c + 1], floss[c], r, gr, NULL, NULL);
            for (int k = 0; k < 3; k++) force[c + k] = -gr[k];
        }
}

/* test hook (tests/test_general_models_cpu.py): the solver alone, J row-major [nc][nv] */
void or_solve_rows(OrModel *m, int nv, const double *M, const double *fs, int nc, const double *Jflat, const double *aref,
                   const double *D, const int *kind, const double *floss, double *a, double *force) {
    static double J[MAXC][MAXV];
    for (int c = 0; c < nc; c++)
        for (int j = 0; j < MAXV; j++) J[c][j] = j < nv ? Jflat[c * nv + j] : 0.0;
    solve_rows(m, nv, M, fs, nc, J, aref, D, kind, floss, a, force);
}

/* The surface point of a solid cylinder - axis from p0 along d (end to end), radius r - nearest to a point c (world
 * coordinates): outside, the point of the solid nearest to c (axial coordinate clamped to the caps, radial one to the
 * radius) and the normal from it to c; inside, the nearest of the side and the two caps, its outward normal.  len = signed
 * distance of c from the surface.  Returns 0 when c lies on the surface to rounding.  (MuJoCo 2.0 hands sphere / capsule
 * against cylinder to its convex collider; this is the closed form a later MuJoCo's mjc_SphereCylinder computes, restated
 * as a scheme of its own.) */
static int cyl_point(const double *p0, const double *d, double r, const double *c, double *q, double *n, double *len) {
    double L = sqrt(dot3(d, d)), u[3], w[3], rv[3], rh[3];
    for (int i = 0; i < 3; i++) { u[i] = d[i] / L; w[i] = c[i] - p0[i]; }
    double z = dot3(w, u);
    for (int i = 0; i < 3; i++) rv[i] = w[i] - z * u[i];
    double rho = sqrt(dot3(rv, rv));
    if (rho > 1e-14) { for (int i = 0; i < 3; i++) rh[i] = rv[i] / rho; }
    else {                                  /* on the axis: a fixed direction across it */
        double e[3] = {0, 0, 0}, pr;
        if (u[0] < 0.9 && u[0] > -0.9) e[0] = 1; else e[1] = 1;
        pr = dot3(e, u);
        for (int i = 0; i < 3; i++) rh[i] = e[i] - pr * u[i];
        pr = sqrt(dot3(rh, rh));
        for (int i = 0; i < 3; i++) rh[i] /= pr;
    }
    if (!(z > 0 && z < L && rho < r)) {
        double zc = z < 0 ? 0 : (z > L ? L : z), rc = rho < r ? rho : r, diff[3];
        for (int i = 0; i < 3; i++) { q[i] = p0[i] + zc * u[i] + rc * rh[i]; diff[i] = c[i] - q[i]; }
        *len = sqrt(dot3(diff, diff));
        if (*len < 1e-14) return 0;
        for (int i = 0; i < 3; i++) n[i] = diff[i] / *len;
        return 1;
    }
    double ds = r - rho, db = z, dt = L - z;
    if (ds <= db && ds <= dt) {
        for (int i = 0; i < 3; i++) { n[i] = rh[i]; q[i] = p0[i] + z * u[i] + r * rh[i]; }
        *len = -ds;
    } else if (db <= dt) {
        for (int i = 0; i < 3; i++) { n[i] = -u[i]; q[i] = p0[i] + rho * rh[i]; }
        *len = -db;
    } else {
        for (int i = 0; i < 3; i++) { n[i] = u[i]; q[i] = p0[i] + L * u[i] + rho * rh[i]; }
        *len = -dt;
    }
    return 1;
}
int or_cyl_point(const double *p0, const double *d, double r, const double *c, double *q, double *n, double *len) {      /* (test hook) */
    return cyl_point(p0, d, r, c, q, n, len);
}

/* closest points of two segments p1 + s d1, p2 + t d2, s, t in [0, 1] (a sphere is a segment of length 0); parallel
 * segments take s = 0 */
static double clamp01(double x) { return x < 0 ? 0 : (x > 1 ? 1 : x); }
static void seg_seg(const double *p1, const double *d1, const double *p2, const double *d2, double *s, double *t) {
    double r[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
    double a = dot3(d1, d1), e = dot3(d2, d2), f = dot3(d2, r);
    const double EPS = 1e-18;
    if (a <= EPS && e <= EPS) { *s = *t = 0; return; }
    if (a <= EPS) { *s = 0; *t = clamp01(f / e); return; }
    double c = dot3(d1, r);
    if (e <= EPS) { *t = 0; *s = clamp01(-c / a); return; }
    double b = dot3(d1, d2), denom = a * e - b * b;
    *s = denom > 1e-12 * a * e ? clamp01((b * f - c * e) / denom) : 0.0;
    *t = (b * (*s) + f) / e;
    if (*t < 0) { *t = 0; *s = clamp01(-c / a); }
    else if (*t > 1) { *t = 1; *s = clamp01((b - c) / a); }
}

/* constraint rows of ONE contact (MuJoCo mj_instantiateContact + mj_makeConstraint for a contact): normal n (from body B
 * to body A; B = 0 is the world), contact point cp, signed distance dist, friction mu (0: condim 1, one frictionless
 * row; else condim 3, MuJoCo's default pyramidal cone: rows Jn +- mu Jt_k in the frame mju_makeFrame builds from the
 * normal and - if it is not zero - the axis hint; diagApprox = tran (1 + mu^2), all four rows R = 2 mu^2 R_first).
 * J = the relative velocity of the two bodies' material points at cp. */
static void contact_rows(const OrModel *m, const Kin *k, const double *v, const double *n, const double *cp, int bA, int bB,
                         double dist, double margin, double mu, const double *solref, const double *solimp,
                         const double *axis_hint, double (*J)[MAXV], double *aref,
                         double *D, int *kind, double *floss, int *pnc) {
    int nv = m->nv, nc = *pnc;
    double Jp[3 * MAXV], Jq[3 * MAXV];
    jacobian(m, k, bA, cp, Jp, NULL);
    if (bB > 0) {
        jacobian(m, k, bB, cp, Jq, NULL);
        for (int i = 0; i < 3 * nv; i++) Jp[i] -= Jq[i];
    }
    double tran = m->body_invweight0[bA] + m->body_invweight0[bB];
    if (mu <= 0) {
        double jv = 0;
        for (int j = 0; j < nv; j++) {
            J[nc][j] = n[0] * Jp[j] + n[1] * Jp[nv + j] + n[2] * Jp[2 * nv + j];
            jv += J[nc][j] * v[j];
        }
        row_params_set(m, solref, solimp, dist, margin, tran, jv, &D[nc], &aref[nc]);
        kind[nc] = ROW_UNI;
        floss[nc] = 0;
        nc++;
    } else {
        double t1[3], t2[3], ax[3] = {axis_hint[0], axis_hint[1], axis_hint[2]};
        if (sqrt(dot3(ax, ax)) < 0.5) {
            ax[0] = 0; ax[1] = 0; ax[2] = 0;
            if (n[1] < 0.5 && n[1] > -0.5) ax[1] = 1; else ax[2] = 1;
        }
        double pr = dot3(n, ax), nr = 0;
        for (int i = 0; i < 3; i++) { t1[i] = ax[i] - pr * n[i]; nr += t1[i] * t1[i]; }
        nr = sqrt(nr);
        if (nr < MJ_MINVAL) { t1[0] = 1; t1[1] = 0; t1[2] = 0; }
        else for (int i = 0; i < 3; i++) t1[i] /= nr;
        cross3(n, t1, t2);
        if (m->cone == 1) {
            /* ELLIPTIC cone, condim 3 (mj_instantiateContact / mj_makeImpedance [EXT]): three rows - the normal, whose
             * position is the distance, and the two tangents, whose position is 0 - on the Jacobians themselves (friction
             * does not scale them); diagApprox of the normal row = tran; the friction rows take R = R_normal / impratio
             * (R_j = R_1 f_0^2 / f_{j-1}^2 further on: one friction coefficient here), their reference acceleration is
             * pure damping, -B J_t v.  The friction coefficient travels in floss[] of the normal row. */
            double jvk[3] = {0, 0, 0}, D0, Dk, ak;
            for (int j = 0; j < nv; j++) {
                J[nc][j] = n[0] * Jp[j] + n[1] * Jp[nv + j] + n[2] * Jp[2 * nv + j];
                J[nc + 1][j] = t1[0] * Jp[j] + t1[1] * Jp[nv + j] + t1[2] * Jp[2 * nv + j];
                J[nc + 2][j] = t2[0] * Jp[j] + t2[1] * Jp[nv + j] + t2[2] * Jp[2 * nv + j];
                for (int r = 0; r < 3; r++) jvk[r] += J[nc + r][j] * v[j];
            }
            row_params_set(m, solref, solimp, dist, margin, tran, jvk[0], &D0, &aref[nc]);
            D[nc] = D0;
            kind[nc] = ROW_ELL;
            floss[nc] = mu;
            for (int r = 1; r < 3; r++) {
                row_params_set(m, solref, solimp, 0.0, 0.0, tran, jvk[r], &Dk, &ak);
                aref[nc + r] = ak;                          /* (r = 0: the spring term vanishes) */
                D[nc + r] = D0 * (m->impratio > MJ_MINVAL ? m->impratio : MJ_MINVAL);
                kind[nc + r] = ROW_ELLT;
                floss[nc + r] = 0;
            }
            *pnc = nc + 3;
            return;
        }
        double D0, a0;
        row_params_set(m, solref, solimp, dist, margin, tran * (1 + mu * mu), 0.0, &D0, &a0);
        double Rpy = 2 * mu * mu / D0;
        for (int kk = 0; kk < 2; kk++) {
            const double *tt = kk == 0 ? t1 : t2;
            for (int sg = 1; sg >= -1; sg -= 2) {
                double jvr = 0, Dd, ar;
                for (int j = 0; j < nv; j++) {
                    double jn = n[0] * Jp[j] + n[1] * Jp[nv + j] + n[2] * Jp[2 * nv + j];
                    double jt = tt[0] * Jp[j] + tt[1] * Jp[nv + j] + tt[2] * Jp[2 * nv + j];
                    J[nc][j] = jn + sg * mu * jt;
                    jvr += J[nc][j] * v[j];
                }
                row_params_set(m, solref, solimp, dist, margin, tran * (1 + mu * mu), jvr, &Dd, &ar);
                D[nc] = 1 / Rpy;
                aref[nc] = ar;
                kind[nc] = ROW_UNI;
                floss[nc] = 0;
                nc++;
            }
        }
    }
    *pnc = nc;
}

/* The surface point of a solid box (half sizes h, in its own frame) nearest to a point `loc` given in that frame
 * (mjc_SphereBox's geometry): outside - the clamped point, normal from it to `loc`; inside - the nearest face, its outward
 * normal.  len = signed distance of `loc` from the surface (negative inside).  Returns 0 when `loc` lies ON the surface to
 * rounding (no normal). */
static int box_point(const double *h, const double *loc, double *cl, double *nb, double *len) {
    int inside = 1;
    for (int i = 0; i < 3; i++) {
        cl[i] = loc[i] < -h[i] ? -h[i] : (loc[i] > h[i] ? h[i] : loc[i]);
        if (cl[i] != loc[i]) inside = 0;
    }
    if (inside) {       /* the face the point is nearest to: surface point on it, outward normal */
        int kk = 0;
        double best = 1e300;
        for (int i = 0; i < 3; i++) {
            double gap = h[i] - fabs(loc[i]);
            if (gap < best) { best = gap; kk = i; }
        }
        double sg = loc[kk] >= 0 ? 1.0 : -1.0;
        cl[kk] = sg * h[kk];
        nb[0] = nb[1] = nb[2] = 0;
        nb[kk] = sg;
        *len = -best;
        return 1;
    }
    for (int i = 0; i < 3; i++) nb[i] = loc[i] - cl[i];
    *len = sqrt(dot3(nb, nb));
    if (*len < 1e-14) return 0;
    for (int i = 0; i < 3; i++) nb[i] /= *len;
    return 1;
}

/* The parameter t in [0, 1] at which the segment a + t b (box frame) comes nearest to the solid box of half sizes h: the
 * squared distance f(t) = sum_i max(0, |a_i + t b_i| - h_i)^2 is convex and piecewise quadratic - its pieces end where a
 * coordinate crosses a face plane (at most six break points) - so every piece is minimised in closed form and the least
 * minimum taken (the first one where several pieces tie: the lowest t) - or, when the segment passes through the box, a
 * point of its stretch inside. */
static double seg_box_param(const double *h, const double *a, const double *b) {
    double bp[8];
    int n = 8;
    bp[0] = 0.0;
    bp[7] = 1.0;
    for (int i = 0; i < 3; i++)
        for (int sg = 0; sg < 2; sg++) {
            double t = 1.0;         /* (a coordinate that does not move has no crossing: a duplicate of the end point) */
            if (b[i] != 0.0) t = ((sg ? h[i] : -h[i]) - a[i]) / b[i];
            bp[1 + 2 * i + sg] = t < 0 ? 0.0 : (t > 1 ? 1.0 : t);
        }
    for (int i = 1; i < n; i++)
        for (int j = i; j > 0 && bp[j] < bp[j - 1]; j--) { double t = bp[j]; bp[j] = bp[j - 1]; bp[j - 1] = t; }
    double best_f = 1e300, best_t = 0.0, zlo = 2.0, zhi = -1.0;
    for (int k = 0; k + 1 < n; k++) {
        double u = bp[k], w = bp[k + 1], mid = 0.5 * (u + w), B = 0, C = 0;
        double off[3];
        int act[3];
        for (int i = 0; i < 3; i++) {
            double si = a[i] + mid * b[i];
            act[i] = si > h[i] ? 1 : (si < -h[i] ? -1 : 0);
            off[i] = a[i] - act[i] * h[i];
            if (act[i]) { B += b[i] * b[i]; C += b[i] * off[i]; }
        }
        double tc = u;
        if (B > 0) { tc = -C / B; tc = tc < u ? u : (tc > w ? w : tc); }
        double f = 0;
        for (int i = 0; i < 3; i++)
            if (act[i]) { double e = off[i] + tc * b[i]; f += e * e; }
        if (f < best_f) { best_f = f; best_t = tc; }
        if (!act[0] && !act[1] && !act[2]) {        /* (no coordinate outside its faces on this piece: the axis runs INSIDE the box here) */
            if (u < zlo) zlo = u;
            if (w > zhi) zhi = w;
        }
    }
    /* an axis that enters the box has distance 0 on a whole stretch [zlo, zhi]: a point well inside it, whose nearest face is
     * defined whatever the rounding (an end point of the stretch lies ON the surface: inside or outside by one ulp) - three
     * eighths of the way rather than the middle, which lies exactly between two faces whenever the axis runs straight through
     * the box's centre plane (the sign of a zero would pick the face) */
    if (zhi >= zlo) return zlo + 0.375 * (zhi - zlo);
    return best_t;
}

/* Two boxes (round 5; mjc_BoxBox [EXT] is restated as a scheme of its own, not MuJoCo's routine): box 0 (centre c0, axes = the
 * COLUMNS of R0, half sizes h0) against box 1.  Separating-axis test over the 15 candidate axes (3 + 3 face normals, 9 edge
 * cross products; an edge axis must beat the best face axis by 5 % of its magnitude, as in most SAT colliders, so that faces
 * resting on each other take the face path); then
 *   a FACE axis: the other box's face most opposed to it is clipped against the four side planes of the reference face
 *                (Sutherland-Hodgman); the clipped polygon's corners within the margin are contacts, four of them at most
 *                (corners k n / 4 of an n-gon), each midway between the corner and the reference face;
 *   an EDGE axis: the closest points of the two edges, one contact midway.
 * n (out): unit normal from box 1 towards box 0.  Returns the number of contacts (pos[k], dist[k]). */
static int box_box(const double *c0, const double *R0, const double *h0, const double *c1, const double *R1, const double *h1,
                   double margin, double *n, double (*pos)[3], double *dist) {
    double A[3][3], B[3][3], d[3], hA[3] = {h0[0], h0[1], h0[2]}, hB[3] = {h1[0], h1[1], h1[2]};
    for (int i = 0; i < 3; i++)
        for (int k = 0; k < 3; k++) { A[i][k] = R0[3 * k + i]; B[i][k] = R1[3 * k + i]; }     /* A[i] = axis i of box 0, world */
    for (int k = 0; k < 3; k++) d[k] = c1[k] - c0[k];
    /* ---- the axis of least penetration (largest separation) */
    double best = -1e300, bestL[3] = {0, 0, 1};
    int code = -1;                          /* 0..2 face of box 0, 3..5 face of box 1, 6 + 3 i + j edge i of box 0 x edge j of box 1 */
    for (int t = 0; t < 6; t++) {
        const double *L = t < 3 ? A[t] : B[t - 3];
        double ra = 0, rb = 0;
        for (int i = 0; i < 3; i++) { ra += hA[i] * fabs(dot3(L, A[i])); rb += hB[i] * fabs(dot3(L, B[i])); }
        double sep = fabs(dot3(L, d)) - ra - rb;
        if (sep > best) { best = sep; code = t; memcpy(bestL, L, 24); }
    }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double L[3];
            cross3(A[i], B[j], L);
            double ln = sqrt(dot3(L, L));
            if (ln < 1e-9) continue;                                /* parallel edges: a face axis covers them */
            for (int k = 0; k < 3; k++) L[k] /= ln;
            double ra = 0, rb = 0;
            for (int k = 0; k < 3; k++) { ra += hA[k] * fabs(dot3(L, A[k])); rb += hB[k] * fabs(dot3(L, B[k])); }
            double sep = fabs(dot3(L, d)) - ra - rb;
            if (sep > best + 0.05 * fabs(best) + 1e-12) { best = sep; code = 6 + 3 * i + j; memcpy(bestL, L, 24); }
        }
    if (!(best < margin)) return 0;
    /* L points from box 0 to box 1; the contact normal from box 1 to box 0 */
    double sgn = dot3(bestL, d) < 0 ? -1.0 : 1.0, L[3];
    for (int k = 0; k < 3; k++) { L[k] = sgn * bestL[k]; n[k] = -L[k]; }
    if (code >= 6) {
        int i = (code - 6) / 3, j = (code - 6) % 3;
        double pa[3], pb[3];
        for (int k = 0; k < 3; k++) { pa[k] = c0[k]; pb[k] = c1[k]; }
        for (int a = 0; a < 3; a++) {
            if (a != i) { double sg = dot3(L, A[a]) > 0 ? 1.0 : -1.0; for (int k = 0; k < 3; k++) pa[k] += sg * hA[a] * A[a][k]; }
            if (a != j) { double sg = dot3(L, B[a]) > 0 ? -1.0 : 1.0; for (int k = 0; k < 3; k++) pb[k] += sg * hB[a] * B[a][k]; }
        }
        /* the two edges as segments: start at the edge centre minus the half length */
        double sa[3], da[3], sb[3], db[3], ta, tb;
        for (int k = 0; k < 3; k++) {
            sa[k] = pa[k] - hA[i] * A[i][k]; da[k] = 2 * hA[i] * A[i][k];
            sb[k] = pb[k] - hB[j] * B[j][k]; db[k] = 2 * hB[j] * B[j][k];
        }
        seg_seg(sa, da, sb, db, &ta, &tb);
        double qa[3], qb[3];
        for (int k = 0; k < 3; k++) { qa[k] = sa[k] + ta * da[k]; qb[k] = sb[k] + tb * db[k]; }
        double dd = 0;
        for (int k = 0; k < 3; k++) dd += (qb[k] - qa[k]) * L[k];
        dist[0] = dd;
        for (int k = 0; k < 3; k++) pos[0][k] = 0.5 * (qa[k] + qb[k]);
        return dd < margin ? 1 : 0;
    }
    /* ---- face contact: the reference box owns the axis */
    int ref1 = code >= 3, fi = code % 3;
    const double (*RA)[3] = ref1 ? B : A;       /* reference box axes / incident box axes */
    const double (*IA)[3] = ref1 ? A : B;
    const double *hr = ref1 ? hB : hA, *hi = ref1 ? hA : hB, *cr = ref1 ? c1 : c0, *ci = ref1 ? c0 : c1;
    double nr[3];                               /* outward normal of the reference face: towards the incident box */
    for (int k = 0; k < 3; k++) nr[k] = ref1 ? -L[k] : L[k];
    /* incident face: the incident box's face whose outward normal is most opposed to nr */
    int ii = 0;
    double most = -1;
    for (int a = 0; a < 3; a++) { double c = fabs(dot3(nr, IA[a])); if (c > most) { most = c; ii = a; } }
    double si = dot3(nr, IA[ii]) > 0 ? -1.0 : 1.0;
    int i1 = (ii + 1) % 3, i2 = (ii + 2) % 3;
    double poly[16][3], tmp[16][3];
    int np_ = 4;
    for (int c = 0; c < 4; c++) {
        double s1 = (c == 0 || c == 3) ? 1.0 : -1.0, s2 = c < 2 ? 1.0 : -1.0;
        for (int k = 0; k < 3; k++) poly[c][k] = ci[k] + si * hi[ii] * IA[ii][k] + s1 * hi[i1] * IA[i1][k] + s2 * hi[i2] * IA[i2][k];
    }
    int r1 = (fi + 1) % 3, r2 = (fi + 2) % 3;
    for (int side = 0; side < 4 && np_ > 0; side++) {
        /* side plane: u . (p - cr) <= hu */
        const double *u = RA[side < 2 ? r1 : r2];
        double sg = (side & 1) ? -1.0 : 1.0, hu = hr[side < 2 ? r1 : r2];
        int nt = 0;
        for (int c = 0; c < np_; c++) {
            const double *P = poly[c], *Q = poly[(c + 1) % np_];
            double dp = 0, dq = 0;
            for (int k = 0; k < 3; k++) { dp += sg * u[k] * (P[k] - cr[k]); dq += sg * u[k] * (Q[k] - cr[k]); }
            dp -= hu; dq -= hu;
            if (dp <= 0) { memcpy(tmp[nt++], P, 24); }
            if ((dp < 0 && dq > 0) || (dp > 0 && dq < 0)) {
                double t = dp / (dp - dq);
                for (int k = 0; k < 3; k++) tmp[nt][k] = P[k] + t * (Q[k] - P[k]);
                nt++;
            }
        }
        np_ = nt;
        memcpy(poly, tmp, sizeof(double) * 3 * nt);
    }
    /* the clipped corners within the margin of the reference face */
    double keep[16][3], kd[16];
    int nk = 0;
    for (int c = 0; c < np_; c++) {
        double dd = -hr[fi];
        for (int k = 0; k < 3; k++) dd += nr[k] * (poly[c][k] - cr[k]);
        if (dd < margin) { memcpy(keep[nk], poly[c], 24); kd[nk++] = dd; }
    }
    int nout = nk < 4 ? nk : 4;
    for (int c = 0; c < nout; c++) {
        int src = nk <= 4 ? c : (c * nk) / 4;
        dist[c] = kd[src];
        for (int k = 0; k < 3; k++) pos[c][k] = keep[src][k] - 0.5 * kd[src] * nr[k];
    }
    return nout;
}

/* (test hook) box_box on flat arrays: R row-major; returns the count, n[3], pos[4][3], dist[4] */
int or_box_box(const double *c0, const double *R0, const double *h0, const double *c1, const double *R1, const double *h1, double margin,
               double *n, double *pos, double *dist) {
    double P[4][3];
    int k = box_box(c0, R0, h0, c1, R1, h1, margin, n, P, dist);
    memcpy(pos, P, sizeof(P));
    return k;
}

double or_seg_box_param(const double *h, const double *a, const double *b) { return seg_box_param(h, a, b); }      /* (test hook) */

/* ---------------------------------------------------------------- one mj_step */
static int or_is_bad(double x) { return !(x <= 1e10 && x >= -1e10); }      /* mju_isBad: NaN, or beyond mjMAXVAL */
static void or_reset_data(const OrModel *m, double *q, double *v, double *ctrl) {     /* mj_resetData, the part the path reads */
    memcpy(q, m->qpos0, sizeof(double) * m->nq);
    memset(v, 0, sizeof(double) * m->nv);
    memset(ctrl, 0, sizeof(double) * m->nu);
}

/* q, v updated in place; site_out (optional) = finger site position computed from the q the step
 * STARTED with (MuJoCo runs kinematics before integrating and MjSim.step() does not call
 * mj_forward afterwards, so data.site_xpos lags qpos by one substep). */
/* mj_forward + mj_Euler.  check_acc: MuJoCo's mj_checkAcc - when the acceleration mj_forward arrives at (d->qacc: the
 * constraint solver's, M^-1 qfrc_smooth without rows) holds a NaN or an entry beyond mjMAXVAL = 1e10, return 1 BEFORE
 * integrating (the caller resets and runs the substep again, as mj_step does). */
static int step_impl(OrModel *m, double *q, double *v, const double *ctrl, double *site_out, double *diag, int check_acc) {
    int nv = m->nv;
    Kin k;
    double M[MAXV * MAXV], bias[MAXV], fs[MAXV];
    kinematics(m, q, &k);
    if (site_out) {                 /* [0:3] the site, [3:6] the object's axis (task 2) in the world */
        double t[3];
        matvec3(k.xmat[m->site_body], m->site_pos, t);
        for (int i = 0; i < 3; i++) site_out[i] = k.xpos[m->site_body][i] + t[i];
        matvec3(k.xmat[m->site_body], m->site_axis, site_out + 3);
    }
    mass_matrix(m, &k, M);
    rne(m, &k, v, NULL, bias);
    /* passive: joint dampers and springs (MuJoCo mj_passive) */
    for (int j = 0; j < nv; j++) {
        fs[j] = -bias[j] - m->damping[j] * v[j];
        if (m->stiffness[j] != 0) fs[j] -= m->stiffness[j] * (q[m->dof_qadr[j]] - m->springref[j]);
    }
    /* ... and the medium: viscous and drag forces on the box of equal inertia of every body, evaluated in the body's
     * inertial frame at its centre of mass, applied there (mj_passive, inertia-box fluid model) */
    if (m->density > 0 || m->viscosity > 0) {
        const double PI = 3.14159265358979323846;
        for (int b = 1; b < m->nbody; b++) {
            if (m->mass[b] <= MJ_MINVAL) continue;
            double Jp[3 * MAXV], Jr[3 * MAXV], vc[3] = {0, 0, 0}, w[3] = {0, 0, 0}, X[9], lv[3], lw[3], lf[3] = {0, 0, 0}, lt[3] = {0, 0, 0};
            jacobian(m, &k, b, k.xipos[b], Jp, Jr);
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < nv; j++) { vc[i] += Jp[i * nv + j] * v[j]; w[i] += Jr[i * nv + j] * v[j]; }
            matmul3(k.xmat[b], m->iR[b], X);
            for (int i = 0; i < 3; i++) {
                lv[i] = X[i] * vc[0] + X[3 + i] * vc[1] + X[6 + i] * vc[2];
                lw[i] = X[i] * w[0] + X[3 + i] * w[1] + X[6 + i] * w[2];
            }
            const double *bx = m->ibox[b];
            if (m->viscosity > 0) {
                double diam = (bx[0] + bx[1] + bx[2]) / 3.0;
                for (int i = 0; i < 3; i++) {
                    lt[i] += -PI * diam * diam * diam * m->viscosity * lw[i];
                    lf[i] += -3.0 * PI * diam * m->viscosity * lv[i];
                }
            }
            if (m->density > 0) {
                for (int i = 0; i < 3; i++) {
                    int j1 = (i + 1) % 3, j2 = (i + 2) % 3;
                    lf[i] -= 0.5 * m->density * bx[j1] * bx[j2] * fabs(lv[i]) * lv[i];
                    lt[i] -= m->density * bx[i] * (bx[j1] * bx[j1] * bx[j1] * bx[j1] + bx[j2] * bx[j2] * bx[j2] * bx[j2]) *
                             fabs(lw[i]) * lw[i] / 64.0;
                }
            }
            double wf[3], wt[3];
            matvec3(X, lf, wf);
            matvec3(X, lt, wt);
            for (int j = 0; j < nv; j++)
                for (int i = 0; i < 3; i++) fs[j] += Jp[i * nv + j] * wf[i] + Jr[i * nv + j] * wt[i];
        }
    }
    for (int a = 0; a < m->nu; a++) {                                          /* actuators */
        double u = ctrl[a];
        if (m->ctrllimited[a]) {
            if (u < m->ctrl_lo[a]) u = m->ctrl_lo[a];
            if (u > m->ctrl_hi[a]) u = m->ctrl_hi[a];
        }
        const double *b = m->act_bias[a];
        if (m->act_tendon[a]) {
            /* an actuator on a fixed tendon: length = gear sum coef q, moment arm of dof i = gear * coef_i */
            int t = m->act_dof[a];
            double len = 0, vel = 0;
            for (int i = 0; i < m->tn_n[t]; i++) {
                len += m->tn_coef[t][i] * q[m->dof_qadr[m->tn_dof[t][i]]];
                vel += m->tn_coef[t][i] * v[m->tn_dof[t][i]];
            }
            double frc = m->act_gain[a] * u + b[0] + b[1] * m->gear[a] * len + b[2] * m->gear[a] * vel;
            if (m->forcelimited[a]) frc = frc < m->forcerange[a][0] ? m->forcerange[a][0] : (frc > m->forcerange[a][1] ? m->forcerange[a][1] : frc);
            for (int i = 0; i < m->tn_n[t]; i++) fs[m->tn_dof[t][i]] += m->gear[a] * m->tn_coef[t][i] * frc;
            continue;
        }
        /* on a joint: length = gear q, velocity = gear v, the force acts through the gear */
        int d = m->act_dof[a];
        double len = (b[1] != 0.0) ? m->gear[a] * q[m->dof_qadr[d]] : 0.0;
        double frc = m->act_gain[a] * u + b[0] + b[1] * len + b[2] * m->gear[a] * v[d];
        if (m->forcelimited[a]) frc = frc < m->forcerange[a][0] ? m->forcerange[a][0] : (frc > m->forcerange[a][1] ? m->forcerange[a][1] : frc);
        fs[d] += m->gear[a] * frc;
    }
    /* constraint rows, in MuJoCo's order: equality, friction loss, limits, contacts */
    static const double ZERO3[3] = {0, 0, 0};
    double J[MAXC][MAXV], aref[MAXC], D[MAXC], floss[MAXC], force[MAXC], qacc[MAXV];
    int kind[MAXC];
    memset(kind, 0, sizeof(kind));          /* (ROW_UNI unless a row says otherwise) */
    memset(floss, 0, sizeof(floss));
    int nc = 0;
    for (int e = 0; e < m->neq; e++) {
        if (m->eq_type[e] == 1 || m->eq_type[e] == 2) {
            /* connect: the anchor as a point of body 1 and as a point of body 2 coincide; rows along the world axes,
             * J = jacp(body 1, p1) - jacp(body 2, p2), diagApprox = the two bodies' translational invweight0 */
            int b1 = m->eq_o1[e], b2 = m->eq_o2[e];
            double p1[3], p2[3], t[3], J1[3 * MAXV], J2[3 * MAXV];
            matvec3(k.xmat[b1], m->eq_anchor[e][0], t);
            for (int i = 0; i < 3; i++) p1[i] = k.xpos[b1][i] + t[i];
            matvec3(k.xmat[b2], m->eq_anchor[e][1], t);
            for (int i = 0; i < 3; i++) p2[i] = k.xpos[b2][i] + t[i];
            jacobian(m, &k, b1, p1, J1, NULL);
            jacobian(m, &k, b2, p2, J2, NULL);
            for (int i = 0; i < 3; i++) {
                double jv = 0;
                for (int j = 0; j < nv; j++) {
                    J[nc][j] = J1[i * nv + j] - (b2 > 0 ? J2[i * nv + j] : 0.0);
                    jv += J[nc][j] * v[j];
                }
                row_params_set(m, m->eq_solref[e], m->eq_solimp[e], p1[i] - p2[i], 0.0,
                               m->body_invweight0[b1] + m->body_invweight0[b2], jv, &D[nc], &aref[nc]);
                kind[nc] = ROW_EQ;
                floss[nc] = 0;
                nc++;
            }
            if (m->eq_type[e] == 2) {
                /* weld, rotation (MuJoCo mj_instantiateEquality, mjEQ_WELD): the error quaternion inv(q2) q1 qrel is the
                 * identity at qpos0; its vector part is the residual (three rows, in body 2's frame), its derivative
                 * 1/2 inv(q2) (0, w1 - w2) q1 qrel the Jacobian: with (w, v) the error quaternion and u = R2' (jacr1 -
                 * jacr2)_j, column j is 1/2 (w u + u x v).  diagApprox = the two bodies' ROTATIONAL invweight0. */
                double E[9], T[9], R2t[9], qe[4], Jr1[3 * MAXV], Jr2[3 * MAXV], Jd[3 * MAXV];
                for (int i = 0; i < 3; i++)
                    for (int j = 0; j < 3; j++) R2t[3 * i + j] = k.xmat[b2][3 * j + i];
                matmul3(R2t, k.xmat[b1], T);
                matmul3(T, m->eq_relR[e], E);
                mat2quat(E, qe);
                jacobian(m, &k, b1, k.xpos[b1], Jd, Jr1);
                jacobian(m, &k, b2, k.xpos[b2], Jd, Jr2);
                for (int j = 0; j < nv; j++) {
                    double dw[3] = {Jr1[j] - (b2 > 0 ? Jr2[j] : 0.0), Jr1[nv + j] - (b2 > 0 ? Jr2[nv + j] : 0.0),
                                    Jr1[2 * nv + j] - (b2 > 0 ? Jr2[2 * nv + j] : 0.0)}, u[3], uxv[3];
                    matvec3(R2t, dw, u);
                    cross3(u, qe + 1, uxv);
                    for (int i = 0; i < 3; i++) J[nc + i][j] = 0.5 * (qe[0] * u[i] + uxv[i]);
                }
                for (int i = 0; i < 3; i++) {
                    double jv = 0;
                    for (int j = 0; j < nv; j++) jv += J[nc][j] * v[j];
                    row_params_set(m, m->eq_solref[e], m->eq_solimp[e], qe[1 + i], 0.0,
                                   m->body_invweight0r[b1] + m->body_invweight0r[b2], jv, &D[nc], &aref[nc]);
                    kind[nc] = ROW_EQ;
                    floss[nc] = 0;
                    nc++;
                }
            }
        } else {
            /* joint: q1 - q1_0 = poly(q2 - q2_0) (qpos0 = the joints' ref) */
            int d1 = m->eq_o1[e], d2 = m->eq_o2[e];
            const double *pc = m->eq_poly[e];
            double x = d2 >= 0 ? q[m->dof_qadr[d2]] - m->qpos0[m->dof_qadr[d2]] : 0.0;
            double poly = pc[0] + x * (pc[1] + x * (pc[2] + x * (pc[3] + x * pc[4])));
            double dpoly = pc[1] + x * (2 * pc[2] + x * (3 * pc[3] + x * 4 * pc[4]));
            memset(J[nc], 0, sizeof(J[nc]));
            J[nc][d1] = 1.0;
            if (d2 >= 0) J[nc][d2] = -dpoly;
            double jv = v[d1] - (d2 >= 0 ? dpoly * v[d2] : 0.0);
            row_params_set(m, m->eq_solref[e], m->eq_solimp[e], q[m->dof_qadr[d1]] - m->qpos0[m->dof_qadr[d1]] - poly, 0.0,
                           m->dof_invweight0[d1] + (d2 >= 0 ? m->dof_invweight0[d2] : 0.0), jv, &D[nc], &aref[nc]);
            kind[nc] = ROW_EQ;
            floss[nc] = 0;
            nc++;
        }
    }
    /* friction loss (mj_instantiateFriction): one row per dof with frictionloss > 0, J = e_j, pos = 0 */
    for (int j = 0; j < nv; j++) {
        if (!(m->frictionloss[j] > 0)) continue;
        memset(J[nc], 0, sizeof(J[nc]));
        J[nc][j] = 1.0;
        row_params_set(m, m->dof_solref_f[j], m->dof_solimp_f[j], 0.0, 0.0, m->dof_invweight0[j], v[j], &D[nc], &aref[nc]);
        kind[nc] = ROW_FRIC;
        floss[nc] = m->frictionloss[j];
        nc++;
    }
    /* joint limits: MuJoCo mj_instantiateLimit (dist < margin, the joint's own); hinge and slide joints */
    for (int j = 0; j < nv; j++) {
        if (!m->limited[j] || m->dof_qadr[j] < 0) continue;
        double qj = q[m->dof_qadr[j]];
        for (int side = -1; side <= 1; side += 2) {
            double dist = side * (m->range[j][(side + 1) / 2] - qj);
            if (dist < m->jmargin[j]) {
                memset(J[nc], 0, sizeof(J[nc]));
                J[nc][j] = -side;
                row_params_set(m, m->dof_solref_l[j], m->dof_solimp_l[j], dist, m->jmargin[j], m->dof_invweight0[j], -side * v[j], &D[nc], &aref[nc]);
                nc++;
            }
        }
    }
    /* ball-joint limits (mj_instantiateLimit, mjJNT_BALL [EXT]): the joint quaternion as axis * angle (mju_quat2Vel with
     * dt = 1: angle = 2 atan2(|xyz|, w), wrapped to (-pi, pi]), dist = max(range) - |angle|; one row, J = -axis over the
     * joint's three dofs (angular velocity in the child body's frame), diagApprox = the dofs' invweight0 */
    for (int b = 1; b <= m->nbody; b++) {
        if (m->jtype[b] != JBALL || !m->ball_limited[b]) continue;
        const double *qq = q + m->qadr[b];
        double ax[3] = {qq[1], qq[2], qq[3]}, sn = sqrt(dot3(ax, ax));
        if (sn < MJ_MINVAL) continue;
        double ang = 2 * atan2(sn, qq[0]);
        if (ang > OR_PI) ang -= 2 * OR_PI;
        double sgn = ang < 0 ? -1.0 : 1.0, value = fabs(ang);
        double dist = m->ball_range[b] - value;
        if (dist < 0) {
            int d0 = m->dofid[b];
            double jv = 0;
            memset(J[nc], 0, sizeof(J[nc]));
            for (int i = 0; i < 3; i++) {
                J[nc][d0 + i] = -sgn * ax[i] / sn;
                jv += J[nc][d0 + i] * v[d0 + i];
            }
            row_params_set(m, m->dof_solref_l[d0], m->dof_solimp_l[d0], dist, 0.0, m->dof_invweight0[d0], jv, &D[nc], &aref[nc]);
            nc++;
        }
    }
    /* tendon limits: length = sum coef q; rows like the joint limits' with J = +-coef, diagApprox = tendon_invweight0 */
    for (int t = 0; t < m->ntendon; t++) {
        if (!m->tn_limited[t]) continue;
        double len = 0, lv = 0;
        for (int i = 0; i < m->tn_n[t]; i++) {
            len += m->tn_coef[t][i] * q[m->dof_qadr[m->tn_dof[t][i]]];
            lv += m->tn_coef[t][i] * v[m->tn_dof[t][i]];
        }
        for (int side = -1; side <= 1; side += 2) {
            double dist = side * (m->tn_range[t][(side + 1) / 2] - len);
            if (dist < m->tn_margin[t]) {
                memset(J[nc], 0, sizeof(J[nc]));
                for (int i = 0; i < m->tn_n[t]; i++) J[nc][m->tn_dof[t][i]] += -side * m->tn_coef[t][i];
                row_params_set(m, m->tn_solref[t], m->tn_solimp[t], dist, m->tn_margin[t], m->tn_invweight0[t], -side * lv, &D[nc], &aref[nc]);
                nc++;
            }
        }
    }
    /* plane-sphere contacts (mjc_PlaneSphere / the two ends mjc_PlaneCapsule tests / the corners mjc_PlaneBox tests +
     * mj_instantiateContact): margin = max of the two geom margins, gap = 0, included when dist < margin */
    int box_count = 0;          /* contacts of the box whose corners are being walked (mjc_PlaneBox keeps at most four) */
    for (int s = 0; m->has_plane && s < m->nsphere; s++) {
        int b = m->sph_body[s];
        double c[3], t[3];
        double margin = (m->plane_margin > m->sph_margin[s] ? m->plane_margin : m->sph_margin[s]) -
                        (m->plane_gap > m->sph_gap[s] ? m->plane_gap : m->sph_gap[s]);       /* includemargin = margin - gap */
        if (m->sph_kind[s] == 2) {
            /* a cylinder on the plane (mjc_PlaneCylinder [EXT], restated from its published description: the lowest point of
             * the lower cap's rim, the point below it on the other cap's rim, and two more points of the lower cap's rim 120
             * degrees to either side - a triangle under a cylinder that stands, a line under one that lies): point sph_k of
             * the four; a point counts while its distance is below the margin, the later ones only if the first does */
            double ctr[3], a[3], vec[3], n[3] = {m->plane_n[0], m->plane_n[1], m->plane_n[2]};
            matvec3(k.xmat[b], m->sph_ctr[s], t);
            for (int i = 0; i < 3; i++) ctr[i] = k.xpos[b][i] + t[i];
            matvec3(k.xmat[b], m->sph_axis[s], a);
            double pa = dot3(n, a);
            if (pa > 0) { for (int i = 0; i < 3; i++) a[i] = -a[i]; pa = -pa; }      /* the axis points down the normal */
            for (int i = 0; i < 3; i++) vec[i] = n[i] - pa * a[i];                   /* the normal within the cap's plane */
            double len = sqrt(dot3(vec, vec)), r = m->sph_r[s], hh = m->sph_hh[s];
            if (len < 1e-12) {          /* upright: any direction of the cap's plane - the frame tangent of the axis */
                double ax[3] = {0, 0, 0};
                if (a[1] < 0.5 && a[1] > -0.5) ax[1] = 1; else ax[2] = 1;
                double pr = dot3(a, ax);
                for (int i = 0; i < 3; i++) vec[i] = ax[i] - pr * a[i];
                len = sqrt(dot3(vec, vec));
            }
            for (int i = 0; i < 3; i++) vec[i] *= r / len;
            double d0 = 0;
            for (int i = 0; i < 3; i++) d0 += n[i] * (ctr[i] - m->plane_pos[i]);
            double pv = dot3(n, vec), d1 = d0 + pa * hh - pv;
            if (!(d1 < margin)) continue;
            double pt[3], dist;
            int kk = m->sph_k[s];
            if (kk == 0) {
                for (int i = 0; i < 3; i++) pt[i] = ctr[i] + a[i] * hh - vec[i];
                dist = d1;
            } else if (kk == 1) {
                for (int i = 0; i < 3; i++) pt[i] = ctr[i] - a[i] * hh - vec[i];
                dist = d0 - pa * hh - pv;
            } else {
                double v1[3];
                cross3(vec, a, v1);
                double sc = (kk == 2 ? 1.0 : -1.0) * sqrt(3.0) / 2;
                for (int i = 0; i < 3; i++) pt[i] = ctr[i] + a[i] * hh + 0.5 * vec[i] + sc * v1[i];
                dist = d0 + pa * hh + 0.5 * pv;
            }
            if (dist < margin) {
                double cp[3];
                for (int i = 0; i < 3; i++) cp[i] = pt[i] - n[i] * 0.5 * dist;
                contact_rows(m, &k, v, m->plane_n, cp, b, 0, dist, margin, m->sph_mu[s], m->sph_solref[s], m->sph_solimp[s], ZERO3, J, aref, D, kind, floss, &nc);
            }
            continue;
        }
        matvec3(k.xmat[b], m->sph_pos[s], t);
        for (int i = 0; i < 3; i++) c[i] = k.xpos[b][i] + t[i] - m->plane_pos[i];
        double dist = dot3(c, m->plane_n) - m->sph_r[s];
        if (m->sph_kind[s] == 1) {
            /* a box's corner (mjc_PlaneBox [EXT]): corners on the upper side of the box centre are skipped, and the box
             * contributes its first four contacts in corner order */
            if (m->sph_k[s] == 0) box_count = 0;
            double tc[3], rel[3];
            matvec3(k.xmat[b], m->sph_ctr[s], tc);
            for (int i = 0; i < 3; i++) rel[i] = t[i] - tc[i];
            if (dot3(rel, m->plane_n) > 0 || box_count >= 4) continue;
            if (dist < margin) box_count++;
        }
        if (dist < margin) {
            double cp[3], ax[3];
            for (int i = 0; i < 3; i++) cp[i] = k.xpos[b][i] + t[i] - m->plane_n[i] * (m->sph_r[s] + 0.5 * dist);
            matvec3(k.xmat[b], m->sph_axis[s], ax);
            contact_rows(m, &k, v, m->plane_n, cp, b, 0, dist, margin, m->sph_mu[s], m->sph_solref[s], m->sph_solimp[s], ax, J, aref, D, kind, floss, &nc);
        }
    }
    /* geom-geom contacts (mjc_SphereSphere / SphereCapsule / CapsuleCapsule): the closest points of the two segments,
     * one contact; normal from the second geom to the first, contact point midway between the surfaces; the contact
     * frame comes from the normal alone (mju_makeFrame).  A sphere against a box (mjc_SphereBox): the closest point of
     * the box to the sphere's centre (centre outside the box), or the nearest face (centre inside). */
    for (int p = 0; p < m->npair; p++) {
        double o[2][3], d[2][3];
        for (int e = 0; e < 2; e++) {
            int b = m->pair_body[p][e];
            double t[3];
            matvec3(k.xmat[b], m->pair_a[p][e], t);
            for (int i = 0; i < 3; i++) o[e][i] = k.xpos[b][i] + t[i];
            matvec3(k.xmat[b], m->pair_d[p][e], d[e]);
        }
        double c1[3], c2[3], diff[3], len;
        if (m->pair_cyl[p] >= 0) {
            /* a sphere or a capsule against a cylinder (a scheme of its own: MuJoCo 2.0 uses its convex collider here): the
             * cylinder's surface point nearest to the sphere's centre (cyl_point); a capsule brings up to three such contacts -
             * the point of its axis nearest to the cylinder's AXIS, and its two ends where they are not that point */
            int ec = m->pair_cyl[p], es = 1 - ec;
            int capsule = dot3(m->pair_d[p][es], m->pair_d[p][es]) > 0;
            double ts = 0, tc = 0;
            if (capsule) seg_seg(o[es], d[es], o[ec], d[ec], &ts, &tc);
            for (int cand = 0; cand < (capsule ? 3 : 1); cand++) {
                double tt = cand == 0 ? ts : (cand == 1 ? 0.0 : 1.0);
                if (cand > 0 && tt == ts) continue;
                double ps[3], q[3], nn[3];
                for (int i = 0; i < 3; i++) ps[i] = o[es][i] + tt * d[es][i];
                if (!cyl_point(o[ec], d[ec], m->pair_cyl_r[p], ps, q, nn, &len)) continue;
                for (int i = 0; i < 3; i++) {
                    diff[i] = es == 0 ? nn[i] : -nn[i];                 /* nn points from the cylinder to the sphere */
                    c1[i] = es == 0 ? ps[i] : q[i];
                    c2[i] = es == 0 ? q[i] : ps[i];
                }
                double dist = len - m->pair_r[p][0] - m->pair_r[p][1];
                if (dist < m->pair_margin[p]) {
                    double cp[3];
                    for (int i = 0; i < 3; i++) cp[i] = c2[i] + diff[i] * (m->pair_r[p][1] + 0.5 * dist);
                    contact_rows(m, &k, v, diff, cp, m->pair_body[p][0], m->pair_body[p][1], dist, m->pair_margin[p], m->pair_mu[p], m->pair_solref[p], m->pair_solimp[p],
                                 ZERO3, J, aref, D, kind, floss, &nc);
                }
            }
            continue;
        }
        if (m->pair_box[p] == 2) {
            double Rw0[9], Rw1[9], n[3], P4[4][3], d4[4];
            matmul3(k.xmat[m->pair_body[p][0]], m->pair_R[p], Rw0);
            matmul3(k.xmat[m->pair_body[p][1]], m->pair_R2[p], Rw1);
            int nct = box_box(o[0], Rw0, m->pair_half[p], o[1], Rw1, m->pair_half2[p], m->pair_margin[p], n, P4, d4);
            for (int c = 0; c < nct; c++)
                contact_rows(m, &k, v, n, P4[c], m->pair_body[p][0], m->pair_body[p][1], d4[c], m->pair_margin[p], m->pair_mu[p], m->pair_solref[p],
                             m->pair_solimp[p], ZERO3, J, aref, D, kind, floss, &nc);
            continue;
        }
        if (m->pair_box[p] >= 0) {
            /* a sphere or a capsule against a box.  Sphere (mjc_SphereBox): the box's surface point nearest to the centre.
             * Capsule (mjc_CapsuleBox [EXT]; restated as a scheme of its own, not MuJoCo's routine): up to three contacts - where
             * the capsule's axis comes nearest to the box, and its two ends (each an end sphere against the box) where they
             * are not that point themselves */
            int eb = m->pair_box[p], es = 1 - eb, bb = m->pair_body[p][eb];
            double Rw[9], a_loc[3], b_loc[3], rel[3];
            matmul3(k.xmat[bb], m->pair_R[p], Rw);
            for (int i = 0; i < 3; i++) rel[i] = o[es][i] - o[eb][i];
            for (int i = 0; i < 3; i++) {
                a_loc[i] = Rw[i] * rel[0] + Rw[3 + i] * rel[1] + Rw[6 + i] * rel[2];
                b_loc[i] = Rw[i] * d[es][0] + Rw[3 + i] * d[es][1] + Rw[6 + i] * d[es][2];
            }
            int capsule = dot3(m->pair_d[p][es], m->pair_d[p][es]) > 0;
            double tstar = capsule ? seg_box_param(m->pair_half[p], a_loc, b_loc) : 0.0;
            for (int cand = 0; cand < (capsule ? 3 : 1); cand++) {
                double tt = cand == 0 ? tstar : (cand == 1 ? 0.0 : 1.0);
                if (cand > 0 && tt == tstar) continue;
                double loc[3], cl[3], nb[3], cb[3], nw[3], ps[3];
                for (int i = 0; i < 3; i++) { loc[i] = a_loc[i] + tt * b_loc[i]; ps[i] = o[es][i] + tt * d[es][i]; }
                if (!box_point(m->pair_half[p], loc, cl, nb, &len)) continue;
                matvec3(Rw, cl, cb);
                matvec3(Rw, nb, nw);
                for (int i = 0; i < 3; i++) cb[i] += o[eb][i];          /* the box's nearest surface point, world */
                /* as two "closest points" c1 (geom 0's side) - c2 (geom 1's side) along the normal from geom 1 to geom 0 */
                for (int i = 0; i < 3; i++) {
                    double sgn = es == 0 ? 1.0 : -1.0;                  /* nw points from the box to the sphere */
                    diff[i] = sgn * nw[i];
                    c1[i] = es == 0 ? ps[i] : cb[i];
                    c2[i] = es == 0 ? cb[i] : ps[i];
                }
                double dist = len - m->pair_r[p][0] - m->pair_r[p][1];
                if (dist < m->pair_margin[p]) {
                    double cp[3];
                    for (int i = 0; i < 3; i++) cp[i] = c2[i] + diff[i] * (m->pair_r[p][1] + 0.5 * dist);
                    contact_rows(m, &k, v, diff, cp, m->pair_body[p][0], m->pair_body[p][1], dist, m->pair_margin[p], m->pair_mu[p], m->pair_solref[p], m->pair_solimp[p],
                                 ZERO3, J, aref, D, kind, floss, &nc);
                }
            }
            continue;
        }
        double sA, sB;
        seg_seg(o[0], d[0], o[1], d[1], &sA, &sB);
        for (int i = 0; i < 3; i++) {
            c1[i] = o[0][i] + sA * d[0][i];
            c2[i] = o[1][i] + sB * d[1][i];
            diff[i] = c1[i] - c2[i];
        }
        len = sqrt(dot3(diff, diff));
        if (len < 1e-14) continue;                  /* coincident axes: no normal */
        double dist = len - m->pair_r[p][0] - m->pair_r[p][1];
        if (dist < m->pair_margin[p]) {
            double n[3], cp[3];
            for (int i = 0; i < 3; i++) {
                n[i] = diff[i] / len;
                cp[i] = c2[i] + n[i] * (m->pair_r[p][1] + 0.5 * dist);
            }
            contact_rows(m, &k, v, n, cp, m->pair_body[p][0], m->pair_body[p][1], dist, m->pair_margin[p], m->pair_mu[p], m->pair_solref[p], m->pair_solimp[p],
                         ZERO3, J, aref, D, kind, floss, &nc);
        }
    }
    solve_rows(m, nv, M, fs, nc, J, aref, D, kind, floss, qacc, force);
    if (check_acc)
        for (int j = 0; j < nv; j++)
            if (or_is_bad(qacc[j])) return 1;
    /* mj_Euler with implicit joint damping: (M + h diag(damping)) qacc' = qfrc_smooth + qfrc_constraint */
    double rhs[MAXV];
    memcpy(rhs, fs, sizeof(double) * nv);
    for (int c = 0; c < nc; c++)
        for (int j = 0; j < nv; j++) rhs[j] += J[c][j] * force[c];
    for (int j = 0; j < nv; j++) M[j * nv + j] += m->timestep * m->damping[j];
    chol(M, nv);
    chol_solve(M, nv, rhs);
    for (int j = 0; j < nv; j++) v[j] += m->timestep * rhs[j];
    /* mj_integratePos: hinge / slide q += h v; ball / free orientation: q <- q * exp(h w / 2), w in the body frame */
    for (int b = 1; b < m->nbody; b++) {
        int j = m->dofid[b];
        if (j < 0) continue;
        double *qq = q + m->qadr[b];
        if (m->jtype[b] == JBALL) {
            quat_integrate(qq, v + j, m->timestep);
        } else if (m->jtype[b] == JFREE) {
            for (int i = 0; i < 3; i++) qq[i] += m->timestep * v[j + i];
            quat_integrate(qq + 3, v + j + 3, m->timestep);
        } else {
            qq[0] += m->timestep * v[j];
        }
    }
    if (diag) {
        diag[0] = nc;
        for (int j = 0; j < nv; j++) diag[1 + j] = qacc[j];
    }
    return 0;
}

/* mj_step [EXT] = mj_checkPos, mj_checkVel, mj_forward, mj_checkAcc, mj_Euler.  The checks (engine_forward.c) look for a
 * NaN or an entry beyond mjMAXVAL = 1e10 (mju_isBad) in qpos / qvel / qacc; on one they issue the "simulation is unstable"
 * warning and call mj_resetData - qpos = qpos0, qvel = 0, ctrl = 0 (act, time, warm start too) - after which mj_checkAcc
 * runs mj_forward again and the step goes on from the reset state.  ctrl is IN-OUT: mjrl's do_simulation writes
 * data.ctrl once per env step and then calls sim.step() frame_skip times (mjrl mujoco_env.py [EXT]; the call site is
 * reacher_env.py:30), so a reset leaves the remaining substeps of that env step with zero controls.  Returns the number
 * of resets (0 .. 2).  (Under mujoco-py's DEFAULT warning callback the same warning raises MujocoException out of
 * sim.step() and the reference's worker dies, gym_env_wrapper.py:125-153 catches nothing; what is restated here is
 * MuJoCo's own behaviour, which is what runs with mujoco_py.ignore_mujoco_warnings.) */
int or_step_mj(OrModel *m, double *q, double *v, double *ctrl, double *site_out, double *diag) {
    int resets = 0, bad = 0;
    for (int i = 0; i < m->nq; i++) bad |= or_is_bad(q[i]);
    for (int i = 0; i < m->nv; i++) bad |= or_is_bad(v[i]);
    if (bad) {
        or_reset_data(m, q, v, ctrl);
        resets++;
    }
    if (step_impl(m, q, v, ctrl, site_out, diag, 1)) {
        or_reset_data(m, q, v, ctrl);
        resets++;
        step_impl(m, q, v, ctrl, site_out, diag, 0);
    }
    m->resets += resets;
    return resets;
}

/* one mj_step with controls that are not written back (a reset zeroes a private copy) */
void or_step(OrModel *m, double *q, double *v, const double *ctrl, double *site_out, double *diag) {
    double u[MAXV];
    memcpy(u, ctrl, sizeof(double) * m->nu);
    or_step_mj(m, q, v, u, site_out, diag);
}

/* ---------------------------------------------------------------- run-time model edits (dynamics randomization)
 * gym_env_wrapper.py:367-416 writes model.body_mass / body_inertia / dof_damping / geom_size in place;
 * MuJoCo does not re-run mj_setConst afterwards, so dof/body_invweight0 keep their load-time values. */
void or_set_body_mass(OrModel *m, int body, double mass) { m->mass[body] = mass; body_box(m, body); }
void or_set_body_inertia(OrModel *m, int body, const double *I9) { memcpy(m->inertia[body], I9, 72); body_box(m, body); }
void or_set_sphere_mu(OrModel *m, int s, double mu) { m->sph_mu[s] = mu; }
void or_set_sphere_pos(OrModel *m, int s, const double *xyz) { memcpy(m->sph_pos[s], xyz, 24); }
void or_set_dof_damping(OrModel *m, int dof, double d) { m->damping[dof] = d; }
void or_set_dof_frictionloss(OrModel *m, int dof, double f) { m->frictionloss[dof] = f; }
void or_set_sphere_radius(OrModel *m, int s, double r) { m->sph_r[s] = r; }

/* number of OpenMP threads or_rollout uses (n > 0 sets it first); 1 without OpenMP */
int or_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

/* ---------------------------------------------------------------- accessors for tests */
int or_nv(const OrModel *m) { return m->nv; }
int or_nq(const OrModel *m) { return m->nq; }
int or_nbody(const OrModel *m) { return m->nbody; }
int or_dobs(const OrModel *m) { return m->task == 1 ? m->nq + m->nv - m->obs_skip : m->nq + m->nv + 6; }
void or_qpos0(const OrModel *m, double *q) { memcpy(q, m->qpos0, sizeof(double) * m->nq); }
int or_npair(const OrModel *m) { return m->npair; }
void or_get_inertial(const OrModel *m, double *mass, double *ipos, double *inertia) {
    for (int b = 0; b < m->nbody; b++) {
        mass[b] = m->mass[b];
        memcpy(ipos + 3 * b, m->ipos[b], 24);
        memcpy(inertia + 9 * b, m->inertia[b], 72);
    }
}
void or_get_invweight0(const OrModel *m, double *dof, double *body) {
    memcpy(dof, m->dof_invweight0, sizeof(double) * m->nv);
    memcpy(body, m->body_invweight0, sizeof(double) * m->nbody);
}
void or_get_invweight0_rot(const OrModel *m, double *body) { memcpy(body, m->body_invweight0r, sizeof(double) * m->nbody); }
void or_get_newton_stats(const OrModel *m, long *out) {
    out[0] = m->newton_calls; out[1] = m->newton_iters; out[2] = m->newton_fail;
}
long or_get_resets(const OrModel *m) { return m->resets; }
void or_mass_matrix(const OrModel *m, const double *q, double *M) {
    Kin k;
    kinematics(m, q, &k);
    mass_matrix(m, &k, M);
}
void or_rne(const OrModel *m, const double *q, const double *v, const double *a, double *tau) {
    Kin k;
    kinematics(m, q, &k);
    rne(m, &k, v, a, tau);
}
void or_site(const OrModel *m, const double *q, double *xyz) {
    Kin k;
    kinematics(m, q, &k);
    double t[3];
    matvec3(k.xmat[m->site_body], m->site_pos, t);
    for (int i = 0; i < 3; i++) xyz[i] = k.xpos[m->site_body][i] + t[i];
}
/* total kinetic energy 1/2 v^T M v (armature included) */
double or_kinetic(const OrModel *m, const double *q, const double *v) {
    double M[MAXV * MAXV], e = 0;
    or_mass_matrix(m, q, M);
    for (int i = 0; i < m->nv; i++)
        for (int j = 0; j < m->nv; j++) e += 0.5 * v[i] * M[i * m->nv + j] * v[j];
    return e;
}

/* ---------------------------------------------------------------- env step + rollout */
/* Reacher7DOFEnv.step (reacher_env.py:29-39): frame_skip x mj_step, then
 * reward = -(|h-g|_1 + 5 |h-g|_2) with h = data.site_xpos[finger] (lagging one substep) and
 * obs = [qpos, qvel, h, h - g] (reacher_env.py:41-47). */
static double env_step(OrModel *m, double *q, double *v, const double *u, const double *target, double *obs) {
    if (m->task == 1) {
        /* SwimmerEnv.step / HalfCheetahEnv.step (swimmer.py:10-19, half_cheetah.py:10-19): reward = forward progress
         * of qpos[0] over the env step / dt - c * |a|^2 (the action as given, unclipped), obs = [qpos[skip:], qvel] */
        int nv = m->nv, nq = m->nq, sk = m->obs_skip;
        double x0 = q[0], c = 0, uu[MAXV];
        memcpy(uu, u, sizeof(double) * m->nu);              /* data.ctrl: written once per env step, zeroed by a reset */
        for (int s = 0; s < m->frame_skip; s++) or_step_mj(m, q, v, uu, NULL, NULL);
        for (int a = 0; a < m->nu; a++) c += u[a] * u[a];
        if (obs) {
            memcpy(obs, q + sk, sizeof(double) * (nq - sk));
            memcpy(obs + nq - sk, v, sizeof(double) * nv);
        }
        return (q[0] - x0) / (m->timestep * m->frame_skip) - m->ctrl_cost * c;
    }
    double h[3] = {0, 0, 0}, h6[6], uu[MAXV];
    memcpy(uu, u, sizeof(double) * m->nu);
    for (int s = 0; s < m->frame_skip; s++) { or_step_mj(m, q, v, uu, h6, NULL); memcpy(h, h6, 24); }
    double d[3] = {h[0] - target[0], h[1] - target[1], h[2] - target[2]};
    double l1 = fabs(d[0]) + fabs(d[1]) + fabs(d[2]), l2 = sqrt(dot3(d, d));
    if (obs) {
        int nv = m->nv, nq = m->nq;
        memcpy(obs, q, sizeof(double) * nq);
        memcpy(obs + nq, v, sizeof(double) * nv);
        memcpy(obs + nq + nv, h, 24);
        memcpy(obs + nq + nv + 3, d, 24);
    }
    /* task 2: the shape of pen-v0's reward (object to its target position, object axis to its target direction; both
     * from the kinematics the last substep started with, like the site) */
    if (m->task == 2) return -l2 + dot3(h6 + 3, m->target_dir);
    return -l1 - 5.0 * l2;
}

double or_env_step(OrModel *m, double *q, double *v, const double *u, const double *target, double *obs) {
    return env_step(m, q, v, u, target, obs);
}

/* GymEnvWrapper.rollout (gym_env_wrapper.py:89-156), mode "open_loop":
 * every particle restarts from (qp0, qv0); obs[b,0] is the fresh observation after
 * set_env_state (which ends with sim.forward(), reacher_env.py:99); act is the UNCLIPPED
 * mean + noise (gym_env_wrapper.py:151); done is always False (reacher_env.py:39).
 * Any of obs / next_obs / act / done may be NULL.  OpenMP over particles. */
static void rollout_impl(OrModel *m, const double *qp0, const double *qv0, const double *target, long P, int H,
                         const double *mean, const double *noise, double *obs, double *rew, double *act,
                         double *done, double *next_obs, int closed_loop);

void or_rollout(OrModel *m, const double *qp0, const double *qv0, const double *target, long P, int H,
                const double *mean, const double *noise, double *obs, double *rew, double *act,
                double *done, double *next_obs) {
    rollout_impl(m, qp0, qv0, target, P, H, mean, noise, obs, rew, act, done, next_obs, 0);
}

/* mode "closed_loop_linear" (gym_env_wrapper.py:135-136): mean is a (d_obs+1) x nu weight matrix and
 * the nominal action of a step is mean.T @ append(curr_obs, 1.0) */
void or_rollout_cl(OrModel *m, const double *qp0, const double *qv0, const double *target, long P, int H,
                   const double *weights, const double *noise, double *obs, double *rew, double *act,
                   double *done, double *next_obs) {
    rollout_impl(m, qp0, qv0, target, P, H, weights, noise, obs, rew, act, done, next_obs, 1);
}

static void rollout_impl(OrModel *m, const double *qp0, const double *qv0, const double *target, long P, int H,
                         const double *mean, const double *noise, double *obs, double *rew, double *act,
                         double *done, double *next_obs, int closed_loop) {
    int nv = m->nv, nq = m->nq, nu = m->nu, dobs = or_dobs(m);
    double h0[3] = {0, 0, 0};
    if (m->task != 1) or_site(m, qp0, h0);
    long calls = 0, iters = 0, fails = 0, resets = 0;
#pragma omp parallel for schedule(static) reduction(+ : calls, iters, fails, resets)
    for (long b = 0; b < P; b++) {
        OrModel loc = *m;       /* private statistics */
        loc.newton_calls = loc.newton_iters = loc.newton_fail = loc.resets = 0;
        double q[MAXQ], v[MAXV], cur[MAXQ + MAXV + 6], nxt[MAXQ + MAXV + 6], u[MAXV];
        memcpy(q, qp0, sizeof(double) * nq);
        memcpy(v, qv0, sizeof(double) * nv);
        if (m->task == 1) {
            memcpy(cur, q + m->obs_skip, sizeof(double) * (nq - m->obs_skip));
            memcpy(cur + nq - m->obs_skip, v, sizeof(double) * nv);
        } else {
            memcpy(cur, q, sizeof(double) * nq);
            memcpy(cur + nq, v, sizeof(double) * nv);
            for (int i = 0; i < 3; i++) { cur[nq + nv + i] = h0[i]; cur[nq + nv + 3 + i] = h0[i] - target[i]; }
        }
        for (int t = 0; t < H; t++) {
            for (int a = 0; a < nu; a++) {
                double ma;
                if (closed_loop) {
                    ma = mean[dobs * nu + a];
                    for (int k = 0; k < dobs; k++) ma += mean[k * nu + a] * cur[k];
                } else {
                    ma = mean[t * nu + a];
                }
                u[a] = ma + (noise ? noise[(b * H + t) * nu + a] : 0.0);
            }
            double r = env_step(&loc, q, v, u, target, nxt);
            long o = (b * H + t);
            if (obs) memcpy(obs + o * dobs, cur, sizeof(double) * dobs);
            if (next_obs) memcpy(next_obs + o * dobs, nxt, sizeof(double) * dobs);
            rew[o] = r;
            if (act) memcpy(act + o * nu, u, sizeof(double) * nu);
            if (done) done[o] = 0.0;
            memcpy(cur, nxt, sizeof(double) * dobs);
        }
        calls += loc.newton_calls; iters += loc.newton_iters; fails += loc.newton_fail; resets += loc.resets;
    }
    m->newton_calls += calls; m->newton_iters += iters; m->newton_fail += fails; m->resets += resets;
}
