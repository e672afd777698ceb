/*
 * rsik_oracle.c — CPU restatement (plain C99) of the analytic IK solve path of
 * pollen-robotics/reachy2_symbolic_ik.  TEST INFRASTRUCTURE: see rsik_oracle.h.
 *
 * Line references: S: = symbolic_ik.py, U: = utils.py, C: = control_ik.py under
 * /root/reference/src/reachy2_symbolic_ik/.  Third-party arithmetic on the path
 * (scipy.spatial.transform.Rotation 1.15.3, numpy 2.2.6 linspace/isclose/lstsq) is
 * restated from its published algorithms and pinned by tests/golden (npz files).
 */
#include "rsik_oracle.h"

#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_PI 3.141592653589793 /* == math.pi == np.pi */

/* ------------------------------------------------------------------ small linear algebra */
typedef struct { double m[3][3]; } mat3;
typedef struct { double m[4][4]; } mat4;

static double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static double norm3(const double a[3]) { return sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
static void cross3(const double a[3], const double b[3], double o[3]) {
    double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}
static mat3 mat3_mul(mat3 a, mat3 b) {
    mat3 r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
    return r;
}
static mat3 mat3_T(mat3 a) {
    mat3 r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r.m[i][j] = a.m[j][i];
    return r;
}
static void mat3_apply(mat3 a, const double v[3], double o[3]) {
    double x = a.m[0][0] * v[0] + a.m[0][1] * v[1] + a.m[0][2] * v[2];
    double y = a.m[1][0] * v[0] + a.m[1][1] * v[1] + a.m[1][2] * v[2];
    double z = a.m[2][0] * v[0] + a.m[2][1] * v[1] + a.m[2][2] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
static mat3 mat3_eye(void) {
    mat3 r = {{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}};
    return r;
}
static mat3 rot_x(double a) {
    double c = cos(a), s = sin(a);
    mat3 r = {{{1, 0, 0}, {0, c, -s}, {0, s, c}}};
    return r;
}
static mat3 rot_y(double a) {
    double c = cos(a), s = sin(a);
    mat3 r = {{{c, 0, s}, {0, 1, 0}, {-s, 0, c}}};
    return r;
}
static mat3 rot_z(double a) {
    double c = cos(a), s = sin(a);
    mat3 r = {{{c, -s, 0}, {s, c, 0}, {0, 0, 1}}};
    return r;
}
/* scipy R.from_euler("xyz", [a,b,c]).as_matrix(): lower-case = extrinsic => Rz(c) Ry(b) Rx(a). */
static mat3 from_euler_xyz_extrinsic(const double e[3]) { return mat3_mul(rot_z(e[2]), mat3_mul(rot_y(e[1]), rot_x(e[0]))); }

/* U:12-23 make_homogenous_matrix_from_rotation_matrix */
static mat4 hom(const double p[3], mat3 Rm) {
    mat4 T;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) T.m[i][j] = Rm.m[i][j];
        T.m[i][3] = p[i];
    }
    T.m[3][0] = 0.0; T.m[3][1] = 0.0; T.m[3][2] = 0.0; T.m[3][3] = 1.0;
    return T;
}
static mat4 mat4_mul(mat4 a, mat4 b) {
    mat4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double s = 0.0;
            for (int k = 0; k < 4; k++) s += a.m[i][k] * b.m[k][j];
            r.m[i][j] = s;
        }
    return r;
}
static void mat4_apply(mat4 T, const double v[4], double o[4]) {
    double t[4];
    for (int i = 0; i < 4; i++) t[i] = T.m[i][0] * v[0] + T.m[i][1] * v[1] + T.m[i][2] * v[2] + T.m[i][3] * v[3];
    memcpy(o, t, sizeof t);
}

/* numpy.isclose(a, b) with default rtol=1e-5, atol=1e-8: |a-b| <= atol + rtol*|b| */
static int np_isclose(double a, double b) { return fabs(a - b) <= (1e-8 + 1e-5 * fabs(b)); }

/* CPython / numpy float modulo (result takes the divisor's sign). */
double orc_pymod(double a, double b) {
    double m = fmod(a, b);
    if (m != 0.0) {
        if ((b < 0) != (m < 0)) m += b;
    } else {
        m = copysign(0.0, b);
    }
    return m;
}

/* U:486-490 */
double orc_angle_diff(double a, double b) {
    double d = a - b;
    d = orc_pymod(d + ORC_PI, 2 * ORC_PI) - ORC_PI;
    return d;
}

/* U:468-474 */
int orc_is_valid_angle(double angle, const double interval[2]) {
    if (orc_pymod(interval[0], 2 * ORC_PI) == orc_pymod(interval[1], 2 * ORC_PI)) return 1;
    if (interval[0] < interval[1]) return (interval[0] <= angle) && (angle <= interval[1]);
    return (interval[0] <= angle) || (angle <= interval[1]);
}

/* U:93-112 */
double orc_limit_theta_to_interval(double theta, double previous_theta, const double interval[2]) {
    theta = orc_pymod(theta, 2 * ORC_PI);
    if (theta > ORC_PI) theta -= 2 * ORC_PI;
    previous_theta = orc_pymod(previous_theta, 2 * ORC_PI); /* normalised, never used (Q12) */
    if (previous_theta > ORC_PI) previous_theta -= 2 * ORC_PI;
    (void)previous_theta;
    if (orc_is_valid_angle(theta, interval)) return theta;
    double posDiff = orc_angle_diff(theta, interval[1]);
    double negDiff = orc_angle_diff(theta, interval[0]);
    if (fabs(posDiff) < fabs(negDiff)) return interval[1];
    return interval[0];
}

/* U:59-81 */
static mat3 rotation_matrix_from_vector(const double vect[3]) {
    double n = norm3(vect);
    double v2[3] = {vect[0] / n, vect[1] / n, vect[2] / n};
    const double v1[3] = {1.0, 0.0, 0.0};
    if (np_isclose(v1[0], v2[0]) && np_isclose(v1[1], v2[1]) && np_isclose(v1[2], v2[2])) return mat3_eye();
    if (np_isclose(v1[0], -v2[0]) && np_isclose(v1[1], -v2[1]) && np_isclose(v1[2], -v2[2])) {
        mat3 r = {{{-1, 0, 0}, {0, 1, 0}, {0, 0, -1}}};
        return r;
    }
    double v[3];
    cross3(v1, v2, v);
    double c = dot3(v1, v2);
    double s = norm3(v);
    mat3 k = {{{0, -v[2], v[1]}, {v[2], 0, -v[0]}, {-v[1], v[0], 0}}};
    mat3 kk = mat3_mul(k, k);
    double h = (1 - c) / (s * s);
    mat3 r = mat3_eye();
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r.m[i][j] = r.m[i][j] + k.m[i][j] + kk.m[i][j] * h;
    return r;
}
void orc_rotation_matrix_from_vector(const double vect[3], double Rm[9]) {
    mat3 r = rotation_matrix_from_vector(vect);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Rm[3 * i + j] = r.m[i][j];
}

/* ------------------------------------------------------------------ scipy Rotation pieces */
/* quaternions are scipy's scalar-last (x, y, z, w) */
typedef struct { double x, y, z, w; } quat;

static quat quat_mul(quat p, quat q) { /* p (x) q : rotation q applied first, then p */
    quat r;
    r.w = p.w * q.w - p.x * q.x - p.y * q.y - p.z * q.z;
    r.x = p.w * q.x + p.x * q.w + p.y * q.z - p.z * q.y;
    r.y = p.w * q.y - p.x * q.z + p.y * q.w + p.z * q.x;
    r.z = p.w * q.z + p.x * q.y - p.y * q.x + p.z * q.w;
    return r;
}
static quat quat_axis(int axis, double angle) {
    quat q = {0, 0, 0, cos(angle / 2)};
    double s = sin(angle / 2);
    if (axis == 0) q.x = s; else if (axis == 1) q.y = s; else q.z = s;
    return q;
}
/* Rotation.from_matrix, first half (scipy >= 1.12): a matrix whose Gramian M M^T is not the identity
 * (np.isclose with atol = 1e-12 and the default rtol = 1e-5, i.e. 1e-12 off the diagonal) is replaced by the
 * solution of the orthogonal Procrustes problem, U V^T of its SVD.  Here U V^T = M (M^T M)^(-1/2), with the inverse
 * square root taken through a cyclic Jacobi eigen-decomposition of the symmetric 3x3 M^T M. */
static int gram_is_identity(mat3 M) {
    mat3 G = mat3_mul(M, mat3_T(M));
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double e = (i == j) ? 1.0 : 0.0;
            if (!(fabs(G.m[i][j] - e) <= 1e-12 + 1e-5 * e)) return 0;
        }
    return 1;
}
static mat3 procrustes_rotation(mat3 M) {
    mat3 S = mat3_mul(mat3_T(M), M), V = mat3_eye();
    for (int sweep = 0; sweep < 30; sweep++) {
        double off = fabs(S.m[0][1]) + fabs(S.m[0][2]) + fabs(S.m[1][2]);
        if (off < 1e-300) break;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                if (S.m[p][q] == 0.0) continue;
                double th = (S.m[q][q] - S.m[p][p]) / (2 * S.m[p][q]);
                double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1));
                double c = 1 / sqrt(t * t + 1), sn = t * c;
                mat3 J = mat3_eye();
                J.m[p][p] = c; J.m[q][q] = c; J.m[p][q] = sn; J.m[q][p] = -sn;
                S = mat3_mul(mat3_T(J), mat3_mul(S, J));
                V = mat3_mul(V, J);
            }
    }
    mat3 D = {{{1 / sqrt(S.m[0][0]), 0, 0}, {0, 1 / sqrt(S.m[1][1]), 0}, {0, 0, 1 / sqrt(S.m[2][2])}}};
    return mat3_mul(M, mat3_mul(V, mat3_mul(D, mat3_T(V))));
}
/* Rotation.from_matrix, second half: orthogonal matrix -> quaternion (Markley 2008, scipy _rotation.pyx) */
static quat quat_from_matrix(mat3 M) {
    if (!gram_is_identity(M)) M = procrustes_rotation(M);
    double tr = M.m[0][0] + M.m[1][1] + M.m[2][2];
    double dec[4] = {M.m[0][0], M.m[1][1], M.m[2][2], tr};
    int choice = 0;
    for (int i = 1; i < 4; i++)
        if (dec[i] > dec[choice]) choice = i;
    double q[4];
    if (choice != 3) {
        int i = choice, j = (i + 1) % 3, k = (j + 1) % 3;
        q[i] = 1 - dec[3] + 2 * M.m[i][i];
        q[j] = M.m[j][i] + M.m[i][j];
        q[k] = M.m[k][i] + M.m[i][k];
        q[3] = M.m[k][j] - M.m[j][k];
    } else {
        q[0] = M.m[2][1] - M.m[1][2];
        q[1] = M.m[0][2] - M.m[2][0];
        q[2] = M.m[1][0] - M.m[0][1];
        q[3] = 1 + dec[3];
    }
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    quat r = {q[0] / n, q[1] / n, q[2] / n, q[3] / n};
    return r;
}
/* Rotation.as_euler (scipy >= 1.12, Bernardes & Viollet 2022, "_compute_euler_from_quat").
 * seq: three axis indices in the order given by the caller; extrinsic: lower-case sequence. */
static void quat_as_euler(quat Q, const int seq_in[3], int extrinsic, double angles[3]) {
    int seq[3] = {seq_in[0], seq_in[1], seq_in[2]};
    if (!extrinsic) { int t = seq[0]; seq[0] = seq[2]; seq[2] = t; }
    int i = seq[0], j = seq[1], k = seq[2];
    int symmetric = (i == k);
    if (symmetric) k = 3 - i - j;
    int sign = (i - j) * (j - k) * (k - i) / 2;
    double q[4] = {Q.x, Q.y, Q.z, Q.w};
    double a, b, c, d;
    if (symmetric) {
        a = q[3]; b = q[i]; c = q[j]; d = q[k] * sign;
    } else {
        a = q[3] - q[j]; b = q[i] + q[k] * sign; c = q[j] + q[3]; d = q[k] * sign - q[i];
    }
    int first = extrinsic ? 0 : 2, third = extrinsic ? 2 : 0;
    angles[1] = 2 * atan2(hypot(c, d), hypot(a, b));
    int kase;
    if (fabs(angles[1]) <= 1e-7) kase = 1;
    else if (fabs(angles[1] - ORC_PI) <= 1e-7) kase = 2;
    else kase = 0;
    double half_sum = atan2(b, a), half_diff = atan2(d, c);
    if (kase == 0) {
        angles[first] = half_sum - half_diff;
        angles[third] = half_sum + half_diff;
    } else {
        angles[2] = 0;
        if (kase == 1) angles[0] = 2 * half_sum;
        else angles[0] = 2 * half_diff * (extrinsic ? -1 : 1);
    }
    if (!symmetric) {
        angles[third] *= sign;
        angles[1] -= ORC_PI / 2;
    }
    for (int n = 0; n < 3; n++) {
        if (angles[n] < -ORC_PI) angles[n] += 2 * ORC_PI;
        else if (angles[n] > ORC_PI) angles[n] -= 2 * ORC_PI;
    }
}

/* U:84-90: R.from_matrix(M[:3,:3]).as_euler("xyz") */
void orc_euler_from_matrix_xyz(const double Rm[9], double eul[3]) {
    mat3 M;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) M.m[i][j] = Rm[3 * i + j];
    const int seq[3] = {0, 1, 2};
    quat_as_euler(quat_from_matrix(M), seq, 1, eul);
}

/* U:508-519 limit_orbita3d_joints: intrinsic XYZ -> ZYZ, clamp beta, ZYZ -> XYZ */
void orc_limit_orbita3d_joints(const double in[3], double max_angle, double out[3]) {
    /* from_euler("XYZ") intrinsic: R = Rx(a) Ry(b) Rz(c) */
    quat q = quat_mul(quat_mul(quat_axis(0, in[0]), quat_axis(1, in[1])), quat_axis(2, in[2]));
    const int zyz[3] = {2, 1, 2};
    double nj[3];
    quat_as_euler(q, zyz, 0, nj);
    nj[1] = fmin(max_angle, fmax(-max_angle, nj[1]));
    quat q2 = quat_mul(quat_mul(quat_axis(2, nj[0]), quat_axis(1, nj[1])), quat_axis(2, nj[2]));
    const int xyz[3] = {0, 1, 2};
    quat_as_euler(q2, xyz, 0, out);
}

/* ------------------------------------------------------------------ SymbolicIK.__init__ */
int orc_arm_ndoubles(void) { return (int)(sizeof(orc_arm_t) / sizeof(double)); }
int orc_solver_ndoubles(void) { return (int)(sizeof(orc_solver_t) / sizeof(double)); }

static double radians(double deg) { return deg * (ORC_PI / 180.0); }

void orc_arm_init(orc_arm_t *arm, int is_left, const double shoulder_position[3],
                  const double shoulder_orientation_deg[3], double upper_arm_size, double forearm_size,
                  const double tip_position[3], double elbow_limit_deg, double wrist_limit_deg,
                  double projection_margin, double backward_limit, double normal_vector_margin,
                  double singularity_offset, double singularity_limit_coeff) {
    memset(arm, 0, sizeof *arm);
    arm->side = is_left ? -1.0 : 1.0;
    memcpy(arm->shoulder_position, shoulder_position, 3 * sizeof(double));
    memcpy(arm->shoulder_orientation_offset, shoulder_orientation_deg, 3 * sizeof(double));
    arm->upper_arm_size = upper_arm_size;
    arm->forearm_size = forearm_size;
    memcpy(arm->tip_position, tip_position, 3 * sizeof(double));
    arm->gripper_size = norm3(tip_position);                                /* S:64 */
    arm->max_arm_length = upper_arm_size + forearm_size + arm->gripper_size; /* S:65 */
    arm->projection_margin = projection_margin;
    arm->normal_vector_margin = normal_vector_margin;
    arm->backward_limit = backward_limit;
    arm->elbow_limit = elbow_limit_deg;
    arm->shoulder_wrist_min_distance =                                       /* S:73-77 */
        sqrt(upper_arm_size * upper_arm_size + forearm_size * forearm_size -
             2 * upper_arm_size * forearm_size * cos(radians(180 - elbow_limit_deg)));
    arm->wrist_limit = wrist_limit_deg;
    arm->singularity_offset = singularity_offset;
    arm->singularity_limit_coeff = singularity_limit_coeff;
    /* U:26-43 get_singularity_position */
    double off_rad[3] = {radians(shoulder_orientation_deg[0]), radians(shoulder_orientation_deg[1]),
                         radians(shoulder_orientation_deg[2])};
    mat4 T = hom(shoulder_position, from_euler_xyz_extrinsic(off_rad));
    double e4[4] = {0.0, -upper_arm_size * arm->side, 0.0, 1.0}, o4[4];
    mat4_apply(T, e4, o4);
    memcpy(arm->elbow_singularity_position, o4, 3 * sizeof(double));
    double w4[4] = {0.0, -(upper_arm_size + forearm_size) * arm->side, 0.0, 1.0};
    mat4_apply(T, w4, o4);
    memcpy(arm->wrist_singularity_position, o4, 3 * sizeof(double));
}

void orc_arm_init_default(orc_arm_t *arm, int is_left, double singularity_offset) {
    /* S:38-51 default ik_parameters; S:30-36 default kwargs */
    double sp[3] = {0.0, is_left ? 0.2 : -0.2, 0.0};
    double so[3] = {is_left ? 15.0 : -15.0, 0.0, is_left ? -10.0 : 10.0};
    double tip[3] = {-0.0, 0.0, 0.10};
    orc_arm_init(arm, is_left, sp, so, 0.28, 0.28, tip, 127.0, 42.5, 1e-8, 0.02, 1e-7, singularity_offset, 1.0);
}

/* ------------------------------------------------------------------ SymbolicIK geometry */
/* S:418-425 */
static void get_wrist_position(const orc_arm_t *arm, const double pos[3], const double eul[3], double wrist[3]) {
    mat4 T = hom(pos, from_euler_xyz_extrinsic(eul));
    double p[4] = {-arm->tip_position[0], arm->tip_position[1], arm->tip_position[2], 1.0}, o[4];
    mat4_apply(T, p, o);
    wrist[0] = o[0]; wrist[1] = o[1]; wrist[2] = o[2];
}

/* S:284-307.  Returns state code (0 == in reach). goal_out = possibly modified position. */
static int is_pose_in_robot_reach(const orc_arm_t *arm, const double pos[3], double goal_out[3]) {
    double gp[3] = {pos[0], pos[1], pos[2]};
    double dv[3] = {pos[0] - arm->shoulder_position[0], pos[1] - arm->shoulder_position[1],
                    pos[2] - arm->shoulder_position[2]};
    double d = norm3(dv);
    int state = ORC_STATE_REACHABLE;
    if (d > arm->max_arm_length) {
        double nd = norm3(dv) + arm->projection_margin;
        for (int i = 0; i < 3; i++) gp[i] = arm->shoulder_position[i] + (dv[i] / nd) * arm->max_arm_length;
        state = ORC_STATE_POSE_OUT_OF_REACH;
    }
    if (gp[0] < arm->backward_limit) {
        gp[0] = arm->backward_limit;
        state = ORC_STATE_BACKWARD_POSE;
    }
    memcpy(goal_out, gp, sizeof gp);
    return state;
}

/* S:337-349: moves sv->wrist_position radially, returns shifted goal position */
static void reduce_goal_pose_no_limits(const orc_arm_t *arm, orc_solver_t *sv, const double pos[3],
                                       double d_shoulder_wrist, double d_max, double pos_out[3]) {
    double nd = fabs(d_shoulder_wrist) + arm->projection_margin;
    double nw[3];
    for (int i = 0; i < 3; i++) {
        double dir = (sv->wrist_position[i] - arm->shoulder_position[i]) / nd;
        nw[i] = arm->shoulder_position[i] + dir * d_max;
    }
    for (int i = 0; i < 3; i++) {
        double diff = nw[i] - sv->wrist_position[i];
        pos_out[i] = pos[i] + diff;
    }
    memcpy(sv->wrist_position, nw, sizeof nw);
}

/* S:366-399.  Returns 0 if no circle (d > u+f). */
static int get_intersection_circle(const orc_arm_t *arm, const orc_solver_t *sv, double center[3], double *radius,
                                   double normal[3]) {
    double P[3] = {sv->wrist_position[0] - arm->shoulder_position[0], sv->wrist_position[1] - arm->shoulder_position[1],
                   sv->wrist_position[2] - arm->shoulder_position[2]};
    double u = arm->upper_arm_size, f = arm->forearm_size;
    double d = sqrt(P[0] * P[0] + P[1] * P[1] + P[2] * P[2]);
    if (d > u + f) return 0;
    double e[3] = {0.0, -asin(P[2] / d), atan2(P[1], P[0])};
    mat3 M = from_euler_xyz_extrinsic(e);
    *radius = 1 / (2 * d) * sqrt(4 * (d * d) * (u * u) - ((d * d) - (f * f) + (u * u)) * ((d * d) - (f * f) + (u * u)));
    double Pc[3] = {((d * d) - (f * f) + (u * u)) / (2 * d), 0, 0}, Psc[3];
    mat3_apply(M, Pc, Psc);
    for (int i = 0; i < 3; i++) center[i] = Psc[i] + arm->shoulder_position[i];
    double ex[3] = {1.0, 0.0, 0.0};
    mat3_apply(M, ex, normal);
    return 1;
}

/* S:401-416 */
static void get_limitation_wrist_circle(const orc_arm_t *arm, const orc_solver_t *sv, const double goal_pos[3],
                                        double center[3], double *radius, double normal[3]) {
    for (int i = 0; i < 3; i++) normal[i] = sv->wrist_position[i] - goal_pos[i];
    *radius = sin(radians(arm->wrist_limit)) * arm->forearm_size;
    double nn = norm3(normal);
    double h = sqrt(arm->forearm_size * arm->forearm_size - (*radius) * (*radius));
    for (int i = 0; i < 3; i++) center[i] = sv->wrist_position[i] + normal[i] / nn * h;
}

/* S:570-586 intersection_point via np.linalg.lstsq on the 3x2 system [v1, -v2] t = p02 - p01.
 * Restated as a thin QR (Gram-Schmidt with one re-orthogonalisation) least-squares solve.  Returns 0 if "empty". */
static int intersection_point(const double v1[3], const double p01[3], const double v2[3], const double p02[3],
                              double out[3]) {
    double a1[3] = {v1[0], v1[1], v1[2]}, a2[3] = {-v2[0], -v2[1], -v2[2]};
    double b[3] = {p02[0] - p01[0], p02[1] - p01[1], p02[2] - p01[2]};
    double r11 = norm3(a1);
    double q1[3] = {a1[0] / r11, a1[1] / r11, a1[2] / r11};
    double r12 = dot3(q1, a2);
    double a2p[3] = {a2[0] - r12 * q1[0], a2[1] - r12 * q1[1], a2[2] - r12 * q1[2]};
    /* one re-orthogonalisation (planes a few 1e-6 apart, just outside normal_vector_margin: the system's condition number is
     * 1 / that, and a single Gram-Schmidt pass leaves that much of q1 in the second column — G18's nearly parallel pairs) */
    double fix = dot3(q1, a2p);
    for (int i = 0; i < 3; i++) a2p[i] -= fix * q1[i];
    r12 += fix;
    double r22 = norm3(a2p);
    double q2[3] = {a2p[0] / r22, a2p[1] / r22, a2p[2] / r22};
    double t2 = dot3(q2, b) / r22;
    double t1 = (dot3(q1, b) - r12 * t2) / r11;
    /* np.all(np.isclose(params, params[0])) */
    if (np_isclose(t1, t1) && np_isclose(t2, t1)) return 0;
    for (int i = 0; i < 3; i++) out[i] = v1[i] * t1 + p01[i];
    return 1;
}

/* S:588-606 */
static int points_of_nearest_approach(const double p1[3], const double n1[3], const double p2[3], const double n2[3],
                                      double q[3], double v[3]) {
    cross3(n1, n2, v);
    double nv = norm3(v);
    for (int i = 0; i < 3; i++) v[i] /= nv;
    double vect1[3], vect2[3];
    cross3(v, n1, vect1);
    cross3(v, n2, vect2);
    return intersection_point(vect1, p1, vect2, p2, q);
}

/* S:608-645.  Returns number of points (0 == None). */
static int intersection_circle_line_3d_vd(const double center[3], double radius, const double direction[3],
                                          const double point_on_line[3], double pts[2][3]) {
    double w[3] = {point_on_line[0] - center[0], point_on_line[1] - center[1], point_on_line[2] - center[2]};
    double a = dot3(direction, direction);
    double b = 2 * dot3(direction, w);
    double c = dot3(w, w) - radius * radius;
    double disc = b * b - 4 * a * c;
    if (disc < 0) return 0;
    if (disc == 0) {
        double t = -b / (2 * a);
        for (int i = 0; i < 3; i++) pts[0][i] = point_on_line[i] + t * direction[i];
        return 1;
    }
    double t1 = (-b + sqrt(disc)) / (2 * a);
    double t2 = (-b - sqrt(disc)) / (2 * a);
    for (int i = 0; i < 3; i++) {
        pts[0][i] = point_on_line[i] + t1 * direction[i];
        pts[1][i] = point_on_line[i] + t2 * direction[i];
    }
    return 2;
}

/* S:511-568 */
static void get_interval_from_intersection(int npts, double pts[2][3], mat4 T_intersection_torso,
                                           mat4 T_torso_intersection, mat4 T_limitation_torso, double radius2,
                                           double interval[2]) {
    if (npts == 1) {
        double p[4] = {pts[0][0], pts[0][1], pts[0][2], 1}, l[4];
        mat4_apply(T_intersection_torso, p, l);
        double angle = atan2(l[2], l[1]);
        interval[0] = angle; interval[1] = angle;
        return;
    }
    double p1[4] = {pts[0][0], pts[0][1], pts[0][2], 1}, p2[4] = {pts[1][0], pts[1][1], pts[1][2], 1}, l1[4], l2[4];
    mat4_apply(T_intersection_torso, p1, l1);
    mat4_apply(T_intersection_torso, p2, l2);
    double angle1 = atan2(l1[2], l1[1]);
    double angle2 = atan2(l2[2], l2[1]);
    if (angle2 < angle1) { double t = angle1; angle1 = angle2; angle2 = t; } /* sorted() */
    double angle_test = (angle1 + angle2) / 2;
    double tp[4] = {0, cos(angle_test) * radius2, sin(angle_test) * radius2, 1}, tt[4], tl[4];
    mat4_apply(T_torso_intersection, tp, tt);
    mat4_apply(T_limitation_torso, tt, tl);
    if (tl[0] > 0) { interval[0] = angle1; interval[1] = angle2; }
    else { interval[0] = angle2; interval[1] = angle1; }
}

/* S:427-509.  Returns 1 and fills interval if the circles are linked, 0 for the empty interval. */
static int are_circles_linked(const orc_arm_t *arm, const orc_solver_t *sv, const double c2[3], double radius2,
                              const double n2_in[3], const double c1[3], double radius1, const double n1_in[3],
                              double interval[2]) {
    double p1[3], p2[3];
    for (int i = 0; i < 3; i++) {
        p1[i] = c1[i] - sv->wrist_position[i];
        p2[i] = c2[i] - sv->wrist_position[i];
    }
    double N1[3] = {n1_in[0], n1_in[1], n1_in[2]}, N2[3] = {n2_in[0], n2_in[1], n2_in[2]};

    mat3 R_torso_intersection = rotation_matrix_from_vector(N2);
    mat4 T_torso_intersection = hom(p2, R_torso_intersection);
    mat3 R_intersection_torso = mat3_T(R_torso_intersection);
    double Pit[3], np2[3] = {p2[0], p2[1], p2[2]};
    { /* np.dot(-R_intersection_torso, p2) */
        mat3 neg = R_intersection_torso;
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) neg.m[i][j] = -neg.m[i][j];
        mat3_apply(neg, np2, Pit);
    }
    mat4 T_intersection_torso = hom(Pit, R_intersection_torso);

    mat3 R_torso_limitation = rotation_matrix_from_vector(N1);
    mat3 R_limitation_torso = mat3_T(R_torso_limitation);
    double Plt[3];
    {
        mat3 neg = R_limitation_torso;
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) neg.m[i][j] = -neg.m[i][j];
        mat3_apply(neg, p1, Plt);
    }
    mat4 T_limitation_torso = hom(Plt, R_limitation_torso);

    double pc2[4] = {p2[0], p2[1], p2[2], 1}, plic[4];
    mat4_apply(T_limitation_torso, pc2, plic);
    int side_ok = plic[0] > 0;

    if (N1[0] != 0 || N1[1] != 0 || N1[2] != 0) { double n = norm3(N1); for (int i = 0; i < 3; i++) N1[i] /= n; }
    if (N2[0] != 0 || N2[1] != 0 || N2[2] != 0) { double n = norm3(N2); for (int i = 0; i < 3; i++) N2[i] /= n; }

    double mg = arm->normal_vector_margin;
    int par_minus = 1, par_plus = 1;
    for (int i = 0; i < 3; i++) {
        if (!(fabs(N2[i] - N1[i]) < mg)) par_minus = 0;
        if (!(fabs(N2[i] + N1[i]) < mg)) par_plus = 0;
    }
    if (par_minus || par_plus) {
        if (side_ok) { interval[0] = -ORC_PI; interval[1] = ORC_PI; return 1; }
        return 0;
    }
    double q[3], v[3];
    if (!points_of_nearest_approach(p1, N1, p2, N2, q, v)) {
        if (side_ok) { interval[0] = -ORC_PI; interval[1] = ORC_PI; return 1; }
        return 0;
    }
    double pts[2][3];
    int npts = intersection_circle_line_3d_vd(p1, radius1, v, q, pts);
    if (npts == 0) {
        if (side_ok) { interval[0] = -ORC_PI; interval[1] = ORC_PI; return 1; }
        return 0;
    }
    get_interval_from_intersection(npts, pts, T_intersection_torso, T_torso_intersection, T_limitation_torso, radius2,
                                   interval);
    return 1;
}

/* S:121-282 */
/* include/rsik.h "Rows that are not numbers": the convention for what the reference answers with an exception */
static int all_finite(const double *v, int n) {
    for (int i = 0; i < n; i++)
        if (!isfinite(v[i])) return 0;
    return 1;
}
static int matrix_is_numbers(const double M[16]) { /* the rotation and the translation: what C:212-216 reads */
    for (int r = 0; r < 3; r++)
        if (!all_finite(&M[4 * r], 4)) return 0;
    return 1;
}

int orc_is_reachable(const orc_arm_t *arm, orc_solver_t *sv, const double pos[3], const double eul[3], int *reachable,
                     double interval[2]) {
    double goal[3];
    *reachable = 0;
    interval[0] = NAN; interval[1] = NAN;
    if (!all_finite(pos, 3) || !all_finite(eul, 3)) return ORC_STATE_INVALID_INPUT; /* self.* untouched */
    int st = is_pose_in_robot_reach(arm, pos, goal);
    if (st != ORC_STATE_REACHABLE) return st; /* S:130-132: self.* untouched */
    memcpy(sv->goal_pos, goal, sizeof goal);
    memcpy(sv->goal_eul, eul, 3 * sizeof(double));
    get_wrist_position(arm, goal, eul, sv->wrist_position);
    if (sv->wrist_position[0] < arm->backward_limit) { /* S:146-153 */
        double diff = arm->backward_limit - sv->wrist_position[0];
        goal[0] = goal[0] + diff;
        sv->wrist_position[0] = sv->wrist_position[0] + diff;
        memcpy(sv->goal_pos, goal, sizeof goal);
    }
    double dv[3] = {sv->wrist_position[0] - arm->shoulder_position[0], sv->wrist_position[1] - arm->shoulder_position[1],
                    sv->wrist_position[2] - arm->shoulder_position[2]};
    double d_shoulder_wrist = norm3(dv);
    if (d_shoulder_wrist > arm->upper_arm_size + arm->forearm_size) return ORC_STATE_WRIST_OUT_OF_RANGE;
    if (d_shoulder_wrist < arm->shoulder_wrist_min_distance) { /* S:166-171 */
        double ng[3];
        reduce_goal_pose_no_limits(arm, sv, goal, d_shoulder_wrist, arm->shoulder_wrist_min_distance, ng);
        memcpy(goal, ng, sizeof ng);
        get_wrist_position(arm, goal, eul, sv->wrist_position);
        memcpy(sv->goal_pos, goal, sizeof goal);
    }
    double c2[3], r2, n2[3], c1[3], r1, n1[3];
    int have = get_intersection_circle(arm, sv, c2, &r2, n2);
    get_limitation_wrist_circle(arm, sv, goal, c1, &r1, n1);
    if (!have) return ORC_STATE_SHOULD_NOT_HAPPEN;
    memcpy(sv->circle_center, c2, sizeof c2);
    sv->circle_radius = r2;
    memcpy(sv->circle_normal, n2, sizeof n2);
    if (are_circles_linked(arm, sv, c2, r2, n2, c1, r1, n1, interval)) {
        *reachable = 1;
        return ORC_STATE_REACHABLE;
    }
    interval[0] = NAN; interval[1] = NAN;
    return ORC_STATE_LIMITED_BY_WRIST;
}

/* S:85-119 */
int orc_is_reachable_no_limits(const orc_arm_t *arm, orc_solver_t *sv, const double pos[3], const double eul[3]) {
    double goal[3];
    (void)is_pose_in_robot_reach(arm, pos, goal);
    memcpy(sv->goal_pos, goal, sizeof goal);
    memcpy(sv->goal_eul, eul, 3 * sizeof(double));
    get_wrist_position(arm, goal, eul, sv->wrist_position);
    if (sv->wrist_position[0] < arm->backward_limit) { /* S:94-98: wrist recomputed, not shifted */
        double diff = arm->backward_limit - sv->wrist_position[0];
        goal[0] = goal[0] + diff;
        get_wrist_position(arm, goal, eul, sv->wrist_position);
        memcpy(sv->goal_pos, goal, sizeof goal);
    }
    double dv[3] = {sv->wrist_position[0] - arm->shoulder_position[0], sv->wrist_position[1] - arm->shoulder_position[1],
                    sv->wrist_position[2] - arm->shoulder_position[2]};
    double d = norm3(dv);
    double uf = arm->upper_arm_size + arm->forearm_size;
    if (d > uf) { /* S:102-105: self.goal_pose updated, local goal_pose NOT (Q4); self.wrist_position moved */
        double ng[3];
        reduce_goal_pose_no_limits(arm, sv, goal, d, uf, ng);
        memcpy(sv->goal_pos, ng, sizeof ng);
    }
    if (d < arm->shoulder_wrist_min_distance) { /* S:107-112 */
        double ng[3];
        reduce_goal_pose_no_limits(arm, sv, goal, d, arm->shoulder_wrist_min_distance, ng);
        memcpy(goal, ng, sizeof ng);
        get_wrist_position(arm, goal, eul, sv->wrist_position);
        memcpy(sv->goal_pos, goal, sizeof goal);
    }
    double c2[3], r2, n2[3];
    if (!get_intersection_circle(arm, sv, c2, &r2, n2)) return 0;
    memcpy(sv->circle_center, c2, sizeof c2);
    sv->circle_radius = r2;
    memcpy(sv->circle_normal, n2, sizeof n2);
    return 1;
}

/* S:684-695 */
void orc_get_elbow_position(const orc_solver_t *sv, double theta, double out[3]) {
    mat4 T = hom(sv->circle_center, rotation_matrix_from_vector(sv->circle_normal));
    double p[4] = {0, sv->circle_radius * cos(theta), sv->circle_radius * sin(theta), 1}, o[4];
    mat4_apply(T, p, o);
    out[0] = o[0]; out[1] = o[1]; out[2] = o[2];
}

/* U:477-483 */
static void make_projection_on_plane(const double P_plane[3], const double nrm[3], const double point[3], double out[3]) {
    double v[3] = {point[0] - P_plane[0], point[1] - P_plane[1], point[2] - P_plane[2]};
    double dist = dot3(v, nrm);
    for (int i = 0; i < 3; i++) out[i] = point[i] - dist * nrm[i];
}

/* S:647-682 */
static void make_elbow_projection(const orc_arm_t *arm, const double goal_pos[3], const double elbow[3],
                                  double new_goal[3], double new_elbow[3]) {
    double alpha = atan2(-arm->singularity_limit_coeff, 1);
    double e[3] = {0, alpha, 0};
    mat3 M_limits = from_euler_xyz_extrinsic(e);
    mat4 T_limits = hom(arm->elbow_singularity_position, M_limits);
    double p0[4] = {0, 0, -arm->singularity_offset, 1}, P_limits[4];
    mat4_apply(T_limits, p0, P_limits);
    T_limits = hom(P_limits, M_limits);
    double n1[4] = {1, 0, 0, 1}, n2[4] = {0, 1, 0, 1};
    mat4_apply(T_limits, n1, n1);
    mat4_apply(T_limits, n2, n2);
    double v1[3] = {n1[0] - P_limits[0], n1[1] - P_limits[1], n1[2] - P_limits[2]};
    double v2[3] = {n2[0] - P_limits[0], n2[1] - P_limits[1], n2[2] - P_limits[2]};
    double v3[3];
    cross3(v1, v2, v3);
    double n3 = norm3(v3);
    for (int i = 0; i < 3; i++) v3[i] /= n3;
    double pc[3];
    make_projection_on_plane(P_limits, v3, arm->shoulder_position, pc);
    double sc[3] = {arm->shoulder_position[0] - pc[0], arm->shoulder_position[1] - pc[1], arm->shoulder_position[2] - pc[2]};
    double nsc = norm3(sc);
    double radius = sqrt(arm->upper_arm_size * arm->upper_arm_size - nsc * nsc);
    double pe[3];
    make_projection_on_plane(P_limits, v3, elbow, pe);
    double V[3] = {pe[0] - pc[0], pe[1] - pc[1], pe[2] - pc[2]};
    double nV = norm3(V);
    for (int i = 0; i < 3; i++) new_elbow[i] = pc[i] + radius * (V[i] / nV);
    for (int i = 0; i < 3; i++) new_goal[i] = goal_pos[i] + (new_elbow[i] - elbow[i]);
}

/* S:697-863 */
int orc_get_joints(const orc_arm_t *arm, orc_solver_t *sv, double theta, const double previous_joints[7],
                   double joints[7], double elbow_out[3]) {
    static const double zero7[7] = {0, 0, 0, 0, 0, 0, 0};
    if (!previous_joints) previous_joints = zero7;
    int projected = 0;
    orc_get_elbow_position(sv, theta, sv->elbow_position);
    if (sv->elbow_position[2] > (sv->elbow_position[0] - arm->elbow_singularity_position[0]) * arm->singularity_limit_coeff +
                                    arm->elbow_singularity_position[2] - arm->singularity_offset) { /* S:708-718 */
        double ng[3], ne[3];
        make_elbow_projection(arm, sv->goal_pos, sv->elbow_position, ng, ne);
        memcpy(sv->goal_pos, ng, sizeof ng);
        memcpy(sv->elbow_position, ne, sizeof ne);
        get_wrist_position(arm, sv->goal_pos, sv->goal_eul, sv->wrist_position);
        projected = 1;
    }
    const double *goal_orientation = sv->goal_eul;
    double P_torso_elbow[4] = {sv->elbow_position[0], sv->elbow_position[1], sv->elbow_position[2], 1};
    double P_torso_wrist[4] = {sv->wrist_position[0], sv->wrist_position[1], sv->wrist_position[2], 1};
    double zero3[3] = {0.0, 0.0, 0.0};

    /* S:728-738 */
    double off[3] = {radians(arm->shoulder_orientation_offset[0]), radians(arm->shoulder_orientation_offset[1]),
                     radians(arm->shoulder_orientation_offset[2])};
    double e_off[3] = {0.0, ORC_PI / 2, 0.0};
    mat3 M_torso_shoulder = mat3_mul(from_euler_xyz_extrinsic(off), from_euler_xyz_extrinsic(e_off));
    mat3 M_shoulder_torso = mat3_T(M_torso_shoulder);
    double P_shoulder_torso[3];
    {
        mat3 neg = M_shoulder_torso;
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) neg.m[i][j] = -neg.m[i][j];
        mat3_apply(neg, arm->shoulder_position, P_shoulder_torso);
    }
    mat4 T_shoulder_torso = hom(P_shoulder_torso, M_shoulder_torso);

    double P_shoulder_elbow[4];
    mat4_apply(T_shoulder_torso, P_torso_elbow, P_shoulder_elbow);
    double shoulder_pitch;
    if (P_shoulder_elbow[0] == 0 && P_shoulder_elbow[2] == 0) shoulder_pitch = previous_joints[0];
    else shoulder_pitch = -atan2(P_shoulder_elbow[2], P_shoulder_elbow[0]);

    double e1[3] = {0.0, -shoulder_pitch, 0.0};
    mat4 T_shoulderPitch_torso = mat4_mul(hom(zero3, from_euler_xyz_extrinsic(e1)), T_shoulder_torso);
    double P_shoulderPitch_elbow[4];
    mat4_apply(T_shoulderPitch_torso, P_torso_elbow, P_shoulderPitch_elbow);
    double shoulder_roll = atan2(P_shoulderPitch_elbow[1], P_shoulderPitch_elbow[0]);

    double e2[3] = {0.0, 0.0, -shoulder_roll};
    mat4 T_elbow_torso = mat4_mul(hom(zero3, from_euler_xyz_extrinsic(e2)), T_shoulderPitch_torso);
    T_elbow_torso.m[0][3] -= arm->upper_arm_size; /* S:776-777 */

    double P_elbow_wrist[4];
    mat4_apply(T_elbow_torso, P_torso_wrist, P_elbow_wrist);
    double elbow_yaw;
    if (P_elbow_wrist[1] == 0 && P_elbow_wrist[2] == 0) elbow_yaw = previous_joints[2];
    else elbow_yaw = -ORC_PI / 2 + atan2(P_elbow_wrist[2], -P_elbow_wrist[1]);

    double e3[3] = {elbow_yaw, 0.0, 0.0};
    mat4 T_elbowYaw_torso = mat4_mul(hom(zero3, from_euler_xyz_extrinsic(e3)), T_elbow_torso);
    double P_elbowYaw_wrist[4];
    mat4_apply(T_elbowYaw_torso, P_torso_wrist, P_elbowYaw_wrist);
    double elbow_pitch = -atan2(P_elbowYaw_wrist[2], P_elbowYaw_wrist[0]);

    double e4[3] = {0.0, -elbow_pitch, 0.0};
    mat4 T_wrist_torso = mat4_mul(hom(zero3, from_euler_xyz_extrinsic(e4)), T_elbowYaw_torso);
    T_wrist_torso.m[0][3] -= arm->forearm_size; /* S:805-806 */

    mat3 M_goal = from_euler_xyz_extrinsic(goal_orientation);
    mat4 T_torso_goalPose = hom(sv->goal_pos, M_goal);
    double P_goalPose_tip[4] = {-arm->tip_position[0], arm->tip_position[1], 0, 1.0}, P_torso_tip[4];
    mat4_apply(T_torso_goalPose, P_goalPose_tip, P_torso_tip);

    double P_wrist_tip[4];
    mat4_apply(T_wrist_torso, P_torso_tip, P_wrist_tip);
    double wrist_roll = ORC_PI - atan2(P_wrist_tip[1], -P_wrist_tip[0]);
    if (wrist_roll > ORC_PI) wrist_roll = wrist_roll - 2 * ORC_PI;

    double e5[3] = {0.0, 0.0, -wrist_roll};
    mat4 T_wristRoll_torso = mat4_mul(hom(zero3, from_euler_xyz_extrinsic(e5)), T_wrist_torso);
    double P_wristRoll_tip[4];
    mat4_apply(T_wristRoll_torso, P_torso_tip, P_wristRoll_tip);
    double wrist_pitch = atan2(P_wristRoll_tip[2], P_wristRoll_tip[0]);

    double e6[3] = {0.0, wrist_pitch, 0.0};
    mat4 T_tip_torso = mat4_mul(hom(zero3, from_euler_xyz_extrinsic(e6)), T_wristRoll_torso);
    T_tip_torso.m[0][3] -= arm->tip_position[2]; /* S:836-837 */

    double P_goal_point[4] = {0.1, 0.0, 0.0, 1.0}, P_torso_point[4], P_tip_point[4];
    mat4 T_torso_goal = hom(P_torso_tip, M_goal);
    mat4_apply(T_torso_goal, P_goal_point, P_torso_point);
    mat4_apply(T_tip_torso, P_torso_point, P_tip_point);
    double wrist_yaw = -atan2(P_tip_point[1], P_tip_point[2]);

    joints[0] = shoulder_pitch; joints[1] = shoulder_roll; joints[2] = elbow_yaw; joints[3] = elbow_pitch;
    joints[4] = wrist_roll; joints[5] = -wrist_pitch; joints[6] = -wrist_yaw;
    double el = radians(arm->elbow_limit);
    if (joints[3] > el) joints[3] = el;
    if (joints[3] < -el) joints[3] = -el;
    memcpy(elbow_out, sv->elbow_position, 3 * sizeof(double));
    return projected;
}

/* ------------------------------------------------------------------ theta selection */
/* U:443-465 (effective predicate, Q10) */
static int is_elbow_ok(const orc_arm_t *arm, const double e[3]) {
    int ok = e[1] * arm->side < -0.2;
    ok = ok && (e[2] < (e[0] - arm->elbow_singularity_position[0]) * arm->singularity_limit_coeff +
                           arm->elbow_singularity_position[2] - arm->singularity_offset);
    return ok;
}

/* U:334-396.  Returns 1 if a theta was found. */
int orc_get_best_discrete_theta(const orc_arm_t *arm, const orc_solver_t *sv, double previous_theta,
                                const double interval[2], int nb, double preferred_theta, double *theta_out) {
    double e[3];
    if (orc_is_valid_angle(preferred_theta, interval)) {
        orc_get_elbow_position(sv, preferred_theta, e);
        if (is_elbow_ok(arm, e)) { *theta_out = preferred_theta; return 1; }
    }
    double a, b;
    if (fabs(fabs(interval[0]) + fabs(interval[1]) - 2 * ORC_PI) < 0.00001) { a = ORC_PI / 2; b = ORC_PI / 2 + 2 * ORC_PI; }
    else if (interval[0] < interval[1]) { a = interval[0]; b = interval[1]; }
    else { a = interval[0]; b = interval[1] + 2 * ORC_PI; }
    /* np.linspace(a, b, nb): y = arange(nb)*step + a, y[-1] = b */
    double step = (b - a) / (double)(nb - 1);
    int found = 0;
    double best_theta = 0.0, best_distance = INFINITY;
    for (int i = 0; i < nb; i++) {
        double theta = (double)i * step + a;
        if (i == nb - 1 && nb > 1) theta = b;
        orc_get_elbow_position(sv, theta, e);
        if (is_elbow_ok(arm, e)) {
            double distance = fabs(orc_angle_diff(theta, preferred_theta));
            if (distance < best_distance) { best_theta = theta; best_distance = distance; found = 1; }
        }
    }
    if (found) { *theta_out = best_theta; return 1; }
    *theta_out = previous_theta;
    return 0;
}

/* ---- the same helpers on explicit arguments, as utils.py exposes them (G18: tests/test_oracle_golden.py) ---- */
/* U:443-465 */
int orc_is_elbow_ok_args(const double e[3], double side, double singularity_offset, double singularity_limit_coeff, const double esp[3]) {
    int ok = e[1] * side < -0.2;
    ok = ok && (e[2] < (e[0] - esp[0]) * singularity_limit_coeff + esp[2] - singularity_offset);
    return ok;
}
/* U:493-505 */
void orc_allow_multiturn(const double new_joints[7], const double prev_joints[7], double out[7]) {
    for (int i = 0; i < 7; i++) out[i] = prev_joints[i] + orc_angle_diff(new_joints[i], prev_joints[i]);
}
/* U:535-568.  Returns the cause bits of include/rsik.h's RSIK_EMERGENCY_* (1 shoulder pitch, 2 elbow yaw, 4 wrist yaw). */
int orc_multiturn_safety_check(const double joints[7], const double limits[3], double out[7]) {
    const int idx[3] = {0, 2, 6};
    int cause = 0;
    memcpy(out, joints, 7 * sizeof(double));
    for (int k = 0; k < 3; k++) {
        if (out[idx[k]] > limits[k]) { out[idx[k]] = limits[k]; cause |= 1 << k; }
        if (out[idx[k]] < -limits[k]) { out[idx[k]] = -limits[k]; cause |= 1 << k; }
    }
    return cause;
}
/* U:571-589.  Returns emergency_stop; out = joints, or previous_joints when not continuous. */
int orc_continuity_check(const double joints[7], const double previous_joints[7], const double max_change[7], double out[7]) {
    int disc = 0;
    for (int i = 0; i < 7; i++)
        if (fabs(orc_angle_diff(joints[i], previous_joints[i])) > max_change[i]) disc = 1;
    memcpy(out, disc ? previous_joints : joints, 7 * sizeof(double));
    return disc;
}
/* U:334-396 on an explicit intersection circle (what the get_elbow_position argument reads, S:684-695).  Returns found;
 * *worked: "preferred_theta worked!". */
int orc_best_discrete_theta_circle(double previous_theta, const double interval[2], int nb, double preferred_theta, double side,
                                   double singularity_offset, double singularity_limit_coeff, const double esp[3], const double circle[7],
                                   double *theta_out, int *worked) {
    orc_solver_t sv;
    memset(&sv, 0, sizeof sv);
    memcpy(sv.circle_center, circle, 3 * sizeof(double));
    sv.circle_radius = circle[3];
    memcpy(sv.circle_normal, circle + 4, 3 * sizeof(double));
    orc_arm_t arm;
    memset(&arm, 0, sizeof arm);
    arm.side = side;
    arm.singularity_offset = singularity_offset;
    arm.singularity_limit_coeff = singularity_limit_coeff;
    memcpy(arm.elbow_singularity_position, esp, 3 * sizeof(double));
    *worked = 0;
    if (orc_is_valid_angle(preferred_theta, interval)) {
        double e[3];
        orc_get_elbow_position(&sv, preferred_theta, e);
        if (is_elbow_ok(&arm, e)) *worked = 1;
    }
    return orc_get_best_discrete_theta(&arm, &sv, previous_theta, interval, nb, preferred_theta, theta_out);
}
/* S:588-606 / S:608-645 on explicit operands */
int orc_points_of_nearest_approach(const double p1[3], const double n1[3], const double p2[3], const double n2[3], double q[3], double v[3]) {
    return points_of_nearest_approach(p1, n1, p2, n2, q, v);
}
int orc_intersection_circle_line(const double center[3], double radius, const double direction[3], const double point_on_line[3], double pts[6]) {
    double p[2][3];
    const int k = intersection_circle_line_3d_vd(center, radius, direction, point_on_line, p);
    for (int i = 0; i < k; i++) memcpy(pts + 3 * i, p[i], 3 * sizeof(double));
    return k;
}

/* ------------------------------------------------------------------ ControlIK */
static int np_allclose_eye3(const double M[16]) { /* np.allclose(M[:3,:3], np.eye(3)) */
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            if (!np_isclose(M[4 * i + j], i == j ? 1.0 : 0.0)) return 0;
    return 1;
}

/* C:212-217 */
static void matrix_to_pose(const double M[16], double pos[3], double eul[3]) {
    pos[0] = M[3]; pos[1] = M[7]; pos[2] = M[11];
    if (np_allclose_eye3(M)) { eul[0] = 0; eul[1] = 0; eul[2] = 0; return; }
    double Rm[9] = {M[0], M[1], M[2], M[4], M[5], M[6], M[8], M[9], M[10]};
    orc_euler_from_matrix_xyz(Rm, eul);
}

/* C:225-252 */
static void interval_limit_for(const orc_arm_t *arm, int constrained_mode, double lim[2], double *preferred_theta) {
    if (constrained_mode == 0) { lim[0] = 3 * ORC_PI / 4; lim[1] = -2 * ORC_PI / 6; }
    else { lim[0] = -4 * ORC_PI / 5; lim[1] = 0; }
    if (arm->side < 0) {
        double a = -ORC_PI - lim[1], b = -ORC_PI - lim[0];
        lim[0] = a; lim[1] = b;
        if (lim[0] < -ORC_PI) lim[0] = orc_pymod(lim[0], 2 * ORC_PI);
        if (lim[1] < -ORC_PI) lim[1] = orc_pymod(lim[1], 2 * ORC_PI);
        if (lim[0] > ORC_PI) lim[0] = orc_pymod(lim[0], -2 * ORC_PI);
        if (lim[1] > ORC_PI) lim[1] = orc_pymod(lim[1], -2 * ORC_PI);
        *preferred_theta = -ORC_PI - *preferred_theta;
    }
}

/* C:225-252 exported for the tests: the control interval (and the mirrored preferred theta) of an arm / mode */
void orc_interval_limit(const orc_arm_t *arm, int constrained_mode, double preferred_theta, double out[3]) {
    interval_limit_for(arm, constrained_mode, out, &preferred_theta);
    out[2] = preferred_theta;
}

/* C:464-497 safety_checks (+ U:493-505 allow_multiturn, U:535-568 multiturn_safety_check) */
static void safety_checks(const double in[7], const double previous_sol[7], double max_angle, double out[7],
                          int *emergency_stop) {
    double j[7];
    memcpy(j, in, sizeof j);
    double w[3];
    orc_limit_orbita3d_joints(&in[4], max_angle, w);
    j[4] = w[0]; j[5] = w[1]; j[6] = w[2];
    for (int i = 0; i < 7; i++) j[i] = previous_sol[i] + orc_angle_diff(j[i], previous_sol[i]);
    const double lim = 6 * ORC_PI;
    const int idx[3] = {0, 2, 6};
    for (int k = 0; k < 3; k++) {
        int i = idx[k];
        if (j[i] > lim) { j[i] = lim; *emergency_stop = 1; }
        if (j[i] < -lim) { j[i] = -lim; *emergency_stop = 1; }
    }
    memcpy(out, j, sizeof j);
}

/* C:162-274 + C:409-462 */
int orc_control_discrete(const orc_arm_t *arm, const double M[16], int nb, double preferred_theta, int constrained_mode,
                         const double previous_sol[7], const double current_joints[7], double previous_theta,
                         double orbita3d_max_angle, double joints[7], int *reachable, int *emergency_stop) {
    double pos[3], eul[3], lim[2];
    if (!matrix_is_numbers(M)) { /* the reference raises (C:215 / S:580): no joints, no verdict on them */
        for (int i = 0; i < 7; i++) joints[i] = NAN;
        *reachable = 0;
        *emergency_stop = 0;
        return ORC_STATE_INVALID_INPUT;
    }
    matrix_to_pose(M, pos, eul);
    interval_limit_for(arm, constrained_mode, lim, &preferred_theta);
    orc_solver_t sv;
    memset(&sv, 0, sizeof sv);
    double interval[2];
    int ok;
    int state = orc_is_reachable(arm, &sv, pos, eul, &ok, interval);
    double theta = 0.0;
    if (ok) {
        ok = orc_get_best_discrete_theta(arm, &sv, previous_theta, interval, nb, preferred_theta, &theta);
        if (!ok) state = ORC_STATE_LIMITED_BY_SHOULDER;
    }
    double raw[7], el[3];
    if (ok) {
        theta = orc_limit_theta_to_interval(theta, previous_theta, lim);
        orc_get_joints(arm, &sv, theta, previous_sol, raw, el);
    } else {
        memcpy(raw, current_joints, sizeof raw);
    }
    *emergency_stop = 0;
    safety_checks(raw, previous_sol, orbita3d_max_angle, joints, emergency_stop);
    *reachable = ok;
    return state;
}

/* U:267-331.  n_current == 7: flat list (C:322-324).  n_current == 14: the constructor's
 * accidental 2x7 broadcast (Q15, C:152-158): only joints[0], joints[1] are compared, each
 * against all 7 entries of row 0 / row 1. */
static double joints_distance(const double joints[7], const double *cur, int n_current) {
    double s = 0.0;
    if (n_current == 7) {
        for (int i = 0; i < 7; i++) { double d = orc_angle_diff(joints[i], cur[i]); s += d * d; }
    } else {
        for (int i = 0; i < 2; i++)
            for (int k = 0; k < 7; k++) { double d = orc_angle_diff(joints[i], cur[7 * i + k]); s += d * d; }
    }
    return sqrt(s);
}

double orc_get_best_theta_to_current_joints(const orc_arm_t *arm, orc_solver_t *sv, const double *current_joints,
                                            int n_current, double preferred_theta) {
    double low = -ORC_PI, high = ORC_PI;
    if (arm->side < 0) { low = 0; high = 2 * ORC_PI; }
    const double tolerance = 0.01;
    double j[7], e[3];
    orc_get_joints(arm, sv, preferred_theta, NULL, j, e);
    if (joints_distance(j, current_joints, n_current) < tolerance) return preferred_theta;
    while ((high - low) > tolerance) {
        double mid1 = low + (high - low) / 3;
        double mid2 = high - (high - low) / 3;
        double j1[7], j2[7];
        orc_get_joints(arm, sv, mid1, NULL, j1, e);
        orc_get_joints(arm, sv, mid2, NULL, j2, e);
        double f1 = joints_distance(j1, current_joints, n_current);
        double f2 = joints_distance(j2, current_joints, n_current);
        if (f1 < f2) high = mid2; else low = mid1;
    }
    double best = (low + high) / 2;
    orc_get_joints(arm, sv, best, NULL, j, e); /* U:324: one more (state-mutating, Q1) call */
    return best;
}

/* U:115-127 */
static double tend_to_preferred_theta(double previous_theta, double d_theta_max, double goal_theta) {
    if (fabs(orc_angle_diff(goal_theta, previous_theta)) < d_theta_max) return goal_theta;
    double ad = orc_angle_diff(goal_theta, previous_theta);
    double sign = ad / fabs(ad);
    return previous_theta + sign * d_theta_max;
}

/* U:220-264.  Returns 1/0 = is_reachable; theta in *theta_out. */
static int get_best_continuous_theta2(const orc_arm_t *arm, const orc_solver_t *sv, double previous_theta,
                                      const double interval[2], int nb, double d_theta_max, double preferred_theta,
                                      double *theta_out) {
    double theta_goal;
    if (!orc_get_best_discrete_theta(arm, sv, previous_theta, interval, nb, preferred_theta, &theta_goal)) {
        *theta_out = previous_theta;
        return 0;
    }
    if (fabs(orc_angle_diff(theta_goal, previous_theta)) < d_theta_max) { *theta_out = theta_goal; return 1; }
    double ad = orc_angle_diff(theta_goal, previous_theta);
    double sign = ad / fabs(ad);
    *theta_out = previous_theta + sign * d_theta_max;
    return 1;
}

/* C:276-407 (+ the wrapper C:212-252).  Returns the state code; 8 = emergency stop latched. */
int orc_control_continuous_step(const orc_arm_t *arm, orc_cont_state_t *cs, const double M[16], int timed_out,
                                double preferred_theta_arg, double preferred_theta_self, int constrained_mode,
                                const double current_joints[7], const double current_pose[16], double d_theta_max,
                                double orbita3d_max_angle, double joints[7], int *reachable) {
    if (cs->emergency_stop != 0.0) { /* C:205-210 */
        memcpy(joints, cs->previous_sol, 7 * sizeof(double));
        *reachable = 0;
        return ORC_STATE_EMERGENCY;
    }
    double pos[3], eul[3], lim[2];
    const int numbers = matrix_is_numbers(M);
    if (numbers) matrix_to_pose(M, pos, eul);
    double pref = preferred_theta_arg;
    interval_limit_for(arm, constrained_mode, lim, &pref);
    int state = ORC_STATE_EMPTY;
    orc_solver_t sv;
    memset(&sv, 0, sizeof sv);
    if (timed_out) { cs->has_previous_sol = 0.0; cs->init = 1.0; }
    if (cs->has_previous_sol == 0.0) { /* C:306-325 */
        memcpy(cs->previous_sol, current_joints, 7 * sizeof(double));
        cs->has_previous_sol = 1.0;
        double cpos[3], ceul[3];
        matrix_to_pose(current_pose, cpos, ceul);
        orc_is_reachable_no_limits(arm, &sv, cpos, ceul);
        cs->previous_theta = orc_get_best_theta_to_current_joints(arm, &sv, current_joints, 7, pref);
    }
    if (!numbers) {
        /* include/rsik.h "Rows that are not numbers" (the start-up branch above does not read the goal and has run): no joints;
         * previous_sol, init and the latch stay; previous_theta takes the step of a search that found nothing (U:252-264 with
         * goal = previous_theta, then U:93-112) */
        cs->previous_theta = orc_limit_theta_to_interval(cs->previous_theta, cs->previous_theta, lim);
        for (int i = 0; i < 7; i++) joints[i] = NAN;
        *reachable = 0;
        return ORC_STATE_INVALID_INPUT;
    }
    double interval[2];
    int ok;
    int st_reach = orc_is_reachable(arm, &sv, pos, eul, &ok, interval);
    double theta, raw[7], el[3];
    if (ok) {
        ok = get_best_continuous_theta2(arm, &sv, cs->previous_theta, interval, 10, d_theta_max, preferred_theta_self, &theta);
        if (!ok) state = ORC_STATE_LIMITED_BY_SHOULDER;
        theta = orc_limit_theta_to_interval(theta, cs->previous_theta, lim);
        cs->previous_theta = theta;
        orc_get_joints(arm, &sv, theta, cs->previous_sol, raw, el);
    } else {
        /* C:371-373 "always True" with ControlIK's own solvers; false needs a solver whose projection_margin pushes the
         * wrist beyond u + f (symbolic_ik.py:343-345): the reference then raises RuntimeError (C:385-387) before it
         * touches previous_theta / previous_sol / init -> reported as ORC_STATE_NOT_REACHABLE_NO_LIMITS, state unchanged
         * but for the start-up branch above (which has run, as in the reference). */
        if (!orc_is_reachable_no_limits(arm, &sv, pos, eul)) {
            for (int i = 0; i < 7; i++) joints[i] = NAN;
            *reachable = 0;
            return ORC_STATE_NOT_REACHABLE_NO_LIMITS;
        }
        theta = tend_to_preferred_theta(cs->previous_theta, d_theta_max, pref);
        theta = orc_limit_theta_to_interval(theta, cs->previous_theta, lim);
        cs->previous_theta = theta;
        orc_get_joints(arm, &sv, theta, cs->previous_sol, raw, el);
        state = st_reach;
    }
    int estop = 0;
    safety_checks(raw, cs->previous_sol, orbita3d_max_angle, joints, &estop);
    if (estop) cs->emergency_stop = 1.0;
    if (cs->init == 0.0) { /* U:571-589 continuity_check */
        static const double maxc[7] = {0.5, 0.5, 0.5, 0.5, 1.0, 1.0, 1.0};
        int disc = 0;
        for (int i = 0; i < 7; i++)
            if (fabs(orc_angle_diff(joints[i], cs->previous_sol[i])) > maxc[i]) disc = 1;
        if (disc) {
            cs->emergency_stop = 1.0;
            memcpy(joints, cs->previous_sol, 7 * sizeof(double));
        }
    }
    cs->init = 0.0;
    if (cs->emergency_stop == 0.0) memcpy(cs->previous_sol, joints, 7 * sizeof(double));
    *reachable = ok;
    return state;
}

/* ------------------------------------------------------------------ batch drivers */
int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_solve_batch(const orc_arm_t *arm_r, const orc_arm_t *arm_l, long n, const double *px, const double *py,
                     const double *pz, const double *ex, const double *ey, const double *ez, const uint8_t *arm_id,
                     int theta_policy, const double *theta_in, const double *previous_joints, double *joints,
                     double *interval, double *elbow, uint8_t *reachable, uint8_t *state, uint8_t *projected,
                     int nthreads) {
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
#else
    (void)nthreads;
#endif
    for (long i = 0; i < n; i++) {
        const orc_arm_t *arm = (arm_id && arm_id[i]) ? arm_l : arm_r;
        orc_solver_t sv;
        memset(&sv, 0, sizeof sv);
        double pos[3] = {px[i], py[i], pz[i]}, eul[3] = {ex[i], ey[i], ez[i]};
        double itv[2], j[7], el[3];
        int ok, proj = 0;
        int st = orc_is_reachable(arm, &sv, pos, eul, &ok, itv);
        for (int k = 0; k < 7; k++) j[k] = NAN;
        el[0] = el[1] = el[2] = NAN;
        if (ok) {
            double theta;
            if (theta_policy == 1) theta = theta_in[i];
            else if (theta_policy == 2) {
                double a = itv[0], b = itv[1];
                if (a > b) b += 2 * ORC_PI;
                theta = a + theta_in[i] * (b - a);
            } else theta = itv[0];
            proj = orc_get_joints(arm, &sv, theta, previous_joints, j, el);
        }
        if (joints) memcpy(&joints[7 * i], j, sizeof j);
        if (interval) { interval[2 * i] = itv[0]; interval[2 * i + 1] = itv[1]; }
        if (elbow) memcpy(&elbow[3 * i], el, sizeof el);
        if (reachable) reachable[i] = (uint8_t)ok;
        if (state) state[i] = (uint8_t)st;
        if (projected) projected[i] = (uint8_t)proj;
    }
}

/* Trajectory batch of the continuous mode (BASELINE config 5): n_traj independent trajectories (one orc_cont_state_t
 * each, updated in place), n_steps control steps each, step s of trajectory k at M[(s * n_traj + k) * 16].  The first
 * step of every trajectory is called with `timed_out` (C:296-304) and `current_pose` (16 doubles, the same for every trajectory:
 * what a ControlIK holds in previous_pose, C:38-57, 294) or, with NULL, the trajectory's own first matrix; current_joints = the
 * state's previous_sol.  joints [n_steps][n_traj][7], reachable / state
 * [n_steps][n_traj].  Trajectories are independent: OpenMP over k. */
void orc_control_continuous_run_batch(const orc_arm_t *arm, orc_cont_state_t *cs, long n_traj, long n_steps, const double *M,
                                      int first_step_timed_out, double preferred_theta_arg, double preferred_theta_self,
                                      int constrained_mode, double d_theta_max, double orbita3d_max_angle, double *joints,
                                      uint8_t *reachable, uint8_t *state, int nthreads, const double *current_pose) {
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
#else
    (void)nthreads;
#endif
    for (long k = 0; k < n_traj; k++) {
        for (long s_ = 0; s_ < n_steps; s_++) {
            const double *Mk = &M[(s_ * n_traj + k) * 16];
            double cur[7];
            memcpy(cur, cs[k].previous_sol, sizeof cur);
            int ok;
            int st = orc_control_continuous_step(arm, &cs[k], Mk, (s_ == 0 && first_step_timed_out) ? 1 : 0, preferred_theta_arg,
                                                 preferred_theta_self, constrained_mode, cur, current_pose ? current_pose : &M[k * 16], d_theta_max,
                                                 orbita3d_max_angle, &joints[(s_ * n_traj + k) * 7], &ok);
            reachable[s_ * n_traj + k] = (uint8_t)ok;
            state[s_ * n_traj + k] = (uint8_t)st;
        }
    }
}

void orc_control_discrete_batch(const orc_arm_t *arm_r, const orc_arm_t *arm_l, long n, const double *M,
                                const uint8_t *arm_id, int nb, double preferred_theta, int constrained_mode,
                                const double *previous_sol_2x7, const double *current_joints, double orbita3d_max_angle,
                                double *joints, uint8_t *reachable, uint8_t *state, uint8_t *emergency, int nthreads) {
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
#else
    (void)nthreads;
#endif
    for (long i = 0; i < n; i++) {
        int a = (arm_id && arm_id[i]) ? 1 : 0;
        const orc_arm_t *arm = a ? arm_l : arm_r;
        const double *prev = &previous_sol_2x7[7 * a];
        const double *cur = current_joints ? &current_joints[7 * i] : prev;
        int ok, es;
        int st = orc_control_discrete(arm, &M[16 * i], nb, preferred_theta, constrained_mode, prev, cur, 0.0,
                                      orbita3d_max_angle, &joints[7 * i], &ok, &es);
        reachable[i] = (uint8_t)ok;
        state[i] = (uint8_t)st;
        if (emergency) emergency[i] = (uint8_t)es;
    }
}
