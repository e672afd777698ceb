/*
 * rsik.h — C ABI of the MI355X-native batched analytic IK solver for the Reachy 2 arms.
 *
 * This is the drop-in boundary for ONE path of pollen-robotics/reachy2_symbolic_ik:
 *   SymbolicIK.is_reachable + theta_to_joints_func (= SymbolicIK.get_joints)
 *   ControlIK.symbolic_inverse_kinematics, control_type "discrete" (and "continuous" as state-carrying steps)
 *
 * The reference has no FFI: its boundary is the Python API of two classes
 * (src/reachy2_symbolic_ik/symbolic_ik.py:25-863, control_ik.py:27-497).  Each entry point below
 * cites the reference interface it replaces.  Plain C: pointers and sizes only, no C++/torch types.
 *
 * Conventions
 *   - Every function returns an int status: RSIK_OK (0) or a negative RSIK_E_* code;
 *     rsik_last_error(ctx) gives the message.  Unreachable poses are DATA (reachable/state
 *     arrays), never errors — the reference returns (False, [], None, state) rather than raising
 *     (symbolic_ik.py:130-132,161,263).
 *   - All `const double*` / `double*` / `uint8_t*` batch arguments are DEVICE pointers (HBM) owned
 *     by the caller unless the name ends in `_host`.  The library never frees or retains them.
 *   - Work is enqueued on the context's HIP stream (rsik_set_stream) and is asynchronous;
 *     rsik_sync waits for it.
 *   - Streams: calls of one context may be issued on different streams (rsik_set_stream between them).  The continuous run keeps
 *     a workspace, dependency words and side streams in the context: a run issued on another stream than the run before it
 *     waits for that run's end first.  hipGraphs recorded from ONE context's continuous runs share that workspace: replay
 *     them one at a time (or record them from different contexts).
 *   - Threads: a context is NOT thread-safe (it holds the stream, the arm constants, the options and the
 *     continuous pipeline's workspace that the next call uses; the reference's objects are not re-entrant either,
 *     symbolic_ik.py:143-144,185).  Use it from one thread at a time; several contexts — one per thread, per stream or
 *     per GPU — are independent and may be used concurrently.
 *   - Angles in radians, lengths in metres, everything IEEE float64.
 *   - Rows of unreachable poses in joints / interval / elbow are filled with NaN.
 */
#ifndef RSIK_H
#define RSIK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSIK_ABI_VERSION 7

/* ---- status codes ---- */
#define RSIK_OK 0
#define RSIK_E_INVALID (-1)   /* bad argument (null pointer, bad enum, n < 0 ...) */
#define RSIK_E_NO_DEVICE (-2) /* no usable HIP device / device id out of range */
#define RSIK_E_HIP (-3)       /* a HIP runtime call failed; see rsik_last_error */
#define RSIK_E_NOT_SET (-4)   /* arm constants were not uploaded for an arm the call needs */

/* ---- per-pose state codes (uint8); strings are the reference's own ---- */
#define RSIK_STATE_REACHABLE 0           /* "reachable"                         symbolic_ik.py:234 */
#define RSIK_STATE_POSE_OUT_OF_REACH 1   /* "Pose out of reach"                 symbolic_ik.py:300 */
#define RSIK_STATE_BACKWARD_POSE 2       /* "Backward pose"                     symbolic_ik.py:306 */
#define RSIK_STATE_WRIST_OUT_OF_RANGE 3  /* "wrist out of range"                symbolic_ik.py:159 */
#define RSIK_STATE_LIMITED_BY_WRIST 4    /* "limited by wrist"                  symbolic_ik.py:262 */
#define RSIK_STATE_SHOULD_NOT_HAPPEN 5   /* "out of reach - should not happen"  symbolic_ik.py:281 */
#define RSIK_STATE_LIMITED_BY_SHOULDER 6 /* "limited by shoulder"               control_ik.py:363,452 */
#define RSIK_STATE_EMPTY 7               /* ""  (continuous mode, reachable)    control_ik.py:297 */
#define RSIK_STATE_EMERGENCY 8           /* emergency stop latched              control_ik.py:205-210 */
#define RSIK_STATE_NOT_REACHABLE_NO_LIMITS 9 /* continuous mode: is_reachable_no_limits failed, where the reference raises
                                              * RuntimeError("Pose not reachable in symbolic IK. ...") control_ik.py:385-387.
                                              * Joints NaN, trajectory state untouched.  Needs projection_margin <= 0. */
#define RSIK_STATE_INVALID_INPUT 10       /* the row's goal (pose or matrix) holds a NaN or an infinity: see "Rows that are not
                                          * numbers" below.  Where the reference raises numpy.linalg.LinAlgError
                                          * (symbolic_ik.py:580, reached through the comparisons of :284-307, which are all
                                          * false for a NaN) or ValueError (scipy's Rotation.from_matrix, control_ik.py:215). */

/* ---- Rows that are not numbers ----
 * A batch is data: a row with a NaN or an infinity in it is answered in that row, never with an error code, never by stalling,
 * and it changes nothing in any other row — every other row's outputs (and, in the continuous entry points, its trajectory
 * state) are bit for bit those of a run in which the bad row held an ordinary goal.
 *
 *  the goal — the six pose entries of rsik_solve / rsik_reach_state, the twelve matrix entries (rotation and translation) of
 *  rsik_control_discrete / rsik_control_continuous_step / rsik_control_continuous_run — holds a NaN or +-infinity:
 *      state RSIK_STATE_INVALID_INPUT, reachable 0, joints / elbow / interval NaN, emergency 0; rsik_reach_state leaves the
 *      solver-state row's geometry as it was.  (The reference raises for a NaN anywhere and for an infinite angle; for an
 *      infinite POSITION its is_reachable returns "Pose out of reach" / "Backward pose" with a projected pose that is itself
 *      NaN, symbolic_ik.py:292-307, and the control entry points raise on it a few lines later: one code for all of them.)
 *      Continuous mode, a step with such a goal: the latch comes first as always (a latched trajectory reports
 *      RSIK_STATE_EMERGENCY and previous_sol); otherwise the step reports the code and NaN joints, previous_sol / init / the
 *      latch stay as they were — the next good step is judged against the last good one — and previous_theta takes the step
 *      of a search that found nothing (utils.py:252-264 with goal = previous_theta, then limit_theta_to_interval: it stays
 *      where it is once inside the control interval).  The start-up branch of a timed-out step (control_ik.py:296-325) does
 *      not read the goal and runs before this.
 *  anything else that is not a number — theta_in, previous_joints, current_joints, current_pose, a row of cont_state or of a
 *  solver state: IEEE arithmetic as the reference would do it: the outputs of THAT row which depend on the value are NaN or
 *  what comparisons that are all false select (a NaN never trips an emergency stop, a NaN previous_theta stays NaN); the state
 *  code is one of the codes above.  A poisoned cont_state row stays poisoned until the caller re-initialises it (timed_out). */

/* ---- why an emergency stop tripped: one bit per message the reference appends to ControlIK.emergency_state
 * (utils.multiturn_safety_check utils.py:544-566, utils.continuity_check utils.py:584-586) ---- */
#define RSIK_EMERGENCY_SHOULDER_PITCH 1  /* "EMERGENCY STOP: shoulder pitch limit reached"  joint 0 beyond +-6 pi */
#define RSIK_EMERGENCY_ELBOW_YAW 2       /* "EMERGENCY STOP: elbow yaw limit reached"       joint 2 */
#define RSIK_EMERGENCY_WRIST_YAW 4       /* "EMERGENCY STOP: wrist yaw limit reached"       joint 6 */
#define RSIK_EMERGENCY_CONTINUITY 8      /* " EMERGENCY STOP: joints are not continuous ..." continuous mode only */

/* ---- arms ---- */
#define RSIK_ARM_R 0
#define RSIK_ARM_L 1

/* ---- theta policies for rsik_solve (which elbow angle get_joints is evaluated at) ---- */
#define RSIK_THETA_INTERVAL0 0 /* theta = interval[0]  (README.md:85, ik_benchmarks.py:27-31) */
#define RSIK_THETA_EXPLICIT 1  /* theta = theta_in[i] */
#define RSIK_THETA_FRACTION 2  /* theta = i0 + theta_in[i] * (i1' - i0), i1' = i1 (+2pi if wrapped) */
#define RSIK_THETA_NONE 3      /* is_reachable only: joints/elbow are not written */

/* ---- constrained modes (control_ik.py:225-232) ---- */
#define RSIK_MODE_UNCONSTRAINED 0
#define RSIK_MODE_LOW_ELBOW 1

/*
 * Per-arm constant block: a flat array of RSIK_ARM_CONSTS_COUNT doubles computed once on the host
 * from SymbolicIK.__init__'s arguments (symbolic_ik.py:26-83, utils.py:26-43) plus the
 * pose-independent parts of get_joints / make_elbow_projection (symbolic_ik.py:728-738, 653-672).
 * Offsets:
 */
enum {
    RSIK_C_SHOULDER = 0,      /* [3] shoulder_position                                   */
    RSIK_C_UPPER_ARM = 3,     /* upper_arm_size u                                         */
    RSIK_C_FOREARM = 4,       /* forearm_size f                                           */
    RSIK_C_TIPL = 5,          /* [3] wrist offset in the goal frame (-tip_x, tip_y, tip_z) symbolic_ik.py:422 */
    RSIK_C_MAX_LEN = 8,       /* max_arm_length                                           */
    RSIK_C_MIN_DIST = 9,      /* shoulder_wrist_min_distance                              */
    RSIK_C_BACKWARD = 10,     /* backward_limit                                           */
    RSIK_C_PROJ_MARGIN = 11,  /* projection_margin                                        */
    RSIK_C_NORMAL_MARGIN = 12,/* normal_vector_margin                                     */
    RSIK_C_UPF = 13,          /* u + f                                                    */
    RSIK_C_WRIST_R = 14,      /* sin(radians(wrist_limit)) * f          symbolic_ik.py:413 */
    RSIK_C_WRIST_AX = 15,     /* sqrt(f^2 - r^2)                        symbolic_ik.py:414 */
    RSIK_C_MST = 16,          /* [9] M_shoulder_torso row-major         symbolic_ik.py:728-736 */
    RSIK_C_TSH = 25,          /* [3] P_shoulder_torso = -M_shoulder_torso . s  symbolic_ik.py:737 */
    RSIK_C_ES = 28,           /* [3] elbow_singularity_position                           */
    RSIK_C_SING_OFFSET = 31,  /* singularity_offset                                       */
    RSIK_C_SING_COEFF = 32,   /* singularity_limit_coeff                                  */
    RSIK_C_ELBOW_LIMIT = 33,  /* radians(elbow_limit)                                     */
    RSIK_C_SIDE = 34,         /* +1 r_arm, -1 l_arm                                       */
    RSIK_C_PLANE_P = 35,      /* [3] P_limits                           symbolic_ik.py:657 */
    RSIK_C_PLANE_N = 38,      /* [3] v3 (unit plane normal)             symbolic_ik.py:668-669 */
    RSIK_C_PROJ_CENTER = 41,  /* [3] projected_center                   symbolic_ik.py:671 */
    RSIK_C_PROJ_RADIUS = 44,  /* radius (NaN if the plane misses the shoulder sphere)  symbolic_ik.py:672 */
    RSIK_C_TIP_Z = 45,        /* tip_position[2]                        symbolic_ik.py:837 */
    RSIK_C_INV_U = 46,        /* 1 / upper_arm_size        (lengths that hold by construction) */
    RSIK_C_INV_F = 47,        /* 1 / forearm_size                                          */
    RSIK_C_INV_TIPZ = 48,     /* 1 / |tip_position[2]|                                     */
    RSIK_C_INV_GRIP = 49,     /* 1 / gripper_size = 1 / |tip_position|                     */
    RSIK_C_MAX_LEN_SQ = 50,   /* largest double x with sqrt(x) <= max_arm_length: |v| > max_arm_length <=> v.v > x, bit for bit */
    RSIK_C_INV_MIN_DIST = 51, /* 1 / shoulder_wrist_min_distance                           */
    RSIK_C_PLANE_K = 52,      /* es_z - singularity_offset - singularity_limit_coeff * es_x: the elbow is above the singularity
                                 plane (symbolic_ik.py:708-713) iff e_z > coeff * e_x + this */
    RSIK_ARM_CONSTS_COUNT = 53
};

typedef struct rsik_ctx rsik_ctx;

/* ---- lifecycle ---- */
int rsik_abi_version(void);
/* "RSIK_SRC_HASH=<32 hex digits>": hash of the sources and flags the library was built from (csrc/build.py), so a host
 * can refuse a stale binary.  Not in the reference (pure Python, nothing to build). */
const char *rsik_build_id(void);
int rsik_arm_consts_count(void);
/* Number of HIP devices visible; 0 when there is none (never fails). */
int rsik_device_count(void);
/* Creates a context bound to one GPU.  Replaces constructing SymbolicIK / ControlIK objects
 * (symbolic_ik.py:26, control_ik.py:28): constants are uploaded separately with rsik_set_arm. */
int rsik_create(int device_id, rsik_ctx **out);
int rsik_destroy(rsik_ctx *ctx);
/* Message of the last failing call on this context ("" if none).  ctx == NULL: last rsik_create failure. */
const char *rsik_last_error(const rsik_ctx *ctx);
/* Use the caller's hipStream_t (e.g. torch's current stream) for all launches; NULL = default stream. */
int rsik_set_stream(rsik_ctx *ctx, void *hip_stream);
/* Waits for the context's stream; also frees the workspaces that continuous runs have outgrown since the last call, and reports
 * (RSIK_E_HIP) a theta kernel of an overlapping continuous run that gave up its bounded wait for its prepare kernel (cannot happen:
 * everything it waits for is issued before it; bounded so that it cannot hang the device either). */
int rsik_sync(rsik_ctx *ctx);
/* Uploads one arm's constant block (host pointer).  Replaces SymbolicIK.__init__ (symbolic_ik.py:26-83). */
int rsik_set_arm(rsik_ctx *ctx, int arm, const double *consts_host, int count);

/* ---- per-context options ---- */
/* RSIK_OPT_EULER_ROUNDTRIP.  ControlIK turns the goal matrix into extrinsic xyz Euler angles
 * (utils.get_euler_from_homogeneous_matrix, utils.py:84-90, called at control_ik.py:215) and SymbolicIK rebuilds the
 * rotation from them (symbolic_ik.py:420).  For a proper rotation matrix away from gimbal lock that round trip is the
 * identity to rounding.  RSIK_EULER_AUTO (default): the control kernels consume M[:3,:3] directly and make the round
 * trip (SciPy's nearest-rotation / matrix -> quaternion -> Euler algorithms) only where it changes the result — the
 * matrix is not orthonormal to 1e-12, or the pitch is within 1e-5 of +-pi/2 (at the lock: yaw := 0).
 * RSIK_EULER_ALWAYS / RSIK_EULER_NEVER force either behaviour. */
#define RSIK_EULER_AUTO 0
#define RSIK_EULER_ALWAYS 1
#define RSIK_EULER_NEVER 2
#define RSIK_OPT_EULER_ROUNDTRIP 0
/* Kernel-variant selectors.  Every value gives the same results (to rounding where a comment says so); the defaults
 * (0) let the library choose.  They exist so that tests can force each variant in turn and A/B timings can pin one.
 *   RSIK_OPT_SWEEP_MODE     rsik_control_discrete's grid search (utils.py:366-396): 0 = chosen per wave, 1 = always the
 *                           exhaustive wave-cooperative sweep, 2 = always the per-lane search
 *   RSIK_OPT_NO_TIPZ        non-zero: never use the goal stage specialised for tip_x = tip_y = 0
 *   RSIK_OPT_NO_MIRROR      non-zero: mixed r/l launches read every constant per lane (no mirror-image shortcut)
 *   RSIK_OPT_CONT_RUN_MODE  rsik_control_continuous_run: 0 = chosen by the library, 1 = the phased trajectory pipeline
 *                           (kernels per block of steps on four streams), 2 = one launch of the step kernel per control
 *                           step, exactly what n_steps calls of rsik_control_continuous_step issue */
#define RSIK_OPT_SWEEP_MODE 1
#define RSIK_OPT_NO_TIPZ 2
#define RSIK_OPT_NO_MIRROR 3
#define RSIK_OPT_CONT_RUN_MODE 4
#define RSIK_CONT_RUN_AUTO 0
#define RSIK_CONT_RUN_PHASED 1
#define RSIK_CONT_RUN_STEPS 2
/* Tuning of rsik_control_continuous_run (results do not depend on it):
 *   RSIK_OPT_CONT_BLOCK_STEPS  control steps per block: 0 (default) = a third of the run, at least 64 and — launch by launch — at most
 *                              512 (half of the run under capture); n > 0 = n (rounded up to the sequential phases' batch of steps) */
#define RSIK_OPT_CONT_BLOCK_STEPS 5
/*   RSIK_OPT_CONT_PHASED_VARIANT  phased pipeline issued launch by launch, a bit mask (0 = the default form; results do not depend on it):
 *                              1  its streams tied by hipEvents (as a run recorded into a hipGraph always is) instead of by
 *                                 hipStreamWriteValue32 / hipStreamWaitValue32 on words in device memory
 *                              2  the joints kernel of a block NOT held until the theta kernel of the next block has started
 *                              4 / 8 / 16  (runs that overlap, RSIK_OPT_CONT_GOALS_RESIDENT) the next run's prepare kernels not held at all /
 *                                 held until the previous run's last chain kernel has FINISHED / its theta kernels behind stream waits
 *                                 instead of waiting for their prepare kernels themselves — the steps docs/experiments.md R6.2 measures
 *                              64 the joints phase without its turn hints (R6.3; joints then differ in their last bits) */
#define RSIK_OPT_CONT_PHASED_VARIANT 6
#define RSIK_PHASED_EDGES_BY_EVENT 1
#define RSIK_PHASED_NO_THETA_FIRST 2
/*   RSIK_OPT_CONT_GOALS_RESIDENT  rsik_control_continuous_run issued launch by launch, run after run on one stream (results do not
 *                              depend on it).  0 (default): every run's four streams meet at its start and at its end — a run begins
 *                              when everything queued ahead of it, the previous run included, has finished.  1: a PROMISE by the caller
 *                              that lets consecutive runs of one shape overlap: the goal matrices `m12_steps` (and `arm`) handed to a
 *                              run were complete before the PREVIOUS continuous run of this context was issued, and nothing the caller
 *                              queued on the stream since that call reads or writes this run's `reachable_steps` / `state_steps` (they
 *                              may be the previous run's own buffers: each block's rows are then written behind that run's chain kernel
 *                              of the same rows).  The prepare phase of run k + 1 — a function of the goals alone — then runs beside the
 *                              joints and chain kernels of run k, in the workspace slots run k does not use (all eight are kept); the
 *                              start-up kernel, the theta, joints and chain phases of run k + 1 wait for run k's end as before (the
 *                              trajectory state: control_ik.py:80-83, 276-407 carries it from call to call), and the caller's stream
 *                              continues behind each run's last kernel as before.  Takes effect behind a run of the same n, block
 *                              length and stream, issued with value-word edges into the same workspace, on a context no hipGraph
 *                              points into; otherwise the run is issued as with 0 (rsik_control_continuous_last_form tells). */
#define RSIK_OPT_CONT_GOALS_RESIDENT 7
#define RSIK_OPT_COUNT 8
int rsik_set_option(rsik_ctx *ctx, int option, int value);
int rsik_get_option(const rsik_ctx *ctx, int option, int *value);

/* ---- device memory helpers for hosts without their own allocator (torch users do not need them) ---- */
int rsik_malloc(rsik_ctx *ctx, size_t bytes, void **dev_ptr);
int rsik_free(rsik_ctx *ctx, void *dev_ptr);
int rsik_memcpy_h2d(rsik_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int rsik_memcpy_d2h(rsik_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);

/*
 * rsik_solve — fused SymbolicIK.is_reachable (symbolic_ik.py:121-282) + one call of the returned
 * theta_to_joints_func = SymbolicIK.get_joints (symbolic_ik.py:697-863) per pose, canonical
 * "fresh is_reachable, then exactly one get_joints" semantics.
 *
 *   n            number of poses
 *   pose_soa     6 device arrays of n doubles: px, py, pz, roll, pitch, yaw — the reference's
 *                goal_pose = [[x,y,z],[roll,pitch,yaw]], extrinsic xyz Euler (README.md:73-75)
 *   arm          device array of n uint8 (RSIK_ARM_R / RSIK_ARM_L) or NULL
 *   arm_uniform  arm used for every pose when arm == NULL
 *   theta_policy RSIK_THETA_*;  theta_in: device array of n doubles for EXPLICIT / FRACTION, else NULL
 *   previous_joints_host  7 doubles (host) or NULL for zeros — get_joints' previous_joints argument
 *   joints       [n,7] row-major or NULL      (shoulder_pitch, shoulder_roll, elbow_yaw, elbow_pitch,
 *                                               wrist_roll, wrist_pitch, wrist_yaw)
 *   interval     [n,2] or NULL                 theta interval, i0 > i1 means wrap-around (README.md:83-84)
 *   elbow        [n,3] or NULL                 elbow position returned by get_joints
 *   reachable    [n] uint8 or NULL
 *   state        [n] uint8 RSIK_STATE_* or NULL
 */
int rsik_solve(rsik_ctx *ctx, int64_t n, const double *const pose_soa[6], const uint8_t *arm, int arm_uniform,
               int theta_policy, const double *theta_in, const double *previous_joints_host, double *joints,
               double *interval, double *elbow, uint8_t *reachable, uint8_t *state);

/*
 * rsik_control_discrete — ControlIK.symbolic_inverse_kinematics(name, M, "discrete", ...)
 * (control_ik.py:162-274 -> symbolic_inverse_kinematics_discrete :409-462 -> safety_checks :464-497),
 * which is a pure function of its inputs and constructor-time constants.
 *
 *   m12_soa          12 device arrays of n doubles: R00,R01,R02,R10,...,R22 (row-major rotation of the
 *                    4x4 goal matrix M) then tx,ty,tz (M[:3,3])
 *   arm/arm_uniform  as above (name "r_arm"/"l_arm")
 *   nb_search_points ControlIK.nb_search_points (control_ik.py:64), >= 2
 *   preferred_theta  the preferred_theta argument (r-arm convention; mirrored for l inside, control_ik.py:252)
 *   constrained_mode RSIK_MODE_*
 *   previous_sol_host 14 doubles: ControlIK.previous_sol["r_arm"], ["l_arm"] (control_ik.py:136,140)
 *   current_joints   [n,7] device or NULL (=> previous_sol of the pose's arm, control_ik.py:237-238)
 *   orbita3d_max_angle  wrist cone half-angle in radians (control_ik.py:84)
 *   joints [n,7], reachable [n], state [n] as above; emergency [n] uint8 or NULL: non-zero where
 *   multiturn_safety_check tripped (utils.py:535-568), as RSIK_EMERGENCY_* cause bits.
 */
int rsik_control_discrete(rsik_ctx *ctx, int64_t n, const double *const m12_soa[12], const uint8_t *arm,
                          int arm_uniform, int nb_search_points, double preferred_theta, int constrained_mode,
                          const double *previous_sol_host, const double *current_joints, double orbita3d_max_angle,
                          double *joints, uint8_t *reachable, uint8_t *state, uint8_t *emergency);

/*
 * rsik_control_continuous_step — ControlIK.symbolic_inverse_kinematics(name, M, "continuous", ...)
 * (control_ik.py:162-274 -> symbolic_inverse_kinematics_continuous :276-407) for n independent trajectories,
 * one control step per call.  The reference keeps previous_theta / previous_sol / init / emergency_stop on the
 * ControlIK object (control_ik.py:60-83); here they live in a caller-owned device array
 * cont_state[RSIK_CONT_STATE_ROWS][n] (SoA): row 0 previous_theta, rows 1-7 previous_sol, row 8 init (0/1),
 * row 9 emergency_stop (0/1), row 10 has_previous_sol (0/1); written only by the step that trips an emergency stop:
 * row 11 its RSIK_EMERGENCY_* cause bits and rows 12-18 the joints that failed the continuity check (the "joints" of
 * the reference's message, utils.py:584-586; its "previous_joints" are rows 1-7).
 *
 *   m12_soa               goal matrices of this step (layout as rsik_control_discrete)
 *   current_pose_m12_soa  current_pose of trajectories that (re)initialise this step, or NULL (=> the goal matrix)
 *   timed_out             [n] uint8 or NULL: 1 replaces the reference's wall-clock test
 *                         abs(t - last_call_t) > call_timeout (control_ik.py:296-304)
 *   preferred_theta       the preferred_theta argument (r convention; used by the init search and the unreachable branch)
 *   preferred_theta_self_host  2 doubles: ControlIK.preferred_theta["r_arm"], ["l_arm"] (control_ik.py:133-139),
 *                         used by the reachable branch (Q14)
 *   current_joints        [n,7] device or NULL (=> previous_sol)
 *   state codes           RSIK_STATE_EMPTY when reachable, RSIK_STATE_LIMITED_BY_SHOULDER, the is_reachable state when
 *                         unreachable, RSIK_STATE_EMERGENCY while the emergency stop is latched.
 */
#define RSIK_CONT_STATE_ROWS 19
int rsik_control_continuous_step(rsik_ctx *ctx, int64_t n, const double *const m12_soa[12],
                                 const double *const current_pose_m12_soa[12], const uint8_t *arm, int arm_uniform,
                                 const uint8_t *timed_out, double preferred_theta, const double *preferred_theta_self_host,
                                 int constrained_mode, double d_theta_max, const double *current_joints,
                                 double orbita3d_max_angle, double *cont_state, double *joints, uint8_t *reachable,
                                 uint8_t *state);

/*
 * rsik_control_continuous_run — n_steps consecutive control steps of n trajectories from one host call (the whole
 * trajectory batch resident in HBM).  The result equals n_steps calls of rsik_control_continuous_step (flags, state
 * codes and the carried theta bit for bit, joints to the last bits: see phase 3), but the work is not done step by
 * step: most of a control step does not depend on the previous one — the goal conversion, is_reachable /
 * is_reachable_no_limits and the 10-point search for the target theta (control_ik.py:327-388 up to the rate limiter)
 * are functions of the pose alone — so the batch is solved in four phases per block of steps:
 *   1. prepare   one thread per (step, trajectory), chip-filling: is_reachable + the search for the target theta; the
 *                step's goal for the rate limiter (the search's theta / the preferred theta / "stay") -> workspace
 *   2. theta     one thread per trajectory, sequential over steps: the d_theta_max rate limiter and
 *                limit_theta_to_interval (the only recurrence on previous_theta)
 *   3. joints    one thread per (step, trajectory), a wave = 8 consecutive steps of 8 trajectories: the circle of
 *                is_reachable / is_reachable_no_limits re-derived from the goal matrix, get_joints at the limited theta,
 *                the Orbita3D cone clamp, and allow_multiturn INSIDE the 8-step chunk (whole turns relative to the step
 *                before, a shuffle prefix sum); continuity / limit / singularity events are detected, not decided
 *   4. chain     eight lanes per trajectory (one per joint), sequential over CHUNKS: checks each chunk's first step
 *                against previous_sol (continuity_check, the +-6 pi clamp, the emergency latch) and finds the whole
 *                turns the chunk sits away from it, which it adds to the chunk's rows where they are not zero (fp64 atomic
 *                adds that nobody waits for); only a chunk with an event is walked step by step with the reference's own
 *                sequence of operations (also: steps whose get_joints hit an exact singularity)
 * A run is cut into blocks of steps: three — of at most 512 steps — when it is issued launch by launch, two when it is being captured
 * into a hipGraph (RSIK_OPT_CONT_BLOCK_STEPS overrides).  Flags, state codes, the carried theta and the latch do not depend on the cut,
 * nor do the joints of a run of up to eight blocks; beyond that the joints can differ in their last bits from cut to cut (each is
 * within 2 ulp of the step kernel's: from the ninth block on phase 3 writes a wound trajectory's rows where phase 4 left previous_sol
 * eight blocks earlier, one rounding instead of two).  Runs of one shape issued one after the other can overlap:
 * RSIK_OPT_CONT_GOALS_RESIDENT.
 * The phases of neighbouring blocks overlap on four streams (the caller's and three of the context's).  Issued launch by
 * launch the streams are tied by words in device memory (hipStreamWriteValue32 behind the producer, hipStreamWaitValue32
 * ahead of the consumer: a third of an event's latency) and a block's joints kernel is held until the NEXT block's theta
 * kernel has started (its lone 276-register waves cannot get onto a chip that a chip-filling kernel holds); while the
 * caller's stream is capturing, by events, which is all a capture takes.
 * A trajectory whose emergency stop is latched (control_ik.py:205-210: previous_sol, not reachable, the emergency state for every
 * goal until "unfreeze") is not walked by phase 4: its steps of a block are filled in at once — on entry where it was latched before
 * the block, behind the chunk in which it latches otherwise.
 * The workspace
 * (17 bytes per step and trajectory + 1 per 8-step chunk, of up to eight blocks in flight, + 16 per trajectory), the side streams and the
 * events belong to the context: they are created by the first call that needs them, or ahead of time by
 * rsik_control_continuous_reserve.  A call can be captured into a hipGraph (the side streams join the capture through
 * the events the call records) provided it has nothing to create: reserve first, or run a call of at least that size
 * first — otherwise the call fails with RSIK_E_INVALID instead of allocating inside the capture.  A captured graph
 * stays valid after later, larger calls (the workspace it points into is kept until rsik_destroy / _release; an
 * outgrown workspace that only runs already issued can use is freed by the next rsik_sync).  The call never waits for
 * the device — with one exception in 4e9 launch-by-launch runs of a context: the words carry a 32-bit run number, and before it
 * wraps the call drains what the context has issued and starts them over.  A solver whose
 * projection_margin is not positive (RSIK_STATE_NOT_REACHABLE_NO_LIMITS possible) is run step by step.  At most 30 Mi
 * trajectories per call.  4096 trajectories x 1000 steps: see DESIGN.md section 4.
 * A goal that is not numbers: see "Rows that are not numbers" above (the step is reported, the trajectory goes on).
 *   m12_steps        device [n_steps][12][n]: the goal matrices of every step
 *   current_pose_m12_soa / current_joints   used by the first step only (see rsik_control_continuous_step)
 *   first_step_timed_out  non-zero: every trajectory (re)initialises on the first step (the reference's behaviour for
 *                    the first call after construction, control_ik.py:160,298)
 *   joints_steps     device [n_steps][n][7]; reachable_steps / state_steps device [n_steps][n] or NULL
 */
int rsik_control_continuous_run(rsik_ctx *ctx, int64_t n, int64_t n_steps, const double *m12_steps,
                                const double *const current_pose_m12_soa[12], const uint8_t *arm, int arm_uniform,
                                int first_step_timed_out, double preferred_theta, const double *preferred_theta_self_host,
                                int constrained_mode, double d_theta_max, const double *current_joints,
                                double orbita3d_max_angle, double *cont_state, double *joints_steps,
                                uint8_t *reachable_steps, uint8_t *state_steps);

/*
 * rsik_control_continuous_last_form — how the last rsik_control_continuous_run of this context was issued (nothing is returned
 * through the run's own arguments: the results do not depend on the form, the cost does).  RSIK_CONT_FORM_STEPS_NO_LIMITS_CAN_FAIL:
 * the solver's projection_margin is not positive, so is_reachable_no_limits can fail (symbolic_ik.py:343-345; the reference then
 * raises on purpose, control_ik.py:385-387, here RSIK_STATE_NOT_REACHABLE_NO_LIMITS) — an outcome the pipeline's phases do not carry:
 * the run was n_steps launches of the step kernel whatever RSIK_OPT_CONT_RUN_MODE said, 10-30 x the pipeline's time.
 */
#define RSIK_CONT_FORM_NONE 0                      /* no run yet */
#define RSIK_CONT_FORM_PHASED 1                    /* the trajectory pipeline, launch by launch */
#define RSIK_CONT_FORM_PHASED_OVERLAPPED 2         /* ... its prepare phase started beside the previous run (RSIK_OPT_CONT_GOALS_RESIDENT) */
#define RSIK_CONT_FORM_PHASED_CAPTURED 3           /* ... recorded into a hipGraph */
#define RSIK_CONT_FORM_STEPS 4                     /* one launch of the step kernel per control step, as RSIK_OPT_CONT_RUN_MODE asked */
#define RSIK_CONT_FORM_STEPS_NO_LIMITS_CAN_FAIL 5  /* ... because the arm's projection margin is not positive */
int rsik_control_continuous_last_form(const rsik_ctx *ctx);

/*
 * rsik_control_continuous_reserve — creates what rsik_control_continuous_run(n, n_steps) would create on first use
 * (workspace, side streams, events), so that the run itself allocates nothing: call it before capturing such a run into
 * a hipGraph on a context that has not yet run one of that size.  Uses RSIK_OPT_CONT_BLOCK_STEPS as set at the time.
 */
int rsik_control_continuous_reserve(rsik_ctx *ctx, int64_t n, int64_t n_steps);

/*
 * rsik_control_continuous_release — waits for the device and frees everything rsik_control_continuous_run keeps in the
 * context between calls: the workspace and the workspaces it has outgrown.  After it, hipGraphs captured from this
 * context's continuous runs must not be replayed any more.  (The context otherwise holds one workspace, of the largest
 * run so far, grown geometrically.)
 */
int rsik_control_continuous_release(rsik_ctx *ctx);

/*
 * rsik_stage — the stages of SymbolicIK.is_reachable as the reference exposes them: public methods that take their operands
 * as arguments (and self.wrist_position, which a caller may assign), timed one by one by src/benchmark/ik_benchmarks.py:36-130.
 * The fused kernels never form these intermediates; this entry point does, with the reference's own sequence of operations, on
 * whatever operands it is given.  Row-major: row i reads in[i * in_stride ...], writes out[i * out_stride ...] (device pointers,
 * or pinned host memory the device can address); `arm`: whose constants (RSIK_ARM_R / RSIK_ARM_L).  Per row, in -> out:
 *   RSIK_STAGE_POSE_IN_REACH        is_pose_in_robot_reach symbolic_ik.py:284-307: position 3, euler 3 -> in reach 0/1, position 3 (pulled
 *                                   back / clamped), state code (RSIK_STATE_EMPTY = the reference's "")
 *   RSIK_STAGE_WRIST_POSITION       get_wrist_position :418-425: position 3, euler 3 -> wrist 3
 *   RSIK_STAGE_LIMITATION_CIRCLE    get_limitation_wrist_circle :401-416: wrist 3, goal position 3 -> centre 3, radius, normal 3 (not normalised)
 *   RSIK_STAGE_INTERSECTION_CIRCLE  get_intersection_circle :366-399: wrist 3 -> found 0/1 (0 = None), centre 3, radius, normal 3
 *   RSIK_STAGE_CIRCLES_LINKED       are_circles_linked :427-568: wrist 3, intersection circle (centre 3, radius, normal 3), limitation circle
 *                                   (the same seven) -> how many numbers the returned array holds (0 or 2), the interval 2
 *   RSIK_STAGE_NEAREST_APPROACH     points_of_nearest_approach :588-606: p1 3, normal1 3, p2 3, normal2 3 -> q found 0/1 (0 = []), q 3, v 3
 *   RSIK_STAGE_CIRCLE_LINE          intersection_circle_line_3d_vd :608-645: centre 3, radius, direction 3, point on line 3 -> points 0/1/2, point 3, point 3
 *   RSIK_STAGE_ROTATION_FROM_VECTOR utils.rotation_matrix_from_vector utils.py:59-81: vector 3 -> 3 x 3 row-major
 * and the policy layer's helpers as utils.py exposes them — scalar functions of explicit arguments that callers of the reference import
 * (src/example/test_ik.py:16-21, test_go_to.py:10-13); `arm` is not read by these (nor by the three stages before them: a context
 * no arm was uploaded to runs them):
 *   RSIK_STAGE_ANGLE_DIFF           angle_diff utils.py:486-490: a, b -> the difference in [-pi, pi)
 *   RSIK_STAGE_IS_VALID_ANGLE       is_valid_angle :468-474: angle, interval 2 -> 0/1
 *   RSIK_STAGE_LIMIT_THETA_TO_INTERVAL  limit_theta_to_interval :93-112: theta, previous_theta, interval 2 -> theta, "theta in interval" 0/1
 *   RSIK_STAGE_IS_ELBOW_OK          is_elbow_ok :443-465: elbow 3, side (+1 r / -1 l), singularity_offset, singularity_limit_coeff,
 *                                   elbow_singularity_position 3 -> 0/1
 *   RSIK_STAGE_ALLOW_MULTITURN      allow_multiturn :493-505: new joints 7, previous joints 7 -> joints 7
 *   RSIK_STAGE_LIMIT_ORBITA3D_JOINTS  limit_orbita3d_joints :508-519: roll, pitch, yaw (intrinsic XYZ), max angle -> the three angles in the cone
 *   RSIK_STAGE_MULTITURN_SAFETY_CHECK  multiturn_safety_check :535-568: joints 7, the three limits -> joints 7, RSIK_EMERGENCY_* bits
 *   RSIK_STAGE_CONTINUITY_CHECK     continuity_check :571-589: joints 7, previous joints 7, max angular change 7 -> joints 7, stop 0/1
 *   RSIK_STAGE_BEST_DISCRETE_THETA  get_best_discrete_theta :334-396: previous_theta, interval 2, nb_search_points, preferred_theta, side,
 *                                   singularity_offset, singularity_limit_coeff, elbow_singularity_position 3, the intersection circle its
 *                                   get_elbow_position argument reads (centre 3, radius, normal 3) -> found 0/1, theta, preferred worked 0/1
 *                                   (nb_search_points outside [0, 2^20] or not a number: an empty grid)
 */
#define RSIK_STAGE_POSE_IN_REACH 0
#define RSIK_STAGE_WRIST_POSITION 1
#define RSIK_STAGE_LIMITATION_CIRCLE 2
#define RSIK_STAGE_INTERSECTION_CIRCLE 3
#define RSIK_STAGE_CIRCLES_LINKED 4
#define RSIK_STAGE_NEAREST_APPROACH 5
#define RSIK_STAGE_CIRCLE_LINE 6
#define RSIK_STAGE_ROTATION_FROM_VECTOR 7
#define RSIK_STAGE_ANGLE_DIFF 8
#define RSIK_STAGE_IS_VALID_ANGLE 9
#define RSIK_STAGE_LIMIT_THETA_TO_INTERVAL 10
#define RSIK_STAGE_IS_ELBOW_OK 11
#define RSIK_STAGE_ALLOW_MULTITURN 12
#define RSIK_STAGE_LIMIT_ORBITA3D_JOINTS 13
#define RSIK_STAGE_MULTITURN_SAFETY_CHECK 14
#define RSIK_STAGE_CONTINUITY_CHECK 15
#define RSIK_STAGE_BEST_DISCRETE_THETA 16
#define RSIK_STAGE_COUNT 17
int rsik_stage(rsik_ctx *ctx, int op, int64_t n, int arm, const double *in, int in_stride, double *out, int out_stride);

/*
 * Solver-state entry points: the scalar drop-in API.  A SymbolicIK object keeps self.goal_pose,
 * self.wrist_position and self.intersection_circle between is_reachable() and the closure it returns
 * (symbolic_ik.py:143-144,185,235), and get_joints() mutates them when the elbow projection fires
 * (symbolic_ik.py:714-718).  That state lives in a caller-owned device array, one row of
 * RSIK_SOLVER_STATE_STRIDE doubles per solver instance:
 *   0-2 goal position, 3-5 goal euler, 6-8 wrist position, 9-11 circle centre, 12 circle radius,
 *   13-15 circle normal, 16-18 elbow position of the last get_joints, 19 projection flag,
 *   20-21 interval, 22 reachable (0/1), 23 state code of the last is_reachable (so that a scalar caller can fetch a
 *   call's results and the updated state with one download), 24-30 joints of the last get_joints, 31 reserved.
 */
#define RSIK_SOLVER_STATE_STRIDE 32

/* SymbolicIK.is_reachable (no_limits == 0, symbolic_ik.py:121-282) or is_reachable_no_limits
 * (no_limits != 0, symbolic_ik.py:85-119).  Only the fields the reference would have assigned are
 * written into solver_state (an early "Pose out of reach" leaves the row untouched). */
int rsik_reach_state(rsik_ctx *ctx, int64_t n, const double *const pose_soa[6], const uint8_t *arm, int arm_uniform,
                     int no_limits, double *solver_state, double *interval, uint8_t *reachable, uint8_t *state);
/* SymbolicIK.get_joints(theta, previous_joints) on stored state (symbolic_ik.py:697-863); updates the
 * row like the reference updates self.  previous_joints: [n,7] device or NULL for zeros.  joints / elbow may be NULL
 * (the row's slots 24-30 / 16-18 carry them too). */
int rsik_joints_from_state(rsik_ctx *ctx, int64_t n, double *solver_state, const uint8_t *arm, int arm_uniform,
                           const double *theta, const double *previous_joints, double *joints, double *elbow);
/* SymbolicIK.get_elbow_position(theta) on stored state (symbolic_ik.py:684-695). */
int rsik_elbow_from_state(rsik_ctx *ctx, int64_t n, const double *solver_state, const double *theta, double *elbow);

/* utils.get_euler_from_homogeneous_matrix for a batch (utils.py:84-90): goal matrices (m12_soa, layout as
 * rsik_control_discrete) -> pose_soa = 6 device arrays px,py,pz,roll,pitch,yaw (the input layout of rsik_solve),
 * Euler angles = Rotation.from_matrix(M[:3,:3]).as_euler("xyz").  identity_shortcut != 0 adds ControlIK's
 * np.allclose(M[:3,:3], I) => [0,0,0] (control_ik.py:212-214). */
int rsik_matrix_to_pose(rsik_ctx *ctx, int64_t n, const double *const m12_soa[12], int identity_shortcut,
                        double *const pose_soa[6]);

/* Forward kinematics of the arm: the chain SymbolicIK.get_joints inverts (symbolic_ik.py:728-848):
 * joints [n,7] -> goal position [n,3] and goal rotation [n,9] (row-major), torso frame.  Either output may be NULL.
 * The reference has no FK (its examples use placo/reachy2_sdk for that, src/example/test_dk.py); this is the
 * self-contained monitor of SURVEY 8 f-4. */
int rsik_forward_kinematics(rsik_ctx *ctx, int64_t n, const double *joints, const uint8_t *arm, int arm_uniform,
                            double *position, double *rotation);
/* FK(joints) against the goal it was solved for: err[n,2] = (|position error| in m, rotation error in rad).
 * goal_soa: 6 columns (px,py,pz,roll,pitch,yaw) for RSIK_GOAL_POSE6, 12 columns (R row-major, t) for RSIK_GOAL_M12.
 * Rows whose joints are NaN (unreachable poses) give NaN. */
#define RSIK_GOAL_POSE6 0
#define RSIK_GOAL_M12 1
int rsik_fk_residual(rsik_ctx *ctx, int64_t n, int goal_kind, const double *const *goal_soa, const double *joints,
                     const uint8_t *arm, int arm_uniform, double *err);

/*
 * Multi-GPU.  Poses are independent, so a job is sharded across the GPUs of a node — one process and one
 * rsik context per GPU — with no data-path collective; the only exchange is the all-gather of the final arrays
 * (joints [n,7], state [n]) over RCCL / xGMI.  The reference has nothing distributed (single-threaded Python).  Python
 * hosts use torch.distributed (backend "nccl" = RCCL, reachy2_symbolic_ik_amd/distributed.py); these four entry points
 * give a C host the same step without linking RCCL itself (librccl.so is opened with dlopen on first use).
 *   rsik_comm_unique_id   fills 128 bytes (ncclUniqueId) on ONE rank; the host carries them to the others
 *   rsik_comm_init_rank   collective over all ranks: creates this rank's communicator on the context's GPU
 *   rsik_allgather        every rank contributes bytes_per_rank bytes at `send`; rank r's block lands at
 *                         recv + r * bytes_per_rank on every rank (send may be recv + rank * bytes_per_rank: in place).
 *                         Enqueued on the context's stream like the kernels; device pointers.
 */
int rsik_comm_unique_id(void *id128);
int rsik_comm_init_rank(rsik_ctx *ctx, int nranks, int rank, const void *id128, void **comm);
int rsik_comm_destroy(rsik_ctx *ctx, void *comm);
int rsik_allgather(rsik_ctx *ctx, void *comm, const void *send, void *recv, size_t bytes_per_rank);

/* Test hook: evaluates the kernels' own elementary functions (csrc/rsik_math.hpp) on device arrays so their
 * accuracy can be measured against the host libm.  op: 0 reciprocal, 1 sqrt (out0, out1 two variants),
 * 2 reciprocal sqrt, 3 atan2(a, b), 4 sincos(a) -> out0 = sin, out1 = cos, 5 out0 = a mod 2pi (Python
 * semantics), out1 = angle_diff(a, b) (utils.py:486-490), 6 fp64 FMA issue-rate calibration (8 x 2048 dependent fma per element, scripts/valu_peak.py),
 * 7 the solve path's table atan2 of a UNIT vector: out0 = atan2(a, b) for a^2 + b^2 = 1,
 * 8 clock monitor: n waves each wait a[0] ticks of the 100 MHz counter (clamped to 5 s); out0[w] = shader-clock ticks,
 *   out1[w] = 100 MHz ticks that passed (core clock = out0 / out1 x 100 MHz); run it on a side stream beside a load.
 * Not part of the reference surface. */
int rsik_debug_math(rsik_ctx *ctx, int op, int64_t n, const double *a, const double *b, double *out0, double *out1);

#ifdef __cplusplus
}
#endif
#endif /* RSIK_H */
