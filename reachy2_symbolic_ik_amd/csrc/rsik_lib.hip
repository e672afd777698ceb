// rsik_lib.hip — the C ABI of include/rsik.h: context, argument checks and launches.  The kernels (gfx950) live in the
// rsik_kernel_*.hpp files next to it, the per-pose mathematics in rsik_device.hpp / rsik_math.hpp.
//
// Kernel shape: one pose per lane, 256-thread workgroups (4 wave64), SoA float64 inputs so that
// every global load is a fully coalesced 512-B wave access; the [n,7] / [n,3] row outputs are
// transposed through LDS so that each wave writes its 3584-B / 1536-B slab with unit-stride stores.
// Per-arm constants travel in the kernarg segment (scalar loads, wave-uniform).
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "rsik_device.hpp"

// the kernels, by entry point (one translation unit)
#include "rsik_kernel_solve.hpp"
#include "rsik_kernel_discrete.hpp"
#include "rsik_kernel_continuous.hpp"
#include "rsik_kernel_pipeline.hpp"
#include "rsik_kernel_state.hpp"
#include "rsik_kernel_stages.hpp"

// =====================================================================================
// C ABI
// =====================================================================================
struct rsik_ctx {
    int device;
    int compute_units;  // of the device (256 on MI355X): which launches are a single round
    hipStream_t stream;
    bool have_arm[2];
    rsik::ArmC arms[2];
    int options[RSIK_OPT_COUNT];
    void* ws;          // workspace of rsik_control_continuous_run (device: the pipeline's block slots), grown on demand
    size_t ws_bytes;
    bool ws_captured;                // a run recorded into a hipGraph points into the current workspace
    std::vector<void*> retired_ws;   // outgrown workspaces a captured hipGraph may still point into: kept until rsik_destroy / _release
    std::vector<void*> outgrown_ws;  // outgrown workspaces only runs already issued can use: freed by the next rsik_sync / _release / rsik_destroy
    hipEvent_t run_done;             // recorded behind every continuous run issued launch by launch: the next run, if it comes on
    hipStream_t run_stream;          // ANOTHER stream, waits for it (the workspace, the words and the side streams are the context's)
    bool have_run_done;
    unsigned* edge_words;            // the phased pipeline's dependency words (device): see cont_edges
    size_t edge_count;
    unsigned edge_seq;               // runs issued with them: the value a word must reach
    int can_wait_value;              // hipDeviceAttributeCanUseStreamWaitValue
    hipStream_t side[3];             // the pipeline's own streams (prepare / joints / chain), created on first use
    // Bookkeeping across continuous runs issued launch by launch with value-word edges (RSIK_OPT_CONT_GOALS_RESIDENT, see
    // rsik_control_continuous_run): which chain kernel used each workspace slot last, and what the last run wrote besides.
    struct SlotUse { size_t word; unsigned seq; } slot_use[8];  // seq 0: nobody since the streams last met
    int slot_next;                   // the slot the next overlapped run's first block takes
    struct LastRun {
        bool valid;                  // a phased run issued launch by launch with value words; nothing since has made it useless
        unsigned seq;
        hipStream_t stream;
        const void* ws;
        const unsigned* words;
        int64_t n, n_steps, T, n_blocks;
        size_t slot_bytes;
        int slots;
        const uint8_t *state_lo, *state_hi, *reach_lo, *reach_hi;  // the rows its prepare and chain kernels wrote
    } last_run;
    int last_run_form;               // RSIK_CONT_FORM_* of the last rsik_control_continuous_run (rsik_control_continuous_last_form)
    std::vector<hipEvent_t> events;  // reusable, timing disabled
    bool have_side;
    std::string err;
};

static thread_local std::string g_create_err;

static int fail(rsik_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    else g_create_err = msg;
    return code;
}
static int hip_fail(rsik_ctx* ctx, hipError_t e, const char* what) {
    return fail(ctx, RSIK_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define RSIK_HIP(ctx, call)                                   \
    do {                                                      \
        hipError_t e_ = (call);                               \
        if (e_ != hipSuccess) return hip_fail(ctx, e_, #call); \
    } while (0)

// the run number at which the words that tie the streams of rsik_control_continuous_run start over (a test build sets it to a handful)
#ifndef RSIK_EDGE_SEQ_WRAP
#define RSIK_EDGE_SEQ_WRAP 0xfffffff0u
#endif

extern "C" {

int rsik_abi_version(void) { return RSIK_ABI_VERSION; }
#ifndef RSIK_SOURCE_HASH
#define RSIK_SOURCE_HASH "00000000000000000000000000000000"
#endif
const char* rsik_build_id(void) { return "RSIK_SRC_HASH=" RSIK_SOURCE_HASH; }
int rsik_arm_consts_count(void) { return RSIK_ARM_CONSTS_COUNT; }

int rsik_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int rsik_create(int device_id, rsik_ctx** out) {
    if (!out) return fail(nullptr, RSIK_E_INVALID, "rsik_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(nullptr, RSIK_E_NO_DEVICE, "rsik_create: no HIP device available");
    if (device_id < 0 || device_id >= n) return fail(nullptr, RSIK_E_NO_DEVICE, "rsik_create: device id out of range");
    rsik_ctx* c = new (std::nothrow) rsik_ctx();
    if (!c) return fail(nullptr, RSIK_E_INVALID, "rsik_create: out of host memory");
    c->device = device_id;
    c->compute_units = 256;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cus > 0) c->compute_units = cus;
    }
    c->stream = nullptr;
    c->have_arm[0] = c->have_arm[1] = false;
    for (int k = 0; k < RSIK_OPT_COUNT; k++) c->options[k] = 0;
    c->ws = nullptr;
    c->ws_bytes = 0;
    c->ws_captured = false;
    c->have_run_done = false;
    c->run_stream = nullptr;
    c->edge_words = nullptr;
    c->edge_count = 0;
    c->edge_seq = 0;
    c->can_wait_value = 0;
    (void)hipDeviceGetAttribute(&c->can_wait_value, hipDeviceAttributeCanUseStreamWaitValue, device_id);
    c->have_side = false;
    for (auto& u : c->slot_use) u = {0, 0};
    c->slot_next = 0;
    c->last_run = {};
    c->last_run_form = RSIK_CONT_FORM_NONE;
    for (auto& st : c->side) st = nullptr;
    *out = c;
    return RSIK_OK;
}

int rsik_destroy(rsik_ctx* ctx) {
    if (ctx && hipSetDevice(ctx->device) == hipSuccess) {
        if (ctx->ws) (void)hipFree(ctx->ws);
        if (ctx->edge_words) (void)hipFree(ctx->edge_words);
        if (ctx->have_run_done) (void)hipEventDestroy(ctx->run_done);
        for (void* w : ctx->retired_ws) (void)hipFree(w);
        for (void* w : ctx->outgrown_ws) (void)hipFree(w);
        for (hipEvent_t e : ctx->events) (void)hipEventDestroy(e);
        if (ctx->have_side)
            for (hipStream_t st : ctx->side) (void)hipStreamDestroy(st);
    }
    delete ctx;
    return RSIK_OK;
}

const char* rsik_last_error(const rsik_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int rsik_set_stream(rsik_ctx* ctx, void* hip_stream) {
    if (!ctx) return RSIK_E_INVALID;
    ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
    return RSIK_OK;
}

int rsik_sync(rsik_ctx* ctx) {
    if (!ctx) return RSIK_E_INVALID;
    RSIK_HIP(ctx, hipSetDevice(ctx->device));
    RSIK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // a theta kernel that gave up waiting for its prepare kernel (cannot happen; the wait is bounded so that it cannot hang either)
    if (ctx->edge_words) {
        unsigned gave_up = 0;
        RSIK_HIP(ctx, hipMemcpyAsync(&gave_up, ctx->edge_words + 3, sizeof gave_up, hipMemcpyDeviceToHost, ctx->stream));
        RSIK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (gave_up != 0) {
            (void)hipMemsetAsync(ctx->edge_words + 3, 0, sizeof gave_up, ctx->stream);
            (void)hipStreamSynchronize(ctx->stream);
            return fail(ctx, RSIK_E_HIP, "rsik_sync: a theta kernel of rsik_control_continuous_run waited a second for its prepare kernel and went on without it: the results of that run are invalid");
        }
    }
    // workspaces that continuous runs outgrew: whatever was issued into them has finished now
    if (!ctx->outgrown_ws.empty()) {
        if (ctx->have_run_done) RSIK_HIP(ctx, hipEventSynchronize(ctx->run_done));
        for (void* w : ctx->outgrown_ws) (void)hipFree(w);
        ctx->outgrown_ws.clear();
    }
    return RSIK_OK;
}

int rsik_set_arm(rsik_ctx* ctx, int arm, const double* consts_host, int count) {
    if (!ctx) return RSIK_E_INVALID;
    if (arm != RSIK_ARM_R && arm != RSIK_ARM_L) return fail(ctx, RSIK_E_INVALID, "rsik_set_arm: arm must be 0 (r) or 1 (l)");
    if (!consts_host || count != RSIK_ARM_CONSTS_COUNT)
        return fail(ctx, RSIK_E_INVALID, "rsik_set_arm: expected RSIK_ARM_CONSTS_COUNT doubles");
    std::memcpy(ctx->arms[arm].v, consts_host, sizeof(double) * RSIK_ARM_CONSTS_COUNT);
    ctx->have_arm[arm] = true;
    return RSIK_OK;
}

int rsik_set_option(rsik_ctx* ctx, int option, int value) {
    if (!ctx) return RSIK_E_INVALID;
    if (option < 0 || option >= RSIK_OPT_COUNT) return fail(ctx, RSIK_E_INVALID, "rsik_set_option: unknown option");
    static const int max_value[RSIK_OPT_COUNT] = {RSIK_EULER_NEVER, 2, 1, 1, RSIK_CONT_RUN_STEPS, 65535, 127, 1};
    if (value < 0 || value > max_value[option]) return fail(ctx, RSIK_E_INVALID, "rsik_set_option: value out of range");
    ctx->options[option] = value;
    return RSIK_OK;
}
int rsik_get_option(const rsik_ctx* ctx, int option, int* value) {
    if (!ctx || !value || option < 0 || option >= RSIK_OPT_COUNT) return RSIK_E_INVALID;
    *value = ctx->options[option];
    return RSIK_OK;
}

int rsik_malloc(rsik_ctx* ctx, size_t bytes, void** dev_ptr) {
    if (!ctx || !dev_ptr) return RSIK_E_INVALID;
    RSIK_HIP(ctx, hipSetDevice(ctx->device));
    RSIK_HIP(ctx, hipMalloc(dev_ptr, bytes));
    return RSIK_OK;
}
int rsik_free(rsik_ctx* ctx, void* dev_ptr) {
    if (!ctx) return RSIK_E_INVALID;
    RSIK_HIP(ctx, hipSetDevice(ctx->device));
    RSIK_HIP(ctx, hipFree(dev_ptr));
    return RSIK_OK;
}
int rsik_memcpy_h2d(rsik_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes) {
    if (!ctx) return RSIK_E_INVALID;
    RSIK_HIP(ctx, hipSetDevice(ctx->device));
    RSIK_HIP(ctx, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    RSIK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RSIK_OK;
}
int rsik_memcpy_d2h(rsik_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes) {
    if (!ctx) return RSIK_E_INVALID;
    RSIK_HIP(ctx, hipSetDevice(ctx->device));
    RSIK_HIP(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    RSIK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RSIK_OK;
}

static int check_arms(rsik_ctx* ctx, const uint8_t* arm, int arm_uniform, const char* who) {
    if (arm) {
        if (!ctx->have_arm[0] || !ctx->have_arm[1])
            return fail(ctx, RSIK_E_NOT_SET, std::string(who) + ": per-pose arm ids need both arms' constants (rsik_set_arm)");
    } else {
        if (arm_uniform != RSIK_ARM_R && arm_uniform != RSIK_ARM_L)
            return fail(ctx, RSIK_E_INVALID, std::string(who) + ": arm_uniform must be 0 (r) or 1 (l)");
        if (!ctx->have_arm[arm_uniform])
            return fail(ctx, RSIK_E_NOT_SET, std::string(who) + ": constants of the requested arm were not uploaded");
    }
    return RSIK_OK;
}

int rsik_solve(rsik_ctx* ctx, int64_t n, const double* const pose_soa[6], const uint8_t* arm, int arm_uniform,
               int theta_policy, const double* theta_in, const double* previous_joints_host, double* joints,
               double* interval, double* elbow, uint8_t* reachable, uint8_t* state) {
    if (!ctx) return RSIK_E_INVALID;
    if (n < 0) return fail(ctx, RSIK_E_INVALID, "rsik_solve: n < 0");
    if (theta_policy < RSIK_THETA_INTERVAL0 || theta_policy > RSIK_THETA_NONE)
        return fail(ctx, RSIK_E_INVALID, "rsik_solve: unknown theta_policy");
    if ((theta_policy == RSIK_THETA_EXPLICIT || theta_policy == RSIK_THETA_FRACTION) && !theta_in && n > 0)
        return fail(ctx, RSIK_E_INVALID, "rsik_solve: theta_in is required for this theta_policy");
    int rc = check_arms(ctx, arm, arm_uniform, "rsik_solve");
    if (rc != RSIK_OK) return rc;
    if (n == 0) return RSIK_OK;
    if (!pose_soa) return fail(ctx, RSIK_E_INVALID, "rsik_solve: pose_soa is NULL");
    rsik::SolveArgs K;
    K.n = n;
    for (int k = 0; k < 6; k++) {
        if (!pose_soa[k]) return fail(ctx, RSIK_E_INVALID, "rsik_solve: a pose_soa column is NULL");
        K.in[k] = pose_soa[k];
    }
    K.arm = arm;
    K.theta_policy = theta_policy;
    K.theta_in = theta_in;
    for (int k = 0; k < 7; k++) K.prev[k] = previous_joints_host ? previous_joints_host[k] : 0.0;
    K.joints = joints; K.interval = interval; K.elbow = elbow; K.reachable = reachable; K.state = state;
    if (arm) { K.arms[0] = ctx->arms[0]; K.arms[1] = ctx->arms[1]; }
    else { K.arms[0] = ctx->arms[arm_uniform]; K.arms[1] = ctx->arms[arm_uniform]; }
    RSIK_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t tile = (int64_t)rsik::kBlock;
    const int64_t blocks = (n + tile - 1) / tile;
    if (blocks > 0x7fffffffLL) return fail(ctx, RSIK_E_INVALID, "rsik_solve: n too large for one launch");
    dim3 grid((unsigned)blocks), block(rsik::kBlock);
    // tip offset along the goal z axis only (the default arm / the URDF): the specialised goal stage applies
    const bool tipz = K.arms[0].v[RSIK_C_TIPL] == 0.0 && K.arms[0].v[RSIK_C_TIPL + 1] == 0.0 &&
                      K.arms[1].v[RSIK_C_TIPL] == 0.0 && K.arms[1].v[RSIK_C_TIPL + 1] == 0.0 && !ctx->options[RSIK_OPT_NO_TIPZ];
    // mixed launch: do the two blocks agree in everything that has no handedness (arm_const_is_sided)?
    bool mirror = arm != nullptr && !ctx->options[RSIK_OPT_NO_MIRROR];
    for (int i = 0; mirror && i < RSIK_ARM_CONSTS_COUNT; i++)
        if (!rsik::arm_const_is_sided(i) && std::memcmp(&K.arms[0].v[i], &K.arms[1].v[i], sizeof(double)) != 0) mirror = false;
    if (arm) {
        if (mirror) {
            if (tipz) hipLaunchKernelGGL((rsik::solve_kernel<2, true>), grid, block, 0, ctx->stream, K);
            else hipLaunchKernelGGL((rsik::solve_kernel<2, false>), grid, block, 0, ctx->stream, K);
        } else {
            if (tipz) hipLaunchKernelGGL((rsik::solve_kernel<1, true>), grid, block, 0, ctx->stream, K);
            else hipLaunchKernelGGL((rsik::solve_kernel<1, false>), grid, block, 0, ctx->stream, K);
        }
    } else {
        if (tipz) hipLaunchKernelGGL((rsik::solve_kernel<0, true>), grid, block, 0, ctx->stream, K);
        else hipLaunchKernelGGL((rsik::solve_kernel<0, false>), grid, block, 0, ctx->stream, K);
    }
    RSIK_HIP(ctx, hipGetLastError());
    return RSIK_OK;
}

// Python float modulo (sign of the divisor), used for the l-arm limit wrap (C:243-250).
static double host_pymod(double a, double b) {
    double m = std::fmod(a, b);
    if (m != 0.0) {
        if ((b < 0) != (m < 0)) m += b;
    } else {
        m = std::copysign(0.0, b);
    }
    return m;
}

// C:225-252: interval_limit per constrained mode, mirrored and re-wrapped for the left arm.
static void control_limits(int arm, int constrained_mode, double preferred_theta, double lim[2], double* pref) {
    const double pi = rsik::kPi;
    if (constrained_mode == RSIK_MODE_UNCONSTRAINED) { lim[0] = 3 * pi / 4; lim[1] = -2 * pi / 6; }
    else { lim[0] = -4 * pi / 5; lim[1] = 0; }
    if (arm == RSIK_ARM_L) {
        double a = -pi - lim[1], b = -pi - lim[0];
        lim[0] = a; lim[1] = b;
        if (lim[0] < -pi) lim[0] = host_pymod(lim[0], 2 * pi);
        if (lim[1] < -pi) lim[1] = host_pymod(lim[1], 2 * pi);
        if (lim[0] > pi) lim[0] = host_pymod(lim[0], -2 * pi);
        if (lim[1] > pi) lim[1] = host_pymod(lim[1], -2 * pi);
        preferred_theta = -pi - preferred_theta;
    }
    *pref = preferred_theta;
}

// The theta phase's specialised step (continuous_next_theta_lean) replaces limit_theta_to_interval's choice of the nearer
// interval end — |angle_diff(theta, l1)| < |angle_diff(theta, l0)|, U:105-111 — by one comparison with a threshold.  Here
// that threshold is found with the reference's own arithmetic (Python's float `%`), by bisection over the doubles of the
// gap, and the equivalence is then checked on a sample of the gap and on the doubles around the threshold; an interval
// for which it does not hold (or a rate limit the step's range analysis does not cover) keeps the generic step.
static double host_angle_diff(double a, double b) { return host_pymod((a - b) + rsik::kPi, 2 * rsik::kPi) - rsik::kPi; }
static int theta_snap_plan(double l0, double l1, double d_theta_max, double* tdag) {
    const double pi = rsik::kPi;
    *tdag = 0.0;
    if (!(d_theta_max >= 0.0 && d_theta_max < 3.0)) return rsik::kSnapGeneric;
    if (!(std::fabs(l0) <= pi && std::fabs(l1) <= pi)) return rsik::kSnapGeneric;
    if (l0 == l1 || (std::fabs(l0) == pi && std::fabs(l1) == pi)) return rsik::kSnapGeneric;  // the whole circle (U:468-474)
    auto nearer_is_l1 = [&](double t) { return std::fabs(host_angle_diff(t, l1)) < std::fabs(host_angle_diff(t, l0)); };
    const bool wrap = !(l0 < l1);
    // the stretch of the gap that starts at l1: up to l0 (wrap) or up to pi (the rest, (-pi, l0), must answer l0)
    double lo = l1, hi = wrap ? l0 : pi;
    if (!(lo < hi)) return rsik::kSnapGeneric;
    if (!nearer_is_l1(std::nextafter(lo, hi)) || nearer_is_l1(hi)) return rsik::kSnapGeneric;
    lo = std::nextafter(lo, hi);
    while (std::nextafter(lo, hi) < hi) {
        const double mid = lo + (hi - lo) / 2;
        if (nearer_is_l1(mid)) lo = mid; else hi = mid;
    }
    const double t = hi;  // the smallest double of the stretch for which l1 is not the nearer end
    auto agrees = [&](double x) {
        const bool valid = wrap ? (l0 <= x || x <= l1) : (l0 <= x && x <= l1);
        if (valid || !(x > -pi && x <= pi)) return true;
        const bool want = nearer_is_l1(x);
        const bool got = wrap ? (x < t) : (x >= l0 && x < t);  // (below l0 the specialised step answers l0)
        return want == got;
    };
    double x = t;
    for (int k = 0; k < 64; k++) { x = std::nextafter(x, -4.0); if (!agrees(x)) return rsik::kSnapGeneric; }
    x = t;
    for (int k = 0; k < 64; k++) { if (!agrees(x)) return rsik::kSnapGeneric; x = std::nextafter(x, 4.0); }
    const int samples = 4096;
    for (int k = 0; k <= samples; k++) {
        if (!agrees(-pi + (2 * pi) * k / samples)) return rsik::kSnapGeneric;
        if (!agrees(std::nextafter(l1, 4.0) + (t - l1) * k / samples)) return rsik::kSnapGeneric;
    }
    *tdag = t;
    return wrap ? rsik::kSnapWrap : rsik::kSnapInner;
}

static int launch_dims(rsik_ctx* ctx, int64_t n, dim3* grid, const char* who, int threads = rsik::kBlock) {
    const int64_t blocks = (n + threads - 1) / threads;
    if (blocks > 0x7fffffffLL) return fail(ctx, RSIK_E_INVALID, std::string(who) + ": n too large for one launch");
    *grid = dim3((unsigned)blocks);
    return RSIK_OK;
}

// Can the singularity-plane half of is_elbow_ok (utils.py:459-464) fail at all?  The elbow lies on the sphere of
// radius u around the shoulder, so e_z - c e_x <= s_z - c s_x + u sqrt(1 + c^2); when that bound stays below the
// plane's right-hand side (the non-DVT offset -1.01: by a metre) the test is compiled out of the launch.
static bool singularity_plane_binds(const rsik::ArmC (&arms)[2]) {
    for (int slot = 0; slot < 2; slot++) {
        const double* c = arms[slot].v;
        const double sc = c[RSIK_C_SING_COEFF];
        const double rhs = c[RSIK_C_ES + 2] - c[RSIK_C_SING_OFFSET] - sc * c[RSIK_C_ES];
        const double reach_max = c[RSIK_C_SHOULDER + 2] - sc * c[RSIK_C_SHOULDER] + c[RSIK_C_UPPER_ARM] * std::sqrt(1.0 + sc * sc);
        if (!(rhs > reach_max + 1e-6)) return true;
    }
    return false;
}

int rsik_control_discrete(rsik_ctx* ctx, int64_t n, const double* const m12_soa[12], const uint8_t* arm,
                          int arm_uniform, int nb_search_points, double preferred_theta, int constrained_mode,
                          const double* previous_sol_host, const double* current_joints, double orbita3d_max_angle,
                          double* joints, uint8_t* reachable, uint8_t* state, uint8_t* emergency) {
    if (!ctx) return RSIK_E_INVALID;
    if (n < 0) return fail(ctx, RSIK_E_INVALID, "rsik_control_discrete: n < 0");
    if (nb_search_points < 2) return fail(ctx, RSIK_E_INVALID, "rsik_control_discrete: nb_search_points must be >= 2");
    if (constrained_mode != RSIK_MODE_UNCONSTRAINED && constrained_mode != RSIK_MODE_LOW_ELBOW)
        return fail(ctx, RSIK_E_INVALID, "rsik_control_discrete: unknown constrained_mode");
    if (!previous_sol_host) return fail(ctx, RSIK_E_INVALID, "rsik_control_discrete: previous_sol_host is NULL");
    int rc = check_arms(ctx, arm, arm_uniform, "rsik_control_discrete");
    if (rc != RSIK_OK) return rc;
    if (n == 0) return RSIK_OK;
    if (!m12_soa || !joints) return fail(ctx, RSIK_E_INVALID, "rsik_control_discrete: m12_soa / joints is NULL");
    rsik::DiscreteArgs K;
    K.n = n;
    for (int k = 0; k < 12; k++) {
        if (!m12_soa[k]) return fail(ctx, RSIK_E_INVALID, "rsik_control_discrete: an m12_soa column is NULL");
        K.in[k] = m12_soa[k];
    }
    K.arm = arm;
    K.nb = nb_search_points;
    int lg = 0;
    while ((1 << lg) < nb_search_points && lg < 6) lg++;
    K.log2p = lg;
    K.sweep_mode = ctx->options[RSIK_OPT_SWEEP_MODE];  // 0 unless a test / A-B run forces one of the two strategies
    K.euler_roundtrip = ctx->options[RSIK_OPT_EULER_ROUNDTRIP];
    // one round = every workgroup resident at once: 4 workgroups per compute unit (their LDS slabs)
    K.stagger = n <= (int64_t)ctx->compute_units * 4 * rsik::kDiscBlock ? 1 : 0;
    for (int slot = 0; slot < 2; slot++) {
        const int a = arm ? slot : arm_uniform;
        control_limits(a, constrained_mode, preferred_theta, K.lim[slot], &K.pref[slot]);
        K.pref_cs[slot] = std::cos(K.pref[slot]);  // np.cos / np.sin of the reference (U:359-360), once per launch
        K.pref_sn[slot] = std::sin(K.pref[slot]);
        for (int k = 0; k < 7; k++) K.prev_sol[slot][k] = previous_sol_host[7 * a + k];
        for (int k = 0; k < 3; k++) {
            K.prev_cs[slot][k] = std::cos(K.prev_sol[slot][4 + k]);
            K.prev_sn[slot][k] = std::sin(K.prev_sol[slot][4 + k]);
        }
        K.arms[slot] = ctx->arms[a];
    }
    const bool plane_binds = singularity_plane_binds(K.arms);
    K.current_joints = current_joints;
    K.max_angle = orbita3d_max_angle;
    K.cos_max = std::cos(orbita3d_max_angle);
    K.sin_max = std::sin(orbita3d_max_angle);
    K.joints = joints; K.reachable = reachable; K.state = state; K.emergency = emergency;
    RSIK_HIP(ctx, hipSetDevice(ctx->device));
    dim3 grid, block(rsik::kDiscBlock);
    rc = launch_dims(ctx, n, &grid, "rsik_control_discrete", rsik::kDiscBlock);
    if (rc != RSIK_OK) return rc;
    if (arm) {
        if (plane_binds) hipLaunchKernelGGL((rsik::control_discrete_kernel<true, true>), grid, block, 0, ctx->stream, K);
        else hipLaunchKernelGGL((rsik::control_discrete_kernel<true, false>), grid, block, 0, ctx->stream, K);
    } else {
        if (plane_binds) hipLaunchKernelGGL((rsik::control_discrete_kernel<false, true>), grid, block, 0, ctx->stream, K);
        else hipLaunchKernelGGL((rsik::control_discrete_kernel<false, false>), grid, block, 0, ctx->stream, K);
    }
    RSIK_HIP(ctx, hipGetLastError());
    return RSIK_OK;
}

// Arguments shared by the continuous-mode launches (validated once).
static int fill_continuous(rsik_ctx* ctx, const char* who, rsik::ContinuousArgs& K, int64_t n, const double* const m12_soa[12],
                           const double* const current_pose_m12_soa[12], const uint8_t* arm, int arm_uniform,
                           const uint8_t* timed_out, int first_timed_out, double preferred_theta,
                           const double* preferred_theta_self_host, int constrained_mode, double d_theta_max,
                           const double* current_joints, double orbita3d_max_angle, double* cont_state, double* joints,
                           uint8_t* reachable, uint8_t* state) {
    if (constrained_mode != RSIK_MODE_UNCONSTRAINED && constrained_mode != RSIK_MODE_LOW_ELBOW)
        return fail(ctx, RSIK_E_INVALID, std::string(who) + ": unknown constrained_mode");
    if (!preferred_theta_self_host) return fail(ctx, RSIK_E_INVALID, std::string(who) + ": preferred_theta_self_host is NULL");
    int rc = check_arms(ctx, arm, arm_uniform, who);
    if (rc != RSIK_OK) return rc;
    if (!m12_soa || !cont_state || !joints) return fail(ctx, RSIK_E_INVALID, std::string(who) + ": m12_soa / cont_state / joints is NULL");
    std::memset(&K, 0, sizeof K);
    K.n = n;
    K.first_timed_out = first_timed_out;
    for (int k = 0; k < 12; k++) {
        if (!m12_soa[k]) return fail(ctx, RSIK_E_INVALID, std::string(who) + ": an m12_soa column is NULL");
        K.in[k] = m12_soa[k];
        K.cur_pose[k] = current_pose_m12_soa ? current_pose_m12_soa[k] : nullptr;
        if (current_pose_m12_soa && !current_pose_m12_soa[k])
            return fail(ctx, RSIK_E_INVALID, std::string(who) + ": a current_pose column is NULL");
    }
    K.arm = arm;
    K.timed_out = timed_out;
    K.euler_roundtrip = ctx->options[RSIK_OPT_EULER_ROUNDTRIP];
    for (int slot = 0; slot < 2; slot++) {
        const int a = arm ? slot : arm_uniform;
        control_limits(a, constrained_mode, preferred_theta, K.lim[slot], &K.pref_arg[slot]);
        K.pref_self[slot] = preferred_theta_self_host[a];
        K.pref_self_cs[slot] = std::cos(K.pref_self[slot]);  // np.cos / np.sin of the reference (U:359-360)
        K.pref_self_sn[slot] = std::sin(K.pref_self[slot]);
        K.arms[slot] = ctx->arms[a];
    }
    K.d_theta_max = d_theta_max;
    K.current_joints = current_joints;
    K.max_angle = orbita3d_max_angle;
    K.cos_max = std::cos(orbita3d_max_angle);
    K.sin_max = std::sin(orbita3d_max_angle);
    K.st = cont_state; K.joints = joints; K.reachable = reachable; K.state = state;
    return RSIK_OK;
}

int rsik_control_continuous_step(rsik_ctx* ctx, int64_t n, const double* const m12_soa[12],
                                 const double* const current_pose_m12_soa[12], const uint8_t* arm, int arm_uniform,
                                 const uint8_t* timed_out, double preferred_theta, const double* preferred_theta_self_host,
                                 int constrained_mode, double d_theta_max, const double* current_joints,
                                 double orbita3d_max_angle, double* cont_state, double* joints, uint8_t* reachable,
                                 uint8_t* state) {
    if (!ctx) return RSIK_E_INVALID;
    if (n < 0) return fail(ctx, RSIK_E_INVALID, "rsik_control_continuous_step: n < 0");
    if (n == 0) {
        int rc0 = check_arms(ctx, arm, arm_uniform, "rsik_control_continuous_step");
        return rc0;
    }
    rsik::ContinuousArgs K;
    int rc = fill_continuous(ctx, "rsik_control_continuous_step", K, n, m12_soa, current_pose_m12_soa, arm, arm_uniform, timed_out,
                             0, preferred_theta, preferred_theta_self_host, constrained_mode, d_theta_max, current_joints,
                             orbita3d_max_angle, cont_state, joints, reachable, state);
    if (rc != RSIK_OK) return rc;
    RSIK_HIP(ctx, hipSetDevice(ctx->device));
    dim3 grid, block(rsik::kBlock);
    rc = launch_dims(ctx, n, &grid, "rsik_control_continuous_step");
    if (rc != RSIK_OK) return rc;
    {
        const bool pb = singularity_plane_binds(K.arms);
        if (arm) { if (pb) hipLaunchKernelGGL((rsik::control_continuous_kernel<true, true>), grid, block, 0, ctx->stream, K); else hipLaunchKernelGGL((rsik::control_continuous_kernel<true, false>), grid, block, 0, ctx->stream, K); }
        else { if (pb) hipLaunchKernelGGL((rsik::control_continuous_kernel<false, true>), grid, block, 0, ctx->stream, K); else hipLaunchKernelGGL((rsik::control_continuous_kernel<false, false>), grid, block, 0, ctx->stream, K); }
    }
    RSIK_HIP(ctx, hipGetLastError());
    return RSIK_OK;
}

// How rsik_control_continuous_run cuts a run of n trajectories x n_steps steps into blocks, and what it needs for that.
struct ContPlan {
    int64_t T;                           // steps per block (the last one may be shorter)
    std::vector<int64_t> block_t0, block_T;
    size_t per_step, chunks_per_block, slot_bytes, carry_bytes, need;
    int slots;
    size_t n_events;
};
constexpr int kContSlots = 8;
static_assert(kContSlots == sizeof(rsik_ctx::slot_use) / sizeof(rsik_ctx::slot_use[0]), "rsik_ctx::slot_use holds one entry per workspace slot");  // workspace slots in flight (block b + 8 reuses the slot of block b once its chain phase has finished)
// `capturing`: the call is being recorded into a hipGraph.  A replay executes the dependency DAG with 15-40 us per edge
// whatever the streams were, so fewer, longer blocks pay there (4096 x 1000 steps replayed: 0.379 ms with two blocks,
// 0.383 with three, 0.395 with four); launched eagerly four blocks are best (0.43 against 0.46 with two: more overlap for
// the same host-side issue cost).  Round 5, after the value-word edges and the theta-first hold: three blocks are level with or 1-2 %
// ahead of four in every sweep (blocks of 256 / 352 steps: 0.370 / 0.363, 0.378 / 0.374, 0.373 / 0.367 ms on three boxes).
// `all_slots`: the workspace holds kContSlots slots whatever the number of blocks (RSIK_OPT_CONT_GOALS_RESIDENT: the next run's
// blocks take the slots this run's do not).
static int cont_plan(rsik_ctx* ctx, const char* who, int64_t n, int64_t n_steps, bool capturing, ContPlan& P, bool all_slots = false) {
    // (the sequential phases address a block's arrays through 2 GB buffer windows: rows of n * 56 bytes, blocks of <= 384 MB
    // of workspace, i.e. <= 1.3 GB of joints; every block costs the host four launches, so blocks are as long as that allows)
    if (n > (int64_t)30 << 20) return fail(ctx, RSIK_E_INVALID, std::string(who) + ": more than 30 Mi trajectories in one call");
    P.per_step = (size_t)n * (2 * sizeof(double) + 1);
    int64_t T_max = (int64_t)((size_t)384 << 20) / (int64_t)P.per_step;
    if (T_max < 1) T_max = 1;
    if (T_max > 65535) T_max = 65535;  // gridDim.y
    // block size: a third of the run (a quarter until round 5), half of it under capture (the phases of neighbouring blocks overlap: more blocks, shorter fill and drain;
    // fewer blocks, fewer of the ~12 us hand-overs between dependent launches: 4096 x 1000 steps take 0.49 / 0.48 / 0.46 /
    // 0.48 / 0.50 ms with blocks of 128 / 192 / 256 / 512 / 1000 steps), a multiple of the theta batch and of the joint
    // chunk; RSIK_OPT_CONT_BLOCK_STEPS overrides
    const int64_t parts = capturing ? 2 : 3;
    int64_t T = ctx->options[RSIK_OPT_CONT_BLOCK_STEPS] > 0 ? ctx->options[RSIK_OPT_CONT_BLOCK_STEPS] : (n_steps + parts - 1) / parts;
    if (T < 64 && ctx->options[RSIK_OPT_CONT_BLOCK_STEPS] == 0) T = 64;
    // (round 6, runs of 2 000 ... 16 000 steps launch by launch: blocks of 512 steps 0.340-0.377 ms per 1000 steps where a third of
    // the run took 0.362-0.450 and blocks of 256 / 352 0.36-0.41 — profiles/r06/config5_long_runs.txt)
    if (!capturing && ctx->options[RSIK_OPT_CONT_BLOCK_STEPS] == 0 && T > 512) T = 512;
    T = (T + rsik::kSeqBatch - 1) / rsik::kSeqBatch * rsik::kSeqBatch;
    if (T > T_max) T = T_max >= rsik::kSeqBatch ? T_max / rsik::kSeqBatch * rsik::kSeqBatch : T_max;
    if (T > n_steps) T = n_steps;
    P.T = T;
    P.block_t0.clear(); P.block_T.clear();
    for (int64_t t0 = 0; t0 < n_steps; t0 += T) {
        P.block_t0.push_back(t0);
        P.block_T.push_back(n_steps - t0 < T ? n_steps - t0 : T);
    }
    const int64_t n_blocks = (int64_t)P.block_t0.size();
    P.chunks_per_block = ((size_t)T + rsik::kJointChunk - 1) / rsik::kJointChunk;
    P.slot_bytes = (((size_t)T * P.per_step + P.chunks_per_block * (size_t)n + 255) / 256) * 256 + (((size_t)n * sizeof(unsigned) + 255) / 256) * 256;  // (+ the slot's turn hints)
    P.slots = (n_blocks < kContSlots && !all_slots) ? (int)n_blocks : kContSlots;
    P.carry_bytes = (((size_t)n * 2 * sizeof(double) + 255) / 256) * 256 + (((size_t)n * sizeof(unsigned) + 255) / 256) * 256;  // theta_carry, turn_hint
    P.need = P.slot_bytes * P.slots + P.carry_bytes;
    P.n_events = 4 + 6 * (size_t)n_blocks;  // per run 4, per block: prepared, theta, joints, chain, "theta / chain has started" (words only)
    return RSIK_OK;
}
// Workspace, side streams and events for a plan.  Nothing here may happen while the caller's stream is capturing (device
// allocation, stream and event creation are not capturable): a capture needs rsik_control_continuous_reserve, or an
// earlier run of at least this size, first.  An outgrown workspace is retired, not freed: a hipGraph captured earlier
// still points into it.
static int cont_resources(rsik_ctx* ctx, const char* who, size_t need, bool want_streams, size_t n_events) {
    const bool grow = ctx->ws_bytes < need, streams = want_streams && !ctx->have_side, events = ctx->events.size() < n_events;
    if (!grow && !streams && !events) return RSIK_OK;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(ctx->stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
        return fail(ctx, RSIK_E_INVALID, std::string(who) + ": the stream is capturing and this run needs a larger workspace / its streams / "
                    "more events than the context holds: call rsik_control_continuous_reserve(ctx, n, n_steps) before the capture");
    if (grow) {
        // geometric growth (a sweep over rising sizes reallocates a logarithmic number of times)
        size_t want = need;
        if (ctx->ws_bytes > 0 && want < ctx->ws_bytes + ctx->ws_bytes / 2) want = ctx->ws_bytes + ctx->ws_bytes / 2;
        void* fresh = nullptr;
        if (hipMalloc(&fresh, want) != hipSuccess) {
            (void)hipGetLastError();
            want = need;
            RSIK_HIP(ctx, hipMalloc(&fresh, want));
        }
        if (ctx->ws) {
            if (ctx->ws_captured) {
                // a hipGraph recorded from this context still points into the old workspace: kept until rsik_destroy or
                // rsik_control_continuous_release
                ctx->retired_ws.push_back(ctx->ws);
            } else {
                // nothing but runs already issued can use it: freed once they are known to have finished (rsik_sync, _release,
                // rsik_destroy) — not here: a device-wide wait and a free inside an asynchronous call would stall every stream of
                // the process and invalidate a capture some other thread has open
                ctx->outgrown_ws.push_back(ctx->ws);
            }
        }
        ctx->ws = fresh;
        ctx->ws_bytes = want;
        ctx->ws_captured = false;
    }
    if (streams) {
        for (auto& st : ctx->side) RSIK_HIP(ctx, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        ctx->have_side = true;
    }
    while (ctx->events.size() < n_events) {
        hipEvent_t e;
        // (hipEventReleaseToDevice / hipEventDisableSystemFence measured: 0.443 / 0.428 against 0.429-0.439 ms per pass — the
        // ~12 us between dependent launches on different streams are not the cache write-back of the event's release)
        RSIK_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->events.push_back(e);
    }
    return RSIK_OK;
}

#ifdef RSIK_PIPE_TIMING
// diagnostic builds: the phase kernels' first-start / last-end stamps of the PREVIOUS run are printed (RSIK_PIPE_TIMING_PRINT).  Two
// stamp areas take turns, and a run clears the area of the run AFTER it: with RSIK_OPT_CONT_GOALS_RESIDENT a run's prepare kernels
// can execute before the caller's stream has reached that run's start.
static unsigned long long* pipe_timing_begin(rsik_ctx* ctx, int64_t n_blocks) {
    static unsigned long long* pipe_t = nullptr;  // [2 areas][2][5 * 64]: min stamps, then max stamps
    static int64_t pipe_prev_blocks = 0, pipe_run = 0;
    auto clear = [&](unsigned long long* area, hipStream_t st) {
        (void)hipMemsetAsync(area, 0xff, 320 * sizeof(unsigned long long), st);
        (void)hipMemsetAsync(area + 320, 0, 320 * sizeof(unsigned long long), st);
    };
    if (!pipe_t) {
        if (hipMalloc(&pipe_t, 2 * 640 * sizeof(unsigned long long)) != hipSuccess) pipe_t = nullptr;
        if (pipe_t) { clear(pipe_t, ctx->stream); clear(pipe_t + 640, ctx->stream); (void)hipStreamSynchronize(ctx->stream); }
    }
    if (!pipe_t) return nullptr;
    unsigned long long* const mine = pipe_t + 640 * (pipe_run & 1), * const other = pipe_t + 640 * ((pipe_run + 1) & 1);
    if (getenv("RSIK_PIPE_TIMING_PRINT") && pipe_prev_blocks > 0) {  // (that run has been synchronised by now)
        unsigned long long h[640];
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, other, sizeof h, hipMemcpyDeviceToHost);
        unsigned long long base = ~0ull;
        for (int k = 0; k < 320; k++) if (h[k] < base) base = h[k];
        static const char* names[5] = {"prepare", "theta", "joints", "chain", "turns"};
        for (int64_t b = 0; b < pipe_prev_blocks && b < 64; b++)
            for (int ph = 0; ph < 5; ph++)
                if (h[b * 5 + ph] != ~0ull)
                    fprintf(stderr, "[pipe] %-8s(%lld) %8.2f -> %8.2f us\n", names[ph], (long long)b, (h[b * 5 + ph] - base) / 100.0, (h[320 + b * 5 + ph] - base) / 100.0);
    }
    clear(other, ctx->stream);  // (for the run after this one)
    pipe_prev_blocks = n_blocks;
    pipe_run += 1;
    return mine;
}
#endif

int rsik_control_continuous_reserve(rsik_ctx* ctx, int64_t n, int64_t n_steps) {
    const char* who = "rsik_control_continuous_reserve";
    if (!ctx) return RSIK_E_INVALID;
    if (n < 0 || n_steps < 0) return fail(ctx, RSIK_E_INVALID, std::string(who) + ": negative size");
    if (n == 0 || n_steps == 0) return RSIK_OK;
    RSIK_HIP(ctx, hipSetDevice(ctx->device));
    // what an eager run and what a captured run of this size need (their block sizes differ): the larger of each
    ContPlan P, Pc;
    int rc = cont_plan(ctx, who, n, n_steps, false, P, ctx->options[RSIK_OPT_CONT_GOALS_RESIDENT] != 0);
    if (rc != RSIK_OK) return rc;
    if ((rc = cont_plan(ctx, who, n, n_steps, true, Pc)) != RSIK_OK) return rc;
    if (Pc.need > P.need) P.need = Pc.need;
    if (Pc.n_events > P.n_events) P.n_events = Pc.n_events;
    return cont_resources(ctx, who, P.need, true, P.n_events);
}

int rsik_control_continuous_release(rsik_ctx* ctx) {
    if (!ctx) return RSIK_E_INVALID;
    RSIK_HIP(ctx, hipSetDevice(ctx->device));
    RSIK_HIP(ctx, hipDeviceSynchronize());
    for (void* w : ctx->retired_ws) (void)hipFree(w);
    ctx->retired_ws.clear();
    for (void* w : ctx->outgrown_ws) (void)hipFree(w);
    ctx->outgrown_ws.clear();
    if (ctx->ws) (void)hipFree(ctx->ws);
    ctx->ws = nullptr;
    ctx->ws_bytes = 0;
    ctx->ws_captured = false;
    ctx->last_run.valid = false;
    for (auto& u : ctx->slot_use) u = {0, 0};
    return RSIK_OK;
}

// Two runs of one context share its workspace, words and side streams: a run issued on another stream than the one before it
// waits for that one's end (runs on one stream are ordered by the stream; hipGraphs recorded from one context must not be
// replayed concurrently: include/rsik.h).
static int cont_run_begin(rsik_ctx* ctx, bool capturing) {
    // Workspaces and word arrays that earlier runs outgrew: a caller that synchronises through its own framework never calls
    // rsik_sync, so they are also let go here, without waiting — when the last run issued is known to have finished (every run
    // before it has, then: runs of one context are ordered).  Never inside a capture.
    if (!capturing && !ctx->outgrown_ws.empty() && ctx->have_run_done && hipEventQuery(ctx->run_done) == hipSuccess) {
        for (void* w : ctx->outgrown_ws) (void)hipFree(w);
        ctx->outgrown_ws.clear();
    }
    (void)hipGetLastError();  // (hipErrorNotReady is not an error)
    if (capturing || !ctx->have_run_done || ctx->run_stream == ctx->stream) return RSIK_OK;
    RSIK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->run_done, 0));
    return RSIK_OK;
}
// `where`: the stream whose last operation marks the run's end — the chain stream behind the last chain kernel (every phase of every
// block is ahead of it), so that the record is not one more operation between this run's end and the next run's first kernel on the
// caller's stream (round 6: that stretch is on the critical path of runs that overlap); the caller's stream where a run failed part-way.
static int cont_run_end(rsik_ctx* ctx, bool capturing, hipStream_t where) {
    if (capturing) return RSIK_OK;
    if (!ctx->have_run_done) {
        RSIK_HIP(ctx, hipEventCreateWithFlags(&ctx->run_done, hipEventDisableTiming));
        ctx->have_run_done = true;
    }
    RSIK_HIP(ctx, hipEventRecord(ctx->run_done, where));
    ctx->run_stream = ctx->stream;
    return RSIK_OK;
}

// The whole trajectory batch: the phased pipeline (include/rsik.h), or — RSIK_CONT_RUN_STEPS — one launch of the step
// kernel per control step.
int rsik_control_continuous_run(rsik_ctx* ctx, int64_t n, int64_t n_steps, const double* m12_steps,
                                const double* const current_pose_m12_soa[12], const uint8_t* arm, int arm_uniform,
                                int first_step_timed_out, double preferred_theta, const double* preferred_theta_self_host,
                                int constrained_mode, double d_theta_max, const double* current_joints,
                                double orbita3d_max_angle, double* cont_state, double* joints_steps,
                                uint8_t* reachable_steps, uint8_t* state_steps) {
    const char* who = "rsik_control_continuous_run";
    if (!ctx) return RSIK_E_INVALID;
    if (n < 0 || n_steps < 0) return fail(ctx, RSIK_E_INVALID, "rsik_control_continuous_run: negative size");
    if (n == 0 || n_steps == 0) return check_arms(ctx, arm, arm_uniform, who);
    if (!m12_steps || !joints_steps) return fail(ctx, RSIK_E_INVALID, "rsik_control_continuous_run: NULL buffer");
    const double* cols[12];
    for (int c = 0; c < 12; c++) cols[c] = m12_steps + (size_t)c * (size_t)n;  // step 0; step s is 12 n doubles further
    rsik::ContinuousArgs K0;
    int rc = fill_continuous(ctx, who, K0, n, cols, current_pose_m12_soa, arm, arm_uniform, nullptr, first_step_timed_out ? 1 : 0,
                             preferred_theta, preferred_theta_self_host, constrained_mode, d_theta_max, current_joints,
                             orbita3d_max_angle, cont_state, joints_steps, reachable_steps, state_steps);
    if (rc != RSIK_OK) return rc;
    RSIK_HIP(ctx, hipSetDevice(ctx->device));
    dim3 grid, block(rsik::kBlock);
    rc = launch_dims(ctx, n, &grid, who);
    if (rc != RSIK_OK) return rc;
    // is_reachable_no_limits can only fail (C:385-387) for a projection margin that lets the pulled-back wrist land beyond
    // u + f (S:343-345); the pipeline's phases do not carry that outcome, the step kernel does.
    bool no_limits_can_fail = false;
    for (int slot = 0; slot < 2; slot++) no_limits_can_fail = no_limits_can_fail || !(K0.arms[slot].v[RSIK_C_PROJ_MARGIN] > 1e-12);
    if (ctx->options[RSIK_OPT_CONT_RUN_MODE] == RSIK_CONT_RUN_STEPS || no_limits_can_fail) {
        // (what the run was issued as is the caller's to know: rsik_control_continuous_last_form — a solver whose projection margin
        // lets is_reachable_no_limits fail gets n_steps launches whatever RSIK_OPT_CONT_RUN_MODE says)
        ctx->last_run_form = ctx->options[RSIK_OPT_CONT_RUN_MODE] == RSIK_CONT_RUN_STEPS ? RSIK_CONT_FORM_STEPS : RSIK_CONT_FORM_STEPS_NO_LIMITS_CAN_FAIL;
        ctx->last_run.valid = false;  // (this run's outputs are written on the caller's stream: the next phased run forks behind them)
        for (int64_t k = 0; k < n_steps; k++) {
            rsik::ContinuousArgs K = K0;
            for (int c = 0; c < 12; c++) K.in[c] = m12_steps + ((size_t)k * 12 + c) * (size_t)n;
            if (k > 0) {
                K.first_timed_out = 0;
                K.current_joints = nullptr;
                for (int c = 0; c < 12; c++) K.cur_pose[c] = nullptr;
            }
            K.joints = joints_steps + (size_t)k * n * 7;
            K.reachable = reachable_steps ? reachable_steps + (size_t)k * n : nullptr;
            K.state = state_steps ? state_steps + (size_t)k * n : nullptr;
            {
                const bool pb = singularity_plane_binds(K.arms);
                if (arm) { if (pb) hipLaunchKernelGGL((rsik::control_continuous_kernel<true, true>), grid, block, 0, ctx->stream, K); else hipLaunchKernelGGL((rsik::control_continuous_kernel<true, false>), grid, block, 0, ctx->stream, K); }
                else { if (pb) hipLaunchKernelGGL((rsik::control_continuous_kernel<false, true>), grid, block, 0, ctx->stream, K); else hipLaunchKernelGGL((rsik::control_continuous_kernel<false, false>), grid, block, 0, ctx->stream, K); }
            }
        }
        RSIK_HIP(ctx, hipGetLastError());
        return RSIK_OK;
    }
    bool capturing = false;
    {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        capturing = hipStreamIsCapturing(ctx->stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
    }
    if ((rc = cont_run_begin(ctx, capturing)) != RSIK_OK) return rc;
    // ---- phased pipeline.  The four phases of a block run on four streams (theta on the caller's, the others on the
    // context's own), ordered by events: prepare(b) -> theta(b) -> joints(b) -> chain(b), theta(b) after theta(b-1),
    // chain(b) after chain(b-1).  The two sequential phases (a lone wave per SIMD on a few CUs) then run beside each other
    // and beside the chip-filling ones of the neighbouring blocks.  Exactly four streams: the runtime multiplexes streams
    // onto four hardware queues, and a fifth stream shares a queue with another one — measured with theta on a stream of
    // its own: theta(b + 1) queued up behind chain(b)'s wait for joints(b), 0.85 -> 1.28 ms per 1000-step pass.  (Giving
    // the sequential phases compute units of their own with hipExtStreamCreateWithCUMask was measured too: every kernel
    // got slower, 2.4 ms per pass.)
    // A run is cut into blocks of steps; up to eight workspace slots are in flight (block b + 8 reuses the slot of block b
    // once its last phase has finished).
    // RSIK_OPT_CONT_GOALS_RESIDENT (rsik.h): the prepare phase of this run need not wait for the previous run's end
    const bool resident = !capturing && ctx->options[RSIK_OPT_CONT_GOALS_RESIDENT] != 0;
    ContPlan P;
    if ((rc = cont_plan(ctx, who, n, n_steps, capturing, P, resident)) != RSIK_OK) return rc;
    {
        // the context holds what BOTH forms of a run of this size need, so that a run that was first issued eagerly can be
        // captured afterwards (and the other way round) without creating anything
        ContPlan other, both = P;
        if ((rc = cont_plan(ctx, who, n, n_steps, !capturing, other, capturing && ctx->options[RSIK_OPT_CONT_GOALS_RESIDENT] != 0)) != RSIK_OK) return rc;
        if (other.need > both.need) both.need = other.need;
        if (other.n_events > both.n_events) both.n_events = other.n_events;
        if ((rc = cont_resources(ctx, who, both.need, true, both.n_events)) != RSIK_OK) return rc;
    }
    if (capturing) ctx->ws_captured = true;
    const std::vector<int64_t>&block_t0 = P.block_t0, &block_T = P.block_T;
    const int64_t n_blocks = (int64_t)block_t0.size();
    const size_t slot_bytes = P.slot_bytes, carry_bytes = P.carry_bytes, chunks_per_block = P.chunks_per_block;
    const int slots = P.slots;
    (void)carry_bytes;
#ifdef RSIK_PIPE_TIMING
    unsigned long long* const pipe_t = pipe_timing_begin(ctx, n_blocks);
#endif
    hipStream_t s_main = ctx->stream, s_theta = ctx->stream, s_prep = ctx->side[0], s_joints = ctx->side[1], s_chain = ctx->side[2];
    // Dependencies between the streams.  Recorded into a hipGraph they are events (the only form a capture takes).  Issued
    // launch by launch they are words in device memory: the producer's stream writes this run's sequence number behind its
    // kernel (hipStreamWriteValue32), the consumer's stream waits for the word to reach it (hipStreamWaitValue32) — measured
    // on an otherwise idle chip (scripts/probes/edge_probe.hip): the dependent kernel starts 3.8 us after its parent's end,
    // against 10.6 us behind an event (15-55 us inside a pass).  Words are per (kind, block) and only ever grow.
    const int variant = ctx->options[RSIK_OPT_CONT_PHASED_VARIANT];
    const bool by_value = !capturing && ctx->can_wait_value != 0 && !(variant & RSIK_PHASED_EDGES_BY_EVENT);
    if (by_value) {
        const size_t need_words = P.n_events;
        if (ctx->edge_count < need_words) {
            ctx->last_run.valid = false;  // (its words are not these: this run forks behind it)
            // (the old words: runs already issued still wait on them and write them — freed like an outgrown workspace)
            if (ctx->edge_words) { ctx->outgrown_ws.push_back(ctx->edge_words); ctx->edge_words = nullptr; ctx->edge_count = 0; }
            RSIK_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->edge_words), need_words * 2 * sizeof(unsigned)));
            RSIK_HIP(ctx, hipMemset(ctx->edge_words, 0, need_words * 2 * sizeof(unsigned)));
            ctx->edge_count = need_words * 2;
            ctx->edge_seq = 0;
        }
        if (ctx->edge_seq >= RSIK_EDGE_SEQ_WRAP) {
            // A word only ever grows and every wait is "word >= a run's number": before the 32-bit number wraps (4e9 runs: weeks of a
            // control loop that issues a run per tick) everything issued drains, the words start over from zero and this run forks
            // behind the caller's stream like a first one.
            RSIK_HIP(ctx, hipStreamSynchronize(ctx->stream));
            for (auto& st : ctx->side) RSIK_HIP(ctx, hipStreamSynchronize(st));
            // (word 3 is not a sequence number: a theta kernel's "gave up waiting" mark, rsik_sync's to read and clear)
            RSIK_HIP(ctx, hipMemset(ctx->edge_words, 0, 3 * sizeof(unsigned)));
            RSIK_HIP(ctx, hipMemset(ctx->edge_words + 4, 0, (ctx->edge_count - 4) * sizeof(unsigned)));
            ctx->edge_seq = 0;
            ctx->last_run.valid = false;
        }
        ctx->edge_seq += 1;
    }
    const unsigned seq = ctx->edge_seq;
    // (a word's meaning does not depend on the run's number of blocks: a word is only ever written from one stream, in issue order,
    // so its value never goes back — runs that overlap, below, rely on it)
    // kinds: 0 prepared, 1 theta, 2 joints, 3 chain, 4 the theta kernel has started, 5 the chain kernel has started; ids 0-3 are the
    // run's: 0 the start-up kernel is done, 1 the prepare stream may fork, 2 the start-up kernel has started
    auto edge_id = [&](int kind, int64_t b) { return 4 + 6 * (size_t)b + kind; };
    constexpr size_t kInitStartedId = 2, kTimeoutId = 3;
    // Does this run's prepare phase start without waiting for the previous run's end?  Only behind a run of the same shape issued the
    // same way on the same stream into the same workspace and words (anything else: the streams meet first, as always).
    rsik_ctx::LastRun& L = ctx->last_run;
    const bool overlap = resident && by_value && L.valid && L.stream == ctx->stream && L.ws == ctx->ws && L.words == ctx->edge_words &&
                         L.slot_bytes == P.slot_bytes && L.slots == slots && !ctx->ws_captured;
    const uint8_t* const st_lo = state_steps, * const st_hi = state_steps ? state_steps + (size_t)n_steps * (size_t)n : nullptr;
    const uint8_t* const rc_lo = reachable_steps, * const rc_hi = reachable_steps ? reachable_steps + (size_t)n_steps * (size_t)n : nullptr;
    bool alias_any = false, alias_same = false;
    if (overlap) {
        auto meet = [](const uint8_t* a0, const uint8_t* a1, const uint8_t* b0, const uint8_t* b1) { return a0 && b0 && a0 < b1 && b0 < a1; };
        alias_any = meet(st_lo, st_hi, L.state_lo, L.state_hi) || meet(st_lo, st_hi, L.reach_lo, L.reach_hi) ||
                    meet(rc_lo, rc_hi, L.state_lo, L.state_hi) || meet(rc_lo, rc_hi, L.reach_lo, L.reach_hi);
        alias_same = alias_any && st_lo == L.state_lo && rc_lo == L.reach_lo && n == L.n && n_steps == L.n_steps && P.T == L.T;
    } else {
        for (auto& u : ctx->slot_use) u = {0, 0};  // the streams meet at this run's start: every slot is free
    }
    const int slot_base = overlap ? ctx->slot_next : 0;
    // A run that overlaps the one before it: its theta kernels wait for their prepare kernels themselves (cont_theta_kernel), and the
    // joints kernel of a block takes "theta of the NEXT block has started" for "theta of this block is done" (same stream: it is) —
    // so that nothing stands between two theta kernels on the caller's stream.  Only there: K overlapping 1000-step passes 0.334-0.342
    // against 0.341-0.346 ms with stream waits, same box; a run on its own is level (0.357-0.371 / 0.363-0.378), a long one — 8 000 /
    // 16 000 steps in blocks of 512 — slower, 0.348 / 0.425 against 0.340 / 0.377 ms per 1000 steps (profiles/r06/config5_long_runs.txt).
    // (variant bit 16, timing experiments and the A/B tests: stream waits and writes there too)
    const bool theta_waits = by_value && overlap && !(variant & 16);
    const unsigned last_seq = L.seq;
    const int64_t last_blocks = L.n_blocks;
    auto signal = [&](hipStream_t st, size_t id) -> hipError_t {
        if (by_value) return hipStreamWriteValue32(st, ctx->edge_words + id, seq, 0);
        return hipEventRecord(ctx->events[id], st);
    };
    auto wait_for = [&](hipStream_t st, size_t id) -> hipError_t {
        if (by_value) return hipStreamWaitValue32(st, ctx->edge_words + id, seq, hipStreamWaitValueGte, 0xffffffffu);
        return hipStreamWaitEvent(st, ctx->events[id], 0);
    };
    rsik::ContRunArgs R;
    std::memset(&R, 0, sizeof R);
    R.n = n;
    R.m12_steps = m12_steps;
    R.arm = arm;
    R.euler_roundtrip = K0.euler_roundtrip;
    for (int slot = 0; slot < 2; slot++) {
        R.pref_arg[slot] = K0.pref_arg[slot]; R.pref_self[slot] = K0.pref_self[slot];
        R.pref_self_cs[slot] = K0.pref_self_cs[slot]; R.pref_self_sn[slot] = K0.pref_self_sn[slot];
        R.lim[slot][0] = K0.lim[slot][0]; R.lim[slot][1] = K0.lim[slot][1];
        R.arms[slot] = K0.arms[slot];
    }
    R.d_theta_max = d_theta_max;
    R.max_angle = K0.max_angle; R.cos_max = K0.cos_max; R.sin_max = K0.sin_max;
    R.st = cont_state; R.joints = joints_steps; R.reachable = reachable_steps; R.state = state_steps;
    R.theta_carry = reinterpret_cast<double*>(static_cast<char*>(ctx->ws) + slot_bytes * slots);
    R.no_turn_hint = (variant & 64) ? 1 : 0;
    R.run_turn_hint = reinterpret_cast<unsigned*>(static_cast<char*>(ctx->ws) + slot_bytes * slots + (((size_t)n * 2 * sizeof(double) + 255) / 256) * 256);
    const dim3 grid8((unsigned)((n * 8 + rsik::kChainBlock - 1) / rsik::kChainBlock));  // (n <= 30 Mi: fits)
    // What a pass really looks like was measured with in-kernel stamps (a -DRSIK_PIPE_TIMING build,
    // scripts/probes/c5_untraced_timeline.py; the profiler's kernel trace delays launches and shows another schedule): a
    // dependency between launches on DIFFERENT streams costs the dependent kernel 15-40 us after its last parent has
    // finished, launch by launch and in a graph replay alike (theta(b) -> joints(b): 31-41 us in a replay), a kernel
    // behind its predecessor on the SAME stream 4-7 us.  Keeping the whole critical chain on one in-order stream (init,
    // theta(b), joints(b) alternately, prepares beside it) removes those hand-overs but also the overlap of theta(b + 1)
    // with joints(b): 0.42 ms launch by launch (the best eager figure) but 0.40-0.43 replayed, against 0.39 for the
    // overlapped form below, which stays.  A block that reuses a workspace slot can only be issued once the block that
    // frees it has been (its event must have been recorded).
    const bool plane_binds = singularity_plane_binds(R.arms);
    // the theta phase's step, specialised for the control interval where that is proven equivalent (single-arm launches)
    int snap_kind = rsik::kSnapGeneric;
    if (!arm) snap_kind = theta_snap_plan(R.lim[0][0], R.lim[0][1], d_theta_max, &R.snap_tdag);
#ifdef RSIK_PIPE_TIMING
    R.tmin = pipe_t; R.tmax = pipe_t ? pipe_t + 320 : nullptr;
#endif
    auto set_block = [&](int64_t b) {
#ifdef RSIK_PIPE_TIMING
        R.tslot = (int)(b < 64 ? b : 63);
#endif
        R.t0 = block_t0[b];
        R.T = block_T[b];
        R.first_block = b == 0;
        R.last_block = b == n_blocks - 1;
        R.ws = reinterpret_cast<double*>(static_cast<char*>(ctx->ws) + slot_bytes * (size_t)((slot_base + b) % slots));
        R.gw = R.ws + (size_t)R.T * (size_t)n;
        R.flags = reinterpret_cast<uint8_t*>(R.gw + (size_t)R.T * (size_t)n);
        R.chunk_event = R.flags + (size_t)R.T * (size_t)n;
        // (the slot's turn hints sit at its end, whatever the block's length; a block that is the first to use its slot in this run
        // reads the run's own)
        R.slot_turn_hint = reinterpret_cast<uint8_t*>(R.ws) + slot_bytes - (((size_t)n * sizeof(unsigned) + 255) / 256) * 256;
        R.turn_hint = b < slots ? R.run_turn_hint : reinterpret_cast<unsigned*>(R.slot_turn_hint);
    };
    const int64_t head = n_blocks < slots ? n_blocks : slots;  // blocks with a workspace slot of their own: issued phase by phase
    auto issue_prepare = [&](int64_t b) -> int {
        set_block(b);
        const dim3 grid2(grid.x, (unsigned)R.T);
        // the slot's previous block is done (launch by launch: whichever run it belonged to)
        if (by_value) {
            const rsik_ctx::SlotUse u = ctx->slot_use[(slot_base + b) % slots];
            if (u.seq != 0) RSIK_HIP(ctx, hipStreamWaitValue32(s_prep, ctx->edge_words + u.word, u.seq, hipStreamWaitValueGte, 0xffffffffu));
            ctx->slot_use[(slot_base + b) % slots] = {edge_id(3, b), seq};
            // a run that overlaps the one before it and writes the same reachable / state rows: behind that run's chain kernel of
            // the same rows (the same cut), or of its last block
            // (and not before that run's last joints kernel has finished: started earlier, this run's prepare kernels share the chip
            // with that run's joints kernels, which its end — and with it this run's start-up — waits for: K passes took 0.39-0.41 ms
            // each instead of 0.36-0.39; behind it they fill the chip while that run's last chain kernel and this run's start-up
            // search, lone waves both, have it to themselves)
            // (variant bits 4 / 8, timing experiments: no such wait / the last chain kernel's END)
            if (overlap && b == 0 && !(variant & 4)) RSIK_HIP(ctx, hipStreamWaitValue32(s_prep, ctx->edge_words + edge_id((variant & 8) ? 3 : 5, last_blocks - 1), last_seq, hipStreamWaitValueGte, 0xffffffffu));
            // ... and the later ones leave the chip to the lone waves ahead of them on the critical path — the start-up search, then
            // theta(0), theta(1) ...: prepare(1) is held until the start-up kernel has started, prepare(b) until theta(b - 2) has (each
            // issued before this wait, issue_all's order for a run that overlaps)
            if (overlap && b == 1 && !(variant & 4)) RSIK_HIP(ctx, hipStreamWaitValue32(s_prep, ctx->edge_words + kInitStartedId, seq, hipStreamWaitValueGte, 0xffffffffu));
            if (overlap && b >= 2 && !(variant & 4)) RSIK_HIP(ctx, hipStreamWaitValue32(s_prep, ctx->edge_words + edge_id(4, b - 2), seq, hipStreamWaitValueGte, 0xffffffffu));
            if (alias_same) RSIK_HIP(ctx, hipStreamWaitValue32(s_prep, ctx->edge_words + edge_id(3, b), last_seq, hipStreamWaitValueGte, 0xffffffffu));
            else if (alias_any && b == 0) RSIK_HIP(ctx, hipStreamWaitValue32(s_prep, ctx->edge_words + edge_id(3, last_blocks - 1), last_seq, hipStreamWaitValueGte, 0xffffffffu));
        } else if (b >= slots) {
            RSIK_HIP(ctx, wait_for(s_prep, edge_id(3, b - slots)));
        }
        if (arm) { if (plane_binds) hipLaunchKernelGGL((rsik::cont_prepare_kernel<true, true>), grid2, block, 0, s_prep, R); else hipLaunchKernelGGL((rsik::cont_prepare_kernel<true, false>), grid2, block, 0, s_prep, R); }
        else { if (plane_binds) hipLaunchKernelGGL((rsik::cont_prepare_kernel<false, true>), grid2, block, 0, s_prep, R); else hipLaunchKernelGGL((rsik::cont_prepare_kernel<false, false>), grid2, block, 0, s_prep, R); }
        RSIK_HIP(ctx, signal(s_prep, edge_id(0, b)));
        return RSIK_OK;
    };
    auto issue_theta = [&](int64_t b) -> int {
        set_block(b);
        // (launch by launch: the kernel says when it has started — the joints kernel of the block before is held until then)
        R.started_word = by_value ? ctx->edge_words + edge_id(4, b) : nullptr;
        R.started_seq = seq;
        if (theta_waits) {
            R.wait_word = ctx->edge_words + edge_id(0, b);
            R.wait_seq = seq;
            R.timeout_word = ctx->edge_words + kTimeoutId;
        } else {
            R.wait_word = nullptr;
            RSIK_HIP(ctx, wait_for(s_theta, edge_id(0, b)));
        }
        const dim3 grid_t((unsigned)((n + rsik::kThetaBlock - 1) / rsik::kThetaBlock)), block_t(rsik::kThetaBlock);
        if (arm) hipLaunchKernelGGL((rsik::cont_theta_kernel<true, rsik::kSnapGeneric>), grid_t, block_t, 0, s_theta, R);
        else if (snap_kind == rsik::kSnapInner) hipLaunchKernelGGL((rsik::cont_theta_kernel<false, rsik::kSnapInner>), grid_t, block_t, 0, s_theta, R);
        else if (snap_kind == rsik::kSnapWrap) hipLaunchKernelGGL((rsik::cont_theta_kernel<false, rsik::kSnapWrap>), grid_t, block_t, 0, s_theta, R);
        else hipLaunchKernelGGL((rsik::cont_theta_kernel<false, rsik::kSnapGeneric>), grid_t, block_t, 0, s_theta, R);
        if (!theta_waits || b == n_blocks - 1) RSIK_HIP(ctx, signal(s_theta, edge_id(1, b)));
        return RSIK_OK;
    };
    auto issue_back = [&](int64_t b) -> int {  // joints(b), chain(b)
        set_block(b);
        // (a wave = 8 trajectories x 8 steps: n / 8 groups, 4 per workgroup)
        const dim3 grid2((unsigned)((n + 8 * (rsik::kBlock / 64) - 1) / (8 * (rsik::kBlock / 64))), (unsigned)((R.T + rsik::kJointChunk - 1) / rsik::kJointChunk));
        // joints(b) needs theta(b).  Launch by launch it is held a little longer: until theta(b + 1) has STARTED (which is after
        // theta(b)'s end: same stream).  The theta kernel's lone waves want 276 registers each — a SIMD that holds six waves of a
        // chip-filling kernel has none to give — so a theta kernel that becomes ready together with a joints kernel and loses the
        // race for the chip only gets in when that kernel drains: theta(b + 1) ran behind joints(b), not beside it (measured with
        // in-kernel stamps: a third of a pass).  Let in first, it has its SIMDs before the chip fills up.
        // (measured and not kept — with the theta phase as one persistent launch, docs/experiments.md A.4: the first joints kernel held until the last prepare kernel has
        // completed, so that the prepare kernels — which every later phase of a block waits for — have the chip to themselves:
        // 0.424 against 0.371 ms per pass, the joints kernels then run one behind the other with a stream operation's ~15 us
        // between them; higher stream priority for the prepare and chain streams: no difference)
        // (theta(b + 1) has been ISSUED before this wait — issue_all's order: streams can share a hardware queue, and a wait that
        // sat in one ahead of the launch it waits for would wait for ever)
        if (!theta_waits || b == n_blocks - 1) RSIK_HIP(ctx, wait_for(s_joints, edge_id(1, b)));
        else RSIK_HIP(ctx, hipStreamWaitValue32(s_joints, ctx->edge_words + edge_id(4, b + 1), seq, hipStreamWaitValueGte, 0xffffffffu));
        if (by_value && !theta_waits && !(variant & RSIK_PHASED_NO_THETA_FIRST) && b + 1 < n_blocks)
            RSIK_HIP(ctx, hipStreamWaitValue32(s_joints, ctx->edge_words + edge_id(4, b + 1), seq, hipStreamWaitValueGte, 0xffffffffu));
        if (arm) hipLaunchKernelGGL(rsik::cont_joints_kernel<true>, grid2, block, 0, s_joints, R);
        else hipLaunchKernelGGL(rsik::cont_joints_kernel<false>, grid2, block, 0, s_joints, R);
        RSIK_HIP(ctx, signal(s_joints, edge_id(2, b)));
        RSIK_HIP(ctx, wait_for(s_chain, edge_id(2, b)));
        R.chain_started_word = by_value ? ctx->edge_words + edge_id(5, b) : nullptr;
        R.started_seq = seq;
        if (arm) hipLaunchKernelGGL(rsik::cont_chain_kernel<true>, grid8, dim3(rsik::kChainBlock), 0, s_chain, R);
        else hipLaunchKernelGGL(rsik::cont_chain_kernel<false>, grid8, dim3(rsik::kChainBlock), 0, s_chain, R);
        RSIK_HIP(ctx, signal(s_chain, edge_id(3, b)));
        return RSIK_OK;
    };
    // (Re)initialisation of the trajectories that start here (C:296-325: the start-up search for previous_theta, ~55 us
    // of lone waves), then the pipeline's streams join in.  The prepare phase depends on the goal matrices alone, not on
    // the trajectory state: its stream forks off BEFORE the initialisation (behind whatever the caller queued ahead of
    // this call), so prepare(0) runs beside it and theta(0) starts when both are done; the joints and chain streams fork
    // behind it.
    // RSIK_OPT_CONT_GOALS_RESIDENT, behind a run of the same shape: the prepare stream does not fork at all — it carries on behind
    // the previous run's prepare kernels, so this run's run beside that run's joints and chain kernels (its slots and output rows
    // are waited for one by one, issue_prepare); the start-up kernel and everything behind it wait for the previous run's end
    // as they must (the trajectory state).
    auto issue_start = [&]() -> int {
        if (!overlap) {
            RSIK_HIP(ctx, signal(s_main, 1));
            RSIK_HIP(ctx, wait_for(s_prep, 1));
        }
        {
            // two lanes per trajectory where get_joints cannot move the solver's state (no elbow projection possible)
            const bool pair = !singularity_plane_binds(K0.arms);
            dim3 grid_init = grid;
            int rc_ = RSIK_OK;
            if (pair && (rc_ = launch_dims(ctx, n * 2, &grid_init, who)) != RSIK_OK) return rc_;
            K0.started_word = by_value ? ctx->edge_words + kInitStartedId : nullptr;
            K0.started_seq = seq;
            if (arm) { if (pair) hipLaunchKernelGGL((rsik::cont_init_kernel<true, true>), grid_init, block, 0, s_main, K0); else hipLaunchKernelGGL((rsik::cont_init_kernel<true, false>), grid_init, block, 0, s_main, K0); }
            else { if (pair) hipLaunchKernelGGL((rsik::cont_init_kernel<false, true>), grid_init, block, 0, s_main, K0); else hipLaunchKernelGGL((rsik::cont_init_kernel<false, false>), grid_init, block, 0, s_main, K0); }
            // (the joints and chain streams' first kernels wait for theta(0), which is behind this kernel on its stream)
            if (!theta_waits) RSIK_HIP(ctx, signal(s_main, 0));
        }
        if (!theta_waits) {
            RSIK_HIP(ctx, wait_for(s_joints, 0));
            RSIK_HIP(ctx, wait_for(s_chain, 0));
        }
        return RSIK_OK;
    };
    // Issue order of the blocks that have a workspace slot of their own (it is also the order of the nodes in a captured
    // graph): prepare(0), theta(0), then the other prepares back to back, the other thetas, then joints + chain of
    // every block.  Measured on graph replays of 4096 x 1000 steps against three other orders (prepare / theta
    // alternating: 0.394-0.411 ms; all prepares, all thetas: 0.398-0.401; thetas and backs alternating: 0.414-0.416):
    // 0.387-0.392 ms.
    auto issue_all = [&]() -> int {
        int rc_ = issue_start();
        if (rc_ != RSIK_OK) return rc_;
        if ((rc_ = issue_prepare(0)) != RSIK_OK) return rc_;
        if ((rc_ = issue_theta(0)) != RSIK_OK) return rc_;
        if (overlap) {
            // (a run that overlaps the one before it holds prepare(b) until theta(b - 2) has started: that one is issued first)
            for (int64_t b = 1; b < head; b++) {
                if ((rc_ = issue_prepare(b)) != RSIK_OK) return rc_;
                if ((rc_ = issue_theta(b)) != RSIK_OK) return rc_;
            }
        } else {
            for (int64_t b = 1; b < head; b++)
                if ((rc_ = issue_prepare(b)) != RSIK_OK) return rc_;
            for (int64_t b = 1; b < head; b++)
                if ((rc_ = issue_theta(b)) != RSIK_OK) return rc_;
        }
        // joints + chain of every block; a block beyond the head (it reuses a workspace slot: its prepare kernel waits for the
        // chain kernel of the block `slots` before it, issued by then) has its prepare and theta kernels issued just ahead of
        // the joints kernel of the block BEFORE it, so that that one can be held until the theta kernel has started, like the
        // head's (round 6: a 16 384-step run in blocks of 512 had its theta kernels start 60-100 us late, behind whichever
        // chip-filling kernel was draining, profiles/r06/config5_long_runs.txt)
        for (int64_t b = 0; b < n_blocks; b++) {
            if (b + 1 >= head && b + 1 < n_blocks) {
                if ((rc_ = issue_prepare(b + 1)) != RSIK_OK) return rc_;
                if ((rc_ = issue_theta(b + 1)) != RSIK_OK) return rc_;
            }
            if ((rc_ = issue_back(b)) != RSIK_OK) return rc_;
        }
        // the caller's stream continues once the last chain (hence every phase of every block) is done
        RSIK_HIP(ctx, wait_for(s_main, edge_id(3, n_blocks - 1)));
        RSIK_HIP(ctx, hipGetLastError());
        return RSIK_OK;
    };
    if ((rc = issue_all()) != RSIK_OK) {
        // A failure part-way leaves value waits queued on the context's streams whose words nobody is going to write (an event
        // that was never recorded is no wait at all; a word is one).  Every word of the context is raised to this run's number
        // from a stream of its own, so that the streams drain and no later call (rsik_sync, _release, rsik_destroy) hangs on
        // them; the run's outputs are unspecified, the error is the caller's to see.
        if (by_value) {
            const std::string first_error = ctx->err;
            hipStream_t fresh = nullptr;
            if (hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking) == hipSuccess) {
                (void)hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ctx->edge_words), (int)seq, ctx->edge_count, fresh);
                (void)hipStreamSynchronize(fresh);
                (void)hipStreamDestroy(fresh);
            }
            (void)hipGetLastError();
            ctx->err = first_error;
        }
        ctx->last_run.valid = false;
        for (auto& u : ctx->slot_use) u = {0, 0};
        // (what was issued before the failure is still running: rsik_sync and the next run's housekeeping wait for THIS point)
        if (!capturing) {
            const std::string first_error = ctx->err;
            (void)cont_run_end(ctx, false, ctx->stream);
            (void)hipGetLastError();
            ctx->err = first_error;
        }
        return rc;
    }
    if (capturing) {
        ctx->last_run_form = RSIK_CONT_FORM_PHASED_CAPTURED;
    } else {
        ctx->last_run_form = overlap ? RSIK_CONT_FORM_PHASED_OVERLAPPED : RSIK_CONT_FORM_PHASED;
        L.valid = by_value;
        L.seq = seq; L.stream = ctx->stream; L.ws = ctx->ws; L.words = ctx->edge_words;
        L.n = n; L.n_steps = n_steps; L.T = P.T; L.n_blocks = n_blocks; L.slot_bytes = P.slot_bytes; L.slots = slots;
        L.state_lo = st_lo; L.state_hi = st_hi; L.reach_lo = rc_lo; L.reach_hi = rc_hi;
        ctx->slot_next = (int)((slot_base + n_blocks) % slots);
    }
    return cont_run_end(ctx, capturing, s_chain);
}

int rsik_control_continuous_last_form(const rsik_ctx* ctx) { return ctx ? ctx->last_run_form : RSIK_CONT_FORM_NONE; }

int rsik_matrix_to_pose(rsik_ctx* ctx, int64_t n, const double* const m12_soa[12], int identity_shortcut,
                        double* const pose_soa[6]) {
    if (!ctx) return RSIK_E_INVALID;
    if (n < 0) return fail(ctx, RSIK_E_INVALID, "rsik_matrix_to_pose: n < 0");
    if (n == 0) return RSIK_OK;
    if (!m12_soa || !pose_soa) return fail(ctx, RSIK_E_INVALID, "rsik_matrix_to_pose: NULL column table");
    rsik::MatrixToPoseArgs K;
    K.n = n;
    K.identity_shortcut = identity_shortcut;
    for (int k = 0; k < 12; k++) {
        if (!m12_soa[k]) return fail(ctx, RSIK_E_INVALID, "rsik_matrix_to_pose: an m12_soa column is NULL");
        K.in[k] = m12_soa[k];
    }
    for (int k = 0; k < 6; k++) {
        if (!pose_soa[k]) return fail(ctx, RSIK_E_INVALID, "rsik_matrix_to_pose: a pose_soa column is NULL");
        K.out[k] = pose_soa[k];
    }
    RSIK_HIP(ctx, hipSetDevice(ctx->device));
    dim3 grid, block(rsik::kBlock);
    int rc = launch_dims(ctx, n, &grid, "rsik_matrix_to_pose");
    if (rc != RSIK_OK) return rc;
    hipLaunchKernelGGL(rsik::matrix_to_pose_kernel, grid, block, 0, ctx->stream, K);
    RSIK_HIP(ctx, hipGetLastError());
    return RSIK_OK;
}

static int fill_state_args(rsik_ctx* ctx, rsik::StateArgs* K, int64_t n, const uint8_t* arm, int arm_uniform,
                           const char* who) {
    if (n < 0) return fail(ctx, RSIK_E_INVALID, std::string(who) + ": n < 0");
    int rc = check_arms(ctx, arm, arm_uniform, who);
    if (rc != RSIK_OK) return rc;
    std::memset(K, 0, sizeof *K);
    K->n = n;
    K->arm = arm;
    for (int slot = 0; slot < 2; slot++) K->arms[slot] = ctx->arms[arm ? slot : arm_uniform];
    return RSIK_OK;
}

int rsik_reach_state(rsik_ctx* ctx, int64_t n, const double* const pose_soa[6], const uint8_t* arm, int arm_uniform,
                     int no_limits, double* solver_state, double* interval, uint8_t* reachable, uint8_t* state) {
    if (!ctx) return RSIK_E_INVALID;
    rsik::StateArgs K;
    int rc = fill_state_args(ctx, &K, n, arm, arm_uniform, "rsik_reach_state");
    if (rc != RSIK_OK) return rc;
    if (n == 0) return RSIK_OK;
    if (!pose_soa || !solver_state) return fail(ctx, RSIK_E_INVALID, "rsik_reach_state: pose_soa / solver_state is NULL");
    for (int k = 0; k < 6; k++) {
        if (!pose_soa[k]) return fail(ctx, RSIK_E_INVALID, "rsik_reach_state: a pose_soa column is NULL");
        K.in[k] = pose_soa[k];
    }
    K.no_limits = no_limits ? 1 : 0;
    K.solver_state = solver_state; K.interval = interval; K.reachable = reachable; K.state = state;
    RSIK_HIP(ctx, hipSetDevice(ctx->device));
    dim3 grid, block(rsik::kBlock);
    rc = launch_dims(ctx, n, &grid, "rsik_reach_state");
    if (rc != RSIK_OK) return rc;
    if (arm) hipLaunchKernelGGL(rsik::reach_state_kernel<true>, grid, block, 0, ctx->stream, K);
    else hipLaunchKernelGGL(rsik::reach_state_kernel<false>, grid, block, 0, ctx->stream, K);
    RSIK_HIP(ctx, hipGetLastError());
    return RSIK_OK;
}

int rsik_joints_from_state(rsik_ctx* ctx, int64_t n, double* solver_state, const uint8_t* arm, int arm_uniform,
                           const double* theta, const double* previous_joints, double* joints, double* elbow) {
    if (!ctx) return RSIK_E_INVALID;
    rsik::StateArgs K;
    int rc = fill_state_args(ctx, &K, n, arm, arm_uniform, "rsik_joints_from_state");
    if (rc != RSIK_OK) return rc;
    if (n == 0) return RSIK_OK;
    if (!solver_state || !theta)  // joints may be NULL: the row's slots 24-30 carry them too
        return fail(ctx, RSIK_E_INVALID, "rsik_joints_from_state: solver_state / theta is NULL");
    K.solver_state = solver_state; K.theta = theta; K.prev = previous_joints; K.joints = joints; K.elbow = elbow;
    RSIK_HIP(ctx, hipSetDevice(ctx->device));
    dim3 grid, block(rsik::kBlock);
    rc = launch_dims(ctx, n, &grid, "rsik_joints_from_state");
    if (rc != RSIK_OK) return rc;
    if (arm) hipLaunchKernelGGL(rsik::joints_state_kernel<true>, grid, block, 0, ctx->stream, K);
    else hipLaunchKernelGGL(rsik::joints_state_kernel<false>, grid, block, 0, ctx->stream, K);
    RSIK_HIP(ctx, hipGetLastError());
    return RSIK_OK;
}

int rsik_elbow_from_state(rsik_ctx* ctx, int64_t n, const double* solver_state, const double* theta, double* elbow) {
    if (!ctx) return RSIK_E_INVALID;
    if (n < 0) return fail(ctx, RSIK_E_INVALID, "rsik_elbow_from_state: n < 0");
    if (n == 0) return RSIK_OK;
    if (!solver_state || !theta || !elbow)
        return fail(ctx, RSIK_E_INVALID, "rsik_elbow_from_state: solver_state / theta / elbow is NULL");
    rsik::StateArgs K;
    std::memset(&K, 0, sizeof K);
    K.n = n;
    K.solver_state = const_cast<double*>(solver_state); K.theta = theta; K.elbow = elbow;
    RSIK_HIP(ctx, hipSetDevice(ctx->device));
    dim3 grid, block(rsik::kBlock);
    int rc = launch_dims(ctx, n, &grid, "rsik_elbow_from_state");
    if (rc != RSIK_OK) return rc;
    hipLaunchKernelGGL(rsik::elbow_state_kernel, grid, block, 0, ctx->stream, K);
    RSIK_HIP(ctx, hipGetLastError());
    return RSIK_OK;
}

int rsik_stage(rsik_ctx* ctx, int op, int64_t n, int arm, const double* in, int in_stride, double* out, int out_stride) {
    const char* who = "rsik_stage";
    if (!ctx) return RSIK_E_INVALID;
    // doubles a row takes and gives, by stage (include/rsik.h)
    static const int need_in[RSIK_STAGE_COUNT] = {6, 6, 6, 3, 17, 12, 10, 3, 2, 3, 4, 9, 14, 4, 10, 21, 18},
                     need_out[RSIK_STAGE_COUNT] = {5, 3, 7, 8, 3, 7, 7, 9, 1, 1, 2, 1, 7, 3, 8, 8, 3};
    // (stages 5 on read no arm constant: they do not need an arm to have been set)
    if (op < 0 || op >= RSIK_STAGE_COUNT) return fail(ctx, RSIK_E_INVALID, std::string(who) + ": unknown stage");
    if (n < 0) return fail(ctx, RSIK_E_INVALID, std::string(who) + ": n < 0");
    int rc = RSIK_OK;
    if (op <= RSIK_STAGE_CIRCLES_LINKED) {  // (the stages that read arm constants; the others run on a context no arm was uploaded to)
        if ((rc = check_arms(ctx, nullptr, arm, who)) != RSIK_OK) return rc;
    } else if (arm != RSIK_ARM_R && arm != RSIK_ARM_L) {
        return fail(ctx, RSIK_E_INVALID, std::string(who) + ": arm must be 0 (r) or 1 (l)");
    }
    if (n == 0) return RSIK_OK;
    if (!in || !out) return fail(ctx, RSIK_E_INVALID, std::string(who) + ": NULL buffer");
    if (in_stride < need_in[op] || out_stride < need_out[op])
        return fail(ctx, RSIK_E_INVALID, std::string(who) + ": stage " + std::to_string(op) + " reads " + std::to_string(need_in[op]) + " and writes " +
                    std::to_string(need_out[op]) + " doubles per row");
    rsik::StageArgs K;
    std::memset(&K, 0, sizeof K);
    K.n = n; K.op = op; K.in = in; K.out = out; K.in_stride = in_stride; K.out_stride = out_stride;
    if (ctx->have_arm[arm]) K.arms[0] = K.arms[1] = ctx->arms[arm];  // (else zeros: the utils helpers read none of it)
    RSIK_HIP(ctx, hipSetDevice(ctx->device));
    dim3 grid, block(rsik::kBlock);
    rc = launch_dims(ctx, n, &grid, who);
    if (rc != RSIK_OK) return rc;
    hipLaunchKernelGGL(rsik::stage_kernel, grid, block, 0, ctx->stream, K);
    RSIK_HIP(ctx, hipGetLastError());
    return RSIK_OK;
}

static int launch_fk(rsik_ctx* ctx, rsik::FkArgs& K, int64_t n, const uint8_t* arm, int arm_uniform, const char* who) {
    int rc = check_arms(ctx, arm, arm_uniform, who);
    if (rc != RSIK_OK) return rc;
    K.n = n;
    K.arm = arm;
    for (int slot = 0; slot < 2; slot++) K.arms[slot] = ctx->arms[arm ? slot : arm_uniform];
    RSIK_HIP(ctx, hipSetDevice(ctx->device));
    dim3 grid, block(rsik::kBlock);
    rc = launch_dims(ctx, n, &grid, who);
    if (rc != RSIK_OK) return rc;
    if (arm) hipLaunchKernelGGL(rsik::fk_kernel<true>, grid, block, 0, ctx->stream, K);
    else hipLaunchKernelGGL(rsik::fk_kernel<false>, grid, block, 0, ctx->stream, K);
    RSIK_HIP(ctx, hipGetLastError());
    return RSIK_OK;
}

int rsik_forward_kinematics(rsik_ctx* ctx, int64_t n, const double* joints, const uint8_t* arm, int arm_uniform,
                            double* position, double* rotation) {
    if (!ctx) return RSIK_E_INVALID;
    if (n < 0) return fail(ctx, RSIK_E_INVALID, "rsik_forward_kinematics: n < 0");
    if (n == 0) return RSIK_OK;
    if (!joints || (!position && !rotation))
        return fail(ctx, RSIK_E_INVALID, "rsik_forward_kinematics: joints or both outputs are NULL");
    rsik::FkArgs K;
    std::memset(&K, 0, sizeof K);
    K.joints = joints; K.pos = position; K.rot = rotation;
    return launch_fk(ctx, K, n, arm, arm_uniform, "rsik_forward_kinematics");
}

int rsik_fk_residual(rsik_ctx* ctx, int64_t n, int goal_kind, const double* const* goal_soa, const double* joints,
                     const uint8_t* arm, int arm_uniform, double* err) {
    if (!ctx) return RSIK_E_INVALID;
    if (n < 0) return fail(ctx, RSIK_E_INVALID, "rsik_fk_residual: n < 0");
    if (goal_kind != RSIK_GOAL_POSE6 && goal_kind != RSIK_GOAL_M12)
        return fail(ctx, RSIK_E_INVALID, "rsik_fk_residual: goal_kind must be RSIK_GOAL_POSE6 or RSIK_GOAL_M12");
    if (n == 0) return RSIK_OK;
    if (!goal_soa || !joints || !err) return fail(ctx, RSIK_E_INVALID, "rsik_fk_residual: goal_soa / joints / err is NULL");
    rsik::FkArgs K;
    std::memset(&K, 0, sizeof K);
    const int cols = goal_kind == RSIK_GOAL_M12 ? 12 : 6;
    for (int k = 0; k < cols; k++) {
        if (!goal_soa[k]) return fail(ctx, RSIK_E_INVALID, "rsik_fk_residual: a goal_soa column is NULL");
        K.goal[k] = goal_soa[k];
    }
    K.goal_kind = goal_kind; K.joints = joints; K.err = err;
    return launch_fk(ctx, K, n, arm, arm_uniform, "rsik_fk_residual");
}

int rsik_debug_math(rsik_ctx* ctx, int op, int64_t n, const double* a, const double* b, double* out0, double* out1) {
    if (!ctx) return RSIK_E_INVALID;
    if (n < 0 || op < 0 || op > 8) return fail(ctx, RSIK_E_INVALID, "rsik_debug_math: bad op or n");
    if (n == 0) return RSIK_OK;
    if (!a || !out0 || ((op == 3 || op == 5 || op == 6 || op == 7) && !b)) return fail(ctx, RSIK_E_INVALID, "rsik_debug_math: NULL operand");
    RSIK_HIP(ctx, hipSetDevice(ctx->device));
    dim3 grid, block(rsik::kBlock);
    if (op == 8) {  // clock monitor: n waves, one per 64-thread workgroup so that they spread over the chip
        if (!out1 || n > 4096) return fail(ctx, RSIK_E_INVALID, "rsik_debug_math: op 8 needs out1 and n <= 4096 waves");
        hipLaunchKernelGGL(rsik::clock_monitor_kernel, dim3((unsigned)n), dim3(64), 0, ctx->stream, a, n, out0, out1);
        RSIK_HIP(ctx, hipGetLastError());
        return RSIK_OK;
    }
    int rc = launch_dims(ctx, n, &grid, "rsik_debug_math");
    if (rc != RSIK_OK) return rc;
    hipLaunchKernelGGL(rsik::debug_math_kernel, grid, block, 0, ctx->stream, op, n, a, b, out0, out1);
    RSIK_HIP(ctx, hipGetLastError());
    return RSIK_OK;
}

#include "rsik_comm.hpp"

}  // extern "C"
