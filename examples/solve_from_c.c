/*
 * solve_from_c.c — the C ABI of include/rsik.h used from plain C (no HIP headers, no Python, no torch):
 * upload one arm's constants and a small pose batch, run the fused is_reachable + get_joints solve, read the results.
 *
 *   gcc -std=c99 -I include examples/solve_from_c.c -L reachy2_symbolic_ik_amd/csrc -lrsik_hip \
 *       -Wl,-rpath,$PWD/reachy2_symbolic_ik_amd/csrc -o solve_from_c
 *   ./solve_from_c consts.bin            # RSIK_ARM_CONSTS_COUNT doubles written by ArmGeometry(...).pack().tofile(...)
 *
 * Prints one line per pose: reachable, state code, interval, seven joints (what tests/test_gpu_parity.py compares with
 * the Python drop-in class).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "rsik.h"

#define CHECK(call)                                                                          \
    do {                                                                                     \
        int rc_ = (call);                                                                    \
        if (rc_ != RSIK_OK) {                                                                \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, rsik_last_error(ctx));      \
            return 1;                                                                        \
        }                                                                                    \
    } while (0)

int main(int argc, char **argv) {
    rsik_ctx *ctx = NULL;
    double consts[RSIK_ARM_CONSTS_COUNT];
    FILE *f;
    if (argc < 2 || !(f = fopen(argv[1], "rb"))) { fprintf(stderr, "usage: solve_from_c consts.bin\n"); return 2; }
    if (fread(consts, sizeof(double), RSIK_ARM_CONSTS_COUNT, f) != RSIK_ARM_CONSTS_COUNT) { fprintf(stderr, "short constants file\n"); return 2; }
    fclose(f);

    /* poses: the reference's README pose, its benchmark pose, a point behind the torso and a far one (README.md:73-75,
     * src/benchmark/ik_benchmarks.py:13-14) — column-major (SoA): px[n], py[n], pz[n], roll[n], pitch[n], yaw[n] */
    enum { N = 4 };
    const double pi = 3.141592653589793;
    double host[6][N] = {
        {0.55, 0.3, -0.3, 1.5},
        {-0.3, -0.1, -0.2, -0.2},
        {-0.15, 0.1, 0.0, 0.0},
        {0.0, 20 * pi / 180, 0.0, 0.0},
        {-pi / 2, -50 * pi / 180, 0.0, 0.0},
        {0.0, 20 * pi / 180, 0.0, 0.0},
    };

    if (rsik_create(0, &ctx) != RSIK_OK) { fprintf(stderr, "rsik_create: %s\n", rsik_last_error(NULL)); return 1; }
    CHECK(rsik_set_arm(ctx, RSIK_ARM_R, consts, RSIK_ARM_CONSTS_COUNT));

    void *d_in, *d_joints, *d_interval, *d_reach, *d_state;
    CHECK(rsik_malloc(ctx, sizeof host, &d_in));
    CHECK(rsik_malloc(ctx, N * 7 * sizeof(double), &d_joints));
    CHECK(rsik_malloc(ctx, N * 2 * sizeof(double), &d_interval));
    CHECK(rsik_malloc(ctx, N, &d_reach));
    CHECK(rsik_malloc(ctx, N, &d_state));
    CHECK(rsik_memcpy_h2d(ctx, d_in, host, sizeof host));
    const double *cols[6];
    for (int k = 0; k < 6; k++) cols[k] = (const double *)d_in + (size_t)k * N;

    CHECK(rsik_solve(ctx, N, cols, NULL, RSIK_ARM_R, RSIK_THETA_INTERVAL0, NULL, NULL, (double *)d_joints,
                     (double *)d_interval, NULL, (uint8_t *)d_reach, (uint8_t *)d_state));
    CHECK(rsik_sync(ctx));

    double joints[N][7], interval[N][2];
    uint8_t reach[N], state[N];
    CHECK(rsik_memcpy_d2h(ctx, joints, d_joints, sizeof joints));
    CHECK(rsik_memcpy_d2h(ctx, interval, d_interval, sizeof interval));
    CHECK(rsik_memcpy_d2h(ctx, reach, d_reach, sizeof reach));
    CHECK(rsik_memcpy_d2h(ctx, state, d_state, sizeof state));
    for (int i = 0; i < N; i++) {
        printf("%d %d %.17g %.17g", reach[i], state[i], interval[i][0], interval[i][1]);
        for (int k = 0; k < 7; k++) printf(" %.17g", joints[i][k]);
        printf("\n");
    }
    rsik_free(ctx, d_in); rsik_free(ctx, d_joints); rsik_free(ctx, d_interval); rsik_free(ctx, d_reach); rsik_free(ctx, d_state);
    rsik_destroy(ctx);
    return 0;
}
