/* Sanitizer driver of the CPU oracle's LM path (test infrastructure): builds a synthetic problem in plain C -- a
 * mono problem (TS.cpp:247-282) and a 4-camera ring rig whose frames are seen by two adjacent cameras
 * (multi_calib.cpp:155-218) -- and runs orc_solve on it.  Compiled together with oracle/tscm_oracle.c by
 * tests/test_oracle_sanitizers.py:
 *   gcc   -fsanitize=address,undefined                      sequential path
 *   clang -fsanitize=thread -fopenmp (libomp + archer)      OpenMP path (orc_set_num_threads)
 * usage: oracle_san <threads> ; exit 0 and "clean" if both solves converge to the noise floor. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tscm_oracle.h"

static unsigned long long rng_state = 0x9e3779b97f4a7c15ull;
static double urand(void)
{
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (double)(rng_state >> 11) / 9007199254740992.0;
}
static double nrand(void) { return sqrt(-2.0 * log(urand() + 1e-300)) * cos(6.283185307179586 * urand()); }

/* intrinsics of EpipolarRectify/calib.yaml cam0 */
static const double kIntr[9] = { 431.29641731951233, 430.77528857601646, 646.53015901902177, 521.20451427825685,
                                 -0.27125775332873053, -0.087861849854000834, 0.56023435889162265, 0.0, 0.0 };

static int run(int C, int views_per_cam, int mono)
{
    const int n = 54, B = mono ? views_per_cam : C * views_per_cam / 2, V = mono ? B : 2 * B, N = V * n;
    double *bxy = malloc(sizeof(double) * 2 * n), *u = malloc(sizeof(double) * N), *v = malloc(sizeof(double) * N);
    int *vc = malloc(sizeof(int) * V), *vb = malloc(sizeof(int) * V), *vo = malloc(sizeof(int) * V), *vn = malloc(sizeof(int) * V);
    double *cam = calloc(6 * C, sizeof(double)), *intr = malloc(sizeof(double) * 9 * C), *brd = malloc(sizeof(double) * 6 * B);
    unsigned char *cconst = calloc(C, 1);
    for (int j = 0; j < n; ++j) { bxy[2 * j] = 45.0 * (j % 9) - 180.0; bxy[2 * j + 1] = 45.0 * (j / 9) - 112.5; }
    for (int m = 0; m < C; ++m) {
        memcpy(intr + 9 * m, kIntr, sizeof(kIntr));
        if (!mono && m > 0) { cam[6 * m + 1] = 0.25 * m; cam[6 * m + 3] = 60.0 * m; cam[6 * m + 5] = -20.0 * m; }   /* a gentle arc: every board stays in view */
    }
    cconst[0] = 1;                                               /* multi_calib.cpp:186 */
    int k = 0, view = 0;
    for (int b = 0; b < B; ++b) {
        double *rt = brd + 6 * b;
        rt[0] = 0.3 * nrand(); rt[1] = 0.3 * nrand(); rt[2] = 0.3 * nrand();
        rt[3] = 150.0 * (urand() - 0.5); rt[4] = 100.0 * (urand() - 0.5); rt[5] = 450.0 + 200.0 * urand();
        const int cams[2] = { mono ? 0 : b % C, mono ? 0 : (b + 1) % C };
        for (int q = 0; q < (mono ? 1 : 2); ++q, ++view) {
            const int m = cams[q];
            vc[view] = m; vb[view] = b; vo[view] = k; vn[view] = n;
            for (int j = 0; j < n; ++j, ++k) {
                const double p[3] = { bxy[2 * j], bxy[2 * j + 1], 0.0 };
                double pw[3], pc[3], uv[2];
                orc_angle_axis_rotate_point(rt, p, pw);
                for (int a = 0; a < 3; ++a) pw[a] += rt[3 + a];
                orc_angle_axis_rotate_point(cam + 6 * m, pw, pc);
                for (int a = 0; a < 3; ++a) pc[a] += cam[6 * m + 3 + a];
                orc_project(intr + 9 * m, pc, uv);
                u[k] = uv[0] + 0.1 * nrand(); v[k] = uv[1] + 0.1 * nrand();
            }
        }
    }
    /* initial guess = ground truth perturbed */
    for (int m = 0; m < C; ++m) {
        for (int a = 0; a < 7; ++a) intr[9 * m + a] *= 1.0 + 0.01 * nrand();
        if (!mono && m > 0) for (int a = 0; a < 6; ++a) cam[6 * m + a] += (a < 3 ? 0.005 : 2.0) * nrand();
    }
    for (int b = 0; b < B; ++b) for (int a = 0; a < 6; ++a) brd[6 * b + a] += (a < 3 ? 0.005 : 2.0) * nrand();
    orc_problem P;
    memset(&P, 0, sizeof(P));
    P.n_cameras = C; P.n_boards = B; P.n_points = n; P.n_views = V;
    P.board_xy = bxy; P.view_camera = vc; P.view_board = vb; P.view_offset = vo; P.view_count = vn; P.obs_u = u; P.obs_v = v;
    P.cam_rt = cam; P.intr = intr; P.board_rt = brd; P.cam_pose_constant = mono ? NULL : cconst; P.mono = mono;
    orc_options o;
    orc_default_options(&o, mono);
    orc_summary *s = calloc(1, sizeof(orc_summary));
    const int rc = orc_solve(&P, &o, s);
    const double rmse = orc_rmse(&P);
    printf("%s: rc %d, %d iterations, %s rmse %.4f px\n", mono ? "mono" : "rig", rc, s->num_iterations - 1, s->message, rmse);
    const int ok = rc == 0 && s->termination_type == ORC_CONVERGENCE && rmse < 0.15 && rmse > 0.11;   /* sqrt(2) * 0.1 px of noise, minus the fitted degrees of freedom */
    free(bxy); free(u); free(v); free(vc); free(vb); free(vo); free(vn); free(cam); free(intr); free(brd); free(cconst); free(s);
    return ok;
}

int main(int argc, char **argv)
{
    orc_set_num_threads(argc > 1 ? atoi(argv[1]) : 1);
    const int a = run(1, 20, 1), b = run(4, 12, 0);
    if (a && b) { printf("clean\n"); return 0; }
    return 1;
}
