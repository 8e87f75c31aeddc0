/* A plain-C consumer of libgoldfish_hip.so: no Python, no torch -- what a cgo / JNI / Fortran binding would do.
 * One flat bicubic patch [0,2] x [0,1], 4 x 2 elements, body force (0, 0, -3) per unit area, u = 0:
 *   - the residual of the undeformed plate is minus the consistent load, so sum_a R_(a,z) = +3 * area = 6
 *     (partition of unity of the rational basis) and the in-plane sums vanish;
 *   - K is symmetric: x . (K y) == y . (K x) through gf_apply.
 * Exit code 0 = all checks passed.  Built and run by tests/test_gpu_api.py::test_plain_c_consumer_of_the_c_abi. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "goldfish_hip.h"

#define CHECK(call) do { if ((call) != 0) { fprintf(stderr, "%s failed: %s\n", #call, gf_last_error()); return 2; } } while (0)

int main(void) {
    enum { P = 3, NELU = 4, NELV = 2, NU = NELU + P, NV = NELV + P, NCP = NU * NV };
    double ku[NU + P + 1], kv[NV + P + 1], knots[NU + NV + 2 * P + 2];
    for (int i = 0; i < NU + P + 1; ++i) { int k = i - P; if (k < 0) k = 0; if (k > NELU) k = NELU; ku[i] = (double)k / NELU; }
    for (int i = 0; i < NV + P + 1; ++i) { int k = i - P; if (k < 0) k = 0; if (k > NELV) k = NELV; kv[i] = (double)k / NELV; }
    for (int i = 0; i < NU + P + 1; ++i) knots[i] = ku[i];
    for (int i = 0; i < NV + P + 1; ++i) knots[NU + P + 1 + i] = kv[i];
    int32_t degree[2] = {P, P}, ncp[2] = {NU, NV};
    int64_t knot_off[3] = {0, NU + P + 1, NU + NV + 2 * P + 2}, cp_off[2] = {0, NCP};
    double weights[NCP], young[1] = {2.0e5}, poisson[1] = {0.3}, body[3] = {0.0, 0.0, -3.0};
    for (int a = 0; a < NCP; ++a) weights[a] = 1.0;
    gf_model_desc d = {0};
    d.n_patches = 1; d.degree = degree; d.ncp = ncp; d.knot_off = knot_off; d.knots = knots; d.cp_off = cp_off;
    d.weights = weights; d.young = young; d.poisson = poisson; d.body_force = body;
    gf_handle* h = NULL;
    CHECK(gf_create(&d, 0, &h));
    if (gf_total_cp(h) != NCP || gf_num_dofs(h) != 3 * NCP || gf_num_elements(h) != NELU * NELV) { fprintf(stderr, "sizes\n"); return 3; }
    /* control points at the Greville abscissae: the identity map of [0,2] x [0,1] */
    double cx[NCP], cy[NCP], cz[NCP], th[NCP], u[3 * NCP];
    for (int j = 0; j < NV; ++j) for (int i = 0; i < NU; ++i) {
        double gu = 0, gv = 0;
        for (int k = 1; k <= P; ++k) { gu += ku[i + k]; gv += kv[j + k]; }
        cx[i + j * NU] = 2.0 * gu / P; cy[i + j * NU] = gv / P; cz[i + j * NU] = 0.0; th[i + j * NU] = 0.05;
    }
    for (int k = 0; k < 3 * NCP; ++k) u[k] = 0.0;
    CHECK(gf_set_cp(h, 0, cx, NCP)); CHECK(gf_set_cp(h, 1, cy, NCP)); CHECK(gf_set_cp(h, 2, cz, NCP));
    CHECK(gf_set_thickness(h, th, NCP)); CHECK(gf_set_u(h, u, 3 * NCP));
    CHECK(gf_assemble(h, GF_ASM_R | GF_ASM_K)); CHECK(gf_sync(h));
    double R[3 * NCP], s[3] = {0, 0, 0};
    CHECK(gf_get_residual(h, R, 3 * NCP));
    for (int a = 0; a < NCP; ++a) for (int i = 0; i < 3; ++i) s[i] += R[3 * a + i];
    if (fabs(s[2] - 6.0) > 1e-11 || fabs(s[0]) > 1e-11 || fabs(s[1]) > 1e-11) { fprintf(stderr, "load sums %g %g %g\n", s[0], s[1], s[2]); return 4; }
    double x[3 * NCP], y[3 * NCP], Kx[3 * NCP], Ky[3 * NCP], xKy = 0, yKx = 0;
    for (int k = 0; k < 3 * NCP; ++k) { x[k] = sin(0.7 * k); y[k] = cos(1.3 * k); Kx[k] = 0; Ky[k] = 0; }
    CHECK(gf_apply(h, GF_MAT_K, 0, x, 3 * NCP, Kx, 3 * NCP)); CHECK(gf_apply(h, GF_MAT_K, 0, y, 3 * NCP, Ky, 3 * NCP));
    for (int k = 0; k < 3 * NCP; ++k) { xKy += x[k] * Ky[k]; yKx += y[k] * Kx[k]; }
    if (fabs(xKy - yKx) > 1e-10 * fabs(xKy)) { fprintf(stderr, "K not symmetric: %g %g\n", xKy, yKx); return 5; }
    if (gf_set_u(h, u, 7) == 0) { fprintf(stderr, "bad size accepted\n"); return 6; }        /* errors are return codes + gf_last_error */
    gf_destroy(h);
    printf("c consumer ok: sum R_z = %.12f, x.Ky = %.6e\n", s[2], xKy);
    return 0;
}
