// Host-side entry points into kl_point.hpp (the same header the kernels include), used by
// tests/test_pointwise_forms.py to check the device closed forms on the CPU.
#include "kl_point.hpp"
// the strain evaluation takes the displacement derivatives dz = z - Z (kl_strains); these wrappers keep the (z, Z) signature of the tests and form the difference
// here (the test states are well conditioned); gfh_kl_strains / gfh_pen_rot_measures take the displacement quantities themselves
static void diff15(const double* z, const double* Z, double* dz) { for (int k = 0; k < 15; ++k) dz[k] = z[k] - Z[k]; }
extern "C" {
void gfh_kl_strains(const double* Z, const double* dz, double* eps, double* kap) {
    double z[15], n[3], N[3], j, Jn, Dn[3][6];
    for (int k = 0; k < 15; ++k) z[k] = Z[k] + dz[k];
    gf::normal_derivs(z, z + 3, n, j, Dn); gf::normal_derivs(Z, Z + 3, N, Jn, Dn);
    gf::kl_strains(Z, dz, n, N, j, Jn, eps, kap);
}
void gfh_pen_rot_measures(const double* Y, const double* dY, const double* tau, double* e) { gf::pen_rot_measures(Y, dY, tau, e[0], e[1]); }
void gfh_shell_point(const double* z, const double* Z, double t, double E, double nu, double* im, double* Pzz, double* PzZ, double* Pz, double* Pzt) {
    for (int k = 0; k < gf::IM_SIZE; ++k) im[k] = 0.0;
    double dz[15]; diff15(z, Z, dz);
    gf::shell_point(z, Z, dz, t, E, nu, im);
    for (int r = 0; r < 15; ++r) { Pz[r] = im[gf::IM_PZ + r]; Pzt[r] = gf::pzt_entry(im, r); }
    for (int r = 0; r < 15; ++r) for (int s = 0; s < 15; ++s) { Pzz[15 * r + s] = gf::pzz_entry(im, r, s); PzZ[15 * r + s] = gf::pzZ_entry(im, r, s); }
}
void gfh_shell_point_cols(const double* z, const double* Z, double t, double E, double nu, double* im) {
    for (int k = 0; k < gf::IM_SIZE; ++k) im[k] = 0.0;
    double dz[15]; diff15(z, Z, dz);
    for (int ic = 0; ic < 3; ++ic) {
        const double d[3] = {ic == 0 ? 1.0 : 0.0, ic == 1 ? 1.0 : 0.0, ic == 2 ? 1.0 : 0.0};
        gf::shell_point_cols(z, Z, dz, t, E, nu, ic, d, ic == 0, im);
    }
}
void gfh_penalty_point(const double* y, const double* Y, const double* tau, double ad, double ar, double dt, double* out) {
    const int tan[12] = {3, 4, 5, 6, 7, 8, 12, 13, 14, 15, 16, 17};
    double dY[12];
    for (int k = 0; k < 12; ++k) dY[k] = y[tan[k]] - Y[k];
    gf::penalty_point(y, Y, dY, tau, ad, ar, dt, out);
}
void gfh_shell_energy_point(const double* z, const double* Z, double t, double E, double nu, double* out) { double dz[15]; diff15(z, Z, dz); gf::shell_energy_point(z, Z, dz, t, E, nu, out); }
void gfh_shell_stress_point(const double* z, const double* Z, double t, double E, double nu, double sgn, int measure, double* out) { double dz[15]; diff15(z, Z, dz); gf::shell_stress_point(z, Z, dz, t, E, nu, sgn, measure, out); }
// value and directional derivative of the vertex gradient: seeds dy[18], dY[12], dtau[2]
void gfh_penalty_grad_dual(const double* y, const double* Y, const double* tau, const double* dy, const double* dY, const double* dtau,
                           double ad, double ar, double dt, double* gr, double* dgr) {
    gf::Dual yd[18], Yd[12], td[2], g[18];
    for (int k = 0; k < 18; ++k) yd[k] = {y[k], dy[k]};
    for (int k = 0; k < 12; ++k) Yd[k] = {Y[k], dY[k]};
    for (int k = 0; k < 2; ++k) td[k] = {tau[k], dtau[k]};
    gf::penalty_grad_t<gf::Dual>(yd, Yd, td, ad, ar, dt, g);
    for (int k = 0; k < 18; ++k) { gr[k] = g[k].v; dgr[k] = g[k].d; }
}
int gfh_sizes(int which) { return which == 0 ? gf::IM_SIZE : which == 1 ? gf::PB_STRIDE : gf::PB_SIZE; }
}
