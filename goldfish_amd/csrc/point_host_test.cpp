// Host-side entry points into kl_point.hpp (the same header the kernels include), used by
// tests/test_pointwise_forms.py to check the device closed forms on the CPU.
#include "kl_point.hpp"
extern "C" {
void gfh_shell_point(const double* z, const double* Z, double t, double E, double nu, double* im, double* Pzz, double* PzZ, double* Pz, double* Pzt) {
    for (int k = 0; k < gf::IM_SIZE; ++k) im[k] = 0.0;
    gf::shell_point(z, Z, t, E, nu, im);
    for (int r = 0; r < 15; ++r) { Pz[r] = im[gf::IM_PZ + r]; Pzt[r] = gf::pzt_entry(im, r); }
    for (int r = 0; r < 15; ++r) for (int s = 0; s < 15; ++s) { Pzz[15 * r + s] = gf::pzz_entry(im, r, s); PzZ[15 * r + s] = gf::pzZ_entry(im, r, s); }
}
void gfh_shell_point_cols(const double* z, const double* Z, double t, double E, double nu, double* im) {
    for (int k = 0; k < gf::IM_SIZE; ++k) im[k] = 0.0;
    for (int ic = 0; ic < 3; ++ic) {
        const double d[3] = {ic == 0 ? 1.0 : 0.0, ic == 1 ? 1.0 : 0.0, ic == 2 ? 1.0 : 0.0};
        gf::shell_point_cols(z, Z, t, E, nu, ic, d, ic == 0, im);
    }
}
void gfh_penalty_point(const double* y, const double* Y, const double* tau, double ad, double ar, double dt, double* out) {
    gf::penalty_point(y, Y, tau, ad, ar, dt, out);
}
void gfh_shell_energy_point(const double* z, const double* Z, double t, double E, double nu, double* out) { gf::shell_energy_point(z, Z, t, E, nu, out); }
void gfh_shell_stress_point(const double* z, const double* Z, double t, double E, double nu, double sgn, int measure, double* out) { gf::shell_stress_point(z, Z, t, E, nu, sgn, measure, out); }
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
