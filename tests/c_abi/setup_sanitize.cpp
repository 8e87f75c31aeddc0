// Host-side preprocessing (goldfish_amd/csrc/gf_setup.hpp: element tables, neighbour lists, mortar-vertex tables, owner
// lists -- the index-heavy part of gf_create) under AddressSanitizer + UBSan on the CPU (GPU sanitizers are not available
// on the pool).  Reads a model dumped by tests/test_host_logic.py (flat binary: header of int64 counts, then the arrays of
// gf_model_desc in declaration order) and runs HostModel::build.  Exit code 0 = built without a sanitizer report.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../goldfish_amd/csrc/gf_setup.hpp"

template <class T> static std::vector<T> rd(FILE* f, int64_t n) {
    std::vector<T> v((size_t)n);
    if (n > 0 && fread(v.data(), sizeof(T), (size_t)n, f) != (size_t)n) { fprintf(stderr, "short read\n"); exit(3); }
    return v;
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    int64_t hd[8];                     // n_patches, n_knots, total_cp, n_zero, n_pl, n_if, npts, n_owned
    if (fread(hd, sizeof(int64_t), 8, f) != 8) return 3;
    const int64_t np = hd[0], nk = hd[1], tcp = hd[2], nz = hd[3], npl = hd[4], ni = hd[5], npts = hd[6];
    auto degree = rd<int32_t>(f, 2 * np); auto ncp = rd<int32_t>(f, 2 * np);
    auto knot_off = rd<int64_t>(f, 2 * np + 1); auto knots = rd<double>(f, nk);
    auto cp_off = rd<int64_t>(f, np + 1); auto weights = rd<double>(f, tcp);
    auto young = rd<double>(f, np); auto poisson = rd<double>(f, np); auto body = rd<double>(f, 3 * np);
    auto zero = rd<int64_t>(f, nz); auto pl_dof = rd<int64_t>(f, npl); auto pl_val = rd<double>(f, npl);
    auto if_patch = rd<int32_t>(f, 2 * ni); auto if_off = rd<int64_t>(f, ni + 1);
    auto if_xi = rd<double>(f, 4 * npts); auto if_tau = rd<double>(f, 2 * npts); auto if_wt = rd<double>(f, npts);
    auto if_alpha = rd<double>(f, 2 * ni); auto load_proj = rd<double>(f, 3 * np);
    fclose(f);
    gf_model_desc d = {};
    d.n_patches = (int32_t)np; d.degree = degree.data(); d.ncp = ncp.data(); d.knot_off = knot_off.data(); d.knots = knots.data();
    d.cp_off = cp_off.data(); d.weights = weights.data(); d.young = young.data(); d.poisson = poisson.data(); d.body_force = body.data();
    d.n_zero_dofs = nz; d.zero_dofs = nz ? zero.data() : nullptr; d.n_point_loads = npl; d.pl_dof = npl ? pl_dof.data() : nullptr; d.pl_val = npl ? pl_val.data() : nullptr;
    d.n_interfaces = (int32_t)ni;
    if (ni > 0) { d.if_patch = if_patch.data(); d.if_off = if_off.data(); d.if_xi = if_xi.data(); d.if_tau = if_tau.data(); d.if_wt = if_wt.data(); d.if_alpha = if_alpha.data(); }
    d.n_owned_patches = (int32_t)hd[7]; d.load_proj = load_proj.data();
    try {
        gf::HostModel H;
        H.build(&d);
        printf("built: %lld cps, %lld elements, %lld mortar points, %zu coupling entries, %zu visit entries\n", (long long)H.total_cp, (long long)H.nelem,
               (long long)H.npts, H.nb_c.size(), H.pen_entries.size());
    } catch (const std::exception& e) { fprintf(stderr, "build failed: %s\n", e.what()); return 4; }
    return 0;
}
