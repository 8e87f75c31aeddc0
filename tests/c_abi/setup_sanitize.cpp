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
        const int seg = argc > 2 ? atoi(argv[2]) : 5;
        // tables of the row-record path + a dry run of its addressing: the element kernel's stores (which pair lands at which position of
        // which row record) against the gather's reads (every position a control point reads must have been written, by exactly the
        // pair the gather attributes to it; every written position is read exactly once as "a as A")
        if (H.degree <= 3) {
            H.build_rec(seg);
            const int P = H.degree, P1 = P + 1, NPOS = 112;
            std::vector<int64_t> recA((size_t)H.rec_items.size() * H.rec_rows * NPOS, -1), recB(recA.size(), -1);
            std::vector<int> nread(recA.size(), 0);
            for (size_t item = 0; item < H.rec_items.size(); ++item) {
                const gf::WalkItem& it = H.rec_items[item];
                const gf::PatchDev& Pt = H.patches[it.patch];
                const int ivf = H.ints[Pt.spv + it.ev0] - P;
                std::vector<char> held(64 * 4, 0);                              // accumulator slots that hold a pair not yet stored
                for (int t = 0; t < it.nel; ++t) {
                    const int ev = it.ev0 + t, iv0 = H.ints[Pt.spv + ev] - P;
                    const int iv0n = t + 1 < it.nel ? H.ints[Pt.spv + ev + 1] - P : iv0 + 4;
                    for (int lane = 0; lane < 64; ++lane) for (int rr = 0; rr < P1; ++rr) {
                        const int x = lane & 15, kk = lane >> 4, jub = x >> 2, sb = x & 3;
                        const int rowa = iv0 + ((kk - iv0) & 3), rowb = iv0 + ((sb - iv0) & 3);
                        if (!((rowa - iv0) < P1 && (rowb - iv0) < P1 && jub < P1)) continue;
                        const int rho = std::min(rowa, rowb);
                        if (rho >= iv0n) continue;
                        const int sl = rho & 3; const bool own = kk == sl; const int rk = ((kk - sl) & 3) - 1;
                        const int row = rho - ivf, pos = own ? rr * 16 + x : 64 + (rr * 3 + rk) * 4 + jub;
                        if (row < 0 || row >= H.rec_rows) { fprintf(stderr, "rec: row beyond the item's records\n"); return 6; }
                        const size_t w = ((size_t)item * H.rec_rows + row) * NPOS + pos;
                        if (recA[w] >= 0) { fprintf(stderr, "rec: position stored twice\n"); return 6; }
                        recA[w] = Pt.cp_off + (it.iu0 + rr) + int64_t(rowa) * Pt.nu; recB[w] = Pt.cp_off + (it.iu0 + jub) + int64_t(rowb) * Pt.nu;
                    }
                }
            }
            for (int64_t a = 0; a < H.owned_cp; ++a) {
                const gf::CpDesc& c = H.cp_desc[a]; const gf::RecCp& rc = H.rec_cp[a]; const gf::PatchDev& Pt = H.patches[c.patch];
                std::vector<int> gotA(H.nb_ptr_s[a + 1] - H.nb_ptr_s[a], 0), gotB(gotA.size(), 0);
                auto slot_of = [&](int64_t b) { for (int64_t k = H.nb_ptr_s[a]; k < H.nb_ptr_s[a + 1]; ++k) if (H.nb_s[k] == b) return int(k - H.nb_ptr_s[a]); return -1; };
                for (int n = 0; n < rc.nit; ++n) {
                    const int row = rc.it[n].row, iu0 = int(rc.it[n].info & 255u) + c.i0, rra = int((rc.it[n].info >> 8) & 255u); const unsigned pm = rc.it[n].info >> 16;
                    if (rra != c.ia - iu0 || rra < 0 || rra > P) { fprintf(stderr, "rec: wrong strip offset\n"); return 6; }
                    auto check = [&](int drow, int pos, int64_t A, int64_t B, bool asA) {
                        const int64_t w = (int64_t(row) + drow) * NPOS + pos;
                        if (w < 0 || (size_t)w >= recA.size() || recA[w] != A || recB[w] != B) { fprintf(stderr, "rec: the gather reads a position that does not hold its pair\n"); return false; }
                        const int sl = slot_of(asA ? B : A);
                        if (sl < 0) { fprintf(stderr, "rec: pair outside the neighbour box\n"); return false; }
                        if (asA) { nread[w]++; gotA[sl]++; } else gotB[sl]++;
                        return true;
                    };
                    auto cp = [&](int iu, int jv) { return Pt.cp_off + iu + int64_t(jv) * Pt.nu; };
                    for (int cc = 0; cc < 16; ++cc) {                                   // G1
                        const int jub = cc >> 2, dv = (cc - c.ja) & 3;
                        if (jub < P1 && dv < P1 && ((pm >> (3 + dv)) & 1) && !check(0, rra * 16 + cc, a, cp(iu0 + jub, c.ja + dv), true)) return 6;
                    }
                    for (int rk = 0; rk < 3; ++rk) for (int jub = 0; jub < 4; ++jub)    // G3
                        if (jub < P1 && rk + 1 < P1 && ((pm >> (2 - rk)) & 1) && !check(-1 - rk, 64 + (rra * 3 + rk) * 4 + jub, a, cp(iu0 + jub, c.ja - 1 - rk), true)) return 6;
                    for (int rk = 0; rk < 3; ++rk) for (int rr = 0; rr < 4; ++rr)       // G2
                        if (rr < P1 && rk + 1 < P1 && ((pm >> (4 + rk)) & 1) && !check(0, 64 + (rr * 3 + rk) * 4 + rra, cp(iu0 + rr, c.ja + 1 + rk), a, false)) return 6;
                    for (int d = 0; d < 4; ++d) for (int rr = 0; rr < 4; ++rr)          // G4
                        if (rr < P1 && d < P1 && ((pm >> (3 - d)) & 1) && !check(-d, rr * 16 + 4 * rra + (c.ja & 3), cp(iu0 + rr, c.ja - d), a, false)) return 6;
                }
                for (size_t k = 0; k < gotA.size(); ++k) if (gotA[k] < 1 || gotA[k] != gotB[k]) { fprintf(stderr, "rec: a box entry receives no (or unbalanced) contributions\n"); return 6; }
            }
            for (size_t w = 0; w < recA.size(); ++w) if ((recA[w] >= 0) != (nread[w] == 1)) { fprintf(stderr, "rec: a stored pair is not read exactly once\n"); return 6; }
            printf("rec: %zu items, %d rows per item, dry run ok\n", H.rec_items.size(), H.rec_rows);
        }
        printf("built: %lld cps, %lld elements, %lld mortar points, %zu coupling entries, %zu visit entries\n", (long long)H.total_cp, (long long)H.nelem,
               (long long)H.npts, H.nb_c.size(), H.pen_entries.size());
    } catch (const std::exception& e) { fprintf(stderr, "build failed: %s\n", e.what()); return 4; }
    return 0;
}
