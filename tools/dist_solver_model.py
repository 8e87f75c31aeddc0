"""Measurement + model (not product): the distributed factorisation (goldfish_amd/_dsolver.py) of C4's K for 2, 4, 8 ranks from the symbolic phase alone -- work of
the replicated top, largest per-rank share of the subtrees, bytes of the Schur-complement all-gather and of the boundary-contribution all-gather -- priced with the
rates measured on one MI355X (profiles/r05_v7_bench.json: 43 TFLOP/s for a factorisation; all-gather over xGMI taken as 300 GB/s per GPU).
Round 5: the K VALUE EXCHANGE is a term of the model -- this code replicates K's values on every rank before it factors
(ShardedDeviceModel.refresh_k_values: one all-gather of the owned value rows as device buffers + one device gather into the global CSR order, 9 x 8 bytes per
block of the pattern): (world - 1) / world of K's bytes in per rank over xGMI, and a pass over 2 x K's bytes at HBM rate for the permutation."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goldfish_amd import _lib, _nd, _solver, _dsolver, geometry as G
from goldfish_amd.model import arrays_from_spec
n = int(os.environ.get("GF_ND_PATCHES", "16"))
if os.environ.get("GF_MODEL", "c4") == "c5":            # the 1024-patch quartic fuselage skin: its factors do not fit one GPU -- what do they take on 8?
    os.environ.setdefault("GF_SCRATCH_GB", "16")        # only the pattern is needed here: small record chunks
    spec, name = G.synthetic_fuselage(32, 32, nel=53, p=4, jitter=2), "C5"
else:
    spec, name = G.synthetic_shell(n, n, nel=48, p=3, jitter=2), "C4"
A = arrays_from_spec(spec)
D = _lib.DeviceModel(A)
nb_ptr, nb = D.cp_graph()
from goldfish_amd import sharding
cp_off = np.concatenate([[0], np.cumsum([p_.ncp for p_ in spec.patches])])
X = np.stack([A.cp_hom[f] / A.weights for f in range(3)], 1)
D.close()
sym = _nd.nested_dissection_native(nb_ptr, nb, X, leaf=128)[0]
ne, nbd, be, bb = sym.front_dofs()
bt = be + bb
flop = 2.0 * 64 ** 3 * _dsolver.front_work(sym)
tiles = bt * (bt + 1) // 2
RATE, XGMI, HBM = 43e12, 300e9, 4.5e12       # end of round 5: 43.3 - 43.8 TFLOP/s per factorisation (profiles/r05_v7_bench.json)
k_bytes = 9.0 * 8.0 * float(nb_ptr[-1])                  # K's values: 9 doubles per block of the control-point pattern
print(name + ": %d fronts, %.2f Tflop, %.1f GB of tiles; one GPU: %.0f ms per factorisation at %.0f TFLOP/s" % (sym.nfronts, flop.sum() / 1e12, tiles.sum() * 32768 / 1e9, flop.sum() / RATE * 1e3, RATE / 1e12))
for world in ((8, 16) if name == "C5" else (2, 4, 8)):
    # round 5: the tree the distributed solver uses follows the patch partition (partition_tree); K is not replicated, only the separator rows travel
    part = sharding.partition_patches(spec, world)
    owner_cp = np.repeat(part, np.diff(cp_off))
    symp, ownerp, rootsp = _dsolver.partition_tree(nb_ptr, nb, X, owner_cp, world, leaf=128)
    nep, nbdp, bep, bbp = symp.front_dofs()
    btp = bep + bbp
    flopp = 2.0 * 64 ** 3 * _dsolver.front_work(symp)
    tilesp = btp * (btp + 1) // 2
    topp = ownerp == -1
    perp = np.array([flopp[ownerp == r].sum() for r in range(world)])
    memp = np.array([tilesp[ownerp == r].sum() for r in range(world)]) * 32768.0
    schurp = np.array([bbp[t] * (bbp[t] + 1) // 2 for t in rootsp]) * 32768.0
    S_cps = int(sum(symp.elim_off[t + 1] - symp.elim_off[t] for t in np.flatnonzero(topp)))
    rows_S = float(sum(nb_ptr[a + 1] - nb_ptr[a] for t in np.flatnonzero(topp) for a in symp.elim[symp.elim_off[t]:symp.elim_off[t + 1]])) * 72.0
    # if every separator front were factored by ONE rank instead of all of them (tree-parallel top: fronts of one level of the rank hierarchy side by side):
    # the critical path = the largest front of every level
    depth = np.zeros(symp.nfronts, np.int64)
    for t in range(symp.nfronts - 1, -1, -1):
        if symp.parent[t] >= 0:
            depth[t] = depth[symp.parent[t]] + 1
    crit = sum(flopp[(topp) & (depth == d)].max() for d in np.unique(depth[topp])) if topp.any() else 0.0
    t_subp, t_topp, t_agp, t_kp = perp.max() / RATE, flopp[topp].sum() / RATE, schurp.sum() * (world - 1) / world / XGMI, rows_S * (world - 1) / world / XGMI
    print("%d ranks, partition-following tree (what _dsolver.py runs): %d subtree roots below %d separator fronts (%d control points = %.1f %% of the model); %.2f Tflop in all "
          "(free dissection: %.2f); subtrees %.2f Tflop (largest share %.2f, imbalance %.2f), separators %.2f Tflop (replicated); K values that travel: <= %.2f GB (the separators' rows; "
          "replicated K: %.2f GB); Schur all-gather %.2f GB; factor memory per rank %.1f GB (own) + %.1f GB (top) + %.1f GB (stubs); if every rank repeated the separator fronts: K rows %.1f + subtrees %.0f "
          "+ Schur all-gather %.0f + top %.0f = %.0f ms; AS BUILT (every separator front by one rank, the fronts of a level side by side): critical path of the top %.2f Tflop = %.0f ms -> "
          "modelled factorisation %.0f ms"
          % (world, len(rootsp), int(topp.sum()), S_cps, 100.0 * S_cps / (nb_ptr.size - 1), flopp.sum() / 1e12, flop.sum() / 1e12, perp.sum() / 1e12, perp.max() / 1e12, perp.max() / perp.mean(),
             flopp[topp].sum() / 1e12, rows_S / 1e9, k_bytes / 1e9, schurp.sum() / 1e9, memp.max() / 1e9, tilesp[topp].sum() * 32768 / 1e9, schurp.sum() / 1e9,
             t_kp * 1e3, t_subp * 1e3, t_agp * 1e3, t_topp * 1e3, (t_kp + t_subp + t_agp + t_topp) * 1e3, crit / 1e12, crit / RATE * 1e3,
             (t_kp + t_subp + t_agp + crit / RATE) * 1e3), flush=True)
    owner, roots = _dsolver.split_tree(sym, world)
    top = owner == -1
    per = np.array([flop[owner == r].sum() for r in range(world)])
    mem = np.array([tiles[owner == r].sum() for r in range(world)]) * 32768.0
    schur = np.array([bb[t] * (bb[t] + 1) // 2 for t in roots]) * 32768.0
    fb = np.array([nbd[t] for t in roots]) * 8.0
    t_sub, t_top, t_ag = per.max() / RATE, flop[top].sum() / RATE, schur.sum() * (world - 1) / world / XGMI
    t_k = k_bytes * (world - 1) / world / XGMI + 2.0 * k_bytes / HBM
    print("%d ranks, round 4's scheme (subtrees of the free dissection dealt by work, K replicated): %d subtrees below %d top fronts; subtrees %.2f Tflop (largest share %.2f, imbalance %.2f), top %.2f Tflop (replicated); Schur all-gather %.2f GB, boundary "
          "contributions %.1f MB per solve; factor memory per rank %.1f GB (own) + %.1f GB (top) + %.1f GB (stubs); modelled factorisation: K value exchange (%.2f GB replicated) %.0f + subtrees %.0f + Schur all-gather %.0f + top %.0f = %.0f ms"
          % (world, len(roots), int(top.sum()), per.sum() / 1e12, per.max() / 1e12, per.max() / per.mean(), flop[top].sum() / 1e12, schur.sum() / 1e9, fb.sum() / 1e6,
             mem.max() / 1e9, tiles[top].sum() * 32768 / 1e9, schur.sum() / 1e9, k_bytes / 1e9, t_k * 1e3, t_sub * 1e3, t_ag * 1e3, t_top * 1e3, (t_k + t_sub + t_ag + t_top) * 1e3), flush=True)
