"""Minimal IGES reader for rational B-spline surfaces (entity 128) -- SURVEY.md 8(f) N2.

The reference imports CAD geometry through pythonOCC (``read_igs_file`` / ``BSpline_surface`` of PENGoLINS'
OCC preprocessing, used e.g. in demos_om/shape_opt/T-beam/T_beam_shape_opt_wint.py:69-78); here the same
surfaces are read directly from the file.  Supported: fixed-format ASCII IGES, directory entries of type 128
(trimmed-surface wrappers 144 and groups 402 are skipped: GOLDFISH uses the untrimmed B-spline surfaces),
Hollerith strings in the global section, D/E exponents.  Returns goldfish_amd.splines.NURBSPatch objects."""
import re

import numpy as np

from ..splines import NURBSPatch


def _sections(path):
    sec = {"S": [], "G": [], "D": [], "P": [], "T": []}
    with open(path, "r") as f:
        for line in f:
            line = line.rstrip("\n").ljust(80)
            if line[72] in sec:
                sec[line[72]].append(line)
    return sec


def _delims(glob):
    """Parameter and record delimiters from the global section (defaults ',' and ';')."""
    pd, rd = ",", ";"
    if glob.startswith("1H"):
        pd = glob[2]
        rest = glob[3 + 1:] if glob[3] == pd else glob[3:]
        if rest.startswith("1H"):
            rd = rest[2]
    elif glob.startswith(","):
        if glob[1:].startswith("1H"):
            rd = glob[3]
    return pd, rd


def _num(tok):
    tok = tok.strip()
    if not tok:
        return 0.0
    return float(re.sub(r"[dD]", "E", tok))


def read_iges_surfaces(path):
    """All entity-128 surfaces of the file, in directory order."""
    sec = _sections(path)
    pd, rd = _delims("".join(l[:72] for l in sec["G"]).lstrip())
    pdata = {}
    for l in sec["P"]:
        de = int(l[64:72])
        pdata.setdefault(de, []).append(l[:64])
    out = []
    D = sec["D"]
    for k in range(0, len(D) - 1, 2):
        etype = int(D[k][0:8])
        if etype != 128:
            continue
        seq = int(D[k][73:80])
        txt = "".join(pdata[seq])
        txt = txt[:txt.index(rd)] if rd in txt else txt
        v = [t for t in txt.split(pd)]
        assert int(_num(v[0])) == 128
        K1, K2, M1, M2 = (int(_num(x)) for x in v[1:5])
        n1, n2 = K1 + 1, K2 + 1
        a, b = n1 + M1 + 1, n2 + M2 + 1                   # knot counts
        pos = 10
        U = np.array([_num(x) for x in v[pos:pos + a]]); pos += a
        V = np.array([_num(x) for x in v[pos:pos + b]]); pos += b
        W = np.array([_num(x) for x in v[pos:pos + n1 * n2]]); pos += n1 * n2
        X = np.array([_num(x) for x in v[pos:pos + 3 * n1 * n2]]).reshape(n1 * n2, 3); pos += 3 * n1 * n2
        u0, u1, v0, v1 = (_num(x) for x in v[pos:pos + 4])
        # control points are stored with the first index fastest, like NURBSPatch (flat = i + j * n_u)
        U = (U - U[0]) / (U[-1] - U[0]); V = (V - V[0]) / (V[-1] - V[0])
        cp_hom = np.concatenate([X * W[:, None], W[:, None]], 1).reshape(n2, n1, 4).transpose(1, 0, 2)
        out.append(NURBSPatch((M1, M2), [U, V], cp_hom))
    return out
