#!/usr/bin/env python3
"""Generates tests/golden/kat_ts.json: 50-digit known-answer values of the Triple Sphere
formulas the reference evaluates (TS.h:117-125 projection, TS.h:39-57 unprojection,
TS.h:100-131 mono functor, multi_calib.h:146-195 multi functor with exact Rodrigues
rotations), computed with mpmath -- independent of both the oracle and the HIP code.
Intrinsics: EpipolarRectify/calib.yaml:7-10 (cam0).  Run:  python tests/golden/make_kat_ts.py
"""
import json
import os

from mpmath import mp, mpf, sqrt, sin, cos, matrix

mp.dps = 50
I = [mpf(x) for x in ("431.29641731951233", "430.77528857601646", "646.53015901902177", "521.20451427825685",
                      "-0.27125775332873053", "-0.087861849854000834", "0.56023435889162265")]


def project(P):
    X, Y, Z = P
    fx, fy, cx, cy, xi, lam, al = I
    d1 = sqrt(X * X + Y * Y + Z * Z)
    d2 = sqrt(X * X + Y * Y + (Z + xi * d1) ** 2)
    d3 = sqrt(X * X + Y * Y + (Z + xi * d1 + lam * d2) ** 2)
    k = Z + xi * d1 + lam * d2 + al / (1 - al) * d3
    return fx * X / k + cx, fy * Y / k + cy


def unproject(u, v):
    fx, fy, cx, cy, xi, lam, al = I
    mx, my = (u - cx) / fx, (v - cy) / fy
    ks = al / (1 - al)
    r2 = mx * mx + my * my
    g = (ks + sqrt(1 + (1 - ks * ks) * r2)) / (r2 + 1)
    yita = lam * (g - ks) + sqrt(((g - ks) ** 2 - 1) * lam * lam + 1)
    mz = yita * (g - ks)
    mu = xi * (mz - lam) + sqrt(xi * xi * ((mz - lam) ** 2 - 1) + 1)
    return mu * yita * g * mx, mu * yita * g * my, mu * (mz - lam) - xi


def rot(w, p):
    th = sqrt(sum(a * a for a in w))
    k = [a / th for a in w]
    kxp = [k[1] * p[2] - k[2] * p[1], k[2] * p[0] - k[0] * p[2], k[0] * p[1] - k[1] * p[0]]
    kd = sum(a * b for a, b in zip(k, p))
    return [p[i] * cos(th) + kxp[i] * sin(th) + k[i] * kd * (1 - cos(th)) for i in range(3)]


S = lambda x: mp.nstr(x, 30)
out = {"intrinsics": [float(x) for x in I], "project": [], "functor": []}
for P in ((mpf("0.1"), mpf("-0.2"), mpf(1)), (mpf(300), mpf(-150), mpf(400)), (mpf(500), mpf(200), mpf(-50)),
          (mpf(-40), mpf(10), mpf(120)), (mpf(0), mpf(0), mpf(250))):
    u, v = project(P)
    ray = unproject(u, v)
    n = sqrt(sum(a * a for a in P))
    err = max(abs(ray[i] - P[i] / n) for i in range(3))
    assert err < mpf(10) ** -40, err
    out["project"].append({"P": [float(a) for a in P], "u": S(u), "v": S(v)})

board_pt = (mpf(90), mpf(45), mpf(0))
brt = [mpf("0.1"), mpf("-0.2"), mpf("0.3"), mpf(-200), mpf(-100), mpf(600)]
crt = [mpf("0.02"), mpf("1.5"), mpf("-0.03"), mpf(311), mpf("-3.2"), mpf(-302)]
Pw = [a + b for a, b in zip(rot(brt[:3], board_pt), brt[3:])]
u, v = project(Pw)
out["functor"].append({"kind": "mono", "board_pt": [90.0, 45.0], "rt": [float(a) for a in brt],
                       "Pc": [S(a) for a in Pw], "u": S(u), "v": S(v)})
Pc = [a + b for a, b in zip(rot(crt[:3], Pw), crt[3:])]
u, v = project(Pc)
out["functor"].append({"kind": "multi", "board_pt": [90.0, 45.0], "board_rt": [float(a) for a in brt],
                       "cam_rt": [float(a) for a in crt], "Pc": [S(a) for a in Pc], "u": S(u), "v": S(v)})
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat_ts.json")
json.dump(out, open(path, "w"), indent=1)
print("wrote", path)
