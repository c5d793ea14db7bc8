"""Round 4: candidates tested per evaluation (per lane) of the Program.fs structure for the walk's early exits: 0 none, 1 after Items.[0] only (round 3), 2 after every evaluation,
3 with the subtract's sphere feeding the cap too (round 4).  Result: 9.58 / 5.93 / 5.57 / 3.62.  Usage: python lazy_sim.py"""
import numpy as np, sys
import os; exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'coop_sim.py')).read().split("# camera")[0])
W = 1000
nps = abs(np.sin(30.0))
pos = np.array([0, 0, -10.0]); fw = np.array([0, 0, 1.0]); right = np.array([1.0, 0, 0]) * nps; up = np.array([0, 1.0, 0]) * nps
eps = 0.01
def fold2(p, cap, mode):
    k = cell_of(p); ctr, o, lb = lists[k]
    dtc = np.linalg.norm(p - ctr)
    mn = torus(p, o[0]); tested = 0; ev = 1
    if mode >= 1 and mn <= cap: return mn, tested, ev
    for j in range(1, len(o)):
        tested += 1
        if not (mn > lb[j] - dtc): break
        i = o[j]
        if mn > np.linalg.norm(C[i] - p) - BR[i]:
            d = torus(p, i); ev += 1; mn = min(mn, d)
            if mode >= 2 and mn <= cap: break
    return mn, tested, ev
rs2 = np.random.RandomState(5)
tot = {0: [0, 0], 1: [0, 0], 2: [0, 0], 3: [0, 0]}
sup = (S1c, S1r * 1.001 + 0.02)
n = 0
for _ in range(2500):
    x = rs2.randint(0, W); y = rs2.randint(0, W)
    d = fw + (x / W - 0.5) * right + (y / W - 0.5) * up; d /= np.linalg.norm(d)
    o = pos.copy(); ln = 30.0
    while ln > 0:
        w = o - sup[0]; cc = w @ w - (sup[1] + eps) ** 2; b = w @ d
        if cc > 0 and (b >= 0 or cc - b * b > 0): break
        d1 = np.linalg.norm(o - S1c) - S1r; d2 = np.linalg.norm(o - S2c) - S2r
        cap = d1 if d1 >= eps else -np.inf
        # mode 3: cap also from subtract: value = max(-d2, max(U,d1)); if -d2>=eps no hit; U irrelevant once max(mn,d1) <= -d2
        cap3 = max(cap, -d2 if -d2 >= eps else -np.inf)
        for mode in (0, 1, 2):
            mn, t, e = fold2(o, cap, mode); tot[mode][0] += t; tot[mode][1] += e
        mn3, t, e = fold2(o, cap3, 2); tot[3][0] += t; tot[3][1] += e
        mn, _, _ = fold2(o, -np.inf, 0)
        A = max(mn, d1)
        v = max(-d2, A); n += 1
        if v < eps: break
        o = o + d * v; ln -= v
print("evals", n)
for m in tot: print("mode", m, "tested/eval %.2f prim evals/eval %.2f" % (tot[m][0] / n, tot[m][1] / n))
