"""Round 4: evaluations per ray of the Program.fs structure with the support sphere of the union (round 3) and of the intersect's sphere child (round 4), numpy model.
Result: 13.88 -> 10.57 evaluations per ray (the kernel then measured 13.6 -> 10.3).  Usage: python esc_sim.py"""
import numpy as np, sys
import os; exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'coop_sim.py')).read().split("# camera")[0])
W = 1000
nps = abs(np.sin(30.0))
pos = np.array([0, 0, -10.0]); fw = np.array([0, 0, 1.0]); right = np.array([1.0, 0, 0]) * nps; up = np.array([0, 1.0, 0]) * nps
L = -np.array([-0.5, -1, 1.0]); L /= np.linalg.norm(L)
PL = np.array([-0.5, 0, -2.0])
eps = 0.01
supU = (C.mean(0), (np.linalg.norm(C - C.mean(0), axis=1) + BR).max())
def padded(c, r): return c, r * 1.001 + 0.01 + 1e-3 * np.abs(c).sum()
def never(sup, o, d, ln):
    c, r = sup; w = o - c; re = r + eps
    ww = w @ w; cc = ww - re * re; dd = d @ d
    if not cc > 0: return False
    b = w @ d
    if b >= 0: return True
    return cc * dd - b * b > 0
def run(sup, nrays=3000, lazy_any=False):
    rs2 = np.random.RandomState(5)
    ev = 0; tested = 0; walk_stop = 0; rays = 0
    for _ in range(nrays):
        x = rs2.randint(0, W); y = rs2.randint(0, W)
        d = fw + (x / W - 0.5) * right + (y / W - 0.5) * up; d /= np.linalg.norm(d)
        stack = [(pos.copy(), d, 30.0, 'M')]
        while stack:
            o, d, ln, ph = stack.pop(); rays += 1
            while True:
                if ln <= 0 or never(sup, o, d, ln): break
                v, U, t, e, k, dtc = value(o); ev += 1; tested += t
                if v < eps:
                    if ph == 'M':
                        ev += 4
                        hp = o - d * eps
                        # normal approx: gradient
                        h = 1e-3; g = np.array([value(hp + h * np.eye(3)[a])[0] - value(hp)[0] for a in range(3)]); n = g / (np.linalg.norm(g) + 1e-30)
                        if n @ L > 0: stack.append((hp, L, 1000.0, 'S'))
                        df = PL - hp
                        if n @ (df / np.linalg.norm(df)) > 0: stack.append((hp, df / (df @ df), np.linalg.norm(df), 'S'))
                    break
                o = o + d * v; ln -= v
    return ev, tested, rays
for name, sup in (("union support", padded(*supU)), ("S1 support", padded(S1c, S1r))):
    ev, tested, rays = run(sup)
    print(name, "radius %.2f" % sup[1], "evals", ev, "tested", tested, "rays", rays, "evals/ray %.2f" % (ev / rays))
