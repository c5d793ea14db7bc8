"""Round 4, before building anything: would a wave-cooperative walk of the grid union (all lanes of a cell group test the cell's candidates together, survivors walked in
lockstep) test fewer candidates per wave than the per-lane pair walk?  numpy model of the Program.fs structure (1000 random tori, the reference's grid, 8x8 tiles marched in lockstep
through march / normal / shadow phases).  Result (1000^2, 40 tiles): 2.38 cell groups per wave-evaluation, 6.16 survivors per group = 14.7 per wave-evaluation against 7.1 pair trips = 14.2
candidate slots of the per-lane walk: no saving, not built.  Usage: python coop_sim.py [size] [tiles]"""
import numpy as np, sys
rs = np.random.RandomState(19)
F = np.float32
N = 1000
def in_ball(n, R):
    out = []
    while len(out) < n:
        v = rs.uniform(-1, 1, 3)
        if v @ v <= 1: out.append(v * R)
    return np.array(out)
def on_sphere(n):
    out = []
    while len(out) < n:
        v = rs.uniform(-1, 1, 3); l = v @ v
        if 0.01 <= l <= 1: out.append(v / np.sqrt(l))
    return np.array(out)
C = in_ball(N, 4.0); Nn = on_sphere(N); R = rs.uniform(0.1, 0.4, N); r = rs.uniform(0.1, 0.3, N)
BR = R + r
def torus(p, i):   # p (...,3), i index array broadcast
    c = C[i]; n = Nn[i]
    dp = ((p - c) * n).sum(-1)
    q = (p - c) - dp[..., None] * n
    dc = np.sqrt((q * q).sum(-1)) - R[i]
    return np.sqrt(dp * dp + dc * dc) - r[i]
# grid
amin = (C - BR[:, None]).min(0); amax = (C + BR[:, None]).max(0); asz = amax - amin
cs0 = 1.5 * BR.mean(); cnt = max(1, int(np.ceil(asz[0] / cs0))); cell = asz / cnt
print("grid", cnt, "cell", cell)
half = np.linalg.norm(cell / 2)
lists = {}
for ix in range(cnt):
    for iy in range(cnt):
        for iz in range(cnt):
            ctr = amin + cell * 0.5 + cell * np.array([ix, iy, iz])
            d = np.linalg.norm(C - ctr, axis=1)
            ub = (d + BR).min() + half
            lb = d - BR
            keep = np.nonzero(lb < ub)[0]
            o = keep[np.argsort(lb[keep], kind='stable')]
            lists[(ix, iy, iz)] = (ctr, o, lb[o])
print("avg list", np.mean([len(v[1]) for v in lists.values()]))
S1c = np.zeros(3); S1r = 3.5; S2c = np.array([-0.5, 1, -2.0]); S2r = 2.5
def cell_of(p):
    c = np.floor((p - amin) / cell).astype(int)
    return tuple(np.clip(c, 0, cnt - 1))
def fold(p):
    """returns mn, n_tested(lb pass incl the failing one), n_eval, cellkey"""
    k = cell_of(p); ctr, o, lb = lists[k]
    dtc = np.linalg.norm(p - ctr)
    mn = torus(p, o[0]); tested = 0; ev = 1
    for j in range(1, len(o)):
        tested += 1
        if not (mn > lb[j] - dtc): break
        i = o[j]
        if mn > np.linalg.norm(C[i] - p) - BR[i]:
            d = torus(p, i); ev += 1; mn = min(mn, d)
    return mn, tested, ev, k, dtc
def value(p):
    U, t, e, k, dtc = fold(p)
    d1 = np.linalg.norm(p - S1c) - S1r
    A = max(U, d1) if U < np.linalg.norm(p - S1c) + S1r else U
    return max(-(np.linalg.norm(p - S2c) - S2r), A), U, t, e, k, dtc
# camera
W = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
nps = abs(np.sin(30.0))
pos = np.array([0, 0, -10.0]); fw = np.array([0, 0, 1.0]); right = np.array([1.0, 0, 0]) * nps; up = np.array([0, 1.0, 0]) * nps
L = -np.array([-0.5, -1, 1.0]); L /= np.linalg.norm(L)
eps = 0.01
ntiles = int(sys.argv[2]) if len(sys.argv) > 2 else 60
stats = dict(wave_evals=0, lane_evals=0, groups=0, lane_tested=0, lane_eval=0, wave_trips=0, surv=0, surv_eval_lanes=0, passes=0, pass_cands=0, surv2=0)
tl = W // 8
for t in range(ntiles):
    # pick a tile that likely hits the object: centre region
    tx = rs.randint(tl // 6, tl - tl // 6); ty = rs.randint(tl // 6, tl - tl // 6)
    lanes = []
    for i in range(64):
        x = tx * 8 + (i >> 3); y = ty * 8 + (i & 7)
        d = fw + (x / W - 0.5) * right + (y / W - 0.5) * up; d /= np.linalg.norm(d)
        lanes.append(dict(o=pos.copy(), d=d, len=30.0, ph='M', k=0))
    while True:
        act = [l for l in lanes if l['ph'] != 'D']
        if not act: break
        pts = []
        for l in act:
            if l['ph'] in ('M', 'S'): q = l['o']
            else:
                base = l['o'] - l['d'] * eps; h = eps * 0.125; q = base.copy()
                if l['ph'] in ('NX', 'NY', 'NZ'): q['XYZ'.index(l['ph'][1])] += h
            pts.append(q)
        res = [value(q) for q in pts]
        stats['wave_evals'] += 1; stats['lane_evals'] += len(act)
        # per-lane walk cost
        stats['lane_tested'] += sum(r_[2] for r_ in res); stats['lane_eval'] += sum(r_[3] for r_ in res)
        stats['wave_trips'] += max((r_[2] + 1) // 2 + 0 for r_ in res)   # pairs per trip, wave max (plus evaluations resume... approx)
        # coop per group
        keys = {}
        for q, r_ in zip(pts, res): keys.setdefault(r_[4], []).append((q, r_))
        stats['groups'] += len(keys)
        for k, mem in keys.items():
            ctr, o, lb = lists[k]
            P = np.array([m[0] for m in mem]); m0 = np.array([torus(p_, o[0]) for p_ in P])
            M = m0.max(); DTC = max(m[1][5] for m in mem)
            q0 = 0.5 * (P[0] + P[np.argmax(((P - P[0]) ** 2).sum(1))]); rho = np.sqrt(((P - q0) ** 2).sum(1).max())
            nlb = int(np.searchsorted(lb - DTC, M, side='left'))   # candidates with lb - DTC < M
            nlb = max(nlb, 1)
            stats['passes'] += (nlb - 1 + 63) // 64; stats['pass_cands'] += nlb - 1
            idx = o[1:nlb]
            md = np.linalg.norm(C[idx] - q0, axis=1) - rho - BR[idx]
            sv = int((md < M).sum())
            stats['surv'] += sv
            # tighter: use per-candidate min over lanes? (exact per lane test is what survivors walk does)
        for l, r_ in zip(act, res):
            v = r_[0]
            if l['ph'] in ('M', 'S'):
                if v < eps:
                    if l['ph'] == 'M': l['ph'] = 'NX'
                    else: l['ph'] = 'D'
                else:
                    l['o'] = l['o'] + l['d'] * v; l['len'] -= v
                    if l['len'] <= 0 or (np.linalg.norm(l['o']) > 6.5 and l['o'] @ l['d'] > 0): l['ph'] = 'D'
            elif l['ph'] == 'NX': l['ph'] = 'NY'
            elif l['ph'] == 'NY': l['ph'] = 'NZ'
            elif l['ph'] == 'NZ': l['ph'] = 'NC'
            elif l['ph'] == 'NC':
                l['o'] = l['o'] - l['d'] * eps; l['d'] = L; l['len'] = 1000.0; l['ph'] = 'S'
s = stats
print(s)
we = s['wave_evals']
print("lanes/wave-eval %.1f  groups/wave-eval %.2f  tested/lane-eval %.1f  evals/lane-eval %.2f  wave trips %.1f" % (s['lane_evals'] / we, s['groups'] / we, s['lane_tested'] / s['lane_evals'], s['lane_eval'] / s['lane_evals'], s['wave_trips'] / we))
print("coop: passes/group %.2f  cands/group %.1f  survivors/group %.2f  survivors/wave-eval %.2f" % (s['passes'] / s['groups'], s['pass_cands'] / s['groups'], s['surv'] / s['groups'], s['surv'] / we))
