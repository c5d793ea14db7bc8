"""System.Random(seed) as shipped by .NET (the seeded constructor keeps the legacy Knuth subtractive
generator in .NET 6: `Net5CompatSeedImpl`), re-implemented so that the reference's console scene
(src/FrayTracer.Console/Program.fs:14, `System.Random(19)`) can be rebuilt without a .NET runtime
(SURVEY.md §8f-1, Appendix B).

WRITTEN FROM MEMORY OF THE BCL SOURCE; no .NET runtime exists in the build image to confirm it.  The
only checks are the widely quoted first outputs `Random(0).Next() = 1559595546` and
`Random(42).Next() = 1434747710` (tests/test_dotnet_random.py) — treat the console scene as best effort.
"""
import numpy as np

MBIG = 2147483647
MSEED = 161803398
F = np.float32


class Random:
    def __init__(self, seed):
        seed = int(seed)
        subtraction = MBIG if seed == -2147483648 else abs(seed)
        sa = [0] * 56
        mj = MSEED - subtraction
        sa[55] = mj
        mk = 1
        ii = 0
        for _ in range(1, 55):
            ii += 21
            if ii >= 55:
                ii -= 55
            sa[ii] = mk
            mk = mj - mk
            if mk < 0:
                mk += MBIG
            mj = sa[ii]
        for _ in range(1, 5):
            for i in range(1, 56):
                n = i + 30
                if n >= 55:
                    n -= 55
                sa[i] -= sa[1 + n]
                if sa[i] < 0:
                    sa[i] += MBIG
        self._sa, self._inext, self._inextp = sa, 0, 21

    def Next(self):
        i = self._inext + 1
        if i >= 56:
            i = 1
        p = self._inextp + 1
        if p >= 56:
            p = 1
        r = self._sa[i] - self._sa[p]
        if r == MBIG:
            r -= 1
        if r < 0:
            r += MBIG
        self._sa[i] = r
        self._inext, self._inextp = i, p
        return r

    def NextDouble(self):
        return self.Next() * (1.0 / MBIG)

    # ---- the reference's extension members (src/FrayTracer/Random.fs:8-40) ------------------------
    def range_01(self):
        return F(self.NextDouble())

    def range(self, lo, hi):
        lo, hi = F(lo), F(hi)
        return F(lo + F(self.range_01() * F(hi - lo)))

    def _vec(self):
        return np.array([self.range(-1, 1), self.range(-1, 1), self.range(-1, 1)], F)

    def pointInBall(self, radius):
        while True:
            v = self._vec()
            if F(F(v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]) <= F(1.0):
                return v * F(radius)

    def pointOnSphere(self, radius):
        while True:
            v = self._vec()
            l = F(F(v[0] * v[0] + v[1] * v[1]) + v[2] * v[2])
            if F(0.01) <= l <= F(1.0):
                return v / np.sqrt(l) * F(radius)
