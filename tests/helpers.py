import numpy as np


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bit_equal(got, want, what=""):
    got = np.ascontiguousarray(got, dtype=np.float32)
    want = np.ascontiguousarray(want, dtype=np.float32)
    assert got.shape == want.shape, (got.shape, want.shape)
    g, w = got.view(np.uint32), want.view(np.uint32)
    # NaNs compare by NaN-ness (payload is not part of the contract), everything else bit for bit
    same = (g == w) | (np.isnan(got) & np.isnan(want))
    if not same.all():
        idx = np.argwhere(~same)
        first = tuple(idx[0])
        raise AssertionError(f"{what}: {len(idx)} of {got.size} values differ; first at {first}: "
                             f"got {got[first]!r} ({g[first]:#010x}) want {want[first]!r} ({w[first]:#010x})")
