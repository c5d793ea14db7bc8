"""How far can "the F# reference on some libm" be from the oracle?  SdfForm.unionSmooth calls MathF.Exp / MathF.Log
(SdfForm.fs:80,82) — platform libm in .NET, not bit-stable — and oracle and kernel substitute one fixed exp / log
(DESIGN.md section 2).  The oracle can be switched to the C library's expf / logf (orc_set_libm): this test renders the
headline scene (C3: unionSmooth of 256 spheres) both ways and keeps the measured distance on record — most pixels
identical, a fraction of a per cent beyond 1e-4 relative, none beyond 2e-3, no ray changing between hit and miss.
It bounds the effect of the substitution; it does not pin parity with the F# program (there is no .NET here)."""
import numpy as np

from fraytracer_amd import synthetic as syn

EPS, LEN = syn.EPSILON, syn.RAY_LENGTH


def test_c3_with_the_c_librarys_exp_and_log(oracle):
    scene, _ = syn.config3()
    cam = syn.default_camera()
    os_ = oracle.Oracle().scene(scene)
    n = 192
    try:
        oracle.lib.orc_set_libm(0)
        a, ca = os_.render(EPS, LEN, n, n, cam.as_array())
        oracle.lib.orc_set_libm(1)
        b, cb = os_.render(EPS, LEN, n, n, cam.as_array())
    finally:
        oracle.lib.orc_set_libm(0)
    assert ca["hits_primary"] == cb["hits_primary"] and ca["hits_shadow"] == cb["hits_shadow"]      # no hit / miss flips
    same = (a.view(np.uint32) == b.view(np.uint32)).all(axis=2)
    rel = (np.abs(a.astype(np.float64) - b) / np.maximum(np.abs(a.astype(np.float64)), 1e-3)).max(axis=2)
    frac_same, frac_over, worst = float(same.mean()), float((rel > 1e-4).mean()), float(rel.max())
    print(f"C3 {n}x{n}: identical {frac_same:.3f}, > 1e-4 relative {frac_over:.4f}, max {worst:.2e}")
    assert frac_same > 0.70
    assert 0.0 < frac_over < 0.01            # the substitution is visible, at the fraction-of-a-per-cent level
    assert worst < 2e-3
