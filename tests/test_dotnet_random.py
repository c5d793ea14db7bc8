"""System.Random port (fraytracer_amd/dotnet_random.py).  The two expected values are the commonly quoted
first outputs of .NET's seeded generator, recalled from memory — no .NET runtime exists here to confirm
them, so this pins the port against regressions rather than against .NET itself."""
import numpy as np

from fraytracer_amd.dotnet_random import Random
from fraytracer_amd import synthetic as syn


def test_first_outputs_of_known_seeds():
    assert Random(0).Next() == 1559595546
    assert Random(42).Next() == 1434747710
    assert Random(1).Next() == 534011718


def test_the_widely_quoted_sequence_of_seed_0():
    """`new Random(0)`: the first ten Next() values and the first NextDouble() as they are quoted all over the .NET
    literature (written down from memory BEFORE the port was run on them — an independent implementation and an
    independent recollection agreeing on 11 numbers is the strongest pin available without a .NET runtime)."""
    r = Random(0)
    assert [r.Next() for _ in range(10)] == [1559595546, 1755192844, 1649316166, 1198642031, 442452829,
                                              1200195957, 1945678308, 949569752, 2099272109, 587775847]
    assert abs(Random(0).NextDouble() - 0.72624326996796) < 5e-15


def test_samples_are_in_range_and_deterministic():
    a, b = Random(19), Random(19)
    xs = [a.NextDouble() for _ in range(2000)]
    assert xs == [b.NextDouble() for _ in range(2000)]
    assert 0.0 <= min(xs) and max(xs) < 1.0 and 0.45 < float(np.mean(xs)) < 0.55


def test_console_scene_builds():
    scene, size = syn.console_scene(n=50)
    assert (size.X, size.Y) == (1000, 1000)
    assert scene.Object.kind == "subtract" and scene.Object.kids[0].kind == "intersect"
    union = scene.Object.kids[0].kids[0]
    assert union.kind == "union" and len(union.kids) == 50
    c = np.array([k.kids[1].args[0] for k in union.kids], np.float32)
    assert np.all(np.linalg.norm(c, axis=1) <= 4.0 + 1e-5)                # pointInBall 4.0f
