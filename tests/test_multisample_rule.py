"""Camera::multisample's stopping rule (camera.rs:273-310) on the ORACLE, pinned by sample counts derived from the
reference source by hand (tests/golden/make_multisample_counts.py): the reference seeds its sum of squares with the
first sample un-squared (camera.rs:277), so a flat pixel of colour c takes ~ sqrt(mean(c - c^2)) / SE samples - 51 for
mid-grey at SE = 1e-2 as IEEE f64 rounds the boundary case n = 50 - and not min_samples."""
import numpy as np
import pytest

import oracle_lib as orc
from multisample_cases import CASES, H, V, constant_scene


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_sample_counts(case):
    world, cam = constant_scene(case)
    img, st = orc.render(world, cam, n_threads=2)
    assert st["samples"] == case["samples"] * H * V
    assert st["primary"] == st["samples"]
    assert np.allclose(img, np.asarray(case["mean"])[None, None, :], rtol=0, atol=1e-12)


def test_a_squared_seed_would_stop_at_min_samples():
    """What the fixture rules out: with sos seeded by c * c a flat pixel has zero variance and takes min_samples."""
    grey = next(c for c in CASES if c["name"].startswith("mid grey"))
    assert grey["samples"] > 10 * grey["min_samples"]
