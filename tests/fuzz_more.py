"""More seeds of the random-scene parity test than the suite runs by default (not collected by `pytest tests`: run it by
name).  FUZZ_SEEDS="first:count" selects the seeds, default 1000:200."""
import os

import pytest

from test_gpu_parity import test_random_scenes_against_brute_force as one_scene
from test_gpu_parity import ctx                # noqa: F401  (fixture)

_first, _count = (int(v) for v in os.environ.get("FUZZ_SEEDS", "1000:200").split(":"))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(_first, _first + _count)))
def test_more_random_scenes(ctx, oracle, seed):   # noqa: F811
    one_scene(ctx, oracle, seed)
