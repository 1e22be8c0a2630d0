import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
FIXTURES = ["ref_4x40_g100_seed1", "ref_4x40_g200_seed2", "ref_4x20_g100_seed3_episode",
            "ref_16x200_g100_seed4_parts", "ref_4x40_g100_seed5_walls"]


@pytest.fixture(scope="session", params=FIXTURES)
def golden(request):
    from fixture_io import load_fixture, regenerate_draws

    fx = load_fixture(os.path.join(GOLDEN_DIR, request.param + ".npz"))
    fx["name"] = request.param
    fx["draws"] = regenerate_draws(fx)
    return fx
