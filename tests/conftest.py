import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SCENES = os.path.join(ROOT, "scenes")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    import flux_amd
    return flux_amd._lib.lib.flux_device_count() > 0


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly, not skip: the product has no fallback.
    pass


@pytest.fixture(scope="session")
def flux():
    import flux_amd
    return flux_amd


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle
    return oracle


@pytest.fixture(scope="session")
def demo1(flux):
    return flux.load_scene(os.path.join(SCENES, "demo1.yml"))


@pytest.fixture(scope="session")
def demo2(flux):
    return flux.load_scene(os.path.join(SCENES, "demo2.yml"))


def small_scene(sd, width, height):
    """Same camera/field of view as `sd` rendered at width x height (pixel_size scaled)."""
    import copy
    s = copy.deepcopy(sd)
    scale = sd.output_settings.image_width / width
    s.output_settings.image_width = width
    s.output_settings.image_height = height
    s.output_settings.pixel_size = sd.output_settings.pixel_size * scale
    return s


@pytest.fixture(scope="session")
def small():
    return small_scene


def max_abs_diff(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))))
