import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG_NAME = "editable-gaussian-reflections_amd"
# Team help (egr_set_team_help) is ON by default, and with it the ORDER of exact depth ties of bounce rays depends on timing (as upstream, where it is the
# PPLL's insertion order). Many tests assert bit-level equalities - a frame rendered twice, a partitioned against a whole-image tracer, strands, fuse_live -
# so the suite pins help OFF at creation (read by egr_create) and the tests of the help protocols switch it on explicitly
# (test_team_help_changes_the_list_order_only, test_team_help_does_not_drift_training, the two-rank bench); smoke() and bench.py run the default.
os.environ.setdefault("EGR_TEAM_HELP", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module(PKG_NAME)


@pytest.fixture(scope="session")
def syn():
    return importlib.import_module(PKG_NAME + ".synthetic")


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.build()
    return oracle
