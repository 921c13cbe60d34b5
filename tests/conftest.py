import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG_NAME = "editable-gaussian-reflections_amd"
# Team help (egr_set_team_help) is automatic by default: ON for under-filled ranks of a partition - and a test image is small enough to make every
# partitioned tracer of this suite such a rank. With help the ORDER of exact depth ties of bounce rays depends on timing, and many tests assert
# bit-level equalities between a partitioned and a whole-image tracer; so the suite pins help OFF at creation (read by egr_create) and the tests of the
# help protocols switch it on explicitly (test_team_help_changes_the_list_order_only, test_team_help_does_not_drift_training, the two-rank bench).
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
