import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_oracle():
    from oracle import oracle
    oracle.build()


_KNOBS = ("ys", "nw", "min_rows", "split_finish_kernel", "inlaunch_tiles", "jac_per_class", "mfma", "traj_fused", "xf", "jac_one_sweep", "train_grid", "fkk", "jt_waves", "hess_ys", "xm", "traj_ys", "traj_across", "owner_poll", "solve_threads", "qt", "giveup_inject")


@pytest.fixture
def knob():
    """knob(name, value): override one launch-geometry rule of libdcx through dcx_debug_set (value < 0 restores the
    rule); every knob goes back to its rule when the test ends"""
    from diffco_amd import _lib
    lib = _lib.load()

    def set_knob(name, value):
        _lib.check(lib.dcx_debug_set(name.encode(), int(value)))
    yield set_knob
    for k in _KNOBS:
        lib.dcx_debug_set(k.encode(), -1)
