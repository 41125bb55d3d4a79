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


_KNOBS = ("ys", "nw", "min_rows", "split_finish_kernel", "inlaunch_tiles", "jac_per_class", "mfma", "traj_fused", "xf", "jac_one_sweep", "train_grid", "fkk", "jt_waves", "hess_ys", "hess_form", "xm", "traj_ys", "traj_across", "owner_poll", "solve_threads", "qt", "giveup_inject", "skew", "skew8")


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


# ---- the process that runs against diffco_amd/libdcx_matrix.so (tests/test_gpu_matrix_forms.py starts it) --------------------
# That library is built for the two widths that HAVE matrix-core forms (12 and 16; every other width is a stub that answers
# hipErrorNotSupported), so a case on another width cannot run there: it is skipped IN THAT PROCESS (its "expanded" / "direct"
# legs ran in the main one against the shipped library).
_MATRIX_LIB = "matrix" in os.path.basename(os.environ.get("DCX_LIB", ""))


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_call(item):
    outcome = yield
    if _MATRIX_LIB and outcome.excinfo is not None:
        msg = str(outcome.excinfo[1])
        if "not supported" in msg or "hipErrorNotSupported" in msg:
            outcome.force_exception(pytest.skip.Exception("this width is a stub in the matrix-forms library"))
