"""The matrix-core leg of the parity suite under the driver's eyes.

The MFMA forms of the sweep - the gradient fold and K[B,S] . W[S,C] on v_mfma_f32_16x16x4_f32 (MF), the expanded form's distance
GEMM as bf16 x 3 split operands on v_mfma_f32_16x16x32_bf16 (XM) - are measured slower than the VALU forms
(profiles/r03_mfma_ab.txt, profiles/r06_mfma_ab.txt) and are not in the shipped library; `__graft_entry__.build()` builds them into
diffco_amd/libdcx_matrix.so for the widths that have them (12 and 16).  A process maps one libdcx, so this test starts a second
python with DCX_LIB pointing at that library and runs the "mfma" leg of tests/test_gpu_parity.py plus the three tests that need the
`mfma` / `xm` knobs there; cases on other widths are stubs in that library and skip inside the child (their other two legs ran
in this process against the shipped library)."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "diffco_amd", "libdcx_matrix.so")


def test_matrix_core_leg_of_the_parity_suite_in_its_own_process():
    assert os.path.exists(LIB), f"{LIB} is missing: __graft_entry__.build() makes it (make ONLY_WIDTHS='12 16' EXTRA=-DDCX_WITH_MATRIX_FORMS)"
    env = dict(os.environ, DCX_LIB=LIB)
    for k in ("DCX_MFMA", "DCX_XM", "DCX_XF"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-q", "-m", "gpu", "-p", "no:cacheprovider",
           "-k", "mfma or matrix_core", "-x"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout or "")[-3000:] + (r.stderr or "")[-1500:]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "matrix_forms_pytest.txt"), "w") as f:
        f.write(r.stdout or "")
    assert r.returncode == 0, tail
    m = re.search(r"(\d+) passed", r.stdout)
    assert m and int(m.group(1)) >= 120, tail     # the D = 12 / 16 cases of the leg + the three matrix-only tests (147 in round 6)
    assert "failed" not in r.stdout.splitlines()[-1], tail
    # and the library this process runs on really has no matrix-core form behind the knob
    from diffco_amd import _lib
    if "matrix" not in os.path.basename(_lib.LIB_PATH):
        with pytest.raises(_lib.DcxUnsupported):
            _lib.check(_lib.load().dcx_debug_set(b"mfma", 1))
