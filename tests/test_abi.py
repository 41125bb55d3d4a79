"""CPU-side checks of the boundary: libdcx.so loads and exports every symbol include/dcx.h declares,
the ctypes mirror of dcx_fk_desc has the C layout, and the product fails loudly without a GPU.
No compute calls (there is no GPU here)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "dcx.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dcx_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from diffco_amd import _lib
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), f"libdcx.so does not export {n}"
    assert set(names) == set(_lib.SYMBOLS), "ctypes table and header disagree"
    assert lib.dcx_version() == 109
    assert isinstance(lib.dcx_device_count(), int)


def test_fk_desc_layout_matches_c(tmp_path):
    from diffco_amd._fkdesc import FkDesc
    prog = tmp_path / "sz.c"
    prog.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "dcx.h"\n'
                    'int main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu", sizeof(dcx_fk_desc), offsetof(dcx_fk_desc, n_chains),'
                    ' offsetof(dcx_fk_desc, base), offsetof(dcx_fk_desc, pt_off), offsetof(dcx_fk_desc, keypoints),'
                    ' offsetof(dcx_fk_desc, t_n_chains), offsetof(dcx_fk_desc, t_base), offsetof(dcx_fk_desc, t_scale),'
                    ' offsetof(dcx_fk_desc, t_fixed), offsetof(dcx_fk_desc, t_axis));}')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    assert [int(x) for x in out] == [ctypes.sizeof(FkDesc), FkDesc.n_chains.offset, FkDesc.base.offset,
                                     FkDesc.pt_off.offset, FkDesc.keypoints.offset, FkDesc.t_n_chains.offset,
                                     FkDesc.t_base.offset, FkDesc.t_scale.offset, FkDesc.t_fixed.offset,
                                     FkDesc.t_axis.offset]


def test_traj_struct_layouts_match_c(tmp_path):
    from diffco_amd._lib import TrajOpts, TrajState
    prog = tmp_path / "sz2.c"
    prog.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "dcx.h"\n'
                    'int main(){printf("%zu %zu %zu %zu %zu %zu", sizeof(dcx_traj_state), offsetof(dcx_traj_state, path),'
                    ' offsetof(dcx_traj_state, col_score), offsetof(dcx_traj_state, steps), sizeof(dcx_traj_opts),'
                    ' offsetof(dcx_traj_opts, valid_tol));}')
    exe = tmp_path / "sz2"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)], check=True)
    out = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    assert out == [ctypes.sizeof(TrajState), TrajState.path.offset, TrajState.col_score.offset, TrajState.steps.offset,
                   ctypes.sizeof(TrajOpts), TrajOpts.valid_tol.offset]


def test_escape_opts_layout_matches_c(tmp_path):
    from diffco_amd._lib import EscapeOpts
    prog = tmp_path / "sz3.c"
    prog.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "dcx.h"\n'
                    'int main(){printf("%zu %zu %zu %zu %zu", sizeof(dcx_escape_opts), offsetof(dcx_escape_opts, eps),'
                    ' offsetof(dcx_escape_opts, n_steps), offsetof(dcx_escape_opts, joint), offsetof(dcx_escape_opts, wrap_mask));}')
    exe = tmp_path / "sz3"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)], check=True)
    out = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    assert out == [ctypes.sizeof(EscapeOpts), EscapeOpts.eps.offset, EscapeOpts.n_steps.offset, EscapeOpts.joint.offset,
                   EscapeOpts.wrap_mask.offset]


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_no_gpu_means_loud_failure_not_cpu_fallback():
    from diffco_amd import _lib, kernel, model
    from diffco_amd.kernel_perceptrons import DiffCo
    lib = _lib.load()
    # the C ABI itself reports the missing device
    h = ctypes.c_void_p()
    kp = (ctypes.c_float * 2)(1.0, 1.0)
    buf = (ctypes.c_float * 12)()
    rc = lib.dcx_model_create(ctypes.byref(h), 0, None, 1, kp, ctypes.cast(buf, ctypes.c_void_p),
                              ctypes.cast(buf, ctypes.c_void_p), 1, 12, 1)
    assert rc == 4 and b"no HIP device" in lib.dcx_last_error()  # DCX_ERR_NO_DEVICE
    rob = model.BaxterLeftArmFK()
    with pytest.raises(_lib.DcxError):
        rob.fkine(torch.zeros(2, 7))
    with pytest.raises(_lib.DcxError):
        kernel.RQKernel(10.0)(torch.zeros(2, 3), torch.zeros(4, 3))
    dc = DiffCo(kernel_func=kernel.RQKernel(10.0), transform=rob.fkine)
    dc.support_points = torch.zeros(4, 7)
    dc.support_transformed = torch.zeros(4, 4, 3)
    dc.gains = torch.ones(4)
    with pytest.raises(_lib.DcxError):
        dc.score(torch.zeros(2, 7))


def test_argument_errors_before_device_use():
    from diffco_amd import _lib
    lib = _lib.load()
    h = ctypes.c_void_p()
    kp = (ctypes.c_float * 2)(1.0, 1.0)
    assert lib.dcx_model_create(None, 0, None, 1, kp, None, None, 0, 12, 1) == 1      # out == NULL
    assert lib.dcx_model_create(ctypes.byref(h), 0, None, 1, kp, None, None, 5, 12, 1) == 1   # NULL supports
    assert lib.dcx_model_create(ctypes.byref(h), 0, None, 1, kp, None, None, 0, 500, 1) == 2  # D unsupported
    assert lib.dcx_model_create(ctypes.byref(h), 0, None, 9, kp, None, None, 0, 12, 1) == 1   # kernel kind
    assert b"kernel_kind" in lib.dcx_last_error()
    assert lib.dcx_score(None, None, 0, None, None) == 1


def test_oracle_is_not_imported_by_the_product():
    """the shipped package must not reach the oracle (or any CPU path) — static check of its sources"""
    pkg = os.path.join(ROOT, "diffco_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.lower().replace("no oracle", ""), f"{f} mentions the oracle"
    code = "import sys; import diffco_amd; assert not any(m.startswith('oracle') for m in sys.modules), 'oracle imported'"
    subprocess.run([sys.executable, "-c", code], check=True, cwd=ROOT)


def test_tree_description_is_validated_before_any_device_work():
    """argument checking of DCX_FK_TREE descriptions happens on the host, ahead of the device: a malformed tree is
    DCX_ERR_INVALID / DCX_ERR_UNSUPPORTED with a message even on a box without a GPU"""
    from diffco_amd import _fkdesc as fd
    from diffco_amd import _lib
    lib = _lib.load()
    ident = fd.IDENTITY_BASE

    def good():
        return fd.tree_desc(2, [dict(joints=[dict(type=fd.DCX_J_REV_Z, q=0, fixed=ident),
                                             dict(type=fd.DCX_J_PRISMATIC, q=1, fixed=ident, axis=(0, 0, 1))])],
                            [(0, 0, (0, 0, 0)), (0, 1, (0.1, 0, 0))])

    def rc_of(desc):
        buf = (ctypes.c_float * 64)()
        p = ctypes.cast(buf, ctypes.c_void_p)
        return lib.dcx_fkine(0, ctypes.byref(desc), p, 1, p, None), lib.dcx_last_error().decode()

    rc, msg = rc_of(good())
    assert rc in (0, 4), msg  # fine, or "no device" on a CPU box — never an argument error
    d = good()
    d.t_type[1] = 9
    assert rc_of(d)[0] == 1 and "joint type" in rc_of(d)[1]
    d = good()
    d.t_q[0] = 5
    assert rc_of(d)[0] == 1 and "t_q" in rc_of(d)[1]
    d = good()
    d.pt_frame[1] = 7
    assert rc_of(d)[0] == 1 and "missing frame" in rc_of(d)[1]
    d = good()
    d.t_n_chains = 17
    assert rc_of(d)[0] == 2  # DCX_ERR_UNSUPPORTED: more chains than DCX_MAX_TREE_CHAINS
    d = good()
    d.t_chain_len[0] = 0
    assert rc_of(d)[0] == 1
    with pytest.raises(ValueError, match="outside"):
        fd.tree_desc(1, [dict(joints=[dict(type=fd.DCX_J_REV_X, q=3, fixed=ident)])], [(0, 0, (0, 0, 0))])
    with pytest.raises(ValueError, match="does not exist"):
        fd.tree_desc(1, [dict(joints=[dict(type=fd.DCX_J_REV_X, q=0, fixed=ident)])], [(0, 2, (0, 0, 0))])


def test_profiled_kernel_names_are_kernels_of_this_library():
    """bench.py's `roofline.traffic` is a constant read from profiles/pmc_<workload>.json (HBM bytes per launch from rocprofv3
    --pmc passes of an earlier run).  It silently goes stale when the kernel behind it changes shape: the
    kernel each of those files names must still be a kernel of the library as built - same template arguments, to the last
    flag - or the counters have to be collected again (tools/gpu_profile.sh) before the number may stay in the line."""
    import glob
    import json
    import re
    import shutil
    import subprocess
    if shutil.which("nm") is None:
        import pytest
        pytest.skip("nm not available")
    syms = subprocess.run(["nm", "-C", "--defined-only", os.path.join(ROOT, "diffco_amd", "libdcx.so")], capture_output=True, text=True,
                          check=True).stdout
    have = {re.sub(r"\s+", "", m.group(1)) for m in re.finditer(r"\bvoid (dcx::\w+<[^>]*>)\(", syms)}
    assert any(k.startswith("dcx::score_kernel<12,1,1,1,1024") for k in have)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "pmc_*.json")))
    assert files
    for f in files:
        name = re.sub(r"\s+", "", json.load(open(f))["kernel"].split("  ")[0].split(" (")[0])
        assert name in have, f"{os.path.basename(f)} names {name}, which this libdcx.so does not contain: re-collect the counters"
    # the matrix-core forms' counters (bench.py `roofline.mfma`, profiles/mfma_*.json) name kernels of the matrix-forms library, which
    # build() makes beside the shipped one (round 3's file named kernels no library contained any more)
    mlib = os.path.join(ROOT, "diffco_amd", "libdcx_matrix.so")
    assert os.path.exists(mlib), "diffco_amd/libdcx_matrix.so is missing: __graft_entry__.build() makes it"
    msyms = subprocess.run(["nm", "-C", "--defined-only", mlib], capture_output=True, text=True, check=True).stdout
    mhave = {re.sub(r"\s+", "", m.group(1)) for m in re.finditer(r"\bvoid (dcx::\w+<[^>]*>)\(", msyms)}
    mfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "mfma_*.json")))
    assert mfiles
    for f in mfiles:
        for key, form in (json.load(open(f)).get("forms") or {}).items():
            name = re.sub(r"\s+", "", re.sub(r"^void ", "", form["kernel"]).split("(")[0])
            assert name in mhave, f"{os.path.basename(f)} [{key}] names {name}, which libdcx_matrix.so does not contain"
            assert name not in have, f"{name} is a matrix-core form: it must not be in the shipped library"
