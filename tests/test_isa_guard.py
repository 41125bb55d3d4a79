"""Code-generation guard (CPU only: hipcc cross-compiles): the hot loop of the fused sweep is hand-scheduled around the
wave's ~100 SGPRs and a 64-VGPR budget, and small source changes have silently cost 30 % before (a scalar branch added to
the loop made the allocator park row SGPRs in VGPR lanes; a second accumulator set spilled the lane index to scratch).
This compiles the headline instantiations in both forms and checks what the profiles rely on:
no v_readlane / v_writelane / scratch traffic inside the sweep loop, the expanded form's instruction count, no scratch
beyond the FK tree's own frame."""
import os
import re
import shutil
import subprocess
from collections import Counter

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "diffco_amd", "csrc")

SRC = """#include "dcx_internal.h"
namespace dcx {
template __global__ void score_kernel<12, KF_POLY1, 1, MODE_GRAD_ROW, 1024, false, true>(const ScoreArgs);
template __global__ void score_kernel<12, KF_POLY1, 1, MODE_GRAD_ROW, 1024, false, false>(const ScoreArgs);
template __global__ void score_kernel<12, KF_RQ2, 5, MODE_GRAD_UP, 1024, false, false>(const ScoreArgs);
template __global__ void score_kernel<6, KF_RQ2, 1, MODE_GRAD_ROW, 1024, false, false>(const ScoreArgs);
}
"""


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    d = tmp_path_factory.mktemp("isa")
    src, out = d / "k.hip", d / "k.s"
    src.write_text(SRC)
    subprocess.run(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I", CSRC, "-S", "--cuda-device-only",
                    str(src), "-o", str(out)], check=True, stderr=subprocess.DEVNULL)
    return out.read_text()


def _kernels(txt):
    for m in re.finditer(r"^(_ZN3dcx12score_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi1024ELb0ELb(\d)ELb0ELb0EEEvNS_9ScoreArgsE):", txt, re.M):
        body = txt[m.end():txt.index(".Lfunc_end", m.end())].split("\n")
        meta = txt[txt.index(".name:           " + m.group(1)):]
        yield dict(D=int(m.group(2)), KF=int(m.group(3)), C=int(m.group(4)), MODE=int(m.group(5)), XF=int(m.group(6)), body=body,
                   scratch_bytes=int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", meta).group(1)),
                   vgpr=int(re.search(r"\.vgpr_count:\s+(\d+)", meta).group(1)))


def _sweep_loop(body):
    """the smallest backward-branch range that loads support rows through the scalar cache and holds at least two
    quarter-rate ops (the two-row pipeline of wide rows; the four-row pipeline has four)"""
    labels = {m.group(1): n for n, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    best = None
    for n, l in enumerate(body):
        m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < n:
            seg = body[labels[m.group(1)]:n + 1]
            quarter = sum(("v_rsq_f32" in x) or ("v_rcp_f32" in x) for x in seg)
            if quarter >= 2 and sum("s_load_dword" in x for x in seg) >= 2 and (best is None or len(seg) < len(best)):
                best = seg
    assert best is not None, "sweep loop not found"
    return Counter(x.split()[0] for x in (y.strip() for y in best) if x and not x.startswith((".", ";")))


def test_sweep_loops_have_no_lane_parking_and_no_scratch(isa):
    seen = 0
    for k in _kernels(isa):
        c = _sweep_loop(k["body"])
        tag = f"D={k['D']} KF={k['KF']} C={k['C']} MODE={k['MODE']} XF={k['XF']}"
        assert c["v_readlane_b32"] + c["v_writelane_b32"] == 0, (tag, "SGPRs parked in VGPR lanes inside the sweep loop")
        assert sum(v for op, v in c.items() if op.startswith("scratch_")) == 0, (tag, "scratch traffic inside the sweep loop")
        assert k["vgpr"] <= 64, (tag, k["vgpr"])          # 8 waves per SIMD at the narrow shapes
        assert k["scratch_bytes"] <= 44, (tag, k["scratch_bytes"])   # the noinline tree-FK frame and nothing else
        seen += 1
    assert seen == 4


def test_expanded_form_instruction_count(isa):
    """four rows per loop iteration: 12 v_pk_fma per row in the expanded form and no packed add on the hot path; the
    direct form carries 6 v_pk_add + 12 v_pk_fma per row"""
    ks = {k["XF"]: _sweep_loop(k["body"]) for k in _kernels(isa) if k["D"] == 12 and k["C"] == 1}
    valu = {xf: sum(v for op, v in c.items() if op.startswith("v_")) for xf, c in ks.items()}
    assert ks[0]["v_pk_add_f32"] == 24 and ks[0]["v_pk_fma_f32"] == 48 and valu[0] <= 100, (ks[0], valu[0])
    # the expanded loop range also holds the flush block (6 v_pk_fma) and may hold one rare correction block
    assert 48 <= ks[1]["v_pk_fma_f32"] <= 70 and ks[1]["v_rsq_f32"] <= 6 and valu[1] <= 135, (ks[1], valu[1])


SRC_R4 = """#include "dcx_internal.h"
namespace dcx {
template __global__ void score_kernel<12, KF_RQ2, 5, MODE_GRAD_ROW, 1024, false, true>(const ScoreArgs);
template __global__ void score_kernel<12, KF_RQ2, 5, MODE_GRAD_ROW, 1024, false, false>(const ScoreArgs);
template __global__ void score_kernel<12, KF_RQ2, 1, MODE_GRAD_ROW, 1024, false, true>(const ScoreArgs);
template __global__ void score_kernel<6, KF_RQ2, 1, MODE_GRAD_ROW, 1024, false, false>(const ScoreArgs);
}
"""


def test_round4_sweeps_rq_folded_expanded_and_the_two_buffer_pipeline(tmp_path):
    """Round 4 (score_kernel.h, "the two-buffer pipeline", sweep_eval): what the counter passes of config #3 / #4 led to,
    held in the generated code.  (a) RQKernel(p = 2) with its constants folded: 15 VALU instructions per pair at D = 6
    (config #4; 17 before; 12 since round 5's pair2), no multiply by gamma left in the loop; (b) config #3's loop (D = 12, C = 5) in the expanded
    form: 22 per pair (28 in round 3's direct form) and in the direct form 26; (c) its pipeline intact: each row's scalar
    loads stand BEFORE the other row's body (two v_rcp per iteration, each preceded by the loads of the row after it), and
    no s_mov copies of row registers, no lane parking"""
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    src, out = tmp_path / "k.hip", tmp_path / "k.s"
    src.write_text(SRC_R4)
    subprocess.run(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I", CSRC, "-S", "--cuda-device-only",
                    str(src), "-o", str(out)], check=True, stderr=subprocess.DEVNULL)
    ks = {(k["D"], k["C"], k["XF"]): k for k in _kernels(out.read_text())}
    assert set(ks) == {(12, 5, 1), (12, 5, 0), (12, 1, 1), (6, 1, 0)}

    def loop(key):
        k = ks[key]
        labels = {m.group(1): n for n, l in enumerate(k["body"]) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
        best = None
        for n, l in enumerate(k["body"]):
            m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < n:
                seg = [x.strip() for x in k["body"][labels[m.group(1)]:n + 1]]
                seg = [x for x in seg if x and not x.startswith((".", ";"))]
                if sum("v_rcp_f32" in x for x in seg) >= 2 and sum("s_load_dword" in x for x in seg) >= 2 and (best is None or len(seg) < len(best)):
                    best = seg
        assert best is not None
        return best

    def per_row(seg):
        rows = sum("v_rcp_f32" in x for x in seg)
        return sum(x.startswith("v_") for x in seg) / rows, sum(x.startswith("s_mov") for x in seg), rows

    # round 5: the two rows of a stage share every packed instruction (pair2): 12 per pair, no add of halves.  Round 6: the rows
    # arrive PAIR-INTERLEAVED, so every operand pair (r0_k, r1_k) is an aligned SGPR pair of the stage's load - the 22 s_mov per
    # four rows that used to build them are gone (one is the loop's own), and with them a third of the loop's scalar instructions
    seg6 = loop((6, 1, 0))
    v, movs, rows = per_row(seg6)
    assert rows == 4 and v <= 12.0 and movs <= 2, (v, movs)
    salu = sum(x.startswith("s_") and not x.startswith(("s_load", "s_nop", "s_waitcnt")) for x in seg6)
    assert salu <= 20, salu          # address arithmetic + loop control: 16 in round 6 (38 with the s_mov)
    assert sum("s_load_dword" in x for x in seg6) <= 4     # the two rows of a stage arrive together
    assert not any(x.startswith(("v_add_f32", "v_mov_b32", "v_readlane", "v_writelane")) for x in seg6)
    v, movs, rows = per_row(loop((12, 1, 1)))
    assert rows == 4 and v <= 19.0, v                     # incl. the flush block's share (direct form: 24)
    for key, vmax in (((12, 5, 1), 25.0), ((12, 5, 0), 26.0)):   # the expanded loop range holds the flush block (7 VALU / 2 rows)
        seg = loop(key)
        v, movs, rows = per_row(seg)
        assert rows == 2 and v <= vmax and movs <= 2, (key, v, movs)
        assert not any("v_readlane" in x or "v_writelane" in x for x in seg)
        # load placement: [loads of row B] ... rcp(A) ... [loads of row A'] ... rcp(B)
        marks = [("L" if "s_load_dword" in x else "R") for x in seg if "s_load_dword" in x or "v_rcp_f32" in x]
        compact = "".join(c for n, c in enumerate(marks) if n == 0 or marks[n - 1] != c)
        assert compact == "LRLR", (key, "".join(marks))
        assert ks[key]["vgpr"] <= 64 and ks[key]["scratch_bytes"] <= 44


def test_handover_publishes_write_through_and_drains_before_the_counter(isa):
    """The in-launch hand-over of a split launch (score_kernel.h) orders the partial-row stores before the arrival
    counter WITHOUT a release fence: it relies on gfx942 / gfx950 lowering agent-scope atomic stores to `global_store sc1`
    (write-through) and on those stores being counted by vmcnt.  Hold the generated code to exactly that: every store of
    the publish is `sc1`, an `s_waitcnt vmcnt(0)` stands between the last of them and every `global_atomic_add` on the
    counter, the owner's re-read uses `sc1` loads, and no L2 write-back / L1 invalidate (`buffer_wbl2` / `buffer_inv`)
    hides in the kernel as a sign that the source went back to fences without this test being revisited."""
    seen = 0
    for k in _kernels(isa):
        body = [l.strip() for l in k["body"]]
        atomics = [n for n, l in enumerate(body) if l.startswith("global_atomic_add")]
        assert atomics, "arrival counter not found"
        for at in atomics:
            back = body[:at]
            # (4-byte stores: the 8-byte `global_store_dwordx2 ... sc1` belong to the owner-polls protocol, checked below)
            # (and not the owner-polls give-up flag: that one is a SYSTEM-scope store, `sc0 sc1`, to host memory)
            st = max(n for n, l in enumerate(back) if l.startswith("global_store_dword ") and " sc1" in l and " sc0" not in l)
            between = back[st + 1:]
            assert any(l.startswith("s_waitcnt") and "vmcnt(0)" in l for l in between), "no vmcnt(0) between the publish and the counter"
            # the publish run in front of the counter is write-through only
            run = []
            for l in reversed(back[:st + 1]):
                if l.startswith("global_store_dword "):
                    if " sc0" not in l:
                        run.append(l)
                elif l.startswith("global_store_dwordx2"):
                    continue
                elif l.startswith(("s_waitcnt", "v_", "s_", ";", "ds_")) or not l:
                    continue
                else:
                    break
            assert run and all(" sc1" in l for l in run), run
        assert any(l.startswith("global_load_dword ") and " sc1" in l for l in body), "the re-read must use agent-scope (sc1) loads"
        # round 4, the owner-polls protocol: value and tag travel as ONE 8-byte agent-scope store / load (single-copy atomic:
        # that is the whole ordering argument), in the same kernels
        assert any(l.startswith("global_store_dwordx2") and " sc1" in l for l in body), "(value, tag) words must be 8-byte sc1 stores"
        assert any(l.startswith("global_load_dwordx2") and " sc1" in l for l in body), "(value, tag) words must be polled with 8-byte sc1 loads"
        # round 5: an owner that gives up waiting tells the host (system-scope store of the flag), and compares the tag with the
        # launch's own number, not with zero
        assert any(l.startswith("global_store_dword ") and " sc0 sc1" in l for l in body), "the give-up flag must be a system-scope store"
        assert not any(l.startswith(("buffer_wbl2", "buffer_inv")) for l in body)
        seen += 1
    assert seen == 4


@pytest.mark.parametrize("src,flags", [("tools/contraction_ubench.hip", []), ("tools/mfma_coissue_ubench.hip", []),
                                       ("tools/mfma_coissue_ubench.hip", ["-DBF16"]), ("tools/lone_wave_ubench.hip", []),
                                       ("tools/tile16_ubench.hip", ["-std=c++17"]),
                                       ("tools/solve_probe.hip", ["-std=c++17", "-DDCX_SOLVE_TS", "-I" + CSRC]),
                                       ("tools/clock_probe.hip", []),
                                       ("tools/sweep_body_ubench.hip", ["-std=c++17", "-I" + CSRC]),
                                       ("tools/sweep_body_ubench.hip", ["-std=c++17", "-DDCX_EXP_NO_LOADS", "-I" + CSRC])])
def test_measurement_tools_still_compile_for_gfx950(tmp_path, src, flags):
    """the micro-benchmarks the profiles cite are part of the evidence: they must keep building (device code only, no GPU)"""
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    out = tmp_path / "t.s"
    subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-S", "--cuda-device-only"] + flags + [os.path.join(ROOT, src), "-o", str(out)],
                   check=True, stderr=subprocess.DEVNULL)
    text = out.read_text()
    if "contraction" in src:   # the three matrix-core variants really issue matrix instructions
        assert text.count("v_mfma_f32_16x16x4_f32") >= 8 and "v_mfma_f32_16x16x32_bf16" in text
    elif "tile16" in src:      # the LDS-operand sweep reads whole float4s; the scalar one goes through the scalar cache
        assert "ds_read_b128" in text and "s_load_dwordx" in text
    elif "solve_probe" in src:  # both workgroup sizes of the solve, with the phase stamps
        assert text.count("lu_solve_kernel") >= 2 and "s_memrealtime" in text
    elif "clock_probe" in src:  # both clocks are read inside the kernel
        assert "s_memtime" in text and "s_memrealtime" in text and "v_pk_fma_f32" in text
    elif "sweep_body" in src:   # the product's own sweep: rows through the scalar cache (or, in the no-load build, almost none)
        n_loads = text.count("s_load_dwordx")
        assert "v_pk_fma_f32" in text and "v_rcp_f32" in text and n_loads > 0
    elif "mfma_coissue" in src:
        assert ("v_mfma_f32_16x16x32_bf16" if flags else "v_mfma_f32_16x16x4_f32") in text and "s_getreg_b32" in text


def test_dense_solve_keeps_its_panel_in_registers(tmp_path):
    """csrc/solve_kernels.hip (dcx_solve), both workgroup sizes: what profiles/r04_solve.txt relies on.  The panel's 64
    doubles per thread are registers (three source forms ended in scratch: a column loop indexes the array dynamically,
    `if (r == rp) v[r]` chains become v[rp], non-inlined phases spill callee-saved registers per call); the pivot search is a
    DPP reduction (no shuffle through LDS in the kernel at all); the pivot row is published by plain ds_write_b64 and read
    back wide; the rank-nb update is fp64 FMAs against LDS reads, never more than its budget of registers."""
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    out = tmp_path / "s.s"
    subprocess.run(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I", CSRC, "-S", "--cuda-device-only",
                    os.path.join(CSRC, "solve_kernels.hip"), "-o", str(out)], check=True, stderr=subprocess.DEVNULL)
    txt = out.read_text()
    kernels = list(re.finditer(r"^(_ZN3dcx\S*nt(256|512)\S*lu_solve_kernel\S*):", txt, re.M))
    assert sorted(m.group(2) for m in kernels) == ["256", "512"]
    for m in kernels:
        body = txt[m.end():txt.index(".Lfunc_end", m.end())]
        meta = txt[txt.index(".name:           " + m.group(1)):]
        scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", meta).group(1))
        vgpr = int(re.search(r"\.vgpr_count:\s+(\d+)", meta).group(1))
        spills = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", meta).group(1))
        nt = int(m.group(2))
        # 256 threads: one wave per SIMD, the whole file; 512: two waves, 256 registers each - a handful of spills tolerated
        assert vgpr <= 512 and scratch <= (0 if nt == 256 else 64) and spills <= (0 if nt == 256 else 12), (nt, vgpr, scratch, spills)
        ops = Counter(l.split()[0] for l in body.split("\n") if l.startswith("\t") and not l.startswith("\t."))
        assert ops["ds_bpermute_b32"] == 0 and ops["ds_permute_b32"] == 0, "a wave reduction went back through the LDS crossbar"
        assert sum(v for k, v in ops.items() if k.endswith("_dpp")) >= 12 * 60, "the pivot search lost its DPP reduction"
        assert ops["ds_max_u64"] == 32 + 16 + 8 + 4, ops["ds_max_u64"]          # one per panel column and width
        assert ops["v_rcp_f64_e32"] >= 60 and ops["v_div_scale_f64"] == 0, "a pivot reciprocal became an IEEE division"
        assert ops["scratch_load_dwordx2"] + ops["scratch_load_dwordx4"] <= (0 if nt == 256 else 16)
        assert ops["v_fma_f64"] + ops["v_fmac_f64_e32"] > 2500                    # the unrolled panel columns and updates


def test_tile_of_16_configurations_sweeps_from_lds_without_scratch(tmp_path):
    """score_kernel<..., QT> (round 4): the rows come from LDS as whole float4s (three ds_read_b128 + the weight per pair at
    D = 12), two row buffers in registers, no scratch and no lane parking inside the sweep loop."""
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    src, out = tmp_path / "q.hip", tmp_path / "q.s"
    src.write_text('#include "dcx_internal.h"\nnamespace dcx {\n'
                   "template __global__ void score_kernel<12, KF_POLY1, 1, MODE_GRAD_ROW, 1024, false, false, false, true>(const ScoreArgs);\n"
                   "template __global__ void score_kernel<12, KF_RQ2, 1, MODE_GRAD_ROW, 1024, false, false, false, true>(const ScoreArgs);\n}\n")
    subprocess.run(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I", CSRC, "-S", "--cuda-device-only",
                    str(src), "-o", str(out)], check=True, stderr=subprocess.DEVNULL)
    txt = out.read_text()
    names = re.findall(r"^(_ZN3dcx12score_kernelILi12ELi\dELi1ELi1ELi1024ELb0ELb0ELb0ELb1EEEvNS_9ScoreArgsE):", txt, re.M)
    assert len(names) == 2
    for name in names:
        start = txt.index(name + ":")
        body = txt[start:txt.index(".Lfunc_end", start)].split("\n")
        meta = txt[txt.index(".name:           " + name):]
        assert int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", meta).group(1)) <= 160   # (the FK tree's own frame)
        assert int(re.search(r"\.vgpr_count:\s+(\d+)", meta).group(1)) <= 128                   # four waves per SIMD
        ops = Counter(l.split()[0] for l in body if l.startswith("\t") and not l.startswith("\t."))
        assert ops["ds_read_b128"] + ops["ds_read2_b64"] >= 6      # two row buffers x three float4s
        # the sweep loop: backward branch range holding the LDS row reads and the kernel function's quarter-rate op
        labels = {m.group(1): n for n, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
        loops = []
        for n, l in enumerate(body):
            m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < n:
                seg = body[labels[m.group(1)]:n + 1]
                if sum("ds_read_b128" in x for x in seg) >= 6 and any(("v_rsq_f32" in x) or ("v_rcp_f32" in x) for x in seg):
                    loops.append(seg)
        assert loops, "no sweep loop over LDS rows found"
        seg = min(loops, key=len)
        assert not any(("scratch_" in x) or ("v_readlane" in x) or ("v_writelane" in x) for x in seg)
