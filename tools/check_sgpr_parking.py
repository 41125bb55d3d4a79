#!/usr/bin/env python3
"""tools/check_sgpr_parking.py — developer check (no GPU needed): compile every width of the fused sweep to ISA and
report the instantiations whose inner sweep loop contains v_readlane / v_writelane (SGPRs parked in VGPR lanes: the
scalar pipeline keeps more rows in flight than the wave has SGPRs) or scratch traffic.

    python tools/check_sgpr_parking.py [widths...]
"""
import os
import re
import subprocess
import sys
from collections import Counter
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "diffco_amd", "csrc")
WIDTHS = [2, 4, 6, 8, 12, 16, 18, 21, 24, 27, 30, 32, 36, 42, 48, 54, 60, 64, 72, 84, 96]


def compile_width(d):
    out = f"/tmp/dcx_check_D{d}.s"
    subprocess.run(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", f"-DDCX_INST_D={d}", "-DDCX_INST_PART=0", "-S",
                    "--cuda-device-only", "-o", out, "score_inst.hip"] + sys.argv[1:0], cwd=CSRC, check=True,
                   stderr=subprocess.DEVNULL)
    return d, out


def loops(body):
    labels = {m.group(1): n for n, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    for n, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < n:
            yield labels[m.group(1)], n


def main():
    widths = [int(a) for a in sys.argv[1:]] or WIDTHS
    bad = 0
    with ThreadPoolExecutor(8) as ex:
        for d, path in ex.map(compile_width, widths):
            txt = open(path).read()
            for m in re.finditer(r"^(_ZN3dcx12score_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELb(\d)ELb\dEEEvNS_9ScoreArgsE):", txt, re.M):
                name, D, KF, CC, MODE = m.group(1), *map(int, m.group(2, 3, 4, 5))
                MF, XF = int(m.group(7)), int(m.group(8))
                body = txt[m.end():txt.index(".Lfunc_end", m.end())].split("\n")
                best = None
                for a, b in loops(body):
                    seg = body[a:b + 1]
                    if sum("s_load_dword" in x for x in seg) >= 2 and any("v_pk_fma" in x or "v_fma" in x or "v_fmac" in x for x in seg):
                        if best is None or len(seg) < len(best):
                            best = seg
                if best is None:
                    continue
                c = Counter(l.split()[0] for l in (x.strip() for x in best) if l and not l.startswith((".", ";")))
                valu = sum(v for k, v in c.items() if k.startswith("v_"))
                parked = c["v_readlane_b32"] + c["v_writelane_b32"]
                scratch = sum(v for k, v in c.items() if k.startswith("scratch_"))
                if parked or scratch:
                    bad += 1
                    print(f"D={D:<3} KF={KF} C={CC} MODE={MODE} MF={MF} XF={XF}: {parked} lane moves, {scratch} scratch ops among {valu} VALU "
                          f"instructions of the sweep loop")
    print(f"{bad} instantiation(s) with SGPR parking or scratch inside the sweep loop")


if __name__ == "__main__":
    main()
