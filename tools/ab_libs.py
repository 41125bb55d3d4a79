#!/usr/bin/env python3
"""tools/ab_libs.py <lib,lib,...> <workload[:batch],...> [rounds] — developer tool (GPU box): bench.py's kernel time of several
builds of libdcx (names under devlibs/: `v1` = devlibs/libdcx_v1.so; `default` = the shipped library), interleaved over
`rounds` rounds on the same box, one fresh process per measurement.  Extra `KEY=VALUE` arguments go into the environment."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if "=" not in a]
envs = dict(a.split("=", 1) for a in sys.argv[1:] if "=" in a)
libs = args[0].split(",")
wls = args[1].split(",")
rounds = int(args[2]) if len(args) > 2 else 2
res = {}
for r in range(rounds):
    for wl in wls:
        name, _, batch = wl.partition(":")
        for lib in libs:
            env = dict(os.environ, **envs)
            if lib != "default":
                env["DCX_LIB"] = os.path.join(ROOT, "devlibs", f"libdcx_{lib}.so")
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", name, "--no-cpu-baseline", "--no-configs", "--steps", "100"]
            if batch:
                cmd += ["--batch", batch]
            p = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT)
            try:
                d = json.loads(p.stdout.strip().splitlines()[-1])
                us, frac = d["roofline"]["kernel_ms"] * 1e3, d["roofline"]["frac"]
            except Exception:   # noqa: BLE001
                us, frac = float("nan"), float("nan")
                print(p.stderr[-400:], file=sys.stderr)
            res.setdefault((wl, lib), []).append(us)
            print(f"round {r + 1}  {wl:<18} {lib:<10} kernel {us:9.2f} us   frac {frac:.4f}", flush=True)
print()
for wl in wls:
    print(f"{wl:<18} " + "   ".join(f"{lib}: {min(res[(wl, lib)]):8.2f}" for lib in libs) + "   (us, best of rounds)")
