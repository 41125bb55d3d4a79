#!/usr/bin/env python3
"""tools/sweep.py — developer A/B harness for the fused sweep (GPU box only).

Times `dcx_score_grad` for a grid of (library variant, workload, batch, waves-per-block) in ONE process
with interleaved rounds (median and min of HIP-event times), so small deltas are comparable.
Each variant is a separately built libdcx (DCX_LIB), loaded in a fresh subprocess.

    python tools/sweep.py --libs diffco_amd/libdcx.so variants/libdcx_v1.so --batches 65536 1048576 --nw 0 2 4 8
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(args):
    sys.path.insert(0, ROOT)
    import ctypes as Ct
    import statistics

    import torch
    import bench
    from diffco_amd import _lib
    lib = _lib.require_gpu()
    dev = torch.device("cuda", 0)
    out = []
    for wl in args.workloads:
        for B in args.batches:
            w = bench.make_workload(wl, B, dev)
            m, q = w["model"], w["q"]
            score = torch.empty((w["B"], w["C"]), device=dev)
            grad = torch.empty((w["B"], w["dof"]), device=dev)
            st = Ct.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

            def launch():
                cur = Ct.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
                _lib.check(lib.dcx_score_grad(m._h, Ct.c_void_p(q.data_ptr()), w["B"], None,
                                              Ct.c_void_p(score.data_ptr()), Ct.c_void_p(grad.data_ptr()), cur))
            times = {nw: [] for nw in args.nw}
            graphs = {}
            if args.graph:
                # capture `inner` back-to-back launches per geometry into a HIP graph: replay time is GPU time,
                # free of the Python/ctypes launch overhead that dominates below ~50 us per call
                side = torch.cuda.Stream(dev)
                for nw in args.nw:
                    lib.dcx_debug_set(b"nw", nw if nw else -1)
                    with torch.cuda.stream(side):   # warm up ON the capture stream (per-stream scratch, lazy init)
                        launch()
                    torch.cuda.synchronize()
                    gr = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gr, stream=side):
                        st = Ct.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
                        for _ in range(args.inner):
                            launch()
                    graphs[nw] = gr
                st = Ct.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            for rnd in range(args.rounds + 1):
                for nw in args.nw:
                    lib.dcx_debug_set(b"nw", nw if nw else -1)
                    if not args.graph:
                        launch()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    if args.graph:
                        graphs[nw].replay()
                    else:
                        for _ in range(args.inner):
                            launch()
                    e1.record()
                    torch.cuda.synchronize()
                    if rnd:
                        times[nw].append(e0.elapsed_time(e1) / args.inner)
            for nw in args.nw:
                med, mn = statistics.median(times[nw]), min(times[nw])
                F = bench.flops_per_eval(w["D"], w["C"], w["S"])
                out.append(dict(lib=os.path.basename(_lib.LIB_PATH), workload=wl, B=w["B"], nw=nw, ms_med=round(med, 5),
                                ms_min=round(mn, 5), Mevals=round(w["B"] / med / 1e3, 1),
                                tflops=round(F * w["B"] / med / 1e9, 2)))
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", nargs="+", default=[os.path.join(ROOT, "diffco_amd", "libdcx.so")])
    ap.add_argument("--workloads", nargs="+", default=["headline"])
    ap.add_argument("--batches", nargs="+", type=int, default=[65536])
    ap.add_argument("--nw", nargs="+", type=int, default=[0], help="waves per block; 0 = library heuristic")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--inner", type=int, default=20)
    ap.add_argument("--graph", action="store_true", help="time HIP-graph replays (GPU time without host launch overhead)")
    ap.add_argument("--child", action="store_true")
    args = ap.parse_args()
    if args.child:
        return child(args)
    rows = []
    for libp in args.libs:
        env = dict(os.environ, DCX_LIB=os.path.abspath(libp))
        cmd = [sys.executable, os.path.abspath(__file__), "--child", "--workloads", *args.workloads, "--batches",
               *map(str, args.batches), "--nw", *map(str, args.nw), "--rounds", str(args.rounds), "--inner", str(args.inner)] + (["--graph"] if args.graph else [])
        r = subprocess.run(cmd, env=env, capture_output=True, text=True)
        if r.returncode != 0:
            print(f"[{libp}] FAILED\n{r.stderr[-2000:]}")
            continue
        rows += json.loads(r.stdout.strip().splitlines()[-1])
    print(f"{'lib':<18}{'workload':<12}{'B':>9}{'nw':>4}{'ms(med)':>10}{'ms(min)':>10}{'M evals/s':>11}{'TFLOP/s':>9}")
    for r in rows:
        print(f"{r['lib']:<18}{r['workload']:<12}{r['B']:>9}{r['nw']:>4}{r['ms_med']:>10.4f}{r['ms_min']:>10.4f}"
              f"{r['Mevals']:>11.1f}{r['tflops']:>9.2f}")


if __name__ == "__main__":
    main()
