"""Where a ForwardKinematicsDiffCo.update round spends its time, line by line (developer tool; GPU box).

A small line timer on sys.settrace for the functions named below: the time between two consecutive line events of a traced
frame is charged to the first of the two lines (calls made from the line included).  Usage: python tools/facade_lines.py
"""
import collections
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from helpers import urdf_robot  # noqa: E402
from diffco_amd import collision_checkers, kernel_perceptrons  # noqa: E402
from diffco_amd.collision_checkers import ForwardKinematicsDiffCo  # noqa: E402

TRACED = {kernel_perceptrons.DiffCo.train_perceptron.__code__, kernel_perceptrons.DiffCo.fit_poly.__code__,
          kernel_perceptrons.DiffCo.jump_start_initialize.__code__,
          collision_checkers.RBFDiffCo.fit.__code__, collision_checkers.RBFDiffCo.update.__code__,
          collision_checkers.RBFDiffCo._generate_dataset.__code__, ForwardKinematicsDiffCo._generate_dataset.__code__}
cost = collections.defaultdict(float)
last = {}


def tracer(frame, event, arg):
    if frame.f_code not in TRACED:
        return None

    def local(fr, ev, a):
        now = time.perf_counter()
        key = id(fr)
        if key in last:
            ln, t = last[key]
            cost[(fr.f_code.co_name, ln)] += now - t
        if ev == "return":
            last.pop(key, None)
        else:
            last[key] = (fr.f_lineno, time.perf_counter())
        return local
    last[id(frame)] = (frame.f_lineno, time.perf_counter())
    return local


def main():
    dev = torch.device("cuda", 0)
    urob = urdf_robot("urdf_panda")
    k_tip = urob.unique_position_link_names.index("panda_virtual_ee_link")
    centre = torch.tensor([0.35, 0.0, 0.55], device=dev)

    def ground_truth(qq):
        return ((urob.fkine(qq.to(dev))[:, :, k_tip] - centre).norm(dim=1) < 0.35).float().cpu()
    torch.manual_seed(0)
    fk = ForwardKinematicsDiffCo(robot=urob, gamma=10, gt_check_func=ground_truth)
    fk.fit(num_samples=1500, verify_ratio=0.2, fix_joints=[7], fix_joint_values=[0.04])
    fk.update(num_samples=200, verify=0.2)
    torch.cuda.synchronize()
    rounds = 6
    if os.environ.get("NOGC"):
        import gc
        gc.disable()
    sys.settrace(tracer)
    for _ in range(rounds):
        fk.update(num_samples=200, verify=0.2)
    sys.settrace(None)
    import linecache
    for (fn, ln), t in sorted(cost.items(), key=lambda kv: -kv[1])[:24]:
        code = [c for c in TRACED if c.co_name == fn][0]
        print(f"{t / rounds * 1e3:8.3f} ms  {fn}:{ln}  {linecache.getline(code.co_filename, ln).strip()[:110]}")


if __name__ == "__main__":
    main()
