import sys, os, ctypes as C
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from diffco_amd import _lib, _ops
from helpers import urdf_robot
import test_gpu_traj as T
lib = _lib.require_gpu()
rob = urdf_robot("urdf_panda")
R, W, S = 6, 30, 400
g = torch.Generator().manual_seed(S)
lim = rob.limits
sup_q = torch.rand((S, rob.dof), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
desc = rob.fk_desc()
sup = _ops.fkine(desc, sup_q.cuda()).reshape(S, -1)
model = _ops.ScoreModel(desc, 1, 1.0, 1.0, sup, (0.02 * torch.randn(S, generator=g)).cuda())
paths = T._random_paths(rob, R, W, seed=R * W)
s0, _ = model.score_grad_raw(paths.reshape(-1, rob.dof).cuda())
opt = _lib.TrajOpts(0.02, 0.9, 0.999, 1e-8, 1, 10, 10, 10, float(s0.median()), 0.3, 1e9, 0.35)
lib.dcx_debug_set(b"nw", 16); lib.dcx_debug_set(b"ys", 1)
for iters in (1, 2, 5):
  for xf in (1, 0):
    lib.dcx_debug_set(b"xf", xf)
    outs = []
    for fused in (0, 1):
        lib.dcx_debug_set(b"traj_fused", fused)
        st, bufs = T._traj_state(model, rob, paths)
        stream = C.c_void_p(torch.cuda.current_stream(model.dev).cuda_stream)
        _lib.check(lib.dcx_traj_adam_run(model._h, C.byref(st), C.byref(opt), 1, iters, stream))
        torch.cuda.synchronize()
        outs.append({k: v.clone() for k, v in bufs.items() if k not in ("col_score", "col_grad", "limits")})
    a, b = outs
    print("iters", iters, "xf", xf, {k: float((a[k].float() - b[k].float()).abs().max()) for k in a})
