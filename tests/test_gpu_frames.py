"""utils.DH2mat / utils.euler2mat on the HIP path (dcx_dh_frames, dcx_euler_frames; reference utils.py:66-75, 15-38) against
tests/golden/frames.npz: the reference's own fp32 outputs, an fp64 evaluation of the reference functions, and fp64 autograd
vector-Jacobian products (tools/make_golden.py gen_frames)."""
import numpy as np
import pytest
import torch

from helpers import load, relerr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["baxter_left", "panda"])
def test_dh2mat_and_vjp(name):
    from diffco_amd import utils
    d = load("frames")
    q = torch.from_numpy(d[f"{name}_q"])
    par = [torch.from_numpy(d[f"{name}_{k}"]) for k in ("a", "d", "sa", "ca")]
    for dev in ("cpu", "cuda"):       # CPU tensors go to the GPU and come back, like every op of the package
        qq = q.clone().to(dev).requires_grad_(True)
        T = utils.DH2mat(qq, *(p.to(dev) for p in par))
        assert T.shape == (len(q), q.shape[1], 4, 4) and T.device.type == dev and T.dtype == torch.float32
        assert relerr(T.detach().cpu().numpy(), d[f"{name}_T64"]) < 2e-7
        assert relerr(T.detach().cpu().numpy(), d[f"{name}_T32"]) < 3e-7
        # structural entries are exact
        assert float((T[:, :, 3, :3].abs().max() + (T[:, :, 3, 3] - 1).abs().max() + T[:, :, 2, 0].abs().max()).item()) == 0.0
        (gq,) = torch.autograd.grad((T * torch.from_numpy(d[f"{name}_gT"]).to(dev)).sum(), qq)
        assert relerr(gq.cpu().numpy(), d[f"{name}_gq64"]) < 1e-6
    # float64 in, float64 out (computed in fp32 on the device, like the score path)
    T64 = utils.DH2mat(q.double(), *par)
    assert T64.dtype == torch.float64 and relerr(T64.numpy(), d[f"{name}_T64"]) < 2e-7
    # a user-written robot in the reference's style (model.py:225-241): DH2mat -> bmm chain -> masked translations
    from helpers import make_robot
    rob = make_robot(name)
    tfs = utils.DH2mat(q.cuda(), *(p.cuda() for p in par))
    tmp, pts = tfs[:, 0], []
    mask = [True, False, True, False, True, False, True] if name == "baxter_left" else [True, False, True, True, True, False, True]
    for i in range(q.shape[1]):
        if i:
            tmp = torch.bmm(tmp, tfs[:, i])
        if mask[i]:
            pts.append(tmp[:, :3, 3])
    chain = torch.stack(pts, dim=1)
    theta0 = torch.tensor([0, np.pi / 2, 0, 0, 0, 0, 0], dtype=torch.float32) if name == "baxter_left" else torch.zeros(7)
    fk = rob.fkine((q - theta0).cuda())
    n = chain.shape[1]
    assert relerr(chain.cpu().numpy(), fk[:, :n].cpu().numpy()) < 2e-6


def test_euler2mat_and_vjp():
    from diffco_amd import utils
    d = load("frames")
    phi = torch.from_numpy(d["euler_phi"])
    for dev in ("cpu", "cuda"):
        pp = phi.clone().to(dev).requires_grad_(True)
        R = utils.euler2mat(pp)
        assert R.shape == (len(phi), 3, 3) and R.device.type == dev
        assert relerr(R.detach().cpu().numpy(), d["euler_R64"]) < 3e-7 and relerr(R.detach().cpu().numpy(), d["euler_R32"]) < 4e-7
        (gp,) = torch.autograd.grad((R * torch.from_numpy(d["euler_gR"]).to(dev)).sum(), pp)
        assert relerr(gp.cpu().numpy(), d["euler_gphi64"]) < 1e-6
    assert torch.equal(utils.euler2mat(torch.zeros(3).cuda()).cpu(), torch.eye(3)[None])
    # rotations: R R^T = 1
    R = utils.euler2mat(phi.cuda())
    assert float((R @ R.transpose(1, 2) - torch.eye(3, device="cuda")).abs().max()) < 1e-6
    assert utils.euler2mat(phi.reshape(8, 8, 3)).shape == (64, 3, 3)          # the reference's reshape((-1, 3))
