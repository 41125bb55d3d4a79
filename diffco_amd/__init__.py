"""diffco_amd — MI355X-native implementation of DiffCo's score(+gradient) hot path.

Same public surface as the reference package for that path (`diffco.kernel`, `diffco.model`,
`diffco.optim`, `diffco.utils`, `DiffCo`, `MultiDiffCo`, `DiffCoBeta`); the arithmetic runs in
hand-written HIP kernels for gfx950 behind the C ABI of include/dcx.h (libdcx.so).
Importing the package needs neither the library nor a GPU; calling a score/FK/kernel op does,
and raises loudly otherwise — there is no CPU fallback.
"""
from . import kernel, model, utils  # noqa: F401
from .kernel_perceptrons import DiffCo  # noqa: F401
from .deprecated import MultiDiffCo, DiffCoBeta  # noqa: F401
from . import deprecated  # noqa: F401
from . import optim  # noqa: F401
from . import sharded  # noqa: F401
from . import traj  # noqa: F401
from . import escape  # noqa: F401
from . import urdf  # noqa: F401
from .urdf import URDFRobotFK, MultiURDFRobotFK  # noqa: F401
from . import collision_checkers  # noqa: F401
from .collision_checkers import RBFDiffCo, ForwardKinematicsDiffCo  # noqa: F401
from .traj import fused_adam_traj_optimize  # noqa: F401

__all__ = ["kernel", "model", "utils", "optim", "sharded", "traj", "escape", "fused_adam_traj_optimize", "urdf", "URDFRobotFK", "MultiURDFRobotFK", "collision_checkers", "RBFDiffCo", "ForwardKinematicsDiffCo", "deprecated", "DiffCo", "MultiDiffCo", "DiffCoBeta"]
