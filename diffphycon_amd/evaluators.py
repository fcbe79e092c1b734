"""Post-sampling PDE evaluators with the reference's call signatures, running on libdpc.

`burgers_numeric_solve_free` replaces dataset/apps/generate_burgers.py:207-299 (fp32 explicit Euler; all
`ceil(T/dt)` steps run inside ONE kernel launch, one wavefront per trajectory)."""
import torch

from . import _lib


def burgers_numeric_solve_free(u0, f, visc, T, dt=1e-4, num_t=10, mode=None):
    """u0 [N,s], f [N,num_t,s] (device tensors) -> trajectory [N,num_t+1,s] with u0 prepended."""
    if mode == "const":
        raise ValueError("mode='const' is rejected by the reference as well (generate_burgers.py:226)")
    assert f.size(1) == num_t, "check number of time interval"
    assert u0.size(0) == f.size(0)
    n, s = u0.shape
    u0 = u0.contiguous().float()
    f = f.reshape(n, num_t, s).contiguous().float()
    out = torch.empty(n, num_t + 1, s, device=u0.device, dtype=torch.float32)
    _lib.check(_lib.lib().dpc_burgers_fd(_lib.ptr(u0), _lib.ptr(f), _lib.ptr(out), n, s, num_t, float(visc), float(T),
                                         float(dt), _lib.stream()))
    return out
