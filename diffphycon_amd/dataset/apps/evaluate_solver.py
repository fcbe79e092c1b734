"""Smoke PDE evaluator with the reference's call surface (dataset/apps/evaluate_solver.py), running on libdpc.

    sim = init_sim_128()
    densitys, zero_densitys, velocitys, c1, c2, smoke_out = solver(sim, init_velocity_(), init_density, c1, c2, 256)

`solver` keeps the reference signature and return tuple (:205-310) for ONE trajectory; `solver_batch` evaluates a
batch of trajectories in one launch (one persistent workgroup per trajectory) and can sub-sample its outputs the way
`multi_evaluate` consumes them (inference/inference_2d_smoke.py:388-390), which is what the inference script uses.
There is no CPU implementation: tensors must live on the GPU (NumPy inputs are uploaded to the current device and
NumPy arrays are returned, as the reference does)."""
import ctypes as C

import numpy as np
import torch

from ... import _lib

N128 = 127
RIM = 16
# (size (y, x), origin (y, x)) — build_obstacles_pi_128, evaluate_solver.py:32-63
OBSTACLES_128 = [
    ((1, 96), (16, 16)),
    ((8, 1), (16, 16)), ((16, 1), (40, 16)), ((40, 1), (72, 16)),
    ((8, 1), (16, 112)), ((16, 1), (40, 112)), ((40, 1), (72, 112)),
    ((1, 8), (112, 16)), ((1, 16), (112, 40)), ((1, 16), (112, 72)), ((1, 8), (112, 104)),
    ((16, 1), (64, 48)), ((16, 1), (96, 48)), ((16, 1), (64, 80)), ((16, 1), (96, 80)),
    ((1, 128 - 40 - 40), (40, 40)),
]
# (y, x, len_y, len_x) — get_bucket_mask, evaluate_solver.py:151-152
BUCKETS_128 = [(112, 24 - 2, 127 - 112, 16 + 4), (112, 56 - 2, 127 - 112, 16 + 4), (112, 88 - 2, 127 - 112, 16 + 4),
               (24 - 2, 0, 16 + 4, 16), (56 - 2, 0, 16 + 4, 16), (24 - 2, 112, 16 + 4, 127 - 112),
               (56 - 2, 112, 16 + 4, 127 - 112)]


class FluidSimulation:
    """The slice of phi.flow.FluidSimulation the evaluator needs: a fully open [n, n] domain with obstacle masks
    (phi/flow.py:47-198).  Masks are int8 [1, n, n, 1] as in phi; `device_masks` uploads them once per device."""

    def __init__(self, shape, buckets=BUCKETS_128, rim=RIM, target_bucket=1):
        assert len(shape) == 2 and shape[0] == shape[1], "square 2-D domains only"
        self._dimensions = list(shape)
        self._fluid_mask = np.ones((1, shape[0], shape[1], 1), np.int8)
        self._active_mask = np.ones((1, shape[0], shape[1], 1), np.int8)
        self.buckets = list(buckets)
        self.rim = rim
        self.target_bucket = target_bucket
        self._dev = {}

    @property
    def dimensions(self):
        return self._dimensions

    def set_obstacle(self, mask_or_size, origin=None):
        """phi/flow.py:171-198 (extent + origin form)."""
        if isinstance(mask_or_size, int):
            mask_or_size = [mask_or_size] * 2
        origin = [0, 0] if origin is None else list(origin)
        sl = (0, slice(origin[0], origin[0] + mask_or_size[0]), slice(origin[1], origin[1] + mask_or_size[1]), 0)
        self._fluid_mask[sl] = 0
        self._active_mask[sl] = 0
        self._dev.clear()

    def device_masks(self, device):
        key = str(device)
        if key not in self._dev:
            self._dev[key] = (torch.from_numpy(self._fluid_mask[0, ..., 0].copy()).to(device),
                              torch.from_numpy(self._active_mask[0, ..., 0].copy()).to(device))
        return self._dev[key]

    def domain_struct(self, device):
        fl, ac = self.device_masks(device)
        d = _lib.SmokeDomain()
        d.n, d.rim, d.n_buckets, d.target_bucket = self._dimensions[0], self.rim, len(self.buckets), self.target_bucket
        for k, r in enumerate(self.buckets):
            for q in range(4):
                d.bucket_rect[k][q] = int(r[q])
        d.fluid_d, d.active_d = fl.data_ptr(), ac.data_ptr()
        return d, (fl, ac)


def build_obstacles_pi_128(sim):
    for size, origin in OBSTACLES_128:
        sim.set_obstacle(size, origin)


def init_sim_128():
    """evaluate_solver.py:94-97."""
    sim = FluidSimulation([N128] * 2)
    build_obstacles_pi_128(sim)
    return sim


def init_velocity_():
    """evaluate_solver.py:103-115: float32 [1,128,128,2], (vx, vy) = (0, 0.8)."""
    v = np.empty((1, 128, 128, 2), np.float32)
    v[..., 0] = 0
    v[..., 1] = 0.8
    return v


def get_bucket_mask():
    """evaluate_solver.py:150-171."""
    lst, concat, set_zero = [], np.zeros((128, 128)), np.ones((128, 128))
    for (y, x, ly, lx) in BUCKETS_128:
        m = np.zeros((128, 128))
        m[y:y + ly, x:x + lx] = 1
        concat[y:y + ly, x:x + lx] = 1
        set_zero[y:y + ly, x:x + lx] = 0
        lst.append(m)
    return lst, concat, set_zero


def _dev_f32(a, device):
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=torch.float32).contiguous()
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)


def solver_batch(sim, init_velocity, init_density, c1, c2, per_timelength, dt=1, accuracy=1e-8, max_iterations=500,
                 frame_stride=1, space_stride=1, density_dtype=torch.float64, want_velocity=True,
                 want_zero_density=True, device=None, return_cg_iterations=False):
    """Batched `solver`: init_density [B,nx,nx], c1/c2 [B,nt,nx,nx], init_velocity [128,128,2] (shared) or
    [B,128,128,2].  Returns device tensors (densitys [B,T',W',W'], zero_densitys, velocitys [B,T',W',W',2],
    smoke_out [B,T']) with T' = ceil(per_timelength/frame_stride), W' = 128/space_stride."""
    if device is None:
        device = c1.device if isinstance(c1, torch.Tensor) and c1.is_cuda else torch.device("cuda", torch.cuda.current_device())
    c1, c2 = _dev_f32(c1, device), _dev_f32(c2, device)
    d0 = _dev_f32(init_density, device)
    B, nt, nx = c1.shape[0], c1.shape[1], c1.shape[2]
    assert d0.shape == (B, nx, nx) and c2.shape == c1.shape and c1.shape[3] == nx
    v0 = _dev_f32(init_velocity, device).reshape(-1, 128, 128, 2)
    assert v0.shape[0] in (1, B)
    vstride = 0 if v0.shape[0] == 1 else 128 * 128 * 2
    T = int(per_timelength)
    To, Wo = -(-T // frame_stride), 128 // space_stride
    dens = torch.empty(B, To, Wo, Wo, device=device, dtype=density_dtype)
    zdens = torch.empty_like(dens) if want_zero_density else None
    vel = torch.empty(B, To, Wo, Wo, 2, device=device, dtype=torch.float64) if want_velocity else None
    smoke = torch.empty(B, To, device=device, dtype=torch.float64)
    iters = torch.zeros(B, max(T - 1, 1), device=device, dtype=torch.int32)
    L = _lib.lib()
    dom, keep = sim.domain_struct(device)
    ws = _lib.workspace(L.dpc_smoke_workspace_bytes(dom.n, B), device)
    out = _lib.SmokeOut()
    out.densitys, out.zero_densitys = dens.data_ptr(), (zdens.data_ptr() if zdens is not None else None)
    out.velocitys, out.smoke_out, out.cg_iters = (vel.data_ptr() if vel is not None else None), smoke.data_ptr(), iters.data_ptr()
    out.density_f32 = 1 if density_dtype == torch.float32 else 0
    out.frame_stride, out.space_stride = frame_stride, space_stride
    _lib.check(L.dpc_smoke_rollout(C.byref(dom), _lib.ptr(v0), vstride, _lib.ptr(d0), _lib.ptr(c1), _lib.ptr(c2), B, nx, nt,
                                   T, float(dt), float(accuracy), int(max_iterations), C.byref(out),
                                   C.c_void_p(ws.data_ptr()), ws.numel(), _lib.stream()))
    res = (dens, zdens, vel, smoke)
    return res + (iters,) if return_cg_iterations else res


def solver(sim, init_velocity, init_density, c1, c2, per_timelength, dt=1):
    """Reference signature and return tuple (evaluate_solver.py:205-310): NumPy in, NumPy out, one trajectory.
    densitys/zero_densitys [T,128,128] f64, velocitys [T,128,128,2] f64, c1/c2 tiled to [T,128,128] (input dtype),
    smoke_out_record [T,128,128] f64."""
    c1 = np.asarray(c1)
    c2 = np.asarray(c2)
    nt, nx = c1.shape[0], c1.shape[1]
    T = per_timelength
    ti, si = int(T / nt), int(128 / nx)
    dens, zdens, vel, smoke = solver_batch(sim, np.asarray(init_velocity).reshape(1, 128, 128, 2),
                                           np.asarray(init_density)[None], c1[None], c2[None], T, dt=dt)
    c1t = np.tile(c1.reshape(nt, 1, nx, 1, nx, 1), (1, ti, 1, si, 1, si)).reshape(T, 128, 128)
    c2t = np.tile(c2.reshape(nt, 1, nx, 1, nx, 1), (1, ti, 1, si, 1, si)).reshape(T, 128, 128)
    rec = np.tile(smoke[0].cpu().numpy()[:, None, None], (1, 128, 128))
    return dens[0].cpu().numpy(), zdens[0].cpu().numpy(), vel[0].cpu().numpy(), c1t, c2t, rec


def pressure_solve(sim, div, accuracy=1e-8, max_iterations=500):
    """SparseCGPressureSolver on div f64 [B,n,n] (device) -> (pressure f64 [B,n,n], iterations int32 [B])
    (phi/solver/sparse.py:88-128, base.py:56-104)."""
    assert div.is_cuda and div.dtype == torch.float64
    p = div.contiguous().clone()
    B = p.shape[0]
    L = _lib.lib()
    dom, keep = sim.domain_struct(p.device)
    ws = _lib.workspace(L.dpc_smoke_workspace_bytes(dom.n, B), p.device)
    its = torch.zeros(B, device=p.device, dtype=torch.int32)
    _lib.check(L.dpc_smoke_pressure_solve(C.byref(dom), C.c_void_p(p.data_ptr()), B, float(accuracy), int(max_iterations),
                                          C.c_void_p(its.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(), _lib.stream()))
    return p, its


def advect(velocity, density, dt=1):
    """StaggeredGrid(velocity).advect(density, dt) for one field: velocity f64 [n+1,n+1,2], density f32 [n,n]."""
    assert velocity.is_cuda and velocity.dtype == torch.float64 and density.dtype == torch.float32
    n = density.shape[-1]
    out = torch.empty_like(density)
    _lib.check(_lib.lib().dpc_smoke_advect(C.c_void_p(velocity.contiguous().data_ptr()), _lib.ptr(density.contiguous()),
                                           _lib.ptr(out), n, float(dt), _lib.stream()))
    return out


def domain_tables(sim, device):
    """Integer tables the kernels derive from the masks: stencil byte [n,n], velocity-mask bits [n+1,n+1],
    bucket id [n+1,n+1] (uint8)."""
    L = _lib.lib()
    dom, keep = sim.domain_struct(device)
    n = dom.n
    cf = torch.zeros(n, n, dtype=torch.uint8, device=device)
    vm = torch.zeros(n + 1, n + 1, dtype=torch.uint8, device=device)
    bk = torch.zeros(n + 1, n + 1, dtype=torch.uint8, device=device)
    ws = _lib.workspace(L.dpc_smoke_workspace_bytes(n, 1), device)
    _lib.check(L.dpc_smoke_domain_tables(C.byref(dom), C.c_void_p(cf.data_ptr()), C.c_void_p(vm.data_ptr()),
                                         C.c_void_p(bk.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(), _lib.stream()))
    return cf, vm, bk
