"""Generate tests/golden/phi_*.npz by running the REFERENCE smoke evaluator (vendored phi + evaluate_solver.py) in the
build container, through tools/refshim.py.  Data only: inputs + the reference's outputs.

    python tools/gen_golden_phi.py

Also prints the oracle-vs-reference comparison (the pinned tests in tests/test_oracle_smoke.py repeat it from the files).
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import refshim  # noqa: E402

refshim.install()
import matplotlib  # noqa: E402

matplotlib.use("Agg")
import numpy as np  # noqa: E402

_stub = refshim._stub
if "PIL" not in sys.modules:
    try:
        import PIL  # noqa: F401
    except ImportError:
        p = _stub("PIL")
        p.Image = _stub("PIL.Image")

from dataset.apps import evaluate_solver as E  # noqa: E402
from phi.math.nd import StaggeredGrid  # noqa: E402
from phi.solver.sparse import SparseCGPressureSolver, sparse_pressure_matrix, sparse_cg  # noqa: E402
from oracle import smoke_solver as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print("wrote", path, f"{os.path.getsize(path) / 1024:.0f} KiB")


def synthetic_controls(rng, nt, nx=64, amp=1.5):
    """Smooth-ish random controls with a few large values so that back-traced coordinates leave the grid."""
    c = rng.standard_normal((nt, nx, nx)).astype(np.float32) * amp
    c[:, :3, :] *= 3.0
    c[:, -3:, :] *= 3.0
    return c


def main():
    rng = np.random.default_rng(7)
    sim = E.init_sim_128()
    dom = O.init_sim_128()

    # ---- D2 masks (int, bit-exact)
    lst, concat, set_zero = E.get_bucket_mask()
    save("phi_masks", fluid=sim._fluid_mask[0, ..., 0], active=sim._active_mask[0, ..., 0],
         vmask=sim._velocity_mask.staggered[0], bucket_list=np.stack(lst).astype(np.int8),
         bucket_concat=concat.astype(np.int8), set_zero=set_zero.astype(np.int8))
    assert np.array_equal(dom.fluid, sim._fluid_mask[0, ..., 0]) and np.array_equal(dom.vmask, sim._velocity_mask.staggered[0])

    # ---- D4 pressure matrix + CG iterates
    am = sim._boundary.pad_active(sim._active_mask)
    fm = sim._boundary.pad_fluid(sim._fluid_mask)
    A = sparse_pressure_matrix([127, 127], am, fm)
    Ad = A.toarray() if False else None  # noqa: F841  (16129^2 dense is too large; compare through mat-vecs instead)
    v0 = rng.standard_normal((1, 128, 128, 2)) * 0.7
    vm = sim.with_boundary_conditions(StaggeredGrid(v0.copy()))
    div = vm.divergence()
    probe = rng.standard_normal(127 * 127)
    Ap_ref = A.dot(probe)
    assert np.array_equal(O.apply_A(dom, probe.reshape(127, 127)).ravel(), Ap_ref), "apply_A differs"
    iters = {}
    for k in (1, 2, 3, 10, 50, 500):
        p, it = sparse_cg(div.copy(), A, k, None, 1e-8)
        iters[k] = (p[0, ..., 0].copy(), it)
        po, ito = O.conjugate_gradient(dom, O.divergence(v0[0] * dom.vmask), 1e-8, k)
        print(f"CG max_it={k}: ref iters {it}, oracle iters {ito}, bit-exact {np.array_equal(po, p[0, ..., 0])}, "
              f"max diff {np.abs(po - p[0, ..., 0]).max():.3e}")
    acc = {}
    for a_ in (1e-2, 1e-4):                    # early exit through the max|r| < accuracy test
        p, it = sparse_cg(div.copy(), A, 500, None, a_)
        po, ito = O.conjugate_gradient(dom, O.divergence(v0[0] * dom.vmask), a_, 500)
        print(f"CG accuracy={a_}: ref iters {it}, oracle iters {ito}, bit-exact {np.array_equal(po, p[0, ..., 0])}")
        acc[a_] = (p[0, ..., 0].copy(), it)
    vfree = sim.divergence_free(StaggeredGrid(v0.copy()), solver=SparseCGPressureSolver(), accuracy=1e-8)
    vfree = sim.with_boundary_conditions(vfree).staggered[0]
    vo = O.divergence_free(dom, v0[0]) * dom.vmask
    print("divergence_free bit-exact:", np.array_equal(vo, vfree))
    save("phi_pressure", v0=v0[0], probe=probe, Ap=Ap_ref, div=O.divergence(v0[0] * dom.vmask),
         p1=iters[1][0], p2=iters[2][0], p3=iters[3][0], p10=iters[10][0], p50=iters[50][0], pfinal=iters[500][0],
         iters_final=np.int64(iters[500][1]), vfree=vfree,
         p_acc1e2=acc[1e-2][0], iters_acc1e2=np.int64(acc[1e-2][1]), p_acc1e4=acc[1e-4][0],
         iters_acc1e4=np.int64(acc[1e-4][1]))

    # ---- D3 advect (incl. the (N-1, N] -> 0 quirk)
    dens = np.zeros((127, 127), np.float32)
    dens[100:127, 90:127] = rng.random((27, 37)).astype(np.float32)
    dens[5:40, 0:30] = rng.random((35, 30)).astype(np.float32)
    vbig = rng.standard_normal((1, 128, 128, 2)) * 2.0
    adv = StaggeredGrid(vbig).advect(dens.reshape(1, 127, 127, 1), dt=1)[0, ..., 0]
    ao = O.advect(vbig[0], dens)
    print("advect bit-exact:", np.array_equal(ao, adv), adv.dtype, "zeros from the quirk:",
          int(((adv == 0) & (dens > 0)).sum()))
    save("phi_advect", vel=vbig[0], dens=dens, out=adv)

    # ---- D1 rollout: 8 frames from 4 control frames, 64^2 controls (x2 in time, x2 in space)
    nt, T = 4, 8
    c1, c2 = synthetic_controls(rng, nt), synthetic_controls(rng, nt)
    c1[:, 8:56, 8:56] = 0
    c2[:, 8:56, 8:56] = 0
    d0 = np.zeros((64, 64), np.float32)
    d0[50:60, 20:40] = 1.0            # overlaps the target bucket columns once advected downwards
    d0[12:20, 2:10] = 0.5             # next to a side bucket
    ref = E.solver(sim, E.init_velocity_(), d0.copy(), c1.copy(), c2.copy(), per_timelength=T)
    its = []
    orc = O.solver(dom, O.init_velocity_(), d0.copy(), c1.copy(), c2.copy(), per_timelength=T, info=its)
    names = ["densitys", "zero_densitys", "velocitys", "c1", "c2", "smoke_out"]
    for n_, a, b in zip(names, ref, orc):
        print(f"rollout {n_}: bit-exact {np.array_equal(a, b)} max diff {np.abs(a - b).max():.3e} {a.dtype} {a.shape}")
    print("CG iterations per step:", its, "smoke_out:", ref[5][:, 0, 0])
    save("phi_rollout", c1=c1, c2=c2, d0=d0, per_timelength=np.int64(T), densitys=ref[0].astype(np.float32),
         zero_densitys=ref[1].astype(np.float32), velocitys=ref[2], smoke_out=ref[5][:, 0, 0], cg_iters=np.array(its))
    assert np.array_equal(ref[0].astype(np.float32).astype(np.float64), ref[0])

    # ---- second rollout: weak controls, the CG converges below the 500-iteration cap in the last step
    rng2 = np.random.default_rng(3)
    c1b = (rng2.standard_normal((4, 64, 64)) * 0.05).astype(np.float32)
    c2b = (rng2.standard_normal((4, 64, 64)) * 0.05).astype(np.float32)
    d0b = np.zeros((64, 64), np.float32)
    d0b[50:60, 20:40] = 1
    refb = E.solver(sim, E.init_velocity_(), d0b.copy(), c1b.copy(), c2b.copy(), per_timelength=T)
    itsb = []
    orcb = O.solver(dom, O.init_velocity_(), d0b.copy(), c1b.copy(), c2b.copy(), per_timelength=T, info=itsb)
    print("rollout b bit-exact:", all(np.array_equal(a, b) for a, b in zip(refb, orcb)), "CG iterations:", itsb)
    save("phi_rollout_b", c1=c1b, c2=c2b, d0=d0b, per_timelength=np.int64(T), density_last=refb[0][-1].astype(np.float32),
         zero_density_last=refb[1][-1].astype(np.float32), velocity_last=refb[2][-1], smoke_out=refb[5][:, 0, 0],
         cg_iters=np.array(itsb))

    # np.sum restatement
    for n in (7, 100, 4096, 16129, 16384, 20000):
        a = rng.standard_normal(n) * 10.0 ** rng.integers(-6, 6, n)
        assert O.np_sum(a) == np.sum(a), n
    print("np_sum restatement ok")


if __name__ == "__main__":
    main()
