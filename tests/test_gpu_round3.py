"""GPU parity tests added in round 3 (all through the C ABI, against the CPU oracle or against properties):

  * the layer-systolic shortwave solver (kernels_rte_sw_sys.hip, the default up to 60 layers) against the two-pass kernel
    and the oracle: layer counts, orientations, ragged tiles, per-band albedos, direct flux, solver switches, both
    arithmetic modes; the tail split (bit-identical); position independence; graph capture without warm-up;
  * host threads sharing the solver scratch pool.

Tolerances as in test_gpu_parity.py: fluxes 1e-9 W m-2 against the oracle (north_star: 1e-6).  The solvers restate
RTE-RRTMGP, which is not in the reference tree: parity unpinned."""
import threading

import numpy as np
import pytest

from test_gpu_round2 import T, FLUX_ATOL

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def default_options(pkg):
    pkg.reset_solver_options()
    pkg.set_solver_option("sw_solver", 0)
    pkg.set_solver_option("sw_tail_split", 1)
    yield
    pkg.reset_solver_options()
    pkg.set_solver_option("sw_solver", 0)
    pkg.set_solver_option("sw_tail_split", 1)
    pkg.set_arithmetic(pkg.FAST)


def sw_inputs(rng, ncol, nlay, ng, nband=2, g_zero=False):
    tau = rng.uniform(0.001, 2.0, (ng, nlay, ncol))
    ssa = rng.uniform(0.0, 0.999, (ng, nlay, ncol))
    g = np.zeros((ng, nlay, ncol)) if g_zero else rng.uniform(0.0, 0.8, (ng, nlay, ncol)) * (rng.uniform(size=(1, 1, ncol)) < 0.5)
    mu0 = rng.uniform(0.05, 1.0, ncol)
    toa = rng.uniform(10, 100, (ng, ncol))
    albd, albf = rng.uniform(0.05, 0.4, (ncol, nband)), rng.uniform(0.05, 0.4, (ncol, nband))
    edges = np.linspace(0, ng, nband + 1).astype(int)
    b2g = np.array([[edges[b] + 1, edges[b + 1]] for b in range(nband)], dtype=np.int32)
    return tau, ssa, g, mu0, toa, albd, albf, b2g


def run_sw(pkg, gpu, inp, top_at_1, with_dir=True):
    import torch
    tau, ssa, g, mu0, toa, albd, albf, b2g = inp
    ng, nlay, ncol = tau.shape
    t = T(gpu)
    op = pkg.OpticalProps2str(); op.tau, op.ssa, op.g = t(tau), t(ssa), t(g)
    op.band2gpt = b2g
    n = 3 if with_dir else 2
    fl = pkg.FluxesBroadband(*(torch.full((nlay + 1, ncol), -1., dtype=torch.float64, device=gpu) for _ in range(n)))
    assert pkg.rte_sw(op, top_at_1, t(mu0), t(toa), t(albd), t(albf), fl) == ""
    torch.cuda.synchronize()
    out = [fl.flux_up.cpu().numpy(), fl.flux_dn.cpu().numpy()]
    if with_dir:
        out.append(fl.flux_dn_dir.cpu().numpy())
    return out


def oracle_sw(oracle_mod, inp, top_at_1, options=None):
    tau, ssa, g, mu0, toa, albd, albf, b2g = inp
    ng = tau.shape[0]
    band = np.zeros(ng, dtype=int)
    for b, (lo, hi) in enumerate(b2g):
        band[lo - 1:hi] = b
    return oracle_mod.rte_sw(tau, ssa, g, mu0, toa, albd[:, band].T.copy(), albf[:, band].T.copy(), top_at_1=top_at_1, options=options)


@pytest.mark.parametrize("ncol,nlay,ng,top_at_1", [
    (1, 60, 27, True), (63, 60, 27, False), (65, 60, 5, True), (1000, 60, 27, True), (20000, 60, 27, False),
    (300, 1, 3, True), (300, 4, 7, False), (300, 5, 2, True), (700, 47, 14, True), (700, 59, 9, False), (130, 6, 1, True),
])
def test_systolic_rte_sw_vs_oracle_and_two_pass(pkg, gpu, oracle_mod, ncol, nlay, ng, top_at_1):
    """The layer-systolic solver against the oracle (1e-9 W m-2) and against the two-pass kernel (same arithmetic per
    (column, g-point); the g-point sums are ordered differently: a few ulp), with the tail split (tail tiles one
    g-point per block, summed in order: the same bits) switched on and off, in both arithmetic modes."""
    rng = np.random.default_rng(1000 * ncol + nlay)
    inp = sw_inputs(rng, ncol, nlay, ng, nband=min(2, ng))
    ref = oracle_sw(oracle_mod, inp, top_at_1)
    scale = max(1.0, float(np.max(ref[1])) / 1000.0)
    for arith in (pkg.FAST, pkg.REFERENCE_ORDER):
        pkg.set_arithmetic(arith)
        res = {}
        for solver, split in ((0, 1), (0, 0), (1, 0)):
            pkg.set_solver_option("sw_solver", solver)
            pkg.set_solver_option("sw_tail_split", split)
            res[solver, split] = run_sw(pkg, gpu, inp, top_at_1)
        for key, out in res.items():
            for a, b in zip(out, ref):
                assert np.max(np.abs(a - b)) < FLUX_ATOL * scale, (key, arith)
        for a, b, c in zip(res[0, 1], res[0, 0], res[1, 0]):
            assert np.array_equal(a, b) and np.allclose(b, c, rtol=1e-13, atol=1e-12)


def test_systolic_rte_sw_columns_do_not_depend_on_their_position(pkg, gpu):
    """A column's fluxes are the same bits wherever it sits in a call (shuffle), whatever the size of the call and whether
    or not it falls into a tail tile; calls repeat bit for bit."""
    rng = np.random.default_rng(5)
    ncol, nlay, ng = 3000, 60, 27
    inp = sw_inputs(rng, ncol, nlay, ng, g_zero=True)
    perm = rng.permutation(ncol)
    shuf = tuple(x[..., perm] if i < 5 else (x[perm] if i < 7 else x) for i, x in enumerate(inp))
    for split in (1, 0):
        pkg.set_solver_option("sw_tail_split", split)
        a = run_sw(pkg, gpu, inp, True)
        b = run_sw(pkg, gpu, inp, True)
        c = run_sw(pkg, gpu, shuf, True)
        for x, y, z in zip(a, b, c):
            assert np.array_equal(x, y) and np.array_equal(x[:, perm], z)
    big = sw_inputs(np.random.default_rng(6), 40000, nlay, ng, g_zero=True)
    whole = run_sw(pkg, gpu, big, True)                          # more tiles than CUs: sequential sum
    part = tuple(x[..., 17000:18000] if i < 5 else (x[17000:18000] if i < 7 else x) for i, x in enumerate(big))
    piece = run_sw(pkg, gpu, part, True)
    for x, y in zip(whole, piece):
        assert np.array_equal(x[:, 17000:18000], y)


def test_systolic_rte_sw_switches_and_no_direct_flux(pkg, gpu, oracle_mod):
    """Version switches (direct-beam clamps, k floor) on the layer-systolic solver against the oracle with the same
    switches; a call without flux_dir gives the same flux_up / flux_dn."""
    rng = np.random.default_rng(21)
    inp = sw_inputs(rng, 500, 60, 9)
    inp[1][:, ::7, :] = 1.0 - 1e-9     # nearly conservative layers: the k floor and the clamps matter
    for clamp, kfl in ((0, 1e-12), (1, 1e-12), (1, 1e-4)):
        pkg.set_solver_option("sw_dir_clamp", clamp)
        pkg.set_solver_option("sw_k_floor", kfl)
        opt = oracle_mod.solver_options()
        opt.sw_dir_clamp, opt.sw_k_floor = clamp, kfl
        ref = oracle_sw(oracle_mod, inp, True, options=opt)
        scale = max(1.0, float(np.max(ref[1])) / 1000.0)
        out = run_sw(pkg, gpu, inp, True)
        for a, b in zip(out, ref):
            assert np.max(np.abs(a - b)) < FLUX_ATOL * scale * 10, (clamp, kfl)
        two = run_sw(pkg, gpu, inp, True, with_dir=False)
        assert np.array_equal(two[0], out[0]) and np.array_equal(two[1], out[1])


def test_systolic_rte_sw_captured_without_warm_up(pkg, gpu):
    """The layer-systolic solver keeps nothing in global scratch: a call is captured in a HIP graph on a stream that
    has never seen it (whole tiles inside the capture -- the partial sums of the optional tail split are not allocated
    there -- and the same bits as the eager call with the split)."""
    import torch
    rng = np.random.default_rng(3)
    inp = sw_inputs(rng, 20000, 60, 9)       # 313 tiles: one round of 256 and 57 tail tiles
    tau, ssa, g, mu0, toa, albd, albf, b2g = inp
    t = T(gpu)
    op = pkg.OpticalProps2str(); op.tau, op.ssa, op.g = t(tau), t(ssa), t(g)
    op.band2gpt = b2g
    args = (t(mu0), t(toa), t(albd), t(albf))
    fl = pkg.FluxesBroadband(*(torch.zeros((61, 20000), dtype=torch.float64, device=gpu) for _ in range(3)))
    assert pkg.rte_sw(op, True, *args, fl) == ""
    torch.cuda.synchronize()
    ref = fl.flux_up.clone()
    fresh = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    fl.flux_up.zero_()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=fresh):
        assert pkg.rte_sw(op, True, *args, fl) == ""
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(fl.flux_up, ref)


def test_host_threads_share_the_solver_scratch(pkg, gpu, oracle_mod):
    """two host threads call rte_sw (two-pass kernel: always needs the scratch ring) through ECCKD_HOST with
    growing column counts, so that each keeps outgrowing the block the other may be about to launch on; every call
    must still return the oracle's fluxes."""
    pkg.set_solver_option("sw_solver", 1)
    sizes = [40, 700, 90, 2500, 300, 6000, 1500, 9000]
    cases = {}
    for n in sizes:
        inp = sw_inputs(np.random.default_rng(n), n, 60, 6, nband=1)
        cases[n] = (inp, oracle_sw(oracle_mod, inp, True))
    errors = []

    def worker(order):
        import numpy as np
        try:
            for n in order:
                inp, ref = cases[n]
                tau, ssa, g, mu0, toa, albd, albf, b2g = inp
                op = pkg.OpticalProps2str(); op.tau, op.ssa, op.g = tau, ssa, g
                op.band2gpt = b2g
                fl = pkg.FluxesBroadband(np.empty((61, n)), np.empty((61, n)), np.empty((61, n)))
                msg = pkg.rte_sw(op, True, mu0, toa, albd, albf, fl, device=0)
                if msg:
                    errors.append(msg)
                elif max(np.max(np.abs(fl.flux_up - ref[0])), np.max(np.abs(fl.flux_dn - ref[1]))) > FLUX_ATOL:
                    errors.append("wrong fluxes for %d columns" % n)
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(sizes * 3,)), threading.Thread(target=worker, args=(sizes[::-1] * 3,))]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:3]
    pkg.release_scratch(0)
