"""GPU parity of scn_small_step (csrc/scn_small.hip): the whole gradient step of a micro-batch on a small complex in one launch --
one workgroup per trajectory, activations resident in LDS through every layer, the readout, the cross-entropy and the backward
(TE:137-152, STM:42-56 on the reference's own sizes, TE:86-90) -- against (a) the CPU oracle and (b) the layer-by-layer kernels
the same trainer runs with ops.SMALL_STEP = False.  Tolerance: north_star's 1e-5 (fp32) for the oracle, 2e-6 of the largest
gradient entry between the two GPU paths (they differ in summation order only).
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import scone_oracle as so

pytestmark = pytest.mark.gpu

TOL = 1e-5


@pytest.fixture(autouse=True)
def _every_size():
    """The trainer takes the one-launch step up to ops.SMALL_STEP_MAX_EDGES edges by default (where it is the faster path); these
    tests run it on |E| = 1001 -- eight row tiles per wave, the largest kernel instance but one."""
    from scone_gcn_amd import ops
    old = ops.SMALL_STEP_MAX_EDGES
    ops.SMALL_STEP_MAX_EDGES = 1 << 30
    yield
    ops.SMALL_STEP_MAX_EDGES = old


@pytest.fixture(scope="module")
def sc1(cfg1):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from scone_gcn_amd.complex import SimplicialComplex
    from scone_gcn_amd.synthetic_data_gen import Complex
    cx = Complex(n_nodes=cfg1["n_nodes"], edges=cfg1["edges"].astype(np.int64), faces=cfg1["faces"].astype(np.int64),
                 coords=cfg1["coords"])
    return SimplicialComplex(cx)


def _maxdiff(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b))) / max(1.0, float(np.max(np.abs(b))))


def _net(sc1, cfg1, model, layers, scale, small, graph=False, flip=False):
    from scone_gcn_amd import ops
    from scone_gcn_amd import scone_trajectory_model as stm
    from scone_gcn_amd import trajectory_experiments as te
    ops.SMALL_STEP = small
    shifts, readout, flips = te.setup_from_complex(sc1, model, flip_edges=flip)
    flows = te.apply_flips(cfg1["flows"], flips)
    inputs = [readout, cfg1["last_nodes"], flows]
    stm.reseed(1030)
    net = stm.Scone_GCN(1, 1e-2, 100, 0.0, verbose=False)
    net.use_graph = graph
    net.setup(te.MODEL_FUNCS[model], layers, shifts, inputs, cfg1["targets"], None, cfg1["train_mask"], model_type=model)
    with torch.no_grad():
        for w in net.weights:
            w.mul_(scale)
    return net, inputs, flips


def _step(net, inputs, y, sel):
    from scone_gcn_amd import ops
    staged = net.stage(inputs, y, sel)
    with ops.KernelTimer() as kt:                                   # (also keeps the graph path out of the way)
        loss = float(net.grad_step_staged(inputs, staged, len(sel), apply=False))
    return loss, net._flat_g.cpu().numpy().copy(), kt.table()


@pytest.mark.parametrize("model,act,n_layers,n_traj", [("scone", "tanh", 3, 27), ("scone", "tanh", 2, 8), ("scone", "relu", 4, 13),
                                                       ("ebli", "leaky_relu", 3, 21), ("scone", "tanh", 3, 1)])
def test_small_step_matches_the_oracle_and_the_layer_by_layer_kernels(cfg1, sc1, model, act, n_layers, n_traj):
    """|E| = 1001 (not a multiple of the 16-row tiles), hidden 16, trajectory counts that leave the last slab partly padding.  The
    ebli case runs the same kernel on the L1 / L1^2 pair: rows of up to several dozen entries."""
    from scone_gcn_amd import ops
    from scone_gcn_amd import trajectory_experiments as te
    layers = [(3, 16)] * n_layers
    sel = np.arange(5, 5 + n_traj)
    scale = 8.0 if model == "scone" else 1.0
    plan = old_act = None
    try:
        res = {}
        for small in (True, False):
            net, inputs, _ = _net(sc1, cfg1, model, layers, scale, small)
            plan = net._plan(inputs)                                # (cached per complex: the activation is put back below)
            old_act = plan.act if old_act is None else old_act
            plan.act = act
            res[small] = _step(net, inputs, cfg1["targets"], sel)
            w = [a.cpu().numpy().astype(np.float64) for a in net.weights]
        used = [k for k in res[True][2] if k.startswith("small_step")]
        assert used and not any(k.startswith("small_step") for k in res[False][2]), (res[True][2].keys(), res[False][2].keys())
        assert not any(k.startswith("conv_") for k in res[True][2])
    finally:
        ops.SMALL_STEP = True
        if plan is not None:
            plan.act = old_act
    (la, ga, _), (lb, gb, _) = res[True], res[False]
    gmax = np.abs(gb).max()
    assert gmax > 1e-4
    assert abs(la - lb) <= 2e-6 * max(1.0, abs(lb)) and np.abs(ga - gb).max() <= 2e-6 * max(gmax, 1.0)
    # the oracle on the same trajectories
    B1, B2 = cfg1["B1"], cfg1["B2"]
    shifts_o = so.scone_shifts(B1, B2, None) if model == "scone" else so.ebli_shifts(B1, B2, None)
    nb, _ = so.neighborhoods(cfg1["edges"], cfg1["n_nodes"])
    Bc = so.make_Bconds(B1, nb, None)
    mask = np.ones(n_traj, int)
    oact = act
    ref_loss, ref_g = so.scone_loss_and_grad(w, shifts_o[0], shifts_o[1], Bc, cfg1["last_nodes"][sel], cfg1["flows"][sel],
                                             cfg1["targets"][sel], mask, 0.0, oact)
    assert abs(la - ref_loss) <= TOL * max(1.0, abs(ref_loss))
    got, off = [], 0
    for g in ref_g:
        got.append(ga[off:off + g.size].reshape(g.shape))
        off += g.size
    for k, (a, b) in enumerate(zip(got, ref_g)):
        assert _maxdiff(a, b) <= TOL, "weight %d" % k


def test_small_step_is_bitwise_reproducible_and_flip_invariant(cfg1, sc1):
    """Two launches on the same buffers give identical bits (fixed reduction order: waves, then trajectories); with tanh the loss
    does not depend on the edge orientation (-flip_edges, TE:214-219, 288-296)."""
    sel = np.arange(100, 164)
    net, inputs, _ = _net(sc1, cfg1, "scone", [(3, 16)] * 3, 8.0, True)
    a = _step(net, inputs, cfg1["targets"], sel)
    b = _step(net, inputs, cfg1["targets"], sel)
    assert a[0] == b[0] and np.array_equal(a[1], b[1])
    netf, inputsf, flips = _net(sc1, cfg1, "scone", [(3, 16)] * 3, 8.0, True, flip=True)
    assert flips is not None and (np.asarray(flips) < 0).any()
    c = _step(netf, inputsf, cfg1["targets"], sel)
    assert abs(a[0] - c[0]) <= 1e-6 * max(1.0, abs(a[0]))
    assert np.abs(np.abs(a[1]) - np.abs(c[1])).max() <= 2e-6 * np.abs(a[1]).max() + 1e-7     # W_0, W_1, W_2 gradients keep their values


def test_two_workgroups_per_trajectory_agree_with_one(cfg1, sc1):
    """scn_small_step's paired form (|E| = 1001, 100 trajectories: the reference's own batch, TE:86-90; 200 workgroups that hand each
    other their rows after every layer) against the same launch with one workgroup per trajectory (scn_small_step_pairing(1)):
    the same loss and gradients up to the order of the weight-gradient sums, and bitwise the same from launch to launch."""
    from scone_gcn_amd import _lib
    lib = _lib.load()
    sel = np.arange(60, 160)
    net, inputs, _ = _net(sc1, cfg1, "scone", [(3, 16)] * 3, 8.0, True)
    try:
        assert lib.scn_small_step_pairing(1) == 0
        one = _step(net, inputs, cfg1["targets"], sel)
        assert lib.scn_small_step_pairing(0) == 0
        two = [_step(net, inputs, cfg1["targets"], sel) for _ in range(3)]
    finally:
        lib.scn_small_step_pairing(0)
    assert any(k.startswith("small_step") for k in one[2]) and any(k.startswith("small_step") for k in two[0][2])
    for t in two[1:]:
        assert t[0] == two[0][0] and np.array_equal(t[1], two[0][1])
    gmax = np.abs(one[1]).max()
    assert gmax > 1e-4 and np.isfinite(two[0][0])
    assert abs(two[0][0] - one[0]) <= 2e-6 * max(1.0, abs(one[0])) and np.abs(two[0][1] - one[1]).max() <= 2e-6 * max(gmax, 1.0)
    assert not np.array_equal(two[0][1], one[1])                     # (2 N partials instead of N: the knob did switch the form)
    assert lib.scn_small_step_pairing(2) != 0


def test_small_step_inside_the_replayed_graph(cfg1, sc1):
    """The graph-replayed optimiser step (Scone_GCN._graph_accumulate) captures the one-launch step: three Adam steps replayed equal
    three eager ones bit for bit, and the weights move."""
    res = {}
    for graph in (False, True):
        net, inputs, _ = _net(sc1, cfg1, "scone", [(3, 16)] * 3, 8.0, True, graph=graph)
        staged = net.stage(inputs, cfg1["targets"], np.arange(40, 104))
        w0 = net._flat_w.cpu().numpy().copy()
        out = []
        for _ in range(3):
            loss = float(net.grad_step_staged(inputs, staged, 64))
            out.append((loss, net._flat_g.cpu().numpy().copy(), net._flat_w.cpu().numpy().copy()))
        assert (len(net._graphs) > 0) == graph
        assert np.abs(out[-1][2] - w0).max() > 1e-3
        res[graph] = out
    for (la, ga, wa), (lb, gb, wb) in zip(res[False], res[True]):
        assert la == lb and np.array_equal(ga, gb) and np.array_equal(wa, wb)


def test_optimiser_step_inside_the_summing_launch_gives_the_bits_of_the_separate_one(cfg1, sc1):
    """scn_small_step_adam (one micro-batch, one rank: the launch that sums the weight gradient applies the ridge + Adam update, the
    step index read from device memory) against the same gradient launches followed by scn_adam_step_dev (net.collective_always
    keeps the update out of the gradient launches): weights, moments, gradients and loss bit for bit over four steps, replayed and
    eager, with the step index moved by the host in between (train() sets the loop index, STM:310)."""
    res = {}
    for graph in (False, True):
        for fused in (True, False):
            net, inputs, _ = _net(sc1, cfg1, "scone", [(3, 16)] * 3, 8.0, True, graph=graph)
            net.collective_always = not fused
            staged = net.stage(inputs, cfg1["targets"], np.arange(10, 110))
            out = []
            for i in range(4):
                if i == 2:
                    net._step = 7                                     # the host moves the index: the device copy has to follow
                loss = float(net.grad_step_staged(inputs, staged, 100))
                out.append((loss, net._flat_g.cpu().numpy().copy(), net._flat_w.cpu().numpy().copy(), net._m.cpu().numpy().copy(),
                            net._v.cpu().numpy().copy(), int(net._step_dev[0].item()), net._step))
            assert out[-1][5] == 9 and out[-1][6] == 9
            assert net._adam_in_graph(True) == fused
            res[(graph, fused)] = out
    ref = res[(False, False)]
    assert np.abs(ref[-1][2] - ref[0][2]).max() > 1e-4
    for key, out in res.items():
        for a, b in zip(out, ref):
            assert a[0] == b[0] and all(np.array_equal(x, y) for x, y in zip(a[1:5], b[1:5])), key


_FIRST_CALL_SCRIPT = r"""
import sys
import numpy as np, torch
sys.path.insert(0, sys.argv[1])
from scone_gcn_amd import ops
from scone_gcn_amd import scone_trajectory_model as stm
from scone_gcn_amd import synthetic_data_gen as g
from scone_gcn_amd import trajectory_experiments as te
from scone_gcn_amd.complex import SimplicialComplex
ops.SMALL_STEP_MAX_EDGES = 1 << 30
cx = g.random_SC_graph(250)
paths = g.generate_random_walks(cx, m=12, seed=3)
flows, choice, last, _, _ = g.path_dataset(cx, paths, seed=4)
outs = []
for captured in (True, False):    # the CAPTURED call comes first: it is the first scn_small_step of this process
    sc = SimplicialComplex(cx)    # new shifts => a plan on NEW handles (scn_conv_create), no scn_small_step on them yet
    shifts, readout, _ = te.setup_from_complex(sc, "scone")
    y = np.zeros((12, sc.max_degree, 1))
    y[np.arange(12), choice, 0] = 1.0
    inputs = [readout, last, flows]
    stm.reseed(1030)
    net = stm.Scone_GCN(1, 1e-2, 12, 0.0, verbose=False)
    net.setup(te.MODEL_FUNCS["scone"], [(3, 16)] * 3, shifts, inputs, y, None, np.ones(12, int), model_type="scone")
    with torch.no_grad():
        for w in net.weights:
            w.mul_(20.0)
    (x, last_d, yt, _), = net.stage(inputs, y, np.arange(12))
    plan = net._plan(inputs)
    loss = torch.zeros(1, dtype=torch.float64, device=x.device)
    torch.cuda.synchronize()
    if captured:
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            assert plan.small_step(x, last_d, yt, -1.0 / 12, net.weights, net._grads, loss, overwrite=True)
        gr.replay()
    else:
        assert plan.small_step(x, last_d, yt, -1.0 / 12, net.weights, net._grads, loss, overwrite=True)
    torch.cuda.synchronize()
    outs.append((float(loss), net._flat_g.cpu().numpy().copy()))
(la, ga), (lb, gb) = outs
assert la == lb and abs(la) > 1e-3, (la, lb)
assert np.array_equal(ga, gb) and float(np.abs(ga).max()) > 1e-4
print("first-call capture ok", la)
"""


def test_first_call_on_a_fresh_handle_can_be_captured_into_a_graph(tmp_path):
    """include/scone_hip.h: no launch allocates, copies or synchronises.  scn_small_step's entry pack (col, val_lower, val_upper per
    entry) is built by scn_conv_create*, so the FIRST scn_small_step of a fresh process on a fresh handle can be captured into a HIP
    graph; the replay equals an eager call on another fresh handle bit for bit."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "first_call.py"
    script.write_text(_FIRST_CALL_SCRIPT)
    r = subprocess.run([sys.executable, str(script), root], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "first-call capture ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_small_step_refuses_what_it_does_not_serve(cfg1, sc1):
    """Hidden 32, a complex beyond the LDS, a neighbourhood bound beyond the item list: scn_small_step_supported says no and the
    entry point returns SCN_ERR_UNSUPPORTED (the trainer then runs the layer-by-layer kernels)."""
    from scone_gcn_amd import _lib, ops
    from scone_gcn_amd import trajectory_experiments as te
    lib = _lib.load()
    shifts, readout, _ = te.setup_from_complex(sc1, "scone")
    plan = ops.get_scone_plan(shifts[0], shifts[1], readout, "tanh", ops.default_device())
    h = plan.conv.handle
    assert lib.scn_small_step_supported(h, 3, 16, plan.max_deg, plan.max_items) == 1
    assert lib.scn_small_step_supported(h, 3, 32, plan.max_deg, plan.max_items) == 0
    assert lib.scn_small_step_supported(h, 1, 16, plan.max_deg, plan.max_items) == 0
    assert lib.scn_small_step_supported(h, 7, 16, plan.max_deg, plan.max_items) == 0
    assert lib.scn_small_step_supported(h, 3, 16, 65, plan.max_items) == 0
    assert lib.scn_small_step_supported(h, 3, 16, plan.max_deg, 513) == 0
    assert lib.scn_small_step_workspace(1001, 0, 3) == 0
    net, inputs, _ = _net(sc1, cfg1, "scone", [(3, 32)] * 3, 8.0, True)
    _, _, table = _step(net, inputs, cfg1["targets"], np.arange(16))
    assert not any(k.startswith("small_step") for k in table) and any(k.startswith("conv_fwd") for k in table)


@pytest.mark.parametrize("n_pts,n_traj", [(130, 37), (250, 12), (80, 5), (440, 9), (200, 11), (330, 100), (250, 160)])
def test_small_step_on_other_complex_sizes_against_the_layer_by_layer_kernels(n_pts, n_traj):
    """The kernel's instances by row tiles per wave: |E| ~ 320 (three, the drifter complex's size), ~ 620 (six), ~ 190 (two for
    some waves, one for others) and ~ 1100 (nine: the largest complex whose two activation buffers fit the LDS).  Above 384 edges
    and up to 128 trajectories two workgroups share a trajectory (alternating blocks of eight row tiles: two blocks each at
    |E| ~ 500, three at ~ 620, four at ~ 820 with a whole batch of 100 -- 200 workgroups resident together --, five at ~ 1100);
    160 trajectories at |E| ~ 620 are past that form's limit and run one workgroup each."""
    from scone_gcn_amd import ops
    from scone_gcn_amd import scone_trajectory_model as stm
    from scone_gcn_amd import synthetic_data_gen as g
    from scone_gcn_amd import trajectory_experiments as te
    from scone_gcn_amd.complex import SimplicialComplex
    cx = g.random_SC_graph(n_pts)
    sc = SimplicialComplex(cx)
    paths = g.generate_random_walks(cx, m=n_traj, seed=3)
    flows, choice, last, _, _ = g.path_dataset(cx, paths, seed=3)
    y = np.zeros((n_traj, sc.max_degree, 1))
    y[np.arange(n_traj), choice, 0] = 1.0
    shifts, readout, _ = te.setup_from_complex(sc, "scone")
    inputs = [readout, last, flows]
    res = {}
    try:
        for small in (True, False):
            ops.SMALL_STEP = small
            stm.reseed(1030)
            net = stm.Scone_GCN(1, 1e-2, n_traj, 0.0, verbose=False)
            net.use_graph = False
            net.setup(te.scone_func, [(3, 16)] * 3, shifts, inputs, y, None, np.ones(n_traj, int), model_type="scone")
            with torch.no_grad():
                for w in net.weights:
                    w.mul_(8.0)
            res[small] = _step(net, inputs, y, np.arange(n_traj))
    finally:
        ops.SMALL_STEP = True
    assert any(k.startswith("small_step") for k in res[True][2]), (cx.n_edges, list(res[True][2]))
    (la, ga, _), (lb, gb, _) = res[True], res[False]
    gmax = np.abs(gb).max()
    assert gmax > 1e-4
    assert abs(la - lb) <= 2e-6 * max(1.0, abs(lb)) and np.abs(ga - gb).max() <= 2e-6 * max(gmax, 1.0)


def test_host_batches_through_the_replayed_graph_take_the_one_launch_step():
    """Scone_GCN.grad_step on a drifter-sized complex (|E| = 319: the default rule applies): host batches of four different sizes go
    through the fixed-address staging buffers (filled by scn_host_stage_batch), the captured graph holds scn_scatter_flows +
    scn_small_step, and four Adam steps end at the weights of the layer-by-layer kernels (summation order apart)."""
    from scone_gcn_amd import ops
    from scone_gcn_amd import scone_trajectory_model as stm
    from scone_gcn_amd import synthetic_data_gen as g
    from scone_gcn_amd import trajectory_experiments as te
    from scone_gcn_amd.complex import SimplicialComplex
    keep = ops.SMALL_STEP_MAX_EDGES
    ops.SMALL_STEP_MAX_EDGES = 960                                  # the shipped rule, whatever the fixture set
    cx = g.random_SC_graph(130)
    sc = SimplicialComplex(cx)
    N = 160
    paths = g.generate_random_walks(cx, m=N, seed=5)
    flows, choice, last, _, _ = g.path_dataset(cx, paths, seed=5)
    y = np.zeros((N, sc.max_degree, 1))
    y[np.arange(N), choice, 0] = 1.0
    shifts, readout, _ = te.setup_from_complex(sc, "scone")
    inputs = [readout, last, flows]
    res = {}
    try:
        for small in (True, False):
            ops.SMALL_STEP = small
            stm.reseed(1030)
            net = stm.Scone_GCN(1, 1e-2, 64, 5e-5, verbose=False)
            net.setup(te.scone_func, [(3, 16)] * 3, shifts, inputs, y, None, np.ones(N, int), model_type="scone")
            with torch.no_grad():
                for w in net.weights:
                    w.mul_(8.0)
            rs = np.random.RandomState(2)
            out = []
            for step in range(4):
                m = np.zeros(N, int)
                m[rs.choice(N, 64 - 9 * step, replace=False)] = 1
                net._step = step
                out.append((float(net.grad_step(inputs, y, m)), net._flat_g.cpu().numpy().copy(), net._flat_w.cpu().numpy().copy()))
            assert len(net._graphs) > 0
            res[small] = out
    finally:
        ops.SMALL_STEP = True
        ops.SMALL_STEP_MAX_EDGES = keep
    for (la, ga, wa), (lb, gb, wb) in zip(res[True], res[False]):
        gmax = np.abs(gb).max()
        assert gmax > 1e-5
        assert abs(la - lb) <= 1e-6 * max(1.0, abs(lb)) and np.abs(ga - gb).max() <= 2e-6 * max(gmax, 1.0) and np.abs(wa - wb).max() <= 5e-6
