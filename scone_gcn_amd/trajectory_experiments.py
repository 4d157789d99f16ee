"""Host-side mirror of the reference's model functions and experiment driver
(trajectory_analysis/trajectory_experiments.py = TE).

Same names, argument order and error behaviour as the reference's call surface for the hot path:
    scone_func(weights, S_lower, S_upper, Bcond_func, last_node, flow)                    TE:137-152
    ebli_func (weights, S_lower, S_upper, Bcond_func, last_node, flow)                    TE:155-170
    bunch_func(weights, S_00, S_10, S_01, S_11, S_21, S_12, S_22, nbrhoods, last_node, flow)   TE:173-203
    hyperparams(), data_setup(), train_model()                                            TE:78-117, 206-311, 313-510
The functions are batched natively (the reference wraps them in vmap, STM:256): last_node (N,), flow (N, E, 1)
or SparseFlows -> log-probabilities (N, D, 1).  Per-sample arguments (scalar last_node, flow (E, 1)) work too
and return (D, 1).  All math runs in libscone_hip.so on the current CUDA (ROCm) device; nothing falls back to CPU.
"""
import os
import sys

import numpy as np
import torch

from . import ops
from .complex import Bconds, Shift, SimplicialComplex, adopt_bconds, adopt_shift, identity_layout
from .synthetic_data_gen import SparseFlows

MODEL_ACT = {"scone": "tanh", "ebli": "leaky_relu", "bunch": "relu"}


def hyperparams(args=None):
    """Parse `-name value` flags with the reference's names and defaults (TE:78-117)."""
    args = sys.argv if args is None else args
    hp = {'model': 'scone', 'epochs': 1000, 'learning_rate': 0.001, 'weight_decay': 0.00005, 'batch_size': 100,
          'hidden_layers': [(3, 16), (3, 16), (3, 16)], 'describe': 1, 'reverse': 0, 'load_data': 1,
          'load_model': 0, 'markov': 0, 'model_name': 'model', 'regional': 0, 'flip_edges': 0,
          'data_folder_suffix': 'working', 'multi_graph': '', 'holes': 1,
          'skip_mode': 'dense'}          # not in the reference: dense | zeros | field (exact zero-skipping, Scone_GCN)
    for i in range(len(args) - 1):
        if args[i] and args[i][0] == '-':
            name = args[i][1:]
            if name == 'hidden_layers':
                nums = list(map(int, args[i + 1].split("_")))
                hp['hidden_layers'] = [(nums[j], nums[j + 1]) for j in range(0, len(nums), 2)]
            elif name in ['model_name', 'data_folder_suffix', 'multi_graph', 'model', 'skip_mode']:
                hp[name] = str(args[i + 1])
            else:
                try:
                    hp[name] = float(args[i + 1])
                except ValueError:
                    pass
    return hp


HYPERPARAMS = hyperparams([])


# ----------------------------------------------------------------------------------------------
# model functions
# ----------------------------------------------------------------------------------------------

def _prep_batch(last_node, flow):
    single = False
    if isinstance(flow, SparseFlows):
        ln = np.atleast_1d(np.asarray(last_node.cpu() if torch.is_tensor(last_node) else last_node))
        return ln, flow, single
    f = flow if torch.is_tensor(flow) else np.asarray(flow)
    if f.ndim == 2:                                   # per-sample call: flow (E, 1), scalar last_node
        single = True
        f = f[None]
    ln = np.atleast_1d(np.asarray(last_node.cpu() if torch.is_tensor(last_node) else last_node))
    return ln, f, single


def _run_batched(fn_cls, plan, weights, last_nodes, flow, widths, rows_total):
    device = plan.device
    w = ops.as_device_weights(weights, device)
    x, N = ops.flows_to_slabs(flow, plan.layout, device)
    if len(last_nodes) != N:
        raise ValueError("last_node and flow disagree on the number of trajectories")
    last_dev = ops._last_nodes_dev(last_nodes, x.shape[0] * ops.NS, device)
    mb = ops.micro_batch_size(rows_total, widths, N, device=device)
    sl = mb // ops.NS
    outs = []
    for s0 in range(0, x.shape[0], sl):
        outs.append(fn_cls.apply(plan, x[s0:s0 + sl], last_dev[s0 * ops.NS:(s0 + sl) * ops.NS], *w))
    logp = torch.cat(outs) if len(outs) > 1 else outs[0]
    return logp[:N].unsqueeze(-1)


def resolve_operands(model_type, shifts, readout):
    """Native operands for what a caller passed: Shift / Bconds objects go through untouched; dense ndarrays, torch tensors
    or scipy matrices (the reference's L1_lower, L1_upper, S_ab -- TE:240-257) and a plain Bcond_func closure (TE:298-303)
    are wrapped once and cached on the object (small complexes; identity row order)."""
    shifts = list(shifts)
    if all(isinstance(m, Shift) for m in shifts):
        layout = shifts[0].layout
    elif model_type == 'bunch':
        sizes = (shifts[0].shape[0], shifts[3].shape[0], shifts[6].shape[0])          # S_00 (V,V), S_11 (E,E), S_22 (F,F)
        layout = next((m.layout for m in shifts if isinstance(m, Shift)), None) or identity_layout(sizes)
        lv = [(0, 0), (0, 1), (1, 0), (1, 1), (1, 2), (2, 1), (2, 2)]
        shifts = [adopt_shift(m, layout, r, c) for m, (r, c) in zip(shifts, lv)]
    else:
        E = shifts[0].shape[0]
        layout = next((m.layout for m in shifts if isinstance(m, Shift)), None) or identity_layout((1, E, 1))
        shifts = [adopt_shift(m, layout, 1, 1) for m in shifts]
    if model_type == 'bunch':
        return shifts, np.asarray(readout.cpu() if torch.is_tensor(readout) else readout)
    return shifts, adopt_bconds(readout, shifts[0].shape[0], layout)


def _scone_like(weights, S_lower, S_upper, Bcond_func, last_node, flow, act):
    n_layers = (len(weights) - 1) / 3
    assert n_layers % 1 == 0, 'wrong number of weights'                    # TE:141-142 / 159-160
    (S_lower, S_upper), bconds = resolve_operands('scone', [S_lower, S_upper], Bcond_func)
    plan = ops.get_scone_plan(S_lower, S_upper, bconds, act, ops.default_device())
    ln, f, single = _prep_batch(last_node, flow)
    ln = ops.remap_last_nodes(plan, ln)
    widths = [1] + [int(weights[3 * i].shape[1]) for i in range(int(n_layers))]
    out = _run_batched(ops._SconeFn, plan, weights, ln, f, widths, plan.n_edges)
    return out[0] if single else out


def scone_func(weights, S_lower, S_upper, Bcond_func, last_node, flow):
    """Forward pass of the SCoNe model with variable number of layers (TE:137-152); tanh activation."""
    return _scone_like(weights, S_lower, S_upper, Bcond_func, last_node, flow, "tanh")


def ebli_func(weights, S_lower, S_upper, Bcond_func, last_node, flow):
    """Forward pass of the Ebli (SNN) model (TE:155-170); leaky_relu(0.01) activation."""
    return _scone_like(weights, S_lower, S_upper, Bcond_func, last_node, flow, "leaky_relu")


def bunch_func(weights, S_00, S_10, S_01, S_11, S_21, S_12, S_22, nbrhoods, last_node, flow):
    """Forward pass of the Bunch (SCCONV) model (TE:173-203); relu on all three levels."""
    n_layers = (len(weights)) / 7
    assert n_layers % 1 == 0, 'wrong number of weights'                    # TE:177-178
    shifts, nbrhoods = resolve_operands('bunch', [S_00, S_10, S_01, S_11, S_21, S_12, S_22], nbrhoods)
    plan = ops.get_bunch_plan(shifts, nbrhoods, ops.default_device())
    ln, f, single = _prep_batch(last_node, flow)
    widths = [1] + [int(weights[7 * i].shape[1]) for i in range(int(n_layers))]
    out = _run_batched(ops._BunchFn, plan, weights, ln, f, widths, sum(plan.sizes))
    return out[0] if single else out


MODEL_FUNCS = {"scone": scone_func, "ebli": ebli_func, "bunch": bunch_func}


# ----------------------------------------------------------------------------------------------
# data setup (TE:206-311) on an in-memory or on-disk dataset
# ----------------------------------------------------------------------------------------------

def setup_from_complex(sc, model='scone', flip_edges=False, flips=None):
    """shifts + readout operand for a SimplicialComplex (TE:214-219, 240-260, 270-309).  flips: an explicit +-1 vector
    (data_setup draws it from the trainer's global stream like the reference); flip_edges=True draws it under seed 1."""
    if flips is None:
        flips = sc.flip_vector(1) if flip_edges else None
    if model == 'scone':
        shifts = sc.scone_shifts(flips)
    elif model == 'ebli':
        shifts = sc.ebli_shifts(flips)
    elif model == 'bunch':
        shifts = sc.bunch_shifts()
    else:
        raise Exception('invalid model type')                              # TE:260
    readout = sc.nbrhoods if model == 'bunch' else sc.bconds(flips)        # TE:305-309
    return shifts, readout, flips


def apply_flips(flows, flips):
    """X -> X F (TE:292-296)."""
    if flips is None:
        return flows
    if isinstance(flows, SparseFlows):
        return SparseFlows(flows.ptr, flows.idx, (flows.val * flips[flows.idx]).astype(np.float32), flows.n_edges)
    return np.asarray(flows) * np.asarray(flips).reshape(1, -1, 1)


def data_setup(hops=(1,), load=True, folder_suffix='schaub', hp=None):
    """Imports and sets up flow, target, and shift matrices for model training (TE:206-311).  Returns the reference's
    eleven values (TE:311): inputs_all, y_all, train_mask, test_mask, shifts, G_undir, E_lookup, nbrhoods, n_nbrs,
    target_nodes_all, prefixes -- shifts as sparse Shift objects and Bconds_func as a Bconds object (both also answer the
    dense questions the reference asks of them: .shape, @, .toarray(), Bconds_func(n))."""
    from .complex import UndirGraph
    from .dataset_io import load_dataset, generate_dataset, load_prefixes
    from .synthetic_data_gen import flow_to_path
    hp = HYPERPARAMS if hp is None else hp
    if hp['flip_edges']:                                                   # TE:214-219: the GLOBAL stream is reseeded with 1 and
        from . import scone_trajectory_model as stm                        # the flips are drawn from it, so the weights drawn
        stm.reseed(1)                                                      # afterwards (STM:237) continue that stream
    if not load:
        generate_dataset(400, 1000, folder=folder_suffix, holes=bool(hp['holes']))
        raise Exception('Data generation done')                            # TE:225
    inputs_all, y_all, target_nodes_all = [], [], []
    sc = None
    for h in hops:
        folder = 'trajectory_data_' + str(h) + 'hop_' + folder_suffix
        X, (B1, B2), y, train_mask, test_mask, coords, last_nodes, target_nodes = load_dataset(folder)
        if sc is None:
            sc = SimplicialComplex.from_incidence(B1, B2, coords=coords)
        target_nodes_all.append(target_nodes)
        inputs_all.append([None, np.array(last_nodes), X])
        y_all.append(y)
    flips = None
    if hp['flip_edges']:
        from . import scone_trajectory_model as stm
        flips = stm._RNG.choice([1, -1], size=sc.cx.n_edges, replace=True, p=[0.8, 0.2]).astype(np.float64)
    shifts, readout, flips = setup_from_complex(sc, hp['model'], flips=flips)
    for i in range(len(inputs_all)):
        inputs_all[i][-1] = apply_flips(inputs_all[i][-1], flips)
        inputs_all[i][0] = readout
    last_nodes = inputs_all[0][1]
    n_nbrs = sc.n_nbrs(last_nodes)
    edges = [tuple(e) for e in sc.cx.edges.tolist()]
    E_lookup = {e: i for i, e in enumerate(edges)}                          # TE:263-268
    prefixes = load_prefixes('trajectory_data_1hop_' + folder_suffix)      # TE:281-284
    if prefixes is None:
        X0 = inputs_all[0][-1]
        dense = X0.todense() if isinstance(X0, SparseFlows) else np.asarray(X0)
        if flips is not None:
            dense = dense * flips.reshape(1, -1, 1)                         # undo X F: paths live on the unflipped orientation
        prefixes = [flow_to_path(dense[i], edges, last_nodes[i]) for i in range(len(last_nodes))]
    return inputs_all, y_all, train_mask, test_mask, shifts, UndirGraph(sc), E_lookup, sc.nbrhoods, n_nbrs, \
        target_nodes_all, prefixes


def train_model(hp=None):
    """Trains a model to predict the next node in each input path (TE:313-510, Markov block excluded)."""
    from .scone_trajectory_model import Scone_GCN
    hp = hyperparams() if hp is None else hp
    inputs_all, y_all, train_mask, test_mask, shifts, G_undir, E_lookup, nbrhoods, n_nbrs, target_nodes_all, prefixes = \
        data_setup(hops=(1, 2), load=hp['load_data'], folder_suffix=hp['data_folder_suffix'], hp=hp)
    (inputs_1hop, inputs_2hop), (y_1hop, y_2hop) = inputs_all, y_all
    in_axes = tuple(([None] * len(shifts)) + [None, None, 0, 0])          # TE:325
    scone = Scone_GCN(hp['epochs'], hp['learning_rate'], hp['batch_size'], hp['weight_decay'],
                      skip_mode=hp.get('skip_mode', 'dense'))
    if hp['model'] not in MODEL_FUNCS:
        raise Exception('invalid model')                                   # TE:445
    model_func = MODEL_FUNCS[hp['model']]
    scone.setup(model_func, hp['hidden_layers'], shifts, inputs_1hop, y_1hop, in_axes, train_mask,
                model_type=hp['model'])
    if hp['regional']:                                                     # TE:449-453
        train_mask = np.array([1 if i % 3 == 1 else 0 for i in range(len(y_1hop))])
        test_mask = np.array([1 if i % 3 == 2 else 0 for i in range(len(y_1hop))])
    if hp['describe'] == 1:                                                # TE:456-461
        print('Graph nodes: {}, edges: {}, avg degree: {}'.format(len(G_undir.nodes), len(G_undir.edges),
                                                                  np.average([G_undir.degree[node] for node in
                                                                              G_undir.nodes])))
        print('Training paths: {}, Test paths: {}'.format(train_mask.sum(), test_mask.sum()))
        print('Model: {}'.format(hp['model']))
    os.makedirs('models', exist_ok=True)
    path = os.path.join('models', hp['model_name'] + '.npz')
    if hp['load_model']:                                                   # TE:464-476
        scone.load_weights(path)
        if hp['epochs'] != 0:
            scone.train(inputs_1hop, y_1hop, train_mask, test_mask, n_nbrs)
            scone.save_weights(path)
        (train_loss, train_acc), (test_loss, test_acc) = scone.test(inputs_1hop, y_1hop, train_mask, n_nbrs), \
            scone.test(inputs_1hop, y_1hop, test_mask, n_nbrs)
    else:
        train_loss, train_acc, test_loss, test_acc = scone.train(inputs_1hop, y_1hop, train_mask, test_mask, n_nbrs)
        scone.save_weights(path)                                           # TE:482-486
    print('standard test set:')                                            # TE:489-494
    train_2target, test_2target = scone.two_target_accuracy(shifts, inputs_1hop, y_1hop, train_mask, n_nbrs), \
        scone.two_target_accuracy(shifts, inputs_1hop, y_1hop, test_mask, n_nbrs)
    scone.test(inputs_1hop, y_1hop, test_mask, n_nbrs)
    print('2-target accs:', train_2target, test_2target)
    results = {"train_2target": train_2target, "test_2target": test_2target,
               "log_probs": scone._predict(scone.weights, inputs_1hop).cpu().numpy()}      # final predictions, kept for inspection
    if hp['reverse']:                                                      # TE:497-504
        from .dataset_io import load_reverse
        rev_flows_in, rev_targets_1hop, rev_last_nodes = load_reverse('trajectory_data_1hop_' + hp['data_folder_suffix'])
        rev_n_nbrs = np.asarray([len(G_undir[n]) for n in rev_last_nodes])                # TE:501
        print('Reverse experiment:')
        # (as in the reference, the reversed flows are fed as stored -- NOT multiplied by F under -flip_edges, TE:499-504)
        rev_inputs = [inputs_1hop[0], rev_last_nodes, rev_flows_in]
        results["reverse"] = scone.test(rev_inputs, rev_targets_1hop, test_mask, rev_n_nbrs)
        results["reverse_log_probs"] = scone._predict(scone.weights, rev_inputs).cpu().numpy()
    scone.experiment_results = results
    return scone, (train_loss, train_acc, test_loss, test_acc)


if __name__ == '__main__':
    train_model()
