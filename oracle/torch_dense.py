"""Dense torch-CPU restatement of the reference formulation -- TEST INFRASTRUCTURE ONLY.

Second, independent oracle: the same per-sample functions as the reference (TE:137-203), written with
dense (E, E) shift matrices and torch autograd instead of a hand-derived backward.  Used (a) to cross
check oracle/scone_oracle.py's hand-written gradients, (b) as the "dense-faithful" CPU baseline B1 of
BASELINE.md section 3 (full-N forward, mask afterwards, autograd backward, Adam) in bench.py.

Never imported by scone_gcn_amd/.  "parity unpinned" in the sense of oracle/scone_oracle.py's header.
"""
import torch


def relu(x):
    return torch.clamp_min(x, 0)                                  # TE:124-125


def tanh(x):
    return torch.tanh(x)                                          # TE:130-131


def leaky_relu(x):
    return torch.where(x >= 0, x, 0.01 * x)                       # TE:133-134


def _conv(weights, S_lower, S_upper, flow, act):
    n_layers = (len(weights) - 1) / 3
    assert n_layers % 1 == 0, "wrong number of weights"           # TE:141-142
    cur = flow                                                    # (N, E, C)
    for i in range(int(n_layers)):
        cur = cur @ weights[3 * i] \
            + (S_lower @ cur) @ weights[3 * i + 1] \
            + (S_upper @ cur) @ weights[3 * i + 2]                # TE:145-147 (left-assoc: shift first)
        cur = act(cur)
    return cur


def scone_func(weights, S_lower, S_upper, B1_ext, nbrhoods, last_nodes, flows, act=tanh):
    """Batched scone_func (TE:137-152).  B1_ext = B1 with the zero row appended (TE:288)."""
    cur = _conv(weights, S_lower, S_upper, flows, act)
    Bc = B1_ext[nbrhoods[last_nodes]]                             # (N, D, E)   TE:298-303
    logits = (Bc @ cur) @ weights[-1]                             # TE:151
    return logits - torch.logsumexp(logits, dim=1, keepdim=True)  # TE:152


def ebli_func(weights, S_lower, S_upper, B1_ext, nbrhoods, last_nodes, flows):
    return scone_func(weights, S_lower, S_upper, B1_ext, nbrhoods, last_nodes, flows, act=leaky_relu)


def bunch_func(weights, shifts, nbrhoods, last_nodes, flows):
    """Batched bunch_func (TE:173-203)."""
    S_00, S_10, S_01, S_11, S_21, S_12, S_22 = shifts
    n_layers = len(weights) / 7
    assert n_layers % 1 == 0, "wrong number of weights"
    N = flows.shape[0]
    cur = [flows.new_zeros((N, S_00.shape[1], 1)), flows, flows.new_zeros((N, S_22.shape[1], 1))]
    for i in range(int(n_layers)):
        w = weights[7 * i: 7 * i + 7]
        nxt = [(S_00 @ cur[0]) @ w[0] + (S_10 @ cur[1]) @ w[1],
               (S_01 @ cur[0]) @ w[2] + (S_11 @ cur[1]) @ w[3] + (S_21 @ cur[2]) @ w[4],
               (S_12 @ cur[1]) @ w[5] + (S_22 @ cur[2]) @ w[6]]
        cur = [relu(c) for c in nxt]
    nodes_out = cur[0]                                            # (N, V, 1)
    idx = nbrhoods[last_nodes] % nodes_out.shape[1]               # -1 wraps to the last node (TE:201)
    logits = torch.gather(nodes_out, 1, idx[:, :, None])
    return logits - torch.logsumexp(logits, dim=1, keepdim=True)


def loss_fn(preds, y, mask, weights, weight_decay):
    """STM:42-56: masked cross entropy + ridge over every weight."""
    m = mask.bool()
    ce = -(preds[m] * y[m]).sum() / m.sum()
    return ce + weight_decay * sum((w ** 2).sum() for w in weights)
