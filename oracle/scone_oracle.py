"""CPU oracle for the SCoNe hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (scone_gcn_amd/) never imports it and has no CPU fallback.

This is a plain NumPy (fp64 by default) restatement of the reference's algorithm for the path
named by BASELINE.json.north_star.  Abbreviations: TE = trajectory_analysis/trajectory_experiments.py,
STM = trajectory_analysis/scone_trajectory_model.py, BMM = trajectory_analysis/bunch_model_matrices.py,
SDG = trajectory_analysis/synthetic_data_gen.py (all under the reference tree).

PARITY PINNING.  The arithmetic of the path lives in JAX (un-vendored, un-pinned; README.md:6) and
the reference holds no tests, golden outputs or stored models, so forward/backward VALUES are
"parity unpinned" by the reference itself.  What IS pinned by fixtures generated
from the reference's own NumPy code (tests/golden/make_golden.py):
  * B1/B2 incidence matrices, flows, targets, last nodes of config 1     (SDG:82-161, 327-373)
  * the 7 Bunch shift matrices                                             (BMM:71-135)
  * the first weights drawn under seed 1030                                (STM:15, 237)
The forward/backward restatement below is pinned instead by: a finite-difference gradient check,
an independent torch-autograd dense restatement (oracle/torch_dense.py), dense == CSR equivalence,
and the invariances SURVEY.md section 4 lists (tests/test_oracle.py).

Everything here is batched over trajectories exactly the way `vmap(model, in_axes)` batches the
per-sample function (STM:256, TE:325): shifts / weights / Bconds broadcast, last_node and flow mapped.
"""
import numpy as np

# ----------------------------------------------------------------------------------------------
# activations (TE:124-134)
# ----------------------------------------------------------------------------------------------

def relu(x):
    return np.maximum(x, 0)                       # TE:124-125


def sigmoid(x):
    return 1 / (1 + np.exp(-x))                   # TE:127-128 (unused by the models)


def tanh(x):
    return np.tanh(x)                             # TE:130-131


def leaky_relu(x):
    return np.where(x >= 0, x, 0.01 * x)          # TE:133-134


ACTIVATIONS = {"tanh": tanh, "relu": relu, "leaky_relu": leaky_relu}


def act_grad(name, z):
    """d act / d z evaluated at the pre-activation z.

    relu: jax.numpy.maximum splits the gradient 0.5/0.5 at an exact tie (lax.max JVP); rows that are
    exactly zero only occur where every contributing input is zero, so the tie value never reaches a
    weight gradient (zero-support propagation) -- kept for fidelity.
    leaky_relu is a `where` select: slope 1 at z == 0.
    """
    if name == "tanh":
        return 1.0 - np.tanh(z) ** 2
    if name == "relu":
        return np.where(z > 0, 1.0, np.where(z == 0, 0.5, 0.0))
    if name == "leaky_relu":
        return np.where(z >= 0, 1.0, 0.01)
    raise ValueError(name)


def logsumexp(a, axis):
    m = np.max(a, axis=axis, keepdims=True)
    return m + np.log(np.sum(np.exp(a - m), axis=axis, keepdims=True))


# ----------------------------------------------------------------------------------------------
# operators (TE:214-219, 240-257; BMM:44-135), dense like the reference
# ----------------------------------------------------------------------------------------------

def hodge_laplacians(B1, B2):
    """L1_lower = B1^T B1, L1_upper = B2 B2^T  (TE:240-241)."""
    return B1.T @ B1, B2 @ B2.T


def flip_matrix(n_edges):
    """F = diag(flips), flips ~ choice([1,-1], p=[.8,.2]) under seed 1  (TE:216-219)."""
    rs = np.random.RandomState(1)
    flips = rs.choice([1, -1], size=n_edges, replace=True, p=[0.8, 0.2])
    return np.diag(flips).astype(np.float64)


def scone_shifts(B1, B2, F=None):
    L_lo, L_up = hodge_laplacians(B1, B2)
    if F is not None:                              # TE:242-244
        L_lo, L_up = F @ L_lo @ F, F @ L_up @ F
    return [L_lo, L_up]                            # TE:247-248


def ebli_shifts(B1, B2, F=None):
    L_lo, L_up = scone_shifts(B1, B2, F)
    L1 = L_lo + L_up
    return [L1, L1 @ L1]                           # TE:251-253


def bunch_shifts(B1, B2):
    """Restatement of compute_shift_matrices (BMM:118-135) and compute_bunch_matrices (BMM:71-116).

    The reference inverts DIAGONAL matrices with dense inv/pinv; here the same diagonals are inverted
    element-wise (pinv of a diagonal = reciprocal where non-zero).  Pinned entry-wise against the
    reference's own output in tests/golden/cfg1_bunch.npz.
    """
    absB1, absB2 = np.abs(B1), np.abs(B2)
    d2_2 = np.maximum(absB2.sum(axis=1), 1)                 # compute_D2(B2)  BMM:44-51, 79   (E)
    d2_1 = np.maximum(absB1.sum(axis=1), 1)                 # compute_D2(B1)  BMM:80          (V)
    d1 = 2 * (absB1 * d2_2[None, :]).sum(axis=1)            # compute_D1      BMM:62-69, 82   (V)
    d5 = absB2.sum(axis=1)                                  # compute_D5      BMM:53-60, 85   (E)
    nF = B2.shape[1]
    d3 = np.full(nF, 1.0 / 3.0)                             # BMM:83
    d4 = np.ones(nF)                                        # BMM:84

    def pinv_diag(d):
        out = np.zeros_like(d, dtype=np.float64)
        nz = d != 0
        out[nz] = 1.0 / d[nz]
        return out

    d1_p, d5_p, d2_2_i, d2_1_i = pinv_diag(d1), pinv_diag(d5), 1.0 / d2_2, 1.0 / d2_1
    D = np.diag
    L0u = B1 @ B1.T @ D(d2_1_i)                             # BMM:92 (D3_n = I)
    L1u = D(d2_2) @ B1.T @ D(d1_p) @ B1                     # BMM:93
    L1d = B2 @ D(d3) @ B2.T @ D(d2_2_i)                     # BMM:94
    L2d = D(d4) @ B2.T @ D(d5_p) @ B2                       # BMM:95
    A0u = D(d2_1) - L0u @ D(d2_1)                           # BMM:100
    A1u = D(d2_2) - L1u @ D(d2_2)                           # BMM:101
    A1d = D(d2_2_i) - D(d2_2_i) @ L1d                       # BMM:102
    A2d = D(1.0 / d4) - D(1.0 / d4) @ L2d                   # BMM:103
    I = np.identity
    A0u_n = (A0u + I(len(d2_1))) @ D(1.0 / (d2_1 + 1))      # BMM:111
    A1u_n = (A1u + I(len(d2_2))) @ D(1.0 / (d2_2 + 1))      # BMM:112
    A1d_n = D(d2_2 + 1) @ (A1d + I(len(d2_2)))              # BMM:113
    A2d_n = D(d4 + 1) @ (A2d + I(nF))                       # BMM:114
    S_00 = A0u_n                                            # BMM:125
    S_10 = D(d1_p) @ B1                                     # BMM:126
    S_01 = D(d2_2) @ B1.T @ D(d1_p)                         # BMM:128
    S_11 = A1d_n + A1u_n                                    # BMM:129
    S_21 = B2 @ D(d3)                                       # BMM:130
    S_12 = D(d4) @ B2.T @ D(d5_p)                           # BMM:132
    S_22 = A2d_n                                            # BMM:133
    return [S_00, S_10, S_01, S_11, S_21, S_12, S_22]


# ----------------------------------------------------------------------------------------------
# readout inputs (TE:270-303)
# ----------------------------------------------------------------------------------------------

def neighborhoods(edges, n_nodes):
    """nbrhoods (V, D) sorted neighbours padded with -1 (TE:273-279); also max degree."""
    adj = [[] for _ in range(n_nodes)]
    for a, b in np.asarray(edges):
        adj[int(a)].append(int(b))
        adj[int(b)].append(int(a))
    D = max(len(a) for a in adj)
    tab = -np.ones((n_nodes, D), dtype=np.int64)
    for v, a in enumerate(adj):
        a = sorted(a)
        tab[v, :len(a)] = a
    return tab, D


def make_Bconds(B1, nbrhoods, F=None):
    """B1_jax = B1 with a zero row appended so that index -1 selects zeros (TE:288); optional flip (TE:291)."""
    B1_ext = np.concatenate([B1, np.zeros((1, B1.shape[1]))], axis=0)
    if F is not None:
        B1_ext = B1_ext @ F

    def Bconds_func(n):                            # TE:298-303
        return B1_ext[nbrhoods[n]]
    return Bconds_func


# ----------------------------------------------------------------------------------------------
# weights (STM:215-242) and Adam (jax.experimental.optimizers.adam as used at STM:300)
# ----------------------------------------------------------------------------------------------

def weight_shapes(in_channels, hidden_layers, out_channels, model_type="scone"):
    shapes = []
    shapes += [(in_channels, hidden_layers[0][1])] * hidden_layers[0][0]          # STM:224
    for i in range(len(hidden_layers) - 1):                                       # STM:226-228
        shapes += [(hidden_layers[i][1], hidden_layers[i + 1][1])] * hidden_layers[i + 1][0]
    if model_type == "bunch":                                                     # STM:230-233
        shapes += [(hidden_layers[-1][1], out_channels)] * hidden_layers[-1][0]
    else:
        shapes += [(hidden_layers[-1][1], out_channels)]
    return shapes


def generate_weights(in_channels, hidden_layers, out_channels, model_type="scone", seed=1030):
    """0.01 * randn in list order under the module-level seed 1030 (STM:15, 235-237)."""
    rs = np.random.RandomState(seed)
    return [0.01 * rs.randn(*s) for s in weight_shapes(in_channels, hidden_layers, out_channels, model_type)]


class Adam:
    """m,v EMA, bias correction with (i+1), eps outside the sqrt: the optimizer STM:300-326 drives."""

    def __init__(self, weights, step_size, b1=0.9, b2=0.999, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps = step_size, b1, b2, eps
        self.x = [np.array(w, dtype=np.float64) for w in weights]
        self.m = [np.zeros_like(w) for w in self.x]
        self.v = [np.zeros_like(w) for w in self.x]

    def update(self, i, grads):
        for k, g in enumerate(grads):
            self.m[k] = (1 - self.b1) * g + self.b1 * self.m[k]
            self.v[k] = (1 - self.b2) * g * g + self.b2 * self.v[k]
            mhat = self.m[k] / (1 - self.b1 ** (i + 1))
            vhat = self.v[k] / (1 - self.b2 ** (i + 1))
            self.x[k] = self.x[k] - self.lr * mhat / (np.sqrt(vhat) + self.eps)
        return self.x


# ----------------------------------------------------------------------------------------------
# forward: scone / ebli (TE:137-170), batched.  Returns log-probabilities (N, D, 1).
# ----------------------------------------------------------------------------------------------

def _apply_shift(S, H):
    """S @ H for every sample: S (R, E) dense ndarray or scipy sparse; H (N, E, C) -> (N, R, C)."""
    # dense or scipy sparse alike: fold the batch into columns, ONE (R, E) x (E, N*C) product (BLAS / CSR SpMM) --
    # the sums of `S @ H[n]` per sample (TE:146-147) in fp64; an un-optimised einsum spent minutes of the GPU suite here
    N, E, C = H.shape
    Y = S @ H.transpose(1, 0, 2).reshape(E, N * C)
    return np.ascontiguousarray(np.asarray(Y).reshape(S.shape[0], N, C).transpose(1, 0, 2))


def conv_forward(weights, S_lower, S_upper, flow, act="tanh", keep=False):
    """The layer loop of scone_func / ebli_func (TE:143-149 / 161-167): SpMM first, then the dense product."""
    n_layers = (len(weights) - 1) / 3
    assert n_layers % 1 == 0, "wrong number of weights"                           # TE:141-142
    f = ACTIVATIONS[act]
    cur = flow
    saved = []
    for i in range(int(n_layers)):
        lo, up = _apply_shift(S_lower, cur), _apply_shift(S_upper, cur)
        z = cur @ weights[3 * i] + lo @ weights[3 * i + 1] + up @ weights[3 * i + 2]
        if keep:
            saved.append((cur, lo, up, z))
        cur = f(z)
    return (cur, saved) if keep else cur


def readout_scone(H, W_last, Bconds, last_nodes):
    """logits = Bcond(last) @ H @ W_last ; out = logits - logsumexp(logits) over ALL D rows (TE:151-152)."""
    N = H.shape[0]
    Bc = np.stack([Bconds(int(last_nodes[n])) for n in range(N)])                 # (N, D, E)
    logits = np.einsum("nde,nec->ndc", Bc, H) @ W_last                            # (N, D, 1)
    return logits - logsumexp(logits, axis=1), Bc, logits


def scone_forward(weights, S_lower, S_upper, Bconds, last_nodes, flows, act="tanh"):
    H = conv_forward(weights, S_lower, S_upper, flows, act)
    return readout_scone(H, weights[-1], Bconds, last_nodes)[0]


def ebli_forward(weights, S_lower, S_upper, Bconds, last_nodes, flows):
    return scone_forward(weights, S_lower, S_upper, Bconds, last_nodes, flows, act="leaky_relu")   # TE:167


# ----------------------------------------------------------------------------------------------
# forward: bunch (TE:173-203)
# ----------------------------------------------------------------------------------------------

BUNCH_SRC = [0, 1, 0, 1, 2, 1, 2]      # input level of weight slot k   (TE:184-192)
BUNCH_DST = [0, 0, 1, 1, 1, 2, 2]      # output level of weight slot k


def bunch_conv_forward(weights, shifts, flow, keep=False):
    n_layers = len(weights) / 7
    assert n_layers % 1 == 0, "wrong number of weights"                           # TE:177-178
    N = flow.shape[0]
    S_00, S_22 = shifts[0], shifts[6]
    cur = [np.zeros((N, S_00.shape[1], 1)), flow, np.zeros((N, S_22.shape[1], 1))]  # TE:179
    saved = []
    for i in range(int(n_layers)):
        g = [_apply_shift(shifts[k], cur[BUNCH_SRC[k]]) for k in range(7)]
        z = [None, None, None]
        for k in range(7):
            t = g[k] @ weights[7 * i + k]
            z[BUNCH_DST[k]] = t if z[BUNCH_DST[k]] is None else z[BUNCH_DST[k]] + t
        if keep:
            saved.append((cur, g, z))
        cur = [relu(c) for c in z]                                                # TE:195
    return (cur, saved) if keep else cur


def bunch_forward(weights, shifts, nbrhoods, last_nodes, flows):
    cur = bunch_conv_forward(weights, shifts, flows)
    nodes_out = cur[0]                                                            # (N, V, 1)
    idx = np.asarray(nbrhoods)[np.asarray(last_nodes)]                            # (N, D), -1 wraps (TE:201)
    logits = np.take_along_axis(nodes_out, (idx % nodes_out.shape[1])[:, :, None], axis=1)
    return logits - logsumexp(logits, axis=1)


# ----------------------------------------------------------------------------------------------
# loss (STM:42-56), accuracy (STM:59-71)
# ----------------------------------------------------------------------------------------------

def ridge(weights):
    """||W[:k]||^2 + ||W[k:-1]||^2 + ||W[-1]||^2 of stacked lists = plain sum of squares of every weight."""
    return sum(float(np.sum(np.asarray(w) ** 2)) for w in weights)


def loss_from_preds(preds, y, mask, weights, weight_decay):
    m = np.asarray(mask).astype(bool)
    return -np.sum(preds[m] * y[m]) / np.sum(m) + weight_decay * ridge(weights)   # STM:54 / 56


def accuracy_from_preds(preds, y, mask, n_nbrs):
    m = np.asarray(mask).astype(bool)
    p = np.array(preds, dtype=np.float64)
    for i in range(len(p)):
        p[i, n_nbrs[i]:] = -100                                                   # STM:67-68
    return float(np.mean(np.argmax(p[m], axis=1) == np.argmax(y[m], axis=1)))     # STM:63, 70-71


def two_target_accuracy_from_preds(preds, y, mask, n_nbrs, rs, random_targets=None):
    """STM:73-108 on given log-probabilities; rs = the global NumPy stream (onp.random), random_targets = the cached draw
    (self.random_targets) or None.  Returns (accuracy, random_targets).  Quirks kept: the redraw compares against the
    PREDICTED choice, and pred_choice (masked, a jax array) is indexed by the unmasked i -- jax clamps an out-of-range
    index to the last element.  (A node with a single neighbour whose prediction is slot 0 would loop forever in the
    reference; such rows keep their draw here.)"""
    N = len(preds)
    n_nbrs = np.asarray(n_nbrs)
    if random_targets is None:
        random_targets = rs.randint(0, high=n_nbrs, size=N)                       # STM:79
    p = np.array(preds, dtype=np.float64)
    for i in range(N):
        p[i, n_nbrs[i]:] = -100                                                   # STM:84-85
    m = np.asarray(mask) == 1
    pred_choice = np.argmax(p[m], axis=1).reshape(-1)                             # STM:87
    for i in range(N):                                                            # STM:89-91
        pc = pred_choice[min(i, len(pred_choice) - 1)]
        while n_nbrs[i] > 1 and random_targets[i] == pc:
            random_targets[i] = rs.randint(0, high=n_nbrs[i])
    rows = np.arange(N)
    random_probs = p[rows, random_targets, 0]                                     # STM:94-95
    true_probs = p[rows, np.argmax(y, axis=1).reshape(N), 0]                      # STM:97-98
    t, r = true_probs[m], random_probs[m]
    return float((np.sum(t > r) + 0.5 * np.sum(t == r)) / np.sum(m)), random_targets   # STM:101-108


# ----------------------------------------------------------------------------------------------
# hand-derived backward of loss(weights) for scone / ebli  (what grad(self.loss) computes, STM:307)
# ----------------------------------------------------------------------------------------------

def scone_loss_and_grad(weights, S_lower, S_upper, Bconds, last_nodes, flows, y, mask, weight_decay,
                        act="tanh"):
    """Returns (loss, [dL/dW_k]).  Operators need not be symmetric (transposes are explicit)."""
    H, saved = conv_forward(weights, S_lower, S_upper, flows, act, keep=True)
    out, Bc, logits = readout_scone(H, weights[-1], Bconds, last_nodes)
    m = np.asarray(mask).astype(bool)
    loss = loss_from_preds(out, y, m, weights, weight_decay)

    d_out = np.where(m[:, None, None], -y / np.sum(m), 0.0)                       # (N, D, 1)
    soft = np.exp(out)
    d_logits = d_out - soft * np.sum(d_out, axis=1, keepdims=True)
    BH = np.einsum("nde,nec->ndc", Bc, H)                                         # (N, D, C)
    grads = [None] * len(weights)
    grads[-1] = np.einsum("ndc,ndo->co", BH, d_logits)
    dH = np.einsum("nde,ndo->neo", Bc, d_logits) @ weights[-1].T                  # (N, E, C)

    S_lo_T = S_lower.T
    S_up_T = S_upper.T
    for i in reversed(range(len(saved))):
        cur, lo, up, z = saved[i]
        dz = dH * act_grad(act, z)
        grads[3 * i] = np.einsum("nec,neo->co", cur, dz)
        grads[3 * i + 1] = np.einsum("nec,neo->co", lo, dz)
        grads[3 * i + 2] = np.einsum("nec,neo->co", up, dz)
        if i > 0:
            dH = (dz @ weights[3 * i].T
                  + _apply_shift(S_lo_T, dz @ weights[3 * i + 1].T)
                  + _apply_shift(S_up_T, dz @ weights[3 * i + 2].T))
    for k in range(len(weights)):
        grads[k] = grads[k] + 2 * weight_decay * weights[k]
    return loss, grads


def bunch_loss_and_grad(weights, shifts, nbrhoods, last_nodes, flows, y, mask, weight_decay):
    cur, saved = bunch_conv_forward(weights, shifts, flows, keep=True)
    nodes_out = cur[0]
    V = nodes_out.shape[1]
    idx = np.asarray(nbrhoods)[np.asarray(last_nodes)] % V
    logits = np.take_along_axis(nodes_out, idx[:, :, None], axis=1)
    out = logits - logsumexp(logits, axis=1)
    m = np.asarray(mask).astype(bool)
    loss = loss_from_preds(out, y, m, weights, weight_decay)

    d_out = np.where(m[:, None, None], -y / np.sum(m), 0.0)
    d_logits = d_out - np.exp(out) * np.sum(d_out, axis=1, keepdims=True)
    d_cur = [np.zeros_like(c) for c in cur]
    N = flows.shape[0]
    for n in range(N):
        np.add.at(d_cur[0][n, :, 0], idx[n], d_logits[n, :, 0])

    grads = [None] * len(weights)
    shifts_T = [S.T for S in shifts]
    for i in reversed(range(len(saved))):
        x, g, z = saved[i]
        dz = [d_cur[a] * act_grad("relu", z[a]) for a in range(3)]
        d_prev = [np.zeros_like(x[a]) for a in range(3)]
        for k in range(7):
            a, b = BUNCH_SRC[k], BUNCH_DST[k]
            grads[7 * i + k] = np.einsum("nrc,nro->co", g[k], dz[b])
            if i > 0:
                d_prev[a] = d_prev[a] + _apply_shift(shifts_T[k], dz[b] @ weights[7 * i + k].T)
        d_cur = d_prev
    for k in range(len(weights)):
        grads[k] = grads[k] + 2 * weight_decay * weights[k]
    return loss, grads


# ----------------------------------------------------------------------------------------------
# batch mask of one optimiser step (STM:313, 319-322)
# ----------------------------------------------------------------------------------------------

def draw_batch_mask(rs, N, batch_size, train_mask):
    bm = np.array([1] * batch_size + [0] * (N - batch_size))
    rs.shuffle(bm)
    return np.logical_and(bm, train_mask)


# ----------------------------------------------------------------------------------------------
# dataset helpers shared by tests: dense arrays from the committed golden fixtures
# ----------------------------------------------------------------------------------------------

def dense_from_coo(row, col, val, shape):
    M = np.zeros(tuple(int(s) for s in shape))
    M[row, col] = val
    return M


def flows_from_ragged(ptr, idx, val, n_edges):
    N = len(ptr) - 1
    X = np.zeros((N, n_edges, 1))
    for i in range(N):
        X[i, idx[ptr[i]:ptr[i + 1]], 0] = val[ptr[i]:ptr[i + 1]]
    return X


def onehot_targets(choice, D):
    y = np.zeros((len(choice), D, 1))
    y[np.arange(len(choice)), choice, 0] = 1.0
    return y
