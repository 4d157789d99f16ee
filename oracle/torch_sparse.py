"""Sparse torch-CPU restatement of the reference formulation -- TEST INFRASTRUCTURE ONLY.

CPU baseline B2 of SURVEY.md section 8(d): the same math as the reference's scone_func (TE:137-152) and loss
(STM:42-56) in fp32 -- the reference's compute type (JAX default, x64 off) -- with the dense (E, E) shift matrices
replaced by torch.sparse_csr tensors, because dense shifts do not exist beyond E ~ 1e4 (4 TB each at E = 1M).
Forward, autograd backward and the ridge term; multi-threaded through torch's intra-op pool (torch.get_num_threads()).

Never imported by scone_gcn_amd/.  "parity unpinned" in the sense of oracle/scone_oracle.py's header; cross-checked
against that fp64 oracle in tests/test_oracle.py.
"""
import numpy as np
import torch


def csr_tensor(m, dtype=torch.float32):
    """scipy sparse matrix -> torch.sparse_csr tensor."""
    m = m.tocsr()
    m.sort_indices()
    return torch.sparse_csr_tensor(torch.from_numpy(m.indptr.astype(np.int64)), torch.from_numpy(m.indices.astype(np.int64)),
                                   torch.from_numpy(m.data.astype(np.float64)).to(dtype), size=m.shape)


class _Shift(torch.autograd.Function):
    """Y = S @ H for H (E, K) dense; backward S^T @ dY (the transposed operator is passed in: CSR stays CSR)."""

    @staticmethod
    def forward(ctx, S, S_T, H):
        ctx.S_T = S_T
        return torch.sparse.mm(S, H)

    @staticmethod
    def backward(ctx, dY):
        return None, None, torch.sparse.mm(ctx.S_T, dY.contiguous())


def scone_forward(weights, S_lower, S_upper, S_lower_T, S_upper_T, inc_rows, last_nodes, flows, act=torch.tanh):
    """Batched scone_func.  flows (N, E, 1) fp32; S_* torch.sparse_csr (E, E); inc_rows(n) -> (edge idx, slot d,
    sign, D): Bcond(last) as sparse rows (make_inc_rows).
    Layout: activations as (E, N, C) so that S @ H is one CSR x dense product over all trajectories."""
    n_layers = (len(weights) - 1) / 3
    assert n_layers % 1 == 0, "wrong number of weights"                          # TE:141-142
    N, E, _ = flows.shape
    cur = flows.permute(1, 0, 2).contiguous()                                    # (E, N, C_in)
    for i in range(int(n_layers)):
        c = cur.shape[2]
        flat = cur.reshape(E, N * c)
        lo = _Shift.apply(S_lower, S_lower_T, flat).reshape(E, N, c)
        up = _Shift.apply(S_upper, S_upper_T, flat).reshape(E, N, c)
        cur = act(cur @ weights[3 * i] + lo @ weights[3 * i + 1] + up @ weights[3 * i + 2])   # TE:145-149
    hw = (cur @ weights[-1]).squeeze(-1)                                         # (E, N): H W_last  (TE:151, re-associated)
    logits = []
    for n in range(N):
        e_idx, slot, sign, D = inc_rows(int(last_nodes[n]))
        lg = torch.zeros(D, dtype=hw.dtype)
        if len(e_idx):
            lg = lg.index_add(0, torch.from_numpy(slot), hw[torch.from_numpy(e_idx), n] * torch.from_numpy(sign).to(hw.dtype))
        logits.append(lg)
    logits = torch.stack(logits)                                                 # (N, D); padding slots keep logit 0
    return (logits - torch.logsumexp(logits, dim=1, keepdim=True)).unsqueeze(-1) # TE:152


def make_inc_rows(B1_csr, nbrhoods):
    """Bcond(n) = B1_ext[nbrhoods[n]] (TE:288, 298-303) as sparse pieces: (edge idx, slot d, sign, D); slot d of a
    padding entry (-1 -> the appended zero row) simply has no pieces."""
    B1 = B1_csr.tocsr()
    nb = np.asarray(nbrhoods)
    D = nb.shape[1]

    def rows(n):
        es, ds, sg = [], [], []
        for d, v in enumerate(nb[n]):
            if v < 0:
                continue
            j0, j1 = B1.indptr[v], B1.indptr[v + 1]
            es.append(B1.indices[j0:j1].astype(np.int64))
            sg.append(B1.data[j0:j1].astype(np.float64))
            ds.append(np.full(j1 - j0, d, np.int64))
        if not es:
            z = np.zeros(0, np.int64)
            return z, z, np.zeros(0), D
        return np.concatenate(es), np.concatenate(ds), np.concatenate(sg), D
    return rows


def loss_and_grad(weights, S_lower, S_upper, S_lower_T, S_upper_T, inc_rows, last_nodes, flows, y, weight_decay):
    """STM:42-56 + grad(loss) (STM:307) on a batch where every trajectory is in the mask."""
    ws = [w.detach().clone().requires_grad_(True) for w in weights]
    out = scone_forward(ws, S_lower, S_upper, S_lower_T, S_upper_T, inc_rows, last_nodes, flows)
    loss = -(out * y).sum() / out.shape[0] + weight_decay * sum((w ** 2).sum() for w in ws)
    loss.backward()
    return float(loss.detach()), [w.grad for w in ws]
