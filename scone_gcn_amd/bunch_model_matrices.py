"""Sparse, closed-form Bunch (SCCONV) shift operators.

Same arithmetic as the reference's compute_bunch_matrices / compute_shift_matrices
(trajectory_analysis/bunch_model_matrices.py, BMM:71-135) but every D matrix there is DIAGONAL, so the
dense inv/pinv calls (O(E^3), BMM:88-111) reduce to element-wise reciprocals and the seven shifts come
out as scipy CSR matrices with ~2-11 non-zeros per row.  Entry-wise parity with the reference's own
output is pinned by tests/golden/cfg1_bunch.npz.
"""
import numpy as np
import scipy.sparse as sp


def _diag(v):
    return sp.diags(np.asarray(v, np.float64))


def _pinv_diag(d):
    out = np.zeros_like(d, dtype=np.float64)
    nz = d != 0
    out[nz] = 1.0 / d[nz]
    return out


def compute_D2(B):
    """max(rowsum|B|, 1) as a vector (BMM:44-51)."""
    return np.maximum(np.asarray(abs(B).sum(axis=1)).ravel(), 1.0)


def compute_D5(B2):
    """rowsum|B2| as a vector (BMM:53-60)."""
    return np.asarray(abs(B2).sum(axis=1)).ravel()


def compute_D1(B1, d2):
    """2 * rowsum(|B1| D2) as a vector (BMM:62-69)."""
    return 2.0 * np.asarray((abs(B1) @ _diag(d2)).sum(axis=1)).ravel()


def compute_bunch_matrices(B1, B2):
    """(A0u_n, A1u_n, A1d_n, A2d_n), (d1_pinv, d2_2, d3, d4, d5_pinv) -- diagonals returned as vectors (BMM:71-116)."""
    B1, B2 = sp.csr_matrix(B1, dtype=np.float64), sp.csr_matrix(B2, dtype=np.float64)
    nF = B2.shape[1]
    d2_2, d2_1 = compute_D2(B2), compute_D2(B1)                      # BMM:79-80
    d1 = compute_D1(B1, d2_2)                                        # BMM:82
    d3, d4 = np.full(nF, 1.0 / 3.0), np.ones(nF)                     # BMM:83-84
    d5 = compute_D5(B2)                                              # BMM:85
    d1_p, d5_p = _pinv_diag(d1), _pinv_diag(d5)                      # BMM:88-89

    L0u = B1 @ B1.T @ _diag(1.0 / d2_1)                              # BMM:92
    L1u = _diag(d2_2) @ B1.T @ _diag(d1_p) @ B1                      # BMM:93
    L1d = B2 @ _diag(d3) @ B2.T @ _diag(1.0 / d2_2)                  # BMM:94
    L2d = _diag(d4) @ B2.T @ _diag(d5_p) @ B2                        # BMM:95

    A0u = _diag(d2_1) - L0u @ _diag(d2_1)                            # BMM:100
    A1u = _diag(d2_2) - L1u @ _diag(d2_2)                            # BMM:101
    A1d = _diag(1.0 / d2_2) - _diag(1.0 / d2_2) @ L1d                # BMM:102
    A2d = _diag(1.0 / d4) - _diag(1.0 / d4) @ L2d                    # BMM:103

    I = sp.identity
    A0u_n = (A0u + I(len(d2_1))) @ _diag(1.0 / (d2_1 + 1.0))         # BMM:111
    A1u_n = (A1u + I(len(d2_2))) @ _diag(1.0 / (d2_2 + 1.0))         # BMM:112
    A1d_n = _diag(d2_2 + 1.0) @ (A1d + I(len(d2_2)))                 # BMM:113
    A2d_n = _diag(d4 + 1.0) @ (A2d + I(nF))                          # BMM:114
    return (A0u_n, A1u_n, A1d_n, A2d_n), (d1_p, d2_2, d3, d4, d5_p)


def compute_shift_matrices(B1, B2):
    """S_00, S_10, S_01, S_11, S_21, S_12, S_22 as scipy CSR (BMM:118-135)."""
    B1, B2 = sp.csr_matrix(B1, dtype=np.float64), sp.csr_matrix(B2, dtype=np.float64)
    (A0u_n, A1u_n, A1d_n, A2d_n), (d1_p, d2_2, d3, d4, d5_p) = compute_bunch_matrices(B1, B2)
    S_00 = A0u_n
    S_10 = _diag(d1_p) @ B1
    S_01 = _diag(d2_2) @ B1.T @ _diag(d1_p)
    S_11 = A1d_n + A1u_n
    S_21 = B2 @ _diag(d3)
    S_12 = _diag(d4) @ B2.T @ _diag(d5_p)
    S_22 = A2d_n
    out = []
    for S in (S_00, S_10, S_01, S_11, S_21, S_12, S_22):
        S = sp.csr_matrix(S)
        S.sum_duplicates()
        S.eliminate_zeros()
        S.sort_indices()
        out.append(S)
    return tuple(out)
