"""Spectral checks of a Muon update (test infrastructure).

Five bf16 Newton-Schulz iterations amplify the weak singular directions of a low-rank gradient together with their rounding
noise, so the element-wise relative L2 distance between two bf16 evaluations of the SAME update can reach O(1) on rank-16 LoRA
gradients and bounds nothing there.  What the update must satisfy whatever the rounding (reference:
/root/reference/src/whisper_finetune/model/optimizer.py:163-239 -> muon.MuonWithAuxAdam; restated in
oracle/whisper_oracle.py `muon_update`):

  U = sqrt(max(1, rows / cols)) * NS5(M / |M|_F),   M = the momentum-mixed gradient

  1. gains: along every STRONG singular pair (p_i, q_i) of M (sigma_i >= 0.1 sigma_max) the update's gain p_i^T U q_i / scale sits
     in the Newton-Schulz-5 band and within `gain_tol` of the scalar iteration f^5(sigma_i / |M|_F);
  2. subspace: U lives in M's row and column spaces (energy outside <= `leak_tol` + twice the oracle's own bf16 leak), and on the
     strong subspace it agrees with the oracle's bf16 evaluation to `proj_tol` (the rounding noise lives in the weak directions);
  3. sign: <U, M> > 0, with the nuclear-norm-weighted mean gain in [0.6, 1.25].
"""
from __future__ import annotations

import torch

NS_COEFFS = (3.4445, -4.7750, 2.0315)
NS_BAND = (0.68, 1.14)  # f^5 over the normalised singular values a strong direction can have (checked in the CPU test)


def ns5_scalar(x: torch.Tensor, steps: int = 5) -> torch.Tensor:
    a, b, c = NS_COEFFS
    x = x.double()
    for _ in range(steps):
        x = a * x + b * x ** 3 + c * x ** 5
    return x


def spectral_violations(U: torch.Tensor, M: torch.Tensor, U_ref: torch.Tensor | None = None, *, strong: float = 0.1,
                        gain_tol: float = 0.1, leak_tol: float = 5e-2, proj_tol: float = 5e-2) -> list[str]:
    """U: the update under test (what is subtracted from the parameter, before lr), M: the matrix that went into the
    orthogonalisation (any positive multiple), U_ref: the oracle's update for the projection check.  -> list of violated
    properties (empty = pass)."""
    U, M = U.double().reshape(M.shape[0], -1), M.double().reshape(M.shape[0], -1)
    rows, cols = M.shape
    scale = max(1.0, rows / cols) ** 0.5
    Uh = U / scale
    P, S, Qt = torch.linalg.svd(M, full_matrices=False)
    fro = S.norm()
    if fro == 0:
        return [] if U.norm() == 0 else ["update of a zero gradient is not zero"]
    k = int((S >= strong * S[0]).sum())
    Pk, Qk = P[:, :k], Qt[:k].T
    out = []
    gains = torch.einsum("ik,ij,jk->k", Pk, Uh, Qk)
    want = ns5_scalar(S[:k] / fro)
    if not bool(((gains > NS_BAND[0] - gain_tol) & (gains < NS_BAND[1] + gain_tol)).all()):
        out.append(f"gains outside the NS-5 band: min {gains.min():.3f} max {gains.max():.3f}")
    if float((gains - want).abs().max()) > gain_tol:
        out.append(f"gains off the scalar iteration by {float((gains - want).abs().max()):.3f}")
    r = int((S > 1e-6 * S[0]).sum())
    Pr, Qr = P[:, :r], Qt[:r].T

    def leak_of(X):
        return float((X - Pr @ (Pr.T @ X @ Qr) @ Qr.T).norm() / (X.norm() + 1e-300))

    # bf16 rounding noise is not confined to M's row / column space and the iteration amplifies it with the weak directions
    # (20 % of the oracle's own update for a spectrum spread of 1e-3): the allowance follows the oracle's bf16 evaluation
    Rh = None if U_ref is None else U_ref.double().reshape(rows, cols) / scale
    leak, allow = leak_of(Uh), leak_tol + (0.0 if Rh is None else 2.0 * leak_of(Rh))
    if leak > allow:
        out.append(f"{leak:.3f} of the update lies outside the gradient's row / column space (allowed {allow:.3f})")
    if U_ref is not None:
        a, b = Pk.T @ Uh @ Qk, Pk.T @ Rh @ Qk
        d = float((a - b).norm() / (b.norm() + 1e-300))
        if d > proj_tol:
            out.append(f"strong-subspace projection differs from the oracle's by {d:.3f}")
    inner = float((Uh * M).sum())
    mean_gain = inner / float(S.sum())
    if not (inner > 0 and 0.6 <= mean_gain <= 1.25):
        out.append(f"<U, M> / |M|_* = {mean_gain:.3f} (sign / overall scale)")
    return out


def negative_controls(U: torch.Tensor, M: torch.Tensor) -> dict:
    """The three wrong updates the check must reject: negated, without the sqrt(max(1, rows/cols)) factor (only different when
    rows > cols), and Newton-Schulz applied to the un-normalised matrix (at a gradient norm of 1e-4)."""
    rows, cols = M.shape[0], M.reshape(M.shape[0], -1).shape[1]
    ctl = {"negated": -U}
    if rows > cols:
        ctl["unscaled"] = U / (rows / cols) ** 0.5
    a, b, c = NS_COEFFS
    # (for |M|_F between ~0.03 and ~1.2 the iteration converges without the normalisation and the "wrong" update is the right
    # one; adapter gradients in training sit orders of magnitude below that: the control is evaluated at |M|_F = 1e-4)
    X = M.double().reshape(rows, cols)
    X = X * (1e-4 / X.norm())
    tr = rows > cols
    X = X.T if tr else X
    for _ in range(5):
        A = X @ X.T
        X = a * X + (b * A + c * A @ A) @ X
    X = X.T if tr else X
    ctl["unnormalised"] = X * max(1.0, rows / cols) ** 0.5
    return ctl
