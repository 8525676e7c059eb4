"""CPU ORACLE — TEST INFRASTRUCTURE ONLY: 8-bit AdamW with block-wise dynamic-quantised state.

**PARITY UNPINNED.**  The reference gets this optimizer from a third-party package that is neither vendored in
/root/reference nor installed in this image: `bitsandbytes` (reference call site: model/optimizer.py:241-256,
`bnb.optim.Adam8bit` / `bnb.optim.AdamW8bit(parameters, **optimizer_conf["params"])`; selected by `optimizer.8bit: True`,
configs/config_turbo_best.yaml:66; no pinned version: pyproject.toml lists it without a bound).  The reference's tests hold
no golden vector for it.  This file restates the PUBLISHED algorithm (Dettmers et al., "8-bit Optimizers via Block-wise
Quantization", ICLR 2022, §2 and App. on dynamic tree quantisation; bitsandbytes' `functional.create_dynamic_map` and its
block-wise 2-state update) and the GPU kernel is checked against this restatement plus properties no implementation of the
scheme can violate (tests/test_adam8bit.py): monotone 256-entry maps, round-trip error bounds, an update that stays within one
quantisation step of the fp32 AdamW update.

Only tests/ may import this file; the product path (whisper-finetune_amd/) never does.

The scheme
  * state tensors m (signed) and v (unsigned) are stored as one byte per element in blocks of 2 048 elements, each block with
    an fp32 absmax; byte c of a block decodes to qmap[c] * absmax;
  * qmap = dynamic (tree) quantisation: 7 decades 10^-6 .. 10^0, decade i holding 2^i (signed) or 2^(i+1) (unsigned) values
    linearly spaced in (0.1, 1) * 10^(i-6) (both signs for the signed map), plus the two codes 0 and 1;
  * a step dequantises a block, applies Adam's moment updates in fp32, takes the new absmax of the block, updates the
    parameters (decoupled weight decay after the update, as the package's kernel does), and re-quantises the moments to the
    nearest map entry of (value / new absmax);
  * tensors with fewer than 4 096 elements keep fp32 state (the package's `min_8bit_size`).
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np

BLOCK = 2048
MIN_8BIT_SIZE = 4096


def create_dynamic_map(signed: bool = True, max_exponent_bits: int = 7, total_bits: int = 8) -> np.ndarray:
    """The 256-entry dynamic quantisation map, ascending (bitsandbytes functional.create_dynamic_map, restated).
    signed: 127 positive + 127 negative values + {0, 1}; unsigned: 254 positive values + {0, 1}."""
    data = []
    non_sign_bits = total_bits - 1
    additional_items = 2 ** (non_sign_bits - max_exponent_bits) - 1
    i = 0
    for i in range(max_exponent_bits):
        fraction_items = int(2 ** (i + non_sign_bits - max_exponent_bits) + 1 if signed
                             else 2 ** (i + non_sign_bits - max_exponent_bits + 1) + 1)
        boundaries = np.linspace(0.1, 1, fraction_items, dtype=np.float64)
        means = (boundaries[:-1] + boundaries[1:]) / 2.0
        scale = 10.0 ** (-(max_exponent_bits - 1) + i)
        data += (scale * means).tolist()
        if signed:
            data += (-scale * means).tolist()
    if additional_items > 0:
        boundaries = np.linspace(0.1, 1, additional_items + 1, dtype=np.float64)
        means = (boundaries[:-1] + boundaries[1:]) / 2.0
        scale = 10.0 ** (-(max_exponent_bits - 1) + i)
        data += (scale * means).tolist()
        if signed:
            data += (-scale * means).tolist()
    data.append(0.0)
    data.append(1.0)
    assert len(data) == 2 ** total_bits, len(data)
    return np.sort(np.asarray(data, dtype=np.float64)).astype(np.float32)


def quantize_to_map(x: np.ndarray, qmap: np.ndarray) -> np.ndarray:
    """Nearest map entry (ties to the lower code): code = number of midpoints strictly below x."""
    mid = ((qmap[:-1].astype(np.float32) + qmap[1:].astype(np.float32)) * np.float32(0.5)).astype(np.float32)
    return np.searchsorted(mid, x.astype(np.float32), side="left").astype(np.uint8)


def quantize_blockwise(x: np.ndarray, qmap: np.ndarray, block: int = BLOCK) -> Tuple[np.ndarray, np.ndarray]:
    """-> (codes uint8 [n], absmax f32 [ceil(n / block)])."""
    x = np.asarray(x, dtype=np.float32).ravel()
    n = x.size
    nb = (n + block - 1) // block
    codes = np.zeros(n, dtype=np.uint8)
    absmax = np.zeros(nb, dtype=np.float32)
    for b in range(nb):
        seg = x[b * block:(b + 1) * block]
        am = np.float32(np.max(np.abs(seg))) if seg.size else np.float32(0)
        absmax[b] = am
        inv = np.float32(1.0) / am if am > 0 else np.float32(0)
        codes[b * block:(b + 1) * block] = quantize_to_map(seg * inv, qmap)
    return codes, absmax


def dequantize_blockwise(codes: np.ndarray, absmax: np.ndarray, qmap: np.ndarray, block: int = BLOCK) -> np.ndarray:
    codes = np.asarray(codes).ravel()
    out = qmap[codes].astype(np.float32)
    for b in range(absmax.size):
        out[b * block:(b + 1) * block] *= absmax[b]
    return out


class Adam8bitState:
    """State of ONE tensor (flat), bitsandbytes' names: state1 / state2 (uint8 codes), absmax1 / absmax2, qmap1 / qmap2."""

    def __init__(self, n: int):
        nb = (n + BLOCK - 1) // BLOCK
        self.step = 0
        self.state1 = np.zeros(n, dtype=np.uint8)
        self.state2 = np.zeros(n, dtype=np.uint8)
        self.absmax1 = np.zeros(nb, dtype=np.float32)
        self.absmax2 = np.zeros(nb, dtype=np.float32)
        self.qmap1 = create_dynamic_map(True)
        self.qmap2 = create_dynamic_map(False)


def adamw8bit_step(p: np.ndarray, g: np.ndarray, st: Adam8bitState, lr: float, beta1: float, beta2: float, eps: float,
                   weight_decay: float, gnorm_scale: float = 1.0) -> np.ndarray:
    """One block-wise 8-bit AdamW step on a flat fp32 tensor (returns the new parameters; `st` is updated in place).
    fp32 arithmetic in the order of the package's block-wise 2-state kernel:
        g *= gnorm_scale
        s1 = s1 * beta1 + (1 - beta1) * g ;  s2 = s2 * beta2 + (1 - beta2) * g * g
        p += step_size * s1 / (sqrt(s2) + correction2 * eps),  step_size = -lr * correction2 / correction1,
             correction1 = 1 - beta1^t, correction2 = sqrt(1 - beta2^t)
        p *= 1 - lr * weight_decay            (only when weight_decay > 0)"""
    f = np.float32
    p = np.asarray(p, dtype=np.float32).ravel().copy()
    g = np.asarray(g, dtype=np.float32).ravel() * f(gnorm_scale)
    st.step += 1
    c1 = f(1.0 - beta1 ** st.step)
    c2 = f(math.sqrt(1.0 - beta2 ** st.step))
    step_size = f(-lr) * c2 / c1
    s1 = dequantize_blockwise(st.state1, st.absmax1, st.qmap1)
    s2 = dequantize_blockwise(st.state2, st.absmax2, st.qmap2)
    s1 = s1 * f(beta1) + f(1.0 - beta1) * g
    s2 = s2 * f(beta2) + f(1.0 - beta2) * g * g
    p = p + step_size * (s1 / (np.sqrt(s2) + c2 * f(eps)))
    if weight_decay > 0:
        p = p * f(1.0 - lr * weight_decay)
    st.state1, st.absmax1 = quantize_blockwise(s1, st.qmap1)
    st.state2, st.absmax2 = quantize_blockwise(s2, st.qmap2)
    return p.astype(np.float32)


def adamw32_step(p, g, m, v, step, lr, beta1, beta2, eps, weight_decay):
    """The same update with fp32 state (what the 8-bit step approximates; also the path of tensors below MIN_8BIT_SIZE)."""
    f = np.float32
    c1 = f(1.0 - beta1 ** step)
    c2 = f(math.sqrt(1.0 - beta2 ** step))
    m = m * f(beta1) + f(1.0 - beta1) * g
    v = v * f(beta2) + f(1.0 - beta2) * g * g
    p = p + (f(-lr) * c2 / c1) * (m / (np.sqrt(v) + c2 * f(eps)))
    if weight_decay > 0:
        p = p * f(1.0 - lr * weight_decay)
    return p.astype(np.float32), m.astype(np.float32), v.astype(np.float32)
