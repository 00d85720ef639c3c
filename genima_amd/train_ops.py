"""Eager wrappers of the training-side C ABI entry points (include/genima_hip.h "training-side kernels") on torch CUDA tensors.

Used by genima_amd/training.py (the ControlNet fine-tune step).  Like engine.py: torch is memory/stream plumbing only.
"""
from __future__ import annotations

import os

import ctypes as C
from typing import Optional

import torch

from ._lib import OUT_F32, OUT_ROWMAJOR, AttnBwdDesc, GemmDesc, GroupNormDesc, WgradDesc, check
from .engine import Engine, _ptr

F16, F32 = torch.float16, torch.float32


def _rup(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def gemm(E: Engine, a, w, out, M: int, N: int, K: int, lda: int, ldw: int, ldo: int, *, bias=None, residual=None, ldr: int = 0,
         f32_out: bool = False, accumulate: bool = False, batch: int = 0, batch_inner: int = 1, a_bs=(0, 0), w_bs=(0, 0), out_bs=(0, 0),
         act: int = 0, a_off: int = 0, w_off: int = 0, out_off: int = 0):
    """Raw dense GEMM out[m, n] = sum_k a[m*lda + k] * w[n*ldw + k] (+ epilogue), optionally batched / f32 / accumulating.
    a_off / w_off / out_off: element offsets added to the base pointers (column slices of wider matrices)."""
    d = GemmDesc()
    d.a, d.w, d.out = a.data_ptr() + 2 * a_off, w.data_ptr() + 2 * w_off, out.data_ptr() + out_off * out.element_size()
    d.bias, d.residual = _ptr(bias), _ptr(residual)
    d.M, d.N, d.K, d.lda, d.ldw, d.ldo, d.ldr = M, N, K, lda, ldw, ldo, ldr
    d.out_mode = OUT_F32 if f32_out else OUT_ROWMAJOR
    d.accumulate, d.act, d.out_scale = int(accumulate), act, 1.0
    if batch > 1:
        d.batch, d.batch_inner = batch, batch_inner
        d.a_bs, d.a_bs2 = a_bs
        d.w_bs, d.w_bs2 = w_bs
        d.out_bs, d.out_bs2 = out_bs
    if E.autotune:
        E.apply_plan(d, _tuned_tile(E, d, out))
    else:
        from .engine import _tune_table
        E.apply_plan(d, _tune_table().get(E._tune_key(d), 0))
    nb = int(E.lib.gn_gemm_workspace_bytes(C.byref(d)))
    if nb > 0:
        d.workspace = E._workspace(nb).data_ptr()
    E.run_gemm(d)
    return out


def _tuned_tile(E: Engine, d: GemmDesc, out: torch.Tensor) -> int:
    """Per-shape tile choice (Engine._autotune).  The timing launches write to a scratch copy of the output region, never to the real
    (possibly accumulating) destination."""
    from .engine import _tune_table
    key = E._tune_key(d)
    hit = _tune_table().get(key)
    if hit is not None and not (os.environ.get("GN_RETUNE") and key not in E._retuned):  # (GN_RETUNE: re-race the named tiles against the incumbent, once)
        return hit
    scratch = torch.empty_like(out)
    real_out, real_acc = d.out, d.accumulate
    d.out = scratch.data_ptr() + (real_out - out.data_ptr())
    d.accumulate = 0
    try:
        return E._autotune(d, key)
    finally:
        d.out, d.accumulate = real_out, real_acc


def transpose2d(E: Engine, x: torch.Tensor, rows: int, cols: int, *, ld_in: Optional[int] = None, batch: int = 1, in_bs: int = 0,
                in_off: int = 0, pad_to: int = 8, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x viewed as [batch][rows, cols] (row stride ld_in, element offset in_off) -> [batch][cols, rup(rows, pad_to)], zero padded
    (the pad keeps the transposed matrix usable as a GEMM operand whose reduction length must be a multiple of 8)."""
    ld_in = cols if ld_in is None else ld_in
    ld_out = _rup(rows, pad_to)
    if out is None:
        shape = (batch, cols, ld_out) if batch > 1 else (cols, ld_out)
        if ld_out != rows and ld_out <= _rup(rows, 64):  # the kernel writes the padding itself (no torch fill launch in front of it)
            out = torch.empty(shape, dtype=F16, device=E.device)
            check(E.lib.gn_transpose2d_zpad(E._ctx, x.data_ptr() + 2 * in_off, _ptr(out), rows, cols, ld_in, ld_out, batch, in_bs, cols * ld_out),
                  "gn_transpose2d_zpad")
            return out
        out = (torch.zeros if ld_out != rows else torch.empty)(shape, dtype=F16, device=E.device)
    check(E.lib.gn_transpose2d(E._ctx, x.data_ptr() + 2 * in_off, _ptr(out), rows, cols, ld_in, ld_out, batch, in_bs, cols * ld_out), "gn_transpose2d")
    return out


def wgrad(E: Engine, dy: torch.Tensor, x: torch.Tensor, dw: torch.Tensor, *, ksize: int = 0, stride: int = 1, pad: int = 0, tile: int = 0,
          dbias: Optional[torch.Tensor] = None, dshift: Optional[torch.Tensor] = None, shift_groups: int = 0):
    """dw [N, K] f32 += dy^T . X with both operands in their forward layout (csrc/gemm_tn.hip: no transposed copies).
    Linear (ksize == 0): dy [R, N], x [R, K].  Conv: dy [B, Ho, Wo, N] (or [R, N]), x NHWC [B, H, W, C], K = ksize^2 * C.
    dbias [N] / dshift [shift_groups, N] (f32, accumulated): the column sums of dy over all rows / per block of R / shift_groups rows,
    taken from the fragments the kernel loads anyway (R / shift_groups must be a multiple of 64)."""
    d = WgradDesc()
    N = dy.shape[-1]
    R = dy.numel() // N
    d.dy, d.x, d.dw = _ptr(dy), _ptr(x), _ptr(dw)
    d.R, d.N, d.ld_dy, d.ld_dw, d.tile = R, N, N, dw.stride(0), tile
    if ksize:
        B, H, W, Cc = x.shape
        Ho, Wo = (H + 2 * pad - ksize) // stride + 1, (W + 2 * pad - ksize) // stride + 1
        assert R == B * Ho * Wo, (R, B, Ho, Wo)
        d.conv, d.B, d.H, d.W, d.C, d.KH, d.KW, d.stride, d.pad, d.Ho, d.Wo = 1, B, H, W, Cc, ksize, ksize, stride, pad, Ho, Wo
        d.K = ksize * ksize * Cc
    else:
        d.K = x.shape[-1]
        d.ld_x = d.K
        assert x.numel() // d.K == R
    d.dbias, d.dshift, d.shift_groups = _ptr(dbias), _ptr(dshift), shift_groups if dshift is not None else 0
    if tile == 0:
        from .engine import _tune_table
        key = f"wg|{int(d.conv)}|{R}|{N}|{int(d.K)}|{int(d.C)}|{int(d.KH)}|{int(d.stride)}"
        plan = _tune_table().get(key)
        retune = os.environ.get("GN_RETUNE_WGRAD")  # e.g. GN_RETUNE_WGRAD=3,4: race the named tiles against the incumbent, once per shape
        if E.autotune and (plan is None or (retune and key not in E._retuned)):
            E._retuned.add(key)
            plan = _tune_wgrad(E, d, dw, key, incumbent=plan, tiles=[int(t) for t in retune.split(",")] if (retune and plan is not None) else None)
        if plan:
            d.tile, d.splitk = plan % 100, plan // 100
    _run_wgrad(E, d)
    return dw


def _run_wgrad(E: Engine, d: WgradDesc):
    nb = int(E.lib.gn_wgrad_workspace_bytes(C.byref(d)))
    d.workspace = E._workspace(nb).data_ptr() if nb > 0 else None
    check(E.lib.gn_wgrad(E._ctx, C.byref(d)), "gn_wgrad")


def _tune_wgrad(E: Engine, d: WgradDesc, dw: torch.Tensor, key: str, incumbent=None, tiles=None) -> int:
    """Race the tiles (or ``tiles`` against ``incumbent``: GN_RETUNE_WGRAD) and a few row splits of gn_wgrad on this shape (into a scratch copy of dw) and remember the winner in the
    GEMM tune table (value = tile + 100 * row split, split 0 = the library's heuristic)."""
    from . import engine as _eng
    real, real_sums = d.dw, (d.dbias, d.dshift)
    scratch = torch.empty_like(dw)
    d.dw, d.dbias, d.dshift = scratch.data_ptr(), None, None
    best, best_ms = 0, float("inf")
    e0, e1 = E.event(), E.event()
    try:
        plans = [t + 100 * sk for t in (tiles or (1, 2, 3, 4)) for sk in (0, 64, 32, 16, 8)]
        if incumbent is not None:
            plans = [int(incumbent)] + [q for q in plans if q != int(incumbent)]
        for plan in plans:
            tile, sk = plan % 100, plan // 100
            if (sk and sk * 512 > d.R) or (tile == 3 and (d.N < 128 or d.K < 256)) or (tile == 4 and (d.N < 256 or d.K < 128)):
                continue
            d.tile, d.splitk = tile, sk
            _run_wgrad(E, d)
            ms = float("inf")
            for _ in range(2):
                E.event_record(e0)
                for _ in range(3):
                    _run_wgrad(E, d)
                E.event_record(e1)
                ms = min(ms, E.event_elapsed_ms(e0, e1))
            if ms < (best_ms if incumbent is None or best == 0 else 0.97 * best_ms):  # a challenger must beat the incumbent by 3 %
                best, best_ms = tile + 100 * sk, ms
    finally:
        d.dw, (d.dbias, d.dshift) = real, real_sums
        E.lib.gn_event_destroy(e0)
        E.lib.gn_event_destroy(e1)
    _eng._tune_table()[key] = best
    _eng._tune_dirty[0] = True
    return best


def wgrad_ok(N: int, K: int, conv_C: int = 0) -> bool:
    """Shapes gn_wgrad takes (the others keep the transposed-copy GEMM path)."""
    return N % 8 == 0 and K % 8 == 0 and conv_C % 8 == 0


def transpose2d_colsum(E: Engine, x: torch.Tensor, rows: int, cols: int, sums) -> torch.Tensor:
    """x [rows, cols] -> x^T [cols, rows], and for every (tensor, groups) of ``sums``: tensor[groups, cols] += the column sums of each
    of the ``groups`` row blocks -- in ONE pass over x (the bias / time-shift gradients ride on the transpose the weight gradient
    needs).  Falls back to transpose2d + colsum when a row block is not a multiple of 64 rows."""
    sums = [(s, g) for s, g in sums if s is not None]
    if not sums:
        return transpose2d(E, x, rows, cols)
    if cols % 8 != 0 or any(rows % g != 0 or (rows // g) % 64 != 0 for _, g in sums):
        for s, g in sums:
            colsum(E, x, s, g, rows // g, cols, cols)
        return transpose2d(E, x, rows, cols)
    out = torch.empty((cols, rows), dtype=F16, device=x.device)
    ws = E._workspace((rows // 64) * cols * 4)
    assert len(sums) <= 2
    (s0, g0), (s1, g1) = sums[0], (sums[1] if len(sums) > 1 else (None, 0))
    check(E.lib.gn_transpose2d_colsum(E._ctx, _ptr(x), _ptr(out), rows, cols, cols, rows, _ptr(s0), g0, _ptr(s1), g1, _ptr(ws)), "gn_transpose2d_colsum")
    return out


def conv_weight_dgrad(E: Engine, w: torch.Tensor, taps: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Packed conv weight [Cout, taps*Cin] -> the data-gradient conv's weight [Cin, taps*Cout]: in/out channels swapped and the
    taps rotated by 180 degrees (out[ci][taps-1-t][co] = w[co][t][ci]); one batched tile transpose, batch = taps."""
    Cout, K = w.shape
    Cin = K // taps
    if out is None:
        out = torch.empty((Cin, taps * Cout), dtype=F16, device=E.device)
    check(E.lib.gn_transpose2d(E._ctx, _ptr(w), out.data_ptr() + 2 * (taps - 1) * Cout, Cout, Cin, taps * Cin, taps * Cout, taps, Cin, -Cout),
          "gn_transpose2d(conv weight)")
    return out


def im2col_t(E: Engine, x: torch.Tensor, ksize: int, stride: int, pad: int) -> torch.Tensor:
    B, H, W, Cc = x.shape
    Ho, Wo = (H + 2 * pad - ksize) // stride + 1, (W + 2 * pad - ksize) // stride + 1
    out = torch.empty((ksize * ksize * Cc, B * Ho * Wo), dtype=F16, device=E.device)
    check(E.lib.gn_im2col_t(E._ctx, _ptr(x), _ptr(out), B, H, W, Cc, ksize, stride, pad), "gn_im2col_t")
    return out


def colsum(E: Engine, x: torch.Tensor, out: torch.Tensor, nb: int, rows_per_batch: int, cols: int, ld: int, accumulate: bool = True):
    ws = E._workspace(int(E.lib.gn_colsum_workspace_bytes(nb, rows_per_batch, cols)))
    check(E.lib.gn_colsum_f32(E._ctx, _ptr(x), _ptr(out), nb, rows_per_batch, cols, ld, _ptr(ws), int(accumulate)), "gn_colsum_f32")
    return out


def act_bwd(E: Engine, dy: torch.Tensor, z: torch.Tensor, act: int) -> torch.Tensor:
    dz = torch.empty_like(dy)
    check(E.lib.gn_act_bwd(E._ctx, _ptr(dy), _ptr(z), _ptr(dz), dy.numel(), act), "gn_act_bwd")
    return dz


def geglu_fwd(E: Engine, hg: torch.Tensor, block: int = 0) -> torch.Tensor:
    Hd = hg.shape[-1] // 2
    out = torch.empty(tuple(hg.shape[:-1]) + (Hd,), dtype=F16, device=E.device)
    check(E.lib.gn_geglu_fwd(E._ctx, _ptr(hg), _ptr(out), hg.numel() // (2 * Hd), Hd, block), "gn_geglu_fwd")
    return out


def geglu_bwd(E: Engine, dy: torch.Tensor, hg: torch.Tensor, block: int = 0) -> torch.Tensor:
    Hd = hg.shape[-1] // 2
    dhg = torch.empty_like(hg)
    check(E.lib.gn_geglu_bwd(E._ctx, _ptr(dy), _ptr(hg), _ptr(dhg), hg.numel() // (2 * Hd), Hd, block), "gn_geglu_bwd")
    return dhg


def softmax_rows_masked(E: Engine, s: torch.Tensor, scale: float, valid: int):
    cols = s.shape[-1]
    check(E.lib.gn_softmax_rows_masked(E._ctx, _ptr(s), s.numel() // cols, cols, cols, float(scale), valid), "gn_softmax_rows_masked")
    return s


def softmax_bwd(E: Engine, p: torch.Tensor, dp: torch.Tensor, scale: float):
    cols = p.shape[-1]
    check(E.lib.gn_softmax_bwd(E._ctx, _ptr(p), _ptr(dp), p.numel() // cols, cols, p.stride(-2), float(scale)), "gn_softmax_bwd")
    return dp


def attention_bwd(E: Engine, q, q_off: int, k, k_off: int, v, o, d_o, lse, heads: int, nk_valid: int, dq, dk, dv):
    """Flash-attention backward.  q [B, N, ldq] / k [B, Nkr, ldk] (head columns start at q_off / k_off), v / o / d_o [B, *, C];
    dq / dk are written at the same column offsets of buffers shaped like q / k (they may be one buffer), dv like v."""
    B, N, ldq = q.shape
    Nkr, ldk = k.shape[1], k.shape[2]
    Cc = v.shape[-1]
    D = Cc // heads
    delta = torch.empty((B, heads, N), dtype=F32, device=E.device)
    d = AttnBwdDesc()
    d.q, d.k, d.v, d.o, d.d_o = q.data_ptr() + 2 * q_off, k.data_ptr() + 2 * k_off, _ptr(v), _ptr(o), _ptr(d_o)
    d.lse, d.delta = _ptr(lse), _ptr(delta)  # (qt / kt / dot stay NULL: the kernels transpose out of their LDS tiles)
    d.dq, d.dk, d.dv = dq.data_ptr() + 2 * q_off, dk.data_ptr() + 2 * k_off, _ptr(dv)
    d.q_bs, d.k_bs, d.v_bs, d.o_bs, d.do_bs = q.stride(0), k.stride(0), v.stride(0), o.stride(0), d_o.stride(0)
    d.q_rs, d.k_rs, d.v_rs, d.o_rs, d.do_rs = q.stride(1), k.stride(1), v.stride(1), o.stride(1), d_o.stride(1)
    d.dq_bs, d.dk_bs, d.dv_bs = dq.stride(0), dk.stride(0), dv.stride(0)
    d.dq_rs, d.dk_rs, d.dv_rs = dq.stride(1), dk.stride(1), dv.stride(1)
    d.B, d.heads, d.Nq, d.Nk, d.Nk_rows, d.D, d.scale = B, heads, N, nk_valid, Nkr, D, float(D) ** -0.5
    check(E.lib.gn_attention_bwd(E._ctx, C.byref(d)), "gn_attention_bwd")


def layernorm_bwd(E: Engine, x, gamma, dy, dgamma: Optional[torch.Tensor] = None, dbeta: Optional[torch.Tensor] = None,
                  eps: float = 1e-5, add: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dgamma / dbeta: f32 [C] views inside the flat gradient buffer (accumulated), or None.  add: the gradient x already holds from
    another branch (same shape): the result is their f16 sum, without a separate add launch."""
    Cc = x.shape[-1]
    M = x.numel() // Cc
    dx = torch.empty_like(x)
    ws = E._workspace(int(E.lib.gn_layernorm_bwd_workspace_bytes(M, Cc))) if dgamma is not None else None
    check(E.lib.gn_layernorm_bwd(E._ctx, _ptr(x), _ptr(gamma), _ptr(dy), _ptr(dx), _ptr(dgamma), _ptr(dbeta), _ptr(ws), M, Cc, eps, _ptr(add)), "gn_layernorm_bwd")
    return dx


class GNSaved:
    """What a training-mode GroupNorm keeps for its backward."""
    __slots__ = ("desc", "x", "x2", "gamma", "beta", "stats", "scsh", "keep")


def groupnorm_fwd_train(E: Engine, x, gamma, beta, groups: int, eps: float, act: int, x2=None):
    B, C1 = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * C1)
    C2 = x2.shape[-1] if x2 is not None else 0
    out = torch.empty(tuple(x.shape[:-1]) + (C1 + C2,), dtype=F16, device=E.device)
    s = GNSaved()
    s.stats = torch.empty((B, groups, 2), dtype=F32, device=E.device)
    s.scsh = torch.empty((B, C1 + C2, 2), dtype=F32, device=E.device)
    d = GroupNormDesc()
    d.x, d.x2, d.gamma, d.beta, d.y = _ptr(x), _ptr(x2), _ptr(gamma), _ptr(beta), _ptr(out)
    d.B, d.HW, d.C1, d.C2, d.groups, d.act, d.eps = B, HW, C1, C2, groups, act, eps
    d.save_stats, d.save_scsh = s.stats.data_ptr(), s.scsh.data_ptr()
    d.workspace = E._workspace(int(E.lib.gn_groupnorm_workspace_bytes(C.byref(d)))).data_ptr()
    check(E.lib.gn_groupnorm_fwd(E._ctx, C.byref(d)), "gn_groupnorm_fwd")
    s.desc, s.x, s.x2, s.gamma, s.beta = d, x, x2, gamma, beta
    return out, s


def groupnorm_bwd(E: Engine, s: GNSaved, dy, need_dx: bool = True, need_dx2: bool = True, dgamma: Optional[torch.Tensor] = None,
                  dbeta: Optional[torch.Tensor] = None, add: Optional[torch.Tensor] = None, add2: Optional[torch.Tensor] = None):
    """-> (dx, dx2).  dgamma / dbeta: f32 [C] views (accumulated), or None.  add / add2: gradients x / x2 already hold (summed in)."""
    d = s.desc
    dx = torch.empty_like(s.x) if need_dx else None
    dx2 = torch.empty_like(s.x2) if (s.x2 is not None and need_dx2) else None
    Cc = d.C1 + d.C2
    ws = E._workspace(int(E.lib.gn_groupnorm_bwd_workspace_bytes(d.B, d.HW, Cc)))
    check(E.lib.gn_groupnorm_bwd(E._ctx, C.byref(d), _ptr(dy), _ptr(dx), _ptr(dx2), _ptr(s.scsh), _ptr(s.stats), _ptr(dgamma), _ptr(dbeta), _ptr(ws),
                                 _ptr(add if need_dx else None), _ptr(add2 if dx2 is not None else None)), "gn_groupnorm_bwd")
    return dx, dx2


def concat_channels(E: Engine, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """[B, H, W, Ca] ++ [B, H, W, Cb] -> [B, H, W, Ca + Cb] (torch.cat(dim=1) of the NCHW reference): two strided copies."""
    Bn, H, Wd, Ca = a.shape
    Cb = b.shape[-1]
    out = torch.empty((Bn, H, Wd, Ca + Cb), dtype=F16, device=E.device)
    P = Bn * H * Wd
    E.copy4d(a, out, (1, 1, 1, P), (0, 0, 0, Ca), (0, 0, 0, Ca + Cb), Ca)
    E.copy4d(b, out[..., Ca:], (1, 1, 1, P), (0, 0, 0, Cb), (0, 0, 0, Ca + Cb), Cb)
    return out


def upsample_nearest2x(E: Engine, x: torch.Tensor) -> torch.Tensor:
    """F.interpolate(scale_factor=2, mode="nearest") on NHWC as four strided copies (out[b, 2y+dy, 2x+dx] = x[b, y, x])."""
    Bn, H, Wd, Cc = x.shape
    out = torch.empty((Bn, 2 * H, 2 * Wd, Cc), dtype=F16, device=E.device)
    for dy in (0, 1):
        for dx in (0, 1):
            E.copy4d(x, out[:, dy:, dx:], (1, Bn, H, Wd), (0, H * Wd * Cc, Wd * Cc, Cc), (0, 4 * H * Wd * Cc, 4 * Wd * Cc, 2 * Cc), Cc)
    return out


def zero_upsample2x(E: Engine, x: torch.Tensor) -> torch.Tensor:
    B, H, W, Cc = x.shape
    out = torch.empty((B, 2 * H, 2 * W, Cc), dtype=F16, device=E.device)
    check(E.lib.gn_zero_upsample2x(E._ctx, _ptr(x), _ptr(out), B, H, W, Cc), "gn_zero_upsample2x")
    return out


def sumpool2x2(E: Engine, x: torch.Tensor) -> torch.Tensor:
    B, H2, W2, Cc = x.shape
    out = torch.empty((B, H2 // 2, W2 // 2, Cc), dtype=F16, device=E.device)
    check(E.lib.gn_sumpool2x2(E._ctx, _ptr(x), _ptr(out), B, H2 // 2, W2 // 2, Cc), "gn_sumpool2x2")
    return out


def mse_loss(E: Engine, pred: torch.Tensor, target: torch.Tensor, C_valid: int, grad_scale: float = 1.0):
    """pred [..., ldp] (first C_valid channels valid), target [..., ldt] -> (loss f32 [1] device tensor, dpred like pred).
    dpred = grad_scale * d(mean squared error)/dpred (grad_scale = the loss scale)."""
    ldp, ldt = pred.shape[-1], target.shape[-1]
    pixels = pred.numel() // ldp
    dpred = torch.empty_like(pred)
    loss = torch.empty(1, dtype=F32, device=E.device)
    ws = E._workspace(4096)
    check(E.lib.gn_mse_loss(E._ctx, _ptr(pred), _ptr(target), _ptr(dpred), _ptr(loss), _ptr(ws), pixels, C_valid, ldp, ldt, float(grad_scale)), "gn_mse_loss")
    return loss, dpred


def sumsq(E: Engine, x: torch.Tensor, out: torch.Tensor):
    ws = E._workspace(8192)
    check(E.lib.gn_sumsq_f32(E._ctx, _ptr(x), x.numel(), _ptr(out), _ptr(ws)), "gn_sumsq_f32")
    return out


def clip_coef(E: Engine, sumsq_t: torch.Tensor, clip: torch.Tensor, max_norm: float, inv_scale: float = 1.0):
    """clip: f32 [3] = (coefficient, unscaled norm, found_inf)."""
    check(E.lib.gn_clip_coef(E._ctx, _ptr(sumsq_t), _ptr(clip), float(max_norm), float(inv_scale)), "gn_clip_coef")
    return clip


def adamw(E: Engine, param, grad, m, v, lr, beta1, beta2, eps, wd, step: int, clip: Optional[torch.Tensor] = None, grad_scale: float = 1.0,
          half_out: Optional[torch.Tensor] = None, zero_grad: bool = False):
    """half_out: f16 working copy refreshed in the same pass (no cast pass over the master); zero_grad: the gradient is cleared in the same pass."""
    check(E.lib.gn_adamw_flat(E._ctx, _ptr(param), _ptr(grad), _ptr(m), _ptr(v), param.numel(), lr, beta1, beta2, eps, wd, step, _ptr(clip), grad_scale,
                              _ptr(half_out), int(zero_grad)), "gn_adamw_flat")


def latent_sample(E: Engine, moments: torch.Tensor, eps: torch.Tensor, C_lat: int, scale: float, ld_out: int = 8) -> torch.Tensor:
    """moments [..., >= 2*C_lat] (mean | logvar), eps [..., >= C_lat] -> f16 [..., ld_out] scaled posterior sample, zero padded."""
    out = torch.empty(tuple(moments.shape[:-1]) + (ld_out,), dtype=F16, device=E.device)
    pixels = moments.numel() // moments.shape[-1]
    check(E.lib.gn_latent_sample(E._ctx, _ptr(moments), _ptr(eps), _ptr(out), pixels, C_lat, moments.shape[-1], eps.shape[-1], ld_out, float(scale)),
          "gn_latent_sample")
    return out


def cast_f32_f16(E: Engine, x: torch.Tensor, out: torch.Tensor):
    check(E.lib.gn_cast_f32_f16(E._ctx, _ptr(x), _ptr(out), x.numel()), "gn_cast_f32_f16")
    return out


def ema_flat(E, shadow: torch.Tensor, param: torch.Tensor, one_minus_decay: float):
    """shadow <- shadow - one_minus_decay * (shadow - param) on flat fp32 buffers (diffusers EMAModel.step)."""
    assert shadow.numel() == param.numel() and shadow.numel() % 4 == 0
    check(E.lib.gn_ema_flat(E._ctx, _ptr(shadow), _ptr(param), shadow.numel(), float(one_minus_decay)), "gn_ema_flat")


def fill_f32(E: Engine, x: torch.Tensor, v: float = 0.0):
    check(E.lib.gn_fill_f32(E._ctx, _ptr(x), x.numel(), float(v)), "gn_fill_f32")
    return x


# ---- ACT controller update (csrc/act_train.hip) ------------------------------------------------------------------------------------------
def film_bwd(E: Engine, dy, x, gamma, beta, rows_per_film: int, act: int, dx, dz=None, dzx=None):
    Cc = x.shape[-1]
    check(E.lib.gn_film_bwd(E._ctx, _ptr(dy), _ptr(x), _ptr(gamma), _ptr(beta), gamma.stride(0), rows_per_film, x.numel() // Cc, Cc, act,
                            _ptr(dx), _ptr(dz), _ptr(dzx)), "gn_film_bwd")


def dropout(E: Engine, x: torch.Tensor, keep_mask: torch.Tensor, scale: float) -> torch.Tensor:
    out = torch.empty_like(x)
    check(E.lib.gn_dropout(E._ctx, _ptr(x), _ptr(keep_mask), _ptr(out), x.numel(), float(scale)), "gn_dropout")
    return out


def add_f32_to_f16(E: Engine, src: torch.Tensor, dst_flat: torch.Tensor, ld_dst: int, B: int, Cc: int):
    check(E.lib.gn_add_f32_to_f16(E._ctx, _ptr(src), src.stride(0), _ptr(dst_flat), ld_dst, B, Cc), "gn_add_f32_to_f16")


def cvae_sample(E: Engine, info: torch.Tensor, eps: torch.Tensor, L: int, ldz: int) -> torch.Tensor:
    B = info.shape[0]
    z = torch.zeros((B, ldz), dtype=F16, device=E.device)
    check(E.lib.gn_cvae_sample(E._ctx, _ptr(info), info.stride(0), _ptr(eps), _ptr(z), ldz, B, L), "gn_cvae_sample")
    return z


def cvae_bwd(E: Engine, info: torch.Tensor, eps: torch.Tensor, dz: torch.Tensor, L: int, kl_scale: float) -> torch.Tensor:
    B = info.shape[0]
    dinfo = torch.zeros_like(info)
    check(E.lib.gn_cvae_bwd(E._ctx, _ptr(info), info.stride(0), _ptr(eps), _ptr(dz), dz.stride(0), _ptr(dinfo), B, L, float(kl_scale)), "gn_cvae_bwd")
    return dinfo


def act_loss(E: Engine, a_hat: torch.Tensor, actions: torch.Tensor, info: torch.Tensor, T_valid: int, A: int, L: int, kl_weight: float,
             grad_scale: float, is_pad: Optional[torch.Tensor] = None):
    """a_hat f16 [B, T_rows, ld] -> (out4 f32 = (loss, l1, gripper, kl), d_a_hat f16 like a_hat)."""
    B, Tr, ld = a_hat.shape
    out4 = torch.zeros(4, dtype=F32, device=E.device)
    d = torch.empty_like(a_hat)
    check(E.lib.gn_act_loss(E._ctx, _ptr(a_hat), ld, Tr * ld, _ptr(actions), _ptr(is_pad), _ptr(info), info.stride(0), B, T_valid, Tr, A, L,
                            float(kl_weight), float(grad_scale), _ptr(out4), _ptr(d)), "gn_act_loss")
    return out4, d
