"""-m gpu: gn_program_get_gemm / gn_program_set_gemm_plan -- the in-call tuner's two entry points (tools/incall_tune.py).  A recorded
gn_gemm op can be read back and its tile / K split replaced: a tile never changes the result (K is walked alike), a K split changes only
the summation order."""
import ctypes as C

import pytest
import torch

from genima_amd._lib import GemmDesc
from genima_amd.engine import Engine
from util import assert_close, randn_h

pytestmark = pytest.mark.gpu


def test_recorded_gemm_plan_can_be_read_back_and_replaced():
    E = Engine("cuda:0", record=True)
    x, w, b = randn_h(512, 1280), randn_h(640, 1280, scale=0.03), randn_h(640)
    y = E.linear(x, w, b, name="lin")
    g = E.conv2d(randn_h(2, 16, 16, 128), randn_h(128, 9 * 128, scale=0.03), randn_h(128), name="conv")
    n = E.num_ops
    descs = []
    for i in range(n):
        d = GemmDesc()
        assert E.lib.gn_program_get_gemm(E._prog, i, C.byref(d)) == 0
        descs.append(d)
    assert (descs[0].M, descs[0].N, descs[0].K, descs[0].conv) == (512, 640, 1280, 0)
    assert (descs[1].M, descs[1].N, descs[1].K, descs[1].conv) == (512, 128, 1152, 1)
    E.run()
    E.synchronize()
    ref = x.float() @ w.float().t() + b.float()
    assert_close(y, ref, 1e-3)
    g_rec = g.clone()  # (the recorded plan of the conv may split K: its sums differ from the unsplit ones in the last bit)
    # every block tile gives the same bits (no K split: the workspace stays untouched)
    y0 = g0 = None
    for tile in (10, 9, 11, 17, 18, 23, 24):
        for i in range(n):
            assert E.lib.gn_program_set_gemm_plan(E._prog, i, tile, 1, None) == 0
        y.zero_(); g.zero_()
        E.run()
        E.synchronize()
        if y0 is None:
            y0, g0 = y.clone(), g.clone()
            assert_close(g0, g_rec, 1e-3)
        assert torch.equal(y, y0) and torch.equal(g, g0), tile
    # a K split needs a workspace, changes the summation order only
    ws = torch.empty(4 * 512 * 640, dtype=torch.float32, device="cuda")
    assert E.lib.gn_program_set_gemm_plan(E._prog, 0, 18, 4, C.c_void_p(ws.data_ptr())) == 0
    y.zero_()
    E.run()
    E.synchronize()
    assert_close(y, ref, 1e-3)
    assert E.lib.gn_program_get_gemm(E._prog, n, C.byref(GemmDesc())) != 0           # out of range


def test_plan_that_drops_an_attached_fusion_is_refused_and_leaves_the_op_untouched():
    """gn_program_set_gemm_plan validates the NEW plan against the fusions attached to the op (norm_out lives in the split-K reduce): a
    plan that no longer splits K is refused with an error text, the recorded op keeps its plan, and the program still replays."""
    from genima_amd._lib import ACT_SILU
    from genima_amd.engine import Norm
    E = Engine("cuda:0", record=True)
    E.autotune = False
    x, w, b = randn_h(4, 8, 8, 1280, seed=1), randn_h(1280, 9 * 1280, seed=2, scale=0.01), randn_h(1280, seed=3)
    gamma, beta = randn_h(1280, seed=4), randn_h(1280, seed=5)
    h, y = E.conv2d(x, w, b, splitk=4, norm_out=Norm(gamma, beta, 32, 1e-5, ACT_SILU), name="c")
    ops = [i for i in range(E.num_ops) if E.lib.gn_program_get_gemm(E._prog, i, C.byref(GemmDesc())) == 0]
    assert len(ops) == 1
    d0 = GemmDesc()
    assert E.lib.gn_program_get_gemm(E._prog, ops[0], C.byref(d0)) == 0 and d0.norm_out.y
    assert E.lib.gn_gemm_plan_valid(C.byref(d0)) == 1
    E.run(); E.synchronize()
    y_ref = y.clone()
    ws = torch.empty(8 * 256 * 1280, dtype=torch.float32, device="cuda")
    assert E.lib.gn_program_set_gemm_plan(E._prog, ops[0], 9, 1, None) != 0, "a plan without a K split cannot carry norm_out"
    assert b"norm_out" in E.lib.gn_last_error()
    d1 = GemmDesc()
    assert E.lib.gn_program_get_gemm(E._prog, ops[0], C.byref(d1)) == 0
    assert (d1.tile, d1.splitk) == (d0.tile, d0.splitk), "a refused plan leaves the op as it was"
    assert E.lib.gn_program_set_gemm_plan(E._prog, ops[0], 18, 8, C.c_void_p(ws.data_ptr())) == 0   # another split plan is fine
    y.zero_()
    E.run(); E.synchronize()
    assert_close(y, y_ref.float(), 1e-3)


def test_add_multi_equals_the_single_adds():
    """gn_add_multi: up to 16 independent f16 adds in one launch (the UNet's skip + ControlNet-residual additions), eager and recorded."""
    for record in (False, True):
        E = Engine("cuda:0", record=record)
        shapes = [(2, 64, 64, 320)] * 3 + [(2, 32, 32, 640)] * 3 + [(2, 16, 16, 1280)] * 3 + [(2, 8, 8, 1280)] * 4
        pairs = [(randn_h(*s, seed=2 * i), randn_h(*s, seed=2 * i + 1)) for i, s in enumerate(shapes)]
        outs = E.add_multi(pairs, name="t")
        if record:
            E.run()
        E.synchronize()
        for (a, b), o in zip(pairs, outs):
            assert torch.equal(o, (a.float() + b.float()).half())
