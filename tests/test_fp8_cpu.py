"""CPU pins of the fp8 oracle (oracle/fp8_torch.py): OCP e4m3 known answers (the same codes tools/probes/mfma_fp8_layout.hip
feeds the MFMA on the GPU), tie handling, the division the scales use, and the size of the quantisation error."""
import torch

from oracle import fp8_torch


def test_e4m3_known_answers_and_ties():
    vals = torch.tensor([0.0, 1.0, 2.0, 0.5, -1.0, 1.5, 3.0, -2.0, 448.0, 2.0 ** -9, 21.0, 23.0, 21.000002])
    codes = vals.to(torch.float8_e4m3fn).view(torch.uint8).tolist()
    #          0     1     2    .5    -1   1.5    3    -2    max  min-sub  tie->even(20)  tie->even(24)  just above the tie -> 22
    assert codes == [0x00, 0x38, 0x40, 0x30, 0xB8, 0x3C, 0x44, 0xC0, 0x7E, 0x01, 0x5A, 0x5C, 0x5B]


def test_quantize_rows_scales_and_zero_row():
    x = torch.tensor([[0.0, 0.0, 0.0, 0.0], [1.0, -2.0, 0.5, 4.0], [5.78125, 0.27099609375, 0.0, 0.0]])
    q, s = fp8_torch.quantize_rows(x)
    assert s[0] == 1.0 and int(q[0].view(torch.uint8).sum()) == 0
    assert s[1] == torch.tensor(4.0) / 448.0 and q[1].float().tolist() == [112.0, -224.0, 56.0, 448.0]
    # 0.27099609375 * (448 / 5.78125) is exactly 21.0 in f32 with an IEEE division: the tie goes to the even neighbour 20
    # (a reciprocal-then-multiply division gives 21.000002 -> 22: the oracle must divide tensor by tensor)
    assert q[2].float().tolist()[:2] == [448.0, 20.0]


def test_linear_fp8_error_vs_exact():
    g = torch.Generator().manual_seed(0)
    x, w = torch.randn(64, 256, generator=g), torch.randn(32, 256, generator=g) / 16
    y = fp8_torch.linear_fp8(x.half().float(), w.half().float())
    ref = x.half().double() @ w.half().double().t()
    err = float((y.double() - ref).norm() / ref.norm())
    assert 0.01 < err < 0.06, err  # e4m3: 2^-4 relative steps on both operands, averaged over K
