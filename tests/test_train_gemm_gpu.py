"""csrc/train_gemm.hip — the bf16 MFMA contractions of the training path (forward / data gradient: pdm_tg_gemm_nt, weight
gradient: pdm_tg_wgrad) against the contract the bf16-emulating checker states (oracle/cpu_detector.py::_MatmulBf16): bf16
operands, exact products, fp32 accumulation, one rounding of the forward result, fp32 weight gradient.  Integer-valued
operands make every fp32 sum exact, so the comparisons are BIT-exact and any fragment-layout / transposition / tile-edge
mistake shows as a wrong number, not as noise; a random-data case bounds the fp32 summation-order difference."""
import numpy as np
import pytest
import torch

from pdm_ssd_amd import train_gemm as tg

pytestmark = pytest.mark.gpu


def ints(shape, lo, hi, seed, dev):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi + 1, shape, generator=g).float().to(dev)


@pytest.mark.parametrize("R,K,N,ldx", [(1000, 64, 32, 64), (300, 72, 200, 80), (129, 8, 8, 8), (4096, 1536, 512, 1536),
                                        (20000, 128, 128, 128), (77, 200, 136, 208), (70000, 8, 16, 8), (9000, 104, 64, 104),
                                        (513, 32, 48, 32), (255, 16, 24, 16)])   # all three tile shapes: N <= 32, <= 64, wider
def test_gemm_nt_exact_on_integer_data(dev, R, K, N, ldx):
    xs = torch.zeros((R, ldx), dtype=torch.bfloat16, device=dev)
    xs[:, :K] = ints((R, K), -4, 4, 1, dev).bfloat16()
    xs[:, K:] = 7.0                                   # pad columns of the storage must not be read
    x = xs[:, :K]
    w32 = ints((N, K), -3, 3, 2, dev)                 # asymmetric: a swapped row / column shows
    w32[0, :] = 1.0; w32[:, 0] += torch.arange(N, device=dev) % 3
    w = tg.pack_weight(w32)
    assert w.shape == (N, (K + 7) // 8 * 8) and torch.equal(w[:, :K].float(), w32)
    y, st = tg.gemm_nt(x, w, stats=True)
    want = (x.float() @ w32.t()).bfloat16()           # every sum below 2^24: exact in fp32
    assert torch.equal(y, want)
    s = st.double().sum(0)
    np.testing.assert_allclose(s[:, 0].cpu().numpy(), want.double().sum(0).cpu().numpy(), rtol=1e-6, atol=1e-3)
    np.testing.assert_allclose(s[:, 1].cpu().numpy(), want.double().square().sum(0).cpu().numpy(), rtol=1e-6, atol=1e-3)
    # the data gradient is the same call on the transposed weights
    wt = tg.pack_weight(w32, transposed=True)
    assert torch.equal(wt[:, :N].float(), w32.t())
    dy = ints((R, N), -3, 3, 3, dev).bfloat16()
    dx = tg.gemm_nt(dy, wt)                           # (K, ld >= N): the pad columns lie beyond the contraction
    assert torch.equal(dx, (dy.float() @ w32).bfloat16())


def test_gemm_nt_bias_and_strided_output(dev):
    R, K, N = 513, 96, 24
    x = ints((R, K), -4, 4, 5, dev).bfloat16()
    w32 = ints((N, K), -3, 3, 6, dev)
    bias = torch.linspace(-2.0, 2.0, N, device=dev) + 1e-3          # rounded to bf16 before the add
    out = torch.full((R, 32), -1.0, dtype=torch.bfloat16, device=dev)
    y = tg.gemm_nt(x, tg.pack_weight(w32), bias=bias, out=out[:, :N])
    want = (x.float() @ w32.t() + bias.bfloat16().float()).bfloat16()
    assert torch.equal(y, want) and torch.equal(out[:, :N], want)
    assert bool((out[:, N:] == -1.0).all())                           # columns beyond N untouched


@pytest.mark.parametrize("R,K,N", [(5000, 64, 32), (64, 8, 8), (4097, 136, 264), (70000, 128, 256), (1, 16, 16)])
def test_wgrad_exact_on_integer_data(dev, R, K, N):
    x = ints((R, K), -4, 4, 7, dev).bfloat16()
    dy = ints((R, N), -3, 3, 8, dev).bfloat16()
    dy[:, 0] = 1.0; x[:, -1] = torch.arange(R, device=dev).remainder(5).bfloat16()
    dw = tg.wgrad(dy, x)
    want = (dy.double().t() @ x.double()).float()     # |sums| < 2^24: exact
    assert torch.equal(dw, want)
    dw2 = tg.wgrad(dy, x, out=dw.clone(), accumulate=True)
    assert torch.equal(dw2, 2 * want)
    assert torch.equal(tg.wgrad(dy, x), dw)           # slabs summed in a fixed order: bit-reproducible


def test_random_data_within_fp32_summation_order(dev):
    torch.manual_seed(0)
    R, K, N = 30000, 256, 192
    x = torch.randn(R, K, device=dev).bfloat16()
    w32 = torch.randn(N, K, device=dev) * 0.1
    y = tg.gemm_nt(x, tg.pack_weight(w32))
    ref = x.double() @ w32.bfloat16().double().t()
    err = (y.double() - ref).abs() / ref.abs().clamp(min=1e-2)
    assert float(err.max()) <= 2.0 ** -8                 # one bf16 rounding (2^-9) plus fp32 accumulation noise
    flips = float((y != ref.float().bfloat16()).float().mean())
    assert flips <= 1e-3                                 # a rounding flips only where the fp32 sum sits on a bf16 tie
    dy = torch.randn(R, N, device=dev).bfloat16()
    dw = tg.wgrad(dy, x)
    refw = dy.double().t() @ x.double()
    assert float((dw.double() - refw).norm() / refw.norm()) <= 1e-5


def test_row_view_recognises_the_training_layouts(dev):
    a = torch.zeros(2, 16, 5, 4, dtype=torch.bfloat16, device=dev).contiguous(memory_format=torch.channels_last)
    v = tg.row_view(a)
    assert v is not None and v.shape == (40, 16) and v.data_ptr() == a.data_ptr()
    assert tg.row_view(torch.zeros(2, 16, 5, 4, dtype=torch.bfloat16, device=dev)) is None     # channel-major: not rows
    b = torch.zeros(7, 24, dtype=torch.bfloat16, device=dev)
    assert tg.row_view(b) is b and tg.row_view(b[:, :16]).shape == (7, 16)
    assert tg.usable(10, 64, 32) and not tg.usable(10, 99, 32)


@pytest.mark.parametrize("R,N,ld", [(524288, 8, 8), (1000, 24, 32), (77, 512, 512), (1, 64, 64), (30000, 200, 200)])
def test_colsum_exact_on_integer_data(dev, R, N, ld):
    ys = torch.zeros((R, ld), dtype=torch.bfloat16, device=dev)
    ys[:, :N] = ints((R, N), -3, 3, 11, dev).bfloat16()
    ys[:, N:] = 5.0
    got = tg.colsum(ys[:, :N])
    assert torch.equal(got, ys[:, :N].double().sum(0).float())      # |sums| < 2^24: exact in fp32
