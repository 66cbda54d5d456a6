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


def _bn_coef(N, dev, seed):
    """(4, N) [mean | invstd | scale | shift] with integer means and power-of-two invstd (g xhat sums then stay exact on integer data)"""
    g = torch.Generator().manual_seed(seed)
    mean = torch.randint(-2, 3, (N,), generator=g).float()
    invstd = torch.tensor([0.5, 1.0, 2.0])[torch.randint(0, 3, (N,), generator=g)]
    gamma = torch.randint(-2, 3, (N,), generator=g).float()          # zero and negative gammas: the mask follows scale's sign
    shift = torch.randint(-3, 4, (N,), generator=g).float() * 0.5
    return torch.stack([mean, invstd, gamma * invstd, shift]).contiguous().to(dev)


def _bwd_stats_reference(bx, y, coef):
    """the operator these epilogues replace: pdm_bn_relu_backward_stats over (x, dy)"""
    from pdm_ssd_amd import _native
    R, N = y.shape
    grads = torch.empty((4, N), dtype=torch.float32, device=y.device)
    part = torch.empty((_native.lib().pdm_bn_parts(0, R, N, 1), N, 2), dtype=torch.float32, device=y.device)
    _native.call("pdm_bn_relu_backward_stats", torch.cuda.current_stream().cuda_stream, 1, 0, R, N, 1, bx.data_ptr(), y.data_ptr(),
                 coef.data_ptr(), grads.data_ptr(), part.data_ptr(), 1)
    return grads


@pytest.mark.parametrize("R,K,N", [(1000, 64, 32), (129, 8, 8), (300, 72, 200), (20000, 128, 128), (9000, 104, 64), (513, 32, 48),
                                   (70000, 16, 24), (4096, 512, 264), (77, 200, 136)])   # all three tile shapes, ragged edges
def test_gemm_nt_with_bn_backward_statistics_in_the_epilogue(dev, R, K, N):
    """pdm_tg_gemm_nt_bs: the product is unchanged and the sums its epilogue leaves (sum g, sum g xhat with g = y [bn(bx) > 0])
    give the grads (dgamma, dbeta, p, q) of pdm_bn_relu_backward_stats over the same tensors — BIT for bit on integer data (every
    fp32 sum exact), to summation order on random data."""
    x = ints((R, K), -3, 3, 21, dev).bfloat16()
    w32 = ints((N, K), -2, 2, 22, dev)
    w = tg.pack_weight(w32)
    bx = ints((R, N), -4, 4, 23, dev).bfloat16()
    coef = _bn_coef(N, dev, 24)
    y, part = tg.gemm_nt_bs(x, w, bx, coef)
    assert torch.equal(y, tg.gemm_nt(x, w)) and part.shape == (part.shape[0], N, 2)
    got, want = tg.bn_bwd_finalize(R, coef, part), _bwd_stats_reference(bx, y, coef)
    assert torch.equal(got, want)
    # random data: same values up to the order of the fp32 partial sums
    x = torch.randn(R, K, device=dev).bfloat16()
    w = tg.pack_weight(torch.randn(N, K, device=dev) * 0.2)
    bx = (torch.randn(R, N, device=dev) * 1.5 + 0.3).bfloat16()
    coef = torch.stack([torch.randn(N) * 0.3, torch.rand(N) + 0.5, torch.randn(N), torch.randn(N) * 0.3]).contiguous().to(dev)
    y, part = tg.gemm_nt_bs(x, w, bx, coef)
    got, want = tg.bn_bwd_finalize(R, coef, part), _bwd_stats_reference(bx, y, coef)
    scale = want.abs().amax(1, keepdim=True).clamp_min(1e-6)
    assert float(((got - want).abs() / scale).max()) < 2e-5


@pytest.mark.parametrize("R,K,N", [(1000, 64, 32), (5000, 128, 64), (300, 72, 200), (20000, 256, 128), (513, 32, 48), (4097, 40, 256)])
def test_gemm_nt_dy_with_bn_backward_statistics_in_the_epilogue(dev, R, K, N):
    """pdm_tg_gemm_nt_dy_bs (the data gradient that forms its operand from an unformed BatchNorm gradient AND whose product is the
    next BatchNorm's gradient): dX and dYout bit-equal to pdm_tg_gemm_nt_dy, grads equal to the reduce operator's as above."""
    dz = ints((R, K), -3, 3, 31, dev).bfloat16()
    yp = ints((R, K), -4, 4, 32, dev).bfloat16()
    icoef = _bn_coef(K, dev, 33)
    igrads = torch.zeros((4, K), device=dev)
    igrads[2] = ints((K,), -1, 1, 34, dev) * 0.5; igrads[3] = ints((K,), -1, 1, 35, dev) * 0.25
    w = tg.pack_weight(ints((N, K), -2, 2, 36, dev))
    bx = ints((R, N), -4, 4, 37, dev).bfloat16()
    coef = _bn_coef(N, dev, 38)
    dx0, dy0 = tg.gemm_nt_dy(dz, yp, icoef, igrads, w)
    dx, dy, part = tg.gemm_nt_dy(dz, yp, icoef, igrads, w, bs=(bx, coef))
    assert torch.equal(dx, dx0) and torch.equal(dy, dy0)
    got, want = tg.bn_bwd_finalize(R, coef, part), _bwd_stats_reference(bx, dx, coef)
    scale = want.abs().amax(1, keepdim=True).clamp_min(1e-6)
    assert float(((got - want).abs() / scale).max()) < 2e-5


def test_packed_weight_pairs_are_cached_and_refreshed_together(dev):
    """tg.pack_weight_pair keeps the bf16 pair of a parameter until the parameter changes (version counter) and then repacks every
    cached pair in ONE launch (pdm_tg_pack_weight_many): the pairs always equal a fresh pack of the current values; views of a
    parameter (reshape) share its entry; a dead parameter's entry is dropped and its address can be reused."""
    torch.manual_seed(3)
    shapes = [(32, 16), (20, 24), (200, 136), (8, 8), (512, 1536)]
    params = [torch.nn.Parameter(torch.randn(n, k, device=dev)) for n, k in shapes]
    conv = torch.nn.Parameter(torch.randn(64, 40, 1, 1, device=dev))

    def pairs():
        out = [tg.pack_weight_pair(p, (p.shape[0] + 7) // 8 * 8, (p.shape[1] + 7) // 8 * 8) for p in params]
        out.append(tg.pack_weight_pair(conv.reshape(64, -1), 64, 40))
        return out

    def fresh():
        out = [tg._pack_weight_pair_now(p.detach(), (p.shape[0] + 7) // 8 * 8, (p.shape[1] + 7) // 8 * 8) for p in params]
        out.append(tg._pack_weight_pair_now(conv.detach().reshape(64, -1), 64, 40))
        return out

    assert tg.PACK_CACHE
    first = pairs()
    for (a, b), (c, d) in zip(first, fresh()):
        assert torch.equal(a, c) and torch.equal(b, d)
    again = pairs()
    assert all(a.data_ptr() == c.data_ptr() for (a, _), (c, _) in zip(first, again))      # no new buffers, no launches
    with torch.no_grad():
        for p in params:
            p.mul_(1.5).add_(0.25)                     # an optimizer step: every version counter moves
        conv.add_(1.0)
    for (a, b), (c, d) in zip(pairs(), fresh()):
        assert torch.equal(a, c) and torch.equal(b, d)
    # torch's fused multi-tensor optimizer moves no version counter either: a pair asked for the FIRST time since the last refresh, after
    # an optimizer step, must still be current (two models taking turns: the generator / discriminator pattern)
    other = torch.nn.Parameter(torch.randn(40, 24, device=dev))
    pair_other = lambda: tg.pack_weight_pair(other, 40, 24)
    pair_other()
    for p in params + [other]:
        p.grad = torch.randn_like(p)
    opt_a, opt_b = torch.optim.AdamW(params, lr=0.1, fused=True), torch.optim.AdamW([other], lr=0.1, fused=True)
    opt_a.step()
    for (a, b), (c, d) in zip(pairs()[:-1], fresh()[:-1]):      # refresh (clears what was handed out)
        assert torch.equal(a, c) and torch.equal(b, d)
    opt_b.step()                                               # `other` changes; it has not been asked for since the refresh
    a, b = pair_other()
    c, d = tg._pack_weight_pair_now(other.detach(), 40, 24)
    assert torch.equal(a, c) and torch.equal(b, d)
    # an edit through .data moves no version counter and no optimizer hook: the next pass over the same parameters (a pair asked for
    # AGAIN since the last refresh) refreshes anyway; tg.invalidate_packs() is the explicit form
    pairs()
    params[3].data.mul_(-2.0)
    for (a, b), (c, d) in zip(pairs(), fresh()):
        assert torch.equal(a, c) and torch.equal(b, d)
    params[4].data.add_(1.0)
    tg.invalidate_packs()
    for (a, b), (c, d) in zip(pairs(), fresh()):
        assert torch.equal(a, c) and torch.equal(b, d)
    # one parameter changes alone; then one dies and another is born (possibly at its address)
    with torch.no_grad():
        params[2].zero_()
    for (a, b), (c, d) in zip(pairs(), fresh()):
        assert torch.equal(a, c) and torch.equal(b, d)
    del first, again
    params[0] = torch.nn.Parameter(torch.randn(32, 16, device=dev))
    with torch.no_grad():
        params[1].neg_()
    for (a, b), (c, d) in zip(pairs(), fresh()):
        assert torch.equal(a, c) and torch.equal(b, d)


@pytest.mark.parametrize("R,K,N,ns", [(4096, 64, 32, 16), (2048, 32, 64, 32), (4096, 96, 128, 16), (1024, 200, 256, 32), (512, 264, 512, 64),
                                      (6400, 16, 24, 128), (48, 8, 8, 16)])
def test_gemm_nt_leaves_the_pooled_operators_group_extremes(dev, R, K, N, ns):
    """pdm_tg_gemm_nt_pool: the product and its column sums are those of pdm_tg_gemm_nt; per group of ns consecutive rows and channel the
    epilogue leaves max / min of the rounded outputs and the FIRST index attaining each (ties: small integers make many) — all three
    tile shapes, a last row tile that is not full."""
    x = ints((R, K), -2, 2, 41, dev).bfloat16()
    w = tg.pack_weight(ints((N, K), -1, 1, 42, dev))
    y0, st0 = tg.gemm_nt(x, w, stats=True)
    y, st, (keep, idx) = tg.gemm_nt(x, w, stats=True, pool_ns=ns)
    assert torch.equal(y, y0) and torch.equal(st, st0)
    g = y.float().view(R // ns, ns, N)
    mx, imx = g.max(1)
    mn, imn = g.min(1)
    assert torch.equal(keep[0].float(), mx) and torch.equal(keep[1].float(), mn)
    first_max = (g == mx[:, None, :]).float().argmax(1)       # first index attaining the extreme
    first_min = (g == mn[:, None, :]).float().argmax(1)
    assert torch.equal(idx[0].long(), first_max) and torch.equal(idx[1].long(), first_min)
