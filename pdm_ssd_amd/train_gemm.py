"""Python side of csrc/train_gemm.hip: the bf16 contractions of the training path on this library's own MFMA kernels.

Rows layout throughout: a channels-last activation tensor IS a (rows, channels) matrix.  No fallback: the calls raise
when the HIP library is missing or an argument does not fit (callers test `usable()` first and otherwise keep torch)."""
import torch

from . import _native


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def usable(rows, k, n):
    """shapes the kernels take: channel counts multiples of 8 (16-byte rows)"""
    return rows > 0 and k > 0 and n > 0 and k % 8 == 0 and n % 8 == 0


def row_view(x):
    """(rows, C) view of a bf16 tensor whose storage is rows x C with the channel fastest: a contiguous (R, C) matrix, a
    channels-last (B, C, H, W) tensor or a (B, C, L) tensor stored (B, L, C).  None when the layout is something else."""
    if x.dim() == 2:
        return x if x.stride(1) == 1 and x.stride(0) >= x.shape[1] else None
    if x.dim() >= 3:
        xm = x.movedim(1, -1)
        if xm.is_contiguous():
            return xm.reshape(-1, x.shape[1])
    return None


def pack_weight(w, transposed=False, pad_to=None):
    """fp32 (N, K) parameter -> bf16 (N, ld) copy (round to nearest even), or its transpose (K, ld); ld = the row length
    rounded up to a multiple of 8 (pad_to overrides), pad columns zero."""
    assert w.dim() == 2 and w.is_cuda and w.dtype == torch.float32
    w = w.detach().contiguous()
    N, K = w.shape
    inner = N if transposed else K
    ld = (inner + 7) // 8 * 8 if pad_to is None else pad_to
    out = torch.empty((K if transposed else N, ld), dtype=torch.bfloat16, device=w.device)
    _native.call("pdm_tg_pack_weight", _stream(w), N, K, w.data_ptr(), 0 if transposed else out.data_ptr(), 0 if transposed else ld,
                 out.data_ptr() if transposed else 0, ld if transposed else 0)
    return out


def pack_weight_into(w, out, transposed=False):
    """the same into the top-left block of an existing bf16 matrix `out` (rows >= N or K, row stride = out.stride(0)); what
    lies outside the block is left as it is (callers that need zero padding allocate zeros)"""
    assert w.dim() == 2 and w.is_cuda and w.dtype == torch.float32 and out.dtype == torch.bfloat16 and out.stride(1) == 1
    w = w.detach().contiguous()
    N, K = w.shape
    assert out.shape[0] >= (K if transposed else N) and out.shape[1] >= (N if transposed else K)
    ld = out.stride(0)
    # the kernel writes ld columns per row: rows of the block are rewritten whole (pad columns zero)
    _native.call("pdm_tg_pack_weight", _stream(w), N, K, w.data_ptr(), 0 if transposed else out.data_ptr(), 0 if transposed else ld,
                 out.data_ptr() if transposed else 0, ld if transposed else 0)
    return out


def pack_weight_pair(w, rows_to, cols_to):
    """fp32 (N, K) parameter -> (wb (rows_to, cols_to), wt (cols_to, rows_to)) bf16: the weight for the forward product and
    its transpose for the data gradient, zero padded (rows_to >= N, cols_to >= K, both multiples of 8), in ONE launch."""
    assert w.dim() == 2 and w.is_cuda and w.dtype == torch.float32
    w = w.detach().contiguous()
    N, K = w.shape
    buf = torch.empty((2, rows_to * cols_to), dtype=torch.bfloat16, device=w.device)     # the kernel writes the padding too
    wb, wt = buf[0].view(rows_to, cols_to), buf[1].view(cols_to, rows_to)
    _native.call("pdm_tg_pack_weight_pair", _stream(w), N, K, w.data_ptr(), wb.data_ptr(), wt.data_ptr(), rows_to, cols_to)
    return wb, wt


def gemm_nt(x, w, bias=None, stats=False, out=None, x_bn_coef=None):
    """x (R, K) bf16 rows (row stride a multiple of 8), w (N, K') bf16 with K' >= K zero padded -> y (R, N) bf16 =
    x . w^T [+ bias], fp32 accumulation, one rounding.  stats=True also returns the column sums (parts, N, 2) fp32 (sum y,
    sum y^2 of the rounded y, one part per persistent workgroup slot): what pdm_bn_relu_forward_stats takes.
    x_bn_coef (4, K) fp32: x is read through BatchNorm + ReLU with these coefficients (the layer before's, unapplied)."""
    R, K = x.shape
    N = w.shape[0]
    assert x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and x.stride(1) == 1 and w.stride(1) == 1
    y = torch.empty((R, N), dtype=torch.bfloat16, device=x.device) if out is None else out
    st = None
    if stats:
        st = torch.empty((_native.lib().pdm_tg_stats_parts(R, N), N, 2), dtype=torch.float32, device=x.device)
    _native.call("pdm_tg_gemm_nt", _stream(x), R, K, N, x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), y.data_ptr(), y.stride(0),
                 0 if bias is None else bias.data_ptr(), 0 if st is None else st.data_ptr(),
                 0 if x_bn_coef is None else x_bn_coef.data_ptr())
    return (y, st) if stats else y


def gemm_nt_dy(dz, yp, coef, grads, w):
    """Data gradient straight behind a BatchNorm + ReLU backward: dz (R, K) bf16 = the gradient of relu(bn(yp)), yp (R, K) bf16 the
    BatchNorm's input, coef / grads (4, K) fp32 from pdm_bn_finalize_stats / pdm_bn_relu_backward_stats, w (N, K) bf16 ->
    (dx (R, N) bf16 = dy . w^T, dy (R, K) bf16) with dy = the BatchNorm + ReLU backward of dz, formed while dz and yp are read
    (bit for bit pdm_bn_relu_backward's dx)."""
    R, K = dz.shape
    N = w.shape[0]
    assert dz.dtype == torch.bfloat16 and yp.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and yp.shape == dz.shape
    assert dz.stride(1) == 1 and yp.stride(1) == 1 and w.stride(1) == 1 and coef.shape == (4, K) and grads.shape == (4, K)
    dx = torch.empty((R, N), dtype=torch.bfloat16, device=dz.device)
    dy = torch.empty((R, K), dtype=torch.bfloat16, device=dz.device)
    _native.call("pdm_tg_gemm_nt_dy", _stream(dz), R, K, N, dz.data_ptr(), dz.stride(0), yp.data_ptr(), yp.stride(0), w.data_ptr(),
                 w.stride(0), dx.data_ptr(), dx.stride(0), dy.data_ptr(), dy.stride(0), coef.data_ptr(), grads.data_ptr())
    return dx, dy


def wgrad(dy, x, out=None, accumulate=False, x_bn_coef=None):
    """dy (R, N) bf16, x (R, K) bf16 -> dW (N, K) fp32 = dy^T . x (fp32 accumulation, deterministic)."""
    R, N = dy.shape
    K = x.shape[1]
    assert dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and x.shape[0] == R and dy.stride(1) == 1 and x.stride(1) == 1
    dw = torch.empty((N, K), dtype=torch.float32, device=x.device) if out is None else out
    nbytes = _native.lib().pdm_tg_wgrad_ws_bytes(R, K, N)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x.device)
    _native.call("pdm_tg_wgrad", _stream(x), R, K, N, dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), dw.data_ptr(),
                 1 if accumulate else 0, ws.data_ptr(), nbytes, 0 if x_bn_coef is None else x_bn_coef.data_ptr())
    return dw


def colsum(y):
    """y (R, N) bf16 rows (N a multiple of 8, <= 512) -> (N,) fp32 column sums, fixed summation order (a bias gradient)."""
    R, N = y.shape
    assert y.dtype == torch.bfloat16 and y.stride(1) == 1 and N % 8 == 0 and N <= 512
    out = torch.empty((N,), dtype=torch.float32, device=y.device)
    ws = torch.empty((max(_native.lib().pdm_tg_colsum_ws_floats(R, N), 4),), dtype=torch.float32, device=y.device)
    _native.call("pdm_tg_colsum", _stream(y), R, N, y.data_ptr(), y.stride(0), out.data_ptr(), ws.data_ptr())
    return out
