"""Host side of the fused fp32-MFMA inference kernels (pdm_sa_mlp_fused / pdm_fp_mlp_fused).

Folds eval-mode BatchNorm into the 1x1-conv weights, pads every width to a multiple of 16 and packs
the weights in the per-lane order the kernels read (include/pdmssd_hip.h).  The packed form is cached
per module and rebuilt whenever a parameter or buffer changes (tensor version counters).
Semantics being fused: /root/reference/pcdet/ops/pointnet2/pointnet2_batch/pointnet2_modules.py:37-52
(SA scale) and :153-170 (FP module).
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn

from . import _native

# bench.py installs a dict here to tally the FLOPs each entry point executes (2 * real cin * cout per position and
# layer, padding excluded); None = no accounting
FLOP_COUNTER = None


def _count(name, positions, pk):
    if FLOP_COUNTER is not None:
        FLOP_COUNTER[name] = FLOP_COUNTER.get(name, 0.0) + float(positions) * pk.flops_per_position


def _pad16(c):
    return (c + 15) // 16 * 16


def fold_conv_bn(conv, bn):
    """(W' (Cout,Cin) fp64, shift (Cout) fp64) with y = relu(W' x + shift) == relu(bn(conv(x))) in eval mode."""
    w = conv.weight.detach().double().reshape(conv.out_channels, conv.in_channels).cpu()
    if conv.bias is not None:
        cb = conv.bias.detach().double().cpu()
    else:
        cb = torch.zeros(conv.out_channels, dtype=torch.float64)
    if bn is None:
        return w, cb
    s = bn.weight.detach().double().cpu() / torch.sqrt(bn.running_var.detach().double().cpu() + bn.eps)
    shift = bn.bias.detach().double().cpu() + (cb - bn.running_mean.detach().double().cpu()) * s
    return w * s[:, None], shift


def split_shared_mlp(seq):
    """nn.Sequential of [Conv, BatchNorm, ReLU]* -> list of (conv, bn); None if the pattern differs."""
    mods = list(seq)
    if len(mods) % 3 != 0 or not mods:
        return None
    out = []
    for i in range(0, len(mods), 3):
        conv, bn, act = mods[i], mods[i + 1], mods[i + 2]
        if not isinstance(conv, (nn.Conv1d, nn.Conv2d)) or not isinstance(bn, (nn.BatchNorm1d, nn.BatchNorm2d)) \
                or not isinstance(act, nn.ReLU):
            return None
        if any(k != 1 for k in conv.kernel_size) or conv.groups != 1:
            return None
        out.append((conv, bn))
    return out


def pack_layer(w, shift):
    """w (Cout,Cin) -> (packed float32 array [Cout_pad*Cin_pad], padded shift, Cin_pad, Cout_pad)."""
    cout, cin = w.shape
    cp, kp = _pad16(cout), _pad16(cin)
    wp = np.zeros((cp, kp), dtype=np.float32)
    wp[:cout, :cin] = w
    # [mb][oc][kb][g][s] -> [mb][kb][g][oc][s]; lane = g*16 + oc
    packed = wp.reshape(cp // 16, 16, kp // 16, 4, 4).transpose(0, 2, 3, 1, 4)
    bp = np.zeros(cp, dtype=np.float32)
    bp[:cout] = shift
    return np.ascontiguousarray(packed).reshape(-1), bp, kp, cp


class PackedMLP:
    def __init__(self, layers, device, in_perm=None, min_in=0):
        """layers: list of (conv, bn).  in_perm: optional input-channel selection / permutation of the first
        layer (a subset keeps only those columns; an empty one leaves `min_in` all-zero columns)."""
        ws, bs, dims = [], [], []
        self.flops_per_position = 0
        for li, (conv, bn) in enumerate(layers):
            w, shift = fold_conv_bn(conv, bn)
            if li == 0 and in_perm is not None:
                w = w[:, in_perm]
            self.flops_per_position += 2 * w.shape[0] * w.shape[1]
            if li == 0 and w.shape[1] < min_in:
                w = torch.cat([w, torch.zeros(w.shape[0], min_in - w.shape[1], dtype=w.dtype)], dim=1)
            pw, pb, kp, cp = pack_layer(w.numpy().astype(np.float32), shift.numpy().astype(np.float32))
            if li == 0:
                dims.append(kp)
            else:
                assert kp == dims[-1], "consecutive layer widths disagree"
            dims.append(cp)
            ws.append(pw)
            bs.append(pb)
        self.nlayers = len(layers)
        self.dims = dims
        self.cout = layers[-1][0].out_channels
        self.cin = layers[0][0].in_channels if in_perm is None else len(in_perm)
        self.dims_c = (ctypes.c_int * len(dims))(*dims)
        self.wpack = torch.from_numpy(np.concatenate(ws)).to(device)
        self.bias = torch.from_numpy(np.concatenate(bs)).to(device)

    @property
    def dims_ptr(self):
        return ctypes.cast(self.dims_c, ctypes.c_void_p)


class PackedPre:
    """First-layer column blocks of several MLPs stacked into one linear map: z = [W_a f | W_b f | ...] (no bias,
    no activation), each block padded to a multiple of 16 outputs.  offsets[i] = column of block i in a z row."""

    def __init__(self, first_layers, cols, device):
        blocks, self.offsets, o = [], [], 0
        self.flops_per_position = 0
        for conv, bn in first_layers:
            w, _ = fold_conv_bn(conv, bn)          # the shift stays with the consuming kernel
            w = w[:, cols].numpy().astype(np.float32)
            self.flops_per_position += 2 * w.shape[0] * w.shape[1]
            wp = np.zeros((_pad16(w.shape[0]), w.shape[1]), dtype=np.float32)
            wp[:w.shape[0]] = w
            blocks.append(wp)
            self.offsets.append(o)
            o += wp.shape[0]
        self.width = o
        self.cin = len(cols)
        pw, pb, kp, cp = pack_layer(np.concatenate(blocks, axis=0), np.zeros(o, dtype=np.float32))
        assert cp == o
        self.dims = [kp, cp]
        self.dims_c = (ctypes.c_int * 2)(kp, cp)
        self.wpack = torch.from_numpy(pw).to(device)
        self.bias = torch.from_numpy(pb).to(device)
        self.nlayers = 1
        self.cout = o

    @property
    def dims_ptr(self):
        return ctypes.cast(self.dims_c, ctypes.c_void_p)


def _state_key(module):
    return tuple((t.data_ptr(), t._version) for t in list(module.parameters()) + list(module.buffers()))


def cached_pack(owner, slot, seq, device, in_perm=None):
    """PackedMLP of `seq`, cached on `owner` under `slot`, rebuilt when any tensor of `seq` changed."""
    cache = owner.__dict__.setdefault('_pdm_fused_cache', {})
    key = (_state_key(seq), str(device))
    hit = cache.get(slot)
    if hit is not None and hit[0] == key:
        return hit[1]
    layers = split_shared_mlp(seq)
    if layers is None:
        cache[slot] = (key, None)
        return None
    packed = PackedMLP(layers, device, in_perm)
    cache[slot] = (key, packed)
    return packed


# ---------------------------------------------------------------------------------------------------------------------
# fp32 emulated on the bf16 matrix pipe (csrc/rows_chain_x3.hip): every fp32 value as three bf16 pieces.  Opt-in.

def _bf16_rne(x):
    """float32 array -> its bfloat16 rounding (round to nearest even) as a float32 array."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split_bf16x3(w):
    """w (float32) -> (hi, mid, lo) float32 arrays, each exactly representable in bfloat16, hi + mid + lo == w (fp32)."""
    w = np.ascontiguousarray(w, dtype=np.float32)
    hi = _bf16_rne(w)
    r1 = w - hi
    mid = _bf16_rne(r1)
    lo = _bf16_rne(r1 - mid)
    return hi, mid, lo


def _x3_fragments(wp):
    """wp (16 NMB, 32 NK) float32 -> uint16 [piece, mb, kb, lane, j]: lane = kg * 16 + row holds W[16 mb + row][32 kb + 16 (j >> 2) +
    4 kg + (j & 3)], the A operand of v_mfma_f32_16x16x32_bf16 whose B operand is the lane's own accumulator quads of output
    blocks 2 kb and 2 kb + 1 of the layer before."""
    M, K = wp.shape
    pieces = []
    for piece in split_bf16x3(wp):
        bits = (piece.view(np.uint32) >> 16).astype(np.uint16)                       # exact: the low 16 bits are zero
        f = bits.reshape(M // 16, 16, K // 32, 2, 4, 4).transpose(0, 2, 4, 1, 3, 5)  # [mb, row, kb, h, kg, i] -> [mb, kb, kg, row, h, i]
        pieces.append(np.ascontiguousarray(f).reshape(M // 16, K // 32, 64, 8))
    return np.stack(pieces)


class PackedMLPx3:
    """Weight stream of pdm_rows_mlp_x3 for a three-layer stack 128 -> 256 -> 256 -> <= 16 (BatchNorm folded): chunks of 24
    fragments (4 output blocks x 2 k-blocks x 3 pieces; the last layer's single output block x 8 k-blocks x 3 pieces is one
    chunk), in the order the kernel consumes them; biases fp32, padded widths, layers back to back."""

    def __init__(self, layers, device):
        assert len(layers) == 3, "PackedMLPx3: three layers"
        chunks, biases, dims = [], [], []
        self.flops_per_position = 0
        for li, (conv, bn) in enumerate(layers):
            w, shift = fold_conv_bn(conv, bn)
            w, shift = w.numpy().astype(np.float32), shift.numpy().astype(np.float32)
            self.flops_per_position += 2 * w.shape[0] * w.shape[1]
            kp = (w.shape[1] + 31) // 32 * 32
            cp = 16 if li == 2 else (w.shape[0] + 63) // 64 * 64
            wp = np.zeros((cp, kp), dtype=np.float32)
            wp[:w.shape[0], :w.shape[1]] = w
            bp = np.zeros(cp, dtype=np.float32)
            bp[:w.shape[0]] = shift
            if li == 0:
                dims.append(kp)
            else:
                assert kp == dims[-1], "consecutive layer widths disagree"
            dims.append(cp)
            fr = _x3_fragments(wp)                                   # [piece, mb, kb, 64, 8]
            nmb, nkb = fr.shape[1], fr.shape[2]
            if li < 2:
                c = fr.reshape(3, nmb // 4, 4, nkb // 2, 2, 64, 8).transpose(1, 3, 4, 2, 0, 5, 6)   # [mg, kg, kbi, i, piece, lane, j]
            else:
                assert nmb == 1 and nkb == 8
                c = fr[:, 0].transpose(1, 0, 2, 3)                   # [kb, piece, lane, j]
            chunks.append(np.ascontiguousarray(c).reshape(-1))
            biases.append(bp)
        assert dims == [128, 256, 256, 16], f"PackedMLPx3: only 128 -> 256 -> 256 -> <= 16 is instantiated, got {dims}"
        self.nlayers, self.dims = 3, dims
        self.cout, self.cin = layers[-1][0].out_channels, layers[0][0].in_channels
        self.dims_c = (ctypes.c_int * len(dims))(*dims)
        self.wstream = torch.from_numpy(np.concatenate(chunks).view(np.int16)).to(device)
        self.bias = torch.from_numpy(np.concatenate(biases)).to(device)

    @property
    def dims_ptr(self):
        return ctypes.cast(self.dims_c, ctypes.c_void_p)


def rows_forward_x3(pk, in_pm, out_pm, relu_last=False):
    """Per-row three-layer MLP with fp32 emulated by three bf16 pieces per operand (six partial products, fp32 accumulation)."""
    cin = in_pm.shape[-1]
    rows = in_pm.numel() // cin
    assert pk.cin == cin and in_pm.is_contiguous() and out_pm.is_contiguous() and in_pm.dtype == torch.float32
    _count(f"pdm_rows_mlp_x3[{pk.nlayers} layers, {cin} in, {rows} rows]", rows, pk)
    _native.call("pdm_rows_mlp_x3", _stream(in_pm), rows, cin, in_pm.data_ptr(), pk.nlayers, pk.dims_ptr, pk.wstream.data_ptr(),
                 pk.wstream.numel() * 2, pk.bias.data_ptr(), 1 if relu_last else 0, out_pm.data_ptr(), out_pm.shape[-1], pk.cout)


def cached_layers_x3(owner, slot, key_module, layers_fn, device):
    cache = owner.__dict__.setdefault('_pdm_fused_cache', {})
    key = (_state_key(key_module), str(device))
    hit = cache.get(slot)
    if hit is not None and hit[0] == key:
        return hit[1]
    packed = PackedMLPx3(layers_fn(), device)
    cache[slot] = (key, packed)
    return packed


def cached_layers(owner, slot, key_module, layers_fn, device):
    """PackedMLP of layers_fn() (a list of (conv, bn|None)), cached on `owner` under `slot` and rebuilt when any
    tensor of `key_module` changed."""
    cache = owner.__dict__.setdefault('_pdm_fused_cache', {})
    key = (_state_key(key_module), str(device))
    hit = cache.get(slot)
    if hit is not None and hit[0] == key:
        return hit[1]
    packed = PackedMLP(layers_fn(), device)
    cache[slot] = (key, packed)
    return packed


def cached_pre_packs(owner, slot, seqs, device, pre_cols, keep_cols, min_in=0):
    """(PackedPre over the first layers of `seqs` restricted to `pre_cols`, [PackedMLP of each seq with only
    `keep_cols` left in its first layer]); cached on `owner`; None when a seq is not a [conv, bn, relu]* chain."""
    cache = owner.__dict__.setdefault('_pdm_fused_cache', {})
    key = (tuple(_state_key(seq) for seq in seqs), str(device), tuple(pre_cols), tuple(keep_cols))
    hit = cache.get(slot)
    if hit is not None and hit[0] == key:
        return hit[1]
    layer_lists = [split_shared_mlp(seq) for seq in seqs]
    if any(l is None for l in layer_lists):
        cache[slot] = (key, None)
        return None
    pre = PackedPre([l[0] for l in layer_lists], list(pre_cols), device)
    packs = [PackedMLP(l, device, list(keep_cols), min_in) for l in layer_lists]
    cache[slot] = (key, (pre, packs))
    return pre, packs


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def sa_scale_forward(pk, xyz, new_xyz, feat_pm, idx, out_pm, out_coff):
    """One SA scale: xyz (B,N,3), new_xyz (B,M,3), feat_pm (B,N,Cin)|None, idx (B,M,ns) int32 ->
    out_pm[:, :, out_coff:out_coff+pk.cout] (B,M,stride)."""
    B, N, _ = xyz.shape
    M, ns = idx.shape[1], idx.shape[2]
    cin = 0 if feat_pm is None else feat_pm.shape[2]
    assert pk.cin == cin + 3
    _count("pdm_sa_mlp_fused", B * M * ns, pk)
    _native.call("pdm_sa_mlp_fused", _stream(xyz), B, N, M, cin, ns, xyz.data_ptr(), new_xyz.data_ptr(),
                 0 if feat_pm is None else feat_pm.data_ptr(), idx.data_ptr(), pk.nlayers, pk.dims_ptr,
                 pk.wpack.data_ptr(), pk.bias.data_ptr(), out_pm.data_ptr(), out_pm.shape[2], out_coff, pk.cout)


def fp_forward(pk, known_pm, skip_pm, idx, weight, out_pm):
    """One FP module: known_pm (B,m,C2), skip_pm (B,n,C1)|None, idx/weight (B,n,3) -> out_pm (B,n,Cout)."""
    B, m, ck = known_pm.shape
    n = idx.shape[1]
    cs = 0 if skip_pm is None else skip_pm.shape[2]
    assert pk.cin == ck + cs
    _count("pdm_fp_mlp_fused", B * n, pk)
    _native.call("pdm_fp_mlp_fused", _stream(known_pm), B, n, m, ck, cs, known_pm.data_ptr(),
                 0 if skip_pm is None else skip_pm.data_ptr(), idx.data_ptr(), weight.data_ptr(), pk.nlayers,
                 pk.dims_ptr, pk.wpack.data_ptr(), pk.bias.data_ptr(), out_pm.data_ptr(), out_pm.shape[2], pk.cout)


def rows_forward(pk, in_pm, out_pm, relu_last=True):
    """Per-row MLP: in_pm (..., cin) contiguous rows -> out_pm (..., stride), pk.cout columns written."""
    cin = in_pm.shape[-1]
    rows = in_pm.numel() // cin
    assert pk.cin == cin and in_pm.is_contiguous() and out_pm.is_contiguous()
    _count(f"pdm_rows_mlp_fused[{pk.nlayers} layers, {cin} in, {rows} rows]", rows, pk)   # bench.py keeps the shapes apart
    _native.call("pdm_rows_mlp_fused", _stream(in_pm), rows, cin, in_pm.data_ptr(), pk.nlayers, pk.dims_ptr,
                 pk.wpack.data_ptr(), pk.bias.data_ptr(), 1 if relu_last else 0, out_pm.data_ptr(),
                 out_pm.shape[-1], pk.cout)


def rows_forward_pair(pk_a, pk_b, in_pm, out_a, out_b, relu_last=True):
    """Two per-row MLPs of equal widths over the same rows (the point head's class and box stacks): one launch where
    rows_chain.hip has an instantiation, two otherwise — bit-identical to two rows_forward calls either way."""
    cin = in_pm.shape[-1]
    rows = in_pm.numel() // cin
    assert pk_a.cin == cin == pk_b.cin and in_pm.is_contiguous() and out_a.is_contiguous() and out_b.is_contiguous()
    assert pk_a.nlayers == pk_b.nlayers and list(pk_a.dims) == list(pk_b.dims), "rows_forward_pair: the two stacks must have equal (padded) widths"
    tag = f"pdm_rows_mlp_fused_pair[{pk_a.nlayers} layers, {cin} in, {rows} rows]"    # bench.py's OpTimer forms the same name
    _count(tag, rows, pk_a)
    _count(tag, rows, pk_b)
    _native.call("pdm_rows_mlp_fused_pair", _stream(in_pm), rows, cin, in_pm.data_ptr(), pk_a.nlayers, pk_a.dims_ptr,
                 pk_a.wpack.data_ptr(), pk_a.bias.data_ptr(), pk_b.wpack.data_ptr(), pk_b.bias.data_ptr(), 1 if relu_last else 0,
                 out_a.data_ptr(), out_a.shape[-1], pk_a.cout, out_b.data_ptr(), out_b.shape[-1], pk_b.cout)


class DeferredFP:
    """The backbone's last FP module, prepared but not yet run: its output rows `out_pm` (B, n, C) are allocated and handed on
    as point_features, and are filled either together with the point head's stacks (fp_head_forward: one launch) or by
    materialize().  Whoever asked the backbone to defer (batch_dict['defer_last_fp']) owns that call."""

    def __init__(self, pk, z, skip_pm, idx, weight, out_pm):
        self.pk, self.z, self.skip_pm, self.idx, self.weight, self.out_pm = pk, z, skip_pm, idx, weight, out_pm
        self.done = False

    def materialize(self):
        if not self.done:
            fp_forward_pre(self.pk, self.z, self.skip_pm, self.idx, self.weight, self.out_pm)
            self.done = True
        return self.out_pm

    def fits_head(self, pk_a, pk_b):
        B, n, _ = self.out_pm.shape
        cs = 0 if self.skip_pm is None else self.skip_pm.shape[2]
        return (not self.done and cs <= 4 and list(self.pk.dims) == [16, 128, 128] and self.pk.cout == 128 and B * n >= 32768
                and list(pk_a.dims) == [128, 256, 256, 16] == list(pk_b.dims) and self.z.shape[2] % 4 == 0 and self.z.shape[2] >= 128)


def fp_head_forward(d, pk_a, pk_b, out_a, out_b, relu_last=False):
    """pdm_fp_head_fused: the deferred FP module `d` and the two per-row stacks over its output in one launch."""
    assert d.fits_head(pk_a, pk_b) and out_a.is_contiguous() and out_b.is_contiguous()
    B, m, _ = d.z.shape
    n = d.idx.shape[1]
    cs = 0 if d.skip_pm is None else d.skip_pm.shape[2]
    rows = B * n
    _count("pdm_fp_mlp_fused_pre", rows, d.pk)
    tag = f"pdm_rows_mlp_fused_pair[{pk_a.nlayers} layers, {pk_a.cin} in, {rows} rows]"
    _count(tag, rows, pk_a)
    _count(tag, rows, pk_b)
    _native.call("pdm_fp_head_fused", _stream(d.z), B, n, m, cs, d.z.data_ptr(), d.z.shape[2],
                 0 if d.skip_pm is None else d.skip_pm.data_ptr(), d.idx.data_ptr(), d.weight.data_ptr(), d.pk.dims_ptr,
                 d.pk.wpack.data_ptr(), d.pk.bias.data_ptr(), d.out_pm.data_ptr(), d.out_pm.shape[2], d.pk.cout, pk_a.dims_ptr,
                 pk_a.wpack.data_ptr(), pk_a.bias.data_ptr(), pk_b.wpack.data_ptr(), pk_b.bias.data_ptr(), 1 if relu_last else 0,
                 out_a.data_ptr(), out_a.shape[-1], pk_a.cout, out_b.data_ptr(), out_b.shape[-1], pk_b.cout)
    d.done = True


def sa_scale_forward_pre(pk, xyz, new_xyz, z, z_coff, idx, out_pm, out_coff):
    """SA scale whose first layer's feature part was applied to the source points: z (B,N,width)."""
    B, N, _ = xyz.shape
    M, ns = idx.shape[1], idx.shape[2]
    assert pk.cin == 3 and z.shape[0] == B and z.shape[1] == N
    _count("pdm_sa_mlp_fused_pre", B * M * ns, pk)
    _native.call("pdm_sa_mlp_fused_pre", _stream(xyz), B, N, M, ns, xyz.data_ptr(), new_xyz.data_ptr(), z.data_ptr(),
                 z.shape[2], z_coff, idx.data_ptr(), pk.nlayers, pk.dims_ptr, pk.wpack.data_ptr(), pk.bias.data_ptr(),
                 out_pm.data_ptr(), out_pm.shape[2], out_coff, pk.cout)


def sa_pack(idx, n):
    """Compacted neighbour list of one SA scale (csrc/sa_pack.hip): idx (B,M,ns) int32 with ns in (16, 32), n = points
    per source cloud -> (pack (rows,2) int32, meta (8,) int32).  Padding copies of the first hit are dropped, so the
    fused kernels run the shared MLP over the distinct neighbours only (bit-identical pooled result)."""
    B, M, ns = idx.shape
    assert idx.dtype == torch.int32 and idx.is_contiguous() and ns in (16, 32)
    l = _native.lib()
    rows = l.pdm_sa_pack_rows(B, M, ns)
    pack = torch.empty((rows, 2), dtype=torch.int32, device=idx.device)
    meta = torch.empty((8,), dtype=torch.int32, device=idx.device)
    ws_bytes = l.pdm_sa_pack_workspace_bytes(B, M)
    ws = torch.empty((max(ws_bytes, 16),), dtype=torch.uint8, device=idx.device)
    _native.call("pdm_sa_pack", _stream(idx), B, n, M, ns, idx.data_ptr(), ws.data_ptr(), ws_bytes, pack.data_ptr(),
                 meta.data_ptr())
    return pack, meta


def sa_pack_pair(idx0, idx1, n):
    """sa_pack of the two scales of an MSG level in one count -> scan -> fill sequence: [(pack, meta), (pack, meta)]."""
    assert idx0.shape[:2] == idx1.shape[:2] and all(i.dtype == torch.int32 and i.is_contiguous() and i.shape[2] in (16, 32) for i in (idx0, idx1))
    B, M = idx0.shape[:2]
    l = _native.lib()
    ws_bytes = l.pdm_sa_pack_workspace_bytes(B, M)
    outs, keep = [], []
    for idx in (idx0, idx1):
        rows = l.pdm_sa_pack_rows(B, M, idx.shape[2])
        outs.append((torch.empty((rows, 2), dtype=torch.int32, device=idx.device), torch.empty((8,), dtype=torch.int32, device=idx.device),
                     torch.empty((max(ws_bytes, 16),), dtype=torch.uint8, device=idx.device)))
    I2, P2 = ctypes.c_int * 2, ctypes.c_void_p * 2
    keep = [I2(idx0.shape[2], idx1.shape[2]), P2(idx0.data_ptr(), idx1.data_ptr()), P2(outs[0][2].data_ptr(), outs[1][2].data_ptr()),
            P2(outs[0][0].data_ptr(), outs[1][0].data_ptr()), P2(outs[0][1].data_ptr(), outs[1][1].data_ptr())]
    c = [ctypes.cast(k, ctypes.c_void_p) for k in keep]
    _native.call("pdm_sa_pack_pair", _stream(idx0), B, n, M, c[0], c[1], c[2], ws_bytes, c[3], c[4])
    return [(outs[0][0], outs[0][1]), (outs[1][0], outs[1][1])]


def sa_scale_forward_packed(pk, xyz, new_xyz, feat_pm, z, z_coff, packed, ns, out_pm, out_coff):
    """One SA scale over a compacted neighbour list `packed` = sa_pack(idx, N); z None = unhoisted form (feat_pm rows),
    else the hoisted form of sa_scale_forward_pre."""
    pack, meta = packed
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    cin = 0 if (feat_pm is None or z is not None) else feat_pm.shape[2]
    assert pk.cin == cin + 3
    if FLOP_COUNTER is not None:
        _count("pdm_sa_mlp_packed", int(meta[6].item()), pk)   # rows actually run (a sync: accounting passes only)
    _native.call("pdm_sa_mlp_packed", _stream(xyz), B, N, M, cin, ns, xyz.data_ptr(), new_xyz.data_ptr(),
                 0 if cin == 0 else feat_pm.data_ptr(), 0 if z is None else z.data_ptr(),
                 0 if z is None else z.shape[2], z_coff, pack.data_ptr(), meta.data_ptr(), pk.nlayers, pk.dims_ptr,
                 pk.wpack.data_ptr(), pk.bias.data_ptr(), out_pm.data_ptr(), out_pm.shape[2], out_coff, pk.cout)


def sa_level_forward_packed(pks, xyz, new_xyz, feat_pm, z, z_coffs, packeds, nss, out_pm, out_coffs):
    """Both scales of an MSG level over their compacted neighbour lists in one launch (pdm_sa_mlp_packed_pair; two launches
    inside the library when the scales do not map to the same kernel): arguments per scale as lists of two."""
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    cin = 0 if (feat_pm is None or z is not None) else feat_pm.shape[2]
    assert len(pks) == 2 and all(pk.cin == cin + 3 for pk in pks)
    if FLOP_COUNTER is not None:
        for pk, (pack, meta) in zip(pks, packeds):
            _count("pdm_sa_mlp_packed_pair", int(meta[6].item()), pk)   # rows actually run (a sync: accounting passes only)
    I2, P2 = ctypes.c_int * 2, ctypes.c_void_p * 2
    keep = [I2(*nss), I2(*z_coffs), P2(*[p.data_ptr() for p, _ in packeds]), P2(*[m.data_ptr() for _, m in packeds]),
            I2(*[pk.nlayers for pk in pks]), P2(*[pk.dims_ptr.value for pk in pks]), P2(*[pk.wpack.data_ptr() for pk in pks]),
            P2(*[pk.bias.data_ptr() for pk in pks]), I2(*out_coffs), I2(*[pk.cout for pk in pks])]
    c = [ctypes.cast(k, ctypes.c_void_p) for k in keep]
    _native.call("pdm_sa_mlp_packed_pair", _stream(xyz), B, N, M, cin, c[0], xyz.data_ptr(), new_xyz.data_ptr(),
                 0 if cin == 0 else feat_pm.data_ptr(), 0 if z is None else z.data_ptr(), 0 if z is None else z.shape[2], c[1], c[2], c[3],
                 c[4], c[5], c[6], c[7], out_pm.data_ptr(), out_pm.shape[2], c[8], c[9])


def fp_forward_pre(pk, z, skip_pm, idx, weight, out_pm):
    """FP module whose first layer's known-feature part was applied to the known points: z (B,m,width)."""
    B, m, _ = z.shape
    n = idx.shape[1]
    cs = 0 if skip_pm is None else skip_pm.shape[2]
    assert pk.cin == cs
    _count("pdm_fp_mlp_fused_pre", B * n, pk)
    _native.call("pdm_fp_mlp_fused_pre", _stream(z), B, n, m, cs, z.data_ptr(), z.shape[2],
                 0 if skip_pm is None else skip_pm.data_ptr(), idx.data_ptr(), weight.data_ptr(), pk.nlayers,
                 pk.dims_ptr, pk.wpack.data_ptr(), pk.bias.data_ptr(), out_pm.data_ptr(), out_pm.shape[2], pk.cout)
