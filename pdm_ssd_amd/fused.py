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
    def __init__(self, layers, device, in_perm=None):
        """layers: list of (conv, bn).  in_perm: optional input-channel permutation of the first layer."""
        ws, bs, dims = [], [], []
        for li, (conv, bn) in enumerate(layers):
            w, shift = fold_conv_bn(conv, bn)
            if li == 0 and in_perm is not None:
                w = w[:, in_perm]
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
        self.cin = layers[0][0].in_channels
        self.dims_c = (ctypes.c_int * len(dims))(*dims)
        self.wpack = torch.from_numpy(np.concatenate(ws)).to(device)
        self.bias = torch.from_numpy(np.concatenate(bs)).to(device)

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


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def sa_scale_forward(pk, xyz, new_xyz, feat_pm, idx, out_pm, out_coff):
    """One SA scale: xyz (B,N,3), new_xyz (B,M,3), feat_pm (B,N,Cin)|None, idx (B,M,ns) int32 ->
    out_pm[:, :, out_coff:out_coff+pk.cout] (B,M,stride)."""
    B, N, _ = xyz.shape
    M, ns = idx.shape[1], idx.shape[2]
    cin = 0 if feat_pm is None else feat_pm.shape[2]
    assert pk.cin == cin + 3
    _native.call("pdm_sa_mlp_fused", _stream(xyz), B, N, M, cin, ns, xyz.data_ptr(), new_xyz.data_ptr(),
                 0 if feat_pm is None else feat_pm.data_ptr(), idx.data_ptr(), pk.nlayers, pk.dims_ptr,
                 pk.wpack.data_ptr(), pk.bias.data_ptr(), out_pm.data_ptr(), out_pm.shape[2], out_coff, pk.cout)


def fp_forward(pk, known_pm, skip_pm, idx, weight, out_pm):
    """One FP module: known_pm (B,m,C2), skip_pm (B,n,C1)|None, idx/weight (B,n,3) -> out_pm (B,n,Cout)."""
    B, m, ck = known_pm.shape
    n = idx.shape[1]
    cs = 0 if skip_pm is None else skip_pm.shape[2]
    assert pk.cin == ck + cs
    _native.call("pdm_fp_mlp_fused", _stream(known_pm), B, n, m, ck, cs, known_pm.data_ptr(),
                 0 if skip_pm is None else skip_pm.data_ptr(), idx.data_ptr(), weight.data_ptr(), pk.nlayers,
                 pk.dims_ptr, pk.wpack.data_ptr(), pk.bias.data_ptr(), out_pm.data_ptr(), out_pm.shape[2], pk.cout)
