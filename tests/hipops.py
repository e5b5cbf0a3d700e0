"""Thin Python entry points to single HIP operators of libymk_hip.so (NHWC fp32 on device).

They exist so the parity tests can exercise exactly the kernels the models use, one at a time.
Inputs/outputs here are NCHW torch tensors on a HIP device; the layout change is done with torch.
"""

from __future__ import annotations

import torch

from yomitoku_amd import _lib

ACT = {"none": 0, "relu": 1, "silu": 2, "sigmoid": 3, "gelu": 4}


def _nhwc(x: torch.Tensor) -> torch.Tensor:
    return x.permute(0, 2, 3, 1).contiguous()


def _nchw(x: torch.Tensor) -> torch.Tensor:
    return x.permute(0, 3, 1, 2).contiguous()


def conv2d(x, weight, scale=None, bias=None, residual=None, stride=1, padding=0, dilation=1, act="none"):
    """y = act(scale * conv(x, weight) + bias + residual); x NCHW on device, weight OIHW (any device)."""
    lib = _lib.load()
    assert x.is_cuda
    n, c, h, w = x.shape
    cout, cin, kh, kw = weight.shape
    assert cin == c
    tap4 = c <= 4
    xh = _nhwc(x.float())
    if tap4 and c < 4:
        xh = torch.cat([xh, xh.new_zeros(n, h, w, 4 - c)], dim=-1).contiguous()
    cpad = xh.shape[-1]
    if not tap4 and cpad % 4:
        raise ValueError("channels must be a multiple of 4")
    oh = (h + 2 * padding - dilation * (kh - 1) - 1) // stride + 1
    ow = (w + 2 * padding - dilation * (kw - 1) - 1) // stride + 1
    y = torch.empty((n, oh, ow, cout), dtype=torch.float32, device=x.device)
    wh = weight.detach().float().cpu().contiguous()
    sh = scale.detach().float().cpu().contiguous() if scale is not None else None
    bh = bias.detach().float().cpu().contiguous() if bias is not None else None
    rh = _nhwc(residual.float()) if residual is not None else None
    with torch.cuda.device(x.device):
        _lib.check(
            lib.ymk_op_conv2d(
                xh.data_ptr(), n, h, w, cpad, wh.data_ptr(), cout, cin, kh, kw, _lib.ptr(sh), _lib.ptr(bh),
                _lib.ptr(rh), stride, padding, dilation, ACT[act], 1 if tap4 else 0, y.data_ptr(),
                _lib.current_stream_ptr(),
            ),
            "ymk_op_conv2d",
        )
    return _nchw(y)


def conv1x1_astat(x, weight, scale=None, bias=None, residual=None, act="none", reps=1, ln=None):
    """The A-stationary kernel at any row count (ymk_op_conv1x1_astat): x [M, C] on the device, weight [Cout, C] -> (y [M, Cout],
    ms of the last of `reps` launches).  ln = (gamma [C], beta [C], eps): LayerNorm folded into the operand load."""
    import ctypes

    lib = _lib.load()
    assert x.is_cuda and x.dim() == 2
    m, c = x.shape
    cout = weight.shape[0]
    xh = x.float().contiguous()
    wh = weight.detach().float().cpu().contiguous()
    sh = scale.detach().float().cpu().contiguous() if scale is not None else None
    bh = bias.detach().float().cpu().contiguous() if bias is not None else None
    rh = residual.float().contiguous() if residual is not None else None
    gh = ln[0].detach().float().cpu().contiguous() if ln is not None else None
    beh = ln[1].detach().float().cpu().contiguous() if ln is not None else None
    y = torch.empty((m, cout), dtype=torch.float32, device=x.device)
    ms = ctypes.c_float()
    with torch.cuda.device(x.device):
        _lib.check(lib.ymk_op_conv1x1_astat(xh.data_ptr(), m, c, wh.data_ptr(), cout, _lib.ptr(sh), _lib.ptr(bh), _lib.ptr(rh), ACT[act],
                                            _lib.ptr(gh), _lib.ptr(beh), float(ln[2]) if ln is not None else 0.0,
                                            y.data_ptr(), reps, ctypes.byref(ms), _lib.current_stream_ptr()), "ymk_op_conv1x1_astat")
    return y, float(ms.value)


def vit_mlp(x, gamma, beta, eps, w1, b1, w2, b2, reps=1):
    """ymk_op_vit_mlp: x [M, D] on the device -> (x + fc2(gelu(fc1(layer_norm(x)))) [M, D], ms of the last launch)."""
    import ctypes

    lib = _lib.load()
    m, d = x.shape
    f = w1.shape[0]
    xh = x.float().contiguous()
    host = [t.detach().float().cpu().contiguous() for t in (gamma, beta, w1, b1, w2, b2)]
    y = torch.empty_like(xh)
    ms = ctypes.c_float()
    with torch.cuda.device(x.device):
        _lib.check(lib.ymk_op_vit_mlp(xh.data_ptr(), m, d, f, host[0].data_ptr(), host[1].data_ptr(), float(eps), host[2].data_ptr(),
                                      host[3].data_ptr(), host[4].data_ptr(), host[5].data_ptr(), y.data_ptr(), reps, ctypes.byref(ms),
                                      _lib.current_stream_ptr()), "ymk_op_vit_mlp")
    return y, float(ms.value)


def maxpool3x3s2(x):
    lib = _lib.load()
    n, c, h, w = x.shape
    xh = _nhwc(x.float())
    y = torch.empty((n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, c), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(
            lib.ymk_op_maxpool3x3s2(xh.data_ptr(), n, h, w, c, y.data_ptr(), _lib.current_stream_ptr()),
            "ymk_op_maxpool3x3s2",
        )
    return _nchw(y)


def upsample_bilinear(x, size, add=None):
    lib = _lib.load()
    n, c, h, w = x.shape
    oh, ow = size
    xh = _nhwc(x.float())
    ah = _nhwc(add.float()) if add is not None else None
    y = torch.empty((n, oh, ow, c), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(
            lib.ymk_op_upsample_bilinear(
                xh.data_ptr(), n, h, w, c, oh, ow, _lib.ptr(ah), y.data_ptr(), _lib.current_stream_ptr()
            ),
            "ymk_op_upsample_bilinear",
        )
    return _nchw(y)


def layernorm(x, weight, bias, eps):
    lib = _lib.load()
    d = x.shape[-1]
    xs = x.float().contiguous()
    y = torch.empty_like(xs)
    g = weight.float().to(x.device).contiguous()
    b = bias.float().to(x.device).contiguous()
    with torch.cuda.device(x.device):
        _lib.check(
            lib.ymk_op_layernorm(xs.data_ptr(), xs.numel() // d, d, g.data_ptr(), b.data_ptr(), float(eps), y.data_ptr(),
                                 _lib.current_stream_ptr()),
            "ymk_op_layernorm",
        )
    return y


def attention(q, k, v, heads, scale=None, mask_qk=None, key_padding_mask=None, use_small=False):
    """q [B, Lq, D], k/v [B, Lk, D] on device -> [B, Lq, D]; masks are bool tensors (True = blocked)."""
    lib = _lib.load()
    b, lq, d = q.shape
    lk = k.shape[1]
    hd = d // heads
    scale = hd**-0.5 if scale is None else scale
    qs, ks, vs = q.float().contiguous(), k.float().contiguous(), v.float().contiguous()
    o = torch.empty_like(qs)
    m = mask_qk.to(device=q.device, dtype=torch.uint8).contiguous() if mask_qk is not None else None
    kp = key_padding_mask.to(device=q.device, dtype=torch.uint8).contiguous() if key_padding_mask is not None else None
    with torch.cuda.device(q.device):
        _lib.check(
            lib.ymk_op_attention(qs.data_ptr(), ks.data_ptr(), vs.data_ptr(), o.data_ptr(), b, heads, lq, lk, hd,
                                 float(scale), _lib.ptr(m), _lib.ptr(kp), 1 if use_small else 0,
                                 _lib.current_stream_ptr()),
            "ymk_op_attention",
        )
    return o
