"""Layout helpers shared by the tests (NCHW <-> packed multi-level NHWC, OIHW <-> KRSC)."""
import torch


def pack_levels(levels_nchw, dtype, cpad=None):
    """list of (B,C,H,W) fp32 -> (rows, Cpad) packed level-major NHWC tensor."""
    outs = []
    for x in levels_nchw:
        b, c, h, w = x.shape
        cp = cpad or c
        t = torch.zeros(b, h, w, cp, dtype=torch.float32)
        t[..., :c] = x.permute(0, 2, 3, 1)
        outs.append(t.reshape(b * h * w, cp))
    return torch.cat(outs, 0).to(dtype).contiguous()


def unpack_levels(packed, batch, shapes, c=None):
    """(rows, C) -> list of (B,C,H,W) fp32 for shapes [(H,W),...]."""
    outs, r = [], 0
    for (h, w) in shapes:
        n = batch * h * w
        t = packed[r:r + n].float().reshape(batch, h, w, -1).permute(0, 3, 1, 2)
        outs.append(t[:, :c] if c else t)
        r += n
    return outs


def w_to_krsc(w_oihw, dtype, cin_pad=None, cout_pad=None):
    o, i, kh, kw = w_oihw.shape
    ip, op = cin_pad or i, cout_pad or o
    t = torch.zeros(op, kh, kw, ip, dtype=torch.float32)
    t[:o, :, :, :i] = w_oihw.permute(0, 2, 3, 1)
    return t.to(dtype).contiguous()


def w_to_dgrad(w_oihw, dtype, cin_pad=None, cout_pad=None):
    """wt[ci][ky][kx][co]"""
    o, i, kh, kw = w_oihw.shape
    ip, op = cin_pad or i, cout_pad or o
    t = torch.zeros(ip, kh, kw, op, dtype=torch.float32)
    t[:i, :, :, :o] = w_oihw.permute(1, 2, 3, 0)
    return t.to(dtype).contiguous()


def round_to(x, dtype):
    return x.to(dtype).float()
