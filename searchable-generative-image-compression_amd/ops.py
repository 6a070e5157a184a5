"""Thin Python wrappers over the transform kernels of libsgic (C ABI, include/sgic.h).  Tensors are torch
CUDA tensors used purely as device-memory handles; all arithmetic happens in the HIP kernels."""
import ctypes

import torch

from ._lib import call, require_gpu

ACT_NONE, ACT_GELU, ACT_SILU, ACT_TANH, ACT_LRELU = 0, 1, 2, 3, 4

# Optional live profiling of the dominant kernel (bench.py): when PROFILE is a list, every gemm() launch is
# bracketed by HIP events on the launch stream and (flops, start, end) is appended.
PROFILE = None


def _rows(t):
    """a 2-D fp32 CUDA view whose last dim is contiguous -> (tensor, leading dimension)"""
    assert t.dim() == 2 and t.stride(1) == 1 and t.dtype == torch.float32 and t.is_cuda, (t.shape, t.stride(), t.dtype)
    return t, t.stride(0)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _cl(v):
    return ctypes.c_long(int(v))


def empty(*shape, like=None, dtype=torch.float32, device=None):
    return torch.empty(*shape, dtype=dtype, device=like.device if like is not None else device)


def gemm(a, w, bias=None, residual=None, act=ACT_NONE, out=None, M=None, a_seg=(0, 0), c_seg=(0, 0)):
    """out[M,N] = act(a[M,K] @ w[N,K]^T + bias) + residual.  a_seg/c_seg = (seg, seg_stride) row maps:
    logical row m of A (resp. C) lives at physical row (m // seg) * seg_stride + m % seg."""
    require_gpu()
    a, lda = _rows(a)
    w, ldw = _rows(w)
    K = a.shape[1]
    if M is None:
        M = a.shape[0]
    N = w.shape[0]
    assert w.shape[1] == K, (a.shape, w.shape)
    if out is None:
        assert c_seg[0] == 0
        out = torch.empty(M, N, device=a.device, dtype=torch.float32)
    out, ldc = _rows(out)
    assert out.shape[1] == N
    ldr = 0
    if residual is not None:
        residual, ldr = _rows(residual)
        assert residual.shape[1] == N and residual.shape[0] >= M
    if bias is not None:
        assert bias.shape == (N,) and bias.is_contiguous()
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    call("sgic_gemm_f32", _p(a), lda, _p(w), ldw, _p(bias), _p(residual), ldr, _p(out), ldc, M, N, K, act,
         a_seg[0], a_seg[1], c_seg[0], c_seg[1])
    if PROFILE is not None:
        e1.record()
        PROFILE.append((2.0 * M * N * K, e0, e1))
    return out


def layernorm(x, gamma, beta, out=None, eps=1e-5, act=ACT_NONE, M=None, x_seg=(0, 0), y_seg=(0, 0)):
    x, ldx = _rows(x)
    C = x.shape[1]
    if M is None:
        M = x.shape[0]
    if out is None:
        assert y_seg[0] == 0
        out = torch.empty(M, C, device=x.device, dtype=torch.float32)
    out, ldy = _rows(out)
    assert gamma.shape == (C,) and beta.shape == (C,)
    call("sgic_layernorm_f32", _p(x), ldx, x_seg[0], x_seg[1], _p(gamma), _p(beta), _p(out), ldy, y_seg[0], y_seg[1],
         M, C, float(eps), act)
    return out


def attention(q, k, v, out, L, nseq, nheads, rowmap=None, bias=None, biasvar=None, scale=0.125):
    """q,k,v,out: 2-D row-strided views (rows x nheads*64)."""
    q, ldq = _rows(q)
    k, ldk = _rows(k)
    v, ldv = _rows(v)
    out, ldo = _rows(out)
    if rowmap is not None:
        assert rowmap.dtype == torch.int32 and rowmap.numel() == nseq * L and rowmap.is_contiguous()
    if bias is not None:
        assert bias.dim() == 3 and bias.shape[1] == L and bias.shape[2] == L and bias.is_contiguous()
    if biasvar is not None:
        assert biasvar.dtype == torch.int32 and biasvar.numel() == nseq
    call("sgic_attention_f32", _p(q), ldq, _p(k), ldk, _p(v), ldv, _p(out), ldo, L, nseq, nheads, _p(rowmap), _p(bias),
         _p(biasvar), float(scale))
    return out


def im2col_patch(x, P, mul=1.0, add=0.0, tile16=False, out=None):
    assert x.dim() == 4 and x.is_contiguous() and x.dtype == torch.float32
    B, C, H, W = x.shape
    if out is None:
        out = torch.empty(B * (H // P) * (W // P), C * P * P, device=x.device, dtype=torch.float32)
    call("sgic_im2col_patch", _p(x), B, C, H, W, P, float(mul), float(add), int(tile16), _p(out))
    return out


def assemble_tokens(emb, cls, pos, lat, latpos, N, P, T, D, out=None):
    if out is None:
        out = torch.empty(N * (1 + P + T), D, device=emb.device, dtype=torch.float32)
    call("sgic_assemble_tokens", _p(emb), _p(cls), _p(pos), _p(lat), _p(latpos), N, P, T, D, _p(out))
    return out


def add_rows_bcast(inp, iseg, vec, out, oseg, Nn, Lr):
    inp, ldi = _rows(inp)
    out, ldo = _rows(out)
    D = inp.shape[1]
    assert out.shape[1] == D
    if vec is not None:
        assert vec.is_contiguous() and vec.numel() == Lr * D
    call("sgic_add_rows_bcast", _p(inp), ldi, iseg, _p(vec), _p(out), ldo, oseg, Nn, Lr, D)
    return out


def dwconv(x, w_kkc, bias, prescale, B, H, W, k, tile16=False, out=None):
    """x: (B*H*W, C) rows; w_kkc: (k*k, C)"""
    x, ldx = _rows(x)
    C = x.shape[1]
    assert ldx == C and w_kkc.shape == (k * k, C) and w_kkc.is_contiguous()
    if out is None:
        out = torch.empty_like(x)
    call("sgic_dwconv_nhwc", _p(x), _p(w_kkc), _p(bias), _p(prescale), _p(out), B, H, W, C, k, int(tile16))
    return out


def im2col_2x2(x, B, H, W, tile16=False, out=None):
    x, ldx = _rows(x)
    C = x.shape[1]
    assert ldx == C
    if out is None:
        out = torch.empty(B * (H // 2) * (W // 2), 4 * C, device=x.device, dtype=torch.float32)
    call("sgic_im2col_2x2", _p(x), B, H, W, C, int(tile16), _p(out))
    return out


def gated_lrelu(x, out=None):
    x, ldx = _rows(x)
    M, C2x2 = x.shape
    assert ldx == C2x2
    if out is None:
        out = torch.empty(M, C2x2 // 2, device=x.device, dtype=torch.float32)
    call("sgic_gated_lrelu", _p(x), _p(out), M, C2x2 // 2)
    return out


def colop(x, v, mode, out=None):
    """mode 0: x * v ; mode 1: x / max(v, 0.5);  v: (vrows, C) broadcast with row m % vrows"""
    x, ldx = _rows(x)
    v, ldv = _rows(v)
    M, C = x.shape
    if out is None:
        out = torch.empty(M, C, device=x.device, dtype=torch.float32)
    out, ldy = _rows(out)
    call("sgic_colop", _p(x), ldx, _p(v), ldv, v.shape[0], _p(out), ldy, M, C, mode)
    return out


def fake2d_transpose(x_base, seq_stride, N, T, D):
    out = torch.empty(N * T, D, device=x_base.device, dtype=torch.float32)
    call("sgic_fake2d_transpose", _p(x_base), _cl(seq_stride), _p(out), N, T, D)
    return out


def vq_argmin(z, codebook, l2norm=True):
    z, ldz = _rows(z)
    M, dim = z.shape
    assert codebook.is_contiguous() and codebook.shape[1] == dim
    idx = torch.empty(M, dtype=torch.int32, device=z.device)
    call("sgic_vq_argmin", _p(z), ldz, _p(codebook), codebook.shape[0], dim, M, int(l2norm), _p(idx))
    return idx


def l2norm_u8(x):
    x, ldx = _rows(x)
    M, D = x.shape
    unit = torch.empty(M, D, device=x.device, dtype=torch.float32)
    q = torch.empty(M, D, device=x.device, dtype=torch.uint8)
    call("sgic_l2norm_u8", _p(x), ldx, M, D, _p(unit), _p(q))
    return unit, q


# ---- entropy-side kernels (device-resident batch API) ----
def quant_step(y, scales, means, ld_sm, yhat, ld_yhat, B, H, W, C, k, thr, sym, idx):
    call("sgic_quant_step", _p(y), _p(scales), _p(means), ld_sm, _p(yhat), ld_yhat, B, H, W, C, k,
         float(-1.0 if thr is None else thr), _p(sym), _p(idx))


def index_step(scales, ld_sm, B, H, W, C, k, thr, idx):
    call("sgic_index_step", _p(scales), ld_sm, B, H, W, C, k, float(-1.0 if thr is None else thr), _p(idx))


def dequant_step(sym, means, ld_sm, yhat, ld_yhat, B, H, W, C, k):
    call("sgic_dequant_step", _p(sym), _p(means), ld_sm, _p(yhat), ld_yhat, B, H, W, C, k)


def rans_encode_batch(table, sym, idx, B, n, cap=None):
    """-> (out (B,cap) u8, off (B,) i32, len (B,) i32, err (B,) i32); streams are end-aligned in their slot"""
    if cap is None:
        cap = 2 * n + 64
    dev = sym.device
    out = torch.empty(B, cap, dtype=torch.uint8, device=dev)
    meta = torch.empty(3, B, dtype=torch.int32, device=dev)
    call("sgic_rans_encode_batch", table, _p(sym), _p(idx), B, n, _p(out), cap, _p(meta[0]), _p(meta[1]), _p(meta[2]))
    return out, meta


def pack12_batch(idx_i32, B, n):
    from ._lib import lib
    nb = lib.sgic_pack12_size(n)
    out = torch.empty(B, nb, dtype=torch.uint8, device=idx_i32.device)
    call("sgic_pack12_batch", _p(idx_i32), B, n, _p(out))
    return out
