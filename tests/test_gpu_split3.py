"""GPU tests of the bf16x3 split GEMM (csrc/gemm_split.hip, through the C ABI): an fp32 GEMM in accuracy, computed on the
bf16 matrix pipe.
  * the split is exact: p0 + p1 + p2 == x for every fp32 x (checked in fp64);
  * error against an fp64 reference is bounded by the error of the plain fp32 path (sgic_gemm_f32: a k-ordered fmaf chain)
    on the same inputs -- the tolerance that makes this "fp32" rather than "reduced precision";
  * results are bitwise independent of the tile mode and of M (batch invariance), like the fp32 kernel;
  * epilogue variants (bias / activation / residual / in-place residual / row maps / ragged M, N) against torch fp64."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import sgic_amd  # noqa
    from sgic_amd import ops as O
    return O


def _bf16_bits_to_f64(t):
    return (t.to(torch.int32) << 16).view(torch.float32).double()


def test_split_is_exact(ops):
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(257, 1024, device="cuda", generator=g) * torch.exp(4 * torch.randn(257, 1, device="cuda", generator=g))
    x[0, :8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 3.0e-39, 65504.0, 1e30, -1e-30], device="cuda")   # incl. an fp32 subnormal
    p = ops.split3(x)
    torch.cuda.synchronize()
    s = _bf16_bits_to_f64(p[0]) + _bf16_bits_to_f64(p[1]) + _bf16_bits_to_f64(p[2])
    d = (s - x.double()).abs()
    big = x.abs() > 1e-30            # below that the low pieces are bf16 subnormals (may flush): far under any fp32 ulp that matters
    assert float(d[big].max()) == 0.0
    assert float(d[~big].max()) <= 1e-30


@pytest.mark.parametrize("M,N,K", [(256, 256, 4096), (512, 1024, 1024), (289, 768, 768)])
def test_error_not_above_fp32_path(ops, M, N, K):
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) * 0.03
    ref = a.double() @ w.double().T
    c3 = ops.gemm(a, w, precision="split3")
    c1 = ops.gemm(a, w, precision="f32")
    torch.cuda.synchronize()
    e3 = float((c3.double() - ref).pow(2).mean().sqrt())
    e1 = float((c1.double() - ref).pow(2).mean().sqrt())
    scale = float(ref.pow(2).mean().sqrt())
    print(f"\n({M},{N},{K}): rms error / rms(C): split3 {e3 / scale:.3e}   fp32 MFMA chain {e1 / scale:.3e}")
    assert e3 <= 1.10 * e1 + 1e-9 * scale          # measured 0.85-0.9x
    assert float((c3.double() - ref).abs().max()) <= 1.5 * float((c1.double() - ref).abs().max())


def test_bitwise_independent_of_tile_and_batch(ops):
    g = torch.Generator(device="cuda").manual_seed(11)
    M, N, K = 12000, 768, 1024      # 94 x 3 tiles of 128x256: one whole round of the 256 CUs + a tail (modes 6 / 7 split at row 10880)
    a = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) * 0.05
    bias = torch.randn(N, device="cuda", generator=g)
    outs = [ops.gemm(a, w, bias, act=ops.ACT_GELU, precision="split3", tile=t) for t in ops.SPLIT3_MODES]
    torch.cuda.synchronize()
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    # a single "image" (rows 289..578) computed alone == the same rows inside the batch
    one = ops.gemm(a[289:578].contiguous(), w, bias, act=ops.ACT_GELU, precision="split3")
    tail = ops.gemm(a[11000:].contiguous(), w, bias, act=ops.ACT_GELU, precision="split3", tile=6)
    torch.cuda.synchronize()
    assert torch.equal(one, outs[0][289:578]) and torch.equal(tail, outs[0][11000:])


@pytest.mark.parametrize("act", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("res", [False, True])
def test_epilogues_vs_torch(ops, act, res):
    g = torch.Generator(device="cuda").manual_seed(act * 2 + res)
    M, N, K = 333, 200, 96          # ragged M and N (N % 32 != 0), three K slices
    a = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) * 0.2
    bias = torch.randn(N, device="cuda", generator=g)
    r = torch.randn(M, N, device="cuda", generator=g) if res else None
    for tile in (1, 2, 3, 4, 5, 6, 7):
        out = ops.gemm(a, w, bias, residual=r, act=act, precision="split3", tile=tile)
        torch.cuda.synchronize()
        z = a.double() @ w.double().T + bias.double()
        z = {0: lambda v: v, 1: lambda v: torch.nn.functional.gelu(v), 2: lambda v: torch.nn.functional.silu(v), 3: torch.tanh,
             4: lambda v: torch.nn.functional.leaky_relu(v, 0.01)}[act](z)
        if res:
            z = z + r.double()
        assert float((out.double() - z).abs().max()) < 2e-5 * max(1.0, float(z.abs().max()))


def test_inplace_residual_and_row_maps(ops):
    g = torch.Generator(device="cuda").manual_seed(5)
    n, L, Lt, D = 6, 40, 24, 128
    w = torch.randn(D, D, device="cuda", generator=g) * 0.1
    X = torch.randn(n * L, D, device="cuda", generator=g)
    X0 = X.clone()
    t = torch.randn(n * L, D, device="cuda", generator=g)
    ops.gemm(t, w, residual=X, out=X, precision="split3")                       # in place: X += t W^T
    torch.cuda.synchronize()
    assert float((X.double() - (X0.double() + t.double() @ w.double().T)).abs().max()) < 1e-4
    # A rows read from / C rows written to the [:, :Lt] token slice of an (n, L, D) buffer
    J = torch.randn(n * L, D, device="cuda", generator=g)
    y = ops.gemm(J, w, M=n * Lt, a_seg=(Lt, L), precision="split3")
    sl = J.reshape(n, L, D)[:, :Lt].reshape(n * Lt, D)
    torch.cuda.synchronize()
    assert float((y.double() - sl.double() @ w.double().T).abs().max()) < 1e-4
    out = torch.zeros(n * L, D, device="cuda")
    ops.gemm(sl.contiguous(), w, out=out, M=n * Lt, c_seg=(Lt, L), precision="split3")
    torch.cuda.synchronize()
    o3 = out.reshape(n, L, D)
    assert float((o3[:, :Lt].reshape(n * Lt, D).double() - sl.double() @ w.double().T).abs().max()) < 1e-4
    assert float(o3[:, Lt:].abs().max()) == 0.0


def test_weight_cache_follows_inplace_updates(ops):
    g = torch.Generator(device="cuda").manual_seed(9)
    a = torch.randn(64, 64, device="cuda", generator=g)
    w = torch.randn(64, 64, device="cuda", generator=g)
    y1 = ops.gemm(a, w, precision="split3")
    w.mul_(2.0)                                             # bumps the tensor version: the cached planes are not reused
    y2 = ops.gemm(a, w, precision="split3")
    torch.cuda.synchronize()
    assert torch.allclose(y2, 2.0 * y1, rtol=1e-6, atol=1e-6)


def test_planes_producers_match_the_two_step_path(ops):
    """layernorm / GEMM epilogues that write bf16x3 planes directly == fp32 output followed by the split pass, bitwise"""
    g = torch.Generator(device="cuda").manual_seed(21)
    M, C, N = 700, 1024, 768
    x = torch.randn(M, C, device="cuda", generator=g) * 3 + 0.5
    gamma = torch.randn(C, device="cuda", generator=g)
    beta = torch.randn(C, device="cuda", generator=g)
    w1 = torch.randn(N, C, device="cuda", generator=g) * 0.05
    b1 = torch.randn(N, device="cuda", generator=g)
    w2 = torch.randn(C, N, device="cuda", generator=g) * 0.05
    old = ops.PRECISION
    ops.set_precision("split3")
    try:
        h_pl = ops.layernorm(x, gamma, beta, to_gemm=True)
        assert isinstance(h_pl, ops.Planes)
        h = ops.layernorm(x, gamma, beta)
        torch.cuda.synchronize()
        assert torch.equal(h_pl.float(), h)                                 # the planes carry the fp32 value exactly
        for tile in (1, 2, 3, 4, 5, 6, 7):
            f_pl = ops.gemm(h_pl, w1, b1, act=ops.ACT_GELU, to_gemm=True, tile=tile)
            f = ops.gemm(h, w1, b1, act=ops.ACT_GELU, tile=tile)
            torch.cuda.synchronize()
            # the same planes as the split pass makes of the fp32 output (tiny GELU tails, < 1e-30, lose their subnormal low pieces
            # either way, so the comparison is between the planes, not with f itself)
            assert isinstance(f_pl, ops.Planes) and torch.equal(f_pl.rowmajor(), ops.split3(f))
            assert float((f_pl.float() - f).abs().max()) < 1e-30
            y_pl = ops.gemm(f_pl, w2, residual=x, tile=tile)
            y = ops.gemm(f, w2, residual=x, tile=tile)
            torch.cuda.synchronize()
            assert torch.equal(y_pl, y)
        # SiLU-fused LN and a row-mapped input (cross blocks)
        t_pl = ops.layernorm(x, gamma, beta, act=ops.ACT_SILU, M=2 * 300, x_seg=(300, 350), to_gemm=True)
        t = ops.layernorm(x, gamma, beta, act=ops.ACT_SILU, M=2 * 300, x_seg=(300, 350))
        torch.cuda.synchronize()
        assert torch.equal(t_pl.float(), t)
    finally:
        ops.set_precision(old)


@pytest.mark.parametrize("L,nseq,heads", [(289, 3, 4), (256, 4, 2), (50, 5, 12), (545, 2, 3), (290, 2, 2)])
def test_attention_planes_output(ops, L, nseq, heads):
    """attention writing the out-projection's operand as planes == its fp32 output put through the split pass (incl. the
    ragged VALU rows of L = 289 / 545 / 290)"""
    g = torch.Generator(device="cuda").manual_seed(L + heads)
    D = heads * 64
    qkv = torch.randn(nseq * L, 3 * D, device="cuda", generator=g)
    old = ops.PRECISION
    ops.set_precision("split3")
    try:
        ref = ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], None, L, nseq, heads)
        pl = ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], None, L, nseq, heads, to_gemm=True)
        torch.cuda.synchronize()
        assert isinstance(pl, ops.Planes) and torch.is_tensor(ref)
        assert torch.equal(pl.rowmajor(), ops.split3(ref))
    finally:
        ops.set_precision(old)


def test_split_launch_modes_with_row_map_residual_and_planes(ops):
    """modes 6 / 7 (whole rounds of big tiles + a tail launch): the tail addresses C / R / the planes through its row base"""
    g = torch.Generator(device="cuda").manual_seed(31)
    n, Lt, L, K, N = 4, 3000, 3500, 256, 768
    M = n * Lt                                   # 94 m-tiles x 3 (or 6) n-tiles: a whole round + a tail at row 10880
    a = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) * 0.1
    bias = torch.randn(N, device="cuda", generator=g)
    r = torch.randn(M, N, device="cuda", generator=g)
    ref = {}
    for tile in (1, 6, 7, 8, 9, 10, 11, 12, 13, 16, 17, 26, 27, 28, 29, 30, 31):
        out = torch.zeros(n * L, N, device="cuda")
        ops.gemm(a, w, bias, out=out, M=M, c_seg=(Lt, L), precision="split3", tile=tile)
        y = ops.gemm(a, w, bias, residual=r, act=ops.ACT_SILU, precision="split3", tile=tile)
        old = ops.PRECISION
        ops.set_precision("split3")
        try:
            pl = ops.gemm(a, w, bias, residual=r, to_gemm=True, tile=tile)
        finally:
            ops.set_precision(old)
        torch.cuda.synchronize()
        ref[tile] = (out.clone(), y.clone(), pl.t.clone())
    for tile in (6, 7, 8, 9, 10, 11, 12, 13, 16, 17, 26, 27, 28, 29, 30, 31):
        assert all(torch.equal(p, q) for p, q in zip(ref[1], ref[tile])), tile
    o3 = ref[1][0].reshape(n, L, N)
    want = (a.double() @ w.double().T + bias.double()).reshape(n, Lt, N)
    assert float((o3[:, :Lt].double() - want).abs().max()) < 1e-4 and float(o3[:, Lt:].abs().max()) == 0.0


@pytest.mark.perf
def test_single_tile_latency(ops):
    """the B = 1 building block under the split arithmetic: one 64x64 tile, K = 512 (the 32x32 latency mode with 64-k stages), timed
    by the dispatch's own timestamps.  A timing check, not parity: marked `perf`, so conftest collects it after every parity test.
    What the timestamps of a lone ~5 us dispatch contain is measured by tools/micro/launch_latency.hip: the
    in-kernel span (first wave's start -> last wave's end) and the back-to-back rate are ~5 us, the timestamp pair of an isolated
    dispatch adds the packet processor's per-dispatch work in front of the first wave and the end-of-kernel cache write-back and
    completion signal behind the last (8.6 us at best inside a compress step, 10.9-11.8 us after an idle gap on the driver's box
    in round 2).  The bound is that in-situ figure with 2x headroom for box-to-box differences, and the operand is pre-split so
    that no producer kernel sits in front of the timed dispatch."""
    a = torch.randn(64, 512, device="cuda")
    w = torch.randn(64, 512, device="cuda")
    ap = ops.split3_planes(a)                     # pre-split: the timed dispatch has no split3_rows_kernel in front of it
    for _ in range(10):
        ops.gemm(ap, w, precision="split3", tile=4)
    torch.cuda.synchronize()
    ops.profile_begin(16)
    for _ in range(8):
        ops.gemm(ap, w, precision="split3", tile=4)
    recs = ops.profile_end()
    best = min(ms for _, ms, _ in recs)
    print(f"[latency] 64x64x512 split GEMM, dispatch timestamps: best {best * 1e3:.1f} us, all "
          f"{[round(ms * 1e3, 1) for _, ms, _ in recs]}")
    assert len(recs) == 8 and best < 0.024, f"64x64x512 split GEMM took {best * 1e3:.1f} us"


@pytest.mark.parametrize("M,N,K", [(70, 201, 32), (33, 7, 64), (257, 130, 128), (1, 4, 96), (300, 64, 192)])
def test_edge_shapes_every_mode(ops, M, N, K):
    """one or two K stages (incl. a single 64-k stage of the latency tile), N not a multiple of 4 (narrow epilogue), M = 1"""
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g)
    bias = torch.randn(N, device="cuda", generator=g)
    r = torch.randn(M, N, device="cuda", generator=g)
    ref = a.double() @ w.double().T + bias.double() + r.double()
    first = None
    for tile in ops.SPLIT3_MODES:
        out = ops.gemm(a, w, bias, residual=r, precision="split3", tile=tile)
        torch.cuda.synchronize()
        assert float((out.double() - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max())), tile
        first = out if first is None else first
        assert torch.equal(out, first), tile


def test_row_major_and_slice_major_planes_agree(ops):
    """A and W planes in the plain row-major layouts [3][M][K] / [3][N][K] (sgic_split3_f32, opts->a_packed = w_packed = 0: what a C
    caller without the packing pass hands over) and in the slice-major layout ops.gemm uses (sgic_split3_pack_f32 and the producers'
    outputs, a_packed = w_packed = 1) give bitwise the same product in every combination, for a register-staged tile (5, 4), an
    LDS-DMA tile (1), the ring kernel (19, 23) and a two-launch mode (8: the second launch starts inside the A planes)"""
    from sgic_amd._lib import call, launch_opts
    from sgic_amd.ops import _p
    g = torch.Generator(device="cuda").manual_seed(5)
    M, N, K = 11200, 520, 448     # 88 x 3 tiles of 128x256: mode 8 = 85 whole m-tiles (one round of 256 CUs) + a second launch from row 10880
    a = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) * 0.1
    bias = torch.randn(N, device="cuda", generator=g)
    ref = ops.gemm(a, w, bias, precision="split3", tile=2)
    planes = {(0, "a"): ops.split3(a), (0, "w"): ops.split3(w), (1, "a"): ops.split3_planes(a).t, (1, "w"): ops.split3_planes(w).t}
    for tile in (1, 5, 19, 23, 4, 8, 27, 30, 31):
        for apk in (0, 1):
            for wpk in (0, 1):
                out = torch.empty(M, N, device="cuda")
                call("sgic_gemm_split3_f32", None, 0, 0, 0, _p(planes[apk, "a"]), _p(planes[wpk, "w"]), _p(bias), None, 0, _p(out), N, None,
                     M, N, K, 0, 0, 0, launch_opts(tile, 0, None, wpk, apk))
                torch.cuda.synchronize()
                assert torch.equal(out, ref), (tile, apk, wpk)
