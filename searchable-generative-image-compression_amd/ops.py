"""Thin Python wrappers over the transform kernels of libsgic (C ABI, include/sgic.h).  Tensors are torch
CUDA tensors used purely as device-memory handles; all arithmetic happens in the HIP kernels."""
import ctypes
import json
import os

import torch

from ._lib import call, check, launch_opts, lib, require_gpu

ACT_NONE, ACT_GELU, ACT_SILU, ACT_TANH, ACT_LRELU = 0, 1, 2, 3, 4

# Optional live profiling of the dominant kernel (bench.py): while a window is open every GEMM / conv launch carries
# the profiler handle in its sgic_launch_opts, so its dispatch is timed by its own event pair, and (flops, shape key)
# is appended to PROFILE.  Host-side state of the launching thread (one process per GPU, one launching thread).
PROFILE = None
_PROFILER = None


def _profiler():
    global _PROFILER
    if _PROFILER is None:
        h = ctypes.c_void_p(0)
        check(lib.sgic_profiler_create(ctypes.byref(h)), "sgic_profiler_create")
        _PROFILER = h
    return _PROFILER


def profile_begin(max_launches=8192):
    """open a profile window: every GEMM / conv launch from now on is timed by its own dispatch (sgic_profiler_begin)"""
    global PROFILE
    finalize_autotune()
    check(lib.sgic_profiler_begin(_profiler(), int(max_launches)), "sgic_profiler_begin")
    PROFILE = []
    del PROFILE_TILES[:]


def profile_end():
    """close the window -> list of (flops, milliseconds, shape key) per launch, in launch order"""
    global PROFILE
    if PROFILE is None:
        raise RuntimeError("profile_end() without an open profile window")
    recs, PROFILE = PROFILE, None
    buf = (ctypes.c_float * max(1, len(recs)))()
    n = ctypes.c_int(0)
    check(lib.sgic_profiler_end(_profiler(), buf, len(recs), ctypes.byref(n)), "sgic_profiler_end")
    if n.value != len(recs):
        raise RuntimeError(f"profile window: {len(recs)} launches recorded on the host, {n.value} timed on the device")
    return [(fl, float(buf[i]), key) for i, (fl, key) in enumerate(recs)]


def _opts(tile=0, attn=0, prof=None, w_packed=0, a_packed=0):
    """prof: an explicit profiler handle (the tuner's probes); default: the bench's window when one is open"""
    return launch_opts(tile, attn, prof if prof is not None else (_PROFILER if PROFILE is not None else None), w_packed, a_packed)


def _rows(t):
    """a 2-D fp32 CUDA view whose last dim is contiguous -> (tensor, leading dimension)"""
    assert t.dim() == 2 and t.stride(1) == 1 and t.dtype == torch.float32 and t.is_cuda, (t.shape, t.stride(), t.dtype)
    return t, (t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1]))   # a 1-row view may carry any stride


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _cl(v):
    return ctypes.c_long(int(v))


def empty(*shape, like=None, dtype=torch.float32, device=None):
    return torch.empty(*shape, dtype=dtype, device=like.device if like is not None else device)


# Per-shape launch-mode autotuning ("measure, don't guess"): the first time a GEMM / conv / attention shape is seen and
# neither the tile cache nor a neighbouring M of the same (N, K, epilogue) family knows it, every launch mode of the
# kernel is timed with HIP events on the launch stream and the fastest is remembered.  All modes give bitwise identical
# results (the k order is fixed), so tuning never changes an output; the mode travels PER CALL in sgic_launch_opts (the
# library has no global launch state).  Picks persist in a JSON tile cache, so a second process -- or a new image
# geometry whose GEMMs only differ in M -- does not race again:
#   1. the in-tree cache  sgic_amd/tile_cache_gfx950.json  (committed; measured on an MI355X),
#   2. the user cache     $SGIC_TILE_CACHE or ~/.cache/sgic_amd/tile_cache_gfx950.json  (read after 1, written by
#      save_tile_cache(); bench.py and the CLI drivers call it at exit).
# AUTOTUNE = False uses the cache / built-in heuristic only.  Single-threaded by design (one launching thread).
AUTOTUNE = True
TUNE_MODES = (1, 2, 3, 4, 5, 7, 9, 10, 11, 12, 13, 14, 15)   # {128x128, 128x64} x {double, single LDS buffer}, staggered wide, mixed 128x128 + 64x64 tail, persistent (plain, mixed), 64x64, 16x16x4 latency kernel
ATTN_MODES = (1, 2, 3, 4, 5, 6, 7)                         # {single, double}-buffered K/V ring x start-up stagger {0, 4096, 8192} cycles; 7 = three row blocks per workgroup
_TILE = {}        # key -> mode (exact shapes)
_FAMILY = {}      # key without M -> {M: mode}
_DIRTY = False
_INTREE_CACHE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tile_cache_gfx950.json")


def _user_cache_path():
    return os.environ.get("SGIC_TILE_CACHE") or os.path.join(os.path.expanduser("~"), ".cache", "sgic_amd", "tile_cache_gfx950.json")


def _remember(key, mode, dirty=True):
    global _DIRTY
    _TILE[key] = mode
    _FAMILY.setdefault((key[0],) + tuple(key[2:]), {})[key[1]] = mode
    _DIRTY = _DIRTY or dirty


_MAX_MODE = {"gemm": 15, "conv3x3": 15, "gemm3": 31, "conv3": 31, "attn": 7, "attn3": 7}   # per kind of key


def _load_tile_cache():
    for path in (_INTREE_CACHE, _user_cache_path()):
        try:
            with open(path) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        for k, mode in d.get("picks", {}).items():
            parts = k.split("|")
            try:
                key = (parts[0],) + tuple(int(x) for x in parts[1:])
                mode = int(mode)
            except ValueError:
                continue
            # a cache written by another build may hold modes this library does not have: such a pick is dropped (re-raced)
            if not 0 <= mode <= _MAX_MODE.get(parts[0], -1):
                continue
            _remember(key, mode, dirty=False)


def save_tile_cache(path=None):
    """persist the picks of this process (merged over what the file already holds); atomic replace"""
    global _DIRTY
    if not _DIRTY and path is None:
        return None
    finalize_autotune()
    path = path or _user_cache_path()
    picks = {}
    try:
        with open(path) as f:
            picks = json.load(f).get("picks", {})
    except (OSError, ValueError):
        pass
    picks.update({"|".join(str(int(x)) if not isinstance(x, str) else x for x in k): int(v) for k, v in _TILE.items()})
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "w") as f:
            json.dump({"device": "gfx950", "format": "kind|M|...: launch mode (sgic_launch_opts)", "picks": dict(sorted(picks.items()))}, f, indent=0)
        os.replace(tmp, path)
    except OSError:
        return None
    _DIRTY = False
    return path


def _lookup(key):
    """exact pick, else the pick of the nearest M (within 2x) of the same (N, K, epilogue) family, else None"""
    m = _TILE.get(key)
    if m is not None:
        return m
    fam = _FAMILY.get((key[0],) + tuple(key[2:]))
    if fam:
        M = key[1]
        near = min(fam, key=lambda x: abs(x - M))
        if M <= 2 * near and near <= 2 * M:
            _TILE[key] = fam[near]          # not persisted: a borrowed pick, not a measurement
            return fam[near]
    return None


def tile_of(key, dev=None):
    """launch mode in use for a profile-record shape key (bench.py's by_shape table)"""
    if key and key[0] == "batched":
        return 0
    if key and key[0] in ("conv3x3", "conv3"):
        return _TILE.get(key)
    M, N, K, res, act = key[:5]
    kind = "gemm3" if len(key) > 5 and key[5] == "s3" else "gemm"
    return _TILE.get((kind, M, N, K, int(bool(res)), act))


_load_tile_cache()


def _save_at_exit():
    """every entry point (CLIs, service, tools, bench) leaves its new measurements behind for the next process"""
    try:
        if _DIRTY:
            save_tile_cache()
    except Exception:   # noqa: BLE001 -- never turn a clean exit into a failing one
        pass


import atexit  # noqa: E402

atexit.register(_save_at_exit)


_TUNE_PROF = None


def _tune_profiler():
    """a profiler handle of the tuner's own (the bench's window must not see the probes)"""
    global _TUNE_PROF
    if _TUNE_PROF is None:
        h = ctypes.c_void_p(0)
        check(lib.sgic_profiler_create(ctypes.byref(h)), "sgic_profiler_create")
        _TUNE_PROF = h
    return _TUNE_PROF


def _tune(key, launch, modes=None):
    """two interleaved passes over the modes, best-of per mode: a single short sample mis-ranks modes that are within
    a few percent of each other (clock ramp, cold L2 after the previous mode's different tile walk).
    Short launches are timed by the DISPATCH's own timestamps (sgic_profiler: hipExtLaunchKernel start / stop events), i.e.
    the kernel's duration alone: host-side event pairs around Python-issued launches cannot rank kernels shorter than the
    ~13 us a launch costs on the host, which is every GEMM of a single-image request."""
    prof = _tune_profiler()
    modes = TUNE_MODES if modes is None else modes
    reps = 3
    times = {}
    for _ in range(2):
        if M_big(key):
            # launches of >= ~50 us: host-side event pairs around four back-to-back launches, which also charge a mode for
            # what it costs BETWEEN kernels (the ramp of 512 persistent workgroups, the drain); measured better in the model
            for mode in modes:
                launch(mode)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    launch(mode)
                e1.record()
                e1.synchronize()
                t = e0.elapsed_time(e1)
                times[mode] = min(t, times.get(mode, t))
            continue
        check(lib.sgic_profiler_begin(prof, len(modes) * reps), "sgic_profiler_begin")
        for mode in modes:
            launch(mode)                               # warm, untimed
            for _ in range(reps):
                launch(mode, prof)
        buf = (ctypes.c_float * (len(modes) * reps))()
        n = ctypes.c_int(0)
        check(lib.sgic_profiler_end(prof, buf, len(modes) * reps, ctypes.byref(n)), "sgic_profiler_end")
        if n.value != len(modes) * reps:
            raise RuntimeError(f"tile tuner: {n.value} timed launches, expected {len(modes) * reps}")
        for i, mode in enumerate(modes):
            t = min(buf[i * reps + r] for r in range(reps))
            times[mode] = min(t, times.get(mode, t))
    order = sorted(times, key=times.get)
    best = order[0]
    _remember(key, best)
    # second stage, in context: the leaders of the isolated race (within 6 % of the best) are re-timed on the next real
    # occurrences of this shape, i.e. with the caches in the state the surrounding kernels leave them in
    cands = [m for m in order[:CTX_CANDS] if times[m] <= 1.06 * times[best]]
    if CTX_REPS > 0 and len(cands) > 1 and M_big(key):
        _CTX[key] = {"cands": cands, "t": {m: [] for m in cands}, "i": 0}
    return best


CTX_CANDS, CTX_REPS = 3, 2
_CTX = {}


def M_big(key):
    """in-context re-timing synchronises the stream once per sample: only worth it for launches of >= ~50 us"""
    if key[0] in ("conv3x3", "conv3"):
        return True
    return 2.0 * key[1] * key[2] * key[3] >= 5e9


def _ctx_launch(key, launch):
    """one in-context sample of the next candidate mode; picks the winner when every candidate has CTX_REPS samples"""
    c = _CTX[key]
    mode = c["cands"][c["i"] % len(c["cands"])]
    c["i"] += 1
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    launch(mode)
    e1.record()
    e1.synchronize()
    c["t"][mode].append(e0.elapsed_time(e1))
    if c["i"] >= len(c["cands"]) * CTX_REPS:
        _remember(key, min(c["cands"], key=lambda m: min(c["t"][m])))
        del _CTX[key]


def finalize_autotune():
    """close every pending in-context race with the samples it has (call before a timed region)"""
    for key in list(_CTX):
        c = _CTX.pop(key)
        done = [m for m in c["cands"] if c["t"][m]]
        if done:
            _remember(key, min(done, key=lambda m: min(c["t"][m])))


def _pick_and_launch(key, launch, big_enough, restore=None, modes=None):
    """shared tile-mode policy of gemm() / conv3x3(): cache -> family neighbour -> race (outside profile windows) -> heuristic.
    Returns True when the launch was already done by an in-context sample."""
    tile = 0
    if big_enough:
        tile = _lookup(key)
        if tile is None:
            if not AUTOTUNE or PROFILE is not None:
                tile = 0      # never tune inside a profile window (its probes would take event slots): built-in heuristic
            elif restore is not None:
                # in-place residual GEMMs (out is residual) are not idempotent: save / restore the buffer around tuning
                saved = restore.clone()
                tile = _tune(key, launch, modes)
                restore.copy_(saved)
            else:
                tile = _tune(key, launch, modes)
        elif key in _CTX and PROFILE is None:
            _ctx_launch(key, launch)
            return True
    global LAST_TILE
    LAST_TILE = tile
    launch(tile)
    return False


LAST_TILE = 0          # launch mode of the most recent _pick_and_launch (0 = the library's heuristic)
PROFILE_TILES = []     # launch mode actually used by each record of the current / last profile window (tools/pmc_by_shape.py)


# ---- GEMM arithmetic: "f32" = the exact-fp32 MFMA (csrc/gemm.hip), "split3" = fp32-accurate bf16x3 split on the bf16 matrix
# pipe (csrc/gemm_split.hip).  A process-level choice (SGIC_GEMM or set_precision): which kernel family a GEMM takes may
# depend on (N, K) and on the operands' kinds, never on M, so single-image and batched requests stay bitwise identical.
PRECISION = os.environ.get("SGIC_GEMM", "split3")
assert PRECISION in ("f32", "split3"), f"SGIC_GEMM={PRECISION!r}: expected f32 or split3"
SPLIT3_MODES = (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31)
SPLIT3_RING_MODES = (18, 19, 20, 21, 22, 23, 24, 25)   # small tiles, deep LDS-DMA ring: raced only for launches that cannot fill the chip (_split3_modes)
SPLIT3_SMALL_M = 2048


def _split3_modes(M):
    """candidate launch modes of a split GEMM with M rows: the ring kernel's 32..80-row tiles are for single-image sized launches"""
    return SPLIT3_MODES if M <= SPLIT3_SMALL_M else tuple(m for m in SPLIT3_MODES if m not in SPLIT3_RING_MODES)


_W3 = {}          # (data_ptr, N, K, ldw, version) -> (planes, w): the weight is pinned so its address cannot be reused
_W3_BYTES = 0
_W3_LIMIT = 24 << 30
_A3_WS = {}       # (device, stream) -> workspace for the activation planes


def set_precision(p):
    global PRECISION
    assert p in ("f32", "split3"), p
    PRECISION = p


class Planes:
    """an activation held as bf16x3 planes (int16 bit patterns) in the slice-major layout [3, cols / 32, rows, 32] (csrc/split3.h):
    the A operand of a split GEMM written directly by its producer (layernorm / attention / gemm epilogue / split3_planes), so no
    fp32 copy and no separate split pass exists"""
    __slots__ = ("t", "rows", "cols")

    def __init__(self, rows, cols, device, buf=None):
        n = 3 * rows * cols
        if buf is not None and buf.numel() >= n:
            self.t = buf
        else:
            self.t = torch.empty(n, device=device, dtype=torch.int16)
        self.rows, self.cols = rows, cols

    @property
    def shape(self):
        return (self.rows, self.cols)

    @property
    def device(self):
        return self.t.device

    def rowmajor(self):
        """the planes as a [3, rows, cols] tensor (tests / debugging): what ops.split3 returns for the same values"""
        p = self.t[:3 * self.rows * self.cols].view(3, self.cols // 32, self.rows, 32)
        return p.permute(0, 2, 1, 3).reshape(3, self.rows, self.cols)

    def float(self):
        """the fp32 values (tests / debugging): the sum of the three planes"""
        p = self.rowmajor()
        f = lambda q: (q.to(torch.int32) << 16).view(torch.float32)
        return (f(p[0]) + f(p[1])) + f(p[2])


def planes_ok(cols, precision=None):
    """may a producer hand its [rows, cols] output to the next GEMM as planes?"""
    return (precision or PRECISION) == "split3" and cols % 32 == 0


def split3(x, ld=None, rows=None, seg=(0, 0), out=None):
    """x[rows, cols] fp32 -> planes [3, rows, cols] (uint16 = bf16 bit patterns), x = p0 + p1 + p2 exactly"""
    x, ldx = _rows(x)
    rows = x.shape[0] if rows is None else rows
    cols = x.shape[1]
    if out is None:
        out = torch.empty(3, rows, cols, device=x.device, dtype=torch.int16)
    call("sgic_split3_f32", _p(x), ldx if ld is None else ld, rows, cols, seg[0], seg[1], _p(out))
    return out


def split3_planes(x, out=None):
    """x[rows, cols] fp32 -> a Planes object (slice-major layout): a pre-split A operand for gemm()"""
    x, ldx = _rows(x)
    rows, cols = x.shape
    assert cols % 32 == 0
    pl = out if isinstance(out, Planes) and out.shape == (rows, cols) else Planes(rows, cols, x.device)
    call("sgic_split3_pack_f32", _p(x), ldx, rows, cols, _p(pl.t))
    return pl


def weight_planes(w, ldw):
    """bf16x3 planes of a constant operand, split once (keyed by address + in-place version), in the slice-major layout
    [3][K / 32][N][32] of sgic_split3_pack_f32: consumers pass w_packed=1 (sgic_launch_opts)"""
    global _W3_BYTES
    N, K = w.shape
    key = (w.data_ptr(), N, K, ldw, w._version)
    hit = _W3.get(key)
    if hit is None:
        planes = torch.empty(3, N, K, device=w.device, dtype=torch.int16)
        call("sgic_split3_pack_f32", _p(w), ldw, N, K, _p(planes))
        nbytes = planes.numel() * 2
        while _W3 and _W3_BYTES + nbytes > _W3_LIMIT:      # oldest first
            old = next(iter(_W3))
            _W3_BYTES -= _W3.pop(old)[0].numel() * 2
        _W3[key] = hit = (planes, w)
        _W3_BYTES += nbytes
    return hit[0]


def _a3_workspace(dev, n_u16):
    key = (dev, _stream_key())
    ws = _A3_WS.get(key)
    if ws is None or ws.numel() < n_u16:
        ws = torch.empty(max(n_u16, 1 << 24), device=dev, dtype=torch.int16)
        _A3_WS[key] = ws
    return ws


def _stream_key():
    from ._lib import stream
    return stream().value


def gemm(a, w, bias=None, residual=None, act=ACT_NONE, out=None, M=None, a_seg=(0, 0), c_seg=(0, 0), tile=None, w_const=True,
         precision=None, to_gemm=False):
    """out[M,N] = act(a[M,K] @ w[N,K]^T + bias) + residual.  a_seg/c_seg = (seg, seg_stride) row maps:
    logical row m of A (resp. C) lives at physical row (m // seg) * seg_stride + m % seg.
    tile: force a launch mode (tests / tools); default = cache / autotuner.
    w_const: w is a constant (a model weight): under PRECISION == "split3" its bf16x3 planes are cached.  A `w` whose contents
    change between calls (the search index) must pass False and takes the fp32 MFMA kernel.
    a may be a Planes object (written by a producer called with to_gemm=True).  to_gemm=True: the result feeds only another
    GEMM as its A operand -> returned as Planes when the split path is active (else the fp32 tensor as always); `out` may
    then be a Planes object to reuse."""
    require_gpu()
    a_planes = a if isinstance(a, Planes) else None
    if a_planes is not None:
        K, lda = a_planes.cols, a_planes.cols
        assert a_seg[0] == 0 and (M is None or M == a_planes.rows)
        M = a_planes.rows
        dev = a_planes.device
    else:
        a, lda = _rows(a)
        K = a.shape[1]
        if M is None:
            M = a.shape[0]
        dev = a.device
    w, ldw = _rows(w)
    N = w.shape[0]
    assert w.shape[1] == K, ((M, K), w.shape)
    use3 = (precision or PRECISION) == "split3" and w_const and K % 32 == 0
    assert a_planes is None or use3, "a Planes operand needs the split path"
    out_planes = None
    if to_gemm and use3 and N % 32 == 0 and c_seg[0] == 0:
        out_planes = out if isinstance(out, Planes) and out.shape == (M, N) else Planes(M, N, dev, buf=out.t if isinstance(out, Planes) else None)
        out, ldc = None, N
    else:
        if isinstance(out, Planes):
            out = None
        if out is None:
            assert c_seg[0] == 0
            out = torch.empty(M, N, device=dev, dtype=torch.float32)
        out, ldc = _rows(out)
        assert out.shape[1] == N
    ldr = 0
    if residual is not None:
        residual, ldr = _rows(residual)
        assert residual.shape[1] == N and residual.shape[0] >= M
    if bias is not None:
        assert bias.shape == (N,) and bias.is_contiguous()
    inplace = residual is not None and out is not None and out.data_ptr() == residual.data_ptr()

    if use3:
        wp = weight_planes(w, ldw)
        ap = a_planes.t if a_planes is not None else _a3_workspace(dev, 3 * M * K)
        a_f32 = None if a_planes is not None else a
        cp = out_planes.t if out_planes is not None else None

        def launch3(mode, prof=None):
            call("sgic_gemm_split3_f32", _p(a_f32), lda, a_seg[0], a_seg[1], _p(ap), _p(wp), _p(bias), _p(residual), ldr, _p(out), ldc,
                 _p(cp), M, N, K, act, c_seg[0], c_seg[1], _opts(tile=mode, prof=prof, w_packed=1, a_packed=1))

        if tile is not None:
            launch3(tile)
        else:
            key = ("gemm3", M, N, K, int(residual is not None), act)
            if _pick_and_launch(key, launch3, M * N >= (1 << 16), restore=out if inplace else None, modes=_split3_modes(M)):
                return out_planes if out_planes is not None else out
        if PROFILE is not None:
            PROFILE.append((2.0 * M * N * K, (M, N, K, residual is not None, act, "s3")))
            PROFILE_TILES.append(tile if tile is not None else LAST_TILE)
        return out_planes if out_planes is not None else out

    def launch(mode, prof=None):
        call("sgic_gemm_f32", _p(a), lda, _p(w), ldw, _p(bias), _p(residual), ldr, _p(out), ldc, M, N, K, act,
             a_seg[0], a_seg[1], c_seg[0], c_seg[1], _opts(tile=mode, prof=prof))

    if tile is not None:
        launch(tile)
    else:
        key = ("gemm", M, N, K, int(residual is not None), act)
        if _pick_and_launch(key, launch, M * N >= (1 << 16), restore=out if inplace else None):
            return out
    if PROFILE is not None:   # the launch took the next event pair of the open profile window (profile_begin)
        PROFILE.append((2.0 * M * N * K, (M, N, K, residual is not None, act)))
        PROFILE_TILES.append(tile if tile is not None else LAST_TILE)
    return out


def layernorm(x, gamma, beta, out=None, eps=1e-5, act=ACT_NONE, M=None, x_seg=(0, 0), y_seg=(0, 0), to_gemm=False):
    """to_gemm=True: the output feeds only a GEMM as its A operand -> written directly as bf16x3 Planes when the split path is
    active and the width allows it (`out` may be a Planes object to reuse); otherwise the fp32 tensor as always."""
    x, ldx = _rows(x)
    C = x.shape[1]
    if M is None:
        M = x.shape[0]
    assert gamma.shape == (C,) and beta.shape == (C,)
    if to_gemm and planes_ok(C) and C % 256 == 0 and C <= 2048 and y_seg[0] == 0:
        pl = out if isinstance(out, Planes) and out.shape == (M, C) else Planes(M, C, x.device, buf=out.t if isinstance(out, Planes) else None)
        call("sgic_layernorm_split3_f32", _p(x), ldx, x_seg[0], x_seg[1], _p(gamma), _p(beta), _p(pl.t), M, C, float(eps), act)
        return pl
    if isinstance(out, Planes):
        out = None
    if out is None:
        assert y_seg[0] == 0
        out = torch.empty(M, C, device=x.device, dtype=torch.float32)
    out, ldy = _rows(out)
    call("sgic_layernorm_f32", _p(x), ldx, x_seg[0], x_seg[1], _p(gamma), _p(beta), _p(out), ldy, y_seg[0], y_seg[1],
         M, C, float(eps), act)
    return out


def attention(q, k, v, out, L, nseq, nheads, rowmap=None, bias=None, biasvar=None, scale=0.125, mode=None, to_gemm=False, precision=None):
    """q,k,v,out: 2-D row-strided views (rows x nheads*64).  mode: force attn_mode (tests / tools).
    to_gemm=True: the output feeds only the out-projection GEMM -> written as bf16x3 Planes over q's row space when the split
    path is active (`out` may be None or a Planes object to reuse); otherwise into the fp32 `out` as always."""
    q, ldq = _rows(q)
    k, ldk = _rows(k)
    v, ldv = _rows(v)
    D = nheads * 64
    pl = None
    if to_gemm and planes_ok(D):
        rows = q.shape[0]
        pl = out if isinstance(out, Planes) and out.shape == (rows, D) else Planes(rows, D, q.device, buf=out.t if isinstance(out, Planes) else None)
    else:
        if out is None or isinstance(out, Planes):
            out = torch.empty(q.shape[0], D, device=q.device, dtype=torch.float32)
        out, ldo = _rows(out)
    if rowmap is not None:
        assert rowmap.dtype == torch.int32 and rowmap.numel() == nseq * L and rowmap.is_contiguous()
    if bias is not None:
        assert bias.dim() == 3 and bias.shape[1] == L and bias.shape[2] == L and bias.is_contiguous()
    if biasvar is not None:
        assert biasvar.dtype == torch.int32 and biasvar.numel() == nseq

    s3 = 8 if (precision or PRECISION) == "split3" else 0     # attn_mode bit 3: S^T = K Q^T as a bf16x3 split product

    def launch(m):
        if pl is not None:
            call("sgic_attention_split3_f32", _p(q), ldq, _p(k), ldk, _p(v), ldv, _p(pl.t), ctypes.c_long(pl.rows), L, nseq, nheads,
                 _p(rowmap), _p(bias), _p(biasvar), float(scale), launch_opts(0, m | s3, None))
        else:
            call("sgic_attention_f32", _p(q), ldq, _p(k), ldk, _p(v), ldv, _p(out), ldo, L, nseq, nheads, _p(rowmap), _p(bias),
                 _p(biasvar), float(scale), launch_opts(0, m | s3, None))

    # K/V ring depth (single buffer + more workgroups per CU vs double buffer + one barrier per tile) depends on L and
    # on how many workgroups the launch has -> tuned per shape like the GEMM tiles; results are identical.
    if mode is None:
        mode = 0
        if nseq * nheads * L >= 4096:
            key = ("attn3" if s3 else "attn", nseq, L, nheads, int(bias is not None))
            mode = _lookup(key)
            if mode is None:
                mode = 0
                if AUTOTUNE and PROFILE is None:
                    best_t = None
                    for cand in ATTN_MODES:
                        launch(cand)
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(3):
                            launch(cand)
                        e1.record()
                        e1.synchronize()
                        t = e0.elapsed_time(e1)
                        if best_t is None or t < best_t:
                            mode, best_t = cand, t
                    _remember(key, mode)
    launch(mode)
    return pl if pl is not None else out


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


def scale_indexes(scales_flat, idx_out, thr):
    call("sgic_scale_to_index", _p(scales_flat), _cl(scales_flat.numel()), float(-1.0 if thr is None else thr), _p(idx_out))


def index_step(scales, ld_sm, B, H, W, C, k, thr, idx):
    call("sgic_index_step", _p(scales), ld_sm, B, H, W, C, k, float(-1.0 if thr is None else thr), _p(idx))


def index_margins(scales, ld_sm, B, H, W, C, k, thr, margin, alt):
    call("sgic_index_margins", _p(scales), ld_sm, B, H, W, C, k, float(-1.0 if thr is None else thr), _p(margin), _p(alt))


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
    nb = lib.sgic_pack12_size(n)
    out = torch.empty(B, nb, dtype=torch.uint8, device=idx_i32.device)
    call("sgic_pack12_batch", _p(idx_i32), B, n, _p(out))
    return out


# ---- decode-side kernels ----
def gemm_batched(a, lda, sa, w, ldw, sw, out, ldc, sc, M, N, K, batch, bias=None, residual=None, ldr=0, sr=0, act=ACT_NONE):
    """batch of GEMMs on raw pointers/strides (elements); a, w, out, residual are tensors used as base pointers"""
    call("sgic_gemm_batched_f32", _p(a), lda, _cl(sa), _p(w), ldw, _cl(sw), _p(bias), _p(residual), ldr, _cl(sr), _p(out), ldc,
         _cl(sc), M, N, K, act, batch, _opts())
    if PROFILE is not None:
        PROFILE.append((2.0 * M * N * K * batch, ("batched", batch, M, N, K, residual is not None, act)))
        PROFILE_TILES.append(0)
    return out


class HaloPlanes:
    """the zero-halo input of a 3x3 convolution held as slice-major bf16x3 planes [3, C / 32, B (H+2) (W+2), 32] (int16 bit patterns),
    written directly by groupnorm / halo_copy when the convolution runs as an implicit split GEMM"""
    __slots__ = ("t", "B", "H", "W", "C")

    def __init__(self, t, B, H, W, C):
        self.t, self.B, self.H, self.W, self.C = t, B, H, W, C


_HALO3 = {}


def halo_planes_buffer(dev, B, H, W, C):
    """cached like halo_buffer: producers only write the interior, the border planes stay zero"""
    key = (str(dev), B, H, W, C)
    if key not in _HALO3:
        _HALO3[key] = HaloPlanes(torch.zeros(3 * B * (H + 2) * (W + 2) * C, device=dev, dtype=torch.int16), B, H, W, C)
    return _HALO3[key]


def conv_planes_ok(Cin, Cout, residual=False, precision=None):
    """does conv3x3 with these channels run as an implicit split GEMM (so its producer may write halo planes directly)?"""
    return (precision or PRECISION) == "split3" and Cin % 32 == 0 and not (Cout == 3 and Cin == 128 and not residual)


def conv3x3(x_halo, w, bias, B, H, W, Cin, Cout, residual=None, act=ACT_NONE, out=None, tile=None, precision=None):
    """x_halo: zero-halo NHWC buffer (B, H+2, W+2, Cin); w: (Cout, 9*Cin) in (ky,kx,cin) order -> (B*H*W, Cout)"""
    hp = x_halo if isinstance(x_halo, HaloPlanes) else None
    if hp is not None:
        assert (hp.B, hp.H, hp.W, hp.C) == (B, H, W, Cin) and conv_planes_ok(Cin, Cout, residual is not None, precision)
        dev = hp.t.device
    else:
        assert x_halo.is_contiguous() and x_halo.numel() == B * (H + 2) * (W + 2) * Cin
        dev = x_halo.device
    assert w.shape == (Cout, 9 * Cin) and w.is_contiguous()
    if out is None:
        out = torch.empty(B * H * W, Cout, device=dev, dtype=torch.float32)
    out, ldc = _rows(out)
    ldr = 0
    if residual is not None:
        residual, ldr = _rows(residual)

    if conv_planes_ok(Cin, Cout, residual is not None, precision):
        # implicit split GEMM: the weight's planes are cached; the halo planes come from the producer, or from one split pass
        # over the fp32 halo buffer's rows
        rows = B * (H + 2) * (W + 2)
        wp = weight_planes(w, 9 * Cin)
        if hp is not None:
            ws = hp.t
        else:
            ws = _a3_workspace(dev, 3 * rows * Cin)
            call("sgic_split3_pack_f32", _p(x_halo), Cin, rows, Cin, _p(ws))

        def launch3(mode, prof=None):
            call("sgic_conv3x3_split3_f32", _p(ws), _p(wp), _p(bias), _p(residual), ldr, _p(out), ldc, B, H, W, Cin, Cout, act,
                 _opts(tile=mode, prof=prof, w_packed=1, a_packed=1))

        if tile is not None:
            launch3(tile)
        else:
            key = ("conv3", B * H * W, H, W, Cin, Cout, int(residual is not None), act)
            if _pick_and_launch(key, launch3, B * H * W * Cout >= (1 << 20), modes=(1, 2, 3, 5, 10, 11, 14, 15, 16, 17, 30)):
                return out
        if PROFILE is not None:
            PROFILE.append((2.0 * B * H * W * Cout * 9 * Cin, (B * H * W, Cout, 9 * Cin, residual is not None, act, "s3", "conv")))
            PROFILE_TILES.append(tile if tile is not None else LAST_TILE)
        return out

    def launch(mode, prof=None):
        call("sgic_conv3x3_f32", _p(x_halo), _p(w), _p(bias), _p(residual), ldr, _p(out), ldc, B, H, W, Cin, Cout, act,
             _opts(tile=mode, prof=prof))

    if tile is not None:
        launch(tile)
    else:
        # key[1] = M = B*H*W so that a different batch of the same geometry borrows the pick (family lookup)
        key = ("conv3x3", B * H * W, H, W, Cin, Cout, int(residual is not None), act)
        if _pick_and_launch(key, launch, B * H * W * Cout >= (1 << 20)):   # conv outputs never alias their residual
            return out
    if PROFILE is not None:   # the implicit-GEMM convolution is the same kernel: M = B*H*W, N = Cout, K = 9*Cin
        PROFILE.append((2.0 * B * H * W * Cout * 9 * Cin, (B * H * W, Cout, 9 * Cin, residual is not None, act, "f32", "conv")))
        PROFILE_TILES.append(tile if tile is not None else LAST_TILE)
    return out


_GN_WS = {}


def groupnorm(x, gamma, beta, B, H, W, swish=True, halo=False, groups=32, eps=1e-6, out=None, to_conv=None):
    """x (B*H*W, C) plain NHWC -> normalised (+swish); halo=True writes the interior of a (B,H+2,W+2,C) buffer
    whose border must already be zero (pass `out`, or a fresh zero buffer is allocated)."""
    x, ldx = _rows(x)
    C = x.shape[1]
    assert ldx == C
    dev = x.device
    key = (str(dev), B, C)
    if key not in _GN_WS:
        _GN_WS[key] = (torch.empty(B * 64 * C * 2, dtype=torch.float64, device=dev),
                       torch.empty(B * groups * 2, dtype=torch.float32, device=dev))
    ws, stats = _GN_WS[key]
    if halo and to_conv is not None and conv_planes_ok(C, to_conv[0], to_conv[1]):
        # to_conv = (Cout, has_residual) of the 3x3 convolution that consumes this halo buffer: written as its operand planes
        hp = halo_planes_buffer(dev, B, H, W, C)
        call("sgic_groupnorm_nhwc_split3", _p(x), _p(gamma), _p(beta), B, H, W, C, groups, float(eps), int(swish), _p(ws), _p(stats),
             _p(hp.t))
        return hp
    if out is None:
        out = halo_buffer(dev, B, H, W, C) if halo else torch.empty(B * H * W, C, device=dev)
    call("sgic_groupnorm_nhwc", _p(x), _p(gamma), _p(beta), B, H, W, C, groups, float(eps), int(swish), int(halo), _p(ws),
         _p(stats), _p(out))
    return out


_HALO = {}


def halo_buffer(dev, B, H, W, C):
    """zero-halo NHWC buffer (B, H+2, W+2, C), cached per shape: producers only ever write the interior, so the
    border stays zero and no per-call memset is needed.  Uses are strictly sequential on one stream (a halo
    buffer is consumed by the conv right after it is produced)."""
    key = (str(dev), B, H, W, C)
    if key not in _HALO:
        _HALO[key] = torch.zeros(B, H + 2, W + 2, C, device=dev, dtype=torch.float32)
    return _HALO[key]


def halo_copy(x, B, H, W, C, upsample=False, tile16=False, out=None, to_conv=None):
    s = 2 if upsample else 1
    if to_conv is not None and conv_planes_ok(C, to_conv[0], to_conv[1]):
        hp = halo_planes_buffer(x.device, B, H * s, W * s, C)
        call("sgic_halo_copy_split3", _p(x), B, H, W, C, int(upsample), int(tile16), _p(hp.t))
        return hp
    if out is None:
        out = halo_buffer(x.device, B, H * s, W * s, C)
    call("sgic_halo_copy", _p(x), B, H, W, C, int(upsample), int(tile16), _p(out))
    return out


def softmax_rows(x, L, scale=1.0, out=None):
    assert x.is_contiguous()
    M = x.numel() // L
    if out is None:
        out = torch.empty_like(x)
    call("sgic_softmax_rows", _p(x), _p(out), _cl(M), L, float(scale))
    return out


def pixel_shuffle2_tm16(x, B, H, W, C):
    out = torch.empty(B * 4 * H * W, C, device=x.device, dtype=torch.float32)
    call("sgic_pixel_shuffle2_tm16", _p(x), B, H, W, C, _p(out))
    return out


def assemble_dec_tokens(emb, cls, mask, pos, latpos, N, P, T, D):
    out = torch.empty(N * (1 + P + T), D, device=emb.device, dtype=torch.float32)
    call("sgic_assemble_dec_tokens", _p(emb), _p(cls), _p(mask), _p(pos), _p(latpos), N, P, T, D, _p(out))
    return out


def codebook_gather_norm(idx, codebook, ld=None):
    M, dim = idx.numel(), codebook.shape[1]
    ld = ld or dim
    out = torch.empty(M, ld, device=idx.device, dtype=torch.float32)
    call("sgic_codebook_gather_norm", _p(idx), _p(codebook), M, dim, ld, _p(out))
    return out


def nhwc3_to_nchw_clamp(x, ld, B, H, W):
    out = torch.empty(B, 3, H, W, device=x.device, dtype=torch.float32)
    call("sgic_nhwc3_to_nchw_clamp", _p(x), ld, B, H, W, _p(out))
    return out


def rans_decode_init(streams, cap, off, ln, B):
    state = torch.zeros(B, 4, dtype=torch.int32, device=streams.device)
    call("sgic_rans_decode_init_batch", _p(streams), cap, _p(off), _p(ln), B, _p(state))
    return state


def rans_decode_step(table, streams, cap, off, ln, B, state, idx, n, idx_stride, out, out_stride):
    call("sgic_rans_decode_batch", table, _p(streams), cap, _p(off), _p(ln), B, _p(state), _p(idx), n, idx_stride, _p(out),
         out_stride)


def unpack12_batch(streams_u8, B, n):
    out = torch.empty(B, n, dtype=torch.int32, device=streams_u8.device)
    call("sgic_unpack12_batch", _p(streams_u8), B, n, _p(out))
    return out


def pad_replicate(x, pl, pr, pt, pb):
    """F.pad(x, (pl, pr, pt, pb), mode="replicate") for a contiguous (B, C, H, W) fp32 device tensor"""
    assert x.dim() == 4 and x.is_contiguous() and x.dtype == torch.float32
    B, C, H, W = x.shape
    if pl == pr == pt == pb == 0:
        return x
    out = torch.empty(B, C, H + pt + pb, W + pl + pr, device=x.device, dtype=torch.float32)
    call("sgic_pad_replicate", _p(x), _p(out), B * C, H, W, int(pl), int(pr), int(pt), int(pb))
    return out


def u8hwc_to_f32chw_pad(x_u8, pl=0, pr=0, pt=0, pb=0):
    """(B,H,W,3) u8 device tensor -> (B,3,H+pt+pb,W+pl+pr) fp32 in [-1,1]: ToTensor, *2-1, replicate pad (compress.py:161-164,258-261)"""
    assert x_u8.dim() == 4 and x_u8.shape[3] == 3 and x_u8.dtype == torch.uint8 and x_u8.is_contiguous() and x_u8.is_cuda
    B, H, W, _ = x_u8.shape
    out = torch.empty(B, 3, H + pt + pb, W + pl + pr, device=x_u8.device, dtype=torch.float32)
    call("sgic_u8hwc_to_f32chw_pad", _p(x_u8), _p(out), B, H, W, int(pl), int(pr), int(pt), int(pb))
    return out


def topk_rows(scores, k):
    """scores (nq, n) fp32 on device (consumed) -> (top scores (nq,k), indices (nq,k) int32)"""
    nq, n = scores.shape
    assert scores.is_contiguous()
    os_ = torch.empty(nq, k, device=scores.device, dtype=torch.float32)
    oi = torch.empty(nq, k, device=scores.device, dtype=torch.int32)
    call("sgic_topk_rows", _p(scores), nq, n, k, _p(os_), _p(oi))
    return os_, oi


def jpeg_decode_batch(params, scan, tabs, segs, quant, B, H, W, total_blocks, plane_bytes, max_blocks, out=None, check=True):
    """baseline JPEG batch -> (B,H,W,3) u8 on the device (csrc/jpeg.hip; descriptors built by sgic_amd.jpeg.JpegBatch)"""
    require_gpu()
    dev = torch.device("cuda", torch.cuda.current_device()) if not params.is_cuda else params.device   # inputs may be pinned host memory
    coef = torch.empty(total_blocks * 64, dtype=torch.int16, device=dev)
    wparams = torch.empty(B * 64, dtype=torch.int32, device=dev)
    wquant = torch.empty(B * 256, dtype=torch.int16, device=dev)
    planes = torch.empty(max(16, plane_bytes), dtype=torch.uint8, device=dev)
    if out is None:
        out = torch.empty(B, H, W, 3, dtype=torch.uint8, device=dev)
    assert out.shape == (B, H, W, 3) and out.dtype == torch.uint8 and out.is_contiguous()
    err = torch.empty(B, dtype=torch.int32, device=dev)
    call("sgic_jpeg_decode_batch", _p(params), _p(scan), _p(tabs), _p(segs), _p(quant), _p(wparams), _p(wquant), _p(coef), _p(planes), _p(out),
         _p(err), B, H, W, int(max_blocks))
    if check:
        e = err.cpu().numpy()
        if e.any():
            raise RuntimeError(f"corrupt JPEG stream(s) in the batch: error codes {e.tolist()}")
    return out, err
