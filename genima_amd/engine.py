"""Engine: the host-side launcher over the C ABI (include/genima_hip.h).

PyTorch-ROCm is used for device memory, streams and interop only: every op takes torch CUDA tensors, passes their
``data_ptr()`` + shapes to libgenima_hip.so and never touches a torch compute kernel.  Two modes:

  * eager  (``Engine(device)``):   each call enqueues the kernel immediately on the current stream (kernel parity tests);
  * record (``Engine(device, record=True)``): ops are appended to a ``gn_program`` with all buffers pre-allocated and
    name-keyed (so the unrolled denoise steps reuse one set of activation buffers); ``run()`` replays the whole program from
    C++ without returning to Python, ``capture()`` + ``launch()`` replay it as one hipGraph.

Layout conventions: activations NHWC f16 ``[B, H, W, C]`` == token-major ``[B, H*W, C]``; conv weights packed
``[Cout, KH*KW*Cin]`` (packing.py); Linear weights ``[out, in]``.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import json
import os
from typing import Dict, Optional, Tuple

import torch

from . import _lib
from ._lib import (ACT_GEGLU, ACT_GELU, ACT_NONE, ACT_QUICK_GELU, ACT_RELU, ACT_SILU, OUT_BATCH_TRANSPOSED, OUT_ROWMAJOR, TBLOCK_FRONT, TBLOCK_MID,
                   TBLOCK_TAIL, AttnDesc, ConvGnDesc, GemmDesc, GenimaHipError, GroupNormDesc, NormOut, StatsSink, TBlockDesc, TBlockTapeSrc, check)

F16 = torch.float16


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


TUNE_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_tune_gfx950.json")
_tune_cache: Optional[dict] = None
_tune_dirty = [False]


def _tune_table() -> dict:
    global _tune_cache
    if _tune_cache is None:
        try:
            with open(TUNE_PATH) as f:
                _tune_cache = {k: int(v) for k, v in json.load(f).items()}
        except Exception:
            _tune_cache = {}
    return _tune_cache


def save_tune_table():
    """Persist newly measured tile choices next to the package (and under gpurun_out/ so a GPU-box run can be committed)."""
    if not _tune_dirty[0]:
        return
    for path in (TUNE_PATH, os.path.join(os.path.dirname(os.path.dirname(TUNE_PATH)), "gpurun_out", "gemm_tune_gfx950.json")):
        try:
            if os.path.isdir(os.path.dirname(path)):
                tmp = f"{path}.{os.getpid()}.tmp"  # several ranks may save at once: each writes its own file, the rename is atomic
                with open(tmp, "w") as f:
                    json.dump(dict(sorted(_tune_table().items())), f, indent=0)
                os.replace(tmp, path)
        except OSError:
            pass
    _tune_dirty[0] = False


class Norm:
    """A GroupNorm (+ activation) standing in front of a conv / Linear: ``Engine.conv2d(x, w, ..., norm=Norm(...))`` normalises x on the way
    into the GEMM where the GroupNorm bridge applies, else runs the GroupNorm launch (buffer ``name``) first."""
    __slots__ = ("gamma", "beta", "groups", "eps", "act", "name")

    def __init__(self, gamma, beta, groups: int, eps: float, act: int = ACT_NONE, name: Optional[str] = None):
        self.gamma, self.beta, self.groups, self.eps, self.act, self.name = gamma, beta, int(groups), float(eps), int(act), name


class _Writer:
    """The recorded op that wrote a tensor (GroupNorm bridge: its statistics sink is attached when the consuming GroupNorm is recorded)."""
    __slots__ = ("op", "index", "numel", "sunk", "rdiv", "gemm")

    def __init__(self, op, index, numel, rdiv=1, gemm=False):
        self.op, self.index, self.numel, self.sunk, self.rdiv, self.gemm = op, index, numel, False, rdiv, gemm  # rdiv: its rows per sample = the tensor's / rdiv


class Engine:
    STATS_ARENA_BYTES = 16 << 20  # the GroupNorm bridge's statistics blocks of one recorded program (the used prefix is cleared at the top of every replay)
    STATS_LINE = 16               # int64 words per (replica, sample, group) line: GN_STATS_LINE

    def __init__(self, device="cuda:0", record: bool = False, autotune: Optional[bool] = None, gn_bridge: Optional[bool] = None):
        self.lib = _lib.load()
        self.autotune = (record or os.environ.get("GN_AUTOTUNE") == "1") if autotune is None else autotune
        self.hoist_time_shifts = os.environ.get("GN_HOIST_TIME_SHIFTS", "1") != "0"  # pipeline: all steps' time shifts in one pass (A/B switch)
        self.up_phases = os.environ.get("GN_UP_PHASES", "1") != "0"  # graphs: upsample + 3x3 conv as four 2x2 phase convs (A/B switch)
        self.up_phases_one_launch = os.environ.get("GN_UP_PHASES_ONE_LAUNCH", "1") != "0"
        self.up_phases_min_rows = int(os.environ.get("GN_UP_PHASES_MIN_ROWS", "1024"))  # source pixels x batch below which the 3x3 launch stays
        self.conv_gn = os.environ.get("GN_CONV_GN", "1") != "0"  # graphs: GroupNorm-apply + SiLU inside the consuming 3x3 conv (csrc/conv_gn.hip; A/B switch)
        # it pays where the GroupNorm launch is HBM-expensive against its conv (tools/bench_conv_gn.py, MI355X, B = 8): 512^2 x 128 -> 128
        # 1251 -> 915 us, 512^2 x 256 -> 128 2066 -> 1657 us; a wash at 256^2 x 256 (789 -> 774), a loss below (the SiLU of the 1.4x halo patch is
        # VALU time beside the MFMAs of a 128-wide output tile)
        self.conv_gn_min_hw = int(os.environ.get("GN_CONV_GN_MIN_HW", str(512 * 512)))
        # graphs: ResnetBlock2D's conv_shortcut inside conv2's K loop (gn_gemm_desc.k_append, packing `conv2sc`; A/B switch)
        self.k_append = os.environ.get("GN_K_APPEND", "1") != "0"
        self.k_append_min_rows = int(os.environ.get("GN_K_APPEND_MIN_ROWS", "0"))
        self.add_multi_on = os.environ.get("GN_ADD_MULTI", "1") != "0"  # graphs: the UNet's skip + ControlNet-residual adds as one launch
        self.zero_conv_split = os.environ.get("GN_ZERO_CONV_SPLIT", "1") != "0"  # deferred zero convs on both streams while the side stream is free
        self.side_free_max_rows = int(os.environ.get("GN_SIDE_FREE_MAX_ROWS", "4096"))  # decoder shortcuts on the idle side stream up to this many latent rows
        self.tblock = os.environ.get("GN_TBLOCK", "1") != "0"  # graphs: fused transformer-block chains at C = 320 (csrc/tblock.hip; A/B switch)
        # one workgroup per 128 rows streams the chain's whole weight tape: it pays once the rows fill the chip (tools/bench_tblock.py on MI355X:
        # tail 147 vs 201 us at 32768 rows, 121 vs 113 at 16384, 114 vs 64 at 8192)
        self.tblock_min_rows = int(os.environ.get("GN_TBLOCK_MIN_ROWS", "24576"))
        self.tblock_front_on = os.environ.get("GN_TBLOCK_FRONT", "1") != "0"  # the GroupNorm + proj_in + q | k | v chain (A/B switch of its own)
        self.ln_fold = os.environ.get("GN_LN_FOLD", "1") != "0"  # graphs: LayerNorm folded into the consuming Linear (A/B switch)
        # graphs: self-attention takes V row-major out of one plain q | k | v launch (gn_attn_desc.v_rowmajor) instead of the two-destination
        # launch + V^T.  Measured neutral in the call (107.59 vs 107.67 ms tiled b8, same box) although the kernel alone is 4-7 % faster at
        # 4096 keys: off by default, kept for hosts that hold V row-major
        self.rowmajor_v = os.environ.get("GN_ROWMAJOR_V", "0") == "1"
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise GenimaHipError("the Genima HIP engine needs a ROCm device (torch device 'cuda:N'); there is no CPU path")
        if not torch.cuda.is_available():
            raise GenimaHipError("no ROCm device visible to torch; the Genima HIP path has no CPU fallback")
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        self.stream = torch.cuda.current_stream(self.device)
        self._ctx = C.c_void_p()
        check(self.lib.gn_ctx_create(idx, C.c_void_p(self.stream.cuda_stream), C.byref(self._ctx)), "gn_ctx_create")
        self.record = record
        self._prog = C.c_void_p()
        if record:
            check(self.lib.gn_program_create(self._ctx, C.byref(self._prog)), "gn_program_create")
        self.buffers: Dict[str, torch.Tensor] = {}
        self._keep = []  # tensors referenced by recorded ops
        self._scope = []
        self._zpool = None  # eager: [pre-zeroed slab, next free element] while an emitter holds zero_pool()
        self.captured = False
        self.meta = []  # per recorded op: dict(kind, flops, bytes) -- algorithmic work for the roofline accounting
        # ---- the GroupNorm bridge (csrc/gn_bridge.h; recorded programs): statistics out of the op that writes a tensor, GroupNorm-apply inside
        # the op that reads it.  OFF by default (GN_BRIDGE=1 / Engine(gn_bridge=True) switch it on): measured on MI355X it does not pay inside the
        # call -- the producers' statistics tails (a reduction + a few device-scope atomics at the end of latency-bound launches: +3 us each at
        # B = 1, +8 us of 80 at B = 8) cost what the cheaper GroupNorm launches save (profiles/r05_v4_bridge_*)
        self.gn_bridge = record and (os.environ.get("GN_BRIDGE", "0") == "1" if gn_bridge is None else bool(gn_bridge))
        # rows (B x H x W) up to which the consuming conv / Linear normalises its own A tiles (ring kernels, gn_gemm_desc.norm_in); above, ONE
        # coalesced apply launch reads the producers' statistics (gn_groupnorm_desc.stats_in)
        self.gn_fuse_max_rows = int(os.environ.get("GN_BRIDGE_FUSE_MAX_ROWS", "0"))
        # the bridge pays where the single-launch GroupNorm (one workgroup per (sample, group) slab) leaves most of the chip idle -- the eval loop's
        # B = 1 call: 22 -> 8.6 us per 64 x 64 x 320 GroupNorm; at B = 8 the statistics tail costs the producing convs more (8 us of 80) than the
        # apply launch saves (profiles/r05_v3_bridge_*): gated by slabs = B x groups
        self.gn_bridge_max_slabs = int(os.environ.get("GN_BRIDGE_MAX_SLABS", "64"))
        # GroupNorm inside the split-K reduce of the launch that wrote its input (gn_gemm_desc.norm_out; recorded programs; GN_REDUCE_FUSE=0: off)
        self.gn_reduce_fuse = record and os.environ.get("GN_REDUCE_FUSE", "1") != "0"
        # one workgroup per (sample, group) slab gathers the partial slabs: it needs the slabs to fill the chip (B = 8: 256 workgroups; at B = 1 the
        # 32 of them lose to the parallel reduce + GroupNorm pair: 1024 x 640 x 5760 49 vs 29 + 12 us, profiles/r05_v7_reduce_gn_ops_*)
        self.gn_reduce_fuse_min_slabs = int(os.environ.get("GN_REDUCE_FUSE_MIN_SLABS", "128"))
        # ... except at the deepest latent levels (rows per sample <= this): there a slab is a few KB, the gather is short, and the GroupNorm launch it
        # replaces is pure latency (7 - 8 us each at B = 1: 185 launches of the tiled call at the 8 x 8 / 16 x 16 levels)
        self.gn_reduce_fuse_small_hw = int(os.environ.get("GN_REDUCE_FUSE_SMALL_HW", "64"))  # measured: profiles/r06_v4_gn_small_hw_ab.txt (single view 19.8 -> 19.3 ms, tiled B = 1 30.5 -> 30.2; 256 / 1024 lose on the tiled call)
        self._writer: Dict[int, _Writer] = {}
        self._stats_arena = None
        self._stats_used = 0
        if self.gn_bridge:
            self._stats_arena = torch.zeros(self.STATS_ARENA_BYTES // 8, dtype=torch.int64, device=self.device)
            self._memset_op = self.num_ops
            check(self.lib.gn_program_add_memset(self._prog, _ptr(self._stats_arena), self.STATS_ARENA_BYTES), "gn_program_add_memset")
            self.meta.append(dict(kind="memset", flops=0.0, bytes=float(self.STATS_ARENA_BYTES), shape=()))

    # ------------------------------------------------------------------------------------------------ housekeeping
    def __del__(self):
        try:
            if self._prog:
                self.lib.gn_program_destroy(self._prog)
            if self._ctx:
                self.lib.gn_ctx_destroy(self._ctx)
        except Exception:
            pass

    def use_stream(self, stream: torch.cuda.Stream):
        self.stream = stream
        check(self.lib.gn_ctx_set_stream(self._ctx, C.c_void_p(stream.cuda_stream)), "gn_ctx_set_stream")

    class _Scope:
        def __init__(self, eng, name):
            self.eng, self.name = eng, name

        def __enter__(self):
            self.eng._scope.append(self.name)

        def __exit__(self, *a):
            self.eng._scope.pop()

    def scope(self, name: str):
        return Engine._Scope(self, name)

    @contextlib.contextmanager
    def zero_pool(self, numel: int):
        """Eager mode: the zero-padded outputs requested inside the block (``buf(zero=True)``: transposed V^T matrices whose pad columns no
        kernel writes) are cut from ONE pre-zeroed f16 slab of ``numel`` elements (+ 64 per request) -- one fill launch per emitter instead
        of one per layer (the fine-tune step ran 39 of them per step in front of its CLIP / cross-attention projections).  Recording engines
        keep their persistent named buffers and ignore this."""
        if self.record or numel <= 0:
            yield
            return
        old = self._zpool
        self._zpool = [torch.zeros(int(numel), dtype=F16, device=self.device), 0]
        try:
            yield
        finally:
            self._zpool = old

    def buf(self, name: Optional[str], shape, dtype=F16, zero: bool = False) -> torch.Tensor:
        """Output buffer.  Eager: a fresh tensor.  Record: a persistent tensor keyed by the scoped name."""
        shape = tuple(int(s) for s in shape)
        if not self.record or name is None:
            pool = self._zpool
            if zero and pool is not None and dtype == F16:  # eager: a slice of the emitter's one pre-zeroed slab (zero_pool)
                n = 1
                for v in shape:
                    n *= v
                off = pool[1]
                if off + n <= pool[0].numel():
                    pool[1] = off + (n + 63) // 64 * 64
                    return pool[0][off:off + n].view(shape)
            return (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.device)
        key = "/".join(self._scope + [name])
        t = self.buffers.get(key)
        if t is None or tuple(t.shape) != shape or t.dtype != dtype:
            t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.device)
            self.buffers[key] = t
        self._wrote(t)  # whoever asks for an output buffer is about to write it
        return t

    def _wrote(self, *ts):
        """A recorded op is about to (over)write these tensors: forget who wrote them before -- a GroupNorm recorded later must not fuse into (or take
        statistics from) a producer whose output has been replaced since.  buf() calls it for every buffer it hands out; ops that take an
        explicit ``out=`` / work in place call it themselves."""
        if self._writer:
            for t in ts:
                if t is not None:
                    self._writer.pop(t.data_ptr(), None)

    def _keepalive(self, *ts):
        if self.record:
            self._keep.extend(t for t in ts if t is not None)

    @property
    def num_ops(self) -> int:
        return int(self.lib.gn_program_num_ops(self._prog)) if self.record else 0

    def _trim_memset(self):
        """The statistics arena's memset clears what the program came to use (at least one line), not the whole arena."""
        if self.gn_bridge and self._stats_arena is not None and not self.captured:
            used = max(self._stats_used * 8, 128)
            check(self.lib.gn_program_set_memset_bytes(self._prog, self._memset_op, used), "gn_program_set_memset_bytes")
            self.meta[self._memset_op]["bytes"] = float(used)

    def run(self, first: int = 0, last: int = -1):
        self._trim_memset()
        check(self.lib.gn_program_run(self._prog, first, last), "gn_program_run")

    def capture(self):
        self._trim_memset()
        check(self.lib.gn_program_capture(self._prog), "gn_program_capture")
        self.captured = True

    def launch(self):
        check(self.lib.gn_program_launch(self._prog), "gn_program_launch")

    def synchronize(self):
        check(self.lib.gn_stream_synchronize(self._ctx), "gn_stream_synchronize")

    # ------------------------------------------------------------------------------------------------ events
    def event(self):
        ev = C.c_void_p()
        check(self.lib.gn_event_create(C.byref(ev)), "gn_event_create")
        return ev

    def event_record(self, ev):
        check(self.lib.gn_event_record(self._ctx, ev), "gn_event_record")

    def event_elapsed_ms(self, a, b) -> float:
        ms = C.c_float()
        check(self.lib.gn_event_elapsed_ms(a, b, C.byref(ms)), "gn_event_elapsed_ms")
        return float(ms.value)

    # ------------------------------------------------------------------------------------------------ GEMM family
    # ---- per-shape tile autotuning ---------------------------------------------------------------------------------
    # The best block tile of the implicit-GEMM kernel depends on (M, N, K) and on how the tile grid fills 256 CUs / 8 XCDs in
    # ways a closed-form heuristic misses (tools/bench_gemm.py: up to 1.6x between configurations on hot-path shapes), so a
    # recording engine times every configuration once per distinct problem and remembers the winner.  Tile choice does not
    # change the per-element summation order (K is walked identically), only split-K does, and split-K is a deterministic
    # function of (shape, tile).  The table measured on MI355X ships as genima_amd/gemm_tune_gfx950.json.
    _retuned = set()  # shapes already re-raced in this process (GN_RETUNE)
    N_TILE_CFGS = 25  # 1..6 register-staged, 7..14 LDS-DMA, 15 ping-pong 256x256, 16..22 3-stage ring, 23 2-stage 128x160, 24 2-stage 128x320 on 8 waves, 25 persistent skewed ping-pong (csrc/gemm.hip kCfg, gemm_pp.hip, gemm_s3.hip, gemm_ppp.hip)

    @staticmethod
    def _tune_key(d: GemmDesc) -> str:
        key = "|".join(str(int(v)) for v in (d.conv, d.M, d.N, d.K, d.C1, d.C2, d.KH, d.stride, d.upsample2x, d.act,
                                             d.out_mode, 1 if d.residual else 0))
        if d.batch > 1:  # batched (attention backward) and accumulating (weight gradient) problems: suffixes keep older keys valid
            key += f"|b{int(d.batch)}"
        if d.accumulate:
            key += "|acc"
        if d.fp8:
            key += "|fp8"
        if d.out2:
            key += f"|o2{int(d.split_n)}"
        if d.ln_c1:
            key += "|ln"
        if d.k_append:
            key += "|ka"
        if d.norm_in.stats:
            key += "|gn"
        return key

    @staticmethod
    def _ppp_candidate(d: GemmDesc) -> bool:
        """Is the persistent skewed ping-pong tile (25, csrc/gemm_ppp.hip) worth racing?  (The library decides eligibility; this only keeps shapes it
        would map back onto tile 15 out of the race.)"""
        if d.M % 256 or d.N % 256 or d.K % 64 or d.K < 256 or d.out_mode != OUT_ROWMAJOR or d.out2 or d.fp8 or d.k_append or d.a2:
            return False
        if (d.act == ACT_GEGLU) != bool(d.ln_c1):  # the feed-forward variant takes the LayerNorm fold and GEGLU together
            return False
        if d.batch > 1 and not d.up_phases:
            return False
        return (d.M // 256) * (d.N // 256) * (4 if d.up_phases else 1) >= 256

    @staticmethod
    def apply_plan(d: GemmDesc, plan: int):
        """A tune-table value is ``tile + 100 * splitk`` (splitk 0 = the library's heuristic for that tile)."""
        d.tile, d.splitk = int(plan) % 100, int(plan) // 100
        cap = int(os.environ.get("GN_PROBE_SPLITK_CAP", "0"))  # probe: 1 = never split K, n = at most n slices where the table names more
        if cap == 1 or (cap > 1 and d.splitk > cap):
            d.splitk = cap

    def _autotune(self, d: GemmDesc, key: Optional[str] = None) -> int:
        """-> the plan (``tile + 100 * splitk``) of this problem: from the table, or measured now and remembered."""
        key = key or self._tune_key(d)
        table = _tune_table()
        challengers = [int(c) for c in os.environ.get("GN_RETUNE", "").split(",") if c.strip()]  # e.g. GN_RETUNE=15: race new tiles
        sk_chal = [int(c) for c in os.environ.get("GN_RETUNE_SK", "").split(",") if c.strip()]      # e.g. GN_RETUNE_SK=10,12,16: race deeper K splits
        if key in table and not ((challengers or sk_chal) and key not in self._retuned):          # against each shape's incumbent
            return table[key]
        resplit = key in table and bool(sk_chal)
        cands = (1, 2, 5, 6, 7, 8, 9, 12, 16, 19, 25) if d.act == ACT_GEGLU else range(1, self.N_TILE_CFGS + 1)
        if key in table:
            self._retuned.add(key)
            cands = [table[key]] + [c for c in challengers if c != table[key] % 100 and (d.act != ACT_GEGLU or c in (1, 2, 5, 6, 7, 8, 9, 12, 16, 19, 25))]
        if not self._ppp_candidate(d):  # tile 25 needs whole 256 x 256 tiles, at least one per CU (the library would run tile 15 instead: no second race of it)
            cands = [c for c in cands if c % 100 != 25]
        if d.fp8:  # the fp8 kernel exists for the six LDS-DMA block tiles 256x256 .. 256x64
            cands = (7, 8, 9, 12) if d.act == ACT_GEGLU else range(7, 13)
        if d.ln_c1:  # the LayerNorm fold lives in the LDS-DMA kernels (the library maps the other tiles onto them)
            cands = [c for c in cands if c % 100 >= 7 and c % 100 != 15]
        if d.k_append:  # so does the appended 1x1 segment
            cands = [c for c in cands if c % 100 >= 7]
        if d.norm_in.stats:  # the normalising A path lives in the ring kernels; a row tile spans at most four samples
            rps = d.Ho * d.Wo if d.conv else d.norm_in.rows_per_sample
            bmn = {16: (128, 128), 17: (128, 64), 18: (64, 64), 19: (256, 64), 20: (128, 160), 21: (64, 160), 22: (64, 320)}
            ct = (d.C1 if d.k_append else d.C1 + d.C2) if d.conv else (d.K - d.C2 if d.k_append else d.K)

            def fits(c):  # the ring + the scale / shift table of the samples a row tile touches inside the CU's 160 KB of LDS
                bm, bn = bmn[c]
                return bm <= 4 * rps and 3 * (bm + bn) * 128 + max(1, bm // rps) * ct * 8 <= 160 * 1024
            cands = [c for c in bmn if fits(c) and (not challengers or c in [table.get(key, 0) % 100] + challengers)]
        e0, e1 = self.event(), self.event()

        def race(plan: int) -> float:
            self.apply_plan(d, plan)
            nb = int(self.lib.gn_gemm_workspace_bytes(C.byref(d)))
            d.workspace = self._workspace(nb).data_ptr() if nb > 0 else None
            check(self.lib.gn_gemm(self._ctx, C.byref(d)), "gn_gemm(autotune)")
            ms = float("inf")
            for _rep in range(3 if challengers else 2):  # min of the timed batches: one stray hiccup must not decide the table
                self.event_record(e0)
                for _ in range(3):
                    check(self.lib.gn_gemm(self._ctx, C.byref(d)), "gn_gemm(autotune)")
                self.event_record(e1)
                ms = min(ms, self.event_elapsed_ms(e0, e1))
            return ms

        best, best_ms = 0, float("inf")
        for c in cands:
            ms = race(c)
            if ms < best_ms:
                best, best_ms = c, ms
        # K splits: the library's heuristic aims at ~1.5 workgroups per CU; for the long-K, few-tile launches (the 16x16 / 8x8 latent
        # levels) the right count is whatever makes the grid fit the 256 CUs.  Raced for the winning tile and -- because its heuristic
        # split (made for the 2-3-workgroups-per-CU tiles) can leave the one-workgroup-per-CU ping-pong tile a 1.6-round grid that loses
        # the tile race although 256 x 256 with the right split wins (conv 1280 -> 1280 @ 16 x 16: tile 9 92 us, tile 15 / 5 slices 78) --
        # for tile 15 as well wherever its output grid is smaller than the chip.  3 % margin against noise.
        if d.K >= 1024 and d.act != ACT_GEGLU and d.out_mode == OUT_ROWMAJOR and d.batch <= 1 and not d.fp8 and not d.out2 and not d.ln_c1:  # gn_gemm pins sk = 1 under ln_c1
            tiles = [best % 100]
            pp_blocks = -(-d.M // 256) * -(-d.N // 256)
            if 15 in cands and 15 not in tiles and d.K >= 2048 and d.K % 64 == 0 and pp_blocks < 128 and not d.norm_in.stats:
                tiles.append(15)
            for tile in tiles:
                for sk in (sk_chal if resplit else (1, 2, 3, 4, 5, 6, 8)):
                    if sk * 512 > d.K or tile + 100 * sk == best or (tile == 15 and tile != tiles[0] and not 128 < pp_blocks * sk <= 256):
                        continue
                    ms = race(tile + 100 * sk)
                    if ms < 0.97 * best_ms:
                        best, best_ms = tile + 100 * sk, ms
        self.lib.gn_event_destroy(e0)
        self.lib.gn_event_destroy(e1)
        table[key] = best
        _tune_dirty[0] = True
        return best

    def _gemm(self, d: GemmDesc, keep):
        if d.tile == 0 and d.splitk == 0:
            # autotuning engines measure unknown shapes; every engine uses a tile that was already measured on this architecture
            self.apply_plan(d, self._autotune(d) if self.autotune else (0 if getattr(self, "no_table", False) else _tune_table().get(self._tune_key(d), 0)))
        ws_bytes = int(self.lib.gn_gemm_workspace_bytes(C.byref(d)))
        ws = None
        if ws_bytes > 0:
            ws = self._workspace(ws_bytes)
            d.workspace = ws.data_ptr()
        if self.record:
            if (d.out_mode == OUT_ROWMAJOR and not d.out2 and d.act != ACT_GEGLU and not d.fp8 and (d.batch <= 1 or d.up_phases)
                    and not d.out_row_width or d.up_phases):
                # the op that wrote this tensor: a GroupNorm recorded later may move into its reduce (norm_out) or take its statistics (bridge)
                # (a phase conv launch writes the whole upsampled tensor: 4 phases x M rows)
                if d.up_phases or int(d.ldo) == int(d.N):  # a dense tensor (a slice of a wider buffer is not what a GroupNorm reads)
                    self._writer[int(d.out)] = _Writer(self.num_ops, 0, int(d.M) * int(d.N) * (4 if d.up_phases else 1), 4 if d.up_phases else 1, gemm=True)
            check(self.lib.gn_program_add_gemm(self._prog, C.byref(d)), "gn_program_add_gemm")
            self._keepalive(*keep, ws)
            kind = (f"conv{d.KH}x{d.KW}" if d.conv else "linear")
            n_out = d.N // 2 if d.act == ACT_GEGLU else d.N
            # a phase conv of an upsampling 3x3 conv (KH = 2, strided output view) does 4 / 9 of the reference algorithm's multiply-adds
            ref_scale = 9.0 / 4.0 if (d.conv and d.KH == 2 and d.out_row_width) else 1.0
            nph = 4.0 if d.up_phases else 1.0  # the four phases of an upsampling conv in one launch
            self.meta.append(dict(kind=kind, flops=nph * 2.0 * d.M * d.N * d.K,
                                  bytes=2.0 * (d.M * d.K / max(1, d.KH * d.KW if d.conv else 1) + nph * d.N * d.K + nph * d.M * n_out),
                                  shape=(int(d.M), int(d.N), int(d.K)), ref_flops=nph * 2.0 * d.M * d.N * d.K * ref_scale))
        else:
            self.run_gemm(d)

    def run_gemm(self, d: GemmDesc):
        """Eager launch; with ``self.gemm_log`` set (a list) each launch is bracketed by HIP events for the per-shape tables."""
        log = getattr(self, "gemm_log", None)
        if log is None:
            check(self.lib.gn_gemm(self._ctx, C.byref(d)), "gn_gemm")
            return
        e0, e1 = self.event(), self.event()
        self.event_record(e0)
        check(self.lib.gn_gemm(self._ctx, C.byref(d)), "gn_gemm")
        self.event_record(e1)
        log.append((self._tune_key(d) + f"|t{int(d.tile)}", 2.0 * d.M * d.N * d.K * max(1, d.batch), e0, e1))

    def gemm_log_report(self):
        """-> {key: (calls, total ms, TFLOP/s)} from the events collected in ``self.gemm_log`` (synchronises)."""
        self.synchronize()
        agg = {}
        for key, fl, e0, e1 in self.gemm_log:
            ms = self.event_elapsed_ms(e0, e1)
            c = agg.setdefault(key, [0, 0.0, 0.0])
            c[0] += 1
            c[1] += ms
            c[2] += fl
            self.lib.gn_event_destroy(e0)
            self.lib.gn_event_destroy(e1)
        self.gemm_log = []
        return {k: (c[0], c[1], c[2] / (c[1] * 1e9) if c[1] > 0 else 0.0) for k, c in agg.items()}

    def _workspace(self, nbytes: int) -> torch.Tensor:
        """f32 split-K / GroupNorm scratch: one shared grow-only buffer per stream (ops on one stream run in order)."""
        n = _round_up(nbytes, 256) // 4
        # one buffer per stream: `_on_side` is False (the main stream), True (a recorded program's side stream) or a tag naming one of the
        # trainer's extra streams -- they run CONCURRENTLY with each other (the next step's front starts under the previous step's backward)
        side = getattr(self, "_on_side", False)
        key = "__workspace__" if not side else ("__workspace_side__" if side is True else f"__workspace_{side}__")
        cur = self.buffers.get(key)
        if cur is None or cur.numel() < n:
            if self.record and cur is not None:
                self._keep.append(cur)  # earlier recorded ops still point at the old buffer
            cur = torch.empty(max(n, 1 << 20), dtype=torch.float32, device=self.device)
            self.buffers[key] = cur
        return cur

    # ---- two-stream sections of a recorded program (gn_program_add_fork / main / join); no-ops when executing eagerly
    def fork(self):
        """Ops recorded from here run on the program's side stream (after everything recorded so far)."""
        if self.record:
            check(self.lib.gn_program_add_fork(self._prog), "gn_program_add_fork")
            self.meta.append(dict(kind="stream", op="fork", flops=0.0, bytes=0.0, shape=()))  # meta stays index-aligned with the op list
            self._on_side = True

    def main(self):
        """Back to the main stream; the side stream keeps running concurrently until join()."""
        if self.record:
            check(self.lib.gn_program_add_main(self._prog), "gn_program_add_main")
            self.meta.append(dict(kind="stream", op="main", flops=0.0, bytes=0.0, shape=()))
            self._on_side = False

    def join(self):
        if self.record:
            check(self.lib.gn_program_add_join(self._prog), "gn_program_add_join")
            self.meta.append(dict(kind="stream", op="join", flops=0.0, bytes=0.0, shape=()))
            self._on_side = False

    # ---- the GroupNorm bridge: producer statistics for a GroupNorm over x (| x2) ---------------------------------------------------------
    def bridge_stats(self, x: torch.Tensor, x2: Optional[torch.Tensor], groups: int) -> Optional[torch.Tensor]:
        """-> the int64 [B, groups, 2] statistics block that the recorded ops which wrote x (and x2) will fill, or None where the bridge does
        not apply (eager engine, an input this program did not write with a gn_gemm / gn_add_multi, a producer that already feeds another
        GroupNorm, arena full): the caller then runs the GroupNorm's own statistics passes."""
        if not self.gn_bridge:
            return None
        C1 = x.shape[-1]
        C2 = x2.shape[-1] if x2 is not None else 0
        B = x.shape[0]
        if (C1 + C2) % groups or ((C1 + C2) // groups) % 2 or B * groups > self.gn_bridge_max_slabs:
            return None
        rps = x.numel() // (B * C1)
        srcs = [(x, 0)] + ([(x2, C1)] if x2 is not None else [])
        ws = []
        for t, _ in srcs:
            w = self._writer.get(t.data_ptr())
            if w is None or w.sunk or w.numel != t.numel() or not t.is_contiguous() or rps % w.rdiv:
                return None
            ws.append(w)
        # a 128-byte line per (replica, sample, group); few samples = few lines for the producers' atomics to serialise on: replicas
        R = max(1, 8 // B)
        n = R * B * groups * self.STATS_LINE
        if (self._stats_used + n) * 8 > self.STATS_ARENA_BYTES:
            return None
        slot = self._stats_arena[self._stats_used:self._stats_used + n].view(R, B, groups, self.STATS_LINE)
        self._stats_used += n
        for (t, coff), w in zip(srcs, ws):
            sk = StatsSink()
            sk.stats, sk.cpg, sk.coff, sk.groups, sk.rows_per_sample = slot.data_ptr(), (C1 + C2) // groups, coff, groups, rps // w.rdiv
            sk.samples, sk.replicas = B, R
            check(self.lib.gn_program_set_sink(self._prog, w.op, w.index, C.byref(sk), t.shape[-1]), "gn_program_set_sink")
            w.sunk = True
        return slot

    @staticmethod
    def _set_sink(d: GemmDesc, sink):
        """sink = (stats int64 [B, groups, 2] (zeroed), cpg, coff, rows_per_sample): explicit producer side of the bridge (eager calls / tests)."""
        if sink is not None:
            st, cpg, coff, rps = sink  # st: int64 [replicas, samples, groups, 16]
            d.sink.stats, d.sink.cpg, d.sink.coff, d.sink.groups, d.sink.rows_per_sample = st.data_ptr(), int(cpg), int(coff), int(st.shape[2]), int(rps)
            d.sink.samples, d.sink.replicas = int(st.shape[1]), int(st.shape[0])

    def _set_norm_out(self, d: GemmDesc, norm_out: Optional["Norm"], out: torch.Tensor, rows_per_sample: int):
        """Explicit gn_gemm_desc.norm_out (eager calls / tests): -> the normalised output tensor, or None.  The plan must split K."""
        if norm_out is None:
            return None
        y = self.buf(norm_out.name, out.shape)
        n = d.norm_out
        n.y, n.gamma, n.beta, n.eps, n.groups, n.act, n.rows_per_sample = _ptr(y), _ptr(norm_out.gamma), _ptr(norm_out.beta), norm_out.eps, norm_out.groups, norm_out.act, int(rows_per_sample)
        return y

    def _norm_in(self, d: GemmDesc, x: torch.Tensor, x2: Optional[torch.Tensor], norm: "Norm", rows: int, stats: Optional[torch.Tensor] = None) -> bool:
        """Try to put ``norm`` (a GroupNorm over x | x2) inside the gn_gemm ``d`` reads them with (gn_gemm_desc.norm_in).  -> done?
        ``stats``: an explicit, already filled statistics block (eager calls / tests) instead of the recorded producers'."""
        if stats is None and (not self.gn_bridge or rows > self.gn_fuse_max_rows or not getattr(self, "gn_fuse", True)):
            return False
        C1 = x.shape[-1]
        C2 = x2.shape[-1] if x2 is not None else 0
        B = x.shape[0]
        ni = d.norm_in
        ni.gamma, ni.beta, ni.eps, ni.groups, ni.cpg, ni.act = _ptr(norm.gamma), _ptr(norm.beta), norm.eps, norm.groups, (C1 + C2) // norm.groups, norm.act
        ni.rows_per_sample = x.numel() // (B * C1)
        ni.samples, ni.replicas = B, (int(stats.shape[0]) if stats is not None else max(1, 8 // B))
        ni.stats = None
        if (C1 + C2) % norm.groups or not self.lib.gn_gemm_norm_in_supported(C.byref(d)):
            if stats is not None:
                raise GenimaHipError("norm_in: unsupported problem (gn_gemm_norm_in_supported)")
            return False
        st = stats if stats is not None else self.bridge_stats(x, x2, norm.groups)
        if st is None:
            return False
        ni.stats = st.data_ptr()
        return True

    def linear(self, x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, *, act: int = ACT_NONE,
               residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, name: Optional[str] = None,
               transposed_out: bool = False, rows_per_batch: int = 0, pad_cols: int = 0, splitk: int = 0,
               split_n: int = 0, out2: Optional[torch.Tensor] = None, ln_c1: Optional[torch.Tensor] = None, ln_eps: float = 1e-5,
               append: Optional[torch.Tensor] = None, norm: Optional[Norm] = None, sink=None, norm_stats: Optional[torch.Tensor] = None,
               norm_out: Optional[Norm] = None):
        """y = act(x @ w.T + bias) (+ residual).  x: [..., K] contiguous f16, w: [N, K].
        transposed_out: y[b, n, m_local] with row stride ``pad_cols`` (>= rows_per_batch; V^T for the attention kernel).
        split_n > 0: ONE launch with two destinations (the q | k | v projections of a self-attention block): columns [0, split_n)
        row-major -> y [.., split_n], columns [split_n, N) batch-transposed -> y2 [b, N - split_n, pad_cols]; returns (y, y2).
        ln_c1 (f32 [N]): LayerNorm folded into this Linear -- x holds the RAW rows, w the gamma-scaled weight, bias c2 (packing.fold_layernorm;
        gn_gemm_desc.ln_c1): y = act(LayerNorm(x) @ W.T + b) without the LayerNorm launch or its round trip through memory."""
        K = x.shape[-1]
        M = x.numel() // K
        N = w.shape[0]
        if append is not None:  # y = [x | append] @ w.T: gn_gemm_desc.k_append (dense) -- two Linears without a nonlinearity between them as one
            assert ln_c1 is None and append.numel() // append.shape[-1] == M and x.shape[-1] % 64 == 0
            K += append.shape[-1]
        assert w.shape[1] == K, (w.shape, x.shape)
        fp8w = self._fp8_weights.get(w.data_ptr()) if self._fp8_weights else None
        if fp8w is not None and append is None and not transposed_out and splitk == 0 and M >= self.fp8_min_rows and not self.record:
            xq, xs = self._fp8_activation(x)
            return self.linear_fp8(xq, xs, fp8w[0], fp8w[1], bias, act=act, residual=residual, out=out, name=name)
        n_out = N // 2 if act == ACT_GEGLU else N
        d = GemmDesc()
        if split_n:
            assert not transposed_out and rows_per_batch > 0 and M % rows_per_batch == 0 and 0 < split_n < N
            nb = M // rows_per_batch
            ld2 = pad_cols if pad_cols else _round_up(rows_per_batch, 64)
            if out is None:
                out = self.buf(name, tuple(x.shape[:-1]) + (split_n,))
            if out2 is None:
                out2 = self.buf(None if name is None else name + ".t", (nb, N - split_n, ld2), zero=ld2 != rows_per_batch)  # only pad columns need the zeros
            d.out_mode, d.ldo, d.rows_per_batch = OUT_ROWMAJOR, out.stride(-2), rows_per_batch
            d.out2, d.ldo2, d.split_n = _ptr(out2), ld2, split_n
        elif transposed_out:
            assert rows_per_batch > 0 and M % rows_per_batch == 0
            nb = M // rows_per_batch
            ld = pad_cols if pad_cols else _round_up(rows_per_batch, 64)
            if out is None:
                out = self.buf(name, (nb, N, ld), zero=ld != rows_per_batch)  # only pad columns need the zeros
            d.out_mode, d.ldo, d.rows_per_batch = OUT_BATCH_TRANSPOSED, ld, rows_per_batch
        else:
            if out is None:
                out = self.buf(name, tuple(x.shape[:-1]) + (n_out,))
            d.out_mode, d.ldo = OUT_ROWMAJOR, out.stride(-2) if out.dim() > 1 else n_out
        d.a, d.w, d.bias, d.residual, d.out = _ptr(x), _ptr(w), _ptr(bias), _ptr(residual), _ptr(out)
        d.M, d.N, d.K = M, N, K
        d.lda, d.ldw = x.stride(-2) if x.dim() > 1 else K, w.stride(0)
        if append is not None:
            d.k_append, d.a2, d.C2, d.lda2 = 1, _ptr(append), append.shape[-1], append.stride(-2)
        d.ldr = residual.stride(-2) if residual is not None else 0
        d.act, d.splitk, d.out_scale = act, splitk, 1.0
        if ln_c1 is not None:
            assert bias is not None and ln_c1.dtype == torch.float32 and ln_c1.numel() == N and not transposed_out
            d.ln_c1, d.ln_eps = _ptr(ln_c1), float(ln_eps)
        if norm is not None:  # x: RAW [B, rows, K] tokens of a tensor a GroupNorm stands in front of (Transformer2DModel.norm -> proj_in)
            assert x.dim() == 3 and append is None and ln_c1 is None
            if not self._norm_in(d, x, None, norm, M, norm_stats):
                d.norm_in.stats = None
                n = self.groupnorm(x, norm.gamma, norm.beta, norm.groups, norm.eps, act=norm.act, name=norm.name)
                d.a, d.lda = _ptr(n), n.stride(-2)
                x = n
        self._set_sink(d, sink)
        y = self._set_norm_out(d, norm_out, out, rows_per_batch if rows_per_batch else (x.shape[1] if x.dim() == 3 else M))
        self._gemm(d, (x, w, bias, residual, out, out2, ln_c1, append, None if norm is None else norm.gamma, None if norm is None else norm.beta, y))
        if y is not None:
            return out, y
        return (out, out2) if split_n else out

    # ---- fused chains of a transformer block's Linears (csrc/tblock.hip): one launch keeps 128 rows of the residual stream in LDS ----------
    def tblock_supported(self, M: int, C: int) -> bool:
        return bool(self.lib.gn_tblock_supported(TBLOCK_TAIL, int(M), int(C)))

    def _tblock(self, d: TBlockDesc, keep, flops: float, nbytes: float):
        if self.record:
            check(self.lib.gn_program_add_tblock(self._prog, C.byref(d)), "gn_program_add_tblock")
            self._keepalive(*keep)
            self.meta.append(dict(kind="tblock", flops=flops, bytes=nbytes, shape=(int(d.M), int(d.C), int(d.kind)), ref_flops=flops))
        else:
            check(self.lib.gn_tblock(self._ctx, C.byref(d)), "gn_tblock")

    def tblock_front(self, x: torch.Tensor, scsh: torch.Tensor, tape: torch.Tensor, rows_per_batch: int, *, ln_eps: float = 1e-5,
                     name: Optional[str] = None):
        """(h, qk, vt): h = GroupNorm(x) @ Wi.T + bi with the GroupNorm applied from its (scale, shift) pairs ``scsh`` (groupnorm_stats) on the
        rows in LDS; q | k = LayerNorm1(h) @ [Wq | Wk].T -> qk [.., 2C]; V^T [B, C, pad64(rows_per_batch)] -- the operands of the self-attention
        kernel.  One launch (GN_TBLOCK_FRONT; ``tape`` = packing.pack_tblock_front_tape)."""
        Cc = x.shape[-1]
        M = x.numel() // Cc
        nb = M // rows_per_batch
        ld = _round_up(rows_per_batch, 64)
        h = self.buf(name, x.shape)
        qk = self.buf(None if name is None else name + ".qk", tuple(x.shape[:-1]) + (2 * Cc,))
        vt = self.buf(None if name is None else name + ".vt", (nb, Cc, ld), zero=ld != rows_per_batch)
        d = TBlockDesc()
        d.kind, d.C, d.M = TBLOCK_FRONT, Cc, M
        d.a, d.scsh, d.out, d.out2, d.out3, d.tape, d.tape_bytes = _ptr(x), _ptr(scsh), _ptr(h), _ptr(qk), _ptr(vt), _ptr(tape), tape.numel() * tape.element_size()
        d.lda, d.ldo, d.ldo2, d.ldo3, d.rows_per_batch, d.ln_eps = x.stride(-2), h.stride(-2), qk.stride(-2), ld, rows_per_batch, float(ln_eps)
        self._tblock(d, (x, scsh, h, qk, vt, tape), 2.0 * M * Cc * Cc * 4, 2.0 * M * Cc * 5 + tape.numel())
        return h, qk, vt

    def tblock_mid(self, a: torch.Tensor, res: torch.Tensor, tape: torch.Tensor, *, ln_eps: float = 1e-5, name: Optional[str] = None):
        """(h1, q): h1 = a @ Wo.T + bo + res (attn1.to_out.0 + residual), q = LayerNorm2(h1) @ Wq.T (attn2.to_q) -- one launch
        (gn_tblock_desc GN_TBLOCK_MID; ``tape`` = packing.pack_tblock_mid_tape).  a, res: [..., 320] f16 contiguous rows."""
        Cc = a.shape[-1]
        M = a.numel() // Cc
        h1 = self.buf(name, a.shape)
        q = self.buf(None if name is None else name + ".q", a.shape)
        d = TBlockDesc()
        d.kind, d.C, d.M = TBLOCK_MID, Cc, M
        d.a, d.res1, d.out, d.out2, d.tape, d.tape_bytes = _ptr(a), _ptr(res), _ptr(h1), _ptr(q), _ptr(tape), tape.numel() * tape.element_size()
        d.lda, d.ldr1, d.ldo, d.ldo2, d.ln_eps = a.stride(-2), res.stride(-2), h1.stride(-2), q.stride(-2), float(ln_eps)
        self._tblock(d, (a, res, h1, q, tape), 2.0 * M * Cc * Cc * 2, 2.0 * M * Cc * 4 + tape.numel())
        return h1, q

    def tblock_tail(self, a: torch.Tensor, res: torch.Tensor, res_out: torch.Tensor, tape: torch.Tensor, *, ln_eps: float = 1e-5,
                    name: Optional[str] = None) -> torch.Tensor:
        """out = proj_out(ff(h2) + h2) + res_out with h2 = a @ Wo.T + bo + res (attn2.to_out.0 + residual; norm3 + GEGLU feed-forward +
        residual; proj_out + the transformer's input) -- one launch (GN_TBLOCK_TAIL; ``tape`` = packing.pack_tblock_tail_tape)."""
        Cc = a.shape[-1]
        M = a.numel() // Cc
        out = self.buf(name, a.shape)
        d = TBlockDesc()
        d.kind, d.C, d.M = TBLOCK_TAIL, Cc, M
        d.a, d.res1, d.res2, d.out, d.tape, d.tape_bytes = _ptr(a), _ptr(res), _ptr(res_out), _ptr(out), _ptr(tape), tape.numel() * tape.element_size()
        d.lda, d.ldr1, d.ldr2, d.ldo, d.ln_eps = a.stride(-2), res.stride(-2), res_out.stride(-2), out.stride(-2), float(ln_eps)
        self._tblock(d, (a, res, res_out, out, tape), 2.0 * M * Cc * Cc * (2 + 8 + 4), 2.0 * M * Cc * 4 + tape.numel())
        return out

    # ---- fp8 (OCP e4m3) Linear: SURVEY section 8 a15 / BASELINE configs[4] "fp8 MFMA" -------------------------------------------
    _fp8_weights = None   # {weight data_ptr: (bytes, scales, weight)} of the Linears that run on the fp8 MFMA (enable_fp8)
    fp8_min_rows = 1024   # below this the Linear is launch-bound and the extra quantisation launch does not pay

    def enable_fp8(self, weights):
        """Route every later ``linear(x, w)`` whose ``w`` is one of ``weights`` ([N, K] f16, frozen) through the fp8 MFMA: the
        weights are quantised once (per-output-channel scales), activations per call (per-token scales, cached while the same
        tensor feeds several Linears, e.g. q / k / v).  Eager engines only."""
        table = {} if self._fp8_weights is None else self._fp8_weights
        for w in weights:
            if w.dim() == 2 and w.shape[1] % 8 == 0 and w.shape[0] % 4 == 0 and w.data_ptr() not in table:
                wq, ws = self.quantize_fp8(w)
                table[w.data_ptr()] = (wq, ws, w)  # holding w keeps its address from being reused
        self._fp8_weights = table
        self._fp8_act = None

    def _fp8_activation(self, x: torch.Tensor):
        c = self._fp8_act
        if c is not None and c[0] is x:  # the held reference keeps x's storage from being recycled under the cache
            return c[1], c[2]
        xq, xs = self.quantize_fp8(x)
        self._fp8_act = (x, xq, xs)
        return xq, xs

    def quantize_fp8(self, x: torch.Tensor, *, name: Optional[str] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """Row-wise dynamic e4m3 quantisation of x [..., K] f16 -> (bytes [..., round_up(K, 16)] uint8, scales [rows] f32).
        Activations: per-token scales; a weight [N, K]: per-output-channel scales (quantise once, keep)."""
        K = x.shape[-1]
        rows = x.numel() // K
        Kp = _round_up(K, 16)
        q = self.buf(None if name is None else name + ".q", tuple(x.shape[:-1]) + (Kp,), dtype=torch.uint8)
        s = self.buf(None if name is None else name + ".s", (_round_up(rows, 4),), dtype=torch.float32)
        if self.record:
            raise GenimaHipError("quantize_fp8 is an eager op (the fp8 Linear serves the training forward, which is not recorded)")
        self._small("quantize_fp8_rows", (x, q, s), _ptr(x), x.stride(-2) if x.dim() > 1 else K, rows, K, _ptr(q), Kp, _ptr(s))
        return q, s

    def linear_fp8(self, xq: torch.Tensor, xs: torch.Tensor, wq: torch.Tensor, ws: torch.Tensor,
                   bias: Optional[torch.Tensor] = None, *, act: int = ACT_NONE, residual: Optional[torch.Tensor] = None,
                   out: Optional[torch.Tensor] = None, name: Optional[str] = None) -> torch.Tensor:
        """y = act(dequant(xq) @ dequant(wq).T + bias) (+ residual) on the fp8 MFMA.  xq [..., Kp] / wq [N, Kp] uint8 e4m3 bytes
        with their row scales xs / ws (quantize_fp8); y f16 [..., N]."""
        Kp = xq.shape[-1]
        M = xq.numel() // Kp
        N = wq.shape[0]
        assert wq.shape[1] == Kp and xq.dtype == torch.uint8 and wq.dtype == torch.uint8, (xq.shape, wq.shape)
        n_out = N // 2 if act == ACT_GEGLU else N
        if out is None:
            out = self.buf(name, tuple(xq.shape[:-1]) + (n_out,))
        else:
            self._wrote(out)
        d = GemmDesc()
        d.a, d.w, d.bias, d.residual, d.out = _ptr(xq), _ptr(wq), _ptr(bias), _ptr(residual), _ptr(out)
        d.M, d.N, d.K = M, N, Kp
        d.lda, d.ldw = xq.stride(-2) if xq.dim() > 1 else Kp, wq.stride(0)
        d.ldo = out.stride(-2) if out.dim() > 1 else n_out
        d.ldr = residual.stride(-2) if residual is not None else 0
        d.act, d.out_mode, d.out_scale = act, OUT_ROWMAJOR, 1.0
        d.fp8, d.scale_a, d.scale_w = 1, _ptr(xs), _ptr(ws)
        self._gemm(d, (xq, xs, wq, ws, bias, residual, out))
        return out

    def conv2d(self, x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, *, ksize: int = 3,
               stride: int = 1, pad: Tuple[int, int, int, int] = None, x2: Optional[torch.Tensor] = None,
               shift: Optional[torch.Tensor] = None, ldshift: int = 0, residual: Optional[torch.Tensor] = None,
               act: int = ACT_NONE, upsample2x: bool = False, out_scale: float = 1.0, out: Optional[torch.Tensor] = None,
               name: Optional[str] = None, splitk: int = 0, residual_before_act: bool = False, up_phases: bool = False,
               append: Optional[torch.Tensor] = None, append2: Optional[torch.Tensor] = None, norm: Optional[Norm] = None, sink=None,
               norm_stats: Optional[torch.Tensor] = None, norm_out: Optional[Norm] = None) -> torch.Tensor:
        """``append`` [B, H, W, C2] (+ ``append2`` [B, H, W, C3], the rest of a concatenated input): a 1x1 conv appended along K
        (gn_gemm_desc.k_append) -- w = [Cout, k*k*C1 + C2 + C3], the 1x1 weight behind the packed k x k weight (packing: ``*.conv2sc.weight``):
        ResnetBlock2D's conv2(h) + conv_shortcut(x) as one launch.
        NHWC conv.  ``up_phases``: w is [4][Cout][4 * Cin] -- the four phase convs of an Upsample2D as ONE launch (conv2d_up2x).  x: [B, H, W, C1] (x2: [B, H, W, C2] virtually concatenated), w: packed [Cout, k*k*(C1+C2)].
        pad = (top, left, bottom, right); default k//2 all round.  shift: [B, ldshift or Cout] per-batch channel shift."""
        if norm is not None and not self.gn_bridge and norm_stats is None:  # (no bridge: the GroupNorm launch, then the plain conv on its output)
            x = self.groupnorm(x, norm.gamma, norm.beta, norm.groups, norm.eps, act=norm.act, x2=None if append is not None else x2, name=norm.name)
            norm, x2 = None, (x2 if append is not None else None)
        B, H, W, C1 = x.shape
        gn_x2 = x2  # the concat partner under the GroupNorm (an appended k_append source is not)
        if append is not None:
            assert x2 is None and stride == 1 and not upsample2x and not up_phases and tuple(append.shape[:3]) == (B, H, W)
            x2 = append
            gn_x2 = None
        C2 = x2.shape[-1] if x2 is not None else 0
        N = w.shape[1] if up_phases else w.shape[0]
        k = ksize
        if pad is None:
            pad = (k // 2,) * 4
        Hin, Win = (2 * H, 2 * W) if upsample2x else (H, W)
        Ho = (Hin + pad[0] + pad[2] - k) // stride + 1
        Wo = (Win + pad[1] + pad[3] - k) // stride + 1
        if out is None:
            out = self.buf(name, (B, Ho, Wo, N))
        else:
            self._wrote(out)
        d = GemmDesc()
        d.a, d.a2, d.w, d.bias, d.shift, d.residual, d.out = (_ptr(x), _ptr(x2), _ptr(w), _ptr(bias), _ptr(shift),
                                                              _ptr(residual), _ptr(out))
        C3 = append2.shape[-1] if append2 is not None else 0
        d.M, d.N, d.K = B * Ho * Wo, N, (k * k * C1 + C2 + C3) if append is not None else k * k * (C1 + C2)
        d.k_append, d.a3, d.C3 = int(append is not None), _ptr(append2), C3
        assert w.shape[-1] == d.K, (tuple(w.shape), d.K)
        d.ldw, d.ldo, d.ldshift = w.stride(-2), out.stride(-2), ldshift
        if up_phases:
            d.batch, d.batch_inner, d.w_bs, d.up_phases = 4, 1, w.stride(0), 1
        if out.dim() == 4 and out.stride(1) != Wo * out.stride(2):
            # a strided view of a larger image (one phase of an upsampling conv writes every other pixel of every other row):
            # two-level row pitch, gn_gemm_desc.out_row_width / ldo_hi
            assert out.stride(0) == Ho * out.stride(1) and residual is None, "strided output views: whole image rows, no residual"
            d.out_row_width, d.ldo_hi = Wo, out.stride(1)
        d.ldr = residual.stride(-2) if residual is not None else 0
        d.conv, d.B, d.H, d.W, d.C1, d.C2 = 1, B, H, W, C1, C2
        d.KH, d.KW, d.stride, d.pad_t, d.pad_l, d.Ho, d.Wo = k, k, stride, pad[0], pad[1], Ho, Wo
        d.upsample2x, d.act, d.out_mode, d.rows_per_batch, d.splitk, d.out_scale = (int(upsample2x), act, OUT_ROWMAJOR,
                                                                                     Ho * Wo, splitk, out_scale)
        d.residual_before_act = int(residual_before_act)
        self._set_sink(d, sink)
        if norm is not None and not self._norm_in(d, x, gn_x2, norm, B * H * W, norm_stats):
            # the GroupNorm as its own launch (one coalesced apply pass where the producers left their statistics), the conv on its output
            d.norm_in.stats = None
            n = self.groupnorm(x, norm.gamma, norm.beta, norm.groups, norm.eps, act=norm.act, x2=gn_x2, name=norm.name)
            if gn_x2 is not None:  # the GroupNorm wrote the concatenation
                d.a, d.a2, d.C1, d.C2 = _ptr(n), None, C1 + C2, 0
            else:
                d.a = _ptr(n)
            x = n
        y = self._set_norm_out(d, norm_out, out, Ho * Wo)
        self._gemm(d, (x, x2, w, bias, shift, residual, out, append2, None if norm is None else norm.gamma, None if norm is None else norm.beta, y))
        return out if y is None else (out, y)

    def conv2d_up2x(self, x: torch.Tensor, w4: torch.Tensor, bias: Optional[torch.Tensor] = None, *, name: Optional[str] = None) -> torch.Tensor:
        """conv3x3(nearest_upsample_2x(x)) as its four phase convs (packing.pack_upsample_phases): phase (dy, dx) is a 2x2 conv over the
        source pixels (top / left padding 1 - dy / 1 - dx) whose results are written straight to pixels (2y + dy, 2x + dx) of the output --
        4 / 9 of the multiply-adds of the fused-upsample 3x3 launch it replaces (diffusers Upsample2D)."""
        B, H, W, _ = x.shape
        N = w4.shape[1]
        out = self.buf(name, (B, 2 * H, 2 * W, N))
        if getattr(self, "up_phases_one_launch", True):  # blockIdx.z = phase: one launch instead of four
            self.conv2d(x, w4, bias, ksize=2, pad=(1, 1, 0, 0), out=out[:, 0::2, 0::2, :], up_phases=True)
            return out
        for dy in (0, 1):
            for dx in (0, 1):
                self.conv2d(x, w4[2 * dy + dx], bias, ksize=2, pad=(1 - dy, 1 - dx, dy, dx), out=out[:, dy::2, dx::2, :])
        return out

    # ------------------------------------------------------------------------------------------------ weight repacking (gn_pack_*)
    def pack_conv_weight(self, w: torch.Tensor) -> torch.Tensor:
        """OIHW f32 / f16 device tensor -> [round_up(O, 8), KH * KW * round_up(I, 8)] f16 (gn_pack_conv_weight)."""
        assert w.dim() == 4 and w.is_cuda and w.dtype in (torch.float32, torch.float16) and not self.record
        w = w.contiguous()
        O, I, KH, KW = w.shape
        out = torch.empty((_round_up(O, 8), KH * KW * _round_up(I, 8)), dtype=F16, device=self.device)
        check(self.lib.gn_pack_conv_weight(self._ctx, _ptr(w), int(w.dtype == torch.float16), _ptr(out), O, I, KH, KW), "gn_pack_conv_weight")
        return out

    def pack_geglu(self, w: torch.Tensor, b: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """GEGLU.proj weight [2H, K] and bias [2H] (hidden rows, then gate rows) -> 32-row blocks hidden | gate, f16 (gn_pack_geglu_rows)."""
        assert w.dim() == 2 and w.is_cuda and w.dtype == b.dtype and w.dtype in (torch.float32, torch.float16) and not self.record
        w, b = w.contiguous(), b.contiguous()
        H, K = w.shape[0] // 2, w.shape[1]
        wp, bp = torch.empty((2 * H, K), dtype=F16, device=self.device), torch.empty((2 * H,), dtype=F16, device=self.device)
        f16 = int(w.dtype == torch.float16)
        check(self.lib.gn_pack_geglu_rows(self._ctx, _ptr(w), f16, _ptr(wp), H, K), "gn_pack_geglu_rows")
        check(self.lib.gn_pack_geglu_rows(self._ctx, _ptr(b), f16, _ptr(bp), H, 1), "gn_pack_geglu_rows")
        return wp, bp

    def pack_tblock_tape(self, kind: int, w_a, b_a, w_ln, c1, c2, w2=None, b2=None, w_p=None, b_p=None) -> torch.Tensor:
        """The weight tape of a fused transformer-block chain (gn_pack_tblock_tape) from packed f16 device tensors (c1 f32) -> uint8 tensor."""
        assert not self.record
        ts = [None if t is None else t.contiguous() for t in (w_a, b_a, w_ln, c1, c2, w2, b2, w_p, b_p)]
        for t, want in zip(ts, (F16, F16, F16, torch.float32, F16, F16, F16, F16, F16)):
            assert t is None or (t.is_cuda and t.dtype == want), (None if t is None else (t.dtype, t.device))
        Cc = ts[0].shape[1]
        nbytes = int(self.lib.gn_tblock_tape_bytes(kind, Cc))
        tape = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        d = TBlockTapeSrc()
        d.kind, d.C = kind, Cc
        d.w_a, d.b_a, d.w_ln, d.c1, d.c2, d.w2, d.b2, d.w_p, d.b_p = (_ptr(t) for t in ts)
        check(self.lib.gn_pack_tblock_tape(self._ctx, C.byref(d), _ptr(tape), nbytes), "gn_pack_tblock_tape")
        self._pack_keep = ts  # (the sources stay alive until the stream has run the kernel: the caller synchronises before dropping the engine)
        return tape

    # ------------------------------------------------------------------------------------------------ attention
    def attention(self, q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, heads: int, *, Nk: Optional[int] = None,
                  causal: bool = False, out: Optional[torch.Tensor] = None, name: Optional[str] = None,
                  lse: Optional[torch.Tensor] = None, v_rowmajor: bool = False) -> torch.Tensor:
        """q: [B, Nq, heads*D] view (last dim contiguous, may be a column slice), k: [B, Nk, heads*D] view,
        vt: [B, heads*D, Nk_pad] (V transposed) -- or, with v_rowmajor (D = 64), V itself as a [B, Nk, heads*D] view.
        Returns o [B, Nq, heads*D].  lse: optional f32 [B, heads, Nq] (training)."""
        B, Nq, Cq = q.shape
        D = Cq // heads
        Nk = k.shape[1] if Nk is None else Nk
        if out is None:
            out = self.buf(name, (B, Nq, Cq))
        else:
            self._wrote(out)
        d = AttnDesc()
        d.q, d.k, d.vt, d.o = _ptr(q), _ptr(k), _ptr(vt), _ptr(out)
        d.q_bs, d.k_bs, d.vt_bs, d.o_bs = q.stride(0), k.stride(0), vt.stride(0), out.stride(0)
        d.q_rs, d.k_rs, d.vt_rs, d.o_rs = q.stride(1), k.stride(1), vt.stride(1), out.stride(1)
        d.B, d.heads, d.Nq, d.Nk, d.D, d.causal, d.scale = B, heads, Nq, Nk, D, int(causal), float(D) ** -0.5
        d.lse = _ptr(lse)
        d.v_rowmajor = int(v_rowmajor)
        if self.record:
            check(self.lib.gn_program_add_attention(self._prog, C.byref(d)), "gn_program_add_attention")
            self._keepalive(q, k, vt, out)
            fl = 4.0 * B * heads * Nq * Nk * D * (0.5 if causal else 1.0)
            self.meta.append(dict(kind="attention", flops=fl, bytes=2.0 * B * Cq * (2 * Nq + 2 * Nk), shape=(B, heads, Nq, Nk, D)))
        else:
            check(self.lib.gn_attention_fwd(self._ctx, C.byref(d)), "gn_attention_fwd")
        return out

    def attention_fp8(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, *, out: Optional[torch.Tensor] = None,
                      name: Optional[str] = None, lse: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Self-/cross-attention on the fp8 MFMA (D = 64; opt-in, the fp8 training forward -- csrc/attention_fp8.hip): q [B, Nq, heads*64],
        k / v [B, Nk, heads*64] f16 views (column slices of a q | k | v projection are fine).  One launch makes the e4m3 operands
        (scale folded into q8, V transposed into the MFMA key order), one runs the attention.  Eager only."""
        if self.record:
            raise GenimaHipError("attention_fp8 is an eager op (it serves the training forward, which is not recorded)")
        B, Nq, Cq = q.shape
        Nk = k.shape[1]
        if Cq != heads * 64:
            raise GenimaHipError(f"attention_fp8: head dim {Cq // heads} unsupported (64)")
        if Nq != Nk:
            raise GenimaHipError("attention_fp8: one row count for q and k / v (self-attention)")
        Np = (Nk + 63) // 64 * 64
        q8 = self.buf((name or "attn8") + ".q8", (B, Nq, Cq), dtype=torch.uint8)
        k8 = self.buf((name or "attn8") + ".k8", (B, Nk, Cq), dtype=torch.uint8)
        v8t = self.buf((name or "attn8") + ".v8t", (B, Cq, Np), dtype=torch.uint8)
        check(self.lib.gn_attention_fp8_quantize(self._ctx, _ptr(q), _ptr(k), _ptr(v), q.stride(1), k.stride(1), v.stride(1), q.stride(0),
                                                  k.stride(0), v.stride(0), B, Nq, heads, 64.0 ** -0.5, _ptr(q8), _ptr(k8), _ptr(v8t), Np),
              "gn_attention_fp8_quantize")
        if out is None:
            out = self.buf(name, (B, Nq, Cq))
        else:
            self._wrote(out)
        d = AttnDesc()
        d.q, d.k, d.vt, d.o = _ptr(q8), _ptr(k8), _ptr(v8t), _ptr(out)
        d.q_bs, d.k_bs, d.vt_bs, d.o_bs = q8.stride(0), k8.stride(0), v8t.stride(0), out.stride(0)
        d.q_rs, d.k_rs, d.vt_rs, d.o_rs = q8.stride(1), k8.stride(1), v8t.stride(1), out.stride(1)
        d.B, d.heads, d.Nq, d.Nk, d.D, d.causal, d.scale = B, heads, Nq, Nk, 64, 0, 1.0
        d.lse = _ptr(lse)
        d.v_rowmajor = 0
        check(self.lib.gn_attention_fp8_fwd(self._ctx, C.byref(d)), "gn_attention_fp8_fwd")
        return out

    # ------------------------------------------------------------------------------------------------ norms
    def groupnorm(self, x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float, *,
                  act: int = ACT_NONE, x2: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                  name: Optional[str] = None, stats_in: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x: [B, H, W, C1] or [B, HW, C1] (x2 optional concat source).  Returns act(GN(cat)) [.., C1+C2].
        stats_in: an explicit, filled statistics block of the GroupNorm bridge (eager calls / tests)."""
        B, C1 = x.shape[0], x.shape[-1]
        HW = x.numel() // (B * C1)
        C2 = x2.shape[-1] if x2 is not None else 0
        if out is None:
            out = self.buf(name, tuple(x.shape[:-1]) + (C1 + C2,))
        else:
            self._wrote(out)
        d = GroupNormDesc()
        d.x, d.x2, d.gamma, d.beta, d.y = _ptr(x), _ptr(x2), _ptr(gamma), _ptr(beta), _ptr(out)
        d.B, d.HW, d.C1, d.C2, d.groups, d.act, d.eps = B, HW, C1, C2, groups, act, eps
        ws = self._workspace(int(self.lib.gn_groupnorm_workspace_bytes(C.byref(d))))
        d.workspace = ws.data_ptr()
        if stats_in is None and x2 is None and self._norm_out(x, gamma, beta, groups, eps, act, out, B, HW, C1):
            return out  # the launch that wrote x normalises it in its split-K reduce: no GroupNorm op
        st = stats_in
        if st is None and getattr(self, "gn_apply_from_stats", True) and gamma.data_ptr() % 16 == 0 and beta.data_ptr() % 16 == 0:
            st = self.bridge_stats(x, x2, groups)
        if st is not None:  # the producers of x (| x2) leave the statistics: ONE coalesced apply launch
            d.stats_in, d.stats_replicas = st.data_ptr(), int(st.shape[0])
        if self.record:
            check(self.lib.gn_program_add_groupnorm(self._prog, C.byref(d)), "gn_program_add_groupnorm")
            self._keepalive(x, x2, gamma, beta, out, ws)
            self.meta.append(dict(kind="groupnorm", flops=0.0, bytes=2.0 * 2 * B * HW * (C1 + C2), shape=(B, HW, C1 + C2) + (("st",) if st is not None else ())))
        else:
            check(self.lib.gn_groupnorm_fwd(self._ctx, C.byref(d)), "gn_groupnorm_fwd")
        return out

    def _norm_out(self, x, gamma, beta, groups, eps, act, out, B, HW, Cc) -> bool:
        """Recorded programs: move this GroupNorm (+ activation) into the split-K reduce of the gn_gemm that wrote x (gn_gemm_desc.norm_out).
        -> done?  (No: x was not written by a K-split launch of this program, is also being normalised elsewhere, or the slab does not fit.)"""
        if not (self.record and self.gn_reduce_fuse) or (B * groups < self.gn_reduce_fuse_min_slabs and HW > self.gn_reduce_fuse_small_hw):
            return False
        w = self._writer.get(x.data_ptr())
        if w is None or not w.gemm or w.sunk or w.rdiv != 1 or w.numel != x.numel() or not x.is_contiguous() or not out.is_contiguous():
            return False
        n = NormOut()
        n.y, n.gamma, n.beta, n.eps, n.groups, n.act, n.rows_per_sample = _ptr(out), _ptr(gamma), _ptr(beta), float(eps), int(groups), int(act), int(HW)
        d = GemmDesc()
        if self.lib.gn_program_get_gemm(self._prog, w.op, C.byref(d)) != 0 or d.norm_out.y:
            return False
        d.norm_out = n
        if not self.lib.gn_gemm_norm_out_supported(C.byref(d)):
            return False
        check(self.lib.gn_program_set_norm_out(self._prog, w.op, C.byref(n)), "gn_program_set_norm_out")
        w.sunk = True
        self._keepalive(gamma, beta, out)
        m = self.meta[w.op]
        m["norm_out"] = (B, HW, Cc)
        m["bytes"] = m.get("bytes", 0.0) + 2.0 * B * HW * Cc  # the normalised tensor's write joins the launch
        return True

    def groupnorm_stats(self, x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float, *,
                        name: Optional[str] = None) -> torch.Tensor:
        """Statistics-only GroupNorm (gn_groupnorm_fwd with y == NULL): -> f32 [B, C, 2] per-(sample, channel) (scale, shift) such that
        GN(x)[b, .., c] = x * scale + shift; conv2d_gn applies them (and the SiLU) to its input patch in LDS.  One read of x, no write."""
        B, Cc = x.shape[0], x.shape[-1]
        HW = x.numel() // (B * Cc)
        scsh = self.buf(name, (B, Cc, 2), dtype=torch.float32)
        d = GroupNormDesc()
        d.x, d.gamma, d.beta, d.y, d.save_scsh = _ptr(x), _ptr(gamma), _ptr(beta), None, _ptr(scsh)
        d.B, d.HW, d.C1, d.C2, d.groups, d.act, d.eps = B, HW, Cc, 0, groups, ACT_NONE, eps
        ws = self._workspace(int(self.lib.gn_groupnorm_workspace_bytes(C.byref(d))))
        d.workspace = ws.data_ptr()
        if self.record:
            check(self.lib.gn_program_add_groupnorm(self._prog, C.byref(d)), "gn_program_add_groupnorm")
            self._keepalive(x, gamma, beta, scsh, ws)
            self.meta.append(dict(kind="groupnorm", flops=0.0, bytes=2.0 * B * HW * Cc, shape=(B, HW, Cc, 0)))
        else:
            check(self.lib.gn_groupnorm_fwd(self._ctx, C.byref(d)), "gn_groupnorm_fwd")
        return scsh

    def conv2d_gn_supported(self, x: torch.Tensor, cout: int) -> bool:
        B, H, W, Cin = x.shape
        return bool(self.lib.gn_conv3x3_gn_supported(B, H, W, Cin, cout))

    def conv2d_gn(self, x: torch.Tensor, scsh: Optional[torch.Tensor], w: torch.Tensor, bias: Optional[torch.Tensor] = None, *,
                  act: int = ACT_SILU, residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                  name: Optional[str] = None) -> torch.Tensor:
        """conv3x3(act(x * scale + shift)) + bias (+ residual) with the GroupNorm-apply and the SiLU done on the conv's LDS patch
        (gn_conv3x3_gn, csrc/conv_gn.hip).  x: RAW NHWC [B, H, W, Cin]; scsh from groupnorm_stats (None: plain conv); w packed [Cout, 9 Cin]."""
        B, H, W, Cin = x.shape
        N = w.shape[0]
        assert w.shape[1] == 9 * Cin, (tuple(w.shape), Cin)
        if out is None:
            out = self.buf(name, (B, H, W, N))
        else:
            self._wrote(out)
        d = ConvGnDesc()
        d.x, d.scsh, d.w, d.bias, d.residual, d.out = _ptr(x), _ptr(scsh), _ptr(w), _ptr(bias), _ptr(residual), _ptr(out)
        d.ldr, d.ldo = (residual.stride(-2) if residual is not None else 0), out.stride(-2)
        d.B, d.H, d.W, d.Cin, d.Cout, d.act = B, H, W, Cin, N, act
        if self.record:
            check(self.lib.gn_program_add_conv3x3_gn(self._prog, C.byref(d)), "gn_program_add_conv3x3_gn")
            self._keepalive(x, scsh, w, bias, residual, out)
            M, K = B * H * W, 9 * Cin
            self.meta.append(dict(kind="conv3x3", flops=2.0 * M * N * K, bytes=2.0 * (M * Cin + N * K + M * N), shape=(M, N, K), ref_flops=2.0 * M * N * K))
        else:
            check(self.lib.gn_conv3x3_gn(self._ctx, C.byref(d)), "gn_conv3x3_gn")
        return out

    def layernorm(self, x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5, *,
                  out: Optional[torch.Tensor] = None, name: Optional[str] = None) -> torch.Tensor:
        Cc = x.shape[-1]
        M = x.numel() // Cc
        if out is None:
            out = self.buf(name, x.shape)
        else:
            self._wrote(out)
        args = (_ptr(x), _ptr(gamma), _ptr(beta), _ptr(out), M, Cc, eps)
        if self.record:
            check(self.lib.gn_program_add_layernorm(self._prog, *args), "gn_program_add_layernorm")
            self._keepalive(x, gamma, beta, out)
            self.meta.append(dict(kind="layernorm", flops=0.0, bytes=2.0 * 2 * M * Cc, shape=(M, Cc)))
        else:
            check(self.lib.gn_layernorm_fwd(self._ctx, *args), "gn_layernorm_fwd")
        return out

    # ------------------------------------------------------------------------------------------------ small ops
    def _small(self, fn_name: str, keep, *args, nbytes: Optional[float] = None):
        """``nbytes``: the bytes the op actually touches when that is not the size of its operands (gathers read a few rows of a table)."""
        if self.record:
            check(getattr(self.lib, "gn_program_add_" + fn_name)(self._prog, *args), "gn_program_add_" + fn_name)
            self._keepalive(*keep)
            if nbytes is None:
                nbytes = float(sum(t.numel() * t.element_size() for t in keep if t is not None))
            self.meta.append(dict(kind=fn_name, flops=0.0, bytes=float(nbytes), shape=()))
        else:
            check(getattr(self.lib, "gn_" + fn_name)(self._ctx, *args), "gn_" + fn_name)

    def timestep_embedding(self, t: torch.Tensor, dim: int, flip_sin_to_cos=True, freq_shift=0.0, *, name=None):
        """t: f32 [B] device tensor -> f16 [B, dim]."""
        out = self.buf(name, (t.numel(), dim))
        self._small("timestep_embedding", (t, out), _ptr(t), _ptr(out), t.numel(), dim, int(flip_sin_to_cos), float(freq_shift))
        return out

    def scale_pad(self, x: torch.Tensor, scale: float, cpad: int, *, out=None, name=None):
        """x: [..., C] -> [..., cpad] = x*scale zero-padded on channels."""
        Cc = x.shape[-1]
        if out is None:
            out = self.buf(name, tuple(x.shape[:-1]) + (cpad,))
        else:
            self._wrote(out)
        self._small("scale_pad", (x, out), _ptr(x), _ptr(out), x.numel() // Cc, Cc, cpad, float(scale))
        return out

    def scale_cat_pad(self, x: torch.Tensor, c1: int, x2: torch.Tensor, c2: int, cpad: int, scale: float = 1.0, scale2: float = 1.0, *,
                      out=None, name=None):
        """out[..., :c1] = x[..., :c1] * scale | out[..., c1:c1+c2] = x2[..., :c2] * scale2 | zeros up to cpad (the pixel rows of x / x2
        may be wider than c1 / c2: padded latents, VAE moments)."""
        pixels = x.numel() // x.shape[-1]
        if out is None:
            out = self.buf(name, tuple(x.shape[:-1]) + (cpad,))
        else:
            self._wrote(out)
        self._small("scale_cat_pad", (x, x2, out), _ptr(x), _ptr(x2), _ptr(out), pixels, c1, x.stride(-2), c2, x2.stride(-2), cpad,
                    float(scale), float(scale2))
        return out

    def euler_step(self, x: torch.Tensor, eps: torch.Tensor, sigma: float, sigma_next: float):
        """In place: x[..., C] <- x + eps[..., :C] * (sigma_next - sigma)."""
        Cc = x.shape[-1]
        self._wrote(x)
        self._small("euler_step", (x, eps), _ptr(x), _ptr(eps), x.numel() // Cc, Cc, eps.stride(-2), float(sigma), float(sigma_next))
        return x

    def add_noise(self, x0, noise, sqrt_ac, sqrt_1mac, *, out=None, name=None):
        """out[b] = sqrt_ac[b] * x0[b] + sqrt_1mac[b] * noise[b] (DDPM add_noise; also the ancestral sampler's x + sigma_up * noise);
        out may alias x0."""
        if out is None:
            out = self.buf(name, x0.shape)
        else:
            self._wrote(out)
        B = x0.shape[0]
        self._small("add_noise", (x0, noise, sqrt_ac, sqrt_1mac, out), _ptr(x0), _ptr(noise), _ptr(sqrt_ac), _ptr(sqrt_1mac), _ptr(out), B, x0.numel() // B)
        return out

    def image_u8_to_f16(self, img: torch.Tensor, cpad: int = 8, mul: float = 1.0, add: float = 0.0, *, out=None, name=None):
        """uint8 [B, H, W, 3] -> f16 [B, H, W, cpad] = v/255*mul + add (channels >= 3 zero)."""
        if out is None:
            out = self.buf(name, tuple(img.shape[:-1]) + (cpad,))
        else:
            self._wrote(out)
        self._small("image_u8_to_f16", (img, out), _ptr(img), _ptr(out), img.numel() // 3, cpad, float(mul), float(add))
        return out

    def image_f16_to_u8(self, x: torch.Tensor, *, out=None, name=None):
        """f16 [B, H, W, ld>=3] -> uint8 [B, H, W, 3] (VaeImageProcessor.postprocess numerics)."""
        if out is None:
            out = self.buf(name, tuple(x.shape[:-1]) + (3,), dtype=torch.uint8)
        else:
            self._wrote(out)
        self._small("image_f16_to_u8", (x, out), _ptr(x), _ptr(out), x.numel() // x.shape[-1], x.stride(-2))
        return out

    def add(self, a: torch.Tensor, b: torch.Tensor, *, out=None, name=None):
        if out is None:
            out = self.buf(name, a.shape)
        else:
            self._wrote(out)
        self._small("add", (a, b, out), _ptr(a), _ptr(b), _ptr(out), a.numel())
        return out

    def add_multi(self, pairs, *, name=None, sinks=None):
        """[(a, b), ...] (<= 16) -> [a + b, ...] as ONE launch (gn_add_multi): independent small adds that would each pay a launch boundary.
        sinks (eager calls / tests): per pair None or (stats, cpg, coff, rows_per_sample) -- gn_add_multi_stats."""
        n = len(pairs)
        if sinks is not None:
            assert not self.record
            outs = [torch.empty_like(a) for a, _ in pairs]
            A = (C.c_void_p * n)(*[_ptr(a) for a, _ in pairs])
            Bp = (C.c_void_p * n)(*[_ptr(b) for _, b in pairs])
            O = (C.c_void_p * n)(*[_ptr(o) for o in outs])
            N = (C.c_int64 * n)(*[a.numel() for a, _ in pairs])
            Cs = (C.c_int32 * n)(*[a.shape[-1] for a, _ in pairs])
            S = (StatsSink * n)()
            for i, sk in enumerate(sinks):
                if sk is not None:
                    st, cpg, coff, rps = sk
                    S[i].stats, S[i].cpg, S[i].coff, S[i].groups, S[i].rows_per_sample = st.data_ptr(), int(cpg), int(coff), int(st.shape[2]), int(rps)
                    S[i].samples, S[i].replicas = int(st.shape[1]), int(st.shape[0])
            check(self.lib.gn_add_multi_stats(self._ctx, A, Bp, O, N, Cs, S, n), "gn_add_multi_stats")
            return outs
        outs = [self.buf(None if name is None else f"{name}{i}", a.shape) for i, (a, _) in enumerate(pairs)]
        A = (C.c_void_p * n)(*[_ptr(a) for a, _ in pairs])
        Bp = (C.c_void_p * n)(*[_ptr(b) for _, b in pairs])
        O = (C.c_void_p * n)(*[_ptr(o) for o in outs])
        N = (C.c_int64 * n)(*[a.numel() for a, _ in pairs])
        keep = tuple(t for pr in pairs for t in pr) + tuple(outs)
        if self.gn_bridge:
            for i, o in enumerate(outs):
                self._writer[o.data_ptr()] = _Writer(self.num_ops, i, o.numel())
        self._small("add_multi", keep, A, Bp, O, N, n)
        return outs

    def act(self, x: torch.Tensor, act: int, *, out=None, name=None):
        if out is None:
            out = self.buf(name, x.shape)
        else:
            self._wrote(out)
        self._small("act", (x, out), _ptr(x), _ptr(out), x.numel(), act)
        return out

    def film(self, x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, rows_per_film: int, act: int = ACT_NONE, *, out=None, name=None):
        """out = act((1 + gamma[b]) * x + beta[b]) with b = row // rows_per_film; x [..., C] f16, gamma / beta [B, C] views (row stride =
        the FiLM feature buffer's width) of one tensor."""
        Cc = x.shape[-1]
        rows = x.numel() // Cc
        assert gamma.stride(0) == beta.stride(0) and gamma.stride(-1) == 1 and rows % rows_per_film == 0
        if out is None:
            out = self.buf(name, x.shape)
        else:
            self._wrote(out)
        self._small("film", (x, gamma, beta, out), _ptr(x), _ptr(out), _ptr(gamma), _ptr(beta), gamma.stride(0), rows_per_film, rows, Cc, act)
        return out

    def embedding(self, ids: torch.Tensor, tok: torch.Tensor, pos: torch.Tensor, *, name=None):
        """ids int32 [B, L] -> tok[ids] + pos[:L]  f16 [B, L, D]."""
        B, L = ids.shape
        D = tok.shape[1]
        out = self.buf(name, (B, L, D))
        # B * L gathered table rows + L position rows read, B * L rows written (not the whole 100 MB token table)
        self._small("embedding", (ids, tok, pos, out), _ptr(ids), _ptr(tok), _ptr(pos), _ptr(out), B, L, D,
                    nbytes=4.0 * B * L + 2.0 * D * (2 * B * L + L))
        return out

    def softmax_rows(self, x: torch.Tensor, scale: float = 1.0):
        """In-place softmax(scale * x) over the last dim of a 2-D-viewable f16 tensor."""
        cols = x.shape[-1]
        self._small("softmax_rows", (x,), _ptr(x), x.numel() // cols, cols, x.stride(-2), float(scale))
        return x

    def image_normalize_u8(self, img: torch.Tensor, mean, std, cpad: int = 8, *, name=None):
        """uint8 [..., 3] -> f16 [..., cpad] = (v/255 - mean_c) / std_c."""
        out = self.buf(name, tuple(img.shape[:-1]) + (cpad,))
        m = [1.0 / (255.0 * s_) for s_ in std]
        a = [-mu / s_ for mu, s_ in zip(mean, std)]
        self._small("image_normalize_u8", (img, out), _ptr(img), _ptr(out), img.numel() // 3, cpad, *m, *a)
        return out

    def gather_rows(self, x: torch.Tensor, idx: torch.Tensor, *, name=None):
        """x [B, L, D], idx int32 [B] -> [B, D]."""
        B, L, D = x.shape
        out = self.buf(name, (B, D))
        self._small("gather_rows", (x, idx, out), _ptr(x), _ptr(idx), _ptr(out), B, L, D, nbytes=4.0 * B + 4.0 * B * D)
        return out

    def argmax_rows(self, x: torch.Tensor, *, name=None):
        """int32 [rows, cols] -> int32 [rows]: first index of each row's maximum."""
        rows, cols = x.shape
        out = self.buf(name, (rows,), dtype=torch.int32)
        self._small("argmax_rows_i32", (x, out), _ptr(x), _ptr(out), rows, cols)
        return out

    def copy4d(self, src: torch.Tensor, dst: torch.Tensor, sizes, in_strides, out_strides, L: int):
        """dst[i0*os0 + i1*os1 + i2*os2 + i3*os3 + :L] = src[i0*is0 + ... + :L] over the 4-D index space ``sizes`` (f16 elements)."""
        arr = (C.c_int64 * 4)
        s, i, o = arr(*sizes), arr(*in_strides), arr(*out_strides)
        self._wrote(dst)
        self._small("copy4d", (src, dst), _ptr(src), _ptr(dst), s, i, o, L)
        return dst

    def maxpool3x3s2(self, x: torch.Tensor, *, name=None):
        B, H, W, Cc = x.shape
        out = self.buf(name, (B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, Cc))
        self._small("maxpool3x3s2", (x, out), _ptr(x), _ptr(out), B, H, W, Cc)
        return out
