"""One exact GP whose covariance matrix is too large for a single GPU: column-panel block-cyclic
Cholesky over the ranks of a ``torch.distributed`` group (backend "nccl" = RCCL over xGMI; "gloo" in
the CPU tests).  BASELINE config 4 (N = 262 144 on 4 x MI355X); the reference's
one-GP-over-several-GPUs switch (``n_devices``, ``src/batt_models/cell_gp.py:37-47``).

Distribution.  The matrix is cut into column panels of width ``nb``; panel ``j`` lives on rank
``j % world`` and keeps ONLY the rows from its own diagonal down plus the 64-row augmented block whose row 0
carries ``y^T`` - a rank holds ~``4 N^2 / world`` bytes (69 GB per rank at N = 262 144, world = 4).
Right-looking factorisation, one exchange per panel step:

    for k in panels:                                   (panel k is already on every rank)
        owner(k+1): C_{k+1} -= P_k ...; factor panel k+1 locally; pack it                     (look-ahead)
        all ranks:  start the broadcast of packed panel k+1 [(Npad + 64 - (k+1) nb) x nb + flag]  <- RCCL, async
        every rank: C_j -= P_k[rows >= j nb] P_k[rows of j]^T  for each of ITS panels j > k   (MFMA)

so each rank receives ~4 N^2 (w-1)/w bytes in total and runs 1/w of the N^3/3 flops.

The whole factorisation is ENQUEUED: no host synchronisation inside the panel loop.  The engine's HIP stream is
handed to torch as an ``ExternalStream`` and made current, so torch's buffer operations and its RCCL collectives are
ordered with the engine's kernels by the streams themselves: the broadcast of panel k+1 waits (on the device) for the
kernels that pack it and runs on RCCL's stream next to the rank-``nb`` updates of step k; the kernels of step k+1 wait
(on the device) for that broadcast.  A failed pivot sets a device flag that turns the owner's later kernels into no-ops
and travels behind the packed panel to poison the other ranks' pipelines; the host looks at the flags ONCE per
factorisation attempt (an all-reduce MAX) and walks the jitter ladder like the single-GPU engine.

The forward solve rides along in the augmented row (``z^T`` comes out of the factorisation), ``log det`` and the ``z``
segments are combined with small all-reduces.  Prediction pushes the query block through the factor right-looking:
the owner of panel k finishes ``E_k`` (reduce of the pending contributions), updates its own accumulator for all later
columns, and adds its share of ``mean = V^T z`` and ``var = k_** - rowsumsq(V^T)``.

The numerical work is done by a *backend*: ``DeviceBackend`` drives the HIP kernels of libbattgp.so on torch-owned
device buffers (torch = container + communicator only); the tests supply a numpy backend to check the distributed
algorithm on CPU/gloo.
"""

from __future__ import annotations

import contextlib
import ctypes as C
import math
import time

import numpy as np

AUG = 64  # rows of the augmented block (BGP_AUG)


def round_up(v: int, q: int) -> int:
    return (v + q - 1) // q * q


class PanelLayout:
    def __init__(self, n: int, nb: int, world: int, ride: int = 0):
        if nb % 64 != 0 or nb < 64:
            raise ValueError("nb must be a positive multiple of 64")
        if ride % 64 != 0 or ride < 0:
            raise ValueError("ride must be a non-negative multiple of 64")
        self.n, self.nb, self.world, self.ride = n, nb, world, ride
        self.npad = round_up(n, 64)
        self.npanels = -(-self.npad // nb)
        # rows of panel 0: the padded matrix + the augmented block (+ the cross-covariance rows of a fit_predict, which
        # ride through the factorisation below it and come out as K_*X L^-T)
        self.nrows = self.npad + AUG + ride

    def owner(self, j: int) -> int:
        return j % self.world

    def local_index(self, j: int) -> int:
        return j // self.world

    def col0(self, j: int) -> int:
        return j * self.nb

    def width(self, j: int) -> int:
        return min(self.nb, self.npad - j * self.nb)

    def local_panels(self, rank: int) -> list[int]:
        return [j for j in range(self.npanels) if j % self.world == rank]

    def rows_from(self, j: int) -> int:
        """rows of panel j from its diagonal down, augmented block (and riding rows) included"""
        return self.nrows - self.col0(j)

    def sigma_rows(self, j: int) -> int:
        """rows of panel j inside the matrix = offset of its augmented block; the riding rows start AUG further down"""
        return self.npad - self.col0(j)

    def ld(self, j: int) -> int:
        """leading dimension of stored panel j: its own height (no rows above its diagonal block are kept), bumped off
        large power-of-two strides (all columns of a tile in one HBM channel)"""
        r = self.rows_from(j)
        return r + 64 if (r >= 2048 and r % 512 == 0) else r

    def offsets(self, rank: int) -> tuple[dict, int]:
        """element offset of every local panel in the rank's store, and the store's size"""
        off, total = {}, 0
        for j in self.local_panels(rank):
            off[j] = total
            total += self.ld(j) * self.width(j)
        return off, total


class DeviceBackend:
    """HIP kernels through the C-ABI on torch-owned device memory.  Every torch operation on those buffers and every
    collective is issued with the ENGINE's stream current (``torch.cuda.ExternalStream`` around ``bgp_get_stream``), so
    kernels, copies and RCCL calls are ordered on the device."""

    def __init__(self, engine, device):
        import torch

        from . import _lib

        self.torch = torch
        self.eng = engine
        self.lib = _lib.load()
        self.h = engine._h
        self.device = device
        ptr = self.lib.bgp_get_stream(self.h, 0)
        self.stream = torch.cuda.ExternalStream(int(ptr), device=device)

    def on_stream(self):
        return self.torch.cuda.stream(self.stream)

    def _chk(self, rc, what):
        self.eng._check(rc, what)

    def zeros(self, n):
        with self.on_stream():
            return self.torch.zeros(int(n), dtype=self.torch.float64, device=self.device)

    def empty(self, n):
        with self.on_stream():
            return self.torch.empty(int(n), dtype=self.torch.float64, device=self.device)

    def upload(self, a):
        with self.on_stream():
            return self.torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64)).to(self.device)

    def to_host(self, t):
        with self.on_stream():
            return t.detach().cpu().numpy()

    def sync(self):
        """drain the engine's streams"""
        self._chk(self.lib.bgp_sync(self.h), "bgp_sync")

    @staticmethod
    def _p(t, off=0):
        return C.c_void_p(t.data_ptr() + 8 * int(off))

    # ---- factorisation ------------------------------------------------------------------------------
    def fill_panel(self, store, off, ld, x_dev, n, d, col0, ncols, nsig, y_dev, extra_diag):
        """panel at store[off:], leading dimension ld: the nsig Sigma rows [col0, npad) of columns [col0, col0 + ncols),
        then the augmented block"""
        self._chk(
            self.lib.bgp_fill_block_dev(self.h, self._p(x_dev), n, d, col0, col0, nsig, ncols, self._p(store, off), ld, float(extra_diag)),
            "bgp_fill_block_dev",
        )
        self._chk(
            self.lib.bgp_aug_rows_dev(self.h, self._p(y_dev), n, col0, ncols, self._p(store, off + nsig), ld),
            "bgp_aug_rows_dev",
        )

    def ride_fill(self, store, off, ld, xq_dev, m, nrows, x_dev, n, d, col0, ncols):
        """the riding block [nrows, ncols] at store[off:]: k(xq_i, x_j) for the panel's columns, zero in the padding"""
        self._chk(
            self.lib.bgp_cross_block_dev(self.h, self._p(xq_dev), m, nrows, self._p(x_dev), n, d, col0, ncols, self._p(store, off), ld),
            "bgp_cross_block_dev",
        )

    def flag_reset(self):
        self._chk(self.lib.bgp_flag_reset_dev(self.h), "bgp_flag_reset_dev")

    def factor_pack(self, store, off, ld, rows, nbk, inv, pbuf, col0):
        """factor the panel (asynchronously) and leave its packed copy + the failure flag slot in pbuf"""
        self._chk(
            self.lib.bgp_factor_pack_panel_async_dev(
                self.h, self._p(store, off), ld, rows, nbk, self._p(inv), self._p(pbuf), col0, self._p(pbuf, nbk * rows)
            ),
            "bgp_factor_pack_panel_async_dev",
        )

    def flag_merge(self, pbuf, idx):
        self._chk(self.lib.bgp_flag_merge_dev(self.h, self._p(pbuf, idx)), "bgp_flag_merge_dev")

    def flag_read(self) -> int:
        out = C.c_int(0)
        self._chk(self.lib.bgp_flag_read(self.h, C.byref(out)), "bgp_flag_read")
        return int(out.value)

    def update_panels(self, store, items, pbuf, ldp, nbk, flag_idx):
        """items: (c_off, ldc, rows_j, nbj, p_off) per local panel - all updates of one step in one C call"""
        if not items:
            return
        desc = np.ascontiguousarray([[c_off, rows_j, nbj, p_off, ldc] for c_off, ldc, rows_j, nbj, p_off in items], dtype=np.int64)
        self._chk(
            self.lib.bgp_update_panels_dev(
                self.h, self._p(store), desc.ctypes.data_as(C.POINTER(C.c_int64)), len(items), self._p(pbuf), ldp, nbk,
                self._p(pbuf, flag_idx) if flag_idx is not None else None,
            ),
            "bgp_update_panels_dev",
        )

    def diag_logsum(self, store, off, ld, nbk) -> float:
        out = C.c_double(0.0)
        self._chk(self.lib.bgp_diag_logsum_dev(self.h, self._p(store, off), ld, nbk, C.byref(out)), "bgp_diag_logsum_dev")
        return float(out.value)

    def aug_row(self, store, off, ld, nsig, nbk):
        """z segment = row 0 of the augmented block under the panel's nsig matrix rows (strided view, copied)"""
        with self.on_stream():
            return store.as_strided((nbk,), (ld,), off + nsig).clone()

    # ---- collectives (issued with the engine's stream current) ------------------------------------------
    def bcast_start(self, dist, t, src):
        with self.on_stream():
            return dist.broadcast(t, src=src, async_op=True)

    def bcast_wait(self, work):
        with self.on_stream():
            work.wait()  # device-side wait of the engine's stream; the host does not block (RCCL)

    def allreduce(self, dist, t, op=None):
        with self.on_stream():
            dist.all_reduce(t) if op is None else dist.all_reduce(t, op=op)

    def reduce_start(self, dist, t, dst):
        with self.on_stream():
            return dist.reduce(t, dst=dst, async_op=True)

    def allgather_start(self, dist, recv, send):
        """recv [world * len(send)] <- every rank's send, in rank order"""
        with self.on_stream():
            return dist.all_gather_into_tensor(recv, send, async_op=True)

    def unshuffle(self, recv, wt, world, nslots, nb, ncols):
        """The gathered chunks - rank r's [nslots * nb, ncols] column-major block holds its panels' pieces in slot order -
        into the natural row order of wt [world * nslots * nb, ncols]: rows of slot i of rank r go to panel i * world + r.
        One strided device copy (a buffer operation like the broadcast buffers' slicing; no arithmetic)."""
        with self.on_stream():
            cnt = world * nslots * nb * ncols
            src = recv[:cnt].view(world, ncols, nslots, nb).permute(1, 2, 0, 3)
            wt[:cnt].view(ncols, nslots, world, nb).copy_(src)

    # ---- gradient: Sigma^-1 in place over the distributed factor ------------------------------------------
    def gemm(self, mode, c, coff, ldc, a, aoff, lda, b, boff, ldb, m, n, k, lower=0, btri=0):
        """C (op)= A B^T on the MFMA kernel: mode 0 ``-=``, 1 ``=``, 2 ``-=`` by atomics (deep k)"""
        self._chk(
            self.lib.bgp_gemm_nt_async_dev(self.h, mode, self._p(c, coff), ldc, self._p(a, aoff), lda, self._p(b, boff), ldb, m, n, k, lower, btri),
            "bgp_gemm_nt_async_dev",
        )

    def block_copy(self, src, soff, lds, rows, cols, dst, doff, ldd, trans=0, scale=1.0, tri=0):
        self._chk(
            self.lib.bgp_block_copy_dev(self.h, self._p(src, soff), lds, rows, cols, self._p(dst, doff), ldd, trans, float(scale), tri),
            "bgp_block_copy_dev",
        )

    def panel_inverse(self, store, off, ld, nbk, inv, out):
        self._chk(self.lib.bgp_panel_inverse_dev(self.h, self._p(store, off), ld, nbk, self._p(inv), self._p(out)), "bgp_panel_inverse_dev")

    def gemv_t(self, a, aoff, ld, rows, ncols, x, xoff, out, ooff):
        self._chk(self.lib.bgp_gemv_t_dev(self.h, self._p(a, aoff), ld, rows, ncols, self._p(x, xoff), self._p(out, ooff)), "bgp_gemv_t_dev")

    def grad_acc(self):
        return self.zeros(int(self.lib.bgp_grad_nacc()))

    def grad_reduce(self, x_dev, n, d, r0, nrows, ncols, store, off, ld, alpha, acc):
        self._chk(
            self.lib.bgp_grad_reduce_block_dev(self.h, self._p(x_dev), n, d, r0, nrows, ncols, self._p(store, off), ld, self._p(alpha), self._p(acc), 1),
            "bgp_grad_reduce_block_dev",
        )

    def grad_finish(self, acc, d):
        a = np.ascontiguousarray(self.to_host(acc), dtype=np.float64)
        nhyp = int(self.eng.hyp.size)
        g = np.zeros(nhyp)
        self._chk(self.lib.bgp_grad_finish(self.h, a.ctypes.data_as(C.POINTER(C.c_double)), d, g.ctypes.data_as(C.POINTER(C.c_double)), nhyp), "bgp_grad_finish")
        return g

    # ---- prediction ---------------------------------------------------------------------------
    def cross_fill(self, xq_dev, m, mpad, x_dev, n, d, npad, out, lde):
        # E[m, i] = k(xq_m, x_i): rows = queries (offset 0), columns = training points
        p = self.lib.bgp_fill_dev
        self._chk(p(self.h, self._p(xq_dev), m, self._p(x_dev), n, d, self._p(out), lde, 0, 0.0), "bgp_fill_dev")

    def solve_panel(self, e, eoff, lde, me, store, off, ld, nbk, inv):
        self._chk(
            self.lib.bgp_solve_panel_dev(self.h, self._p(e, eoff), lde, me, self._p(store, off), ld, nbk, self._p(inv)),
            "bgp_solve_panel_dev",
        )

    def update_rows(self, w, woff, lde, me, ek, ldek, store, off, ld, nrows, nbk):
        # W[:, cols] -= E_k L[rows, panel k]^T   (store[off:] = the first of those rows of the panel)
        self._chk(
            self.lib.bgp_gemm_nt_sub_async_dev(self.h, self._p(w, woff), lde, self._p(ek), ldek, self._p(store, off), ld, me, nrows, nbk, 0),
            "bgp_gemm_nt_sub_async_dev",
        )

    def sumsq(self, v, n) -> float:
        """sum of squares of a vector, on the device (a 1 x n row block through the row-dot kernel)"""
        out = self.zeros(1)
        self._chk(self.lib.bgp_rowdot_dev(self.h, self._p(v), 1, 1, int(n), None, self._p(out)), "bgp_rowdot_dev")
        return float(self.to_host(out)[0])

    def var_finish(self, xq_dev, m, d, ssq, min_var):
        out = self.empty(m)
        self._chk(self.lib.bgp_var_finish_dev(self.h, self._p(xq_dev), m, d, self._p(ssq), float(min_var), self._p(out)), "bgp_var_finish_dev")
        return out

    def rowdot(self, e, lde, m, n, vec, voff, out, eoff=0):
        self._chk(
            self.lib.bgp_rowdot_dev(self.h, self._p(e, eoff), lde, m, n, self._p(vec, voff) if vec is not None else None, self._p(out)),
            "bgp_rowdot_dev",
        )

    def add_into(self, acc, t):
        with self.on_stream():
            acc += t

    def copy_into(self, dst, src):
        with self.on_stream():
            dst.copy_(src)

    def set_segment(self, vec, c0, seg):
        with self.on_stream():
            vec[c0 : c0 + seg.shape[0]] = seg

    def scalar(self, value):
        with self.on_stream():
            return self.torch.full((1,), float(value), dtype=self.torch.float64, device=self.device)


class ShardedExactGP:
    """Zero-mean exact GP with the factor sharded over the ranks of ``dist`` (``None`` = 1 rank)."""

    def __init__(self, backend, dist, rank: int, world: int, nb: int = 512, max_tries: int = 3, jitter0: float = 1e-8):
        self.be, self.dist, self.rank, self.world, self.nb = backend, dist, rank, world, nb
        self.max_tries, self.jitter0 = max_tries, jitter0
        self.lay = None
        self.lml = None
        self.jitter = 0.0
        self._times = {}
        self._shape = None
        self._factor_consumed = False  # lml_grad() has turned the stored factor into Sigma^-1
        self._comm = {}                # phase -> kind -> [calls, payload bytes, bytes received by THIS rank]
        self._phase = "other"

    def set_hyp(self, hyp) -> None:
        """New hyper-parameters for the next :meth:`fit` (every rank must pass the same vector).  The model is unfitted
        from here on: :meth:`predict` / :meth:`lml_grad` raise until :meth:`fit` has run with them (the single-GPU
        engine does the same in ``bgp_set_kernel``)."""
        self.hyp = np.asarray(hyp, dtype=np.float64).copy()
        eng = getattr(self, "engine", None)
        if eng is not None:
            eng.set_hyp(self.hyp)
        self.lml = None
        self._factor_consumed = False

    # ---- bookkeeping of the exchange (host side; the only scaling evidence obtainable without a multi-GPU node) ----
    def _count(self, kind: str, nbytes: int, root: int | None = None) -> None:
        """One collective of `nbytes` payload.  Bytes RECEIVED by this rank under the ring / chain algorithms RCCL uses on
        point-to-point xGMI links: broadcast - the payload on every rank but the root; all-gather of equal chunks -
        (w-1) chunks; all-reduce - 2 (w-1)/w of the payload; reduce - (w-1)/w of it (average over the chain)."""
        w = self.world
        if kind == "broadcast":
            recv = 0 if root == self.rank else nbytes
        elif kind == "all_gather":
            recv = (w - 1) * nbytes  # nbytes = one rank's chunk
        elif kind == "all_reduce":
            recv = 2 * nbytes * (w - 1) // w
        elif kind == "reduce":
            recv = nbytes * (w - 1) // w
        else:
            raise ValueError(kind)
        c = self._comm.setdefault(self._phase, {}).setdefault(kind, [0, 0, 0])
        c[0] += 1
        c[1] += int(nbytes)
        c[2] += int(recv)

    def comm_bytes(self, reset: bool = False) -> dict:
        """``{phase: {kind: (calls, payload_bytes, received_bytes)}}`` since the last reset; phases: ``fit`` (fill,
        factorisation, z / log det), ``predict``, ``grad_a`` (M = L^-1), ``grad_alpha``, ``grad_b`` (P = M^T M),
        ``grad_reduce``."""
        out = {ph: {k: tuple(v) for k, v in kinds.items()} for ph, kinds in self._comm.items()}
        if reset:
            self._comm = {}
        return out

    def _allreduce(self, t, op=None):
        if self.dist is not None:
            self._count("all_reduce", 8 * int(t.shape[0]))
            self.be.allreduce(self.dist, t, op=op)

    def timers(self) -> dict:
        """host wall-clock of the last fit / predict on this rank (seconds)"""
        return dict(self._times)

    def close(self) -> None:
        eng = getattr(self, "engine", None)
        if eng is not None:
            self.be.sync()
            self.store = self.pbufs = self.inv = self.z = self.nbuf = self.sbuf = None
            self.wk = self._xq_dev = self.x_dev = self.y_dev = None  # (every device buffer goes before the engine does)
            self._shape = None
            eng.close()
            self.engine = None

    # ---- fit ------------------------------------------------------------------------------------
    def _allocate(self, n: int, d: int, ride: int = 0):
        lay = self.lay = PanelLayout(n, self.nb, self.world, ride)
        if self._shape == (n, d, ride):
            return lay
        be = self.be
        mine = lay.local_panels(self.rank)
        self.poff, total = lay.offsets(self.rank)
        self.store = be.empty(max(1, total))
        self.inv = {j: be.empty((lay.width(j) // 64) * 4096) for j in mine}
        # packed panel + failing-minor flag slot; two of them: the broadcast of panel k+1 is in flight while panel k is read.
        # The gradient's step (B) reuses them as the gathered / the assembled transposed row block, whose leading dimension
        # is a whole number of rounds of `world` panels: up to (world - 1) nb rows more than the matrix has
        nslots = -(-lay.npanels // self.world)
        per_buf = max(lay.nb * lay.nrows + 1, self.world * nslots * lay.nb * lay.nb)
        self.pbufs = [be.empty(per_buf), be.empty(per_buf)]
        self.sbuf = be.empty(nslots * lay.nb * lay.nb) if self.dist is not None else None  # this rank's pieces of one row block
        self.wk = [be.empty(lay.nb * lay.nb) for _ in range(4)]  # nb x nb work blocks of the gradient
        self.nbuf, self._per_buf = None, per_buf  # the gradient's negated row block (allocated by the first lml_grad)
        self._shape = (n, d, ride)
        return lay

    def fit(self, x: np.ndarray, y: np.ndarray) -> float:
        t_start = time.perf_counter()
        self._upload(x, y, None)
        self._phase = "fit"
        self._fit_resident()
        self._times["fit_s"] = time.perf_counter() - t_start
        return self.lml

    def _upload(self, x, y, xq):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64).reshape(-1)
        n, d = x.shape
        be = self.be
        self._xq_dev, self._m = None, 0
        ride = 0
        if xq is not None:
            xq = np.ascontiguousarray(xq, dtype=np.float64)
            self._m = xq.shape[0]
            ride = round_up(self._m, 64)
        self._allocate(n, d, ride)
        self.n, self.d = n, d
        self.x_dev, self.y_dev = be.upload(x), be.upload(y)
        if xq is not None:
            self._xq_dev = be.upload(xq)

    def fit_predict(self, x: np.ndarray, y: np.ndarray, xq: np.ndarray, min_var: float = 1e-10):
        """Fit and the first prediction in ONE pass over the panels - the sharded form of ``bgp_fit_predict``, which is the
        reference's actual flow (the model is built lazily and the first ``predict`` triggers the factorisation,
        ``src/batt_models/battcellgp_full.py:171-173``): the cross-covariance rows ``K(xq, X)`` are appended below the
        augmented block of every panel and ride through the factorisation, so ``V^T = K_*X L^-T`` comes out of the
        Cholesky itself, spread over the owners of the panels.  ``mean = V^T z`` and ``var = k_** - rowsumsq(V^T)`` are
        local row-dots per panel + two small all-reduces: no right-looking pass over the factor, no per-panel reduce, and
        the ``M N^2`` flop of the solve run inside the rank-``nb`` updates, spread over all ranks.  Each broadcast grows
        by ``M_pad x nb`` doubles.  Returns ``(lml, mean, var)``, identical on every rank; the model stays fitted
        (:meth:`predict` with other queries walks the stored factor)."""
        t_start = time.perf_counter()
        self._upload(x, y, xq)
        self._phase = "fit"
        self._fit_resident()
        lay, be = self.lay, self.be
        m, ride = self._m, lay.ride
        mean_p, var_p, tmp = be.zeros(ride), be.zeros(ride), be.zeros(ride)
        for j in lay.local_panels(self.rank):
            c0, w = lay.col0(j), lay.width(j)
            eoff = self.poff[j] + lay.sigma_rows(j) + AUG  # V^T[:, panel j]: [ride, w], leading dimension of the panel
            be.rowdot(self.store, lay.ld(j), ride, w, self.z, c0, tmp, eoff=eoff)
            be.add_into(mean_p, tmp)
            be.rowdot(self.store, lay.ld(j), ride, w, None, 0, tmp, eoff=eoff)
            be.add_into(var_p, tmp)
        self._allreduce(mean_p)
        self._allreduce(var_p)
        mean = be.to_host(mean_p)[:m]
        var = be.to_host(be.var_finish(self._xq_dev, m, self.d, var_p, min_var))[:m]
        self._times["fit_predict_s"] = time.perf_counter() - t_start
        return self.lml, mean, var

    def _fit_resident(self) -> float:
        """fill + jittered factorisation + z + LML on the resident inputs"""
        be, lay, n, d = self.be, self.lay, self.n, self.d
        mine = lay.local_panels(self.rank)
        self._factor_consumed = False
        jitter = 0.0
        for attempt in range(self.max_tries + 1):
            be.flag_reset()
            for j in mine:
                be.fill_panel(self.store, self.poff[j], lay.ld(j), self.x_dev, n, d, lay.col0(j), lay.width(j), lay.sigma_rows(j), self.y_dev, jitter)
                if lay.ride:
                    be.ride_fill(self.store, self.poff[j] + lay.sigma_rows(j) + AUG, lay.ld(j), self._xq_dev, self._m, lay.ride,
                                 self.x_dev, n, d, lay.col0(j), lay.width(j))
            self._enqueue_factorisation()
            info = self._collect_flag()  # the ONE host synchronisation of the attempt
            if info == 0:
                break
            if attempt == self.max_tries:
                from .engine import NotPSDError

                raise NotPSDError(f"matrix not positive definite (leading minor {info}) after jitter up to {jitter:.1e}")
            jitter = self.jitter0 * 10.0**attempt
        self.jitter = jitter

        # z^T (augmented row) and log det: every rank contributes the segments of its panels
        z = be.zeros(lay.npad)
        logdet = 0.0
        for j in mine:
            c0, w = lay.col0(j), lay.width(j)
            be.set_segment(z, c0, be.aug_row(self.store, self.poff[j], lay.ld(j), lay.sigma_rows(j), w))
            logdet += be.diag_logsum(self.store, self.poff[j], lay.ld(j), w)
        ld_t = be.scalar(logdet)
        self._allreduce(z)
        self._allreduce(ld_t)
        self.z = z
        zz = be.sumsq(z, lay.npad)
        self.lml = -0.5 * zz - float(be.to_host(ld_t)[0]) - 0.5 * n * math.log(2.0 * math.pi)
        return self.lml

    def _collect_flag(self) -> int:
        """failing leading minor seen by ANY rank (0 = none): drains this rank's pipeline, then one MAX all-reduce"""
        flag = self.be.flag_read()
        if self.dist is None:
            return flag
        t = self.be.scalar(flag)
        self._allreduce(t, op=self.dist.ReduceOp.MAX)
        return int(self.be.to_host(t)[0])

    def _enqueue_factorisation(self) -> None:
        """Right-looking factorisation with a one-panel look-ahead on the exchange: while every rank applies
        panel k to its own panels, the owner of panel k+1 has already updated, factored and packed that
        panel and its broadcast is in flight, so the xGMI transfer of panel k+1 hides behind the rank-nb updates of
        step k.  ONE collective per step (the failing-minor flag travels as the element behind the packed panel) and
        no host synchronisation: everything is ordered by the engine's stream."""
        lay, be = self.lay, self.be
        mine = set(lay.local_panels(self.rank))
        pbufs = self.pbufs

        def factor_and_pack(k, buf):
            be.factor_pack(self.store, self.poff[k], lay.ld(k), lay.rows_from(k), lay.width(k), self.inv[k], buf, lay.col0(k))

        def start_bcast(k, buf):
            if self.dist is None:
                return None
            cnt = lay.width(k) * lay.rows_from(k) + 1
            self._count("broadcast", 8 * cnt, lay.owner(k))
            return be.bcast_start(self.dist, buf[:cnt], lay.owner(k))

        if self.rank == lay.owner(0):
            factor_and_pack(0, pbufs[0])
        work = start_bcast(0, pbufs[0])
        for k in range(lay.npanels):
            cur, nxt = pbufs[k % 2], pbufs[(k + 1) % 2]
            c0, nbk, rows = lay.col0(k), lay.width(k), lay.rows_from(k)
            flag_idx = nbk * rows
            if work is not None:
                be.bcast_wait(work)
            if self.rank != lay.owner(k):
                be.flag_merge(cur, flag_idx)  # a failure upstream poisons this rank's pipeline as well
            if k == lay.npanels - 1:
                break

            def item(j):
                return (self.poff[j], lay.ld(j), lay.rows_from(j), lay.width(j), lay.col0(j) - c0)

            todo = sorted(p for p in mine if p > k)
            if (k + 1) in mine:  # look-ahead: my next panel first, then factor and ship it
                be.update_panels(self.store, [item(k + 1)], cur, rows, nbk, flag_idx)
                factor_and_pack(k + 1, nxt)
                todo.remove(k + 1)
            work = start_bcast(k + 1, nxt)
            be.update_panels(self.store, [item(j) for j in todo], cur, rows, nbk, flag_idx)

    # ---- predict ----------------------------------------------------------------------------------
    PREDICT_ROWS = 4096  # query rows per right-looking pass of predict()

    def predict(self, xq: np.ndarray, min_var: float = 1e-10):
        """(mean, var) of the latent f at xq, identical on every rank.

        Right-looking over the panels with a ONE-STEP LOOK-AHEAD on the exchange, like the factorisation: at step k the
        owner solves ``E_k``, applies it to the columns of panel k+1 FIRST, the reduce of block k+1 to its owner starts
        (asynchronously, on the communicator's stream), and the owner's update of all later columns runs underneath it.

        The pass keeps an ``M_pad x N_pad`` accumulator on EVERY rank, so the queries go through in blocks of
        ``PREDICT_ROWS`` rows: 8.6 GB per rank at N = 262 144 whatever M is - the ``add_time_steps`` shape (M ~ N,
        ``src/batt_models/battgp_full.py:86-96``) would otherwise ask every rank for a second N x N buffer."""
        if self.lml is None:
            raise RuntimeError("predict: fit first (set_hyp() leaves the model unfitted)")
        t_start = time.perf_counter()
        if self._factor_consumed:  # same data, same hyper-parameters, same ladder: the factor comes back as it was
            self._phase = "fit"
            self._fit_resident()
        self._phase = "predict"
        xq = np.ascontiguousarray(xq, dtype=np.float64)
        step = max(16, int(self.PREDICT_ROWS))
        parts = [self._predict_block(xq[r0 : r0 + step], min_var) for r0 in range(0, max(1, xq.shape[0]), step)]
        mean = np.concatenate([p[0] for p in parts])
        var = np.concatenate([p[1] for p in parts])
        self._times["predict_s"] = time.perf_counter() - t_start
        return mean, var

    def _predict_block(self, xq: np.ndarray, min_var: float):
        """one right-looking pass for a block of query rows (see :meth:`predict`)"""
        lay, be = self.lay, self.be
        m = xq.shape[0]
        mpad = round_up(m, 16)
        lde = mpad
        xq_dev = be.upload(xq)
        # W_r: rank 0 starts from the cross-covariance, the others from zero; sum_r W_r[:, k] at step k
        # is K_*X[:, k] minus every contribution of the panels before k
        w = be.zeros(mpad * lay.npad)
        if self.rank == 0:
            be.cross_fill(xq_dev, m, mpad, self.x_dev, self.n, self.d, lay.npad, w, lde)
        mean_p, var_p, tmp = be.zeros(mpad), be.zeros(mpad), be.zeros(mpad)
        ek = be.empty(mpad * lay.nb)

        def start_reduce(k):
            if self.dist is None:
                return None
            c0, nbk = lay.col0(k), lay.width(k)
            self._count("reduce", 8 * nbk * lde, lay.owner(k))
            return be.reduce_start(self.dist, w[c0 * lde : (c0 + nbk) * lde], lay.owner(k))

        work = start_reduce(0)
        for k in range(lay.npanels):
            owner, c0, nbk = lay.owner(k), lay.col0(k), lay.width(k)
            mine = self.rank == owner
            if work is not None:
                be.bcast_wait(work)
            nxt = lay.width(k + 1) if k + 1 < lay.npanels else 0
            if mine:
                be.copy_into(ek[: nbk * lde], w[c0 * lde : (c0 + nbk) * lde])
                be.solve_panel(ek, 0, lde, mpad, self.store, self.poff[k], lay.ld(k), nbk, self.inv[k])
                if nxt:  # what block k+1 still misses from this rank
                    be.update_rows(w, (c0 + nbk) * lde, lde, mpad, ek, lde, self.store, self.poff[k] + nbk, lay.ld(k), nxt, nbk)
            if nxt:
                work = start_reduce(k + 1)
            if mine:
                rest = lay.npad - (c0 + nbk + nxt)
                if rest > 0:
                    be.update_rows(w, (c0 + nbk + nxt) * lde, lde, mpad, ek, lde, self.store, self.poff[k] + nbk + nxt, lay.ld(k), rest, nbk)
                be.rowdot(ek, lde, mpad, nbk, self.z, c0, tmp)
                be.add_into(mean_p, tmp)
                be.rowdot(ek, lde, mpad, nbk, None, 0, tmp)
                be.add_into(var_p, tmp)
        self._allreduce(mean_p)
        self._allreduce(var_p)
        mean = be.to_host(mean_p)[:m]
        var = be.to_host(be.var_finish(xq_dev, m, self.d, var_p, min_var))[:m]
        return mean, var

    # ---- gradient ---------------------------------------------------------------------------------
    def lml_grad(self) -> np.ndarray:
        """``d lml / d theta`` of the last fit (layout of ``hyp``), identical on every rank: the analytic
        ``1/2 tr((alpha alpha^T - Sigma^-1) dSigma/dtheta)`` that one backward pass gives the reference
        (``src/gp/training.py:39-41``), with ``Sigma^-1`` formed IN PLACE over the distributed factor - the two steps of
        the single-GPU ``bgp_lml_grad`` over block-cyclic column panels:

        (A) ``M = L^-1``, right-looking.  Step k: the owner packs ``[M_kk ; L[K1:, k]]`` (the shape of a factor panel) and
            broadcasts it; every rank transforms row block k of its panels ``j < k`` (``X_kj <- M_kk X_kj``) and applies
            ONE rank-``nb`` update ``X[K1:, j] -= L[K1:, k] X_kj`` per panel on the MFMA kernel; the owner writes its own
            panel ``[M_kk ; -L[K1:, k] M_kk]``.  The pack and broadcast of panel k+1 run ahead of the updates of step k.
        (B) ``P = M^T M``, row blocks from the top.  Step k: row block k of ``M`` (spread over the owners of the panels
            ``j <= k``) is assembled, transposed, into ``Wt[K1, nb]`` on every rank by ONE ALL-GATHER, issued one step
            ahead: each rank packs the pieces of ITS panels slot by slot (panel j -> slot j // world), the gathered
            chunks are put into the natural row order by one strided copy (panel = slot * world + rank); every rank adds
            the rank-``nb`` SYRK ``P[J0:K0, j] += Wt[J0:K0] Wt[j]^T`` to its panels ``j < k`` - run as ``-= Wt (-Wt)^T``
            against a negated copy of the block, on the same kernel and atomic epilogue as the factorisation's updates -
            transforms their row block k (``M_kj <- M_kk^T M_kj``), the owner forms ``P_kk``.

        ``alpha = M^T z`` is local per panel between (A) and (B); the reduction over each local panel's lower trapezoid
        re-evaluates the kernel derivatives, and ONE all-reduce of the few accumulators ends it.  Per rank: ``2/3 N^3 /
        world`` flop, ~``4 N^2 (w-1)/w`` bytes received in each of (A) and (B) (:meth:`comm_bytes` counts them), no second
        ``N^2`` buffer (the packed-panel buffers of the factorisation are reused; one more panel-sized buffer holds the
        negated row block of step (B)).  The factor is consumed: the next :meth:`predict` re-runs the factorisation on
        the resident inputs."""
        if self.lml is None:
            raise RuntimeError("lml_grad: fit first (set_hyp() leaves the model unfitted)")
        if self._factor_consumed:
            self._phase = "fit"
            self._fit_resident()
        t_start = time.perf_counter()
        lay, be, dist = self.lay, self.be, self.dist
        world = self.world
        mine = lay.local_panels(self.rank)
        nb, npad = lay.nb, lay.npad
        mk, mt, t1, t3 = self.wk
        pbufs = self.pbufs
        self._factor_consumed = True
        self._phase = "grad_a"

        def geom(k):
            c0, nbk = lay.col0(k), lay.width(k)
            return c0, nbk, c0 + nbk, npad - c0  # K0, width, K1, rows of the panel inside the matrix

        # ---- (A) M = L^-1 ------------------------------------------------------------------------------
        def pack(k, buf):
            K0, nbk, K1, R = geom(k)
            off, ld = self.poff[k], lay.ld(k)
            be.panel_inverse(self.store, off, ld, nbk, self.inv[k], mk)
            be.block_copy(mk, 0, nbk, nbk, nbk, buf, 0, R)
            if R > nbk:
                be.block_copy(self.store, off + nbk, ld, R - nbk, nbk, buf, nbk, R)

        def start_bcast(k, buf):
            if dist is None:
                return None
            _, nbk, _, R = geom(k)
            self._count("broadcast", 8 * R * nbk, lay.owner(k))
            return be.bcast_start(dist, buf[: R * nbk], lay.owner(k))

        if self.rank == lay.owner(0):
            pack(0, pbufs[0])
        work = start_bcast(0, pbufs[0])
        for k in range(lay.npanels):
            cur, nxt = pbufs[k % 2], pbufs[(k + 1) % 2]
            K0, nbk, K1, R = geom(k)
            if work is not None:
                be.bcast_wait(work)
            if k + 1 < lay.npanels:  # look-ahead of the exchange: panel k+1 of L is untouched until its own step
                if self.rank == lay.owner(k + 1):
                    pack(k + 1, nxt)
                work = start_bcast(k + 1, nxt)
            below = R - nbk
            for j in (p for p in mine if p < k):
                J0, nbj = lay.col0(j), lay.width(j)
                off, ld = self.poff[j], lay.ld(j)
                blk = off + (K0 - J0)  # row block k of panel j: [nbk, nbj]
                be.block_copy(self.store, blk, ld, nbk, nbj, t1, 0, nb, trans=1)  # t1 = X_kj^T [nbj, nbk]
                be.gemm(1, t3, 0, nb, t1, 0, nb, cur, 0, R, nbj, nbk, nbk, btri=1)  # t3 = X_kj^T M_kk^T
                be.block_copy(t3, 0, nb, nbj, nbk, self.store, blk, ld, trans=1)
                if below > 0:
                    be.gemm(2 if nbk >= 256 else 0, self.store, off + (K1 - J0), ld, cur, nbk, R, t3, 0, nb, below, nbj, nbk)
            if k in mine:
                off, ld = self.poff[k], lay.ld(k)
                if below > 0:
                    be.block_copy(cur, 0, R, nbk, nbk, mt, 0, nbk, trans=1, scale=-1.0)  # -M_kk^T
                    be.gemm(1, self.store, off + nbk, ld, cur, nbk, R, mt, 0, nbk, below, nbk, nbk)
                be.block_copy(cur, 0, R, nbk, nbk, self.store, off, ld)

        # ---- alpha = M^T z: each panel's columns against the rows it stores -----------------------------------
        self._phase = "grad_alpha"
        alpha = be.zeros(npad)
        for j in mine:
            J0, nbj = lay.col0(j), lay.width(j)
            be.gemv_t(self.store, self.poff[j], lay.ld(j), npad - J0, nbj, self.z, J0, alpha, J0)
        self._allreduce(alpha)

        # ---- (B) P = M^T M -----------------------------------------------------------------------------
        # Wt = pbufs[0]: (row block k of M)^T in natural row order, leading dimension ldw(k) = a whole number of rounds of
        # `world` panels (>= K1); pbufs[1]: the gathered chunks; sbuf: this rank's chunk.  All three single-buffered: the
        # gather of block k+1 is issued behind the copy that empties pbufs[1], and Wt is rewritten only at the next step.
        # (One rank: the chunk IS the block - packed straight into Wt, which then alternates between the two buffers,
        # block k+1 being packed while step k still reads block k.)
        self._phase = "grad_b"
        gathered = pbufs[1]
        if self.nbuf is None:
            self.nbuf = be.empty(self._per_buf)
        wneg = self.nbuf

        def slots(k):
            return -(-(k + 1) // world)  # panels 0..k dealt round-robin: slots per rank

        def gather(k):
            """my pieces of (row block k of M)^T, slot by slot; then the all-gather of every rank's chunk"""
            K0, nbk, K1, _ = geom(k)
            rows = slots(k) * nb  # rows of a chunk
            dst = self.sbuf if dist is not None else pbufs[k % 2]
            for j in (p for p in mine if p < k):
                J0, nbj = lay.col0(j), lay.width(j)
                be.block_copy(self.store, self.poff[j] + (K0 - J0), lay.ld(j), nbk, nbj, dst, (j // world) * nb, rows, trans=1)
            if k in mine:
                be.block_copy(self.store, self.poff[k], lay.ld(k), nbk, nbk, dst, (k // world) * nb, rows, trans=1, tri=1)  # M_kk^T
            if dist is None:
                return None
            self._count("all_gather", 8 * rows * nbk)
            return be.allgather_start(dist, gathered[: world * rows * nbk], self.sbuf[: rows * nbk])

        work = gather(0)
        for k in range(lay.npanels):
            K0, nbk, K1, _ = geom(k)
            ldw = world * slots(k) * nb
            wt = pbufs[0] if dist is not None else pbufs[k % 2]
            if work is not None:
                be.bcast_wait(work)
                be.unshuffle(gathered, wt, world, slots(k), nb, nbk)
            if k + 1 < lay.npanels:  # row block k+1 is untouched by step k: its exchange runs under this step's updates
                work = gather(k + 1)
            first = min((p for p in mine if p < k), default=None)
            if first is not None:  # -Wt[J0:K0] from my first panel's rows on: the B operand of this step's SYRKs
                F0 = lay.col0(first)
                be.block_copy(wt, F0, ldw, K0 - F0, nbk, wneg, F0, ldw, scale=-1.0)
            for j in (p for p in mine if p < k):
                J0, nbj = lay.col0(j), lay.width(j)
                off, ld = self.poff[j], lay.ld(j)
                be.gemm(2 if nbk >= 256 else 0, self.store, off, ld, wt, J0, ldw, wneg, J0, ldw, K0 - J0, nbj, nbk, lower=1)
                be.gemm(1, t3, 0, nb, wt, J0, ldw, wt, K0, ldw, nbj, nbk, nbk)  # t3 = M_kj^T M_kk
                be.block_copy(t3, 0, nb, nbj, nbk, self.store, off + (K0 - J0), ld, trans=1)
            if k in mine:
                be.gemm(1, self.store, self.poff[k], lay.ld(k), wt, K0, ldw, wt, K0, ldw, nbk, nbk, nbk)  # P_kk = M_kk^T M_kk

        # ---- reduction over the local panels, one all-reduce of the accumulators ----------------------------
        self._phase = "grad_reduce"
        acc = be.grad_acc()
        for j in mine:
            J0, nbj = lay.col0(j), lay.width(j)
            be.grad_reduce(self.x_dev, self.n, self.d, J0, npad - J0, nbj, self.store, self.poff[j], lay.ld(j), alpha, acc)
        self._allreduce(acc)
        grad = be.grad_finish(acc, self.d)
        self._times["grad_s"] = time.perf_counter() - t_start
        return grad


def make_sharded_gp(kernel_id: int, hyp, nb: int = 512, backend_name: str | None = None, local_rank: int | None = None):
    """Build a :class:`ShardedExactGP` for the calling process: rank/world from the launcher's
    environment (torch.distributed.run), one GPU per rank (LOCAL_RANK), RCCL communicator."""
    import torch

    from . import parallel
    from .engine import ExactGPEngine

    rank, world, env_local = parallel.env_rank_world()
    local_rank = env_local if local_rank is None else local_rank
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    dist = parallel.init(backend_name or "nccl", device=device)
    eng = ExactGPEngine(kernel_id, hyp, device=local_rank)
    eng.set_options(nb_outer=max(512, nb))
    gp = ShardedExactGP(DeviceBackend(eng, device), dist, rank, world, nb=nb)
    gp.engine = eng  # keep the handle alive
    gp.kernel_id, gp.hyp = kernel_id, np.asarray(hyp, dtype=np.float64)
    return gp
