"""One exact GP whose covariance matrix is too large for a single GPU: column-panel block-cyclic
Cholesky over the ranks of a ``torch.distributed`` group (backend "nccl" = RCCL over xGMI; "gloo" in
the CPU tests).  BASELINE config 4 (N = 262 144 on 4 x MI355X); SURVEY section 8(e).

Distribution.  The matrix is cut into column panels of width ``nb``; panel ``j`` lives on rank
``j % world`` with ALL its rows from the diagonal down plus the 64-row augmented block whose row 0
carries ``y^T``.  Right-looking factorisation, one exchange per panel step:

    for k in panels:                                   (panel k is already on every rank)
        owner(k+1): C_{k+1} -= P_k ...; factor panel k+1 locally; pack it                     (look-ahead)
        all ranks:  start the broadcast of packed panel k+1 [(Npad + 64 - (k+1) nb) x nb + flag]  <- RCCL, async
        every rank: C_j -= P_k[rows >= j nb] P_k[rows of j]^T  for each of ITS panels j > k   (MFMA)
        wait for the broadcast

so each rank receives ~4 N^2 (w-1)/w bytes in total and runs 1/w of the N^3/3 flops.  The forward
solve rides along in the augmented row (``z^T`` comes out of the factorisation), ``log det`` and the
``z`` segments are combined with small all-reduces.  Prediction pushes the query block through the
factor right-looking: the owner of panel k finishes ``E_k`` (reduce of the pending contributions),
updates its own accumulator for all later columns, and adds its share of ``mean = V^T z`` and
``var = k_** - rowsumsq(V^T)``.

The numerical work is done by a *backend*: ``DeviceBackend`` drives the HIP kernels of libbattgp.so
on torch-owned device buffers (torch = container + communicator only); the tests supply a numpy
backend to check the distributed algorithm on CPU/gloo.
"""

from __future__ import annotations

import ctypes as C
import math

import numpy as np

AUG = 64  # rows of the augmented block (BGP_AUG)


def round_up(v: int, q: int) -> int:
    return (v + q - 1) // q * q


class PanelLayout:
    def __init__(self, n: int, nb: int, world: int):
        if nb % 64 != 0 or nb < 64:
            raise ValueError("nb must be a positive multiple of 64")
        self.n, self.nb, self.world = n, nb, world
        self.npad = round_up(n, 64)
        self.npanels = -(-self.npad // nb)
        self.nrows = self.npad + AUG  # leading dimension of every stored column

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
        """rows of panel j from its diagonal down, augmented block included"""
        return self.nrows - self.col0(j)


class DeviceBackend:
    """HIP kernels through the C-ABI on torch-owned device memory."""

    def __init__(self, engine, device):
        import torch

        from . import _lib

        self.torch = torch
        self.eng = engine
        self.lib = _lib.load()
        self.h = engine._h
        self.device = device

    def _chk(self, rc, what):
        self.eng._check(rc, what)

    def zeros(self, n):
        # the fill runs on torch's stream, the engine's kernels on their own: finish it before handing
        # the buffer to the engine
        t = self.torch.zeros(int(n), dtype=self.torch.float64, device=self.device)
        self.after_comm()
        return t

    def empty(self, n):
        return self.torch.empty(int(n), dtype=self.torch.float64, device=self.device)

    def upload(self, a):
        t = self.torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=self.device)
        self.after_comm()
        return t

    def to_host(self, t):
        return t.detach().cpu().numpy()

    def sync(self):
        """drain the engine's streams (its kernels run outside torch's stream)"""
        self._chk(self.lib.bgp_sync(self.h), "bgp_sync")

    def after_comm(self):
        """RCCL work is ordered on torch's stream, the HIP kernels run on the engine's: make the received
        data visible before the next engine launch"""
        self.torch.cuda.current_stream(self.device).synchronize()

    @staticmethod
    def _p(t, off=0):
        return C.c_void_p(t.data_ptr() + 8 * int(off))

    def fill_panel(self, store, ld, lcol0, x_dev, n, d, col0, ncols, rows, y_dev, extra_diag):
        """store[(i-col0) ... ]: columns [col0, col0+ncols) rows [col0, npad) + augmented block"""
        base = lcol0 * ld + col0
        self._chk(
            self.lib.bgp_fill_block_dev(self.h, self._p(x_dev), n, d, col0, col0, rows - AUG, ncols, self._p(store, base), ld, float(extra_diag)),
            "bgp_fill_block_dev",
        )
        self._chk(
            self.lib.bgp_aug_rows_dev(self.h, self._p(y_dev), n, col0, ncols, self._p(store, lcol0 * ld + (ld - AUG)), ld),
            "bgp_aug_rows_dev",
        )

    def factor_panel(self, store, ld, lcol0, col0, rows, nbk, inv):
        info = C.c_int(0)
        self._chk(
            self.lib.bgp_factor_panel_dev(self.h, self._p(store, lcol0 * ld + col0), ld, rows, nbk, self._p(inv), C.byref(info)),
            "bgp_factor_panel_dev",
        )
        return int(info.value)

    def factor_pack(self, store, ld, lcol0, col0, rows, nbk, inv, pbuf):
        """factor the panel and leave its packed copy in pbuf (one C call; see bgp_factor_pack_panel_dev)"""
        info = C.c_int(0)
        self._chk(
            self.lib.bgp_factor_pack_panel_dev(
                self.h, self._p(store, lcol0 * ld + col0), ld, rows, nbk, self._p(inv), self._p(pbuf), C.byref(info)
            ),
            "bgp_factor_pack_panel_dev",
        )
        return int(info.value)

    def pack_panel(self, store, ld, lcol0, col0, rows, nbk, pbuf):
        """pbuf[r + c*rows] = store[(col0 + r) + (lcol0 + c)*ld]"""
        t = self.torch
        src = store.as_strided((nbk, rows), (ld, 1), lcol0 * ld + col0)
        pbuf[: nbk * rows].view(nbk, rows).copy_(src)
        t.cuda.current_stream(self.device).synchronize()

    def update_panel(self, store, ld, lcol0, colj, rows_j, nbj, pbuf, ldp, off, nbk):
        self._chk(
            self.lib.bgp_gemm_nt_sub_async_dev(
                self.h, self._p(store, lcol0 * ld + colj), ld, self._p(pbuf, off), ldp, self._p(pbuf, off), ldp, rows_j, nbj, nbk, 1
            ),
            "bgp_gemm_nt_sub_async_dev",
        )

    def update_panels(self, store, ld, items, pbuf, ldp, nbk):
        """items: (lcol0, colj, rows_j, nbj, p_off) per local panel - all updates of one step in one C call"""
        if not items:
            return
        desc = np.ascontiguousarray([[lc * ld + cj, rows_j, nbj, off] for lc, cj, rows_j, nbj, off in items], dtype=np.int64)
        self._chk(
            self.lib.bgp_update_panels_dev(
                self.h, self._p(store), ld, desc.ctypes.data_as(C.POINTER(C.c_int64)), len(items), self._p(pbuf), ldp, nbk
            ),
            "bgp_update_panels_dev",
        )

    def diag_logsum(self, store, ld, lcol0, col0, nbk) -> float:
        out = C.c_double(0.0)
        self._chk(self.lib.bgp_diag_logsum_dev(self.h, self._p(store, lcol0 * ld + col0), ld, nbk, C.byref(out)), "bgp_diag_logsum_dev")
        return float(out.value)

    def aug_row(self, store, ld, lcol0, nbk):
        """z segment = row 0 of the augmented block under the panel (strided view, copied)"""
        return store.as_strided((nbk,), (ld,), lcol0 * ld + (ld - AUG)).clone()

    # ---- prediction ---------------------------------------------------------------------------
    def cross_fill(self, xq_dev, m, mpad, x_dev, n, d, npad, out, lde):
        # E[m, i] = k(xq_m, x_i): rows = queries (offset 0), columns = training points
        p = self.lib.bgp_fill_dev
        self._chk(p(self.h, self._p(xq_dev), m, self._p(x_dev), n, d, self._p(out), lde, 0, 0.0), "bgp_fill_dev")

    def solve_panel(self, e, lde, ecol0, me, store, ld, lcol0, col0, nbk, inv):
        self._chk(
            self.lib.bgp_solve_panel_dev(self.h, self._p(e, ecol0 * lde), lde, me, self._p(store, lcol0 * ld + col0), ld, nbk, self._p(inv)),
            "bgp_solve_panel_dev",
        )

    def update_rows(self, w, lde, wcol0, me, ek, ldek, store, ld, lcol0, row0, nrows, nbk):
        # W[:, wcol0 : wcol0+nrows] -= E_k L[row0 : row0+nrows, panel k]^T
        self._chk(
            self.lib.bgp_gemm_nt_sub_async_dev(
                self.h, self._p(w, wcol0 * lde), lde, self._p(ek), ldek, self._p(store, lcol0 * ld + row0), ld, me, nrows, nbk, 0
            ),
            "bgp_gemm_nt_sub_async_dev",
        )

    def sumsq(self, v, n) -> float:
        """sum of squares of a vector, on the device (a 1 x n row block through the row-dot kernel)"""
        out = self.zeros(1)
        self._chk(self.lib.bgp_rowdot_dev(self.h, self._p(v), 1, 1, int(n), None, self._p(out)), "bgp_rowdot_dev")
        self.sync()
        return float(out[0])

    def var_finish(self, xq_dev, m, d, ssq, min_var):
        out = self.empty(m)
        self._chk(self.lib.bgp_var_finish_dev(self.h, self._p(xq_dev), m, d, self._p(ssq), float(min_var), self._p(out)), "bgp_var_finish_dev")
        self.sync()
        return out

    def rowdot(self, e, lde, m, n, vec, out):
        self._chk(self.lib.bgp_rowdot_dev(self.h, self._p(e), lde, m, n, self._p(vec) if vec is not None else None, self._p(out)), "bgp_rowdot_dev")


class ShardedExactGP:
    """Zero-mean exact GP with the factor sharded over the ranks of ``dist`` (``None`` = 1 rank)."""

    def __init__(self, backend, dist, rank: int, world: int, nb: int = 512, max_tries: int = 3, jitter0: float = 1e-8):
        self.be, self.dist, self.rank, self.world, self.nb = backend, dist, rank, world, nb
        self.max_tries, self.jitter0 = max_tries, jitter0
        self.lay = None
        self.lml = None
        self.jitter = 0.0

    def set_hyp(self, hyp) -> None:
        """New hyper-parameters for the next :meth:`fit` (every rank must pass the same vector)."""
        self.hyp = np.asarray(hyp, dtype=np.float64).copy()
        eng = getattr(self, "engine", None)
        if eng is not None:
            eng.set_hyp(self.hyp)

    # ---- collectives (torch.distributed on backend buffers; numpy buffers are wrapped in place) ------
    @staticmethod
    def _t(buf):
        if isinstance(buf, np.ndarray):
            import torch

            return torch.from_numpy(buf)
        return buf

    def _bcast(self, t, src):
        if self.dist is not None:
            self.dist.broadcast(self._t(t), src=src)
            self.be.after_comm()

    def _allreduce(self, t):
        if self.dist is not None:
            self.dist.all_reduce(self._t(t))
            self.be.after_comm()

    def _reduce(self, t, dst):
        if self.dist is not None:
            self.dist.reduce(self._t(t), dst=dst)
            self.be.after_comm()

    # ---- fit ------------------------------------------------------------------------------------
    def fit(self, x: np.ndarray, y: np.ndarray) -> float:
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64).reshape(-1)
        n, d = x.shape
        be = self.be
        lay = self.lay = PanelLayout(n, self.nb, self.world)
        self.n, self.d = n, d
        ld = lay.nrows
        mine = lay.local_panels(self.rank)
        self.x_dev, self.y_dev = be.upload(x), be.upload(y)
        ncols_local = sum(lay.width(j) for j in mine)
        self.store = be.empty(max(1, ncols_local) * ld)
        self.lcol0 = {j: lay.local_index(j) * lay.nb for j in mine}
        self.inv = {j: be.empty((lay.width(j) // 64) * 4096) for j in mine}
        pbufs = [be.empty(lay.nb * ld + 1), be.empty(lay.nb * ld + 1)]  # packed panel + failing-minor flag

        jitter = 0.0
        for attempt in range(self.max_tries + 1):
            for j in mine:
                be.fill_panel(self.store, ld, self.lcol0[j], self.x_dev, n, d, lay.col0(j), lay.width(j), lay.rows_from(j), self.y_dev, jitter)
            info = self._factor(pbufs)
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
            z[c0 : c0 + w] = be.aug_row(self.store, ld, self.lcol0[j], w)
            logdet += be.diag_logsum(self.store, ld, self.lcol0[j], c0, w)
        ld_t = be.zeros(1)
        ld_t[0] = logdet
        self._allreduce(z)
        self._allreduce(ld_t)
        self.z = z
        zz = be.sumsq(z, lay.npad)
        self.lml = -0.5 * zz - float(ld_t[0]) - 0.5 * n * math.log(2.0 * math.pi)
        return self.lml

    def _factor(self, pbufs) -> int:
        """Right-looking factorisation with a one-panel look-ahead on the exchange: while every rank applies
        panel k to its own panels, the owner of panel k+1 has already updated, factored and packed that
        panel and its broadcast is in flight (``async_op``), so the xGMI transfer of panel k+1 hides behind
        the rank-nb updates of step k.  ONE collective per step: the failing-minor flag travels as the
        element behind the packed panel."""
        lay, be, ld = self.lay, self.be, self.lay.nrows
        mine = set(lay.local_panels(self.rank))

        def factor_and_pack(k, buf):
            c0, nbk, rows = lay.col0(k), lay.width(k), lay.rows_from(k)
            info = be.factor_pack(self.store, ld, self.lcol0[k], c0, rows, nbk, self.inv[k], buf)
            buf[nbk * rows] = float(info + c0 if info else 0)
            be.after_comm()

        def start_bcast(k, buf):
            if self.dist is None:
                return None
            n_el = lay.width(k) * lay.rows_from(k) + 1
            return self.dist.broadcast(self._t(buf[:n_el]), src=lay.owner(k), async_op=True)

        if self.rank == lay.owner(0):
            factor_and_pack(0, pbufs[0])
        work = start_bcast(0, pbufs[0])
        for k in range(lay.npanels):
            cur, nxt = pbufs[k % 2], pbufs[(k + 1) % 2]
            c0, nbk, rows = lay.col0(k), lay.width(k), lay.rows_from(k)
            if work is not None:
                work.wait()
                be.after_comm()
            flag = float(cur[nbk * rows])
            if flag != 0.0:
                return int(flag)
            if k == lay.npanels - 1:
                break
            todo = sorted(p for p in mine if p > k)
            if (k + 1) in mine:  # look-ahead: my next panel first, then factor and ship it
                cj, nbj = lay.col0(k + 1), lay.width(k + 1)
                be.update_panel(self.store, ld, self.lcol0[k + 1], cj, lay.rows_from(k + 1), nbj, cur, rows, cj - c0, nbk)
                be.sync()
                factor_and_pack(k + 1, nxt)
                todo.remove(k + 1)
            work = start_bcast(k + 1, nxt)
            be.update_panels(
                self.store, ld, [(self.lcol0[j], lay.col0(j), lay.rows_from(j), lay.width(j), lay.col0(j) - c0) for j in todo], cur, rows, nbk
            )
            be.sync()
        return 0

    # ---- predict ----------------------------------------------------------------------------------
    def predict(self, xq: np.ndarray, min_var: float = 1e-10):
        """(mean, var) of the latent f at xq, identical on every rank."""
        lay, be, ld = self.lay, self.be, self.lay.nrows
        xq = np.ascontiguousarray(xq, dtype=np.float64)
        m = xq.shape[0]
        mpad = round_up(m, 16)
        lde = mpad
        xq_dev = be.upload(xq)
        # W_r: rank 0 starts from the cross-covariance, the others from zero; sum_r W_r[:, k] at step k
        # is K_*X[:, k] minus every contribution of the panels before k
        w = be.zeros(mpad * lay.npad)
        if self.rank == 0:
            be.cross_fill(xq_dev, m, mpad, self.x_dev, self.n, self.d, lay.npad, w, lde)
            be.sync()
        mean_p, var_p, tmp = be.zeros(mpad), be.zeros(mpad), be.zeros(mpad)
        ek = be.empty(mpad * lay.nb)
        for k in range(lay.npanels):
            owner, c0, nbk = lay.owner(k), lay.col0(k), lay.width(k)
            blk = w[c0 * lde : (c0 + nbk) * lde]
            self._reduce(blk, owner)
            if self.rank == owner:
                ek[: nbk * lde] = blk
                be.after_comm()  # container copy (torch stream) must land before the engine's kernels read it
                be.solve_panel(ek, lde, 0, mpad, self.store, ld, self.lcol0[k], c0, nbk, self.inv[k])
                rest = lay.npad - (c0 + nbk)
                if rest > 0:
                    be.update_rows(w, lde, c0 + nbk, mpad, ek, lde, self.store, ld, self.lcol0[k], c0 + nbk, rest, nbk)
                be.rowdot(ek, lde, mpad, nbk, self.z[c0 : c0 + nbk], tmp)
                be.sync()
                mean_p += tmp
                be.after_comm()
                be.rowdot(ek, lde, mpad, nbk, None, tmp)
                be.sync()
                var_p += tmp
                be.after_comm()  # ... and the accumulations before `tmp` / `w` are touched again
        self._allreduce(mean_p)
        self._allreduce(var_p)
        mean = be.to_host(mean_p)[:m]
        var = be.to_host(be.var_finish(xq_dev, m, self.d, var_p, min_var))[:m]
        return mean, var


def make_sharded_gp(kernel_id: int, hyp, nb: int = 512, backend_name: str | None = None):
    """Build a :class:`ShardedExactGP` for the calling process: rank/world from the launcher's
    environment (torch.distributed.run), one GPU per rank (LOCAL_RANK), RCCL communicator."""
    import torch

    from . import parallel
    from .engine import ExactGPEngine

    rank, world, local_rank = parallel.env_rank_world()
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    dist = parallel.init(backend_name or "nccl", device=device)
    eng = ExactGPEngine(kernel_id, hyp, device=local_rank)
    eng.set_options(nb_outer=max(512, nb))
    gp = ShardedExactGP(DeviceBackend(eng, device), dist, rank, world, nb=nb)
    gp.engine = eng  # keep the handle alive
    gp.kernel_id, gp.hyp = kernel_id, np.asarray(hyp, dtype=np.float64)
    return gp
