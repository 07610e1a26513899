"""The sharded (column-panel block-cyclic) Cholesky GP on CPU: 2 ranks over gloo with a numpy backend
that stands in for the HIP kernels, checked against the single-process oracle.  This validates the
distributed algorithm - ownership, broadcast / reduce schedule, augmented-row forward solve,
right-looking prediction pass - exactly as it runs over RCCL on the GPUs."""

import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class NumpyBackend:
    """Same interface as battgp_amd.sharded.DeviceBackend, numpy arithmetic (tests only)."""

    def __init__(self, kernel_id, hyp):
        from oracle import kernels as K

        self.K, self.kid, self.hyp = K, kernel_id, np.asarray(hyp, dtype=np.float64)

    def zeros(self, n):
        return np.zeros(int(n))

    def empty(self, n):
        return np.full(int(n), np.nan)

    def upload(self, a):
        return np.ascontiguousarray(a, dtype=np.float64)

    def to_host(self, t):
        return np.array(t)

    def sync(self):
        pass

    def after_comm(self):
        pass

    @staticmethod
    def _cm(buf, off, rows, cols, ld):
        """column-major view [rows, cols] at element offset `off`"""
        return np.lib.stride_tricks.as_strided(buf[off:], (rows, cols), (8, 8 * ld))

    def fill_panel(self, store, ld, lcol0, x, n, d, col0, ncols, rows, y, extra_diag):
        aug = 64
        nr = rows - aug
        blk = np.zeros((nr, ncols))
        idx_r = np.arange(col0, col0 + nr)
        idx_c = np.arange(col0, col0 + ncols)
        vr, vc = idx_r < n, idx_c < n
        if vr.any() and vc.any():
            blk[np.ix_(vr, vc)] = self.K.kernel_matrix(self.kid, self.hyp, x[idx_r[vr]], x[idx_c[vc]])
        for c in range(ncols):
            j = col0 + c
            blk[c, c] = blk[c, c] + self.hyp[0] + extra_diag if j < n else 1.0
        self._cm(store, lcol0 * ld + col0, nr, ncols, ld)[:] = blk
        a = np.zeros((aug, ncols))
        a[0, vc] = y[idx_c[vc]]
        self._cm(store, lcol0 * ld + (ld - aug), aug, ncols, ld)[:] = a

    def factor_panel(self, store, ld, lcol0, col0, rows, nbk, inv):
        p = self._cm(store, lcol0 * ld + col0, rows, nbk, ld)
        try:
            l11 = np.linalg.cholesky(np.tril(p[:nbk]) + np.tril(p[:nbk], -1).T)
        except np.linalg.LinAlgError:
            return 1
        p[:nbk] = l11
        p[nbk:] = np.linalg.solve(l11, p[nbk:].T).T
        return 0

    def factor_pack(self, store, ld, lcol0, col0, rows, nbk, inv, pbuf):
        info = self.factor_panel(store, ld, lcol0, col0, rows, nbk, inv)
        if info == 0:
            self.pack_panel(store, ld, lcol0, col0, rows, nbk, pbuf)
        return info

    def pack_panel(self, store, ld, lcol0, col0, rows, nbk, pbuf):
        pbuf[: nbk * rows] = self._cm(store, lcol0 * ld + col0, rows, nbk, ld).T.reshape(-1)

    def update_panel(self, store, ld, lcol0, colj, rows_j, nbj, pbuf, ldp, off, nbk):
        pm = self._cm(pbuf, off, rows_j, nbk, ldp)
        self._cm(store, lcol0 * ld + colj, rows_j, nbj, ld)[:] -= pm @ pm[:nbj].T

    def update_panels(self, store, ld, items, pbuf, ldp, nbk):
        for lc, cj, rows_j, nbj, off in items:
            self.update_panel(store, ld, lc, cj, rows_j, nbj, pbuf, ldp, off, nbk)

    def diag_logsum(self, store, ld, lcol0, col0, nbk):
        return float(np.sum(np.log(np.diag(self._cm(store, lcol0 * ld + col0, nbk, nbk, ld)))))

    def aug_row(self, store, ld, lcol0, nbk):
        return self._cm(store, lcol0 * ld + (ld - 64), 1, nbk, ld)[0].copy()

    def cross_fill(self, xq, m, mpad, x, n, d, npad, out, lde):
        self._cm(out, 0, m, n, lde)[:] = self.K.kernel_matrix(self.kid, self.hyp, xq, x)

    def solve_panel(self, e, lde, ecol0, me, store, ld, lcol0, col0, nbk, inv):
        l11 = np.tril(self._cm(store, lcol0 * ld + col0, nbk, nbk, ld))
        ek = self._cm(e, ecol0 * lde, me, nbk, lde)
        ek[:] = np.linalg.solve(l11, ek.T).T

    def update_rows(self, w, lde, wcol0, me, ek, ldek, store, ld, lcol0, row0, nrows, nbk):
        l21 = self._cm(store, lcol0 * ld + row0, nrows, nbk, ld)
        self._cm(w, wcol0 * lde, me, nrows, lde)[:] -= self._cm(ek, 0, me, nbk, ldek) @ l21.T

    def sumsq(self, v, n):
        return float(np.dot(v[:n], v[:n]))

    def var_finish(self, xq, m, d, ssq, min_var):
        var = self.K.kernel_diag(self.kid, self.hyp, xq) - ssq[:m]
        return np.maximum(var, min_var) if min_var >= 0 else var

    def rowdot(self, e, lde, m, n, vec, out):
        em = self._cm(e, 0, m, n, lde)
        out[:m] = em @ vec if vec is not None else np.einsum("ij,ij->i", em, em)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, nb, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from battgp_amd import parallel, synthetic
    from battgp_amd.sharded import ShardedExactGP
    from test_sharded_cpu import NumpyBackend

    dist = parallel.init("gloo") if world > 1 else None
    x, y = synthetic.make_cell_data(n, seed=42)
    xq = synthetic.make_query(x, 21)
    gp = ShardedExactGP(NumpyBackend(0, synthetic.HYP_BATTGP), dist, rank, world, nb=nb)
    lml = gp.fit(x, y)
    mean, var = gp.predict(xq)
    q.put((rank, lml, mean.tolist(), var.tolist()))
    if dist is not None:
        parallel.barrier(dist)
        dist.destroy_process_group()


def _run(world, n, nb):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, nb, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=170) for _ in range(world))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    return res


@pytest.mark.timeout(200)
@pytest.mark.parametrize("world,n,nb", [(2, 330, 64), (2, 500, 128), (3, 449, 64)])
def test_sharded_gp_matches_oracle(world, n, nb):
    from battgp_amd import synthetic
    from oracle import kernels as K
    from oracle.exact_gp import OracleGP

    res = _run(world, n, nb)
    x, y = synthetic.make_cell_data(n, seed=42)
    xq = synthetic.make_query(x, 21)
    ref = OracleGP(K.KERNEL_BATTGP, synthetic.HYP_BATTGP, x, y).fit()
    m_ref, v_ref = ref.predict(xq)
    for rank, lml, mean, var in res:
        assert abs(lml - ref.lml) < 1e-9 * abs(ref.lml), (rank, lml, ref.lml)
        assert np.linalg.norm(np.array(mean) - m_ref) < 1e-8 * np.linalg.norm(m_ref)
        assert np.max(np.abs(np.array(var) - v_ref)) < 1e-9 * synthetic.OUTPUTSCALE_RBF
    assert res[0][1:] == res[1][1:]  # every rank ends with identical results


def test_panel_layout():
    from battgp_amd.sharded import PanelLayout

    lay = PanelLayout(1000, 128, 4)
    assert lay.npad == 1024 and lay.npanels == 8 and lay.nrows == 1088
    assert [lay.owner(j) for j in range(8)] == [0, 1, 2, 3, 0, 1, 2, 3]
    assert lay.local_panels(1) == [1, 5] and lay.local_index(5) == 1
    assert lay.rows_from(3) == 1088 - 384
    lay = PanelLayout(200, 128, 2)
    assert lay.npad == 256 and lay.width(1) == 128 and lay.npanels == 2
    lay = PanelLayout(330, 128, 2)
    assert lay.npad == 384 and lay.npanels == 3 and lay.width(2) == 128
