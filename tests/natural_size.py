"""Checks of one fit + predict at a BASELINE config's NATURAL size, where no CPU factorisation is affordable: every
covariance entry they use is evaluated by the ORACLE's kernel code on the host (oracle/kernels.py) - never by the device
`kfun`, which the engine's own `bgp_residuals` shares with the fill.  Test infrastructure, shared by
tests/test_gpu_pins_and_sizes.py (configs 2 and 3) and tests/test_gpu_y_config_sizes.py (configs 4 and 5)."""

import numpy as np
import torch

from battgp_amd import synthetic
from battgp_amd.engine import ExactGPEngine
from oracle import kernels as K

REL = 1e-6  # north_star tolerance (LML, posterior mean)


def oracle_sigma_entries(kid, hyp, x, ii, jj):
    """Sigma_ij for index pairs, evaluated by the oracle's kernel code row by row (no N x N matrix)."""
    out = np.empty(len(ii))
    for q, (i, j) in enumerate(zip(ii, jj)):
        out[q] = K.kernel_matrix(kid, hyp, x[i : i + 1], x[j : j + 1])[0, 0] + (K.noise(hyp) if i == j else 0.0)
    return out


def natural_size_checks(kid, hyp, n, m=300, nblocks=24, blk=64, npairs=48, nrows_solve=32, seed=None, slab=0, nb=-1):
    """`slab` / `nb`: bgp_set_layout / outer panel width for the toy-size rehearsals (0 / -1 = the automatic choices, which
    is what the natural sizes run).  Returns (figures + the live engine, (x, y, xq, mean, var)); the caller closes the engine."""
    rng = np.random.default_rng(n + kid)
    x, y = synthetic.make_cell_data(n, seed=seed)
    xq = synthetic.make_query(x, m)
    e = ExactGPEngine(kid, hyp)
    if slab:
        e.set_layout(slab)
    if nb > 0:
        e.set_options(nb_outer=nb)
    out = {}
    try:
        # (1) sampled blocks of the matrix the fit factors (~1e5 entries) against the oracle's kernel code
        tx = torch.from_numpy(x).cuda()
        buf = torch.empty((blk, blk), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        worst = 0.0
        for b in range(nblocks):
            r0 = int(rng.integers(0, n - blk)) & ~1
            c0 = r0 if b % 6 == 0 else int(rng.integers(0, n - blk)) & ~1  # every 6th block sits on the diagonal
            e.fill_block_device(tx.data_ptr(), n, 4, r0, c0, blk, blk, buf.data_ptr(), blk)
            e.sync()
            got = buf.cpu().numpy().T  # column-major [blk, blk]
            want = K.kernel_matrix(kid, hyp, x[r0 : r0 + blk], x[c0 : c0 + blk])
            if r0 == c0:
                want[np.diag_indices(blk)] += K.noise(hyp)
            err = np.max(np.abs(got - want) / (np.abs(want) + 1e-300 + 1e-16 * np.abs(want).max()))
            worst = max(worst, float(err))
        out["fill_rel"] = worst
        assert worst <= 2e-13, worst
        del buf, tx
        torch.cuda.empty_cache()

        # (2) the factorisation at the natural size (automatic scheme / NB / layout)
        lml, mean, var = e.fit_predict(x, y, xq, min_var=-1.0)
        assert e.jitter == 0.0
        out["lml"] = lml
        alpha = e.alpha()
        diag = e.factor_diag()

        # (3) sampled (L L^T)_ij against oracle-evaluated Sigma_ij
        ii = rng.integers(0, n, npairs)
        jj = np.array([rng.integers(0, i + 1) for i in ii])
        ii[0], jj[0] = n - 1, n - 1
        ii[1], jj[1] = n - 1, 0
        ii[2], jj[2] = n // 2, n // 2
        rows = np.unique(np.concatenate((ii, jj)))
        lrows = e.factor_rows(rows)
        pos = {int(r): k for k, r in enumerate(rows)}
        llt = np.array([lrows[pos[int(i)]] @ lrows[pos[int(j)]] for i, j in zip(ii, jj)])
        sig = oracle_sigma_entries(kid, hyp, x, ii, jj)
        scale = np.sqrt(oracle_sigma_entries(kid, hyp, x, ii, ii) * oracle_sigma_entries(kid, hyp, x, jj, jj))
        out["llt"] = float(np.max(np.abs(llt - sig) / scale))
        assert out["llt"] <= 1e-12, out["llt"]
        assert np.allclose(diag[rows], lrows[np.arange(len(rows)), rows], rtol=0, atol=0)  # same numbers, two getters

        # (4) sampled rows of Sigma alpha = y with oracle-evaluated rows of Sigma
        rs = rng.integers(0, n, nrows_solve)
        rs[0], rs[1] = 0, n - 1
        res = np.empty(nrows_solve)
        for q, i in enumerate(rs):
            krow = K.kernel_matrix(kid, hyp, x[i : i + 1], x)[0]
            krow[i] += K.noise(hyp)
            res[q] = krow @ alpha - y[i]
        out["solve_rows"] = float(np.max(np.abs(res)) / np.max(np.abs(y)))
        assert out["solve_rows"] <= 1e-7, out["solve_rows"]  # cond(Sigma) ~ 1e8 at these sizes

        # (5) LML and posterior mean re-derived on the host from alpha, diag(L) and oracle kernels
        lml_host = -0.5 * float(y @ alpha) - float(np.sum(np.log(diag))) - 0.5 * n * np.log(2.0 * np.pi)
        assert abs(lml - lml_host) <= REL * abs(lml_host), (lml, lml_host)
        mean_host = K.kernel_matrix(kid, hyp, xq, x) @ alpha
        out["mean_rel"] = float(np.linalg.norm(mean - mean_host) / np.linalg.norm(mean_host))
        assert out["mean_rel"] <= REL, out["mean_rel"]
        prior = K.kernel_diag(kid, hyp, xq)
        assert np.all(var <= prior * (1 + 1e-12)) and np.all(var > -1e-9 * prior.max())
        out["engine"] = e
        return out, (x, y, xq, mean, var)
    except BaseException:
        e.close()
        raise
