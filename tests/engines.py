"""Test-only row-block engine over the CPU oracle, with the interface of
active_matrix_factorization_amd.sharding.HipEngine.  It lets the world_size-2 gloo tests
exercise the sharding / exchange logic on a machine without a GPU.  Never used by the product."""
import numpy as np

from oracle import oracle as orc


class OracleEngine(object):
    def __init__(self, ratings, n, m, d, sigma_sq=1., sigma_u_sq=10., sigma_v_sq=10.,
                 subtract_mean=False, mean_rating=0.):
        self.ratings, self.n, self.m, self.d = np.asarray(ratings, float), n, m, d
        self.kw = dict(sig=sigma_sq, sig_u=sigma_u_sq, sig_v=sigma_v_sq, subtract_mean=subtract_mean,
                       mean_rating=mean_rating)

    # arrays are plain NumPy here (the product's HipEngine keeps DeviceArrays)
    def put(self, host, rows=None):
        host = np.array(host, np.float64)
        if rows is not None and rows > host.shape[0]:
            host = np.vstack([host, np.zeros((rows - host.shape[0],) + host.shape[1:])])
        return host

    def get(self, arr, n):
        return np.array(arr[:n])

    def grad_rows(self, side, lo, hi, U, V, slot=0):
        gU, gV = orc.pmf_gradient(self.ratings, U[:self.n], V[:self.m], **self.kw)
        return gU if side == 0 else gV

    def trial_block(self, X, G, lo, hi, lr, bs):
        out = np.zeros((bs, self.d))
        out[:hi - lo] = X[lo:hi] + lr * G[lo:hi]
        return out

    def ll_terms(self, U, V, u_lo, u_hi, v_lo, v_hi):
        r = self.ratings
        U, V = U[:self.n], V[:self.m]
        sel = r[(r[:, 0] >= u_lo) & (r[:, 0] < u_hi)]
        sq = orc.pmf_sq_error(sel, U, V, self.kw['subtract_mean'], self.kw['mean_rating']) if len(sel) else 0.
        return np.array([sq, (U[u_lo:u_hi] ** 2).sum(), (V[v_lo:v_hi] ** 2).sum()])

    def ll_partial(self, U, V, u_lo, u_hi, v_lo, v_hi):
        return np.concatenate([self.ll_terms(U, V, u_lo, u_hi, v_lo, v_hi), [0.]])

    def small_to_host(self, arr):
        return np.asarray(arr)

    def as_rows(self, gathered):
        return gathered.reshape((gathered.shape[0] * gathered.shape[1],) + gathered.shape[2:])

    def hold_grid(self, rows_2d, dtype=np.float64):
        rows_2d = np.asarray(rows_2d)
        return None if rows_2d.shape[0] == 0 else rows_2d.astype(dtype)

    def hold(self, host):
        return np.array(host, np.float64)

    def score_resident(self, cross, mask, mean_u_rows, mean_items, diag_u_rows, diag_items, omega, m, want):
        rec = np.zeros(4)
        idx = rec[2:3].view(np.int64)
        idx[0] = -1
        if cross is not None:
            tot, bv, bi = self.score_rows(mean_u_rows, mean_items[:m], cross, diag_u_rows, diag_items[:m], omega,
                                          mask, want)
            rec[0], rec[1], idx[0] = tot, bv, bi
        return rec

    def gibbs_rows(self, side, lo, hi, oth, alpha, mu, beta, mean_sub, z, bs):
        rows = self.n if side == 0 else self.m
        ptr, idx, rat = orc.csr_by_row(self.ratings, side, rows)
        oth = np.ascontiguousarray(oth[:(self.m if side == 0 else self.n)])
        out = np.zeros((bs, self.d))
        for t in range(lo, hi):
            out[t - lo] = orc.sample_feature(oth, idx[ptr[t]:ptr[t + 1]], rat[ptr[t]:ptr[t + 1]], alpha,
                                             mu, beta, mean_sub, z[t])
        return out

    def score_rows(self, mean_users, mean_items, cross_rows, diag_users, diag_items, omega, mask_rows,
                   want):
        nl, m = cross_rows.shape
        if nl == 0:
            return 0., 0., -1
        _, pv = orc.mn_pred_means_vars_compact(nl, m, np.vstack([mean_users, mean_items]), cross_rows,
                                               np.concatenate([diag_users, diag_items]), omega)
        mask = np.ones((nl, m), bool) if mask_rows is None else mask_rows.astype(bool)
        if not mask.any():
            return 0., 0., -1
        cand = np.where(mask, pv, -np.inf if want > 0 else np.inf)
        k = int(np.argmax(cand)) if want > 0 else int(np.argmin(cand))
        return float(pv[mask].sum()), float(cand.ravel()[k]), k
