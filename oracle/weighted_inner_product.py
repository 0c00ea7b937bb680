"""Restatement of /root/reference/src/proofs/weighted_inner_product.rs (test infrastructure).

`WeightedInnerProdArg::{prove, verify, fast_verify}` and `weighted_inner_product`.
Randomness draw order (tape): per round d_L then d_R (:106, :134); final r, s, delta, eta (:213-219).
"""
from . import ec
from .curv import sha256
from .ec import Q
from .inner_product import OK, _fold_skip_zero

WEIGHTED_INNER_PROD_ERROR = "WeightedInnerProdError"


def weighted_inner_product(a, b, y):
    """weighted_inner_product.rs:518-536: sum a_i b_i y^(i+1) mod q."""
    assert len(a) == len(b), "weighted_inner_product(a,b): lengths of vectors do not match"
    out = 0
    yp = y
    for i in range(len(a)):
        out = (out + (a[i] * b[i] % Q) * yp % Q) % Q
        yp = yp * y            # the reference's y_powers are unreduced BigInts (:525-527)
    return out


class WeightedInnerProdArg:
    """weighted_inner_product.rs:35-44."""

    def __init__(self, L, R, a_tag, b_tag, r_prime, s_prime, delta_prime):
        self.L, self.R = list(L), list(R)
        self.a_tag, self.b_tag = a_tag, b_tag
        self.r_prime, self.s_prime, self.delta_prime = r_prime, s_prime, delta_prime

    # weighted_inner_product.rs:47-268
    @staticmethod
    def prove(G, H, g, h, P, a, b, alpha, y, tape, L_vec=None, R_vec=None, trace=None):
        L_vec = [] if L_vec is None else L_vec
        R_vec = [] if R_vec is None else R_vec
        G, H, a, b = list(G), list(H), list(a), list(b)
        order = Q
        y_inv = pow(y, -1, order)
        while True:
            n = len(G)
            assert len(H) == n and len(a) == n and len(b) == n
            assert n & (n - 1) == 0 and n > 0
            if n == 1:
                break
            n //= 2
            # powers_y[n-1] = y^n, powers_yinv[n-1] = y^-n   (:71-76, :91, :94)
            y_n = pow(y, n, order)
            yinv_n = pow(y_inv, n, order)
            a_L, a_R = a[:n], a[n:]
            b_L, b_R = b[:n], b[n:]
            G_L, G_R = G[:n], G[n:]
            H_L, H_R = H[:n], H[n:]
            yn_aR = [y_n * a_R[i] % order for i in range(n)]
            yninv_aL = [yinv_n * a_L[i] % order for i in range(n)]
            c_L = weighted_inner_product(a_L, b_R, y)
            c_R = weighted_inner_product(yn_aR, b_L, y)

            # L = <yninv_aL * G_R> + <b_R * H_L> + c_L g + d_L h      (:103-126)
            g_cL = ec.mul(g, c_L)
            d_L = tape.random_scalar()
            g_cL_h_dL = ec.add(g_cL, ec.mul(h, d_L))
            L = _fold_skip_zero(_fold_skip_zero(g_cL_h_dL, G_R, yninv_aL), H_L, b_R)
            # R = <yn_aR * G_L> + <b_L * H_R> + c_R g + d_R h         (:131-154)
            g_cR = ec.mul(g, c_R)
            d_R = tape.random_scalar()
            g_cR_h_dR = ec.add(g_cR, ec.mul(h, d_R))
            R = _fold_skip_zero(_fold_skip_zero(g_cR_h_dR, G_L, yn_aR), H_R, b_L)

            x = sha256().chain_points([L, R, g, h]).result_scalar()   # :158
            x_sq = x * x % order
            x_sq_inv = pow(x_sq, -1, order)
            x_inv = pow(x, -1, order)
            if trace is not None:
                trace.append({"L": L, "R": R, "x": x})
            a = [(a_L[i] * x % order + yn_aR[i] * x_inv % order) % order for i in range(n)]
            b = [(b_R[i] * x % order + b_L[i] * x_inv % order) % order for i in range(n)]
            alpha = (alpha + (x_sq * d_L % order + x_sq_inv * d_R % order) % order) % order
            x_yinv = x * yinv_n % order                              # :187
            gl = ec.mul_batch(G_L, [x_inv] * n)
            gr = ec.mul_batch(G_R, [x_yinv] * n)
            G = [ec.add(gr[i], gl[i]) for i in range(n)]
            hl = ec.mul_batch(H_L, [x] * n)
            hr = ec.mul_batch(H_R, [x_inv] * n)
            H = [ec.add(hl[i], hr[i]) for i in range(n)]
            L_vec.append(L)
            R_vec.append(R)

        r = tape.random_scalar()
        s = tape.random_scalar()
        delta = tape.random_scalar()
        eta = tape.random_scalar()
        a_sy_b_ry = ((a[0] * s % order) * y % order + (b[0] * r % order) * y % order) % order
        A = ec.add(ec.add(ec.add(ec.mul(G[0], r), ec.mul(H[0], s)), ec.mul(g, a_sy_b_ry)), ec.mul(h, delta))
        r_sy = y * (r * s % order) % order
        B = ec.add(ec.mul(g, r_sy), ec.mul(h, eta))
        e = sha256().chain_points([A, B, g, h]).result_scalar()      # :243
        e_sq = e * e % order
        r_prime = (r + a[0] * e % order) % order
        s_prime = (s + b[0] * e % order) % order
        delta_prime = (eta + (delta * e % order + alpha * e_sq % order) % order) % order
        return WeightedInnerProdArg(L_vec, R_vec, A, B, r_prime, s_prime, delta_prime)

    # weighted_inner_product.rs:270-377
    def verify(self, g_vec, hi_tag, g, h, P, y):
        G, H = list(g_vec), list(hi_tag)
        order = Q
        y_inv = pow(y, -1, order)
        Ls, Rs = list(self.L), list(self.R)
        while True:
            n = len(G)
            assert len(H) == n and n & (n - 1) == 0 and n > 0
            if n == 1:
                break
            n //= 2
            G_L, G_R, H_L, H_R = G[:n], G[n:], H[:n], H[n:]
            x = sha256().chain_points([Ls[0], Rs[0], g, h]).result_scalar()
            x_inv = pow(x, -1, order)
            x_sq = x * x % order
            x_inv_sq = x_inv * x_inv % order
            x_yinv = x * pow(y_inv, n, order) % order
            gl = ec.mul_batch(G_L, [x_inv] * n)
            gr = ec.mul_batch(G_R, [x_yinv] * n)
            G = [ec.add(gr[i], gl[i]) for i in range(n)]
            hl = ec.mul_batch(H_L, [x] * n)
            hr = ec.mul_batch(H_R, [x_inv] * n)
            H = [ec.add(hl[i], hr[i]) for i in range(n)]
            P = ec.add(ec.add(ec.mul(Ls[0], x_sq), ec.mul(Rs[0], x_inv_sq)), P)
            Ls, Rs = Ls[1:], Rs[1:]
        e = sha256().chain_points([self.a_tag, self.b_tag, g, h]).result_scalar()
        e_sq = e * e % order
        left = ec.add(ec.add(ec.mul(P, e_sq), ec.mul(self.a_tag, e)), self.b_tag)
        er_prime = e * self.r_prime % order
        es_prime = e * self.s_prime % order
        yrs_prime = (self.s_prime * self.r_prime % order) * y % order
        right = ec.add(ec.add(ec.add(ec.mul(G[0], er_prime), ec.mul(H[0], es_prime)),
                              ec.mul(g, yrs_prime)), ec.mul(h, self.delta_prime))
        return OK if left == right else WEIGHTED_INNER_PROD_ERROR

    # weighted_inner_product.rs:386-515
    def fast_verify_terms(self, g_vec, hi_tag, g, h, P, y):
        """Returns (init = h*delta', points, scalars, rhs) for the check at :500-510."""
        n = len(g_vec)
        order = Q
        assert len(hi_tag) == n and n & (n - 1) == 0 and n > 0
        y_inv = pow(y, -1, order)
        powers_yinv = [pow(y_inv, i + 1, order) for i in range(n)]
        lg_n = len(self.L)
        assert lg_n <= 64
        e = sha256().chain_points([self.a_tag, self.b_tag, g, h]).result_scalar()
        e_sq = e * e % order
        x_sq_vec, m_esq_xsq, m_esq_xinvsq = [], [], []
        allinv, all_ = 1, 1
        for Li, Ri in zip(self.L, self.R):
            x = sha256().chain_points([Li, Ri, g, h]).result_scalar()
            x_inv = pow(x, -1, order)
            x_sq = x * x % order
            x_inv_sq = x_inv * x_inv % order
            x_sq_vec.append(x_sq)
            m_esq_xsq.append((0 - e_sq * x_sq % order) % order)
            m_esq_xinvsq.append((0 - e_sq * x_inv_sq % order) % order)
            allinv = allinv * x_inv % order
            all_ = all_ * x % order
        s, sg, sh = [allinv], [allinv], [all_]
        for i in range(1, n):
            lg_i = i.bit_length() - 1
            k = 1 << lg_i
            s.append(s[i - k] * x_sq_vec[(lg_n - 1) - lg_i] % order)
            sg.append(s[i] * powers_yinv[i - 1] % order)
            sh.append(pow(s[i], -1, order))
        er_times_sg = [self.r_prime * (e * sg[i] % order) % order for i in range(n)]
        es_times_sg = [self.s_prime * (e * sh[i] % order) % order for i in range(n)]
        r_times_s_y = (self.r_prime * self.s_prime % order) * y % order
        scalars = er_times_sg + es_times_sg + m_esq_xsq + m_esq_xinvsq + [r_times_s_y]
        points = list(g_vec) + list(hi_tag) + list(self.L) + list(self.R) + [g]
        h_delta_prime = ec.mul(h, self.delta_prime)
        rhs = ec.add(ec.add(ec.mul(P, e_sq), ec.mul(self.a_tag, e)), self.b_tag)
        return h_delta_prime, points, scalars, rhs

    def fast_verify(self, g_vec, hi_tag, g, h, P, y):
        init, points, scalars, rhs = self.fast_verify_terms(g_vec, hi_tag, g, h, P, y)
        lhs = ec.msm_naive(points, scalars, init)
        return OK if lhs == rhs else WEIGHTED_INNER_PROD_ERROR
