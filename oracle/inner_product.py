"""Restatement of /root/reference/src/proofs/inner_product.rs (test infrastructure).

`InnerProductArg::{prove, verify, fast_verify}` and `inner_product`, following the file line by
line; group operations go through oracle.ec (one full scalar multiplication per term, as the
reference does).  Errors: returns "Ok" / raises nothing; `InnerProductError` is the string
"InnerProductError" (src/lib.rs:30-35).
"""
from . import ec
from .curv import sha256
from .ec import Q

OK = "Ok"
INNER_PRODUCT_ERROR = "InnerProductError"


def inner_product(a, b):
    """inner_product.rs:348-361."""
    assert len(a) == len(b), "inner_product(a,b): lengths of vectors do not match"
    out = 0
    for x, y in zip(a, b):
        out = (out + x * y % Q) % Q
    return out


def _fold_skip_zero(acc, points, scalars):
    """The `if x.1 != &BigInt::zero()` fold of inner_product.rs:73-90."""
    pts, ks = [], []
    for pt, k in zip(points, scalars):
        if k != 0:
            pts.append(pt)
            ks.append(k % Q)  # Scalar::from(&BigInt) reduces mod q
    return ec.msm_naive(pts, ks, acc)


class InnerProductArg:
    """inner_product.rs:27-33."""

    def __init__(self, L, R, a_tag, b_tag):
        self.L = list(L)
        self.R = list(R)
        self.a_tag = a_tag
        self.b_tag = b_tag

    # inner_product.rs:36-169 (recursion unrolled into a loop; same order of operations)
    @staticmethod
    def prove(G, H, ux, P, a, b, L_vec=None, R_vec=None, trace=None):
        L_vec = [] if L_vec is None else L_vec
        R_vec = [] if R_vec is None else R_vec
        G, H, a, b = list(G), list(H), list(a), list(b)
        while True:
            n = len(G)
            assert len(H) == n and len(a) == n and len(b) == n
            assert n & (n - 1) == 0 and n > 0
            if n == 1:
                return InnerProductArg(L_vec, R_vec, a[0], b[0])
            n //= 2
            a_L, a_R = a[:n], a[n:]
            b_L, b_R = b[:n], b[n:]
            G_L, G_R = G[:n], G[n:]
            H_L, H_R = H[:n], H[n:]

            c_L = inner_product(a_L, b_R)
            c_R = inner_product(a_R, b_L)

            # L = <a_L * G_R> + <b_R * H_L> + c_L * ux       (:70-90)
            ux_CL = ec.mul(ux, c_L)
            aL_GR = _fold_skip_zero(ux_CL, G_R, a_L)
            L = _fold_skip_zero(aL_GR, H_L, b_R)
            # R = <a_R * G_L> + <b_L * H_R> + c_R * ux       (:95-115)
            ux_CR = ec.mul(ux, c_R)
            aR_GL = _fold_skip_zero(ux_CR, G_L, a_R)
            R = _fold_skip_zero(aR_GL, H_R, b_L)

            x = sha256().chain_points([L, R, ux]).result_scalar()   # :117
            x_inv = pow(x, -1, Q)                                     # :120 (.unwrap(): x != 0)
            if trace is not None:
                trace.append({"L": L, "R": R, "x": x})

            a = [(a_L[i] * x % Q + a_R[i] * x_inv % Q) % Q for i in range(n)]   # :122-128
            b = [(b_R[i] * x % Q + b_L[i] * x_inv % Q) % Q for i in range(n)]   # :131-137
            # G_new[i] = x^-1 G_L[i] + x G_R[i]; H_new[i] = x H_L[i] + x^-1 H_R[i]   (:140-155)
            gl = ec.mul_batch(G_L, [x_inv] * n)
            gr = ec.mul_batch(G_R, [x] * n)
            G = [ec.add(gr[i], gl[i]) for i in range(n)]
            hl = ec.mul_batch(H_L, [x] * n)
            hr = ec.mul_batch(H_R, [x_inv] * n)
            H = [ec.add(hl[i], hr[i]) for i in range(n)]
            L_vec.append(L)
            R_vec.append(R)

    # inner_product.rs:171-244
    def verify(self, g_vec, hi_tag, ux, P):
        G, H = list(g_vec), list(hi_tag)
        Ls, Rs = list(self.L), list(self.R)
        while True:
            n = len(G)
            assert len(H) == n and n & (n - 1) == 0 and n > 0
            if n == 1:
                break
            n //= 2
            G_L, G_R, H_L, H_R = G[:n], G[n:], H[:n], H[n:]
            x = sha256().chain_points([Ls[0], Rs[0], ux]).result_scalar()
            x_inv = pow(x, -1, Q)
            x_sq = x * x % Q
            x_inv_sq = x_inv * x_inv % Q
            gl = ec.mul_batch(G_L, [x_inv] * n)
            gr = ec.mul_batch(G_R, [x] * n)
            G = [ec.add(gr[i], gl[i]) for i in range(n)]
            hl = ec.mul_batch(H_L, [x] * n)
            hr = ec.mul_batch(H_R, [x_inv] * n)
            H = [ec.add(hl[i], hr[i]) for i in range(n)]
            Lx_sq = ec.mul(Ls[0], x_sq)
            Rx_sq_inv = ec.mul(Rs[0], x_inv_sq)
            P = ec.add(ec.add(Lx_sq, Rx_sq_inv), P)          # :222
            Ls, Rs = Ls[1:], Rs[1:]
        a_fe = self.a_tag % Q
        b_fe = self.b_tag % Q
        c = a_fe * b_fe % Q
        P_calc = ec.add(ec.add(ec.mul(G[0], a_fe), ec.mul(H[0], b_fe)), ec.mul(ux, c))
        return OK if P == P_calc else INNER_PRODUCT_ERROR

    # inner_product.rs:253-345
    def fast_verify_terms(self, g_vec, hi_tag, ux, P):
        """Returns (init_point, points, scalars) of the single MSM at :334-338."""
        n = len(g_vec)
        assert len(hi_tag) == n and n & (n - 1) == 0 and n > 0
        lg_n = len(self.L)
        assert lg_n <= 64
        x_sq_vec, minus_x_sq_vec, minus_x_inv_sq_vec = [], [], []
        allinv = 1
        for Li, Ri in zip(self.L, self.R):
            x = sha256().chain_points([Li, Ri, ux]).result_scalar()
            x_inv = pow(x, -1, Q)
            x_sq = x * x % Q
            x_inv_sq = x_inv * x_inv % Q
            x_sq_vec.append(x_sq)
            minus_x_sq_vec.append((0 - x_sq) % Q)
            minus_x_inv_sq_vec.append((0 - x_inv_sq) % Q)
            allinv *= x_inv                       # unreduced in the reference (:293)
        s = [allinv]
        for i in range(1, n):
            lg_i = i.bit_length() - 1
            k = 1 << lg_i
            s.append(s[i - k] * x_sq_vec[(lg_n - 1) - lg_i])   # unreduced (:305)
        a_times_s = [s[i] * self.a_tag % Q for i in range(n)]
        b_div_s = [pow(s[i], -1, Q) * self.b_tag % Q for i in range(n)]
        scalars = a_times_s + b_div_s + minus_x_sq_vec + minus_x_inv_sq_vec
        points = list(g_vec) + list(hi_tag) + list(self.L) + list(self.R)
        c = self.a_tag * self.b_tag % Q
        ux_c = ec.mul(ux, c)
        return ux_c, points, scalars

    def fast_verify(self, g_vec, hi_tag, ux, P):
        ux_c, points, scalars = self.fast_verify_terms(g_vec, hi_tag, ux, P)
        expect_P = ec.msm_naive(points, scalars, ux_c)
        return OK if P == expect_P else INNER_PRODUCT_ERROR
