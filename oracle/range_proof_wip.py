"""Restatement of /root/reference/src/proofs/range_proof_wip.rs (test infrastructure).

`StatementRP::generate_bases`, `RangeProofWIP::{prove, verify, aggregated_verify}` (Bulletproofs+).
Tape draw order: alpha (:150), then the WIP argument's draws.
"""
from . import ec
from .curv import sha256, bpplus_generators
from .ec import Q
from .inner_product import OK
from .range_proof import RANGE_PROOF_ERROR
from .weighted_inner_product import WeightedInnerProdArg


class StatementRP:
    """range_proof_wip.rs:38-45."""

    def __init__(self, g_vec, h_vec, G, H, bit_length):
        self.g_vec, self.h_vec, self.G, self.H, self.bit_length = list(g_vec), list(h_vec), G, H, bit_length

    @staticmethod
    def generate_bases(init_seed, num_of_proofs, bit_length):
        """range_proof_wip.rs:48-91."""
        g_vec, h_vec, G, H = bpplus_generators(bit_length, num_of_proofs, init_seed)
        return StatementRP(g_vec, h_vec, G, H, bit_length)


def _common_scalars(n, m, y_bn, z_bn):
    """y_powers, y_powers_rev, d, y_pow (= y^(nm+1)), sum term used as scalar_G / zeta
    (range_proof_wip.rs:175-224, 298-336, 428-455, 521-531)."""
    order = Q
    nm = n * m
    z_sq = z_bn * z_bn % order
    y_powers = [pow(y_bn, i + 1, order) for i in range(nm)]
    y_powers_rev = y_powers[::-1]
    vec_2n = [pow(2, i, order) for i in range(n)]
    vec_z2m = [pow(z_sq, i + 1, order) for i in range(m)]      # unreduced in the reference
    d = [vec_2n[i % n] * vec_z2m[i // n] % order for i in range(nm)]
    y_pow = pow(y_bn, nm + 1, order)
    z_minus_z2 = (z_bn - z_sq) % order
    z_y_pow = z_bn * y_pow % order
    zeta = 0
    for i in range(nm):
        zeta = (zeta + (z_minus_z2 * y_powers[i] % order - d[i] * z_y_pow % order) % order) % order
    return y_powers, y_powers_rev, vec_z2m, d, y_pow, zeta


class RangeProofWIP:
    """range_proof_wip.rs:94-98."""

    def __init__(self, A, weighted_inner_product_proof):
        self.A = A
        self.weighted_inner_product_proof = weighted_inner_product_proof

    # range_proof_wip.rs:101-274
    @staticmethod
    def prove(stmt, secret, blinding, tape, trace=None):
        secret = list(secret)
        num_of_proofs = len(secret)
        bit_length = stmt.bit_length
        nm = num_of_proofs * bit_length
        g_vec, h_vec, G, H = stmt.g_vec, stmt.h_vec, stmt.G, stmt.H
        N = len(g_vec)
        order = Q
        assert len(h_vec) == N
        assert N & (N - 1) == 0 and N > 0

        secret.reverse()
        secret_agg = 0
        for x in secret:
            secret_agg = (secret_agg << bit_length) + x
        aL = [(secret_agg >> i) % 2 for i in range(nm)]
        aR = [(aL[i] - 1) % order for i in range(nm)]
        secret_bits = [(aL[i] & 1) == 1 for i in range(nm)]

        alpha = tape.random_scalar()          # :150
        A = ec.mul(H, alpha)
        for gi, bit in zip(g_vec, secret_bits):
            if bit:
                A = ec.add(A, gi)
        for hi, bit in zip(h_vec, secret_bits):
            if not bit:
                A = ec.sub(A, hi)

        y_bn = sha256().chain_points([A]).result_scalar()                 # :167
        yG = ec.mul(ec.G, y_bn)
        z_bn = sha256().chain_points([A, yG]).result_scalar()             # :171

        y_powers, y_powers_rev, vec_z2m, d, y_pow, zeta = _common_scalars(bit_length, num_of_proofs, y_bn, z_bn)
        # NB: `secret` was reversed in place at :126, and :206/:212 index the reversed vector
        # against blinding[i]/vec_z2m[i] exactly as written in the reference.
        ip_blinding = 0
        for i in range(num_of_proofs):
            ip_blinding = (ip_blinding + blinding[i] * vec_z2m[i]) % order
        scalar_H = ip_blinding * y_pow % order
        ip_secrets = 0
        for i in range(num_of_proofs):
            ip_secrets = (ip_secrets + secret[i] * vec_z2m[i]) % order
        scalar1_G = ip_secrets * y_pow % order
        scalar_G = (scalar1_G + zeta) % order

        scalars_g_vec = [(0 - z_bn) % order] * nm
        scalars_h_vec = [(d[i] * y_powers_rev[i] % order + z_bn) % order for i in range(nm)]
        A_hat_scalars = scalars_g_vec + scalars_h_vec + [scalar_G, scalar_H]
        A_hat_bases = list(g_vec) + list(h_vec) + [G, H]
        A_hat = ec.msm_naive(A_hat_bases, A_hat_scalars, A)               # :249-251

        aL_hat = [(aL[i] + scalars_g_vec[i]) % order for i in range(nm)]
        aR_hat = [(aR[i] + scalars_h_vec[i]) % order for i in range(nm)]
        alpha_hat = (alpha + scalar_H) % order
        if trace is not None:
            trace.update({"y": y_bn, "z": z_bn, "A_hat": A_hat, "rounds": []})
        wip = WeightedInnerProdArg.prove(g_vec, h_vec, G, H, A_hat, aL_hat, aR_hat, alpha_hat, y_bn, tape,
                                         [], [], trace=None if trace is None else trace["rounds"])
        return RangeProofWIP(A, wip)

    # range_proof_wip.rs:276-382
    def verify(self, stmt, ped_com):
        bit_length = stmt.bit_length
        num_of_proofs = len(ped_com)
        nm = num_of_proofs * bit_length
        g_vec, h_vec, G, H = stmt.g_vec, stmt.h_vec, stmt.G, stmt.H
        order = Q
        y_bn = sha256().chain_points([self.A]).result_scalar()
        yG = ec.mul(ec.G, y_bn)
        z_bn = sha256().chain_points([self.A, yG]).result_scalar()
        y_powers, y_powers_rev, vec_z2m, d, y_pow, scalar_G = _common_scalars(bit_length, num_of_proofs, y_bn, z_bn)
        scalars_g_vec = [(0 - z_bn) % order] * nm
        scalars_h_vec = [(d[i] * y_powers_rev[i] % order + z_bn) % order for i in range(nm)]
        sum_com = ec.msm_naive(ped_com, [y_pow * vec_z2m[i] % order for i in range(num_of_proofs)], self.A)
        A_hat = ec.msm_naive(list(g_vec[:nm]) + list(h_vec[:nm]) + [G],
                             scalars_g_vec + scalars_h_vec + [scalar_G], sum_com)
        v = self.weighted_inner_product_proof.fast_verify(g_vec, h_vec, G, H, A_hat, y_bn)
        return OK if v == OK else RANGE_PROOF_ERROR

    def aggregated_verify_terms(self, stmt, ped_com, trace=None):
        """range_proof_wip.rs:388-602: (init = h*delta', points, scalars) of the single MSM."""
        wip = self.weighted_inner_product_proof
        P = self.A
        n = stmt.bit_length
        m = len(ped_com)
        nm = m * n
        G, H, g, h = stmt.g_vec, stmt.h_vec, stmt.G, stmt.H
        order = Q
        lg_nm = len(wip.L)
        assert len(G) == nm
        assert len(H) == nm
        assert nm & (nm - 1) == 0 and nm > 0, "(n*m) must be a power of two!"
        assert lg_nm <= 64

        y_bn = sha256().chain_points([self.A]).result_scalar()
        yG = ec.mul(ec.G, y_bn)
        z_bn = sha256().chain_points([self.A, yG]).result_scalar()
        y_powers, y_powers_rev, vec_z2m, d, y_pow, zeta = _common_scalars(n, m, y_bn, z_bn)
        y_inv = pow(y_bn, -1, order)
        powers_yinv = [pow(y_inv, i + 1, order) for i in range(nm)]

        e_bn = sha256().chain_points([wip.a_tag, wip.b_tag, g, h]).result_scalar()
        e_sq = e_bn * e_bn % order
        x_sq_vec, m_esq_xsq, m_esq_xinvsq, xs = [], [], [], []
        allinv, all_ = 1, 1
        for Li, Ri in zip(wip.L, wip.R):
            x = sha256().chain_points([Li, Ri, g, h]).result_scalar()
            xs.append(x)
            x_inv = pow(x, -1, order)
            x_sq = x * x % order
            x_inv_sq = x_inv * x_inv % order
            x_sq_vec.append(x_sq)
            m_esq_xsq.append((0 - e_sq * x_sq % order) % order)
            m_esq_xinvsq.append((0 - e_sq * x_inv_sq % order) % order)
            allinv = allinv * x_inv % order
            all_ = all_ * x % order
        s, sg, sh = [allinv], [allinv], [all_]
        for i in range(1, nm):
            lg_i = i.bit_length() - 1
            k = 1 << lg_i
            s.append(s[i - k] * x_sq_vec[(lg_nm - 1) - lg_i] % order)
            sg.append(s[i] * powers_yinv[i - 1] % order)
            sh.append(pow(s[i], -1, order))

        e_sq_zeta = zeta * e_sq % order
        r_times_s_y = (wip.r_prime * wip.s_prime % order) * y_bn % order
        scalar_g = (r_times_s_y - e_sq_zeta) % order
        e_rprime = wip.r_prime * e_bn % order
        e_sq_z = z_bn * e_sq % order
        scalar_g_vec = [(e_rprime * sg[i] % order + e_sq_z) % order for i in range(nm)]
        e_sprime = wip.s_prime * e_bn % order
        scalar_h_vec = [(((e_sprime * sh[i] % order) - e_sq_z) % order
                         - (d[i] * y_powers_rev[i] % order) * e_sq % order) % order for i in range(nm)]
        minus_e_sq = (0 - e_sq) % order
        scalar_P = minus_e_sq
        minus_e_sq_y_pow = minus_e_sq * y_pow % order
        scalar_com = [vec_z2m[i] * minus_e_sq_y_pow % order for i in range(m)]
        scalar_A = (0 - e_bn) % order

        scalars = scalar_g_vec + scalar_h_vec + [scalar_g] + m_esq_xsq + m_esq_xinvsq + [scalar_P] + scalar_com + [scalar_A]
        points = list(G) + list(H) + [g] + list(wip.L) + list(wip.R) + [P] + list(ped_com) + [wip.a_tag]
        if trace is not None:
            trace.update({"y": y_bn, "z": z_bn, "e": e_bn, "xs": xs, "yG": yG})
        h_delta_prime = ec.mul(h, wip.delta_prime)                 # :598
        return h_delta_prime, points, scalars

    def aggregated_verify(self, stmt, ped_com):
        init, points, scalars = self.aggregated_verify_terms(stmt, ped_com)
        lhs = ec.msm_naive(points, scalars, init)
        return OK if lhs == self.weighted_inner_product_proof.b_tag else RANGE_PROOF_ERROR
