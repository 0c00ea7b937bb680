"""Restatement of /root/reference/src/proofs/range_proof.rs (test infrastructure).

`RangeProof::{prove, verify, fast_verify, aggregated_verify}`; randomness comes from a Tape in
the reference's draw order (alpha, rho, SR[0..nm), SL[0..nm), tau1, tau2).
"""
from . import ec
from .curv import sha256
from .ec import Q
from .inner_product import InnerProductArg, OK

RANGE_PROOF_ERROR = "RangeProofError"


class RangeProof:
    """range_proof.rs:32-42."""

    def __init__(self, A, S, T1, T2, tau_x, miu, tx, inner_product_proof):
        self.A, self.S, self.T1, self.T2 = A, S, T1, T2
        self.tau_x, self.miu, self.tx = tau_x, miu, tx
        self.inner_product_proof = inner_product_proof

    # range_proof.rs:45-263
    @staticmethod
    def prove(g_vec, h_vec, G, H, secret, blinding, bit_length, tape, trace=None):
        secret = list(secret)
        num_of_proofs = len(secret)
        nm = num_of_proofs * bit_length
        alpha = tape.random_scalar()          # :57
        rho = tape.random_scalar()            # :58
        g_vec, h_vec = list(g_vec), list(h_vec)

        A = ec.mul(H, alpha)
        S = ec.mul(H, rho)
        order = Q

        secret.reverse()                      # :70
        secret_agg = 0
        for x in secret:
            secret_agg = (secret_agg << bit_length) + x
        aL = [(secret_agg >> i) % 2 for i in range(nm)]
        aR = [(aL[i] - 1) % order for i in range(nm)]
        secret_bits = [(aL[i] & 1) == 1 for i in range(nm)]

        for gi, bit in zip(g_vec, secret_bits):     # :93-102
            if bit:
                A = ec.add(A, gi)
        for hi, bit in zip(h_vec, secret_bits):     # :104-107
            if not bit:
                A = ec.sub(A, hi)

        SR = [tape.random_scalar() for _ in range(nm)]   # :109-111  (SR before SL)
        SL = [tape.random_scalar() for _ in range(nm)]   # :112-114

        for i in range(nm):                               # :116-122
            g_vec_i_SLi = ec.mul(g_vec[i], SL[i])
            h_vec_i_SRi = ec.mul(h_vec[i], SR[i])
            S = ec.add(S, ec.add(h_vec_i_SRi, g_vec_i_SLi))

        y = sha256().chain_points([A, S]).result_scalar()       # :124
        yG = ec.mul(ec.G, y)
        z = sha256().chain_points([yG]).result_scalar()         # :127
        z_bn = z

        yi = [pow(y, i, order) for i in range(nm)]              # :131-133 (iterate)
        t2 = 0
        for i in range(nm):
            t2 = (t2 + SR[i] * yi[i] % order * SL[i]) % order   # :135-137
        vec_2n = [pow(2, i, order) for i in range(bit_length)]

        t1 = 0
        for i in range(nm):                                     # :145-162
            t1_1 = (aR[i] + z_bn) % order
            t1_2 = t1_1 * yi[i] % order
            t1_3 = SL[i] * t1_2 % order
            t1_4 = (aL[i] - z_bn) % order
            t1_5 = SR[i] * yi[i] % order
            t1_6 = t1_4 * t1_5 % order
            j = i // bit_length + 2
            k = i % bit_length
            z_index = pow(z_bn, j, order)
            t1_7 = z_index * vec_2n[k] % order
            t1_8 = t1_7 * SL[i] % order
            t1 = (t1 + (t1_3 + (t1_6 + t1_8) % order) % order) % order

        tau1 = tape.random_scalar()           # :164
        tau2 = tape.random_scalar()           # :165
        T1 = ec.add(ec.mul(G, t1), ec.mul(H, tau1))
        T2 = ec.add(ec.mul(G, t2), ec.mul(H, tau2))

        fs_challenge = sha256().chain_points([T1, T2, G, H]).result_scalar()    # :171
        fs_challenge_square = fs_challenge * fs_challenge % order
        taux_1 = fs_challenge * tau1 % order
        taux_2 = fs_challenge_square * tau2 % order
        taux_3 = taux_2
        for i in range(num_of_proofs):                          # :175-182
            z_j = pow(z_bn, (2 + i) % order, order)
            taux_3 = (taux_3 + z_j * blinding[i]) % order
        tau_x = (taux_1 + taux_3) % order
        miu = (rho * fs_challenge + alpha) % order

        Lp = [((SL[i] * fs_challenge) % order + (aL[i] - z_bn) % order) % order for i in range(nm)]
        Rp = []
        for i in range(nm):                                     # :194-207
            Rp_1 = SR[i] * fs_challenge % order
            j = i // bit_length + 2
            k = i % bit_length
            Rp_2 = pow(z_bn, j, order) * vec_2n[k] % order
            Rp_3 = ((z_bn + aR[i]) % order + Rp_1) % order
            Rp_4 = yi[i] * Rp_3 % order
            Rp.append((Rp_4 + Rp_2) % order)
        tx = 0
        for l, r in zip(Lp, Rp):
            tx = (tx + l * r) % order

        challenge_x = sha256().chain_bigint(tau_x).chain_bigint(miu).chain_bigint(tx).result_scalar()
        Gx = ec.mul(G, challenge_x)           # :219
        P = ec.mul(Gx, tx)                    # :221

        yi_inv = [pow(v, -1, order) for v in yi]                # :223-229
        hi_tag = ec.mul_batch(h_vec, yi_inv)                    # :231-233
        P = ec.msm_naive(g_vec, Lp, P)                          # :236-239
        P = ec.msm_naive(hi_tag, Rp, P)                         # :241-244
        if trace is not None:
            trace.update({"y": y, "z": z, "x": fs_challenge, "x_u": challenge_x, "Gx": Gx, "P": P,
                          "t1": t1, "t2": t2, "rounds": []})
        ipp = InnerProductArg.prove(g_vec, hi_tag, Gx, P, Lp, Rp, [], [],
                                    trace=None if trace is None else trace["rounds"])
        return RangeProof(A, S, T1, T2, tau_x, miu, tx, ipp)

    # shared prefix of verify (:274-391) and fast_verify (:409-526)
    def _verify_common(self, g_vec, h_vec, G, H, ped_com, bit_length):
        num_of_proofs = len(ped_com)
        nm = num_of_proofs * bit_length
        order = Q
        y = sha256().chain_points([self.A, self.S]).result_scalar()
        yG = ec.mul(ec.G, y)
        z = sha256().chain_points([yG]).result_scalar()
        z_bn = z
        z_minus = (order - z) % order
        z_squared = pow(z, 2, order)
        yi = [pow(y, i, order) for i in range(nm)]
        scalar_mul_yn = sum(yi) % order
        vec_2n = [pow(2, i, order) for i in range(bit_length)]
        scalar_mul_2n = sum(vec_2n) % order
        z_cubed_scalar_mul_2n = 0
        for i in range(num_of_proofs):
            z_j = pow(z_bn, (3 + i) % order, order)
            z_cubed_scalar_mul_2n = (z_cubed_scalar_mul_2n + z_j * scalar_mul_2n) % order
        z_minus_zsq = (z_bn - z_squared) % order
        delta = (z_minus_zsq * scalar_mul_yn % order - z_cubed_scalar_mul_2n) % order

        yi_inv = [pow(v, -1, order) for v in yi]
        hi_tag = ec.mul_batch(h_vec, yi_inv)

        fs_challenge = sha256().chain_points([self.T1, self.T2, G, H]).result_scalar()
        fs_challenge_square = fs_challenge * fs_challenge % order

        # eq 65
        left_side = ec.add(ec.mul(G, self.tx), ec.mul(H, self.tau_x))
        Gdelta = ec.mul(G, delta)
        Tx = ec.mul(self.T1, fs_challenge)
        Tx_sq = ec.mul(self.T2, fs_challenge_square)
        vec_ped_zm = [ec.mul(ped_com[i], pow(z_bn, 2 + i, order)) for i in range(num_of_proofs)]
        ped_com_sum = vec_ped_zm.pop(0)        # `.remove(0)` panics for m == 0 (:352)
        for x in vec_ped_zm:
            ped_com_sum = ec.add(ped_com_sum, x)
        right_side = ec.add(ec.add(ec.add(ped_com_sum, Gdelta), Tx), Tx_sq)

        challenge_x = sha256().chain_bigint(self.tau_x).chain_bigint(self.miu).chain_bigint(self.tx).result_scalar()
        Gx = ec.mul(G, challenge_x)
        P = ec.mul(Gx, self.tx)
        minus_miu = (order - self.miu) % order
        Hmiu = ec.mul(H, minus_miu)
        Sx = ec.mul(self.S, fs_challenge)
        P = ec.add(ec.add(ec.add(Hmiu, P), self.A), Sx)

        p1_scalars = []
        for i in range(nm):
            z_yn = z_bn * yi[i] % order
            j = i // bit_length
            k = i % bit_length
            z_j = pow(z_bn, 2 + j, order)
            p1_scalars.append((z_yn + z_j * vec_2n[k]) % order)
        P1 = ec.msm_naive(hi_tag, p1_scalars, P)
        P = ec.msm_naive(g_vec, [z_minus] * nm, P1)
        return hi_tag, Gx, P, left_side == right_side

    def verify(self, g_vec, h_vec, G, H, ped_com, bit_length):
        """range_proof.rs:265-398."""
        hi_tag, Gx, P, eq65 = self._verify_common(g_vec, h_vec, G, H, ped_com, bit_length)
        v = self.inner_product_proof.verify(g_vec, hi_tag, Gx, P)
        return OK if (v == OK and eq65) else RANGE_PROOF_ERROR

    def fast_verify(self, g_vec, h_vec, G, H, ped_com, bit_length):
        """range_proof.rs:400-535."""
        hi_tag, Gx, P, eq65 = self._verify_common(g_vec, h_vec, G, H, ped_com, bit_length)
        v = self.inner_product_proof.fast_verify(g_vec, hi_tag, Gx, P)
        return OK if (v == OK and eq65) else RANGE_PROOF_ERROR

    def aggregated_verify_terms(self, g_vec, h_vec, G, H, ped_com, bit_length, trace=None):
        """range_proof.rs:537-774: returns (init = H*scalar_H, points, scalars) of the single MSM."""
        n = bit_length
        m = len(ped_com)
        nm = m * n
        ipp = self.inner_product_proof
        lg_nm = len(ipp.L)
        order = Q
        assert len(g_vec) == nm
        assert len(h_vec) == nm
        assert nm & (nm - 1) == 0 and nm > 0, "(n*m) must be a power of two!"
        assert lg_nm <= 64

        y = sha256().chain_points([self.A, self.S]).result_scalar()
        y_bn = y
        y_inv_bn = pow(y_bn, -1, order)
        yG = ec.mul(ec.G, y)
        z = sha256().chain_points([yG]).result_scalar()
        z_bn = z
        z_squared = pow(z_bn, 2, order)
        challenge_x = sha256().chain_points([self.T1, self.T2, G, H]).result_scalar()
        challenge_x_sq = challenge_x * challenge_x % order
        x_u_fe = sha256().chain_bigint(self.tau_x).chain_bigint(self.miu).chain_bigint(self.tx).result_scalar()
        ux = ec.mul(G, x_u_fe)                                    # :588
        challenge_ver_bn = sha256().chain_points([self.A, self.S, self.T1, self.T2, G, H]).result_scalar()

        # The reference keeps these as UNREDUCED BigInt products (:597-619); values are equal
        # mod q, which is all that reaches the group, so they are reduced here.
        z2_vec = [pow(z_bn, 2 + i, order) for i in range(m)]
        y_vec = [pow(y_bn, i, order) for i in range(nm)]
        sum_y_pow = sum(y_vec) % order
        vec_2n = [pow(2, i, order) for i in range(n)]
        y_inv_vec = [pow(y_inv_bn, i, order) for i in range(nm)]
        d = [vec_2n[i % n] * z2_vec[i // n] % order for i in range(nm)]
        sum_d = sum(d) % order
        z_minus_zsq = (z_bn - z_squared) % order
        delta = (z_minus_zsq * sum_y_pow % order - sum_d * z_bn % order) % order

        x_sq_vec, minus_x_sq_vec, minus_x_inv_sq_vec = [], [], []
        xs = []
        allinv = 1
        for Li, Ri in zip(ipp.L, ipp.R):
            x = sha256().chain_points([Li, Ri, ux]).result_scalar()    # :656
            xs.append(x)
            x_inv = pow(x, -1, order)
            x_sq = x * x % order
            x_inv_sq = x_inv * x_inv % order
            x_sq_vec.append(x_sq)
            minus_x_sq_vec.append((0 - x_sq) % order)
            minus_x_inv_sq_vec.append((0 - x_inv_sq) % order)
            allinv = allinv * x_inv % order
        s = [allinv]
        for i in range(1, nm):
            lg_i = i.bit_length() - 1
            k = 1 << lg_i
            s.append(s[i - k] * x_sq_vec[(lg_nm - 1) - lg_i] % order)
        a_times_s = [s[i] * ipp.a_tag % order for i in range(nm)]
        b_times_sinv = [pow(s[i], -1, order) * ipp.b_tag % order for i in range(nm)]
        scalar_g_vec = [(a_times_s[i] + z_bn) % order for i in range(nm)]
        scalar_h_vec = [((y_inv_vec[i] * ((b_times_sinv[i] - d[i]) % order)) % order - z_bn) % order
                        for i in range(nm)]
        ab = ipp.a_tag * ipp.b_tag % order
        scalar_G1 = x_u_fe * ((ab - self.tx) % order) % order
        scalar_G2 = challenge_ver_bn * ((delta - self.tx) % order) % order
        scalar_G = (scalar_G1 + scalar_G2) % order
        scalar_H = (self.miu - challenge_ver_bn * self.tau_x % order) % order
        scalar_S = (0 - challenge_x) % order
        scalar_coms = [challenge_ver_bn * z2_vec[i] % order for i in range(m)]
        scalar_T1 = challenge_ver_bn * challenge_x % order
        scalar_T2 = challenge_ver_bn * challenge_x_sq % order

        scalars = (scalar_g_vec + scalar_h_vec + [scalar_G, scalar_S] + minus_x_sq_vec
                   + minus_x_inv_sq_vec + scalar_coms + [scalar_T1, scalar_T2])
        points = (list(g_vec) + list(h_vec) + [G, self.S] + list(ipp.L) + list(ipp.R)
                  + list(ped_com) + [self.T1, self.T2])
        if trace is not None:
            trace.update({"y": y, "z": z, "x": challenge_x, "x_u": x_u_fe, "c": challenge_ver_bn,
                          "xs": xs, "scalar_H": scalar_H, "yG": yG, "ux": ux})
        H_times_scalar_H = ec.mul(H, scalar_H)                    # :770
        return H_times_scalar_H, points, scalars

    def aggregated_verify(self, g_vec, h_vec, G, H, ped_com, bit_length):
        """range_proof.rs:537-782."""
        init, points, scalars = self.aggregated_verify_terms(g_vec, h_vec, G, H, ped_com, bit_length)
        lhs = ec.msm_naive(points, scalars, init)                 # :772-774
        return OK if lhs == self.A else RANGE_PROOF_ERROR         # :777
