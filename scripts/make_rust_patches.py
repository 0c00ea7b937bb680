#!/usr/bin/env python3
"""Writes rust/patches/*.patch: unified diffs against the reference crate (/root/reference) that route every
MSM-shaped fold, every base folding and every Scalar::random() of the four protocol files through src/accel.rs
(rust/patches/accel.rs), i.e. through libbpmsm.so when the cargo feature `gpu` is on.

Edits are given as (first line, last line, replacement) against the reference's line numbers, so this script holds
only the new text; the context lines of the diffs come from the reference at generation time.  Apply with
    cd bulletproofs && for p in <repo>/rust/patches/00*.patch; do patch -p1 < $p; done && cp -r <repo>/rust/bpmsm* .
tests/test_rust_patches.py checks that every hunk still applies to /root/reference.
"""
import difflib, os, sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rust", "patches")
RND = "crate::accel::random_scalar()"

def sub_random(lines, lo, hi):
    """replace Scalar::<Secp256k1>::random() on the given 1-based lines (non-test code only)"""
    return [(n, n, [lines[n - 1].replace("Scalar::<Secp256k1>::random()", RND)]) for n in range(lo, hi + 1)
            if "Scalar::<Secp256k1>::random()" in lines[n - 1]]

def edits_for(path, lines):
    E = []
    if path == "Cargo.toml":
        i = next(k for k, l in enumerate(lines) if l.startswith("generic-array")) + 1
        E.append((i + 1, i, ['bpmsm = { path = "bpmsm", optional = true }\n']))
        j = next(k for k, l in enumerate(lines) if l.startswith("default = [")) + 1
        E.append((j + 1, j, ['# every MSM / base folding through libbpmsm.so (MI355X); off: the original CPU folds\n', 'gpu = ["bpmsm"]\n']))
    elif path == "src/lib.rs":
        i = next(k for k, l in enumerate(lines) if l.startswith("extern crate sha2;")) + 1
        E.append((i + 1, i, ['#[cfg(feature = "gpu")]\n', 'extern crate bpmsm;\n']))
        j = next(k for k, l in enumerate(lines) if l.startswith("pub mod proofs;")) + 1
        E.append((j, j - 1, ['pub mod accel;\n']))
    elif path == "src/proofs/range_proof.rs":
        E += sub_random(lines, 57, 58) + sub_random(lines, 110, 113) + sub_random(lines, 164, 165)
        # A = H*alpha + sum_{bit} g_i - sum_{!bit} h_i   (range_proof.rs:93-107): a signed-unit MSM
        E.append((93, 107, ["        A = crate::accel::signed_unit_sum(&g_vec, &h_vec, &secret_bits, A);\n"]))
        # S = H*rho + <SL, g> + <SR, h>   (range_proof.rs:116-122)
        E.append((116, 122, [
            "        let SL_bn = SL.iter().map(|s| s.to_bigint()).collect::<Vec<BigInt>>();\n",
            "        let SR_bn = SR.iter().map(|s| s.to_bigint()).collect::<Vec<BigInt>>();\n",
            "        S = crate::accel::msm2(&g_vec, &SL_bn, &h_vec, &SR_bn, S);\n",
            "        let _ = &mut index;\n"]))
        # y*G with the secp256k1 generator (range_proof.rs:125-126), x_u*G / challenge_x*G with the caller's G (:219, :588)
        E.append((125, 126, ["        let yG: Point<Secp256k1> = crate::accel::generator_mul(&y);\n"]))
        E.append((219, 219, ["        let Gx = crate::accel::fixed_base_mul(G, &challenge_x);\n"]))
        E.append((588, 588, ["        let ux = crate::accel::fixed_base_mul(G, &x_u_fe);\n"]))
        # hi_tag[i] = h_i * y^-i in prove, verify and fast_verify (range_proof.rs:231-233, 327-329, 462-464)
        for a in (231, 327, 462):
            E.append((a, a + 2, ["        let hi_tag = crate::accel::scalar_mul_batch(&h_vec, &yi_inv);\n"]))
        # P' = P' g^l h'^r (range_proof.rs:236-244): two folds -> one MSM
        E.append((235, 244, ["        // P' = P' g^l h'^r\n", "        let P = crate::accel::msm2(&g_vec, &Lp, &hi_tag, &Rp, P);\n"]))
        # verify / fast_verify: P1 = P + <z y^i + z^(2+j) 2^k, hi_tag>, P = P1 + <-z, g_vec> (range_proof.rs:375-391, 510-526):
        # the closure that forms the scalar is kept (reference lines, taken at generation time), its point product and the two
        # folds become one MSM
        for a in (375, 510):
            body = lines[a:a + 8]  # .map(|i| { ... let zyn_zsq2n = ...;   (lines a+1 .. a+8)
            E.append((a, a + 16,
                      ["        let P1_scalars = (0..nm)\n"] + body +
                      ["                zyn_zsq2n\n", "            })\n", "            .collect::<Vec<BigInt>>();\n",
                       "        let z_minus_bn = vec![z_minus_fe.to_bigint(); nm];\n",
                       "        let P = crate::accel::msm2(&hi_tag, &P1_scalars, &g_vec, &z_minus_bn, P);\n"]))
        # aggregated_verify's single multi-exponentiation (range_proof.rs:771-774)
        E.append((771, 774, ["        let lhs = crate::accel::msm(&points, &scalars, H_times_scalar_H);\n"]))
    elif path == "src/proofs/inner_product.rs":
        # L, R of a round (inner_product.rs:73-90, 98-115)
        E.append((73, 90, ["            let L = crate::accel::msm2(G_R, a_L, H_L, b_R, ux_CL);\n"]))
        E.append((98, 115, ["            let R = crate::accel::msm2(G_L, a_R, H_R, b_L, ux_CR);\n"]))
        for a in (140, 203):  # G_new, H_new in prove and in the recursive verify
            E.append((a, a + 6, ["            let G_new = crate::accel::fold_bases(G_L, G_R, &x_inv_fe, &x);\n"]))
            E.append((a + 9, a + 15, ["            let H_new = crate::accel::fold_bases(H_L, H_R, &x, &x_inv_fe);\n"]))
        # fast_verify's single multi-exponentiation (inner_product.rs:334-338)
        E.append((334, 338, ["        let expect_P = crate::accel::msm(&points, &scalars, ux_c);\n"]))
    elif path == "src/proofs/weighted_inner_product.rs":
        E += sub_random(lines, 106, 106) + sub_random(lines, 134, 134) + sub_random(lines, 213, 219)
        E.append((109, 126, ["            let L = crate::accel::msm2(G_R, &yninv_aL, H_L, b_R, g_cL_h_dL);\n"]))
        E.append((137, 154, ["            let R = crate::accel::msm2(G_L, &yn_aR, H_R, b_L, g_cR_h_dR);\n"]))
        for a in (189, 312):  # G_hat, H_hat in prove and in the recursive verify
            E.append((a, a + 6, ["            let G_hat = crate::accel::fold_bases(G_L, G_R, &x_inv_fe, &x_yinv_fe);\n"]))
            E.append((a + 9, a + 15, ["            let H_hat = crate::accel::fold_bases(H_L, H_R, &x, &x_inv_fe);\n"]))
        E.append((501, 504, ["        let lhs = crate::accel::msm(&points, &scalars, h_delta_prime);\n"]))
    elif path == "src/proofs/range_proof_wip.rs":
        E += sub_random(lines, 150, 150)
        # A = H*alpha + sum_{bit} g_i - sum_{!bit} h_i   (range_proof_wip.rs:152-165)
        E.append((152, 165, ["        A = crate::accel::signed_unit_sum(&g_vec, &h_vec, &secret_bits, A);\n"]))
        # A_hat in prove (:249-251) and in verify (:370-372)
        E.append((249, 251, ["        let A_hat = crate::accel::msm(&A_hat_bases, &A_hat_scalars, A.clone());\n"]))
        E.append((370, 372, ["        let A_hat = crate::accel::msm(&A_hat_bases, &A_hat_scalars, sum_com);\n"]))
        E.append((599, 602, ["        let lhs = crate::accel::msm(&points, &scalars, h_delta_prime);\n"]))
    return E

def apply(lines, edits):
    out = list(lines)
    for lo, hi, new in sorted(edits, key=lambda e: -e[0]):  # bottom-up keeps the line numbers valid; hi = lo - 1 inserts
        out[lo - 1:hi] = new
    return out

PATCHES = [
    ("0001-accel-shim.patch", ["Cargo.toml", "src/lib.rs"]),
    ("0002-range_proof.patch", ["src/proofs/range_proof.rs"]),
    ("0003-inner_product.patch", ["src/proofs/inner_product.rs"]),
    ("0004-weighted_inner_product.patch", ["src/proofs/weighted_inner_product.rs"]),
    ("0005-range_proof_wip.patch", ["src/proofs/range_proof_wip.rs"]),
]

def main():
    os.makedirs(OUT, exist_ok=True)
    for name, files in PATCHES:
        text = ""
        for f in files:
            old = open(os.path.join(REF, f)).read().splitlines(keepends=True)
            new = apply(old, edits_for(f, old))
            text += "".join(difflib.unified_diff(old, new, "a/" + f, "b/" + f, n=3))
        if name.startswith("0001"):
            acc = open(os.path.join(OUT, "accel.rs")).read().splitlines(keepends=True)
            text += "".join(difflib.unified_diff([], acc, "/dev/null", "b/src/accel.rs", n=0))
        open(os.path.join(OUT, name), "w").write(text)
        print(name, text.count("\n@@"), "hunks")

if __name__ == "__main__":
    main()
