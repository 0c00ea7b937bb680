//! Safe wrappers the patched reference calls.  Everything a method body of
//! `src/proofs/*.rs` needs at the operator level: `msm`, `fold_bases`, `scalar_mul_batch`, `fixed_base_mul`; and at the
//! method level: `Gens`, `prove`, `verify`, `verify_batch`.
use bpmsm_sys as sys;
use curv::arithmetic::traits::*;
use curv::elliptic::curves::{secp256_k1::Secp256k1, Point, Scalar};
use curv::BigInt;
use std::os::raw::c_int;

#[derive(Debug)]
pub struct Error(pub c_int, pub String);

fn check(rc: c_int) -> Result<(), Error> {
    if rc == sys::BP_OK {
        return Ok(());
    }
    let msg = unsafe { std::ffi::CStr::from_ptr(sys::bp_last_error()) }.to_string_lossy().into_owned();
    Err(Error(rc, msg))
}

/// `Point` -> 64 bytes x || y (identity = 64 zero bytes).
pub fn point_bytes(p: &Point<Secp256k1>) -> [u8; 64] {
    let mut out = [0u8; 64];
    if !p.is_zero() {
        out.copy_from_slice(&p.to_bytes(false)[1..65]);
    }
    out
}
pub fn point_from(b: &[u8]) -> Point<Secp256k1> {
    if b.iter().all(|x| *x == 0) {
        return Point::zero();
    }
    let mut enc = [4u8; 65];
    enc[1..].copy_from_slice(&b[..64]);
    Point::from_bytes(&enc).expect("libbpmsm returns canonical curve points")
}
/// `BigInt` (possibly unreduced, as the reference's scalar vectors are: range_proof.rs:597-619) -> 32 bytes big-endian.
pub fn scalar_bytes(s: &BigInt) -> [u8; 32] {
    let r = Scalar::<Secp256k1>::from(s).to_bigint();
    let v = r.to_bytes();
    let mut out = [0u8; 32];
    out[32 - v.len()..].copy_from_slice(&v);
    out
}

pub struct Ctx(*mut sys::bp_ctx);
unsafe impl Send for Ctx {}

impl Ctx {
    pub fn new(device: i32) -> Result<Ctx, Error> {
        let mut p = std::ptr::null_mut();
        check(unsafe { sys::bp_ctx_create(device, &mut p) })?;
        Ok(Ctx(p))
    }

    /// `(0..n).map(|i| &points[i] * &Scalar::from(&scalars[i])).fold(init, |acc, x| acc + x)`
    pub fn msm(&self, points: &[Point<Secp256k1>], scalars: &[BigInt], init: Option<&Point<Secp256k1>>) -> Result<Point<Secp256k1>, Error> {
        assert_eq!(points.len(), scalars.len());
        let pb: Vec<u8> = points.iter().flat_map(|p| point_bytes(p).to_vec()).collect();
        let sb: Vec<u8> = scalars.iter().flat_map(|s| scalar_bytes(s).to_vec()).collect();
        let ib = init.map(point_bytes);
        let mut out = [0u8; 64];
        check(unsafe {
            sys::bp_msm(self.0, pb.as_ptr(), sb.as_ptr(), points.len(), ib.as_ref().map_or(std::ptr::null(), |b| b.as_ptr()), out.as_mut_ptr())
        })?;
        Ok(point_from(&out))
    }

    /// `G_new[i] = s_lo * lo[i] + s_hi * hi[i]` (inner_product.rs:140-155, weighted_inner_product.rs:189-204)
    pub fn fold_bases(&self, lo: &[Point<Secp256k1>], hi: &[Point<Secp256k1>], s_lo: &BigInt, s_hi: &BigInt) -> Result<Vec<Point<Secp256k1>>, Error> {
        assert_eq!(lo.len(), hi.len());
        let lb: Vec<u8> = lo.iter().flat_map(|p| point_bytes(p).to_vec()).collect();
        let hb: Vec<u8> = hi.iter().flat_map(|p| point_bytes(p).to_vec()).collect();
        let mut out = vec![0u8; 64 * lo.len()];
        check(unsafe {
            sys::bp_fold_bases(self.0, lb.as_ptr(), hb.as_ptr(), lo.len(), scalar_bytes(s_lo).as_ptr(), scalar_bytes(s_hi).as_ptr(), out.as_mut_ptr())
        })?;
        Ok(out.chunks(64).map(point_from).collect())
    }

    /// `hi_tag[i] = h_i * y^-i` (range_proof.rs:231-233)
    pub fn scalar_mul_batch(&self, points: &[Point<Secp256k1>], scalars: &[BigInt]) -> Result<Vec<Point<Secp256k1>>, Error> {
        assert_eq!(points.len(), scalars.len());
        let pb: Vec<u8> = points.iter().flat_map(|p| point_bytes(p).to_vec()).collect();
        let sb: Vec<u8> = scalars.iter().flat_map(|s| scalar_bytes(s).to_vec()).collect();
        let mut out = vec![0u8; 64 * points.len()];
        check(unsafe { sys::bp_scalar_mul_batch(self.0, pb.as_ptr(), sb.as_ptr(), points.len(), out.as_mut_ptr()) })?;
        Ok(out.chunks(64).map(point_from).collect())
    }

    /// `Point::generator() * s` for each s (range_proof.rs:126, 219, 588)
    pub fn fixed_base_mul(&self, scalars: &[BigInt]) -> Result<Vec<Point<Secp256k1>>, Error> {
        let sb: Vec<u8> = scalars.iter().flat_map(|s| scalar_bytes(s).to_vec()).collect();
        let mut out = vec![0u8; 64 * scalars.len()];
        check(unsafe { sys::bp_fixed_base_mul_batch(self.0, sb.as_ptr(), scalars.len(), out.as_mut_ptr()) })?;
        Ok(out.chunks(64).map(point_from).collect())
    }
}
impl Drop for Ctx {
    fn drop(&mut self) {
        unsafe { sys::bp_ctx_destroy(self.0) }
    }
}

/// Resident generators + fixed-base table (`g_vec`, `h_vec`, `G`, `H` of every reference call).
pub struct Gens(*mut sys::bp_gens);
unsafe impl Send for Gens {}
unsafe impl Sync for Gens {}

impl Gens {
    pub fn upload(ctx: &Ctx, g_vec: &[Point<Secp256k1>], h_vec: &[Point<Secp256k1>], g: &Point<Secp256k1>, h: &Point<Secp256k1>) -> Result<Gens, Error> {
        assert_eq!(g_vec.len(), h_vec.len());
        let gb: Vec<u8> = g_vec.iter().flat_map(|p| point_bytes(p).to_vec()).collect();
        let hb: Vec<u8> = h_vec.iter().flat_map(|p| point_bytes(p).to_vec()).collect();
        let mut out = std::ptr::null_mut();
        check(unsafe {
            sys::bp_gens_create(ctx.0, gb.as_ptr(), hb.as_ptr(), g_vec.len(), point_bytes(g).as_ptr(), point_bytes(h).as_ptr(), 0, &mut out)
        })?;
        Ok(Gens(out))
    }
}
impl Drop for Gens {
    fn drop(&mut self) {
        unsafe { sys::bp_gens_destroy(self.0) }
    }
}

/// `RangeProof::aggregated_verify` for `nproofs` canonical proof encodings that share `gens`:
/// one verdict byte each (0 = Ok, 3 = RangeProofError, 255 = would not deserialize).  `plus` selects Bulletproofs+.
pub fn verify_batch(ctx: &Ctx, gens: &Gens, n_bits: usize, m: usize, proofs: &[u8], proof_len: usize, coms: &[Point<Secp256k1>],
                    plus: bool) -> Result<Vec<u8>, Error> {
    let nproofs = proofs.len() / proof_len;
    assert_eq!(coms.len(), nproofs * m);
    let cb: Vec<u8> = coms.iter().flat_map(|p| point_bytes(p).to_vec()).collect();
    let mut verdicts = vec![0u8; nproofs];
    let f = if plus { sys::bp_wip_rangeproof_verify_batch } else { sys::bp_rangeproof_verify_batch };
    check(unsafe { f(ctx.0, gens.0, n_bits as c_int, m as c_int, proofs.as_ptr(), proof_len, cb.as_ptr(), nproofs, verdicts.as_mut_ptr()) })?;
    Ok(verdicts)
}

/// `RangeProof::prove` / `RangeProofWIP::prove` with the randomness drawn by the caller in the reference's order
/// (alpha, rho, SR[nm], SL[nm], tau1, tau2 / alpha, (d_L, d_R) per round, r, s, delta, eta): canonical proof bytes.
pub fn prove(ctx: &Ctx, gens: &Gens, n_bits: usize, secrets: &[BigInt], blindings: &[BigInt], tape: &[Scalar<Secp256k1>], plus: bool) -> Result<Vec<u8>, Error> {
    let sb: Vec<u8> = secrets.iter().flat_map(|s| scalar_bytes(s).to_vec()).collect();
    let bb: Vec<u8> = blindings.iter().flat_map(|s| scalar_bytes(s).to_vec()).collect();
    let tb: Vec<u8> = tape.iter().flat_map(|s| scalar_bytes(&s.to_bigint()).to_vec()).collect();
    let mut out = vec![0u8; 4096];
    let mut len = 0usize;
    let f = if plus { sys::bp_wip_rangeproof_prove } else { sys::bp_rangeproof_prove };
    check(unsafe {
        f(ctx.0, gens.0, n_bits as c_int, sb.as_ptr(), bb.as_ptr(), secrets.len() as c_int, tb.as_ptr(), tb.len(), out.as_mut_ptr(), out.len(), &mut len)
    })?;
    out.truncate(len);
    Ok(out)
}
