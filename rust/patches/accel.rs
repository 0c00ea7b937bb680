//! `src/accel.rs` of the patched bulletproof-kzen crate (added by 0001-accel-shim.patch).
//!
//! Every multi-scalar multiplication and base folding of the four protocol files goes through this
//! module.  With the cargo feature `gpu` the calls go to libbpmsm.so (include/bpmsm.h) through the
//! `bpmsm` wrapper crate; without it they are the crate's original fold idiom, so the patched crate
//! builds and behaves exactly like the unpatched one on a machine without a GPU.
//!
//! `random_scalar()` replaces every `Scalar::<Secp256k1>::random()` of the provers.  It draws from a
//! thread-local tape when one was installed with `set_tape` (the reference's draw order: alpha, rho,
//! SR[0..nm), SL[0..nm), tau1, tau2 for Bulletproofs; alpha, (d_L, d_R) per round, r, s, delta, eta
//! for Bulletproofs+) and from the OS generator otherwise.  With a tape the proof bytes are a pure
//! function of the inputs, which is what cross-implementation vectors need.
use curv::elliptic::curves::{secp256_k1::Secp256k1, Point, Scalar};
use curv::BigInt;
use std::cell::RefCell;
use std::collections::VecDeque;

thread_local! {
    static TAPE: RefCell<Option<VecDeque<Scalar<Secp256k1>>>> = RefCell::new(None);
}

/// Install (Some) or remove (None) the randomness tape of the calling thread.
pub fn set_tape(tape: Option<Vec<Scalar<Secp256k1>>>) {
    TAPE.with(|t| *t.borrow_mut() = tape.map(VecDeque::from));
}

/// `Scalar::random()` with the tape hook.  Panics when an installed tape runs dry: a silent fall back
/// to the OS generator would make the proof bytes irreproducible.
pub fn random_scalar() -> Scalar<Secp256k1> {
    TAPE.with(|t| match t.borrow_mut().as_mut() {
        Some(q) => q.pop_front().expect("accel: randomness tape exhausted"),
        None => Scalar::<Secp256k1>::random(),
    })
}

#[cfg(feature = "gpu")]
mod gpu {
    use bpmsm::Ctx;
    thread_local! {
        // one context (one HIP stream) per host thread; BPMSM_DEVICE selects the GPU
        pub static CTX: Ctx = Ctx::new(std::env::var("BPMSM_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0))
            .expect("accel: bp_ctx_create failed (libbpmsm has no CPU fallback)");
    }
}

/// init + sum_i scalars[i] * points[i]; scalars need not be reduced (the library reduces mod q like
/// `Scalar::from(&BigInt)`), zero scalars and identity points are legal.
pub fn msm(points: &[Point<Secp256k1>], scalars: &[BigInt], init: Point<Secp256k1>) -> Point<Secp256k1> {
    assert_eq!(points.len(), scalars.len());
    #[cfg(feature = "gpu")]
    {
        gpu::CTX.with(|c| c.msm(points, scalars, Some(&init))).expect("accel: bp_msm")
    }
    #[cfg(not(feature = "gpu"))]
    {
        (0..points.len())
            .map(|i| &points[i] * &Scalar::<Secp256k1>::from(&scalars[i]))
            .fold(init, |acc, x| acc + x)
    }
}

/// init + <s1, p1> + <s2, p2>: the L / R commitments of the (weighted) inner-product rounds.  The
/// original folds skip zero scalars; 0 * P is the identity, so the result is the same.
pub fn msm2(
    p1: &[Point<Secp256k1>],
    s1: &[BigInt],
    p2: &[Point<Secp256k1>],
    s2: &[BigInt],
    init: Point<Secp256k1>,
) -> Point<Secp256k1> {
    let mut points = Vec::with_capacity(p1.len() + p2.len());
    points.extend_from_slice(p1);
    points.extend_from_slice(p2);
    let mut scalars = Vec::with_capacity(s1.len() + s2.len());
    scalars.extend_from_slice(s1);
    scalars.extend_from_slice(s2);
    msm(&points, &scalars, init)
}

/// out[i] = s_lo * lo[i] + s_hi * hi[i]: the base folding of every round.
pub fn fold_bases(
    lo: &[Point<Secp256k1>],
    hi: &[Point<Secp256k1>],
    s_lo: &Scalar<Secp256k1>,
    s_hi: &Scalar<Secp256k1>,
) -> Vec<Point<Secp256k1>> {
    assert_eq!(lo.len(), hi.len());
    #[cfg(feature = "gpu")]
    {
        gpu::CTX
            .with(|c| c.fold_bases(lo, hi, &s_lo.to_bigint(), &s_hi.to_bigint()))
            .expect("accel: bp_fold_bases")
    }
    #[cfg(not(feature = "gpu"))]
    {
        (0..lo.len()).map(|i| &hi[i] * s_hi + &lo[i] * s_lo).collect()
    }
}

/// init + sum_{bits[i]} g[i] - sum_{!bits[i]} h[i]: the bit commitment A of both provers (range_proof.rs:93-107,
/// range_proof_wip.rs:152-165).  On the GPU one MSM over g || h with scalars 1 / 0 for g and 0 / q-1 for h.
pub fn signed_unit_sum(g: &[Point<Secp256k1>], h: &[Point<Secp256k1>], bits: &[bool], init: Point<Secp256k1>) -> Point<Secp256k1> {
    assert!(g.len() == bits.len() && h.len() == bits.len());
    #[cfg(feature = "gpu")]
    {
        let one = BigInt::from(1u32);
        let zero = BigInt::from(0u32);
        let minus_one = Scalar::<Secp256k1>::group_order() - &one;
        let mut scalars: Vec<BigInt> = bits.iter().map(|b| if *b { one.clone() } else { zero.clone() }).collect();
        scalars.extend(bits.iter().map(|b| if *b { zero.clone() } else { minus_one.clone() }));
        let mut points = Vec::with_capacity(2 * bits.len());
        points.extend_from_slice(g);
        points.extend_from_slice(h);
        msm(&points, &scalars, init)
    }
    #[cfg(not(feature = "gpu"))]
    {
        let acc = g.iter().zip(bits).fold(init, |acc, x| if *x.1 { acc + x.0 } else { acc });
        h.iter().zip(bits).fold(acc, |acc, x| if !*x.1 { acc - x.0 } else { acc })
    }
}

/// out[i] = points[i] * scalars[i]: hi_tag of prove / verify / fast_verify (range_proof.rs:231-233, 327-329, 462-464).
pub fn scalar_mul_batch(points: &[Point<Secp256k1>], scalars: &[Scalar<Secp256k1>]) -> Vec<Point<Secp256k1>> {
    assert_eq!(points.len(), scalars.len());
    #[cfg(feature = "gpu")]
    {
        let bn: Vec<BigInt> = scalars.iter().map(|s| s.to_bigint()).collect();
        gpu::CTX.with(|c| c.scalar_mul_batch(points, &bn)).expect("accel: bp_scalar_mul_batch")
    }
    #[cfg(not(feature = "gpu"))]
    {
        (0..points.len()).map(|i| &points[i] * &scalars[i]).collect()
    }
}

/// `Point::generator() * s`: the point y*G hashed into z (range_proof.rs:125-126).
pub fn generator_mul(s: &Scalar<Secp256k1>) -> Point<Secp256k1> {
    #[cfg(feature = "gpu")]
    {
        gpu::CTX.with(|c| c.fixed_base_mul(&[s.to_bigint()])).expect("accel: bp_fixed_base_mul_batch").remove(0)
    }
    #[cfg(not(feature = "gpu"))]
    {
        Point::<Secp256k1>::generator() * s
    }
}

/// `base * s` for the caller's G (range_proof.rs:219, 588): the fixed-base kernel when G is the secp256k1 generator (every
/// reference test and bench), one variable-base multiplication otherwise.
pub fn fixed_base_mul(base: &Point<Secp256k1>, s: &Scalar<Secp256k1>) -> Point<Secp256k1> {
    #[cfg(feature = "gpu")]
    {
        if *base == Point::<Secp256k1>::generator().to_point() {
            generator_mul(s)
        } else {
            scalar_mul_batch(std::slice::from_ref(base), std::slice::from_ref(s)).remove(0)
        }
    }
    #[cfg(not(feature = "gpu"))]
    {
        base * s
    }
}
