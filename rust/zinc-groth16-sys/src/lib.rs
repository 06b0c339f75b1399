//! FFI binding of `include/zinc_groth16.h` plus `create_random_proof`, a drop-in for
//! `franklin_crypto::bellman::groth16::create_random_proof` at
//! zinc-vm/src/core/circuit/facade.rs:153 and zinc-vm/src/core/contract/facade.rs:387.
//!
//! NOT COMPILED IN THIS REPOSITORY'S ENVIRONMENT (no Rust toolchain): kept as the source a zinc-vm
//! maintainer adds.  Field elements are passed as-is: ff_ce's `Fr` is `#[repr]`-compatible with
//! 4 little-endian u64 Montgomery limbs, which is the ABI's element layout.
#![allow(non_camel_case_types)]
use franklin_crypto::bellman::groth16::{Parameters, Proof};
use franklin_crypto::bellman::pairing::bn256::{Bn256, Fq, Fq2, Fr, G1Affine, G2Affine};
use franklin_crypto::bellman::pairing::ff::{Field, PrimeField};
use franklin_crypto::bellman::pairing::CurveAffine;
use franklin_crypto::bellman::{Circuit, ConstraintSystem, Index, LinearCombination, SynthesisError, Variable};
use rand::Rng;
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)] pub struct zg16_ctx { _p: [u8; 0] }
#[repr(C)] pub struct zg16_pk { _p: [u8; 0] }
#[repr(C)] pub struct zg16_r1cs { _p: [u8; 0] }

#[repr(C)]
pub struct zg16_pk_view {
    pub alpha_g1: *const u64, pub beta_g1: *const u64, pub beta_g2: *const u64,
    pub delta_g1: *const u64, pub delta_g2: *const u64,
    pub h: *const u64, pub n_h: usize, pub l: *const u64, pub n_l: usize,
    pub a: *const u64, pub n_a: usize, pub b_g1: *const u64, pub n_b_g1: usize,
    pub b_g2: *const u64, pub n_b_g2: usize,
}
#[repr(C)] pub struct zg16_csr { pub row_ptr: *const u64, pub col: *const u32, pub coeff: *const u64 }
#[repr(C)]
pub struct zg16_r1cs_view { pub n_rows: usize, pub n_inputs: usize, pub n_aux: usize, pub a: zg16_csr, pub b: zg16_csr, pub c: zg16_csr }
#[repr(C)] pub struct zg16_proof { pub a: [u64; 8], pub b: [u64; 16], pub c: [u64; 8] }
#[repr(C)]
pub struct zg16_vk_view { pub alpha_g1: *const u64, pub beta_g2: *const u64, pub gamma_g2: *const u64, pub delta_g2: *const u64, pub ic: *const u64, pub n_ic: usize }

extern "C" {
    pub fn zg16_ctx_create(device_id: c_int, out: *mut *mut zg16_ctx) -> c_int;
    pub fn zg16_ctx_destroy(ctx: *mut zg16_ctx);
    pub fn zg16_last_error(ctx: *const zg16_ctx) -> *const c_char;
    pub fn zg16_pk_upload(ctx: *mut zg16_ctx, view: *const zg16_pk_view, shard_index: c_int, shard_count: c_int, out: *mut *mut zg16_pk) -> c_int;
    pub fn zg16_pk_free(pk: *mut zg16_pk);
    pub fn zg16_r1cs_upload(ctx: *mut zg16_ctx, view: *const zg16_r1cs_view, out: *mut *mut zg16_r1cs) -> c_int;
    pub fn zg16_r1cs_free(r: *mut zg16_r1cs);
    pub fn zg16_prove(ctx: *mut zg16_ctx, pk: *const zg16_pk, r1cs: *const zg16_r1cs, z_in: *const c_void, z_aux: *const c_void,
                      witness_on_device: c_int, r: *const u64, s: *const u64, out: *mut zg16_proof) -> c_int;
    /// `Parameters::<Bn256>::read(file, checked)` on the GPU: `data` is the proving-key file
    /// (zinc-vm/src/zvm/arguments/command/prove.rs:73-76).
    pub fn zg16_pk_upload_bytes(ctx: *mut zg16_ctx, data: *const u8, len: usize, checked: c_int,
                                shard_index: c_int, shard_count: c_int, out: *mut *mut zg16_pk) -> c_int;
    /// For a host that keeps bellman's own `ProvingAssignment`: a, b, c are the evaluation vectors
    /// `enforce` pushed (n_rows Fr each, the `input_i * 0 = 0` rows included).
    pub fn zg16_prove_from_evals(ctx: *mut zg16_ctx, pk: *const zg16_pk, r1cs: *const zg16_r1cs,
                                 a: *const c_void, b: *const c_void, c: *const c_void, evals_on_device: c_int,
                                 z_in: *const c_void, z_aux: *const c_void, witness_on_device: c_int,
                                 r: *const u64, s: *const u64, out: *mut zg16_proof) -> c_int;
    /// bellman `Proof::write` (the 128 bytes `zvm prove` prints as hex, prove.rs:171-174).
    pub fn zg16_proof_encode(proof: *const zg16_proof, out: *mut u8) -> c_int;
    /// The process-wide context of a GPU and its cache of resident keys / circuits: what lets a caller that
    /// cannot keep a handle (prove.rs:73-88, zinc-tester proof_check.rs:122) pay for the key upload once.
    pub fn zg16_default_ctx(device_id: c_int, out: *mut *mut zg16_ctx) -> c_int;
    pub fn zg16_pk_cached(ctx: *mut zg16_ctx, view: *const zg16_pk_view, full_digest: c_int, out: *mut *mut zg16_pk, was_cached: *mut c_int) -> c_int;
    /// bellman `verify_proof` (zinc-vm/src/core/facade.rs:32-38); host arithmetic, no context.
    pub fn zg16_verify_proof(vk: *const zg16_vk_view, proof: *const zg16_proof, public_inputs: *const u64, n_inputs: usize, accepted: *mut c_int) -> c_int;
    /// A serial number no later key handle of the process repeats; `zg16_pk_cache_holds` wants it beside the address.
    pub fn zg16_pk_id(pk: *const zg16_pk) -> u64;
    pub fn zg16_pk_cache_holds(ctx: *mut zg16_ctx, pk: *const zg16_pk, id: u64) -> c_int;
    /// The shape the resident matrices were uploaded with (any pointer may be null).
    pub fn zg16_r1cs_shape(r: *const zg16_r1cs, n_rows: *mut usize, n_inputs: *mut usize, n_aux: *mut usize) -> c_int;
    pub fn zg16_r1cs_cached(ctx: *mut zg16_ctx, circuit_id: *const u8, circuit_id_len: usize, view: *const zg16_r1cs_view,
                            out: *mut *mut zg16_r1cs, was_cached: *mut c_int) -> c_int;
}

/// Plays the role of bellman's `ProvingAssignment`, but records the matrices instead of evaluating
/// rows on the host.  Sits *below* zinc-vm's `DedupCS(LoggingCS(cs))` wrappers
/// (zinc-vm/src/core/circuit/synthesizer.rs:34), so every row arrives in canonical form.
pub struct RecordingAssignment {
    pub z_in: Vec<Fr>, pub z_aux: Vec<Fr>,
    row_ptr: [Vec<u64>; 3], col: [Vec<u32>; 3], coeff: [Vec<Fr>; 3],
    /// The circuit's matrices are already resident on the GPU: `enforce` returns without calling its closures, so
    /// no LinearCombination is built and synthesis pays for witness generation only.
    witness_only: bool,
}

impl RecordingAssignment {
    pub fn new() -> Self { Self::with_mode(false) }
    pub fn with_mode(witness_only: bool) -> Self {
        RecordingAssignment { z_in: vec![], z_aux: vec![], row_ptr: [vec![0], vec![0], vec![0]], col: Default::default(), coeff: Default::default(), witness_only }
    }
    fn push(&mut self, k: usize, lc: &LinearCombination<Bn256>) {
        for (var, coeff) in lc.as_ref() {
            let c = match var.get_unchecked() { Index::Input(i) => i as u32, Index::Aux(i) => (i as u32) | 0x8000_0000 };
            self.col[k].push(c);
            self.coeff[k].push(*coeff);
        }
        self.row_ptr[k].push(self.col[k].len() as u64);
    }
}

impl ConstraintSystem<Bn256> for RecordingAssignment {
    type Root = Self;
    fn alloc<F, A, AR>(&mut self, _: A, f: F) -> Result<Variable, SynthesisError>
    where F: FnOnce() -> Result<Fr, SynthesisError>, A: FnOnce() -> AR, AR: Into<String> {
        self.z_aux.push(f()?);
        Ok(Variable::new_unchecked(Index::Aux(self.z_aux.len() - 1)))
    }
    fn alloc_input<F, A, AR>(&mut self, _: A, f: F) -> Result<Variable, SynthesisError>
    where F: FnOnce() -> Result<Fr, SynthesisError>, A: FnOnce() -> AR, AR: Into<String> {
        self.z_in.push(f()?);
        Ok(Variable::new_unchecked(Index::Input(self.z_in.len() - 1)))
    }
    fn enforce<A, AR, LA, LB, LC>(&mut self, _: A, a: LA, b: LB, c: LC)
    where A: FnOnce() -> AR, AR: Into<String>,
          LA: FnOnce(LinearCombination<Bn256>) -> LinearCombination<Bn256>,
          LB: FnOnce(LinearCombination<Bn256>) -> LinearCombination<Bn256>,
          LC: FnOnce(LinearCombination<Bn256>) -> LinearCombination<Bn256> {
        if self.witness_only { return; }
        let (a, b, c) = (a(LinearCombination::zero()), b(LinearCombination::zero()), c(LinearCombination::zero()));
        self.push(0, &a); self.push(1, &b); self.push(2, &c);
    }
    fn push_namespace<NR, N>(&mut self, _: N) where NR: Into<String>, N: FnOnce() -> NR {}
    fn pop_namespace(&mut self) {}
    fn get_root(&mut self) -> &mut Self::Root { self }
}

fn g1(p: &G1Affine) -> [u64; 8] { if p.is_zero() { [0; 8] } else { let (x, y) = p.into_xy_unchecked(); unsafe { std::mem::transmute([x, y]) } } }
fn g2(p: &G2Affine) -> [u64; 16] { if p.is_zero() { [0; 16] } else { let (x, y) = p.into_xy_unchecked(); unsafe { std::mem::transmute([x, y]) } } }

fn status_to_error(st: c_int) -> SynthesisError {
    match st {
        1 => SynthesisError::PolynomialDegreeTooLarge,
        2 => SynthesisError::UnexpectedIdentity,
        3 => SynthesisError::AssignmentMissing,
        4 => SynthesisError::MalformedVerifyingKey,
        _ => SynthesisError::IoError(std::io::Error::new(std::io::ErrorKind::Other, "zinc_groth16_hip device error")),
    }
}

/// Host-side memo of converted keys: `Parameters<Bn256>` keeps its vectors behind `Arc`s, so `params.clone()`
/// (zinc-tester/src/runners/proof_check.rs:122) shares them and the `Arc` addresses identify the key.  The entry
/// holds clones of the `Arc`s, which keeps those addresses from being reused while it is cached.
struct KeyMemo {
    h: std::sync::Arc<Vec<G1Affine>>, l: std::sync::Arc<Vec<G1Affine>>, a: std::sync::Arc<Vec<G1Affine>>,
    b_g1: std::sync::Arc<Vec<G1Affine>>, b_g2: std::sync::Arc<Vec<G2Affine>>,
    pk: usize, // *mut zg16_pk owned by the library's cache
    id: u64,   // zg16_pk_id(pk) when it was handed out: the address alone may since belong to another key
}
lazy_static::lazy_static! {
    /// One memo for the whole process (zinc-tester proves on a rayon pool: a per-thread memo would convert every
    /// key once per thread, and a thread's stale entry could outlive the key it names).  Only touched under
    /// `PROVE_LOCK`.  `lazy_static!` because the reference builds with rust:1.47 (docker/Dockerfile:1), where
    /// `Mutex::new` is not yet a const fn; zinc-vm already depends on the crate.
    static ref KEY_MEMO: std::sync::Mutex<Vec<KeyMemo>> = std::sync::Mutex::new(Vec::new());
    static ref PROVE_LOCK: std::sync::Mutex<()> = std::sync::Mutex::new(());
}

/// The resident handle of `params` on `ctx`: converted and handed to `zg16_pk_cached` the first time this set of
/// `Arc`s is seen, looked up by address afterwards (no conversion, no upload).
unsafe fn resident_key(ctx: *mut zg16_ctx, params: &Parameters<Bn256>) -> Result<*mut zg16_pk, c_int> {
    use std::sync::Arc;
    let mut memo = KEY_MEMO.lock().unwrap_or_else(|poisoned| poisoned.into_inner());
    let known = memo.iter().find(|e|
        Arc::ptr_eq(&e.h, &params.h) && Arc::ptr_eq(&e.l, &params.l) && Arc::ptr_eq(&e.a, &params.a) &&
        Arc::ptr_eq(&e.b_g1, &params.b_g1) && Arc::ptr_eq(&e.b_g2, &params.b_g2)).map(|e| (e.pk, e.id));
    if let Some((pk, id)) = known {
        // the library may have evicted the key since, and handed its address to a newer one: only a handle it
        // still holds under the same serial number is used without a view
        if zg16_pk_cache_holds(ctx, pk as *const zg16_pk, id) != 0 { return Ok(pk as *mut zg16_pk); }
    }
    // entries whose key the library no longer holds are of no use to anyone
    memo.retain(|e| zg16_pk_cache_holds(ctx, e.pk as *const zg16_pk, e.id) != 0);
    let conv1 = |v: &Vec<G1Affine>| v.iter().map(g1).collect::<Vec<_>>();
    let (h, l, a, b1) = (conv1(&params.h), conv1(&params.l), conv1(&params.a), conv1(&params.b_g1));
    let b2: Vec<[u64; 16]> = params.b_g2.iter().map(g2).collect();
    let (al, be1, de1) = (g1(&params.vk.alpha_g1), g1(&params.vk.beta_g1), g1(&params.vk.delta_g1));
    let (be2, de2) = (g2(&params.vk.beta_g2), g2(&params.vk.delta_g2));
    let pkv = zg16_pk_view { alpha_g1: al.as_ptr(), beta_g1: be1.as_ptr(), beta_g2: be2.as_ptr(), delta_g1: de1.as_ptr(), delta_g2: de2.as_ptr(),
        h: h.as_ptr() as *const u64, n_h: h.len(), l: l.as_ptr() as *const u64, n_l: l.len(), a: a.as_ptr() as *const u64, n_a: a.len(),
        b_g1: b1.as_ptr() as *const u64, n_b_g1: b1.len(), b_g2: b2.as_ptr() as *const u64, n_b_g2: b2.len() };
    let mut pk = std::ptr::null_mut();
    let mut hit: c_int = 0;
    let st = zg16_pk_cached(ctx, &pkv, 0, &mut pk, &mut hit);
    if st != 0 { return Err(st); }
    memo.retain(|e| e.pk != pk as usize);
    memo.push(KeyMemo { h: params.h.clone(), l: params.l.clone(), a: params.a.clone(), b_g1: params.b_g1.clone(), b_g2: params.b_g2.clone(),
                        pk: pk as usize, id: zg16_pk_id(pk) });
    if memo.len() > 8 { memo.remove(0); }
    Ok(pk)
}

/// Same signature shape and semantics as bellman's `create_proof`, plus `circuit_id`: bytes that name the circuit
/// (e.g. a digest of the bytecode and the method name; its structure does not depend on the inputs).  Once the
/// library holds the matrices of a named circuit, synthesis runs in witness-only mode.
pub fn create_proof_named<C: Circuit<Bn256>>(circuit: C, params: &Parameters<Bn256>, r: Fr, s: Fr, circuit_id: Option<&[u8]>) -> Result<Proof<Bn256>, SynthesisError> {
    // The default context, its caches and the handles they hand out are shared by every thread of the process
    // (zinc-tester runs its cases on a rayon pool): one proof at a time goes through here.
    let _one_proof_at_a_time = PROVE_LOCK.lock().unwrap_or_else(|poisoned| poisoned.into_inner());
    unsafe {
        let mut ctx = std::ptr::null_mut();
        let st = zg16_default_ctx(0, &mut ctx);
        if st != 0 { return Err(status_to_error(st)); }
        let mut mat: *mut zg16_r1cs = std::ptr::null_mut();
        let mut hit: c_int = 0;
        if let Some(id) = circuit_id {
            let st = zg16_r1cs_cached(ctx, id.as_ptr(), id.len(), std::ptr::null(), &mut mat, &mut hit);
            if st != 0 { return Err(status_to_error(st)); }
        }
        let mut cs = RecordingAssignment::with_mode(!mat.is_null());
        cs.alloc_input(|| "", || Ok(Fr::one()))?;
        circuit.synthesize(&mut cs)?;
        for i in 0..cs.z_in.len() {
            cs.enforce(|| "", |lc| lc + Variable::new_unchecked(Index::Input(i)), |lc| lc, |lc| lc);
        }
        let pk = resident_key(ctx, params).map_err(status_to_error)?;
        if mat.is_null() {
            let csr = |k: usize| zg16_csr { row_ptr: cs.row_ptr[k].as_ptr(), col: cs.col[k].as_ptr(), coeff: cs.coeff[k].as_ptr() as *const u64 };
            let rv = zg16_r1cs_view { n_rows: cs.row_ptr[0].len() - 1, n_inputs: cs.z_in.len(), n_aux: cs.z_aux.len(), a: csr(0), b: csr(1), c: csr(2) };
            let (idp, idl) = match circuit_id { Some(id) => (id.as_ptr(), id.len()), None => (std::ptr::null(), 0) };
            let st = zg16_r1cs_cached(ctx, idp, idl, &rv, &mut mat, &mut hit);
            if st != 0 { return Err(status_to_error(st)); }
        }
        // zg16_prove reads as many witness elements as the resident circuit has variables: an id reused for a
        // circuit of another shape fails here instead of reading past the witness buffers
        let (mut n_in, mut n_aux) = (0usize, 0usize);
        let st = zg16_r1cs_shape(mat, std::ptr::null_mut(), &mut n_in, &mut n_aux);
        if st != 0 { return Err(status_to_error(st)); }
        if n_in != cs.z_in.len() || n_aux != cs.z_aux.len() { return Err(SynthesisError::AssignmentMissing); }
        let mut out = std::mem::zeroed::<zg16_proof>();
        let st = zg16_prove(ctx, pk, mat, cs.z_in.as_ptr() as *const c_void, cs.z_aux.as_ptr() as *const c_void, 0,
                            &r as *const Fr as *const u64, &s as *const Fr as *const u64, &mut out);
        if st != 0 { return Err(status_to_error(st)); }
        // Coordinates are Montgomery limbs in ff_ce's own layout (`Fq` is four u64, `Fq2` is { c0, c1 }), so the
        // sized reinterpretations below are exact: [u64; 8] = [Fq; 2] = (x, y), [u64; 16] = [Fq2; 2].
        let [ax, ay]: [Fq; 2] = std::mem::transmute(out.a);
        let [bx, by]: [Fq2; 2] = std::mem::transmute(out.b);
        let [cx, cy]: [Fq; 2] = std::mem::transmute(out.c);
        Ok(Proof { a: G1Affine::from_xy_unchecked(ax, ay), b: G2Affine::from_xy_unchecked(bx, by), c: G1Affine::from_xy_unchecked(cx, cy) })
    }
}

/// bellman's `create_proof(circuit, params, r, s)`.
pub fn create_proof<C: Circuit<Bn256>>(circuit: C, params: &Parameters<Bn256>, r: Fr, s: Fr) -> Result<Proof<Bn256>, SynthesisError> {
    create_proof_named(circuit, params, r, s, None)
}

/// Drop-in for `groth16::create_random_proof` (r first, then s, both `rng.gen::<Fr>()`).
pub fn create_random_proof<C: Circuit<Bn256>, R: Rng>(circuit: C, params: &Parameters<Bn256>, rng: &mut R) -> Result<Proof<Bn256>, SynthesisError> {
    let r: Fr = rng.gen();
    let s: Fr = rng.gen();
    create_proof(circuit, params, r, s)
}
