//! Functionality for logical operations.  (`/root/reference/src/ciphertext/logic.rs`)
//!
//! In the reference these helpers are the glue between tfhe-rs calls; inside the engine they are fused into the compiled
//! circuits of the string operations.  They are kept as functions of their own (each one small circuit) because the
//! reference's crate has them, and because code written against it may call them.

use super::{op1, BooleanBlock, Ct, FheUsize};
use crate::server_key::ServerKey;

// Returns `a ? b : 0`.  (logic.rs:13-16; the reference is generic over `IntegerRadixCiphertext`, the engine's
// circuit is compiled for FheUsize-sized operands)
pub fn if_then_else_zero(k: &ServerKey, a: &BooleanBlock, b: &FheUsize) -> FheUsize {
    FheUsize(op1(k, "if_then_else_zero", &[&a.0, &b.0], &[]))
}

// Returns `a ? b : 0`, where `b` is a scalar.  (logic.rs:19-25)
pub fn scalar_if_then_else_zero<Scalar: Into<u64>>(k: &ServerKey, a: &BooleanBlock, b: Scalar) -> FheUsize {
    FheUsize(op1(k, "scalar_if_then_else_zero", &[&a.0], &[b.into() as i64]))
}

// Returns `a ? b : c`.  (logic.rs:28-38)
pub fn if_then_else_bool(
    k: &ServerKey,
    a: &BooleanBlock,
    b: &BooleanBlock,
    c: &BooleanBlock,
) -> BooleanBlock {
    BooleanBlock(op1(k, "if_then_else_bool", &[&a.0, &b.0, &c.0], &[]))
}

// Returns true if any of the elements of `v` is true.  (logic.rs:41-51)
pub fn any(k: &ServerKey, v: &[BooleanBlock]) -> BooleanBlock {
    let parts: Vec<&Ct> = v.iter().map(|b| &b.0).collect();
    let all = Ct::concat(k, &parts);
    BooleanBlock(op1(k, "any", &[&all], &[]))
}

// Returns true if all of the elements of `v` are true.  (logic.rs:54-65)
pub fn all(k: &ServerKey, v: &[BooleanBlock]) -> BooleanBlock {
    let parts: Vec<&Ct> = v.iter().map(|b| &b.0).collect();
    let all = Ct::concat(k, &parts);
    BooleanBlock(op1(k, "all", &[&all], &[]))
}
