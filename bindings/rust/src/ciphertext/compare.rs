//! Functionality for string comparison.  (`/root/reference/src/ciphertext/compare.rs`)

use super::{op1, BooleanBlock, FheString};
use crate::server_key::ServerKey;

impl FheString {
    /// Returns whether `self` is empty. The result is an encryption of 1 if
    /// this is the case and an encryption of 0 otherwise.  (compare.rs:15-17)
    pub fn is_empty(&self, k: &ServerKey) -> BooleanBlock {
        BooleanBlock(op1(k, "is_empty", &[&self.ct], &[]))
    }

    /// Returns `self == s`. The result is an encryption of 1 if this is the
    /// case and an encryption of 0 otherwise.  (compare.rs:21-45)
    pub fn eq(&self, k: &ServerKey, s: &FheString) -> BooleanBlock {
        BooleanBlock(op1(k, "eq", &[&self.ct, &s.ct], &[]))
    }

    /// Returns `self != s`. The result is an encryption of 1 if this is the
    /// case and an encryption of 0 otherwise.  (compare.rs:49-52)
    pub fn ne(&self, k: &ServerKey, s: &FheString) -> BooleanBlock {
        BooleanBlock(op1(k, "ne", &[&self.ct, &s.ct], &[]))
    }

    /// Returns `self <= s`. The result is an encryption of 1 if this is the
    /// case and an encryption of 0 otherwise.  (compare.rs:56-59)
    pub fn le(&self, k: &ServerKey, s: &FheString) -> BooleanBlock {
        BooleanBlock(op1(k, "le", &[&self.ct, &s.ct], &[]))
    }

    /// Returns `self < s`. The result is an encryption of 1 if this is the case
    /// and an encryption of 0 otherwise.  (compare.rs:63-99)
    pub fn lt(&self, k: &ServerKey, s: &FheString) -> BooleanBlock {
        BooleanBlock(op1(k, "lt", &[&self.ct, &s.ct], &[]))
    }

    /// Returns `self >= s`. The result is an encryption of 1 if this is the
    /// case and an encryption of 0 otherwise.  (compare.rs:102-105)
    pub fn ge(&self, k: &ServerKey, s: &FheString) -> BooleanBlock {
        BooleanBlock(op1(k, "ge", &[&self.ct, &s.ct], &[]))
    }

    /// Returns `self > s`. The result is an encryption of 1 if this is the
    /// case and an encryption of 0 otherwise.  (compare.rs:108-111)
    pub fn gt(&self, k: &ServerKey, s: &FheString) -> BooleanBlock {
        BooleanBlock(op1(k, "gt", &[&self.ct, &s.ct], &[]))
    }

    /// Returns whether `self` and `s` are equal when ignoring case.  (compare.rs:113-116)
    pub fn eq_ignore_ascii_case(&self, k: &ServerKey, s: &FheString) -> BooleanBlock {
        BooleanBlock(op1(k, "eq_ignore_ascii_case", &[&self.ct, &s.ct], &[]))
    }

    /// Returns whether `self[i..i+s.len]` and `s` are equal.  (compare.rs:119-151)
    pub fn substr_eq(&self, k: &ServerKey, i: usize, s: &FheString) -> BooleanBlock {
        BooleanBlock(op1(k, "substr_eq", &[&self.ct, &s.ct], &[i as i64]))
    }
}
