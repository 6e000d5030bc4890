//! Functionality for string insertion.  (`/root/reference/src/ciphertext/insert.rs`)

use super::{op1, FheString, FheUsize};
use crate::server_key::ServerKey;

impl FheString {
    /// Returns `self + s`.  (insert.rs:14-17)
    pub fn add(&self, k: &ServerKey, s: &FheString) -> FheString {
        FheString::wrap(op1(k, "add", &[&self.ct, &s.ct], &[]), k)
    }

    /// Returns `self` repeated `n` times up to a maximum length of `l`.  (insert.rs:20-45)
    pub fn repeat(&self, k: &ServerKey, n: &FheUsize, l: usize) -> FheString {
        FheString::wrap(op1(k, "repeat", &[&self.ct, &n.0], &[l as i64]), k)
    }

    /// Returns a copy of `self` where `s` is inserted at the given index.
    ///
    /// # Panics
    /// Panics on index out of bounds.  (insert.rs:48-124)
    pub fn insert(&self, k: &ServerKey, index: &FheUsize, s: &FheString) -> FheString {
        FheString::wrap(op1(k, "insert", &[&self.ct, &index.0, &s.ct], &[]), k)
    }
}
