//! Functionality for string replacement.  (`/root/reference/src/ciphertext/replace.rs`)

use super::{op1, FheString, FheUsize};
use crate::server_key::ServerKey;

impl FheString {
    /// Returns `self` where `p` is replaced by `s` up to length `l`.  (replace.rs:17-20)
    pub fn replace(&self, k: &ServerKey, p: &FheString, s: &FheString, l: usize) -> FheString {
        FheString::wrap(op1(k, "replace", &[&self.ct, &p.ct, &s.ct], &[l as i64]), k)
    }

    /// Returns `self` where `p` is replaced by `s` up to length `l`, at most `n_max` times.  (replace.rs:23-33)
    pub fn replacen(
        &self,
        k: &ServerKey,
        p: &FheString,
        s: &FheString,
        n_max: &FheUsize,
        l: usize,
    ) -> FheString {
        FheString::wrap(op1(k, "replacen", &[&self.ct, &p.ct, &s.ct, &n_max.0], &[l as i64]), k)
    }
}
