//! Functionality for string search.  (`/root/reference/src/ciphertext/search.rs`)

use super::{index_of_unchecked, op, op1, rindex_of_unchecked, BooleanBlock, FheAsciiChar, FheOption, FheString, FheUsize};
use crate::server_key::ServerKey;

impl FheString {
    /// Returns whether `self` contains the string `s`. The result is an
    /// encryption of 1 if this is the case and an encryption of 0 otherwise.  (search.rs:17-20)
    ///
    /// With a communicator attached to the key (`fhes_server_key_set_comm`, one process per GPU) the per-position
    /// `substr_eq` grid of `find_all` (search.rs:33-41) runs sharded over the ranks; every rank makes the same call.
    pub fn contains(&self, k: &ServerKey, s: &FheString) -> BooleanBlock {
        BooleanBlock(op1(k, "contains", &[&self.ct, &s.ct], &[]))
    }

    /// Returns the index of the first occurrence of `s`, if existent.  (search.rs:23-28)
    pub fn find(&self, k: &ServerKey, s: &FheString) -> FheOption<FheUsize> {
        let mut o = op(k, "find", &[&self.ct, &s.ct], &[]).into_iter();
        FheOption {
            is_some: BooleanBlock(o.next().expect("is_some")),
            val: FheUsize(o.next().expect("val")),
        }
    }

    /// Returns the index of the last occurence of `s`, if existent.  (search.rs:155-173)
    pub fn rfind(&self, k: &ServerKey, s: &FheString) -> FheOption<FheUsize> {
        let mut o = op(k, "rfind", &[&self.ct, &s.ct], &[]).into_iter();
        FheOption {
            is_some: BooleanBlock(o.next().expect("is_some")),
            val: FheUsize(o.next().expect("val")),
        }
    }

    /// Searches `self` for the first index `i` with `p(self[i]) == 1`.  (search.rs:197-203; the closure runs on every
    /// slot -- terminator and padding included, as `index_of_unchecked` iterates the whole `Vec` --, the index scan on
    /// the device)
    pub fn find_pred_unchecked(
        &self,
        k: &ServerKey,
        p: fn(&ServerKey, &FheAsciiChar) -> BooleanBlock,
    ) -> FheOption<FheUsize> {
        index_of_unchecked(k, &self.chars(k), p)
    }

    /// Searches `self` for the first index `i` with `p(self[i]) == 1` in
    /// reverse direction.  (search.rs:207-213)
    pub fn rfind_pred_unchecked(
        &self,
        k: &ServerKey,
        p: fn(&ServerKey, &FheAsciiChar) -> BooleanBlock,
    ) -> FheOption<FheUsize> {
        rindex_of_unchecked(k, &self.chars(k), p)
    }

    /// Returns whether `self` starts with the string `s`.  (search.rs:216-218)
    pub fn starts_with(&self, k: &ServerKey, s: &FheString) -> BooleanBlock {
        BooleanBlock(op1(k, "starts_with", &[&self.ct, &s.ct], &[]))
    }

    /// Returns whether `self` ends with the string `s`. The result is an
    /// encryption of 1 if this is the case and an encryption of 0 otherwise.  (search.rs:222-244, as written there:
    /// see DESIGN.md 5.7 for the padded-string behaviour the engine reproduces)
    pub fn ends_with(&self, k: &ServerKey, s: &FheString) -> BooleanBlock {
        BooleanBlock(op1(k, "ends_with", &[&self.ct, &s.ct], &[]))
    }
}
