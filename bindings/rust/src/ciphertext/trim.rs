//! Functionality for string trimming.  (`/root/reference/src/ciphertext/trim.rs`)

use super::{op, op1, BooleanBlock, FheAsciiChar, FheOption, FheString, FheUsize};
use crate::server_key::ServerKey;

impl FheAsciiChar {
    /// Returns whether `self` is a whitespace character (9..=13 or 32).  (trim.rs:15-25)
    pub fn is_whitespace(&self, k: &ServerKey) -> BooleanBlock {
        BooleanBlock(op1(k, "is_whitespace", &[&self.0 .0], &[]))
    }
}

impl FheString {
    /// Returns `self[i..]` where `i` is the index of the first non-whitespace
    /// character.  (trim.rs:31-40)
    pub fn trim_start(&self, k: &ServerKey) -> FheString {
        FheString::wrap(op1(k, "trim_start", &[&self.ct], &[]), k)
    }

    /// Returns `self[..i+1]` where `i` is the index of the last non-whitespace
    /// character.  (trim.rs:43-59)
    pub fn trim_end(&self, k: &ServerKey) -> FheString {
        FheString::wrap(op1(k, "trim_end", &[&self.ct], &[]), k)
    }

    /// Returns `self[i..j+1]` where `i` is the index of the first
    /// non-whitespace character and `j` is the index of the last non-whitespace
    /// character.  (trim.rs:62-85)
    pub fn trim(&self, k: &ServerKey) -> FheString {
        FheString::wrap(op1(k, "trim", &[&self.ct], &[]), k)
    }

    /// If `self` starts with `s`, returns `self[s.len..]`.  (trim.rs:89-97)
    pub fn strip_prefix(&self, k: &ServerKey, s: &FheString) -> FheOption<FheString> {
        let mut o = op(k, "strip_prefix", &[&self.ct, &s.ct], &[]).into_iter();
        FheOption {
            is_some: BooleanBlock(o.next().expect("is_some")),
            val: FheString::wrap(o.next().expect("val"), k),
        }
    }

    /// If `self` ends with `s`, returns `self[..self.len - s.len]`.  (trim.rs:101-116)
    pub fn strip_suffix(&self, k: &ServerKey, s: &FheString) -> FheOption<FheString> {
        let mut o = op(k, "strip_suffix", &[&self.ct, &s.ct], &[]).into_iter();
        FheOption {
            is_some: BooleanBlock(o.next().expect("is_some")),
            val: FheString::wrap(o.next().expect("val"), k),
        }
    }

    /// Returns `self[..index]`.  (trim.rs:119-121)
    pub fn truncate(&self, k: &ServerKey, index: &FheUsize) -> FheString {
        FheString::wrap(op1(k, "truncate", &[&self.ct, &index.0], &[]), k)
    }
}
