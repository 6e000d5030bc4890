//! Functionality for string conversion.  (`/root/reference/src/ciphertext/convert.rs`)

use super::{op1, BooleanBlock, FheAsciiChar, FheString, RadixCiphertext};
use crate::server_key::ServerKey;

impl FheAsciiChar {
    /// Returns whether `self` is uppercase.  (convert.rs:14-19)
    pub fn is_uppercase(&self, k: &ServerKey) -> BooleanBlock {
        BooleanBlock(op1(k, "is_uppercase", &[&self.0 .0], &[]))
    }

    /// Returns whether `self` is lowercase.  (convert.rs:22-27)
    pub fn is_lowercase(&self, k: &ServerKey) -> BooleanBlock {
        BooleanBlock(op1(k, "is_lowercase", &[&self.0 .0], &[]))
    }

    /// Returns the lowercase representation of `self`.  (convert.rs:30-36)
    pub fn to_lowercase(&self, k: &ServerKey) -> FheAsciiChar {
        FheAsciiChar(RadixCiphertext(op1(k, "char_to_lowercase", &[&self.0 .0], &[])))
    }

    /// Returns the uppercase representation of `self`.  (convert.rs:39-45)
    pub fn to_uppercase(&self, k: &ServerKey) -> FheAsciiChar {
        FheAsciiChar(RadixCiphertext(op1(k, "char_to_uppercase", &[&self.0 .0], &[])))
    }
}

impl FheString {
    /// Returns a copy of `self` where uppercase characters have been replaced
    /// by their lowercase counterparts.  (convert.rs:51-55)
    pub fn to_lowercase(&self, k: &ServerKey) -> FheString {
        FheString::wrap(op1(k, "to_lowercase", &[&self.ct], &[]), k)
    }

    /// Returns a copy of `self` where lowercase characters have been replaced
    /// by their uppercase counterparts.  (convert.rs:58-62)
    pub fn to_uppercase(&self, k: &ServerKey) -> FheString {
        FheString::wrap(op1(k, "to_uppercase", &[&self.ct], &[]), k)
    }
}
