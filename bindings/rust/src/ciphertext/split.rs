//! Functionality for string splitting.  (`/root/reference/src/ciphertext/split.rs`)

use super::{op, op1, BooleanBlock, Ct, FheOption, FheString, FheUsize};
use crate::client_key::ClientKey;
use crate::server_key::ServerKey;

/// An encrypted vector of substrings of an encrypted reference string.  (split.rs:21-28)
///
/// The reference keeps `Vec<FheOption<FheStringSlice{start, end}>>`; the engine returns the same data as three objects:
/// `is_some[n]`, `start[n][num_blocks_usize]`, `end[n][num_blocks_usize]`.
pub struct FheStringSliceVector {
    /// The reference string.
    s: FheString,
    is_some: Ct,
    start: Ct,
    end: Ct,
    num_blocks_usize: usize,
    /// `reverse()` takes no key in the reference (split.rs:229-231); the order is applied at the next use
    reversed: bool,
}

impl FheStringSliceVector {
    fn from_parts(s: &FheString, k: &ServerKey, mut o: std::vec::IntoIter<Ct>) -> Self {
        FheStringSliceVector {
            s: s.clone(),
            is_some: o.next().expect("is_some"),
            start: o.next().expect("start"),
            end: o.next().expect("end"),
            num_blocks_usize: k.num_blocks_usize,
            reversed: false,
        }
    }

    /// runs `f` on the three objects in their current element order
    fn with_parts<R>(&self, k: &ServerKey, f: impl FnOnce(&Ct, &Ct, &Ct) -> R) -> R {
        if self.reversed {
            let mut o = op(k, "slices_reverse", &[&self.is_some, &self.start, &self.end], &[]).into_iter();
            let (a, b, c) = (o.next().expect("is_some"), o.next().expect("start"), o.next().expect("end"));
            f(&a, &b, &c)
        } else {
            f(&self.is_some, &self.start, &self.end)
        }
    }

    /// Returns the number of substrings contained in this vector.  (split.rs:32-43)
    pub fn len(&self, k: &ServerKey) -> FheUsize {
        FheUsize(op1(k, "slices_len", &[&self.is_some, &self.start, &self.end], &[]))
    }

    /// Returns the substring stored at index `i`, if existent.  (split.rs:46-83, as written there: DESIGN.md 5.7)
    pub fn get(&self, k: &ServerKey, i: &FheUsize) -> FheOption<FheString> {
        let mut o = self.with_parts(k, |a, b, c| op(k, "slices_get", &[a, b, c, &self.s.ct, &i.0], &[])).into_iter();
        FheOption {
            is_some: BooleanBlock(o.next().expect("is_some")),
            val: FheString::wrap(o.next().expect("val"), k),
        }
    }

    /// Truncate the vector to the first `i` elements.  (split.rs:86-113)
    pub fn truncate(&mut self, k: &ServerKey, i: &FheUsize) {
        let mut o = self.with_parts(k, |a, b, c| op(k, "slices_truncate", &[a, b, c, &i.0], &[])).into_iter();
        self.is_some = o.next().expect("is_some");
        self.start = o.next().expect("start");
        self.end = o.next().expect("end");
        self.reversed = false;
    }

    /// Decrypts this vector.  (split.rs:208-226)
    pub fn decrypt(&self, k: &ClientKey) -> Vec<String> {
        let s_dec = self.s.decrypt(k);
        let n = self.is_some.num_blocks();
        let bu = self.num_blocks_usize;
        let some = decrypt_blocks(k, &self.is_some);
        let st = decrypt_blocks(k, &self.start);
        let en = decrypt_blocks(k, &self.end);
        let msg_bits = k.params().message_bits as usize;
        let msg_mask = (1usize << msg_bits) - 1;
        // little-endian digits of `msg_bits` bits each (mod.rs:439-441: RadixClientKey::decrypt)
        let radix = |digits: &[u8]| -> usize {
            let mut v = 0usize;
            for (j, &d) in digits.iter().enumerate() {
                v += (d as usize & msg_mask) << (msg_bits * j);
            }
            v
        };
        let mut out = (0..n)
            .filter_map(|i| match some[i] & 1 {
                0 => None,
                _ => {
                    let start = radix(&st[i * bu..(i + 1) * bu]);
                    let end = radix(&en[i * bu..(i + 1) * bu]);
                    let slice = s_dec.get(start..end).unwrap_or_default();
                    log::trace!("decrypt slice: [{start}, {end}]");
                    Some(slice.to_string())
                }
            })
            .collect::<Vec<_>>();
        if self.reversed {
            out.reverse();
        }
        out
    }

    /// Reverses the order of the elements.  (split.rs:229-231)
    pub fn reverse(&mut self) {
        self.reversed = !self.reversed;
    }
}

fn decrypt_blocks(k: &ClientKey, c: &Ct) -> Vec<u8> {
    let mut v = vec![0u8; c.num_blocks()];
    let rc = unsafe { crate::ffi::fhes_ct_decrypt_blocks(k.k, c.0, v.as_mut_ptr()) };
    assert!(rc == 0, "{}", crate::last_error());
    v
}

impl FheString {
    fn split_family(&self, k: &ServerKey, name: &str, p: &FheString) -> FheStringSliceVector {
        FheStringSliceVector::from_parts(self, k, op(k, name, &[&self.ct, &p.ct], &[]).into_iter())
    }

    /// Splits `self` at each occurrence of `p` into a vector of substrings.
    ///
    /// # Limitations
    /// If p.len == 0, the result is undefined.  (split.rs:339-341; the engine reproduces what the reference's tests
    /// pin for the empty pattern, tests/split.rs:33-47)
    pub fn split(&self, k: &ServerKey, p: &FheString) -> FheStringSliceVector {
        self.split_family(k, "split", p)
    }

    /// Splits `self` at each occurrence of `p` into a vector of substrings and
    /// returns the elements in reverse order.  (split.rs:348-352)
    pub fn rsplit(&self, k: &ServerKey, p: &FheString) -> FheStringSliceVector {
        self.split_family(k, "rsplit", p)
    }

    /// Splits `self` at each occurrence of `p` into a vector of substrings
    /// where the pattern is included at the end of each substring.  (split.rs:359-361)
    pub fn split_inclusive(&self, k: &ServerKey, p: &FheString) -> FheStringSliceVector {
        self.split_family(k, "split_inclusive", p)
    }

    /// Splits `self` at each occurrence of `p` into a vector of substrings of
    /// at most length `n`.  (split.rs:368-373)
    pub fn splitn(&self, k: &ServerKey, n: &FheUsize, p: &FheString) -> FheStringSliceVector {
        FheStringSliceVector::from_parts(self, k, op(k, "splitn", &[&self.ct, &n.0, &p.ct], &[]).into_iter())
    }

    /// Splits `self` at each occurrence of `p` into a vector of substrings of
    /// at most length `n` in reverse order.  (split.rs:380-387)
    pub fn rsplitn(&self, k: &ServerKey, n: &FheUsize, p: &FheString) -> FheStringSliceVector {
        FheStringSliceVector::from_parts(self, k, op(k, "rsplitn", &[&self.ct, &n.0, &p.ct], &[]).into_iter())
    }

    /// Splits `self` at each occurrence of `p` into a vector of substrings
    /// where the last substring is skipped if empty.  (split.rs:394-398)
    pub fn split_terminator(&self, k: &ServerKey, p: &FheString) -> FheStringSliceVector {
        self.split_family(k, "split_terminator", p)
    }

    /// Splits `self` at each occurrence of `p` into a vector of substrings in
    /// reverse order where the last substring is skipped if empty.  (split.rs:405-411)
    pub fn rsplit_terminator(&self, k: &ServerKey, p: &FheString) -> FheStringSliceVector {
        self.split_family(k, "rsplit_terminator", p)
    }

    /// Splits `self` at each occurrence of ascii whitespace into a vector of
    /// substrings.  (split.rs:415-473)
    pub fn split_ascii_whitespace(&self, k: &ServerKey) -> FheStringSliceVector {
        FheStringSliceVector::from_parts(self, k, op(k, "split_ascii_whitespace", &[&self.ct], &[]).into_iter())
    }

    fn split_once_opt(&self, k: &ServerKey, p: &FheString, reverse: bool) -> FheOption<(FheString, FheString)> {
        let mut o = op(k, if reverse { "rsplit_once" } else { "split_once" }, &[&self.ct, &p.ct], &[]).into_iter();
        FheOption {
            is_some: BooleanBlock(o.next().expect("is_some")),
            val: (FheString::wrap(o.next().expect("prefix"), k), FheString::wrap(o.next().expect("suffix"), k)),
        }
    }

    /// Splits `self` on the first occurrence of the specified delimiter and
    /// returns prefix before delimiter and suffix after delimiter.  (split.rs:505-507)
    pub fn split_once(&self, k: &ServerKey, p: &FheString) -> FheOption<(FheString, FheString)> {
        self.split_once_opt(k, p, false)
    }

    /// Splits `self` on the last occurrence of the specified delimiter and
    /// returns prefix before delimiter and suffix after delimiter.  (split.rs:511-513)
    pub fn rsplit_once(&self, k: &ServerKey, p: &FheString) -> FheOption<(FheString, FheString)> {
        self.split_once_opt(k, p, true)
    }
}
