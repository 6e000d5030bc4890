//! This module covers the ciphertext types and their operations.  (`/root/reference/src/ciphertext/mod.rs`)
//!
//! The reference represents an `FheString` as `Vec<FheAsciiChar>` of tfhe-rs radix ciphertexts; here every value is
//! ONE engine object (`fhes_ct`: a range of LWE ciphertexts, host- or device-resident) and every operation is one
//! compiled circuit run on the GPU (`fhes_string_op`).  Public names, arguments and results are the reference's.

use std::error::Error;

use crate::client_key::{ClientKey, Key};
use crate::server_key::ServerKey;
use crate::{ffi, last_error};

mod compare;
mod convert;
mod insert;
pub mod logic;
mod replace;
mod search;
pub mod split;
mod trim;

/// Owned handle of an engine ciphertext object.
pub(crate) struct Ct(pub(crate) *mut ffi::FhesCt);

// The reference's own signatures need its ciphertext types to be `Sync` (`index_of_unchecked<T: Sync>`, mod.rs:292; `&FheString`
// shared by rayon workers, search.rs:34-40).  The engine object changes residency behind `&self` (first use as an operand
// uploads, decrypt downloads): those changes are serialised inside the object (csrc/objects.cpp, `fhes_ct::mu`), everything
// else is read-only, so shared access from several threads is sound.
unsafe impl Send for Ct {}
unsafe impl Sync for Ct {}

impl Ct {
    pub(crate) fn num_blocks(&self) -> usize {
        unsafe { ffi::fhes_ct_num_blocks(self.0) as usize }
    }

    pub(crate) fn to_bytes(&self) -> Vec<u8> {
        let n = unsafe { ffi::fhes_ct_serialize(self.0, std::ptr::null_mut(), 0) };
        assert!(n > 0, "{}", last_error());
        let mut buf = vec![0u8; n as usize];
        let m = unsafe { ffi::fhes_ct_serialize(self.0, buf.as_mut_ptr(), n) };
        assert!(m == n, "{}", last_error());
        buf
    }

    pub(crate) fn from_bytes(data: &[u8]) -> Result<Self, Box<dyn Error>> {
        let h = unsafe { ffi::fhes_ct_deserialize(data.as_ptr(), data.len() as i64) };
        if h.is_null() {
            return Err(last_error().into());
        }
        Ok(Ct(h))
    }

    /// `count` blocks from `first` as a new object (device-side copy).
    pub(crate) fn slice(&self, k: &ServerKey, first: usize, count: usize) -> Ct {
        let h = unsafe { ffi::fhes_ct_slice(k.k, self.0, first as i32, count as i32) };
        assert!(!h.is_null(), "{}", last_error());
        Ct(h)
    }

    pub(crate) fn concat(k: &ServerKey, parts: &[&Ct]) -> Ct {
        let ptrs: Vec<*mut ffi::FhesCt> = parts.iter().map(|p| p.0).collect();
        let h = unsafe { ffi::fhes_ct_concat(k.k, ptrs.as_ptr(), ptrs.len() as i32) };
        assert!(!h.is_null(), "{}", last_error());
        Ct(h)
    }
}

impl Clone for Ct {
    /// Device-side copy (`fhes_ct_clone`): a device-resident object is duplicated in the arena, nothing crosses PCIe; the
    /// reference clones ciphertexts freely (compare.rs:155-160, mod.rs:250-263, split.rs:322-325).
    fn clone(&self) -> Self {
        let h = unsafe { ffi::fhes_ct_clone(self.0) };
        assert!(!h.is_null(), "{}", last_error());
        Ct(h)
    }
}

impl Drop for Ct {
    fn drop(&mut self) {
        unsafe { ffi::fhes_ct_destroy(self.0) }
    }
}

impl std::fmt::Debug for Ct {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "Ct({} blocks)", self.num_blocks())
    }
}

/// One compiled circuit on the device: `fhes_string_op`.  Operation names and operand order are data of the engine
/// (`fhes_op_signature`); homomorphic operations are infallible in the reference, so a failure here (no device,
/// out of memory) panics with the engine's message.
pub(crate) fn op(k: &ServerKey, name: &str, args: &[&Ct], ints: &[i64]) -> Vec<Ct> {
    let cname = std::ffi::CString::new(name).expect("operation name");
    let ptrs: Vec<*mut ffi::FhesCt> = args.iter().map(|a| a.0).collect();
    let mut outs: [*mut ffi::FhesCt; 4] = [std::ptr::null_mut(); 4];
    let mut n = 0i32;
    let rc = unsafe {
        ffi::fhes_string_op(
            k.k,
            cname.as_ptr(),
            ptrs.as_ptr(),
            ptrs.len() as i32,
            if ints.is_empty() { std::ptr::null() } else { ints.as_ptr() },
            ints.len() as i32,
            outs.as_mut_ptr(),
            outs.len() as i32,
            &mut n,
        )
    };
    assert!(rc == 0, "{name}: {}", last_error());
    outs[..n as usize].iter().map(|&h| Ct(h)).collect()
}

pub(crate) fn op1(k: &ServerKey, name: &str, args: &[&Ct], ints: &[i64]) -> Ct {
    op(k, name, args, ints).into_iter().next().expect("one result")
}

/// tfhe-rs's `BooleanBlock` as far as the reference's API shows it: one block holding 0 or 1.
#[derive(Clone, Debug)]
pub struct BooleanBlock(pub(crate) Ct);

/// tfhe-rs's `RadixCiphertext` as far as the reference's API shows it (client_key.rs:23-33).
#[derive(Clone, Debug)]
pub struct RadixCiphertext(pub(crate) Ct);

/// FheAsciiChar is a wrapper type for `RadixCiphertext`.  (mod.rs:30-31)
#[derive(Clone, Debug)]
pub struct FheAsciiChar(pub(crate) RadixCiphertext);

impl FheAsciiChar {
    // Creates an encryption of `c`.  (mod.rs:35-37)
    pub fn new(k: &ClientKey, c: u8) -> Self {
        Self(k.encrypt(c))
    }
}

/// FheString is a wrapper type for `Vec<FheAsciiChar>`. It is assumed to be
/// 0-terminated.  (mod.rs:40-43)
#[derive(Clone, Debug)]
pub struct FheString {
    pub(crate) ct: Ct,
    /// blocks per character (lib.rs:60-61): `max_len` takes no key in the reference
    pub(crate) num_blocks_char: usize,
}

impl FheString {
    /// ASCII value of the string termination character.
    #[allow(dead_code)]
    const TERMINATOR: u8 = 0;

    pub(crate) fn wrap(ct: Ct, k: &ServerKey) -> Self {
        FheString { ct, num_blocks_char: k.num_blocks_char }
    }

    /// Creates a new FheString from an ascii string using the provided key. The
    /// input string must only contain ascii characters and must not contain any
    /// zero values. The result is padded to the specified length.  (mod.rs:58-92; the five validation errors of
    /// :59-70 come back from the engine with the reference's wording)
    pub fn new(k: &ClientKey, s: &str, l: Option<usize>) -> Result<Self, Box<dyn Error>> {
        if !s.is_ascii() {
            return Err("string is not ascii".into());
        } else if s.chars().any(|x| x as u8 == Self::TERMINATOR) {
            return Err("string contains terminator character".into());
        }
        let cs = std::ffi::CString::new(s)?;
        let mut status = 0i32;
        let h = unsafe { ffi::fhes_string_encrypt(k.k, cs.as_ptr(), l.map_or(-1, |v| v as i64), &mut status) };
        if h.is_null() {
            return Err(last_error().into()); // "string length exceeds maximum length", "pad length exceeds ...", ...
        }
        Ok(FheString { ct: Ct(h), num_blocks_char: k.num_blocks_char() })
    }

    /// Similar to `FheString::new`, but only creates a trivial encryption that
    /// does not actually hide the plaintext.  (mod.rs:96-119)
    pub fn new_trivial(k: &ServerKey, s: &str) -> Result<Self, Box<dyn Error>> {
        if !s.is_ascii() {
            return Err("string is not ascii".into());
        } else if s.chars().any(|x| x as u8 == Self::TERMINATOR) {
            return Err("string contains terminator character".into());
        }
        let cs = std::ffi::CString::new(s)?;
        let mut status = 0i32;
        let h = unsafe { ffi::fhes_string_trivial(k.k, cs.as_ptr(), &mut status) };
        if h.is_null() {
            return Err(last_error().into());
        }
        Ok(FheString::wrap(Ct(h), k))
    }

    /// (mod.rs:121-129)
    pub fn decrypt(&self, k: &ClientKey) -> String {
        let mut buf = vec![0u8; self.ct.num_blocks() + 8];
        let rc = unsafe { ffi::fhes_string_decrypt(k.k, self.ct.0, buf.as_mut_ptr() as *mut _, buf.len() as i64) };
        assert!(rc == 0, "{}", last_error());
        let end = buf.iter().position(|&b| b == 0).unwrap_or(buf.len());
        String::from_utf8(buf[..end].to_vec()).unwrap()
    }

    /// Returns the length of `self`.  (mod.rs:132-156)
    pub fn len(&self, k: &ServerKey) -> FheUsize {
        FheUsize(op1(k, "len", &[&self.ct], &[]))
    }

    /// Returns an upper bound on the length of `self`.  (mod.rs:159-162)
    pub fn max_len(&self) -> usize {
        // Substract 1 because strings are 0-terminated.
        self.ct.num_blocks() / self.num_blocks_char - 1
    }

    /// Returns `self[..index]`.  (mod.rs:165-180)
    pub fn substr_to(&self, k: &ServerKey, index: &FheUsize) -> FheString {
        FheString::wrap(op1(k, "substr_to", &[&self.ct, &index.0], &[]), k)
    }

    /// Returns `self[index..]`.  (mod.rs:183-195)
    pub fn substr_from(&self, k: &ServerKey, index: &FheUsize) -> FheString {
        FheString::wrap(op1(k, "substr_from", &[&self.ct, &index.0], &[]), k)
    }

    /// Returns `self[start..end]`.  (mod.rs:198-213)
    pub fn substr_end(&self, k: &ServerKey, start: &FheUsize, end: &FheUsize) -> FheString {
        FheString::wrap(op1(k, "substr_end", &[&self.ct, &start.0, &end.0], &[]), k)
    }

    /// Returns the character at the given index. Returns 0 if the index is out
    /// of bounds.  (mod.rs:217-237)
    pub fn char_at(&self, k: &ServerKey, i: &FheUsize) -> FheAsciiChar {
        FheAsciiChar(RadixCiphertext(op1(k, "char_at", &[&self.ct, &i.0], &[])))
    }

    /// Returns the maximum length of an FheString when using key `k`.  (mod.rs:240-244)
    pub fn max_len_with_key<K: Key>(k: &K) -> usize {
        // We can only handle strings for which the length can be represented by
        // an FheUsize.
        FheUsize::max(k)
    }

    /// The characters of `self` (terminator and padding included), each as an object of its own.
    pub(crate) fn chars(&self, k: &ServerKey) -> Vec<FheAsciiChar> {
        let bc = self.num_blocks_char;
        (0..self.ct.num_blocks() / bc)
            .map(|i| FheAsciiChar(RadixCiphertext(self.ct.slice(k, i * bc, bc))))
            .collect()
    }

    /// serde (mod.rs:42 derives it): u32 blocks-per-character + the ciphertext blob of the C ABI.
    pub fn to_bytes(&self) -> Vec<u8> {
        let mut v = (self.num_blocks_char as u32).to_le_bytes().to_vec();
        v.extend_from_slice(&self.ct.to_bytes());
        v
    }

    pub fn from_bytes(data: &[u8]) -> Result<Self, Box<dyn Error>> {
        if data.len() < 4 {
            return Err("truncated FheString blob".into());
        }
        let nbc = u32::from_le_bytes([data[0], data[1], data[2], data[3]]) as usize;
        let ct = Ct::from_bytes(&data[4..])?;
        if nbc == 0 || ct.num_blocks() % nbc != 0 || ct.num_blocks() == 0 {
            return Err("FheString blob: block count is not a multiple of the character width".into());
        }
        Ok(FheString { ct, num_blocks_char: nbc })
    }
}

/// Given `v` and `Enc(i)`, return `v[i]`. Returns `0` if `i` is out of bounds.  (mod.rs:271-287; the engine's
/// `"element_at_bool"` circuit over one BooleanBlock per element)
pub fn element_at_bool(k: &ServerKey, v: &[BooleanBlock], i: &FheUsize) -> BooleanBlock {
    let parts: Vec<&Ct> = v.iter().map(|b| &b.0).collect();
    let all = Ct::concat(k, &parts);
    BooleanBlock(op1(k, "element_at_bool", &[&all, &i.0], &[]))
}

/// Searches `v` for the first index `i` with `p(v[i]) == 1`.
///
/// Expects that `p` returns an encryption of either 0 or 1.  (mod.rs:292-298)
pub fn index_of_unchecked<T: Sync>(
    k: &ServerKey,
    v: &[T],
    p: fn(&ServerKey, &T) -> BooleanBlock,
) -> FheOption<FheUsize> {
    index_of_unchecked_with_options(k, v, p, false)
}

/// Searches `v` for the last index `i` with `p(v[i]) == 1`.
///
/// Expects that `p` returns an encryption of either 0 or 1.  (mod.rs:303-309)
pub fn rindex_of_unchecked<T: Sync>(
    k: &ServerKey,
    v: &[T],
    p: fn(&ServerKey, &T) -> BooleanBlock,
) -> FheOption<FheUsize> {
    index_of_unchecked_with_options(k, v, p, true)
}

/// (mod.rs:315-355) The closure runs on the caller's side (a Rust closure cannot cross the C boundary); the index
/// scan over its results -- `pi * i`, the `index = b ? index : pi_mul_i` / `b |= pi` chain -- is the engine's
/// `"find_pred"` / `"rfind_pred"` circuit over one BooleanBlock per element.
fn index_of_unchecked_with_options<T: Sync>(
    k: &ServerKey,
    v: &[T],
    p: fn(&ServerKey, &T) -> BooleanBlock,
    reverse: bool,
) -> FheOption<FheUsize> {
    let flags: Vec<BooleanBlock> = v.iter().map(|x| p(k, x)).collect();
    let parts: Vec<&Ct> = flags.iter().map(|b| &b.0).collect();
    let all = Ct::concat(k, &parts);
    let mut o = op(k, if reverse { "rfind_pred" } else { "find_pred" }, &[&all], &[]).into_iter();
    FheOption {
        is_some: BooleanBlock(o.next().expect("is_some")),
        val: FheUsize(o.next().expect("val")),
    }
}

/// FheOption represents an encrypted option type.  (mod.rs:358-363)
pub struct FheOption<T> {
    /// Whether this option decrypts to `Some` or `None`.
    pub is_some: BooleanBlock,
    /// The optional value.
    pub val: T,
}

impl FheOption<RadixCiphertext> {
    pub fn decrypt(&self, k: &ClientKey) -> Option<u64> {
        let is_some = k.decrypt_bool(&self.is_some);
        match is_some {
            true => {
                let val = k.decrypt::<u64>(&self.val);
                Some(val)
            }
            false => None,
        }
    }
}

impl FheOption<FheUsize> {
    pub fn decrypt(&self, k: &ClientKey) -> Option<usize> {
        let is_some = k.decrypt_bool(&self.is_some);
        match is_some {
            true => Some(k.decrypt_usize(&self.val)),
            false => None,
        }
    }
}

impl FheOption<FheString> {
    pub fn decrypt(&self, k: &ClientKey) -> Option<String> {
        let is_some = k.decrypt_bool(&self.is_some);
        match is_some {
            true => {
                let val = self.val.decrypt(k);
                Some(val)
            }
            false => None,
        }
    }
}

impl FheOption<(FheString, FheString)> {
    /// `split_once` / `rsplit_once` (split.rs:505-513): the reference's callers decrypt the two halves themselves
    /// (examples/cmd/main.rs); provided here for symmetry with the other instantiations.
    pub fn decrypt(&self, k: &ClientKey) -> Option<(String, String)> {
        match k.decrypt_bool(&self.is_some) {
            true => Some((self.val.0.decrypt(k), self.val.1.decrypt(k))),
            false => None,
        }
    }
}

/// (mod.rs:401-402)
#[derive(Clone, Debug)]
pub struct FheUsize(pub(crate) Ct);

impl FheUsize {
    /// Creates a new encrypted usize with value `v`.  (mod.rs:406-423: encrypt as num_blocks_char blocks, then extend
    /// with trivial zero blocks or trim to num_blocks_usize -- `fhes_usize_encrypt` does the same on the host,
    /// without the reference's per-call server-keygen)
    pub fn new(k: &ClientKey, v: usize) -> Self {
        let h = unsafe { ffi::fhes_usize_encrypt(k.k, v as u64) };
        assert!(!h.is_null(), "{}", last_error());
        FheUsize(Ct(h))
    }

    /// Creates a new trivial usize ciphertext with value `v`.  (mod.rs:426-429)
    pub fn new_trivial(k: &ServerKey, v: usize) -> FheUsize {
        let h = unsafe { ffi::fhes_usize_trivial(k.k, v as u64) };
        assert!(!h.is_null(), "{}", last_error());
        FheUsize(Ct(h))
    }

    /// Creates a new trivial usize ciphertext with value `b ? 1 : 0`.  (mod.rs:431-433)
    pub fn new_from_bool(k: &ServerKey, b: &BooleanBlock) -> FheUsize {
        FheUsize(op1(k, "usize_from_bool", &[&b.0], &[]))
    }

    /// (mod.rs:435-437)
    pub fn max<K: Key>(k: &K) -> usize {
        k.msg_mod().pow(k.num_blocks_usize() as u32) - 1
    }

    /// (mod.rs:439-441)
    pub fn decrypt(&self, k: &ClientKey) -> usize {
        k.decrypt_usize(self)
    }

    pub fn to_bytes(&self) -> Vec<u8> {
        self.0.to_bytes()
    }

    pub fn from_bytes(data: &[u8]) -> Result<Self, Box<dyn Error>> {
        Ok(FheUsize(Ct::from_bytes(data)?))
    }
}

#[cfg(feature = "serde")]
mod serde_impls {
    use super::*;
    macro_rules! blob_serde {
        ($t:ty) => {
            impl serde::Serialize for $t {
                fn serialize<S: serde::Serializer>(&self, s: S) -> Result<S::Ok, S::Error> {
                    s.serialize_bytes(&self.to_bytes())
                }
            }
            impl<'de> serde::Deserialize<'de> for $t {
                fn deserialize<D: serde::Deserializer<'de>>(d: D) -> Result<Self, D::Error> {
                    let bytes: Vec<u8> = serde::Deserialize::deserialize(d)?;
                    <$t>::from_bytes(&bytes).map_err(serde::de::Error::custom)
                }
            }
        };
    }
    blob_serde!(FheString);
    blob_serde!(FheUsize);
    impl FheAsciiChar {
        pub fn to_bytes(&self) -> Vec<u8> {
            self.0 .0.to_bytes()
        }
        pub fn from_bytes(data: &[u8]) -> Result<Self, Box<dyn Error>> {
            Ok(FheAsciiChar(RadixCiphertext(Ct::from_bytes(data)?)))
        }
    }
    blob_serde!(FheAsciiChar);
}
