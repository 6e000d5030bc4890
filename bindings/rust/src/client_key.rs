//! This module covers the client key that is used for encryption and
//! decryption.  (`/root/reference/src/client_key.rs`)

use crate::ciphertext::{BooleanBlock, Ct, FheUsize, RadixCiphertext};
use crate::{ffi, last_error, ClassicPBSParameters};

/// A key used by the client for string encryption and decryption.  (client_key.rs:15-19)
///
/// Key generation, encryption and decryption are host code in the reference and stay host code in the engine
/// (`csrc/client.cpp`); secret keys and encryption randomness come from the OS (`getrandom`).
pub struct ClientKey {
    pub(crate) k: *mut ffi::FhesClientKey,
    pub(crate) num_blocks_usize: usize,
}

// the handle is only read after generation; encryption draws independent randomness per call under its own lock
unsafe impl Send for ClientKey {}
unsafe impl Sync for ClientKey {}

impl ClientKey {
    /// `gen_keys_radix`'s client half (lib.rs:64): `fhes_client_key_generate` with a production seed of 0.
    pub(crate) fn generate(p: ClassicPBSParameters, l: usize) -> Self {
        let k = unsafe { ffi::fhes_client_key_generate(&p, l as u64, 0) };
        assert!(!k.is_null(), "{}", last_error());
        let num_blocks_usize = unsafe { ffi::fhes_client_key_num_blocks_usize(k) } as usize;
        ClientKey { k, num_blocks_usize }
    }

    /// Deterministic key for tests (seed != 0; never for production use).
    pub fn generate_seeded(p: ClassicPBSParameters, l: usize, seed: u64) -> Self {
        let k = unsafe { ffi::fhes_client_key_generate(&p, l as u64, seed) };
        assert!(!k.is_null(), "{}", last_error());
        let num_blocks_usize = unsafe { ffi::fhes_client_key_num_blocks_usize(k) } as usize;
        ClientKey { k, num_blocks_usize }
    }

    /// Encrypt a single element.  (client_key.rs:23-28)
    pub fn encrypt<T: Into<u64>>(&self, message: T) -> RadixCiphertext {
        let h = unsafe { ffi::fhes_radix_encrypt(self.k, message.into()) };
        assert!(!h.is_null(), "{}", last_error());
        RadixCiphertext(Ct(h))
    }

    /// Decrypt a single element.  (client_key.rs:31-33)
    pub fn decrypt<T: FromRadixValue>(&self, ct: &RadixCiphertext) -> T {
        T::from_radix_value(self.decrypt_u64(&ct.0))
    }

    /// Decrypt a usize.  (client_key.rs:36-38)
    pub fn decrypt_usize(&self, ct: &FheUsize) -> usize {
        self.decrypt_u64(&ct.0) as usize
    }

    /// Decrypt a boolean value.  (client_key.rs:41-43)
    pub fn decrypt_bool(&self, ct: &BooleanBlock) -> bool {
        let mut out = 0i32;
        let rc = unsafe { ffi::fhes_bool_decrypt(self.k, ct.0 .0, &mut out) };
        assert!(rc == 0, "{}", last_error());
        out != 0
    }

    pub(crate) fn decrypt_u64(&self, ct: &Ct) -> u64 {
        let mut out = 0u64;
        let rc = unsafe { ffi::fhes_radix_decrypt(self.k, ct.0, &mut out) };
        assert!(rc == 0, "{}", last_error());
        out
    }

    pub(crate) fn params(&self) -> ClassicPBSParameters {
        let mut p = crate::PARAM_MESSAGE_2_CARRY_2_KS_PBS;
        let rc = unsafe { ffi::fhes_client_key_params(self.k, &mut p) };
        assert!(rc == 0, "{}", last_error());
        p
    }

    pub(crate) fn num_blocks_char(&self) -> usize {
        unsafe { ffi::fhes_client_key_num_blocks_char(self.k) as usize }
    }

    /// serde (client_key.rs:15 derives it): the versioned blob of `fhes_client_key_serialize`.
    pub fn to_bytes(&self) -> Vec<u8> {
        let n = unsafe { ffi::fhes_client_key_serialize(self.k, std::ptr::null_mut(), 0) };
        assert!(n > 0, "{}", last_error());
        let mut buf = vec![0u8; n as usize];
        let m = unsafe { ffi::fhes_client_key_serialize(self.k, buf.as_mut_ptr(), n) };
        assert!(m == n, "{}", last_error());
        buf
    }

    pub fn from_bytes(data: &[u8]) -> Result<Self, Box<dyn std::error::Error>> {
        let k = unsafe { ffi::fhes_client_key_deserialize(data.as_ptr(), data.len() as i64) };
        if k.is_null() {
            return Err(last_error().into());
        }
        let num_blocks_usize = unsafe { ffi::fhes_client_key_num_blocks_usize(k) } as usize;
        Ok(ClientKey { k, num_blocks_usize })
    }
}

impl Clone for ClientKey {
    fn clone(&self) -> Self {
        Self::from_bytes(&self.to_bytes()).expect("client key blob round trip")
    }
}

impl std::fmt::Debug for ClientKey {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        f.debug_struct("ClientKey").field("num_blocks_usize", &self.num_blocks_usize).finish()
    }
}

impl Drop for ClientKey {
    fn drop(&mut self) {
        unsafe { ffi::fhes_client_key_destroy(self.k) }
    }
}

#[cfg(feature = "serde")]
impl serde::Serialize for ClientKey {
    fn serialize<S: serde::Serializer>(&self, s: S) -> Result<S::Ok, S::Error> {
        s.serialize_bytes(&self.to_bytes())
    }
}

#[cfg(feature = "serde")]
impl<'de> serde::Deserialize<'de> for ClientKey {
    fn deserialize<D: serde::Deserializer<'de>>(d: D) -> Result<Self, D::Error> {
        let bytes: Vec<u8> = serde::Deserialize::deserialize(d)?;
        ClientKey::from_bytes(&bytes).map_err(serde::de::Error::custom)
    }
}

/// What `ClientKey::decrypt::<T>` may produce: the unsigned integers, wrapping like tfhe-rs's recomposition.
pub trait FromRadixValue {
    fn from_radix_value(v: u64) -> Self;
}
macro_rules! from_radix_value {
    ($($t:ty),*) => { $(impl FromRadixValue for $t { fn from_radix_value(v: u64) -> Self { v as $t } })* };
}
from_radix_value!(u8, u16, u32, u64, u128, usize);

/// A trait for operations common on client key and server key.  (client_key.rs:47-52)
pub trait Key {
    /// Returns the message space modulus.
    fn msg_mod(&self) -> usize;
    /// Returns the number of blocks used to store an encrypted usize.
    fn num_blocks_usize(&self) -> usize;
}

impl Key for ClientKey {
    fn msg_mod(&self) -> usize {
        1usize << self.params().message_bits
    }

    fn num_blocks_usize(&self) -> usize {
        self.num_blocks_usize
    }
}
