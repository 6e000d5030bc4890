//! This module covers the server key that is used for computation.  (`/root/reference/src/server_key.rs`)

use std::fmt::Debug;

use crate::client_key::{ClientKey, Key};
use crate::{ffi, last_error, ClassicPBSParameters};

/// A key used by the server for computation on encrypted strings.  (server_key.rs:11-17)
///
/// The reference's `k: tfhe::integer::ServerKey` is the engine's device context here: bootstrapping and keyswitching
/// keys resident in HBM, compiled circuits cached per (operation, operand shapes).
pub struct ServerKey {
    pub(crate) k: *mut ffi::FhesServerKey,
    pub(crate) msg_mod: usize,
    pub(crate) num_blocks_char: usize,
    pub(crate) num_blocks_usize: usize,
}

// The reference requires `ServerKey: Sync` (nested rayon workers: search.rs:34-40, compare.rs:27-42, mod.rs:292).
// Every entry point of the C ABI may be called from any thread; concurrent fhes_string_op calls overlap on the
// device (execution lanes) and identical concurrent calls are combined into one batched run.
unsafe impl Send for ServerKey {}
unsafe impl Sync for ServerKey {}

impl ServerKey {
    /// `gen_keys_radix`'s server half (lib.rs:64): builds the bootstrapping / keyswitching keys from `ck` and uploads
    /// them to HIP device `device` (one process per GPU: `device = LOCAL_RANK`).
    pub fn new_on_device(ck: &ClientKey, device: i32) -> Self {
        let k = unsafe { ffi::fhes_server_key_create(ck.k, device) };
        assert!(!k.is_null(), "{}", last_error());
        Self::from_handle(k)
    }

    fn from_handle(k: *mut ffi::FhesServerKey) -> Self {
        let mut p = crate::PARAM_MESSAGE_2_CARRY_2_KS_PBS;
        let (mut nbc, mut nbu) = (0i32, 0i32);
        let rc = unsafe { ffi::fhes_server_key_info(k, &mut p, &mut nbc, &mut nbu) };
        assert!(rc == 0, "{}", last_error());
        ServerKey { k, msg_mod: 1usize << p.message_bits, num_blocks_char: nbc as usize, num_blocks_usize: nbu as usize }
    }

    /// Parameter set of this key.
    pub fn parameters(&self) -> ClassicPBSParameters {
        let mut p = crate::PARAM_MESSAGE_2_CARRY_2_KS_PBS;
        let rc = unsafe { ffi::fhes_server_key_info(self.k, &mut p, std::ptr::null_mut(), std::ptr::null_mut()) };
        assert!(rc == 0, "{}", last_error());
        p
    }

    /// Opt-in: run the parallel-prefix form of every circuit that has a serial scan (identical results, tens instead
    /// of thousands of dependent levels).  Default off: the reference's own circuits.
    pub fn set_logdepth(&self, enabled: bool) {
        let rc = unsafe { ffi::fhes_server_key_set_logdepth(self.k, enabled as i32) };
        assert!(rc == 0, "{}", last_error());
    }

    /// serde (server_key.rs:11 derives it): THIS key as one blob of standard-domain words -- whether it was made from a
    /// client key, restored from a blob or received by a key broadcast; no client key needed (`fhes_server_key_serialize`).
    pub fn to_bytes(&self) -> Vec<u8> {
        let n = unsafe { ffi::fhes_server_key_serialize(self.k, std::ptr::null_mut(), 0) };
        assert!(n > 0, "{}", last_error());
        let mut buf = vec![0u8; n as usize];
        let m = unsafe { ffi::fhes_server_key_serialize(self.k, buf.as_mut_ptr(), n) };
        assert!(m == n, "{}", last_error());
        buf
    }

    /// The same blob made on the client, where the key was generated, without a device
    /// (`fhes_client_key_serialize_server_key`).
    pub fn serialize_from(ck: &ClientKey) -> Vec<u8> {
        let n = unsafe { ffi::fhes_client_key_serialize_server_key(ck.k, std::ptr::null_mut(), 0) };
        assert!(n > 0, "{}", last_error());
        let mut buf = vec![0u8; n as usize];
        let m = unsafe { ffi::fhes_client_key_serialize_server_key(ck.k, buf.as_mut_ptr(), n) };
        assert!(m == n, "{}", last_error());
        buf
    }

    /// Restores a key on `device` from a blob of [`ServerKey::to_bytes`] / [`ServerKey::serialize_from`]; needs no client key.
    pub fn from_bytes(data: &[u8], device: i32) -> Result<Self, Box<dyn std::error::Error>> {
        let k = unsafe { ffi::fhes_server_key_deserialize(data.as_ptr(), data.len() as i64, device) };
        if k.is_null() {
            return Err(last_error().into());
        }
        Ok(Self::from_handle(k))
    }

    /// The device context under this key (Level 0 / Level 1 of the ABI, multi-GPU communicators).
    pub fn ctx(&self) -> *mut ffi::FhesCtx {
        unsafe { ffi::fhes_server_key_ctx(self.k) }
    }
}

impl Drop for ServerKey {
    fn drop(&mut self) {
        unsafe { ffi::fhes_server_key_destroy(self.k) }
    }
}

impl Debug for ServerKey {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        f.debug_struct("ServerKey").finish()
    }
}

#[cfg(feature = "serde")]
impl serde::Serialize for ServerKey {
    fn serialize<S: serde::Serializer>(&self, s: S) -> Result<S::Ok, S::Error> {
        serde::Serialize::serialize(&self.to_bytes(), s)
    }
}

#[cfg(feature = "serde")]
impl<'de> serde::Deserialize<'de> for ServerKey {
    fn deserialize<D: serde::Deserializer<'de>>(d: D) -> Result<Self, D::Error> {
        let bytes: Vec<u8> = serde::Deserialize::deserialize(d)?;
        ServerKey::from_bytes(&bytes, 0).map_err(serde::de::Error::custom)
    }
}

impl Key for ServerKey {
    fn msg_mod(&self) -> usize {
        self.msg_mod
    }

    fn num_blocks_usize(&self) -> usize {
        self.num_blocks_usize
    }
}
