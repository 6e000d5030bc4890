//! `fhe_string` over the MI355X engine: the public surface of the reference crate
//! (`/root/reference/src/lib.rs:34-89`) with every homomorphic operation executed by `libfhe_string_amd.so`
//! through its C ABI (`include/fhe_string_amd.h`, declared in [`ffi`]).
//!
//! **Uncompiled**: the build image has no Rust toolchain.  `tests/test_rust_binding.py` keeps this tree in step
//! with the header and with the reference's `pub fn` list; the same calls run compiled in the Python mirror
//! (`fhe_string_amd/api.py`), which the GPU tests exercise.
//!
//! ```ignore
//! use fhe_string::{generate_keys, StringEncryption};
//! let (client_key, server_key) = generate_keys();
//! let input = "a,b,c".encrypt(&client_key, Some(8)).unwrap();
//! let pattern = ",".encrypt(&client_key, None).unwrap();
//! let found = input.find(&server_key, &pattern);
//! assert_eq!(found.decrypt(&client_key), Some(1));
//! ```
use std::error::Error;

pub use ciphertext::{split::FheStringSliceVector, FheAsciiChar, FheOption, FheString, FheUsize};
pub use ciphertext::{BooleanBlock, RadixCiphertext};
pub use client_key::{ClientKey, Key};
pub use server_key::ServerKey;

mod ciphertext;
mod client_key;
pub mod ffi;
mod server_key;

/// The parameter block of the C ABI under the name the reference's signature uses
/// (`generate_keys_with_params(p: ClassicPBSParameters, l: usize)`, lib.rs:50).
pub type ClassicPBSParameters = ffi::FhesParams;

/// tfhe-rs 0.5 `PARAM_MESSAGE_2_CARRY_2_KS_PBS` (the set lib.rs:45 fixes), field for field what
/// `fhes_params_default` returns (checked by tests/test_rust_binding.py).
pub const PARAM_MESSAGE_2_CARRY_2_KS_PBS: ClassicPBSParameters = ClassicPBSParameters {
    lwe_dim: 742,
    glwe_dim: 1,
    poly_size: 2048,
    pbs_base_log: 23,
    pbs_level: 1,
    ks_base_log: 3,
    ks_level: 5,
    message_bits: 2,
    carry_bits: 2,
    grouping_factor: 1,
    lwe_noise_std: 0.000007069849454709433,
    glwe_noise_std: 0.00000000000000029403601535432533,
};

/// Multi-bit sets of BASELINE config 5 (`fhes_params_multibit`; the reference cannot select them, lib.rs:50).
pub fn multi_bit_parameters(grouping_factor: i32) -> Option<ClassicPBSParameters> {
    let mut p = PARAM_MESSAGE_2_CARRY_2_KS_PBS;
    let rc = unsafe { ffi::fhes_params_multibit(&mut p, grouping_factor) };
    (rc == 0).then_some(p)
}

/// Generates a fresh key pair using the default parameters. The maximum string
/// length is set to 255.  (lib.rs:44-46)
pub fn generate_keys() -> (ClientKey, ServerKey) {
    generate_keys_with_params(PARAM_MESSAGE_2_CARRY_2_KS_PBS, 255)
}

/// Generates a fresh key pair using encryption scheme parameters `p` and
/// maximum string length `l`.  (lib.rs:50-77; the block counts of :51-62 are derived by the engine:
/// `fhes_client_key_num_blocks_char / _usize`.)  The server key lives on HIP device 0; use
/// [`ServerKey::new_on_device`] for another one (one process per GPU).
pub fn generate_keys_with_params(p: ClassicPBSParameters, l: usize) -> (ClientKey, ServerKey) {
    let client_key = ClientKey::generate(p, l);
    let server_key = ServerKey::new_on_device(&client_key, 0);
    (client_key, server_key)
}

/// Support for string encryption.  (lib.rs:80-89)
pub trait StringEncryption {
    /// Encrypt `self` into an `FheString`.
    fn encrypt(&self, k: &ClientKey, l: Option<usize>) -> Result<FheString, Box<dyn Error>>;
}

impl StringEncryption for str {
    fn encrypt(&self, k: &ClientKey, l: Option<usize>) -> Result<FheString, Box<dyn Error>> {
        FheString::new(k, self, l)
    }
}

/// The paths under which code written against the reference names tfhe-rs's types (its example does:
/// `use tfhe::{integer::BooleanBlock, shortint::prelude::PARAM_MESSAGE_2_CARRY_2_KS_PBS};`, examples/cmd/main.rs:5): with this crate
/// the same line reads `use fhe_string::tfhe::{...}`.
pub mod tfhe {
    pub mod integer {
        pub use crate::{BooleanBlock, RadixCiphertext};
    }
    pub mod shortint {
        pub use crate::ClassicPBSParameters;
        pub mod parameters {
            pub use crate::PARAM_MESSAGE_2_CARRY_2_KS_PBS;
        }
        pub mod prelude {
            pub use crate::PARAM_MESSAGE_2_CARRY_2_KS_PBS;
        }
    }
}

/// Message of the last failing ABI call on this thread (`fhes_last_error`).
pub(crate) fn last_error() -> String {
    unsafe {
        let p = ffi::fhes_last_error();
        if p.is_null() {
            String::from("unknown error")
        } else {
            std::ffi::CStr::from_ptr(p).to_string_lossy().into_owned()
        }
    }
}
