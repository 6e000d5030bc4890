//! Runs string operations in the encrypted domain on the GPU and checks each against `std::str` -- the counterpart of the
//! reference's `examples/cmd` (same flags: --input --pattern [--substitution _] [--pad N] [--n 3] [--l 255] [--logdepth]),
//! written against this crate.  (Uncompiled, like the rest of the tree.)
//!
//!     cargo run --example cmd -- --input " A bcbc " --pattern "bc"
use std::time::Instant;

use fhe_string::tfhe::shortint::prelude::PARAM_MESSAGE_2_CARRY_2_KS_PBS;
use fhe_string::{generate_keys_with_params, FheString, FheUsize};

fn arg(name: &str) -> Option<String> {
    let args: Vec<String> = std::env::args().collect();
    args.iter().position(|a| a == name).and_then(|i| args.get(i + 1).cloned())
}

fn report<T: PartialEq + std::fmt::Debug>(name: &str, t0: Instant, got: T, want: T) {
    let ok = if got == want { "ok" } else { "MISMATCH" };
    println!("{name:<24} {:>8.1} ms  {ok}  {:?}", t0.elapsed().as_secs_f64() * 1e3, got);
}

fn main() {
    env_logger::init();
    let input = arg("--input").expect("--input");
    let pattern = arg("--pattern").expect("--pattern");
    let substitution = arg("--substitution").unwrap_or_else(|| "_".to_string());
    let pad: Option<usize> = arg("--pad").map(|v| v.parse().expect("--pad"));
    let n: usize = arg("--n").map_or(3, |v| v.parse().expect("--n"));
    let l: usize = arg("--l").map_or(255, |v| v.parse().expect("--l"));

    let (ck, sk) = generate_keys_with_params(PARAM_MESSAGE_2_CARRY_2_KS_PBS, l);
    if std::env::args().any(|a| a == "--logdepth") {
        sk.set_logdepth(true);
    }
    let a = FheString::new(&ck, &input, pad).unwrap();
    let p = FheString::new(&ck, &pattern, None).unwrap();
    let s = FheString::new(&ck, &substitution, None).unwrap();
    let n_enc = FheUsize::new(&ck, n);

    let t = Instant::now();
    report("len", t, a.len(&sk).decrypt(&ck), input.len());
    let t = Instant::now();
    report("contains", t, ck.decrypt_bool(&a.contains(&sk, &p)), input.contains(&pattern));
    let t = Instant::now();
    report("find", t, a.find(&sk, &p).decrypt(&ck), input.find(&pattern));
    let t = Instant::now();
    report("rfind", t, a.rfind(&sk, &p).decrypt(&ck), input.rfind(&pattern));
    let t = Instant::now();
    report("starts_with", t, ck.decrypt_bool(&a.starts_with(&sk, &p)), input.starts_with(&pattern));
    let t = Instant::now();
    report("eq", t, ck.decrypt_bool(&a.eq(&sk, &p)), input == pattern);
    let t = Instant::now();
    report("lt", t, ck.decrypt_bool(&a.lt(&sk, &p)), input < pattern);
    let t = Instant::now();
    report("trim", t, a.trim(&sk).decrypt(&ck), input.trim().to_string());
    let t = Instant::now();
    report("to_uppercase", t, a.to_uppercase(&sk).decrypt(&ck), input.to_uppercase());
    if !pattern.is_empty() {
        let t = Instant::now();
        report("split", t, a.split(&sk, &p).decrypt(&ck), input.split(&pattern).map(String::from).collect::<Vec<_>>());
        let t = Instant::now();
        report("splitn", t, a.splitn(&sk, &n_enc, &p).decrypt(&ck), input.splitn(n, &pattern).map(String::from).collect::<Vec<_>>());
        let want = input.replace(&pattern, &substitution);
        let t = Instant::now();
        report("replace", t, a.replace(&sk, &p, &s, want.len()).decrypt(&ck), want);
    }
    let t = Instant::now();
    report("strip_prefix", t, a.strip_prefix(&sk, &p).decrypt(&ck), input.strip_prefix(pattern.as_str()).map(String::from));
}
