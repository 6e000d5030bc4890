"""exact_pbs.py -- EXACT-INTEGER statement of the programmable bootstrap (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

A second statement of the TFHE layer next to oracle/tfhe_oracle.c, written from the published scheme (CGGI: Chillotti, Gama,
Georgieva, Izabachene, "TFHE: Fast Fully Homomorphic Encryption over the Torus"; the tfhe-rs 0.5 conventions the reference's
`tfhe = "0.5.0"` dependency uses, /root/reference/Cargo.toml:7, for the closest-representable signed decomposition, the
modulus switch and the multi-bit bundle) with NO floating point anywhere: every negacyclic product of the external product is a
schoolbook convolution in Z_{2^64}[X] / (X^N + 1).  What it pins (tests/test_exact_pbs.py):

  * keyswitch, modulus switch, sample extract: integers only -- the C oracle must agree BIT FOR BIT;
  * blind rotate (classic CMUX chain and the multi-bit bundle form): the C oracle (and with it the HIP kernels, which equal the
    oracle bit for bit) differs from this exact evaluation only by the rounding of its f64 transforms -- the tests bound that
    difference far below the scheme's own noise, coefficient by coefficient.  So the freedom left in "the spec" of DESIGN.md
    section 4 is the rounding of the FFT (which tfhe-rs's own FFT has too, differently) and nothing else: decomposition,
    rotation signs, key layout, accumulation order of rows and the monomial degrees of the multi-bit bundle are all checked
    against an evaluation that contains no transform, no twist table and no approximation of a monomial's spectrum.

Nothing here shares code with tfhe_oracle.c or fhe_string_amd/csrc; only the key LAYOUTS (oracle/tfhe_oracle.h:69-73) are common,
because the keys are handed over as data.  Speed is not a goal: one bootstrap at the real parameter set takes tens of seconds."""
import numpy as np

U64 = np.uint64
MASK64 = (1 << 64) - 1


def closest_digits(x, base_log, levels):
    """Signed decomposition of the words x (uint64 array) by definition: q = round(x / 2^(64 - levels*base_log)) with ties up,
    modulo B^levels; q = sum_l d_l B^(levels - 1 - l) with every d_l in [-B/2, B/2) (balanced digits, carries upwards; a carry
    out of the top digit wraps with the torus).  Returns int64 [levels][len(x)], index 0 = the MOST significant digit, the one
    that belongs to the gadget element 2^(64 - base_log)."""
    total = base_log * levels
    B = 1 << base_log
    q = [((int(v) + (1 << (63 - total))) >> (64 - total)) % (1 << total) for v in np.asarray(x, dtype=U64).tolist()]
    out = np.zeros((levels, len(q)), dtype=np.int64)
    for j, v in enumerate(q):
        for l in range(levels - 1, -1, -1):
            d = v % B
            if d >= B // 2:
                d -= B
            v = (v - d) // B
            out[l, j] = d
    return out


def closest_digits_fast(x, base_log, levels):
    """the same digits with array operations (checked against closest_digits in the tests)"""
    total = base_log * levels
    B = np.int64(1 << base_log)
    x = np.asarray(x, dtype=U64)
    q = (((x >> U64(64 - total - 1)) + U64(1)) >> U64(1)) & U64((1 << total) - 1)
    v = q.astype(np.int64)
    out = np.zeros((levels, len(x)), dtype=np.int64)
    for l in range(levels - 1, -1, -1):
        d = v & (B - 1)
        d = np.where(d >= B // 2, d - B, d)
        v = (v - d) >> np.int64(base_log)
        out[l] = d
    return out


def negacyclic_mul(d, g):
    """(d * g) mod (X^N + 1, 2^64): d small signed integers (int64), g torus words (uint64); exact"""
    n = len(g)
    full = np.convolve(np.asarray(d, dtype=np.int64).astype(U64), np.asarray(g, dtype=U64))   # wraps modulo 2^64
    out = full[:n].copy()
    out[:n - 1] -= full[n:]
    return out


def monomial_mul(p, t):
    """X^t * p in Z_{2^64}[X] / (X^N + 1), t in [0, 2N)"""
    n = len(p)
    t %= 2 * n
    out = np.empty_like(p)
    s = t % n
    out[s:] = p[:n - s]
    out[:s] = U64(0) - p[n - s:]
    return (U64(0) - out) if t >= n else out


def modswitch(v, poly_size):
    """round(v * 2N / 2^64) mod 2N, ties up"""
    two_n = 2 * poly_size
    return ((int(v) * two_n + (1 << 63)) >> 64) % two_n


def keyswitch(ct_big, ksk, n_small, base_log, levels):
    """LWE keyswitch big -> small key: (0, b) - sum_i sum_l d_{i,l} KSK[i][l]   (ksk: uint64 [big][levels][n_small + 1])"""
    big = len(ct_big) - 1
    ksk = np.asarray(ksk, dtype=U64).reshape(big, levels, n_small + 1)
    dig = closest_digits_fast(ct_big[:big], base_log, levels)          # [levels][big]
    out = np.zeros(n_small + 1, dtype=U64)
    out[n_small] = ct_big[big]
    for l in range(levels):
        out -= (dig[l].astype(U64)[:, None] * ksk[:, l, :]).sum(axis=0, dtype=U64)
    return out


def external_product(acc_like, ggsw, base_log, levels):
    """sum over rows r and levels l of digits(acc_like[r])_l * GGSW[l][r][col]   (ggsw: uint64 [levels][k+1][k+1][N])"""
    kp1, n = acc_like.shape
    out = np.zeros((kp1, n), dtype=U64)
    for r in range(kp1):
        dig = closest_digits_fast(acc_like[r], base_log, levels)
        for l in range(levels):
            if not dig[l].any():
                continue
            for col in range(kp1):
                out[col] += negacyclic_mul(dig[l], ggsw[l, r, col])
    return out


def blind_rotate(small, lut_poly, bsk, glwe_dim, poly_size, base_log, levels):
    """classic CGGI blind rotate: acc = X^{-b~} (0, .., 0, lut); for every mask element acc += (X^{a~_i} acc - acc) [x] BSK_i.
    small: the keyswitched ciphertext (uint64 [n + 1]); bsk: uint64 [n][levels][k+1][k+1][N]."""
    n = len(small) - 1
    kp1, N = glwe_dim + 1, poly_size
    bsk = np.asarray(bsk, dtype=U64).reshape(n, levels, kp1, kp1, N)
    acc = np.zeros((kp1, N), dtype=U64)
    acc[glwe_dim] = monomial_mul(np.asarray(lut_poly, dtype=U64), (2 * N - modswitch(small[n], N)) % (2 * N))
    for i in range(n):
        a = modswitch(small[i], N)
        if a == 0:
            continue
        diff = np.stack([monomial_mul(acc[r], a) - acc[r] for r in range(kp1)])
        acc = acc + external_product(diff, bsk[i], base_log, levels)
    return acc


def blind_rotate_multibit(small, lut_poly, bsk, glwe_dim, poly_size, base_log, levels, g):
    """multi-bit blind rotate (tfhe-rs multi_bit_blind_rotate_assign): per group of g mask elements the bundle
    G_0 + sum_{c >= 1} X^{deg_c} G_c with deg_c = modswitch(sum of the group's mask elements selected by the bits of c, the
    first element by the HIGHEST bit) -- the sum taken in Z_{2^64} BEFORE the switch --, then acc <- bundle [x] acc.
    bsk: uint64 [n / g][2^g][levels][k+1][k+1][N], in the standard domain: the monomial acts by exact rotation."""
    n = len(small) - 1
    kp1, N = glwe_dim + 1, poly_size
    bsk = np.asarray(bsk, dtype=U64).reshape(n // g, 1 << g, levels, kp1, kp1, N)
    acc = np.zeros((kp1, N), dtype=U64)
    acc[glwe_dim] = monomial_mul(np.asarray(lut_poly, dtype=U64), (2 * N - modswitch(small[n], N)) % (2 * N))
    for j in range(n // g):
        bundle = bsk[j, 0].copy()
        for c in range(1, 1 << g):
            s = sum(int(small[j * g + t]) for t in range(g) if (c >> (g - 1 - t)) & 1) & MASK64
            deg = modswitch(s, N)
            gc = bsk[j, c]
            for l in range(levels):
                for r in range(kp1):
                    for col in range(kp1):
                        bundle[l, r, col] += monomial_mul(gc[l, r, col], deg)
        acc = external_product(acc, bundle, base_log, levels)
    return acc


def sample_extract(acc):
    """the LWE ciphertext of the constant coefficient under the flattened GLWE key"""
    kp1, N = acc.shape
    out = np.zeros((kp1 - 1) * N + 1, dtype=U64)
    for r in range(kp1 - 1):
        out[r * N] = acc[r, 0]
        out[r * N + 1:(r + 1) * N] = U64(0) - acc[r, :0:-1]
    out[(kp1 - 1) * N] = acc[kp1 - 1, 0]
    return out


def pbs(ct_big, lut_poly, bsk, ksk, p):
    """keyswitch -> blind rotate -> sample extract with the fields of a parameter struct p (lwe_dim, glwe_dim, poly_size,
    pbs_base_log, pbs_level, ks_base_log, ks_level, grouping_factor)"""
    small = keyswitch(np.asarray(ct_big, dtype=U64), ksk, p.lwe_dim, p.ks_base_log, p.ks_level)
    if p.grouping_factor > 1:
        acc = blind_rotate_multibit(small, lut_poly, bsk, p.glwe_dim, p.poly_size, p.pbs_base_log, p.pbs_level, p.grouping_factor)
    else:
        acc = blind_rotate(small, lut_poly, bsk, p.glwe_dim, p.poly_size, p.pbs_base_log, p.pbs_level)
    return sample_extract(acc), small, acc
