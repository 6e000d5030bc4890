"""Index model of the latency blind rotate with an LDS-FREE in-wave part (round 5, second layout): the C <-> D exchange through LDS
is replaced by register <-> lane-bit transposes on lane bits 3 and 2 (bank-masked row shifts) and D <-> E works on lane bit 1 (quad
permute).  Same checks as lat_layouts.py: the spectrum / round trip against the plain stage-by-stage transform, the layouts after
every transpose, and the position of a point in the engine layout of the Fourier key."""
import numpy as np

from lat_layouts import LOGM, M, T, PSI, TW, bank_check_read_b128, bank_check_write_b128, butterflies_fwd, butterflies_inv, check_layout, \
    check_same_points, dpp_fwd, dpp_inv, exchange, pA, pB, pC, permlane_swap, ref_forward, ref_inverse


def pD2(q, e):   # e = (p3, p2); l5 = p6, l4 = p7, l3 = p5, l2 = p4, l1 = p1, l0 = p0
    w, l = q >> 6, q & 63
    b = lambda k: (l >> k) & 1
    return w * 256 + b(4) * 128 + b(5) * 64 + b(3) * 32 + b(2) * 16 + (e >> 1) * 8 + (e & 1) * 4 + b(1) * 2 + b(0)


def pE2(q, e):   # e = (p3, p1); l1 = p2; rest as pD2
    w, l = q >> 6, q & 63
    b = lambda k: (l >> k) & 1
    return w * 256 + b(4) * 128 + b(5) * 64 + b(3) * 32 + b(2) * 16 + (e >> 1) * 8 + b(1) * 4 + (e & 1) * 2 + b(0)


def lane_transpose(x, ebit, lbit):
    """register bit `ebit` <-> lane bit `lbit` (lbit < 4: row shifts / quad permutes with lane selection), the semantics of the
    permlane swaps: the first register of a pair ends up with the data of lanes whose bit was 0"""
    return permlane_swap(x, ebit, lbit)


def run():
    for f in (pD2, pE2):
        check_layout(f)
    pts = np.array([[pC(q, e) for e in range(4)] for q in range(T)])
    pts = lane_transpose(lane_transpose(pts, 1, 3), 0, 2)
    check_same_points(pts, pD2)
    pts = lane_transpose(pts, 0, 1)
    check_same_points(pts, pE2)
    pts = lane_transpose(pts, 0, 1)
    check_same_points(pts, pD2)
    pts = lane_transpose(lane_transpose(pts, 0, 2), 1, 3)
    check_same_points(pts, pC)

    rng = np.random.default_rng(2)
    a = rng.normal(size=M) + 1j * rng.normal(size=M)
    ident = lambda p: p
    x = np.array([[a[pA(q, e)] for e in range(4)] for q in range(T)])
    x = butterflies_fwd(x, pA, 1); x = butterflies_fwd(x, pA, 0)
    x = exchange(x, pA, pB, ident)
    x = butterflies_fwd(x, pB, 1); x = butterflies_fwd(x, pB, 0)
    x = permlane_swap(permlane_swap(x, 0, 5), 1, 4)
    x = butterflies_fwd(x, pC, 1); x = butterflies_fwd(x, pC, 0)
    x = lane_transpose(lane_transpose(x, 1, 3), 0, 2)
    x = butterflies_fwd(x, pD2, 1); x = butterflies_fwd(x, pD2, 0)
    x = lane_transpose(x, 0, 1)
    x = butterflies_fwd(x, pE2, 0)
    x = dpp_fwd(x, pE2)
    ref = ref_forward(a)
    got = np.zeros(M, dtype=complex)
    for q in range(T):
        for e in range(4):
            got[pE2(q, e)] = x[q, e]
    assert np.allclose(got, ref), np.abs(got - ref).max()
    y = dpp_inv(x, pE2)
    y = butterflies_inv(y, pE2, 0)
    y = lane_transpose(y, 0, 1)
    y = butterflies_inv(y, pD2, 0); y = butterflies_inv(y, pD2, 1)
    y = lane_transpose(lane_transpose(y, 0, 2), 1, 3)
    y = butterflies_inv(y, pC, 0); y = butterflies_inv(y, pC, 1)
    y = permlane_swap(permlane_swap(y, 0, 5), 1, 4)
    y = butterflies_inv(y, pB, 0); y = butterflies_inv(y, pB, 1)
    y = exchange(y, pB, pA, ident)
    y = butterflies_inv(y, pA, 0); y = butterflies_inv(y, pA, 1)
    refi = ref_inverse(ref)
    goti = np.zeros(M, dtype=complex)
    for q in range(T):
        for e in range(4):
            goti[pA(q, e)] = y[q, e]
    assert np.allclose(goti, refi)
    # twiddle index formulas used by the kernel (forward: psi[2^s + block]; inverse: tw[(bits below) << (9 - bit)])
    for q in range(T):
        w, l = q >> 6, q & 63
        b = lambda k: (l >> k) & 1
        blk16 = w * 16 + b(4) * 8 + b(5) * 4 + b(3) * 2 + b(2)
        for e in range(4):
            p = pD2(q, e)
            assert p >> 4 == blk16 and (p >> 3) == 2 * blk16 + (e >> 1)
            assert (p & 3) == b(1) * 2 + b(0)
            p = pE2(q, e)
            assert (p >> 2) == blk16 * 4 + (e >> 1) * 2 + b(1)
            assert (p >> 1) == blk16 * 8 + (e >> 1) * 4 + b(1) * 2 + (e & 1)
            e8 = (e >> 1) * 4 + b(1) * 2 + (e & 1)
            q7 = blk16 * 2 + b(0)
            assert (((q7 >> 1) << 4) | (e8 << 1) | (q7 & 1)) == p
    print("exchange 1  write L_A / read L_B:", bank_check_write_b128(pA, ident), bank_check_read_b128(pB, ident))
    print("exchange 1' write L_B / read L_A:", bank_check_write_b128(pB, ident), bank_check_read_b128(pA, ident))
    print("layout v2 ok")


if __name__ == "__main__":
    run()
