"""Index model of the 512-thread latency blind rotate (csrc/kernel_br_lat.hip.h): 256 threads per polynomial, 4 points per
thread, M = 1024.  Simulates the data movement (LDS exchanges with their swizzles, v_permlane32/16_swap transposes, the
lane-pair stages) with complex128 arithmetic and compares the spectrum / the round trip with a plain stage-by-stage transform
that follows DESIGN.md 4.3.  Checks indices (which operands meet in which butterfly with which twiddle), not bits."""
import numpy as np

LOGM, M, N = 10, 1024, 2048
T = 256


def tables():
    t = np.arange(2 * N)
    root = np.exp(1j * np.pi * t / N)

    def bitrev(m):
        f = 0
        for b in range(LOGM):
            f |= ((m >> b) & 1) << (LOGM - 1 - b)
        return f

    def slot_exp(m):
        return (1 - 4 * bitrev(m)) % (2 * N)
    psi = np.zeros(M, dtype=complex)
    for l in range(LOGM):
        nb, half = 1 << l, M >> (l + 1)
        for b in range(nb):
            psi[nb + b] = root[(slot_exp(b * 2 * half) * half) % (2 * N)]
    tw = np.exp(-2j * np.pi * np.arange(M // 2) / M)
    return root, psi, tw


ROOT, PSI, TW = tables()


def ref_forward(a):
    x = a.copy()
    for beta in range(LOGM - 1, -1, -1):
        sh = LOGM - 1 - beta
        for p in range(M):
            if p & (1 << beta):
                continue
            p1 = p | (1 << beta)
            t = PSI[(1 << sh) + (p >> (beta + 1))]
            u, v = x[p], x[p1]
            if beta == 0:
                x[p], x[p1] = u + t * v, v + (-np.conj(t)) * u
            else:
                x[p], x[p1] = u + t * v, u - t * v
    return x


def ref_inverse(s):
    x = s.copy()
    for beta in range(LOGM):
        sh = LOGM - 1 - beta
        for p in range(M):
            if p & (1 << beta):
                continue
            p1 = p | (1 << beta)
            u, v = x[p], x[p1]
            if beta == 0:
                x[p], x[p1] = u + v, u - v
            else:
                w = TW[(p & ((1 << beta) - 1)) << sh]
                x[p], x[p1] = u + v * np.conj(w), u - v * np.conj(w)
    return x


def bits(v, *idx):
    return [(v >> i) & 1 for i in idx]


# ---- layouts: point index held by thread q (w = q >> 6, l = q & 63) in register e ----
def pA(q, e):
    return q + 256 * e


def pB(q, e):
    w, l = q >> 6, q & 63
    l5, l4 = (l >> 5) & 1, (l >> 4) & 1
    return w * 256 + e * 64 + l4 * 32 + l5 * 16 + (l & 15)


def pC(q, e):
    w, l = q >> 6, q & 63
    l5, l4 = (l >> 5) & 1, (l >> 4) & 1
    return w * 256 + l4 * 128 + l5 * 64 + e * 16 + (l & 15)


def pD(q, e):
    w, l = q >> 6, q & 63
    l5, l0 = (l >> 5) & 1, l & 1
    return w * 256 + ((l >> 1) & 15) * 16 + e * 4 + l5 * 2 + l0


def pE(q, e):
    w, l = q >> 6, q & 63
    l5, l0 = (l >> 5) & 1, l & 1
    e1, e0 = e >> 1, e & 1
    return w * 256 + ((l >> 1) & 15) * 16 + e1 * 8 + l5 * 4 + e0 * 2 + l0


def fswz(p):
    return p ^ (((p >> 4) & 7) << 1)


def check_layout(f):
    seen = sorted(f(q, e) for q in range(T) for e in range(4))
    assert seen == list(range(M)), f.__name__


for f in (pA, pB, pC, pD, pE):
    check_layout(f)


def exchange(x, f_from, f_to, slot):
    buf = np.zeros(M, dtype=complex)
    for q in range(T):
        for e in range(4):
            buf[slot(f_from(q, e))] = x[q, e]
    y = np.zeros_like(x)
    for q in range(T):
        for e in range(4):
            y[q, e] = buf[slot(f_to(q, e))]
    return y


def permlane_swap(x, ebit, lbit):
    """v_permlane32_swap (lbit 5) / v_permlane16_swap (lbit 4) on the register pairs that differ in e bit `ebit`"""
    y = x.copy()
    for w in range(4):
        for e in range(4):
            if e & (1 << ebit):
                continue
            e1 = e | (1 << ebit)
            for l in range(64):
                q = w * 64 + l
                if (l >> lbit) & 1:        # odd row of the first operand <- even row of the second
                    y[q, e] = x[w * 64 + (l ^ (1 << lbit)), e1]
                else:                       # even row of the second operand <- odd row of the first
                    y[q, e1] = x[w * 64 + (l ^ (1 << lbit)), e]
    return y


def check_same_points(x_pts, f):
    for q in range(T):
        for e in range(4):
            assert x_pts[q, e] == f(q, e), (f.__name__, q, e, x_pts[q, e], f(q, e))


def bank_check_write_b128(f, slot):
    """ds_write_b128: groups of 8 contiguous lanes, 32 banks of 4 B -> the 8 slots of a group distinct mod 8"""
    worst = 1
    for e in range(4):
        for w in range(4):
            for g in range(8):
                s = [slot(f(w * 64 + g * 8 + i, e)) % 8 for i in range(8)]
                worst = max(worst, max(s.count(v) for v in s))
    return worst


READ_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
READ_GROUPS += [[l + 32 for l in g] for g in READ_GROUPS]


def bank_check_read_b128(f, slot):
    worst = 1
    for e in range(4):
        for w in range(4):
            for g in READ_GROUPS:
                s = [slot(f(w * 64 + l, e)) % 16 for l in g]
                worst = max(worst, max(s.count(v) for v in s))
    return worst


def butterflies_fwd(x, f, ebit):
    """forward stage on the index bit that register bit `ebit` holds in layout f"""
    y = x.copy()
    for q in range(T):
        for e in range(4):
            if e & (1 << ebit):
                continue
            e1 = e | (1 << ebit)
            p, p1 = f(q, e), f(q, e1)
            beta = (p ^ p1).bit_length() - 1
            assert p1 == p | (1 << beta) and not p & (1 << beta)
            t = PSI[(1 << (LOGM - 1 - beta)) + (p >> (beta + 1))]
            u, v = x[q, e], x[q, e1]
            y[q, e], y[q, e1] = u + t * v, u - t * v
    return y


def butterflies_inv(x, f, ebit):
    y = x.copy()
    for q in range(T):
        for e in range(4):
            if e & (1 << ebit):
                continue
            e1 = e | (1 << ebit)
            p, p1 = f(q, e), f(q, e1)
            beta = (p ^ p1).bit_length() - 1
            assert p1 == p | (1 << beta) and not p & (1 << beta)
            w = TW[(p & ((1 << beta) - 1)) << (LOGM - 1 - beta)]
            u, v = x[q, e], x[q, e1]
            y[q, e], y[q, e1] = u + v * np.conj(w), u - v * np.conj(w)
    return y


def dpp_fwd(x, f):
    """lane-pair stage of the forward transform: own + c * partner (lane ^ 1)"""
    y = x.copy()
    for q in range(T):
        for e in range(4):
            p = f(q, e)
            assert f(q ^ 1, e) == p ^ 1
            t = PSI[M // 2 + (p >> 1)]
            c = (-np.conj(t)) if (p & 1) else t
            y[q, e] = x[q, e] + c * x[q ^ 1, e]
    return y


def dpp_inv(x, f):
    y = x.copy()
    for q in range(T):
        for e in range(4):
            p = f(q, e)
            sgn = -1.0 if (p & 1) else 1.0
            y[q, e] = x[q, e] * sgn + x[q ^ 1, e]
    return y


def run():
    rng = np.random.default_rng(1)
    a = rng.normal(size=M) + 1j * rng.normal(size=M)
    ident = lambda p: p
    # conflicts of every LDS exchange
    print("exchange 1  write L_A / read L_B (identity):", bank_check_write_b128(pA, ident), bank_check_read_b128(pB, ident))
    print("exchange 2  write L_C / read L_D (fswz):    ", bank_check_write_b128(pC, fswz), bank_check_read_b128(pD, fswz))
    print("exchange 2' write L_D / read L_C (fswz):    ", bank_check_write_b128(pD, fswz), bank_check_read_b128(pC, fswz))
    print("exchange 1' write L_B / read L_A (identity):", bank_check_write_b128(pB, ident), bank_check_read_b128(pA, ident))

    # track which point each register holds, through the permutes
    pts = np.array([[pB(q, e) for e in range(4)] for q in range(T)])
    pts = permlane_swap(permlane_swap(pts, 0, 5), 1, 4)
    check_same_points(pts, pC)
    pts = np.array([[pD(q, e) for e in range(4)] for q in range(T)])
    pts = permlane_swap(pts, 0, 5)
    check_same_points(pts, pE)
    # and back
    pts = permlane_swap(np.array([[pE(q, e) for e in range(4)] for q in range(T)]), 0, 5)
    check_same_points(pts, pD)
    pts = np.array([[pC(q, e) for e in range(4)] for q in range(T)])
    pts = permlane_swap(permlane_swap(pts, 0, 5), 1, 4)
    check_same_points(pts, pB)

    # ---- forward ----
    x = np.array([[a[pA(q, e)] for e in range(4)] for q in range(T)])
    x = butterflies_fwd(x, pA, 1)      # bit 9
    x = butterflies_fwd(x, pA, 0)      # bit 8
    x = exchange(x, pA, pB, ident)
    x = butterflies_fwd(x, pB, 1)      # 7
    x = butterflies_fwd(x, pB, 0)      # 6
    x = permlane_swap(permlane_swap(x, 0, 5), 1, 4)
    x = butterflies_fwd(x, pC, 1)      # 5
    x = butterflies_fwd(x, pC, 0)      # 4
    x = exchange(x, pC, pD, fswz)
    x = butterflies_fwd(x, pD, 1)      # 3
    x = butterflies_fwd(x, pD, 0)      # 2
    x = permlane_swap(x, 0, 5)
    x = butterflies_fwd(x, pE, 0)      # 1
    x = dpp_fwd(x, pE)                 # 0
    ref = ref_forward(a)
    got = np.zeros(M, dtype=complex)
    for q in range(T):
        for e in range(4):
            got[pE(q, e)] = x[q, e]
    print("forward max err", np.abs(got - ref).max())
    assert np.allclose(got, ref)

    # ---- inverse ----
    y = dpp_inv(x, pE)
    y = butterflies_inv(y, pE, 0)      # 1
    y = permlane_swap(y, 0, 5)
    y = butterflies_inv(y, pD, 0)      # 2
    y = butterflies_inv(y, pD, 1)      # 3
    y = exchange(y, pD, pC, fswz)
    y = butterflies_inv(y, pC, 0)      # 4
    y = butterflies_inv(y, pC, 1)      # 5
    y = permlane_swap(permlane_swap(y, 0, 5), 1, 4)
    y = butterflies_inv(y, pB, 0)      # 6
    y = butterflies_inv(y, pB, 1)      # 7
    y = exchange(y, pB, pA, ident)
    y = butterflies_inv(y, pA, 0)      # 8
    y = butterflies_inv(y, pA, 1)      # 9
    refi = ref_inverse(ref)
    goti = np.zeros(M, dtype=complex)
    for q in range(T):
        for e in range(4):
            goti[pA(q, e)] = y[q, e]
    print("inverse max err", np.abs(goti - refi).max())
    assert np.allclose(goti, refi)
    # the engine layout of the Fourier key (bsk_to_engine_layout_kernel): element (e8, q7) holds point
    # ((q7 >> 1) << 4) | (e8 << 1) | (q7 & 1); a thread of this kernel finds point p at e8 * 128 + q7 with
    for q in range(T):
        for e in range(4):
            p = pE(q, e)
            w, l = q >> 6, q & 63
            e8 = (e >> 1) * 4 + ((l >> 5) & 1) * 2 + (e & 1)
            q7 = w * 32 + (l & 31)
            assert (((q7 >> 1) << 4) | (e8 << 1) | (q7 & 1)) == p
    print("ok")


if __name__ == "__main__":
    run()
