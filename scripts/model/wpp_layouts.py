"""Index model of the wave-per-polynomial blind rotate (csrc/kernel_br_wpp.hip.h): ONE wave (64 lanes) per polynomial,
16 points per lane, M = 1024; register passes of 4 | 4 | 2 index bits, every transform exchange local to its wave, ONE padded
slot function for every exchange.  Simulates the data movement with complex128 arithmetic, applies the twiddles by the index
formulas the kernel uses (including the quarter-turn derivations) and compares with the plain stage-by-stage transform of
lat_layouts.py.  Checks indices and LDS bank rules, not bits."""
import numpy as np

from lat_layouts import LOGM, M, PSI, READ_GROUPS, TW, ref_forward, ref_inverse

L = 64
E = 16


# ---- layouts: point held by lane l in register e ----
def p1(l, e):   # e = (p9 p8 p7 p6), l = p5..p0
    return e * 64 + l


def p2(l, e):   # e = (p5 p4 p3 p2), l5..l2 = p9..p6, l1 l0 = p1 p0
    return (l >> 2) * 64 + e * 4 + (l & 3)


def p3(l, e):   # e = (p3 p2 p1 p0), l = p9..p4
    return l * 16 + e


def slot1(p):   # exchange L1 <-> L2: four slots of padding per 64 points (the lanes of L2 differ in p9..p6 and p1 p0)
    return p + 4 * (p >> 6)


def slot2(p):   # exchange L2 <-> L3 and the spectrum hand-over: one slot per 16 points (the lanes of L3 differ in p9..p4)
    return p + (p >> 4)


for f in (p1, p2, p3):
    assert sorted(f(l, e) for l in range(L) for e in range(E)) == list(range(M)), f.__name__
for sl in (slot1, slot2):
    assert len({sl(p) for p in range(M)}) == M and max(sl(p) for p in range(M)) < 1088


def bank_write(f, slot):
    worst = 1
    for e in range(E):
        for g in range(8):
            s = [slot(f(g * 8 + i, e)) % 8 for i in range(8)]
            worst = max(worst, max(s.count(v) for v in s))
    return worst


def bank_read(f, slot):
    worst = 1
    for e in range(E):
        for g in READ_GROUPS:
            s = [slot(f(l, e)) % 16 for l in g]
            worst = max(worst, max(s.count(v) for v in s))
    return worst


def exchange(x, f_from, f_to, slot):
    buf = np.zeros(1088, dtype=complex)
    for l in range(L):
        for e in range(E):
            buf[slot(f_from(l, e))] = x[l, e]
    y = np.zeros_like(x)
    for l in range(L):
        for e in range(E):
            y[l, e] = buf[slot(f_to(l, e))]
    return y


def neg_i(w):
    return complex(w.imag, -w.real)


# ---- twiddles by the kernel's formulas: dictionaries (stage bit, e with that bit clear) -> value ----
def fwd_twiddle(layout, l, beta, e):
    """forward twiddle of the butterfly (e, e | bit) on index bit beta; `layout` 1, 2, 3.  Loads only EVEN blocks, the odd block's
    twiddle is -i times the even one."""
    if layout == 1:
        blk = {9: 0, 8: e >> 3, 7: e >> 2, 6: e >> 1}[beta]          # index bits above beta (all register bits)
        base = {9: 1, 8: 2, 7: 4, 6: 8}[beta]
    elif layout == 2:
        h = l >> 2                                                   # p9..p6
        blk = {5: h, 4: h * 2 + (e >> 3), 3: h * 4 + (e >> 2), 2: h * 8 + (e >> 1)}[beta]
        base = {5: 16, 4: 32, 3: 64, 2: 128}[beta]
    else:
        blk = {1: l * 4 + (e >> 2), 0: l * 8 + (e >> 1)}[beta]
        base = {1: 256, 0: 512}[beta]
    if beta == 9 or (layout == 2 and beta == 5):
        return PSI[base + blk]
    return PSI[base + blk] if blk % 2 == 0 else neg_i(PSI[base + blk - 1])


def inv_twiddle(layout, l, beta, e):
    """inverse twiddle tw[(bits below beta) << (9 - beta)]; the register bit just below beta adds M/4 to the index: -i times."""
    if layout == 3:
        assert beta == 1
        return TW[(e & 1) << 8]
    if layout == 2:
        low = {2: (l & 3), 3: (e & 1) * 4 + (l & 3), 4: (e & 3) * 4 + (l & 3), 5: (e & 7) * 4 + (l & 3)}[beta]
        top = {2: 0, 3: e & 1, 4: (e >> 1) & 1, 5: (e >> 2) & 1}[beta]
    else:
        low = {6: l, 7: (e & 1) * 64 + l, 8: (e & 3) * 64 + l, 9: (e & 7) * 64 + l}[beta]
        top = {6: 0, 7: e & 1, 8: (e >> 1) & 1, 9: (e >> 2) & 1}[beta]
    sh = LOGM - 1 - beta
    if not top:
        return TW[low << sh]
    return neg_i(TW[(low - (1 << (beta - 1))) << sh])


def stage_fwd(x, f, layout, ebit):
    y = x.copy()
    for l in range(L):
        for e in range(E):
            if e & (1 << ebit):
                continue
            e1 = e | (1 << ebit)
            p, q = f(l, e), f(l, e1)
            beta = (p ^ q).bit_length() - 1
            assert q == p | (1 << beta) and not p & (1 << beta)
            t = fwd_twiddle(layout, l, beta, e)
            assert np.isclose(t, PSI[(1 << (LOGM - 1 - beta)) + (p >> (beta + 1))]), (layout, l, beta, e)
            u, v = x[l, e], x[l, e1]
            if beta == 0:
                y[l, e], y[l, e1] = u + t * v, v + (-np.conj(t)) * u
            else:
                y[l, e], y[l, e1] = u + t * v, u - t * v
    return y


def stage_inv(x, f, layout, ebit):
    y = x.copy()
    for l in range(L):
        for e in range(E):
            if e & (1 << ebit):
                continue
            e1 = e | (1 << ebit)
            p, q = f(l, e), f(l, e1)
            beta = (p ^ q).bit_length() - 1
            assert q == p | (1 << beta) and not p & (1 << beta)
            u, v = x[l, e], x[l, e1]
            if beta == 0:
                y[l, e], y[l, e1] = u + v, u - v
            else:
                w = inv_twiddle(layout, l, beta, e)
                assert np.isclose(w, TW[(p & ((1 << beta) - 1)) << (LOGM - 1 - beta)]), (layout, l, beta, e)
                y[l, e], y[l, e1] = u + v * np.conj(w), u - v * np.conj(w)
    return y


def run():
    rng = np.random.default_rng(2)
    a = rng.normal(size=M) + 1j * rng.normal(size=M)
    for name, f, sl in (("L1 / slot1", p1, slot1), ("L2 / slot1", p2, slot1), ("L2 / slot2", p2, slot2), ("L3 / slot2", p3, slot2)):
        print(f"{name}: ds_write_b128 worst {bank_write(f, sl)}-way, ds_read_b128 worst {bank_read(f, sl)}-way")
        assert bank_write(f, sl) == 1 and bank_read(f, sl) == 1
        # slot addresses are base(lane) + constant(e)
        for l in range(L):
            for e in range(E):
                assert sl(f(l, e)) - sl(f(l, 0)) == sl(f(0, e)) - sl(f(0, 0)), (name, l, e)
        print("   constant(e) =", [sl(f(0, e)) - sl(f(0, 0)) for e in range(E)])
    x = np.array([[a[p1(l, e)] for e in range(E)] for l in range(L)])
    for b in (3, 2, 1, 0):
        x = stage_fwd(x, p1, 1, b)      # bits 9 8 7 6
    x = exchange(x, p1, p2, slot1)
    for b in (3, 2, 1, 0):
        x = stage_fwd(x, p2, 2, b)      # bits 5 4 3 2
    x = exchange(x, p2, p3, slot2)
    for b in (1, 0):
        x = stage_fwd(x, p3, 3, b)      # bits 1 0
    ref = ref_forward(a)
    got = np.zeros(M, dtype=complex)
    for l in range(L):
        for e in range(E):
            got[p3(l, e)] = x[l, e]
    print("forward max err", np.abs(got - ref).max())
    assert np.allclose(got, ref)
    y = x
    for b in (0, 1):
        y = stage_inv(y, p3, 3, b)
    y = exchange(y, p3, p2, slot2)
    for b in (0, 1, 2, 3):
        y = stage_inv(y, p2, 2, b)
    y = exchange(y, p2, p1, slot1)
    for b in (0, 1, 2, 3):
        y = stage_inv(y, p1, 1, b)
    refi = ref_inverse(ref)
    goti = np.zeros(M, dtype=complex)
    for l in range(L):
        for e in range(E):
            goti[p1(l, e)] = y[l, e]
    print("inverse max err", np.abs(goti - refi).max())
    assert np.allclose(goti, refi)
    # the engine layout of the Fourier key: element (e8, q7) holds point ((q7 >> 1) << 4) | (e8 << 1) | (q7 & 1)
    for l in range(L):
        for e in range(E):
            e8, q7 = e >> 1, l * 2 + (e & 1)
            assert (((q7 >> 1) << 4) | (e8 << 1) | (q7 & 1)) == p3(l, e)
    # rotated reads of the image in plain order: consecutive lanes read consecutive 8-byte words (conflict-free by construction)
    print("ok")


if __name__ == "__main__":
    run()
