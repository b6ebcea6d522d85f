"""Algorithmic FLOP / byte counts of one scene-step - the figures the roofline
fraction in bench.py is computed from.  Dense general-Q formulation exactly as the reference
performs it; a multiply-add counts 2; `it` = PDIPM loop iterations executed by that scene."""


def flops_forward(nz, m, e, it):
    k = e + m
    P = (2.0 / 3) * nz ** 3 + 2 * nz * nz * m + 2 * m * m * nz + m * m
    if e > 0:
        P += 2 * nz * nz * e + 2 * e * e * nz + 2 * m * nz * e + (2.0 / 3) * e ** 3 + 2 * e ** 3 + 6 * m * e * e + 2 * m * m * e
    Fk = (2.0 / 3) * m ** 3 + m
    Sk = 4 * nz * nz + 4 * m * nz + 4 * e * nz + 2 * k * k + 6 * m
    Rk = 2 * nz * nz + 4 * m * nz + 4 * e * nz + 2 * m * m + 10 * m
    I = Fk + 2 * Sk + Rk + 20 * m
    return P + Fk + Sk + it * I


def flops_backward(nz, m, e):
    k = e + m
    Fk = (2.0 / 3) * m ** 3 + m
    Sk = 4 * nz * nz + 4 * m * nz + 4 * e * nz + 2 * k * k + 6 * m
    return Fk + Sk + 4 * m * nz + m * m + 4 * e * nz + 2 * nz * nz


def bytes_forward(nz, m, e, word=4):
    bytes_in = word * (nz * nz + nz + m * nz + m + e * nz + e + m * m)
    bytes_out = word * (nz + 2 * m + e)
    return bytes_in + bytes_out


def bytes_backward(nz, m, e, word=4):
    rd = word * (nz * nz + m * nz + e * nz + m * m) + word * (nz + 2 * m + e) + word * nz
    wr = word * (nz * nz + nz + m * nz + m + e * nz + e + m * m)
    return rd + wr


def bytes_fused_step(nb, nc, word=4):
    """Fused boundary: contact list + body state in, new velocity + pose out."""
    return word * (14 * nb + 7 * nc) + 8 * nc + word * 6 * nb
