"""Test infrastructure: a plain-Python restatement of the reference's recover_cells (kzg/src/das.rs:566-657, with
shift_poly / coset_fft / coset_ifft :455-492 and vanishing_polynomial_for_missing_cells :494-564) on Python integers.
It exists to pin ONE behaviour the reference's vectors do not exercise: a provided cell element equal to Fr::null()
((2^256 - 1) mod r, blst/src/types/fr.rs:36-38) is treated as missing when fewer than 128 cells are given
(das.rs:611-617) and kept as it is when all 128 are (:172-181).  Pinned itself on the reference's
recover_cells_and_kzg_proofs vectors (tests/test_oracle_golden.py).  Never imported by the product."""
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
NULL = (2 ** 256 - 1) % R
CELL = 64
CELLS = 128
E = CELL * CELLS


def _brev(i, bits):
    return int(format(i, "0%db" % bits)[::-1], 2)


def _fft(vals, w):
    """iterative radix-2, natural order in and out: out[k] = sum vals[j] w^(jk)  (blst/src/fft_fr.rs:14-108)"""
    n = len(vals)
    bits = n.bit_length() - 1
    a = [vals[_brev(i, bits)] for i in range(n)]
    m = 2
    while m <= n:
        wm = pow(w, n // m, R)
        tw = [1] * (m // 2)
        for j in range(1, m // 2):
            tw[j] = tw[j - 1] * wm % R
        for k in range(0, n, m):
            for j in range(m // 2):
                t = tw[j] * a[k + j + m // 2] % R
                u = a[k + j]
                a[k + j] = (u + t) % R
                a[k + j + m // 2] = (u - t) % R
        m *= 2
    return a


def _ifft(vals, w):
    n_inv = pow(len(vals), R - 2, R)
    return [v * n_inv % R for v in _fft(vals, pow(w, R - 2, R))]


def _shift(poly, f):
    out, p = list(poly), 1
    for i in range(1, len(out)):
        p = p * f % R
        out[i] = out[i] * p % R
    return out


def recover_cells(provided, root8192):
    """provided: {cell index: [64 canonical ints]} with 64 <= len < 128; root8192: the primitive 8192-th root the
    settings use (SCALE2_ROOT_OF_UNITY[13]).  Returns the 128 recovered cells as lists of ints."""
    assert CELLS // 2 <= len(provided) < CELLS
    flat = [None] * E
    for c, vals in provided.items():
        flat[c * CELL:(c + 1) * CELL] = list(vals)
    cells_brp = [flat[_brev(i, 13)] for i in range(E)]  # missing: None ("null")
    missing = [_brev(i, 7) for i in range(CELLS) if i not in provided]
    roots = [pow(root8192, m * CELL, R) for m in missing]
    short = [(-roots[0]) % R]  # compute_vanishing_polynomial_from_roots
    for i in range(1, len(roots)):
        neg = (-roots[i]) % R
        short.append((neg + short[i - 1]) % R)
        for j in range(i - 1, 0, -1):
            short[j] = (short[j] * neg + short[j - 1]) % R
        short[0] = short[0] * neg % R
    short.append(1)
    vanishing = [0] * E
    for i, c in enumerate(short):
        vanishing[i * CELL] = c
    v_eval = _fft(vanishing, root8192)
    ez = [0 if (x is None or x == NULL) else x * v_eval[i] % R for i, x in enumerate(cells_brp)]
    ez_coeffs = _ifft(ez, root8192)
    over_coset = _fft(_shift(ez_coeffs, 7), root8192)
    v_coset = _fft(_shift(vanishing, 7), root8192)
    quot = [a * pow(b, R - 2, R) % R for a, b in zip(over_coset, v_coset)]
    coeffs = _shift(_ifft(quot, root8192), pow(7, R - 2, R))
    out = _fft(coeffs, root8192)
    out = [out[_brev(i, 13)] for i in range(E)]
    return [out[c * CELL:(c + 1) * CELL] for c in range(CELLS)]


def cell_to_ints(cell: bytes):
    return [int.from_bytes(cell[32 * j:32 * j + 32], "big") for j in range(CELL)]


def ints_to_cell(vals):
    return b"".join(int(v).to_bytes(32, "big") for v in vals)
