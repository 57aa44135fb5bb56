"""block_data_<j>.bin — the Boost-binary flavour `pmp2sdp` writes by default
(src/pmp2sdp/Pmp2sdp_Parameters/Pmp2sdp_Parameters.cxx:36-39).

Layout restated from the writer (src/pmp2sdp/write_block_data.cxx:89-103), the serialisers
(src/sdpb_util/boost_serialization.hxx:19-86) and the reader
(src/sdp_solve/SDP/read_block_data/SDP_Block_Data.cxx:35-47); the container is a
boost::archive::binary_oarchive (x86-64, little endian):

  header     u64 22, "serialization::archive", u16 library version,
             u8 sizeof(int)=4, u8 sizeof(long)=8, u8 sizeof(float)=4, u8 sizeof(double)=8, i32 1
  u64        El::gmp::Precision()                (mp_bitcnt_t; must match --precision as GMP rounds it)
  Matrix     B  (constraint_matrix, P x N)       first El::Matrix: u8 tracking=0, u32 class version=0
  vector     c  (constraint_constants, P)        u64 count, u32 item_version, elements
  Matrix     bilinear_bases_even, bilinear_bases_odd
  El::Matrix = El::Int height, width, ldim (4 bytes each; 8 accepted), then ldim*width numbers
               column-major
  BigFloat   the first one in the archive is preceded by u8 tracking=0, u32 class version=1;
             every one: u8 is_zero; unless zero: El::BigFloat::Serialize bytes =
             i32 _mp_prec, i32 _mp_size, i64 _mp_exp, num_limbs x u64 _mp_d with
             num_limbs = (max(p,53)+127)/64 + 1  (test/src/integration_tests/util/Float.cxx:13-20)

PINNING: neither Boost nor the Elemental fork is in /root/reference and its test data holds no
.bin file, so this layout cannot be checked against a reference-written sample here; it is
self-consistent (writer <-> reader round trip, tests/test_sdp_formats.py) and the reader rejects any
file whose size, precision word or matrix shapes disagree with it instead of guessing.
"""
from __future__ import annotations

import io
import struct
from typing import List, Tuple

SIGNATURE = b"serialization::archive"
LIBRARY_VERSION = 19  # Boost 1.74+; any value is accepted on read


def gmp_precision(precision_bits: int) -> int:
    """El::gmp::Precision() after mpf_set_default_prec(p): 64 * (prec_limbs - 1) with
    prec_limbs = (max(53,p) + 127) / 64  (GMP's __GMPF_BITS_TO_PREC / PREC_TO_BITS)."""
    return 64 * ((max(53, precision_bits) + 127) // 64) - 64


def num_limbs(precision_bits: int) -> int:
    return (max(precision_bits, 53) + 127) // 64 + 1


# ---------------------------------------------------------------- numbers <-> exact decimals
def _record_to_decimal(size: int, exp: int, limbs: Tuple[int, ...]) -> str:
    n = abs(size)
    if n == 0:
        return "0"
    m = 0
    for i in range(n):
        m |= limbs[i] << (64 * i)
    e2 = 64 * (exp - n)            # value = m * 2^e2, exactly
    if e2 >= 0:
        digits = str(m << e2)
    else:
        k = -e2
        t = m & -m                  # strip common factors of two first
        s = min(k, t.bit_length() - 1)
        m >>= s
        k -= s
        digits = str(m * 5 ** k)
        if k:
            digits = digits.rjust(k + 1, "0")
            digits = digits[:-k] + "." + digits[-k:]
    return ("-" if size < 0 else "") + digits


def _decimal_to_record(text: str, precision_bits: int):
    """Truncating conversion (toward zero) of a decimal string to GMP's layout at precision p."""
    import mpmath
    nl = num_limbs(precision_bits)
    keep = 64 * nl
    old = mpmath.mp.prec
    mpmath.mp.prec = keep + 64
    try:
        v = mpmath.mpf(text)
    finally:
        mpmath.mp.prec = old
    if v == 0:
        return 0, 0, (0,) * nl
    sign, man, e, bc = v._mpf_
    # value = man * 2^e; limb exponent: smallest exp with |value| < 2^(64 exp)
    top = bc + e
    exp = -((-top) // 64)
    # mantissa as nl limbs just below 2^(64 exp), truncated
    shift = 64 * (exp - nl) - e     # man * 2^e = M * 2^(64 (exp - nl))  ->  M = man >> shift
    M = man >> shift if shift >= 0 else man << (-shift)
    # mpf keeps only the limbs it needs: drop low zero limbs
    n = nl
    while n and M & ((1 << 64) - 1) == 0 and M:
        M >>= 64
        n -= 1
    limbs = tuple((M >> (64 * i)) & ((1 << 64) - 1) for i in range(n)) + (0,) * (nl - n)
    return (-n if sign else n), exp, limbs


# ---------------------------------------------------------------- reader
class _In:
    def __init__(self, data: bytes):
        self.b = data
        self.o = 0

    def take(self, fmt: str):
        v = struct.unpack_from("<" + fmt, self.b, self.o)
        self.o += struct.calcsize("<" + fmt)
        return v if len(v) > 1 else v[0]

    def raw(self, n: int) -> bytes:
        r = self.b[self.o:self.o + n]
        if len(r) != n:
            raise ValueError("block_data.bin: truncated file")
        self.o += n
        return r


class BinFormatError(ValueError):
    pass


def read_block_data_bin(data: bytes, precision_bits: int, shape_hint=None) -> dict:
    """-> {"B": rows, "c": list, "bilinear_bases_even": rows, "bilinear_bases_odd": rows} with every
    number as an exact decimal string (B row-major P x N like the JSON flavour)."""
    f = _In(data)
    if f.take("Q") != len(SIGNATURE) or f.raw(len(SIGNATURE)) != SIGNATURE:
        raise BinFormatError("not a boost binary archive (signature missing)")
    f.take("H")
    if f.take("BBBB") != (4, 8, 4, 8) or f.take("i") != 1:
        raise BinFormatError("archive written on an unsupported platform (type sizes / endianness)")
    prec = f.take("Q")
    if prec != gmp_precision(precision_bits):   # SDP_Block_Data.cxx:41-43
        raise BinFormatError(f"Read GMP precision: {prec}, expected: {gmp_precision(precision_bits)}")
    nl = num_limbs(precision_bits)
    state = {"float_seen": False, "matrix_seen": False, "int": None}

    def number():
        if not state["float_seen"]:
            state["float_seen"] = True
            if f.take("B") != 0 or f.take("I") != 1:
                raise BinFormatError("unexpected class preamble of El::BigFloat (tracking, version)")
        if f.take("B"):
            return "0"
        mp_prec, size, exp = f.take("iiq")
        if mp_prec != nl - 1 or abs(size) > nl:
            raise BinFormatError(f"El::BigFloat record does not fit --precision {precision_bits} "
                                 f"(_mp_prec {mp_prec}, _mp_size {size}, {nl} limbs expected)")
        return _record_to_decimal(size, exp, f.take(f"{nl}Q") if nl > 1 else (f.take("Q"),))

    def matrix():
        if not state["matrix_seen"]:
            state["matrix_seen"] = True
            if f.take("B") != 0 or f.take("I") != 0:
                raise BinFormatError("unexpected class preamble of El::Matrix (tracking, version)")
        if state["int"] is None:     # El::Int is int unless Elemental was built with 64-bit ints
            h4 = struct.unpack_from("<iii", f.b, f.o)
            state["int"] = "i" if (0 <= h4[0] <= h4[2] and h4[1] >= 0 and h4[2] < 1 << 28) else "q"
        h, w, ld = f.take(state["int"] * 3)
        if not (0 <= h <= ld and w >= 0):
            raise BinFormatError(f"bad El::Matrix header {h} x {w}, ldim {ld}")
        cols = [[number() for _ in range(ld)][:h] for _ in range(w)]
        return [[cols[c][r] for c in range(w)] for r in range(h)]   # row-major rows

    B = matrix()
    count = f.take("Q")
    f.take("I")                       # item_version
    c = [number() for _ in range(count)]
    even = matrix()
    odd = matrix()
    if f.o != len(f.b):
        raise BinFormatError(f"{len(f.b) - f.o} trailing bytes: the file does not follow the assumed layout")
    return {"B": B, "c": c, "bilinear_bases_even": even, "bilinear_bases_odd": odd}


# ---------------------------------------------------------------- writer (round trips, converters)
def write_block_data_bin(block: dict, precision_bits: int) -> bytes:
    out = io.BytesIO()
    w = out.write
    w(struct.pack("<Q", len(SIGNATURE)) + SIGNATURE + struct.pack("<H", LIBRARY_VERSION))
    w(struct.pack("<BBBBi", 4, 8, 4, 8, 1))
    w(struct.pack("<Q", gmp_precision(precision_bits)))
    nl = num_limbs(precision_bits)
    state = {"float_seen": False, "matrix_seen": False}

    def number(text):
        if not state["float_seen"]:
            state["float_seen"] = True
            w(struct.pack("<BI", 0, 1))
        size, exp, limbs = _decimal_to_record(text, precision_bits)
        if size == 0:
            w(b"\x01")
            return
        w(b"\x00" + struct.pack(f"<iiq{nl}Q", nl - 1, size, exp, *limbs))

    def matrix(rows: List[List[str]], width=None):
        if not state["matrix_seen"]:
            state["matrix_seen"] = True
            w(struct.pack("<BI", 0, 0))
        h = len(rows)
        wd = len(rows[0]) if rows else (width or 0)
        w(struct.pack("<iii", h, wd, max(h, 1)))
        for c in range(wd):
            col = [rows[r][c] for r in range(h)] + ["0"] * (max(h, 1) - h)
            for v in col:
                number(v)

    matrix(block["B"])
    w(struct.pack("<QI", len(block["c"]), 0))
    for v in block["c"]:
        number(v)
    k = len(block["bilinear_bases_even"][0]) if block["bilinear_bases_even"] else 0
    matrix(block["bilinear_bases_even"], k)
    matrix(block["bilinear_bases_odd"], k)
    return out.getvalue()
