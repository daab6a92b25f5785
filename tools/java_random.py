"""java.util.Random as the JDK specifies it, so that the reference's seeded test inputs
(`new Random(42)` + `nextGaussian()`) can be regenerated without a JVM.

The class documentation of java.util.Random fixes the algorithm bit for bit:
  * state: 48-bit linear congruential generator, seed scrambled with 0x5DEECE66D,
        next(bits) = (seed = (seed * 0x5DEECE66D + 0xB) mod 2^48) >>> (48 - bits)
  * nextDouble() = ((next(26) << 27) + next(27)) * 2^-53
  * nextGaussian(): the polar method of Box, Muller and Marsaglia (Knuth, TAOCP vol. 2, 3.4.1 C) with the
    second value of each pair cached:
        do { v1 = 2 nextDouble() - 1; v2 = 2 nextDouble() - 1; s = v1 v1 + v2 v2 } while (s >= 1 || s == 0)
        multiplier = StrictMath.sqrt(-2 StrictMath.log(s) / s); return v1 multiplier, keep v2 multiplier
    (java.util.Random keeps this legacy form in every JDK, 17+ included.)
StrictMath.log is fdlibm's e_log.c (ported to Java as FdLibm.Log in recent JDKs); it is restated below operation by
operation, because the C library's log behind math.log is a different implementation and may differ in the last bit.
StrictMath.sqrt is the correctly rounded square root, which math.sqrt is too.
"""
import math
import struct

_MASK48 = (1 << 48) - 1


def _hi_lo(x):
    b = struct.unpack("<Q", struct.pack("<d", x))[0]
    hi = (b >> 32) & 0xFFFFFFFF
    if hi >= 0x80000000:
        hi -= 1 << 32
    return hi, b & 0xFFFFFFFF


def _with_hi(x, hi):
    b = struct.unpack("<Q", struct.pack("<d", x))[0]
    return struct.unpack("<d", struct.pack("<Q", ((hi & 0xFFFFFFFF) << 32) | (b & 0xFFFFFFFF)))[0]


def strict_log(x):
    """fdlibm __ieee754_log (StrictMath.log) for finite x > 0, in the operation order of e_log.c."""
    ln2_hi = 6.93147180369123816490e-01
    ln2_lo = 1.90821492927058770002e-10
    two54 = 1.80143985094819840000e+16
    Lg1 = 6.666666666666735130e-01
    Lg2 = 3.999999999940941908e-01
    Lg3 = 2.857142874366239149e-01
    Lg4 = 2.222219843214978396e-01
    Lg5 = 1.818357216161805012e-01
    Lg6 = 1.531383769920937332e-01
    Lg7 = 1.479819860511658591e-01
    if not (x > 0.0) or math.isinf(x):
        raise ValueError("strict_log is restated for finite positive arguments only")
    hx, _ = _hi_lo(x)
    k = 0
    if hx < 0x00100000:  # subnormal
        k -= 54
        x *= two54
        hx, _ = _hi_lo(x)
    k += (hx >> 20) - 1023
    hx &= 0x000FFFFF
    i = (hx + 0x95F64) & 0x100000
    x = _with_hi(x, hx | (i ^ 0x3FF00000))  # normalize x or x/2
    k += i >> 20
    f = x - 1.0
    if (0x000FFFFF & (2 + hx)) < 3:  # |f| < 2**-20
        if f == 0.0:
            if k == 0:
                return 0.0
            dk = float(k)
            return dk * ln2_hi + dk * ln2_lo
        R = f * f * (0.5 - 0.33333333333333333 * f)
        if k == 0:
            return f - R
        dk = float(k)
        return dk * ln2_hi - ((R - dk * ln2_lo) - f)
    s = f / (2.0 + f)
    dk = float(k)
    z = s * s
    i = hx - 0x6147A
    w = z * z
    j = 0x6B851 - hx
    t1 = w * (Lg2 + w * (Lg4 + w * Lg6))
    t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)))
    i |= j
    R = t2 + t1
    if i > 0:
        hfsq = 0.5 * f * f
        if k == 0:
            return f - (hfsq - s * (hfsq + R))
        return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f)
    if k == 0:
        return f - s * (f - R)
    return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f)


class JavaRandom:
    def __init__(self, seed):
        self.seed = (seed ^ 0x5DEECE66D) & _MASK48
        self._have_next = False
        self._next_gaussian = 0.0

    def _next(self, bits):
        self.seed = (self.seed * 0x5DEECE66D + 0xB) & _MASK48
        return self.seed >> (48 - bits)  # callers below use 26 and 27 bits: no sign to restore

    def next_double(self):
        return ((self._next(26) << 27) + self._next(27)) * (1.0 / (1 << 53))

    def next_gaussian(self):
        if self._have_next:
            self._have_next = False
            return self._next_gaussian
        while True:
            v1 = 2 * self.next_double() - 1
            v2 = 2 * self.next_double() - 1
            s = v1 * v1 + v2 * v2
            if s < 1 and s != 0:
                break
        multiplier = math.sqrt(-2 * strict_log(s) / s)
        self._next_gaussian = v2 * multiplier
        self._have_next = True
        return v1 * multiplier
