"""Physical constants, bit-for-bit those of africanus/constants/consts.py:6-9."""
import math

__all__ = ["c", "minus_two_pi_over_c", "two_pi_over_c"]

# Lightspeed
c = 2.99792458e8

two_pi_over_c = 2 * math.pi / c
minus_two_pi_over_c = -two_pi_over_c
