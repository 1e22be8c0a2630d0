#!/usr/bin/env python3
"""Emit the Taylor coefficients used by csrc/lean_math.h as exact hex-float literals.

exp2:      2^r      = sum_k (ln2)^k / k! * r^k                       |r| <= 1/2,  k = 0..13
sincospi:  sin(pi r) = r * sum_k (-1)^k pi^(2k+1)/(2k+1)! * r^(2k)   |r| <= 1/4,  k = 0..7
           cos(pi r) =     sum_k (-1)^k pi^(2k)  /(2k)!   * r^(2k)   |r| <= 1/4,  k = 0..8
Truncation errors (first dropped term, relative to the result): 2^r: (ln2/2)^14/14! = 4e-18;
sin: (pi/4)^16/17! = 6e-17 of r*pi; cos: (pi/4)^18/18! = 2e-18.  60-digit decimal arithmetic, then round-to-nearest
to float64 (float(str(Decimal)) is correctly rounded).
"""
from decimal import Decimal, getcontext

getcontext().prec = 60
PI = Decimal("3.14159265358979323846264338327950288419716939937510582097494")
LN2 = Decimal("0.693147180559945309417232121458176568075500134360255254120680")


def fact(n):
    r = Decimal(1)
    for i in range(2, n + 1):
        r *= i
    return r


def emit(name, vals):
    print("// %s" % name)
    print("    " + ", ".join(float(str(v)).hex() for v in vals))


emit("EXP2[k] = ln2^k/k!, k=0..13", [LN2 ** k / fact(k) for k in range(14)])
emit("SINPI[k] = (-1)^k pi^(2k+1)/(2k+1)!, k=0..7", [(-1) ** k * PI ** (2 * k + 1) / fact(2 * k + 1) for k in range(8)])
emit("COSPI[k] = (-1)^k pi^(2k)/(2k)!, k=0..8", [(-1) ** k * PI ** (2 * k) / fact(2 * k) for k in range(9)])
