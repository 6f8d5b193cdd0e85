#!/usr/bin/env python3
"""Development aid: print the polynomial tables of csrc/ymc_numerics.h.

Needs mpmath (present in the build container; not needed at run time).  The
tables are Taylor coefficients rounded to nearest binary64:

    sin(pi r) = r * sum_{i=0..8} S[i] r^(2i),   S[i] = (-1)^i pi^(2i+1)/(2i+1)!
    cos(pi r) =     sum_{i=0..8} C[i] r^(2i),   C[i] = (-1)^i pi^(2i)/(2i)!
    exp(r)    =     sum_{n=0..13} r^n / n!

On |r| <= 1/4 the first neglected terms are pi^19/19! r^18 (relative 1e-19 of
sin) and pi^18/18! r^18 (2e-18 absolute), i.e. far below 2^-53.
"""
import mpmath as mp

mp.mp.prec = 200


def hexd(x):
    return float(x).hex()


def main():
    print("/* sinpi */")
    for i in range(9):
        n = 2 * i + 1
        c = (-1) ** i * mp.pi ** n / mp.factorial(n)
        print(f"    {hexd(c)}, /* {mp.nstr(c, 20)} */")
    print("/* cospi */")
    for i in range(9):
        n = 2 * i
        c = (-1) ** i * mp.pi ** n / mp.factorial(n)
        print(f"    {hexd(c)}, /* {mp.nstr(c, 20)} */")
    print("/* exp */")
    for n in range(13, -1, -1):
        c = 1 / mp.factorial(n)
        print(f"    {hexd(c)}, /* 1/{n}! */")
    print("/* ln2 split, 1/ln2 */")
    ln2 = mp.log(2)
    hi = float.fromhex("0x1.62e42fee00000p-1")
    print("ln2_hi", hexd(hi), "ln2_lo", hexd(ln2 - mp.mpf(hi)), "inv_ln2", hexd(1 / ln2))
    print("exp overflow threshold", hexd(mp.log(mp.mpf(2) ** 1024)), mp.nstr(mp.log(mp.mpf(2) ** 1024), 25))


if __name__ == "__main__":
    main()
