#!/usr/bin/env python
"""Why six bf16 piece products reproduce an fp32 product (csrc/wino_mm.hip, option "wino_x6"): a numpy model of the two
matrix-pipe paths against an fp64 reference, no GPU needed.

  x = x1 + x2 + x3 with x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2) is EXACT for every finite fp32 x (3 x 8 significand
  bits), by truncation as well as by round-to-nearest.  a*b = sum_ij a_i b_j; the terms a2b3, a3b2, a3b3 are <= 2^-25 |ab| and
  are dropped; bf16 x bf16 products are exact in fp32; both pipes accumulate in fp32.

Model: native = fp32 accumulation of exact products in steps of k = 2 (v_mfma_f32_32x32x2_f32), split = fp32 accumulation in
steps of one piece-product group of k = 16 (v_mfma_f32_32x32x16_bf16), i.e. 128 vs 96 roundings for K = 256.
    python tools/split_bf16_numerics.py"""
import numpy as np


def bf16_rne(x):
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000).astype(np.uint32).view(np.float32)


def bf16_trunc(x):
    return (np.asarray(x, np.float32).view(np.uint32) & np.uint32(0xffff0000)).view(np.float32)


def split(x, f):
    a1 = f(x)
    r = (x - a1).astype(np.float32)
    a2 = f(r)
    r2 = (r - a2).astype(np.float32)
    a3 = f(r2)
    return a1, a2, a3, (r2 - a3)


TERMS6 = [(0, 2), (2, 0), (1, 1), (0, 1), (1, 0), (0, 0)]                      # smallest first, as the kernel issues them
TERMS9 = [(2, 2), (1, 2), (2, 1)] + TERMS6


def main():
    rng = np.random.default_rng(1)
    M, K, N = 1024, 256, 256
    for name, (sa, sb, positive) in (("V ~ N(0,3), U ~ N(0,0.02)", (3, 0.02, False)), ("all positive (worst case for a bias)", (1, 1, True)),
                                     ("V ~ N(0,100), U ~ N(0,1e-3)", (100, 1e-3, False))):
        A = (rng.standard_normal((M, K)) * sa).astype(np.float32)
        B = (rng.standard_normal((K, N)) * sb).astype(np.float32)
        if positive:
            A, B = np.abs(A), np.abs(B)
        ref = A.astype(np.float64) @ B.astype(np.float64)
        sc = np.abs(ref).max()
        acc = np.zeros((M, N), np.float32)
        for k in range(0, K, 2):
            acc = (acc + (A[:, k:k + 2].astype(np.float64) @ B[k:k + 2].astype(np.float64))).astype(np.float32)
        print("%s\n   native fp32 MFMA model        : max %.3e  rms %.3e   (relative to max |ref|)" % (
            name, np.abs(acc - ref).max() / sc, np.sqrt(((acc - ref) ** 2).mean()) / sc))
        a, b = split(A, bf16_trunc), split(B, bf16_rne)                    # as in the kernel: A truncated in the loader, B rounded once
        assert np.abs(a[3]).max() == 0 and np.abs(b[3]).max() == 0, "the three pieces must reproduce the operand exactly"
        for tn, T in (("6 piece products", TERMS6), ("9 piece products", TERMS9)):
            acc = np.zeros((M, N), np.float32)
            for k in range(0, K, 16):
                for i, j in T:
                    acc = (acc + (a[i][:, k:k + 16].astype(np.float64) @ b[j][k:k + 16].astype(np.float64))).astype(np.float32)
            print("   split bf16, %-18s: max %.3e  rms %.3e" % (tn, np.abs(acc - ref).max() / sc, np.sqrt(((acc - ref) ** 2).mean()) / sc))


if __name__ == "__main__":
    main()
