"""A plain textbook Lloyd iteration for k = 2 in float64 -- test infrastructure, independent of both the library and
oracle/abw_oracle.c (which restates the library's fp32 arithmetic): two squared distances per point, the nearer centroid
wins (tie -> cluster 0), both means recomputed from their members, an empty cluster keeps its centroid.  Only the seeding
is shared with the library, because it defines which run is meant: restart r draws its two seed points with TinyMT32
(reference src/TinyMT-src-1.0.1/tinymt/tinymt32.c, parameters of reference src/misc.cpp:29-31) seeded base_seed + r:
u0 % N, then the next u % N that differs.

Array-library agnostic: `xp` is numpy (CPU tests, small sizes) or torch (float64 on the GPU box, full sizes)."""
import numpy as np


def tinymt32_stream(seed, n):
    """TinyMT32 as published (tinymt32.h:70-107, tinymt32.c:62-76), pure Python."""
    M = 0xFFFFFFFF
    mat1, mat2, tmat = 0x8f7011ee, 0xfc78ff1f, 0x3793fdff
    s = [seed & M, mat1, mat2, tmat]
    for i in range(1, 8):
        s[i & 3] ^= (i + 1812433253 * (s[(i - 1) & 3] ^ (s[(i - 1) & 3] >> 30))) & M
    if (s[0] & 0x7fffffff) == 0 and s[1] == 0 and s[2] == 0 and s[3] == 0:
        s = [ord("T"), ord("I"), ord("N"), ord("Y")]

    def next_state():
        y = s[3]
        x = (s[0] & 0x7fffffff) ^ s[1] ^ s[2]
        x ^= (x << 1) & M
        y ^= (y >> 1) ^ x
        s[0], s[1], s[2], s[3] = s[1], s[2], (x ^ ((y << 10) & M)) & M, y & M
        if y & 1:
            s[1] ^= mat1
            s[2] ^= mat2
    for _ in range(8):
        next_state()
    out = []
    for _ in range(n):
        next_state()
        t0 = s[3]
        t1 = (s[0] + (s[2] >> 8)) & M
        t0 ^= t1
        if t1 & 1:
            t0 ^= tmat
        out.append(t0 & M)
    return out


def seed_points(base_seed, r, N):
    u = tinymt32_stream((base_seed + r) & 0xFFFFFFFF, 66)
    i0 = u[0] % N
    i1 = i0
    for v in u[1:65]:
        if i1 != i0:
            break
        i1 = v % N
    return i0, i1


def lloyd(xp, x, base_seed, r, iters, block=1 << 16):
    """x: float64 [d, N].  Yields after every assignment step (iteration 0 = by the seed points, ...):
    (it, member_of_cluster1 bool[N], c0, c1, inertia) -- iters + 1 items."""
    d, N = x.shape
    i0, i1 = seed_points(base_seed, r, N)
    c0, c1 = x[:, i0] + 0, x[:, i1] + 0
    for it in range(iters + 1):
        mem_blocks, d_best = [], 0.0
        for b in range(0, N, block):
            xb = x[:, b:b + block]
            d0 = ((xb - c0[:, None]) ** 2).sum(0)
            d1 = ((xb - c1[:, None]) ** 2).sum(0)
            m = d1 < d0
            mem_blocks.append(m)
            d_best = d_best + float(xp.where(m, d1, d0).sum())
        mem = xp.concatenate(mem_blocks) if hasattr(xp, "concatenate") else xp.cat(mem_blocks)
        yield it, mem, c0, c1, d_best
        if it == iters:
            break
        n1 = int(mem.sum())
        if n1 > 0:
            c1 = (x * mem).sum(1) / n1
        if N - n1 > 0:
            c0 = (x * (~mem)).sum(1) / (N - n1)
