"""The arithmetic claims conv_x3.hip (UMT_PREC_FP32_BF16X3) rests on, checked in numpy on the CPU:

  1. an fp32 number is the EXACT sum of three bf16 numbers obtained by round-to-nearest-even splitting,
     and both remainders are exact fp32 subtractions;
  2. |x1| <= 2^-8 |x0|-ish and |x2| <= 2^-16: the three cross terms the kernel drops are <= 2^-23 |a b|;
  3. a dot product evaluated from the six kept bf16 x bf16 products with fp32 accumulation is as close to the
     float64 result as a plain fp32 fmaf chain is.
"""
import numpy as np


def bf16_rne(x):
    """fp32 -> bf16 (round to nearest even), returned as fp32 with the low 16 bits zero."""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, np.float32)
    x0 = bf16_rne(x)
    r1 = x - x0
    x1 = bf16_rne(r1)
    r2 = r1 - x1
    x2 = bf16_rne(r2)
    return x0, x1, x2, r1, r2


def _samples(n, seed):
    rng = np.random.default_rng(seed)
    mant = rng.uniform(1.0, 2.0, n).astype(np.float32)
    expo = rng.integers(-30, 30, n)
    sign = rng.choice([-1.0, 1.0], n).astype(np.float32)
    return sign * np.ldexp(mant, expo).astype(np.float32)


def test_three_bf16_pieces_sum_exactly_to_the_fp32_number():
    x = np.concatenate([_samples(200000, 1), np.float32([0.0, 1.0, -1.0, 2.0 ** -100, 1.0e38, 255.0 / 256.0, 1.0 + 2.0 ** -23])])
    x0, x1, x2, r1, r2 = split3(x)
    # the remainders are exact: computed in float64 they are the same numbers
    assert np.array_equal(r1.astype(np.float64), x.astype(np.float64) - x0.astype(np.float64))
    assert np.array_equal(r2.astype(np.float64), r1.astype(np.float64) - x1.astype(np.float64))
    # and the third piece holds all of the second remainder
    s = x0.astype(np.float64) + x1.astype(np.float64) + x2.astype(np.float64)
    assert np.array_equal(s, x.astype(np.float64))
    # below 2^-110 the low pieces underflow bf16's range (same exponent range as fp32, 16 fewer fraction bits): the sum is
    # then off by less than one bf16 denormal step, 2^-133 - twenty orders of magnitude below anything the network holds
    tiny = np.float32([3.0e-39, -1.7e-38, 2.0 ** -120 * 1.2345])
    t0, t1, t2, _, _ = split3(tiny)
    err = np.abs(t0.astype(np.float64) + t1.astype(np.float64) + t2.astype(np.float64) - tiny.astype(np.float64))
    assert np.all(err <= 2.0 ** -133)


def test_piece_magnitudes_and_dropped_cross_terms():
    a, b = _samples(100000, 2), _samples(100000, 3)
    a0, a1, a2, _, _ = split3(a)
    b0, b1, b2, _, _ = split3(b)
    f = np.float64
    assert np.all(np.abs(a1.astype(f)) <= 2.0 ** -8 * np.abs(a.astype(f)) * (1 + 2.0 ** -7))
    assert np.all(np.abs(a2.astype(f)) <= 2.0 ** -16 * np.abs(a.astype(f)) * (1 + 2.0 ** -6))
    dropped = a1.astype(f) * b2.astype(f) + a2.astype(f) * b1.astype(f) + a2.astype(f) * b2.astype(f)
    assert np.all(np.abs(dropped) <= 2.0 ** -23 * np.abs(a.astype(f) * b.astype(f)))
    # every kept product of two bf16 numbers (8-bit significands) is exact in fp32
    for p, q in ((a0, b0), (a0, b1), (a1, b0), (a0, b2), (a1, b1), (a2, b0)):
        assert np.array_equal((p * q).astype(f), p.astype(f) * q.astype(f))


def test_six_term_dot_product_is_as_accurate_as_an_fp32_chain():
    rng = np.random.default_rng(4)
    K, N = 2304, 400                      # 256 channels x 9 taps
    A = (rng.standard_normal((N, K)) * rng.uniform(0.1, 3.0, (N, 1))).astype(np.float32)
    A = np.maximum(A, 0)                  # post-ReLU activations
    B = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    truth = (A.astype(np.float64) * B.astype(np.float64)).sum(axis=1)
    chain = np.zeros(N, np.float32)
    for k in range(K):                    # fp32 multiply + add per step (an fmaf chain rounds once per step: no worse)
        chain = (chain + A[:, k] * B[:, k]).astype(np.float32)
    a = split3(A)[:3]
    b = split3(B)[:3]
    acc = np.zeros(N, np.float32)
    for k0 in range(0, K, 16):            # one matrix instruction = 16 exact products summed into the fp32 accumulator
        for s in (2, 1, 0):
            for pa in range(s + 1):
                part = (a[pa][:, k0:k0 + 16].astype(np.float64) * b[s - pa][:, k0:k0 + 16].astype(np.float64)).sum(axis=1)
                acc = (acc.astype(np.float64) + part).astype(np.float32)
    scale = np.abs(A.astype(np.float64) * B.astype(np.float64)).sum(axis=1)
    e_chain = np.abs(chain - truth) / scale
    e_split = np.abs(acc - truth) / scale
    assert e_split.max() < 2.0 ** -22
    assert np.sqrt((e_split ** 2).mean()) <= 1.5 * np.sqrt((e_chain ** 2).mean()) + 1e-9
