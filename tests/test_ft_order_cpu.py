"""The fine-tune kernels order the four keys of a pass by ranks computed from comparisons (mulut_amd/csrc/mulut_ft.hip: ft_order_code)
instead of the reference's 24-branch cascade of strict '>' tests (sr/model.py:191-282).  Gradients depend on the order at TIES, so the two
must agree for every tie pattern: checked here exhaustively on a restatement of both (the kernels themselves are held to reference-made
fixtures with integer inputs -- ties everywhere -- in tests/test_gpu_finetune.py)."""
import itertools


def cascade(fa, fb, fc, fd):
    """sr/model.py:191-282: the simplex case by strict comparisons, as the order of the key ids 0..3 (first = largest f)"""
    fab, fac, fad, fbc, fbd, fcd = fa > fb, fa > fc, fa > fd, fb > fc, fb > fd, fc > fd
    if fab and fbc:
        return (0, 1, 2, 3) if fcd else (0, 1, 3, 2) if fbd else (0, 3, 1, 2) if fad else (3, 0, 1, 2)
    if fab and fac:
        return (0, 2, 1, 3) if fbd else (0, 2, 3, 1) if fcd else (0, 3, 2, 1) if fad else (3, 0, 2, 1)
    if fab:
        return (2, 0, 1, 3) if fbd else (2, 0, 3, 1) if fad else (2, 3, 0, 1) if fcd else (3, 2, 0, 1)
    if fac:
        return (1, 0, 2, 3) if fcd else (1, 0, 3, 2) if fad else (1, 3, 0, 2) if fbd else (3, 1, 0, 2)
    if fbc:
        return (1, 2, 0, 3) if fad else (1, 2, 3, 0) if fcd else (1, 3, 2, 0) if fbd else (3, 1, 2, 0)
    return (2, 1, 0, 3) if fad else (2, 1, 3, 0) if fbd else (2, 3, 1, 0) if fcd else (3, 2, 1, 0)


def by_ranks(f):
    """ft_order_code: rank_i = #{j > i: f_j >= f_i} + #{j < i: f_j > f_i}; returns the key ids by rank"""
    lt = lambda x, y: int(x < y)      # noqa: E731  (the kernel: sign bit of the int32 difference of the float patterns)
    a, b, c, d = f
    s10, s20, s30, s21, s31, s32 = lt(b, a), lt(c, a), lt(d, a), lt(c, b), lt(d, b), lt(d, c)
    ranks = (3 - s10 - s20 - s30, s10 + 2 - s21 - s31, s20 + s21 + 1 - s32, s30 + s31 + s32)
    assert sorted(ranks) == [0, 1, 2, 3]
    code = (1 << (2 * ranks[1])) | (2 << (2 * ranks[2])) | (3 << (2 * ranks[3]))
    return tuple((code >> (2 * j)) & 3 for j in range(4))


def test_rank_order_equals_the_reference_cascade_for_every_tie_pattern():
    n = 0
    for f in itertools.product(range(4), repeat=4):      # four levels: every ordering and every tie pattern of four values
        assert by_ranks(f) == cascade(*f), f
        n += 1
    assert n == 256
    for f in itertools.product((0.0, 0.5, 7.25, 15.999), repeat=4):
        assert by_ranks(f) == cascade(*f), f
