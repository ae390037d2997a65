"""Host side of estimate_pan_core_size() that needs no GPU: the permutations are drawn from numpy's
global legacy generator exactly as the reference draws them (pangenome_analysis.py:84-85) -- by libpgx from
the generator's MT19937 state -- and the generator is left in the state the reference leaves it in."""
import numpy as np
import pytest

from pangenomix_amd import pangenome_analysis as pa


@pytest.mark.parametrize('seed,S,n_iter', [(0, 400, 50), (3, 1, 4), (7, 2, 9), (11, 37, 200), (5, 1000, 3), (2, 65, 700)])
def test_native_shuffles_equal_numpys(seed, S, n_iter):
    np.random.seed(seed)
    np.random.random(seed % 5)                      # start somewhere inside the 624-word block
    want = pa._draw_permutations_numpy(S, n_iter)
    tail_want = np.random.random(3)
    np.random.seed(seed)
    np.random.random(seed % 5)
    got = pa.draw_permutations(S, n_iter)
    assert got.dtype == np.int32 and np.array_equal(got, want)
    assert np.array_equal(np.random.random(3), tail_want)      # the stream continues where numpy's would


def test_shuffles_across_many_regenerations():
    np.random.seed(123)
    want = pa._draw_permutations_numpy(400, 1000)              # ~ 450 k draws = 700 regenerations of the state
    state_want = np.random.get_state()
    np.random.seed(123)
    got = pa.draw_permutations(400, 1000)
    state_got = np.random.get_state()
    assert np.array_equal(got, want)
    assert np.array_equal(state_got[1], state_want[1]) and state_got[2] == state_want[2]
    assert sorted(got[17].tolist()) == list(range(400))


def test_empty_requests_leave_the_generator_alone():
    np.random.seed(9)
    before = np.random.get_state()
    assert pa.draw_permutations(0, 5).shape == (5, 0) and pa.draw_permutations(7, 0).shape == (0, 7)
    after = np.random.get_state()
    assert np.array_equal(before[1], after[1]) and before[2] == after[2]
