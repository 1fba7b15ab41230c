"""The wavefront schedule on paper: a model of what csrc/tvdn_stream.hip does per chunk -- which
rows of which iteration level a launch reads and writes, in rings of R+2 rows per level (R+K+3 for the input) -- checked
for many (rows of the cube, chunk height, depth): every row of every level is produced exactly once, every read finds the
row it expects still in its ring slot (nothing was overwritten while live), and every row goes home from the last level.
No GPU: this is the arithmetic the ring capacities rest on."""
import itertools

import pytest


def simulate(N0, R, K):
    """One pass of K levels over a cube of N0 rows in chunks of R rows.  Ring contents are tags (level, row)."""
    cap, ocap = R + 2, R + K + 3
    recon = [dict() for _ in range(K + 1)]          # level -> {slot: (level, row)}
    state = [dict() for _ in range(K + 2)]          # index level + 1 (levels -1 .. K)
    orig = {}
    produced = set()
    home = []
    n_chunks = (N0 + K + R - 1) // R
    for c in range(n_chunks):
        u0, u1 = c * R, min((c + 1) * R, N0)
        for g in range(u0, u1):                     # upload: input, recon level 0, state levels 0 and -1
            orig[g % ocap] = ("orig", g)
            recon[0][g % cap] = (0, g)
            state[1][g % cap] = (0, g)
            state[0][g % cap] = (-1, g)
        for j in range(K):                          # level j -> j + 1, trailing by one row per level
            lo, hi = max(0, c * R - (j + 1)), min(N0, (c + 1) * R - (j + 1))
            if lo >= hi:
                continue
            assert hi - lo <= R
            # reads: recon level j rows lo-1 .. hi (clipped to the cube: the faces apply the boundary rule instead)
            for g in range(max(0, lo - 1), min(N0, hi + 1)):
                assert recon[j].get(g % cap) == (j, g), ("recon", N0, R, K, c, j, g, recon[j].get(g % cap))
            # reads: accumulator state of levels j and j-1 at rows lo .. hi (the look-ahead row included)
            for g in range(lo, min(N0, hi + 1)):
                assert state[j + 1].get(g % cap) == (j, g), ("state", N0, R, K, c, j, g)
                assert state[j].get(g % cap) == (j - 1, g), ("state-prev", N0, R, K, c, j, g)
            for g in range(lo, hi):
                assert orig.get(g % ocap) == ("orig", g), ("orig", N0, R, K, c, j, g)
            # writes: recon and state of level j + 1 at rows lo .. hi - 1 (different rings from the ones read)
            for g in range(lo, hi):
                assert (j + 1, g) not in produced
                produced.add((j + 1, g))
                recon[j + 1][g % cap] = (j + 1, g)
                state[j + 2][g % cap] = (j + 1, g)
        lo, hi = max(0, c * R - K), min(N0, (c + 1) * R - K)
        for g in range(lo, hi):                     # rows that reached the last level go home: recon, d_K, d_K-1
            assert recon[K].get(g % cap) == (K, g) and state[K + 1].get(g % cap) == (K, g)
            assert state[K].get(g % cap) == (K - 1, g), ("home-prev", N0, R, K, c, g)
            home.append(g)
    assert produced == {(j, g) for j in range(1, K + 1) for g in range(N0)}
    assert home == list(range(N0))
    return n_chunks


@pytest.mark.parametrize("N0,R,K", [c for c in itertools.product((1, 2, 3, 5, 8, 17, 40), (1, 2, 3, 4, 7, 16), (1, 2, 3, 5, 9, 30))])
def test_every_row_of_every_level_once_and_never_overwritten_while_live(N0, R, K):
    simulate(N0, R, K)


def test_a_ring_one_row_shorter_would_lose_rows():
    """R + 2 is tight: with R + 1 slots the row a launch still has to read has already been overwritten."""
    import types
    src = simulate.__code__
    g = dict(simulate.__globals__)
    # same model, rings of R + 1
    import inspect
    text = inspect.getsource(simulate).replace("cap, ocap = R + 2, R + K + 3", "cap, ocap = R + 1, R + K + 3").replace("def simulate", "def shorter")
    exec(text, g)
    with pytest.raises(AssertionError):
        g["shorter"](17, 4, 5)
