"""`ops.kmeans_column_slabs` (host arithmetic of the data-parallel Lloyd, no GPU): how the columns of a matrix are dealt to the ranks in whole SC-KM2
segments, and that every deal satisfies what `sc_kmeans_fit_cols` checks (include/streamchat_hip.h, ABI 8)."""
import pytest

from streamchat_amd import ops

G, S = ops.KM_GROUP, ops.KM_SEGMENTS


@pytest.mark.parametrize("D", [576 * 3584, 2048 * 67 + 520, 2048 * 33, 2048 * 40, 8192, 2048 * 1000 + 8, 2048 * 32, 2048 * 31 + 1])
@pytest.mark.parametrize("world", [1, 2, 3, 4, 5, 7, 8, 16])
def test_slabs_are_whole_segments_and_cover_the_matrix(D, world):
    r = ops.kmeans_column_slabs(D, world)
    ng = (D + G - 1) // G
    sg = (ng + S - 1) // S
    live = (ng + sg - 1) // sg
    if live < world:
        assert r is None
        return
    seg_groups, slabs = r
    assert seg_groups == sg and len(slabs) == world
    assert slabs[0][0] == 0 and slabs[0][2] == 0 and slabs[-1][3] == D and slabs[-1][0] + slabs[-1][1] == S
    for (s0, c, lo, hi), nxt in zip(slabs, slabs[1:] + [None]):
        assert c > 0 and hi > lo and lo == s0 * sg * G
        ngl = (hi - lo + G - 1) // G
        assert ngl <= c * sg                                   # the library's first check
        if nxt is not None:                                    # a slab that is not the matrix's last one: whole groups, all its segments full
            assert nxt[0] == s0 + c and nxt[2] == hi and (hi - lo) % G == 0 and ngl == c * sg
    counts = [c for _, c, _, _ in slabs[:-1]]
    assert not counts or max(counts) - min(counts) <= 1       # the live segments are dealt as evenly as they go


def test_the_shipped_merge_shape_at_eight_ranks():
    sg, slabs = ops.kmeans_column_slabs(576 * 3584, 8)
    assert sg == 32 and [c for _, c, _, _ in slabs] == [4] * 8
    assert [hi - lo for _, _, lo, hi in slabs] == [262144] * 7 + [229376]
